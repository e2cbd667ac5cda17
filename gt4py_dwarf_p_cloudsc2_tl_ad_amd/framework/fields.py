"""DataArray-like field wrapper and allocation helpers (sympl-style `DataArray`s in the reference:
`.data` is indexed `[:, 0, :]` by the harnesses, tangent_linear/validation.py:243-249)."""
from __future__ import annotations

import contextlib
from typing import Any, Iterator, Sequence, Tuple

import numpy as np
import torch

from .. import storage
from .backends import backend_device
from .grid import DimSymbol


class FieldTensor(torch.Tensor):
    """torch.Tensor whose global reductions answer NumPy's calling convention.

    The reference's harnesses apply NumPy functions to `.data` slices, e.g.
    `np.abs(np.sum(field_nl_p - field_nl))` (tangent_linear/validation.py:253-261).  `np.sum(x)` calls
    `x.sum(axis=None, dtype=None, out=None)`: here that runs the reduction ON THE DEVICE (fp64
    accumulation) and returns a Python float, so a Taylor-test norm moves 8 bytes over PCIe instead of
    a 72 MB field."""

    def sum(self, *args, axis=None, dtype=None, out=None, keepdims=False, **kwargs):
        base = self.as_subclass(torch.Tensor)
        if args or kwargs or axis is not None or keepdims or out is not None:
            if axis is not None:
                kwargs["dim"] = axis
            if keepdims:
                kwargs["keepdim"] = True
            return base.sum(*args, **kwargs)
        return float(base.sum(dtype=torch.float64))


class DataArray:
    """Minimal stand-in for the sympl `DataArray`: `.data` (torch tensor), `.dims`, `.attrs['units']`."""

    def __init__(self, data: torch.Tensor, dims: Tuple[DimSymbol, ...], units: str = ""):
        if isinstance(data, torch.Tensor) and not isinstance(data, FieldTensor):
            data = data.as_subclass(FieldTensor)
        self.data = data
        self.dims = tuple(dims)
        self.attrs = {"units": units}

    @property
    def shape(self):
        return tuple(self.data.shape)

    def __repr__(self) -> str:
        return f"DataArray(dims={self.dims}, shape={self.shape}, units={self.attrs['units']!r})"


def _dtype(gt4py_config, dtype_name: str):
    return getattr(gt4py_config.dtypes, dtype_name)


def allocate(computational_grid, dims: Sequence[DimSymbol], gt4py_config, dtype_name: str = "float") -> torch.Tensor:
    """Zero storage for `dims`.  3-D fields ALWAYS get nz+1 levels (full-level fields carry a zero
    padding level, as in the reference where every kernel runs on domain (nx, 1, nz+1))."""
    device = backend_device(gt4py_config)
    dt = _dtype(gt4py_config, dtype_name)
    nx, nz = computational_grid.nx, computational_grid.nz
    names = [d.name for d in dims]
    if names == ["I", "J", "K"]:
        return storage.zeros(nx, nz, dt, device)
    if names == ["K"]:
        return storage.zeros_k(nz, dt, device)
    if names == ["I", "J"]:
        return torch.zeros((nx, 1), dtype=storage.torch_dtype(dt), device=device)
    raise ValueError(f"unsupported grid dims {tuple(dims)}")


def gt_zeros(computational_grid, dims, *, gt4py_config, dtype_name: str = "float") -> torch.Tensor:
    return allocate(computational_grid, dims, gt4py_config, dtype_name)


@contextlib.contextmanager
def managed_temporary_storage(computational_grid, *specs, gt4py_config) -> Iterator[Tuple[Any, ...]]:
    """The reference hands 2-D scratch fields to its stencils (nonlinear/microphysics.py:131-133).  The
    native kernels keep that state in registers, so no memory is allocated: placeholders are yielded
    and the stencil objects ignore every `tmp_*` argument."""
    yield tuple(None for _ in specs)


def assign(lhs, rhs) -> None:
    """`assign(storage[:], array)` (tangent_linear/microphysics.py:67-71)."""
    if isinstance(lhs, torch.Tensor):
        lhs.copy_(torch.as_tensor(np.asarray(rhs), dtype=lhs.dtype))
    else:
        lhs[...] = rhs


def to_numpy(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)
