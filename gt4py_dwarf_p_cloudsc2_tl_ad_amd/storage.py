"""Field storages of the MI355X build: PyTorch-ROCm tensors, physical layout [level][column].

The reference's fields are GT4Py storages of logical shape ``(nx, ny, nz+1)`` whose ``.data`` the
harnesses index as ``[:, 0, :]`` (/root/reference/src/cloudsc2_gt4py/physics/tangent_linear/validation.py:243-249,
adjoint/validation.py:217-220).  Here a field is ONE contiguous ``(nz+1, nx)`` allocation in HBM
(column index fastest, so a wave64 reading one level of 64 adjacent columns issues one coalesced
request) exposed as the permuted view ``(nx, 1, nz+1)`` - the same indexing works unchanged.

All 3-D storages have nz+1 levels, as in the reference (every kernel runs on
``domain=(nx, 1, nz+1)``, nonlinear/microphysics.py:168-169); K-vectors (`f_eta`, `klevel`) have nz+1
entries too.
"""
from __future__ import annotations

from typing import Any, Tuple

import numpy as np
import torch

_TORCH_DTYPES = {
    np.dtype("float64"): torch.float64,
    np.dtype("float32"): torch.float32,
    np.dtype("int64"): torch.int64,
    np.dtype("int32"): torch.int32,
}


def torch_dtype(dtype: Any) -> torch.dtype:
    if isinstance(dtype, torch.dtype):
        return dtype
    return _TORCH_DTYPES[np.dtype(dtype)]


def logical_view(kc: torch.Tensor) -> torch.Tensor:
    """(nz+1, nx) physical tensor -> (nx, 1, nz+1) logical view (no copy)."""
    if kc.dim() != 2:
        raise ValueError(f"expected a 2-D [level][column] tensor, got shape {tuple(kc.shape)}")
    return kc.unsqueeze(1).permute(2, 1, 0)


def klayout(field: torch.Tensor) -> torch.Tensor:
    """(nx, 1, nz+1) logical view -> (nz+1, nx) physical view (no copy)."""
    if field.dim() != 3 or field.shape[1] != 1:
        raise ValueError(f"expected a (nx, 1, nz+1) field, got shape {tuple(field.shape)}")
    return field.permute(2, 1, 0).squeeze(1)


def zeros(nx: int, nz: int, dtype: Any, device: Any) -> torch.Tensor:
    """Zero-initialised 3-D field, logical shape (nx, 1, nz+1)."""
    return logical_view(torch.zeros((nz + 1, nx), dtype=torch_dtype(dtype), device=device))


def zeros_k(nz: int, dtype: Any, device: Any) -> torch.Tensor:
    """Zero-initialised K-vector with nz+1 entries."""
    return torch.zeros((nz + 1,), dtype=torch_dtype(dtype), device=device)


def from_klayout(array_kc: Any, dtype: Any, device: Any) -> torch.Tensor:
    """Copy a host/device ``[level][column]`` array into a new field storage."""
    t = torch.as_tensor(array_kc)
    t = t.to(device=device, dtype=torch_dtype(dtype)).contiguous()
    return logical_view(t)


def field_geometry(field: torch.Tensor) -> Tuple[int, int, int]:
    """(nx, nlev, lev_stride) of a logical (nx, 1, nlev) field; raises if it is not column-fastest."""
    if field.dim() != 3 or field.shape[1] != 1:
        raise ValueError(f"field must have logical shape (nx, 1, nz+1), got {tuple(field.shape)}")
    nx, _, nlev = field.shape
    if nx > 1 and field.stride(0) != 1:
        raise ValueError(
            f"field is not column-fastest (stride over columns = {field.stride(0)}); "
            "allocate it with storage.zeros / storage.from_klayout"
        )
    ls = field.stride(2) if nlev > 1 else nx
    return nx, nlev, ls
