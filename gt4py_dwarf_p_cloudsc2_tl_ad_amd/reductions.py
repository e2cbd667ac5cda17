"""Validation-norm reductions on the device (C ABI `cloudsc2_field_sums_*`, `cloudsc2_column_dots_*`).

The reference's harnesses reduce whole fields on the host with NumPy - `TaylorTest.get_field_norm`
(/root/reference/src/cloudsc2_gt4py/physics/tangent_linear/validation.py:250-261: `np.sum(field_nl_p - field_nl)`,
`np.sum(field_tl)`) and `SymmetryTest.get_norm1 / get_norm2` (adjoint/validation.py:167-215: per-column sums over levels
of products of fields).  Here ONE kernel launch handles up to 16 fields / pairs (with torch ops the Taylor test alone
spent 200 small launches and as many temporaries per run).  Same arithmetic: the difference is formed in the field type,
the accumulation is in double; per-workgroup partial sums are added in a fixed order (deterministic, no atomics).

Host tensors (the test-only oracle backend of the CPU suite) take the equivalent torch expressions; GPU tensors always go
through the library - there is no silent fallback for them."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from .storage import field_geometry

MAX_FIELDS = 16
_SFX = {torch.float64: "f64", torch.float32: "f32"}


def _plain(t: torch.Tensor) -> torch.Tensor:
    return t.as_subclass(torch.Tensor)


def _geometry(fields: Sequence[torch.Tensor], what: str):
    first = fields[0]
    geo = field_geometry(first)
    for f in fields[1:]:
        if field_geometry(f) != geo or f.dtype != first.dtype or f.device != first.device:
            raise ValueError(f"{what}: all fields must share (nx, nlev, lev_stride), dtype and device")
    if first.dtype not in _SFX:
        raise TypeError(f"{what}: unsupported dtype {first.dtype}")
    return geo


def field_sums(a: Sequence[torch.Tensor], b: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """float64 device vector: entry f = sum over all nz+1 levels and all columns of a[f] - b[f] (b is None: of a[f])."""
    a = [_plain(x) for x in a]
    b = None if b is None else [_plain(x) for x in b]
    if not 1 <= len(a) <= MAX_FIELDS or (b is not None and len(b) != len(a)):
        raise ValueError(f"field_sums: 1..{MAX_FIELDS} fields, as many subtrahends as minuends")
    nx, nlev, ls = _geometry(a + (b or []), "field_sums")
    first = a[0]
    if not first.is_cuda:      # host tensors: the test-only oracle backend
        return torch.stack([(x if b is None else x - y).sum(dtype=torch.float64) for x, y in zip(a, b or a)])
    lib = _lib.load()
    nf = len(a)
    blocks = int(lib.cloudsc2_field_sums_blocks(nx, nlev))
    part = torch.empty((blocks, nf), dtype=torch.float64, device=first.device)
    with torch.cuda.device(first.device):
        rc = getattr(lib, "cloudsc2_field_sums_" + _SFX[first.dtype])(
            nx, nlev, ls, nf, _lib.ptr_array([x.data_ptr() for x in a]),
            None if b is None else _lib.ptr_array([x.data_ptr() for x in b]), part.data_ptr(),
            int(torch.cuda.current_stream(first.device).cuda_stream))
    _lib.check(rc, "field_sums")
    return part.sum(dim=0)


def column_dots(a: Sequence[torch.Tensor], b: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """float64 device vector of nx entries: entry c = sum over pairs p and all nz+1 levels of a[p][c, 0, k] * b[p][c, 0, k]
    (b is None: squares), the products formed in double."""
    a = [_plain(x) for x in a]
    b = a if b is None else [_plain(x) for x in b]
    if not a or len(a) != len(b):
        raise ValueError("column_dots: as many left as right factors, at least one pair")
    nx, nlev, ls = _geometry(a + b, "column_dots")
    first = a[0]
    if not first.is_cuda:
        out = None
        for x, y in zip(a, b):
            s = (x[:, 0, :].to(torch.float64) * y[:, 0, :].to(torch.float64)).sum(dim=1)
            out = s if out is None else out + s
        return out
    lib = _lib.load()
    out = torch.empty((int(lib.cloudsc2_column_dots_chunks(nlev)), nx), dtype=torch.float64, device=first.device)
    fn = getattr(lib, "cloudsc2_column_dots_" + _SFX[first.dtype])
    stream = int(torch.cuda.current_stream(first.device).cuda_stream)
    with torch.cuda.device(first.device):
        for i in range(0, len(a), MAX_FIELDS):
            pa, pb = a[i:i + MAX_FIELDS], b[i:i + MAX_FIELDS]
            rc = fn(nx, nlev, ls, len(pa), _lib.ptr_array([x.data_ptr() for x in pa]),
                    _lib.ptr_array([x.data_ptr() for x in pb]), out.data_ptr(), 1 if i else 0, stream)
            _lib.check(rc, "column_dots")
    return out.sum(dim=0)          # level-chunk partials, fixed order
