"""Stencil objects of the MI355X build: the counterpart of what the reference obtains from
``self.compile_stencil(name, externals)`` (nonlinear/microphysics.py:79, tangent_linear/microphysics.py:92,
adjoint/microphysics.py:89, common/saturation.py:54, common/increment.py:47-49,146 under
/root/reference/src/cloudsc2_gt4py/physics/).

A stencil object is called exactly like a GT4Py `StencilObject`: keyword field arguments named as
in the gtscript signature, then ``dt=`` (or ``f=``), ``origin=``, ``domain=``, ``validate_args=``,
``exec_info=``; it returns ``None`` and writes its outputs in place.  Each call is one launch of a
hand-written gfx950 kernel through the C ABI of libcloudsc2_hip.so.  There is no CPU path: without
the library (or without a GPU) the call raises.

Error behaviour mirrors GT4Py's: argument problems found with ``validate_args=True`` raise
`ValueError` / `TypeError` before anything is launched; launch failures raise `RuntimeError`.
"""
from __future__ import annotations

import ctypes
from typing import Any, Callable, Dict, Mapping, Optional, Sequence

import torch

from . import _lib
from .params import make_params
from .storage import field_geometry

# gtscript parameter names in C-ABI order (include/cloudsc2_hip.h enums)
NL_IN = ("ap", "aph", "lu", "lude", "mfd", "mfu", "q", "qi", "ql", "qsat", "supsat", "t",
         "tnd_cml_q", "tnd_cml_qi", "tnd_cml_ql", "tnd_cml_t")
NL_OUT = ("clc", "covptot", "fhpsl", "fhpsn", "fplsl", "fplsn", "tnd_q", "tnd_qi", "tnd_ql", "tnd_t")
INC = ("aph", "ap", "q", "qsat", "t", "ql", "qi", "lude", "lu", "mfu", "mfd",
       "tnd_cml_t", "tnd_cml_q", "tnd_cml_ql", "tnd_cml_qi", "supsat")

_SFX = {torch.float64: "f64", torch.float32: "f32"}

#: scratch / bookkeeping arguments of the gtscript signatures that the native kernels keep in
#: registers (accepted for call compatibility, never touched)
_IGNORED_PREFIX = "tmp_"


def _windows_overlap(pa: int, pb: int, itemsize: int, nx: int, nlev: int, ls: int) -> bool:
    """Do two [level][column] windows (nx columns, nlev levels, level stride ls elements) that start at byte
    addresses pa / pb share an element?  Exact for windows of one allocation (column windows with lev_stride > nx
    interleave without touching); a misaligned pair is judged by its byte ranges."""
    d = pb - pa
    if d % itemsize:
        span = ((nlev - 1) * ls + nx) * itemsize
        return pa < pb + span and pb < pa + span
    q, r = divmod(d // itemsize, ls)            # d = q * ls + r, 0 <= r < ls
    return (r < nx and abs(q) < nlev) or (ls - r < nx and abs(q + 1) < nlev)


def _current_stream_ptr(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class HipStencil:
    """Base of the callable stencil objects."""

    name: str = ""
    scalar_name: str = "dt"
    nlev_offset: int = 1  # domain[2] = nz + nlev_offset

    def __init__(self, externals: Mapping[str, Any]):
        self.externals = dict(externals)
        self.params = make_params(self.externals)
        self._lib = _lib.load()  # raises if the HIP library is missing: no fallback

    # -- argument handling ---------------------------------------------------------------
    def _field_names(self) -> Sequence[str]:
        raise NotImplementedError

    def _geometry(self, fields: Mapping[str, torch.Tensor], domain, origin, validate: bool):
        first = next(iter(fields.values()))
        nx, nlev, ls = field_geometry(first)
        nz = nlev - 1
        if origin is not None and tuple(origin) != (0, 0, 0):
            raise ValueError(f"{self.name}: only origin=(0, 0, 0) is supported, got {tuple(origin)}")
        if domain is not None:
            d = tuple(int(x) for x in domain)
            if d != (nx, 1, nz + self.nlev_offset):
                raise ValueError(
                    f"{self.name}: domain {d} does not match the storages "
                    f"(expected {(nx, 1, nz + self.nlev_offset)})"
                )
        if validate:
            dev, dt = first.device, first.dtype
            if dev.type != "cuda":
                raise ValueError(f"{self.name}: fields must live on the GPU, got device {dev}")
            if dt not in _SFX:
                raise TypeError(f"{self.name}: unsupported dtype {dt}")
            for n, f in fields.items():
                if not isinstance(f, torch.Tensor):
                    raise TypeError(f"{self.name}: argument {n} is not a torch.Tensor")
                if f.device != dev or f.dtype != dt:
                    raise TypeError(f"{self.name}: argument {n} has device/dtype {f.device}/{f.dtype}, "
                                    f"expected {dev}/{dt}")
                g = field_geometry(f)
                if g != (nx, nlev, ls):
                    raise ValueError(f"{self.name}: argument {n} has (nx, nlev, lev_stride) = {g}, "
                                     f"expected {(nx, nlev, ls)}")
            self._check_disjoint(fields, nx, nlev, ls, first.element_size())
        elif nlev > 1:
            # ALWAYS checked, validate_args or not (a few microseconds): the C ABI takes ONE level stride for every field of
            # the call, and storages may carry a padded level pitch (storage.level_pitch) - a field allocated another way
            # (a dense torch tensor beside storage.zeros fields at a ragged nx) would be addressed with the wrong stride
            for n, f in fields.items():
                if f.stride(2) != ls:
                    raise ValueError(f"{self.name}: argument {n} has level stride {f.stride(2)}, the call's is {ls} - "
                                     "allocate every field of a call the same way (storage.zeros / storage.from_klayout)")
        return nx, nz, ls, first.dtype, first.device

    def _check_disjoint(self, fields: Mapping[str, torch.Tensor], nx: int, nlev: int, ls: int, itemsize: int) -> None:
        """Debug check (validate_args=True): an output of a call must not share storage with any other field of the
        same call.  The kernels stream level by level and cloudsc2_ad re-reads its inputs in the backward sweep, so an
        in-place call would silently compute from overwritten data (SURVEY.md 5 "race detection"; the reference's
        drivers never alias inputs and outputs of one call, adjoint/validation.py:149-150 rebinds TL outputs as AD
        *inputs* only)."""
        if nx == 0:
            return
        ptrs = {n: f.data_ptr() for n, f in fields.items()}
        # the O(n^2) window test (26 outputs x 72 fields for cloudsc2_ad: ~1 ms of host Python) runs once per distinct set
        # of storages; the drivers call a stencil on the same storages run after run
        key = (tuple(ptrs.items()), nx, nlev, ls, itemsize)
        seen = self.__dict__.setdefault("_disjoint_ok", set())
        if key in seen:
            return
        outs = [n for n in fields if n.startswith("out_")]
        for o in outs:
            for n, p in ptrs.items():
                if n == o or (n in outs and n < o):      # every unordered pair once
                    continue
                if _windows_overlap(ptrs[o], p, itemsize, nx, nlev, ls):
                    raise ValueError(f"{self.name}: output '{o}' overlaps '{n}' in memory - inputs and outputs of "
                                     "one call must be disjoint storages")
        if len(seen) >= 64:
            seen.clear()
        seen.add(key)

    def _collect(self, kwargs: Dict[str, Any]) -> Dict[str, torch.Tensor]:
        fields = {}
        for n in self._field_names():
            if n not in kwargs:
                raise TypeError(f"{self.name}: missing field argument '{n}'")
            fields[n] = kwargs.pop(n)
        for n in list(kwargs):
            if n.startswith(_IGNORED_PREFIX):
                kwargs.pop(n)
        return fields

    def _kvec(self, v: Any, nz: int, dtype, device, validate: bool) -> torch.Tensor:
        if validate:
            if not isinstance(v, torch.Tensor) or v.dim() != 1 or v.shape[0] < nz + 1:
                raise ValueError(f"{self.name}: in_eta must be a 1-D tensor with >= {nz + 1} entries")
            if v.dtype != dtype or v.device != device or not v.is_contiguous():
                raise TypeError(f"{self.name}: in_eta must be contiguous {dtype} on {device}")
        return v

    def __call__(self, **kwargs: Any) -> None:
        origin = kwargs.pop("origin", None)
        domain = kwargs.pop("domain", None)
        validate = bool(kwargs.pop("validate_args", True))
        exec_info: Optional[dict] = kwargs.pop("exec_info", None)
        self._validate = validate
        scalar = 0.0
        if self.scalar_name:
            if self.scalar_name not in kwargs:
                raise TypeError(f"{self.name}: missing scalar argument '{self.scalar_name}'")
            scalar = float(kwargs.pop(self.scalar_name))
        eta = kwargs.pop("in_eta", None)
        fields = self._collect(kwargs)
        if kwargs:
            raise TypeError(f"{self.name}: unexpected arguments {sorted(kwargs)}")
        nx, nz, ls, dtype, device = self._geometry(fields, domain, origin, validate)
        if eta is not None:
            eta = self._kvec(eta, nz, dtype, device, validate)
        ev = None
        if exec_info is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        with torch.cuda.device(device):
            rc = self._launch(fields, eta, scalar, nx, nz, ls, _SFX[dtype], _current_stream_ptr(device))
        _lib.check(rc, self.name)
        if ev is not None:
            ev[1].record()
            rec = exec_info.setdefault(self.name, {"ncalls": 0, "events": []})
            rec["ncalls"] += 1
            rec["events"].append(ev)

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream) -> int:
        raise NotImplementedError

    def _set_nlev(self, nz: int) -> None:
        """`NLEV` is an external of the TL / AD stencils (tangent_linear/microphysics.py:85: the grid's nz).  When the
        caller supplied one it must agree with the storages (checked with validate_args=True; otherwise the storages
        win, as they must for the kernels' bounds); when it did not, it is derived from them."""
        given = self.externals.get("NLEV")
        if given is not None and int(given) != nz and getattr(self, "_validate", True):
            raise ValueError(f"{self.name}: external NLEV={int(given)} does not match the storages (nz={nz})")
        self.params.NLEV = nz

    def _fn(self, base: str, sfx: str) -> Callable:
        return getattr(self._lib, f"cloudsc2_{base}_{sfx}")


def _ptrs(fields: Mapping[str, torch.Tensor], names: Sequence[str]):
    return _lib.ptr_array([fields[n].data_ptr() for n in names])


class Cloudsc2NLStencil(HipStencil):
    """`cloudsc2_nl` - nonlinear/_stencils/cloudsc2.py:24-60 (signature), :93-399 (body)."""

    name = "cloudsc2_nl"

    def _field_names(self):
        return tuple("in_" + n for n in NL_IN) + tuple("out_" + n for n in NL_OUT)

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError("cloudsc2_nl: missing field argument 'in_eta'")
        return self._fn("nl", sfx)(
            ctypes.byref(self.params), nx, nz, ls,
            _ptrs(fields, ["in_" + n for n in NL_IN]), eta.data_ptr(),
            _ptrs(fields, ["out_" + n for n in NL_OUT]), scalar, stream)


class Cloudsc2NLSaturationStencil(HipStencil):
    """BUILD EXTENSION `cloudsc2_nl_saturation`: `saturation` + `cloudsc2_nl` in one launch (C ABI
    `cloudsc2_nl_fused_*` with `qsat_out`).  Arguments of `cloudsc2_nl` minus `in_qsat`, plus `out_qsat`."""

    name = "cloudsc2_nl_saturation"

    def _field_names(self):
        return (tuple("in_" + n for n in NL_IN if n != "qsat") + ("out_qsat",)
                + tuple("out_" + n for n in NL_OUT))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError(f"{self.name}: missing field argument 'in_eta'")
        ins = _lib.ptr_array([0 if n == "qsat" else fields["in_" + n].data_ptr() for n in NL_IN])
        return self._fn("nl_fused", sfx)(
            ctypes.byref(self.params), nx, nz, ls, ins, None, 0.0, fields["out_qsat"].data_ptr(), eta.data_ptr(),
            _ptrs(fields, ["out_" + n for n in NL_OUT]), scalar, stream)


class Cloudsc2NLPerturbedStencil(HipStencil):
    """BUILD EXTENSION `cloudsc2_nl_perturbed`: `perturbed_state` + `cloudsc2_nl` in one launch: the inputs are
    read as in_X + f * in_X_i.  Arguments of `cloudsc2_nl` plus the 16 `in_*_i` fields and the scalar `f`."""

    name = "cloudsc2_nl_perturbed"

    def __call__(self, **kwargs: Any) -> None:
        if "f" not in kwargs:
            raise TypeError(f"{self.name}: missing scalar argument 'f'")
        self._pf = float(kwargs.pop("f"))
        super().__call__(**kwargs)

    def _field_names(self):
        return (tuple("in_" + n for n in NL_IN) + tuple("in_" + n + "_i" for n in NL_IN)
                + tuple("out_" + n for n in NL_OUT))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError(f"{self.name}: missing field argument 'in_eta'")
        return self._fn("nl_fused", sfx)(
            ctypes.byref(self.params), nx, nz, ls, _ptrs(fields, ["in_" + n for n in NL_IN]),
            _ptrs(fields, ["in_" + n + "_i" for n in NL_IN]), self._pf, None, eta.data_ptr(),
            _ptrs(fields, ["out_" + n for n in NL_OUT]), scalar, stream)


class Cloudsc2NLTaylorStencil(HipStencil):
    """BUILD EXTENSION `cloudsc2_nl_taylor`: perturbed NL run + the Taylor test's reduction in one launch (C ABI
    `cloudsc2_nl_taylor_*`).  Fields: the 16 `in_*`, the 16 `in_*_i`, the 10 unperturbed outputs `ref_*` (read-only);
    scalar `f`; `out_partials`: contiguous float64 tensor of shape (taylor_blocks(nx), 10) that receives, per
    workgroup, sum(NL(in + f in_i) - ref) for the 10 outputs in NL_OUT order.  Nothing else is written."""

    name = "cloudsc2_nl_taylor"

    def __call__(self, **kwargs: Any) -> None:
        if "f" not in kwargs or "out_partials" not in kwargs:
            raise TypeError(f"{self.name}: missing argument 'f' / 'out_partials'")
        self._pf = float(kwargs.pop("f"))
        self._partials = kwargs.pop("out_partials")
        super().__call__(**kwargs)

    def _field_names(self):
        return (tuple("in_" + n for n in NL_IN) + tuple("in_" + n + "_i" for n in NL_IN)
                + tuple("ref_" + n for n in NL_OUT))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError(f"{self.name}: missing field argument 'in_eta'")
        part = self._partials
        need = taylor_blocks(nx)
        if (not isinstance(part, torch.Tensor) or part.dtype != torch.float64 or not part.is_contiguous()
                or part.device != fields["in_ap"].device or part.numel() < need * len(NL_OUT)):
            raise ValueError(f"{self.name}: out_partials must be a contiguous float64 device tensor with >= "
                             f"{need} x {len(NL_OUT)} elements")
        return self._fn("nl_taylor", sfx)(
            ctypes.byref(self.params), nx, nz, ls, _ptrs(fields, ["in_" + n for n in NL_IN]),
            _ptrs(fields, ["in_" + n + "_i" for n in NL_IN]), self._pf, eta.data_ptr(),
            _ptrs(fields, ["ref_" + n for n in NL_OUT]), part.data_ptr(), scalar, stream)


class Cloudsc2NLTaylorMultiStencil(HipStencil):
    """BUILD EXTENSION `cloudsc2_nl_taylor_multi`: the Taylor test's perturbed NL runs for ALL step sizes (C ABI
    `cloudsc2_nl_taylor_multi_*`: up to 5 step sizes share one pass over the 42 words of a level).  Fields as
    `cloudsc2_nl_taylor`; `fs`: the step sizes; `out_partials`: contiguous float64 tensor of shape
    (taylor_blocks(nx), len(fs), 10) receiving, per workgroup and step size, sum(NL(in + f in_i) - ref) in NL_OUT order.
    With `f_inc=<factor>` instead of the 16 `in_*_i` fields, `state_increment` is fused in as well: the increments are formed
    in the kernel as f_inc * in (external IGNORE_SUPSAT zeroes the supsat increment)."""

    name = "cloudsc2_nl_taylor_multi"

    def __call__(self, **kwargs: Any) -> None:
        if "fs" not in kwargs or "out_partials" not in kwargs:
            raise TypeError(f"{self.name}: missing argument 'fs' / 'out_partials'")
        self._fs = [float(x) for x in kwargs.pop("fs")]
        self._partials = kwargs.pop("out_partials")
        self._f_inc = kwargs.pop("f_inc", None)
        super().__call__(**kwargs)

    def _field_names(self):
        incs = () if getattr(self, "_f_inc", None) is not None else tuple("in_" + n + "_i" for n in NL_IN)
        return tuple("in_" + n for n in NL_IN) + incs + tuple("ref_" + n for n in NL_OUT)

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError(f"{self.name}: missing field argument 'in_eta'")
        part, nf = self._partials, len(self._fs)
        need = taylor_blocks(nx) * nf * len(NL_OUT)
        if (not isinstance(part, torch.Tensor) or part.dtype != torch.float64 or not part.is_contiguous()
                or part.device != fields["in_ap"].device or part.numel() < need):
            raise ValueError(f"{self.name}: out_partials must be a contiguous float64 device tensor with >= {need} elements")
        pf = (ctypes.c_double * max(nf, 1))(*self._fs)
        fused_inc = self._f_inc is not None
        return self._fn("nl_taylor_multi", sfx)(
            ctypes.byref(self.params), nx, nz, ls, _ptrs(fields, ["in_" + n for n in NL_IN]),
            None if fused_inc else _ptrs(fields, ["in_" + n + "_i" for n in NL_IN]),
            float(self._f_inc) if fused_inc else 0.0, nf, pf, eta.data_ptr(),
            _ptrs(fields, ["ref_" + n for n in NL_OUT]), part.data_ptr(), scalar, stream)


def taylor_blocks(nx: int) -> int:
    """Number of per-workgroup partial rows `cloudsc2_nl_taylor` writes for nx columns."""
    return int(_lib.load().cloudsc2_nl_taylor_blocks(int(nx)))


class Cloudsc2TLStencil(HipStencil):
    """`cloudsc2_tl` - tangent_linear/_stencils/cloudsc2.py:23-90 (signature), :124-774 (body)."""

    name = "cloudsc2_tl"

    def _field_names(self):
        return (tuple("in_" + n for n in NL_IN) + tuple("in_" + n + "_i" for n in NL_IN)
                + tuple("out_" + n for n in NL_OUT) + tuple("out_" + n + "_i" for n in NL_OUT))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError("cloudsc2_tl: missing field argument 'in_eta'")
        self._set_nlev(nz)
        return self._fn("tl", sfx)(
            ctypes.byref(self.params), nx, nz, ls,
            _ptrs(fields, ["in_" + n for n in NL_IN]), _ptrs(fields, ["in_" + n + "_i" for n in NL_IN]),
            eta.data_ptr(),
            _ptrs(fields, ["out_" + n for n in NL_OUT]), _ptrs(fields, ["out_" + n + "_i" for n in NL_OUT]),
            scalar, stream)


class Cloudsc2TLIncrementedStencil(HipStencil):
    """BUILD EXTENSION `cloudsc2_tl_incremented`: `state_increment` + `cloudsc2_tl` in one launch (C ABI
    `cloudsc2_tl_incremented_*`): the perturbations are formed in the kernel as f * in_X (external IGNORE_SUPSAT: the
    supsat perturbation is 0).  Arguments of `cloudsc2_tl` minus the 16 `in_*_i` fields, plus the scalar `f`."""

    name = "cloudsc2_tl_incremented"

    def __call__(self, **kwargs: Any) -> None:
        if "f" not in kwargs:
            raise TypeError(f"{self.name}: missing scalar argument 'f'")
        self._f = float(kwargs.pop("f"))
        super().__call__(**kwargs)

    def _field_names(self):
        return (tuple("in_" + n for n in NL_IN) + tuple("out_" + n for n in NL_OUT)
                + tuple("out_" + n + "_i" for n in NL_OUT))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError(f"{self.name}: missing field argument 'in_eta'")
        self._set_nlev(nz)
        return self._fn("tl_incremented", sfx)(
            ctypes.byref(self.params), nx, nz, ls, _ptrs(fields, ["in_" + n for n in NL_IN]), self._f, eta.data_ptr(),
            _ptrs(fields, ["out_" + n for n in NL_OUT]), _ptrs(fields, ["out_" + n + "_i" for n in NL_OUT]),
            scalar, stream)


class Cloudsc2ADStencil(HipStencil):
    """`cloudsc2_ad` - adjoint/_stencils/cloudsc2.py:24-90 (signature), :124-996 (body)."""

    name = "cloudsc2_ad"

    def _field_names(self):
        return (tuple("in_" + n for n in NL_IN) + tuple("in_" + n + "_i" for n in NL_OUT)
                + tuple("out_" + n for n in NL_OUT) + tuple("out_" + n + "_i" for n in NL_IN))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError("cloudsc2_ad: missing field argument 'in_eta'")
        self._set_nlev(nz)
        return self._fn("ad", sfx)(
            ctypes.byref(self.params), nx, nz, ls,
            _ptrs(fields, ["in_" + n for n in NL_IN]), _ptrs(fields, ["in_" + n + "_i" for n in NL_OUT]),
            eta.data_ptr(),
            _ptrs(fields, ["out_" + n for n in NL_OUT]), _ptrs(fields, ["out_" + n + "_i" for n in NL_IN]),
            scalar, stream)


class Cloudsc2ADFromTrajectoryStencil(HipStencil):
    """BUILD EXTENSION `cloudsc2_ad_from_trajectory`: `cloudsc2_ad` without its forward sweep (C ABI
    `cloudsc2_ad_from_trajectory_*`).  Fields of `cloudsc2_ad` minus the ten `out_*` NL outputs, plus `traj_fplsl` /
    `traj_fplsn`: the flux outputs of a cloudsc2_nl / cloudsc2_tl call on the same state (read-only).  Only the 16
    `out_*_i` adjoints are written.  Driver switches only (no evaporation block)."""

    name = "cloudsc2_ad_from_trajectory"

    def _field_names(self):
        return (tuple("in_" + n for n in NL_IN) + tuple("in_" + n + "_i" for n in NL_OUT) + ("traj_fplsl", "traj_fplsn")
                + tuple("out_" + n + "_i" for n in NL_IN))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        if eta is None:
            raise TypeError(f"{self.name}: missing field argument 'in_eta'")
        self._set_nlev(nz)
        return self._fn("ad_from_trajectory", sfx)(
            ctypes.byref(self.params), nx, nz, ls,
            _ptrs(fields, ["in_" + n for n in NL_IN]), _ptrs(fields, ["in_" + n + "_i" for n in NL_OUT]),
            eta.data_ptr(), fields["traj_fplsl"].data_ptr(), fields["traj_fplsn"].data_ptr(),
            _ptrs(fields, ["out_" + n + "_i" for n in NL_IN]), scalar, stream)


class SaturationStencil(HipStencil):
    """`saturation` - common/_stencils/saturation.py:23-42; domain (nx, 1, nz)."""

    name = "saturation"
    scalar_name = ""
    nlev_offset = 0

    def _field_names(self):
        return ("in_ap", "in_t", "out_qsat")

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        return self._fn("saturation", sfx)(
            ctypes.byref(self.params), nx, nz, ls,
            fields["in_ap"].data_ptr(), fields["in_t"].data_ptr(), fields["out_qsat"].data_ptr(), stream)


class StateIncrementStencil(HipStencil):
    """`state_increment` - common/_stencils/state_increment.py:22-80."""

    name = "state_increment"
    scalar_name = "f"

    def _field_names(self):
        return tuple("in_" + n for n in INC) + tuple("out_" + n + "_i" for n in INC)

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        return self._fn("state_increment", sfx)(
            ctypes.byref(self.params), nx, nz, ls,
            _ptrs(fields, ["in_" + n for n in INC]), _ptrs(fields, ["out_" + n + "_i" for n in INC]),
            scalar, stream)


class PerturbedStateStencil(HipStencil):
    """`perturbed_state` - common/_stencils/perturbed_state.py:22-91."""

    name = "perturbed_state"
    scalar_name = "f"

    def _field_names(self):
        return (tuple("in_" + n for n in INC) + tuple("in_" + n + "_i" for n in INC)
                + tuple("out_" + n for n in INC))

    def _launch(self, fields, eta, scalar, nx, nz, ls, sfx, stream):
        return self._fn("perturbed_state", sfx)(
            ctypes.byref(self.params), nx, nz, ls,
            _ptrs(fields, ["in_" + n for n in INC]), _ptrs(fields, ["in_" + n + "_i" for n in INC]),
            _ptrs(fields, ["out_" + n for n in INC]), scalar, stream)


#: registry keyed by the names the reference registers with `@stencil_collection(name)`
STENCILS: Dict[str, type] = {
    "cloudsc2_nl": Cloudsc2NLStencil,
    "cloudsc2_nl_saturation": Cloudsc2NLSaturationStencil,   # build extension (fused)
    "cloudsc2_nl_perturbed": Cloudsc2NLPerturbedStencil,     # build extension (fused)
    "cloudsc2_nl_taylor": Cloudsc2NLTaylorStencil,           # build extension (fused + reduction)
    "cloudsc2_nl_taylor_multi": Cloudsc2NLTaylorMultiStencil,   # build extension (all step sizes, fused + reduction)
    "cloudsc2_tl": Cloudsc2TLStencil,
    "cloudsc2_tl_incremented": Cloudsc2TLIncrementedStencil,    # build extension (state_increment fused in)
    "cloudsc2_ad": Cloudsc2ADStencil,
    "cloudsc2_ad_from_trajectory": Cloudsc2ADFromTrajectoryStencil,   # build extension (no forward sweep)
    "saturation": SaturationStencil,
    "state_increment": StateIncrementStencil,
    "perturbed_state": PerturbedStateStencil,
}


def compile_stencil(name: str, externals: Optional[Mapping[str, Any]] = None) -> HipStencil:
    """Counterpart of the components' ``self.compile_stencil(name, externals)``.

    Nothing is compiled at run time: the externals select a prebuilt kernel instantiation
    (boolean switches) and fill the `Cloudsc2Params` struct (numeric values); unknown externals are
    ignored, as GT4Py ignores externals a stencil does not import."""
    try:
        cls = STENCILS[name]
    except KeyError:
        raise KeyError(f"unknown stencil '{name}'; available: {sorted(STENCILS)}") from None
    return cls(externals or {})


def finalize_exec_info(exec_info: Optional[dict]) -> None:
    """Resolve the recorded HIP events into `total_run_time` (seconds) per stencil."""
    if not exec_info:
        return
    if not any(isinstance(r, dict) and r.get("events") for r in exec_info.values()):
        return
    torch.cuda.synchronize()
    for rec in exec_info.values():
        if isinstance(rec, dict) and "events" in rec:
            tot = rec.get("total_run_time", 0.0)
            for a, b in rec["events"]:
                tot += a.elapsed_time(b) * 1e-3
            rec["total_run_time"] = tot
            rec["events"] = []
