"""Deterministic synthetic atmospheric columns for CLOUDSC2 (stand-in for `data/input.h5`).

The reference reads its 100 input columns, the timestep and every physical parameter from
`data/input.h5` (/root/reference/src/cloudsc2_gt4py/setup.py:47-70,
/root/reference/src/cloudsc2_gt4py/iox.py:212-244), which is not shipped with the reference
checkout (/root/reference/.MISSING_LARGE_BLOBS:1).  This module generates columns with the same
field set, layout and value ranges so that every branch of the NL/TL/AD stencils is exercised
(warm rain, melting snow, all-ice columns, clear / partial / overcast cloud, convective
detrainment with and without updraught condensate below, supersaturation).

Properties the rest of the code relies on:
  * every value is a pure function of (seed, field, level, GLOBAL column index): a rank that owns
    columns [c0, c0+n) generates exactly the slice of the global problem, and global column 0
    (the one `EtaLevels` reads, common/diagnostics.py:42-45) is the same for every shard count;
  * the same code runs on NumPy arrays and on torch tensors (CPU or the GPU), so the bench can
    build 4M-column states directly in HBM;
  * fields come out in the build's physical layout ``[level][column]`` with ``nz+1`` levels; the
    padding level of full-level fields is 0 (NL relies on ``lu[nz] < ZEPS2``,
    nonlinear/_stencils/cloudsc2.py:212).
"""
from __future__ import annotations

import math
from typing import Any, Dict, Mapping, Optional

import numpy as np

from .params import default_externals

SEED = 20240807

#: state fields read by `get_state` (setup.py:48-65) that the stencils actually use
STATE_FIELDS = (
    "f_ap", "f_aph", "f_lu", "f_lude", "f_mfd", "f_mfu", "f_q", "f_qi", "f_ql",
    "f_supsat", "f_t", "f_tnd_cml_q", "f_tnd_cml_qi", "f_tnd_cml_ql", "f_tnd_cml_t",
)
_FIELD_ID = {name: i + 1 for i, name in enumerate(STATE_FIELDS)}
_M32 = 0xFFFFFFFF


class _NP:
    """Minimal array namespace over numpy."""

    pi = math.pi

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)

    def arange_i64(self, lo, hi):
        return np.arange(lo, hi, dtype=np.int64)

    def to_real(self, x):
        return x.astype(np.float64)

    def cast(self, x):
        return np.ascontiguousarray(x.astype(self.dtype))

    exp = staticmethod(np.exp)
    log = staticmethod(np.log)
    sqrt = staticmethod(np.sqrt)
    cos = staticmethod(np.cos)
    where = staticmethod(np.where)
    minimum = staticmethod(np.minimum)
    maximum = staticmethod(np.maximum)

    def zeros_like(self, x):
        return np.zeros_like(x)


class _Torch:
    pi = math.pi

    def __init__(self, dtype, device):
        import torch

        self.torch = torch
        self.dtype = {np.dtype("float64"): torch.float64, np.dtype("float32"): torch.float32}[
            np.dtype(dtype)
        ]
        self.device = device

    def arange_i64(self, lo, hi):
        return self.torch.arange(lo, hi, dtype=self.torch.int64, device=self.device)

    def to_real(self, x):
        return x.to(self.torch.float64)

    def cast(self, x):
        return x.to(self.dtype).contiguous()

    def exp(self, x):
        return self.torch.exp(x)

    def log(self, x):
        return self.torch.log(x)

    def sqrt(self, x):
        return self.torch.sqrt(x)

    def cos(self, x):
        return self.torch.cos(x)

    def where(self, c, a, b):
        t = self.torch
        if not t.is_tensor(a):
            a = t.as_tensor(a, dtype=t.float64, device=self.device)
        if not t.is_tensor(b):
            b = t.as_tensor(b, dtype=t.float64, device=self.device)
        return t.where(c, a, b)

    def minimum(self, a, b):
        t = self.torch
        if not t.is_tensor(b):
            return t.clamp(a, max=b)
        if not t.is_tensor(a):
            return t.clamp(b, max=a)
        return t.minimum(a, b)

    def maximum(self, a, b):
        t = self.torch
        if not t.is_tensor(b):
            return t.clamp(a, min=b)
        if not t.is_tensor(a):
            return t.clamp(b, min=a)
        return t.maximum(a, b)

    def zeros_like(self, x):
        return self.torch.zeros_like(x)


def _hash_u01(xp, field_id: int, salt: int, kk, cc, seed: int):
    """murmur3-fmix32 of (seed, field, salt, level, column) -> uniform in (0, 1).

    Integer arithmetic is done in int64 with explicit 32-bit masking, which is bit-identical in
    NumPy and torch (wrap-around of the 64-bit product does not touch the low 32 bits)."""
    x = (cc * 0x9E3779B1 + kk * 0x85EBCA77 + (field_id * 0xC2B2AE3D + salt * 0x27D4EB2F + seed)) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    x = x ^ (x >> 16)
    return (xp.to_real(x) + 0.5) * (1.0 / 4294967296.0)


def sigma_half_levels(nz: int) -> np.ndarray:
    """Fixed smooth sigma grid: sigma[0] = 0 (model top) ... sigma[nz] = 1 (surface)."""
    x = np.arange(nz + 1, dtype=np.float64) / nz
    return x**2.2


def make_state(
    nx: int,
    nz: int = 137,
    *,
    col0: int = 0,
    ncols: Optional[int] = None,
    dtype: Any = np.float64,
    device: Optional[Any] = None,
    seed: int = SEED,
    externals: Optional[Mapping[str, Any]] = None,
    regime: str = "mixed",
) -> Dict[str, Any]:
    """Columns [col0, col0+ncols) of the global nx-column synthetic problem.

    Returns ``{name: array[(nz+1), ncols]}`` for the 15 `STATE_FIELDS` (NumPy arrays when
    `device` is None, torch tensors on `device` otherwise).  `f_qsat` and `f_eta` are *not*
    produced here: they come from the saturation / eta-level operators, as in the drivers
    (/root/reference/drivers/run_nonlinear.py:76-94).

    `regime`: "mixed" (default) spans warm-rain, melting and all-ice columns - used by every parity
    test; "cold" keeps the surface below freezing (243..268 K), i.e. the regime of the reference's own
    100-column sample as its golden outputs show it (snow only: no rain, no melting - SURVEY.md F6) -
    used as the stand-in for `data/input.h5` by the drivers."""
    if regime not in ("mixed", "cold"):
        raise ValueError(f"unknown regime {regime!r}")
    if ncols is None:
        ncols = nx - col0
    if col0 < 0 or ncols < 0 or col0 + ncols > nx:
        raise ValueError(f"column range [{col0}, {col0 + ncols}) outside [0, {nx})")
    ext = dict(default_externals())
    if externals:
        ext.update(externals)
    xp = _NP(dtype) if device is None else _Torch(dtype, device)

    cols = xp.arange_i64(col0, col0 + ncols)
    levs = xp.arange_i64(0, nz + 1)
    cc = cols[None, :]
    kk = levs[:, None]

    def u(name, salt=0):
        return _hash_u01(xp, _FIELD_ID[name], salt, kk, cc, seed)

    def ucol(name, salt=0):
        return _hash_u01(xp, _FIELD_ID[name], salt, kk[:1] * 0 + 1000003, cc, seed)

    def normal(name, salt=0):
        return xp.sqrt(-2.0 * xp.log(u(name, salt))) * xp.cos(2.0 * xp.pi * u(name, salt + 1))

    sig_h_np = sigma_half_levels(nz)
    sig_f_np = np.concatenate([0.5 * (sig_h_np[:-1] + sig_h_np[1:]), [0.0]])
    if device is None:
        sig_h = sig_h_np[:, None]
        sig_f = sig_f_np[:, None]
    else:
        t = xp.torch
        sig_h = t.as_tensor(sig_h_np, dtype=t.float64, device=device)[:, None]
        sig_f = t.as_tensor(sig_f_np, dtype=t.float64, device=device)[:, None]
    full = (kk < nz)  # mask of real full levels (level nz of full-level fields is padding)

    # pressure
    ps = 9.7e4 + 6.0e3 * ucol("f_aph")
    aph = ps * sig_h
    ap = ps * sig_f  # = 0.5 (aph[k] + aph[k+1]) on full levels, 0 on the padding level

    # temperature: 6.5 K/km lapse-rate troposphere over an isothermal stratosphere + noise
    if regime == "cold":
        ts = 288.0 + (-45.0 + 25.0 * ucol("f_t"))
    else:
        ts = 288.0 + (-25.0 + 35.0 * ucol("f_t"))
    t_trop = ts * xp.maximum(sig_f, 1e-6) ** 0.190263
    t_strat = 216.65 + 0.0 * t_trop
    tt = xp.maximum(t_trop, t_strat) + 1.5 * normal("f_t", 2)
    trop = t_trop > t_strat

    # specific humidity from a relative humidity w.r.t. a mixed-phase Tetens formula
    rtt = ext["RTT"]
    alfa = xp.minimum(1.0, ((xp.maximum(ext["RTICE"], xp.minimum(ext["RTWAT"], tt)) - ext["RTICE"])
                            * ext["RTWAT_RTICE_R"]) ** 2)
    ew = ext["R2ES"] * (alfa * xp.exp(ext["R3LES"] * (tt - rtt) / (tt - ext["R4LES"]))
                        + (1.0 - alfa) * xp.exp(ext["R3IES"] * (tt - rtt) / (tt - ext["R4IES"])))
    qs = xp.minimum(ew / xp.maximum(ap, 1.0), 0.5)
    qs = qs / (1.0 - ext["RETV"] * qs)
    rh = xp.where(trop, 0.2 + 0.95 * u("f_q"), 0.02 + 0.2 * u("f_q"))
    q = rh * qs

    # condensate: sparse, phase by temperature
    has_c = (u("f_ql", 1) < 0.35) & trop
    cond = xp.where(has_c, 4.0e-5 * u("f_ql", 2) ** 2, 0.0)
    ql = cond * alfa
    qi = cond * (1.0 - alfa)

    # convection: detrained condensate, updraught condensate, mass fluxes
    conv = (u("f_lude", 1) < 0.2) & trop
    lude = xp.where(conv, 2.0e-6 * u("f_lude", 2), 0.0)
    lu = xp.where((u("f_lu", 1) < 0.35) & trop, 3.0e-4 * u("f_lu", 2), 0.0)
    mfu = xp.where(trop, 3.0e-2 * u("f_mfu") ** 2, 0.0)
    mfd = xp.where(trop, -1.0e-2 * u("f_mfd") ** 2, 0.0)

    supsat = xp.where(u("f_supsat", 1) < 0.01, 1.0e-5 * u("f_supsat", 2), 0.0)

    dt = 3600.0
    tnd_t = 2.0e-4 * normal("f_tnd_cml_t")
    tnd_q = 0.02 * normal("f_tnd_cml_q") * q / dt
    tnd_ql = 0.1 * normal("f_tnd_cml_ql") * ql / dt
    tnd_qi = 0.1 * normal("f_tnd_cml_qi") * qi / dt

    def fl(x):  # full-level field: zero padding level
        return xp.cast(xp.where(full, x, 0.0))

    return {
        "f_ap": fl(ap),
        "f_aph": xp.cast(aph),
        "f_lu": fl(lu),
        "f_lude": fl(lude),
        "f_mfd": fl(mfd),
        "f_mfu": fl(mfu),
        "f_q": fl(q),
        "f_qi": fl(qi),
        "f_ql": fl(ql),
        "f_supsat": fl(supsat),
        "f_t": fl(tt),
        "f_tnd_cml_q": fl(tnd_q),
        "f_tnd_cml_qi": fl(tnd_qi),
        "f_tnd_cml_ql": fl(tnd_ql),
        "f_tnd_cml_t": fl(tnd_t),
    }


def eta_levels(nz: int = 137, *, seed: int = SEED, dtype: Any = np.float64) -> np.ndarray:
    """`f_eta` of the synthetic problem: ap[col 0, k] / aph[col 0, nz] for k < nz (and 0 in the
    padding slot), i.e. what `EtaLevels` (common/diagnostics.py:42-45) computes from GLOBAL column 0.
    Computed on the host from a one-column state so every rank gets identical values."""
    s = make_state(1, nz, col0=0, ncols=1, dtype=dtype, seed=seed)
    eta = np.zeros(nz + 1, dtype=dtype)
    eta[:nz] = s["f_ap"][:nz, 0] / s["f_aph"][nz, 0]  # division in the field dtype, as the reference
    return eta
