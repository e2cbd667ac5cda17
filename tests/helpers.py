"""Shared helpers of the test-suite: synthetic cases, oracle drivers, comparison with tolerances.

The oracle (oracle/cloudsc2_numpy.py) is imported ONLY here and in the tests - it is the checker.
"""
from __future__ import annotations

import os
import sys
from typing import Dict, Mapping

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import DEFAULT_TIMESTEP_S, default_externals  # noqa: E402
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state  # noqa: E402
from oracle import cloudsc2_numpy as oracle  # noqa: E402

NL_IN = ("ap", "aph", "lu", "lude", "mfd", "mfu", "q", "qi", "ql", "qsat", "supsat", "t",
         "tnd_cml_q", "tnd_cml_qi", "tnd_cml_ql", "tnd_cml_t")
NL_OUT = ("clc", "covptot", "fhpsl", "fhpsn", "fplsl", "fplsn", "tnd_q", "tnd_qi", "tnd_ql", "tnd_t")

# Tolerances of the HIP-vs-oracle comparison (stated here once, used by every parity test).
# fp64: both sides evaluate the same formulas in IEEE double; they differ by FMA contraction and
#       by libm-vs-ocml exp/tanh/pow (<= 1-2 ulp each), amplified through cancellations such as
#       (qlwc - ql)/dt.  Errors are measured relative to the field's own scale:
#       |a - b| <= RTOL * |b| + ATOL_REL * max|b|.
# fp32: same with single-precision epsilons.  r04: 4 x tighter than the 2e-3 / 2e-4 of rounds 1-3.  Measured over the whole
#       GPU suite: everything but three isolated `out_clc` points passes at 2e-4 / 2e-5 (clc = 1 - sqrt(...) amplifies a
#       1-ulp difference of its argument to 2.3e-4 near the overcast threshold); worst cases by column scale are in
#       profiles/r02/fp32_errors.txt (<= 2e-5), and the executed-reference vectors are held tighter still
#       (tests/test_reference_exec_f32.py: 3e-5 / 1e-4 of a column's scale).
TOL = {
    np.dtype("float64"): dict(rtol=1e-9, atol_rel=1e-11),
    np.dtype("float32"): dict(rtol=5e-4, atol_rel=5e-5),
}


def externals(**over) -> Dict:
    e = default_externals()
    e.update(over)
    return e


def nl_case(nx: int, nz: int = 137, dtype=np.float64, seed: int = 20240807, ext: Mapping = None,
            col0: int = 0, ncols: int = None, total: int = None):
    """Host-side NL inputs in [k][col] layout: dict in_* (incl. in_qsat from the oracle's
    saturation), eta, dt."""
    ext = ext or externals()
    total = total if total is not None else nx
    s = make_state(total, nz, col0=col0, ncols=ncols if ncols is not None else nx, dtype=dtype, seed=seed)
    eta = eta_levels(nz, seed=seed, dtype=dtype)
    qsat = np.zeros_like(s["f_t"])
    oracle.saturation(s["f_ap"], s["f_t"], qsat, ext)
    fields = {"in_" + k[2:]: v for k, v in s.items()}
    fields["in_qsat"] = qsat
    return fields, eta, DEFAULT_TIMESTEP_S


def run_oracle_nl(fields, eta, dt, ext):
    F = dict(fields)
    for n in NL_OUT:
        F["out_" + n] = np.zeros_like(fields["in_ap"])
    oracle.cloudsc2_nl(F, eta, dt, ext)
    return {n: F["out_" + n] for n in NL_OUT}


def to_device(fields: Mapping[str, np.ndarray], device):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage

    return {k: storage.from_klayout(v, v.dtype, device) for k, v in fields.items()}


def from_device(t):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage

    return storage.klayout(t).cpu().numpy()


def assert_close(name: str, got: np.ndarray, want: np.ndarray, dtype=None, scale: float = None,
                 rtol_mul: float = 1.0):
    dtype = np.dtype(dtype or want.dtype)
    tol = TOL[dtype]
    scale = float(np.max(np.abs(want))) if scale is None else scale
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    bound = rtol_mul * (tol["rtol"] * np.abs(want.astype(np.float64)) + tol["atol_rel"] * scale)
    bad = err > bound
    assert not np.isnan(got).any(), f"{name}: NaN in result"
    if bad.any():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.size} points outside tolerance; worst at {i}: "
            f"got {got[i]!r}, want {want[i]!r}, |err| {err[i]:.3e}, bound {bound[i]:.3e}, scale {scale:.3e}"
        )
    return float(err.max() / scale) if scale > 0 else 0.0


# ----------------------------------------------------------------------------------------------
# TL / AD cases
# ----------------------------------------------------------------------------------------------
def increments(fields, f: float = 0.01, ignore_supsat: bool = False):
    """`state_increment` (common/_stencils/state_increment.py:61-80): x_i = f * x for the 16 inputs."""
    out = {k + "_i": (f * v).astype(v.dtype) for k, v in fields.items()}
    if ignore_supsat:
        out["in_supsat_i"] = np.zeros_like(fields["in_supsat"])
    return out


def run_oracle_tl(fields, fields_i, eta, dt, ext):
    F = dict(fields)
    F.update(fields_i)
    for n in NL_OUT:
        F["out_" + n] = np.zeros_like(fields["in_ap"])
        F["out_" + n + "_i"] = np.zeros_like(fields["in_ap"])
    oracle.cloudsc2_tl(F, eta, dt, ext)
    return ({n: F["out_" + n] for n in NL_OUT}, {n: F["out_" + n + "_i"] for n in NL_OUT})


def run_oracle_ad(fields, forcing, eta, dt, ext):
    """forcing: dict NL_OUT name -> adjoint forcing array.  Returns (nl_outputs, adjoint_outputs)."""
    F = dict(fields)
    for n in NL_OUT:
        F["in_" + n + "_i"] = forcing[n].copy()
        F["out_" + n] = np.zeros_like(fields["in_ap"])
    for n in NL_IN:
        F["out_" + n + "_i"] = np.zeros_like(fields["in_ap"])
    oracle.cloudsc2_ad(F, eta, dt, ext)
    return ({n: F["out_" + n] for n in NL_OUT}, {n: F["out_" + n + "_i"] for n in NL_IN})


def nlev_of(name: str, nz: int) -> int:
    """number of levels a stencil writes for a field: half-level fields (aph, fluxes) nz+1, others nz"""
    half = name in ("aph", "fhpsl", "fhpsn", "fplsl", "fplsn")
    return nz + 1 if half else nz


def taylor_norms(nl0, nlp_of_f2, tl_i, f2s):
    """TaylorTest.get_norm (tangent_linear/validation.py:219-261) for each factor2."""
    import sys as _sys

    names = ("tnd_t", "tnd_q", "tnd_ql", "tnd_qi", "clc", "fhpsl", "fhpsn", "fplsl", "fplsn", "covptot")
    norms = []
    for f2 in f2s:
        nlp = nlp_of_f2(f2)
        tot, cnt = 0.0, 0
        for n in names:
            den = abs(f2 * tl_i[n].sum())
            norm = abs((nlp[n] - nl0[n]).sum()) / den if den > _sys.float_info.epsilon else 0
            cnt += norm > 0
            tot += norm
        norms.append(tot / cnt if cnt else 0.0)
    return np.array(norms)


def taylor_verdict(norms):
    """TaylorTest.validate scoring (tangent_linear/validation.py:183-217) -> (passed, message)."""
    e = np.abs(1 - np.asarray(norms, dtype=float))
    start = -1
    for i in range(e.size):
        if start == -1 and e[i] < 0.5:
            start = i
    if start == -1 or start > 3:
        return False, "The test failed with error 13."
    test, negat = -10, 1
    for i in range(start, e.size - 1):
        tmp_negat = int(e[i + 1] < e[i])
        if negat > tmp_negat:
            test += 10
        negat = tmp_negat
    if test == -10:
        test = 11
    if np.min(e[start:]) > 1e-5:
        test += 7
    if np.min(e[start:]) > 1e-6:
        test += 5
    if test > 5:
        return False, f"The test failed with error {test}."
    return True, f"The test passed with penalty {test}. HOORAY!"


def symmetry_norm3(tl_out_i, fields_i, ad_out_i, dtype=np.float64):
    """SymmetryTest norms (adjoint/validation.py:157-215): per-column |norm1 - norm2| / (eps norm2)."""
    norm1 = sum((tl_out_i[n].astype(np.float64) ** 2).sum(axis=0) for n in NL_OUT)
    norm2 = sum((fields_i["in_" + n + "_i"].astype(np.float64) * ad_out_i[n].astype(np.float64)).sum(axis=0)
                for n in NL_IN)
    eps = np.finfo(dtype).eps
    with np.errstate(divide="ignore", invalid="ignore"):
        norm3 = np.where(norm2 == 0, abs(norm1 - norm2) / eps, abs(norm1 - norm2) / (eps * norm2))
    return norm1, norm2, norm3
