"""Physical parameters ("externals") of the CLOUDSC2 stencils and their C-ABI image.

The reference bakes these into each compiled stencil as GT4Py externals
(/root/reference/src/cloudsc2_gt4py/physics/nonlinear/microphysics.py:62-79,
 tangent_linear/microphysics.py:73-92, adjoint/microphysics.py:73-89,
 common/saturation.py:51-54, common/increment.py:47-49); their names and defaults are the
pydantic models of /root/reference/src/cloudsc2_gt4py/iox.py:25-209.  Here the numeric ones are a
plain struct handed to the HIP kernels by value and the boolean ones select a kernel
instantiation (see include/cloudsc2_hip.h).

The authoritative values live in the reference's `data/input.h5`, which is not part of the
reference checkout (/root/reference/.MISSING_LARGE_BLOBS:1).  `default_externals()` therefore
returns the *provisional* IFS constants of SURVEY.md Appendix E; two of them (RLSTT and
RLSTT/RCPD) are confirmed by the golden output files (tests/test_golden_invariants.py).
Every report produced with them must say "synthetic-parameters".
"""
from __future__ import annotations

import ctypes
import math
from typing import Any, Dict, Mapping

# order == field order of `struct Cloudsc2Params` in include/cloudsc2_hip.h
_REAL_FIELDS = (
    "R2ES", "R3IES", "R3LES", "R4IES", "R4LES", "R5IES", "R5LES",
    "R5ALSCP", "R5ALVCP", "RALSDCP", "RALVDCP",
    "RTICE", "RTWAT", "RTWAT_RTICE_R", "RTICECU", "RTWAT_RTICECU_R", "RVTMP2",
    "RCPD", "RD", "RETV", "RG", "RLMLT", "RLSTT", "RLVTT", "RTT",
    "RCLCRIT", "RKCONV", "RLMIN", "RPECONS", "RLPTRC",
    "ZEPS1", "ZEPS2", "ZQMAX", "ZSCAL", "QMAX",
)
_INT_FIELDS = (
    "LPHYLIN", "LDRAIN1D", "LEVAPLS2", "LREGCL", "ICALL", "KFLAG", "IGNORE_SUPSAT", "NLEV",
    "AD_TRAJ_FIX",  # build extension (not a reference external), see include/cloudsc2_hip.h
)

ABI_VERSION = 3


class Cloudsc2Params(ctypes.Structure):
    """ctypes mirror of `struct Cloudsc2Params` (include/cloudsc2_hip.h)."""

    _fields_ = [(n, ctypes.c_double) for n in _REAL_FIELDS] + [
        (n, ctypes.c_int32) for n in _INT_FIELDS
    ]


def default_externals() -> Dict[str, Any]:
    """Provisional parameter set + the literals the components add."""
    RD = 287.0597
    RV = 461.5250
    RTT = 273.16
    RLVTT = 2.5008e6
    RLSTT = 2.8345e6
    RCPD = 3.5 * RD
    R3LES, R3IES, R4LES, R4IES = 17.502, 22.587, 32.19, -0.7
    R5LES = R3LES * (RTT - R4LES)
    R5IES = R3IES * (RTT - R4IES)
    RTWAT = RTT
    RTICE = RTT - 23.0
    RTICECU = RTT - 23.0
    ext: Dict[str, Any] = dict(
        # YOMCST (iox.py:48-57)
        RG=9.80665, RD=RD, RV=RV, RCPD=RCPD, RETV=RV / RD - 1.0,
        RLVTT=RLVTT, RLSTT=RLSTT, RLMLT=RLSTT - RLVTT, RTT=RTT,
        # YOETHF (iox.py:25-45)
        R2ES=611.21 * RD / RV, R3LES=R3LES, R3IES=R3IES, R4LES=R4LES, R4IES=R4IES,
        R5LES=R5LES, R5IES=R5IES,
        R5ALVCP=R5LES * RLVTT / RCPD, R5ALSCP=R5IES * RLSTT / RCPD,
        RALVDCP=RLVTT / RCPD, RALSDCP=RLSTT / RCPD, RALFDCP=(RLSTT - RLVTT) / RCPD,
        RTWAT=RTWAT, RTICE=RTICE, RTICECU=RTICECU,
        RTWAT_RTICE_R=1.0 / (RTWAT - RTICE), RTWAT_RTICECU_R=1.0 / (RTWAT - RTICECU),
        RKOOP1=2.583, RKOOP2=0.48116e-2, RVTMP2=0.0,
        # YRECLDP (iox.py:60-181): only the four the stencils import
        RLMIN=1.0e-8, RKCONV=1.0 / 6000.0, RCLCRIT=4.0e-4, RPECONS=5.547e-5,
        # YREPHLI (iox.py:184-201)
        RLPTRC=RTICE + (RTWAT - RTICE) / math.sqrt(2.0), LPHYLIN=True,
        # YRNCL / YRPHNC (iox.py:204-209)
        LREGCL=True, LEVAPLS2=False,
        # literals set by the components
        ICALL=0, LDRAIN1D=False, ZEPS1=1.0e-12, ZEPS2=1.0e-10, ZQMAX=0.5, ZSCAL=0.9,
        QMAX=0.5, KFLAG=1, IGNORE_SUPSAT=False, NLEV=137,
        AD_TRAJ_FIX=0,
    )
    return ext


DEFAULT_TIMESTEP_S = 3600.0  # provisional (PTSPHY lives in the missing input.h5)


def make_params(externals: Mapping[str, Any]) -> Cloudsc2Params:
    """Fill the C struct from a flat externals dict; unknown keys are ignored, missing ones
    take the provisional default (so `saturation`'s smaller externals dict works too)."""
    base = default_externals()
    p = Cloudsc2Params()
    for n in _REAL_FIELDS:
        setattr(p, n, float(externals.get(n, base[n])))
    for n in _INT_FIELDS:
        setattr(p, n, int(externals.get(n, base[n])))
    return p


def params_as_dict(p: Cloudsc2Params) -> Dict[str, Any]:
    return {n: getattr(p, n) for n, _ in p._fields_}
