"""ORACLE / CPU BASELINE loader (test infrastructure, NOT product code): ctypes binding of
oracle/_build/libcloudsc2_oracle.so, the plain-C + OpenMP restatement of `saturation` and `cloudsc2_nl`
(oracle/cloudsc2_nl_omp.c).  Same [level][column] host arrays as oracle/cloudsc2_numpy.py.
Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_int64, c_void_p
from typing import Dict, Mapping

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libcloudsc2_oracle.so")

_NL_IN = ("ap", "aph", "lu", "lude", "mfd", "mfu", "q", "qi", "ql", "qsat", "supsat", "t",
          "tnd_cml_q", "tnd_cml_qi", "tnd_cml_ql", "tnd_cml_t")
_NL_OUT = ("clc", "covptot", "fhpsl", "fhpsn", "fplsl", "fplsn", "tnd_q", "tnd_qi", "tnd_ql", "tnd_t")
_lib = None


def build() -> str:
    """(Re)build the library with the committed recipe (oracle/Makefile)."""
    subprocess.run(["make", "-C", HERE, "--no-print-directory"], check=True)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = ctypes.CDLL(LIB_PATH)
        lib.cs2c_max_threads.restype = c_int
        lib.cs2c_saturation.restype = c_int
        lib.cs2c_saturation.argtypes = [c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int]
        lib.cs2c_nl.restype = c_int
        lib.cs2c_nl.argtypes = [c_void_p, c_int, c_int, c_int64, POINTER(c_void_p), c_void_p, POINTER(c_void_p),
                                c_double, c_int]
        _lib = lib
    return _lib


def max_threads() -> int:
    return int(load().cs2c_max_threads())


def _check(a: np.ndarray) -> np.ndarray:
    if a.dtype != np.float64 or not a.flags.c_contiguous:
        raise ValueError("the C restatement takes C-contiguous float64 [level][column] arrays")
    return a


def _params(externals: Mapping):
    # the struct definition is shared with the product's ctypes mirror (a data declaration, not a compute path)
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import make_params

    return make_params(dict(externals))


def saturation(in_ap: np.ndarray, in_t: np.ndarray, out_qsat: np.ndarray, externals: Mapping, nthreads: int = 0) -> None:
    nzp1, nx = _check(in_ap).shape
    p = _params(dict(externals, NLEV=nzp1 - 1))
    rc = load().cs2c_saturation(ctypes.addressof(p), nx, nzp1 - 1, nx, _check(in_ap).ctypes.data,
                                _check(in_t).ctypes.data, _check(out_qsat).ctypes.data, nthreads)
    if rc != 0:
        raise ValueError(f"cs2c_saturation failed ({rc})")


def cloudsc2_nl(fields: Dict[str, np.ndarray], in_eta: np.ndarray, dt: float, externals: Mapping,
                nthreads: int = 0) -> None:
    nzp1, nx = _check(fields["in_ap"]).shape
    p = _params(dict(externals, NLEV=nzp1 - 1))
    ins = (c_void_p * 16)(*[_check(fields["in_" + n]).ctypes.data for n in _NL_IN])
    outs = (c_void_p * 10)(*[_check(fields["out_" + n]).ctypes.data for n in _NL_OUT])
    eta = np.ascontiguousarray(in_eta, dtype=np.float64)
    rc = load().cs2c_nl(ctypes.addressof(p), nx, nzp1 - 1, nx, ins, eta.ctypes.data, outs, float(dt), nthreads)
    if rc != 0:
        raise ValueError(f"cs2c_nl failed ({rc})")
