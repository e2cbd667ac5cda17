"""ctypes binding of libcloudsc2_hip.so (the C ABI declared in include/cloudsc2_hip.h).

There is NO fallback: if the shared library is missing or stale this module raises, and every
stencil object built on top of it is unusable.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C gt4py_dwarf_p_cloudsc2_tl_ad_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_void_p
from typing import Optional

from .params import ABI_VERSION, Cloudsc2Params

LIB_NAME = "libcloudsc2_hip.so"
#: CLOUDSC2_HIP_LIB overrides the library file (dev / A-B builds made by profiles/build_variants.sh); same checks apply
LIB_PATH = os.environ.get("CLOUDSC2_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

NL_NUM_IN = 16
NL_NUM_OUT = 10
INC_NUM = 16

#: every symbol include/cloudsc2_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "cloudsc2_abi_version", "cloudsc2_params_sizeof", "cloudsc2_last_error", "cloudsc2_last_kernel",
    "cloudsc2_device_count",
    "cloudsc2_nl_f64", "cloudsc2_nl_f32",
    "cloudsc2_nl_fused_f64", "cloudsc2_nl_fused_f32",
    "cloudsc2_nl_taylor_blocks", "cloudsc2_nl_taylor_f64", "cloudsc2_nl_taylor_f32",
    "cloudsc2_nl_taylor_multi_f64", "cloudsc2_nl_taylor_multi_f32",
    "cloudsc2_field_sums_blocks", "cloudsc2_field_sums_f64", "cloudsc2_field_sums_f32",
    "cloudsc2_column_dots_chunks", "cloudsc2_column_dots_f64", "cloudsc2_column_dots_f32",
    "cloudsc2_tl_f64", "cloudsc2_tl_f32",
    "cloudsc2_tl_incremented_f64", "cloudsc2_tl_incremented_f32",
    "cloudsc2_ad_f64", "cloudsc2_ad_f32",
    "cloudsc2_ad_from_trajectory_f64", "cloudsc2_ad_from_trajectory_f32",
    "cloudsc2_saturation_f64", "cloudsc2_saturation_f32",
    "cloudsc2_state_increment_f64", "cloudsc2_state_increment_f32",
    "cloudsc2_perturbed_state_f64", "cloudsc2_perturbed_state_f32",
)

_lib: Optional[ctypes.CDLL] = None


class Cloudsc2LibraryError(RuntimeError):
    pass


def _declare(lib: ctypes.CDLL) -> None:
    PP = POINTER(Cloudsc2Params)
    lib.cloudsc2_abi_version.restype = c_int32
    lib.cloudsc2_abi_version.argtypes = []
    lib.cloudsc2_params_sizeof.restype = c_int32
    lib.cloudsc2_params_sizeof.argtypes = []
    lib.cloudsc2_last_error.restype = c_char_p
    lib.cloudsc2_last_error.argtypes = []
    lib.cloudsc2_last_kernel.restype = c_char_p
    lib.cloudsc2_last_kernel.argtypes = []
    lib.cloudsc2_device_count.restype = c_int32
    lib.cloudsc2_device_count.argtypes = []
    lib.cloudsc2_nl_taylor_blocks.restype = c_int32
    lib.cloudsc2_nl_taylor_blocks.argtypes = [c_int32]
    lib.cloudsc2_field_sums_blocks.restype = c_int32
    lib.cloudsc2_field_sums_blocks.argtypes = [c_int32, c_int32]
    lib.cloudsc2_column_dots_chunks.restype = c_int32
    lib.cloudsc2_column_dots_chunks.argtypes = [c_int32]
    for sfx, real in (("f64", c_double), ("f32", c_float)):
        del real  # device pointers travel as integers (void*), never dereferenced on the host
        parr = POINTER(c_void_p)
        common = [PP, c_int32, c_int32, c_int64]
        f = getattr(lib, f"cloudsc2_nl_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, c_void_p, parr, c_double, c_void_p]
        f = getattr(lib, f"cloudsc2_nl_fused_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, parr, c_double, c_void_p, c_void_p, parr, c_double, c_void_p]
        f = getattr(lib, f"cloudsc2_nl_taylor_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, parr, c_double, c_void_p, parr, c_void_p, c_double, c_void_p]
        f = getattr(lib, f"cloudsc2_nl_taylor_multi_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, parr, c_double, c_int32, POINTER(c_double), c_void_p, parr, c_void_p, c_double, c_void_p]
        f = getattr(lib, f"cloudsc2_field_sums_{sfx}")
        f.restype = c_int32
        f.argtypes = [c_int32, c_int32, c_int64, c_int32, parr, parr, c_void_p, c_void_p]
        f = getattr(lib, f"cloudsc2_column_dots_{sfx}")
        f.restype = c_int32
        f.argtypes = [c_int32, c_int32, c_int64, c_int32, parr, parr, c_void_p, c_int32, c_void_p]
        f = getattr(lib, f"cloudsc2_tl_incremented_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, c_double, c_void_p, parr, parr, c_double, c_void_p]
        for name in ("tl", "ad"):
            f = getattr(lib, f"cloudsc2_{name}_{sfx}")
            f.restype = c_int32
            f.argtypes = common + [parr, parr, c_void_p, parr, parr, c_double, c_void_p]
        f = getattr(lib, f"cloudsc2_ad_from_trajectory_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, parr, c_void_p, c_void_p, c_void_p, parr, c_double, c_void_p]
        f = getattr(lib, f"cloudsc2_saturation_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [c_void_p, c_void_p, c_void_p, c_void_p]
        f = getattr(lib, f"cloudsc2_state_increment_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, parr, c_double, c_void_p]
        f = getattr(lib, f"cloudsc2_perturbed_state_{sfx}")
        f.restype = c_int32
        f.argtypes = common + [parr, parr, parr, c_double, c_void_p]


def load() -> ctypes.CDLL:
    """Load (once) and return the library; raises `Cloudsc2LibraryError` if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Cloudsc2LibraryError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `make -C gt4py_dwarf_p_cloudsc2_tl_ad_amd/csrc` or `__graft_entry__.build()`). "
            "There is no CPU fallback."
        )
    # torch ships its own libamdhip64 (same soname as /opt/rocm's): import it first so that the
    # kernels, torch's allocator and torch's streams all live in ONE HIP runtime.
    import torch  # noqa: F401

    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise Cloudsc2LibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    missing = [s for s in EXPORTED_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise Cloudsc2LibraryError(f"{LIB_PATH} lacks symbols {missing}: stale build, rebuild it")
    _declare(lib)
    if lib.cloudsc2_abi_version() != ABI_VERSION:
        raise Cloudsc2LibraryError(
            f"ABI version mismatch: library {lib.cloudsc2_abi_version()}, python {ABI_VERSION}"
        )
    if lib.cloudsc2_params_sizeof() != ctypes.sizeof(Cloudsc2Params):
        raise Cloudsc2LibraryError(
            f"Cloudsc2Params size mismatch: library {lib.cloudsc2_params_sizeof()}, "
            f"python {ctypes.sizeof(Cloudsc2Params)}"
        )
    _lib = lib
    return lib


def last_error() -> str:
    return load().cloudsc2_last_error().decode("utf-8", "replace")


def last_kernel() -> str:
    """Name of the kernel the last stencil call of this thread launched (diagnostics, e.g. for bench.py's record)."""
    return load().cloudsc2_last_kernel().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    """Map a C-ABI return code to the exception the reference's stencil call would raise."""
    if rc == 0:
        return
    msg = f"{what}: {last_error()} (code {rc})"
    if rc in (-1, -2):
        raise ValueError(msg)
    raise RuntimeError(msg)


def ptr_array(ptrs) -> ctypes.Array:
    arr = (c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
