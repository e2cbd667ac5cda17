"""Run one UNMODIFIED reference driver (/root/reference/drivers/run_*.py) with `--backend hip` in the GPU-less build
container, against a RECORDING FAKE of libcloudsc2_hip.so.

What is real: the reference's drivers, its component classes (Cloudsc2NL / Cloudsc2TL / Cloudsc2AD, Saturation,
StateIncrement, PerturbedState, the Taylor / symmetry harnesses), this build's `ifs_physics_common` shim, its backend
registry, and the product's stencil objects (`stencils.HipStencil.__call__` and every `_launch`).
What is fake, FOR THIS TEST ONLY: the C-ABI entry points (they record their arguments and return 0 - nothing is computed,
outputs stay zero), the device of backend "hip" (host tensors), the HIP stream and `torch.cuda.device`.

  python tests/run_reference_on_recording_hip.py <record.json> run_nonlinear.py --num-cols 64 ...

The record holds, per stencil call, the keyword names and data pointers the REFERENCE component passed (taken at
`HipStencil.__call__`) and, per C-ABI call, the entry point's name and the arguments it received."""
import contextlib
import ctypes
import json
import os
import runpy
import sys

sys.dont_write_bytecode = True   # importing the reference package must not leave __pycache__ in the read-only checkout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("CLOUDSC2_REFERENCE", "/root/reference")

STENCIL_CALLS = []      # what the reference's components handed to the stencil objects
ABI_CALLS = []          # what reached the C ABI


def _plain(a):
    """a ctypes argument as JSON data"""
    if isinstance(a, ctypes.Array):
        return [int(x or 0) for x in a]
    if hasattr(a, "_obj"):                                   # ctypes.byref(Cloudsc2Params)
        p = a._obj
        return {"params": {n: getattr(p, n) for n, _ in p._fields_}}
    if a is None:
        return None
    return a


class RecordingLib:
    """stands where ctypes.CDLL(libcloudsc2_hip.so) stands in the product"""

    def cloudsc2_last_error(self):
        return b""

    def cloudsc2_last_kernel(self):
        return b"recording fake"

    def __getattr__(self, name):
        if not name.startswith("cloudsc2_"):
            raise AttributeError(name)

        def entry(*args):
            ABI_CALLS.append({"entry": name, "args": [_plain(a) for a in args]})
            return 0

        return entry


def install_fakes():
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, stencils
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework import backends

    fake = RecordingLib()
    _lib._lib = fake                                        # `_lib.load()` returns what is already loaded
    backends.register_backend("hip", backends._hip_compile, torch.device("cpu"))     # host storages, this test only
    stencils._current_stream_ptr = lambda device: 0
    torch.cuda.device = lambda device: contextlib.nullcontext()

    class NoEvent:                                          # exec_info brackets every launch with a HIP event pair
        def __init__(self, enable_timing=False):
            pass

        def record(self, stream=None):
            pass

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return 0.0

    torch.cuda.Event = NoEvent
    torch.cuda.synchronize = lambda device=None: None

    real_call = stencils.HipStencil.__call__

    def recording_call(self, **kwargs):
        rec = {"stencil": self.name, "externals": {k: v for k, v in self.externals.items() if isinstance(v, (bool, int))},
               "kwargs": sorted(kwargs), "fields": {}, "scalars": {}, "abi_first": len(ABI_CALLS)}
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                rec["fields"][k] = {"ptr": v.data_ptr(), "shape": list(v.shape), "strides": list(v.stride()),
                                    "dtype": str(v.dtype)}
            elif k in ("dt", "f"):
                rec["scalars"][k] = {"value": float(v), "type": type(v).__name__}
            elif k in ("origin", "domain"):
                rec[k] = [int(x) for x in v]
            elif k == "validate_args":
                rec[k] = bool(v)
        real_call(self, **kwargs)
        rec["abi_last"] = len(ABI_CALLS)
        STENCIL_CALLS.append(rec)

    stencils.HipStencil.__call__ = recording_call


def main() -> None:
    record, driver = sys.argv[1], os.path.join(REFERENCE, "drivers", sys.argv[2])
    for p in (os.path.join(REFERENCE, "drivers"), os.path.join(REFERENCE, "src"), os.path.join(ROOT, "shim"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    install_fakes()
    sys.argv = [driver] + sys.argv[3:]
    try:
        runpy.run_path(driver, run_name="__main__")
    except SystemExit as exc:                                # click ends the command with SystemExit(0)
        if exc.code not in (0, None):
            raise
    finally:
        with open(record, "w") as fh:
            json.dump({"stencil_calls": STENCIL_CALLS, "abi_calls": ABI_CALLS}, fh)


if __name__ == "__main__":
    main()
