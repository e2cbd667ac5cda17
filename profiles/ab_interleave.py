#!/usr/bin/env python3
"""Does cloudsc2_nl run slower right after `saturation` than in a back-to-back train?  (dev tool)
Times NL with per-launch HIP events in both patterns, queue kept full so host gaps do not count."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals, make_params
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_IN, NL_OUT
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

nx, nz, dev = 65536, 137, torch.device("cuda:0")
import ctypes as _ct
lib = _ct.CDLL(os.path.abspath(sys.argv[1])) if len(sys.argv) > 1 else _lib.load()
_lib._declare(lib)
print("library:", sys.argv[1] if len(sys.argv) > 1 else "in-tree")
p = make_params(default_externals())
s = make_state(nx, nz, device=dev)
eta = torch.as_tensor(eta_levels(nz), device=dev)
f = {"in_" + k[2:]: storage.logical_view(v) for k, v in s.items()}
f["in_qsat"] = storage.zeros(nx, nz, np.float64, dev)
outs = {n: storage.zeros(nx, nz, np.float64, dev) for n in NL_OUT}
pin = _lib.ptr_array([f["in_" + n].data_ptr() for n in NL_IN]); pout = _lib.ptr_array([outs[n].data_ptr() for n in NL_OUT])
st = torch.cuda.current_stream().cuda_stream
def sat(): lib.cloudsc2_saturation_f64(ctypes.byref(p), nx, nz, nx, f["in_ap"].data_ptr(), f["in_t"].data_ptr(), f["in_qsat"].data_ptr(), st)
def nl(): lib.cloudsc2_nl_f64(ctypes.byref(p), nx, nz, nx, pin, eta.data_ptr(), pout, 3600.0, st)
scratch = torch.empty(1 << 26, dtype=torch.float64, device=dev)
small = torch.empty(1 << 17, dtype=torch.float64, device=dev)
ap2, t2, q2 = (torch.rand((nz + 1) * nx, dtype=torch.float64, device=dev) + 200.0 for _ in range(3))
def sat_other(): lib.cloudsc2_saturation_f64(ctypes.byref(p), nx, nz, nx, ap2.data_ptr(), t2.data_ptr(), q2.data_ptr(), st)
def run(pattern, n=20):
    evs = []
    for _ in range(n):
        if pattern == "sat": sat()
        if pattern == "sat_other": sat_other()
        if pattern == "fill512M": scratch.fill_(1.0)
        if pattern == "fill1M": small.fill_(1.0)
        if pattern == "read512M": scratch.sum()
        if pattern == "idle": torch.cuda._sleep(200000)
        if pattern.startswith("fillMB"): scratch[: int(pattern[6:]) * 131072].fill_(1.0)
        if pattern == "sat+read": sat(); scratch.sum()
        if pattern == "sat+idle": sat(); torch.cuda._sleep(400000)
        if pattern == "sat+nl": sat(); nl()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); nl(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs][3:]) * 1e3
    return np.median(t), t.min()
sat(); nl(); torch.cuda.synchronize()
for rnd in range(2):
    for pat in ("train", "sat", "fillMB64", "idle"):
        print(rnd, f"{pat:10s} NL median %.1f us  min %.1f us" % run(pat))
