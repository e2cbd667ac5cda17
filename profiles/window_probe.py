#!/usr/bin/env python3
"""Where does the fixed cost of bench.py's timed window go?  (20 steps report 0.417-0.424 ms per
step, 100 steps 0.363-0.379 ms.)  Reproduces the window (W warm-up steps, synchronize, K steps, synchronize) with one HIP
event pair per step and a host time stamp per enqueue, for several ways of entering the window.
  python profiles/window_probe.py > profiles/r02/window_probe.txt"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage  # noqa: E402
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals  # noqa: E402
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_OUT, compile_stencil  # noqa: E402
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state  # noqa: E402

dev = torch.device("cuda:0")
nx, nz = 65536, 137
ext = default_externals()
s = make_state(nx, nz, device=dev)
eta = torch.as_tensor(eta_levels(nz), device=dev)
f = {k: storage.logical_view(v) for k, v in s.items()}
qsat = storage.zeros(nx, nz, np.float64, dev)
ins = {"in_" + k[2:]: v for k, v in f.items()}
ins["in_qsat"] = qsat
outs = {"out_" + n: storage.zeros(nx, nz, np.float64, dev) for n in NL_OUT}
sat = compile_stencil("saturation", ext)
nl = compile_stencil("cloudsc2_nl", ext)
com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)


def step():
    sat(in_ap=f["f_ap"], in_t=f["f_t"], out_qsat=qsat, domain=(nx, 1, nz), **com)
    nl(**ins, **outs, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)


def window(W, K, idle_ms=0.0, events=True, label=""):
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    if idle_ms:
        time.sleep(idle_ms * 1e-3)
    evs, host = [], []
    t0 = time.perf_counter()
    for _ in range(K):
        if events:
            a = torch.cuda.Event(enable_timing=True)
            a.record()
        host.append(time.perf_counter() - t0)
        step()
        if events:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            evs.append((a, b))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    line = f"{label:44s} W={W:3d} K={K:3d}: wall {1e3 * (t2 - t0) / K:.4f} ms/step (enqueue done after {1e3 * (t1 - t0):.2f} ms)"
    if events:
        ms = [a.elapsed_time(b) for a, b in evs]
        gap0 = evs[0][0].elapsed_time(evs[-1][1])
        line += (f"; GPU first..last {gap0:.3f} ms; per-step events: first {ms[0]:.3f}, 2nd {ms[1]:.3f}, 3rd {ms[2]:.3f}, "
                 f"mean of the rest {sum(ms[3:]) / len(ms[3:]):.4f}, max {max(ms):.3f}")
    print(line, flush=True)


gc.collect()
gc.disable()
for _ in range(50):
    step()
torch.cuda.synchronize()
print(torch.cuda.get_device_name(0))
for rep in range(2):
    window(5, 20, label="busy GPU -> 5 warm-up -> window")
    window(20, 100, label="busy GPU -> 20 warm-up -> window")
    window(5, 20, events=False, label="same, no per-step events")
    window(20, 100, events=False, label="same, no per-step events")
    window(5, 20, idle_ms=2.0, label="2 ms host pause after the opening sync")
    window(5, 20, idle_ms=50.0, label="50 ms host pause after the opening sync")
    time.sleep(0.2)
    window(5, 20, label="200 ms idle BEFORE the warm-up")
    time.sleep(0.2)
    window(20, 20, label="200 ms idle BEFORE a 20-step warm-up")
    time.sleep(0.2)
    window(60, 20, label="200 ms idle BEFORE a 60-step warm-up")
