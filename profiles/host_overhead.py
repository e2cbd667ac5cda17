import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_OUT, compile_stencil
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state
dev = torch.device("cuda:0"); nx, nz = 65536, 137
ext = default_externals()
s = make_state(nx, nz, device=dev); eta = torch.as_tensor(eta_levels(nz), device=dev)
f = {k: storage.logical_view(v) for k, v in s.items()}
qsat = storage.zeros(nx, nz, np.float64, dev)
ins = {"in_" + k[2:]: v for k, v in f.items()}; ins["in_qsat"] = qsat
outs = {"out_" + n: storage.zeros(nx, nz, np.float64, dev) for n in NL_OUT}
sat = compile_stencil("saturation", ext); nl = compile_stencil("cloudsc2_nl", ext)
def step():
    sat(in_ap=f["f_ap"], in_t=f["f_t"], out_qsat=qsat, origin=(0, 0, 0), domain=(nx, 1, nz), validate_args=False, exec_info=None)
    nl(**ins, **outs, in_eta=eta, dt=3600.0, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=False, exec_info=None)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue per step %.1f us; total per step %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
# GPU-side time of the same 50 steps by events
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): step()
b.record(); torch.cuda.synchronize()
print("event time per step %.1f us" % (a.elapsed_time(b) / 50 * 1e3))
# how much of the wall-clock figure is the wake-up latency of a blocking synchronize()?
for mode in ("blocking synchronize", "event-query spin, then synchronize"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step()
    if mode.startswith("event"):
        ev = torch.cuda.Event(); ev.record()
        while not ev.query():
            pass
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-36s total per step %.1f us" % (mode, (t2 - t0) / 50 * 1e6))
