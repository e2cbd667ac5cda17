#!/usr/bin/env python3
"""Does `storage.tune_placement` pay for the bench's step, and does the gain hold?  Tunes the placement of the 26 fields of
(saturation, cloudsc2_nl) with the step itself as the objective, then times default-arena fields, separately allocated
fields and the tuned fields alternately (rounds x 50 steps, HIP events).
  python profiles/tune_probe.py [--cols=65536] [--precision=double] > profiles/r02/tune_probe_<n>.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    import __graft_entry__ as ge

    ge.build()
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_IN, NL_OUT, compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    nx = int(opts.get("cols", 65536))
    np_dtype = np.float64 if opts.get("precision", "double") == "double" else np.float32
    nz, dev, dt = 137, torch.device("cuda:0"), 3600.0
    ext = default_externals()
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    sat = compile_stencil("saturation", ext)
    nl = compile_stencil("cloudsc2_nl", ext)
    order = ["in_" + n for n in NL_IN] + ["out_" + n for n in NL_OUT]
    sources = {"in_" + k[2:]: v for k, v in s.items()}

    def step(F):
        sat(in_ap=F["in_ap"], in_t=F["in_t"], out_qsat=F["in_qsat"], domain=(nx, 1, nz), **com)
        nl(**F, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)

    def build(arena):
        storage.set_arena_capacity(32 if arena else 0)
        F = {k: storage.from_klayout(v, np_dtype, dev) for k, v in sources.items()}
        F["in_qsat"] = storage.zeros(nx, nz, np_dtype, dev)
        F.update({"out_" + n: storage.zeros(nx, nz, np_dtype, dev) for n in NL_OUT})
        return F

    sets = {"separate": build(False), "arena default": build(True)}
    for _ in range(80):
        step(sets["separate"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tuned, rep = storage.tune_placement(nx, nz, np_dtype, dev, order, sources, step)
    torch.cuda.synchronize()
    print(f"{nx} columns, {np.dtype(np_dtype).name}, {torch.cuda.get_device_name(0)}; tuning took {time.perf_counter() - t0:.2f} s: {rep}")
    sets["tuned"] = tuned
    res = {k: [] for k in sets}
    for r in range(6):
        for k, F in sets.items():
            step(F)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                step(F)
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / 50 * 1e3)
    for k, v in res.items():
        print(f"  {k:14s} step median {np.median(v):7.1f} us  (min {min(v):.1f}, max {max(v):.1f})  -> {nx / np.median(v):.1f} M columns/s")
    same = all(torch.equal(sets["tuned"]["out_" + n], sets["separate"]["out_" + n]) for n in NL_OUT)
    print("  results bit-identical across placements:", same)


if __name__ == "__main__":
    main()
