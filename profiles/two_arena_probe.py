#!/usr/bin/env python3
"""The fast shifts of a torch arena are junctions of its physical backing (profiles/r02/placement_structure.txt 1, 7).  Can the
junction be MADE: fields 0..G-1 of cloudsc2_nl in allocation A, fields G.. in a second, separate allocation B (each with 192 MB
or dense 2-MB-slab spacing, stagger 2304)?  NL kernel by HIP events, several B allocations, alternating.
  python profiles/two_arena_probe.py [--cols=65536] [--precision=double] [--rounds=3]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    import __graft_entry__ as ge

    ge.build()
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals, make_params
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_IN, NL_OUT
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    nx = int(opts.get("cols", 65536))
    rounds = int(opts.get("rounds", 3))
    prec = opts.get("precision", "double")
    np_dtype = np.float64 if prec == "double" else np.float32
    tdt = storage.torch_dtype(np_dtype)
    sfx = "f64" if prec == "double" else "f32"
    item = np.dtype(np_dtype).itemsize
    nz, dev = 137, torch.device("cuda:0")
    lib = _lib.load()
    p = make_params(dict(default_externals(), NLEV=nz))
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    qsat_src = storage.zeros(nx, nz, np_dtype, dev)
    getattr(lib, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, nx, s["f_ap"].data_ptr(), s["f_t"].data_ptr(),
                                                 qsat_src.data_ptr(), stream)
    src = {n: (s["f_" + n] if n != "qsat" else storage.klayout(qsat_src)) for n in NL_IN}
    fn = getattr(lib, "cloudsc2_nl_" + sfx)
    nf = len(NL_IN) + len(NL_OUT)
    two_mb = 2 << 20
    fbytes = (nz + 1) * nx * item
    slab = (fbytes + 65536 + two_mb - 1) // two_mb * two_mb
    wide = max(slab, 192 << 20)
    size = nf * wide + 2 * two_mb
    A = torch.zeros(size, dtype=torch.uint8, device=dev)
    filler = [torch.zeros(int(g) << 30, dtype=torch.uint8, device=dev) for g in opts.get("fill_gb", "3,7").split(",")]
    Bs = []
    for f in filler + [None]:                      # B allocations made at different moments (other allocations in between)
        Bs.append(torch.zeros(size, dtype=torch.uint8, device=dev))
    print(f"cloudsc2_nl {prec} {nx} columns; A at {A.data_ptr():#x}, B at " + ", ".join(f"{b.data_ptr():#x}" for b in Bs))

    def views(G, B, sp):
        vs = []
        for i in range(nf):
            buf = A if i < G else B
            base = (-buf.data_ptr()) % two_mb
            o = base + i * sp + (i * 2304) % 65536
            vs.append(buf[o:o + fbytes].view(tdt).view(nz + 1, nx))
        for n, v in zip(NL_IN, vs):
            v.copy_(src[n])
        return vs

    def timed(vs):
        pin = _lib.ptr_array([v.data_ptr() for v in vs[:len(NL_IN)]])
        pout = _lib.ptr_array([v.data_ptr() for v in vs[len(NL_IN):]])
        for _ in range(3):
            assert fn(ctypes.byref(p), nx, nz, nx, pin, eta.data_ptr(), pout, 3600.0, stream) == 0
        ts = []
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn(ctypes.byref(p), nx, nz, nx, pin, eta.data_ptr(), pout, 3600.0, stream)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 5)
        return float(np.median(ts)) * 1e3

    for _ in range(60):
        timed(views(nf, A, slab))
    for sp, tag in ((wide, "192 MB spacing"), (slab, "dense")):
        for rnd in range(2):
            print(f"  {tag:15s} round {rnd}: all in A {timed(views(nf, A, sp)):6.1f} us;", end="")
            for G in (24, 22, 20, 16, 10, 4):
                print(f"  G={G:2d}: " + " / ".join(f"{timed(views(G, B, sp)):5.1f}" for B in Bs), end="")
            print(flush=True)


if __name__ == "__main__":
    main()
