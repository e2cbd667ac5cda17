#!/usr/bin/env python3
"""Does the LEVEL PITCH (elements between two levels of a field; the C ABI's `level_stride`) matter?  At the headline size
a level is 65 536 x 8 B = 512 KB, at 524 288 fp32 columns 2 MB: powers of two, so every level of a field starts at the same
offset inside a 512 KB / 2 MB frame.  ONE arena; the 26 fields of cloudsc2_nl on 2-MB slabs (+ `extra` x 2 MB, stagger
2304 B), each stored with pitch = nx + pad elements; median of `rounds` x 5 launches by HIP events.
  python profiles/pitch_scan.py [--cols=65536] [--precision=double] [--rounds=5] [--extras=0,3]"""
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
    rounds = int(opts.get("rounds", 5))
    prec = opts.get("precision", "double")
    extras = [int(x) for x in opts.get("extras", "0,3").split(",")]
    pads = [int(x) for x in opts.get("pads", "0,32,64,128,288,576,1056,2112,4160,8256,16448").split(",")]
    np_dtype = np.float64 if prec == "double" else np.float32
    sfx = "f64" if prec == "double" else "f32"
    item = np.dtype(np_dtype).itemsize
    nz, dev = 137, torch.device("cuda:0")
    lib = _lib.load()
    p = make_params(dict(default_externals(), NLEV=nz))
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    nfields = len(NL_IN) + len(NL_OUT)
    two_mb = 2 << 20
    maxf = (nz + 1) * (nx + max(pads)) * item
    arena_bytes = nfields * (maxf + (max(extras) + 2) * two_mb) + (4 << 20)
    spads = [int(x) for x in opts.get("spads", "0").split(",")]        # bytes added to the slab spacing (any multiple of 256)
    arena_bytes += nfields * max(spads)
    raw = None
    if opts.get("contiguous"):   # --contiguous=1: a PHYSICALLY contiguous arena (hipExtMallocWithFlags, hipDeviceMallocContiguous):
        import ctypes as C       # the same physical layout in every process, so what is measured is the layout's, not the lottery's

        hip = C.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
        hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        ptr = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(ptr), arena_bytes, 0x4)
        assert rc == 0 and ptr.value, f"contiguous allocation of {arena_bytes >> 20} MiB failed: {rc}"

        class Raw:
            __cuda_array_interface__ = {"shape": (arena_bytes,), "typestr": "|u1", "data": (ptr.value, False), "version": 2}

        raw = Raw()
        arena = torch.as_tensor(raw, device=dev).view(storage.torch_dtype(np_dtype))
        arena.zero_()
        print(f"contiguous arena {arena_bytes >> 20} MiB at {ptr.value:#x}")
    else:
        arena = torch.zeros(arena_bytes // item, dtype=storage.torch_dtype(np_dtype), device=dev)
    base0 = (-arena.data_ptr()) % two_mb
    stream = torch.cuda.current_stream().cuda_stream
    qsat_src = storage.zeros(nx, nz, np_dtype, dev)
    getattr(lib, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, nx, s["f_ap"].data_ptr(), s["f_t"].data_ptr(),
                                                 qsat_src.data_ptr(), stream)
    src = {n: (s["f_" + n] if n != "qsat" else storage.klayout(qsat_src)) for n in NL_IN}
    fn = getattr(lib, "cloudsc2_nl_" + sfx)

    def run(pad, extra, spad=0):
        ls = nx + pad
        fbytes = (nz + 1) * ls * item
        slab = (fbytes + 65536 + two_mb - 1) // two_mb * two_mb + extra * two_mb + spad
        views = []
        for i in range(nfields):
            o = (base0 + i * slab + (i * 2304) % 65536) // item
            views.append(arena[o:o + (nz + 1) * ls].view(nz + 1, ls)[:, :nx])
        for n, v in zip(NL_IN, views):
            v.copy_(src[n])
        for v in views[len(NL_IN):]:
            v.zero_()            # rows the stencil leaves untouched (level nz of full-level outputs) must not carry an older layout's bytes
        pin = _lib.ptr_array([v.data_ptr() for v in views[:len(NL_IN)]])
        pout = _lib.ptr_array([v.data_ptr() for v in views[len(NL_IN):]])
        for _ in range(3):
            rc = fn(ctypes.byref(p), nx, nz, ls, pin, eta.data_ptr(), pout, 3600.0, stream)
            assert rc == 0, _lib.last_error()
        ts = []
        for _ in range(rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn(ctypes.byref(p), nx, nz, ls, pin, eta.data_ptr(), pout, 3600.0, stream)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 5)
        return float(np.median(ts)) * 1e3, _lib.last_kernel(), [v.clone() for v in views[len(NL_IN):]]

    print(f"cloudsc2_nl {prec} {nx} columns: level pitch = nx + pad elements ({item} B each); 2-MB slabs + extra x 2 MB, stagger 2304")
    for _ in range(40):      # reach steady clocks
        run(0, 0)
    ref = None
    for extra in extras:
        for spad in spads:
            for pad in pads + [pads[0]]:
                t, kern, outs = run(pad, extra, spad)
                if ref is None:
                    ref = outs
                same = all(bool(torch.equal(a, b)) for a, b in zip(ref, outs))
                print(f"  extra {extra:2d}  spacing +{spad:8d} B  pitch pad {pad:6d} ({pad * item:7d} B)  {t:8.1f} us   "
                      f"{3567 * item * nx / t / 1e3:7.1f} GB/s   {kern}  results equal: {same}", flush=True)


if __name__ == "__main__":
    main()
