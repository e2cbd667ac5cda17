#!/usr/bin/env python3
"""Is the placement lottery a matter of PHYSICAL contiguity (page-table fragment size -> TLB reach)?  The same dense
placement of the 26 cloudsc2_nl fields (2-MB slabs, stagger 2304) in (a) a torch allocation, (b) a
hipExtMallocWithFlags(hipDeviceMallocContiguous) allocation, (c) a plain hipMalloc, alternately; NL kernel by HIP events.
  python profiles/contig_probe.py [--cols=65536] [--precision=double] [--rounds=5]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class RawDevice:
    """a device allocation exposed to torch through __cuda_array_interface__ (no ownership transfer)"""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


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
    np_dtype = np.float64 if prec == "double" else np.float32
    tdt = storage.torch_dtype(np_dtype)
    sfx = "f64" if prec == "double" else "f32"
    item = np.dtype(np_dtype).itemsize
    nz, dev = 137, torch.device("cuda:0")
    lib = _lib.load()
    hip_path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
    hip = ctypes.CDLL(hip_path)
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    p = make_params(dict(default_externals(), NLEV=nz))
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    qsat_src = storage.zeros(nx, nz, np_dtype, dev)
    getattr(lib, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, nx, s["f_ap"].data_ptr(), s["f_t"].data_ptr(),
                                                 qsat_src.data_ptr(), stream)
    src = {n: (s["f_" + n] if n != "qsat" else storage.klayout(qsat_src)) for n in NL_IN}
    fn = getattr(lib, "cloudsc2_nl_" + sfx)
    nfields = len(NL_IN) + len(NL_OUT)
    two_mb = 2 << 20
    fbytes = (nz + 1) * nx * item
    slab = (fbytes + 65536 + two_mb - 1) // two_mb * two_mb
    need = nfields * slab + 2 * two_mb
    print(f"cloudsc2_nl {prec} {nx} columns; 26 fields dense on 2-MB slabs, stagger 2304; arena {need >> 20} MiB; {hip_path}")

    def views_of(flat_u8):
        base = (-flat_u8.data_ptr()) % two_mb
        vs = []
        for i in range(nfields):
            o = base + i * slab + (i * 2304) % 65536
            vs.append(flat_u8[o:o + fbytes].view(tdt).view(nz + 1, nx))
        for n, v in zip(NL_IN, vs):
            v.copy_(src[n])
        for v in vs[len(NL_IN):]:
            v.zero_()
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

    if "scan" in opts:           # --scan=1: slab spacing +e x 2 MB, e = 0..63, in TWO contiguous arenas and one torch arena:
        # is the best spacing the same in every physically contiguous allocation (then it is a rule, not a lottery)?
        emax = 63
        big = nfields * (slab + emax * two_mb) + 2 * two_mb
        ar = {"torch": torch.zeros(big, dtype=torch.uint8, device=dev)}
        raws = []
        for name in ("contiguous A", "contiguous B"):
            ptr = ctypes.c_void_p()
            rc = hip.hipExtMallocWithFlags(ctypes.byref(ptr), big, 0x4)
            print(f"  {name}: rc {rc}, ptr {ptr.value and hex(ptr.value)}, {big >> 20} MiB")
            if rc == 0 and ptr.value:
                raws.append(RawDevice(ptr.value, big))
                ar[name] = torch.as_tensor(raws[-1], device=dev)

        def views_e(flat_u8, e):
            base = (-flat_u8.data_ptr()) % two_mb
            vs = []
            for i in range(nfields):
                o = base + i * (slab + e * two_mb) + (i * 2304) % 65536
                vs.append(flat_u8[o:o + fbytes].view(tdt).view(nz + 1, nx))
            for n, v in zip(NL_IN, vs):
                v.copy_(src[n])
            return vs

        for _ in range(60):
            timed(views_e(ar["torch"], 0))
        res = {k: [] for k in ar}
        for e in range(emax + 1):
            row = []
            for k, a in ar.items():
                t = timed(views_e(a, e))
                res[k].append(t)
                row.append(f"{k} {t:7.1f}")
            print(f"  +{e:2d} x 2 MB   " + "   ".join(row), flush=True)
        ks = list(res)
        for i in range(len(ks)):
            for j in range(i + 1, len(ks)):
                print(f"  correlation {ks[i]} ~ {ks[j]}: {np.corrcoef(res[ks[i]], res[ks[j]])[0, 1]:+.3f}")
        for k in ks:
            o = np.argsort(res[k])[:5]
            print(f"  {k}: fastest spacings " + ", ".join(f"+{int(e)} ({res[k][e]:.1f} us)" for e in o) + f"; median {np.median(res[k]):.1f} us")
        torch.cuda.synchronize()
        del ar
        for raw in raws:
            hip.hipFree(ctypes.c_void_p(raw.ptr))
        return
    arenas = {}
    arenas["torch"] = torch.zeros(need, dtype=torch.uint8, device=dev)
    raws = []
    for name, flag in (("hipMalloc", None), ("contiguous", 0x4), ("contiguous #2", 0x4)):
        ptr = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(ptr), need) if flag is None else hip.hipExtMallocWithFlags(ctypes.byref(ptr), need, flag)
        print(f"  {name}: rc {rc}, ptr {ptr.value and hex(ptr.value)}")
        if rc == 0 and ptr.value:
            raw = RawDevice(ptr.value, need)
            raws.append(raw)
            arenas[name] = torch.as_tensor(raw, device=dev)
    views = {k: views_of(a) for k, a in arenas.items()}
    for _ in range(60):
        timed(views["torch"])
    ref = None
    for rnd in range(3):
        for k, vs in views.items():
            t = timed(vs)
            outs = [v.clone() for v in vs[len(NL_IN):]]
            if ref is None:
                ref = outs
            same = all(bool(torch.equal(a, b)) for a, b in zip(ref, outs))
            print(f"  round {rnd}  {k:14s} {t:8.1f} us   {3567 * item * nx / t / 1e3:7.1f} GB/s   results equal: {same}", flush=True)
    torch.cuda.synchronize()
    del views, arenas
    for raw in raws:
        hip.hipFree(ctypes.c_void_p(raw.ptr))


if __name__ == "__main__":
    main()
