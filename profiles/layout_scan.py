#!/usr/bin/env python3
"""How much does the PLACEMENT of the fields of a cloudsc2_nl call in HBM matter, and which placement is best?

ONE arena is allocated once; every candidate places the 26 fields of the call at base_i = i * S inside it (S = the field
size + a padding, or an explicit list of 2-MB-slab + stagger rules), copies the same state in and times the NL kernel
(median of `rounds` x 5 launches by HIP events).  Same physical memory for every candidate, so differences are the
placement's, not the allocation lottery's (profiles/r02/layout_experiment.txt).
  python profiles/layout_scan.py [--cols=65536] [--precision=double] [--rounds=5] > profiles/r02/layout_scan.txt"""
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
    np_dtype = np.float64 if prec == "double" else np.float32
    sfx = "f64" if prec == "double" else "f32"
    item = np.dtype(np_dtype).itemsize
    nz, dev = 137, torch.device("cuda:0")
    lib = _lib.load()
    p = make_params(dict(default_externals(), NLEV=nz))
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    fbytes = (nz + 1) * nx * item
    nfields = len(NL_IN) + len(NL_OUT)
    arena_bytes = nfields * (fbytes + (8 << 20)) + (4 << 20)
    keep_raw = []

    def alloc(nbytes):
        """the arena: a torch allocation, or with --contiguous=1 PHYSICALLY contiguous memory (hipExtMallocWithFlags,
        hipDeviceMallocContiguous) wrapped through __cuda_array_interface__"""
        if not opts.get("contiguous"):
            return torch.zeros(nbytes // item, dtype=storage.torch_dtype(np_dtype), device=dev)
        hip = ctypes.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
        hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        ptr = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(ptr), nbytes, 0x4)
        assert rc == 0 and ptr.value, f"contiguous allocation of {nbytes >> 20} MiB failed: {rc}"

        class Raw:
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr.value, False), "version": 2}

        keep_raw.append(Raw())
        print(f"contiguous arena {nbytes >> 20} MiB at {ptr.value:#x}")
        t = torch.as_tensor(keep_raw[-1], device=dev).view(storage.torch_dtype(np_dtype))
        t.zero_()
        return t

    arena = alloc(arena_bytes)
    base0 = (-arena.data_ptr()) % (2 << 20)              # start the placements on a 2 MB boundary of the address space
    stream = torch.cuda.current_stream().cuda_stream
    qsat_src = storage.zeros(nx, nz, np_dtype, dev)
    getattr(lib, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, nx, s["f_ap"].data_ptr(), s["f_t"].data_ptr(),
                                                 qsat_src.data_ptr(), stream)
    src = {n: (s["f_" + n] if n != "qsat" else storage.klayout(qsat_src)) for n in NL_IN}

    def place(offsets):
        nonlocal arena, base0
        """views at the given byte offsets (relative to the 2-MB-aligned start); returns (in ptrs, out ptrs)"""
        views = []
        for i, off in enumerate(offsets):
            o = (base0 + off) // item
            views.append(arena[o:o + (nz + 1) * nx].view(nz + 1, nx))
        for n, v in zip(NL_IN, views[:len(NL_IN)]):
            v.copy_(src[n])
        return (_lib.ptr_array([v.data_ptr() for v in views[:len(NL_IN)]]),
                _lib.ptr_array([v.data_ptr() for v in views[len(NL_IN):]]))

    def time_layout(offsets):
        assert max(offsets) + fbytes + base0 <= arena_bytes and all(o % 16 == 0 for o in offsets)
        pin, pout = place(offsets)
        fn = getattr(lib, "cloudsc2_nl_" + sfx)
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

    two_mb = 2 << 20
    slab2m = (fbytes + 65536 + two_mb - 1) // two_mb * two_mb
    cands = {}
    for pad in (0, 256, 512, 1024, 2304, 4352, 8448, 16640, 33024, 65792, 131328, 262400, 524544, 1048576, 1048832,
                1050880, 1057024, 1065216, 1081600, 1114368):
        cands[f"dense+{pad}"] = [i * (fbytes + pad) for i in range(nfields)]
    for st in (0, 256, 1280, 2304, 4352, 8448, 12544, 16640, 20736, 33024):
        cands[f"slab2m stagger {st}"] = [i * slab2m + (i * st) % 65536 for i in range(nfields)]
    for extra in (1, 2, 3):      # wider slab spacing (+ extra x 2 MB) with the 2304 / 8448 staggers
        for st in (2304, 8448):
            cands[f"slab2m+{extra}x2MB stagger {st}"] = [i * (slab2m + extra * two_mb) + (i * st) % 65536 for i in range(nfields)]
    if "spacings" in opts:       # second pass: --spacings=a,b,... (multiples of 1 MB between field starts) x --staggers=...
        cands = {"slab2m stagger 0": cands["slab2m stagger 0"]}
        for sp in [int(x) for x in opts["spacings"].split(",")]:
            for st in [int(x) for x in opts.get("staggers", "2304").split(",")]:
                if sp * (1 << 20) >= fbytes + 65536:
                    cands[f"spacing {sp} MB stagger {st}"] = [i * sp * (1 << 20) + (i * st) % 65536 for i in range(nfields)]
    if "shifts" in opts:         # --shifts=a,b,... (MB): the SAME relative placement (2-MB slabs, stagger 2304) moved through the arena
        rel = [i * slab2m + (i * 2304) % 65536 for i in range(nfields)]
        rel192 = [i * 192 * (1 << 20) + (i * 2304) % 65536 for i in range(nfields)]
        cands = {"slab2m stagger 0": cands["slab2m stagger 0"]}
        for sh in [int(x) for x in opts["shifts"].split(",")]:
            cands[f"dense (70 MB) shifted by {sh} MB"] = [o + sh * (1 << 20) for o in rel]
            cands[f"192 MB spacing shifted by {sh} MB"] = [o + sh * (1 << 20) for o in rel192]
    if "random" in opts:         # --random=N: N placements with a random 256-B-granular offset (< 4 MB) per field on 76 MB slabs
        rng = np.random.default_rng(int(opts.get("seed", 1)))
        cands = {"slab2m stagger 0": cands["slab2m stagger 0"],
                 "slab2m stagger 2304": [i * slab2m + (i * 2304) % 65536 for i in range(nfields)]}
        for e in (61, 62):
            cands[f"spacing +{e}x2MB stagger 2304"] = [i * (slab2m + e * two_mb) + (i * 2304) % 65536 for i in range(nfields)]
        for j in range(int(opts["random"])):
            cands[f"random #{j}"] = [i * (slab2m + 2 * two_mb) + int(rng.integers(0, 16384)) * 256 for i in range(nfields)]
    if "split" in opts:          # --split=<GB> [--arena_gb=N]: is there a coarse ABSOLUTE-address structure?  For every multiple X of
        # <GB> GiB (absolute virtual address) inside an N-GiB arena: inputs packed right below X and outputs right above it,
        # against all 26 fields below X and all above X (dense 2-MB slabs, stagger 2304)
        step_b = int(float(opts["split"]) * (1 << 30))
        want = int(opts.get("arena_gb", 80)) << 30
        if want > arena_bytes:
            del arena
            arena_bytes = want
            arena = alloc(arena_bytes)
            base0 = (-arena.data_ptr()) % (2 << 20)
        start = arena.data_ptr() + base0
        print(f"arena {arena_bytes >> 30} GiB at {arena.data_ptr():#x}; placements start at {start:#x}")
        cands = {"slab2m stagger 0": cands["slab2m stagger 0"]}
        nin = len(NL_IN)
        x = (start // step_b + 1) * step_b
        while x + (nfields + 1) * slab2m < arena.data_ptr() + arena_bytes:
            rel = x - start
            if rel >= nfields * slab2m:
                tag = f"{x / (1 << 30):7.2f} GiB"
                cands[f"in | out split at {tag}"] = ([rel - (nin - i) * slab2m + (i * 2304) % 65536 for i in range(nin)]
                                                     + [rel + (i - nin) * slab2m + (i * 2304) % 65536 for i in range(nin, nfields)])
                cands[f"all below        {tag}"] = [rel - (nfields - i) * slab2m + (i * 2304) % 65536 for i in range(nfields)]
                cands[f"all above        {tag}"] = [rel + i * slab2m + (i * 2304) % 65536 for i in range(nfields)]
                cands[f"half | half at   {tag}"] = [rel + (i - nfields // 2) * slab2m + (i * 2304) % 65536 for i in range(nfields)]
            x += step_b
    if "jump" in opts:           # --jump=<first field of the group> [--contiguous=1]: 192 MB spacing, the fields from that index on moved
        # by an extra offset D - imitates the junction between two physical blocks that the fast shifts of a torch arena straddle
        g0 = int(opts["jump"])
        mb = 1 << 20
        base_sp = 192 * mb
        ds = [0] + [k * 256 * mb for k in range(1, 25)] + [2 * mb, 6 * mb, 10 * mb, 18 * mb, 34 * mb, 66 * mb, 130 * mb, 258 * mb, 514 * mb,
                                                          1026 * mb, 2050 * mb, 4098 * mb, 98 * mb, 354 * mb, 866 * mb, 1890 * mb, 3426 * mb]
        cands = {"slab2m stagger 0": cands["slab2m stagger 0"]}
        for d in ds:
            cands[f"fields {g0}.. moved by {d // mb:5d} MB"] = [i * base_sp + (i * 2304) % 65536 + (d if i >= g0 else 0) for i in range(nfields)]
    arena_need = max(max(v) for v in cands.values()) + fbytes + base0 + (1 << 20)
    if arena_need > arena_bytes:
        del arena
        arena_bytes = arena_need
        arena = alloc(arena_bytes)
        base0 = (-arena.data_ptr()) % (2 << 20)
    print(f"cloudsc2_nl {prec} {nx} columns, {torch.cuda.get_device_name(0)}; field {fbytes} B, 2-MB slab {slab2m} B; median of {rounds} x 5 launches")
    ref = "slab2m stagger 0"
    order = [ref] + [k for k in cands if k != ref] + [ref]
    res = []
    for k in order:
        t = time_layout(cands[k])
        res.append((k, t))
        print(f"  {k:34s} {t:8.1f} us   {3567 * item * nx / t / 1e3:7.1f} GB/s", flush=True)
    best = sorted(res, key=lambda r: r[1])[:5]
    print("best:", ", ".join(f"{k} ({t:.1f} us)" for k, t in best))


if __name__ == "__main__":
    main()
