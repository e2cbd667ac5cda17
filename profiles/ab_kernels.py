#!/usr/bin/env python3
"""A/B timing of kernel variants of ANY stencil of the path in ONE process, interleaved rounds (deltas from separate
runs on different boxes are not comparable: the pool's boxes differ by more than most steps are worth).

  python profiles/ab_kernels.py <nl|tl|ad> name1=path/to/lib1.so name2=path/to/lib2.so ... [--cols=N] [--rounds=R]
                                [--precision=double|single]

Each library is a build of gt4py_dwarf_p_cloudsc2_tl_ad_amd/csrc with different -D switches (profiles/build_variants.sh;
`base=gt4py_dwarf_p_cloudsc2_tl_ad_amd/libcloudsc2_hip.so` is the shipped one).  Prints a checksum comparison against
the first library (so a variant that changes results shows), then per-variant median / min kernel time (HIP events) and
the algorithmic rate."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORDS = {"nl": 3567, "tl": 7134, "ad": 7134}


def main():
    import numpy as np
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals, make_params
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, NL_IN, NL_OUT
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    which = sys.argv[1]
    args = [a for a in sys.argv[2:] if not a.startswith("--")]
    opts = dict(a[2:].split("=") for a in sys.argv[2:] if a.startswith("--") and "=" in a)
    nx = int(opts.get("cols", 65536))
    rounds = int(opts.get("rounds", 15))
    prec = opts.get("precision", "double")
    np_dtype = np.float64 if prec == "double" else np.float32
    sfx = "f64" if prec == "double" else "f32"
    nz = 137
    dev = torch.device("cuda:0")
    libs = {}
    for a in args:
        name, path = a.split("=", 1)
        lib = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
        _lib._declare(lib)
        libs[name] = lib
    first = next(iter(libs.values()))
    ext = dict(default_externals(), NLEV=nz)
    for k, v in opts.items():
        if k.isupper():
            ext[k] = type(ext.get(k, 0))(int(v))
    p = make_params(ext)
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    # --layouts=a,b,...: field placements to compare (interleaved like the libraries).  "separate" = one torch allocation
    # per field (the default); "arena:<stagger>[:<lspad>]" = ONE big allocation, field i starting <stagger> bytes (a
    # multiple of 16) further into its own slab than field i-1, level stride nx + <lspad> elements.
    layouts = opts.get("layouts", "separate").split(",")
    stream = torch.cuda.current_stream().cuda_stream
    P = lambda d, names: _lib.ptr_array([d[n].data_ptr() for n in names])  # noqa: E731
    item = np.dtype(np_dtype).itemsize

    def make_set(layout):
        layout = layout.split("#")[0]        # "name#tag": the tag only makes repeated layouts distinct sets
        if layout.startswith("separate"):
            # "separate[:<stagger>]": one torch allocation per field, field i starting i * <stagger> bytes (mod 64 KB) into it
            ls = nx
            sep_stagger = int(layout.split(":")[1]) if ":" in layout else 0
            count = [0]

            def Z():
                i = count[0]
                count[0] += 1
                off = (i * sep_stagger) % 65536 // item
                buf = torch.zeros((nz + 1) * nx + 65536 // item, dtype=storage.torch_dtype(np_dtype), device=dev)
                return storage.logical_view(buf[off:off + (nz + 1) * nx].view(nz + 1, nx))
        else:
            parts = layout.split(":")
            stagger = int(parts[1]) if len(parts) > 1 else 0
            ls = nx + (int(parts[2]) if len(parts) > 2 else 0)
            slab = (nz + 1) * ls + stagger // item
            if parts[0] == "arena2m":       # slabs start on 2 MB boundaries of the arena, + i * stagger (mod 64 KB) inside them;
                two_mb = (2 << 20) // item  # "arena2m:<stagger>:<lspad>:<extra>": slab spacing + <extra> x 2 MB
                slab = ((nz + 1) * ls + 65536 // item + two_mb - 1) // two_mb * two_mb
                slab += (int(parts[3]) if len(parts) > 3 else 0) * two_mb
            jitter = None
            if parts[0] == "arenaperm":     # "arenaperm:<seed>:<span>": 2-MB slabs as arena2m (stagger 2304), plus an IRREGULAR extra
                two_mb = (2 << 20) // item  # offset of r_i x 2 MB per field, r_i random in [0, span): no arithmetic progression
                seed, span = int(parts[1]), int(parts[2]) if len(parts) > 2 else 32
                stagger = 2304
                ls = nx
                slab = ((nz + 1) * ls + 65536 // item + two_mb - 1) // two_mb * two_mb + span * two_mb
                jitter = np.random.default_rng(seed).integers(0, span, size=80) * two_mb
            big = torch.zeros(80 * slab, dtype=storage.torch_dtype(np_dtype), device=dev)
            count = [0]

            def Z():
                i = count[0]
                count[0] += 1
                o = i * slab + ((i * stagger) % 65536 // item if parts[0] in ("arena2m", "arenaperm") else 0)
                if jitter is not None:
                    o += int(jitter[i])
                return storage.logical_view(big[o:o + (nz + 1) * ls].view(nz + 1, ls)[:, :nx])
        f = {}
        for k, v in s.items():
            t = Z()
            storage.klayout(t).copy_(v)
            f[k[2:]] = t
        f["qsat"] = Z()
        assert getattr(first, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, ls, f["ap"].data_ptr(), f["t"].data_ptr(),
                                                           f["qsat"].data_ptr(), stream) == 0
        fi = {n: Z() for n in INC}
        pinc = make_params(dict(ext, IGNORE_SUPSAT=True))
        assert getattr(first, "cloudsc2_state_increment_" + sfx)(ctypes.byref(pinc), nx, nz, ls, P(f, INC), P(fi, INC), 0.01, stream) == 0
        out = {n: Z() for n in NL_OUT}
        out_i = {n: Z() for n in NL_OUT}
        adj = {n: Z() for n in NL_IN}
        # TL once with the first library: its perturbation outputs force the adjoint
        assert getattr(first, "cloudsc2_tl_" + sfx)(ctypes.byref(p), nx, nz, ls, P(f, NL_IN), P(fi, NL_IN), eta.data_ptr(),
                                                   P(out, NL_OUT), P(out_i, NL_OUT), 3600.0, stream) == 0
        forcing = {n: Z() for n in NL_OUT}
        for n in NL_OUT:
            storage.klayout(forcing[n]).copy_(storage.klayout(out_i[n]))
        return dict(ls=ls, f=f, fi=fi, out=out, out_i=out_i, adj=adj, forcing=forcing)

    sets = {l: make_set(l) for l in layouts}

    def call(lib, S):
        ls, f, fi, out, out_i, adj, forcing = (S[k] for k in ("ls", "f", "fi", "out", "out_i", "adj", "forcing"))
        if which == "nl":
            rc = getattr(lib, "cloudsc2_nl_" + sfx)(ctypes.byref(p), nx, nz, ls, P(f, NL_IN), eta.data_ptr(), P(out, NL_OUT), 3600.0, stream)
        elif which == "tl":
            rc = getattr(lib, "cloudsc2_tl_" + sfx)(ctypes.byref(p), nx, nz, ls, P(f, NL_IN), P(fi, NL_IN), eta.data_ptr(),
                                                    P(out, NL_OUT), P(out_i, NL_OUT), 3600.0, stream)
        else:
            rc = getattr(lib, "cloudsc2_ad_" + sfx)(ctypes.byref(p), nx, nz, ls, P(f, NL_IN), P(forcing, NL_OUT), eta.data_ptr(),
                                                    P(out, NL_OUT), P(adj, NL_IN), 3600.0, stream)
        assert rc == 0, (rc, lib.cloudsc2_last_error())

    def checksum(S):
        fields = list(S["out"].values()) + (list(S["out_i"].values()) if which == "tl" else []) + \
            (list(S["adj"].values()) if which == "ad" else [])
        return torch.stack([storage.klayout(o)[:nz].double().abs().sum() for o in fields]).cpu().numpy()

    combos = [(ln, lib, sn, S) for ln, lib in libs.items() for sn, S in sets.items()]
    ref = None
    for ln, lib, sn, S in combos:
        for _ in range(3):
            call(lib, S)
        torch.cuda.synchronize()
        chk = checksum(S)
        ref = chk if ref is None else ref
        kern = lib.cloudsc2_last_kernel().decode() if hasattr(lib, "cloudsc2_last_kernel") else "?"
        with np.errstate(all="ignore"):
            print(f"{ln:>14s} {sn:18s} [{kern}] checksum rel diff vs first: {np.max(np.abs(chk - ref) / (np.abs(ref) + 1e-300)):.2e}")
    times = {(ln, sn): [] for ln, _, sn, _ in combos}
    for r in range(rounds):
        for ln, lib, sn, S in combos:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                call(lib, S)
            b.record()
            torch.cuda.synchronize()
            times[(ln, sn)].append(a.elapsed_time(b) / 5)
    nbytes = WORDS[which] * np.dtype(np_dtype).itemsize * nx
    print(f"cloudsc2_{which} {prec} {nx} columns, {torch.cuda.get_device_name(0)}, {rounds} interleaved rounds x 5 launches")
    for (ln, sn), t in times.items():
        t = np.array(t)
        name = ln if len(sets) == 1 else f"{ln} {sn}"
        print(f"{name:>24s}: median {np.median(t)*1e3:8.1f} us  min {t.min()*1e3:8.1f} us  "
              f"-> {nbytes/np.median(t)/1e6:7.1f} GB/s algorithmic ({nbytes/np.median(t)/1e6/80:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
