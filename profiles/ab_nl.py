#!/usr/bin/env python3
"""A/B timing of cloudsc2_nl kernel variants in ONE process, interleaved rounds (perf deltas from
separate runs on different boxes are not comparable: DVFS / device spread).

  python profiles/ab_nl.py name1=path/to/lib1.so name2=path/to/lib2.so ... [--cols N] [--rounds R]

Each library is a build of gt4py_dwarf_p_cloudsc2_tl_ad_amd/csrc with different -DCS2_NL_* switches
(see profiles/build_variants.sh).  Prints per-variant median / min kernel time (HIP events).
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals, make_params
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_IN, NL_OUT
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
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
    ext = default_externals()
    p = make_params(ext)
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    arena = int(opts.get("arena", 0))
    ls = nx
    if arena:
        # one allocation [level][slot][column] for all 26 fields of the call: lev_stride = slots * nx
        slots = 26
        ls = slots * nx
        big = torch.zeros((nz + 1, slots, nx), dtype=storage.torch_dtype(np_dtype), device=dev)
        names = [k for k in s] + ["f_qsat"]
        f = {}
        for i, k in enumerate(names):
            if k in s:
                big[:, i, :] = s[k]
            f["in_" + k[2:]] = storage.logical_view(big[:, i, :])
        out_base = len(names)
    else:
        f = {"in_" + k[2:]: storage.logical_view(v) for k, v in s.items()}
        f["in_qsat"] = storage.zeros(nx, nz, np_dtype, dev)
    first = next(iter(libs.values()))
    getattr(first, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, ls, f["in_ap"].data_ptr(), f["in_t"].data_ptr(),
                                   f["in_qsat"].data_ptr(), None)
    if arena:
        outs = {n: storage.logical_view(big[:, out_base + i, :]) for i, n in enumerate(NL_OUT)}
    else:
        outs = {n: storage.zeros(nx, nz, np_dtype, dev) for n in NL_OUT}
    pin = _lib.ptr_array([f["in_" + n].data_ptr() for n in NL_IN])
    pout = _lib.ptr_array([outs[n].data_ptr() for n in NL_OUT])
    stream = torch.cuda.current_stream().cuda_stream

    step = int(opts.get("step", 0))      # --step=1: time the driver's step (saturation + cloudsc2_nl of the SAME library)

    def call(lib):
        if step:
            rc = getattr(lib, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, ls, f["in_ap"].data_ptr(), f["in_t"].data_ptr(),
                                                            f["in_qsat"].data_ptr(), stream)
            assert rc == 0, rc
        rc = getattr(lib, "cloudsc2_nl_" + sfx)(ctypes.byref(p), nx, nz, ls, pin, eta.data_ptr(), pout, 3600.0, stream)
        assert rc == 0, rc

    ref = None
    for name, lib in libs.items():
        for _ in range(3):
            call(lib)
        torch.cuda.synchronize()
        chk = torch.stack([storage.klayout(o)[:nz].double().abs().sum() for o in outs.values()]).cpu().numpy()
        if ref is None:
            ref = chk
        print(f"{name:>12s} checksum rel diff vs first: {np.abs(chk - ref).max() / np.abs(ref).max():.2e}")
    times = {n: [] for n in libs}
    for r in range(rounds):
        for name, lib in libs.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                call(lib)
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b) / 5)
    if step:
        # where a step's time goes: HIP events around each of the two launches of a 40-step train per library
        for name, lib in libs.items():
            evs = []
            for _ in range(42):
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                rc = getattr(lib, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, ls, f["in_ap"].data_ptr(),
                                                                f["in_t"].data_ptr(), f["in_qsat"].data_ptr(), stream)
                e1.record()
                rc |= getattr(lib, "cloudsc2_nl_" + sfx)(ctypes.byref(p), nx, nz, ls, pin, eta.data_ptr(), pout, 3600.0, stream)
                e2.record()
                assert rc == 0
                evs.append((e0, e1, e2))
            torch.cuda.synchronize()
            sat = np.median([a.elapsed_time(b) for a, b, _ in evs[2:]]) * 1e3
            nl_ = np.median([b.elapsed_time(c) for _, b, c in evs[2:]]) * 1e3
            print(f"{name:>12s}: inside the step  saturation {sat:6.1f} us   cloudsc2_nl {nl_:6.1f} us")
    bytes_ = 3567 * np.dtype(np_dtype).itemsize * nx
    for name, t in times.items():
        t = np.array(t)
        print(f"{name:>12s}: median {np.median(t)*1e3:8.1f} us  min {t.min()*1e3:8.1f} us  "
              f"-> {bytes_/np.median(t)/1e6:7.1f} GB/s ({bytes_/np.median(t)/1e6/8000*100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
