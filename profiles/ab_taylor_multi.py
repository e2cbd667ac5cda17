#!/usr/bin/env python3
"""A/B of `cloudsc2_nl_taylor_multi` builds (step sizes per launch: -DCS2_NL_MULTI_NF=...) in ONE process, interleaved:
  python profiles/ab_taylor_multi.py nf5=gt4py_dwarf_p_cloudsc2_tl_ad_amd/libcloudsc2_hip.so nf3=build/variants/lib_nf3.so ...
Times the ten step sizes of the Taylor test (2 launches at NF = 5) and, for reference, ten launches of the one-step kernel."""
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
    nx, rounds = int(opts.get("cols", 65536)), int(opts.get("rounds", 9))
    prec = opts.get("precision", "double")
    np_dtype, sfx = (np.float64, "f64") if prec == "double" else (np.float32, "f32")
    nz, dev = 137, torch.device("cuda:0")
    libs = {}
    for a in args:
        name, path = a.split("=", 1)
        lib = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
        _lib._declare(lib)
        libs[name] = lib
    first = next(iter(libs.values()))
    p = make_params(dict(default_externals(), NLEV=nz))
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    Z = lambda: storage.zeros(nx, nz, np_dtype, dev)  # noqa: E731
    f = {k[2:]: storage.from_klayout(v, np_dtype, dev) for k, v in s.items()}
    f["qsat"] = Z()
    P = lambda d, names: _lib.ptr_array([d[n].data_ptr() for n in names])  # noqa: E731
    assert getattr(first, "cloudsc2_saturation_" + sfx)(ctypes.byref(p), nx, nz, nx, f["ap"].data_ptr(), f["t"].data_ptr(),
                                                       f["qsat"].data_ptr(), stream) == 0
    fi = {n: Z() for n in NL_IN}
    for n in NL_IN:
        storage.klayout(fi[n]).copy_(0.01 * storage.klayout(f[n]))
    ref = {n: Z() for n in NL_OUT}
    assert getattr(first, "cloudsc2_nl_" + sfx)(ctypes.byref(p), nx, nz, nx, P(f, NL_IN), eta.data_ptr(), P(ref, NL_OUT),
                                                3600.0, stream) == 0
    f2s = [10.0 ** -(i + 1) for i in range(10)]
    pf = (ctypes.c_double * 10)(*f2s)
    nb = first.cloudsc2_nl_taylor_blocks(nx)
    part = torch.zeros((nb, 10, 10), dtype=torch.float64, device=dev)
    part1 = torch.zeros((nb, 10), dtype=torch.float64, device=dev)

    def multi(lib):
        rc = getattr(lib, "cloudsc2_nl_taylor_multi_" + sfx)(ctypes.byref(p), nx, nz, nx, P(f, NL_IN), P(fi, NL_IN), 0.0, 10, pf,
                                                             eta.data_ptr(), P(ref, NL_OUT), part.data_ptr(), 3600.0, stream)
        assert rc == 0, lib.cloudsc2_last_error()

    def multi_inc(lib):
        rc = getattr(lib, "cloudsc2_nl_taylor_multi_" + sfx)(ctypes.byref(p), nx, nz, nx, P(f, NL_IN), None, 0.01, 10, pf,
                                                             eta.data_ptr(), P(ref, NL_OUT), part.data_ptr(), 3600.0, stream)
        assert rc == 0, lib.cloudsc2_last_error()

    def single(lib):
        for f2 in f2s:
            rc = getattr(lib, "cloudsc2_nl_taylor_" + sfx)(ctypes.byref(p), nx, nz, nx, P(f, NL_IN), P(fi, NL_IN), f2,
                                                           eta.data_ptr(), P(ref, NL_OUT), part1.data_ptr(), 3600.0, stream)
            assert rc == 0

    sums = {}
    for name, lib in libs.items():
        for _ in range(2):
            multi(lib)
        torch.cuda.synchronize()
        sums[name] = part.sum(dim=0).cpu().numpy()
    base = next(iter(sums.values()))
    for name, v in sums.items():
        print(f"{name:>10s}: max |sum - first build's| / max|sum| = {np.abs(v - base).max() / np.abs(base).max():.2e}")
    times = {n: [] for n in libs}
    times.update({n + "+inc": [] for n in libs})
    times.update({n + " one-step x10": [] for n in libs})
    for _ in range(rounds):
        for name, lib in (list(libs.items()) + [(n + "+inc", l) for n, l in libs.items()]
                          + [(n + " one-step x10", l) for n, l in libs.items()]):
            fn = single if name.endswith("one-step x10") else (multi_inc if name.endswith("+inc") else multi)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                fn(lib)
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b) / 3)
    print(f"ten step sizes of the Taylor test, {prec} {nx} columns, {rounds} interleaved rounds x 3:")
    for name, ts in times.items():
        ts = sorted(ts)
        print(f"{name:>22s}: median {ts[len(ts) // 2]:8.3f} ms  min {ts[0]:8.3f} ms  ({ts[len(ts) // 2] / 10 * 1e3:7.1f} us per step size)")


if __name__ == "__main__":
    main()
