#!/usr/bin/env python3
"""Worst error of the fp32 HIP kernels against the executed-reference fp32 fixtures (tests/golden/reference_exec_f32.npz),
as a fraction of each column's scale - the evidence behind the bounds of tests/test_reference_exec_f32.py.
  python profiles/measure_fp32_errors.py > profiles/r02/fp32_errors.txt   (on the GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worst(got, want):
    scale = np.abs(want.astype(np.float64)).max(axis=0, keepdims=True)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return float(np.max(err / (scale + 1e-300))), float(err.max() / (np.abs(want).max() + 1e-300))


def main():
    import torch

    import __graft_entry__ as ge

    ge.build()
    from helpers import NL_IN, NL_OUT, externals, nlev_of
    from test_hip_nl import run_hip_nl
    from test_hip_tl_ad import run_hip_ad, run_hip_tl

    dev = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_exec_f32.npz"))
    NZ = 137
    fields = {"in_" + n: g["in_" + n] for n in NL_IN}
    eta, dt = g["eta"], float(g["dt"])
    nx = fields["in_ap"].shape[1]
    print(f"{torch.cuda.get_device_name(0)}; fraction of the column scale | of the field scale")
    for tag, fl in (("nl", {}), ("nl_evap", dict(LEVAPLS2=True))):
        got = run_hip_nl(fields, eta, dt, externals(**fl), dev, nx, NZ)
        for n in NL_OUT:
            k = nlev_of(n, NZ)
            print(f"{tag:10s} out_{n:12s} %.2e | %.2e" % worst(got[n][:k], g[f"{tag}_out_{n}"][:k]))
    for tag, fl, inc, d in (("tl_noreg", dict(LREGCL=False), "inc", dt), ("tl_sym", {}, "inc_nosupsat", dt),
                            ("evap60_tl", dict(LEVAPLS2=True), "evap60_inc", 60.0)):
        fi = {"in_" + n + "_i": g[f"{inc}_{n}_i"] for n in NL_IN}
        got, got_i = run_hip_tl(fields, fi, eta, d, externals(NLEV=NZ, **fl), dev, nx, NZ)
        for n in NL_OUT:
            k = nlev_of(n, NZ)
            print(f"{tag:10s} out_{n:12s} %.2e | %.2e" % worst(got[n][:k], g[f"{tag}_out_{n}"][:k]),
                  f"  out_{n}_i %.2e | %.2e" % worst(got_i[n][:k], g[f"{tag}_out_{n}_i"][:k]))
    for tag, fl, tl_tag, d in (("ad", {}, "tl_sym", dt), ("evap60_ad", dict(LEVAPLS2=True), "evap60_tl", 60.0)):
        forcing = {n: g[f"{tl_tag}_out_{n}_i"] for n in NL_OUT}
        got, got_i = run_hip_ad(fields, forcing, eta, d, externals(NLEV=NZ, **fl), dev, nx, NZ)
        for n in NL_OUT:
            k = nlev_of(n, NZ)
            print(f"{tag:10s} out_{n:12s} %.2e | %.2e" % worst(got[n][:k], g[f"{tag}_out_{n}"][:k]))
        for n in NL_IN:
            k = 138 if n in ("aph", "lu") else 137
            print(f"{tag:10s} out_{n + '_i':14s} %.2e | %.2e" % worst(got_i[n][:k], g[f"{tag}_out_{n}_i"][:k]))


if __name__ == "__main__":
    main()
