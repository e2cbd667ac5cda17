#!/usr/bin/env python3
"""How are the placements `storage.tune_placement` tries distributed - is a fast one rare or common, and does it exist on
every lease?  Times EVERY candidate of the tuner's grid (64 spacings x 2 staggers x up to 4 shifts) for the
(saturation, cloudsc2_nl) step and prints the distribution of the first-pass times.
  python profiles/placement_distribution.py [--cols=524288] [--precision=single]"""
import os
import sys

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
    nx = int(opts.get("cols", 524288))
    np_dtype = np.float64 if opts.get("precision", "single") == "double" else np.float32
    nz, dev, dt = 137, torch.device("cuda:0"), 3600.0
    ext = default_externals()
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    sat = compile_stencil("saturation", ext)
    nl = compile_stencil("cloudsc2_nl", ext)

    def step(F):
        sat(in_ap=F["in_ap"], in_t=F["in_t"], out_qsat=F["in_qsat"], domain=(nx, 1, nz), **com)
        nl(**F, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)

    order = ["in_" + n for n in NL_IN] + ["out_" + n for n in NL_OUT]
    src = {"in_" + k[2:]: v for k, v in s.items()}
    kw = {}
    if "shifts_gb" in opts:      # --shifts_gb=0,8,16,...  --spacings=0,16,32,...: a wider / coarser grid than the tuner's default
        kw.update(shifts_mb=tuple(int(x) * 1024 for x in opts["shifts_gb"].split(",")), max_arena_bytes=230 << 30,
                  max_shift_spans=1e9)
    if "spacings" in opts:
        kw.update(spacings=tuple(int(x) for x in opts["spacings"].split(",")))
    F, rep = storage.tune_placement(nx, nz, np_dtype, dev, order, src, step, budget_s=float(opts.get("budget", 120)),
                                    keep_all=True, **kw)
    allc = rep.pop("first_pass_all")
    ts = np.array([c[0] for c in allc])
    print(f"(saturation, cloudsc2_nl) {np.dtype(np_dtype).name} {nx} columns: {len(allc)} placements, step ms "
          f"min {ts.min():.4f}  p5 {np.percentile(ts, 5):.4f}  median {np.median(ts):.4f}  p95 {np.percentile(ts, 95):.4f}  max {ts.max():.4f}; "
          f"default {rep['first_pass_default_ms']:.4f}")
    lo, hi = ts.min(), ts.max()
    edges = np.linspace(lo, hi, 13)
    hist, _ = np.histogram(ts, bins=edges)
    for a, b, h in zip(edges[:-1], edges[1:], hist):
        print(f"  {a:8.4f} - {b:8.4f} ms  {h:4d}  " + "#" * int(60 * h / max(hist.max(), 1)))
    for sh in sorted({c[3] for c in allc}):
        sub = np.array([c[0] for c in allc if c[3] == sh])
        print(f"  shift {sh:6d} MB: min {sub.min():.4f}  median {np.median(sub):.4f}  ({len(sub)} placements)")
    for st in sorted({c[2] for c in allc}):
        sub = np.array([c[0] for c in allc if c[2] == st])
        print(f"  stagger {st:5d} B: min {sub.min():.4f}  median {np.median(sub):.4f}")
    print("  fastest ten:", sorted(allc)[:10])
    print("  report:", {k: rep[k] for k in ("candidates", "default_ms", "tuned_ms", "extra_spacing_x2MB", "stagger_bytes", "shift_MB")})


if __name__ == "__main__":
    main()
