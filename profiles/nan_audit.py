#!/usr/bin/env python3
"""Non-finite values at BASELINE configs[4] scale: cloudsc2_nl / _tl / _ad in fp32 on 524 288 synthetic columns (the
per-GPU shard on 8 GPUs).  For every column in which a HIP kernel produces a NaN / inf, the fp32 ORACLE is run on the same
column: a non-finite value the oracle also produces is a property of the reference's formulas in single precision (e.g.
a division by a cloud-fraction term that rounds to 0), one it does not produce would be a kernel defect.
  python profiles/nan_audit.py [--cols=524288] > profiles/r02/nan_audit_f32.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch

    import __graft_entry__ as ge

    ge.build()
    from helpers import NL_IN, NL_OUT, externals, run_oracle_ad, run_oracle_nl, run_oracle_tl
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    nx = int(opts.get("cols", 524288))
    prec = opts.get("precision", "single")
    dt_np = np.float32 if prec == "single" else np.float64
    nz, dev, dt = 137, torch.device("cuda:0"), 3600.0
    ext = externals(NLEV=nz)
    parts = [make_state(nx, nz, col0=c, ncols=min(131072, nx - c), dtype=dt_np, device=dev) for c in range(0, nx, 131072)]
    s = {k: torch.cat([p[k] for p in parts], dim=1) for k in parts[0]}
    del parts
    eta = torch.as_tensor(eta_levels(nz, dtype=dt_np), device=dev)
    Z = lambda: storage.zeros(nx, nz, dt_np, dev)  # noqa: E731
    f = {"in_" + k[2:]: storage.logical_view(v) for k, v in s.items()}
    f["in_qsat"] = Z()
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    compile_stencil("saturation", ext)(in_ap=f["in_ap"], in_t=f["in_t"], out_qsat=f["in_qsat"], domain=(nx, 1, nz), **com)
    inc = {"out_" + n + "_i": Z() for n in INC}
    compile_stencil("state_increment", {"IGNORE_SUPSAT": True})(**{"in_" + n: f["in_" + n] for n in INC}, **inc, f=0.01,
                                                                 domain=(nx, 1, nz + 1), **com)
    fi = {"in_" + n + "_i": inc["out_" + n + "_i"] for n in NL_IN}
    nl_out = {"out_" + n: Z() for n in NL_OUT}
    compile_stencil("cloudsc2_nl", ext)(**f, **nl_out, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)
    tl_out = {"out_" + n + sfx: Z() for n in NL_OUT for sfx in ("", "_i")}
    compile_stencil("cloudsc2_tl", ext)(**f, **fi, **tl_out, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)
    ad_in = {"in_" + n + "_i": tl_out["out_" + n + "_i"] for n in NL_OUT}
    ad_out = {"out_" + n: Z() for n in NL_OUT}
    ad_out.update({"out_" + n + "_i": Z() for n in NL_IN})
    compile_stencil("cloudsc2_ad", ext)(**f, **ad_in, **ad_out, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)
    torch.cuda.synchronize()
    print(f"{nx} columns x {nz} levels, {np.dtype(dt_np).name}, {torch.cuda.get_device_name(0)}")
    bad_cols = {}
    for kind, outs in (("nl", nl_out), ("tl", tl_out), ("ad", ad_out)):
        cols = torch.zeros(nx, dtype=torch.bool, device=dev)
        per_field = {}
        for name, t in outs.items():
            m = ~torch.isfinite(storage.klayout(t))
            if bool(m.any()):
                per_field[name] = int(m.sum())
                cols |= m.any(dim=0)
        idx = torch.nonzero(cols).flatten().cpu().numpy()
        bad_cols[kind] = idx
        print(f"cloudsc2_{kind}: {len(idx)} of {nx} columns hold a non-finite output; points per field: {per_field}")
    eta_h = eta.cpu().numpy()
    for kind in ("tl", "ad"):
        idx = bad_cols[kind][:16]
        if len(idx) == 0:
            continue
        sel = torch.as_tensor(idx, device=dev)
        host = {k: storage.klayout(v)[:, sel].cpu().numpy() for k, v in f.items()}
        host_i = {k: storage.klayout(v)[:, sel].cpu().numpy() for k, v in fi.items()}
        if kind == "tl":
            o, oi = run_oracle_tl(host, host_i, eta_h, dt, ext)
            hip = {n: storage.klayout(tl_out["out_" + n + "_i"])[:, sel].cpu().numpy() for n in NL_OUT}
        else:
            forcing = {n: storage.klayout(tl_out["out_" + n + "_i"])[:, sel].cpu().numpy() for n in NL_OUT}
            o, oi = run_oracle_ad(host, forcing, eta_h, dt, ext)
            hip = {n: storage.klayout(ad_out["out_" + n + "_i"])[:, sel].cpu().numpy() for n in NL_IN}
        for j, c in enumerate(idx):
            hb = sorted(n for n in hip if not np.isfinite(hip[n][:, j]).all())
            ob = sorted(n for n in oi if not np.isfinite(oi[n][:, j]).all())
            print(f"  {kind} column {int(c)}: HIP non-finite in {len(hb)} fields, oracle in {len(ob)} fields"
                  + ("" if hb == ob else f"  (differ: HIP {hb} vs oracle {ob})"))


if __name__ == "__main__":
    main()
