#!/usr/bin/env python3
"""Per-kernel timing of every stencil of the path at a given size (HIP events, back-to-back launches).
  python profiles/bench_kernels.py [--cols 65536] [--precision double]
Prints one line per kernel: mean time, algorithmic GB/s (SURVEY.md 8a word counts), % of the 8 TB/s peak."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORDS = {"saturation": 411, "state_increment": 4416, "perturbed_state": 6624, "cloudsc2_nl": 3567,
         "cloudsc2_tl": 7134, "cloudsc2_ad": 7134,
         # sequences and the fused build extensions that replace them (words = those of the FUSED form)
         "saturation+nl": 3567, "nl_saturation(fused)": 3567,               # 15 in + qsat out + 10 out
         "perturbed+nl": 5759, "nl_perturbed(fused)": 5759}                 # 2 x 16 in + 10 out


def main():
    import numpy as np
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, NL_IN, NL_OUT, compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    nx = int(opts.get("cols", 65536))
    dt_np = np.float64 if opts.get("precision", "double") == "double" else np.float32
    nz, dev = 137, torch.device("cuda:0")
    ext = dict(default_externals(), NLEV=nz)
    s = make_state(nx, nz, dtype=dt_np, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=dt_np), device=dev)
    Z = lambda: storage.zeros(nx, nz, dt_np, dev)  # noqa: E731
    f = {"in_" + k[2:]: storage.logical_view(v) for k, v in s.items()}
    f["in_qsat"] = Z()
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    sat = compile_stencil("saturation", ext)
    calls = {}
    calls["saturation"] = lambda: sat(in_ap=f["in_ap"], in_t=f["in_t"], out_qsat=f["in_qsat"], domain=(nx, 1, nz), **com)
    calls["saturation"]()
    inc_out = {"out_" + n + "_i": Z() for n in INC}
    inc = compile_stencil("state_increment", {"IGNORE_SUPSAT": True})
    calls["state_increment"] = lambda: inc(**{"in_" + n: f["in_" + n] for n in INC}, **inc_out, f=0.01,
                                           domain=(nx, 1, nz + 1), **com)
    calls["state_increment"]()
    fi = {"in_" + n + "_i": inc_out["out_" + n + "_i"] for n in INC}
    per_out = {"out_" + n: Z() for n in INC}
    per = compile_stencil("perturbed_state", {})
    calls["perturbed_state"] = lambda: per(**{"in_" + n: f["in_" + n] for n in INC}, **fi, **per_out, f=1e-3,
                                           domain=(nx, 1, nz + 1), **com)
    nl_out = {"out_" + n: Z() for n in NL_OUT}
    nl = compile_stencil("cloudsc2_nl", ext)
    calls["cloudsc2_nl"] = lambda: nl(**f, **nl_out, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
    calls["saturation+nl"] = lambda: (calls["saturation"](), calls["cloudsc2_nl"]())
    nls = compile_stencil("cloudsc2_nl_saturation", ext)
    f_noq = {k: v for k, v in f.items() if k != "in_qsat"}
    calls["nl_saturation(fused)"] = lambda: nls(**f_noq, out_qsat=f["in_qsat"], **nl_out, in_eta=eta, dt=3600.0,
                                                domain=(nx, 1, nz + 1), **com)
    fp = {"in_" + n: per_out["out_" + n] for n in NL_IN}
    calls["perturbed+nl"] = lambda: (calls["perturbed_state"](),
                                     nl(**fp, **nl_out, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com))
    nlp = compile_stencil("cloudsc2_nl_perturbed", ext)
    fi_nl = {"in_" + n + "_i": inc_out["out_" + n + "_i"] for n in NL_IN}
    calls["nl_perturbed(fused)"] = lambda: nlp(**f, **fi_nl, **nl_out, f=1e-3, in_eta=eta, dt=3600.0,
                                               domain=(nx, 1, nz + 1), **com)
    tl_out = {"out_" + n: Z() for n in NL_OUT}
    tl_out.update({"out_" + n + "_i": Z() for n in NL_OUT})
    tl = compile_stencil("cloudsc2_tl", ext)
    calls["cloudsc2_tl"] = lambda: tl(**f, **fi, **tl_out, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
    calls["cloudsc2_tl"]()
    ad_in = {"in_" + n + "_i": tl_out["out_" + n + "_i"] for n in NL_OUT}
    ad_out = {"out_" + n: Z() for n in NL_OUT}
    ad_out.update({"out_" + n + "_i": Z() for n in NL_IN})
    ad = compile_stencil("cloudsc2_ad", ext)
    calls["cloudsc2_ad"] = lambda: ad(**f, **ad_in, **ad_out, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
    wsize = np.dtype(dt_np).itemsize
    print(f"{nx} columns x {nz} levels, {np.dtype(dt_np).name}, {torch.cuda.get_device_name(0)}")
    for name, fn in calls.items():
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        gbs = WORDS[name] * wsize * nx / (ms * 1e-3) / 1e9
        print(f"  {name:16s} {ms*1e3:9.1f} us   {gbs:7.1f} GB/s algorithmic   {gbs/80:5.1f} % of 8 TB/s")


if __name__ == "__main__":
    main()
