#!/usr/bin/env python3
"""Field placement A/B in ONE process: the same state as (a) one `torch` allocation per field (`CLOUDSC2_FIELD_ARENA=0`, what
round 1 ran on) and (b) slabs of a `storage.FieldArena` (2 MB slab starts + i x 2 304 B stagger, the default now), several
independent instances of each, timed alternately: the `saturation` + `cloudsc2_nl` step as bench.py runs it, cloudsc2_tl and
cloudsc2_ad launches (HIP events, median over rounds).
  python profiles/ab_placement.py [--cols=65536] [--precision=double] [--instances=3] > profiles/r02/placement_ab.txt"""
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
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, NL_IN, NL_OUT, compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    nx = int(opts.get("cols", 65536))
    ninst = int(opts.get("instances", 3))
    rounds = int(opts.get("rounds", 6))
    np_dtype = np.float64 if opts.get("precision", "double") == "double" else np.float32
    nz, dev, dt = 137, torch.device("cuda:0"), 3600.0
    ext = dict(default_externals(), NLEV=nz)
    s = make_state(nx, nz, dtype=np_dtype, device=dev)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=dev)
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    sat = compile_stencil("saturation", ext)
    nl = compile_stencil("cloudsc2_nl", ext)
    tl = compile_stencil("cloudsc2_tl", ext)
    ad = compile_stencil("cloudsc2_ad", ext)
    inc = compile_stencil("state_increment", {"IGNORE_SUPSAT": True})

    def instance(arena):
        storage.set_arena_capacity(64 if arena else 0)       # 64 slabs: all fields of an instance in ONE arena
        Z = lambda: storage.zeros(nx, nz, np_dtype, dev)  # noqa: E731
        f = {"in_" + k[2:]: storage.from_klayout(v, np_dtype, dev) for k, v in s.items()}
        f["in_qsat"] = Z()
        out = {"out_" + n: Z() for n in NL_OUT}
        fi = {"out_" + n + "_i": Z() for n in INC}
        out_i = {"out_" + n + "_i": Z() for n in NL_OUT}
        adj = {"out_" + n + "_i": Z() for n in NL_IN}
        sat(in_ap=f["in_ap"], in_t=f["in_t"], out_qsat=f["in_qsat"], domain=(nx, 1, nz), **com)
        inc(**{"in_" + n: f["in_" + n] for n in INC}, **fi, f=0.01, domain=(nx, 1, nz + 1), **com)
        fin = {"in_" + n + "_i": fi["out_" + n + "_i"] for n in NL_IN}
        frc = {"in_" + n + "_i": out_i["out_" + n + "_i"] for n in NL_OUT}
        calls = {
            "sat+nl step": lambda: (sat(in_ap=f["in_ap"], in_t=f["in_t"], out_qsat=f["in_qsat"], domain=(nx, 1, nz), **com),
                                    nl(**f, **out, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)),
            "cloudsc2_nl": lambda: nl(**f, **out, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com),
            "cloudsc2_tl": lambda: tl(**f, **fin, **out, **out_i, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com),
            "cloudsc2_ad": lambda: ad(**f, **frc, **out, **adj, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com),
        }
        calls["cloudsc2_tl"]()       # the adjoint forcing
        return calls

    insts = []
    for i in range(ninst):
        insts.append((f"separate #{i}", instance(False)))
        insts.append((f"arena    #{i}", instance(True)))
    storage.set_arena_capacity(32)
    for _ in range(60):
        insts[0][1]["sat+nl step"]()
    torch.cuda.synchronize()
    print(f"{nx} columns x {nz} levels, {np.dtype(np_dtype).name}, {torch.cuda.get_device_name(0)}; median of {rounds} rounds x 10 calls, us")
    names = list(insts[0][1])
    res = {(lab, n): [] for lab, _ in insts for n in names}
    for r in range(rounds):
        for n in names:
            for lab, calls in insts:
                fn = calls[n]
                fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    fn()
                b.record()
                torch.cuda.synchronize()
                res[(lab, n)].append(a.elapsed_time(b) / 10 * 1e3)
    print(f"{'':14s}" + "".join(f"{n:>14s}" for n in names))
    for lab, _ in insts:
        print(f"{lab:14s}" + "".join(f"{np.median(res[(lab, n)]):14.1f}" for n in names))
    for kind in ("separate", "arena"):
        print(f"{kind + ' mean':14s}" + "".join(
            f"{np.mean([np.median(res[(lab, n)]) for lab, _ in insts if lab.startswith(kind)]):14.1f}" for n in names))


if __name__ == "__main__":
    main()
