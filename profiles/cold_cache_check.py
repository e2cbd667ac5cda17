#!/usr/bin/env python3
"""How much of a cloudsc2_nl launch time depends on what earlier launches left in the 256 MB memory-side cache?
Times the (saturation, cloudsc2_nl) pattern of bench.py with and without a 1 GiB fill between steps (the fill evicts
everything); HIP events around the NL launch only.   python profiles/cold_cache_check.py [lib.so ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_OUT, compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    dev = torch.device("cuda:0")
    nx, nz = 65536, 137
    ext = default_externals()
    s = make_state(nx, nz, device=dev)
    eta = torch.as_tensor(eta_levels(nz), device=dev)
    f = {k: storage.logical_view(v) for k, v in s.items()}
    qsat = storage.zeros(nx, nz, np.float64, dev)
    ins = {"in_" + k[2:]: v for k, v in f.items()}
    ins["in_qsat"] = qsat
    outs = {"out_" + n: storage.zeros(nx, nz, np.float64, dev) for n in NL_OUT}
    sat = compile_stencil("saturation", ext)
    nl = compile_stencil("cloudsc2_nl", ext)
    junk = torch.empty(1 << 27, dtype=torch.float64, device=dev)   # 1 GiB
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    for flush in (False, True):
        evs = []
        for it in range(40):
            if flush:
                junk.fill_(float(it))
            sat(in_ap=f["f_ap"], in_t=f["f_t"], out_qsat=qsat, domain=(nx, 1, nz), **com)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            nl(**ins, **outs, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs[5:])
        print(f"  {'1 GiB fill before every step' if flush else 'bench.py pattern            '}: cloudsc2_nl median {t[len(t) // 2] * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
