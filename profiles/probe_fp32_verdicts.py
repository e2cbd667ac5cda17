#!/usr/bin/env python3
"""What do the reference's two verdicts say in SINGLE precision - on the HIP kernels and on the fp32 oracle, same columns?
(dev probe behind tests/test_drivers.py::test_validation_drivers_in_single_precision)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_backend
from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test, run_taylor_test

oracle_backend.register("numpy")
for source in ("auto", "synthetic"):
    for cols in (2048,):
        for backend in ("hip", "numpy"):
            t = run_taylor_test.main(["--backend", backend, "--num-cols", str(cols), "--precision", "single", "--input", source])
            print(f"@@ taylor {source} {cols} {backend}: passed={t['passed']} norms={np.array2string(np.asarray(t['norms']), precision=6)}")
            s = run_symmetry_test.main(["--backend", backend, "--num-cols", str(cols), "--precision", "single", "--input", source])
            print(f"@@ symmetry {source} {cols} {backend}: passed={s['passed']} detail={s['detail']}")
for source in ("auto",):
    t = run_taylor_test.main(["--backend", "hip", "--num-cols", "65536", "--precision", "single", "--input", source])
    print(f"@@ taylor {source} 65536 hip: passed={t['passed']} norms={np.array2string(np.asarray(t['norms']), precision=6)}")
    s = run_symmetry_test.main(["--backend", "hip", "--num-cols", "65536", "--precision", "single", "--input", source])
    print(f"@@ symmetry {source} 65536 hip: passed={s['passed']} detail={s['detail']}")
