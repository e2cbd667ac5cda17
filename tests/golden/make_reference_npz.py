"""Convert the reference's golden HDF5 outputs to .npz fixtures.

Run ONCE in the build container with the interpreter that has h5py
(`/opt/conda/bin/python3.9 tests/golden/make_reference_npz.py`); the GPU box has
neither /root/reference nor h5py, so the .npz files are what the tests read.

Source data (data, not code): /root/reference/data/reference_{double,single}.h5
 - outputs of the Fortran CLOUDSC2 NL dwarf for the 100-column / 137-level case,
   named as in /root/reference/src/cloudsc2_gt4py/physics/nonlinear/reference.py:28-55.
The matching inputs (`data/input.h5`) are a missing large blob
(/root/reference/.MISSING_LARGE_BLOBS:1), so these files pin layout + invariants only.
"""
import os
import sys

import h5py
import numpy as np

SRC = "/root/reference/data"
DST = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    for prec in ("double", "single"):
        out = {}
        with h5py.File(os.path.join(SRC, f"reference_{prec}.h5"), "r") as f:
            for name in sorted(f.keys()):
                out[name] = np.asarray(f[name])
        path = os.path.join(DST, f"reference_{prec}.npz")
        np.savez_compressed(path, **out)
        print(path, {k: (v.shape, str(v.dtype)) for k, v in out.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
