"""Writes tests/golden/input_standin.h5: a REAL HDF5 file (written by the HDF5 library through h5py) with exactly the
dataset names, layouts and parameter names the reference reads from its `data/input.h5`
(/root/reference/src/cloudsc2_gt4py/setup.py:28-70: PA, PAP, PAPH, PLU, PLUDE, PMFD, PMFU, PQ, PSUPSAT, PT,
TENDENCY_CML_Q, TENDENCY_CML_T as (K, IJ), PCLV and TENDENCY_CML_CLD as (D5, K, IJ); iox.py:212-244: KLEV, KLON, PTSPHY
and every field of the six parameter models, the YRECLDP_ / YREPHLI_ ones with their prefix).

The VALUES are the build's own 100-column synthetic stand-in (`framework.iox.synthetic_dataset()`: seeded columns,
provisional parameters) - `data/input.h5` itself is a missing blob of the reference checkout.  The file exists so that the
reader path a real `input.h5` would take (HDF5 file -> framework/h5lite.py or h5py -> HDF5GridOperator / HDF5Operator ->
drivers) is exercised on a genuine HDF5 file today: tests/test_input_file.py requires `run_nonlinear --input <this file>`
to be bit-equal to the in-memory stand-in, on the CPU (oracle backend) and on the GPU (hip).

Two interpreters: the package needs torch (this image's /usr/bin/python3), h5py lives only in /opt/conda/bin/python3.9.

    python tests/golden/make_input_standin.py            # stage 1 here, stage 2 in the conda interpreter

The parameter NAMES are taken from the text of the reference's iox.py at generation time (class bodies of the six
pydantic models: `NAME: type`); only names and types are read, values come from `params.default_externals()` (0 / False
for the ~150 the stencils never import, as the in-memory stand-in resolves them).  Data only is committed."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "input_standin.h5")
REF_IOX = "/root/reference/src/cloudsc2_gt4py/iox.py"
H5PY_PYTHON = "/opt/conda/bin/python3.9"
PREFIX = {"YrecldpParams": "YRECLDP_", "YrephliParams": "YREPHLI_"}


def model_fields():
    """[(dataset name, 'float' | 'int' | 'bool')] for every field of the reference's parameter models."""
    out, cls = [], None
    for line in open(REF_IOX):
        m = re.match(r"class (\w+Params)\(BaseModel\):", line)
        if m:
            cls = m.group(1)
            continue
        if line.startswith("class "):
            cls = None
        m = re.match(r"    (\w+): (float|int|bool)\b", line)
        if cls and m:
            out.append((PREFIX.get(cls, "") + m.group(1), m.group(2)))
    return out


def stage1():
    sys.path.insert(0, ROOT)
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.iox import synthetic_dataset

    d = synthetic_dataset()
    data = {}
    for k in ("KLEV", "KLON"):
        data[k] = np.asarray(d[k], dtype=np.int32).reshape(1)
    data["PTSPHY"] = np.asarray(d["PTSPHY"], dtype=np.float64).reshape(1)
    for k in ("PA", "PAP", "PAPH", "PLU", "PLUDE", "PMFD", "PMFU", "PQ", "PSUPSAT", "PT", "TENDENCY_CML_Q",
              "TENDENCY_CML_T", "PCLV", "TENDENCY_CML_CLD"):
        data[k] = np.ascontiguousarray(d[k], dtype=np.float64)
    fields = model_fields()
    assert len(fields) > 150, len(fields)
    for name, kind in fields:
        v = np.asarray(d[name] if name in d else 0.0).reshape(-1)[0]
        # logicals and integers as 32-bit integers, reals as float64: what an HDF5 file written from Fortran holds
        data[name] = (np.array([float(v)], dtype=np.float64) if kind == "float"
                      else np.array([int(bool(v)) if kind == "bool" else int(v)], dtype=np.int32))
    with tempfile.TemporaryDirectory() as tmp:
        npz = os.path.join(tmp, "standin.npz")
        np.savez(npz, **data)
        subprocess.run([H5PY_PYTHON, os.path.abspath(__file__), "--write", npz, OUT], check=True)
    print(OUT, os.path.getsize(OUT), "bytes,", len(data), "datasets")


def stage2(npz, out):
    import h5py

    data = np.load(npz)
    with h5py.File(out, "w", libver="earliest") as f:     # old-style groups, contiguous datasets: the library's defaults
        for k in data.files:
            f.create_dataset(k, data=data[k])


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--write":
        stage2(sys.argv[2], sys.argv[3])
    else:
        stage1()
