"""The C ABI used from a program that knows nothing about Python or PyTorch (examples/nl_from_cpp.cpp): built with
hipcc on the GPU box, run as a child process, and held against the Python path on the very same analytic columns."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import NL_IN, NL_OUT, externals, to_device

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.gpu


def analytic_state(nx, nz=137):
    """Same closed-form profiles as examples/nl_from_cpp.cpp."""
    k = np.arange(nz + 1, dtype=np.float64)[:, None]
    c = np.arange(nx, dtype=np.float64)[None, :]
    sh, sf, x = k / nz, (k + 0.5) / nz, c / nx
    ps = 98000.0 + 4000.0 * x
    full = (k < nz)
    f = {n: np.zeros((nz + 1, nx)) for n in NL_IN}
    f["aph"] = ps * sh * sh * (3.0 - 2.0 * sh) + 0.0 * x
    ap = ps * sf * sf * (3.0 - 2.0 * sf) + 1.0
    t = 215.0 + 75.0 * sf * sf + 6.0 * np.sin(7.0 * x + 3.0 * sf) - 8.0 * x
    es = 611.21 * np.exp(17.502 * (t - 273.16) / (t - 32.19))
    rh = 0.35 + 0.75 * np.sin(5.0 * x + 4.0 * sf) ** 2.0
    ki = np.arange(nz + 1)[:, None]
    f["ap"], f["t"], f["q"] = ap * full, t * full, rh * 0.622 * es / ap * full
    f["ql"] = np.where(ki % 5 == 0, 2e-5 * x, 0.0) * full
    f["qi"] = np.where(ki % 7 == 0, 1e-5 * (1.0 - x), 0.0) * full
    f["lude"] = np.where(ki % 11 == 3, 1e-6 * x, 0.0) * full
    f["lu"] = np.where(ki % 11 == 4, 1e-4 * x, 0.0) * full
    f["mfu"], f["mfd"] = 0.01 * sf * x * full, -0.005 * sf * (1.0 - x) * full
    f["tnd_cml_t"] = 1e-5 * np.sin(9.0 * x + sf) * full
    f["tnd_cml_q"] = 1e-9 * np.cos(4.0 * x + 2.0 * sf) * full
    eta = np.where(np.arange(nz + 1) < nz, (np.arange(nz + 1) + 0.5) / nz, 0.0)
    return {"in_" + n: np.ascontiguousarray(v) for n, v in f.items()}, eta


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_cpp_program_matches_the_python_path(gpu, tmp_path):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx, nz = 320, 137
    exe = str(tmp_path / "nl_from_cpp")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([HIPCC, "-O2", "--offload-arch=gfx950", "-w", os.path.join(ROOT, "examples", "nl_from_cpp.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lcloudsc2_hip", "-Wl,-rpath," + libdir,
                    "-o", exe], check=True, timeout=300)
    ext = externals()
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import Cloudsc2Params

    struct_fields = {n for n, _ in Cloudsc2Params._fields_} - {"NLEV"}
    args = [f"{k}={float(v)!r}" for k, v in ext.items() if k in struct_fields]
    p = subprocess.run([exe, str(nx)] + args, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    got = {ln.split()[0]: (float(ln.split()[1]), float(ln.split()[2])) for ln in p.stdout.strip().splitlines()}
    assert set(got) == set(NL_OUT)

    fields, eta = analytic_state(nx, nz)
    dev = to_device(fields, gpu)
    com = dict(origin=(0, 0, 0), validate_args=True, exec_info=None)
    compile_stencil("saturation", ext)(in_ap=dev["in_ap"], in_t=dev["in_t"], out_qsat=dev["in_qsat"], domain=(nx, 1, nz), **com)
    outs = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl", ext)(**dev, **outs, in_eta=torch.as_tensor(eta, device=gpu), dt=3600.0,
                                         domain=(nx, 1, nz + 1), **com)
    torch.cuda.synchronize()
    assert got["clc"][1] > 0 and got["fplsl"][1] + got["fplsn"][1] > 0      # clouds and precipitation do occur
    for n in NL_OUT:
        o = storage.klayout(outs["out_" + n]).cpu().numpy()
        s, a = float(o.sum()), float(np.abs(o).sum())
        # the inputs agree to an ulp of libm (exp / sin / pow on the two hosts' code paths), the kernels are the same
        assert abs(got[n][1] - a) <= 1e-9 * max(a, 1e-300), (n, got[n], a)
        assert abs(got[n][0] - s) <= 1e-9 * max(a, 1e-300), (n, got[n], s)
