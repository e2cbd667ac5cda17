"""The build's own drivers / components / harnesses (gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers, .physics,
.harness) - the mirror of the reference's L3-L5 for use where the reference checkout is absent.

CPU part: with the test-only oracle backend the mirror must print what the UNMODIFIED reference classes
print on the same inputs (compared number by number when /root/reference is present).
GPU part: BASELINE configs 2, 3, 4 at full size (65 536 columns x 137 levels, fp64) through the HIP backend.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/drivers")


@pytest.fixture(scope="module")
def oracle_numpy_backend():
    import oracle_backend

    oracle_backend.register("numpy")


def _norms(text):
    return [float(x) for x in re.findall(r"norm = ([0-9.eE+-]+)", text)]


def test_mirror_taylor_driver_matches_reference_classes(oracle_numpy_backend, capsys):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_taylor_test

    ctx = run_taylor_test.main(["--backend", "numpy", "--num-cols", "64"])
    out = capsys.readouterr().out
    mine = _norms(out)
    err = np.abs(1 - np.array(mine))
    assert err.min() < 1e-6 and np.all(np.diff(err[:6]) < 0)        # V shape down to the round-off regime
    assert len(mine) == 10
    if HAVE_REF:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_reference_driver.py"),
                            "run_taylor_test.py", "--num-cols", "64"], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        ref = _norms(p.stdout)
        # same kernels (oracle), same inputs; only the reduction order differs (torch vs NumPy sums)
        np.testing.assert_allclose(mine[:8], ref[:8], rtol=1e-7)


def test_mirror_symmetry_driver(oracle_numpy_backend, capsys):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test

    ctx = run_symmetry_test.main(["--backend", "numpy", "--num-cols", "128"])     # stand-in dataset, tiled
    assert ctx["passed"] and ctx["detail"]["max_error_eps"] < 100
    ctx = run_symmetry_test.main(["--backend", "numpy", "--num-cols", "300", "--input", "synthetic"])
    assert not ctx["passed"] and ctx["detail"]["columns_passing"] >= 290        # literal Q4/Q5 semantics
    ctx = run_symmetry_test.main(["--backend", "numpy", "--num-cols", "300", "--input", "synthetic", "--ad-traj-fix"])
    assert ctx["passed"] and ctx["detail"]["columns_passing"] == 300
    assert "The symmetry test passed. HOORAY!" in capsys.readouterr().out


def test_mirror_nonlinear_driver(oracle_numpy_backend, capsys, tmp_path):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear

    ctx = run_nonlinear.main(["--backend", "numpy", "--num-cols", "48", "--num-runs", "2",
                              "--output-csv-file", str(tmp_path / "p.csv")])
    out = capsys.readouterr().out
    assert "Performance: 48 columns, 2 runs" in out and "== Validation:" in out
    assert (tmp_path / "p.csv").exists()
    clc = ctx["diags"]["f_clc"].data
    assert float(clc.min()) >= 0 and float(clc.max()) <= 1


def test_unknown_backend_is_rejected():
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear

    with pytest.raises(ValueError, match="not available"):
        run_nonlinear.main(["--backend", "gt:gpu", "--num-cols", "8"])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_config2_nonlinear_65536(gpu, capsys):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear

    ctx = run_nonlinear.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "5", "--input", "synthetic",
                              "--disable-validation"])
    out = capsys.readouterr().out
    assert "Performance: 65536 columns, 5 runs" in out
    assert np.mean(ctx["runtimes_ms"]) < 5.0
    import torch

    for d in (ctx["tends"], ctx["diags"]):
        for k, v in d.items():
            assert bool(torch.isfinite(v.data.as_subclass(torch.Tensor)).all()), k


@pytest.mark.gpu
def test_config3_taylor_test_65536(gpu, capsys):
    """run_taylor_test protocol at 65 536 columns: (a) the reader path - 100-column stand-in dataset tiled to
    65 536 like the reference tiles its 100-column input.h5; (b) 65 536 DISTINCT mixed-regime columns, where a
    discontinuous scheme always has a few of its 9 million points sitting on a branch threshold: the V shape
    then only emerges once factor2 is small enough that no point flips (norm -> 1 to 1e-6 or better)."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_taylor_test

    ctx = run_taylor_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "2"])
    out = capsys.readouterr().out
    err = np.abs(1 - ctx["norms"])
    assert ">>> Taylor test: Start" in out and "<<< Taylor test: End" in out
    assert err.min() < 1e-6 and np.all(np.diff(err[:6]) < 0), ctx["norms"]
    print(out)
    ctx = run_taylor_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "1", "--input", "synthetic"])
    print(capsys.readouterr().out)
    assert np.abs(1 - ctx["norms"]).min() < 1e-6, ctx["norms"]


@pytest.mark.gpu
def test_config4_symmetry_test_65536(gpu, capsys):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test

    ctx = run_symmetry_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "2"])
    assert ctx["passed"], ctx["detail"]                      # reference-literal AD on the tiled stand-in dataset
    out = capsys.readouterr().out
    assert "The symmetry test passed. HOORAY!" in out
    print(out)
    ctx = run_symmetry_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "1", "--input", "synthetic",
                                  "--ad-traj-fix"])
    assert ctx["passed"], ctx["detail"]                      # 65 536 distinct mixed-regime columns, consistent AD
    ctx = run_symmetry_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "1", "--input", "synthetic"])
    d = ctx["detail"]
    assert d["columns_passing"] / d["columns"] > 0.95, d     # reference-literal AD: only RTT-crossing columns fail
    print(capsys.readouterr().out)


@pytest.mark.gpu
def test_numpy_style_reductions_on_device_fields(gpu):
    """The reference's harnesses call NumPy on `.data` slices (np.sum(field_tl), np.abs(np.sum(a - b)),
    tangent_linear/validation.py:253-261; to_numpy(x)[:, 0, :], adjoint/validation.py:217-220).  With the hip
    backend `.data` is a FieldTensor on the GPU: the reductions must run there and hand back host scalars."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.fields import DataArray, FieldTensor, to_numpy
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.grid import I, J, K

    a = DataArray(storage.zeros(300, 137, np.float64, gpu), (I, J, K))
    b = DataArray(storage.zeros(300, 137, np.float64, gpu), (I, J, K))
    a.data.as_subclass(torch.Tensor).copy_(torch.rand(a.data.shape, dtype=torch.float64, device=gpu))
    b.data.as_subclass(torch.Tensor).copy_(torch.rand(b.data.shape, dtype=torch.float64, device=gpu))
    fa, fb = a.data[:, 0, :], b.data[:, 0, :]
    assert isinstance(fa, FieldTensor) and fa.is_cuda
    want = float(np.sum(to_numpy(a.data)[:, 0, :] - to_numpy(b.data)[:, 0, :]))
    got = np.abs(np.sum(fa - fb))
    assert isinstance(got, float) and abs(got - abs(want)) <= 1e-9 * max(1.0, abs(want))
    den = np.abs(1e-3 * np.sum(fa))
    assert den > 0 and isinstance(den, float)


@pytest.mark.gpu
def test_exec_info_and_stencil_csv_on_hip(gpu, tmp_path, capsys):
    """`exec_info` bookkeeping through the hip stencils (HIP events) and the per-stencil CSV the drivers write
    (run_nonlinear.py:221-232)."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear

    csv = tmp_path / "stencils.csv"
    run_nonlinear.main(["--backend", "hip", "--num-cols", "4096", "--num-runs", "3", "--disable-validation",
                        "--output-csv-file-stencils", str(csv)])
    text = csv.read_text()
    assert "cloudsc2_nl" in text and "saturation" in text
    rows = [r.split(",") for r in text.strip().splitlines()[1:]]
    for r in rows:
        assert int(r[7]) == 3 and 0.0 < float(r[8]) < 50.0          # 3 timed calls each, a sane mean in ms
        assert float(r[9]) > 0.0 and 0.0 < float(r[10]) < 100.0     # algorithmic GB/s, % of the 8 TB/s roofline


@pytest.mark.gpu
def test_fused_driver_variants_match_the_unfused_ones(gpu, capsys):
    """`--fused` (saturation inside NL; perturbation inside NL) are build extensions: same results, fewer launches."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear, run_taylor_test

    a = run_nonlinear.main(["--backend", "hip", "--num-cols", "4096", "--num-runs", "2", "--disable-validation"])
    b = run_nonlinear.main(["--backend", "hip", "--num-cols", "4096", "--num-runs", "2", "--disable-validation", "--fused"])
    for d in ("tends", "diags"):
        for k, v in a[d].items():
            assert torch.equal(v.data, b[d][k].data), k
    t0 = run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--disable-validation"])
    t1 = run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--disable-validation", "--fused"])
    np.testing.assert_allclose(t1["norms"], t0["norms"], rtol=1e-12)
    capsys.readouterr()


@pytest.mark.gpu
def test_taylor_driver_with_fused_norms(gpu, capsys):
    """`--fused-norms`: the whole perturbation step (perturb, NL, difference, sums) is one kernel launch; the verdict
    and the norms are those of the unfused harness (summation order differs, so compare the error |1 - norm| loosely)."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_taylor_test

    t0 = run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--disable-validation"])
    t1 = run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--disable-validation", "--fused-norms"])
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.harness import taylor_verdict

    assert taylor_verdict(t0["norms"])[0] and taylor_verdict(t1["norms"])[0]
    np.testing.assert_allclose(t1["norms"][:6], t0["norms"][:6], rtol=1e-7)
    np.testing.assert_allclose(t1["norms"], t0["norms"], rtol=1e-2)
    capsys.readouterr()


@pytest.mark.gpu
def test_nonlinear_driver_with_hip_graph(gpu, capsys):
    """`--graph`: the timed region replayed from a captured HIP graph gives the same fields as the eager loop."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear

    base = ["--backend", "hip", "--num-cols", "4096", "--num-runs", "3", "--disable-validation"]
    a = run_nonlinear.main(base)
    for extra in (["--graph"], ["--graph", "--fused"]):
        b = run_nonlinear.main(base + extra)
        assert len(b["runtimes_ms"]) == 3
        for d in ("tends", "diags"):
            for k, v in a[d].items():
                assert torch.equal(v.data, b[d][k].data), (extra, k)
    capsys.readouterr()
