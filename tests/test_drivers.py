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


def test_taylor_and_symmetry_drivers_write_the_performance_csv(oracle_numpy_backend, capsys, tmp_path):
    """`--output-csv-file` of the two validation drivers (run_taylor_test.py:109-124 variant "tl-<backend>",
    run_symmetry_test.py:106-121 variant "ad-<backend>"): the reference's columns + columns/s, GB/s, % of the roofline"""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test, run_taylor_test

    csv, scsv = tmp_path / "perf.csv", tmp_path / "stencils.csv"
    run_taylor_test.main(["--backend", "numpy", "--num-cols", "32", "--num-runs", "2", "--output-csv-file", str(csv),
                          "--output-csv-file-stencils", str(scsv)])
    run_symmetry_test.main(["--backend", "numpy", "--num-cols", "32", "--num-runs", "2", "--output-csv-file", str(csv),
                            "--output-csv-file-stencils", str(scsv)])
    capsys.readouterr()
    srows = [r.split(",") for r in scsv.read_text().strip().splitlines()[1:]]
    calls = {(r[2], r[6]): int(r[7]) for r in srows}
    assert calls[("tl-numpy", "cloudsc2_nl")] == 22 and calls[("tl-numpy", "perturbed_state")] == 20      # 2 runs x (1 + 10) / x 10
    assert calls[("tl-numpy", "cloudsc2_tl")] == 2 and calls[("ad-numpy", "cloudsc2_ad")] == 2 and calls[("ad-numpy", "state_increment")] == 2
    rows = [r.split(",") for r in csv.read_text().strip().splitlines()]
    assert rows[0][:4] == ["host", "precision", "variant", "num_cols"] and len(rows) == 3
    assert rows[1][2] == "tl-numpy" and rows[2][2] == "ad-numpy" and rows[1][3] == rows[2][3] == "32"
    assert float(rows[1][7]) > 0 and float(rows[2][7]) > 0 and float(rows[1][12]) > 0      # mean ms, algorithmic GB/s
    # bytes behind the GB/s column: 939 504 B per column for the Taylor run, 152 760 B for the symmetry call
    assert float(rows[1][12]) * float(rows[1][7]) * 1e-3 * 1e9 / 32 == pytest.approx(939504, rel=1e-9)
    assert float(rows[2][12]) * float(rows[2][7]) * 1e-3 * 1e9 / 32 == pytest.approx(152760, rel=1e-9)


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
@pytest.mark.parametrize("num_cols", [4096, 65536])
def test_config2_nonlinear_with_golden_comparison(gpu, capsys, num_cols):
    """BASELINE configs[1] through the driver WITH its validation step (run_nonlinear.py:139-147): `validate()` runs on
    the HIP output fields against data/reference_double.h5 (its .npz conversion on the GPU box) through the `f_qv -> f_q`
    mapping.  The golden file's inputs (data/input.h5) are not available, so the comparison of the values says nothing
    about the kernels; asserted is what the inputs cannot change: every reference field finds its counterpart with the
    right shape, `f_covptot` agrees exactly (0 without the evaporation block on both sides), and the HIP outputs obey the
    reference's own identities fhpsn = -RLSTT fplsn, fhpsl = -RLVTT fplsl."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals

    ctx = run_nonlinear.main(["--backend", "hip", "--num-cols", str(num_cols), "--num-runs", "5", "--input", "synthetic"])
    out = capsys.readouterr().out
    print(out)
    assert f"Performance: {num_cols} columns, 5 runs" in out and "== Validation:" in out
    assert np.min(ctx["runtimes_ms"]) < 5.0          # a sane time once the box is warm (the first runs may sit in the clock ramp)
    rep = ctx["validation"]
    assert set(rep) == {"f_qi", "f_ql", "f_qv", "f_t", "f_clc", "f_covptot", "f_fhpsl", "f_fhpsn", "f_fplsl", "f_fplsn"}
    assert rep["f_qv"]["as"] == "f_q"                                   # nonlinear/reference.py:32 vs microphysics.py:106
    for name, r in rep.items():
        assert r["shape"] == r["ref_shape"] == (num_cols, 1, 138), (name, r)
        assert np.isfinite(r["max_abs_err"]), name
        assert f"  {name:12s}" in out                                    # one printed line per field
    assert rep["f_covptot"]["max_abs_err"] == 0.0 and rep["f_covptot"]["ok"]
    for d in (ctx["tends"], ctx["diags"]):
        for k, v in d.items():
            t = v.data.as_subclass(torch.Tensor)
            assert t.is_cuda and bool(torch.isfinite(t).all()), k
    ext = default_externals()
    dg = {k: v.data.as_subclass(torch.Tensor) for k, v in ctx["diags"].items()}
    assert torch.equal(dg["f_fhpsn"], -dg["f_fplsn"] * ext["RLSTT"])
    assert torch.equal(dg["f_fhpsl"], -dg["f_fplsl"] * ext["RLVTT"])
    assert float(dg["f_fplsn"].max()) > 0 and float(dg["f_fplsl"].max()) > 0      # a real signal on both fluxes
    # the golden fields are tiled to the run's columns like any input (100-column file): column j == column j + 100
    ref = ctx["diags_ref"]["f_fplsn"].data.as_subclass(torch.Tensor)
    assert ref.is_cuda and torch.equal(ref[:100], ref[100:200])


@pytest.mark.gpu
def test_config3_taylor_test_65536(gpu, capsys):
    """run_taylor_test protocol at 65 536 columns.  (a) The reader path - the 100-column stand-in dataset tiled to
    65 536 like the reference tiles its 100-column input.h5 - must earn the reference's own verdict, "The test passed
    with penalty ..." (tangent_linear/validation.py:183-217).  (b) 65 536 DISTINCT mixed-regime columns: a
    discontinuous scheme always has a few of its 9 million points sitting on a branch threshold, so the reference's
    scoring rule may say "failed with error 10" there (non-monotone head) although the norm still converges to 1 to
    1e-6 - whatever it says, the oracle must say the same on the same columns (next test)."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_taylor_test

    ctx = run_taylor_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "2"])
    out = capsys.readouterr().out
    err = np.abs(1 - ctx["norms"])
    assert ">>> Taylor test: Start" in out and "<<< Taylor test: End" in out
    assert ctx["passed"] is True and "The test passed with penalty" in out, out
    assert err.min() < 1e-6 and np.all(np.diff(err[:6]) < 0), ctx["norms"]
    ctx = run_taylor_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "1", "--input", "synthetic"])
    out2 = capsys.readouterr().out
    assert np.abs(1 - ctx["norms"]).min() < 1e-6, ctx["norms"]
    assert ("The test passed with penalty" in out2) == bool(ctx["passed"])
    assert ("The test failed with error" in out2) == (not ctx["passed"])
    print(out + out2)


@pytest.mark.gpu
@pytest.mark.parametrize("source,cols", [("auto", 2048), ("synthetic", 2048), ("synthetic", 16384)])
def test_taylor_verdict_on_hip_equals_the_oracles(gpu, capsys, oracle_numpy_backend, source, cols):
    """The reference's Taylor verdict, HIP kernels vs the oracle on the SAME columns (2 048: a size the NumPy oracle
    follows in seconds; 16 384 distinct mixed-regime columns = BASELINE configs[0] size): same verdict
    string, same norms down to the round-off regime."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_taylor_test

    args = ["--num-cols", str(cols), "--num-runs", "1", "--input", source]
    hip = run_taylor_test.main(["--backend", "hip"] + args)
    out_hip = capsys.readouterr().out
    ref = run_taylor_test.main(["--backend", "numpy"] + args)
    out_ref = capsys.readouterr().out
    verdict = lambda o: [l for l in o.splitlines() if l.startswith(("The test passed", "The test failed"))]  # noqa: E731
    assert verdict(out_hip) == verdict(out_ref) and len(verdict(out_hip)) == 1, (verdict(out_hip), verdict(out_ref))
    assert hip["passed"] == ref["passed"]
    if source == "auto":
        assert hip["passed"] is True and "The test passed with penalty" in out_hip
    # the first norms agree to ~1e-8 (different exp / reciprocals); deep in the round-off regime only the error level
    np.testing.assert_allclose(hip["norms"][:6], ref["norms"][:6], rtol=1e-6)
    eh, er = np.abs(1 - hip["norms"]), np.abs(1 - ref["norms"])
    assert abs(np.log10(eh.min()) - np.log10(er.min())) < 1.5
    print(out_hip)


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
    # reference-literal AD on DISTINCT mixed-regime columns: the columns whose saturation adjustment crosses RTT fail
    # (quirks Q4/Q5 - a property of the reference, reproduced by the oracle too), so the reference's verdict here is
    # "failed" and only > 95 % of the columns pass
    assert d["columns_passing"] / d["columns"] > 0.95, d
    out = capsys.readouterr().out
    assert ("The symmetry test failed." in out) == (not ctx["passed"])
    print(out)


@pytest.mark.gpu
def test_symmetry_verdict_on_hip_equals_the_oracles_at_16384_columns(gpu, capsys, oracle_numpy_backend):
    """The "> 95 % of distinct columns" statement of the reference-literal AD, held against the ORACLE on the same 16 384
    mixed-regime columns: the same verdict, the same number of passing columns (a column sitting
    exactly on the 1e4 eps threshold may flip: <= 0.2 % allowed), and with the consistent freezing tests every column
    passes on both sides."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test

    args = ["--num-cols", "16384", "--num-runs", "1", "--input", "synthetic"]
    hip = run_symmetry_test.main(["--backend", "hip"] + args)
    ref = run_symmetry_test.main(["--backend", "numpy"] + args)
    capsys.readouterr()
    assert hip["passed"] == ref["passed"] is False                     # quirks Q4/Q5: a property of the reference
    dh, dr = hip["detail"], ref["detail"]
    assert dh["columns"] == dr["columns"] == 16384
    assert abs(dh["columns_passing"] - dr["columns_passing"]) <= 0.002 * 16384, (dh, dr)
    assert dr["columns_passing"] / dr["columns"] > 0.95
    hip = run_symmetry_test.main(["--backend", "hip", "--ad-traj-fix"] + args)
    ref = run_symmetry_test.main(["--backend", "numpy", "--ad-traj-fix"] + args)
    capsys.readouterr()
    assert hip["passed"] and ref["passed"] and hip["detail"]["columns_passing"] == ref["detail"]["columns_passing"] == 16384


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
def test_stencil_csv_of_the_validation_drivers_on_hip(gpu, tmp_path, capsys):
    """per-stencil CSV (HIP events in exec_info) of run_taylor_test in its fused-all mode and of run_symmetry_test --fused: the
    build extensions appear under their own stencil names with their own byte counts"""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test, run_taylor_test

    csv = tmp_path / "stencils.csv"
    run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--num-runs", "2", "--fused-all",
                          "--output-csv-file-stencils", str(csv)])
    run_symmetry_test.main(["--backend", "hip", "--num-cols", "4096", "--num-runs", "2", "--fused",
                            "--output-csv-file-stencils", str(csv)])
    capsys.readouterr()
    rows = [r.split(",") for r in csv.read_text().strip().splitlines()[1:]]
    calls = {(r[2], r[6]): int(r[7]) for r in rows}
    assert calls[("tl-hip", "cloudsc2_nl_taylor_multi")] == 2 and calls[("tl-hip", "cloudsc2_tl_incremented")] == 2
    assert calls[("tl-hip", "cloudsc2_nl")] == 2 and ("tl-hip", "state_increment") not in calls
    assert calls[("ad-hip", "cloudsc2_tl_incremented")] == 2 and calls[("ad-hip", "cloudsc2_ad_from_trajectory")] == 2
    assert ("ad-hip", "cloudsc2_ad") not in calls               # the timed calls run the adjoint sweep alone (r04)
    for r in rows:
        assert 0.0 < float(r[8]) < 50.0 and float(r[9]) > 0.0


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
    t1 = run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--disable-validation", "--fused-stored"])
    np.testing.assert_allclose(t1["norms"], t0["norms"], rtol=1e-12)
    assert t1["harness"].fused and not t1["harness"].fused_norms and t1["harness"].tends_nl_p      # perturbed outputs kept
    capsys.readouterr()


@pytest.mark.gpu
def test_taylor_driver_with_fused_norms(gpu, capsys):
    """`--fused` (= `--fused-norms`, its older name): the whole perturbation step (perturb, NL, difference, sums) is one kernel
    launch; the verdict and the norms are those of the unfused harness (summation order differs, so compare the error
    |1 - norm| loosely)."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_taylor_test
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.harness import taylor_verdict

    t0 = run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--disable-validation"])
    for flag in ("--fused", "--fused-norms"):
        t1 = run_taylor_test.main(["--backend", "hip", "--num-cols", "4096", "--disable-validation", flag])
        assert t1["harness"].fused_norms and not t1["harness"].tends_nl_p           # nothing stored per step size
        assert taylor_verdict(t0["norms"])[0] and taylor_verdict(t1["norms"])[0]
        assert taylor_verdict(t0["norms"])[1] == taylor_verdict(t1["norms"])[1]     # the reference's verdict string, unchanged
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


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["double", "single"])
def test_eta_levels_on_the_device_match_the_oracle(gpu, precision, capsys):
    """`EtaLevels` (common/diagnostics.py:42-45) as the drivers run it on HIP storages: f_eta[k] =
    ap[column 0, k] / aph[column 0, nz], one device slice operation - bit-equal to the oracle's level loop and to the
    host-side `synthetic.eta_levels` the sharded runs use (global column 0)."""
    import argparse

    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers._common import add_common_options, setup
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels
    from oracle import cloudsc2_numpy as oracle

    ap = argparse.ArgumentParser()
    add_common_options(ap)
    for source in ("synthetic", "auto"):
        args = ap.parse_args(["--backend", "hip", "--num-cols", "333", "--precision", precision, "--input", source])
        ctx = setup(args)
        st = ctx["state"]
        eta = st["f_eta"].data.as_subclass(torch.Tensor)
        assert eta.is_cuda and eta.shape == (138,)
        want = oracle.eta_levels(storage.klayout(st["f_ap"].data).cpu().numpy(), storage.klayout(st["f_aph"].data).cpu().numpy())
        assert want.dtype == (np.float64 if precision == "double" else np.float32)
        assert np.array_equal(eta.cpu().numpy(), want), source
        assert 0.0 < want[0] < 1e-3 and 0.99 < want[136] < 1.0 and want[137] == 0.0 and np.all(np.diff(want[:137]) > 0)
        if source == "synthetic":
            assert np.array_equal(want, eta_levels(137, dtype=want.dtype))
    capsys.readouterr()


@pytest.mark.gpu
def test_nonlinear_driver_with_tuned_field_placement(gpu, capsys):
    """`--tune-placement`: the driver's state, diagnostics and tendencies are re-placed in HBM (storage.tune_placement, the
    timed region as the objective); the fields it ends with are those of the plain run, bit for bit."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_nonlinear

    base = ["--backend", "hip", "--num-cols", "8192", "--num-runs", "3", "--disable-validation", "--input", "synthetic"]
    a = run_nonlinear.main(base)
    b = run_nonlinear.main(base + ["--tune-placement"])
    out = capsys.readouterr().out
    assert "field placement: UNTUNED" in out and " -> TUNED " in out and "cost: a " in out and b["placement"]["fields"] >= 26
    assert b["placement"]["tuned_ms"] <= b["placement"]["default_ms"]
    for d in ("tends", "diags"):
        for k, v in a[d].items():
            assert torch.equal(v.data, b[d][k].data), k


@pytest.mark.gpu
def test_taylor_and_symmetry_drivers_with_tuned_field_placement(gpu, capsys):
    """`--tune-placement` on the two validation drivers: every field of the test (state, increments, perturbed state, NL / TL / AD
    outputs) is re-placed with a whole run as the objective; norms, verdicts and adjoints are those of the plain run, bit for bit."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test, run_taylor_test

    base = ["--backend", "hip", "--num-cols", "4096", "--num-runs", "2", "--input", "synthetic"]
    a = run_taylor_test.main(base)
    b = run_taylor_test.main(base + ["--tune-placement"])
    out = capsys.readouterr().out
    assert out.count("field placement: UNTUNED") == 1 and b["placement"]["fields"] >= 80
    assert np.array_equal(a["norms"], b["norms"]) and a["passed"] == b["passed"]
    c = run_symmetry_test.main(base)
    d = run_symmetry_test.main(base + ["--tune-placement"])
    out = capsys.readouterr().out
    assert out.count("field placement: UNTUNED") == 1 and d["placement"]["fields"] >= 70
    assert c["passed"] == d["passed"] and c["detail"] == d["detail"]
    for k, v in c["state"].items():
        if hasattr(v, "data") and isinstance(v.data, torch.Tensor):
            assert torch.equal(v.data, d["state"][k].data), k


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["auto", "synthetic"])
def test_validation_drivers_in_single_precision_agree_with_the_fp32_oracle(gpu, capsys, oracle_numpy_backend, source):
    """`--precision single` through run_taylor_test / run_symmetry_test (/root/reference/drivers/run_taylor_test.py:146-151,
    :170-176 `with_precision`; run_symmetry_test.py likewise): the fp32 HIP kernels against the fp32 oracle on the SAME
    2 048 columns.  In single precision the Taylor norms leave the convergent regime after two or three step sizes (the
    perturbation sinks below float32 rounding), so the reference's scoring rule sees noise in its tail on BOTH sides; what
    must agree is the convergent head of the norms, the level at which they bottom out, and the verdict's pass / fail."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test, run_taylor_test

    args = ["--num-cols", "2048", "--num-runs", "1", "--input", source, "--precision", "single"]
    hip = run_taylor_test.main(["--backend", "hip"] + args)
    out_hip = capsys.readouterr().out
    ref = run_taylor_test.main(["--backend", "numpy"] + args)
    out_ref = capsys.readouterr().out
    verdict = lambda o: [l for l in o.splitlines() if l.startswith(("The test passed", "The test failed"))]  # noqa: E731
    assert len(verdict(out_hip)) == 1 and len(verdict(out_ref)) == 1
    assert hip["passed"] == ref["passed"], (verdict(out_hip), verdict(out_ref))
    eh, er = np.abs(1 - np.asarray(hip["norms"])), np.abs(1 - np.asarray(ref["norms"]))
    np.testing.assert_allclose(hip["norms"][:2], ref["norms"][:2], rtol=2e-3)
    assert abs(np.log10(eh.min()) - np.log10(er.min())) < 1.5, (eh, er)
    sh = run_symmetry_test.main(["--backend", "hip"] + args)
    sr = run_symmetry_test.main(["--backend", "numpy"] + args)
    capsys.readouterr()
    assert sh["passed"] == sr["passed"], (sh["detail"], sr["detail"])
    assert abs(sh["detail"]["columns_passing"] - sr["detail"]["columns_passing"]) <= 0.02 * sr["detail"]["columns"]
    print(out_hip)


@pytest.mark.gpu
def test_validation_drivers_in_single_precision_at_65536_columns(gpu, capsys):
    """BASELINE configs[2] / [3] sizes in single precision on the reader path: both drivers run to their verdict, the norms
    are finite, the Taylor norms start convergent (|1 - norm| falls over the first step sizes) and the symmetry error stays
    at the float32 rounding level on the tiled stand-in dataset."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test, run_taylor_test

    t = run_taylor_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "2", "--precision", "single"])
    out = capsys.readouterr().out
    norms = np.asarray(t["norms"])
    assert np.all(np.isfinite(norms)) and ("The test passed" in out or "The test failed" in out)
    err = np.abs(1 - norms)
    assert err[1] < err[0] < 0.5 and err.min() < 1e-2, norms
    s = run_symmetry_test.main(["--backend", "hip", "--num-cols", "65536", "--num-runs", "2", "--precision", "single"])
    out = capsys.readouterr().out
    assert ("The symmetry test passed. HOORAY!" in out) == bool(s["passed"])
    assert np.isfinite(s["detail"]["max_error_eps"]) and s["detail"]["columns"] == 65536
    assert s["detail"]["columns_passing"] / s["detail"]["columns"] > 0.95, s["detail"]


@pytest.mark.gpu
def test_timing_brackets_do_not_drain_the_pipeline(gpu):
    """`timing()` records a HIP event pair instead of synchronising on both sides (the reference opens 22 brackets per
    Taylor run): the interval it reports for asynchronous GPU work is the device time of that work, a bracket around host-side
    work reports host time, nested / repeated labels add up, and nothing synchronises until the time is asked for."""
    import time

    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.timing import Timer, timing

    x = torch.zeros(1 << 26, dtype=torch.float64, device=gpu)          # 512 MB: a fill takes ~0.1-0.2 ms
    Timer.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        with timing("fill"):
            for _ in range(5):
                x.add_(1.0)
    enqueue_s = time.perf_counter() - t0
    pending = len(Timer._pending)
    ms = Timer.get_time("fill")                                          # resolves: ONE synchronisation
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        x.add_(1.0)
    b.record()
    torch.cuda.synchronize()
    want = a.elapsed_time(b)
    assert pending == 20 and not Timer._pending
    assert 0.7 * want < ms < 1.5 * want, (ms, want)                      # the device time of the 100 fills
    assert enqueue_s * 1e3 < ms                                          # the host ran ahead: no drain per bracket
    with timing("host"):
        time.sleep(0.05)
    assert 45.0 < Timer.get_time("host") < 80.0                          # host-side work: the host clock
    Timer.reset()
    assert Timer.get_time("fill") == 0.0
