"""GPU parity of cloudsc2_tl / cloudsc2_ad / state_increment / perturbed_state against the oracle,
plus the reference's own acceptance tests run on the HIP kernels: the TL Taylor test
(tangent_linear/validation.py:150-217) and the AD symmetry test (adjoint/validation.py:132-165)."""
import numpy as np
import pytest

from helpers import (NL_IN, NL_OUT, assert_close, externals, from_device, increments, nl_case, nlev_of,
                     run_oracle_ad, run_oracle_nl, run_oracle_tl, symmetry_norm3, taylor_norms, taylor_verdict,
                     to_device)

pytestmark = pytest.mark.gpu
INC = ("aph", "ap", "q", "qsat", "t", "ql", "qi", "lude", "lu", "mfu", "mfd",
       "tnd_cml_t", "tnd_cml_q", "tnd_cml_ql", "tnd_cml_qi", "supsat")


def _nan_outputs(names, nx, nz, dtype, device):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage

    outs = {n: storage.zeros(nx, nz, dtype, device) for n in names}
    for o in outs.values():
        o.fill_(float("nan"))
    return outs


def run_hip_nl(dev_fields, eta_d, dt, ext, nx, nz, dtype, device):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    outs = _nan_outputs(["out_" + n for n in NL_OUT], nx, nz, dtype, device)
    compile_stencil("cloudsc2_nl", ext)(**dev_fields, **outs, in_eta=eta_d, dt=dt, origin=(0, 0, 0),
                                         domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    return outs


def run_hip_tl(fields, fields_i, eta, dt, ext, device, nx, nz):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    dtype = fields["in_ap"].dtype
    dev = to_device({**fields, **fields_i}, device)
    outs = _nan_outputs(["out_" + n for n in NL_OUT] + ["out_" + n + "_i" for n in NL_OUT], nx, nz, dtype, device)
    compile_stencil("cloudsc2_tl", ext)(
        **dev, **outs, in_eta=torch.as_tensor(eta, device=device), tmp_klevel=None, tmp_aph_s=None,
        dt=dtype.type(dt), origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    torch.cuda.synchronize()
    return ({n: from_device(outs["out_" + n]) for n in NL_OUT},
            {n: from_device(outs["out_" + n + "_i"]) for n in NL_OUT})


def run_hip_ad(fields, forcing, eta, dt, ext, device, nx, nz):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    dtype = fields["in_ap"].dtype
    dev = to_device(fields, device)
    frc = to_device({"in_" + n + "_i": forcing[n] for n in NL_OUT}, device)
    frc_before = {k: v.clone() for k, v in frc.items()}
    outs = _nan_outputs(["out_" + n for n in NL_OUT] + ["out_" + n + "_i" for n in NL_IN], nx, nz, dtype, device)
    compile_stencil("cloudsc2_ad", ext)(
        **dev, **frc, **outs, in_eta=torch.as_tensor(eta, device=device), tmp_klevel=None,
        dt=dtype.type(dt), origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    torch.cuda.synchronize()
    for k in frc:  # documented choice (Q1): the forcings are left untouched
        assert torch.equal(frc[k], frc_before[k]), k
    return ({n: from_device(outs["out_" + n]) for n in NL_OUT},
            {n: from_device(outs["out_" + n + "_i"]) for n in NL_IN})


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("regcl", [True, False])
@pytest.mark.parametrize("nx", [64, 333])
def test_tl_matches_oracle(gpu, nx, regcl, dtype):
    ext = externals(LREGCL=regcl, NLEV=137)
    fields, eta, dt = nl_case(nx, dtype=dtype)
    fi = increments(fields, 0.01)
    want, want_i = run_oracle_tl(fields, fi, eta, dt, ext)
    got, got_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        k = nlev_of(n, 137)
        assert_close(f"tl out_{n}", got[n][:k], want[n][:k], dtype)
        # perturbation outputs: compare on the scale of the field's perturbation
        assert_close(f"tl out_{n}_i", got_i[n][:k], want_i[n][:k], dtype, rtol_mul=100.0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("flags", [dict(), dict(LREGCL=False), dict(AD_TRAJ_FIX=1)])
def test_ad_matches_oracle(gpu, flags, dtype):
    nx = 200
    ext = externals(NLEV=137, **flags)
    fields, eta, dt = nl_case(nx, dtype=dtype)
    fi = increments(fields, 0.01, ignore_supsat=True)
    _, forcing = run_oracle_tl(fields, fi, eta, dt, ext)
    want, want_i = run_oracle_ad(fields, forcing, eta, dt, ext)
    got, got_i = run_hip_ad(fields, forcing, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        k = nlev_of(n, 137)
        assert_close(f"ad out_{n}", got[n][:k], want[n][:k], dtype)
    for n in NL_IN:
        # out_aph_i and out_lu_i are written on all nz+1 levels (adjoint/_stencils/cloudsc2.py:970-986)
        k = 138 if n in ("aph", "lu") else 137
        assert_close(f"ad out_{n}_i{flags}", got_i[n][:k], want_i[n][:k], dtype, rtol_mul=1000.0)
        if k == 137:
            assert np.isnan(got_i[n][137]).all(), n


def test_taylor_test_on_the_gpu(gpu):
    """BASELINE config 3 protocol at a size the oracle can follow (512 columns): saturation -> NL ->
    state_increment(0.01) -> TL -> 10 x (perturbed_state(f2) -> NL), all through the HIP stencils;
    norms as TaylorTest.get_norm.  LREGCL = False as TaylorTest forces (validation.py:84-85)."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx, nz = 512, 137
    ext = externals(LREGCL=False, NLEV=nz)
    fields, eta, dt = nl_case(nx)
    dev = to_device(fields, gpu)
    eta_d = torch.as_tensor(eta, device=gpu)
    # increments on the device
    st = {"in_" + n: dev["in_" + n] for n in INC}
    inc = {"out_" + n + "_i": storage.zeros(nx, nz, np.float64, gpu) for n in INC}
    compile_stencil("state_increment", {"IGNORE_SUPSAT": False})(
        **st, **inc, f=0.01, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    dev_i = {"in_" + n + "_i": inc["out_" + n + "_i"] for n in INC}
    nl0 = run_hip_nl(dev, eta_d, dt, ext, nx, nz, np.float64, gpu)
    tl = _nan_outputs(["out_" + n for n in NL_OUT] + ["out_" + n + "_i" for n in NL_OUT], nx, nz, np.float64, gpu)
    compile_stencil("cloudsc2_tl", ext)(**dev, **dev_i, **tl, in_eta=eta_d, dt=dt, origin=(0, 0, 0),
                                         domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    pert = compile_stencil("perturbed_state", {})
    sp = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in INC}

    def host(d, suffix=""):
        return {n: from_device(d["out_" + n + suffix]) for n in NL_OUT}

    nl0_h, tl_i_h = host(nl0), host(tl, "_i")
    for n in NL_OUT:  # padding levels hold NaN by construction of the test: mask them
        k = nlev_of(n, nz)
        nl0_h[n], tl_i_h[n] = nl0_h[n][:k], tl_i_h[n][:k]

    def nlp(f2):
        pert(**st, **dev_i, **sp, f=f2, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True,
             exec_info=None)
        p = run_hip_nl({"in_" + n: sp["out_" + n] for n in INC}, eta_d, dt, ext, nx, nz, np.float64, gpu)
        h = host(p)
        return {n: h[n][:nlev_of(n, nz)] for n in NL_OUT}

    f2s = tuple(10.0 ** -i for i in range(1, 11))
    norms = taylor_norms(nl0_h, nlp, tl_i_h, f2s)
    err = np.abs(1 - norms)
    assert err.min() < 1e-6, norms
    assert err[4:9].max() < 1e-4, norms
    # the same experiment with the oracle gives the same norms (the kernels ARE the reference's math)
    fi = increments(fields, 0.01)
    o_nl0 = run_oracle_nl(fields, eta, dt, ext)
    _, o_tl_i = run_oracle_tl(fields, fi, eta, dt, ext)
    o_norms = taylor_norms(o_nl0, lambda f2: run_oracle_nl({k: fields[k] + f2 * fi[k + "_i"] for k in fields},
                                                           eta, dt, ext), o_tl_i, f2s)
    np.testing.assert_allclose(norms[:7], o_norms[:7], rtol=1e-6)
    print("Taylor norms (HIP):", norms, taylor_verdict(norms))


def test_symmetry_test_on_the_gpu(gpu):
    """BASELINE config 4 protocol (512 columns): state_increment(0.01, ignore_supsat) -> TL -> AD on the
    TL outputs; norm3 = |norm1 - norm2| / (eps norm2) per column, pass iff max < 1e4
    (adjoint/validation.py:157-165).  With the reference's literal freezing tests the columns whose
    adjustment crosses RTT fail, exactly as in the oracle; with AD_TRAJ_FIX every column passes."""
    nx = 512
    fields, eta, dt = nl_case(nx)
    fi = increments(fields, 0.01, ignore_supsat=True)
    ext = externals(NLEV=137)
    _, tl_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        tl_i[n][nlev_of(n, 137):] = 0.0
    _, ad_i = run_hip_ad(fields, tl_i, eta, dt, ext, gpu, nx, 137)
    for n in NL_IN:
        ad_i[n][(138 if n in ("aph", "lu") else 137):] = 0.0
    _, _, norm3 = symmetry_norm3(tl_i, fi, ad_i)
    # oracle verdict per column must be the same as the kernels' verdict
    _, o_tl_i = run_oracle_tl(fields, fi, eta, dt, ext)
    _, o_ad_i = run_oracle_ad(fields, o_tl_i, eta, dt, ext)
    _, _, o_norm3 = symmetry_norm3(o_tl_i, fi, o_ad_i)
    assert np.array_equal(norm3 < 1e4, o_norm3 < 1e4)
    assert np.mean(norm3 < 1e4) > 0.95
    _, ad_i = run_hip_ad(fields, tl_i, eta, dt, externals(NLEV=137, AD_TRAJ_FIX=1), gpu, nx, 137)
    for n in NL_IN:
        ad_i[n][(138 if n in ("aph", "lu") else 137):] = 0.0
    _, _, norm3 = symmetry_norm3(tl_i, fi, ad_i)
    assert norm3.max() < 1e4, norm3.max()
    print(f"symmetry test (HIP, AD_TRAJ_FIX): max error {norm3.max():.3e} x eps")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_increment_and_perturbed_state(gpu, dtype):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx, nz = 130, 137
    fields, _, _ = nl_case(nx, dtype=dtype)
    dev = to_device(fields, gpu)
    st = {"in_" + n: dev["in_" + n] for n in INC}
    for ignore in (False, True):
        inc = {"out_" + n + "_i": storage.zeros(nx, nz, dtype, gpu) for n in INC}
        compile_stencil("state_increment", {"IGNORE_SUPSAT": ignore})(
            **st, **inc, f=dtype(0.01), origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True,
            exec_info=None)
        for n in INC:
            want = (dtype(0.01) * fields["in_" + n]).astype(dtype)
            if n == "supsat" and ignore:
                want = np.zeros_like(want)
            assert np.array_equal(from_device(inc["out_" + n + "_i"]), want), n  # one multiply: bit-exact
    out = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in INC}
    compile_stencil("perturbed_state", {})(
        **st, **{"in_" + n + "_i": inc["out_" + n + "_i"] for n in INC}, **out, f=dtype(1e-3),
        origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    torch.cuda.synchronize()
    for n in INC:
        xi = from_device(inc["out_" + n + "_i"])
        want = fields["in_" + n] + dtype(1e-3) * xi
        got = from_device(out["out_" + n])
        # a + f*b is contracted to one fma on the GPU: agree to 1 ulp
        np.testing.assert_allclose(got, want, rtol=4 * np.finfo(dtype).eps, atol=0)


def _assert_close_by_column(name, got, want, tol):
    """|got - want| <= tol * max_k |want[k, col]|: every column is judged on its own scale (the evaporation block's
    perturbations span 40 orders of magnitude between columns)."""
    assert not np.isnan(got).any(), f"{name}: NaN"
    scale = np.abs(want).max(axis=0, keepdims=True)
    err = np.abs(got - want)
    bad = err > tol * scale + np.finfo(np.float64).tiny
    assert not bad.any(), (f"{name}: {int(bad.sum())} points, worst {np.nanmax(err / (scale + 1e-300)):.2e} "
                           f"of the column scale (tol {tol:.0e})")


@pytest.mark.parametrize("dt", [1.0, 60.0])
@pytest.mark.parametrize("sw", [dict(LEVAPLS2=True, LREGCL=False), dict(LDRAIN1D=True, LREGCL=True)])
def test_tl_evaporation_block_matches_oracle(gpu, sw, dt):
    """LEVAPLS2 / LDRAIN1D: the precipitation-evaporation block of cloudsc2_tl (tangent_linear/_stencils/
    cloudsc2.py:528-616).  Increments are NOT proportional to the state (a uniform 1 % scaling is nearly a symmetry of
    the scheme: several perturbations then cancel to rounding noise) and dt is 1 s / 60 s: the reference's b_i carries a
    dt**2 where the derivative has dt (:565-569), so at dt = 3600 s the recurrence amplifies rounding noise by ~3600 per
    level and no two implementations agree on the perturbations (see the next test)."""
    nx = 333
    ext = externals(NLEV=137, **sw)
    fields, eta, _ = nl_case(nx, ext=ext)
    rng = np.random.default_rng(5)
    fi = {k: v * rng.uniform(0.5, 1.5, size=v.shape) for k, v in increments(fields, 0.01).items()}
    want, want_i = run_oracle_tl(fields, fi, eta, dt, ext)
    got, got_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, 137)
    assert np.abs(want["covptot"]).max() > 0 and np.abs(want_i["covptot"]).max() > 0   # the block is exercised
    for n in NL_OUT:
        k = nlev_of(n, 137)
        _assert_close_by_column(f"tl-evap out_{n}", got[n][:k], want[n][:k], 1e-9)
        _assert_close_by_column(f"tl-evap out_{n}_i", got_i[n][:k], want_i[n][:k], 1e-9)


def test_tl_evaporation_block_at_the_driver_timestep(gpu):
    """dt = 3600 s (the drivers' timestep): the trajectory half of cloudsc2_tl must still match, and it must equal the
    NL kernel's result for the same switches; the perturbations only have to be finite and of the oracle's magnitude
    (they reach 1e57: the dt**2 quirk above)."""
    nx = 333
    ext = externals(NLEV=137, LEVAPLS2=True, LREGCL=False)
    fields, eta, dt = nl_case(nx, ext=ext)
    fi = increments(fields, 0.01)
    want, want_i = run_oracle_tl(fields, fi, eta, dt, ext)
    got, got_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        k = nlev_of(n, 137)
        _assert_close_by_column(f"tl-evap traj out_{n}", got[n][:k], want[n][:k], 1e-9)
        assert np.isfinite(got_i[n][:k]).all()
        assert np.abs(got_i[n][:k] - want_i[n][:k]).max() <= 1e-9 * np.abs(want_i[n][:k]).max()


@pytest.mark.parametrize("sw", [dict(LEVAPLS2=True, LREGCL=False), dict(LDRAIN1D=True, LREGCL=True),
                                dict(LEVAPLS2=True, LREGCL=True, AD_TRAJ_FIX=1)])
def test_ad_evaporation_block_matches_oracle(gpu, sw):
    """cloudsc2_ad with the evaporation block (adjoint/_stencils/cloudsc2.py:357-394, :635-719) at dt = 60 s, forced
    with the oracle's TL output perturbations of non-proportional increments.  (The reference's AD evaporation block is
    NOT the transpose of its TL one - the symmetry norm is O(1e16) eps already in the oracle - so there is no symmetry
    test for this switch, only parity.)"""
    nx, dt = 333, 60.0
    ext = externals(NLEV=137, **sw)
    fields, eta, _ = nl_case(nx, ext=ext)
    rng = np.random.default_rng(5)
    fi = {k: v * rng.uniform(0.5, 1.5, size=v.shape) for k, v in increments(fields, 0.01, ignore_supsat=True).items()}
    _, forcing = run_oracle_tl(fields, fi, eta, dt, ext)
    want, want_i = run_oracle_ad(fields, forcing, eta, dt, ext)
    got, got_i = run_hip_ad(fields, forcing, eta, dt, ext, gpu, nx, 137)
    assert np.abs(want["covptot"]).max() > 0
    for n in NL_OUT:
        k = nlev_of(n, 137)
        _assert_close_by_column(f"ad-evap out_{n}", got[n][:k], want[n][:k], 1e-9)
    for n in NL_IN:
        k = 138 if n in ("aph", "lu") else 137
        # out_lu_i inherits the cancellation inside a_clc (tests/test_reference_exec.py)
        _assert_close_by_column(f"ad-evap out_{n}_i", got_i[n][:k], want_i[n][:k], 1e-4 if n == "lu" else 1e-8)


@pytest.mark.parametrize("kind", ["tl", "ad"])
def test_evaporation_block_fp32_instantiations(gpu, kind):
    """fp32 builds of the TL / AD evaporation block: the trajectory half must match the fp32 oracle within the fp32
    tolerance of tests/helpers.py and every perturbation / adjoint output must be finite (their values are not compared:
    in single precision the block's cancellations leave few significant digits even at dt = 60 s)."""
    nx, dt = 256, 60.0
    ext = externals(NLEV=137, LEVAPLS2=True, LREGCL=True)
    fields, eta, _ = nl_case(nx, dtype=np.float32, ext=ext)
    rng = np.random.default_rng(9)
    fi = {k: (v * rng.uniform(0.5, 1.5, size=v.shape)).astype(np.float32)
          for k, v in increments(fields, 0.01, ignore_supsat=True).items()}
    want, want_i = run_oracle_tl(fields, fi, eta, dt, ext)
    if kind == "tl":
        got, got_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, 137)
        outs_i = {n: got_i[n][:nlev_of(n, 137)] for n in NL_OUT}
    else:
        forcing = {n: np.nan_to_num(want_i[n], posinf=0.0, neginf=0.0).astype(np.float32) for n in NL_OUT}
        want, _ = run_oracle_ad(fields, forcing, eta, dt, ext)   # the AD stencil's own trajectory (quirks Q4 / Q10)
        got, got_i = run_hip_ad(fields, forcing, eta, dt, ext, gpu, nx, 137)
        outs_i = {n: got_i[n][:(138 if n in ("aph", "lu") else 137)] for n in NL_IN}
    assert np.abs(want["covptot"]).max() > 0
    for n in NL_OUT:
        k = nlev_of(n, 137)
        assert_close(f"{kind}-evap fp32 out_{n}", got[n][:k], want[n][:k], np.float32)
    for n, v in outs_i.items():
        assert np.isfinite(v).all(), n


@pytest.mark.parametrize("regcl", [True, False])
def test_tl_with_general_increments(gpu, regcl):
    """Increments proportional to the state (what `state_increment` produces, and what most tests above use) make many TL
    terms cancel - a uniform 1 % scaling is nearly a symmetry of the scheme.  Here every input field gets its own random
    increment (sign included) at the driver's dt = 3600 s, default switches, each column judged on its own scale."""
    nx = 512
    ext = externals(NLEV=137, LREGCL=regcl)
    fields, eta, dt = nl_case(nx, ext=ext, seed=31)
    rng = np.random.default_rng(23)
    fi = {}
    for k, v in fields.items():
        fi[k + "_i"] = v * rng.uniform(-0.02, 0.02, size=v.shape)
    fi["in_t_i"] = rng.normal(0.0, 0.3, size=fields["in_t"].shape) * (fields["in_t"] != 0)
    want, want_i = run_oracle_tl(fields, fi, eta, dt, ext)
    got, got_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        k = nlev_of(n, 137)
        _assert_close_by_column(f"tl general out_{n}", got[n][:k], want[n][:k], 1e-9)
        _assert_close_by_column(f"tl general out_{n}_i", got_i[n][:k], want_i[n][:k], 1e-8)


@pytest.mark.parametrize("flags", [dict(), dict(LREGCL=False), dict(AD_TRAJ_FIX=1)])
def test_ad_with_general_forcing(gpu, flags):
    """Adjoint forcings that come out of a TL run with proportional increments leave whole paths of cloudsc2_ad nearly
    unexercised (in_clc_i is rounding noise there, in_covptot_i is zero).  Here all ten forcing fields are independent
    random fields of the size of the corresponding NL output; every adjoint output is judged per column."""
    nx = 512
    ext = externals(NLEV=137, **flags)
    fields, eta, dt = nl_case(nx, ext=ext, seed=37)
    nl = run_oracle_nl(fields, eta, dt, ext)
    rng = np.random.default_rng(29)
    forcing = {}
    for n in NL_OUT:
        scale = max(float(np.abs(nl[n]).max()), 1e-30) if n != "covptot" else 1.0
        forcing[n] = rng.normal(0.0, 1.0, size=nl[n].shape) * scale
        forcing[n][nlev_of(n, 137):] = 0.0
    want, want_i = run_oracle_ad(fields, forcing, eta, dt, ext)
    got, got_i = run_hip_ad(fields, forcing, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        k = nlev_of(n, 137)
        _assert_close_by_column(f"ad general out_{n}", got[n][:k], want[n][:k], 1e-9)
    for n in NL_IN:
        k = 138 if n in ("aph", "lu") else 137
        _assert_close_by_column(f"ad general out_{n}_i{flags}", got_i[n][:k], want_i[n][:k], 1e-7)


def test_symmetry_with_general_increments(gpu):
    """<TL dx, TL dx> == <dx, AD TL dx> for an arbitrary dx (every field its own random increment, signs included;
    supsat excluded as in the reference's test because of quirk Q7), not only for the proportional increments of the
    reference's symmetry test: with AD_TRAJ_FIX the adjoint is the transpose of TL in every column."""
    nx = 512
    ext = externals(NLEV=137, AD_TRAJ_FIX=1)
    fields, eta, dt = nl_case(nx, seed=41)
    rng = np.random.default_rng(43)
    fi = {k + "_i": v * rng.uniform(-0.02, 0.02, size=v.shape) for k, v in fields.items()}
    fi["in_t_i"] = rng.normal(0.0, 0.3, size=fields["in_t"].shape) * (fields["in_t"] != 0)
    fi["in_supsat_i"] = np.zeros_like(fields["in_supsat"])
    _, tl_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        tl_i[n][nlev_of(n, 137):] = 0.0
    _, ad_i = run_hip_ad(fields, tl_i, eta, dt, ext, gpu, nx, 137)
    for n in NL_IN:
        ad_i[n][(138 if n in ("aph", "lu") else 137):] = 0.0
    norm1, norm2, norm3 = symmetry_norm3(tl_i, fi, ad_i)
    assert (norm1 > 0).all()
    assert norm3.max() < 1e4, (norm3.max(), int(np.argmax(norm3)))
    print(f"symmetry with general increments: max error {norm3.max():.3e} x eps")


def test_taylor_with_general_increments(gpu):
    """TL against finite differences of NL for an arbitrary dx (the reference's Taylor protocol and scoring, but every
    field with its own random increment), the perturbed NL runs through the fused `cloudsc2_nl_perturbed` kernel."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx, nz = 2048, 137
    ext = externals(NLEV=nz, LREGCL=False)                # no regularisation in the Taylor test (validation.py:84-85)
    fields, eta, dt = nl_case(nx, seed=47)
    rng = np.random.default_rng(53)
    fi = {k + "_i": v * rng.uniform(-0.02, 0.02, size=v.shape) for k, v in fields.items()}
    fi["in_t_i"] = rng.normal(0.0, 0.3, size=fields["in_t"].shape) * (fields["in_t"] != 0)
    nl0, tl_i = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, nz)
    dev = to_device({**fields, **fi}, gpu)
    eta_d = torch.as_tensor(eta, device=gpu)
    nlp = compile_stencil("cloudsc2_nl_perturbed", ext)
    outs = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}

    def perturbed(f2):
        nlp(**dev, **outs, in_eta=eta_d, f=f2, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True,
            exec_info=None)
        torch.cuda.synchronize()
        return {n: storage.klayout(outs["out_" + n]).cpu().numpy() for n in NL_OUT}

    for n in NL_OUT:            # rows a stencil does not write are not part of the sums
        nl0[n][nlev_of(n, nz):] = 0.0
        tl_i[n][nlev_of(n, nz):] = 0.0
    f2s = [10.0 ** -(i + 1) for i in range(10)]
    norms = taylor_norms(nl0, perturbed, tl_i, f2s)
    ok, msg = taylor_verdict(norms)
    print("taylor norms (general increments):", " ".join(f"{x:.8f}" for x in norms), msg)
    assert ok, (norms, msg)


def test_tl_ad_on_strided_column_windows(gpu):
    """lev_stride > nx: TL and AD on a column window of wider storages (a rank's shard addressed in place) give the
    bits of the same columns copied into contiguous storages; columns outside the window are not touched."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx, nz, W, c0 = 190, 137, 300, 37      # not a multiple of 64: both calls take the register-prefetch kernels (bit-equal)
    ext = externals(NLEV=nz)
    fields, eta, dt = nl_case(nx, seed=59)
    fi = increments(fields, 0.01, ignore_supsat=True)
    eta_d = torch.as_tensor(eta, device=gpu)
    tdt = storage.torch_dtype(np.float64)

    def wide(v=None):
        t = torch.full((nz + 1, W), 7.0, dtype=tdt, device=gpu)
        if v is not None:
            t[:, c0:c0 + nx] = torch.as_tensor(v, device=gpu)
        return t

    def win(t):
        return storage.logical_view(t[:, c0:c0 + nx])

    com = dict(in_eta=eta_d, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    # contiguous reference calls
    dev = to_device({**fields, **fi}, gpu)
    tl_c = {"out_" + n + s: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT for s in ("", "_i")}
    compile_stencil("cloudsc2_tl", ext)(**dev, **tl_c, **com)
    frc = {"in_" + n + "_i": tl_c["out_" + n + "_i"] for n in NL_OUT}
    ad_c = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    ad_c.update({"out_" + n + "_i": storage.zeros(nx, nz, np.float64, gpu) for n in NL_IN})
    compile_stencil("cloudsc2_ad", ext)(**{k: v for k, v in dev.items() if not k.endswith("_i")}, **frc, **ad_c, **com)
    # the same through windows of wide storages
    wi = {k: wide(v) for k, v in {**fields, **fi}.items()}
    tl_w = {k: wide() for k in tl_c}
    compile_stencil("cloudsc2_tl", ext)(**{k: win(v) for k, v in wi.items()}, **{k: win(v) for k, v in tl_w.items()}, **com)
    frc_w = {"in_" + n + "_i": win(tl_w["out_" + n + "_i"]) for n in NL_OUT}
    ad_w = {k: wide() for k in ad_c}
    compile_stencil("cloudsc2_ad", ext)(**{k: win(v) for k, v in wi.items() if not k.endswith("_i")}, **frc_w,
                                         **{k: win(v) for k, v in ad_w.items()}, **com)
    torch.cuda.synchronize()
    for name, c, w in [(k, tl_c[k], tl_w[k]) for k in tl_c] + [(k, ad_c[k], ad_w[k]) for k in ad_c]:
        half = name.replace("out_", "").replace("_i", "") in ("fhpsl", "fhpsn", "fplsl", "fplsn", "aph", "lu")
        rows = nz + 1 if half and not name.startswith("out_tnd") else nz
        assert torch.equal(storage.klayout(c)[:rows], w[:rows, c0:c0 + nx]), name
        assert bool((w[:, :c0] == 7.0).all()) and bool((w[:, c0 + nx:] == 7.0).all()), name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nz,sw", [(137, dict(LEVAPLS2=True)), (137, dict(LREGCL=False)), (5, {}), (3, {}), (2, {})])
def test_tl_lds_ring_and_register_prefetch_paths_agree(gpu, dtype, nz, sw):
    """cloudsc2_tl has two load paths (csrc/cloudsc2_tl.hip): the LDS-DMA ring (whole waves, 16-byte aligned rows) and
    the register prefetch (everything else).  The same 320 columns presented (a) as aligned contiguous storages and (b)
    as a window that starts at column 1 of wider storages (misaligned rows -> register path) must agree 100x tighter
    than the HIP-vs-oracle tolerance (same level function, but fma contraction may differ between the two contexts),
    and both must match the oracle.  nz = 2, 3 exercise the head / tail of the ring (two slots per wave)."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx = 320
    ext = externals(NLEV=nz, **sw)
    fields, eta, dt = nl_case(nx, nz=nz, dtype=dtype, seed=5)
    if sw.get("LEVAPLS2"):
        dt = 60.0        # the evaporation block amplifies rounding noise at 3600 s (docs/DESIGN_r03_detail.md 3.3)
    fi = increments(fields, 0.01)
    want, want_i = run_oracle_tl(fields, fi, eta, dt, ext)
    tl = compile_stencil("cloudsc2_tl", ext)
    eta_d = torch.as_tensor(eta, device=gpu)
    com = dict(in_eta=eta_d, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    names = ["out_" + n + s for n in NL_OUT for s in ("", "_i")]
    aligned = to_device({**fields, **fi}, gpu)
    out_a = _nan_outputs(names, nx, nz, dtype, gpu)
    tl(**aligned, **out_a, **com)
    assert _lib.last_kernel() == "cs2::tl_ring_kernel"
    tdt = storage.torch_dtype(dtype)
    wide = {k: torch.zeros((nz + 1, nx + 3), dtype=tdt, device=gpu) for k in aligned}
    for k, v in {**fields, **fi}.items():
        wide[k][:, 1:nx + 1] = torch.as_tensor(v, device=gpu)
    out_w = {k: torch.zeros((nz + 1, nx + 3), dtype=tdt, device=gpu) for k in names}
    tl(**{k: storage.logical_view(v[:, 1:nx + 1]) for k, v in wide.items()},
       **{k: storage.logical_view(v[:, 1:nx + 1]) for k, v in out_w.items()}, **com)
    assert _lib.last_kernel() == "cs2::tl_kernel"
    torch.cuda.synchronize()
    for n in NL_OUT:
        k = nlev_of(n, nz)
        for sfx, ref, mul in (("", want, 1.0), ("_i", want_i, 100.0)):
            a = from_device(out_a["out_" + n + sfx])
            w = out_w["out_" + n + sfx][:, 1:nx + 1].cpu().numpy()
            assert_close(f"ring vs register out_{n}{sfx}[nz={nz}]", a[:k], w[:k], dtype, rtol_mul=1e-2 * mul)
            assert_close(f"ring out_{n}{sfx}[nz={nz}]", a[:k], ref[n][:k], dtype, rtol_mul=mul)
            assert (out_w["out_" + n + sfx][:, 0] == 0).all() and (out_w["out_" + n + sfx][:, nx + 1:] == 0).all()
