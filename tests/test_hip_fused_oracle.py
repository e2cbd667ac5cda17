"""Build extensions held DIRECTLY against the oracle ("fused entry points are compared with other HIP kernels
only"), plus the round-3 extensions: the validation-norm reduction kernels, the all-step-sizes Taylor kernel
(`cloudsc2_nl_taylor_multi`) and the HIP-graph / fused-all modes of the harnesses.

Oracle side (CPU, NumPy restatement): saturation + cloudsc2_nl on x, on x + f x_i (perturbed_state restated), and the
sums the reference's TaylorTest forms from them (tangent_linear/validation.py:239-261).  Tolerances: tests/helpers.py."""
import numpy as np
import pytest

from helpers import NL_OUT, assert_close, externals, nl_case, nlev_of, oracle, run_oracle_nl, to_device

pytestmark = pytest.mark.gpu


def _oracle_perturbed(fields, fields_i, f2, eta, dt, ext):
    """perturbed_state (common/_stencils/perturbed_state.py:75-91) then cloudsc2_nl, both by the oracle"""
    dtype = fields["in_ap"].dtype.type
    st = {k[3:]: v for k, v in fields.items()}
    st_i = {k[3:]: fields_i[k + "_i"] for k in fields}
    merged = dict(st)
    merged.update({k + "_i": v for k, v in st_i.items()})
    out = {k: np.zeros_like(v) for k, v in st.items()}
    oracle.perturbed_state(merged, out, dtype(f2))
    return run_oracle_nl({"in_" + k: v for k, v in out.items()}, eta, dt, ext)


def _general_increments(fields, seed=5):
    rng = np.random.default_rng(seed)
    return {k + "_i": (v * rng.uniform(-0.02, 0.02, size=v.shape)).astype(v.dtype) for k, v in fields.items()}


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nx", [333, 1024])     # register-prefetch path / LDS-ring path
def test_fused_saturation_variant_matches_the_oracle(gpu, dtype, nx):
    """`cloudsc2_nl_saturation` against oracle.saturation + oracle.cloudsc2_nl (not against the unfused HIP kernels)"""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nz, ext = 137, externals()
    fields, eta, dt = nl_case(nx, dtype=dtype, seed=11)      # in_qsat = oracle.saturation(ap, t)
    want = run_oracle_nl(fields, eta, dt, ext)
    dev = to_device({k: v for k, v in fields.items() if k != "in_qsat"}, gpu)
    qsat = storage.zeros(nx, nz, dtype, gpu)
    outs = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl_saturation", ext)(**dev, out_qsat=qsat, **outs, in_eta=torch.as_tensor(eta, device=gpu),
                                                    dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True,
                                                    exec_info=None)
    torch.cuda.synchronize()
    assert_close("fused qsat", storage.klayout(qsat).cpu().numpy()[:nz], fields["in_qsat"][:nz], dtype)
    for n in NL_OUT:
        k = nlev_of(n, nz)
        assert_close(f"fused-saturation out_{n}", storage.klayout(outs["out_" + n]).cpu().numpy()[:k], want[n][:k], dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("sw", [dict(), dict(LEVAPLS2=True)])
def test_fused_perturbation_variant_matches_the_oracle(gpu, dtype, sw):
    """`cloudsc2_nl_perturbed` with GENERAL increments against oracle.perturbed_state + oracle.cloudsc2_nl"""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx, nz, f2 = 500, 137, 0.1
    ext = externals(**sw)
    fields, eta, dt = nl_case(nx, dtype=dtype, seed=12, ext=ext)
    inc = _general_increments(fields)
    want = _oracle_perturbed(fields, inc, f2, eta, dt, ext)
    outs = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl_perturbed", ext)(**to_device(fields, gpu), **to_device(inc, gpu), **outs, f=f2,
                                                   in_eta=torch.as_tensor(eta, device=gpu), dt=dt, origin=(0, 0, 0),
                                                   domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    torch.cuda.synchronize()
    for n in NL_OUT:
        k = nlev_of(n, nz)
        # fp32 with the evaporation block: the parity tests' own allowance for that block (tests/test_hip_nl.py)
        mul = 4.0 if (sw and dtype == np.float32) else 1.0
        assert_close(f"fused-perturbed out_{n}", storage.klayout(outs["out_" + n]).cpu().numpy()[:k], want[n][:k], dtype,
                     rtol_mul=mul)


def _oracle_taylor_sums(fields, inc, f2s, eta, dt, ext):
    """sum(NL(x + f x_i) - NL(x)) per output field and step size, the differences formed in the field type
    (np.sum(field_nl_p - field_nl), tangent_linear/validation.py:255), accumulated in double"""
    base = run_oracle_nl(fields, eta, dt, ext)
    rows, mags = [], []
    for f2 in f2s:
        p = _oracle_perturbed(fields, inc, f2, eta, dt, ext)
        rows.append([float((p[n] - base[n]).sum(dtype=np.float64)) for n in NL_OUT])
        mags.append([float(np.abs(p[n] - base[n]).sum(dtype=np.float64)) for n in NL_OUT])
    return base, np.array(rows), np.array(mags)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("sw", [dict(), dict(LEVAPLS2=True)])
def test_taylor_kernels_match_the_oracle_and_each_other(gpu, dtype, sw):
    """`cloudsc2_nl_taylor` (one step size per launch) and `cloudsc2_nl_taylor_multi` (all step sizes, 5 per launch; 7
    step sizes = launches of 5 and 2) against the ORACLE's sums, and against each other."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil, taylor_blocks

    nx, nz = 700, 137
    f2s = (0.5, 0.1, 1e-2, 1e-3, 1e-4, 1e-6, 1e-9)
    ext = externals(**sw)
    fields, eta, dt = nl_case(nx, dtype=dtype, seed=13, ext=ext)
    inc = _general_increments(fields, seed=6)
    base, want, mag = _oracle_taylor_sums(fields, inc, f2s, eta, dt, ext)
    dev, dev_i = to_device(fields, gpu), to_device(inc, gpu)
    com = dict(in_eta=torch.as_tensor(eta, device=gpu), dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1),
               validate_args=True, exec_info=None)
    ref = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl", ext)(**dev, **ref, **com)
    refs = {"ref_" + n: ref["out_" + n] for n in NL_OUT}
    keep = {k: v.clone() for k, v in ref.items()}
    nb = taylor_blocks(nx)
    single = torch.full((len(f2s), nb, len(NL_OUT)), float("nan"), dtype=torch.float64, device=gpu)
    one = compile_stencil("cloudsc2_nl_taylor", ext)
    for j, f2 in enumerate(f2s):
        one(**dev, **dev_i, **refs, out_partials=single[j], f=f2, **com)
    multi = torch.full((nb, len(f2s), len(NL_OUT)), float("nan"), dtype=torch.float64, device=gpu)
    compile_stencil("cloudsc2_nl_taylor_multi", ext)(**dev, **dev_i, **refs, out_partials=multi, fs=f2s, **com)
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(ref[k], keep[k]), k                       # the reference outputs are read-only
    got1 = single.sum(dim=1).cpu().numpy()
    gotm = multi.sum(dim=0).cpu().numpy()
    assert not np.isnan(gotm).any() and not np.isnan(got1).any()
    # the two kernels run the same level function on the same words and reduce in the same order
    assert np.allclose(gotm, got1, rtol=0, atol=1e-13 * mag.max()) if dtype == np.float64 else np.allclose(gotm, got1, rtol=1e-6, atol=1e-7 * mag.max())
    # against the oracle: every difference carries the kernels' pointwise tolerance (helpers.TOL) on BOTH runs
    tol = dict(rtol=1e-9, atol_rel=1e-11) if dtype == np.float64 else dict(rtol=2e-3, atol_rel=2e-4)
    for fi, n in enumerate(NL_OUT):
        scale = float(np.abs(base[n]).sum(dtype=np.float64))        # sum of |field|: what the pointwise bound adds up to
        bound = 2.0 * (tol["rtol"] + tol["atol_rel"]) * scale + 1e-300
        assert np.all(np.abs(gotm[:, fi] - want[:, fi]) <= bound), (n, gotm[:, fi], want[:, fi], bound)
        assert np.all(np.abs(got1[:, fi] - want[:, fi]) <= bound), (n, got1[:, fi], want[:, fi], bound)
    assert np.abs(want).max() > 0
    # where the perturbation is far above rounding the sums agree to many digits (fp64)
    if dtype == np.float64:
        big = np.abs(want) > 1e-6 * mag.max()
        assert np.all(np.abs(gotm[big] - want[big]) <= 1e-6 * np.abs(want[big]))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ignore_supsat", [False, True])
def test_tl_with_fused_increment_equals_separate_calls_and_the_oracle(gpu, dtype, ignore_supsat):
    """`cloudsc2_tl_incremented` (state_increment inside the TL kernel) against state_increment then cloudsc2_tl - 100x
    (fp32: 10x) tighter than the HIP-vs-oracle tolerance (the increments are the stored products exactly, `rounded_product`; the level
    function is the same source, but hipcc's fma contraction depends on the kernel it is inlined into, as for the ring and
    register paths of cloudsc2_nl) - and against the oracle's state_increment + cloudsc2_tl within the TL tolerance; the
    multi-step Taylor kernel with `f_inc` against the same kernel fed with stored increments."""
    import torch

    from helpers import increments, run_oracle_tl
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, compile_stencil, taylor_blocks

    nx, nz, f1 = 600, 137, 0.01
    ext = externals(NLEV=nz, IGNORE_SUPSAT=ignore_supsat, LREGCL=not ignore_supsat)
    fields, eta, dt = nl_case(nx, dtype=dtype, seed=21)
    want_nl, want_tl = run_oracle_tl(fields, increments(fields, f1, ignore_supsat), eta, dt, ext)
    dev = to_device(fields, gpu)
    com = dict(origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    eta_d = torch.as_tensor(eta, device=gpu)
    Z = lambda: storage.zeros(nx, nz, dtype, gpu)  # noqa: E731
    inc = {"out_" + n + "_i": Z() for n in INC}
    compile_stencil("state_increment", ext)(**{"in_" + n: dev["in_" + n] for n in INC}, **inc, f=f1, **com)
    dev_i = {"in_" + n + "_i": inc["out_" + n + "_i"] for n in INC}
    sep = {**{"out_" + n: Z() for n in NL_OUT}, **{"out_" + n + "_i": Z() for n in NL_OUT}}
    compile_stencil("cloudsc2_tl", ext)(**dev, **dev_i, **sep, in_eta=eta_d, dt=dt, **com)
    fus = {**{"out_" + n: Z() for n in NL_OUT}, **{"out_" + n + "_i": Z() for n in NL_OUT}}
    compile_stencil("cloudsc2_tl_incremented", ext)(**dev, **fus, in_eta=eta_d, dt=dt, f=f1, **com)
    torch.cuda.synchronize()
    for n in NL_OUT:
        k = nlev_of(n, nz)
        # trajectory: 100x (fp64) / 10x (fp32: one point of 82 200 sits at 1e-2, clc = 1 - sqrt(..) amplifies an ulp) tighter
        # than the HIP-vs-oracle tolerance; perturbation fields are differences of nearly equal numbers
        for sfx_, mul in (("", 1e-2 if dtype == np.float64 else 1e-1), ("_i", 1.0)):
            a_ = storage.klayout(fus["out_" + n + sfx_]).cpu().numpy()[:k]
            b_ = storage.klayout(sep["out_" + n + sfx_]).cpu().numpy()[:k]
            assert_close(f"tl-incremented vs separate out_{n}{sfx_}", a_, b_, dtype, rtol_mul=mul)
    for n in NL_OUT:
        k = nlev_of(n, nz)
        assert_close(f"tl-incremented out_{n}", storage.klayout(fus["out_" + n]).cpu().numpy()[:k], want_nl[n][:k], dtype)
        scale = float(np.abs(want_tl[n][:k]).max())
        assert_close(f"tl-incremented out_{n}_i", storage.klayout(fus["out_" + n + "_i"]).cpu().numpy()[:k], want_tl[n][:k],
                     dtype, scale=scale, rtol_mul=100.0)      # perturbation fields: the TL tests' allowance (helpers / DESIGN 4)
    f2s = (1e-1, 1e-3, 1e-5)
    refs = {"ref_" + n: sep["out_" + n] for n in NL_OUT}
    multi = compile_stencil("cloudsc2_nl_taylor_multi", ext)
    pa = torch.zeros((taylor_blocks(nx), len(f2s), len(NL_OUT)), dtype=torch.float64, device=gpu)
    pb = torch.zeros_like(pa)
    multi(**dev, **dev_i, **refs, out_partials=pa, fs=f2s, in_eta=eta_d, dt=dt, **com)
    multi(**dev, **refs, out_partials=pb, fs=f2s, f_inc=f1, in_eta=eta_d, dt=dt, **com)
    torch.cuda.synchronize()
    sa, sb = pa.sum(dim=0).cpu().numpy(), pb.sum(dim=0).cpu().numpy()
    mag = float(np.abs(sa).max())
    assert mag > 0 and np.all(np.abs(sa - sb) <= (1e-11 if dtype == np.float64 else 1e-4) * mag), (sa, sb)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_reduction_kernels_match_torch(gpu, dtype):
    """`field_sums` / `column_dots` on aligned fields, on a column window of wider storages (lev_stride > nx) and on a
    size that is no multiple of the workgroup; 10 and 16 fields; with and without subtrahend"""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.reductions import column_dots, field_sums

    td = storage.torch_dtype(dtype)
    g = torch.Generator(device="cpu").manual_seed(3)
    for nx, nz, pad in ((1000, 137, 0), (777, 20, 5), (64, 3, 0)):
        wide = [torch.randn((nz + 1, nx + pad), generator=g, dtype=torch.float64).to(td).to(gpu) for _ in range(32)]
        fields = [storage.logical_view(w[:, pad // 2: pad // 2 + nx]) for w in wide]
        a, b = fields[:16], fields[16:]
        for nf in (10, 16, 1):
            got = field_sums(a[:nf], b[:nf]).cpu().numpy()
            want = np.array([float((x - y).sum(dtype=torch.float64)) for x, y in zip(a[:nf], b[:nf])])
            mag = np.array([float((x - y).abs().sum(dtype=torch.float64)) for x, y in zip(a[:nf], b[:nf])])
            assert np.all(np.abs(got - want) <= 1e-13 * mag), (nx, nf)
            got = field_sums(a[:nf]).cpu().numpy()
            want = np.array([float(x.sum(dtype=torch.float64)) for x in a[:nf]])
            assert np.all(np.abs(got - want) <= 1e-13 * np.array([float(x.abs().sum(dtype=torch.float64)) for x in a[:nf]]))
        for pairs in (10, 16, 20):          # 20 pairs = two launches, the second accumulating
            aa, bb = (a + b)[:pairs], (b + a)[:pairs]
            got = column_dots(aa, bb).cpu().numpy()
            want = sum((x[:, 0, :].double() * y[:, 0, :].double()).sum(dim=1) for x, y in zip(aa, bb)).cpu().numpy()
            mag = sum((x[:, 0, :].double() * y[:, 0, :].double()).abs().sum(dim=1) for x, y in zip(aa, bb)).cpu().numpy()
            assert got.shape == (nx,) and np.all(np.abs(got - want) <= 1e-13 * mag)
        sq = column_dots(a[:10]).cpu().numpy()
        want = sum((x[:, 0, :].double() ** 2).sum(dim=1) for x in a[:10]).cpu().numpy()
        assert np.all(np.abs(sq - want) <= 1e-13 * want)
    with pytest.raises(ValueError):
        field_sums([fields[0]] * 17)


def _taylor_ctx(argv):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_taylor_test

    return run_taylor_test.main(["--backend", "hip", "--num-runs", "2"] + argv)


@pytest.mark.parametrize("precision", ["double", "single"])
def test_taylor_driver_modes_agree(gpu, precision, capsys):
    """plain, --fused (= --fused-norms), --fused-stored, --fused-all, --graph (alone and with the fused modes): the same norms (summation
    order apart) and the same verdict of the reference's scoring rule, on the reader path and on distinct columns"""
    for inp, cols in (("auto", 4096), ("synthetic", 3000)):
        base = _taylor_ctx(["--num-cols", str(cols), "--input", inp, "--precision", precision])
        verdict = [l for l in capsys.readouterr().out.splitlines() if l.startswith("The test ")]
        for extra in (["--fused"], ["--fused-norms"], ["--fused-stored"], ["--fused-all"], ["--graph"], ["--fused", "--graph"],
                      ["--fused-stored", "--graph"], ["--fused-all", "--graph"]):
            ctx = _taylor_ctx(["--num-cols", str(cols), "--input", inp, "--precision", precision] + extra)
            out = [l for l in capsys.readouterr().out.splitlines() if l.startswith("The test ")]
            # norms: ratios of sums; a different summation order moves them by rounding only while the perturbation is
            # well above the noise floor of the field type (the last step sizes are noise in every mode)
            n0, n1 = np.asarray(base["norms"]), np.asarray(ctx["norms"])
            k = 6 if precision == "double" else 2
            assert np.allclose(n1[:k], n0[:k], rtol=1e-7 if precision == "double" else 1e-3, atol=0), (extra, n0, n1)
            if precision == "double":
                assert out[0] == verdict[0], (extra, out, verdict)
            assert ctx["passed"] == base["passed"] or precision == "single"


def test_symmetry_driver_graph_mode(gpu, capsys):
    """--graph replays the four launches of the timed call: the adjoint fields it leaves are those of the plain call"""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import run_symmetry_test

    a = run_symmetry_test.main(["--backend", "hip", "--num-cols", "4096", "--num-runs", "3"])
    for extra in (["--graph"], ["--fused"], ["--fused", "--graph"]):
        b = run_symmetry_test.main(["--backend", "hip", "--num-cols", "4096", "--num-runs", "3"] + extra)
        assert a["passed"] and b["passed"]
        assert a["detail"]["columns_passing"] == b["detail"]["columns_passing"] == 4096
        for dct in ("tends_ad", "diags_ad", "tends_tl", "diags_tl"):
            da, db = getattr(a["harness"], dct), getattr(b["harness"], dct)
            assert set(da) == set(db) and len(da) >= 4
            for k in da:
                if hasattr(da[k], "data"):
                    if "--fused" in extra:      # another kernel instantiation: same arithmetic, contraction may differ
                        x, y = da[k].data.as_subclass(torch.Tensor), db[k].data.as_subclass(torch.Tensor)
                        assert float((x - y).abs().max()) <= 1e-9 * max(float(y.abs().max()), 1e-300), (extra, dct, k)
                    else:                        # graph replay of the same launches: bit for bit
                        assert torch.equal(da[k].data, db[k].data), (extra, dct, k)
    assert "HOORAY" in capsys.readouterr().out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("flags", [dict(), dict(LREGCL=False, AD_TRAJ_FIX=1)])
def test_ad_from_trajectory_equals_cloudsc2_ad_and_the_oracle(gpu, flags, dtype):
    """BUILD EXTENSION `cloudsc2_ad_from_trajectory` (r04): cloudsc2_ad without its forward sweep, fed with the flux outputs of
    a call on the same state.  (a) With the fluxes cloudsc2_ad itself wrote, the 16 adjoint fields are the BITS of
    cloudsc2_ad's; (b) with the fluxes of the cloudsc2_tl call that precedes it in the symmetry test
    (adjoint/validation.py:135-151) they agree to rounding and match the oracle's cloudsc2_ad like cloudsc2_ad does; (c) the
    forcings and the trajectory fields are left untouched, nothing but the adjoints is written; (d) the evaporation block is
    refused by name."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil
    from helpers import NL_IN, from_device, increments, run_oracle_ad, run_oracle_tl
    from test_hip_tl_ad import run_hip_ad, run_hip_tl

    lev = lambda n: nz + 1 if n in ("aph", "lu") else nz  # noqa: E731  (out_aph_i / out_lu_i are written on all nz+1 levels)
    nx, nz = 333, 137
    ext = externals(NLEV=nz, **flags)
    fields, eta, dt = nl_case(nx, dtype=dtype, seed=83)
    fi = increments(fields, 0.01, ignore_supsat=True)
    _, tl_i = run_oracle_tl(fields, fi, eta, dt, ext)
    forcing = {n: tl_i[n] for n in NL_OUT}                       # the symmetry test's forcing: the TL perturbation outputs
    want_nl, want_adj = run_oracle_ad(fields, forcing, eta, dt, ext)
    ad_nl, ad_adj = run_hip_ad(fields, forcing, eta, dt, ext, gpu, nx, nz)
    tl_nl, _ = run_hip_tl(fields, fi, eta, dt, ext, gpu, nx, nz)

    dev = to_device(fields, gpu)
    frc = to_device({"in_" + n + "_i": forcing[n] for n in NL_OUT}, gpu)
    st = compile_stencil("cloudsc2_ad_from_trajectory", ext)
    com = dict(in_eta=torch.as_tensor(eta, device=gpu), dt=dtype(dt), origin=(0, 0, 0), domain=(nx, 1, nz + 1),
               validate_args=True, exec_info=None)

    def run(traj):
        tr = to_device({"traj_fplsl": traj["fplsl"], "traj_fplsn": traj["fplsn"]}, gpu)
        before = {k: v.clone() for k, v in {**frc, **tr}.items()}
        outs = {"out_" + n + "_i": storage.from_klayout(np.full((nz + 1, nx), np.nan, dtype=dtype), dtype, gpu) for n in NL_IN}
        st(**dev, **frc, **tr, **outs, **com)
        torch.cuda.synchronize()
        assert _lib.last_kernel() == "cs2::ad_kernel<trajectory>"
        for k, v in before.items():
            assert torch.equal({**frc, **tr}[k], v), k                                        # (c) read-only
        return {n: from_device(outs["out_" + n + "_i"]) for n in NL_IN}

    own = run(ad_nl)                                              # (a) cloudsc2_ad's own recomputed fluxes
    for n in NL_IN:
        assert np.array_equal(own[n][:lev(n)], ad_adj[n][:lev(n)], equal_nan=True), n
    from_tl = run(tl_nl)                                          # (b) the TL call's fluxes
    rt = 1e3 if dtype == np.float64 else 1e2
    for n in NL_IN:
        k = lev(n)
        assert_close(f"ad_from_trajectory(TL fluxes) vs cloudsc2_ad out_{n}_i", from_tl[n][:k], ad_adj[n][:k], dtype, rtol_mul=rt)
        assert_close(f"ad_from_trajectory vs oracle out_{n}_i", from_tl[n][:k], want_adj[n][:k], dtype, rtol_mul=rt)
    with pytest.raises(ValueError, match="no evaporation"):      # (d)
        compile_stencil("cloudsc2_ad_from_trajectory", externals(NLEV=nz, LEVAPLS2=True))(
            **dev, **frc, **to_device({"traj_fplsl": ad_nl["fplsl"], "traj_fplsn": ad_nl["fplsn"]}, gpu),
            **{"out_" + n + "_i": storage.zeros(nx, nz, dtype, gpu) for n in NL_IN}, **com)
