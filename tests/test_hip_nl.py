"""GPU parity of the hand-written cloudsc2_nl / saturation kernels against the oracle.

Reads like the reference's own NL check (drivers/run_nonlinear.py:139-147: run saturation + NL,
compare tendencies and diagnostics field by field) with the oracle in the role of the golden file.
"""
import numpy as np
import pytest

from helpers import (NL_OUT, assert_close, externals, from_device, nl_case, run_oracle_nl, to_device)

pytestmark = pytest.mark.gpu


def run_hip_nl(fields, eta, dt, ext, device, nx, nz):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    dev = to_device(fields, device)
    dtype = fields["in_ap"].dtype
    outs = {"out_" + n: storage.zeros(nx, nz, dtype, device) for n in NL_OUT}
    for o in outs.values():
        o.fill_(float("nan"))  # every element the kernel owns must be written
    st = compile_stencil("cloudsc2_nl", ext)
    st(**dev, **outs, in_eta=torch.as_tensor(eta, device=device), tmp_aph_s=None, tmp_covptot=None,
       tmp_rfl=None, tmp_sfl=None, tmp_trpaus=None, dt=dtype.type(dt), origin=(0, 0, 0),
       domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    torch.cuda.synchronize()
    return {n: from_device(outs["out_" + n]) for n in NL_OUT}


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nx", [1, 64, 100, 333, 1024])
def test_nl_matches_oracle(gpu, nx, dtype):
    ext = externals()
    fields, eta, dt = nl_case(nx, dtype=dtype)
    want = run_oracle_nl(fields, eta, dt, ext)
    got = run_hip_nl(fields, eta, dt, ext, gpu, nx, 137)
    for n in NL_OUT:
        full = n in ("fhpsl", "fhpsn", "fplsl", "fplsn")
        nlev = 138 if full else 137
        assert_close(f"out_{n}[nx={nx},{np.dtype(dtype)}]", got[n][:nlev], want[n][:nlev], dtype)
        if not full:
            assert np.isnan(got[n][137]).all(), "padding level of a full-level output must stay untouched"


@pytest.mark.parametrize("flags", [dict(LEVAPLS2=True), dict(LDRAIN1D=True), dict(LPHYLIN=False),
                                   dict(LPHYLIN=False, LEVAPLS2=True)])
def test_nl_switches(gpu, flags):
    ext = externals(**flags)
    fields, eta, dt = nl_case(256, seed=11)
    want = run_oracle_nl(fields, eta, dt, ext)
    got = run_hip_nl(fields, eta, dt, ext, gpu, 256, 137)
    if flags.get("LEVAPLS2") or flags.get("LDRAIN1D"):
        assert (want["covptot"] > 0).any(), "case must exercise the evaporation block"
    for n in NL_OUT:
        nlev = 138 if n.startswith("f") else 137
        assert_close(f"out_{n}{flags}", got[n][:nlev], want[n][:nlev])


def test_nl_short_columns(gpu):
    """nz != 137: the kernel takes nz from the storages (odd and even level counts)."""
    for nz in (7, 20):
        ext = externals()
        fields, eta, dt = nl_case(70, nz=nz, seed=5)
        want = run_oracle_nl(fields, eta, dt, ext)
        got = run_hip_nl(fields, eta, dt, ext, gpu, 70, nz)
        for n in NL_OUT:
            nlev = nz + 1 if n.startswith("f") else nz
            assert_close(f"out_{n}[nz={nz}]", got[n][:nlev], want[n][:nlev])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nx", [300, 301])
def test_saturation_matches_oracle(gpu, dtype, nx):
    """nx = 300: the 16-byte-per-lane kernel (aligned rows); nx = 301: the scalar kernel."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil
    from oracle import cloudsc2_numpy as oracle

    for flags in (dict(), dict(LPHYLIN=False, KFLAG=1), dict(LPHYLIN=False, KFLAG=0)):
        ext = externals(**flags)
        fields, _, _ = nl_case(nx, dtype=dtype)
        want = np.zeros_like(fields["in_t"])
        oracle.saturation(fields["in_ap"], fields["in_t"], want, ext)
        dev = to_device({k: fields[k] for k in ("in_ap", "in_t")}, gpu)
        out = storage.zeros(nx, 137, dtype, gpu)
        compile_stencil("saturation", ext)(**dev, out_qsat=out, origin=(0, 0, 0), domain=(nx, 1, 137),
                                            validate_args=True, exec_info=None)
        torch.cuda.synchronize()
        got = from_device(out)
        assert_close(f"qsat{flags}", got[:137], want[:137], dtype)
        assert np.all(got[137] == 0)


def test_nl_properties_at_full_size(gpu):
    """BASELINE config 2 size (65 536 x 137, fp64): size-independent properties instead of the oracle:
    flux/enthalpy relation (cloudsc2.py:396-399), ranges, shard invariance (a 512-column slice run on
    its own reproduces the same columns bit for bit), and the first 256 columns against the oracle."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    nx, nz = 65536, 137
    ext = externals()
    s = make_state(nx, nz, device=gpu)
    eta = torch.as_tensor(eta_levels(nz), device=gpu)
    f = {k: storage.logical_view(v) for k, v in s.items()}
    qsat = storage.zeros(nx, nz, np.float64, gpu)
    compile_stencil("saturation", ext)(in_ap=f["f_ap"], in_t=f["f_t"], out_qsat=qsat, origin=(0, 0, 0),
                                        domain=(nx, 1, nz), validate_args=True, exec_info=None)
    ins = {"in_" + k[2:]: v for k, v in f.items()}
    ins["in_qsat"] = qsat
    outs = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    nl = compile_stencil("cloudsc2_nl", ext)
    nl(**ins, **outs, in_eta=eta, dt=3600.0, origin=(0, 0, 0), domain=(nx, 1, nz + 1),
       validate_args=True, exec_info=None)
    torch.cuda.synchronize()
    o = {n: storage.klayout(outs["out_" + n]) for n in NL_OUT}
    assert not any(torch.isnan(v).any().item() for v in o.values())
    assert torch.equal(o["fhpsn"], -o["fplsn"] * ext["RLSTT"])
    assert torch.equal(o["fhpsl"], -o["fplsl"] * ext["RLVTT"])
    assert (o["covptot"] == 0).all() and (o["clc"] >= 0).all() and (o["clc"] <= 1).all()
    assert (o["fplsl"] >= 0).all() and (o["fplsn"] >= 0).all()
    # shard invariance: columns [4096, 4608) alone
    c0, n = 4096, 512
    sub_in = {k: storage.logical_view(storage.klayout(v)[:, c0:c0 + n].contiguous()) for k, v in ins.items()}
    sub_out = {"out_" + m: storage.zeros(n, nz, np.float64, gpu) for m in NL_OUT}
    nl(**sub_in, **sub_out, in_eta=eta, dt=3600.0, origin=(0, 0, 0), domain=(n, 1, nz + 1),
       validate_args=True, exec_info=None)
    for m in NL_OUT:
        assert torch.equal(storage.klayout(sub_out["out_" + m]), o[m][:, c0:c0 + n]), m
    # strided views (lev_stride > nx) of the big storages give the same answer without a copy
    view_in = {k: storage.logical_view(storage.klayout(v)[:, c0:c0 + n]) for k, v in ins.items()}
    view_out = {"out_" + m: storage.logical_view(storage.klayout(outs["out_" + m])[:, c0:c0 + n]) for m in NL_OUT}
    ref = {m: o[m][:, c0:c0 + n].clone() for m in NL_OUT}
    nl(**view_in, **view_out, in_eta=eta, dt=3600.0, origin=(0, 0, 0), domain=(n, 1, nz + 1),
       validate_args=True, exec_info=None)
    for m in NL_OUT:
        assert torch.equal(o[m][:, c0:c0 + n], ref[m]), m
    # first 256 columns against the oracle
    host = {k: storage.klayout(v)[:, :256].cpu().numpy() for k, v in ins.items()}
    want = run_oracle_nl(host, eta.cpu().numpy(), 3600.0, ext)
    for m in NL_OUT:
        nlev = 138 if m.startswith("f") else 137
        assert_close(f"full-size out_{m}", o[m][:nlev, :256].cpu().numpy(), want[m][:nlev])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nx", [1000, 1024])     # register-prefetch path / LDS-ring path of both kernels
def test_fused_saturation_variant_equals_separate_calls(gpu, dtype, nx):
    """Build extension `cloudsc2_nl_saturation` (one launch) == `saturation` then `cloudsc2_nl` (two launches)."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nz = 137
    ext = externals()
    fields, eta, dt = nl_case(nx, dtype=dtype)
    dev = to_device(fields, gpu)
    eta_d = torch.as_tensor(eta, device=gpu)
    com = dict(origin=(0, 0, 0), validate_args=True, exec_info=None)
    qsat = storage.zeros(nx, nz, dtype, gpu)
    compile_stencil("saturation", ext)(in_ap=dev["in_ap"], in_t=dev["in_t"], out_qsat=qsat, domain=(nx, 1, nz), **com)
    outs = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    ins = dict(dev)
    ins["in_qsat"] = qsat
    compile_stencil("cloudsc2_nl", ext)(**ins, **outs, in_eta=eta_d, dt=dt, domain=(nx, 1, nz + 1), **com)
    qsat2 = storage.zeros(nx, nz, dtype, gpu)
    outs2 = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    ins2 = {k: v for k, v in dev.items() if k != "in_qsat"}
    compile_stencil("cloudsc2_nl_saturation", ext)(**ins2, out_qsat=qsat2, **outs2, in_eta=eta_d, dt=dt,
                                                    domain=(nx, 1, nz + 1), **com)
    torch.cuda.synchronize()
    assert torch.equal(qsat2, qsat)                      # same arithmetic as the saturation kernel: bit-identical
    for n in NL_OUT:
        assert torch.equal(outs2["out_" + n], outs["out_" + n]), n


def test_fused_perturbation_variant_equals_separate_calls(gpu):
    """Build extension `cloudsc2_nl_perturbed` == `perturbed_state` then `cloudsc2_nl`."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, compile_stencil

    nx, nz, f2 = 777, 137, 1e-3
    ext = externals()
    fields, eta, dt = nl_case(nx)
    dev = to_device(fields, gpu)
    dev_i = to_device({k + "_i": (0.01 * v) for k, v in fields.items()}, gpu)
    eta_d = torch.as_tensor(eta, device=gpu)
    com = dict(origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    pert = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in INC}
    compile_stencil("perturbed_state", {})(**{"in_" + n: dev["in_" + n] for n in INC},
                                            **{"in_" + n + "_i": dev_i["in_" + n + "_i"] for n in INC}, **pert, f=f2, **com)
    outs = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl", ext)(**{"in_" + n: pert["out_" + n] for n in INC}, **outs, in_eta=eta_d, dt=dt, **com)
    outs2 = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl_perturbed", ext)(**dev, **dev_i, **outs2, in_eta=eta_d, dt=dt, f=f2, **com)
    torch.cuda.synchronize()
    for n in NL_OUT:
        assert torch.equal(outs2["out_" + n], outs["out_" + n]), n   # x + f*x_i is one fma in both paths


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nz", [137, 5, 3, 2])
def test_lds_ring_and_register_prefetch_paths_agree(gpu, dtype, nz):
    """cloudsc2_nl has two load paths (csrc/cloudsc2_nl.hip): the LDS-DMA ring (whole waves, 16-byte aligned rows) and
    the register prefetch (everything else).  The same 320 columns presented (a) as aligned contiguous storages and
    (b) as a window that starts at column 1 of wider storages (misaligned rows -> register path) must agree 100x
    tighter than the HIP-vs-oracle tolerance (the two kernels inline the same level function, but fma contraction may
    differ between the two contexts), and both must match the oracle.  nz = 2 is below the ring depth (register path
    both times)."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx = 320
    ext = externals(LEVAPLS2=True) if nz == 137 else externals()
    fields, eta, dt = nl_case(nx, nz=nz, dtype=dtype, seed=3)
    want = run_oracle_nl(fields, eta, dt, ext)
    nl = compile_stencil("cloudsc2_nl", ext)
    eta_d = torch.as_tensor(eta, device=gpu)
    com = dict(in_eta=eta_d, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    aligned = to_device(fields, gpu)
    out_a = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    nl(**aligned, **out_a, **com)
    wide = {k: torch.zeros((nz + 1, nx + 3), dtype=storage.torch_dtype(dtype), device=gpu) for k in fields}
    for k, v in fields.items():
        wide[k][:, 1:nx + 1] = torch.as_tensor(v, device=gpu)
    shifted = {k: storage.logical_view(v[:, 1:nx + 1]) for k, v in wide.items()}
    out_w = {"out_" + n: torch.zeros((nz + 1, nx + 3), dtype=storage.torch_dtype(dtype), device=gpu) for n in NL_OUT}
    nl(**shifted, **{k: storage.logical_view(v[:, 1:nx + 1]) for k, v in out_w.items()}, **com)
    torch.cuda.synchronize()
    for n in NL_OUT:
        a = storage.klayout(out_a["out_" + n]).cpu().numpy()
        w = out_w["out_" + n][:, 1:nx + 1].cpu().numpy()
        nlev = nz + 1 if n.startswith("f") else nz
        assert_close(f"ring vs register out_{n}[nz={nz}]", a[:nlev], w[:nlev], dtype, rtol_mul=1e-2)
        assert_close(f"ring out_{n}[nz={nz}]", a[:nlev], want[n][:nlev], dtype)
        assert (out_w["out_" + n][:, 0] == 0).all() and (out_w["out_" + n][:, nx + 1:] == 0).all()  # window respected


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_taylor_reduction_variant_matches_separate_calls(gpu, dtype):
    """`cloudsc2_nl_taylor` (build extension): perturbed NL run with the Taylor test's ten sums formed in the kernel
    epilogue == perturbed_state -> cloudsc2_nl -> (out_p - out).sum() done with separate kernels; the reference
    outputs are left untouched.  Sums differ only by summation order."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil, taylor_blocks

    nx, nz, f2 = 777, 137, 1e-2
    ext = externals()
    fields, eta, dt = nl_case(nx, dtype=dtype)
    dev = to_device(fields, gpu)
    inc = {k + "_i": storage.from_klayout(0.01 * storage.klayout(v), dtype, gpu) for k, v in dev.items()}   # same level pitch
    eta_d = torch.as_tensor(eta, device=gpu)
    com = dict(in_eta=eta_d, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    ref = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl", ext)(**dev, **ref, **com)
    pert = {"out_" + n: storage.zeros(nx, nz, dtype, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl_perturbed", ext)(**dev, **inc, **pert, f=f2, **com)
    want = torch.stack([(pert["out_" + n].double() - ref["out_" + n].double()).sum() for n in NL_OUT]).cpu().numpy()
    keep = {k: v.clone() for k, v in ref.items()}
    part = torch.full((taylor_blocks(nx), len(NL_OUT)), float("nan"), dtype=torch.float64, device=gpu)
    assert taylor_blocks(nx) == -(-nx // 256)
    compile_stencil("cloudsc2_nl_taylor", ext)(**dev, **inc, **{"ref_" + n: ref["out_" + n] for n in NL_OUT},
                                                out_partials=part, f=f2, **com)
    torch.cuda.synchronize()
    got = part.sum(dim=0).cpu().numpy()
    for k in ref:
        assert torch.equal(ref[k], keep[k]), k
    mag = torch.stack([(pert["out_" + n].double() - ref["out_" + n].double()).abs().sum() for n in NL_OUT]).cpu().numpy()
    tol = 1e-12 if dtype == np.float64 else 1e-6
    assert np.all(np.abs(got - want) <= tol * mag + 1e-300), (got, want)
    assert np.abs(want).max() > 0


def test_large_grid_takes_the_shallow_ring(gpu):
    """More than ~1.5 workgroups per CU: launch_nl selects ring depth 2 (two workgroups resident per CU).  131 072
    columns through that path against (a) the same columns run as a small separate call (deep ring) and (b) the oracle."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    nx, nz = 131072, 137
    ext = externals()
    s = make_state(nx, nz, device=gpu)
    eta = torch.as_tensor(eta_levels(nz), device=gpu)
    ins = {"in_" + k[2:]: storage.logical_view(v) for k, v in s.items()}
    qsat = storage.zeros(nx, nz, np.float64, gpu)
    compile_stencil("saturation", ext)(in_ap=ins["in_ap"], in_t=ins["in_t"], out_qsat=qsat, origin=(0, 0, 0),
                                        domain=(nx, 1, nz), validate_args=True, exec_info=None)
    ins["in_qsat"] = qsat
    outs = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    nl = compile_stencil("cloudsc2_nl", ext)
    nl(**ins, **outs, in_eta=eta, dt=3600.0, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True,
       exec_info=None)
    c0, n = 70000 - 70000 % 64, 512
    sub_in = {k: storage.logical_view(storage.klayout(v)[:, c0:c0 + n].contiguous()) for k, v in ins.items()}
    sub_out = {"out_" + m: storage.zeros(n, nz, np.float64, gpu) for m in NL_OUT}
    nl(**sub_in, **sub_out, in_eta=eta, dt=3600.0, origin=(0, 0, 0), domain=(n, 1, nz + 1), validate_args=True,
       exec_info=None)
    torch.cuda.synchronize()
    host = {k: storage.klayout(v).cpu().numpy() for k, v in sub_in.items()}
    want = run_oracle_nl(host, eta.cpu().numpy(), 3600.0, ext)
    for m in NL_OUT:
        nlev = 138 if m.startswith("f") else 137
        big = storage.klayout(outs["out_" + m])[:nlev, c0:c0 + n].cpu().numpy()
        small = storage.klayout(sub_out["out_" + m])[:nlev].cpu().numpy()
        assert_close(f"depth-2 vs depth-3 ring out_{m}", big, small, np.float64, rtol_mul=1e-2)
        assert_close(f"depth-2 ring out_{m}", big, want[m][:nlev])


def test_ring_kernel_is_deterministic_under_load(gpu):
    """The LDS-ring kernels wait on hand-counted vmcnt values; a wrong count would show as run-to-run differences
    (a lane reading a slot before its DMA landed).  200 launches at the headline size, alternating with the
    fused-saturation ring kernel and a large copy that perturbs memory latency, must all give the first launch's bits."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    nx, nz = 65536, 137
    ext = externals()
    s = make_state(nx, nz, device=gpu)
    eta = torch.as_tensor(eta_levels(nz), device=gpu)
    ins = {"in_" + k[2:]: storage.logical_view(v) for k, v in s.items()}
    qsat = storage.zeros(nx, nz, np.float64, gpu)
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    compile_stencil("saturation", ext)(in_ap=ins["in_ap"], in_t=ins["in_t"], out_qsat=qsat, domain=(nx, 1, nz), **com)
    nl = compile_stencil("cloudsc2_nl", ext)
    nls = compile_stencil("cloudsc2_nl_saturation", ext)
    ins_q = dict(ins, in_qsat=qsat)
    ref = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    nl(**ins_q, **ref, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
    out = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    q2 = storage.zeros(nx, nz, np.float64, gpu)
    big_a = torch.empty(1 << 26, dtype=torch.float64, device=gpu)
    big_b = torch.empty_like(big_a)
    bad = torch.zeros((), dtype=torch.int64, device=gpu)
    for it in range(200):
        if it % 3 == 2:
            big_b.copy_(big_a, non_blocking=True)
        if it % 2:
            nls(**ins, out_qsat=q2, **out, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
            bad += (q2 != qsat).sum()
        else:
            nl(**ins_q, **out, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
        for n in ("out_tnd_t", "out_fplsn", "out_clc"):
            bad += (out[n] != ref[n]).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0


def test_config5_shard_fp32(gpu):
    """BASELINE config 5, one GPU's share: CLOUDSC2-NL fp32 on 524 288 columns x 137 levels (4 194 304 columns over 8 GPUs).
    Size-independent checks on the whole shard (finite, cover in [0, 1], non-negative fluxes, enthalpy-flux identity),
    equality with a separate call on a 512-column block, and the oracle on that block."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    total, nx, nz, rank = 4194304, 524288, 137, 3
    ext = externals()
    s = make_state(total, nz, col0=rank * nx, ncols=nx, dtype=np.float32, device=gpu)
    eta = torch.as_tensor(eta_levels(nz, dtype=np.float32), device=gpu)
    ins = {"in_" + k[2:]: storage.logical_view(v) for k, v in s.items()}
    qsat = storage.zeros(nx, nz, np.float32, gpu)
    com = dict(origin=(0, 0, 0), validate_args=True, exec_info=None)
    compile_stencil("saturation", ext)(in_ap=ins["in_ap"], in_t=ins["in_t"], out_qsat=qsat, domain=(nx, 1, nz), **com)
    ins["in_qsat"] = qsat
    outs = {"out_" + n: storage.zeros(nx, nz, np.float32, gpu) for n in NL_OUT}
    nl = compile_stencil("cloudsc2_nl", ext)
    nl(**ins, **outs, in_eta=eta, dt=3600.0, domain=(nx, 1, nz + 1), **com)
    o = {n: storage.klayout(outs["out_" + n]) for n in NL_OUT}
    for n in NL_OUT:
        assert bool(torch.isfinite(o[n]).all()), n
    assert float(o["clc"].min()) >= 0.0 and float(o["clc"].max()) <= 1.0
    assert float(o["fplsl"].min()) >= -1e-12 and float(o["fplsn"].min()) >= -1e-12
    assert torch.allclose(o["fhpsl"], -o["fplsl"] * np.float32(ext["RLVTT"]), rtol=1e-6, atol=0.0)
    assert torch.allclose(o["fhpsn"], -o["fplsn"] * np.float32(ext["RLSTT"]), rtol=1e-6, atol=0.0)
    c0, n = 300032, 512
    sub_in = {k: storage.logical_view(storage.klayout(v)[:, c0:c0 + n].contiguous()) for k, v in ins.items()}
    sub_out = {"out_" + m: storage.zeros(n, nz, np.float32, gpu) for m in NL_OUT}
    nl(**sub_in, **sub_out, in_eta=eta, dt=3600.0, domain=(n, 1, nz + 1), **com)
    torch.cuda.synchronize()
    host = {k: storage.klayout(v).cpu().numpy() for k, v in sub_in.items()}
    want = run_oracle_nl(host, eta.cpu().numpy(), 3600.0, ext)
    for m in NL_OUT:
        nlev = 138 if m.startswith("f") else 137
        big = o[m][:nlev, c0:c0 + n].cpu().numpy()
        small = storage.klayout(sub_out["out_" + m])[:nlev].cpu().numpy()
        assert_close(f"shard vs block out_{m}", big, small, np.float32, rtol_mul=1e-2)
        assert_close(f"shard out_{m}", big, want[m][:nlev], np.float32)


def test_empty_calls_are_no_ops(gpu):
    """Zero columns (an empty rank of a ragged partition): every stencil accepts zero-size storages and does nothing."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, NL_IN, compile_stencil

    nz = 137
    ext = externals(NLEV=nz)
    z = lambda: storage.zeros(0, nz, np.float64, gpu)  # noqa: E731
    eta = torch.zeros(nz + 1, dtype=torch.float64, device=gpu)
    com = dict(origin=(0, 0, 0), validate_args=True, exec_info=None)
    compile_stencil("saturation", ext)(in_ap=z(), in_t=z(), out_qsat=z(), domain=(0, 1, nz), **com)
    ins = {"in_" + n: z() for n in NL_IN}
    ins_i = {"in_" + n + "_i": z() for n in NL_IN}
    compile_stencil("cloudsc2_nl", ext)(**ins, **{"out_" + n: z() for n in NL_OUT}, in_eta=eta, dt=3600.0,
                                         domain=(0, 1, nz + 1), **com)
    compile_stencil("cloudsc2_tl", ext)(**ins, **ins_i, **{"out_" + n: z() for n in NL_OUT},
                                         **{"out_" + n + "_i": z() for n in NL_OUT}, in_eta=eta, dt=3600.0,
                                         domain=(0, 1, nz + 1), **com)
    compile_stencil("cloudsc2_ad", ext)(**ins, **{"in_" + n + "_i": z() for n in NL_OUT},
                                         **{"out_" + n: z() for n in NL_OUT}, **{"out_" + n + "_i": z() for n in NL_IN},
                                         in_eta=eta, dt=3600.0, domain=(0, 1, nz + 1), **com)
    compile_stencil("state_increment", {"IGNORE_SUPSAT": False})(
        **{"in_" + n: z() for n in INC}, **{"out_" + n + "_i": z() for n in INC}, f=0.01, domain=(0, 1, nz + 1), **com)
    torch.cuda.synchronize()


@pytest.mark.parametrize("sw", [dict(), dict(LEVAPLS2=True)])
def test_fused_perturbation_with_general_increments(gpu, sw):
    """The fused perturbed variants must perturb EVERYTHING the stencil reads, also what it reads outside the level loop:
    the tropopause pre-scan's t / tnd_cml_t, aph at the top half level and (evaporation block) at the surface.  With
    increments proportional to the state none of that shows (the ordering t[k] > t[k+1] is scale-invariant, aph[0] = 0);
    here the increments are random, of order one and aph[0] != 0, so the tropopause level moves in many columns.
    `cloudsc2_nl_perturbed` == perturbed_state then cloudsc2_nl, and `cloudsc2_nl_taylor` sums the same differences."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, compile_stencil, taylor_blocks

    nx, nz, f2 = 640, 137, 1.0
    ext = externals(**sw)
    fields, eta, dt = nl_case(nx, ext=ext)
    fields["in_aph"][0] = 150.0 + 10.0 * np.arange(nx) / nx            # a model top that is not at p = 0
    rng = np.random.default_rng(17)
    inc = {}
    for k, v in fields.items():
        inc[k + "_i"] = (v * rng.uniform(-0.02, 0.02, size=v.shape)).astype(v.dtype)
    inc["in_t_i"] = rng.normal(0.0, 2.5, size=fields["in_t"].shape) * (fields["in_t"] != 0)      # +-2.5 K: moves trpaus
    inc["in_aph_i"][0] = 3.0
    dev = to_device({**fields, **inc}, gpu)
    eta_d = torch.as_tensor(eta, device=gpu)
    com = dict(origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    pert = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in INC}
    compile_stencil("perturbed_state", {})(**{"in_" + n: dev["in_" + n] for n in INC},
                                            **{"in_" + n + "_i": dev["in_" + n + "_i"] for n in INC}, **pert, f=f2, **com)
    nl = compile_stencil("cloudsc2_nl", ext)
    outs = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    nl(**{"in_" + n: pert["out_" + n] for n in INC}, **outs, in_eta=eta_d, dt=dt, **com)
    ref = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    nl(**{k: v for k, v in dev.items() if not k.endswith("_i")}, **ref, in_eta=eta_d, dt=dt, **com)
    # the perturbation must really move the tropopause somewhere, or this test proves nothing
    moved = (storage.klayout(outs["out_clc"]) != storage.klayout(ref["out_clc"])).any(dim=0).float().mean()
    assert float(moved) > 0.5
    outs2 = {"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT}
    compile_stencil("cloudsc2_nl_perturbed", ext)(**dev, **outs2, in_eta=eta_d, f=f2, dt=dt, **com)
    for n in NL_OUT:
        assert torch.equal(outs2["out_" + n], outs["out_" + n]), n
    part = torch.empty((taylor_blocks(nx), len(NL_OUT)), dtype=torch.float64, device=gpu)
    compile_stencil("cloudsc2_nl_taylor", ext)(**dev, **{"ref_" + n: ref["out_" + n] for n in NL_OUT},
                                                out_partials=part, in_eta=eta_d, f=f2, dt=dt, **com)
    torch.cuda.synchronize()
    got = part.sum(dim=0).cpu().numpy()
    want = np.array([float((outs["out_" + n].double() - ref["out_" + n].double()).sum()) for n in NL_OUT])
    mag = np.array([float((outs["out_" + n].double() - ref["out_" + n].double()).abs().sum()) for n in NL_OUT])
    assert np.all(np.abs(got - want) <= 1e-12 * mag + 1e-300), (got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,nx", [(np.float64, 1), (np.float64, 2), (np.float64, 66), (np.float64, 333), (np.float64, 1000),
                                      (np.float64, 65499), (np.float64, 65500), (np.float32, 3), (np.float32, 4),
                                      (np.float32, 68), (np.float32, 1001), (np.float32, 20004)])
def test_ragged_last_wave_takes_the_ring(gpu, dtype, nx):
    """r03: a call whose nx is NOT a multiple of 64 keeps the LDS-ring path (`nl_ring_kernel<.., RAGGED>`) as long as its rows
    are 16-byte aligned and hold the last DMA-wide group of columns (lev_stride >= nx rounded up to 2 fp64 / 4 fp32 columns -
    what `storage.zeros` allocates): the dead lanes of the partly filled last wave fetch that group, compute on the copy and
    do not store.  Held against the oracle; the storages are wider than nx and the columns beyond nx must stay untouched - a
    store that ran off the window would show there, a DMA that ran off the row would fault."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nz = 137
    per16 = 16 // np.dtype(dtype).itemsize
    pad = 8 + (-(nx + 8)) % per16               # lev_stride = nx + pad: a multiple of the DMA width, rows 16-byte aligned
    ext = externals()
    fields, eta, dt = nl_case(nx, dtype=dtype, seed=41)
    big = nx > 4096
    want = run_oracle_nl({k: v[:, :2048] for k, v in fields.items()} if big else fields, eta, dt, ext)
    td = storage.torch_dtype(dtype)
    wide_in = {k: torch.full((nz + 1, nx + pad), 7.0, dtype=td, device=gpu) for k in fields}
    for k, v in fields.items():
        wide_in[k][:, :nx] = torch.as_tensor(v, device=gpu)
    wide_out = {"out_" + n: torch.full((nz + 1, nx + pad), -3.0, dtype=td, device=gpu) for n in NL_OUT}
    view = lambda d: {k: storage.logical_view(v[:, :nx]) for k, v in d.items()}  # noqa: E731
    compile_stencil("cloudsc2_nl", ext)(**view(wide_in), **view(wide_out), in_eta=torch.as_tensor(eta, device=gpu), dt=dt,
                                         origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    torch.cuda.synchronize()
    assert _lib.last_kernel() == ("cs2::nl_ring_kernel<ragged>" if nx % 64 else "cs2::nl_ring_kernel")
    ncmp = 2048 if big else nx
    for n in NL_OUT:
        got = wide_out["out_" + n].cpu().numpy()
        k = nz + 1 if n.startswith("f") else nz
        assert_close(f"ragged ring out_{n} nx={nx}", got[:k, :ncmp], want[n][:k], dtype)
        assert np.all(got[:, nx:] == -3.0), n                              # nothing written beyond the window
        assert np.all(np.isfinite(got[:k, :nx])), n
    if big:      # the last (partly filled) wave too: its columns against the oracle on those columns alone
        tail = {k: np.ascontiguousarray(v[:, nx - 60:]) for k, v in fields.items()}
        want_t = run_oracle_nl(tail, eta, dt, ext)
        for n in NL_OUT:
            k = nz + 1 if n.startswith("f") else nz
            assert_close(f"ragged ring tail out_{n}", wide_out["out_" + n][:k, nx - 60:nx].cpu().numpy(), want_t[n][:k], dtype)
