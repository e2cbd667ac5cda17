"""Fields of 4 GiB and more.  The column kernels address a field element with a byte offset from the
field's base; in 32 bits that stops at (nz+1) * lev_stride * itemsize = 2^32 - 3.89 M fp64 columns at 137 levels, far below
what 288 GB of HBM hold.  Since r04 `cloudsc2_nl` / `_tl` / `_ad` switch to a 64-bit-offset instantiation of their
register-path kernel for such fields (`cs2::*_kernel<big>`) instead of refusing the call; the C ABI is unchanged.  What
overflows is the LEVEL STRIDE, not nx, so the cheap way to exercise the path is a narrow column window of a wide
allocation: every field of a call is its own 1 000-column window of ONE 4.6 GB buffer."""
import numpy as np
import pytest

from helpers import NL_IN, NL_OUT, assert_close, externals, from_device, increments, nl_case, run_oracle_nl, to_device

NZ = 137
WIDE = 4_194_304            # columns of the wide allocation: 138 x 4 194 304 x 8 B = 4.63 GB per field > 2^32


@pytest.mark.gpu
@pytest.mark.parametrize("switches", [{}, {"LEVAPLS2": True, "LREGCL": False}],
                         ids=["driver-defaults", "evaporation-block-no-LREGCL"])
def test_nl_tl_ad_on_windows_of_a_4_6_GB_allocation(gpu, switches):
    """NL, TL and AD through 1 000-column windows whose level stride is 4 194 304 fp64 elements: the launchers pick the
    `<big>` kernels, and their results are the BITS of the same columns in dense storages (same arithmetic on the same
    words; NL additionally against the oracle)."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    nx = 1000                               # not a multiple of 64: the dense calls take the register-path kernels too
    ext = externals(NLEV=NZ, **switches)    # the evaporation block adds accesses of its own (aph[nz], the parked cover of AD)
    fields, eta, dt = nl_case(nx, seed=71)
    fi = increments(fields, 0.01, ignore_supsat=True)
    eta_d = torch.as_tensor(eta, device=gpu)
    wide = torch.zeros((NZ + 1, WIDE), dtype=torch.float64, device=gpu)
    assert wide.numel() * 8 > 2 ** 32
    slot = [0]

    def win(v=None):
        c0 = slot[0] * 1100 + 37              # windows [37, 1037) of 1 100-column slots: 100 untouched columns between them
        slot[0] += 1
        t = wide[:, c0:c0 + nx]
        if v is not None:
            t.copy_(torch.as_tensor(v, device=gpu))
        return storage.logical_view(t)

    com = dict(in_eta=eta_d, dt=dt, origin=(0, 0, 0), domain=(nx, 1, NZ + 1), validate_args=True, exec_info=None)
    dev = to_device({**fields, **fi}, gpu)
    big = {k: win(v) for k, v in {**fields, **fi}.items()}
    assert all(v.stride(2) == WIDE for v in big.values())

    # ---- NL
    nl = compile_stencil("cloudsc2_nl", ext)
    nl_d = {"out_" + n: storage.zeros(nx, NZ, np.float64, gpu) for n in NL_OUT}
    nl(**{k: v for k, v in dev.items() if not k.endswith("_i")}, **nl_d, **com)
    nl_b = {"out_" + n: win() for n in NL_OUT}
    nl(**{k: v for k, v in big.items() if not k.endswith("_i")}, **nl_b, **com)
    assert _lib.last_kernel() == "cs2::nl_kernel<big>"
    want = run_oracle_nl(fields, eta, dt, ext)
    for n in NL_OUT:
        assert torch.equal(nl_b["out_" + n], nl_d["out_" + n]), n
        nlev = NZ + 1 if n.startswith("f") else NZ
        assert_close("big out_" + n, from_device(nl_b["out_" + n])[:nlev], want[n][:nlev])
    # ---- TL
    tl = compile_stencil("cloudsc2_tl", ext)
    tl_d = {"out_" + n + s: storage.zeros(nx, NZ, np.float64, gpu) for n in NL_OUT for s in ("", "_i")}
    tl(**dev, **tl_d, **com)
    tl_b = {k: win() for k in tl_d}
    tl(**big, **tl_b, **com)
    assert _lib.last_kernel() == "cs2::tl_kernel<big>"
    for k in tl_d:
        assert torch.equal(tl_b[k], tl_d[k]), k
    # ---- AD, forced with the TL perturbation outputs
    ad = compile_stencil("cloudsc2_ad", ext)
    ad_d = {"out_" + n: storage.zeros(nx, NZ, np.float64, gpu) for n in NL_OUT}
    ad_d.update({"out_" + n + "_i": storage.zeros(nx, NZ, np.float64, gpu) for n in NL_IN})
    ad(**{k: v for k, v in dev.items() if not k.endswith("_i")}, **{"in_" + n + "_i": tl_d["out_" + n + "_i"] for n in NL_OUT},
       **ad_d, **com)
    ad_b = {k: win() for k in ad_d}
    ad(**{k: v for k, v in big.items() if not k.endswith("_i")}, **{"in_" + n + "_i": tl_b["out_" + n + "_i"] for n in NL_OUT},
       **ad_b, **com)
    assert _lib.last_kernel() == "cs2::ad_kernel<big>"
    for k in ad_d:
        assert torch.equal(ad_b[k], ad_d[k]), k
    # the fused build extensions keep 32-bit offsets and say so
    with pytest.raises(ValueError, match="2\\^32"):
        compile_stencil("cloudsc2_nl_saturation", ext)(
            **{k: v for k, v in big.items() if not k.endswith("_i") and k != "in_qsat"}, out_qsat=win(), **nl_b, **com)
    slots_used = slot[0]
    assert slots_used * 1100 <= WIDE
    # nothing outside the windows was written: the gaps between windows are still zero
    slots = wide[:, :slots_used * 1100].view(NZ + 1, slots_used, 1100)
    assert float(slots[:, :, :37].abs().sum()) == 0.0 and float(slots[:, :, 37 + nx:].abs().sum()) == 0.0


@pytest.mark.gpu
def test_nl_at_4194304_fp64_columns(gpu):
    """cloudsc2_nl on 4 194 304 fp64 columns x 137 levels: 26 fields of 4.63 GB = 120 GB resident on one GPU, ONE call.
    Property checks at full size (every output finite, the padding level untouched, fluxes non-negative) and the first and
    the last 512-column windows against the oracle on the same columns."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    free = torch.cuda.mem_get_info(gpu)[0]
    if free < 150e9:
        pytest.skip(f"needs ~125 GB of free device memory, {free / 1e9:.0f} GB available")
    nx, chunk = WIDE, 262144
    ext = externals()
    dt = 3600.0
    eta = eta_levels(NZ, dtype=np.float64)
    eta_d = torch.as_tensor(eta, device=gpu)
    F = {"in_" + n: storage.zeros(nx, NZ, np.float64, gpu) for n in NL_IN}
    for c in range(0, nx, chunk):                                   # the state, generated on the device chunk by chunk
        part = make_state(nx, NZ, col0=c, ncols=chunk, dtype=np.float64, device=gpu)
        for k, v in part.items():
            if "in_" + k[2:] in F:
                storage.klayout(F["in_" + k[2:]])[:, c:c + chunk] = v
        del part
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    compile_stencil("saturation", ext)(in_ap=F["in_ap"], in_t=F["in_t"], out_qsat=F["in_qsat"], domain=(nx, 1, NZ), **com)
    out = {"out_" + n: storage.zeros(nx, NZ, np.float64, gpu) for n in NL_OUT}
    for v in out.values():
        storage.klayout(v)[NZ].fill_(-7.0)                        # the padding level of full-level fields must stay as it is
    assert storage.klayout(F["in_ap"]).numel() * 8 > 2 ** 32
    nl = compile_stencil("cloudsc2_nl", ext)
    nl(**F, **out, in_eta=eta_d, dt=dt, domain=(nx, 1, NZ + 1), **com)
    torch.cuda.synchronize()
    assert _lib.last_kernel() == "cs2::nl_kernel<big>"
    for n in NL_OUT:
        k = storage.klayout(out["out_" + n])
        half = n.startswith("f")
        assert bool(torch.isfinite(k[:NZ + 1 if half else NZ]).all()), n
        if not half:
            assert bool((k[NZ] == -7.0).all()), n
    assert float(storage.klayout(out["out_fplsl"]).min()) >= 0.0 and float(storage.klayout(out["out_fplsn"]).min()) >= 0.0
    assert float(storage.klayout(out["out_clc"])[:NZ].max()) <= 1.0
    for c0 in (0, nx - 512):                                        # the first and the last window against the oracle
        host = {k: storage.klayout(v)[:, c0:c0 + 512].cpu().numpy() for k, v in F.items()}
        want = run_oracle_nl(host, eta, dt, ext)
        for n in NL_OUT:
            nlev = NZ + 1 if n.startswith("f") else NZ
            got = storage.klayout(out["out_" + n])[:nlev, c0:c0 + 512].cpu().numpy()
            assert_close(f"4M columns, window {c0}: out_{n}", got, want[n][:nlev])
