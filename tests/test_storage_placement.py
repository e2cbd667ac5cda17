"""`storage.FieldArena` / `storage.tune_placement`: WHERE the fields of a call sit in HBM (docs/TUNING_LOG.md 3.7).  Placement never
changes results - the GPU test holds a tuned placement bit-equal to separately allocated fields - only how the 26+ concurrent
streams of a call fall onto HBM channels and banks."""
import numpy as np
import pytest


def test_field_arena_geometry_on_the_host():
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage

    a = storage.FieldArena(100, 137, np.float64, "cpu", capacity=5)
    fields = [a.zeros() for _ in range(5)]
    assert a.free_slots == 0
    with pytest.raises(RuntimeError, match="full"):
        a.zeros()
    two_mb = 2 << 20
    for i, f in enumerate(fields):
        assert f.shape == (100, 1, 138) and storage.field_geometry(f) == (100, 138, 128)     # level pitch padded to 512 B
        assert f.data_ptr() % two_mb == (i * 2304) % 65536          # 2 MB slab start + i x 2 304 B stagger
        assert float(f.abs().sum()) == 0.0
    assert fields[1].data_ptr() - fields[0].data_ptr() == a.slab + 2304 and a.slab % two_mb == 0
    fields[0].fill_(1.0)
    assert float(fields[1].abs().sum()) == 0.0                      # slabs do not overlap
    with pytest.raises(ValueError, match="multiple of 16"):
        storage.FieldArena(8, 4, np.float64, "cpu", capacity=2, stagger=100)
    # automatic arenas are off by default and never used for host fields
    assert storage._ARENA_CAPACITY == 0 or storage._arena_for(8, 4, torch.float64, torch.device("cpu")) is None


@pytest.mark.gpu
def test_tuned_placement_is_bit_identical_and_reports_what_it_did(gpu):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_IN, NL_OUT, compile_stencil
    from helpers import externals, nl_case

    nx, nz = 2048, 137
    ext = externals()
    fields, eta, dt = nl_case(nx, seed=3)
    eta_d = torch.as_tensor(eta, device=gpu)
    sat = compile_stencil("saturation", ext)
    nl = compile_stencil("cloudsc2_nl", ext)
    com = dict(origin=(0, 0, 0), validate_args=True, exec_info=None)

    def step(F):
        sat(in_ap=F["in_ap"], in_t=F["in_t"], out_qsat=F["in_qsat"], domain=(nx, 1, nz), **com)
        nl(**F, in_eta=eta_d, dt=dt, domain=(nx, 1, nz + 1), **com)

    order = ["in_" + n for n in NL_IN] + ["out_" + n for n in NL_OUT]
    sources = {k: torch.as_tensor(v, device=gpu) for k, v in fields.items() if k != "in_qsat"}
    tuned, rep = storage.tune_placement(nx, nz, np.float64, gpu, order, sources, step, spacings=(0, 1, 2, 3),
                                        staggers=(2304, 8448), wide_spacings=(9,), wide_shifts_mb=(0, 64), budget_s=0.5)
    assert rep["candidates"] >= 4 * 2 + 1 * 2 * 2           # the spacing x stagger x shift grid + the (optional) wide family
    assert rep["tuned_ms"] <= rep["default_ms"] and rep["stagger_bytes"] in (2304, 8448)
    assert rep["extra_spacing_x2MB"] in (0, 1, 2, 3, 9)
    assert set(tuned) == set(order)
    for k, src in sources.items():
        assert torch.equal(storage.klayout(tuned[k]), src), k                       # inputs copied in
    ptrs = sorted(t.data_ptr() for t in tuned.values())
    assert all(b - a >= (nz + 1) * nx * 8 for a, b in zip(ptrs, ptrs[1:]))           # disjoint slabs
    step(tuned)
    sep = {k: storage.from_klayout(v, np.float64, gpu) for k, v in sources.items()}
    sep["in_qsat"] = storage.zeros(nx, nz, np.float64, gpu)
    sep.update({"out_" + n: storage.zeros(nx, nz, np.float64, gpu) for n in NL_OUT})
    step(sep)
    torch.cuda.synchronize()
    for n in NL_OUT:
        assert torch.equal(tuned["out_" + n], sep["out_" + n]), n
    assert torch.equal(tuned["in_qsat"], sep["in_qsat"])
    # the second stage (a second arena searched at shifts further out when the first gained too little), forced here
    # at a small size: whichever stage wins, the fields it returns hold the inputs and compute the same results
    tuned2, rep2 = storage.tune_placement(nx, nz, np.float64, gpu, order, sources, step, spacings=(0, 1), staggers=(2304,),
                                          budget_s=0.3, extend_shifts_mb=(64, 128), extend_below_gain=1.0,
                                          extend_min_span_bytes=0)
    st2 = rep2["second_stage"]
    assert st2["candidates"] >= 2 and st2["shift_MB"] in (0, 64, 128) and isinstance(st2["chosen"], bool)
    assert ("first_stage" in rep2) == st2["chosen"] and rep2["tuned_ms"] > 0
    for k, src in sources.items():
        assert torch.equal(storage.klayout(tuned2[k]), src), k
    # the stage re-times both winners AFTER placing them - the returned fields must be back to the documented
    # state all the same (inputs = sources, output-only fields zero), whichever stage won
    for k in order:
        if k not in sources:
            assert float(tuned2[k].abs().sum()) == 0.0, k
    step(tuned2)
    torch.cuda.synchronize()
    for n in NL_OUT:
        assert torch.equal(tuned2["out_" + n], sep["out_" + n]), n
    # the second stage runs only where the arena cap can reach its shifts - with a cap below the first
    # extension shift it is SKIPPED (no `second_stage` in the report, one arena), and when it runs the report says so (above)
    span = len(order) * rep2.get("slab_bytes", rep["slab_bytes"])
    tuned3, rep3 = storage.tune_placement(nx, nz, np.float64, gpu, order, sources, step, spacings=(0, 1), staggers=(2304,),
                                          budget_s=0.3, extend_shifts_mb=(64, 128), extend_below_gain=1.0,
                                          extend_min_span_bytes=0, shifts_mb=(0,), max_arena_bytes=span + (48 << 20))
    assert "second_stage" not in rep3 and "first_stage" not in rep3 and rep3["shift_MB"] == 0
    assert rep3["arena_bytes"] <= span + (48 << 20)


@pytest.mark.gpu
def test_tuner_sizes_its_arena_to_the_free_memory(gpu, monkeypatch):
    """A shared or nearly full device: the tuner asks for at most 60 % of what is free (fewer shifts, narrower spacings)
    and says so clearly when even the densest placement does not fit - bench.py then runs on plain allocations."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage

    nx, nz, order = 1024, 15, ["a", "b", "c"]
    src = {"a": torch.ones(nz + 1, nx, dtype=torch.float64, device=gpu), "b": None, "c": None}

    def launch(F):
        F["c"].copy_(F["a"])

    slab = 2 << 20                                   # one 2-MB slab per field at this size
    real = torch.cuda.mem_get_info(gpu)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (int((3 * (slab + 2 * slab) + slab) / 0.6) + 1, real[1]))
    F, rep = storage.tune_placement(nx, nz, np.float64, gpu, order, src, launch, budget_s=0.2)
    assert rep["extra_spacing_x2MB"] <= 2 and rep["shift_MB"] == 0 and rep["arena_bytes"] <= 3 * 3 * slab + slab
    assert torch.equal(storage.klayout(F["a"]), src["a"]) and float(F["c"].abs().sum()) == 0.0
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (slab, real[1]))
    with pytest.raises(RuntimeError, match="do not fit"):
        storage.tune_placement(nx, nz, np.float64, gpu, order, src, launch, budget_s=0.2)


def test_a_failing_tuner_leaves_the_callers_state_untouched(monkeypatch, capsys):
    """`tune_field_placement` re-points every DataArray at candidate placements while it tunes; when the
    objective or the tuner raises, the DataArrays must be back on their ORIGINAL storages with their ORIGINAL contents
    (candidate runs modify inout fields), the report must carry the reason and the drivers go on untuned.  CPU test: the
    tuner is replaced by one that runs the objective once on a scratch placement and then fails."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import _common
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.fields import DataArray
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.grid import I, J, K

    nx, nz = 8, 5
    mk = lambda v: DataArray(storage.logical_view(torch.full((nz + 1, nx), float(v), dtype=torch.float64)), (I, J, K), "")  # noqa: E731
    state = {"f_a": mk(1.0), "f_b": mk(2.0), "time": 0}
    diags = {"f_b": state["f_b"], "f_c": mk(3.0)}               # f_b is shared by two dicts
    before = {k: (v.data.data_ptr(), v.data.clone()) for k, v in {**state, **diags}.items() if hasattr(v, "data")}

    def objective():                                            # an inout stencil: modifies what it is pointed at
        state["f_a"].data.add_(10.0)
        diags["f_c"].data.mul_(2.0)

    def failing_tuner(nx_, nz_, dtype, device, order, sources, launch, **kw):
        scratch = {n: storage.logical_view(sources[n].clone()) for n in order}
        launch(scratch)                                         # the DataArrays now point into `scratch`
        assert state["f_a"].data.data_ptr() == scratch[order[0]].data_ptr()
        raise RuntimeError("tune_placement: 3 fields of 2097152 B do not fit the arena cap of 1048576 B")

    monkeypatch.setattr(storage, "tune_placement", failing_tuner)
    rep = _common.tune_field_placement([state, diags], objective, _any_device=True)
    assert rep["fields"] == 3 and "do not fit" in rep["error"]
    for k, v in {**state, **diags}.items():
        if hasattr(v, "data"):
            assert v.data.data_ptr() == before[k][0] and torch.equal(v.data, before[k][1]), k
    assert state["f_b"] is diags["f_b"]
    _common.report_placement(rep)
    assert "NOT tuned" in capsys.readouterr().out

    # an objective that raises by itself (a failing stencil) is handled the same way
    def boom():
        raise ValueError("cloudsc2_nl: output 'out_clc' overlaps 'in_t' in memory")

    def plain_tuner(nx_, nz_, dtype, device, order, sources, launch, **kw):
        launch({n: storage.logical_view(sources[n].clone()) for n in order})

    monkeypatch.setattr(storage, "tune_placement", plain_tuner)
    rep = _common.tune_field_placement([state, diags], boom, _any_device=True)
    assert "overlaps" in rep["error"]
    assert all(v.data.data_ptr() == before[k][0] for k, v in {**state, **diags}.items() if hasattr(v, "data"))
