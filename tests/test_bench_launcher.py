"""`python bench.py --gpus N` invoked PLAINLY must produce the N-rank line: the parent starts
`python -m torch.distributed.run` as a child before it touches torch or the GPU and relays rank 0's JSON line.
Rehearsed here without a GPU through `--dry-run` (gloo, shard bookkeeping and reductions only, no kernels)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


DRY = ("--dry-run", "--cpu-cols", "256", "--cpu-budget-s", "0.2")     # the CPU-baseline leg runs, on a sample of a fraction of a second


def _nrank_record_is_complete(d, world, real_kernels):
    """an N > 1 line carries everything the N = 1 line does - `cpu_baseline` (rank 0, after the closing
    barrier), `roofline`, every rank's own placement outcome, and which collective backend joined how many ranks."""
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["rank"] == 0 and cb["world_size"] == world
    assert cb["affinity_cores"] >= cb["cores"] and "cgroup_cpu_quota_cores" in cb and cb["numpy_1core"]["cores"] == 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["bytes_per_launch"] > 0
    reps = d["per_rank_placement"]
    assert [x["rank"] for x in reps] == list(range(world))
    for x in reps:
        assert "mode" in x and "chosen" in x and "startup_s" in x and x["ms_per_step"] > 0, x
    assert len(d["per_rank_ms"]) == world
    if real_kernels:
        assert r["frac"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel"].startswith("cs2::")
        assert cb["parity_check"]["passed"] is True
        assert all(x["nl_kernel_ms"] > 0 and x["startup_s"]["to_first_step_s"] > 0 for x in reps)
        assert d["startup_s_max_over_ranks"] >= max(x["startup_s"]["to_first_step_s"] for x in reps) - 1e-9
    assert (d["rccl_ranks"] == world) != (d.get("rehearsal_ranks") == world)        # exactly one of the two says N


def _run(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=600, env=e)
    return p


def test_plain_invocation_with_two_gpus_starts_two_ranks():
    p = _run("--gpus", "2", "--steps", "3", "--warmup", "1", *DRY)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                       # ONE JSON line on stdout, everything else on stderr
    d = json.loads(lines[0])
    _nrank_record_is_complete(d, 2, real_kernels=False)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["dtype"] == "f64" and d["dry_run"] is True and d["value"] is None
    assert d["config"]["columns_per_gpu"] == 65536 and d["config"]["columns_total"] == 131072
    # the two ranks own different slices of ONE global problem (col0 = 0 and 65536) and the same eta
    assert d["shard_check"]["col0_sum"] == 65536.0
    assert d["shard_check"]["eta_sum_x_world"] == pytest.approx(2 * d["shard_check"]["eta_sum"], rel=1e-12)
    assert "torch.distributed.run" in p.stderr             # the launcher announced the child it started


def test_config5_is_the_fixed_fp32_problem_split_over_the_gpus():
    p = _run("--gpus", "2", "--config", "5", *DRY)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["scaling"] == "strong" and d["dtype"] == "f32" and d["n_gpus"] == 2
    assert d["config"]["columns_per_gpu"] == 4194304 // 2 and d["config"]["columns_total"] == 4194304
    assert "configs[4]" in d["config"]["workload"] and "fp32" in d["metric"]
    q = _run("--gpus", "1", "--config", "5", *DRY)
    assert q.returncode == 0, q.stderr[-3000:]
    d1 = json.loads(q.stdout.strip().splitlines()[-1])
    assert d1["config"]["columns_per_gpu"] == 4194304 and d1["n_gpus"] == 1 and d1["rccl_ranks"] is None


def test_eight_ranks_config5_rehearsal():
    """The shape of the driver's 8-GPU scaling run of BASELINE configs[4], rehearsed on gloo: 8 ranks x 524 288 columns."""
    p = _run("--gpus", "8", "--config", "5", *DRY)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _nrank_record_is_complete(d, 8, real_kernels=False)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["config"]["columns_per_gpu"] == 524288
    assert d["shard_check"]["col0_sum"] == 524288.0 * sum(range(8))          # ranks own columns [r * 524288, (r+1) * 524288)


def test_eight_ranks_default_mode_rehearsal_carries_per_rank_times():
    """The default (weak, fp64) mode at the driver's N = 8, rehearsed on gloo without kernels: 8 x 65 536 columns, and the
    record carries every rank's own time (a straggler must be visible in the first real SCALE line)."""
    p = _run("--gpus", "8", *DRY)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    _nrank_record_is_complete(d, 8, real_kernels=False)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["config"]["columns_per_gpu"] == 65536 and d["config"]["columns_total"] == 8 * 65536
    assert d["per_rank_ms"] == pytest.approx([1.0 * (r + 1) for r in range(8)]) and d["per_rank_ms_min_max"] == pytest.approx([1.0, 8.0])


def test_every_rank_sizes_its_placement_arena_against_its_own_device(monkeypatch):
    """under WORLD_SIZE = 8 a rank's `tune_placement` arena must be bounded by what ITS device has free
    (60 %), whatever the other ranks do.  (a) the planning function never exceeds the bound, down to "does not fit";
    (b) `tune_placement` asks `torch.cuda.mem_get_info` about the device it was given - bench.py hands it cuda:LOCAL_RANK."""
    import numpy as np
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage

    slab = -(-(138 * 65536 * 8 + 65536) // (2 << 20)) * (2 << 20)
    kw = dict(spacings=tuple(range(64)), staggers=(2304, 8448), shifts_mb=(0, 4096, 8192, 12288), wide_spacings=(),
              wide_shifts_mb=(), max_arena_bytes=40 << 30, max_shift_spans=4.0)
    for free in (288e9, 100e9, 30e9, 8e9, 4e9):
        grid, need, _ = storage.plan_placement_grid(26, slab, int(free), **kw)
        assert need <= 0.6 * free and need <= 40 << 30 and len(grid) >= 2, (free, need)
        assert need >= 26 * slab                                   # and it does hold the 26 fields
    with pytest.raises(RuntimeError, match="do not fit the arena cap"):
        storage.plan_placement_grid(26, slab, int(2e9), **kw)
    asked = []

    def fake_mem_get_info(dev):
        asked.append(torch.device(dev))
        return (int(2e9), int(288e9))                              # almost nothing free on that device

    monkeypatch.setattr(torch.cuda, "mem_get_info", fake_mem_get_info)
    with pytest.raises(RuntimeError, match="do not fit the arena cap"):     # refused BEFORE anything is allocated
        storage.tune_placement(65536, 137, np.float64, torch.device("cuda", 5), ["f%d" % i for i in range(26)], {}, None)
    assert asked == [torch.device("cuda", 5)]
    src = open(BENCH).read()
    assert "torch.cuda.set_device(local_rank)" in src and 'device = torch.device("cuda", local_rank)' in src


def test_concurrent_builds_of_eight_ranks_run_make_once(tmp_path):
    """the N ranks of a multi-GPU bench all call `__graft_entry__.build()` at start-up; the file lock
    must leave exactly ONE `make` of the HIP library when it is stale.  Rehearsed with 8 processes whose `make` and ISA
    check are stand-ins that only log (a real rebuild takes minutes); the library's source hash is made stale for the
    test and is valid again afterwards."""
    import textwrap

    lib = os.path.join(ROOT, "gt4py_dwarf_p_cloudsc2_tl_ad_amd", "libcloudsc2_hip.so")
    if not os.path.exists(lib) or not os.path.exists(lib + ".srchash"):
        pytest.skip("the library has not been built yet")
    log = tmp_path / "make.log"
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(f"""
        import os, sys, time, types
        sys.path.insert(0, {ROOT!r})
        import __graft_entry__ as ge
        real_run = ge.subprocess.run
        def fake_run(cmd, *a, **kw):
            if cmd[0] == "make" and "csrc" in cmd[2]:
                with open({str(log)!r}, "a") as fh:
                    fh.write("make %d\\n" % os.getpid())
                time.sleep(0.5)                      # a build in progress: the other ranks are waiting for the lock
                return types.SimpleNamespace(returncode=0)
            return real_run(cmd, *a, **kw)
        ge.subprocess.run = fake_run
        fake = types.ModuleType("check_ring_isa")
        fake.check_all = lambda *a, **k: {{"stand-in": True}}
        sys.modules["check_ring_isa"] = fake
        ge.build()
        """))
    saved = open(lib + ".srchash").read()
    try:
        with open(lib + ".srchash", "w") as fh:
            fh.write("stale\n")
        procs = [subprocess.Popen([sys.executable, str(child)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for _ in range(8)]
        outs = [p.communicate(timeout=600) for p in procs]
        assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
        assert log.read_text().count("make") == 1, log.read_text()
        assert open(lib + ".srchash").read() == saved        # marked as built from the (unchanged) sources again
    finally:
        with open(lib + ".srchash", "w") as fh:
            fh.write(saved)


def test_world_size_mismatch_and_bad_splits_are_refused():
    p = _run("--gpus", "2", *DRY, env={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)
    p = _run("--gpus", "3", "--config", "5", *DRY)
    assert p.returncode != 0 and "do not split" in p.stderr


def test_launcher_parent_never_imports_torch():
    """The parent of a plain `--gpus N` run must not hold a HIP context when it starts the ranks: it imports nothing
    beyond the standard library before `launch_ranks` (checked on the module source: no torch import at module level,
    and main() branches to the launcher before its first torch import)."""
    src = open(BENCH).read()
    head = src[:src.index("def parse_args")]
    assert "import torch" not in head
    main_src = src[src.index("def main("):]
    assert main_src.index("launch_ranks(args, argv)") < main_src.index("import torch")


@pytest.mark.gpu
def test_bench_line_carries_the_contract_on_the_gpu(gpu):
    """`python bench.py` (small sizes for speed) as a child process on the GPU: ONE JSON line with the driver's contract
    keys, the `roofline` / `roofline_tl` / `roofline_ad` / `roofline_nl_f32` objects and a `cpu_baseline`."""
    p = _run("--steps", "3", "--warmup", "1", "--cpu-cols", "256")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["unit"] == "columns/s" and d["value"] > 1e7 and d["outputs_finite"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "extra_rooflines_error" not in d and "extra_rooflines_f32_error" not in d, d.get("extra_rooflines_f32_error")
    for k in ("roofline", "roofline_tl", "roofline_ad", "roofline_nl_f32", "roofline_tl_f32", "roofline_ad_f32"):
        r = d[k]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and 0.2 < r["frac"] < 1.0, (k, r)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel"].startswith("cs2::")
    assert d["roofline"]["kernel"] == "cs2::nl_ring_kernel" and d["roofline_tl"]["kernel"] == "cs2::tl_kernel"
    # run_taylor_test.py / run_symmetry_test.py `--precision single` at the per-GPU shard of BASELINE configs[4]
    for k, kern in (("roofline_tl_f32", "cs2::tl_ring_kernel"), ("roofline_ad_f32", "cs2::ad_kernel")):
        assert d[k]["kernel"] == kern and d[k]["dtype"] == "f32" and d[k]["columns"] == 524288, d[k]
        assert d[k]["bytes_per_launch"] == 7134 * 4 * 524288
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    pc = d["cpu_baseline"]["parity_check"]       # the HIP step on the baseline's own columns, held to the C restatement
    assert pc["passed"] and pc["points_outside_tolerance"] == 0 and pc["columns"] == 256 and pc["fields"] == 11
    assert 0 <= pc["max_err_over_field_scale"] < 1e-10
    assert d["fused_step"]["results_equal_unfused"] is True
    # the tuner's winner was held against plain allocations before the timed region, in the headline and in every leg, and
    # the record says which one was timed; the untuned figure stands beside `value`
    for rep in (d["placement"], d["roofline_tl"]["placement"], d["roofline_ad"]["placement"], d["roofline_nl_f32"]["placement"],
                d["roofline_tl_f32"]["placement"], d["roofline_ad_f32"]["placement"]):
        assert rep["chosen"].startswith(("tuned arena", "plain allocations")), rep
        assert rep["recheck_tuned_ms"] > 0 and rep["recheck_plain_ms"] > 0
        assert (rep["mode"] == "separate") == rep["chosen"].startswith("plain")
    assert d["value_default_placement"] > 1e7


@pytest.mark.gpu
def test_two_ranks_with_real_kernels_reproduce_the_one_process_result(gpu):
    """`--collective gloo`: the N-rank path with REAL kernels on fewer GPUs than ranks (both ranks on this box's one GPU; barrier
    and reductions through gloo).  Two shards of 8 192 columns must give the validation norms of ONE process on the same
    16 384 global columns - the shards are slices of one problem, eta comes from global column 0, the SUM all-reduce adds up."""
    common = ("--steps", "3", "--warmup", "1", "--no-extra-rooflines", "--placement", "separate")
    one = _run("--cols", "16384", "--cpu-cols", "0", "--no-roofline-events", *common)
    assert one.returncode == 0, one.stderr[-3000:]
    two = _run("--gpus", "2", "--collective", "gloo", "--cols", "8192", "--cpu-cols", "256", "--cpu-budget-s", "0.5", *common)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, two.stdout[-2000:]
    d1, d2 = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(lines[0])
    _nrank_record_is_complete(d2, 2, real_kernels=True)      # cpu_baseline (+ parity check), roofline, per-rank outcomes at N = 2
    assert "[bench] rank 1: import" in two.stderr            # every rank prints its start-up times
    assert d2["n_gpus"] == 2 and d2["rehearsal_ranks"] == 2 and d2["rccl_ranks"] is None
    assert d2["collective_backend"].startswith("gloo (rehearsal")
    assert d2["config"]["columns_per_gpu"] == 8192 and d2["config"]["columns_total"] == 16384 == d1["config"]["columns_total"]
    assert d2["outputs_finite"] is True and d2["value"] > 0
    for name, v1 in d1["validation_norm"].items():
        assert d2["validation_norm"][name] == pytest.approx(v1, rel=1e-12, abs=1e-300), name


@pytest.mark.gpu
@pytest.mark.parametrize("config", [3, 4])
def test_bench_lines_of_configs_3_and_4(gpu, config):
    """`python bench.py --config 3 | 4`: BASELINE configs[2] / configs[3] as timed steps with the same JSON contract; the
    record carries the reference's verdict string, a sequence-level `roofline` (sum of the algorithmic bytes of the step's
    stencils / the step's time) and the opt-in variants, each with its own verdict."""
    p = _run("--config", str(config), "--steps", "3", "--warmup", "1", "--cols", "8192")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "verdict", "variants", "per_rank_ms"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["value"] > 0 and d["vs_baseline"] is None
    assert f"configs[{config - 1}]" in d["config"]["workload"] and d["config"]["columns_per_gpu"] == 8192
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["bytes_per_column"] == (939504 if config == 3 else 152760)             # SURVEY 8d, VERDICT r02 item 3
    assert d["verdict"]["passed"] is True
    if config == 3:
        assert d["verdict"]["verdict"].startswith("The test passed with penalty") and len(d["verdict"]["norms"]) == 10
        assert r["stencils_one_step"]["cloudsc2_nl"]["ncalls"] == 11 and r["stencils_one_step"]["perturbed_state"]["ncalls"] == 10
        for name in ("graph", "fused", "fused_graph", "fused_stored_graph", "fused_all", "fused_all_graph"):
            v = d["variants"][name]
            assert "error" not in v and v["verdict"] == d["verdict"]["verdict"], (name, v)
    else:
        assert d["verdict"]["verdict"] == "The symmetry test passed. HOORAY!" and d["verdict"]["max_error_eps"] < 100
        assert set(r["stencils_one_step"]) == {"saturation", "state_increment", "cloudsc2_tl", "cloudsc2_ad"}
        for name in ("graph", "fused", "fused_graph"):
            assert "error" not in d["variants"][name] and d["variants"][name]["passed"] is True, (name, d["variants"][name])


@pytest.mark.gpu
def test_config5_whole_problem_on_one_gpu(gpu):
    """BASELINE configs[4] at its FULL size - 4 194 304 fp32 columns x 137 levels, 60 GB of fields - resident on the one GPU
    of the box (`--config 5 --gpus 1`; the 8-GPU split of the same global problem needs hardware the pool does not have):
    the record names the configuration, every output is finite, and the NL kernel is the fp32 ring at a sane rate."""
    p = _run("--config", "5", "--steps", "3", "--warmup", "1", "--cpu-cols", "0", "--no-extra-rooflines",
             "--placement", "separate")
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["scaling"] == "strong"
    assert d["config"]["columns_total"] == 4194304 == d["config"]["columns_per_gpu"] and "configs[4]" in d["config"]["workload"]
    assert d["outputs_finite"] is True and d["value"] > 1.5e8
    r = d["roofline"]
    assert r["kernel"] == "cs2::nl_ring_kernel" and r["dtype"] == "f32" and r["columns"] == 4194304 and 0.4 < r["frac"] < 1.0
    assert r["bytes_per_launch"] == 14268 * 4194304
    # the same global problem split over 4 ranks (real kernels, the ranks share this GPU, gloo for the reductions): the
    # shards are slices of ONE problem, so the validation norms of the split run are those of the whole
    q = _run("--config", "5", "--gpus", "4", "--collective", "gloo", "--steps", "3", "--warmup", "1", "--cpu-cols", "0",
             "--no-extra-rooflines", "--no-roofline-events", "--placement", "separate")
    assert q.returncode == 0, q.stderr[-3000:]
    d4 = json.loads([l for l in q.stdout.splitlines() if l.strip()][0])
    assert d4["n_gpus"] == 4 and d4["rehearsal_ranks"] == 4 and d4["config"]["columns_per_gpu"] == 1048576
    assert d4["config"]["columns_total"] == 4194304 and len(d4["per_rank_ms"]) == 4 and d4["outputs_finite"] is True
    for name, v1 in d["validation_norm"].items():
        assert d4["validation_norm"][name] == pytest.approx(v1, rel=1e-9, abs=1e-300), name


@pytest.mark.gpu
def test_a_rank_that_started_too_slowly_skips_the_tuner_and_says_so(gpu):
    """start-up is bounded - a rank that needed longer than `--startup-budget-s` to reach the placement
    tuner (a cold node paging in torch, eight ranks queueing for the build lock) runs on plain allocations, and the record
    says so for that rank; here the budget is 0 s, so the one rank must skip."""
    p = _run("--steps", "3", "--warmup", "1", "--cols", "8192", "--cpu-cols", "0", "--no-extra-rooflines",
             "--startup-budget-s", "0")
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert d["placement"]["mode"] == "separate" and "--startup-budget-s 0" in d["placement"]["tune_skipped"]
    r0 = d["per_rank_placement"][0]
    assert r0["mode"] == "separate" and "skipped" in r0["chosen"] and r0["startup_s"]["tune_s"] == 0.0
    assert d["value"] > 0 and d["value_default_placement"] == d["value"] and d["default_placement_is_value"] is True
    assert "[bench] rank 0: import" in p.stderr


@pytest.mark.gpu
def test_one_rank_under_torchrun_takes_the_rccl_path(gpu):
    """The driver starts the N-rank bench as `python -m torch.distributed.run ... bench.py --gpus N`.  With ONE rank on the
    box's one GPU that command runs the real RCCL code path end to end (two RCCL ranks cannot share a GPU): communicator
    creation with its banner kept off stdout, barrier, the per-rank time / MAX / SUM all-reduces on device tensors,
    `all_gather_object` of the rank reports, the rendezvous-store wait around rank 0's CPU baseline, the closing barrier."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "1", "--steps", "3", "--warmup", "1", "--cols", "8192",
           "--no-extra-rooflines", "--cpu-cols", "256", "--cpu-budget-s", "0.3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]               # the RCCL banner did not land on stdout
    d = json.loads(lines[0])
    assert d["collective_backend"] == "nccl (RCCL)" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    _nrank_record_is_complete(d, 1, real_kernels=True)
    assert d["value"] > 0 and d["outputs_finite"] is True
