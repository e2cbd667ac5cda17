"""What `bench.py` (repo root) is built from, apart from its command line, the CPU-baseline leg (the only code that may import
`oracle/`, so it stays in bench.py), the dry run and the headline `main()`: constants of the roofline accounting (SURVEY.md
8a / 8d), the record skeleton, the PMC-traffic lookup, the resident synthetic state, the end-of-run protocol of an N-rank job
and the BASELINE configs[2] / configs[3] bench (`--config 3 / 4`).  Nothing here imports torch at module level: the plain
`python bench.py --gpus N` parent must be able to import it and start its ranks without touching the GPU."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NL_WORDS_PER_COL = 3567          # 15*137 + 138 read, 6*137 + 4*138 written
TLAD_WORDS_PER_COL = 7134        # rows a3 / a5: twice the NL count (state + perturbation / adjoint fields)
SAT_WORDS_PER_COL = 411          # 2 in, 1 out over 137 levels
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
CONFIG5_COLUMNS = 4194304        # BASELINE.json configs[4]
METRIC = "columns/sec at 137 levels fp64; achieved HBM GB/s vs MI355X roofline"
METRIC_C3 = ("columns/sec through one TL Taylor-test run (run_taylor_test.py) at 137 levels fp64 (BASELINE configs[2]); "
             "achieved HBM GB/s of the run's stencil sequence vs MI355X roofline")
METRIC_C4 = ("columns/sec through one AD symmetry-test call (run_symmetry_test.py) at 137 levels fp64 (BASELINE configs[3]); "
             "achieved HBM GB/s of the call's stencil sequence vs MI355X roofline")
INC_WORDS_PER_COL = 4416         # state_increment: 16 in + 16 out over 138 levels
PERT_WORDS_PER_COL = 6624        # perturbed_state: 32 in + 16 out over 138 levels
METRIC_C5 = ("columns/sec at 137 levels fp32, 4 194 304 columns sharded over the GPUs (BASELINE configs[4]); "
             "achieved HBM GB/s vs MI355X roofline")


class StdoutToStderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; the bench contract is
    ONE JSON line on stdout, so fd 1 is pointed at stderr while the communicator comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def gather_rank_reports(dist, world, report):
    """every rank's own outcome (placement chosen, start-up times, kernel time) in rank order: a rank that fell back to plain
    allocations or skipped the tuner must be readable in the record, not guessed from a straggler in `per_rank_ms`"""
    if dist is None:
        return [report]
    out = [None] * world
    dist.all_gather_object(out, report)
    return out


def rank0_then_everyone(dist, rank, work, key="bench/rank0_done", timeout_s=900):
    """`work()` on rank 0 while the other ranks WAIT without spinning (blocked on the rendezvous store's socket, not in a
    collective: an RCCL barrier would burn a host core per waiting rank and disturb the CPU baseline being timed)."""
    if dist is None:
        return work()
    import datetime

    from torch.distributed import distributed_c10d as c10d

    store = c10d._get_default_store()
    res = None
    if rank == 0:
        try:
            res = work()
        finally:
            store.set(key, "1")       # also when work() raised: the other ranks must not sit out the timeout
    else:
        store.wait([key], datetime.timedelta(seconds=timeout_s))
    dist.barrier()      # everyone is here within milliseconds; the store's host may now go away
    return res


# ------------------------------------------------------------------------------------------------ PMC traffic
_PMC_FILES = (
    # (columns, precision) -> summaries of the separate FETCH_SIZE / WRITE_SIZE passes, newest round first
    ((65536, "double"), ("profiles/r04/nl_fp64_65536_pmc.json", "profiles/r04/all_kernels_fp64_65536_pmc.json",
                         "profiles/r03/nl_fp64_65536_pmc.json", "profiles/r03/all_kernels_fp64_65536_pmc.json",
                         "profiles/r02/nl_fp64_65536_pmc.json", "profiles/r02/all_kernels_fp64_65536_pmc.json",
                         "profiles/r01/nl_fp64_65536_pmc.json", "profiles/r01/all_kernels_fp64_65536_pmc.json")),
    ((524288, "single"), ("profiles/r04/all_kernels_fp32_524288_pmc.json", "profiles/r03/all_kernels_fp32_524288_pmc.json",
                          "profiles/r02/all_kernels_fp32_524288_pmc.json")),
)


def pmc_traffic(kernel_substr: str, nx: int, precision: str):
    """HBM bytes per launch of the kernel whose name contains `kernel_substr`, from the rocprofv3 PMC passes committed
    under profiles/ (separate FETCH_SIZE / WRITE_SIZE passes, profiles/run_rocprof*.sh).  gfx950 correction
    (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-B requests of a wide streaming read at 64 B -> doubled;
    WRITE_SIZE is exact; both are in KiB.  (None, None) when no committed summary matches this workload."""
    for key, files in _PMC_FILES:
        if key != (nx, precision):
            continue
        for rel in files:
            try:
                with open(os.path.join(ROOT, rel)) as fh:
                    pm = json.load(fh)
                k = [v for n, v in pm.items() if kernel_substr in n][0]
                fetch = k["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0
                write = k["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
            except (OSError, KeyError, IndexError, ValueError):
                continue
            return 2.0 * fetch + write, (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {rel}: "
                                         f"2 x {fetch / 1e9:.3f} GB read + {write / 1e9:.3f} GB written per launch")
    return None, None


def roofline_entry(kernel: str, words_per_col: int, wsize: int, nx: int, precision: str, ms: float, **extra):
    nbytes = words_per_col * wsize * nx
    achieved = nbytes / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic(kernel.split("::")[-1], nx, precision)
    d = {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
         "traffic_source": src, "bytes_per_launch": nbytes, "avg_launch_ms": ms,
         "kernel_columns_per_s": nx / (ms * 1e-3), "columns": nx, "dtype": "f64" if wsize == 8 else "f32"}
    d.update(extra)
    return d


# ------------------------------------------------------------------------------------------------ state
def make_resident_state(total, nz, col0, nx, np_dtype, device, chunk=262144):
    """This rank's slice [col0, col0 + nx) of the global `total`-column synthetic problem, generated on the device
    in column chunks (the generator works in float64 temporaries: bounded at ~0.3 GB each, whatever nx is)."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import make_state

    if nx <= chunk:
        return make_state(total, nz, col0=col0, ncols=nx, dtype=np_dtype, device=device)
    out = None
    for c in range(0, nx, chunk):
        n = min(chunk, nx - c)
        part = make_state(total, nz, col0=col0 + c, ncols=n, dtype=np_dtype, device=device)
        if out is None:
            out = {k: torch.empty((nz + 1, nx), dtype=v.dtype, device=device) for k, v in part.items()}
        for k, v in part.items():
            out[k][:, c:c + n] = v
        del part
    return out


def base_record(args, world, nx, nz, value, ms_per_step, ranks, backend):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import DEFAULT_TIMESTEP_S

    c5 = args.config == 5
    total = nx * world
    return {
        "metric": METRIC_C5 if c5 else METRIC,
        "value": value,
        "unit": "columns/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if c5 else "weak",
        "vs_baseline": None,
        "dtype": "f64" if args.precision == "double" else "f32",
        "data": "synthetic columns + synthetic-parameters (reference data/input.h5 unavailable)",
        "config": {
            "workload": (f"BASELINE configs[4]: CLOUDSC2-NL (saturation + cloudsc2_nl) fp32, {total} cols x {nz} lev "
                         f"sharded across {world} GPU(s), {nx} columns per GPU" if c5 else
                         f"BASELINE configs[1]: CLOUDSC2-NL (saturation + cloudsc2_nl), {nx} cols x {nz} lev per GPU, "
                         f"{args.precision}, {world} GPU(s), {total} columns total"),
            "columns_per_gpu": nx, "columns_total": total, "levels": nz, "timestep_s": DEFAULT_TIMESTEP_S,
            "parallelism": f"column-sharded x{world}, no data-path collective",
        },
        "rccl_ranks": ranks,
        "collective_backend": backend,
    }


# ------------------------------------------------------------------------------------------------ configs 3 / 4
def harness_bench(args, rank, local_rank, world):
    """BASELINE configs[2] / configs[3]: the TL Taylor test / the AD symmetry test as the timed step, through the
    harness classes that mirror the reference's (`harness.TaylorTest`, `harness.SymmetryTest`) on the state the driver
    mirror builds (`drivers._common.setup`: the reader path by default).  Same protocol as the headline: pre-warm >= 25 ms,
    W warm-up steps, EXACTLY K steps between barrier + synchronize pairs, MAX over ranks."""
    import argparse
    import gc

    import numpy as np
    import torch

    import __graft_entry__ as ge

    ge.build()
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.drivers import _common
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.harness import SymmetryTest, TaylorTest, taylor_verdict
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import finalize_exec_info

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.collective == "gloo":
        raise SystemExit("--config 3 / 4: the harnesses all-reduce device tensors; the gloo rehearsal covers configs 2 / 5")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist

        with StdoutToStderr():
            dist.init_process_group("nccl", device_id=device)
            dist.barrier()
            torch.cuda.synchronize()

    taylor = args.config == 3
    nx, nz = args.cols, args.nlev
    wsize = 8 if args.precision == "double" else 4
    ns = argparse.Namespace(backend="hip", enable_checks=False, enable_validation=True, num_cols=nx, num_runs=1,
                            precision=args.precision, host_alias=None, output_csv_file=None, output_csv_file_stencils=None,
                            input=args.input, atol=None, rtol=None)
    with StdoutToStderr():          # the set-up's notices must not land on stdout (ONE JSON line)
        ctx = _common.setup(ns)
    if ctx["nz"] != nz:
        raise SystemExit(f"the input has {ctx['nz']} levels, --nlev says {nz}")
    cfg, grid, state, dt, p = ctx["config"], ctx["grid"], ctx["state"], ctx["dt"], ctx["params"]
    gcfg = cfg.gt4py_config
    f2s = tuple(10 ** -(i + 1) for i in range(10))                     # run_taylor_test.py:76

    def make(**kw):
        common = dict(yoethf_params=p["yoethf"], yomcst_params=p["yomcst"], yrecldp_params=p["yrecldp"],
                      yrephli_params=p["yrephli"], yrncl_params=p["yrncl"], yrphnc_params=p["yrphnc"],
                      enable_checks=False, gt4py_config=gcfg, **kw)
        if taylor:
            return TaylorTest(grid, factor1=0.01, factor2s=f2s, kflag=1, lphylin=True, ldrain1d=False, **common)
        return SymmetryTest(grid, factor=0.01, kflag=1, lphylin=True, ldrain1d=False, **common)

    sat_b, nl_b, tl_b = SAT_WORDS_PER_COL, NL_WORDS_PER_COL, TLAD_WORDS_PER_COL
    pnl_words = 2 * 2193 + 1374                    # perturbed NL run fused: 32 fields read, 10 references read (or 10 written)
    if taylor:
        seq_words = {"plain": sat_b + nl_b + INC_WORDS_PER_COL + tl_b + 10 * (PERT_WORDS_PER_COL + nl_b),     # 117 438
                     "fused": sat_b + nl_b + INC_WORDS_PER_COL + tl_b + 10 * pnl_words,
                     # fused_all: state_increment is fused into cloudsc2_tl (16 + 20 fields) and into the two multi-step launches
                     # (16 state + 10 reference fields read each)
                     "fused_all": sat_b + nl_b + (2193 + 2 * 1374) + 2 * (2193 + 1374)}
        # fused: perturbation in the NL loads + the sums in the NL epilogue, one launch per step size (r04 default of --fused);
        # fused_stored: the r03 meaning of "fused" (perturbed outputs stored, sums as separate launches)
        variants = [("graph", dict(graph=True), "plain"), ("fused", dict(fused=True), "fused"),
                    ("fused_graph", dict(fused=True, graph=True), "fused"),
                    ("fused_stored_graph", dict(fused=True, store_perturbed=True, graph=True), "fused"),
                    ("fused_all", dict(fused_all=True), "fused_all"),
                    ("fused_all_graph", dict(fused_all=True, graph=True), "fused_all")]
        what = ("saturation + cloudsc2_nl + state_increment + cloudsc2_tl + 10 x (perturbed_state + cloudsc2_nl) + the "
                "norms' reductions (tangent_linear/validation.py:150-181)")
    else:
        seq_words = {"plain": sat_b + INC_WORDS_PER_COL + tl_b + tl_b,                                         # 19 095
                     # fused: state_increment inside cloudsc2_tl (16 fields read, 20 written) + cloudsc2_ad from the TL call's
                     # trajectory (16 inputs + 10 forcings + 2 fluxes read, 16 adjoints written = 44 words per level)
                     "fused": sat_b + (2193 + 2 * 1374) + (2193 + 1374 + 2 * 138 + 16 * 137 + 2)}
        variants = [("graph", dict(graph=True), "plain"), ("fused", dict(fused=True), "fused"),
                    ("fused_graph", dict(fused=True, graph=True), "fused")]
        what = "saturation + state_increment + cloudsc2_tl + cloudsc2_ad (adjoint/validation.py:135-151, validation off)"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_window(h, steps, warmup):
        """(seconds for `steps` steps, verdict info) of harness `h`; the first call allocates and - Taylor - yields the norms"""
        if taylor:
            norms = h.run(state, dt)
            ok, verdict = taylor_verdict(norms)
            info = {"norms": [float(x) for x in norms], "verdict": verdict, "passed": bool(ok)}
            step = lambda: h.run(state, dt)  # noqa: E731
        else:
            with StdoutToStderr():
                ok = h(state, dt, enable_validation=True)
            info = {"verdict": "The symmetry test passed. HOORAY!" if ok else "The symmetry test failed.",
                    "passed": bool(ok), **(h.last or {})}
            step = lambda: h(state, dt, enable_validation=False)  # noqa: E731
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - t1, 1e-4)
        for _ in range(max(0, int(0.025 / one) + 1 - warmup)):      # >= 25 ms of the same work right before the warm-up
            step()
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        return el, info

    gc.collect()
    gc.disable()
    head = make()
    elapsed, info = run_window(head, args.steps, args.warmup)
    per_rank_ms = [1e3 * elapsed / args.steps]
    if dist is not None:
        t = torch.zeros(world, dtype=torch.float64, device=device)
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [1e3 * float(x) / args.steps for x in t.cpu()]
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gc.enable()

    # per-stencil device times of ONE instrumented step (HIP events around every stencil call)
    gcfg.exec_info = {}
    if taylor:
        head.run(state, dt)
    else:
        head(state, dt, enable_validation=False)
    finalize_exec_info(gcfg.exec_info)
    kernels = {k: {"ncalls": v["ncalls"], "device_ms": 1e3 * v.get("total_run_time", 0.0)}
               for k, v in gcfg.exec_info.items() if isinstance(v, dict) and "ncalls" in v}
    gcfg.exec_info = None
    last_kernel = _lib.last_kernel()

    var_out = {}
    if not args.no_variants and world == 1:
        for name, kw, seq in variants:
            try:
                h = make(**kw)
                el, vinfo = run_window(h, args.steps, args.warmup)
                ms = 1e3 * el / args.steps
                nbytes = seq_words[seq] * wsize * nx
                var_out[name] = {"ms_per_step": ms, "value": nx * args.steps / el, "unit": "columns/s",
                                 "bytes_per_step": nbytes, "achieved_GBs": nbytes / (ms * 1e-3) / 1e9,
                                 "frac_of_8TBs": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, **vinfo}
                del h
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001 - a variant must not cost the headline line
                var_out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                torch.cuda.empty_cache()

    if rank == 0:
        total = nx * world
        ms = 1e3 * elapsed / args.steps
        nbytes = seq_words["plain"] * wsize * nx
        res = base_record(args, world, nx, nz, value=total * args.steps / elapsed, ms_per_step=ms,
                          ranks=dist.get_world_size() if dist is not None else None,
                          backend="nccl (RCCL)" if dist is not None else "none (single process)")
        res["metric"] = METRIC_C3 if taylor else METRIC_C4
        res["data"] = ctx["source"]
        res["config"]["workload"] = (
            f"BASELINE configs[{2 if taylor else 3}]: CLOUDSC2-{'TL Taylor test' if taylor else 'AD symmetry test'} "
            f"({'run_taylor_test.py' if taylor else 'run_symmetry_test.py'}), {nx} cols x {nz} lev per GPU, {args.precision}, "
            f"{world} GPU(s); one step = {what}")
        res["config"]["input"] = args.input
        res["per_rank_ms"] = per_rank_ms
        res["per_rank_ms_min_max"] = [min(per_rank_ms), max(per_rank_ms)]
        res["verdict"] = info
        ach = nbytes / (ms * 1e-3) / 1e9
        res["roofline"] = {"kernel": "sequence: " + what, "last_kernel": last_kernel, "bound": "hbm", "achieved": ach,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "bytes_per_launch": nbytes, "bytes_per_column": seq_words["plain"] * wsize,
                           "avg_launch_ms": ms, "columns": nx, "dtype": "f64" if wsize == 8 else "f32",
                           "what": "algorithmic bytes of the step's whole stencil sequence (SURVEY.md 8d per-stencil "
                                   "figures; the norms' reductions are not counted) / the step's wall time, host side included",
                           "stencils_one_step": kernels,
                           "device_ms_one_step": sum(k["device_ms"] for k in kernels.values())}
        res["variants"] = var_out
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()
