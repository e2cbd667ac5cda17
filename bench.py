#!/usr/bin/env python3
"""Headline benchmark: CLOUDSC2-NL fp64, 65 536 columns x 137 levels per GPU (BASELINE.json configs[1]).

One "step" = the reference driver's timed region (/root/reference/drivers/run_nonlinear.py:115-119):
`saturation(state)` followed by `cloudsc2_nl(state, dt)` on the same resident state, i.e. two kernel
launches through the C ABI of libcloudsc2_hip.so.  Inputs are synthetic columns generated directly
in HBM (gt4py_dwarf_p_cloudsc2_tl_ad_amd/synthetic.py; the reference's data/input.h5 is not
available) and are resident before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]            # N > 1: starts its own N ranks (see below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W  # same thing, ranks started by the caller

Modes
  default      weak scaling: every rank owns `--cols` (65 536) fp64 columns of a global N*cols-column problem.
  --config 5   BASELINE.json configs[4]: ONE global problem of 4 194 304 fp32 columns split N ways
               (`cols = 4 194 304 / N`, "scaling": "strong", "dtype": "f32").
  --config 3   BASELINE.json configs[2]: one step = one `TaylorTest.run` (/root/reference/src/cloudsc2_gt4py/physics/
               tangent_linear/validation.py:150-181: saturation, cloudsc2_nl, state_increment, cloudsc2_tl, then ten times
               perturbed_state + cloudsc2_nl + the norm) on 65 536 fp64 columns per GPU; the record carries the reference's
               verdict string and, beside the headline (the reference's own call sequence, launched eagerly), the opt-in
               variants: HIP-graph replay, perturbation fused into the NL loads, all step sizes in two launches.
  --config 4   BASELINE.json configs[3]: one step = the timed call of run_symmetry_test.py:94-98
               (`SymmetryTest(state, dt, enable_validation=False)`: saturation, state_increment, cloudsc2_tl, cloudsc2_ad);
               the validated call that precedes it supplies the verdict.

N > 1: one process per GPU.  Columns are independent, so there is NO data-path collective; RCCL is used only for
the barrier, the max-over-ranks time and the final validation-norm all-reduce.  When `--gpus N > 1` is given to a
plain `python bench.py` (no WORLD_SIZE in the environment), this process starts `python -m torch.distributed.run`
with N ranks as a CHILD process - before torch is imported or any GPU call is made here - relays rank 0's JSON
line and exits with the child's return code.

Field placement: before anything is timed, `storage.tune_placement` measures where the 26 fields of the step should sit in
HBM for this process (same kernels, same arguments, bit-identical results; the spacing between the fields' starting
addresses decides how 26 concurrent streams fall onto HBM channels and banks, worth up to 10 %, docs/TUNING_LOG.md 3.7) and the
state is placed there; `--placement separate` gives one torch allocation per field instead.  The record says what was done
(`placement`).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel of the step (the `cloudsc2_nl` kernel; its name
is the one the launcher reports): algorithmic bytes per launch (3 567 words per column) / the kernel's
mean duration measured with HIP events on the launch stream.  At N = 1 the line also carries `roofline_tl`,
`roofline_ad` (65 536 fp64 columns), `roofline_nl_f32` (524 288 fp32 columns = the per-GPU shard of config 5 on 8
GPUs) and `cpu_baseline` (the C/OpenMP and NumPy restatements on a bounded sample).

Layout: this file holds the command line, the self-launcher, the CPU-baseline leg (the only code that may touch `oracle/`),
the dry run and the headline `main()`; the roofline constants, the record skeleton, the PMC lookup, the N-rank end-of-run
protocol and the `--config 3 / 4` bench live in `gt4py_dwarf_p_cloudsc2_tl_ad_amd/benchlib.py`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gt4py_dwarf_p_cloudsc2_tl_ad_amd.benchlib import (  # noqa: E402  (no torch import at module level there)
    CONFIG5_COLUMNS, HBM_PEAK_GBS, NL_WORDS_PER_COL, TLAD_WORDS_PER_COL, StdoutToStderr, base_record, gather_rank_reports,
    harness_bench, make_resident_state, rank0_then_everyone, roofline_entry)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=2,
                    help="2 (default): BASELINE configs[1], 65 536 fp64 columns per GPU, weak scaling; "
                         "3: BASELINE configs[2], the TL Taylor test as the step; 4: BASELINE configs[3], the AD symmetry "
                         "test as the step; 5: BASELINE configs[4], 4 194 304 fp32 columns split over --gpus, strong scaling")
    ap.add_argument("--input", default="auto",
                    help="--config 3 / 4: 'auto' = the reader path (data/input.h5 if present, else its 100-column stand-in, "
                         "tiled to the columns: the configuration on which the reference's verdicts pass), 'synthetic' = "
                         "distinct mixed-regime columns, or the path of an HDF5 input file")
    ap.add_argument("--no-variants", action="store_true", help="--config 3 / 4: time the headline sequence only")
    ap.add_argument("--no-placement-recheck", action="store_true",
                    help="--placement tuned: do not hold the tuner's winner against plain allocations before the timed region")
    ap.add_argument("--tune-shifts-mb", default=None,
                    help="dev: comma-separated whole-placement shifts (MB) for the placement tuner instead of 0/4/8/12 GB, "
                         "with the arena allowed to grow to 96 GB (profiles/tuner_ab.sh)")
    ap.add_argument("--cols", type=int, default=None, help="columns per GPU (default 65536; --config 5: 4194304 / gpus)")
    ap.add_argument("--nlev", type=int, default=137)
    ap.add_argument("--precision", choices=["double", "single"], default=None)
    ap.add_argument("--cpu-cols", type=int, default=16384,
                    help="columns of the CPU-baseline sample (0 disables the baseline)")
    ap.add_argument("--cpu-budget-s", type=float, default=8.0,
                    help="seconds the C/OpenMP CPU-baseline loop runs for (the NumPy leg gets 0.75 x, the all-cores leg 0.5 x)")
    ap.add_argument("--startup-budget-s", type=float, default=120.0,
                    help="a rank that needed longer than this to get from process start to the placement tuner skips the tuner "
                         "(plain allocations; the record says so): keeps an N-rank run on a cold node inside the driver's clock")
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--no-extra-rooflines", action="store_true",
                    help="skip the TL / AD / fp32-NL kernel legs (N = 1 only)")
    ap.add_argument("--placement", choices=["tuned", "arena", "separate"], default="tuned",
                    help="where the step's 26 fields sit in HBM: 'tuned' (default) = storage.tune_placement calibrates the "
                         "spacing between field starts for this process with the step itself as the objective, before the "
                         "timed region; 'arena' = the default FieldArena layout; 'separate' = one torch allocation per field")
    ap.add_argument("--collective", choices=["rccl", "gloo"], default="rccl",
                    help="'gloo': REHEARSAL of the N-rank path on fewer GPUs than ranks - kernels run for real (rank r on "
                         "cuda:(r mod device count), so ranks may share a GPU), the barrier and the reductions go through gloo "
                         "on host tensors; the record says so and its value is not a scaling figure")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing rehearsal without a GPU: gloo instead of RCCL, shard bookkeeping and the "
                         "reductions only, NO kernels (value is null) - used by the CPU tests of the launcher")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.tune_shifts_mb is not None:
        try:
            args.tune_shifts_mb = tuple(int(x) for x in args.tune_shifts_mb.split(","))
            assert all(0 <= x <= 131072 for x in args.tune_shifts_mb) and args.tune_shifts_mb
        except (ValueError, AssertionError):
            ap.error("--tune-shifts-mb: a comma-separated list of shifts in MB (0 .. 131072)")
    if args.config == 5:
        if args.precision not in (None, "single"):
            ap.error("--config 5 is the fp32 configuration")
        args.precision = "single"
        if args.cols is None:
            if CONFIG5_COLUMNS % args.gpus:
                ap.error(f"--config 5: {CONFIG5_COLUMNS} columns do not split over {args.gpus} GPUs")
            args.cols = CONFIG5_COLUMNS // args.gpus
    else:
        args.precision = args.precision or "double"
        args.cols = 65536 if args.cols is None else args.cols
    return args


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args, argv) -> int:
    """Plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE): start the N ranks as a child torchrun job.
    Nothing here touches torch or the GPU, so the parent never holds a HIP context; the child's stdout is scanned
    for rank 0's JSON line (the only thing written to this process's stdout), everything else goes to stderr."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only mode this pool's driver supports
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    seen = 0
    for line in p.stdout:
        if line.lstrip().startswith('{"metric"'):
            seen += 1
            if seen == 1:
                print(line.rstrip("\n"), flush=True)
                continue
        sys.stderr.write(line)
    rc = p.wait()
    if rc == 0 and seen != 1:
        print(f"[bench] expected ONE JSON line from rank 0, saw {seen}", file=sys.stderr)
        return 1
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cpu_share():
    """(cores in this process's affinity mask, CPU quota of its cgroup in cores or None): what "the node's host cores" are
    for THIS process.  A container may show every core of the node in the mask and still be throttled to a quota."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            with open(path) as fh:
                q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        except (OSError, ValueError):
            pass
    return aff, quota


def cpu_baseline(cols: int, nz: int, np_dtype, budget_s: float = 8.0, hip_step=None):
    """CPU baselines on the host, saturation + cloudsc2_nl on `cols` synthetic columns (BASELINE configs[0] size):
      * headline `value`: the plain-C + OpenMP restatement (oracle/cloudsc2_nl_omp.c, SURVEY 8d "restatement B")
        on the host cores of this GPU's share, fp64, repeated for ~`budget_s` s;
      * `numpy_1core`: the NumPy restatement (GT4Py-numpy-like execution shape, single-threaded) for ~6 s.
    Both are the checker (pinned to the executed reference source by tests/), timed here only as baselines.
    `hip_step(fields, eta, dt)` (the product path on the device, given by main()) is run on the SAME columns and held to
    the C restatement's outputs with the parity tests' fp64 tolerance (tests/helpers.py: |a-b| <= 1e-9 |b| + 1e-11 max|b|);
    the outcome goes into the record as `cpu_baseline.parity_check` - the checker checking, nothing it returns is shipped."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import NL_OUT, externals, nl_case, run_oracle_nl
    from oracle import cloudsc2_c
    from oracle import cloudsc2_numpy as oracle

    ext = externals()
    run_oracle_nl(*nl_case(64, nz, np_dtype), ext)  # warm-up (imports, allocator)
    fields, eta, dt = nl_case(cols, nz, np_dtype)

    def loop(step, budget):
        step()
        runs, t0 = 0, time.perf_counter()
        while True:
            step()
            runs += 1
            el = time.perf_counter() - t0
            if el >= budget or runs >= 500:
                return runs, el

    def numpy_step():
        oracle.saturation(fields["in_ap"], fields["in_t"], fields["in_qsat"], ext)
        run_oracle_nl(fields, eta, dt, ext)

    n_runs, n_el = loop(numpy_step, 0.75 * budget_s)
    numpy_rate = cols * n_runs / n_el
    what = f"saturation + cloudsc2_nl, {cols} cols x {nz} lev"
    aff, quota = host_cpu_share()
    res = {"unit": "columns/s", "host_cores": os.cpu_count(), "affinity_cores": aff, "cgroup_cpu_quota_cores": quota,
           "kind": "port", "rank": int(os.environ.get("RANK", "0")), "world_size": int(os.environ.get("WORLD_SIZE", "1")),
           "numpy_1core": {"value": numpy_rate, "unit": "columns/s", "cores": 1,
                           "sample": f"NumPy restatement, {what} {np.dtype(np_dtype).name}, {n_runs} runs in {n_el:.1f} s"}}
    try:
        threads = max(1, min(aff, 16 if quota is None else int(quota + 0.5)))
        F = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in fields.items()}
        for n in NL_OUT:
            F["out_" + n] = np.zeros_like(F["in_ap"])

        def c_step():
            cloudsc2_c.saturation(F["in_ap"], F["in_t"], F["in_qsat"], ext, nthreads=threads)
            cloudsc2_c.cloudsc2_nl(F, eta, dt, ext, nthreads=threads)

        c_runs, c_el = loop(c_step, budget_s)
        if hip_step is not None:
            from helpers import TOL
            tol = TOL[np.dtype("float64")]
            got = hip_step(F, eta, dt)
            worst, bad = 0.0, 0
            for n in ("qsat",) + tuple(NL_OUT):
                want = F["in_qsat"] if n == "qsat" else F["out_" + n]
                scale = float(np.max(np.abs(want)))
                err = np.abs(got[n] - want)
                bad += int((err > tol["rtol"] * np.abs(want) + tol["atol_rel"] * scale).sum()) + int(np.isnan(got[n]).sum())
                worst = max(worst, float(err.max() / scale) if scale > 0 else 0.0)
            res["parity_check"] = {"columns": cols, "fields": 1 + len(NL_OUT), "points_outside_tolerance": bad,
                                   "max_err_over_field_scale": worst, "rtol": tol["rtol"], "atol_rel": tol["atol_rel"],
                                   "passed": bad == 0,
                                   "what": "HIP saturation + cloudsc2_nl against oracle/cloudsc2_nl_omp.c on the same columns"}
        # the same restatement on EVERY core of the affinity mask (north_star: "the node's host cores") - only where the mask
        # is not known to exceed the cgroup's CPU quota: one thread per core of a mask the process is throttled below
        # measures the throttle, not the cores (r03: 53 k columns/s on "256 threads" of a 16-core share)
        all_threads = aff
        if all_threads > threads and quota is None:
            def c_step_all():
                cloudsc2_c.saturation(F["in_ap"], F["in_t"], F["in_qsat"], ext, nthreads=all_threads)
                cloudsc2_c.cloudsc2_nl(F, eta, dt, ext, nthreads=all_threads)

            a_runs, a_el = loop(c_step_all, 0.5 * budget_s)
            res["all_cores"] = {"value": cols * a_runs / a_el, "unit": "columns/s", "cores": all_threads,
                                "sample": f"same restatement and columns with one thread per core of this process's affinity "
                                          f"mask ({all_threads}; no cgroup CPU quota visible), {a_runs} runs in {a_el:.1f} s"}
        elif all_threads > threads:
            res["all_cores"] = {"value": None, "cores": all_threads,
                                "sample": f"not taken: the affinity mask shows {all_threads} cores but the cgroup's CPU quota is "
                                          f"{quota:.1f} cores - `value` ({threads} threads) is what the host gives this process"}
        res.update(value=cols * c_runs / c_el, cores=threads,
                   sample=f"plain-C + OpenMP restatement (oracle/cloudsc2_nl_omp.c, {threads} threads, scalar libm, "
                          f"-O2), {what} float64, {c_runs} runs in {c_el:.1f} s, synthetic-parameters")
    except Exception as exc:  # the C library is optional test infrastructure: fall back to the NumPy figure
        res.update(value=numpy_rate, cores=1, sample=res["numpy_1core"]["sample"] + f" (C restatement unavailable: {exc})")
    return res


def dry_run(args, rank, world):
    """Launcher / rendezvous / shard bookkeeping rehearsal on the CPU (gloo): no kernels, no GPU."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        dist.init_process_group("gloo")
    nx, nz = args.cols, args.nlev
    np_dtype = np.float64 if args.precision == "double" else np.float32
    n = min(nx, 64)      # a few columns of this rank's slice: enough to show the shards differ and tile the global problem
    s = make_state(nx * world, nz, col0=rank * nx, ncols=n, dtype=np_dtype)
    eta = eta_levels(nz, dtype=np_dtype)
    t = torch.tensor([float(np.abs(s["f_t"]).sum()), float(rank * nx), float(eta.sum())], dtype=torch.float64)
    first = t.clone()
    per_rank = torch.zeros(world, dtype=torch.float64)
    per_rank[rank] = 0.001 * (rank + 1)                 # stand-in for this rank's elapsed time: rank r "took" r + 1 ms
    if dist.is_initialized():
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)  # the gather of the per-rank times (main() does the same)
        tm = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    # the same end-of-run protocol as main(): every rank's report gathered in rank order, then the CPU baseline on rank 0
    # while the other ranks wait on the rendezvous store
    mine = {"rank": rank, "device": None, "mode": "dry run", "chosen": "none (no kernels)", "ms_per_step": 1.0 * (rank + 1),
            "nl_kernel_ms": None, "startup_s": {"to_first_step_s": 0.0}}
    rank_reports = gather_rank_reports(dist if dist.is_initialized() else None, world, mine)

    def rank0_record():
        res = base_record(args, world, nx, nz, value=None, ms_per_step=None,
                          ranks=dist.get_world_size() if dist.is_initialized() else None, backend="gloo (dry run)")
        res["per_rank_ms"] = [1e3 * float(x) for x in per_rank]
        res["per_rank_ms_min_max"] = [min(res["per_rank_ms"]), max(res["per_rank_ms"])]
        res["per_rank_placement"] = rank_reports
        res.update(dry_run=True, data="dry run: no kernels were launched",
                   shard_check={"col0_sum": float(t[1]), "eta_sum_x_world": float(t[2]), "eta_sum": float(first[2])})
        wsize = 8 if args.precision == "double" else 4
        res["roofline"] = {"kernel": None, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                           "traffic": None, "bytes_per_launch": NL_WORDS_PER_COL * wsize * nx, "columns": nx,
                           "note": "dry run: no kernel was launched; the object's shape is what main() fills in"}
        if args.cpu_cols > 0:
            res["cpu_baseline"] = cpu_baseline(args.cpu_cols, nz, np.float64, budget_s=args.cpu_budget_s)
        print(json.dumps(res), flush=True)

    rank0_then_everyone(dist if dist.is_initialized() else None, rank, rank0_record)
    if dist.is_initialized():
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))       # nothing has touched torch / the GPU in this process

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.config in (3, 4):
        return harness_bench(args, rank, local_rank, world)

    t_start = time.perf_counter()
    import numpy as np
    import torch

    import __graft_entry__ as ge

    startup = {"import_s": time.perf_counter() - t_start}
    t1 = time.perf_counter()
    ge.build()
    startup["build_s"] = time.perf_counter() - t1
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib, storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import DEFAULT_TIMESTEP_S, default_externals
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, NL_IN, NL_OUT, compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    rehearsal = args.collective == "gloo"
    if rehearsal:
        local_rank %= torch.cuda.device_count()          # ranks may share a GPU: every rank still runs its own shard's kernels
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    red_device = torch.device("cpu") if rehearsal else device      # where the (three, tiny) reductions' tensors live
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # one process per GPU under torch.distributed.run; backend "nccl" is RCCL on ROCm.  (Also taken
        # for a 1-rank torchrun launch, which exercises the collective code path on a single GPU.)
        import torch.distributed as dist

        with StdoutToStderr():
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=device)
            dist.barrier()          # creates the RCCL communicator (and prints its banner) now
            torch.cuda.synchronize()

    np_dtype = np.float64 if args.precision == "double" else np.float32
    wsize = np.dtype(np_dtype).itemsize
    nx, nz = args.cols, args.nlev
    total = nx * world
    ext = default_externals()
    dt = DEFAULT_TIMESTEP_S
    com = dict(origin=(0, 0, 0), validate_args=False, exec_info=None)
    last_kernel = lambda: _lib.last_kernel()  # noqa: E731

    # resident state: this rank's slice [rank*nx, (rank+1)*nx) of the global problem
    t1 = time.perf_counter()
    s = make_resident_state(total, nz, rank * nx, nx, np_dtype, device)
    torch.cuda.synchronize()
    startup["state_s"] = time.perf_counter() - t1
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=device)  # from GLOBAL column 0
    sat = compile_stencil("saturation", ext)
    _nl = compile_stencil("cloudsc2_nl", ext)
    nl_launches = [0]        # cloudsc2_nl launches in this rank's working precision so far (the placement tuner's included):
                             # lets a kernel trace of this command be cut to the launches of the event-timed pass

    def nl(**kw):
        if kw["in_ap"].dtype == storage.torch_dtype(np_dtype):
            nl_launches[0] += 1
        return _nl(**kw)

    def step_on(F):
        sat(in_ap=F["in_ap"], in_t=F["in_t"], out_qsat=F["in_qsat"], domain=(nx, 1, nz), **com)
        nl(**F, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)

    def faster_of_tuned_and_plain(Ft, launch, report, dtype):
        """One more candidate for a tuned leg: plain allocations, what the drivers give a caller who does not opt in.  On a
        lease where the allocator dealt well they beat the best placement of the arena (r03: 189 against 186 M columns/s
        at the headline size; cloudsc2_nl fp32 at 524 288 columns 1.27-1.49 ms against 1.59 ms on leases where the arena
        holds no fast placement), and a record whose opt-in path is slower than its default path describes the tuner, not
        the kernels.  Five interleaved rounds of the leg's own launch decide; the loser is freed.  Returns the fields."""
        if args.no_placement_recheck:
            return Ft
        try:
            Fs = {k: storage.from_klayout(storage.klayout(v).clone(), dtype, device) for k, v in Ft.items()}

            def round_ms(fields, n):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(n):
                    launch(fields)
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / n

            n = max(5, min(20, int(8.0 / max(round_ms(Ft, 3), 1e-3))))       # ~8 ms of work per round
            round_ms(Fs, n)
            tt, ts = [], []
            for _ in range(5):
                tt.append(round_ms(Ft, n))
                ts.append(round_ms(Fs, n))
            t_tuned, t_plain = sorted(tt)[2], sorted(ts)[2]
            report.update(recheck_tuned_ms=t_tuned, recheck_plain_ms=t_plain)
            if t_plain < 0.995 * t_tuned:
                report.update(mode="separate", chosen="plain allocations (faster than the tuner's winner on this lease)",
                              arena_GB=0.0)
                Ft = Fs
            else:
                report["chosen"] = "tuned arena"
            del Fs
            torch.cuda.empty_cache()
        except RuntimeError as exc:
            report["recheck_error"] = f"{type(exc).__name__}: {exc}"[:200]
            torch.cuda.empty_cache()
        return Ft

    # Placement of the step's 26 fields in HBM (gt4py_dwarf_p_cloudsc2_tl_ad_amd/storage.py: FieldArena, tune_placement).
    # The kernels, their arguments and their results are the same for every placement; what changes is how the 26
    # concurrent streams fall onto HBM channels and banks (docs/TUNING_LOG.md 3.7).
    order = ["in_" + n for n in NL_IN] + ["out_" + n for n in NL_OUT]
    sources = {"in_" + k[2:]: v for k, v in s.items()}
    placement = {"mode": args.placement}
    F = None
    if args.placement == "tuned" and time.perf_counter() - t_start > args.startup_budget_s:
        placement = {"mode": "separate", "tune_skipped": f"start-up took {time.perf_counter() - t_start:.0f} s before the tuner "
                                                         f"(> --startup-budget-s {args.startup_budget_s:.0f})"}
    elif args.placement == "tuned":
        try:
            tkw = {}
            if args.tune_shifts_mb:      # dev A/B of the shift range: explicit, validated (was an environment switch)
                tkw = dict(shifts_mb=args.tune_shifts_mb, max_arena_bytes=96 << 30, max_shift_spans=1e9)
            t_tune = time.perf_counter()
            F, rep = storage.tune_placement(nx, nz, np_dtype, device, order, sources, step_on, **tkw)
            torch.cuda.synchronize()
            placement.update(rep)
            # what the opt-in costs: the arena kept alive for the state (the 26 fields themselves are 1.9 GB at the
            # headline size) and the wall time of the calibration, outside the timed window
            placement.update(arena_GB=rep.get("arena_bytes", 0) / 1e9, tuning_s=time.perf_counter() - t_tune)
        except RuntimeError as exc:      # e.g. a shared device without room for the arena: say so, run on plain allocations
            placement = {"mode": "separate", "tune_error": f"{type(exc).__name__}: {exc}"[:300]}
            torch.cuda.empty_cache()
    if F is not None and placement.get("mode") == "tuned":
        F = faster_of_tuned_and_plain(F, step_on, placement, np_dtype)
    if F is None:
        old_cap = storage.set_arena_capacity(32 if placement["mode"] == "arena" else 0)
        F = {k: storage.from_klayout(v, np_dtype, device) for k, v in sources.items()}
        F["in_qsat"] = storage.zeros(nx, nz, np_dtype, device)
        F.update({"out_" + n: storage.zeros(nx, nz, np_dtype, device) for n in NL_OUT})
        storage.set_arena_capacity(old_cap)
    del s, sources
    startup["tune_s"] = placement.get("tuning_s", 0.0)
    startup["to_first_step_s"] = time.perf_counter() - t_start
    print(f"[bench] rank {rank}: import {startup['import_s']:.1f} s, build {startup['build_s']:.1f} s, state "
          f"{startup['state_s']:.1f} s, placement {startup['tune_s']:.1f} s ({placement.get('chosen', placement['mode'])}); "
          f"{startup['to_first_step_s']:.1f} s from process start to the first step", file=sys.stderr, flush=True)
    ins = {k: v for k, v in F.items() if k.startswith("in_")}
    outs = {k: v for k, v in F.items() if k.startswith("out_")}
    qsat = F["in_qsat"]

    def sat_only():
        sat(in_ap=F["in_ap"], in_t=F["in_t"], out_qsat=qsat, domain=(nx, 1, nz), **com)

    def nl_only():
        nl(**ins, **outs, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)

    def step():
        sat_only()
        nl_only()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def event_times(fn, reps, before=None):
        """mean HIP-event interval around each `fn()` of a train of `reps` (+2 discarded) launches; everything is
        enqueued before the first synchronisation, so the GPU never waits for the host"""
        evs = []
        for _ in range(reps + 2):
            if before is not None:
                before()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        times = [a.elapsed_time(b) for a, b in evs][2:]
        return sum(times) / len(times)

    # dominant-kernel duration: HIP events on the launch stream (torch's current stream, the one the C ABI
    # launches on) around EVERY cloudsc2_nl launch of a pass over the timed region's pattern
    # (saturation, cloudsc2_nl, ...); rocprofv3 --kernel-trace of this command reports the same average (profiles/).
    import gc

    gc.collect()        # once, here: nothing below may pause the host for milliseconds right before the timed window
    gc.disable()        # (reference counting still frees every tensor; the cyclic collector is not needed)
    nl_ms = nl_train_ms = copy_gbs = None
    nl_kernel_name = None
    extra = {}
    if not args.no_roofline_events:
        for _ in range(70):          # untimed, >= 25 ms of the same work: the GPU needs 10-15 ms after an idle period to
            step()                   # reach its steady clocks (profiles/r02/window_probe.txt); 20 steps were not enough
        nl_kernel_name = last_kernel()
        reps = max(10, min(args.steps, 50))
        nl_window_first = nl_launches[0] + 2          # event_times discards its first two intervals
        nl_ms = event_times(nl_only, reps, before=sat_only)
        nl_window = [nl_window_first, nl_launches[0] - 1]
        # the same kernel in a back-to-back train (no other kernel in between), for reference
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            nl_only()
        b.record()
        torch.cuda.synchronize()
        nl_train_ms = a.elapsed_time(b) / reps
        # a 1 GiB torch device-to-device copy on this box (read + write bytes), for orientation only: it is SLOWER than the
        # NL kernel's own stream rate, i.e. not a ceiling
        src = torch.empty(1 << 27, dtype=torch.float64, device=device)
        dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            dst.copy_(src)
        b.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * src.numel() * 8 * 5 / (a.elapsed_time(b) * 1e-3) / 1e9
        del src, dst

        def extra_rooflines():
            out = {}
            extn = dict(ext, NLEV=nz)
            KL = storage.klayout
            tuned = args.placement == "tuned"

            def steady(call, ms=40.0):
                """>= `ms` of the leg's own launches, back to back, right before its event-timed pass: allocating and
                freeing fields (the placement recheck, torch.cuda.empty_cache) leaves the GPU idle for tens of ms, and the
                first 10-15 ms of work after an idle period run at lower clocks (profiles/r02/window_probe.txt) - the
                first version of the recheck read cloudsc2_ad 5 % slow for that reason alone"""
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    call()
                b.record()
                torch.cuda.synchronize()
                for _ in range(max(3, int(ms / max(a.elapsed_time(b) / 3, 1e-3)))):
                    call()

            def tl_ad_legs(n_c, dt_np, ins_c, eta_c, sfx, reps):
                """`roofline_tl<sfx>` / `roofline_ad<sfx>`: cloudsc2_tl on (state, 0.01 x state), then cloudsc2_ad forced with
                the TL perturbation outputs, `n_c` columns of dtype `dt_np`, each on its own (tuned or default) placement"""
                w = np.dtype(dt_np).itemsize
                prec = "double" if w == 8 else "single"
                Zc = lambda: storage.zeros(n_c, nz, dt_np, device)  # noqa: E731

                def placed(order_, sources_, launch_):
                    if tuned:
                        Ft_, rep_ = storage.tune_placement(n_c, nz, dt_np, device, order_, sources_, launch_)
                        rep_["mode"] = "tuned"
                        return faster_of_tuned_and_plain(Ft_, launch_, rep_, dt_np), rep_
                    Fd = {k: (storage.from_klayout(sources_[k], dt_np, device) if sources_.get(k) is not None else Zc())
                          for k in order_}
                    return Fd, {"mode": args.placement}

                inc_out = {"out_" + n + "_i": Zc() for n in INC}
                compile_stencil("state_increment", {"IGNORE_SUPSAT": True})(
                    **{"in_" + n: ins_c["in_" + n] for n in INC}, **inc_out, f=0.01, domain=(n_c, 1, nz + 1), **com)
                tl = compile_stencil("cloudsc2_tl", extn)
                tl_order = (["in_" + n for n in NL_IN] + ["in_" + n + "_i" for n in NL_IN] + ["out_" + n for n in NL_OUT]
                            + ["out_" + n + "_i" for n in NL_OUT])
                tl_src = {"in_" + n: KL(ins_c["in_" + n]) for n in NL_IN}
                tl_src.update({"in_" + n + "_i": KL(inc_out["out_" + n + "_i"]) for n in NL_IN})
                tl_launch = lambda Ft: tl(**Ft, in_eta=eta_c, dt=dt, domain=(n_c, 1, nz + 1), **com)  # noqa: E731
                Ft, tl_rep = placed(tl_order, tl_src, tl_launch)
                del inc_out, tl_src
                tl_call = lambda: tl_launch(Ft)  # noqa: E731
                steady(tl_call)
                tl_name = last_kernel()
                tl_ms = event_times(tl_call, reps)
                out["roofline_tl" + sfx] = roofline_entry(tl_name, TLAD_WORDS_PER_COL, w, n_c, prec, tl_ms, placement=tl_rep)
                ad = compile_stencil("cloudsc2_ad", extn)
                ad_order = (["in_" + n for n in NL_IN] + ["in_" + n + "_i" for n in NL_OUT] + ["out_" + n for n in NL_OUT]
                            + ["out_" + n + "_i" for n in NL_IN])
                ad_src = {"in_" + n: KL(ins_c["in_" + n]) for n in NL_IN}
                ad_src.update({"in_" + n + "_i": KL(Ft["out_" + n + "_i"]) for n in NL_OUT})      # forced with the TL perturbations
                ad_launch = lambda Fa: ad(**Fa, in_eta=eta_c, dt=dt, domain=(n_c, 1, nz + 1), **com)  # noqa: E731
                Fa, ad_rep = placed(ad_order, ad_src, ad_launch)
                del Ft, ad_src
                ad_call = lambda: ad_launch(Fa)  # noqa: E731
                steady(ad_call)
                ad_name = last_kernel()
                ad_ms = event_times(ad_call, reps)
                out["roofline_ad" + sfx] = roofline_entry(ad_name, TLAD_WORDS_PER_COL, w, n_c, prec, ad_ms, placement=ad_rep)
                del Fa
                torch.cuda.empty_cache()

            tl_ad_legs(nx, np_dtype, ins, eta, "", 20)
            # cloudsc2_nl fp32 at the per-GPU shard of BASELINE configs[4] on 8 GPUs (524 288 columns)
            n32 = CONFIG5_COLUMNS // 8
            s32 = make_resident_state(n32, nz, 0, n32, np.float32, device)
            eta32 = torch.as_tensor(eta_levels(nz, dtype=np.float32), device=device)

            def step32(F32):
                sat(in_ap=F32["in_ap"], in_t=F32["in_t"], out_qsat=F32["in_qsat"], domain=(n32, 1, nz), **com)
                nl(**F32, in_eta=eta32, dt=dt, domain=(n32, 1, nz + 1), **com)

            src32 = {"in_" + k[2:]: v for k, v in s32.items()}
            if tuned:
                F32, rep32 = storage.tune_placement(n32, nz, np.float32, device, order, src32, step32)
                rep32["mode"] = "tuned"
                F32 = faster_of_tuned_and_plain(F32, step32, rep32, np.float32)
            else:
                F32 = {k: storage.from_klayout(v, np.float32, device) for k, v in src32.items()}
                F32["in_qsat"] = storage.zeros(n32, nz, np.float32, device)
                F32.update({"out_" + n: storage.zeros(n32, nz, np.float32, device) for n in NL_OUT})
                rep32 = {"mode": args.placement}
            del s32, src32
            sat32 = lambda: sat(in_ap=F32["in_ap"], in_t=F32["in_t"], out_qsat=F32["in_qsat"],  # noqa: E731
                                domain=(n32, 1, nz), **com)
            nl32 = lambda: nl(**F32, in_eta=eta32, dt=dt, domain=(n32, 1, nz + 1), **com)  # noqa: E731
            steady(lambda: step32(F32))
            name32 = last_kernel()
            ms32 = event_times(nl32, 10, before=sat32)
            out["roofline_nl_f32"] = roofline_entry(name32, NL_WORDS_PER_COL, 4, n32, "single", ms32, placement=rep32,
                                                      what="per-GPU shard of BASELINE configs[4] on 8 GPUs, timed "
                                                           "inside the (saturation, cloudsc2_nl) pattern")
            # cloudsc2_tl / cloudsc2_ad fp32 at the same shard (run_taylor_test.py / run_symmetry_test.py `--precision single`)
            ins32 = {k: v for k, v in F32.items() if k.startswith("in_")}
            del F32
            try:
                tl_ad_legs(n32, np.float32, ins32, eta32, "_f32", 10)
            except Exception as exc:  # noqa: BLE001 - e.g. a shared device without room: the fp64 legs stand
                out["extra_rooflines_f32_error"] = f"{type(exc).__name__}: {exc}"[:300]
            del ins32
            torch.cuda.empty_cache()
            return out

    # ---- the timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize pairs.
    # Clock state: this GPU runs the step ~12 % slower (0.425 vs 0.375 ms) for the first ~10-15 ms of work that follows
    # an idle period of >= ~50 ms, whatever caused it - a garbage collection, an allocation, the host building the next
    # state (profiles/window_probe.py, profiles/r02/window_probe.txt: a 200 ms pause before a 5-step warm-up leaves all
    # 20 timed steps slow; before a 60-step warm-up none).  Round 1 collected garbage between the warm-up and the
    # window and reported 0.417 ms for 20 steps where 100 steps gave 0.379.  The window must see the steady state of a
    # long run, so (a) the collector ran once at the top and stays off, and (b) the step is repeated back to back for
    # >= 25 ms (PREWARM steps, the same launches as the warm-up, reported in the record) immediately before the W
    # warm-up steps; nothing but the opening barrier separates them from the K timed steps.
    prewarm = max(0, 70 - args.warmup)
    for _ in range(prewarm):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0      # this rank's K steps, from the common start to its own completion
    barrier()                               # closing bracket; the job time is the MAX over ranks taken below
    gc.enable()
    per_rank_ms = [1e3 * elapsed / args.steps]
    if dist is not None:
        # every rank's own time for its K steps (a straggler must be visible in the record), then the job time = MAX
        t = torch.zeros(world, dtype=torch.float64, device=red_device)
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [1e3 * float(x) / args.steps for x in t.cpu()]
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same step on the DEFAULT placement (one torch allocation per field - what `storage.zeros` and the drivers give a
    # caller who does not opt into the tuner), timed right after the headline window on a copy of the state: reported
    # beside `value` so that the record never describes only the opt-in path
    default_placement = None
    if placement.get("mode") == "tuned" and not args.no_roofline_events and nx <= 1 << 20:
        try:
            Fd = {k: storage.from_klayout(storage.klayout(v).clone(), np_dtype, device) for k, v in F.items()}
            for _ in range(max(70, args.warmup)):
                step_on(Fd)
            torch.cuda.synchronize()
            td = time.perf_counter()
            for _ in range(args.steps):
                step_on(Fd)
            torch.cuda.synchronize()
            td = time.perf_counter() - td
            default_placement = {"ms_per_step": 1e3 * td / args.steps, "value_this_rank": nx * args.steps / td}
            del Fd
        except RuntimeError as exc:
            default_placement = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            torch.cuda.empty_cache()

    # the same step as ONE launch (build extension: saturation evaluated inside the NL kernel, stencil
    # `cloudsc2_nl_saturation`); reported beside the headline, never as `value`
    fused = None
    if not args.no_roofline_events and world == 1:
        try:
            nls = compile_stencil("cloudsc2_nl_saturation", ext)
            qsat2 = storage.zeros(nx, nz, np_dtype, device)
            outs2 = {"out_" + n: storage.zeros(nx, nz, np_dtype, device) for n in NL_OUT}
            ins2 = {k: v for k, v in ins.items() if k != "in_qsat"}

            def fused_step():
                nls(**ins2, out_qsat=qsat2, **outs2, in_eta=eta, dt=dt, domain=(nx, 1, nz + 1), **com)

            for _ in range(args.warmup):
                fused_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                fused_step()
            torch.cuda.synchronize()
            fel = time.perf_counter() - t1
            same = all(bool(torch.equal(outs2[k], outs[k])) for k in outs) and bool(torch.equal(qsat2, qsat))
            fused = {"ms_per_step": 1e3 * fel / args.steps, "value": nx * args.steps / fel, "unit": "columns/s",
                     "kernel": last_kernel(), "results_equal_unfused": same,
                     "what": "saturation + cloudsc2_nl as one launch (cloudsc2_nl_fused_*)"}
            del qsat2, outs2
        except (ValueError, RuntimeError) as exc:   # e.g. fields of 4 GiB and more: the fused extension keeps 32-bit offsets
            fused = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            torch.cuda.empty_cache()

    # ---- the other kernels of the path, each against its own roofline (N = 1, headline configuration only), AFTER the
    # timed window: their tuning runs other sizes and frees GBs of arenas, and a 20-step window that followed them was
    # 2 % slower than one that did not (profiles/r02/window_vs_extras.txt).  A failure here (e.g. out of memory on a
    # shared device) must not cost the headline line: it is recorded and the run goes on.
    if (not args.no_roofline_events and world == 1 and not args.no_extra_rooflines and args.config == 2
            and nx <= 131072):
        try:
            extra.update(extra_rooflines())
        except Exception as exc:  # noqa: BLE001
            extra["extra_rooflines_error"] = f"{type(exc).__name__}: {exc}"
            torch.cuda.empty_cache()

    # validation norm (the only data reduction across ranks): sum of every NL output
    norm = torch.stack([storage.klayout(outs["out_" + n]).double().abs().sum() for n in NL_OUT])
    finite = all(bool(torch.isfinite(storage.klayout(v)[: nz]).all()) for v in outs.values())
    if dist is not None:
        norm = norm.to(red_device)
        dist.all_reduce(norm, op=dist.ReduceOp.SUM)
    norm = [float(x) for x in norm.cpu()]
    mine = {"rank": rank, "device": f"cuda:{local_rank}", "ms_per_step": per_rank_ms[rank] if dist is not None else per_rank_ms[0],
            "nl_kernel_ms": nl_ms, "startup_s": startup}
    mine.update({k: placement[k] for k in ("mode", "chosen", "arena_GB", "tuning_s", "tuned_ms", "default_ms", "recheck_tuned_ms",
                                            "recheck_plain_ms", "tune_error", "tune_skipped", "recheck_error") if k in placement})
    mine.setdefault("chosen", {"separate": "plain allocations", "arena": "default arena"}.get(placement["mode"], placement["mode"])
                    + (" (tuner failed or skipped)" if "tune_error" in placement or "tune_skipped" in placement else " (as requested)"))
    rank_reports = gather_rank_reports(dist, world, mine)

    def rank0_record():
        res = base_record(args, world, nx, nz, value=total * args.steps / elapsed,
                          ms_per_step=1e3 * elapsed / args.steps,
                          ranks=dist.get_world_size() if dist is not None else None,
                          backend=("gloo (rehearsal: kernels real, ranks may share a GPU - not a scaling figure)" if rehearsal
                                   else "nccl (RCCL)") if dist is not None else "none (single process)")
        if rehearsal and dist is not None:
            res["rccl_ranks"], res["rehearsal_ranks"] = None, dist.get_world_size()
        res["prewarm_steps"] = prewarm
        res["per_rank_ms"] = per_rank_ms
        res["per_rank_ms_min_max"] = [min(per_rank_ms), max(per_rank_ms)]
        res["placement"] = placement
        # every rank's own placement outcome, start-up times and kernel time, in rank order (rank 0's `placement` above is
        # the full report of ONE rank; a rank that skipped the tuner or fell back to plain allocations shows here)
        res["per_rank_placement"] = rank_reports
        res["startup_s_max_over_ranks"] = max(r["startup_s"]["to_first_step_s"] for r in rank_reports)
        if default_placement is None and placement.get("mode") == "separate":
            # the timed region itself ran on plain allocations (asked for, or they beat the tuner's winner, or the tuner
            # failed / was skipped): the two figures coincide
            res["value_default_placement"] = res["value"]
            res["ms_per_step_default_placement"] = res["ms_per_step"]
            res["default_placement_is_value"] = True
        if default_placement is not None:
            # rank 0's figure on separate allocations, scaled to the job (weak scaling: every rank does the same work)
            if "value_this_rank" in default_placement and rehearsal and dist is not None:
                # ranks sharing ONE GPU: rank 0 timed this window while the others were already done - not a job figure
                res["value_default_placement"] = None
                res["default_placement_note"] = "not taken in a rehearsal (ranks share one GPU)"
            elif "value_this_rank" in default_placement:
                res["value_default_placement"] = default_placement["value_this_rank"] * world
                res["ms_per_step_default_placement"] = default_placement["ms_per_step"]
            else:
                res["value_default_placement"] = None
                res["default_placement_error"] = default_placement.get("error")
        res["outputs_finite"] = finite
        res["validation_norm"] = dict(zip(NL_OUT, norm))
        if nl_ms is not None:
            res["roofline"] = roofline_entry(nl_kernel_name, NL_WORDS_PER_COL, wsize, nx, args.precision, nl_ms,
                                             avg_launch_ms_back_to_back=nl_train_ms, box_torch_copy_GBs=copy_gbs,
                                             launch_window=nl_window)
        res.update(extra)
        if fused is not None:
            res["fused_step"] = fused
        if args.cpu_cols > 0:
            # the CPU baseline beside the GPU figure at EVERY world size (north_star: "in the same run"), on rank 0, after
            # the closing barrier and every reduction; the other ranks wait in rank0_then_everyone without spinning
            def hip_step(Fh, eta_h, dt_h):
                """saturation + cloudsc2_nl through the stencil objects on host fields in [k][col] layout (fp64)"""
                n_c = Fh["in_ap"].shape[1]
                D = {k: storage.from_klayout(np.ascontiguousarray(v), np.float64, device) for k, v in Fh.items()
                     if k.startswith("in_")}
                D["in_qsat"] = storage.zeros(n_c, nz, np.float64, device)
                O = {"out_" + n: storage.zeros(n_c, nz, np.float64, device) for n in NL_OUT}
                s64 = compile_stencil("saturation", ext)
                n64 = compile_stencil("cloudsc2_nl", ext)
                s64(in_ap=D["in_ap"], in_t=D["in_t"], out_qsat=D["in_qsat"], domain=(n_c, 1, nz), **com)
                n64(**D, **O, in_eta=torch.as_tensor(np.asarray(eta_h, dtype=np.float64), device=device), dt=dt_h,
                    domain=(n_c, 1, nz + 1), **com)
                torch.cuda.synchronize()
                got = {n: storage.klayout(O["out_" + n]).cpu().numpy() for n in NL_OUT}
                got["qsat"] = storage.klayout(D["in_qsat"]).cpu().numpy()
                return got

            budget = args.cpu_budget_s if time.perf_counter() - t_start < 200.0 else min(args.cpu_budget_s, 3.0)
            res["cpu_baseline"] = cpu_baseline(args.cpu_cols, nz, np.float64, budget_s=budget, hip_step=hip_step)
        res["wall_s_rank0"] = time.perf_counter() - t_start
        print(json.dumps(res), flush=True)

    rank0_then_everyone(dist, rank, rank0_record)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
