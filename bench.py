#!/usr/bin/env python3
"""Headline benchmark: CLOUDSC2-NL fp64, 65 536 columns x 137 levels per GPU (BASELINE.json configs[1]).

One "step" = the reference driver's timed region (/root/reference/drivers/run_nonlinear.py:115-119):
`saturation(state)` followed by `cloudsc2_nl(state, dt)` on the same resident state, i.e. two kernel
launches through the C ABI of libcloudsc2_hip.so.  Inputs are synthetic columns generated directly
in HBM (gt4py_dwarf_p_cloudsc2_tl_ad_amd/synthetic.py; the reference's data/input.h5 is not
available) and are resident before the timed region starts.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, every rank owns `--cols` columns of a global N*cols-column problem
(weak scaling; columns are independent, so there is NO data-path collective).  RCCL is used only
for the barrier / max-over-ranks timing and for the final validation-norm all-reduce.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (`nl_ring_kernel`, the LDS-ring load path of cloudsc2_nl): algorithmic bytes
per launch (28 536 B/column fp64) / the kernel's mean duration measured with HIP
events on the launch stream.  `cpu_baseline` times the C/OpenMP and NumPy restatements on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NL_WORDS_PER_COL = 3567          # 15*137 + 138 read, 6*137 + 4*138 written
SAT_WORDS_PER_COL = 411          # 2 in, 1 out over 137 levels
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cols", type=int, default=65536, help="columns per GPU")
    ap.add_argument("--nlev", type=int, default=137)
    ap.add_argument("--precision", choices=["double", "single"], default="double")
    ap.add_argument("--cpu-cols", type=int, default=16384,
                    help="columns of the CPU-baseline sample (0 disables the baseline)")
    ap.add_argument("--no-roofline-events", action="store_true")
    return ap.parse_args()


def cpu_baseline(cols: int, nz: int, np_dtype, budget_s: float = 8.0):
    """CPU baselines on the host, saturation + cloudsc2_nl on `cols` synthetic columns (BASELINE configs[0] size):
      * headline `value`: the plain-C + OpenMP restatement (oracle/cloudsc2_nl_omp.c, SURVEY 8d "restatement B")
        on the host cores of this GPU's share, fp64, repeated for ~`budget_s` s;
      * `numpy_1core`: the NumPy restatement (GT4Py-numpy-like execution shape, single-threaded) for ~6 s.
    Both are the checker (pinned to the executed reference source by tests/), timed here only as baselines."""
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

    n_runs, n_el = loop(numpy_step, 6.0)
    numpy_rate = cols * n_runs / n_el
    what = f"saturation + cloudsc2_nl, {cols} cols x {nz} lev"
    res = {"unit": "columns/s", "host_cores": os.cpu_count(), "kind": "port",
           "numpy_1core": {"value": numpy_rate, "unit": "columns/s", "cores": 1,
                           "sample": f"NumPy restatement, {what} {np.dtype(np_dtype).name}, {n_runs} runs in {n_el:.1f} s"}}
    try:
        threads = min(len(os.sched_getaffinity(0)), 16)
        F = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in fields.items()}
        for n in NL_OUT:
            F["out_" + n] = np.zeros_like(F["in_ap"])

        def c_step():
            cloudsc2_c.saturation(F["in_ap"], F["in_t"], F["in_qsat"], ext, nthreads=threads)
            cloudsc2_c.cloudsc2_nl(F, eta, dt, ext, nthreads=threads)

        c_runs, c_el = loop(c_step, budget_s)
        res.update(value=cols * c_runs / c_el, cores=threads,
                   sample=f"plain-C + OpenMP restatement (oracle/cloudsc2_nl_omp.c, {threads} threads, scalar libm), "
                          f"{what} float64, {c_runs} runs in {c_el:.1f} s, synthetic-parameters")
    except Exception as exc:  # the C library is optional test infrastructure: fall back to the NumPy figure
        res.update(value=numpy_rate, cores=1, sample=res["numpy_1core"]["sample"] + f" (C restatement unavailable: {exc})")
    return res


class _StdoutToStderr:
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


def pmc_traffic(nx: int, precision: str):
    """HBM bytes per cloudsc2_nl launch from the rocprofv3 PMC passes committed under profiles/ (separate
    FETCH_SIZE / WRITE_SIZE passes of this same command, profiles/run_rocprof.sh).  gfx950 correction
    (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-B requests of a wide streaming read at 64 B ->
    doubled; WRITE_SIZE is exact; both are in KiB.  None when no summary matches this workload."""
    if nx != 65536 or precision != "double":
        return None, None
    path = os.path.join(ROOT, "profiles", "r01", "nl_fp64_65536_pmc.json")
    try:
        with open(path) as fh:
            pm = json.load(fh)
        k = ([v for n, v in pm.items() if "nl_ring_kernel" in n] or [v for n, v in pm.items() if "nl_kernel" in n])[0]
        fetch = k["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0
        write = k["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
    except (OSError, KeyError, IndexError, ValueError):
        return None, None
    return 2.0 * fetch + write, (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {os.path.relpath(path, ROOT)}: "
                                 f"2 x {fetch / 1e9:.3f} GB read + {write / 1e9:.3f} GB written per launch")


def main():
    args = parse_args()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import __graft_entry__ as ge

    ge.build()
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import DEFAULT_TIMESTEP_S, default_externals
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_OUT, compile_stencil
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.synthetic import eta_levels, make_state

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # one process per GPU under torch.distributed.run; backend "nccl" is RCCL on ROCm.  (Also taken
        # for a 1-rank torchrun launch, which exercises the collective code path on a single GPU.)
        import torch.distributed as dist

        with _StdoutToStderr():
            dist.init_process_group("nccl", device_id=device)
            dist.barrier()          # creates the RCCL communicator (and prints its banner) now
            torch.cuda.synchronize()

    np_dtype = np.float64 if args.precision == "double" else np.float32
    wsize = np.dtype(np_dtype).itemsize
    nx, nz = args.cols, args.nlev
    total = nx * world
    ext = default_externals()
    dt = DEFAULT_TIMESTEP_S

    # resident state: this rank's slice [rank*nx, (rank+1)*nx) of the global problem
    s = make_state(total, nz, col0=rank * nx, ncols=nx, dtype=np_dtype, device=device)
    eta = torch.as_tensor(eta_levels(nz, dtype=np_dtype), device=device)  # from GLOBAL column 0
    f = {k: storage.logical_view(v) for k, v in s.items()}
    qsat = storage.zeros(nx, nz, np_dtype, device)
    ins = {"in_" + k[2:]: v for k, v in f.items()}
    ins["in_qsat"] = qsat
    outs = {"out_" + n: storage.zeros(nx, nz, np_dtype, device) for n in NL_OUT}
    sat = compile_stencil("saturation", ext)
    nl = compile_stencil("cloudsc2_nl", ext)

    def step():
        sat(in_ap=f["f_ap"], in_t=f["f_t"], out_qsat=qsat, origin=(0, 0, 0), domain=(nx, 1, nz),
            validate_args=False, exec_info=None)
        nl(**ins, **outs, in_eta=eta, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1),
           validate_args=False, exec_info=None)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # dominant-kernel duration: HIP events on the launch stream (torch's current stream, the one the C ABI
    # launches on) around EVERY cloudsc2_nl launch of a second pass over the timed region's pattern
    # (saturation, cloudsc2_nl, ...).  All launches and event records are enqueued before the first
    # synchronisation, so the GPU never waits for the host and an event interval is kernel time only;
    # rocprofv3 --kernel-trace of this command reports the same average (profiles/).
    nl_ms = None
    nl_train_ms = None
    copy_gbs = None
    if not args.no_roofline_events:
        def nl_only():
            nl(**ins, **outs, in_eta=eta, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1),
               validate_args=False, exec_info=None)

        for _ in range(20):          # untimed: bring the GPU out of its idle power state first
            step()
        reps = max(10, min(args.steps, 50))
        evs = []
        for _ in range(reps):
            sat(in_ap=f["f_ap"], in_t=f["f_t"], out_qsat=qsat, origin=(0, 0, 0), domain=(nx, 1, nz),
                validate_args=False, exec_info=None)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            nl_only()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        times = [a.elapsed_time(b) for a, b in evs][2:]
        nl_ms = sum(times) / len(times)
        # the same kernel in a back-to-back train (no other kernel in between), for reference
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            nl_only()
        b.record()
        torch.cuda.synchronize()
        nl_train_ms = a.elapsed_time(b) / reps
        # this box's streaming-copy ceiling (1 GiB device-to-device copy, read + write bytes)
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

    # ---- the timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize pairs.  It runs AFTER the
    # event-timed passes above on purpose: those ~150 launches bring the GPU out of its idle power state, so that the
    # wall-clock figure is not dominated by the clock ramp of the first milliseconds (with 5 warm-up steps straight
    # from idle the same 50 steps measured 377 us per step instead of 343 us, profiles/host_overhead.py).
    import gc

    for _ in range(args.warmup):
        step()
    gc.collect()
    gc.disable()        # no collector pause inside the wall-clock window (every stencil call builds small ctypes arrays)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0      # this rank's K steps, from the common start to its own completion
    barrier()                               # closing bracket; the job time is the MAX over ranks taken below
    gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same step as ONE launch (build extension: saturation evaluated inside the NL kernel, stencil
    # `cloudsc2_nl_saturation`); reported beside the headline, never as `value`
    fused = None
    if not args.no_roofline_events and world == 1:
        nls = compile_stencil("cloudsc2_nl_saturation", ext)
        qsat2 = storage.zeros(nx, nz, np_dtype, device)
        outs2 = {"out_" + n: storage.zeros(nx, nz, np_dtype, device) for n in NL_OUT}
        ins2 = {k: v for k, v in ins.items() if k != "in_qsat"}

        def fused_step():
            nls(**ins2, out_qsat=qsat2, **outs2, in_eta=eta, dt=dt, origin=(0, 0, 0), domain=(nx, 1, nz + 1),
                validate_args=False, exec_info=None)

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
                 "results_equal_unfused": same, "what": "saturation + cloudsc2_nl as one launch (cloudsc2_nl_fused_*)"}
        del qsat2, outs2

    # validation norm (the only data reduction across ranks): sum of every NL output
    norm = torch.stack([storage.klayout(outs["out_" + n]).double().abs().sum() for n in NL_OUT])
    finite = all(bool(torch.isfinite(storage.klayout(v)[: nz]).all()) for v in outs.values())
    if dist is not None:
        dist.all_reduce(norm, op=dist.ReduceOp.SUM)
    norm = [float(x) for x in norm.cpu()]

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total * args.steps / elapsed
        res = {
            "metric": "columns/sec at 137 levels fp64; achieved HBM GB/s vs MI355X roofline",
            "value": value,
            "unit": "columns/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == "double" else "f32",
            "data": "synthetic columns + synthetic-parameters (reference data/input.h5 unavailable)",
            "config": {
                "workload": f"CLOUDSC2-NL (saturation + cloudsc2_nl), {nx} cols x {nz} lev per GPU, "
                            f"{args.precision}, {world} GPU(s), {total} columns total",
                "columns_per_gpu": nx, "levels": nz, "timestep_s": dt,
                "parallelism": f"column-sharded x{world}, no data-path collective",
            },
            "outputs_finite": finite,
            "validation_norm": dict(zip(NL_OUT, norm)),
        }
        if nl_ms is not None:
            nl_bytes = NL_WORDS_PER_COL * wsize * nx
            achieved = nl_bytes / (nl_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(nx, args.precision)
            res["roofline"] = {
                "kernel": "cs2::nl_ring_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                "traffic_source": traffic_src,
                "bytes_per_launch": nl_bytes, "avg_launch_ms": nl_ms,
                "kernel_columns_per_s": nx / (nl_ms * 1e-3),
                "avg_launch_ms_back_to_back": nl_train_ms,
                "box_copy_ceiling_GBs": copy_gbs,
            }
        if fused is not None:
            res["fused_step"] = fused
        if world == 1 and args.cpu_cols > 0:
            res["cpu_baseline"] = cpu_baseline(args.cpu_cols, nz, np_dtype)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
