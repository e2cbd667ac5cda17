"""TL Taylor-test driver: counterpart of /root/reference/drivers/run_taylor_test.py:41-196."""
from __future__ import annotations

import argparse
import statistics

from ..framework.timing import Timer
from ..harness import TaylorTest
from ._common import add_common_options, init_distributed_from_env, report_placement, setup, tune_field_placement


def core(args):
    ctx = setup(args)
    cfg, p = ctx["config"], ctx["params"]
    tt = TaylorTest(ctx["grid"], factor1=0.01, factor2s=tuple(10 ** -(i + 1) for i in range(10)), kflag=1,
                    lphylin=True, ldrain1d=False, yoethf_params=p["yoethf"], yomcst_params=p["yomcst"],
                    yrecldp_params=p["yrecldp"], yrephli_params=p["yrephli"], yrncl_params=p["yrncl"],
                    yrphnc_params=p["yrphnc"], enable_checks=cfg.sympl_enable_checks, gt4py_config=cfg.gt4py_config,
                    fused=args.fused or args.fused_stored, fused_norms=args.fused_norms, fused_all=args.fused_all,
                    graph=args.graph, store_perturbed=args.fused_stored)
    norms = tt.run(ctx["state"], ctx["dt"])                  # warm-up; these norms are the validated ones
    if args.output_csv_file_stencils is not None and not args.graph:
        cfg.gt4py_config.reset_exec_info()                    # run_taylor_test.py:93: per-stencil HIP events from here on
    if args.tune_placement:
        # build extension (docs/TUNING_LOG.md 3.7): the ~90 fields of the test are re-placed in HBM where a whole run is fastest
        # (a captured HIP graph holds the OLD field addresses: candidates are timed eagerly and the graph is re-captured
        # on the fields' final placement)
        graph, tt.graph, tt._graphed = tt.graph, False, None
        try:
            ctx["placement"] = tune_field_placement(
                [ctx["state"], tt.diags_sat, tt.state_i, tt.state_p, tt.tends_nl, tt.diags_nl, tt.tends_tl, tt.diags_tl,
                 tt.tends_nl_p, tt.diags_nl_p], lambda: tt.run(ctx["state"], ctx["dt"]), budget_s=8.0)
        finally:
            tt.graph = graph
        report_placement(ctx["placement"], "run")
    runtimes = []
    for _ in range(cfg.num_runs):
        Timer.reset()
        tt.run(ctx["state"], ctx["dt"])
        runtimes.append(Timer.get_time("run"))
    ok = tt.validate(norms) if cfg.enable_validation else None
    mean = statistics.fmean(runtimes)
    std = statistics.stdev(runtimes) if len(runtimes) > 1 else 0.0
    print(f"\nThe test completed in {mean:.3f} ± {std:.3f} ms.")
    io = ctx["io_config"]
    if io.output_csv_file is not None:          # run_taylor_test.py:109-124 (+ the build's GB/s and roofline columns)
        from ..framework.output import write_performance_to_csv

        n = len(tt.f2s)
        head = ["saturation", "cloudsc2_nl"]
        if tt.fused_all:
            seq = head + ["cloudsc2_tl_incremented", "cloudsc2_nl_taylor_multi"]
        elif tt.fused_norms:
            seq = head + ["state_increment", "cloudsc2_tl"] + ["cloudsc2_nl_taylor"] * n
        elif tt.fused:
            seq = head + ["state_increment", "cloudsc2_tl"] + ["cloudsc2_nl_perturbed"] * n
        else:
            seq = head + ["state_increment", "cloudsc2_tl"] + ["perturbed_state", "cloudsc2_nl"] * n
        write_performance_to_csv(io.output_csv_file, io.host_name, cfg.precision, "tl-" + cfg.gt4py_config.backend,
                                 ctx["nx"], cfg.num_threads, 1, cfg.num_runs, mean, std, 0, 0, stencils=seq)
    if args.output_csv_file_stencils is not None:      # run_taylor_test.py, end of main(): one row per stencil from exec_info
        from ..framework.output import write_stencils_performance_to_csv

        write_stencils_performance_to_csv(args.output_csv_file_stencils, ctx["io_config"].host_name, cfg.precision,
                                          "tl-" + cfg.gt4py_config.backend, ctx["nx"], cfg.num_threads, cfg.num_runs,
                                          cfg.gt4py_config.exec_info, key_patterns=["cloudsc", "increment", "perturbed", "saturation"])
    ctx.update(norms=norms, passed=ok, runtimes_ms=runtimes, harness=tt)
    return ctx


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    add_common_options(ap)
    ap.add_argument("--fused", action="store_true",
                    help="apply the perturbation inside the NL kernel's loads and form the ten difference sums in its epilogue: "
                         "one launch per step size, nothing stored (build extension cloudsc2_nl_taylor)")
    ap.add_argument("--fused-norms", action="store_true", help="older name of --fused")
    ap.add_argument("--fused-stored", action="store_true",
                    help="perturbation inside the NL kernel's loads, perturbed outputs stored, sums as separate launches "
                         "(build extension cloudsc2_nl_perturbed; norms bit-equal to the unfused sequence)")
    ap.add_argument("--fused-all", action="store_true",
                    help="all ten perturbed runs in two launches that share the loads of a level and form the sums in "
                         "their epilogue (build extension cloudsc2_nl_taylor_multi)")
    ap.add_argument("--graph", action="store_true",
                    help="capture a run's kernel sequence in a HIP graph and replay it (one host call per run)")
    ap.add_argument("--tune-placement", action="store_true",
                    help="measure and fix the HBM placement of the test's fields for this process (storage.tune_placement)")
    args = ap.parse_args(argv)
    init_distributed_from_env()
    return core(args)


if __name__ == "__main__":
    main()
