"""AD symmetry-test driver: counterpart of /root/reference/drivers/run_symmetry_test.py:41-194."""
from __future__ import annotations

import argparse
import statistics

from ..framework.timing import timing
from ..harness import SymmetryTest
from ._common import add_common_options, init_distributed_from_env, report_placement, setup, tune_field_placement


def core(args):
    ctx = setup(args)
    cfg, p = ctx["config"], ctx["params"]
    st = SymmetryTest(ctx["grid"], factor=0.01, kflag=1, lphylin=True, ldrain1d=False, yoethf_params=p["yoethf"],
                      yomcst_params=p["yomcst"], yrecldp_params=p["yrecldp"], yrephli_params=p["yrephli"],
                      yrncl_params=p["yrncl"], yrphnc_params=p["yrphnc"], enable_checks=cfg.sympl_enable_checks,
                      gt4py_config=cfg.gt4py_config, ad_traj_fix=args.ad_traj_fix, graph=args.graph, fused=args.fused)
    ok = st(ctx["state"], ctx["dt"], enable_validation=cfg.enable_validation)   # warm-up + the validated call
    if args.output_csv_file_stencils is not None and not args.graph:
        cfg.gt4py_config.reset_exec_info()                    # run_symmetry_test.py:92: per-stencil HIP events from here on
    if args.tune_placement:
        # build extension (docs/TUNING_LOG.md 3.7): the ~80 fields of the test are re-placed in HBM where a whole run is fastest
        # (a captured HIP graph holds the OLD field addresses: candidates are timed eagerly, the graph is captured afterwards)
        graph, st.graph, st._graphed = st.graph, False, None
        try:
            ctx["placement"] = tune_field_placement(
                [ctx["state"], st.diags_sat, st.state_i, st.tends_tl, st.diags_tl, st.tends_ad, st.diags_ad],
                lambda: st(ctx["state"], ctx["dt"], enable_validation=False), budget_s=8.0)
        finally:
            st.graph = graph
        report_placement(ctx["placement"], "run")
    from ..framework.timing import Timer

    if args.graph:      # the capture (a warm-up on the capture stream + the recording) happens on the first call: not timed
        st(ctx["state"], ctx["dt"], enable_validation=False)
    Timer.reset()
    for i in range(cfg.num_runs):      # one HIP-event bracket per run (run_symmetry_test.py:94-98); read after the loop
        with timing(f"run_{i}"):
            st(ctx["state"], ctx["dt"], enable_validation=False)
    runtimes = [Timer.get_time(f"run_{i}", units="ms") for i in range(cfg.num_runs)]
    mean = statistics.fmean(runtimes)
    std = statistics.stdev(runtimes) if len(runtimes) > 1 else 0.0
    print(f"\nThe test completed in {mean:.3f} ± {std:.3f} ms.")
    io = ctx["io_config"]
    if io.output_csv_file is not None:          # run_symmetry_test.py:106-121 (+ the build's GB/s and roofline columns)
        from ..framework.output import write_performance_to_csv

        seq = (["saturation", "cloudsc2_tl_incremented",
                "cloudsc2_ad_from_trajectory" if st.cloudsc2_ad_from_trajectory is not None else "cloudsc2_ad"] if st.fused
               else ["saturation", "state_increment", "cloudsc2_tl", "cloudsc2_ad"])
        write_performance_to_csv(io.output_csv_file, io.host_name, cfg.precision, "ad-" + cfg.gt4py_config.backend,
                                 ctx["nx"], cfg.num_threads, 1, cfg.num_runs, mean, std, 0, 0, stencils=seq)
    if args.output_csv_file_stencils is not None:      # run_symmetry_test.py, end of main(): one row per stencil from exec_info
        from ..framework.output import write_stencils_performance_to_csv

        write_stencils_performance_to_csv(args.output_csv_file_stencils, ctx["io_config"].host_name, cfg.precision,
                                          "ad-" + cfg.gt4py_config.backend, ctx["nx"], cfg.num_threads, cfg.num_runs,
                                          cfg.gt4py_config.exec_info, key_patterns=["cloudsc", "increment", "saturation"])
    ctx.update(passed=ok, detail=st.last, runtimes_ms=runtimes, harness=st)
    return ctx


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    add_common_options(ap)
    ap.add_argument("--ad-traj-fix", action="store_true",
                    help="use the AD kernel whose freezing tests match NL/TL (build extension, docs/DESIGN_r03_detail.md 3.3)")
    ap.add_argument("--fused", action="store_true",
                    help="timed call: state_increment fused into cloudsc2_tl (build extension cloudsc2_tl_incremented) and "
                         "cloudsc2_ad without its forward sweep, fed with the TL call's fluxes (cloudsc2_ad_from_trajectory)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the timed call (saturation, state_increment, cloudsc2_tl, cloudsc2_ad) in a HIP graph and "
                         "replay it (one host call per run)")
    ap.add_argument("--tune-placement", action="store_true",
                    help="measure and fix the HBM placement of the test's fields for this process (storage.tune_placement)")
    args = ap.parse_args(argv)
    init_distributed_from_env()
    return core(args)


if __name__ == "__main__":
    main()
