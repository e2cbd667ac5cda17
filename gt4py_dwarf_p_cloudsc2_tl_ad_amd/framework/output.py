"""Performance report and CSV output (run_nonlinear.py:121-137, :221-232)."""
from __future__ import annotations

import csv
import os
import statistics
from typing import Any, Dict, Optional, Sequence, Tuple

# Floating-point work per column used for the MFLOPS figure: the build's own count of the NL scheme
# (~1e3 fp64 operation-equivalents per level-point x 137 levels), NOT an upstream value.
FLOPS_PER_COLUMN = 1.0e3 * 137
# Algorithmic HBM words per column (each input read once, each output written once; SURVEY.md 8a/8d, nz = 137)
WORDS_PER_COLUMN = {"saturation": 411, "state_increment": 4416, "perturbed_state": 6624, "cloudsc2_nl": 3567,
                    "cloudsc2_nl_saturation": 3567, "cloudsc2_nl_perturbed": 5760, "cloudsc2_nl_taylor": 5760,
                    # per CALL with the Taylor test's ten step sizes = two launches of five, increments formed in the kernel:
                    # each launch reads the 16 state + 10 reference fields once
                    "cloudsc2_nl_taylor_multi": 2 * 3567,
                    "cloudsc2_tl": 7134, "cloudsc2_tl_incremented": 4941, "cloudsc2_ad": 7134,
                    # 16 inputs + 10 forcings + 2 fluxes read, 16 adjoints written
                    "cloudsc2_ad_from_trajectory": 2193 + 1374 + 2 * 138 + 2194}
HBM_PEAK_GBS = 8000.0   # MI355X spec peak the roofline columns are quoted against
_WORD = {"double": 8, "single": 4}


def algorithmic_gbs(stencils: Sequence[str], precision: str, num_cols: int, ms: float) -> float:
    words = sum(WORDS_PER_COLUMN.get(s, 0) for s in stencils)
    return words * _WORD.get(precision, 8) * num_cols / (ms * 1e-3) / 1e9 if ms > 0 else 0.0


def print_performance(num_cols: int, runtimes_ms: Sequence[float]) -> Tuple[float, float, float, float]:
    mean = statistics.fmean(runtimes_ms)
    std = statistics.stdev(runtimes_ms) if len(runtimes_ms) > 1 else 0.0
    mflops = [FLOPS_PER_COLUMN * num_cols / (t * 1e-3) / 1e6 for t in runtimes_ms]
    mf_mean = statistics.fmean(mflops)
    mf_std = statistics.stdev(mflops) if len(mflops) > 1 else 0.0
    print(f"Performance: {num_cols} columns, {len(runtimes_ms)} runs: {mean:.3f} +/- {std:.3f} ms per run, "
          f"{num_cols / (mean * 1e-3):.4g} columns/s, ~{mf_mean:.0f} MFLOPS")
    return mean, std, mf_mean, mf_std


def _append(path: str, header: Sequence[str], row: Sequence[Any]) -> None:
    new = not os.path.exists(path)
    with open(path, "a", newline="") as f:
        w = csv.writer(f)
        if new:
            w.writerow(header)
        w.writerow(row)


def write_performance_to_csv(path, host, precision, variant, num_cols, num_threads, nproma, num_runs,
                             runtime_mean, runtime_stddev, mflops_mean, mflops_stddev,
                             stencils: Sequence[str] = ("saturation", "cloudsc2_nl")) -> None:
    """The reference's columns (run_nonlinear.py:123-137) + algorithmic GB/s and % of the HBM roofline of the timed
    region (`stencils` = what one run launches)."""
    gbs = algorithmic_gbs(stencils, precision, num_cols, runtime_mean)
    _append(path, ["host", "precision", "variant", "num_cols", "num_threads", "nproma", "num_runs",
                   "runtime_mean_ms", "runtime_stddev_ms", "mflops_mean", "mflops_stddev", "columns_per_s",
                   "algorithmic_GBs", "pct_hbm_roofline"],
            [host, precision, variant, num_cols, num_threads, nproma, num_runs, runtime_mean, runtime_stddev,
             mflops_mean, mflops_stddev, num_cols / (runtime_mean * 1e-3), gbs, 100.0 * gbs / HBM_PEAK_GBS])


def write_stencils_performance_to_csv(path, host, precision, variant, num_cols, num_threads, num_runs,
                                      exec_info: Optional[Dict[str, Any]], key_patterns: Sequence[str]) -> None:
    from ..stencils import finalize_exec_info

    finalize_exec_info(exec_info)
    for name, rec in (exec_info or {}).items():
        if isinstance(rec, dict) and any(p in name for p in key_patterns):
            calls = max(rec.get("ncalls", 0), 1)
            mean_ms = 1e3 * rec.get("total_run_time", 0.0) / calls
            gbs = algorithmic_gbs([name], precision, num_cols, mean_ms)
            _append(path, ["host", "precision", "variant", "num_cols", "num_threads", "num_runs", "stencil",
                           "ncalls", "mean_ms", "algorithmic_GBs", "pct_hbm_roofline"],
                    [host, precision, variant, num_cols, num_threads, num_runs, name, rec.get("ncalls", 0),
                     mean_ms, gbs, 100.0 * gbs / HBM_PEAK_GBS])
