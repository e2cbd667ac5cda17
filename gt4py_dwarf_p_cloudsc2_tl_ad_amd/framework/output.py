"""Performance report and CSV output (run_nonlinear.py:121-137, :221-232)."""
from __future__ import annotations

import csv
import os
import statistics
from typing import Any, Dict, Optional, Sequence, Tuple

# Floating-point work per column used for the MFLOPS figure: the build's own count of the NL scheme
# (~1e3 fp64 operation-equivalents per level-point x 137 levels), NOT an upstream value.
FLOPS_PER_COLUMN = 1.0e3 * 137


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
                             runtime_mean, runtime_stddev, mflops_mean, mflops_stddev) -> None:
    _append(path, ["host", "precision", "variant", "num_cols", "num_threads", "nproma", "num_runs",
                   "runtime_mean_ms", "runtime_stddev_ms", "mflops_mean", "mflops_stddev"],
            [host, precision, variant, num_cols, num_threads, nproma, num_runs, runtime_mean, runtime_stddev,
             mflops_mean, mflops_stddev])


def write_stencils_performance_to_csv(path, host, precision, variant, num_cols, num_threads, num_runs,
                                      exec_info: Optional[Dict[str, Any]], key_patterns: Sequence[str]) -> None:
    from ..stencils import finalize_exec_info

    finalize_exec_info(exec_info)
    for name, rec in (exec_info or {}).items():
        if isinstance(rec, dict) and any(p in name for p in key_patterns):
            calls = max(rec.get("ncalls", 0), 1)
            _append(path, ["host", "precision", "variant", "num_cols", "num_threads", "num_runs", "stencil",
                           "ncalls", "mean_ms"],
                    [host, precision, variant, num_cols, num_threads, num_runs, name, rec.get("ncalls", 0),
                     1e3 * rec.get("total_run_time", 0.0) / calls])
