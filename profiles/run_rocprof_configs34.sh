#!/bin/bash
# Kernel trace + stats of `bench.py --config 3` (Taylor test) and `--config 4` (symmetry test), variants included, so that the
# per-kernel average durations of the step's stencil sequence (and of cs2::nl_taylor_multi_kernel, cs2::field_sums_kernel,
# cs2::tl_kernel<inc>) can be held against the record the same profiled run prints.  Run on the GPU box from the repo root:
#   bash profiles/run_rocprof_configs34.sh <tag>
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
for C in 3 4; do
  OUT=gpurun_out/prof_${TAG}_c$C
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config $C --steps 20 --warmup 5 > $OUT/trace.log 2>&1
done
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
for c in (3, 4):
    rows = []
    for f in glob.glob(f"gpurun_out/prof_{tag}_c{c}/trace/*/*kernel_stats.csv"):
        rows += [r for r in csv.DictReader(open(f)) if "cs2::" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(f"gpurun_out/prof_{tag}_c{c}/kernel_stats_cs2.csv", "w", newline="") as o:
        if rows:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    print(f"--config {c}:")
    for r in rows:
        print(f"  {r['Name'].split('(')[0][:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
    line = [l for l in open(f"gpurun_out/prof_{tag}_c{c}/trace.log") if l.startswith('{"metric"')]
    if line:
        open(f"gpurun_out/prof_{tag}_c{c}/bench_under_rocprof.json", "w").write(line[-1])
PY
