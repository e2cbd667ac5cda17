#!/bin/bash
# Everything a round's record needs from ONE GPU lease (run on the GPU box from the repo root):
#   bash profiles/collect_round.sh r03
# GPU tests, the bench at the driver's two (steps, warmup) settings + default, rocprofv3 stats + PMC passes, summaries.
# Two leases since r03 (one call is capped at 20 minutes):  bash profiles/collect_round.sh r03 a   (tests, bench lines, rehearsals)
#                                                           bash profiles/collect_round.sh r03 b   (rocprofv3 passes, summaries)
TAG=${1:-r03}; PART=${2:-all}; D=gpurun_out/${TAG}_collect
mkdir -p $D
if [ "$PART" != "b" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q > $D/pytest_gpu.log 2>&1; tail -3 $D/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $D/bench_20_5.json 2> $D/bench_20_5.err
python bench.py --steps 100 --warmup 20 > $D/bench_100_20.json 2> $D/bench_100_20.err
python bench.py > $D/bench_default.json 2> $D/bench_default.err
python bench.py --config 3 --steps 20 --warmup 5 > $D/bench_config3.json 2> $D/bench_config3.err
python bench.py --config 4 --steps 20 --warmup 5 > $D/bench_config4.json 2> $D/bench_config4.err
# the N-rank path with REAL kernels on this box's one GPU (gloo carries barrier + reductions; at most 6 processes may have the GPU open, the launcher included: 4 ranks)
python bench.py --gpus 4 --collective gloo --steps 20 --warmup 5 --cpu-cols 2048 --cpu-budget-s 1 --no-extra-rooflines > $D/bench_rehearsal_4ranks_one_gpu.json 2> $D/bench_rehearsal_4ranks.err
# (the 8-rank dry run needs no GPU and is taken in the build container: `python bench.py --gpus 8 --dry-run`; on a GPU box
#  its nine processes map the HIP runtime and can trip the pool's limit of 6 processes per GPU)
fi
[ "$PART" = "a" ] && exit 0
bash profiles/run_rocprof_configs34.sh $TAG > $D/rocprof_configs34.log 2>&1
bash profiles/run_rocprof.sh $TAG > $D/rocprof.log 2>&1
bash profiles/run_rocprof_all.sh $TAG 65536 double >> $D/rocprof.log 2>&1
bash profiles/run_rocprof_all.sh $TAG 524288 single >> $D/rocprof.log 2>&1
python profiles/summarize.py $TAG $D/summary > $D/summary.txt 2>&1
tail -40 $D/summary.txt
# the raw rocprofv3 output (>100 MB of CSV) stays on the box unless KEEP_RAW=1: gpurun merges back at most 64 MiB
for C in 3 4; do cp gpurun_out/prof_${TAG}_c$C/kernel_stats_cs2.csv $D/summary/config${C}_kernel_stats.csv; cp gpurun_out/prof_${TAG}_c$C/bench_under_rocprof.json $D/summary/config${C}_bench_under_rocprof.json; done 2>/dev/null
[ -n "$KEEP_RAW" ] || rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_all_${TAG}_* gpurun_out/prof_${TAG}_c3/trace gpurun_out/prof_${TAG}_c4/trace
