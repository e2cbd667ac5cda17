#!/bin/bash
# Everything a round's record needs from ONE GPU lease (run on the GPU box from the repo root):
#   bash profiles/collect_round.sh r02
# GPU tests, the bench at the driver's two (steps, warmup) settings + default, rocprofv3 stats + PMC passes, summaries.
TAG=${1:-r02}; D=gpurun_out/${TAG}_collect
mkdir -p $D
timeout -k 10 900 python -m pytest tests -m gpu -q > $D/pytest_gpu.log 2>&1; tail -3 $D/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $D/bench_20_5.json 2> $D/bench_20_5.err
python bench.py --steps 100 --warmup 20 > $D/bench_100_20.json 2> $D/bench_100_20.err
python bench.py > $D/bench_default.json 2> $D/bench_default.err
bash profiles/run_rocprof.sh $TAG > $D/rocprof.log 2>&1
bash profiles/run_rocprof_all.sh $TAG 65536 double >> $D/rocprof.log 2>&1
bash profiles/run_rocprof_all.sh $TAG 524288 single >> $D/rocprof.log 2>&1
python profiles/summarize.py $TAG $D/summary > $D/summary.txt 2>&1
tail -40 $D/summary.txt
# the raw rocprofv3 output (>100 MB of CSV) stays on the box unless KEEP_RAW=1: gpurun merges back at most 64 MiB
[ -n "$KEEP_RAW" ] || rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_all_${TAG}_*
