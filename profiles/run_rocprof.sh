#!/bin/bash
# Profiling recipe for the headline bench (run on the GPU box from the repo root):
#   bash profiles/run_rocprof.sh <tag>
# pass 1: kernel trace + stats of THE bench command; passes 2-4: PMC counters (own runs, no other trace domains).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# pass 1 profiles THE bench command (default steps / warmup, HIP-event legs included; only the CPU baseline is skipped), so
# each kernel's average duration in the stats can be held against the `roofline*` entries of the JSON line that the same
# profiled run prints into trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-cols 0 > $OUT/trace.log 2>&1
ARGS="bench.py --steps 20 --warmup 3 --cpu-cols 0 --no-roofline-events"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
