#!/bin/bash
# rocprofv3 kernel trace + PMC passes of every stencil kernel (run on the GPU box from the repo root):
#   bash profiles/run_rocprof_all.sh <tag> [cols] [precision]     e.g.  r02 65536 double   /   r02 524288 single
set -u
TAG=${1:-r02}; COLS=${2:-65536}; PREC=${3:-double}
OUT=gpurun_out/prof_all_${TAG}_${PREC}_${COLS}
mkdir -p $OUT
export TMPDIR=/tmp
A="profiles/bench_kernels.py --cols=$COLS --precision=$PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $A > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $A > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $A > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $A > $OUT/pmc_sq.log 2>&1
tail -12 $OUT/trace.log
