#!/bin/bash
# rocprofv3 kernel-trace of every stencil kernel at the BASELINE size (run on the GPU box from the repo root).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_all_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/bench_kernels.py > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 profiles/bench_kernels.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 profiles/bench_kernels.py > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 profiles/bench_kernels.py > $OUT/pmc_sq.log 2>&1
tail -8 $OUT/trace.log
