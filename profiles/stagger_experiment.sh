#!/bin/bash
# Field-placement experiment (one process per kernel, layouts interleaved): separate torch allocations vs one arena with
# different staggers / level-stride paddings.   bash profiles/stagger_experiment.sh   (on the GPU box)
mkdir -p gpurun_out/r02d; L=base=build/variants/lib_base.so; O=gpurun_out/r02d/layouts2.txt; : > $O
for k in nl tl ad; do
  python profiles/ab_kernels.py $k $L --rounds=10 --layouts=separate,arena:1048576,arena:3145728,arena:5242880,arena:1052672,arena:1114112,arena:524288,separate 2>&1 | grep -v "Warning\|amdgpu.ids\|^ *print" >> $O || exit 1
done
cat $O
