#!/bin/bash
# A/B of the tuner's shift range over fresh processes of ONE lease: default grid (shifts 0/4/8/12 GB) vs shifts 0..32 GB in 4 GB steps
# (bench.py --tune-shifts-mb, which passes shifts_mb / max_arena_bytes to storage.tune_placement).  Usage: bash profiles/tuner_ab.sh [extra bench args]
mkdir -p gpurun_out/tuner_ab
WIDE=0,4096,8192,12288,16384,20480,24576,28672,32768
for i in 1 2 3 4; do
for w in "" $WIDE; do
python bench.py --steps 100 --warmup 20 --cpu-cols 0 --no-extra-rooflines ${w:+--tune-shifts-mb $w} "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
p=d['placement']
print('shifts=%-8s run $i: %.1f M  %.4f ms/step  NL %.4f (%.3f)  tuned_ms %.4f default_ms %.4f  e=%s st=%s shift=%s MB  cands %s'%('${w:0:8}' or 'default',d['value']/1e6,d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],p['tuned_ms'],p['default_ms'],p['extra_spacing_x2MB'],p['stagger_bytes'],p['shift_MB'],p['candidates']))
"
done
done
