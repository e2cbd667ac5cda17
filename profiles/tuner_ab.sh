#!/bin/bash
# A/B of the tuner's candidate families over fresh processes of ONE lease: narrow-only grid (CLOUDSC2_TUNE_WIDE=0) vs narrow + wide.
mkdir -p gpurun_out/tuner_ab
for i in 1 2 3 4; do
for w in 0 1; do
CLOUDSC2_TUNE_WIDE=$w python bench.py --steps 100 --warmup 20 --cpu-cols 0 --no-extra-rooflines 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
p=d['placement']
print('wide=$w run $i: %.1f M  %.4f ms/step  NL %.4f (%.3f)  tuned_ms %.4f default_ms %.4f  e=%s st=%s shift=%s MB  cands %s'%(d['value']/1e6,d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],p['tuned_ms'],p['default_ms'],p['extra_spacing_x2MB'],p['stagger_bytes'],p['shift_MB'],p['candidates']))
"
done
done
