mkdir -p gpurun_out/r02x
for cfg in "20 5 --no-extra-rooflines" "100 20 --no-extra-rooflines" "20 5" "100 20" "20 5 --no-extra-rooflines" "100 20 --no-extra-rooflines"; do
set -- $cfg
python bench.py --steps $1 --warmup $2 $3 --cpu-cols 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$cfg', '%.1f M  %.4f ms/step  tuned %.4f  NLev %.4f'%(d['value']/1e6,d['ms_per_step'],d['placement']['tuned_ms'],d['roofline']['avg_launch_ms']))
"
done
