#!/bin/bash
# One lease: the whole GPU suite, the fp32 verdict probe, the multi-step Taylor kernel A/B, the three bench lines (configs 3, 4, 2).
D=gpurun_out/${1:-r03c}; mkdir -p $D
timeout -k 10 800 python -m pytest tests -m gpu -q > $D/pytest_gpu.log 2>&1; tail -15 $D/pytest_gpu.log
timeout -k 10 300 python profiles/probe_fp32_verdicts.py > $D/fp32_verdicts.txt 2>&1; grep "@@" $D/fp32_verdicts.txt
if ls build/variants/lib_nf4.so > /dev/null 2>&1; then
timeout -k 10 200 python profiles/ab_taylor_multi.py nf5=gt4py_dwarf_p_cloudsc2_tl_ad_amd/libcloudsc2_hip.so nf4=build/variants/lib_nf4.so nf3=build/variants/lib_nf3.so nf2=build/variants/lib_nf2.so > $D/ab_taylor_multi.txt 2>&1; tail -8 $D/ab_taylor_multi.txt
fi
python bench.py --config 3 --steps 20 --warmup 5 > $D/bench_c3.json 2> $D/bench_c3.err
python bench.py --config 4 --steps 20 --warmup 5 > $D/bench_c4.json 2> $D/bench_c4.err
python bench.py --steps 20 --warmup 5 > $D/bench_c2.json 2> $D/bench_c2.err
python - "$D" <<PY
import json, sys
D = sys.argv[1]
for c in (3, 4):
    try:
        d = json.load(open(f"{D}/bench_c{c}.json"))
    except Exception as e:
        print(c, "no json", e); continue
    print(c, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["verdict"].get("verdict"), d["roofline"]["device_ms_one_step"])
    for k, v in d["variants"].items(): print("   ", k, v.get("ms_per_step"), v.get("frac_of_8TBs"), v.get("verdict"), v.get("error"))
d = json.load(open(f"{D}/bench_c2.json"))
print(2, d["value"], d["ms_per_step"], d.get("value_default_placement"), d.get("ms_per_step_default_placement"), d["roofline"]["frac"], d["roofline_tl"]["frac"], d["roofline_ad"]["frac"], d["roofline_nl_f32"]["frac"], d["placement"].get("arena_GB"), d["placement"].get("tuning_s"), d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores"))
PY
