#!/bin/bash
# SQ counters of the Taylor-test kernels (one --pmc pass with --kernel-trace only), from `bench.py --config 3`:
#   bash profiles/run_pmc_config3.sh <tag>
set -u
TAG=${1:-r03}
OUT=gpurun_out/pmc_${TAG}_c3
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --config 3 --steps 5 --warmup 2 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --config 3 --steps 5 --warmup 2 > $OUT/pmc_sq2.log 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
pm = {}
for p in sorted(glob.glob(f"{out}/pmc_*/*/*counter_collection.csv")):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if "cs2::" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        pm.setdefault(k, {}).update({c: {"mean_per_dispatch": sum(x) / len(x), "dispatches": len(x)} for c, x in v.items()})
json.dump(pm, open(f"{out}/config3_pmc.json", "w"), indent=1)
for k, v in pm.items():
    g = lambda c: v.get(c, {}).get("mean_per_dispatch", float("nan"))
    print(f"{k[:64]:64s} waves {g('SQ_WAVES'):8.0f} VALU/wave {g('SQ_INSTS_VALU') / g('SQ_WAVES'):9.0f} SALU/wave {g('SQ_INSTS_SALU') / g('SQ_WAVES'):8.0f} "
          f"LDS/wave {g('SQ_INSTS_LDS') / g('SQ_WAVES'):8.0f} VALU-active {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.0%} waiting {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.0%}")
PY
