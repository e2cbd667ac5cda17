#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries kept under profiles/rNN/.
  python profiles/summarize.py <tag> <round-dir>     e.g.  python profiles/summarize.py r02 profiles/r02"""
import collections, csv, glob, json, os, sys

tag, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def stats(pattern, dest):
    rows = []
    for f in glob.glob(pattern):
        rows += [r for r in csv.DictReader(open(f)) if "cs2::" in r["Name"]]
    if rows:
        with open(dest, "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    return rows


def pmc(pattern, dest):
    pm = {}
    for p in sorted(glob.glob(pattern)):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            if "cs2::" in r["Kernel_Name"]:
                d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in d.items():
            pm.setdefault(k, {}).update({c: {"mean_per_dispatch": sum(x) / len(x), "dispatches": len(x)} for c, x in v.items()})
    if pm:
        json.dump(pm, open(dest, "w"), indent=1)
    return pm


def show(pm):
    for k, v in pm.items():
        fs = v.get("FETCH_SIZE", {}).get("mean_per_dispatch", 0) * 1024
        ws = v.get("WRITE_SIZE", {}).get("mean_per_dispatch", 0) * 1024
        line = f"  {k[:56]}: FETCH_SIZE x2 = {2 * fs / 1e9:.3f} GB, WRITE_SIZE = {ws / 1e9:.3f} GB"
        if "SQ_INSTS_VALU" in v and "SQ_WAVE_CYCLES" in v:
            wc, va, wa = (v[c]["mean_per_dispatch"] for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"))
            line += f", VALU/wave {v['SQ_INSTS_VALU']['mean_per_dispatch'] / v['SQ_WAVES']['mean_per_dispatch']:.0f}, VALU-active {va / wc:.0%}, waiting {wa / wc:.0%}"
        print(line)


for d in sorted(glob.glob(f"gpurun_out/prof_all_{tag}_*")):
    sfx = d.split(f"prof_all_{tag}_")[1]                       # e.g. double_65536
    prec, cols = sfx.split("_")
    name = f"all_kernels_{'fp64' if prec == 'double' else 'fp32'}_{cols}"
    for r in stats(f"{d}/trace/*/*kernel_stats.csv", f"{out}/{name}_kernel_stats.csv"):
        print(f"bench_kernels.py {sfx}:", r["Name"].split("(")[0][:56], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
    show(pmc(f"{d}/pmc_*/*/*counter_collection.csv", f"{out}/{name}_pmc.json"))
for r in stats(f"gpurun_out/prof_{tag}/trace/*/*kernel_stats.csv", f"{out}/bench_fp64_65536_kernel_stats.csv"):
    print("bench.py        :", r["Name"].split("(")[0][:56], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
show(pmc(f"gpurun_out/prof_{tag}/pmc_*/*/*counter_collection.csv", f"{out}/nl_fp64_65536_pmc.json"))

def window_average(trace_glob, first, last, needle):
    """mean duration (us) of launches [first, last] (0-based, in time order) of the kernel whose name contains `needle`,
    from the per-dispatch kernel trace - the launches bench.py's event-timed pass covers (`roofline.launch_window`)"""
    rows = []
    for f in glob.glob(trace_glob):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    sel = rows[first:last + 1]
    return (sum(e - b for b, e in sel) / len(sel) / 1e3, len(sel), len(rows)) if sel else (None, 0, len(rows))


# the bench line printed by the profiled run itself (pass 1), for the stats-vs-events comparison
try:
    for line in open(f"gpurun_out/prof_{tag}/trace.log", errors="replace"):
        if line.startswith('{"metric"'):
            d = json.loads(line)
            json.dump(d, open(f"{out}/bench_under_rocprof.json", "w"))
            for k in ("roofline", "roofline_tl", "roofline_ad", "roofline_nl_f32"):
                if k in d:
                    print(f"profiled run's own events: {k}: {d[k]['kernel']} avg_launch_ms {d[k]['avg_launch_ms']:.4f}")
            print("ms_per_step", d["ms_per_step"])
            win = d["roofline"].get("launch_window")
            if win:
                prec = "double" if d["dtype"] == "f64" else "float"
                avg, n, tot = window_average(f"gpurun_out/prof_{tag}/trace/*/*kernel_trace.csv", win[0], win[1],
                                             f"nl_ring_kernel<{prec}")
                if avg is None:
                    avg, n, tot = window_average(f"gpurun_out/prof_{tag}/trace/*/*kernel_trace.csv", win[0], win[1],
                                                 f"nl_kernel<{prec}")
                # the fused-saturation instantiation (..., true>) is launched after the event pass, so an index window over
                # all nl_ring_kernel<prec> launches in time order is the unfused ones of the pass
                if avg is not None:
                    msg = (f"kernel trace, launches {win[0]}..{win[1]} of {tot} ({n} launches = the event-timed pass): "
                           f"{avg:.1f} us  vs events {d['roofline']['avg_launch_ms'] * 1e3:.1f} us")
                    print(msg)
                    open(f"{out}/bench_nl_event_window.txt", "w").write(msg + "\n")
except OSError:
    pass
