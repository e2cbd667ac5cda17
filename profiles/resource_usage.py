#!/usr/bin/env python3
"""Register / LDS / scratch usage of every kernel instantiation, from hipcc's own remarks (no GPU needed):
  python profiles/resource_usage.py [cloudsc2_tl.hip ...] [-- -DCS2_SOME_SWITCH=1]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gt4py_dwarf_p_cloudsc2_tl_ad_amd", "csrc")
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
files = args or ["cloudsc2_nl.hip", "cloudsc2_tl.hip", "cloudsc2_ad.hip", "cloudsc2_aux.hip"]
for f in files:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", f,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra, cwd=CSRC, capture_output=True, text=True)
    cur = {}
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void ", "", name)
            name = name[:name.index("(")] if "(" in name else name
            cur = {"name": name}
        else:
            cur[k] = v
        if k.startswith("LDS Size"):
            print(f"{cur['name']:70s} SGPR {cur.get('TotalSGPRs'):>4} VGPR {cur.get('VGPRs'):>4} AGPR {cur.get('AGPRs'):>4} "
                  f"scratch {cur.get('ScratchSize [bytes/lane]'):>4} occ {cur.get('Occupancy [waves/SIMD]'):>2} LDS {v}")
    if p.returncode:
        sys.stderr.write(p.stderr[-3000:])
        sys.exit(p.returncode)
