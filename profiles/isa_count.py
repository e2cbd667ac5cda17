#!/usr/bin/env python3
"""Instruction mix of the level loops of one compiled kernel, from `hipcc -S` output (no GPU needed):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S csrc/cloudsc2_ad.hip -o /tmp/ad.s   (+ the file's Makefile flags)
    python profiles/isa_count.py /tmp/ad.s _ZN3cs29ad_kernelIfLb0ELb0ELb0ELb0E
prints every loop of >= 300 lines (the level loops) with its VALU / SALU / vector-memory / LDS instruction counts and the 40
most frequent opcodes.  This is how round 4 found that 16-17 % of a TL / AD level's VALU instructions were `v_readlane_b32`
fetching spilled field pointers back (docs/TUNING_LOG.md 3.11)."""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
# find function
start = None
for i,l in enumerate(src):
    if l.startswith(want) and ": ; @" in l:
        start = i
        break
assert start is not None
end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
body = src[start:end+1]
# find labels and backward branches -> loops
labels = {}
for i,l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i,l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i))
print("function lines", len(body), "loops", [(a,b,b-a) for a,b in loops])
for a,b in loops:
    if b-a < 300: continue
    c = collections.Counter()
    for l in body[a:b+1]:
        l = l.strip()
        if not l or l.startswith(('.', ';', '//')) or l.endswith(':'): continue
        op = l.split()[0]
        c[op] += 1
    tot = sum(c.values())
    valu = sum(v for k,v in c.items() if k.startswith('v_'))
    print(f"loop {a}-{b}: {tot} instrs, VALU {valu}, SALU {sum(v for k,v in c.items() if k.startswith('s_'))}, vmem {sum(v for k,v in c.items() if k.startswith(('global_','buffer_')))}, ds {sum(v for k,v in c.items() if k.startswith('ds_'))}")
    for k,v in c.most_common(40): print(f"   {k:28s}{v}")
