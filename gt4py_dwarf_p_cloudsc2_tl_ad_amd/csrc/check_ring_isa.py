"""Build-time guard of the compiled level loops: the hand-counted `s_waitcnt vmcnt(N)` of the LDS-ring kernels
(cloudsc2_nl.hip, cloudsc2_tl.hip) and the prefetch distance of the register-path kernels.

The ring kernels wait for "all but the N youngest vector-memory operations"; N is counted by hand from what the level loop
issues (NI LDS-DMAs + the level's stores).  That count is only right for the code hipcc actually emitted, so it is checked
on the compiled gfx950 assembly (no GPU needed):

  * tl_ring_kernel (two slots per wave, EXACT wait `vmcnt(NI + 20)`): the loop body around the wait holds exactly NI
    `global_load_lds_dwordx4`, 20 `global_store`, no ordinary load, and no vector-memory wait hipcc added on its own;
  * nl_ring_kernel (three / two slots, wait one level of stores stricter than exact): hipcc multiplies the level body per
    cache-policy path (the in_qsat DMA, the optional pre-scan DMA), so lines cannot simply be counted; the kernel's
    control-flow graph is walked instead: on EVERY static path from one steady-state ring wait to the next there are
    >= NSTORE stores, the paths that issue any DMA issue >= NI of them, there is no ordinary load and no vector-memory
    wait other than the three hand-written ones (NFULL, NHEAD, 0).  (RAGGED instantiations: the exec-masked stores of
    a partly filled last wave are counted as issued - see `_cfg`.)

  * register-path kernels (tl_kernel, nl_kernel, nl_taylor_multi_kernel, ad_kernel; `check_prefetch_distance`): in every
    level loop the first wait that reaches into the batch of prefetch loads for the next level comes at least 60
    instructions behind the batch - i.e. nothing consumes a prefetched word at the load site.

Used by tests/test_ring_isa.py (CPU suite) and by `__graft_entry__.build()` whenever it really recompiles the library, so a
library built by a different hipcc cannot ship with a wrong count.  `python check_ring_isa.py` runs it by hand."""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.dirname(os.path.abspath(__file__))
HIPCC = "/opt/rocm/bin/hipcc"


def compile_flags(src: str):
    """the flags the Makefile compiles `src` with (`make -n -B <object>`): the check must read the code generation that
    ships, per-file switches included (cloudsc2_tl / cloudsc2_ad are built with -fno-slp-vectorize)"""
    obj = os.path.splitext(src)[0] + ".o"
    p = subprocess.run(["make", "-n", "-B", obj], cwd=CSRC, check=True, capture_output=True, text=True)
    line = next(l for l in p.stdout.splitlines() if " -c " in l and src in l)
    words = line.split()
    flags, skip = [], False
    for w in words[1:]:
        if skip:
            skip = False
        elif w in ("-c", src):
            continue
        elif w == "-o":
            skip = True
        elif w not in ("-fPIC",):
            flags.append(w)
    return flags


def compile_to_asm(src: str, out_dir: str) -> str:
    out = os.path.join(out_dir, src + ".s")
    subprocess.run([HIPCC] + compile_flags(src) + ["--cuda-device-only", "-S", src, "-o", out],
                   cwd=CSRC, check=True, capture_output=True)
    with open(out) as fh:
        return fh.read()


def _kernels(asm, prefix):
    for m in re.finditer(r"^(_ZN3cs2\w+):", asm, flags=re.M):
        name = m.group(1)
        if prefix in name:
            end = asm.index(".end_amdhsa_kernel", m.end()) if ".end_amdhsa_kernel" in asm[m.end():] else len(asm)
            yield name, asm[m.end():end].split("\n")


def _innermost_loop_around(lines, idx, outermost=False):
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"(\.LBB\d+_\d+):", l.strip()))}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"\s*s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= idx <= i:
            smaller = best is None or (i - labels[m.group(1)]) < (best[1] - best[0])
            if best is None or smaller != outermost:
                best = (labels[m.group(1)], i)
    return best


def check_tl_ring(asm):
    """Raises AssertionError when a compiled tl_ring_kernel does not match its hand-counted wait; returns the number of
    instantiations checked."""
    seen = 0
    for tname, ni in (("d", 16), ("f", 8)):
        expected = ni + 20            # two slots per wave: (RD-1) x (NI DMAs + 20 stores)
        for name, lines in _kernels(asm, "tl_ring_kernelI" + tname):
            waits = [i for i, l in enumerate(lines) if re.search(rf"s_waitcnt vmcnt\({expected}\)\s*$", l)]
            assert len(waits) == 1, (name, "steady-state wait not found exactly once", len(waits))
            lo, hi = _innermost_loop_around(lines, waits[0])
            body = lines[lo:hi + 1]
            stores = sum("global_store" in l for l in body)
            dmas = sum("global_load_lds_dwordx4" in l for l in body)
            plain_loads = sum(bool(re.search(r"global_load_dword", l)) for l in body)
            assert stores == 20 and dmas == ni and plain_loads == 0, (name, stores, dmas, plain_loads)
            vm_waits = [re.search(r"vmcnt\((\d+)\)", l).group(1) for l in body if "s_waitcnt" in l and "vmcnt" in l]
            # the loop's only vector-memory waits: the counted one, the tail's drain (0), the second half's no-op (63) and -
            # unless hipcc peeled the first iteration - the head's (NI: no stores counted yet)
            assert set(vm_waits) - {str(ni)} == {"0", "63", str(expected)} and vm_waits.count("0") == 1, (name, vm_waits)
            seen += 1
    return seen


def _cfg(lines, execz_never_taken=False):
    """Nodes of a kernel's control-flow graph, split at labels, branches and vector-memory waits.  Every node is
    (kind counts, wait value or None, successors).  `execz_never_taken`: drop the taken edge of `s_cbranch_execz` (the
    skip around an exec-masked region) - for the RAGGED ring kernels, whose stores sit behind `if (live)`: a wave with
    no live lane has retired at the top of the kernel, so inside the level loop that branch is never taken and the masked
    stores are always issued."""
    starts = {0}
    label_at = {}
    for i, l in enumerate(lines):
        t = l.strip()
        if (m := re.match(r"(\.LBB\d+_\d+):", t)):
            starts.add(i)
            label_at[m.group(1)] = i
        if re.match(r"s_c?branch|s_endpgm", t):
            starts.add(i + 1)
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            starts.add(i)
            starts.add(i + 1)
    order = sorted(x for x in starts if x < len(lines))
    nodes = {}
    for a, b in zip(order, order[1:] + [len(lines)]):
        body = [l.strip() for l in lines[a:b]]
        wait = None
        if body and body[0].startswith("s_waitcnt") and "vmcnt" in body[0]:
            wait = int(re.search(r"vmcnt\((\d+)\)", body[0]).group(1))
        succ = []
        last = next((t for t in reversed(body) if t and not t.startswith(";")), "")
        m = re.match(r"s_(c?)branch\w* (\.LBB\d+_\d+)", last)
        if m:
            if not (execz_never_taken and last.startswith("s_cbranch_execz")):
                succ.append(label_at[m.group(2)])
            if m.group(1):
                succ.append(b)
        elif not last.startswith("s_endpgm"):
            succ.append(b)
        # the hand-written ring waits are the ones fused with the slot's LDS reads in one asm statement
        ring_wait = wait is not None and b < len(lines) and lines[b].strip().startswith("ds_read_b")
        nodes[a] = dict(ring_wait=ring_wait, dma=sum("global_load_lds" in t for t in body), st=sum(t.startswith("global_store") for t in body),
                        ld=sum(bool(re.match(r"global_load_dword", t)) for t in body), wait=wait,
                        succ=[x for x in succ if x < len(lines)])
    return nodes


def _between_waits(nodes, start, nfull):
    """Over all static paths from the ring wait `start` to the next steady-state ring wait (paths end at ANY vector-memory
    wait; only those ending at a vmcnt(NFULL) ring wait count): min / max stores, max ordinary loads, max LDS-DMAs, the min
    of LDS-DMAs over the paths that issue at least one (the static graph also contains the infeasible combination
    "`more` false at the issue, true at the wait" - the two tests are the same uniform condition - which issues none), and
    the set of wait values met."""
    import sys

    sys.setrecursionlimit(20000)
    INF = 10 ** 9
    seen_waits = set()
    memo = {}
    BUSY = object()

    def walk(n):
        if n in memo:
            # a node still being expanded = a static cycle through the level loop that bypasses all three waits (the same
            # infeasible combination of the uniform `more` tests): ignored
            return None if memo[n] is BUSY else memo[n]
        nd = nodes[n]
        if nd["wait"] is not None:
            seen_waits.add(nd["wait"])
            memo[n] = dict(st=(0, 0), ld=0, dma_max=0, dma_min_any=0, dma_min_some=INF) \
                if (nd["wait"] == nfull and nd["ring_wait"]) else None
            return memo[n]
        memo[n] = BUSY
        res = [r for r in (walk(s) for s in nd["succ"]) if r is not None]
        if not res:
            memo[n] = None
            return None
        d = nd["dma"]
        any_min = min(r["dma_min_any"] for r in res)
        memo[n] = dict(st=(nd["st"] + min(r["st"][0] for r in res), nd["st"] + max(r["st"][1] for r in res)),
                       ld=nd["ld"] + max(r["ld"] for r in res), dma_max=d + max(r["dma_max"] for r in res),
                       dma_min_any=d + any_min,
                       dma_min_some=(d + any_min) if d > 0 else min(r["dma_min_some"] for r in res))
        return memo[n]

    res = [r for r in (walk(s) for s in nodes[start]["succ"]) if r is not None]
    assert res, "the steady-state wait is not inside a loop"
    return dict(st=(min(r["st"][0] for r in res), max(r["st"][1] for r in res)), ld=max(r["ld"] for r in res),
                dma_max=max(r["dma_max"] for r in res), dma_min_some=min(r["dma_min_some"] for r in res),
                waits=seen_waits)


def check_nl_ring(asm):
    """Raises AssertionError when a compiled nl_ring_kernel does not issue what its hand-counted waits assume.  Returns the
    number of instantiations checked."""
    seen = 0
    for name, lines in _kernels(asm, "nl_ring_kernelI"):
        m = re.search(r"nl_ring_kernelI([df])Lb[01]ELb[01]ELb[01]ELi(\d)ELb([01])ELb([01])E", name)
        t, rd, satf, ragged = m.group(1), int(m.group(2)), m.group(3) == "1", m.group(4) == "1"
        ni = 8 if t == "d" else 4
        nstore = 10
        nfull, nhead = (rd - 1) * ni + (rd - 2) * nstore, (rd - 1) * ni
        if ragged:
            nfull = nhead       # the ragged instantiations count no stores at all: their wait is independent of how the
                                # exec-masked stores of a level were lowered
        nodes = _cfg(lines, execz_never_taken=ragged)
        steady = [n for n, nd in nodes.items() if nd["wait"] == nfull and nd["ring_wait"]]
        assert steady, (name, f"no s_waitcnt vmcnt({nfull})")
        for n in steady:
            r = _between_waits(nodes, n, nfull)
            want_st = nstore + (1 if satf else 0)
            # NI DMAs per level, + 1 for the pre-scan pair, + 1 static only: the default-policy / nt alternatives of the
            # in_qsat DMA, which hipcc lays out as a fall-through behind an always-taken s_cbranch_execnz
            assert ni <= r["dma_min_some"] and r["dma_max"] <= ni + 2, (name, "LDS-DMAs per level", r)
            if ragged:
                # the level's stores sit behind `if (live)`: hipcc lowers that to exec-masked regions that a wave WITHOUT a
                # live lane would skip (s_cbranch_execz / execnz in several shapes) - and such a wave has retired at the
                # top of the kernel.  So only the maximum is held statically (all masked regions entered = every store of
                # the level issued); the aligned instantiations above pin the exact count of the same source.
                # (since r04 the ragged wait does not depend on this count; it is still pinned as a description of the code)
                assert r["st"][1] == want_st, (name, "stores per level (ragged)", r)
            else:
                assert r["st"] == (want_st, want_st), (name, "stores per level", r)
            assert r["ld"] == 0, (name, "ordinary loads inside the level loop", r)
            assert r["waits"] <= {0, nhead, nfull}, (name, "vector-memory waits in the level loop", sorted(r["waits"]))
        seen += 1
    return seen


def _is_instr(line):
    t = line.strip()
    return bool(t) and not t.startswith((";", ".", "//")) and not t.endswith(":")


def check_prefetch_distance(asm, prefix, min_instr=60, min_batch=14, skip=None):
    """The register-path kernels (tl_kernel, ad_kernel, nl_kernel, nl_taylor_multi_kernel) request level k+1's words
    before level k is computed.  That only hides latency if nothing WAITS for those words until the level's arithmetic
    is done.  Two things made hipcc wait early in r03 (docs/TUNING_LOG.md 3.9): arithmetic on a loaded word at the load
    site (cloudsc2_ad's flux forcings: `s_waitcnt vmcnt(11)` 10 instructions behind the batch of 26 loads, on every level
    of sweep 2) and a rotating register buffer (cloudsc2_nl, fixed earlier in r03).  Checked on the compiled ISA: inside
    every level loop, the first vector-memory wait in program order that reaches into a batch of >= `min_batch` ordinary
    loads (`vmcnt(N)` with N below the operations issued since the batch began) comes at least `min_instr` instructions
    behind it.  Not checked: nl_kernel<EVAP = true, FUSE = 3> (the opt-in fused-norms kernel with the evaporation block no
    driver enables: at 256 VGPRs hipcc sinks loads to their uses there).  Returns the number of (kernel, batch) pairs checked."""
    seen = 0
    for name, lines in _kernels(asm, prefix):
        if skip and re.search(skip, name):
            continue
        labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"(\.LBB\d+_\d+):", l.strip()))}
        i = 0
        while i < len(lines):
            if "global_load_dword" not in lines[i] or "global_load_lds" in lines[i]:
                i += 1
                continue
            # a batch: loads separated by at most 24 other instructions (address arithmetic, v_readlane of spilled pointers)
            j, last, n, gap = i, i, 0, 0
            while j < len(lines) and gap <= 24:
                if "global_load_dword" in lines[j] and "global_load_lds" not in lines[j]:
                    last, n, gap = j, n + 1, 0
                elif _is_instr(lines[j]):
                    gap += 1
                j += 1
            loop = _innermost_loop_around(lines, last)
            # level loops only (hundreds of instructions of physics): the tropopause pre-scan's short loop consumes what it loads
            if n >= min_batch and loop is not None and loop[1] - loop[0] >= 600:
                # the first wait that reaches INTO the batch: vmcnt(N) with N below the operations issued since its first load
                # (program order: unconditional branches are followed, conditional ones fall through)
                dist, k, issued, hops = 0, last + 1, n, 0
                while k < len(lines):
                    m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", lines[k])
                    if m and int(m.group(1)) < issued:
                        break
                    b = re.match(r"\s*s_branch (\.LBB\d+_\d+)", lines[k])
                    if b and b.group(1) in labels and hops < 64:
                        k, hops = labels[b.group(1)], hops + 1
                        continue
                    issued += "global_load" in lines[k] or "global_store" in lines[k]
                    dist += _is_instr(lines[k])
                    k += 1
                assert k == len(lines) or dist >= min_instr, (
                    name, f"a batch of {n} loads in a level loop is waited for {dist} instructions later", lines[k].strip())
                seen += 1
            i = last + 1
    return seen


def kernel_resources(asm, key):
    """{'NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy'} of the one kernel whose mangled name contains `key`, from the
    metadata comments hipcc writes behind the kernel."""
    start = next(m.start() for m in re.finditer(r"^(_ZN3cs2\w+):", asm, flags=re.M) if key in m.group(1))
    tail = asm[start:asm.index(".end_amdhsa_kernel", start) + 4000]
    return {k: int(re.search(rf"; {k}: (\d+)", tail).group(1)) for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy")}


def check_resources(asm_nl, asm_tl, asm_ad) -> dict:
    """Register budgets the launch heuristics and DESIGN's numbers rest on: no scratch in the kernels the drivers' defaults
    run, and cloudsc2_ad fp32 at three waves per SIMD (what its LDS parking buys; r03: two more live words took it to 170
    VGPRs = two waves without anybody noticing, until the register cap was requested)."""
    out = {}
    for what, asm, key in (("nl_ring f64", asm_nl, "nl_ring_kernelIdLb0ELb1ELb1ELi3ELb0ELb0E"),
                           ("nl_ring f32", asm_nl, "nl_ring_kernelIfLb0ELb1ELb0ELi2ELb0ELb0E"),
                           ("tl f64", asm_tl, "9tl_kernelIdLb1ELb0ELb0ELb0E"), ("tl_ring f32", asm_tl, "tl_ring_kernelIfLb1ELb0E"),
                           ("ad f64", asm_ad, "9ad_kernelIdLb1ELb0ELb0ELb0ELb0E"), ("ad f32", asm_ad, "9ad_kernelIfLb1ELb0ELb0ELb0ELb0E")):
        r = kernel_resources(asm, key)
        assert r["ScratchSize"] == 0, (what, "spills to scratch", r)
        out[what] = r
    assert out["ad f32"]["Occupancy"] >= 3, out["ad f32"]
    return out


_ASM_VMEM = re.compile(r"^\s*((?:global|flat|buffer|scratch)_(?:load|store|atomic)\w*)\s+(.*?)\s*(?:;.*)?$")


def check_inline_asm_vmem(asm: str, where: str = "") -> int:
    """Every vector-memory instruction that comes from an INLINE-ASM block (hipcc brackets those with `;;#ASMSTART` /
    `;;#ASMEND` in its -S output) must address memory as `v_off, s[base:base+1]`: a single 32-bit VGPR byte offset plus the
    field's base pointer in an SGPR pair.  Round 3 lost a GPU (memory access fault on 0x4000 / 0xa000 = the byte offset of
    a workgroup's first column, i.e. an address WITHOUT its base, docs/TUNING_LOG.md 3.10) to a hand-written
    `global_store_dwordx2 ... sc1` whose ISA text nobody had read.  Compiler-generated accesses are not judged here (hipcc
    legitimately uses the `v[a:b], off` form with a full 64-bit per-lane address); `flat_*`, `buffer_*` and `scratch_*`
    have no business in these kernels' asm at all.  Returns the number of inline-asm memory instructions seen (all well
    formed); raises AssertionError on the first one that is not."""
    n, inside = 0, False
    for ln, line in enumerate(asm.split("\n"), 1):
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            inside = True
        elif s.startswith(";;#ASMEND"):
            inside = False
        elif inside:
            m = _ASM_VMEM.match(s)
            if not m:
                continue
            op, operands = m.group(1), [x.strip() for x in m.group(2).split(",")]
            ok = op.startswith("global_") and len(operands) >= 3
            if ok:
                # global_load*  vdst, vaddr, saddr [mods]   /   global_store*  vaddr, vdata, saddr [mods]
                vaddr = operands[1] if op.startswith(("global_load", "global_atomic")) else operands[0]
                saddr = operands[2].split()[0]
                ok = re.fullmatch(r"v\d+", vaddr) is not None and re.fullmatch(r"s\[\d+:\d+\]", saddr) is not None
            assert ok, (f"{where}:{ln}: inline-asm memory instruction `{s}` does not use the `v_off, s[base:base+1]` form - "
                        "read profiles/README.md (rule on hand-written memory instructions) before this goes to a GPU")
            n += 1
    return n


def check_all(out_dir=None) -> dict:
    """Compile both ring sources to assembly and check every instantiation; raises AssertionError on a mismatch."""
    with tempfile.TemporaryDirectory() as tmp:
        d = out_dir or tmp
        asm_tl, asm_nl = compile_to_asm("cloudsc2_tl.hip", d), compile_to_asm("cloudsc2_nl.hip", d)
        asm_ad = compile_to_asm("cloudsc2_ad.hip", d)
    n_tl = check_tl_ring(asm_tl)
    n_nl = check_nl_ring(asm_nl)
    assert n_tl == 8 and n_nl == 64, (n_tl, n_nl)      # T x REG x EVAP; T x EVAP x LIN x depth {3, 2} x SATF x RAGGED
    pf = {"tl_kernel": check_prefetch_distance(asm_tl, "9tl_kernelI"),
          "nl_kernel": check_prefetch_distance(asm_nl, "9nl_kernelI", skip=r"Lb1ELb[01]ELb[01]ELi3E"),
          "nl_taylor_multi_kernel": check_prefetch_distance(asm_nl, "nl_taylor_multi_kernelI"),
          "ad_kernel": check_prefetch_distance(asm_ad, "9ad_kernelI")}
    # the batch detection is a heuristic (loads a few instructions apart, loops of >= 600 lines): it must have seen every
    # instantiation's level loop(s) - first of all the ones the drivers' defaults run (fp64, LREGCL, no evaporation)
    assert check_prefetch_distance(asm_tl, "9tl_kernelIdLb1ELb0ELb0ELb0E") == 1, "cloudsc2_tl fp64 default: level loop not seen"
    assert check_prefetch_distance(asm_ad, "9ad_kernelIdLb1ELb0ELb0ELb0ELb0E") == 2, "cloudsc2_ad fp64 default: two sweeps not seen"
    assert check_prefetch_distance(asm_nl, "9nl_kernelIdLb0ELb1ELb1ELi2ELb0E") == 1, "perturbed cloudsc2_nl fp64: loop not seen"
    # (r04: + the BIG instantiations - 64-bit offsets - of the plain register-path kernels: tl 16 + 8, nl 28 + 8, ad 2 x 32;
    #  + the 8 trajectory instantiations of ad_kernel, one sweep each)
    assert pf["tl_kernel"] >= 24 and pf["nl_kernel"] >= 36 and pf["nl_taylor_multi_kernel"] >= 64 and pf["ad_kernel"] >= 72, pf
    assert check_prefetch_distance(asm_ad, "9ad_kernelIdLb1ELb0ELb0ELb0ELb1E") == 1, "cloudsc2_ad_from_trajectory fp64: sweep not seen"
    res = check_resources(asm_nl, asm_tl, asm_ad)
    n_asm = sum(check_inline_asm_vmem(a, f) for a, f in ((asm_nl, "cloudsc2_nl.hip"), (asm_tl, "cloudsc2_tl.hip"),
                                                         (asm_ad, "cloudsc2_ad.hip")))
    return {"tl_ring_kernel": n_tl, "nl_ring_kernel": n_nl, "register_path_prefetch_batches": pf,
            "ad_f32_occupancy": res["ad f32"]["Occupancy"], "inline_asm_vmem": n_asm}


if __name__ == "__main__":
    print("ring ISA check:", check_all())
    sys.exit(0)
