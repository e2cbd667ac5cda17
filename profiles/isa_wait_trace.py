"""Compact trace of one compiled kernel's memory operations and waits (from `hipcc -S` output):
    python profiles/isa_wait_trace.py /tmp/isa/tl.s tl_kernelIdLb1ELb0ELb0E
prints, in program order, labels, branches, runs of global loads (Ln) / stores (Sn) / LDS-DMAs (Dn) and every
`s_waitcnt vmcnt(N)` (WN).  A `W0` behind a level's stores means the wave drains its stores before the next level's
loads can be issued: what docs/TUNING_LOG.md 3.9 is about."""
import re
import sys


def trace(path, key, loops_only=True):
    L = open(path).read().split("\n")
    start = next(i for i, l in enumerate(L) if l.startswith("_ZN3cs2") and key in l.split(":")[0])
    end = next(i for i in range(start, len(L)) if ".end_amdhsa_kernel" in L[i])
    ev, cnt = [], {"L": 0, "S": 0, "D": 0}

    def flush():
        s = " ".join(f"{k}{v}" for k, v in cnt.items() if v)
        if s:
            ev.append(s)
        for k in cnt:
            cnt[k] = 0

    for l in L[start + 1:end]:
        s = l.strip()
        if re.match(r"\.LBB\d+_\d+:", s):
            flush()
            ev.append("\n" + s.split(":")[0] + (" (loop header)" if "Loop Header" in s else "") + ":")
        elif s.startswith("global_load_lds"):
            cnt["D"] += 1
        elif s.startswith("global_load"):
            cnt["L"] += 1
        elif s.startswith("global_store"):
            cnt["S"] += 1
        elif s.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m:
                flush()
                ev.append("W" + m.group(1))
        elif re.match(r"s_c?branch", s):
            flush()
            ev.append(s.split()[0][2:] + "->" + s.split()[1])
    flush()
    meta = [l.strip() for l in L[end - 60:end + 40] if re.search(r"(NumVgprs|NumAgprs|ScratchSize|Occupancy):", l)]
    return " ".join(ev), meta


if __name__ == "__main__":
    t, meta = trace(sys.argv[1], sys.argv[2])
    print(t)
    print("\n".join(meta))
