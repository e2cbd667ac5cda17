"""The hand-counted `s_waitcnt vmcnt(N)` of the cloudsc2_tl LDS-ring kernel is only right if the compiled level loop
issues exactly the operations the count assumes: NI LDS-DMAs and 20 stores per level, and no wait hipcc added on its own.
This test compiles csrc/cloudsc2_tl.hip to gfx950 assembly (no GPU needed) and checks exactly that, so a compiler or source
change that breaks the count fails here and not as silent data corruption on the GPU.  (nl_ring_kernel's wait is one level of
stores short of exact by design and hipcc multiplies its loop body per load-policy path, so a static count says nothing
there; its guard is the determinism soak and the ring-vs-register tests on the GPU.)"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gt4py_dwarf_p_cloudsc2_tl_ad_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _compile(tmp_path_factory, src):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available on this machine (the prebuilt library travelled with the snapshot)")
    out = tmp_path_factory.mktemp("isa") / (src + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src,
                    "-o", str(out)], cwd=CSRC, check=True, capture_output=True)
    return out.read_text()


@pytest.fixture(scope="module")
def tl_asm(tmp_path_factory):
    return _compile(tmp_path_factory, "cloudsc2_tl.hip")


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


@pytest.mark.parametrize("tname,ni", [("d", 16), ("f", 8)])
def test_tl_ring_loop_matches_the_hand_counted_wait(tl_asm, tname, ni):
    expected = ni + 20            # two slots per wave: (RD-1) x (NI DMAs + 20 stores)
    seen = 0
    for name, lines in _kernels(tl_asm, "tl_ring_kernelI" + tname):
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
    assert seen == 4              # REG x EVAP instantiations

