"""The hand-counted `s_waitcnt vmcnt(N)` of the LDS-ring kernels (cloudsc2_tl and the headline cloudsc2_nl) are only right
if the compiled level loops issue exactly the operations the counts assume.  The checks themselves live beside the sources
(gt4py_dwarf_p_cloudsc2_tl_ad_amd/csrc/check_ring_isa.py - `__graft_entry__.build()` runs them whenever it recompiles);
this file compiles the two sources to gfx950 assembly (no GPU needed) and runs them in the CPU suite, plus two mutation
tests showing that the NL guard really fails on a dropped store and on a foreign wait."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gt4py_dwarf_p_cloudsc2_tl_ad_amd", "csrc"))
import check_ring_isa as isa  # noqa: E402


def _compile(tmp_path_factory, src):
    if not os.path.exists(isa.HIPCC):
        pytest.skip("hipcc not available on this machine (the prebuilt library travelled with the snapshot)")
    return isa.compile_to_asm(src, str(tmp_path_factory.mktemp("isa")))


@pytest.fixture(scope="module")
def tl_asm(tmp_path_factory):
    return _compile(tmp_path_factory, "cloudsc2_tl.hip")


@pytest.fixture(scope="module")
def nl_asm(tmp_path_factory):
    return _compile(tmp_path_factory, "cloudsc2_nl.hip")


def test_tl_ring_loop_matches_the_hand_counted_wait(tl_asm):
    assert isa.check_tl_ring(tl_asm) == 8            # T in {double, float} x REG x EVAP


def test_nl_ring_loop_matches_the_hand_counted_waits(nl_asm):
    assert isa.check_nl_ring(nl_asm) == 64           # T x EVAP x LIN x ring depth {3, 2} x SATF x RAGGED


def test_nl_ring_guard_detects_a_dropped_store_and_a_foreign_wait(nl_asm):
    import re

    k0 = nl_asm.index("nl_ring_kernelIdLb0ELb1ELb1ELi3ELb0ELb0E")      # the headline instantiation (aligned whole waves)
    # the first hand-written ring wait of that kernel (a vmcnt wait fused with the slot's ds_reads) ...
    w = re.compile(r"s_waitcnt vmcnt\(\d+\)\n\s*ds_read_b64").search(nl_asm, k0).start()
    # ... and the last store before it in the listing: one of the level's ten stores (the loop is rotated: stores on top)
    i = nl_asm.rindex("global_store_dwordx2", k0, w)
    j = nl_asm.rindex("\n", 0, i) + 1
    with pytest.raises(AssertionError):
        isa.check_nl_ring(nl_asm[:j] + "\ts_nop 0 ; " + nl_asm[j:].lstrip())           # one store of a level gone
    with pytest.raises(AssertionError):
        isa.check_nl_ring(nl_asm[:j] + "\ts_waitcnt vmcnt(5)\n" + nl_asm[j:])          # a wait hipcc added on its own
