"""Self-tests of tests/golden/gtscript_exec.py - the executor that pins the oracle to the reference's stencil source.

GT4Py is not installed, so that executor is the build's own reading of gtscript semantics ("parity
unpinned").  These tests state each rule it implements on stencils of the BUILD's own (written in the reference's
dialect, parsed from text, never imported) with results worked out by hand, so the reading is explicit and a change to the
executor that alters a rule fails here rather than silently shifting every golden vector.  Rules = GT4Py cartesian
semantics as documented for gtscript: FORWARD / BACKWARD are sequential in k with every statement applied to the whole
horizontal plane, intervals of one computation are visited in k order, PARALLEL blocks here are pointwise, a field `if` masks
the assignments of its body, conditions on externals are resolved statically, temporaries are fields that persist between
computations, `@gtscript.function`s are inlined."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from gtscript_exec import Executor, load_definitions  # noqa: E402

SRC = '''
from gt4py.cartesian import gtscript
from ifs_physics_common.stencil import stencil_collection, function_collection


@function_collection("f_pair")
@gtscript.function
def f_pair(a, b):
    from __externals__ import SCALE

    s = a + b
    if a > b:
        d = SCALE * (a - b)
    else:
        d = 0.0
    return s, d


@stencil_collection("cumsum_down")
def cumsum_down(in_a: gtscript.Field["float"], out_s: gtscript.Field["float"]):
    with computation(FORWARD):
        with interval(0, 1):
            out_s[0, 0, 0] = in_a[0, 0, 0]
        with interval(1, None):
            out_s[0, 0, 0] = out_s[0, 0, -1] + in_a[0, 0, 0]


@stencil_collection("cumsum_up")
def cumsum_up(in_a: gtscript.Field["float"], out_s: gtscript.Field["float"]):
    with computation(BACKWARD):
        with interval(-1, None):
            out_s[0, 0, 0] = in_a[0, 0, 0]
        with interval(0, -1):
            out_s[0, 0, 0] = out_s[0, 0, 1] + in_a[0, 0, 0]


@stencil_collection("masks")
def masks(in_a: gtscript.Field["float"], out_x: gtscript.Field["float"], out_y: gtscript.Field["float"]):
    from __externals__ import FLAG, THRESH

    with computation(PARALLEL), interval(...):
        out_x[0, 0, 0] = -1.0
        if in_a > THRESH:
            out_x[0, 0, 0] = in_a[0, 0, 0] * 2.0
            if in_a > 2.0 * THRESH:
                out_y[0, 0, 0] = 7.0
        else:
            out_y[0, 0, 0] = 3.0
        if FLAG:
            out_x[0, 0, 0] = out_x[0, 0, 0] + 100.0


@stencil_collection("temps_and_carries")
def temps_and_carries(
    in_a: gtscript.Field["float"],
    in_eta: gtscript.Field[gtscript.K, "float"],
    out_b: gtscript.Field["float"],
    out_c: gtscript.Field["float"],
    tmp_carry: gtscript.Field[gtscript.IJ, "float"],
    *,
    dt: "float",
):
    with computation(FORWARD), interval(0, 1):
        tmp_carry[0, 0] = 0.0
    with computation(FORWARD), interval(0, -1):
        tmp = in_a[0, 0, 0] * dt + in_eta[0]
        tmp_carry[0, 0] = tmp_carry[0, 0] + tmp
        out_b[0, 0, 0] = tmp_carry[0, 0]
    with computation(BACKWARD), interval(0, -1):
        out_c[0, 0, 0] = tmp[0, 0, 0] + tmp[0, 0, 1] + never_assigned[0, 0, 0]


@stencil_collection("uses_function")
def uses_function(in_a: gtscript.Field["float"], in_b: gtscript.Field["float"], out_s: gtscript.Field["float"],
                  out_d: gtscript.Field["float"]):
    with computation(PARALLEL), interval(...):
        s, d = f_pair(in_a, in_b)
        out_s[0, 0, 0] = s
        out_d[0, 0, 0] = d ** 2.0 + max(in_a[0, 0, 0], 1.0) - min(in_b[0, 0, 0], 0.0)
'''


@pytest.fixture(scope="module")
def defs(tmp_path_factory):
    p = tmp_path_factory.mktemp("gts") / "mini.py"
    p.write_text(SRC)
    return load_definitions([str(p)])


def _field(nz, nx, seed):
    return np.random.default_rng(seed).uniform(-2.0, 2.0, size=(nz + 1, nx))


def test_forward_and_backward_are_sequential_in_k_and_intervals_follow_k_order(defs):
    nz, nx = 6, 5
    a = _field(nz, nx, 1)
    s = np.zeros_like(a)
    Executor(defs, {}).run("cumsum_down", {"in_a": a, "out_s": s}, {}, nz)
    assert np.allclose(s, np.cumsum(a, axis=0), rtol=0, atol=1e-15)
    s2 = np.zeros_like(a)
    Executor(defs, {}).run("cumsum_up", {"in_a": a, "out_s": s2}, {}, nz)            # interval(-1, None) = last level first
    assert np.allclose(s2, np.cumsum(a[::-1], axis=0)[::-1], rtol=0, atol=1e-15)


def test_domain_levels_restrict_the_vertical_domain(defs):
    nz, nx = 6, 3
    a = _field(nz, nx, 2)
    s = np.full_like(a, 9.0)
    Executor(defs, {}).run("cumsum_down", {"in_a": a, "out_s": s}, {}, nz, domain_levels=nz)   # as `saturation`: nz levels
    assert np.allclose(s[:nz], np.cumsum(a[:nz], axis=0)) and (s[nz] == 9.0).all()             # level nz untouched


@pytest.mark.parametrize("flag", [False, True])
def test_field_ifs_mask_assignments_and_external_ifs_are_static(defs, flag):
    nz, nx = 3, 8
    a = _field(nz, nx, 3)
    x, y = np.zeros_like(a), np.full_like(a, -5.0)
    Executor(defs, {"FLAG": flag, "THRESH": 0.5}).run("masks", {"in_a": a, "out_x": x, "out_y": y}, {}, nz)
    want_x = np.where(a > 0.5, 2.0 * a, -1.0) + (100.0 if flag else 0.0)
    want_y = np.where(a > 0.5, np.where(a > 1.0, 7.0, -5.0), 3.0)        # untouched points keep what the storage held
    assert np.array_equal(x, want_x) and np.array_equal(y, want_y)


def test_temporaries_persist_ij_fields_carry_and_unassigned_temporaries_read_zero(defs):
    nz, nx = 5, 4
    a = _field(nz, nx, 4)
    eta = np.linspace(0.1, 0.9, nz + 1)
    b, c = np.zeros_like(a), np.zeros_like(a)
    carry = np.full(nx, 123.0)
    Executor(defs, {}).run("temps_and_carries", {"in_a": a, "in_eta": eta, "out_b": b, "out_c": c, "tmp_carry": carry},
                           {"dt": 2.0}, nz)
    tmp = np.zeros_like(a)
    tmp[:nz] = a[:nz] * 2.0 + eta[:nz, None]                              # interval(0, -1): the last level is not visited
    assert np.allclose(b[:nz], np.cumsum(tmp[:nz], axis=0)) and (b[nz] == 0).all()
    assert np.allclose(carry, tmp[:nz].sum(axis=0))                       # the IJ field holds the last carried value
    want_c = np.zeros_like(a)
    want_c[:nz] = tmp[:nz] + tmp[1:nz + 1]                                # tmp[nz] was never assigned: reads 0
    assert np.allclose(c, want_c)


def test_functions_are_inlined_with_their_own_externals_and_masked_locals(defs):
    nz, nx = 2, 6
    a, b = _field(nz, nx, 5), _field(nz, nx, 6)
    s, d = np.zeros_like(a), np.zeros_like(a)
    Executor(defs, {"SCALE": 3.0}).run("uses_function", {"in_a": a, "in_b": b, "out_s": s, "out_d": d}, {}, nz)
    dd = np.where(a > b, 3.0 * (a - b), 0.0)
    assert np.allclose(s, a + b) and np.allclose(d, dd ** 2.0 + np.maximum(a, 1.0) - np.minimum(b, 0.0))
    with pytest.raises(KeyError, match="SCALE"):
        Executor(defs, {}).run("uses_function", {"in_a": a, "in_b": b, "out_s": s, "out_d": d}, {}, nz)


def test_float32_runs_keep_every_field_valued_quantity_in_float32(defs):
    nz, nx = 3, 5
    a = _field(nz, nx, 7).astype(np.float32)
    b = _field(nz, nx, 8).astype(np.float32)
    s, d = np.zeros_like(a), np.zeros_like(a)
    temps = Executor(defs, {"SCALE": 1.0 / 3.0}, dtype=np.float32).run(
        "uses_function", {"in_a": a, "in_b": b, "out_s": s, "out_d": d}, {}, nz)
    assert s.dtype == d.dtype == np.float32 and all(t.dtype == np.float32 for t in temps.values())
    # the external 1/3 is a weak scalar: the product is formed in float32, not in double and rounded afterwards
    dd = np.where(a > b, np.float32(1.0 / 3.0) * (a - b), np.float32(0.0)).astype(np.float32)
    want = (dd ** np.float32(2.0) + np.maximum(a, np.float32(1.0)) - np.minimum(b, np.float32(0.0))).astype(np.float32)
    assert np.array_equal(d, want)
