"""The REFERENCE's own component classes drive the product's stencil objects into the C ABI.

`tests/run_reference_on_recording_hip.py` runs the three UNMODIFIED reference drivers with `--backend hip` in the GPU-less
build container; libcloudsc2_hip.so's entry points are replaced by a recording fake (nothing is computed) and the storages
of backend "hip" are host tensors, for this test only.  Checked here, against include/cloudsc2_hip.h read as text:

  * `Cloudsc2NL/TL/AD.array_call`, `Saturation`, `StateIncrement`, `PerturbedState` of the reference
    (nonlinear/microphysics.py:123-172, tangent_linear/microphysics.py:162-242, adjoint/microphysics.py:159-238,
    common/saturation.py:67-76, common/increment.py:93-132,219-261) each reach exactly ONE `cloudsc2_*_f64` call;
  * the pointer arrays are in the header's NL_IN_* / NL_OUT_* / INC_* enum order and hold the storages the reference
    passed under the matching gtscript keyword; `nx`, `nz`, `lev_stride`, `dt` / `f` and the externals arrive;
  * the gtscript scratch arguments (`tmp_*`, incl. `tmp_klevel`) are accepted and never reach the ABI;
  * `dt=` arrives as the dtype scalar the reference builds (`gt4py_config.dtypes.float(...)`).

Skipped on the GPU box (no reference checkout there); the same stencil objects run for real in the `-m gpu` tests."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "drivers")),
                                reason="reference checkout not present (GPU box)")

NX, NZ = 64, 137


def _enum(prefix):
    """field names of an `enum { PREFIX_A, PREFIX_B, ..., PREFIX_NUM... }` of the C header, in order, lower case"""
    text = open(os.path.join(ROOT, "include", "cloudsc2_hip.h")).read()
    for body in re.findall(r"enum\s*\{(.*?)\}", text, flags=re.S):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = [x.strip() for x in body.split(",") if x.strip()]
        if names and names[0].startswith(prefix) and all(n.startswith(prefix) or "NUM" in n for n in names):
            return [n[len(prefix):].lower() for n in names if "NUM" not in n]
    raise AssertionError(f"no enum {prefix}* in include/cloudsc2_hip.h")


NL_IN, NL_OUT, INC = _enum("NL_IN_"), _enum("NL_OUT_"), _enum("INC_")


@pytest.fixture(scope="module")
def records(tmp_path_factory):
    out = {}
    d = tmp_path_factory.mktemp("recording_hip")
    for key, driver in (("nl", "run_nonlinear.py"), ("tl", "run_taylor_test.py"), ("ad", "run_symmetry_test.py")):
        rec = d / f"{key}.json"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_reference_on_recording_hip.py"), str(rec), driver,
                            "--backend", "hip", "--num-cols", str(NX), "--num-runs", "1"],
                           capture_output=True, text=True, timeout=900, cwd=ROOT,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        out[key] = json.loads(rec.read_text())
        out[key]["stdout"] = p.stdout
    return out


def _calls(rec, stencil):
    got = []
    for c in rec["stencil_calls"]:
        if c["stencil"] == stencil:
            assert c["abi_last"] - c["abi_first"] == 1, c          # one stencil call = one C-ABI call
            got.append((c, rec["abi_calls"][c["abi_first"]]))
    assert got, f"the reference never called {stencil}"
    return got


def _ptrs(call, names):
    return [call["fields"][n]["ptr"] for n in names]


def _common(call, abi, entry, nlev_offset):
    """entry point, (params, nx, nz, lev_stride), origin / domain / validate_args as the reference passes them"""
    assert abi["entry"] == entry
    a = abi["args"]
    assert a[1] == NX and a[2] == NZ
    first = next(iter(call["fields"].values()))
    assert a[3] == first["strides"][2] >= NX                 # the level stride of the storages, in elements
    assert first["shape"] == [NX, 1, NZ + 1] and first["strides"][0] == 1 and first["dtype"] == "torch.float64"
    assert call["origin"] == [0, 0, 0] and call["domain"] == [NX, 1, NZ + nlev_offset]
    assert call["validate_args"] is False and a[-1] == 0     # the drivers' default (--disable-checks); the fake stream
    return a


def test_saturation_component_reaches_the_abi(records):
    for call, abi in _calls(records["nl"], "saturation"):
        a = _common(call, abi, "cloudsc2_saturation_f64", 0)
        assert a[4:7] == _ptrs(call, ["in_ap", "in_t", "out_qsat"])
        assert a[0]["params"]["LPHYLIN"] == 1 and a[0]["params"]["KFLAG"] == 1          # run_nonlinear.py:84-85


def test_nonlinear_component_reaches_the_abi(records):
    pairs = _calls(records["nl"], "cloudsc2_nl")
    assert len(pairs) == 2                                    # the warm-up call and the one timed run
    for call, abi in pairs:
        a = _common(call, abi, "cloudsc2_nl_f64", 1)
        assert a[4] == _ptrs(call, ["in_" + n for n in NL_IN])
        assert a[5] == call["fields"]["in_eta"]["ptr"] and call["fields"]["in_eta"]["shape"] == [NZ + 1]
        assert a[6] == _ptrs(call, ["out_" + n for n in NL_OUT])
        assert a[7] == 3600.0 and call["scalars"]["dt"]["type"] == "float64"            # a dtype scalar, not a Python float
        # the five 2-D scratch arguments of managed_temporary_storage (placeholders in this build: the kernels carry that
        # state in registers, framework/fields.py) are passed by the reference, accepted, and never reach the ABI
        tmp = [k for k in call["kwargs"] if k.startswith("tmp_")]
        assert sorted(tmp) == ["tmp_aph_s", "tmp_covptot", "tmp_rfl", "tmp_sfl", "tmp_trpaus"]
        assert not any(t in call["fields"] for t in tmp)
        assert len(set(a[4] + a[6])) == 26                    # 26 distinct storages
        p = a[0]["params"]
        assert p["LPHYLIN"] == 1 and p["LDRAIN1D"] == 0 and p["ZQMAX"] == 0.5 and p["ZSCAL"] == 0.9


def test_increment_components_reach_the_abi(records):
    for call, abi in _calls(records["tl"], "state_increment"):
        a = _common(call, abi, "cloudsc2_state_increment_f64", 1)
        assert a[4] == _ptrs(call, ["in_" + n for n in INC])
        assert a[5] == _ptrs(call, ["out_" + n + "_i" for n in INC])
        assert a[6] == 0.01 and a[0]["params"]["IGNORE_SUPSAT"] == 0      # factor1; the Taylor harness keeps the default (False)
    for call, abi in _calls(records["ad"], "state_increment"):
        assert abi["args"][0]["params"]["IGNORE_SUPSAT"] == 1             # adjoint/validation.py:119: ignore_supsat=True
    fs = []
    for call, abi in _calls(records["tl"], "perturbed_state"):
        a = _common(call, abi, "cloudsc2_perturbed_state_f64", 1)
        assert a[4] == _ptrs(call, ["in_" + n for n in INC])
        assert a[5] == _ptrs(call, ["in_" + n + "_i" for n in INC])
        assert a[6] == _ptrs(call, ["out_" + n for n in INC])
        fs.append(a[7])
    assert fs[:10] == pytest.approx([10.0 ** -(i + 1) for i in range(10)], rel=1e-15)   # run_taylor_test.py:76


def test_tangent_linear_component_reaches_the_abi(records):
    for call, abi in _calls(records["tl"], "cloudsc2_tl"):
        a = _common(call, abi, "cloudsc2_tl_f64", 1)
        assert a[4] == _ptrs(call, ["in_" + n for n in NL_IN])
        assert a[5] == _ptrs(call, ["in_" + n + "_i" for n in NL_IN])
        assert a[6] == call["fields"]["in_eta"]["ptr"]
        assert a[7] == _ptrs(call, ["out_" + n for n in NL_OUT])
        assert a[8] == _ptrs(call, ["out_" + n + "_i" for n in NL_OUT])
        assert a[9] == 3600.0 and call["scalars"]["dt"]["type"] == "float64"
        tmp = [k for k in call["kwargs"] if k.startswith("tmp_")]
        assert "tmp_klevel" in tmp and len(tmp) >= 10         # 2-D scratch pairs + the K-index vector
        # tmp_klevel is a REAL storage the reference fills (tangent_linear/microphysics.py:67-71): accepted, never passed on
        flat = [x for arg in a for x in (arg if isinstance(arg, list) else [arg])]
        assert call["fields"]["tmp_klevel"]["ptr"] not in flat
        assert a[0]["params"]["NLEV"] == NZ                   # external NLEV = the grid's nz (tangent_linear/microphysics.py:85)
    # the Taylor harness ran to its verdict on the (all-zero) fake outputs: 1 + 10 NL runs per TaylorTest.run
    assert sum(c["stencil"] == "cloudsc2_nl" for c in records["tl"]["stencil_calls"]) % 11 == 0
    assert "<<< Taylor test: End" in records["tl"]["stdout"]


def test_adjoint_component_reaches_the_abi(records):
    tl_out_i = None
    for call, abi in _calls(records["ad"], "cloudsc2_tl"):
        tl_out_i = abi["args"][8]
    for call, abi in _calls(records["ad"], "cloudsc2_ad"):
        a = _common(call, abi, "cloudsc2_ad_f64", 1)
        assert a[4] == _ptrs(call, ["in_" + n for n in NL_IN])
        assert a[5] == _ptrs(call, ["in_" + n + "_i" for n in NL_OUT])       # the adjoint forcing, NL_OUT_* order
        assert a[6] == call["fields"]["in_eta"]["ptr"]
        assert a[7] == _ptrs(call, ["out_" + n for n in NL_OUT])
        assert a[8] == _ptrs(call, ["out_" + n + "_i" for n in NL_IN])       # adjoint of the 16 inputs, NL_IN_* order
        assert a[9] == 3600.0
        assert "tmp_klevel" in call["kwargs"]
        assert a[0]["params"]["NLEV"] == NZ
        # adjoint/validation.py:149-150: the TL perturbation outputs are rebound as the AD forcing - same storages
        assert a[5] == tl_out_i
    assert "symmetry test" in records["ad"]["stdout"]
