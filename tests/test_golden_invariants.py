"""What the reference's golden output files pin without `data/input.h5`.

The golden files hold OUTPUTS of the Fortran CLOUDSC2-NL dwarf for 100 columns x 137 levels
(/root/reference/data/reference_{double,single}.h5, converted by tests/golden/make_reference_npz.py).
Their inputs are a missing blob, so they cannot be reproduced; they still pin the output layout, the
flux/enthalpy-flux relation of nonlinear/_stencils/cloudsc2.py:396-399 (hence RLSTT, RLVTT), the
behaviour of `out_covptot` under the driver flags, and value ranges.  The same invariants are then
required of the oracle's output on synthetic columns.
"""
import os

import numpy as np
import pytest

from helpers import externals, nl_case, run_oracle_nl

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["double", "single"])
def golden(request):
    return request.param, np.load(os.path.join(HERE, "golden", f"reference_{request.param}.npz"))


def test_layout(golden):
    prec, g = golden
    assert int(g["KLEV"][0]) == 137 and int(g["KLON"][0]) == 100
    dt = np.float64 if prec == "double" else np.float32
    for n in ("PCLC", "PCOVPTOT", "TENDENCY_LOC_Q", "TENDENCY_LOC_T"):
        assert g[n].shape == (137, 100) and g[n].dtype == dt
    for n in ("PFHPSL", "PFHPSN", "PFPLSL", "PFPLSN"):
        assert g[n].shape == (138, 100) and g[n].dtype == dt  # half levels: (K-1/2, IJ)
    assert g["TENDENCY_LOC_CLD"].shape == (5, 137, 100)


def test_enthalpy_flux_relation_pins_latent_heats(golden):
    prec, g = golden
    e = externals()
    if prec == "double":
        # cloudsc2.py:399  out_fhpsn = -out_fplsn * RLSTT  -- bit-exact in the fp64 file
        assert np.array_equal(g["PFHPSN"], -g["PFPLSN"] * e["RLSTT"])
        assert np.array_equal(g["PFHPSL"], -g["PFPLSL"] * e["RLVTT"])
    else:
        np.testing.assert_allclose(g["PFHPSN"], -g["PFPLSN"] * np.float32(e["RLSTT"]), rtol=2e-7, atol=0)


def test_flux_and_cover_invariants(golden):
    prec, g = golden
    # the fp32 file carries round-off dust of order -1e-23 in the fluxes
    floor = 0.0 if prec == "double" else -1e-19
    assert np.all(g["PCOVPTOT"] == 0.0)          # evaporation block is dead under the driver flags
    assert np.all(g["PFPLSL"][0] == 0) and np.all(g["PFPLSN"][0] == 0)   # nothing enters at the top
    assert np.all(g["PFPLSN"] >= floor) and np.all(g["PFPLSL"] >= floor)
    assert np.all(np.diff(g["PFPLSN"], axis=0) >= floor)  # cold sample: snow never melts or evaporates
    assert np.all((g["PCLC"] >= 0) & (g["PCLC"] <= 1))
    assert np.all(g["TENDENCY_LOC_CLD"][2:] == 0)      # only ql (0) and qi (1) carry tendencies


def test_latent_heat_ratio_pins_rcpd():
    g = np.load(os.path.join(HERE, "golden", "reference_double.npz"))
    e = externals()
    tq, tt = g["TENDENCY_LOC_Q"], g["TENDENCY_LOC_T"]
    m = np.abs(tq) > 1e-12
    ratio = -tt[m] / tq[m]
    # pure sublimation/deposition points give exactly RLSTT/RCPD (RVTMP2 = 0)
    want = e["RLSTT"] / e["RCPD"]
    # ... and that value (2821.215) is the mode of the distribution: ~3 % of the points hit it to 1e-6
    frac = np.mean(np.abs(ratio / want - 1.0) < 1e-6)
    assert frac > 0.02, frac
    vals, counts = np.unique(np.round(ratio, 3), return_counts=True)
    assert abs(vals[np.argmax(counts)] - want) < 2e-3


def test_oracle_obeys_the_same_invariants():
    e = externals()
    fields, eta, dt = nl_case(96)
    out = run_oracle_nl(fields, eta, dt, e)
    assert np.all(out["covptot"] == 0.0)
    assert np.array_equal(out["fhpsn"][1:], -out["fplsn"][1:] * e["RLSTT"])
    assert np.array_equal(out["fhpsl"][1:], -out["fplsl"][1:] * e["RLVTT"])
    assert np.all(out["fhpsl"][0] == 0) and np.all(out["fhpsn"][0] == 0)
    assert np.all((out["clc"] >= 0) & (out["clc"] <= 1))
    assert np.all(out["fplsn"] >= 0) and np.all(out["fplsl"] >= 0)
