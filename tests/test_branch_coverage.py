"""Branch-coverage evidence for the synthetic fixtures ("branch-coverage counters").

The parity tests are only as strong as the regimes their inputs reach.  The oracle's `cloudsc2_nl` can count, per
branch of the scheme, the grid points that took it (`branch_counts`, no effect on results); this test fails if any
branch of nonlinear/_stencils/cloudsc2.py:93-399 is not exercised by (a) the 40 columns of the executed-reference
fixtures (tests/golden/reference_exec*.npz) and (b) `nl_case`, the generator behind every HIP-vs-oracle test."""
import os

import numpy as np
import pytest

from helpers import NL_IN, NL_OUT, externals, nl_case
from oracle import cloudsc2_numpy as oracle

HERE = os.path.dirname(os.path.abspath(__file__))

ALWAYS = ("clear", "partial", "overcast",                       # Le Treut & Li regimes (:196-207)
          "cold_fwat", "ice_supersaturation", "esdp_clipped",   # t < RTT (:141-155), t < RTICE (:189-193), ZQMAX clip
          "detrainment", "detrainment_without_updraught_condensate",      # :210-215 both outcomes of the lu test
          "subsidence_evaporates_all_condensate", "subsidence_evaporates_part",   # :224 min() both ways
          "snow_enters_level", "melting", "melting_all_snow", "melting_part_of_snow",   # :238-246
          "autoconversion", "new_precip_as_snow", "new_precip_as_rain", "rain_refreezes",   # :249-285
          "adjustment_condenses", "adjustment_evaporates", "adjustment_warm_branch",   # :347-364
          "adjustment_crosses_RTT", "adjustment_precip_as_snow", "adjustment_precip_as_rain")
EVAP_ONLY = ("evaporation", "evaporation_of_all_precip")        # :288-321 (LEVAPLS2 or LDRAIN1D)


def _counts(fields, eta, dt, ext):
    F = dict(fields)
    for n in NL_OUT:
        F["out_" + n] = np.zeros_like(fields["in_ap"])
    bc = {}
    oracle.cloudsc2_nl(F, eta, dt, ext, branch_counts=bc)
    return bc


def _check(bc, names, what):
    missing = [n for n in names if bc.get(n, 0) == 0]
    assert not missing, f"{what}: no grid point reaches {missing}; counts: {bc}"


@pytest.mark.parametrize("evap", [False, True])
def test_golden_fixture_columns_reach_every_nl_branch(evap):
    g = np.load(os.path.join(HERE, "golden", "reference_exec.npz"))
    fields = {"in_" + n: g["in_" + n] for n in NL_IN}
    bc = _counts(fields, g["eta"], float(g["dt"]), externals(LEVAPLS2=evap))
    names = ALWAYS + (EVAP_ONLY if evap else ())
    if evap:
        # with evaporation thinning the snow flux, none of the 40 columns melts only PART of its snow; that outcome of
        # min(sfl, cons * max(t - meltp2, 0)) (:239) is reached by the same columns without evaporation (this test,
        # evap=False) and, with evaporation, by nl_case(256) below
        names = tuple(n for n in names if n != "melting_part_of_snow")
    _check(bc, names, "reference_exec.npz (40 columns)")
    total = fields["in_ap"].shape[1] * 137
    assert bc["clear"] + bc["partial"] + bc["overcast"] == total


@pytest.mark.parametrize("evap", [False, True])
@pytest.mark.parametrize("nx", [256])
def test_nl_case_reaches_every_nl_branch(nx, evap):
    fields, eta, dt = nl_case(nx)
    bc = _counts(fields, eta, dt, externals(LEVAPLS2=evap))
    _check(bc, ALWAYS + (EVAP_ONLY if evap else ()), f"nl_case({nx})")


def test_counting_does_not_change_results():
    fields, eta, dt = nl_case(64)
    outs = []
    for bc in (None, {}):
        F = dict(fields)
        for n in NL_OUT:
            F["out_" + n] = np.zeros_like(fields["in_ap"])
        oracle.cloudsc2_nl(F, eta, dt, externals(), branch_counts=bc)
        outs.append({n: F["out_" + n] for n in NL_OUT})
    for n in NL_OUT:
        assert np.array_equal(outs[0][n], outs[1][n]), n
