"""The plain-C + OpenMP restatement (oracle/cloudsc2_nl_omp.c, the multi-threaded CPU baseline of bench.py - a plain port, not a tuned CPU code) against the
NumPy oracle, which is itself bit-identical to the executed reference source (tests/test_reference_exec.py).
Both evaluate the same IEEE-double statements (the C build uses -ffp-contract=off); what is left are libm-vs-NumPy
exp/tanh/pow differences (<= 1 ulp each) amplified through cancellations - hence 1e-12 of each field's scale."""
import numpy as np
import pytest
from helpers import NL_OUT, externals, nl_case, run_oracle_nl

from oracle import cloudsc2_c, cloudsc2_numpy


def _close(name, got, want):
    scale = max(np.abs(want).max(), np.finfo(np.float64).tiny)
    err = np.abs(got - want).max() / scale
    assert err <= 1e-12, f"{name}: {err:.2e} of field scale"


@pytest.mark.parametrize("sw", [dict(), dict(LEVAPLS2=True), dict(LDRAIN1D=True), dict(LPHYLIN=False),
                                dict(LPHYLIN=False, LEVAPLS2=True)])
@pytest.mark.parametrize("nx", [37, 512])
def test_c_nl_matches_numpy_oracle(sw, nx):
    ext = externals(**sw)
    fields, eta, dt = nl_case(nx, 137, np.float64, ext=ext)
    want = run_oracle_nl(fields, eta, dt, ext)
    F = dict(fields)
    for n in NL_OUT:
        F["out_" + n] = np.full_like(fields["in_ap"], np.nan)
    cloudsc2_c.cloudsc2_nl(F, eta, dt, ext, nthreads=3)
    for n in NL_OUT:
        got = F["out_" + n]
        if n in ("fplsl", "fplsn"):          # level 0 is not written (Q2), exactly like the reference
            assert np.isnan(got[0]).all()
            got, w = got[1:], want[n][1:]
        elif n in ("fhpsl", "fhpsn"):
            w = want[n]
        else:                                # full-level fields: padding level nz untouched
            assert np.isnan(got[-1]).all()
            got, w = got[:-1], want[n][:-1]
        _close(n, got, w)


@pytest.mark.parametrize("sw", [dict(), dict(LPHYLIN=False, KFLAG=1), dict(LPHYLIN=False, KFLAG=0)])
def test_c_saturation_matches_numpy_oracle(sw):
    ext = externals(**sw)
    fields, _, _ = nl_case(300, 137, np.float64)
    want = np.zeros_like(fields["in_t"])
    cloudsc2_numpy.saturation(fields["in_ap"], fields["in_t"], want, ext)
    got = np.zeros_like(want)
    cloudsc2_c.saturation(fields["in_ap"], fields["in_t"], got, ext, nthreads=2)
    _close("qsat", got, want)


def test_thread_count_does_not_change_results():
    ext = externals()
    fields, eta, dt = nl_case(200, 137, np.float64)
    outs = []
    for nt in (1, 4):
        F = dict(fields)
        for n in NL_OUT:
            F["out_" + n] = np.zeros_like(fields["in_ap"])
        cloudsc2_c.cloudsc2_nl(F, eta, dt, ext, nthreads=nt)
        outs.append(F)
    for n in NL_OUT:
        assert np.array_equal(outs[0]["out_" + n], outs[1]["out_" + n])
