"""ORACLE (test infrastructure, NOT product code): NumPy restatement of the CLOUDSC2 stencils.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module; the product package never does (it fails loudly when the HIP library is missing).

What it restates (statement by statement, from the source text; the reference itself cannot be
imported here because `gt4py` and `ifs_physics_common` are not installed - SURVEY.md F2):

  saturation       /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/saturation.py:23-42
                   /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/fcttre.py:22-57
  cloudsc2_nl      /root/reference/src/cloudsc2_gt4py/physics/nonlinear/_stencils/cloudsc2.py:93-399
  f_cuadjtqs_nl    /root/reference/src/cloudsc2_gt4py/physics/nonlinear/_stencils/cuadjtqs.py:22-68
  cloudsc2_tl      /root/reference/src/cloudsc2_gt4py/physics/tangent_linear/_stencils/cloudsc2.py:124-774
  f_cuadjtqs_tl    /root/reference/src/cloudsc2_gt4py/physics/tangent_linear/_stencils/cuadjtqs.py:22-84
  cloudsc2_ad      /root/reference/src/cloudsc2_gt4py/physics/adjoint/_stencils/cloudsc2.py:124-996
  f_cuadjtqs_ad    /root/reference/src/cloudsc2_gt4py/physics/adjoint/_stencils/cuadjtqs.py:22-158
  state_increment  /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/state_increment.py:61-80
  perturbed_state  /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/perturbed_state.py:75-91
  eta levels       /root/reference/src/cloudsc2_gt4py/physics/common/diagnostics.py:42-45

Execution shape = GT4Py's numpy backend: the vertical loop is explicit, every gtscript statement
is one whole-array NumPy statement over the columns, field `if`s are `np.where` masks, temporaries
of the AD stencil are full (nz+1, nx) arrays that persist between the forward and the backward
computation (and read as 0 where never assigned).

Array layout: ``a[k, col]`` with nz+1 levels (`interval(0, -1)` = k in [0, nz)).

PARITY PINNING.  `data/input.h5` is absent, so the reference's golden HDF5 files pin only layout and
invariants (tests/test_golden_invariants.py): parity against data/reference_{double,single}.h5 is
UNPINNED.  The formulas are pinned instead by the reference's own source: tests/golden/
reference_exec.npz holds the outputs of the unmodified reference stencils executed by the build's
gtscript executor (tests/golden/gtscript_exec.py, make_reference_exec.py) on seeded columns, and this
module reproduces them BIT FOR BIT (tests/test_reference_exec.py; one AD field to 1 ulp) - plus the
TL Taylor test against NL and the AD dot-product test against TL (tests/test_oracle_consistency.py).
"""
from __future__ import annotations

from typing import Any, Dict, Mapping

import numpy as np

_ERR = dict(divide="ignore", invalid="ignore", over="ignore", under="ignore")


def _ext(externals: Mapping[str, Any], dtype) -> Dict[str, Any]:
    """Externals as Python scalars (GT4Py bakes them in as Python doubles / bools)."""
    out = {}
    for k, v in externals.items():
        out[k] = v
    return out


# --------------------------------------------------------------------------------------
# small operators
# --------------------------------------------------------------------------------------
def eta_levels(ap: np.ndarray, aph: np.ndarray) -> np.ndarray:
    """common/diagnostics.py:42-45: eta[k] = ap[col 0, k] / aph[col 0, nz], k < nz."""
    nz = ap.shape[0] - 1
    eta = np.zeros(nz + 1, dtype=ap.dtype)
    for k in range(nz):
        eta[k] = ap[k, 0] / aph[nz, 0]
    return eta


def f_foealfa(t, e):
    """fcttre.py:22-27"""
    return np.minimum(1.0, ((np.maximum(e["RTICE"], np.minimum(e["RTWAT"], t)) - e["RTICE"])
                            * e["RTWAT_RTICE_R"]) ** 2.0)


def f_foealfcu(t, e):
    """fcttre.py:30-35"""
    return np.minimum(1.0, ((np.maximum(e["RTICECU"], np.minimum(e["RTWAT"], t)) - e["RTICECU"])
                            * e["RTWAT_RTICECU_R"]) ** 2.0)


def f_foeewm(t, e):
    """fcttre.py:38-46"""
    return e["R2ES"] * (
        f_foealfa(t, e) * np.exp(e["R3LES"] * (t - e["RTT"]) / (t - e["R4LES"]))
        + (1.0 - f_foealfa(t, e)) * (np.exp(e["R3IES"] * (t - e["RTT"]) / (t - e["R4IES"])))
    )


def f_foeewmcu(t, e):
    """fcttre.py:49-57"""
    return e["R2ES"] * (
        f_foealfcu(t, e) * np.exp(e["R3LES"] * (t - e["RTT"]) / (t - e["R4LES"]))
        + (1.0 - f_foealfcu(t, e)) * (np.exp(e["R3IES"] * (t - e["RTT"]) / (t - e["R4IES"])))
    )


def saturation(in_ap, in_t, out_qsat, externals) -> None:
    """saturation.py:23-42 on domain (nx, 1, nz): levels 0 .. nz-1 of the (nz+1)-level arrays."""
    e = externals
    nz = in_ap.shape[0] - 1
    _WORK["dtype"] = in_ap.dtype
    with np.errstate(**_ERR):
        t = in_t[:nz]
        ap = in_ap[:nz]
        if e["LPHYLIN"]:
            alfa = f_foealfa(t, e)
            foeewl = e["R2ES"] * np.exp(e["R3LES"] * (t - e["RTT"]) / (t - e["R4LES"]))
            foeewi = e["R2ES"] * np.exp(e["R3IES"] * (t - e["RTT"]) / (t - e["R4IES"]))
            foeew = alfa * foeewl + (1.0 - alfa) * foeewi
            qs = np.minimum(foeew / ap, e["QMAX"])
        else:
            ew = f_foeewmcu(t, e) if e["KFLAG"] == 1 else f_foeewm(t, e)
            qs = np.minimum(ew / ap, e["QMAX"])
        out_qsat[:nz] = qs / (1.0 - e["RETV"] * qs)


_INCR = ("aph", "ap", "q", "qsat", "t", "ql", "qi", "lude", "lu", "mfu", "mfd",
         "tnd_cml_t", "tnd_cml_q", "tnd_cml_ql", "tnd_cml_qi", "supsat")


def state_increment(state: Mapping[str, np.ndarray], out: Dict[str, np.ndarray], f, ignore_supsat) -> None:
    """state_increment.py:61-80, all nz+1 levels; keys are the gtscript names without in_/out_."""
    for n in _INCR:
        if n == "supsat" and ignore_supsat:
            out[n + "_i"][...] = 0.0
        else:
            out[n + "_i"][...] = f * state[n]


def perturbed_state(state: Mapping[str, np.ndarray], out: Dict[str, np.ndarray], f) -> None:
    """perturbed_state.py:75-91"""
    for n in _INCR:
        out[n][...] = state[n] + f * state[n + "_i"]


# --------------------------------------------------------------------------------------
# saturation adjustment
# --------------------------------------------------------------------------------------
def _cuadjtqs_nl_0(ap, t, q, z3es, z4es, z5alcp, zaldcp, e):
    """nonlinear/_stencils/cuadjtqs.py:24-37"""
    foeew = e["R2ES"] * np.exp(z3es * (t - e["RTT"]) / (t - z4es))
    qsat = np.minimum(foeew / ap, e["ZQMAX"])
    cor = 1.0 / (1.0 - e["RETV"] * qsat)
    qsat = qsat * cor
    z2s = z5alcp / (t - z4es) ** 2.0
    cond = (q - qsat) / (1.0 + qsat * cor * z2s)
    t = t + zaldcp * cond
    q = q - cond
    return t, q


def f_cuadjtqs_nl(ap, t, q, e):
    """nonlinear/_stencils/cuadjtqs.py:40-68 (ICALL == 0 only, as the reference)."""
    assert e["ICALL"] == 0
    warm = t > e["RTT"]
    z3es = _where(warm, e["R3LES"], e["R3IES"])
    z4es = _where(warm, e["R4LES"], e["R4IES"])
    z5alcp = _where(warm, e["R5ALVCP"], e["R5ALSCP"])
    zaldcp = _where(warm, e["RALVDCP"], e["RALSDCP"])
    t, q = _cuadjtqs_nl_0(ap, t, q, z3es, z4es, z5alcp, zaldcp, e)
    t, q = _cuadjtqs_nl_0(ap, t, q, z3es, z4es, z5alcp, zaldcp, e)
    return t, q


# --------------------------------------------------------------------------------------
# shared pieces of the three sweeps
# --------------------------------------------------------------------------------------
def _trpaus(eta, t, nz, dtype):
    """cloudsc2.py:107-111 - eta of the LAST level k in [0, nz-2] with 0.1<eta<0.4 and t[k]>t[k+1]."""
    nx = t.shape[1]
    trpaus = np.full(nx, 0.1, dtype=dtype)
    for k in range(nz - 1):
        if eta[k] > 0.1 and eta[k] < 0.4:
            trpaus = _where(t[k] > t[k + 1], eta[k], trpaus)
    return trpaus


def _crh2(eta_k, trpaus):
    """cloudsc2.py:166-186 - critical relative humidity (depends on level and trpaus only)."""
    rh1 = 1.0
    rh2 = (0.35 + 0.14 * ((trpaus - 0.25) / 0.15) ** 2.0
           + 0.04 * np.minimum(trpaus - 0.25, 0.0) / 0.15)
    rh3 = 1.0
    deta2 = 0.3
    bound1 = trpaus + deta2
    deta1 = 0.09 + 0.16 * (0.4 - trpaus) / 0.3
    bound2 = 1.0 - deta1
    crh2 = _where(
        eta_k < trpaus,
        rh3,
        _where(
            eta_k < bound1,
            rh3 + (rh2 - rh3) * (eta_k - trpaus) / deta2,
            _where(eta_k < bound2, rh2, rh1 + (rh2 - rh1) * np.sqrt((1.0 - eta_k) / deta1)),
        ),
    )
    return crh2


NL_INPUTS = ("in_ap", "in_aph", "in_lu", "in_lude", "in_mfd", "in_mfu", "in_q", "in_qi", "in_ql",
             "in_qsat", "in_supsat", "in_t", "in_tnd_cml_q", "in_tnd_cml_qi", "in_tnd_cml_ql",
             "in_tnd_cml_t")
NL_OUTPUTS = ("out_clc", "out_covptot", "out_fhpsl", "out_fhpsn", "out_fplsl", "out_fplsn",
              "out_tnd_q", "out_tnd_qi", "out_tnd_ql", "out_tnd_t")


def _where(cond, a, b):
    """np.where in the working precision.  With two scalar branches (externals are Python floats) NumPy returns a
    float64 array, which would silently promote everything downstream of it in a float32 run; every field-valued
    quantity of the stencils is held in the field precision instead (fp32 semantics: tests/golden/gtscript_exec.py,
    make_reference_exec.py).  In a float64 run this is np.where."""
    r = np.where(cond, a, b)
    if r.dtype == np.float64 and _WORK["dtype"] == np.float32:
        return r.astype(np.float32)
    return r


_WORK = {"dtype": np.dtype(np.float64)}   # set by each stencil entry point from its fields


def _count(bc, name, mask) -> None:
    if bc is not None:
        bc[name] = bc.get(name, 0) + int(np.count_nonzero(mask))


def cloudsc2_nl(fields: Dict[str, np.ndarray], in_eta: np.ndarray, dt, externals, branch_counts=None) -> None:
    """nonlinear/_stencils/cloudsc2.py:93-399.  `fields` holds the 16 `in_*` and 10 `out_*`
    arrays (nz+1, nx); outputs are written in place.  `out_fplsl[0]`/`out_fplsn[0]` are NOT
    written, exactly as the reference.
    `branch_counts` (a dict, optional): receives the number of grid points that took each branch of the scheme -
    the coverage evidence of tests/test_branch_coverage.py; it does not change any result."""
    e = externals
    bc = branch_counts
    F = fields
    in_ap, in_aph, in_lu, in_lude = F["in_ap"], F["in_aph"], F["in_lu"], F["in_lude"]
    in_mfd, in_mfu, in_qsat = F["in_mfd"], F["in_mfu"], F["in_qsat"]
    dtype = in_ap.dtype
    _WORK["dtype"] = dtype
    nz = in_ap.shape[0] - 1
    nx = in_ap.shape[1]
    dt = dtype.type(dt)
    eta = in_eta
    LEV = e["LEVAPLS2"] or e["LDRAIN1D"]
    RG, RTT, RCPD = e["RG"], e["RTT"], e["RCPD"]
    ZEPS1, ZEPS2 = e["ZEPS1"], e["ZEPS2"]

    with np.errstate(**_ERR):
        # :93-100
        tmp_rfl = np.zeros(nx, dtype)
        tmp_sfl = np.zeros(nx, dtype)
        tmp_covptot = np.zeros(nx, dtype)
        tmp_aph_s = in_aph[nz].copy()
        # :102-104
        t3d = F["in_t"][:nz] + dt * F["in_tnd_cml_t"][:nz]
        # :107-111
        tmp_trpaus = _trpaus(eta, t3d, nz, dtype)

        fplsl = np.zeros((nz, nx), dtype)
        fplsn = np.zeros((nz, nx), dtype)

        for k in range(nz):
            t = t3d[k]
            ap = in_ap[k]
            qs_in = in_qsat[k]
            # :115-117
            q = F["in_q"][k] + dt * F["in_tnd_cml_q"][k] + F["in_supsat"][k]
            ql = F["in_ql"][k] + dt * F["in_tnd_cml_ql"][k]
            qi = F["in_qi"][k] + dt * F["in_tnd_cml_qi"][k]
            # :120-124
            ckcodtl = 2.0 * e["RKCONV"] * dt
            ckcodti = 5.0 * e["RKCONV"] * dt
            cons2 = 1.0 / (RG * dt)
            cons3 = e["RLVTT"] / RCPD
            meltp2 = RTT + 2.0
            # :127
            scalm = e["ZSCAL"] * max(eta[k] - 0.2, ZEPS1) ** 0.2
            # :130-134
            dp = in_aph[k + 1] - in_aph[k]
            zz = RCPD + RCPD * e["RVTMP2"] * q
            lfdcp = e["RLMLT"] / zz
            lsdcp = e["RLSTT"] / zz
            lvdcp = e["RLVTT"] / zz
            # :141-160
            if e["LPHYLIN"] or e["LDRAIN1D"]:
                cold = t < RTT
                fwat = _where(cold, 0.545 * (np.tanh(0.17 * (t - e["RLPTRC"])) + 1.0), 1.0)
                z3es = _where(cold, e["R3IES"], e["R3LES"])
                z4es = _where(cold, e["R4IES"], e["R4LES"])
                foeew = e["R2ES"] * np.exp(z3es * (t - RTT) / (t - z4es))
                esdp = np.minimum(foeew / ap, e["ZQMAX"])
            else:
                fwat = f_foealfa(t, e)
                foeew = f_foeewm(t, e)
                esdp = foeew / ap
            facw = e["R5LES"] / ((t - e["R4LES"]) ** 2.0)
            faci = e["R5IES"] / ((t - e["R4IES"]) ** 2.0)
            fac = fwat * facw + (1.0 - fwat) * faci
            dqsdtemp = fac * qs_in / (1.0 - e["RETV"] * esdp)
            corqs = 1.0 + cons3 * dqsdtemp
            # :163
            qlim = np.minimum(q, qs_in)
            # :166-186
            crh2 = _crh2(eta[k], tmp_trpaus)
            # :189-193
            qsat = _where(t < e["RTICE"], qs_in * (1.8 - 0.003 * t), qs_in)
            qcrit = crh2 * qsat
            # :196-207
            qt = q + ql + qi
            clear = qt < qcrit
            overcast = (~clear) & (qt >= qsat)
            qpd = qsat - qt
            qcd = qsat - qcrit
            clc_p = 1.0 - np.sqrt(qpd / (qcd - scalm * (qt - qcrit)))
            _count(bc, "clear", clear)
            _count(bc, "overcast", overcast)
            _count(bc, "partial", ~clear & ~overcast)
            _count(bc, "cold_fwat", t < RTT)
            _count(bc, "ice_supersaturation", t < e["RTICE"])
            if e["LPHYLIN"] or e["LDRAIN1D"]:
                _count(bc, "esdp_clipped", foeew / ap > e["ZQMAX"])
            clc = _where(clear, 0.0, _where(overcast, 1.0, clc_p))
            qc = _where(clear, 0.0, _where(
                overcast, (1.0 - scalm) * (qsat - qcrit),
                (scalm * qpd + (1.0 - scalm) * qcd) * (clc_p ** 2.0)))
            # :210-215
            gdp = RG / (in_aph[k + 1] - in_aph[k])
            lude = dt * in_lude[k] * gdp
            lo1 = (lude >= e["RLMIN"]) & (in_lu[k + 1] >= ZEPS2)
            _count(bc, "detrainment", lo1)
            _count(bc, "detrainment_without_updraught_condensate", (lude >= e["RLMIN"]) & ~lo1)
            clc = _where(lo1, clc + (1.0 - clc) * (1.0 - np.exp(-lude / in_lu[k + 1])), clc)
            qc = _where(lo1, qc + lude, qc)
            # :218-224
            rho = ap / (e["RD"] * t)
            rodqsdp = -rho * qs_in / (ap - e["RETV"] * foeew)
            ldcp = fwat * lvdcp + (1.0 - fwat) * lsdcp
            dtdzmo = RG * (1.0 / RCPD - ldcp * rodqsdp) / (1.0 + ldcp * dqsdtemp)
            dqsdz = dqsdtemp * dtdzmo - RG * rodqsdp
            dqc = np.minimum(dt * dqsdz * (in_mfu[k] + in_mfd[k]) / rho, qc)
            _count(bc, "subsidence_evaporates_all_condensate", (dqc >= qc) & (qc > 0.0))
            _count(bc, "subsidence_evaporates_part", (dqc < qc) & (dqc > 0.0))
            qc = qc - dqc
            # :227-230
            qlwc = qc * fwat
            qiwc = qc * (1.0 - fwat)
            condl = (qlwc - ql) / dt
            condi = (qiwc - qi) / dt
            # :234-235
            tmp_covptot = np.maximum(tmp_covptot, clc)
            covpclr = np.maximum(tmp_covptot - clc, 0.0)
            # :238-246
            melt = tmp_sfl != 0.0
            cons = cons2 * dp / lfdcp
            snmlt = np.minimum(tmp_sfl, cons * np.maximum(t - meltp2, 0.0))
            _count(bc, "snow_enters_level", melt)
            _count(bc, "melting", melt & (snmlt > 0.0))
            _count(bc, "melting_all_snow", melt & (snmlt > 0.0) & (snmlt >= tmp_sfl))
            _count(bc, "melting_part_of_snow", melt & (snmlt > 0.0) & (snmlt < tmp_sfl))
            rfln = _where(melt, tmp_rfl + snmlt, tmp_rfl)
            sfln = _where(melt, tmp_sfl - snmlt, tmp_sfl)
            t = _where(melt, t - snmlt / cons, t)
            # :249-272
            cloudy = clc > ZEPS2
            lcrit = 1.9 * e["RCLCRIT"] if LEV else 2.0 * e["RCLCRIT"]
            cldl = qlwc / clc
            dl = ckcodtl * (1.0 - np.exp(-((cldl / lcrit) ** 2.0)))
            prr = _where(cloudy, qlwc - clc * cldl * np.exp(-dl), 0.0)
            qlwc = _where(cloudy, qlwc - prr, qlwc)
            icrit = 0.0001 if LEV else 2.0 * e["RCLCRIT"]
            cldi = qiwc / clc
            di = ckcodti * np.exp(0.025 * (t - RTT)) * (1.0 - np.exp(-((cldi / icrit) ** 2.0)))
            prs = _where(cloudy, qiwc - clc * cldi * np.exp(-di), 0.0)
            qiwc = _where(cloudy, qiwc - prs, qiwc)
            # :275-285
            dr = cons2 * dp * (prr + prs)
            frz = t < RTT
            _count(bc, "autoconversion", cloudy)
            _count(bc, "new_precip_as_snow", frz & (dr > 0.0))
            _count(bc, "new_precip_as_rain", ~frz & (dr > 0.0))
            _count(bc, "rain_refreezes", frz & (prr > 0.0))
            rfreeze = _where(frz, cons2 * dp * prr, 0.0)
            fwatr = _where(frz, 0.0, 1.0)
            rfln = rfln + fwatr * dr
            sfln = sfln + (1.0 - fwatr) * dr
            # :288-321
            prtot = rfln + sfln
            if LEV:
                ev = (prtot > ZEPS2) & (covpclr > ZEPS2)
                preclr = prtot * covpclr / tmp_covptot
                qe = qs_in - (qs_in - qlim) * covpclr / ((1.0 - clc) ** 2.0)
                beta = RG * e["RPECONS"] * (
                    np.sqrt(ap / tmp_aph_s) / 0.00509 * preclr / covpclr) ** 0.5777
                b = dt * beta * (qs_in - qe) / (1.0 + dt * beta * corqs)
                dtgdp = dt * RG / (in_aph[k + 1] - in_aph[k])
                dpr = np.minimum(covpclr * b / dtgdp, preclr)
                preclr = preclr - dpr
                _count(bc, "evaporation", ev)
                _count(bc, "evaporation_of_all_precip", ev & (preclr <= 0.0))
                tmp_covptot = _where(ev & (preclr <= 0.0), clc, tmp_covptot)
                out_covptot_k = _where(ev, tmp_covptot, 0.0)
                evapr = _where(ev, dpr * rfln / prtot, 0.0)
                rfln = rfln - evapr
                evaps = _where(ev, dpr * sfln / prtot, 0.0)
                sfln = sfln - evaps
            else:
                out_covptot_k = np.zeros(nx, dtype)
                evapr = np.zeros(nx, dtype)
                evaps = np.zeros(nx, dtype)
            # :328-344
            dqdt = -(condl + condi) + (in_lude[k] + evapr + evaps) * gdp
            dtdt = (lvdcp * condl + lsdcp * condi
                    - (lvdcp * evapr + lsdcp * evaps
                       + in_lude[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                       - (lsdcp - lvdcp) * rfreeze) * gdp)
            t = t + dt * dtdt
            q = q + dt * dqdt
            qold = q
            # :347
            t_pre = t
            t, q = f_cuadjtqs_nl(ap, t, q, e)
            # :350-364
            dq = np.maximum(qold - q, 0.0)
            dr2 = cons2 * dp * dq
            frz2 = t < RTT
            _count(bc, "adjustment_condenses", dq > 0.0)
            _count(bc, "adjustment_evaporates", qold < q)
            _count(bc, "adjustment_warm_branch", t_pre > RTT)
            _count(bc, "adjustment_crosses_RTT", (t_pre > RTT) != (t > RTT))
            _count(bc, "adjustment_precip_as_snow", frz2 & (dq > 0.0))
            _count(bc, "adjustment_precip_as_rain", ~frz2 & (dq > 0.0))
            rfreeze2 = _where(frz2, fwat * dr2, 0.0)
            fwatr = _where(frz2, 0.0, 1.0)
            rn = fwatr * dr2
            sn = (1.0 - fwatr) * dr2
            condl = condl + fwatr * dq / dt
            condi = condi + (1.0 - fwatr) * dq / dt
            rfln = rfln + rn
            sfln = sfln + sn
            rfreeze = rfreeze + rfreeze2
            # :367-380
            F["out_clc"][k] = clc
            F["out_covptot"][k] = out_covptot_k
            F["out_tnd_q"][k] = -(condl + condi) + (in_lude[k] + evapr + evaps) * gdp
            F["out_tnd_t"][k] = (lvdcp * condl + lsdcp * condi
                                 - (lvdcp * evapr + lsdcp * evaps
                                    + in_lude[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                                    - (lsdcp - lvdcp) * rfreeze) * gdp)
            F["out_tnd_ql"][k] = (qlwc - ql) / dt
            F["out_tnd_qi"][k] = (qiwc - qi) / dt
            # :383-388
            fplsl[k] = rfln
            fplsn[k] = sfln
            tmp_rfl = rfln
            tmp_sfl = sfln

        # :391-399
        F["out_fhpsl"][0] = 0.0
        F["out_fhpsn"][0] = 0.0
        F["out_fplsl"][1:] = fplsl
        F["out_fplsn"][1:] = fplsn
        F["out_fhpsl"][1:] = -F["out_fplsl"][1:] * e["RLVTT"]
        F["out_fhpsn"][1:] = -F["out_fplsn"][1:] * e["RLSTT"]


# --------------------------------------------------------------------------------------
# tangent-linear
# --------------------------------------------------------------------------------------
def _cuadjtqs_tl_0(ap, ap_i, t, t_i, q, q_i, z3es, z4es, z5alcp, zaldcp, e):
    """tangent_linear/_stencils/cuadjtqs.py:22-52"""
    qp = 1.0 / ap
    qp_i = -ap_i / ap ** 2.0
    foeew = e["R2ES"] * np.exp(z3es * (t - e["RTT"]) / (t - z4es))
    foeew_i = foeew * z3es * t_i * (e["RTT"] - z4es) / (t - z4es) ** 2
    qsat = qp * foeew
    qsat_i = qp_i * foeew + qp * foeew_i
    clip = qsat > e["ZQMAX"]
    qsat = _where(clip, e["ZQMAX"], qsat)
    qsat_i = _where(clip, 0.0, qsat_i)
    cor = 1.0 / (1.0 - e["RETV"] * qsat)
    cor_i = e["RETV"] * qsat_i / (1.0 - e["RETV"] * qsat) ** 2.0
    qsat_i = qsat_i * cor + qsat * cor_i
    qsat = qsat * cor
    z2s = z5alcp / (t - z4es) ** 2.0
    z2s_i = -2.0 * z5alcp * t_i / (t - z4es) ** 3.0
    cond = (q - qsat) / (1.0 + qsat * cor * z2s)
    cond_i = (q_i - qsat_i) / (1.0 + qsat * cor * z2s) - (q - qsat) * (
        qsat_i * cor * z2s + qsat * cor_i * z2s + qsat * cor * z2s_i
    ) / (1.0 + qsat * cor * z2s) ** 2.0
    t = t + zaldcp * cond
    t_i = t_i + zaldcp * cond_i
    q = q - cond
    q_i = q_i - cond_i
    return t, t_i, q, q_i


def f_cuadjtqs_tl(ap, ap_i, t, t_i, q, q_i, e):
    """tangent_linear/_stencils/cuadjtqs.py:55-84"""
    assert e["ICALL"] == 0
    warm = t > e["RTT"]
    z3es = _where(warm, e["R3LES"], e["R3IES"])
    z4es = _where(warm, e["R4LES"], e["R4IES"])
    z5alcp = _where(warm, e["R5ALVCP"], e["R5ALSCP"])
    zaldcp = _where(warm, e["RALVDCP"], e["RALSDCP"])
    t, t_i, q, q_i = _cuadjtqs_tl_0(ap, ap_i, t, t_i, q, q_i, z3es, z4es, z5alcp, zaldcp, e)
    t, t_i, q, q_i = _cuadjtqs_tl_0(ap, ap_i, t, t_i, q, q_i, z3es, z4es, z5alcp, zaldcp, e)
    return t, t_i, q, q_i


def cloudsc2_tl(fields: Dict[str, np.ndarray], in_eta: np.ndarray, dt, externals) -> None:
    """tangent_linear/_stencils/cloudsc2.py:124-774.  `fields`: the 16 `in_*`, their 16 `in_*_i`
    twins, the 10 `out_*` and 10 `out_*_i` arrays (nz+1, nx)."""
    e = externals
    F = fields
    in_ap, in_ap_i, in_aph, in_aph_i = F["in_ap"], F["in_ap_i"], F["in_aph"], F["in_aph_i"]
    in_lu, in_lu_i, in_lude, in_lude_i = F["in_lu"], F["in_lu_i"], F["in_lude"], F["in_lude_i"]
    in_mfd, in_mfd_i, in_mfu, in_mfu_i = F["in_mfd"], F["in_mfd_i"], F["in_mfu"], F["in_mfu_i"]
    in_qsat, in_qsat_i = F["in_qsat"], F["in_qsat_i"]
    dtype = in_ap.dtype
    _WORK["dtype"] = dtype
    nz = in_ap.shape[0] - 1
    nx = in_ap.shape[1]
    dt = dtype.type(dt)
    eta = in_eta
    LEV = e["LEVAPLS2"] or e["LDRAIN1D"]
    LREGCL = e["LREGCL"]
    NLEV = e.get("NLEV", nz)
    RG, RTT, RCPD, RETV, RD = e["RG"], e["RTT"], e["RCPD"], e["RETV"], e["RD"]
    RLVTT, RLSTT, RLMLT, RVTMP2 = e["RLVTT"], e["RLSTT"], e["RLMLT"], e["RVTMP2"]
    ZEPS1, ZEPS2, ZQMAX = e["ZEPS1"], e["ZEPS2"], e["ZQMAX"]
    z = lambda: np.zeros(nx, dtype)  # noqa: E731

    with np.errstate(**_ERR):
        # :124-135
        tmp_rfl, tmp_rfl_i, tmp_sfl, tmp_sfl_i = z(), z(), z(), z()
        tmp_covptot, tmp_covptot_i = z(), z()
        tmp_aph_s = in_aph[nz].copy()
        tmp_aph_s_i = in_aph_i[nz].copy()
        # :137-140
        t3d = F["in_t"][:nz] + dt * F["in_tnd_cml_t"][:nz]
        t3d_i = F["in_t_i"][:nz] + dt * F["in_tnd_cml_t_i"][:nz]
        # :143-147
        tmp_trpaus = _trpaus(eta, t3d, nz, dtype)

        fplsl = np.zeros((nz, nx), dtype)
        fplsl_i = np.zeros((nz, nx), dtype)
        fplsn = np.zeros((nz, nx), dtype)
        fplsn_i = np.zeros((nz, nx), dtype)

        for k in range(nz):
            t, t_i = t3d[k], t3d_i[k]
            ap, ap_i = in_ap[k], in_ap_i[k]
            qs_in, qs_in_i = in_qsat[k], in_qsat_i[k]
            # :151-156
            q = F["in_q"][k] + dt * F["in_tnd_cml_q"][k] + F["in_supsat"][k]
            q_i = F["in_q_i"][k] + dt * F["in_tnd_cml_q_i"][k] + F["in_supsat_i"][k]
            ql = F["in_ql"][k] + dt * F["in_tnd_cml_ql"][k]
            ql_i = F["in_ql_i"][k] + dt * F["in_tnd_cml_ql_i"][k]
            qi = F["in_qi"][k] + dt * F["in_tnd_cml_qi"][k]
            qi_i = F["in_qi_i"][k] + dt * F["in_tnd_cml_qi_i"][k]
            # :159-165
            ckcodtl = 2.0 * e["RKCONV"] * dt
            ckcodti = 5.0 * e["RKCONV"] * dt
            ckcodtla = ckcodtl / 100.0
            ckcodtia = ckcodti / 100.0
            cons2 = 1.0 / (RG * dt)
            cons3 = RLVTT / RCPD
            meltp2 = RTT + 2.0
            # :168
            scalm = e["ZSCAL"] * max(eta[k] - 0.2, ZEPS1) ** 0.2
            # :171-180
            dp = in_aph[k + 1] - in_aph[k]
            dp_i = in_aph_i[k + 1] - in_aph_i[k]
            zz = 1.0 / (RCPD + RCPD * RVTMP2 * q)
            zz_i = -RCPD * RVTMP2 * q_i / (RCPD + RCPD * RVTMP2 * q) ** 2.0
            lfdcp, lfdcp_i = RLMLT * zz, RLMLT * zz_i
            lsdcp, lsdcp_i = RLSTT * zz, RLSTT * zz_i
            lvdcp, lvdcp_i = RLVTT * zz, RLVTT * zz_i
            # :189-205
            cold = t < RTT
            fwat = _where(cold, 0.545 * (np.tanh(0.17 * (t - e["RLPTRC"])) + 1.0), 1.0)
            fwat_i = _where(cold, 0.545 * 0.17 * t_i / np.cosh(0.17 * (t - e["RLPTRC"])) ** 2.0, 0.0)
            z3es = _where(cold, e["R3IES"], e["R3LES"])
            z4es = _where(cold, e["R4IES"], e["R4LES"])
            foeew = e["R2ES"] * np.exp(z3es * (t - RTT) / (t - z4es))
            foeew_i = z3es * (RTT - z4es) * t_i * foeew / (t - z4es) ** 2.0
            esdp = foeew / ap
            esdp_i = foeew_i / ap - foeew * ap_i / (ap ** 2.0)
            clip = esdp > ZQMAX
            esdp = _where(clip, ZQMAX, esdp)
            esdp_i = _where(clip, 0.0, esdp_i)
            # :207-222
            facw = e["R5LES"] / (t - e["R4LES"]) ** 2.0
            facw_i = -2.0 * e["R5LES"] * t_i / (t - e["R4LES"]) ** 3.0
            faci = e["R5IES"] / (t - e["R4IES"]) ** 2.0
            faci_i = -2.0 * e["R5IES"] * t_i / (t - e["R4IES"]) ** 3.0
            fac = fwat * facw + (1.0 - fwat) * faci
            fac_i = fwat_i * (facw - faci) + fwat * facw_i + (1.0 - fwat) * faci_i
            cor = 1.0 / (1.0 - RETV * esdp)
            cor_i = RETV * esdp_i / (1.0 - RETV * esdp) ** 2.0
            dqsdtemp = fac * cor * qs_in
            dqsdtemp_i = fac_i * cor * qs_in + fac * cor_i * qs_in + fac * cor * qs_in_i
            corqs = 1.0 + cons3 * dqsdtemp
            corqs_i = cons3 * dqsdtemp_i
            # :225-230
            qgt = q > qs_in
            qlim = _where(qgt, qs_in, q)
            qlim_i = _where(qgt, qs_in_i, q_i)
            # :233-253
            crh2 = _crh2(eta[k], tmp_trpaus)
            # :256-265
            vcold = t < e["RTICE"]
            supsat = _where(vcold, 1.8 - 0.003 * t, 1.0)
            supsat_i = _where(vcold, -0.003 * t_i, 0.0)
            qsat = qs_in * supsat
            qsat_i = qs_in_i * supsat + qs_in * supsat_i
            qcrit = crh2 * qsat
            qcrit_i = crh2 * qsat_i
            # :268-306
            qt = q + ql + qi
            qt_i = q_i + ql_i + qi_i
            clear = qt < qcrit
            overcast = (~clear) & (qt >= qsat)
            partial = (~clear) & (~overcast)
            qpd = qsat - qt
            qpd_i = qsat_i - qt_i
            qcd = qsat - qcrit
            qcd_i = qsat_i - qcrit_i
            den = qcd - scalm * (qt - qcrit)
            tmp1 = np.sqrt(qpd / den)
            clc_p = 1.0 - tmp1
            clc_p_i = (-0.5 / tmp1
                       * (qpd_i * den - qpd * (qcd_i - scalm * (qt_i - qcrit_i)))
                       / den ** 2.0)
            if LREGCL:
                rat = qpd / qcd
                yyy = np.minimum(0.3, 3.5 * np.sqrt(rat * (1.0 - scalm * (1.0 - rat)) ** 3.0) / (1.0 - scalm))
                clc_p_i = clc_p_i * yyy
            qc_p = (scalm * qpd + (1.0 - scalm) * qcd) * clc_p ** 2.0
            qc_p_i = ((scalm * qpd_i + (1.0 - scalm) * qcd_i) * clc_p ** 2.0
                      + 2.0 * (scalm * qpd + (1.0 - scalm) * qcd) * clc_p * clc_p_i)
            clc = _where(clear, 0.0, _where(overcast, 1.0, clc_p))
            clc_i = _where(partial, clc_p_i, 0.0)
            qc = _where(clear, 0.0, _where(overcast, (1.0 - scalm) * (qsat - qcrit), qc_p))
            qc_i = _where(clear, 0.0, _where(overcast, (1.0 - scalm) * (qsat_i - qcrit_i), qc_p_i))
            # :309-325
            gdp = RG / (in_aph[k + 1] - in_aph[k])
            gdp_i = -RG * (in_aph_i[k + 1] - in_aph_i[k]) / (in_aph[k + 1] - in_aph[k]) ** 2.0
            lude = dt * in_lude[k] * gdp
            lude_i = dt * (in_lude_i[k] * gdp + in_lude[k] * gdp_i)
            lo1 = (k < NLEV - 1) & (lude >= e["RLMIN"]) & (in_lu[k + 1] >= ZEPS2)
            tmp2 = np.exp(-lude / in_lu[k + 1])
            clc_i = _where(
                lo1,
                clc_i + (-clc_i * (1 - tmp2) + (1.0 - clc) * tmp2
                         * (lude_i / in_lu[k + 1] - lude * in_lu_i[k + 1] / in_lu[k + 1] ** 2.0)),
                clc_i)
            clc = _where(lo1, clc + (1.0 - clc) * (1.0 - tmp2), clc)
            qc = _where(lo1, qc + lude, qc)
            qc_i = _where(lo1, qc_i + lude_i, qc_i)
            # :328-354
            fac1 = 1.0 / (RD * t)
            rho = ap * fac1
            rho_i = (ap_i - ap * t_i / t) * fac1
            fac2 = 1.0 / (ap - RETV * foeew)
            rodqsdp = -rho * qs_in * fac2
            rodqsdp_i = (-rho_i * qs_in - rho * qs_in_i
                         + rho * qs_in * (ap_i - RETV * foeew_i) * fac2) * fac2
            ldcp = fwat * lvdcp + (1.0 - fwat) * lsdcp
            ldcp_i = fwat_i * (lvdcp - lsdcp) + fwat * lvdcp_i + (1.0 - fwat) * lsdcp_i
            fac3 = 1.0 / (1.0 + ldcp * dqsdtemp)
            dtdzmo = RG * (1.0 / RCPD - ldcp * rodqsdp) * fac3
            dtdzmo_i = (-(RG * (ldcp_i * rodqsdp + ldcp * rodqsdp_i)
                          + dtdzmo * (ldcp_i * dqsdtemp + ldcp * dqsdtemp_i)) * fac3)
            dqsdz = dqsdtemp * dtdzmo - RG * rodqsdp
            dqsdz_i = dqsdtemp_i * dtdzmo + dqsdtemp * dtdzmo_i - RG * rodqsdp_i
            # :356-373
            tmp3 = dt * dqsdz * (in_mfu[k] + in_mfd[k]) / rho
            lo3 = tmp3 < qc
            dqc_a_i = (dt * (dqsdz_i * (in_mfu[k] + in_mfd[k]) + dqsdz * (in_mfu_i[k] + in_mfd_i[k]))
                       - tmp3 * rho_i) / rho
            if LREGCL:
                dqc_a_i = dqc_a_i * 0.1
            dqc = _where(lo3, tmp3, qc)
            dqc_i = _where(lo3, dqc_a_i, qc_i)
            qc = qc - dqc
            qc_i = qc_i - dqc_i
            # :376-386
            qlwc = qc * fwat
            qlwc_i = qc_i * fwat + qc * fwat_i
            qiwc = qc * (1.0 - fwat)
            qiwc_i = qc_i * (1.0 - fwat) - qc * fwat_i
            condl = (qlwc - ql) / dt
            condl_i = (qlwc_i - ql_i) / dt
            condi = (qiwc - qi) / dt
            condi_i = (qiwc_i - qi_i) / dt
            # :390-397
            up = clc > tmp_covptot
            tmp_covptot = _where(up, clc, tmp_covptot)
            tmp_covptot_i = _where(up, clc_i, tmp_covptot_i)
            covpclr = tmp_covptot - clc
            covpclr_i = tmp_covptot_i - clc_i
            neg = covpclr < 0.0
            covpclr = _where(neg, 0.0, covpclr)
            covpclr_i = _where(neg, 0.0, covpclr_i)
            # :400-427
            melt = tmp_sfl != 0.0
            cons = cons2 * dp / lfdcp
            cons_i = cons2 * (dp_i * lfdcp - dp * lfdcp_i) / lfdcp ** 2
            warm = t > meltp2
            z2s = _where(warm, cons * (t - meltp2), 0.0)
            z2s_i = _where(warm, cons_i * (t - meltp2) + cons * t_i, 0.0)
            allm = tmp_sfl <= z2s
            snmlt = _where(allm, tmp_sfl, z2s)
            snmlt_i = _where(allm, tmp_sfl_i, z2s_i)
            rfln = _where(melt, tmp_rfl + snmlt, tmp_rfl)
            rfln_i = _where(melt, tmp_rfl_i + snmlt_i, tmp_rfl_i)
            sfln = _where(melt, tmp_sfl - snmlt, tmp_sfl)
            sfln_i = _where(melt, tmp_sfl_i - snmlt_i, tmp_sfl_i)
            t_i = _where(melt, t_i - (snmlt_i * cons - snmlt * cons_i) / cons ** 2, t_i)
            t = _where(melt, t - snmlt / cons, t)
            # :429-503
            cloudy = clc > ZEPS2
            lcrit = 1.9 * e["RCLCRIT"] if LEV else 2.0 * e["RCLCRIT"]
            cldl = qlwc / clc
            cldl_i = qlwc_i / clc - qlwc * clc_i / clc ** 2.0
            ltmp4 = np.exp(-((cldl / lcrit) ** 2.0))
            dl = ckcodtl * (1.0 - ltmp4)
            ltmp5 = np.exp(-dl)
            if LREGCL:
                dl_i = (2.0 * ckcodtla / lcrit ** 2.0) * ltmp4 * cldl * cldl_i
            else:
                dl_i = (2.0 * ckcodtl / lcrit ** 2.0) * ltmp4 * cldl * cldl_i
            qlnew = clc * cldl * ltmp5
            qlnew_i = clc_i * cldl * ltmp5 + clc * cldl_i * ltmp5 - clc * cldl * ltmp5 * dl_i
            prr = _where(cloudy, qlwc - qlnew, 0.0)
            prr_i = _where(cloudy, qlwc_i - qlnew_i, 0.0)
            qlwc = _where(cloudy, qlwc - prr, qlwc)
            qlwc_i = _where(cloudy, qlwc_i - prr_i, qlwc_i)
            icrit = 0.0001 if LEV else 2.0 * e["RCLCRIT"]
            cldi = qiwc / clc
            cldi_i = qiwc_i / clc - qiwc * clc_i / clc ** 2.0
            itmp41 = np.exp(-((cldi / icrit) ** 2.0))
            itmp42 = np.exp(0.025 * (t - RTT))
            di = ckcodti * itmp42 * (1.0 - itmp41)
            itmp5 = np.exp(-di)
            di_i = ((ckcodtia if LREGCL else ckcodti) * itmp42
                    * (itmp41 * (2.0 * cldi * cldi_i / icrit ** 2.0 - 0.025 * t_i) + 0.025 * t_i))
            qinew = clc * cldi * itmp5
            qinew_i = clc_i * cldi * itmp5 + clc * cldi_i * itmp5 - clc * cldi * itmp5 * di_i
            prs = _where(cloudy, qiwc - qinew, 0.0)
            prs_i = _where(cloudy, qiwc_i - qinew_i, 0.0)
            qiwc = _where(cloudy, qiwc - prs, qiwc)
            qiwc_i = _where(cloudy, qiwc_i - prs_i, qiwc_i)
            # :506-523
            dr = cons2 * dp * (prr + prs)
            dr_i = cons2 * (dp_i * (prr + prs) + dp * (prr_i + prs_i))
            frz = t < RTT
            rfreeze = _where(frz, cons2 * dp * prr, 0.0)
            rfreeze_i = _where(frz, cons2 * (dp_i * prr + dp * prr_i), 0.0)
            fwatr = _where(frz, 0.0, 1.0)
            fwatr_i = 0.0
            rfln = rfln + fwatr * dr
            rfln_i = rfln_i + fwatr_i * dr + fwatr * dr_i
            sfln = sfln + (1.0 - fwatr) * dr
            sfln_i = sfln_i + (-fwatr_i * dr + (1.0 - fwatr) * dr_i)
            # :526-616
            prtot = rfln + sfln
            prtot_i = rfln_i + sfln_i
            if LEV:
                ev = (prtot > ZEPS2) & (covpclr > ZEPS2)
                preclr = prtot * covpclr / tmp_covptot
                preclr_i = ((prtot_i * covpclr + prtot * covpclr_i) / tmp_covptot
                            - prtot * covpclr * tmp_covptot_i / tmp_covptot ** 2.0)
                qe = qs_in - (qs_in - qlim) * covpclr / (1.0 - clc) ** 2.0
                qe_i = (qs_in_i
                        - (qs_in_i * covpclr - qlim_i * covpclr + (qs_in - qlim) * covpclr_i)
                        / (1.0 - clc) ** 2.0
                        - 2.0 * (qs_in - qlim) * covpclr * clc_i / (1.0 - clc) ** 3.0)
                tmp6 = np.sqrt(ap / tmp_aph_s)
                beta = RG * e["RPECONS"] * (tmp6 * preclr / (0.00509 * covpclr)) ** 0.5777
                beta_i = (0.5777 * RG * e["RPECONS"] / 0.00509
                          * (0.00509 * covpclr / (tmp6 * preclr)) ** 0.4223
                          * ((tmp6 * preclr_i + 0.5 * preclr * ap_i / tmp6
                              - 0.5 * preclr * tmp6 * tmp_aph_s_i / tmp_aph_s) / covpclr
                             - tmp6 * preclr * covpclr_i / covpclr ** 2))
                b = dt * beta * (qs_in - qe) / (1.0 + dt * beta * corqs)
                b_i = (dt * (beta_i * (qs_in - qe) + beta * (qs_in_i - qe_i)) / (1.0 + dt * beta * corqs)
                       - dt ** 2.0 * b * (beta_i * corqs + beta * corqs_i) / (1 + dt * beta * corqs))
                dtgdp = dt * RG / (in_aph[k + 1] - in_aph[k])
                dtgdp_i = (-dt * RG * (in_aph_i[k + 1] - in_aph_i[k])
                           / (in_aph[k + 1] - in_aph[k]) ** 2.0)
                dpr = covpclr * b / dtgdp
                dpr_i = (covpclr_i * b + covpclr * b_i) / dtgdp - covpclr * b * dtgdp_i / dtgdp ** 2
                cap = dpr > preclr
                dpr = _where(cap, preclr, dpr)
                dpr_i = _where(cap, preclr_i, dpr_i)
                preclr = preclr - dpr
                preclr_i = preclr_i - dpr_i
                reset = ev & (preclr <= 0.0)
                tmp_covptot = _where(reset, clc, tmp_covptot)
                tmp_covptot_i = _where(reset, clc_i, tmp_covptot_i)
                out_covptot_k = _where(ev, tmp_covptot, 0.0)
                out_covptot_k_i = _where(ev, tmp_covptot_i, 0.0)
                evapr = _where(ev, dpr * rfln / prtot, 0.0)
                evapr_i = _where(ev, (dpr_i * rfln + dpr * rfln_i) / prtot
                                   - dpr * rfln * prtot_i / prtot ** 2, 0.0)
                rfln = rfln - evapr
                rfln_i = rfln_i - evapr_i
                evaps = _where(ev, dpr * sfln / prtot, 0.0)
                evaps_i = _where(ev, (dpr_i * sfln + dpr * sfln_i) / prtot
                                   - dpr * sfln * prtot_i / prtot ** 2, 0.0)
                sfln = sfln - evaps
                sfln_i = sfln_i - evaps_i
            else:
                out_covptot_k, out_covptot_k_i = z(), z()
                evapr, evapr_i, evaps, evaps_i = z(), z(), z(), z()
            # :619-659
            dqdt = -(condl + condi) + (in_lude[k] + evapr + evaps) * gdp
            dqdt_i = (-(condl_i + condi_i) + (in_lude_i[k] + evapr_i + evaps_i) * gdp
                      + (in_lude[k] + evapr + evaps) * gdp_i)
            tmp7 = (lvdcp * evapr + lsdcp * evaps
                    + in_lude[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                    - (lsdcp - lvdcp) * rfreeze)
            dtdt = lvdcp * condl + lsdcp * condi - tmp7 * gdp
            dtdt_i = (lvdcp_i * condl + lvdcp * condl_i + lsdcp_i * condi + lsdcp * condi_i
                      - (lvdcp_i * evapr + lvdcp * evapr_i + lsdcp_i * evaps + lsdcp * evaps_i
                         + in_lude_i[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                         + in_lude[k] * (fwat_i * (lvdcp - lsdcp) + fwat * lvdcp_i + (1.0 - fwat) * lsdcp_i)
                         - (lsdcp_i - lvdcp_i) * rfreeze - (lsdcp - lvdcp) * rfreeze_i) * gdp
                      - tmp7 * gdp_i)
            t = t + dt * dtdt
            t_i = t_i + dt * dtdt_i
            q = q + dt * dqdt
            q_i = q_i + dt * dqdt_i
            qold, qold_i = q, q_i
            # :662
            t, t_i, q, q_i = f_cuadjtqs_tl(ap, ap_i, t, t_i, q, q_i, e)
            # :664-673
            pos = qold >= q
            dq = _where(pos, qold - q, 0.0)
            dq_i = _where(pos, (qold_i - q_i) * (0.7 if LREGCL else 1.0), 0.0)
            dr2 = cons2 * dp * dq
            dr2_i = cons2 * (dp_i * dq + dp * dq_i)
            # :677-703
            frz2 = t < RTT
            rfreeze2 = _where(frz2, fwat * dr2, 0.0)
            rfreeze2_i = _where(frz2, fwat_i * dr2 + fwat * dr2_i, 0.0)
            fwatr = _where(frz2, 0.0, 1.0)
            fwatr_i = 0.0
            rn = fwatr * dr2
            rn_i = fwatr_i * dr2 + fwatr * dr2_i
            sn = (1.0 - fwatr) * dr2
            sn_i = -fwatr_i * dr2 + (1.0 - fwatr) * dr2_i
            condl = condl + fwatr * dq / dt
            condl_i = condl_i + (fwatr_i * dq + fwatr * dq_i) / dt
            condi = condi + (1.0 - fwatr) * dq / dt
            condi_i = condi_i + (-fwatr_i * dq + (1.0 - fwatr) * dq_i) / dt
            rfln = rfln + rn
            rfln_i = rfln_i + rn_i
            sfln = sfln + sn
            sfln_i = sfln_i + sn_i
            rfreeze = rfreeze + rfreeze2
            rfreeze_i = rfreeze_i + rfreeze2_i
            # :706-741
            F["out_clc"][k] = clc
            F["out_clc_i"][k] = clc_i
            F["out_covptot"][k] = out_covptot_k
            F["out_covptot_i"][k] = out_covptot_k_i
            F["out_tnd_q"][k] = -(condl + condi) + (in_lude[k] + evapr + evaps) * gdp
            F["out_tnd_q_i"][k] = (-(condl_i + condi_i) + (in_lude_i[k] + evapr_i + evaps_i) * gdp
                                   + (in_lude[k] + evapr + evaps) * gdp_i)
            tmp8 = (lvdcp * evapr + lsdcp * evaps
                    + in_lude[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                    - (lsdcp - lvdcp) * rfreeze)
            F["out_tnd_t"][k] = lvdcp * condl + lsdcp * condi - tmp8 * gdp
            F["out_tnd_t_i"][k] = (
                lvdcp_i * condl + lvdcp * condl_i + lsdcp_i * condi + lsdcp * condi_i
                - (lvdcp_i * evapr + lvdcp * evapr_i + lsdcp_i * evaps + lsdcp * evaps_i
                   + in_lude_i[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                   + in_lude[k] * (fwat_i * (lvdcp - lsdcp) + fwat * lvdcp_i + (1.0 - fwat) * lsdcp_i)
                   - (lsdcp_i - lvdcp_i) * rfreeze - (lsdcp - lvdcp) * rfreeze_i) * gdp
                - tmp8 * gdp_i)
            F["out_tnd_ql"][k] = (qlwc - ql) / dt
            F["out_tnd_ql_i"][k] = (qlwc_i - ql_i) / dt
            F["out_tnd_qi"][k] = (qiwc - qi) / dt
            F["out_tnd_qi_i"][k] = (qiwc_i - qi_i) / dt
            # :744-753
            fplsl[k], fplsl_i[k], fplsn[k], fplsn_i[k] = rfln, rfln_i, sfln, sfln_i
            tmp_rfl, tmp_rfl_i, tmp_sfl, tmp_sfl_i = rfln, rfln_i, sfln, sfln_i

        # :756-774
        for n in ("fplsl", "fplsn", "fhpsl", "fhpsn"):
            F["out_" + n][0] = 0.0
            F["out_" + n + "_i"][0] = 0.0
        F["out_fplsl"][1:] = fplsl
        F["out_fplsl_i"][1:] = fplsl_i
        F["out_fplsn"][1:] = fplsn
        F["out_fplsn_i"][1:] = fplsn_i
        F["out_fhpsl"][1:] = -F["out_fplsl"][1:] * RLVTT
        F["out_fhpsl_i"][1:] = -F["out_fplsl_i"][1:] * RLVTT
        F["out_fhpsn"][1:] = -F["out_fplsn"][1:] * RLSTT
        F["out_fhpsn_i"][1:] = -F["out_fplsn_i"][1:] * RLSTT


# --------------------------------------------------------------------------------------
# adjoint
# --------------------------------------------------------------------------------------
def f_cuadjtqs_ad(ap, ap_i, t, t_i, q, q_i, e):
    """adjoint/_stencils/cuadjtqs.py:22-158: recompute both adjustment iterations keeping the
    intermediates, then reverse them.  Returns (ap_i, t, t_i, q, q_i)."""
    assert e["ICALL"] == 0
    R2ES, RETV, RTT, ZQMAX = e["R2ES"], e["RETV"], e["RTT"], e["ZQMAX"]
    warm = t > RTT
    z3es = _where(warm, e["R3LES"], e["R3IES"])
    z4es = _where(warm, e["R4LES"], e["R4IES"])
    z5alcp = _where(warm, e["R5ALVCP"], e["R5ALSCP"])
    zaldcp = _where(warm, e["RALVDCP"], e["RALSDCP"])

    # :53-71 first iteration ("b")
    targ = t
    foeew = R2ES * np.exp(z3es * (targ - RTT) / (targ - z4es))
    foeew_b = foeew
    qsat = foeew / ap
    ltest2 = qsat > ZQMAX
    qsat = _where(ltest2, ZQMAX, qsat)
    cor = 1.0 / (1.0 - RETV * qsat)
    qsat_d = qsat
    qsat = qsat * cor
    targ_b = targ
    z2s = z5alcp / (targ - z4es) ** 2.0
    qsat_b, cor_b, z2s_b, q_b = qsat, cor, z2s, q
    cond1 = (q - qsat) / (1.0 + qsat * cor * z2s)
    t = t + zaldcp * cond1
    q = q - cond1
    # :73-91 second iteration ("a")
    targ = t
    foeew = R2ES * np.exp(z3es * (targ - RTT) / (targ - z4es))
    foeew_a = foeew
    qsat = foeew / ap
    ltest1 = qsat > ZQMAX
    qsat = _where(ltest1, ZQMAX, qsat)
    cor = 1.0 / (1.0 - RETV * qsat)
    qsat_c = qsat
    qsat = qsat * cor
    targ_a = targ
    z2s = z5alcp / (targ - z4es) ** 2.0
    qsat_a, cor_a, z2s_a, q_a = qsat, cor, z2s, q
    cond1 = (q - qsat) / (1.0 + qsat * cor * z2s)
    t = t + zaldcp * cond1
    q = q - cond1

    # :93-124 reverse of the second iteration
    cond1_i = -q_i + zaldcp * t_i
    qsat, cor, z2s = qsat_a, cor_a, z2s_a
    q_i = q_i + cond1_i / (1.0 + qsat * cor * z2s)
    qsat_i = (-cond1_i / (1.0 + qsat * cor * z2s)
              - cond1_i * (q_a - qsat) * cor * z2s / (1.0 + qsat * cor * z2s) ** 2.0)
    cor_i = -cond1_i * (q_a - qsat) * qsat * z2s / (1.0 + qsat * cor * z2s) ** 2.0
    z2s_i = -cond1_i * (q_a - qsat) * qsat * cor / (1.0 + qsat * cor * z2s) ** 2.0
    targ = targ_a
    targ_i = -2.0 * z2s_i * z5alcp / (targ - z4es) ** 3.0
    qsat = qsat_c
    cor_i = cor_i + qsat_i * qsat
    qsat_i = qsat_i * cor
    qsat_i = qsat_i + cor_i * RETV / (1.0 - RETV * qsat) ** 2.0
    qsat_i = _where(ltest1, 0.0, qsat_i)
    foeew_i = qsat_i / ap
    foeew = foeew_a
    qp_i = qsat_i * foeew
    targ_i = targ_i + (foeew_i * R2ES * z3es * (RTT - z4es)
                       * np.exp(z3es * (targ - RTT) / (targ - z4es)) / (targ - z4es) ** 2.0)
    t_i = t_i + targ_i
    # :126-156 reverse of the first iteration
    cond1_i = -q_i + zaldcp * t_i
    qsat, cor, z2s = qsat_b, cor_b, z2s_b
    q_i = q_i + cond1_i / (1.0 + qsat * cor * z2s)
    qsat_i = (-cond1_i / (1.0 + qsat * cor * z2s)
              - cond1_i * (q_b - qsat) * cor * z2s / (1.0 + qsat * cor * z2s) ** 2.0)
    cor_i = -cond1_i * (q_b - qsat) * qsat * z2s / (1.0 + qsat * cor * z2s) ** 2.0
    z2s_i = -cond1_i * (q_b - qsat) * qsat * cor / (1.0 + qsat * cor * z2s) ** 2.0
    targ = targ_b
    targ_i = -2.0 * z2s_i * z5alcp / (targ - z4es) ** 3.0
    qsat = qsat_d
    cor_i = cor_i + qsat_i * qsat
    qsat_i = qsat_i * cor
    qsat_i = qsat_i + cor_i * RETV / (1.0 - RETV * qsat) ** 2.0
    qsat_i = _where(ltest2, 0.0, qsat_i)
    foeew_i = qsat_i / ap
    foeew = foeew_b
    qp_i = qp_i + qsat_i * foeew
    targ_i = targ_i + (foeew_i * R2ES * z3es * (RTT - z4es)
                       * np.exp(z3es * (targ - RTT) / (targ - z4es)) / (targ - z4es) ** 2.0)
    t_i = t_i + targ_i
    ap_i = ap_i - qp_i / ap ** 2.0
    return ap_i, t, t_i, q, q_i


AD_ADJ_IN = tuple("in_" + n + "_i" for n in
                  ("clc", "covptot", "fhpsl", "fhpsn", "fplsl", "fplsn", "tnd_q", "tnd_qi", "tnd_ql", "tnd_t"))
AD_ADJ_OUT = tuple("out_" + n[3:] + "_i" for n in NL_INPUTS)


def cloudsc2_ad(fields: Dict[str, np.ndarray], in_eta: np.ndarray, dt, externals) -> None:
    """adjoint/_stencils/cloudsc2.py:124-996.

    `fields`: the 16 trajectory inputs `in_*`, the 10 adjoint forcings `in_{clc,...}_i`
    (NOT modified here; the reference zeroes them in place, Q1), the 10 NL outputs `out_*` and the
    16 adjoint outputs `out_{ap,aph,...,tnd_cml_t}_i`.  Outputs the reference only accumulates into
    (`out_lude_i`, :526) start from 0.  Temporaries never assigned at a level read as 0 (Q8)."""
    e = externals
    F = fields
    in_ap, in_aph, in_lu, in_lude = F["in_ap"], F["in_aph"], F["in_lu"], F["in_lude"]
    in_mfd, in_mfu, in_qsat = F["in_mfd"], F["in_mfu"], F["in_qsat"]
    dtype = in_ap.dtype
    _WORK["dtype"] = dtype
    nz = in_ap.shape[0] - 1
    nx = in_ap.shape[1]
    dt = dtype.type(dt)
    eta = in_eta
    LEV = e["LEVAPLS2"] or e["LDRAIN1D"]
    LREGCL = e["LREGCL"]
    NLEV = e.get("NLEV", nz)
    # build extension (include/cloudsc2_hip.h): use the NL/TL freezing tests instead of Q4/Q5
    FIX = bool(e.get("AD_TRAJ_FIX", 0))
    RG, RTT, RCPD, RETV, RD = e["RG"], e["RTT"], e["RCPD"], e["RETV"], e["RD"]
    RLVTT, RLSTT, RLMLT, RVTMP2 = e["RLVTT"], e["RLSTT"], e["RLMLT"], e["RVTMP2"]
    ZEPS1, ZEPS2, ZQMAX = e["ZEPS1"], e["ZEPS2"], e["ZQMAX"]

    class Traj(dict):
        """3-D temporaries of the stencil: zero-initialised (nz+1, nx) arrays created on first use."""

        def __missing__(self, key):
            self[key] = np.zeros((nz + 1, nx), dtype)
            return self[key]

    T = Traj()

    with np.errstate(**_ERR):
        # ------------------------------------------------------------------ forward (:124-475)
        tmp_covptotp = np.zeros(nx, dtype)
        tmp_rfln = np.zeros(nx, dtype)
        tmp_sfln = np.zeros(nx, dtype)
        tmp_aph_s = in_aph[nz].copy()
        t3d = F["in_t"][:nz] + dt * F["in_tnd_cml_t"][:nz]      # :135
        T["t2"][:nz] = t3d                                      # :137
        tmp_trpaus = _trpaus(eta, t3d, nz, dtype)               # :140-144

        ckcodtl = 2.0 * e["RKCONV"] * dt
        ckcodti = 5.0 * e["RKCONV"] * dt
        cons2 = 1.0 / (RG * dt)
        cons3 = RLVTT / RCPD
        meltp2 = RTT + 2.0
        lcrit = 1.9 * e["RCLCRIT"] if LEV else 2.0 * e["RCLCRIT"]
        icrit = 0.0001 if LEV else 2.0 * e["RCLCRIT"]
        scalm_k = np.array([e["ZSCAL"] * max(eta[k] - 0.2, ZEPS1) ** 0.2 for k in range(nz)], dtype)

        for k in range(nz):
            ap, qs_in = in_ap[k], in_qsat[k]
            t = t3d[k]
            t2 = t
            rfl = tmp_rfln                                      # :149-150
            sfl = tmp_sfln
            q = F["in_q"][k] + dt * F["in_tnd_cml_q"][k] + F["in_supsat"][k]
            ql = F["in_ql"][k] + dt * F["in_tnd_cml_ql"][k]
            qi = F["in_qi"][k] + dt * F["in_tnd_cml_qi"][k]
            q2 = q
            scalm = scalm_k[k]
            dp = in_aph[k + 1] - in_aph[k]
            zz = RCPD + RCPD * RVTMP2 * q
            lfdcp, lsdcp, lvdcp = RLMLT / zz, RLSTT / zz, RLVTT / zz
            # :181-197
            cold = t < RTT
            fwat = _where(cold, 0.545 * (np.tanh(0.17 * (t2 - e["RLPTRC"])) + 1.0), 1.0)
            z3es = _where(cold, e["R3IES"], e["R3LES"])
            z4es = _where(cold, e["R4IES"], e["R4LES"])
            foeew = e["R2ES"] * np.exp(z3es * (t2 - RTT) / (t2 - z4es))
            esdp1 = foeew / ap
            esdp = np.minimum(esdp1, ZQMAX)
            facw = e["R5LES"] / (t2 - e["R4LES"]) ** 2.0
            faci = e["R5IES"] / (t2 - e["R4IES"]) ** 2.0
            fac = fwat * facw + (1.0 - fwat) * faci
            cor = 1.0 / (1.0 - RETV * esdp)
            dqsdtemp = fac * cor * qs_in
            corqs = 1.0 + cons3 * dqsdtemp
            qlim = np.minimum(q2, qs_in)                        # :200
            crh2 = _crh2(eta[k], tmp_trpaus)                    # :203-223
            supsat = _where(t2 < e["RTICE"], 1.8 - 0.003 * t2, 1.0)   # :226-231
            qsat = qs_in * supsat
            qcrit = crh2 * qsat
            # :234-252
            qt = q + ql + qi
            clear = qt <= qcrit
            overcast = (~clear) & (qt >= qsat)
            partial = (~clear) & (~overcast)
            qcd = _where(partial, qsat - qcrit, 0.0)
            qpd = _where(partial, qsat - qt, 0.0)
            tmp3 = _where(partial, np.sqrt(qpd / (qcd - scalm * (qt - qcrit))), 0.0)
            clc = _where(clear, 0.0, _where(overcast, 1.0, 1.0 - tmp3))
            qc1 = _where(clear, 0.0, _where(overcast, (1.0 - scalm) * (qsat - qcrit),
                                                (scalm * qpd + (1.0 - scalm) * qcd) * clc ** 2.0))
            # :255-263
            gdp = RG / (in_aph[k + 1] - in_aph[k])
            lude = dt * in_lude[k] * gdp
            lo1 = (lude >= e["RLMIN"]) & (in_lu[k + 1] >= ZEPS2)
            out_clc = _where(lo1, clc + (1.0 - clc) * (1.0 - np.exp(-lude / in_lu[k + 1])), clc)
            qc2 = _where(lo1, qc1 + lude, qc1)
            # :266-277
            fac1 = 1.0 / (RD * t2)
            rho = ap * fac1
            fac2 = 1.0 / (ap - RETV * foeew)
            rodqsdp = -rho * qs_in * fac2
            ldcp = fwat * lvdcp + (1.0 - fwat) * lsdcp
            fac3 = 1.0 / (1.0 + ldcp * dqsdtemp)
            dtdzmo = RG * (1.0 / RCPD - ldcp * rodqsdp) * fac3
            dqsdz = dqsdtemp * dtdzmo - RG * rodqsdp
            fac4 = 1.0 / rho
            lo3 = dt * dqsdz * (in_mfu[k] + in_mfd[k]) * fac4 < qc2
            dqc = np.minimum(dt * dqsdz * (in_mfu[k] + in_mfd[k]) * fac4, qc2)
            qc3 = qc2 - dqc
            # :280-290
            qlwc1 = qc3 * fwat
            qiwc1 = qc3 * (1.0 - fwat)
            condl1 = (qlwc1 - ql) / dt
            condi1 = (qiwc1 - qi) / dt
            covptot1 = np.maximum(tmp_covptotp, out_clc)
            covptot = covptot1
            covpclr1 = covptot - out_clc
            covpclr = np.maximum(covpclr1, 0.0)
            # :293-302
            melt = sfl != 0.0
            cons = _where(melt, cons2 * dp / lfdcp, 0.0)
            z2s = _where(melt, cons * np.maximum(t2 - meltp2, 0.0), 0.0)
            snmlt = _where(melt, np.minimum(sfl, z2s), 0.0)
            tmp_rfln = _where(melt, rfl + snmlt, rfl)
            tmp_sfln = _where(melt, sfl - snmlt, sfl)
            t = _where(melt, t2 - snmlt / cons, t)
            # :305-337
            cloudy = out_clc > ZEPS2
            cldl = _where(cloudy, qlwc1 / out_clc, 0.0)
            ltmp1 = _where(cloudy, np.exp(-((cldl / lcrit) ** 2.0)), 0.0)
            dl = ckcodtl * (1.0 - ltmp1)
            ltmp2 = _where(cloudy, np.exp(-dl), 0.0)
            qlnew = out_clc * cldl * ltmp2
            prr = _where(cloudy, qlwc1 - qlnew, 0.0)
            qlwc = _where(cloudy, qlwc1 - prr, qlwc1)
            cldi = _where(cloudy, qiwc1 / out_clc, 0.0)
            itmp11 = _where(cloudy, np.exp(-((cldi / icrit) ** 2.0)), 0.0)
            itmp12 = _where(cloudy, np.exp(0.025 * (t - RTT)), 0.0)
            di = ckcodti * itmp12 * (1.0 - itmp11)
            itmp2 = _where(cloudy, np.exp(-di), 0.0)
            qinew = out_clc * cldi * itmp2
            prs = _where(cloudy, qiwc1 - qinew, 0.0)
            qiwc = _where(cloudy, qiwc1 - prs, qiwc1)
            # :340-353
            dr1 = cons2 * dp * (prr + prs)
            frz = t < RTT
            rfreeze1 = _where(frz, cons2 * dp * prr, 0.0)
            fwatr1 = _where(frz, 0.0, 1.0)
            tmp_rfln = tmp_rfln + fwatr1 * dr1
            tmp_sfln = tmp_sfln + (1.0 - fwatr1) * dr1
            rfln2, sfln2 = tmp_rfln, tmp_sfln
            # :356-394
            prtot = tmp_rfln + tmp_sfln
            out_covptot = np.zeros(nx, dtype)
            if LEV:
                ev = (prtot > ZEPS2) & (covpclr > ZEPS2)
                preclr1 = prtot * covpclr / covptot1
                qe = qs_in - (qs_in - qlim) * covpclr / (1.0 - out_clc) ** 2.0
                beta = RG * e["RPECONS"] * (
                    np.sqrt(ap / tmp_aph_s) / 0.00509 * preclr1 / covpclr) ** 0.5777
                b = dt * beta * (qs_in - qe) / (1.0 + dt * beta * corqs)
                dtgdp = dt * RG / (in_aph[k + 1] - in_aph[k])
                dpr1 = covpclr * b / dtgdp
                dpr = np.minimum(dpr1, preclr1)
                preclr = preclr1 - dpr
                covptot = _where(ev & (preclr <= 0.0), out_clc, covptot)
                out_covptot = _where(ev, covptot, 0.0)
                evapr = _where(ev, dpr * rfln2 / prtot, 0.0)
                tmp_rfln = tmp_rfln - evapr
                evaps = _where(ev, dpr * sfln2 / prtot, 0.0)
                tmp_sfln = tmp_sfln - evaps
                for n_, v_ in (("preclr1", preclr1), ("qe", qe), ("beta", beta), ("b", b), ("dtgdp", dtgdp),
                               ("dpr1", dpr1), ("dpr", dpr), ("preclr", preclr)):
                    T[n_][k] = _where(ev, v_, 0.0)
            else:
                evapr = np.zeros(nx, dtype)
                evaps = np.zeros(nx, dtype)
            # :401-419
            dqdt = -(condl1 + condi1) + (in_lude[k] + evapr + evaps) * gdp
            dtdt = (lvdcp * condl1 + lsdcp * condi1
                    - (lvdcp * evapr + lsdcp * evaps
                       + in_lude[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                       - (lsdcp - lvdcp) * rfreeze1) * gdp)
            t3 = t + dt * dtdt
            q = q2 + dt * dqdt
            told, qold, qold1 = t3, q, q
            # :422
            t, q = f_cuadjtqs_nl(ap, t3, q, e)
            # :425-439
            dq = np.maximum(qold1 - q, 0.0)
            dr2 = cons2 * dp * dq
            frz2 = (t < RTT) if FIX else (t3 < RTT)             # Q4: literal = pre-adjustment temperature
            rfreeze2 = _where(frz2, fwat * dr2, 0.0)
            fwatr2 = _where(frz2, 0.0, 1.0)
            rn = fwatr2 * dr2
            sn = (1.0 - fwatr2) * dr2
            condl2 = condl1 + fwatr2 * dq / dt
            condi2 = condi1 + (1.0 - fwatr2) * dq / dt
            tmp_rfln = tmp_rfln + rn
            tmp_sfln = tmp_sfln + sn
            rfreeze3 = rfreeze1 + rfreeze2
            # :442-455
            F["out_clc"][k] = out_clc
            F["out_covptot"][k] = out_covptot
            F["out_tnd_q"][k] = -(condl2 + condi2) + (in_lude[k] + evapr + evaps) * gdp
            F["out_tnd_t"][k] = (lvdcp * condl2 + lsdcp * condi2
                                 - (lvdcp * evapr + lsdcp * evaps
                                    + in_lude[k] * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
                                    - (lsdcp - lvdcp) * rfreeze3) * gdp)
            F["out_tnd_ql"][k] = (qlwc - ql) / dt
            F["out_tnd_qi"][k] = (qiwc - qi) / dt
            tmp_covptotp = covptot                               # :458
            # keep the level's trajectory (the stencil's 3-D temporaries)
            loc = locals()
            for n_ in ("rfl", "sfl", "q2", "lfdcp", "lsdcp", "lvdcp", "fwat", "foeew", "esdp1", "facw",
                       "faci", "fac", "cor", "dqsdtemp", "corqs", "qlim", "crh2", "supsat", "qsat", "qcrit",
                       "qt", "qcd", "qpd", "tmp3", "clc", "gdp", "lude", "out_clc", "fac1", "rho", "fac2",
                       "rodqsdp", "ldcp", "fac3", "dtdzmo", "dqsdz", "fac4", "lo3", "dqc", "qc3", "qlwc1",
                       "qiwc1", "condl1", "condi1", "covptot1", "covptot", "covpclr1", "covpclr", "cons",
                       "z2s", "snmlt", "cldl", "ltmp1", "ltmp2", "prr", "cldi", "itmp11", "itmp12", "itmp2",
                       "prs", "rfreeze1", "fwatr1", "rfln2", "sfln2", "prtot", "evapr", "evaps", "t3", "told",
                       "qold", "qold1", "t", "q", "dq", "dr2", "fwatr2", "condl2", "condi2", "rfreeze3", "dp"):
                T[n_][k] = loc[n_]
            T["scalm"][k] = scalm

        # :459-475  (rfl/sfl at level nz take the value left by level nz-1)
        T["rfl"][nz] = tmp_rfln
        T["sfl"][nz] = tmp_sfln
        for n in ("fplsl", "fplsn", "fhpsl", "fhpsn"):
            F["out_" + n][0] = 0.0
        F["out_fplsl"][1:] = T["rfl"][1:]
        F["out_fplsn"][1:] = T["sfl"][1:]
        F["out_fhpsl"][1:] = -F["out_fplsl"][1:] * RLVTT
        F["out_fhpsn"][1:] = -F["out_fplsn"][1:] * RLSTT

        # ------------------------------------------------------------------ backward (:479-996)
        # local copies of the adjoint forcings (the stencil updates them in place)
        a_clc = F["in_clc_i"].copy()
        a_covptot = F["in_covptot_i"].copy()
        a_fplsn = F["in_fplsn_i"] - F["in_fhpsn_i"] * RLSTT       # :481-484
        a_fplsl = F["in_fplsl_i"] - F["in_fhpsl_i"] * RLVTT
        a_tnd_q, a_tnd_qi = F["in_tnd_q_i"], F["in_tnd_qi_i"]
        a_tnd_ql, a_tnd_t = F["in_tnd_ql_i"], F["in_tnd_t_i"]

        O = {n: np.zeros((nz + 1, nx), dtype) for n in
             ("ap", "aph", "lu", "lude", "mfd", "mfu", "q", "qi", "ql", "qsat", "supsat", "t")}
        Z = lambda: np.zeros((nz + 1, nx), dtype)  # noqa: E731
        rfl_i3, sfl_i3, covptot_i3 = Z(), Z(), Z()
        daph_i3, dp_i3, dlu_i3 = Z(), Z(), Z()
        lvdcp_i3, lsdcp_i3, lfdcp_i3 = Z(), Z(), Z()
        tmp_aph_s_i = np.zeros(nx, dtype)
        tmp_rfln_i = np.zeros(nx, dtype)
        tmp_sfln_i = np.zeros(nx, dtype)
        ckcodtla = ckcodtl / 100.0
        ckcodtia = ckcodti / 100.0

        for k in range(nz - 1, -1, -1):
            g = lambda n: T[n][k]  # noqa: E731
            ap, qs_in, lude_in = in_ap[k], in_qsat[k], in_lude[k]
            lvdcp, lsdcp, lfdcp, fwat, gdp = g("lvdcp"), g("lsdcp"), g("lfdcp"), g("fwat"), g("gdp")
            evapr, evaps = g("evapr"), g("evaps")
            rfreeze3, rfreeze1 = g("rfreeze3"), g("rfreeze1")
            dp, t2, scalm = g("dp"), T["t2"][k], T["scalm"][k]
            out_clc, clc = g("out_clc"), g("clc")
            # :500-501
            tmp_rfln_i = tmp_rfln_i + rfl_i3[k + 1] + a_fplsl[k + 1]
            tmp_sfln_i = tmp_sfln_i + sfl_i3[k + 1] + a_fplsn[k + 1]
            # :504-511
            o_qi = -a_tnd_qi[k] / dt
            qiwc_i = a_tnd_qi[k] / dt
            o_ql = -a_tnd_ql[k] / dt
            qlwc_i = a_tnd_ql[k] / dt
            # :514-533
            tt = a_tnd_t[k]
            hh = (lvdcp * evapr + lsdcp * evaps + lude_in * (fwat * lvdcp + (1.0 - fwat) * lsdcp))
            gdp_i = -tt * (hh - (lsdcp - lvdcp) * rfreeze3)
            condl_i = tt * lvdcp
            condi_i = tt * lsdcp
            evapr_i = -tt * lvdcp * gdp
            evaps_i = -tt * lsdcp * gdp
            lvdcp_i = tt * (g("condl2") - evapr * gdp)
            lsdcp_i = tt * (g("condi2") - evaps * gdp)
            o_lude = -tt * gdp * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
            lvdcp_i = lvdcp_i - tt * lude_in * gdp * fwat
            lsdcp_i = lsdcp_i - tt * lude_in * gdp * (1.0 - fwat)
            fwat_i = -tt * lude_in * gdp * (lvdcp - lsdcp)
            lvdcp_i = lvdcp_i - tt * rfreeze3 * gdp
            lsdcp_i = lsdcp_i + tt * rfreeze3 * gdp
            rfreeze_i = tt * (lsdcp - lvdcp) * gdp
            # :536-542
            tq = a_tnd_q[k]
            gdp_i = gdp_i + tq * (lude_in + evapr + evaps)
            o_lude = o_lude + tq * gdp
            evapr_i = evapr_i + tq * gdp
            evaps_i = evaps_i + tq * gdp
            condl_i = condl_i - tq
            condi_i = condi_i - tq
            # :566-592
            rn_i = tmp_rfln_i
            sn_i = tmp_sfln_i
            fwatr2 = g("fwatr2")
            dq_i = (fwatr2 * condl_i + (1.0 - fwatr2) * condi_i) / dt
            dr2_i = fwatr2 * rn_i + (1.0 - fwatr2) * sn_i
            c577 = (g("t") < RTT) if FIX else (g("t3") < RTT)
            fwat_i = _where(c577, fwat_i + g("dr2") * rfreeze_i, fwat_i)
            dr2_i = _where(c577, dr2_i + fwat * rfreeze_i, dr2_i)
            dq_i = dq_i + cons2 * dp * dr2_i
            dp_i = cons2 * g("dq") * dr2_i
            pos = g("qold1") >= g("q")
            if LREGCL:
                dq_i = _where(pos, dq_i * 0.7, dq_i)
            qold_i = _where(pos, dq_i, 0.0)
            o_q = _where(pos, -dq_i, 0.0)
            # :594-598
            o_ap, _t, o_t, _q, o_q = f_cuadjtqs_ad(ap, np.zeros(nx, dtype), g("told"), np.zeros(nx, dtype),
                                                   g("qold"), o_q, e)
            # :601-633
            o_q = o_q + qold_i
            dqdt_i = dt * o_q
            dtdt_i = dt * o_t
            gdp_i = gdp_i - dtdt_i * (hh - (lsdcp - lvdcp) * rfreeze1)
            condl_i = condl_i + dtdt_i * lvdcp
            condi_i = condi_i + dtdt_i * lsdcp
            evapr_i = evapr_i - dtdt_i * lvdcp * gdp
            evaps_i = evaps_i - dtdt_i * lsdcp * gdp
            lvdcp_i = lvdcp_i + dtdt_i * (g("condl1") - evapr * gdp)
            lsdcp_i = lsdcp_i + dtdt_i * (g("condi1") - evaps * gdp)
            o_lude = o_lude - dtdt_i * gdp * (fwat * lvdcp + (1.0 - fwat) * lsdcp)
            lvdcp_i = lvdcp_i - dtdt_i * lude_in * gdp * fwat
            lsdcp_i = lsdcp_i - dtdt_i * lude_in * gdp * (1.0 - fwat)
            fwat_i = fwat_i - dtdt_i * lude_in * gdp * (lvdcp - lsdcp)
            lvdcp_i = lvdcp_i - dtdt_i * rfreeze1 * gdp
            lsdcp_i = lsdcp_i + dtdt_i * rfreeze1 * gdp
            rfreeze_i = rfreeze_i + dtdt_i * (lsdcp - lvdcp) * gdp
            gdp_i = gdp_i + dqdt_i * (lude_in + evapr + evaps)
            o_lude = o_lude + dqdt_i * gdp
            evapr_i = evapr_i + dqdt_i * gdp
            evaps_i = evaps_i + dqdt_i * gdp
            condl_i = condl_i - dqdt_i
            condi_i = condi_i - dqdt_i
            # :635-719
            prtot, covpclr = g("prtot"), g("covpclr")
            a_clc_k = a_clc[k]
            if LEV:
                ev = (prtot > ZEPS2) & (covpclr > ZEPS2)
                dpr, sfln2, rfln2 = g("dpr"), g("sfln2"), g("rfln2")
                preclr, dpr1, preclr1 = g("preclr"), g("dpr1"), g("preclr1")
                b, dtgdp, beta, corqs, qe = g("b"), g("dtgdp"), g("beta"), g("corqs"), g("qe")
                covptot1, qlim = g("covptot1"), g("qlim")
                e_evaps_i = evaps_i - tmp_sfln_i
                e_sfln_i = tmp_sfln_i + dpr * e_evaps_i / prtot
                dpr_i = sfln2 * e_evaps_i / prtot
                prtot_i = -dpr * sfln2 * e_evaps_i / prtot ** 2.0
                e_evapr_i = evapr_i - tmp_rfln_i
                e_rfln_i = tmp_rfln_i + dpr * e_evapr_i / prtot
                dpr_i = dpr_i + rfln2 * e_evapr_i / prtot
                prtot_i = prtot_i - dpr * rfln2 * e_evapr_i / prtot ** 2.0
                cov_i = covptot_i3[k + 1] + a_covptot[k]
                e_clc = _where(preclr <= 0, a_clc_k + cov_i, a_clc_k)
                cov_i = _where(preclr <= 0, 0.0, cov_i)
                capped = dpr1 > preclr1
                preclr_i = _where(capped, dpr_i, 0.0)
                dpr_i = _where(capped, 0.0, dpr_i)
                b_i = covpclr * dpr_i / dtgdp
                covpclr_i = b * dpr_i / dtgdp
                dtgdp_i = -covpclr * b * dpr_i / dtgdp ** 2.0
                daph_i = dt * RG * dtgdp_i / (in_aph[k + 1] - in_aph[k])
                tmp1 = 1.0 + dt * beta * corqs
                beta_i = (dt * (qs_in - qe) * b_i / tmp1
                          - (dt ** 2.0) * beta * (qs_in - qe) * corqs * b_i / tmp1 ** 2.0)
                o_qsat = dt * beta * b_i / tmp1
                qe_i = -dt * beta * b_i / tmp1
                corqs_i = -(dt ** 2.0) * beta * (qs_in - qe) * beta * b_i / tmp1 ** 2.0
                sq_ = np.sqrt(ap / tmp_aph_s)
                xx = (0.5777 * (RG * e["RPECONS"] / 0.00509)
                      * (0.00509 * covpclr / (preclr1 * sq_)) ** 0.4223)
                preclr_i = preclr_i + xx * sq_ * beta_i / covpclr
                e_ap = o_ap + 0.5 * xx * preclr1 * beta_i / (covpclr * np.sqrt(ap * tmp_aph_s))
                e_aphs = tmp_aph_s_i - 0.5 * xx * preclr1 * sq_ * beta_i / (covpclr * tmp_aph_s)
                covpclr_i = covpclr_i + (
                    -(xx * preclr1 * sq_ * beta_i / covpclr ** 2.0)
                    - (qs_in - qlim) * qe_i / (1.0 - out_clc) ** 2.0) + prtot * preclr_i / covptot1
                o_qsat = o_qsat + qe_i - covpclr * qe_i / (1.0 - out_clc) ** 2.0
                qlim_i = covpclr * qe_i / (1.0 - out_clc) ** 2.0
                e_clc = e_clc - 2.0 * (qs_in - qlim) * covpclr * qe_i / (1.0 - out_clc) ** 3.0
                prtot_i = prtot_i + covpclr * preclr_i / covptot1
                cov_i = cov_i - prtot * covpclr * preclr_i / covptot1 ** 2.0
                # merge with the `else` branch (:711-719)
                evaps_i = _where(ev, e_evaps_i, evaps_i)
                evapr_i = _where(ev, e_evapr_i, evapr_i)
                tmp_sfln_i = _where(ev, e_sfln_i, tmp_sfln_i)
                tmp_rfln_i = _where(ev, e_rfln_i, tmp_rfln_i)
                a_clc_k = _where(ev, e_clc, a_clc_k)
                o_ap = _where(ev, e_ap, o_ap)
                tmp_aph_s_i = _where(ev, e_aphs, tmp_aph_s_i)
                corqs_i = _where(ev, corqs_i, 0.0)
                covpclr_i = _where(ev, covpclr_i, 0.0)
                covptot_i = _where(ev, cov_i, 0.0)
                daph_i = _where(ev, daph_i, 0.0)
                o_qsat = _where(ev, o_qsat, 0.0)
                prtot_i = _where(ev, prtot_i, 0.0)
                qlim_i = _where(ev, qlim_i, 0.0)
            else:
                corqs_i = np.zeros(nx, dtype)
                covpclr_i = np.zeros(nx, dtype)
                covptot_i = np.zeros(nx, dtype)
                daph_i = np.zeros(nx, dtype)
                o_qsat = np.zeros(nx, dtype)
                prtot_i = np.zeros(nx, dtype)
                qlim_i = np.zeros(nx, dtype)
            # :722-736
            tmp_rfln_i = tmp_rfln_i + prtot_i
            tmp_sfln_i = tmp_sfln_i + prtot_i
            fwatr1 = g("fwatr1")
            dr_i = fwatr1 * tmp_rfln_i + (1.0 - fwatr1) * tmp_sfln_i
            prr, prs = g("prr"), g("prs")
            c729 = (fwatr1 == 0.0) if FIX else (g("t") < RTT)     # Q5: literal = post-adjustment temperature
            dp_i = _where(c729, dp_i + rfreeze_i * cons2 * prr, dp_i)
            prr_i = _where(c729, rfreeze_i * cons2 * dp, 0.0)
            prr_i = prr_i + cons2 * dp * dr_i
            prs_i = cons2 * dp * dr_i
            dp_i = dp_i + cons2 * (prr + prs) * dr_i
            # :738-782
            cloudy = out_clc > ZEPS2
            cldi, itmp2, itmp12, itmp11 = g("cldi"), g("itmp2"), g("itmp12"), g("itmp11")
            c_prs_i = prs_i - qiwc_i
            c_qiwc_i = qiwc_i + c_prs_i
            qinew_i = -c_prs_i
            c_clc = a_clc_k + qinew_i * cldi * itmp2
            cldi_i = qinew_i * out_clc * itmp2
            di_i = -qinew_i * out_clc * cldi * itmp2
            itmp4 = ckcodtia if LREGCL else ckcodti
            c_t = o_t + 0.025 * itmp4 * itmp12 * (1.0 - itmp11) * di_i
            cldi_i = cldi_i + 2.0 * itmp4 * itmp12 * itmp11 * cldi * di_i / icrit ** 2.0
            c_qiwc_i = c_qiwc_i + cldi_i / out_clc
            c_clc = c_clc - g("qiwc1") * cldi_i / out_clc ** 2.0
            cldl, ltmp1, ltmp2 = g("cldl"), g("ltmp1"), g("ltmp2")
            c_prr_i = prr_i - qlwc_i
            c_qlwc_i = qlwc_i + c_prr_i
            qlnew_i = -c_prr_i
            c_clc = c_clc + qlnew_i * cldl * ltmp2
            cldl_i = qlnew_i * out_clc * ltmp2
            dl_i = -qlnew_i * out_clc * cldl * ltmp2
            ltmp4 = ckcodtla if LREGCL else ckcodtl
            cldl_i = cldl_i + 2.0 * ltmp4 * ltmp1 * cldl * dl_i / lcrit ** 2.0
            c_qlwc_i = c_qlwc_i + cldl_i / out_clc
            c_clc = c_clc - g("qlwc1") * cldl_i / out_clc ** 2.0
            qiwc_i = _where(cloudy, c_qiwc_i, qiwc_i)
            qlwc_i = _where(cloudy, c_qlwc_i, qlwc_i)
            a_clc_k = _where(cloudy, c_clc, a_clc_k)
            o_t = _where(cloudy, c_t, o_t)
            # :785-806
            sfl = g("sfl")
            melt = sfl != 0.0
            cons, snmlt, z2s = g("cons"), g("snmlt"), g("z2s")
            snmlt_i = -o_t / cons + tmp_rfln_i - tmp_sfln_i
            cons_i = o_t * snmlt / cons ** 2.0
            m_rfl_i = tmp_rfln_i
            m_sfl_i = tmp_sfln_i
            allm = sfl <= z2s
            m_sfl_i = _where(allm, m_sfl_i + snmlt_i, m_sfl_i)
            z2s_i = _where(allm, 0.0, snmlt_i)
            warm = t2 > meltp2
            m_t = _where(warm, o_t + cons * z2s_i, o_t)
            cons_i = _where(warm, cons_i + (t2 - meltp2) * z2s_i, cons_i)
            m_dp_i = dp_i + cons2 * cons_i / lfdcp
            m_lfdcp_i = -cons2 * dp * cons_i / lfdcp ** 2.0
            rfl_i3[k] = _where(melt, m_rfl_i, 0.0)
            sfl_i3[k] = _where(melt, m_sfl_i, 0.0)
            tmp_rfln_i = _where(melt, 0.0, tmp_rfln_i)
            tmp_sfln_i = _where(melt, 0.0, tmp_sfln_i)
            o_t = _where(melt, m_t, o_t)
            dp_i = _where(melt, m_dp_i, dp_i)
            lfdcp_i = _where(melt, m_lfdcp_i, 0.0)
            # :810-817
            covpclr_i = _where(g("covpclr1") < 0.0, 0.0, covpclr_i)
            covptot_i = covptot_i + covpclr_i
            a_clc_k = a_clc_k - covpclr_i
            c815 = out_clc > g("covptot")
            a_clc_k = _where(c815, a_clc_k + covptot_i, a_clc_k)
            covptot_i = _where(c815, 0.0, covptot_i)
            covptot_i3[k] = covptot_i
            # :820-825
            qiwc_i = qiwc_i + condi_i / dt
            o_qi = o_qi - condi_i / dt
            qlwc_i = qlwc_i + condl_i / dt
            o_ql = o_ql - condl_i / dt
            qc_i = fwat * qlwc_i + (1.0 - fwat) * qiwc_i
            fwat_i = fwat_i + g("qc3") * (qlwc_i - qiwc_i)
            # :828-842
            lo3 = g("lo3") != 0
            fac4, dqsdz = g("fac4"), g("dqsdz")
            dqc_i = -qc_i
            l_dqc_i = dqc_i * 0.1 if LREGCL else dqc_i
            dqsdz_i = _where(lo3, dt * l_dqc_i * (in_mfd[k] + in_mfu[k]) * fac4, 0.0)
            o_mfd = _where(lo3, dt * l_dqc_i * dqsdz * fac4, 0.0)
            o_mfu = o_mfd
            rho_i = _where(lo3, -l_dqc_i * g("dqc") * fac4, 0.0)
            qc_i = _where(lo3, qc_i, qc_i + dqc_i)
            # :844-855
            dqsdtemp, dtdzmo, ldcp, fac3 = g("dqsdtemp"), g("dtdzmo"), g("ldcp"), g("fac3")
            rodqsdp, fac2, fac1, rho = g("rodqsdp"), g("fac2"), g("fac1"), g("rho")
            dtdzmo_i = dqsdz_i * dqsdtemp
            dqsdtemp_i = dqsdz_i * dtdzmo - dtdzmo * dtdzmo_i * ldcp * fac3
            rodqsdp_i = -RG * (dqsdz_i + dtdzmo_i * ldcp * fac3)
            ldcp_i = -dtdzmo_i * (RG * rodqsdp + dtdzmo * dqsdtemp) * fac3
            fwat_i = fwat_i + ldcp_i * (lvdcp - lsdcp)
            lvdcp_i = lvdcp_i + fwat * ldcp_i
            lsdcp_i = lsdcp_i + (1.0 - fwat) * ldcp_i
            rho_i = rho_i - rodqsdp_i * qs_in * fac2
            o_qsat = o_qsat - rodqsdp_i * rho * fac2
            o_ap = o_ap + rodqsdp_i * rho * qs_in * fac2 ** 2.0 + rho_i * fac1
            foeew_i = -RETV * rodqsdp_i * rho * qs_in * fac2 ** 2.0
            o_t = o_t - rho_i * ap * fac1 / t2
            # :858-877
            lude = g("lude")
            lu1 = in_lu[k + 1]
            lo1 = (k < NLEV - 1) & (lude >= e["RLMIN"]) & (lu1 >= ZEPS2)
            ex = np.exp(-lude / lu1)
            lude_i = _where(lo1, qc_i + (1.0 - clc) / lu1 * ex * a_clc_k, 0.0)
            dlu_i = _where(lo1, (1.0 - clc) * lude / lu1 ** 2.0 * ex * a_clc_k, 0.0)
            a_clc_k = _where(lo1, a_clc_k * (1.0 - (1.0 - ex)), a_clc_k)
            o_lude = o_lude + dt * gdp * lude_i
            gdp_i = gdp_i + dt * lude_in * lude_i
            daph_i = daph_i + RG * gdp_i / (in_aph[k + 1] - in_aph[k]) ** 2.0
            # :880-918
            qt, qcrit, qsat = g("qt"), g("qcrit"), g("qsat")
            qpd, qcd, tmp3 = g("qpd"), g("qcd"), g("tmp3")
            clear = qt < qcrit
            overcast = (~clear) & (qt >= qsat)
            partial = (~clear) & (~overcast)
            p_qpd_i = scalm * qc_i * clc ** 2.0
            p_qcd_i = (1.0 - scalm) * qc_i * clc ** 2.0
            p_clc = a_clc_k + 2.0 * (scalm * qpd + (1.0 - scalm) * qcd) * clc * qc_i
            if LREGCL:
                rat = qpd / qcd
                yyy = np.minimum(0.3, 3.5 * np.sqrt(rat * (1.0 - scalm * (1.0 - rat)) ** 3.0) / (1.0 - scalm))
                p_clc = p_clc * yyy
            den = qcd - scalm * (qt - qcrit)
            p_qpd_i = p_qpd_i - 0.5 / tmp3 * p_clc / den
            p_qcd_i = p_qcd_i + 0.5 / tmp3 * qpd * p_clc / den ** 2.0
            p_qt_i = (-0.5 / tmp3 * (qpd * scalm * p_clc) / den ** 2.0) - p_qpd_i
            p_qcrit_i = (0.5 / tmp3 * (qpd * scalm * p_clc) / den ** 2.0) - p_qcd_i
            p_qsat_i = p_qcd_i + p_qpd_i
            qt_i = _where(partial, p_qt_i, 0.0)
            qsat_i = _where(clear, 0.0, _where(overcast, (1.0 - scalm) * qc_i, p_qsat_i))
            qcrit_i = _where(clear, 0.0, _where(overcast, -(1.0 - scalm) * qc_i, p_qcrit_i))
            # :920-938
            o_q = o_q + qt_i
            o_ql = o_ql + qt_i
            o_qi = o_qi + qt_i
            qsat_i = qsat_i + qcrit_i * g("crh2")
            o_qsat = o_qsat + qsat_i * g("supsat")
            supsat_i = qsat_i * qs_in
            o_t = _where(t2 < e["RTICE"], o_t - 0.003 * supsat_i, o_t)
            qgt = g("q2") > qs_in
            o_qsat = _where(qgt, o_qsat + qlim_i, o_qsat)
            o_q = _where(qgt, o_q, o_q + qlim_i)
            # :941-967
            fac, cor, facw, faci, foeew = g("fac"), g("cor"), g("facw"), g("faci"), g("foeew")
            dqsdtemp_i = dqsdtemp_i + cons3 * corqs_i
            o_qsat = o_qsat + fac * cor * dqsdtemp_i
            cor_i = fac * qs_in * dqsdtemp_i
            fac_i = cor * qs_in * dqsdtemp_i
            esdp_i = RETV * cor_i * cor ** 2.0
            facw_i = fwat * fac_i
            faci_i = (1.0 - fwat) * fac_i
            fwat_i = fwat_i + (facw - faci) * fac_i
            o_t = o_t - 2.0 * (e["R5IES"] * faci_i / (t2 - e["R4IES"]) ** 3.0
                               + e["R5LES"] * facw_i / (t2 - e["R4LES"]) ** 3.0)
            esdp_i = _where(g("esdp1") > ZQMAX, 0.0, esdp_i)
            foeew_i = foeew_i + esdp_i / ap
            o_ap = o_ap - esdp_i * foeew / ap ** 2.0
            cold = t2 < RTT
            z3es = _where(cold, e["R3IES"], e["R3LES"])
            z4es = _where(cold, e["R4IES"], e["R4LES"])
            o_t = o_t + z3es * (RTT - z4es) * foeew_i * foeew / (t2 - z4es) ** 2.0
            o_t = _where(cold, o_t + 0.545 * 0.17 * fwat_i / np.cosh(0.17 * (t2 - e["RLPTRC"])) ** 2.0, o_t)
            # keep what later computations read
            daph_i3[k], dp_i3[k], dlu_i3[k] = daph_i, dp_i, dlu_i
            lvdcp_i3[k], lsdcp_i3[k], lfdcp_i3[k] = lvdcp_i, lsdcp_i, lfdcp_i
            O["ap"][k], O["t"][k], O["q"][k], O["ql"][k], O["qi"][k] = o_ap, o_t, o_q, o_ql, o_qi
            O["qsat"][k], O["lude"][k], O["mfd"][k], O["mfu"][k] = o_qsat, o_lude, o_mfd, o_mfu

        # :970-986 corrections to staggered fields
        tmp_aph_s_i = tmp_aph_s_i + (-daph_i3[nz - 1] + dp_i3[nz - 1])
        O["aph"][nz] = tmp_aph_s_i
        O["lu"][nz] = -dlu_i3[nz - 1]
        for k in range(nz - 1, 0, -1):
            O["aph"][k] = daph_i3[k] - daph_i3[k - 1] - dp_i3[k] + dp_i3[k - 1]
            O["lu"][k] = -dlu_i3[k - 1]
        O["aph"][0] = daph_i3[0] - dp_i3[0]
        O["lu"][0] = 0.0
        # :988-996
        zzv = RLVTT * lvdcp_i3[:nz] + RLSTT * lsdcp_i3[:nz] + RLMLT * lfdcp_i3[:nz]
        O["q"][:nz] = O["q"][:nz] + (-zzv * RCPD * RVTMP2 / (RCPD + RCPD * RVTMP2 * T["q"][:nz]) ** 2.0)
        O["supsat"][:nz] = dt * O["q"][:nz]

        for n in ("ap", "lude", "mfd", "mfu", "q", "qi", "ql", "qsat", "supsat", "t"):
            F["out_" + n + "_i"][:nz] = O[n][:nz]
        F["out_aph_i"][...] = O["aph"]
        F["out_lu_i"][...] = O["lu"]
        F["out_tnd_cml_t_i"][:nz] = dt * O["t"][:nz]
        F["out_tnd_cml_q_i"][:nz] = dt * O["q"][:nz]
        F["out_tnd_cml_ql_i"][:nz] = dt * O["ql"][:nz]
        F["out_tnd_cml_qi_i"][:nz] = dt * O["qi"][:nz]
