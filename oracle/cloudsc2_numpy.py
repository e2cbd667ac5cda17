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

PARITY PINNING.  `data/input.h5` is absent, so the golden files pin only layout and invariants
(tests/test_golden_invariants.py).  The formulas are pinned by executing the reference's own
stencil source through the build's small gtscript interpreter (tests/golden/…), by
the TL Taylor test against NL and by the AD dot-product test against TL.
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
    z3es = np.where(warm, e["R3LES"], e["R3IES"])
    z4es = np.where(warm, e["R4LES"], e["R4IES"])
    z5alcp = np.where(warm, e["R5ALVCP"], e["R5ALSCP"])
    zaldcp = np.where(warm, e["RALVDCP"], e["RALSDCP"])
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
            trpaus = np.where(t[k] > t[k + 1], eta[k], trpaus)
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
    crh2 = np.where(
        eta_k < trpaus,
        rh3,
        np.where(
            eta_k < bound1,
            rh3 + (rh2 - rh3) * (eta_k - trpaus) / deta2,
            np.where(eta_k < bound2, rh2, rh1 + (rh2 - rh1) * np.sqrt((1.0 - eta_k) / deta1)),
        ),
    )
    return crh2


NL_INPUTS = ("in_ap", "in_aph", "in_lu", "in_lude", "in_mfd", "in_mfu", "in_q", "in_qi", "in_ql",
             "in_qsat", "in_supsat", "in_t", "in_tnd_cml_q", "in_tnd_cml_qi", "in_tnd_cml_ql",
             "in_tnd_cml_t")
NL_OUTPUTS = ("out_clc", "out_covptot", "out_fhpsl", "out_fhpsn", "out_fplsl", "out_fplsn",
              "out_tnd_q", "out_tnd_qi", "out_tnd_ql", "out_tnd_t")


def cloudsc2_nl(fields: Dict[str, np.ndarray], in_eta: np.ndarray, dt, externals) -> None:
    """nonlinear/_stencils/cloudsc2.py:93-399.  `fields` holds the 16 `in_*` and 10 `out_*`
    arrays (nz+1, nx); outputs are written in place.  `out_fplsl[0]`/`out_fplsn[0]` are NOT
    written, exactly as the reference."""
    e = externals
    F = fields
    in_ap, in_aph, in_lu, in_lude = F["in_ap"], F["in_aph"], F["in_lu"], F["in_lude"]
    in_mfd, in_mfu, in_qsat = F["in_mfd"], F["in_mfu"], F["in_qsat"]
    dtype = in_ap.dtype
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
                fwat = np.where(cold, 0.545 * (np.tanh(0.17 * (t - e["RLPTRC"])) + 1.0), 1.0)
                z3es = np.where(cold, e["R3IES"], e["R3LES"])
                z4es = np.where(cold, e["R4IES"], e["R4LES"])
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
            qsat = np.where(t < e["RTICE"], qs_in * (1.8 - 0.003 * t), qs_in)
            qcrit = crh2 * qsat
            # :196-207
            qt = q + ql + qi
            clear = qt < qcrit
            overcast = (~clear) & (qt >= qsat)
            qpd = qsat - qt
            qcd = qsat - qcrit
            clc_p = 1.0 - np.sqrt(qpd / (qcd - scalm * (qt - qcrit)))
            clc = np.where(clear, 0.0, np.where(overcast, 1.0, clc_p))
            qc = np.where(clear, 0.0, np.where(
                overcast, (1.0 - scalm) * (qsat - qcrit),
                (scalm * qpd + (1.0 - scalm) * qcd) * (clc_p ** 2.0)))
            # :210-215
            gdp = RG / (in_aph[k + 1] - in_aph[k])
            lude = dt * in_lude[k] * gdp
            lo1 = (lude >= e["RLMIN"]) & (in_lu[k + 1] >= ZEPS2)
            clc = np.where(lo1, clc + (1.0 - clc) * (1.0 - np.exp(-lude / in_lu[k + 1])), clc)
            qc = np.where(lo1, qc + lude, qc)
            # :218-224
            rho = ap / (e["RD"] * t)
            rodqsdp = -rho * qs_in / (ap - e["RETV"] * foeew)
            ldcp = fwat * lvdcp + (1.0 - fwat) * lsdcp
            dtdzmo = RG * (1.0 / RCPD - ldcp * rodqsdp) / (1.0 + ldcp * dqsdtemp)
            dqsdz = dqsdtemp * dtdzmo - RG * rodqsdp
            dqc = np.minimum(dt * dqsdz * (in_mfu[k] + in_mfd[k]) / rho, qc)
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
            rfln = np.where(melt, tmp_rfl + snmlt, tmp_rfl)
            sfln = np.where(melt, tmp_sfl - snmlt, tmp_sfl)
            t = np.where(melt, t - snmlt / cons, t)
            # :249-272
            cloudy = clc > ZEPS2
            lcrit = 1.9 * e["RCLCRIT"] if LEV else 2.0 * e["RCLCRIT"]
            cldl = qlwc / clc
            dl = ckcodtl * (1.0 - np.exp(-((cldl / lcrit) ** 2.0)))
            prr = np.where(cloudy, qlwc - clc * cldl * np.exp(-dl), 0.0)
            qlwc = np.where(cloudy, qlwc - prr, qlwc)
            icrit = 0.0001 if LEV else 2.0 * e["RCLCRIT"]
            cldi = qiwc / clc
            di = ckcodti * np.exp(0.025 * (t - RTT)) * (1.0 - np.exp(-((cldi / icrit) ** 2.0)))
            prs = np.where(cloudy, qiwc - clc * cldi * np.exp(-di), 0.0)
            qiwc = np.where(cloudy, qiwc - prs, qiwc)
            # :275-285
            dr = cons2 * dp * (prr + prs)
            frz = t < RTT
            rfreeze = np.where(frz, cons2 * dp * prr, 0.0)
            fwatr = np.where(frz, 0.0, 1.0)
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
                tmp_covptot = np.where(ev & (preclr <= 0.0), clc, tmp_covptot)
                out_covptot_k = np.where(ev, tmp_covptot, 0.0)
                evapr = np.where(ev, dpr * rfln / prtot, 0.0)
                rfln = rfln - evapr
                evaps = np.where(ev, dpr * sfln / prtot, 0.0)
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
            t, q = f_cuadjtqs_nl(ap, t, q, e)
            # :350-364
            dq = np.maximum(qold - q, 0.0)
            dr2 = cons2 * dp * dq
            frz2 = t < RTT
            rfreeze2 = np.where(frz2, fwat * dr2, 0.0)
            fwatr = np.where(frz2, 0.0, 1.0)
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
