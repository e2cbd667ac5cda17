// cloudsc2_tl as a hand-written CDNA4 kernel: NL trajectory + tangent-linear perturbation in one
// forward sweep.  Restates
// /root/reference/src/cloudsc2_gt4py/physics/tangent_linear/_stencils/cloudsc2.py:124-774 and
// /root/reference/src/cloudsc2_gt4py/physics/tangent_linear/_stencils/cuadjtqs.py:22-84
// (line numbers below refer to the first file).  Same execution model as cloudsc2_nl.hip: one lane
// per column, carried state (rfl, sfl, covptot and their perturbations) in registers, level k+1
// prefetched while level k is computed, flux shift fused into the sweep.
//
// Template flags: REG = LREGCL (regularisations :294-301, :367-368, :444-448, :475-487, :667-668);
// EVAP = LEVAPLS2 or LDRAIN1D (precipitation-evaporation block :528-616 and the 1.9 RCLCRIT / 1e-4
// autoconversion thresholds; never enabled by the reference's own drivers).
#include "cloudsc2_common.hpp"

#ifndef CS2_TL_DIAG
#define CS2_TL_DIAG 0   // diagnostics only (wrong results): 1 = the kernel's memory traffic without the physics
#endif

#ifndef CS2_TL_DRAIN_LEFT
#define CS2_TL_DRAIN_LEFT 0   // stores that may stay in flight at the drain: 0 / 5 / 10 / 15 / all = 694.8 / 697.7 / 706.1 / 714.1 / 717.5 us
#endif
#ifndef CS2_TL_DRAIN
#define CS2_TL_DRAIN 1   // register-path kernel: drain the level's stores before the next level is requested (see drain_vmem)
#endif

namespace cs2 {

template <typename T>
struct TLIn {
    T ap, aph1, lu1, lude, mfd, mfu, q, qi, ql, qsat, supsat, t, tq, tqi, tql, tt;
};

template <typename T, typename O>
__device__ __forceinline__ TLIn<T> tl_load(const CPtrs<T, NL_NUM_IN>& in, O lsb, O o) {
    TLIn<T> x;
    x.ap = ldg(in.p[NL_IN_AP], o);
    x.aph1 = ldg(in.p[NL_IN_APH], o + lsb);
    x.lu1 = ldg(in.p[NL_IN_LU], o + lsb);
    x.lude = ldg(in.p[NL_IN_LUDE], o);
    x.mfd = ldg(in.p[NL_IN_MFD], o);
    x.mfu = ldg(in.p[NL_IN_MFU], o);
    x.q = ldg(in.p[NL_IN_Q], o);
    x.qi = ldg(in.p[NL_IN_QI], o);
    x.ql = ldg(in.p[NL_IN_QL], o);
    x.qsat = ldg(in.p[NL_IN_QSAT], o);
    x.supsat = ldg(in.p[NL_IN_SUPSAT], o);
    x.t = ldg(in.p[NL_IN_T], o);
    x.tq = ldg(in.p[NL_IN_TND_CML_Q], o);
    x.tqi = ldg(in.p[NL_IN_TND_CML_QI], o);
    x.tql = ldg(in.p[NL_IN_TND_CML_QL], o);
    x.tt = ldg(in.p[NL_IN_TND_CML_T], o);
    return x;
}

// state_increment (common/_stencils/state_increment.py:61-80) applied on the fly: x_i = f * x, supsat_i = 0 with
// IGNORE_SUPSAT - the very products the stand-alone increment kernel stores (cloudsc2_aux.hip), so the fused variant
// (INC, C ABI cloudsc2_tl_incremented_*) feeds the level function exactly the words the separate calls would load.
template <typename T>
__device__ __forceinline__ TLIn<T> tl_increment(const TLIn<T>& x, T f, bool zero_supsat) {
    TLIn<T> y;   // rounded_product: these are the STORED products of the increment kernel, never half of an fma
#define CS2_P(m) y.m = rounded_product<T>(f, x.m)
    CS2_P(ap); CS2_P(aph1); CS2_P(lu1); CS2_P(lude); CS2_P(mfd); CS2_P(mfu); CS2_P(q); CS2_P(qi); CS2_P(ql); CS2_P(qsat);
    CS2_P(t); CS2_P(tq); CS2_P(tqi); CS2_P(tql); CS2_P(tt);
#undef CS2_P
    y.supsat = zero_supsat ? T(0.0) : rounded_product<T>(f, x.supsat);
    return y;
}

template <typename T>
struct TLCarry {
    T rfl, rfl_i, sfl, sfl_i, covptot, covptot_i, aph_k, aph_k_i;
    T aph_s, aph_s_i;  // surface pressure and its perturbation (:126-127), read by the evaporation block only
};

template <typename T>
struct TLOut {
    T clc, clc_i, tnd_q, tnd_q_i, tnd_t, tnd_t_i, tnd_ql, tnd_ql_i, tnd_qi, tnd_qi_i, rfln, rfln_i, sfln, sfln_i;
    T covptot, covptot_i;  // :185-186, non-zero only where the evaporation block ran (:586-587)
};

// One iteration of tangent_linear/_stencils/cuadjtqs.py:22-52 with shared reciprocals
// (rap = 1/ap from the caller, r = 1/(t - z4es) serves the exponent, foeew_i, z2s and z2s_i).
template <typename T>
__device__ __forceinline__ void tl_cuadj_iter(const Ext<T>& e, const ExpK<T>& xk, T rap, T ap_i, T& t, T& t_i, T& q,
                                              T& q_i, T z3es, T z4es, T z5alcp, T zaldcp) {
    const T qp_i = -ap_i * rap * rap;
    const T r = frcp<T>(t - z4es);
    const T foeew = e.R2ES * fexp<T>(xk, z3es * (t - e.RTT) * r);
    const T foeew_i = foeew * z3es * t_i * (e.RTT - z4es) * r * r;
    T qsat = rap * foeew;
    T qsat_i = qp_i * foeew + rap * foeew_i;
    if (qsat > e.ZQMAX) {
        qsat = e.ZQMAX;
        qsat_i = T(0.0);
    }
    const T cor = frcp<T>(T(1.0) - e.RETV * qsat);
    const T cor_i = e.RETV * qsat_i * cor * cor;
    qsat_i = qsat_i * cor + qsat * cor_i;
    qsat *= cor;
    const T z2s = z5alcp * r * r;
    const T z2s_i = T(-2.0) * z5alcp * t_i * r * r * r;
    const T rden = frcp<T>(T(1.0) + qsat * cor * z2s);
    const T cond = (q - qsat) * rden;
    const T cond_i = (q_i - qsat_i) * rden -
                     (q - qsat) * (qsat_i * cor * z2s + qsat * cor_i * z2s + qsat * cor * z2s_i) * rden * rden;
    t += zaldcp * cond;
    t_i += zaldcp * cond_i;
    q -= cond;
    q_i -= cond_i;
}

// One level of the forward sweep (:149-753) for one column.  x = trajectory inputs, y = perturbations.
// Same algebra as the reference; divisions are x * frcp(y) with shared reciprocals, tanh/cosh come
// from ONE exponential: with ex = exp(-0.34 (t - RLPTRC)), rr = 1/(1 + ex):
//   0.545 (tanh u + 1) = 1.09 rr,   1 / cosh(u)^2 = 4 ex rr^2      (u = 0.17 (t - RLPTRC)).
template <typename T, bool REG, bool EVAP>
__device__ __forceinline__ TLOut<T> tl_level(const Ext<T>& e, const NLK<T>& kc, const ExpK<T>& xk, const TLIn<T>& x,
                                             const TLIn<T>& y, int k, T eta_k, T scalm, const CrhCol<T>& crh, T dt,
                                             TLCarry<T>& c) {
    TLOut<T> o;
#if CS2_TL_DIAG == 1
    {
        const T s1 = x.ap + x.aph1 + x.lu1 + x.lude + x.mfd + x.mfu + x.q + x.qi + x.ql + x.qsat + x.supsat + x.t + x.tq + x.tqi +
                     x.tql + x.tt;
        const T s2 = y.ap + y.aph1 + y.lu1 + y.lude + y.mfd + y.mfu + y.q + y.qi + y.ql + y.qsat + y.supsat + y.t + y.tq + y.tqi +
                     y.tql + y.tt;
        o.clc = s1; o.clc_i = s2; o.tnd_q = s1 + s2; o.tnd_q_i = s1 - s2; o.tnd_t = s1 * s2; o.tnd_t_i = s2 - s1;
        o.tnd_ql = s1 + T(1); o.tnd_ql_i = s2 + T(1); o.tnd_qi = s1 + T(2); o.tnd_qi_i = s2 + T(2);
        o.rfln = c.rfl + s1; o.rfln_i = c.rfl_i + s2; o.sfln = c.sfl + s2; o.sfln_i = c.sfl_i + s1;
        o.covptot = s1 - T(1); o.covptot_i = s2 - T(1);
        c.rfl = o.rfln; c.rfl_i = o.rfln_i; c.sfl = o.sfln; c.sfl_i = o.sfln_i; c.aph_k = x.aph1; c.aph_k_i = y.aph1;
        return o;
    }
#endif
    // :139-140, :151-156
    T t = x.t + dt * x.tt;
    T t_i = y.t + dt * y.tt;
    T q = x.q + dt * x.tq + x.supsat;
    T q_i = y.q + dt * y.tq + y.supsat;
    const T ql = x.ql + dt * x.tql;
    const T ql_i = y.ql + dt * y.tql;
    const T qi = x.qi + dt * x.tqi;
    const T qi_i = y.qi + dt * y.tqi;
    // :171-180
    const T dp = x.aph1 - c.aph_k;
    const T dp_i = y.aph1 - c.aph_k_i;
    const T rdp = frcp<T>(dp);
    const T zden = e.RCPD + e.RCPD * e.RVTMP2 * q;
    const T zz = frcp<T>(zden);
    const T zz_i = -e.RCPD * e.RVTMP2 * q_i * zz * zz;
    const T lfdcp = e.RLMLT * zz, lfdcp_i = e.RLMLT * zz_i;
    const T lsdcp = e.RLSTT * zz, lsdcp_i = e.RLSTT * zz_i;
    const T lvdcp = e.RLVTT * zz, lvdcp_i = e.RLVTT * zz_i;
    // :189-205
    const T rl = frcp<T>(t - e.R4LES);
    const T ri = frcp<T>(t - e.R4IES);
    const T rap = frcp<T>(x.ap);
    T fwat, fwat_i, z3es, z4es, r4;
    if (t < e.RTT) {
        const T ex = fexp<T>(xk, -kc.fw2 * (t - e.RLPTRC));
        const T rr = frcp<T>(T(1.0) + ex);
        fwat = T(1.09) * rr;
        fwat_i = T(0.545) * T(0.17) * t_i * (T(4.0) * ex * rr * rr);
        z3es = e.R3IES;
        z4es = e.R4IES;
        r4 = ri;
    } else {
        fwat = T(1.0);
        fwat_i = T(0.0);
        z3es = e.R3LES;
        z4es = e.R4LES;
        r4 = rl;
    }
    const T foeew = e.R2ES * fexp<T>(xk, z3es * (t - e.RTT) * r4);
    const T foeew_i = z3es * (e.RTT - z4es) * t_i * foeew * r4 * r4;
    T esdp = foeew * rap;
    T esdp_i = foeew_i * rap - foeew * y.ap * rap * rap;
    T cor;
    if (esdp > e.ZQMAX) {
        esdp = e.ZQMAX;
        esdp_i = T(0.0);
        cor = kc.cormax;
    } else {
        cor = frcp<T>(T(1.0) - e.RETV * esdp);
    }
    // :207-222
    const T facw = e.R5LES * rl * rl;
    const T facw_i = T(-2.0) * e.R5LES * t_i * rl * rl * rl;
    const T faci = e.R5IES * ri * ri;
    const T faci_i = T(-2.0) * e.R5IES * t_i * ri * ri * ri;
    const T fac = fwat * facw + (T(1.0) - fwat) * faci;
    const T fac_i = fwat_i * (facw - faci) + fwat * facw_i + (T(1.0) - fwat) * faci_i;
    const T cor_i = e.RETV * esdp_i * cor * cor;
    const T dqsdtemp = fac * cor * x.qsat;
    const T dqsdtemp_i = fac_i * cor * x.qsat + fac * cor_i * x.qsat + fac * cor * y.qsat;
    // (:221-230 corqs / qlim feed only the evaporation block and are formed there)
    // :233-253
    const T crh2 = crh2_at(crh, eta_k);
    // :256-265
    T supsat, supsat_i;
    if (t < e.RTICE) {
        supsat = T(1.8) - T(0.003) * t;
        supsat_i = T(-0.003) * t_i;
    } else {
        supsat = T(1.0);
        supsat_i = T(0.0);
    }
    const T qsat = x.qsat * supsat;
    const T qsat_i = y.qsat * supsat + x.qsat * supsat_i;
    const T qcrit = crh2 * qsat;
    const T qcrit_i = crh2 * qsat_i;
    // :268-306
    const T qt = q + ql + qi;
    const T qt_i = q_i + ql_i + qi_i;
    T clc, clc_i, qc, qc_i;
    if (qt < qcrit) {
        clc = T(0.0);
        clc_i = T(0.0);
        qc = T(0.0);
        qc_i = T(0.0);
    } else if (qt >= qsat) {
        clc = T(1.0);
        clc_i = T(0.0);
        qc = (T(1.0) - scalm) * (qsat - qcrit);
        qc_i = (T(1.0) - scalm) * (qsat_i - qcrit_i);
    } else {
        const T qpd = qsat - qt;
        const T qpd_i = qsat_i - qt_i;
        const T qcd = qsat - qcrit;
        const T qcd_i = qsat_i - qcrit_i;
        const T den = qcd - scalm * (qt - qcrit);
        const T rden = frcp<T>(den);
        const T tmp1 = rsqrt_<T>(qpd * rden);
        clc = T(1.0) - tmp1;
        clc_i = T(-0.5) * frcp<T>(tmp1) * (qpd_i * den - qpd * (qcd_i - scalm * (qt_i - qcrit_i))) * rden * rden;
        if constexpr (REG) {
            const T rat = qpd * frcp<T>(qcd);
            const T yyy = rmin<T>(T(0.3), T(3.5) * rsqrt_<T>(rat * cube(T(1.0) - scalm * (T(1.0) - rat))) *
                                              frcp<T>(T(1.0) - scalm));
            clc_i *= yyy;
        }
        qc = (scalm * qpd + (T(1.0) - scalm) * qcd) * sq(clc);
        qc_i = (scalm * qpd_i + (T(1.0) - scalm) * qcd_i) * sq(clc) +
               T(2.0) * (scalm * qpd + (T(1.0) - scalm) * qcd) * clc * clc_i;
    }
    // :309-325 convective detrainment
    const T gdp = e.RG * rdp;
    const T gdp_i = -e.RG * dp_i * rdp * rdp;
    const T lude = dt * x.lude * gdp;
    const T lude_i = dt * (y.lude * gdp + x.lude * gdp_i);
    if (k < e.NLEV - 1 && lude >= e.RLMIN && x.lu1 >= e.ZEPS2) {
        const T rlu = frcp<T>(x.lu1);
        const T tmp2 = fexp<T>(xk, -lude * rlu);
        clc_i += -clc_i * (T(1.0) - tmp2) + (T(1.0) - clc) * tmp2 * (lude_i * rlu - lude * y.lu1 * rlu * rlu);
        clc += (T(1.0) - clc) * (T(1.0) - tmp2);
        qc += lude;
        qc_i += lude_i;
    }
    // :328-354 compensating subsidence
    const T rt = frcp<T>(t);
    const T fac1 = rt * kc.rRD;
    const T rho = x.ap * fac1;
    const T rho_i = (y.ap - x.ap * t_i * rt) * fac1;
    const T fac2 = frcp<T>(x.ap - e.RETV * foeew);
    const T rodqsdp = -rho * x.qsat * fac2;
    const T rodqsdp_i = (-rho_i * x.qsat - rho * y.qsat + rho * x.qsat * (y.ap - e.RETV * foeew_i) * fac2) * fac2;
    const T ldcp = fwat * lvdcp + (T(1.0) - fwat) * lsdcp;
    const T ldcp_i = fwat_i * (lvdcp - lsdcp) + fwat * lvdcp_i + (T(1.0) - fwat) * lsdcp_i;
    const T fac3 = frcp<T>(T(1.0) + ldcp * dqsdtemp);
    const T dtdzmo = e.RG * (kc.rRCPD - ldcp * rodqsdp) * fac3;
    const T dtdzmo_i = -(e.RG * (ldcp_i * rodqsdp + ldcp * rodqsdp_i) +
                         dtdzmo * (ldcp_i * dqsdtemp + ldcp * dqsdtemp_i)) * fac3;
    const T dqsdz = dqsdtemp * dtdzmo - e.RG * rodqsdp;
    const T dqsdz_i = dqsdtemp_i * dtdzmo + dqsdtemp * dtdzmo_i - e.RG * rodqsdp_i;
    // :356-373
    const T rrho = e.RD * t * rap;
    const T tmp3 = dt * dqsdz * (x.mfu + x.mfd) * rrho;
    T dqc, dqc_i;
    if (tmp3 < qc) {
        dqc = tmp3;
        dqc_i = (dt * (dqsdz_i * (x.mfu + x.mfd) + dqsdz * (y.mfu + y.mfd)) - dqc * rho_i) * rrho;
        if constexpr (REG) dqc_i *= T(0.1);
    } else {
        dqc = qc;
        dqc_i = qc_i;
    }
    qc -= dqc;
    qc_i -= dqc_i;
    // :376-386
    T qlwc = qc * fwat;
    T qlwc_i = qc_i * fwat + qc * fwat_i;
    T qiwc = qc * (T(1.0) - fwat);
    T qiwc_i = qc_i * (T(1.0) - fwat) - qc * fwat_i;
    T condl = (qlwc - ql) * kc.rdt;
    T condl_i = (qlwc_i - ql_i) * kc.rdt;
    T condi = (qiwc - qi) * kc.rdt;
    T condi_i = (qiwc_i - qi_i) * kc.rdt;
    // :390-397 maximum overlap (covpclr feeds only the evaporation block)
    if (clc > c.covptot) {
        c.covptot = clc;
        c.covptot_i = clc_i;
    }
    // :400-427 melting of incoming snow;  1/lfdcp = zden / RLMLT
    T rfln, rfln_i, sfln, sfln_i;
    if (c.sfl != T(0.0)) {
        const T ilf = zden * kc.rRLMLT;
        const T cons = kc.cons2 * dp * ilf;
        const T cons_i = kc.cons2 * (dp_i * lfdcp - dp * lfdcp_i) * ilf * ilf;
        T z2s, z2s_i;
        if (t > kc.meltp2) {
            z2s = cons * (t - kc.meltp2);
            z2s_i = cons_i * (t - kc.meltp2) + cons * t_i;
        } else {
            z2s = T(0.0);
            z2s_i = T(0.0);
        }
        T snmlt, snmlt_i;
        if (c.sfl <= z2s) {
            snmlt = c.sfl;
            snmlt_i = c.sfl_i;
        } else {
            snmlt = z2s;
            snmlt_i = z2s_i;
        }
        rfln = c.rfl + snmlt;
        rfln_i = c.rfl_i + snmlt_i;
        sfln = c.sfl - snmlt;
        sfln_i = c.sfl_i - snmlt_i;
        const T rcons = frcp<T>(cons);
        t -= snmlt * rcons;
        t_i -= (snmlt_i * cons - snmlt * cons_i) * rcons * rcons;
    } else {
        rfln = c.rfl;
        rfln_i = c.rfl_i;
        sfln = c.sfl;
        sfln_i = c.sfl_i;
    }
    // :429-503 autoconversion
    T prr = T(0.0), prr_i = T(0.0), prs = T(0.0), prs_i = T(0.0);
    if (clc > e.ZEPS2) {
        const T rclc = frcp<T>(clc);
        const T cldl = qlwc * rclc;
        const T cldl_i = qlwc_i * rclc - qlwc * clc_i * rclc * rclc;
        const T ltmp4 = fexp<T>(xk, -sq(cldl * kc.rlcrit));
        const T dl = kc.ckcodtl * (T(1.0) - ltmp4);
        const T ltmp5 = fexp<T>(xk, -dl);
        const T dl_i = (T(2.0) * (REG ? kc.ckcodtl * T(0.01) : kc.ckcodtl) * kc.rlcrit * kc.rlcrit) * ltmp4 * cldl * cldl_i;
        const T qlnew = clc * cldl * ltmp5;
        const T qlnew_i = clc_i * cldl * ltmp5 + clc * cldl_i * ltmp5 - clc * cldl * ltmp5 * dl_i;
        prr = qlwc - qlnew;
        prr_i = qlwc_i - qlnew_i;
        qlwc -= prr;
        qlwc_i -= prr_i;
        const T cldi = qiwc * rclc;
        const T cldi_i = qiwc_i * rclc - qiwc * clc_i * rclc * rclc;
        const T itmp41 = fexp<T>(xk, -sq(cldi * kc.ricrit));
        const T itmp42 = fexp<T>(xk, T(0.025) * (t - e.RTT));
        const T di = kc.ckcodti * itmp42 * (T(1.0) - itmp41);
        const T itmp5 = fexp<T>(xk, -di);
        const T di_i = (REG ? kc.ckcodti * T(0.01) : kc.ckcodti) * itmp42 *
                       (itmp41 * (T(2.0) * cldi * cldi_i * kc.ricrit * kc.ricrit - T(0.025) * t_i) + T(0.025) * t_i);
        const T qinew = clc * cldi * itmp5;
        const T qinew_i = clc_i * cldi * itmp5 + clc * cldi_i * itmp5 - clc * cldi * itmp5 * di_i;
        prs = qiwc - qinew;
        prs_i = qiwc_i - qinew_i;
        qiwc -= prs;
        qiwc_i -= prs_i;
    }
    // :506-523 new precipitation
    const T dr = kc.cons2 * dp * (prr + prs);
    const T dr_i = kc.cons2 * (dp_i * (prr + prs) + dp * (prr_i + prs_i));
    T rfreeze, rfreeze_i;
    if (t < e.RTT) {
        rfreeze = kc.cons2 * dp * prr;
        rfreeze_i = kc.cons2 * (dp_i * prr + dp * prr_i);
        sfln += dr;
        sfln_i += dr_i;
    } else {
        rfreeze = T(0.0);
        rfreeze_i = T(0.0);
        rfln += dr;
        rfln_i += dr_i;
    }
    // :526-616 precipitation evaporation (LEVAPLS2 or LDRAIN1D).  q / q_i are still the first guess of :154-156
    // here, so corqs (:221-222) and qlim (:225-230) are formed in place; covpclr is :392-397.
    T evapr = T(0.0), evapr_i = T(0.0), evaps = T(0.0), evaps_i = T(0.0);
    o.covptot = T(0.0);
    o.covptot_i = T(0.0);
    if constexpr (EVAP) {
        T covpclr = c.covptot - clc;
        T covpclr_i = c.covptot_i - clc_i;
        if (covpclr < T(0.0)) {
            covpclr = T(0.0);
            covpclr_i = T(0.0);
        }
        const T prtot = rfln + sfln;
        const T prtot_i = rfln_i + sfln_i;
        if (prtot > e.ZEPS2 && covpclr > e.ZEPS2) {
            const T corqs = T(1.0) + kc.cons3 * dqsdtemp;
            const T corqs_i = kc.cons3 * dqsdtemp_i;
            T qlim, qlim_i;
            if (q > x.qsat) {
                qlim = x.qsat;
                qlim_i = y.qsat;
            } else {
                qlim = q;
                qlim_i = q_i;
            }
            // trajectory quantities with discrete consequences (the covptot reset below, a flux that must become
            // exactly 0 when everything evaporates) use IEEE division like the reference, not x * frcp(y)
            const T rcov = frcp<T>(c.covptot);
            const T preclr = prtot * covpclr / c.covptot;
            const T preclr_i = (prtot_i * covpclr + prtot * covpclr_i) * rcov - prtot * covpclr * c.covptot_i * rcov * rcov;
            const T romc = frcp<T>(T(1.0) - clc);
            const T qe = x.qsat - (x.qsat - qlim) * covpclr * romc * romc;
            const T qe_i = y.qsat - (y.qsat * covpclr - qlim_i * covpclr + (x.qsat - qlim) * covpclr_i) * romc * romc -
                           T(2.0) * (x.qsat - qlim) * covpclr * clc_i * romc * romc * romc;
            const T tmp6 = rsqrt_<T>(x.ap * frcp<T>(c.aph_s));
            const T rcp = frcp<T>(covpclr);
            const T arg = tmp6 * preclr * rcp * T(1.0 / 0.00509);
            const T pw = rpow<T>(arg, T(0.5777));
            const T beta = e.RG * e.RPECONS * pw;
            // (0.00509 covpclr / (tmp6 preclr))^0.4223 = arg^-0.4223 = pw / arg
            const T beta_i = T(0.5777 / 0.00509) * e.RG * e.RPECONS * pw * frcp<T>(arg) *
                             ((tmp6 * preclr_i + T(0.5) * preclr * y.ap * frcp<T>(tmp6) -
                               T(0.5) * preclr * tmp6 * c.aph_s_i * frcp<T>(c.aph_s)) * rcp -
                              tmp6 * preclr * covpclr_i * rcp * rcp);
            const T rden = frcp<T>(T(1.0) + dt * beta * corqs);
            const T b = dt * beta * (x.qsat - qe) * rden;
            // the reference's dt**2 factor in the second term is kept (tangent_linear/_stencils/cloudsc2.py:565-569)
            const T b_i = dt * (beta_i * (x.qsat - qe) + beta * (y.qsat - qe_i)) * rden -
                          dt * dt * b * (beta_i * corqs + beta * corqs_i) * rden;
            const T rdtgdp = dp * frcp<T>(kc.rgdt);  // 1 / dtgdp = dp / (dt RG)
            const T dtgdp_i = -dt * e.RG * dp_i * rdp * rdp;
            T dpr = covpclr * b * rdtgdp;
            T dpr_i = (covpclr_i * b + covpclr * b_i) * rdtgdp - covpclr * b * dtgdp_i * rdtgdp * rdtgdp;
            // preclr - dpr <= 0 (:580) <=> dpr >= preclr; written as a comparison because `preclr - dpr` may be
            // contracted into an fma with the product that defines preclr, which would leave a rounding residue
            const bool all_evaporates = dpr >= preclr;
            if (dpr > preclr) {
                dpr = preclr;
                dpr_i = preclr_i;
            }
            if (all_evaporates) {
                c.covptot = clc;
                c.covptot_i = clc_i;
            }
            o.covptot = c.covptot;
            o.covptot_i = c.covptot_i;
            const T rpr = frcp<T>(prtot);
            evapr = dpr * rfln / prtot;
            evapr_i = (dpr_i * rfln + dpr * rfln_i) * rpr - dpr * rfln * prtot_i * rpr * rpr;
            rfln -= evapr;
            rfln_i -= evapr_i;
            evaps = dpr * sfln / prtot;
            evaps_i = (dpr_i * sfln + dpr * sfln_i) * rpr - dpr * sfln * prtot_i * rpr * rpr;
            sfln -= evaps;
            sfln_i -= evaps_i;
        }
    }
    // :619-659   (src = in_lude + evapr + evaps;  hh = the latent-heat weighted sum of the same three sources)
    const T src = x.lude + evapr + evaps;
    const T src_i = y.lude + evapr_i + evaps_i;
    const T hh = x.lude * ldcp + lvdcp * evapr + lsdcp * evaps;
    const T hh_i = y.lude * ldcp + x.lude * ldcp_i + lvdcp_i * evapr + lvdcp * evapr_i + lsdcp_i * evaps + lsdcp * evaps_i;
    const T dqdt = -(condl + condi) + src * gdp;
    const T dqdt_i = -(condl_i + condi_i) + src_i * gdp + src * gdp_i;
    const T tmp7 = hh - (lsdcp - lvdcp) * rfreeze;
    const T dtdt = lvdcp * condl + lsdcp * condi - tmp7 * gdp;
    const T dtdt_i = lvdcp_i * condl + lvdcp * condl_i + lsdcp_i * condi + lsdcp * condi_i -
                     (hh_i - (lsdcp_i - lvdcp_i) * rfreeze - (lsdcp - lvdcp) * rfreeze_i) * gdp - tmp7 * gdp_i;
    t += dt * dtdt;
    t_i += dt * dtdt_i;
    q += dt * dqdt;
    q_i += dt * dqdt_i;
    const T qold = q;
    const T qold_i = q_i;
    // :662 (tangent_linear/_stencils/cuadjtqs.py:55-84)
    {
        T a3, a4, a5, ad;
        if (t > e.RTT) {
            a3 = e.R3LES; a4 = e.R4LES; a5 = e.R5ALVCP; ad = e.RALVDCP;
        } else {
            a3 = e.R3IES; a4 = e.R4IES; a5 = e.R5ALSCP; ad = e.RALSDCP;
        }
        tl_cuadj_iter(e, xk, rap, y.ap, t, t_i, q, q_i, a3, a4, a5, ad);
        tl_cuadj_iter(e, xk, rap, y.ap, t, t_i, q, q_i, a3, a4, a5, ad);
    }
    // :664-673
    T dq, dq_i;
    if (qold >= q) {
        dq = qold - q;
        dq_i = qold_i - q_i;
        if constexpr (REG) dq_i *= T(0.7);
    } else {
        dq = T(0.0);
        dq_i = T(0.0);
    }
    const T dr2 = kc.cons2 * dp * dq;
    const T dr2_i = kc.cons2 * (dp_i * dq + dp * dq_i);
    // :677-703
    if (t < e.RTT) {
        rfreeze += fwat * dr2;
        rfreeze_i += fwat_i * dr2 + fwat * dr2_i;
        condi += dq * kc.rdt;
        condi_i += dq_i * kc.rdt;
        sfln += dr2;
        sfln_i += dr2_i;
    } else {
        condl += dq * kc.rdt;
        condl_i += dq_i * kc.rdt;
        rfln += dr2;
        rfln_i += dr2_i;
    }
    // :706-741
    o.clc = clc;
    o.clc_i = clc_i;
    o.tnd_q = -(condl + condi) + src * gdp;
    o.tnd_q_i = -(condl_i + condi_i) + src_i * gdp + src * gdp_i;
    const T tmp8 = hh - (lsdcp - lvdcp) * rfreeze;
    o.tnd_t = lvdcp * condl + lsdcp * condi - tmp8 * gdp;
    o.tnd_t_i = lvdcp_i * condl + lvdcp * condl_i + lsdcp_i * condi + lsdcp * condi_i -
                (hh_i - (lsdcp_i - lvdcp_i) * rfreeze - (lsdcp - lvdcp) * rfreeze_i) * gdp - tmp8 * gdp_i;
    o.tnd_ql = (qlwc - ql) * kc.rdt;
    o.tnd_ql_i = (qlwc_i - ql_i) * kc.rdt;
    o.tnd_qi = (qiwc - qi) * kc.rdt;
    o.tnd_qi_i = (qiwc_i - qi_i) * kc.rdt;
    // :744-753
    o.rfln = rfln;
    o.rfln_i = rfln_i;
    o.sfln = sfln;
    o.sfln_i = sfln_i;
    c.rfl = rfln;
    c.rfl_i = rfln_i;
    c.sfl = sfln;
    c.sfl_i = sfln_i;
    c.aph_k = x.aph1;
    c.aph_k_i = y.aph1;
    return o;
}

template <typename T, typename O>
__device__ __forceinline__ void tl_store(const MPtrs<T, NL_NUM_OUT>& out, const MPtrs<T, NL_NUM_OUT>& out_i,
                                         const Ext<T>& e, O lsb, O i, const TLOut<T>& o) {
    stg(out.p[NL_OUT_CLC], i, o.clc);
    stg(out_i.p[NL_OUT_CLC], i, o.clc_i);
    stg(out.p[NL_OUT_COVPTOT], i, o.covptot);   // :185-186 (only the evaporation block sets it non-zero)
    stg(out_i.p[NL_OUT_COVPTOT], i, o.covptot_i);
    stg(out.p[NL_OUT_TND_Q], i, o.tnd_q);
    stg(out_i.p[NL_OUT_TND_Q], i, o.tnd_q_i);
    stg(out.p[NL_OUT_TND_T], i, o.tnd_t);
    stg(out_i.p[NL_OUT_TND_T], i, o.tnd_t_i);
    stg(out.p[NL_OUT_TND_QL], i, o.tnd_ql);
    stg(out_i.p[NL_OUT_TND_QL], i, o.tnd_ql_i);
    stg(out.p[NL_OUT_TND_QI], i, o.tnd_qi);
    stg(out_i.p[NL_OUT_TND_QI], i, o.tnd_qi_i);
    // :766-774
    stg(out.p[NL_OUT_FPLSL], i + lsb, o.rfln);
    stg(out_i.p[NL_OUT_FPLSL], i + lsb, o.rfln_i);
    stg(out.p[NL_OUT_FPLSN], i + lsb, o.sfln);
    stg(out_i.p[NL_OUT_FPLSN], i + lsb, o.sfln_i);
    stg(out.p[NL_OUT_FHPSL], i + lsb, -o.rfln * e.RLVTT);
    stg(out_i.p[NL_OUT_FHPSL], i + lsb, -o.rfln_i * e.RLVTT);
    stg(out.p[NL_OUT_FHPSN], i + lsb, -o.sfln * e.RLSTT);
    stg(out_i.p[NL_OUT_FHPSN], i + lsb, -o.sfln_i * e.RLSTT);
}

// INC: the perturbation fields are not read but formed as finc * in (state_increment fused in, see tl_increment):
// 16 input streams instead of 32, no landing buffer for the second 16.
// BIG: 64-bit byte offsets (fields of 4 GiB and more, see offset_t in cloudsc2_common.hpp); instantiated without INC only.
template <typename T, bool REG, bool EVAP, bool INC = false, bool BIG = false>
__global__ void __launch_bounds__(kColBlock)
tl_kernel(Ext<T> e, NLK<T> kc, ExpK<T> xk, int nx, int nz, int64_t ls, CPtrs<T, NL_NUM_IN> in,
          CPtrs<T, NL_NUM_IN> in_i, const T* __restrict__ eta, MPtrs<T, NL_NUM_OUT> out, MPtrs<T, NL_NUM_OUT> out_i,
          T dt, T finc, int zero_supsat_i) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);
    if constexpr (sizeof(T) == 8) {
        // fp64 constants of the level loop -> VGPRs (see pin_vgpr in cloudsc2_common.hpp)
        pin_vgpr(e.RCPD); pin_vgpr(e.RLSTT); pin_vgpr(e.RLVTT); pin_vgpr(e.RLMLT); pin_vgpr(e.R4LES);
        pin_vgpr(e.R4IES); pin_vgpr(e.RTT); pin_vgpr(e.R3IES); pin_vgpr(e.R3LES); pin_vgpr(e.R2ES);
        pin_vgpr(e.ZQMAX); pin_vgpr(e.RETV); pin_vgpr(e.R5LES); pin_vgpr(e.R5IES); pin_vgpr(e.RG);
        pin_vgpr(e.RD); pin_vgpr(kc.rdt); pin_vgpr(kc.cons2); pin_vgpr(kc.rRD); pin_vgpr(kc.rRCPD); pin_vgpr(dt);
        pin_vgpr(xk.l2e); pin_vgpr(xk.ln2h); pin_vgpr(xk.ln2l); pin_vgpr(xk.c12); pin_vgpr(xk.c11);
        pin_vgpr(xk.c10); pin_vgpr(xk.c9); pin_vgpr(xk.c8); pin_vgpr(xk.c7); pin_vgpr(xk.c6);
        pin_vgpr(xk.c5); pin_vgpr(xk.c4); pin_vgpr(xk.c3);
    }

    const int gcol = xcd_block() * kColBlock + threadIdx.x;
    // Lanes past the last column retire here (no workgroup barrier follows build_level_table).  They must not be carried
    // along under an `if (live)` around the stores: that branch is a merge point for hipcc's wait-count insertion, which
    // then drains every store of a level (`s_waitcnt vmcnt(0)`) before the next level's words are handed over
    // (docs/TUNING_LOG.md 3.9).
    if (gcol >= nx) return;
    using O = offset_t<BIG>;
#if CS2_TL_DIAG == 2
    const O lsb = nz < 0 ? O(ls) : O(0);   // diagnostics only (wrong results): every level reads and writes level 0 -
#else                                                  // cache-resident rows, the kernel's time without HBM
    const O lsb = O(ls) * O(sizeof(T));
#endif
    const O colb = O(gcol) * O(sizeof(T));

    const T trpaus = trpaus_prescan<T, false, O>(in.p[NL_IN_T], in.p[NL_IN_TND_CML_T], lsb, colb, dt, s_eta, klo, khi);
    const CrhCol<T> crh = crh_setup<T>(trpaus);

    // :124-135
    TLCarry<T> c;
    c.rfl = c.rfl_i = c.sfl = c.sfl_i = c.covptot = c.covptot_i = T(0.0);
    c.aph_k = ldg(in.p[NL_IN_APH], colb);
    c.aph_s = EVAP ? ldg(in.p[NL_IN_APH], O(nz) * lsb + colb) : T(1.0);
    if constexpr (INC) {
        c.aph_k_i = rounded_product<T>(finc, c.aph_k);
        c.aph_s_i = EVAP ? rounded_product<T>(finc, c.aph_s) : T(0.0);
    } else {
        c.aph_k_i = ldg(in_i.p[NL_IN_APH], colb);
        c.aph_s_i = EVAP ? ldg(in_i.p[NL_IN_APH], O(nz) * lsb + colb) : T(0.0);
    }

    // :757-765
    stg(out.p[NL_OUT_FPLSL], colb, T(0.0));
    stg(out_i.p[NL_OUT_FPLSL], colb, T(0.0));
    stg(out.p[NL_OUT_FPLSN], colb, T(0.0));
    stg(out_i.p[NL_OUT_FPLSN], colb, T(0.0));
    stg(out.p[NL_OUT_FHPSL], colb, T(0.0));
    stg(out_i.p[NL_OUT_FHPSL], colb, T(0.0));
    stg(out.p[NL_OUT_FHPSN], colb, T(0.0));
    stg(out_i.p[NL_OUT_FHPSN], colb, T(0.0));

    O o = colb;
    if constexpr (INC) {
        TLIn<T> xa = tl_load<T, O>(in, lsb, o);
        for (int k = 0; k < nz; ++k) {
            TLIn<T> xn = xa;
            if (k + 1 < nz) xn = tl_load<T, O>(in, lsb, o + lsb);
            const TLIn<T> ya = tl_increment<T>(xa, finc, zero_supsat_i != 0);
            const TLOut<T> r = tl_level<T, REG, EVAP>(e, kc, xk, xa, ya, k, s_eta[k], s_scalm[k], crh, dt, c);
            tl_store<T, O>(out, out_i, e, lsb, o, r);
            if constexpr (CS2_TL_DRAIN != 0) drain_vmem<CS2_TL_DRAIN_LEFT>();
            xa = xn;
            o += lsb;
        }
    } else {
        TLIn<T> xa = tl_load<T, O>(in, lsb, o), ya = tl_load<T, O>(in_i, lsb, o);
        for (int k = 0; k < nz; ++k) {
            TLIn<T> xn = xa, yn = ya;
            if (k + 1 < nz) {
                xn = tl_load<T, O>(in, lsb, o + lsb);
                yn = tl_load<T, O>(in_i, lsb, o + lsb);
            }
            const TLOut<T> r = tl_level<T, REG, EVAP>(e, kc, xk, xa, ya, k, s_eta[k], s_scalm[k], crh, dt, c);
            tl_store<T, O>(out, out_i, e, lsb, o, r);
            if constexpr (CS2_TL_DRAIN != 0) drain_vmem<CS2_TL_DRAIN_LEFT>();
            xa = xn;
            ya = yn;
            o += lsb;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-ring variant of the sweep.  Same prologue, same tl_level / tl_store; only the way the 32 input words of a level
// (16 trajectory + 16 perturbation fields) reach the lane differs - the load path of cloudsc2_nl.hip's nl_ring_kernel,
// restated for 32 fields:
//   * every wave owns RD slots of 32 fields x 64 columns in LDS (16 KB per slot in fp64); level k + RD - 1 is requested
//     with LDS-DMA (`global_load_lds_dwordx4`, no VGPR destination) while level k is computed.  In fp64 RD = 2 is what
//     fits a CU (4 waves x 2 x 16 KB + table): the same one level ahead as the register prefetch of tl_kernel, but
//     without its 32-double landing buffer (64 VGPRs) - the register path runs at 254-256 VGPRs + 36-96 AGPRs;
//   * one DMA instruction moves 16 B per lane = NPL columns (2 fp64 / 4 fp32): lane group g of instruction i fetches
//     field i*NPL + g of the concatenated list [in[0..15], in_i[0..15]], so field f of a slot starts at
//     f * 64 * sizeof(T) and the lane reads its own column with ds_read_b64 / _b32;
//   * the waits are counted by hand (vmcnt retires in order on gfx9): when level k is read, the operations younger than
//     its DMAs are the (RD-1) x NI DMAs of the levels in flight and the (RD-1) x 20 stores of the levels computed since.
//     Unlike nl_ring_kernel (three slots, one level of stores of slack) the count is EXACT here: with two slots a
//     stricter wait would expose the completion latency of the previous level's stores on every level.  tl_store
//     issues exactly kTLStores stores per level (20 distinct fields; tests/test_ring_isa.py counts them in the compiled ISA, and __graft_entry__.build() runs that check whenever it recompiles).
// Used when the launcher can guarantee 16-byte aligned rows and whole waves (launch_tl); every other call takes the
// register-prefetch kernel above.  Results are bit-identical (same arithmetic on the same words).
#ifndef CS2_TL_RING
#define CS2_TL_RING 2   // slots per wave in fp64 (0 disables the variant)
#endif
#ifndef CS2_TL_RING_F32
#define CS2_TL_RING_F32 2
#endif
#ifndef CS2_TL_RING_AUX
#define CS2_TL_RING_AUX (CS2_NT & 1 ? 2 : 0)   // cache policy of the input DMAs: 2 = nt (every byte is read once)
#endif
typedef __attribute__((address_space(3))) void* tl_lds_ptr;
typedef const __attribute__((address_space(1))) void* tl_glb_ptr;
constexpr int kTLFields = 2 * NL_NUM_IN;    // input words per level and column
constexpr int kTLStores = 2 * NL_NUM_OUT;   // stores per level (tl_store)

template <typename T>
struct TLRingGeom {
    static constexpr int NPL = 16 / int(sizeof(T));             // columns per lane per DMA = fields per DMA
    static constexpr int NI = kTLFields / NPL;                   // DMA instructions per level
    static constexpr int SLOT = kTLFields * 64 * int(sizeof(T));  // the 32 input fields of one level
};

// Wait until at most N vector-memory operations are outstanding, then read this lane's column of 16 fields of the slot
// at LDS byte address `a` (+ table entries eta[k] at `ta`, scalm[k] at `tb` when TAB).  Two statements per level (x, y):
// the wait and the LDS reads live in ONE asm statement with a memory clobber - hipcc would otherwise drain vmcnt(0)
// before every LDS read that follows an LDS-DMA, and no store may move across the wait.
template <int N>
__device__ __forceinline__ void tl_ring_read16(uint32_t a, TLIn<double>& x) {
    asm volatile(
        "s_waitcnt vmcnt(%17)\n\t"
            "ds_read_b64 %0, %16\n\t"
            "ds_read_b64 %1, %16 offset:512\n\t"
            "ds_read_b64 %2, %16 offset:1024\n\t"
            "ds_read_b64 %3, %16 offset:1536\n\t"
            "ds_read_b64 %4, %16 offset:2048\n\t"
            "ds_read_b64 %5, %16 offset:2560\n\t"
            "ds_read_b64 %6, %16 offset:3072\n\t"
            "ds_read_b64 %7, %16 offset:3584\n\t"
            "ds_read_b64 %8, %16 offset:4096\n\t"
            "ds_read_b64 %9, %16 offset:4608\n\t"
            "ds_read_b64 %10, %16 offset:5120\n\t"
            "ds_read_b64 %11, %16 offset:5632\n\t"
            "ds_read_b64 %12, %16 offset:6144\n\t"
            "ds_read_b64 %13, %16 offset:6656\n\t"
            "ds_read_b64 %14, %16 offset:7168\n\t"
            "ds_read_b64 %15, %16 offset:7680\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x.ap), "=&v"(x.aph1), "=&v"(x.lu1), "=&v"(x.lude), "=&v"(x.mfd), "=&v"(x.mfu), "=&v"(x.q), "=&v"(x.qi),
          "=&v"(x.ql), "=&v"(x.qsat), "=&v"(x.supsat), "=&v"(x.t), "=&v"(x.tq), "=&v"(x.tqi), "=&v"(x.tql), "=&v"(x.tt)
        : "v"(a), "n"(N)
        : "memory");
}
template <int N>
__device__ __forceinline__ void tl_ring_read16(uint32_t a, TLIn<float>& x) {
    asm volatile(
        "s_waitcnt vmcnt(%17)\n\t"
            "ds_read_b32 %0, %16\n\t"
            "ds_read_b32 %1, %16 offset:256\n\t"
            "ds_read_b32 %2, %16 offset:512\n\t"
            "ds_read_b32 %3, %16 offset:768\n\t"
            "ds_read_b32 %4, %16 offset:1024\n\t"
            "ds_read_b32 %5, %16 offset:1280\n\t"
            "ds_read_b32 %6, %16 offset:1536\n\t"
            "ds_read_b32 %7, %16 offset:1792\n\t"
            "ds_read_b32 %8, %16 offset:2048\n\t"
            "ds_read_b32 %9, %16 offset:2304\n\t"
            "ds_read_b32 %10, %16 offset:2560\n\t"
            "ds_read_b32 %11, %16 offset:2816\n\t"
            "ds_read_b32 %12, %16 offset:3072\n\t"
            "ds_read_b32 %13, %16 offset:3328\n\t"
            "ds_read_b32 %14, %16 offset:3584\n\t"
            "ds_read_b32 %15, %16 offset:3840\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x.ap), "=&v"(x.aph1), "=&v"(x.lu1), "=&v"(x.lude), "=&v"(x.mfd), "=&v"(x.mfu), "=&v"(x.q), "=&v"(x.qi),
          "=&v"(x.ql), "=&v"(x.qsat), "=&v"(x.supsat), "=&v"(x.t), "=&v"(x.tq), "=&v"(x.tqi), "=&v"(x.tql), "=&v"(x.tt)
        : "v"(a), "n"(N)
        : "memory");
}
// eta[k] (LDS byte address ta) and scalm[k] (tb) from the level table; the table is written before the only workgroup
// barrier and never again, so no vector-memory wait is involved
__device__ __forceinline__ void tl_table_read(uint32_t ta, uint32_t tb, double& eta_k, double& scalm_k) {
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(eta_k), "=&v"(scalm_k) : "v"(ta), "v"(tb) : "memory");
}
__device__ __forceinline__ void tl_table_read(uint32_t ta, uint32_t tb, float& eta_k, float& scalm_k) {
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(eta_k), "=&v"(scalm_k) : "v"(ta), "v"(tb) : "memory");
}

template <typename T, bool REG, bool EVAP, int RD>
__global__ void __launch_bounds__(kColBlock, 1)
tl_ring_kernel(Ext<T> e, NLK<T> kc, ExpK<T> xk, int nx, int nz, int64_t ls, CPtrs<T, NL_NUM_IN> in,
               CPtrs<T, NL_NUM_IN> in_i, const T* __restrict__ eta, MPtrs<T, NL_NUM_OUT> out, MPtrs<T, NL_NUM_OUT> out_i,
               T dt) {
    using G = TLRingGeom<T>;
    static_assert(kColBlock % 64 == 0 && RD >= 2, "whole waves, at least one level in flight");
    static_assert((RD - 1) * (G::NI + kTLStores) < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);
    if constexpr (sizeof(T) == 8) {
        pin_vgpr(e.RCPD); pin_vgpr(e.RLSTT); pin_vgpr(e.RLVTT); pin_vgpr(e.RLMLT); pin_vgpr(e.R4LES);
        pin_vgpr(e.R4IES); pin_vgpr(e.RTT); pin_vgpr(e.R3IES); pin_vgpr(e.R3LES); pin_vgpr(e.R2ES);
        pin_vgpr(e.ZQMAX); pin_vgpr(e.RETV); pin_vgpr(e.R5LES); pin_vgpr(e.R5IES); pin_vgpr(e.RG);
        pin_vgpr(e.RD); pin_vgpr(kc.rdt); pin_vgpr(kc.cons2); pin_vgpr(kc.rRD); pin_vgpr(kc.rRCPD); pin_vgpr(dt);
        pin_vgpr(xk.l2e); pin_vgpr(xk.ln2h); pin_vgpr(xk.ln2l); pin_vgpr(xk.c12); pin_vgpr(xk.c11);
        pin_vgpr(xk.c10); pin_vgpr(xk.c9); pin_vgpr(xk.c8); pin_vgpr(xk.c7); pin_vgpr(xk.c6);
        pin_vgpr(xk.c5); pin_vgpr(xk.c4); pin_vgpr(xk.c3);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wcol0 = xcd_block() * kColBlock + wave * 64;   // first column of this wave
    if (wcol0 >= nx) return;                                // nx % 64 == 0 (launcher): whole waves retire; the only
                                                            // workgroup barrier is inside build_level_table
    const uint32_t lsb = uint32_t(ls) * uint32_t(sizeof(T));
    const uint32_t colb = uint32_t(wcol0 + lane) * uint32_t(sizeof(T));

    const T trpaus = trpaus_prescan<T>(in.p[NL_IN_T], in.p[NL_IN_TND_CML_T], lsb, colb, dt, s_eta, klo, khi);
    const CrhCol<T> crh = crh_setup<T>(trpaus);

    // :124-135
    TLCarry<T> c;
    c.rfl = c.rfl_i = c.sfl = c.sfl_i = c.covptot = c.covptot_i = T(0.0);
    c.aph_k = ldg(in.p[NL_IN_APH], colb);
    c.aph_k_i = ldg(in_i.p[NL_IN_APH], colb);
    c.aph_s = EVAP ? ldg(in.p[NL_IN_APH], uint32_t(nz) * lsb + colb) : T(1.0);
    c.aph_s_i = EVAP ? ldg(in_i.p[NL_IN_APH], uint32_t(nz) * lsb + colb) : T(0.0);
    // :757-765
    stg(out.p[NL_OUT_FPLSL], colb, T(0.0));
    stg(out_i.p[NL_OUT_FPLSL], colb, T(0.0));
    stg(out.p[NL_OUT_FPLSN], colb, T(0.0));
    stg(out_i.p[NL_OUT_FPLSN], colb, T(0.0));
    stg(out.p[NL_OUT_FHPSL], colb, T(0.0));
    stg(out_i.p[NL_OUT_FHPSL], colb, T(0.0));
    stg(out.p[NL_OUT_FHPSN], colb, T(0.0));
    stg(out_i.p[NL_OUT_FHPSN], colb, T(0.0));
    // consume the prologue's ordinary loads BEFORE the first DMA is issued: hipcc drains vmcnt(0) at the first use of
    // an ordinary load's result while an LDS-DMA is in flight, which would empty the ring inside level 0
    pin_vgpr(c.aph_k);
    pin_vgpr(c.aph_k_i);
    if constexpr (EVAP) { pin_vgpr(c.aph_s); pin_vgpr(c.aph_s_i); }
    { T tp = crh.rh2; pin_vgpr(tp); }

    // per-lane DMA sources: lane group g of instruction i walks field i*NPL + g of [in | in_i], NPL adjacent columns
    // per lane; aph and lu are read one half level below (aph[k+1], lu[k+1]: :171, :311)
    constexpr int LPG = 64 / G::NPL;   // lanes per group
    const int g = lane / LPG, l = lane % LPG;
    const char* src[G::NI];
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        const int f0 = i * G::NPL;                       // first field of this instruction (compile-time)
        const T* base = f0 < NL_NUM_IN ? in.p[f0] : in_i.p[f0 - NL_NUM_IN];
        int f = f0;
#pragma unroll
        for (int j = 1; j < G::NPL; ++j)
            if (g == j) {
                base = (f0 + j) < NL_NUM_IN ? in.p[f0 + j] : in_i.p[f0 + j - NL_NUM_IN];
                f = f0 + j;
            }
        const int fn = f % NL_NUM_IN;
        const uint32_t lev1 = (fn == NL_IN_APH || fn == NL_IN_LU) ? lsb : 0u;
        src[i] = reinterpret_cast<const char*>(base) + (uint32_t(wcol0 + G::NPL * l) * uint32_t(sizeof(T)) + lev1);
    }
    // LDS: [eta | scalm table][pad][wave 0: RD slots][wave 1: RD slots] ...
    const uint32_t tab_bytes = (2u * uint32_t(nz + 1) * uint32_t(sizeof(T)) + 1023u) & ~1023u;
    const uint32_t ring0 = tab_bytes + uint32_t(wave) * uint32_t(RD * G::SLOT);
    auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            __builtin_amdgcn_global_load_lds((tl_glb_ptr)src[i],
                                             (tl_lds_ptr)(&smem_raw[ring0 + uint32_t(slot * G::SLOT + i * 1024)]), 16, 0,
                                             CS2_TL_RING_AUX);
            src[i] += lsb;
        }
    };
#pragma unroll
    for (int j = 0; j < RD - 1; ++j)
        if (j < nz) issue(j);

    // operations younger than level k's DMAs when level k is read: see the header of this kernel
    constexpr int NFULL = (RD - 1) * (G::NI + kTLStores);
    constexpr int NHEAD = (RD - 1) * G::NI;   // the first RD-1 levels: no stores counted (stricter = safe)
    const uint32_t rd_lane = ring0 + uint32_t(lane) * uint32_t(sizeof(T));
    const uint32_t tb_off = uint32_t(nz + 1) * uint32_t(sizeof(T));
    constexpr uint32_t YOFF = uint32_t(NL_NUM_IN) * 64u * uint32_t(sizeof(T));   // the perturbation half of a slot
    uint32_t o = colb;
    int slot = 0, pslot = RD - 1;
    for (int k = 0; k < nz; ++k) {
        const bool more = k + RD - 1 < nz;
        if (more) issue(pslot);
        TLIn<T> x, y;
        T eta_k, scalm_k;
        const uint32_t a = rd_lane + uint32_t(slot * G::SLOT);
        const uint32_t ta = uint32_t(k) * uint32_t(sizeof(T));
        if (!more) tl_ring_read16<0>(a, x);   // tail: drain
        else if (k < RD - 1) tl_ring_read16<NHEAD>(a, x);
        else tl_ring_read16<NFULL>(a, x);
        tl_ring_read16<63>(a + YOFF, y);      // same level, already covered by the wait above (63 = no further wait)
        tl_table_read(ta, ta + tb_off, eta_k, scalm_k);
        const TLOut<T> r = tl_level<T, REG, EVAP>(e, kc, xk, x, y, k, eta_k, scalm_k, crh, dt, c);
        tl_store<T, uint32_t>(out, out_i, e, lsb, o, r);
        o += lsb;
        slot = slot + 1 == RD ? 0 : slot + 1;
        pslot = pslot + 1 == RD ? 0 : pslot + 1;
    }
}

template <typename T>
int launch_tl(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* const* in, const T* const* in_i,
              const T* eta, T* const* out, T* const* out_i, double dt, hipStream_t stream, double inc_f) {
    // in_i == nullptr: the fused state_increment variant - perturbations formed in the kernel as T(inc_f) * in
    const bool inc = in_i == nullptr;
    const T tinc = static_cast<T>(inc_f);
    const int zsi = p.IGNORE_SUPSAT ? 1 : 0;
    const bool evap = p.LEVAPLS2 || p.LDRAIN1D;
    const Ext<T> e = make_ext<T>(p);
    CPtrs<T, NL_NUM_IN> ci, cii;
    MPtrs<T, NL_NUM_OUT> co, coi;
    for (int i = 0; i < NL_NUM_IN; ++i) { ci.p[i] = in[i]; cii.p[i] = inc ? nullptr : in_i[i]; }
    for (int i = 0; i < NL_NUM_OUT; ++i) { co.p[i] = out[i]; coi.p[i] = out_i[i]; }
    const dim3 grid((nx + kColBlock - 1) / kColBlock), block(kColBlock);
    const size_t smem = 2 * size_t(nz + 1) * sizeof(T);
    const T tdt = static_cast<T>(dt);
    const NLK<T> kc = make_nlk<T>(p, dt, evap);
    const ExpK<T> xk = make_expk<T>();
    const bool big = !fits_u32_offsets<T>(nz, ls);
    if (big) {       // fields of 4 GiB and more: the register-path kernel with 64-bit offsets (not the fused-increment variant)
        if (inc) return -2;
#define CS2_TL_BIG(REG, EVAP)                                                                                      \
    hipLaunchKernelGGL((tl_kernel<T, REG, EVAP, false, true>), grid, block, smem, stream, e, kc, xk, nx, nz, ls,   \
                       ci, cii, eta, co, coi, tdt, tinc, zsi)
        if (p.LREGCL) {
            if (evap) CS2_TL_BIG(true, true); else CS2_TL_BIG(true, false);
        } else {
            if (evap) CS2_TL_BIG(false, true); else CS2_TL_BIG(false, false);
        }
#undef CS2_TL_BIG
        note_kernel("cs2::tl_kernel<big>");
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    constexpr int kRing = sizeof(T) == 8 ? CS2_TL_RING : CS2_TL_RING_F32;
    if constexpr (kRing >= 2) {
        // LDS-ring variant: whole waves, 16-byte aligned rows of every input field (the DMA moves 16 B per lane)
        using G = TLRingGeom<T>;
        bool ring = !inc && nx % 64 == 0 && nz >= kRing && (ls * int64_t(sizeof(T))) % 16 == 0;
        for (int i = 0; i < NL_NUM_IN && ring; ++i)
            ring = reinterpret_cast<uintptr_t>(in[i]) % 16 == 0 && reinterpret_cast<uintptr_t>(in_i[i]) % 16 == 0;
        const size_t tab = (2 * size_t(nz + 1) * sizeof(T) + 1023) & ~size_t(1023);
        const size_t rsmem = tab + size_t(kColBlock / 64) * kRing * G::SLOT;
        ring = ring && rsmem <= size_t(160) * 1024;   // LDS of a CU; very tall columns take the register path
        int dev = 0;
        if (ring)
            if (const int rc = current_device(dev)) return rc;
#ifndef CS2_TL_RING_ALWAYS
#define CS2_TL_RING_ALWAYS 0   // A/B switch: 1 = take the ring whenever it is legal, whatever the occupancy
#endif
        if (ring && sizeof(T) == 8 && !CS2_TL_RING_ALWAYS) {
            // fp64 (two slots per wave = the register path's one level ahead): the ring wins wherever part of the chip is
            // latency-bound - 8 192 .. 49 152 columns -3.5 .. -6 %, 98 304 (1.5 workgroups per CU) -3.7 % - and loses
            // 1.5-2 % when every CU holds the same number of workgroups for the whole launch and HBM is saturated
            // (65 536: 720 vs 710 us, 131 072: 1 518 vs 1 488 us; profiles/r02/ab_tl_ring.txt).  Rule: register path when
            // at least 3/4 of the launch's workgroup rounds are full.  fp32 (8 KB slots): the ring wins at every size
            // measured (65 536: 337 vs 389 us; 524 288: 2 986 vs 3 049 us).
            const int64_t c = device_cus(dev), gx = grid.x;
            const int64_t full = gx / c, rounds = (gx + c - 1) / c;
            ring = 4 * full < 3 * rounds;
        }
        if (ring) {
#define CS2_TL_RING_LAUNCH(REG, EVAP)                                                                                \
    do {                                                                                                             \
        auto kern = tl_ring_kernel<T, REG, EVAP, kRing>;                                                             \
        /* > 64 KB of dynamic LDS needs the opt-in: once per instantiation, device and size */                       \
        static std::atomic<size_t> attr_set[kMaxDevices] = {};                                                       \
        if (!lds_opt_in(kern, attr_set, dev, rsmem)) return -1;                                                      \
        hipLaunchKernelGGL(kern, grid, block, rsmem, stream, e, kc, xk, nx, nz, ls, ci, cii, eta, co, coi, tdt);     \
    } while (0)
            if (p.LREGCL) {
                if (evap) CS2_TL_RING_LAUNCH(true, true); else CS2_TL_RING_LAUNCH(true, false);
            } else {
                if (evap) CS2_TL_RING_LAUNCH(false, true); else CS2_TL_RING_LAUNCH(false, false);
            }
#undef CS2_TL_RING_LAUNCH
            note_kernel("cs2::tl_ring_kernel");
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
    }
#define CS2_TL_LAUNCH(REG, EVAP, INCV)                                                                                 \
    hipLaunchKernelGGL((tl_kernel<T, REG, EVAP, INCV>), grid, block, smem, stream, e, kc, xk, nx, nz, ls, ci, cii, eta, \
                       co, coi, tdt, tinc, zsi)
#define CS2_TL_LAUNCH_I(REG, EVAP)                                               \
    do {                                                                         \
        if (inc) CS2_TL_LAUNCH(REG, EVAP, true); else CS2_TL_LAUNCH(REG, EVAP, false); \
    } while (0)
    if (p.LREGCL) {
        if (evap) CS2_TL_LAUNCH_I(true, true); else CS2_TL_LAUNCH_I(true, false);
    } else {
        if (evap) CS2_TL_LAUNCH_I(false, true); else CS2_TL_LAUNCH_I(false, false);
    }
#undef CS2_TL_LAUNCH_I
#undef CS2_TL_LAUNCH
    note_kernel(inc ? "cs2::tl_kernel<inc>" : "cs2::tl_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template int launch_tl<double>(const Cloudsc2Params&, int, int, int64_t, const double* const*, const double* const*,
                               const double*, double* const*, double* const*, double, hipStream_t, double);
template int launch_tl<float>(const Cloudsc2Params&, int, int, int64_t, const float* const*, const float* const*,
                              const float*, float* const*, float* const*, double, hipStream_t, double);

}  // namespace cs2
