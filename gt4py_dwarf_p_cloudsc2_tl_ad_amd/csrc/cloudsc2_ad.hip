// cloudsc2_ad as a hand-written CDNA4 kernel.  Restates
// /root/reference/src/cloudsc2_gt4py/physics/adjoint/_stencils/cloudsc2.py:124-996 and
// /root/reference/src/cloudsc2_gt4py/physics/adjoint/_stencils/cuadjtqs.py:22-158
// (line numbers below refer to the first file unless prefixed "cuadjtqs").
//
// GT4Py keeps ~80 trajectory temporaries of the forward computation as full 3-D fields for the
// backward computation.  Here NOTHING but the stencil's own outputs goes to HBM:
//   sweep 1 (k = 0 .. nz-1)  recomputes the NL trajectory and writes the 10 NL outputs (:146-475);
//   sweep 2 (k = nz-1 .. 0)  re-reads level k's 16 state inputs + the snow flux that entered the
//                            level (= out_fplsn[k], written by sweep 1 of the same lane), recomputes
//                            that level's local trajectory in registers and runs the adjoint
//                            statements (:479-996) on it.
// The only loop-carried trajectory values are the fluxes entering a level, and those are outputs.
// Backward carries: tmp_rfln_i, tmp_sfln_i, rfl_i/sfl_i of the level below, and daph_i/dp_i/dlu_i of
// the level below for the staggered corrections (:970-986), which are fused into the sweep.
//
// Deliberate, documented choices:
//   Q1  adjoint forcings (in_*_i) are read-only here; the reference zeroes them in place.
//   Q8  rfl_i / sfl_i of a level that does not melt read as 0 (unassigned GT4Py temporaries).
//   out_lude_i is written (=), not accumulated into prior storage contents (:526 uses -= on an
//       output the stencil never initialises; zero-initialised storage gives the same result).
//   Q4/Q5 are reproduced literally when FIX = false (the reference's behaviour): the second
//       freezing test uses the pre-adjustment t3 (:427, :577) and the adjoint of rfreeze1 tests the
//       post-adjustment t (:729).  FIX = true (Cloudsc2Params.AD_TRAJ_FIX, a build extension) uses
//       the tests of the NL/TL stencils instead, which makes AD the exact transpose of TL.
// Template flags: REG = LREGCL; EVAP = LEVAPLS2 or LDRAIN1D (evaporation block :357-394 / :635-709).
// With EVAP the precipitation cover entering a level (tmp_covptotp, :458) is a second loop-carried
// trajectory value that is NOT an output (out_covptot is 0 on levels without evaporation): sweep 1
// parks it in level k of out_mfd_i, sweep 2 reads it back before it overwrites that element.
#include "cloudsc2_common.hpp"

#ifndef CS2_AD_DIAG
#define CS2_AD_DIAG 0   // diagnostics only (wrong results): 1 = the kernel's memory traffic without the physics
#endif
// Tuning switches (A/B-tested with profiles/ab_kernels.py; the defaults are the fastest measured set).
// CS2_AD_PARK: sweep 2 parks the part of a level's recomputed trajectory that the adjoint statements only read in
// their second half (34 values per column, listed in CS2_AD_PARK_LIST) in LDS between ad_forward and that half, instead
// of carrying them through the register-pressure peak (cuadj_bwd + the autoconversion adjoint), where hipcc otherwise
// shuttles them through AGPRs (v_accvgpr_write / _read occupy a VALU issue slot each; ds_write / ds_read do not).
// Measured (profiles/r02/ab_ad_variants.txt, one process, interleaved): fp64 65 536 columns 922.9 us parked vs 911.7 us
// not parked - the fp64 kernel runs at 94 % of the time its memory traffic alone takes (CS2_AD_DIAG = 1), so issue slots
// are not what it waits for; fp32 524 288 columns 3 689 us parked vs 3 817 us (199 -> 166 VGPRs: 3 waves per SIMD
// instead of 2).  Hence the default: bit 0 = park in the fp64 kernels (off), bit 1 = in the fp32 kernels (on).
#ifndef CS2_AD_PARK
#define CS2_AD_PARK 2
#endif
template <typename T>
constexpr bool kADPark = ((CS2_AD_PARK) & (sizeof(T) == 8 ? 1 : 2)) != 0;
#define CS2_AD_PARK_LIST(X)                                                                                            \
    X(fac4) X(dqsdz) X(dqc) X(dqsdtemp) X(dtdzmo) X(fac3) X(rodqsdp) X(fac2) X(rho) X(fac1) X(rt) X(clc) X(rlu) X(exlu) \
    X(lude) X(qt) X(qcrit) X(qsat) X(qpd) X(qcd) X(tmp3) X(rden) X(crh2) X(supsat) X(fac) X(cor) X(facw) X(faci) X(ri)  \
    X(rl) X(esdp1) X(foeew) X(sech2) X(qc3)
#define CS2_AD_PARK_COUNT 34
// CS2_AD_KEEP_MB: cache-resident turnaround.  Sweep 1 ends at level nz-1 and sweep 2 starts there, re-reading the 16
// inputs + 2 fluxes of every level; the last levels sweep 1 touched can still be in the 256 MB memory-side cache when
// sweep 2 asks for them - if neither side marked them non-temporal.  The launcher turns this budget into a number of
// bottom levels (18 words x level stride each) whose sweep-1 loads, flux stores and sweep-2 loads use the default
// policy; every other access of the kernel stays non-temporal.  0 = off.
#ifndef CS2_AD_KEEP_MB
#define CS2_AD_KEEP_MB 0
#endif
#ifndef CS2_AD_DRAIN
#define CS2_AD_DRAIN 0  // bit 0 / bit 1: drain the level's stores before the next level is requested in sweep 1 / sweep 2
#endif
#ifndef CS2_AD_LANDED
#define CS2_AD_LANDED 0 // 1: aph_k marked as landed before sweep 1's loop (see landed()): sweep 1 then keeps its stores in
#endif                  // flight across levels - measured slower (+1.0 ... +1.5 %, docs/TUNING_LOG.md 3.9)
#ifndef CS2_AD_PIN
#define CS2_AD_PIN 3    // fp64 constants pinned in VGPRs: bit 0 = the physical constants, bit 1 = the exp coefficients
#endif

namespace cs2 {

template <typename T>
struct ADIn {
    T ap, aph1, lu1, lude, mfd, mfu, q, qi, ql, qsat, supsat, t, tq, tqi, tql, tt;
};

// `keep`: wave-uniform; the level's 16 words are loaded with the default cache policy instead of non-temporally
// (CS2_AD_KEEP_MB below: the levels where sweep 1 ends are the levels where sweep 2 starts).
template <typename T, typename FP, typename O>
__device__ __forceinline__ ADIn<T> ad_load(const FP& F, O lsb, O o, bool keep = false) {
    ADIn<T> x;
    if (keep) {
        x.ap = ldg_keep(F.in(NL_IN_AP), o);
        x.aph1 = ldg_keep(F.in(NL_IN_APH), o + lsb);
        x.lu1 = ldg_keep(F.in(NL_IN_LU), o + lsb);
        x.lude = ldg_keep(F.in(NL_IN_LUDE), o);
        x.mfd = ldg_keep(F.in(NL_IN_MFD), o);
        x.mfu = ldg_keep(F.in(NL_IN_MFU), o);
        x.q = ldg_keep(F.in(NL_IN_Q), o);
        x.qi = ldg_keep(F.in(NL_IN_QI), o);
        x.ql = ldg_keep(F.in(NL_IN_QL), o);
        x.qsat = ldg_keep(F.in(NL_IN_QSAT), o);
        x.supsat = ldg_keep(F.in(NL_IN_SUPSAT), o);
        x.t = ldg_keep(F.in(NL_IN_T), o);
        x.tq = ldg_keep(F.in(NL_IN_TND_CML_Q), o);
        x.tqi = ldg_keep(F.in(NL_IN_TND_CML_QI), o);
        x.tql = ldg_keep(F.in(NL_IN_TND_CML_QL), o);
        x.tt = ldg_keep(F.in(NL_IN_TND_CML_T), o);
        return x;
    }
    x.ap = ldg(F.in(NL_IN_AP), o);
    x.aph1 = ldg(F.in(NL_IN_APH), o + lsb);
    x.lu1 = ldg(F.in(NL_IN_LU), o + lsb);
    x.lude = ldg(F.in(NL_IN_LUDE), o);
    x.mfd = ldg(F.in(NL_IN_MFD), o);
    x.mfu = ldg(F.in(NL_IN_MFU), o);
    x.q = ldg(F.in(NL_IN_Q), o);
    x.qi = ldg(F.in(NL_IN_QI), o);
    x.ql = ldg(F.in(NL_IN_QL), o);
    x.qsat = ldg(F.in(NL_IN_QSAT), o);
    x.supsat = ldg(F.in(NL_IN_SUPSAT), o);
    x.t = ldg(F.in(NL_IN_T), o);
    x.tq = ldg(F.in(NL_IN_TND_CML_Q), o);
    x.tqi = ldg(F.in(NL_IN_TND_CML_QI), o);
    x.tql = ldg(F.in(NL_IN_TND_CML_QL), o);
    x.tt = ldg(F.in(NL_IN_TND_CML_T), o);
    return x;
}
// store / load with the cache policy chosen by a wave-uniform flag (sweep 1's fluxes that sweep 2 reads back)
template <typename T, typename O>
__device__ __forceinline__ void stg_sel(T* base, O boff, T v, bool keep) {
    if (keep) *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + boff) = v;
    else stg(base, boff, v);
}
template <typename T, typename O>
__device__ __forceinline__ T ldg_sel(const T* base, O boff, bool keep) {
    return keep ? ldg_keep(base, boff) : ldg(base, boff);
}

// Saved state of the two saturation-adjustment iterations (cuadjtqs:53-91), including the
// reciprocals the reverse sweep divides by: r = 1/(targ - z4es), rden = 1/(1 + qsat cor z2s).
template <typename T>
struct CuadjSav {
    T z4es, z5alcp, zaldcp, dfo;  // dfo = z3es * (RTT - z4es)
    T foeew_b, qsat_d, r_b, qsat_b, cor_b, z2s_b, q_b, rden_b;
    T foeew_a, qsat_c, r_a, qsat_a, cor_a, z2s_a, q_a, rden_a;
    bool ltest2, ltest1;
};

template <typename T>
__device__ __forceinline__ void cuadj_fwd_save(const Ext<T>& e, const ExpK<T>& xk, T rap, T& t, T& q,
                                               CuadjSav<T>& s) {
    T z3es;
    if (t > e.RTT) {
        z3es = e.R3LES; s.z4es = e.R4LES; s.z5alcp = e.R5ALVCP; s.zaldcp = e.RALVDCP;
    } else {
        z3es = e.R3IES; s.z4es = e.R4IES; s.z5alcp = e.R5ALSCP; s.zaldcp = e.RALSDCP;
    }
    s.dfo = z3es * (e.RTT - s.z4es);
    {
        const T r = frcp<T>(t - s.z4es);
        const T foeew = e.R2ES * fexp<T>(xk, z3es * (t - e.RTT) * r);
        s.foeew_b = foeew;
        T qsat = foeew * rap;
        s.ltest2 = qsat > e.ZQMAX;
        if (s.ltest2) qsat = e.ZQMAX;
        const T cor = frcp<T>(T(1.0) - e.RETV * qsat);
        s.qsat_d = qsat;
        qsat *= cor;
        s.r_b = r;
        const T z2s = s.z5alcp * r * r;
        const T rden = frcp<T>(T(1.0) + qsat * cor * z2s);
        s.qsat_b = qsat; s.cor_b = cor; s.z2s_b = z2s; s.q_b = q; s.rden_b = rden;
        const T cond1 = (q - qsat) * rden;
        t += s.zaldcp * cond1;
        q -= cond1;
    }
    {
        const T r = frcp<T>(t - s.z4es);
        const T foeew = e.R2ES * fexp<T>(xk, z3es * (t - e.RTT) * r);
        s.foeew_a = foeew;
        T qsat = foeew * rap;
        s.ltest1 = qsat > e.ZQMAX;
        if (s.ltest1) qsat = e.ZQMAX;
        const T cor = frcp<T>(T(1.0) - e.RETV * qsat);
        s.qsat_c = qsat;
        qsat *= cor;
        s.r_a = r;
        const T z2s = s.z5alcp * r * r;
        const T rden = frcp<T>(T(1.0) + qsat * cor * z2s);
        s.qsat_a = qsat; s.cor_a = cor; s.z2s_a = z2s; s.q_a = q; s.rden_a = rden;
        const T cond1 = (q - qsat) * rden;
        t += s.zaldcp * cond1;
        q -= cond1;
    }
}

// Reverse of one iteration (cuadjtqs:93-124 / :126-156); returns this iteration's qp_i contribution.
// `cor` = 1 / (1 - RETV qsat_pre) is the saved value, so RETV / (1 - RETV qsat_pre)^2 = RETV cor^2.
template <typename T>
__device__ __forceinline__ T cuadj_bwd_iter(const Ext<T>& e, const CuadjSav<T>& s, T rap, T& t_i, T& q_i, T qsat,
                                            T cor, T z2s, T q_sav, T r, T rden, T qsat_pre, bool ltest, T foeew) {
    const T cond1_i = -q_i + s.zaldcp * t_i;
    q_i += cond1_i * rden;
    const T w = cond1_i * (q_sav - qsat) * rden * rden;
    T qsat_i = -cond1_i * rden - w * cor * z2s;
    T cor_i = -w * qsat * z2s;
    const T z2s_i = -w * qsat * cor;
    T targ_i = T(-2.0) * z2s_i * s.z5alcp * r * r * r;
    cor_i += qsat_i * qsat_pre;
    qsat_i *= cor;
    qsat_i += cor_i * e.RETV * cor * cor;
    if (ltest) qsat_i = T(0.0);
    const T foeew_i = qsat_i * rap;
    const T qp_i = qsat_i * foeew;
    // R2ES * exp(z3es (targ - RTT) / (targ - z4es)) is the saved foeew (cuadjtqs:117-122)
    targ_i += foeew_i * foeew * s.dfo * r * r;
    t_i += targ_i;
    return qp_i;
}

template <typename T>
__device__ __forceinline__ void cuadj_bwd(const Ext<T>& e, const CuadjSav<T>& s, T rap, T& ap_i, T& t_i, T& q_i) {
    T qp_i = cuadj_bwd_iter(e, s, rap, t_i, q_i, s.qsat_a, s.cor_a, s.z2s_a, s.q_a, s.r_a, s.rden_a, s.qsat_c,
                            s.ltest1, s.foeew_a);
    qp_i += cuadj_bwd_iter(e, s, rap, t_i, q_i, s.qsat_b, s.cor_b, s.z2s_b, s.q_b, s.r_b, s.rden_b, s.qsat_d,
                           s.ltest2, s.foeew_b);
    ap_i -= qp_i * rap * rap;
}

// Local trajectory of one level (:149-458): everything the backward statements read, including the
// reciprocals they divide by (computed once in the forward part).
template <typename T>
struct ADTraj {
    T t2, q2, ql, qi, dp, rdp, rzz, lfdcp, lsdcp, lvdcp, fwat, sech2, foeew, esdp1, rl, ri, rap, facw, faci, fac, cor;
    T dqsdtemp, crh2, supsat, qsat, qcrit, qt, qcd, qpd, tmp3, rden, clc, gdp, lude, rlu, exlu, out_clc;
    T fac1, rt, rho, fac2, rodqsdp, ldcp, fac3, dtdzmo, dqsdz, fac4, dqc, qc3;
    T qlwc1, qiwc1, condl1, condi1, cons, rcons, z2s, snmlt;
    T rclc, cldl, ltmp1, ltmp2, prr, cldi, itmp11, itmp12, itmp2, prs;
    T rfreeze1, fwatr1, t3, qold, dq, dr2, fwatr2, condl2, condi2, rfreeze3;
    T t_post, q_post, rfln, sfln, tnd_q, tnd_t, tnd_ql, tnd_qi;
    // evaporation block (EVAP only)
    T covptot1, covpclr1, covpclr, covptot, out_covptot, prtot, rfln2, sfln2, corqs, qlim, preclr1, romc, qe, sq, xx,
        beta, rtmp1, b, dpr, evapr, evaps;
    bool ev, capped, all_evaporates;
    bool lo1, lo3, melt, cloudy, t3_cold, tpost_cold, t2_cold;
    int cls;  // 0 clear (qt <= qcrit), 1 overcast, 2 partial
    CuadjSav<T> adj;
};

template <typename T, bool FIX, bool EVAP>
__device__ __forceinline__ void ad_forward(const Ext<T>& e, const NLK<T>& kc, const ExpK<T>& xk, const ADIn<T>& x,
                                           T aph_k, int k, T eta_k, T scalm, const CrhCol<T>& crh, T dt, T rfl, T sfl,
                                           T covptot_in, T aph_s, ADTraj<T>& r) {
#if CS2_AD_DIAG == 1
    {
        const T s1 = x.ap + x.aph1 + x.lu1 + x.lude + x.mfd + x.mfu + x.q + x.qi + x.ql + x.qsat + x.supsat + x.t + x.tq + x.tqi +
                     x.tql + x.tt + aph_k;
        r.out_clc = s1; r.out_covptot = s1 + T(1); r.tnd_q = s1 + T(2); r.tnd_t = s1 + T(3); r.tnd_ql = s1 + T(4); r.tnd_qi = s1 + T(5);
        r.rfln = rfl + s1; r.sfln = sfl + s1; r.covptot = covptot_in; r.t2 = s1;
        return;
    }
#endif
    // :135-137, :153-157
    T t = x.t + dt * x.tt;
    r.t2 = t;
    T q = x.q + dt * x.tq + x.supsat;
    r.ql = x.ql + dt * x.tql;
    r.qi = x.qi + dt * x.tqi;
    r.q2 = q;
    // :170-174
    r.dp = x.aph1 - aph_k;
    r.rdp = frcp<T>(r.dp);
    const T zz = e.RCPD + e.RCPD * e.RVTMP2 * q;
    r.rzz = frcp<T>(zz);
    r.lfdcp = e.RLMLT * r.rzz;
    r.lsdcp = e.RLSTT * r.rzz;
    r.lvdcp = e.RLVTT * r.rzz;
    // :181-197;  with ex = exp(-0.34 (t2 - RLPTRC)), rr = 1/(1+ex): 0.545 (tanh u + 1) = 1.09 rr and
    // 1/cosh(u)^2 = 4 ex rr^2 (needed by the adjoint, :967)
    r.rl = frcp<T>(r.t2 - e.R4LES);
    r.ri = frcp<T>(r.t2 - e.R4IES);
    r.rap = frcp<T>(x.ap);
    r.t2_cold = t < e.RTT;
    T z3es, r4;
    if (r.t2_cold) {
        const T ex = fexp<T>(xk, -kc.fw2 * (r.t2 - e.RLPTRC));
        const T rr = frcp<T>(T(1.0) + ex);
        r.fwat = T(1.09) * rr;
        r.sech2 = T(4.0) * ex * rr * rr;
        z3es = e.R3IES;
        r4 = r.ri;
    } else {
        r.fwat = T(1.0);
        r.sech2 = T(0.0);
        z3es = e.R3LES;
        r4 = r.rl;
    }
    r.foeew = e.R2ES * fexp<T>(xk, z3es * (r.t2 - e.RTT) * r4);
    r.esdp1 = r.foeew * r.rap;
    r.facw = e.R5LES * r.rl * r.rl;
    r.faci = e.R5IES * r.ri * r.ri;
    r.fac = r.fwat * r.facw + (T(1.0) - r.fwat) * r.faci;
    r.cor = (r.esdp1 > e.ZQMAX) ? kc.cormax : frcp<T>(T(1.0) - e.RETV * r.esdp1);
    r.dqsdtemp = r.fac * r.cor * x.qsat;
    // :203-231
    r.crh2 = crh2_at(crh, eta_k);
    r.supsat = (r.t2 < e.RTICE) ? T(1.8) - T(0.003) * r.t2 : T(1.0);
    r.qsat = x.qsat * r.supsat;
    r.qcrit = r.crh2 * r.qsat;
    // :234-252
    r.qt = q + r.ql + r.qi;
    T qc1;
    r.rden = T(0.0);
    if (r.qt <= r.qcrit) {
        r.cls = 0;
        r.clc = T(0.0);
        qc1 = T(0.0);
        r.qcd = T(0.0);
        r.qpd = T(0.0);
        r.tmp3 = T(0.0);
    } else if (r.qt >= r.qsat) {
        r.cls = 1;
        r.clc = T(1.0);
        qc1 = (T(1.0) - scalm) * (r.qsat - r.qcrit);
        r.qcd = T(0.0);
        r.qpd = T(0.0);
        r.tmp3 = T(0.0);
    } else {
        r.cls = 2;
        r.qcd = r.qsat - r.qcrit;
        r.qpd = r.qsat - r.qt;
        r.rden = frcp<T>(r.qcd - scalm * (r.qt - r.qcrit));
        r.tmp3 = rsqrt_<T>(r.qpd * r.rden);
        r.clc = T(1.0) - r.tmp3;
        qc1 = (scalm * r.qpd + (T(1.0) - scalm) * r.qcd) * sq(r.clc);
    }
    // :255-263
    r.gdp = e.RG * r.rdp;
    r.lude = dt * x.lude * r.gdp;
    r.lo1 = r.lude >= e.RLMIN && x.lu1 >= e.ZEPS2;
    T qc2;
    if (r.lo1) {
        r.rlu = frcp<T>(x.lu1);
        r.exlu = fexp<T>(xk, -r.lude * r.rlu);
        r.out_clc = r.clc + (T(1.0) - r.clc) * (T(1.0) - r.exlu);
        qc2 = qc1 + r.lude;
    } else {
        r.rlu = T(0.0);
        r.exlu = T(0.0);
        r.out_clc = r.clc;
        qc2 = qc1;
    }
    // :266-277
    r.rt = frcp<T>(r.t2);
    r.fac1 = r.rt * kc.rRD;
    r.rho = x.ap * r.fac1;
    r.fac2 = frcp<T>(x.ap - e.RETV * r.foeew);
    r.rodqsdp = -r.rho * x.qsat * r.fac2;
    r.ldcp = r.fwat * r.lvdcp + (T(1.0) - r.fwat) * r.lsdcp;
    r.fac3 = frcp<T>(T(1.0) + r.ldcp * r.dqsdtemp);
    r.dtdzmo = e.RG * (kc.rRCPD - r.ldcp * r.rodqsdp) * r.fac3;
    r.dqsdz = r.dqsdtemp * r.dtdzmo - e.RG * r.rodqsdp;
    r.fac4 = e.RD * r.t2 * r.rap;
    const T sub = dt * r.dqsdz * (x.mfu + x.mfd) * r.fac4;
    r.lo3 = sub < qc2;
    r.dqc = rmin<T>(sub, qc2);
    r.qc3 = qc2 - r.dqc;
    // :280-283
    r.qlwc1 = r.qc3 * r.fwat;
    r.qiwc1 = r.qc3 * (T(1.0) - r.fwat);
    r.condl1 = (r.qlwc1 - r.ql) * kc.rdt;
    r.condi1 = (r.qiwc1 - r.qi) * kc.rdt;
    // :284-290 maximum overlap
    r.covptot1 = rmax<T>(covptot_in, r.out_clc);
    r.covptot = r.covptot1;
    r.covpclr1 = r.covptot1 - r.out_clc;
    r.covpclr = rmax<T>(r.covpclr1, T(0.0));
    // :293-302 melting of incoming snow;  cons = cons2 dp / lfdcp = cons2 dp zz / RLMLT
    r.melt = sfl != T(0.0);
    T rfln, sfln;
    if (r.melt) {
        r.cons = kc.cons2 * r.dp * zz * kc.rRLMLT;
        r.rcons = frcp<T>(r.cons);
        r.z2s = r.cons * rmax<T>(r.t2 - kc.meltp2, T(0.0));
        r.snmlt = rmin<T>(sfl, r.z2s);
        rfln = rfl + r.snmlt;
        sfln = sfl - r.snmlt;
        t = r.t2 - r.snmlt * r.rcons;
    } else {
        r.cons = T(1.0);
        r.rcons = T(1.0);
        r.z2s = T(0.0);
        r.snmlt = T(0.0);
        rfln = rfl;
        sfln = sfl;
    }
    // :305-337 autoconversion
    r.cloudy = r.out_clc > e.ZEPS2;
    T qlwc = r.qlwc1, qiwc = r.qiwc1;
    if (r.cloudy) {
        r.rclc = frcp<T>(r.out_clc);
        r.cldl = r.qlwc1 * r.rclc;
        r.ltmp1 = fexp<T>(xk, -sq(r.cldl * kc.rlcrit));
        const T dl = kc.ckcodtl * (T(1.0) - r.ltmp1);
        r.ltmp2 = fexp<T>(xk, -dl);
        const T qlnew = r.out_clc * r.cldl * r.ltmp2;
        r.prr = r.qlwc1 - qlnew;
        qlwc = r.qlwc1 - r.prr;
        r.cldi = r.qiwc1 * r.rclc;
        r.itmp11 = fexp<T>(xk, -sq(r.cldi * kc.ricrit));
        r.itmp12 = fexp<T>(xk, T(0.025) * (t - e.RTT));
        const T di = kc.ckcodti * r.itmp12 * (T(1.0) - r.itmp11);
        r.itmp2 = fexp<T>(xk, -di);
        const T qinew = r.out_clc * r.cldi * r.itmp2;
        r.prs = r.qiwc1 - qinew;
        qiwc = r.qiwc1 - r.prs;
    } else {
        r.rclc = r.cldl = r.ltmp1 = r.ltmp2 = r.cldi = r.itmp11 = r.itmp12 = r.itmp2 = T(0.0);
        r.prr = T(0.0);
        r.prs = T(0.0);
    }
    // :340-353
    const T dr1 = kc.cons2 * r.dp * (r.prr + r.prs);
    if (t < e.RTT) {
        r.rfreeze1 = kc.cons2 * r.dp * r.prr;
        r.fwatr1 = T(0.0);
        sfln += dr1;
    } else {
        r.rfreeze1 = T(0.0);
        r.fwatr1 = T(1.0);
        rfln += dr1;
    }
    // :356-394 precipitation evaporation.  IEEE divisions where the result has discrete consequences (see cloudsc2_tl.hip)
    r.evapr = T(0.0);
    r.evaps = T(0.0);
    r.out_covptot = T(0.0);
    r.ev = false;
    if constexpr (EVAP) {
        r.prtot = rfln + sfln;
        r.rfln2 = rfln;
        r.sfln2 = sfln;
        r.ev = r.prtot > e.ZEPS2 && r.covpclr > e.ZEPS2;
        if (r.ev) {
            r.corqs = T(1.0) + kc.cons3 * r.dqsdtemp;  // :199
            r.qlim = rmin<T>(r.q2, x.qsat);             // :200
            r.preclr1 = r.prtot * r.covpclr / r.covptot1;
            r.romc = frcp<T>(T(1.0) - r.out_clc);
            r.qe = x.qsat - (x.qsat - r.qlim) * r.covpclr * r.romc * r.romc;
            r.sq = rsqrt_<T>(x.ap * frcp<T>(aph_s));
            const T arg = r.sq * T(1.0 / 0.00509) * r.preclr1 * frcp<T>(r.covpclr);
            const T pw = rpow<T>(arg, T(0.5777));
            r.beta = e.RG * e.RPECONS * pw;
            // 0.5777 (RG RPECONS / 0.00509) (0.00509 covpclr / (preclr1 sq))^0.4223, and x^-0.4223 = x^0.5777 / x
            r.xx = T(0.5777 / 0.00509) * e.RG * e.RPECONS * pw * frcp<T>(arg);
            r.rtmp1 = frcp<T>(T(1.0) + dt * r.beta * r.corqs);
            r.b = dt * r.beta * (x.qsat - r.qe) * r.rtmp1;
            const T dpr1 = r.covpclr * r.b * r.dp * frcp<T>(kc.rgdt);  // / dtgdp, dtgdp = dt RG / dp
            r.capped = dpr1 > r.preclr1;
            r.dpr = r.capped ? r.preclr1 : dpr1;
            r.all_evaporates = dpr1 >= r.preclr1;  // preclr = preclr1 - dpr <= 0
            if (r.all_evaporates) r.covptot = r.out_clc;
            r.out_covptot = r.covptot;
            r.evapr = r.dpr * r.rfln2 / r.prtot;
            rfln -= r.evapr;
            r.evaps = r.dpr * r.sfln2 / r.prtot;
            sfln -= r.evaps;
        }
    }
    // :401-419
    const T hh = x.lude * r.ldcp + r.lvdcp * r.evapr + r.lsdcp * r.evaps;
    const T src = x.lude + r.evapr + r.evaps;
    const T dqdt = -(r.condl1 + r.condi1) + src * r.gdp;
    const T dtdt = r.lvdcp * r.condl1 + r.lsdcp * r.condi1 - (hh - (r.lsdcp - r.lvdcp) * r.rfreeze1) * r.gdp;
    r.t3 = t + dt * dtdt;
    q = r.q2 + dt * dqdt;
    r.qold = q;
    // :422
    t = r.t3;
    cuadj_fwd_save(e, xk, r.rap, t, q, r.adj);
    r.t_post = t;
    r.q_post = q;
    r.t3_cold = r.t3 < e.RTT;
    r.tpost_cold = t < e.RTT;
    // :425-439
    r.dq = rmax<T>(r.qold - q, T(0.0));
    r.dr2 = kc.cons2 * r.dp * r.dq;
    const bool frz2 = FIX ? r.tpost_cold : r.t3_cold;  // Q4
    T rfreeze2;
    if (frz2) {
        rfreeze2 = r.fwat * r.dr2;
        r.fwatr2 = T(0.0);
        sfln += r.dr2;
    } else {
        rfreeze2 = T(0.0);
        r.fwatr2 = T(1.0);
        rfln += r.dr2;
    }
    r.condl2 = r.condl1 + r.fwatr2 * r.dq * kc.rdt;
    r.condi2 = r.condi1 + (T(1.0) - r.fwatr2) * r.dq * kc.rdt;
    r.rfreeze3 = r.rfreeze1 + rfreeze2;
    // :442-455
    r.tnd_q = -(r.condl2 + r.condi2) + src * r.gdp;
    r.tnd_t = r.lvdcp * r.condl2 + r.lsdcp * r.condi2 - (hh - (r.lsdcp - r.lvdcp) * r.rfreeze3) * r.gdp;
    r.tnd_ql = (qlwc - r.ql) * kc.rdt;
    r.tnd_qi = (qiwc - r.qi) * kc.rdt;
    r.rfln = rfln;
    r.sfln = sfln;
}

// Adjoint forcing of one level: the perturbations of the 10 NL outputs that level k feeds.
template <typename T>
struct ADForce {
    T clc, tnd_q, tnd_qi, tnd_ql, tnd_t;
    // flux and enthalpy-flux forcings at half level k+1, RAW: ad_load_force is the prefetch of the NEXT level, and any
    // arithmetic on a loaded word at the load site makes hipcc wait for it there - a full memory latency on every level
    // (r03: `s_waitcnt vmcnt(11)` right behind the 26 loads of the batch).  ad_flux_forcing combines them (:481-484)
    // where the level is computed.
    T fplsl1, fplsn1, fhpsl1, fhpsn1;
    T covptot;                                            // read by the evaporation block only
};

template <typename T, bool EVAP, typename FP, typename O>
__device__ __forceinline__ ADForce<T> ad_load_force(const FP& F, const Ext<T>& e, O lsb, O o) {
    ADForce<T> f;
    f.clc = ldg(F.adj(NL_OUT_CLC), o);
    f.tnd_q = ldg(F.adj(NL_OUT_TND_Q), o);
    f.tnd_qi = ldg(F.adj(NL_OUT_TND_QI), o);
    f.tnd_ql = ldg(F.adj(NL_OUT_TND_QL), o);
    f.tnd_t = ldg(F.adj(NL_OUT_TND_T), o);
    f.fplsl1 = ldg(F.adj(NL_OUT_FPLSL), o + lsb);
    f.fhpsl1 = ldg(F.adj(NL_OUT_FHPSL), o + lsb);
    f.fplsn1 = ldg(F.adj(NL_OUT_FPLSN), o + lsb);
    f.fhpsn1 = ldg(F.adj(NL_OUT_FHPSN), o + lsb);
    f.covptot = EVAP ? ldg(F.adj(NL_OUT_COVPTOT), o) : T(0.0);
    return f;
}

// :481-484: the flux forcings a level sees are in_fpls*_i - RL*TT * in_fhps*_i (formed when the level is computed).
template <typename T>
__device__ __forceinline__ void ad_flux_forcing(const ADForce<T>& f, const Ext<T>& e, T& fplsl1, T& fplsn1) {
    fplsl1 = f.fplsl1 - f.fhpsl1 * e.RLVTT;
    fplsn1 = f.fplsn1 - f.fhpsn1 * e.RLSTT;
}

template <typename T>
struct ADBack {   // carried from level k+1 to level k in the backward sweep
    T tmp_rfln_i, tmp_sfln_i, rfl_i, sfl_i, daph_i, dp_i;
    T aph_s, covptot_i, aph_s_i;  // EVAP only: surface pressure;: covptot_i of the level below (:653), accumulated tmp_aph_s_i (:686)
};

template <typename T>
struct ADOut {
    T ap, t, q, ql, qi, qsat, lude, mfd, mfu, aph1, lu1;
};

// LDS parking of trajectory values across the first half of ad_backward (CS2_AD_PARK): one 8-byte (4-byte) slot per
// value and lane, [value][lane] - consecutive lanes hit consecutive banks.  The empty asm statements with a memory
// clobber pin the stores before, and the loads after, everything in between: hipcc may neither forward the stored values
// in registers nor hoist the loads back up to the stores.
template <typename T>
__device__ __forceinline__ void ad_park(T* __restrict__ lds, const ADTraj<T>& r) {
    int j = 0;
#define CS2_X(f) lds[(j++) * kColBlock] = r.f;
    CS2_AD_PARK_LIST(CS2_X)
#undef CS2_X
    asm volatile("" ::: "memory");
}
template <typename T>
__device__ __forceinline__ void ad_unpark(const T* __restrict__ lds, ADTraj<T>& r) {
    asm volatile("" ::: "memory");
    int j = 0;
#define CS2_X(f) r.f = lds[(j++) * kColBlock];
    CS2_AD_PARK_LIST(CS2_X)
#undef CS2_X
}

// Backward statements of one level (:494-967 + this level's share of :970-996).  Divisions use the
// reciprocals saved with the trajectory.
template <typename T, bool REG, bool FIX, bool EVAP>
__device__ __forceinline__ ADOut<T> ad_backward(const Ext<T>& e, const NLK<T>& kc, const ADIn<T>& x, int k, T scalm,
                                                T dt, T sfl, ADTraj<T>& r, const ADForce<T>& f, ADBack<T>& b,
                                                const T* park_lds = nullptr) {
    ADOut<T> o;
#if CS2_AD_DIAG == 1
    {
        T d_fl, d_fn;
        ad_flux_forcing<T>(f, e, d_fl, d_fn);
        const T s2 = f.clc + f.tnd_q + f.tnd_qi + f.tnd_ql + f.tnd_t + d_fl + d_fn + r.t2 + sfl + b.tmp_rfln_i;
        o.ap = s2; o.t = s2 + T(1); o.q = s2 + T(2); o.ql = s2 + T(3); o.qi = s2 + T(4); o.qsat = s2 + T(5); o.lude = s2 + T(6);
        o.mfd = s2 + T(7); o.mfu = s2 + T(8); o.aph1 = s2 + T(9); o.lu1 = s2 + T(10);
        b.tmp_rfln_i = s2;
        return o;
    }
#endif
    const T ckcodtla = kc.ckcodtl * T(0.01);
    const T ckcodtia = kc.ckcodti * T(0.01);
    const T cons2 = kc.cons2, rdt = kc.rdt;
    const T lvdcp = r.lvdcp, lsdcp = r.lsdcp, fwat = r.fwat, gdp = r.gdp;
    // :500-501
    T f_fplsl1, f_fplsn1;
    ad_flux_forcing<T>(f, e, f_fplsl1, f_fplsn1);
    T tmp_rfln_i = b.tmp_rfln_i + b.rfl_i + f_fplsl1;
    T tmp_sfln_i = b.tmp_sfln_i + b.sfl_i + f_fplsn1;
    // :504-511
    T o_qi = -f.tnd_qi * rdt;
    T qiwc_i = f.tnd_qi * rdt;
    T o_ql = -f.tnd_ql * rdt;
    T qlwc_i = f.tnd_ql * rdt;
    // :514-533
    const T tt = f.tnd_t;
    const T mix = r.ldcp;  // fwat * lvdcp + (1 - fwat) * lsdcp
    const T evapr = EVAP ? r.evapr : T(0.0), evaps = EVAP ? r.evaps : T(0.0);
    const T hh = x.lude * mix + lvdcp * evapr + lsdcp * evaps;
    const T src = x.lude + evapr + evaps;
    T gdp_i = -tt * (hh - (lsdcp - lvdcp) * r.rfreeze3);
    T condl_i = tt * lvdcp;
    T condi_i = tt * lsdcp;
    T evapr_i = -tt * lvdcp * gdp;
    T evaps_i = -tt * lsdcp * gdp;
    T lvdcp_i = tt * (r.condl2 - evapr * gdp);
    T lsdcp_i = tt * (r.condi2 - evaps * gdp);
    T o_lude = -tt * gdp * mix;
    lvdcp_i -= tt * x.lude * gdp * fwat;
    lsdcp_i -= tt * x.lude * gdp * (T(1.0) - fwat);
    T fwat_i = -tt * x.lude * gdp * (lvdcp - lsdcp);
    lvdcp_i -= tt * r.rfreeze3 * gdp;
    lsdcp_i += tt * r.rfreeze3 * gdp;
    T rfreeze_i = tt * (lsdcp - lvdcp) * gdp;
    // :536-542
    const T tq = f.tnd_q;
    gdp_i += tq * src;
    o_lude += tq * gdp;
    evapr_i += tq * gdp;
    evaps_i += tq * gdp;
    condl_i -= tq;
    condi_i -= tq;
    // :566-592
    T dq_i = (r.fwatr2 * condl_i + (T(1.0) - r.fwatr2) * condi_i) * rdt;
    T dr2_i = r.fwatr2 * tmp_rfln_i + (T(1.0) - r.fwatr2) * tmp_sfln_i;
    if (FIX ? r.tpost_cold : r.t3_cold) {  // :577
        fwat_i += r.dr2 * rfreeze_i;
        dr2_i += fwat * rfreeze_i;
    }
    dq_i += cons2 * r.dp * dr2_i;
    T dp_i = cons2 * r.dq * dr2_i;
    T qold_i, o_q;
    if (r.qold >= r.q_post) {
        if constexpr (REG) dq_i *= T(0.7);
        qold_i = dq_i;
        o_q = -dq_i;
    } else {
        qold_i = T(0.0);
        o_q = T(0.0);
    }
    // :594-598
    T o_ap = T(0.0), o_t = T(0.0);
    cuadj_bwd(e, r.adj, r.rap, o_ap, o_t, o_q);
    // :601-633
    o_q += qold_i;
    const T dqdt_i = dt * o_q;
    const T dtdt_i = dt * o_t;
    gdp_i -= dtdt_i * (hh - (lsdcp - lvdcp) * r.rfreeze1);
    condl_i += dtdt_i * lvdcp;
    condi_i += dtdt_i * lsdcp;
    evapr_i -= dtdt_i * lvdcp * gdp;
    evaps_i -= dtdt_i * lsdcp * gdp;
    lvdcp_i += dtdt_i * (r.condl1 - evapr * gdp);
    lsdcp_i += dtdt_i * (r.condi1 - evaps * gdp);
    o_lude -= dtdt_i * gdp * mix;
    lvdcp_i -= dtdt_i * x.lude * gdp * fwat;
    lsdcp_i -= dtdt_i * x.lude * gdp * (T(1.0) - fwat);
    fwat_i -= dtdt_i * x.lude * gdp * (lvdcp - lsdcp);
    lvdcp_i -= dtdt_i * r.rfreeze1 * gdp;
    lsdcp_i += dtdt_i * r.rfreeze1 * gdp;
    rfreeze_i += dtdt_i * (lsdcp - lvdcp) * gdp;
    gdp_i += dqdt_i * src;
    o_lude += dqdt_i * gdp;
    evapr_i += dqdt_i * gdp;
    evaps_i += dqdt_i * gdp;
    condl_i -= dqdt_i;
    condi_i -= dqdt_i;
    // :635-719 evaporation block; without it (or on a level where it did not run, :710-719)
    // corqs_i = covpclr_i = covptot_i = daph_i = out_qsat_i = prtot_i = qlim_i = 0
    T daph_i = T(0.0);
    T o_qsat = T(0.0);
    T a_clc = f.clc;
    T corqs_i = T(0.0), covpclr_i = T(0.0), covptot_i = T(0.0), qlim_i = T(0.0);
    if constexpr (EVAP) {
        if (r.ev) {
            const T rpr = frcp<T>(r.prtot);
            const T e_evaps_i = evaps_i - tmp_sfln_i;
            tmp_sfln_i += r.dpr * e_evaps_i * rpr;
            T dpr_i = r.sfln2 * e_evaps_i * rpr;
            T prtot_i = -r.dpr * r.sfln2 * e_evaps_i * rpr * rpr;
            const T e_evapr_i = evapr_i - tmp_rfln_i;
            tmp_rfln_i += r.dpr * e_evapr_i * rpr;
            dpr_i += r.rfln2 * e_evapr_i * rpr;
            prtot_i -= r.dpr * r.rfln2 * e_evapr_i * rpr * rpr;
            T cov_i = b.covptot_i + f.covptot;  // :653
            if (r.all_evaporates) {             // preclr <= 0
                a_clc += cov_i;
                cov_i = T(0.0);
            }
            T preclr_i = T(0.0);
            if (r.capped) {  // dpr1 > preclr1
                preclr_i = dpr_i;
                dpr_i = T(0.0);
            }
            const T rdtgdp = r.dp * frcp<T>(kc.rgdt);
            const T b_i = r.covpclr * dpr_i * rdtgdp;
            covpclr_i = r.b * dpr_i * rdtgdp;
            const T dtgdp_i = -r.covpclr * r.b * dpr_i * rdtgdp * rdtgdp;
            daph_i = dt * e.RG * dtgdp_i * r.rdp;
            const T dqe = x.qsat - r.qe;
            // the reference's dt**2 factors are kept (:667-672)
            const T beta_i = dt * dqe * b_i * r.rtmp1 - dt * dt * r.beta * dqe * r.corqs * b_i * r.rtmp1 * r.rtmp1;
            o_qsat = dt * r.beta * b_i * r.rtmp1;
            T qe_i = -dt * r.beta * b_i * r.rtmp1;
            corqs_i = -dt * dt * r.beta * dqe * r.beta * b_i * r.rtmp1 * r.rtmp1;
            const T rcp = frcp<T>(r.covpclr);
            const T raphs = frcp<T>(b.aph_s);
            preclr_i += r.xx * r.sq * beta_i * rcp;
            // sqrt(ap aph_s) = sq aph_s
            o_ap += T(0.5) * r.xx * r.preclr1 * beta_i * rcp * frcp<T>(r.sq) * raphs;
            b.aph_s_i -= T(0.5) * r.xx * r.preclr1 * r.sq * beta_i * rcp * raphs;
            const T rcov1 = frcp<T>(r.covptot1);
            const T romc2 = r.romc * r.romc;
            covpclr_i += -(r.xx * r.preclr1 * r.sq * beta_i * rcp * rcp) - (x.qsat - r.qlim) * qe_i * romc2 +
                         r.prtot * preclr_i * rcov1;
            o_qsat += qe_i - r.covpclr * qe_i * romc2;
            qlim_i = r.covpclr * qe_i * romc2;
            a_clc -= T(2.0) * (x.qsat - r.qlim) * r.covpclr * qe_i * romc2 * r.romc;
            prtot_i += r.covpclr * preclr_i * rcov1;
            cov_i -= r.prtot * r.covpclr * preclr_i * rcov1 * rcov1;
            covptot_i = cov_i;
            // :722-723
            tmp_rfln_i += prtot_i;
            tmp_sfln_i += prtot_i;
        }
    }
    // :722-736
    const T dr_i = r.fwatr1 * tmp_rfln_i + (T(1.0) - r.fwatr1) * tmp_sfln_i;
    T prr_i;
    if (FIX ? (r.fwatr1 == T(0.0)) : r.tpost_cold) {  // :729 (Q5)
        dp_i += rfreeze_i * cons2 * r.prr;
        prr_i = rfreeze_i * cons2 * r.dp;
    } else {
        prr_i = T(0.0);
    }
    prr_i += cons2 * r.dp * dr_i;
    T prs_i = cons2 * r.dp * dr_i;
    dp_i += cons2 * (r.prr + r.prs) * dr_i;
    // :738-782
    if (r.cloudy) {
        prs_i -= qiwc_i;
        qiwc_i += prs_i;
        const T qinew_i = -prs_i;
        a_clc += qinew_i * r.cldi * r.itmp2;
        T cldi_i = qinew_i * r.out_clc * r.itmp2;
        const T di_i = -qinew_i * r.out_clc * r.cldi * r.itmp2;
        const T itmp4 = REG ? ckcodtia : kc.ckcodti;
        o_t += T(0.025) * itmp4 * r.itmp12 * (T(1.0) - r.itmp11) * di_i;
        cldi_i += T(2.0) * itmp4 * r.itmp12 * r.itmp11 * r.cldi * di_i * kc.ricrit * kc.ricrit;
        qiwc_i += cldi_i * r.rclc;
        a_clc -= r.qiwc1 * cldi_i * r.rclc * r.rclc;
        prr_i -= qlwc_i;
        qlwc_i += prr_i;
        const T qlnew_i = -prr_i;
        a_clc += qlnew_i * r.cldl * r.ltmp2;
        T cldl_i = qlnew_i * r.out_clc * r.ltmp2;
        const T dl_i = -qlnew_i * r.out_clc * r.cldl * r.ltmp2;
        const T ltmp4 = REG ? ckcodtla : kc.ckcodtl;
        cldl_i += T(2.0) * ltmp4 * r.ltmp1 * r.cldl * dl_i * kc.rlcrit * kc.rlcrit;
        qlwc_i += cldl_i * r.rclc;
        a_clc -= r.qlwc1 * cldl_i * r.rclc * r.rclc;
    }
    // :785-806 melting of incoming snow;  1/lfdcp = 1/(RLMLT rzz)
    T lfdcp_i;
    if (r.melt) {
        const T snmlt_i = -o_t * r.rcons + tmp_rfln_i - tmp_sfln_i;
        T cons_i = o_t * r.snmlt * r.rcons * r.rcons;
        b.rfl_i = tmp_rfln_i;
        tmp_rfln_i = T(0.0);
        b.sfl_i = tmp_sfln_i;
        tmp_sfln_i = T(0.0);
        T z2s_i;
        if (sfl <= r.z2s) {
            b.sfl_i += snmlt_i;
            z2s_i = T(0.0);
        } else {
            z2s_i = snmlt_i;
        }
        if (r.t2 > kc.meltp2) {
            o_t += r.cons * z2s_i;
            cons_i += (r.t2 - kc.meltp2) * z2s_i;
        }
        const T ilf = kc.rRLMLT * (e.RCPD + e.RCPD * e.RVTMP2 * r.q2);  // 1 / lfdcp
        dp_i += cons2 * cons_i * ilf;
        lfdcp_i = -cons2 * r.dp * cons_i * ilf * ilf;
    } else {
        lfdcp_i = T(0.0);
        b.rfl_i = T(0.0);  // Q8
        b.sfl_i = T(0.0);
    }
    b.tmp_rfln_i = tmp_rfln_i;
    b.tmp_sfln_i = tmp_sfln_i;
    // :810-817 (covpclr_i = covptot_i = 0 without the evaporation block -> no contribution)
    if constexpr (EVAP) {
        if (r.covpclr1 < T(0.0)) covpclr_i = T(0.0);
        covptot_i += covpclr_i;
        a_clc -= covpclr_i;
        if (r.out_clc > r.covptot) {  // :815, literal: tests the cover AFTER the evaporation block's reset
            a_clc += covptot_i;
            covptot_i = T(0.0);
        }
        b.covptot_i = covptot_i;
    }
    if constexpr (kADPark<T>) ad_unpark<T>(park_lds, r);   // the second half starts here: it reads the parked values
    // :820-825
    qiwc_i += condi_i * rdt;
    o_qi -= condi_i * rdt;
    qlwc_i += condl_i * rdt;
    o_ql -= condl_i * rdt;
    T qc_i = fwat * qlwc_i + (T(1.0) - fwat) * qiwc_i;
    fwat_i += r.qc3 * (qlwc_i - qiwc_i);
    // :828-842
    T dqc_i = -qc_i;
    T dqsdz_i, rho_i;
    if (r.lo3) {
        if constexpr (REG) dqc_i *= T(0.1);
        dqsdz_i = dt * dqc_i * (x.mfd + x.mfu) * r.fac4;
        o.mfd = dt * dqc_i * r.dqsdz * r.fac4;
        o.mfu = o.mfd;
        rho_i = -dqc_i * r.dqc * r.fac4;
    } else {
        qc_i += dqc_i;
        dqsdz_i = T(0.0);
        o.mfd = T(0.0);
        o.mfu = T(0.0);
        rho_i = T(0.0);
    }
    // :844-855
    const T dtdzmo_i = dqsdz_i * r.dqsdtemp;
    T dqsdtemp_i = dqsdz_i * r.dtdzmo - r.dtdzmo * dtdzmo_i * r.ldcp * r.fac3;
    const T rodqsdp_i = -e.RG * (dqsdz_i + dtdzmo_i * r.ldcp * r.fac3);
    const T ldcp_i = -dtdzmo_i * (e.RG * r.rodqsdp + r.dtdzmo * r.dqsdtemp) * r.fac3;
    fwat_i += ldcp_i * (lvdcp - lsdcp);
    lvdcp_i += fwat * ldcp_i;
    lsdcp_i += (T(1.0) - fwat) * ldcp_i;
    rho_i -= rodqsdp_i * x.qsat * r.fac2;
    o_qsat -= rodqsdp_i * r.rho * r.fac2;
    o_ap += rodqsdp_i * r.rho * x.qsat * sq(r.fac2) + rho_i * r.fac1;
    T foeew_i = -e.RETV * rodqsdp_i * r.rho * x.qsat * sq(r.fac2);
    o_t -= rho_i * x.ap * r.fac1 * r.rt;
    // :858-877 convective detrainment (exp(-lude/lu1) is the saved exlu; lo1 of the forward part is the same test)
    T lude_i, dlu_i;
    if (k < e.NLEV - 1 && r.lo1) {
        lude_i = qc_i + (T(1.0) - r.clc) * r.rlu * r.exlu * a_clc;
        dlu_i = (T(1.0) - r.clc) * r.lude * r.rlu * r.rlu * r.exlu * a_clc;
        a_clc *= T(1.0) - (T(1.0) - r.exlu);
    } else {
        lude_i = T(0.0);
        dlu_i = T(0.0);
    }
    o_lude += dt * gdp * lude_i;
    gdp_i += dt * x.lude * lude_i;
    daph_i += e.RG * gdp_i * r.rdp * r.rdp;
    // :880-918 Le Treut & Li cloud fraction
    T qt_i = T(0.0), qsat_i, qcrit_i;
    if (r.qt < r.qcrit) {
        qsat_i = T(0.0);
        qcrit_i = T(0.0);
    } else if (r.qt >= r.qsat) {
        qsat_i = (T(1.0) - scalm) * qc_i;
        qcrit_i = -(T(1.0) - scalm) * qc_i;
    } else {
        T qpd_i = scalm * qc_i * sq(r.clc);
        T qcd_i = (T(1.0) - scalm) * qc_i * sq(r.clc);
        a_clc += T(2.0) * (scalm * r.qpd + (T(1.0) - scalm) * r.qcd) * r.clc * qc_i;
        if constexpr (REG) {
            const T rat = r.qpd * frcp<T>(r.qcd);
            const T yyy = rmin<T>(T(0.3), T(3.5) * rsqrt_<T>(rat * cube(T(1.0) - scalm * (T(1.0) - rat))) *
                                              frcp<T>(T(1.0) - scalm));
            a_clc *= yyy;
        }
        // Q10: reached with cls == 2 (tmp3 > 0, rden saved); at qt == qcrit exactly the reference divides by 0
        const T h = T(0.5) * frcp<T>(r.tmp3) * a_clc;
        qpd_i -= h * r.rden;
        qcd_i += h * r.qpd * r.rden * r.rden;
        qt_i = -h * r.qpd * scalm * r.rden * r.rden - qpd_i;
        qcrit_i = h * r.qpd * scalm * r.rden * r.rden - qcd_i;
        qsat_i = qcd_i + qpd_i;
    }
    // :920-938
    o_q += qt_i;
    o_ql += qt_i;
    o_qi += qt_i;
    qsat_i += qcrit_i * r.crh2;
    o_qsat += qsat_i * r.supsat;
    const T supsat_i = qsat_i * x.qsat;
    if (r.t2 < e.RTICE) o_t -= T(0.003) * supsat_i;
    if constexpr (EVAP) {  // :933-938
        if (r.q2 > x.qsat) o_qsat += qlim_i;
        else o_q += qlim_i;
        dqsdtemp_i += kc.cons3 * corqs_i;  // :941
    }
    // :941-967
    o_qsat += r.fac * r.cor * dqsdtemp_i;
    const T cor_i = r.fac * x.qsat * dqsdtemp_i;
    const T fac_i = r.cor * x.qsat * dqsdtemp_i;
    T esdp_i = e.RETV * cor_i * sq(r.cor);
    const T facw_i = fwat * fac_i;
    const T faci_i = (T(1.0) - fwat) * fac_i;
    fwat_i += (r.facw - r.faci) * fac_i;
    o_t -= T(2.0) * (e.R5IES * faci_i * cube(r.ri) + e.R5LES * facw_i * cube(r.rl));
    if (r.esdp1 > e.ZQMAX) esdp_i = T(0.0);
    foeew_i += esdp_i * r.rap;
    o_ap -= esdp_i * r.foeew * r.rap * r.rap;
    if (r.t2_cold) {
        o_t += e.R3IES * (e.RTT - e.R4IES) * foeew_i * r.foeew * r.ri * r.ri;
        o_t += T(0.545) * T(0.17) * fwat_i * r.sech2;
    } else {
        o_t += e.R3LES * (e.RTT - e.R4LES) * foeew_i * r.foeew * r.rl * r.rl;
    }
    // :988-991
    const T zzv = e.RLVTT * lvdcp_i + e.RLSTT * lsdcp_i + e.RLMLT * lfdcp_i;
    o_q += -zzv * e.RCPD * e.RVTMP2 / sq(e.RCPD + e.RCPD * e.RVTMP2 * r.q_post);
    // :970-986 staggered corrections for half level k+1 (uses level k+1's daph_i / dp_i)
    o.aph1 = b.daph_i - daph_i - b.dp_i + dp_i;
    o.lu1 = -dlu_i;
    b.daph_i = daph_i;
    b.dp_i = dp_i;
    o.ap = o_ap;
    o.t = o_t;
    o.q = o_q;
    o.ql = o_ql;
    o.qi = o_qi;
    o.qsat = o_qsat;
    o.lude = o_lude;
    return o;
}

// Kernel arguments as ONE struct (kernarg offset 0): the field pointers are fetched from the kernarg segment at their
// point of use (KernArgs in cloudsc2_common.hpp) instead of living in - and being spilled from - SGPRs.
template <typename T>
struct ADArgs {
    Ext<T> e;
    NLK<T> kc;
    ExpK<T> xk;
    int nx, nz;
    int64_t ls;
    CPtrs<T, NL_NUM_IN> in;
    CPtrs<T, NL_NUM_OUT> adj;
    const T* eta;
    MPtrs<T, NL_NUM_OUT> out;
    MPtrs<T, NL_NUM_IN> oadj;
    T dt;
    int keep_from;
    const T* traj_l;   // TRAJ instantiation only: the rain / snow fluxes ENTERING each level (= out_fplsl / out_fplsn of a
    const T* traj_n;   // cloudsc2_nl / cloudsc2_tl call on the same state), read instead of recomputed by sweep 1
};
#ifndef CS2_AD_KARG
#define CS2_AD_KARG 1   // 1: field pointers are re-read from the kernarg segment (scalar loads) on every level
#endif
template <typename T>
struct ADFields : KernArgs<ADArgs<T>> {
    __device__ __forceinline__ void fresh() { KernArgs<ADArgs<T>>::template fresh<(CS2_AD_KARG != 0)>(); }
    __device__ __forceinline__ const T* in(int i) const { return this->ka->in.p[i]; }
    __device__ __forceinline__ const T* adj(int i) const { return this->ka->adj.p[i]; }
    __device__ __forceinline__ T* out(int i) const { return this->ka->out.p[i]; }
    __device__ __forceinline__ const T* outc(int i) const { return this->ka->out.p[i]; }
    __device__ __forceinline__ T* oadj(int i) const { return this->ka->oadj.p[i]; }
    __device__ __forceinline__ const T* oadjc(int i) const { return this->ka->oadj.p[i]; }
    __device__ __forceinline__ const T* traj_l() const { return this->ka->traj_l; }
    __device__ __forceinline__ const T* traj_n() const { return this->ka->traj_n; }
};

// fp32 without the evaporation block: three waves per SIMD (<= 168 VGPRs) is what the LDS parking was built for (+4.7 %,
// DESIGN 3.5); r03's two extra raw forcing words took the unconstrained allocation to 170 VGPRs = two waves, so it is asked for.
// BIG: 64-bit byte offsets (fields of 4 GiB and more, see offset_t in cloudsc2_common.hpp).
// TRAJ (BUILD EXTENSION, C ABI cloudsc2_ad_from_trajectory_*): the symmetry test calls cloudsc2_tl and then cloudsc2_ad on
// the same state (adjoint/validation.py:135-151), and the only loop-carried trajectory values sweep 2 needs from sweep 1
// are the two precipitation fluxes entering each level - which the TL call has just written as out_fplsl / out_fplsn.
// With TRAJ the kernel skips sweep 1 (no NL outputs are written) and reads those two fields from `traj_l` / `traj_n`:
// 44 words per level and column instead of 70.  Without the evaporation block only (its parked cover has no such source).
template <typename T, bool REG, bool FIX, bool EVAP, bool BIG = false, bool TRAJ = false>
__global__ void __launch_bounds__(kColBlock, (sizeof(T) == 4 && !EVAP) ? 3 : 1)
ad_kernel(const ADArgs<T> A) {
    Ext<T> e = A.e;
    NLK<T> kc = A.kc;
    ExpK<T> xk = A.xk;
    const int nx = A.nx, nz = A.nz, keep_from = A.keep_from;
    const int64_t ls = A.ls;
    const T* __restrict__ eta = A.eta;
    T dt = A.dt;
    ADFields<T> F;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);
    if constexpr (sizeof(T) == 8 && (CS2_AD_PIN & 1)) {
        // fp64 constants of the level loops -> VGPRs (see pin_vgpr in cloudsc2_common.hpp)
        pin_vgpr(e.RCPD); pin_vgpr(e.RLSTT); pin_vgpr(e.RLVTT); pin_vgpr(e.R4LES); pin_vgpr(e.R4IES);
        pin_vgpr(e.RTT); pin_vgpr(e.R3IES); pin_vgpr(e.R3LES); pin_vgpr(e.R2ES); pin_vgpr(e.ZQMAX);
        pin_vgpr(e.RETV); pin_vgpr(e.R5LES); pin_vgpr(e.R5IES); pin_vgpr(e.RG); pin_vgpr(e.RD);
        pin_vgpr(kc.rdt); pin_vgpr(kc.cons2); pin_vgpr(kc.rRD); pin_vgpr(kc.rRCPD); pin_vgpr(dt);
    }
    if constexpr (sizeof(T) == 8 && (CS2_AD_PIN & 2)) {
        pin_vgpr(xk.l2e); pin_vgpr(xk.ln2h); pin_vgpr(xk.ln2l); pin_vgpr(xk.c12); pin_vgpr(xk.c11);
        pin_vgpr(xk.c10); pin_vgpr(xk.c9); pin_vgpr(xk.c8); pin_vgpr(xk.c7); pin_vgpr(xk.c6);
        pin_vgpr(xk.c5); pin_vgpr(xk.c4); pin_vgpr(xk.c3);
    }

    const int gcol = xcd_block() * kColBlock + threadIdx.x;
    // CS2_AD_PARK: this lane's parking slots follow the level table (8-byte aligned)
    T* const park_lds = s_scalm + (nz + 1) + threadIdx.x;
    (void)park_lds;
    if (gcol >= nx) return;  // no later workgroup barrier: whole lanes may retire
    using O = offset_t<BIG>;
#if CS2_AD_DIAG == 2
    const O lsb = nz < 0 ? O(ls) : O(0);   // diagnostics only (wrong results): every level reads and writes level 0 -
#else                                                  // cache-resident rows, the kernel's time without HBM
    const O lsb = O(ls) * O(sizeof(T));
#endif
    const O colb = O(gcol) * O(sizeof(T));

    const T trpaus = trpaus_prescan<T, false, O>(F.in(NL_IN_T), F.in(NL_IN_TND_CML_T), lsb, colb, dt, s_eta, klo, khi);
    const CrhCol<T> crh = crh_setup<T>(trpaus);

    static_assert(!(TRAJ && EVAP), "the trajectory variant has no source for the evaporation block's parked cover");
    // ---------------- sweep 1: trajectory + NL outputs (:146-475)
    const T aph_s = EVAP ? ldg(F.in(NL_IN_APH), O(nz) * lsb + colb) : T(1.0);
#define park F.oadj(NL_IN_MFD)  // EVAP: level k holds the cover entering level k until sweep 2 overwrites it
    if constexpr (!TRAJ) {
        stg(F.out(NL_OUT_FPLSL), colb, T(0.0));
        stg(F.out(NL_OUT_FPLSN), colb, T(0.0));
        stg(F.out(NL_OUT_FHPSL), colb, T(0.0));
        stg(F.out(NL_OUT_FHPSN), colb, T(0.0));
        T rfl = T(0.0), sfl = T(0.0), covptot = T(0.0);
        T aph_k = ldg(F.in(NL_IN_APH), colb);
        O o = colb;
        ADIn<T> xa = ad_load<T>(F, lsb, o, 0 >= keep_from);
        if constexpr (CS2_AD_LANDED != 0) landed(aph_k);
        for (int k = 0; k < nz; ++k) {
            F.fresh();
            ADIn<T> xn = xa;
            const bool keep_n = k + 1 >= keep_from;   // level k+1 (and the fluxes entering it) stay cacheable
            if (k + 1 < nz) xn = ad_load<T>(F, lsb, o + lsb, keep_n);
            ADTraj<T> r;
            ad_forward<T, FIX, EVAP>(e, kc, xk, xa, aph_k, k, s_eta[k], s_scalm[k], crh, dt, rfl, sfl, covptot, aph_s,
                                     r);
            if constexpr (EVAP) stg(park, o, covptot);
            covptot = r.covptot;
            stg(F.out(NL_OUT_CLC), o, r.out_clc);
            stg(F.out(NL_OUT_COVPTOT), o, r.out_covptot);
            stg(F.out(NL_OUT_TND_Q), o, r.tnd_q);
            stg(F.out(NL_OUT_TND_T), o, r.tnd_t);
            stg(F.out(NL_OUT_TND_QL), o, r.tnd_ql);
            stg(F.out(NL_OUT_TND_QI), o, r.tnd_qi);
            stg_sel(F.out(NL_OUT_FPLSL), o + lsb, r.rfln, keep_n);
            stg_sel(F.out(NL_OUT_FPLSN), o + lsb, r.sfln, keep_n);
            stg(F.out(NL_OUT_FHPSL), o + lsb, -r.rfln * e.RLVTT);
            stg(F.out(NL_OUT_FHPSN), o + lsb, -r.sfln * e.RLSTT);
            if constexpr ((CS2_AD_DRAIN & 1) != 0) drain_vmem();
            rfl = r.rfln;
            sfl = r.sfln;
            aph_k = xa.aph1;
            xa = xn;
            o += lsb;
        }
    }   // !TRAJ

    // ---------------- sweep 2: adjoint (:479-996), k = nz-1 .. 0
    ADBack<T> b;
    b.tmp_rfln_i = b.tmp_sfln_i = b.rfl_i = b.sfl_i = b.daph_i = b.dp_i = T(0.0);
    b.covptot_i = b.aph_s_i = T(0.0);
    b.aph_s = aph_s;
    {
        int k = nz - 1;
        O o = O(k) * lsb + colb;
        ADIn<T> xa = ad_load<T>(F, lsb, o, k >= keep_from);
        ADForce<T> fa = ad_load_force<T, EVAP>(F, e, lsb, o);
        T aph_k = ldg_sel(F.in(NL_IN_APH), o, k >= keep_from);
        T sfl = ldg_sel(TRAJ ? F.traj_n() : F.outc(NL_OUT_FPLSN), o, k >= keep_from);
        T rfl = ldg_sel(TRAJ ? F.traj_l() : F.outc(NL_OUT_FPLSL), o, k >= keep_from);
        T cov = EVAP ? ldg(F.oadjc(NL_IN_MFD), o) : T(0.0);
        for (; k >= 0; --k) {
            F.fresh();
            ADIn<T> xn = xa;
            ADForce<T> fn = fa;
            T aph_n = aph_k, sfl_n = sfl, rfl_n = rfl, cov_n = cov;
            if (k > 0) {
                const O om = o - lsb;
                const bool keep_m = k - 1 >= keep_from;
                xn = ad_load<T>(F, lsb, om, keep_m);
                xn.aph1 = aph_k;   // aph[k]: already here as this level's upper half level (the load above is dropped)
                fn = ad_load_force<T, EVAP>(F, e, lsb, om);
                aph_n = ldg_sel(F.in(NL_IN_APH), om, keep_m);
                sfl_n = ldg_sel(TRAJ ? F.traj_n() : F.outc(NL_OUT_FPLSN), om, keep_m);
                rfl_n = ldg_sel(TRAJ ? F.traj_l() : F.outc(NL_OUT_FPLSL), om, keep_m);
                if constexpr (EVAP) cov_n = ldg(F.oadjc(NL_IN_MFD), om);
            }
            ADTraj<T> r;
            ad_forward<T, FIX, EVAP>(e, kc, xk, xa, aph_k, k, s_eta[k], s_scalm[k], crh, dt, rfl, sfl, cov, aph_s, r);
            if constexpr (kADPark<T>) ad_park<T>(park_lds, r);
            const ADOut<T> a = ad_backward<T, REG, FIX, EVAP>(e, kc, xa, k, s_scalm[k], dt, sfl, r, fa, b, park_lds);
            stg(F.oadj(NL_IN_AP), o, a.ap);
            stg(F.oadj(NL_IN_T), o, a.t);
            stg(F.oadj(NL_IN_Q), o, a.q);
            stg(F.oadj(NL_IN_QL), o, a.ql);
            stg(F.oadj(NL_IN_QI), o, a.qi);
            stg(F.oadj(NL_IN_QSAT), o, a.qsat);
            stg(F.oadj(NL_IN_LUDE), o, a.lude);
            stg(F.oadj(NL_IN_MFD), o, a.mfd);
            stg(F.oadj(NL_IN_MFU), o, a.mfu);
            stg(F.oadj(NL_IN_SUPSAT), o, dt * a.q);           // :992 (Q7, literal)
            stg(F.oadj(NL_IN_TND_CML_T), o, dt * a.t);        // :993-996
            stg(F.oadj(NL_IN_TND_CML_Q), o, dt * a.q);
            stg(F.oadj(NL_IN_TND_CML_QL), o, dt * a.ql);
            stg(F.oadj(NL_IN_TND_CML_QI), o, dt * a.qi);
            stg(F.oadj(NL_IN_APH), o + lsb, a.aph1);
            stg(F.oadj(NL_IN_LU), o + lsb, a.lu1);
            if constexpr ((CS2_AD_DRAIN & 2) != 0) drain_vmem();
            xa = xn;
            fa = fn;
            aph_k = aph_n;
            sfl = sfl_n;
            rfl = rfl_n;
            cov = cov_n;
            o -= lsb;
        }
    }
    if constexpr (EVAP) {  // :970-971: out_aph_i[nz] also receives the accumulated tmp_aph_s_i
        const O on = O(nz) * lsb + colb;
        stg(F.oadj(NL_IN_APH), on, ldg(F.oadjc(NL_IN_APH), on) + b.aph_s_i);
    }
    // :982-986 top half level
    stg(F.oadj(NL_IN_APH), colb, b.daph_i - b.dp_i);
    stg(F.oadj(NL_IN_LU), colb, T(0.0));
#undef park
}

// traj_l / traj_n != nullptr: the trajectory variant (TRAJ above); `out` may then be nullptr (nothing of it is touched).
template <typename T>
int launch_ad(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* const* in, const T* const* in_adj,
              const T* eta, T* const* out, T* const* out_adj, double dt, hipStream_t stream, const T* traj_l,
              const T* traj_n) {
    const bool traj = traj_l != nullptr && traj_n != nullptr;
    const bool evap = p.LEVAPLS2 || p.LDRAIN1D;
    const Ext<T> e = make_ext<T>(p);
    CPtrs<T, NL_NUM_IN> ci;
    CPtrs<T, NL_NUM_OUT> ca;
    MPtrs<T, NL_NUM_OUT> co;
    MPtrs<T, NL_NUM_IN> coa;
    for (int i = 0; i < NL_NUM_IN; ++i) { ci.p[i] = in[i]; coa.p[i] = out_adj[i]; }
    for (int i = 0; i < NL_NUM_OUT; ++i) { ca.p[i] = in_adj[i]; co.p[i] = out ? out[i] : nullptr; }
    const dim3 grid((nx + kColBlock - 1) / kColBlock), block(kColBlock);
    const size_t smem = 2 * size_t(nz + 1) * sizeof(T) + (kADPark<T> ? size_t(CS2_AD_PARK_COUNT) * kColBlock * sizeof(T) : 0);
    const T tdt = static_cast<T>(dt);
    const NLK<T> kc = make_nlk<T>(p, dt, evap);
    const ExpK<T> xk = make_expk<T>();
    const bool big = !fits_u32_offsets<T>(nz, ls);
    if (smem > size_t(160) * 1024) return -2;
    int dev = 0;
    if (smem > size_t(64) * 1024)
        if (const int rc = current_device(dev)) return rc;
    const bool reg = p.LREGCL != 0;
    const bool fix = p.AD_TRAJ_FIX != 0;
    // cache-resident turnaround (CS2_AD_KEEP_MB): bottom levels whose 18 re-read words per column fit the budget
    int keep_from = nz;
    if (CS2_AD_KEEP_MB > 0) {
        const uint64_t per_level = uint64_t(18) * uint64_t(ls) * sizeof(T);
        const int levels = int((uint64_t(CS2_AD_KEEP_MB) << 20) / (per_level ? per_level : 1));
        keep_from = levels >= nz ? 0 : nz - levels;
    }
    if (traj && (evap || big)) return -2;   // the trajectory variant: driver switches, 32-bit offsets
    const ADArgs<T> args = {e, kc, xk, nx, nz, ls, ci, ca, eta, co, coa, tdt, keep_from, traj_l, traj_n};
#define CS2_AD_LAUNCH(R, F, E)                                                                                         \
    do {                                                                                                               \
        if (big) CS2_AD_LAUNCH_B(R, F, E, true); else CS2_AD_LAUNCH_B(R, F, E, false);                                 \
    } while (0)
#define CS2_AD_LAUNCH_B(R, F, E, B)                                                                                    \
    do {                                                                                                               \
        auto kern = ad_kernel<T, R, F, E, B>;                                                                             \
        if (smem > size_t(64) * 1024) { /* > 64 KB of dynamic LDS needs the opt-in: once per instantiation and device */ \
            static std::atomic<size_t> attr_set[kMaxDevices] = {};                                                     \
            if (!lds_opt_in(kern, attr_set, dev, smem)) return -1;                                                     \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, grid, block, smem, stream, args);         \
    } while (0)
#define CS2_AD_LAUNCH_T(R, F)                                                                                          \
    do {                                                                                                               \
        auto kern = ad_kernel<T, R, F, false, false, true>;                                                            \
        if (smem > size_t(64) * 1024) {                                                                                \
            static std::atomic<size_t> attr_set[kMaxDevices] = {};                                                     \
            if (!lds_opt_in(kern, attr_set, dev, smem)) return -1;                                                     \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, grid, block, smem, stream, args);                                                     \
    } while (0)
#define CS2_AD_LAUNCH_E(R, F) \
    do {                      \
        if (traj) CS2_AD_LAUNCH_T(R, F); \
        else if (evap) CS2_AD_LAUNCH(R, F, true); else CS2_AD_LAUNCH(R, F, false); \
    } while (0)
    if (reg && !fix) CS2_AD_LAUNCH_E(true, false);
    else if (!reg && !fix) CS2_AD_LAUNCH_E(false, false);
    else if (reg && fix) CS2_AD_LAUNCH_E(true, true);
    else CS2_AD_LAUNCH_E(false, true);
#undef CS2_AD_LAUNCH_E
#undef CS2_AD_LAUNCH_T
#undef CS2_AD_LAUNCH
#undef CS2_AD_LAUNCH_B
    note_kernel(traj ? "cs2::ad_kernel<trajectory>" : big ? "cs2::ad_kernel<big>" : "cs2::ad_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template int launch_ad<double>(const Cloudsc2Params&, int, int, int64_t, const double* const*, const double* const*,
                               const double*, double* const*, double* const*, double, hipStream_t, const double*,
                               const double*);
template int launch_ad<float>(const Cloudsc2Params&, int, int, int64_t, const float* const*, const float* const*,
                              const float*, float* const*, float* const*, double, hipStream_t, const float*, const float*);

}  // namespace cs2
