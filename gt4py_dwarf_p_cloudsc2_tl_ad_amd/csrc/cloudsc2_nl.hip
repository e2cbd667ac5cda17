// cloudsc2_nl as a hand-written CDNA4 kernel.
//
// Restates /root/reference/src/cloudsc2_gt4py/physics/nonlinear/_stencils/cloudsc2.py:93-399
// (line numbers in the comments below refer to that file) for one column per lane:
//
//   prologue   LDS table of eta/scalm, surface pressure, tropopause pre-scan (:93-111)
//   k-sweep    k = 0 .. nz-1, carried scalars rfl / sfl / covptot in registers (:113-388);
//              level k+1's 16 input words are requested from HBM before level k is computed
//              (software prefetch: at 65 536 columns there is exactly one wave per SIMD, so
//              memory latency can only hide behind this wave's own arithmetic);
//   flux shift out_fpls*[k+1] = flux leaving level k, enthalpy fluxes (:391-399) - fused into the
//              sweep, so the fluxes are written once and never re-read.
//
// Two kernels share nl_level / nl_store: nl_kernel (register prefetch; any shape, and the perturbed / Taylor fused
// variants) and nl_ring_kernel (LDS-DMA ring, two levels in flight, tropopause pre-scan overlapped with the sweep;
// 16-byte aligned rows, any nx - the path the headline configuration takes).  launch_nl picks.
//
// Template flags: EVAP = LEVAPLS2 or LDRAIN1D (precipitation evaporation block :288-321 and the
// 1.9*RCLCRIT / 1e-4 thresholds :250-266), LIN = LPHYLIN or LDRAIN1D (:141-155).
#include "cloudsc2_common.hpp"

// Tuning switches (A/B-tested with profiles/ab_nl.py; the defaults are the fastest measured set).
#ifndef CS2_NL_FEXP
#define CS2_NL_FEXP 1   // 1: cs2::fexp (coefficients as kernel arguments), 0: ocml exp
#endif
#ifndef CS2_NL_PINK
#define CS2_NL_PINK 1   // pin the named physical constants in VGPRs (fp64 only)
#endif
#ifndef CS2_NL_DIAG
#define CS2_NL_DIAG 0   // diagnostics only (wrong results): 1 = memory traffic without the physics,
#endif                  // 2 = physics without HBM traffic (inputs from 2 cached levels, no stores)
#ifndef CS2_NL_PINX
#define CS2_NL_PINX 1   // pin the exp coefficients in VGPRs (fp64 only)
#endif

#ifndef CS2_NL_DRAIN
#define CS2_NL_DRAIN 1   // register-path kernel: drain the level's stores before the next level is requested (see drain_vmem)
#endif

namespace cs2 {

template <typename T>
__device__ __forceinline__ T nl_exp(const ExpK<T>& xk, T x) {
#if CS2_NL_FEXP
    return fexp<T>(xk, x);
#else
    return rexp<T>(x);
#endif
}

template <typename T>
struct NLIn {
    T ap, aph1, lu1, lude, mfd, mfu, q, qi, ql, qsat, supsat, t, tq, tqi, tql, tt;
};

// `o` = byte offset of (level k, this lane's column); `lsb` = level stride in bytes (`O`: uint32_t, or uint64_t in the BIG
// instantiation).  SKIPQ: in_qsat is not read (the fused-saturation variant computes it).
template <typename T, typename O>
__device__ __forceinline__ NLIn<T> nl_load_impl(const CPtrs<T, NL_NUM_IN>& in, O lsb, O o, bool skipq,
                                                 bool keepq = false) {
    NLIn<T> x;
    x.ap = ldg(in.p[NL_IN_AP], o);
    x.aph1 = ldg(in.p[NL_IN_APH], o + lsb);
    x.lu1 = ldg(in.p[NL_IN_LU], o + lsb);
    x.lude = ldg(in.p[NL_IN_LUDE], o);
    x.mfd = ldg(in.p[NL_IN_MFD], o);
    x.mfu = ldg(in.p[NL_IN_MFU], o);
    x.q = ldg(in.p[NL_IN_Q], o);
    x.qi = ldg(in.p[NL_IN_QI], o);
    x.ql = ldg(in.p[NL_IN_QL], o);
    // in_qsat was just written by `saturation`: when the field fits the memory-side cache (keepq, launcher) let the load hit it
    x.qsat = skipq ? T(0.0) : (keepq ? ldg_keep(in.p[NL_IN_QSAT], o) : ldg(in.p[NL_IN_QSAT], o));
    x.supsat = ldg(in.p[NL_IN_SUPSAT], o);
    x.t = ldg(in.p[NL_IN_T], o);
    x.tq = ldg(in.p[NL_IN_TND_CML_Q], o);
    x.tqi = ldg(in.p[NL_IN_TND_CML_QI], o);
    x.tql = ldg(in.p[NL_IN_TND_CML_QL], o);
    x.tt = ldg(in.p[NL_IN_TND_CML_T], o);
    return x;
}

template <typename T, bool SKIPQ, typename O>
__device__ __forceinline__ NLIn<T> nl_load(const CPtrs<T, NL_NUM_IN>& in, O lsb, O o, bool keepq = false) {
    return nl_load_impl<T, O>(in, lsb, o, SKIPQ, keepq);
}

// perturbed_state (common/_stencils/perturbed_state.py:75-91) applied on the fly: x + f * x_i.
template <typename T>
__device__ __forceinline__ NLIn<T> nl_perturb(const NLIn<T>& a, const NLIn<T>& b, T f) {
    NLIn<T> x;
    x.ap = a.ap + f * b.ap; x.aph1 = a.aph1 + f * b.aph1; x.lu1 = a.lu1 + f * b.lu1; x.lude = a.lude + f * b.lude;
    x.mfd = a.mfd + f * b.mfd; x.mfu = a.mfu + f * b.mfu; x.q = a.q + f * b.q; x.qi = a.qi + f * b.qi;
    x.ql = a.ql + f * b.ql; x.qsat = a.qsat + f * b.qsat; x.supsat = a.supsat + f * b.supsat; x.t = a.t + f * b.t;
    x.tq = a.tq + f * b.tq; x.tqi = a.tqi + f * b.tqi; x.tql = a.tql + f * b.tql; x.tt = a.tt + f * b.tt;
    return x;
}

// saturation, LPHYLIN form (common/_stencils/saturation.py:30-35,42 + f_foealfa, fcttre.py:22-27): the very function
// the stand-alone saturation kernels evaluate (cs2::saturation_point, contraction pinned), so the fused variant
// reproduces their bits.
template <typename T>
__device__ __forceinline__ T nl_saturation(const Ext<T>& e, const ExpK<T>& xk, T ap, T t) {
    return saturation_point<T, 0>(e, xk, t, ap);
}

template <typename T>
struct NLCarry {
    T rfl, sfl, covptot, aph_k;
};

template <typename T>
struct NLOut {
    T clc, covptot, tnd_q, tnd_t, tnd_ql, tnd_qi, rfln, sfln;
};

// One iteration of the saturation adjustment (nonlinear/_stencils/cuadjtqs.py:24-37) with shared
// reciprocals: r = 1/(t - z4es) serves the exponent and z2s, rap = 1/ap is the caller's.
template <typename T>
__device__ __forceinline__ void nl_cuadj_iter(const Ext<T>& e, const ExpK<T>& xk, T rap, T& t, T& q, T z3es, T z4es, T z5alcp,
                                              T zaldcp) {
    const T r = frcp<T>(t - z4es);
    const T foeew = e.R2ES * nl_exp<T>(xk, z3es * (t - e.RTT) * r);
    T qsat = rmin<T>(foeew * rap, e.ZQMAX);
    const T cor = frcp<T>(T(1.0) - e.RETV * qsat);
    qsat *= cor;
    const T z2s = z5alcp * r * r;
    const T cond = (q - qsat) * frcp<T>(T(1.0) + qsat * cor * z2s);
    t += zaldcp * cond;
    q -= cond;
}

// One level of the forward sweep (:113-388) for one column.  Algebraically the reference's
// statements; divisions are x * frcp(y) with the reciprocals shared (1/zz, 1/(t-R4LES), 1/(t-R4IES),
// 1/ap, 1/t, 1/dp, 1/clc), 0.545 (tanh(u) + 1) is evaluated as 1.09 / (1 + exp(-2u)).
template <typename T, bool EVAP, bool LIN>
__device__ __forceinline__ NLOut<T> nl_level(const Ext<T>& e, const NLK<T>& kc, const ExpK<T>& xk, const NLIn<T>& x, T eta_k, T scalm,
                                             const CrhCol<T>& crh, T dt, T aph_s, NLCarry<T>& c) {
    NLOut<T> o;
#if CS2_NL_DIAG == 1
    {
        const T s1 = x.ap + x.aph1 + x.lu1 + x.lude + x.mfd + x.mfu + x.q + x.qi;
        const T s2 = x.ql + x.qsat + x.supsat + x.t + x.tq + x.tqi + x.tql + x.tt;
        o.clc = s1; o.covptot = s2; o.tnd_q = s1 + s2; o.tnd_t = s1 - s2; o.tnd_ql = s1 * s2; o.tnd_qi = s2 - s1;
        o.rfln = c.rfl + s1; o.sfln = c.sfl + s2; c.rfl = o.rfln; c.sfl = o.sfln; c.aph_k = x.aph1;
        return o;
    }
#endif
    // :104, :115-117 first guess
    T t = x.t + dt * x.tt;
    T q = x.q + dt * x.tq + x.supsat;
    const T ql = x.ql + dt * x.tql;
    const T qi = x.qi + dt * x.tqi;
    // :130-134
    const T dp = x.aph1 - c.aph_k;
    const T rdp = frcp<T>(dp);
    const T zz = e.RCPD + e.RCPD * e.RVTMP2 * q;
    const T rzz = frcp<T>(zz);
    const T lsdcp = e.RLSTT * rzz;
    const T lvdcp = e.RLVTT * rzz;
    // :141-160 dqs/dT correction factor
    const T rl = frcp<T>(t - e.R4LES);
    const T ri = frcp<T>(t - e.R4IES);
    const T rap = frcp<T>(x.ap);
    T fwat, foeew, cor;
    if constexpr (LIN) {
        T z3es, r4;
        if (t < e.RTT) {
            fwat = T(1.09) * frcp<T>(T(1.0) + nl_exp<T>(xk, -kc.fw2 * (t - e.RLPTRC)));
            z3es = e.R3IES;
            r4 = ri;
        } else {
            fwat = T(1.0);
            z3es = e.R3LES;
            r4 = rl;
        }
        foeew = e.R2ES * nl_exp<T>(xk, z3es * (t - e.RTT) * r4);
        const T esdp = foeew * rap;
        cor = (esdp > e.ZQMAX) ? kc.cormax : frcp<T>(T(1.0) - e.RETV * esdp);
    } else {
        // f_foealfa / f_foeewm, common/_stencils/fcttre.py:22-46
        fwat = rmin<T>(T(1.0), sq((rmax<T>(e.RTICE, rmin<T>(e.RTWAT, t)) - e.RTICE) * e.RTWAT_RTICE_R));
        foeew = e.R2ES * (fwat * nl_exp<T>(xk, e.R3LES * (t - e.RTT) * rl) +
                          (T(1.0) - fwat) * nl_exp<T>(xk, e.R3IES * (t - e.RTT) * ri));
        cor = frcp<T>(T(1.0) - e.RETV * (foeew * rap));
    }
    const T facw = e.R5LES * rl * rl;
    const T faci = e.R5IES * ri * ri;
    const T fac = fwat * facw + (T(1.0) - fwat) * faci;
    const T dqsdtemp = fac * x.qsat * cor;
    // :166-186
    const T crh2 = crh2_at(crh, eta_k);
    // :189-193
    const T qsat = (t < e.RTICE) ? x.qsat * (T(1.8) - T(0.003) * t) : x.qsat;
    const T qcrit = crh2 * qsat;
    // :196-207 Le Treut & Li cloud fraction
    const T qt = q + ql + qi;
    T clc, qc;
    if (qt < qcrit) {
        clc = T(0.0);
        qc = T(0.0);
    } else if (qt >= qsat) {
        clc = T(1.0);
        qc = (T(1.0) - scalm) * (qsat - qcrit);
    } else {
        const T qpd = qsat - qt;
        const T qcd = qsat - qcrit;
        clc = T(1.0) - rsqrt_<T>(qpd * frcp<T>(qcd - scalm * (qt - qcrit)));
        qc = (scalm * qpd + (T(1.0) - scalm) * qcd) * sq(clc);
    }
    // :210-215 convective detrainment
    const T gdp = e.RG * rdp;
    const T lude = dt * x.lude * gdp;
    if (lude >= e.RLMIN && x.lu1 >= e.ZEPS2) {
        clc += (T(1.0) - clc) * (T(1.0) - nl_exp<T>(xk, -lude * frcp<T>(x.lu1)));
        qc += lude;
    }
    // :218-224 compensating subsidence
    const T rt = frcp<T>(t);
    const T rho = x.ap * rt * kc.rRD;
    const T rodqsdp = -rho * x.qsat * frcp<T>(x.ap - e.RETV * foeew);
    const T ldcp = fwat * lvdcp + (T(1.0) - fwat) * lsdcp;
    const T dtdzmo = e.RG * (kc.rRCPD - ldcp * rodqsdp) * frcp<T>(T(1.0) + ldcp * dqsdtemp);
    const T dqsdz = dqsdtemp * dtdzmo - e.RG * rodqsdp;
    const T rrho = e.RD * t * rap;
    const T dqc = rmin<T>(dt * dqsdz * (x.mfu + x.mfd) * rrho, qc);
    qc -= dqc;
    // :227-230
    T qlwc = qc * fwat;
    T qiwc = qc * (T(1.0) - fwat);
    T condl = (qlwc - ql) * kc.rdt;
    T condi = (qiwc - qi) * kc.rdt;
    // :234-235 maximum overlap
    c.covptot = rmax<T>(c.covptot, clc);
    // :238-246 melting of incoming snow; cons = cons2 * dp / lfdcp = cons2 * dp * zz / RLMLT
    T rfln, sfln;
    if (c.sfl != T(0.0)) {
        const T cons = kc.cons2 * dp * zz * kc.rRLMLT;
        const T snmlt = rmin<T>(c.sfl, cons * rmax<T>(t - kc.meltp2, T(0.0)));
        rfln = c.rfl + snmlt;
        sfln = c.sfl - snmlt;
        t -= snmlt * (kc.rgdt * rdp * e.RLMLT * rzz);
    } else {
        rfln = c.rfl;
        sfln = c.sfl;
    }
    // :249-272 autoconversion
    T prr = T(0.0), prs = T(0.0);
    if (clc > e.ZEPS2) {
        const T rclc = frcp<T>(clc);
        const T cldl = qlwc * rclc;
        const T dl = kc.ckcodtl * (T(1.0) - nl_exp<T>(xk, -sq(cldl * kc.rlcrit)));
        prr = qlwc - clc * cldl * nl_exp<T>(xk, -dl);
        qlwc -= prr;
        const T cldi = qiwc * rclc;
        const T di = kc.ckcodti * nl_exp<T>(xk, T(0.025) * (t - e.RTT)) * (T(1.0) - nl_exp<T>(xk, -sq(cldi * kc.ricrit)));
        prs = qiwc - clc * cldi * nl_exp<T>(xk, -di);
        qiwc -= prs;
    }
    // :275-285 new precipitation
    const T dr = kc.cons2 * dp * (prr + prs);
    T rfreeze;
    if (t < e.RTT) {
        rfreeze = kc.cons2 * dp * prr;
        sfln += dr;
    } else {
        rfreeze = T(0.0);
        rfln += dr;
    }
    // :288-321 precipitation evaporation
    T evapr = T(0.0), evaps = T(0.0);
    o.covptot = T(0.0);
    if constexpr (EVAP) {
        const T covpclr = rmax<T>(c.covptot - clc, T(0.0));
        const T prtot = rfln + sfln;
        if (prtot > e.ZEPS2 && covpclr > e.ZEPS2) {
            const T corqs = T(1.0) + kc.cons3 * dqsdtemp;  // :160
            const T qlim = rmin<T>(q, x.qsat);             // :163
            T preclr = prtot * covpclr / c.covptot;
            const T qe = x.qsat - (x.qsat - qlim) * covpclr / sq(T(1.0) - clc);
            const T beta = e.RG * e.RPECONS *
                           rpow<T>(rsqrt_<T>(x.ap / aph_s) / T(0.00509) * preclr / covpclr, T(0.5777));
            const T b = dt * beta * (x.qsat - qe) / (T(1.0) + dt * beta * corqs);
            const T dtgdp = dt * e.RG * rdp;
            const T dpr = rmin<T>(covpclr * b / dtgdp, preclr);
            preclr -= dpr;
            if (preclr <= T(0.0)) c.covptot = clc;
            o.covptot = c.covptot;
            // IEEE division: when everything evaporates (dpr == preclr == prtot) the flux must become exactly 0, as in
            // the reference - an approximate reciprocal leaves a 1e-16-relative residue that later levels carry along
            evapr = dpr * rfln / prtot;
            rfln -= evapr;
            evaps = dpr * sfln / prtot;
            sfln -= evaps;
        }
    }
    // :328-344 first-guess T and q after cloud processes
    const T ludeh = x.lude * ldcp;  // in_lude * (fwat * lvdcp + (1 - fwat) * lsdcp)
    const T dqdt = -(condl + condi) + (x.lude + evapr + evaps) * gdp;
    const T dtdt = lvdcp * condl + lsdcp * condi -
                   (lvdcp * evapr + lsdcp * evaps + ludeh - (lsdcp - lvdcp) * rfreeze) * gdp;
    t += dt * dtdt;
    q += dt * dqdt;
    const T qold = q;
    // :347 saturation adjustment (nonlinear/_stencils/cuadjtqs.py:40-68)
    {
        T z3es, z4es, z5alcp, zaldcp;
        if (t > e.RTT) {
            z3es = e.R3LES; z4es = e.R4LES; z5alcp = e.R5ALVCP; zaldcp = e.RALVDCP;
        } else {
            z3es = e.R3IES; z4es = e.R4IES; z5alcp = e.R5ALSCP; zaldcp = e.RALSDCP;
        }
        nl_cuadj_iter(e, xk, rap, t, q, z3es, z4es, z5alcp, zaldcp);
        nl_cuadj_iter(e, xk, rap, t, q, z3es, z4es, z5alcp, zaldcp);
    }
    // :350-364
    const T dq = rmax<T>(qold - q, T(0.0));
    const T dr2 = kc.cons2 * dp * dq;
    if (t < e.RTT) {
        rfreeze += fwat * dr2;
        condi += dq * kc.rdt;
        sfln += dr2;
    } else {
        condl += dq * kc.rdt;
        rfln += dr2;
    }
    // :367-380 output tendencies
    o.clc = clc;
    o.tnd_q = -(condl + condi) + (x.lude + evapr + evaps) * gdp;
    o.tnd_t = lvdcp * condl + lsdcp * condi -
              (lvdcp * evapr + lsdcp * evaps + ludeh - (lsdcp - lvdcp) * rfreeze) * gdp;
    o.tnd_ql = (qlwc - ql) * kc.rdt;
    o.tnd_qi = (qiwc - qi) * kc.rdt;
    // :383-388
    o.rfln = rfln;
    o.sfln = sfln;
    c.rfl = rfln;
    c.sfl = sfln;
    c.aph_k = x.aph1;
    return o;
}

template <typename T, typename O>
__device__ __forceinline__ void nl_store(const MPtrs<T, NL_NUM_OUT>& out, const Ext<T>& e, O lsb, O i,
                                         const NLOut<T>& o) {
    stg(out.p[NL_OUT_CLC], i, o.clc);
    stg(out.p[NL_OUT_COVPTOT], i, o.covptot);
    stg(out.p[NL_OUT_TND_Q], i, o.tnd_q);
    stg(out.p[NL_OUT_TND_T], i, o.tnd_t);
    stg(out.p[NL_OUT_TND_QL], i, o.tnd_ql);
    stg(out.p[NL_OUT_TND_QI], i, o.tnd_qi);
    // :391-399 fluxes leave level k through half level k+1
    stg(out.p[NL_OUT_FPLSL], i + lsb, o.rfln);
    stg(out.p[NL_OUT_FPLSN], i + lsb, o.sfln);
    stg(out.p[NL_OUT_FHPSL], i + lsb, -o.rfln * e.RLVTT);
    stg(out.p[NL_OUT_FHPSN], i + lsb, -o.sfln * e.RLSTT);
}

// Difference of an enthalpy flux against its stored reference, as the unfused sequence forms it: the flux is rounded to T
// first (nl_store writes -flux * L), then the reference is subtracted - no fused multiply-subtract across the two.
template <typename T>
__device__ __forceinline__ T enthalpy_diff(T flux, T latent, T ref) {
#pragma clang fp contract(off)
    const T h = -flux * latent;
    return h - ref;
}

#ifndef CS2_F32_WAVES
#define CS2_F32_WAVES 1   // minimum waves per SIMD requested for the fp32 instantiations (register cap)
#endif
// FUSE selects the fused variants (build extensions; results identical to the
// separate stencil calls):
//   1  `saturation` fused in: in_qsat is not read but computed from (in_ap, in_t) exactly as
//      common/_stencils/saturation.py:29-42 (LPHYLIN form) and written to `qsat_out` - the driver's
//      timed region (saturation + cloudsc2_nl, run_nonlinear.py:117-118) becomes ONE launch;
//   2  `perturbed_state` fused in: every input is read as in + pf * in_i
//      (common/_stencils/perturbed_state.py:75-91) - the Taylor test's ten perturbed NL runs no longer
//      write and re-read a perturbed copy of the 16-field state.
//   3  = 2 + the Taylor test's reduction in the epilogue: nothing is stored; `out` holds the UNPERTURBED NL
//      outputs (read-only) and every workgroup writes the 10 sums  sum_{k, col in block}(NL(x + pf x_i) - NL(x))
//      (tangent_linear/validation.py:239-249) to partials[block][field] in double precision.  One partial per
//      workgroup, summed by the caller: deterministic, no atomics.
// BIG: 64-bit byte offsets (fields of 4 GiB and more, see offset_t); instantiated for FUSE = 0 only.
template <typename T, bool EVAP, bool LIN, bool PINK, int FUSE, bool BIG = false>
__global__ void __launch_bounds__(kColBlock, (sizeof(T) == 4 ? CS2_F32_WAVES : 1))
nl_kernel(Ext<T> e, NLK<T> kc, ExpK<T> xk, int nx, int nz, int64_t ls, CPtrs<T, NL_NUM_IN> in, const T* __restrict__ eta,
          MPtrs<T, NL_NUM_OUT> out, T dt, CPtrs<T, NL_NUM_IN> in_i, T pf, T* __restrict__ qsat_out,
          double* __restrict__ partials, int keepq) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);
    if constexpr (PINK && CS2_NL_PINK) {
        // constants of the level loop -> VGPRs (see pin_vgpr)
        pin_vgpr(e.RCPD); pin_vgpr(e.RLSTT); pin_vgpr(e.RLVTT); pin_vgpr(e.R4LES); pin_vgpr(e.R4IES);
        pin_vgpr(e.RTT); pin_vgpr(e.RLPTRC); pin_vgpr(e.R3IES); pin_vgpr(e.R3LES); pin_vgpr(e.R2ES);
        pin_vgpr(e.ZQMAX); pin_vgpr(e.RETV); pin_vgpr(e.R5LES); pin_vgpr(e.R5IES); pin_vgpr(e.RTICE);
        pin_vgpr(e.RG); pin_vgpr(e.RD); pin_vgpr(e.R5ALVCP); pin_vgpr(e.RALVDCP); pin_vgpr(e.R5ALSCP);
        pin_vgpr(e.RALSDCP); pin_vgpr(kc.rdt); pin_vgpr(kc.cons2); pin_vgpr(kc.rRD); pin_vgpr(kc.rRCPD);
        pin_vgpr(kc.cormax); pin_vgpr(kc.fw2); pin_vgpr(dt);
    }
    if constexpr (PINK && CS2_NL_PINX && CS2_NL_FEXP) {
        pin_vgpr(xk.l2e); pin_vgpr(xk.ln2h); pin_vgpr(xk.ln2l); pin_vgpr(xk.c12); pin_vgpr(xk.c11);
        pin_vgpr(xk.c10); pin_vgpr(xk.c9); pin_vgpr(xk.c8); pin_vgpr(xk.c7); pin_vgpr(xk.c6);
        pin_vgpr(xk.c5); pin_vgpr(xk.c4); pin_vgpr(xk.c3);
    }

    const int gcol = xcd_block() * kColBlock + threadIdx.x;
    // Lanes past the last column retire here.  They must not be carried along under an `if (live)` around the stores: that
    // branch is a merge point for hipcc's wait-count insertion, which then drains every store of a level
    // (`s_waitcnt vmcnt(0)`) before the next level's words are handed over (docs/TUNING_LOG.md 3.9).  FUSE == 3 ends in
    // a workgroup reduction (a barrier): its dead lanes shadow the last column and add 0 to the sums; it has no stores.
    if constexpr (FUSE != 3)
        if (gcol >= nx) return;
#if CS2_NL_DIAG == 2
    const bool live = nz < 0;                // never true at run time: no stores
#else
    constexpr bool live = true;
#endif
    const double wlive = gcol < nx ? 1.0 : 0.0;
    const int col = (gcol < nx) ? gcol : nx - 1;
    using O = offset_t<BIG>;
#if CS2_NL_DIAG == 2
    const O lsb = 0;                         // every level reads level 0 (cache-resident)
#else
    const O lsb = O(ls) * O(sizeof(T));
#endif
    const O colb = O(col) * O(sizeof(T));

    // everything the fused-perturbed variants read outside nl_load is perturbed here as well: the pre-scan's t and
    // tnd_cml_t, aph at the top half level and at the surface
    constexpr bool PERTURBED = FUSE == 2 || FUSE == 3;
    const T trpaus = PERTURBED ? trpaus_prescan<T, true, O>(in.p[NL_IN_T], in.p[NL_IN_TND_CML_T], lsb, colb, dt, s_eta, klo,
                                                         khi, in_i.p[NL_IN_T], in_i.p[NL_IN_TND_CML_T], pf)
                               : trpaus_prescan<T, false, O>(in.p[NL_IN_T], in.p[NL_IN_TND_CML_T], lsb, colb, dt, s_eta, klo, khi);
    const CrhCol<T> crh = crh_setup<T>(trpaus);

    // :93-100
    NLCarry<T> c;
    c.rfl = T(0.0);
    c.sfl = T(0.0);
    c.covptot = T(0.0);
    c.aph_k = ldg(in.p[NL_IN_APH], colb);
    T aph_s = EVAP ? ldg(in.p[NL_IN_APH], O(nz) * lsb + colb) : T(1.0);
    if constexpr (PERTURBED) {
        c.aph_k = c.aph_k + pf * ldg(in_i.p[NL_IN_APH], colb);
        if constexpr (EVAP) aph_s = aph_s + pf * ldg(in_i.p[NL_IN_APH], O(nz) * lsb + colb);
    }

    if (live && FUSE != 3) {
        // top half level: no flux enters the column (:392-394; out_fpls*[0] written as 0, the
        // value the reference relies on from zero-initialised storage - SURVEY.md App. B Q2)
        stg(out.p[NL_OUT_FPLSL], colb, T(0.0));
        stg(out.p[NL_OUT_FPLSN], colb, T(0.0));
        stg(out.p[NL_OUT_FHPSL], colb, T(0.0));
        stg(out.p[NL_OUT_FHPSN], colb, T(0.0));
    }

    // Level sweep with ONE level of software prefetch: level k+1's words are requested before level k is computed and
    // handed over by register copies at the end of the level (`xa = xn`, the shape of tl_kernel / ad_kernel), so the only
    // consumer of a prefetched word is that copy, a whole level of arithmetic later.  r03: the earlier form - a rotating
    // buffer `buf[PD + 1]` in a loop unrolled PD + 1 times, no copies - compiled to `s_waitcnt vmcnt(0)` right behind the
    // loads of every second level (the compiled ISA shows it: 16 / 32 loads, then the wait 0 / 19 instructions later), i.e.
    // every other level paid the full HBM latency.  Deeper register prefetch was measured in r01 (spills; 1 is best).
    constexpr bool PERT = FUSE == 2 || FUSE == 3;
    NLIn<T> xa = nl_load<T, FUSE == 1, O>(in, lsb, colb, keepq != 0);
    NLIn<T> xia;
    if constexpr (PERT) xia = nl_load<T, false, O>(in_i, lsb, colb);
    landed(c.aph_k);   // first read inside the loop: see landed()
    if constexpr (EVAP) landed(aph_s);
    double acc[FUSE == 3 ? NL_NUM_OUT : 1] = {};
    O o = colb;  // byte offset of (level k, column)
    for (int k = 0; k < nz; ++k) {
        NLIn<T> xn = xa, xin = xia;
        if (k + 1 < nz) {
            xn = nl_load<T, FUSE == 1, O>(in, lsb, o + lsb, keepq != 0);
            if constexpr (PERT) xin = nl_load<T, false, O>(in_i, lsb, o + lsb);
        }
        NLIn<T> x = xa;
        if constexpr (PERT) x = nl_perturb<T>(xa, xia, pf);
        T ref[FUSE == 3 ? NL_NUM_OUT : 1];
        if constexpr (FUSE == 3) {   // requested before the physics, consumed after it
#pragma unroll
            for (int f = 0; f < NL_NUM_OUT; ++f) {
                const bool half = f == NL_OUT_FPLSL || f == NL_OUT_FPLSN || f == NL_OUT_FHPSL || f == NL_OUT_FHPSN;
                ref[f] = ldg(const_cast<const T*>(out.p[f]), half ? o + lsb : o);
            }
        }
        if constexpr (FUSE == 1) {
            x.qsat = nl_saturation<T>(e, xk, x.ap, x.t);
            if (live) stg(qsat_out, o, x.qsat);
        }
        const NLOut<T> r = nl_level<T, EVAP, LIN>(e, kc, xk, x, s_eta[k], s_scalm[k], crh, dt, aph_s, c);
        if constexpr (FUSE == 3) {
            // wlive = 1 (exact) or 0 (a lane past the last column: its shadow of column nx-1 is finite)
            acc[NL_OUT_CLC] += wlive * double(r.clc - ref[NL_OUT_CLC]);
            acc[NL_OUT_COVPTOT] += wlive * double(r.covptot - ref[NL_OUT_COVPTOT]);
            acc[NL_OUT_TND_Q] += wlive * double(r.tnd_q - ref[NL_OUT_TND_Q]);
            acc[NL_OUT_TND_T] += wlive * double(r.tnd_t - ref[NL_OUT_TND_T]);
            acc[NL_OUT_TND_QL] += wlive * double(r.tnd_ql - ref[NL_OUT_TND_QL]);
            acc[NL_OUT_TND_QI] += wlive * double(r.tnd_qi - ref[NL_OUT_TND_QI]);
            acc[NL_OUT_FPLSL] += wlive * double(r.rfln - ref[NL_OUT_FPLSL]);
            acc[NL_OUT_FPLSN] += wlive * double(r.sfln - ref[NL_OUT_FPLSN]);
            acc[NL_OUT_FHPSL] += wlive * double(enthalpy_diff<T>(r.rfln, e.RLVTT, ref[NL_OUT_FHPSL]));
            acc[NL_OUT_FHPSN] += wlive * double(enthalpy_diff<T>(r.sfln, e.RLSTT, ref[NL_OUT_FHPSN]));
        } else {
            if (live) nl_store<T, O>(out, e, lsb, o, r);
            if constexpr (CS2_NL_DRAIN != 0) drain_vmem();
        }
        xa = xn;
        if constexpr (PERT) xia = xin;
        o += lsb;
    }
    if constexpr (FUSE == 3) {
        // workgroup reduction of the 10 sums: wave shuffle, then one LDS hop (the level table is no longer needed)
        __shared__ double s_red[kColBlock / 64][NL_NUM_OUT];
#pragma unroll
        for (int f = 0; f < NL_NUM_OUT; ++f) {
            double v = acc[f];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][f] = v;
        }
        __syncthreads();
        if (threadIdx.x < NL_NUM_OUT) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < kColBlock / 64; ++w) v += s_red[w][threadIdx.x];
            partials[size_t(blockIdx.x) * NL_NUM_OUT + threadIdx.x] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-ring variant of the sweep (plain cloudsc2_nl, FUSE = 0).  Same prologue, same nl_level / nl_store; only the
// way the 16 input words of a level reach the lane differs:
//   * every wave owns RD slots of 16 fields x 64 columns in LDS; the inputs of level k + RD - 1 are requested with
//     LDS-DMA (`global_load_lds_dwordx4`, no VGPR destination) while level k is computed, so RD - 1 levels are in
//     flight instead of the one level the register prefetch can afford at 240 VGPRs - at one wave per SIMD
//     (65 536 columns) that is the only source of memory-level parallelism (profiles/microbench_ring.hip);
//   * one DMA instruction moves 16 B per lane = NPL columns (2 fp64 / 4 fp32): the wave's lanes are split into
//     NPL groups, group g fetches field i*NPL + g for the wave's 64 columns, so the LDS image of instruction i
//     is [field i*NPL .. i*NPL+NPL-1][64 columns] (the DMA destination is lane-linear) and field f of a slot
//     starts at f * 64 * sizeof(T); the lane then reads its own column with ds_read_b64 / _b32;
//   * the waits are counted by hand: the DMAs of level k are older than (RD-1) x (NI DMAs + 10 stores), and vmcnt
//     retires in order on gfx9 (the wait used is one level of stores stricter than that, see NFULL).  The wait and the LDS reads live in ONE asm statement with a memory clobber: hipcc
//     would otherwise drain vmcnt(0) before every LDS read that follows an LDS-DMA, and no store may move across
//     the wait (the count must never exceed the operations really issued after level k's DMAs).
// Used when the launcher can guarantee 16-byte aligned rows that hold the last DMA-wide column group (launch_nl; a partly
// filled last wave: RAGGED); every other call takes the register-prefetch kernel above.  Results are bit-identical (same arithmetic on the same words).
#ifndef CS2_NL_RING
#define CS2_NL_RING 3   // slots per wave (levels in flight + the one being computed); 0 disables the variant
#endif
#ifndef CS2_NL_RING_AUX
#define CS2_NL_RING_AUX (CS2_NT & 1 ? 2 : 0)   // cache policy of the input DMAs: 2 = nt (every byte is read once)
#endif
// 1: the sweep's own reads of t / tnd_cml_t on the tropopause-window levels use the default cache policy, hoping to hit
// the rows the overlapped pre-scan fetched.  Measured and left OFF: it gains 9 us only in a back-to-back train of
// launches, i.e. from what the PREVIOUS launch left in the memory-side cache; with the cache evicted between steps
// it is 4 us slower (profiles/cold_cache_check.py).  The committed configuration times the same with or without
// that eviction - only in_qsat, produced inside the step, is meant to be found in cache.
#ifndef CS2_NL_PS_REUSE
#define CS2_NL_PS_REUSE 0
#endif
// Cache policy of the DMA that carries in_qsat when the field fits the memory-side cache (`keepq`, see qsat_fits_cache):
// default (0), not nt - `saturation` wrote the field just before (run_nonlinear.py:117-118).  -1: never, same policy
// as the other inputs.
#ifndef CS2_NL_QSAT_AUX
#define CS2_NL_QSAT_AUX 0
#endif
// Which input fields (bit f = field NL_IN_*) take that policy: the DMA instruction that carries any of them does.  Default:
// in_qsat only.  (A/B switch, r04: in_ap / in_t, which `saturation` read just before, measured in profiles/r04/ab_keep_*.)
#ifndef CS2_NL_KEEP_FIELDS
#define CS2_NL_KEEP_FIELDS (1 << NL_IN_QSAT)
#endif
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <typename T>
struct RingGeom {
    static constexpr int NPL = 16 / int(sizeof(T));               // columns per lane per DMA = fields per DMA
    static constexpr int NI = NL_NUM_IN / NPL;                    // DMA instructions per level
    static constexpr int DATA = NL_NUM_IN * 64 * int(sizeof(T));  // the 16 input fields of one level
    static constexpr int SLOT = DATA + 1024;                      // + the tropopause pre-scan pair (one more DMA)
    static constexpr int NSTORE = NL_NUM_OUT;                     // stores per level (nl_store)
};

// Wait until at most N vector-memory operations are outstanding, then read this lane's column of the slot at LDS
// byte address `a`: the 16 input fields, then the pre-scan pair (first-guess inputs t / tnd_cml_t of level klo + k + 1,
// fields 16 and 17 of the slot); plus the table entries eta[k] (`ta`), scalm[k] (`tb`), eta[klo + k] (`tc`).
template <int N>
__device__ __forceinline__ void ring_read(uint32_t a, uint32_t ta, uint32_t tb, uint32_t tc, NLIn<double>& x, double& eta_k,
                                          double& scalm_k, double& ps_t, double& ps_tt, double& eta_ps) {
    asm volatile(
        "s_waitcnt vmcnt(%25)\n\t"
            "ds_read_b64 %0, %21\n\t"
            "ds_read_b64 %1, %21 offset:512\n\t"
            "ds_read_b64 %2, %21 offset:1024\n\t"
            "ds_read_b64 %3, %21 offset:1536\n\t"
            "ds_read_b64 %4, %21 offset:2048\n\t"
            "ds_read_b64 %5, %21 offset:2560\n\t"
            "ds_read_b64 %6, %21 offset:3072\n\t"
            "ds_read_b64 %7, %21 offset:3584\n\t"
            "ds_read_b64 %8, %21 offset:4096\n\t"
            "ds_read_b64 %9, %21 offset:4608\n\t"
            "ds_read_b64 %10, %21 offset:5120\n\t"
            "ds_read_b64 %11, %21 offset:5632\n\t"
            "ds_read_b64 %12, %21 offset:6144\n\t"
            "ds_read_b64 %13, %21 offset:6656\n\t"
            "ds_read_b64 %14, %21 offset:7168\n\t"
            "ds_read_b64 %15, %21 offset:7680\n\t"
            "ds_read_b64 %18, %21 offset:8192\n\t"
            "ds_read_b64 %19, %21 offset:8704\n\t"
            "ds_read_b64 %16, %22\n\t"
            "ds_read_b64 %17, %23\n\t"
            "ds_read_b64 %20, %24\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x.ap), "=&v"(x.aph1), "=&v"(x.lu1), "=&v"(x.lude), "=&v"(x.mfd), "=&v"(x.mfu), "=&v"(x.q), "=&v"(x.qi),
          "=&v"(x.ql), "=&v"(x.qsat), "=&v"(x.supsat), "=&v"(x.t), "=&v"(x.tq), "=&v"(x.tqi), "=&v"(x.tql), "=&v"(x.tt),
          "=&v"(eta_k), "=&v"(scalm_k), "=&v"(ps_t), "=&v"(ps_tt), "=&v"(eta_ps)
        : "v"(a), "v"(ta), "v"(tb), "v"(tc), "n"(N)
        : "memory");
}
template <int N>
__device__ __forceinline__ void ring_read(uint32_t a, uint32_t ta, uint32_t tb, uint32_t tc, NLIn<float>& x, float& eta_k,
                                          float& scalm_k, float& ps_t, float& ps_tt, float& eta_ps) {
    asm volatile(
        "s_waitcnt vmcnt(%25)\n\t"
            "ds_read_b32 %0, %21\n\t"
            "ds_read_b32 %1, %21 offset:256\n\t"
            "ds_read_b32 %2, %21 offset:512\n\t"
            "ds_read_b32 %3, %21 offset:768\n\t"
            "ds_read_b32 %4, %21 offset:1024\n\t"
            "ds_read_b32 %5, %21 offset:1280\n\t"
            "ds_read_b32 %6, %21 offset:1536\n\t"
            "ds_read_b32 %7, %21 offset:1792\n\t"
            "ds_read_b32 %8, %21 offset:2048\n\t"
            "ds_read_b32 %9, %21 offset:2304\n\t"
            "ds_read_b32 %10, %21 offset:2560\n\t"
            "ds_read_b32 %11, %21 offset:2816\n\t"
            "ds_read_b32 %12, %21 offset:3072\n\t"
            "ds_read_b32 %13, %21 offset:3328\n\t"
            "ds_read_b32 %14, %21 offset:3584\n\t"
            "ds_read_b32 %15, %21 offset:3840\n\t"
            "ds_read_b32 %18, %21 offset:4096\n\t"
            "ds_read_b32 %19, %21 offset:4352\n\t"
            "ds_read_b32 %16, %22\n\t"
            "ds_read_b32 %17, %23\n\t"
            "ds_read_b32 %20, %24\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x.ap), "=&v"(x.aph1), "=&v"(x.lu1), "=&v"(x.lude), "=&v"(x.mfd), "=&v"(x.mfu), "=&v"(x.q), "=&v"(x.qi),
          "=&v"(x.ql), "=&v"(x.qsat), "=&v"(x.supsat), "=&v"(x.t), "=&v"(x.tq), "=&v"(x.tqi), "=&v"(x.tql), "=&v"(x.tt),
          "=&v"(eta_k), "=&v"(scalm_k), "=&v"(ps_t), "=&v"(ps_tt), "=&v"(eta_ps)
        : "v"(a), "v"(ta), "v"(tb), "v"(tc), "n"(N)
        : "memory");
}

// SATF: the fused-saturation variant (FUSE = 1 of nl_kernel): in_qsat is not read - its half of the (ql, qsat) DMA
// re-reads ql, the same bytes the other half fetches - but computed from (ap, t) and written to qsat_out.
// RAGGED (r03): nx is not a multiple of 64 - the last wave is partly filled.  Its dead lanes fetch the last DMA-wide group
// of columns that holds a valid column instead of running off the row (the launcher guarantees lev_stride >= nx rounded up
// to the DMA width, so that group lies inside every row; its columns beyond nx are row padding), compute on that copy and
// do not store.  The stores are then exec-masked but still ISSUED by a
// wave that has any live lane - and a wave without one retires at the top - so the hand-counted waits hold unchanged.
// RAGGED = false is the code of the aligned whole-wave call, instruction for instruction.
template <typename T, bool EVAP, bool LIN, bool PINK, int RD, bool SATF, bool RAGGED = false>
__global__ void __launch_bounds__(kColBlock, 1)
nl_ring_kernel(Ext<T> e, NLK<T> kc, ExpK<T> xk, int nx, int nz, int64_t ls, CPtrs<T, NL_NUM_IN> in,
               const T* __restrict__ eta, MPtrs<T, NL_NUM_OUT> out, T dt, T* __restrict__ qsat_out, int keepq) {
    using G = RingGeom<T>;
    static_assert(kColBlock % 64 == 0 && RD >= 2, "whole waves, at least one level in flight");
    static_assert((RD - 1) * (G::NI + G::NSTORE) < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);
    if constexpr (PINK && CS2_NL_PINK) {
        pin_vgpr(e.RCPD); pin_vgpr(e.RLSTT); pin_vgpr(e.RLVTT); pin_vgpr(e.R4LES); pin_vgpr(e.R4IES);
        pin_vgpr(e.RTT); pin_vgpr(e.RLPTRC); pin_vgpr(e.R3IES); pin_vgpr(e.R3LES); pin_vgpr(e.R2ES);
        pin_vgpr(e.ZQMAX); pin_vgpr(e.RETV); pin_vgpr(e.R5LES); pin_vgpr(e.R5IES); pin_vgpr(e.RTICE);
        pin_vgpr(e.RG); pin_vgpr(e.RD); pin_vgpr(e.R5ALVCP); pin_vgpr(e.RALVDCP); pin_vgpr(e.R5ALSCP);
        pin_vgpr(e.RALSDCP); pin_vgpr(kc.rdt); pin_vgpr(kc.cons2); pin_vgpr(kc.rRD); pin_vgpr(kc.rRCPD);
        pin_vgpr(kc.cormax); pin_vgpr(kc.fw2); pin_vgpr(dt);
    }
    if constexpr (PINK && CS2_NL_PINX && CS2_NL_FEXP) {
        pin_vgpr(xk.l2e); pin_vgpr(xk.ln2h); pin_vgpr(xk.ln2l); pin_vgpr(xk.c12); pin_vgpr(xk.c11);
        pin_vgpr(xk.c10); pin_vgpr(xk.c9); pin_vgpr(xk.c8); pin_vgpr(xk.c7); pin_vgpr(xk.c6);
        pin_vgpr(xk.c5); pin_vgpr(xk.c4); pin_vgpr(xk.c3);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wcol0 = xcd_block() * kColBlock + wave * 64;   // first column of this wave
    if (wcol0 >= nx) return;                                // whole waves retire; the only workgroup barrier is inside
                                                            // build_level_table
    const uint32_t lsb = uint32_t(ls) * uint32_t(sizeof(T));
    const bool live = !RAGGED || wcol0 + lane < nx;         // RAGGED: lanes beyond nx shadow the last column, stores masked
    const uint32_t colb = uint32_t(live ? wcol0 + lane : nx - 1) * uint32_t(sizeof(T));

    // Tropopause pre-scan (:107-111), overlapped with the sweep.  crh2 (:166-186) is 1 on every level with
    // eta <= 0.1 whatever trpaus (>= 0.1) turns out to be, i.e. on all levels above the window [klo, khi] of a
    // top-down eta grid; the window's first-guess temperatures are therefore fetched by ONE extra DMA per level while
    // levels 0 .. nps-1 are computed, and trpaus is final before the sweep reaches level klo.  Needs nps <= klo and
    // eta <= 0.1 above the window (uniform tests); otherwise the pre-scan runs here, before the sweep.
    const int nps = khi >= klo ? khi - klo + 1 : 0;
    bool overlap = nps > 0 && nps <= klo;
    for (int k = 0; k < klo && overlap; ++k) overlap = s_eta[k] <= T(0.1);
    T trpaus = T(0.1);
    T tk_ps = T(0.0);
    if (overlap) {
        const uint32_t o0 = uint32_t(klo) * lsb + colb;
        tk_ps = ldg(in.p[NL_IN_T], o0) + dt * ldg(in.p[NL_IN_TND_CML_T], o0);
    } else {
        trpaus = trpaus_prescan<T>(in.p[NL_IN_T], in.p[NL_IN_TND_CML_T], lsb, colb, dt, s_eta, klo, khi);
    }
    CrhCol<T> crh = crh_setup<T>(overlap ? T(2.0) : trpaus);   // trpaus = 2 > every eta: crh2_at returns 1

    // :93-100
    NLCarry<T> c;
    c.rfl = T(0.0);
    c.sfl = T(0.0);
    c.covptot = T(0.0);
    c.aph_k = ldg(in.p[NL_IN_APH], colb);
    const T aph_s = EVAP ? ldg(in.p[NL_IN_APH], uint32_t(nz) * lsb + colb) : T(1.0);
    if (live) {
        stg(out.p[NL_OUT_FPLSL], colb, T(0.0));
        stg(out.p[NL_OUT_FPLSN], colb, T(0.0));
        stg(out.p[NL_OUT_FHPSL], colb, T(0.0));
        stg(out.p[NL_OUT_FHPSN], colb, T(0.0));
    }
    // consume the prologue's ordinary loads BEFORE the first DMA is issued: hipcc drains vmcnt(0) at the first use of
    // an ordinary load's result while an LDS-DMA is in flight, which would empty the ring inside level 0
    pin_vgpr(c.aph_k);
    pin_vgpr(tk_ps);
    if constexpr (EVAP) { T a = aph_s; pin_vgpr(a); }

    // per-lane DMA sources: lane group g of instruction i walks field i*NPL + g, NPL adjacent columns per lane;
    // aph and lu are read one half level below (aph[k+1], lu[k+1]: :130, :212)
    constexpr int LPG = 64 / G::NPL;   // lanes per group
    const int g = lane / LPG, l = lane % LPG;
    // first of this lane's NPL columns; RAGGED: a lane whose group lies beyond nx fetches the last group that holds a valid
    // column instead (a group boundary, so 16-byte aligned like every other source; inside the row: see the launcher)
    const int last_group = ((nx - 1) / G::NPL) * G::NPL;
    const int dcol = (RAGGED && wcol0 + G::NPL * l > last_group) ? last_group : wcol0 + G::NPL * l;
    const char* src[G::NI];
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        const T* base = in.p[i * G::NPL];
        int f = i * G::NPL;
#pragma unroll
        for (int j = 1; j < G::NPL; ++j)
            if (g == j) {
                base = in.p[i * G::NPL + j];
                f = i * G::NPL + j;
            }
        if (SATF && f == NL_IN_QSAT) base = in.p[NL_IN_QL];
        const uint32_t lev1 = (f == NL_IN_APH || f == NL_IN_LU) ? lsb : 0u;
        src[i] = reinterpret_cast<const char*>(base) + (uint32_t(dcol) * uint32_t(sizeof(T)) + lev1);
    }
    // the pre-scan pair: lane groups alternate between t and tnd_cml_t of level klo + 1 + (level being fetched)
    const char* src_ps = reinterpret_cast<const char*>((g & 1) ? in.p[NL_IN_TND_CML_T] : in.p[NL_IN_T]) +
                         (uint32_t(dcol) * uint32_t(sizeof(T)) + uint32_t(klo + 1) * lsb);
    const int nps_dma = overlap ? nps : 0;
    // LDS: [eta | scalm table][pad][wave 0: RD slots][wave 1: RD slots] ...
    const uint32_t tab_bytes = (2u * uint32_t(nz + 1) * uint32_t(sizeof(T)) + 1023u) & ~1023u;
    const uint32_t ring0 = tab_bytes + uint32_t(wave) * uint32_t(RD * G::SLOT);
    auto issue = [&](int slot, int level) {
        const bool in_window = nps_dma > 0 && level > klo && level <= khi + 1;   // uniform
#pragma unroll
        for (int i = 0; i < G::NI; ++i) {
            // (the cache-policy operand must be a literal constant at each call site)
            if (CS2_NL_QSAT_AUX >= 0 && keepq && ((CS2_NL_KEEP_FIELDS >> (i * G::NPL)) & ((1 << G::NPL) - 1)) != 0)   // uniform
                __builtin_amdgcn_global_load_lds((glb_void_ptr)src[i],
                                                 (lds_void_ptr)(&smem_raw[ring0 + uint32_t(slot * G::SLOT + i * 1024)]),
                                                 16, 0, CS2_NL_QSAT_AUX >= 0 ? CS2_NL_QSAT_AUX : 0);
            else if (CS2_NL_PS_REUSE && in_window && (i == NL_IN_T / G::NPL || i == NL_IN_TND_CML_T / G::NPL))
                // the pre-scan DMA fetched this row of t / tnd_cml_t (default policy) a few tens of levels ago
                __builtin_amdgcn_global_load_lds((glb_void_ptr)src[i],
                                                 (lds_void_ptr)(&smem_raw[ring0 + uint32_t(slot * G::SLOT + i * 1024)]),
                                                 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((glb_void_ptr)src[i],
                                                 (lds_void_ptr)(&smem_raw[ring0 + uint32_t(slot * G::SLOT + i * 1024)]),
                                                 16, 0, CS2_NL_RING_AUX);
            src[i] += lsb;
        }
        if (level < nps_dma) {   // uniform
            __builtin_amdgcn_global_load_lds((glb_void_ptr)src_ps,
                                             (lds_void_ptr)(&smem_raw[ring0 + uint32_t(slot * G::SLOT + G::DATA)]), 16, 0,
                                             0);   // default cache policy: the sweep reads these rows again
            src_ps += lsb;
        }
    };
#pragma unroll
    for (int j = 0; j < RD - 1; ++j)
        if (j < nz) issue(j, j);

    // Operations younger than level k's DMAs when level k is read (k >= RD-1): (RD-1) x (NI DMAs + NSTORE stores).
    // The wait must never ALLOW more than were really issued, so it is set one level of stores short of that: the
    // stores it additionally retires are >= RD-1 levels old (long complete), and the count stays valid even if a
    // future compiler merged or dropped stores of a level.  First RD-1 levels: no stores counted at all.
    constexpr int NFULL = (RD - 1) * G::NI + (RD - 2) * G::NSTORE;
    constexpr int NHEAD = (RD - 1) * G::NI;
    const uint32_t rd_lane = ring0 + uint32_t(lane) * uint32_t(sizeof(T));
    const uint32_t tb_off = uint32_t(nz + 1) * uint32_t(sizeof(T));
    uint32_t o = colb;
    int slot = 0, pslot = RD - 1;
    for (int k = 0; k < nz; ++k) {
        const bool more = k + RD - 1 < nz;
        if (more) issue(pslot, k + RD - 1);
        NLIn<T> x;
        T eta_k, scalm_k, ps_t, ps_tt, eta_ps;
        const uint32_t a = rd_lane + uint32_t(slot * G::SLOT);
        const uint32_t ta = uint32_t(k) * uint32_t(sizeof(T));
        const uint32_t tc = uint32_t(klo + k < nz ? klo + k : nz) * uint32_t(sizeof(T));
        if (!more) ring_read<0>(a, ta, ta + tb_off, tc, x, eta_k, scalm_k, ps_t, ps_tt, eta_ps);   // tail: drain
        else if (k < RD - 1) ring_read<NHEAD>(a, ta, ta + tb_off, tc, x, eta_k, scalm_k, ps_t, ps_tt, eta_ps);
        // RAGGED: the level's stores are exec-masked (`if (live)`), so their COUNT is not something the wait may rely on -
        // the steady state counts none (the NHEAD wait: every store of earlier levels retired).  Not the headline path.
        else ring_read<(RAGGED ? NHEAD : NFULL)>(a, ta, ta + tb_off, tc, x, eta_k, scalm_k, ps_t, ps_tt, eta_ps);
        if (k < nps_dma) {            // pre-scan of level klo + k (:107-111); the slot holds level klo + k + 1
            const T tk1 = ps_t + dt * ps_tt;
            if (eta_ps > T(0.1) && eta_ps < T(0.4) && tk_ps > tk1) trpaus = eta_ps;
            tk_ps = tk1;
        } else if (k == nps_dma && overlap) {
            crh = crh_setup<T>(trpaus);   // final before the sweep reaches level klo (nps <= klo)
        }
        if constexpr (SATF) {
            x.qsat = nl_saturation<T>(e, xk, x.ap, x.t);
            if (live) stg(qsat_out, o, x.qsat);   // an 11th store per level: not counted in NFULL (under-counting is safe)
        }
        const NLOut<T> r = nl_level<T, EVAP, LIN>(e, kc, xk, x, eta_k, scalm_k, crh, dt, aph_s, c);
        if (live) nl_store<T, uint32_t>(out, e, lsb, o, r);
        o += lsb;
        slot = slot + 1 == RD ? 0 : slot + 1;
        pslot = pslot + 1 == RD ? 0 : pslot + 1;
    }
}

template <typename T>
int launch_nl(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* const* in, const T* eta, T* const* out,
              double dt, hipStream_t stream, const T* const* in_i = nullptr, double pf = 0.0, T* qsat_out = nullptr,
              double* partials = nullptr) {
    const Ext<T> e = make_ext<T>(p);
    CPtrs<T, NL_NUM_IN> ci, cii;
    MPtrs<T, NL_NUM_OUT> co;
    for (int i = 0; i < NL_NUM_IN; ++i) {
        ci.p[i] = in[i];
        cii.p[i] = in_i ? in_i[i] : nullptr;
    }
    for (int i = 0; i < NL_NUM_OUT; ++i) co.p[i] = out[i];
    const dim3 grid((nx + kColBlock - 1) / kColBlock), block(kColBlock);
    const size_t smem = 2 * size_t(nz + 1) * sizeof(T);
    const bool evap = p.LEVAPLS2 || p.LDRAIN1D;
    const bool lin = p.LPHYLIN || p.LDRAIN1D;
    const T tdt = static_cast<T>(dt);
    const T tpf = static_cast<T>(pf);
    const NLK<T> kc = make_nlk<T>(p, dt, evap);
    const ExpK<T> xk = make_expk<T>();
    const int fuse = qsat_out ? 1 : (in_i ? (partials ? 3 : 2) : 0);
    const bool big = !fits_u32_offsets<T>(nz, ls);
    if (big && fuse != 0) return -2;          // the fused build extensions keep 32-bit offsets
    const int keepq = qsat_fits_cache<T>(nz, ls) ? 1 : 0;   // in_qsat: default cache policy only when the field fits
    if (fuse == 1 && !p.LPHYLIN) return -2;   // only the LPHYLIN form of `saturation` is fused
#define CS2_NL_LAUNCH(EV, LN, FU)                                                                                 \
    hipLaunchKernelGGL((nl_kernel<T, EV, LN, sizeof(T) == 8, FU>), grid, block, smem, stream, e, kc, xk, nx, nz, \
                       ls, ci, eta, co, tdt, cii, tpf, qsat_out, partials, keepq)
#define CS2_NL_FLAGS(FU)                                   \
    do {                                                   \
        if (evap && lin) CS2_NL_LAUNCH(true, true, FU);    \
        else if (evap && !lin) CS2_NL_LAUNCH(true, false, FU); \
        else if (!evap && lin) CS2_NL_LAUNCH(false, true, FU); \
        else CS2_NL_LAUNCH(false, false, FU);              \
    } while (0)
#if CS2_NL_RING
    // LDS-ring variant: whole waves, 16-byte aligned rows of every input field (the DMA moves 16 B per lane).
    // Ring depth by grid size (profiles/ab_nl.py, fp64, same box): with at most ~1.5 workgroups per CU the deep ring
    // wins (32 768 columns: 244 us vs 259 us at depth 2 vs 299 us register prefetch; 98 304: 609 / 633 / 651 us); with
    // more, LDS occupancy matters more than depth - depth 2 keeps two workgroups resident per CU (131 072 columns:
    // 655 us vs 692 us at depth 3; 262 144: 1 286 vs 1 350 us).
    bool ring_deep = true;
    // 16-byte aligned rows that hold whole DMA-wide groups of columns up to the last valid one (lev_stride >= nx rounded up
    // to 2 fp64 / 4 fp32 columns: `storage.zeros` pads the level pitch to 512 B); a partly filled last wave takes the
    // RAGGED instantiation
    constexpr int kNPL = RingGeom<T>::NPL;
    bool ring = !big && fuse <= 1 && nx > 0 && nz >= CS2_NL_RING && (ls * int64_t(sizeof(T))) % 16 == 0 &&
                ls >= int64_t((nx + kNPL - 1) / kNPL) * kNPL;
    const bool ragged = nx % 64 != 0;
    for (int i = 0; i < NL_NUM_IN && ring; ++i)
        ring = (fuse == 1 && i == NL_IN_QSAT) || reinterpret_cast<uintptr_t>(in[i]) % 16 == 0;
    if (ring) {
        using G = RingGeom<T>;
        int dev = 0;
        if (const int rc = current_device(dev)) return rc;
        const bool deep = int64_t(grid.x) * 2 <= int64_t(device_cus(dev)) * 3;
        ring_deep = deep;
        const int depth = deep ? CS2_NL_RING : 2;
        const size_t tab = (2 * size_t(nz + 1) * sizeof(T) + 1023) & ~size_t(1023);
        const size_t rsmem = tab + size_t(kColBlock / 64) * depth * G::SLOT;
        ring = rsmem <= size_t(160) * 1024;   // LDS of a CU; very tall columns (table > 46 KB) take the register path
    }
    if (ring) {
        using G = RingGeom<T>;
        int dev = 0;
        if (const int rc = current_device(dev)) return rc;
        const bool deep = ring_deep;
        const int depth = deep ? CS2_NL_RING : 2;
        const size_t tab = (2 * size_t(nz + 1) * sizeof(T) + 1023) & ~size_t(1023);
        const size_t rsmem = tab + size_t(kColBlock / 64) * depth * G::SLOT;
#define CS2_NL_RING_LAUNCH(EV, LN, RD, SF)                                                                           \
    do {                                                                                                             \
        if (ragged) CS2_NL_RING_LAUNCH_R(EV, LN, RD, SF, true);                                                      \
        else CS2_NL_RING_LAUNCH_R(EV, LN, RD, SF, false);                                                            \
    } while (0)
#define CS2_NL_RING_LAUNCH_R(EV, LN, RD, SF, RG)                                                                     \
    do {                                                                                                             \
        auto kern = nl_ring_kernel<T, EV, LN, sizeof(T) == 8, RD, SF, RG>;                                           \
        /* > 64 KB of dynamic LDS needs the opt-in: once per instantiation, device and size */                       \
        static std::atomic<size_t> attr_set[kMaxDevices] = {};                                                       \
        if (!lds_opt_in(kern, attr_set, dev, rsmem)) return -1;                                                      \
        hipLaunchKernelGGL(kern, grid, block, rsmem, stream, e, kc, xk, nx, nz, ls, ci, eta, co, tdt, qsat_out,      \
                           keepq);                                                                                   \
    } while (0)
#define CS2_NL_RING_FLAGS(RD, SF)                                          \
    do {                                                                   \
        if (evap && lin) CS2_NL_RING_LAUNCH(true, true, RD, SF);           \
        else if (evap && !lin) CS2_NL_RING_LAUNCH(true, false, RD, SF);    \
        else if (!evap && lin) CS2_NL_RING_LAUNCH(false, true, RD, SF);    \
        else CS2_NL_RING_LAUNCH(false, false, RD, SF);                     \
    } while (0)
        if (fuse == 1 && !p.LPHYLIN) return -2;   // only the LPHYLIN form of `saturation` is fused
        if (fuse == 0) {
            if (deep) CS2_NL_RING_FLAGS(CS2_NL_RING, false);
            else CS2_NL_RING_FLAGS(2, false);
        } else {
            if (deep) CS2_NL_RING_FLAGS(CS2_NL_RING, true);
            else CS2_NL_RING_FLAGS(2, true);
        }
#undef CS2_NL_RING_FLAGS
#undef CS2_NL_RING_LAUNCH
#undef CS2_NL_RING_LAUNCH_R
        note_kernel(ragged ? "cs2::nl_ring_kernel<ragged>" : "cs2::nl_ring_kernel");
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
#endif
    if (big) {       // fields of 4 GiB and more: the register-path kernel with 64-bit offsets
#define CS2_NL_BIG(EV, LN)                                                                                         \
    hipLaunchKernelGGL((nl_kernel<T, EV, LN, sizeof(T) == 8, 0, true>), grid, block, smem, stream, e, kc, xk, nx,  \
                       nz, ls, ci, eta, co, tdt, cii, tpf, qsat_out, partials, keepq)
        if (evap && lin) CS2_NL_BIG(true, true);
        else if (evap && !lin) CS2_NL_BIG(true, false);
        else if (!evap && lin) CS2_NL_BIG(false, true);
        else CS2_NL_BIG(false, false);
#undef CS2_NL_BIG
        note_kernel("cs2::nl_kernel<big>");
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (fuse == 0) CS2_NL_FLAGS(0);
    else if (fuse == 1) CS2_NL_FLAGS(1);
    else if (fuse == 2) CS2_NL_FLAGS(2);
    else CS2_NL_FLAGS(3);
#undef CS2_NL_FLAGS
#undef CS2_NL_LAUNCH
    note_kernel("cs2::nl_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template int launch_nl<double>(const Cloudsc2Params&, int, int, int64_t, const double* const*, const double*,
                               double* const*, double, hipStream_t, const double* const*, double, double*, double*);
template int launch_nl<float>(const Cloudsc2Params&, int, int, int64_t, const float* const*, const float*,
                              float* const*, double, hipStream_t, const float* const*, double, float*, double*);

// ------------------------------------------------------------------------------------------------
// The Taylor test's perturbed runs for NF step sizes in ONE launch (build extension, C ABI cloudsc2_nl_taylor_multi_*;
// tangent_linear/validation.py:162-176 + :239-249 of the reference).  The perturbed NL run of one step size is bound by
// its 42 words per level and column (16 state + 16 increment fields read, 10 outputs written or 10 reference outputs
// read), ten times per Taylor run.  Here a lane loads the 42 words of a level ONCE and evaluates the level for NF step
// sizes on them - NF independent carried states (rfl, sfl, covptot, aph_k) and tropopause profiles in registers - so the
// ten perturbed runs cost 2 x 42 words instead of 10 x 42 and the kernel becomes bound by its arithmetic (NF x the NL
// physics at one wave per SIMD), which also gives the VALU NF independent chains to interleave.  The 10 x NF running sums
// sum_k (NL(x + f_j x_i) - NL(x)) live in LDS, [sum][lane] (conflict-free, no VGPRs), and are reduced per workgroup exactly
// as the one-step variant (nl_kernel FUSE = 3) does: partials[(block * nf_total + f0 + j) * 10 + field], fixed order,
// no atomics.  Same nl_level / nl_perturb / trpaus arithmetic as the one-step kernels on the same words.
template <typename T, int NF>
struct PFs { T f[NF]; };

// INC (here and in the kernel): the increment fields are not read but formed as finc * state (state_increment fused in,
// common/_stencils/state_increment.py:61-80 - the same products the stand-alone increment kernel stores)
template <typename T, int NF, bool INC>
__device__ __forceinline__ void trpaus_prescan_multi(const T* __restrict__ pt, const T* __restrict__ ptt,
                                                     const T* __restrict__ pt_i, const T* __restrict__ ptt_i, uint32_t lsb,
                                                     uint32_t colb, T dt, const T* s_eta, int klo, int khi,
                                                     const PFs<T, NF>& pf, T finc, T (&trpaus)[NF]) {
#pragma unroll
    for (int j = 0; j < NF; ++j) trpaus[j] = T(0.1);
    if (klo > khi) return;
    constexpr int CH = 8;
    const uint32_t o0 = uint32_t(klo) * lsb + colb;
    const T a0 = ldg(pt, o0), b0 = ldg(ptt, o0);
    const T ai0 = INC ? rounded_product<T>(finc, a0) : ldg(pt_i, o0);
    const T bi0 = INC ? rounded_product<T>(finc, b0) : ldg(ptt_i, o0);
    T tk[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const T t0 = a0 + pf.f[j] * ai0;
        const T tt0 = b0 + pf.f[j] * bi0;
        tk[j] = t0 + dt * tt0;
    }
    for (int k0 = klo; k0 <= khi; k0 += CH) {
        T a[CH], b[CH], ai[CH], bi[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int kk = (k0 + i < khi ? k0 + i : khi) + 1;   // the tail re-reads level khi + 1 (cache hits)
            const uint32_t oi = uint32_t(kk) * lsb + colb;
            a[i] = ldg(pt, oi);
            b[i] = ldg(ptt, oi);
            if constexpr (!INC) {
                ai[i] = ldg(pt_i, oi);
                bi[i] = ldg(ptt_i, oi);
            }
        }
        if constexpr (INC) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                ai[i] = rounded_product<T>(finc, a[i]);
                bi[i] = rounded_product<T>(finc, b[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = k0 + i;
            if (k <= khi) {
                const T ek = s_eta[k];
                const bool in_window = ek > T(0.1) && ek < T(0.4);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const T aj = a[i] + pf.f[j] * ai[i];
                    const T bj = b[i] + pf.f[j] * bi[i];
                    const T tk1 = aj + dt * bj;
                    if (in_window && tk[j] > tk1) trpaus[j] = ek;
                    tk[j] = tk1;
                }
            }
        }
    }
}

template <typename T>
__device__ __forceinline__ NLIn<T> nl_increment(const NLIn<T>& a, T f, bool zero_supsat) {
    NLIn<T> x;   // rounded_product: the STORED products of the increment kernel, never half of an fma (nl_perturb's x + f2 * x_i)
#define CS2_P(m) x.m = rounded_product<T>(f, a.m)
    CS2_P(ap); CS2_P(aph1); CS2_P(lu1); CS2_P(lude); CS2_P(mfd); CS2_P(mfu); CS2_P(q); CS2_P(qi); CS2_P(ql); CS2_P(qsat);
    CS2_P(t); CS2_P(tq); CS2_P(tqi); CS2_P(tql); CS2_P(tt);
#undef CS2_P
    x.supsat = zero_supsat ? T(0.0) : rounded_product<T>(f, a.supsat);
    return x;
}

template <typename T>
__device__ __forceinline__ void nl_load_refs(const CPtrs<T, NL_NUM_OUT>& ref, uint32_t lsb, uint32_t o, T (&r)[NL_NUM_OUT]) {
#pragma unroll
    for (int f = 0; f < NL_NUM_OUT; ++f) {
        const bool half = f == NL_OUT_FPLSL || f == NL_OUT_FPLSN || f == NL_OUT_FHPSL || f == NL_OUT_FHPSN;
        r[f] = ldg(ref.p[f], half ? o + lsb : o);
    }
}

// The running sums of one step size: ten fp64 words in LDS, field f of this lane at `addr + f * kColBlock * 8`.  Left to
// hipcc, every `sum += d` became ds_read / s_waitcnt lgkmcnt(0) / add / ds_write back to back - fifty exposed LDS round
// trips per level, a third of the kernel's time at one wave per SIMD (PMC: VALU-active 69 %).  So the ten reads of a step
// size are ISSUED here, before its level is evaluated, and only waited for (acc_wait) when the differences are ready: the
// latency hides behind ~570 VALU instructions.  The asm is a compiler barrier for memory operations (the previous level's
// ds_write of the same words must stay ahead of it); lgkmcnt(0) also covers whatever LDS / scalar loads hipcc has in flight.
#ifndef CS2_NL_MULTI_PREISSUE
#define CS2_NL_MULTI_PREISSUE 0   // measured: 2.34 vs 2.28 ms for the ten step sizes - the LDS round trips are not what the kernel waits for
#endif
__device__ __forceinline__ void acc_issue(uint32_t addr, double (&v)[NL_NUM_OUT]) {
    static_assert(NL_NUM_OUT == 10 && kColBlock * 8 == 2048, "offsets below are f * kColBlock * sizeof(double)");
    asm volatile(
        "ds_read_b64 %0, %10\n\t"
        "ds_read_b64 %1, %10 offset:2048\n\t"
        "ds_read_b64 %2, %10 offset:4096\n\t"
        "ds_read_b64 %3, %10 offset:6144\n\t"
        "ds_read_b64 %4, %10 offset:8192\n\t"
        "ds_read_b64 %5, %10 offset:10240\n\t"
        "ds_read_b64 %6, %10 offset:12288\n\t"
        "ds_read_b64 %7, %10 offset:14336\n\t"
        "ds_read_b64 %8, %10 offset:16384\n\t"
        "ds_read_b64 %9, %10 offset:18432"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
          "=&v"(v[9])
        : "v"(addr)
        : "memory");
}
__device__ __forceinline__ void acc_wait(double (&v)[NL_NUM_OUT]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                   "+v"(v[9]));
}

// Kernel arguments as ONE struct (kernarg offset 0): the 42 field pointers are read from the kernarg segment at their point
// of use (KernArgs in cloudsc2_common.hpp) instead of living in - and being spilled from - SGPRs.  This kernel is bound by
// its VALU instructions, and 384 of the 3 996 of a level (NF = 5, fp64) were v_readlane_b32 fetching spilled SGPRs back.
template <typename T, int NF>
struct NLMArgs {
    Ext<T> e;
    NLK<T> kc;
    ExpK<T> xk;
    int nx, nz;
    int64_t ls;
    CPtrs<T, NL_NUM_IN> in, in_i;
    CPtrs<T, NL_NUM_OUT> ref;
    const T* eta;
    T dt;
    PFs<T, NF> pf;
    double* partials;
    int nf_total, f0;
    T finc;
    int zero_supsat_i;
};
#ifndef CS2_NL_MULTI_KARG
#define CS2_NL_MULTI_KARG 1   // 1: field pointers are re-read from the kernarg segment (scalar loads) on every level
#endif

template <typename T, bool EVAP, bool LIN, bool PINK, int NF, bool INC>
__global__ void __launch_bounds__(kColBlock, 1)
nl_taylor_multi_kernel(const NLMArgs<T, NF> A) {
    Ext<T> e = A.e;
    NLK<T> kc = A.kc;
    ExpK<T> xk = A.xk;
    const int nx = A.nx, nz = A.nz, nf_total = A.nf_total, f0 = A.f0, zero_supsat_i = A.zero_supsat_i;
    const int64_t ls = A.ls;
    const T* __restrict__ eta = A.eta;
    T dt = A.dt;
    const PFs<T, NF> pf = A.pf;
    double* __restrict__ partials = A.partials;
    const T finc = A.finc;
    KernArgs<NLMArgs<T, NF>> K;
    const auto P_in = [&](int i) -> const T* { return K->in.p[i]; };
    const auto P_ini = [&](int i) -> const T* { return K->in_i.p[i]; };
    const auto P_ref = [&](int i) -> const T* { return K->ref.p[i]; };
    // the 16 words of a level through a pointer source (nl_load's statement order)
    const auto load16 = [&](const auto& ptr, uint32_t lsb_, uint32_t o_) {
        NLIn<T> x;
        x.ap = ldg(ptr(NL_IN_AP), o_);
        x.aph1 = ldg(ptr(NL_IN_APH), o_ + lsb_);
        x.lu1 = ldg(ptr(NL_IN_LU), o_ + lsb_);
        x.lude = ldg(ptr(NL_IN_LUDE), o_);
        x.mfd = ldg(ptr(NL_IN_MFD), o_);
        x.mfu = ldg(ptr(NL_IN_MFU), o_);
        x.q = ldg(ptr(NL_IN_Q), o_);
        x.qi = ldg(ptr(NL_IN_QI), o_);
        x.ql = ldg(ptr(NL_IN_QL), o_);
        x.qsat = ldg(ptr(NL_IN_QSAT), o_);
        x.supsat = ldg(ptr(NL_IN_SUPSAT), o_);
        x.t = ldg(ptr(NL_IN_T), o_);
        x.tq = ldg(ptr(NL_IN_TND_CML_Q), o_);
        x.tqi = ldg(ptr(NL_IN_TND_CML_QI), o_);
        x.tql = ldg(ptr(NL_IN_TND_CML_QL), o_);
        x.tt = ldg(ptr(NL_IN_TND_CML_T), o_);
        return x;
    };
    const auto load_refs = [&](uint32_t lsb_, uint32_t o_, T (&r)[NL_NUM_OUT]) {
#pragma unroll
        for (int f = 0; f < NL_NUM_OUT; ++f) {
            const bool half = f == NL_OUT_FPLSL || f == NL_OUT_FPLSN || f == NL_OUT_FHPSL || f == NL_OUT_FHPSN;
            r[f] = ldg(P_ref(f), half ? o_ + lsb_ : o_);
        }
    };
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    // running sums: [step size j][output field f][lane], after the level table (16-byte aligned)
    double* const s_acc = reinterpret_cast<double*>(smem_raw + ((2u * uint32_t(nz + 1) * uint32_t(sizeof(T)) + 15u) & ~15u)) +
                          threadIdx.x;
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);
#pragma unroll
    for (int i = 0; i < NF * NL_NUM_OUT; ++i) s_acc[i * kColBlock] = 0.0;   // this lane's own slots: no barrier needed
    if constexpr (PINK && CS2_NL_PINK) {
        pin_vgpr(e.RCPD); pin_vgpr(e.RLSTT); pin_vgpr(e.RLVTT); pin_vgpr(e.R4LES); pin_vgpr(e.R4IES);
        pin_vgpr(e.RTT); pin_vgpr(e.RLPTRC); pin_vgpr(e.R3IES); pin_vgpr(e.R3LES); pin_vgpr(e.R2ES);
        pin_vgpr(e.ZQMAX); pin_vgpr(e.RETV); pin_vgpr(e.R5LES); pin_vgpr(e.R5IES); pin_vgpr(e.RTICE);
        pin_vgpr(e.RG); pin_vgpr(e.RD); pin_vgpr(e.R5ALVCP); pin_vgpr(e.RALVDCP); pin_vgpr(e.R5ALSCP);
        pin_vgpr(e.RALSDCP); pin_vgpr(kc.rdt); pin_vgpr(kc.cons2); pin_vgpr(kc.rRD); pin_vgpr(kc.rRCPD);
        pin_vgpr(kc.cormax); pin_vgpr(kc.fw2); pin_vgpr(dt);
    }
    if constexpr (PINK && CS2_NL_PINX && CS2_NL_FEXP) {
        pin_vgpr(xk.l2e); pin_vgpr(xk.ln2h); pin_vgpr(xk.ln2l); pin_vgpr(xk.c12); pin_vgpr(xk.c11);
        pin_vgpr(xk.c10); pin_vgpr(xk.c9); pin_vgpr(xk.c8); pin_vgpr(xk.c7); pin_vgpr(xk.c6);
        pin_vgpr(xk.c5); pin_vgpr(xk.c4); pin_vgpr(xk.c3);
    }
    const int gcol = xcd_block() * kColBlock + threadIdx.x;
    const bool live = gcol < nx;
    const int col = live ? gcol : nx - 1;   // dead lanes shadow the last column and add nothing
    const uint32_t lsb = uint32_t(ls) * uint32_t(sizeof(T));
    const uint32_t colb = uint32_t(col) * uint32_t(sizeof(T));

    // LDS byte address of this lane's first running sum (the low 32 bits of a generic pointer into LDS are its LDS offset)
    const uint32_t acc_addr = uint32_t(reinterpret_cast<uintptr_t>(s_acc));
    (void)acc_addr;
    T trpaus[NF];
    trpaus_prescan_multi<T, NF, INC>(P_in(NL_IN_T), P_in(NL_IN_TND_CML_T), P_ini(NL_IN_T), P_ini(NL_IN_TND_CML_T), lsb, colb,
                                     dt, s_eta, klo, khi, pf, finc, trpaus);
    CrhCol<T> crh[NF];
    NLCarry<T> c[NF];
    T aph_s[NF];
    {
        const T aph0 = ldg(P_in(NL_IN_APH), colb);
        const T aph0_i = INC ? rounded_product<T>(finc, aph0) : ldg(P_ini(NL_IN_APH), colb);
        T aphs = T(1.0), aphs_i = T(0.0);
        if constexpr (EVAP) {
            aphs = ldg(P_in(NL_IN_APH), uint32_t(nz) * lsb + colb);
            aphs_i = INC ? rounded_product<T>(finc, aphs) : ldg(P_ini(NL_IN_APH), uint32_t(nz) * lsb + colb);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            crh[j] = crh_setup<T>(trpaus[j]);
            c[j].rfl = T(0.0);
            c[j].sfl = T(0.0);
            c[j].covptot = T(0.0);
            c[j].aph_k = aph0 + pf.f[j] * aph0_i;
            aph_s[j] = EVAP ? aphs + pf.f[j] * aphs_i : T(1.0);
        }
    }

    // level k+1's 26-42 words are requested before level k's NF evaluations and handed over by register copies at the top of
    // the next level (`cur = next`): the copy is the only consumer of the prefetched words, so they have a whole level of
    // arithmetic (~5 us) to arrive.  (A two-way unrolled double buffer without copies - the shape nl_kernel uses - made
    // hipcc consume ten of the 26 words 76 instructions after requesting them: VALU-active 69 %.)
#ifndef CS2_NL_MULTI_COPYBUF
#define CS2_NL_MULTI_COPYBUF 1
#endif
    NLIn<T> xa = load16(P_in, lsb, colb), xb;
    if constexpr (!INC) xb = load16(P_ini, lsb, colb);
    T xr[NL_NUM_OUT];
    load_refs(lsb, colb, xr);
    uint32_t o = colb;
    for (int k = 0; k < nz; ++k) {
        // fp64 only: -4 ... -5 % there (2.24 -> 2.12 ms for the ten step sizes at 65 536 columns; bit-identical sums); in fp32 at
        // 524 288 columns the kernel waits for HBM and the per-level scalar loads measured 0 ... +2 % (profiles/r04/ab_multi_*)
        K.template fresh<(CS2_NL_MULTI_KARG != 0 && sizeof(T) == 8)>();
        NLIn<T> na = xa, nb = xb;
        T nr[NL_NUM_OUT];
#pragma unroll
        for (int f = 0; f < NL_NUM_OUT; ++f) nr[f] = xr[f];
        if (k + 1 < nz) {
            na = load16(P_in, lsb, o + lsb);
            if constexpr (!INC) nb = load16(P_ini, lsb, o + lsb);
            load_refs(lsb, o + lsb, nr);
        }
        const T eta_k = s_eta[k], scalm_k = s_scalm[k];
        NLIn<T> inc_k;
        if constexpr (INC) inc_k = nl_increment<T>(xa, finc, zero_supsat_i != 0);
        const NLIn<T>& xi = INC ? inc_k : xb;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            double* const a = s_acc + j * NL_NUM_OUT * kColBlock;
            double sum[NL_NUM_OUT];
#if CS2_NL_MULTI_PREISSUE
            acc_issue(acc_addr + uint32_t(j * NL_NUM_OUT * kColBlock * sizeof(double)), sum);
#endif
            const NLIn<T> x = nl_perturb<T>(xa, xi, pf.f[j]);
            const NLOut<T> r = nl_level<T, EVAP, LIN>(e, kc, xk, x, eta_k, scalm_k, crh[j], dt, aph_s[j], c[j]);
            double d[NL_NUM_OUT];
            d[NL_OUT_CLC] = double(r.clc - xr[NL_OUT_CLC]);
            d[NL_OUT_COVPTOT] = double(r.covptot - xr[NL_OUT_COVPTOT]);
            d[NL_OUT_TND_Q] = double(r.tnd_q - xr[NL_OUT_TND_Q]);
            d[NL_OUT_TND_T] = double(r.tnd_t - xr[NL_OUT_TND_T]);
            d[NL_OUT_TND_QL] = double(r.tnd_ql - xr[NL_OUT_TND_QL]);
            d[NL_OUT_TND_QI] = double(r.tnd_qi - xr[NL_OUT_TND_QI]);
            d[NL_OUT_FPLSL] = double(r.rfln - xr[NL_OUT_FPLSL]);
            d[NL_OUT_FPLSN] = double(r.sfln - xr[NL_OUT_FPLSN]);
            d[NL_OUT_FHPSL] = double(enthalpy_diff<T>(r.rfln, e.RLVTT, xr[NL_OUT_FHPSL]));
            d[NL_OUT_FHPSN] = double(enthalpy_diff<T>(r.sfln, e.RLSTT, xr[NL_OUT_FHPSN]));
#if CS2_NL_MULTI_PREISSUE
            acc_wait(sum);
#else
#pragma unroll
            for (int f = 0; f < NL_NUM_OUT; ++f) sum[f] = a[f * kColBlock];
#endif
            if (live) {
#pragma unroll
                for (int f = 0; f < NL_NUM_OUT; ++f) a[f * kColBlock] = sum[f] + d[f];
            }
        }
        xa = na;
        if constexpr (!INC) xb = nb;
#pragma unroll
        for (int f = 0; f < NL_NUM_OUT; ++f) xr[f] = nr[f];
        o += lsb;
    }
    // workgroup reduction of the NF x 10 sums: wave shuffle, then one LDS hop
    __shared__ double s_red[kColBlock / 64][NF * NL_NUM_OUT];
#pragma unroll
    for (int i = 0; i < NF * NL_NUM_OUT; ++i) {
        double v = s_acc[i * kColBlock];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NF * NL_NUM_OUT) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kColBlock / 64; ++w) v += s_red[w][threadIdx.x];
        const int j = threadIdx.x / NL_NUM_OUT, f = threadIdx.x % NL_NUM_OUT;
        partials[(size_t(blockIdx.x) * size_t(nf_total) + size_t(f0 + j)) * NL_NUM_OUT + f] = v;
    }
}

#ifndef CS2_NL_MULTI_NF
#define CS2_NL_MULTI_NF 5   // step sizes per launch: 10 x NF fp64 running sums x 256 lanes must fit the CU's LDS
#endif
template <typename T>
int launch_nl_taylor_multi(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* const* in, const T* const* in_i,
                           int nf, const double* pfs, const T* eta, const T* const* ref_out, double* partials, double dt,
                           hipStream_t stream, double inc_f) {
    const bool inc = in_i == nullptr;      // fused state_increment: in_i = T(inc_f) * in, formed in the kernel
    const T tinc = static_cast<T>(inc_f);
    const int zsi = p.IGNORE_SUPSAT ? 1 : 0;
    const Ext<T> e = make_ext<T>(p);
    CPtrs<T, NL_NUM_IN> ci, cii;
    CPtrs<T, NL_NUM_OUT> cr;
    for (int i = 0; i < NL_NUM_IN; ++i) { ci.p[i] = in[i]; cii.p[i] = inc ? nullptr : in_i[i]; }
    for (int i = 0; i < NL_NUM_OUT; ++i) cr.p[i] = ref_out[i];
    const dim3 grid((nx + kColBlock - 1) / kColBlock), block(kColBlock);
    const bool evap = p.LEVAPLS2 || p.LDRAIN1D;
    const bool lin = p.LPHYLIN || p.LDRAIN1D;
    const T tdt = static_cast<T>(dt);
    const NLK<T> kc = make_nlk<T>(p, dt, evap);
    const ExpK<T> xk = make_expk<T>();
    if (!fits_u32_offsets<T>(nz, ls)) return -2;
    int dev = 0;
    if (const int rc = current_device(dev)) return rc;
    const size_t tab = (2 * size_t(nz + 1) * sizeof(T) + 15) & ~size_t(15);
#define CS2_NLM_LAUNCH(EV, LN, NFV)                                                                                    \
    do {                                                                                                                \
        if (inc) CS2_NLM_LAUNCH_I(EV, LN, NFV, true); else CS2_NLM_LAUNCH_I(EV, LN, NFV, false);                        \
    } while (0)
#define CS2_NLM_LAUNCH_I(EV, LN, NFV, INCV)                                                                            \
    do {                                                                                                                \
        auto kern = nl_taylor_multi_kernel<T, EV, LN, sizeof(T) == 8, NFV, INCV>;                                       \
        const size_t smem = tab + size_t(NFV) * NL_NUM_OUT * kColBlock * sizeof(double);                                \
        if (smem + sizeof(double) * (kColBlock / 64) * NFV * NL_NUM_OUT > size_t(160) * 1024) return -2;                \
        static std::atomic<size_t> attr_set[kMaxDevices] = {};                                                          \
        if (smem > size_t(64) * 1024 && !lds_opt_in(kern, attr_set, dev, smem)) return -1;                              \
        PFs<T, NFV> pf;                                                                                                 \
        for (int j = 0; j < NFV; ++j) pf.f[j] = static_cast<T>(pfs[f0 + j]);                                            \
        const NLMArgs<T, NFV> margs = {e, kc, xk, nx, nz, ls, ci, cii, cr, eta, tdt, pf, partials, nf, f0, tinc, zsi};  \
        hipLaunchKernelGGL(kern, grid, block, smem, stream, margs);                                                     \
    } while (0)
#define CS2_NLM_FLAGS(NFV)                                            \
    do {                                                              \
        if (evap && lin) CS2_NLM_LAUNCH(true, true, NFV);             \
        else if (evap && !lin) CS2_NLM_LAUNCH(true, false, NFV);      \
        else if (!evap && lin) CS2_NLM_LAUNCH(false, true, NFV);      \
        else CS2_NLM_LAUNCH(false, false, NFV);                       \
    } while (0)
    for (int f0 = 0; f0 < nf;) {
        const int left = nf - f0;
        if (left >= CS2_NL_MULTI_NF) { CS2_NLM_FLAGS(CS2_NL_MULTI_NF); f0 += CS2_NL_MULTI_NF; }
        else if (left >= 3 && CS2_NL_MULTI_NF > 3) { CS2_NLM_FLAGS(3); f0 += 3; }
        else if (left == 2) { CS2_NLM_FLAGS(2); f0 += 2; }
        else { CS2_NLM_FLAGS(1); f0 += 1; }
    }
#undef CS2_NLM_FLAGS
#undef CS2_NLM_LAUNCH
#undef CS2_NLM_LAUNCH_I
    note_kernel("cs2::nl_taylor_multi_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template int launch_nl_taylor_multi<double>(const Cloudsc2Params&, int, int, int64_t, const double* const*,
                                            const double* const*, int, const double*, const double*, const double* const*,
                                            double*, double, hipStream_t, double);
template int launch_nl_taylor_multi<float>(const Cloudsc2Params&, int, int, int64_t, const float* const*,
                                           const float* const*, int, const double*, const float*, const float* const*,
                                           double*, double, hipStream_t, double);

}  // namespace cs2
