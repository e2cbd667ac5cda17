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
// Template flags: EVAP = LEVAPLS2 or LDRAIN1D (precipitation evaporation block :288-321 and the
// 1.9*RCLCRIT / 1e-4 thresholds :250-266), LIN = LPHYLIN or LDRAIN1D (:141-155).
#include "cloudsc2_common.hpp"

namespace cs2 {

template <typename T>
struct NLIn {
    T ap, aph1, lu1, lude, mfd, mfu, q, qi, ql, qsat, supsat, t, tq, tqi, tql, tt;
};

template <typename T>
__device__ __forceinline__ NLIn<T> nl_load(const CPtrs<T, NL_NUM_IN>& in, int64_t ls, int col, int k) {
    const int64_t o = int64_t(k) * ls + col;
    NLIn<T> x;
    x.ap = in.p[NL_IN_AP][o];
    x.aph1 = in.p[NL_IN_APH][o + ls];
    x.lu1 = in.p[NL_IN_LU][o + ls];
    x.lude = in.p[NL_IN_LUDE][o];
    x.mfd = in.p[NL_IN_MFD][o];
    x.mfu = in.p[NL_IN_MFU][o];
    x.q = in.p[NL_IN_Q][o];
    x.qi = in.p[NL_IN_QI][o];
    x.ql = in.p[NL_IN_QL][o];
    x.qsat = in.p[NL_IN_QSAT][o];
    x.supsat = in.p[NL_IN_SUPSAT][o];
    x.t = in.p[NL_IN_T][o];
    x.tq = in.p[NL_IN_TND_CML_Q][o];
    x.tqi = in.p[NL_IN_TND_CML_QI][o];
    x.tql = in.p[NL_IN_TND_CML_QL][o];
    x.tt = in.p[NL_IN_TND_CML_T][o];
    return x;
}

template <typename T>
struct NLCarry {
    T rfl, sfl, covptot, aph_k;
};

template <typename T>
struct NLOut {
    T clc, covptot, tnd_q, tnd_t, tnd_ql, tnd_qi, rfln, sfln;
};

// Level-independent derived constants (one set per launch, computed on the host in double).
template <typename T>
struct NLK {
    T rdt, ckcodtl, ckcodti, cons2, rgdt, cons3, meltp2, rlcrit, ricrit, rRD, rRCPD, rRLMLT, cormax, fw2;
};

template <typename T>
inline NLK<T> make_nlk(const Cloudsc2Params& p, double dt, bool evap) {
    NLK<T> k;
    k.rdt = T(1.0 / dt);
    k.ckcodtl = T(2.0 * p.RKCONV * dt);              // :120
    k.ckcodti = T(5.0 * p.RKCONV * dt);              // :121
    k.cons2 = T(1.0 / (p.RG * dt));                  // :122
    k.rgdt = T(p.RG * dt);
    k.cons3 = T(p.RLVTT / p.RCPD);                   // :123
    k.meltp2 = T(p.RTT + 2.0);                       // :124
    k.rlcrit = T(1.0 / ((evap ? 1.9 : 2.0) * p.RCLCRIT));     // :250-253
    k.ricrit = T(1.0 / (evap ? 0.0001 : 2.0 * p.RCLCRIT));    // :263-266
    k.rRD = T(1.0 / p.RD);
    k.rRCPD = T(1.0 / p.RCPD);
    k.rRLMLT = T(1.0 / p.RLMLT);
    k.cormax = T(1.0 / (1.0 - p.RETV * p.ZQMAX));    // 1 / (1 - RETV * esdp) when esdp is clipped at ZQMAX
    k.fw2 = T(2.0 * 0.17);                           // tanh(u) + 1 = 2 / (1 + exp(-2u)), u = 0.17 (t - RLPTRC)
    return k;
}

// One iteration of the saturation adjustment (nonlinear/_stencils/cuadjtqs.py:24-37) with shared
// reciprocals: r = 1/(t - z4es) serves the exponent and z2s, rap = 1/ap is the caller's.
template <typename T>
__device__ __forceinline__ void nl_cuadj_iter(const Ext<T>& e, T rap, T& t, T& q, T z3es, T z4es, T z5alcp,
                                              T zaldcp) {
    const T r = frcp<T>(t - z4es);
    const T foeew = e.R2ES * rexp<T>(z3es * (t - e.RTT) * r);
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
__device__ __forceinline__ NLOut<T> nl_level(const Ext<T>& e, const NLK<T>& kc, const NLIn<T>& x, T eta_k, T scalm,
                                             const CrhCol<T>& crh, T dt, T aph_s, NLCarry<T>& c) {
    NLOut<T> o;
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
            fwat = T(1.09) * frcp<T>(T(1.0) + rexp<T>(-kc.fw2 * (t - e.RLPTRC)));
            z3es = e.R3IES;
            r4 = ri;
        } else {
            fwat = T(1.0);
            z3es = e.R3LES;
            r4 = rl;
        }
        foeew = e.R2ES * rexp<T>(z3es * (t - e.RTT) * r4);
        const T esdp = foeew * rap;
        cor = (esdp > e.ZQMAX) ? kc.cormax : frcp<T>(T(1.0) - e.RETV * esdp);
    } else {
        // f_foealfa / f_foeewm, common/_stencils/fcttre.py:22-46
        fwat = rmin<T>(T(1.0), sq((rmax<T>(e.RTICE, rmin<T>(e.RTWAT, t)) - e.RTICE) * e.RTWAT_RTICE_R));
        foeew = e.R2ES * (fwat * rexp<T>(e.R3LES * (t - e.RTT) * rl) +
                          (T(1.0) - fwat) * rexp<T>(e.R3IES * (t - e.RTT) * ri));
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
        clc += (T(1.0) - clc) * (T(1.0) - rexp<T>(-lude * frcp<T>(x.lu1)));
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
        const T dl = kc.ckcodtl * (T(1.0) - rexp<T>(-sq(cldl * kc.rlcrit)));
        prr = qlwc - clc * cldl * rexp<T>(-dl);
        qlwc -= prr;
        const T cldi = qiwc * rclc;
        const T di = kc.ckcodti * rexp<T>(T(0.025) * (t - e.RTT)) * (T(1.0) - rexp<T>(-sq(cldi * kc.ricrit)));
        prs = qiwc - clc * cldi * rexp<T>(-di);
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
            const T rprtot = frcp<T>(prtot);
            evapr = dpr * rfln * rprtot;
            rfln -= evapr;
            evaps = dpr * sfln * rprtot;
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
        nl_cuadj_iter(e, rap, t, q, z3es, z4es, z5alcp, zaldcp);
        nl_cuadj_iter(e, rap, t, q, z3es, z4es, z5alcp, zaldcp);
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

template <typename T>
__device__ __forceinline__ void nl_store(const MPtrs<T, NL_NUM_OUT>& out, const Ext<T>& e, int64_t ls,
                                         int col, int k, const NLOut<T>& o) {
    const int64_t i = int64_t(k) * ls + col;
    out.p[NL_OUT_CLC][i] = o.clc;
    out.p[NL_OUT_COVPTOT][i] = o.covptot;
    out.p[NL_OUT_TND_Q][i] = o.tnd_q;
    out.p[NL_OUT_TND_T][i] = o.tnd_t;
    out.p[NL_OUT_TND_QL][i] = o.tnd_ql;
    out.p[NL_OUT_TND_QI][i] = o.tnd_qi;
    // :391-399 fluxes leave level k through half level k+1
    out.p[NL_OUT_FPLSL][i + ls] = o.rfln;
    out.p[NL_OUT_FPLSN][i + ls] = o.sfln;
    out.p[NL_OUT_FHPSL][i + ls] = -o.rfln * e.RLVTT;
    out.p[NL_OUT_FHPSN][i + ls] = -o.sfln * e.RLSTT;
}

// Tropopause pre-scan (:107-111): eta of the LAST level k in the window with t[k] > t[k+1].
template <typename T>
__device__ __forceinline__ T nl_trpaus(const T* __restrict__ pt, const T* __restrict__ ptt, int64_t ls,
                                       int col, T dt, const T* s_eta, int klo, int khi) {
    T trpaus = T(0.1);
    if (klo <= khi) {
        T tk = pt[int64_t(klo) * ls + col] + dt * ptt[int64_t(klo) * ls + col];
        for (int k = klo; k <= khi; ++k) {
            const int64_t o1 = int64_t(k + 1) * ls + col;
            const T tk1 = pt[o1] + dt * ptt[o1];
            const T ek = s_eta[k];
            if (ek > T(0.1) && ek < T(0.4) && tk > tk1) trpaus = ek;
            tk = tk1;
        }
    }
    return trpaus;
}

template <typename T, bool EVAP, bool LIN, bool PINK>
__global__ void __launch_bounds__(kWave)
nl_kernel(Ext<T> e, NLK<T> kc, int nx, int nz, int64_t ls, CPtrs<T, NL_NUM_IN> in, const T* __restrict__ eta,
          MPtrs<T, NL_NUM_OUT> out, T dt) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);
    if constexpr (PINK) {
        // constants of the level loop -> VGPRs (see pin_vgpr)
        pin_vgpr(e.RCPD); pin_vgpr(e.RLSTT); pin_vgpr(e.RLVTT); pin_vgpr(e.R4LES); pin_vgpr(e.R4IES);
        pin_vgpr(e.RTT); pin_vgpr(e.RLPTRC); pin_vgpr(e.R3IES); pin_vgpr(e.R3LES); pin_vgpr(e.R2ES);
        pin_vgpr(e.ZQMAX); pin_vgpr(e.RETV); pin_vgpr(e.R5LES); pin_vgpr(e.R5IES); pin_vgpr(e.RTICE);
        pin_vgpr(e.RG); pin_vgpr(e.RD); pin_vgpr(e.R5ALVCP); pin_vgpr(e.RALVDCP); pin_vgpr(e.R5ALSCP);
        pin_vgpr(e.RALSDCP); pin_vgpr(kc.rdt); pin_vgpr(kc.cons2); pin_vgpr(kc.rRD); pin_vgpr(kc.rRCPD);
        pin_vgpr(kc.cormax); pin_vgpr(kc.fw2); pin_vgpr(dt);
    }

    const int gcol = blockIdx.x * kWave + threadIdx.x;
    const bool live = gcol < nx;
    const int col = live ? gcol : nx - 1;  // dead lanes shadow the last column, stores masked

    const T trpaus = nl_trpaus<T>(in.p[NL_IN_T], in.p[NL_IN_TND_CML_T], ls, col, dt, s_eta, klo, khi);
    const CrhCol<T> crh = crh_setup<T>(trpaus);

    // :93-100
    NLCarry<T> c;
    c.rfl = T(0.0);
    c.sfl = T(0.0);
    c.covptot = T(0.0);
    c.aph_k = in.p[NL_IN_APH][col];
    const T aph_s = EVAP ? in.p[NL_IN_APH][int64_t(nz) * ls + col] : T(1.0);

    if (live) {
        // top half level: no flux enters the column (:392-394; out_fpls*[0] written as 0, the
        // value the reference relies on from zero-initialised storage - SURVEY.md App. B Q2)
        out.p[NL_OUT_FPLSL][col] = T(0.0);
        out.p[NL_OUT_FPLSN][col] = T(0.0);
        out.p[NL_OUT_FHPSL][col] = T(0.0);
        out.p[NL_OUT_FHPSN][col] = T(0.0);
    }

    NLIn<T> a = nl_load<T>(in, ls, col, 0);
    NLIn<T> b = a;
    for (int k = 0; k < nz; k += 2) {
        const bool has_b = (k + 1 < nz);
        if (has_b) b = nl_load<T>(in, ls, col, k + 1);
        {
            const NLOut<T> o = nl_level<T, EVAP, LIN>(e, kc, a, s_eta[k], s_scalm[k], crh, dt, aph_s, c);
            if (live) nl_store<T>(out, e, ls, col, k, o);
        }
        if (has_b) {
            if (k + 2 < nz) a = nl_load<T>(in, ls, col, k + 2);
            const NLOut<T> o = nl_level<T, EVAP, LIN>(e, kc, b, s_eta[k + 1], s_scalm[k + 1], crh, dt, aph_s, c);
            if (live) nl_store<T>(out, e, ls, col, k + 1, o);
        }
    }
}

template <typename T>
int launch_nl(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* const* in, const T* eta,
              T* const* out, double dt, hipStream_t stream) {
    const Ext<T> e = make_ext<T>(p);
    CPtrs<T, NL_NUM_IN> ci;
    MPtrs<T, NL_NUM_OUT> co;
    for (int i = 0; i < NL_NUM_IN; ++i) ci.p[i] = in[i];
    for (int i = 0; i < NL_NUM_OUT; ++i) co.p[i] = out[i];
    const dim3 grid((nx + kWave - 1) / kWave), block(kWave);
    const size_t smem = 2 * size_t(nz + 1) * sizeof(T);
    const bool evap = p.LEVAPLS2 || p.LDRAIN1D;
    const bool lin = p.LPHYLIN || p.LDRAIN1D;
    const T tdt = static_cast<T>(dt);
    const NLK<T> kc = make_nlk<T>(p, dt, evap);
#define CS2_NL_LAUNCH(EV, LN) \
    hipLaunchKernelGGL((nl_kernel<T, EV, LN, sizeof(T) == 8>), grid, block, smem, stream, e, kc, nx, nz, ls, ci, eta, co, tdt)
    if (evap && lin) CS2_NL_LAUNCH(true, true);
    else if (evap && !lin) CS2_NL_LAUNCH(true, false);
    else if (!evap && lin) CS2_NL_LAUNCH(false, true);
    else CS2_NL_LAUNCH(false, false);
#undef CS2_NL_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template int launch_nl<double>(const Cloudsc2Params&, int, int, int64_t, const double* const*, const double*,
                               double* const*, double, hipStream_t);
template int launch_nl<float>(const Cloudsc2Params&, int, int, int64_t, const float* const*, const float*,
                              float* const*, double, hipStream_t);

}  // namespace cs2
