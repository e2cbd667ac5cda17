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

// One level of the forward sweep (:113-388) for one column.
template <typename T, bool EVAP, bool LIN>
__device__ __forceinline__ NLOut<T> nl_level(const Ext<T>& e, const NLIn<T>& x, T eta_k, T scalm,
                                             const CrhCol<T>& crh, T dt, T aph_s, NLCarry<T>& c) {
    NLOut<T> o;
    // :104, :115-117 first guess
    T t = x.t + dt * x.tt;
    T q = x.q + dt * x.tq + x.supsat;
    const T ql = x.ql + dt * x.tql;
    const T qi = x.qi + dt * x.tqi;
    // :120-124
    const T ckcodtl = T(2.0) * e.RKCONV * dt;
    const T ckcodti = T(5.0) * e.RKCONV * dt;
    const T cons2 = T(1.0) / (e.RG * dt);
    const T cons3 = e.RLVTT / e.RCPD;
    const T meltp2 = e.RTT + T(2.0);
    // :130-134
    const T dp = x.aph1 - c.aph_k;
    const T zz = e.RCPD + e.RCPD * e.RVTMP2 * q;
    const T lfdcp = e.RLMLT / zz;
    const T lsdcp = e.RLSTT / zz;
    const T lvdcp = e.RLVTT / zz;
    // :141-160 dqs/dT correction factor
    T fwat, foeew, esdp;
    if constexpr (LIN) {
        T z3es, z4es;
        if (t < e.RTT) {
            fwat = T(0.545) * (rtanh<T>(T(0.17) * (t - e.RLPTRC)) + T(1.0));
            z3es = e.R3IES;
            z4es = e.R4IES;
        } else {
            fwat = T(1.0);
            z3es = e.R3LES;
            z4es = e.R4LES;
        }
        foeew = e.R2ES * rexp<T>(z3es * (t - e.RTT) / (t - z4es));
        esdp = rmin<T>(foeew / x.ap, e.ZQMAX);
    } else {
        // f_foealfa / f_foeewm, common/_stencils/fcttre.py:22-46
        fwat = rmin<T>(T(1.0), sq((rmax<T>(e.RTICE, rmin<T>(e.RTWAT, t)) - e.RTICE) * e.RTWAT_RTICE_R));
        foeew = e.R2ES * (fwat * rexp<T>(e.R3LES * (t - e.RTT) / (t - e.R4LES)) +
                          (T(1.0) - fwat) * rexp<T>(e.R3IES * (t - e.RTT) / (t - e.R4IES)));
        esdp = foeew / x.ap;
    }
    const T facw = e.R5LES / sq(t - e.R4LES);
    const T faci = e.R5IES / sq(t - e.R4IES);
    const T fac = fwat * facw + (T(1.0) - fwat) * faci;
    const T dqsdtemp = fac * x.qsat / (T(1.0) - e.RETV * esdp);
    // :163
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
        clc = T(1.0) - rsqrt_<T>(qpd / (qcd - scalm * (qt - qcrit)));
        qc = (scalm * qpd + (T(1.0) - scalm) * qcd) * sq(clc);
    }
    // :210-215 convective detrainment
    const T gdp = e.RG / (x.aph1 - c.aph_k);
    const T lude = dt * x.lude * gdp;
    if (lude >= e.RLMIN && x.lu1 >= e.ZEPS2) {
        clc += (T(1.0) - clc) * (T(1.0) - rexp<T>(-lude / x.lu1));
        qc += lude;
    }
    // :218-224 compensating subsidence
    const T rho = x.ap / (e.RD * t);
    const T rodqsdp = -rho * x.qsat / (x.ap - e.RETV * foeew);
    const T ldcp = fwat * lvdcp + (T(1.0) - fwat) * lsdcp;
    const T dtdzmo = e.RG * (T(1.0) / e.RCPD - ldcp * rodqsdp) / (T(1.0) + ldcp * dqsdtemp);
    const T dqsdz = dqsdtemp * dtdzmo - e.RG * rodqsdp;
    const T dqc = rmin<T>(dt * dqsdz * (x.mfu + x.mfd) / rho, qc);
    qc -= dqc;
    // :227-230
    T qlwc = qc * fwat;
    T qiwc = qc * (T(1.0) - fwat);
    T condl = (qlwc - ql) / dt;
    T condi = (qiwc - qi) / dt;
    // :234-235 maximum overlap
    c.covptot = rmax<T>(c.covptot, clc);
    const T covpclr = rmax<T>(c.covptot - clc, T(0.0));
    // :238-246 melting of incoming snow
    T rfln, sfln;
    if (c.sfl != T(0.0)) {
        const T cons = cons2 * dp / lfdcp;
        const T snmlt = rmin<T>(c.sfl, cons * rmax<T>(t - meltp2, T(0.0)));
        rfln = c.rfl + snmlt;
        sfln = c.sfl - snmlt;
        t -= snmlt / cons;
    } else {
        rfln = c.rfl;
        sfln = c.sfl;
    }
    // :249-272 autoconversion
    T prr = T(0.0), prs = T(0.0);
    if (clc > e.ZEPS2) {
        const T lcrit = EVAP ? T(1.9) * e.RCLCRIT : T(2.0) * e.RCLCRIT;
        const T cldl = qlwc / clc;
        const T dl = ckcodtl * (T(1.0) - rexp<T>(-sq(cldl / lcrit)));
        prr = qlwc - clc * cldl * rexp<T>(-dl);
        qlwc -= prr;
        const T icrit = EVAP ? T(0.0001) : T(2.0) * e.RCLCRIT;
        const T cldi = qiwc / clc;
        const T di = ckcodti * rexp<T>(T(0.025) * (t - e.RTT)) * (T(1.0) - rexp<T>(-sq(cldi / icrit)));
        prs = qiwc - clc * cldi * rexp<T>(-di);
        qiwc -= prs;
    }
    // :275-285 new precipitation
    const T dr = cons2 * dp * (prr + prs);
    T rfreeze, fwatr;
    if (t < e.RTT) {
        rfreeze = cons2 * dp * prr;
        fwatr = T(0.0);
    } else {
        rfreeze = T(0.0);
        fwatr = T(1.0);
    }
    rfln += fwatr * dr;
    sfln += (T(1.0) - fwatr) * dr;
    // :288-321 precipitation evaporation
    T evapr = T(0.0), evaps = T(0.0);
    o.covptot = T(0.0);
    if constexpr (EVAP) {
        const T prtot = rfln + sfln;
        if (prtot > e.ZEPS2 && covpclr > e.ZEPS2) {
            const T corqs = T(1.0) + cons3 * dqsdtemp;  // :160
            const T qlim = rmin<T>(q, x.qsat);          // :163
            T preclr = prtot * covpclr / c.covptot;
            const T qe = x.qsat - (x.qsat - qlim) * covpclr / sq(T(1.0) - clc);
            const T beta = e.RG * e.RPECONS *
                           rpow<T>(rsqrt_<T>(x.ap / aph_s) / T(0.00509) * preclr / covpclr, T(0.5777));
            const T b = dt * beta * (x.qsat - qe) / (T(1.0) + dt * beta * corqs);
            const T dtgdp = dt * e.RG / (x.aph1 - c.aph_k);
            const T dpr = rmin<T>(covpclr * b / dtgdp, preclr);
            preclr -= dpr;
            if (preclr <= T(0.0)) c.covptot = clc;
            o.covptot = c.covptot;
            evapr = dpr * rfln / prtot;
            rfln -= evapr;
            evaps = dpr * sfln / prtot;
            sfln -= evaps;
        }
    }
    // :328-344 first-guess T and q after cloud processes
    const T dqdt = -(condl + condi) + (x.lude + evapr + evaps) * gdp;
    const T dtdt = lvdcp * condl + lsdcp * condi -
                   (lvdcp * evapr + lsdcp * evaps + x.lude * (fwat * lvdcp + (T(1.0) - fwat) * lsdcp) -
                    (lsdcp - lvdcp) * rfreeze) * gdp;
    t += dt * dtdt;
    q += dt * dqdt;
    const T qold = q;
    // :347 saturation adjustment
    cuadjtqs_nl(e, x.ap, t, q);
    // :350-364
    const T dq = rmax<T>(qold - q, T(0.0));
    const T dr2 = cons2 * dp * dq;
    T rfreeze2;
    if (t < e.RTT) {
        rfreeze2 = fwat * dr2;
        fwatr = T(0.0);
    } else {
        rfreeze2 = T(0.0);
        fwatr = T(1.0);
    }
    const T rn = fwatr * dr2;
    const T sn = (T(1.0) - fwatr) * dr2;
    condl += fwatr * dq / dt;
    condi += (T(1.0) - fwatr) * dq / dt;
    rfln += rn;
    sfln += sn;
    rfreeze += rfreeze2;
    // :367-380 output tendencies
    o.clc = clc;
    o.tnd_q = -(condl + condi) + (x.lude + evapr + evaps) * gdp;
    o.tnd_t = lvdcp * condl + lsdcp * condi -
              (lvdcp * evapr + lsdcp * evaps + x.lude * (fwat * lvdcp + (T(1.0) - fwat) * lsdcp) -
               (lsdcp - lvdcp) * rfreeze) * gdp;
    o.tnd_ql = (qlwc - ql) / dt;
    o.tnd_qi = (qiwc - qi) / dt;
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

template <typename T, bool EVAP, bool LIN>
__global__ void __launch_bounds__(kWave)
nl_kernel(Ext<T> e, int nx, int nz, int64_t ls, CPtrs<T, NL_NUM_IN> in, const T* __restrict__ eta,
          MPtrs<T, NL_NUM_OUT> out, T dt) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T* s_eta = reinterpret_cast<T*>(smem_raw);
    T* s_scalm = s_eta + (nz + 1);
    int klo, khi;
    build_level_table<T>(eta, nz, e, s_eta, s_scalm, klo, khi);

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
            const NLOut<T> o = nl_level<T, EVAP, LIN>(e, a, s_eta[k], s_scalm[k], crh, dt, aph_s, c);
            if (live) nl_store<T>(out, e, ls, col, k, o);
        }
        if (has_b) {
            if (k + 2 < nz) a = nl_load<T>(in, ls, col, k + 2);
            const NLOut<T> o = nl_level<T, EVAP, LIN>(e, b, s_eta[k + 1], s_scalm[k + 1], crh, dt, aph_s, c);
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
#define CS2_NL_LAUNCH(EV, LN) \
    hipLaunchKernelGGL((nl_kernel<T, EV, LN>), grid, block, smem, stream, e, nx, nz, ls, ci, eta, co, tdt)
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
