/*
 * ORACLE / CPU BASELINE (test infrastructure, NOT product code): plain-C + OpenMP restatement ("restatement B",
 * SURVEY.md 8d) of the two stencils of the reference driver's timed region, fp64:
 *
 *   saturation    /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/saturation.py:23-42
 *                 (+ f_foealfa / f_foeewm / f_foeewmcu, common/_stencils/fcttre.py:22-57)
 *   cloudsc2_nl   /root/reference/src/cloudsc2_gt4py/physics/nonlinear/_stencils/cloudsc2.py:93-399
 *   f_cuadjtqs_nl /root/reference/src/cloudsc2_gt4py/physics/nonlinear/_stencils/cuadjtqs.py:22-68
 *
 * Only tests/, __graft_entry__.{build,smoke}() and the cpu_baseline leg of bench.py may build, load or call it.
 * It is pinned through oracle/cloudsc2_numpy.py (itself bit-identical to the executed reference source,
 * tests/test_reference_exec.py): tests/test_oracle_c.py requires agreement to ~1 ulp-level tolerances on
 * all externals combinations.  Same data layout as the product ([level][column], nz+1 levels, include/
 * cloudsc2_hip.h) so that the same host arrays feed both.
 *
 * Execution shape: OpenMP threads over blocks of CB adjacent columns; inside a block the level loop is
 * outermost and the carried precipitation state lives in small per-block arrays - i.e. what a hand-written
 * CPU port of the scheme would look like, as opposed to the statement-at-a-time NumPy restatement.
 *
 * Build: gcc -O3 -march=native -fopenmp -fPIC -shared oracle/cloudsc2_nl_omp.c -Iinclude -lm -o oracle/_build/libcloudsc2_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "cloudsc2_hip.h"

#define CB 16 /* columns per block: 16 doubles = two cache lines per field row */

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double sq(double a) { return a * a; }

int cs2c_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* fcttre.py:22-27 / :30-35 */
static inline double foealfa(const Cloudsc2Params* e, double t) {
    return dmin(1.0, sq((dmax(e->RTICE, dmin(e->RTWAT, t)) - e->RTICE) * e->RTWAT_RTICE_R));
}
static inline double foealfcu(const Cloudsc2Params* e, double t) {
    return dmin(1.0, sq((dmax(e->RTICECU, dmin(e->RTWAT, t)) - e->RTICECU) * e->RTWAT_RTICECU_R));
}
/* fcttre.py:38-46 / :49-57 */
static inline double foeewm_with(const Cloudsc2Params* e, double t, double alfa) {
    return e->R2ES * (alfa * exp(e->R3LES * (t - e->RTT) / (t - e->R4LES)) +
                      (1.0 - alfa) * exp(e->R3IES * (t - e->RTT) / (t - e->R4IES)));
}

/* saturation.py:23-42 on levels 0 .. nz-1 */
int cs2c_saturation(const Cloudsc2Params* e, int nx, int nz, int64_t ls, const double* ap, const double* t,
                    double* qsat, int nthreads) {
    if (!e || !ap || !t || !qsat || nx < 1 || nz < 1 || ls < nx) return -1;
    if (nthreads < 1) nthreads = cs2c_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int k = 0; k < nz; ++k) {
        for (int c = 0; c < nx; ++c) {
            const int64_t i = (int64_t)k * ls + c;
            const double tt = t[i];
            double qs;
            if (e->LPHYLIN) {
                const double alfa = foealfa(e, tt);
                const double foeewl = e->R2ES * exp(e->R3LES * (tt - e->RTT) / (tt - e->R4LES));
                const double foeewi = e->R2ES * exp(e->R3IES * (tt - e->RTT) / (tt - e->R4IES));
                qs = dmin((alfa * foeewl + (1.0 - alfa) * foeewi) / ap[i], e->QMAX);
            } else {
                const double alfa = e->KFLAG == 1 ? foealfcu(e, tt) : foealfa(e, tt);
                qs = dmin(foeewm_with(e, tt, alfa) / ap[i], e->QMAX);
            }
            qsat[i] = qs / (1.0 - e->RETV * qs);
        }
    }
    return 0;
}

/* cuadjtqs.py:24-37, one Newton step */
static inline void cuadjtqs_step(const Cloudsc2Params* e, double ap, double* t, double* q, double z3es, double z4es,
                                 double z5alcp, double zaldcp) {
    const double foeew = e->R2ES * exp(z3es * (*t - e->RTT) / (*t - z4es));
    double qsat = dmin(foeew / ap, e->ZQMAX);
    const double cor = 1.0 / (1.0 - e->RETV * qsat);
    qsat = qsat * cor;
    const double z2s = z5alcp / sq(*t - z4es);
    const double cond = (*q - qsat) / (1.0 + qsat * cor * z2s);
    *t = *t + zaldcp * cond;
    *q = *q - cond;
}

/* cloudsc2.py:166-186 */
static inline double crh2_of(double eta, double trpaus) {
    const double rh2 = 0.35 + 0.14 * sq((trpaus - 0.25) / 0.15) + 0.04 * dmin(trpaus - 0.25, 0.0) / 0.15;
    const double deta2 = 0.3, bound1 = trpaus + deta2;
    const double deta1 = 0.09 + 0.16 * (0.4 - trpaus) / 0.3, bound2 = 1.0 - deta1;
    if (eta < trpaus) return 1.0;
    if (eta < bound1) return 1.0 + (rh2 - 1.0) * (eta - trpaus) / deta2;
    if (eta < bound2) return rh2;
    return 1.0 + (rh2 - 1.0) * sqrt((1.0 - eta) / deta1);
}

int cs2c_nl(const Cloudsc2Params* e, int nx, int nz, int64_t ls, const double* const* in, const double* eta,
            double* const* out, double dt, int nthreads) {
    if (!e || !in || !out || !eta || nx < 1 || nz < 2 || ls < nx || e->ICALL != 0) return -1;
    for (int i = 0; i < NL_NUM_IN; ++i)
        if (!in[i]) return -1;
    for (int i = 0; i < NL_NUM_OUT; ++i)
        if (!out[i]) return -1;
    if (nthreads < 1) nthreads = cs2c_max_threads();
    const int LEV = e->LEVAPLS2 || e->LDRAIN1D, LIN = e->LPHYLIN || e->LDRAIN1D;
    const double RG = e->RG, RTT = e->RTT, RCPD = e->RCPD;
    /* :120-124 */
    const double ckcodtl = 2.0 * e->RKCONV * dt, ckcodti = 5.0 * e->RKCONV * dt;
    const double cons2 = 1.0 / (RG * dt), cons3 = e->RLVTT / RCPD, meltp2 = RTT + 2.0;
    const double lcrit = LEV ? 1.9 * e->RCLCRIT : 2.0 * e->RCLCRIT; /* :251-254 */
    const double icrit = LEV ? 0.0001 : 2.0 * e->RCLCRIT;           /* :262-265 */
    double* scalm = (double*)malloc(sizeof(double) * (size_t)nz);
    if (!scalm) return -1;
    for (int k = 0; k < nz; ++k) scalm[k] = e->ZSCAL * pow(dmax(eta[k] - 0.2, e->ZEPS1), 0.2); /* :127 */

    const double *AP = in[NL_IN_AP], *APH = in[NL_IN_APH], *LU = in[NL_IN_LU], *LUDE = in[NL_IN_LUDE],
                 *MFD = in[NL_IN_MFD], *MFU = in[NL_IN_MFU], *Q = in[NL_IN_Q], *QI = in[NL_IN_QI], *QL = in[NL_IN_QL],
                 *QSAT = in[NL_IN_QSAT], *SUPSAT = in[NL_IN_SUPSAT], *T = in[NL_IN_T], *TQ = in[NL_IN_TND_CML_Q],
                 *TQI = in[NL_IN_TND_CML_QI], *TQL = in[NL_IN_TND_CML_QL], *TT = in[NL_IN_TND_CML_T];
    const int nblk = (nx + CB - 1) / CB;

#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int b = 0; b < nblk; ++b) {
        const int c0 = b * CB, nc = (nx - c0) < CB ? (nx - c0) : CB;
        double rfl[CB], sfl[CB], covptot[CB], trpaus[CB], aph_s[CB];
        /* :93-100 and :107-111 */
        for (int j = 0; j < nc; ++j) {
            rfl[j] = sfl[j] = covptot[j] = 0.0;
            aph_s[j] = APH[(int64_t)nz * ls + c0 + j];
            trpaus[j] = 0.1;
        }
        for (int k = 0; k < nz - 1; ++k) {
            if (!(eta[k] > 0.1 && eta[k] < 0.4)) continue;
            for (int j = 0; j < nc; ++j) {
                const int64_t i = (int64_t)k * ls + c0 + j;
                if (T[i] + dt * TT[i] > T[i + ls] + dt * TT[i + ls]) trpaus[j] = eta[k];
            }
        }
        for (int j = 0; j < nc; ++j) { /* :396-397 at k = 0 (fplsl/fplsn[0] are not written, Q2) */
            out[NL_OUT_FHPSL][c0 + j] = 0.0;
            out[NL_OUT_FHPSN][c0 + j] = 0.0;
        }
        for (int k = 0; k < nz; ++k) {
            for (int j = 0; j < nc; ++j) {
                const int64_t i = (int64_t)k * ls + c0 + j;
                const double ap = AP[i], qs_in = QSAT[i], aph0 = APH[i], aph1 = APH[i + ls];
                double t = T[i] + dt * TT[i];                 /* :104 */
                double q = Q[i] + dt * TQ[i] + SUPSAT[i];     /* :115 */
                const double ql = QL[i] + dt * TQL[i], qi = QI[i] + dt * TQI[i];
                /* :130-134 */
                const double dp = aph1 - aph0;
                const double zz = RCPD + RCPD * e->RVTMP2 * q;
                const double lfdcp = e->RLMLT / zz, lsdcp = e->RLSTT / zz, lvdcp = e->RLVTT / zz;
                /* :141-160 */
                double fwat, foeew, esdp;
                if (LIN) {
                    const int cold = t < RTT;
                    fwat = cold ? 0.545 * (tanh(0.17 * (t - e->RLPTRC)) + 1.0) : 1.0;
                    const double z3es = cold ? e->R3IES : e->R3LES, z4es = cold ? e->R4IES : e->R4LES;
                    foeew = e->R2ES * exp(z3es * (t - RTT) / (t - z4es));
                    esdp = dmin(foeew / ap, e->ZQMAX);
                } else {
                    fwat = foealfa(e, t);
                    foeew = foeewm_with(e, t, fwat);
                    esdp = foeew / ap;
                }
                const double facw = e->R5LES / sq(t - e->R4LES), faci = e->R5IES / sq(t - e->R4IES);
                const double fac = fwat * facw + (1.0 - fwat) * faci;
                const double dqsdtemp = fac * qs_in / (1.0 - e->RETV * esdp);
                const double corqs = 1.0 + cons3 * dqsdtemp;
                const double qlim = dmin(q, qs_in); /* :163 */
                const double crh2 = crh2_of(eta[k], trpaus[j]);
                /* :189-207 */
                const double qsat = t < e->RTICE ? qs_in * (1.8 - 0.003 * t) : qs_in;
                const double qcrit = crh2 * qsat, qt = q + ql + qi;
                double clc, qc;
                if (qt < qcrit) {
                    clc = 0.0;
                    qc = 0.0;
                } else if (qt >= qsat) {
                    clc = 1.0;
                    qc = (1.0 - scalm[k]) * (qsat - qcrit);
                } else {
                    const double qpd = qsat - qt, qcd = qsat - qcrit;
                    clc = 1.0 - sqrt(qpd / (qcd - scalm[k] * (qt - qcrit)));
                    qc = (scalm[k] * qpd + (1.0 - scalm[k]) * qcd) * sq(clc);
                }
                /* :210-215 */
                const double gdp = RG / (aph1 - aph0);
                const double lude_in = LUDE[i], lude = dt * lude_in * gdp, lu1 = LU[i + ls];
                if (lude >= e->RLMIN && lu1 >= e->ZEPS2) {
                    clc = clc + (1.0 - clc) * (1.0 - exp(-lude / lu1));
                    qc = qc + lude;
                }
                /* :218-224 */
                const double rho = ap / (e->RD * t);
                const double rodqsdp = -rho * qs_in / (ap - e->RETV * foeew);
                const double ldcp = fwat * lvdcp + (1.0 - fwat) * lsdcp;
                const double dtdzmo = RG * (1.0 / RCPD - ldcp * rodqsdp) / (1.0 + ldcp * dqsdtemp);
                const double dqsdz = dqsdtemp * dtdzmo - RG * rodqsdp;
                const double dqc = dmin(dt * dqsdz * (MFU[i] + MFD[i]) / rho, qc);
                qc = qc - dqc;
                /* :227-230 */
                double qlwc = qc * fwat, qiwc = qc * (1.0 - fwat);
                double condl = (qlwc - ql) / dt, condi = (qiwc - qi) / dt;
                /* :234-235 */
                covptot[j] = dmax(covptot[j], clc);
                const double covpclr = dmax(covptot[j] - clc, 0.0);
                /* :238-246 */
                double rfln = rfl[j], sfln = sfl[j];
                if (sfl[j] != 0.0) {
                    const double cons = cons2 * dp / lfdcp;
                    const double snmlt = dmin(sfl[j], cons * dmax(t - meltp2, 0.0));
                    rfln = rfln + snmlt;
                    sfln = sfln - snmlt;
                    t = t - snmlt / cons;
                }
                /* :249-272 */
                double prr = 0.0, prs = 0.0;
                if (clc > e->ZEPS2) {
                    const double cldl = qlwc / clc;
                    const double dl = ckcodtl * (1.0 - exp(-sq(cldl / lcrit)));
                    prr = qlwc - clc * cldl * exp(-dl);
                    qlwc = qlwc - prr;
                    const double cldi = qiwc / clc;
                    const double di = ckcodti * exp(0.025 * (t - RTT)) * (1.0 - exp(-sq(cldi / icrit)));
                    prs = qiwc - clc * cldi * exp(-di);
                    qiwc = qiwc - prs;
                }
                /* :275-285 */
                const double dr = cons2 * dp * (prr + prs);
                double rfreeze = 0.0, fwatr = 1.0;
                if (t < RTT) {
                    rfreeze = cons2 * dp * prr;
                    fwatr = 0.0;
                }
                rfln = rfln + fwatr * dr;
                sfln = sfln + (1.0 - fwatr) * dr;
                /* :288-321 */
                double evapr = 0.0, evaps = 0.0, covptot_out = 0.0;
                if (LEV) {
                    const double prtot = rfln + sfln;
                    if (prtot > e->ZEPS2 && covpclr > e->ZEPS2) {
                        double preclr = prtot * covpclr / covptot[j];
                        const double qe = qs_in - (qs_in - qlim) * covpclr / sq(1.0 - clc);
                        const double beta =
                            RG * e->RPECONS * pow(sqrt(ap / aph_s[j]) / 0.00509 * preclr / covpclr, 0.5777);
                        const double bb = dt * beta * (qs_in - qe) / (1.0 + dt * beta * corqs);
                        const double dtgdp = dt * RG / (aph1 - aph0);
                        const double dpr = dmin(covpclr * bb / dtgdp, preclr);
                        preclr = preclr - dpr;
                        if (preclr <= 0.0) covptot[j] = clc;
                        covptot_out = covptot[j];
                        evapr = dpr * rfln / prtot;
                        rfln = rfln - evapr;
                        evaps = dpr * sfln / prtot;
                        sfln = sfln - evaps;
                    }
                }
                /* :328-344 */
                const double dqdt = -(condl + condi) + (lude_in + evapr + evaps) * gdp;
                const double dtdt = lvdcp * condl + lsdcp * condi -
                                    (lvdcp * evapr + lsdcp * evaps + lude_in * (fwat * lvdcp + (1.0 - fwat) * lsdcp) -
                                     (lsdcp - lvdcp) * rfreeze) * gdp;
                t = t + dt * dtdt;
                q = q + dt * dqdt;
                const double qold = q;
                /* :347 (cuadjtqs.py:40-68, ICALL == 0) */
                {
                    const int warm = t > RTT;
                    const double z3es = warm ? e->R3LES : e->R3IES, z4es = warm ? e->R4LES : e->R4IES;
                    const double z5alcp = warm ? e->R5ALVCP : e->R5ALSCP, zaldcp = warm ? e->RALVDCP : e->RALSDCP;
                    cuadjtqs_step(e, ap, &t, &q, z3es, z4es, z5alcp, zaldcp);
                    cuadjtqs_step(e, ap, &t, &q, z3es, z4es, z5alcp, zaldcp);
                }
                /* :350-364 */
                const double dq = dmax(qold - q, 0.0), dr2 = cons2 * dp * dq;
                double rfreeze2 = 0.0;
                fwatr = 1.0;
                if (t < RTT) {
                    rfreeze2 = fwat * dr2;
                    fwatr = 0.0;
                }
                condl = condl + fwatr * dq / dt;
                condi = condi + (1.0 - fwatr) * dq / dt;
                rfln = rfln + fwatr * dr2;
                sfln = sfln + (1.0 - fwatr) * dr2;
                rfreeze = rfreeze + rfreeze2;
                /* :367-380 */
                out[NL_OUT_CLC][i] = clc;
                out[NL_OUT_COVPTOT][i] = covptot_out;
                out[NL_OUT_TND_Q][i] = -(condl + condi) + (lude_in + evapr + evaps) * gdp;
                out[NL_OUT_TND_T][i] = lvdcp * condl + lsdcp * condi -
                                       (lvdcp * evapr + lsdcp * evaps +
                                        lude_in * (fwat * lvdcp + (1.0 - fwat) * lsdcp) - (lsdcp - lvdcp) * rfreeze) * gdp;
                out[NL_OUT_TND_QL][i] = (qlwc - ql) / dt;
                out[NL_OUT_TND_QI][i] = (qiwc - qi) / dt;
                /* :383-399: carry + shifted flux outputs */
                rfl[j] = rfln;
                sfl[j] = sfln;
                out[NL_OUT_FPLSL][i + ls] = rfln;
                out[NL_OUT_FPLSN][i + ls] = sfln;
                out[NL_OUT_FHPSL][i + ls] = -rfln * e->RLVTT;
                out[NL_OUT_FHPSN][i + ls] = -sfln * e->RLSTT;
            }
        }
    }
    free(scalm);
    return 0;
}
