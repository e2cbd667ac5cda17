/*
 * cloudsc2_hip.h - C ABI of the MI355X-native CLOUDSC2 column-physics engine (libcloudsc2_hip.so).
 *
 * Each entry point replaces ONE GT4Py stencil object of the reference, i.e. the callable that
 * `compile_stencil(name, externals)` returns and that the components invoke with keyword field
 * arguments (`self.cloudsc2(in_ap=..., ..., dt=..., origin=(0,0,0), domain=(nx,1,nz+1), ...)`).
 * The reference call sites are cited per function below (paths relative to
 * /root/reference/src/cloudsc2_gt4py/physics/).
 *
 * Conventions shared by every entry point
 *   - plain C types only; no torch / HIP types in the signatures (`stream` is a hipStream_t
 *     passed as void*, NULL = the legacy default stream);
 *   - all field pointers are DEVICE pointers (HBM) owned by the caller; the library never
 *     allocates, frees or copies fields;
 *   - field layout is [level][column]: element (column c, level k) of a field lives at
 *     ptr[k * lev_stride + c], 0 <= c < nx, 0 <= k <= nz (every field has nz+1 levels, as every
 *     reference storage has, nonlinear/microphysics.py:168-169; the padding level nz of a
 *     full-level field is never written and, for `lu`, must be 0, nonlinear/_stencils/cloudsc2.py:212);
 *   - `eta` is a device vector of nz+1 values (`in_eta`, gtscript.Field[K]);
 *   - pointer-array arguments (`in`, `out`) are HOST arrays of device pointers in the order of
 *     the enum documented with the function (= the order of the gtscript signature);
 *   - the boolean externals LPHYLIN / LDRAIN1D / LEVAPLS2 / LREGCL / IGNORE_SUPSAT select a kernel
 *     instantiation, the numeric externals travel by value in `Cloudsc2Params`;
 *   - return value: 0 on success, <0 on error (CLOUDSC2_E_*); `cloudsc2_last_error()` returns a
 *     thread-local message.  Kernels are launched asynchronously on `stream`;
 *   - threading: the library is written for ONE host thread per process and device (the one-process-per-GPU
 *     model of the drivers).  Error text and kernel-name diagnostics are thread-local; the per-device facts the
 *     launchers cache (CU count, the >64 KiB LDS opt-in of the ring kernels) are relaxed atomics whose only race
 *     is a repeated, idempotent query, so a second thread launching on the same device is safe.  Device ordinals
 *     >= 64 are refused (CLOUDSC2_E_UNSUPPORTED);
 *   - inputs and outputs of one call must not overlap (no in-place calls): the kernels stream level by level
 *     and cloudsc2_ad re-reads its inputs in its second sweep.  The Python stencil objects check this when
 *     called with validate_args=True.
 */
#ifndef CLOUDSC2_HIP_H
#define CLOUDSC2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLOUDSC2_ABI_VERSION 3

#define CLOUDSC2_OK 0
#define CLOUDSC2_E_ARG (-1)      /* bad argument (null pointer, nx/nz/stride out of range)      */
#define CLOUDSC2_E_UNSUPPORTED (-2) /* no kernel for this call: ICALL != 0, or a FUSED build extension (cloudsc2_nl_fused_*,
                                       _nl_taylor*, _tl_incremented_*) on fields of 4 GiB or more ((nz+1) * lev_stride *
                                       sizeof(element) must be < 2^32 there).  cloudsc2_nl / _tl / _ad themselves take
                                       fields of any size: beyond 4 GiB they run their 64-bit-offset instantiation */
#define CLOUDSC2_E_LAUNCH (-3)   /* HIP reported an error at launch                              */
#define CLOUDSC2_E_NODEVICE (-4) /* no HIP device visible                                        */

/* Numeric + boolean externals (names = /root/reference/src/cloudsc2_gt4py/iox.py:25-209 and the
 * literals of nonlinear/microphysics.py:68-78, common/saturation.py:51, common/increment.py:47-49).
 * Mirrored field-for-field by `Cloudsc2Params` in gt4py_dwarf_p_cloudsc2_tl_ad_amd/params.py. */
typedef struct Cloudsc2Params {
    double R2ES, R3IES, R3LES, R4IES, R4LES, R5IES, R5LES;
    double R5ALSCP, R5ALVCP, RALSDCP, RALVDCP;
    double RTICE, RTWAT, RTWAT_RTICE_R, RTICECU, RTWAT_RTICECU_R, RVTMP2;
    double RCPD, RD, RETV, RG, RLMLT, RLSTT, RLVTT, RTT;
    double RCLCRIT, RKCONV, RLMIN, RPECONS, RLPTRC;
    double ZEPS1, ZEPS2, ZQMAX, ZSCAL, QMAX;
    int32_t LPHYLIN, LDRAIN1D, LEVAPLS2, LREGCL, ICALL, KFLAG, IGNORE_SUPSAT, NLEV;
    /* Build extension, NOT a reference external (default 0 = reproduce the reference literally).
     * 1: cloudsc2_ad uses the freezing tests of the NL/TL stencils (post-adjustment t < RTT at
     * adjoint/_stencils/cloudsc2.py:427,:577; the forward test of :343 at :729), which makes AD the
     * exact transpose of TL also in columns where the saturation adjustment crosses RTT
     * (SURVEY.md Appendix B Q4/Q5). */
    int32_t AD_TRAJ_FIX;
} Cloudsc2Params;

int32_t cloudsc2_abi_version(void);
int32_t cloudsc2_params_sizeof(void);
const char* cloudsc2_last_error(void);
/* diagnostics: name of the kernel the calling thread's last successful entry-point call enqueued, e.g.
 * "cs2::nl_ring_kernel" (LDS-ring load path) or "cs2::nl_kernel" (register prefetch); "" before the first launch */
const char* cloudsc2_last_kernel(void);
/* number of HIP devices visible to the library's runtime (0 if none / runtime unusable) */
int32_t cloudsc2_device_count(void);

/* ---- cloudsc2_nl : nonlinear/_stencils/cloudsc2.py:24-399, called at nonlinear/microphysics.py:134-172 */
enum { /* order of `in` */
    NL_IN_AP, NL_IN_APH, NL_IN_LU, NL_IN_LUDE, NL_IN_MFD, NL_IN_MFU, NL_IN_Q, NL_IN_QI, NL_IN_QL,
    NL_IN_QSAT, NL_IN_SUPSAT, NL_IN_T, NL_IN_TND_CML_Q, NL_IN_TND_CML_QI, NL_IN_TND_CML_QL,
    NL_IN_TND_CML_T, NL_NUM_IN
};
enum { /* order of `out` */
    NL_OUT_CLC, NL_OUT_COVPTOT, NL_OUT_FHPSL, NL_OUT_FHPSN, NL_OUT_FPLSL, NL_OUT_FPLSN,
    NL_OUT_TND_Q, NL_OUT_TND_QI, NL_OUT_TND_QL, NL_OUT_TND_T, NL_NUM_OUT
};
int32_t cloudsc2_nl_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                        const double* const* in, const double* eta, double* const* out, double dt,
                        void* stream);
int32_t cloudsc2_nl_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                        const float* const* in, const float* eta, float* const* out, double dt,
                        void* stream);

/* ---- cloudsc2_nl, fused variants (BUILD EXTENSIONS - the reference has no such stencil).
 * Exactly one of `qsat_out` / `in_i` is non-NULL; results are those of the separate stencil calls.
 *   qsat_out != NULL : `saturation` (LPHYLIN form, common/_stencils/saturation.py:30-35,42) is evaluated inside the
 *                      NL kernel from in[NL_IN_AP], in[NL_IN_T]; in[NL_IN_QSAT] is not read (may be NULL) and the
 *                      result is written to qsat_out: the driver's timed region (run_nonlinear.py:117-118) in ONE launch.
 *   in_i != NULL     : every input is read as in[f] + pf * in_i[f] (perturbed_state, perturbed_state.py:75-91): the
 *                      Taylor test's perturbed NL runs (tangent_linear/validation.py:166-176) without the perturbed copy. */
int32_t cloudsc2_nl_fused_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                              const double* const* in, const double* const* in_i, double pf, double* qsat_out,
                              const double* eta, double* const* out, double dt, void* stream);
int32_t cloudsc2_nl_fused_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                              const float* const* in, const float* const* in_i, double pf, float* qsat_out,
                              const float* eta, float* const* out, double dt, void* stream);

/* ---- perturbed NL run + Taylor-test reduction (BUILD EXTENSION, SURVEY.md 8f rank 1, third item).
 * Replaces, per step size, perturbed_state + cloudsc2_nl + the ten field differences and sums of
 * TaylorTest.run / get_field_norm (tangent_linear/validation.py:166-176, :239-249): NL is evaluated on in + pf * in_i,
 * nothing is stored, and workgroup b writes  partials[b * NL_NUM_OUT + f] = sum over its columns and all levels of
 * (NL(in + pf in_i) - ref_out)[f]  in double precision (f in NL_OUT_* order; ref_out = the unperturbed NL outputs,
 * read-only).  `partials` is a DEVICE array of cloudsc2_nl_taylor_blocks(nx) * NL_NUM_OUT doubles; the caller adds
 * the blocks (fixed order: deterministic). */
int32_t cloudsc2_nl_taylor_blocks(int32_t nx);
int32_t cloudsc2_nl_taylor_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                               const double* const* in, const double* const* in_i, double pf, const double* eta,
                               const double* const* ref_out, double* partials, double dt, void* stream);
int32_t cloudsc2_nl_taylor_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                               const float* const* in, const float* const* in_i, double pf, const float* eta,
                               const float* const* ref_out, double* partials, double dt, void* stream);

/* ---- the Taylor test's perturbed runs for SEVERAL step sizes per launch (BUILD EXTENSION).
 * Replaces the whole loop of TaylorTest.run (tangent_linear/validation.py:162-176: perturbed_state, cloudsc2_nl, get_norm
 * per step size): a lane loads the 16 state + 16 increment + 10 reference words of a level once and evaluates the level
 * for up to 5 step sizes on them (internally ceil(nf / 5) launches), so the perturbed runs are bound by arithmetic instead
 * of re-streaming 42 words per level, column and step size.  `in_i` may be NULL: the increments are then formed in the
 * kernel as T(inc_f) * in (state_increment fused in, state_increment.py:61-80; p->IGNORE_SUPSAT zeroes the supsat
 * increment), otherwise inc_f is ignored.  `pf`: HOST array of the nf step sizes; `partials`: DEVICE
 * array of cloudsc2_nl_taylor_blocks(nx) * nf * NL_NUM_OUT doubles, partials[(b * nf + j) * NL_NUM_OUT + f] = sum over
 * workgroup b's columns and all levels of (NL(in + pf[j] in_i) - ref_out)[f]; the caller adds the blocks. */
int32_t cloudsc2_nl_taylor_multi_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                     const double* const* in, const double* const* in_i, double inc_f, int32_t nf,
                                     const double* pf, const double* eta, const double* const* ref_out, double* partials,
                                     double dt, void* stream);
int32_t cloudsc2_nl_taylor_multi_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                     const float* const* in, const float* const* in_i, double inc_f, int32_t nf,
                                     const double* pf, const float* eta, const float* const* ref_out, double* partials,
                                     double dt, void* stream);

/* ---- validation-norm reductions (BUILD EXTENSIONS; the reference reduces on the host with NumPy).
 * field_sums : per field f < nfields (<= 16), sum over nlev levels and nx columns of a[f] - b[f] (b == NULL: of a[f]);
 *              the difference is formed in the field type, accumulated in double - TaylorTest.get_field_norm's
 *              np.sum(field_nl_p - field_nl) and np.sum(field_tl) (tangent_linear/validation.py:250-261).  Workgroup w
 *              writes partials[w * nfields + f]; `partials` holds cloudsc2_field_sums_blocks(nx, nlev) * nfields doubles.
 * column_dots: per column c, the sum over pairs p < npairs (<= 16) and nlev levels of a[p][k][c] * b[p][k][c] in double -
 *              SymmetryTest.get_norm1 / get_norm2 (adjoint/validation.py:167-215) - delivered as level-chunk partials:
 *              out[j * nx + c] (+)= the sum over chunk j's levels, j < cloudsc2_column_dots_chunks(nlev); the caller adds
 *              the chunks.  `accumulate` != 0 adds to what `out` holds (for more than 16 pairs: a second call). */
int32_t cloudsc2_field_sums_blocks(int32_t nx, int32_t nlev);
int32_t cloudsc2_column_dots_chunks(int32_t nlev);
int32_t cloudsc2_field_sums_f64(int32_t nx, int32_t nlev, int64_t lev_stride, int32_t nfields, const double* const* a,
                                const double* const* b, double* partials, void* stream);
int32_t cloudsc2_field_sums_f32(int32_t nx, int32_t nlev, int64_t lev_stride, int32_t nfields, const float* const* a,
                                const float* const* b, double* partials, void* stream);
int32_t cloudsc2_column_dots_f64(int32_t nx, int32_t nlev, int64_t lev_stride, int32_t npairs, const double* const* a,
                                 const double* const* b, double* out, int32_t accumulate, void* stream);
int32_t cloudsc2_column_dots_f32(int32_t nx, int32_t nlev, int64_t lev_stride, int32_t npairs, const float* const* a,
                                 const float* const* b, double* out, int32_t accumulate, void* stream);

/* ---- saturation : common/_stencils/saturation.py:23-42, called at common/saturation.py:67-76
 * (domain nx x 1 x nz: level nz of out_qsat is not written) */
int32_t cloudsc2_saturation_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                const double* ap, const double* t, double* qsat, void* stream);
int32_t cloudsc2_saturation_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                const float* ap, const float* t, float* qsat, void* stream);

/* ---- state_increment : common/_stencils/state_increment.py:22-80, called at common/increment.py:93-132
 * ---- perturbed_state : common/_stencils/perturbed_state.py:22-91, called at common/increment.py:219-261
 * Field order of the 16-entry arrays (same for in / in_i / out): */
enum {
    INC_APH, INC_AP, INC_Q, INC_QSAT, INC_T, INC_QL, INC_QI, INC_LUDE, INC_LU, INC_MFU, INC_MFD,
    INC_TND_CML_T, INC_TND_CML_Q, INC_TND_CML_QL, INC_TND_CML_QI, INC_SUPSAT, INC_NUM
};
int32_t cloudsc2_state_increment_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                     const double* const* in, double* const* out_i, double f, void* stream);
int32_t cloudsc2_state_increment_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                     const float* const* in, float* const* out_i, double f, void* stream);
int32_t cloudsc2_perturbed_state_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                     const double* const* in, const double* const* in_i,
                                     double* const* out, double f, void* stream);
int32_t cloudsc2_perturbed_state_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                     const float* const* in, const float* const* in_i,
                                     float* const* out, double f, void* stream);

/* ---- cloudsc2_tl : tangent_linear/_stencils/cloudsc2.py:23-774, called at tangent_linear/microphysics.py:162-242
 * `in` / `in_i`: the 16 NL inputs and their perturbations, NL_IN_* order;
 * `out` / `out_i`: the 10 NL outputs and their perturbations, NL_OUT_* order. */
int32_t cloudsc2_tl_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                        const double* const* in, const double* const* in_i, const double* eta,
                        double* const* out, double* const* out_i, double dt, void* stream);
int32_t cloudsc2_tl_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                        const float* const* in, const float* const* in_i, const float* eta,
                        float* const* out, float* const* out_i, double dt, void* stream);

/* ---- cloudsc2_tl with state_increment fused in (BUILD EXTENSION).  The harnesses call state_increment and cloudsc2_tl back
 * to back on the same state (tangent_linear/validation.py:159-164, adjoint/validation.py:138-143); here the perturbation
 * fields are not read but formed in the kernel as in_i[f] = T(f) * in[f] (state_increment.py:61-80; with
 * p->IGNORE_SUPSAT the supsat perturbation is 0, :77-80): 16 input streams instead of 32 and no increment launch.  The
 * increments are the very products the increment kernel would store; the outputs equal those of the two separate calls up
 * to the compiler's fma contraction of the shared level function (ulps). */
int32_t cloudsc2_tl_incremented_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                    const double* const* in, double f, const double* eta, double* const* out,
                                    double* const* out_i, double dt, void* stream);
int32_t cloudsc2_tl_incremented_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                    const float* const* in, double f, const float* eta, float* const* out,
                                    float* const* out_i, double dt, void* stream);

/* ---- cloudsc2_ad : adjoint/_stencils/cloudsc2.py:24-996, called at adjoint/microphysics.py:159-238
 * `in`     : the 16 NL inputs (trajectory), NL_IN_* order;
 * `in_adj` : adjoint forcing = perturbations of the 10 NL outputs, NL_OUT_* order
 *            (in_clc_i, in_covptot_i, in_fhpsl_i, in_fhpsn_i, in_fplsl_i, in_fplsn_i,
 *             in_tnd_q_i, in_tnd_qi_i, in_tnd_ql_i, in_tnd_t_i); NOT modified (the reference
 *            zeroes them in place, adjoint/_stencils/cloudsc2.py:481-484 ...; nothing reads them
 *            afterwards);
 * `out`    : the 10 NL outputs recomputed along the trajectory, NL_OUT_* order;
 * `out_adj`: adjoint of the 16 inputs, NL_IN_* order (out_ap_i ... out_tnd_cml_t_i). */
int32_t cloudsc2_ad_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                        const double* const* in, const double* const* in_adj, const double* eta,
                        double* const* out, double* const* out_adj, double dt, void* stream);
int32_t cloudsc2_ad_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                        const float* const* in, const float* const* in_adj, const float* eta,
                        float* const* out, float* const* out_adj, double dt, void* stream);

/* ---- cloudsc2_ad WITHOUT its forward sweep (BUILD EXTENSION).  The symmetry test calls cloudsc2_tl and then cloudsc2_ad on
 * the same state (adjoint/validation.py:135-151).  The kernel behind cloudsc2_ad_* recomputes the NL trajectory in a first
 * sweep only to obtain, per level, the rain / snow fluxes entering it - and those are the NL outputs out_fplsl / out_fplsn
 * the TL call has just written.  Here they are READ (`traj_fplsl`, `traj_fplsn`: (nz+1)-level fields as cloudsc2_nl /
 * cloudsc2_tl write them) and the forward sweep is skipped: 44 words per level and column instead of 70.  Nothing but
 * `out_adj` is written (the recomputed NL outputs of cloudsc2_ad_* are not produced: they are the TL call's).  With fluxes
 * that come from cloudsc2_ad_*'s own `out` the adjoints are bit-identical to that call's; with a TL call's they agree to
 * rounding (the two kernels contract the same formulas differently).  Driver switches only: LEVAPLS2 / LDRAIN1D are
 * refused (CLOUDSC2_E_UNSUPPORTED), as are fields of 4 GiB and more. */
int32_t cloudsc2_ad_from_trajectory_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                        const double* const* in, const double* const* in_adj, const double* eta,
                                        const double* traj_fplsl, const double* traj_fplsn, double* const* out_adj, double dt,
                                        void* stream);
int32_t cloudsc2_ad_from_trajectory_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t lev_stride,
                                        const float* const* in, const float* const* in_adj, const float* eta,
                                        const float* traj_fplsl, const float* traj_fplsn, float* const* out_adj, double dt,
                                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDSC2_HIP_H */
