// extern "C" boundary of libcloudsc2_hip.so - see include/cloudsc2_hip.h for the contract.
// Argument checking happens here (on the host, before anything is launched): a kernel is only
// launched when the shapes it assumes hold, because a faulting kernel can take the whole node down.
#include <cstdarg>
#include <cstdio>

#include "cloudsc2_common.hpp"

namespace cs2 {
template <typename T>
int launch_nl(const Cloudsc2Params&, int, int, int64_t, const T* const*, const T*, T* const*, double, hipStream_t,
              const T* const*, double, T*, double*);
template <typename T>
int launch_nl_taylor_multi(const Cloudsc2Params&, int, int, int64_t, const T* const*, const T* const*, int, const double*,
                           const T*, const T* const*, double*, double, hipStream_t, double);
int field_sums_blocks(int, int);
int column_dots_chunks(int);
template <typename T>
int launch_field_sums(int, int, int64_t, int, const T* const*, const T* const*, double*, hipStream_t);
template <typename T>
int launch_column_dots(int, int, int64_t, int, const T* const*, const T* const*, double*, int, hipStream_t);
template <typename T>
int launch_tl(const Cloudsc2Params&, int, int, int64_t, const T* const*, const T* const*, const T*, T* const*,
              T* const*, double, hipStream_t, double);
template <typename T>
int launch_ad(const Cloudsc2Params&, int, int, int64_t, const T* const*, const T* const*, const T*, T* const*,
              T* const*, double, hipStream_t, const T* traj_l = nullptr, const T* traj_n = nullptr);
template <typename T>
int launch_saturation(const Cloudsc2Params&, int, int, int64_t, const T*, const T*, T*, hipStream_t);
template <typename T>
int launch_increment(const Cloudsc2Params&, int, int, int64_t, const T* const*, T* const*, double, hipStream_t);
template <typename T>
int launch_perturb(int, int, int64_t, const T* const*, const T* const*, T* const*, double, hipStream_t);
}  // namespace cs2

namespace {

thread_local char g_err[512] = "";
thread_local const char* g_kernel = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

constexpr int kMaxLevels = 4095;  // LDS table of the register-path kernels: 2 * (nz+1) * 8 B <= 64 KiB (no opt-in needed)

int check_common(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls) {
    if (!p) return fail(CLOUDSC2_E_ARG, "%s: params is NULL", fn);
    if (nx < 0) return fail(CLOUDSC2_E_ARG, "%s: nx=%d < 0", fn, nx);
    if (nz < 2 || nz > kMaxLevels) return fail(CLOUDSC2_E_ARG, "%s: nz=%d outside [2, %d]", fn, nz, kMaxLevels);
    if (ls < nx) return fail(CLOUDSC2_E_ARG, "%s: lev_stride=%lld < nx=%d", fn, (long long)ls, nx);
    return 0;
}

template <typename T>
int check_ptrs(const char* fn, const char* what, const T* const* arr, int n) {
    if (!arr) return fail(CLOUDSC2_E_ARG, "%s: %s is NULL", fn, what);
    for (int i = 0; i < n; ++i)
        if (!arr[i]) return fail(CLOUDSC2_E_ARG, "%s: %s[%d] is NULL", fn, what, i);
    return 0;
}

int launched(const char* fn, int rc) {
    if (rc == 0) return CLOUDSC2_OK;
    if (rc == -2)   // the launchers' refusals: a fused build extension on fields of 4 GiB or more, or a device ordinal >= 64
        return fail(CLOUDSC2_E_UNSUPPORTED,
                    "%s: this build extension keeps 32-bit byte offsets: (nz+1) * lev_stride * sizeof(element) must be < 2^32 "
                    "bytes per field (the plain stencils cloudsc2_nl / _tl / _ad have no such limit: use them, or narrower "
                    "allocations) (or: HIP device ordinal >= 64, LDS demand beyond one CU)", fn);
    return fail(CLOUDSC2_E_LAUNCH, "%s: HIP launch failed: %s", fn, hipGetErrorString(hipPeekAtLastError()));
}

template <typename T>
int nl_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
            const T* eta, T* const* out, double dt, void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (int rc = check_ptrs(fn, "in", in, NL_NUM_IN)) return rc;
    if (int rc = check_ptrs(fn, "out", const_cast<const T* const*>(out), NL_NUM_OUT)) return rc;
    if (!eta) return fail(CLOUDSC2_E_ARG, "%s: eta is NULL", fn);
    if (p->ICALL != 0) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: ICALL=%d (the reference implements ICALL == 0 only)", fn, p->ICALL);
    if (!(dt > 0.0)) return fail(CLOUDSC2_E_ARG, "%s: dt=%g must be > 0", fn, dt);
    return launched(fn, cs2::launch_nl<T>(*p, nx, nz, ls, in, eta, out, dt, static_cast<hipStream_t>(stream), nullptr,
                                          0.0, nullptr, nullptr));
}

// Fused variants of cloudsc2_nl (build extensions): exactly one of `qsat_out` / `in_i` is non-NULL.
template <typename T>
int nl_fused_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
                  const T* const* in_i, double pf, T* qsat_out, const T* eta, T* const* out, double dt, void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (int rc = check_ptrs(fn, "out", const_cast<const T* const*>(out), NL_NUM_OUT)) return rc;
    if (!in) return fail(CLOUDSC2_E_ARG, "%s: in is NULL", fn);
    if ((qsat_out != nullptr) == (in_i != nullptr))
        return fail(CLOUDSC2_E_ARG, "%s: exactly one of qsat_out (fused saturation) and in_i (fused perturbation) must be given", fn);
    for (int i = 0; i < NL_NUM_IN; ++i)
        if (!in[i] && !(qsat_out && i == NL_IN_QSAT)) return fail(CLOUDSC2_E_ARG, "%s: in[%d] is NULL", fn, i);
    if (in_i)
        if (int rc = check_ptrs(fn, "in_i", in_i, NL_NUM_IN)) return rc;
    if (!eta) return fail(CLOUDSC2_E_ARG, "%s: eta is NULL", fn);
    if (p->ICALL != 0) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: ICALL=%d unsupported", fn, p->ICALL);
    if (qsat_out && !p->LPHYLIN)
        return fail(CLOUDSC2_E_UNSUPPORTED, "%s: only the LPHYLIN form of saturation is available fused", fn);
    if (!(dt > 0.0)) return fail(CLOUDSC2_E_ARG, "%s: dt=%g must be > 0", fn, dt);
    return launched(fn, cs2::launch_nl<T>(*p, nx, nz, ls, in, eta, out, dt, static_cast<hipStream_t>(stream), in_i, pf,
                                          qsat_out, nullptr));
}

// Perturbed NL run + Taylor-test reduction (build extension): see include/cloudsc2_hip.h.
template <typename T>
int nl_taylor_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
                   const T* const* in_i, double pf, const T* eta, const T* const* ref_out, double* partials, double dt,
                   void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (int rc = check_ptrs(fn, "in", in, NL_NUM_IN)) return rc;
    if (int rc = check_ptrs(fn, "in_i", in_i, NL_NUM_IN)) return rc;
    if (int rc = check_ptrs(fn, "ref_out", ref_out, NL_NUM_OUT)) return rc;
    if (!eta) return fail(CLOUDSC2_E_ARG, "%s: eta is NULL", fn);
    if (!partials) return fail(CLOUDSC2_E_ARG, "%s: partials is NULL", fn);
    if (p->ICALL != 0) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: ICALL=%d unsupported", fn, p->ICALL);
    if (!(dt > 0.0)) return fail(CLOUDSC2_E_ARG, "%s: dt=%g must be > 0", fn, dt);
    // the kernel only READS the reference outputs; the launcher's pointer pack is the mutable one
    T* refs[NL_NUM_OUT];
    for (int i = 0; i < NL_NUM_OUT; ++i) refs[i] = const_cast<T*>(ref_out[i]);
    return launched(fn, cs2::launch_nl<T>(*p, nx, nz, ls, in, eta, refs, dt, static_cast<hipStream_t>(stream), in_i, pf,
                                          nullptr, partials));
}

// The Taylor test's perturbed runs, several step sizes per launch (build extension): see include/cloudsc2_hip.h.
template <typename T>
int nl_taylor_multi_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
                         const T* const* in_i, double inc_f, int32_t nf, const double* pf, const T* eta,
                         const T* const* ref_out, double* partials, double dt, void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nf < 0 || nf > 64) return fail(CLOUDSC2_E_ARG, "%s: nf=%d outside [0, 64]", fn, nf);
    if (nx == 0 || nf == 0) return CLOUDSC2_OK;
    if (int rc = check_ptrs(fn, "in", in, NL_NUM_IN)) return rc;
    if (in_i)      // NULL: the increments are formed in the kernel as inc_f * in (state_increment fused in)
        if (int rc = check_ptrs(fn, "in_i", in_i, NL_NUM_IN)) return rc;
    if (int rc = check_ptrs(fn, "ref_out", ref_out, NL_NUM_OUT)) return rc;
    if (!eta) return fail(CLOUDSC2_E_ARG, "%s: eta is NULL", fn);
    if (!pf) return fail(CLOUDSC2_E_ARG, "%s: pf is NULL", fn);
    if (!partials) return fail(CLOUDSC2_E_ARG, "%s: partials is NULL", fn);
    if (p->ICALL != 0) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: ICALL=%d unsupported", fn, p->ICALL);
    if (!(dt > 0.0)) return fail(CLOUDSC2_E_ARG, "%s: dt=%g must be > 0", fn, dt);
    if (nz > 2000) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: nz=%d: the level table and the running sums must share the LDS", fn, nz);
    return launched(fn, cs2::launch_nl_taylor_multi<T>(*p, nx, nz, ls, in, in_i, nf, pf, eta, ref_out, partials, dt,
                                                       static_cast<hipStream_t>(stream), inc_f));
}

template <typename T>
int sums_impl(const char* fn, int32_t nx, int32_t nlev, int64_t ls, int32_t nf, const T* const* a, const T* const* b,
              double* partials, void* stream) {
    if (nx < 0 || nlev < 1 || ls < nx) return fail(CLOUDSC2_E_ARG, "%s: nx=%d nlev=%d lev_stride=%lld", fn, nx, nlev, (long long)ls);
    if (nf < 1 || nf > 16) return fail(CLOUDSC2_E_ARG, "%s: nfields=%d outside [1, 16]", fn, nf);
    if (nx == 0) return CLOUDSC2_OK;
    if (int rc = check_ptrs(fn, "a", a, nf)) return rc;
    if (b)
        if (int rc = check_ptrs(fn, "b", b, nf)) return rc;
    if (!partials) return fail(CLOUDSC2_E_ARG, "%s: partials is NULL", fn);
    return launched(fn, cs2::launch_field_sums<T>(nx, nlev, ls, nf, a, b, partials, static_cast<hipStream_t>(stream)));
}

template <typename T>
int dots_impl(const char* fn, int32_t nx, int32_t nlev, int64_t ls, int32_t np, const T* const* a, const T* const* b,
              double* out, int32_t accumulate, void* stream) {
    if (nx < 0 || nlev < 1 || ls < nx) return fail(CLOUDSC2_E_ARG, "%s: nx=%d nlev=%d lev_stride=%lld", fn, nx, nlev, (long long)ls);
    if (np < 1 || np > 16) return fail(CLOUDSC2_E_ARG, "%s: npairs=%d outside [1, 16]", fn, np);
    if (nx == 0) return CLOUDSC2_OK;
    if (int rc = check_ptrs(fn, "a", a, np)) return rc;
    if (int rc = check_ptrs(fn, "b", b, np)) return rc;
    if (!out) return fail(CLOUDSC2_E_ARG, "%s: out is NULL", fn);
    return launched(fn, cs2::launch_column_dots<T>(nx, nlev, ls, np, a, b, out, accumulate, static_cast<hipStream_t>(stream)));
}

// `fused_inc`: the state_increment-fused variant (cloudsc2_tl_incremented_*): in_i is absent, perturbations = inc_f * in
template <typename T>
int tl_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
            const T* const* in_i, const T* eta, T* const* out, T* const* out_i, double dt, void* stream,
            bool fused_inc = false, double inc_f = 0.0) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (int rc = check_ptrs(fn, "in", in, NL_NUM_IN)) return rc;
    if (!fused_inc)
        if (int rc = check_ptrs(fn, "in_i", in_i, NL_NUM_IN)) return rc;
    if (int rc = check_ptrs(fn, "out", const_cast<const T* const*>(out), NL_NUM_OUT)) return rc;
    if (int rc = check_ptrs(fn, "out_i", const_cast<const T* const*>(out_i), NL_NUM_OUT)) return rc;
    if (!eta) return fail(CLOUDSC2_E_ARG, "%s: eta is NULL", fn);
    if (p->ICALL != 0) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: ICALL=%d unsupported", fn, p->ICALL);
    if (!(dt > 0.0)) return fail(CLOUDSC2_E_ARG, "%s: dt=%g must be > 0", fn, dt);
    if (p->NLEV != nz) return fail(CLOUDSC2_E_ARG, "%s: NLEV=%d != nz=%d", fn, p->NLEV, nz);
    return launched(fn, cs2::launch_tl<T>(*p, nx, nz, ls, in, fused_inc ? nullptr : in_i, eta, out, out_i, dt,
                                          static_cast<hipStream_t>(stream), inc_f));
}

template <typename T>
int ad_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
            const T* const* in_adj, const T* eta, T* const* out, T* const* out_adj, double dt, void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (int rc = check_ptrs(fn, "in", in, NL_NUM_IN)) return rc;
    if (int rc = check_ptrs(fn, "in_adj", in_adj, NL_NUM_OUT)) return rc;
    if (int rc = check_ptrs(fn, "out", const_cast<const T* const*>(out), NL_NUM_OUT)) return rc;
    if (int rc = check_ptrs(fn, "out_adj", const_cast<const T* const*>(out_adj), NL_NUM_IN)) return rc;
    if (!eta) return fail(CLOUDSC2_E_ARG, "%s: eta is NULL", fn);
    if (p->ICALL != 0) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: ICALL=%d unsupported", fn, p->ICALL);
    if (!(dt > 0.0)) return fail(CLOUDSC2_E_ARG, "%s: dt=%g must be > 0", fn, dt);
    if (p->NLEV != nz) return fail(CLOUDSC2_E_ARG, "%s: NLEV=%d != nz=%d", fn, p->NLEV, nz);
    return launched(fn, cs2::launch_ad<T>(*p, nx, nz, ls, in, in_adj, eta, out, out_adj, dt, static_cast<hipStream_t>(stream)));
}

template <typename T>
int ad_traj_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
                 const T* const* in_adj, const T* eta, const T* traj_fplsl, const T* traj_fplsn, T* const* out_adj, double dt,
                 void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;
    if (int rc = check_ptrs(fn, "in", in, NL_NUM_IN)) return rc;
    if (int rc = check_ptrs(fn, "in_adj", in_adj, NL_NUM_OUT)) return rc;
    if (int rc = check_ptrs(fn, "out_adj", const_cast<const T* const*>(out_adj), NL_NUM_IN)) return rc;
    if (!eta || !traj_fplsl || !traj_fplsn) return fail(CLOUDSC2_E_ARG, "%s: eta / traj_fplsl / traj_fplsn is NULL", fn);
    if (p->ICALL != 0) return fail(CLOUDSC2_E_UNSUPPORTED, "%s: ICALL=%d unsupported", fn, p->ICALL);
    if (p->LEVAPLS2 || p->LDRAIN1D)
        return fail(CLOUDSC2_E_UNSUPPORTED, "%s: the trajectory variant covers the driver switches only (no evaporation "
                    "block: its parked precipitation cover has no counterpart among the NL outputs) - use cloudsc2_ad", fn);
    if (!(dt > 0.0)) return fail(CLOUDSC2_E_ARG, "%s: dt=%g must be > 0", fn, dt);
    if (p->NLEV != nz) return fail(CLOUDSC2_E_ARG, "%s: NLEV=%d != nz=%d", fn, p->NLEV, nz);
    return launched(fn, cs2::launch_ad<T>(*p, nx, nz, ls, in, in_adj, eta, nullptr, out_adj, dt,
                                          static_cast<hipStream_t>(stream), traj_fplsl, traj_fplsn));
}

template <typename T>
int sat_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* ap, const T* t,
             T* qsat, void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (!ap || !t || !qsat) return fail(CLOUDSC2_E_ARG, "%s: NULL field pointer", fn);
    return launched(fn, cs2::launch_saturation<T>(*p, nx, nz, ls, ap, t, qsat, static_cast<hipStream_t>(stream)));
}

template <typename T>
int inc_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
             T* const* out, double f, void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (int rc = check_ptrs(fn, "in", in, INC_NUM)) return rc;
    if (int rc = check_ptrs(fn, "out_i", const_cast<const T* const*>(out), INC_NUM)) return rc;
    return launched(fn, cs2::launch_increment<T>(*p, nx, nz, ls, in, out, f, static_cast<hipStream_t>(stream)));
}

template <typename T>
int per_impl(const char* fn, const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const T* const* in,
             const T* const* in_i, T* const* out, double f, void* stream) {
    if (int rc = check_common(fn, p, nx, nz, ls)) return rc;
    if (nx == 0) return CLOUDSC2_OK;   // empty call: nothing to check or launch (zero-size tensors carry NULL pointers)
    if (int rc = check_ptrs(fn, "in", in, INC_NUM)) return rc;
    if (int rc = check_ptrs(fn, "in_i", in_i, INC_NUM)) return rc;
    if (int rc = check_ptrs(fn, "out", const_cast<const T* const*>(out), INC_NUM)) return rc;
    return launched(fn, cs2::launch_perturb<T>(nx, nz, ls, in, in_i, out, f, static_cast<hipStream_t>(stream)));
}

}  // namespace

namespace cs2 {
void note_kernel(const char* name) { g_kernel = name; }
}  // namespace cs2

extern "C" {

int32_t cloudsc2_abi_version(void) { return CLOUDSC2_ABI_VERSION; }
int32_t cloudsc2_params_sizeof(void) { return (int32_t)sizeof(Cloudsc2Params); }
const char* cloudsc2_last_error(void) { return g_err; }
const char* cloudsc2_last_kernel(void) { return g_kernel; }
int32_t cloudsc2_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int32_t cloudsc2_nl_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                        const double* eta, double* const* out, double dt, void* stream) {
    return nl_impl<double>("cloudsc2_nl_f64", p, nx, nz, ls, in, eta, out, dt, stream);
}
int32_t cloudsc2_nl_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                        const float* eta, float* const* out, double dt, void* stream) {
    return nl_impl<float>("cloudsc2_nl_f32", p, nx, nz, ls, in, eta, out, dt, stream);
}
int32_t cloudsc2_nl_fused_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                              const double* const* in_i, double pf, double* qsat_out, const double* eta,
                              double* const* out, double dt, void* stream) {
    return nl_fused_impl<double>("cloudsc2_nl_fused_f64", p, nx, nz, ls, in, in_i, pf, qsat_out, eta, out, dt, stream);
}
int32_t cloudsc2_nl_fused_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                              const float* const* in_i, double pf, float* qsat_out, const float* eta,
                              float* const* out, double dt, void* stream) {
    return nl_fused_impl<float>("cloudsc2_nl_fused_f32", p, nx, nz, ls, in, in_i, pf, qsat_out, eta, out, dt, stream);
}
int32_t cloudsc2_nl_taylor_blocks(int32_t nx) { return nx <= 0 ? 0 : (nx + cs2::kColBlock - 1) / cs2::kColBlock; }
int32_t cloudsc2_nl_taylor_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                               const double* const* in_i, double pf, const double* eta, const double* const* ref_out,
                               double* partials, double dt, void* stream) {
    return nl_taylor_impl<double>("cloudsc2_nl_taylor_f64", p, nx, nz, ls, in, in_i, pf, eta, ref_out, partials, dt, stream);
}
int32_t cloudsc2_nl_taylor_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                               const float* const* in_i, double pf, const float* eta, const float* const* ref_out,
                               double* partials, double dt, void* stream) {
    return nl_taylor_impl<float>("cloudsc2_nl_taylor_f32", p, nx, nz, ls, in, in_i, pf, eta, ref_out, partials, dt, stream);
}
int32_t cloudsc2_nl_taylor_multi_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                                     const double* const* in_i, double inc_f, int32_t nf, const double* pf, const double* eta,
                                     const double* const* ref_out, double* partials, double dt, void* stream) {
    return nl_taylor_multi_impl<double>("cloudsc2_nl_taylor_multi_f64", p, nx, nz, ls, in, in_i, inc_f, nf, pf, eta, ref_out,
                                        partials, dt, stream);
}
int32_t cloudsc2_nl_taylor_multi_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                                     const float* const* in_i, double inc_f, int32_t nf, const double* pf, const float* eta,
                                     const float* const* ref_out, double* partials, double dt, void* stream) {
    return nl_taylor_multi_impl<float>("cloudsc2_nl_taylor_multi_f32", p, nx, nz, ls, in, in_i, inc_f, nf, pf, eta, ref_out,
                                       partials, dt, stream);
}
int32_t cloudsc2_field_sums_blocks(int32_t nx, int32_t nlev) { return cs2::field_sums_blocks(nx, nlev); }
int32_t cloudsc2_column_dots_chunks(int32_t nlev) { return cs2::column_dots_chunks(nlev); }
int32_t cloudsc2_field_sums_f64(int32_t nx, int32_t nlev, int64_t ls, int32_t nf, const double* const* a,
                                const double* const* b, double* partials, void* stream) {
    return sums_impl<double>("cloudsc2_field_sums_f64", nx, nlev, ls, nf, a, b, partials, stream);
}
int32_t cloudsc2_field_sums_f32(int32_t nx, int32_t nlev, int64_t ls, int32_t nf, const float* const* a,
                                const float* const* b, double* partials, void* stream) {
    return sums_impl<float>("cloudsc2_field_sums_f32", nx, nlev, ls, nf, a, b, partials, stream);
}
int32_t cloudsc2_column_dots_f64(int32_t nx, int32_t nlev, int64_t ls, int32_t np, const double* const* a,
                                 const double* const* b, double* out, int32_t accumulate, void* stream) {
    return dots_impl<double>("cloudsc2_column_dots_f64", nx, nlev, ls, np, a, b, out, accumulate, stream);
}
int32_t cloudsc2_column_dots_f32(int32_t nx, int32_t nlev, int64_t ls, int32_t np, const float* const* a,
                                 const float* const* b, double* out, int32_t accumulate, void* stream) {
    return dots_impl<float>("cloudsc2_column_dots_f32", nx, nlev, ls, np, a, b, out, accumulate, stream);
}
int32_t cloudsc2_tl_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                        const double* const* in_i, const double* eta, double* const* out, double* const* out_i,
                        double dt, void* stream) {
    return tl_impl<double>("cloudsc2_tl_f64", p, nx, nz, ls, in, in_i, eta, out, out_i, dt, stream);
}
int32_t cloudsc2_tl_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                        const float* const* in_i, const float* eta, float* const* out, float* const* out_i,
                        double dt, void* stream) {
    return tl_impl<float>("cloudsc2_tl_f32", p, nx, nz, ls, in, in_i, eta, out, out_i, dt, stream);
}
int32_t cloudsc2_tl_incremented_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                                    double f, const double* eta, double* const* out, double* const* out_i, double dt,
                                    void* stream) {
    return tl_impl<double>("cloudsc2_tl_incremented_f64", p, nx, nz, ls, in, nullptr, eta, out, out_i, dt, stream, true, f);
}
int32_t cloudsc2_tl_incremented_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                                    double f, const float* eta, float* const* out, float* const* out_i, double dt,
                                    void* stream) {
    return tl_impl<float>("cloudsc2_tl_incremented_f32", p, nx, nz, ls, in, nullptr, eta, out, out_i, dt, stream, true, f);
}
int32_t cloudsc2_ad_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                        const double* const* in_adj, const double* eta, double* const* out,
                        double* const* out_adj, double dt, void* stream) {
    return ad_impl<double>("cloudsc2_ad_f64", p, nx, nz, ls, in, in_adj, eta, out, out_adj, dt, stream);
}
int32_t cloudsc2_ad_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                        const float* const* in_adj, const float* eta, float* const* out, float* const* out_adj,
                        double dt, void* stream) {
    return ad_impl<float>("cloudsc2_ad_f32", p, nx, nz, ls, in, in_adj, eta, out, out_adj, dt, stream);
}
int32_t cloudsc2_ad_from_trajectory_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* const* in,
                                        const double* const* in_adj, const double* eta, const double* traj_fplsl,
                                        const double* traj_fplsn, double* const* out_adj, double dt, void* stream) {
    return ad_traj_impl<double>("cloudsc2_ad_from_trajectory_f64", p, nx, nz, ls, in, in_adj, eta, traj_fplsl, traj_fplsn,
                                out_adj, dt, stream);
}
int32_t cloudsc2_ad_from_trajectory_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* const* in,
                                        const float* const* in_adj, const float* eta, const float* traj_fplsl,
                                        const float* traj_fplsn, float* const* out_adj, double dt, void* stream) {
    return ad_traj_impl<float>("cloudsc2_ad_from_trajectory_f32", p, nx, nz, ls, in, in_adj, eta, traj_fplsl, traj_fplsn,
                               out_adj, dt, stream);
}
int32_t cloudsc2_saturation_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const double* ap,
                                const double* t, double* qsat, void* stream) {
    return sat_impl<double>("cloudsc2_saturation_f64", p, nx, nz, ls, ap, t, qsat, stream);
}
int32_t cloudsc2_saturation_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls, const float* ap,
                                const float* t, float* qsat, void* stream) {
    return sat_impl<float>("cloudsc2_saturation_f32", p, nx, nz, ls, ap, t, qsat, stream);
}
int32_t cloudsc2_state_increment_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls,
                                     const double* const* in, double* const* out_i, double f, void* stream) {
    return inc_impl<double>("cloudsc2_state_increment_f64", p, nx, nz, ls, in, out_i, f, stream);
}
int32_t cloudsc2_state_increment_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls,
                                     const float* const* in, float* const* out_i, double f, void* stream) {
    return inc_impl<float>("cloudsc2_state_increment_f32", p, nx, nz, ls, in, out_i, f, stream);
}
int32_t cloudsc2_perturbed_state_f64(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls,
                                     const double* const* in, const double* const* in_i, double* const* out,
                                     double f, void* stream) {
    return per_impl<double>("cloudsc2_perturbed_state_f64", p, nx, nz, ls, in, in_i, out, f, stream);
}
int32_t cloudsc2_perturbed_state_f32(const Cloudsc2Params* p, int32_t nx, int32_t nz, int64_t ls,
                                     const float* const* in, const float* const* in_i, float* const* out,
                                     double f, void* stream) {
    return per_impl<float>("cloudsc2_perturbed_state_f32", p, nx, nz, ls, in, in_i, out, f, stream);
}

}  // extern "C"
