// Pointwise helper stencils: saturation, state_increment, perturbed_state.
//   saturation       /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/saturation.py:23-42
//                    (+ f_foealfa / f_foeewm / f_foeewmcu, common/_stencils/fcttre.py:22-57)
//   state_increment  /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/state_increment.py:61-80
//   perturbed_state  /root/reference/src/cloudsc2_gt4py/physics/common/_stencils/perturbed_state.py:75-91
// All three are pure streaming kernels: grid.y walks the levels, grid.x the columns, so each wave
// issues fully coalesced row segments of the [level][column] storage.
#include "cloudsc2_common.hpp"

namespace cs2 {

constexpr int kAuxBlock = 256;
// qsat stores: `keep` != 0 (launcher: the field fits the memory-side cache, qsat_fits_cache) -> default cache policy, the
// consumer (cloudsc2_nl) follows immediately and finds the field there; otherwise non-temporal like every other store.
// CS2_SAT_NT_STORE = 1 forces non-temporal stores (A/B switch).
#ifndef CS2_SAT_NT_STORE
#define CS2_SAT_NT_STORE 0
#endif

// MODE 0: LPHYLIN; MODE 1: not LPHYLIN, KFLAG == 1 (f_foeewmcu); MODE 2: not LPHYLIN, KFLAG != 1 (f_foeewm)
template <typename T, int MODE>
__global__ void __launch_bounds__(kAuxBlock)
saturation_kernel(Ext<T> e, ExpK<T> xk, int nx, int64_t ls, const T* __restrict__ ap, const T* __restrict__ t,
                  T* __restrict__ qsat, int keep) {
    const int col = blockIdx.x * kAuxBlock + threadIdx.x;
    if (col >= nx) return;
    const int64_t i = int64_t(blockIdx.y) * ls + col;
    const T r = saturation_point<T, MODE>(e, xk, ntload(t + i), ntload(ap + i));
    if (keep && !CS2_SAT_NT_STORE) qsat[i] = r;
    else ntstore(qsat + i, r);
}

// Vector form for aligned storages: 16 bytes per lane (2 fp64 / 4 fp32 columns) and LPT levels per thread, all loads
// issued before the first use - 4x (fp64) / 8x (fp32) fewer workgroups and 2 x LPT independent requests per lane
// instead of 2.  Same arithmetic per point as saturation_kernel.
constexpr int kSatLPT = 4;
template <typename T, int MODE>
__global__ void __launch_bounds__(kAuxBlock)
saturation_vec_kernel(Ext<T> e, ExpK<T> xk, int nxv, int nz, int64_t ls, const T* __restrict__ ap, const T* __restrict__ t,
                      T* __restrict__ qsat, int keep) {
    constexpr int V = 16 / int(sizeof(T));
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int cv = blockIdx.x * kAuxBlock + threadIdx.x;   // index of this lane's group of V columns
    if (cv >= nxv) return;
    const int k0 = blockIdx.y * kSatLPT;
    vec_t va[kSatLPT], vt[kSatLPT];
#pragma unroll
    for (int j = 0; j < kSatLPT; ++j) {
        const int k = k0 + j < nz ? k0 + j : nz - 1;
        const int64_t i = int64_t(k) * ls + int64_t(cv) * V;
        va[j] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(ap + i));
        vt[j] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(t + i));
    }
#pragma unroll
    for (int j = 0; j < kSatLPT; ++j) {
        if (k0 + j >= nz) break;
        vec_t r;
#pragma unroll
        for (int v = 0; v < V; ++v) r[v] = saturation_point<T, MODE>(e, xk, vt[j][v], va[j][v]);
        vec_t* dst = reinterpret_cast<vec_t*>(qsat + int64_t(k0 + j) * ls + int64_t(cv) * V);
        if (keep && !CS2_SAT_NT_STORE) *dst = r;
        else __builtin_nontemporal_store(r, dst);
    }
}

template <typename T>
int launch_saturation(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* ap, const T* t,
                      T* qsat, hipStream_t stream) {
    const Ext<T> e = make_ext<T>(p);
    const ExpK<T> xk = make_expk<T>();
    constexpr int V = 16 / int(sizeof(T));
    const int keep = qsat_fits_cache<T>(nz, ls) ? 1 : 0;
    const bool vec = nx % V == 0 && (ls * int64_t(sizeof(T))) % 16 == 0 && reinterpret_cast<uintptr_t>(ap) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(t) % 16 == 0 && reinterpret_cast<uintptr_t>(qsat) % 16 == 0;
    if (vec) {
        const int nxv = nx / V;
        const dim3 vgrid((nxv + kAuxBlock - 1) / kAuxBlock, (nz + kSatLPT - 1) / kSatLPT), vblock(kAuxBlock);
        if (p.LPHYLIN)
            hipLaunchKernelGGL((saturation_vec_kernel<T, 0>), vgrid, vblock, 0, stream, e, xk, nxv, nz, ls, ap, t, qsat, keep);
        else if (p.KFLAG == 1)
            hipLaunchKernelGGL((saturation_vec_kernel<T, 1>), vgrid, vblock, 0, stream, e, xk, nxv, nz, ls, ap, t, qsat, keep);
        else
            hipLaunchKernelGGL((saturation_vec_kernel<T, 2>), vgrid, vblock, 0, stream, e, xk, nxv, nz, ls, ap, t, qsat, keep);
        note_kernel("cs2::saturation_vec_kernel");
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const dim3 grid((nx + kAuxBlock - 1) / kAuxBlock, nz), block(kAuxBlock);
    if (p.LPHYLIN)
        hipLaunchKernelGGL((saturation_kernel<T, 0>), grid, block, 0, stream, e, xk, nx, ls, ap, t, qsat, keep);
    else if (p.KFLAG == 1)
        hipLaunchKernelGGL((saturation_kernel<T, 1>), grid, block, 0, stream, e, xk, nx, ls, ap, t, qsat, keep);
    else
        hipLaunchKernelGGL((saturation_kernel<T, 2>), grid, block, 0, stream, e, xk, nx, ls, ap, t, qsat, keep);
    note_kernel("cs2::saturation_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T>
__global__ void __launch_bounds__(kAuxBlock)
increment_kernel(int nx, int64_t ls, CPtrs<T, INC_NUM> in, MPtrs<T, INC_NUM> out, T f, int ignore_supsat) {
    const int col = blockIdx.x * kAuxBlock + threadIdx.x;
    if (col >= nx) return;
    const int64_t i = int64_t(blockIdx.y) * ls + col;
    T v[INC_NUM];
#pragma unroll
    for (int n = 0; n < INC_NUM; ++n) v[n] = ntload(in.p[n] + i);
#pragma unroll
    for (int n = 0; n < INC_NUM; ++n) {
        T r = f * v[n];
        if (n == INC_SUPSAT && ignore_supsat) r = T(0.0);
        ntstore(out.p[n] + i, r);
    }
}

template <typename T>
int launch_increment(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* const* in,
                     T* const* out, double f, hipStream_t stream) {
    CPtrs<T, INC_NUM> ci;
    MPtrs<T, INC_NUM> co;
    for (int i = 0; i < INC_NUM; ++i) { ci.p[i] = in[i]; co.p[i] = out[i]; }
    const dim3 grid((nx + kAuxBlock - 1) / kAuxBlock, nz + 1), block(kAuxBlock);
    hipLaunchKernelGGL((increment_kernel<T>), grid, block, 0, stream, nx, ls, ci, co, static_cast<T>(f),
                       int(p.IGNORE_SUPSAT));
    note_kernel("cs2::increment_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T>
__global__ void __launch_bounds__(kAuxBlock)
perturb_kernel(int nx, int64_t ls, CPtrs<T, INC_NUM> in, CPtrs<T, INC_NUM> in_i, MPtrs<T, INC_NUM> out, T f) {
    const int col = blockIdx.x * kAuxBlock + threadIdx.x;
    if (col >= nx) return;
    const int64_t i = int64_t(blockIdx.y) * ls + col;
    T a[INC_NUM], b[INC_NUM];
#pragma unroll
    for (int n = 0; n < INC_NUM; ++n) { a[n] = ntload(in.p[n] + i); b[n] = ntload(in_i.p[n] + i); }
#pragma unroll
    for (int n = 0; n < INC_NUM; ++n) ntstore(out.p[n] + i, a[n] + f * b[n]);
}

template <typename T>
int launch_perturb(int nx, int nz, int64_t ls, const T* const* in, const T* const* in_i, T* const* out,
                   double f, hipStream_t stream) {
    CPtrs<T, INC_NUM> ci, cii;
    MPtrs<T, INC_NUM> co;
    for (int i = 0; i < INC_NUM; ++i) { ci.p[i] = in[i]; cii.p[i] = in_i[i]; co.p[i] = out[i]; }
    const dim3 grid((nx + kAuxBlock - 1) / kAuxBlock, nz + 1), block(kAuxBlock);
    hipLaunchKernelGGL((perturb_kernel<T>), grid, block, 0, stream, nx, ls, ci, cii, co, static_cast<T>(f));
    note_kernel("cs2::perturb_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

#define CS2_INST(T)                                                                                           \
    template int launch_saturation<T>(const Cloudsc2Params&, int, int, int64_t, const T*, const T*, T*,      \
                                      hipStream_t);                                                           \
    template int launch_increment<T>(const Cloudsc2Params&, int, int, int64_t, const T* const*, T* const*,   \
                                     double, hipStream_t);                                                    \
    template int launch_perturb<T>(int, int, int64_t, const T* const*, const T* const*, T* const*, double,   \
                                   hipStream_t);
CS2_INST(double)
CS2_INST(float)
#undef CS2_INST

}  // namespace cs2
