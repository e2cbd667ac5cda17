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
// CS2_SAT_KEEP_LOADS (r04): with `keep`, in_ap / in_t are LOADED with the default cache policy too.  cloudsc2_nl reads the same
// two fields right after (run_nonlinear.py:117-118); allocated in the 256 MB memory-side cache by this kernel's loads, they
// are served from there to the NL kernel's (non-temporal) LDS-DMAs: inside the step cloudsc2_nl 329.5 -> 314.4 us, the step
// 368.9 -> 349.9 us on plain allocations (profiles/r04/ab_keep_step*.txt).  Making the NL kernel's OWN ap / t DMAs cacheable
// instead does the opposite (+5 %): each DMA carries a second field (aph, supsat) that then sweeps the cache.
#ifndef CS2_SAT_KEEP_LOADS
#define CS2_SAT_KEEP_LOADS 1
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
    if ((keep & 1) && !CS2_SAT_NT_STORE) qsat[i] = r;
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
        if (CS2_SAT_KEEP_LOADS && (keep & 2)) {   // uniform: ap / t stay in the memory-side cache for the cloudsc2_nl that follows
            va[j] = *reinterpret_cast<const vec_t*>(ap + i);
            vt[j] = *reinterpret_cast<const vec_t*>(t + i);
        } else {
            va[j] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(ap + i));
            vt[j] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(t + i));
        }
    }
#pragma unroll
    for (int j = 0; j < kSatLPT; ++j) {
        if (k0 + j >= nz) break;
        vec_t r;
#pragma unroll
        for (int v = 0; v < V; ++v) r[v] = saturation_point<T, MODE>(e, xk, vt[j][v], va[j][v]);
        vec_t* dst = reinterpret_cast<vec_t*>(qsat + int64_t(k0 + j) * ls + int64_t(cv) * V);
        if ((keep & 1) && !CS2_SAT_NT_STORE) *dst = r;
        else __builtin_nontemporal_store(r, dst);
    }
}

template <typename T>
int launch_saturation(const Cloudsc2Params& p, int nx, int nz, int64_t ls, const T* ap, const T* t,
                      T* qsat, hipStream_t stream) {
    const Ext<T> e = make_ext<T>(p);
    const ExpK<T> xk = make_expk<T>();
    constexpr int V = 16 / int(sizeof(T));
    // bit 0: the qsat stores stay cacheable (the field fits beside the streams passing through); bit 1: the ap / t loads too
    // (the three fields cloudsc2_nl shares with this kernel fit the 256 MB memory-side cache together)
    const uint64_t field_bytes = uint64_t(nz + 1) * uint64_t(ls) * sizeof(T);
    const int keep = (qsat_fits_cache<T>(nz, ls) ? 1 : 0) | (3 * field_bytes <= (uint64_t(240) << 20) ? 2 : 0);
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


// ---- validation-norm reductions (build extensions behind cloudsc2_field_sums_* / cloudsc2_column_dots_*) ---------------
// The harnesses of the reference reduce whole fields on the host with NumPy: TaylorTest.get_field_norm forms
// sum(field_nl_p - field_nl) and sum(field_tl) per output field (tangent_linear/validation.py:250-261), SymmetryTest
// forms per-column sums over levels of products of fields (adjoint/validation.py:167-215).  Done with torch those are
// two to three kernels and a temporary per field (200 small launches per Taylor run); here ONE launch handles up to
// kSumFields fields / pairs.  Arithmetic as the reference: the difference (product) is formed in the field type, the
// accumulation is in double.
constexpr int kSumFields = 16;
constexpr int kSumLevels = 18;   // levels per workgroup of field_sums_kernel (138 levels -> 8 chunks)

template <typename T>
__global__ void __launch_bounds__(kAuxBlock)
field_sums_kernel(int nx, int nlev, int64_t ls, int nf, CPtrs<T, kSumFields> a, CPtrs<T, kSumFields> b, int has_b,
                  double* __restrict__ partials) {
    const int col = blockIdx.x * kAuxBlock + threadIdx.x;
    const int k0 = blockIdx.y * kSumLevels;
    const int k1 = k0 + kSumLevels < nlev ? k0 + kSumLevels : nlev;
    double acc[kSumFields];
#pragma unroll
    for (int f = 0; f < kSumFields; ++f) acc[f] = 0.0;
    if (col < nx) {
        for (int k = k0; k < k1; ++k) {
            const int64_t i = int64_t(k) * ls + col;
#pragma unroll
            for (int f = 0; f < kSumFields; ++f)
                if (f < nf) {   // uniform
                    T v = ntload(a.p[f] + i);
                    if (has_b) v = v - ntload(b.p[f] + i);
                    acc[f] += double(v);
                }
        }
    }
    __shared__ double s_red[kAuxBlock / 64][kSumFields];
#pragma unroll
    for (int f = 0; f < kSumFields; ++f) {
        double v = acc[f];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][f] = v;
    }
    __syncthreads();
    if (threadIdx.x < nf) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kAuxBlock / 64; ++w) v += s_red[w][threadIdx.x];
        partials[(size_t(blockIdx.y) * gridDim.x + blockIdx.x) * size_t(nf) + threadIdx.x] = v;
    }
}

int field_sums_blocks(int nx, int nlev) {
    if (nx <= 0 || nlev <= 0) return 0;
    return ((nx + kAuxBlock - 1) / kAuxBlock) * ((nlev + kSumLevels - 1) / kSumLevels);
}

template <typename T>
int launch_field_sums(int nx, int nlev, int64_t ls, int nf, const T* const* a, const T* const* b, double* partials,
                      hipStream_t stream) {
    CPtrs<T, kSumFields> ca, cb;
    for (int i = 0; i < kSumFields; ++i) {
        ca.p[i] = i < nf ? a[i] : nullptr;
        cb.p[i] = (b && i < nf) ? b[i] : nullptr;
    }
    const dim3 grid((nx + kAuxBlock - 1) / kAuxBlock, (nlev + kSumLevels - 1) / kSumLevels), block(kAuxBlock);
    hipLaunchKernelGGL((field_sums_kernel<T>), grid, block, 0, stream, nx, nlev, ls, nf, ca, cb, b ? 1 : 0, partials);
    note_kernel("cs2::field_sums_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// partial[chunk][col] = sum over pairs p and the chunk's kSumLevels levels k of a_p[k][col] * b_p[k][col]  (b == a:
// squares); one lane per column and level chunk, so that a 65 536-column call is 2 048 workgroups with independent loads
// in flight instead of 256 lanes walking 138 levels (first version: 0.86 ms for 0.7 GB; the caller adds the chunks).
template <typename T>
__global__ void __launch_bounds__(kAuxBlock)
column_dots_kernel(int nx, int nlev, int64_t ls, int np, CPtrs<T, kSumFields> a, CPtrs<T, kSumFields> b,
                   double* __restrict__ out, int accumulate) {
    const int col = blockIdx.x * kAuxBlock + threadIdx.x;
    if (col >= nx) return;
    const int k0 = blockIdx.y * kSumLevels;
    const int k1 = k0 + kSumLevels < nlev ? k0 + kSumLevels : nlev;
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) {
        const int64_t i = int64_t(k) * ls + col;
        T va[kSumFields], vb[kSumFields];
#pragma unroll
        for (int f = 0; f < kSumFields; ++f)
            if (f < np) {   // uniform
                va[f] = ntload(a.p[f] + i);
                vb[f] = a.p[f] == b.p[f] ? va[f] : ntload(b.p[f] + i);
            }
#pragma unroll
        for (int f = 0; f < kSumFields; ++f)
            if (f < np) acc += double(va[f]) * double(vb[f]);
    }
    double* o = out + size_t(blockIdx.y) * size_t(nx) + col;
    *o = accumulate ? *o + acc : acc;
}

int column_dots_chunks(int nlev) { return nlev <= 0 ? 0 : (nlev + kSumLevels - 1) / kSumLevels; }

template <typename T>
int launch_column_dots(int nx, int nlev, int64_t ls, int np, const T* const* a, const T* const* b, double* out,
                       int accumulate, hipStream_t stream) {
    CPtrs<T, kSumFields> ca, cb;
    for (int i = 0; i < kSumFields; ++i) {
        ca.p[i] = i < np ? a[i] : nullptr;
        cb.p[i] = i < np ? b[i] : nullptr;
    }
    const dim3 grid((nx + kAuxBlock - 1) / kAuxBlock, column_dots_chunks(nlev)), block(kAuxBlock);
    hipLaunchKernelGGL((column_dots_kernel<T>), grid, block, 0, stream, nx, nlev, ls, np, ca, cb, out, accumulate);
    note_kernel("cs2::column_dots_kernel");
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

#define CS2_INST(T)                                                                                           \
    template int launch_saturation<T>(const Cloudsc2Params&, int, int, int64_t, const T*, const T*, T*,      \
                                      hipStream_t);                                                           \
    template int launch_increment<T>(const Cloudsc2Params&, int, int, int64_t, const T* const*, T* const*,   \
                                     double, hipStream_t);                                                    \
    template int launch_perturb<T>(int, int, int64_t, const T* const*, const T* const*, T* const*, double,   \
                                   hipStream_t);                                                              \
    template int launch_field_sums<T>(int, int, int64_t, int, const T* const*, const T* const*, double*,     \
                                      hipStream_t);                                                           \
    template int launch_column_dots<T>(int, int, int64_t, int, const T* const*, const T* const*, double*,    \
                                       int, hipStream_t);
CS2_INST(double)
CS2_INST(float)
#undef CS2_INST

}  // namespace cs2
