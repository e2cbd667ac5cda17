// Shared device-side definitions for the CLOUDSC2 HIP kernels (gfx950 / CDNA4 only).
//
// Execution model used by every column kernel in this directory:
//   * one lane = one atmospheric column, one wave64 = 64 adjacent columns, workgroup = 4 waves (kColBlock = 256
//     threads: one wave per SIMD of a CU);
//   * fields are [level][column], so every per-level access of a wave is one fully coalesced
//     512-B (fp64) / 256-B (fp32) request;
//   * the vertical loop is sequential per lane; the loop-carried precipitation state lives in
//     registers, level-only quantities (eta, scalm) live in a small LDS table built per workgroup.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "../../include/cloudsc2_hip.h"

namespace cs2 {

constexpr int kWave = 64;  // CDNA wavefront
// Workgroup of the column kernels: 4 waves = one per SIMD of a CU.  One-wave workgroups are placed
// SIMD by SIMD by the dispatcher, and after a kernel with a very large grid (e.g. `saturation`) ~10 % of
// them double up on a SIMD while other SIMDs stay empty (profiles/census_placement.hip), which cost
// cloudsc2_nl +65 us in the driver loop; the 4 waves of a 256-thread workgroup always go to the 4
// different SIMDs of their CU.
#ifndef CS2_COL_BLOCK
#define CS2_COL_BLOCK 256
#endif
constexpr int kColBlock = CS2_COL_BLOCK;

// XCD-aware workgroup -> column-block mapping.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8
// share one, MI355X_MICROARCH.md "Workgroup dispatch"); with the identity mapping every XCD's L2 and fabric port sees
// every eighth 2-KB piece of each field row.  Remapped, the workgroups of one XCD own one contiguous eighth of the
// columns.  Measured on cloudsc2_nl (profiles/r02/ab_xcd_remap.txt, one process): neutral at 65 536 fp64 columns (one
// workgroup per CU, everything in lockstep), -1.4 .. -2.2 % at 262 144 fp64 / 524 288 fp32 columns.  Grids that are not
// a multiple of 8 keep the identity mapping (every block index must stay below gridDim.x).
#ifndef CS2_XCD_REMAP
#define CS2_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_block() {
    const int b = blockIdx.x, n = gridDim.x;
    if (CS2_XCD_REMAP && (n & 7) == 0) return (b & 7) * (n >> 3) + (b >> 3);
    return b;
}

// ---- math in the working precision --------------------------------------------------------
template <typename T> __device__ __forceinline__ T rexp(T x);
template <> __device__ __forceinline__ double rexp<double>(double x) { return exp(x); }
template <> __device__ __forceinline__ float rexp<float>(float x) { return expf(x); }
template <typename T> __device__ __forceinline__ T rtanh(T x);
template <> __device__ __forceinline__ double rtanh<double>(double x) { return tanh(x); }
template <> __device__ __forceinline__ float rtanh<float>(float x) { return tanhf(x); }
template <typename T> __device__ __forceinline__ T rcosh(T x);
template <> __device__ __forceinline__ double rcosh<double>(double x) { return cosh(x); }
template <> __device__ __forceinline__ float rcosh<float>(float x) { return coshf(x); }
template <typename T> __device__ __forceinline__ T rsqrt_(T x);
template <> __device__ __forceinline__ double rsqrt_<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float rsqrt_<float>(float x) { return sqrtf(x); }
template <typename T> __device__ __forceinline__ T rpow(T x, T y);
template <> __device__ __forceinline__ double rpow<double>(double x, double y) { return pow(x, y); }
template <> __device__ __forceinline__ float rpow<float>(float x, float y) { return powf(x, y); }
// Reciprocal by hardware seed + Newton steps (v_rcp_f64 + 4 fma; v_rcp_f32 + 2 fma): <= ~1 ulp, about
// half the instructions of the IEEE-correct division sequence hipcc emits for `a / b`.  The column
// kernels are VALU-issue-bound in fp64 (profiles/), so divisions are expressed as x * frcp(y) and
// reciprocals shared between the expressions that divide by the same quantity.
template <typename T> __device__ __forceinline__ T frcp(T x);
template <> __device__ __forceinline__ double frcp<double>(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
template <> __device__ __forceinline__ float frcp<float>(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    r = fmaf(fmaf(-x, r, 1.0f), r, r);
    return r;
}
// exp() for the level loops.  ocml's exp re-materialises its 11 polynomial coefficients into VGPRs
// on every call (destructive v_fmac form: 2 v_mov per coefficient, ~42 VALU per call); here the
// coefficients travel as kernel arguments (`ExpK`), stay in registers across the loop and the
// evaluation is 19 VALU: n = rint(x log2 e), r = x - n ln2 (two-term Cody-Waite), degree-12 Taylor
// polynomial in Horner form on |r| <= ln2/2 (truncation 3e-17 relative), ldexp.  The argument is
// clamped to [-746, 710] first (exp(-746) = 0, exp(710) = inf in double): the autoconversion terms
// exp(-(cld/crit)^2) reach arguments of -1e20, where the Cody-Waite reduction would lose all bits.
template <typename T>
struct ExpK {
    T l2e, ln2h, ln2l, c12, c11, c10, c9, c8, c7, c6, c5, c4, c3;
};
template <typename T>
inline ExpK<T> make_expk() {
    ExpK<T> k;
    k.l2e = T(1.4426950408889634);
    k.ln2h = T(6.93147180369123816490e-01);
    k.ln2l = T(1.90821492927058770002e-10);
    k.c12 = T(1.0 / 479001600.0); k.c11 = T(1.0 / 39916800.0); k.c10 = T(1.0 / 3628800.0);
    k.c9 = T(1.0 / 362880.0); k.c8 = T(1.0 / 40320.0); k.c7 = T(1.0 / 5040.0); k.c6 = T(1.0 / 720.0);
    k.c5 = T(1.0 / 120.0); k.c4 = T(1.0 / 24.0); k.c3 = T(1.0 / 6.0);
    return k;
}
template <typename T> __device__ __forceinline__ T fexp(const ExpK<T>& k, T x);
template <> __device__ __forceinline__ double fexp<double>(const ExpK<double>& k, double x) {
    {
        // clamp; a NaN argument must stay NaN (v_max / v_min return their non-NaN operand, which would turn exp(NaN)
        // into exp(-746) = 0 where the reference propagates the NaN): its high word is patched back in - 2 VALU
        const double xc = __builtin_fmin(__builtin_fmax(x, -746.0), 710.0);
        int hi = __double2hiint(xc);
        if (x != x) hi = 0x7FF80000;
        x = __hiloint2double(hi, __double2loint(xc));
    }
    const double n = __builtin_rint(x * k.l2e);
    double r = __builtin_fma(-n, k.ln2h, x);
    r = __builtin_fma(-n, k.ln2l, r);
    double p = __builtin_fma(k.c12, r, k.c11);
    p = __builtin_fma(p, r, k.c10);
    p = __builtin_fma(p, r, k.c9);
    p = __builtin_fma(p, r, k.c8);
    p = __builtin_fma(p, r, k.c7);
    p = __builtin_fma(p, r, k.c6);
    p = __builtin_fma(p, r, k.c5);
    p = __builtin_fma(p, r, k.c4);
    p = __builtin_fma(p, r, k.c3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, static_cast<int>(n));
}
template <> __device__ __forceinline__ float fexp<float>(const ExpK<float>&, float x) { return expf(x); }
// A product that is ROUNDED as a product, whatever consumes it: contraction is switched off for this multiply, so the
// backend cannot fold it into a following add / subtract as an fma.  The kernels that fuse `state_increment` in
// (x_i = f * x formed in registers instead of loaded) use it to reproduce the stored products bit for bit.
template <typename T>
__device__ __forceinline__ T rounded_product(T f, T x) {
#pragma clang fp contract(off)
    return f * x;
}
template <typename T> __device__ __forceinline__ T rmin(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ __forceinline__ T rmax(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T sq(T x) { return x * x; }
template <typename T> __device__ __forceinline__ T cube(T x) { return x * x * x; }

// ---- externals in the working precision -----------------------------------------------------
template <typename T>
struct Ext {
    T R2ES, R3IES, R3LES, R4IES, R4LES, R5IES, R5LES;
    T R5ALSCP, R5ALVCP, RALSDCP, RALVDCP;
    T RTICE, RTWAT, RTWAT_RTICE_R, RTICECU, RTWAT_RTICECU_R, RVTMP2;
    T RCPD, RD, RETV, RG, RLMLT, RLSTT, RLVTT, RTT;
    T RCLCRIT, RKCONV, RLMIN, RPECONS, RLPTRC;
    T ZEPS1, ZEPS2, ZQMAX, ZSCAL, QMAX;
    int32_t NLEV;
};

template <typename T>
inline Ext<T> make_ext(const Cloudsc2Params& p) {
    Ext<T> e;
#define CS2_CP(n) e.n = static_cast<T>(p.n)
    CS2_CP(R2ES); CS2_CP(R3IES); CS2_CP(R3LES); CS2_CP(R4IES); CS2_CP(R4LES); CS2_CP(R5IES); CS2_CP(R5LES);
    CS2_CP(R5ALSCP); CS2_CP(R5ALVCP); CS2_CP(RALSDCP); CS2_CP(RALVDCP);
    CS2_CP(RTICE); CS2_CP(RTWAT); CS2_CP(RTWAT_RTICE_R); CS2_CP(RTICECU); CS2_CP(RTWAT_RTICECU_R); CS2_CP(RVTMP2);
    CS2_CP(RCPD); CS2_CP(RD); CS2_CP(RETV); CS2_CP(RG); CS2_CP(RLMLT); CS2_CP(RLSTT); CS2_CP(RLVTT); CS2_CP(RTT);
    CS2_CP(RCLCRIT); CS2_CP(RKCONV); CS2_CP(RLMIN); CS2_CP(RPECONS); CS2_CP(RLPTRC);
    CS2_CP(ZEPS1); CS2_CP(ZEPS2); CS2_CP(ZQMAX); CS2_CP(ZSCAL); CS2_CP(QMAX);
#undef CS2_CP
    e.NLEV = p.NLEV;
    return e;
}

// ---- saturation (common/_stencils/saturation.py:23-42 + fcttre.py:22-57), one point -------------------------
// Shared by saturation_kernel / saturation_vec_kernel (cloudsc2_aux.hip) and the fused-saturation NL variant
// (cloudsc2_nl.hip).  Contraction is switched off inside: every caller then evaluates the literal sequence of
// multiplies and adds below (frcp / fexp spell their fma's out), so the three kernels agree bit for bit whatever
// their surrounding code looks like.  MODE 0: LPHYLIN; 1: not LPHYLIN, KFLAG == 1 (f_foeewmcu); 2: f_foeewm.
template <typename T>
__device__ __forceinline__ T foealfa(const Ext<T>& e, T t) {
#pragma clang fp contract(off)
    return rmin<T>(T(1.0), sq((rmax<T>(e.RTICE, rmin<T>(e.RTWAT, t)) - e.RTICE) * e.RTWAT_RTICE_R));
}
template <typename T>
__device__ __forceinline__ T foealfcu(const Ext<T>& e, T t) {
#pragma clang fp contract(off)
    return rmin<T>(T(1.0), sq((rmax<T>(e.RTICECU, rmin<T>(e.RTWAT, t)) - e.RTICECU) * e.RTWAT_RTICECU_R));
}
template <typename T, int MODE>
__device__ __forceinline__ T saturation_point(const Ext<T>& e, const ExpK<T>& xk, T tt, T app) {
#pragma clang fp contract(off)
    const T rap = frcp<T>(app);
    const T dl = tt - e.R4LES, di = tt - e.R4IES, dtt = tt - e.RTT;
    const T al = e.R3LES * dtt, ai = e.R3IES * dtt;
    const T foeewl = fexp<T>(xk, al * frcp<T>(dl));
    const T foeewi = fexp<T>(xk, ai * frcp<T>(di));
    T qs;
    if constexpr (MODE == 0) {
        const T alfa = foealfa(e, tt);
        const T wl = alfa * (e.R2ES * foeewl);
        const T wi = (T(1.0) - alfa) * (e.R2ES * foeewi);
        qs = rmin<T>((wl + wi) * rap, e.QMAX);
    } else {
        const T alfa = (MODE == 1) ? foealfcu(e, tt) : foealfa(e, tt);
        const T wl = alfa * foeewl;
        const T wi = (T(1.0) - alfa) * foeewi;
        qs = rmin<T>(e.R2ES * (wl + wi) * rap, e.QMAX);
    }
    const T den = T(1.0) - e.RETV * qs;
    return qs * frcp<T>(den);
}

// Pin a wave-uniform value in a VGPR.  The fp64 kernels use ~45 named double constants; together with
// the 26 field pointers that is far more than the 102 SGPRs of a wave, and hipcc then spills SGPRs to
// VGPR lanes and pays two v_readlane per 64-bit constant per use.  A constant that lives in a VGPR
// pair is a plain VALU operand (no extra instruction); at one wave per SIMD there are VGPRs to spare.
template <typename T>
__device__ __forceinline__ void pin_vgpr(T& x) {
    asm volatile("" : "+v"(x));
}

// "This loaded word has landed": the value must be in its register here, so hipcc waits for its load HERE.  Used on
// words loaded before a level loop and first read inside it: a load still pending at the loop header makes hipcc's
// wait-count insertion settle for `s_waitcnt vmcnt(0)` at the first use INSIDE the loop - on every level, i.e. the
// prefetch of the next level is waited for in the middle of the current one (docs/TUNING_LOG.md 3.9).
template <typename T>
__device__ __forceinline__ void landed(T& x) {
    asm volatile("" : "+v"(x));
}

// "Every memory operation of this wave has completed": the level's stores are drained before the next level's loads are
// requested.  Costs the stores' completion latency once per level and buys HBM a cleaner read / write phase structure;
// which one wins is measured per kernel (CS2_*_DRAIN switches, docs/TUNING_LOG.md 3.9).
template <int LEFT = 0>   // LEFT: how many of the youngest operations may stay in flight (0 = a full drain)
__device__ __forceinline__ void drain_vmem() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEFT) : "memory");
}

// Field access by 32-bit BYTE offset from a uniform base pointer: hipcc then emits the
// `global_load_dwordx2 v, v_off, s[base:base+1]` form (no per-access 64-bit VALU address arithmetic).
// The launchers guarantee (nz+1) * lev_stride * sizeof(T) < 2^32.
// CS2_NT (bit 0: loads, bit 1: stores) selects non-temporal accesses: every field element is touched
// exactly once per launch, and streaming reads/writes that do not linger in L2 measured faster on the
// mixed 16-read / 10-write stream pattern (profiles/microbench_stream.hip; A/B in profiles/ab_nl.py).
#ifndef CS2_NT
#define CS2_NT 3
#endif
// The offset type `O` is uint32_t in every kernel but the BIG instantiations (fields of 4 GiB and more, see kBigOffsets):
// there it is uint64_t and hipcc emits the `v[a:a+1], off` form after a 64-bit VALU add per access.
template <typename T, typename O>
__device__ __forceinline__ T ldg(const T* base, O boff) {
    const T* a = reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + boff);
#if CS2_NT & 1
    return __builtin_nontemporal_load(a);
#else
    return *a;
#endif
}
template <typename T, typename O>
__device__ __forceinline__ void stg(T* base, O boff, T v) {
    T* a = reinterpret_cast<T*>(reinterpret_cast<char*>(base) + boff);
#if CS2_NT & 2
    __builtin_nontemporal_store(v, a);
#else
    *a = v;
#endif
}
// Default-policy load for the ONE field with a producer just upstream: `in_qsat` is written by `saturation` right before
// cloudsc2_nl reads it (run_nonlinear.py:117-118) and, at 72 MB for 65 536 columns, is still in the 256 MB
// memory-side cache - unless the store or the load is marked non-temporal (measured: saturation + NL 391 -> 370 us).
template <typename T, typename O>
__device__ __forceinline__ T ldg_keep(const T* base, O boff) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + boff);
}
// streaming access for the pointwise helper kernels (64-bit indexing, same CS2_NT policy)
template <typename T>
__device__ __forceinline__ T ntload(const T* a) {
#if CS2_NT & 1
    return __builtin_nontemporal_load(a);
#else
    return *a;
#endif
}
template <typename T>
__device__ __forceinline__ void ntstore(T* a, T v) {
#if CS2_NT & 2
    __builtin_nontemporal_store(v, a);
#else
    *a = v;
#endif
}
// Producer-consumer residency of `qsat` (saturation -> cloudsc2_nl, run_nonlinear.py:117-118): default cache policy
// on that one store / load pays when the field fits the 256 MB memory-side cache beside the streams passing through
// (72 MB at 65 536 fp64 columns: saturation + NL 391 -> 370 us) and costs 3-4 % when it does not (289 MB at
// 524 288 fp32 columns: 1 598 -> 1 662 us), so both launchers decide with the same rule.
template <typename T>
inline bool qsat_fits_cache(int nz, int64_t ls) {
    return static_cast<uint64_t>(nz + 1) * static_cast<uint64_t>(ls) * sizeof(T) <= (uint64_t(128) << 20);
}
template <typename T>
inline bool fits_u32_offsets(int nz, int64_t ls) {
    return static_cast<uint64_t>(nz + 1) * static_cast<uint64_t>(ls) * sizeof(T) <= 0xFFFFFFFFull;
}
// Fields of 4 GiB and more ((nz+1) * lev_stride * sizeof(T) > 2^32 - 1: 3.89 M fp64 columns at 137 levels): the plain
// stencils (cloudsc2_nl / _tl / _ad) switch to the BIG instantiation of their register-path kernel, whose byte offsets are
// 64-bit; the fused build extensions and the LDS-ring kernels keep 32-bit offsets and refuse such a call
// (CLOUDSC2_E_UNSUPPORTED).  Columns are independent, so a caller can also split such a call into column windows of a
// narrower allocation - but NOT into windows of the same allocation: the level stride, not nx, is what overflows.
template <bool BIG>
using offset_t = typename std::conditional<BIG, uint64_t, uint32_t>::type;

// ---- field pointer bundles (kernel arguments, by value) --------------------------------------
template <typename T, int N>
struct CPtrs { const T* p[N]; };
template <typename T, int N>
struct MPtrs { T* p[N]; };

// Field pointers fetched from the KERNARG SEGMENT at their point of use (cloudsc2_ad).  The AD kernel takes 52 field
// pointers: 104 SGPRs, more than a wave has (102), before the first constant.  Passed as ordinary by-value arguments they
// are all preloaded at kernel entry, and hipcc spills the excess to VGPR lanes and restores each with two v_readlane_b32 -
// VALU issue slots - per memory instruction: 290 of the 1 800 VALU instructions of a cloudsc2_ad level in fp32, 396 of
// 2 386 in fp64 (r04, counted on the ISA).  With the arguments as ONE struct (kernarg offset 0) and this view, a pointer is
// read with a scalar load (s_load_dwordx16: SMEM, no VALU slot) right before the loads / stores that use it and its SGPRs
// are free again afterwards.  `fresh()` makes the compiler forget what it knows about the kernarg pointer, so the scalar
// loads behind `K->...` cannot be hoisted above that point (out of the level loop and back into long-lived SGPRs): call
// it once per level.  Worth -2 ... -4 % on cloudsc2_ad fp32 and 0 ... -1.6 % in fp64 (the kernels wait for HBM, not for
// issue slots: docs/TUNING_LOG.md 3.11); the same change on cloudsc2_tl measured +0.7 % and was not kept.
template <typename ARGS>
struct KernArgs {
    typedef const __attribute__((address_space(4))) ARGS KA;
    KA* ka;
    __device__ __forceinline__ KernArgs() : ka((KA*)__builtin_amdgcn_kernarg_segment_ptr()) {}
    template <bool ON = true>
    __device__ __forceinline__ void fresh() {
        if constexpr (ON) asm volatile("" : "+s"(ka));
    }
    __device__ __forceinline__ KA* operator->() const { return ka; }
};

// diagnostics: the launchers record the name of the kernel they enqueued (cloudsc2_last_kernel(), thread-local)
void note_kernel(const char* name);

// ---- per-device facts the launchers cache (host side) ---------------------------------------------------------
// The header asks for one host thread per device, but nothing here breaks if two threads launch on one device (a
// capture thread beside the main thread): the caches are relaxed atomics, so a reader sees either "not yet known" and
// repeats the (idempotent) query / hipFuncSetAttribute, or the final value - never a torn one.  Devices are indexed
// directly; an ordinal beyond kMaxDevices is refused instead of aliased.
constexpr int kMaxDevices = 64;
inline int current_device(int& dev) {
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return (dev >= 0 && dev < kMaxDevices) ? 0 : -2;
}
inline int device_cus(int dev) {
    static std::atomic<int> cus[kMaxDevices] = {};
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// > 64 KB of dynamic LDS needs an opt-in per kernel function: done once per instantiation (`done` is that
// instantiation's own static array), device and size; sizes only grow.
template <typename Kern>
inline bool lds_opt_in(Kern kern, std::atomic<size_t>* done, int dev, size_t bytes) {
    if (done[dev].load(std::memory_order_relaxed) >= bytes) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(bytes)) != hipSuccess)
        return false;
    size_t cur = done[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !done[dev].compare_exchange_weak(cur, bytes, std::memory_order_relaxed)) {}
    return true;
}

// Per-level LDS table: eta[k] and scalm[k] = ZSCAL * max(eta[k]-0.2, ZEPS1)^0.2
// (nonlinear/_stencils/cloudsc2.py:127).  `pow` is evaluated once per level per workgroup instead
// of once per level per column.  Also returns the tropopause search window [klo, khi]: the levels
// k in [0, nz-2] with 0.1 < eta[k] < 0.4 (cloudsc2.py:109-110); klo > khi when the window is empty.
template <typename T>
__device__ __forceinline__ void build_level_table(const T* __restrict__ eta, int nz, const Ext<T>& e,
                                                  T* s_eta, T* s_scalm, int& klo, int& khi) {
    for (int k = threadIdx.x; k <= nz; k += blockDim.x) {
        T ek = eta[k];
        s_eta[k] = ek;
        s_scalm[k] = e.ZSCAL * rpow<T>(rmax<T>(ek - T(0.2), e.ZEPS1), T(0.2));
    }
    __syncthreads();
    klo = nz;
    khi = -1;
    for (int k = 0; k < nz - 1; ++k) {
        T ek = s_eta[k];
        if (ek > T(0.1) && ek < T(0.4)) {
            if (k < klo) klo = k;
            khi = k;
        }
    }
}

// (line references in this block: nonlinear/_stencils/cloudsc2.py)
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

// Tropopause pre-scan (nonlinear/_stencils/cloudsc2.py:107-111, same statement in the TL and AD stencils): eta of the
// LAST level k of the window [klo, khi] (the levels with 0.1 < eta < 0.4) whose first-guess temperature exceeds that
// of level k+1.  The column's first-guess t of up to 43 levels is needed before the sweep reaches level klo, and
// with one wave per SIMD a load-compare-load loop exposes the full HBM latency once per level (22 us of a 337 us
// kernel, measured).  The loads of CH levels are therefore issued back to back and compared afterwards: 3 round
// trips instead of 42.  (klo, khi) come from build_level_table; khi <= nz - 2, so level khi + 1 exists.
// PERT: the state is read as x + pf * x_i (the fused perturbed_state variants), exactly as nl_perturb forms it.
template <typename T, bool PERT = false, typename O = uint32_t>
__device__ __forceinline__ T trpaus_prescan(const T* __restrict__ pt, const T* __restrict__ ptt, O lsb,
                                            O colb, T dt, const T* s_eta, int klo, int khi,
                                            const T* __restrict__ pt_i = nullptr, const T* __restrict__ ptt_i = nullptr,
                                            T pf = T(0.0)) {
    T trpaus = T(0.1);
    if (klo > khi) return trpaus;
    constexpr int CH = PERT ? 8 : 16;
    const O o0 = O(klo) * lsb + colb;
    T t0 = ldg(pt, o0), tt0 = ldg(ptt, o0);
    if constexpr (PERT) {
        t0 = t0 + pf * ldg(pt_i, o0);
        tt0 = tt0 + pf * ldg(ptt_i, o0);
    }
    T tk = t0 + dt * tt0;
    for (int k0 = klo; k0 <= khi; k0 += CH) {
        T a[CH], b[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kk = (k0 + j < khi ? k0 + j : khi) + 1;   // the tail re-reads level khi + 1 (cache hits)
            const O oj = O(kk) * lsb + colb;
            a[j] = ldg(pt, oj);
            b[j] = ldg(ptt, oj);
            if constexpr (PERT) {
                a[j] = a[j] + pf * ldg(pt_i, oj);
                b[j] = b[j] + pf * ldg(ptt_i, oj);
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int k = k0 + j;
            if (k <= khi) {
                const T tk1 = a[j] + dt * b[j];
                const T ek = s_eta[k];
                if (ek > T(0.1) && ek < T(0.4) && tk > tk1) trpaus = ek;
                tk = tk1;
            }
        }
    }
    return trpaus;
}

// Critical relative humidity profile (nonlinear/_stencils/cloudsc2.py:166-186); rh2/deta1 depend on
// the column's tropopause eta only and are hoisted out of the level loop by the callers.
template <typename T>
struct CrhCol { T trpaus, rh2, bound1, deta1, bound2; };

template <typename T>
__device__ __forceinline__ CrhCol<T> crh_setup(T trpaus) {
    CrhCol<T> c;
    c.trpaus = trpaus;
    c.rh2 = T(0.35) + T(0.14) * sq((trpaus - T(0.25)) / T(0.15)) +
            T(0.04) * rmin<T>(trpaus - T(0.25), T(0.0)) / T(0.15);
    c.bound1 = trpaus + T(0.3);
    c.deta1 = T(0.09) + T(0.16) * (T(0.4) - trpaus) / T(0.3);
    c.bound2 = T(1.0) - c.deta1;
    return c;
}

template <typename T>
__device__ __forceinline__ T crh2_at(const CrhCol<T>& c, T eta) {
    const T rh1 = T(1.0), rh3 = T(1.0);
    if (eta < c.trpaus) return rh3;
    if (eta < c.bound1) return rh3 + (c.rh2 - rh3) * (eta - c.trpaus) / T(0.3);
    if (eta < c.bound2) return c.rh2;
    return rh1 + (c.rh2 - rh1) * rsqrt_<T>((T(1.0) - eta) / c.deta1);
}

// One iteration of the saturation adjustment (nonlinear/_stencils/cuadjtqs.py:24-37).
template <typename T>
__device__ __forceinline__ void cuadjtqs_nl_0(const Ext<T>& e, T ap, T& t, T& q, T z3es, T z4es,
                                              T z5alcp, T zaldcp) {
    T foeew = e.R2ES * rexp<T>(z3es * (t - e.RTT) / (t - z4es));
    T qsat = rmin<T>(foeew / ap, e.ZQMAX);
    T cor = T(1.0) / (T(1.0) - e.RETV * qsat);
    qsat *= cor;
    T z2s = z5alcp / sq(t - z4es);
    T cond = (q - qsat) / (T(1.0) + qsat * cor * z2s);
    t += zaldcp * cond;
    q -= cond;
}

// nonlinear/_stencils/cuadjtqs.py:40-68 (ICALL == 0, the only branch the reference implements).
template <typename T>
__device__ __forceinline__ void cuadjtqs_nl(const Ext<T>& e, T ap, T& t, T& q) {
    T z3es, z4es, z5alcp, zaldcp;
    if (t > e.RTT) {
        z3es = e.R3LES; z4es = e.R4LES; z5alcp = e.R5ALVCP; zaldcp = e.RALVDCP;
    } else {
        z3es = e.R3IES; z4es = e.R4IES; z5alcp = e.R5ALSCP; zaldcp = e.RALSDCP;
    }
    cuadjtqs_nl_0(e, ap, t, q, z3es, z4es, z5alcp, zaldcp);
    cuadjtqs_nl_0(e, ap, t, q, z3es, z4es, z5alcp, zaldcp);
}

}  // namespace cs2
