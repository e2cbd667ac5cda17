// Micro-benchmark: does an LDS-staged prefetch ring (LDS-DMA, `global_load_lds_dwordx4`) lift the cloudsc2_nl access
// pattern above what the depth-1 register prefetch reaches?  (dev tool; build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 profiles/microbench_ring.hip -o /tmp/microbench_ring && /tmp/microbench_ring)
//
// Pattern = the NL kernel's: 16 input + 10 output fields [level][column] fp64, 256-thread workgroups, one wave per
// 64 columns walking 137 levels.  Variants:
//   reg<1>  : 16 x 8-byte loads into registers one level ahead (what cloudsc2_nl.hip does), 10 nt stores
//   ring<D> : every wave owns D LDS slots of 8 KB; one `global_load_lds_dwordx4` brings 2 fields x 64 columns
//             (lanes 0-31: field 2i, lanes 32-63: field 2i+1, 2 columns per lane); D-1 levels in flight; counted
//             s_waitcnt vmcnt; the lane then reads its own column of the 16 fields with ds_read_b64.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NF_IN = 16, NF_OUT = 10;
struct Ptrs {
    const double* in[NF_IN];
    double* out[NF_OUT];
};
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ double work(const double* v, int W) {
    double s = 0.0;
#pragma unroll
    for (int f = 0; f < NF_IN; ++f) s += v[f];
    // W dependent fp64 fmas emulate the issue time of the physics between loads and stores
    for (int i = 0; i < W; ++i) s = __builtin_fma(s, 0.999999, 1e-9);
    return s;
}

template <int W>
__global__ void __launch_bounds__(256) reg_kernel(Ptrs p, int nlev, unsigned lsb) {
    unsigned o = (blockIdx.x * 256 + threadIdx.x) * 8u;
    double a[NF_IN], b[NF_IN];
#pragma unroll
    for (int f = 0; f < NF_IN; ++f) a[f] = __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(p.in[f]) + o));
    double acc = 0.0;
    for (int k = 0; k < nlev; ++k) {
        if (k + 1 < nlev) {
#pragma unroll
            for (int f = 0; f < NF_IN; ++f) b[f] = __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(p.in[f]) + o + lsb));
        }
        const double s = work(a, W) + acc;
#pragma unroll
        for (int f = 0; f < NF_OUT; ++f) __builtin_nontemporal_store(s + f, reinterpret_cast<double*>(reinterpret_cast<char*>(p.out[f]) + o));
        acc = s * 1e-3;
#pragma unroll
        for (int f = 0; f < NF_IN; ++f) a[f] = b[f];
        o += lsb;
    }
}

template <int D, int W>
__global__ void __launch_bounds__(256) ring_kernel(Ptrs p, int nlev, unsigned lsb) {
    extern __shared__ __align__(16) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned col0 = blockIdx.x * 256 + wave * 64;
    const int h = lane >> 5, l = lane & 31;
    unsigned o = (col0 + lane) * 8u;                    // this lane's own column (stores)
    // per-lane source pointers: lanes 0-31 walk field 2i, lanes 32-63 field 2i+1, two adjacent columns per lane
    const char* src[NF_IN / 2];
#pragma unroll
    for (int i = 0; i < NF_IN / 2; ++i)
        src[i] = reinterpret_cast<const char*>(h ? p.in[2 * i + 1] : p.in[2 * i]) + (col0 + 2 * l) * 8u;
    const unsigned ringo = unsigned(wave) * (D * 8192);  // byte offset of this wave's ring in LDS
    auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NF_IN / 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                             (lds_ptr_t)(&smem[ringo + slot * 8192 + i * 1024]), 16, 0, 2);
            src[i] += lsb;
        }
    };
#pragma unroll
    for (int k = 0; k < D - 1; ++k) issue(k);
    double acc = 0.0;
    int slot = 0, pslot = D - 1;
    for (int k = 0; k < nlev; ++k) {
        if (k + D - 1 < nlev) {
            issue(pslot);
            // level k's 8 DMAs are older than everything issued since: (D-1) x (8 DMAs + 10 stores)
            __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 1) * 18) & 0xF) | ((((D - 1) * 18) >> 4) << 14));
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // tail: vmcnt(0)
        }
        // LDS reads in inline asm: the compiler must not see them as LDS accesses, or it drains vmcnt(0) before each
        double v[NF_IN];
        const unsigned a = ringo + unsigned(slot) * 8192u + unsigned(lane) * 8u;
        asm volatile(
            "ds_read_b64 %0, %16\n\tds_read_b64 %1, %16 offset:512\n\t"
            "ds_read_b64 %2, %16 offset:1024\n\tds_read_b64 %3, %16 offset:1536\n\t"
            "ds_read_b64 %4, %16 offset:2048\n\tds_read_b64 %5, %16 offset:2560\n\t"
            "ds_read_b64 %6, %16 offset:3072\n\tds_read_b64 %7, %16 offset:3584\n\t"
            "ds_read_b64 %8, %16 offset:4096\n\tds_read_b64 %9, %16 offset:4608\n\t"
            "ds_read_b64 %10, %16 offset:5120\n\tds_read_b64 %11, %16 offset:5632\n\t"
            "ds_read_b64 %12, %16 offset:6144\n\tds_read_b64 %13, %16 offset:6656\n\t"
            "ds_read_b64 %14, %16 offset:7168\n\tds_read_b64 %15, %16 offset:7680\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
              "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]),
              "=&v"(v[15])
            : "v"(a)
            : "memory");
        const double s = work(v, W) + acc;
#pragma unroll
        for (int f = 0; f < NF_OUT; ++f) __builtin_nontemporal_store(s + f, reinterpret_cast<double*>(reinterpret_cast<char*>(p.out[f]) + o));
        acc = s * 1e-3;
        o += lsb;
        slot = slot + 1 == D ? 0 : slot + 1;
        pslot = pslot + 1 == D ? 0 : pslot + 1;
    }
}

__global__ void fill_kernel(double* a, int f, int nx, int nlev1) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < size_t(nx) * nlev1) a[i] = 1e3 * f + double(i / nx) + 1e-6 * double(i % nx);
}

template <typename F>
float timeit(F launch, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

template <int W>
void suite(const Ptrs& p, int nx, int nlev, double bytes) {
    const unsigned lsb = unsigned(nx) * 8u;
    const dim3 grid(nx / 256), block(256);
    auto rep = [&](const char* name, float ms) { printf("  W=%4d %-22s %8.1f us  %7.1f GB/s\n", W, name, ms * 1e3, bytes / (ms * 1e-3) / 1e9); };
    rep("reg depth 1", timeit([&] { hipLaunchKernelGGL((reg_kernel<W>), grid, block, 0, 0, p, nlev, lsb); }, 20));
    rep("ring D=2", timeit([&] { hipLaunchKernelGGL((ring_kernel<2, W>), grid, block, 2 * 32768, 0, p, nlev, lsb); }, 20));
    rep("ring D=3", timeit([&] { hipLaunchKernelGGL((ring_kernel<3, W>), grid, block, 3 * 32768, 0, p, nlev, lsb); }, 20));
    rep("ring D=4", timeit([&] { hipLaunchKernelGGL((ring_kernel<4, W>), grid, block, 4 * 32768, 0, p, nlev, lsb); }, 20));
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 65536, nlev = 137;
    Ptrs p;
    const size_t bytes = size_t(nx) * (nlev + 1) * 8;
    for (int f = 0; f < NF_IN; ++f) {
        hipMalloc((void**)&p.in[f], bytes);
        hipLaunchKernelGGL(fill_kernel, dim3((size_t(nx) * (nlev + 1) + 255) / 256), dim3(256), 0, 0,
                           const_cast<double*>(p.in[f]), f, nx, nlev + 1);
    }
    for (int f = 0; f < NF_OUT; ++f) { hipMalloc((void**)&p.out[f], bytes); hipMemset(p.out[f], 0, bytes); }
    hipFuncSetAttribute((const void*)ring_kernel<3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    hipFuncSetAttribute((const void*)ring_kernel<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    hipFuncSetAttribute((const void*)ring_kernel<3, 600>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    hipFuncSetAttribute((const void*)ring_kernel<4, 600>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    hipFuncSetAttribute((const void*)ring_kernel<3, 1200>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    hipFuncSetAttribute((const void*)ring_kernel<4, 1200>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
    {   // the ring variants must produce exactly what the register variant produces
        const unsigned lsb = unsigned(nx) * 8u;
        const dim3 grid(nx / 256), block(256);
        std::vector<double> ref(size_t(nx) * nlev), got(size_t(nx) * nlev);
        hipLaunchKernelGGL((reg_kernel<0>), grid, block, 0, 0, p, nlev, lsb);
        hipMemcpy(ref.data(), p.out[3], ref.size() * 8, hipMemcpyDeviceToHost);
        auto check = [&](const char* name) {
            hipMemcpy(got.data(), p.out[3], got.size() * 8, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < ref.size(); ++i) bad += got[i] != ref[i];
            printf("  check %-10s %s (%zu mismatches, sample %.6f)\n", name, bad ? "FAILED" : "ok", bad, got[ref.size() / 2]);
            hipMemset(p.out[3], 0, got.size() * 8);
        };
        hipMemset(p.out[3], 0, got.size() * 8);
        hipLaunchKernelGGL((ring_kernel<2, 0>), grid, block, 2 * 32768, 0, p, nlev, lsb); check("ring D=2");
        hipLaunchKernelGGL((ring_kernel<3, 0>), grid, block, 3 * 32768, 0, p, nlev, lsb); check("ring D=3");
        hipLaunchKernelGGL((ring_kernel<4, 0>), grid, block, 4 * 32768, 0, p, nlev, lsb); check("ring D=4");
    }
    const double tot = double(NF_IN + NF_OUT) * nx * nlev * 8;
    printf("%d columns x %d levels, 16 in + 10 out fp64 fields; W = dependent fma per level (issue-time stand-in)\n", nx, nlev);
    suite<0>(p, nx, nlev, tot);
    suite<600>(p, nx, nlev, tot);
    suite<1200>(p, nx, nlev, tot);
    return 0;
}
