// Micro-benchmark: what HBM bandwidth does the cloudsc2_nl ACCESS PATTERN reach on MI355X, as a
// function of the per-lane access width?  (dev tool; build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 profiles/microbench_stream.hip -o build/microbench_stream && build/microbench_stream)
//
// Pattern: NF_IN input fields + NF_OUT output fields, each [nlev][nx] doubles; one wave64 per 64
// columns walks the levels top to bottom (1024 waves at nx = 65536 = one wave per SIMD).
//   mode 0: every lane loads/stores its own column with 8-byte accesses (dwordx2), 512 B per wave-instr
//   mode 1: loads are 16-byte (dwordx4): half-wave h loads field 2i+h, lane l covers columns 2l, 2l+1
//   mode 2: mode 1 + 16-byte stores in the same arrangement
//   mode 3/4: loads only (8 B / 16 B);  mode 5/6: stores only (8 B / 16 B)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NF_IN = 16, NF_OUT = 10;
struct Ptrs {
    const double* in[NF_IN];
    double* out[NF_OUT];
};

template <int MODE, int NT = 0>
__global__ void __launch_bounds__(64) stream_kernel(Ptrs p, int nx, int nlev, unsigned lsb) {
    const int lane = threadIdx.x;
    const int col0 = blockIdx.x * 64;
    double acc = 0.0;
    if constexpr (MODE == 0 || MODE == 3 || MODE == 5) {
        unsigned o = unsigned(col0 + lane) * 8u;
        for (int k = 0; k < nlev; ++k) {
            double s = acc;
            if constexpr (MODE != 5) {
#pragma unroll
                for (int f = 0; f < NF_IN; ++f) {
                    const double* a = reinterpret_cast<const double*>(reinterpret_cast<const char*>(p.in[f]) + o);
                    s += (NT & 1) ? __builtin_nontemporal_load(a) : *a;
                }
            }
            if constexpr (MODE != 3) {
#pragma unroll
                for (int f = 0; f < NF_OUT; ++f) {
                    double* a = reinterpret_cast<double*>(reinterpret_cast<char*>(p.out[f]) + o);
                    if (NT & 2) __builtin_nontemporal_store(s + f, a); else *a = s + f;
                }
            }
            acc = s * 1e-3;
            o += lsb;
        }
    } else {
        const int h = lane >> 5, l = lane & 31;
        unsigned o = unsigned(col0 + 2 * l) * 8u;
        for (int k = 0; k < nlev; ++k) {
            double s = acc;
            if constexpr (MODE != 6) {
#pragma unroll
                for (int f = 0; f < NF_IN; f += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(h ? p.in[f + 1] : p.in[f]) + o);
                    s += v.x + v.y;
                }
            }
            if constexpr (MODE == 1) {
                const unsigned o1 = unsigned(col0 + lane) * 8u + unsigned(k) * lsb;
#pragma unroll
                for (int f = 0; f < NF_OUT; ++f) *reinterpret_cast<double*>(reinterpret_cast<char*>(p.out[f]) + o1) = s + f;
            } else if constexpr (MODE == 2 || MODE == 6) {
#pragma unroll
                for (int f = 0; f < NF_OUT; f += 2) {
                    double2 v;
                    v.x = s + f;
                    v.y = s - f;
                    *reinterpret_cast<double2*>(reinterpret_cast<char*>(h ? p.out[f + 1] : p.out[f]) + o) = v;
                }
            }
            acc = s * 1e-3;
            o += lsb;
        }
    }
    if (acc == 12345.678) p.out[0][col0 + lane] = acc;  // keep loads alive in load-only modes
}

// Column-blocked layouts: every wave streams contiguous memory.
//   BLK 0: per field [block][level][64 columns]   (26 contiguous 70-KB runs per wave)
//   BLK 1: one arena [block][level][field][64]     (one contiguous 1.8-MB run per wave; `in` and `out` arenas)
template <int BLK>
__global__ void __launch_bounds__(256) blocked_kernel(Ptrs p, int nx, int nlev) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    double acc = 0.0;
    if (BLK == 0) {
        size_t o = size_t(wave) * (nlev + 1) * 64 + lane;
        for (int k = 0; k < nlev; ++k) {
            double s = acc;
#pragma unroll
            for (int f = 0; f < NF_IN; ++f) s += __builtin_nontemporal_load(p.in[f] + o);
#pragma unroll
            for (int f = 0; f < NF_OUT; ++f) __builtin_nontemporal_store(s + f, p.out[f] + o);
            acc = s * 1e-3;
            o += 64;
        }
    } else {
        size_t oi = size_t(wave) * (nlev + 1) * NF_IN * 64 + lane, oo = size_t(wave) * (nlev + 1) * NF_OUT * 64 + lane;
        for (int k = 0; k < nlev; ++k) {
            double s = acc;
#pragma unroll
            for (int f = 0; f < NF_IN; ++f) s += __builtin_nontemporal_load(p.in[0] + oi + f * 64);
#pragma unroll
            for (int f = 0; f < NF_OUT; ++f) __builtin_nontemporal_store(s + f, p.out[0] + oo + f * 64);
            acc = s * 1e-3;
            oi += NF_IN * 64;
            oo += NF_OUT * 64;
        }
    }
    if (acc == 12345.678) p.out[0][lane] = acc;
}
template <int BLK>
float run_blocked(const Ptrs& p, int nx, int nlev, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(blocked_kernel<BLK>, dim3(nx / 256), dim3(256), 0, 0, p, nx, nlev);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(blocked_kernel<BLK>, dim3(nx / 256), dim3(256), 0, 0, p, nx, nlev);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

template <int MODE, int NT = 0>
float run(const Ptrs& p, int nx, int nlev, int iters, unsigned lsb) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<MODE, NT>), dim3(nx / 64), dim3(64), 0, 0, p, nx, nlev, lsb);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((stream_kernel<MODE, NT>), dim3(nx / 64), dim3(64), 0, 0, p, nx, nlev, lsb);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 65536, nlev = 137;
    Ptrs p;
    const size_t bytes = size_t(nx) * (nlev + 1) * 8;
    for (int f = 0; f < NF_IN; ++f) { hipMalloc((void**)&p.in[f], bytes); hipMemset((void*)p.in[f], 0, bytes); }
    for (int f = 0; f < NF_OUT; ++f) { hipMalloc((void**)&p.out[f], bytes); hipMemset(p.out[f], 0, bytes); }
    const double rd = double(NF_IN) * nx * nlev * 8, wr = double(NF_OUT) * nx * nlev * 8;
    struct R { const char* name; float ms; double bytes; };
    {
        const unsigned lsb = unsigned(nx) * 8u;
        R r[7];
        r[0] = {"8B loads + 8B stores  ", run<0>(p, nx, nlev, 20, lsb), rd + wr};
        r[1] = {"16B loads + 8B stores ", run<1>(p, nx, nlev, 20, lsb), rd + wr};
        r[2] = {"16B loads + 16B stores", run<2>(p, nx, nlev, 20, lsb), rd + wr};
        r[3] = {"8B loads only         ", run<3>(p, nx, nlev, 20, lsb), rd};
        r[4] = {"16B loads only        ", run<4>(p, nx, nlev, 20, lsb), rd};
        r[5] = {"8B stores only        ", run<5>(p, nx, nlev, 20, lsb), wr};
        r[6] = {"16B stores only       ", run<6>(p, nx, nlev, 20, lsb), wr};
        printf("separate allocations, [level][column] each:\n");
        R n1 = {"8B nt-loads + 8B stores ", run<0, 1>(p, nx, nlev, 20, lsb), rd + wr};
        R n2 = {"8B loads + 8B nt-stores ", run<0, 2>(p, nx, nlev, 20, lsb), rd + wr};
        R n3 = {"8B nt-loads + nt-stores ", run<0, 3>(p, nx, nlev, 20, lsb), rd + wr};
        for (auto& x : {n1, n2, n3}) printf("  %s  %8.1f us  %7.1f GB/s\n", x.name, x.ms * 1e3, x.bytes / (x.ms * 1e-3) / 1e9);
        for (auto& x : r) printf("  %s  %8.1f us  %7.1f GB/s\n", x.name, x.ms * 1e3, x.bytes / (x.ms * 1e-3) / 1e9);
    }
    {
        // arena layout: inputs interleaved per level [level][field][column], same for outputs
        Ptrs q;
        double *ain, *aout;
        hipMalloc((void**)&ain, bytes * NF_IN);
        hipMalloc((void**)&aout, bytes * NF_OUT);
        hipMemset(ain, 0, bytes * NF_IN);
        hipMemset(aout, 0, bytes * NF_OUT);
        for (int f = 0; f < NF_IN; ++f) q.in[f] = ain + size_t(f) * nx;
        for (int f = 0; f < NF_OUT; ++f) q.out[f] = aout + size_t(f) * nx;
        // two different level strides (inputs vs outputs) are not expressible with one lsb: use the
        // larger one for both, i.e. pad the output arena to NF_IN fields per level
        hipFree(aout);
        hipMalloc((void**)&aout, bytes * NF_IN);
        hipMemset(aout, 0, bytes * NF_IN);
        for (int f = 0; f < NF_OUT; ++f) q.out[f] = aout + size_t(f) * nx;
        const unsigned lsb = unsigned(nx) * 8u * NF_IN;
        R r[4];
        r[0] = {"8B loads + 8B stores  ", run<0>(q, nx, nlev, 20, lsb), rd + wr};
        r[1] = {"16B loads + 16B stores", run<2>(q, nx, nlev, 20, lsb), rd + wr};
        r[2] = {"8B loads only         ", run<3>(q, nx, nlev, 20, lsb), rd};
        r[3] = {"8B stores only        ", run<5>(q, nx, nlev, 20, lsb), wr};
        printf("arena [level][field][column]:\n");
        for (auto& x : r) printf("  %s  %8.1f us  %7.1f GB/s\n", x.name, x.ms * 1e3, x.bytes / (x.ms * 1e-3) / 1e9);
    }
    {
        R r0 = {"blocked per field [blk][lev][64]        ", run_blocked<0>(p, nx, nlev, 20), rd + wr};
        Ptrs q;
        double *ain, *aout;
        hipMalloc((void**)&ain, bytes * NF_IN);
        hipMalloc((void**)&aout, bytes * NF_OUT);
        hipMemset(ain, 0, bytes * NF_IN);
        hipMemset(aout, 0, bytes * NF_OUT);
        for (int f = 0; f < NF_IN; ++f) q.in[f] = ain;
        for (int f = 0; f < NF_OUT; ++f) q.out[f] = aout;
        R r1 = {"blocked arena [blk][lev][field][64]     ", run_blocked<1>(q, nx, nlev, 20), rd + wr};
        printf("column-blocked layouts (256-thread workgroups, nt loads + stores):\n");
        for (auto& x : {r0, r1}) printf("  %s  %8.1f us  %7.1f GB/s\n", x.name, x.ms * 1e3, x.bytes / (x.ms * 1e-3) / 1e9);
    }
    return 0;
}
