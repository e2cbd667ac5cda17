// Where do the 1024 one-wave workgroups of a column kernel land, depending on the kernel that ran before?
// (dev tool: hipcc -O3 --offload-arch=gfx950 profiles/census_placement.hip -o /tmp/census && /tmp/census)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(64) census(unsigned* out, long long spin) {
    unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
__global__ void fill(double* p, size_t n) {
    size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i < n) p[i] = 1.0;
}
static void report(const char* name, const std::vector<unsigned>& h) {
    std::map<unsigned, int> per_cu, per_xcc, per_simd;
    for (size_t b = 0; b < h.size() / 2; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
        unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
        per_xcc[xcc]++;
        per_simd[(xcc << 20) | (se << 12) | (sh << 8) | (cu << 4) | ((hw >> 4) & 3)]++;
    }
    std::map<int, int> hs;
    for (auto& kv : per_simd) hs[kv.second]++;
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("%-28s CUs used %3zu | waves/CU histogram:", name, per_cu.size());
    for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
    printf(" | SIMDs used %zu, waves/SIMD histogram:", per_simd.size());
    for (auto& kv : hs) printf(" %d:%d", kv.first, kv.second);
    printf(" | per XCC:");
    for (auto& kv : per_xcc) printf(" %d", kv.second);
    printf("\n");
}
int main() {
    unsigned* d; hipMalloc(&d, 2 * 1024 * sizeof(unsigned));
    double* big; size_t n = size_t(64) << 20; hipMalloc(&big, n * 8);
    std::vector<unsigned> h(2048);
    for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, 0, d, 200000LL);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        report("after idle", h);
        hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, big, n);
        hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, 0, d, 200000LL);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        report("after fill (262144 WGs)", h);
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, big, size_t(2048) * 256);
        hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, 0, d, 200000LL);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        report("after fill (2048 WGs)", h);
        hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, 0, d, 200000LL);
        hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, 0, d, 200000LL);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        report("after census (train)", h);
    }
    return 0;
}
