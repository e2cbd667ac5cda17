// Calling the engine from C++ without Python or PyTorch: the C ABI of include/cloudsc2_hip.h is the whole interface.
//
//   hipcc -O2 --offload-arch=gfx950 examples/nl_from_cpp.cpp -Iinclude \
//         -Lgt4py_dwarf_p_cloudsc2_tl_ad_amd -lcloudsc2_hip -Wl,-rpath,$PWD/gt4py_dwarf_p_cloudsc2_tl_ad_amd -o /tmp/nl_from_cpp
//   /tmp/nl_from_cpp [columns]
//
// Builds an analytic column state on the host (closed-form profiles, so that tests/test_cpp_example.py can rebuild
// the very same numbers in NumPy), copies it to the device, runs `saturation` + `cloudsc2_nl` on a HIP stream and
// prints one checksum per output field.  The parameter values are the build's provisional set
// (gt4py_dwarf_p_cloudsc2_tl_ad_amd/params.py), passed on the command line by the test as "NAME=value" pairs so
// that this file carries no second copy of them.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cloudsc2_hip.h"

#define HIP_CHECK(x)                                                                     \
    do {                                                                                 \
        hipError_t err_ = (x);                                                           \
        if (err_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_));               \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

static bool set_param(Cloudsc2Params& p, const std::string& name, double v) {
#define D(n) if (name == #n) { p.n = v; return true; }
#define I(n) if (name == #n) { p.n = static_cast<int32_t>(v); return true; }
    D(R2ES) D(R3IES) D(R3LES) D(R4IES) D(R4LES) D(R5IES) D(R5LES) D(R5ALSCP) D(R5ALVCP) D(RALSDCP) D(RALVDCP)
    D(RTICE) D(RTWAT) D(RTWAT_RTICE_R) D(RTICECU) D(RTWAT_RTICECU_R) D(RVTMP2) D(RCPD) D(RD) D(RETV) D(RG) D(RLMLT)
    D(RLSTT) D(RLVTT) D(RTT) D(RCLCRIT) D(RKCONV) D(RLMIN) D(RPECONS) D(RLPTRC) D(ZEPS1) D(ZEPS2) D(ZQMAX) D(ZSCAL)
    D(QMAX) I(LPHYLIN) I(LDRAIN1D) I(LEVAPLS2) I(LREGCL) I(ICALL) I(KFLAG) I(IGNORE_SUPSAT) I(NLEV) I(AD_TRAJ_FIX)
#undef D
#undef I
    return false;
}

int main(int argc, char** argv) {
    int nx = 256;
    const int nz = 137;
    Cloudsc2Params p;
    std::memset(&p, 0, sizeof p);
    for (int i = 1; i < argc; ++i) {
        const char* eq = std::strchr(argv[i], '=');
        if (!eq) { nx = std::atoi(argv[i]); continue; }
        if (!set_param(p, std::string(argv[i], eq - argv[i]), std::atof(eq + 1))) {
            std::fprintf(stderr, "unknown parameter %s\n", argv[i]);
            return 2;
        }
    }
    p.NLEV = nz;
    if (cloudsc2_abi_version() != CLOUDSC2_ABI_VERSION || cloudsc2_params_sizeof() != int32_t(sizeof(Cloudsc2Params))) {
        std::fprintf(stderr, "header / library mismatch\n");
        return 2;
    }
    // ---- analytic state, [level][column], nz+1 levels per field (same formulas in tests/test_cpp_example.py)
    const size_t n = size_t(nz + 1) * nx;
    std::vector<std::vector<double>> h(NL_NUM_IN, std::vector<double>(n, 0.0));
    std::vector<double> eta(nz + 1, 0.0);
    for (int k = 0; k <= nz; ++k) {
        const double sh = double(k) / nz;                       // half-level sigma: 0 at the top, 1 at the surface
        const double sf = (k + 0.5) / nz;                       // full-level sigma
        if (k < nz) eta[k] = sf;
        for (int c = 0; c < nx; ++c) {
            const size_t i = size_t(k) * nx + c;
            const double x = double(c) / nx;
            const double ps = 98000.0 + 4000.0 * x;
            h[NL_IN_APH][i] = ps * sh * sh * (3.0 - 2.0 * sh);   // smooth, strictly increasing, 0 at the top
            if (k == nz) continue;                               // padding level of the full-level fields stays 0
            const double ap = ps * sf * sf * (3.0 - 2.0 * sf) + 1.0;
            const double t = 215.0 + 75.0 * sf * sf + 6.0 * std::sin(7.0 * x + 3.0 * sf) - 8.0 * x;
            h[NL_IN_AP][i] = ap;
            h[NL_IN_T][i] = t;
            const double es = 611.21 * std::exp(17.502 * (t - 273.16) / (t - 32.19));
            const double rh = 0.35 + 0.75 * std::pow(std::sin(5.0 * x + 4.0 * sf), 2.0);
            h[NL_IN_Q][i] = rh * 0.622 * es / ap;
            h[NL_IN_QL][i] = (k % 5 == 0) ? 2e-5 * x : 0.0;
            h[NL_IN_QI][i] = (k % 7 == 0) ? 1e-5 * (1.0 - x) : 0.0;
            h[NL_IN_LUDE][i] = (k % 11 == 3) ? 1e-6 * x : 0.0;
            h[NL_IN_LU][i] = (k % 11 == 4) ? 1e-4 * x : 0.0;    // lu[k+1] pairs with lude[k]
            h[NL_IN_MFU][i] = 0.01 * sf * x;
            h[NL_IN_MFD][i] = -0.005 * sf * (1.0 - x);
            h[NL_IN_SUPSAT][i] = 0.0;
            h[NL_IN_TND_CML_T][i] = 1e-5 * std::sin(9.0 * x + sf);
            h[NL_IN_TND_CML_Q][i] = 1e-9 * std::cos(4.0 * x + 2.0 * sf);
            h[NL_IN_TND_CML_QL][i] = 0.0;
            h[NL_IN_TND_CML_QI][i] = 0.0;
        }
    }
    // ---- device buffers
    std::vector<double*> d_in(NL_NUM_IN), d_out(NL_NUM_OUT);
    double* d_eta;
    for (int f = 0; f < NL_NUM_IN; ++f) {
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_in[f]), n * sizeof(double)));
        HIP_CHECK(hipMemcpy(d_in[f], h[f].data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    for (int f = 0; f < NL_NUM_OUT; ++f) {
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_out[f]), n * sizeof(double)));
        HIP_CHECK(hipMemset(d_out[f], 0, n * sizeof(double)));
    }
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_eta), (nz + 1) * sizeof(double)));
    HIP_CHECK(hipMemcpy(d_eta, eta.data(), (nz + 1) * sizeof(double), hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP_CHECK(hipStreamCreate(&stream));
    // ---- the two calls of the reference driver's timed region
    int rc = cloudsc2_saturation_f64(&p, nx, nz, nx, d_in[NL_IN_AP], d_in[NL_IN_T], d_in[NL_IN_QSAT], stream);
    if (rc == CLOUDSC2_OK)
        rc = cloudsc2_nl_f64(&p, nx, nz, nx, d_in.data(), d_eta, d_out.data(), 3600.0, stream);
    if (rc != CLOUDSC2_OK) {
        std::fprintf(stderr, "cloudsc2 call failed (%d): %s\n", rc, cloudsc2_last_error());
        return 1;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    static const char* names[NL_NUM_OUT] = {"clc", "covptot", "fhpsl", "fhpsn", "fplsl", "fplsn", "tnd_q", "tnd_qi", "tnd_ql", "tnd_t"};
    std::vector<double> out(n);
    for (int f = 0; f < NL_NUM_OUT; ++f) {
        HIP_CHECK(hipMemcpy(out.data(), d_out[f], n * sizeof(double), hipMemcpyDeviceToHost));
        double sum = 0.0, asum = 0.0;
        for (size_t i = 0; i < n; ++i) { sum += out[i]; asum += std::fabs(out[i]); }
        std::printf("%s %.17e %.17e\n", names[f], sum, asum);
    }
    return 0;
}
