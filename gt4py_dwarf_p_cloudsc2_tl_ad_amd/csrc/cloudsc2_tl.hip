// placeholder until the tl kernel lands (same round); the C-ABI reports CLOUDSC2_E_LAUNCH.
#include "cloudsc2_common.hpp"
namespace cs2 {
template <typename T>
int launch_tl(const Cloudsc2Params&, int, int, int64_t, const T* const*, const T* const*, const T*, T* const*,
              T* const*, double, hipStream_t) { return -1; }
template int launch_tl<double>(const Cloudsc2Params&, int, int, int64_t, const double* const*, const double* const*, const double*, double* const*, double* const*, double, hipStream_t);
template int launch_tl<float>(const Cloudsc2Params&, int, int, int64_t, const float* const*, const float* const*, const float*, float* const*, float* const*, double, hipStream_t);
}
