// lqr_tiny.hip -- gfx950 launch of lqr_tiny_body.h: one lane per problem for n_ctrl = 1,
// n_state <= 6 (pendulum / cart-pole iLQR, their slew-augmented variants).
#include "lqr_common.h"
#include "lqr_tiny_body.h"

namespace mpclqr {
namespace {

template <typename real, int NS>
__global__ void __launch_bounds__(64) lqr_step_tiny_kernel(StepParams<real> p)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= p.B) return;
    tiny::lqr_step_problem<real, NS>(p, b, p.Kk);
}

template <typename real, int NS> int launch_ns(const StepParams<real> &p, hipStream_t st)
{
    hipLaunchKernelGGL((lqr_step_tiny_kernel<real, NS>), dim3((p.B + 63) / 64), dim3(64), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(hipGetErrorString(e));
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

}  // namespace

bool tiny_supported(int ns, int nc) { return tiny::shape_supported(ns, nc); }

template <typename real> int launch_step_tiny(const StepParams<real> &p, hipStream_t st)
{
    switch (p.ns) {
    case 1: return launch_ns<real, 1>(p, st);
    case 2: return launch_ns<real, 2>(p, st);
    case 3: return launch_ns<real, 3>(p, st);
    case 4: return launch_ns<real, 4>(p, st);
    case 5: return launch_ns<real, 5>(p, st);
    case 6: return launch_ns<real, 6>(p, st);
    }
    set_last_error("tiny kernel: n_state out of range");
    return MPC_E_DIMS;
}
template int launch_step_tiny<float>(const StepParams<float> &, hipStream_t);
template int launch_step_tiny<double>(const StepParams<double> &, hipStream_t);

}  // namespace mpclqr
