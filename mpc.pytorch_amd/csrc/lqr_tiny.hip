// lqr_tiny.hip -- gfx950 launch of lqr_tiny_body.h: one lane per problem for n_ctrl = 1,
// n_state <= 6 (pendulum / cart-pole iLQR, their slew-augmented variants).
#include "lqr_common.h"
#include "lqr_tiny_body.h"

namespace mpclqr {
namespace {

// G aligned lanes of a wavefront share a problem: exchanges are row shuffles, the gains the group's first
// lane wrote to the scratch become visible to the others through an agent-scope fence.
struct GroupLanes {
    int G_, g_, base_;
    __device__ int G() const { return G_; }
    __device__ int g() const { return g_; }
    __device__ void gather(double mine, double *all) const
    {
        for (int i = 0; i < G_; ++i) all[i] = __shfl(mine, base_ + i);
    }
    // The group's lanes sit in one wavefront, the workgroup IS that wavefront: what one lane stored is visible to the
    // others once the stores have left the wavefront (workgroup scope: a wait, no cache maintenance).  An agent-scope
    // fence here writes the XCD's L2 back -- once per wavefront, microseconds each.
    __device__ void gains_visible() const
    {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
};

template <typename real, int NS>
__global__ void __launch_bounds__(64) lqr_step_tiny_kernel(StepParams<real> p, int G)
{
    const int gid = blockIdx.x * 64 + threadIdx.x;
    int b = gid / G;
    const bool active = b < p.B;
    if (!active) b = p.B - 1;                // whole idle groups shadow the last problem, storing nothing
    GroupLanes L{G, gid % G, (int)(threadIdx.x & 63) & ~(G - 1)};
    tiny::lqr_step_problem<real, NS>(p, b, p.Kk, p.Kk + (long)p.T * (NS + 1) * p.B, L, active);
}

// lanes per problem: one per line-search trial of a round, a power of two <= 8
inline int trial_lanes(int max_ls) { return max_ls <= 1 ? 1 : (max_ls == 2 ? 2 : (max_ls <= 4 ? 4 : 8)); }

template <typename real, int NS> int launch_ns(const StepParams<real> &p, hipStream_t st)
{
    const int G = trial_lanes(p.max_ls);
    const long lanes = (long)p.B * G;
    hipLaunchKernelGGL((lqr_step_tiny_kernel<real, NS>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, p, G);
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
