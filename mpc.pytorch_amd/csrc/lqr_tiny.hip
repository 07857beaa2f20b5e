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
    __device__ void gains_visible() const
    {
        __threadfence();
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
    tiny::lqr_step_problem<real, NS>(p, b, p.Kk, L, active);
}

// ---- the whole iLQR solve in one launch (lqr_tiny_body.h: ilqr_iterate_problem) ----------------------------------------
// One wavefront per workgroup; after every iteration the workgroups exchange the two batch-wide words of the stop test
// through `sync` (agent-scope atomics + a spin on the arrival count).  Every workgroup has to be resident for that:
// the launcher refuses grids beyond one wavefront per SIMD of the device.
template <typename real> struct DuBits;
template <> struct DuBits<float> { typedef unsigned int type; };
template <> struct DuBits<double> { typedef unsigned long long type; };

template <typename real, int NS>
__global__ void __launch_bounds__(64) ilqr_env_tiny_kernel(StepParams<real> p, tiny::IlqrArgs<real> a, int G)
{
    typedef typename DuBits<real>::type bits_t;
    const int gid = blockIdx.x * 64 + threadIdx.x;
    int b = gid / G;
    const bool active = b < p.B;
    if (!active) b = p.B - 1;
    GroupLanes L{G, gid % G, (int)(threadIdx.x & 63) & ~(G - 1)};
    // the first nominal: util.get_traj of u_init through the simulator (mpc/mpc.py:251)
    if (active && L.g() == 0) tiny::env_traj_problem<real, NS>(p, b, a.ua, a.xa);
    L.gains_visible();
    int n_not_improved = 0, it = 0;
    for (;;) {
        bool improved = false;
        real du = 0;
        tiny::ilqr_iterate_problem<real, NS>(p, a, b, it, p.Kk, L, active, improved, du);
        L.gains_visible();                                    // the new nominal and the best costs, for the group's other lanes
        // this wavefront's share of the two words: any improvement, the largest ||du|| (a NaN sorts above every number)
        const bool imp = __ballot(improved && active) != 0ull;
        real m = active ? du : (real)0;
        if (!(m == m)) m = (real)NAN;
        bits_t mb;
        __builtin_memcpy(&mb, &m, sizeof(mb));
        mb &= ~((bits_t)1 << (sizeof(bits_t) * 8 - 1));
        for (int off = 32; off > 0; off >>= 1) {
            const bits_t o = __shfl_xor(mb, off);
            mb = o > mb ? o : mb;
        }
        int *sy = a.sync + 4 * it;
        int flag = 0;
        bits_t mx = 0;
        if (threadIdx.x == 0) {
            if (imp) __hip_atomic_fetch_or(sy + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max((bits_t *)(sy + 2), mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sy, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(sy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) __builtin_amdgcn_s_sleep(4);
            flag = __hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mx = __hip_atomic_load((bits_t *)(sy + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        flag = __shfl(flag, 0);
        mx = __shfl(mx, 0);
        real max_du;
        __builtin_memcpy(&max_du, &mx, sizeof(mx));
        if (!tiny::ilqr_continue(it, a.lqr_iter, flag != 0, (double)max_du, (double)a.eps, a.not_improved_lim, n_not_improved)) break;
        ++it;
    }
    if (gid == 0 && a.n_iter_out) *a.n_iter_out = it + 1;
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

template <typename real, int NS> int launch_ilqr_ns(const StepParams<real> &p, const tiny::IlqrArgs<real> &a, hipStream_t st)
{
    const int G = trial_lanes(p.max_ls);
    const long lanes = (long)p.B * G;
    hipLaunchKernelGGL((ilqr_env_tiny_kernel<real, NS>), dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, p, a, G);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(hipGetErrorString(e));
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

bool tiny_supported(int ns, int nc) { return tiny::shape_supported(ns, nc); }

long ilqr_tiny_wavefronts(int B, int max_ls) { return ((long)B * trial_lanes(max_ls) + 63) / 64; }

template <typename real>
int launch_ilqr_env_tiny(const StepParams<real> &p, real *xa, real *ua, real *xb, real *ub, real *best_x, real *best_u,
                         real *best_cost, real *best_du, int lqr_iter, real eps, real best_cost_eps, int not_improved_lim,
                         int *sync, int *n_iter_out, hipStream_t st)
{
    tiny::IlqrArgs<real> a;
    a.xa = xa; a.ua = ua; a.xb = xb; a.ub = ub;
    a.best_x = best_x; a.best_u = best_u; a.best_cost = best_cost; a.best_du = best_du;
    a.lqr_iter = lqr_iter; a.not_improved_lim = not_improved_lim; a.eps = eps; a.best_cost_eps = best_cost_eps;
    a.sync = sync; a.n_iter_out = n_iter_out;
    switch (p.ns) {
    case 1: return launch_ilqr_ns<real, 1>(p, a, st);
    case 2: return launch_ilqr_ns<real, 2>(p, a, st);
    case 3: return launch_ilqr_ns<real, 3>(p, a, st);
    case 4: return launch_ilqr_ns<real, 4>(p, a, st);
    case 5: return launch_ilqr_ns<real, 5>(p, a, st);
    case 6: return launch_ilqr_ns<real, 6>(p, a, st);
    }
    set_last_error("ilqr kernel: n_state out of range");
    return MPC_E_DIMS;
}
template int launch_ilqr_env_tiny<float>(const StepParams<float> &, float *, float *, float *, float *, float *, float *, float *,
                                         float *, int, float, float, int, int *, int *, hipStream_t);
template int launch_ilqr_env_tiny<double>(const StepParams<double> &, double *, double *, double *, double *, double *, double *,
                                          double *, double *, int, double, double, int, int *, int *, hipStream_t);

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
