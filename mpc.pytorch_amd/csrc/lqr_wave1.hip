// lqr_wave1.hip -- gfx950 launch of lqr_wave1_body.h: a 16-lane row per problem, the problem in LDS, for n_ctrl = 1,
// n_state <= 6, float32 (pendulum / cart-pole iLQR at the batch sizes it is run at: the lane-per-problem kernel then
// has a handful of wavefronts on the whole chip and its time is one lane's instruction count).
#include "lqr_common.h"

#define MPC_DEV __device__ __forceinline__

namespace mpclqr {
namespace wv {
extern __shared__ float g_sm[];
MPC_DEV float &sm(int i) { return g_sm[i]; }
MPC_DEV int lane() { return (int)threadIdx.x; }
MPC_DEV int problem() { return (int)blockIdx.x; }
MPC_DEV long long clock() { return (long long)__builtin_readcyclecounter(); }
// lane N of the caller's 16-lane row (DPP row_newbcast)
template <int N> MPC_DEV float bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + N, 0xf, 0xf, true));
}
// acc += bcast_N(src) * mul.  A v_mov_b32_dpp and a v_fmac the compiler schedules among the neighbouring chains and
// whose DPP wait states it places itself: measured 4 % faster here than `s_nop 1; v_fmac_f32_dpp` in an asm block it cannot
// see into (57.8 against 60.5 us per cart-pole step).
template <int N> MPC_DEV void fmac_bcast(float &acc, float src, float mul) { acc = fmaf(bcast<N>(src), mul, acc); }
// sum over the sixteen lanes of the row, the same value in all of them
MPC_DEV double row_sum_f64(double x)
{
    for (int off = 8; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
MPC_DEV double shfl_xor_f64(double x, int off) { return __shfl_xor(x, off); }
MPC_DEV double readlane_f64(double x, int src) { return __shfl(x, src); }
MPC_DEV float readlane(float x, int src) { return __shfl(x, src); }
MPC_DEV unsigned long long ballot(bool c) { return __ballot(c); }
MPC_DEV bool any(bool c) { return __ballot(c) != 0ull; }
MPC_DEV int ctz64(unsigned long long m) { return __builtin_ctzll(m); }
// the workgroup is this one wavefront and its LDS instructions execute in order: the phases only have to be kept apart
MPC_DEV void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
}  // namespace wv
}  // namespace mpclqr

#include "lqr_wave1_body.h"

namespace mpclqr {
namespace {

template <int NS>
__global__ void __launch_bounds__(64) lqr_step_wave1_kernel(StepParams<float> p)
{
    wave1::step_wave<NS>(p);
}

template <int NS> int launch_ns(const StepParams<float> &p, hipStream_t st)
{
    const size_t lds = (size_t)wave1::layout(p).total * 16;      // four problems
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lqr_step_wave1_kernel<NS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((lqr_step_wave1_kernel<NS>), dim3((unsigned)((p.B + 3) / 4)), dim3(64), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(hipGetErrorString(e));
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

}  // namespace

bool wave1_supported(const StepParams<float> &p) { return wave1::shape_supported(p); }
long wave1_lds_bytes(const StepParams<float> &p) { return (long)wave1::layout(p).total * 16; }

int launch_step_wave1(const StepParams<float> &p, hipStream_t st)
{
    switch (p.ns) {
    case 1: return launch_ns<1>(p, st);
    case 2: return launch_ns<2>(p, st);
    case 3: return launch_ns<3>(p, st);
    case 4: return launch_ns<4>(p, st);
    case 5: return launch_ns<5>(p, st);
    case 6: return launch_ns<6>(p, st);
    }
    set_last_error("wave-per-problem kernel: n_state out of range");
    return MPC_E_DIMS;
}

}  // namespace mpclqr
