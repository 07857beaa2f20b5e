// lqr_mfma16.hip -- gfx950 binding of the fused MFMA LQR step (lqr_mfma16_body.h).
//
// One 64-lane wavefront per problem, one workgroup per wavefront, no LDS: at the headline
// shape (n_state=12, n_ctrl=4, T=50, B=4096) that is 4096 wavefronts = 16 per CU = 4 per
// SIMD, all resident at once (__launch_bounds__(64, 4) keeps the kernel at <= 128 VGPRs).
// Every per-timestep block is streamed from HBM exactly once per pass with loads issued two
// timesteps ahead of their use; the matrix work runs on v_mfma_f32_16x16x4_f32.
#include <string>
#include "lqr_common.h"

#define MPC_DEV __device__ __forceinline__

namespace mpclqr {
namespace wv {
typedef float f32x4 __attribute__((ext_vector_type(4)));
MPC_DEV int lane() { return (int)threadIdx.x; }
MPC_DEV int problem() { return (int)blockIdx.x; }
MPC_DEV f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
MPC_DEV float readlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
MPC_DEV int readlane_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
MPC_DEV float shfl_xor(float x, int m) { return __shfl_xor(x, m, 64); }
MPC_DEV float rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);     // one Newton step: <= 1 ulp
}
// an opaque register-to-register identity: keeps hipcc from folding a chain of selects back into scalar mask logic
MPC_DEV void pin(float &x) { asm volatile("" : "+v"(x)); }
MPC_DEV bool uniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
MPC_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
MPC_DEV unsigned long long ballot(bool c) { return __ballot(c); }
MPC_DEV int ctz64(unsigned long long m) { return __builtin_ctzll(m); }
// ---- HBM -> LDS staging --------------------------------------------------------------------
// One ring of NSTAGE stage buffers per wavefront (= per workgroup).
#define MPC_LDS_BYTES (4 * 2288 + 64)
__shared__ __attribute__((aligned(16))) char g_stage[MPC_LDS_BYTES];
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
// 16 (4) bytes per lane from the lane's own global address to LDS[off + 16 (4) * lane]
MPC_DEV void dma16(const void *g, unsigned off)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage + off), 16, 0, 0);
}
MPC_DEV void dma4(const void *g, unsigned off)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage + off), 4, 0, 0);
}
MPC_DEV float lds_f32(unsigned off) { return *(const float *)(g_stage + off); }
MPC_DEV f32x4 lds_f32x4(unsigned off) { return *(const f32x4 *)(g_stage + off); }
MPC_DEV void lds_store_f32(unsigned off, float v) { *(float *)(g_stage + off) = v; }
MPC_DEV void lds_store_f32x4(unsigned off, f32x4 v) { *(f32x4 *)(g_stage + off) = v; }
// LDS traffic between lanes of ONE wave: DS instructions execute in program order, so a compiler
// barrier is all the ordering there is to ask for.
MPC_DEV void lds_sync() { asm volatile("" ::: "memory"); }
// Wait until at most N of this wave's vector-memory operations are outstanding.  hipcc does not
// order LDS reads behind an LDS-DMA by itself; this is the ordering point (and a compiler barrier).
template <int N> MPC_DEV void dma_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; counted" ::"n"(N) : "memory");
}
MPC_DEV void fence_own_stores()
{
    // same-CU visibility of this wave's own global stores to its later loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
}  // namespace wv
}  // namespace mpclqr

#include "lqr_mfma16_body.h"

namespace mpclqr {
namespace {

// MODE: 0 unconstrained, 1 unconstrained + u_zero_I, 2 box-constrained (pnqp in the sweep)
// (the box-constrained instantiation needs ~140 registers: at four waves per SIMD -- 128 -- it spilled 16-21 of them to
//  scratch memory; three waves per SIMD keep it in the register file, tests/test_isa_lint.py)
template <bool FULL, int MODE>
__global__ void __launch_bounds__(64, MODE == 2 ? 3 : 4) lqr_step_mfma16_kernel(StepParams<float> p)
{
    mfma16::step_problem<FULL, MODE>(p);
}

template <bool FULL>
void launch_mode(const StepParams<float> &p, hipStream_t st)
{
    if (p.bound_mode != MPC_BOUND_NONE)
        hipLaunchKernelGGL((lqr_step_mfma16_kernel<FULL, 2>), dim3(p.B), dim3(64), 0, st, p);
    else if (p.zero_mask)
        hipLaunchKernelGGL((lqr_step_mfma16_kernel<FULL, 1>), dim3(p.B), dim3(64), 0, st, p);
    else
        hipLaunchKernelGGL((lqr_step_mfma16_kernel<FULL, 0>), dim3(p.B), dim3(64), 0, st, p);
}

}  // namespace

bool mfma16_supported(const StepParams<float> &p)
{
    return p.ns >= 1 && p.ns <= 12 && p.nc >= 1 && p.nc <= 4 && p.max_ls >= 1 && p.max_ls <= 16 && p.T >= 1;
}

int launch_step_mfma16(const StepParams<float> &p, hipStream_t st)
{
    if (!mfma16_supported(p)) { set_last_error("mfma16: needs n_state <= 12, n_ctrl <= 4, max_linesearch_iter <= 16"); return MPC_E_DIMS; }
    if (!p.Kk) { set_last_error("mfma16: gain workspace missing"); return MPC_E_NULL; }
    static_assert(MPC_LDS_BYTES == mfma16::LDS_TOTAL, "LDS layout out of sync");
    if (!p.new_x || !p.new_u) { set_last_error("mfma16: new_x / new_u is NULL"); return MPC_E_NULL; }
    if (p.ns == 12 && p.nc == 4)
        launch_mode<true>(p, st);
    else
        launch_mode<false>(p, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_step_mfma16_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

}  // namespace mpclqr
