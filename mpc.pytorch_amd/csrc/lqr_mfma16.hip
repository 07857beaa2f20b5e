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
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <typename T> struct vec4_of;
template <> struct vec4_of<float> { typedef f32x4 type; };
template <> struct vec4_of<double> { typedef f64x4 type; };
MPC_DEV int lane() { return (int)threadIdx.x; }
MPC_DEV int problem() { return (int)blockIdx.x; }
MPC_DEV f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// (round 5) v_mfma_f64_16x16x4_f64: lane 16 g + j holds A[i=j][k=g], B[k=g][j] like the float32 instruction, but D[4r+g][j] in element r
// (float32: D[4g+r][j]; measured, tools/ubench/mfma_f64_probe.hip) -- lqr_mfma16_body.h's SLOT_T
MPC_DEV f64x4 mfma(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
MPC_DEV float readlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
MPC_DEV double readlane(double x, int l)
{
    const long long v = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)v, l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
MPC_DEV int readlane_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
MPC_DEV float shfl_xor(float x, int m) { return __shfl_xor(x, m, 64); }
MPC_DEV double shfl_xor(double x, int m) { return __shfl_xor(x, m, 64); }
MPC_DEV float rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);     // one Newton step: <= 1 ulp
}
MPC_DEV double rcp(double x) { return 1.0 / x; }
// an opaque register-to-register identity: keeps hipcc from folding a chain of selects back into scalar mask logic
MPC_DEV void pin(float &x) { asm volatile("" : "+v"(x)); }
MPC_DEV void pin(double &x) { asm volatile("" : "+v"(x)); }
MPC_DEV bool uniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
MPC_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
MPC_DEV unsigned long long ballot(bool c) { return __ballot(c); }
MPC_DEV int ctz64(unsigned long long m) { return __builtin_ctzll(m); }
// ---- HBM -> LDS staging --------------------------------------------------------------------
// One ring of NSTAGE stage buffers per wavefront (= per workgroup); one array per element type, so that the float32 kernels keep
// their 9 KiB (four wavefronts per SIMD) beside the float64 instantiation's 18 KiB.
#define MPC_LDS_BYTES(ES) (4 * 572 * (ES) + 16 * (ES))
__shared__ __attribute__((aligned(16))) char g_stage[MPC_LDS_BYTES(4)];
__shared__ __attribute__((aligned(16))) char g_stage_d[MPC_LDS_BYTES(8)];
template <typename T> MPC_DEV char *stage_of();
template <> MPC_DEV char *stage_of<float>() { return g_stage; }
template <> MPC_DEV char *stage_of<double>() { return g_stage_d; }
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
// 16 (4) bytes per lane from the lane's own global address to LDS[off + 16 (4) * lane]
template <typename T> MPC_DEV void dma16r(const void *g, unsigned off)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(stage_of<T>() + off), 16, 0, 0);
}
template <typename T> MPC_DEV void dma4r(const void *g, unsigned off)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(stage_of<T>() + off), 4, 0, 0);
}
template <typename T> MPC_DEV T lds_r(unsigned off) { return *(const T *)(stage_of<T>() + off); }
template <typename T> MPC_DEV typename vec4_of<T>::type lds_r4(unsigned off);
template <> MPC_DEV f32x4 lds_r4<float>(unsigned off) { return *(const f32x4 *)(g_stage + off); }
// (four doubles as two 16-byte accesses of their own: a 32-byte vector access is split by the compiler, and the halves lose the
// memory operand that tells its waitcnt pass they cannot alias an LDS-DMA in flight -- it then drains the queue, s_waitcnt vmcnt(0),
// in front of each of them, in every timestep: tools/isa_lint.py)
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <> MPC_DEV f64x4 lds_r4<double>(unsigned off)
{
    const f64x2 a = *(const f64x2 *)(g_stage_d + off), b = *(const f64x2 *)(g_stage_d + off + 16);
    return f64x4{a[0], a[1], b[0], b[1]};
}
MPC_DEV void lds_store_r(unsigned off, float v) { *(float *)(g_stage + off) = v; }
MPC_DEV void lds_store_r(unsigned off, double v) { *(double *)(g_stage_d + off) = v; }
MPC_DEV void lds_store_r4(unsigned off, f32x4 v) { *(f32x4 *)(g_stage + off) = v; }
MPC_DEV void lds_store_r4(unsigned off, f64x4 v)
{
    *(f64x2 *)(g_stage_d + off) = f64x2{v[0], v[1]};
    *(f64x2 *)(g_stage_d + off + 16) = f64x2{v[2], v[3]};
}
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
// ... and once more in float64 (namespace mfma16d)
#define MPC_M16_F64
#include "lqr_mfma16_body.h"
#undef MPC_M16_F64

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
// float64 (round 5): twice the registers per value and 18 KiB of LDS per wavefront: two wavefronts per SIMD
template <bool FULL, int MODE>
__global__ void __launch_bounds__(64, 2) lqr_step_mfma16_f64_kernel(StepParams<double> p)
{
    mfma16d::step_problem<FULL, MODE>(p);
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
template <bool FULL>
void launch_mode(const StepParams<double> &p, hipStream_t st)
{
    if (p.bound_mode != MPC_BOUND_NONE)
        hipLaunchKernelGGL((lqr_step_mfma16_f64_kernel<FULL, 2>), dim3(p.B), dim3(64), 0, st, p);
    else if (p.zero_mask)
        hipLaunchKernelGGL((lqr_step_mfma16_f64_kernel<FULL, 1>), dim3(p.B), dim3(64), 0, st, p);
    else
        hipLaunchKernelGGL((lqr_step_mfma16_f64_kernel<FULL, 0>), dim3(p.B), dim3(64), 0, st, p);
}

template <typename real>
int launch_any(const StepParams<real> &p, hipStream_t st)
{
    if (!mfma16_supported(p)) { set_last_error("mfma16: needs n_state <= 12, n_ctrl <= 4, max_linesearch_iter <= 16"); return MPC_E_DIMS; }
    if (!p.Kk) { set_last_error("mfma16: gain workspace missing"); return MPC_E_NULL; }
    if (!p.new_x || !p.new_u) { set_last_error("mfma16: new_x / new_u is NULL"); return MPC_E_NULL; }
    // (the 16-byte staging DMAs of the full 12/4 layout need 16-byte aligned blocks; every other case goes word by word)
    const bool aligned = (((uintptr_t)p.C | (uintptr_t)p.c | (uintptr_t)p.F | (uintptr_t)p.f | (uintptr_t)p.cur_x | (uintptr_t)p.cur_u |
                           (uintptr_t)p.lo | (uintptr_t)p.hi | (uintptr_t)p.Kk) & 15) == 0 &&
                         ((p.C_st | p.C_sb | p.c_st | p.c_sb | p.F_st | p.F_sb | p.f_st | p.f_sb) * (long)sizeof(real)) % 16 == 0;
    if (p.ns == 12 && p.nc == 4 && (sizeof(real) == 4 || aligned))
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

}  // namespace

bool mfma16_supported(const StepParams<float> &p)
{
    return p.ns >= 1 && p.ns <= 12 && p.nc >= 1 && p.nc <= 4 && p.max_ls >= 1 && p.max_ls <= 16 && p.T >= 1;
}
bool mfma16_supported(const StepParams<double> &p)
{
    return p.ns >= 1 && p.ns <= 12 && p.nc >= 1 && p.nc <= 4 && p.max_ls >= 1 && p.max_ls <= 16 && p.T >= 1 && !p.env.kind;
}

int launch_step_mfma16(const StepParams<float> &p, hipStream_t st)
{
    static_assert(MPC_LDS_BYTES(4) == mfma16::LDS_TOTAL && MPC_LDS_BYTES(8) == mfma16d::LDS_TOTAL, "LDS layout out of sync");
    return launch_any(p, st);
}
int launch_step_mfma16(const StepParams<double> &p, hipStream_t st) { return launch_any(p, st); }

}  // namespace mpclqr
