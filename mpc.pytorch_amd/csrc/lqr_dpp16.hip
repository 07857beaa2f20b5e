// lqr_dpp16.hip -- gfx950 binding of the 4-problems-per-wave DPP LQR step (lqr_dpp16_body.h).
//
// One 64-lane wavefront (= one workgroup) owns FOUR problems, one per 16-lane DPP row.  The headline
// shape (n_state=12, n_ctrl=4, T=50, B=4096) is 1024 wavefronts = one per SIMD, each with the whole
// 512-entry VGPR file and 36 KiB of LDS (4 workgroups per CU).  Matrix-matrix products are batched 4x4 outer
// products on the matrix core (mfma4); matrix-vector products are blocks of v_fmac_f32_dpp ... row_newbcast:N
// written as inline asm (hipcc does not fold a DPP mov into v_fmac); each block opens with s_nop 1 because the
// assembler's hazard padding does not look inside asm (VALU write of a VGPR -> DPP read of it needs 2 wait
// states).
#include <string>
#include "lqr_common.h"

#define MPC_DEV __device__ __forceinline__
#define DPPM " row_mask:0xf bank_mask:0xf\n"

namespace mpclqr {
namespace wv {
typedef float f32x4 __attribute__((ext_vector_type(4)));
MPC_DEV int lane() { return (int)threadIdx.x; }
MPC_DEV int problem() { return (int)blockIdx.x; }
MPC_DEV unsigned long long clock() { return (unsigned long long)clock64(); }
MPC_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // 1 ulp
// an opaque register-to-register identity: keeps hipcc from folding a chain of selects back into scalar mask logic
MPC_DEV void pin(float &x) { asm volatile("" : "+v"(x)); }
MPC_DEV bool uniform(bool c) { return c; }
MPC_DEV int uniform(int v) { return v; }
MPC_DEV bool any(bool c) { return __ballot(c) != 0ull; }

// ---- batched 4x4 outer products on the matrix core -------------------------------------------------
// v_mfma_f32_4x4x1_16b_f32 with cbsz=2: sixteen independent 4x4 rank-1 updates, four per 16-lane row; the A
// vector of all four comes from lanes 4*ABID .. 4*ABID+3 of the row, B stays in its lane:
//     d[v] (lane l) = fma(a of lane (l & ~15) + 4*ABID + v,  b of lane l,  c[v])       (one rounding)
// i.e. rows 4*ABID .. +3 of a 16-column outer product whose columns sit one per lane -- the layout every
// matrix of this kernel already has.  One issue slot moves 256 FMAs (a v_fmac_f32_dpp moves 64); measured
// 8.3 clocks per instruction against 5.7 (tools/ubench/).
template <int ABID> MPC_DEV f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, ABID, 0);
}

// nothing is scheduled across this point
MPC_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// ---- DPP row broadcasts ----------------------------------------------------------------------
template <int N> MPC_DEV float bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + N, 0xf, 0xf, true));
}
// acc += bcast_N(src) * mul
template <int N> MPC_DEV void fmac_bcast(float &acc, float src, float mul)
{
    asm("s_nop 1\n"
        "v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3" DPPM
        : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
}
// acc += sum_i bcast_i(src) * mul[i]   (two accumulation chains)
MPC_DEV void dot_bcast16(float &acc, float src, const float (&m)[16])
{
    float t;
    asm("s_nop 1\n"
        "v_mul_f32_dpp %1, %2, %4 row_newbcast:1" DPPM
        "v_fmac_f32_dpp %0, %2, %3 row_newbcast:0" DPPM
        "v_fmac_f32_dpp %1, %2, %6 row_newbcast:3" DPPM
        "v_fmac_f32_dpp %0, %2, %5 row_newbcast:2" DPPM
        "v_fmac_f32_dpp %1, %2, %8 row_newbcast:5" DPPM
        "v_fmac_f32_dpp %0, %2, %7 row_newbcast:4" DPPM
        "v_fmac_f32_dpp %1, %2, %10 row_newbcast:7" DPPM
        "v_fmac_f32_dpp %0, %2, %9 row_newbcast:6" DPPM
        "v_fmac_f32_dpp %1, %2, %12 row_newbcast:9" DPPM
        "v_fmac_f32_dpp %0, %2, %11 row_newbcast:8" DPPM
        "v_fmac_f32_dpp %1, %2, %14 row_newbcast:11" DPPM
        "v_fmac_f32_dpp %0, %2, %13 row_newbcast:10" DPPM
        "v_fmac_f32_dpp %1, %2, %16 row_newbcast:13" DPPM
        "v_fmac_f32_dpp %0, %2, %15 row_newbcast:12" DPPM
        "v_fmac_f32_dpp %1, %2, %18 row_newbcast:15" DPPM
        "v_fmac_f32_dpp %0, %2, %17 row_newbcast:14" DPPM
        "v_add_f32 %0, %0, %1\n"
        : "+v"(acc), "=&v"(t)
        : "v"(src), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]),
          "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]));
}
// acc += sum_b bcast_{12+b}(src) * m[b]   (the four control lanes)
MPC_DEV void dot_bcast_u4(float &acc, float src, const float (&m)[4])
{
    float t;
    asm("s_nop 1\n"
        "v_mul_f32_dpp %1, %2, %4 row_newbcast:13" DPPM
        "v_fmac_f32_dpp %0, %2, %3 row_newbcast:12" DPPM
        "v_fmac_f32_dpp %1, %2, %6 row_newbcast:15" DPPM
        "v_fmac_f32_dpp %0, %2, %5 row_newbcast:14" DPPM
        "v_add_f32 %0, %0, %1\n"
        : "+v"(acc), "=&v"(t)
        : "v"(src), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]));
}
MPC_DEV void dot_bcast12(float &acc, float src, const float (&m)[12])
{
    float t;
    asm("s_nop 1\n"
        "v_mul_f32_dpp %1, %2, %4 row_newbcast:1" DPPM
        "v_fmac_f32_dpp %0, %2, %3 row_newbcast:0" DPPM
        "v_fmac_f32_dpp %1, %2, %6 row_newbcast:3" DPPM
        "v_fmac_f32_dpp %0, %2, %5 row_newbcast:2" DPPM
        "v_fmac_f32_dpp %1, %2, %8 row_newbcast:5" DPPM
        "v_fmac_f32_dpp %0, %2, %7 row_newbcast:4" DPPM
        "v_fmac_f32_dpp %1, %2, %10 row_newbcast:7" DPPM
        "v_fmac_f32_dpp %0, %2, %9 row_newbcast:6" DPPM
        "v_fmac_f32_dpp %1, %2, %12 row_newbcast:9" DPPM
        "v_fmac_f32_dpp %0, %2, %11 row_newbcast:8" DPPM
        "v_fmac_f32_dpp %1, %2, %14 row_newbcast:11" DPPM
        "v_fmac_f32_dpp %0, %2, %13 row_newbcast:10" DPPM
        "v_add_f32 %0, %0, %1\n"
        : "+v"(acc), "=&v"(t)
        : "v"(src), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]),
          "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
}
// sum over the 16 lanes of the row, result in every lane (compiler-visible DPP: hazards handled)
MPC_DEV float row_sum(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));   // row_half_mirror
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true));   // row_mirror
    return x;
}

MPC_DEV double dpp_f64(double x, int) { return x; }
template <int CTRL> MPC_DEV double mov_dpp_f64(double x)
{
    const unsigned long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
MPC_DEV double row_sum_f64(double x)
{
    x += mov_dpp_f64<0xB1>(x);
    x += mov_dpp_f64<0x4E>(x);
    x += mov_dpp_f64<0x141>(x);
    x += mov_dpp_f64<0x140>(x);
    return x;
}

// ---- HBM -> LDS staging --------------------------------------------------------------------
#define MPC_DPP16_LDS (4 * 9216)
__shared__ __attribute__((aligned(16))) char g_stage16[MPC_DPP16_LDS];
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
// cache policy bits of the stage DMAs (see dma16_at).  C is read exactly once per launch: nt (measured
// 129-132 -> 123-124 us).  F in the rollout is its second and last read -- yet nt there is 6 % SLOWER: the first
// part of the rollout finds the blocks the sweep touched last still in the Infinity Cache.
#ifndef MPC_DPP16_C_AUX
#define MPC_DPP16_C_AUX 2
#endif
#ifndef MPC_DPP16_FR_AUX
#define MPC_DPP16_FR_AUX 0
#endif
// results nobody in this launch reads again
MPC_DEV void store_out(float *g, float v)
{
#ifdef MPC_DPP16_OUT_CACHED
    *g = v;
#else
    __builtin_nontemporal_store(v, g);
#endif
}
// KKT kernel: C and F are read exactly once, dC / dF written exactly once -- yet the plain policies win
// (measured: nt loads +10 us, non-temporal stores +90 us), so these stay default.  What did pay is the shape of
// the stores: one register = one row segment of 16 consecutive floats per problem, written as a coalesced dword
// store (184 -> 139 us against four 16-byte stores per lane at a 64-byte stride).
#ifndef MPC_KKT_LD_AUX
#define MPC_KKT_LD_AUX 0
#endif
MPC_DEV void dma16_once(const void *g, unsigned off)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage16 + off), 16, 0, MPC_KKT_LD_AUX);
}
MPC_DEV void store_f32_out(float *g, float v) { *g = v; }
MPC_DEV void dma16_if(bool active, const void *g, unsigned off)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage16 + off), 16, 0, 0);
}
// The immediate offset of an LDS-DMA moves the LDS destination together with the global source
// (tools/ubench/dma_offset_probe.hip).  With the source pointer biased by -IMM once, at set-up, every DMA of a
// stage names the same LDS anchor `mid` and differs only in IMM: one M0 write per stage instead of one per
// instruction.  `g` is the biased pointer (true source - IMM), -4096 <= IMM < 4096.
enum { DMA_PLAIN = 0, DMA_C = 1, DMA_LAST = 2 };
template <int IMM, int KIND = DMA_PLAIN> MPC_DEV void dma16_at(const void *g, unsigned mid)
{
    static_assert(IMM >= -4096 && IMM < 4096, "13-bit signed immediate");
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage16 + mid), 16, IMM,
                                     KIND == DMA_C ? MPC_DPP16_C_AUX : (KIND == DMA_LAST ? MPC_DPP16_FR_AUX : 0));
}
template <int IMM> MPC_DEV void dma16_at_if(bool active, const void *g, unsigned mid)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage16 + mid), 16, IMM, 0);
}
// a dword at a wave-uniform address, through the scalar cache (s_load_dword: lgkmcnt, not vmcnt); the memory is
// read-only for the lifetime of the launch
MPC_DEV unsigned load_uniform_u32(const unsigned *g)
{
    typedef const __attribute__((address_space(4))) unsigned const_u32_t;
    return *(const_u32_t *)(unsigned long)g;
}
MPC_DEV float lds_f32(unsigned off) { return *(const float *)(g_stage16 + off); }
MPC_DEV f32x4 lds_f32x4(unsigned off) { return *(const f32x4 *)(g_stage16 + off); }
MPC_DEV void store_f32x4(float *g, f32x4 v) { *(f32x4 *)g = v; }
template <int N> MPC_DEV void dma_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
    // (the trailing comment marks the wait as written by hand: tools/isa_lint.py looks for the UNMARKED vmcnt(0) the
    // compiler puts in front of a vector load it cannot count across a loop)
    asm volatile("s_waitcnt vmcnt(%0) ; counted" ::"n"(N) : "memory");
}
MPC_DEV void fence_own_stores()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
}  // namespace wv
}  // namespace mpclqr

#include "lqr_dpp16_body.h"

namespace mpclqr {
namespace {

// MODE: 0 unconstrained, 1 unconstrained + u_zero_I, 2 box-constrained (pnqp in the sweep)
template <int MODE>
__global__ void __launch_bounds__(64, 1) lqr_step_dpp16_kernel(StepParams<float> p)
{
    dpp16::step_wave<MODE>(p);
}

__global__ void __launch_bounds__(64, 1) lqr_kkt_dpp16_kernel(StepParams<float> p, dpp16::KktArgs k)
{
    dpp16::kkt_wave(p, k);
}

}  // namespace

bool kkt_dpp16_supported(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx,
                         const float *dC, const float *dF)
{
    auto al = [](const void *q, long st, long sb) { return ((uintptr_t)q % 16 == 0) && (st % 4 == 0) && (sb % 4 == 0); };
    if (!(p.ns == 12 && p.nc == 4 && p.T >= 1)) return false;
    if (!al(p.C, p.C_st, p.C_sb) || !al(p.c, p.c_st, p.c_sb)) return false;
    if (p.T > 1 && !al(p.F, p.F_st, p.F_sb)) return false;
    return al(p.cur_x, 0, 0) && al(p.cur_u, 0, 0) && al(dx, 0, 0) && al(du, 0, 0) && al(dl_dx, 0, 0) && al(dC, 0, 0) &&
           (p.T == 1 || al(dF, 0, 0));
}

int launch_kkt_dpp16(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx, float *dC,
                     float *dc, float *dF, float *df, float *dx_init, hipStream_t st)
{
    dpp16::KktArgs k;
    k.dx = dx; k.du = du; k.dl_dx = dl_dx; k.dC = dC; k.dc = dc; k.dF = dF; k.df = df; k.dx_init = dx_init;
    hipLaunchKernelGGL(lqr_kkt_dpp16_kernel, dim3((p.B + 3) / 4), dim3(64), 0, st, p, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_kkt_dpp16_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

namespace {
}

bool dpp16_supported(const StepParams<float> &p)
{
    // 16-byte DMA granules: every block the kernel streams must start on a 16-byte boundary
    auto al = [](const void *q, long st, long sb) { return ((uintptr_t)q % 16 == 0) && (st % 4 == 0) && (sb % 4 == 0); };
    if (!(p.ns == 12 && p.nc == 4 && p.T >= 1 && p.max_ls >= 1 && p.max_ls <= 16)) return false;
    if (!al(p.C, p.C_st, p.C_sb) || !al(p.c, p.c_st, p.c_sb)) return false;
    if (p.T > 1 && !al(p.F, p.F_st, p.F_sb)) return false;
    if (p.f && !al(p.f, p.f_st, p.f_sb)) return false;
    if (!al(p.cur_x, 0, 0) || !al(p.cur_u, 0, 0)) return false;
    if (p.bound_mode == MPC_BOUND_TENSOR && (!al(p.lo, 0, 0) || !al(p.hi, 0, 0))) return false;
    if (p.zero_mask && (uintptr_t)p.zero_mask % 4 != 0) return false;
    return true;
}

int launch_step_dpp16(const StepParams<float> &p, hipStream_t st)
{
    if (!dpp16_supported(p)) { set_last_error("dpp16: needs n_state = 12, n_ctrl = 4, fp32, 16-byte aligned blocks"); return MPC_E_DIMS; }
    if (!p.Kk || (uintptr_t)p.Kk % 16 != 0) { set_last_error("dpp16: gain workspace missing or misaligned"); return MPC_E_NULL; }
    if (!p.new_x || !p.new_u) { set_last_error("dpp16: new_x / new_u is NULL"); return MPC_E_NULL; }
    static_assert(MPC_DPP16_LDS == dpp16::LDS_TOTAL, "LDS layout out of sync");
    const dim3 grid((p.B + 3) / 4), block(64);
    if (p.bound_mode != MPC_BOUND_NONE) hipLaunchKernelGGL((lqr_step_dpp16_kernel<2>), grid, block, 0, st, p);
    else if (p.zero_mask) hipLaunchKernelGGL((lqr_step_dpp16_kernel<1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((lqr_step_dpp16_kernel<0>), grid, block, 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_step_dpp16_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

}  // namespace mpclqr
