// lqr_mfma40.hip -- gfx950 binding of the register-resident MFMA sweep for n_state = 32, n_ctrl = 8
// (lqr_mfma40_body.h): one wavefront per problem and per workgroup, a 3-slot LDS-DMA sweep ring (36 KiB: 4 wavefronts per
// CU) or a 2-slot one (26 KiB: 6 per CU).
#include <string>
#include "lqr_common.h"

#define MPC_DEV __device__ __forceinline__

namespace mpclqr {
namespace wv {
typedef float f32x4 __attribute__((ext_vector_type(4)));
MPC_DEV int lane() { return (int)threadIdx.x; }
MPC_DEV int problem() { return (int)blockIdx.x; }
MPC_DEV unsigned long long clock() { return (unsigned long long)clock64(); }
MPC_DEV f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
MPC_DEV float readlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
MPC_DEV float shfl_xor(float x, int m) { return __shfl_xor(x, m, 64); }
MPC_DEV bool uniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
MPC_DEV bool any(bool c) { return __ballot(c) != 0ull; }
MPC_DEV unsigned long long ballot(bool c) { return __ballot(c); }
// sum over the four 16-lane rows, result in every lane: two row swaps (gfx950 v_permlane{16,32}_swap), no LDS
MPC_DEV float sum_rows(float x)
{
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// rows 0 and 1 (lanes 0..15, 16..31) of x, each copied to all four rows: the same two row swaps
MPC_DEV void rows01(float x, float &r0, float &r1)
{
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);     // rows {0,1,0,1}
    auto b = __builtin_amdgcn_permlane16_swap(a[0], a[0], false, false);
    r0 = __uint_as_float(b[0]);
    r1 = __uint_as_float(b[1]);
}
// lo = {a.row0, b.row0, a.row2, b.row2}, hi = {a.row1, b.row1, a.row3, b.row3} (rows of 16 lanes): v_permlane16_swap
// exchanges the odd rows of its first operand with the even rows of its second
MPC_DEV void swap16(float a, float b, float &lo, float &hi)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
// lo = {a.lanes 0..31, b.lanes 0..31}: v_permlane32_swap exchanges the upper half of its first operand with the lower half of
// its second (the other result, the two upper halves, is not wanted where this is used)
MPC_DEV float lower_halves(float a, float b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]);
}
// DPP row broadcast: lane N of the caller's 16-lane row
template <int N> MPC_DEV float bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + N, 0xf, 0xf, true));
}
// acc += bcast_N(src) * mul (src written long before: no DPP read-after-write wait states needed)
template <int N> MPC_DEV void fmac_bcast_settled(float &acc, float src, float mul)
{
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
}
// the same right behind the instruction that wrote src
template <int N> MPC_DEV void fmac_bcast(float &acc, float src, float mul)
{
    asm("s_nop 1\n"
        "v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(N));
}
// sum over the 16 lanes of a row, in every lane of it
MPC_DEV float row_sum(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));   // row_half_mirror
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true));   // row_mirror
    return x;
}
// the sum over lanes 0..7 of a row, in lanes 0..7 of it (lanes 8..15: their own): the first three of row_sum's four steps --
// for the box QP's vectors, which live in lanes 0..7 and whose sums are read in lane 0 (wv::first_lane)
MPC_DEV float row_sum8(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));   // row_half_mirror
    return x;
}
// lane 0's truth value, as a wave-uniform one (uniform() says "every lane holds this"; this says "lane 0 decides")
MPC_DEV bool first_lane(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
MPC_DEV float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32 as it comes: 1 ulp
MPC_DEV float rcp(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);     // one Newton step: <= 1 ulp
}
// a dword at a wave-uniform address through the scalar cache (s_load_dword: lgkmcnt, not vmcnt); read-only data
MPC_DEV unsigned load_uniform_u32(const unsigned *g)
{
    typedef const __attribute__((address_space(4))) unsigned const_u32_t;
    return *(const_u32_t *)(unsigned long)g;
}
// nothing is scheduled across this point
MPC_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// an opaque register-to-register identity (see mfma40::pick)
MPC_DEV void pin(float &x) { asm volatile("" : "+v"(x)); }
// Compiled three times (Makefile): the step kernels on the three-slot sweep ring (launch_step_mfma40), the same with
// -DMPC_MFMA40_SWEEP_NSTAGE=2 (launch_step_mfma40_ring2), and with -DMPC_MFMA40_KKT the fused KKT backward (three slots).
#ifndef MPC_MFMA40_SWEEP_NSTAGE
#define MPC_MFMA40_SWEEP_NSTAGE 3              // (lqr_mfma40_body.h: why the sweep looks two timesteps ahead)
#endif
// the sweep's slots (12032 B each) or the pricing rollout's two (13056 B), + the layout-turn words; the fused backward's
// second ring (31.5 KiB) lies inside its three sweep slots
#define MPC_MFMA40_LDS_STEP ((MPC_MFMA40_SWEEP_NSTAGE * 12032 > 2 * 13056 ? MPC_MFMA40_SWEEP_NSTAGE * 12032 : 2 * 13056) + 512)
// (the padded fused backward sweeps on two slots: its second ring alone sets the size then)
#if defined(MPC_MFMA40_KKT) && MPC_MFMA40_LDS_STEP < 3 * 10752
#define MPC_MFMA40_LDS (3 * 10752)
#else
#define MPC_MFMA40_LDS MPC_MFMA40_LDS_STEP
#endif
__shared__ __attribute__((aligned(16))) char g_stage40[MPC_MFMA40_LDS];
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
MPC_DEV void dma16(const void *g, unsigned off)
{
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage40 + off), 16, 0, 0);
}
MPC_DEV void dma16_if(bool active, const void *g, unsigned off)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage40 + off), 16, 0, 0);
}
// the same with an immediate offset, which on gfx950 moves source AND destination: instructions of one block share pointer and M0
template <int IMM> MPC_DEV void dma16_at(const void *g, unsigned off)
{
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage40 + off), 16, IMM, 0);
}
template <int IMM> MPC_DEV void dma16_at_if(bool active, const void *g, unsigned off)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage40 + off), 16, IMM, 0);
}
// ... of a block read exactly once per launch: the non-temporal policy (aux bit 1; -DMPC_MFMA40_C_AUX=0 for the A/B)
#ifndef MPC_MFMA40_C_AUX
#define MPC_MFMA40_C_AUX 2
#endif
template <int IMM> MPC_DEV void dma16_once_at(const void *g, unsigned off)
{
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage40 + off), 16, IMM, MPC_MFMA40_C_AUX);
}
// ---- the padded instantiation's staging (lqr_mfma40_body.h, PADK) -------------------------------------------------------------
// G bytes per lane from `base + voff` (base wave-uniform: a raw buffer of `nbytes`, voff per lane) to LDS offset `off` + G * lane.
// A lane whose voff lies beyond the buffer writes ZERO (the hardware's range check; measured, tools/ubench/buffer_lds_probe.hip):
// that is the zero padding of the kernel's 40 x 40 / 32 x 40 blocks, at no instruction.
// a wave-uniform pointer / word the compiler has moved to vector registers, back in scalar ones (two / one v_readfirstlane): the dword build's
// C block address is computed on the vector ALU next to the timestep counter, and a buffer descriptor made from it was then wrapped in a
// readfirstlane LOOP -- per gather, 25 a timestep (round 6)
MPC_DEV const void *uniform_ptr(const void *q)
{
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void *)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
MPC_DEV unsigned uniform_u32(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
template <int G> MPC_DEV void dma_buf(bool active, const void *base, unsigned nbytes, unsigned voff, unsigned off)
{
    static_assert(G == 4 || G == 16, "dword or dwordx4");
    if (active) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)nbytes, 0x00020000);
        if constexpr (G == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(g_stage40 + off), 4, (int)voff, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(g_stage40 + off), 16, (int)voff, 0, 0, 0);
    }
}
// a dword to `base + voff` through a raw buffer of `nbytes`: a lane whose voff lies beyond the buffer stores nothing
MPC_DEV const void *uniform_ptr(const void *q);
MPC_DEV unsigned uniform_u32(unsigned x);
MPC_DEV void st_buf(float *base, unsigned nbytes, unsigned voff, float v)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)nbytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)voff, 0, 0);
}
// ... where `base` is the same for every lane of the wave (a row of one problem and timestep): the address is wave-uniform, but the compiler
// computes it next to the timestep counter on the vector ALU and then wraps the store in a readfirstlane loop -- see uniform_ptr.  (st_buf
// itself stays general: the copy-out of a parked trajectory addresses a different row per lane.)
MPC_DEV void st_buf_u(float *base, unsigned nbytes, unsigned voff, float v)
{
    st_buf((float *)uniform_ptr(base), uniform_u32(nbytes), voff, v);
}
MPC_DEV void dma4_if(bool active, const void *g, unsigned off)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage40 + off), 4, 0, 0);
}
MPC_DEV void lds_store_f32x4(unsigned off, f32x4 v) { *(f32x4 *)(g_stage40 + off) = v; }
MPC_DEV float lds_f32(unsigned off) { return *(const float *)(g_stage40 + off); }
MPC_DEV f32x4 lds_f32x4(unsigned off) { return *(const f32x4 *)(g_stage40 + off); }
MPC_DEV void lds_store_f32(unsigned off, float v) { *(float *)(g_stage40 + off) = v; }
MPC_DEV void store_f32x4(float *g, f32x4 v) { *(f32x4 *)g = v; }
MPC_DEV void fence_own_stores()
{
    // same-CU visibility of this wave's own global stores (the gains) to its later LDS-DMA reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// DS instructions of one wave execute in program order: a compiler barrier is all there is to ask for
MPC_DEV void lds_sync() { asm volatile("" ::: "memory"); }
// hipcc does not order LDS reads behind an LDS-DMA by itself; this is the ordering point
template <int N> MPC_DEV void dma_wait()
{
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
    asm volatile("s_waitcnt vmcnt(%0) ; counted" ::"n"(N) : "memory");
}
// acc = max(acc, |a|, |b|): one v_max3_f32 with source modifiers (fmaxf() costs a canonicalising v_max per operand)
MPC_DEV void absmax3(float &acc, float a, float b)
{
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(acc) : "v"(a), "v"(b));
}
}  // namespace wv
}  // namespace mpclqr

#include "lqr_mfma40_body.h"
static_assert(mpclqr::mfma40::LDS_TOTAL <= MPC_MFMA40_LDS, "the staging array is smaller than the body's rings");

#ifdef MPC_MFMA40_KKT
namespace mpclqr {
namespace {
static_assert(mfma40::KLDS_TOTAL <= MPC_MFMA40_LDS && mfma40::LDS_TOTAL <= MPC_MFMA40_LDS, "the fused KKT kernel's rings do not fit");
// MODE 0: no bounds; 1: controls on a bound pinned in the nested solve
template <int MODE> __global__ void __launch_bounds__(64, 1) lqr_kkt_fused_mfma40_kernel(StepParams<float> p, float *K, float *k,
                                                                                        mfma40::KktArgs40 kx)
{
    mfma40::kkt_fused_wave<MODE>(p, K, k, kx);
}
}  // namespace

#ifdef MPC_MFMA40_PAD
// the PADDED instantiation (round 6; -DMPC_MFMA40_KKT -DMPC_MFMA40_PAD=4 on the two-slot sweep ring, lqr_mfma40_padkkt.o): any n_state <= 32,
// n_ctrl <= 8, dword gathers -- nothing but 4-byte alignment asked of the caller's blocks; the workspace (the library's padded layout) on 16
#if MPC_MFMA40_PAD == 4
bool kkt_fused_mfma40_pad_supported(const StepParams<float> &p, const float *ws)
{
    return p.ns >= 1 && p.ns <= 32 && p.nc >= 1 && p.nc <= 8 && p.T >= 1 && !p.env.kind && ((uintptr_t)ws & 15) == 0 &&
           (p.bound_mode != MPC_BOUND_TENSOR || ((((uintptr_t)p.lo | (uintptr_t)p.hi) & 3) == 0));
}
// ... with 16-byte gathers (lqr_mfma40_pad16kkt.o): rows of C and F and the x | u boundary on 16 bytes -- a quarter of the staging instructions
bool kkt_fused_mfma40_pad16_supported(const StepParams<float> &p, const float *ws)
{
    auto al = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    return kkt_fused_mfma40_pad_supported(p, ws) && p.ns % 4 == 0 && p.nc % 4 == 0 && al(p.C) && (p.T == 1 || al(p.F)) && p.C_st % 4 == 0 &&
           p.C_sb % 4 == 0 && p.F_st % 4 == 0 && p.F_sb % 4 == 0;
}
#define MPC_KF40_LAUNCH launch_kkt_fused_mfma40_pad
#else
#define MPC_KF40_LAUNCH launch_kkt_fused_mfma40_pad16
#endif
#else
#define MPC_KF40_LAUNCH launch_kkt_fused_mfma40
// floats: K [T,B,8,32] | k [T,B,8] | V [T,B,1024] | v,g [T,B,64] | (dx [T,B,32] | du [T,B,8] when the caller keeps none)
int64_t kkt_fused_mfma40_workspace_bytes(int T, int B) { return (int64_t)T * B * (256 + 8 + 1024 + 64 + 40) * 4 + 64; }

bool kkt_fused_mfma40_supported(const StepParams<float> &p, const float *dl_dx, const float *dl_du, const float *dC,
                                const float *dF, const float *ws)
{
    auto al = [](const void *q, long st, long sb) { return ((uintptr_t)q % 16 == 0) && (st % 4 == 0) && (sb % 4 == 0); };
    if (!(p.ns == 32 && p.nc == 8 && p.T >= 1)) return false;
    if (!al(p.C, p.C_st, p.C_sb) || !al(p.c, p.c_st, p.c_sb)) return false;
    if (p.T > 1 && !al(p.F, p.F_st, p.F_sb)) return false;
    if (p.bound_mode == MPC_BOUND_TENSOR && ((((uintptr_t)p.lo | (uintptr_t)p.hi) & 3) != 0)) return false;
    if (p.env.kind) return false;           // (zero_mask / has_delta: the forward's, dropped by the launcher -- mpc/lqr_step.py:322-340)
    if (p.c_st >= (1L << 30) || (long)p.B * 64 >= (1L << 30)) return false;       // (32-bit record steps, lqr_mfma40_body.h: Stream)
    return al(p.cur_x, 0, 0) && al(p.cur_u, 0, 0) && al(dl_dx, 0, 0) && al(dl_du, 0, 0) && al(dC, 0, 0) && al(ws, 0, 0) &&
           (p.T == 1 || al(dF, 0, 0));
}

#endif

// the nested step with lambda and dlambda riding along (one launch), then the outer products (kkt_wave.hip)
int MPC_KF40_LAUNCH(const StepParams<float> &p_in, const float *dl_dx, const float *dl_du, float *dC, float *dc, float *dF,
                            float *df, float *dx_init, float *dx_out, float *du_out, float *ws, float decay, int max_ls,
                            hipStream_t st)
{
    StepParams<float> p = p_in;
    const long TB = (long)p.T * p.B;
    float *K = ws, *k = K + TB * 256, *V = k + TB * 8, *vg = V + TB * 1024, *dx = vg + TB * 64, *du = dx + TB * 32;
    if (dx_out && du_out) { dx = dx_out; du = du_out; }
    p.new_x = dx;
    p.new_u = du;
    p.ls_decay = decay;
    p.max_ls = max_ls;
    p.old_costs = nullptr;
    p.qp_iters = nullptr;
    p.c_symmetric = true;
    p.zero_mask = nullptr;
    p.has_delta = 0;
    mfma40::KktArgs40 kx;
    kx.dl_dx = dl_dx; kx.dl_du = dl_du; kx.dF = dF; kx.df = df; kx.dx_init = dx_init; kx.Vws = V; kx.vgws = vg;
    if (p.bound_mode != MPC_BOUND_NONE) hipLaunchKernelGGL((lqr_kkt_fused_mfma40_kernel<1>), dim3(p.B), dim3(64), 0, st, p, K, k, kx);
    else hipLaunchKernelGGL((lqr_kkt_fused_mfma40_kernel<0>), dim3(p.B), dim3(64), 0, st, p, K, k, kx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_kkt_fused_mfma40_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
#ifdef MPC_KF40_NO_OUTER          // (diagnostic build: the fused kernel without its memory-bound neighbour, tools/ab_cfg5b.py)
    return MPC_OK;
#endif
    return launch_kkt_outer(p, dx, du, dC, dc, dF, st);
}
}  // namespace mpclqr
#else
namespace mpclqr {
namespace {

// MODE: 0 unconstrained, 1 unconstrained + u_zero_I, 2 box-constrained (pnqp8 in the sweep)
template <int MODE> __global__ void __launch_bounds__(64, 1) lqr_step_mfma40_kernel(StepParams<float> p)
{
    mfma40::step_wave<MODE>(p, p.K, p.k);
}

}  // namespace

#ifdef MPC_MFMA40_PAD
// ---- the padded instantiation (two more compilations of this file: -DMPC_MFMA40_PAD=4 / =16, both on the two-slot sweep ring) ----
#if MPC_MFMA40_PAD == 4
// any n_state <= 32, n_ctrl <= 8 in float32: every staging access is a dword, nothing but 4-byte alignment is asked for
bool mfma40_pad_supported(const StepParams<float> &p)
{
    return p.ns >= 1 && p.ns <= 32 && p.nc >= 1 && p.nc <= 8 && p.T >= 1 && p.max_ls >= 1 && p.max_ls <= 16 && !p.env.kind &&
           !(p.bound_mode != MPC_BOUND_NONE && p.zero_mask) &&
           (p.bound_mode != MPC_BOUND_TENSOR || ((((uintptr_t)p.lo | (uintptr_t)p.hi) & 3) == 0));
}
// ... with 16-byte gathers: rows of C and F and the x | u boundary on 16 bytes
bool mfma40_pad16_supported(const StepParams<float> &p)
{
    auto al = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    return p.ns % 4 == 0 && p.nc % 4 == 0 && al(p.C) && (p.T == 1 || al(p.F)) && p.C_st % 4 == 0 && p.C_sb % 4 == 0 &&
           p.F_st % 4 == 0 && p.F_sb % 4 == 0;
}
#define MPC_MFMA40_LAUNCH launch_step_mfma40_pad4
#else
bool mfma40_pad_supported(const StepParams<float> &p);
#define MPC_MFMA40_LAUNCH launch_step_mfma40_pad16
#endif
// p.K / p.k: the kernel's own padded gains [T,B,8,32] / [T,B,8] (16-byte aligned, workspace); p.K_user / p.k_user the caller's
int MPC_MFMA40_LAUNCH(const StepParams<float> &p, hipStream_t st)
{
    if (!mfma40_pad_supported(p)) { set_last_error("mfma40 (padded): needs fp32, n_state <= 32, n_ctrl <= 8, max_linesearch_iter <= 16, no simulator"); return MPC_E_DIMS; }
    if (!p.K || !p.k || (!p.sweep_only && (!p.new_x || !p.new_u))) { set_last_error("mfma40 (padded): K / k / new_x / new_u missing"); return MPC_E_NULL; }
    if (((uintptr_t)p.K & 15) || ((uintptr_t)p.k & 15)) { set_last_error("mfma40 (padded): the gain workspace must be 16-byte aligned"); return MPC_E_ARG; }
    if (p.bound_mode != MPC_BOUND_NONE)
        hipLaunchKernelGGL(lqr_step_mfma40_kernel<2>, dim3(p.B), dim3(64), 0, st, p);
    else if (p.zero_mask)
        hipLaunchKernelGGL(lqr_step_mfma40_kernel<1>, dim3(p.B), dim3(64), 0, st, p);
    else
        hipLaunchKernelGGL(lqr_step_mfma40_kernel<0>, dim3(p.B), dim3(64), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_step_mfma40_kernel (padded): ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}
}  // namespace mpclqr
#else
#if MPC_MFMA40_SWEEP_NSTAGE == 3
bool mfma40_supported(const StepParams<float> &p)
{
    auto al = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    // (the record's per-lane pointers step by 32-bit byte counts, lqr_mfma40_body.h: Stream)
    auto fits = [](long elems) { return elems >= 0 && elems < (1L << 30); };
    return p.ns == 32 && p.nc == 8 && p.T >= 1 && p.max_ls >= 1 && p.max_ls <= 16 && !p.env.kind &&
           fits(p.c_st) && fits(p.f ? p.f_st : 0) && fits((long)p.B * 64) &&
           !(p.bound_mode != MPC_BOUND_NONE && p.zero_mask) &&
           al(p.C) && al(p.c) && (p.T == 1 || al(p.F)) && al(p.cur_x) && al(p.cur_u) && p.C_st % 4 == 0 && p.C_sb % 4 == 0 &&
           p.c_st % 4 == 0 && p.c_sb % 4 == 0 && p.F_st % 4 == 0 && p.F_sb % 4 == 0 &&
           (!p.f || (al(p.f) && p.f_st % 4 == 0 && p.f_sb % 4 == 0)) && al(p.x_init) &&
           // mask bytes and bound rows are read as dwords through the scalar path
           (!p.zero_mask || ((uintptr_t)p.zero_mask & 3) == 0) &&
           (p.bound_mode != MPC_BOUND_TENSOR || ((((uintptr_t)p.lo | (uintptr_t)p.hi) & 3) == 0));
}

#define MPC_MFMA40_LAUNCH launch_step_mfma40
#else
bool mfma40_supported(const StepParams<float> &p);
#define MPC_MFMA40_LAUNCH launch_step_mfma40_ring2
#endif
int MPC_MFMA40_LAUNCH(const StepParams<float> &p, hipStream_t st)
{
    if (!mfma40_supported(p)) { set_last_error("mfma40: needs fp32, n_state = 32, n_ctrl = 8, 16-byte aligned blocks"); return MPC_E_DIMS; }
    if (!p.K || !p.k || (!p.sweep_only && (!p.new_x || !p.new_u))) { set_last_error("mfma40: K / k / new_x / new_u missing"); return MPC_E_NULL; }
    if (((uintptr_t)p.K & 15) || ((uintptr_t)p.k & 15) || ((uintptr_t)p.new_x & 15) || ((uintptr_t)p.new_u & 15)) {
        set_last_error("mfma40: outputs must be 16-byte aligned");
        return MPC_E_ARG;
    }
    if (p.bound_mode != MPC_BOUND_NONE)
        hipLaunchKernelGGL(lqr_step_mfma40_kernel<2>, dim3(p.B), dim3(64), 0, st, p);
    else if (p.zero_mask)
        hipLaunchKernelGGL(lqr_step_mfma40_kernel<1>, dim3(p.B), dim3(64), 0, st, p);
    else
        hipLaunchKernelGGL(lqr_step_mfma40_kernel<0>, dim3(p.B), dim3(64), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_step_mfma40_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

}  // namespace mpclqr
#endif
#endif
