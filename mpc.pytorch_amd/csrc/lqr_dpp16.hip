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
MPC_DEV unsigned long long ballot(bool c) { return __ballot(c); }
// the value of lane l (wave-uniform l)
MPC_DEV float readlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), __builtin_amdgcn_readfirstlane(l))); }
MPC_DEV double readlane_f64(double x, int l)
{
    const unsigned long long b = __double_as_longlong(x);
    const int ll = __builtin_amdgcn_readfirstlane(l);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, ll), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), ll);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

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
// the same where src was written at least three instructions earlier (or has just been read through DPP by the instruction in
// front): no read-after-write wait states -- each s_nop is an issue slot on a chain that has nothing to fill it with (round 4)
template <int N> MPC_DEV void fmac_bcast_settled(float &acc, float src, float mul)
{
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3" DPPM
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

// max over the 16 lanes of the row, result in every lane (x >= 0, no NaN handling wanted)
MPC_DEV float row_max(float x)
{
    x = __builtin_fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)));
    x = __builtin_fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true)));
    x = __builtin_fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true)));
    x = __builtin_fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true)));
    return x;
}
// acc = max(acc, |a|, |b|): one v_max3_f32 with source modifiers (fmaxf() costs a canonicalising v_max per operand)
MPC_DEV void absmax3(float &acc, float a, float b)
{
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(acc) : "v"(a), "v"(b));
}

// Four row sums in one go: lane j of a row comes back with sum over the row's 16 lanes of p_a, a = j & 3.
// Two exchanges inside the quads (each lane keeps the addend of "its" a and hands over the other), then the four quads
// of the row are added with two rotations: 6 selects + 5 DPP adds (four separate row_sum calls: 16 DPP adds).
MPC_DEV float quad_sums(float p0, float p1, float p2, float p3, int j)
{
    const bool o1 = (j & 1) != 0, o2 = (j & 2) != 0;
    const float t0 = o1 ? p1 : p0, t1 = o1 ? p0 : p1, t2 = o1 ? p3 : p2, t3 = o1 ? p2 : p3;
    const float r01 = t0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t1), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    const float r23 = t2 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t3), 0xB1, 0xf, 0xf, true));
    const float u0 = o2 ? r23 : r01, u1 = o2 ? r01 : r23;
    float r = u0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u1), 0x4E, 0xf, 0xf, true));            // quad_perm [2,3,0,1]
    r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x124, 0xf, 0xf, true));                      // row_ror:4
    r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x128, 0xf, 0xf, true));                      // row_ror:8
    return r;
}

// Sum over the four quads of a row (lanes j, j+4, j+8, j+12 of the same row): two row rotations.
MPC_DEV float ring_sum(float r)
{
    r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x124, 0xf, 0xf, true));                      // row_ror:4
    r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x128, 0xf, 0xf, true));                      // row_ror:8
    return r;
}

// ---- the gains of the whole horizon in the accumulation half of the register file (mode 0) ----------------------
// a[4t .. 4t+3] = the four gain registers of timestep t.  The compiler never allocates AccVGPRs in this kernel (its
// MFMAs accumulate in VGPRs, -amdgpu-mfma-vgpr-form, and it stays under 256 VGPRs -- tests/test_isa_lint.py checks
// that nothing else touches a[...]); the clobber lists put the registers into the kernel's allocation.
MPC_DEV void rg_put(int t, f32x4 v)
{
    switch (t) {
        case 0: asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a0", "a1", "a2", "a3"); break;
        case 1: asm volatile("v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %1\n v_accvgpr_write_b32 a6, %2\n v_accvgpr_write_b32 a7, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a4", "a5", "a6", "a7"); break;
        case 2: asm volatile("v_accvgpr_write_b32 a8, %0\n v_accvgpr_write_b32 a9, %1\n v_accvgpr_write_b32 a10, %2\n v_accvgpr_write_b32 a11, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a8", "a9", "a10", "a11"); break;
        case 3: asm volatile("v_accvgpr_write_b32 a12, %0\n v_accvgpr_write_b32 a13, %1\n v_accvgpr_write_b32 a14, %2\n v_accvgpr_write_b32 a15, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a12", "a13", "a14", "a15"); break;
        case 4: asm volatile("v_accvgpr_write_b32 a16, %0\n v_accvgpr_write_b32 a17, %1\n v_accvgpr_write_b32 a18, %2\n v_accvgpr_write_b32 a19, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a16", "a17", "a18", "a19"); break;
        case 5: asm volatile("v_accvgpr_write_b32 a20, %0\n v_accvgpr_write_b32 a21, %1\n v_accvgpr_write_b32 a22, %2\n v_accvgpr_write_b32 a23, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a20", "a21", "a22", "a23"); break;
        case 6: asm volatile("v_accvgpr_write_b32 a24, %0\n v_accvgpr_write_b32 a25, %1\n v_accvgpr_write_b32 a26, %2\n v_accvgpr_write_b32 a27, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a24", "a25", "a26", "a27"); break;
        case 7: asm volatile("v_accvgpr_write_b32 a28, %0\n v_accvgpr_write_b32 a29, %1\n v_accvgpr_write_b32 a30, %2\n v_accvgpr_write_b32 a31, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a28", "a29", "a30", "a31"); break;
        case 8: asm volatile("v_accvgpr_write_b32 a32, %0\n v_accvgpr_write_b32 a33, %1\n v_accvgpr_write_b32 a34, %2\n v_accvgpr_write_b32 a35, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a32", "a33", "a34", "a35"); break;
        case 9: asm volatile("v_accvgpr_write_b32 a36, %0\n v_accvgpr_write_b32 a37, %1\n v_accvgpr_write_b32 a38, %2\n v_accvgpr_write_b32 a39, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a36", "a37", "a38", "a39"); break;
        case 10: asm volatile("v_accvgpr_write_b32 a40, %0\n v_accvgpr_write_b32 a41, %1\n v_accvgpr_write_b32 a42, %2\n v_accvgpr_write_b32 a43, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a40", "a41", "a42", "a43"); break;
        case 11: asm volatile("v_accvgpr_write_b32 a44, %0\n v_accvgpr_write_b32 a45, %1\n v_accvgpr_write_b32 a46, %2\n v_accvgpr_write_b32 a47, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a44", "a45", "a46", "a47"); break;
        case 12: asm volatile("v_accvgpr_write_b32 a48, %0\n v_accvgpr_write_b32 a49, %1\n v_accvgpr_write_b32 a50, %2\n v_accvgpr_write_b32 a51, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a48", "a49", "a50", "a51"); break;
        case 13: asm volatile("v_accvgpr_write_b32 a52, %0\n v_accvgpr_write_b32 a53, %1\n v_accvgpr_write_b32 a54, %2\n v_accvgpr_write_b32 a55, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a52", "a53", "a54", "a55"); break;
        case 14: asm volatile("v_accvgpr_write_b32 a56, %0\n v_accvgpr_write_b32 a57, %1\n v_accvgpr_write_b32 a58, %2\n v_accvgpr_write_b32 a59, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a56", "a57", "a58", "a59"); break;
        case 15: asm volatile("v_accvgpr_write_b32 a60, %0\n v_accvgpr_write_b32 a61, %1\n v_accvgpr_write_b32 a62, %2\n v_accvgpr_write_b32 a63, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a60", "a61", "a62", "a63"); break;
        case 16: asm volatile("v_accvgpr_write_b32 a64, %0\n v_accvgpr_write_b32 a65, %1\n v_accvgpr_write_b32 a66, %2\n v_accvgpr_write_b32 a67, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a64", "a65", "a66", "a67"); break;
        case 17: asm volatile("v_accvgpr_write_b32 a68, %0\n v_accvgpr_write_b32 a69, %1\n v_accvgpr_write_b32 a70, %2\n v_accvgpr_write_b32 a71, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a68", "a69", "a70", "a71"); break;
        case 18: asm volatile("v_accvgpr_write_b32 a72, %0\n v_accvgpr_write_b32 a73, %1\n v_accvgpr_write_b32 a74, %2\n v_accvgpr_write_b32 a75, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a72", "a73", "a74", "a75"); break;
        case 19: asm volatile("v_accvgpr_write_b32 a76, %0\n v_accvgpr_write_b32 a77, %1\n v_accvgpr_write_b32 a78, %2\n v_accvgpr_write_b32 a79, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a76", "a77", "a78", "a79"); break;
        case 20: asm volatile("v_accvgpr_write_b32 a80, %0\n v_accvgpr_write_b32 a81, %1\n v_accvgpr_write_b32 a82, %2\n v_accvgpr_write_b32 a83, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a80", "a81", "a82", "a83"); break;
        case 21: asm volatile("v_accvgpr_write_b32 a84, %0\n v_accvgpr_write_b32 a85, %1\n v_accvgpr_write_b32 a86, %2\n v_accvgpr_write_b32 a87, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a84", "a85", "a86", "a87"); break;
        case 22: asm volatile("v_accvgpr_write_b32 a88, %0\n v_accvgpr_write_b32 a89, %1\n v_accvgpr_write_b32 a90, %2\n v_accvgpr_write_b32 a91, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a88", "a89", "a90", "a91"); break;
        case 23: asm volatile("v_accvgpr_write_b32 a92, %0\n v_accvgpr_write_b32 a93, %1\n v_accvgpr_write_b32 a94, %2\n v_accvgpr_write_b32 a95, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a92", "a93", "a94", "a95"); break;
        case 24: asm volatile("v_accvgpr_write_b32 a96, %0\n v_accvgpr_write_b32 a97, %1\n v_accvgpr_write_b32 a98, %2\n v_accvgpr_write_b32 a99, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a96", "a97", "a98", "a99"); break;
        case 25: asm volatile("v_accvgpr_write_b32 a100, %0\n v_accvgpr_write_b32 a101, %1\n v_accvgpr_write_b32 a102, %2\n v_accvgpr_write_b32 a103, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a100", "a101", "a102", "a103"); break;
        case 26: asm volatile("v_accvgpr_write_b32 a104, %0\n v_accvgpr_write_b32 a105, %1\n v_accvgpr_write_b32 a106, %2\n v_accvgpr_write_b32 a107, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a104", "a105", "a106", "a107"); break;
        case 27: asm volatile("v_accvgpr_write_b32 a108, %0\n v_accvgpr_write_b32 a109, %1\n v_accvgpr_write_b32 a110, %2\n v_accvgpr_write_b32 a111, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a108", "a109", "a110", "a111"); break;
        case 28: asm volatile("v_accvgpr_write_b32 a112, %0\n v_accvgpr_write_b32 a113, %1\n v_accvgpr_write_b32 a114, %2\n v_accvgpr_write_b32 a115, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a112", "a113", "a114", "a115"); break;
        case 29: asm volatile("v_accvgpr_write_b32 a116, %0\n v_accvgpr_write_b32 a117, %1\n v_accvgpr_write_b32 a118, %2\n v_accvgpr_write_b32 a119, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a116", "a117", "a118", "a119"); break;
        case 30: asm volatile("v_accvgpr_write_b32 a120, %0\n v_accvgpr_write_b32 a121, %1\n v_accvgpr_write_b32 a122, %2\n v_accvgpr_write_b32 a123, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a120", "a121", "a122", "a123"); break;
        case 31: asm volatile("v_accvgpr_write_b32 a124, %0\n v_accvgpr_write_b32 a125, %1\n v_accvgpr_write_b32 a126, %2\n v_accvgpr_write_b32 a127, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a124", "a125", "a126", "a127"); break;
        case 32: asm volatile("v_accvgpr_write_b32 a128, %0\n v_accvgpr_write_b32 a129, %1\n v_accvgpr_write_b32 a130, %2\n v_accvgpr_write_b32 a131, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a128", "a129", "a130", "a131"); break;
        case 33: asm volatile("v_accvgpr_write_b32 a132, %0\n v_accvgpr_write_b32 a133, %1\n v_accvgpr_write_b32 a134, %2\n v_accvgpr_write_b32 a135, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a132", "a133", "a134", "a135"); break;
        case 34: asm volatile("v_accvgpr_write_b32 a136, %0\n v_accvgpr_write_b32 a137, %1\n v_accvgpr_write_b32 a138, %2\n v_accvgpr_write_b32 a139, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a136", "a137", "a138", "a139"); break;
        case 35: asm volatile("v_accvgpr_write_b32 a140, %0\n v_accvgpr_write_b32 a141, %1\n v_accvgpr_write_b32 a142, %2\n v_accvgpr_write_b32 a143, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a140", "a141", "a142", "a143"); break;
        case 36: asm volatile("v_accvgpr_write_b32 a144, %0\n v_accvgpr_write_b32 a145, %1\n v_accvgpr_write_b32 a146, %2\n v_accvgpr_write_b32 a147, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a144", "a145", "a146", "a147"); break;
        case 37: asm volatile("v_accvgpr_write_b32 a148, %0\n v_accvgpr_write_b32 a149, %1\n v_accvgpr_write_b32 a150, %2\n v_accvgpr_write_b32 a151, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a148", "a149", "a150", "a151"); break;
        case 38: asm volatile("v_accvgpr_write_b32 a152, %0\n v_accvgpr_write_b32 a153, %1\n v_accvgpr_write_b32 a154, %2\n v_accvgpr_write_b32 a155, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a152", "a153", "a154", "a155"); break;
        case 39: asm volatile("v_accvgpr_write_b32 a156, %0\n v_accvgpr_write_b32 a157, %1\n v_accvgpr_write_b32 a158, %2\n v_accvgpr_write_b32 a159, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a156", "a157", "a158", "a159"); break;
        case 40: asm volatile("v_accvgpr_write_b32 a160, %0\n v_accvgpr_write_b32 a161, %1\n v_accvgpr_write_b32 a162, %2\n v_accvgpr_write_b32 a163, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a160", "a161", "a162", "a163"); break;
        case 41: asm volatile("v_accvgpr_write_b32 a164, %0\n v_accvgpr_write_b32 a165, %1\n v_accvgpr_write_b32 a166, %2\n v_accvgpr_write_b32 a167, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a164", "a165", "a166", "a167"); break;
        case 42: asm volatile("v_accvgpr_write_b32 a168, %0\n v_accvgpr_write_b32 a169, %1\n v_accvgpr_write_b32 a170, %2\n v_accvgpr_write_b32 a171, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a168", "a169", "a170", "a171"); break;
        case 43: asm volatile("v_accvgpr_write_b32 a172, %0\n v_accvgpr_write_b32 a173, %1\n v_accvgpr_write_b32 a174, %2\n v_accvgpr_write_b32 a175, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a172", "a173", "a174", "a175"); break;
        case 44: asm volatile("v_accvgpr_write_b32 a176, %0\n v_accvgpr_write_b32 a177, %1\n v_accvgpr_write_b32 a178, %2\n v_accvgpr_write_b32 a179, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a176", "a177", "a178", "a179"); break;
        case 45: asm volatile("v_accvgpr_write_b32 a180, %0\n v_accvgpr_write_b32 a181, %1\n v_accvgpr_write_b32 a182, %2\n v_accvgpr_write_b32 a183, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a180", "a181", "a182", "a183"); break;
        case 46: asm volatile("v_accvgpr_write_b32 a184, %0\n v_accvgpr_write_b32 a185, %1\n v_accvgpr_write_b32 a186, %2\n v_accvgpr_write_b32 a187, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a184", "a185", "a186", "a187"); break;
        case 47: asm volatile("v_accvgpr_write_b32 a188, %0\n v_accvgpr_write_b32 a189, %1\n v_accvgpr_write_b32 a190, %2\n v_accvgpr_write_b32 a191, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a188", "a189", "a190", "a191"); break;
        case 48: asm volatile("v_accvgpr_write_b32 a192, %0\n v_accvgpr_write_b32 a193, %1\n v_accvgpr_write_b32 a194, %2\n v_accvgpr_write_b32 a195, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a192", "a193", "a194", "a195"); break;
        case 49: asm volatile("v_accvgpr_write_b32 a196, %0\n v_accvgpr_write_b32 a197, %1\n v_accvgpr_write_b32 a198, %2\n v_accvgpr_write_b32 a199, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a196", "a197", "a198", "a199"); break;
        case 50: asm volatile("v_accvgpr_write_b32 a200, %0\n v_accvgpr_write_b32 a201, %1\n v_accvgpr_write_b32 a202, %2\n v_accvgpr_write_b32 a203, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a200", "a201", "a202", "a203"); break;
        case 51: asm volatile("v_accvgpr_write_b32 a204, %0\n v_accvgpr_write_b32 a205, %1\n v_accvgpr_write_b32 a206, %2\n v_accvgpr_write_b32 a207, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a204", "a205", "a206", "a207"); break;
        case 52: asm volatile("v_accvgpr_write_b32 a208, %0\n v_accvgpr_write_b32 a209, %1\n v_accvgpr_write_b32 a210, %2\n v_accvgpr_write_b32 a211, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a208", "a209", "a210", "a211"); break;
        case 53: asm volatile("v_accvgpr_write_b32 a212, %0\n v_accvgpr_write_b32 a213, %1\n v_accvgpr_write_b32 a214, %2\n v_accvgpr_write_b32 a215, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a212", "a213", "a214", "a215"); break;
        case 54: asm volatile("v_accvgpr_write_b32 a216, %0\n v_accvgpr_write_b32 a217, %1\n v_accvgpr_write_b32 a218, %2\n v_accvgpr_write_b32 a219, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a216", "a217", "a218", "a219"); break;
        case 55: asm volatile("v_accvgpr_write_b32 a220, %0\n v_accvgpr_write_b32 a221, %1\n v_accvgpr_write_b32 a222, %2\n v_accvgpr_write_b32 a223, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a220", "a221", "a222", "a223"); break;
        case 56: asm volatile("v_accvgpr_write_b32 a224, %0\n v_accvgpr_write_b32 a225, %1\n v_accvgpr_write_b32 a226, %2\n v_accvgpr_write_b32 a227, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a224", "a225", "a226", "a227"); break;
        case 57: asm volatile("v_accvgpr_write_b32 a228, %0\n v_accvgpr_write_b32 a229, %1\n v_accvgpr_write_b32 a230, %2\n v_accvgpr_write_b32 a231, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a228", "a229", "a230", "a231"); break;
        case 58: asm volatile("v_accvgpr_write_b32 a232, %0\n v_accvgpr_write_b32 a233, %1\n v_accvgpr_write_b32 a234, %2\n v_accvgpr_write_b32 a235, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a232", "a233", "a234", "a235"); break;
        case 59: asm volatile("v_accvgpr_write_b32 a236, %0\n v_accvgpr_write_b32 a237, %1\n v_accvgpr_write_b32 a238, %2\n v_accvgpr_write_b32 a239, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a236", "a237", "a238", "a239"); break;
        case 60: asm volatile("v_accvgpr_write_b32 a240, %0\n v_accvgpr_write_b32 a241, %1\n v_accvgpr_write_b32 a242, %2\n v_accvgpr_write_b32 a243, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a240", "a241", "a242", "a243"); break;
        case 61: asm volatile("v_accvgpr_write_b32 a244, %0\n v_accvgpr_write_b32 a245, %1\n v_accvgpr_write_b32 a246, %2\n v_accvgpr_write_b32 a247, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a244", "a245", "a246", "a247"); break;
        case 62: asm volatile("v_accvgpr_write_b32 a248, %0\n v_accvgpr_write_b32 a249, %1\n v_accvgpr_write_b32 a250, %2\n v_accvgpr_write_b32 a251, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a248", "a249", "a250", "a251"); break;
        case 63: asm volatile("v_accvgpr_write_b32 a252, %0\n v_accvgpr_write_b32 a253, %1\n v_accvgpr_write_b32 a254, %2\n v_accvgpr_write_b32 a255, %3" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a252", "a253", "a254", "a255"); break;
    }
}
MPC_DEV f32x4 rg_get(int t)
{
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    switch (t) {
        case 0: asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 1: asm volatile("v_accvgpr_read_b32 %0, a4\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_read_b32 %2, a6\n v_accvgpr_read_b32 %3, a7" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 2: asm volatile("v_accvgpr_read_b32 %0, a8\n v_accvgpr_read_b32 %1, a9\n v_accvgpr_read_b32 %2, a10\n v_accvgpr_read_b32 %3, a11" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 3: asm volatile("v_accvgpr_read_b32 %0, a12\n v_accvgpr_read_b32 %1, a13\n v_accvgpr_read_b32 %2, a14\n v_accvgpr_read_b32 %3, a15" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 4: asm volatile("v_accvgpr_read_b32 %0, a16\n v_accvgpr_read_b32 %1, a17\n v_accvgpr_read_b32 %2, a18\n v_accvgpr_read_b32 %3, a19" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 5: asm volatile("v_accvgpr_read_b32 %0, a20\n v_accvgpr_read_b32 %1, a21\n v_accvgpr_read_b32 %2, a22\n v_accvgpr_read_b32 %3, a23" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 6: asm volatile("v_accvgpr_read_b32 %0, a24\n v_accvgpr_read_b32 %1, a25\n v_accvgpr_read_b32 %2, a26\n v_accvgpr_read_b32 %3, a27" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 7: asm volatile("v_accvgpr_read_b32 %0, a28\n v_accvgpr_read_b32 %1, a29\n v_accvgpr_read_b32 %2, a30\n v_accvgpr_read_b32 %3, a31" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 8: asm volatile("v_accvgpr_read_b32 %0, a32\n v_accvgpr_read_b32 %1, a33\n v_accvgpr_read_b32 %2, a34\n v_accvgpr_read_b32 %3, a35" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 9: asm volatile("v_accvgpr_read_b32 %0, a36\n v_accvgpr_read_b32 %1, a37\n v_accvgpr_read_b32 %2, a38\n v_accvgpr_read_b32 %3, a39" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 10: asm volatile("v_accvgpr_read_b32 %0, a40\n v_accvgpr_read_b32 %1, a41\n v_accvgpr_read_b32 %2, a42\n v_accvgpr_read_b32 %3, a43" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 11: asm volatile("v_accvgpr_read_b32 %0, a44\n v_accvgpr_read_b32 %1, a45\n v_accvgpr_read_b32 %2, a46\n v_accvgpr_read_b32 %3, a47" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 12: asm volatile("v_accvgpr_read_b32 %0, a48\n v_accvgpr_read_b32 %1, a49\n v_accvgpr_read_b32 %2, a50\n v_accvgpr_read_b32 %3, a51" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 13: asm volatile("v_accvgpr_read_b32 %0, a52\n v_accvgpr_read_b32 %1, a53\n v_accvgpr_read_b32 %2, a54\n v_accvgpr_read_b32 %3, a55" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 14: asm volatile("v_accvgpr_read_b32 %0, a56\n v_accvgpr_read_b32 %1, a57\n v_accvgpr_read_b32 %2, a58\n v_accvgpr_read_b32 %3, a59" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 15: asm volatile("v_accvgpr_read_b32 %0, a60\n v_accvgpr_read_b32 %1, a61\n v_accvgpr_read_b32 %2, a62\n v_accvgpr_read_b32 %3, a63" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 16: asm volatile("v_accvgpr_read_b32 %0, a64\n v_accvgpr_read_b32 %1, a65\n v_accvgpr_read_b32 %2, a66\n v_accvgpr_read_b32 %3, a67" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 17: asm volatile("v_accvgpr_read_b32 %0, a68\n v_accvgpr_read_b32 %1, a69\n v_accvgpr_read_b32 %2, a70\n v_accvgpr_read_b32 %3, a71" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 18: asm volatile("v_accvgpr_read_b32 %0, a72\n v_accvgpr_read_b32 %1, a73\n v_accvgpr_read_b32 %2, a74\n v_accvgpr_read_b32 %3, a75" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 19: asm volatile("v_accvgpr_read_b32 %0, a76\n v_accvgpr_read_b32 %1, a77\n v_accvgpr_read_b32 %2, a78\n v_accvgpr_read_b32 %3, a79" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 20: asm volatile("v_accvgpr_read_b32 %0, a80\n v_accvgpr_read_b32 %1, a81\n v_accvgpr_read_b32 %2, a82\n v_accvgpr_read_b32 %3, a83" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 21: asm volatile("v_accvgpr_read_b32 %0, a84\n v_accvgpr_read_b32 %1, a85\n v_accvgpr_read_b32 %2, a86\n v_accvgpr_read_b32 %3, a87" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 22: asm volatile("v_accvgpr_read_b32 %0, a88\n v_accvgpr_read_b32 %1, a89\n v_accvgpr_read_b32 %2, a90\n v_accvgpr_read_b32 %3, a91" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 23: asm volatile("v_accvgpr_read_b32 %0, a92\n v_accvgpr_read_b32 %1, a93\n v_accvgpr_read_b32 %2, a94\n v_accvgpr_read_b32 %3, a95" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 24: asm volatile("v_accvgpr_read_b32 %0, a96\n v_accvgpr_read_b32 %1, a97\n v_accvgpr_read_b32 %2, a98\n v_accvgpr_read_b32 %3, a99" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 25: asm volatile("v_accvgpr_read_b32 %0, a100\n v_accvgpr_read_b32 %1, a101\n v_accvgpr_read_b32 %2, a102\n v_accvgpr_read_b32 %3, a103" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 26: asm volatile("v_accvgpr_read_b32 %0, a104\n v_accvgpr_read_b32 %1, a105\n v_accvgpr_read_b32 %2, a106\n v_accvgpr_read_b32 %3, a107" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 27: asm volatile("v_accvgpr_read_b32 %0, a108\n v_accvgpr_read_b32 %1, a109\n v_accvgpr_read_b32 %2, a110\n v_accvgpr_read_b32 %3, a111" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 28: asm volatile("v_accvgpr_read_b32 %0, a112\n v_accvgpr_read_b32 %1, a113\n v_accvgpr_read_b32 %2, a114\n v_accvgpr_read_b32 %3, a115" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 29: asm volatile("v_accvgpr_read_b32 %0, a116\n v_accvgpr_read_b32 %1, a117\n v_accvgpr_read_b32 %2, a118\n v_accvgpr_read_b32 %3, a119" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 30: asm volatile("v_accvgpr_read_b32 %0, a120\n v_accvgpr_read_b32 %1, a121\n v_accvgpr_read_b32 %2, a122\n v_accvgpr_read_b32 %3, a123" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 31: asm volatile("v_accvgpr_read_b32 %0, a124\n v_accvgpr_read_b32 %1, a125\n v_accvgpr_read_b32 %2, a126\n v_accvgpr_read_b32 %3, a127" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 32: asm volatile("v_accvgpr_read_b32 %0, a128\n v_accvgpr_read_b32 %1, a129\n v_accvgpr_read_b32 %2, a130\n v_accvgpr_read_b32 %3, a131" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 33: asm volatile("v_accvgpr_read_b32 %0, a132\n v_accvgpr_read_b32 %1, a133\n v_accvgpr_read_b32 %2, a134\n v_accvgpr_read_b32 %3, a135" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 34: asm volatile("v_accvgpr_read_b32 %0, a136\n v_accvgpr_read_b32 %1, a137\n v_accvgpr_read_b32 %2, a138\n v_accvgpr_read_b32 %3, a139" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 35: asm volatile("v_accvgpr_read_b32 %0, a140\n v_accvgpr_read_b32 %1, a141\n v_accvgpr_read_b32 %2, a142\n v_accvgpr_read_b32 %3, a143" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 36: asm volatile("v_accvgpr_read_b32 %0, a144\n v_accvgpr_read_b32 %1, a145\n v_accvgpr_read_b32 %2, a146\n v_accvgpr_read_b32 %3, a147" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 37: asm volatile("v_accvgpr_read_b32 %0, a148\n v_accvgpr_read_b32 %1, a149\n v_accvgpr_read_b32 %2, a150\n v_accvgpr_read_b32 %3, a151" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 38: asm volatile("v_accvgpr_read_b32 %0, a152\n v_accvgpr_read_b32 %1, a153\n v_accvgpr_read_b32 %2, a154\n v_accvgpr_read_b32 %3, a155" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 39: asm volatile("v_accvgpr_read_b32 %0, a156\n v_accvgpr_read_b32 %1, a157\n v_accvgpr_read_b32 %2, a158\n v_accvgpr_read_b32 %3, a159" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 40: asm volatile("v_accvgpr_read_b32 %0, a160\n v_accvgpr_read_b32 %1, a161\n v_accvgpr_read_b32 %2, a162\n v_accvgpr_read_b32 %3, a163" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 41: asm volatile("v_accvgpr_read_b32 %0, a164\n v_accvgpr_read_b32 %1, a165\n v_accvgpr_read_b32 %2, a166\n v_accvgpr_read_b32 %3, a167" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 42: asm volatile("v_accvgpr_read_b32 %0, a168\n v_accvgpr_read_b32 %1, a169\n v_accvgpr_read_b32 %2, a170\n v_accvgpr_read_b32 %3, a171" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 43: asm volatile("v_accvgpr_read_b32 %0, a172\n v_accvgpr_read_b32 %1, a173\n v_accvgpr_read_b32 %2, a174\n v_accvgpr_read_b32 %3, a175" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 44: asm volatile("v_accvgpr_read_b32 %0, a176\n v_accvgpr_read_b32 %1, a177\n v_accvgpr_read_b32 %2, a178\n v_accvgpr_read_b32 %3, a179" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 45: asm volatile("v_accvgpr_read_b32 %0, a180\n v_accvgpr_read_b32 %1, a181\n v_accvgpr_read_b32 %2, a182\n v_accvgpr_read_b32 %3, a183" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 46: asm volatile("v_accvgpr_read_b32 %0, a184\n v_accvgpr_read_b32 %1, a185\n v_accvgpr_read_b32 %2, a186\n v_accvgpr_read_b32 %3, a187" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 47: asm volatile("v_accvgpr_read_b32 %0, a188\n v_accvgpr_read_b32 %1, a189\n v_accvgpr_read_b32 %2, a190\n v_accvgpr_read_b32 %3, a191" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 48: asm volatile("v_accvgpr_read_b32 %0, a192\n v_accvgpr_read_b32 %1, a193\n v_accvgpr_read_b32 %2, a194\n v_accvgpr_read_b32 %3, a195" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 49: asm volatile("v_accvgpr_read_b32 %0, a196\n v_accvgpr_read_b32 %1, a197\n v_accvgpr_read_b32 %2, a198\n v_accvgpr_read_b32 %3, a199" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 50: asm volatile("v_accvgpr_read_b32 %0, a200\n v_accvgpr_read_b32 %1, a201\n v_accvgpr_read_b32 %2, a202\n v_accvgpr_read_b32 %3, a203" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 51: asm volatile("v_accvgpr_read_b32 %0, a204\n v_accvgpr_read_b32 %1, a205\n v_accvgpr_read_b32 %2, a206\n v_accvgpr_read_b32 %3, a207" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 52: asm volatile("v_accvgpr_read_b32 %0, a208\n v_accvgpr_read_b32 %1, a209\n v_accvgpr_read_b32 %2, a210\n v_accvgpr_read_b32 %3, a211" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 53: asm volatile("v_accvgpr_read_b32 %0, a212\n v_accvgpr_read_b32 %1, a213\n v_accvgpr_read_b32 %2, a214\n v_accvgpr_read_b32 %3, a215" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 54: asm volatile("v_accvgpr_read_b32 %0, a216\n v_accvgpr_read_b32 %1, a217\n v_accvgpr_read_b32 %2, a218\n v_accvgpr_read_b32 %3, a219" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 55: asm volatile("v_accvgpr_read_b32 %0, a220\n v_accvgpr_read_b32 %1, a221\n v_accvgpr_read_b32 %2, a222\n v_accvgpr_read_b32 %3, a223" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 56: asm volatile("v_accvgpr_read_b32 %0, a224\n v_accvgpr_read_b32 %1, a225\n v_accvgpr_read_b32 %2, a226\n v_accvgpr_read_b32 %3, a227" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 57: asm volatile("v_accvgpr_read_b32 %0, a228\n v_accvgpr_read_b32 %1, a229\n v_accvgpr_read_b32 %2, a230\n v_accvgpr_read_b32 %3, a231" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 58: asm volatile("v_accvgpr_read_b32 %0, a232\n v_accvgpr_read_b32 %1, a233\n v_accvgpr_read_b32 %2, a234\n v_accvgpr_read_b32 %3, a235" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 59: asm volatile("v_accvgpr_read_b32 %0, a236\n v_accvgpr_read_b32 %1, a237\n v_accvgpr_read_b32 %2, a238\n v_accvgpr_read_b32 %3, a239" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 60: asm volatile("v_accvgpr_read_b32 %0, a240\n v_accvgpr_read_b32 %1, a241\n v_accvgpr_read_b32 %2, a242\n v_accvgpr_read_b32 %3, a243" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 61: asm volatile("v_accvgpr_read_b32 %0, a244\n v_accvgpr_read_b32 %1, a245\n v_accvgpr_read_b32 %2, a246\n v_accvgpr_read_b32 %3, a247" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 62: asm volatile("v_accvgpr_read_b32 %0, a248\n v_accvgpr_read_b32 %1, a249\n v_accvgpr_read_b32 %2, a250\n v_accvgpr_read_b32 %3, a251" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
        case 63: asm volatile("v_accvgpr_read_b32 %0, a252\n v_accvgpr_read_b32 %1, a253\n v_accvgpr_read_b32 %2, a254\n v_accvgpr_read_b32 %3, a255" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)); break;
    }
    return f32x4{r0, r1, r2, r3};
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
#ifndef MPC_DPP16_NSTAGE
#define MPC_DPP16_NSTAGE 4
#endif
// (MPC_KF_LDS_BYTES: the compilation that holds only the PADDED fused KKT kernel -- Makefile, lqr_dpp16_padkkt.o -- sweeps on two slots
// like every padded sweep but gives its second pass the deep ring's 36 KiB)
#ifdef MPC_KF_LDS_BYTES
#define MPC_DPP16_LDS (MPC_KF_LDS_BYTES)
#else
#define MPC_DPP16_LDS (MPC_DPP16_NSTAGE * 9216)
#endif
// the KKT kernel shares this file's staging array: it lives in the compilation whose array fits its ring
#ifndef MPC_KKT16_NSTAGE
#define MPC_KKT16_NSTAGE 4
#endif
#if !defined(MPC_DPP16_WITH_KKT) && !defined(MPC_DPP16_NO_KKT) && MPC_DPP16_NSTAGE == 4
#define MPC_DPP16_WITH_KKT 1
#endif
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
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Gradients (dC, dF, dc, df, dx, du) are written once and never read back by the kernel: NON-TEMPORAL stores.  Round 4,
// same box, fused backward at the headline shape: 218 -> 185 us (0.46 -> 0.54 of its roofline) -- as cached stores the 380 MB of
// gradients pushed the (V, v, lambda, g) records pass 2 reads back, and the F blocks of both passes, out of the Infinity Cache.
// -DMPC_KF_CACHED_GRADS restores the cached form for the A/B.  (store_f32_out stays cached: the workspace records.)
MPC_DEV void store_f32x2_out(float *g, float a, float b)                                        // 8-byte aligned
{
#ifdef MPC_KF_CACHED_GRADS
    *(f32x2 *)g = f32x2{a, b};
#else
    __builtin_nontemporal_store(f32x2{a, b}, (f32x2 *)g);
#endif
}
MPC_DEV void store_f32_grad(float *g, float v)
{
#ifdef MPC_KF_CACHED_GRADS
    *g = v;
#else
    __builtin_nontemporal_store(v, g);
#endif
}
// 16 bytes of a gradient block at 4-byte alignment (the padded fused backward's runs: global memory takes unaligned vector accesses)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
MPC_DEV void store_f32x4_grad(float *g, float __attribute__((ext_vector_type(4))) v)
{
#ifdef MPC_KF_CACHED_GRADS
    *(f32x4_a4 *)g = v;
#else
    __builtin_nontemporal_store(v, (f32x4_a4 *)g);
#endif
}
// the value of the neighbouring lane j ^ 1 (quad_perm [1,0,3,2])
MPC_DEV float swap1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true)); }
MPC_DEV void dma16_if(bool active, const void *g, unsigned off)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage16 + off), 16, 0, 0);
}
// ---- the padded instantiation's staging (lqr_dpp16_body.h, PADK; the same two primitives as lqr_mfma40.hip) ----------------
// 4 bytes per lane from `base + voff` (base wave-uniform: a raw buffer of `nbytes`, voff per lane) to LDS offset `off` + 4 * lane.
// A lane whose voff lies beyond the buffer writes ZERO (the hardware's range check; measured, tools/ubench/buffer_lds_probe.hip).
template <int G> MPC_DEV void dma_buf(bool active, const void *base, unsigned nbytes, unsigned voff, unsigned off)
{
    static_assert(G == 4, "dword gathers");
    if (active) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)nbytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(g_stage16 + off), 4, (int)voff, 0, 0, 0);
    }
}
MPC_DEV void dma4_if(bool active, const void *g, unsigned off)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage16 + off), 4, 0, 0);
}
// ... with the instruction's immediate offset IMM: it moves the LDS destination and the buffer offset together (like dma16_at: one M0 per
// group of gathers instead of one per gather -- three instructions a gather became one).  `voff` = source offset - IMM + BIAS, `base_biased` =
// the block's base - BIAS, `nbytes_biased` = its size + BIAS: the bias keeps voff non-negative, the hardware's range check of voff + IMM
// against the record count still sends every offset beyond the block to zero.  `anchor`: the LDS address IMM counts from.
template <int IMM, int BIAS> MPC_DEV void dma_buf_at(const void *base_biased, unsigned nbytes_biased, unsigned voff, unsigned anchor)
{
    static_assert(IMM >= 0 && IMM < 4096, "12-bit unsigned immediate");
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base_biased, (short)0, (int)nbytes_biased, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(g_stage16 + anchor), 4, (int)voff, 0, IMM, 0);
}
template <int IMM> MPC_DEV void dma4_at_if(bool active, const void *g, unsigned anchor)
{
    if (active) __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(g_stage16 + anchor), 4, IMM, 0);
}
MPC_DEV void lds_store_f32x4(unsigned off, f32x4 v) { *(f32x4 *)(g_stage16 + off) = v; }
MPC_DEV void lds_store_f32(unsigned off, float v) { *(float *)(g_stage16 + off) = v; }
// DS instructions of one wave execute in program order: a compiler barrier is all there is to ask for
MPC_DEV void lds_sync() { asm volatile("" ::: "memory"); }
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
// a word of flag bits travelling in a float slot of a record (moves and selects keep the bits)
MPC_DEV float bits_f32(unsigned u) { return __uint_as_float(u); }
MPC_DEV unsigned f32_bits(float f) { return __float_as_uint(f); }
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

// MODE: 0 unconstrained with the gains in registers (T <= 64), 1 unconstrained + u_zero_I, 2 box-constrained (pnqp in the
// sweep), 3 unconstrained with the gains through memory (any T)
template <int MODE>
__global__ void __launch_bounds__(64, 1) lqr_step_dpp16_kernel(StepParams<float> p)
{
    dpp16::step_wave<MODE>(p);
}

#if MPC_DPP16_NSTAGE == 4 || defined(MPC_DPP16_PAD_KKT)
// the whole of LQRStepFn.backward in one launch (lqr_dpp16_body.h: kkt_fused_wave); MASKED = controls on a bound are pinned
static_assert((int)dpp16::KfP2<false>::SLOTS * (int)dpp16::KfP2<false>::STAGE + (int)dpp16::KfP2<false>::GST <= MPC_DPP16_LDS &&
              (int)dpp16::KfP2<false>::SLOTS >= 5, "the fused KKT kernel's second ring does not fit");
static_assert((int)dpp16::KfP2<true>::SLOTS * (int)dpp16::KfP2<true>::STAGE + (int)dpp16::KfP2<true>::GST <= MPC_DPP16_LDS &&
              (int)dpp16::KfP2<true>::SLOTS >= (dpp16::PADK ? 4 : 5), "the long-horizon fused KKT kernel's second ring does not fit");
static_assert((int)dpp16::LDS_TOTAL <= MPC_DPP16_LDS, "the fused KKT kernel's sweep ring does not fit");
template <bool MASKED>
__global__ void __launch_bounds__(64, 1) lqr_kkt_fused_dpp16_kernel(StepParams<float> p, dpp16::KktFusedArgs k)
{
    dpp16::kkt_fused_wave<MASKED>(p, k);
}
// T > 64 (round 4): the gains through the workspace instead of the accumulation registers (lqr_dpp16_body.h, KfP2<true>)
template <bool MASKED>
__global__ void __launch_bounds__(64, 1) lqr_kkt_fused_long_dpp16_kernel(StepParams<float> p, dpp16::KktFusedArgs k)
{
    dpp16::kkt_fused_wave<MASKED, true>(p, k);
}

#endif

#ifdef MPC_DPP16_WITH_KKT
static_assert(MPC_KKT16_NSTAGE * 8192 <= MPC_DPP16_LDS, "the KKT kernel's ring does not fit this compilation's staging array");
__global__ void __launch_bounds__(64, 1) lqr_kkt_dpp16_kernel(StepParams<float> p, dpp16::KktArgs k)
{
    dpp16::kkt_wave(p, k);
}

#endif
}  // namespace

#ifdef MPC_DPP16_WITH_KKT
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

#endif

#ifdef MPC_DPP16_PAD_KKT
// the PADDED instantiation of the fused KKT backward (round 6): any n_state <= 12, n_ctrl <= 4, no alignment asked of the caller's blocks
// (dword gathers and dword stores); the workspace -- the library's own layout, padded to 12/4 -- on 16 bytes
bool kkt_fused_dpp16_pad_supported(const StepParams<float> &p, const float *ws)
{
    return p.ns >= 1 && p.ns <= 12 && p.nc >= 1 && p.nc <= 4 && p.T >= 1 && (uintptr_t)ws % 16 == 0;
}
#define MPC_KF_LAUNCH launch_kkt_fused_dpp16_pad
#elif MPC_DPP16_NSTAGE == 4
#define MPC_KF_LAUNCH launch_kkt_fused_dpp16
// (built in the compilation with the deep staging array: four sweep stages, six rollout stages, see kkt_fused_wave)
bool kkt_fused_dpp16_supported(const StepParams<float> &p, const float *dl_dx, const float *dl_du, const float *dC,
                               const float *dF, const float *ws)
{
    auto al = [](const void *q, long st, long sb) { return ((uintptr_t)q % 16 == 0) && (st % 4 == 0) && (sb % 4 == 0); };
    if (!(p.ns == 12 && p.nc == 4 && p.T >= 1)) return false;          // (any horizon since round 4: T > 64 on the LONG instantiation)
    if (!al(p.C, p.C_st, p.C_sb) || !al(p.c, p.c_st, p.c_sb)) return false;
    if (p.T > 1 && !al(p.F, p.F_st, p.F_sb)) return false;
    if (p.bound_mode == MPC_BOUND_TENSOR && (!al(p.lo, 0, 0) || !al(p.hi, 0, 0))) return false;
    // (p.zero_mask / p.has_delta: the forward's u_zero_I and delta_u, which the backward does not use -- mpc/lqr_step.py:322-340;
    // launch_kkt_fused_dpp16 clears them)
    return al(p.cur_x, 0, 0) && al(p.cur_u, 0, 0) && al(dl_dx, 0, 0) && al(dl_du, 0, 0) && al(dC, 0, 0) && al(ws, 0, 0) &&
           (p.T == 1 || al(dF, 0, 0));
}

// floats per problem-step: (V | v) 96, (lambda | g) 24, and beyond 64 timesteps the gain record 64
int64_t kkt_fused_dpp16_workspace_bytes(int T, int B)
{
    return (int64_t)T * B * (dpp16::KF_VBLK + 24 + (T > dpp16::RG_STEPS ? 64 : 0)) * 4 + 64;
}
#endif

#if MPC_DPP16_NSTAGE == 4 || defined(MPC_DPP16_PAD_KKT)
int MPC_KF_LAUNCH(const StepParams<float> &p_in, const float *dl_dx, const float *dl_du, float *dC, float *dc, float *dF,
                  float *df, float *dx_init, float *dx_out, float *du_out, float *ws, float decay, int max_ls,
                  hipStream_t st)
{
    StepParams<float> p = p_in;
    p.zero_mask = nullptr;          // the forward's u_zero_I / delta_u are no inputs of the backward (mpc/lqr_step.py:322-340)
    p.has_delta = 0;
    if (dpp16::PADK) p.c_symmetric = 1;          // (the fused backward is only taken under that promise: the plain row layout of C)
    dpp16::KktFusedArgs k;
    k.dl_dx = dl_dx; k.dl_du = dl_du; k.dC = dC; k.dc = dc; k.dF = dF; k.df = df; k.dx_init = dx_init;
    k.dx_out = dx_out; k.du_out = du_out; k.vws = ws; k.decay = decay; k.max_ls = max_ls;
    const dim3 grid((p.B + 3) / 4), block(64);
    if (p.T > dpp16::RG_STEPS) {
        if (p.bound_mode != MPC_BOUND_NONE) hipLaunchKernelGGL((lqr_kkt_fused_long_dpp16_kernel<true>), grid, block, 0, st, p, k);
        else hipLaunchKernelGGL((lqr_kkt_fused_long_dpp16_kernel<false>), grid, block, 0, st, p, k);
    } else if (p.bound_mode != MPC_BOUND_NONE) hipLaunchKernelGGL((lqr_kkt_fused_dpp16_kernel<true>), grid, block, 0, st, p, k);
    else hipLaunchKernelGGL((lqr_kkt_fused_dpp16_kernel<false>), grid, block, 0, st, p, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_kkt_fused_dpp16_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

#endif

#if MPC_DPP16_NSTAGE == 4
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

#endif

// This file is compiled twice (Makefile): with the 4-slot sweep ring (36 KiB of LDS per wave, four waves per CU) as
// launch_step_dpp16, and with -DMPC_DPP16_NSTAGE=2 (18 KiB, eight waves per CU = two per SIMD) as launch_step_dpp16_ring2.
// capi.hip picks: the memory-bound unconstrained step is faster on the short ring at every batch (fewer bytes in
// flight per CU: 83.3 against 86.0 us at B = 4096, 65 against 74 at 1024); the instruction-bound constrained step wants the
// deep ring while there is one wave per SIMD (B <= 4096: 184.8 against 191.0 us) and a second wave per SIMD beyond it
// (B = 6144: 248 against 347 us; 8192: 304 against 371) -- profiles/r02_experiments.md 15.
// ... and a third time with -DMPC_DPP16_PAD (on the 2-slot ring) as launch_step_dpp16_pad: the PADDED instantiation for any n_state <=
// 12, n_ctrl <= 4 (lqr_dpp16_body.h, PADK): dword gathers with the zero padding done by the DMA, no alignment asked of anybody.
#ifndef MPC_DPP16_PAD_KKT
#ifdef MPC_DPP16_PAD
#define MPC_DPP16_LAUNCH launch_step_dpp16_pad
bool dpp16_pad_supported(const StepParams<float> &p)
{
    // (dword gathers: float arrays are 4-byte aligned by construction; u_zero_I is read byte-wise.  12/4 itself is welcome too: blocks or
    // strides that are not 16-byte aligned, which the exact kernel refuses)
    return p.ns >= 1 && p.ns <= 12 && p.nc >= 1 && p.nc <= 4 && p.T >= 1 && p.max_ls >= 1 && p.max_ls <= 16 && !p.env.kind;
}
#define MPC_DPP16_TAKES(p) dpp16_pad_supported(p)
#elif MPC_DPP16_NSTAGE == 4
#define MPC_DPP16_LAUNCH launch_step_dpp16
#else
#define MPC_DPP16_LAUNCH launch_step_dpp16_ring2
bool dpp16_supported(const StepParams<float> &p);
#endif
int MPC_DPP16_LAUNCH(const StepParams<float> &p, hipStream_t st)
{
#ifndef MPC_DPP16_TAKES
#define MPC_DPP16_TAKES(p) dpp16_supported(p)
#endif
    if (!MPC_DPP16_TAKES(p)) { set_last_error(dpp16::PADK ? "dpp16 (padded): needs fp32, n_state <= 12, n_ctrl <= 4, max_linesearch_iter <= 16" : "dpp16: needs n_state = 12, n_ctrl = 4, fp32, 16-byte aligned blocks"); return MPC_E_DIMS; }
    if (!p.Kk || (uintptr_t)p.Kk % 16 != 0) { set_last_error("dpp16: gain workspace missing or misaligned"); return MPC_E_NULL; }
    if (!p.sweep_only && (!p.new_x || !p.new_u)) { set_last_error("dpp16: new_x / new_u is NULL"); return MPC_E_NULL; }
    static_assert(MPC_DPP16_LDS == dpp16::LDS_TOTAL, "LDS layout out of sync");
    const dim3 grid((p.B + 3) / 4), block(64);
    if (p.bound_mode != MPC_BOUND_NONE) hipLaunchKernelGGL((lqr_step_dpp16_kernel<2>), grid, block, 0, st, p);
    else if (p.zero_mask) hipLaunchKernelGGL((lqr_step_dpp16_kernel<1>), grid, block, 0, st, p);
    else if (p.T <= dpp16::RG_STEPS) hipLaunchKernelGGL((lqr_step_dpp16_kernel<0>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((lqr_step_dpp16_kernel<3>), grid, block, 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("lqr_step_dpp16_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}
#endif

}  // namespace mpclqr
