// lqr_mfma40_body.h -- the Riccati sweep for n_state = 32, n_ctrl = 8 (BASELINE config 5), fp32,
// (unconstrained, u_zero_I-masked or box-constrained): ONE wavefront per problem, every matrix product on v_mfma_f32_16x16x4_f32 with all
// operands in registers -- Y = V F, Q = C + F'Y and V = Qxx + Qxu K chain through the accumulators
// without a single cross-lane move of matrix data.  (The generic kernel runs this shape at 4 % of the
// HBM roofline: its time goes into serialised LDS phases, not arithmetic.)
//
// Written against the wave interface `wv::` (mfma / readlane / shfl_xor / LDS / LDS-DMA): lqr_mfma40.hip
// binds it to gfx950, tests/emu/ to the host-side lockstep emulator.
//
// Reference: mpc/lqr_step.py:284-296 (delta-space linear term) + :52-160 (lqr_backward), unconstrained
// branch :84-94.  K, k are written in the reference's layout; the rollout is a separate kernel.
//
// Layout.  tau = [x(32); u(8)] padded to 48 = 3 tiles of 16.  MFMA 16x16x4: lane l = 16 q + r holds
// A[i=r][k=q], B[k=q][j=r] and D[4q+v][r] in accumulator register v.  A 48x48 matrix in "D layout" is
// tiles T[I][J], register v of lane (q,r) = M[16I + 4q + v][16J + r].
//   * contraction order is ours to choose: block kb = (I',v) stands for rows m = 16I' + 4q + v.  With it
//     a D-layout tile IS a B operand (register v of tile (I',J)), a SYMMETRIC D-layout matrix is its own A
//     operand (V[16Im+r][m] = V[m][16Im+r] = register v of tile (I',Im)), and F, loaded once per step as
//     FB[(I',v)][J] = F[16I'+4q+v][16J+r], is the B operand of Y = V F and the A operand (F') of Q += F'Y.
//   * the u-rows of Q (tile row 2, lanes q < 2) are at once the right-hand sides of K = -Quu^-1 Qux and the
//     A operand (Qxu = Qux') of V = Qxx + Qxu K; K comes out of its solve in the B layout that product wants.
//   * vectors live in "row layout" (lane r holds entry 16J + r, the same in all four lane groups) or in
//     "column layout" (lane group q holds entries 16I + 4q + v); matrix-vector products are per-lane partial
//     sums over the tile registers plus a 2-step (across q) or 4-step (across r) butterfly.
//   * the 8x8 Quu is factorised (LDL') spread over lanes (Ldl8V) in every mode; masked / clamped controls are identity
//     rows and columns of the factorised matrix, the box QP (pnqp8v) works on the same lane-spread data.
// C is read as the symmetric matrix the reference documents it to be (mpc/mpc.py:61-68).
#pragma once
#include <math.h>
#include "lqr_params.h"

#ifndef MPC_DEVM
#define MPC_DEVM MPC_DEV
#endif

// 1: start the box QP of timestep t from the solution of timestep t+1 like the reference (the product).
// 0: pnqp's cold start at every timestep whose Quu is positive definite, as the 12/4 kernel does since round 6 -- measured level
// here: 3.27 -> 2.41 trips per QP, but the 8 x 8 LDL' and its solve cost what the saved trip costs (box-constrained step at
// B = 1024: 396 us against 394, gpurun_out r06b) -- so this kernel keeps the reference's start and the reference's path.
#ifndef MPC_MFMA40_QP_WARM
#define MPC_MFMA40_QP_WARM 1
#endif

namespace mpclqr {
namespace mfma40 {

// c ? a : b on two values that both sit in registers.  Written through wv::pin because hipcc otherwise folds a
// select of two elements of a small local array into ONE dynamically indexed load, which puts the array in scratch
// memory: a store, a load and an `s_waitcnt vmcnt(0)` that also drains the staging DMAs, twice per timestep.
MPC_DEV float pick(bool c, float a, float b)
{
    wv::pin(a);
    wv::pin(b);
    return c ? a : b;
}

// One problem per wavefront: its mask bytes and bound rows sit at wave-uniform addresses and come through the
// scalar path.  A vector load would share the vmcnt queue with the staging DMAs, which the compiler cannot count
// across the loop: it drains the queue in front of every use.
MPC_DEV float uniform_f32(const float *g)
{
    const unsigned u = wv::load_uniform_u32((const unsigned *)g);
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}


using wv::f32x4;
constexpr int NS = 32, NC = 8, N = 40;
constexpr int NSTAGE = 2;      // slots of the pricing rollout's LDS-DMA ring: one step in flight (27 KiB per wave -> 6 waves per CU)
// Slots of the SWEEP's ring.  Two (the DMA one timestep = 3.6 us ahead) is enough while the address translations of the
// blocks are cached; behind a kernel that has walked other memory (its own outer-product kernel in the fused backward: 810 MB
// of gradients) every stage starts with TLB misses and a sweep that is one step ahead waits them out, +2.3 us per timestep
// (tools/k40_tlb_probe.py: a 16 us kernel touching one byte per page of 800 MB slows the next launch by 145 us: the
// step at B = 1024 465 us against 342 on three slots, box-constrained 1010 against 658; with warm translations the two rings
// are level).  Three slots (two timesteps ahead, 36 KiB per wave) is therefore the default; the two-slot compilation
// (26 KiB: six wavefronts per CU instead of four) serves the box-constrained step of batches beyond one wavefront per SIMD.
#ifndef MPC_MFMA40_SWEEP_NSTAGE
#define MPC_MFMA40_SWEEP_NSTAGE 3
#endif
constexpr int SNSTAGE = MPC_MFMA40_SWEEP_NSTAGE;
constexpr unsigned OFF_C = 0, OFF_F = 6400, OFF_R = 11520, STAGE_BYTES = 12032;   // (the record: 320 B, 448 B in the fused backward)
// ---- The PADDED instantiation (round 4; -DMPC_MFMA40_PAD=4|16, lqr_mfma40.hip): any n_state <= 32, n_ctrl <= 8 -----------------
// The reference's sweep is shape-agnostic (mpc/lqr_step.py:61-158); rounds 1-3 had fast kernels for exactly 12/4 and 32/8 and a
// generic kernel at 4 % of the roofline for everything between.  Here the 32/8 kernel runs ANY smaller shape: tau is padded to
// [x(32); u(8)] -- state i at slot i, control a at slot 32 + a, zeros elsewhere, an identity on the padded diagonal of Quu --
// and the padding is done BY THE STAGING DMA: every stage is gathered with `buffer_load ... lds`, whose per-lane source offset
// is free and whose out-of-range lanes write ZERO into LDS (measured, tools/ubench/buffer_lds_probe.hip).  LDS therefore
// holds the dense 40 x 40 / 32 x 40 blocks the kernel was written for and nothing downstream of the staging changes; only
// what touches the caller's arrays directly (x_init, bounds, masks, the trajectory and gain outputs) indexes by the true shape.
// PADG = bytes a lane moves per gather instruction: 16 when n_state and n_ctrl are multiples of 4 (rows and the x | u boundary
// are then 16-byte aligned: the same instruction count as the exact kernel), 4 for every other shape.
#ifdef MPC_MFMA40_PAD
constexpr bool PADK = true;
constexpr int PADG = MPC_MFMA40_PAD;
#else
constexpr bool PADK = false;
constexpr int PADG = 16;
#endif
static_assert(PADG == 4 || PADG == 16, "gather granule");
constexpr int CH_C = PADK ? (N * N * 4 + 64 * PADG - 1) / (64 * PADG) : 7;      // gather instructions of a C block (25 | 7)
constexpr int CH_F = PADK ? (NS * N * 4 + 64 * PADG - 1) / (64 * PADG) : 5;     // ... of an F block (20 | 5)
constexpr int CH_R = PADK ? 2 : 1;                                               // ... of a record (dwords: 2 x 64 words)
constexpr int DMA_PER_STAGE = CH_C + CH_F + CH_R;           // 13 = 7 (C) + 5 (F) + 1 (c | x | u) in the exact kernel
// rollout stage: C | F | K_t (1 KiB) | record (c, x_{t+1}, u_t, f_t, k_t)
constexpr unsigned ROFF_K = 11520, ROFF_R = 12544, RSTAGE_BYTES = 13056;
constexpr int RDMA_PER_STAGE = CH_C + CH_F + 1 + CH_R;      // 14 = 7 (C) + 5 (F) + 1 (K) + 1 (record)
static_assert((SNSTAGE - 1) * DMA_PER_STAGE < 64 && RDMA_PER_STAGE < 64, "vmcnt is 6 bits: the dword gather needs the two-slot sweep ring");
constexpr unsigned OFF_SCR = SNSTAGE * STAGE_BYTES > NSTAGE * RSTAGE_BYTES ? SNSTAGE * STAGE_BYTES : NSTAGE * RSTAGE_BYTES;   // 512 B: row -> column layout turns
constexpr unsigned LDS_TOTAL = OFF_SCR + 512;
typedef StepParams<float> P;
// the flags of controls 4w .. 4w+3 of u_zero_I [T,B,8] at (t, b) = tb
MPC_DEV unsigned zero_mask_word(const P &p, long tb, int w)
{
#ifdef MPC_MFMA40_PAD
    // u_zero_I [T,B,nc] bytes at any nc: the byte of control a out of the aligned dword that holds it (scalar loads want
    // 4-byte alignment); padded controls are free (their row of Quu is the identity, they stay at zero)
    unsigned z = 0u;
    const int nc = p.nc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int a = 4 * w + v;
        const unsigned long A = (unsigned long)(p.zero_mask + tb * nc + (a < nc ? a : 0));
        const unsigned word = wv::load_uniform_u32((const unsigned *)(A & ~3ul));
        const unsigned byte = (word >> (8u * (unsigned)(A & 3ul))) & 0xffu;
        z |= (a < nc && byte != 0u) ? (1u << (8 * v)) : 0u;
    }
    return z;
#else
    return wv::load_uniform_u32((const unsigned *)(p.zero_mask + tb * 8) + w);
#endif
}

struct Lane {
    int lane, r, q, b;
};

// ---- LQRStepFn.backward (mpc/lqr_step.py:312-407) fused into the step of its nested problem (kkt_fused_wave below) ----
// The sweep and the lean rollout compiled with KKT = true carry the two costate recursions along; what they need beyond
// the step's own parameters:
struct KktArgs40 {
    const float *dl_dx, *dl_du;     // [T,B,32], [T,B,8]: the nested problem's linear term is c = -(dl_dx | dl_du)  (:315-320, :338)
    float *dF, *df, *dx_init;       // dF [T-1,B,32,40]: lambda_{t+1}, dlambda_{t+1} parked in the first 64 words of block t
                                    // (kkt_outer_kernel's convention, kkt_wave.hip); df [T-1,B,32] or NULL; dx_init [B,32]
    float *Vws;                     // workspace [T,B,1024]: V_t as the sweep holds it (four D-layout tiles, 16 B per lane)
    float *vgws;                    // workspace [T,B,64]: v_t | g_t
};
// pass 2's stage: F | K_t | record (v_{t+1}, g_{t+1}, k_t) | V_{t+1}; three slots, the DMA two timesteps ahead
constexpr unsigned KOFF_F = 0, KOFF_K = 5120, KOFF_R = 6144, KOFF_V = 6656, KSTAGE_BYTES = 10752;
#ifdef MPC_KF40_VFULL
constexpr int KSLOTS = 3, KDMA_PER_STAGE = 11;                 // 5 (F) + 1 (K) + 1 (record) + 4 (V)
#else
constexpr int KSLOTS = 3, KDMA_PER_STAGE_EXACT = 10;           // 5 (F) + 1 (K) + 1 (record) + 3 (V: tiles (0,0), (1,0), (1,1))
#endif
#ifndef MPC_KF40_VFULL
// pass 2 of the fused backward: F (gathered in the padded instantiation: CH_F instructions) + K + record + three V tiles
// (+ 3 in the padded instantiation: u*_t and its tensor bounds, words 0..23 of the record, for the pinned set -- kkt_pinned_lds)
constexpr int KDMA_PER_STAGE = PADK ? CH_F + 5 + 3 : KDMA_PER_STAGE_EXACT;
static_assert((KSLOTS - 2) * KDMA_PER_STAGE < 64, "vmcnt is 6 bits");
#endif
// the constrained modes' record for the rollout that prices without C (rollout_priced): floats per problem-step
constexpr int PREC = 328;                                      // M [8][32] | Quu [8][8] | m [8]
constexpr int PSCR = 40;                                       // behind the records [T,B,PREC]: the second line-search trial's x' | u' [T,B,40]
constexpr unsigned KLDS_TOTAL = KSLOTS * KSTAGE_BYTES;         // 31.5 KiB per wave: four waves per CU

// ---- staging of the padded instantiation ---------------------------------------------------------------------------------
constexpr unsigned OOB_OFF = 0x7fffff00u;          // a source offset beyond every block: the gather writes zero there
// slot of the padded tau = [x(32); u(8)] -> index in the caller's tau = [x(ns); u(nc)], or -1 (padding)
MPC_DEV int pad_tau(int pi, int ns, int nc) { return pi < NS ? (pi < ns ? pi : -1) : (pi - NS < nc ? ns + (pi - NS) : -1); }
struct Gather {
    unsigned coff[CH_C], foff[CH_F];        // source byte offset of this lane's granule of chunk k inside C_t / F_t (OOB_OFF: padding)
    unsigned cbytes, fbytes;                // bytes of the caller's C_t [n,n] and F_t [ns,n] blocks
};
MPC_DEV void gather_init(Gather &g, const P &p, int lane)
{
    const int ns = p.ns, nc = p.nc, n = ns + nc;
    g.cbytes = (unsigned)(n * n * 4);
    g.fbytes = (unsigned)(ns * n * 4);
    constexpr int WPL = PADG / 4;           // words per lane and instruction
#pragma unroll
    for (int k = 0; k < CH_C; ++k) {
        const int w = (64 * k + lane) * WPL, pi = w / N, pj = w - pi * N;
        const int ai = pi < N ? pad_tau(pi, ns, nc) : -1, aj = pad_tau(pj, ns, nc);
        // (16-byte granules: ns, nc multiples of 4 -- a granule is four valid consecutive columns or four padded ones)
        g.coff[k] = (ai >= 0 && aj >= 0) ? (unsigned)(4 * (ai * n + aj)) : OOB_OFF;
    }
#pragma unroll
    for (int k = 0; k < CH_F; ++k) {
        const int w = (64 * k + lane) * WPL, pi = w / N, pj = w - pi * N;
        const int aj = pad_tau(pj, ns, nc);
        g.foff[k] = (pi < ns && aj >= 0) ? (unsigned)(4 * (pi * n + aj)) : OOB_OFF;
    }
}
// the C_t / F_t block at `blk` (wave-uniform) into the dense 40 x 40 / 32 x 40 layout at LDS offset `off`
MPC_DEV void gather_C(const Gather &g, const float *blk_, unsigned off, int lane)
{
    const float *blk = (const float *)wv::uniform_ptr(blk_);
    const unsigned cbytes = wv::uniform_u32(g.cbytes);
#pragma unroll
    for (int k = 0; k < CH_C; ++k) {
        // (a chunk that lies inside the block whole takes every lane: with the lane test in front of it -- always true, but not to the
        // compiler -- hipcc kept the dword build's descriptor in vector registers and wrapped each of the 25 gathers in a readfirstlane
        // loop: 300 instructions a timestep, round 6)
        const bool whole = 64 * (k + 1) * (PADG / 4) <= N * N;
        wv::dma_buf<PADG>(whole || (64 * k + lane) * (PADG / 4) < N * N, blk, cbytes, g.coff[k], off + (unsigned)(64 * PADG * k));
    }
}
MPC_DEV void gather_F(const Gather &g, const float *blk_, unsigned nbytes_, unsigned off)
{
    const float *blk = (const float *)wv::uniform_ptr(blk_);
    const unsigned nbytes = wv::uniform_u32(nbytes_);
#pragma unroll
    for (int k = 0; k < CH_F; ++k) wv::dma_buf<PADG>(true, blk, nbytes, g.foff[k], off + (unsigned)(64 * PADG * k));
}
// The record (the small vectors of a timestep: c | x | u | f | k at words 0, 40, 72, 80, 112 of its 128) gathered dword by
// dword: lane l fetches words l and 64 + l, each from its own array (or sits the instruction out: padding stays the zero
// pad_clear left there).  kind: which timestep index the word follows (0: t, 1: min(t, T-2) -- f has T-1 entries, 2: t + 1).
struct RecMap {
    const char *ptr[2];
    long step[2];
    int kind[2];
    bool act[2];
};
// x_next: the x segment holds x_{t+1} (rollouts) instead of x_t (sweep); with_c / with_f / kin: which segments this pass reads
MPC_DEV void rec_init(RecMap &m, const P &p, int lane, long b, bool with_c, bool x_next, bool with_f, const float *kin)
{
    const int ns = p.ns, nc = p.nc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int w = 64 * j + lane;
        m.ptr[j] = (const char *)p.cur_x;
        m.step[j] = 0;
        m.kind[j] = 0;
        m.act[j] = false;
        if (w < 40) {
            const int a = pad_tau(w, ns, nc);
            if (with_c && a >= 0) { m.act[j] = true; m.ptr[j] = (const char *)(p.c + b * p.c_sb + a); m.step[j] = p.c_st * 4; }
        } else if (w < 72) {
            const int i = w - 40;
            if (i < ns) { m.act[j] = true; m.ptr[j] = (const char *)(p.cur_x + b * ns + i); m.step[j] = (long)p.B * ns * 4; m.kind[j] = x_next ? 2 : 0; }
        } else if (w < 80) {
            const int a = w - 72;
            if (a < nc) { m.act[j] = true; m.ptr[j] = (const char *)(p.cur_u + b * nc + a); m.step[j] = (long)p.B * nc * 4; }
        } else if (w < 112) {
            const int i = w - 80;
            if (with_f && p.f != nullptr && p.T > 1 && i < ns) {
                m.act[j] = true; m.ptr[j] = (const char *)(p.f + b * p.f_sb + i); m.step[j] = p.f_st * 4; m.kind[j] = 1;
            } else if (!with_f && !x_next && p.bound_mode == MPC_BOUND_TENSOR && i < 16 && (i & 7) < nc) {
                // the box-constrained SWEEP (it never looks at f): words 80..87 u_lower_t, 88..95 u_upper_t ride in the record -- as sixteen
                // scalar loads a timestep they sat in scalar registers across the QP's loop (round 6)
                m.act[j] = true; m.ptr[j] = (const char *)((i < 8 ? p.lo : p.hi) + b * nc + (i & 7)); m.step[j] = (long)p.B * nc * 4;
            }
        } else if (w < 120) {
            if (kin) { m.act[j] = true; m.ptr[j] = (const char *)(kin + b * NC + (w - 112)); m.step[j] = (long)p.B * NC * 4; }
            else if (MPC_QP_START && !x_next && p.qp_start && p.bound_mode != MPC_BOUND_NONE && w - 112 < nc) {
                // mpc_lqr_options.qp_start in the sweep's record, where the exact kernel carries it (strides may be 0)
                m.act[j] = true; m.ptr[j] = (const char *)(p.qp_start + b * p.qp_start_sb + (w - 112)); m.step[j] = p.qp_start_st * 4;
            }
        }
    }
}
// ... of the fused backward's sweep (stream_init with kk): words 0..39 r = (dl_dx | dl_du) in the padded tau's order, 40..79 tau*, 80..111
// c_x of the original problem (lambda's constant term)
MPC_DEV void rec_init_kkt(RecMap &m, const P &p, int lane, long b, const float *dl_dx, const float *dl_du)
{
    const int ns = p.ns, nc = p.nc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int w = 64 * j + lane;
        m.ptr[j] = (const char *)p.cur_x;
        m.step[j] = 0;
        m.kind[j] = 0;
        m.act[j] = false;
        if (w < 32) {
            if (w < ns) { m.act[j] = true; m.ptr[j] = (const char *)(dl_dx + b * ns + w); m.step[j] = (long)p.B * ns * 4; }
        } else if (w < 40) {
            if (w - 32 < nc) { m.act[j] = true; m.ptr[j] = (const char *)(dl_du + b * nc + (w - 32)); m.step[j] = (long)p.B * nc * 4; }
        } else if (w < 72) {
            if (w - 40 < ns) { m.act[j] = true; m.ptr[j] = (const char *)(p.cur_x + b * ns + (w - 40)); m.step[j] = (long)p.B * ns * 4; }
        } else if (w < 80) {
            if (w - 72 < nc) { m.act[j] = true; m.ptr[j] = (const char *)(p.cur_u + b * nc + (w - 72)); m.step[j] = (long)p.B * nc * 4; }
        } else if (w < 112) {
            if (w - 80 < ns) { m.act[j] = true; m.ptr[j] = (const char *)(p.c + b * p.c_sb + (w - 80)); m.step[j] = p.c_st * 4; }
        } else if (p.bound_mode == MPC_BOUND_TENSOR) {
            // words 112..119 u_lower_t, 120..127 u_upper_t: the pinned set is decided from the record (kkt_pinned_lds)
            const int a = (w - 112) & 7;
            if (a < nc) { m.act[j] = true; m.ptr[j] = (const char *)((w < 120 ? p.lo : p.hi) + b * nc + a); m.step[j] = (long)p.B * nc * 4; }
        }
    }
}
// The padded fused backward decides its pinned set from LDS: lane a < 8 of the wave looks at u*[a] and its bounds where the stage
// holds them (byte offsets of the three 8-word blocks), a ballot makes the two flag words kkt_pinned_word would have assembled from
// 8-24 scalar loads at the true n_ctrl -- those, live across the inlined sweep, spilled 364 scalar registers and put the masked kernel
// at 1.3x the unmasked one (the exact kernel: 1.1x).
MPC_DEV void kkt_pinned_lds(const P &p, unsigned u_off, unsigned lo_off, unsigned hi_off, int lane, unsigned &zlo, unsigned &zhi)
{
    const int a = lane & 7;
    const float u = wv::lds_f32(u_off + 4u * (unsigned)a);
    float lo = p.lo_s, hi = p.hi_s;
    if (p.bound_mode != MPC_BOUND_SCALAR) {
        lo = wv::lds_f32(lo_off + 4u * (unsigned)a);
        hi = wv::lds_f32(hi_off + 4u * (unsigned)a);
    }
    const bool pinned = (fabsf(u - lo) <= 1e-8f || fabsf(u - hi) <= 1e-8f) && a < p.nc && lane < 8;
    const unsigned m = (unsigned)wv::ballot(pinned);
    zlo = ((m & 1u) ? 1u : 0u) | ((m & 2u) ? 1u << 8 : 0u) | ((m & 4u) ? 1u << 16 : 0u) | ((m & 8u) ? 1u << 24 : 0u);
    zhi = ((m & 16u) ? 1u : 0u) | ((m & 32u) ? 1u << 8 : 0u) | ((m & 64u) ? 1u << 16 : 0u) | ((m & 128u) ? 1u << 24 : 0u);
}
MPC_DEV void rec_issue(const RecMap &m, long tl, long tf, long tx, unsigned off, bool skip_c = false, int lane = 0)
{
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long ti = m.kind[j] == 1 ? tf : (m.kind[j] == 2 ? tx : tl);
        wv::dma4_if(m.act[j] && !(skip_c && j == 0 && lane < 40), m.ptr[j] + ti * m.step[j], off + 256u * (unsigned)j);
    }
}
// every pass of the padded instantiation starts on cleared staging memory: the record words of padded entries are never
// written by the gather above (their lanes sit the instruction out), and the passes lay their slots out differently
MPC_DEV void pad_clear(int lane)
{
    if (!PADK) return;
    wv::lds_sync();
    for (unsigned off = 16u * (unsigned)lane; off + 16 <= LDS_TOTAL; off += 1024) wv::lds_store_f32x4(off, f32x4{0.f, 0.f, 0.f, 0.f});
    wv::lds_sync();
}
// ---- the caller's arrays by their true shape (padded instantiation) or by the kernel's (exact one) ----------------------------
MPC_DEV float ld_xinit(const P &p, long b, int i)
{
    if (PADK) return i < p.ns ? p.x_init[b * p.ns + i] : 0.f;
    return p.x_init[b * NS + i];
}
// four consecutive state / control entries i0 .. i0+3 of timestep-problem tb
// (padded instantiation: through a raw buffer over the ROW new_x[tb] / new_u[tb], whose range check drops the entries beyond the
// true shape -- `if (i < ns) store` put every store in a block of its own and hipcc an `s_waitcnt vmcnt(0)`, a drain of the
// staging gathers, in front of each)
MPC_DEV void st_x4(const P &p, long tb, int i0, f32x4 v)
{
    if (PADK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wv::st_buf_u(p.new_x + tb * p.ns, (unsigned)(4 * p.ns), (unsigned)(4 * (i0 + e)), v[e]);
    } else {
        wv::store_f32x4(p.new_x + tb * NS + i0, v);
    }
}
MPC_DEV void st_u4(const P &p, long tb, int a0, f32x4 v)
{
    if (PADK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wv::st_buf_u(p.new_u + tb * p.nc, (unsigned)(4 * p.nc), (unsigned)(4 * (a0 + e)), v[e]);
    } else {
        wv::store_f32x4(p.new_u + tb * NC + a0, v);
    }
}
// four consecutive entries i0 .. i0+3 of a caller's row of `len` floats (the fused backward's dF / df / dx_init rows): the entries beyond
// the true length are dropped by the buffer's range check (padded instantiation), one 16-byte store otherwise
MPC_DEV void st_row4(float *row, int len, int i0, f32x4 v)
{
    if (PADK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wv::st_buf_u(row, (unsigned)(4 * len), (unsigned)(4 * (i0 + e)), v[e]);
    } else {
        wv::store_f32x4(row + i0, v);
    }
}
// bounds of control a at (t, b) = tb; a padded control is unbounded in every mode (it sits at zero: its row of Quu is the identity)
MPC_DEV float bound_lo(const P &p, long tb, int a)
{
    if (PADK) {
        const int nc = p.nc;
        const float x = p.bound_mode != MPC_BOUND_SCALAR ? uniform_f32(p.lo + tb * nc + (a < nc ? a : nc - 1)) : p.lo_s;
        return a < nc ? x : -3e38f;
    }
    return p.bound_mode != MPC_BOUND_SCALAR ? uniform_f32(p.lo + tb * NC + a) : p.lo_s;
}
MPC_DEV float bound_hi(const P &p, long tb, int a)
{
    if (PADK) {
        const int nc = p.nc;
        const float x = p.bound_mode != MPC_BOUND_SCALAR ? uniform_f32(p.hi + tb * nc + (a < nc ? a : nc - 1)) : p.hi_s;
        return a < nc ? x : 3e38f;
    }
    return p.bound_mode != MPC_BOUND_SCALAR ? uniform_f32(p.hi + tb * NC + a) : p.hi_s;
}

// (round 4) c_ptr, f_ptr (and k_ptr, the workspace pointers of the rollouts) are WAVE-UNIFORM block addresses: `ptr + t * step`
// is scalar arithmetic, and this lane's 16-byte column `lo` joins in one 64-bit vector add per block.  Rounds 1-3 kept per-lane
// pointers: a 64-bit vector multiply-add (quarter rate) per block and timestep, ~47 clocks per DMA instruction issued against
// the 12/4 kernel's 21.  The record's lanes point into different arrays: their pointer stays per lane, the step is 32 bits
// (bytes per timestep < 4 GiB: checked by mfma40_supported) so that the product is ONE v_mad_u64_u32.
MPC_DEV unsigned long rec_off(long t, unsigned step) { return (unsigned long)(unsigned)t * (unsigned long)step; }
struct Stream {
    const char *c_ptr, *f_ptr;              // the problem's C_0 / F_0 (wave-uniform)
    const char *r_ptr;                      // this lane's 16-byte granule of the record, timestep 0
    long c_step, f_step;                    // bytes per timestep
    unsigned r_step, lo;                    // lo = 16 * lane
    bool r_active;
    bool r_is_f;                            // this lane's record granule is f_t (T-1 entries: indexed like F)
    // the padded instantiation stages by gather instead (PADK): wave-uniform block pointers + the per-lane maps
    const float *Cb, *Fb;                   // C_0 / F_0 of this problem
    Gather g;
    RecMap rm;
};

MPC_DEV void stream_init(Stream &d, const P &p, const Lane &L, const KktArgs40 *kk = nullptr, bool with_f = false)
{
    const long b = L.b;
    d.r_is_f = false;
    if (PADK) {
        d.Cb = p.C + b * p.C_sb;
        d.Fb = p.T > 1 ? p.F + b * p.F_sb : d.Cb;
        gather_init(d.g, p, L.lane);
        if (kk) rec_init_kkt(d.rm, p, L.lane, b, kk->dl_dx, kk->dl_du);
        else rec_init(d.rm, p, L.lane, b, true, false, with_f, nullptr);
    }
    d.lo = 16u * (unsigned)L.lane;
    d.c_ptr = (const char *)(p.C + b * p.C_sb);
    d.c_step = p.C_st * 4;
    // T = 1 has no dynamics (F may be NULL): its five DMA slots re-read C, so the wait counts stay the same
    d.f_ptr = p.T > 1 ? (const char *)(p.F + b * p.F_sb) : d.c_ptr;
    d.f_step = p.T > 1 ? p.F_st * 4 : 0;
    // record: lanes 0..9 -> c_t (160 B), 10..17 -> x_t (128 B), 18..19 -> u_t (32 B)
    d.r_active = L.lane < 20;
    if (L.lane < 10) {
        d.r_ptr = (const char *)(p.c + b * p.c_sb) + 16 * L.lane;
        d.r_step = (unsigned)(p.c_st * 4);
    } else if (L.lane < 18) {
        d.r_ptr = (const char *)(p.cur_x + b * NS) + 16 * (L.lane - 10);
        d.r_step = (unsigned)((long)p.B * NS * 4);
    } else {
        d.r_ptr = (const char *)(p.cur_u + b * NC) + 16 * ((L.lane < 20 ? L.lane : 18) - 18);
        d.r_step = (unsigned)((long)p.B * NC * 4);
    }
    if (with_f && L.lane >= 20 && L.lane < 28) {
        // the unvouched step verifies the nominal while it sweeps (sweep_wave): lanes 20..27 carry f_t
        d.r_active = true;
        d.r_is_f = true;
        d.r_ptr = (const char *)(p.f + b * p.f_sb) + 16 * (L.lane - 20);
        d.r_step = (unsigned)(p.f_st * 4);
    }
    if (MPC_QP_START && !kk && p.qp_start && p.bound_mode != MPC_BOUND_NONE && (L.lane == 28 || L.lane == 29)) {
        // mpc_lqr_options.qp_start: the caller's start of timestep t's box QP rides in the record, lanes 28..29 -> bytes 448..479
        // (the array may be the k this very sweep rewrites: stage t is fetched at least one timestep before timestep t stores its own)
        d.r_active = true;
        d.r_ptr = (const char *)(p.qp_start + b * p.qp_start_sb) + 16 * (L.lane - 28);
        d.r_step = (unsigned)(p.qp_start_st * 4);
    }
    if (kk) {
        // the fused backward: lanes 0..7 dl_dx_t, 8..9 dl_du_t (negated where they are read), 10..19 tau*_t as above,
        // 20..27 c_t[0..31] of the ORIGINAL problem (lambda's constant term)
        d.r_active = L.lane < 28;
        if (L.lane < 8) {
            d.r_ptr = (const char *)(kk->dl_dx + b * NS) + 16 * L.lane;
            d.r_step = (unsigned)((long)p.B * NS * 4);
        } else if (L.lane < 10) {
            d.r_ptr = (const char *)(kk->dl_du + b * NC) + 16 * (L.lane - 8);
            d.r_step = (unsigned)((long)p.B * NC * 4);
        } else if (L.lane >= 20) {
            d.r_ptr = (const char *)(p.c + b * p.c_sb) + 16 * ((L.lane < 28 ? L.lane : 20) - 20);
            d.r_step = (unsigned)(p.c_st * 4);
        }
    }
}

// The fused backward pins the controls that sit on a bound (to 1e-8, mpc/lqr_step.py:322-326) -- the u_zero_I of its nested
// solve -- straight from u* and the bounds: the same flag words zero_mask_word would read from a mask array.
MPC_DEV unsigned kkt_pinned_word(const P &p, long tb, int w)
{
    unsigned z = 0u;
    // (padded instantiation: u*, the bounds by the true n_ctrl; a control beyond it is free -- its row of Quu is the identity, it stays at zero)
    const int nc = PADK ? p.nc : NC;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int a = 4 * w + v, as = a < nc ? a : nc - 1;
        const float u = uniform_f32(p.cur_u + tb * nc + as);
        float lo = p.lo_s, hi = p.hi_s;
        if (p.bound_mode != MPC_BOUND_SCALAR) {
            lo = uniform_f32(p.lo + tb * nc + as);
            hi = uniform_f32(p.hi + tb * nc + as);
        }
        const bool pinned = (fabsf(u - lo) <= 1e-8f || fabsf(u - hi) <= 1e-8f) && a < nc;
        z |= pinned ? (1u << (8 * v)) : 0u;
    }
    return z;
}

// NB consecutive KiB of a block, source and destination alike: the immediate of an LDS-DMA moves both (gfx950; measured,
// tools/ubench/dma_offset_probe.hip), so four instructions share one pointer and one M0
template <int NB> MPC_DEV void dma_kib(const char *g, unsigned off)
{
    wv::dma16_at<0>(g, off);
    if (NB > 1) wv::dma16_at<1024>(g, off);
    if (NB > 2) wv::dma16_at<2048>(g, off);
    if (NB > 3) wv::dma16_at<3072>(g, off);
    if (NB > 4) dma_kib<(NB > 4 ? NB - 4 : 1)>(g + 4096, off + 4096);
}
// the same for a block this launch reads exactly ONCE (C in the sweep): streamed past the caches' replacement (round 4: the
// 420 MB of C otherwise push out the F blocks and gains the rollout is about to read again -- the 12/4 kernel's DMA_C)
template <int NB> MPC_DEV void dma_kib_once(const char *g, unsigned off)
{
    wv::dma16_once_at<0>(g, off);
    if (NB > 1) wv::dma16_once_at<1024>(g, off);
    if (NB > 2) wv::dma16_once_at<2048>(g, off);
    if (NB > 3) wv::dma16_once_at<3072>(g, off);
    if (NB > 4) dma_kib_once<(NB > 4 ? NB - 4 : 1)>(g + 4096, off + 4096);
}

MPC_DEV void stage_issue(const P &p, const Stream &d, const Lane &L, int t, int slot)
{
    const unsigned base = (unsigned)slot * STAGE_BYTES;
    const long tl = t;
    const long tf = t < p.T - 1 ? t : (p.T > 1 ? p.T - 2 : 0);
    if (PADK) {
        // (T = 1 has no dynamics: its F instructions gather an empty block -- zeros --, the wait counts stay the same)
        gather_C(d.g, d.Cb + tl * p.C_st, base + OFF_C, L.lane);
        gather_F(d.g, p.T > 1 ? d.Fb + tf * p.F_st : d.Cb, p.T > 1 ? d.g.fbytes : 0u, base + OFF_F);
        rec_issue(d.rm, tl, tf, tl, base + OFF_R);
        return;
    }
    dma_kib_once<6>(d.c_ptr + tl * d.c_step + d.lo, base + OFF_C);
    wv::dma16_at_if<2048>(L.lane < 16, d.c_ptr + tl * d.c_step + 4096 + d.lo, base + OFF_C + 4096);
    dma_kib<5>(d.f_ptr + (p.T > 1 ? tf * d.f_step : 0) + d.lo, base + OFF_F);
    wv::dma16_if(d.r_active, d.r_ptr + rec_off(d.r_is_f ? tf : tl, d.r_step), base + OFF_R);
}

// butterfly sums: across the four lane groups (lanes differing in bits 4,5), across the 16 lanes of a group
MPC_DEV float sum_q(float x) { return wv::sum_rows(x); }
MPC_DEV float sum_r(float x)
{
    x += wv::shfl_xor(x, 1);
    x += wv::shfl_xor(x, 2);
    x += wv::shfl_xor(x, 4);
    x += wv::shfl_xor(x, 8);
    return x;
}

// ---- LDL' of the SPD 8x8 Quu with the matrix spread over lanes ----------------------------------------------------
// All four 16-lane rows hold the same data; inside a row lane a < 8 holds matrix row a.  col[c] = column c of the
// symmetric matrix (lane a: A[a][c]).  Right-looking LDL': per pivot one reciprocal, one scaling and one
// v_fmac_f32_dpp per remaining column (60 instructions against ~170 on wave-uniform values, and no readlanes to make
// the 36 entries uniform first).  The triangular solves of per-lane right-hand sides read L[a][c] as a DPP broadcast
// of lane a of nl[c] = -L[:, c] -- the same 64 multiply-adds as with scalar operands.
struct Ldl8V {
    float nl[8];     // lane a: -L[a][c]  (valid for a > c)
    float inv[8];    // 1 / D_c in every lane
    float sing;      // != 0: a pivot was exactly zero and its unknown dropped out (the reference's pinverse of a Quu with
                     // a zero row and column, mpc/lqr_step.py:88-94; see pivot_inv in lqr_small_math.h)
};
template <int C, int M> struct Ldl8VElim {
    static MPC_DEVM void run(float (&col)[8], float nlc)
    {
        // (col[C] was last written by the previous pivot's first elimination, five or more instructions back: the DPP read needs
        // no wait states of its own)
        wv::fmac_bcast_settled<M>(col[M], col[C], nlc);     // A[a][m] -= L[a][c] A[m][c]
        Ldl8VElim<C, M + 1>::run(col, nlc);
    }
};
template <int C> struct Ldl8VElim<C, 8> { static MPC_DEVM void run(float (&)[8], float) {} };
template <int C, bool PINV> struct Ldl8VPivot {
    static MPC_DEVM void run(Ldl8V &f, float (&col)[8])
    {
        const float d = wv::bcast<C>(col[C]);
        float inv = wv::rcp(d);           // (unrefined v_rcp_f32 here: 280.1 against 280.8 us at config 5 -- not taken)
        if (PINV) {                                         // a zero pivot drops out (the struct's comment)
            const bool ok = d != 0.f;
            f.sing = ok ? f.sing : 1.f;
            inv = ok ? inv : 0.f;
        }
        f.inv[C] = inv;
        f.nl[C] = -(col[C] * f.inv[C]);
        Ldl8VElim<C, C + 1>::run(col, f.nl[C]);
        Ldl8VPivot<C + 1, PINV>::run(f, col);
    }
};
template <bool PINV> struct Ldl8VPivot<8, PINV> { static MPC_DEVM void run(Ldl8V &, float (&)[8]) {} };
// PINV: the unconstrained solve, where the reference takes a pseudo-inverse; the box QP's factorisations (H + 1e-11 I
// on the free set, the reference's LU) are taken as they come
template <bool PINV = false> MPC_DEV void ldl8v(Ldl8V &f, float (&col)[8]) { f.sing = 0.f; Ldl8VPivot<0, PINV>::run(f, col); }

// z_I -= sum_{k < I} L[I][k] z_k
template <int I, int K> struct Ldl8VFwdRow {
    static MPC_DEVM void run(const Ldl8V &f, float (&z)[8])
    {
        wv::fmac_bcast_settled<I>(z[I], f.nl[K], z[K]);
        Ldl8VFwdRow<I, K + 1>::run(f, z);
    }
};
template <int I> struct Ldl8VFwdRow<I, I> { static MPC_DEVM void run(const Ldl8V &, float (&)[8]) {} };
template <int I> struct Ldl8VFwd {
    static MPC_DEVM void run(const Ldl8V &f, float (&z)[8])
    {
        Ldl8VFwdRow<I, 0>::run(f, z);
        Ldl8VFwd<I + 1>::run(f, z);
    }
};
template <> struct Ldl8VFwd<8> { static MPC_DEVM void run(const Ldl8V &, float (&)[8]) {} };
// y_I = z_I / D_I - sum_{k > I} L[k][I] y_k
template <int I, int K> struct Ldl8VBwdRow {
    static MPC_DEVM void run(const Ldl8V &f, float (&y)[8])
    {
        wv::fmac_bcast_settled<K>(y[I], f.nl[I], y[K]);
        Ldl8VBwdRow<I, K + 1>::run(f, y);
    }
};
template <int I> struct Ldl8VBwdRow<I, 8> { static MPC_DEVM void run(const Ldl8V &, float (&)[8]) {} };
template <int I> struct Ldl8VBwd {
    static MPC_DEVM void run(const Ldl8V &f, float (&y)[8])
    {
        y[I] *= f.inv[I];
        Ldl8VBwdRow<I, I + 1>::run(f, y);
        Ldl8VBwd<I - 1>::run(f, y);
    }
};
template <> struct Ldl8VBwd<-1> { static MPC_DEVM void run(const Ldl8V &, float (&)[8]) {} };
// y = S^-1 y, per lane
MPC_DEV void ldl8v_solve(const Ldl8V &f, float (&y)[8])
{
    Ldl8VFwd<1>::run(f, y);
    Ldl8VBwd<7>::run(f, y);
}

// ---- Gauss-Jordan on the same lane-spread matrix: the factorisation of the box QP's trips (round 4) ----------------
// A trip of the QP solves ONE system whose right-hand side is spread over lanes like the matrix (lane a: entry a).  With
// LDL' that costs the factorisation, eight broadcasts to give every lane the whole right-hand side, 64 multiply-adds of
// triangular solves on it and eight selects to spread the answer again.  Eliminating ABOVE the pivot as well (the same
// instruction: lane a < c simply keeps a live multiplier) lets the right-hand side ride along as a ninth column -- one
// DPP multiply-add per pivot -- and leaves the solution where it is wanted: 44 instructions in place of ~90.  The
// multipliers are a factorisation too: the K solve behind the QP applies them to its per-lane right-hand sides at the
// price of the triangular solves (56 + 8 against 64 + 8).  SPD matrix, same pivots as the LDL' -- no pivoting needed.
struct Gj8V {
    float nl[8];     // lane a: -(A[a][c] / d_c) as it stood at pivot c (the multiplier of row a), 0 in lane c itself
    float inv[8];    // 1 / d_c in every lane
};
template <int C> struct Gj8VPivot {
    static MPC_DEVM void run(Gj8V &f, float (&col)[8], float &rhs, float &invd, int r)
    {
        // (v_rcp_f32 unrefined, as in the 12/4 kernel's QP: the eight pivots are one dependent chain, and the Newton step of
        // wv::rcp was two more links of it per pivot -- 1 ulp against 1/2 in a solve that stops at |dx| < 1e-4)
        const float inv = wv::rcp_fast(wv::bcast<C>(col[C]));
        f.inv[C] = inv;
        const float t = -(col[C] * inv);
        f.nl[C] = r == C ? 0.f : t;
        invd = r == C ? inv : invd;
        // (the unpivoted block stays symmetric: A[c][m] is lane m of column c, as in ldl8v)
        Ldl8VElim<C, C + 1>::run(col, f.nl[C]);
        rhs = fmaf(wv::bcast<C>(rhs), f.nl[C], rhs);        // rhs[a] -= l[a] rhs[c]  (rhs was written by the previous pivot)
        Gj8VPivot<C + 1>::run(f, col, rhs, invd, r);
    }
};
template <> struct Gj8VPivot<8> { static MPC_DEVM void run(Gj8V &, float (&)[8], float &, float &, int) {} };
// col: the matrix (destroyed); rhs: lane a holds entry a, on return entry a of the solution
MPC_DEV void gj8v(Gj8V &f, float (&col)[8], float &rhs, int r)
{
    float invd = 0.f;
    Gj8VPivot<0>::run(f, col, rhs, invd, r);
    rhs *= invd;
}
template <int C, int A> struct Gj8VApplyRow {
    static MPC_DEVM void run(const Gj8V &f, float (&z)[8])
    {
        if (A != C) wv::fmac_bcast_settled<A>(z[A], f.nl[C], z[C]);
        Gj8VApplyRow<C, A + 1>::run(f, z);
    }
};
template <int C> struct Gj8VApplyRow<C, 8> { static MPC_DEVM void run(const Gj8V &, float (&)[8]) {} };
template <int C> struct Gj8VApply {
    static MPC_DEVM void run(const Gj8V &f, float (&z)[8])
    {
        Gj8VApplyRow<C, 0>::run(f, z);
        Gj8VApply<C + 1>::run(f, z);
    }
};
template <> struct Gj8VApply<8> { static MPC_DEVM void run(const Gj8V &, float (&)[8]) {} };
// z = A^-1 z for per-lane right-hand sides (every lane its own), from the multipliers of gj8v
MPC_DEV void gj8v_solve(const Gj8V &f, float (&z)[8])
{
    Gj8VApply<0>::run(f, z);
#pragma unroll
    for (int a = 0; a < 8; ++a) z[a] *= f.inv[a];
}
#ifdef MPC_MFMA40_QP_LDL          // (the round-3 form of the QP's linear algebra, kept for A/B timing)
using QpFac = Ldl8V;
#else
using QpFac = Gj8V;
#endif

MPC_DEV float clampf(float x, float lo, float hi)
{
    if (x < lo) x = lo;      // util.eclamp (mpc/util.py:56-70): strict compares, bound written exactly
    if (x > hi) x = hi;
    return x;
}

// The box of a rollout's four controls 4q .. 4q+3 at timestep tb (mpc/lqr_step.py:200-213): the bounds (scalars, or rows of
// the bound tensors) narrowed by the trust region u +- delta_u, and the clamp -- straight-line code (round 4).  Rounds 1-3
// wrote this per control inside `if (MODE == 2 && q < 2) { if (tensor bounds) ..; if (has_delta) ..; clamp }`: an exec-mask
// branch and two uniform ones per control, four times a timestep -- the priced rollout spent 1,300 clocks a timestep there
// (profiles/r04_prof_phases40.log).  max / min pick the same numbers as the reference's compare-and-select chain.
struct Box4 { float lo[4], hi[4]; };
MPC_DEV void box4(Box4 &bx, const P &p, long tb, const Lane &L, const wv::f32x4 &ub)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        bx.lo[v] = p.lo_s;
        bx.hi[v] = p.hi_s;
    }
    if (PADK && p.bound_mode == MPC_BOUND_SCALAR) {
        // (round 6) padded instantiation, scalar bounds: a control beyond n_ctrl is unbounded -- decided per lane group on the vector ALU
        // (as sixteen scalar selects the bounds lived in scalar registers across the rollout's timestep and spilled there)
        const int nc = p.nc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool in = 4 * (L.q & 1) + v < nc;
            bx.lo[v] = in ? p.lo_s : -3e38f;
            bx.hi[v] = in ? p.hi_s : 3e38f;
        }
    } else if (PADK || p.bound_mode != MPC_BOUND_SCALAR) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            bx.lo[v] = pick(L.q == 0, bound_lo(p, tb, v), bound_lo(p, tb, 4 + v));
            bx.hi[v] = pick(L.q == 0, bound_hi(p, tb, v), bound_hi(p, tb, 4 + v));
        }
    }
    const float dlt = p.has_delta ? p.delta_u : 3.0e38f;           // (no trust region: u -+ 3e38 never narrows a bound)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        bx.lo[v] = fmaxf(bx.lo[v], ub[v] - dlt);                    // :202-207
        bx.hi[v] = fminf(bx.hi[v], ub[v] + dlt);
    }
}
// A contraction over the eight controls on the 16x16x4 MFMA: lane group q supplies k = q of each instruction, so two
// instructions cover the controls when group q carries control uctl(q) + 2h in instruction h = 0, 1 (0 4 1 5 | 2 6 3 7) -- the
// order wv::lower_halves gives to D-layout registers 2h, 2h + 1 (controls 4q + v of groups q = 0, 1).  Four instructions with
// half their k on padding in rounds 1-3.
MPC_DEV int uctl(int q) { return 4 * (q & 1) + (q >> 1); }
MPC_DEV float box_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }       // util.eclamp, lo first

// ---- Projected-Newton box QP in 8 unknowns (mpc/pnqp.py:5-82, n_batch = 1; the Armijo restatements of lqr_small_math.h's
// pnqp4) with its vectors spread over lanes (lane a < 8 of every 16-lane row: entry a; lanes 8..15 hold
// zeros and stay free, which makes them inert in every product and row sum).  col0[c]: column c of H (lane a: H[a][c]).
// A trip is ~210 instructions against ~500 on wave-uniform values: the gradient and H d are 8 DPP multiply-adds each, the
// tests / clamps / steps one instruction per vector, the factorisation is ldl8v; only the triangular solve keeps its 64
// multiply-adds (every lane solves the same right-hand side, read as DPP broadcasts).
// Same iterates as mpc/pnqp.py:5-82 (n_batch = 1).  On return xv is the solution, mv the free set (1 / 0) and f
// the factorisation of the iteration that recognised convergence (:56-59), as the reference returns them.
template <int C> struct Pnqp8vMat {
    static MPC_DEVM void run(const float (&col0)[8], float mnv, float dgv, int r, float (&colm)[8])
    {
        colm[C] = (mnv * wv::bcast<C>(mnv)) * col0[C];
        colm[C] = r == C ? dgv : colm[C];
        Pnqp8vMat<C + 1>::run(col0, mnv, dgv, r, colm);
    }
};
template <> struct Pnqp8vMat<8> { static MPC_DEVM void run(const float (&)[8], float, float, int, float (&)[8]) {} };
template <int C> struct Pnqp8vMv {          // acc += H v  (v spread over lanes)
    static MPC_DEVM void run(const float (&col0)[8], float v, float &acc)
    {
        // (v may have been written by the instruction in front of the first product: that one waits out the DPP read-after-write
        // states; the seven that follow it on the accumulator's chain find v settled)
        if (C == 0) wv::fmac_bcast<C>(acc, v, col0[C]);
        else wv::fmac_bcast_settled<C>(acc, v, col0[C]);
        Pnqp8vMv<C + 1>::run(col0, v, acc);
    }
};
template <> struct Pnqp8vMv<8> { static MPC_DEVM void run(const float (&)[8], float, float &) {} };
// out[a] = in[a] * (entry a of the 0 / 1 vector m): rows off the free set drop out of a per-lane right-hand side / solution
template <int A> MPC_DEVM void free_rows(float m, const float (&in)[8], float (&out)[8])
{
    if constexpr (A < 8) {
        out[A] = in[A] * wv::bcast<A>(m);
        free_rows<A + 1>(m, in, out);
    }
}
template <int A> struct Pnqp8vSpread {      // y[a] = entry a of the row's vector v, in every lane
    static MPC_DEVM void run(float v, float (&y)[8])
    {
        y[A] = wv::bcast<A>(v);
        Pnqp8vSpread<A + 1>::run(v, y);
    }
};
template <> struct Pnqp8vSpread<8> { static MPC_DEVM void run(float, float (&)[8]) {} };
// lane a < 8 picks y[a] (y is the same in all lanes), lanes 8..15 get 0
MPC_DEV float gather8(const float (&y)[8], int r)
{
    float v = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) v = pick(r == a, y[a], v);
    return v;
}
MPC_DEV int pnqp8v(const float (&col0)[8], float diagv, float qv, float lbv, float ubv, int n_iter, int r, float &xv,
                   float &mv, QpFac &f, bool &converged)
{
    int it_ret = n_iter - 1;
    converged = false;
    const float dgm = diagv + 1e-11f - 1.f;                        // :47 (identity off the free set)
    bool full = false;                                             // the last trip took its whole Newton step
    for (int it = 0; it < n_iter; ++it) {
        float gv = qv;
        Pnqp8vMv<0>::run(col0, xv, gv);                            // :29
        // :32 clamped = (x == lb & g > 0) | (x == ub & g < 0)
        const float r_lo = (xv == lbv) ? gv : -1.f;
        const float r_hi = (xv == ubv) ? -gv : -1.f;
        const float mnv = (fmaxf(r_lo, r_hi) > 0.f) ? 0.f : 1.f;
        // The confirming iteration costs a gradient, not a factorisation (as in pnqp4_rows, lqr_dpp16_body.h): after a FULL
        // Newton step the gradient vanishes on the free set, so if the free set of the new point is the one just factorised
        // the next step is zero to rounding -- this is the iteration the reference stops in (:56-59), and the factorisation
        // it would recompute is the one at hand.
        if (it > 0 && full && !wv::any(mnv != mv)) {             // (a ballot: every row of the wave holds the same QP)
            converged = true;
            it_ret = it;
            break;
        }
        float colm[8];
        Pnqp8vMat<0>::run(col0, mnv, fmaf(mnv, dgm, 1.f), r, colm);   // :44-48
#ifdef MPC_MFMA40_QP_LDL
        Ldl8V fn;
        ldl8v(fn, colm);
        float y[8];
        Pnqp8vSpread<0>::run(mnv * gv, y);
        ldl8v_solve(fn, y);                                        // :50-54
        const float dxv = -(mnv * gather8(y, r));
#else
        // (rows off the free set are identity rows with a zero right-hand side: their entry of the solution is 0 as it comes)
        Gj8V fn;
        float sol = mnv * gv;
        gj8v(fn, colm, sol, r);                                    // :50-54
        const float dxv = -sol;
#endif
        const float nrm2 = wv::row_sum8(dxv * dxv);       // (the vectors live in lanes 0..7; lane 0 reads the sums)
        float mxv = xv + dxv;
        const bool outside = wv::any((mxv < lbv) | (mxv > ubv));
        f = fn;
        mv = mnv;
        if (wv::first_lane(!(nrm2 >= 1e-8f))) {                    // :56-59
            converged = true;
            it_ret = it;
            break;
        }
        full = !outside;
        if (!full) {                                               // :61-76
            float alpha = 1.f;
            for (int count = 0; count < 10; ++count) {
                mxv = clampf(fmaf(alpha, dxv, xv), lbv, ubv);
                const float dv = mxv - xv;
                float hdv = 0.f;
                Pnqp8vMv<0>::run(col0, dv, hdv);
                const float den = wv::row_sum8(-gv * dv), dhd = wv::row_sum8(dv * hdv);
                const float arm = fmaf(-0.5f, dhd, den) * wv::rcp_fast(den);
                if (wv::first_lane(arm <= 0.1f)) alpha *= 0.1f; else break;
            }
        }
        xv = mxv;                                                  // :78
    }
    return it_ret;
}

// The sweep of one problem: K [T,B,8,32] and k [T,B,8] in the reference layout, old_costs[b].
// MODE 0: unconstrained; 1: u_zero_I mask; 2: box constraints (pnqp8v).
// V_t (the four tiles as they sit in the registers: pass 2 reads them back as A operands) and v_t | g_t (column layout:
// lane group q holds entries 16I + 4q + v) to the fused backward's workspace
MPC_DEV void kkt_store_vvg(const KktArgs40 &kx, long tb, const Lane &L, const wv::f32x4 (&Vd)[2][2], const float (&vcol)[2][4],
                           const float (&gcol)[2][4])
{
#ifdef MPC_KF40_NOVS              // (diagnostic build: what the V stores cost)
    if (tb < 0)
#endif
    // (round 4) V is symmetric: tile (0, 1) is the transpose of tile (1, 0) and stays home -- pass 2 reads it out of the staged
    // (1, 0) tile with the indices swapped (`-DMPC_KF40_VFULL`: all four tiles, as in round 3).  3 KiB per problem-step each way
    // instead of 4 in a kernel that moves 1.9 GB at 5 TB/s.
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) {
#ifndef MPC_KF40_VFULL
            if (I == 0 && J == 1) continue;
#endif
            wv::store_f32x4(kx.Vws + tb * 1024 + (2 * I + J) * 256 + 4 * L.lane, Vd[I][J]);
        }
    if (L.r == 0) {
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            wv::store_f32x4(kx.vgws + tb * 64 + 16 * I + 4 * L.q, wv::f32x4{vcol[I][0], vcol[I][1], vcol[I][2], vcol[I][3]});
            wv::store_f32x4(kx.vgws + tb * 64 + 32 + 16 * I + 4 * L.q, wv::f32x4{gcol[I][0], gcol[I][1], gcol[I][2], gcol[I][3]});
        }
    }
}

// KKT (the fused backward, kkt_fused_wave): the nested problem of LQRStepFn.backward -- zero nominal, c = -(dl_dx | dl_du),
// pinned controls from u* and the bounds (MODE 1) -- with two more vector recursions riding along:
//   lambda_t = C_x tau*_t + c_x + F_x' lambda_{t+1}   (:355-369; nothing the nested solve produces is in it)
//   g_t = F_x' g_{t+1} - Qxu k_t                      (the correction of dlambda when the nested line search ends below 1)
// V_t, v_t, g_t go to the workspace for pass 2 (dlambda_t = V_t dx_t + v_t + (1 - alpha) g_t), lambda_{t+1} into the
// first 32 words of the dF_t block.  v_0, g_0 (row layout) come back through v0g0 for dx_init.
// Diagnostic builds only (-DMPC_MFMA40_PROF, tools/prof_phases40.py): shader-clock totals of the sweep's phases per wave,
// written over K behind the rollout.  Every probe drains the LDS queue: the phases are serialised, upper bounds each.
//   0 DMA wait | 1 C, tau, c from LDS + c_back + nominal cost | 2 F from LDS + Y, Q, q (MFMA) | 3 Quu out of the tile, bounds,
//   factorisation / box QP | 4 K solve, M, gain stores | 5 value update + records | 6 sweep tail | 7 rollout | 8 DMA issue
#ifdef MPC_MFMA40_PROF
struct Prof40 { unsigned long long acc[16], last; };
#define PROF40_MARK(slot) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = wv::clock(); \
                               prof->acc[slot] += n_ - prof->last; prof->last = n_; } while (0)
#else
struct Prof40;
#define PROF40_MARK(slot)
#endif
template <int MODE, bool KKT = false>
MPC_DEV double sweep_wave(const P &p, float *Kout, float *kout, double *w0_out = nullptr, const KktArgs40 *kx = nullptr,
                          float *v0g0 = nullptr, bool *on_dynamics_out = nullptr, Prof40 *prof = nullptr)
{
    Lane L;
    L.lane = wv::lane();
    L.r = L.lane & 15;
    L.q = L.lane >> 4;
    L.b = wv::problem();
    if (L.b >= p.B) return 0.0;
    const int T = p.T;
    // An unconstrained step whose caller does not vouch for the nominal (a bare LQRStep call) checks it here: x_0 = x_init and
    // x_{t+1} = F_t tau_t + f_t to 1e-5 (1 + |x|), the 4-problems-per-wave kernel's test.  F_t and tau_t are in registers for
    // the sweep anyway: 24 multiply-adds and eight 16-lane sums per timestep.  A nominal that passes takes the lean rollout
    // (the line search decided from the sweep, no pricing pass over C): the bare call costs 7 % more than the vouched one
    // instead of 33 %; one that fails is priced from C like the reference and reports MPC_ST_NOMINAL_OFF_DYNAMICS.
    const bool verify = MODE == 0 && !KKT && !p.on_dynamics && on_dynamics_out != nullptr;
    float offdyn = 0.f;                // largest violation this lane has seen (> 0: off the dynamics)
    float xnext[2][4];                 // the nominal state of the timestep worked on last (t + 1), column layout
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int v = 0; v < 4; ++v) xnext[I][v] = 0.f;
    Stream d;
    stream_init(d, p, L, KKT ? kx : nullptr, verify && p.f != nullptr && T > 1);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 Vd[2][2];                    // V, D layout
    float vcol[2][4];                  // v, column layout: vcol[I][v] = v[16I + 4q + v]
    float lcol[2][4], gcol[2][4];      // lambda_{t+1}, g_{t+1} in the same layout (KKT)
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
        for (int J = 0; J < 2; ++J) Vd[I][J] = zero4;
#pragma unroll
        for (int v = 0; v < 4; ++v) vcol[I][v] = lcol[I][v] = gcol[I][v] = 0.f;
    }
    double old_cost = 0.0;
    double w0 = 0.0;                   // sum_t 0.5 qu'k: the value function's predicted change of the cost (unconstrained)
    int qp_total = 0, status = 0;
    float asym = 0.f, cmax = 0.f;      // symmetry test of C (see the tile loads below)
    bool warm = false;
    float kprev_v = 0.f;               // box QP: the previous timestep's solution, spread over lanes (warm start)

    pad_clear(L.lane);
    // the identity on the padded diagonal of Quu (padded instantiation): register v of lane (q, r) of tile (2, 2) is
    // Quu[4q + v][r] -- a control beyond n_ctrl gets H = 1, q = 0, no bounds: it stays at zero and its gains are zero
    float padd[4] = {0.f, 0.f, 0.f, 0.f};
    if (PADK) {
#pragma unroll
        for (int v = 0; v < 4; ++v) padd[v] = (L.q < 2 && 4 * L.q + v == L.r && L.r >= p.nc) ? 1.f : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SNSTAGE - 1; ++i) stage_issue(p, d, L, T - 1 - i >= 0 ? T - 1 - i : 0, i);
    int slot = 0;
    for (int t = T - 1; t >= 0; --t) {
        // keep the next timestep(s) in flight (re-loading step 0 at the tail keeps the wait count fixed)
        stage_issue(p, d, L, t - (SNSTAGE - 1) >= 0 ? t - (SNSTAGE - 1) : 0, (slot + SNSTAGE - 1) % SNSTAGE);
        PROF40_MARK(8);
        wv::dma_wait<(SNSTAGE - 1) * DMA_PER_STAGE>();
        PROF40_MARK(0);
        const unsigned base = (unsigned)slot * STAGE_BYTES;
        const long tb = (long)t * p.B + L.b;
        // (V, v, g) of t+1 leave now, not when they were finished: vector stores share the counter the wait above counts
        // the DMAs with, and a store issued at the end of a timestep would be waited out at the top of the next one
        if (KKT && t < T - 1) kkt_store_vvg(*kx, tb + p.B, L, Vd, vcol, gcol);

        // ---- C in D layout (all nine tiles), tau in both layouts, c in row layout
        // Tile (I, J), registers v = 0..3 of lane (q, r) = C[16I + 4q + v][16J + r] -- read through C's symmetry as
        // C[16J + r][16I + 4q .. + 3]: four consecutive floats of one row, ONE 16-byte LDS read per tile instead of four
        // 4-byte reads (rows are 160 B, so every such quadruple is 16-byte aligned)
        f32x4 Qd[3][3];
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J < 3; ++J) {
                // only tile row / column 2 has padding (rows, columns 40..47)
                const bool in = (I < 2 || L.q < 2) && (J < 2 || L.r < 8);
                // (a padding lane reads on -- into the next row, or F's block of the stage -- and is zeroed below: one lane address + compile-time offsets)
                const f32x4 x = wv::lds_f32x4(base + OFF_C + 4u * (unsigned)(L.r * N + 4 * L.q) + 4u * (unsigned)(16 * J * N + 16 * I));
#pragma unroll
                for (int v = 0; v < 4; ++v) Qd[I][J][v] = in ? x[v] : 0.f;
            }
        // The tiles above hold C[16J + r][16I + 4q + v] where the layout wants C[16I + 4q + v][16J + r]: the same number
        // only if C_t is symmetric, which the reference does not require (mpc/lqr_step.py:68, 294).  Fetch the true entry
        // of the upper-triangle tiles (every unordered pair once) and keep the largest difference, with the largest
        // entry as the scale: MPC_ST_C_ASYMMETRIC at the end of the sweep.  Skipped when the caller vouches for C.
        if (!p.c_symmetric) {
            // (round 5: ONE lane address and compile-time offsets -- a lane of the padding, rows / columns 40..47 of the third tile row /
            // column, reads on into F's block of the stage and is masked in the compare, not in the address: 24 address selects and
            // adds a timestep less; the maxima as v_max3 with |.| modifiers, two entries an instruction)
            const unsigned a0 = base + OFF_C + 4u * (unsigned)(4 * L.q * N + L.r);
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int J = I; J < 3; ++J) {
                    const bool in = (I < 2 || L.q < 2) && (J < 2 || L.r < 8);
                    float dq[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float tr = wv::lds_f32(a0 + 4u * (unsigned)((16 * I + v) * N + 16 * J));
                        const float dd = Qd[I][J][v] - tr;
                        dq[v] = (J < 2) ? dd : (in ? dd : 0.f);          // (I = 2 means J = 2: only the third tile column has padding here)
                    }
                    wv::absmax3(asym, dq[0], dq[1]);
                    wv::absmax3(asym, dq[2], dq[3]);
                    wv::absmax3(cmax, Qd[I][J][0], Qd[I][J][1]);
                    wv::absmax3(cmax, Qd[I][J][2], Qd[I][J][3]);
                }
        }
        float tcol[3][4], trow[3], crow[3];
#pragma unroll
        for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 16 * I + 4 * L.q + v;
                const float x = wv::lds_f32(base + OFF_R + 160 + 4u * (unsigned)i);        // (i up to 47: still inside the record, zeroed below)
                tcol[I][v] = i < N ? x : 0.f;
            }
#pragma unroll
        for (int J = 0; J < 3; ++J) {
            const int j = 16 * J + L.r;
            const float x = wv::lds_f32(base + OFF_R + 160 + 4u * (unsigned)j);
            const float c = wv::lds_f32(base + OFF_R + 4u * (unsigned)j);
            trow[J] = j < N ? x : 0.f;
            crow[J] = j < N ? (KKT ? -c : c) : 0.f;
        }
        // c_back = C tau + c (mpc/lqr_step.py:289-295) in row layout, and the nominal cost (:169)
        // Sums over the four lane groups are deferred: every vector below is kept as this lane group's PARTIAL sum (qpart,
        // lpart, gpart: the products of its rows 16I + 4q + v) and reduced once, where its value is first needed -- q_u in
        // front of the control block, q_x not before v = q_x + Qxu k (one reduction for c_back, F'v and Qxu k together;
        // round 2 reduced each of them: eight two-swap butterflies per timestep, now three).
        float qpart[3];
        float lpart[2] = {0.f, 0.f}, gpart[2] = {0.f, 0.f};
#pragma unroll
        for (int J = 0; J < 3; ++J) {
            qpart[J] = 0.f;
            if (KKT && J == 2) continue;
            float s = 0.f;
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) s = fmaf(Qd[I][J][v], tcol[I][v], s);
            // (the nested nominal of the fused backward is zero: c_back = c = -r; tau there is tau*, C tau* + c_x starts lambda_t)
            if (KKT) lpart[J] = s;
            else qpart[J] = s;                               // this lane group's share of (C tau)[16J + r]
        }
        if (!KKT) {
            // nominal cost (:169): tau'(C tau / 2 + c) -- with the unreduced shares the sum runs over all 64 lanes after the loop
            float s = 0.f;
#pragma unroll
            for (int J = 0; J < 3; ++J) s = fmaf(trow[J], fmaf(0.5f, qpart[J], L.q == 0 ? crow[J] : 0.f), s);
            old_cost += (double)s;
        }
        PROF40_MARK(1);

        if (t < T - 1) {
            // ---- F as FB[(I',v)][J] = F[16I' + 4q + v][16J + r]
            float FB[8][3];
#pragma unroll
            for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int J = 0; J < 3; ++J) {
                        const int m = 16 * Ip + 4 * L.q + v, col = 16 * J + L.r;
                        const bool in = J < 2 || L.r < 8;
                        const float x = wv::lds_f32(base + OFF_F + 4u * (unsigned)(m * N + col));       // (col up to 47: the next row, zeroed below)
                        FB[4 * Ip + v][J] = in ? x : 0.f;
                    }
            if (verify) {
                // F_t tau_t + f_t against x_{t+1}: rows 16I' + 4q + v of F times tau (row layout), summed over the 16 lanes
#pragma unroll
                for (int Ip = 0; Ip < 2; ++Ip) {
                    f32x4 ft = zero4;
                    if (p.f) ft = wv::lds_f32x4(base + OFF_R + 320 + 4u * (unsigned)(16 * Ip + 4 * L.q));
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float s = 0.f;
#pragma unroll
                        for (int J = 0; J < 3; ++J) s = fmaf(FB[4 * Ip + v][J], trow[J], s);
                        const float pred = wv::row_sum(s) + ft[v];
                        float r = fabsf(pred - xnext[Ip][v]) - 1e-5f * (1.f + fabsf(xnext[Ip][v]));
                        r = (r == r) ? r : 1.f;
                        offdyn = r > offdyn ? r : offdyn;
                    }
                }
            }
            // ---- Y = V F   (A operand = V by symmetry: register v of tile (I', Im))
            // (round 5, tried: c_back and q = c_back + F'v issued INSIDE this region, interleaved with the MFMAs by scheduling groups --
            // one MFMA, one or two vector instructions --: 281 -> 283-292 us.  The blocks stay undivided.)
            wv::sched_fence();
            // (three tiles at a time, their accumulation chains interleaved: an MFMA that waits for the previous one's
            // accumulator issues every 40 clocks, an independent one every 32)
            f32x4 Yd[2][3];
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
#pragma unroll
                for (int J = 0; J < 3; ++J) Yd[Im][J] = zero4;
#pragma unroll
                for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int J = 0; J < 3; ++J) Yd[Im][J] = wv::mfma(Vd[Ip][Im][v], FB[4 * Ip + v][J], Yd[Im][J]);
            }
            // ---- Q = C + F'Y  (A operand = F' = FB, B operand = Y in D layout); tiles (0,2), (1,2) are
            // not needed below (Qxu is used through Qux)
#pragma unroll
            for (int I = 0; I < 3; ++I)
#pragma unroll
                for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int J = 0; J < 3; ++J) {
                            if (J == 2 && I < 2) continue;
                            Qd[I][J] = wv::mfma(FB[4 * Ip + v][I], Yd[Ip][J][v], Qd[I][J]);
                        }
            wv::sched_fence();
            // ---- q = c_back + F'v
#pragma unroll
            for (int J = 0; J < 3; ++J) {
                float s = qpart[J];
#pragma unroll
                for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
                    for (int v = 0; v < 4; ++v) s = fmaf(FB[4 * Ip + v][J], vcol[Ip][v], s);
                qpart[J] = s;
            }
            if (KKT) {
                // lambda_t += F_x' lambda_{t+1}, g_t = F_x' g_{t+1} (- Qxu k_t below); lambda_{t+1} into dF_t's block
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    float sl = lpart[J], sg = 0.f;
#pragma unroll
                    for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            sl = fmaf(FB[4 * Ip + v][J], lcol[Ip][v], sl);
                            sg = fmaf(FB[4 * Ip + v][J], gcol[Ip][v], sg);
                        }
                    lpart[J] = sl;
                    gpart[J] = sg;
                }
                if (L.r == 0) {
#pragma unroll
                    for (int I = 0; I < 2; ++I)
                        st_row4(kx->dF + tb * (long)(PADK ? p.ns * (p.ns + p.nc) : NS * N), PADK ? p.ns : NS, 16 * I + 4 * L.q,
                                f32x4{lcol[I][0], lcol[I][1], lcol[I][2], lcol[I][3]});
                }
            }
        }

        PROF40_MARK(2);
        if (PADK) {
#pragma unroll
            for (int v = 0; v < 4; ++v) Qd[2][2][v] += padd[v];
        }
        // ---- Quu, qu; K = -Quu^-1 Qux, k = -Quu^-1 qu   (:84-94; LDL' for the pinverse)
        // Quu stays spread over lanes in every mode (Ldl8V): column c of it is register c & 3 of lane row c >> 2 of the
        // accumulator tile, copied to all four lane rows with two row swaps per register; pinned / clamped controls become
        // identity rows and columns of the factorised matrix (Pnqp8vMat).  (Round 2 read Quu out with 36 readlanes and
        // factorised on wave-uniform values in the constrained modes.)
        Ldl8V facv;
        QpFac qpf;                      // box QP: the factorisation of its last trip
        float kqp_v = 0.f, frq_v = 1.f; // box QP: its solution k and its free set (1 / 0), lane a of every row: entry a
        float qu[8], kk[8];
        bool fr[8];
        // q_u (row layout): the one vector needed before the gains; q_x stays in shares until v takes it
        const float qrow2 = crow[2] + sum_q(qpart[2]);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            qu[a] = wv::readlane(qrow2, a);
            fr[a] = true;
            kk[a] = 0.f;
        }
        if (MODE == 0) {
            float col[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) wv::rows01(Qd[2][2][v], col[v], col[4 + v]);
            ldl8v<true>(facv, col);
            if (facv.sing != 0.f) status |= MPC_ST_QUU_SINGULAR;
        }
        float mkv = 0.f;                // constrained modes: m = qu + Quu k spread over lanes (lane a < 8 of every row: m[a])
        float col1[8];                  // u_zero_I mode: Quu's columns spread over lanes (m and the record need them again)
        if (MODE == 0) {
            // (k comes out of the first of the two K solves below: lane rows 2, 3 carry qu as their right-hand side)
        } else if (MODE == 1) {                          // :99-127: pinned controls drop out of the solve
            unsigned zlo, zhi;
            if (PADK && KKT) {
                kkt_pinned_lds(p, base + OFF_R + 4u * 72u, base + OFF_R + 4u * 112u, base + OFF_R + 4u * 120u, L.lane, zlo, zhi);
            } else {
                zlo = KKT ? kkt_pinned_word(p, tb, 0) : zero_mask_word(p, tb, 0);
                zhi = KKT ? kkt_pinned_word(p, tb, 1) : zero_mask_word(p, tb, 1);
            }
            float frf[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                fr[a] = (((a < 4 ? zlo : zhi) >> (8 * (a & 3))) & 0xffu) == 0u;
                frf[a] = fr[a] ? 1.f : 0.f;
            }
            // the free block of Quu, identity elsewhere, factorised spread over lanes (k rides in the K solve below)
#pragma unroll
            for (int v = 0; v < 4; ++v) wv::rows01(Qd[2][2][v], col1[v], col1[4 + v]);
            float diagv = 0.f;
#pragma unroll
            for (int a = 0; a < 8; ++a) diagv = pick(L.r == a, col1[a], diagv);
            const float mnv = gather8(frf, L.r);
            float colm[8];
            Pnqp8vMat<0>::run(col1, mnv, fmaf(mnv, diagv - 1.f, 1.f), L.r, colm);
            ldl8v(facv, colm);
        } else {                                         // :128-148: box QP, warm start k_{t+1}
            // the QP's data spread over lanes (pnqp8v): H's columns by two row swaps per accumulator register
            float col0[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) wv::rows01(Qd[2][2][v], col0[v], col0[4 + v]);
            float diagv = 0.f;
#pragma unroll
            for (int a = 0; a < 8; ++a) diagv = pick(L.r == a, col0[a], diagv);
            const bool r8 = L.r < 8;
            const float uav = wv::lds_f32(base + OFF_R + 288 + 4u * (unsigned)(r8 ? L.r : 0));
            float lov = p.lo_s, hiv = p.hi_s;
            if (PADK) {
                // (round 6) this lane's own bounds: scalars, or the record's words 80.. / 88.. (rec_init); a control beyond n_ctrl is unbounded
                const unsigned rr = 4u * (unsigned)(r8 ? L.r : 0);
                if (p.bound_mode != MPC_BOUND_SCALAR) {
                    lov = wv::lds_f32(base + OFF_R + 320 + rr);
                    hiv = wv::lds_f32(base + OFF_R + 352 + rr);
                }
                lov = L.r < p.nc ? lov : -3e38f;
                hiv = L.r < p.nc ? hiv : 3e38f;
            } else if (p.bound_mode != MPC_BOUND_SCALAR) {
                float lo8[8], hi8[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    lo8[a] = bound_lo(p, tb, a);
                    hi8[a] = bound_hi(p, tb, a);
                }
                lov = gather8(lo8, L.r);
                hiv = gather8(hi8, L.r);
            }
            float lbv = r8 ? lov - uav : 0.f, ubv = r8 ? hiv - uav : 0.f;
            if (p.has_delta) {                           // :132-134
                lbv = r8 ? fmaxf(lbv, -p.delta_u) : 0.f;
                ubv = r8 ? fminf(ubv, p.delta_u) : 0.f;
            }
            const float qv = r8 ? qrow2 : 0.f;
            float xv = kprev_v;
            if (MPC_QP_START && p.qp_start) {
                // the caller's start (mpc_lqr_options.qp_start; a hint: the solve ends on a confirmed free set whatever it is).
                // (The array may be this kernel's own k of an earlier launch: timestep t's block was fetched before this timestep's
                // store further down.)
                // it rode in with the record (stream_init: lanes 28..29; padded instantiation: rec_init's words 112..), entry r in lane (q, r)
                // like the nominal control
                xv = wv::lds_f32(base + OFF_R + 448 + 4u * (unsigned)(r8 ? L.r : 0));
                xv = (r8 && (!PADK || L.r < p.nc)) ? xv : 0.f;
                xv = (xv == xv) ? xv : 0.f;              // (a NaN would survive the clamp)
            } else if (!MPC_MFMA40_QP_WARM || !warm) {
                // cold start x = -H^-1 q (mpc/pnqp.py:14-19): the first QP of the sweep -- or, with MPC_MFMA40_QP_WARM = 0, every QP
                // whose Quu is positive definite (see the switch's comment: measured level at this shape, not the product)
                float colc[8], y[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) colc[a] = col0[a];
                Ldl8V f0;
                ldl8v(f0, colc);
                Pnqp8vSpread<0>::run(qv, y);
                ldl8v_solve(f0, y);
                // (a Quu that is not positive definite -- a pivot <= 0 or not finite -- keeps the reference's start: where a
                // non-convex QP ends depends on where it starts)
                const float imin = fminf(fminf(fminf(f0.inv[0], f0.inv[1]), fminf(f0.inv[2], f0.inv[3])), fminf(fminf(f0.inv[4], f0.inv[5]), fminf(f0.inv[6], f0.inv[7])));
                const float imax = fmaxf(fmaxf(fmaxf(f0.inv[0], f0.inv[1]), fmaxf(f0.inv[2], f0.inv[3])), fmaxf(fmaxf(f0.inv[4], f0.inv[5]), fmaxf(f0.inv[6], f0.inv[7])));
                const bool cold = !warm || (imin > 0.f && imax < 3.0e38f);
                xv = cold ? -gather8(y, L.r) : kprev_v;
            }
            xv = clampf(xv, lbv, ubv);                   // :23
            bool conv;
            float mv = 1.f;
            const int it = pnqp8v(col0, diagv, qv, lbv, ubv, p.pnqp_iter, L.r, xv, mv, qpf, conv);
            qp_total += 1 + it;                          // :140
            if (!conv) status |= MPC_ST_PNQP_UNCONVERGED;
            warm = true;
            kprev_v = xv;
            // m = qu + Quu k: the QP's gradient at its solution, eight DPP multiply-adds on the lane-spread data
            mkv = qv;
            Pnqp8vMv<0>::run(col0, xv, mkv);
            // (k and the free set stay where the QP left them, spread over lanes: what follows reads entry a as a DPP broadcast
            // of lane a.  Rounds 2-3 made them wave-uniform first: sixteen v_readlane and their wait states per timestep.)
            kqp_v = xv;
            frq_v = mv;
        }
        PROF40_MARK(3);
        f32x4 Kd[2];                    // K, B layout of the value update: register v of lane (q,r) = K[4q+v][16J+r]
        float Kp[2][2];                 // ... its registers 2h, 2h + 1 packed into one full-k MFMA operand (see below)
        f32x4 Md[2];                    // M = Qux + Quu K in the same layout (constrained modes)
        Md[0] = zero4;
        Md[1] = zero4;
        {
            // ONE solve for all 32 columns of Qux (and qu): lane row 0 takes column r of tile 0, row 1 column r of tile 1 --
            // a column's eight entries are registers v of rows 0 and 1 of its tile, and one row swap per register
            // (wv::swap16: lo = {a.row0, b.row0, ..}, hi = {a.row1, b.row1, ..}) hands rows 0 / 1 entries 0..3 and 4..7 of
            // THEIR column; row 2 solves for k (its rows of Qux are padding), row 3 rides along.  The same swap turns the
            // solutions back into the two B-layout tiles.  (Round 2 solved tile by tile with every column in two lanes:
            // two passes of 64 dependent multiply-adds instead of one.)
            float rhs[8], sol[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) wv::swap16(Qd[2][0][v], Qd[2][1][v], rhs[v], rhs[4 + v]);
            if (MODE != 0) {
                // (u_zero_I mode: lane row 2 carries qu, as in the unconstrained mode; the box QP has its k already)
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const float x = (MODE == 1 && L.q == 2) ? qu[a] : rhs[a];
                    sol[a] = fr[a] ? x : 0.f;                                                         // :142-143
                }
                if (MODE == 2) free_rows<0>(frq_v, rhs, sol);
#ifdef MPC_MFMA40_QP_LDL
                ldl8v_solve(MODE == 2 ? qpf : facv, sol);
#else
                if (MODE == 2) gj8v_solve(qpf, sol); else ldl8v_solve(facv, sol);
#endif
                if (MODE == 2) {
                    free_rows<0>(frq_v, sol, sol);
                } else {
#pragma unroll
                    for (int a = 0; a < 8; ++a) sol[a] = fr[a] ? sol[a] : 0.f;
                }
                if (MODE == 1) {
                    float w = 0.f;
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        kk[a] = -wv::readlane(sol[a], 32);
                        w = fmaf(qu[a], kk[a], w);
                    }
                    if (KKT) w0 += 0.5 * (double)w;      // the nested line search's predicted change (pinned controls: k = 0)
                    // m = qu + Quu k on the lane-spread columns
                    const float kv = gather8(kk, L.r);
                    mkv = L.r < NC ? qrow2 : 0.f;
                    Pnqp8vMv<0>::run(col1, kv, mkv);
                    kprev_v = kv;                        // (k spread over lanes, for w0 of the priced rollout below)
                }
            } else {
#pragma unroll
                for (int a = 0; a < 8; ++a) sol[a] = L.q == 2 ? qu[a] : rhs[a];
                ldl8v_solve(facv, sol);
                float w = 0.f;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    kk[a] = -wv::readlane(sol[a], 32);
                    w = fmaf(qu[a], kk[a], w);
                }
                w0 += 0.5 * (double)w;
            }
            float Kc[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) Kc[a] = -sol[a];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float k0, k1;
                wv::swap16(Kc[v], Kc[4 + v], k0, k1);
                Kd[0][v] = L.q < 2 ? k0 : 0.f;
                Kd[1][v] = L.q < 2 ? k1 : 0.f;
            }
#ifndef MPC_MFMA40_KPAD
            // (round 4) A contraction over the eight controls is register v of the operand tiles for v = 0..3, and in each only
            // lane groups q = 0, 1 (controls v and 4 + v) are not padding: half of every MFMA's k.  Moving the lower halves of
            // registers v, v + 1 side by side (one v_permlane32_swap each) makes it TWO full MFMAs per output tile instead of four:
            // M = Qux + Quu K (constrained modes) and V = Qxx + Qxu K, all their operands packed first, then the MFMAs in one
            // run (vector instructions between MFMAs cost more than they hide: profiles/r04_ab_mfma_shadow.log).
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int h = 0; h < 2; ++h) Kp[J][h] = wv::lower_halves(Kd[J][2 * h], Kd[J][2 * h + 1]);
            float Qp[2] = {0.f, 0.f}, Ap[2][2];
            if (MODE != 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) Qp[h] = wv::lower_halves(Qd[2][2][2 * h], Qd[2][2][2 * h + 1]);
            }
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int h = 0; h < 2; ++h) Ap[I][h] = wv::lower_halves(Qd[2][I][2 * h], Qd[2][I][2 * h + 1]);
            wv::sched_fence();
            if (MODE != 0) {
                // M = Qux + Quu K: the Quu tile of Q is, by symmetry, its own A operand, the Qux tiles the accumulators
#pragma unroll
                for (int J = 0; J < 2; ++J) Md[J] = Qd[2][J];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int J = 0; J < 2; ++J) Md[J] = wv::mfma(Qp[h], Kp[J][h], Md[J]);
            }
            // ---- V = Qxx + Qxu K   (:155-158 with K'(Qux + Quu K) = 0; v follows below)
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J) Vd[I][J] = Qd[I][J];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int J = 0; J < 2; ++J) Vd[I][J] = wv::mfma(Ap[I][h], Kp[J][h], Vd[I][J]);
            wv::sched_fence();
#else
            if (MODE != 0) {
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    Md[J] = Qd[2][J];
#pragma unroll
                    for (int v = 0; v < 4; ++v) Md[J] = wv::mfma(Qd[2][2][v], Kd[J][v], Md[J]);
                }
            }
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J) Vd[I][J] = Qd[I][J];
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int J = 0; J < 2; ++J) Vd[I][J] = wv::mfma(Qd[2][I][v], Kd[J][v], Vd[I][J]);
#endif
            if (L.q < 2) {
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) Kout[(tb * NC + 4 * L.q + v) * NS + 16 * J + L.r] = Kd[J][v];
#ifdef MPC_MFMA40_PAD
                // (Kout / kout above are the kernel's own padded gains in the workspace; the caller's K [T,B,nc,ns], if asked for:)
                if (p.K_user) {
                    // (rows of K_t [nc,ns] through a raw buffer over the block: entries beyond the true shape fall outside it)
#pragma unroll
                    for (int J = 0; J < 2; ++J)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const bool in = 4 * L.q + v < p.nc && 16 * J + L.r < p.ns;
                            wv::st_buf(p.K_user + tb * (long)(p.nc * p.ns), (unsigned)(4 * p.nc * p.ns),
                                       in ? (unsigned)(4 * ((4 * L.q + v) * p.ns + 16 * J + L.r)) : OOB_OFF, Kd[J][v]);
                        }
                }
#endif
            }
        }
        if (L.lane < 8) {
            float kv = kk[0];
#pragma unroll
            for (int a = 1; a < 8; ++a) kv = pick(L.lane == a, kk[a], kv);
            if (MODE == 2) kv = kqp_v;
            kout[tb * NC + L.lane] = kv;
#ifdef MPC_MFMA40_PAD
            if (p.k_user) wv::st_buf(p.k_user + tb * p.nc, (unsigned)(4 * p.nc), (unsigned)(4 * L.lane), kv);
#endif
        }

        PROF40_MARK(4);
        // (+ K'(Qux + Quu K) of :155-158 is zero but for rounding in the constrained modes too: K's rows are exactly zero where a
        // control is pinned or clamped, and on the free rows Qux + Quu K = 0 is what the direct solve for K just enforced --
        // sixteen MFMAs that added 1e-7-relative noise.  The vector term K'(qu + Quu k) stays: the box QP stops at |dx| < 1e-4,
        // its gradient on the free set is small, not rounding.)
        float mq[4] = {0.f, 0.f, 0.f, 0.f};         // m[4q + v] for this lane group (constrained modes)
        if (MODE != 0) {
            float lo4[4], hi4[4];
            lo4[0] = wv::bcast<0>(mkv); lo4[1] = wv::bcast<1>(mkv); lo4[2] = wv::bcast<2>(mkv); lo4[3] = wv::bcast<3>(mkv);
            hi4[0] = wv::bcast<4>(mkv); hi4[1] = wv::bcast<5>(mkv); hi4[2] = wv::bcast<6>(mkv); hi4[3] = wv::bcast<7>(mkv);
#pragma unroll
            for (int v = 0; v < 4; ++v) mq[v] = pick(L.q == 0, lo4[v], pick(L.q == 1, hi4[v], 0.f));
        }
        if (MODE != 0 && !KKT && p.Kk) {
            // what the priced rollout of the constrained modes needs (rollout_priced): M, Quu, m of this timestep, and the
            // value function's constant w0 += k'(m + qu) / 2
            float *rec = p.Kk + tb * (long)PREC;
            if (L.q < 2) {
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) rec[(4 * L.q + v) * NS + 16 * J + L.r] = Md[J][v];
                if (L.r < NC) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) rec[256 + (4 * L.q + v) * NC + L.r] = Qd[2][2][v];
                }
            }
            if (L.lane < NC) rec[320 + L.lane] = mkv;
            const float kv = kprev_v;                                     // (k spread over lanes)
            w0 += 0.5 * (double)wv::row_sum(kv * (mkv + (L.r < NC ? qrow2 : 0.f)));
        }
        float vrow[2], lrow[2] = {0.f, 0.f}, grow[2] = {0.f, 0.f};
        float kq4[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (MODE == 2) Pnqp8vSpread<0>::run(kqp_v, kq4);
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float ka = MODE == 2 ? pick(L.q == 0, kq4[v], pick(L.q == 1, kq4[4 + v], 0.f))
                                           : pick(L.q == 0, kk[v], pick(L.q == 1, kk[4 + v], 0.f));
                s = fmaf(Qd[2][J][v], ka, s);
                if (MODE != 0) s = fmaf(Kd[J][v], mq[v], s);      // + K'(qu + Quu k)
            }
            // v = q_x + Qxu k (+ K'(qu + Quu k)): c_back's, F'v's and this step's shares reduced together
            vrow[J] = crow[J] + sum_q(qpart[J] + s);
            if (KKT) {
                lrow[J] = sum_q(lpart[J]) + wv::lds_f32(base + OFF_R + 320 + 4u * (unsigned)(16 * J + L.r));
                grow[J] = sum_q(gpart[J] - s);              // g_t = F_x' g_{t+1} - Qxu k_t: what v just took on
            }
        }
        // row -> column layout through the scratch words
        wv::lds_sync();
        if (L.q == 0) {
            wv::lds_store_f32(OFF_SCR + 4u * (unsigned)L.r, vrow[0]);
            wv::lds_store_f32(OFF_SCR + 64 + 4u * (unsigned)L.r, vrow[1]);
            if (KKT) {
                wv::lds_store_f32(OFF_SCR + 128 + 4u * (unsigned)L.r, lrow[0]);
                wv::lds_store_f32(OFF_SCR + 192 + 4u * (unsigned)L.r, lrow[1]);
                wv::lds_store_f32(OFF_SCR + 256 + 4u * (unsigned)L.r, grow[0]);
                wv::lds_store_f32(OFF_SCR + 320 + 4u * (unsigned)L.r, grow[1]);
            }
        }
        wv::lds_sync();
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const f32x4 w = wv::lds_f32x4(OFF_SCR + 64u * (unsigned)I + 16u * (unsigned)L.q);
#pragma unroll
            for (int v = 0; v < 4; ++v) vcol[I][v] = w[v];
            if (KKT) {
                const f32x4 wl = wv::lds_f32x4(OFF_SCR + 128 + 64u * (unsigned)I + 16u * (unsigned)L.q);
                const f32x4 wg = wv::lds_f32x4(OFF_SCR + 256 + 64u * (unsigned)I + 16u * (unsigned)L.q);
#pragma unroll
                for (int v = 0; v < 4; ++v) { lcol[I][v] = wl[v]; gcol[I][v] = wg[v]; }
            }
        }
        if (KKT && t == 0) {
            kkt_store_vvg(*kx, tb, L, Vd, vcol, gcol);
            if (v0g0) { v0g0[0] = vrow[0]; v0g0[1] = vrow[1]; v0g0[2] = grow[0]; v0g0[3] = grow[1]; }
        }
        if (verify) {
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) xnext[I][v] = tcol[I][v];
        }
        slot = (slot + 1) % SNSTAGE;
        PROF40_MARK(5);
    }
    if (on_dynamics_out) {
        bool on = p.on_dynamics != 0;
        if (verify) {
            // the rollout starts from x_init: the nominal (xnext holds x_0 now) has to as well
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float x0 = ld_xinit(p, L.b, 16 * I + 4 * L.q + v);
                    float r = fabsf(x0 - xnext[I][v]) - 1e-5f * (1.f + fabsf(x0));
                    r = (r == r) ? r : 1.f;
                    offdyn = r > offdyn ? r : offdyn;
                }
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) offdyn = fmaxf(offdyn, wv::shfl_xor(offdyn, sh));
            on = !(offdyn > 0.f);
            if (!on) status |= MPC_ST_NOMINAL_OFF_DYNAMICS;
        }
        *on_dynamics_out = on;
    }
    // the nominal cost: every lane holds the partial sum of its entries and its lane group's rows (one butterfly per sweep)
#pragma unroll
    for (int sh = 1; sh < (KKT ? 1 : 64); sh <<= 1) {
        // (both partners add the same two rounded numbers: every lane ends with the same bits)
        const float hi = (float)old_cost, lo = (float)(old_cost - (double)hi);
        old_cost = ((double)hi + (double)lo) + ((double)wv::shfl_xor(hi, sh) + (double)wv::shfl_xor(lo, sh));
    }
    if (L.lane == 0 && p.old_costs) p.old_costs[L.b] = (float)old_cost;
    if (L.lane == 0 && p.qp_iters) p.qp_iters[L.b] = qp_total;
    if (!p.c_symmetric) {
        // (a tolerance, not a bit test: C = A'A out of a float32 GEMM is symmetric to rounding only)
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            cmax = fmaxf(cmax, wv::shfl_xor(cmax, sh));
            asym = fmaxf(asym, wv::shfl_xor(asym, sh));
        }
        status |= MPC_ST_C_TESTED;
        if (asym > 1e-5f * cmax) status |= MPC_ST_C_ASYMMETRIC;
    }
    if (L.lane == 0 && p.status) p.status[L.b] = status;
    if (w0_out) *w0_out = w0;
    PROF40_MARK(6);
    return old_cost;
}

// ---------------------------------------------------------------------------------------------
// Rollout + line search (mpc/lqr_step.py:164-261, LinDx / QuadCost, no bounds).
// The state is a 32 x 16 matrix in D layout: column r is the trajectory of the trial alpha = decay^r, so
// ONE pass prices every candidate of the line search; u' = K dx, x+ = F tau' + f and C tau' are MFMAs whose
// A operands come straight from LDS and whose outputs are the next operands (same block order trick as
// the sweep: x block (I',v) <-> 16I'+4q+v, u block v <-> 32+4q+v).  Trial 0 (the usual winner) writes its
// trajectory as it goes; any other winner is replayed once with its alpha in every column.
// ---------------------------------------------------------------------------------------------
struct RStream {
    const char *c_ptr, *f_ptr, *k_ptr;      // wave-uniform block addresses (see Stream)
    const char *r_ptr;
    long c_step, f_step, k_step;
    unsigned r_step, lo;
    bool r_active, r_is_f, r_is_x;
    long bidx;                              // the problem (padded fused backward: kstage_issue's u* and bounds)
    // padded instantiation: gathers instead (see Stream)
    const float *Cb, *Fb;
    Gather g;
    RecMap rm;
};

MPC_DEV void rstream_init(RStream &d, const P &p, const Lane &L, const float *Kin, const float *kin)
{
    const long b = L.b;
    if (PADK) {
        d.Cb = p.C + b * p.C_sb;
        d.Fb = p.T > 1 ? p.F + b * p.F_sb : d.Cb;
        gather_init(d.g, p, L.lane);
        rec_init(d.rm, p, L.lane, b, true, true, true, kin);
    }
    d.lo = 16u * (unsigned)L.lane;
    d.c_ptr = (const char *)(p.C + b * p.C_sb);
    d.c_step = p.C_st * 4;
    d.f_ptr = p.T > 1 ? (const char *)(p.F + b * p.F_sb) : d.c_ptr;
    d.f_step = p.T > 1 ? p.F_st * 4 : 0;
    d.k_ptr = (const char *)(Kin + b * (NC * NS));
    d.k_step = (long)p.B * NC * NS * 4;
    // record: lanes 0..9 c_t | 10..17 x_{t+1} | 18..19 u_t | 20..27 f_t | 28..29 k_t
    d.r_active = L.lane < 30;
    d.r_is_f = false;
    d.r_is_x = false;
    if (L.lane < 10) {
        d.r_ptr = (const char *)(p.c + b * p.c_sb) + 16 * L.lane;
        d.r_step = (unsigned)(p.c_st * 4);
    } else if (L.lane < 18) {
        d.r_ptr = (const char *)(p.cur_x + b * NS) + 16 * (L.lane - 10);
        d.r_step = (unsigned)((long)p.B * NS * 4);
        d.r_is_x = true;
    } else if (L.lane < 20) {
        d.r_ptr = (const char *)(p.cur_u + b * NC) + 16 * (L.lane - 18);
        d.r_step = (unsigned)((long)p.B * NC * 4);
    } else if (L.lane < 28) {
        d.r_is_f = true;
        d.r_active = p.f != nullptr && p.T > 1;
        d.r_ptr = p.f ? (const char *)(p.f + b * p.f_sb) + 16 * (L.lane - 20) : d.c_ptr + d.lo;
        d.r_step = (unsigned)(p.f_st * 4);
    } else {
        d.r_ptr = (const char *)(kin + b * NC) + 16 * ((L.lane < 30 ? L.lane : 28) - 28);
        d.r_step = (unsigned)((long)p.B * NC * 4);
    }
}

MPC_DEV void rstage_issue(const P &p, const RStream &d, const Lane &L, int t, int slot)
{
    const unsigned base = (unsigned)slot * RSTAGE_BYTES;
    const long tl = t;
    const long tf = t < p.T - 1 ? t : (p.T > 1 ? p.T - 2 : 0);      // F, f have T-1 entries
    const long tx = t + 1 < p.T ? t + 1 : t;                         // x_{t+1}
    if (PADK) {
        gather_C(d.g, d.Cb + tl * p.C_st, base + OFF_C, L.lane);
        gather_F(d.g, p.T > 1 ? d.Fb + tf * p.F_st : d.Cb, p.T > 1 ? d.g.fbytes : 0u, base + OFF_F);
        wv::dma16(d.k_ptr + tl * d.k_step + d.lo, base + ROFF_K);
        rec_issue(d.rm, tl, tf, tx, base + ROFF_R);
        return;
    }
    dma_kib<6>(d.c_ptr + tl * d.c_step + d.lo, base + OFF_C);
    wv::dma16_at_if<2048>(L.lane < 16, d.c_ptr + tl * d.c_step + 4096 + d.lo, base + OFF_C + 4096);
    dma_kib<5>(d.f_ptr + tf * d.f_step + d.lo, base + OFF_F);
    wv::dma16(d.k_ptr + tl * d.k_step + d.lo, base + ROFF_K);
    wv::dma16_if(d.r_active, d.r_ptr + rec_off(d.r_is_f ? tf : (d.r_is_x ? tx : tl), d.r_step), base + ROFF_R);
}

// One pass over the horizon.  alpha: this lane's (= its column's) step size.  Returns the column's cost and
// squared control change in every lane of the column.
template <int MODE>
MPC_DEV void rollout_pass(const P &p, const RStream &d, const Lane &L, float alpha, bool store, double &cost, float &du2)
{
    const int T = p.T;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 Xd[2], DXd[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Xd[I][v] = ld_xinit(p, L.b, 16 * I + 4 * L.q + v);
        DXd[I] = zero4;
        if (store && L.r == 0) st_x4(p, L.b, 16 * I + 4 * L.q, Xd[I]);
    }
    double cacc = 0.0;
    float dacc = 0.f;
    wv::dma_wait<0>();          // nothing of the sweep / the previous pass may still land in the ring
    pad_clear(L.lane);
    rstage_issue(p, d, L, 0, 0);
    int slot = 0;
    for (int t = 0; t < T; ++t) {
        rstage_issue(p, d, L, t + 1 < T ? t + 1 : T - 1, slot ^ 1);
        wv::dma_wait<RDMA_PER_STAGE>();
        const unsigned base = (unsigned)slot * RSTAGE_BYTES;
        const long tb = (long)t * p.B + L.b;
        const unsigned rec = base + ROFF_R;
        // ---- u' = K dx + u + alpha k   (:192)
        // (operands first, then the matrix-core block undivided: an MFMA <-> VALU turn costs ~15 clocks here,
        // tools/ubench/mfma16_turn.hip)
        f32x4 Ud = zero4;
        {
            float a[8];
#pragma unroll
            for (int J = 0; J < 2; ++J) {        // four consecutive words of K's row r per lane and state tile: one 16-byte LDS read
                const f32x4 x = wv::lds_f32x4(base + ROFF_K + 4u * (unsigned)((L.r < NC ? L.r : 0) * NS + 16 * J + 4 * L.q));
#pragma unroll
                for (int i = 0; i < 4; ++i) a[4 * J + i] = L.r < NC ? x[i] : 0.f;
            }
            wv::sched_fence();
#pragma unroll
            for (int k = 0; k < 8; ++k) Ud = wv::mfma(a[k], DXd[k >> 2][k & 3], Ud);
            wv::sched_fence();
        }
        {
            const unsigned qo = 16u * (unsigned)(L.q < 2 ? L.q : 0);
            const f32x4 ub = wv::lds_f32x4(rec + 288 + qo), kb = wv::lds_f32x4(rec + 448 + qo);
            float s = 0.f;
            unsigned zw = 0u;                    // the zero flags of controls 4q .. 4q+3
            if (MODE == 1) {
                const unsigned zlo = zero_mask_word(p, tb, 0), zhi = zero_mask_word(p, tb, 1);
                zw = L.q == 0 ? zlo : zhi;
            }
            Box4 bx;
            if (MODE == 2) box4(bx, p, tb, L, ub);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float un = L.q < 2 ? Ud[v] + ub[v] + alpha * kb[v] : 0.f;
                if (MODE == 1 && L.q < 2 && ((zw >> (8 * v)) & 0xffu) != 0u) un = 0.f;               // :197-198
                if (MODE == 2) un = L.q < 2 ? box_clamp(un, bx.lo[v], bx.hi[v]) : 0.f;               // :200-213
                const float dd = L.q < 2 ? ub[v] - un : 0.f;
                Ud[v] = un;
                s = fmaf(dd, dd, s);
            }
            dacc += s;
            if (store && L.r == 0 && L.q < 2) st_u4(p, tb, 4 * L.q, Ud);
        }
        // ---- stage cost 0.5 tau'C tau + c'tau   (:230-232); C tau' on MFMA (C read through its symmetry)
        {
            float part = 0.f;
#pragma unroll
            for (int I = 0; I < 3; ++I) {
                f32x4 G = zero4;
                const int col = 16 * I + L.r;
                const bool cin = col < N;
                {
                    float a[12];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float x = wv::lds_f32(base + OFF_C + 4u * (unsigned)((16 * (k >> 2) + 4 * L.q + (k & 3)) * N + (cin ? col : 0)));
                        a[k] = cin ? x : 0.f;
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const bool in = cin && L.q < 2;
                        const float x = wv::lds_f32(base + OFF_C + 4u * (unsigned)((in ? 32 + 4 * L.q + v : 0) * N + (cin ? col : 0)));
                        a[8 + v] = in ? x : 0.f;
                    }
                    wv::sched_fence();
#pragma unroll
                    for (int k = 0; k < 8; ++k) G = wv::mfma(a[k], Xd[k >> 2][k & 3], G);
#pragma unroll
                    for (int v = 0; v < 4; ++v) G = wv::mfma(a[8 + v], Ud[v], G);
                    wv::sched_fence();
                }
                // rows 16I + 4q + v of (C tau') against the same entries of tau' and c
                const bool rin = I < 2 || L.q < 2;
                const f32x4 cv = wv::lds_f32x4(rec + 4u * (unsigned)(rin ? 16 * I + 4 * L.q : 0));
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float tv = I < 2 ? Xd[I][v] : Ud[v];
                    part = fmaf(rin ? tv : 0.f, fmaf(0.5f, G[v], cv[v]), part);
                }
            }
            cacc += (double)part;
        }
        // ---- x+ = F tau' + f   (:216-222)
        if (t < T - 1) {
            const long tb1 = (long)(t + 1) * p.B + L.b;
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                f32x4 acc = zero4;
                if (p.f) acc = wv::lds_f32x4(rec + 320 + 4u * (unsigned)(16 * Im + 4 * L.q));
                const int row = 16 * Im + L.r;
                {
                    float a[12];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        a[k] = wv::lds_f32(base + OFF_F + 4u * (unsigned)(row * N + 16 * (k >> 2) + 4 * L.q + (k & 3)));
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float x = wv::lds_f32(base + OFF_F + 4u * (unsigned)(row * N + (L.q < 2 ? 32 + 4 * L.q + v : 0)));
                        a[8 + v] = L.q < 2 ? x : 0.f;
                    }
                    wv::sched_fence();
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = wv::mfma(a[k], Xd[k >> 2][k & 3], acc);
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc = wv::mfma(a[8 + v], Ud[v], acc);
                    wv::sched_fence();
                }
                if (store && L.r == 0) st_x4(p, tb1, 16 * Im + 4 * L.q, acc);
                DXd[Im] = acc;      // parked here until both tiles are done (the second product still reads Xd)
            }
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                const f32x4 xb = wv::lds_f32x4(rec + 160 + 4u * (unsigned)(16 * Im + 4 * L.q));
                Xd[Im] = DXd[Im];
#pragma unroll
                for (int v = 0; v < 4; ++v) DXd[Im][v] = Xd[Im][v] - xb[v];
            }
        }
        slot ^= 1;
    }
    // column totals: the four lane groups hold disjoint rows
    double c2 = cacc;
    {
        const float hi = (float)c2, lo = (float)(c2 - (double)hi);
        const double o1 = (double)wv::shfl_xor(hi, 16) + (double)wv::shfl_xor(lo, 16);
        c2 += o1;
        const float hi2 = (float)c2, lo2 = (float)(c2 - (double)hi2);
        c2 += (double)wv::shfl_xor(hi2, 32) + (double)wv::shfl_xor(lo2, 32);
    }
    cost = c2;
    du2 = sum_q(dacc);
}

template <int MODE, bool REPLAY>
MPC_DEV void rollout_lean(const P &p, const Lane &L, const float *Kin, const float *kin, double old_cost, double w0,
                          float replay_alpha = 1.f);

template <int MODE> MPC_DEV void rollout_wave(const P &p, const float *Kin, const float *kin, double old_cost)
{
    Lane L;
    L.lane = wv::lane();
    L.r = L.lane & 15;
    L.q = L.lane >> 4;
    L.b = wv::problem();
    if (L.b >= p.B) return;
    RStream d;
    rstream_init(d, p, L, Kin, kin);
    float alpha = 1.f;
    for (int i = 0; i < L.r; ++i) alpha *= p.ls_decay;            // column r tries decay^r
    double cost;
    float du2;
    rollout_pass<MODE>(p, d, L, alpha, true, cost, du2);
    // first trial that is not worse than the nominal, else the last one (:176-179, 247)
    int win = p.max_ls - 1;
    for (int j = p.max_ls - 1; j >= 0; --j) {
        const float cj_hi = wv::readlane((float)cost, j);
        const double cj = (double)cj_hi + (double)wv::readlane((float)(cost - (double)(float)cost), j);
        if (!(cj > old_cost)) win = j;
    }
    const float full2 = wv::readlane(du2, 0);
    float win_alpha = 1.f;
    for (int i = 0; i < win; ++i) win_alpha *= p.ls_decay;
    double win_cost;
    float win_du2;
    // the winner's cost and step length are those of its column in the pricing pass; a winner other than the first trial
    // (which stored as it went) is rolled out once more for its trajectory alone: F, K and the record, no C, no pricing
    win_cost = cost;
    win_du2 = du2;
    if (win != 0) rollout_lean<MODE, true>(p, L, Kin, kin, old_cost, 0.0, win_alpha);
    float wc_hi = wv::readlane((float)win_cost, 0), wc_lo = wv::readlane((float)(win_cost - (double)(float)win_cost), 0);
    float wd = wv::readlane(win_du2, 0);
    for (int j = 1; j < 16; ++j) {                   // (uniform loop; readlane wants a constant lane only in the kernel build)
        const float hj = wv::readlane((float)win_cost, j), lj = wv::readlane((float)(win_cost - (double)(float)win_cost), j);
        const float dj = wv::readlane(win_du2, j);
        if (j == win) {
            wc_hi = hj;
            wc_lo = lj;
            wd = dj;
        }
    }
    const double wc = (double)wc_hi + (double)wc_lo;
    if (L.lane == 0) {
        int status = 0;
        if (!(wc == wc) || fabs(wc) > 3e38) status |= MPC_ST_NONFINITE;
        if (p.costs) p.costs[L.b] = (float)wc;
        if (p.full_du_norm) p.full_du_norm[L.b] = sqrtf(full2);
        if (p.alpha_du_norm) p.alpha_du_norm[L.b] = sqrtf(wd);
        if (p.alphas) p.alphas[L.b] = win_alpha;
        if (p.status) p.status[L.b] |= status;
    }
}

// ---------------------------------------------------------------------------------------------
// Unconstrained step on a nominal that is KNOWN to obey the dynamics (MPC_OPT_NOMINAL_ON_DYNAMICS: what
// MPC.forward hands to every step).  With linear dynamics and an exact quadratic model the cost along the search
// direction is exactly J(alpha) = J(nominal) + (2 alpha - alpha^2) w0, w0 = sum_t 0.5 qu'k from the sweep -- the
// same identity the 4-problems-per-wave kernel prices with (lqr_dpp16_body.h).  So the line search
// (mpc/lqr_step.py:176-179, 247) is decided BEFORE the rollout, and the one pass that follows needs neither C
// (6.4 of the 13 KB a rollout stage streams) nor the 36 MFMAs per step of C tau': stage = F | K_t | record,
// four slots in the same LDS, the DMA three steps ahead.  Column r still rolls out alpha = decay^r (column 0 gives
// full_du_norm, :243-245); the winner's column stores.
// ---------------------------------------------------------------------------------------------
constexpr unsigned LOFF_F = 0, LOFF_K = 5120, LOFF_R = 6144, LSTAGE_BYTES = 6656;
constexpr int LSLOTS = 4, LDMA_PER_STAGE = CH_F + 1 + CH_R;    // 7 = 5 (F) + 1 (K) + 1 (record) in the exact kernel
static_assert((LSLOTS - 2) * LDMA_PER_STAGE < 64, "vmcnt is 6 bits");
static_assert(LSLOTS * LSTAGE_BYTES <= LDS_TOTAL, "lean rollout ring exceeds the wave's LDS");

MPC_DEV void lstage_issue(const P &p, const RStream &d, const Lane &L, int t, int slot)
{
    const unsigned base = (unsigned)slot * LSTAGE_BYTES;
    const long tl = t;
    const long tf = t < p.T - 1 ? t : (p.T > 1 ? p.T - 2 : 0);      // F, f have T-1 entries
    const long tx = t + 1 < p.T ? t + 1 : t;                         // x_{t+1}
    if (PADK) {
        gather_F(d.g, p.T > 1 ? d.Fb + tf * p.F_st : d.Cb, p.T > 1 ? d.g.fbytes : 0u, base + LOFF_F);
        wv::dma16(d.k_ptr + tl * d.k_step + d.lo, base + LOFF_K);
        rec_issue(d.rm, tl, tf, tx, base + LOFF_R, true, L.lane);        // (no c in this pass)
        return;
    }
    dma_kib<5>(d.f_ptr + tf * d.f_step + d.lo, base + LOFF_F);
    wv::dma16(d.k_ptr + tl * d.k_step + d.lo, base + LOFF_K);
    // (lanes 0..9 would carry c_t, which this pass never looks at: they sit the instruction out)
    wv::dma16_if(d.r_active && L.lane >= 10, d.r_ptr + rec_off(d.r_is_f ? tf : (d.r_is_x ? tx : tl), d.r_step), base + LOFF_R);
}

// REPLAY (any MODE): the line search has been decided by a pricing pass (rollout_wave) and its winner was not the
// first trial: every column rolls the winner's step `replay_alpha` with the mode's mask / clamp, column 0 stores; nothing
// is priced and no result scalars are written (the pricing pass has them).
template <int MODE, bool REPLAY>
MPC_DEV void rollout_lean(const P &p, const Lane &L, const float *Kin, const float *kin, double old_cost, double w0,
                          float replay_alpha)
{
    const int T = p.T;
    RStream d;
    rstream_init(d, p, L, Kin, kin);
    float alpha = 1.f;
    for (int i = 0; i < L.r; ++i) alpha *= p.ls_decay;            // column r tries decay^r
    // first trial that is not worse than the nominal, else the last one (:176-179, 247)
    int win = p.max_ls - 1;
    if (!REPLAY) {
        float a = 1.f;
        for (int j = 0; j < p.max_ls; ++j) {
            const double cj = old_cost + (2.0 * (double)a - (double)a * (double)a) * w0;
            if (!(cj > old_cost)) { win = j; break; }
            a *= p.ls_decay;
        }
    }
    float win_alpha = 1.f;
    for (int i = 0; i < win; ++i) win_alpha *= p.ls_decay;
    if (REPLAY) {
        alpha = replay_alpha;
        win = 0;
    }
    const bool store = L.r == win;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 Xd[2], DXd[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Xd[I][v] = ld_xinit(p, L.b, 16 * I + 4 * L.q + v);
        DXd[I] = zero4;
        if (store) st_x4(p, L.b, 16 * I + 4 * L.q, Xd[I]);
    }
    float dacc = 0.f;
    wv::dma_wait<0>();          // nothing of the sweep may still land in the ring
    pad_clear(L.lane);
#pragma unroll
    for (int i = 0; i < LSLOTS - 1; ++i) lstage_issue(p, d, L, i < T ? i : T - 1, i);
    for (int t = 0; t < T; ++t) {
        // stages t+1, t+2 are in flight behind the one needed now (re-issuing the last stage at the tail keeps the count)
        wv::dma_wait<(LSLOTS - 2) * LDMA_PER_STAGE>();
        const unsigned base = (unsigned)(t % LSLOTS) * LSTAGE_BYTES;
        const long tb = (long)t * p.B + L.b;
        const unsigned rec = base + LOFF_R;
        // ---- u' = K dx + u + alpha k   (:192)
        f32x4 Ud = zero4;
        float a[8];
#pragma unroll
        for (int J = 0; J < 2; ++J) {        // four consecutive words of K's row r per lane and state tile: one 16-byte LDS read
            const f32x4 x = wv::lds_f32x4(base + LOFF_K + 4u * (unsigned)((L.r < NC ? L.r : 0) * NS + 16 * J + 4 * L.q));
#pragma unroll
            for (int i = 0; i < 4; ++i) a[4 * J + i] = L.r < NC ? x[i] : 0.f;
        }
        const unsigned qo = 16u * (unsigned)(L.q < 2 ? L.q : 0);
        const f32x4 ub = wv::lds_f32x4(rec + 288 + qo), kb = wv::lds_f32x4(rec + 448 + qo);
        {
            const int tn = t + LSLOTS - 1;
            lstage_issue(p, d, L, tn < T ? tn : T - 1, tn % LSLOTS);
        }
        wv::sched_fence();
        {
            // two half-length accumulation chains side by side (dependent MFMAs issue every 40 clocks, independent 32)
            f32x4 U2 = zero4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Ud = wv::mfma(a[k], DXd[0][k], Ud);
                U2 = wv::mfma(a[4 + k], DXd[1][k], U2);
            }
            wv::sched_fence();
#pragma unroll
            for (int v = 0; v < 4; ++v) Ud[v] += U2[v];
        }
        {
            float s = 0.f;
            unsigned zw = 0u;                    // the zero flags of controls 4q .. 4q+3
            if (MODE == 1) {
                const unsigned zlo = zero_mask_word(p, tb, 0), zhi = zero_mask_word(p, tb, 1);
                zw = L.q == 0 ? zlo : zhi;
            }
            Box4 bx;
            if (MODE == 2) box4(bx, p, tb, L, ub);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float un = L.q < 2 ? Ud[v] + ub[v] + alpha * kb[v] : 0.f;
                if (MODE == 1 && L.q < 2 && ((zw >> (8 * v)) & 0xffu) != 0u) un = 0.f;               // :197-198
                if (MODE == 2) un = L.q < 2 ? box_clamp(un, bx.lo[v], bx.hi[v]) : 0.f;               // :200-213
                const float dd = L.q < 2 ? ub[v] - un : 0.f;
                Ud[v] = un;
                s = fmaf(dd, dd, s);
            }
            dacc += s;
            if (store && L.q < 2) st_u4(p, tb, 4 * L.q, Ud);
        }
        // ---- x+ = F tau' + f   (:216-222)
        if (t < T - 1) {
            const long tb1 = (long)(t + 1) * p.B + L.b;
            // both output tiles at once: operands of the two first, then their accumulation chains interleaved
            f32x4 acc[2];
            float fa[2][12];
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                acc[Im] = zero4;
                if (p.f) acc[Im] = wv::lds_f32x4(rec + 320 + 4u * (unsigned)(16 * Im + 4 * L.q));
                const int row = 16 * Im + L.r;
#pragma unroll
                for (int J = 0; J < 2; ++J) {    // (rows of F are 160 B: every such quadruple is 16-byte aligned)
                    const f32x4 x = wv::lds_f32x4(base + LOFF_F + 4u * (unsigned)(row * N + 16 * J + 4 * L.q));
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[Im][4 * J + i] = x[i];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)        // the control columns as two full-k operands (uctl)
                    fa[Im][8 + h] = wv::lds_f32(base + LOFF_F + 4u * (unsigned)(row * N + 32 + uctl(L.q) + 2 * h));
            }
            wv::sched_fence();
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int Im = 0; Im < 2; ++Im) acc[Im] = wv::mfma(fa[Im][k], Xd[k >> 2][k & 3], acc[Im]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float up = wv::lower_halves(Ud[2 * h], Ud[2 * h + 1]);
#pragma unroll
                for (int Im = 0; Im < 2; ++Im) acc[Im] = wv::mfma(fa[Im][8 + h], up, acc[Im]);
            }
            wv::sched_fence();
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                if (store) st_x4(p, tb1, 16 * Im + 4 * L.q, acc[Im]);
                DXd[Im] = acc[Im];
            }
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                const f32x4 xb = wv::lds_f32x4(rec + 160 + 4u * (unsigned)(16 * Im + 4 * L.q));
                Xd[Im] = DXd[Im];
#pragma unroll
                for (int v = 0; v < 4; ++v) DXd[Im][v] = Xd[Im][v] - xb[v];
            }
        }
    }
    wv::dma_wait<0>();
    if (REPLAY) return;
    const float du2 = sum_q(dacc);
    const float full2 = wv::readlane(du2, 0);
    float wd = full2;
    for (int j = 1; j < 16; ++j) {                   // (uniform loop; readlane wants a constant lane only in the kernel build)
        const float dj = wv::readlane(du2, j);
        if (j == win) wd = dj;
    }
    const double wc = old_cost + (2.0 * (double)win_alpha - (double)win_alpha * (double)win_alpha) * w0;
    if (L.lane == 0) {
        int status = 0;
        if (!(wc == wc) || fabs(wc) > 3e38) status |= MPC_ST_NONFINITE;
        if (p.costs) p.costs[L.b] = (float)wc;
        if (p.full_du_norm) p.full_du_norm[L.b] = sqrtf(full2);
        if (p.alpha_du_norm) p.alpha_du_norm[L.b] = sqrtf(wd);
        if (p.alphas) p.alphas[L.b] = win_alpha;
        if (p.status) p.status[L.b] |= status;
    }
}

// ---------------------------------------------------------------------------------------------
// Constrained step (u_zero_I mask / box bounds) on a vouched-for nominal: the line search priced WITHOUT a second pass over C.
// Around a nominal that obeys the dynamics the cost of ANY rollout is, exactly,
//     J(tau') = J(nominal) + w0 + sum_t [ e'(m + M dx) + e'Quu e / 2 ],   e = du - K dx - k,
//     m = qu + Quu k,  M = Qux + Quu K,  w0 = sum_t k'(m + qu) / 2
// -- V, v of the sweep are the cost-to-go of the affine policy (K, k) whatever pinned or clamped it, and a deviation e from the
// policy at timestep t changes the cost by the bracket (its effect on later states is in their dx) -- the identity the
// 4-problems-per-wave kernel prices with (lqr_dpp16_body.h).  Clamps and masks make e != 0, so unlike the unconstrained case
// the trials have to be rolled out to be priced: all sixteen columns in one pass, stage = F | K | record | M | Quu, m (7.8 KiB
// against the 13 KiB of the pass that reads C), 12 more MFMAs per timestep than the lean pass (M dx: 8, Quu e: 4) against 36.
// The sweep leaves (M, Quu, m) in the record p.Kk [T,B,328].  Trial 0 stores as it goes, another winner is replayed (lean).
// ---------------------------------------------------------------------------------------------
constexpr unsigned POFF_F = 0, POFF_K = 5120, POFF_R = 6144, POFF_M = 6656, POFF_Q = 7680, PSTAGE_BYTES = 8000;
constexpr int PSLOTS = LDS_TOTAL / PSTAGE_BYTES >= 4 ? 4 : 3, PDMA_PER_STAGE = CH_F + 1 + CH_R + 2;   // 9 = 5 (F) + K + record + M + (Quu | m)
static_assert((PSLOTS - 2) * PDMA_PER_STAGE < 64, "vmcnt is 6 bits");
static_assert(PSLOTS * PSTAGE_BYTES <= LDS_TOTAL && PSLOTS >= 3, "priced rollout ring exceeds the wave's LDS");

MPC_DEV void pstage_issue(const P &p, const RStream &d, const Lane &L, const char *m_ptr, int t, int slot)
{
    const unsigned base = (unsigned)slot * PSTAGE_BYTES;
    const long tl = t;
    const long tf = t < p.T - 1 ? t : (p.T > 1 ? p.T - 2 : 0);      // F, f have T-1 entries
    const long tx = t + 1 < p.T ? t + 1 : t;                         // x_{t+1}
    if (PADK) {
        gather_F(d.g, p.T > 1 ? d.Fb + tf * p.F_st : d.Cb, p.T > 1 ? d.g.fbytes : 0u, base + POFF_F);
        wv::dma16(d.k_ptr + tl * d.k_step + d.lo, base + POFF_K);
        rec_issue(d.rm, tl, tf, tx, base + POFF_R, true, L.lane);
    } else {
    dma_kib<5>(d.f_ptr + tf * d.f_step + d.lo, base + POFF_F);
    wv::dma16(d.k_ptr + tl * d.k_step + d.lo, base + POFF_K);
    wv::dma16_if(d.r_active && L.lane >= 10, d.r_ptr + rec_off(d.r_is_f ? tf : (d.r_is_x ? tx : tl), d.r_step), base + POFF_R);
    }
    const char *rec = m_ptr + tl * ((long)p.B * PREC * 4) + d.lo;
    wv::dma16(rec, base + POFF_M);
    wv::dma16_if(L.lane < 18, rec + 1024, base + POFF_Q);
}

template <int MODE>
MPC_DEV void rollout_priced(const P &p, const Lane &L, const float *Kin, const float *kin, double old_cost, double w0,
                            Prof40 *prof = nullptr)
{
    const int T = p.T;
    RStream d;
    rstream_init(d, p, L, Kin, kin);
    const char *m_ptr = (const char *)(p.Kk + (long)L.b * PREC);        // (wave-uniform, see Stream)
    float alpha = 1.f;
    for (int i = 0; i < L.r; ++i) alpha *= p.ls_decay;            // column r tries decay^r
    // Trial 0 (column 0) stores into new_x / new_u as it goes.  Trial 1 (alpha = decay) is the winner in most problems that
    // reject trial 0 (box constraints: one problem in six), and with one wavefront per SIMD the slowest wavefront is the
    // kernel's time: its column parks its trajectory behind the records, from where it is COPIED (ten 16-byte loads per lane
    // in flight at once) instead of replayed by another pass over F and the gains; only a later winner is replayed.
    const bool store = L.r < 2;
    float *const scr = p.Kk + (long)T * p.B * PREC;
    float *const xo = L.r == 0 ? p.new_x + (long)L.b * NS : scr + (long)L.b * PSCR;
    float *const uo = L.r == 0 ? p.new_u + (long)L.b * NC : scr + (long)L.b * PSCR + NS;
    const long xst = L.r == 0 ? (long)p.B * NS : (long)p.B * PSCR;
    const long ust = L.r == 0 ? (long)p.B * NC : (long)p.B * PSCR;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 Xd[2], DXd[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
        for (int v = 0; v < 4; ++v) Xd[I][v] = ld_xinit(p, L.b, 16 * I + 4 * L.q + v);
        DXd[I] = zero4;
        if (PADK && L.r == 0) st_x4(p, L.b, 16 * I + 4 * L.q, Xd[I]);       // (the caller's arrays by their true shape; the scratch is padded)
        else if (store) wv::store_f32x4(xo + 16 * I + 4 * L.q, Xd[I]);
    }
    float dacc = 0.f;
    double cacc = 0.0;
    wv::dma_wait<0>();          // nothing of the sweep may still land in the ring
    pad_clear(L.lane);
#pragma unroll
    for (int i = 0; i < PSLOTS - 1; ++i) pstage_issue(p, d, L, m_ptr, i < T ? i : T - 1, i);
    PROF40_MARK(7);
    for (int t = 0; t < T; ++t) {
        wv::dma_wait<(PSLOTS - 2) * PDMA_PER_STAGE>();
        PROF40_MARK(9);             // 9 rollout: DMA wait | 10 operand reads + DMA issue | 11 K dx, M dx | 12 u', e, store | 13 F reads, x+ | 14 tail
        const unsigned base = (unsigned)(t % PSLOTS) * PSTAGE_BYTES;
        const long tb = (long)t * p.B + L.b;
        const unsigned rec = base + POFF_R;
        // ---- operands of the timestep out of its stage: rows of K and of M (A operands), Quu's columns, m, u, k
        // (the price e'(m + M dx + Quu e / 2) cannot be skipped where e = 0: e = (alpha - 1) k in every column but the first)
        float a[8], am[8], aq[2];
        const unsigned krow = 4u * (unsigned)((L.r < NC ? L.r : 0) * NS + 4 * L.q);
#pragma unroll
        for (int J = 0; J < 2; ++J) {        // (16-byte LDS reads: four consecutive words of row r of K, of M, per state tile)
            const f32x4 x = wv::lds_f32x4(base + POFF_K + krow + 64u * (unsigned)J);
            const f32x4 y = wv::lds_f32x4(base + POFF_M + krow + 64u * (unsigned)J);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[4 * J + i] = L.r < NC ? x[i] : 0.f;
                am[4 * J + i] = L.r < NC ? y[i] : 0.f;
            }
        }
        const bool uq = L.q < 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float x = wv::lds_f32(base + POFF_Q + 4u * (unsigned)((L.r < NC ? L.r : 0) * NC + uctl(L.q) + 2 * h));
            aq[h] = L.r < NC ? x : 0.f;                           // A[i = r][k = q] = Quu[r][uctl(q) + 2h]
        }
        const unsigned qo = 16u * (unsigned)(uq ? L.q : 0);
        const f32x4 ub = wv::lds_f32x4(rec + 288 + qo), kb = wv::lds_f32x4(rec + 448 + qo);
        const f32x4 mm = wv::lds_f32x4(base + POFF_Q + 256 + qo);
        {
            const int tn = t + PSLOTS - 1;
            pstage_issue(p, d, L, m_ptr, tn < T ? tn : T - 1, tn % PSLOTS);
        }
        PROF40_MARK(10);
        // ---- K dx and M dx  (four accumulation chains side by side)
        f32x4 Ud = zero4, G = zero4;
        wv::sched_fence();
        {
            f32x4 U2 = zero4, G2 = zero4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Ud = wv::mfma(a[k], DXd[0][k], Ud);
                U2 = wv::mfma(a[4 + k], DXd[1][k], U2);
                G = wv::mfma(am[k], DXd[0][k], G);
                G2 = wv::mfma(am[4 + k], DXd[1][k], G2);
            }
            wv::sched_fence();
#pragma unroll
            for (int v = 0; v < 4; ++v) { Ud[v] += U2[v]; G[v] += G2[v]; }
        }
#ifdef MPC_MFMA40_PROF
        asm volatile("" :: "v"(Ud[0]), "v"(G[0]));
        PROF40_MARK(11);
#endif
        // ---- u' = clamp / mask (u + K dx + alpha k)   (:192-213),  e = du - K dx - k
        f32x4 e = zero4;
        {
            float s = 0.f;
            unsigned zw = 0u;
            if (MODE == 1) {
                const unsigned zlo = zero_mask_word(p, tb, 0), zhi = zero_mask_word(p, tb, 1);
                zw = L.q == 0 ? zlo : zhi;
            }
            Box4 bx;
            if (MODE == 2) box4(bx, p, tb, L, ub);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float un = uq ? Ud[v] + ub[v] + alpha * kb[v] : 0.f;
                if (MODE == 1 && uq && ((zw >> (8 * v)) & 0xffu) != 0u) un = 0.f;                    // :197-198
                if (MODE == 2) un = uq ? box_clamp(un, bx.lo[v], bx.hi[v]) : 0.f;                   // :200-213
                const float du = uq ? un - ub[v] : 0.f;
                e[v] = uq ? du - Ud[v] - kb[v] : 0.f;
                Ud[v] = un;
                s = fmaf(du, du, s);
            }
            dacc += s;
            if (PADK && L.r == 0) { if (uq) st_u4(p, tb, 4 * L.q, Ud); }
            else if (store && uq) wv::store_f32x4(uo + (long)t * ust + 4 * L.q, Ud);
        }
        PROF40_MARK(12);
        // ---- x+ = F tau' + f   (:216-222)  and  Quu e
        f32x4 H = zero4;
        if (t < T - 1) {
            f32x4 acc[2];
            float fa[2][12];
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                acc[Im] = zero4;
                if (p.f) acc[Im] = wv::lds_f32x4(rec + 320 + 4u * (unsigned)(16 * Im + 4 * L.q));
                const int row = 16 * Im + L.r;
#pragma unroll
                for (int J = 0; J < 2; ++J) {    // (rows of F are 160 B: every such quadruple is 16-byte aligned)
                    const f32x4 x = wv::lds_f32x4(base + POFF_F + 4u * (unsigned)(row * N + 16 * J + 4 * L.q));
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[Im][4 * J + i] = x[i];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)        // the control columns as two full-k operands (uctl)
                    fa[Im][8 + h] = wv::lds_f32(base + POFF_F + 4u * (unsigned)(row * N + 32 + uctl(L.q) + 2 * h));
            }
            wv::sched_fence();
#pragma unroll
            for (int h = 0; h < 2; ++h) H = wv::mfma(aq[h], wv::lower_halves(e[2 * h], e[2 * h + 1]), H);
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int Im = 0; Im < 2; ++Im) acc[Im] = wv::mfma(fa[Im][k], Xd[k >> 2][k & 3], acc[Im]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float up = wv::lower_halves(Ud[2 * h], Ud[2 * h + 1]);
#pragma unroll
                for (int Im = 0; Im < 2; ++Im) acc[Im] = wv::mfma(fa[Im][8 + h], up, acc[Im]);
            }
            wv::sched_fence();
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                if (PADK && L.r == 0) st_x4(p, (long)(t + 1) * p.B + L.b, 16 * Im + 4 * L.q, acc[Im]);
                else if (store) wv::store_f32x4(xo + (long)(t + 1) * xst + 16 * Im + 4 * L.q, acc[Im]);
                const f32x4 xb = wv::lds_f32x4(rec + 160 + 4u * (unsigned)(16 * Im + 4 * L.q));
                Xd[Im] = acc[Im];
#pragma unroll
                for (int v = 0; v < 4; ++v) DXd[Im][v] = acc[Im][v] - xb[v];
            }
        } else {
            wv::sched_fence();
#pragma unroll
            for (int h = 0; h < 2; ++h) H = wv::mfma(aq[h], wv::lower_halves(e[2 * h], e[2 * h + 1]), H);
            wv::sched_fence();
        }
#ifdef MPC_MFMA40_PROF
        asm volatile("" :: "v"(DXd[0][0]), "v"(DXd[1][0]), "v"(H[0]));
        PROF40_MARK(13);
#endif
        // (G was formed with the dx this timestep STARTED from: DXd above is already the next one's)
        {
            float term = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) term = fmaf(e[v], mm[v] + G[v] + 0.5f * H[v], term);
            cacc += (double)(uq ? term : 0.f);
        }
        PROF40_MARK(14);
    }
    wv::dma_wait<0>();
    // column totals: lane groups 0 and 1 hold the two halves of the controls
    double c2 = cacc;
    {
        const float hi = (float)c2, lo = (float)(c2 - (double)hi);
        const double o1 = (double)wv::shfl_xor(hi, 16) + (double)wv::shfl_xor(lo, 16);
        c2 += o1;
        const float hi2 = (float)c2, lo2 = (float)(c2 - (double)hi2);
        c2 += (double)wv::shfl_xor(hi2, 32) + (double)wv::shfl_xor(lo2, 32);
    }
    const double cost = old_cost + w0 + c2;
    const float du2 = sum_q(dacc);
    // first trial that is not worse than the nominal, else the last one (:176-179, 247)
    int win = p.max_ls - 1;
    for (int j = p.max_ls - 1; j >= 0; --j) {
        const double cj = (double)wv::readlane((float)cost, j) + (double)wv::readlane((float)(cost - (double)(float)cost), j);
        if (!(cj > old_cost)) win = j;
    }
    const float full2 = wv::readlane(du2, 0);
    float win_alpha = 1.f;
    for (int i = 0; i < win; ++i) win_alpha *= p.ls_decay;
    if (PADK && win == 1) {
        // padded instantiation: the parked trajectory is [T,B,40] in the kernel's padded slots, the caller's arrays have the
        // true shape -- word by word, sixteen loads in flight
        wv::fence_own_stores();
        enum { CW = 16 };
        const int nword = T * PSCR;
        for (int c0 = 0; c0 < nword; c0 += 64 * CW) {
            float v[CW];
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                const int w = c0 + 64 * i + L.lane, ww = w < nword ? w : nword - 1;
                const int t = ww / PSCR;
                v[i] = scr[((long)t * p.B + L.b) * PSCR + (ww - t * PSCR)];
            }
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                const int w = c0 + 64 * i + L.lane;
                const int ww = w < nword ? w : nword - 1, t = ww / PSCR, e = ww - t * PSCR;
                const long tb = (long)t * p.B + L.b;
                wv::st_buf(p.new_x + tb * p.ns, (unsigned)(4 * p.ns), (w < nword && e < NS) ? (unsigned)(4 * e) : OOB_OFF, v[i]);
                wv::st_buf(p.new_u + tb * p.nc, (unsigned)(4 * p.nc), (w < nword && e >= NS) ? (unsigned)(4 * (e - NS)) : OOB_OFF, v[i]);
            }
        }
    } else if (win == 1) {
        // the parked trajectory -> new_x / new_u: PSCR / 4 = 10 16-byte chunks per timestep (8 of x', 2 of u'), chunk
        // c = lane + 64 i; the tail repeats the last chunk (the same bytes to the same address)
        wv::fence_own_stores();
        enum { CH = 10 };
        const int nchunk = T * (PSCR / 4);
        for (int c0 = 0; c0 < nchunk; c0 += 64 * CH) {
            f32x4 v[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = c0 + 64 * i + L.lane, cc = c < nchunk ? c : nchunk - 1;
                const int t = cc / (PSCR / 4), k = cc - t * (PSCR / 4);
                v[i] = *(const f32x4 *)(scr + ((long)t * p.B + L.b) * PSCR + 4 * k);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = c0 + 64 * i + L.lane, cc = c < nchunk ? c : nchunk - 1;
                const int t = cc / (PSCR / 4), k = cc - t * (PSCR / 4);
                const long tb = (long)t * p.B + L.b;
                wv::store_f32x4(k < NS / 4 ? p.new_x + tb * NS + 4 * k : p.new_u + tb * NC + 4 * (k - NS / 4), v[i]);
            }
        }
    } else if (win != 0) {
        rollout_lean<MODE, true>(p, L, Kin, kin, old_cost, 0.0, win_alpha);
    }
    float wc_hi = wv::readlane((float)cost, 0), wc_lo = wv::readlane((float)(cost - (double)(float)cost), 0);
    float wd = full2;
    for (int j = 1; j < 16; ++j) {                   // (uniform loop; readlane wants a constant lane only in the kernel build)
        const float hj = wv::readlane((float)cost, j), lj = wv::readlane((float)(cost - (double)(float)cost), j);
        const float dj = wv::readlane(du2, j);
        if (j == win) {
            wc_hi = hj;
            wc_lo = lj;
            wd = dj;
        }
    }
    const double wc = (double)wc_hi + (double)wc_lo;
    if (L.lane == 0) {
        int status = 0;
        if (!(wc == wc) || fabs(wc) > 3e38) status |= MPC_ST_NONFINITE;
        if (p.costs) p.costs[L.b] = (float)wc;
        if (p.full_du_norm) p.full_du_norm[L.b] = sqrtf(full2);
        if (p.alpha_du_norm) p.alpha_du_norm[L.b] = sqrtf(wd);
        if (p.alphas) p.alphas[L.b] = win_alpha;
        if (p.status) p.status[L.b] |= status;
    }
}

// ---------------------------------------------------------------------------------------------
// LQRStepFn.backward (mpc/lqr_step.py:312-407) for this shape: the step of its nested problem with the costates riding
// along, then kkt_outer_kernel (kkt_wave.hip) for the outer products.  The reference walks the horizon four times
// (nested sweep, nested rollout, lambda, dlambda); here lambda rides with the sweep (sweep_wave<MODE, true>) and
//     dlambda_t = V_t dx_t + v_t + (1 - alpha) g_t
// -- the costate of the nested LQR problem along its own optimal trajectory is the gradient of its cost-to-go; the
// derivation is in lqr_dpp16_body.h (kkt_fused_wave), which does the same for the 12/4 shape -- with the rollout:
// 16 more MFMAs per timestep (V_{t+1} out of the workspace is its own A operand, the new state tile the B operand)
// instead of a third and fourth pass over C and F.  The nested line search (:176-179, decay 0.2, 10 trials) is decided
// from the sweep's predicted change (2 alpha - alpha^2) w0 like the lean rollout's; column r still rolls alpha = decay^r,
// the winner's column stores dx, du, dlambda.  MODE 0: no bounds; MODE 1: controls on a bound pinned (kkt_pinned_word).
// ---------------------------------------------------------------------------------------------
MPC_DEV void kstage_issue(const P &p, const RStream &d, const char *v_ptr, long v_step, int t, int slot)
{
    const unsigned base = (unsigned)slot * KSTAGE_BYTES;
    const long tl = t;
    const long tf = t < p.T - 1 ? t : (p.T > 1 ? p.T - 2 : 0);      // F has T-1 entries
    const long tx = t + 1 < p.T ? t + 1 : t;                         // (V, v, g) of t+1
    // (padded instantiation: the caller's F by the rollout's gather; K, the record and V are the workspace's own padded layout)
    if (PADK) gather_F(d.g, p.T > 1 ? d.Fb + tf * p.F_st : d.Cb, p.T > 1 ? d.g.fbytes : 0u, base + KOFF_F);
    else dma_kib<5>(d.f_ptr + tf * d.f_step + d.lo, base + KOFF_F);
    wv::dma16(d.k_ptr + tl * d.k_step + d.lo, base + KOFF_K);
    wv::dma16_if(d.r_active, d.r_ptr + rec_off(d.r_is_x ? tx : tl, d.r_step), base + KOFF_R);
    if (PADK) {
        // u*_t | u_lower_t | u_upper_t (tensor bounds; u* again otherwise: the instruction count of a stage is fixed), lane a < n_ctrl
        const int nc = p.nc;
        const unsigned lane = d.lo >> 4;
        const bool act = (int)lane < nc;
        const long o = ((long)tl * p.B + d.bidx) * nc + (act ? (int)lane : 0);
        const bool tb_ = p.bound_mode == MPC_BOUND_TENSOR;
        wv::dma4_if(act, p.cur_u + o, base + KOFF_R);
        wv::dma4_if(act, (tb_ ? p.lo : p.cur_u) + o, base + KOFF_R + 32);
        wv::dma4_if(act, (tb_ ? p.hi : p.cur_u) + o, base + KOFF_R + 64);
    }
#ifdef MPC_KF40_VFULL
    dma_kib<4>(v_ptr + tx * v_step + d.lo, base + KOFF_V);
#else
    {
        // tiles (0,0), (1,0), (1,1): KiB 0, 2, 3 of the record (the slot keeps its 4 KiB; KiB 1 is never written nor read)
        const char *vp = v_ptr + tx * v_step + d.lo;
        wv::dma16_at<0>(vp, base + KOFF_V);
        wv::dma16_at<2048>(vp, base + KOFF_V);
        wv::dma16_at<3072>(vp, base + KOFF_V);
    }
#endif
}

template <int MODE>
MPC_DEV void kkt_pass2(const P &p, const Lane &L, const float *Kin, const float *kin, const KktArgs40 &kx, double w0,
                       const float (&v0g0)[4])
{
    const int T = p.T;
    // record: lanes 10..17 v_{t+1} | 20..27 g_{t+1} | 28..29 k_t
    RStream d;
    d.c_ptr = nullptr; d.c_step = 0;
    d.lo = 16u * (unsigned)L.lane;
    d.bidx = L.b;
    if (PADK) {
        d.Cb = p.C + (long)L.b * p.C_sb;
        d.Fb = T > 1 ? p.F + (long)L.b * p.F_sb : d.Cb;
        gather_init(d.g, p, L.lane);
    }
    // the caller's rows by their true length (padded instantiation)
    const int ns_o = PADK ? p.ns : NS, dfb = PADK ? p.ns * (p.ns + p.nc) : NS * N;
    d.f_ptr = T > 1 ? (const char *)(p.F + (long)L.b * p.F_sb) : (const char *)(p.C + (long)L.b * p.C_sb);
    d.f_step = T > 1 ? p.F_st * 4 : 0;
    d.k_ptr = (const char *)(Kin + (long)L.b * (NC * NS));
    d.k_step = (long)p.B * NC * NS * 4;
    d.r_is_f = false;
    d.r_is_x = L.lane < 28;
    d.r_active = (L.lane >= 10 && L.lane < 18) || (L.lane >= 20 && L.lane < 30);
    if (L.lane >= 28) {
        d.r_ptr = (const char *)(kin + (long)L.b * NC) + 16 * ((L.lane < 30 ? L.lane : 28) - 28);
        d.r_step = (unsigned)((long)p.B * NC * 4);
    } else {
        const int g = L.lane >= 20 ? 8 + (L.lane - 20) : (L.lane >= 10 && L.lane < 18 ? L.lane - 10 : 0);
        d.r_ptr = (const char *)(kx.vgws + (long)L.b * 64) + 16 * g;
        d.r_step = (unsigned)((long)p.B * 64 * 4);
    }
    const char *v_ptr = (const char *)(kx.Vws + (long)L.b * 1024);
    const long v_step = (long)p.B * 4096;

    float alpha = 1.f;
    for (int i = 0; i < L.r; ++i) alpha *= p.ls_decay;            // column r tries decay^r
    // first trial that does not make the nested cost worse, else the last one (:176-179, 247)
    int win = p.max_ls - 1;
    {
        float a = 1.f;
        for (int j = 0; j < p.max_ls; ++j) {
            if (!((2.0 * (double)a - (double)a * (double)a) * w0 > 0.0)) { win = j; break; }
            a *= p.ls_decay;
        }
    }
    const bool store = L.r == win;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // dx_init = -dlambda_0 = -(V_0 0 + v_0 + (1 - alpha) g_0)   (:404)
    {
        // v0g0 is in row layout (lane r of every lane group holds entry 16J + r); the winner's alpha is a wave-uniform number
        float wa = 1.f;
        for (int i = 0; i < win; ++i) wa *= p.ls_decay;
        if (L.q == 0) {
#pragma unroll
            for (int J = 0; J < 2; ++J)
                if (!PADK || 16 * J + L.r < ns_o) kx.dx_init[(long)L.b * ns_o + 16 * J + L.r] = -fmaf(1.f - wa, v0g0[2 + J], v0g0[J]);
        }
    }
    f32x4 Xd[2];                     // dx_t, D layout: column r = trial r (dx_0 = 0: the nested x_init, :327, 338)
    Xd[0] = zero4;
    Xd[1] = zero4;
    f32x4 pU = zero4, pDL[2] = {zero4, zero4};     // du_{t-1}, dlambda_t: stored one timestep late (below)
    if (store) {
#pragma unroll
        for (int I = 0; I < 2; ++I) st_x4(p, L.b, 16 * I + 4 * L.q, zero4);
    }
    wv::dma_wait<0>();          // nothing of the sweep may still land in the ring
#pragma unroll
    for (int i = 0; i < KSLOTS - 1; ++i) kstage_issue(p, d, v_ptr, v_step, i < T ? i : T - 1, i);
    for (int t = 0; t < T; ++t) {
        wv::dma_wait<(KSLOTS - 2) * KDMA_PER_STAGE>();
        const unsigned base = (unsigned)(t % KSLOTS) * KSTAGE_BYTES;
        const long tb = (long)t * p.B + L.b;
        const unsigned rec = base + KOFF_R;
        // ---- du = K dx + alpha k   (:192 around the zero nominal)
        f32x4 Ud = zero4;
        float a[8];
#pragma unroll
        for (int J = 0; J < 2; ++J) {        // four consecutive words of K's row r per lane and state tile: one 16-byte LDS read
            const f32x4 x = wv::lds_f32x4(base + KOFF_K + 4u * (unsigned)((L.r < NC ? L.r : 0) * NS + 16 * J + 4 * L.q));
#pragma unroll
            for (int i = 0; i < 4; ++i) a[4 * J + i] = L.r < NC ? x[i] : 0.f;
        }
        const unsigned qo = 16u * (unsigned)(L.q < 2 ? L.q : 0);
        const f32x4 kb = wv::lds_f32x4(rec + 448 + qo);
        // every operand of this timestep out of its stage first (F's rows as A operands, V_{t+1}'s tiles, v and g), then the
        // matrix-core blocks undivided
        float fa[2][12];
#pragma unroll
        for (int Im = 0; Im < 2; ++Im) {
            const int row = 16 * Im + L.r;
#pragma unroll
            for (int J = 0; J < 2; ++J) {    // (rows of F are 160 B: every such quadruple is 16-byte aligned)
                const f32x4 x = wv::lds_f32x4(base + KOFF_F + 4u * (unsigned)(row * N + 16 * J + 4 * L.q));
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[Im][4 * J + i] = x[i];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)        // the control columns as two full-k operands (uctl)
                fa[Im][8 + h] = wv::lds_f32(base + KOFF_F + 4u * (unsigned)(row * N + 32 + uctl(L.q) + 2 * h));
        }
        f32x4 Vt[2][2];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J) {
#ifndef MPC_KF40_VFULL
                if (I == 0 && J == 1) continue;
#endif
                Vt[I][J] = wv::lds_f32x4(base + KOFF_V + 1024u * (unsigned)(2 * I + J) + 16u * (unsigned)L.lane);
            }
#ifndef MPC_KF40_VFULL
        // tile (0,1), register v of lane (q,r) = V[4q+v][16+r] = V[16+r][4q+v] = register r & 3 of lane (r >> 2, 4q + v) of tile (1,0)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            Vt[0][1][v] = wv::lds_f32(base + KOFF_V + 2048u + 16u * (unsigned)(16 * (L.r >> 2) + 4 * L.q + v) + 4u * (unsigned)(L.r & 3));
#endif
        f32x4 DL[2];
#pragma unroll
        for (int Im = 0; Im < 2; ++Im) {
            const f32x4 v1 = wv::lds_f32x4(rec + 160 + 4u * (unsigned)(16 * Im + 4 * L.q));
            const f32x4 g1 = wv::lds_f32x4(rec + 320 + 4u * (unsigned)(16 * Im + 4 * L.q));
#pragma unroll
            for (int v = 0; v < 4; ++v) DL[Im][v] = fmaf(1.f - alpha, g1[v], v1[v]);
        }
        {
            const int tn = t + KSLOTS - 1;
            kstage_issue(p, d, v_ptr, v_step, tn < T ? tn : T - 1, tn % KSLOTS);
        }
        // what the PREVIOUS timestep produced leaves now: the wait at the top of the next trip counts vector stores with the
        // DMAs, so stores issued here have a timestep of arithmetic to land in (issued at the end of their own timestep they
        // were waited out at once)
        if (store && t > 0) {
            const long tbp = tb - p.B;
            if (L.q < 2) st_u4(p, tbp, 4 * L.q, pU);
#pragma unroll
            for (int Im = 0; Im < 2; ++Im) {
                st_x4(p, tb, 16 * Im + 4 * L.q, Xd[Im]);
                st_row4(kx.dF + tbp * (long)dfb + ns_o, ns_o, 16 * Im + 4 * L.q, pDL[Im]);
                if (kx.df) st_row4(kx.df + tbp * ns_o, ns_o, 16 * Im + 4 * L.q, f32x4{-pDL[Im][0], -pDL[Im][1], -pDL[Im][2], -pDL[Im][3]});   // :397-400
            }
        }
        wv::sched_fence();
        {
            f32x4 U2 = zero4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Ud = wv::mfma(a[k], Xd[0][k], Ud);
                U2 = wv::mfma(a[4 + k], Xd[1][k], U2);
            }
            wv::sched_fence();
#pragma unroll
            for (int v = 0; v < 4; ++v) Ud[v] += U2[v];
        }
        {
            unsigned zw = 0u;                    // the pinned flags of controls 4q .. 4q+3
            if (MODE == 1) {
                unsigned zlo, zhi;
                if (PADK) {
                    kkt_pinned_lds(p, rec, rec + 32, rec + 64, L.lane, zlo, zhi);
                } else {
                    zlo = kkt_pinned_word(p, tb, 0);
                    zhi = kkt_pinned_word(p, tb, 1);
                }
                zw = L.q == 0 ? zlo : zhi;
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float un = L.q < 2 ? fmaf(alpha, kb[v], Ud[v]) : 0.f;
                if (MODE == 1 && ((zw >> (8 * v)) & 0xffu) != 0u) un = 0.f;                          // :197-198
                Ud[v] = un;
            }
            pU = Ud;
        }
        // ---- dx_{t+1} = F dtau   (f = None, :333), dlambda_{t+1} = V_{t+1} dx_{t+1} + v_{t+1} + (1 - alpha) g_{t+1}
        if (t < T - 1) {
            f32x4 acc[2] = {zero4, zero4};
            wv::sched_fence();
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int Im = 0; Im < 2; ++Im) acc[Im] = wv::mfma(fa[Im][k], Xd[k >> 2][k & 3], acc[Im]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float up = wv::lower_halves(Ud[2 * h], Ud[2 * h + 1]);
#pragma unroll
                for (int Im = 0; Im < 2; ++Im) acc[Im] = wv::mfma(fa[Im][8 + h], up, acc[Im]);
            }
            // (A operand = V through its symmetry: register v of tile (I', Im), exactly as Y = V F in the sweep)
#pragma unroll
            for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int Im = 0; Im < 2; ++Im) DL[Im] = wv::mfma(Vt[Ip][Im][v], acc[Ip][v], DL[Im]);
            wv::sched_fence();
            Xd[0] = acc[0];
            Xd[1] = acc[1];
            pDL[0] = DL[0];
            pDL[1] = DL[1];
        }
    }
    if (store && L.q < 2) st_u4(p, (long)(T - 1) * p.B + L.b, 4 * L.q, pU);
    wv::dma_wait<0>();
}

template <int MODE> MPC_DEV void kkt_fused_wave(const P &p, float *K, float *k, const KktArgs40 &kx)
{
    Lane L;
    L.lane = wv::lane();
    L.r = L.lane & 15;
    L.q = L.lane >> 4;
    L.b = wv::problem();
    if (L.b >= p.B) return;
    double w0 = 0.0;
    float v0g0[4] = {0.f, 0.f, 0.f, 0.f};
    (void)sweep_wave<MODE, true>(p, K, k, &w0, &kx, v0g0);
    wv::fence_own_stores();            // K, k, V, v, g come back through the DMA
#ifdef MPC_KF40_P1_ONLY                 // (diagnostic build: where the time goes, tools/ab_kkt40_phases.sh)
    if (w0 != 1.2345e300) return;
#endif
    kkt_pass2<MODE>(p, L, K, k, kx, w0, v0g0);
}

template <int MODE> MPC_DEV void step_wave(const P &p, float *K, float *k)
{
    double w0 = 0.0;
    bool on_dyn = false;
#ifdef MPC_MFMA40_PROF
    Prof40 prof_;
    for (int i_ = 0; i_ < 16; ++i_) prof_.acc[i_] = 0;
    prof_.last = wv::clock();
    const double old_cost = sweep_wave<MODE>(p, K, k, &w0, nullptr, nullptr, &on_dyn, &prof_);
    struct ProfOut {
        const P &p; float *K; Prof40 &pr;
        MPC_DEVM ~ProfOut()
        {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            pr.acc[7] += wv::clock() - pr.last;
            if (wv::lane() == 0 && wv::problem() < p.B)
                for (int i_ = 0; i_ < 16; ++i_) K[(long)wv::problem() * 16 + i_] = (float)pr.acc[i_];
        }
    } prof_out_{p, K, prof_};
#else
    const double old_cost = sweep_wave<MODE>(p, K, k, &w0, nullptr, nullptr, &on_dyn);
#endif
    if (p.sweep_only) return;           // MPC_OPT_SWEEP_ONLY (the sweep has written K, k, old_costs, qp_iters, status)
    wv::fence_own_stores();
#ifdef MPC_CFG5_SWEEP_ONLY
    if (old_cost == 1.2345e300) return;    // (diagnostic build: the sweep alone; keeps old_cost / w0 alive)
    if (old_cost != 1.2345e300) { if (wv::lane() == 0 && p.costs) p.costs[wv::problem()] = (float)(old_cost + w0); return; }
#endif
    if (MODE == 0 && on_dyn) {          // vouched for by the caller, or verified by the sweep
        Lane L;
        L.lane = wv::lane();
        L.r = L.lane & 15;
        L.q = L.lane >> 4;
        L.b = wv::problem();
        if (L.b >= p.B) return;
        rollout_lean<0, false>(p, L, K, k, old_cost, w0);
    } else if (MODE != 0 && on_dyn && p.Kk) {
        // constrained, vouched for, and the caller left room for the (M, Quu, m) record: priced without a pass over C
        Lane L;
        L.lane = wv::lane();
        L.r = L.lane & 15;
        L.q = L.lane >> 4;
        L.b = wv::problem();
        if (L.b >= p.B) return;
#ifdef MPC_MFMA40_PROF
        rollout_priced<MODE>(p, L, K, k, old_cost, w0, &prof_);
#else
        rollout_priced<MODE>(p, L, K, k, old_cost, w0);
#endif
    } else {
        rollout_wave<MODE>(p, K, k, old_cost);
    }
}

}  // namespace mfma40
}  // namespace mpclqr
