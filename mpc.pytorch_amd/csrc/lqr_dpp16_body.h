// lqr_dpp16_body.h -- the fused LQR step at the headline shape (n_state = 12, n_ctrl = 4, fp32):
// FOUR problems per 64-lane wavefront, one problem per 16-lane DPP row, lane j of a row owning
// variable j of tau = [x_0..x_11, u_0..u_3], i.e. column j of every matrix.
//
//   matrix x matrix   batched 4x4 outer products on the matrix core: v_mfma_f32_4x4x1_16b_f32 with cbsz:2 takes
//                     the A vector of the four blocks of a row from lanes 4*abid..+3 of that row and leaves B in
//                     its lane, which is rows 4*abid..+3 of an outer product with one column per lane -- exactly
//                     this layout, so V F, C + F'(VF) and Qxx + Qxu K are sums of such products with no data
//                     movement (wv::mfma4, outer_acc below); 256 FMAs per issue slot
//   matrix x vector   v_fmac_f32 with a DPP row_newbcast operand (one lane of the row broadcast to the other 15
//                     inside the FMA itself): c_back, F'v, the rollout's K dx and F tau
//   4x4 control block LDL', the box QP and the K solve on row-uniform values, shared by four problems per
//                     instruction; nothing is computed 64 times over
//
// With one wavefront per SIMD every instruction costs an issue slot of >= 4.6 clocks whatever its kind
// (tools/ubench/valu_rate.hip): the kernel is bound by its instruction count, and the 4x4x1 form does the
// 384 product FMAs of a timestep in 96 instructions.  lqr_mfma16_body.h (one wave per problem on the 16x16x4
// MFMA, all shapes <= 12/4) is kept as impl 2; DESIGN.md has the measured comparison.
//
// Written against the `wv::` wave interface like lqr_mfma16_body.h: lqr_dpp16.hip binds it to gfx950
// (MFMA builtins, inline-asm blocks of v_fmac_f32_dpp, global_load_lds DMA), tests/emu/ to the host emulator.
//
// What it replaces in locuslab/mpc.pytorch:
//   sweep_step      mpc/lqr_step.py:284-296 (c_back) + :52-160 (lqr_backward)
//   pnqp4           mpc/pnqp.py:5-82 (shared with lqr_mfma16_body.h)
//   rollout_step    mpc/lqr_step.py:164-261 (lqr_forward), mpc/util.py:129-153
//
// Register layout of one row (lane j = variable j):
//   Vc[i]  = V[i][j]      column j of the value Hessian (j < 12)       vv = v[j]
//   Cc[i]  = C[j][i]      (C symmetric, mpc/mpc.py:61-68: row j is read as column j)
//   Fc[m]  = F[m][j]      column j of the dynamics
//   Y[i]   = (V F)[i][j]  = sum_m V[i][m] F[m][j]:   outer_acc(Y, Vc[m], Fc[m])  (V symmetric: lane i holds V[m][i])
//   Q[i]   = (C + F'VF)[i][j] = Cc[i] + sum_m F[m][i] Y[m][j]:   outer_acc(Q, Fc[m], Y[m]);   q = q[j]
//   K[a]   = K[a][j]  (j < 12),  lane 12 carries k = K[.][12]
// Staging: per wave an LDS ring filled by global_load_lds ahead of use with counted vmcnt waits: the sweep 4 slots
// of 9 KiB (4 x {C 1 KiB, F 768 B, small-vector record 256 B}), the rollout 7 (6) packed slots, see RollRing.
//
// One read of C.  The rollout does NOT stream C again to price its trajectory: for a rollout that obeys
// the dynamics (it does by construction) around a nominal that obeys them too, the exact identity
//     J(tau') = J(nominal) + w_0 + sum_t [ e_t'(m_t + M_t dx_t) + 0.5 e_t' Quu_t e_t ],
//     e_t = du_t - K_t dx_t - k_t,  m = qu + Quu k,  M = Qux + Quu K,  w_0 = sum_t (0.5 k'Quu k + qu'k)
// (telescoping Q_t(dx,du) - V_t(dx) over the sweep's own value function) needs only what the sweep
// already holds: Quu_t rides in the spare 48 bytes of the gain record, (m, M) in a second record when
// constraints make them non-zero.  Whether the nominal obeys the dynamics is CHECKED in the rollout
// (|F tau + f - x_next| <= 1e-5 (1 + |x|), one more 16-term product per step); if any problem of the
// wave fails it, the wave prices its rollout the reference's way, 0.5 tau'C tau + c'tau, from a second
// stream of C (mpc/lqr_step.py:230-232).  Either way `costs` is the reference's quantity.
#pragma once
#include <math.h>
#include "lqr_params.h"
#include "lqr_small_math.h"

// statistics hooks of the host emulator build (tests/emu): no-ops in the kernel
#ifndef MPC_STAT
#define MPC_STAT(i)
#endif

// member functions: MPC_DEV is `static inline` in the host emulator build, which a member cannot be
#ifndef MPC_DEVM
#define MPC_DEVM MPC_DEV
#endif

#ifndef MPC_DPP16_NSTAGE
#define MPC_DPP16_NSTAGE 4
#endif
// 1: start the box QP of timestep t from the solution of timestep t+1 like the reference (rounds 1-5; kept for the A/B)
#ifndef MPC_DPP16_QP_WARM
#define MPC_DPP16_QP_WARM 0
#endif

// ---- The PADDED instantiation (round 6; -DMPC_DPP16_PAD, the third compilation of lqr_dpp16.hip): any n_state <= 12, n_ctrl <= 4 --
// Rounds 1-5 ran every shape of that range other than exactly 12/4 on the one-problem-per-wavefront kernel (lqr_mfma16_body.h) at
// half this kernel's rate.  Here tau is padded to [x(12); u(4)] -- state i in lane i, control a in lane 12 + a, zeros elsewhere, an
// identity on the padded diagonal of Quu -- and the padding is done BY THE STAGING DMA, as in the padded 32/8 kernel
// (lqr_mfma40_body.h): C and F are gathered dword by dword with `buffer_load ... lds`, whose per-lane source offset is free and whose
// out-of-range lanes write ZERO into LDS; the small vectors of the record by per-lane `global_load_lds_dword`.  LDS holds the dense
// 16 x 16 / 12 x 16 blocks in the permuted layout the kernel was written for and nothing downstream of the staging changes; only
// what touches the caller's arrays directly (x_init, tensor bounds, masks, qp_start, the trajectory and gain outputs) indexes by the
// true shape.  A stage is 32 DMA instructions instead of 8 (vmcnt is six bits wide): the two-slot ring, one stage ahead -- what the
// exact kernel's unconstrained step runs on anyway.  No 16-byte alignment is asked of anybody's blocks.
#ifdef MPC_DPP16_PAD
#define MPC_DPP16_PADK 1
#else
#define MPC_DPP16_PADK 0
#endif

namespace mpclqr {
namespace dpp16 {
constexpr bool PADK = MPC_DPP16_PADK != 0;
constexpr unsigned OOB_OFF = 0x7fff0000u;          // a source offset beyond every block: the gather writes zero there
enum { PAD_BIAS = 4096 };                          // see wv::dma_buf_at: source offsets are stored as offset - IMM + PAD_BIAS (never negative)
// slot of the padded tau = [x(12); u(4)] -> index in the caller's tau = [x(ns); u(nc)], or -1 (padding)
MPC_DEV int pad_tau(int pj, int ns, int nc) { return pj < 12 ? (pj < ns ? pj : -1) : (pj - 12 < nc ? ns + (pj - 12) : -1); }

// Diagnostic builds only (-DMPC_DPP16_PROF, tools/prof_phases.py): shader-clock totals of the phases of the two
// loops, per wave, written where K would go.  Every probe drains the LDS / scalar queue (s_memtime), so the phases
// are timed back to back, not overlapped -- an upper bound for each.
#ifdef MPC_DPP16_PROF
struct Prof { unsigned long long acc[16]; unsigned long long last; };
#define PROF_DECL Prof prof_; for (int i_ = 0; i_ < 16; ++i_) prof_.acc[i_] = 0; prof_.last = wv::clock()
#define PROF_ARG , Prof &prof_
#define PROF_PASS , prof_
#define PROF_MARK(slot) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = wv::clock(); prof_.acc[slot] += n_ - prof_.last; prof_.last = n_; } while (0)
#define PROF_MARK_ALL(slot) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = wv::clock(); prof_.acc[slot] += n_ - prof_.last; prof_.last = n_; } while (0)
#else
#define PROF_DECL
#define PROF_ARG
#define PROF_PASS
#define PROF_MARK(slot)
#define PROF_MARK_ALL(slot)
#endif

typedef StepParams<float> P;
using wv::f32x4;

// MODE of the step kernel:
//   0  unconstrained, horizon T <= RG_STEPS: the gains of the whole horizon stay in the register file between the
//      sweep and the rollouts (no record written, none read back: 512 of the 3,376 bytes a problem-step moves)
//   1  unconstrained + u_zero_I          2  box-constrained (pnqp in the sweep)
//   3  unconstrained, any T: gains through the record in memory like modes 1 and 2
constexpr bool con(int MODE) { return MODE == 1 || MODE == 2; }       // constrained: (m, M) terms, masks, clamps
constexpr bool rgm(int MODE) { return MODE < 1; }                     // register-resident gains

// The gain record of timestep t as the sweep holds it: lane j < 12 K[0..3][j], lane 12 k[0..3], lanes 13..15 columns
// 1..3 of Quu (Quu[0][0] in element 2 of lane 13).  Registers cannot be indexed by a runtime t, so put / get are
// switches over t whose cases name their registers statically: a compare tree of ~12 scalar instructions around four
// register moves (the array lives in the accumulation half of the 512-entry file).
enum { RG_STEPS = 64 };
// (the storage itself is the wave interface's: wv::rg_put / wv::rg_get name accumulation registers a[4t .. 4t+3] in
// inline assembly on the GPU -- as compiler-visible values the 256 registers were copied around at every loop
// boundary, 112 us against 99 -- and a plain per-lane array in the host emulator)
template <bool ON> struct Gains {};
MPC_DEV void gain_put(Gains<true> &, int t, f32x4 v) { wv::rg_put(t, v); }
MPC_DEV f32x4 gain_get(const Gains<true> &, int t) { return wv::rg_get(t); }
MPC_DEV void gain_put(Gains<false> &, int, f32x4) {}
MPC_DEV f32x4 gain_get(const Gains<false> &, int) { return f32x4{0.f, 0.f, 0.f, 0.f}; }
using mfma16::Sym4;
using mfma16::Ldl4;
using mfma16::ldl4;
using mfma16::ldl4_solve;
using mfma16::eclampf;
using mfma16::sel;

// ---------------------------------------------------------------------------
// The same projected-Newton box QP for the kernel that keeps one problem per 16-lane ROW (lqr_dpp16_body.h):
// four unknowns, every lane of a row computes its problem's (row-uniform) numbers, the four rows of a wave
// differ.  Same iterates and the same answer as pnqp4 / mpc/pnqp.py:5-82; what differs is how the wave gets there:
//   * ONE wave-uniform loop.  A row that has converged freezes its x; every later trip then recomputes that
//     row's gradient, free set and factorisation from the same numbers, so what the last trip leaves behind is
//     valid for all four rows -- no per-row exit, hence no exec-mask bookkeeping, no state carried in lane masks.
//   * the confirming iteration costs a gradient, not a factorisation.  The reference stops in the iteration
//     whose Newton step is shorter than 1e-4 (:56-59).  After a FULL Newton step (x + dx inside the box) the
//     gradient vanishes on the free set, so if the free set of the new point is the one just factorised the next
//     step is zero to rounding (~1e-7): that iteration is recognised from the free-set test alone and returns
//     the factorisation it would have recomputed.
//   * free-set flags are numbers (0 / 1) and the masked matrix is built by multiplication, not by selects on
//     combined lane masks (see ldl4).
// ---------------------------------------------------------------------------
// ---- Gauss-Jordan in the quad layout: the factorisation of the QP's trips (round 4; the 32/8 kernel's Gj8V on four unknowns).
// A trip solves ONE system whose right-hand side lives like the QP's vectors (quad a of the row: entry a).  On row-uniform
// scalars that was ten products for the masked matrix, the LDL' (19), four broadcasts of the right-hand side, sixteen multiply-adds
// of triangular solves and three selects to spread the answer.  With column c of the matrix as ONE register (quad a: A[a][c])
// and elimination above the pivot as well, the right-hand side rides along as a fifth column and the answer is where it is
// wanted.  The multipliers are a factorisation too: the K solve behind the QP applies them to its per-lane right-hand sides.
struct Gj4 {
    float nl[4];     // quad a: -(A[a][c] / d_c) as it stood at pivot c (the multiplier of row a), 0 in quad c itself
    float inv[4];    // 1 / d_c (row-uniform)
};
// z = A^-1 z for per-lane right-hand sides z[0..3] (every lane its own), from the multipliers
MPC_DEV void gj4_solve(const Gj4 &g, float (&z)[4])
{
#define MPC_GJ4_APPLY(c, a) wv::fmac_bcast_settled<4 * (a)>(z[a], g.nl[c], z[c])
    MPC_GJ4_APPLY(0, 1); MPC_GJ4_APPLY(0, 2); MPC_GJ4_APPLY(0, 3);
    MPC_GJ4_APPLY(1, 0); MPC_GJ4_APPLY(1, 2); MPC_GJ4_APPLY(1, 3);
    MPC_GJ4_APPLY(2, 0); MPC_GJ4_APPLY(2, 1); MPC_GJ4_APPLY(2, 3);
    MPC_GJ4_APPLY(3, 0); MPC_GJ4_APPLY(3, 1); MPC_GJ4_APPLY(3, 2);
#undef MPC_GJ4_APPLY
#pragma unroll
    for (int a = 0; a < 4; ++a) z[a] *= g.inv[a];
}
#ifdef MPC_DPP16_QP_LDL           // (the round-3 form of the QP's linear algebra, kept for A/B timing)
using QpFac4 = Ldl4;
#else
using QpFac4 = Gj4;
#endif

MPC_DEV int pnqp4_rows(const Sym4 &s, const float q[4], const float lb[4], const float ub[4], int n_iter, int j,
                       float x[4], bool fr_out[4], QpFac4 &f, bool &converged)
{
    // Everything that is one number per unknown (x, g, the bounds, the free-set flags, the step) lives as ONE register
    // with unknown a in the lanes of quad a (lanes 4a..4a+3 of the row): a clamp, a compare, a gradient update is one
    // instruction instead of four.  H x is four multiply-adds whose x operand is a DPP broadcast of lane 4b against
    // column b of H (quad a holds H[a][b]); sums over the unknowns are two row rotations (wv::ring_sum).  Only the
    // factorisation and its triangular solves stay row-uniform scalars, fed by broadcasts.
    const float reg = 1e-11f;                                       // :47
    const float dg[4] = {s.s00 + reg - 1.f, s.s11 + reg - 1.f, s.s22 + reg - 1.f, s.s33 + reg - 1.f};
    const bool q0 = j < 4, q1 = j < 8, q2 = j < 12;
#define MPC_QV(v0, v1, v2, v3) (q0 ? (v0) : (q1 ? (v1) : (q2 ? (v2) : (v3))))
    const float Hc[4] = {MPC_QV(s.s00, s.s01, s.s02, s.s03), MPC_QV(s.s01, s.s11, s.s12, s.s13),
                         MPC_QV(s.s02, s.s12, s.s22, s.s23), MPC_QV(s.s03, s.s13, s.s23, s.s33)};
    const float lbv = MPC_QV(lb[0], lb[1], lb[2], lb[3]), ubv = MPC_QV(ub[0], ub[1], ub[2], ub[3]);
#ifndef MPC_DPP16_QP_LDL
    const float dgv = MPC_QV(dg[0], dg[1], dg[2], dg[3]);
    const bool qd[4] = {q0, q1 && !q0, q2 && !q1, !q2};            // "this lane belongs to quad c"
#endif
    float xv = MPC_QV(x[0], x[1], x[2], x[3]);
    float mv = -1.f;                                                // free set of the last factorisation (none yet)
    float done = 0.f, conv = 0.f, full = 0.f, it_ret = (float)(n_iter - 1);
    // The gradient H x + q (:29) is carried along: every step d is followed by g += H d, and H d is also what the
    // Armijo test of a projected step needs (:61-76) -- one product serves both.
    // (same order of additions as sym4_mv: the products are bit-identical to the row-uniform form)
    // (the first product reads v through the compiler's DPP form, which waits out v's producer; the three behind it on the
    // accumulator's chain find v settled)
#define MPC_HV(out, v) do { out = wv::bcast<0>(v) * Hc[0]; wv::fmac_bcast_settled<4>(out, v, Hc[1]); \
                            wv::fmac_bcast_settled<8>(out, v, Hc[2]); wv::fmac_bcast_settled<12>(out, v, Hc[3]); } while (0)
    float gv;
    MPC_HV(gv, xv);
    gv += MPC_QV(q[0], q[1], q[2], q[3]);
    MPC_STAT(4);
    for (int it = 0; it < n_iter; ++it) {
        if (done == 0.f) MPC_STAT(0);
        MPC_STAT(5);
        // :32  clamped = (x == lb & g > 0) | (x == ub & g < 0)
        const float r_lo = (xv == lbv) ? gv : -1.f;
        const float r_hi = (xv == ubv) ? -gv : -1.f;
        const float mnv = (fmaxf(r_lo, r_hi) > 0.f) ? 0.f : 1.f;
        const float diff = wv::ring_sum(fabsf(mnv - mv));
        // the free set just factorised, reached by a full Newton step: the step from here is zero -- converged (:56-59)
        const bool confirmed = (diff == 0.f) & (full != 0.f) & (done == 0.f);
        it_ret = confirmed ? (float)it : it_ret;
        conv = confirmed ? 1.f : conv;
        done = confirmed ? 1.f : done;
        if (!wv::any(done == 0.f)) break;
        if (done == 0.f) MPC_STAT(1);
        MPC_STAT(6);
        // :44-54  H_ = H on the free block (+1e-11 I, identity elsewhere), dx = -H_^-1 g_
#ifdef MPC_DPP16_QP_LDL
        const float mn[4] = {wv::bcast<0>(mnv), wv::bcast<4>(mnv), wv::bcast<8>(mnv), wv::bcast<12>(mnv)};
        Ldl4 fn;
        {
            const float a10 = (mn[0] * mn[1]) * s.s01, a20 = (mn[0] * mn[2]) * s.s02, a30 = (mn[0] * mn[3]) * s.s03;
            const float a21 = (mn[1] * mn[2]) * s.s12, a31 = (mn[1] * mn[3]) * s.s13, a32 = (mn[2] * mn[3]) * s.s23;
            const float a00 = fmaf(mn[0], dg[0], 1.f), a11 = fmaf(mn[1], dg[1], 1.f);
            const float a22 = fmaf(mn[2], dg[2], 1.f), a33 = fmaf(mn[3], dg[3], 1.f);
            fn.i0 = wv::rcp(a00);
            fn.l10 = a10 * fn.i0; fn.l20 = a20 * fn.i0; fn.l30 = a30 * fn.i0;
            const float d1 = fmaf(-fn.l10, a10, a11);
            fn.i1 = wv::rcp(d1);
            const float t21 = fmaf(-fn.l20, a10, a21);
            const float t31 = fmaf(-fn.l30, a10, a31);
            fn.l21 = t21 * fn.i1; fn.l31 = t31 * fn.i1;
            const float d2 = fmaf(-fn.l21, t21, fmaf(-fn.l20, a20, a22));
            fn.i2 = wv::rcp(d2);
            const float t32 = fmaf(-fn.l31, t21, fmaf(-fn.l30, a20, a32));
            fn.l32 = t32 * fn.i2;
            const float d3 = fmaf(-fn.l32, t32, fmaf(-fn.l31, t31, fmaf(-fn.l30, a30, a33)));
            fn.i3 = wv::rcp(d3);
        }
        const float rv = mnv * gv;
        float y[4];
        ldl4_solve(fn, wv::bcast<0>(rv), wv::bcast<4>(rv), wv::bcast<8>(rv), wv::bcast<12>(rv), y);
        const float dxv = -(mnv * MPC_QV(y[0], y[1], y[2], y[3]));
#else
        Gj4 fn;
        float col[4];
#define MPC_GJ4_COL(c) do { col[c] = (mnv * wv::bcast<4 * (c)>(mnv)) * Hc[c]; col[c] = qd[c] ? dgq : col[c]; } while (0)
        const float dgq = fmaf(mnv, dgv, 1.f);
        MPC_GJ4_COL(0); MPC_GJ4_COL(1); MPC_GJ4_COL(2); MPC_GJ4_COL(3);
#undef MPC_GJ4_COL
        // (rows off the free set are identity rows with a zero right-hand side: their entry of the solution is 0 as it comes)
        float sol = mnv * gv, invd = 0.f;
#define MPC_GJ4_PIVOT(c) do { const float inv = wv::rcp(wv::bcast<4 * (c)>(col[c])); fn.inv[c] = inv; \
                              const float t = -(col[c] * inv); fn.nl[c] = qd[c] ? 0.f : t; invd = qd[c] ? inv : invd; } while (0)
#define MPC_GJ4_ELIM(c, m) wv::fmac_bcast_settled<4 * (m)>(col[m], col[c], fn.nl[c])   /* the unpivoted block stays symmetric */
#define MPC_GJ4_RHS(c) sol = fmaf(wv::bcast<4 * (c)>(sol), fn.nl[c], sol)
        MPC_GJ4_PIVOT(0); MPC_GJ4_ELIM(0, 1); MPC_GJ4_ELIM(0, 2); MPC_GJ4_ELIM(0, 3); MPC_GJ4_RHS(0);
        MPC_GJ4_PIVOT(1); MPC_GJ4_ELIM(1, 2); MPC_GJ4_ELIM(1, 3); MPC_GJ4_RHS(1);
        MPC_GJ4_PIVOT(2); MPC_GJ4_ELIM(2, 3); MPC_GJ4_RHS(2);
        MPC_GJ4_PIVOT(3); MPC_GJ4_RHS(3);
#undef MPC_GJ4_PIVOT
#undef MPC_GJ4_ELIM
#undef MPC_GJ4_RHS
        const float dxv = -(sol * invd);
#endif
        const float nrm2 = wv::ring_sum(dxv * dxv);
        // what this trip factorised is what the solve returns for every row (frozen rows recompute their own)
        f = fn;
        mv = mnv;
        const bool small = !(nrm2 >= 1e-8f) & (done == 0.f);        // |dx| < 1e-4  (:56-59)
        it_ret = small ? (float)it : it_ret;
        conv = small ? 1.f : conv;
        done = small ? 1.f : done;
        // :61-78  the step: x + dx when that stays inside the box (its Armijo ratio is exactly 1/2, see pnqp4), else the
        // projection of x + alpha dx with alpha = 1, 0.1, ... by the Armijo rule.  d = step taken (0 on a converged row).
        const float live = 1.f - done;
        const float xnv = xv + dxv;
        float xcv = eclampf(xnv, lbv, ubv);
        const float out = wv::ring_sum(fabsf(xcv - xnv));
        float dv = (xcv - xv) * live, hdv;
        MPC_HV(hdv, dv);
        const bool inside = out == 0.f;
        full = inside ? 1.f : 0.f;
        const bool test = !inside & (done == 0.f);
        if (wv::any(test)) {        // (round 6 A/B: computed in every trip instead -- one wave-uniform branch fewer --: 144.8 against 143.4 us)
            if (test) MPC_STAT(2);
            // f(x) - f(m) = -g'd - d'Hd/2 with d = m - x, against 0.1 g'(x - m)
            const float den = wv::ring_sum(-gv * dv), dhd = wv::ring_sum(dv * hdv);
            const float arm = fmaf(-0.5f, dhd, den) * wv::rcp(den);
            // (rare: one QP in a hundred) shorter steps.  Rows that do not need them ride along with their values
            // untouched: the branch stays wave-uniform (no exec-mask bookkeeping around the DPP broadcasts inside)
            bool go = test & (arm <= 0.1f);
            if (wv::any(go)) {
                float alpha = 0.1f;
                for (int count = 1; count < 10; ++count) {
                    if (go) MPC_STAT(3);
                    const float xt = eclampf(fmaf(alpha, dxv, xv), lbv, ubv);
                    const float dt = xt - xv;
                    float hdt;
                    MPC_HV(hdt, dt);
                    const float den2 = wv::ring_sum(-gv * dt), dhd2 = wv::ring_sum(dt * hdt);
                    const float arm2 = fmaf(-0.5f, dhd2, den2) * wv::rcp(den2);
                    xcv = go ? xt : xcv;
                    dv = go ? dt : dv;
                    hdv = go ? hdt : hdv;
                    go = go & (arm2 <= 0.1f);
                    alpha = go ? alpha * 0.1f : alpha;
                    if (!wv::any(go)) break;
                }
            }
        }
        xv = (done != 0.f) ? xv : xcv;                              // :78 (a converged row keeps its x)
        gv += hdv;
    }
#undef MPC_HV
#undef MPC_QV
    x[0] = wv::bcast<0>(xv); x[1] = wv::bcast<4>(xv); x[2] = wv::bcast<8>(xv); x[3] = wv::bcast<12>(xv);
    fr_out[0] = wv::bcast<0>(mv) != 0.f; fr_out[1] = wv::bcast<4>(mv) != 0.f;
    fr_out[2] = wv::bcast<8>(mv) != 0.f; fr_out[3] = wv::bcast<12>(mv) != 0.f;
    converged = conv != 0.f;
    return (int)it_ret;
}

enum {
    // NSTAGE: slots of the sweep's staging ring (the DMA runs NSTAGE - 1 timesteps ahead).  4 in the product; the diagnostic
    // build -DMPC_DPP16_NSTAGE=2 halves the wave's LDS so that a CU holds eight waves (two per SIMD) instead of four.
    SC = 0, SF = 4096, SR = 7168, SG = 8192, STAGE_BYTES = 9216, NSTAGE = MPC_DPP16_NSTAGE, AHEAD = NSTAGE - 1,
    R_c = 0, R_tau = 64, R_f = 128, R_qs = 176, R_lo = 192, R_hi = 208,
    LDS_TOTAL = NSTAGE * STAGE_BYTES,
#if defined(MPC_PAD_PROBE_DUP) && MPC_DPP16_PADK
    DMA_SWEEP = 48
#else
    DMA_SWEEP = MPC_DPP16_PADK ? 32 : 8       // 4 C + 3 F + 1 record (padded instantiation, dwords: 16 + 12 + 4)
#endif
};
// DMA instructions of one rollout stage: F, record, gains (+ C when priced directly, + (m, M) otherwise
// when constraints are present)
#if defined(MPC_PAD_PROBE_DUP) && MPC_DPP16_PADK
#define MPC_PAD_NF 12
#define MPC_PAD_NC 32
#else
#define MPC_PAD_NF 12
#define MPC_PAD_NC 16
#endif
template <int MODE, bool DIRECT> struct RollDma { enum { N = (MPC_DPP16_PADK ? MPC_PAD_NF + 4 : 3 + 1) + (rgm(MODE) ? 0 : 1) + (DIRECT ? (MPC_DPP16_PADK ? MPC_PAD_NC : 4) : (con(MODE) ? 1 : 0)) }; };
// The identity-priced rollout does not stage C: its stage is [(m, M)] | gains | F | record = 5 (6) KiB, packed
// back to back so the same LDS holds 7 (6) stages instead of 4 and the DMA runs 6 (5) timesteps ahead -- a
// rollout step is ~0.45 us, four-deep staging would leave the loads less than an HBM round trip under load.
// Every stage is addressed from its F block (`mid`): the sweep's C sits at mid-4096, the record at mid+3072,
// the packed rollout's gains at mid-1024 and (m, M) at mid-2048; a rollout that prices directly keeps the
// sweep's 9 KiB layout with the gains behind the record.
template <int MODE, bool DIRECT> struct RollRing {
    enum {
        // (register-resident gains: the stage is F | record = 4 KiB, nine stages in flight)
        BYTES = DIRECT ? (int)STAGE_BYTES : (con(MODE) ? 6144 : (rgm(MODE) ? 4096 : 5120)),
#ifdef MPC_DPP16_RSLOTS
        SLOTS = DIRECT ? (int)NSTAGE : (MPC_DPP16_RSLOTS),
#else
        SLOTS = DIRECT ? (int)NSTAGE : (int)LDS_TOTAL / (con(MODE) ? 6144 : (rgm(MODE) ? 4096 : 5120)),
#endif
        FOFF = DIRECT ? (int)SF : (con(MODE) ? 2048 : (rgm(MODE) ? 0 : 1024)),        // F block inside a slot
        GADJ = DIRECT ? 0 : (int)SF - 1024 - (int)SG,                // gains / (m, M) relative to the lane offsets,
        MADJ = DIRECT ? 0 : (int)SF - 2048 - (int)SC                 // which are written for the sweep's layout
    };
};
// anchor of ring slot `slot`, and the base the lane offsets (SC / SF / SR / SG relative) apply to
template <int MODE, bool ROLL, bool DIRECT> MPC_DEV unsigned stage_mid(int slot)
{
    return ROLL ? (unsigned)(slot * (int)RollRing<MODE, DIRECT>::BYTES + (int)RollRing<MODE, DIRECT>::FOFF)
                : (unsigned)(slot * (int)STAGE_BYTES + (int)SF);
}

// every pass of the padded instantiation starts on cleared staging memory: the record words of padded entries are never written by
// the gathers (their lanes sit the instruction out), and the passes lay their slots out differently
MPC_DEV void pad_clear(int lane, unsigned bytes = (unsigned)LDS_TOTAL)
{
    if (!PADK) return;
    wv::lds_sync();
    for (unsigned off = 16u * (unsigned)lane; off + 16 <= bytes; off += 1024) wv::lds_store_f32x4(off, f32x4{0.f, 0.f, 0.f, 0.f});
    wv::lds_sync();
}

// ---- LDS bank conflicts ----------------------------------------------------------------------------------
// Row-major 64-byte rows read as one b128 per lane (lane j = row j) put eight lanes on the same four banks, and
// the four problems' F blocks (768 B apart) alias bank for bank under column reads.  Both go away by permuting
// 16-byte granules, which the DMA does for free:
//   rows read by b128 (C always, F in the rollout): quarter q of row j is stored at quarter q ^ s(j), s(j) = (j>>1)&3
//   F read by columns (sweep, KKT):                 odd problems store row m at row m ^ 1 (the other 16 banks)
//   C also read by columns (the symmetry test):     problem slot k stores row R at row R ^ k -- a row is 16 banks, so the
//                                                   four problems' copies of row i, 1 KiB apart, would meet bank for bank;
//                                                   with rows (i ^ k) & 3 = 0..3 a column read touches all 64 banks once
MPC_DEV int quarter_swizzle(int row) { return (row >> 1) & 3; }
// source granule (of the 64 of a 16x16 block / the 48 of a 12x16 block) for LDS granule position g
MPC_DEV int src_granule_rows(int g) { return (g & ~3) | ((g & 3) ^ quarter_swizzle(g >> 2)); }
MPC_DEV int src_granule_cols(int g, int problem_slot) { return g ^ ((problem_slot & 1) << 2); }
// C: position (row r', quarter q') of problem slot k holds source row R = r' ^ k, quarter q' ^ s(R)
MPC_DEV int src_granule_C(int g, int problem_slot)
{
#ifdef MPC_DPP16_NO_CPERM          // diagnostic build (tools/ab_sym.py): rows where they were, column reads 4-way conflicted
    problem_slot = 0;
#endif
    const int R = (g >> 2) ^ (problem_slot & 3);
    return 4 * R + ((g & 3) ^ quarter_swizzle(R));
}

struct Lane {
    int lane, p, j;       // problem slot in the wave, variable
    int pb;               // problem index (clamped to B-1)
    int b0;               // first problem of the wave (uniform)
    bool live;            // pb is a real problem of this wave
    bool isu;             // j >= 12
    bool ovalid;          // this lane's variable exists in the caller's tau (always, in the exact kernel)
    float padd[4];        // padded instantiation: 1 in lane 12 + a of entry a for a control beyond n_ctrl -- the identity on the padded diagonal of Quu
    int a;                // control index of this lane (j - 12), 0 for state lanes
    // LDS byte offsets inside a stage
    // C and F sit in LDS in granule-permuted order (see lds_swizzle below): a lane's 16-byte DMA destination is
    // fixed, its source is free, so the permutation costs nothing and the reads below are bank-conflict free
    int aCq[4];           // SC + p*1024 + 64 (j ^ p) + 16 (q ^ s(j))  quarter q of row j of C      (b128)
    int aCcol;            // C[i][j], the column through this lane's variable: (aCcol ^ ((i&3) << 6 | s(i) << 4)) + 64 (i & 12)
    int aFe, aFo;         // SF + p*768 + 4 j (+-64 for odd p)   F[m][j] at aFe + 64 m (m even) / aFo + 64 m (m odd)
    int aFq[4];           // SF + p*768 + 64 j' + 16 (q ^ s(j')) quarter q of row j' = min(j, 11) of F (b128, rollout)
    int aRec;             // SR + p*256 + 4 j                   (+R_c: c_j, +R_tau: tau_j)
    int aRecA;            // SR + p*256 + 4 a                   (+R_lo / R_hi)
    int aRecF;            // SR + p*256 + R_f + 4 min(j, 11)
    int aKrow;            // SG + p*256 + 4 a                   (+16 jj: K[a][jj]; +192: k_a)
    int aS[4];            // SG + p*256 + 16 (12 + max(a,b)) + 4 min(a,b)   (Quu[a][b], see lane_init)
    float *out0;          // this lane's element of new_x / new_u at t = 0 ...
    long ostep;           // ... and its stride per timestep
    float *scr0;          // this lane's element of the second trial's trajectory [T,B,16] (workspace) at t = 0
    long ostep1;          // = 16 B
    int aMcol;            // SC + p*256 + 16 j                  (the m block of the stage, this lane's own granule: m for j = 12; + RollRing::MADJ)
    int aGcol;            // SG + p*256 + 16 j                  (gain record, this lane's own granule: column j of K / M)
    int aFm;              // SG + p*256 + 16*13 + 12            (the free-set flags of the row's problem, see sweep_step)
};

// cperm: the C blocks of this launch are staged with their rows permuted per problem slot (src_granule_C): only the
// symmetry test reads columns, so a launch whose caller vouches for C keeps the plain layout (0.5 us of the 83)
MPC_DEV void lane_init(Lane &L, int lane, int wave, int B, bool cperm = true)
{
    L.lane = lane;
    L.p = lane >> 4;
    L.j = lane & 15;
#ifdef MPC_DPP16_NO_CPERM
    const int cp = 0;
#else
    const int cp = cperm ? L.p : 0;          // row permutation of this problem slot's C block (src_granule_C)
#endif
    const int pb = 4 * wave + L.p;
    L.b0 = 4 * wave;
    L.live = pb < B;
    L.pb = L.live ? pb : B - 1;
    L.isu = L.j >= 12;
    L.ovalid = true;
    L.padd[0] = L.padd[1] = L.padd[2] = L.padd[3] = 0.f;
    L.a = L.isu ? L.j - 12 : 0;
    const int jx = L.j < 12 ? L.j : 11;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        L.aCq[q] = SC + L.p * 1024 + 64 * (L.j ^ cp) + 16 * (q ^ quarter_swizzle(L.j));
        L.aFq[q] = SF + L.p * 768 + 64 * jx + 16 * (q ^ quarter_swizzle(jx));
    }
    // source row i sits at row i ^ p = (i & 12) | ((i & 3) ^ p), quarter (j >> 2) ^ s(i): both are XORs into bits 4..7
    L.aCcol = SC + L.p * 1024 + (cp << 6) + ((L.j >> 2) << 4) + 4 * (L.j & 3);
    L.aFe = SF + L.p * 768 + 4 * L.j + ((L.p & 1) ? 64 : 0);
    L.aFo = SF + L.p * 768 + 4 * L.j - ((L.p & 1) ? 64 : 0);
    L.aRec = SR + L.p * 256 + 4 * L.j;
    L.aRecA = SR + L.p * 256 + 4 * L.a;
    L.aRecF = SR + L.p * 256 + R_f + 4 * jx;
    L.aKrow = SG + L.p * 256 + 4 * L.a;
    L.aMcol = SC + L.p * 256 + 16 * L.j;
    L.aGcol = SG + L.p * 256 + 16 * L.j;
    L.aFm = SG + L.p * 256 + 16 * 13 + 12;
    // Quu in the gain record: lanes 13..15 carry columns 1..3 of Quu as they stand in their Q registers
    // (element r of lane 12 + c = Quu[r][c]); Quu[0][0], which only lane 12 (the k lane) holds, replaces the
    // redundant Quu[2][1] in lane 13.  Quu[a][b] is read as (column max(a,b), row min(a,b)).
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int lo = L.a < b ? L.a : b, hi = L.a < b ? b : L.a;
        L.aS[b] = SG + L.p * 256 + (hi == 0 ? 16 * 13 + 4 * 2 : 16 * (12 + hi) + 4 * lo);
    }
}

// ---------------------------------------------------------------------------
// HBM -> LDS staging: lane l of DMA instruction k moves 16 B.
//   C  : instruction k = problem slot k, granule l            (4 instructions)
//   F  : granule G = 64 k + l of the 4 x 48 granules           (3 instructions)
//   rec: problem slot l>>4, granule l&15: 0-3 c | 4-6 x | 7 u | 8-10 f | 11 the QP's start (sweep, mode 2, when given) | 12 lo | 13 hi
//   gains (rollout): problem slot l>>4, granule l&15 of the wave's own record Kk[t][b][16][4]
// Every lane keeps RUNNING source pointers that step along the horizon (one 64-bit add per pointer and timestep;
// t * stride + base per DMA cost three times that).  Every lane takes part in every instruction: a granule nobody
// reads in this pass (c in the identity-priced rollout, f in the sweep, the unused ones) is aliased to a granule of
// the nominal state -- the same cache lines its neighbours fetch, no extra HBM traffic, and no exec-mask
// bookkeeping around the DMA.  Rows of a partial last wave repeat problem B-1 (loads and stores: the same values
// to the same addresses).
// ---------------------------------------------------------------------------
struct Dma {
    // all biased by minus the immediate their instruction carries (see wv::dma16_at)
    const char *c_ptr[4];     // per lane                             imm 1024 k - 4096
    const char *f_ptr[3];     // per lane (column-read order in the sweep, row-read order in the rollout)  imm 1024 k
    const char *r_ptr;        // per lane                             imm 3072
    const char *g_ptr;        // per lane                             imm -1024 (packed rollout)
    const char *g2_ptr;       // per lane: the (m, M) record          imm -2048 (packed rollout)
    long c_step, f_step, g_step;      // bytes per timestep (wave-uniform)
    long g2_step;                     // ... of the m record (16 bytes per problem and timestep)
    long r_step;                      // bytes per timestep of this lane's record source
    long r_step_nof;                  // the same, but 0 on lanes that stream f (a move that leaves F / f in place)
#ifdef MPC_DPP16_PAD
    // the padded instantiation's gathers: wave-uniform block bases of the four problems at the stage the pointers stand on, the
    // per-lane source offsets of the 16 + 12 dword instructions of a C / F stage, and this lane's word of each problem's record
    const char *Cb[4], *Fb[4];
    unsigned coff[16], foff[12];
    unsigned cbytes, fbytes;
    const char *rq[4];
    long rq_step[4];
    bool rq_act[4], rq_isf[4];
#endif
};

// Position the pointers on the first stage of a pass: t = T-1 for the sweep, t = 0 for a rollout.
// F / f have T-1 entries: stage t reads entry min(t, T-2) (stage T-1 never looks at its copy).
template <int MODE, bool ROLL, bool DIRECT>
MPC_DEV void dma_seek(Dma &d, const P &p, const Lane &L, int wave)
{
    const long B = p.B;
    const int T = p.T;
    const long t0 = ROLL ? 0 : T - 1;
    const long tf0 = (ROLL || T < 2) ? 0 : T - 2;
    d.c_step = 4 * p.C_st;
    d.f_step = T > 1 ? 4 * p.F_st : 0;
#ifdef MPC_DPP16_PROBE_ROLL_F          // (diagnostic build only: the rollouts read timestep 0's F at every timestep -- WRONG results, the time of a
    if (ROLL) d.f_step = 0;            //  step whose rollout costs no HBM traffic for F: the bound of every scheme that keeps F on chip)
#endif
    d.g_step = 4 * B * 64;
    d.g2_step = 4 * B * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pbk = 4 * wave + k < p.B ? 4 * wave + k : p.B - 1;
        d.c_ptr[k] = (const char *)(p.C + (long)pbk * p.C_sb) + 16 * src_granule_C(L.lane, p.c_symmetric ? 0 : k) - (1024 * k - 4096) + t0 * d.c_step;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int G = 64 * k + L.lane;
        const int slot = G / 48, gi = G - 48 * slot;
        const int pbk = 4 * wave + slot < p.B ? 4 * wave + slot : p.B - 1;
        const char *fb = T > 1 ? (const char *)(p.F + (long)pbk * p.F_sb) : (const char *)p.C;
        const int perm = ROLL ? src_granule_rows(gi) : src_granule_cols(gi, slot);
        d.f_ptr[k] = fb + (T > 1 ? 16 * perm : 0) - 1024 * k + tf0 * d.f_step;
    }
    {
        const int gi = L.lane & 15;
        const long pb = L.pb;
        const bool want_c = !ROLL || DIRECT;            // the identity-priced rollout never looks at c
        const bool want_f = ROLL && p.f && T > 1;       // the sweep never looks at f
        const bool want_b = MODE == 2 && p.bound_mode == MPC_BOUND_TENSOR;
        const char *q = (const char *)(p.cur_x + pb * 12 + 4 * (gi % 3));       // the alias: a granule of the nominal state
        long st = 4 * B * 12;
        bool is_f = false;
        if (gi < 4) {
            if (want_c) { q = (const char *)(p.c + pb * p.c_sb + 4 * gi); st = 4 * p.c_st; }
        } else if (gi < 7) {
            q = (const char *)(p.cur_x + pb * 12 + 4 * (gi - 4));
        } else if (gi == 7) {
            q = (const char *)(p.cur_u + pb * 4); st = 4 * B * 4;
        } else if (gi < 11) {
            if (want_f) { q = (const char *)(p.f + pb * p.f_sb + 4 * (gi - 8)); st = 4 * p.f_st; is_f = true; }
        } else if (gi == 11) {
            // (mpc_lqr_options.qp_start: strides may be 0 -- one block for the whole batch / horizon -- and the array may be
            // the record this very sweep rewrites: stage t is fetched AHEAD timesteps before timestep t stores its own)
            if (MPC_QP_START && MODE == 2 && !ROLL && p.qp_start) { q = (const char *)(p.qp_start + pb * p.qp_start_sb); st = 4 * p.qp_start_st; }
        } else if (gi == 12 || gi == 13) {
            if (want_b) { q = (const char *)((gi == 12 ? p.lo : p.hi) + pb * 4); st = 4 * B * 4; }
        }
        d.r_ptr = q - 3072 + (is_f ? tf0 : t0) * st;
        d.r_step = st;
        d.r_step_nof = is_f ? 0 : st;
        d.g_ptr = (const char *)(p.Kk + pb * 64 + 4 * gi) + 1024 + t0 * d.g_step;
        // (round 5) the second record is m = qu + Quu k alone, 16 bytes per problem and timestep: every lane of a row fetches its
        // problem's block (one request), so lane 12 finds m at its own granule of the stage's record block (M rides in the gain
        // record, see sweep_step)
        d.g2_ptr = (const char *)(p.Kk + (long)T * B * 64 + pb * 4) + 2048 + t0 * d.g2_step;
    }
#ifdef MPC_DPP16_PAD
    {
        // ---- the padded instantiation's maps (see the head of the file).  LDS position -> source element, through the SAME granule
        // permutations the exact kernel's 16-byte DMA applies (src_granule_C / _cols / _rows), one dword at a time.
        const int ns = p.ns, nc = p.nc, n = ns + nc;
        d.cbytes = (unsigned)(n * n * 4) + (unsigned)PAD_BIAS;
        d.fbytes = (T > 1 ? (unsigned)(ns * n * 4) : 0u) + (unsigned)PAD_BIAS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pbk = 4 * wave + k < p.B ? 4 * wave + k : p.B - 1;
            d.Cb[k] = (const char *)(p.C + (long)pbk * p.C_sb) + t0 * d.c_step - (int)PAD_BIAS;
            d.Fb[k] = (T > 1 ? (const char *)(p.F + (long)pbk * p.F_sb) + tf0 * d.f_step : (const char *)p.C) - (int)PAD_BIAS;
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int slot = kk >> 2, dw = 64 * (kk & 3) + L.lane, g = dw >> 2, e = dw & 3;
            const int gs = src_granule_C(g, p.c_symmetric ? 0 : slot), R = gs >> 2, col = 4 * (gs & 3) + e;
            const int aR = pad_tau(R, ns, nc), ac = pad_tau(col, ns, nc);
            // (stored biased: - the instruction's immediate, 256 kk from the stage's C block, + PAD_BIAS)
            d.coff[kk] = ((aR >= 0 && ac >= 0) ? (unsigned)(4 * (aR * n + ac)) : OOB_OFF) + (unsigned)PAD_BIAS - 256u * (unsigned)kk;
        }
#pragma unroll
        for (int kk = 0; kk < 12; ++kk) {
            const int slot = kk / 3, G = 16 * kk + (L.lane >> 2), gi = G - 48 * slot, e = L.lane & 3;
            const int perm = ROLL ? src_granule_rows(gi) : src_granule_cols(gi, slot);
            const int m = perm >> 2, ac = pad_tau(4 * (perm & 3) + e, ns, nc);
            d.foff[kk] = ((m < ns && ac >= 0) ? (unsigned)(4 * (m * n + ac)) : OOB_OFF) + (unsigned)PAD_BIAS - 256u * (unsigned)kk;      // (immediate: 256 kk from the F block)
        }
        // the record of problem slot kk: lane l = word l of its 64: granule gi = l >> 2 (0-3 c | 4-6 x | 7 u | 8-10 f | 11 the QP's
        // start | 12 lo | 13 hi), entry e = l & 3 of it.  A word with no source sits the instruction out: it keeps the zero that
        // pad_clear left there.
        const int gi = L.lane >> 2, e = L.lane & 3;
        const bool want_c = !ROLL || DIRECT, want_f = ROLL && p.f && T > 1;
        const bool want_b = MODE == 2 && p.bound_mode == MPC_BOUND_TENSOR;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long pbk = 4 * wave + kk < p.B ? 4 * wave + kk : p.B - 1;
            const char *q = (const char *)p.cur_x;
            long st = 0;
            bool act = false, isf = false;
            if (gi < 4) {
                const int a = pad_tau(4 * gi + e, ns, nc);
                if (want_c && a >= 0) { q = (const char *)(p.c + pbk * p.c_sb + a); st = 4 * p.c_st; act = true; }
            } else if (gi < 7) {
                const int i = 4 * (gi - 4) + e;
                if (i < ns) { q = (const char *)(p.cur_x + pbk * ns + i); st = 4 * B * ns; act = true; }
            } else if (gi == 7) {
                if (e < nc) { q = (const char *)(p.cur_u + pbk * nc + e); st = 4 * B * nc; act = true; }
            } else if (gi < 11) {
                const int i = 4 * (gi - 8) + e;
                if (want_f && i < ns) { q = (const char *)(p.f + pbk * p.f_sb + i); st = 4 * p.f_st; act = true; isf = true; }
            } else if (gi == 11) {
                if (MPC_QP_START && MODE == 2 && !ROLL && p.qp_start && e < nc) { q = (const char *)(p.qp_start + pbk * p.qp_start_sb + e); st = 4 * p.qp_start_st; act = true; }
            } else if (gi == 12 || gi == 13) {
                if (want_b && e < nc) { q = (const char *)((gi == 12 ? p.lo : p.hi) + pbk * nc + e); st = 4 * B * nc; act = true; }
            }
            d.rq[kk] = q + (isf ? tf0 : t0) * st - (3072 + 256 * kk);       // (biased by minus the immediate: the record sits 3072 behind the F block)
            d.rq_step[kk] = st;
            d.rq_act[kk] = act;
            d.rq_isf[kk] = isf;
        }
    }
#endif
}

#ifdef MPC_DPP16_PAD
// the gathers of one stage, in the four parts the arithmetic of a timestep takes them in (Feed): `mid` = the stage's F block
template <int MODE, bool ROLL, bool DIRECT, int K> MPC_DEV void pad_part(const Dma &d, unsigned mid)
{
    constexpr bool WITH_C = !ROLL || DIRECT;
    // sweep / direct rollout: C 0-7 | C 8-15 | F 0-7 | F 8-11 + record;   packed rollout: F 0-5 | F 6-11 | record | -
    // (ONE LDS anchor per block -- the stage's C block, its F block -- and the gather's place as the instruction's immediate)
#if defined(MPC_PAD_PROBE_DUP) && MPC_DPP16_PADK          // (diagnostic build only: every gather of C issued twice -- the same results, the time of a stage of 48 instead of 32 instructions)
#define MPC_PAD_C(kk) do { wv::dma_buf_at<256 * (kk), PAD_BIAS>(d.Cb[(kk) >> 2], d.cbytes, d.coff[kk], mid - 4096); \
                           wv::dma_buf_at<256 * (kk), PAD_BIAS>(d.Cb[(kk) >> 2], d.cbytes, d.coff[kk], mid - 4096); } while (0)
#define MPC_PAD_F(kk) wv::dma_buf_at<256 * (kk), PAD_BIAS>(d.Fb[(kk) / 3], d.fbytes, d.foff[kk], mid)
#else
#define MPC_PAD_C(kk) wv::dma_buf_at<256 * (kk), PAD_BIAS>(d.Cb[(kk) >> 2], d.cbytes, d.coff[kk], mid - 4096)
#define MPC_PAD_F(kk) wv::dma_buf_at<256 * (kk), PAD_BIAS>(d.Fb[(kk) / 3], d.fbytes, d.foff[kk], mid)
#endif
#define MPC_PAD_R(kk) wv::dma4_at_if<3072 + 256 * (kk)>(d.rq_act[kk], d.rq[kk], mid)
    if (WITH_C) {
        if (K == 0) { MPC_PAD_C(0); MPC_PAD_C(1); MPC_PAD_C(2); MPC_PAD_C(3); MPC_PAD_C(4); MPC_PAD_C(5); MPC_PAD_C(6); MPC_PAD_C(7); }
        else if (K == 1) { MPC_PAD_C(8); MPC_PAD_C(9); MPC_PAD_C(10); MPC_PAD_C(11); MPC_PAD_C(12); MPC_PAD_C(13); MPC_PAD_C(14); MPC_PAD_C(15); }
        else if (K == 2) { MPC_PAD_F(0); MPC_PAD_F(1); MPC_PAD_F(2); MPC_PAD_F(3); MPC_PAD_F(4); MPC_PAD_F(5); MPC_PAD_F(6); MPC_PAD_F(7); }
        else { MPC_PAD_F(8); MPC_PAD_F(9); MPC_PAD_F(10); MPC_PAD_F(11); MPC_PAD_R(0); MPC_PAD_R(1); MPC_PAD_R(2); MPC_PAD_R(3); }
    } else {
        if (K == 0) { MPC_PAD_F(0); MPC_PAD_F(1); MPC_PAD_F(2); MPC_PAD_F(3); MPC_PAD_F(4); MPC_PAD_F(5); }
        else if (K == 1) { MPC_PAD_F(6); MPC_PAD_F(7); MPC_PAD_F(8); MPC_PAD_F(9); MPC_PAD_F(10); MPC_PAD_F(11); }
        else if (K == 2) { MPC_PAD_R(0); MPC_PAD_R(1); MPC_PAD_R(2); MPC_PAD_R(3); }
    }
#undef MPC_PAD_C
#undef MPC_PAD_F
#undef MPC_PAD_R
}
#endif

// DMA of the stage the pointers stand on into the ring slot anchored at `mid`: exactly DMA_SWEEP /
// RollDma<MODE, DIRECT>::N instructions, all but the directly-priced rollout's gains on one M0.
template <int MODE, bool ROLL, bool DIRECT>
MPC_DEV void stage_issue(const Dma &d, unsigned mid)
{
#ifdef MPC_DPP16_PAD
    pad_part<MODE, ROLL, DIRECT, 0>(d, mid); pad_part<MODE, ROLL, DIRECT, 1>(d, mid);
    pad_part<MODE, ROLL, DIRECT, 2>(d, mid); pad_part<MODE, ROLL, DIRECT, 3>(d, mid);
    if (ROLL && !rgm(MODE)) {
        if (DIRECT) wv::dma16_at<0>(d.g_ptr - 1024, mid + (SG - SF));
        else wv::dma16_at<-1024>(d.g_ptr, mid);
    }
    if (ROLL && !DIRECT && con(MODE)) wv::dma16_at<-2048>(d.g2_ptr, mid);
    return;
#endif
    if (!ROLL || DIRECT) {
        wv::dma16_at<-4096, wv::DMA_C>(d.c_ptr[0], mid);
        wv::dma16_at<-3072, wv::DMA_C>(d.c_ptr[1], mid);
        wv::dma16_at<-2048, wv::DMA_C>(d.c_ptr[2], mid);
        wv::dma16_at<-1024, wv::DMA_C>(d.c_ptr[3], mid);
    }
    wv::dma16_at<0, ROLL ? wv::DMA_LAST : wv::DMA_PLAIN>(d.f_ptr[0], mid);
    wv::dma16_at<1024, ROLL ? wv::DMA_LAST : wv::DMA_PLAIN>(d.f_ptr[1], mid);
    wv::dma16_at<2048, ROLL ? wv::DMA_LAST : wv::DMA_PLAIN>(d.f_ptr[2], mid);
    wv::dma16_at<3072>(d.r_ptr, mid);
    if (ROLL && !rgm(MODE)) {
        if (DIRECT) {
            wv::dma16_at<0>(d.g_ptr - 1024, mid + (SG - SF));
        } else {
            wv::dma16_at<-1024>(d.g_ptr, mid);
            if (con(MODE)) wv::dma16_at<-2048>(d.g2_ptr, mid);
        }
    }
}

// Step the pointers to the next stage of the pass (t-1 in the sweep, t+1 in a rollout).  move_f (wave-uniform):
// the F / f index changes with it (it does not between stages T-1 and T-2).
template <int MODE, bool ROLL, bool DIRECT>
MPC_DEV void stage_move(Dma &d, bool move_f)
{
#ifdef MPC_DPP16_PAD
    {
        const long fsp = move_f ? d.f_step : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ROLL || DIRECT) d.Cb[k] += ROLL ? d.c_step : -d.c_step;
            d.Fb[k] += ROLL ? fsp : -fsp;
            const long rs = (d.rq_isf[k] && !move_f) ? 0 : d.rq_step[k];
            d.rq[k] += ROLL ? rs : -rs;
        }
        if (ROLL && !rgm(MODE)) {
            d.g_ptr += d.g_step;
            if (con(MODE) && !DIRECT) d.g2_ptr += d.g2_step;
        }
        return;
    }
#endif
    const long fs = move_f ? d.f_step : 0;
    const long rs = ROLL ? (move_f ? d.r_step : d.r_step_nof) : d.r_step;
    if (!ROLL || DIRECT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d.c_ptr[k] += ROLL ? d.c_step : -d.c_step;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) d.f_ptr[k] += ROLL ? fs : -fs;
    d.r_ptr += ROLL ? rs : -rs;
    if (ROLL && !rgm(MODE)) {
        d.g_ptr += d.g_step;
        if (con(MODE) && !DIRECT) d.g2_ptr += d.g2_step;
    }
}

// counted wait in the tail of a pass: exactly `rem` (<= K) stages of ND instructions each are newer than the one needed
template <int K, int ND> MPC_DEV void tail_wait(int rem)
{
    if (K == 0 || rem >= K) wv::dma_wait<K * ND>();
    else tail_wait<(K > 0 ? K - 1 : 0), ND>(rem);
}

// The DMA of the stage LOOKAHEAD timesteps away, handed to the arithmetic of the current timestep in four parts of
// two instructions each: issued in one burst of eight, the instructions queue up behind the CU's one texture-address
// unit (shared by the four waves) and each stalls the wave ~27 clocks; spread between the blocks of arithmetic they
// cost their issue slot.  `on` is wave-uniform (no stage left to fetch in the last timesteps of a pass).
template <int MODE, bool ROLL, bool DIRECT> struct Feed {
    Dma &d;
    unsigned mid;
    bool on;
    bool move_f;
    template <int K> MPC_DEVM void part()
    {
        if (!on) return;
#ifdef MPC_DPP16_PAD
        pad_part<MODE, ROLL, DIRECT, K>(d, mid);
        if (ROLL && !DIRECT) {
            if (K == 2) {
                if (!rgm(MODE)) wv::dma16_at<-1024>(d.g_ptr, mid);
                if (con(MODE)) wv::dma16_at<-2048>(d.g2_ptr, mid);
            } else if (K == 3) {
                stage_move<MODE, ROLL, DIRECT>(d, move_f);
            }
        } else if (K == 3) {
            if (ROLL && !rgm(MODE)) wv::dma16_at<0>(d.g_ptr - 1024, mid + (SG - SF));
            stage_move<MODE, ROLL, DIRECT>(d, move_f);
        }
        return;
#endif
        enum { NC = (!ROLL || DIRECT) ? 4 : 0 };      // C instructions of a stage
        if (ROLL && !DIRECT) {
            // packed rollout stage: F0 F1 | F2 REC | G [G2] | moves
            if (K == 0) {
                wv::dma16_at<0, wv::DMA_LAST>(d.f_ptr[0], mid);
                wv::dma16_at<1024, wv::DMA_LAST>(d.f_ptr[1], mid);
            } else if (K == 1) {
                wv::dma16_at<2048, wv::DMA_LAST>(d.f_ptr[2], mid);
                wv::dma16_at<3072>(d.r_ptr, mid);
            } else if (K == 2) {
                if (!rgm(MODE)) wv::dma16_at<-1024>(d.g_ptr, mid);
                if (con(MODE)) wv::dma16_at<-2048>(d.g2_ptr, mid);
            } else {
                stage_move<MODE, ROLL, DIRECT>(d, move_f);
            }
        } else {
            if (K == 0) {
                wv::dma16_at<-4096, wv::DMA_C>(d.c_ptr[0], mid);
                wv::dma16_at<-3072, wv::DMA_C>(d.c_ptr[1], mid);
            } else if (K == 1) {
                wv::dma16_at<-2048, wv::DMA_C>(d.c_ptr[2], mid);
                wv::dma16_at<-1024, wv::DMA_C>(d.c_ptr[3], mid);
            } else if (K == 2) {
                wv::dma16_at<0, ROLL ? wv::DMA_LAST : wv::DMA_PLAIN>(d.f_ptr[0], mid);
                wv::dma16_at<1024, ROLL ? wv::DMA_LAST : wv::DMA_PLAIN>(d.f_ptr[1], mid);
            } else {
                wv::dma16_at<2048, ROLL ? wv::DMA_LAST : wv::DMA_PLAIN>(d.f_ptr[2], mid);
                wv::dma16_at<3072>(d.r_ptr, mid);
                if (ROLL && !rgm(MODE)) wv::dma16_at<0>(d.g_ptr - 1024, mid + (SG - SF));
                stage_move<MODE, ROLL, DIRECT>(d, move_f);
            }
        }
    }
};

struct ZmRaw { unsigned m[4]; };        // the masks of the wave's four problems (uniform)
MPC_DEV ZmRaw zm_fetch(const P &p, const Lane &L, int t)
{
    // u_zero_I [T,B,4] bytes: the four flags of this row's problem as one dword.  The four dwords of a wave come
    // through the scalar path (wave-uniform addresses): a vector load here would sit in the same vmcnt queue as the
    // stage DMAs, and the compiler -- which cannot count those across the loop -- would drain the whole queue in
    // front of every use (measured: the masked kernel at 152 us against 104 us unmasked).
    // Split in two so that a timestep of arithmetic sits between the loads and the first look at their result.
    const int last = p.B - 1 - L.b0;                         // a partial last wave repeats its last problem
    ZmRaw r;
    if (PADK) {
        // u_zero_I [T,B,nc] bytes at any nc: the byte of control a out of the aligned dword that holds it (scalar loads want 4-byte
        // alignment); a padded control is free (its row of Quu is the identity, it stays at zero).  (The aligned dword of the array's
        // last bytes can reach up to three bytes past its end -- inside the same 4-byte word, hence the same page: the one finding of the
        // emulator under AddressSanitizer, as for the padded 32/8 kernel's zero_mask_word.)
        const int nc = p.nc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long b = L.b0 + (k < last ? k : last);
            unsigned z = 0u;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const unsigned long A = (unsigned long)(p.zero_mask + ((long)t * p.B + b) * nc + (a < nc ? a : 0));
                const unsigned word = wv::load_uniform_u32((const unsigned *)(A & ~3ul));
                const unsigned byte = (word >> (8u * (unsigned)(A & 3ul))) & 0xffu;
                z |= (a < nc && byte != 0u) ? (1u << (8 * a)) : 0u;
            }
            r.m[k] = z;
        }
        return r;
    }
    const unsigned *row = (const unsigned *)p.zero_mask + (long)t * p.B + L.b0;
    r.m[0] = wv::load_uniform_u32(row);
    r.m[1] = wv::load_uniform_u32(row + (1 < last ? 1 : last));
    r.m[2] = wv::load_uniform_u32(row + (2 < last ? 2 : last));
    r.m[3] = wv::load_uniform_u32(row + (3 < last ? 3 : last));
    return r;
}
MPC_DEV unsigned zm_pick(const Lane &L, const ZmRaw &r)
{
    return L.p == 0 ? r.m[0] : (L.p == 1 ? r.m[1] : (L.p == 2 ? r.m[2] : r.m[3]));
}

// ---------------------------------------------------------------------------
// Sweep
// ---------------------------------------------------------------------------
// acc[4I .. 4I+3] (lane j) += a[lane 4I+v] * b[lane j]: rows 4I..4I+3 of the outer product a b'
template <int I, int N>
MPC_DEV void outer_rows(float (&acc)[N], float a, float b)
{
    f32x4 c = {acc[4 * I], acc[4 * I + 1], acc[4 * I + 2], acc[4 * I + 3]};
    c = wv::mfma4<I>(a, b, c);
    acc[4 * I] = c[0]; acc[4 * I + 1] = c[1]; acc[4 * I + 2] = c[2]; acc[4 * I + 3] = c[3];
}
// acc (12 or 16 rows, lane j = column j) += a b'
template <int N>
MPC_DEV void outer_acc(float (&acc)[N], float a, float b)
{
    outer_rows<0>(acc, a, b);
    outer_rows<1>(acc, a, b);
    outer_rows<2>(acc, a, b);
    if (N == 16) outer_rows<(N == 16 ? 3 : 0)>(acc, a, b);
}

struct SwStage {
    float Cc[16];
    float Fc[12];
    float cj, tb;
    float lo[4], hi[4];     // bounds of the row's four controls (row-uniform)
    float qs[4];            // the caller's start of this timestep's QP (row-uniform; read only when p.qp_start)
    unsigned zm;
};

template <int MODE>
MPC_DEV void sw_read(SwStage &s, const P &p, const Lane &L, int t, int slot, unsigned zm, float &asym, float &cmax)
{
    const unsigned base = (unsigned)slot * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = wv::lds_f32x4(base + L.aCq[q]);
        s.Cc[4 * q] = v[0]; s.Cc[4 * q + 1] = v[1]; s.Cc[4 * q + 2] = v[2]; s.Cc[4 * q + 3] = v[3];
    }
    // Is C_t symmetric?  Everything below reads row j of C as its column j (the header); the reference does not
    // (mpc/lqr_step.py:68, 294).  Lane j fetches the true column j and keeps the largest difference, and the largest
    // entry as the scale; step_wave turns the two into MPC_ST_C_ASYMMETRIC.  Skipped when the caller vouches for C.
    if (!p.c_symmetric) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const float c0 = wv::lds_f32(base + (unsigned)((L.aCcol ^ (((i & 3) << 6) | (quarter_swizzle(i) << 4))) + 64 * (i & 12)));
            const float c1 = wv::lds_f32(base + (unsigned)((L.aCcol ^ ((((i + 1) & 3) << 6) | (quarter_swizzle(i + 1) << 4))) + 64 * ((i + 1) & 12)));
            wv::absmax3(asym, s.Cc[i] - c0, s.Cc[i + 1] - c1);
            wv::absmax3(cmax, s.Cc[i], s.Cc[i + 1]);
        }
    }
    // at t = T-1 the F slot holds a copy of F[T-2] (stage_issue clamps the index) and nothing looks at it
#pragma unroll
    for (int m = 0; m < 12; ++m) s.Fc[m] = wv::lds_f32(base + ((m & 1) ? L.aFo : L.aFe) + 64 * m);
    s.cj = wv::lds_f32(base + L.aRec + R_c);
    s.tb = wv::lds_f32(base + L.aRec + R_tau);
#pragma unroll
    for (int a = 0; a < 4; ++a) { s.lo[a] = 0.f; s.hi[a] = 0.f; }
    if (MODE == 2) {
        if (p.bound_mode == MPC_BOUND_TENSOR) {
            const f32x4 l = wv::lds_f32x4(base + SR + L.p * 256 + R_lo);
            const f32x4 h = wv::lds_f32x4(base + SR + L.p * 256 + R_hi);
#pragma unroll
            for (int a = 0; a < 4; ++a) { s.lo[a] = l[a]; s.hi[a] = h[a]; }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bool pad = PADK && a >= p.nc;          // (a padded control has no reach: it stays at zero)
                s.lo[a] = pad ? 0.f : p.lo_s;
                s.hi[a] = pad ? 0.f : p.hi_s;
            }
        }
        if (MPC_QP_START && p.qp_start) {
            const f32x4 z = wv::lds_f32x4(base + SR + L.p * 256 + R_qs);
#pragma unroll
            for (int a = 0; a < 4; ++a) s.qs[a] = z[a];
        }
    }
    s.zm = MODE == 1 ? zm : 0u;
}

struct SwState {
    float Vc[12];
    float vv;
    // Kept in double: the rollout is priced as J_nominal + w_0 + (small terms), and J_nominal + w_0 is
    // the difference of two sums that can each be 1e3 times the result.
    double oc;         // nominal-cost partial of this lane
    double w0;         // sum_t (0.5 k'Quu k + qu'k): the sweep's predicted cost change (row-uniform)
    float kprev[4];
    int warm;
    int qp_total;
    int status;
    float asym, cmax;  // max |C[j][i] - C[i][j]| and max |C[j][i]| over this lane's rows so far (the symmetry test)
    float *rec;        // this lane's 16 bytes of the gain record of the current timestep (steps back by rec_step)
    float *rec2;       // the 16 bytes of m = qu + Quu k of this row's problem at the current timestep (constrained modes)
    long rec_step, rec2_step;
};

template <int MODE>
MPC_DEV void sweep_step(const P &p, const Lane &L, const SwStage &s, SwState &st, int t, Feed<MODE, false, false> &feed, Gains<rgm(MODE)> &G PROF_ARG)
{
    const bool last = (t == p.T - 1);
    feed.template part<0>();
    // c_back = C tau + c (mpc/lqr_step.py:289-295) and the nominal stage cost (util.get_cost, :169)
    float cb = s.cj;
    wv::dot_bcast16(cb, s.tb, s.Cc);
    st.oc += (double)(s.tb * (0.5f * (cb + s.cj)));

    float Q[16];
    float q = cb;
#pragma unroll
    for (int i = 0; i < 16; ++i) Q[i] = s.Cc[i];
    if (!last) {
        // Y = V F, Q = C + F'Y, q = c_back + F'v   (:65-70)
        // as sums of outer products on the matrix core: Y = sum_m V[:,m] F[m,:]  (V symmetric: lane i holds
        // V[m][i] = V[i][m] in Vc[m]),  Q += sum_m F[m,:]' Y[m,:]
        float Y[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // folds into the first products' srcC
        wv::sched_fence();
#pragma unroll
        for (int m = 0; m < 12; ++m) outer_acc(Y, st.Vc[m], s.Fc[m]);
        wv::sched_fence();
        feed.template part<1>();
        wv::sched_fence();
#pragma unroll
        for (int m = 0; m < 12; ++m) outer_acc(Q, s.Fc[m], Y[m]);
        wv::sched_fence();          // keep the MFMAs in blocks: every MFMA <-> VALU turn costs ~7 clocks
        feed.template part<2>();
        wv::sched_fence();
        wv::dot_bcast12(q, st.vv, s.Fc);
    } else {
        feed.template part<1>();
        feed.template part<2>();
    }

    if (PADK) {
        // a control beyond n_ctrl: H = 1, q = 0, no reach -- it stays at zero and its gains are zero
#pragma unroll
        for (int a = 0; a < 4; ++a) Q[12 + a] += L.padd[a];
    }
    PROF_MARK(10);              // slot 10: c_back + the products Y, Q, q
    // ---- the 4x4 control block: row-uniform copies out of lanes 12..15 --------------------------
    Sym4 S;
    float qu[4];
#ifdef MPC_DPP16_DPP_BCAST          // (rounds 1-3: fourteen DPP row broadcasts, each a v_mov_b32_dpp behind its wait states)
    S.s00 = wv::bcast<12>(Q[12]); S.s01 = wv::bcast<13>(Q[12]); S.s02 = wv::bcast<14>(Q[12]); S.s03 = wv::bcast<15>(Q[12]);
    S.s11 = wv::bcast<13>(Q[13]); S.s12 = wv::bcast<14>(Q[13]); S.s13 = wv::bcast<15>(Q[13]);
    S.s22 = wv::bcast<14>(Q[14]); S.s23 = wv::bcast<15>(Q[14]);
    S.s33 = wv::bcast<15>(Q[15]);
    qu[0] = wv::bcast<12>(q); qu[1] = wv::bcast<13>(q); qu[2] = wv::bcast<14>(q); qu[3] = wv::bcast<15>(q);
#else
    // Round 4: the 4x4x1 outer product with a unit B operand IS a broadcast -- d[v] (every lane of the row) = a (lane 12 + v) * 1:
    // four lanes' values in one instruction (exact: one multiplication by 1, + 0).  Five instructions for Quu and qu.
    {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 r0 = wv::mfma4<3>(Q[12], 1.f, z4), r1 = wv::mfma4<3>(Q[13], 1.f, z4);
        const f32x4 r2 = wv::mfma4<3>(Q[14], 1.f, z4), r3 = wv::mfma4<3>(Q[15], 1.f, z4), rq = wv::mfma4<3>(q, 1.f, z4);
        S.s00 = r0[0]; S.s01 = r0[1]; S.s02 = r0[2]; S.s03 = r0[3];
        S.s11 = r1[1]; S.s12 = r1[2]; S.s13 = r1[3];
        S.s22 = r2[2]; S.s23 = r2[3];
        S.s33 = r3[3];
        qu[0] = rq[0]; qu[1] = rq[1]; qu[2] = rq[2]; qu[3] = rq[3];
    }
#endif

    bool fr[4] = {true, true, true, true};
    const bool valid[4] = {true, true, true, true};
    Ldl4 f;
    QpFac4 qf;                      // box QP: the factorisation of its last trip
    float kq[4] = {0.f, 0.f, 0.f, 0.f};
    if (!con(MODE)) {
        float sing = 0.f;
        ldl4<false, true>(f, S, fr, 0.f, &sing);                       // :84-94 (pinverse: see pivot_inv)
        if (sing != 0.f) st.status |= MPC_ST_QUU_SINGULAR;
    } else if (MODE == 1) {
        // :99-127 u_zero_I: masked rows and columns drop out
#pragma unroll
        for (int a = 0; a < 4; ++a) fr[a] = ((s.zm >> (8 * a)) & 0xffu) == 0u;
        ldl4<true>(f, S, fr, 0.f);
    } else {
        // :128-141 box constraints in delta space
        float lb[4], ub[4], ubar[4];
#ifdef MPC_DPP16_DPP_BCAST
        ubar[0] = wv::bcast<12>(s.tb); ubar[1] = wv::bcast<13>(s.tb); ubar[2] = wv::bcast<14>(s.tb); ubar[3] = wv::bcast<15>(s.tb);
#else
        {
            const f32x4 ru = wv::mfma4<3>(s.tb, 1.f, f32x4{0.f, 0.f, 0.f, 0.f});
            ubar[0] = ru[0]; ubar[1] = ru[1]; ubar[2] = ru[2]; ubar[3] = ru[3];
        }
#endif
        const float dlt = p.has_delta ? p.delta_u : 3.0e38f;       // :132-134 (no trust region: never the tighter bound)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            lb[a] = fmaxf(s.lo[a] - ubar[a], -dlt);
            ub[a] = fminf(s.hi[a] - ubar[a], dlt);
        }
        if (MPC_QP_START && p.qp_start) {
            // the caller's start (mpc_lqr_options.qp_start): a hint -- the solve below ends on a confirmed free set whatever it is.
            // (a NaN start would survive the clamp and poison the QP: such an entry falls back to the middle of the box's reach)
#pragma unroll
            for (int a = 0; a < 4; ++a) kq[a] = (s.qs[a] == s.qs[a]) ? s.qs[a] : 0.f;
        } else if (!MPC_DPP16_QP_WARM || !st.warm) {
            // cold start x = -H^-1 q (mpc/pnqp.py:14-19) -- at EVERY timestep whose Quu is positive definite (round 6).  The reference
            // hands timestep t the solution of timestep t+1 (mpc/lqr_step.py:137,141); a strictly convex QP has ONE minimiser, so
            // the start decides the trip count and nothing else (qp_start), and the clamped unconstrained minimiser is the
            // better guess of the active set by a whole trip: 2.91 -> 1.99 trips per QP in the first iteration of the benchmark's
            // box-constrained solve, 2.62 -> 1.74, 2.32 -> 1.40, 2.03 -> 1.22 in the next three (tools/qp_start_study.py, float64
            // numpy on the harvested QPs) -- one factorisation and a confirming gradient instead of two and one, for the ~40
            // instructions of an LDL' and its solve.  A Quu that is NOT positive definite (a pivot <= 0 or not finite: an indefinite
            // C) keeps the reference's start: where such a QP ends does depend on where it starts.
            ldl4<false>(f, S, valid, 0.f);
            float y[4];
            ldl4_solve(f, qu[0], qu[1], qu[2], qu[3], y);
            const bool spd = fminf(fminf(f.i0, f.i1), fminf(f.i2, f.i3)) > 0.f && fmaxf(fmaxf(f.i0, f.i1), fmaxf(f.i2, f.i3)) < 3.0e38f;
            const bool cold = !st.warm || spd;
#pragma unroll
            for (int a = 0; a < 4; ++a) kq[a] = cold ? -y[a] : st.kprev[a];
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) kq[a] = st.kprev[a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) kq[a] = eclampf(kq[a], lb[a], ub[a]);
        bool conv = false;
        const int it = pnqp4_rows(S, qu, lb, ub, p.pnqp_iter, L.j, kq, fr, qf, conv);
        st.qp_total += 1 + it;                                      // :140
        if (!conv) st.status |= MPC_ST_PNQP_UNCONVERGED;
        st.warm = 1;
#pragma unroll
        for (int a = 0; a < 4; ++a) st.kprev[a] = kq[a];
    }

    PROF_MARK(11);              // slot 11: the control block's factorisation / the box QP (slot 3 keeps the rest of the step)
    // K[:, j] = -H_free^-1 Qux[:, j]; lane 12 solves for k = -H_free^-1 qu instead
    const bool j12 = L.j == 12;
    float rhs[4], K[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) rhs[a] = sel(j12, qu[a], Q[12 + a]);
    {
        float y[4];
        if (!con(MODE)) {
            ldl4_solve(f, rhs[0], rhs[1], rhs[2], rhs[3], y);
#pragma unroll
            for (int a = 0; a < 4; ++a) K[a] = -y[a];
        } else {
#ifdef MPC_DPP16_QP_LDL
            ldl4_solve(MODE == 2 ? qf : f, fr[0] ? rhs[0] : 0.f, fr[1] ? rhs[1] : 0.f, fr[2] ? rhs[2] : 0.f, fr[3] ? rhs[3] : 0.f, y);
#else
            if (MODE == 2) {
#pragma unroll
                for (int a = 0; a < 4; ++a) y[a] = fr[a] ? rhs[a] : 0.f;
                gj4_solve(qf, y);
            } else {
                ldl4_solve(f, fr[0] ? rhs[0] : 0.f, fr[1] ? rhs[1] : 0.f, fr[2] ? rhs[2] : 0.f, fr[3] ? rhs[3] : 0.f, y);
            }
#endif
#pragma unroll
            for (int a = 0; a < 4; ++a) K[a] = fr[a] ? -y[a] : 0.f;
            if (MODE == 2) {
#pragma unroll
                for (int a = 0; a < 4; ++a) K[a] = sel(j12, kq[a], K[a]);     // k is the QP solution (:136-141)
            }
        }
    }

    // V = Qxx + Qxu K + K'(Qux + Quu K),  v = qx + Qxu k + K'(qu + Quu k)    (:155-158)
    float Vn[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Vn[i] = Q[i];
    float M[4] = {0.f, 0.f, 0.f, 0.f};
    if (con(MODE)) {
        // with a full free set Qux + Quu K vanishes; with masked / clamped controls it does not
#ifdef MPC_DPP16_DPP_BCAST
        M[0] = fmaf(S.s03, K[3], fmaf(S.s02, K[2], fmaf(S.s01, K[1], fmaf(S.s00, K[0], rhs[0]))));
        M[1] = fmaf(S.s13, K[3], fmaf(S.s12, K[2], fmaf(S.s11, K[1], fmaf(S.s01, K[0], rhs[1]))));
        M[2] = fmaf(S.s23, K[3], fmaf(S.s22, K[2], fmaf(S.s12, K[1], fmaf(S.s02, K[0], rhs[2]))));
        M[3] = fmaf(S.s33, K[3], fmaf(S.s23, K[2], fmaf(S.s13, K[1], fmaf(S.s03, K[0], rhs[3]))));
#else
        // M = Qux + Quu K as four rank-1 updates on the matrix core: column b of Quu sits in lanes 12..15 of register Q[12 + b]
        // (Quu symmetric), row b of K one entry per lane -- the same multiply-adds in the same order, four instructions for sixteen
        f32x4 Mv = {rhs[0], rhs[1], rhs[2], rhs[3]};
#pragma unroll
        for (int b = 0; b < 4; ++b) Mv = wv::mfma4<3>(Q[12 + b], K[b], Mv);
        M[0] = Mv[0]; M[1] = Mv[1]; M[2] = Mv[2]; M[3] = Mv[3];
#endif
    }
    wv::sched_fence();
    feed.template part<3>();
    wv::sched_fence();              // the matrix-core block of the value update, undivided
#pragma unroll
    for (int a = 0; a < 4; ++a) outer_acc(Vn, Q[12 + a], K[a]);      // += Qux[a][i] K[a][j]  (Qux = Qxu')
#ifdef MPC_DPP16_KEEP_KM
    // K'(Qux + Quu K) is zero but for rounding: a pinned row of K is exactly zero, a free row of the bracket is what the
    // solve above enforced (lqr_mfma40_body.h dropped the term in round 3; diagnostic switch for the A/B)
    if (con(MODE)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) outer_acc(Vn, K[a], M[a]);       // += K[a][i] M[a][j]
    }
#endif
    wv::sched_fence();
    float vn = q;
    // (K and M were finished in front of the matrix-core block above: settled sources, no wait states)
    wv::fmac_bcast_settled<12>(vn, K[0], Q[12]); wv::fmac_bcast_settled<12>(vn, K[1], Q[13]);
    wv::fmac_bcast_settled<12>(vn, K[2], Q[14]); wv::fmac_bcast_settled<12>(vn, K[3], Q[15]);
    if (con(MODE)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) wv::fmac_bcast_settled<12>(vn, M[a], K[a]);      // += K[a][j] m[a]
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) st.Vc[i] = Vn[i];
    st.vv = vn;
    // the constant of the value function: w_t = w_{t+1} + 0.5 k'Quu k + qu'k = w_{t+1} + 0.5 k'(m + qu),
    // m = qu + Quu k (lane 12 of M; zero without constraints)
    {
        float w = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float ka = wv::bcast<12>(K[a]);
            const float mq = con(MODE) ? wv::bcast<12>(M[a]) + qu[a] : qu[a];
            w = fmaf(ka, 0.5f * mq, w);
        }
        st.w0 += (double)w;
    }

    // gains: the wave's own record Kk[t][b][j][4] = K[.][j] (j < 12), k (j = 12), columns 1..3 of Quu (j = 13..15,
    // Quu[0][0] in place of the redundant Quu[2][1], see lane_init); a second record (m, M) in the same layout when
    // constraints are present; and K [T,B,4,12] / k [T,B,4] in the reference's layout when asked for.
    // (Rows of a partial last wave repeat problem B-1: the same values to the same addresses.)
    {
        const bool quu = L.j >= 13;
        // (round 5, constrained modes) row a of the record is K[a] for a FREE control and M[a] = (Qux + Quu K)[a] for a pinned one:
        // a pinned control's row of K is exactly zero, a free control's row of M is what the solve enforced -- zero but for
        // rounding, and taken as zero from here on -- so one 4 x 12 block and four flag bits carry both.  The rollouts took M from a
        // second 256-byte record: 512 of the 3,936 bytes a problem-step moved (written here, read back there).  m = qu + Quu k
        // (lane 12 of M) keeps a record of its own, 16 bytes; lane 12 of THIS record stays k (mpc_lqr_qp_record).
        f32x4 rec = {sel(quu, Q[12], K[0]), sel(quu, Q[13], K[1]), sel(quu, Q[14], K[2]), sel(quu, Q[15], K[3])};
        if (con(MODE)) {
            const bool below12 = L.j < 12;
#pragma unroll
            for (int a = 0; a < 4; ++a) rec[a] = (below12 && !fr[a]) ? M[a] : rec[a];
            // the free-set flags in a slot of the Quu block nobody reads (element 3 of lane 13 = Quu[3][1], which is read as Quu[1][3])
            const unsigned fm = (fr[0] ? 1u : 0u) | (fr[1] ? 2u : 0u) | (fr[2] ? 4u : 0u) | (fr[3] ? 8u : 0u);
            rec[3] = sel(L.j == 13, wv::bits_f32(fm), rec[3]);
        }
        rec[2] = sel(L.j == 13, S.s00, rec[2]);
        if (rgm(MODE)) {
            gain_put(G, t, rec);                         // stays in the register file until the rollouts
        } else {
            wv::store_f32x4(st.rec, rec);
            st.rec -= st.rec_step;
        }
        if (con(MODE)) {
            if (j12) wv::store_f32x4(st.rec2, f32x4{M[0], M[1], M[2], M[3]});       // m = qu + Quu k
            st.rec2 -= st.rec2_step;
        }
#ifndef MPC_DPP16_PROF
        if (p.K != nullptr) {
            const long tb = (long)t * p.B + L.pb;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (PADK) {
                    if (a < p.nc) {
                        if (L.j < p.ns) p.K[(tb * p.nc + a) * p.ns + L.j] = K[a];
                        else if (j12) p.k[tb * p.nc + a] = K[a];
                    }
                } else if (L.j < 12) p.K[(tb * 4 + a) * 12 + L.j] = K[a];
                else if (j12) p.k[tb * 4 + a] = K[a];
            }
        }
#endif
    }
}

// ---------------------------------------------------------------------------
// Rollout
// ---------------------------------------------------------------------------
struct RoStage {
    float Cr[16];     // row j of C                         (direct pricing only)
    float Fr[16];     // row j of F (state lanes)
    float Kr[12];     // row a of K (control lanes)
    float Mc[4];      // column j of M = Qux + Quu K (j < 12), m = qu + Quu k (j = 12): this lane's own granule of the sweep's
                      // second record (identity pricing, constrained modes)
    float Sr[4];      // row a of Quu                       (identity pricing)
    float cj, tb, fj, kk, lo, hi;
    f32x4 rec;        // register-resident gains (mode 0): the record in the sweep's own layout, see Gains
    bool zm;
};

// Mode 0: k_a and row a of Quu for control lane 12 + a out of the column-layout record -- needed only where a
// trial's step differs from the sweep's policy (alpha < 1), i.e. not in the first pass of a line search.
MPC_DEV void rg_price_terms(RoStage &s, const Lane &L)
{
    const f32x4 r = s.rec;
    const float k0 = wv::bcast<12>(r[0]), k1 = wv::bcast<12>(r[1]), k2 = wv::bcast<12>(r[2]), k3 = wv::bcast<12>(r[3]);
    const int a = L.a;
    s.kk = a == 0 ? k0 : (a == 1 ? k1 : (a == 2 ? k2 : k3));
    // Quu[i][c] = element i of lane 12 + c (c = 1..3), Quu[0][0] = element 2 of lane 13 (lane_init)
    const float q00 = wv::bcast<13>(r[2]);
    const float q01 = wv::bcast<13>(r[0]), q11 = wv::bcast<13>(r[1]);
    const float q02 = wv::bcast<14>(r[0]), q12 = wv::bcast<14>(r[1]), q22 = wv::bcast<14>(r[2]);
    const float q03 = wv::bcast<15>(r[0]), q13 = wv::bcast<15>(r[1]), q23 = wv::bcast<15>(r[2]), q33 = wv::bcast<15>(r[3]);
    s.Sr[0] = a == 0 ? q00 : (a == 1 ? q01 : (a == 2 ? q02 : q03));
    s.Sr[1] = a == 0 ? q01 : (a == 1 ? q11 : (a == 2 ? q12 : q13));
    s.Sr[2] = a == 0 ? q02 : (a == 1 ? q12 : (a == 2 ? q22 : q23));
    s.Sr[3] = a == 0 ? q03 : (a == 1 ? q13 : (a == 2 ? q23 : q33));
}

template <int MODE, bool DIRECT>
MPC_DEV void ro_read(RoStage &s, const P &p, const Lane &L, int t, int slot, unsigned zm)
{
    const unsigned base = stage_mid<MODE, true, DIRECT>(slot) - SF;       // the lane offsets are SF-relative + SF
    const unsigned mrec = base + RollRing<MODE, DIRECT>::MADJ;
    const unsigned gain = base + RollRing<MODE, DIRECT>::GADJ;
    s.cj = 0.f;
    if (DIRECT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = wv::lds_f32x4(base + L.aCq[q]);
            s.Cr[4 * q] = v[0]; s.Cr[4 * q + 1] = v[1]; s.Cr[4 * q + 2] = v[2]; s.Cr[4 * q + 3] = v[3];
        }
        s.cj = wv::lds_f32(base + L.aRec + R_c);
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) s.Sr[b] = rgm(MODE) ? 0.f : wv::lds_f32(gain + L.aS[b]);
        if (con(MODE)) {
            // (round 4: e'(m + M dx) = sum_j [dx_j; 1] (M | m)'e -- each lane needs ITS column, one 16-byte read, and four
            // broadcast multiply-adds of e; rounds 1-3 read row a of M on the control lanes: 13 reads, 12 multiply-adds)
            // (round 5: column j of M is this lane's own granule of the GAIN record, the rows of the pinned controls; m has the
            // 16-byte record of its own, lane 12's granule of the stage's m block)
            const f32x4 mc = wv::lds_f32x4(L.j == 12 ? mrec + L.aMcol : gain + L.aGcol);
            const unsigned fm = wv::f32_bits(wv::lds_f32(gain + L.aFm));
#pragma unroll
            for (int a = 0; a < 4; ++a) s.Mc[a] = (L.j == 12 || !((fm >> a) & 1u)) ? mc[a] : 0.f;
        }
    }
    // (t = T-1: a copy of F[T-2], unused)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = wv::lds_f32x4(base + L.aFq[q]);
        s.Fr[4 * q] = v[0]; s.Fr[4 * q + 1] = v[1]; s.Fr[4 * q + 2] = v[2]; s.Fr[4 * q + 3] = v[3];
    }
    s.fj = p.f ? wv::lds_f32(base + L.aRecF) : 0.f;
    if (!rgm(MODE)) {
#pragma unroll
        for (int jj = 0; jj < 12; ++jj) s.Kr[jj] = wv::lds_f32(gain + L.aKrow + 16 * jj);
        if (con(MODE)) {
            // a pinned control's row of the record is its row of M (sweep_step): its row of K is zero
            const unsigned fm = wv::f32_bits(wv::lds_f32(gain + L.aFm));
            const bool free_a = ((fm >> L.a) & 1u) != 0u;
#pragma unroll
            for (int jj = 0; jj < 12; ++jj) s.Kr[jj] = free_a ? s.Kr[jj] : 0.f;
        }
        s.kk = wv::lds_f32(gain + L.aKrow + 192);
    } else {
        s.kk = 0.f;
    }
    s.tb = wv::lds_f32(base + L.aRec + R_tau);
    s.lo = s.hi = 0.f;
    if (MODE == 2) {
        if (p.bound_mode == MPC_BOUND_TENSOR) {
            s.lo = wv::lds_f32(base + L.aRecA + R_lo);
            s.hi = wv::lds_f32(base + L.aRecA + R_hi);
        } else {
            const bool pad = PADK && !L.ovalid;
            s.lo = pad ? 0.f : p.lo_s;
            s.hi = pad ? 0.f : p.hi_s;
        }
    }
    s.zm = ((zm >> (8 * L.a)) & 0xffu) != 0u;
}

struct RoState {
    float xs;         // x'_t[j] (state lanes)
    float cost, du2;  // per-lane partials (identity pricing: of J - J_nominal - w_0)
    float alpha;      // line-search step of this row's problem
    float pred;       // F tau + f of the nominal at t-1: what the nominal x_t must equal
    float viol;       // > 0 once the nominal broke the dynamics somewhere
    float *out;       // this lane's element of new_x / new_u at the current timestep
    bool price_on;    // (wave-uniform) some row of this pass steps off the sweep's policy: the identity's terms are not all zero
    // the pair pass of the box-constrained line search: a second trial (alpha = decay) rolled out beside the first, its
    // trajectory parked in the workspace
    float xs1, cost1, du21;
    float *out1;
};

// new_u = K dx + u + alpha k (mpc/lqr_step.py:192), zero mask (:197-198), box / delta_u clamp (:200-213);
// control lanes hold row a of K.  `e` = du - K dx - k, the distance from the sweep's policy.
template <int MODE>
MPC_DEV float control_law(const P &p, const Lane &L, const RoStage &s, float xs, float alpha, float &e, float &dx)
{
    dx = L.isu ? 0.f : xs - s.tb;
    float un;
    if (rgm(MODE)) {
        // the record is in the sweep's layout (lane j: K[0..3][j], lane 12: k): K dx + alpha k is ONE reduction over the
        // row per control, lane 12 contributing alpha k_a -- four sums that a two-level exchange lands in the lanes
        // with (j & 3) == a, control lane 12 + a among them
        const float mult = L.j == 12 ? alpha : dx;
        un = s.tb + wv::quad_sums(s.rec[0] * mult, s.rec[1] * mult, s.rec[2] * mult, s.rec[3] * mult, L.j);
    } else {
        un = fmaf(alpha, s.kk, s.tb);
        wv::dot_bcast12(un, dx, s.Kr);
    }
    const float pre = un;
    if (con(MODE) && s.zm) un = 0.f;
    if (MODE == 2) {
        // (straight-line: without a trust region u -+ 3e38 never narrows a bound; max / min pick what the reference's
        // compare-and-select chain picks, :202-207)
        const float dlt = p.has_delta ? p.delta_u : 3.0e38f;
        const float l = fmaxf(s.lo, s.tb - dlt), h = fminf(s.hi, s.tb + dlt);
        un = eclampf(un, l, h);
    }
    e = L.isu ? (un - pre) + (alpha - 1.f) * s.kk : 0.f;
    return un;
}

// The stage's contribution to the trajectory cost.  DIRECT: 0.5 tau'C tau + c'tau (mpc/lqr_step.py:230-232).
// Otherwise e'(m + M dx) + 0.5 e'Quu e (see the header): control lanes only.
template <int MODE, bool DIRECT>
MPC_DEV float stage_price(const Lane &L, const RoStage &s, float tp, float e, float dx)
{
    if (DIRECT) {
        float sq = 0.f;
        wv::dot_bcast16(sq, tp, s.Cr);
        return tp * fmaf(0.5f, sq, s.cj);
    }
    float se = 0.f;
    wv::dot_bcast_u4(se, e, s.Sr);                 // sum_b bcast_{12+b}(e) Sr[b]
    const float quad = L.isu ? e * (0.5f * se) : 0.f;          // e'Quu e / 2, on the control lanes
    if (!con(MODE)) return quad;
    // e'(m + M dx): lane j < 12 adds dx_j (M'e)_j, lane 12 (whose column is m) adds m'e
    float w = 0.f;
    wv::dot_bcast_u4(w, e, s.Mc);
    const float mult = L.j < 12 ? dx : 1.f;
    return L.j <= 12 ? fmaf(mult, w, quad) : quad;          // (lanes 13..15 hold no column of the record)
}

template <int MODE, bool DIRECT, bool CHECK, bool PAIR = false>
MPC_DEV void rollout_step(const P &p, const Lane &L, const RoStage &s, RoState &st, int t, Feed<MODE, true, DIRECT> &feed)
{
    const bool last = (t == p.T - 1);
    feed.template part<0>();
    float e, dx;
    const float un = control_law<MODE>(p, L, s, st.xs, st.alpha, e, dx);
    const float tp = L.isu ? un : st.xs;                             // tau'_t[j]
    feed.template part<1>();
    // (mode 0 on the sweep's own policy, alpha = 1 everywhere: e = 0, the identity's stage terms vanish)
    if (DIRECT || !rgm(MODE) || st.price_on) st.cost += stage_price<MODE, DIRECT>(L, s, tp, e, dx);
    {
        const float d = sel(L.isu, s.tb - un, 0.f);                  // (selects, not branches: lane-dependent
        st.du2 = fmaf(d, d, st.du2);                                 //  branches cost exec-mask bookkeeping)
    }
    if (!PADK || L.ovalid) wv::store_out(st.out, tp);           // new_u (control lanes) / new_x: one store
    st.out += L.ostep;
    feed.template part<2>();
    if (!DIRECT && CHECK) {
        // does the nominal obey x_t = F tau_{t-1} + f_{t-1}?  (the identity above assumes it; CHECK is off when the
        // caller vouches for it, MPC_OPT_NOMINAL_ON_DYNAMICS)
        if (t > 0) {
            float r = fabsf(st.pred - s.tb) - 1e-5f * (1.f + fabsf(s.tb));
            r = (r == r) ? r : 1.f;
            r = sel(L.isu, 0.f, r);
            st.viol = r > st.viol ? r : st.viol;
        }
        if (!last) {
            float pn = s.fj;
            wv::dot_bcast16(pn, s.tb, s.Fr);
            st.pred = pn;
        }
    }
    feed.template part<3>();
    // x_{t+1} = F [x;u] + f  (:216-222)
    if (!last) {
        float xn = s.fj;
        wv::dot_bcast16(xn, tp, s.Fr);
        st.xs = xn;
    }
    if (PAIR) {
        // the second trial, alpha = decay, off the same stage registers
        float e1, dx1;
        const float un1 = control_law<MODE>(p, L, s, st.xs1, p.ls_decay, e1, dx1);
        const float tp1 = L.isu ? un1 : st.xs1;
        st.cost1 += stage_price<MODE, DIRECT>(L, s, tp1, e1, dx1);
        {
            const float d = sel(L.isu, s.tb - un1, 0.f);
            st.du21 = fmaf(d, d, st.du21);
        }
        *st.out1 = tp1;                      // (a plain store: read back by this wave within the launch, line_search)
        st.out1 += L.ostep1;
        if (!last) {
            float xn = s.fj;
            wv::dot_bcast16(xn, tp1, s.Fr);
            st.xs1 = xn;
        }
    }
}

// The line-search trials alpha = decay^k rolled out side by side off ONE pass over the data (read from LDS
// once per timestep, every trial has its own state register): only the costs come out; the accepted
// trial is replayed by a storing pass.
enum { MAX_TRIALS = 15 };
struct Trials {
    float xs[MAX_TRIALS], cost[MAX_TRIALS], alpha[MAX_TRIALS];
    // (round 5, the box-constrained step) the LAST trial of the pass parks its trajectory in the workspace, for the rows still
    // searching: a problem that gets worse for every step size -- what the late iterations of a solve see, two to five problems of
    // 4096 -- ends on exactly that trial, and its wavefront then copies the trajectory out instead of replaying a pass
    float du2_last;
    float *park;
    bool park_row;
    int park_k;             // which of this row's trials is "the last one" (-1: none of them)
};

template <int MODE, bool DIRECT>
MPC_DEV void trials_step(const P &p, const Lane &L, const RoStage &s, Trials &tr, int nt, int t)
{
    const bool last = (t == p.T - 1);
    constexpr bool PARK = MODE == 2 && !DIRECT;
#pragma unroll
    for (int k = 0; k < MAX_TRIALS; ++k) {
        if (k < nt) {
            float e, dx;
            const float un = control_law<MODE>(p, L, s, tr.xs[k], tr.alpha[k], e, dx);
            const float tp = L.isu ? un : tr.xs[k];
            tr.cost[k] += stage_price<MODE, DIRECT>(L, s, tp, e, dx);
            if (PARK && k == tr.park_k) {
                const float d = sel(L.isu, s.tb - un, 0.f);
                tr.du2_last = fmaf(d, d, tr.du2_last);
                if (tr.park_row) *tr.park = tp;          // (a plain store: read back by this wave within the launch)
                tr.park += L.ostep1;
            }
            if (!last) {
                float xn = s.fj;
                wv::dot_bcast16(xn, tp, s.Fr);
                tr.xs[k] = xn;
            }
        }
    }
}

// One pass over the horizon.  MULTI: the nt trials of tr (costs only); otherwise the single trial of st
// (trajectory stored).  Costs come back as full trajectory costs (base = J_nominal + w_0 when priced by
// the identity, 0 when priced directly).
template <int MODE, bool MULTI, bool DIRECT, bool CHECK, bool PAIR = false>
MPC_DEV void rollout_pass(const P &p, const Lane &L, Dma &d, int wave, const Gains<rgm(MODE)> &G, RoState &st, Trials &tr, int nt, double base PROF_ARG)
{
    const int T = p.T;
    float x0 = L.isu ? 0.f : (PADK ? (L.ovalid ? p.x_init[(long)L.pb * p.ns + L.j] : 0.f) : p.x_init[(long)L.pb * 12 + L.j]);
    // have the load land HERE: a vector load still pending when the loop is entered makes the compiler drain the
    // whole DMA queue (s_waitcnt vmcnt(0)) in front of its first use in every trip -- it cannot count across the loop
    wv::pin(x0);
    if (MULTI) {
#pragma unroll
        for (int k = 0; k < MAX_TRIALS; ++k) { tr.xs[k] = x0; tr.cost[k] = 0.f; }
        tr.du2_last = 0.f;
        tr.park = L.scr0;
    } else {
        st.xs = x0;
        st.cost = 0.f;
        st.du2 = 0.f;
        st.pred = 0.f;
        st.out = L.out0;
        if (PAIR) {
            st.xs1 = x0;
            st.cost1 = 0.f;
            st.du21 = 0.f;
            st.out1 = L.scr0;
        }
    }
    st.price_on = MULTI || wv::any(st.alpha != 1.f);
    const bool use_zm = con(MODE) && p.zero_mask != nullptr;
    enum { NS = RollRing<MODE, DIRECT>::SLOTS, LA = NS - 1, ND = RollDma<MODE, DIRECT>::N };
    static_assert((LA - 1) * ND + 2 * LA < 64, "vmcnt is 6 bits");
    unsigned zq[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) zq[i] = 0u;
    dma_seek<MODE, true, DIRECT>(d, p, L, wave);
    pad_clear(L.lane);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        if (i < T) {
            stage_issue<MODE, true, DIRECT>(d, stage_mid<MODE, true, DIRECT>(i));
            stage_move<MODE, true, DIRECT>(d, i + 1 <= T - 2);
            if (use_zm) zq[i] = zm_pick(L, zm_fetch(p, L, i));
        }
    }
    for (int t0 = 0; t0 < T; t0 += NS) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int t = t0 + i;
            if (t < T) {
                // stages t+1 .. min(t + LA - 1, T - 1) are in flight behind the one needed now
                PROF_MARK(7);
                if (T - 1 - t >= LA - 1) wv::dma_wait<(LA - 1) * ND>();
                else tail_wait<(LA >= 2 ? LA - 2 : 0), ND>(T - 1 - t);
                PROF_MARK(4);
                RoStage s;
                ro_read<MODE, DIRECT>(s, p, L, t, i, zq[i]);
                if (rgm(MODE)) {
                    s.rec = gain_get(G, t);
                    if (!DIRECT && st.price_on) rg_price_terms(s, L);
                }
                PROF_MARK(5);
                const int tn = t + LA;
                ZmRaw zr = {{0u, 0u, 0u, 0u}};
                if (use_zm && tn < T) zr = zm_fetch(p, L, tn);
                Feed<MODE, true, DIRECT> feed = {d, stage_mid<MODE, true, DIRECT>((i + LA) % NS), tn < T, tn + 1 <= T - 2};
                PROF_MARK(6);
                if (MULTI) {
                    feed.template part<0>(); feed.template part<1>(); feed.template part<2>(); feed.template part<3>();
                    trials_step<MODE, DIRECT>(p, L, s, tr, nt, t);
                } else {
                    rollout_step<MODE, DIRECT, CHECK, PAIR>(p, L, s, st, t, feed);
                }
                if (use_zm) zq[(i + LA) % NS] = zm_pick(L, zr);
            }
        }
    }
    wv::dma_wait<0>();
    // (round 6) ONE rounding: J_nominal + w_0 + (stage terms) is added up in double and rounded once.  Rounded twice --
    // float(J_nominal + w_0) first, the stage terms onto that -- a trial whose true change is a small DEcrease could come out one
    // ulp ABOVE float(J_nominal) when the first rounding went up, and it did so for every step size alike (the stage terms tend to
    // -w_0 as alpha -> 0): in the late iterations of a box-constrained solve two to five problems of 4096 "got worse for every step
    // size", searched to the end, and their one wavefront was the launch (206-216 us against 148).  Rounding is monotone: with one
    // rounding a trial that does not raise the cost never compares above the nominal's (mpc/lqr_step.py:176-179).
    if (MULTI) {
#pragma unroll
        for (int k = 0; k < MAX_TRIALS; ++k)
            if (k < nt) tr.cost[k] = (float)(base + (double)wv::row_sum(tr.cost[k]));
        tr.du2_last = wv::row_sum(tr.du2_last);
    } else {
        st.cost = (float)(base + (double)wv::row_sum(st.cost));
        st.du2 = wv::row_sum(st.du2);
        if (PAIR) {
            st.cost1 = (float)(base + (double)wv::row_sum(st.cost1));
            st.du21 = wv::row_sum(st.du21);
        }
    }
}

// The lane offsets of row L as if it held problem slot `ps` of the wave (identity-priced rollouts only: F rows, record, gains):
// what lets the four rows of a wavefront roll out trials of ONE problem side by side (line_search).
MPC_DEV Lane lane_as_slot(const Lane &L, int ps, const P &p, int wave)
{
    Lane X = L;
    const int dp = ps - L.p;
    const int pb = 4 * wave + ps < p.B ? 4 * wave + ps : p.B - 1;
    X.p = ps;
    X.pb = pb;
#pragma unroll
    for (int q = 0; q < 4; ++q) X.aFq[q] += dp * 768;
    X.aRec += dp * 256; X.aRecA += dp * 256; X.aRecF += dp * 256;
    X.aKrow += dp * 256; X.aMcol += dp * 256; X.aGcol += dp * 256; X.aFm += dp * 256;
#pragma unroll
    for (int b = 0; b < 4; ++b) X.aS[b] += dp * 256;
    X.scr0 = p.Kk + (long)p.T * p.B * 128 + (long)pb * 16 + L.j;
    return X;
}

// The line search of mpc/lqr_step.py:164-261, every row (problem) on its own.
// :176-179, 247, 252: the step shrinks while the cost got worse; the first trial that did not get worse is
// taken, else the last one.  Backtracking is usually one step deep (box-constrained problems) or runs to
// the end (a nominal that is already optimal): alpha = 1, then alpha = decay on its own, then ALL
// remaining trials in one pass and a replay of the accepted ones.
template <int MODE, bool DIRECT, bool CHECK>
MPC_DEV void line_search(const P &p, const Lane &L, Dma &d, int wave, const Gains<rgm(MODE)> &G, RoState &rs, float old_cost, double base, float &full2 PROF_ARG)
{
    Trials tr;
    rs.alpha = 1.f;
    bool worse0 = false;
    bool copy_parked = false;
    if (MODE == 2 && !DIRECT && p.max_ls >= 2) {
        // Box constraints: one problem in six steps back to alpha = decay, i.e. every second wave -- and the slowest wave
        // is the kernel's time.  The first pass therefore rolls out alpha = 1 AND alpha = decay side by side (one read of
        // the stage, two states; the second trajectory goes to the workspace); a row that takes the second trial copies
        // it over afterwards.  Only a row that rejects both (rare) sends the wave into the remaining trials + a replay.
        rollout_pass<MODE, false, DIRECT, CHECK, true>(p, L, d, wave, G, rs, tr, 0, base PROF_PASS);
        full2 = rs.du2;                                              // :243-245 (the alpha = 1 trial)
        worse0 = rs.cost > old_cost;
        if (!wv::any(worse0)) return;
        const bool worse1 = worse0 && rs.cost1 > old_cost && p.max_ls > 2;
        if (worse0) {
            rs.alpha = p.ls_decay;
            rs.cost = rs.cost1;
            rs.du2 = rs.du21;
        }
        if (wv::any(worse1)) {
            const int nt = p.max_ls - 2;                                 // trials alpha = decay^2 .. decay^(max_ls-1)
            bool ended_on_last = false;
            const unsigned long long wm = wv::ballot(worse1 && L.j == 0);    // bit 16 r: row r is still searching
            const int n_w = (int)__builtin_popcountll(wm);
            // (round 5) ONE row left -- in the late iterations of a solve, one row of one wavefront in a thousand: a problem that gets
            // worse for every step size --: the FOUR rows roll out the trials of that one problem side by side, nt / 4 each (the stage
            // holds all four problems' blocks: a row reads another slot's by its offsets), instead of every row all nt trials of its
            // own problem: a quarter of the pass's arithmetic.  Same step sizes, same rule.  ONE call
            // site serves both forms (every row its own problem = slot L.p, all nt trials): the pass is inlined where it is called.
            const bool shared = n_w == 1 && nt >= 4;
            const int ps1 = (int)(__builtin_ctzll(wm | (1ull << 63)) >> 4);                   // the one row still searching
            const int nt_row = shared ? (nt + 3) / 4 : nt;
            const int r_last = shared ? (nt - 1) / nt_row : 0;
            if (shared) MPC_STAT(7); else MPC_STAT(8);
            {
                const int ps = ps1;
                const Lane X = lane_as_slot(L, shared ? ps : L.p, p, wave);
                const int idx0 = shared ? L.p * nt_row : 0;
                float a = p.ls_decay;
                for (int i = 0; i <= (idx0 < nt - 1 ? idx0 : nt - 1); ++i) a *= p.ls_decay;            // decay^(2 + min(idx0, nt - 1))
#pragma unroll
                for (int k = 0; k < MAX_TRIALS; ++k) {
                    tr.alpha[k] = a;
                    if (idx0 + k + 1 <= nt - 1) a *= p.ls_decay;                                      // (trials past the last repeat it)
                }
                tr.park_k = shared ? (L.p == r_last ? (nt - 1) - r_last * nt_row : -1) : nt - 1;
                tr.park_row = shared || worse1;                                // (a row that took alpha = decay keeps ITS parked trajectory)
                const double base_s = shared ? wv::readlane_f64(base, 16 * ps) : base;
                const float old_s = shared ? wv::readlane(old_cost, 16 * ps) : old_cost;
                rollout_pass<MODE, true, DIRECT, CHECK>(p, X, d, wave, G, rs, tr, nt_row, base_s PROF_PASS);
                // the first trial that did not get worse, else the last
                float acc = 0.f, cacc = 0.f, du2_l = tr.du2_last;
                int kacc = 0;
                bool found = false;
                if (shared) {
                    // trial idx = r nt_row + k sits in row r, slot k
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int k = 0; k < MAX_TRIALS; ++k) {
                            if (k < nt_row && r * nt_row + k < nt && !found) {
                                acc = wv::readlane(tr.alpha[k], 16 * r);
                                cacc = wv::readlane(tr.cost[k], 16 * r);
                                kacc = r * nt_row + k;
                                if (!(cacc > old_s)) found = true;
                            }
                        }
                    }
                    du2_l = wv::readlane(tr.du2_last, 16 * r_last);
                } else {
#pragma unroll
                    for (int k = 0; k < MAX_TRIALS; ++k) {
                        if (k < nt && !found) {
                            acc = tr.alpha[k];
                            cacc = tr.cost[k];
                            kacc = k;
                            if (!(tr.cost[k] > old_cost)) found = true;
                        }
                    }
                }
                if (shared ? L.p == ps : worse1) {
                    rs.alpha = acc;
                    ended_on_last = kacc == nt - 1;
                    if (ended_on_last) { rs.cost = cacc; rs.du2 = du2_l; }
                }
            }
            if (wv::any(worse1 && !ended_on_last)) {
                rollout_pass<MODE, false, DIRECT, CHECK>(p, L, d, wave, G, rs, tr, 0, base PROF_PASS);   // replay: every row stores its accepted trial
                return;
            }
        }
        // every row's accepted trajectory is on hand: trial 0 in new_x / new_u, trial 1 or the last one parked in the workspace
        copy_parked = true;
    }
    if (copy_parked) {
        // parked trajectory -> new_x / new_u, rows that did not take the full step (the stores of this wave's own passes must
        // have landed: they are read back through the vector path)
        PROF_MARK_ALL(9);
        wv::fence_own_stores();
        // COPY_N loads in flight, then their stores: one element per trip (load, wait, store) was a dependent
        // HBM round trip per timestep -- T of them in every second wave, and the slowest wave is the kernel's time
        enum { COPY_N = 16 };
        const float *src = L.scr0;
        float *dst = L.out0;
        const int T = p.T;
        for (int t0 = 0; t0 < T; t0 += COPY_N) {
            float v[COPY_N];
#pragma unroll
            for (int i = 0; i < COPY_N; ++i) {
                const int t = t0 + i < T ? t0 + i : T - 1;
                v[i] = src[(long)t * L.ostep1];
            }
#pragma unroll
            for (int i = 0; i < COPY_N; ++i) {
                const int t = t0 + i < T ? t0 + i : T - 1;
                if (worse0 && (!PADK || L.ovalid)) wv::store_out(dst + (long)t * L.ostep, v[i]);
            }
        }
        PROF_MARK_ALL(12);          // slot 12: copy of the parked trajectory
        return;
    }
#pragma unroll 1
    for (int phase = 0; phase < 3; ++phase) {
        rollout_pass<MODE, false, DIRECT, CHECK>(p, L, d, wave, G, rs, tr, 0, base PROF_PASS);   // rows whose alpha did not change reproduce their result
        if (phase == 0) {
            full2 = rs.du2;                                          // :243-245 (the alpha = 1 trial)
            worse0 = rs.cost > old_cost && p.max_ls > 1;
            if (!wv::any(worse0)) break;
            if (worse0) rs.alpha = p.ls_decay;
        } else if (phase == 1) {
            const bool worse1 = worse0 && rs.cost > old_cost && p.max_ls > 2;
            if (!wv::any(worse1)) break;
            const int nt = p.max_ls - 2;                          // trials alpha = decay^2 .. decay^(max_ls-1)
            float a = p.ls_decay;
#pragma unroll
            for (int k = 0; k < MAX_TRIALS; ++k) { a *= p.ls_decay; tr.alpha[k] = a; }
            rollout_pass<MODE, true, DIRECT, CHECK>(p, L, d, wave, G, rs, tr, nt, base PROF_PASS);
            if (worse1) {
                float acc = tr.alpha[0];
                bool found = false;
#pragma unroll
                for (int k = 0; k < MAX_TRIALS; ++k) {
                    if (k < nt && !found) {
                        acc = tr.alpha[k];
                        if (!(tr.cost[k] > old_cost)) found = true;
                    }
                }
                rs.alpha = acc;
            }
        }
    }
}

template <int MODE>
MPC_DEV void step_wave(const P &p)
{
    const int lane = wv::lane();
    // (which problem group a workgroup -- hence an XCD: block b runs on XCD b % 8 -- takes makes no difference:
    // four remappings measured within 0.5 %, profiles/r02_experiments.md)
    const int wave = wv::problem();          // one workgroup = one wave = four problems
    if (4 * wave >= p.B) return;
    Lane L;
    lane_init(L, lane, wave, p.B, !p.c_symmetric);
    L.out0 = L.isu ? p.new_u + (long)L.pb * 4 + L.a : p.new_x + (long)L.pb * 12 + L.j;
    L.ostep = L.isu ? (long)p.B * 4 : (long)p.B * 12;
    if (PADK) {
        // the caller's arrays by their true shape; a lane of the padding stores nothing (its out0 is never dereferenced)
        L.ovalid = L.isu ? L.a < p.nc : L.j < p.ns;
#pragma unroll
        for (int a = 0; a < 4; ++a) L.padd[a] = (L.j == 12 + a && a >= p.nc) ? 1.f : 0.f;
        L.out0 = L.isu ? p.new_u + (long)L.pb * p.nc + (L.ovalid ? L.a : 0) : p.new_x + (long)L.pb * p.ns + (L.ovalid ? L.j : 0);
        L.ostep = L.isu ? (long)p.B * p.nc : (long)p.B * p.ns;
    }
    L.scr0 = p.Kk + (long)p.T * p.B * 128 + (long)L.pb * 16 + L.j;      // behind the two records
    L.ostep1 = (long)p.B * 16;
    const int T = p.T;
    Dma d;
    Gains<rgm(MODE)> G;
    PROF_DECL;

    // ---- Riccati sweep, t = T-1 .. 0 ------------------------------------------------------------
    SwState ss;
#pragma unroll
    for (int i = 0; i < 12; ++i) ss.Vc[i] = 0.f;
    ss.vv = 0.f;
    ss.oc = 0.0;
    ss.w0 = 0.0;
    ss.warm = 0;
    ss.qp_total = 0;
    ss.status = 0;
    ss.asym = 0.f;
    ss.cmax = 0.f;
    ss.kprev[0] = ss.kprev[1] = ss.kprev[2] = ss.kprev[3] = 0.f;
    ss.rec_step = (long)p.B * 64;
    ss.rec = p.Kk + ((long)(T - 1) * p.B + L.pb) * 64 + 4 * L.j;
    ss.rec2_step = (long)p.B * 4;
    ss.rec2 = p.Kk + (long)T * p.B * 64 + ((long)(T - 1) * p.B + L.pb) * 4;
    {
        unsigned zq[NSTAGE] = {};
        dma_seek<MODE, false, false>(d, p, L, wave);
        pad_clear(L.lane);
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) {
            const int ti = T - 1 - i;
            if (ti >= 0) {
                stage_issue<MODE, false, false>(d, stage_mid<MODE, false, false>(i));
                stage_move<MODE, false, false>(d, ti <= T - 2);           // on to stage ti - 1
                if (MODE == 1) zq[i] = zm_pick(L, zm_fetch(p, L, ti));
            }
        }
        for (int k0 = 0; k0 < T; k0 += NSTAGE) {
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                const int t = T - 1 - (k0 + i);
                if (t >= 0) {
                    // stages t-1 and t-2 (where they exist) are in flight behind the one needed now
                    PROF_MARK(3);
                    if (t >= AHEAD - 1) wv::dma_wait<(AHEAD - 1) * DMA_SWEEP>();
                    else wv::dma_wait<0>();
                    PROF_MARK(0);
                    SwStage s;
                    sw_read<MODE>(s, p, L, t, i, zq[i], ss.asym, ss.cmax);
                    PROF_MARK(1);
                    ZmRaw zr = {{0u, 0u, 0u, 0u}};
                    if (MODE == 1 && t >= AHEAD) zr = zm_fetch(p, L, t - AHEAD);
                    Feed<MODE, false, false> feed = {d, stage_mid<MODE, false, false>((i + AHEAD) % NSTAGE), t >= AHEAD, true};
                    PROF_MARK(2);
                    sweep_step<MODE>(p, L, s, ss, t, feed, G PROF_PASS);
                    if (MODE == 1) zq[(i + AHEAD) % NSTAGE] = zm_pick(L, zr);
                }
            }
        }
        wv::dma_wait<0>();
    }
    PROF_MARK_ALL(8);           // slot 8: set-up + sweep tail; slots 0-3: wait / LDS reads / DMA issue / arithmetic of the sweep
    const double old_cost_d = wv::row_sum_f64(ss.oc);
    const float old_cost = (float)old_cost_d;
    // (a tolerance, not a bit test: C = A'A out of a float32 GEMM is symmetric to rounding only)
    if (!p.c_symmetric) ss.status |= MPC_ST_C_TESTED | (wv::row_max(ss.asym) > 1e-5f * wv::row_max(ss.cmax) ? (int)MPC_ST_C_ASYMMETRIC : 0);

    if (p.sweep_only) {                  // MPC_OPT_SWEEP_ONLY: the caller rolls out itself (a module as true_dynamics)
        if (L.j == 0) {
            if (p.old_costs) p.old_costs[L.pb] = old_cost;
            if (p.qp_iters) p.qp_iters[L.pb] = ss.qp_total;
            if (p.status) p.status[L.pb] = ss.status;
        }
        return;
    }
    // the gains were written by this wave and are re-read through the DMA: drain the stores
    wv::fence_own_stores();

    // ---- line-searched rollout (mpc/lqr_step.py:164-261) --------------------------------------
    RoState rs;
    rs.viol = 0.f;
    float full2 = 0.f;
    if (p.on_dynamics) {
        // the caller vouches for the nominal (MPC_OPT_NOMINAL_ON_DYNAMICS): no verification in the loop
        line_search<MODE, false, false>(p, L, d, wave, G, rs, old_cost, old_cost_d + ss.w0, full2 PROF_PASS);
    } else {
        line_search<MODE, false, true>(p, L, d, wave, G, rs, old_cost, old_cost_d + ss.w0, full2 PROF_PASS);
        // a nominal that does not obey the dynamics voids the identity the pass was priced with: price the
        // rollout the reference's way, from a second stream of C
        const bool off = wv::row_sum(rs.viol > 0.f ? 1.f : 0.f) > 0.f;
        // (round 5) ... and so is a problem one of whose box QPs did not converge -- in practice a Quu that is not positive definite:
        // V then grows to 1e5 times the cost, J_nominal + w0 is what is left of two float32 sums that size, and the identity's price
        // was seen 3 % off the cost of the very trajectory it belongs to (a positive definite problem: 1e-7).  The call that makes
        // no promises checks this premise like the other one; the vouched call trusts its caller with both (the same test on that
        // path -- a second site for the inlined pass -- cost the convex step 4 %: profiles/r05_ab_direct_pricing.log).
        // (NOT checked: a price that is the small difference of large numbers.  The identity's cost carries an absolute error of about
        // 3e-8 (|J_nominal| + |w0|) -- 1 % of the cost was seen on a nominal 1.5e5 times dearer than the step it leads to,
        // tools/emu_fuzz.py; the decisions stand on margins that size.  A test for it here fires on one problem in a few thousand
        // of the benchmark's batch, and one wavefront that re-prices is the launch: 92 -> 123 us.  DESIGN 6.)
        const bool broken = off || (MODE == 2 && (ss.status & MPC_ST_PNQP_UNCONVERGED) != 0);
        if (wv::any(broken)) line_search<MODE, true, false>(p, L, d, wave, G, rs, old_cost, 0.0, full2 PROF_PASS);
        if (off) ss.status |= MPC_ST_NOMINAL_OFF_DYNAMICS;
    }
#ifdef MPC_DPP16_PROF
    PROF_MARK_ALL(9);           // slot 9: between the loops + rollout tail; slots 4-7: the same four phases of the rollout
    if (p.K != nullptr && lane == 0)
        for (int i_ = 0; i_ < 16; ++i_) p.K[(long)wave * 16 + i_] = (float)prof_.acc[i_];
#endif
    int status = ss.status;
    if (!(rs.cost == rs.cost) || fabsf(rs.cost) > 3e38f) status |= MPC_ST_NONFINITE;
    if (L.j == 0) {
        const int b = L.pb;
        if (p.costs) p.costs[b] = rs.cost;
        if (p.old_costs) p.old_costs[b] = old_cost;
        if (p.full_du_norm) p.full_du_norm[b] = sqrtf(full2);
        if (p.alpha_du_norm) p.alpha_du_norm[b] = sqrtf(rs.du2);
        if (p.alphas) p.alphas[b] = rs.alpha;
        if (p.qp_iters) p.qp_iters[b] = ss.qp_total;
        if (p.status) p.status[b] = status;
    }
}


// ---------------------------------------------------------------------------
// KKT backward, closed-form part (mpc/lqr_step.py:346-404) in the same 4-problems-per-wave layout:
// given the solution tau* = (x*, u*) and the KKT solve's dtau = (dx, du) [= the LQR step on (C, -r, F)
// with the active controls pinned, :328-340], one backward pass over t emits
//   dC_t = -0.5 (dtau tau' + tau dtau'),  dc_t = -dtau                                 (:346-353)
//   lambda_t  = Cxx x + Cxu u + c_x + F_x' lambda_{t+1}                               (:355-369)
//   dlambda_t = Cxx dx + Cxu du - r_x + F_x' dlambda_{t+1}                            (:371-385)
//   dF_t = -(dlambda_{t+1} tau' + lambda_{t+1} dtau'),  df_t = -dlambda_{t+1}          (:387-400)
//   dx_init = -dlambda_0                                                               (:404)
// Lane j computes ROW j of dC (symmetric) and, j < 12, row j of dF: 64 contiguous bytes per lane.
// HBM-bound: 3,952 B per problem-step in and out (C, F read once; dC, dF written once).
// ---------------------------------------------------------------------------
struct KktArgs {
    const float *dx, *du, *dl_dx;
    float *dC, *dc, *dF, *df, *dx_init;
};

#ifndef MPC_KKT16_NSTAGE
#define MPC_KKT16_NSTAGE 4
#endif
enum { K_c = 0, K_tau = 64, K_dtau = 128, K_rx = 192, KKT_STAGE = 8192, DMA_KKT = 8, KKT_NSTAGE = MPC_KKT16_NSTAGE,
       KKT_AHEAD = KKT_NSTAGE - 1 };

struct KktDma {
    const char *c_ptr[4];
    const char *f_ptr[3];
    const char *r_ptr;
    long c_step, f_step, r_step;
    bool r_active;
};

MPC_DEV void kkt_dma_init(KktDma &d, const P &p, const KktArgs &k, const Lane &L, int wave)
{
    const long B = p.B;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pbk = 4 * wave + q < p.B ? 4 * wave + q : p.B - 1;
        d.c_ptr[q] = (const char *)(p.C + (long)pbk * p.C_sb) + 16 * src_granule_C(L.lane, 0);
    }
    d.c_step = 4 * p.C_st;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int G = 64 * q + L.lane;
        const int slot = G / 48, gi = G - 48 * slot;
        const int pbk = 4 * wave + slot < p.B ? 4 * wave + slot : p.B - 1;
        d.f_ptr[q] = p.T > 1 ? (const char *)(p.F + (long)pbk * p.F_sb) + 16 * src_granule_cols(gi, slot) : (const char *)p.C;
    }
    d.f_step = 4 * p.F_st;
    // record granules: 0-3 c | 4-6 x* | 7 u* | 8-10 dx | 11 du | 12-14 dl_dx
    const int gi = L.lane & 15;
    const long pb = L.pb;
    d.r_active = gi < 15;
    const char *q = (const char *)p.c;
    long st = 0;
    if (gi < 4) { q = (const char *)(p.c + pb * p.c_sb + 4 * gi); st = 4 * p.c_st; }
    else if (gi < 7) { q = (const char *)(p.cur_x + pb * 12 + 4 * (gi - 4)); st = 4 * B * 12; }
    else if (gi == 7) { q = (const char *)(p.cur_u + pb * 4); st = 4 * B * 4; }
    else if (gi < 11) { q = (const char *)(k.dx + pb * 12 + 4 * (gi - 8)); st = 4 * B * 12; }
    else if (gi == 11) { q = (const char *)(k.du + pb * 4); st = 4 * B * 4; }
    else if (gi < 15) { q = (const char *)(k.dl_dx + pb * 12 + 4 * (gi - 12)); st = 4 * B * 12; }
    d.r_ptr = q;
    d.r_step = st;
}

MPC_DEV void kkt_stage_issue(const P &p, const KktDma &d, int t, int slot)
{
    const unsigned base = (unsigned)slot * KKT_STAGE;
    const long tl = t;
    const long tf = t < p.T - 1 ? t : (p.T > 1 ? p.T - 2 : 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) wv::dma16_once(d.c_ptr[q] + tl * d.c_step, base + SC + 1024 * q);
#pragma unroll
    for (int q = 0; q < 3; ++q) wv::dma16_once(d.f_ptr[q] + (p.T > 1 ? tf * d.f_step : 0), base + SF + 1024 * q);
    wv::dma16_if(d.r_active, d.r_ptr + tl * d.r_step, base + SR);
}

MPC_DEV void kkt_wave(const P &p, const KktArgs &k)
{
    const int lane = wv::lane();
    const int wave = wv::problem();
    if (4 * wave >= p.B) return;
    Lane L;
    lane_init(L, lane, wave, p.B, false);          // (rows of C only: the plain layout)
    const int T = p.T;
    KktDma d;
    kkt_dma_init(d, p, k, L, wave);
    float lam = 0.f, dlam = 0.f;          // lambda_{t+1}[j], dlambda_{t+1}[j]  (state lanes)
#pragma unroll
    for (int i = 0; i < KKT_AHEAD; ++i) kkt_stage_issue(p, d, T - 1 - i >= 0 ? T - 1 - i : 0, i);
    for (int k0 = 0; k0 < T; k0 += KKT_NSTAGE) {
#pragma unroll
        for (int i = 0; i < KKT_NSTAGE; ++i) {
            const int t = T - 1 - (k0 + i);
            if (t >= 0) {
                wv::dma_wait<(KKT_AHEAD - 1) * DMA_KKT>();
                const unsigned base = (unsigned)i * KKT_STAGE;
                float Cr[16], Fc[12];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = wv::lds_f32x4(base + L.aCq[q]);
                    Cr[4 * q] = v[0]; Cr[4 * q + 1] = v[1]; Cr[4 * q + 2] = v[2]; Cr[4 * q + 3] = v[3];
                }
                const bool have = t < T - 1;
                if (have) {
#pragma unroll
                    for (int m = 0; m < 12; ++m) Fc[m] = wv::lds_f32(base + ((m & 1) ? L.aFo : L.aFe) + 64 * m);
                } else {
#pragma unroll
                    for (int m = 0; m < 12; ++m) Fc[m] = 0.f;
                }
                const float cj = wv::lds_f32(base + L.aRec + K_c);
                const float tj = wv::lds_f32(base + L.aRec + K_tau);
                const float dj = wv::lds_f32(base + L.aRec + K_dtau);
                const float rj = wv::lds_f32(base + SR + L.p * 256 + K_rx + 4 * (L.j < 12 ? L.j : 11));
                kkt_stage_issue(p, d, t - KKT_AHEAD >= 0 ? t - KKT_AHEAD : 0, (i + KKT_AHEAD) % KKT_NSTAGE);

                const long tb = (long)t * p.B + L.pb;
                // dF_t, df_t from the costates of t+1.  Lane j holds COLUMN j (regs = rows): every register is then one
                // row segment of 16 consecutive floats per problem and goes out as a fully coalesced dword store.
                if (have) {
                    float col[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    outer_acc(col, -dlam, tj);            // -(dlam tau' + lam dtau')[r][j]
                    outer_acc(col, -lam, dj);
                    if (L.live) {
                        float *dst = k.dF + tb * 192 + L.j;
#pragma unroll
                        for (int r = 0; r < 12; ++r) wv::store_f32_grad(dst + 16 * r, col[r]);
                        if (k.df && L.j < 12) wv::store_f32_grad(k.df + tb * 12 + L.j, -dlam);
                    }
                }
                // dC_t, dc_t: -0.5 (dtau tau' + tau dtau') is symmetric, row j = column j
                {
                    float col[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    outer_acc(col, tj, -0.5f * dj);
                    outer_acc(col, dj, -0.5f * tj);
                    if (L.live) {
                        float *dst = k.dC + tb * 256 + L.j;
#pragma unroll
                        for (int r = 0; r < 16; ++r) wv::store_f32_grad(dst + 16 * r, col[r]);
                        wv::store_f32_grad(k.dc + tb * 16 + L.j, -dj);
                    }
                }
                // costate recursions (rows 0..11 of C; F_x = first 12 columns of F)
                float ln = cj, dn = L.j < 12 ? -rj : 0.f;
                wv::dot_bcast16(ln, tj, Cr);
                wv::dot_bcast16(dn, dj, Cr);
                if (have) {
                    wv::dot_bcast12(ln, lam, Fc);
                    wv::dot_bcast12(dn, dlam, Fc);
                }
                lam = ln;
                dlam = dn;
            }
        }
    }
    wv::dma_wait<0>();
    if (L.live && L.j < 12) k.dx_init[(long)L.pb * 12 + L.j] = -dlam;
}


// ---------------------------------------------------------------------------
// The WHOLE KKT backward in one launch (round 3): LQRStepFn.backward, mpc/lqr_step.py:312-407.
//
// The reference (and the three-launch path: mpc_lqr_kkt_prepare, mpc_lqr_step, mpc_lqr_kkt_grads) solves the KKT
// system as one more LQR step on (C, -r, F) from the zero nominal with the active controls pinned (:328-340) and then
// walks the horizon backwards once more for the two costates (:355-385).  Streamed that way C and F are read three
// times.  Two facts remove the third walk and the kernels in between:
//   * lambda_t = C_x tau*_t + c_x + F_x' lambda_{t+1} (:355-369) involves nothing the nested solve produces: it rides
//     along with the Riccati sweep, which has C_t and F_t staged anyway;
//   * dlambda_t (:371-385) is the costate of the nested LQR problem along ITS OWN optimal trajectory, and for a
//     trajectory that follows the sweep's policy the costate is the gradient of the cost-to-go:
//         dlambda_t = V_t dx_t + v_t
//     (substitute dlambda_{t+1} = V_{t+1} F dtau + v_{t+1} into the recursion: [C + F'V_{t+1}F]_x dtau + q_x, and with
//     du = K dx + alpha k, Qux + Quu K = 0 on the free controls and K = 0 on the pinned ones that is
//     V_t dx_t + v_t - (1 - alpha) Qxu k_t.  alpha < 1 -- a full step that does not decrease the nested cost, i.e. a C
//     that is not positive semi-definite along the sweep -- therefore leaves dlambda_t = V_t dx_t + v_t + (1 - alpha) g_t
//     with g_t = F_x' g_{t+1} - Qxu k_t, a third 12-vector recursion of the sweep that does not depend on alpha).
// Pass 1 (t = T-1 .. 0): the sweep with c_back = -r (the nominal is zero), gains parked in the register file like mode
//     0 of the step kernel, the pinned set taken straight from u* and the bounds (:316-326, no mask array), lambda_t
//     and g_t beside it; V_t (packed upper triangle, 78 floats), v_t, lambda_t and g_t go to the workspace: 384 + 96 B
//     per problem-step against the 1,792 B of reading C_t and F_t again.
// Pass 2 (t = 0 .. T-1): the rollout dtau_t, dx_{t+1} = F dtau, dlambda_{t+1} = V_{t+1} dx_{t+1} + v_{t+1}, and with
//     them every gradient of the timestep: dC_t, dc_t, dF_t, df_t (:346-353, 387-400) as coalesced row segments.
// HBM traffic per problem-step: C, F, F again, the small vectors and 2 x 480 B of (V, v, lambda, g) in; dC, dF, dc, df
// out -- 1.12 GB per launch at the headline shape against 1.30 GB in three launches, and no launch in between.
// The line search of the nested step (:176-179, defaults decay 0.2, 10 trials) is decided from the sweep's predicted
// change of the cost, (2 alpha - alpha^2) w0, like the lean rollout of the 32/8 kernel.
// T <= RG_STEPS, symmetric C (the caller's promise MPC_OPT_C_SYMMETRIC: capi.hip takes the three-launch route
// without it).
// ---------------------------------------------------------------------------
struct KktFusedArgs {
    const float *dl_dx, *dl_du;     // [T,B,12], [T,B,4]
    float *dC, *dc, *dF, *df, *dx_init;
    float *dx_out, *du_out;         // the KKT solve's own (dx, du), optional
    float *vws;                     // workspace: [T,B,96] packed V | v, then [T,B,24] lambda | g
    float decay;                    // linesearch_decay of the nested step (0.2)
    int max_ls;                     // its max_linesearch_iter (10)
};
// Rings: the fused kernel is built in the compilation with the deep staging array (36 KiB).  Pass 1 stages exactly like
// the step kernel's sweep (four slots, the DMA three timesteps ahead, handed to the arithmetic in four parts); pass 2 has
// six of its smaller stages in flight, reads a stage into registers one timestep BEFORE it is worked on and issues a
// timestep's stores right behind that read: vector stores sit in the same vmcnt queue as the stage DMAs, a counted wait
// can only count the loads (reads and writes complete out of order with respect to each other), so a wait pays for
// every store still in flight -- behind the read they have a timestep of arithmetic to land in.
enum { KF_VBLK = 96 };
// LONG (round 4): T > RG_STEPS -- the gains of the whole horizon do not fit the 256 accumulation registers; pass 1 stores the
// record [T,B,64] behind (lambda | g) in the workspace like mode 3 of the step kernel, pass 2's stage carries it as a seventh
// DMA instruction (+ 1 KiB: five slots in the same staging memory instead of six).  T <= 64 is the kernel of round 3, untouched.
// The padded instantiation (round 6): F by 12 dword gathers, tau* by one dword instruction into a block of its own behind the stage
// (TOFF: lane l = word l), lambda | g and (V | v) out of the workspace as before -- 16 (17) instructions a stage, and vmcnt's six
// bits then cap the ring at five slots.  The kernel is built in a compilation of its own (csrc/Makefile, lqr_dpp16_padkkt.o) whose
// staging array has the deep ring's 36 KiB (MPC_KF_LDS_BYTES) although its sweep runs on two slots like every padded sweep.
#ifndef MPC_KF_LDS_BYTES
#define MPC_KF_LDS_BYTES ((int)LDS_TOTAL)
#endif
// GST: the padded instantiation's gradient blocks leave through LDS (behind the ring): a lane holds a COLUMN of dC_t / dF_t, the caller's
// block has rows of n = n_state + n_ctrl floats at no particular alignment -- a dword per lane and row was 23-28 store instructions a
// timestep on rows that straddle cache lines (10/3: 123 of the backward's 301 us; 12/2: 126 of 319).  The four problems' blocks of a
// timestep are ONE contiguous run of 4 n^2 (4 n_state n) floats in the caller's array: the lanes park their columns in LDS (rows of
// 20 dwords: conflict-free ds_write_b128), gather them back in the run's order and store 16 bytes a lane.
template <bool LONG> struct KfP2 {
    enum { TOFF = LONG ? 6656 : 5632, STAGE = TOFF + (MPC_DPP16_PADK ? 256 : 0), DMA = MPC_DPP16_PADK ? (LONG ? 17 : 16) : (LONG ? 7 : 6),
           GST = MPC_DPP16_PADK ? 5120 : 0,
           FIT = ((int)(MPC_KF_LDS_BYTES) - GST) / STAGE, CAP = 63 / DMA + 2,          // (AHEAD - 1) * DMA < 64
           SLOTS0 = FIT >= 6 ? 6 : FIT, SLOTS = SLOTS0 < CAP ? SLOTS0 : CAP, AHEAD = SLOTS - 1, GOFF = 5632 };
};
enum { KF_P2_STAGE = KfP2<false>::STAGE, KF_P2_SLOTS = KfP2<false>::SLOTS };
// row offset of row i in the packed upper triangle of a symmetric 12 x 12: entries (i, j >= i) at tri_off(i) + j - i
MPC_DEV int tri_off(int i) { return 12 * i - (i * (i - 1)) / 2; }

struct KfStage1 {            // pass 1: what a timestep reads out of its stage
    float Cc[16], Fc[12];
    float rj, tj, cx;
    float lo[4], hi[4];
};
struct KfStage2 {            // pass 2
    float Fr[16], Vn[12];
    float tj, l1, g1, v1;
};

template <bool MASKED, bool LONG = false>
MPC_DEV void kkt_fused_wave(const P &p, const KktFusedArgs &k)
{
    enum { KF_P2_STAGE = KfP2<LONG>::STAGE, KF_P2_SLOTS = KfP2<LONG>::SLOTS, KF_P2_AHEAD = KfP2<LONG>::AHEAD, KF_P2_DMA = KfP2<LONG>::DMA };
    static_assert((KF_P2_AHEAD - 1) * KF_P2_DMA < 64, "vmcnt is 6 bits");
    const int lane = wv::lane();
    const int wave = wv::problem();
    if (4 * wave >= p.B) return;
    Lane L;
    lane_init(L, lane, wave, p.B, false);
    const int T = p.T;
    const long B = p.B;
    const int gi = lane & 15;
    const long pb = L.pb;
    const bool xs_lane = L.j < 12;
    const int jx = xs_lane ? L.j : 11;
    // the caller's arrays by their true shape (the padded instantiation; 12 / 4 / 16 in the exact kernel)
    const int ns_o = PADK ? p.ns : 12, nc_o = PADK ? p.nc : 4, n_o = ns_o + nc_o;
    if (PADK) {
        L.ovalid = L.isu ? L.a < nc_o : L.j < ns_o;
#pragma unroll
        for (int a = 0; a < 4; ++a) L.padd[a] = (L.j == 12 + a && a >= nc_o) ? 1.f : 0.f;
    }

    // ---- pass 1: Riccati sweep of the nested problem + lambda + g ------------------------------------------------
    // The step kernel's own sweep staging (Dma / stage_issue / Feed: running pointers, one M0 per stage, the DMA handed
    // to the arithmetic in four parts, C with the read-once policy); only the record's sources differ:
    // granules 0-2 dl_dx, 3 dl_du, 4-6 x*, 7 u*, 8-10 c_x, 12 lo, 13 hi (tensor bounds), the rest aliased to x*
    Dma d;
#ifdef MPC_DPP16_PAD
    {
        // the sweep's own gathers of C and F (dma_seek; the launcher sets c_symmetric: the plain row layout), and the record's words
        // from THIS pass's sources: lane l = word l of a problem's 64: granule l >> 2, entry l & 3
        dma_seek<0, false, false>(d, p, L, wave);
        const long t0 = T - 1;
        const int g4 = lane >> 2, e = lane & 3;
        const bool tb = MASKED && p.bound_mode == MPC_BOUND_TENSOR;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long pbk = 4 * wave + kk < p.B ? 4 * wave + kk : p.B - 1;
            const char *q = (const char *)p.cur_x;
            long st = 0;
            bool act = false;
            if (g4 < 3) {
                const int i = 4 * g4 + e;
                if (i < ns_o) { q = (const char *)(k.dl_dx + pbk * ns_o + i); st = 4 * B * ns_o; act = true; }
            } else if (g4 == 3) {
                if (e < nc_o) { q = (const char *)(k.dl_du + pbk * nc_o + e); st = 4 * B * nc_o; act = true; }
            } else if (g4 < 7) {
                const int i = 4 * (g4 - 4) + e;
                if (i < ns_o) { q = (const char *)(p.cur_x + pbk * ns_o + i); st = 4 * B * ns_o; act = true; }
            } else if (g4 == 7) {
                if (e < nc_o) { q = (const char *)(p.cur_u + pbk * nc_o + e); st = 4 * B * nc_o; act = true; }
            } else if (g4 < 11) {
                const int i = 4 * (g4 - 8) + e;
                if (i < ns_o) { q = (const char *)(p.c + pbk * p.c_sb + i); st = 4 * p.c_st; act = true; }
            } else if (tb && (g4 == 12 || g4 == 13)) {
                if (e < nc_o) { q = (const char *)((g4 == 12 ? p.lo : p.hi) + pbk * nc_o + e); st = 4 * B * nc_o; act = true; }
            }
            d.rq[kk] = q + t0 * st - (3072 + 256 * kk);
            d.rq_step[kk] = st;
            d.rq_act[kk] = act;
            d.rq_isf[kk] = false;
        }
    }
    pad_clear(lane, MPC_KF_LDS_BYTES);
#else
    {
        const long t0 = T - 1, tf0 = T < 2 ? 0 : T - 2;
        d.c_step = 4 * p.C_st;
        d.f_step = T > 1 ? 4 * p.F_st : 0;
        d.g_step = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pbk = 4 * wave + q < p.B ? 4 * wave + q : p.B - 1;
            d.c_ptr[q] = (const char *)(p.C + (long)pbk * p.C_sb) + 16 * src_granule_C(lane, 0) - (1024 * q - 4096) + t0 * d.c_step;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int G = 64 * q + lane;
            const int slot = G / 48, g2 = G - 48 * slot;
            const int pbk = 4 * wave + slot < p.B ? 4 * wave + slot : p.B - 1;
            const char *fb = T > 1 ? (const char *)(p.F + (long)pbk * p.F_sb) : (const char *)p.C;
            d.f_ptr[q] = fb + (T > 1 ? 16 * src_granule_cols(g2, slot) : 0) - 1024 * q + tf0 * d.f_step;
        }
        const bool tb = MASKED && p.bound_mode == MPC_BOUND_TENSOR;
        const char *q = (const char *)(p.cur_x + pb * 12 + 4 * (gi % 3));
        long st = 4 * B * 12;
        if (gi < 3) { q = (const char *)(k.dl_dx + pb * 12 + 4 * gi); }
        else if (gi == 3) { q = (const char *)(k.dl_du + pb * 4); st = 4 * B * 4; }
        else if (gi < 7) { q = (const char *)(p.cur_x + pb * 12 + 4 * (gi - 4)); }
        else if (gi == 7) { q = (const char *)(p.cur_u + pb * 4); st = 4 * B * 4; }
        else if (gi < 11) { q = (const char *)(p.c + pb * p.c_sb + 4 * (gi - 8)); st = 4 * p.c_st; }
        else if (tb && gi == 12) { q = (const char *)(p.lo + pb * 4); st = 4 * B * 4; }
        else if (tb && gi == 13) { q = (const char *)(p.hi + pb * 4); st = 4 * B * 4; }
        d.r_ptr = q - 3072 + t0 * st;
        d.r_step = st;
        d.r_step_nof = st;
        d.g_ptr = d.g2_ptr = d.r_ptr;
    }
#endif

    float Vc[12], vv = 0.f, lam = 0.f, gv = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) Vc[i] = 0.f;
    double w0 = 0.0;
    int status = 0;
    Gains<!LONG> G;
    // LONG: this lane's 16 bytes of the gain record [T,B,16,4] behind (V | v) [T,B,96] and (lambda | g) [T,B,24] in the workspace
    float *const gws = k.vws + (long)T * B * (KF_VBLK + 24);
    float *gp = gws + ((long)(T - 1) * B + pb) * 64 + 4 * L.j;
    // this lane's element (., j) of the packed (V | v) block and of (lambda | g), at t = T-1, stepping back a timestep a trip
    float *vp = k.vws + ((long)(T - 1) * B + pb) * KF_VBLK + L.j;
    float *lp = k.vws + (long)T * B * KF_VBLK + ((long)(T - 1) * B + pb) * 24 + L.j;
    const long vstep = B * KF_VBLK, lstep = B * 24;

#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
        const int ti = T - 1 - i;
        if (ti >= 0) {
            stage_issue<0, false, false>(d, stage_mid<0, false, false>(i));
            stage_move<0, false, false>(d, ti <= T - 2);
        }
    }
    for (int k0 = 0; k0 < T; k0 += NSTAGE) {
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            const int t = T - 1 - (k0 + i);
            if (t >= 0) {
                const bool last = t == T - 1;
                if (t >= AHEAD - 1) wv::dma_wait<(AHEAD - 1) * DMA_SWEEP>();
                else wv::dma_wait<0>();
                KfStage1 s1;
                {
                    const unsigned base = (unsigned)i * STAGE_BYTES;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = wv::lds_f32x4(base + L.aCq[q]);
                        s1.Cc[4 * q] = v[0]; s1.Cc[4 * q + 1] = v[1]; s1.Cc[4 * q + 2] = v[2]; s1.Cc[4 * q + 3] = v[3];
                    }
#pragma unroll
                    for (int m = 0; m < 12; ++m) s1.Fc[m] = wv::lds_f32(base + ((m & 1) ? L.aFo : L.aFe) + 64 * m);
                    s1.rj = wv::lds_f32(base + L.aRec);                                        // (dl_dx | dl_du)[j]
                    s1.tj = wv::lds_f32(base + L.aRec + 64);                                   // tau*[j]
                    s1.cx = wv::lds_f32(base + SR + L.p * 256 + 128 + 4 * jx);                 // c[j], j < 12
#pragma unroll
                    for (int a = 0; a < 4; ++a) { s1.lo[a] = 0.f; s1.hi[a] = 0.f; }
                    if (MASKED) {
                        if (p.bound_mode == MPC_BOUND_TENSOR) {
                            const f32x4 l = wv::lds_f32x4(base + SR + L.p * 256 + R_lo);
                            const f32x4 h = wv::lds_f32x4(base + SR + L.p * 256 + R_hi);
#pragma unroll
                            for (int a = 0; a < 4; ++a) { s1.lo[a] = l[a]; s1.hi[a] = h[a]; }
                        } else {
#pragma unroll
                            for (int a = 0; a < 4; ++a) { s1.lo[a] = p.lo_s; s1.hi[a] = p.hi_s; }
                        }
                    }
                }
                Feed<0, false, false> feed = {d, stage_mid<0, false, false>((i + AHEAD) % NSTAGE), t >= AHEAD, true};
                feed.template part<0>();

                // lambda_t = C_x tau* + c_x + F_x' lambda_{t+1}   (:355-369; rows 0..11 of C, F_x = first 12 columns)
                float ln = s1.cx;
                wv::dot_bcast16(ln, s1.tj, s1.Cc);
                if (!last) wv::dot_bcast12(ln, lam, s1.Fc);

                // the nested problem's sweep step: c_back = C 0 - r = -r   (:328-340, :52-160)
                float Q[16];
                float q = -s1.rj;
#pragma unroll
                for (int r = 0; r < 16; ++r) Q[r] = s1.Cc[r];
                if (!last) {
                    float Y[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    wv::sched_fence();
#pragma unroll
                    for (int m = 0; m < 12; ++m) outer_acc(Y, Vc[m], s1.Fc[m]);
                    wv::sched_fence();
                    feed.template part<1>();
                    wv::sched_fence();
#pragma unroll
                    for (int m = 0; m < 12; ++m) outer_acc(Q, s1.Fc[m], Y[m]);
                    wv::sched_fence();
                    feed.template part<2>();
                    wv::sched_fence();
                    wv::dot_bcast12(q, vv, s1.Fc);
                } else {
                    feed.template part<1>();
                    feed.template part<2>();
                }
                if (PADK) {
                    // a control beyond n_ctrl: H = 1, q = 0, no reach -- its dtau stays at zero
#pragma unroll
                    for (int a = 0; a < 4; ++a) Q[12 + a] += L.padd[a];
                }
                Sym4 S;
                S.s00 = wv::bcast<12>(Q[12]); S.s01 = wv::bcast<13>(Q[12]); S.s02 = wv::bcast<14>(Q[12]); S.s03 = wv::bcast<15>(Q[12]);
                S.s11 = wv::bcast<13>(Q[13]); S.s12 = wv::bcast<14>(Q[13]); S.s13 = wv::bcast<15>(Q[13]);
                S.s22 = wv::bcast<14>(Q[14]); S.s23 = wv::bcast<15>(Q[14]);
                S.s33 = wv::bcast<15>(Q[15]);
                float qu[4];
                qu[0] = wv::bcast<12>(q); qu[1] = wv::bcast<13>(q); qu[2] = wv::bcast<14>(q); qu[3] = wv::bcast<15>(q);
                bool fr[4] = {true, true, true, true};
                Ldl4 f;
                if (MASKED) {
                    // :316-326 a control sitting on a bound (to 1e-8) is pinned in the nested solve
                    const float us[4] = {wv::bcast<12>(s1.tj), wv::bcast<13>(s1.tj), wv::bcast<14>(s1.tj), wv::bcast<15>(s1.tj)};
#pragma unroll
                    for (int a = 0; a < 4; ++a) fr[a] = !(fabsf(us[a] - s1.lo[a]) <= 1e-8f || fabsf(us[a] - s1.hi[a]) <= 1e-8f);
                    ldl4<true>(f, S, fr, 0.f);
                } else {
                    float sing = 0.f;
                    ldl4<false, true>(f, S, fr, 0.f, &sing);
                    if (sing != 0.f) status |= MPC_ST_QUU_SINGULAR;
                }
                const bool j12 = L.j == 12;
                float K[4];
                {
                    float rhs[4], y[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) rhs[a] = sel(j12, qu[a], Q[12 + a]);
                    if (MASKED) {
                        ldl4_solve(f, fr[0] ? rhs[0] : 0.f, fr[1] ? rhs[1] : 0.f, fr[2] ? rhs[2] : 0.f, fr[3] ? rhs[3] : 0.f, y);
#pragma unroll
                        for (int a = 0; a < 4; ++a) K[a] = fr[a] ? -y[a] : 0.f;
                    } else {
                        ldl4_solve(f, rhs[0], rhs[1], rhs[2], rhs[3], y);
#pragma unroll
                        for (int a = 0; a < 4; ++a) K[a] = -y[a];
                    }
                }
                // V = Qxx + Qxu K, v = qx + Qxu k  (:155-158: the K'(Qux + Quu K) terms vanish -- free rows of the bracket
                // are zero, pinned rows of K are)
                float Vn[12];
#pragma unroll
                for (int r = 0; r < 12; ++r) Vn[r] = Q[r];
                wv::sched_fence();
                feed.template part<3>();
                wv::sched_fence();
#pragma unroll
                for (int a = 0; a < 4; ++a) outer_acc(Vn, Q[12 + a], K[a]);
                wv::sched_fence();
                float vn = q;
                wv::fmac_bcast_settled<12>(vn, K[0], Q[12]); wv::fmac_bcast_settled<12>(vn, K[1], Q[13]);
                wv::fmac_bcast_settled<12>(vn, K[2], Q[14]); wv::fmac_bcast_settled<12>(vn, K[3], Q[15]);
                {
                    float w = 0.f;
#pragma unroll
                    for (int a = 0; a < 4; ++a) w = fmaf(wv::bcast<12>(K[a]), 0.5f * qu[a], w);
                    w0 += (double)w;
                }
                // g_t = F_x' g_{t+1} - Qxu k_t (the header): Qxu k_t is what v just took on
                float gn = q - vn;
                if (!last) wv::dot_bcast12(gn, gv, s1.Fc);
                if (LONG) {
                    // (rows of a partial last wave repeat problem B-1: only the live row stores its record, like V, v, lambda below)
                    if (L.live) wv::store_f32x4(gp, f32x4{K[0], K[1], K[2], K[3]});
                    gp -= B * 64;
                } else {
                    gain_put(G, t, f32x4{K[0], K[1], K[2], K[3]});
                }
#pragma unroll
                for (int r = 0; r < 12; ++r) Vc[r] = Vn[r];
                vv = vn;
                lam = ln;
                gv = gn;
                // V_t (upper triangle: register r of lane j is V[r][j], kept for j >= r), v_t, lambda_t, g_t -> workspace
#ifdef MPC_KF_SKIP           // diagnostic builds (tools/ab_kkt_phases.sh): 1 no workspace stores, 2 no gradient stores, 4 pass 1 only
                if (xs_lane && !(MPC_KF_SKIP & 1)) {
#else
                if (xs_lane) {
#endif
#pragma unroll
                    for (int r = 0; r < 12; ++r)
                        if (L.j >= r) wv::store_f32_out(vp + (tri_off(r) - r), Vn[r]);
                    wv::store_f32_out(vp + 78, vn);
                    wv::store_f32_out(lp, ln);
                    wv::store_f32_out(lp + 12, gn);
                }
                vp -= vstep;
                lp -= lstep;
            }
        }
    }
    wv::dma_wait<0>();
    // the nested step's line search (:176-179, 247): cost(alpha) - cost(0) = (2 alpha - alpha^2) w0 along the sweep's
    // direction; the first trial that does not make it worse, else the last one
    float alpha = 1.f;
    {
        const float w0f = (float)w0;
        for (int trial = 1; trial < k.max_ls; ++trial) {
            const bool worse = (2.f * alpha - alpha * alpha) * w0f > 0.f;
            alpha = worse ? alpha * k.decay : alpha;
        }
    }
    // dx_init = -dlambda_0 = -(V_0 0 + v_0 + (1 - alpha) g_0)   (:404)
    if (L.live && xs_lane && (!PADK || L.ovalid)) k.dx_init[pb * ns_o + L.j] = -fmaf(1.f - alpha, gv, vv);
    wv::fence_own_stores();            // (V, v, lambda, g) come back through the DMA
#ifdef MPC_KF_SKIP
    if (MPC_KF_SKIP & 4) return;
#endif

    // ---- pass 2: rollout of the nested solve + every gradient of the timestep -------------------------------------
    // stage (5.5 KiB): F (row order) | record: granules 4-6 x*, 7 u*, 8-10 lambda_{t+1}, 11-13 g_{t+1} | (V | v)_{t+1} x 4
    const char *f2_ptr[3], *r2_ptr, *v2_ptr[2];
    long r2_step;
    bool r2_active, v2_active;
#ifdef MPC_DPP16_PAD
    // F in the rollout's row-read order by the rollout's own gathers (dma_seek: Fb / foff of a ROLL pass, standing on t = 0); tau*
    // by one dword instruction: lane l = entry l & 15 of problem slot l >> 4
    Dma d2;
    dma_seek<0, true, false>(d2, p, L, wave);
    const char *const t2_ptr = L.isu ? (const char *)(p.cur_u + pb * nc_o + (L.ovalid ? L.a : 0)) : (const char *)(p.cur_x + pb * ns_o + (L.ovalid ? L.j : 0));
    const long t2_step = L.isu ? 4 * B * nc_o : 4 * B * ns_o;
    pad_clear(lane, MPC_KF_LDS_BYTES);
#endif
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int Gq = 64 * q + lane;
        const int slot = Gq / 48, g2 = Gq - 48 * slot;
        const int pbk = 4 * wave + slot < p.B ? 4 * wave + slot : p.B - 1;
        f2_ptr[q] = T > 1 ? (const char *)(p.F + (long)pbk * p.F_sb) + 16 * src_granule_rows(g2) : (const char *)p.C;
    }
    {
        r2_active = gi >= (PADK ? 8 : 4) && gi < 14;
        const char *q = (const char *)p.cur_x;
        long st = 0;
        if (gi >= 4 && gi < 7) { q = (const char *)(p.cur_x + pb * 12 + 4 * (gi - 4)); st = 4 * B * 12; }
        else if (gi == 7) { q = (const char *)(p.cur_u + pb * 4); st = 4 * B * 4; }
        else if (gi >= 8 && gi < 14) { q = (const char *)(k.vws + (long)T * B * KF_VBLK + pb * 24 + 4 * (gi - 8)); st = 4 * B * 24; }
        r2_ptr = q;
        r2_step = st;
        // (V | v): 24 granules per problem, 96 per wave: instruction 0 lanes 0..63, instruction 1 lanes 0..31
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int Gq = 64 * q2 + lane;
            const int slot = Gq / 24, g2 = Gq - 24 * slot;
            const int pbk = slot < 4 ? (4 * wave + slot < p.B ? 4 * wave + slot : p.B - 1) : (int)pb;
            v2_ptr[q2] = (const char *)(k.vws + (long)pbk * KF_VBLK) + 16 * (slot < 4 ? g2 : 0);
        }
        v2_active = lane < 32;
    }
    const long v2_step = 4 * B * KF_VBLK, f_step = T > 1 ? 4 * p.F_st : 0;
    auto issue2 = [&](int t, int slot) {
        // stage t holds F_t, tau*_t and lambda / g / V / v of t+1 (t = T-1: copies of T-1 that nothing looks at)
        t = t < T ? t : T - 1;
        const unsigned sb = (unsigned)slot * KF_P2_STAGE;
        const long tf = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
        const long t1 = t < T - 1 ? t + 1 : t;
#ifdef MPC_DPP16_PAD
        {
            const long fo = tf * f_step;
#define MPC_PAD_F2(kk) wv::dma_buf_at<256 * (kk), PAD_BIAS>(d2.Fb[(kk) / 3] + fo, d2.fbytes, d2.foff[kk], sb)
            MPC_PAD_F2(0); MPC_PAD_F2(1); MPC_PAD_F2(2); MPC_PAD_F2(3); MPC_PAD_F2(4); MPC_PAD_F2(5);
            MPC_PAD_F2(6); MPC_PAD_F2(7); MPC_PAD_F2(8); MPC_PAD_F2(9); MPC_PAD_F2(10); MPC_PAD_F2(11);
#undef MPC_PAD_F2
            wv::dma4_if(L.ovalid, t2_ptr + (long)t * t2_step, sb + (unsigned)KfP2<LONG>::TOFF);
        }
#else
#pragma unroll
        for (int q = 0; q < 3; ++q) wv::dma16_once(f2_ptr[q] + tf * f_step, sb + 1024 * q);
#endif
        wv::dma16_if(r2_active, r2_ptr + (gi >= 8 ? t1 : (long)t) * r2_step, sb + 3072);
        wv::dma16_once(v2_ptr[0] + t1 * v2_step, sb + 4096);
        wv::dma16_if(v2_active, v2_ptr[1] + t1 * v2_step, sb + 4096 + 1024);
        if (LONG) wv::dma16_once((const char *)(gws + ((long)t * B + pb) * 64 + 4 * L.j), sb + (unsigned)KfP2<LONG>::GOFF);
    };
    // index of V[i][j] in the packed block, for this lane's column j
    int vidx[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) vidx[i] = 4 * (L.p * KF_VBLK + (i <= jx ? tri_off(i) + jx - i : tri_off(jx) + i - jx));
    auto read2 = [&](KfStage2 &s, int slot) {
        const int sb = slot * KF_P2_STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = wv::lds_f32x4((unsigned)(sb - SF + L.aFq[q]));
            s.Fr[4 * q] = v[0]; s.Fr[4 * q + 1] = v[1]; s.Fr[4 * q + 2] = v[2]; s.Fr[4 * q + 3] = v[3];
        }
        s.tj = PADK ? wv::lds_f32((unsigned)(sb + (int)KfP2<LONG>::TOFF + 4 * lane))             // tau*_t[j] (its own block, see KfP2)
                    : wv::lds_f32((unsigned)(sb + 3072 - SR + L.aRec + 64));
        s.l1 = wv::lds_f32((unsigned)(sb + 3072 + L.p * 256 + 128 + 4 * jx));            // lambda_{t+1}[j]
        s.g1 = wv::lds_f32((unsigned)(sb + 3072 + L.p * 256 + 176 + 4 * jx));            // g_{t+1}[j]
#pragma unroll
        for (int r = 0; r < 12; ++r) s.Vn[r] = wv::lds_f32((unsigned)(sb + 4096 + vidx[r]));
        s.v1 = wv::lds_f32((unsigned)(sb + 4096 + 4 * (L.p * KF_VBLK + 78 + jx)));
    };
    float xs = 0.f;                                            // dx_t[j]: dx_0 = 0
    // Rows r, r+1 of a gradient block leave as ONE 8-byte store per lane: the even lane of a pair writes (row r, columns
    // j, j+1), the odd one (row r+1, columns j-1, j) -- a problem's 16 lanes then cover the two rows, 128 contiguous bytes,
    // where a dword per lane and row wrote 64-byte halves of a cache line with twice the instructions (28 per timestep).
    const bool odd = (L.j & 1) != 0;
    float *dC_p = k.dC + pb * 256 + (L.j & ~1) + (odd ? 16 : 0), *dF_p = k.dF + pb * 192 + (L.j & ~1) + (odd ? 16 : 0);
    float *dc_p = k.dc + pb * 16 + L.j;
    float *df_p = k.df ? k.df + pb * 12 + L.j : nullptr;
    float *so_p = (k.dx_out && k.du_out) ? (L.isu ? k.du_out + pb * 4 + L.a : k.dx_out + pb * 12 + L.j) : nullptr;
    long so_step = L.isu ? B * 4 : B * 12;
    // the padded instantiation: the caller's blocks are [n, n] / [n_state, n] with n = n_state + n_ctrl -- this lane's COLUMN of them,
    // a dword per row (a row of the padding is skipped: wave-uniform), rows of n floats instead of 16
    const int aj = PADK ? (L.isu ? ns_o + L.a : L.j) : L.j;          // this lane's entry of the caller's tau (when ovalid)
#ifdef MPC_DPP16_PAD
    // the staged form (KfP2::GST): this lane's 16 bytes of the wave's run of dC_t (dF_t) in store instruction i are the run's floats
    // 4 (64 i + lane) .. + 3; each is element (row, column) of problem slot q of the run, parked at gst + 1280 q + 80 pad(column) + 4 pad(row)
    // [dC_t is symmetric: the lane that holds column j holds row j], pad(a) = a for a state, 12 + (a - n_state) for a control
    const bool full4 = 4 * wave + 3 < p.B;            // (a ragged last wave keeps the dword stores: its run is shorter)
    constexpr unsigned gst = (unsigned)(KF_P2_SLOTS * KF_P2_STAGE);
    const int n2C = n_o * n_o, n2F = ns_o * n_o;      // 16-byte stores of a full wave's run: 4 n^2 / 4
    unsigned gaC[4][4], gaF[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int e = 4 * (64 * i + lane);
        int q = e / n2C, rem = e - q * n2C, ar = rem / n_o, ac = rem - ar * n_o;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int pr = ar < ns_o ? ar : 12 + (ar - ns_o), pc = ac < ns_o ? ac : 12 + (ac - ns_o);
            gaC[i][v] = gst + (unsigned)(1280 * (q & 3) + 80 * (pc & 15) + 4 * (pr & 15));
            if (++ac == n_o) { ac = 0; if (++ar == n_o) { ar = 0; ++q; } }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int e = 4 * (64 * i + lane);
        int q = e / n2F, rem = e - q * n2F, ar = rem / n_o, ac = rem - ar * n_o;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int pc = ac < ns_o ? ac : 12 + (ac - ns_o);
            gaF[i][v] = gst + (unsigned)(1280 * (q & 3) + 80 * (pc & 15) + 4 * (ar & 15));
            if (++ac == n_o) { ac = 0; if (++ar == ns_o) { ar = 0; ++q; } }
        }
    }
    const unsigned gw = gst + (unsigned)(1280 * L.p + 80 * L.j);        // where this lane parks its column
    float *dCs_p = k.dC + (long)(4 * wave) * n2C + 4 * lane, *dFs_p = k.dF + (long)(4 * wave) * n2F + 4 * lane;
#endif
    if (PADK) {
        dC_p = k.dC + pb * (n_o * n_o) + (L.ovalid ? aj : 0);
        dF_p = k.dF + pb * (ns_o * n_o) + (L.ovalid ? aj : 0);
        dc_p = k.dc + pb * n_o + (L.ovalid ? aj : 0);
        df_p = k.df ? k.df + pb * ns_o + (L.ovalid ? L.j : 0) : nullptr;
        so_p = (k.dx_out && k.du_out) ? (L.isu ? k.du_out + pb * nc_o + (L.ovalid ? L.a : 0) : k.dx_out + pb * ns_o + (L.ovalid ? L.j : 0)) : nullptr;
        so_step = L.isu ? B * nc_o : B * ns_o;
    }
#pragma unroll
    for (int i = 0; i < KF_P2_AHEAD; ++i) issue2(i, i);
    KfStage2 s2;
    wv::dma_wait<(KF_P2_AHEAD - 1) * KF_P2_DMA>();
    read2(s2, 0);
    for (int t0 = 0; t0 < T; t0 += KF_P2_SLOTS) {
#pragma unroll
        for (int i = 0; i < KF_P2_SLOTS; ++i) {
            const int t = t0 + i;
            if (t < T) {
                const bool have = t < T - 1;
                issue2(t + KF_P2_AHEAD, (i + KF_P2_AHEAD) % KF_P2_SLOTS);
                const f32x4 rec = LONG ? wv::lds_f32x4((unsigned)(i * KF_P2_STAGE + (int)KfP2<LONG>::GOFF + 16 * lane)) : gain_get(G, t);
                // du = K dx + alpha k  (:192; pinned controls have zero rows of K and k: they stay at 0)
                const float dxj = L.isu ? 0.f : xs;
                const float mult = L.j == 12 ? alpha : dxj;
                const float du = wv::quad_sums(rec[0] * mult, rec[1] * mult, rec[2] * mult, rec[3] * mult, L.j);
                const float dj = L.isu ? du : xs;                           // dtau_t[j]
                float xn = 0.f, dl1 = 0.f;
                float colF[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (have) {
                    wv::dot_bcast16(xn, dj, s2.Fr);                        // dx_{t+1} = F dtau   (f = None, :333)
                    dl1 = fmaf(1.f - alpha, s2.g1, s2.v1);
                    wv::dot_bcast12(dl1, xn, s2.Vn);                       // dlambda_{t+1} = V dx + v + (1 - alpha) g
                    // dF_t = -(dlambda_{t+1} tau' + lambda_{t+1} dtau'), df_t = -dlambda_{t+1}   (:387-400)
                    outer_acc(colF, -dl1, s2.tj);
                    outer_acc(colF, -s2.l1, dj);
                }
                // dC_t = -0.5 (dtau tau' + tau dtau'), dc_t = -dtau   (:346-353)
                float colC[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                outer_acc(colC, s2.tj, -0.5f * dj);
                outer_acc(colC, dj, -0.5f * s2.tj);
                xs = xn;
                // the next stage has had this timestep to land: read it, THEN send this timestep's gradients off
                wv::dma_wait<(KF_P2_AHEAD - 1) * KF_P2_DMA>();
                read2(s2, (i + 1) % KF_P2_SLOTS);
#ifdef MPC_DPP16_PAD
#ifdef MPC_KF_SKIP
                const bool skip_ = (MPC_KF_SKIP & 2) && t != T - 1;
#else
                const bool skip_ = false;
#endif
                if (full4 && !skip_) {
                    // dC_t: park the columns, gather the run, 16 bytes a lane (DS instructions of a wave execute in order)
                    wv::lds_sync();
#pragma unroll
                    for (int q = 0; q < 4; ++q) wv::lds_store_f32x4(gw + 16u * q, f32x4{colC[4 * q], colC[4 * q + 1], colC[4 * q + 2], colC[4 * q + 3]});
                    wv::lds_sync();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (64 * i < n2C && 64 * i + lane < n2C)
                            wv::store_f32x4_grad(dCs_p + 256 * i, f32x4{wv::lds_f32(gaC[i][0]), wv::lds_f32(gaC[i][1]), wv::lds_f32(gaC[i][2]), wv::lds_f32(gaC[i][3])});
                    }
                    if (have) {
                        wv::lds_sync();
#pragma unroll
                        for (int q = 0; q < 3; ++q) wv::lds_store_f32x4(gw + 16u * q, f32x4{colF[4 * q], colF[4 * q + 1], colF[4 * q + 2], colF[4 * q + 3]});
                        wv::lds_sync();
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            if (64 * i < n2F && 64 * i + lane < n2F)
                                wv::store_f32x4_grad(dFs_p + 256 * i, f32x4{wv::lds_f32(gaF[i][0]), wv::lds_f32(gaF[i][1]), wv::lds_f32(gaF[i][2]), wv::lds_f32(gaF[i][3])});
                        }
                    }
                }
                if (L.live && L.ovalid && !skip_) {
                    if (so_p) wv::store_f32_grad(so_p, dj);
                    if (have) {
                        if (!full4) {
#pragma unroll
                            for (int r = 0; r < 12; ++r)
                                if (r < ns_o) wv::store_f32_grad(dF_p + r * n_o, colF[r]);
                        }
                        if (df_p && xs_lane) wv::store_f32_grad(df_p, -dl1);
                    }
                    if (!full4) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ar = pad_tau(r, ns_o, nc_o);
                            if (ar >= 0) wv::store_f32_grad(dC_p + ar * n_o, colC[r]);
                        }
                    }
                    wv::store_f32_grad(dc_p, -dj);
                }
                dCs_p += B * n2C;
                dFs_p += B * n2F;
                if (so_p) so_p += so_step;
                dC_p += B * (n_o * n_o);
                dF_p += B * (ns_o * n_o);
                dc_p += B * n_o;
                if (df_p) df_p += B * ns_o;
#else
                // (the exchange with the neighbouring lane happens for every lane: rows of a ragged last wave only skip the stores)
                float pF0[6], pF1[6], pC0[8], pC1[8];
#pragma unroll
                for (int r = 0; r < 12; r += 2) {
                    const float n0 = wv::swap1(colF[r]), n1 = wv::swap1(colF[r + 1]);
                    pF0[r / 2] = odd ? n1 : colF[r];
                    pF1[r / 2] = odd ? colF[r + 1] : n0;
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float n0 = wv::swap1(colC[r]), n1 = wv::swap1(colC[r + 1]);
                    pC0[r / 2] = odd ? n1 : colC[r];
                    pC1[r / 2] = odd ? colC[r + 1] : n0;
                }
#ifdef MPC_KF_SKIP
                if (L.live && (!(MPC_KF_SKIP & 2) || t == T - 1)) {
#else
                if (L.live) {
#endif
                    if (so_p) wv::store_f32_grad(so_p, dj);
                    if (have) {
#pragma unroll
                        for (int r = 0; r < 6; ++r) wv::store_f32x2_out(dF_p + 32 * r, pF0[r], pF1[r]);
                        if (df_p && xs_lane) wv::store_f32_grad(df_p, -dl1);
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) wv::store_f32x2_out(dC_p + 32 * r, pC0[r], pC1[r]);
                    wv::store_f32_grad(dc_p, -dj);
                }
                if (so_p) so_p += so_step;
                dC_p += B * 256;
                dF_p += B * 192;
                dc_p += B * 16;
                if (df_p) df_p += B * 12;
#endif
            }
        }
    }
    wv::dma_wait<0>();
    if (L.j == 0 && L.live && p.status) p.status[pb] = status;
}

}  // namespace dpp16
}  // namespace mpclqr
