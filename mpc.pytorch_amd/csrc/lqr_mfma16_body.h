// lqr_mfma16_body.h -- the fused LQR step for n_state <= 12, n_ctrl <= 4, fp32:
// one 64-lane wavefront owns one problem for the whole horizon, every matrix
// product of the step is a v_mfma_f32_16x16x4_f32 (exact fp32), and NO data ever
// moves between lanes except a handful of readlanes of the 4x4 control block.
//
// This file is written against a tiny "wave" interface (namespace wv: lane(),
// mfma(), readlane(), dma16(), lds_f32(), ...).  lqr_mfma16.hip binds it to the
// gfx950 builtins; tests/emu/emu_mfma16.cpp binds it to a host-side 64-fiber
// lockstep emulator so the very same source is parity-tested on a CPU-only box.
//
// What it replaces in locuslab/mpc.pytorch (one launch instead of ~4,000 ATen ops):
//   sweep_step      mpc/lqr_step.py:284-296 (c_back) + :52-160 (lqr_backward)
//   pnqp4           mpc/pnqp.py:5-82 (the in-sweep n_ctrl-dimensional box QP)
//   rollout_*       mpc/lqr_step.py:164-261 (lqr_forward), mpc/util.py:129-153
//
// ---------------------------------------------------------------------------
// Layout.  MFMA 16x16x4 f32: lane l = 16 g + j holds  A[i=j][k=g],  B[k=g][j],
// and D[4g+r][j] in accumulator register r.  The 16 variables of tau = [x;u] are
// given "slots" rho = 0..15 so that u_a sits at rho = 4a (register 0 of lane
// group a) and x_m at rho = 4 (m/3) + 1 + m%3 (registers 1..3):
//
//   * a symmetric matrix held in D layout IS its own A operand (block kb = register
//     kb, contraction slot (g,kb) <-> rho = 4g+kb), so  Y = V F  and  Q = C + F'Y
//     chain through registers with no shuffles; F is loaded once as lane (g,j) ->
//     F[x-slot(g,kb)][var j] and serves as B of the first and A of the second product.
//   * the four u-rows of Q land in register 0 spread over the four lane groups,
//     which is exactly the B operand of K = -Quu^-1 [Qux|qu] and the A operand of
//     Qxu K; K and M = Qux + Quu K come out of their MFMAs in that same layout.
//   * column j = 0 carries the linear terms: Q'[:,0] = q, K'[:,0] = k, M'[:,0] =
//     qu + Quu k, V'[:,0] = v.  (Q[u0][u0], the one entry that loses its place,
//     is rebuilt from 3 FMAs + 4 readlanes.)
//   * the rollout keeps its state as column 0 of the B operand, [u;x] in slot order,
//     so u' = K dx and x+ = F [x';u'] are MFMAs whose output is already the next
//     step's operand.  When the full step (alpha = 1) is rejected, ONE more pass
//     propagates all the remaining line-search candidates at once -- column j is the
//     trial with alpha = decay^j (mpc/lqr_step.py:176-252) -- and a last pass stores
//     the accepted one.
//
// Staging.  Every per-timestep block (C_t 1 KiB, F_t 768 B, and one ~0.5 KiB record
// of the small vectors c, x, u, f, bounds and the gains) is DMA'd HBM -> LDS with
// global_load_lds (16 B per lane, no VGPRs held while in flight) into a 4-slot ring
// per wavefront, three timesteps ahead of its use; the step then picks its MFMA
// operands out of LDS in whatever lane order the layout above wants (a per-lane
// address picks "vector entry in column 0, matrix entry elsewhere", or a word of
// zeros, so no select instructions are spent on it).  The waits are counted
// s_waitcnt vmcnt(N) on the number of NEWER DMA instructions (loads return in order;
// stores in flight only make the wait more conservative).
//
// C is read as the symmetric matrix the reference documents it to be
// (mpc/mpc.py:61-68; its own delta-space gradient C tau + c, :294, is only a
// gradient for symmetric C).
// ---------------------------------------------------------------------------
// (round 5) Included once per element type, like lqr_small_math.h: float / namespace mfma16 by default, double / mfma16d with
// MPC_M16_F64 defined -- the float64 instantiation (v_mfma_f64_16x16x4_f64 has the lane layout of the float32 instruction; LDS offsets and DMA
// granules scale with the element size ES).
#include <math.h>
#include "lqr_params.h"
#include "lqr_small_math.h"
#if (defined(MPC_M16_F64) && !defined(MPC_MFMA16_BODY_F64)) || (!defined(MPC_M16_F64) && !defined(MPC_MFMA16_BODY_F32))
#undef MPC_M16_REAL
#undef MPC_M16_NS
#ifdef MPC_M16_F64
#define MPC_MFMA16_BODY_F64
#define MPC_M16_REAL double
#define MPC_M16_NS mfma16d
#else
#define MPC_MFMA16_BODY_F32
#define MPC_M16_REAL float
#define MPC_M16_NS mfma16
#endif

// 1: start the box QP of timestep t from the solution of timestep t+1 like the reference (rounds 1-5; kept for the A/B)
#ifndef MPC_MFMA16_QP_WARM
#define MPC_MFMA16_QP_WARM 0
#endif

namespace mpclqr {
namespace MPC_M16_NS {

typedef StepParams<real> P;
typedef typename wv::vec4_of<real>::type rx4;
constexpr int ES = (int)sizeof(real);        // bytes per element
constexpr int GE = 16 / ES;                  // elements per 16-byte DMA granule
// Slot numbering.  Element r of lane group g of an accumulator is row 4 g + r of the tile in v_mfma_f32_16x16x4_f32 and row
// 4 r + g in v_mfma_f64_16x16x4_f64 (measured: tools/ubench/mfma_f64_probe.hip); the A / B operand layouts are the same in both.
// Everything below speaks of the slot (g, r); only where a slot becomes a COLUMN or LANE index does its number matter.
constexpr bool SLOT_T = ES == 8;
MPC_DEV constexpr int slot_of(int g, int r) { return SLOT_T ? 4 * r + g : 4 * g + r; }
MPC_DEV int slot_group(int j) { return SLOT_T ? (j & 3) : (j >> 2); }
MPC_DEV int slot_reg(int j) { return SLOT_T ? (j >> 2) : (j & 3); }

// LDS layout (bytes): a ring of NSTAGE stages + 64 B of scratch.  The record of small vectors is
// laid out so that in the n_state = 12, n_ctrl = 4 case every piece starts on a 16-byte granule
// (one DMA lane each).
enum {
    LDS_C = 0, LDS_F = 256 * ES, LDS_V = 448 * ES,
    V_c = 0, V_tau = 16 * ES, V_f = 32 * ES, V_lo = 48 * ES, V_hi = 52 * ES, V_K = 56 * ES, V_zero = 120 * ES,
    STAGE_BYTES = 572 * ES, NSTAGE = 4,
    LDS_SCRATCH = NSTAGE * STAGE_BYTES, LDS_TOTAL = LDS_SCRATCH + 16 * ES,
    C_GRAN = 256 / GE, F_GRAN = 192 / GE,                                  // 16-byte granules of a full C / F block
    DMA_PER_STAGE_FULL = (C_GRAN + 63) / 64 + (F_GRAN + 63) / 64 + 1,      // C, F, record (float: 1 + 1 + 1, double: 2 + 2 + 1)
    DMA_PER_STAGE_MIN = 5       // padded shapes: at least C, F, c, x, u (one instruction each)
};

struct Lane {
    int lane, g, j, ja;
    bool j0, jq;          // j == 0 (vector column), (j & 3) == 0 (u column)
    int col;              // tau index of column slot j (0 if padding)
    bool colv;
    int row[4];           // control index g (r = 0) / state index 3g+r-1 (r >= 1) of row slot 4g+r
    bool rowv[4];
    bool vC[4], vT[4], vK;
    // byte offsets into an LDS stage
    int aC[4];            // LDS_C + C[rowtau r][col]                    (rows 1..3 double as F rows: +LDS_F-LDS_C)
    int aCt[4];           // LDS_C + C[col][rowtau r]: the mirror entry (the symmetry test)
    int aT[4];            // 4 * tau index of row slot r                  (+LDS_V+V_c: c; +V_tau: tau; +V_f: f)
    int aQi[4];           // column 0: c[row r];   elsewhere: C[row r][col]
    int aTb[4];           // column 0: tau[row r]; elsewhere: a word of zeros
    int aFT[4];           // LDS_F + F[x(col)][rowtau r]                  (F as the rollout's A operand)
    int aKA;              // u columns: the 16 B  Kk[ja][4g .. 4g+3];  elsewhere: zeros
    int aKAe[3];          // ... entry by entry, Kk[ja][slot (g, kb)], where the slots of a lane group are not adjacent (SLOT_T)
    int aKk;              // Kk[g][0] = k_g
    int aG;               // 4 * g  (+LDS_V+V_lo / V_hi)
    real eg[4];          // one-hot of g
    real nea[4];         // -1 at (u column, ja == a), else 0
};

MPC_DEV void lane_init(Lane &L, int lane, int ns, int nc)
{
    const int n = ns + nc;
    L.lane = lane;
    L.g = lane >> 4;
    L.j = lane & 15;
    L.j0 = L.j == 0;
    L.jq = slot_reg(L.j) == 0;
    L.ja = slot_group(L.j);
    int coltau;
    if (L.jq) { L.colv = L.ja < nc; coltau = ns + L.ja; }
    else { const int x = 3 * L.ja + slot_reg(L.j) - 1; L.colv = x < ns; coltau = x; }
    if (!L.colv) coltau = 0;
    L.col = coltau;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int tau;
        if (r == 0) { L.rowv[0] = L.g < nc; L.row[0] = L.g; tau = ns + L.g; }
        else { const int x = 3 * L.g + r - 1; L.rowv[r] = x < ns; L.row[r] = x; tau = x; }
        if (!L.rowv[r]) { L.row[r] = 0; tau = 0; }
        L.vC[r] = L.rowv[r] && L.colv;
        L.aC[r] = LDS_C + (L.vC[r] ? ES * (tau * n + coltau) : 0);
        L.aCt[r] = LDS_C + (L.vC[r] ? ES * (coltau * n + tau) : 0);
        L.aT[r] = ES * tau;
        L.aQi[r] = L.j0 ? LDS_V + V_c + L.aT[r] : L.aC[r];
        L.aTb[r] = L.j0 ? LDS_V + V_tau + L.aT[r] : LDS_V + V_zero;
        // rollout: output row = state x(j) (x columns only), contraction slot = (g, r)
        L.vT[r] = L.rowv[r] && L.colv && !L.jq;
        L.aFT[r] = LDS_F + (L.vT[r] ? ES * (coltau * n + tau) : 0);
        L.eg[r] = L.g == r ? 1.f : 0.f;
        L.nea[r] = (L.jq && L.ja == r) ? -1.f : 0.f;
    }
    L.vK = L.jq && L.ja < nc;
    L.aKA = L.vK ? LDS_V + V_K + ES * (16 * L.ja + 4 * L.g) : LDS_V + V_zero;
#pragma unroll
    for (int kb = 1; kb < 4; ++kb) L.aKAe[kb - 1] = L.vK ? LDS_V + V_K + ES * (16 * L.ja + slot_of(L.g, kb)) : LDS_V + V_zero;
    L.aKk = LDS_V + V_K + ES * 16 * L.g;
    L.aG = L.rowv[0] ? ES * L.g : 0;
}

// ---------------------------------------------------------------------------
// HBM -> LDS staging
// ---------------------------------------------------------------------------
// Source of the small-vector record for this lane (n_state = 12, n_ctrl = 4: lane = 16-byte granule):
//   granules 0-3 c | 4-6 x | 7 u | 8-10 f | 12 lo | 13 hi | 14-29 Kk (gains of the sweep)
struct VecDma {
    const char *ptr;      // address for the next timestep to be issued
    long step;            // bytes per timestep (negative in the sweep)
    bool active;
    bool is_f;            // f has T-1 entries: the last record re-reads f[T-2] (never used)
};

template <int MODE, bool ROLL>
MPC_DEV void vecdma_init(VecDma &v, const P &p, int lane, int b, int t0)
{
    v.active = false;
    v.is_f = false;
    const char *q = (const char *)p.c;
    long st = 0;
    const long B = p.B;
    // the record in ELEMENTS: 0-15 c | 16-27 x | 28-31 u | 32-43 f | 48-51 lo | 52-55 hi | 56-119 Kk; lane = 16-byte granule = GE elements
    const int e = lane * GE;
    if (e < 16) {
        v.active = true; q = (const char *)(p.c + (long)b * p.c_sb + e); st = ES * p.c_st;
    } else if (e < 28) {
        v.active = true; q = (const char *)(p.cur_x + (long)b * 12 + (e - 16)); st = (long)ES * B * 12;
    } else if (e < 32) {
        v.active = true; q = (const char *)(p.cur_u + (long)b * 4 + (e - 28)); st = (long)ES * B * 4;
    } else if (e < 44) {
        if (ROLL && p.f && p.T > 1) {
            v.active = true; v.is_f = true;
            q = (const char *)(p.f + (long)b * p.f_sb + (e - 32)); st = ES * p.f_st;
        }
    } else if (e < 48) {
    } else if (e < 56) {
        if (MODE == 2 && p.bound_mode == MPC_BOUND_TENSOR) {
            v.active = true; q = (const char *)((e < 52 ? p.lo : p.hi) + (long)b * 4 + (e < 52 ? e - 48 : e - 52)); st = (long)ES * B * 4;
        }
    } else if (e < 120) {
        if (ROLL) { v.active = true; q = (const char *)(p.Kk + (long)b * 64 + (e - 56)); st = (long)ES * B * 64; }
    }
    v.ptr = q + (long)t0 * st;
    v.step = ROLL ? st : -st;
}

// Issue the DMA of timestep t into ring slot `slot` and advance the record pointer (`advance` is
// false for the re-issues past the end of the horizon, which only keep the DMA count constant).
// Exactly DMA_PER_STAGE_FULL instructions when FULL, at least DMA_PER_STAGE_MIN otherwise.
template <bool FULL, int MODE, bool ROLL>
MPC_DEV void stage_issue(const P &p, VecDma &vd, int lane, int b, int t, int slot, bool advance)
{
    const unsigned base = (unsigned)slot * STAGE_BYTES;
    const int n = p.ns + p.nc;
    const real *Ct = p.C + (long)t * p.C_st + (long)b * p.C_sb;
    // F has T-1 entries; the last timestep re-reads a valid block nobody looks at
    const real *Ft = Ct;
    const int tf = t < p.T - 1 ? t : p.T - 2;
    if (p.T > 1) Ft = p.F + (long)tf * p.F_st + (long)b * p.F_sb;
    if (FULL) {
        const unsigned lo16 = 16u * (unsigned)lane;
#pragma unroll
        for (int i = 0; i < (C_GRAN + 63) / 64; ++i)
            if (lane + 64 * i < C_GRAN) wv::dma16r<real>((const char *)Ct + 1024 * i + lo16, base + LDS_C + 1024 * i);
#pragma unroll
        for (int i = 0; i < (F_GRAN + 63) / 64; ++i)
            if (lane + 64 * i < F_GRAN) wv::dma16r<real>((const char *)Ft + 1024 * i + lo16, base + LDS_F + 1024 * i);
        if (vd.active) {
            const char *src = vd.ptr;
            if (ROLL && vd.is_f && t >= p.T - 1) src -= vd.step;
            wv::dma16r<real>(src, base + LDS_V);
        }
        if (advance) vd.ptr += vd.step;
    } else {
        // element by element, as 4-byte words (a double is two of them): the blocks land densely packed from their LDS base
        constexpr int W = ES / 4;
        const long tb = (long)t * p.B + b;
        const int n2 = n * n * W, nf = p.T > 1 ? p.ns * n * W : 1;
        typedef const float *wp;
#pragma unroll
        for (int i = 0; i < 4 * W; ++i)
            if (i * 64 < n2) { const int e = lane + 64 * i; if (e < n2) wv::dma4r<real>((wp)Ct + e, base + LDS_C + 256 * i); }
#pragma unroll
        for (int i = 0; i < 3 * W; ++i)
            if (i * 64 < nf) { const int e = lane + 64 * i; if (e < nf) wv::dma4r<real>((wp)Ft + e, base + LDS_F + 256 * i); }
        if (lane < n * W) wv::dma4r<real>((wp)(p.c + (long)t * p.c_st + (long)b * p.c_sb) + lane, base + LDS_V + V_c);
        if (lane < p.ns * W) wv::dma4r<real>((wp)(p.cur_x + tb * p.ns) + lane, base + LDS_V + V_tau);
        if (lane < p.nc * W) wv::dma4r<real>((wp)(p.cur_u + tb * p.nc) + lane, base + LDS_V + V_tau + ES * p.ns);
        if (MODE == 2 && p.bound_mode == MPC_BOUND_TENSOR) {
            if (lane < p.nc * W) wv::dma4r<real>((wp)(p.lo + tb * p.nc) + lane, base + LDS_V + V_lo);
            if (lane < p.nc * W) wv::dma4r<real>((wp)(p.hi + tb * p.nc) + lane, base + LDS_V + V_hi);
        }
        if (ROLL) {
            if (p.f && p.T > 1 && lane < p.ns * W)
                wv::dma4r<real>((wp)(p.f + (long)tf * p.f_st + (long)b * p.f_sb) + lane, base + LDS_V + V_f);
#pragma unroll
            for (int i = 0; i < W; ++i) wv::dma4r<real>((wp)(p.Kk + tb * 64) + lane + 64 * i, base + LDS_V + V_K + 256 * i);
        }
    }
}

template <bool FULL, int NEWER_STAGES>
MPC_DEV void stage_wait()
{
    wv::dma_wait<NEWER_STAGES * (FULL ? (int)DMA_PER_STAGE_FULL : (int)DMA_PER_STAGE_MIN)>();
}

MPC_DEV int zm_load(const P &p, const Lane &L, int b, int t)
{
    return L.rowv[0] ? (int)p.zero_mask[((long)t * p.B + b) * p.nc + L.row[0]] : 0;
}

// ---------------------------------------------------------------------------
// Sweep
// ---------------------------------------------------------------------------
struct SwStage {
    real C[4];       // C_t in D layout (register r = row slot 4g+r)
    real Qi[4];      // the same with column 0 replaced by c_t (row layout): accumulator init of Q'
    real Tb[4];      // nominal tau_t in row layout in column 0, zeros elsewhere (B operand of C tau)
    real F[3];       // F_t rows x-slot(g,kb), kb = 1..3, at column var(j)
    real lo, hi;     // control bounds of u_g (tensor or scalar mode)
    int zm;           // u_zero_I of u_g
};

// MODE: 0 = unconstrained, 1 = unconstrained with u_zero_I (the KKT backward's solve), 2 = box bounds
template <bool FULL, int MODE>
MPC_DEV void sw_read(SwStage &s, const P &p, const Lane &L, int t, int slot, int zm, real &asym, real &cmax)
{
    const unsigned base = (unsigned)slot * STAGE_BYTES;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const real v = wv::lds_r<real>(base + L.aC[r]);
        s.C[r] = FULL ? v : sel(L.vC[r], v, 0.f);
        // the D-layout C serves as its own A operand below -- true for a symmetric C only; the reference takes C as it
        // is (mpc/lqr_step.py:68, 294).  Keep the largest |C - C'| entry and the largest |C| (-> MPC_ST_C_ASYMMETRIC).
        if (!p.c_symmetric) {
            const real vt = wv::lds_r<real>(base + L.aCt[r]);
            asym = rmax(asym, rabs(v - vt));       // (padding lanes read one entry twice: 0)
            cmax = rmax(cmax, rabs(s.C[r]));
        }
        const real w = wv::lds_r<real>(base + L.aQi[r]);
        s.Qi[r] = FULL ? w : sel(L.j0 ? L.rowv[r] : L.vC[r], w, 0.f);
        const real u = wv::lds_r<real>(base + L.aTb[r]);
        s.Tb[r] = FULL ? u : sel(L.rowv[r], u, 0.f);
    }
    if (t < p.T - 1) {
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) {
            const real v = wv::lds_r<real>(base + (LDS_F - LDS_C) + L.aC[kb]);
            s.F[kb - 1] = FULL ? v : sel(L.vC[kb], v, 0.f);
        }
    } else {
        s.F[0] = s.F[1] = s.F[2] = 0.f;
    }
    s.lo = s.hi = 0.f;
    if (MODE == 2) {
        if (p.bound_mode == MPC_BOUND_TENSOR) {
            s.lo = wv::lds_r<real>(base + LDS_V + V_lo + L.aG);
            s.hi = wv::lds_r<real>(base + LDS_V + V_hi + L.aG);
        } else {
            s.lo = p.lo_s;
            s.hi = p.hi_s;
        }
    }
    s.zm = MODE == 1 ? zm : 0;
}

struct SwState {
    rx4 Vp;          // V' in D layout: rows/cols = x slots, column 0 = v, u slots = finite junk
    real oc;          // nominal-cost partial (lanes j == 0)
    real kprev[4];    // warm start of the next pnqp = k_{t+1}   (mpc/lqr_step.py:137,141)
    int warm;
    int qp_total;
    int status;
    real asym, cmax;  // symmetry test of C: largest |C[i][j] - C[j][i]|, largest |C[i][j]| seen by this lane
};

template <bool FULL, int MODE>
MPC_DEV void sweep_step(const P &p, const Lane &L, const SwStage &s, SwState &st, int b, int t)
{
    const bool last = (t == p.T - 1);
    const rx4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // c_back - c = C tau  (mpc/lqr_step.py:289-295): contraction over all 16 slots, column 0 only
    rx4 CB = zero4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) CB = wv::mfma(s.C[kb], s.Tb[kb], CB);

    // Y = V_{t+1} F_t   (:65-70), x slots only (kb = 1..3)
    rx4 Y = zero4;
    rx4 Q = {s.Qi[0], s.Qi[1], s.Qi[2], s.Qi[3]};
    real q00p = 0.f;
    if (!last) {
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) Y = wv::mfma(st.Vp[kb], s.F[kb - 1], Y);
        // Q = C + F'Y ; column 0: F'v  (Y's column 0 is swapped for v = V'[:,0])
        q00p = rfma(s.F[2], Y[3], rfma(s.F[1], Y[2], s.F[0] * Y[1]));
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) Q = wv::mfma(s.F[kb - 1], sel(L.j0, st.Vp[kb], Y[kb]), Q);
    }
    // nominal cost 0.5 tau'C tau + c'tau (util.get_cost, mpc/lqr_step.py:169) off the same product:
    // Tb is tau in column 0 and zero elsewhere, Qi is c there
    {
        real ocl = st.oc;
#pragma unroll
        for (int r = 0; r < 4; ++r) ocl = rfma(s.Tb[r], rfma((real)0.5, CB[r], s.Qi[r]), ocl);
        st.oc = ocl;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Q[r] += CB[r];          // CB is exactly 0 outside column 0

    // ---- the 4x4 control block, wave-uniform -------------------------------------------------
    const real U = Q[0];        // lane (g,j): Q[u_g][var j]; lane (g,0): qu[g]
    Sym4 S;
    // Quu[a][b] sits in lane group a at the column slot of u_b
#define MPC_QUU(a, b) wv::readlane(U, 16 * (a) + slot_of(b, 0))
    S.s01 = MPC_QUU(0, 1); S.s02 = MPC_QUU(0, 2); S.s03 = MPC_QUU(0, 3);
    S.s11 = MPC_QUU(1, 1); S.s12 = MPC_QUU(1, 2); S.s13 = MPC_QUU(1, 3);
    S.s22 = MPC_QUU(2, 2); S.s23 = MPC_QUU(2, 3);
    S.s33 = MPC_QUU(3, 3);
#undef MPC_QUU
    S.s00 = wv::readlane(s.C[0], 0);
    if (!last)
        S.s00 += (wv::readlane(q00p, 0) + wv::readlane(q00p, 16)) + (wv::readlane(q00p, 32) + wv::readlane(q00p, 48));

    bool valid[4], fr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) valid[a] = a < p.nc;
    Ldl4 f;
    real kq[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 2) {
        // :84-94 unconstrained / :99-127 masked (u_zero_I): masked rows and columns drop out
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            fr[a] = valid[a];
            if (MODE == 1) fr[a] = fr[a] && (wv::readlane_i(s.zm, 16 * a) == 0);
        }
        real sing = 0.f;
        ldl4<(!FULL || MODE == 1), MODE == 0>(f, S, fr, 0.f, &sing);          // MODE 0: the reference's pinverse (pivot_inv)
        if (MODE == 0 && sing != 0.f) st.status |= MPC_ST_QUU_SINGULAR;
    } else {
        // :128-141 box constraints in delta space
        real qu[4], lb[4], ub[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            qu[a] = wv::readlane(U, 16 * a);
            const real u = wv::readlane(s.Tb[0], 16 * a);
            real l = wv::readlane(s.lo, 16 * a) - u;
            real h = wv::readlane(s.hi, 16 * a) - u;
            if (p.has_delta) {                                      // :132-134
                if (l < -p.delta_u) l = -p.delta_u;
                if (h > p.delta_u) h = p.delta_u;
            }
            lb[a] = valid[a] ? l : 0.f;
            ub[a] = valid[a] ? h : 0.f;
        }
        // (float64 keeps the reference's start, k of timestep t+1: pnqp returns the iterate whose Newton step is shorter than 1e-4
        // WITHOUT taking that step (mpc/pnqp.py:56-59), so where the reference ends -- to within 1e-4 -- does depend on where it
        // started, and the float64 instantiation is held to the reference at 1e-9; in float32 that 1e-4 is inside the tolerance
        // and inside what the reference's own float32 and float64 runs differ by)
        constexpr bool REF_START = MPC_MFMA16_QP_WARM || sizeof(real) == 8;
        if (!REF_START || !st.warm) {
            // cold start x = -H^-1 q (mpc/pnqp.py:14-19) at EVERY timestep whose Quu is positive definite (round 6, see
            // lqr_dpp16_body.h: a better guess of the active set than the reference's start k_{t+1} by a whole trip per QP; a convex
            // QP has one minimiser whatever the start).  Not positive definite: the reference's start.
            ldl4<!FULL>(f, S, valid, 0.f);
            real y[4];
            ldl4_solve(f, valid[0] ? qu[0] : 0.f, valid[1] ? qu[1] : 0.f, valid[2] ? qu[2] : 0.f,
                       valid[3] ? qu[3] : 0.f, y);
            const real imin = rmin(rmin(f.i0, f.i1), rmin(f.i2, f.i3)), imax = rmax(rmax(f.i0, f.i1), rmax(f.i2, f.i3));
            const bool cold = !st.warm || (imin > (real)0 && imax < (real)3.0e38);
#pragma unroll
            for (int a = 0; a < 4; ++a) kq[a] = cold ? -y[a] : st.kprev[a];      // (an invalid row is an identity row with a zero right-hand side: y = 0)
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) kq[a] = st.kprev[a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) kq[a] = eclampf(kq[a], lb[a], ub[a]);
        bool conv = false;
        const int it = pnqp4(S, qu, lb, ub, valid, p.pnqp_iter, kq, fr, f, conv);
        st.qp_total += 1 + it;                                      // :140
        if (!conv) st.status |= MPC_ST_PNQP_UNCONVERGED;
        st.warm = 1;
#pragma unroll
        for (int a = 0; a < 4; ++a) st.kprev[a] = kq[a];
    }

    // A operand of K' = -H_free^-1 [Qux | qu]: lane (i = 4a, g) holds -inv[a][g]
    real Ainv;
    {
        real y[4];
        ldl4_solve(f, L.eg[0], L.eg[1], L.eg[2], L.eg[3], y);
        if (!FULL || MODE != 0) {
            // rows / columns outside the free set are exactly zero (the factor holds identity there)
            const real fg = dot4(L.eg, fr[0] ? 1.f : 0.f, fr[1] ? 1.f : 0.f, fr[2] ? 1.f : 0.f, fr[3] ? 1.f : 0.f);
#pragma unroll
            for (int a = 0; a < 4; ++a) y[a] = fr[a] ? y[a] * fg : 0.f;
        }
        Ainv = dot4(L.nea, y[0], y[1], y[2], y[3]);
    }
    const rx4 Kacc = wv::mfma(Ainv, U, zero4);
    real Kp = Kacc[0];           // lane (g,j): K[g][var j]; lane (g,0): k[g]
    rx4 Vn;
    if (MODE == 0) {
        // K = -Quu^-1 Qux makes Qux + Quu K vanish: V = Qxx + Qxu K, v = qx + Qxu k   (:155-158)
        Vn = wv::mfma(U, Kp, Q);
    } else {
        if (MODE == 2) {
            const real kg = dot4(L.eg, kq[0], kq[1], kq[2], kq[3]);
            Kp = sel(L.j0, kg, Kp);   // k is the QP solution itself (:136-141)
        }
        // A operand Quu (unmasked, :155-158): lane (i = 4a, g) holds Quu[a][g] = U at the same lane,
        // except column 0 where U carries qu: there Quu[0][g] is needed.
        const real s0g = dot4(L.eg, S.s00, S.s01, S.s02, S.s03);
        const real Aquu = L.jq ? sel(L.j0, s0g, U) : 0.f;
        rx4 Min = zero4;
        Min[0] = U;
        const rx4 Macc = wv::mfma(Aquu, Kp, Min);
        const real Mp = Macc[0];     // lane (g,j): (Qux + Quu K)[g][var j]; lane (g,0): qu + Quu k
        // :155-158  V = Qxx + Qxu K + K'(Qux + Quu K),  v = qx + Qxu k + K'(qu + Quu k)
        Vn = wv::mfma(U, Kp, Q);
        Vn = wv::mfma(Kp, Mp, Vn);
    }
    st.Vp = Vn;

    // gains: the wave's own record Kk[t][b][16 g + j] (read back by the rollout), and, when the caller
    // asked for them, K [T,B,nc,ns] / k [T,B,nc] in the reference's layout
    {
        const long tb = (long)t * p.B + b;
        p.Kk[tb * 64 + L.lane] = Kp;
        if (p.K != nullptr) {
            if (L.rowv[0]) {
                if (L.j0) p.k[tb * p.nc + L.g] = Kp;
                else if (!L.jq && L.colv) p.K[(tb * p.nc + L.g) * p.ns + L.col] = Kp;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Rollout
// ---------------------------------------------------------------------------
struct RoStage {
    real C[4];       // C_t, D layout
    real FA[4];      // F_t as A operand: lane (i = x slot, g), block kb -> F[x(i)][var(4g+kb)]
    real KA[3];      // K_t as A operand: lane (i = 4a, g), block kb -> K[a][x-slot(g,kb)]
    real crow[4];
    real xbar[3];    // nominal x_t, row layout
    real frow[3];    // f_t, row layout
    real ubar, kk;   // nominal u_g, feed-forward k_g
    real lo, hi;
    int zm;
};

template <bool FULL, int MODE>
MPC_DEV void ro_read(RoStage &s, const P &p, const Lane &L, int t, int slot, int zm)
{
    const unsigned base = (unsigned)slot * STAGE_BYTES;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const real v = wv::lds_r<real>(base + L.aC[r]);
        s.C[r] = FULL ? v : sel(L.vC[r], v, 0.f);
        const real w = wv::lds_r<real>(base + LDS_V + V_c + L.aT[r]);
        s.crow[r] = FULL ? w : sel(L.rowv[r], w, 0.f);
    }
    if (t < p.T - 1) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const real v = wv::lds_r<real>(base + L.aFT[kb]);
            s.FA[kb] = FULL ? v : sel(L.vT[kb], v, 0.f);      // u columns feed output rows nobody reads
        }
        if (p.f) {
#pragma unroll
            for (int kb = 1; kb < 4; ++kb) {
                const real v = wv::lds_r<real>(base + LDS_V + V_f + L.aT[kb]);
                s.frow[kb - 1] = FULL ? v : sel(L.rowv[kb], v, 0.f);
            }
        } else {
            s.frow[0] = s.frow[1] = s.frow[2] = 0.f;
        }
    } else {
        s.FA[0] = s.FA[1] = s.FA[2] = s.FA[3] = 0.f;
        s.frow[0] = s.frow[1] = s.frow[2] = 0.f;
    }
    if (SLOT_T) {
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) {
            const real v = wv::lds_r<real>(base + L.aKAe[kb - 1]);
            s.KA[kb - 1] = FULL ? v : sel(L.rowv[kb], v, 0.f);
        }
    } else {
        const rx4 kv = wv::lds_r4<real>(base + L.aKA);
        s.KA[0] = FULL ? kv[1] : sel(L.rowv[1], kv[1], 0.f);
        s.KA[1] = FULL ? kv[2] : sel(L.rowv[2], kv[2], 0.f);
        s.KA[2] = FULL ? kv[3] : sel(L.rowv[3], kv[3], 0.f);
    }
#pragma unroll
    for (int kb = 1; kb < 4; ++kb) {
        const real v = wv::lds_r<real>(base + LDS_V + V_tau + L.aT[kb]);
        s.xbar[kb - 1] = FULL ? v : sel(L.rowv[kb], v, 0.f);
    }
    {
        const real v = wv::lds_r<real>(base + LDS_V + V_tau + L.aT[0]);
        s.ubar = FULL ? v : sel(L.rowv[0], v, 0.f);
        const real w = wv::lds_r<real>(base + L.aKk);
        s.kk = FULL ? w : sel(L.rowv[0], w, 0.f);
    }
    s.lo = s.hi = 0.f;
    if (MODE == 2) {
        if (p.bound_mode == MPC_BOUND_TENSOR) {
            s.lo = wv::lds_r<real>(base + LDS_V + V_lo + L.aG);
            s.hi = wv::lds_r<real>(base + LDS_V + V_hi + L.aG);
        } else {
            s.lo = p.lo_s;
            s.hi = p.hi_s;
        }
    }
    s.zm = zm;
}

struct RoState {
    real xrow[3];    // x'_t of trial j (column j), row layout
    real cost;       // partial of this lane (group)
    real du2;
    real alpha;      // step of trial j
};

// MULTI = false: one trial (every column carries it; column 0 is the one read), stage cost on the
//                vector ALU;
// MULTI = true : sixteen trials, alpha_j = decay^j in column j, stage cost = one more 16x16x16 product.
template <bool FULL, int MODE, bool MULTI>
MPC_DEV void rollout_step(const P &p, const Lane &L, const RoStage &s, RoState &st, int b, int t)
{
    const rx4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool last = (t == p.T - 1);
    // new_u = K dx + u + alpha k   (mpc/lqr_step.py:192)
    rx4 Uacc = zero4;
#pragma unroll
    for (int kb = 1; kb < 4; ++kb) Uacc = wv::mfma(s.KA[kb - 1], st.xrow[kb - 1] - s.xbar[kb - 1], Uacc);
    // x_{t+1} = F [x;u] + f  (:216-222): the x part does not wait for u
    rx4 Xacc = zero4;
    Xacc[1] = s.frow[0]; Xacc[2] = s.frow[1]; Xacc[3] = s.frow[2];
    if (!last) {
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) Xacc = wv::mfma(s.FA[kb], st.xrow[kb - 1], Xacc);
    }
    real un = Uacc[0] + rfma(st.alpha, s.kk, s.ubar);
    if (MODE != 0 && s.zm) un = 0.f;                                 // :197-198
    if (MODE == 2) {                                                 // :200-213
        real l = s.lo, h = s.hi;
        if (p.has_delta) {
            const real l2 = s.ubar - p.delta_u, h2 = s.ubar + p.delta_u;
            l = (l2 < l) ? l : l2;
            h = (h2 > h) ? h : h2;
        }
        un = eclampf(un, l, h);
    }
    if (!FULL) un = sel(L.rowv[0], un, 0.f);
    if (!last) Xacc = wv::mfma(s.FA[0], un, Xacc);
    // obj_t = 0.5 tau'C tau + c'tau   (:230-232)
    if (MULTI) {
        rx4 Cacc = wv::mfma(s.C[0], un, zero4);
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) Cacc = wv::mfma(s.C[kb], st.xrow[kb - 1], Cacc);
        real ca = un * rfma((real)0.5, Cacc[0], s.crow[0]);
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) ca = rfma(st.xrow[kb - 1], rfma((real)0.5, Cacc[kb], s.crow[kb]), ca);
        st.cost += ca;
    } else {
        // tau' sits in row layout; hand it to every lane through 64 bytes of LDS: lane (g,j) needs
        // tau'[4g+r] (its four C rows -- it has those) and tau'[slot j] (its C column)
        if (L.j0) {
            if (SLOT_T) {
                wv::lds_store_r(LDS_SCRATCH + (unsigned)ES * (unsigned)slot_of(L.g, 0), un);
#pragma unroll
                for (int kb = 1; kb < 4; ++kb) wv::lds_store_r(LDS_SCRATCH + (unsigned)ES * (unsigned)slot_of(L.g, kb), st.xrow[kb - 1]);
            } else {
                wv::lds_store_r4(LDS_SCRATCH + (unsigned)(4 * ES) * (unsigned)L.g, rx4{un, st.xrow[0], st.xrow[1], st.xrow[2]});
            }
        }
        wv::lds_sync();
        const real tc = wv::lds_r<real>(LDS_SCRATCH + (unsigned)ES * (unsigned)L.j);
        wv::lds_sync();
        const real sq = rfma(s.C[3], st.xrow[2], rfma(s.C[2], st.xrow[1], rfma(s.C[1], st.xrow[0], s.C[0] * un)));
        const real lin = rfma(s.crow[3], st.xrow[2], rfma(s.crow[2], st.xrow[1], rfma(s.crow[1], st.xrow[0], s.crow[0] * un)));
        st.cost = rfma(0.5f * tc, sq, st.cost) + sel(L.j0, lin, 0.f);
        const long tb = (long)t * p.B + b;
        if (L.j0) {
            if (L.rowv[0]) p.new_u[tb * p.nc + L.row[0]] = un;
#pragma unroll
            for (int kb = 1; kb < 4; ++kb)
                if (L.rowv[kb]) p.new_x[tb * p.ns + L.row[kb]] = st.xrow[kb - 1];
        }
    }
    const real d = s.ubar - un;
    st.du2 = rfma(d, d, st.du2);
    if (!last) {
        st.xrow[0] = Xacc[1]; st.xrow[1] = Xacc[2]; st.xrow[2] = Xacc[3];
    }
}

// One pass over the horizon.  MULTI: trial j uses alpha = decay^min(j, max_ls-1), nothing is stored.
// Single: every column uses `alpha` and the trajectory is stored.  cost_j / du2_j: totals of trial j
// (single: the same number in every lane).
template <bool FULL, int MODE, bool MULTI>
MPC_DEV void rollout_pass(const P &p, const Lane &L, int b, real alpha, real &cost_j, real &du2_j)
{
    RoState st;
#pragma unroll
    for (int kb = 1; kb < 4; ++kb) st.xrow[kb - 1] = L.rowv[kb] ? p.x_init[(long)b * p.ns + L.row[kb]] : 0.f;
    st.cost = 0.f;
    st.du2 = 0.f;
    st.alpha = alpha;
    if (MULTI) {
        // alpha_j = decay^min(j, max_ls-1)    (:247)
        real a = 1.f;
        const int e = L.j < p.max_ls - 1 ? L.j : p.max_ls - 1;
        for (int i = 0; i < e; ++i) a *= p.ls_decay;
        st.alpha = a;
    }
    const int T = p.T;
    const int lane = L.lane;
    const bool use_zm = MODE != 0 && p.zero_mask != nullptr;
    VecDma vd;
    vecdma_init<MODE, true>(vd, p, lane, b, 0);
    // ring prologue: timesteps 0, 1, 2 in flight (indices past the horizon re-read the last step)
    int zq[NSTAGE] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ti = i < T ? i : T - 1;
        stage_issue<FULL, MODE, true>(p, vd, lane, b, ti, i, i + 1 < T);
        if (use_zm) zq[i] = zm_load(p, L, b, ti);
    }
    for (int t0 = 0; t0 < T; t0 += NSTAGE) {
#pragma unroll
        for (int i = 0; i < NSTAGE; ++i) {
            const int t = t0 + i;
            if (t < T) {
                stage_wait<FULL, 2>();                       // slots of t+1, t+2 may still be in flight
                RoStage s;
                ro_read<FULL, MODE>(s, p, L, t, i, zq[i]);
                const int tn = t + 3 < T ? t + 3 : T - 1;
                stage_issue<FULL, MODE, true>(p, vd, lane, b, tn, (i + 3) % NSTAGE, t + 4 < T);
                if (use_zm) zq[(i + 3) % NSTAGE] = zm_load(p, L, b, tn);
                rollout_step<FULL, MODE, MULTI>(p, L, s, st, b, t);
            }
        }
    }
    stage_wait<FULL, 0>();
    // sum the four lane groups (and, single trial, the 16 column partials of the quadratic form)
    real c = st.cost, d = st.du2;
    c += wv::shfl_xor(c, 16); d += wv::shfl_xor(d, 16);
    c += wv::shfl_xor(c, 32); d += wv::shfl_xor(d, 32);
    if (!MULTI) {
        c += wv::shfl_xor(c, 1); c += wv::shfl_xor(c, 2); c += wv::shfl_xor(c, 4); c += wv::shfl_xor(c, 8);
    }
    cost_j = c;
    du2_j = d;
}

template <bool FULL, int MODE>
MPC_DEV void step_problem(const P &p)
{
    const int lane = wv::lane();
    const int b = wv::problem();
    if (b >= p.B) return;
    Lane L;
    lane_init(L, lane, p.ns, p.nc);
    const int T = p.T;
    // the words of zeros of every ring slot, once
    if (lane < 4 * NSTAGE) wv::lds_store_r((unsigned)(lane >> 2) * STAGE_BYTES + LDS_V + V_zero + (unsigned)ES * (unsigned)(lane & 3), (real)0);
    wv::lds_sync();

    // ---- Riccati sweep, t = T-1 .. 0 ------------------------------------------------------------
    SwState ss;
    ss.Vp = rx4{0.f, 0.f, 0.f, 0.f};
    ss.oc = 0.f;
    ss.warm = 0;
    ss.qp_total = 0;
    ss.status = 0;
    ss.asym = ss.cmax = 0.f;
    ss.kprev[0] = ss.kprev[1] = ss.kprev[2] = ss.kprev[3] = 0.f;
    {
        VecDma vd;
        vecdma_init<MODE, false>(vd, p, lane, b, T - 1);
        int zq[NSTAGE] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ti = T - 1 - i >= 0 ? T - 1 - i : 0;
            stage_issue<FULL, MODE, false>(p, vd, lane, b, ti, i, T - 2 - i >= 0);
            if (MODE == 1) zq[i] = zm_load(p, L, b, ti);
        }
        for (int k0 = 0; k0 < T; k0 += NSTAGE) {
#pragma unroll
            for (int i = 0; i < NSTAGE; ++i) {
                const int t = T - 1 - (k0 + i);
                if (t >= 0) {
                    stage_wait<FULL, 2>();
                    SwStage s;
                    sw_read<FULL, MODE>(s, p, L, t, i, zq[i], ss.asym, ss.cmax);
                    const int tn = t - 3 >= 0 ? t - 3 : 0;
                    stage_issue<FULL, MODE, false>(p, vd, lane, b, tn, (i + 3) % NSTAGE, t - 4 >= 0);
                    if (MODE == 1) zq[(i + 3) % NSTAGE] = zm_load(p, L, b, tn);
                    sweep_step<FULL, MODE>(p, L, s, ss, b, t);
                }
            }
        }
        stage_wait<FULL, 0>();
    }
    const real old_cost = (wv::readlane(ss.oc, 0) + wv::readlane(ss.oc, 16)) +
                           (wv::readlane(ss.oc, 32) + wv::readlane(ss.oc, 48));
    if (!p.c_symmetric) {
        // (a tolerance, not a bit test: C = A'A out of a float32 GEMM is symmetric to rounding only)
        real m = ss.cmax;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) m = rmax(m, wv::shfl_xor(m, sh));
        ss.status |= MPC_ST_C_TESTED;
        // (float64: a C that is symmetric to float32 rounding only -- A'A out of a float32 product, cast up -- moves the results by
        // 1e-7 against the reference, which uses C as given: far above what float64 callers compare to; it is flagged, and
        // impl = 0 re-solves it the reference's way)
        if (wv::ballot(ss.asym > (real)(ES == 4 ? 1e-5 : 1e-12) * m) != 0ull) ss.status |= MPC_ST_C_ASYMMETRIC;
    }

    // the gains were written by this wave and are re-read through the DMA: drain the stores
    wv::fence_own_stores();

    // ---- line-searched rollout (mpc/lqr_step.py:164-261) ---------------------------------------
    real cost_j, du2_j;
    rollout_pass<FULL, MODE, false>(p, L, b, 1.f, cost_j, du2_j);
    real cost = wv::readlane(cost_j, 0);
    real dun2 = wv::readlane(du2_j, 0);
    const real full2 = dun2;                                        // :243-245 (alpha = 1 trial)
    real alpha = 1.f;
    if (wv::uniform(cost > old_cost) && p.max_ls > 1) {
        // the full step made it worse: run every remaining trial at once and take the first whose
        // cost did not get worse, else the last one (:176-179, 247, 252)
        rollout_pass<FULL, MODE, true>(p, L, b, 1.f, cost_j, du2_j);
        const unsigned long long okm = wv::ballot(!(cost_j > old_cost) && L.g == 0 && L.j >= 1 && L.j < p.max_ls);
        int jstar = p.max_ls - 1;
        if (okm) jstar = wv::ctz64(okm);
        jstar = wv::uniform(jstar);
        for (int i = 0; i < jstar; ++i) alpha *= p.ls_decay;
        rollout_pass<FULL, MODE, false>(p, L, b, alpha, cost_j, du2_j);
        cost = wv::readlane(cost_j, 0);
        dun2 = wv::readlane(du2_j, 0);
    }
    int status = ss.status;
    if (!(cost == cost) || rabs(cost) > (real)(ES == 4 ? 3e38 : 1e300)) status |= MPC_ST_NONFINITE;
    if (lane == 0) {
        if (p.costs) p.costs[b] = cost;
        if (p.old_costs) p.old_costs[b] = old_cost;
        if (p.full_du_norm) p.full_du_norm[b] = rsqrt_of(full2);
        if (p.alpha_du_norm) p.alpha_du_norm[b] = rsqrt_of(dun2);
        if (p.alphas) p.alphas[b] = alpha;
        if (p.qp_iters) p.qp_iters[b] = ss.qp_total;
        if (p.status) p.status[b] = status;
    }
}

}  // namespace MPC_M16_NS
}  // namespace mpclqr
#endif
