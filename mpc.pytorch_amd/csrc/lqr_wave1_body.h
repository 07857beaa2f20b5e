// lqr_wave1_body.h -- one LQR step for problems with ONE control and n_state <= 6 (the reference's pendulum and
// cart-pole iLQR, BASELINE configs 2 and 3) when the batch is far too small to fill the chip with a lane per problem:
// ONE 16-LANE ROW PER PROBLEM (four problems per wavefront), float32, the problems resident in LDS.
//
// A wavefront alone on its SIMD issues one instruction every four clocks whatever the instruction does: the
// lane-per-problem kernel (lqr_tiny_body.h) walks the horizon with ~1400 instructions per timestep (sweep + rollout,
// pendulum) and at B = 1024 .. 4096 its time IS that count times the horizon -- 54 / 97 us per pendulum / cart-pole step.
// Most of those instructions do not depend on the recursion.  Here the step is cut where its dependences are:
//   P1  lane = TIMESTEP (16 at a time): C_t, c_t, the nominal, the delta-space linear term C tau + c
//       (mpc/lqr_step.py:284-296), the nominal's cost (:169), the bounds, and F_t -- the simulator's Jacobian in closed
//       form (mpc/mpc.py:490-549) or the caller's array -- into LDS.
//   P2  the Riccati recursion (:52-160), the only part serial in t: lane j of the row owns COLUMN j of F_t, W = V F,
//       Q = C + F'W and V; every product is a DPP row broadcast fused into a multiply-add; operands come from LDS one
//       step ahead.  nc = 1: the box QP is the scalar pnqp (mpc/pnqp.py:5-82 with n = 1), solved by every lane alike.
//   P3  lane = LINE-SEARCH TRIAL (:176-179, 247: alpha = decay^g): all max_linesearch_iter <= 16 rollouts (:186-241) at
//       once, serial in t only through the dynamics (simulator :223-225 or F, f :216-222); trajectories into LDS.
//   P4  lane = (trial, slice of the horizon): the trials' costs (:230-232) from the LDS copies of C, c.
//   P5  the first trial not worse than the nominal wins (else the last), its trajectory leaves LDS.
// Written against the `wv::` lane interface: lqr_wave1.hip implements it for gfx950, tests/emu/ with lockstep fibers.
#pragma once
#include <type_traits>
#include "lqr_tiny_body.h"

#ifndef MPC_STAT
#define MPC_STAT(i)          // (tests/emu counts events through this)
#endif

namespace mpclqr {
namespace wave1 {

template <int I, int N, class F> MPC_DEV void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// LDS map of one problem, in floats.  Regions a call does not use take no room (f: only the linear true dynamics has one;
// lo / hi: only a box-constrained call; mask: only with u_zero_I), and the gains K_t, k_t overwrite C_t tau_t + c_t, which the
// recursion has consumed by then -- a cart-pole problem (T = 25) takes 9.8 KB, four waves of four problems fit a CU.
struct Layout {
    int C, c, cb, F, f, tau, K, lo, hi, mask, X, total;
};
MPC_HD Layout layout(int ns, int T, int ntrial, bool has_f, bool bounded, bool masked)
{
    const int N = ns + 1;
    Layout L;
    int o = 0;
    L.C = o, o += T * N * N;
    L.c = o, o += T * N;
    L.cb = L.K = o, o += T * N;
    L.F = o, o += T * ns * N;
    L.f = o, o += has_f ? T * ns : 0;
    L.tau = o, o += T * N;
    L.lo = o, o += bounded ? T : 0;
    L.hi = o, o += bounded ? T : 0;
    L.mask = o, o += masked ? T : 0;
    L.X = o, o += ntrial * T * N;
    L.total = o;
    return L;
}
MPC_HD Layout layout(const StepParams<float> &p)
{
    return layout(p.ns, p.T, p.max_ls, !p.env.kind && p.f, p.bound_mode != MPC_BOUND_NONE, p.zero_mask != nullptr);
}
MPC_HD bool shape_supported(const StepParams<float> &p)
{
    if (!(p.nc == 1 && p.ns >= 1 && p.ns <= 6 && p.T >= 1 && p.max_ls >= 1 && p.max_ls <= 16)) return false;
    if ((long)p.T * 49 * 16 > 160 * 1024) return false;                   // (before the sum below could overflow)
    return (long)layout(p).total * 16 <= 150 * 1024;                      // four problems per wavefront
}

// tiny::pnqp1 (mpc/pnqp.py:5-82 with n = 1) without its loops, for the case that is nearly every case: H > 0.  Then the
// iteration is over after at most two passes, and what it leaves behind can be written down directly --
//   pass 0 at x0 = clamp(warm start): clamped with the gradient pointing outward, or a Newton step below 1e-4: done, x = x0;
//   otherwise the Newton step, inside the box as it is or projected onto the bound it crosses (its Armijo ratio is
//   1 - d / (2 dx) > 1/2: the first trial of :64-76 is always accepted), and pass 1 finds nothing left to do --
// with the same operations in the same order as the loop (the free flag of pass 1 is evaluated as the loop does, rounding
// of the residual gradient included).  A divergent branch costs a lone wavefront ~50 clocks (VALU -> SALU -> exec and
// back) and the loop form had a dozen per timestep: 892 of the recursion's 2156 clocks per pendulum step.  Anything
// else (H <= 0, a residual Newton step that is not below 1e-4) sends the wavefront through the loop itself.
MPC_DEV int pnqp1_fast(float H, float q, float lb, float ub, float &x, float &Hfree, bool &is_free, int n_iter, bool &conv)
{
    const float x0 = tiny::clampr<float>(x, lb, ub);               // :23
    const float g0 = H * x0 + q;                                   // :29
    const bool fr0 = !((x0 == lb && g0 > 0) || (x0 == ub && g0 < 0));      // :32
    const float Hf0 = (fr0 ? H : 0.f) + 1e-11f;                    // :44-48
    const float dx0 = -((fr0 ? g0 : 0.f) * env_inv(Hf0));          // :50-51
    const bool done0 = !(tiny::absr<float>(dx0) >= 1e-4f);         // :56-59
    const float xs = x0 + dx0;
    const bool inside = xs >= lb && xs <= ub;
    // the projected step: xn = clamp(x0 + dx0), d = xn - x0, accepted at alpha = 1 iff its Armijo ratio exceeds 0.1 (:64-76)
    const float xc = tiny::clampr<float>(xs, lb, ub), d = xc - x0;
    const float arm = (-g0 * d - 0.5f * H * d * d) * env_inv(-g0 * d);
    const float x1 = inside ? xs : xc;
    const float g1 = H * x1 + q;
    const bool fr1 = !((x1 == lb && g1 > 0) || (x1 == ub && g1 < 0));
    const float Hf1 = (fr1 ? H : 0.f) + 1e-11f;
    const float dx1 = -((fr1 ? g1 : 0.f) * env_inv(Hf1));
    const bool done1 = !(tiny::absr<float>(dx1) >= 1e-4f);
    const bool simple = n_iter >= 2 && H > 0 && (done0 || ((inside || arm > 0.1f) && done1));
    int ret;
    if (wv::any(!simple)) {                                        // (wavefront-uniform and rare: the loop, for everybody)
        MPC_STAT(14);
        ret = tiny::pnqp1<float>(H, q, lb, ub, x, Hfree, is_free, n_iter, conv);
    } else {
        MPC_STAT(15);
        ret = done0 ? 0 : 1;
        x = done0 ? x0 : x1;
        is_free = done0 ? fr0 : fr1;
        Hfree = done0 ? Hf0 : Hf1;
        conv = true;
    }
    return ret;
}

template <int NS> MPC_DEV void step_wave(const StepParams<float> &p)
{
    constexpr int N = NS + 1;
    const int T = p.T, B = p.B, lane = wv::lane(), j = lane & 15, row = lane >> 4;
    int b = wv::problem() * 4 + row;
    const bool active = b < B;
    if (!active) b = B - 1;                            // idle rows shadow the last problem, storing nothing
    const int ntr = p.max_ls;
    Layout L = layout(p);
    const bool has_f = !p.env.kind && p.f, masked_call = p.zero_mask != nullptr;
    {                                                  // this row's region of the LDS
        const int rb = row * L.total;
        L.C += rb, L.c += rb, L.cb += rb, L.F += rb, L.f += rb, L.tau += rb, L.K += rb, L.lo += rb, L.hi += rb, L.mask += rb, L.X += rb;
    }
    const bool bounded = p.bound_mode != MPC_BOUND_NONE;

    // ------------------------------------------------------------------ P1: everything that is independent over t
    double oc = 0;
    for (int t = j; t < T; t += 16) {
        const long tb = (long)t * B + b;
        const float *Ct = p.C + (long)t * p.C_st + (long)b * p.C_sb;
        const float *ct = p.c + (long)t * p.c_st + (long)b * p.c_sb;
        float C[N][N], c[N], tau[N];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) C[i][j] = Ct[i * N + j];
        for (int i = 0; i < N; ++i) c[i] = ct[i];
        for (int i = 0; i < NS; ++i) tau[i] = p.cur_x[tb * NS + i];
        tau[NS] = p.cur_u[tb];
        for (int i = 0; i < N; ++i) {
            float r = 0;
            for (int j = 0; j < N; ++j) {
                r += C[i][j] * tau[j];
                wv::sm(L.C + (t * N + i) * N + j) = C[i][j];
            }
            oc += (double)(0.5f * tau[i] * r + c[i] * tau[i]);
            wv::sm(L.cb + t * N + i) = r + c[i];
            wv::sm(L.c + t * N + i) = c[i];
            wv::sm(L.tau + t * N + i) = tau[i];
        }
        if (t < T - 1) {
            if (p.env.kind && p.env.linearize) {       // F_t = d simulator / d [x;u] at the nominal
                float nxt[NS > 5 ? NS : 5], J[NS * N > 30 ? NS * N : 30];
                env_step<float>(p.env, tau, tau[NS], nxt, J);
                for (int m = 0; m < NS * N; ++m) wv::sm(L.F + t * NS * N + m) = J[m];
            } else {
                const float *Ft = p.F + (long)t * p.F_st + (long)b * p.F_sb;
                for (int m = 0; m < NS * N; ++m) wv::sm(L.F + t * NS * N + m) = Ft[m];
            }
            if (has_f) {
                const float *ft = p.f + (long)t * p.f_st + (long)b * p.f_sb;
                for (int i = 0; i < NS; ++i) wv::sm(L.f + t * NS + i) = ft[i];
            }
        }
        if (bounded) {
            wv::sm(L.lo + t) = p.bound_mode == MPC_BOUND_SCALAR ? p.lo_s : p.lo[tb];
            wv::sm(L.hi + t) = p.bound_mode == MPC_BOUND_SCALAR ? p.hi_s : p.hi[tb];
        }
        if (masked_call) wv::sm(L.mask + t) = p.zero_mask[tb] ? 1.f : 0.f;
    }
    const double old_cost = wv::row_sum_f64(oc);
    wv::lds_sync();

    // ------------------------------------------------------------------ P2: the Riccati recursion
    int status = 0, qp_total = 0;
#ifndef MPC_W1_SKIP                                            // (diagnostic builds, tools/ab_w1_phases.sh: bit 0 = no P2, 1 = no P3, 2 = no P4)
#define MPC_W1_SKIP 0
#endif
    if (!(MPC_W1_SKIP & 1)) {
        const int jc = j < N ? j : N - 1;                     // (columns >= N idle along on column N - 1's data)
        float Vc[NS], vj = 0;                                 // column j of V_{t+1}, element j of v_{t+1}
        for (int i = 0; i < NS; ++i) Vc[i] = 0;
        bool warm = false;
        float kprev = 0;
        struct Stage { float Cc[N], Crn, cb, Fc[NS], u, lo, hi, mask; };
        auto fetch = [&](int t, Stage &g) {
            for (int i = 0; i < N; ++i) g.Cc[i] = wv::sm(L.C + (t * N + i) * N + jc);
            g.Crn = wv::sm(L.C + (t * N + jc) * N + NS);
            g.cb = wv::sm(L.cb + t * N + jc);
            for (int m = 0; m < NS; ++m) g.Fc[m] = t < T - 1 ? wv::sm(L.F + (t * NS + m) * N + jc) : 0.f;
            g.u = wv::sm(L.tau + t * N + NS);
            g.lo = bounded ? wv::sm(L.lo + t) : 0.f;
            g.hi = bounded ? wv::sm(L.hi + t) : 0.f;
            g.mask = masked_call ? wv::sm(L.mask + t) : 0.f;
        };
#ifdef MPC_W1_PROF                                             // (diagnostic build: clock stamps of one timestep of wavefront 0 into qp_iters[8..])
#define W1_STAMP(k) do { if (t == T / 2) stamp[k] = wv::clock(); } while (0)
#define W1_STAMP5 stamp[5] = wv::clock()
        long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
#define W1_STAMP(k) do { } while (0)
#define W1_STAMP5 (void)0
#endif
        auto riccati = [&](int t, const Stage &now) {
            W1_STAMP(0);
            float qj = now.cb, Qc[N], Qjn = now.Crn;          // q[j]; column j of Q; Q[j][NS]
            for (int i = 0; i < N; ++i) Qc[i] = now.Cc[i];
            if (t < T - 1) {                                  // Q = C + F'VF, q = c_back + F'v (:65-70)
                float Wc[NS];                                 // column j of W = V F: W[m][j] = sum_l V[m][l] F[l][j]
                for (int m = 0; m < NS; ++m) Wc[m] = 0;
                static_for<0, NS>([&](auto l) {               // (m inner: NS independent accumulation chains)
                    for (int m = 0; m < NS; ++m) wv::fmac_bcast<l.value>(Wc[m], Vc[m], now.Fc[l.value]);
                });
                for (int m = 0; m < NS; ++m) {                // Q[i][j] += F[m][i] W[m][j]   (i inner: N independent chains)
                    static_for<0, N>([&](auto i) { wv::fmac_bcast<i.value>(Qc[i.value], now.Fc[m], Wc[m]); });
                    wv::fmac_bcast<NS>(Qjn, Wc[m], now.Fc[m]);            // Q[j][NS] += F[m][j] W[m][NS]
                }
                static_for<0, NS>([&](auto m) { wv::fmac_bcast<m.value>(qj, vj, now.Fc[m.value]); });
            }
            W1_STAMP(1);
            const float Quu = wv::bcast<NS>(Qc[NS]), qu = wv::bcast<NS>(qj);
            float Kj, k;
            if (!bounded) {
                const bool masked = now.mask != 0.f;
                const float inv = env_inv(Quu);
                Kj = masked ? 0.f : -(inv * Qc[NS]);          // :86-87; the control pinned (:99-127): K = 0, k = 0
                k = masked ? 0.f : -(inv * qu);
            } else {                                          // :128-148, every lane solves the same scalar QP
                float lb = now.lo - now.u, ub = now.hi - now.u;
                if (p.has_delta) {                            // :132-134
                    if (lb < -p.delta_u) lb = -p.delta_u;
                    if (ub > p.delta_u) ub = p.delta_u;
                }
                float x = warm ? kprev : -(qu * env_inv(Quu));
                float Hf;
                bool is_free, conv;
                const int it = pnqp1_fast(Quu, qu, lb, ub, x, Hf, is_free, p.pnqp_iter, conv);
                qp_total += 1 + it;                           // :140
                if (!conv) status |= MPC_ST_PNQP_UNCONVERGED;
                warm = true;
                k = x;
                Kj = is_free ? -(Qc[NS] * env_inv(Hf)) : 0.f; // :142-146
            }
            kprev = k;
            W1_STAMP(2);
            wv::sm(L.K + t * N + jc) = j < NS ? Kj : k;       // (every lane stores: columns >= N repeat k in column N - 1's slot)
            if ((p.K || p.k) && active) {                     // (a wavefront-uniform test first: mpc.MPC's loop asks for no gains)
                const long tb = (long)t * B + b;
                if (j < NS) {
                    if (p.K) p.K[tb * NS + j] = Kj;
                } else if (j == NS && p.k) {
                    p.k[tb] = k;
                }
            }
            // :155-158 V = Qxx + Qxu K + K'Qux + K'Quu K, v likewise (unmasked Quu, qu)
            W1_STAMP(3);
            const float Mj = Qc[NS] + Quu * Kj, mk = qu + Quu * k;
            static_for<0, NS>([&](auto i) {
                float v = Qc[i.value];
                wv::fmac_bcast<NS>(v, Qc[i.value], Kj);       // Q[i][NS] K[j]
                wv::fmac_bcast<i.value>(v, Kj, Mj);           // K[i] M[j]
                Vc[i.value] = v;
            });
            vj = qj + Qjn * k + Kj * mk;
            W1_STAMP(4);
            if (t == T / 2 - 1) W1_STAMP5;
        };
        // two stages that take turns (the body is compiled twice): the operands of step t - 1 are read from LDS while step t
        // is computed, and nothing is copied from one stage to the other
        Stage sa, sb;
        fetch(T - 1, sa);
        for (int t = T - 1; t >= 0; t -= 2) {
            if (t > 0) fetch(t - 1, sb);
            riccati(t, sa);
            if (t == 0) break;
            if (t > 1) fetch(t - 2, sa);
            riccati(t - 1, sb);
        }
#ifdef MPC_W1_PROF
        if (wv::problem() == 0 && lane == 0 && p.qp_iters)
            for (int k = 0; k < 5; ++k) p.qp_iters[8 + k] = (int)(stamp[k + 1] - stamp[k]);
#endif
    }
    wv::lds_sync();

    // ------------------------------------------------------------------ P3: every line-search trial at once
    float alpha = 1, dun = 0;
    for (int i = 0; i < j && i < ntr; ++i) alpha *= p.ls_decay;         // the same products the sequential search forms
    if (j < ntr && !(MPC_W1_SKIP & 2)) {
        const int xo = L.X + j * T * N;
        float x[NS], dx[NS], da = 0;
        for (int i = 0; i < NS; ++i) {
            x[i] = p.x_init[(long)b * NS + i];
            dx[i] = 0;
        }
        // (a lane walks its trial alone: the operands of step t + 1 -- gains, nominal, bounds: nothing that depends on the
        //  state -- are read from LDS while step t is computed, or every use would wait out an LDS round trip)
        struct Stage { float K[N], u, mask, lo, hi, xn[NS]; };
        auto fetch = [&](int t, Stage &g) {
            for (int i = 0; i < N; ++i) g.K[i] = wv::sm(L.K + t * N + i);
            g.u = wv::sm(L.tau + t * N + NS);
            g.mask = masked_call ? wv::sm(L.mask + t) : 0.f;
            g.lo = bounded ? wv::sm(L.lo + t) : 0.f;
            g.hi = bounded ? wv::sm(L.hi + t) : 0.f;
            for (int i = 0; i < NS; ++i) g.xn[i] = t < T - 1 ? wv::sm(L.tau + (t + 1) * N + i) : 0.f;
        };
        auto advance = [&](int t, const Stage &now) {
            float r = 0;
            for (int i = 0; i < NS; ++i) r += now.K[i] * dx[i];
            const float u = now.u;
            float un = r + u + alpha * now.K[NS];                           // :192
            if (now.mask != 0.f) un = 0;                                    // :197-198
            if (bounded) {                                                  // :200-213
                float l = now.lo, h = now.hi;
                if (p.has_delta) {
                    const float l2 = u - p.delta_u, h2 = u + p.delta_u;
                    l = (l2 < l) ? l : l2;
                    h = (h2 > h) ? h : h2;
                }
                un = tiny::clampr<float>(un, l, h);
            }
            da += (u - un) * (u - un);
            for (int i = 0; i < NS; ++i) wv::sm(xo + t * N + i) = x[i];
            wv::sm(xo + t * N + NS) = un;
            if (t < T - 1) {
                float xn[NS > 5 ? NS : 5];
                if (p.env.kind) {                                           // :223-225
                    env_step<float>(p.env, x, un, xn, nullptr);
                } else {                                                    // :216-222
                    for (int i = 0; i < NS; ++i) {
                        float s = 0;
                        for (int jj = 0; jj < NS; ++jj) s += wv::sm(L.F + (t * NS + i) * N + jj) * x[jj];
                        s += wv::sm(L.F + (t * NS + i) * N + NS) * un;
                        xn[i] = has_f ? s + wv::sm(L.f + t * NS + i) : s;
                    }
                }
                for (int i = 0; i < NS; ++i) {
                    x[i] = xn[i];
                    dx[i] = xn[i] - now.xn[i];
                }
            }
        };
        Stage sa, sb;                                  // (two stages taking turns, as in the recursion)
        fetch(0, sa);
        for (int t = 0; t < T; t += 2) {
            if (t + 1 < T) fetch(t + 1, sb);
            advance(t, sa);
            if (t + 1 >= T) break;
            if (t + 2 < T) fetch(t + 2, sa);
            advance(t + 1, sb);
        }
        dun = sqrtf(da);
    }
    wv::lds_sync();

    // ------------------------------------------------------------------ P4: the trials' costs, (trial, slice) per lane
    int Gp = 1;
    while (Gp < ntr) Gp <<= 1;
    double cost = 0;
    {
        const int g = j & (Gp - 1), slice = j / Gp, nsl = 16 / Gp;
        if (g < ntr && !(MPC_W1_SKIP & 4))
            for (int t = slice; t < T; t += nsl) {
                float tau[N];
                for (int i = 0; i < N; ++i) tau[i] = wv::sm(L.X + (g * T + t) * N + i);
                for (int i = 0; i < N; ++i) {                               // :230-232
                    float s = 0;
                    for (int jj = 0; jj < N; ++jj) s += wv::sm(L.C + (t * N + i) * N + jj) * tau[jj];
                    cost += (double)(0.5f * tau[i] * s + wv::sm(L.c + t * N + i) * tau[i]);
                }
            }
        for (int off = Gp; off < 16; off <<= 1) cost += wv::shfl_xor_f64(cost, off);
    }

    // ------------------------------------------------------------------ P5: the accepted trial
    const unsigned okm = (unsigned)(wv::ballot(j < ntr && !(cost > old_cost)) >> (lane & 48)) & 0xffffu;   // the row's sixteen bits
    const int win = okm ? wv::ctz64(okm) : ntr - 1;                         // nothing helped: the last trial stands
    const int rl = lane & 48;
    const double win_cost = wv::readlane_f64(cost, rl + win);
    const float win_dun = wv::readlane(dun, rl + win), full = wv::readlane(dun, rl), win_alpha = wv::readlane(alpha, rl + win);
    if (!active) return;
    for (int idx = j; idx < T * N; idx += 16) {
        const int t = idx / N, i = idx - t * N;
        const float v = wv::sm(L.X + (win * T + t) * N + i);
        const long tb = (long)t * B + b;
        if (i < NS) p.new_x[tb * NS + i] = v;
        else p.new_u[tb] = v;
    }
    if (j != 0) return;
    if (!(win_cost == win_cost) || tiny::absr<double>(win_cost) > 3e38) status |= MPC_ST_NONFINITE;
    if (p.costs) p.costs[b] = (float)win_cost;
    if (p.old_costs) p.old_costs[b] = (float)old_cost;
    if (p.full_du_norm) p.full_du_norm[b] = full;
    if (p.alpha_du_norm) p.alpha_du_norm[b] = win_dun;
    if (p.alphas) p.alphas[b] = win_alpha;
#ifdef MPC_W1_PROF
    if (b >= 8 && b < 16) return;
#endif
    if (p.qp_iters) p.qp_iters[b] = qp_total;
    if (p.status) p.status[b] = status;
}

}  // namespace wave1
}  // namespace mpclqr
