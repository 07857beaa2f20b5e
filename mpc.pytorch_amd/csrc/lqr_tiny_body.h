// lqr_tiny_body.h -- one LQR step for problems with ONE control and a handful of states
// (the reference's pendulum / cart-pole iLQR, BASELINE configs 2 and 3), ONE LANE PER PROBLEM.
//
// At n = n_state + 1 <= 7 a problem's whole Riccati recursion fits in one lane's registers
// (Q 7x7, V 6x6, F 6x7): no LDS, no barriers, no cross-lane traffic, 64 problems per wavefront.
// The generic kernel spends a wavefront (and ~10 barriers per timestep) on each of these.
// Same semantics as lqr_generic.hip, specialised to n_ctrl = 1:
//   sweep    mpc/lqr_step.py:52-160 (the nc == 1 branches :86-87, :121-123, :144-146)
//   pnqp     mpc/pnqp.py:5-82 with n = 1 (:15-16, :50-51)
//   rollout  mpc/lqr_step.py:164-261, true dynamics = F,f (:216-222) or a shipped simulator (:223-225)
// Plain C++ (no HIP builtins): tests/emu/ compiles this header for the host and checks it against the
// reference's golden outputs on a CPU-only box.
#pragma once
#include <math.h>
#include "lqr_params.h"

namespace mpclqr {
namespace tiny {

template <typename real> MPC_HD real clampr(real x, real lo, real hi)
{
    if (x < lo) x = lo;      // util.eclamp (mpc/util.py:56-70): strict compares, bound written exactly
    if (x > hi) x = hi;
    return x;
}
template <typename real> MPC_HD real absr(real x) { return x < 0 ? -x : x; }

// pnqp for n = 1 (mpc/pnqp.py:5-82).  Returns the iteration index the reference returns; x in/out,
// Hfree = the (regularised) free-set Hessian the returned x belongs to, is_free its free flag.
template <typename real>
MPC_HD int pnqp1(real H, real q, real lb, real ub, real &x, real &Hfree, bool &is_free, int n_iter, bool &conv)
{
    const real GAMMA = (real)0.1;
    x = clampr<real>(x, lb, ub);                                   // :23
    conv = false;
    int ret = n_iter - 1;
    Hfree = H + (real)1e-11;
    is_free = true;
    for (int it = 0; it < n_iter; ++it) {
        const real g = H * x + q;                                  // :29
        const bool ic = (x == lb && g > 0) || (x == ub && g < 0);  // :32
        is_free = !ic;
        Hfree = (is_free ? H : (real)0) + (real)1e-11;             // :44-48
        const real dx = -((is_free ? g : (real)0) * env_inv(Hfree)); // :50-51 (reciprocal + Newton step, see env_inv)
        if (!(absr<real>(dx) >= (real)1e-4)) { conv = true; ret = it; break; }   // :56-59
        if (x + dx >= lb && x + dx <= ub) {                        // Newton step inside the box: its Armijo
            x = x + dx;                                            // ratio is exactly 1/2, no evaluation
            continue;
        }
        real alpha = 1, arm = GAMMA, xn = x;
        int count = 0;
        while (arm <= GAMMA && count < 10) {                       // :64-76
            xn = clampr<real>(x + alpha * dx, lb, ub);
            const real d = xn - x;                                 // f(x) - f(xn) = -g d - H d^2 / 2, without
            arm = (-g * d - (real)0.5 * H * d * d) * env_inv(-g * d);   // subtracting two large objective values
            if (arm <= GAMMA) alpha *= (real)0.1;
            ++count;
        }
        x = xn;                                                    // :78
    }
    return ret;
}

// The group of lanes that shares one problem: lane g of G evaluates line-search trial g of a round.
// On the host (tests/emu) a "group" is one lane, and the rounds below are the reference's sequential passes.
struct OneLane {
    MPC_HD int G() const { return 1; }
    MPC_HD int g() const { return 0; }
    MPC_HD void gather(double mine, double *all) const { all[0] = mine; }
    MPC_HD void gains_visible() const {}
};

// Riccati sweep of one problem.  Kw: gain scratch laid out [T][NS+1][B] (k in row NS) so neighbouring lanes
// touch neighbouring words.  Every lane of a group runs the sweep (same instructions, no extra time); only
// `writer` stores.
template <typename real, int NS>
MPC_HD void sweep_problem(const StepParams<real> &p, int b, real *Kw, bool writer, double &old_cost, int &status,
                          int &qp_total)
{
    constexpr int N = NS + 1;
    const int T = p.T, B = p.B;
    real V[NS][NS], v[NS];
    // trajectory costs are summed in double: the line search compares two nearly equal sums once iLQR
    // is close to its fixed point, and float32 rounding there costs whole extra rollout passes
    old_cost = 0;
    status = 0;
    qp_total = 0;
    bool warm = false;
    real kprev = 0;

    // One lane walks the horizon alone: nothing hides a load's latency but the lane's own arithmetic.  Up to
    // n_state = 4 the stage of the next timestep (C, c, x, u: n^2 + 2n + 1 numbers) is fetched while this one is
    // worked on; wider states would pay for the second copy with register spills.
    constexpr bool PREFETCH = NS <= 4;
    struct Stage { real C[N][N], c[N], tau[N]; };
    // The stages are visited in order, so every array is a per-lane pointer that steps back by one timestep per
    // fetch: a vector register pair each, instead of a scalar base and stride per array -- the kernel's ~14 arrays
    // do not fit the scalar registers, and their spills were a quarter of its instructions.
    const real *Cp = p.C + (long)(T - 1) * p.C_st + (long)b * p.C_sb;
    const real *cp = p.c + (long)(T - 1) * p.c_st + (long)b * p.c_sb;
    const real *xp = p.cur_x + ((long)(T - 1) * B + b) * NS;
    const real *up = p.cur_u + ((long)(T - 1) * B + b);
    auto fetch = [&](int, Stage &g) {
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) g.C[i][j] = Cp[i * N + j];
        for (int i = 0; i < N; ++i) g.c[i] = cp[i];
        for (int i = 0; i < NS; ++i) g.tau[i] = xp[i];
        g.tau[NS] = up[0];
        Cp -= p.C_st;
        cp -= p.c_st;
        xp -= (long)B * NS;
        up -= B;
    };
    Stage ahead;
    if (PREFETCH) fetch(T - 1, ahead);

    // ------------------------------------------------------------------ sweep
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        Stage held;
        if (PREFETCH) {
            held = ahead;
            if (t > 0) fetch(t - 1, ahead);
        } else {
            fetch(t, ahead);
        }
        const Stage &now = PREFETCH ? held : ahead;
        real Q[N][N], q[N], tau[N];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) Q[i][j] = now.C[i][j];
        for (int i = 0; i < N; ++i) tau[i] = now.tau[i];
        for (int i = 0; i < N; ++i) {                  // c_back = C tau + c (:289-295) and the nominal cost (:169)
            real r = 0;
            for (int j = 0; j < N; ++j) r += Q[i][j] * tau[j];
            const real ci = now.c[i];
            old_cost += (double)((real)0.5 * tau[i] * r + ci * tau[i]);
            q[i] = r + ci;
        }
        if (t < T - 1) {                               // Q = C + F'VF, q = c_back + F'v (:65-70)
            real F[NS][N];
            if (p.env.kind && p.env.linearize) {       // F_t = d simulator / d [x;u] at the nominal (mpc/mpc.py:490-549)
                real nxt[NS > 5 ? NS : 5], J[NS * N > 30 ? NS * N : 30];   // (sized for either simulator)
                env_step<real>(p.env, tau, tau[NS], nxt, J);
                for (int m = 0; m < NS; ++m)
                    for (int j = 0; j < N; ++j) F[m][j] = J[m * N + j];
            } else {
                const real *Ft = p.F + (long)t * p.F_st + (long)b * p.F_sb;
                for (int m = 0; m < NS; ++m)
                    for (int j = 0; j < N; ++j) F[m][j] = Ft[m * N + j];
            }
            for (int j = 0; j < N; ++j) {
                real Y[NS];                            // column j of V F
                for (int m = 0; m < NS; ++m) {
                    real r = 0;
                    for (int l = 0; l < NS; ++l) r += V[m][l] * F[l][j];
                    Y[m] = r;
                }
                for (int i = 0; i < N; ++i) {
                    real r = 0;
                    for (int m = 0; m < NS; ++m) r += F[m][i] * Y[m];
                    Q[i][j] += r;
                }
            }
            for (int i = 0; i < N; ++i) {
                real r = 0;
                for (int m = 0; m < NS; ++m) r += F[m][i] * v[m];
                q[i] += r;
            }
        }
        const real Quu = Q[NS][NS], qu = q[NS];
        real K[NS], k;
        if (p.bound_mode == MPC_BOUND_NONE) {
            const bool masked = p.zero_mask && p.zero_mask[tb];
            if (masked) {                              // :99-127 with the control pinned: K = 0, k = 0
                for (int j = 0; j < NS; ++j) K[j] = 0;
                k = 0;
            } else {                                   // :86-87
                const real inv = env_inv(Quu);
                for (int j = 0; j < NS; ++j) K[j] = -(inv * Q[NS][j]);
                k = -(inv * qu);
            }
        } else {                                       // :128-148
            const real u = tau[NS];
            real lb = (p.bound_mode == MPC_BOUND_SCALAR ? p.lo_s : p.lo[tb]) - u;
            real ub = (p.bound_mode == MPC_BOUND_SCALAR ? p.hi_s : p.hi[tb]) - u;
            if (p.has_delta) {                         // :132-134
                if (lb < -p.delta_u) lb = -p.delta_u;
                if (ub > p.delta_u) ub = p.delta_u;
            }
            real x = warm ? kprev : -(qu * env_inv(Quu)); // warm start = k_{t+1} (:137,141) / cold start pnqp.py:15-16
            real Hf;
            bool is_free, conv;
            const int it = pnqp1<real>(Quu, qu, lb, ub, x, Hf, is_free, p.pnqp_iter, conv);
            qp_total += 1 + it;                        // :140
            if (!conv) status |= MPC_ST_PNQP_UNCONVERGED;
            warm = true;
            k = x;
            const real iHf = env_inv(Hf);
            for (int j = 0; j < NS; ++j) K[j] = is_free ? -(Q[NS][j] * iHf) : (real)0;   // :142-146
        }
        kprev = k;
        if (writer) {
            for (int j = 0; j < NS; ++j) Kw[((long)t * N + j) * B + b] = K[j];
            Kw[((long)t * N + NS) * B + b] = k;
            if (p.K) for (int j = 0; j < NS; ++j) p.K[tb * NS + j] = K[j];
            if (p.k) p.k[tb] = k;
        }
        // :155-158 V = Qxx + Qxu K + K'Qux + K'Quu K, v likewise (unmasked Quu, qu)
        real M[NS];
        for (int j = 0; j < NS; ++j) M[j] = Q[NS][j] + Quu * K[j];
        const real mk = qu + Quu * k;
        for (int i = 0; i < NS; ++i) {
            for (int j = 0; j < NS; ++j) V[i][j] = Q[i][j] + Q[i][NS] * K[j] + K[i] * M[j];
            v[i] = q[i] + Q[i][NS] * k + K[i] * mk;
        }
    }

}

// One rollout with step size alpha (mpc/lqr_step.py:186-241): trajectory cost and ||u - u'||.  The trajectory goes
// to ox [T][OB][NS] / ou [T][OB] at column ob (the outputs themselves for trial 0, a scratch column per trial lane
// otherwise); ox == nullptr: nowhere.
template <typename real, int NS>
MPC_HD void rollout_pass(const StepParams<real> &p, int b, const real *Kw, real alpha, real *ox, real *ou, long OB, long ob,
                         double &cost, real &dun)
{
    constexpr int N = NS + 1;
    const int T = p.T, B = p.B;
    const bool store = ox != nullptr;
    real x[NS], dx[NS];
    for (int i = 0; i < NS; ++i) {
        x[i] = p.x_init[(long)b * NS + i];
        dx[i] = 0;
        if (store) ox[ob * NS + i] = x[i];
    }
    real da = 0;
    double ca = 0;
    // (the same one-stage look-ahead as the sweep: gains, nominal control and next nominal state, C, c)
    constexpr bool PREFETCH = NS <= 4;
    struct Stage { real K[N], u, xn[NS], C[N][N], c[N]; };
    // (per-lane pointers stepping forward one timestep per fetch, as in the sweep)
    const real *Cp = p.C + (long)b * p.C_sb;
    const real *cp = p.c + (long)b * p.c_sb;
    const real *Kp = Kw + b;
    const real *up = p.cur_u + b;
    const real *xp = p.cur_x + ((long)B + b) * NS;         // the nominal state of t + 1
    auto fetch = [&](int t, Stage &g) {
        for (int j = 0; j < N; ++j) g.K[j] = Kp[(long)j * B];
        g.u = up[0];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) g.C[i][j] = Cp[i * N + j];
        for (int i = 0; i < N; ++i) g.c[i] = cp[i];
        if (t < T - 1) {
            for (int i = 0; i < NS; ++i) g.xn[i] = xp[i];
        } else {
            for (int i = 0; i < NS; ++i) g.xn[i] = 0;
        }
        Cp += p.C_st;
        cp += p.c_st;
        Kp += (long)N * B;
        up += B;
        xp += (long)B * NS;
    };
    Stage ahead;
    if (PREFETCH) fetch(0, ahead);
    for (int t = 0; t < T; ++t) {
        const long tb = (long)t * B + b;
        Stage held;
        if (PREFETCH) {
            held = ahead;
            if (t + 1 < T) fetch(t + 1, ahead);
        } else {
            fetch(t, ahead);
        }
        const Stage &now = PREFETCH ? held : ahead;
        real r = 0;
        for (int j = 0; j < NS; ++j) r += now.K[j] * dx[j];
        const real u = now.u;
        real un = r + u + alpha * now.K[NS];                                // :192
        if (p.zero_mask && p.zero_mask[tb]) un = 0;                         // :197-198
        if (p.bound_mode != MPC_BOUND_NONE) {                               // :200-213
            real l = p.bound_mode == MPC_BOUND_SCALAR ? p.lo_s : p.lo[tb];
            real h = p.bound_mode == MPC_BOUND_SCALAR ? p.hi_s : p.hi[tb];
            if (p.has_delta) {
                const real l2 = u - p.delta_u, h2 = u + p.delta_u;
                l = (l2 < l) ? l : l2;
                h = (h2 > h) ? h : h2;
            }
            un = clampr<real>(un, l, h);
        }
        if (store) ou[(long)t * OB + ob] = un;
        da += (u - un) * (u - un);
        real tau[N];
        for (int j = 0; j < NS; ++j) tau[j] = x[j];
        tau[NS] = un;
        for (int i = 0; i < N; ++i) {                                       // :230-232
            real s = 0;
            for (int j = 0; j < N; ++j) s += now.C[i][j] * tau[j];
            ca += (double)((real)0.5 * tau[i] * s + now.c[i] * tau[i]);
        }
        if (t < T - 1) {
            real xn[NS > 5 ? NS : 5];
            if (p.env.kind) {                                               // :223-225
                env_step<real>(p.env, x, un, xn, nullptr);
            } else {                                                        // :216-222
                const real *Ft = p.F + (long)t * p.F_st + (long)b * p.F_sb;
                const real *ft = p.f ? p.f + (long)t * p.f_st + (long)b * p.f_sb : nullptr;
                for (int i = 0; i < NS; ++i) {
                    real s = 0;
                    for (int j = 0; j < N; ++j) s += Ft[i * N + j] * tau[j];
                    xn[i] = ft ? s + ft[i] : s;
                }
            }
            const long o1 = ((long)(t + 1) * OB + ob) * NS;
            for (int i = 0; i < NS; ++i) {
                x[i] = xn[i];
                dx[i] = xn[i] - now.xn[i];
                if (store) ox[o1 + i] = xn[i];
            }
        }
    }
    cost = ca;
    dun = sqrt(da);
}

// One problem on a group of lanes.  The reference's line search (:176-179, 247) tries alpha = decay^j for
// j = 0, 1, ... and keeps the first trial whose cost is not worse than the nominal's, else the last one.  A
// group evaluates G consecutive trials at once; trial 0 (alpha = 1, the usual winner) writes its trajectory
// to the outputs as it goes, every other trial into its lane's column of the scratch Tw (x [T][B*G][NS], then
// u [T][B*G]), from where the group copies an accepted one -- rounds 1-2 replayed the accepted trial instead, a whole
// extra rollout pass for every wavefront in which one problem took a shorter step (all of them, in practice:
// 72 -> 52 us per pendulum step, 103 -> 84 per cart-pole step).  `active`: this lane belongs to a real problem
// (idle tail lanes shadow the last one so that the group exchanges stay uniform).
template <typename real, int NS, class Lanes>
MPC_HD void lqr_step_problem(const StepParams<real> &p, int b, real *Kw, real *Tw, const Lanes &L, bool active = true)
{
    const int G = L.G(), g = L.g();
    const bool writer = active && g == 0;
    const long OB = (long)p.B * G, ob = (long)b * G + g;
    real *const tx = Tw, *const tu = Tw + (long)p.T * OB * NS;
    double old_cost;
    int status, qp_total;
    sweep_problem<real, NS>(p, b, Kw, writer, old_cost, status, qp_total);
    L.gains_visible();

    int win = -1;                       // index of the accepted trial
    double cost0 = 0, win_cost = 0;     // cost of trial 0 / of the accepted trial
    real full = 0, win_dun = 0, win_alpha = 1;
    for (int base = 0; base < p.max_ls && win < 0; base += G) {
        const int j = base + g;
        real alpha = 1;
        for (int i = 0; i < j; ++i) alpha *= p.ls_decay;          // the same products the sequential search forms
        double cost = 0;
        real dun = 0;
        if (j < p.max_ls) {              // (ONE call: the lanes of a group differ in where they store, not in what they run)
            const bool first = j == 0;
            real *ox = first ? p.new_x : tx;
            if (!active) ox = nullptr;
            rollout_pass<real, NS>(p, b, Kw, alpha, ox, first ? p.new_u : tu, first ? (long)p.B : OB, first ? (long)b : ob, cost, dun);
        }
        double costs[8], duns[8];
        L.gather(cost, costs);
        L.gather((double)dun, duns);
        const int n = p.max_ls - base < G ? p.max_ls - base : G;
        if (base == 0) { full = (real)duns[0]; cost0 = costs[0]; }    // :243-245
        int w = -1;
        for (int i = 0; i < n && w < 0; ++i)
            if (!(costs[i] > old_cost)) w = i;
        if (w < 0 && base + G >= p.max_ls) w = n - 1;                // nothing helped: the last trial stands
        if (w >= 0) {
            win = base + w;
            win_cost = costs[w];
            win_dun = (real)duns[w];
            win_alpha = 1;
            for (int i = 0; i < win; ++i) win_alpha *= p.ls_decay;
        }
    }
    if (win != 0) {                      // the accepted trial's trajectory: out of its lane's scratch column, a slice per lane
        L.gains_visible();
        const long wb = (long)b * G + win % G;
        if (active)
            for (int t = g; t < p.T; t += G) {
                const long src = (long)t * OB + wb, dst = (long)t * p.B + b;
                for (int i = 0; i < NS; ++i) p.new_x[dst * NS + i] = tx[src * NS + i];
                p.new_u[dst] = tu[src];
            }
    }
    (void)cost0;
    if (!writer) return;
    if (!(win_cost == win_cost) || absr<double>(win_cost) > 3e38) status |= MPC_ST_NONFINITE;
    if (p.costs) p.costs[b] = (real)win_cost;
    if (p.old_costs) p.old_costs[b] = (real)old_cost;
    if (p.full_du_norm) p.full_du_norm[b] = full;
    if (p.alpha_du_norm) p.alpha_du_norm[b] = win_dun;
    if (p.alphas) p.alphas[b] = win_alpha;
    if (p.qp_iters) p.qp_iters[b] = qp_total;
    if (p.status) p.status[b] = status;
}

MPC_HD bool shape_supported(int ns, int nc) { return nc == 1 && ns >= 1 && ns <= 6; }

}  // namespace tiny
}  // namespace mpclqr
