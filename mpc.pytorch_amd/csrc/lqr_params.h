// lqr_params.h -- the kernel parameter block shared by every LQR kernel (plain C++: also
// compiled by the host-side lane emulator in tests/emu/, which has no HIP runtime).
#pragma once
#include <stdint.h>
#include "../../include/mpc_lqr.h"
#include "env_dynamics.h"

// diagnostic builds (-DMPC_QP_START=0): the kernels without the mpc_lqr_options.qp_start branch, for A/B timing of the cold path
#ifndef MPC_QP_START
#define MPC_QP_START 1
#endif

namespace mpclqr {

template <typename real>
struct StepParams {
    int B, T, ns, nc;
    const real *x_init, *C, *c, *F, *f, *cur_x, *cur_u;
    long C_st, C_sb, c_st, c_sb, F_st, F_sb, f_st, f_sb;
    int bound_mode;
    real lo_s, hi_s;
    const real *lo, *hi;
    const uint8_t *zero_mask;
    int has_delta;
    real delta_u;
    real ls_decay;
    int max_ls;
    int pnqp_iter;
    int on_dynamics;             // MPC_OPT_NOMINAL_ON_DYNAMICS: the nominal is known to obey the dynamics
    int sweep_only;              // MPC_OPT_SWEEP_ONLY: gains, nominal cost and QP counts, no rollout
    int c_symmetric;             // MPC_OPT_C_SYMMETRIC: the caller vouches for C = C' (no symmetry test in the fused kernels)
    const real *qp_start;        // MODE 2 of the 12/4 and 32/8 kernels: where the box QP of (t, b) starts, [T,B,nc] through the two
    long qp_start_st, qp_start_sb;   // element strides (mpc_lqr_options.qp_start); NULL = the reference's start (k of timestep t+1)
    const int *gate;             // generic kernel only: solve problem b iff gate[b] & MPC_ST_C_ASYMMETRIC (NULL = every problem)
    real *K_user, *k_user;       // padded 32/8 instantiation only: the caller's K [T,B,nc,ns] / k [T,B,nc] (K, k are then the
                                 // kernel's own padded gains [T,B,8,32] / [T,B,8] in the workspace); NULL = not asked for
    // outputs
    real *new_x, *new_u, *costs, *old_costs, *full_du_norm, *alpha_du_norm, *alphas;
    int *qp_iters, *status;
    real *K, *k;                 // [T,B,nc,ns], [T,B,nc]
    real *Kk;                    // fused MFMA kernel: its own gain record [T,B,4,16] (workspace)
    const real *old_costs_in;    // rollout-only entry point
    EnvDesc<real> env;           // true_dynamics of the rollout (kind 0 = the linear model)
};

template <typename real>
inline void set_env(EnvDesc<real> &e, const mpc_env_dynamics *d)
{
    e.kind = d->kind; e.linearize = d->linearize; e.params = (const real *)d->params; e.dt = (real)d->dt; e.u_max = (real)d->u_max;
}

template <typename real>
inline StepParams<real> make_params(const mpc_lqr_problem *p, const mpc_lqr_options *o,
                                    const mpc_lqr_outputs *out)
{
    StepParams<real> s;
    s.B = p->B; s.T = p->T; s.ns = p->ns; s.nc = p->nc;
    s.x_init = (const real *)p->x_init;
    s.C = (const real *)p->C; s.C_st = p->C_st; s.C_sb = p->C_sb;
    s.c = (const real *)p->c; s.c_st = p->c_st; s.c_sb = p->c_sb;
    s.F = (const real *)p->F; s.F_st = p->F_st; s.F_sb = p->F_sb;
    s.f = (const real *)p->f; s.f_st = p->f_st; s.f_sb = p->f_sb;
    s.cur_x = (const real *)p->cur_x; s.cur_u = (const real *)p->cur_u;
    s.bound_mode = o ? o->bound_mode : 0;
    s.lo_s = o ? (real)o->lo_s : (real)0; s.hi_s = o ? (real)o->hi_s : (real)0;
    s.lo = o ? (const real *)o->lo : nullptr; s.hi = o ? (const real *)o->hi : nullptr;
    s.zero_mask = o ? o->zero_mask : nullptr;
    s.has_delta = (o && o->delta_u == o->delta_u && o->delta_u >= 0) ? 1 : 0;
    s.delta_u = s.has_delta ? (real)o->delta_u : (real)0;
    s.ls_decay = o ? (real)o->linesearch_decay : (real)0.2;
    s.max_ls = o ? o->max_linesearch_iter : 10;
    s.pnqp_iter = (o && o->pnqp_iter > 0) ? o->pnqp_iter : 20;
    s.on_dynamics = (o && (o->flags & MPC_OPT_NOMINAL_ON_DYNAMICS)) ? 1 : 0;
    s.sweep_only = (o && (o->flags & MPC_OPT_SWEEP_ONLY)) ? 1 : 0;
    s.c_symmetric = (o && (o->flags & MPC_OPT_C_SYMMETRIC)) ? 1 : 0;
    s.qp_start = o ? (const real *)o->qp_start : nullptr;
    s.qp_start_st = o ? o->qp_start_st : 0; s.qp_start_sb = o ? o->qp_start_sb : 0;
    s.gate = nullptr;
    s.K_user = nullptr; s.k_user = nullptr;
    s.new_x = out ? (real *)out->new_x : nullptr; s.new_u = out ? (real *)out->new_u : nullptr;
    s.costs = out ? (real *)out->costs : nullptr; s.old_costs = out ? (real *)out->old_costs : nullptr;
    s.full_du_norm = out ? (real *)out->full_du_norm : nullptr;
    s.alpha_du_norm = out ? (real *)out->alpha_du_norm : nullptr;
    s.alphas = out ? (real *)out->alphas : nullptr;
    s.qp_iters = out ? out->qp_iters : nullptr; s.status = out ? out->status : nullptr;
    s.K = out ? (real *)out->K : nullptr; s.k = out ? (real *)out->k : nullptr;
    s.Kk = nullptr;
    s.old_costs_in = nullptr;
    s.env.kind = MPC_ENV_NONE; s.env.linearize = 0; s.env.params = nullptr; s.env.dt = 0; s.env.u_max = 0;
    if (o && o->true_dynamics) set_env(s.env, o->true_dynamics);
    return s;
}

}  // namespace mpclqr
