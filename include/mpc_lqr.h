/*
 * mpc_lqr.h -- C ABI of libmpc_lqr_hip.so: the MI355X (gfx950) batched LQR step.
 *
 * This is the drop-in boundary for the ONE hot path of locuslab/mpc.pytorch:
 * everything `LQRStep(...)(x_init, C, c, F, f)` does (mpc/lqr_step.py:22-409)
 * plus the helpers its caller `MPC.forward` runs on the same tensors
 * (mpc/util.py:102-153, mpc/pnqp.py:5-82).  Plain pointers and sizes only; no
 * torch types.  Every entry point
 *   - takes DEVICE pointers (hipMalloc'ed / torch ROCm storage),
 *   - enqueues on the given hipStream_t (passed as void*) and returns at once,
 *   - allocates nothing, never synchronises (hipGraph-capturable),
 *   - returns 0 on success, a negative MPC_E_* code on a rejected argument.
 *
 * Layout (identical to the reference's tensors): time-major, row-major,
 *   C [T,B,n,n]  c [T,B,n]  F [T-1,B,ns,n]  f [T-1,B,ns] (or NULL)
 *   x [T,B,ns]   u [T,B,nc] x_init [B,ns]   K [T,B,nc,ns]  k [T,B,nc]
 * with n = ns + nc.  C, c, F, f carry explicit element strides for the T and B
 * axes so the `.expand()`ed (stride-0) views MPC.forward builds
 * (mpc/mpc.py:207-221) are read in place; their inner block is contiguous.
 * All other arrays are contiguous.
 */
#ifndef MPC_LQR_H
#define MPC_LQR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_LQR_ABI_VERSION 9

enum { MPC_F32 = 0, MPC_F64 = 1 };
enum { MPC_BOUND_NONE = 0, MPC_BOUND_SCALAR = 1, MPC_BOUND_TENSOR = 2 };
enum {
    MPC_OK = 0,
    MPC_E_DIMS = -1,       /* unsupported / inconsistent sizes            */
    MPC_E_NULL = -2,       /* a required pointer is NULL                  */
    MPC_E_DTYPE = -3,
    MPC_E_LAUNCH = -4,     /* hipLaunch failed; see mpc_lqr_last_error()  */
    MPC_E_ARG = -5
};
/* per-problem status word bits (status[B]) */
enum {
    MPC_ST_PNQP_UNCONVERGED = 1,   /* "pnqp warning: Did not converge" (mpc/pnqp.py:81) at some timestep   */
    MPC_ST_NONFINITE = 2,          /* the returned cost is NaN / inf                                        */
    MPC_ST_NOMINAL_OFF_DYNAMICS = 4,/* informational (the 12/4 and 32/8 kernels): current_x is not the rollout of current_u
                                      from x_init, the trajectory cost was evaluated from a second pass over C */
    MPC_ST_C_ASYMMETRIC = 8,       /* some C_t of this problem is not symmetric (max |C - C'| > 1e-5 max |C|).  The
                                      reference uses C as given (mpc/lqr_step.py:68 Q = C + F'VF, :294 C tau); the fused
                                      kernels (impl 2..5) read it through its symmetry.  impl = 0 re-solves exactly these
                                      problems on the generic kernels in the same call (results are the reference's,
                                      the bit stays set as information); a FORCED fused impl leaves its symmetric-C
                                      results in place and the bit tells the caller they are not the reference's.   */
    MPC_ST_C_TESTED = 32,          /* (round 4) the kernel that solved this problem ran the symmetry test of C (the fused kernels, impl 2, 3, 5,
                                      on a call without MPC_OPT_C_SYMMETRIC): MPC_ST_C_ASYMMETRIC clear then MEANS symmetric.  The
                                      generic, lane-per-problem and row-per-problem kernels use C as given and never test it: without
                                      this bit a clear MPC_ST_C_ASYMMETRIC says nothing, and a caller (mpc.MPC) must not derive the
                                      promise MPC_OPT_C_SYMMETRIC from it.                                                       */
    MPC_ST_QUU_SINGULAR = 16       /* unconstrained solve with n_ctrl > 1 (the reference's pinverse, mpc/lqr_step.py:88-94):
                                      a pivot of Quu's factorisation was exactly zero and its control dropped out (gain 0).
                                      That is the pseudo-inverse when the null space is a coordinate axis -- a control that
                                      enters neither cost nor dynamics, the results are the reference's -- and a different
                                      generalised inverse otherwise (informational either way).                        */
};

/* The simulator dynamics the reference ships (mpc/env_dx/pendulum.py:18-84, cartpole.py:28-96),
 * usable as the `true_dynamics` of a step (mpc/lqr_step.py:223-225) and linearised in closed form
 * (replaces the (T-1)*n_state autograd passes of mpc/mpc.py:514-549).  n_ctrl = 1. */
enum {
    MPC_ENV_NONE = 0,
    MPC_ENV_PENDULUM = 1,        /* PendulumDx(simple=True):  params (g, m, l),       n_state 3 */
    MPC_ENV_PENDULUM_FULL = 2,   /* PendulumDx(simple=False): params (g, m, l, d, b), n_state 3 */
    MPC_ENV_CARTPOLE = 3         /* CartpoleDx: params (gravity, masscart, masspole, length), n_state 5 */
};
typedef struct mpc_env_dynamics {
    int32_t kind;                 /* MPC_ENV_* */
    int32_t linearize;            /* 1: mpc_lqr_step takes the sweep's F_t from the simulator's Jacobian at the nominal
                                     (current_x, current_u), computed in the kernel: MPC.linearize_dynamics
                                     (mpc/mpc.py:490-549) fused into the step; p->F, p->f are ignored (may be NULL).
                                     Lane-per-problem kernel only. */
    const void *params;           /* DEVICE pointer, dtype of the problem, 3 / 5 / 4 values */
    double dt;                    /* 0.05 in both modules */
    double u_max;                 /* max_torque (2.0) / force_mag (100.0): the module clamps u to +-u_max */
} mpc_env_dynamics;

/* The problem data of one LQRStepFn.forward call: (x_init, C, c, F, f) plus the
 * closure state `current_x/current_u` (mpc/lqr_step.py:22-38, 277). */
typedef struct mpc_lqr_problem {
    int32_t B, T, ns, nc;
    int32_t dtype;                    /* MPC_F32 | MPC_F64 */
    int32_t _pad;
    const void *x_init;               /* [B,ns] */
    const void *C; int64_t C_st, C_sb; /* element strides of the T and B axes */
    const void *c; int64_t c_st, c_sb;
    const void *F; int64_t F_st, F_sb; /* only t < T-1 is read (mpc/lqr_step.py:217) */
    const void *f; int64_t f_st, f_sb; /* NULL <=> the reference's empty tensor (mpc/mpc.py:360) */
    const void *cur_x;                /* [T,B,ns] nominal states   */
    const void *cur_u;                /* [T,B,nc] nominal controls */
} mpc_lqr_problem;

/* mpc_lqr_options.flags */
enum {
    MPC_OPT_NOMINAL_ON_DYNAMICS = 1  /* the caller GUARANTEES that cur_x is the rollout of cur_u through (F, f) from
                                        x_init -- what MPC.forward hands to every step (it recomputes x with
                                        util.get_traj, mpc/mpc.py:251) and what a previous step's (new_x, new_u) is.
                                        The 4-problems-per-wave kernel prices its rollout by an identity of the sweep's
                                        value function that holds exactly then; without the flag it verifies the premise
                                        at every timestep (and prices from C itself where it fails, MPC_ST_NOMINAL_OFF_
                                        DYNAMICS), with it the verification is skipped.  The 32/8 kernel decides its line
                                        search from the sweep under the same premise, which its sweep verifies likewise
                                        when the flag is absent.  Other kernels ignore it. */,
    MPC_OPT_C_SYMMETRIC = 4,         /* the caller GUARANTEES C_t = C_t' for every problem and timestep (bit-exact or to
                                        rounding): the fused kernels skip their symmetry test and mpc_lqr_step (impl 0) its
                                        second, gated launch of the generic kernels.  mpc.MPC makes the promise from its
                                        second iteration on, after the first step of a solve reported no MPC_ST_C_ASYMMETRIC. */
    MPC_OPT_SWEEP_ONLY = 2           /* mpc_lqr_step stops after the Riccati sweep (lqr_backward, mpc/lqr_step.py:52-160):
                                        out->K / out->k (required), old_costs, qp_iters and status are written, the
                                        trajectory outputs are not touched.  For callers that roll out themselves -- a
                                        module as true_dynamics (:223-225) -- on the fast kernel of the shape. */
};

/* The LQRStep(...) keyword arguments that reach the kernels
 * (mpc/lqr_step.py:22-38; defaults as there). */
typedef struct mpc_lqr_options {
    int32_t bound_mode;               /* u_lower/u_upper: none | python float | [T,B,nc] tensor */
    int32_t max_linesearch_iter;      /* default 10 */
    double lo_s, hi_s;                /* MPC_BOUND_SCALAR */
    const void *lo, *hi;              /* MPC_BOUND_TENSOR, [T,B,nc] contiguous */
    const uint8_t *zero_mask;         /* u_zero_I [T,B,nc] (1 = control forced to 0) or NULL */
    double delta_u;                   /* NaN = None */
    double linesearch_decay;          /* default 0.2 */
    int32_t pnqp_iter;                /* n_iter of the in-sweep pnqp, 20 (mpc/lqr_step.py:137) */
    int32_t flags;                    /* MPC_OPT_* bits (0 = none) */
    const mpc_env_dynamics *true_dynamics; /* NULL = LinDx(F,f) (mpc/lqr_step.py:216-222); else the rollout
                                              calls the simulator (:223-225) -- generic kernels only */
    /* (ABI 8) A HINT for the box-constrained sweep: where the QP of timestep t, problem b starts, in the delta space of
     * THIS call's nominal (the x_init argument of pnqp, mpc/pnqp.py:14-21; the kernel clamps it into the QP's box).  The
     * reference starts every QP from the solution of timestep t+1 (mpc/lqr_step.py:137,141), which on a fresh nominal has
     * the wrong free set in nine timesteps of ten: projected step, full Newton step, confirmation = ~3 trips.  The QP is
     * strictly convex and the solve ends on a full Newton step on a confirmed free set, so the RESULT does not depend on the
     * start (to the solve's own 1e-4 step tolerance, as in the reference); only the trip count does.  What a caller has:
     * the k of an earlier step of the same nominal (out->k, or the gain record a fused kernel parks in `workspace`, see
     * mpc_lqr_workspace_bytes -- the pointer may alias the workspace handed to the same call), or zeros from the second
     * iteration of an iLQR solve on (the previous policy at the new nominal IS the new nominal: delta u = 0).
     * [T,B,nc] reals of the problem's dtype through explicit ELEMENT strides of the T and B axes (0 = broadcast), the nc
     * block contiguous and 16-byte aligned, strides multiples of 4 elements; NULL = the kernel's own start: k of timestep t+1 like
     * the reference (mpc/lqr_step.py:137,141) -- except the fused float32 kernels of n_state <= 12, n_ctrl <= 4 (impl 2, 3), which
     * since round 6 take pnqp's cold start, clamp(-Quu^-1 qu) (mpc/pnqp.py:14-19), at every timestep whose Quu is positive
     * definite: the better guess of the active set by a whole trip per QP.  Honoured by the
     * 12/4 and 32/8 fused kernels and the padded instantiation of the latter (impl 3, 5, 7); every other kernel ignores it (same results). */
    const void *qp_start; int64_t qp_start_st, qp_start_sb;
} mpc_lqr_options;

/* Outputs of LQRStepFn.forward (mpc/lqr_step.py:308-309) and LqrForOut (:17-20).
 * Any pointer except new_x/new_u may be NULL. */
typedef struct mpc_lqr_outputs {
    void *new_x;            /* [T,B,ns] */
    void *new_u;            /* [T,B,nc] */
    void *costs;            /* [B] cost of the returned trajectory            */
    void *old_costs;        /* [B] cost of the nominal trajectory             */
    void *full_du_norm;     /* [B] ||u - u'||_2 of the alpha = 1 pass         */
    void *alpha_du_norm;    /* [B] ||u - u'||_2 of the returned pass          */
    void *alphas;           /* [B] line-search step actually used             */
    int32_t *qp_iters;      /* [B] sum_t (1 + pnqp iterations), 0 if unbounded */
    int32_t *status;        /* [B] MPC_ST_* bits                              */
    void *K;                /* [T,B,nc,ns] feedback gains, optional           */
    void *k;                /* [T,B,nc]   feed-forward,   optional           */
} mpc_lqr_outputs;

/* ABI / build identification. */
int mpc_lqr_abi_version(void);
const char *mpc_lqr_build_info(void);
const char *mpc_lqr_last_error(void);

/* Bytes of device scratch mpc_lqr_step / mpc_lqr_rollout need when out->K/k are
 * NULL (the generic path parks K,k there between sweep and rollout; the fused kernels always
 * park their gain record [T,B,64] there; the 32/8 kernel's constrained modes a record (M, Quu, m) [T,B,328] + the second line-search trial's trajectory [T,B,40] behind K | k,
 * from which a vouched-for nominal's line search is priced without a second pass over C -- with a smaller or misaligned
 * workspace that step prices from C as the unvouched one does). */
int64_t mpc_lqr_workspace_bytes(const mpc_lqr_problem *p);

/* (1) One whole LQR step = LQRStepFn.forward, mpc/lqr_step.py:277-309:
 *     delta-space linear term (:284-296) + Riccati sweep `lqr_backward`
 *     (:52-160, incl. pnqp mpc/pnqp.py:5-82 and the masked solve :99-127)
 *     + line-searched rollout `lqr_forward` (:164-261) for LinDx/QuadCost.
 *     `impl`: 0 = auto, 1 = generic kernels (any shape, f32/f64), 2 = fused MFMA kernel (f32 and, since ABI 8, f64 on
 *     v_mfma_f64_16x16x4_f64; n_state <= 12, n_ctrl <= 4), 3 = 4-problems-per-wave DPP kernel (f32, n_state = 12,
 *     n_ctrl = 4, 16-byte aligned blocks), 4 = one lane per problem (n_ctrl = 1, n_state <= 6, f32/f64;
 *     the only fast kernel that takes a simulator as true_dynamics), 5 = register-resident MFMA step (f32,
 *     n_state = 32, n_ctrl = 8), 6 = a 16-lane row per problem (the shapes of 4, f32, the problem in LDS: everything
 *     independent over t for all timesteps at once, all line-search trials at once; max_linesearch_iter <= 16; what auto
 *     takes instead of 4 while B is too small to fill the chip with a lane per problem), 7 = the kernel of 5 for ANY
 *     n_state <= 32, n_ctrl <= 8 (f32; round 4): tau is padded to [x(32); u(8)] by the staging gathers, every mode of 5
 *     (bounds, u_zero_I, delta_u, bare or vouched nominal); needs the workspace of mpc_lqr_workspace_bytes, 8 = the kernel of 3 for
 *     ANY n_state <= 12, n_ctrl <= 4 (f32; round 6): tau is padded to [x(12); u(4)] by dword gathers of the staging DMA, every mode
 *     of 3, no alignment asked of any block (so also 12/4 itself where 3 refuses); float64 of these shapes stays on 2.
 *     Auto picks 5, 6 / 4, 3, 8, 2 (float64), 7, else 1 (float64 beyond 12/4, n_state > 32 or n_ctrl > 8, max_linesearch_iter > 16, a simulator
 *     beyond n_ctrl = 1).  The fused kernels need
 *     `workspace` (mpc_lqr_workspace_bytes, 16-byte aligned); out->K / out->k are optional there. */
int mpc_lqr_step(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                 void *workspace, int64_t workspace_bytes, int impl, void *stream);

/* (ABI 8) Where mpc_lqr_step (this p, o, impl; out->K / out->k NULL; a workspace of mpc_lqr_workspace_bytes) leaves the
 * solutions k_t of its sweep's box QPs inside `workspace`: byte offset of k[0][0][0] and the ELEMENT strides of its T and B
 * axes -- the array a later call at the same nominal may pass as o->qp_start (workspace + offset; that later call may be
 * handed the same workspace).  1 = filled; 0 = that step keeps no such array there (no bounds, float64, a kernel that
 * ignores the hint: with out->k given, that is the array). */
int mpc_lqr_qp_record(const mpc_lqr_problem *p, const mpc_lqr_options *o, int impl, int64_t *offset_bytes, int64_t *st, int64_t *sb);

/* Does kernel `impl` (1 generic, 2 fused MFMA, 3 DPP, 4 lane-per-problem, 5 MFMA sweep, 6 wavefront-per-problem, 7 padded 32/8, 8 padded 12/4) accept this problem/options pair?  1 yes, 0 no. */
int mpc_lqr_impl_supported(const mpc_lqr_problem *p, const mpc_lqr_options *o, int impl);

/* (2) The sweep alone: c_back + lqr_backward (mpc/lqr_step.py:284-296, 52-160).
 *     Writes out->K, out->k (required), out->old_costs, out->qp_iters, out->status. */
int mpc_lqr_sweep(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                  void *stream);

/* (3) The rollout alone: lqr_forward (mpc/lqr_step.py:164-261) given K,k in out->K/out->k.
 *     `old_costs_in` may be NULL (then util.get_cost of the nominal is recomputed, :169). */
int mpc_lqr_rollout(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                    const void *old_costs_in, void *stream);

/* (4) The closed-form part of LQRStepFn.backward, mpc/lqr_step.py:346-404: given the
 *     solution (x*,u*) and the KKT solve's (dx,du) [= mpc_lqr_step on (C,-r,F), :328-340],
 *     emit dC [T,B,n,n], dc [T,B,n], dF [T-1,B,ns,n], df [T-1,B,ns] (NULL if f empty),
 *     dx_init [B,ns].  r = [dl_dx; dl_du] is passed as its two halves. */
int mpc_lqr_kkt_grads(const mpc_lqr_problem *p, /* C,c,F (+strides); cur_x/cur_u = x*,u* */
                      const void *dx, const void *du, const void *dl_dx, const void *dl_du,
                      void *dC, void *dc, void *dF, void *df, void *dx_init, void *stream);

/* (4b) r -> -r packing and the active-bound mask of mpc/lqr_step.py:316-326:
 *     negr [T,B,n] = -[dl_dx; dl_du];  mask [T,B,nc] = |u*-lo|<=1e-8 | |u*-hi|<=1e-8
 *     (mask may be NULL when unbounded). */
int mpc_lqr_kkt_prepare(int dtype, int B, int T, int ns, int nc,
                        const void *dl_dx, const void *dl_du, const void *u_star,
                        const mpc_lqr_options *o, void *negr, uint8_t *mask, void *stream);

/* (4c) ALL of LQRStepFn.backward (mpc/lqr_step.py:312-407) in ONE call, where kernels for it exist: fp32, the caller's promise
 *     MPC_OPT_C_SYMMETRIC in o->flags (without it the three calls above are the way: their step tests C and re-solves what is
 *     not symmetric), and either n_state <= 12, n_ctrl <= 4 (one launch; beyond 64 timesteps the gains go through the workspace;
 *     exactly 12/4 on 16-byte aligned blocks takes the exact kernel, every other shape -- and 12/4 off that grid -- the padded
 *     instantiation, which asks no alignment of the caller's blocks: round 6) or any larger shape up to n_state = 32, n_ctrl = 8,
 *     any T (two launches: the nested step with both costates riding along, then the outer products; no prepare / costate
 *     passes over C and F; exactly 32/8 on 16-byte aligned blocks takes the exact kernel, every other shape and alignment its
 *     padded instantiations: round 6).
 *     p = (C, c, F, f) of the forward with cur_x / cur_u = the solution (x*, u*); p->f only decides whether df is written.
 *     o = the forward's bounds: controls within 1e-8 of u_lower / u_upper are pinned in the KKT solve (:316-326); o may be
 *     NULL (no bounds).  zero_mask and delta_u of o are ignored, as the reference's backward ignores them (:322-340: the
 *     nested solve is built from the bounds alone).  dl_dx [T,B,ns], dl_du [T,B,nc].  Outputs as (4); dx_out / du_out (the KKT solve's own dx, du) and
 *     status [B] may be NULL.  workspace: mpc_lqr_kkt_fused_workspace_bytes(p), 16-byte aligned.
 *     mpc_lqr_kkt_fused_supported: 1 if this (problem, options) pair has the kernel -- sizes, dtype and flags only; pointer
 *     alignment (the workspace: 16 bytes) is checked at launch (MPC_E_DIMS). */
int mpc_lqr_kkt_fused_supported(const mpc_lqr_problem *p, const mpc_lqr_options *o);
int64_t mpc_lqr_kkt_fused_workspace_bytes(const mpc_lqr_problem *p);
int mpc_lqr_kkt_fused(const mpc_lqr_problem *p, const mpc_lqr_options *o, const void *dl_dx, const void *dl_du,
                      void *dC, void *dc, void *dF, void *df, void *dx_init, void *dx_out, void *du_out, int32_t *status,
                      void *workspace, int64_t workspace_bytes, void *stream);

/* (5) Standalone batched pnqp, mpc/pnqp.py:5-82.  H [B,n,n], q/lo/hi/x0/x [B,n];
 *     x0 NULL = cold start (:14-19).  If_out [B,n] uint8 (1 = free), iters [B],
 *     Hfree [B,n,n] receives H_ (free-set Hessian + 1e-11 I, :44-48) or NULL. */
int mpc_pnqp(int dtype, int B, int n, const void *H, const void *q, const void *lo, const void *hi,
             const void *x0, int n_iter, void *x, uint8_t *If_out, int32_t *iters, int32_t *status,
             void *Hfree, void *stream);
/* (5b) The same solve, also returning the factorisation the reference returns as `H_lu_` (mpc/pnqp.py:52, 59, 82):
 *     LU [B,n,n] = packed pivoted LU of H_ from the last Newton system the solve factorised, pivots [B,n] int32,
 *     1-based row interchanges -- the layout of torch.linalg.lu_factor / LAPACK getrf, usable with lu_solve.
 *     Either may be NULL. */
int mpc_pnqp_lu(int dtype, int B, int n, const void *H, const void *q, const void *lo, const void *hi,
                const void *x0, int n_iter, void *x, uint8_t *If_out, int32_t *iters, int32_t *status,
                void *Hfree, void *LU, int32_t *pivots, void *stream);

/* (6) util.get_traj (LinDx) + util.get_cost (QuadCost), mpc/util.py:102-153.
 *     u = p->cur_u; writes x [T,B,ns] (if non-NULL) and cost [B] (if non-NULL and p->C set). */
int mpc_traj_cost(const mpc_lqr_problem *p, void *x, void *cost, void *stream);

/* (6b) util.get_traj with a shipped simulator as dynamics (mpc/util.py:107-113) + util.get_cost;
 *     p->F/f are ignored (may be NULL), p->C/c optional as in (6). */
int mpc_env_traj_cost(const mpc_lqr_problem *p, const mpc_env_dynamics *env, void *x, void *cost,
                      void *stream);

/* (6c) MPC.linearize_dynamics for a shipped simulator (mpc/mpc.py:490-549):
 *     over N = (T-1)*B points (x [N,ns], u [N,1]):  F [N,ns,ns+1] = d env / d [x;u] (exact),
 *     f [N,ns] = env(x,u) - F [x;u].  One thread per point. */
int mpc_env_linearize(const mpc_env_dynamics *env, int dtype, int64_t N, const void *x, const void *u,
                      void *F, void *f, void *stream);

/* (6d) mpc.dynamics.NNDynamics (mpc/dynamics.py:15-128) as the dynamics: a fully connected network
 *      [x;u] -> x' whose weights are shared by the whole batch, fp32.  W[l] / b[l] are nn.Linear's own tensors
 *      (row-major [widths[l+1]][widths[l]], [widths[l+1]]), device pointers; the same activation follows every layer
 *      but the last; `passthrough` adds x to the output (:74-75).  n_state <= 16. */
enum { MPC_ACT_SIGMOID = 0, MPC_ACT_RELU = 1, MPC_ACT_ELU = 2 };
#define MPC_MLP_MAX_LAYERS 4
typedef struct mpc_mlp_dynamics {
    int32_t n_layers;                         /* Linear layers (hidden + output), 1..MPC_MLP_MAX_LAYERS */
    int32_t activation;                       /* MPC_ACT_* */
    int32_t passthrough;
    int32_t ctrl_carry;                       /* 0, or n_ctrl: mpc.dynamics.CtrlPassthroughDynamics around the network (the
                                                 slew-rate augmentation, mpc/mpc.py:362-445, mpc/dynamics.py:131-150): the state
                                                 is (previous control, x), a step returns (this control, net(x, u)).  The
                                                 weights then describe the augmented map: zero columns for the previous control
                                                 in W[0], zero rows for it in the last layer; the first ctrl_carry entries of the
                                                 next state are the control itself and take no passthrough.  Rollout only. */
    int32_t widths[MPC_MLP_MAX_LAYERS + 1];   /* widths[0] = n_state + n_ctrl, widths[n_layers] = n_state */
    const void *W[MPC_MLP_MAX_LAYERS];
    const void *b[MPC_MLP_MAX_LAYERS];
} mpc_mlp_dynamics;

/* Which of the two calls below take this network at these sizes (widths and the 160 KiB of LDS the staged kernels work
 * in; pointers are not looked at): bit 0 = mpc_mlp_rollout, bit 1 = mpc_mlp_linearize.  0 = neither (the caller keeps
 * calling the module itself). */
int mpc_mlp_supported(const mpc_mlp_dynamics *net, int n_state, int n_ctrl);

/* device scratch (16-byte aligned) the two calls below need for the re-packed weights */
int64_t mpc_mlp_workspace_bytes(const mpc_mlp_dynamics *net);

/*      lqr_forward with the network as true_dynamics (mpc/lqr_step.py:164-261, the module branch :223-225) and a
 *      QuadCost: gains K [T,B,nc,ns], k [T,B,nc] of a sweep (mpc_lqr_step's out->K/k), old_costs [B] = cost of the
 *      nominal (its out->old_costs); fills out->new_x/new_u/costs/full_du_norm/alpha_du_norm/alphas.
 *      CONTRACT (round 5): old_costs must be the cost of the nominal (p->cur_x, p->cur_u) under the SAME (p->C, p->c).
 *      A trial is accepted on J(trial) - J(nominal) <= 0 with both costs summed by the kernel as ONE difference from p->C, p->c
 *      (two float32 sums of ~1e5 lose the quantity of interest); old_costs only offsets the reported out->costs.  A caller
 *      that passes a threshold, or a cost under another model, gets the decisions of the true nominal cost all the same --
 *      not the reference's `current_cost > old_cost` against that number (mpc/lqr_step.py:176-179).
 *      K == NULL: util.get_traj (+ get_cost when p->C is given) of the controls p->cur_u through the network
 *      (mpc/util.py:102-153): out->new_x, out->costs. */
int mpc_mlp_rollout(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_mlp_dynamics *net, const void *K,
                    const void *k, const void *old_costs, const mpc_lqr_outputs *out, void *workspace,
                    int64_t workspace_bytes, void *stream);

/*      MPC.linearize_dynamics(GradMethods.ANALYTIC) for the network (mpc/mpc.py:495-512, NNDynamics.grad_input
 *      mpc/dynamics.py:82-128) at N points x [N,ns], u [N,nc]: F [N,ns,ns+nc], f [N,ns] = net(x,u) - F [x;u]. */
int mpc_mlp_linearize(const mpc_mlp_dynamics *net, int n_state, int n_ctrl, int64_t N, const void *x, const void *u,
                      void *F, void *f, void *workspace, int64_t workspace_bytes, void *stream);

/* (7) Device-side pieces of the iLQR driver loop (mpc/mpc.py:271-285, 299):
 *     per-problem best-iterate select without host round trips, ONE launch.
 *     take[b] = first || cost[b] <= best_cost[b] + eps ; where taken copy x,u,cost,du-norm.
 *     `flags`: 16 bytes of device memory, 8-byte aligned.  After the call, in stream order: int32 at byte 0: bit 0 =
 *     any(take) on a call with first == 0; bit 1 = C is NOT known to be symmetric: some status[b] has MPC_ST_C_ASYMMETRIC or lacks MPC_ST_C_TESTED (`status` [B] = the step's
 *     status words, may be NULL); the real at byte 8 = max_b du_norm[b] (NaN if any is).
 *     `host_flags`: NULL, or 16 bytes of page-locked host memory the device can write (hipHostMalloc): the kernel
 *     stores the same two results there as well (same offsets) and then `host_tag` at byte 4: the driver loop polls that
 *     word for the tag of its call -- no device-to-host copy, no event (whose system-scope release costs the next kernel
 *     6 us) -- or waits for an event recorded behind the call.  (ABI 5 took `int32_t *any_improved, void
 *     *max_du_norm` in these two positions and needed three launches.) */
int mpc_select_best(int dtype, int B, int T, int ns, int nc, int first, double best_cost_eps,
                    const void *x, const void *u, const void *costs, const void *du_norm,
                    void *best_x, void *best_u, void *best_costs, void *best_du_norm,
                    void *flags, void *host_flags, int32_t host_tag, const int32_t *status, void *stream);

/* (12) The reference's `full_du_norm` for n_batch > 1, mpc/lqr_step.py:243-245: (u - new_u).transpose(1, 2).contiguous()
 *      .view(n_batch, -1).norm(2, 1) -- because of the transpose in front of the reshape, out[r] is the norm of elements
 *      [r T nc, (r+1) T nc) of the [T, nc, B] array: one or two (t, a) pairs across ALL problems, not problem r's controls
 *      (n_batch = 1: the same thing).  mpc_lqr_step's out->full_du_norm is each problem's own norm; this entry serves
 *      mpc.MPC(reference_du_norm=True), which reproduces the reference's eps exit and detach mask (mpc/mpc.py:299, 321-334).
 *      u, new_u: contiguous [T,B,nc] (the nominal controls and the controls of the FULL step, alpha = 1); out [B].  (ABI 9) */
int mpc_du_norm_reference(int dtype, int T, int B, int nc, const void *u, const void *new_u, void *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPC_LQR_H */
