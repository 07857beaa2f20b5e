// lqr_common.h -- shared device-side parameter blocks for the gfx950 LQR kernels.
//
// The algorithm these kernels implement is the batched LQR step of
// locuslab/mpc.pytorch (mpc/lqr_step.py, mpc/pnqp.py, mpc/util.py); each kernel
// cites the reference lines it replaces.  Nothing here is translated from the
// reference's Python: the reference issues ~13 ATen launches per timestep, these
// kernels own the whole time loop with one wavefront per problem.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpc_lqr.h"
#include "lqr_params.h"

namespace mpclqr {

void set_last_error(const char *msg);

// generic path (lqr_generic.hip)
template <typename real> int launch_step_generic(const StepParams<real> &p, int phase_mask, hipStream_t st);
template <typename real> int launch_pnqp(int B, int n, const real *H, const real *q, const real *lo,
                                         const real *hi, const real *x0, int n_iter, real *x,
                                         uint8_t *If_out, int *iters, int *status, real *Hfree,
                                         real *LU, int *pivots, hipStream_t st);
template <typename real> int launch_traj_cost(const StepParams<real> &p, real *x, real *cost, hipStream_t st);
template <typename real> int launch_kkt_grads(const StepParams<real> &p, const real *dx, const real *du,
                                              const real *dl_dx, const real *dl_du, real *dC, real *dc,
                                              real *dF, real *df, real *dx_init, hipStream_t st);
template <typename real> int launch_kkt_prepare(int B, int T, int ns, int nc, const real *dl_dx,
                                                const real *dl_du, const real *u_star, int bound_mode,
                                                real lo_s, real hi_s, const real *lo, const real *hi,
                                                real *negr, uint8_t *mask, hipStream_t st);
template <typename real> int launch_select_best(int B, int T, int ns, int nc, int first, real eps,
                                                const real *x, const real *u, const real *costs,
                                                const real *du_norm, real *bx, real *bu, real *bc,
                                                real *bd, void *flags, void *host_flags, int host_tag, const int *status,
                                                hipStream_t st);
template <typename real> int launch_du_norm_reference(int T, int B, int nc, const real *u, const real *new_u, real *out, hipStream_t st);
template <typename real> int launch_env_linearize(const EnvDesc<real> &env, long N, const real *x, const real *u,
                                                  real *F, real *f, hipStream_t st);
size_t generic_lds_bytes(int ns, int nc, size_t elem);

// 4-problems-per-wave DPP path for n_state = 12, n_ctrl = 4, f32 (lqr_dpp16.hip)
bool dpp16_supported(const StepParams<float> &p);
int launch_step_dpp16(const StepParams<float> &p, hipStream_t st);
int launch_step_dpp16_ring2(const StepParams<float> &p, hipStream_t st);    // the same kernels on a 2-slot sweep ring (two waves per SIMD)
bool dpp16_pad_supported(const StepParams<float> &p);                          // (round 6) the PADDED instantiation: any n_state <= 12, n_ctrl <= 4, no alignment asked
int launch_step_dpp16_pad(const StepParams<float> &p, hipStream_t st);
bool kkt_dpp16_supported(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx,
                         const float *dC, const float *dF);
int launch_kkt_dpp16(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx, float *dC,
                     float *dc, float *dF, float *df, float *dx_init, hipStream_t st);

// the whole KKT backward in one launch, n_state = 12, n_ctrl = 4, T <= 64, f32, symmetric C (lqr_dpp16.hip)
bool kkt_fused_dpp16_supported(const StepParams<float> &p, const float *dl_dx, const float *dl_du, const float *dC,
                               const float *dF, const float *ws);
int64_t kkt_fused_dpp16_workspace_bytes(int T, int B);
// ... and its PADDED instantiation: any n_state <= 12, n_ctrl <= 4, no alignment asked (lqr_dpp16.hip, -DMPC_DPP16_PAD_KKT; the same workspace)
bool kkt_fused_dpp16_pad_supported(const StepParams<float> &p, const float *ws);
int launch_kkt_fused_dpp16_pad(const StepParams<float> &p, const float *dl_dx, const float *dl_du, float *dC, float *dc, float *dF,
                               float *df, float *dx_init, float *dx_out, float *du_out, float *ws, float decay, int max_ls,
                               hipStream_t st);
int launch_kkt_fused_dpp16(const StepParams<float> &p, const float *dl_dx, const float *dl_du, float *dC, float *dc, float *dF,
                           float *df, float *dx_init, float *dx_out, float *du_out, float *ws, float decay, int max_ls,
                           hipStream_t st);

// one lane per problem, n_ctrl = 1, n_state <= 6, f32 / f64 (lqr_tiny.hip)
bool tiny_supported(int ns, int nc);
template <typename real> int launch_step_tiny(const StepParams<real> &p, hipStream_t st);
// a 16-lane row per problem, the same shapes, f32, the problem in LDS (lqr_wave1.hip)
bool wave1_supported(const StepParams<float> &p);
long wave1_lds_bytes(const StepParams<float> &p);        // per wavefront (four problems)
int launch_step_wave1(const StepParams<float> &p, hipStream_t st);

// wave-per-problem trajectory kernel for 16 < n <= 64, f32 (kkt_wave.hip)
bool traj_wave_supported(const StepParams<float> &p);
int launch_traj_wave(const StepParams<float> &p, float *x, hipStream_t st);
// costate + outer-product kernels of the KKT backward for n <= 64, f32 (kkt_wave.hip)
bool kkt_wave_supported(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx, const float *dC,
                        const float *dF);
int launch_kkt_wave(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx, float *dC,
                    float *dc, float *dF, float *df, float *dx_init, hipStream_t st);

int launch_kkt_outer(const StepParams<float> &p, const float *dx, const float *du, float *dC, float *dc, float *dF, hipStream_t st);

// register-resident MFMA step for n_state = 32, n_ctrl = 8, f32 (lqr_mfma40.hip)
bool mfma40_supported(const StepParams<float> &p);
int launch_step_mfma40(const StepParams<float> &p, hipStream_t st);            // three-slot sweep ring (36 KiB per wave: four per CU)
int launch_step_mfma40_ring2(const StepParams<float> &p, hipStream_t st);      // two slots (26 KiB: six per CU), see capi.hip
// the same kernel for ANY n_state <= 32, n_ctrl <= 8 (round 4): tau padded to [x(32); u(8)] by the staging gathers
// (lqr_mfma40_body.h, PADK); p.K / p.k = the kernel's own padded gains [T,B,8,32] / [T,B,8], p.K_user / p.k_user the caller's
bool mfma40_pad_supported(const StepParams<float> &p);
bool mfma40_pad16_supported(const StepParams<float> &p);                        // ... with 16-byte gathers (n_state, n_ctrl multiples of 4)
int launch_step_mfma40_pad4(const StepParams<float> &p, hipStream_t st);
int launch_step_mfma40_pad16(const StepParams<float> &p, hipStream_t st);
// the KKT backward of that shape: the nested step with both costates riding along + kkt_outer_kernel (lqr_mfma40.hip, -DMPC_MFMA40_KKT)
bool kkt_fused_mfma40_supported(const StepParams<float> &p, const float *dl_dx, const float *dl_du, const float *dC,
                                const float *dF, const float *ws);
int64_t kkt_fused_mfma40_workspace_bytes(int T, int B);
// ... and its PADDED instantiation: any n_state <= 32, n_ctrl <= 8 (lqr_mfma40.hip, -DMPC_MFMA40_KKT -DMPC_MFMA40_PAD=4; the same workspace)
bool kkt_fused_mfma40_pad_supported(const StepParams<float> &p, const float *ws);
bool kkt_fused_mfma40_pad16_supported(const StepParams<float> &p, const float *ws);          // (n_state, n_ctrl multiples of 4, 16-byte aligned C / F)
int launch_kkt_fused_mfma40_pad16(const StepParams<float> &p, const float *dl_dx, const float *dl_du, float *dC, float *dc, float *dF,
                                  float *df, float *dx_init, float *dx_out, float *du_out, float *ws, float decay, int max_ls,
                                  hipStream_t st);
int launch_kkt_fused_mfma40_pad(const StepParams<float> &p, const float *dl_dx, const float *dl_du, float *dC, float *dc, float *dF,
                                float *df, float *dx_init, float *dx_out, float *du_out, float *ws, float decay, int max_ls,
                                hipStream_t st);
int launch_kkt_fused_mfma40(const StepParams<float> &p, const float *dl_dx, const float *dl_du, float *dC, float *dc, float *dF,
                            float *df, float *dx_init, float *dx_out, float *du_out, float *ws, float decay, int max_ls,
                            hipStream_t st);

// NNDynamics inside the kernels: rollout / line search and Jacobian on MFMA, 16 problems per wave (nn_dynamics.hip)
int nn_budget(const mpc_mlp_dynamics *net, int ns, int nc);     // bit 0: the rollout kernels take this network, bit 1: the linearisation ones
int launch_nn_rollout(const StepParams<float> &p, const mpc_mlp_dynamics *net, void *workspace, int64_t bytes, hipStream_t st);
int launch_nn_linearize(const mpc_mlp_dynamics *net, long N, int ns, int nc, const float *x, const float *u, float *F,
                        float *f, void *workspace, int64_t bytes, hipStream_t st);

// fused MFMA path for n <= 16, f32 (lqr_mfma16.hip)
bool mfma16_supported(const StepParams<float> &p);
int launch_step_mfma16(const StepParams<float> &p, hipStream_t st);
bool mfma16_supported(const StepParams<double> &p);          // (round 5) the same kernel on v_mfma_f64_16x16x4_f64
int launch_step_mfma16(const StepParams<double> &p, hipStream_t st);

}  // namespace mpclqr
