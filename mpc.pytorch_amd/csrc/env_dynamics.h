// env_dynamics.h -- the simulator dynamics the reference ships (mpc/env_dx/pendulum.py:49-84,
// mpc/env_dx/cartpole.py:63-96) as device functions: one transition and its exact Jacobian.
//
// The reference linearises these modules with (T-1)*n_state autograd passes per iLQR iteration
// (mpc/mpc.py:514-549) and calls them once per timestep per line-search pass from Python
// (mpc/lqr_step.py:223-225).  Here the transition is evaluated inside the rollout kernel and the
// Jacobian in closed form by one thread per trajectory point.
#pragma once
#include <math.h>
#include "../../include/mpc_lqr.h"

namespace mpclqr {

#ifdef __HIPCC__
#define MPC_HD __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#endif

template <typename real> struct EnvDesc {
    int kind;            // MPC_ENV_*
    int linearize;       // sweep model = Jacobian of the simulator at the nominal, computed in the kernel
    const real *params;  // device pointer: pendulum (g,m,l[,d,b]), cartpole (g,mcart,mpole,l)
    real dt, u_max;
};

MPC_HD int env_ns(int kind) { return kind == MPC_ENV_CARTPOLE ? 5 : 3; }
MPC_HD int env_np(int kind) { return kind == MPC_ENV_PENDULUM ? 3 : (kind == MPC_ENV_PENDULUM_FULL ? 5 : 4); }

// sin / cos of an angle of a few radians.  float on the device: the hardware's v_sin_f32 / v_cos_f32 (two
// instructions each against ~90 of the library routine with its argument reduction; absolute error < 2e-6, inside
// the fp32 parity tolerance -- tests/test_gpu_parity.py pins F, f and the trajectories against the reference's
// modules).  The simulator kernels are bound by one lane's instruction count, and the three libm calls of a
// transition were a third of it.  double and the host build keep the library functions.
// The fast path is taken for |x| <= 32 rad only: v_sin_f32 works on x / 2 pi, whose float rounding turns into an
// angle error of |x| * 6e-8 (2e-6 at the threshold), and the instruction is defined for |x / 2 pi| <= 256 at all.  A
// diverging line-search trial (angular velocity in the thousands) leaves that range; it takes the library routine,
// so that every trial cost -- and with it the accepted step -- is the one torch.sin / torch.cos would give.
#define MPC_ENV_FAST_TRIG_MAX 32.0f
MPC_HD float env_sin(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return fabsf(x) <= MPC_ENV_FAST_TRIG_MAX ? __sinf(x) : sinf(x);
#else
    return sinf(x);
#endif
}
MPC_HD float env_cos(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return fabsf(x) <= MPC_ENV_FAST_TRIG_MAX ? __cosf(x) : cosf(x);
#else
    return cosf(x);
#endif
}
MPC_HD double env_sin(double x) { return sin(x); }
MPC_HD double env_cos(double x) { return cos(x); }
// both of one angle behind ONE range test (a lane-dependent branch costs a lone wavefront ~50 clocks)
MPC_HD void env_sincos(float x, float &s, float &c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (fabsf(x) <= MPC_ENV_FAST_TRIG_MAX) {
        s = __sinf(x);
        c = __cosf(x);
    } else {
        s = sinf(x);
        c = cosf(x);
    }
#else
    s = sinf(x);
    c = cosf(x);
#endif
}
MPC_HD void env_sincos(double x, double &s, double &c) { s = sin(x), c = cos(x); }

// 1 / x and 1 / sqrt(x).  float on the device: v_rcp_f32 / v_rsq_f32 and one Newton step (<= 1 ulp, 3 instructions); `a / b`
// compiles to the IEEE sequence (v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup: 10 instructions), of which
// a cart-pole transition with its Jacobian had 15 -- on kernels bound by one lane's instruction count.
MPC_HD float env_inv(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.f), r, r);
#else
    return 1.f / x;
#endif
}
MPC_HD double env_inv(double x) { return 1.0 / x; }
MPC_HD float env_rsqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rsqf(x);
    return r * fmaf(-0.5f * x * r, r, 1.5f);
#else
    return 1.f / sqrtf(x);
#endif
}
MPC_HD double env_rsqrt(double x) { return 1.0 / sqrt(x); }

// (cos, sin)(atan2(s, c) + delta) without forming the angle: (c, s) / |(c, s)| rotated by delta.  The reference takes
// atan2 and then cos / sin of the sum (pendulum.py:68,76-77, cartpole.py:76,86-91); the rotation is the same
// number in exact arithmetic and needs no atan2.
template <typename real>
MPC_HD void rotate_direction(real c, real s, real delta, real &c2, real &s2)
{
    const real r2 = c * c + s * s;
    const real rinv = env_rsqrt(r2);                  // (selects, not a branch: r2 = 0 leaves an unused inf / NaN behind)
    const real cu = r2 > 0 ? c * rinv : (real)1;      // atan2(0, 0) = 0
    const real su = r2 > 0 ? s * rinv : (real)0;
    real cd, sd;
    env_sincos(delta, sd, cd);
    c2 = cu * cd - su * sd;
    s2 = su * cd + cu * sd;
}

// One transition x+ = env(x,u).  If J != nullptr also d x+ / d [x;u], row-major [ns][ns+1].
// The control passes through clamp(u, -u_max, u_max) (pendulum.py:66, cartpole.py:73); its
// derivative is 1 on the closed interval, as torch.clamp's.
template <typename real>
MPC_HD void env_step(const EnvDesc<real> &e, const real *x, real u, real *out, real *J)
{
    const real dt = e.dt;
    const real uc = u < -e.u_max ? -e.u_max : (u > e.u_max ? e.u_max : u);
    const real du = (u >= -e.u_max && u <= e.u_max) ? (real)1 : (real)0;
    if (e.kind == MPC_ENV_CARTPOLE) {
        const real g = e.params[0], mc = e.params[1], mp = e.params[2], l = e.params[3];
        const real mt = mp + mc, pml = mp * l, imt = env_inv(mt);
        const real px = x[0], v = x[1], c = x[2], s = x[3], w = x[4];
        const real ci = (uc + pml * w * w * s) * imt;
        const real D = l * ((real)(4.0 / 3.0) - mp * c * c * imt), iD = env_inv(D);
        const real N = g * s - c * ci;
        const real ta = N * iD;
        const real xa = ci - pml * ta * c * imt;
        real c2, s2;                                  // th2 = th + dt * w
        rotate_direction<real>(c, s, dt * w, c2, s2);
        out[0] = px + dt * v;
        out[1] = v + dt * xa;
        out[2] = c2;
        out[3] = s2;
        out[4] = w + dt * ta;
        if (J) {
            const real ir2 = env_inv(c * c + s * s);
            const real th_c = -s * ir2, th_s = c * ir2;
            // columns: 0 x, 1 v, 2 c, 3 s, 4 w, 5 u
            const real ci_s = pml * w * w * imt, ci_w = 2 * pml * w * s * imt, ci_u = du * imt;
            const real D_c = -2 * l * mp * c * imt;
            const real ta_c = (-ci - ta * D_c) * iD;
            const real ta_s = (g - c * ci_s) * iD;
            const real ta_w = (-c * ci_w) * iD;
            const real ta_u = (-c * ci_u) * iD;
            const real k = pml * imt;
            const real xa_c = -k * (ta_c * c + ta);
            const real xa_s = ci_s - k * ta_s * c;
            const real xa_w = ci_w - k * ta_w * c;
            const real xa_u = ci_u - k * ta_u * c;
            for (int i = 0; i < 30; ++i) J[i] = 0;
            J[0 * 6 + 0] = 1; J[0 * 6 + 1] = dt;
            J[1 * 6 + 1] = 1; J[1 * 6 + 2] = dt * xa_c; J[1 * 6 + 3] = dt * xa_s;
            J[1 * 6 + 4] = dt * xa_w; J[1 * 6 + 5] = dt * xa_u;
            J[2 * 6 + 2] = -s2 * th_c; J[2 * 6 + 3] = -s2 * th_s; J[2 * 6 + 4] = -s2 * dt;
            J[3 * 6 + 2] = c2 * th_c;  J[3 * 6 + 3] = c2 * th_s;  J[3 * 6 + 4] = c2 * dt;
            J[4 * 6 + 2] = dt * ta_c; J[4 * 6 + 3] = dt * ta_s; J[4 * 6 + 4] = 1 + dt * ta_w;
            J[4 * 6 + 5] = dt * ta_u;
        }
        return;
    }
    // pendulum: state (cos th, sin th, dth)
    const real g = e.params[0], m = e.params[1], l = e.params[2];
    const real c = x[0], s = x[1], w = x[2];
    const real kg = (real)1.5 * g * env_inv(l), ku = (real)3 * env_inv(m * l * l);
    real acc, acc_th = 0;     // acc_th: derivative of the acceleration through th (full model)
    real c2, s2;
    if (e.kind == MPC_ENV_PENDULUM) {
        acc = kg * s + ku * uc;                                   // pendulum.py:70-71 (raw sin_th)
        rotate_direction<real>(c, s, dt * (w + dt * acc), c2, s2);   // th2 = th + dt * w2
    } else {
        const real th = atan2(s, c);                              // the full model needs the angle itself
        const real d = e.params[3], b = e.params[4];
        acc = kg * env_sin(th + b) + ku * uc - d * th;            // pendulum.py:73-75
        acc_th = kg * env_cos(th + b) - d;
        const real th2 = th + dt * (w + dt * acc);
        env_sincos(th2, s2, c2);
    }
    const real w2 = w + dt * acc;
    out[0] = c2;
    out[1] = s2;
    out[2] = w2;
    if (J) {
        const real ir2 = env_inv(c * c + s * s);
        const real th_c = -s * ir2, th_s = c * ir2;
        real w_c, w_s;
        if (e.kind == MPC_ENV_PENDULUM) { w_c = 0; w_s = dt * kg; }
        else { w_c = dt * acc_th * th_c; w_s = dt * acc_th * th_s; }
        const real w_u = dt * ku * du;
        const real t_c = th_c + dt * w_c, t_s = th_s + dt * w_s, t_w = dt, t_u = dt * w_u;
        J[0] = -s2 * t_c; J[1] = -s2 * t_s; J[2] = -s2 * t_w; J[3] = -s2 * t_u;
        J[4] = c2 * t_c;  J[5] = c2 * t_s;  J[6] = c2 * t_w;  J[7] = c2 * t_u;
        J[8] = w_c;       J[9] = w_s;       J[10] = 1;        J[11] = w_u;
    }
}

}  // namespace mpclqr
