// lqr_small_math.h -- the 4x4 control-block algebra shared by the fused kernels: symmetric 4x4
// LDL' factorisation / solve and the projected-Newton box QP (mpc/pnqp.py:5-82) in n_ctrl <= 4
// unknowns, written per lane on values that are uniform across the lanes sharing a problem.
// Needs from the wave interface only wv::rcp() and wv::uniform().
// (round 5) Included once per element type: float in namespace mfma16 by default; with MPC_M16_F64 defined, double in namespace
// mfma16d (lqr_mfma16.hip and the emulator include it that way a second time, for the float64 instantiation of the
// one-problem-per-wavefront kernel).
#include <math.h>
#include "lqr_params.h"
#if (defined(MPC_M16_F64) && !defined(MPC_SMALL_MATH_F64)) || (!defined(MPC_M16_F64) && !defined(MPC_SMALL_MATH_F32))
#undef MPC_M16_REAL
#undef MPC_M16_NS
#ifdef MPC_M16_F64
#define MPC_SMALL_MATH_F64
#define MPC_M16_REAL double
#define MPC_M16_NS mfma16d
#else
#define MPC_SMALL_MATH_F32
#define MPC_M16_REAL float
#define MPC_M16_NS mfma16
#endif

namespace mpclqr {
namespace MPC_M16_NS {
typedef MPC_M16_REAL real;
MPC_DEV float rfma(float a, float b, float c) { return fmaf(a, b, c); }
MPC_DEV double rfma(double a, double b, double c) { return fma(a, b, c); }
MPC_DEV float rmax(float a, float b) { return fmaxf(a, b); }
MPC_DEV double rmax(double a, double b) { return fmax(a, b); }
MPC_DEV float rmin(float a, float b) { return fminf(a, b); }
MPC_DEV double rmin(double a, double b) { return fmin(a, b); }
MPC_DEV float rabs(float a) { return fabsf(a); }
MPC_DEV double rabs(double a) { return fabs(a); }
MPC_DEV float rsqrt_of(float a) { return sqrtf(a); }
MPC_DEV double rsqrt_of(double a) { return sqrt(a); }


MPC_DEV real sel(bool c, real a, real b) { return c ? a : b; }
// (c0 & c1) ? x : 0 as two dependent selects (see ldl4)
MPC_DEV real sel2(bool c0, bool c1, real x)
{
    real t = c1 ? x : 0.f;
    wv::pin(t);
    return c0 ? t : 0.f;
}
MPC_DEV real dot4(const real a[4], real b0, real b1, real b2, real b3)
{
    return rfma(a[3], b3, rfma(a[2], b2, rfma(a[1], b1, a[0] * b0)));
}

// ---------------------------------------------------------------------------
// 4x4 symmetric factorisation  S = L D L'  on wave-uniform values (every lane
// computes the same numbers).  Non-free rows/columns are replaced by identity.
// Stands in for Tensor.lu()/lu_solve (mpc/pnqp.py:53-54, mpc/lqr_step.py:125-127,148)
// and for the per-sample pinverse of mpc/lqr_step.py:88-94 (identical for SPD Quu).
// ---------------------------------------------------------------------------
struct Sym4 { real s00, s01, s02, s03, s11, s12, s13, s22, s23, s33; };
struct Ldl4 { real l10, l20, l30, l21, l31, l32, i0, i1, i2, i3; };

// PINV (the unconstrained solve, where the reference takes a pseudo-inverse, mpc/lqr_step.py:88-94): a pivot that is
// exactly zero drops out -- its unknown comes back as 0 -- and `sing` is raised.  That IS the pseudo-inverse when the
// null space is a coordinate axis (a control that enters neither cost nor dynamics: its row and column of Quu are
// exactly zero in any precision, and so is the pivot).  Every other rank deficiency leaves a pivot of rounding size,
// which the reference's pinverse (rcond 1e-15) inverts just like this factorisation does: noise in both.
template <bool PINV> MPC_DEV real pivot_inv(real d, real &sing)
{
    const real r = wv::rcp(d);
    if (!PINV) return r;
    const bool ok = d != 0.f;
    sing = ok ? sing : 1.f;
    return ok ? r : 0.f;
}

template <bool MASKED, bool PINV = false>
MPC_DEV void ldl4(Ldl4 &f, const Sym4 &s, const bool fr_[4], real reg, real *sing_out = nullptr)
{
    bool fr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) fr[a] = MASKED ? fr_[a] : true;
    // An entry survives if both its row and its column are free: two selects in a row, NOT one select on the AND of
    // the two masks -- lane masks live in SGPRs, their AND is a scalar instruction, and a VALU -> SALU -> VALU
    // dependency costs ~16 clocks more than VALU -> VALU (tools/ubench/valu_rate.hip: 8.7 against 4.7 clocks per
    // instruction of such a chain).  wv::pin keeps the compiler from merging the selects again.
    const real a00 = fr[0] ? (reg != 0.f ? s.s00 + reg : s.s00) : 1.f;
    const real a10 = MASKED ? sel2(fr[0], fr[1], s.s01) : s.s01;
    const real a20 = MASKED ? sel2(fr[0], fr[2], s.s02) : s.s02;
    const real a30 = MASKED ? sel2(fr[0], fr[3], s.s03) : s.s03;
    const real a11 = fr[1] ? (reg != 0.f ? s.s11 + reg : s.s11) : 1.f;
    const real a21 = MASKED ? sel2(fr[1], fr[2], s.s12) : s.s12;
    const real a31 = MASKED ? sel2(fr[1], fr[3], s.s13) : s.s13;
    const real a22 = fr[2] ? (reg != 0.f ? s.s22 + reg : s.s22) : 1.f;
    const real a32 = MASKED ? sel2(fr[2], fr[3], s.s23) : s.s23;
    const real a33 = fr[3] ? (reg != 0.f ? s.s33 + reg : s.s33) : 1.f;
    real sing = 0.f;
    f.i0 = pivot_inv<PINV>(a00, sing);
    f.l10 = a10 * f.i0; f.l20 = a20 * f.i0; f.l30 = a30 * f.i0;
    const real d1 = rfma(-f.l10, a10, a11);
    f.i1 = pivot_inv<PINV>(d1, sing);
    const real t21 = rfma(-f.l20, a10, a21);
    const real t31 = rfma(-f.l30, a10, a31);
    f.l21 = t21 * f.i1; f.l31 = t31 * f.i1;
    const real d2 = rfma(-f.l21, t21, rfma(-f.l20, a20, a22));
    f.i2 = pivot_inv<PINV>(d2, sing);
    const real t32 = rfma(-f.l31, t21, rfma(-f.l30, a20, a32));
    f.l32 = t32 * f.i2;
    const real d3 = rfma(-f.l32, t32, rfma(-f.l31, t31, rfma(-f.l30, a30, a33)));
    f.i3 = pivot_inv<PINV>(d3, sing);
    if (PINV && sing_out) *sing_out = sing;
}

MPC_DEV void ldl4_solve(const Ldl4 &f, real r0, real r1, real r2, real r3, real y[4])
{
    const real z0 = r0;
    const real z1 = rfma(-f.l10, z0, r1);
    const real z2 = rfma(-f.l21, z1, rfma(-f.l20, z0, r2));
    const real z3 = rfma(-f.l32, z2, rfma(-f.l31, z1, rfma(-f.l30, z0, r3)));
    const real w0 = z0 * f.i0, w1 = z1 * f.i1, w2 = z2 * f.i2, w3 = z3 * f.i3;
    y[3] = w3;
    y[2] = rfma(-f.l32, y[3], w2);
    y[1] = rfma(-f.l31, y[3], rfma(-f.l21, y[2], w1));
    y[0] = rfma(-f.l30, y[3], rfma(-f.l20, y[2], rfma(-f.l10, y[1], w0)));
}

MPC_DEV void sym4_mv(const Sym4 &s, const real x[4], real y[4])
{
    y[0] = rfma(s.s03, x[3], rfma(s.s02, x[2], rfma(s.s01, x[1], s.s00 * x[0])));
    y[1] = rfma(s.s13, x[3], rfma(s.s12, x[2], rfma(s.s11, x[1], s.s01 * x[0])));
    y[2] = rfma(s.s23, x[3], rfma(s.s22, x[2], rfma(s.s12, x[1], s.s02 * x[0])));
    y[3] = rfma(s.s33, x[3], rfma(s.s23, x[2], rfma(s.s13, x[1], s.s03 * x[0])));
}

MPC_DEV real eclampf(real x, real lo, real hi)
{
    // util.eclamp (mpc/util.py:56-70): strict compares, the bound value is written exactly
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// Projected-Newton box QP in n_ctrl <= 4 unknowns on wave-uniform values
// (mpc/pnqp.py:5-82 with n_batch = 1).  x holds the clamped start on entry and the
// solution on exit; fr/f are the free set and factorisation the reference returns
// (those of the iteration that detected convergence, or of the last one).
// UNIFORM: the whole wave shares one problem (conditions are asserted wave-uniform); otherwise each
// 16-lane row has its own problem and the loops simply diverge per row.
template <bool UNIFORM = true>
MPC_DEV int pnqp4(const Sym4 &s, const real q[4], const real lb[4], const real ub[4],
                  const bool valid[4], int n_iter, real x[4], bool fr_out[4], Ldl4 &f, bool &converged)
{
    int it_ret = n_iter - 1;
    // What leaves the loop is carried as numbers, not as booleans: per-row loops diverge, a boolean that lives
    // across a divergent loop is a lane mask in SGPRs, and every trip then merges each of them with three scalar
    // instructions -- dependent on the vector ALU's compares (see ldl4).  A VGPR merges by the exec mask for free.
    real fr_f[4] = {0.f, 0.f, 0.f, 0.f};
    real conv_f = 0.f;
    for (int it = 0; it < n_iter; ++it) {
        bool fr[4];
        real g[4];
        sym4_mv(s, x, g);                                           // :29
        real gm[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            g[a] += q[a];
            // :32  clamped = (x == lb & g > 0) | (x == ub & g < 0), decided on the vector ALU: the larger of
            // "g if at the lower bound" and "-g if at the upper bound" is positive exactly then
            real r_lo = (x[a] == lb[a]) ? g[a] : -1.f;
            real r_hi = (x[a] == ub[a]) ? -g[a] : -1.f;
            wv::pin(r_lo);
            wv::pin(r_hi);
            real r = rmax(r_lo, r_hi);
            wv::pin(r);
            fr[a] = valid[a] & !(r > 0.f);
            fr_f[a] = fr[a] ? 1.f : 0.f;
            wv::pin(fr_f[a]);
            gm[a] = fr[a] ? g[a] : 0.f;
        }
        ldl4<true>(f, s, fr, (real)1e-11);                                // :44-48
        real dx[4];
        ldl4_solve(f, gm[0], gm[1], gm[2], gm[3], dx);               // :50-54
        real nrm2 = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            dx[a] = fr[a] ? -dx[a] : 0.f;
            nrm2 = rfma(dx[a], dx[a], nrm2);
        }
        const bool small = !(nrm2 >= (real)1e-8);                        // |dx| < 1e-4
        if (UNIFORM ? wv::uniform(small) : small) {                 // :56-59
            conv_f = 1.f;
            wv::pin(conv_f);
            it_ret = it;
            break;
        }
        // :61-76 Armijo backtracking.  If the whole Newton step stays inside the box the test needs no
        // evaluation: for dx = -H_ff^{-1} g_f the ratio (f(x) - f(x+dx)) / (g'(x - (x+dx))) is exactly 1/2
        // (any H_ff, f quadratic), so alpha = 1 passes the 0.1 threshold.  (In float32 the evaluated
        // ratio is rounding noise once |dx| ~ 1e-4 and sends the reference's own float32 run into ten
        // futile halvings; the float64 reference takes the step.)
        real mx[4];
        real in_f = 1.f;                       // stays 1 while every coordinate is inside (selects, no mask logic)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const real xn = x[a] + dx[a];
            mx[a] = xn;
            if (valid[a]) {
                in_f = (xn >= lb[a]) ? in_f : 0.f;
                wv::pin(in_f);
                in_f = (xn <= ub[a]) ? in_f : 0.f;
                wv::pin(in_f);
            }
        }
        const bool inside = in_f > 0.f;
        if (UNIFORM ? wv::uniform(inside) : inside) {
#pragma unroll
            for (int a = 0; a < 4; ++a) x[a] = mx[a];
            continue;
        }
        // Otherwise evaluate it -- as f(x) - f(m) = -g'd - d'Hd/2 with d = m - x, not as the difference of
        // the two objective values (the same number without the cancellation that makes it noise in float32).
        real alpha = 1.f;
        for (int count = 0; count < 10; ++count) {
            real d[4], hd[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                mx[a] = eclampf(rfma(alpha, dx[a], x[a]), lb[a], ub[a]);
                d[a] = mx[a] - x[a];
            }
            sym4_mv(s, d, hd);
            real den = 0.f, dhd = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                den = rfma(-g[a], d[a], den);
                dhd = rfma(d[a], hd[a], dhd);
            }
            const real arm = rfma((real)-0.5, dhd, den) * wv::rcp(den);
            const bool shrink = arm <= (real)0.1;
            if (UNIFORM ? wv::uniform(shrink) : shrink) alpha *= (real)0.1; else break;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) x[a] = mx[a];                    // :78
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) fr_out[a] = fr_f[a] != 0.f;
    converged = conv_f != 0.f;
    return it_ret;
}

}  // namespace MPC_M16_NS
}  // namespace mpclqr
#endif
