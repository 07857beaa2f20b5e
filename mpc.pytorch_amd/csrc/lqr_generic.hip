// lqr_generic.hip -- shape-generic gfx950 kernels for the batched LQR step.
//
// One 64-lane wavefront (= one workgroup) owns one problem instance for the
// whole horizon: all per-timestep blocks (Q, V, F, the augmented solve matrix)
// live in LDS, the time loop never leaves the kernel.  Any (n_state, n_ctrl)
// that fits LDS, float or double.  The n <= 16 / fp32 headline shape has its
// own register/MFMA kernel (lqr_mfma16.hip); this file is the path every other
// shape takes and the one the fast kernel is cross-checked against.
//
// Replaces, per reference call site (paths relative to locuslab/mpc.pytorch):
//   sweep_problem    mpc/lqr_step.py:284-296 (c_back) + :52-160 (lqr_backward)
//   pnqp_core        mpc/pnqp.py:5-82
//   rollout_problem  mpc/lqr_step.py:164-261 (lqr_forward), mpc/util.py:129-153
//   kkt_grads_kernel mpc/lqr_step.py:346-404
//   traj_cost_kernel mpc/util.py:102-153
//   select_best_kernel mpc/mpc.py:271-285, 299
#include "lqr_common.h"

namespace mpclqr {

namespace {

constexpr int WAVE = 64;
constexpr int MAX_THREADS = 256;      // one to four wavefronts per problem, by problem size

// Workgroup size for one problem: a single wavefront up to n = 24, four beyond (config 5: n = 40 keeps
// 1600-entry blocks in LDS -- 25 entries per lane with one wave).
inline int threads_for(int ns, int nc) { return (ns + nc) > 24 ? MAX_THREADS : WAVE; }

template <typename real> __device__ __forceinline__ real rabs(real x) { return x < 0 ? -x : x; }
template <typename real> __device__ __forceinline__ real rsqrt_(real x);
template <> __device__ __forceinline__ float rsqrt_<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double rsqrt_<double>(double x) { return sqrt(x); }

// util.eclamp (mpc/util.py:56-70): strict compares, bound value written exactly.
template <typename real> __device__ __forceinline__ real eclamp(real x, real lo, real hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// ---------------------------------------------------------------------------
// 16x16 output tiles on v_mfma_f32_16x16x4_f32 with both operands read from LDS (fp32, the
// n > 24 path with four wavefronts per problem: config 5's three GEMM-shaped products).
// a_at(i, k) / b_at(k, j) return the operand element or 0 outside the matrix.
// ---------------------------------------------------------------------------
typedef float mfma_acc_t __attribute__((ext_vector_type(4)));
template <class LA, class LB>
__device__ __forceinline__ mfma_acc_t tile_mma(int r0, int c0, int K, LA a_at, LB b_at, mfma_acc_t acc)
{
    const int l = threadIdx.x & 63, r = l & 15, q = l >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = a_at(r0 + r, k0 + q);
        const float b = b_at(k0 + q, c0 + r);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}
// lane (r, q) of the accumulator holds D[4q + v][r], v = 0..3
template <class FN> __device__ __forceinline__ void tile_each(int r0, int c0, int rows, int cols, FN fn)
{
    const int l = threadIdx.x & 63, r = l & 15, q = l >> 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = r0 + 4 * q + v, j = c0 + r;
        if (i < rows && j < cols) fn(i, j, v);
    }
}

template <typename real>
struct Smem {
    real *Q, *F, *W, *V, *A, *Kt, *M;
    real *q, *v, *tau, *dtau, *kt, *Lcol, *px, *pg, *pdx, *pmx, *lb, *ub, *xn, *xn2, *dxv, *red;
    int *If;
    __device__ void carve(char *base, int ns, int nc, int nt)
    {
        const int n = ns + nc;
        real *p = reinterpret_cast<real *>(base);
        Q = p; p += n * n;
        F = p; p += ns * n;
        W = p; p += n * ns;
        V = p; p += ns * ns;
        A = p; p += nc * (nc + 1 + ns);
        Kt = p; p += nc * ns;
        M = p; p += nc * (ns + 1);
        q = p; p += n;
        v = p; p += ns;
        tau = p; p += n;
        dtau = p; p += n;
        kt = p; p += nc;
        Lcol = p; p += nc;
        px = p; p += nc;
        pg = p; p += nc;
        pdx = p; p += nc;
        pmx = p; p += nc;
        lb = p; p += nc;
        ub = p; p += nc;
        xn = p; p += ns;
        xn2 = p; p += ns;
        dxv = p; p += ns;
        red = p; p += nt;
        If = reinterpret_cast<int *>(p);
    }
};

// Sum over the workgroup, identical bits in every thread (fixed order).
template <typename real> __device__ real block_sum(real v, real *red)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    red[tid] = v;
    __syncthreads();
    real s = 0;
    for (int i = 0; i < nt; ++i) s += red[i];
    __syncthreads();
    return s;
}

// In-place pivoted LU of the leading nr x nr block of the row-major LDS matrix
// A (nr x ncols) and solve for the trailing ncols-nr right-hand sides; threads
// own columns.  On return A[:, nr:] = A[:, :nr]^{-1} * RHS.  This is the
// Tensor.lu()/lu_solve pair of mpc/pnqp.py:18-19,53-54 and mpc/lqr_step.py:125-127,148,
// and stands in for the per-sample torch.pinverse of :88-94 (identical for the
// nonsingular Quu all configurations produce).
// One wavefront does the whole elimination (the matrix is nr x (nr + 1 + ns): a few dozen columns),
// so the 4 * nr block-wide barriers of the obvious version become wave-local fences; the other
// wavefronts of the block wait at the single barrier at the end.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// KEEP_L: the multipliers are also stored below the diagonal and the row interchanges recorded (1-based, the
// LAPACK / torch.linalg.lu_factor convention: row p was swapped with row piv[p]; the swap runs over the whole row,
// so earlier multipliers move with it) -- A[:, :nr] is then the packed LU the reference hands back from pnqp
// (mpc/pnqp.py:52, 59, 82).
// sing != nullptr (the unconstrained solve, where the reference takes a pseudo-inverse, mpc/lqr_step.py:88-94): a pivot
// that is exactly zero -- a column with nothing left in it -- drops out: its unknown comes back as 0 and *sing is set.
// That is the pseudo-inverse when the null space is a coordinate axis (a control that enters neither cost nor dynamics);
// see pivot_inv in lqr_small_math.h.
template <typename real, bool KEEP_L = false> __device__ void lu_solve_aug(real *A, int nr, int ncols, real *Lcol, int *piv = nullptr,
                                                                           int *sing = nullptr)
{
    const int tid = threadIdx.x;
    if (tid < WAVE) {
        const int nt = WAVE;
        for (int p = 0; p < nr; ++p) {
            int r = p;
            real best = rabs(A[p * ncols + p]);
            for (int i = p + 1; i < nr; ++i) {
                real a = rabs(A[i * ncols + p]);
                if (a > best) { best = a; r = i; }
            }
            wave_sync();
            if (r != p)
                for (int j = tid; j < ncols; j += nt) {
                    real tmp = A[p * ncols + j];
                    A[p * ncols + j] = A[r * ncols + j];
                    A[r * ncols + j] = tmp;
                }
            wave_sync();
            const real d = A[p * ncols + p];
            const bool dead = sing && d == (real)0;
            if (dead && tid == 0) *sing = 1;
            for (int i = p + 1 + tid; i < nr; i += nt) {
                const real l = dead ? (real)0 : A[i * ncols + p] / d;
                Lcol[i] = l;
                if (KEEP_L) A[i * ncols + p] = l;
            }
            if (KEEP_L && piv && tid == 0) piv[p] = r + 1;
            wave_sync();
            for (int j = p + 1 + tid; j < ncols; j += nt) {
                const real pj = A[p * ncols + j];
                for (int i = p + 1; i < nr; ++i) A[i * ncols + j] -= Lcol[i] * pj;
            }
            wave_sync();
        }
        for (int j = nr + tid; j < ncols; j += nt) {
            for (int k = nr - 1; k >= 0; --k) {
                real x = A[k * ncols + j];
                for (int i = k + 1; i < nr; ++i) x -= A[k * ncols + i] * A[i * ncols + j];
                const real dk = A[k * ncols + k];
                A[k * ncols + j] = (sing && dk == (real)0) ? (real)0 : x / dk;
            }
        }
    }
    __syncthreads();
}

// Projected-Newton box QP for ONE problem (mpc/pnqp.py:5-82 with n_batch = 1):
//   min 0.5 x'Hx + q'x  s.t. lb <= x <= ub.
// H (n x n, leading dim ldH) and qv may live in LDS or global memory.  `rhs`
// (n x nrhs, leading dim ldR, may be NULL) is an extra right-hand side carried
// through every factorisation so that on exit A[:, n+1:] = H_free^{-1} rhs_free
// -- the K = -Quu_free^{-1} Qux of mpc/lqr_step.py:142-148 comes out of the same
// elimination that produced the final Newton step.  x must already hold the
// clamped start.  Returns the iteration index the reference returns.
template <typename real, bool KEEP_L = false>
__device__ int pnqp_core(const real *H, int ldH, const real *qv, const real *rhs, int ldR, int nrhs,
                         int n, int n_iter, real *A, real *Lcol, real *x, real *g, real *dx, real *mx,
                         const real *lb, const real *ub, int *If, bool *converged, int *piv = nullptr)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ncols = n + 1 + nrhs;
    const real GAMMA = (real)0.1;
    int it_ret = n_iter - 1;
    bool conv = false;
    // Is H symmetric (bit for bit)?  The two restatements of the Armijo test below -- a full Newton step passes without
    // evaluation, and f(x) - f(m) = -g'd - d'Hd/2 -- take g = Hx + q for the gradient of the objective, which it is for
    // a symmetric H only.  The reference evaluates obj(x) - obj(m) literally (mpc/pnqp.py:71-73), so a Quu that came out
    // of a non-symmetric C (mpc/lqr_step.py:68 uses C as given) gets the literal quantity here too, with the true
    // gradient (H + H')x/2 + q in place of g -- still without subtracting two large objective values.
    bool hsym = true;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) hsym = hsym & (H[i * ldH + j] == H[j * ldH + i]);
    for (int it = 0; it < n_iter; ++it) {
        // :29-33 gradient, clamped / free sets
        for (int i = tid; i < n; i += nt) {
            real r = 0;
            for (int j = 0; j < n; ++j) r += H[i * ldH + j] * x[j];
            r += qv[i];
            g[i] = r;
            const real xi = x[i];
            const bool ic = ((xi == lb[i]) && (r > 0)) || ((xi == ub[i]) && (r < 0));
            If[i] = ic ? 0 : 1;
        }
        __syncthreads();
        // :44-48 H_ = H on the free block (+1e-11 I), g_ = g on the free set
        for (int e = tid; e < n * ncols; e += nt) {
            const int i = e / ncols, j = e - i * ncols;
            const bool fi = If[i] != 0;
            real val;
            if (j < n) {
                val = (fi && If[j] != 0) ? H[i * ldH + j] : (real)0;
                if (i == j) val += (real)1e-11;
            } else if (j == n) {
                val = fi ? g[i] : (real)0;
            } else {
                val = fi ? rhs[i * ldR + (j - n - 1)] : (real)0;
            }
            A[e] = val;
        }
        __syncthreads();
        lu_solve_aug<real, KEEP_L>(A, n, ncols, Lcol, piv);            // :50-54
        for (int i = tid; i < n; i += nt) dx[i] = -A[i * ncols + n];
        __syncthreads();
        real nrm2 = 0;
        for (int i = 0; i < n; ++i) nrm2 += dx[i] * dx[i];
        if (!(rsqrt_<real>(nrm2) >= (real)1e-4)) {  // :56-59
            conv = true;
            it_ret = it;
            break;
        }
        // :61-76 Armijo backtracking (n_batch = 1 form of the batch-global loop).  A Newton step that
        // stays inside the box passes without evaluation: its Armijo ratio is exactly 1/2 for a quadratic
        // (see lqr_small_math.h); evaluating it in float32 near convergence only measures rounding noise.
        bool inside = true;
        for (int i = 0; i < n; ++i) {
            const real xn = x[i] + dx[i];
            inside = inside & ((xn >= lb[i]) & (xn <= ub[i]));
        }
        if (inside && hsym) {
            __syncthreads();
            for (int i = tid; i < n; i += nt) x[i] += dx[i];
            __syncthreads();
            continue;
        }
        // otherwise evaluate it, as f(x) - f(m) = -g'd - d'Hd/2 with d = m - x (no cancellation of two
        // large objective values)
        real alpha = 1;
        for (int count = 0; count < 10; ++count) {
            for (int i = tid; i < n; i += nt) mx[i] = eclamp<real>(x[i] + alpha * dx[i], lb[i], ub[i]);
            __syncthreads();
            real den = 0, dhd = 0, num = 0;
            for (int i = 0; i < n; ++i) {
                const real di = mx[i] - x[i];
                real r = 0, corr = 0;
                for (int j = 0; j < n; ++j) r += H[i * ldH + j] * (mx[j] - x[j]);
                if (!hsym)
                    for (int j = 0; j < n; ++j) corr += (H[j * ldH + i] - H[i * ldH + j]) * x[j];
                den -= g[i] * di;
                num -= (g[i] + (real)0.5 * corr) * di;
                dhd += di * r;
            }
            const real arm = (num - (real)0.5 * dhd) / den;
            __syncthreads();
            if (arm <= GAMMA) alpha *= (real)0.1; else break;
        }
        for (int i = tid; i < n; i += nt) x[i] = mx[i];   // :78
        __syncthreads();
    }
    *converged = conv;
    return it_ret;
}

// ---------------------------------------------------------------------------
// Riccati sweep for one problem.
// ---------------------------------------------------------------------------
template <typename real>
__device__ void sweep_problem(const StepParams<real> &p, int b, Smem<real> &s, real *Kdst, real *kdst,
                              real &old_cost, int &qp_total, int &status)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    const int ncols = nc + 1 + ns;
    real oc = 0;
    bool warm = false;
    qp_total = 0;
    for (int t = T - 1; t >= 0; --t) {
        const real *Ct = p.C + (long)t * p.C_st + (long)b * p.C_sb;
        const real *ct = p.c + (long)t * p.c_st + (long)b * p.c_sb;
        const long tb = (long)t * B + b;
        for (int e = tid; e < n * n; e += nt) s.Q[e] = Ct[e];
        if (t < T - 1) {
            const real *Ft = p.F + (long)t * p.F_st + (long)b * p.F_sb;
            for (int e = tid; e < ns * n; e += nt) s.F[e] = Ft[e];
        }
        for (int i = tid; i < n; i += nt)
            s.tau[i] = i < ns ? p.cur_x[tb * ns + i] : p.cur_u[tb * nc + (i - ns)];
        __syncthreads();
        // delta-space linear term c_back = C tau + c (mpc/lqr_step.py:289-295) and the
        // nominal cost 0.5 tau'C tau + c'tau (util.get_cost, :169) off the same product.
        for (int i = tid; i < n; i += nt) {
            real r = 0;
            for (int j = 0; j < n; ++j) r += s.Q[i * n + j] * s.tau[j];
            const real ci = ct[i], ti = s.tau[i];
            oc += (real)0.5 * ti * r + ci * ti;
            s.q[i] = r + ci;
        }
        if (t < T - 1) {
            // Q = C + F'VF, q = c_back + F'v  (:65-70; the f-term of :72-74 is dead: f_back=None)
            bool on_mfma = false;
            if constexpr (sizeof(real) == 4) on_mfma = nt > WAVE;
            if (on_mfma) {
                if constexpr (sizeof(real) == 4) {
                    const int wv_id = tid >> 6, nwv = nt >> 6;
                    const int ti_n = (n + 15) / 16, tk_n = (ns + 15) / 16;
                    for (int tile = wv_id; tile < ti_n * tk_n; tile += nwv) {       // W = F'V
                        const int r0 = 16 * (tile / tk_n), c0 = 16 * (tile % tk_n);
                        mfma_acc_t acc = {0.f, 0.f, 0.f, 0.f};
                        acc = tile_mma(r0, c0, ns,
                                       [&](int i, int m) { return (i < n && m < ns) ? s.F[m * n + i] : 0.f; },
                                       [&](int m, int k) { return (m < ns && k < ns) ? s.V[m * ns + k] : 0.f; }, acc);
                        tile_each(r0, c0, n, ns, [&](int i, int k, int v) { s.W[i * ns + k] = acc[v]; });
                    }
                    __syncthreads();
                    for (int tile = wv_id; tile < ti_n * ti_n; tile += nwv) {       // Q += W F
                        const int r0 = 16 * (tile / ti_n), c0 = 16 * (tile % ti_n);
                        mfma_acc_t acc = {0.f, 0.f, 0.f, 0.f};
                        tile_each(r0, c0, n, n, [&](int i, int j, int v) { acc[v] = s.Q[i * n + j]; });
                        acc = tile_mma(r0, c0, ns,
                                       [&](int i, int k) { return (i < n && k < ns) ? s.W[i * ns + k] : 0.f; },
                                       [&](int k, int j) { return (k < ns && j < n) ? s.F[k * n + j] : 0.f; }, acc);
                        tile_each(r0, c0, n, n, [&](int i, int j, int v) { s.Q[i * n + j] = acc[v]; });
                    }
                }
            } else {
                for (int e = tid; e < n * ns; e += nt) {
                    const int i = e / ns, k = e - i * ns;
                    real r = 0;
                    for (int m = 0; m < ns; ++m) r += s.F[m * n + i] * s.V[m * ns + k];
                    s.W[e] = r;
                }
                __syncthreads();
                for (int e = tid; e < n * n; e += nt) {
                    const int i = e / n, j = e - i * n;
                    real r = 0;
                    for (int k = 0; k < ns; ++k) r += s.W[i * ns + k] * s.F[k * n + j];
                    s.Q[e] += r;
                }
            }
            for (int i = tid; i < n; i += nt) {
                real r = 0;
                for (int m = 0; m < ns; ++m) r += s.F[m * n + i] * s.v[m];
                s.q[i] += r;
            }
        }
        __syncthreads();

        const real *Quu = s.Q + ns * n + ns;   // leading dim n
        const real *Qux = s.Q + ns * n;        // leading dim n
        const real *qu = s.q + ns;
        if (p.bound_mode == MPC_BOUND_NONE) {
            // :84-94 unconstrained, :99-127 masked (u_zero_I): [Quu_ | qu_ | Qux_] eliminated at once
            const uint8_t *mk = p.zero_mask ? p.zero_mask + tb * nc : nullptr;
            for (int e = tid; e < nc * ncols; e += nt) {
                const int i = e / ncols, j = e - i * ncols;
                const bool mi = mk && mk[i];
                real val;
                if (j < nc) {
                    const bool mj = mk && mk[j];
                    val = (!mi && !mj) ? Quu[i * n + j] : (real)0;
                    if (mi && i == j) val += (real)1e-8;     // :116
                } else if (j == nc) {
                    val = mi ? (real)0 : qu[i];
                } else {
                    val = mi ? (real)0 : Qux[i * n + (j - nc - 1)];
                }
                s.A[e] = val;
            }
            __syncthreads();
            // (pinverse semantics for the plain unconstrained solve with more than one control; the masked solve and a
            //  single control are an LU / a division in the reference)
            __shared__ int sing_flag;
            const bool pinv = !mk && nc > 1;
            if (pinv && tid == 0) sing_flag = 0;
            lu_solve_aug(s.A, nc, ncols, s.Lcol, (int *)nullptr, pinv ? &sing_flag : (int *)nullptr);
            if (pinv && sing_flag) status |= MPC_ST_QUU_SINGULAR;
        } else {
            // :128-148 box constraints in delta space
            for (int i = tid; i < nc; i += nt) {
                const real u = p.cur_u[tb * nc + i];
                real l = (p.bound_mode == MPC_BOUND_SCALAR ? p.lo_s : p.lo[tb * nc + i]) - u;
                real h = (p.bound_mode == MPC_BOUND_SCALAR ? p.hi_s : p.hi[tb * nc + i]) - u;
                if (p.has_delta) {                            // :132-134
                    if (l < -p.delta_u) l = -p.delta_u;
                    if (h > p.delta_u) h = p.delta_u;
                }
                s.lb[i] = l;
                s.ub[i] = h;
            }
            if (!warm) {
                // cold start x = -H^{-1} q (mpc/pnqp.py:14-19)
                for (int e = tid; e < nc * (nc + 1); e += nt) {
                    const int i = e / (nc + 1), j = e - i * (nc + 1);
                    s.A[e] = j < nc ? Quu[i * n + j] : qu[i];
                }
                __syncthreads();
                lu_solve_aug(s.A, nc, nc + 1, s.Lcol);
                for (int i = tid; i < nc; i += nt) s.px[i] = -s.A[i * (nc + 1) + nc];
            } else {
                for (int i = tid; i < nc; i += nt) s.px[i] = s.kt[i];   // warm start = k_{t+1} (:137,141)
            }
            __syncthreads();
            for (int i = tid; i < nc; i += nt) s.px[i] = eclamp<real>(s.px[i], s.lb[i], s.ub[i]);
            __syncthreads();
            bool conv = false;
            const int it = pnqp_core<real>(Quu, n, qu, Qux, n, ns, nc, p.pnqp_iter, s.A, s.Lcol, s.px, s.pg,
                                           s.pdx, s.pmx, s.lb, s.ub, s.If, &conv);
            qp_total += 1 + it;                               // :140
            if (!conv) status |= MPC_ST_PNQP_UNCONVERGED;
            warm = true;
        }
        for (int e = tid; e < nc * ns; e += nt) {
            const int i = e / ns, j = e - i * ns;
            s.Kt[e] = -s.A[i * ncols + nc + 1 + j];
        }
        for (int i = tid; i < nc; i += nt)
            s.kt[i] = (p.bound_mode == MPC_BOUND_NONE) ? -s.A[i * ncols + nc] : s.px[i];
        __syncthreads();
        {
            real *Kg = Kdst + tb * nc * ns, *kg = kdst + tb * nc;
            for (int e = tid; e < nc * ns; e += nt) Kg[e] = s.Kt[e];
            for (int i = tid; i < nc; i += nt) kg[i] = s.kt[i];
        }
        // :155-158 V = Qxx + Qxu K + K'Qux + K'Quu K ; v likewise (unmasked Quu, qu)
        for (int e = tid; e < nc * (ns + 1); e += nt) {
            const int l = e / (ns + 1), j = e - l * (ns + 1);
            real r = 0;
            for (int l2 = 0; l2 < nc; ++l2)
                r += Quu[l * n + l2] * (j < ns ? s.Kt[l2 * ns + j] : s.kt[l2]);
            s.M[e] = r;
        }
        __syncthreads();
        bool v_on_mfma = false;
        if constexpr (sizeof(real) == 4) v_on_mfma = nt > WAVE;
        if (v_on_mfma) {
            if constexpr (sizeof(real) == 4) {
                // V = Qxx + Qxu K + K'(Qux + Quu K): two products with inner dimension n_ctrl
                const int wv_id = tid >> 6, nwv = nt >> 6, tk_n = (ns + 15) / 16;
                for (int tile = wv_id; tile < tk_n * tk_n; tile += nwv) {
                    const int r0 = 16 * (tile / tk_n), c0 = 16 * (tile % tk_n);
                    mfma_acc_t acc = {0.f, 0.f, 0.f, 0.f};
                    tile_each(r0, c0, ns, ns, [&](int i, int j, int v) { acc[v] = s.Q[i * n + j]; });
                    acc = tile_mma(r0, c0, nc,
                                   [&](int i, int l) { return (i < ns && l < nc) ? s.Q[i * n + ns + l] : 0.f; },
                                   [&](int l, int j) { return (l < nc && j < ns) ? s.Kt[l * ns + j] : 0.f; }, acc);
                    acc = tile_mma(r0, c0, nc,
                                   [&](int i, int l) { return (i < ns && l < nc) ? s.Kt[l * ns + i] : 0.f; },
                                   [&](int l, int j) {
                                       return (l < nc && j < ns) ? Qux[l * n + j] + s.M[l * (ns + 1) + j] : 0.f;
                                   }, acc);
                    tile_each(r0, c0, ns, ns, [&](int i, int j, int v) { s.V[i * ns + j] = acc[v]; });
                }
            }
        } else {
            for (int e = tid; e < ns * ns; e += nt) {
                const int i = e / ns, j = e - i * ns;
                real t1 = 0, t2 = 0, t3 = 0;
                for (int l = 0; l < nc; ++l) {
                    const real kli = s.Kt[l * ns + i];
                    t1 += s.Q[i * n + ns + l] * s.Kt[l * ns + j];
                    t2 += kli * Qux[l * n + j];
                    t3 += kli * s.M[l * (ns + 1) + j];
                }
                s.V[e] = s.Q[i * n + j] + t1 + t2 + t3;
            }
        }
        for (int i = tid; i < ns; i += nt) {
            real t1 = 0, t2 = 0, t3 = 0;
            for (int l = 0; l < nc; ++l) {
                const real kli = s.Kt[l * ns + i];
                t1 += s.Q[i * n + ns + l] * s.kt[l];
                t2 += kli * qu[l];
                t3 += kli * s.M[l * (ns + 1) + ns];
            }
            s.v[i] = s.q[i] + t1 + t2 + t3;
        }
        __syncthreads();
    }
    old_cost = block_sum<real>(oc, s.red);
}

// Nominal cost only (util.get_cost with x given, mpc/util.py:129-153).
template <typename real>
__device__ real nominal_cost(const StepParams<real> &p, int b, Smem<real> &s)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    real oc = 0;
    for (int t = 0; t < T; ++t) {
        const real *Ct = p.C + (long)t * p.C_st + (long)b * p.C_sb;
        const real *ct = p.c + (long)t * p.c_st + (long)b * p.c_sb;
        const long tb = (long)t * B + b;
        for (int e = tid; e < n * n; e += nt) s.Q[e] = Ct[e];
        for (int i = tid; i < n; i += nt)
            s.tau[i] = i < ns ? p.cur_x[tb * ns + i] : p.cur_u[tb * nc + (i - ns)];
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            real r = 0;
            for (int j = 0; j < n; ++j) r += s.Q[i * n + j] * s.tau[j];
            oc += (real)0.5 * s.tau[i] * r + ct[i] * s.tau[i];
        }
        __syncthreads();
    }
    return block_sum<real>(oc, s.red);
}

// ---------------------------------------------------------------------------
// Line-searched rollout for one problem (mpc/lqr_step.py:164-261, LinDx/QuadCost).
// ---------------------------------------------------------------------------
template <typename real>
__device__ void rollout_problem(const StepParams<real> &p, int b, Smem<real> &s, const real *Ksrc,
                                const real *ksrc, real old_cost, int &status)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    real alpha = 1, cost = 0, dun = 0, full = 0;
    for (int pass = 0; pass < p.max_ls; ++pass) {
        for (int i = tid; i < ns; i += nt) {
            const real xi = p.x_init[(long)b * ns + i];
            s.xn[i] = xi;
            s.dxv[i] = 0;
            p.new_x[(long)b * ns + i] = xi;
        }
        real ca = 0, da = 0;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const long tb = (long)t * B + b;
            const real *Ct = p.C + (long)t * p.C_st + (long)b * p.C_sb;
            const real *ct = p.c + (long)t * p.c_st + (long)b * p.c_sb;
            const real *Kg = Ksrc + tb * nc * ns, *kg = ksrc + tb * nc;
            for (int e = tid; e < n * n; e += nt) s.Q[e] = Ct[e];
            if (t < T - 1 && !p.env.kind) {
                const real *Ft = p.F + (long)t * p.F_st + (long)b * p.F_sb;
                for (int e = tid; e < ns * n; e += nt) s.F[e] = Ft[e];
            }
            for (int i = tid; i < nc; i += nt) {
                real r = 0;
                for (int j = 0; j < ns; ++j) r += Kg[i * ns + j] * s.dxv[j];
                const real u = p.cur_u[tb * nc + i];
                real un = r + u + alpha * kg[i];                       // :192
                if (p.zero_mask && p.zero_mask[tb * nc + i]) un = 0;   // :197-198
                if (p.bound_mode != MPC_BOUND_NONE) {                  // :200-213
                    real l = p.bound_mode == MPC_BOUND_SCALAR ? p.lo_s : p.lo[tb * nc + i];
                    real h = p.bound_mode == MPC_BOUND_SCALAR ? p.hi_s : p.hi[tb * nc + i];
                    if (p.has_delta) {
                        const real l2 = u - p.delta_u, h2 = u + p.delta_u;
                        l = (l2 < l) ? l : l2;
                        h = (h2 > h) ? h : h2;
                    }
                    un = eclamp<real>(un, l, h);
                }
                s.tau[ns + i] = un;
                p.new_u[tb * nc + i] = un;
                const real d = u - un;
                da += d * d;
            }
            for (int j = tid; j < ns; j += nt) s.tau[j] = s.xn[j];
            __syncthreads();
            for (int i = tid; i < n; i += nt) {                        // :230-232
                real r = 0;
                for (int j = 0; j < n; ++j) r += s.Q[i * n + j] * s.tau[j];
                ca += (real)0.5 * s.tau[i] * r + ct[i] * s.tau[i];
            }
            if (t < T - 1 && p.env.kind) {                             // :223-225, a shipped simulator
                if (tid == 0) env_step<real>(p.env, s.tau, s.tau[ns], s.xn2, nullptr);
            } else if (t < T - 1) {                                    // :216-222
                const real *ft = p.f ? p.f + (long)t * p.f_st + (long)b * p.f_sb : nullptr;
                for (int i = tid; i < ns; i += nt) {
                    real r = 0;
                    for (int j = 0; j < n; ++j) r += s.F[i * n + j] * s.tau[j];
                    if (ft) r += ft[i];
                    s.xn2[i] = r;
                }
            }
            __syncthreads();
            if (t < T - 1) {
                const long tb1 = (long)(t + 1) * B + b;
                for (int i = tid; i < ns; i += nt) {
                    const real r = s.xn2[i];
                    s.xn[i] = r;
                    s.dxv[i] = r - p.cur_x[tb1 * ns + i];
                    p.new_x[tb1 * ns + i] = r;
                }
            }
            __syncthreads();
        }
        cost = block_sum<real>(ca, s.red);
        dun = rsqrt_<real>(block_sum<real>(da, s.red));
        if (pass == 0) full = dun;                                     // :243-245
        // :176-179, 247, 252: keep shrinking while this problem's cost got worse
        if (cost > old_cost && pass + 1 < p.max_ls) alpha *= p.ls_decay; else break;
    }
    if (!(cost == cost) || rabs(cost) > (real)3e38) status |= MPC_ST_NONFINITE;
    if (tid == 0) {
        if (p.costs) p.costs[b] = cost;
        if (p.old_costs) p.old_costs[b] = old_cost;
        if (p.full_du_norm) p.full_du_norm[b] = full;
        if (p.alpha_du_norm) p.alpha_du_norm[b] = dun;
        if (p.alphas) p.alphas[b] = alpha;
    }
}

// phase_mask: 1 = sweep, 2 = rollout, 3 = both (K,k round-trip through p.K/p.k, L2-resident)
template <typename real>
__global__ void __launch_bounds__(MAX_THREADS, (sizeof(real) == 4 ? 6 : 2)) lqr_step_generic_kernel(StepParams<real> p, int phase_mask)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem<real> s;
    s.carve(smem_raw, p.ns, p.nc, blockDim.x);
    // p.gate (mpc_lqr_step, impl 0, after a fused kernel): a small grid walks the batch and solves only the problems the
    // fused kernel flagged as having a non-symmetric C -- usually none, and the launch is a few microseconds of reading flags
    if (p.gate) {
        // all of this block's flags in ONE round of loads (a dependent load per problem made the empty launch 6.5 us); the
        // block owns b = blockIdx.x + k gridDim.x for EVERY k with b < B (the loop below), so beyond blockDim x gridDim
        // problems (B > 65536 with the 64-thread block) the pre-check takes further rounds -- independent loads, one vote
        int flag = 0;
        for (long mine = (long)blockIdx.x + (long)threadIdx.x * (long)gridDim.x; mine < p.B; mine += (long)blockDim.x * (long)gridDim.x)
            flag |= p.gate[mine] & MPC_ST_C_ASYMMETRIC;
        if (!__syncthreads_or(flag)) return;
    }
    for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    if (p.gate && !(p.gate[b] & MPC_ST_C_ASYMMETRIC)) continue;
    __syncthreads();                      // the previous problem's last readers of the staging area
    int status = p.gate ? (int)(MPC_ST_C_ASYMMETRIC | MPC_ST_C_TESTED) : 0, qp_total = 0;      // (the fused kernel's verdict stays)
    real old_cost = 0;
    if (phase_mask & 1) {
        sweep_problem<real>(p, b, s, p.K, p.k, old_cost, qp_total, status);
        if (threadIdx.x == 0) {
            if (p.qp_iters) p.qp_iters[b] = qp_total;
            if (p.old_costs) p.old_costs[b] = old_cost;
        }
    } else if (p.old_costs_in) {
        old_cost = p.old_costs_in[b];
    } else {
        old_cost = nominal_cost<real>(p, b, s);
    }
    if (phase_mask & 2) {
        // K,k were written by this same wave: make them visible to its own loads.
        __threadfence_block();
        __syncthreads();
        rollout_problem<real>(p, b, s, p.K, p.k, old_cost, status);
    }
    if (threadIdx.x == 0 && p.status) {
        if (phase_mask & 1) p.status[b] = status; else p.status[b] |= status;
    }
    }
}

// ---------------------------------------------------------------------------
// standalone pnqp (mpc/pnqp.py:5-82), one workgroup per problem, H read from HBM/L2
// ---------------------------------------------------------------------------
template <typename real>
__global__ void pnqp_kernel(int B, int n, const real *H, const real *q, const real *lo, const real *hi,
                            const real *x0, int n_iter, real *x_out, uint8_t *If_out, int *iters,
                            int *status, real *Hfree, real *LU, int *pivots)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (b >= B) return;
    real *A = reinterpret_cast<real *>(smem_raw);     // n x (n+1)
    real *Lcol = A + (size_t)n * (n + 1);
    real *x = Lcol + n, *g = x + n, *dx = g + n, *mx = dx + n, *lb = mx + n, *ub = lb + n;
    int *If = reinterpret_cast<int *>(ub + n);
    int *piv = If + n;
    const real *Hb = H + (size_t)b * n * n, *qb = q + (size_t)b * n;
    for (int i = tid; i < n; i += nt) { lb[i] = lo[(size_t)b * n + i]; ub[i] = hi[(size_t)b * n + i]; }
    if (x0 == nullptr) {
        for (int e = tid; e < n * (n + 1); e += nt) {
            const int i = e / (n + 1), j = e - i * (n + 1);
            A[e] = j < n ? Hb[i * n + j] : qb[i];
        }
        __syncthreads();
        lu_solve_aug(A, n, n + 1, Lcol);
        for (int i = tid; i < n; i += nt) x[i] = -A[i * (n + 1) + n];
    } else {
        for (int i = tid; i < n; i += nt) x[i] = x0[(size_t)b * n + i];
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) x[i] = eclamp<real>(x[i], lb[i], ub[i]);
    __syncthreads();
    bool conv = false;
    const int it = pnqp_core<real, true>(Hb, n, qb, nullptr, 0, 0, n, n_iter, A, Lcol, x, g, dx, mx, lb, ub, If, &conv, piv);
    for (int i = tid; i < n; i += nt) {
        x_out[(size_t)b * n + i] = x[i];
        if (If_out) If_out[(size_t)b * n + i] = (uint8_t)If[i];
        if (pivots) pivots[(size_t)b * n + i] = piv[i];
    }
    // the factorisation of the last Newton system, packed as torch.linalg.lu_factor packs it (what the reference
    // returns as H_lu_, mpc/pnqp.py:52-59, 82): nobody has to factor H_ again
    if (LU)
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            LU[(size_t)b * n * n + e] = A[i * (n + 1) + j];
        }
    if (Hfree)
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            real val = (If[i] && If[j]) ? Hb[e] : (real)0;
            if (i == j) val += (real)1e-11;
            Hfree[(size_t)b * n * n + e] = val;
        }
    if (tid == 0) {
        if (iters) iters[b] = it;
        if (status) status[b] = conv ? 0 : MPC_ST_PNQP_UNCONVERGED;
    }
}

// ---------------------------------------------------------------------------
// util.get_traj + util.get_cost (mpc/util.py:102-153)
// ---------------------------------------------------------------------------
template <typename real>
__global__ void __launch_bounds__(MAX_THREADS) traj_cost_kernel(StepParams<real> p, real *x, real *cost)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem<real> s;
    s.carve(smem_raw, p.ns, p.nc, blockDim.x);
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    if (b >= B) return;
    for (int i = tid; i < ns; i += nt) {
        const real xi = p.x_init[(long)b * ns + i];
        s.xn[i] = xi;
        if (x) x[(long)b * ns + i] = xi;
    }
    real ca = 0;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const long tb = (long)t * B + b;
        if (cost) {
            const real *Ct = p.C + (long)t * p.C_st + (long)b * p.C_sb;
            for (int e = tid; e < n * n; e += nt) s.Q[e] = Ct[e];
        }
        if (t < T - 1 && !p.env.kind) {
            const real *Ft = p.F + (long)t * p.F_st + (long)b * p.F_sb;
            for (int e = tid; e < ns * n; e += nt) s.F[e] = Ft[e];
        }
        for (int i = tid; i < n; i += nt) s.tau[i] = i < ns ? s.xn[i] : p.cur_u[tb * nc + (i - ns)];
        __syncthreads();
        if (cost) {
            const real *ct = p.c + (long)t * p.c_st + (long)b * p.c_sb;
            for (int i = tid; i < n; i += nt) {
                real r = 0;
                for (int j = 0; j < n; ++j) r += s.Q[i * n + j] * s.tau[j];
                ca += (real)0.5 * s.tau[i] * r + ct[i] * s.tau[i];
            }
        }
        if (t < T - 1 && p.env.kind) {                                 // mpc/util.py:112-113
            if (tid == 0) env_step<real>(p.env, s.tau, s.tau[ns], s.xn2, nullptr);
        } else if (t < T - 1) {
            const real *ft = p.f ? p.f + (long)t * p.f_st + (long)b * p.f_sb : nullptr;
            for (int i = tid; i < ns; i += nt) {
                real r = 0;
                for (int j = 0; j < n; ++j) r += s.F[i * n + j] * s.tau[j];
                if (ft) r += ft[i];
                s.xn2[i] = r;
            }
        }
        __syncthreads();
        if (t < T - 1)
            for (int i = tid; i < ns; i += nt) {
                s.xn[i] = s.xn2[i];
                if (x) x[((long)(t + 1) * B + b) * ns + i] = s.xn2[i];
            }
        __syncthreads();
    }
    if (cost) {
        const real tot = block_sum<real>(ca, s.red);
        if (tid == 0) cost[b] = tot;
    }
}

// util.get_traj (LinDx) for n_state + n_ctrl <= 16: one problem per 16-lane group, lane i owns state i and
// reads row i of F_t (a group reads one contiguous block per step), tau is exchanged by row shuffles, the
// next step's row is in flight while this one is summed.  The generic kernel above spends a whole
// wavefront and two barriers per step on the same 192 multiply-adds.
template <typename real, bool VEC>
__global__ void __launch_bounds__(256) traj_rows16_kernel(StepParams<real> p, real *x)
{
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, i = threadIdx.x & 15;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    const int b = gid < B ? gid : B - 1;            // idle groups shadow the last problem (no divergent shuffles)
    const bool own = gid < B && i < ns;
    real xi = i < ns ? p.x_init[(long)b * ns + i] : (real)0;
    if (own) x[(long)b * ns + i] = xi;
    if (T < 2) return;
    // The recursion is a chain of one row-times-vector product per timestep: what it waits for is memory.  DEPTH rows of
    // F (and the matching u / f entry) are kept in flight in registers.  Every load of the loop is unconditional (indices
    // clamped, zeros selected afterwards) so that the compiler can COUNT them: with a branch per element it drained the
    // whole queue (`s_waitcnt vmcnt(0)`) once per trip, and the 154 MB of F at the headline shape took 76-90 us.
    constexpr int DEPTH = 4;
    real row[DEPTH][16], aux[DEPTH];
    const int ir = i < ns ? i : 0;
    const bool hasf = p.f != nullptr;
    const bool isu = i >= ns && i < n;
    // lane i < ns: f_t[i];  ns <= i < n: u_t[i - ns];  the rest (and f absent): any valid address, the value is dropped
    const real *aux0 = isu ? p.cur_u + (long)b * nc + (i - ns) : (hasf && i < ns ? p.f + (long)b * p.f_sb + i : p.x_init);
    const long aux_st = isu ? (long)B * nc : (hasf && i < ns ? p.f_st : 0);
    const bool auxv = isu || (hasf && i < ns);
    const real *F0 = p.F + (long)b * p.F_sb + (long)ir * n;
    auto load_stage = [&](int t, int slot) {
        t = t < T - 2 ? t : T - 2;
        const real *Ft = F0 + (long)t * p.F_st;
        if constexpr (VEC) {
            const float4 *F4 = reinterpret_cast<const float4 *>(Ft);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = F4[q];
                row[slot][4 * q] = v.x, row[slot][4 * q + 1] = v.y, row[slot][4 * q + 2] = v.z, row[slot][4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) row[slot][j] = Ft[j < n ? j : 0];
        }
        aux[slot] = aux0[(long)t * aux_st];
    };
    auto advance = [&](int t, int slot) {
        const real a = auxv ? aux[slot] : (real)0;
        const real tau = i < ns ? xi : a;
        real acc = i < ns ? a : (real)0;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += (VEC || j < n ? row[slot][j] : (real)0) * __shfl(tau, j, 16);
        xi = acc;
        if (own) x[((long)(t + 1) * B + b) * ns + i] = xi;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_stage(d, d);
    int t0 = 0;
    for (; t0 + DEPTH <= T - 1; t0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            advance(t0 + d, d);
            load_stage(t0 + d + DEPTH, d);                  // the slot just used takes the stage DEPTH steps on
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
        if (t0 + d < T - 1) advance(t0 + d, d);
}

// ---------------------------------------------------------------------------
// KKT backward, closed-form part (mpc/lqr_step.py:346-404)
// ---------------------------------------------------------------------------
template <typename real>
__global__ void __launch_bounds__(MAX_THREADS)
kkt_grads_kernel(StepParams<real> p, const real *dx, const real *du, const real *dl_dx, const real *dl_du,
                 real *dC, real *dc, real *dF, real *df, real *dx_init)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem<real> s;
    s.carve(smem_raw, p.ns, p.nc, blockDim.x);
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    if (b >= B) return;
    // costates live in s.xn (lam), s.dxv (dlam); next values in s.xn2 / s.v
    real *lam = s.xn, *dlam = s.dxv, *lam2 = s.xn2, *dlam2 = s.v;
    bool have = false;
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        const real *Ct = p.C + (long)t * p.C_st + (long)b * p.C_sb;
        const real *ct = p.c + (long)t * p.c_st + (long)b * p.c_sb;
        for (int e = tid; e < n * n; e += nt) s.Q[e] = Ct[e];
        if (have) {
            const real *Ft = p.F + (long)t * p.F_st + (long)b * p.F_sb;
            for (int e = tid; e < ns * n; e += nt) s.F[e] = Ft[e];
        }
        for (int i = tid; i < n; i += nt) {
            s.tau[i] = i < ns ? p.cur_x[tb * ns + i] : p.cur_u[tb * nc + (i - ns)];
            s.dtau[i] = i < ns ? dx[tb * ns + i] : du[tb * nc + (i - ns)];
        }
        __syncthreads();
        // :346-353  dC_t = -0.5 (dtau tau' + tau dtau'),  dc_t = -dtau
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            dC[tb * n * n + e] = (real)-0.5 * (s.dtau[i] * s.tau[j] + s.tau[i] * s.dtau[j]);
        }
        for (int i = tid; i < n; i += nt) dc[tb * n + i] = -s.dtau[i];
        // :355-385 costate recursions (rows 0..ns-1 of C; F_x = first ns columns of F)
        for (int i = tid; i < ns; i += nt) {
            real r1 = 0, r2 = 0;
            for (int j = 0; j < n; ++j) {
                r1 += s.Q[i * n + j] * s.tau[j];
                r2 += s.Q[i * n + j] * s.dtau[j];
            }
            r1 += ct[i];
            r2 -= dl_dx[tb * ns + i];
            if (have)
                for (int m = 0; m < ns; ++m) {
                    r1 += s.F[m * n + i] * lam[m];
                    r2 += s.F[m * n + i] * dlam[m];
                }
            lam2[i] = r1;
            dlam2[i] = r2;
        }
        // :387-400 dF_t = -(dlam_{t+1} tau_t' + lam_{t+1} dtau_t'),  df_t = -dlam_{t+1}
        if (have) {
            for (int e = tid; e < ns * n; e += nt) {
                const int i = e / n, j = e - i * n;
                dF[tb * ns * n + e] = -(dlam[i] * s.tau[j] + lam[i] * s.dtau[j]);
            }
            if (df)
                for (int i = tid; i < ns; i += nt) df[tb * ns + i] = -dlam[i];
        }
        __syncthreads();
        for (int i = tid; i < ns; i += nt) { lam[i] = lam2[i]; dlam[i] = dlam2[i]; }
        __syncthreads();
        have = true;
    }
    for (int i = tid; i < ns; i += nt) dx_init[(long)b * ns + i] = -dlam[i];   // :404
}

template <typename real>
__global__ void kkt_prepare_kernel(long TB, int ns, int nc, const real *dl_dx, const real *dl_du,
                                   const real *u_star, int bound_mode, real lo_s, real hi_s,
                                   const real *lo, const real *hi, real *negr, uint8_t *mask)
{
    const int n = ns + nc;
    const long total = TB * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long tb = e / n;
        const int i = (int)(e - tb * n);
        if (i < ns) {
            negr[e] = -dl_dx[tb * ns + i];                              // mpc/lqr_step.py:316-320, 339
        } else {
            const long ui = tb * nc + (i - ns);
            negr[e] = -dl_du[ui];
            if (mask) {                                                 // :322-326
                const real l = bound_mode == MPC_BOUND_SCALAR ? lo_s : lo[ui];
                const real h = bound_mode == MPC_BOUND_SCALAR ? hi_s : hi[ui];
                const real u = u_star[ui];
                mask[ui] = (rabs<real>(u - l) <= (real)1e-8 || rabs<real>(u - h) <= (real)1e-8) ? 1 : 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// best-iterate select + convergence reductions (mpc/mpc.py:271-285, 299)
// ---------------------------------------------------------------------------
template <typename real> struct BitsOf;
template <> struct BitsOf<float> { using type = unsigned int; };
template <> struct BitsOf<double> { using type = unsigned long long; };

// One launch (round 3; rounds 1-2: a copy kernel per array + an update kernel + the caller's device-to-host copy of the
// two result words: 22 us per iLQR iteration at the headline shape).
//   Workgroups 1..: each owns kSelPB consecutive problems -- their `take` bits come from costs and the OLD best costs,
// which it then overwrites; their rows of x_t and u_t are one contiguous piece per timestep.
//   Workgroup 0 computes the two batch-wide words on its own, reading all B problems' scalars: no atomics, no ticket, no
// fence (on this chip an agent-scope fence writes an XCD's L2 back; 512 workgroups drawing tickets took 30 us).  It
// races with the owners' updates of best_costs and that is harmless: an owner stores cost[b] there only where the
// problem was taken, and cost[b] <= cost[b] + eps, so the old and the new word give the same `take` (eps >= 0).
constexpr int kSelPB = 8;

template <typename real> struct SelectArgs {
    int B, T, ns, nc, first;
    real eps;
    const real *x, *u, *costs, *du;
    real *bx, *bu, *bc, *bd;
    const int *status;
    unsigned char *flags, *host;      // int32 result bits at byte 0, the maximum at byte 8; host: the same, page-locked
    int host_tag;                     // stored at byte 4 of `host` once the two results are visible there
};

template <typename real, typename vec>
__global__ void __launch_bounds__(256) select_best_kernel(SelectArgs<real> a, int owners_only)
{
    constexpr int VW = sizeof(vec) / sizeof(real);
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && !owners_only) {
        __shared__ real s_max[4];
        __shared__ int s_fl[4];
        const volatile real *bc = a.bc;
        real d = 0;
        int fl = 0;
        for (int b = tid; b < a.B; b += 256) {
            real v = a.du[b];
            if (a.first || a.costs[b] <= bc[b] + a.eps) fl |= 1;
            if (v != v) { fl |= 2; v = 0; }
            d = v > d ? v : d;
            // C is known to be symmetric only where a kernel that TESTS it has run and found nothing (include/mpc_lqr.h)
            if (a.status && (a.status[b] & (MPC_ST_C_ASYMMETRIC | MPC_ST_C_TESTED)) != MPC_ST_C_TESTED) fl |= 4;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const real o = __shfl_down(d, off);
            d = o > d ? o : d;
            fl |= __shfl_down(fl, off);
        }
        if ((tid & 63) == 0) s_max[tid >> 6] = d, s_fl[tid >> 6] = fl;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) d = s_max[w] > d ? s_max[w] : d, fl |= s_fl[w];
            if (fl & 2) d = (real)NAN;                  // a NaN norm anywhere: the maximum is NaN, as torch.max reports it
            const int word = (((fl & 1) && !a.first) ? 1 : 0) | ((fl & 4) ? 2 : 0);
            *reinterpret_cast<int *>(a.flags) = word;
            *reinterpret_cast<real *>(a.flags + 8) = d;
            if (a.host) {
                *reinterpret_cast<volatile int *>(a.host) = word;
                *reinterpret_cast<volatile real *>(a.host + 8) = d;
                __threadfence_system();               // the results first, then the tag a polling host waits for
                *reinterpret_cast<volatile int *>(a.host + 4) = a.host_tag;
                __threadfence_system();
            }
        }
        return;
    }
    __shared__ int s_take[kSelPB];
    const int b0 = (blockIdx.x - (owners_only ? 0 : 1)) * kSelPB;
    const int nb = a.B - b0 < kSelPB ? a.B - b0 : kSelPB;
    if (tid < nb) {
        const int b = b0 + tid;
        const real cnew = a.costs[b];
        const int take = a.first || cnew <= a.bc[b] + a.eps;
        if (take) {
            a.bc[b] = cnew;
            a.bd[b] = a.du[b];
        }
        s_take[tid] = take;
    }
    __syncthreads();
    const int wx = nb * a.ns / VW, wu = nb * a.nc / VW, W = wx + wu;
    const vec *__restrict__ x = reinterpret_cast<const vec *>(a.x), *__restrict__ u = reinterpret_cast<const vec *>(a.u);
    vec *__restrict__ bx = reinterpret_cast<vec *>(a.bx), *__restrict__ bu = reinterpret_cast<vec *>(a.bu);
    const int total = a.T * W;
    for (int it = tid; it < total; it += 256) {
        const int t = it / W, r = it - t * W;
        const bool isx = r < wx;
        const int e = isx ? r : r - wx, d = isx ? a.ns : a.nc;
        if (!s_take[e * VW / d]) continue;
        const long at = ((long)t * a.B + b0) * d / VW + e;
        if (isx) bx[at] = x[at];
        else bu[at] = u[at];
    }
}

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string(what) + ": " + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

}  // namespace

size_t generic_lds_bytes(int ns, int nc, size_t elem)
{
    const size_t n = (size_t)ns + nc;
    size_t cnt = n * n + ns * n + n * ns + (size_t)ns * ns + (size_t)nc * (nc + 1 + ns) + (size_t)nc * ns +
                 (size_t)nc * (ns + 1) + n + ns + n + n + (size_t)nc * 8 + (size_t)ns * 3 + MAX_THREADS;
    return cnt * elem + (size_t)nc * sizeof(int) + 16;
}

template <typename real> int launch_step_generic(const StepParams<real> &p, int phase_mask, hipStream_t st)
{
    const size_t lds = generic_lds_bytes(p.ns, p.nc, sizeof(real));
    if (lds > 160 * 1024) { set_last_error("n_state/n_ctrl too large for the LDS-resident generic kernel"); return MPC_E_DIMS; }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lqr_step_generic_kernel<real>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = p.gate ? (p.B < 1024 ? p.B : 1024) : p.B;
    hipLaunchKernelGGL(lqr_step_generic_kernel<real>, dim3(grid), dim3(threads_for(p.ns, p.nc)), lds, st, p, phase_mask);
    return check_launch("lqr_step_generic_kernel");
}

template <typename real>
int launch_pnqp(int B, int n, const real *H, const real *q, const real *lo, const real *hi, const real *x0,
                int n_iter, real *x, uint8_t *If_out, int *iters, int *status, real *Hfree, real *LU, int *pivots,
                hipStream_t st)
{
    const size_t lds = ((size_t)n * (n + 1) + 7 * (size_t)n) * sizeof(real) + 2 * (size_t)n * sizeof(int) + 16;
    if (lds > 160 * 1024) { set_last_error("pnqp: n too large for LDS"); return MPC_E_DIMS; }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pnqp_kernel<real>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int threads = n <= 32 ? 64 : (n <= 64 ? 128 : 256);
    hipLaunchKernelGGL(pnqp_kernel<real>, dim3(B), dim3(threads), lds, st, B, n, H, q, lo, hi, x0, n_iter, x,
                       If_out, iters, status, Hfree, LU, pivots);
    return check_launch("pnqp_kernel");
}

// util.get_traj through a shipped simulator (mpc/util.py:102-126 with the module of mpc/env_dx): a lane per problem -- the
// transition is ~100 instructions of one lane's arithmetic, and the generic kernel above spends a workgroup and two barriers
// per timestep on it (19 / 33 us at the pendulum / cart-pole sizes against 16 / 24 here: T dependent simulator calls).
template <typename real>
__global__ void __launch_bounds__(64) env_traj_lane_kernel(StepParams<real> p, real *x)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= p.B) return;
    const int ns = p.ns, T = p.T, B = p.B;
    real xi[5], xn[5];
    for (int i = 0; i < ns; ++i) {
        xi[i] = p.x_init[(long)b * ns + i];
        x[(long)b * ns + i] = xi[i];
    }
    real u = T > 1 ? p.cur_u[b] : (real)0;
    for (int t = 0; t < T - 1; ++t) {
        const real un = t + 1 < T - 1 ? p.cur_u[(long)(t + 1) * B + b] : (real)0;     // (next step's control in flight)
        env_step<real>(p.env, xi, u, xn, nullptr);
        for (int i = 0; i < ns; ++i) {
            xi[i] = xn[i];
            x[((long)(t + 1) * B + b) * ns + i] = xn[i];
        }
        u = un;
    }
}

template <typename real> int launch_traj_cost(const StepParams<real> &p, real *x, real *cost, hipStream_t st)
{
    if (!cost && x && p.env.kind && p.nc == 1 && p.ns <= 5) {
        hipLaunchKernelGGL(env_traj_lane_kernel<real>, dim3((unsigned)((p.B + 63) / 64)), dim3(64), 0, st, p, x);
        return check_launch("env_traj_lane_kernel");
    }
    if (!cost && x && !p.env.kind && p.ns + p.nc <= 16) {
        const long groups = p.B;
        const dim3 grid((unsigned)((groups * 16 + 255) / 256));
        bool vec = false;
        if constexpr (sizeof(real) == 4)
            vec = p.ns + p.nc == 16 && ((uintptr_t)p.F & 15) == 0 && p.F_st % 4 == 0 && p.F_sb % 4 == 0;
        if (vec) hipLaunchKernelGGL((traj_rows16_kernel<real, sizeof(real) == 4>), grid, dim3(256), 0, st, p, x);
        else hipLaunchKernelGGL((traj_rows16_kernel<real, false>), grid, dim3(256), 0, st, p, x);
        return check_launch("traj_rows16_kernel");
    }
    const size_t lds = generic_lds_bytes(p.ns, p.nc, sizeof(real));
    if (lds > 160 * 1024) { set_last_error("traj_cost: dims too large"); return MPC_E_DIMS; }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&traj_cost_kernel<real>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(traj_cost_kernel<real>, dim3(p.B), dim3(threads_for(p.ns, p.nc)), lds, st, p, x, cost);
    return check_launch("traj_cost_kernel");
}

template <typename real>
int launch_kkt_grads(const StepParams<real> &p, const real *dx, const real *du, const real *dl_dx,
                     const real *dl_du, real *dC, real *dc, real *dF, real *df, real *dx_init, hipStream_t st)
{
    const size_t lds = generic_lds_bytes(p.ns, p.nc, sizeof(real));
    if (lds > 160 * 1024) { set_last_error("kkt_grads: dims too large"); return MPC_E_DIMS; }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kkt_grads_kernel<real>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kkt_grads_kernel<real>, dim3(p.B), dim3(threads_for(p.ns, p.nc)), lds, st, p, dx, du, dl_dx, dl_du, dC, dc,
                       dF, df, dx_init);
    return check_launch("kkt_grads_kernel");
}

template <typename real>
int launch_kkt_prepare(int B, int T, int ns, int nc, const real *dl_dx, const real *dl_du, const real *u_star,
                       int bound_mode, real lo_s, real hi_s, const real *lo, const real *hi, real *negr,
                       uint8_t *mask, hipStream_t st)
{
    const long TB = (long)T * B;
    const long total = TB * (ns + nc);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kkt_prepare_kernel<real>, dim3(blocks), dim3(256), 0, st, TB, ns, nc, dl_dx, dl_du, u_star,
                       bound_mode, lo_s, hi_s, lo, hi, negr, mask);
    return check_launch("kkt_prepare_kernel");
}

template <typename real>
int launch_select_best(int B, int T, int ns, int nc, int first, real eps, const real *x, const real *u,
                       const real *costs, const real *du_norm, real *bx, real *bu, real *bc, real *bd,
                       void *flags, void *host_flags, int host_tag, const int *status, hipStream_t st)
{
    SelectArgs<real> a{B, T, ns, nc, first, eps, x, u, costs, du_norm, bx, bu, bc, bd, status,
                       (unsigned char *)flags, (unsigned char *)host_flags, host_tag};
    const unsigned owners = (unsigned)((B + kSelPB - 1) / kSelPB);
    bool vec4 = false;
    if constexpr (sizeof(real) == 4)
        vec4 = ns % 4 == 0 && nc % 4 == 0 && (((uintptr_t)x | (uintptr_t)bx | (uintptr_t)u | (uintptr_t)bu) & 15) == 0;
    auto go = [&](unsigned grid, int owners_only) {
        if (vec4) hipLaunchKernelGGL((select_best_kernel<real, float4>), dim3(grid), dim3(256), 0, st, a, owners_only);
        else hipLaunchKernelGGL((select_best_kernel<real, real>), dim3(grid), dim3(256), 0, st, a, owners_only);
    };
    if (eps >= 0) go(1 + owners, 0);
    else {                       // (a negative tolerance voids the argument that lets workgroup 0 run beside the owners)
        go(1, 0);
        go(owners, 1);
    }
    return check_launch("select_best_kernel");
}

// ---------------------------------------------------------------------------
// The reference's full_du_norm for n_batch > 1, mpc/lqr_step.py:243-245:
//     (u - new_u).transpose(1, 2).contiguous().view(n_batch, -1).norm(2, 1)
// The transpose in front of the reshape makes "row r" the T n_ctrl consecutive elements [r T nc, (r+1) T nc) of the [T, nc, B]
// array -- one or two (t, a) pairs across ALL problems, not problem r's controls (for n_batch = 1 the two are the same).  The
// product's own norm is each problem's; this kernel serves `reference_du_norm=True`, which reproduces the reference's
// eps exit and detach mask (mpc/mpc.py:299, 321-334).  One workgroup per row; element f of the transposed array is
// (t, a, b) = (f / (nc B), f / B % nc, f % B).
// ---------------------------------------------------------------------------
template <typename real>
__global__ void __launch_bounds__(256) du_norm_reference_kernel(int T, int B, int nc, const real *u, const real *new_u, real *out)
{
    const long row = (long)T * nc;
    const long f0 = (long)blockIdx.x * row;
    real acc = 0;
    for (long i = threadIdx.x; i < row; i += blockDim.x) {
        const long f = f0 + i;
        const long b = f % B, ta = f / B;          // ta = t nc + a
        const long src = (ta / nc * B + b) * nc + ta % nc;
        const real d = u[src] - new_u[src];
        acc += d * d;
    }
    __shared__ real part[256];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = rsqrt_<real>(part[0]);
}

template <typename real>
int launch_du_norm_reference(int T, int B, int nc, const real *u, const real *new_u, real *out, hipStream_t st)
{
    hipLaunchKernelGGL(du_norm_reference_kernel<real>, dim3(B), dim3(256), 0, st, T, B, nc, u, new_u, out);
    return check_launch("du_norm_reference_kernel");
}

// ---------------------------------------------------------------------------
// MPC.linearize_dynamics for a shipped simulator (mpc/mpc.py:490-549): one thread per
// trajectory point, closed-form Jacobian, f = env(x,u) - F [x;u].
// ---------------------------------------------------------------------------
template <typename real>
__global__ void env_linearize_kernel(EnvDesc<real> env, long N, const real *x, const real *u, real *F, real *f)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int ns = env_ns(env.kind), n = ns + 1;
    real xi[5], out[5], J[30];
    for (int j = 0; j < ns; ++j) xi[j] = x[i * ns + j];
    const real ui = u[i];
    env_step<real>(env, xi, ui, out, J);
    for (int r = 0; r < ns; ++r) {
        real acc = out[r];
        for (int j = 0; j < ns; ++j) acc -= J[r * n + j] * xi[j];
        acc -= J[r * n + ns] * ui;
        f[i * ns + r] = acc;
        for (int j = 0; j < n; ++j) F[(i * ns + r) * n + j] = J[r * n + j];
    }
}

template <typename real>
int launch_env_linearize(const EnvDesc<real> &env, long N, const real *x, const real *u, real *F, real *f,
                         hipStream_t st)
{
    if (N <= 0) return MPC_OK;
    hipLaunchKernelGGL(env_linearize_kernel<real>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, env, N, x,
                       u, F, f);
    return check_launch("env_linearize_kernel");
}

#define INSTANTIATE(real)                                                                                     \
    template int launch_step_generic<real>(const StepParams<real> &, int, hipStream_t);                        \
    template int launch_pnqp<real>(int, int, const real *, const real *, const real *, const real *,          \
                                   const real *, int, real *, uint8_t *, int *, int *, real *, real *, int *, \
                                   hipStream_t);                                                              \
    template int launch_traj_cost<real>(const StepParams<real> &, real *, real *, hipStream_t);                \
    template int launch_kkt_grads<real>(const StepParams<real> &, const real *, const real *, const real *,   \
                                        const real *, real *, real *, real *, real *, real *, hipStream_t);   \
    template int launch_kkt_prepare<real>(int, int, int, int, const real *, const real *, const real *, int,   \
                                          real, real, const real *, const real *, real *, uint8_t *,          \
                                          hipStream_t);                                                       \
    template int launch_env_linearize<real>(const EnvDesc<real> &, long, const real *, const real *, real *,   \
                                            real *, hipStream_t);                                             \
    template int launch_du_norm_reference<real>(int, int, int, const real *, const real *, real *, hipStream_t); \
    template int launch_select_best<real>(int, int, int, int, int, real, const real *, const real *,          \
                                          const real *, const real *, real *, real *, real *, real *, void *, \
                                          void *, int, const int *, hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)

}  // namespace mpclqr
