// nn_dynamics.hip -- mpc.dynamics.NNDynamics (mpc/dynamics.py:15-128) inside the kernels, fp32.
//
// The reference calls the network from Python once per timestep per line-search pass (mpc/lqr_step.py:223-225,
// mpc/util.py:112-113) and builds its Jacobian with one [N, out, in] tensor per layer (mpc/dynamics.py:100-116).
// A fully connected network over a batch that shares its weights is a GEMM, so here a wavefront takes SIXTEEN
// problems (trajectory points) and runs every layer on v_mfma_f32_16x16x4_f32:
//
//   activations live as "features x problems": D[i][j] = z[feature 16 tile + i][problem j].  In the accumulator
//   layout lane (q, r) holds rows 4q..4q+3 of column r -- and with the contraction index of K-block v' taken as
//   feature 4q + v' that same register quadruple IS the B operand of the next layer (the contraction order of an
//   MFMA is free).  The A operand is the weight matrix as nn.Linear stores it: lane (q, r) reads
//   W[16 tile_out + r][16 tile_in + 4q .. +3], one 16-byte load for four MFMAs.  Nothing moves across lanes between
//   layers; activations go through LDS only to be indexable by a runtime tile number (registers are not).
//
//   nn_rollout_kernel    lqr_forward with the network as true_dynamics (mpc/lqr_step.py:164-261): per-problem feedback
//                        u' = K dx + u + alpha k, bounds, the quadratic cost and the per-problem line search around
//                        the shared-weight network; without gains it is util.get_traj / get_cost (mpc/util.py:102-153)
//   nn_linearize_kernel  MPC.linearize_dynamics(ANALYTIC) (mpc/mpc.py:495-512): F = d net / d [x;u] by the chain
//                        W_L diag(s_{L-1}) W_{L-1} ... diag(s_1) W_1 per point, every product after the first an MFMA
//                        whose B operand is the previous product's accumulator; f = net(x, u) - F [x;u]
//
// Weights are re-packed once per call into zero-padded [out_pad][in_pad] blocks (multiples of 16) in the workspace.
#include <string>

#include "lqr_common.h"

namespace mpclqr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// The packed network: one block of floats, layer after layer, W_l as [wp[l+1]][wp[l] + 4] (rows padded by 16 bytes so
// that the sixteen rows a 16-byte-per-lane read touches do not fall on the same LDS banks) followed by b_l [wp[l+1]].
// The block is copied into LDS at kernel start when it fits (WL), else read from global memory through the caches.
struct MlpDesc {
    int L, act, pass;            // Linear layers, MPC_ACT_*, passthrough
    int carry;                   // mpc_mlp_dynamics.ctrl_carry
    int w[MPC_MLP_MAX_LAYERS + 1];   // widths: w[0] = n_state + n_ctrl, w[L] = n_state
    int wp[MPC_MLP_MAX_LAYERS + 1];  // rounded up to 16
    int woff[MPC_MLP_MAX_LAYERS], boff[MPC_MLP_MAX_LAYERS];   // offsets (floats) into the packed block
    int total;                   // floats in the packed block (a multiple of 4)
    const float *packed;         // in the workspace
};

struct PackArgs {
    MlpDesc d;
    const float *W[MPC_MLP_MAX_LAYERS], *b[MPC_MLP_MAX_LAYERS];
    float *dst;
};

__global__ void mlp_pack_kernel(PackArgs a)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int l = 0; l < a.d.L; ++l) {
        const int in = a.d.w[l], out = a.d.w[l + 1], ldp = a.d.wp[l] + 4, outp = a.d.wp[l + 1];
        for (int e = tid; e < outp * ldp; e += nt) {
            const int o = e / ldp, i = e - o * ldp;
            a.dst[a.d.woff[l] + e] = (o < out && i < in) ? a.W[l][o * in + i] : 0.f;
        }
        for (int o = tid; o < outp; o += nt) a.dst[a.d.boff[l] + o] = o < out ? a.b[l][o] : 0.f;
    }
}

// wave-local ordering of LDS traffic: a wavefront's LDS instructions execute in order, only the compiler has to be
// kept from moving them (the per-wave staging areas are never touched by another wave)
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); }

// the activation on one accumulator (one wave-uniform branch per tile, not per element)
__device__ __forceinline__ f32x4 act_fn(f32x4 a, int kind)
{
    if (kind == MPC_ACT_SIGMOID) {
#pragma unroll
        for (int v = 0; v < 4; ++v) a[v] = __builtin_amdgcn_rcpf(1.f + __expf(-a[v]));   // v_exp_f32 + v_rcp_f32 (1 ulp each), no division sequence
    } else if (kind == MPC_ACT_RELU) {
#pragma unroll
        for (int v = 0; v < 4; ++v) a[v] = fmaxf(a[v], 0.f);
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) a[v] = a[v] > 0.f ? a[v] : expm1f(a[v]);   // F.elu, alpha = 1
    }
    return a;
}
// derivative of the activation from its OUTPUT (mpc/dynamics.py:104-112)
__device__ __forceinline__ f32x4 slope_fn(f32x4 z, int kind)
{
    if (kind == MPC_ACT_SIGMOID) {
#pragma unroll
        for (int v = 0; v < 4; ++v) z[v] = z[v] * (1.f - z[v]);
    } else if (kind == MPC_ACT_RELU) {
#pragma unroll
        for (int v = 0; v < 4; ++v) z[v] = z[v] > 0.f ? 1.f : 0.f;
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) z[v] = z[v] > 0.f ? 1.f : z[v] + 1.f;
    }
    return z;
}

// All layers for the sixteen columns staged in tauS (row r = column's [x;u], zero padded to wp[0]).  Hidden
// activations go to zbase: layer l at zbase + l * 16 * ZS when KEEP (the Jacobian needs them all), else ping-pong.
// Returns the output layer's accumulator: features 4q..4q+3 of column r (n_state <= 16: one tile).
// wts: the packed network (LDS or global).  The operand loads of the next K-step are issued before the current
// one's MFMAs: with one wavefront per SIMD nothing else hides their latency.
template <bool KEEP>
__device__ __forceinline__ f32x4 mlp_forward(const MlpDesc &m, const float *wts, const float *tauS, int TS, float *zbase,
                                             int ZS, int q, int r)
{
    const float *in = tauS;
    int is = TS;
    f32x4 res = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < m.L; ++l) {
        const int nin_t = m.wp[l] >> 4, nout_t = m.wp[l + 1] >> 4, ldp = m.wp[l] + 4;
        const float *W = wts + m.woff[l], *bias = wts + m.boff[l];
        float *dst = zbase + (KEEP ? l : (l & 1)) * 16 * ZS;
        const bool last = l + 1 == m.L;
        const float *irow = in + r * is + 4 * q;
        for (int to = 0; to < nout_t; ++to) {
            // four accumulation chains (one per K-block of a 16-byte load), joined at the end
            f32x4 a0 = *reinterpret_cast<const f32x4 *>(bias + 16 * to + 4 * q);
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
            const float *wrow = W + (16 * to + r) * ldp + 4 * q;
            f32x4 a = *reinterpret_cast<const f32x4 *>(wrow);
            f32x4 b = *reinterpret_cast<const f32x4 *>(irow);
            for (int ti = 0; ti < nin_t; ++ti) {
                const int nx = ti + 1 < nin_t ? ti + 1 : ti;
                const f32x4 an = *reinterpret_cast<const f32x4 *>(wrow + 16 * nx);
                const f32x4 bn = *reinterpret_cast<const f32x4 *>(irow + 16 * nx);
                a0 = mfma(a[0], b[0], a0);
                a1 = mfma(a[1], b[1], a1);
                a2 = mfma(a[2], b[2], a2);
                a3 = mfma(a[3], b[3], a3);
                a = an;
                b = bn;
            }
            f32x4 acc = (a0 + a1) + (a2 + a3);
            if (!last) {
                *reinterpret_cast<f32x4 *>(dst + r * ZS + 16 * to + 4 * q) = act_fn(acc, m.act);
            } else if (to == 0) {
                res = acc;
            }
        }
        wave_sync();
        in = dst;
        is = ZS;
    }
    return res;
}

// the packed network into LDS (all waves of the workgroup), or a pointer to it in global memory
template <bool WL> __device__ __forceinline__ const float *stage_weights(const MlpDesc &m, float *wl)
{
    if (!WL) return m.packed;
    for (int e = threadIdx.x * 4; e < m.total; e += blockDim.x * 4)
        *reinterpret_cast<f32x4 *>(wl + e) = *reinterpret_cast<const f32x4 *>(m.packed + e);
    __syncthreads();
    return wl;
}

// One row's share of J_t(tau') - J_t(tau_nominal) for the stage cost 0.5 tau'C tau + c'tau (mpc/lqr_step.py:230-232), any C:
//     0.5 tau''C tau' - 0.5 tb'C tb = 0.5 [ d'(C tau') + tb'(C d) ],   d = tau' - tb,
// from the row products s = (C tau')_i, s2 = (C d)_i, this row's tau'_i, d_i and c_i.  Round 5: the line search's test
// "did the cost get worse" (:176-179) was the comparison of two float32 sums of ~1e4 whose difference, from the third iLQR
// iteration on, is below their rounding (1e-2): a coin per trial and problem, and the unluckiest of the 16 problems of the
// unluckiest wavefront ran all ten trials -- 1.3 ms a rollout instead of 0.13 (profiles/r05_trace_nn_before.txt).  Every
// term here is proportional to d; the cost handed back is J(nominal) + this sum.
__device__ __forceinline__ float line_search_delta(float tau_i, float d_i, float s, float s2, float c_i)
{
    return fmaf(d_i, fmaf(0.5f, s, c_i), 0.5f * (tau_i - d_i) * s2);
}

// ---------------------------------------------------------------------------------------------------------------
// lqr_forward through the network (mpc/lqr_step.py:164-261), sixteen problems per wavefront.
// lane (q, r): problem r of the group, quarter q of every per-problem loop (controls i = q, q+4, ..; cost rows likewise);
// the state x' is the output accumulator: features 4q..4q+3.
// ---------------------------------------------------------------------------------------------------------------
template <bool WL>
__global__ void __launch_bounds__(256) nn_rollout_kernel(StepParams<float> p, MlpDesc m, int TS, int ZS, int wave_floats)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float *wts = stage_weights<WL>(m, lds);
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    float *tauS = lds + (WL ? m.total : 0) + wave * wave_floats, *dxS = tauS + 16 * TS, *zb = dxS + 16 * TS;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long group = (long)blockIdx.x * nwave + wave;
    if (group * 16 >= p.B) return;
    const long b_raw = group * 16 + r;
    const bool valid = b_raw < p.B;
    const long b = valid ? b_raw : p.B - 1;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T;
    const long B = p.B;
    const bool has_gain = p.K != nullptr, has_cost = p.C != nullptr;
    const float old_cost = p.old_costs_in ? p.old_costs_in[b] : 0.f;
    for (int i = lane; i < 16 * TS; i += 64) {
        tauS[i] = 0.f;
        dxS[i] = 0.f;
    }
    wave_sync();
    float alpha = 1.f, cost = 0.f, dun = 0.f, full = 0.f;
    bool active = valid;
    const int max_ls = has_gain ? p.max_ls : 1;
    for (int pass = 0; pass < max_ls; ++pass) {
        float xr[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int f = 4 * q + v;
            xr[v] = f < ns ? p.x_init[b * ns + f] : 0.f;
            if (active && f < ns) p.new_x[b * ns + f] = xr[v];
        }
        float ca = 0.f, da = 0.f;
        for (int t = 0; t < T; ++t) {
            const long tb = (long)t * B + b;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int f = 4 * q + v;
                if (f < ns) {
                    tauS[r * TS + f] = xr[v];
                    dxS[r * TS + f] = has_gain ? xr[v] - p.cur_x[tb * ns + f] : 0.f;      // :227
                }
            }
            wave_sync();
            for (int i = q; i < nc; i += 4) {
                const float u = p.cur_u[tb * nc + i];
                float un = u;
                if (has_gain) {
                    const float *Kr = p.K + (tb * nc + i) * ns;
                    float s = 0.f;
                    for (int j = 0; j < ns; ++j) s = fmaf(Kr[j], dxS[r * TS + j], s);
                    un = s + u + alpha * p.k[tb * nc + i];                                  // :192
                    if (p.zero_mask && p.zero_mask[tb * nc + i]) un = 0.f;                  // :197-198
                    if (p.bound_mode != MPC_BOUND_NONE) {                                   // :200-213
                        float lo = p.bound_mode == MPC_BOUND_SCALAR ? p.lo_s : p.lo[tb * nc + i];
                        float hi = p.bound_mode == MPC_BOUND_SCALAR ? p.hi_s : p.hi[tb * nc + i];
                        if (p.has_delta) {
                            const float l2 = u - p.delta_u, h2 = u + p.delta_u;
                            lo = (l2 < lo) ? lo : l2;
                            hi = (h2 > hi) ? hi : h2;
                        }
                        if (un < lo) un = lo;                                               // util.eclamp
                        if (un > hi) un = hi;
                    }
                    if (active) p.new_u[tb * nc + i] = un;
                }
                tauS[r * TS + ns + i] = un;
                const float d = u - un;
                dxS[r * TS + ns + i] = -d;          // (the control part of tau' - tau_nominal, for the cost difference below; K reads states only)
                da = fmaf(d, d, da);
            }
            wave_sync();
            if (has_cost) {                                                                 // :230-232
                const float *Ct = p.C + (long)t * p.C_st + b * p.C_sb;
                const float *ct = p.c + (long)t * p.c_st + b * p.c_sb;
                for (int i = q; i < n; i += 4) {
                    float s = 0.f, s2 = 0.f;
                    for (int j = 0; j < n; ++j) {
                        const float cij = Ct[i * n + j];
                        s = fmaf(cij, tauS[r * TS + j], s);
                        if (has_gain) s2 = fmaf(cij, dxS[r * TS + j], s2);
                    }
                    if (has_gain) {
                        // the line search compares J(tau') with J(nominal) (:176-179): summed as their DIFFERENCE, see line_search_delta
                        ca += line_search_delta(tauS[r * TS + i], dxS[r * TS + i], s, s2, ct[i]);
                    } else {
                        ca = fmaf(tauS[r * TS + i], fmaf(0.5f, s, ct[i]), ca);
                    }
                }
            }
            if (t < T - 1) {                                                                // :223-225
                const f32x4 o = mlp_forward<false>(m, wts, tauS, TS, zb, ZS, q, r);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int f = 4 * q + v;
                    // mpc/dynamics.py:74-75; an augmented state's first entries are the control just applied (:139-147)
                    const float skip = f < m.carry ? tauS[r * TS + ns + f] : (m.pass ? xr[v] : 0.f);
                    xr[v] = f < ns ? o[v] + skip : 0.f;
                    if (active && f < ns) p.new_x[((long)(t + 1) * B + b) * ns + f] = xr[v];
                }
                wave_sync();
            } else {
                wave_sync();
            }
        }
        ca += __shfl_xor(ca, 16);
        ca += __shfl_xor(ca, 32);
        da += __shfl_xor(da, 16);
        da += __shfl_xor(da, 32);
        const float dn = sqrtf(da);
        if (pass == 0) full = dn;                                                           // :243-245
        if (active) {
            cost = has_gain ? old_cost + ca : ca;       // (with gains `ca` is J(tau') - J(nominal))
            dun = dn;
            // :176-179, 247: keep shrinking while this problem's cost got worse
            if (has_gain && ca > 0.f && pass + 1 < max_ls) alpha *= p.ls_decay; else active = false;
        }
        if (!__any(active)) break;
    }
    if (q == 0 && valid) {
        if (p.costs) p.costs[b] = cost;
        if (p.old_costs) p.old_costs[b] = old_cost;
        if (p.full_du_norm) p.full_du_norm[b] = full;
        if (p.alpha_du_norm) p.alpha_du_norm[b] = dun;
        if (p.alphas) p.alphas[b] = alpha;
        if (p.status && (!(cost == cost) || fabsf(cost) > 3e38f)) p.status[b] |= MPC_ST_NONFINITE;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// F = d net / d [x;u], f = net(x, u) - F [x;u] at N points (mpc/mpc.py:495-512 + mpc/dynamics.py:82-128), sixteen
// points per wavefront: one forward pass for all sixteen, then the chain of one point at a time.  A product
// G_l = diag(s_l) W_l G_{l-1} is [width_l x n]: tile (to, tj) in accumulator layout is rows 16 to + 4q + v, column
// 16 tj + r -- written to this lane's own LDS slot and read back by this lane as the B operand of layer l + 1.
// ---------------------------------------------------------------------------------------------------------------
template <bool WL>
__global__ void __launch_bounds__(512) nn_linearize_kernel(MlpDesc m, long N, int ns, int nc, const float *x, const float *u,
                                                           float *F, float *f, int TS, int ZS, int GT, int wave_floats)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float *wts = stage_weights<WL>(m, lds);
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    float *tauS = lds + (WL ? m.total : 0) + wave * wave_floats, *zb = tauS + 16 * TS;
    f32x4 *G0 = reinterpret_cast<f32x4 *>(zb + (m.L > 1 ? m.L - 1 : 1) * 16 * ZS), *G1 = G0 + GT * 64;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long p0 = ((long)blockIdx.x * nwave + wave) * 16;
    if (p0 >= N) return;
    const long pt = (p0 + r < N) ? p0 + r : N - 1;
    const int n = ns + nc, NTJ = m.wp[0] >> 4;
    for (int f0 = 4 * q; f0 < m.wp[0]; f0 += 16) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int fe = f0 + v;
            tauS[r * TS + fe] = fe < ns ? x[pt * ns + fe] : (fe < n ? u[pt * nc + (fe - ns)] : 0.f);
        }
    }
    wave_sync();
    f32x4 out = mlp_forward<true>(m, wts, tauS, TS, zb, ZS, q, r);
    if (m.pass) {
#pragma unroll
        for (int v = 0; v < 4; ++v) out[v] += (4 * q + v < ns) ? tauS[r * TS + 4 * q + v] : 0.f;
    }
    const int Lh = m.L - 1;
    for (int pp = 0; pp < 16; ++pp) {
        if (p0 + pp >= N) break;
        f32x4 *Gprev = G0, *Gcur = G1;
        for (int l = 0; l < Lh; ++l) {
            const int nout_t = m.wp[l + 1] >> 4, nin_t = m.wp[l] >> 4, ld = m.wp[l] + 4;
            const float *zrow = zb + l * 16 * ZS + pp * ZS;
            const float *W = wts + m.woff[l];
            for (int to = 0; to < nout_t; ++to) {
                const f32x4 s = slope_fn(*reinterpret_cast<const f32x4 *>(zrow + 16 * to + 4 * q), m.act);
                for (int tj = 0; tj < NTJ; ++tj) {
                    f32x4 g;
                    if (l == 0) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) g[v] = s[v] * W[(16 * to + 4 * q + v) * ld + 16 * tj + r];
                    } else {
                        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                        const float *wrow = W + (16 * to + r) * ld + 4 * q;
                        for (int ti = 0; ti < nin_t; ++ti) {
                            const f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + 16 * ti);
                            const f32x4 gp = Gprev[(ti * NTJ + tj) * 64 + lane];
                            a0 = mfma(a[0], gp[0], a0);
                            a1 = mfma(a[1], gp[1], a1);
                            a2 = mfma(a[2], gp[2], a2);
                            a3 = mfma(a[3], gp[3], a3);
                        }
                        g = ((a0 + a1) + (a2 + a3)) * s;
                    }
                    Gcur[(to * NTJ + tj) * 64 + lane] = g;
                }
            }
            f32x4 *sw = Gprev;
            Gprev = Gcur;
            Gcur = sw;
        }
        // output layer: J = W_L G_{L-1}  (n_state <= 16: one row tile)
        float fs[4] = {0.f, 0.f, 0.f, 0.f};
        {
            const int l = Lh, nin_t = m.wp[l] >> 4, ld = m.wp[l] + 4;
            const float *W = wts + m.woff[l];
            for (int tj = 0; tj < NTJ; ++tj) {
                f32x4 J;
                if (Lh == 0) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) J[v] = W[(4 * q + v) * ld + 16 * tj + r];
                } else {
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                    const float *wrow = W + r * ld + 4 * q;
                    for (int ti = 0; ti < nin_t; ++ti) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + 16 * ti);
                        const f32x4 gp = Gprev[(ti * NTJ + tj) * 64 + lane];
                        a0 = mfma(a[0], gp[0], a0);
                        a1 = mfma(a[1], gp[1], a1);
                        a2 = mfma(a[2], gp[2], a2);
                        a3 = mfma(a[3], gp[3], a3);
                    }
                    J = (a0 + a1) + (a2 + a3);
                }
                const int j = 16 * tj + r;
                const float tj_tau = tauS[pp * TS + j];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 4 * q + v;
                    if (m.pass && i == j && j < ns) J[v] += 1.f;                            // mpc/dynamics.py:118-125
                    if (i < ns && j < n) F[((p0 + pp) * ns + i) * n + j] = J[v];
                    fs[v] = fmaf(J[v], tj_tau, fs[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float s = fs[v];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            fs[v] = s;
        }
        if (r == pp) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (4 * q + v < ns) f[(p0 + pp) * ns + 4 * q + v] = out[v] - fs[v];        // mpc/mpc.py:508-509
        }
    }
}

// ===============================================================================================================
// The network the reference builds by default -- ONE hidden layer (mpc/dynamics.py:16: hidden_sizes=[100]) -- with
// n_state + n_ctrl <= 16 and at most 16 HT hidden units: the whole network lives in registers.  A operands of both
// layers, the biases and (for the Jacobian) W_1 in accumulator layout are loaded once per wavefront; the hidden
// activations never leave the accumulators (tile `to` of layer 1's output IS K-step `to` of layer 2's B operand).
// ===============================================================================================================
template <int N> __device__ __forceinline__ float bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + N, 0xf, 0xf, true));   // row_newbcast:N
}

template <int HT> struct NetRegs {
    f32x4 w1[HT], b1[HT], w2[HT], b2;
};

template <int HT> __device__ __forceinline__ void load_net(NetRegs<HT> &R, const MlpDesc &m, int q, int r)
{
    const float *P = m.packed;
    const int ld2 = m.wp[1] + 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int to = 0; to < HT; ++to) {
        const bool in = 16 * to < m.wp[1];                      // (wave-uniform)
        const int tc = in ? to : 0;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(P + m.woff[0] + (16 * tc + r) * 20 + 4 * q);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(P + m.boff[0] + 16 * tc + 4 * q);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(P + m.woff[1] + r * ld2 + 16 * tc + 4 * q);
        R.w1[to] = in ? a : zero;
        R.b1[to] = in ? bb : zero;
        R.w2[to] = in ? c : zero;
    }
    R.b2 = *reinterpret_cast<const f32x4 *>(P + m.boff[1] + 4 * q);
}

// pre[to] = W_1 tau + b_1 (accumulator layout: hidden units 16 to + 4q + v, column r)
template <int HT> __device__ __forceinline__ void layer1(const NetRegs<HT> &R, f32x4 tq, f32x4 (&pre)[HT])
{
#pragma unroll
    for (int to = 0; to < HT; ++to) pre[to] = R.b1[to];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int to = 0; to < HT; ++to) pre[to] = mfma(R.w1[to][v], tq[v], pre[to]);
}
// the same with layer 1's operands (A operand and bias of every tile) read from this lane's own LDS slots instead of
// 8 HT registers: the line-search kernel needs those registers for a timestep's rows of C and K
template <int HT> __device__ __forceinline__ void layer1_lds(const f32x4 *w1s, const f32x4 *b1s, f32x4 tq, f32x4 (&pre)[HT])
{
#pragma unroll
    for (int to = 0; to < HT; ++to) {
        const f32x4 w = w1s[64 * to];
        f32x4 a = b1s[64 * to];
#pragma unroll
        for (int v = 0; v < 4; ++v) a = mfma(w[v], tq[v], a);
        pre[to] = a;
    }
}
// W_2 z (+ init), four accumulation chains
template <int HT> __device__ __forceinline__ f32x4 layer2(const NetRegs<HT> &R, const f32x4 (&z)[HT], f32x4 init)
{
    f32x4 a0 = init, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
#pragma unroll
    for (int to = 0; to < HT; ++to) {
        a0 = mfma(R.w2[to][0], z[to][0], a0);
        a1 = mfma(R.w2[to][1], z[to][1], a1);
        a2 = mfma(R.w2[to][2], z[to][2], a2);
        a3 = mfma(R.w2[to][3], z[to][3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}

// sixteen floats of a row that need not be 16-byte aligned or 16 long (columns >= len come back as 0).  Every load is
// unconditional from a clamped address and masked afterwards: a load under a lane-dependent condition becomes a branch
// around it, and the joins of those branches are where the compiler drains the whole load queue.
__device__ __forceinline__ void load_row(const float *row, int len, bool vec, f32x4 (&o)[4])
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (vec) {
            const bool in = 4 * g < len;
            const f32x4 t = *reinterpret_cast<const f32x4 *>(row + (in ? 4 * g : 0));
            o[g] = in ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = 4 * g + e < len;
                const float t = row[in ? 4 * g + e : 0];
                o[g][e] = in ? t : 0.f;
            }
        }
    }
}
__device__ __forceinline__ float load_if(const float *p, long idx, bool in)
{
    const float t = p[in ? idx : 0];
    return in ? t : 0.f;
}

// lqr_forward through the network, one wavefront per workgroup, sixteen problems per wavefront (see nn_rollout_kernel
// for the roles of the lanes).  What one step needs from memory -- rows 4q..4q+3 of C_t, c_t, row q of K_t, k_t, the
// nominal -- is requested one step ahead, before the network's MFMAs, and consumed after them.
// GAIN: line search with feedback gains (else the controls are the nominal ones); COST: the quadratic cost is summed.
// Rows are read 16 bytes at a time: n_state and n_state + n_ctrl multiples of 4 (else the general kernel runs).
// ACT: the activation as a compile-time constant (the exp / reciprocal of 28 sigmoids per lane and step are the largest
// block of vector work in the step; a runtime switch keeps all three variants and their branches inside the time loop)
template <int HT, bool GAIN, bool COST, int ACT>
__global__ void __launch_bounds__(64) nn_rollout_fast_kernel(StepParams<float> p, MlpDesc m)
{
    __shared__ __attribute__((aligned(16))) float tauS[16 * 20], dxS[16 * 20];
    __shared__ f32x4 netS[GAIN && COST ? 2 * HT * 64 : 1];        // layer 1 of the line-search kernel: [w1 | b1][tile][lane]
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const long b_raw = (long)blockIdx.x * 16 + r;
    const bool valid = b_raw < p.B;
    const long b = valid ? b_raw : p.B - 1;
    const long b_raw_clamped = b;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T;
    const long B = p.B;
    constexpr bool has_gain = GAIN, has_cost = COST, vecC = true, vecK = true;
    const float old_cost = p.old_costs_in ? p.old_costs_in[b] : 0.f;
    NetRegs<HT> R;
    load_net<HT>(R, m, q, r);
    for (int i = lane; i < 16 * 20; i += 64) {
        tauS[i] = 0.f;
        dxS[i] = 0.f;
    }
    if (GAIN && COST) {
        // (this instantiation sits at the 512-register limit: layer 1's operands move to LDS, two 16-byte reads per tile and step)
#pragma unroll
        for (int to = 0; to < HT; ++to) {
            netS[64 * to + lane] = R.w1[to];
            netS[64 * (HT + to) + lane] = R.b1[to];
            R.w1[to] = f32x4{0.f, 0.f, 0.f, 0.f};
            R.b1[to] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    wave_sync();
    float alpha = 1.f, cost = 0.f, dun = 0.f, full = 0.f;
    bool active = valid;
    const int max_ls = has_gain ? p.max_ls : 1;
    const bool own_u = q < nc;                  // this lane computes control i = q (n_ctrl <= 4)
    // The line search as a JOB SCHEDULE over the wave's sixteen problem slots (round 5).  The reference runs pass after pass over the
    // whole batch while any problem got worse (mpc/lqr_step.py:176-179); a wavefront did the same for its 16 problems -- and from the
    // third iLQR iteration on a few problems of most wavefronts need the 4th, 7th or 10th step size (tools/archive/r05_nn_iter_probe.py), so
    // the rollout of 0.14 ms ran 1.3 - 1.7 ms with one or two slots of sixteen at work (profiles/r05_trace_nn_before.txt).  The trials
    // of a problem do not depend on each other: after a pass, the slots nobody needs any more roll out FURTHER trials of the problems
    // that are still searching -- a problem's next S step sizes side by side, S = 16 / (problems searching) --, the first of them with
    // the stores on (as the sequential search would have it); a problem whose accepted trial is not the one in memory replays it
    // once.  Same step sizes (the same chain of float32 multiplications from 1), same acceptance rule -- the first trial that did not
    // get worse, else the last (:247-252) -- same results; three passes where the sequential search took up to max_ls.
    __shared__ int tabR[16], tabS[16];
    long bb = b;                                // the problem this lane's slot rolls out in the CURRENT pass
    float al = 1.f;                             // ... with this step size
    bool store = active;                        // ... writing new_x / new_u
    bool done_all = false;
    // the search state of this slot's OWN problem (what its q = 0 lane says counts; the other three carry copies)
    int next_t = 0, acc_t = -1, stored_t = -1;  // first trial not rolled out yet; the accepted one; the one whose trajectory is in memory
    bool done = !valid;
    // this pass's job of the slot, and where the own problem's jobs sit: slots [base, base + share), trials first_t ..
    int job_t = 0, base = r, share = valid ? 1 : 0, first_t = 0;
    bool job_valid = valid, replaying = false;
    auto chain = [&](int j) { float a = 1.f; for (int i = 0; i < j; ++i) a *= p.ls_decay; return a; };      // :247 alpha *= decay, j times
    auto schedule = [&]() {
        if (!done && acc_t < 0 && next_t >= max_ls) acc_t = max_ls - 1;                     // no trial improved: the last one stands (:250-252)
        const bool need_replay = !done && acc_t >= 0 && stored_t != acc_t;
        const bool searching = !done && acc_t < 0;
        const unsigned mR = (unsigned)__ballot(need_replay && q == 0), mS = (unsigned)__ballot(searching && q == 0);
        if ((mR | mS) == 0u) { done_all = true; return; }
        const int nR = __popc(mR), nS = __popc(mS), free_slots = 16 - nR;
        const int S = nS ? (free_slots >= nS ? free_slots / nS : 1) : 0;
        const unsigned below = (1u << r) - 1u;
        const int rankR = __popc(mR & below), rankS = __popc(mS & below);
        if (q == 0 && need_replay) tabR[rankR] = r;
        if (q == 0 && searching) tabS[rankS] = r;
        wave_sync();
        // the own problem's jobs of this pass
        replaying = need_replay;
        share = need_replay ? 1 : ((searching && rankS * S < free_slots) ? S : 0);
        base = need_replay ? rankR : nR + rankS * S;
        first_t = need_replay ? acc_t : next_t;
        // this slot's job
        const int k = r;
        int home = 0, off = 0;
        bool st = false;
        job_valid = false;
        if (k < nR) {
            home = tabR[k]; st = true; job_valid = true;
        } else if (S > 0 && (k - nR) / S < nS) {
            home = tabS[(k - nR) / S]; off = (k - nR) % S; st = off == 0; job_valid = true;
        }
        const int h_acc = __shfl(acc_t, home), h_next = __shfl(next_t, home);               // (lane `home`: q = 0 of that slot)
        job_t = (k < nR ? h_acc : h_next) + off;
        job_valid = job_valid && job_t < max_ls;
        bb = (long)__shfl((int)b_raw_clamped, home);
        al = chain(job_t);
        store = job_valid && st;
        wave_sync();
    };
    auto update_after_pass = [&](int pass, float ca, float dn) {
        if (!(GAIN && COST)) {                  // util.get_traj / get_cost: one pass, every slot its own problem
            cost = ca; dun = dn; full = dn; alpha = 1.f;
            done_all = true;
            return;
        }
        const unsigned ok = (unsigned)__ballot(q == 0 && job_valid && !(ca > 0.f));         // bit k: the trial in slot k did not get worse
        // the results of the own problem's first job (the one that stored) come from slot `base`
        const float ca0 = __shfl(ca, base), dn0 = __shfl(dn, base);
        if (!done && share > 0) {
            if (pass == 0) full = dn0;                                                      // :243-245 (trial 0 is alpha = 1)
            if (replaying) {
                cost = old_cost + ca0; dun = dn0; alpha = chain(acc_t);
                stored_t = acc_t; done = true;
            } else {
                const unsigned mine = (ok >> base) & ((1u << share) - 1u);
                const int n_run = min(share, max_ls - first_t);
                if (mine != 0u) acc_t = first_t + __ffs((int)mine) - 1;
                stored_t = first_t;
                next_t = first_t + n_run;
                // (what the sequential search would report had it stopped here: the trial in memory)
                cost = old_cost + ca0; dun = dn0; alpha = chain(first_t);
                if (acc_t < 0 && next_t >= max_ls) acc_t = max_ls - 1;
                if (acc_t == stored_t) done = true;
            }
        }
        schedule();
    };
    for (int pass = 0; ; ++pass) {
        const long b = bb;                      // (shadows the slot's own problem inside the pass)
        const float alpha = al;
        const bool active = store;
        f32x4 xr;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int f = 4 * q + v;
            xr[v] = load_if(p.x_init, b * ns + f, f < ns);
            if (active && f < ns) p.new_x[b * ns + f] = xr[v];
        }
        // operands of step t, requested one step ahead, before the network's MFMAs of step t - 1.  (Two steps ahead
        // with a second register set was no faster: at ~460 registers the loads serialise behind register reuse.)
        struct Ops {
            f32x4 Cr[4][4], Kr[4], cq, cx;
            float kq, uq;
        } opA;
        auto request = [&](int t, Ops &o) {
            const long tb = (long)t * B + b;
            if (has_cost) {
                const float *Ct = p.C + (long)t * p.C_st + b * p.C_sb, *ct = p.c + (long)t * p.c_st + b * p.c_sb;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 4 * q + v < n ? 4 * q + v : 0;
                    load_row(Ct + i * n, 4 * q + v < n ? n : 0, vecC, o.Cr[v]);
                    o.cq[v] = load_if(ct, i, 4 * q + v < n);
                }
            }
            if (has_gain) {
                load_row(p.K + (tb * nc + (own_u ? q : 0)) * ns, own_u ? ns : 0, vecK, o.Kr);
                o.kq = load_if(p.k, tb * nc + q, own_u);
#pragma unroll
                for (int v = 0; v < 4; ++v) o.cx[v] = load_if(p.cur_x, tb * ns + 4 * q + v, 4 * q + v < ns);
            }
            o.uq = load_if(p.cur_u, tb * nc + q, own_u);
        };
        float ca = 0.f, da = 0.f;
        auto step = [&](int t, Ops &o) {
            const long tb = (long)t * B + b;
            // the state quad goes first: a quad that straddles n_state also covers control slots, written next
            *reinterpret_cast<f32x4 *>(tauS + r * 20 + 4 * q) = xr;
            *reinterpret_cast<f32x4 *>(dxS + r * 20 + 4 * q) = has_gain ? xr - o.cx : f32x4{0.f, 0.f, 0.f, 0.f};   // :227
            wave_sync();
            if (own_u) {
                const float uq = o.uq;
                float un = uq;
                if (has_gain) {
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 d = *reinterpret_cast<const f32x4 *>(dxS + r * 20 + 4 * g);
                        s0 = fmaf(o.Kr[g][0], d[0], s0);
                        s1 = fmaf(o.Kr[g][1], d[1], s1);
                        s0 = fmaf(o.Kr[g][2], d[2], s0);
                        s1 = fmaf(o.Kr[g][3], d[3], s1);
                    }
                    un = (s0 + s1) + uq + alpha * o.kq;                                     // :192
                    if (p.zero_mask && p.zero_mask[tb * nc + q]) un = 0.f;                  // :197-198
                    if (p.bound_mode != MPC_BOUND_NONE) {                                   // :200-213
                        float lo = p.bound_mode == MPC_BOUND_SCALAR ? p.lo_s : p.lo[tb * nc + q];
                        float hi = p.bound_mode == MPC_BOUND_SCALAR ? p.hi_s : p.hi[tb * nc + q];
                        if (p.has_delta) {
                            const float l2 = uq - p.delta_u, h2 = uq + p.delta_u;
                            lo = (l2 < lo) ? lo : l2;
                            hi = (h2 > hi) ? hi : h2;
                        }
                        if (un < lo) un = lo;                                               // util.eclamp
                        if (un > hi) un = hi;
                    }
                    if (active) p.new_u[tb * nc + q] = un;
                }
                tauS[r * 20 + ns + q] = un;
                const float d = uq - un;
                if (has_gain) dxS[r * 20 + ns + q] = -d;      // (the control part of tau' - tau_nominal; K's padded columns are zero)
                da = fmaf(d, d, da);
            }
            wave_sync();
            const f32x4 tq = *reinterpret_cast<const f32x4 *>(tauS + r * 20 + 4 * q);
            if (has_cost) {                                                                 // :230-232
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 tg = *reinterpret_cast<const f32x4 *>(tauS + r * 20 + 4 * g);
                    const f32x4 dg = has_gain ? *reinterpret_cast<const f32x4 *>(dxS + r * 20 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s[v] = fmaf(o.Cr[v][g][e], tg[e], s[v]);
                            if (has_gain) s2[v] = fmaf(o.Cr[v][g][e], dg[e], s2[v]);
                        }
                }
                if (has_gain) {
                    // J(tau') - J(nominal), summed as a difference (line_search_delta)
                    const f32x4 dq = *reinterpret_cast<const f32x4 *>(dxS + r * 20 + 4 * q);
#pragma unroll
                    for (int v = 0; v < 4; ++v) ca += line_search_delta(tq[v], dq[v], s[v], s2[v], o.cq[v]);
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) ca = fmaf(tq[v], fmaf(0.5f, s[v], o.cq[v]), ca);
                }
            }
            if (t + 1 < T) request(t + 1, o);
            if (t + 1 < T) {
                f32x4 z[HT];                                                                // :223-225
                if (GAIN && COST) layer1_lds<HT>(netS + lane, netS + 64 * HT + lane, tq, z); else layer1<HT>(R, tq, z);
#pragma unroll
                for (int to = 0; to < HT; ++to) z[to] = act_fn(z[to], ACT);
                const f32x4 out = layer2<HT>(R, z, R.b2);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int f = 4 * q + v;
                    const float skip = f < m.carry ? tauS[r * 20 + ns + f] : (m.pass ? xr[v] : 0.f);   // mpc/dynamics.py:74-75, :139-147
                    xr[v] = f < ns ? out[v] + skip : 0.f;
                    if (active && f < ns) p.new_x[((long)(t + 1) * B + b) * ns + f] = xr[v];
                }
            }
            wave_sync();
        };
        request(0, opA);
        for (int t = 0; t < T; ++t) step(t, opA);
        ca += __shfl_xor(ca, 16);
        ca += __shfl_xor(ca, 32);
        da += __shfl_xor(da, 16);
        da += __shfl_xor(da, 32);
        const float dn = sqrtf(da);
        update_after_pass(pass, ca, dn);
        if (done_all) break;
    }
    if (q == 0 && valid) {
        if (p.costs) p.costs[b] = cost;
        if (p.old_costs) p.old_costs[b] = old_cost;
        if (p.full_du_norm) p.full_du_norm[b] = full;
        if (p.alpha_du_norm) p.alpha_du_norm[b] = dun;
        if (p.alphas) p.alphas[b] = alpha;
        if (p.status && (!(cost == cost) || fabsf(cost) > 3e38f)) p.status[b] |= MPC_ST_NONFINITE;
    }
}

// F, f at sixteen points per wavefront, everything in registers.  Per point the chain W_2 diag(s) W_1 is 4 HT MFMAs whose
// B operand is s (of THAT point: a DPP row broadcast of the lane that holds it) times W_1 in accumulator layout.
// f = net - F tau needs no reduction either: net - F tau = W_2 (z - s .* (W_1 tau)) + b_2, one more pass of layer 2.
template <int HT, int PP> struct JacPoints {
    static __device__ __forceinline__ void run(const NetRegs<HT> &R, const f32x4 (&w1d)[HT], const f32x4 (&s)[HT], int pass,
                                               long p0, long N, int ns, int n, int q, int r, float *F)
    {
        if (p0 + PP < N) {
            // all B operands of the point first, then the MFMAs as one block: an MFMA issued right behind the vector
            // instruction that feeds it costs 56 clocks instead of 32 (tools/ubench/mfma16_turn.hip)
            f32x4 g[HT];
#pragma unroll
            for (int to = 0; to < HT; ++to)
#pragma unroll
                for (int v = 0; v < 4; ++v) g[to][v] = bcast<PP>(s[to][v]) * w1d[to][v];
            __builtin_amdgcn_sched_barrier(0);
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
            for (int to = 0; to < HT; ++to) {
                a0 = mfma(R.w2[to][0], g[to][0], a0);
                a1 = mfma(R.w2[to][1], g[to][1], a1);
                a2 = mfma(R.w2[to][2], g[to][2], a2);
                a3 = mfma(R.w2[to][3], g[to][3], a3);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 J = (a0 + a1) + (a2 + a3);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 4 * q + v;
                if (i < ns && r < n) F[((p0 + PP) * ns + i) * n + r] = J[v] + ((pass && i == r) ? 1.f : 0.f);   // mpc/dynamics.py:118-125
            }
        }
        JacPoints<HT, PP + 1>::run(R, w1d, s, pass, p0, N, ns, n, q, r, F);
    }
};
template <int HT> struct JacPoints<HT, 16> {
    static __device__ __forceinline__ void run(const NetRegs<HT> &, const f32x4 (&)[HT], const f32x4 (&)[HT], int, long, long, int, int,
                                               int, int, float *) {}
};

template <int HT, int ACT>
__global__ void __launch_bounds__(64) nn_linearize_fast_kernel(MlpDesc m, long N, int ns, int nc, const float *x, const float *u,
                                                               float *F, float *f)
{
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const long p0 = (long)blockIdx.x * 16;
    const long pt = (p0 + r < N) ? p0 + r : N - 1;
    const int n = ns + nc;
    NetRegs<HT> R;
    load_net<HT>(R, m, q, r);
    f32x4 w1d[HT];                      // W_1 in accumulator layout: rows (hidden units) 16 to + 4q + v, column r
#pragma unroll
    for (int to = 0; to < HT; ++to)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            w1d[to][v] = load_if(m.packed, m.woff[0] + (16 * to + 4 * q + v) * 20 + r, 16 * to < m.wp[1]);
    f32x4 tq;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int fe = 4 * q + v;
        const float tx = load_if(x, pt * ns + fe, fe < ns), tu = load_if(u, pt * nc + (fe - ns), fe >= ns && fe < n);
        tq[v] = fe < ns ? tx : tu;
    }
    f32x4 z[HT], s[HT];
    layer1<HT>(R, tq, z);
#pragma unroll
    for (int to = 0; to < HT; ++to) {
        const f32x4 lin = z[to] - R.b1[to];                     // W_1 tau
        z[to] = act_fn(z[to], ACT);
        s[to] = slope_fn(z[to], ACT);
        z[to] = z[to] - s[to] * lin;
    }
    const f32x4 fv = layer2<HT>(R, z, R.b2);                    // net(x, u) - F [x;u]  (the passthrough x cancels)
    if (p0 + r < N) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (4 * q + v < ns) f[pt * ns + 4 * q + v] = fv[v];                              // mpc/mpc.py:508-509
    }
    JacPoints<HT, 0>::run(R, w1d, s, m.pass, p0, N, ns, n, q, r, F);
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

int pad16(int v) { return (v + 15) & ~15; }

// validates the network, lays the packed block out in the workspace and launches the packing kernel
int mlp_prepare(const mpc_mlp_dynamics *net, int ns, int nc, void *workspace, int64_t bytes, MlpDesc &d, hipStream_t st)
{
    if (!net) { set_last_error("network is NULL"); return MPC_E_NULL; }
    if (net->n_layers < 1 || net->n_layers > MPC_MLP_MAX_LAYERS) { set_last_error("network: 1..4 Linear layers"); return MPC_E_ARG; }
    if (net->activation < MPC_ACT_SIGMOID || net->activation > MPC_ACT_ELU) { set_last_error("network: unknown activation"); return MPC_E_ARG; }
    if (net->widths[0] != ns + nc || net->widths[net->n_layers] != ns) { set_last_error("network: widths[0] must be n_state + n_ctrl, widths[L] n_state"); return MPC_E_DIMS; }
    if (ns > 16) { set_last_error("network kernels: n_state <= 16"); return MPC_E_DIMS; }
    for (int l = 0; l <= net->n_layers; ++l)
        if (net->widths[l] < 1 || net->widths[l] > 4096) { set_last_error("network: layer width out of range"); return MPC_E_DIMS; }
    if (mpc_mlp_workspace_bytes(net) > bytes || !workspace || ((uintptr_t)workspace & 15)) {
        set_last_error("network: workspace too small or not 16-byte aligned (see mpc_mlp_workspace_bytes)");
        return MPC_E_ARG;
    }
    PackArgs a;
    d.L = net->n_layers;
    d.act = net->activation;
    d.pass = net->passthrough ? 1 : 0;
    d.carry = net->ctrl_carry;
    if (d.carry != 0 && d.carry != nc) { set_last_error("network: ctrl_carry must be 0 or n_ctrl"); return MPC_E_ARG; }
    if (d.carry >= ns) { set_last_error("network: ctrl_carry needs n_state (augmented) > n_ctrl"); return MPC_E_DIMS; }
    for (int l = 0; l <= d.L; ++l) {
        d.w[l] = net->widths[l];
        d.wp[l] = pad16(net->widths[l]);
    }
    int off = 0;
    for (int l = 0; l < d.L; ++l) {
        if (!net->W[l] || !net->b[l]) { set_last_error("network: weight / bias pointer is NULL"); return MPC_E_NULL; }
        a.W[l] = (const float *)net->W[l];
        a.b[l] = (const float *)net->b[l];
        d.woff[l] = off; off += d.wp[l + 1] * (d.wp[l] + 4);
        d.boff[l] = off; off += d.wp[l + 1];
    }
    d.total = off;
    d.packed = (const float *)workspace;
    a.d = d;
    a.dst = (float *)workspace;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(64), dim3(256), 0, st, a);
    return check_launch("mlp_pack_kernel");
}

int max_hidden_pad(const MlpDesc &d)
{
    int h = 16;
    for (int l = 1; l < d.L; ++l) h = d.wp[l] > h ? d.wp[l] : h;
    return h;
}

// LDS budget of a workgroup: the packed network (when it is staged) + one staging area per wavefront
constexpr size_t LDS_MAX = 160 * 1024, WEIGHTS_IN_LDS_MAX = 96 * 1024;

template <typename K> void allow_lds(K kernel, size_t lds)
{
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

// Does the general (LDS-staged) kernel of either entry point fit a network of these layer widths?  The same two budget
// tests the launchers below apply -- exported as mpc_mlp_supported so that a caller can ask BEFORE it routes a module here
// (a 1024-unit layer is a legal NNDynamics, it just keeps the host-driven path).  bit 0: rollout, bit 1: linearisation.
int nn_budget(const mpc_mlp_dynamics *net, int ns, int nc)
{
    if (!net || net->n_layers < 1 || net->n_layers > MPC_MLP_MAX_LAYERS || ns < 1 || ns > 16 || nc < 1) return 0;
    if (net->widths[0] != ns + nc || net->widths[net->n_layers] != ns) return 0;
    int wp[MPC_MLP_MAX_LAYERS + 1], total = 0, hid = 16;
    for (int l = 0; l <= net->n_layers; ++l) {
        if (net->widths[l] < 1 || net->widths[l] > 4096) return 0;
        wp[l] = pad16(net->widths[l]);
    }
    for (int l = 0; l < net->n_layers; ++l) total += wp[l + 1] * (wp[l] + 4) + wp[l + 1];
    for (int l = 1; l < net->n_layers; ++l) hid = wp[l] > hid ? wp[l] : hid;
    const int TS = wp[0] + 4, ZS = hid + 4, NTJ = wp[0] >> 4, L = net->n_layers;
    int GT = 1;
    for (int l = 1; l < L; ++l) GT = (wp[l] >> 4) * NTJ > GT ? (wp[l] >> 4) * NTJ : GT;
    const size_t wbytes = (size_t)total * 4;
    const size_t roll = (size_t)(2 * 16 * TS + 2 * 16 * ZS) * 4;
    const size_t lin = (size_t)(16 * TS + (L > 1 ? L - 1 : 1) * 16 * ZS + 2 * GT * 64 * 4) * 4;
    auto fits = [&](size_t per_wave) { return (wbytes <= WEIGHTS_IN_LDS_MAX && wbytes + per_wave <= LDS_MAX) || per_wave <= LDS_MAX; };
    const bool fast = L == 2 && wp[0] == 16 && wp[1] <= 128;          // the register-resident kernels: no staging at all
    return ((fast || fits(roll)) ? 1 : 0) | ((fast || fits(lin)) ? 2 : 0);
}

int launch_nn_rollout(const StepParams<float> &p, const mpc_mlp_dynamics *net, void *workspace, int64_t bytes, hipStream_t st)
{
    MlpDesc d;
    int rc = mlp_prepare(net, p.ns, p.nc, workspace, bytes, d, st);
    if (rc) return rc;
    const bool rows16 = (p.ns & 3) == 0 && ((p.ns + p.nc) & 3) == 0 && (!p.C || ((p.C_st & 3) == 0 && (p.C_sb & 3) == 0));
    if (d.L == 2 && d.wp[0] == 16 && d.wp[1] <= 128 && p.nc <= 4 && rows16) {
        // one hidden layer of <= 128 units, n <= 16: the register-resident kernel
        const unsigned g = (unsigned)(((long)p.B + 15) / 16);
        const int ht = d.wp[1] >> 4;
        const int mode = p.K ? 2 : (p.C ? 1 : 0);
#define MPC_NN_LAUNCH_A(HT_, ACT_)                                                                                                \
        do {                                                                                                                       \
            if (mode == 2) hipLaunchKernelGGL((nn_rollout_fast_kernel<HT_, true, true, ACT_>), dim3(g), dim3(64), 0, st, p, d);      \
            else if (mode == 1) hipLaunchKernelGGL((nn_rollout_fast_kernel<HT_, false, true, ACT_>), dim3(g), dim3(64), 0, st, p, d); \
            else hipLaunchKernelGGL((nn_rollout_fast_kernel<HT_, false, false, ACT_>), dim3(g), dim3(64), 0, st, p, d);              \
        } while (0)
#define MPC_NN_LAUNCH(HT_)                                                                  \
        do {                                                                                 \
            if (d.act == MPC_ACT_SIGMOID) MPC_NN_LAUNCH_A(HT_, MPC_ACT_SIGMOID);             \
            else if (d.act == MPC_ACT_RELU) MPC_NN_LAUNCH_A(HT_, MPC_ACT_RELU);              \
            else MPC_NN_LAUNCH_A(HT_, MPC_ACT_ELU);                                          \
        } while (0)
        if (ht <= 2) MPC_NN_LAUNCH(2);
        else if (ht <= 4) MPC_NN_LAUNCH(4);
        else if (ht <= 7) MPC_NN_LAUNCH(7);
        else MPC_NN_LAUNCH(8);
#undef MPC_NN_LAUNCH_A
#undef MPC_NN_LAUNCH
        return check_launch("nn_rollout_fast_kernel");
    }
    const int TS = d.wp[0] + 4, ZS = max_hidden_pad(d) + 4;
    const int wave_floats = 2 * 16 * TS + 2 * 16 * ZS;
    const size_t wbytes = (size_t)d.total * 4, per_wave = (size_t)wave_floats * 4;
    const bool wl = wbytes <= WEIGHTS_IN_LDS_MAX && wbytes + per_wave <= LDS_MAX;
    if (!wl && per_wave > LDS_MAX) { set_last_error("network: layers too wide for the LDS-resident kernel"); return MPC_E_DIMS; }
    // sixteen problems per wavefront; few groups -> one wavefront per workgroup so that every CU gets one
    const long groups = ((long)p.B + 15) / 16;
    int nw = groups >= 2048 ? 4 : (groups >= 1024 ? 2 : 1);
    while (nw > 1 && (wl ? wbytes : 0) + nw * per_wave > LDS_MAX) nw >>= 1;
    const size_t lds = (wl ? wbytes : 0) + nw * per_wave;
    const unsigned grid = (unsigned)((groups + nw - 1) / nw);
    if (wl) {
        allow_lds(&nn_rollout_kernel<true>, lds);
        hipLaunchKernelGGL(nn_rollout_kernel<true>, dim3(grid), dim3(64 * nw), lds, st, p, d, TS, ZS, wave_floats);
    } else {
        allow_lds(&nn_rollout_kernel<false>, lds);
        hipLaunchKernelGGL(nn_rollout_kernel<false>, dim3(grid), dim3(64 * nw), lds, st, p, d, TS, ZS, wave_floats);
    }
    return check_launch("nn_rollout_kernel");
}

int launch_nn_linearize(const mpc_mlp_dynamics *net, long N, int ns, int nc, const float *x, const float *u, float *F,
                        float *f, void *workspace, int64_t bytes, hipStream_t st)
{
    MlpDesc d;
    if (net && net->ctrl_carry) { set_last_error("mlp_linearize: ctrl_carry describes a rollout only (linearise the network itself)"); return MPC_E_ARG; }
    int rc = mlp_prepare(net, ns, nc, workspace, bytes, d, st);
    if (rc) return rc;
    if (d.L == 2 && d.wp[0] == 16 && d.wp[1] <= 128) {
        const unsigned g = (unsigned)((N + 15) / 16);
        const int ht = d.wp[1] >> 4;
#define MPC_NN_LIN(HT_)                                                                                                             \
        do {                                                                                                                        \
            if (d.act == MPC_ACT_SIGMOID) hipLaunchKernelGGL((nn_linearize_fast_kernel<HT_, MPC_ACT_SIGMOID>), dim3(g), dim3(64), 0, st, d, N, ns, nc, x, u, F, f); \
            else if (d.act == MPC_ACT_RELU) hipLaunchKernelGGL((nn_linearize_fast_kernel<HT_, MPC_ACT_RELU>), dim3(g), dim3(64), 0, st, d, N, ns, nc, x, u, F, f);   \
            else hipLaunchKernelGGL((nn_linearize_fast_kernel<HT_, MPC_ACT_ELU>), dim3(g), dim3(64), 0, st, d, N, ns, nc, x, u, F, f);                              \
        } while (0)
        if (ht <= 2) MPC_NN_LIN(2);
        else if (ht <= 4) MPC_NN_LIN(4);
        else if (ht <= 7) MPC_NN_LIN(7);
        else MPC_NN_LIN(8);
#undef MPC_NN_LIN
        return check_launch("nn_linearize_fast_kernel");
    }
    const int TS = d.wp[0] + 4, ZS = max_hidden_pad(d) + 4, NTJ = d.wp[0] >> 4;
    int GT = 1;
    for (int l = 1; l < d.L; ++l) GT = (d.wp[l] >> 4) * NTJ > GT ? (d.wp[l] >> 4) * NTJ : GT;
    const int wave_floats = 16 * TS + (d.L > 1 ? d.L - 1 : 1) * 16 * ZS + 2 * GT * 64 * 4;
    const size_t wbytes = (size_t)d.total * 4, per_wave = (size_t)wave_floats * 4;
    const bool wl = wbytes <= WEIGHTS_IN_LDS_MAX && wbytes + per_wave <= LDS_MAX;
    if (!wl && per_wave > LDS_MAX) { set_last_error("network: layers too wide for the LDS-resident kernel"); return MPC_E_DIMS; }
    const long groups = (N + 15) / 16;
    int nw = groups >= 4096 ? 8 : (groups >= 2048 ? 4 : (groups >= 1024 ? 2 : 1));
    while (nw > 1 && (wl ? wbytes : 0) + nw * per_wave > LDS_MAX) nw >>= 1;
    const size_t lds = (wl ? wbytes : 0) + nw * per_wave;
    const unsigned grid = (unsigned)((groups + nw - 1) / nw);
    if (wl) {
        allow_lds(&nn_linearize_kernel<true>, lds);
        hipLaunchKernelGGL(nn_linearize_kernel<true>, dim3(grid), dim3(64 * nw), lds, st, d, N, ns, nc, x, u, F, f, TS, ZS, GT, wave_floats);
    } else {
        allow_lds(&nn_linearize_kernel<false>, lds);
        hipLaunchKernelGGL(nn_linearize_kernel<false>, dim3(grid), dim3(64 * nw), lds, st, d, N, ns, nc, x, u, F, f, TS, ZS, GT, wave_floats);
    }
    return check_launch("nn_linearize_kernel");
}

}  // namespace mpclqr
