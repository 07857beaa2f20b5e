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

struct MlpDesc {
    int L, act, pass;            // Linear layers, MPC_ACT_*, passthrough
    int w[MPC_MLP_MAX_LAYERS + 1];   // widths: w[0] = n_state + n_ctrl, w[L] = n_state
    int wp[MPC_MLP_MAX_LAYERS + 1];  // rounded up to 16
    const float *W[MPC_MLP_MAX_LAYERS];   // packed [wp[l+1]][wp[l]]
    const float *b[MPC_MLP_MAX_LAYERS];   // packed [wp[l+1]]
};

struct PackArgs {
    int L;
    int w[MPC_MLP_MAX_LAYERS + 1], wp[MPC_MLP_MAX_LAYERS + 1];
    const float *W[MPC_MLP_MAX_LAYERS], *b[MPC_MLP_MAX_LAYERS];
    float *Wp[MPC_MLP_MAX_LAYERS], *bp[MPC_MLP_MAX_LAYERS];
};

__global__ void mlp_pack_kernel(PackArgs a)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int l = 0; l < a.L; ++l) {
        const int in = a.w[l], out = a.w[l + 1], inp = a.wp[l], outp = a.wp[l + 1];
        for (int e = tid; e < outp * inp; e += nt) {
            const int o = e / inp, i = e - o * inp;
            a.Wp[l][e] = (o < out && i < in) ? a.W[l][o * in + i] : 0.f;
        }
        for (int o = tid; o < outp; o += nt) a.bp[l][o] = o < out ? a.b[l][o] : 0.f;
    }
}

__device__ __forceinline__ float act_fn(float a, int kind)
{
    if (kind == MPC_ACT_SIGMOID) return 1.f / (1.f + expf(-a));
    if (kind == MPC_ACT_RELU) return fmaxf(a, 0.f);
    return a > 0.f ? a : expm1f(a);                                   // F.elu, alpha = 1
}
// derivative of the activation from its OUTPUT (mpc/dynamics.py:104-112)
__device__ __forceinline__ float slope_fn(float z, int kind)
{
    if (kind == MPC_ACT_SIGMOID) return z * (1.f - z);
    if (kind == MPC_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    return z > 0.f ? 1.f : z + 1.f;
}

// All layers for the sixteen columns staged in tauS (row r = column's [x;u], zero padded to wp[0]).  Hidden
// activations go to zbase: layer l at zbase + l * 16 * ZS when KEEP (the Jacobian needs them all), else ping-pong.
// Returns the output layer's accumulator: features 4q..4q+3 of column r (n_state <= 16: one tile).
template <bool KEEP>
__device__ f32x4 mlp_forward(const MlpDesc &m, const float *tauS, int TS, float *zbase, int ZS, int q, int r)
{
    const float *in = tauS;
    int is = TS;
    f32x4 res = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < m.L; ++l) {
        const int nin_t = m.wp[l] >> 4, nout_t = m.wp[l + 1] >> 4, ld = m.wp[l];
        const float *W = m.W[l], *bias = m.b[l];
        float *dst = zbase + (KEEP ? l : (l & 1)) * 16 * ZS;
        const bool last = l + 1 == m.L;
        for (int to = 0; to < nout_t; ++to) {
            // four accumulation chains (one per K-block of a 16-byte load), joined at the end
            f32x4 a0 = *reinterpret_cast<const f32x4 *>(bias + 16 * to + 4 * q);
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
            const float *wrow = W + (long)(16 * to + r) * ld + 4 * q;
            const float *irow = in + r * is + 4 * q;
            for (int ti = 0; ti < nin_t; ++ti) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + 16 * ti);
                const f32x4 b = *reinterpret_cast<const f32x4 *>(irow + 16 * ti);
                a0 = mfma(a[0], b[0], a0);
                a1 = mfma(a[1], b[1], a1);
                a2 = mfma(a[2], b[2], a2);
                a3 = mfma(a[3], b[3], a3);
            }
            f32x4 acc = (a0 + a1) + (a2 + a3);
            if (!last) {
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = act_fn(acc[v], m.act);
                *reinterpret_cast<f32x4 *>(dst + r * ZS + 16 * to + 4 * q) = acc;
            } else if (to == 0) {
                res = acc;
            }
        }
        __syncthreads();
        in = dst;
        is = ZS;
    }
    return res;
}

// ---------------------------------------------------------------------------------------------------------------
// lqr_forward through the network (mpc/lqr_step.py:164-261), sixteen problems per wavefront.
// lane (q, r): problem r of the group, quarter q of every per-problem loop (controls i = q, q+4, ..; cost rows likewise);
// the state x' is the output accumulator: features 4q..4q+3.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) nn_rollout_kernel(StepParams<float> p, MlpDesc m, int TS, int ZS)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tauS = lds, *dxS = tauS + 16 * TS, *zb = dxS + 16 * TS;
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const int b_raw = blockIdx.x * 16 + r;
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
    __syncthreads();
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
            __syncthreads();
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
                da = fmaf(d, d, da);
            }
            __syncthreads();
            if (has_cost) {                                                                 // :230-232
                const float *Ct = p.C + (long)t * p.C_st + b * p.C_sb;
                const float *ct = p.c + (long)t * p.c_st + b * p.c_sb;
                for (int i = q; i < n; i += 4) {
                    float s = 0.f;
                    for (int j = 0; j < n; ++j) s = fmaf(Ct[i * n + j], tauS[r * TS + j], s);
                    ca = fmaf(tauS[r * TS + i], fmaf(0.5f, s, ct[i]), ca);
                }
            }
            if (t < T - 1) {                                                                // :223-225
                const f32x4 o = mlp_forward<false>(m, tauS, TS, zb, ZS, q, r);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int f = 4 * q + v;
                    xr[v] = f < ns ? o[v] + (m.pass ? xr[v] : 0.f) : 0.f;                   // mpc/dynamics.py:74-75
                    if (active && f < ns) p.new_x[((long)(t + 1) * B + b) * ns + f] = xr[v];
                }
            } else {
                __syncthreads();
            }
        }
        ca += __shfl_xor(ca, 16);
        ca += __shfl_xor(ca, 32);
        da += __shfl_xor(da, 16);
        da += __shfl_xor(da, 32);
        const float dn = sqrtf(da);
        if (pass == 0) full = dn;                                                           // :243-245
        if (active) {
            cost = ca;
            dun = dn;
            // :176-179, 247: keep shrinking while this problem's cost got worse
            if (has_gain && ca > old_cost && pass + 1 < max_ls) alpha *= p.ls_decay; else active = false;
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
__global__ void __launch_bounds__(64) nn_linearize_kernel(MlpDesc m, long N, int ns, int nc, const float *x, const float *u,
                                                          float *F, float *f, int TS, int ZS, int GT)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tauS = lds, *zb = tauS + 16 * TS;
    f32x4 *G0 = reinterpret_cast<f32x4 *>(zb + (m.L > 1 ? m.L - 1 : 1) * 16 * ZS), *G1 = G0 + (long)GT * 64;
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const long p0 = (long)blockIdx.x * 16;
    const long pt = (p0 + r < N) ? p0 + r : N - 1;
    const int n = ns + nc, NTJ = m.wp[0] >> 4;
    for (int f0 = 4 * q; f0 < m.wp[0]; f0 += 16) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int fe = f0 + v;
            tauS[r * TS + fe] = fe < ns ? x[pt * ns + fe] : (fe < n ? u[pt * nc + (fe - ns)] : 0.f);
        }
    }
    __syncthreads();
    f32x4 out = mlp_forward<true>(m, tauS, TS, zb, ZS, q, r);
    if (m.pass) {
#pragma unroll
        for (int v = 0; v < 4; ++v) out[v] += (4 * q + v < ns) ? tauS[r * TS + 4 * q + v] : 0.f;
    }
    const int Lh = m.L - 1;
    for (int pp = 0; pp < 16; ++pp) {
        if (p0 + pp >= N) break;
        f32x4 *Gprev = G0, *Gcur = G1;
        for (int l = 0; l < Lh; ++l) {
            const int nout_t = m.wp[l + 1] >> 4, nin_t = m.wp[l] >> 4, ld = m.wp[l];
            const float *zrow = zb + l * 16 * ZS + pp * ZS;
            const float *W = m.W[l];
            for (int to = 0; to < nout_t; ++to) {
                f32x4 s = *reinterpret_cast<const f32x4 *>(zrow + 16 * to + 4 * q);
#pragma unroll
                for (int v = 0; v < 4; ++v) s[v] = slope_fn(s[v], m.act);
                for (int tj = 0; tj < NTJ; ++tj) {
                    f32x4 g;
                    if (l == 0) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) g[v] = s[v] * W[(long)(16 * to + 4 * q + v) * ld + 16 * tj + r];
                    } else {
                        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                        const float *wrow = W + (long)(16 * to + r) * ld + 4 * q;
                        for (int ti = 0; ti < nin_t; ++ti) {
                            const f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + 16 * ti);
                            const f32x4 gp = Gprev[(long)(ti * NTJ + tj) * 64 + lane];
                            a0 = mfma(a[0], gp[0], a0);
                            a1 = mfma(a[1], gp[1], a1);
                            a2 = mfma(a[2], gp[2], a2);
                            a3 = mfma(a[3], gp[3], a3);
                        }
                        g = ((a0 + a1) + (a2 + a3)) * s;
                    }
                    Gcur[(long)(to * NTJ + tj) * 64 + lane] = g;
                }
            }
            f32x4 *sw = Gprev;
            Gprev = Gcur;
            Gcur = sw;
        }
        // output layer: J = W_L G_{L-1}  (n_state <= 16: one row tile)
        float fs[4] = {0.f, 0.f, 0.f, 0.f};
        {
            const int l = Lh, nin_t = m.wp[l] >> 4, ld = m.wp[l];
            const float *W = m.W[l];
            for (int tj = 0; tj < NTJ; ++tj) {
                f32x4 J;
                if (Lh == 0) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) J[v] = W[(long)(4 * q + v) * ld + 16 * tj + r];
                } else {
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                    const float *wrow = W + (long)r * ld + 4 * q;
                    for (int ti = 0; ti < nin_t; ++ti) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + 16 * ti);
                        const f32x4 gp = Gprev[(long)(ti * NTJ + tj) * 64 + lane];
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

// validates the network, lays the packed weights out in the workspace and launches the packing kernel
int mlp_prepare(const mpc_mlp_dynamics *net, int ns, int nc, void *workspace, int64_t bytes, MlpDesc &d, hipStream_t st)
{
    if (!net) { set_last_error("network is NULL"); return MPC_E_NULL; }
    if (net->n_layers < 1 || net->n_layers > MPC_MLP_MAX_LAYERS) { set_last_error("network: 1..4 Linear layers"); return MPC_E_ARG; }
    if (net->activation < MPC_ACT_SIGMOID || net->activation > MPC_ACT_ELU) { set_last_error("network: unknown activation"); return MPC_E_ARG; }
    if (net->widths[0] != ns + nc || net->widths[net->n_layers] != ns) { set_last_error("network: widths[0] must be n_state + n_ctrl, widths[L] n_state"); return MPC_E_DIMS; }
    if (ns > 16) { set_last_error("network kernels: n_state <= 16"); return MPC_E_DIMS; }
    if (mpc_mlp_workspace_bytes(net) > bytes || !workspace || ((uintptr_t)workspace & 15)) {
        set_last_error("network: workspace too small or not 16-byte aligned (see mpc_mlp_workspace_bytes)");
        return MPC_E_ARG;
    }
    PackArgs a;
    a.L = d.L = net->n_layers;
    d.act = net->activation;
    d.pass = net->passthrough ? 1 : 0;
    float *w = (float *)workspace;
    for (int l = 0; l <= net->n_layers; ++l) {
        if (net->widths[l] < 1 || net->widths[l] > 4096) { set_last_error("network: layer width out of range"); return MPC_E_DIMS; }
        a.w[l] = d.w[l] = net->widths[l];
        a.wp[l] = d.wp[l] = pad16(net->widths[l]);
    }
    for (int l = 0; l < net->n_layers; ++l) {
        if (!net->W[l] || !net->b[l]) { set_last_error("network: weight / bias pointer is NULL"); return MPC_E_NULL; }
        a.W[l] = (const float *)net->W[l];
        a.b[l] = (const float *)net->b[l];
        a.Wp[l] = w; d.W[l] = w; w += (size_t)d.wp[l + 1] * d.wp[l];
        a.bp[l] = w; d.b[l] = w; w += d.wp[l + 1];
    }
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(64), dim3(256), 0, st, a);
    return check_launch("mlp_pack_kernel");
}

int max_hidden_pad(const MlpDesc &d)
{
    int h = 16;
    for (int l = 1; l < d.L; ++l) h = d.wp[l] > h ? d.wp[l] : h;
    return h;
}

}  // namespace

int launch_nn_rollout(const StepParams<float> &p, const mpc_mlp_dynamics *net, void *workspace, int64_t bytes, hipStream_t st)
{
    MlpDesc d;
    int rc = mlp_prepare(net, p.ns, p.nc, workspace, bytes, d, st);
    if (rc) return rc;
    const int TS = d.wp[0] + 4, ZS = max_hidden_pad(d) + 4;
    const size_t lds = ((size_t)2 * 16 * TS + (size_t)2 * 16 * ZS) * sizeof(float);
    if (lds > 160 * 1024) { set_last_error("network: layers too wide for the LDS-resident kernel"); return MPC_E_DIMS; }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nn_rollout_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nn_rollout_kernel, dim3((unsigned)((p.B + 15) / 16)), dim3(64), lds, st, p, d, TS, ZS);
    return check_launch("nn_rollout_kernel");
}

int launch_nn_linearize(const mpc_mlp_dynamics *net, long N, int ns, int nc, const float *x, const float *u, float *F,
                        float *f, void *workspace, int64_t bytes, hipStream_t st)
{
    MlpDesc d;
    int rc = mlp_prepare(net, ns, nc, workspace, bytes, d, st);
    if (rc) return rc;
    const int TS = d.wp[0] + 4, ZS = max_hidden_pad(d) + 4, NTJ = d.wp[0] >> 4;
    int GT = 1;
    for (int l = 1; l < d.L; ++l) GT = (d.wp[l] >> 4) * NTJ > GT ? (d.wp[l] >> 4) * NTJ : GT;
    const size_t lds = ((size_t)16 * TS + (size_t)(d.L > 1 ? d.L - 1 : 1) * 16 * ZS) * sizeof(float) + (size_t)2 * GT * 64 * 16;
    if (lds > 160 * 1024) { set_last_error("network: layers too wide for the LDS-resident kernel"); return MPC_E_DIMS; }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&nn_linearize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nn_linearize_kernel, dim3((unsigned)((N + 15) / 16)), dim3(64), lds, st, d, N, ns, nc, x, u, F, f, TS, ZS, GT);
    return check_launch("nn_linearize_kernel");
}

}  // namespace mpclqr
