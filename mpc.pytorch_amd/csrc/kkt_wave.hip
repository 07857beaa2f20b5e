// kkt_wave.hip -- the closed-form part of LQRStepFn.backward (mpc/lqr_step.py:346-404) for fp32 problems
// with n = n_state + n_ctrl <= 64, split where the arithmetic splits:
//
//   kkt_costate_kernel   the two costate recursions (:355-385), sequential in t: one wavefront per problem,
//                        lane i owns state i; C and F are read column-wise ((C tau)[i] = sum_j C[j][i] tau[j]
//                        by symmetry, (Fx' lam)[i] = sum_m F[m][i] lam[m]): conflict-free LDS reads of blocks
//                        that arrive by LDS-DMA one step ahead; tau[j] / lam[m] reach the lanes as readlanes.
//                        lam_{t+1}, dlam_{t+1} are parked in the first 2 n_state words of the dF_t block.
//   kkt_outer_kernel     dC_t, dc_t, dF_t (:346-353, :387-400): independent over (t, b), one wavefront each,
//                        16 bytes per lane, every output byte written exactly once and fully coalesced
//                        (the block first picks its two costates out of its own dF_t block).
//
// The generic kernel (lqr_generic.hip) does the same with one workgroup per problem and LDS-staged blocks;
// at n = 40 it spends 0.86 ms on what is 1.5 GB of compulsory traffic.
#include <string>
#include "lqr_common.h"

namespace mpclqr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lane_bcast(float x, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// HBM -> LDS, 16 bytes per lane, `bytes` contiguous bytes (a multiple of 16) starting at g
__device__ __forceinline__ void dma_block(const float *g, char *lds, int bytes, int lane)
{
    for (int off = 0; off < bytes; off += 1024)
        if (off + 16 * lane < bytes)
            __builtin_amdgcn_global_load_lds((glb_void_t *)((const char *)g + off + 16 * lane),
                                             (lds_void_t *)(lds + off), 16, 0, 0);
}

// ... 4 bytes per lane, for blocks and rows at any (4-byte) alignment and of any length: the general-shape instantiations (round 6)
__device__ __forceinline__ void dma_block4(const float *g, char *lds, int bytes, int lane)
{
    for (int off = 0; off < bytes; off += 256)
        if (off + 4 * lane < bytes)
            __builtin_amdgcn_global_load_lds((glb_void_t *)((const char *)g + off + 4 * lane), (lds_void_t *)(lds + off), 4, 0, 0);
}

// s_waitcnt vmcnt(n) for a run-time n (the immediate has to be a constant): at most n of the newest vector-memory
// operations may still be in flight
__device__ __forceinline__ void wait_newer(int n)
{
#define MPC_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        MPC_W(0) MPC_W(1) MPC_W(2) MPC_W(3) MPC_W(4) MPC_W(5) MPC_W(6) MPC_W(7) MPC_W(8) MPC_W(9) MPC_W(10) MPC_W(11) MPC_W(12)
        MPC_W(13) MPC_W(14) MPC_W(15) MPC_W(16) MPC_W(17) MPC_W(18) MPC_W(19) MPC_W(20) MPC_W(21) MPC_W(22) MPC_W(23) MPC_W(24)
        MPC_W(25) MPC_W(26) MPC_W(27) MPC_W(28) MPC_W(29) MPC_W(30) MPC_W(31) MPC_W(32) MPC_W(33) MPC_W(34) MPC_W(35) MPC_W(36)
        MPC_W(37) MPC_W(38) MPC_W(39) MPC_W(40) MPC_W(41) MPC_W(42) MPC_W(43) MPC_W(44) MPC_W(45) MPC_W(46) MPC_W(47) MPC_W(48)
        MPC_W(49) MPC_W(50) MPC_W(51) MPC_W(52) MPC_W(53) MPC_W(54) MPC_W(55) MPC_W(56) MPC_W(57) MPC_W(58) MPC_W(59) MPC_W(60)
        MPC_W(61) MPC_W(62) MPC_W(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef MPC_W
}

// NSLOT ring slots of [C_t | F_t | record]; record = four rows of 64 floats: tau* | dtau | c_x | dl_dx.  Everything a
// timestep needs arrives by LDS-DMA NSLOT - 1 steps ahead (round 3: one step ahead and `vmcnt(0)` per step left a
// wavefront waiting out a full HBM round trip every timestep -- 3.8 us per step at config 5, 0.244 ms of its backward);
// the wait counts the DMA instructions of the newer stages, every stage issues the same number of them.
// AL: every block and row on the 16-byte grid (n, n_state multiples of 4, aligned bases and strides): 16 bytes a lane.  !AL (round 6): any
// shape and alignment -- the same stages moved 4 bytes a lane (a record row = one instruction: lane l fetches word l), rows of C read
// word by word.  Rounds 3-5 sent such shapes (13/4, 20/5 ...) to the generic kernel: 0.45 of their 1.0 ms backward.
template <int NSLOT, bool AL>
__global__ void __launch_bounds__(64) kkt_costate_kernel(StepParams<float> p, const float *dx, const float *du,
                                                         const float *dl_dx, float *dF, float *df, float *dx_init)
{
    extern __shared__ __attribute__((aligned(16))) char kkt_lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    const int cbytes = n * n * 4, fbytes = ns * n * 4, slot_bytes = cbytes + fbytes + 1024;
    const int nd = AL ? (cbytes + 1023) / 1024 + (T > 1 ? (fbytes + 1023) / 1024 : 0) + 1      // DMA instructions per stage
                      : (cbytes + 255) / 256 + (T > 1 ? (fbytes + 255) / 256 : 0) + 4;
    const bool st = lane < ns;
    const int li = st ? lane : 0;
    // the record's lane -> source map: row = lane / 16, granule g = lane % 16 of that row
    const int row = lane >> 4, g = lane & 15;
    const float *rsrc = nullptr;
    long rstep = 0;
    if (row == 0) {
        if (4 * g < ns) { rsrc = p.cur_x + (long)b * ns + 4 * g; rstep = (long)B * ns; }
        else if (4 * g < n) { rsrc = p.cur_u + (long)b * nc + (4 * g - ns); rstep = (long)B * nc; }
    } else if (row == 1) {
        if (4 * g < ns) { rsrc = dx + (long)b * ns + 4 * g; rstep = (long)B * ns; }
        else if (4 * g < n) { rsrc = du + (long)b * nc + (4 * g - ns); rstep = (long)B * nc; }
    } else if (row == 2) {
        if (4 * g < ns) { rsrc = p.c + (long)b * p.c_sb + 4 * g; rstep = p.c_st; }
    } else if (4 * g < ns) { rsrc = dl_dx + (long)b * ns + 4 * g; rstep = (long)B * ns; }
    // !AL: the record's four rows word by word -- lane l: tau*[l] | dtau[l] | c[l] (states) | dl_dx[l] (states)
    const float *rs4[4] = {nullptr, nullptr, nullptr, nullptr};
    long rst4[4] = {0, 0, 0, 0};
    if (!AL) {
        if (lane < ns) {
            rs4[0] = p.cur_x + (long)b * ns + lane; rst4[0] = (long)B * ns;
            rs4[1] = dx + (long)b * ns + lane; rst4[1] = (long)B * ns;
            rs4[2] = p.c + (long)b * p.c_sb + lane; rst4[2] = p.c_st;
            rs4[3] = dl_dx + (long)b * ns + lane; rst4[3] = (long)B * ns;
        } else if (lane < n) {
            rs4[0] = p.cur_u + (long)b * nc + (lane - ns); rst4[0] = (long)B * nc;
            rs4[1] = du + (long)b * nc + (lane - ns); rst4[1] = (long)B * nc;
        }
    }
    auto issue = [&](int t, int slot) {
        t = t >= 0 ? t : 0;                                   // past the end: stage 0 again, the count per stage stays fixed
        char *base = kkt_lds + slot * slot_bytes;
        if (AL) {
            dma_block(p.C + (long)t * p.C_st + (long)b * p.C_sb, base, cbytes, lane);
            if (T > 1) dma_block(p.F + (long)(t < T - 1 ? t : T - 2) * p.F_st + (long)b * p.F_sb, base + cbytes, fbytes, lane);
            if (rsrc)
                __builtin_amdgcn_global_load_lds((glb_void_t *)(rsrc + (long)t * rstep), (lds_void_t *)(base + cbytes + fbytes), 16, 0, 0);
        } else {
            dma_block4(p.C + (long)t * p.C_st + (long)b * p.C_sb, base, cbytes, lane);
            if (T > 1) dma_block4(p.F + (long)(t < T - 1 ? t : T - 2) * p.F_st + (long)b * p.F_sb, base + cbytes, fbytes, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q)          // (every row has a lane that takes part: n_state >= 1)
                if (rs4[q])
                    __builtin_amdgcn_global_load_lds((glb_void_t *)(rs4[q] + (long)t * rst4[q]), (lds_void_t *)(base + cbytes + fbytes + 256 * q), 4, 0, 0);
        }
    };
    float lam = 0.f, dlam = 0.f;
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) issue(T - 1 - i, i);
    int slot = 0;
    for (int t = T - 1; t >= 0; --t) {
        wait_newer((NSLOT - 2) * nd);
        const float *Cl = (const float *)(kkt_lds + slot * slot_bytes);
        const float *Fl = (const float *)(kkt_lds + slot * slot_bytes + cbytes);
        const float *Rl = (const float *)(kkt_lds + slot * slot_bytes + cbytes + fbytes);
        const float tau = lane < n ? Rl[lane] : 0.f, d = lane < n ? Rl[64 + lane] : 0.f;
        const float cc = st ? Rl[128 + lane] : 0.f, gx = st ? Rl[192 + lane] : 0.f;
        // the slot the PREVIOUS timestep worked on is free (its reads fed that timestep's products): the stage NSLOT - 1
        // steps on goes there
        issue(t - (NSLOT - 1), (slot + NSLOT - 1) % NSLOT);
        const long tb = (long)t * B + b;
        float r1 = cc, r2 = -gx;
        // (C tau)[i], (C dtau)[i] for the state rows: ROW i of C (mpc/lqr_step.py:355-383, bmv(Ct_xx, xt) + bmv(Ct_xu, ut)),
        // 16 bytes a load.  (Rounds 2-3 read column i -- consecutive lanes, consecutive words -- which is the same numbers
        // only for a symmetric C; the reference uses C as given and an asymmetric C is a supported input: ADVICE r03,
        // fixtures grad_asym_cfg5_f32 / grad_asym_20_4_f32.  mpc_lqr_kkt_grads carries no options, so there is no promise to
        // take the column read back under; the vouched route of this shape is the fused backward, lqr_mfma40_body.h.)
        if (AL) {
            for (int j = 0; j < n; j += 4) {
                const f32x4 cr = *(const f32x4 *)(Cl + li * n + j);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    r1 = fmaf(cr[v], lane_bcast(tau, j + v), r1);
                    r2 = fmaf(cr[v], lane_bcast(d, j + v), r2);
                }
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const float cr = Cl[li * n + j];
                r1 = fmaf(cr, lane_bcast(tau, j), r1);
                r2 = fmaf(cr, lane_bcast(d, j), r2);
            }
        }
        if (t < T - 1) {
            if (AL) {
                for (int m = 0; m < ns; m += 4) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float fm = Fl[(m + v) * n + li];
                        r1 = fmaf(fm, lane_bcast(lam, m + v), r1);
                        r2 = fmaf(fm, lane_bcast(dlam, m + v), r2);
                    }
                }
            } else {
                for (int m = 0; m < ns; ++m) {
                    const float fm = Fl[m * n + li];
                    r1 = fmaf(fm, lane_bcast(lam, m), r1);
                    r2 = fmaf(fm, lane_bcast(dlam, m), r2);
                }
            }
        }
        if (t < T - 1 && st) {
            float *park = dF + tb * (long)(ns * n);
            park[lane] = lam;
            park[ns + lane] = dlam;
            if (df) df[tb * ns + lane] = -dlam;                                   // :397-400
        }
        lam = r1;
        dlam = r2;
        slot = (slot + 1) % NSLOT;
    }
    if (st) dx_init[(long)b * ns + lane] = -dlam;                                 // :404
}

// element e = i n + j of -0.5 (a b' + b a') resp. -(a2 b' + a1 b2') for a run of four consecutive e that may cross rows (n not a multiple of 4)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <bool AL>
__global__ void __launch_bounds__(64) kkt_outer_kernel(StepParams<float> p, const float *dx, const float *du, float *dC,
                                                       float *dc, float *dF)
{
    __shared__ __attribute__((aligned(16))) float sv[4][64];       // tau, dtau, lam_{t+1}, dlam_{t+1}
    const long tb = blockIdx.x;
    const int lane = threadIdx.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    const int t = (int)(tb / B);
    const bool have = t < T - 1;
    float tau = 0.f, d = 0.f;
    if (lane < ns) { tau = p.cur_x[tb * ns + lane]; d = dx[tb * ns + lane]; }
    else if (lane < n) { tau = p.cur_u[tb * nc + (lane - ns)]; d = du[tb * nc + (lane - ns)]; }
    sv[0][lane] = tau;
    sv[1][lane] = d;
    float *dFt = dF + tb * (long)(ns * n);
    if (have && lane < ns) {
        sv[2][lane] = dFt[lane];
        sv[3][lane] = dFt[ns + lane];
    }
    __syncthreads();
    if (lane < n) __builtin_nontemporal_store(-d, dc + tb * n + lane);                                         // :352-353
    // dC_t = -0.5 (dtau tau' + tau dtau')   (:346-351), four consecutive columns per lane
    float *dCt = dC + tb * (long)(n * n);
    for (int e = 4 * lane; e < n * n; e += 256) {
        int i = e / n, j = e - i * n;
        if (AL) {
            const float di = sv[1][i], ti = sv[0][i];
            const f32x4 tj = *(const f32x4 *)&sv[0][j], dj = *(const f32x4 *)&sv[1][j];
            f32x4 o;
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = -0.5f * fmaf(di, tj[v], ti * dj[v]);
            __builtin_nontemporal_store(o, (f32x4 *)(dCt + e));          // (write-once gradients: see store_f32x2_out, lqr_dpp16.hip)
        } else {
            // (round 6) any n: the four elements may cross a row; the block starts on a 4-byte boundary only -- 16 bytes a lane all the
            // same (global memory takes unaligned vector stores), the block's last 1-3 elements one by one
            f32x4 o;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ii = i < n ? i : n - 1;
                o[v] = -0.5f * fmaf(sv[1][ii], sv[0][j], sv[0][ii] * sv[1][j]);
                if (++j == n) { j = 0; ++i; }
            }
            if (e + 4 <= n * n) __builtin_nontemporal_store(o, (f32x4_a4 *)(dCt + e));
            else
#pragma unroll
                for (int v = 0; v < 3; ++v)
                    if (e + v < n * n) __builtin_nontemporal_store(o[v], dCt + e + v);
        }
    }
    // dF_t = -(dlam_{t+1} tau' + lam_{t+1} dtau')   (:387-396)
    if (have) {
        for (int e = 4 * lane; e < ns * n; e += 256) {
            int i = e / n, j = e - i * n;
            if (AL) {
                const float li = sv[2][i], dli = sv[3][i];
                const f32x4 tj = *(const f32x4 *)&sv[0][j], dj = *(const f32x4 *)&sv[1][j];
                f32x4 o;
#pragma unroll
                for (int v = 0; v < 4; ++v) o[v] = -fmaf(dli, tj[v], li * dj[v]);
                __builtin_nontemporal_store(o, (f32x4 *)(dFt + e));
            } else {
                f32x4 o;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int ii = i < ns ? i : ns - 1;
                    o[v] = -fmaf(sv[3][ii], sv[0][j], sv[2][ii] * sv[1][j]);
                    if (++j == n) { j = 0; ++i; }
                }
                if (e + 4 <= ns * n) __builtin_nontemporal_store(o, (f32x4_a4 *)(dFt + e));
                else
#pragma unroll
                    for (int v = 0; v < 3; ++v)
                        if (e + v < ns * n) __builtin_nontemporal_store(o[v], dFt + e + v);
            }
        }
    }
}

// util.get_traj (LinDx, mpc/util.py:114-126) for 16 < n <= 64: one wavefront per problem, lane i sums row i with tau[j] as
// readlane scalars.  F_t and the step's small vector (lane < ns: f_t[lane], else u_t[lane - ns]: one 4-byte LDS-DMA per lane)
// arrive NSLOT - 1 timesteps ahead with counted waits (round 3: one step ahead behind `vmcnt(0)`, and u_t / f_t by vector
// loads whose results the compiler drained the DMA queue for, left the wavefront waiting out a memory round trip per timestep:
// 118 us for the 335 MB of config 5's F).
template <int NSLOT>
__global__ void __launch_bounds__(64) traj_wave_kernel(StepParams<float> p, float *x)
{
    extern __shared__ __attribute__((aligned(16))) char traj_lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int ns = p.ns, nc = p.nc, n = ns + nc, T = p.T, B = p.B;
    const int fbytes = ns * n * 4, slot_bytes = fbytes + 256;
    const int nd = (fbytes + 1023) / 1024 + 1;                   // DMA instructions per stage
    const bool st = lane < ns;
    float xi = st ? p.x_init[(long)b * ns + lane] : 0.f;
    if (st) x[(long)b * ns + lane] = xi;
    if (T < 2) return;
    // the record's source: f_t[lane] for the state lanes (x_init again when there is no f: never used), u_t[lane - ns] above
    const float *rsrc;
    long rstep;
    if (st) {
        rsrc = p.f ? p.f + (long)b * p.f_sb + lane : p.x_init + (long)b * ns + lane;
        rstep = p.f ? p.f_st : 0;
    } else {
        rsrc = p.cur_u + (long)b * nc + (lane < n ? lane - ns : 0);
        rstep = (long)B * nc;
    }
    auto issue = [&](int t, int slot) {
        t = t < T - 1 ? t : T - 2;                               // past the end: the last stage again, the count per stage stays fixed
        char *base = traj_lds + slot * slot_bytes;
        dma_block(p.F + (long)t * p.F_st + (long)b * p.F_sb, base, fbytes, lane);
        __builtin_amdgcn_global_load_lds((glb_void_t *)(rsrc + (long)t * rstep), (lds_void_t *)(base + fbytes), 4, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) issue(i, i);
    int slot = 0;
    for (int t = 0; t < T - 1; ++t) {
        wait_newer((NSLOT - 2) * nd);
        const float *Fl = (const float *)(traj_lds + slot * slot_bytes) + (st ? lane : 0) * n;
        const float rec = ((const float *)(traj_lds + slot * slot_bytes + fbytes))[lane];
        const float tau = st ? xi : (lane < n ? rec : 0.f);
        float acc = (p.f && st) ? rec : 0.f;
        // (the slot the previous timestep worked on is free: its reads fed that timestep's sums)
        issue(t + NSLOT - 1, (slot + NSLOT - 1) % NSLOT);
        for (int j = 0; j < n; j += 4) {
            const f32x4 row = *(const f32x4 *)(Fl + j);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc = fmaf(row[v], lane_bcast(tau, j + v), acc);
        }
        xi = acc;
        if (st) x[((long)(t + 1) * B + b) * ns + lane] = xi;
        slot = (slot + 1) % NSLOT;
    }
}

}  // namespace

bool traj_wave_supported(const StepParams<float> &p)
{
    const int n = p.ns + p.nc;
    return n > 16 && n <= 64 && n % 4 == 0 && p.B > 0 && !p.env.kind &&
           (p.T == 1 || (((uintptr_t)p.F & 15) == 0 && p.F_st % 4 == 0 && p.F_sb % 4 == 0));
}

int launch_traj_wave(const StepParams<float> &p, float *x, hipStream_t st)
{
    const int n = p.ns + p.nc;
    // four slots (the DMA three timesteps ahead) while the whole batch is resident with them (256 CUs x 160 KiB), else two: a
    // batch that runs in rounds anyway hides the latency with more wavefronts per CU (config 5: 81 against 110 us at B = 1024,
    // 557 against 519 at B = 8192)
    const size_t slot = (size_t)p.ns * n * 4 + 256;
    const bool deep = (size_t)((p.B + 255) / 256) * 4 * slot <= 160 * 1024;
    const size_t lds = (deep ? 4 : 2) * slot;
    if (deep) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&traj_wave_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(traj_wave_kernel<4>, dim3(p.B), dim3(64), lds, st, p, x);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&traj_wave_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(traj_wave_kernel<2>, dim3(p.B), dim3(64), lds, st, p, x);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("traj_wave_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

// every block and row on the 16-byte grid: the AL instantiations
static bool kkt_wave_aligned(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx, const float *dC,
                             const float *dF)
{
    const int n = p.ns + p.nc;
    auto al = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
    // (the small vectors travel as 16-byte granules too: rows of x*, u*, dx, du, dl_dx and c start on 16-byte boundaries)
    return n <= 64 && n % 4 == 0 && p.ns % 4 == 0 && p.ns * n >= 2 * p.ns && p.B > 0 && al(p.C) && (p.T == 1 || al(p.F)) &&
           p.C_st % 4 == 0 && p.C_sb % 4 == 0 && p.F_st % 4 == 0 && p.F_sb % 4 == 0 &&
           al(p.c) && p.c_st % 4 == 0 && p.c_sb % 4 == 0 && al(p.cur_x) && al(p.cur_u) && al(dx) && al(du) && al(dl_dx) &&
           ((uintptr_t)dC & 15) == 0 && (p.T == 1 || ((uintptr_t)dF & 15) == 0);
}

// (round 6) any float32 shape with n <= 64: what is not on the 16-byte grid takes the general instantiations (4 bytes a lane)
bool kkt_wave_supported(const StepParams<float> &p, const float *, const float *, const float *, const float *, const float *)
{
    const int n = p.ns + p.nc;
    return n <= 64 && n >= 2 && p.ns >= 1 && p.nc >= 1 && p.B > 0;          // (n >= 2: the costates park in the first 2 n_state words of dF_t)
}

template <int NSLOT, bool AL>
static void launch_costate(const StepParams<float> &p, size_t lds, const float *dx, const float *du, const float *dl_dx, float *dF, float *df,
                           float *dx_init, hipStream_t st)
{
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kkt_costate_kernel<NSLOT, AL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kkt_costate_kernel<NSLOT, AL>), dim3(p.B), dim3(64), lds, st, p, dx, du, dl_dx, dF, df, dx_init);
}

int launch_kkt_wave(const StepParams<float> &p, const float *dx, const float *du, const float *dl_dx, float *dC,
                    float *dc, float *dF, float *df, float *dx_init, hipStream_t st)
{
    const int n = p.ns + p.nc;
    const bool al = kkt_wave_aligned(p, dx, du, dl_dx, dC, dF);
    // three slots (the DMA two timesteps ahead) while four wavefronts of them fit a CU's 160 KiB, else two
    // (the general instantiation: a stage is up to 4x the instructions -- three slots only while two stages stay under vmcnt's 63)
    const size_t slot = (size_t)(n * n + p.ns * n) * 4 + 1024;
    const int nd4 = (n * n * 4 + 255) / 256 + (p.T > 1 ? (p.ns * n * 4 + 255) / 256 : 0) + 4;
    const bool deep = 3 * slot * 4 <= 160 * 1024 && (al || nd4 <= 63);
    const size_t lds = (deep ? 3 : 2) * slot;
    if (deep) {
        if (al) launch_costate<3, true>(p, lds, dx, du, dl_dx, dF, df, dx_init, st);
        else launch_costate<3, false>(p, lds, dx, du, dl_dx, dF, df, dx_init, st);
    } else {
        if (al) launch_costate<2, true>(p, lds, dx, du, dl_dx, dF, df, dx_init, st);
        else launch_costate<2, false>(p, lds, dx, du, dl_dx, dF, df, dx_init, st);
    }
    if (al) hipLaunchKernelGGL(kkt_outer_kernel<true>, dim3((unsigned)((long)p.T * p.B)), dim3(64), 0, st, p, dx, du, dC, dc, dF);
    else hipLaunchKernelGGL(kkt_outer_kernel<false>, dim3((unsigned)((long)p.T * p.B)), dim3(64), 0, st, p, dx, du, dC, dc, dF);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("kkt_wave kernels: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

// the outer products alone: lambda_{t+1}, dlambda_{t+1} already sit in the dF_t blocks (the fused backward of the 32/8
// shape, lqr_mfma40.hip, parks them there itself)
int launch_kkt_outer(const StepParams<float> &p, const float *dx, const float *du, float *dC, float *dc, float *dF, hipStream_t st)
{
    // (the padded fused backward of the 32/8 kernel comes here with any shape: rows of four on the 16-byte grid, or the general form)
    const int n = p.ns + p.nc;
    const bool al = n % 4 == 0 && p.ns % 4 == 0 && ((((uintptr_t)dC | (uintptr_t)dF) & 15) == 0);
    if (al) hipLaunchKernelGGL(kkt_outer_kernel<true>, dim3((unsigned)((long)p.T * p.B)), dim3(64), 0, st, p, dx, du, dC, dc, dF);
    else hipLaunchKernelGGL(kkt_outer_kernel<false>, dim3((unsigned)((long)p.T * p.B)), dim3(64), 0, st, p, dx, du, dC, dc, dF);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error((std::string("kkt_outer_kernel: ") + hipGetErrorString(e)).c_str());
        return MPC_E_LAUNCH;
    }
    return MPC_OK;
}

}  // namespace mpclqr
