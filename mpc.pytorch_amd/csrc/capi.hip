// capi.hip -- extern "C" entry points of libmpc_lqr_hip.so (declared in include/mpc_lqr.h).
// Argument validation + dtype / kernel dispatch; no computation happens on the host.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lqr_common.h"

namespace mpclqr {
static thread_local std::string g_last_error;
void set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }

static int fail(int code, const char *msg)
{
    set_last_error(msg);
    return code;
}

// Read once per process (the first launch): the A/B switch of the 12/4 kernel's staging ring and the number of SIMDs of
// the device the library runs on (MI355X: 256 CUs x 4), which is where the constrained step changes rings.
struct DeviceFacts {
    int ring_force;      // 0 = pick by mode and batch, 2 / 4 = MPC_DPP16_RING
    int simds;
    int ring40_force;    // 0 = pick by mode and batch, 2 / 3 = MPC_MFMA40_RING
    bool sticky;         // false with MPC_DPP16_RING_DYNAMIC set: the switch is re-read at every launch (the tests flip it)
};
static const DeviceFacts &device_facts()
{
    static DeviceFacts f = [] {
        DeviceFacts d;
        d.ring_force = 0;
        d.ring40_force = 0;
        d.simds = 1024;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            d.simds = prop.multiProcessorCount * 4;
        const char *force = getenv("MPC_DPP16_RING");            // "2" / "4": A/B switch, read at load
        if (force && (force[0] == '2' || force[0] == '4')) d.ring_force = force[0] - '0';
        const char *force40 = getenv("MPC_MFMA40_RING");         // "2" / "3": the 32/8 kernel's sweep ring, read at load
        if (force40 && (force40[0] == '2' || force40[0] == '3')) d.ring40_force = force40[0] - '0';
        d.sticky = getenv("MPC_DPP16_RING_DYNAMIC") == nullptr;
        return d;
    }();
    return f;
}
static int dpp16_ring_force()
{
    const DeviceFacts &f = device_facts();
    if (f.sticky) return f.ring_force;
    // MPC_DPP16_RING_DYNAMIC=1 (the test suite): the switch follows the environment from launch to launch
    const char *force = getenv("MPC_DPP16_RING");
    return (force && (force[0] == '2' || force[0] == '4')) ? force[0] - '0' : 0;
}

// The 32/8 kernel's sweep ring: three slots (the DMA two timesteps ahead: address-translation misses behind another kernel
// stay hidden, lqr_mfma40_body.h) wherever four wavefronts per CU are all the batch asks for or the mode gains from them anyway
// (unconstrained / masked: 2.33 against 2.40 ms at B = 8192); the box-constrained step of a larger batch wants six per CU on
// the two-slot ring (round 3: 4.65 against 5.06 ms; round 4, after the QP's rewrite: 3.05 against 3.11).  MPC_MFMA40_RING=2|3 forces one (read like MPC_DPP16_RING).
static int mfma40_ring(const StepParams<float> &sp)
{
    const DeviceFacts &f = device_facts();
    int force = f.ring40_force;
    if (!f.sticky) {                       // MPC_DPP16_RING_DYNAMIC=1 (the test suite): the switch follows the environment
        const char *e = getenv("MPC_MFMA40_RING");
        force = (e && (e[0] == '2' || e[0] == '3')) ? e[0] - '0' : 0;
    }
    if (force) return force;
    // Box-constrained: the three-slot ring while one wave per SIMD holds the batch, six waves per CU on the two-slot ring beyond.
    // Unconstrained / masked (round 4, final kernels, config 5 by HIP events): the two-slot ring keeps 1,536 waves resident, so a
    // batch between 1,025 and 1,536 runs in ONE round on it (B = 1,536: 0.44 against 0.54 ms); beyond, the three-slot ring's
    // shorter rounds win (2,048: 0.57 against 0.72 ms; 4,096: 1.11 against 1.21; 8,192: 2.22 against 2.30).  Up to 1,024 the two-slot
    // ring is 1.4-3 % faster in the sustained state (277 against 281 us) but 30 % SLOWER behind a kernel that emptied the address
    // translations (bench.py's cold_translations row: 0.436 against 0.33 ms) -- what the third slot is there for: it stays.
    if (sp.bound_mode != MPC_BOUND_NONE) return sp.B > f.simds ? 2 : 3;
    return (sp.B > f.simds && sp.B <= f.simds + f.simds / 2) ? 2 : 3;
}

static int check_problem(const mpc_lqr_problem *p, bool need_cost, bool need_nominal, bool need_F = true)
{
    if (!p) return fail(MPC_E_NULL, "problem is NULL");
    if (p->B < 0 || p->T < 1 || p->ns < 1 || p->nc < 1) return fail(MPC_E_DIMS, "need B>=0, T>=1, ns>=1, nc>=1");
    if (p->nc > 64) return fail(MPC_E_DIMS, "n_ctrl > 64 is not supported");
    if (p->dtype != MPC_F32 && p->dtype != MPC_F64) return fail(MPC_E_DTYPE, "dtype must be MPC_F32 or MPC_F64");
    if (p->B == 0) return MPC_OK;
    if (need_cost && (!p->C || !p->c)) return fail(MPC_E_NULL, "C / c is NULL");
    if (need_F && p->T > 1 && !p->F) return fail(MPC_E_NULL, "F is NULL");
    if (!p->x_init) return fail(MPC_E_NULL, "x_init is NULL");
    if (need_nominal && (!p->cur_x || !p->cur_u)) return fail(MPC_E_NULL, "current_x / current_u is NULL");
    return MPC_OK;
}

static int check_env(const mpc_env_dynamics *e, int ns, int nc)
{
    if (e->kind < MPC_ENV_PENDULUM || e->kind > MPC_ENV_CARTPOLE) return fail(MPC_E_ARG, "unknown simulator kind");
    if (!e->params) return fail(MPC_E_NULL, "simulator params is NULL");
    if (nc != 1 || ns != env_ns(e->kind)) return fail(MPC_E_DIMS, "n_state / n_ctrl do not match the simulator");
    if (!(e->dt > 0) || !(e->u_max >= 0)) return fail(MPC_E_ARG, "simulator dt / u_max");
    return MPC_OK;
}

static int check_options(const mpc_lqr_problem *p, const mpc_lqr_options *o)
{
    if (!o) return MPC_OK;
    if (o->bound_mode < MPC_BOUND_NONE || o->bound_mode > MPC_BOUND_TENSOR) return fail(MPC_E_ARG, "bad bound_mode");
    if (o->bound_mode == MPC_BOUND_TENSOR && (!o->lo || !o->hi) && p->B > 0) return fail(MPC_E_NULL, "tensor bounds are NULL");
    if (o->max_linesearch_iter < 1) return fail(MPC_E_ARG, "max_linesearch_iter must be > 0");  // mpc/mpc.py:148
    // mpc/lqr_step.py:195: delta_u without bounds is unimplemented in the reference as well
    if (o->delta_u == o->delta_u && o->delta_u >= 0 && o->bound_mode == MPC_BOUND_NONE)
        return fail(MPC_E_ARG, "delta_u requires u_lower/u_upper (mpc/lqr_step.py:195)");
    if (o->true_dynamics) {
        int rc = check_env(o->true_dynamics, p->ns, p->nc);
        if (rc) return rc;
    }
    return MPC_OK;
}

// floats of the padded 32/8 instantiation's workspace: K [T,B,8,32] | k [T,B,8] | (M, Quu, m) [T,B,328] | second trial [T,B,40]
static int64_t pad_workspace_bytes(int T, int B) { return (int64_t)T * B * (256 + 8 + 328 + 40) * 4; }

// where a caller's missing status array lives inside the workspace: behind the larger of the two uses of it
static int64_t status_scratch_offset(const mpc_lqr_problem *p)
{
    const int64_t e = p->dtype == MPC_F64 ? 8 : 4;
    int64_t generic = ((int64_t)p->T * p->B * p->nc * p->ns + (int64_t)p->T * p->B * p->nc) * e;
    // the 32/8 kernel's constrained modes park (M, Quu, m) behind the gains for the rollout that prices without C
    if (p->dtype == MPC_F32 && p->ns == 32 && p->nc == 8) generic += (int64_t)p->T * p->B * (328 + 40) * 4;   // + the second trial's trajectory
    // its padded instantiation (any n_state <= 32, n_ctrl <= 8): the kernel's own padded gains [T,B,8,32] | [T,B,8] + the same records
    else if (p->dtype == MPC_F32 && p->ns <= 32 && p->nc <= 8) {
        const int64_t pad = pad_workspace_bytes(p->T, p->B);
        generic = generic > pad ? generic : pad;
    }
    const int64_t fused = (int64_t)p->T * p->B * (128 + 16) * 4;     // gain records of the fused kernels + the second
                                                                       // line-search trial's trajectory (box-constrained 12/4 kernel)
    return ((generic > fused ? generic : fused) + 15) & ~(int64_t)15;
}

// After a fused kernel under impl = 0 (auto): the problems it flagged MPC_ST_C_ASYMMETRIC are solved again by the generic
// kernels, which use C exactly as the reference does (mpc/lqr_step.py:68, 294); everything else is left as it is.  A
// compact grid reads the flags -- no host round trip, nothing to synchronise on.
template <typename real>
static int resolve_asymmetric(StepParams<real> sp, int impl, void *workspace, int64_t needK, hipStream_t st)
{
    if (impl != 0 || sp.c_symmetric || !sp.status) return MPC_OK;
    if (sp.env.kind && sp.env.linearize) return MPC_OK;      // no F, f arrays to fall back on: the bit is the answer
    sp.gate = sp.status;
    sp.Kk = nullptr;
    if (!sp.K || !sp.k) {
        sp.K = (real *)workspace;
        sp.k = (real *)((char *)workspace + needK);
    }
    return launch_step_generic<real>(sp, sp.sweep_only ? 1 : 3, st);
}

// may this call take the padded 32/8 instantiation?  (shape and options: mfma40_pad_supported; its workspace: the padded gains
// and the constrained modes' records, 16-byte aligned)
static bool pad_route(const StepParams<float> &sp, const void *workspace, int64_t workspace_bytes)
{
    return mfma40_pad_supported(sp) && workspace && ((uintptr_t)workspace % 16 == 0) && workspace_bytes >= pad_workspace_bytes(sp.T, sp.B);
}
static bool pad_route(const StepParams<double> &, const void *, int64_t) { return false; }

template <typename real>
static int step_impl(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                     void *workspace, int64_t workspace_bytes, int impl, int phase_mask,
                     const void *old_costs_in, hipStream_t st)
{
    StepParams<real> sp = make_params<real>(p, o, out);
    sp.old_costs_in = (const real *)old_costs_in;
    // mpc_lqr_options.qp_start is a hint: an array the staging DMAs cannot fetch (16-byte granules) is ignored, not an error
    if (sp.qp_start && (sp.bound_mode == MPC_BOUND_NONE || (uintptr_t)sp.qp_start % 16 || sp.qp_start_st % 4 || sp.qp_start_sb % 4 ||
                        sp.qp_start_st < 0 || sp.qp_start_sb < 0 || sp.qp_start_st * 4 >= (1ll << 32) || sizeof(real) != 4))
        sp.qp_start = nullptr;
    // the symmetry verdict travels in the status words: a caller that passes none gets them parked behind the workspace
    if (!sp.status && phase_mask == 3 && workspace) {
        const int64_t off = status_scratch_offset(p);
        if (workspace_bytes >= off + (int64_t)p->B * 4) sp.status = (int *)((char *)workspace + off);
    }
    if ((phase_mask & 2) && !sp.sweep_only) {
        if (!sp.new_x || !sp.new_u) return fail(MPC_E_NULL, "new_x / new_u is NULL");
    }
    const int64_t needK = (int64_t)p->T * p->B * p->nc * p->ns * (int64_t)sizeof(real);
    const int64_t needk = (int64_t)p->T * p->B * p->nc * (int64_t)sizeof(real);
    if (sp.sweep_only) {
        if (phase_mask != 3) return fail(MPC_E_ARG, "MPC_OPT_SWEEP_ONLY is an option of mpc_lqr_step");
        if (!sp.K || !sp.k) return fail(MPC_E_NULL, "MPC_OPT_SWEEP_ONLY needs out->K and out->k");
        if (sp.env.kind) return fail(MPC_E_ARG, "MPC_OPT_SWEEP_ONLY with a simulator as true_dynamics");
        // the kernels that can stop after their sweep: the 12/4 and 32/8 fused ones; every other shape takes the generic sweep
        bool fused = false;
        if constexpr (sizeof(real) == 4) fused = (impl == 0 || impl == 3) ? dpp16_supported(sp) : false;
        // (round 6: the padded 12/4 instantiation stops after its sweep like the exact kernel -- where the call reaches it: under impl 0
        // the one-control shapes up to six states belong to the lane-per-problem kernels in front of it, which cannot)
        if constexpr (sizeof(real) == 4)
            fused = fused || ((impl == 8 || (impl == 0 && !tiny_supported(p->ns, p->nc))) && dpp16_pad_supported(sp));
        if constexpr (sizeof(real) == 4) fused = fused || ((impl == 0 || impl == 5) && mfma40_supported(sp));
        // (the padded 32/8 instantiation can stop after its sweep too -- for the shapes that reach it: under impl 0 the 12/4-class
        // shapes belong to kernels in front of it that cannot, and keep the generic sweep they had)
        if constexpr (sizeof(real) == 4)
            fused = fused || (pad_route(sp, workspace, workspace_bytes) &&
                              (impl == 7 || (impl == 0 && !mfma16_supported(sp) && !tiny_supported(p->ns, p->nc))));
        if (!fused) {
            if (impl != 0 && impl != 1) return fail(MPC_E_ARG, "MPC_OPT_SWEEP_ONLY: this kernel cannot stop after its sweep");
            return launch_step_generic<real>(sp, 1, st);
        }
    }
    if (sp.env.kind && sp.env.linearize && !(phase_mask == 3 && tiny_supported(p->ns, p->nc) && (impl == 0 || impl == 4 || impl == 6)))
        return fail(MPC_E_ARG, "in-kernel linearisation needs the lane-per-problem kernel (n_ctrl = 1, n_state <= 6)");
    if (sp.env.kind && (impl == 2 || impl == 3 || impl == 8))
        return fail(MPC_E_ARG, "a simulator as true_dynamics runs on the generic kernels only");
    if ((impl == 4 || impl == 6) && (phase_mask != 3 || !tiny_supported(p->ns, p->nc)))
        return fail(MPC_E_DIMS, "lane-per-problem / wavefront-per-problem kernel needs n_ctrl = 1, n_state <= 6");
    if (impl == 6) {
        bool ok = false;
        if constexpr (sizeof(real) == 4) ok = wave1_supported(sp);
        if (!ok) return fail(MPC_E_DIMS, "the row-per-problem kernel is float32, max_linesearch_iter <= 16, four problems in 150 KiB of LDS");
    }
    if (phase_mask == 3 && (impl == 4 || impl == 6 || (impl == 0 && tiny_supported(p->ns, p->nc)))) {
        // one lane per problem; gains parked in the workspace as [T][ns+1][B], behind them one trajectory column
        // [T][ns+1] per line-search trial lane (<= 8 per problem): 9 (ns + 1) reals per problem-step <= 504 bytes
        if (!workspace || workspace_bytes < (needK + needk) * 9)
            return fail(MPC_E_ARG, "workspace too small (see mpc_lqr_workspace_bytes)");
        sp.Kk = (real *)workspace;
        // (these kernels keep Q and V as general matrices, like the reference: nothing to test, nothing to re-solve)
        if constexpr (sizeof(real) == 4) {
            // A lane per problem leaves the chip empty at the batch sizes iLQR is run at (B = 1024: 16 .. 128 wavefronts
            // on 1024 SIMDs) and its time is one lane's instruction count times the horizon; a 16-lane row per problem
            // does everything that is independent over t for 16 timesteps at once and every line-search trial at once.
            // Worth its sixteen lanes while all its wavefronts are resident together, at most two to a SIMD (measured,
            // profiles/r03_experiments.md section 6: pendulum up to B = 8192, cart-pole -- 39.6 KB of LDS per wavefront --
            // up to 4096; twice that and the lane-per-problem kernel is level or ahead).
            bool wide = impl == 6;
            if (impl == 0 && wave1_supported(sp)) {
                const long per_cu = (160L * 1024) / wave1_lds_bytes(sp);
                wide = (p->B + 3) / 4 <= (device_facts().simds / 4) * (per_cu < 8 ? per_cu : 8);
            }
            if (wide) return launch_step_wave1(sp, st);
        }
        return launch_step_tiny<real>(sp, st);
    }
    if (phase_mask == 3 && impl != 1 && !sp.env.kind) {
        if constexpr (sizeof(real) == 4) {
            // the fused kernels park their gains [T,B,64] in the workspace; out->K / out->k are optional
            const int64_t need = (int64_t)p->T * p->B * (128 + 16) * (int64_t)sizeof(float);   // gain record + (m, M) record + trial trajectory
            const bool have_ws = workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16 == 0);
            if ((sp.K == nullptr) != (sp.k == nullptr)) return fail(MPC_E_NULL, "pass both K and k, or neither");
            sp.Kk = (float *)workspace;
            const bool dpp = dpp16_supported(sp), mfma = mfma16_supported(sp), dpad = dpp16_pad_supported(sp);
            if (impl == 3 && !dpp)
                return fail(MPC_E_DIMS, "DPP kernel needs fp32, n_state = 12, n_ctrl = 4 and 16-byte aligned blocks");
            if (impl == 8 && !dpad)
                return fail(MPC_E_DIMS, "padded DPP kernel needs fp32, n_state <= 12, n_ctrl <= 4, max_linesearch_iter <= 16");
            if (impl == 2 && !mfma)
                return fail(MPC_E_DIMS, "fused MFMA kernel needs fp32, n_state <= 12, n_ctrl <= 4, max_linesearch_iter <= 16");
            if ((impl == 3 || impl == 2 || impl == 8 || dpp || mfma || dpad) && !have_ws)
                return fail(MPC_E_ARG, "workspace too small or not 16-byte aligned (see mpc_lqr_workspace_bytes)");
            if (impl == 3 || (impl == 0 && dpp)) {
                // ring depth (lqr_dpp16.hip): the unconstrained step always on the short ring; the constrained one there
                // only when the batch has more waves (4 problems each) than the chip has SIMDs (256 CUs x 4)
                const bool constrained = sp.bound_mode != MPC_BOUND_NONE || sp.zero_mask != nullptr;
                const int force = dpp16_ring_force();
                const bool ring2 = force ? force == 2 : (!constrained || (sp.B + 3) / 4 > device_facts().simds);
                const int rc = ring2 ? launch_step_dpp16_ring2(sp, st) : launch_step_dpp16(sp, st);
                return rc ? rc : resolve_asymmetric(sp, impl, workspace, needK, st);
            }
            // (round 6) every other float32 shape up to 12/4 -- and 12/4 itself where a block is not 16-byte aligned --: the 12/4
            // kernel's PADDED instantiation, ahead of the one-problem-per-wavefront kernel that served them in rounds 1-5
            if (impl == 8 || (impl == 0 && dpad)) {
                const int rc = launch_step_dpp16_pad(sp, st);
                return rc ? rc : resolve_asymmetric(sp, impl, workspace, needK, st);
            }
            if (impl == 2 || (impl == 0 && mfma)) {
                const int rc = launch_step_mfma16(sp, st);
                return rc ? rc : resolve_asymmetric(sp, impl, workspace, needK, st);
            }
            sp.Kk = nullptr;
        } else {
            // float64 (round 5): n_state <= 12, n_ctrl <= 4 on the one-problem-per-wavefront kernel's float64 instantiation
            // (v_mfma_f64_16x16x4_f64); what every test and gradient check of the reference runs in (tests/test_mpc.py .double())
            if (impl == 3 || impl == 8) return fail(MPC_E_DTYPE, "the 4-problems-per-wave kernel is fp32 only");
            const bool mfma = mfma16_supported(sp);
            if (impl == 2 && !mfma)
                return fail(MPC_E_DIMS, "fused MFMA kernel needs n_state <= 12, n_ctrl <= 4, max_linesearch_iter <= 16");
            const int64_t need = (int64_t)p->T * p->B * 64 * (int64_t)sizeof(double);
            const bool have_ws = workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16 == 0);
            // (a caller written against ABI 7 -- out->K / out->k given, no or a small workspace -- keeps the generic
            // kernel under impl 0; only a FORCED impl 2 insists on the fused kernel's workspace)
            if (impl == 2 && !have_ws)
                return fail(MPC_E_ARG, "workspace too small or not 16-byte aligned (see mpc_lqr_workspace_bytes)");
            if (mfma && have_ws) {
                if ((sp.K == nullptr) != (sp.k == nullptr)) return fail(MPC_E_NULL, "pass both K and k, or neither");
                sp.Kk = (real *)workspace;
                const int rc = launch_step_mfma16(sp, st);
                return rc ? rc : resolve_asymmetric(sp, impl, workspace, needK, st);
            }
        }
    }
    if constexpr (sizeof(real) == 4) {
        // Every float32 shape up to 32/8 that has no kernel of its own (13/4, 20/5, 24/8, 32/4, 8/6 ...): the 32/8 kernel's
        // PADDED instantiation (lqr_mfma40_body.h, PADK) -- the reference's sweep is shape-agnostic (mpc/lqr_step.py:61-158), and
        // between the hand-tuned shapes the generic kernel ran at 4 % of the roofline (missing 1)
        if (impl == 7 && !(phase_mask == 3 && pad_route(sp, workspace, workspace_bytes)))
            return fail(MPC_E_DIMS, "padded MFMA kernel needs fp32, n_state <= 32, n_ctrl <= 8, max_linesearch_iter <= 16, no simulator, and the "
                                    "workspace of mpc_lqr_workspace_bytes (16-byte aligned)");
        if (phase_mask == 3 && (impl == 7 || (impl == 0 && !mfma40_supported(sp))) && pad_route(sp, workspace, workspace_bytes)) {
            if ((sp.K == nullptr) != (sp.k == nullptr)) return fail(MPC_E_NULL, "pass both K and k, or neither");
            StepParams<float> q = sp;
            const int64_t TB = (int64_t)p->T * p->B;
            q.K_user = sp.K;
            q.k_user = sp.k;
            q.K = (float *)workspace;
            q.k = q.K + TB * 256;
            q.Kk = q.k + TB * 8;
            const int rc = mfma40_pad16_supported(q) ? launch_step_mfma40_pad16(q, st) : launch_step_mfma40_pad4(q, st);
            return rc ? rc : resolve_asymmetric(sp, impl, workspace, needK, st);
        }
    }
    if (!sp.K || !sp.k) {
        if (phase_mask != 3) return fail(MPC_E_NULL, "K / k is NULL");
        if (!workspace || workspace_bytes < needK + needk)
            return fail(MPC_E_ARG, "workspace too small (see mpc_lqr_workspace_bytes)");
        sp.K = (real *)workspace;
        sp.k = (real *)((char *)workspace + needK);
    }
    if constexpr (sizeof(real) == 4) {
        // n_state = 32, n_ctrl = 8, unconstrained: register-resident MFMA sweep and rollout (lqr_mfma40_body.h)
        if (impl == 5 && !(phase_mask == 3 && mfma40_supported(sp)))
            return fail(MPC_E_DIMS, "MFMA kernel needs fp32, n_state = 32, n_ctrl = 8, 16-byte aligned blocks, no simulator");
        if (phase_mask == 3 && (impl == 5 || impl == 0) && mfma40_supported(sp)) {
            // (M, Quu, m) record of the constrained modes: behind K | k where the workspace has the room (rollout_priced)
            sp.Kk = nullptr;
            if ((sp.bound_mode != MPC_BOUND_NONE || sp.zero_mask) && workspace && ((uintptr_t)workspace % 16 == 0) &&
                workspace_bytes >= needK + needk + (int64_t)p->T * p->B * (328 + 40) * 4)
                sp.Kk = (real *)((char *)workspace + needK + needk);
            const int rc = mfma40_ring(sp) == 2 ? launch_step_mfma40_ring2(sp, st) : launch_step_mfma40(sp, st);
            return rc ? rc : resolve_asymmetric(sp, impl, workspace, needK, st);
        }
    } else if (impl == 5) {
        return fail(MPC_E_DTYPE, "the MFMA sweep is fp32 only");
    }
    return launch_step_generic<real>(sp, phase_mask, st);
}
}  // namespace mpclqr

using namespace mpclqr;

extern "C" {

int mpc_lqr_abi_version(void) { return MPC_LQR_ABI_VERSION; }

const char *mpc_lqr_build_info(void)
{
    return "libmpc_lqr_hip gfx950 (CDNA4) | kernels: lqr_step_generic<f32,f64>, lqr_step_mfma16<f32,f64>, lqr_step_dpp16<f32>, lqr_step_dpp16_padded<f32>, "
           "lqr_step_tiny<f32,f64>, lqr_step_wave1<f32>, lqr_step_mfma40<f32>, lqr_step_mfma40_padded<f32>, nn_rollout<f32>, nn_linearize<f32>, env_linearize, kkt_grads, kkt_fused<f32>, kkt_fused_padded<f32>, pnqp, traj_cost, select_best | built " __DATE__ " " __TIME__;
}

const char *mpc_lqr_last_error(void) { return g_last_error.c_str(); }

int64_t mpc_lqr_workspace_bytes(const mpc_lqr_problem *p)
{
    if (!p) return 0;
    // + one status word per problem for callers that pass out->status = NULL (see step_impl)
    return status_scratch_offset(p) + (int64_t)p->B * 4 + 256;
}

int mpc_lqr_step(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                 void *workspace, int64_t workspace_bytes, int impl, void *stream)
{
    const bool inline_lin = o && o->true_dynamics && o->true_dynamics->linearize;
    int rc = check_problem(p, true, true, !inline_lin);
    if (rc) return rc;
    if ((rc = check_options(p, o))) return rc;
    if (!out) return fail(MPC_E_NULL, "outputs is NULL");
    if (p->B == 0) return MPC_OK;
    hipStream_t st = (hipStream_t)stream;
    return p->dtype == MPC_F32 ? step_impl<float>(p, o, out, workspace, workspace_bytes, impl, 3, nullptr, st)
                               : step_impl<double>(p, o, out, workspace, workspace_bytes, impl, 3, nullptr, st);
}

int mpc_lqr_impl_supported(const mpc_lqr_problem *p, const mpc_lqr_options *o, int impl)
{
    if (!p || check_problem(p, false, false) != MPC_OK || check_options(p, o) != MPC_OK) return 0;
    if (impl == 1) return generic_lds_bytes(p->ns, p->nc, p->dtype == MPC_F64 ? 8 : 4) <= 160 * 1024;
    if (impl == 4) return tiny_supported(p->ns, p->nc) ? 1 : 0;
    if (impl == 6) return (p->dtype == MPC_F32 && tiny_supported(p->ns, p->nc) && wave1_supported(make_params<float>(p, o, nullptr))) ? 1 : 0;
    if (impl == 5) {
        if (p->dtype != MPC_F32) return 0;
        mpc_lqr_outputs out;
        memset(&out, 0, sizeof(out));
        StepParams<float> sp = make_params<float>(p, o, &out);
        return (sp.ns == 32 && sp.nc == 8 && !sp.env.kind) ? 1 : 0;
    }
    if (impl == 7) {
        if (p->dtype != MPC_F32) return 0;
        mpc_lqr_outputs out;
        memset(&out, 0, sizeof(out));
        return mfma40_pad_supported(make_params<float>(p, o, &out)) ? 1 : 0;
    }
    if (impl == 8) {
        if (p->dtype != MPC_F32) return 0;
        mpc_lqr_outputs out;
        memset(&out, 0, sizeof(out));
        return dpp16_pad_supported(make_params<float>(p, o, &out)) ? 1 : 0;
    }
    if (impl == 2 && p->dtype == MPC_F64) {
        mpc_lqr_outputs out;
        memset(&out, 0, sizeof(out));
        return mfma16_supported(make_params<double>(p, o, &out)) ? 1 : 0;
    }
    if (impl == 2 || impl == 3) {
        if (p->dtype != MPC_F32) return 0;
        mpc_lqr_outputs out;
        memset(&out, 0, sizeof(out));
        StepParams<float> sp = make_params<float>(p, o, &out);
        if (impl == 2) return mfma16_supported(sp) ? 1 : 0;
        // sizes only: the alignment of the actual tensors is checked at launch
        return (sp.ns == 12 && sp.nc == 4) ? 1 : 0;
    }
    return 0;
}

int mpc_lqr_qp_record(const mpc_lqr_problem *p, const mpc_lqr_options *o, int impl, int64_t *offset_bytes, int64_t *st, int64_t *sb)
{
    // mirrors the routing of step_impl for a box-constrained float32 step whose out->K / out->k are NULL
    if (!p || !o || !offset_bytes || !st || !sb) return 0;
    if (check_problem(p, false, false) != MPC_OK || check_options(p, o) != MPC_OK) return 0;
    if (p->dtype != MPC_F32 || o->bound_mode == MPC_BOUND_NONE || o->true_dynamics || p->B == 0) return 0;
    mpc_lqr_outputs out;
    memset(&out, 0, sizeof(out));
    const StepParams<float> sp = make_params<float>(p, o, &out);
    const int64_t TB = (int64_t)p->T * p->B;
    if ((impl == 0 || impl == 3) && p->ns == 12 && p->nc == 4 && dpp16_supported(sp)) {
        // the 12/4 kernel's gain record Kk[t][b][lane 0..15][4]: lane 12 holds k.  (only where that kernel really takes
        // the call -- blocks or strides that are not 16-byte aligned go to the one-problem-per-wavefront kernel under impl 0, whose
        // record keeps k elsewhere: "0 = not filled" below; a forced impl 3 fails at launch with MPC_E_DIMS)
        *offset_bytes = 48 * 4; *st = (int64_t)p->B * 64; *sb = 64;
        return 1;
    }
    if (impl == 3) return 0;
    if (impl == 0 && (tiny_supported(p->ns, p->nc) || mfma16_supported(sp))) return 0;
    if ((impl == 0 || impl == 5) && p->ns == 32 && p->nc == 8) {
        *offset_bytes = TB * 256 * 4; *st = (int64_t)p->B * 8; *sb = 8;       // K [T,B,8,32] | k [T,B,8]
        return 1;
    }
    if ((impl == 0 || impl == 7) && mfma40_pad_supported(sp)) {
        *offset_bytes = TB * 256 * 4; *st = (int64_t)p->B * 8; *sb = 8;       // the padded gains: K [T,B,8,32] | k [T,B,8] (entries >= n_ctrl zero)
        return 1;
    }
    return 0;
}

int mpc_lqr_sweep(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out, void *stream)
{
    int rc = check_problem(p, true, true);
    if (rc) return rc;
    if ((rc = check_options(p, o))) return rc;
    if (!out) return fail(MPC_E_NULL, "outputs is NULL");
    if (p->B == 0) return MPC_OK;
    hipStream_t st = (hipStream_t)stream;
    return p->dtype == MPC_F32 ? step_impl<float>(p, o, out, nullptr, 0, 1, 1, nullptr, st)
                               : step_impl<double>(p, o, out, nullptr, 0, 1, 1, nullptr, st);
}

int mpc_lqr_rollout(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                    const void *old_costs_in, void *stream)
{
    int rc = check_problem(p, true, true);
    if (rc) return rc;
    if ((rc = check_options(p, o))) return rc;
    if (!out) return fail(MPC_E_NULL, "outputs is NULL");
    if (p->B == 0) return MPC_OK;
    hipStream_t st = (hipStream_t)stream;
    return p->dtype == MPC_F32 ? step_impl<float>(p, o, out, nullptr, 0, 1, 2, old_costs_in, st)
                               : step_impl<double>(p, o, out, nullptr, 0, 1, 2, old_costs_in, st);
}

int mpc_lqr_kkt_grads(const mpc_lqr_problem *p, const void *dx, const void *du, const void *dl_dx,
                      const void *dl_du, void *dC, void *dc, void *dF, void *df, void *dx_init, void *stream)
{
    int rc = check_problem(p, true, true);
    if (rc) return rc;
    if (p->B == 0) return MPC_OK;
    if (!dx || !du || !dl_dx || !dl_du || !dC || !dc || !dx_init) return fail(MPC_E_NULL, "kkt_grads: NULL argument");
    if (p->T > 1 && !dF) return fail(MPC_E_NULL, "kkt_grads: dF is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == MPC_F32) {
        StepParams<float> sp = make_params<float>(p, nullptr, nullptr);
        // headline shape: the 4-problems-per-wave DPP kernel (lqr_dpp16_body.h: kkt_wave)
        if (kkt_dpp16_supported(sp, (const float *)dx, (const float *)du, (const float *)dl_dx, (const float *)dC,
                                (const float *)dF))
            return launch_kkt_dpp16(sp, (const float *)dx, (const float *)du, (const float *)dl_dx, (float *)dC,
                                    (float *)dc, (float *)dF, (float *)df, (float *)dx_init, st);
        // other shapes up to n = 64: costate recursion per wavefront + fully parallel outer products
        if (kkt_wave_supported(sp, (const float *)dx, (const float *)du, (const float *)dl_dx, (const float *)dC, (const float *)dF))
            return launch_kkt_wave(sp, (const float *)dx, (const float *)du, (const float *)dl_dx, (float *)dC,
                                   (float *)dc, (float *)dF, (float *)df, (float *)dx_init, st);
        return launch_kkt_grads<float>(sp, (const float *)dx, (const float *)du, (const float *)dl_dx,
                                       (const float *)dl_du, (float *)dC, (float *)dc, (float *)dF, (float *)df,
                                       (float *)dx_init, st);
    }
    StepParams<double> sp = make_params<double>(p, nullptr, nullptr);
    return launch_kkt_grads<double>(sp, (const double *)dx, (const double *)du, (const double *)dl_dx,
                                    (const double *)dl_du, (double *)dC, (double *)dc, (double *)dF, (double *)df,
                                    (double *)dx_init, st);
}

int mpc_lqr_kkt_fused_supported(const mpc_lqr_problem *p, const mpc_lqr_options *o)
{
    if (!p || check_problem(p, false, false) != MPC_OK || check_options(p, o) != MPC_OK) return 0;
    // (12/4: any horizon since round 4; every shape UP TO 12/4 since round 6: the padded instantiation, lqr_dpp16.hip -DMPC_DPP16_PAD_KKT)
    // ... and every shape up to 32/8: the padded instantiation of the 32/8 kernel's fused backward, lqr_mfma40.hip -DMPC_MFMA40_KKT -DMPC_MFMA40_PAD=4
    const bool s12 = p->ns >= 1 && p->ns <= 12 && p->nc >= 1 && p->nc <= 4, s32 = p->ns >= 1 && p->ns <= 32 && p->nc >= 1 && p->nc <= 8;
    if (p->dtype != MPC_F32 || !(s12 || s32)) return 0;
    if (!o || !(o->flags & MPC_OPT_C_SYMMETRIC)) return 0;
    // (u_zero_I and delta_u of the FORWARD are not inputs of the backward: the reference's nested solve is built from the bounds
    // alone, `u_zero_I = I` from u* and the bounds, `delta_u = None`, mpc/lqr_step.py:322-340 -- the launchers drop them; rounds
    // 1-3 refused such options here and the caller fell to three launches for nothing)
    if (o->true_dynamics) return 0;
    return 1;
}

int64_t mpc_lqr_kkt_fused_workspace_bytes(const mpc_lqr_problem *p)
{
    if (!p) return 0;
    return (p->ns > 12 || p->nc > 4) ? kkt_fused_mfma40_workspace_bytes(p->T, p->B) : kkt_fused_dpp16_workspace_bytes(p->T, p->B);
}

int mpc_lqr_kkt_fused(const mpc_lqr_problem *p, const mpc_lqr_options *o, const void *dl_dx, const void *dl_du,
                      void *dC, void *dc, void *dF, void *df, void *dx_init, void *dx_out, void *du_out, int32_t *status,
                      void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_problem(p, true, true);
    if (rc) return rc;
    if ((rc = check_options(p, o))) return rc;
    if (!mpc_lqr_kkt_fused_supported(p, o))
        return fail(MPC_E_DIMS, "mpc_lqr_kkt_fused: needs fp32, n_state <= 32, n_ctrl <= 8, and "
                                "MPC_OPT_C_SYMMETRIC (otherwise: mpc_lqr_kkt_prepare + mpc_lqr_step + mpc_lqr_kkt_grads)");
    if (p->B == 0) return MPC_OK;
    if (!dl_dx || !dl_du || !dC || !dc || !dx_init) return fail(MPC_E_NULL, "kkt_fused: NULL argument");
    if (p->T > 1 && !dF) return fail(MPC_E_NULL, "kkt_fused: dF is NULL");
    if ((df != nullptr) != (p->f != nullptr && p->T > 1)) return fail(MPC_E_NULL, "kkt_fused: df goes with f");
    if ((dx_out == nullptr) != (du_out == nullptr)) return fail(MPC_E_NULL, "kkt_fused: pass both dx_out and du_out, or neither");
    if (!workspace || workspace_bytes < mpc_lqr_kkt_fused_workspace_bytes(p))
        return fail(MPC_E_ARG, "workspace too small (see mpc_lqr_kkt_fused_workspace_bytes)");
    mpc_lqr_outputs out;
    memset(&out, 0, sizeof(out));
    out.status = status;
    StepParams<float> sp = make_params<float>(p, o, &out);
    if (p->ns > 12 || p->nc > 4) {
        // config 5's shape: the nested step with the costates riding along, then the outer-product kernel (two launches)
        if (!kkt_fused_mfma40_supported(sp, (const float *)dl_dx, (const float *)dl_du, (const float *)dC, (const float *)dF,
                                        (const float *)workspace) ||
            (dx_out && (((uintptr_t)dx_out | (uintptr_t)du_out) & 15)) || ((uintptr_t)dx_init & 15) || (df && ((uintptr_t)df & 15))) {
            // every other shape up to 32/8, and 32/8 itself off the 16-byte grid: the padded instantiation (round 6)
            if (!kkt_fused_mfma40_pad_supported(sp, (const float *)workspace))
                return fail(MPC_E_DIMS, "mpc_lqr_kkt_fused: the workspace must be 16-byte aligned (tensor bounds 4-byte aligned)");
            if (kkt_fused_mfma40_pad16_supported(sp, (const float *)workspace))
                return launch_kkt_fused_mfma40_pad16(sp, (const float *)dl_dx, (const float *)dl_du, (float *)dC, (float *)dc, (float *)dF,
                                                     (float *)df, (float *)dx_init, (float *)dx_out, (float *)du_out, (float *)workspace, 0.2f,
                                                     10, (hipStream_t)stream);
            return launch_kkt_fused_mfma40_pad(sp, (const float *)dl_dx, (const float *)dl_du, (float *)dC, (float *)dc, (float *)dF,
                                               (float *)df, (float *)dx_init, (float *)dx_out, (float *)du_out, (float *)workspace, 0.2f,
                                               10, (hipStream_t)stream);
        }
        return launch_kkt_fused_mfma40(sp, (const float *)dl_dx, (const float *)dl_du, (float *)dC, (float *)dc, (float *)dF,
                                       (float *)df, (float *)dx_init, (float *)dx_out, (float *)du_out, (float *)workspace, 0.2f,
                                       10, (hipStream_t)stream);
    }
    if (!kkt_fused_dpp16_supported(sp, (const float *)dl_dx, (const float *)dl_du, (const float *)dC, (const float *)dF,
                                   (const float *)workspace)) {
        // every other shape up to 12/4, and 12/4 itself where a block is not 16-byte aligned: the padded instantiation (round 6)
        if (!kkt_fused_dpp16_pad_supported(sp, (const float *)workspace))
            return fail(MPC_E_DIMS, "mpc_lqr_kkt_fused: the workspace must be 16-byte aligned");
        return launch_kkt_fused_dpp16_pad(sp, (const float *)dl_dx, (const float *)dl_du, (float *)dC, (float *)dc, (float *)dF,
                                          (float *)df, (float *)dx_init, (float *)dx_out, (float *)du_out, (float *)workspace, 0.2f, 10,
                                          (hipStream_t)stream);
    }
    // the nested solve is a plain LQRStep(...) in the reference (:328-338): linesearch_decay 0.2, max_linesearch_iter 10
    return launch_kkt_fused_dpp16(sp, (const float *)dl_dx, (const float *)dl_du, (float *)dC, (float *)dc, (float *)dF,
                                  (float *)df, (float *)dx_init, (float *)dx_out, (float *)du_out, (float *)workspace, 0.2f, 10,
                                  (hipStream_t)stream);
}

int mpc_env_traj_cost(const mpc_lqr_problem *p, const mpc_env_dynamics *env, void *x, void *cost, void *stream)
{
    if (!p || !env) return fail(MPC_E_NULL, "problem / simulator is NULL");
    if (p->B < 0 || p->T < 1) return fail(MPC_E_DIMS, "need B>=0, T>=1");
    if (p->dtype != MPC_F32 && p->dtype != MPC_F64) return fail(MPC_E_DTYPE, "dtype must be MPC_F32 or MPC_F64");
    int rc = check_env(env, p->ns, p->nc);
    if (rc) return rc;
    if (p->B == 0) return MPC_OK;
    if (!p->x_init || !p->cur_u) return fail(MPC_E_NULL, "x_init / u is NULL");
    if (cost && (!p->C || !p->c)) return fail(MPC_E_NULL, "cost requested but C / c is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == MPC_F32) {
        StepParams<float> sp = make_params<float>(p, nullptr, nullptr);
        set_env(sp.env, env);
        return launch_traj_cost<float>(sp, (float *)x, (float *)cost, st);
    }
    StepParams<double> sp = make_params<double>(p, nullptr, nullptr);
    set_env(sp.env, env);
    return launch_traj_cost<double>(sp, (double *)x, (double *)cost, st);
}

int mpc_env_linearize(const mpc_env_dynamics *env, int dtype, int64_t N, const void *x, const void *u, void *F,
                      void *f, void *stream)
{
    if (!env) return fail(MPC_E_NULL, "simulator is NULL");
    if (dtype != MPC_F32 && dtype != MPC_F64) return fail(MPC_E_DTYPE, "bad dtype");
    if (env->kind < MPC_ENV_PENDULUM || env->kind > MPC_ENV_CARTPOLE) return fail(MPC_E_ARG, "unknown simulator kind");
    int rc = check_env(env, env_ns(env->kind), 1);
    if (rc) return rc;
    if (N < 0) return fail(MPC_E_DIMS, "N < 0");
    if (N == 0) return MPC_OK;
    if (!x || !u || !F || !f) return fail(MPC_E_NULL, "env_linearize: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPC_F32) {
        EnvDesc<float> e;
        set_env(e, env);
        return launch_env_linearize<float>(e, (long)N, (const float *)x, (const float *)u, (float *)F, (float *)f, st);
    }
    EnvDesc<double> e;
    set_env(e, env);
    return launch_env_linearize<double>(e, (long)N, (const double *)x, (const double *)u, (double *)F, (double *)f, st);
}

int64_t mpc_mlp_workspace_bytes(const mpc_mlp_dynamics *net)
{
    if (!net || net->n_layers < 1 || net->n_layers > MPC_MLP_MAX_LAYERS) return 0;
    int64_t fl = 0;
    for (int l = 0; l < net->n_layers; ++l) {
        const int64_t in = (net->widths[l] + 15) & ~15, out = (net->widths[l + 1] + 15) & ~15;
        fl += out * (in + 4) + out;          // rows padded by 16 bytes (LDS banks), see nn_dynamics.hip
    }
    return fl * 4 + 256;
}

int mpc_mlp_supported(const mpc_mlp_dynamics *net, int n_state, int n_ctrl) { return nn_budget(net, n_state, n_ctrl); }

int mpc_mlp_rollout(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_mlp_dynamics *net, const void *K,
                    const void *k, const void *old_costs, const mpc_lqr_outputs *out, void *workspace,
                    int64_t workspace_bytes, void *stream)
{
    if (!p || !out) return fail(MPC_E_NULL, "problem / outputs is NULL");
    if (p->B < 0 || p->T < 1 || p->ns < 1 || p->nc < 1) return fail(MPC_E_DIMS, "need B>=0, T>=1, ns>=1, nc>=1");
    if (p->dtype != MPC_F32) return fail(MPC_E_DTYPE, "the network kernels are fp32 only");
    if (p->B == 0) return MPC_OK;
    if (!p->x_init || !p->cur_u) return fail(MPC_E_NULL, "x_init / current_u is NULL");
    if ((p->C == nullptr) != (p->c == nullptr)) return fail(MPC_E_NULL, "pass both C and c, or neither");
    if (!out->new_x) return fail(MPC_E_NULL, "new_x is NULL");
    if ((K == nullptr) != (k == nullptr)) return fail(MPC_E_NULL, "pass both K and k, or neither");
    if (K) {
        if (!p->cur_x || !p->C || !old_costs || !out->new_u)
            return fail(MPC_E_NULL, "line search needs current_x, C, c, old_costs and new_u");
        int rc = check_options(p, o);
        if (rc) return rc;
        if (o && o->true_dynamics) return fail(MPC_E_ARG, "mpc_mlp_rollout: options.true_dynamics must be NULL");
    }
    StepParams<float> sp = make_params<float>(p, K ? o : nullptr, out);
    sp.K = (float *)K;
    sp.k = (float *)k;
    sp.old_costs_in = (const float *)old_costs;
    return launch_nn_rollout(sp, net, workspace, workspace_bytes, (hipStream_t)stream);
}

int mpc_mlp_linearize(const mpc_mlp_dynamics *net, int n_state, int n_ctrl, int64_t N, const void *x, const void *u,
                      void *F, void *f, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (n_state < 1 || n_ctrl < 1 || N < 0) return fail(MPC_E_DIMS, "need n_state>=1, n_ctrl>=1, N>=0");
    if (N == 0) return MPC_OK;
    if (!x || !u || !F || !f) return fail(MPC_E_NULL, "mlp_linearize: NULL argument");
    return launch_nn_linearize(net, (long)N, n_state, n_ctrl, (const float *)x, (const float *)u, (float *)F, (float *)f,
                               workspace, workspace_bytes, (hipStream_t)stream);
}

int mpc_lqr_kkt_prepare(int dtype, int B, int T, int ns, int nc, const void *dl_dx, const void *dl_du,
                        const void *u_star, const mpc_lqr_options *o, void *negr, uint8_t *mask, void *stream)
{
    if (dtype != MPC_F32 && dtype != MPC_F64) return fail(MPC_E_DTYPE, "bad dtype");
    if (B < 0 || T < 1 || ns < 1 || nc < 1) return fail(MPC_E_DIMS, "bad dims");
    if (B == 0) return MPC_OK;
    if (!dl_dx || !dl_du || !negr) return fail(MPC_E_NULL, "kkt_prepare: NULL argument");
    const int mode = o ? o->bound_mode : MPC_BOUND_NONE;
    if (mask && (mode == MPC_BOUND_NONE || !u_star)) return fail(MPC_E_ARG, "kkt_prepare: mask needs bounds and u*");
    if (mask && mode == MPC_BOUND_TENSOR && (!o->lo || !o->hi)) return fail(MPC_E_NULL, "tensor bounds are NULL");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPC_F32)
        return launch_kkt_prepare<float>(B, T, ns, nc, (const float *)dl_dx, (const float *)dl_du,
                                         (const float *)u_star, mode, o ? (float)o->lo_s : 0.f,
                                         o ? (float)o->hi_s : 0.f, o ? (const float *)o->lo : nullptr,
                                         o ? (const float *)o->hi : nullptr, (float *)negr, mask, st);
    return launch_kkt_prepare<double>(B, T, ns, nc, (const double *)dl_dx, (const double *)dl_du,
                                      (const double *)u_star, mode, o ? o->lo_s : 0., o ? o->hi_s : 0.,
                                      o ? (const double *)o->lo : nullptr, o ? (const double *)o->hi : nullptr,
                                      (double *)negr, mask, st);
}

int mpc_pnqp_lu(int dtype, int B, int n, const void *H, const void *q, const void *lo, const void *hi,
                const void *x0, int n_iter, void *x, uint8_t *If_out, int32_t *iters, int32_t *status,
                void *Hfree, void *LU, int32_t *pivots, void *stream)
{
    if (dtype != MPC_F32 && dtype != MPC_F64) return fail(MPC_E_DTYPE, "bad dtype");
    if (B < 0 || n < 1) return fail(MPC_E_DIMS, "bad dims");
    if (B == 0) return MPC_OK;
    if (!H || !q || !lo || !hi || !x) return fail(MPC_E_NULL, "pnqp: NULL argument");
    if (n_iter < 1) n_iter = 20;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPC_F32)
        return launch_pnqp<float>(B, n, (const float *)H, (const float *)q, (const float *)lo, (const float *)hi,
                                  (const float *)x0, n_iter, (float *)x, If_out, iters, status, (float *)Hfree,
                                  (float *)LU, pivots, st);
    return launch_pnqp<double>(B, n, (const double *)H, (const double *)q, (const double *)lo, (const double *)hi,
                               (const double *)x0, n_iter, (double *)x, If_out, iters, status, (double *)Hfree,
                               (double *)LU, pivots, st);
}

int mpc_pnqp(int dtype, int B, int n, const void *H, const void *q, const void *lo, const void *hi,
             const void *x0, int n_iter, void *x, uint8_t *If_out, int32_t *iters, int32_t *status,
             void *Hfree, void *stream)
{
    return mpc_pnqp_lu(dtype, B, n, H, q, lo, hi, x0, n_iter, x, If_out, iters, status, Hfree, nullptr, nullptr, stream);
}

int mpc_traj_cost(const mpc_lqr_problem *p, void *x, void *cost, void *stream)
{
    if (!p) return fail(MPC_E_NULL, "problem is NULL");
    int rc = check_problem(p, cost != nullptr, false);
    if (rc) return rc;
    if (p->B == 0) return MPC_OK;
    if (!p->cur_u) return fail(MPC_E_NULL, "traj_cost: u (cur_u) is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == MPC_F32) {
        StepParams<float> sp = make_params<float>(p, nullptr, nullptr);
        if (!cost && x && traj_wave_supported(sp)) return launch_traj_wave(sp, (float *)x, st);
        return launch_traj_cost<float>(sp, (float *)x, (float *)cost, st);
    }
    StepParams<double> sp = make_params<double>(p, nullptr, nullptr);
    return launch_traj_cost<double>(sp, (double *)x, (double *)cost, st);
}

int mpc_select_best(int dtype, int B, int T, int ns, int nc, int first, double best_cost_eps, const void *x,
                    const void *u, const void *costs, const void *du_norm, void *best_x, void *best_u,
                    void *best_costs, void *best_du_norm, void *flags, void *host_flags, int32_t host_tag,
                    const int32_t *status, void *stream)
{
    if (dtype != MPC_F32 && dtype != MPC_F64) return fail(MPC_E_DTYPE, "bad dtype");
    if (B < 0 || T < 1 || ns < 1 || nc < 1) return fail(MPC_E_DIMS, "bad dims");
    if (B == 0) return MPC_OK;
    if (!x || !u || !costs || !du_norm || !best_x || !best_u || !best_costs || !best_du_norm || !flags)
        return fail(MPC_E_NULL, "select_best: NULL argument");
    if ((uintptr_t)flags & 7 || (uintptr_t)host_flags & 7) return fail(MPC_E_ARG, "select_best: the flags blocks must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPC_F32)
        return launch_select_best<float>(B, T, ns, nc, first, (float)best_cost_eps, (const float *)x,
                                         (const float *)u, (const float *)costs, (const float *)du_norm,
                                         (float *)best_x, (float *)best_u, (float *)best_costs,
                                         (float *)best_du_norm, flags, host_flags, host_tag, status, st);
    return launch_select_best<double>(B, T, ns, nc, first, best_cost_eps, (const double *)x, (const double *)u,
                                      (const double *)costs, (const double *)du_norm, (double *)best_x,
                                      (double *)best_u, (double *)best_costs, (double *)best_du_norm,
                                      flags, host_flags, host_tag, status, st);
}

int mpc_du_norm_reference(int dtype, int T, int B, int nc, const void *u, const void *new_u, void *out, void *stream)
{
    if (dtype != MPC_F32 && dtype != MPC_F64) return fail(MPC_E_DTYPE, "bad dtype");
    if (B < 0 || T < 1 || nc < 1) return fail(MPC_E_DIMS, "bad dims");
    if (B == 0) return MPC_OK;
    if (!u || !new_u || !out) return fail(MPC_E_NULL, "du_norm_reference: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MPC_F32) return launch_du_norm_reference<float>(T, B, nc, (const float *)u, (const float *)new_u, (float *)out, st);
    return launch_du_norm_reference<double>(T, B, nc, (const double *)u, (const double *)new_u, (double *)out, st);
}

}  // extern "C"
