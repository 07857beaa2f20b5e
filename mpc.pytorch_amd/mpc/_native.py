"""ctypes binding of libmpc_lqr_hip.so (C ABI: include/mpc_lqr.h) + the tensor-level backend.

torch is used here for device memory and the current HIP stream only; every
computation on the LQR hot path happens inside the shared library's gfx950
kernels.  There is NO CPU / eager fallback: if the library is missing or the
tensors are not on a ROCm device the calls raise.
"""
import ctypes
import math
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmpc_lqr_hip.so"

MPC_F32, MPC_F64 = 0, 1
BOUND_NONE, BOUND_SCALAR, BOUND_TENSOR = 0, 1, 2
ST_PNQP_UNCONVERGED, ST_NONFINITE, ST_NOMINAL_OFF_DYNAMICS, ST_C_ASYMMETRIC, ST_QUU_SINGULAR, ST_C_TESTED = 1, 2, 4, 8, 16, 32
IMPL_AUTO, IMPL_GENERIC, IMPL_MFMA16, IMPL_DPP16, IMPL_TINY, IMPL_MFMA40, IMPL_WAVE1, IMPL_MFMA40_PAD, IMPL_DPP16_PAD = 0, 1, 2, 3, 4, 5, 6, 7, 8

ABI_VERSION = 9      # include/mpc_lqr.h: MPC_LQR_ABI_VERSION

_vp, _i32, _i64, _f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double


class Problem(ctypes.Structure):
    _fields_ = [("B", _i32), ("T", _i32), ("ns", _i32), ("nc", _i32), ("dtype", _i32), ("_pad", _i32),
                ("x_init", _vp),
                ("C", _vp), ("C_st", _i64), ("C_sb", _i64),
                ("c", _vp), ("c_st", _i64), ("c_sb", _i64),
                ("F", _vp), ("F_st", _i64), ("F_sb", _i64),
                ("f", _vp), ("f_st", _i64), ("f_sb", _i64),
                ("cur_x", _vp), ("cur_u", _vp)]


class EnvDynamics(ctypes.Structure):
    _fields_ = [("kind", _i32), ("linearize", _i32), ("params", _vp), ("dt", _f64), ("u_max", _f64)]


class Options(ctypes.Structure):
    _fields_ = [("bound_mode", _i32), ("max_linesearch_iter", _i32), ("lo_s", _f64), ("hi_s", _f64),
                ("lo", _vp), ("hi", _vp), ("zero_mask", _vp), ("delta_u", _f64),
                ("linesearch_decay", _f64), ("pnqp_iter", _i32), ("flags", _i32),
                ("true_dynamics", ctypes.POINTER(EnvDynamics)),
                ("qp_start", _vp), ("qp_start_st", _i64), ("qp_start_sb", _i64)]


ENV_PENDULUM, ENV_PENDULUM_FULL, ENV_CARTPOLE = 1, 2, 3
OPT_NOMINAL_ON_DYNAMICS = 1          # mpc_lqr_options.flags
OPT_SWEEP_ONLY = 2
OPT_C_SYMMETRIC = 4                  # the caller vouches for C = C' (no symmetry test, no gated re-solve)


class EnvSpec:
    """What a shipped simulator module hands to the kernels: kind (ENV_*), its parameter tensor,
    the integration step and the control clamp (mpc/env_dx/pendulum.py:23-37, cartpole.py:36-49)."""

    def __init__(self, kind, params, dt, u_max, linearize=False):
        self.kind, self.params, self.dt, self.u_max = int(kind), params, float(dt), float(u_max)
        self.linearize = bool(linearize)     # the step kernel linearises the simulator itself (F, f not passed)
        self.n_state = 5 if self.kind == ENV_CARTPOLE else 3
        self.n_ctrl = 1

    def to_struct(self, like):
        prm = _device_copy_of(self.params, like.device, like.dtype)
        e = EnvDynamics()
        e.kind, e.params, e.dt, e.u_max = self.kind, prm.data_ptr(), self.dt, self.u_max
        e.linearize = int(self.linearize)
        return e, prm


_PARAM_COPIES = {}


def _device_copy_of(t, device, dtype):
    """`t` on `device` as `dtype`, contiguous.  A simulator's parameter block (five numbers) usually lives on the host and
    MPC.forward asks for it three times per solve: the copy is kept while the tensor is the same object, its version counter
    has not moved (an optimiser step moves it) AND its contents still equal the snapshot taken with the copy -- edits through
    `params.data`, `.data.clamp_()` or a numpy alias do not move the counter.  The comparison is a few
    microseconds for a block this small; tensors of more than 64 elements, or on another device than the host, are not
    cached at all.  `invalidate_param_copies()` drops every cached copy."""
    if t.device == device and t.dtype == dtype and t.is_contiguous():
        return t.detach()
    if t.numel() > 64 or t.device.type != "cpu":
        return t.detach().to(device=device, dtype=dtype).contiguous()
    import weakref
    key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, str(device), dtype)
    hit = _PARAM_COPIES.get(key)
    if hit is not None and hit[0]() is t and hit[1] == t._version and torch.equal(hit[3], t.detach()):
        return hit[2]
    if len(_PARAM_COPIES) > 64:
        _PARAM_COPIES.clear()
    c = t.detach().to(device=device, dtype=dtype).contiguous()
    _PARAM_COPIES[key] = (weakref.ref(t), t._version, c, t.detach().clone())
    return c


def invalidate_param_copies():
    """Forget every cached device copy of a simulator parameter block (see _device_copy_of)."""
    _PARAM_COPIES.clear()


MLP_MAX_LAYERS = 4
ACT_CODES = {"sigmoid": 0, "relu": 1, "elu": 2}


class MlpDynamics(ctypes.Structure):          # struct mpc_mlp_dynamics
    _fields_ = [("n_layers", _i32), ("activation", _i32), ("passthrough", _i32), ("ctrl_carry", _i32),
                ("widths", _i32 * (MLP_MAX_LAYERS + 1)), ("W", _vp * MLP_MAX_LAYERS), ("b", _vp * MLP_MAX_LAYERS)]


class MlpSpec:
    """What mpc.dynamics.NNDynamics hands to the kernels: its Linear layers' weights and biases (as they are NOW:
    the struct is rebuilt for every call, a training loop changes them between calls), the activation and the
    passthrough flag (mpc/dynamics.py:15-36)."""

    def __init__(self, weights, biases, activation, passthrough, ctrl_carry=0):
        self.weights, self.biases = list(weights), list(biases)
        self.activation, self.passthrough = activation, bool(passthrough)
        self.ctrl_carry = int(ctrl_carry)      # CtrlPassthroughDynamics around the network (see `augmented`)
        self.n_state = self.weights[-1].shape[0]
        self.n_ctrl = self.weights[0].shape[1] - self.n_state

    def augmented(self):
        """The same network as the dynamics of the slew-rate augmentation (mpc/dynamics.py:131-150): state (previous
        control, x), a step returns (this control, net(x, u)).  Zero columns for the previous control in the first layer,
        zero rows for it in the last; the kernels write the control itself into those rows (`ctrl_carry`)."""
        nc = self.n_ctrl
        W, b = list(self.weights), list(self.biases)
        W[0] = torch.cat((W[0].new_zeros(W[0].shape[0], nc), W[0]), 1) if len(W) > 1 else W[0]
        if len(W) == 1:
            W[0] = torch.cat((W[0].new_zeros(nc, W[0].shape[1] + nc),
                              torch.cat((W[0].new_zeros(W[0].shape[0], nc), W[0]), 1)), 0)
        else:
            W[-1] = torch.cat((W[-1].new_zeros(nc, W[-1].shape[1]), W[-1]), 0)
        b[-1] = torch.cat((b[-1].new_zeros(nc), b[-1]), 0)
        return MlpSpec(W, b, self.activation, self.passthrough, ctrl_carry=nc)

    @staticmethod
    def supported(weights, activation, like):
        """fp32 on the device of `like`, at most four layers, n_state <= 16, and layer widths whose staging areas fit the
        160 KiB of LDS the kernels work in -- the library's own test (mpc_mlp_supported: both the rollout and the
        linearisation kernel must take the network, e.g. NNDynamics(4, 1, [1024]) does not and keeps the module path)."""
        if not (like.is_cuda and like.dtype == torch.float32 and 1 <= len(weights) <= MLP_MAX_LAYERS
                and activation in ACT_CODES and weights[-1].shape[0] <= 16
                and all(W.shape[0] <= 4096 for W in weights)
                and all(W.is_cuda and W.dtype == torch.float32 for W in weights)):
            return False
        return MlpSpec.widths_supported([weights[0].shape[1]] + [W.shape[0] for W in weights])

    @staticmethod
    def widths_supported(widths):
        """mpc_mlp_supported for a network of these layer widths ([n_state + n_ctrl, hidden..., n_state])."""
        e = MlpDynamics()
        e.n_layers = len(widths) - 1
        for l, w in enumerate(widths):
            e.widths[l] = int(w)
        ns = int(widths[-1])
        return int(load().mpc_mlp_supported(ctypes.byref(e), ns, int(widths[0]) - ns)) == 3

    def to_struct(self, like):
        e = MlpDynamics()
        keep = []
        e.n_layers, e.activation, e.passthrough = len(self.weights), ACT_CODES[self.activation], int(self.passthrough)
        e.ctrl_carry = self.ctrl_carry
        e.widths[0] = self.weights[0].shape[1]
        for l, (W, b) in enumerate(zip(self.weights, self.biases)):
            W = W.detach().to(device=like.device, dtype=torch.float32).contiguous()
            b = b.detach().to(device=like.device, dtype=torch.float32).contiguous()
            keep += [W, b]
            e.widths[l + 1] = W.shape[0]
            e.W[l], e.b[l] = W.data_ptr(), b.data_ptr()
        nbytes = int(load().mpc_mlp_workspace_bytes(ctypes.byref(e)))
        ws = torch.empty(nbytes, device=like.device, dtype=torch.uint8)
        keep.append(ws)
        return e, ws, nbytes, keep


class Outputs(ctypes.Structure):
    _fields_ = [("new_x", _vp), ("new_u", _vp), ("costs", _vp), ("old_costs", _vp), ("full_du_norm", _vp),
                ("alpha_du_norm", _vp), ("alphas", _vp), ("qp_iters", _vp), ("status", _vp),
                ("K", _vp), ("k", _vp)]


EXPORTS = ("mpc_lqr_abi_version", "mpc_lqr_build_info", "mpc_lqr_last_error", "mpc_lqr_workspace_bytes",
           "mpc_lqr_step", "mpc_lqr_impl_supported", "mpc_lqr_qp_record", "mpc_lqr_sweep", "mpc_lqr_rollout", "mpc_lqr_kkt_grads", "mpc_lqr_kkt_prepare",
           "mpc_pnqp", "mpc_pnqp_lu", "mpc_traj_cost", "mpc_env_traj_cost", "mpc_env_linearize", "mpc_select_best",
           "mpc_mlp_workspace_bytes", "mpc_mlp_rollout", "mpc_mlp_linearize",
           "mpc_mlp_supported", "mpc_lqr_kkt_fused_supported", "mpc_lqr_kkt_fused_workspace_bytes", "mpc_lqr_kkt_fused",
           "mpc_du_norm_reference")

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib_path():
    return os.environ.get("MPC_LQR_HIP_LIB", os.path.join(_HERE, LIB_NAME))


def load():
    """dlopen the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            "%s not found at %s -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % (LIB_NAME, path))
    L = ctypes.CDLL(path)
    for name in EXPORTS:
        getattr(L, name)   # AttributeError if the ABI is incomplete
    L.mpc_lqr_build_info.restype = ctypes.c_char_p
    L.mpc_lqr_last_error.restype = ctypes.c_char_p
    L.mpc_lqr_workspace_bytes.restype = _i64
    L.mpc_lqr_workspace_bytes.argtypes = [ctypes.POINTER(Problem)]
    PP, OP, UP = ctypes.POINTER(Problem), ctypes.POINTER(Options), ctypes.POINTER(Outputs)
    L.mpc_lqr_step.argtypes = [PP, OP, UP, _vp, _i64, ctypes.c_int, _vp]
    L.mpc_lqr_impl_supported.argtypes = [PP, OP, ctypes.c_int]
    L.mpc_lqr_qp_record.argtypes = [PP, OP, ctypes.c_int] + [ctypes.POINTER(_i64)] * 3
    L.mpc_lqr_sweep.argtypes = [PP, OP, UP, _vp]
    L.mpc_lqr_rollout.argtypes = [PP, OP, UP, _vp, _vp]
    L.mpc_du_norm_reference.argtypes = [ctypes.c_int] * 4 + [_vp] * 4
    L.mpc_lqr_kkt_grads.argtypes = [PP] + [_vp] * 10
    L.mpc_lqr_kkt_prepare.argtypes = [ctypes.c_int] * 5 + [_vp, _vp, _vp, OP, _vp, _vp, _vp]
    L.mpc_lqr_kkt_fused_supported.argtypes = [PP, OP]
    L.mpc_lqr_kkt_fused_workspace_bytes.restype = _i64
    L.mpc_lqr_kkt_fused_workspace_bytes.argtypes = [PP]
    L.mpc_lqr_kkt_fused.argtypes = [PP, OP] + [_vp] * 11 + [_i64, _vp]
    L.mpc_pnqp.argtypes = [ctypes.c_int] * 3 + [_vp] * 5 + [ctypes.c_int] + [_vp] * 6
    L.mpc_pnqp_lu.argtypes = [ctypes.c_int] * 3 + [_vp] * 5 + [ctypes.c_int] + [_vp] * 8
    L.mpc_traj_cost.argtypes = [PP, _vp, _vp, _vp]
    L.mpc_env_traj_cost.argtypes = [PP, ctypes.POINTER(EnvDynamics), _vp, _vp, _vp]
    L.mpc_env_linearize.argtypes = [ctypes.POINTER(EnvDynamics), ctypes.c_int, _i64, _vp, _vp, _vp, _vp, _vp]
    L.mpc_select_best.argtypes = [ctypes.c_int] * 6 + [_f64] + [_vp] * 10 + [ctypes.c_int32, _vp, _vp]
    MP = ctypes.POINTER(MlpDynamics)
    L.mpc_mlp_workspace_bytes.restype = _i64
    L.mpc_mlp_workspace_bytes.argtypes = [MP]
    L.mpc_mlp_supported.argtypes = [MP, ctypes.c_int, ctypes.c_int]
    L.mpc_mlp_rollout.argtypes = [PP, OP, MP, _vp, _vp, _vp, UP, _vp, _i64, _vp]
    L.mpc_mlp_linearize.argtypes = [MP, ctypes.c_int, ctypes.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]
    if L.mpc_lqr_abi_version() != ABI_VERSION:
        raise RuntimeError("libmpc_lqr_hip ABI version mismatch")
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().mpc_lqr_last_error().decode()))


def _dtype_code(t):
    if t.dtype == torch.float32:
        return MPC_F32
    if t.dtype == torch.float64:
        return MPC_F64
    raise TypeError("the LQR kernels take float32 or float64 tensors, got %s" % t.dtype)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _require_device(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "mpc (MI355X build): the LQR step runs only on ROCm device tensors; got a %s tensor. "
                "There is no CPU fallback -- move the problem to the GPU." % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("all tensors of one LQR problem must live on the same device")
    return dev


def _block_strided(t, inner_dims):
    """Return (tensor, stride_T, stride_B) with the trailing `inner_dims` dims densely packed.
    Stride-0 (expanded) leading axes are kept as they are -- the kernels honour them."""
    inner = t.shape[-inner_dims:]
    want = []
    acc = 1
    for d in reversed(inner):
        want.append(acc)
        acc *= d
    want = tuple(reversed(want))
    ok = all(t.shape[-inner_dims + i] == 1 or t.stride()[-inner_dims + i] == want[i] for i in range(inner_dims))
    if not ok:
        t = t.contiguous()
    return t, t.stride(0), t.stride(1)


class StepOptions:
    """The LQRStep keyword arguments that reach the kernels (mpc/lqr_step.py:22-38 of the reference)."""

    def __init__(self, u_lower=None, u_upper=None, u_zero_I=None, delta_u=None, linesearch_decay=0.2,
                 max_linesearch_iter=10, pnqp_iter=20, true_dynamics=None, nominal_on_dynamics=False, sweep_only=False,
                 c_symmetric=False, qp_start=None):
        assert (u_lower is None) == (u_upper is None)
        # mpc_lqr_options.qp_start: a [T,B,nc] tensor (any T / B strides, stride-0 views included; last axis contiguous) the box
        # QPs of the sweep start from, in the delta space of this call's nominal -- a hint: results do not depend on it
        self.qp_start = qp_start
        # the caller guarantees C_t = C_t' (MPC_OPT_C_SYMMETRIC): mpc.MPC does from its second iteration on, once the first
        # step of the solve has reported no MPC_ST_C_ASYMMETRIC; a bare LQRStep never does
        self.c_symmetric = bool(c_symmetric)
        # the caller guarantees cur_x = rollout of cur_u through (F, f): MPC.forward's nominal always is (mpc/mpc.py:251)
        self.nominal_on_dynamics = bool(nominal_on_dynamics)
        self.sweep_only = bool(sweep_only)      # mpc_lqr_step stops after the Riccati sweep: K, k, old_costs, qp_iters only
        self.u_lower, self.u_upper, self.u_zero_I = u_lower, u_upper, u_zero_I
        self.true_dynamics = true_dynamics      # EnvSpec: the rollout calls the simulator, not F,f
        self.delta_u, self.linesearch_decay = delta_u, linesearch_decay
        self.max_linesearch_iter, self.pnqp_iter = max_linesearch_iter, pnqp_iter

    def to_struct(self, T, B, nc, like):
        """-> (Options, keepalive list)"""
        keep = []
        o = Options()
        o.max_linesearch_iter = int(self.max_linesearch_iter)
        o.linesearch_decay = float(self.linesearch_decay)
        o.delta_u = float("nan") if self.delta_u is None else float(self.delta_u)
        o.pnqp_iter = int(self.pnqp_iter)
        o.flags = ((OPT_NOMINAL_ON_DYNAMICS if self.nominal_on_dynamics else 0) | (OPT_SWEEP_ONLY if self.sweep_only else 0)
                   | (OPT_C_SYMMETRIC if self.c_symmetric else 0))
        lo, hi = self.u_lower, self.u_upper
        if lo is None:
            o.bound_mode = BOUND_NONE
        elif isinstance(lo, (float, int)) and isinstance(hi, (float, int)):
            o.bound_mode, o.lo_s, o.hi_s = BOUND_SCALAR, float(lo), float(hi)
        else:
            def full(v):
                if not torch.is_tensor(v):
                    v = torch.full((1,), float(v))
                v = v.detach().to(device=like.device, dtype=like.dtype)
                return v.expand(T, B, nc).contiguous()
            lo_t, hi_t = full(lo), full(hi)
            keep += [lo_t, hi_t]
            o.bound_mode, o.lo, o.hi = BOUND_TENSOR, lo_t.data_ptr(), hi_t.data_ptr()
        if self.u_zero_I is not None:
            m = self.u_zero_I.detach().to(device=like.device)
            m = (m != 0).to(torch.uint8).expand(T, B, nc).contiguous()
            keep.append(m)
            o.zero_mask = m.data_ptr()
        if self.true_dynamics is not None:
            e, prm = self.true_dynamics.to_struct(like)
            keep += [e, prm]
            o.true_dynamics = ctypes.pointer(e)
        if self.qp_start is not None and lo is not None:
            q = self.qp_start
            if tuple(q.shape) != (T, B, nc) or q.dtype != like.dtype or q.device != like.device:
                raise ValueError("qp_start must be a [T,B,n_ctrl] tensor of the problem's dtype and device")
            if nc > 1 and q.stride(2) != 1:
                q = q.contiguous()
            keep.append(q)
            o.qp_start, o.qp_start_st, o.qp_start_sb = q.data_ptr(), q.stride(0), q.stride(1)
        return o, keep


class HipBackend:
    """Tensor-level front of the C ABI.  All methods enqueue on torch's current stream."""

    name = "hip-gfx950"

    # -- helpers -------------------------------------------------------------------------------
    @staticmethod
    def _problem(x_init, C, c, F, f, cur_x, cur_u):
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_init.shape[1]
        nc = n - ns
        keep = []
        p = Problem()
        p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, _dtype_code(C)
        xi = x_init.detach().contiguous(); keep.append(xi); p.x_init = xi.data_ptr()
        Cc, p.C_st, p.C_sb = _block_strided(C.detach(), 2); keep.append(Cc); p.C = Cc.data_ptr()
        cc, p.c_st, p.c_sb = _block_strided(c.detach(), 1); keep.append(cc); p.c = cc.data_ptr()
        if T > 1 and F is not None:
            Fc, p.F_st, p.F_sb = _block_strided(F.detach(), 2); keep.append(Fc); p.F = Fc.data_ptr()
        if f is not None and f.numel() > 0:
            fc, p.f_st, p.f_sb = _block_strided(f.detach(), 1); keep.append(fc); p.f = fc.data_ptr()
        if cur_x is not None:
            cx = cur_x.detach().contiguous(); keep.append(cx); p.cur_x = cx.data_ptr()
        if cur_u is not None:
            cu = cur_u.detach().contiguous(); keep.append(cu); p.cur_u = cu.data_ptr()
        return p, keep

    @staticmethod
    def _check_same(C, *others):
        for t in others:
            if t is not None and t.numel() > 0 and (t.dtype != C.dtype or t.device != C.device):
                raise TypeError("all tensors of one LQR problem must share dtype and device")

    def impl_supported(self, ns, nc, dtype, impl, opts=None):
        p = Problem()
        p.B, p.T, p.ns, p.nc = 1, 1, ns, nc
        p.dtype = MPC_F32 if dtype == torch.float32 else MPC_F64
        p.x_init = 16        # the query only looks at sizes and dtype; pointers are never dereferenced
        o = None
        if opts is not None:
            o, _keep = opts.to_struct(1, 1, nc, torch.empty(0, dtype=dtype))
        return bool(load().mpc_lqr_impl_supported(ctypes.byref(p), None if o is None else ctypes.byref(o), int(impl)))

    # -- (1) LQRStepFn.forward ------------------------------------------------------------------
    def lqr_step(self, x_init, C, c, F, f, cur_x, cur_u, opts, want_gains=False, impl=IMPL_AUTO,
                 rollout_problem=None, out_x=None, out_u=None):
        """c_back + Riccati sweep + line-searched rollout.  Returns a dict of device tensors.
        out_x / out_u: contiguous [T,B,ns] / [T,B,nc] tensors the kernel writes the new trajectory into (mpc.shard: views
        of the all-gather's receive buffer).

        rollout_problem: optional (C, c, F, f) the rollout/true cost should use when they differ
        from the sweep's (mpc/lqr_step.py:218-232 reads true_dynamics / true_cost)."""
        dev = _require_device(x_init, C, c, F, cur_x, cur_u)
        self._check_same(C, x_init, c, F, f, cur_x, cur_u)
        L = load()
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_init.shape[1]
        nc = n - ns
        p, keep = self._problem(x_init, C, c, F, f, cur_x, cur_u)
        o, keep_o = opts.to_struct(T, B, nc, C)
        kw = dict(device=dev, dtype=C.dtype)
        if out_x is not None:
            assert out_x.is_contiguous() and out_u.is_contiguous() and tuple(out_x.shape) == (T, B, ns) and tuple(out_u.shape) == (T, B, nc)
            assert out_x.dtype == C.dtype and out_x.device == C.device and out_u.dtype == C.dtype and out_u.device == C.device
        res = dict(new_x=torch.empty(T, B, ns, **kw) if out_x is None else out_x,
                   new_u=torch.empty(T, B, nc, **kw) if out_u is None else out_u,
                   costs=torch.empty(B, **kw), old_costs=torch.empty(B, **kw),
                   full_du_norm=torch.empty(B, **kw), alpha_du_norm=torch.empty(B, **kw),
                   alphas=torch.empty(B, **kw),
                   qp_iters=torch.zeros(B, device=dev, dtype=torch.int32),
                   status=torch.zeros(B, device=dev, dtype=torch.int32))
        out = Outputs()
        for k in ("new_x", "new_u", "costs", "old_costs", "full_du_norm", "alpha_du_norm", "alphas",
                  "qp_iters", "status"):
            setattr(out, k, res[k].data_ptr())
        split = rollout_problem is not None
        ws = None
        if want_gains or split:
            res["K"] = torch.empty(T, B, nc, ns, **kw)
            res["k"] = torch.empty(T, B, nc, **kw)
            out.K, out.k = res["K"].data_ptr(), res["k"].data_ptr()
        st = _stream(dev)
        if split:
            _check(L.mpc_lqr_sweep(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), st), "mpc_lqr_sweep")
            rC, rc_, rF, rf = rollout_problem
            p2, keep2 = self._problem(x_init, rC, rc_, rF, rf, cur_x, cur_u)
            _check(L.mpc_lqr_rollout(ctypes.byref(p2), ctypes.byref(o), ctypes.byref(out), None, st),
                   "mpc_lqr_rollout")
            keep += keep2
        else:
            # scratch for the gains (the fused MFMA kernel always parks its own record there)
            nbytes = int(L.mpc_lqr_workspace_bytes(ctypes.byref(p)))
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            _check(L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), _ptr(ws), nbytes,
                                  int(impl), st), "mpc_lqr_step")
        res["_keep"] = (keep, keep_o, ws)
        return res

    def plan_step(self, x_init, C, c, F, f, cur_x, cur_u, opts, impl=IMPL_AUTO, out_x=None, out_u=None,
                  workspace=None):
        """Pre-bind one LQR step: argument structs and output buffers are built once, `plan()` is
        then a single C call (no allocation, hipGraph-capturable).  Outputs are overwritten by
        every call -- clone what must survive.  out_x / out_u: write the new trajectory into these
        (contiguous) tensors, e.g. the nominal buffers of the NEXT iteration's plan."""
        dev = _require_device(x_init, C, c, F, cur_x, cur_u)
        self._check_same(C, x_init, c, F, f, cur_x, cur_u)
        L = load()
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_init.shape[1]
        nc = n - ns
        p, keep = self._problem(x_init, C, c, F, f, cur_x, cur_u)
        o, keep_o = opts.to_struct(T, B, nc, C)
        kw = dict(device=dev, dtype=C.dtype)
        res = dict(new_x=torch.empty(T, B, ns, **kw) if out_x is None else out_x,
                   new_u=torch.empty(T, B, nc, **kw) if out_u is None else out_u)
        res.update(self._per_problem_outputs(B, dev, C.dtype))
        assert res["new_x"].is_contiguous() and res["new_u"].is_contiguous()
        out = Outputs()
        for k in res:
            setattr(out, k, res[k].data_ptr())
        nbytes = int(L.mpc_lqr_workspace_bytes(ctypes.byref(p)))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8) if workspace is None else workspace
        assert ws.numel() >= nbytes
        return self._bind_plan(p, o, out, res, ws, nbytes, int(impl), dev, (keep, keep_o))

    @staticmethod
    def _per_problem_outputs(B, dev, dtype):
        """The seven [B] outputs of a step as views of two allocations (one fill): a plan is built per MPC.forward call, and
        nine allocator calls + two fill launches per plan were a third of the host time in front of the first step."""
        fl = torch.empty(5, B, device=dev, dtype=dtype)
        it = torch.zeros(2, B, device=dev, dtype=torch.int32)
        return dict(costs=fl[0], old_costs=fl[1], full_du_norm=fl[2], alpha_du_norm=fl[3], alphas=fl[4],
                    qp_iters=it[0], status=it[1])

    @staticmethod
    def _bind_plan(p, o, out, res, ws, nbytes, impl, dev, keep):
        pp, op, up, wp = ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), ws.data_ptr()
        fn = load().mpc_lqr_step

        def run(stream=None):
            # stream: the raw handle of the stream to enqueue on (a loop that looked it up once); default: torch's current one
            rc = fn(pp, op, up, wp, nbytes, impl, torch.cuda.current_stream(dev).cuda_stream if stream is None else stream)
            if rc != 0:
                _check(rc, "mpc_lqr_step")
            return res
        run.outputs = res
        run._keep = (p, o, out, keep, ws)
        run._bind = (nbytes, impl, dev)
        return run

    @staticmethod
    def qp_record(plan):
        """The solutions k_t [T,B,n_ctrl] of the box QPs of `plan`'s sweep, as a strided VIEW of the plan's workspace
        (mpc_lqr_qp_record): what a later step at the same nominal passes as StepOptions(qp_start=...) -- that step may be
        a variant of `plan` on the very same workspace (it reads timestep t's block before it stores its own).  None when the
        step keeps no such record (no bounds, float64, a kernel without the hint)."""
        p, o, _out, _keep, ws = plan._keep
        _nbytes, impl, _dev = plan._bind
        off, st, sb = _i64(0), _i64(0), _i64(0)
        if not load().mpc_lqr_qp_record(ctypes.byref(p), ctypes.byref(o), int(impl), ctypes.byref(off), ctypes.byref(st), ctypes.byref(sb)):
            return None
        flat = ws[off.value:].view(torch.float32)
        return flat.as_strided((p.T, p.B, p.nc), (st.value, sb.value, 1))

    def plan_variant(self, plan, opts=None, cur_x=None, cur_u=None, out_x=None, out_u=None):
        """A second plan over the SAME problem tensors as `plan`, differing in the nominal it reads (`cur_x`, `cur_u`), the
        buffers it writes (`out_x`, `out_u`) and / or its options: the structs are copied and patched, nothing is walked or
        checked again (MPC.forward's ping-pong partner and its MPC_OPT_C_SYMMETRIC twins).  Without new out_x / out_u it also
        shares the per-problem outputs of `plan` -- the two must then never be in flight together with different readers
        (MPC.forward runs them on one stream, each consumed by the select that follows it)."""
        p0, o0, out0, keep0, ws = plan._keep
        nbytes, impl, dev = plan._bind
        p = Problem.from_buffer_copy(p0)
        out = Outputs.from_buffer_copy(out0)
        res = dict(plan.outputs)
        keep = [keep0]
        T, B, ns, nc = p.T, p.B, p.ns, p.nc
        if cur_x is not None:
            assert cur_x.is_contiguous() and cur_u.is_contiguous() and tuple(cur_x.shape) == (T, B, ns) and tuple(cur_u.shape) == (T, B, nc)
            p.cur_x, p.cur_u = cur_x.data_ptr(), cur_u.data_ptr()
            keep += [cur_x, cur_u]
        if out_x is not None:
            assert out_x.is_contiguous() and out_u.is_contiguous() and tuple(out_x.shape) == (T, B, ns) and tuple(out_u.shape) == (T, B, nc)
            res["new_x"], res["new_u"] = out_x, out_u
            res.update(self._per_problem_outputs(B, dev, out_x.dtype))
            for k in res:
                setattr(out, k, res[k].data_ptr())
        o = o0
        if opts is not None:
            o, keep_o = opts.to_struct(T, B, nc, res["new_x"])
            keep.append(keep_o)
        return self._bind_plan(p, o, out, res, ws, nbytes, impl, dev, keep)

    def lqr_sweep(self, x_init, C, c, F, cur_x, cur_u, opts):
        """c_back + lqr_backward only -> dict(K, k, old_costs, qp_iters, status)."""
        dev = _require_device(x_init, C, c, F, cur_x, cur_u)
        self._check_same(C, x_init, c, F, cur_x, cur_u)
        L = load()
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_init.shape[1]
        nc = n - ns
        p, keep = self._problem(x_init, C, c, F, None, cur_x, cur_u)
        o, keep_o = opts.to_struct(T, B, nc, C)
        kw = dict(device=dev, dtype=C.dtype)
        res = dict(K=torch.empty(T, B, nc, ns, **kw), k=torch.empty(T, B, nc, **kw),
                   old_costs=torch.empty(B, **kw), qp_iters=torch.zeros(B, device=dev, dtype=torch.int32),
                   status=torch.zeros(B, device=dev, dtype=torch.int32))
        out = Outputs()
        for k in res:
            setattr(out, k, res[k].data_ptr())
        # mpc_lqr_step with MPC_OPT_SWEEP_ONLY: the fused kernel of the shape stops after its sweep (the 12/4 kernel is
        # ten times the generic sweep behind mpc_lqr_sweep); other shapes take the generic sweep through the same call
        o.flags |= OPT_SWEEP_ONLY
        nbytes = int(L.mpc_lqr_workspace_bytes(ctypes.byref(p)))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        _check(L.mpc_lqr_step(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), _ptr(ws), nbytes, IMPL_AUTO, _stream(dev)),
               "mpc_lqr_step (sweep only)")
        res["_keep"] = (keep, keep_o, ws)
        return res

    def lqr_rollout(self, x_init, C, c, F, f, cur_x, cur_u, K, k, opts, old_costs=None):
        """lqr_forward alone (mpc/lqr_step.py:164-261) given the gains K [T,B,nc,ns], k [T,B,nc] of a sweep: mpc_lqr_rollout.
        Returns dict(new_x, new_u, costs, full_du_norm, alpha_du_norm, alphas)."""
        dev = _require_device(x_init, C, c, F, cur_x, cur_u, K, k)
        self._check_same(C, x_init, c, F, f, cur_x, cur_u, K, k)
        L = load()
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_init.shape[1]
        nc = n - ns
        p, keep = self._problem(x_init, C, c, F, f, cur_x, cur_u)
        o, keep_o = opts.to_struct(T, B, nc, C)
        kw = dict(device=dev, dtype=C.dtype)
        res = dict(new_x=torch.empty(T, B, ns, **kw), new_u=torch.empty(T, B, nc, **kw), costs=torch.empty(B, **kw),
                   full_du_norm=torch.empty(B, **kw), alpha_du_norm=torch.empty(B, **kw), alphas=torch.empty(B, **kw),
                   status=torch.zeros(B, device=dev, dtype=torch.int32))
        out = Outputs()
        for key in res:
            setattr(out, key, res[key].data_ptr())
        Kc, kc = K.contiguous(), k.contiguous()
        out.K, out.k = Kc.data_ptr(), kc.data_ptr()
        oc = None if old_costs is None else old_costs.contiguous()
        _check(L.mpc_lqr_rollout(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), None if oc is None else oc.data_ptr(), _stream(dev)),
               "mpc_lqr_rollout")
        res["_keep"] = (keep, keep_o, Kc, kc, oc)
        return res

    def du_norm_reference(self, u, new_u):
        """The reference's `full_du_norm` for n_batch > 1 (mpc/lqr_step.py:243-245: the transpose in front of the reshape mixes
        the problems of a batch): mpc_du_norm_reference.  u, new_u [T,B,nc] -> [B]."""
        dev = _require_device(u, new_u)
        T, B, nc = u.shape
        uc, nuc = u.contiguous(), new_u.contiguous()
        assert uc.dtype == nuc.dtype and tuple(nuc.shape) == (T, B, nc)
        out = torch.empty(B, device=dev, dtype=u.dtype)
        _check(load().mpc_du_norm_reference(_dtype_code(u), T, B, nc, uc.data_ptr(), nuc.data_ptr(), out.data_ptr(), _stream(dev)),
               "mpc_du_norm_reference")
        return out

    # -- (4) LQRStepFn.backward -----------------------------------------------------------------
    def _zero_nominal(self, T, B, ns, nc, kw):
        """The nested solve's nominal (x, u, x_init) = 0: read-only, so one copy per shape serves every backward
        (three fill kernels less per call)."""
        key = (T, B, ns, nc, kw["dtype"], str(kw["device"]))
        cache = self.__dict__.setdefault("_zero_cache", {})
        z = cache.get(key)
        if z is None:
            if len(cache) >= 8:
                cache.clear()
            z = cache[key] = (torch.zeros(T, B, ns, **kw), torch.zeros(T, B, nc, **kw), torch.zeros(B, ns, **kw))
        return z

    def kkt_backward(self, C, c, F, f, x_star, u_star, dl_dx, dl_du, opts, impl=IMPL_AUTO):
        """dx_init, dC, dc, dF, df of mpc/lqr_step.py:312-407 (reference), all on device."""
        dev = _require_device(C, c, F, x_star, u_star, dl_dx, dl_du)
        L = load()
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_star.shape[2]
        nc = n - ns
        kw = dict(device=dev, dtype=C.dtype)
        code = _dtype_code(C)
        st = _stream(dev)
        dl_dx = dl_dx.detach().to(**kw).contiguous()
        dl_du = dl_du.detach().to(**kw).contiguous()
        x_star = x_star.detach().contiguous()
        u_star = u_star.detach().contiguous()
        o, keep_o = opts.to_struct(T, B, nc, C)
        has_f = f is not None and f.numel() > 0
        if impl == IMPL_AUTO:
            # the whole backward in one launch where a kernel for it exists (12/4 or 32/8, fp32, C vouched symmetric)
            plan = self.plan_kkt_backward(C, c, F, f, x_star, u_star, dl_dx, dl_du, opts, _prepared=True)
            if plan is not None:
                g = plan()
                if g is not None:
                    return g
        negr = torch.empty(T, B, n, **kw)
        mask = None
        if o.bound_mode != BOUND_NONE:
            mask = torch.empty(T, B, nc, device=dev, dtype=torch.uint8)
        _check(L.mpc_lqr_kkt_prepare(code, B, T, ns, nc, dl_dx.data_ptr(), dl_du.data_ptr(), u_star.data_ptr(),
                                     ctypes.byref(o), negr.data_ptr(), _ptr(mask), st), "mpc_lqr_kkt_prepare")
        # nested solve of :328-340: one LQR step on (C, -r, F, f=None) from the zero nominal with
        # the active controls pinned; defaults linesearch_decay=0.2, max_linesearch_iter=10.
        zx, zu, z0 = self._zero_nominal(T, B, ns, nc, kw)
        # (the zero nominal obeys x+ = F tau with f = None: the step may skip verifying it)
        inner = StepOptions(u_zero_I=mask, nominal_on_dynamics=True, c_symmetric=opts.c_symmetric)
        sol = self.lqr_step(z0, C, negr, F, None, zx, zu, inner, impl=impl)
        p, keep = self._problem(z0, C, c, F, f, x_star, u_star)
        dC = torch.empty(T, B, n, n, **kw)
        dc = torch.empty(T, B, n, **kw)
        dF = torch.empty(F.shape, **kw)          # every kernel writes all of it (t < T-1 is all there is)
        df = torch.empty(T - 1, B, ns, **kw) if has_f else None
        dx_init = torch.empty(B, ns, **kw)
        _check(L.mpc_lqr_kkt_grads(ctypes.byref(p), sol["new_x"].data_ptr(), sol["new_u"].data_ptr(),
                                   dl_dx.data_ptr(), dl_du.data_ptr(), dC.data_ptr(), dc.data_ptr(),
                                   dF.data_ptr(), _ptr(df), dx_init.data_ptr(), st), "mpc_lqr_kkt_grads")
        return dict(dx_init=dx_init, dC=dC, dc=dc, dF=dF, df=df, dx=sol["new_x"], du=sol["new_u"],
                    _keep=(keep, keep_o, negr, mask, sol))

    def plan_kkt_backward(self, C, c, F, f, x_star, u_star, dl_dx, dl_du, opts, _prepared=False):
        """Pre-bind the fused KKT backward (mpc_lqr_kkt_fused): argument structs, the gradient buffers and the workspace are
        built once, `plan()` is then a single C call that overwrites the same outputs (clone what must survive) -- what a loop
        over many backward calls of one shape wants (bench.py; an allocation-per-call caller spends as long in the allocator
        as the GPU in the kernel).  Reads dl_dx / dl_du in place at every call.  None where no fused kernel covers the
        problem; plan() returns None if the library refuses the views (misaligned): use kkt_backward then."""
        dev = _require_device(C, c, F, x_star, u_star, dl_dx, dl_du)
        L = load()
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_star.shape[2]
        nc = n - ns
        kw = dict(device=dev, dtype=C.dtype)
        if not _prepared:
            dl_dx = dl_dx.detach().to(**kw).contiguous()
            dl_du = dl_du.detach().to(**kw).contiguous()
            x_star = x_star.detach().contiguous()
            u_star = u_star.detach().contiguous()
        o, keep_o = opts.to_struct(T, B, nc, C)
        has_f = f is not None and f.numel() > 0
        pf, keep_f = self._problem(x_star[0], C, c, F, f, x_star, u_star)
        if not L.mpc_lqr_kkt_fused_supported(ctypes.byref(pf), ctypes.byref(o)):
            return None
        g = dict(dC=torch.empty(T, B, n, n, **kw), dc=torch.empty(T, B, n, **kw), dF=torch.empty(F.shape, **kw),
                 df=torch.empty(T - 1, B, ns, **kw) if has_f and T > 1 else (torch.empty(0, B, ns, **kw) if has_f else None),
                 dx_init=torch.empty(B, ns, **kw), dx=torch.empty(T, B, ns, **kw), du=torch.empty(T, B, nc, **kw))
        nbytes = int(L.mpc_lqr_kkt_fused_workspace_bytes(ctypes.byref(pf)))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        g["_keep"] = (keep_f, keep_o, ws, dl_dx, dl_du, x_star, u_star, pf, o)
        fn = L.mpc_lqr_kkt_fused
        args = (ctypes.byref(pf), ctypes.byref(o), dl_dx.data_ptr(), dl_du.data_ptr(), g["dC"].data_ptr(), g["dc"].data_ptr(),
                g["dF"].data_ptr(), _ptr(g["df"]) if (has_f and T > 1) else None, g["dx_init"].data_ptr(), g["dx"].data_ptr(),
                g["du"].data_ptr(), None, ws.data_ptr(), nbytes)

        def run(stream=None):
            rc = fn(*args, torch.cuda.current_stream(dev).cuda_stream if stream is None else stream)
            if rc == -1:              # MPC_E_DIMS = misaligned views: kkt_backward's three calls take anything
                return None
            if rc != 0:
                _check(rc, "mpc_lqr_kkt_fused")
            return g
        run.outputs = g
        return run

    # -- (5) pnqp -------------------------------------------------------------------------------
    def pnqp(self, H, q, lower, upper, x_init=None, n_iter=20, want_Hfree=True, want_lu=False):
        dev = _require_device(H, q)
        L = load()
        B, n = H.shape[0], H.shape[1]
        kw = dict(device=dev, dtype=H.dtype)
        H = H.detach().contiguous(); q = q.detach().contiguous()

        def full(v):
            if not torch.is_tensor(v):
                v = torch.full((1,), float(v))
            return v.detach().to(**kw).expand(B, n).contiguous()
        lo, hi = full(lower), full(upper)
        x0 = None if x_init is None else x_init.detach().to(**kw).contiguous()
        x = torch.empty(B, n, **kw)
        If = torch.empty(B, n, device=dev, dtype=torch.uint8)
        iters = torch.empty(B, device=dev, dtype=torch.int32)
        status = torch.empty(B, device=dev, dtype=torch.int32)
        Hfree = torch.empty(B, n, n, **kw) if want_Hfree else None
        # want_lu: the packed LU + 1-based pivots of the last Newton system (torch.linalg.lu_factor's layout) --
        # the factor the kernel computed anyway, instead of a rocSOLVER call on Hfree afterwards
        LU = torch.empty(B, n, n, **kw) if want_lu else None
        piv = torch.empty(B, n, device=dev, dtype=torch.int32) if want_lu else None
        _check(L.mpc_pnqp_lu(_dtype_code(H), B, n, H.data_ptr(), q.data_ptr(), lo.data_ptr(), hi.data_ptr(),
                             _ptr(x0), int(n_iter), x.data_ptr(), If.data_ptr(), iters.data_ptr(),
                             status.data_ptr(), _ptr(Hfree), _ptr(LU), _ptr(piv), _stream(dev)), "mpc_pnqp_lu")
        return dict(x=x, If=If, iters=iters, status=status, Hfree=Hfree, LU=LU, pivots=piv)

    # -- (6) get_traj / get_cost ----------------------------------------------------------------
    def traj_cost(self, x_init, u, F, f, C=None, c=None, want_x=True):
        dev = _require_device(x_init, u, F)
        L = load()
        T, B, nc = u.shape
        ns = x_init.shape[1]
        kw = dict(device=dev, dtype=u.dtype)
        want_cost = C is not None
        if want_cost:
            p, keep = self._problem(x_init, C, c, F, f, None, u)
        else:
            p = Problem()
            p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, _dtype_code(u)
            keep = []
            xi = x_init.detach().contiguous(); keep.append(xi); p.x_init = xi.data_ptr()
            if T > 1:
                Fc, p.F_st, p.F_sb = _block_strided(F.detach(), 2); keep.append(Fc); p.F = Fc.data_ptr()
            if f is not None and f.numel() > 0:
                fc, p.f_st, p.f_sb = _block_strided(f.detach(), 1); keep.append(fc); p.f = fc.data_ptr()
            cu = u.detach().contiguous(); keep.append(cu); p.cur_u = cu.data_ptr()
        x = torch.empty(T, B, ns, **kw) if want_x else None
        cost = torch.empty(B, **kw) if want_cost else None
        _check(L.mpc_traj_cost(ctypes.byref(p), _ptr(x), _ptr(cost), _stream(dev)), "mpc_traj_cost")
        return x, cost

    def env_traj_cost(self, x_init, u, env, C=None, c=None, want_x=True):
        """util.get_traj through a shipped simulator (+ util.get_cost for a QuadCost)."""
        dev = _require_device(x_init, u)
        L = load()
        T, B, nc = u.shape
        ns = x_init.shape[1]
        kw = dict(device=dev, dtype=u.dtype)
        p = Problem()
        p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, _dtype_code(u)
        xi = x_init.detach().contiguous(); p.x_init = xi.data_ptr()
        cu = u.detach().contiguous(); p.cur_u = cu.data_ptr()
        keep = [xi, cu]
        if C is not None:
            Cc, p.C_st, p.C_sb = _block_strided(C.detach(), 2); keep.append(Cc); p.C = Cc.data_ptr()
            cc, p.c_st, p.c_sb = _block_strided(c.detach(), 1); keep.append(cc); p.c = cc.data_ptr()
        e, prm = env.to_struct(u)
        x = torch.empty(T, B, ns, **kw) if want_x else None
        cost = torch.empty(B, **kw) if C is not None else None
        _check(L.mpc_env_traj_cost(ctypes.byref(p), ctypes.byref(e), _ptr(x), _ptr(cost), _stream(dev)),
               "mpc_env_traj_cost")
        return x, cost

    def env_linearize(self, env, x, u, out_F=None, out_f=None):
        """x [N,ns], u [N,1] -> F [N,ns,ns+1], f [N,ns] (closed-form Jacobian of the simulator)."""
        dev = _require_device(x, u)
        L = load()
        N, ns = x.shape
        kw = dict(device=dev, dtype=x.dtype)
        x = x.detach().contiguous(); u = u.detach().contiguous()
        F = torch.empty(N, ns, ns + 1, **kw) if out_F is None else out_F
        f = torch.empty(N, ns, **kw) if out_f is None else out_f
        assert F.is_contiguous() and f.is_contiguous() and F.numel() == N * ns * (ns + 1) and f.numel() == N * ns
        e, prm = env.to_struct(x)
        _check(L.mpc_env_linearize(ctypes.byref(e), _dtype_code(x), N, x.data_ptr(), u.data_ptr(),
                                   F.data_ptr(), f.data_ptr(), _stream(dev)), "mpc_env_linearize")
        return F, f

    # -- (6d) NNDynamics in the kernels --------------------------------------------------------
    def mlp_rollout(self, x_init, C, c, K, k, cur_x, cur_u, old_costs, opts, net, out_x=None, out_u=None):
        """lqr_forward with the network as true_dynamics (mpc/lqr_step.py:164-261): gains K, k and the nominal's
        cost come from a sweep (`lqr_step(..., want_gains=True)`).  `old_costs` MUST be the cost of (cur_x, cur_u) under the
        same (C, c): the kernel decides a trial on J(trial) - J(nominal), both summed from C, c as one difference, and uses
        old_costs only as the offset of the reported costs (include/mpc_lqr.h, mpc_mlp_rollout)."""
        dev = _require_device(x_init, C, c, K, k, cur_x, cur_u, old_costs)
        L = load()
        T, B, nc = cur_u.shape
        ns = x_init.shape[1]
        p, keep = self._problem(x_init, C, c, None, None, cur_x, cur_u)
        o, keep_o = opts.to_struct(T, B, nc, C)
        e, ws, nbytes, keep_e = net.to_struct(C)
        kw = dict(device=dev, dtype=C.dtype)
        res = dict(new_x=torch.empty(T, B, ns, **kw) if out_x is None else out_x,
                   new_u=torch.empty(T, B, nc, **kw) if out_u is None else out_u,
                   costs=torch.empty(B, **kw), old_costs=torch.empty(B, **kw),
                   full_du_norm=torch.empty(B, **kw), alpha_du_norm=torch.empty(B, **kw),
                   alphas=torch.empty(B, **kw), status=torch.zeros(B, device=dev, dtype=torch.int32))
        out = Outputs()
        for name in res:
            setattr(out, name, res[name].data_ptr())
        Kc, kc, oc = K.detach().contiguous(), k.detach().contiguous(), old_costs.detach().contiguous()
        _check(L.mpc_mlp_rollout(ctypes.byref(p), ctypes.byref(o), ctypes.byref(e), Kc.data_ptr(), kc.data_ptr(),
                                 oc.data_ptr(), ctypes.byref(out), ws.data_ptr(), nbytes, _stream(dev)),
               "mpc_mlp_rollout")
        res["_keep"] = (keep, keep_o, keep_e, Kc, kc, oc)
        return res

    def mlp_traj_cost(self, x_init, u, net, C=None, c=None):
        """util.get_traj through the network (+ util.get_cost for a QuadCost), mpc/util.py:102-153."""
        dev = _require_device(x_init, u)
        L = load()
        T, B, nc = u.shape
        ns = x_init.shape[1]
        kw = dict(device=dev, dtype=u.dtype)
        p = Problem()
        p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, _dtype_code(u)
        xi = x_init.detach().contiguous(); p.x_init = xi.data_ptr()
        cu = u.detach().contiguous(); p.cur_u = cu.data_ptr()
        keep = [xi, cu]
        if C is not None:
            Cc, p.C_st, p.C_sb = _block_strided(C.detach(), 2); keep.append(Cc); p.C = Cc.data_ptr()
            cc, p.c_st, p.c_sb = _block_strided(c.detach(), 1); keep.append(cc); p.c = cc.data_ptr()
        e, ws, nbytes, keep_e = net.to_struct(u)
        x = torch.empty(T, B, ns, **kw)
        cost = torch.empty(B, **kw) if C is not None else None
        out = Outputs()
        out.new_x, out.costs = x.data_ptr(), _ptr(cost)
        _check(L.mpc_mlp_rollout(ctypes.byref(p), None, ctypes.byref(e), None, None, None, ctypes.byref(out),
                                 ws.data_ptr(), nbytes, _stream(dev)), "mpc_mlp_rollout")
        return x, cost

    def mlp_linearize(self, net, x, u, out_F=None, out_f=None):
        """x [N,ns], u [N,nc] -> F [N,ns,ns+nc], f [N,ns]: NNDynamics.grad_input + the affine term of
        MPC.linearize_dynamics (mpc/dynamics.py:82-128, mpc/mpc.py:495-512), no [N, hidden, n] intermediates."""
        dev = _require_device(x, u)
        L = load()
        N, ns = x.shape
        nc = u.shape[1]
        kw = dict(device=dev, dtype=x.dtype)
        x = x.detach().contiguous(); u = u.detach().contiguous()
        F = torch.empty(N, ns, ns + nc, **kw) if out_F is None else out_F
        f = torch.empty(N, ns, **kw) if out_f is None else out_f
        assert F.is_contiguous() and f.is_contiguous() and F.numel() == N * ns * (ns + nc) and f.numel() == N * ns
        e, ws, nbytes, keep_e = net.to_struct(x)
        _check(L.mpc_mlp_linearize(ctypes.byref(e), ns, nc, N, x.data_ptr(), u.data_ptr(), F.data_ptr(), f.data_ptr(),
                                   ws.data_ptr(), nbytes, _stream(dev)), "mpc_mlp_linearize")
        return F, f

    def plan_network_iteration(self, x_init, C, c, net, opts, nominals, scratch=None):
        """One iLQR iteration on an NNDynamics network (mpc/mpc.py:245-306 with dx a module: util.get_traj is the previous
        rollout's own new_x, then MPC.linearize_dynamics(ANALYTIC) :495-512, lqr_backward :52-160, lqr_forward through the
        network :164-261) bound ONCE per solve: three C calls on the stream per iteration -- mpc_mlp_linearize, mpc_lqr_step
        (MPC_OPT_SWEEP_ONLY, the fused kernel of the shape), mpc_mlp_rollout -- no allocation, no struct rebuilt, no torch op.
        nominals = ((xa, ua), (xb, ub)): iteration k reads nominals[k % 2] and writes nominals[1 - k % 2] (contiguous buffers).
        -> (run(k, stream=None) -> outputs dict of that parity, (outputs_0, outputs_1), vouch_c()): vouch_c() re-binds the two
        sweeps with MPC_OPT_C_SYMMETRIC once the first sweep has reported C symmetric."""
        dev = _require_device(x_init, C, c, nominals[0][0], nominals[0][1])
        L = load()
        T, B, n = C.shape[0], C.shape[1], C.shape[2]
        ns = x_init.shape[1]
        nc = n - ns
        kw = dict(device=dev, dtype=C.dtype)
        N = (T - 1) * B
        F, f = torch.empty(T - 1, B, ns, n, **kw), torch.empty(T - 1, B, ns, **kw)
        K, k = torch.empty(T, B, nc, ns, **kw), torch.empty(T, B, nc, **kw)
        e, mws, mbytes, keep_e = net.to_struct(C)
        import copy
        so = copy.copy(opts)
        so.sweep_only, so.true_dynamics = True, None
        ro = copy.copy(opts)
        ro.sweep_only, ro.true_dynamics = False, None
        ro_struct, keep_ro = ro.to_struct(T, B, nc, C)
        fl, it = torch.empty(2, 5, B, **kw), torch.zeros(2, 2, B, device=dev, dtype=torch.int32)
        sweeps, rolls, outs, keeps = [], [], [], [F, f, K, k, keep_e, keep_ro, fl, it]
        for j in (0, 1):
            cx, cu = nominals[j]
            ox, ou = nominals[1 - j]
            for t_ in (cx, cu, ox, ou):
                assert t_.is_contiguous() and t_.dtype == C.dtype and t_.device == C.device
            assert tuple(cx.shape) == (T, B, ns) and tuple(cu.shape) == (T, B, nc)
            res = dict(new_x=ox, new_u=ou, costs=fl[j, 0], old_costs=fl[j, 1], full_du_norm=fl[j, 2], alpha_du_norm=fl[j, 3],
                       alphas=fl[j, 4], qp_iters=it[j, 0], status=it[j, 1])
            # the sweep: (x_init, C, c, F, f) at this nominal -> K, k, old_costs, qp_iters, status
            sp, keep_p = self._problem(x_init, C, c, F, f, cx, cu)
            sweeps.append([sp, None, None, keep_p])
            # the rollout through the network: new_x, new_u, costs, full_du_norm, alphas
            rp, keep_r = self._problem(x_init, C, c, None, None, cx, cu)
            rout = Outputs()
            for name in ("new_x", "new_u", "costs", "full_du_norm", "alpha_du_norm", "alphas"):
                setattr(rout, name, res[name].data_ptr())
            # (the rollout's status words would overwrite the sweep's: it reports none the driver reads -- left unbound)
            rolls.append((rp, rout, keep_r, cx.data_ptr(), cu.data_ptr()))
            outs.append(res)
        nbytes = int(L.mpc_lqr_workspace_bytes(ctypes.byref(sweeps[0][0])))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)

        def bind_sweeps(o_):
            st_, keep_s = o_.to_struct(T, B, nc, C)
            st_.flags |= OPT_SWEEP_ONLY
            for j in (0, 1):
                sout = Outputs()
                sout.K, sout.k = K.data_ptr(), k.data_ptr()
                sout.old_costs, sout.qp_iters, sout.status = outs[j]["old_costs"].data_ptr(), outs[j]["qp_iters"].data_ptr(), outs[j]["status"].data_ptr()
                sweeps[j][1], sweeps[j][2] = st_, sout
            keeps.append(keep_s)
        bind_sweeps(so)
        step_fn, lin_fn, roll_fn = L.mpc_lqr_step, L.mpc_mlp_linearize, L.mpc_mlp_rollout
        ep, rop, wsp, mwsp = ctypes.byref(e), ctypes.byref(ro_struct), ws.data_ptr(), mws.data_ptr()
        Fp, fp, Kp, kp = F.data_ptr(), f.data_ptr(), K.data_ptr(), k.data_ptr()

        def run(j, stream=None):
            st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
            sp, so_, sout, _ = sweeps[j]
            rp, rout, _, cxp, cup = rolls[j]
            rc = lin_fn(ep, ns, nc, N, cxp, cup, Fp, fp, mwsp, mbytes, st)
            if rc == 0:
                rc = step_fn(ctypes.byref(sp), ctypes.byref(so_), ctypes.byref(sout), wsp, nbytes, IMPL_AUTO, st)
            if rc == 0:
                rc = roll_fn(ctypes.byref(rp), rop, ep, Kp, kp, outs[j]["old_costs"].data_ptr(), ctypes.byref(rout), mwsp, mbytes, st)
            if rc != 0:
                _check(rc, "network iteration (mpc_mlp_linearize / mpc_lqr_step / mpc_mlp_rollout)")
            return outs[j]

        def vouch_c():
            so2 = copy.copy(so)
            so2.c_symmetric = True
            bind_sweeps(so2)
        run._keep = (keeps, sweeps, rolls, ws, mws, x_init, C, c, nominals)
        return run, tuple(outs), vouch_c

    # -- (7) driver reductions ------------------------------------------------------------------
    writes_host_flags = True

    def select_best(self, first, eps, x, u, costs, du_norm, best, flags=None, status=None, host=None, tag=0):
        """In-place update of best = dict(x,u,costs,full_du_norm); returns the two result words (any_improved int32[1],
        max_du real[1]) as device views, without synchronising.  flags: a `select_flags()` block to reuse.  status: the step's status words -- bit 1 of any_improved then reports whether any of them carries
        ST_C_ASYMMETRIC.  host: 16 bytes of pinned host memory (uint8 tensor) the kernel also stores the two words in, followed
        by the int32 `tag` at byte 4 (what a polling caller waits for)."""
        dev = _require_device(x, u, costs, du_norm)
        L = load()
        T, B, ns = x.shape
        nc = u.shape[2]
        if flags is None:
            flags = self.select_flags(dev, x.dtype)
        any_improved, max_du = flags
        # the two views are bytes [0,4) and [8,8+size) of one 16-byte block (select_flags)
        assert max_du.data_ptr() == any_improved.data_ptr() + 8
        if host is not None:
            assert host.is_pinned() and host.numel() * host.element_size() >= 16
        _check(L.mpc_select_best(_dtype_code(x), B, T, ns, nc, int(bool(first)), float(eps),
                                 x.data_ptr(), u.data_ptr(), costs.data_ptr(), du_norm.data_ptr(),
                                 best["x"].data_ptr(), best["u"].data_ptr(), best["costs"].data_ptr(),
                                 best["full_du_norm"].data_ptr(), any_improved.data_ptr(), _ptr(host), int(tag),
                                 _ptr(status), _stream(dev)), "mpc_select_best")
        return any_improved, max_du

    def plan_select(self, eps, sources, best, flags, host=None):
        """mpc_select_best with everything but (which source, first, tag, status on / off) bound once: `sources` are the
        output dicts of the step plans whose results are selected from (MPC.forward's ping-pong pair).  The call per
        iteration is then one C call -- the general entry above spends ~14 us of host time per call on checks and pointer
        look-ups, which an iLQR iteration of 37 us on the device does not hide.  -> sel(k, first, tag=0, with_status=False,
        stream=None)."""
        x0 = sources[0]["new_x"]
        dev = _require_device(x0, best["x"])
        T, B, ns = x0.shape
        nc = sources[0]["new_u"].shape[2]
        any_improved, max_du = flags
        assert max_du.data_ptr() == any_improved.data_ptr() + 8
        if host is not None:
            assert host.is_pinned() and host.numel() * host.element_size() >= 16
        fn = load().mpc_select_best
        head = (_dtype_code(x0), B, T, ns, nc)
        feps = float(eps)
        tails = []
        for r in sources:
            assert r["new_x"].is_contiguous() and r["new_u"].is_contiguous() and r["new_x"].shape == x0.shape
            tails.append(((r["new_x"].data_ptr(), r["new_u"].data_ptr(), r["costs"].data_ptr(), r["full_du_norm"].data_ptr(),
                           best["x"].data_ptr(), best["u"].data_ptr(), best["costs"].data_ptr(), best["full_du_norm"].data_ptr(),
                           any_improved.data_ptr(), _ptr(host)), r["status"].data_ptr()))
        keep = (sources, best, flags, host)

        def sel(k, first, tag=0, with_status=False, stream=None):
            ptrs, st = tails[k]
            rc = fn(*head, 1 if first else 0, feps, *ptrs, tag, st if with_status else None,
                    torch.cuda.current_stream(dev).cuda_stream if stream is None else stream)
            if rc != 0:
                _check(rc, "mpc_select_best")
        sel._keep = keep
        return sel

    @staticmethod
    def select_flags(device, dtype):
        """The 16-byte device block mpc_select_best reports in, as its two views (int32 word, real maximum)."""
        blk = torch.empty(16, dtype=torch.uint8, device=device)
        return blk[0:4].view(torch.int32), blk[8:8 + torch.empty(0, dtype=dtype).element_size()].view(dtype)


_backend = None


def backend():
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend_for_testing(obj):
    """TEST HOOK ONLY.  tests/ use this to drive the host-side logic (MPC.forward, the autograd
    wiring) on a CPU-only box with an oracle-backed stand-in.  The shipped package never calls it
    and has no CPU implementation of its own."""
    global _backend
    prev = _backend
    _backend = obj
    return prev
