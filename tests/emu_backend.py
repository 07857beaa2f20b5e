"""TEST INFRASTRUCTURE: run the fused MFMA kernel SOURCE (mpc.pytorch_amd/csrc/lqr_mfma16_body.h)
on the CPU through the 64-fiber wavefront emulator in tests/emu/emu_mfma16.cpp.

This is how the kernel's lane/register layout algebra is parity-tested on a box without a GPU;
on the GPU box the same source runs on the real v_mfma_f32_16x16x4_f32 (tests/test_gpu_parity.py).
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np

from mpc import _native as N

_HERE = os.path.dirname(os.path.abspath(__file__))
_EMU = os.path.join(_HERE, "emu")
_LIB = None


def build(ring2=False, pad=0):
    """ring2: the 12/4 kernel with its 2-slot sweep ring (-DMPC_DPP16_NSTAGE=2, the second compilation of lqr_dpp16.hip) and
    the 32/8 kernel with its 2-slot sweep ring (-DMPC_MFMA40_SWEEP_NSTAGE=2, the second compilation of lqr_mfma40.hip's step kernels).
    pad = 4 | 16: the 32/8 kernel's PADDED instantiation (-DMPC_MFMA40_PAD=4|16 on the two-slot ring, csrc/Makefile) -- and, round 6, the
    12/4 kernel's (-DMPC_DPP16_PAD: any n_state <= 12, n_ctrl <= 4; kernel name "dpp16_pad")."""
    so = os.path.join(_EMU, "libemu_mfma16_pad%d.so" % pad if pad else ("libemu_mfma16_ring2.so" if ring2 else "libemu_mfma16.so"))
    src = os.path.join(_EMU, "emu_mfma16.cpp")
    csrc = os.path.join(_HERE, "..", "mpc.pytorch_amd", "csrc")
    deps = [src] + [os.path.join(csrc, h) for h in ("lqr_mfma16_body.h", "lqr_dpp16_body.h", "lqr_small_math.h",
                                                    "lqr_params.h", "env_dynamics.h", "lqr_tiny_body.h", "lqr_wave1_body.h", "lqr_mfma40_body.h")]
    def stale():
        return not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps)
    if stale():
        # several test processes (pytest -n) may find the library stale at once: one builds under the lock, into a
        # temporary name moved into place, so that nobody ever loads a half-written file
        import fcntl
        with open(so + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                cxx = "/opt/rocm/lib/llvm/bin/clang++"
                if not os.path.exists(cxx):
                    cxx = shutil.which("clang++")
                assert cxx, "the emulator needs clang++ (ext_vector_type)"
                tmp = so + ".tmp%d" % os.getpid()
                subprocess.check_call([cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]
                                      + (["-DMPC_DPP16_NSTAGE=2", "-DMPC_KKT16_NSTAGE=2", "-DMPC_MFMA40_SWEEP_NSTAGE=2"] if (ring2 or pad) else [])
                                      + (["-DMPC_MFMA40_PAD=%d" % pad, "-DMPC_DPP16_PAD", "-DMPC_KF_LDS_BYTES=36864"] if pad else []) + ["-o", tmp, src])
                os.replace(tmp, so)
    return so


def build_all(workers=4):
    """The four emulator libraries side by side (~55 s each from cold, clang++ -O1): what the CPU suite's first emulator test would
    otherwise wait for one after the other."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(lambda kw: build(**kw), [dict(), dict(ring2=True), dict(pad=4), dict(pad=16)]))


_LIB2 = None
_LIBPAD = {}


def lib_pad(g):
    if g not in _LIBPAD:
        _LIBPAD[g] = ctypes.CDLL(build(pad=g))
    return _LIBPAD[g]



def lib_ring2():
    global _LIB2
    if _LIB2 is None:
        _LIB2 = ctypes.CDLL(build(ring2=True))
        _LIB2.emu_lqr_step_dpp16.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options), ctypes.POINTER(N.Outputs)]
    return _LIB2


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.emu_lqr_step_mfma16.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options),
                                             ctypes.POINTER(N.Outputs), ctypes.c_int]
        _LIB.emu_lqr_step_dpp16.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options),
                                            ctypes.POINTER(N.Outputs)]
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p).value


def lqr_step(x_init, C, c, F, f, cur_x, cur_u, u_lower=None, u_upper=None, u_zero_I=None, delta_u=None,
             linesearch_decay=0.2, max_linesearch_iter=10, pnqp_iter=20, force_general=False, dma_late=False,
             kernel="mfma16", dtype=np.float32, env=None, nominal_on_dynamics=False, c_symmetric=False, qp_start=None):
    """Same signature as oracle.lqr_oracle.lqr_step.  Returns the kernel's outputs.  The fused
    kernels are float32; kernel="tiny" (lane-per-problem body, n_ctrl = 1) also runs in float64 and
    takes env = (kind, params, dt, u_max): a shipped simulator as the rollout's true dynamics."""
    f32 = np.dtype(dtype).type
    assert f32 == np.float32 or kernel in ("tiny", "mfma16")      # (mfma16: its float64 instantiation, round 5)
    C = np.ascontiguousarray(C, f32); c = np.ascontiguousarray(c, f32)
    x_init = np.ascontiguousarray(x_init, f32)
    T, B, n, _ = C.shape
    ns = x_init.shape[1]
    nc = n - ns
    F = np.ascontiguousarray(F, f32) if T > 1 else np.zeros((0, B, ns, n), f32)
    f = None if (f is None or np.asarray(f).size == 0) else np.ascontiguousarray(f, f32)
    cur_x = np.ascontiguousarray(cur_x, f32); cur_u = np.ascontiguousarray(cur_u, f32)
    p = N.Problem()
    p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, (N.MPC_F32 if f32 == np.float32 else N.MPC_F64)
    p.x_init = _ptr(x_init)
    p.C, p.C_st, p.C_sb = _ptr(C), B * n * n, n * n
    p.c, p.c_st, p.c_sb = _ptr(c), B * n, n
    if T > 1:
        p.F, p.F_st, p.F_sb = _ptr(F), B * ns * n, ns * n
    if f is not None:
        p.f, p.f_st, p.f_sb = _ptr(f), B * ns, ns
    p.cur_x, p.cur_u = _ptr(cur_x), _ptr(cur_u)
    o = N.Options()
    o.max_linesearch_iter = int(max_linesearch_iter)
    o.linesearch_decay = float(linesearch_decay)
    o.delta_u = float("nan") if delta_u is None else float(delta_u)
    o.pnqp_iter = int(pnqp_iter)
    o.flags = (N.OPT_NOMINAL_ON_DYNAMICS if nominal_on_dynamics else 0) | (N.OPT_C_SYMMETRIC if c_symmetric else 0)
    keep = []
    if u_lower is None:
        o.bound_mode = N.BOUND_NONE
    elif isinstance(u_lower, float) and isinstance(u_upper, float):
        o.bound_mode, o.lo_s, o.hi_s = N.BOUND_SCALAR, u_lower, u_upper
    else:
        lo = np.ascontiguousarray(np.broadcast_to(np.asarray(u_lower, f32), (T, B, nc)))
        hi = np.ascontiguousarray(np.broadcast_to(np.asarray(u_upper, f32), (T, B, nc)))
        keep += [lo, hi]
        o.bound_mode, o.lo, o.hi = N.BOUND_TENSOR, _ptr(lo), _ptr(hi)
    if u_zero_I is not None:
        zm = np.ascontiguousarray((np.asarray(u_zero_I) != 0).astype(np.uint8))
        keep.append(zm)
        o.zero_mask = _ptr(zm)
    if qp_start is not None:
        # mpc_lqr_options.qp_start: any array that broadcasts to [T,B,nc]; element strides of the T and B axes (0 where broadcast)
        qs = np.asarray(qp_start, f32)
        full = np.broadcast_to(qs, (T, B, nc))
        if full.strides[2] not in (4, 0) or any(st % 16 for st in full.strides[:2]):
            full = np.ascontiguousarray(full)
        if full.strides[2] == 0:
            full = np.ascontiguousarray(full)
        keep.append(full); keep.append(qs)
        o.qp_start, o.qp_start_st, o.qp_start_sb = full.ctypes.data, full.strides[0] // 4, full.strides[1] // 4
    res = dict(new_x=np.full((T, B, ns), np.nan, f32), new_u=np.full((T, B, nc), np.nan, f32),
               costs=np.empty(B, f32), old_costs=np.empty(B, f32), full_du_norm=np.empty(B, f32),
               alpha_du_norm=np.empty(B, f32), alphas=np.empty(B, f32),
               qp_iters=np.zeros(B, np.int32), status=np.zeros(B, np.int32),
               K=np.full((T, B, nc, ns), np.nan, f32), k=np.full((T, B, nc), np.nan, f32))
    out = N.Outputs()
    for key, arr in res.items():
        setattr(out, key, _ptr(arr))
    lib().emu_set_dma_late(int(bool(dma_late)))
    if kernel == "dpp16_ring2":
        lib_ring2().emu_set_dma_late(int(bool(dma_late)))
    if env is not None:
        e = N.EnvDynamics()
        prm = np.ascontiguousarray(env[1], f32)
        keep.append(prm)
        e.kind, e.params, e.dt, e.u_max = int(env[0]), _ptr(prm), float(env[2]), float(env[3])
        e.linearize = int(len(env) > 4 and bool(env[4]))
        keep.append(e)
        o.true_dynamics = ctypes.pointer(e)
    if kernel in ("mfma40_sweep", "mfma40", "mfma40_ring2", "mfma40_pad4", "mfma40_pad16", "mfma40_pad4_sweep", "mfma40_pad16_sweep"):
        # ("mfma40_ring2": the step kernels' second compilation, two sweep slots instead of three; "mfma40_pad4 / _pad16": the
        # padded instantiation for any n_state <= 32, n_ctrl <= 8, dword / 16-byte gathers)
        L40 = lib_pad(4) if "pad4" in kernel else (lib_pad(16) if "pad16" in kernel else (lib_ring2() if kernel == "mfma40_ring2" else lib()))
        L40.emu_set_dma_late(int(bool(dma_late)))
        L40.emu_mfma40_full(int(not kernel.endswith("_sweep")))
        fn = L40.emu_lqr_sweep_mfma40
        fn.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options), ctypes.POINTER(N.Outputs)]
        rc = fn(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out))
    elif kernel == "tiny":
        fn = lib().emu_lqr_step_tiny
        fn.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options), ctypes.POINTER(N.Outputs)]
        rc = fn(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out))
    elif kernel == "wave1":
        fn = lib().emu_lqr_step_wave1
        fn.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options), ctypes.POINTER(N.Outputs)]
        rc = fn(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out))
    elif kernel == "dpp16_pad":
        Lp = lib_pad(4)
        Lp.emu_set_dma_late(int(bool(dma_late)))
        Lp.emu_lqr_step_dpp16.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options), ctypes.POINTER(N.Outputs)]
        rc = Lp.emu_lqr_step_dpp16(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out))
    elif kernel == "dpp16":
        rc = lib().emu_lqr_step_dpp16(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out))
    elif kernel == "dpp16_ring2":
        rc = lib_ring2().emu_lqr_step_dpp16(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out))
    elif f32 == np.float64:
        fn = lib().emu_lqr_step_mfma16_f64
        fn.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options), ctypes.POINTER(N.Outputs), ctypes.c_int]
        rc = fn(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), int(force_general))
    else:
        rc = lib().emu_lqr_step_mfma16(ctypes.byref(p), ctypes.byref(o), ctypes.byref(out), int(force_general))
    assert rc == 0, rc
    return res


def kkt_grads(C, c, F, f, x_star, u_star, dx, du, dl_dx, dma_late=False, ring2=False):
    """The closed-form part of LQRStepFn.backward (mpc/lqr_step.py:346-404) through the emulated
    4-problems-per-wave kernel; n_state = 12, n_ctrl = 4, float32."""
    f32 = np.float32
    cast = lambda a: np.ascontiguousarray(a, f32)
    C, c, F, x_star, u_star, dx, du, dl_dx = map(cast, (C, c, F, x_star, u_star, dx, du, dl_dx))
    T, B, n, _ = C.shape
    ns, nc = 12, 4
    p = N.Problem()
    p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, N.MPC_F32
    x0 = np.zeros((B, ns), f32)
    p.x_init = _ptr(x0)
    p.C, p.C_st, p.C_sb = _ptr(C), B * n * n, n * n
    p.c, p.c_st, p.c_sb = _ptr(c), B * n, n
    if T > 1:
        p.F, p.F_st, p.F_sb = _ptr(F), B * ns * n, ns * n
    p.cur_x, p.cur_u = _ptr(x_star), _ptr(u_star)
    has_f = f is not None and np.asarray(f).size > 0
    out = dict(dC=np.full((T, B, n, n), np.nan, f32), dc=np.full((T, B, n), np.nan, f32),
               dF=np.zeros((max(T - 1, 0), B, ns, n), f32), df=np.full((max(T - 1, 0), B, ns), np.nan, f32) if has_f else None,
               dx_init=np.full((B, ns), np.nan, f32))
    L = lib_ring2() if ring2 else lib()          # ring2: the 2-slot ring the library's KKT kernel is built with
    L.emu_set_dma_late(int(bool(dma_late)))
    vp = ctypes.c_void_p
    L.emu_kkt_dpp16.argtypes = [ctypes.POINTER(N.Problem)] + [vp] * 8
    rc = L.emu_kkt_dpp16(ctypes.byref(p), _ptr(dx), _ptr(du), _ptr(dl_dx), _ptr(out["dC"]), _ptr(out["dc"]),
                         _ptr(out["dF"]), _ptr(out["df"]), _ptr(out["dx_init"]))
    assert rc == 0, rc
    return out


def kkt_fused(C, c, F, f, x_star, u_star, dl_dx, dl_du, u_lower=None, u_upper=None, dma_late=False, ring2=True,
              linesearch_decay=0.2, max_linesearch_iter=10, kernel="dpp16"):
    """ALL of LQRStepFn.backward (mpc/lqr_step.py:312-407) through the emulated fused kernel (kkt_fused_wave,
    lqr_dpp16_body.h): n_state = 12, n_ctrl = 4, float32.  Returns dC, dc, dF, df, dx_init and the KKT
    solve's own (dx, du).  kernel="dpp16_pad": the padded instantiation, any n_state <= 12, n_ctrl <= 4 (the library's
    lqr_dpp16_padkkt.o: two sweep slots, the second pass on 36 KiB)."""
    f32 = np.float32
    cast = lambda a: np.ascontiguousarray(a, f32)
    C, c, F, x_star, u_star, dl_dx, dl_du = map(cast, (C, c, F, x_star, u_star, dl_dx, dl_du))
    T, B, n, _ = C.shape
    ns, nc = (x_star.shape[2], n - x_star.shape[2]) if kernel == "dpp16_pad" else (12, 4)
    p = N.Problem()
    p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, N.MPC_F32
    x0 = np.zeros((B, ns), f32)
    p.x_init = _ptr(x0)
    p.C, p.C_st, p.C_sb = _ptr(C), B * n * n, n * n
    p.c, p.c_st, p.c_sb = _ptr(c), B * n, n
    if T > 1:
        p.F, p.F_st, p.F_sb = _ptr(F), B * ns * n, ns * n
    p.cur_x, p.cur_u = _ptr(x_star), _ptr(u_star)
    o = N.Options()
    o.max_linesearch_iter, o.linesearch_decay, o.delta_u, o.pnqp_iter = int(max_linesearch_iter), float(linesearch_decay), float("nan"), 20
    keep = []
    if u_lower is None:
        o.bound_mode = N.BOUND_NONE
    elif isinstance(u_lower, (float, int)):
        o.bound_mode, o.lo_s, o.hi_s = N.BOUND_SCALAR, float(u_lower), float(u_upper)
    else:
        lo = np.ascontiguousarray(np.broadcast_to(u_lower, (T, B, nc)), f32)
        hi = np.ascontiguousarray(np.broadcast_to(u_upper, (T, B, nc)), f32)
        keep += [lo, hi]
        o.bound_mode, o.lo, o.hi = N.BOUND_TENSOR, _ptr(lo), _ptr(hi)
    has_f = f is not None and np.asarray(f).size > 0
    out = dict(dC=np.full((T, B, n, n), np.nan, f32), dc=np.full((T, B, n), np.nan, f32),
               dF=np.full((max(T - 1, 0), B, ns, n), np.nan, f32), df=np.full((max(T - 1, 0), B, ns), np.nan, f32) if has_f else None,
               dx_init=np.full((B, ns), np.nan, f32), dx=np.full((T, B, ns), np.nan, f32), du=np.full((T, B, nc), np.nan, f32),
               status=np.zeros(B, np.int32))
    L = lib_pad(4) if kernel == "dpp16_pad" else (lib_ring2() if ring2 else lib())
    L.emu_set_dma_late(int(bool(dma_late)))
    vp = ctypes.c_void_p
    L.emu_kkt_fused.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options)] + [vp] * 10
    rc = L.emu_kkt_fused(ctypes.byref(p), ctypes.byref(o), _ptr(dl_dx), _ptr(dl_du), _ptr(out["dC"]), _ptr(out["dc"]),
                         _ptr(out["dF"]), _ptr(out["df"]), _ptr(out["dx_init"]), _ptr(out["dx"]), _ptr(out["du"]), _ptr(out["status"]))
    assert rc == 0, rc
    return out


def kkt_fused_mfma40(C, c, F, f, x_star, u_star, dl_dx, dl_du, u_lower=None, u_upper=None, dma_late=False, sweep3=True, pad=0):
    """LQRStepFn.backward (mpc/lqr_step.py:312-407) at n_state = 32, n_ctrl = 8 through the emulated fused kernel
    (kkt_fused_wave, lqr_mfma40_body.h): dx, du, dx_init, df and the two costates from the kernel; dC, dc, dF are then the
    outer products kkt_outer_kernel (kkt_wave.hip) forms from exactly those vectors (:346-353, :387-396), here in numpy."""
    f32 = np.float32
    cast = lambda a: np.ascontiguousarray(a, f32)
    C, c, x_star, u_star, dl_dx, dl_du = map(cast, (C, c, x_star, u_star, dl_dx, dl_du))
    T, B, n, _ = C.shape
    # pad = 4: the padded instantiation (any n_state <= 32, n_ctrl <= 8; the library's lqr_mfma40_padkkt.o: dword gathers, two-slot sweep ring)
    ns, nc = (x_star.shape[2], n - x_star.shape[2]) if pad else (32, 8)
    p = N.Problem()
    p.B, p.T, p.ns, p.nc, p.dtype = B, T, ns, nc, N.MPC_F32
    x0 = np.zeros((B, ns), f32)
    p.x_init = _ptr(x0)
    p.C, p.C_st, p.C_sb = _ptr(C), B * n * n, n * n
    p.c, p.c_st, p.c_sb = _ptr(c), B * n, n
    if T > 1:
        F = cast(F)
        p.F, p.F_st, p.F_sb = _ptr(F), B * ns * n, ns * n
    p.cur_x, p.cur_u = _ptr(x_star), _ptr(u_star)
    o = N.Options()
    o.max_linesearch_iter, o.linesearch_decay, o.delta_u, o.pnqp_iter = 10, 0.2, float("nan"), 20
    keep = []
    if u_lower is None:
        o.bound_mode = N.BOUND_NONE
    elif isinstance(u_lower, (float, int)):
        o.bound_mode, o.lo_s, o.hi_s = N.BOUND_SCALAR, float(u_lower), float(u_upper)
    else:
        lo = np.ascontiguousarray(np.broadcast_to(u_lower, (T, B, nc)), f32)
        hi = np.ascontiguousarray(np.broadcast_to(u_upper, (T, B, nc)), f32)
        keep += [lo, hi]
        o.bound_mode, o.lo, o.hi = N.BOUND_TENSOR, _ptr(lo), _ptr(hi)
    has_f = f is not None and np.asarray(f).size > 0
    out = dict(dF=np.full((max(T - 1, 0), B, ns, n), np.nan, f32), df=np.full((max(T - 1, 0), B, ns), np.nan, f32) if has_f else None,
               dx_init=np.full((B, ns), np.nan, f32), dx=np.full((T, B, ns), np.nan, f32), du=np.full((T, B, nc), np.nan, f32),
               status=np.zeros(B, np.int32))
    L = lib_pad(pad) if pad else (lib() if sweep3 else lib_ring2())          # (the library builds this kernel with the 3-slot sweep ring)
    L.emu_set_dma_late(int(bool(dma_late)))
    vp = ctypes.c_void_p
    L.emu_kkt_fused_mfma40.argtypes = [ctypes.POINTER(N.Problem), ctypes.POINTER(N.Options)] + [vp] * 8
    rc = L.emu_kkt_fused_mfma40(ctypes.byref(p), ctypes.byref(o), _ptr(dl_dx), _ptr(dl_du), _ptr(out["dF"]), _ptr(out["df"]),
                                _ptr(out["dx_init"]), _ptr(out["dx"]), _ptr(out["du"]), _ptr(out["status"]))
    assert rc == 0, rc
    tau = np.concatenate((x_star, u_star), 2)
    dtau = np.concatenate((out["dx"], out["du"]), 2)
    out["dC"] = -0.5 * (dtau[..., :, None] * tau[..., None, :] + tau[..., :, None] * dtau[..., None, :])
    out["dc"] = -dtau
    if T > 1:
        park = out["dF"].reshape(T - 1, B, ns * n)
        lam1, dlam1 = park[..., :ns].copy(), park[..., ns:2 * ns].copy()
        out["lam1"], out["dlam1"] = lam1, dlam1
        out["dF"] = -(dlam1[..., :, None] * tau[:-1, :, None, :] + lam1[..., :, None] * dtau[:-1, :, None, :])
    return out


def env_linearize(kind, params, dt, u_max, x, u, dtype=np.float64):
    """mpc.pytorch_amd/csrc/env_dynamics.h compiled for the host: next state, F, f at N points."""
    sfx = "f64" if dtype == np.float64 else "f32"
    x = np.ascontiguousarray(x, dtype); u = np.ascontiguousarray(u, dtype).reshape(-1)
    params = np.ascontiguousarray(params, dtype)
    Np, ns = x.shape
    nxt = np.empty((Np, ns), dtype); F = np.empty((Np, ns, ns + 1), dtype); f = np.empty((Np, ns), dtype)
    fn = getattr(lib(), "emu_env_linearize_" + sfx)
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_long] + [ctypes.c_void_p] * 5
    fn.restype = None
    fn(int(kind), _ptr(params), float(dt), float(u_max), Np, _ptr(x), _ptr(u), _ptr(nxt), _ptr(F), _ptr(f))
    return nxt, F, f
