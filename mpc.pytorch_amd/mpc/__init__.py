"""MI355X-native batched differentiable LQR / iLQR solver.

Drop-in for the `mpc` package of locuslab/mpc.pytorch on its LQR hot path:

    from mpc import mpc
    from mpc.mpc import QuadCost, LinDx
    x, u, costs = mpc.MPC(n_state, n_ctrl, T, u_lower=..., u_upper=...)(x_init, QuadCost(C, c), LinDx(F, f))

`mpc.MPC`, `QuadCost`, `LinDx`, `GradMethods`, `mpc.lqr_step.LQRStep`, `mpc.pnqp.pnqp` and
`mpc.util.get_traj/get_cost` keep the reference's names, argument meaning and error behaviour; the
computation behind them is the hand-written gfx950 library libmpc_lqr_hip.so (C ABI:
include/mpc_lqr.h).  ROCm device tensors only -- there is no CPU fallback.
"""
__version__ = "0.1.0"
