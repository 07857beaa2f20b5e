"""Simulator dynamics shipped with the reference (mpc/env_dx/): PendulumDx, CartpoleDx.

Besides the differentiable torch `forward` (what the reference has), each module here carries a
description the gfx950 kernels understand (`native_env()`), so `MPC` can linearise it in closed form
(`grad_input`, one kernel over all (T-1)*B points) and roll it out inside the line-search kernel
instead of calling back into Python once per timestep (mpc/lqr_step.py:223-225)."""
