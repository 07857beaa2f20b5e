#!/usr/bin/env python3
"""Round 6: where should the box QP of timestep t START?  
Harvests the n_ctrl x n_ctrl box QPs of the Riccati sweep over five iLQR iterations of bench.py's box-constrained problem and counts
pnqp trips (the fused kernels' accounting: the confirming trip counts) for several starts -- all reach the same minimiser, the QP
is strictly convex.  Float64 numpy, no GPU; the oracle (test infrastructure) steps the solve from iteration to iteration.
    python tools/qp_start_study.py B n_state n_ctrl T        e.g. 256 12 4 50   /   64 32 8 64
columns: per-problem mean / mean of the max over groups of four problems (what a 12/4 wavefront pays)
  ref    the reference's start, k of timestep t+1 (mpc/lqr_step.py:137,141)      unc   clamp(-Quu^-1 qu) at every timestep (round 6)
  zero   zeros      shift  the previous ITERATION's solution in the new delta space (u_old + k_old - u_new)     prevk  k_old unshifted
  unc+1  unc, then one re-solve on its guessed free set      exact  the solution itself (the floor: one trip)
Log: profiles/r06_qp_start_study.log"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mpc.pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from mpc import _native
import oracle_backend
_native.set_backend_for_testing(oracle_backend.OracleBackend())
from oracle import lqr_oracle as O

def pnqp(H, q, lb, ub, x0, max_iter=20):
    """batched over N: returns x, trips (1 + index of the trip that ended), free set"""
    N = H.shape[0]
    x = np.clip(x0, lb, ub)
    trips = np.zeros(N, int); done = np.zeros(N, bool)
    mprev = -np.ones((N,H.shape[1])); full = np.zeros(N, bool)
    I4 = np.eye(H.shape[1])
    for it in range(max_iter):
        g = np.einsum('nij,nj->ni', H, x) + q
        clamped = ((x == lb) & (g > 0)) | ((x == ub) & (g < 0))
        fr = (~clamped).astype(float)
        same = (np.abs(fr - mprev).sum(1) == 0)
        conf = same & full & ~done
        trips[conf] = it + 1; done |= conf
        if done.all(): break
        Hf = H * fr[:,:,None] * fr[:,None,:] + (1 - fr)[:,:,None] * I4
        dx = -np.linalg.solve(Hf, (fr * g)[:,:,None])[:,:,0] * fr
        small = (np.sqrt((dx**2).sum(1)) < 1e-4) & ~done
        trips[small] = it + 1; done |= small
        mprev = np.where(done[:,None] & ~small[:,None] & ~conf[:,None], mprev, fr)
        xn = x + dx
        xc = np.clip(xn, lb, ub)
        inside = (np.abs(xc - xn).sum(1) == 0)
        full = inside
        # armijo for projected steps
        need = ~inside & ~done
        alpha = np.ones(N)
        for cnt in range(10):
            xt = np.clip(x + alpha[:,None]*dx, lb, ub)
            d = xt - x
            den = -(g*d).sum(1); dhd = np.einsum('ni,nij,nj->n', d, H, d)
            arm = (den - 0.5*dhd) / np.where(den == 0, 1, den)
            bad = need & (arm <= 0.1)
            xc = np.where((need & ~bad)[:,None] | (cnt==0), np.where(need[:,None], xt, xc), xc) if cnt == 0 else np.where((need & ~bad & (alpha<1))[:,None], xt, xc)
            if not bad.any(): break
            # those still bad shrink
            xc = np.where(bad[:,None], xt, xc)
            alpha = np.where(bad, alpha*0.1, alpha)
            need = bad
        x = np.where(done[:,None], x, xc)
    trips[~done] = max_iter
    return x, trips, fr

def sweep(h, x, u, lo, hi, strategies, prev=None):
    """backward Riccati with box QPs; returns dict strategy -> trips [T,B], the record (k solutions)"""
    T, B = u.shape[:2]; nc = u.shape[2]; ns = x.shape[2]
    C, c, F = h["C"], h["c"], h["F"]
    tau = np.concatenate((x, u), 2)
    cb = np.einsum('tbij,tbj->tbi', C, tau) + c
    V = np.zeros((B, ns, ns)); v = np.zeros((B, ns))
    ks = np.zeros((T, B, nc)); out = {s: np.zeros((T, B), int) for s in strategies}
    knext = None
    for t in range(T-1, -1, -1):
        if t == T-1:
            Q = C[t].copy(); q = cb[t].copy()
        else:
            Ft = F[t]
            Q = C[t] + np.einsum('bji,bjk,bkl->bil', Ft, V, Ft)
            q = cb[t] + np.einsum('bji,bj->bi', Ft, v)
        Quu, Qux, qu = Q[:, ns:, ns:], Q[:, ns:, :ns], q[:, ns:]
        lb, ub = lo - u[t], hi - u[t]
        xunc = -np.linalg.solve(Quu, qu[:,:,None])[:,:,0]
        sol = None
        for s in strategies:
            if s == "ref":
                x0 = xunc if knext is None else knext
            elif s == "unc":
                x0 = xunc
            elif s == "zero":
                x0 = np.zeros_like(xunc)
            elif s == "shift":
                x0 = xunc if prev is None else prev["u"][t] + prev["k"][t] - u[t]
            elif s == "prevk":
                x0 = xunc if prev is None else prev["k"][t]
            elif s == "unc+1":   # the unconstrained minimiser clamped, then ONE re-solve on its free set (the clamped ones fixed at their bounds)
                xc = np.clip(xunc, lb, ub); fr = (xc == xunc).astype(float)
                Hf = Quu * fr[:,:,None]*fr[:,None,:] + (1-fr)[:,:,None]*np.eye(nc)
                rhs = fr * (qu + np.einsum('bij,bj->bi', Quu, (1-fr)*xc))
                x0 = np.where(fr > 0, -np.linalg.solve(Hf, rhs[:,:,None])[:,:,0], xc)
            elif s == "exact":
                x0 = sol
            xs, tr, fr = pnqp(Quu, qu, lb, ub, x0)
            out[s][t] = tr
            if sol is None: sol, frs = xs, fr
        ks[t] = sol; knext = sol
        fr = frs
        # K on the free set
        Hf = Quu * fr[:,:,None]*fr[:,None,:] + (1-fr)[:,:,None]*np.eye(nc)
        K = -np.linalg.solve(Hf, fr[:,:,None]*Qux) * fr[:,:,None]
        k = sol
        Qxx, Qxu, qx = Q[:, :ns, :ns], Q[:, :ns, ns:], q[:, :ns]
        V = Qxx + Qxu @ K + K.transpose(0,2,1) @ Qux + K.transpose(0,2,1) @ Quu @ K
        v = qx + np.einsum('bij,bj->bi', Qxu, k) + np.einsum('bji,bj->bi', K, qu) + np.einsum('bji,bjk,bk->bi', K, Quu, k)
    return out, ks

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NS=int(sys.argv[2]); NC=int(sys.argv[3]); TT=int(sys.argv[4])
p = bench.make_problem(NS, NC, TT, B, torch.float32, "cpu", seed=5, u_scale=0.3, clamp=1.0)
h = {k: v.numpy().astype(np.float64) for k, v in p.items()}
x, u = h["cur_x"], h["cur_u"]
strategies = ["ref", "unc", "zero", "shift", "prevk", "unc+1", "exact"]
prev = None
for it in range(5):
    out, ks = sweep(h, x, u, -1.0, 1.0, strategies, prev)
    r = O.lqr_step(h["x_init"], h["C"], h["c"], h["F"], h["f"], x, u, -1.0, 1.0, lockstep=False)
    # wave-level: max over groups of 4 problems
    line = "iter %d  oracle qp/t %.2f |" % (it, float(np.mean(r["n_qp_iter"])) / TT)
    for s in strategies:
        tr = out[s]
        wave = tr.reshape(TT, B // 4, 4).max(2)
        line += " %s %.2f/%.2f" % (s, tr.mean(), wave.mean())
    print(line, flush=True)
    prev = dict(u=u, k=ks)
    x, u = r["new_x"], r["new_u"]
