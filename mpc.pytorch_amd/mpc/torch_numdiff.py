"""Finite-difference gradient / Hessian of a batched scalar- or vector-valued net, all probe points in
one forward call -- mirror of the reference's mpc/torch_numdiff.py (grad :15-28, hess :31-45)."""
import torch


def grad(net, inputs, eps=1e-4):
    """inputs [B,n] -> d net / d inputs, [B,n] (or [B,n,m] for an m-valued net); central differences
    with half-steps eps/2, as the reference."""
    assert inputs.ndimension() == 2
    B, n = inputs.shape
    x = inputs.detach()
    e = 0.5 * eps * torch.eye(n, dtype=x.dtype, device=x.device)
    probes = x.unsqueeze(1) + torch.stack((e, -e)).unsqueeze(1)         # [2, B, n, n]
    fs = net(probes.reshape(2 * B * n, n))
    m = fs.shape[1] if fs.ndimension() > 1 else 1
    fs = fs.reshape(2, B, n, m)
    return ((fs[0] - fs[1]) / eps).squeeze(2)


def hess(net, inputs, eps=1e-4):
    """inputs [B,n] -> second derivatives [B,n,n] (or [B,n,n,m]) by the four-point stencil."""
    assert inputs.ndimension() == 2
    B, n = inputs.shape
    x = inputs.detach()
    e = eps * torch.eye(n, dtype=x.dtype, device=x.device)
    ei, ej = e.unsqueeze(1), e.unsqueeze(0)                             # [n,1,n], [1,n,n]
    shifts = torch.stack((ei + ej, ei - ej, -ei + ej, -ei - ej))         # [4, n, n, n]
    probes = x.view(1, B, 1, 1, n) + shifts.unsqueeze(1)                 # [4, B, n, n, n]
    fs = net(probes.reshape(4 * B * n * n, n))
    m = fs.shape[1] if fs.ndimension() > 1 else 1
    fs = fs.reshape(4, B, n, n, m)
    return ((fs[0] - fs[1] - fs[2] + fs[3]) / (4 * eps * eps)).squeeze(3)
