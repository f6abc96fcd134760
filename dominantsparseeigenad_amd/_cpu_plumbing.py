"""Host-tensor plumbing (BASELINE.json configs[0]: "CPU PyTorch (plumbing, no GPU)").

The reference selects its device with a ``device=`` argument (reference Lanczos.py:3, symeig.py:15,71).
When the caller passes CPU tensors / ``device=cpu`` the primitives run these few torch expressions on
the host, exactly as the reference does; this keeps the reference's own CPU unit tests and example
scripts runnable anywhere.  It is NOT a fallback: a CUDA tensor is never routed here -- the CUDA path
goes to libdsea.so or raises (see engine.py / _lib.py).
"""
from __future__ import annotations

import torch


def lanczos_host(apply_A, k, n, dtype, q0, device):
    """reference Lanczos.py:49-77 on host tensors.  Returns (Qk (n,k), alphas, betas)."""
    Qk = torch.zeros((n, k), dtype=dtype, device=device)
    alphas = torch.zeros(k, dtype=dtype, device=device)
    betas = torch.zeros(max(k - 1, 0), dtype=dtype, device=device)
    q = q0 / torch.norm(q0)
    u = apply_A(q)
    a = torch.matmul(q, u)
    Qk[:, 0], alphas[0] = q, a
    b, q_old = 0, None
    for i in range(1, k):
        r = u - a * q if q_old is None else u - a * q - b * q_old
        Qi = Qk[:, :i]
        r = r - torch.matmul(Qi, torch.matmul(Qi.T, r))
        q_old = q
        b = torch.norm(r)
        q = r / b
        u = apply_A(q)
        a = torch.matmul(q, u)
        Qk[:, i], alphas[i], betas[i - 1] = q, a, b
    return Qk, alphas, betas


def cg_host(apply_A, b, x0, eps, cap, info):
    """reference CG.py:24-41 on host tensors (A d evaluated once per iteration)."""
    x = x0
    r = b - apply_A(x)
    rn = torch.norm(r).item()
    it = 0
    if rn >= eps:
        d = r
        Ad = apply_A(d)
        step = torch.matmul(r, r) / torch.matmul(Ad, d)
        for _ in range(cap):
            it += 1
            x = x + step * d
            r_new = r - step * Ad
            rn = torch.norm(r_new).item()
            if rn < eps:
                break
            ratio = torch.matmul(r_new, r_new) / torch.matmul(r, r)
            r = r_new
            d = r + ratio * d
            Ad = apply_A(d)
            step = torch.matmul(r, r) / torch.matmul(Ad, d)
    info.iters, info.resnorm, info.converged = it, rn, rn < eps
    return x
