"""Independent formulation of the per-bond loss and gradient (torch fp64 autograd).

TEST INFRASTRUCTURE ONLY (see oracle/ref_numpy.py header; sweep parity unpinned).

Plays the role of the reference's own cross-check between its array engine and
its legacy ITensor engine (test/classification.jl:24): the loss is written as
the *definition* - full left-to-right contraction of the whole MPS with the
bond tensor substituted for sites (lid, rid), no caches, no fused loops - and
the gradient comes from reverse-mode autodiff instead of the hand-derived
phi-tilde/yhat formula (src/legacy_itensor/loss_functions.jl:433-491 is the
same mathematics on ITensors).
"""
from __future__ import annotations

import numpy as np
import torch


def _chain_outputs(W, bt5, lid, phi):
    """yhat (N, C): contract sites 0..lid-1, the bond tensor (s_l,a,s_r,b,c), sites rid+1..T-1."""
    T = len(W)
    N = phi.shape[0]
    left = torch.ones((N, 1), dtype=bt5.dtype)
    for j in range(lid):
        left = torch.einsum("ia,is,ask->ik", left, phi[:, j, :], W[j])
    right = torch.ones((N, 1), dtype=bt5.dtype)
    for j in range(T - 1, lid + 1, -1):
        right = torch.einsum("ib,is,ksb->ik", right, phi[:, j, :], W[j])
    return torch.einsum("ia,is,it,ib,satbc->ic", left, phi[:, lid, :], phi[:, lid + 1, :], right, bt5)


def loss_and_grad(W, bt5, lid, phi, label_index, class_distribution, loss="KLD", train_separate=False):
    """Loss at bond (lid, lid+1) as a function of the bond tensor and its gradient.

    KLD  : (1/N) sum_i -log yhat_i[c_i]^2          (loss_functions.jl:322-379)
           train_separate: sum_c mean_{i in c}       (:383-432)
    MSE  : (1/N) sum_i sum_c 0.5 (yhat_i[c]-delta)^2 (:561-619)
    The reference's gradient is dLoss/dB up to the factor convention of its
    hand-written form: for real tensors d(-log y^2)/dB = -2 phi/y while the
    reference accumulates -phi/y (:367); callers compare against grad/2 for KLD.
    """
    Wt = [torch.tensor(np.ascontiguousarray(t), dtype=torch.float64) if t is not None and t.ndim == 3 else None
          for t in W]
    phit = torch.tensor(phi, dtype=torch.float64)
    b = torch.tensor(np.ascontiguousarray(bt5), dtype=torch.float64, requires_grad=True)
    yhat = _chain_outputs(Wt, b, lid, phit)
    N, C = yhat.shape
    idx = torch.tensor(np.asarray(label_index), dtype=torch.long)
    if loss == "KLD":
        per = -torch.log(yhat[torch.arange(N), idx] ** 2)
        if train_separate:
            cd = torch.tensor(np.asarray(class_distribution), dtype=torch.float64)
            val = (per / cd[idx]).sum()
        else:
            val = per.mean()
    elif loss == "MSE":
        y = torch.zeros((N, C), dtype=torch.float64)
        y[torch.arange(N), idx] = 1.0
        val = (0.5 * (yhat - y) ** 2).sum() / N
    else:
        raise ValueError(loss)
    val.backward()
    return float(val.detach()), b.grad.numpy().copy(), yhat.detach().numpy()
