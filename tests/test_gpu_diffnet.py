"""Diffnet++ diffusion layers on the HIP kernels against the fp64 CPU restatement
(oracle/diffnet_oracle.py; parity unpinned — the TensorFlow reference cannot run in this image): scores, loss and the
gradient of EVERY parameter (embeddings, per-edge attention parameters through softmax + SpMM + SDDMM, attention
MLPs), then a few Adam steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _toy(rng, U, I):
    su = rng.integers(0, U, 6 * U)
    sv = rng.integers(0, U, 6 * U)
    ru = np.r_[rng.integers(0, U, 9 * U), np.zeros(150, np.int64)]          # user 0: a row of > 128 consumed items
    ri = np.r_[rng.integers(0, I, 9 * U), rng.choice(I, 150, replace=False)]
    return (su, sv), (ru, ri)


@pytest.mark.parametrize("H", [64, 32])
def test_diffnet_forward_backward_vs_restatement(H):
    from oracle import diffnet_oracle as DO
    from spex_amd.diffnet import DiffnetPlusPlus, LearnedGraph, loss_fn, pairs_to_csr
    rng = np.random.default_rng(H)
    U, I = 220, 300
    (su, sv), (ru, ri) = _toy(rng, U, I)
    csr = {"social": pairs_to_csr(su, sv, U, U), "consumed": pairs_to_csr(ru, ri, U, I),
           "customer": pairs_to_csr(ri, ru, I, U)}
    shapes = {"social": (U, U), "consumed": (U, I), "customer": (I, U)}
    torch.manual_seed(0)
    graphs = {k: LearnedGraph(*csr[k], n_cols=shapes[k][1]) for k in csr}
    model = DiffnetPlusPlus(U, I, H, graphs["social"], graphs["consumed"], graphs["customer"]).cuda()
    with torch.no_grad():                                   # larger than the 0.01 init so that every path matters
        model.user_embedding.mul_(30.0)
        model.item_embedding.mul_(30.0)
    B = 256
    users, items = rng.integers(0, U, B), rng.integers(0, I, B)
    labels = (rng.random(B) < 0.3).astype(np.float32)
    score, lab = model(users.reshape(-1, 1), items.reshape(-1, 1), labels.reshape(-1, 1), 0)
    loss = loss_fn(score, lab)
    loss.backward()

    p64 = {k: v.detach().double().cpu().requires_grad_() for k, v in model.state_dict().items()}
    pats = {}
    for k, (rowptr, col, _) in csr.items():
        rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
        pats[k] = (torch.from_numpy(rows), torch.from_numpy(col.astype(np.int64)), shapes[k])
    ref_score = DO.diffnet_forward(p64, pats, torch.from_numpy(users), torch.from_numpy(items))
    ref_loss = DO.loss(ref_score, torch.from_numpy(labels).double())
    ref_loss.backward()
    assert (score.detach().double().cpu() - ref_score.detach()).abs().max().item() <= 1e-5 * max(1.0, ref_score.abs().max().item())
    assert abs(loss.item() - ref_loss.item()) <= 1e-6
    # the softmax is shift-invariant, so e.g. the bias of a low-level attention Dense has a gradient that is a sum of
    # cancelling per-edge terms (~1e-8 out of terms of ~1e-5): allow fp32 noise relative to the largest gradient
    g_max = max(p64[n].grad.abs().max().item() for n, _ in model.named_parameters())
    for name, prm in model.named_parameters():
        want = p64[name].grad
        assert want is not None, name
        got = prm.grad.double().cpu()
        scale = max(want.abs().max().item(), 1e-12)
        assert (got - want).abs().max().item() <= 2e-5 * scale + 1e-7 * g_max, (name, (got - want).abs().max().item(), scale)
    assert model.snii1.grad.abs().max().item() > 0 and model.icii2.grad.abs().max().item() > 0

    # flag 1 = inference scores; a few optimiser steps reduce the loss
    with torch.no_grad():
        s1 = model(users, items, None, 1)
    assert torch.allclose(s1, score.detach(), rtol=0, atol=1e-6)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    first = None
    for _ in range(15):
        opt.zero_grad()
        sc, lb = model(users, items, labels, 0)
        l = loss_fn(sc, lb)
        l.backward()
        opt.step()
        first = first if first is not None else l.item()
    assert l.item() < first
