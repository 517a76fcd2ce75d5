"""Oracle vs the reference's own outputs (tests/golden/ref_*.npz, made by oracle/make_golden.py)."""
import os
import numpy as np
import torch
import torch.nn.functional as F
from oracle import model as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _pair(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return F.normalize(torch.randn(n, 512, generator=g)), F.normalize(torch.randn(n, 512, generator=g))


def test_loss_known_answers(golden_dir):
    d = _load(golden_dir, "ref_loss.npz")
    for n in (4, 64, 512):
        v, a = _pair(n)
        v.requires_grad_(); a.requires_grad_()
        loss = O.TripletLoss(0.2)(v, a)
        loss.backward()
        assert abs(loss.item() - float(d[f"loss_{n}"])) < 1e-6
        assert abs(v.grad.norm().item() - float(d[f"dVnorm_{n}"])) < 1e-6
        if n <= 64:
            np.testing.assert_allclose(v.detach().numpy(), d[f"V_{n}"], atol=0)
            np.testing.assert_allclose(v.grad.numpy(), d[f"dV_{n}"], atol=1e-7)
            np.testing.assert_allclose(a.grad.numpy(), d[f"dA_{n}"], atol=1e-7)
    # SURVEY 8c known answers
    assert abs(float(d["loss_4"]) - 0.28413501) < 1e-7
    assert abs(float(d["loss_64"]) - 0.40155444) < 1e-7
    assert abs(float(d["loss_512"]) - 0.39339843) < 1e-7


def test_loss_unnormalised_and_margins(golden_dir):
    d = _load(golden_dir, "ref_loss.npz")
    v = torch.tensor(d["rawV"], requires_grad=True)
    a = torch.tensor(d["rawA"], requires_grad=True)
    np.testing.assert_allclose(O.cosine_matrix(v, a).detach().numpy(), d["raw_cos"], atol=1e-6)
    loss = O.TripletLoss(0.2)(v, a)
    loss.backward()
    assert abs(loss.item() - float(d["raw_loss"])) < 1e-6
    np.testing.assert_allclose(v.grad.numpy(), d["raw_dV"], atol=1e-7)
    np.testing.assert_allclose(a.grad.numpy(), d["raw_dA"], atol=1e-7)
    v4, a4 = _pair(4)
    for m, ref in zip(d["margins"], d["margin_losses"]):
        assert abs(O.TripletLoss(float(m))(v4, a4).item() - ref) < 1e-6


def test_triplet_accuracy(golden_dir):
    d = _load(golden_dir, "ref_metrics.npz")
    anc, pos, neg = (torch.tensor(d[k]) for k in ("anchor", "positive", "negative"))
    acc = O.triplet_accuracy(anc, pos, neg)
    np.testing.assert_array_equal(acc.numpy(), d["acc"])
    assert acc[5].item() == 0.5  # exact tie
    np.testing.assert_allclose(O.triplet_accuracy(anc, pos, neg, discrete=False).numpy(), d["diff"], atol=1e-7)


def test_bertadam_trajectories(golden_dir):
    d = _load(golden_dir, "ref_bertadam.npz")
    p = torch.tensor([1.0, -2.0, 3.0])
    st = {}
    for k in range(3):
        O.bertadam_step([p], [2 * p.clone()], st, lr=1e-2, warmup=0.1, t_total=10)
        np.testing.assert_allclose(p.numpy(), d["tiny_traj"][k], atol=1e-7)
    np.testing.assert_allclose(d["tiny_traj"][0], [1.0, -2.0, 3.0])  # warmup_linear(0)=0 -> no-op
    params = [torch.tensor(d[f"p0_{i}"]) for i in range(5)]
    st = {}
    for step in range(6):
        grads = [torch.tensor(d[f"g{step}_{i}"]) for i in range(5)]
        O.bertadam_step(params, grads, st, lr=1e-3, warmup=0.25, t_total=8)
        for i in range(5):
            np.testing.assert_allclose(params[i].numpy(), d[f"p{step + 1}_{i}"], atol=2e-7)
    for i in range(5):
        np.testing.assert_allclose(st[i]["m"].numpy(), d[f"m_{i}"], atol=1e-6)
        np.testing.assert_allclose(st[i]["v"].numpy(), d[f"v_{i}"], rtol=1e-6, atol=1e-9)
    for s, m in zip(d["sched_steps"], d["sched_mult"]):
        assert abs(O.warmup_linear(s / 15000, 0.1) - m) < 1e-12
