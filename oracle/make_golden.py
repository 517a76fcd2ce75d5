"""Generate golden vectors from the LIVE reference (build container only).

Run: `PYTHONPATH=/root/reference python oracle/make_golden.py`
Imports the reference's torch-only modules (`pig.loss`, `pig.metrics`,
`pig.optimization`, `pig.util`) and writes inputs + expected outputs
to `tests/golden/ref_*.npz`.  The fixtures are data only; no reference source ships.
The reference does not exist on the GPU box, so tests read the committed .npz files.
"""
import os
import sys
import warnings
import numpy as np
import torch
import torch.nn.functional as F

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference")
import pig.loss  # noqa: E402
import pig.metrics  # noqa: E402
import pig.optimization  # noqa: E402
import pig.util  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def unit_pair(n, seed=0, d=512):
    g = torch.Generator().manual_seed(seed)
    v = F.normalize(torch.randn(n, d, generator=g))
    a = F.normalize(torch.randn(n, d, generator=g))
    return v, a


def loss_fixture():
    out = {}
    for n in (4, 64, 512):
        v, a = unit_pair(n)
        v.requires_grad_()
        a.requires_grad_()
        loss = pig.loss.TripletLoss(margin=0.2)(v, a)
        loss.backward()
        out[f"loss_{n}"] = np.float64(loss.item())
        out[f"dVnorm_{n}"] = np.float64(v.grad.norm().item())
        out[f"dAnorm_{n}"] = np.float64(a.grad.norm().item())
        if n <= 64:
            out[f"V_{n}"], out[f"A_{n}"] = v.detach().numpy(), a.detach().numpy()
            out[f"dV_{n}"], out[f"dA_{n}"] = v.grad.numpy(), a.grad.numpy()
        else:  # keep the file small: first/last rows only
            out[f"dV_{n}_head"] = v.grad[:4].numpy()
            out[f"dA_{n}_tail"] = a.grad[-4:].numpy()
    # un-normalised inputs exercise cosine_matrix's own normalisation
    g = torch.Generator().manual_seed(7)
    v = (torch.randn(16, 512, generator=g) * 3).requires_grad_()
    a = (torch.randn(16, 512, generator=g) * 0.5).requires_grad_()
    loss = pig.loss.TripletLoss(margin=0.2)(v, a)
    loss.backward()
    out.update(rawV=v.detach().numpy(), rawA=a.detach().numpy(), raw_loss=np.float64(loss.item()),
               raw_dV=v.grad.numpy(), raw_dA=a.grad.numpy(),
               raw_cos=pig.util.cosine_matrix(v.detach(), a.detach()).numpy())
    # margin sweep on the N=4 pair
    v, a = unit_pair(4)
    out["margins"] = np.array([0.0, 0.1, 0.5, 1.0])
    out["margin_losses"] = np.array([pig.loss.TripletLoss(m)(v, a).item() for m in out["margins"]])
    np.savez(os.path.join(OUT, "ref_loss.npz"), **out)


def metrics_fixture():
    g = torch.Generator().manual_seed(3)
    anc = torch.randn(12, 512, generator=g)
    pos = torch.randn(12, 512, generator=g)
    neg = torch.randn(12, 512, generator=g)
    neg[5] = pos[5]  # exact tie -> 0.5
    acc = pig.metrics.triplet_accuracy(anc, pos, neg)
    diff = pig.metrics.triplet_accuracy(anc, pos, neg, discrete=False)
    cand = torch.randn(16, 512, generator=g)
    ref = cand + 0.8 * torch.randn(16, 512, generator=g)
    rec = pig.metrics.recall_at_n(cand, ref, torch.eye(16), n=3)
    rec1n = pig.metrics.recall_at_1_to_n(cand, ref, torch.eye(16), N=4)
    np.savez(os.path.join(OUT, "ref_metrics.npz"), anchor=anc.numpy(), positive=pos.numpy(),
             negative=neg.numpy(), acc=acc.numpy(), diff=diff.numpy(), cand=cand.numpy(),
             ref=ref.numpy(), recall_at_3=rec.numpy(), recall_1_to_4=rec1n.numpy())


def optim_fixture():
    out = {}
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0, 3.0]))
    opt = pig.optimization.BertAdam([p], lr=1e-2, warmup=0.1, t_total=10)
    traj = []
    for _ in range(3):
        opt.zero_grad()
        (p ** 2).sum().backward()
        opt.step()
        traj.append(p.detach().clone().numpy())
    out["tiny_traj"] = np.stack(traj)
    # several tensors, fixed gradients, 6 steps through warmup into decay
    g = torch.Generator().manual_seed(11)
    shapes = [(7,), (5, 3), (4, 2, 3, 3), (1, 1, 9), (33,)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    grads = [[torch.randn(*s, generator=g) * (10.0 if i == 2 else 0.3) for i, s in enumerate(shapes)]
             for _ in range(6)]
    opt = pig.optimization.BertAdam(params, lr=1e-3, warmup=0.25, t_total=8)
    for i, s in enumerate(shapes):
        out[f"p0_{i}"] = params[i].detach().clone().numpy()
    for step in range(6):
        for p_, g_ in zip(params, grads[step]):
            p_.grad = g_.clone()
        opt.step()
        for i in range(len(shapes)):
            out[f"g{step}_{i}"] = grads[step][i].numpy()
            out[f"p{step + 1}_{i}"] = params[i].detach().clone().numpy()
    for i in range(len(shapes)):
        st = opt.state[params[i]]
        out[f"m_{i}"], out[f"v_{i}"] = st["next_m"].numpy(), st["next_v"].numpy()
    steps = np.array([0, 1, 750, 1500, 7500, 15000])
    out["sched_steps"] = steps
    out["sched_mult"] = np.array([pig.optimization.warmup_linear(s / 15000, 0.1) for s in steps])
    np.savez(os.path.join(OUT, "ref_bertadam.npz"), **out)


def collate_fixture():
    """Ragged uint8 clips -> the padded fp32 batch, through the LIVE pig.util.pad_video_batch / pad_audio_batch.
    `featurize` (pig/data.py:66-72) needs moviepy clips, so its frame arithmetic (frame / 255 -> float32, stack,
    (T,H,W,C) -> (C,T,H,W)) is applied here to numpy frames with the same torch calls."""
    rng = np.random.default_rng(5)
    out = {}
    for tag, (Hh, W), Ts, Ls in (("a", (8, 8), [3, 5, 1, 4], [100, 257, 1, 64]), ("b", (5, 7), [2, 1, 3], [33, 7, 50])):
        frames = [rng.integers(0, 256, size=(t, Hh, W, 3), dtype=np.uint8) for t in Ts]
        frames[0][0, 0, 0] = (0, 255, 128)
        audio = [(0.1 * rng.standard_normal((1, l))).astype(np.float32) for l in Ls]
        video = [torch.stack([torch.tensor(f / 255).float() for f in clip]).permute(3, 0, 1, 2) for clip in frames]
        out[f"{tag}_frames"] = np.concatenate([f.reshape(-1) for f in frames])
        out[f"{tag}_T"], out[f"{tag}_hw"] = np.array(Ts), np.array([Hh, W])
        out[f"{tag}_audio"] = np.concatenate([a.reshape(-1) for a in audio])
        out[f"{tag}_L"] = np.array(Ls)
        out[f"{tag}_video_batch"] = pig.util.pad_video_batch(video).numpy()
        out[f"{tag}_audio_batch"] = pig.util.pad_audio_batch([torch.tensor(a) for a in audio]).numpy()
    np.savez_compressed(os.path.join(OUT, "ref_collate.npz"), **out)


if __name__ == "__main__":
    collate_fixture()
    loss_fixture()
    metrics_fixture()
    optim_fixture()
    print("wrote", sorted(os.listdir(OUT)))
