"""Every matrix-core launch of one training step, one kernel at a time (single stream), grouped by (kernel family, mode,
M, N, K): launches, average microseconds, TFLOP/s, total ms -- sorted by total time.  Finds the shapes that run furthest
below the matrix peak.    python tools/prof_gemm_keys.py [--frames 16 --samples 36800 --batch 64]"""
import argparse
import os
import sys
import warnings

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yaml

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--samples", type=int, default=36800)
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()

import pig.models
from peppa_amd import hip as H
from peppa_amd import video as PV
from peppa_amd.data import synthetic_batch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml.safe_load(open(os.path.join(root, "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
net.set_precision(args.dtype)
net._overlap, PV.OVERLAP_WGRAD = False, False
opt = net.configure_optimizers()
b = synthetic_batch(args.batch, args.frames, 112, args.samples).to("cuda")

# keys with M in them: wrap the two entry points
_ig, _wg = H.igemm, H.wgrad
recs = {}


def timed(key, flops, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    recs.setdefault(key, []).append((e0, e1, flops))


def igemm(A, Bt, Cout, M, N, K, g, *a, **kw):
    nb = kw.get("nbatch", 1)
    key = (H._igemm_family(g), H._MODE_NAMES[g.mode], M, N, K, nb, (g.kt, g.kh, g.kw), (g.st, g.sh, g.sw))
    timed(key, 2.0 * M * N * K * nb, lambda: _ig(A, Bt, Cout, M, N, K, g, *a, **kw))


def wgrad(X, dY, dW, M, Ni, Kj, g, *a, **kw):
    nb = kw.get("nbatch", 1)
    key = (H._wgrad_family(g), H._MODE_NAMES[g.mode], M, Ni, Kj, nb, (g.kt, g.kh, g.kw), (g.st, g.sh, g.sw))
    timed(key, 2.0 * M * Ni * Kj * nb, lambda: _wg(X, dY, dW, M, Ni, Kj, g, *a, **kw))


def step(i):
    opt.zero_grad(set_to_none=True)
    net.training_step(b, i).backward()
    opt.step()


for i in range(3):
    step(i)
H.igemm, H.wgrad = igemm, wgrad
for i in range(3):
    step(3 + i)
torch.cuda.synchronize()
H.igemm, H.wgrad = _ig, _wg
rows = []
for key, rs in recs.items():
    secs = sum(a.elapsed_time(b) for a, b, _ in rs) * 1e-3
    fl = sum(f for _, _, f in rs)
    rows.append((secs / 3, len(rs) / 3, secs / len(rs) * 1e6, fl / secs / 1e12, key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total {tot * 1e3:.2f} ms/step of matrix-core launches (each launch bracketed by events: +~5 us)")
print(f"{'ms/step':>8s} {'n/step':>6s} {'avg us':>8s} {'TF/s':>7s}  kernel, mode, M, N, K, batch, taps, stride")
for r in rows[:70]:
    print(f"{r[0] * 1e3:8.3f} {r[1]:6.1f} {r[2]:8.1f} {r[3]:7.1f}  {r[4]}")
