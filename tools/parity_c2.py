"""Real-shape parity report: BASELINE configs[1] geometry at a batch the CPU oracle finishes in
seconds (default B = 8, 3x16x112x112 video + 36 800 audio samples).  Prints, per trunk stage, the free-running relative
L2 error of the HIP activations against the fp32 oracle, then embedding cosine / max-abs, |dloss| and per-stage
gradient errors; next to each the same figure for torch's own bf16 autocast of the oracle.

    python tools/parity_c2.py [--batch 8] [--frames 16] [--size 112] [--samples 36800] [--no-bwd] [--version r2plus1d_18]
"""
import argparse
import copy
import os
import sys
import time
import warnings

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import model as O
import pig.models
from pig.execution import default_config
from peppa_amd.data import synthetic_batch
from peppa_amd import video as PV
from peppa_amd import hip as H
from peppa_amd import layers as L


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def from_cl(y, B, thw, C):
    return y.float().cpu()[:, :C].reshape(B, *thw, C).permute(0, 4, 1, 2, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=112)
    ap.add_argument("--samples", type=int, default=36800)
    ap.add_argument("--version", default="r2plus1d_18")
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--no-autocast", action="store_true")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    cfg["video"]["version"] = args.version
    torch.manual_seed(0)
    ref = O.PeppaPigOracle(cfg, dropout=0.0, layer_drop=0.0).train()
    net = pig.models.PeppaPig(cfg)
    net.load_state_dict(ref.state_dict())
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "layer_drop"):
            m.layer_drop = 0.0
    net = net.cuda().train()
    batch = synthetic_batch(args.batch, args.frames, args.size, args.samples)
    B = args.batch
    rv, hv = ref.video_encoder.video, net.video_encoder.video
    stages = ["stem", "layer1", "layer2", "layer3", "layer4"]
    acts = {}
    hooks = [getattr(rv, s).register_forward_hook(lambda m, i, o, s=s: acts.__setitem__(s, o.detach())) for s in stages]
    sd = copy.deepcopy(ref.state_dict())
    t0 = time.time()
    V32 = ref.encode_video(batch.video)
    A32 = ref.encode_audio(batch.audio)
    loss32 = ref.loss(V32, A32)
    print(f"oracle forward {time.time() - t0:.1f} s; loss {loss32.item():.6f}", flush=True)
    acts32 = dict(acts)
    if not args.no_bwd:
        t0 = time.time()
        loss32.backward()
        print(f"oracle backward {time.time() - t0:.1f} s", flush=True)
    acts16 = None
    if not args.no_autocast:
        ref2 = copy.deepcopy(ref)
        ref2.load_state_dict(sd)
        rv2 = ref2.video_encoder.video
        acts.clear()
        h2 = [getattr(rv2, s).register_forward_hook(lambda m, i, o, s=s: acts.__setitem__(s, o.detach().float())) for s in stages]
        t0 = time.time()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            V16 = ref2.encode_video(batch.video).float()
        acts16 = dict(acts)
        print(f"oracle bf16-autocast forward {time.time() - t0:.1f} s", flush=True)
    for h in hooks:
        h.remove()

    # HIP, stage by stage (free running)
    gb = batch.to("cuda")
    with torch.no_grad():
        x = gb.video
        cur = torch.empty(x.numel() // 3, 8, dtype=torch.bfloat16, device="cuda")
        H.video_normalize_ndhwc(x, cur, *PV.VIDEO_STATS["peppa"])
        thw = tuple(x.shape[2:])
        bn_state = copy.deepcopy(hv.state_dict())
        cur, thw, _ = PV.run_plan(hv.stem_plan(), cur, thw, B, True, False, first=True)
        print(f"{'stage':8s} {'HIP rel-L2':>12s} {'torch-bf16':>12s}")
        y16 = f"{rel(acts16['stem'], acts32['stem']):12.5f}" if acts16 else ""
        print(f"{'stem':8s} {rel(from_cl(cur, B, thw, 64), acts32['stem']):12.5f} {y16}")
        for s, layer in zip(stages[1:], (hv.layer1, hv.layer2, hv.layer3, hv.layer4)):
            for blk in layer:
                cur, thw, _ = PV.run_plan(PV.VideoResNet.block_plan(blk), cur, thw, B, True, False)
            C = acts32[s].shape[1]
            y16 = f"{rel(acts16[s], acts32[s]):12.5f}" if acts16 else ""
            print(f"{s:8s} {rel(from_cl(cur, B, thw, C), acts32[s]):12.5f} {y16}")
        hv.load_state_dict(bn_state)   # undo the running-statistics update of this diagnostic pass
    net.zero_grad(set_to_none=True)
    loss = net.training_step(gb, 0)
    if not args.no_bwd:
        loss.backward()
    torch.cuda.synchronize()
    hv.load_state_dict(bn_state)
    with torch.no_grad():
        Vh = net.encode_video(gb.video).cpu()
        Ah = net.encode_audio(gb.audio).cpu()
    V32d, A32d = V32.detach(), A32.detach()
    cv = F.cosine_similarity(Vh, V32d, dim=1)
    ca = F.cosine_similarity(Ah, A32d, dim=1)
    print(f"video  emb: min cos {cv.min().item():.6f} mean cos {cv.mean().item():.6f} max-abs {(Vh - V32d).abs().max().item():.5f}")
    if acts16:
        c16 = F.cosine_similarity(V16, V32d, dim=1)
        print(f"  torch bf16 autocast: min cos {c16.min().item():.6f} max-abs {(V16 - V32d).abs().max().item():.5f}")
    print(f"audio  emb: min cos {ca.min().item():.6f} max-abs {(Ah - A32d).abs().max().item():.5f}")
    print(f"loss: HIP {loss.item():.6f} oracle {loss32.item():.6f} |d| {abs(loss.item() - loss32.item()):.6f}")
    if not args.no_bwd:
        refp = dict(ref.named_parameters())
        by_stage = {}
        for n, p in net.named_parameters():
            pr = refp[n]
            if pr.grad is None or p.grad is None:
                continue
            parts = n.split(".")
            key = ".".join(parts[:3]) if parts[0] == "video_encoder" and parts[1] == "video" else \
                (".".join(parts[:2]) if parts[0] == "video_encoder" else "audio")
            d = by_stage.setdefault(key, [0.0, 0.0, 0.0])
            d[0] += (p.grad.detach().cpu() - pr.grad).pow(2).sum().item()
            d[1] += pr.grad.pow(2).sum().item()
            d[2] = max(d[2], rel(p.grad, pr.grad))
        print("gradient rel-L2 per stage (all tensors pooled / worst tensor):")
        for k, (e, r, w) in by_stage.items():
            print(f"  {k:32s} {(e / (r + 1e-30)) ** 0.5:9.4f} {w:9.4f}")


if __name__ == "__main__":
    main()
