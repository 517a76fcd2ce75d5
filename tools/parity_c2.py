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
import types
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
    ap.add_argument("--blocks", action="store_true", help="teacher-forced forward/backward of every residual block at this shape")
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
    acts16 = grads16 = None
    if not args.no_autocast:   # torch's own bf16 autocast of the oracle: the yardstick for "what bf16 operands cost"
        ref2 = O.PeppaPigOracle(cfg, dropout=0.0, layer_drop=0.0).train()   # (a second instance: the fp32 graph of `ref`
        ref2.load_state_dict(sd)                                             # is still needed for its backward pass)
        h2 = [getattr(ref2.video_encoder.video, s).register_forward_hook(
            lambda m, i, o, s=s: acts.__setitem__(s, o.detach().float())) for s in stages]
        t0 = time.time()
        gR2 = torch.Generator().manual_seed(77)
        Rv2 = torch.randn(B, 512, generator=gR2)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            V16g = ref2.encode_video(batch.video)
        V16 = V16g.detach().float()
        acts16 = dict(acts)
        grads16 = None
        if not args.no_bwd:
            (V16g.float() * Rv2).sum().backward()
            grads16 = {n: p.grad for n, p in ref2.named_parameters() if p.grad is not None}
        del ref2
        print(f"oracle bf16-autocast forward+backward {time.time() - t0:.1f} s", flush=True)
    # gradients are compared under a SMOOTH objective <V, Rv> + <A, Ra> with fixed random Rv, Ra: the hinge loss of
    # near-identical random-init embeddings is a difference of almost cancelling terms, so its gradient amplifies the
    # forward rounding of V / A by ~30x and says nothing about the backward kernels (measured: 100 % "error")
    gR = torch.Generator().manual_seed(77)
    Rv, Ra = torch.randn(B, 512, generator=gR), torch.randn(B, 512, generator=gR)
    if not args.no_bwd:
        t0 = time.time()
        ((V32 * Rv).sum() + (A32 * Ra).sum()).backward()
        print(f"oracle backward {time.time() - t0:.1f} s", flush=True)
    ref_grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
    for h in hooks:
        h.remove()

    gb = batch.to("cuda")
    if args.blocks:
        bn_state = copy.deepcopy(hv.state_dict())
        blocks_teacher_forced(rv, hv, acts32["stem"], B)
        hv.load_state_dict(bn_state)
    # HIP, stage by stage (free running)
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
    with torch.no_grad():
        loss = net.training_step(gb, 0)
    hv.load_state_dict(bn_state)
    if not args.no_bwd:
        Vg, Ag = net.encode_pair(gb.video, gb.audio)
        ((Vg * Rv.cuda()).sum() + (Ag * Ra.cuda()).sum()).backward()
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
        by_stage = {}
        gmax = max(g.norm().item() for g in ref_grads.values())
        for n, p in net.named_parameters():
            if n not in ref_grads or p.grad is None:
                continue
            pr = types.SimpleNamespace(grad=ref_grads[n])
            parts = n.split(".")
            key = ".".join(parts[:3]) if parts[0] == "video_encoder" and parts[1] == "video" else \
                (".".join(parts[:2]) if parts[0] == "video_encoder" else "audio")
            d = by_stage.setdefault(key, [0.0, 0.0, 0.0, 0.0])
            d[0] += (p.grad.detach().cpu() - pr.grad).pow(2).sum().item()
            d[1] += pr.grad.pow(2).sum().item()
            if grads16 is not None and n in grads16:
                d[3] += (grads16[n].float() - pr.grad).pow(2).sum().item()
            if pr.grad.norm().item() > 1e-4 * gmax:     # (tensors whose true gradient is ~0, e.g. k_proj.bias: skip)
                d[2] = max(d[2], rel(p.grad, pr.grad))
        print("gradient rel-L2 per stage under <V,Rv> + <A,Ra> (all tensors pooled / worst tensor / torch bf16 autocast pooled):")
        for k, (e, r, w, e16) in by_stage.items():
            print(f"  {k:32s} {(e / (r + 1e-30)) ** 0.5:9.4f} {w:9.4f} {(e16 / (r + 1e-30)) ** 0.5:9.4f}")


def to_cl(x, cp):
    B, C = x.shape[:2]
    y = x.permute(0, 2, 3, 4, 1).reshape(-1, C)
    out = torch.zeros(y.shape[0], cp)
    out[:, :C] = y
    return out.to(torch.bfloat16).cuda()


def blocks_teacher_forced(rv, hv, x0, B):
    """Every residual block (and the stem's second unit chain via the layers' inputs) with the ORACLE's activation as
    input and a fixed random output gradient: isolates the kernels chosen at this geometry from the depth effect."""
    rb = lambda t: t.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(5)
    x = x0
    print(f"{'block':10s} {'fwd':>8s} {'dx':>8s} {'worst dW':>9s}   (teacher forced, rel-L2 vs fp32 oracle)")
    for li, (rlayer, hlayer) in enumerate(zip((rv.layer1, rv.layer2, rv.layer3, rv.layer4),
                                              (hv.layer1, hv.layer2, hv.layer3, hv.layer4))):
        for bi, (rblk, hblk) in enumerate(zip(rlayer, hlayer)):
            xin = rb(x.detach()).requires_grad_()
            out = rblk(xin)
            dout = rb(torch.randn(out.shape, generator=g))
            for p in rblk.parameters():
                p.grad = None
            out.backward(dout)
            C = xin.shape[1]
            thw = tuple(xin.shape[2:])
            with torch.no_grad():
                z, thw_o, tape = PV.run_plan(PV.VideoResNet.block_plan(hblk), to_cl(xin.detach(), L.cpad(C)), thw, B, True, True)
                grads = {}
                dx = PV.trunk_backward(tape, to_cl(dout, z.shape[1]), grads)
            torch.cuda.synchronize()
            ef = rel(from_cl(z, B, thw_o, out.shape[1]), out)
            eb = rel(from_cl(dx, B, thw, C), xin.grad)
            ew = max(rel(grads[ph], pr.grad) for pr, ph in zip(rblk.parameters(), hblk.parameters()))
            print(f"layer{li + 1}.{bi}  {ef:8.4f} {eb:8.4f} {ew:9.4f}", flush=True)
            x = out.detach()


if __name__ == "__main__":
    main()
