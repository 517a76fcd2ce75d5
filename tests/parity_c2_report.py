"""Real-shape parity of the HIP path against the fp32 CPU oracle -- measurement code shared by
tests/test_parity_c2_gpu.py and the command line:

    python tests/parity_c2_report.py [--batch 8] [--frames 16] [--size 112] [--samples 36800] [--version r2plus1d_18]
    python tests/parity_c2_report.py --triplets [--clips 128]

BASELINE configs[1] geometry at a batch the oracle finishes in seconds (B = 8, 3x16x112x112 video + 36 800 audio
samples; layer 4 normalises over 784 rows instead of the toy test's 16).  `report()` returns and prints
  * per trunk stage the free-running relative L2 error of the HIP activations, next to torch's own bf16 autocast of the
    oracle (the yardstick for "what bf16 operands cost");
  * every residual block teacher-forced (oracle activation in, fixed random gradient out): forward / dx / worst dW;
  * embedding cosine / max-abs, |dloss| (SURVEY 8d tolerances);
  * per-stage gradient errors under a SMOOTH objective <V,Rv> + <A,Ra> -- the hinge loss of near-identical random-init
    embeddings is a difference of almost cancelling terms and amplifies the forward rounding ~30x (measured 100 %), so
    it says nothing about the backward kernels.
`triplet_flips()` is the "triplet accuracy within +-0.2 %" check of SURVEY 8d on >= 10 000 duration-matched triplets.
Test infrastructure: imports the oracle; nothing in the product does.
"""
import argparse
import copy
import os
import random
import sys
import time
import warnings

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import model as O
import pig.models
import pig.metrics
import pig.triplet
from pig.execution import default_config
from peppa_amd.data import synthetic_batch, synthetic_structured_batch
from peppa_amd import video as PV
from peppa_amd import hip as H
from peppa_amd import layers as L

STAGES = ["stem", "layer1", "layer2", "layer3", "layer4"]


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def from_cl(y, B, thw, C):
    return y.float().cpu()[:, :C].reshape(B, *thw, C).permute(0, 4, 1, 2, 3)


def to_cl(x, cp):
    B, C = x.shape[:2]
    y = x.permute(0, 2, 3, 4, 1).reshape(-1, C)
    out = torch.zeros(y.shape[0], cp)
    out[:, :C] = y
    return out.to(H.act16()).cuda()


def make_cfg(version="r2plus1d_18", static=False):
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    cfg["video"]["version"] = version
    return cfg


def no_dropout(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "layer_drop"):
            m.layer_drop = 0.0
    return net


def build_pair(cfg, seed=0):
    torch.manual_seed(seed)
    ref = O.PeppaPigOracle(cfg, dropout=0.0, layer_drop=0.0).train()
    net = pig.models.PeppaPig(cfg)
    net.load_state_dict(ref.state_dict())
    return ref, no_dropout(net).cuda().train()


def blocks_teacher_forced(rv, hv, x0, B, log=print):
    """Every residual block with the ORACLE's activation as input and a fixed random output gradient: isolates the
    kernels chosen at this geometry from the depth effect.  Returns [(name, fwd, dx, worst dW)]."""
    rb = lambda t: t.to(H.act16()).float()
    g = torch.Generator().manual_seed(5)
    x = x0
    rows = []
    log(f"{'block':10s} {'fwd':>8s} {'dx':>8s} {'worst dW':>9s}   (teacher forced, rel-L2 vs fp32 oracle)")
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
            log(f"layer{li + 1}.{bi}  {ef:8.4f} {eb:8.4f} {ew:9.4f}")
            rows.append((f"layer{li + 1}.{bi}", ef, eb, ew))
            x = out.detach()
    return rows


def report(batch=8, frames=16, size=112, samples=36800, version="r2plus1d_18", bwd=True, autocast=True, blocks=True,
           threads=16, log=print, precision="bf16", cfg=None, pair=None, data=None, hinge=False):
    """precision: "bf16" | "fp16" -- which build of the HIP library runs (the yardstick stays torch's bf16 autocast).
    cfg / pair / data: a prepared config, (oracle, HIP model) pair with equal weights, and ClipBatch (defaults: the
    default config at `version`, a fresh random-init pair, iid-noise clips).  hinge: differentiate the triplet loss itself
    instead of the smooth objective (meaningful once the embeddings are separated, i.e. for a trained model)."""
    prev = H.set_precision(precision)
    try:
        return _report(batch, frames, size, samples, version, bwd, autocast, blocks, threads, log, precision, cfg, pair,
                       data, hinge)
    finally:
        H.set_precision(prev)


def _gkey(n):
    parts = n.split(".")
    if parts[0] == "video_encoder" and parts[1] == "video":
        return ".".join(parts[:3])
    return ".".join(parts[:2]) if parts[0] == "video_encoder" else "audio"


def _report(batch, frames, size, samples, version, bwd, autocast, blocks, threads, log, precision, cfg=None, pair=None,
            data=None, hinge=False):
    torch.set_num_threads(threads)
    cfg = make_cfg(version) if cfg is None else cfg
    ref, net = build_pair(cfg) if pair is None else pair
    net.set_precision(precision)
    data = synthetic_batch(batch, frames, size, samples) if data is None else data
    B = batch = data.video.shape[0]
    ref.zero_grad(set_to_none=True)       # (a prepared pair may come from an earlier report)
    rv, hv = ref.video_encoder.video, net.video_encoder.video
    acts = {}
    hooks = [getattr(rv, s).register_forward_hook(lambda m, i, o, s=s: acts.__setitem__(s, o.detach())) for s in STAGES]
    sd = copy.deepcopy(ref.state_dict())
    res = {}
    t0 = time.time()
    V32 = ref.encode_video(data.video)
    A32 = ref.encode_audio(data.audio)
    loss32 = ref.loss(V32, A32)
    log(f"oracle forward {time.time() - t0:.1f} s; loss {loss32.item():.6f}")
    acts32 = dict(acts)
    gR = torch.Generator().manual_seed(77)
    Rv, Ra = torch.randn(B, 512, generator=gR), torch.randn(B, 512, generator=gR)
    acts16 = grads16 = V16 = None
    if autocast:   # a second oracle instance: the fp32 graph of `ref` is still needed for its backward pass
        ref2 = O.PeppaPigOracle(cfg, dropout=0.0, layer_drop=0.0).train()
        ref2.load_state_dict(sd)
        for s in STAGES:
            getattr(ref2.video_encoder.video, s).register_forward_hook(lambda m, i, o, s=s: acts.__setitem__(s, o.detach().float()))
        t0 = time.time()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            V16g = ref2.encode_video(data.video)
        V16 = V16g.detach().float()
        acts16 = dict(acts)
        if bwd:
            if hinge:
                with torch.no_grad():
                    A16 = ref2.encode_audio(data.audio)
                ref2.loss(V16g.float(), A16).backward()
            else:
                (V16g.float() * Rv).sum().backward()
            grads16 = {n: p.grad for n, p in ref2.named_parameters() if p.grad is not None}
        del ref2
        log(f"oracle bf16-autocast forward+backward {time.time() - t0:.1f} s")
    if bwd:
        t0 = time.time()
        (loss32 if hinge else (V32 * Rv).sum() + (A32 * Ra).sum()).backward()
        log(f"oracle backward {time.time() - t0:.1f} s")
    ref_grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
    for h in hooks:
        h.remove()

    gb = data.to("cuda")
    bn_state = copy.deepcopy(hv.state_dict())
    if blocks:
        res["blocks"] = blocks_teacher_forced(rv, hv, acts32["stem"], B, log)
        hv.load_state_dict(bn_state)
    # HIP, stage by stage (free running)
    with torch.no_grad():
        x = gb.video
        cur, thw, _ = PV.normalized_input(x, "peppa", hv.stem_plan()[0][1])
        cur, thw, _ = PV.run_plan(hv.stem_plan(), cur, thw, B, True, False, first=True)
        log(f"{'stage':8s} {'HIP rel-L2':>12s} {'torch-bf16':>12s}")
        res["stages"] = {}
        for s, layer in zip(STAGES, (None, hv.layer1, hv.layer2, hv.layer3, hv.layer4)):
            if layer is not None:
                for blk in layer:
                    cur, thw, _ = PV.run_plan(PV.VideoResNet.block_plan(blk), cur, thw, B, True, False)
            C = acts32[s].shape[1]
            e = rel(from_cl(cur, B, thw, C), acts32[s])
            y = rel(acts16[s], acts32[s]) if acts16 else float("nan")
            res["stages"][s] = (e, y)
            log(f"{s:8s} {e:12.5f} {y:12.5f}")
        hv.load_state_dict(bn_state)   # undo the running-statistics update of this diagnostic pass
    net.zero_grad(set_to_none=True)
    with torch.no_grad():
        loss = net.training_step(gb, 0)
    hv.load_state_dict(bn_state)
    if bwd:
        Vg, Ag = net.encode_pair(gb.video, gb.audio)
        (net.loss(Vg, Ag) if hinge else (Vg * Rv.cuda()).sum() + (Ag * Ra.cuda()).sum()).backward()
    torch.cuda.synchronize()
    hv.load_state_dict(bn_state)
    with torch.no_grad():
        Vh = net.encode_video(gb.video).cpu()
        Ah = net.encode_audio(gb.audio).cpu()
    V32d, A32d = V32.detach(), A32.detach()
    cv = F.cosine_similarity(Vh, V32d, dim=1)
    ca = F.cosine_similarity(Ah, A32d, dim=1)
    res.update(video_cos=cv.min().item(), video_maxabs=(Vh - V32d).abs().max().item(), audio_cos=ca.min().item(),
               audio_maxabs=(Ah - A32d).abs().max().item(), loss=loss.item(), loss_ref=loss32.item(),
               dloss=abs(loss.item() - loss32.item()))
    log(f"video  emb: min cos {res['video_cos']:.6f} mean cos {cv.mean().item():.6f} max-abs {res['video_maxabs']:.5f}")
    if V16 is not None:
        c16 = F.cosine_similarity(V16, V32d, dim=1)
        res["video_cos_bf16"] = c16.min().item()
        log(f"  torch bf16 autocast: min cos {c16.min().item():.6f} max-abs {(V16 - V32d).abs().max().item():.5f}")
    log(f"audio  emb: min cos {res['audio_cos']:.6f} max-abs {res['audio_maxabs']:.5f}")
    log(f"loss: HIP {loss.item():.6f} oracle {loss32.item():.6f} |d| {res['dloss']:.6f}")
    if bwd:
        by_stage = {}
        gmax = max(g.norm().item() for g in ref_grads.values())
        for n, p in net.named_parameters():
            if n not in ref_grads:
                continue
            if p.grad is None:
                res.setdefault("missing_grads", []).append(n)     # the oracle has a gradient here, the HIP path none
                continue
            rg = ref_grads[n]
            hg = p.grad.detach().float().cpu()
            d = by_stage.setdefault(_gkey(n), [0.0] * 8)
            d[0] += (hg - rg).pow(2).sum().item()
            d[1] += rg.pow(2).sum().item()
            if rg.norm().item() > 1e-4 * gmax:     # (tensors whose true gradient is ~0, e.g. k_proj.bias: skip)
                d[2] = max(d[2], rel(hg, rg))
            d[4] += (hg * rg).sum().item()
            d[5] += hg.pow(2).sum().item()
            if grads16 is not None and n in grads16:
                g16 = grads16[n].float()
                d[3] += (g16 - rg).pow(2).sum().item()
                d[6] += (g16 * rg).sum().item()
                d[7] += g16.pow(2).sum().item()
        what = "the triplet loss" if hinge else "<V,Rv> + <A,Ra>"
        log(f"gradients per stage under {what}, all tensors of a stage pooled, against the fp32 oracle:")
        log(f"  {'stage':32s} {'rel-L2':>8s} {'worst':>8s} {'|g|/|ref|':>9s} {'cosine':>8s}   torch bf16 autocast: "
            f"{'rel-L2':>8s} {'|g|/|ref|':>9s} {'cosine':>8s}")
        res["grads"], res["gstats"] = {}, {}
        for k, (e, r, w, e16, dot, hh, dot16, yy) in by_stage.items():
            r = r + 1e-30
            res["grads"][k] = ((e / r) ** 0.5, w, (e16 / r) ** 0.5 if yy else float("nan"))
            st = dict(ratio=(hh / r) ** 0.5, cos=dot / ((hh * r) ** 0.5 + 1e-30),
                      ratio16=(yy / r) ** 0.5 if yy else float("nan"),
                      cos16=dot16 / ((yy * r) ** 0.5 + 1e-30) if yy else float("nan"))
            res["gstats"][k] = st
            log(f"  {k:32s} {res['grads'][k][0]:8.4f} {w:8.4f} {st['ratio']:9.4f} {st['cos']:8.4f}   "
                f"{'':21s}{res['grads'][k][2]:8.4f} {st['ratio16']:9.4f} {st['cos16']:8.4f}")
        if res.get("missing_grads"):
            log("  NO gradient on the HIP path for:", res["missing_grads"][:8])
        extra = [n for n, p in net.named_parameters() if p.grad is not None and n not in ref_grads]
        res["extra_grads"] = extra
        if extra:
            log("  gradient on the HIP path where the oracle has none:", extra[:8])
    return res


def conditioned_report(steps=300, batch=8, frames=16, size=112, samples=36800, lr=2e-4, pool=6, log=print, threads=16, common_weight=1.0,
                       margin=None):
    """Full-depth parity for a CONDITIONED model: the HIP model is trained `steps` optimizer steps
    (BertAdam) on structured synthetic clips, `pool` batches in rotation; its state -- weights AND BatchNorm running
    statistics -- is loaded into the oracle, and the free-running stage errors, the embeddings and the full-depth
    gradients are compared on one of the training batches.  Answers whether the ~100 % full-depth gradient error of the
    random-init trunk is the chaos of random initialisation (then it shrinks here) or the kernels (then it does not).

    Training objective: every clip is pulled towards a fixed random unit vector of its own (shared by its video and its
    audio, pairwise cosine ~0.5 between clips): smooth, never degenerate.  (The reference's all-negatives hinge loss
    collapses a random-init model within 300 steps -- every clip on one point, loss = 2 m (N-1)/N = 0.35 exactly, true
    gradient 0: measured in round 3 -- which conditions nothing.)  The triplet loss's own gradient is compared as well when
    its hinges are neither all off nor all on."""
    import pig.optimization
    torch.set_num_threads(threads)
    cfg = make_cfg()
    if margin is not None:      # (the triplet loss's margin: only the hinge comparison at the end sees it)
        cfg["margin"] = margin
    ref, net = build_pair(cfg)
    batches = [synthetic_structured_batch(batch, frames, size, samples, seed=100 + k).to("cuda") for k in range(pool)]
    g = torch.Generator().manual_seed(7)
    common = torch.randn(1, 1, 512, generator=g)
    # (common_weight 1.0: pairwise cosine ~0.5 between the clips' targets; smaller: more spread, so that the trained model's
    # diagonal similarities exceed some off-diagonal ones by more than the margin and the triplet loss's hinges are PARTLY active)
    targets = F.normalize(common_weight * common + torch.randn(pool, batch, 512, generator=g), dim=-1).cuda()
    optim = pig.optimization.BertAdam(net.parameters(), lr=lr, warmup=0.05, t_total=2 * steps)
    t0, trace = time.time(), []
    for i in range(steps):
        optim.zero_grad(set_to_none=True)
        V, A = net.encode_pair(batches[i % pool].video, batches[i % pool].audio)
        obj = -((V * targets[i % pool]).sum() + (A * targets[i % pool]).sum()) / (2 * batch)
        obj.backward()
        optim.step()
        if i % max(1, steps // 10) == 0 or i == steps - 1:
            trace.append(round(-obj.item(), 3))
    torch.cuda.synchronize()
    log(f"trained the HIP model {steps} steps on structured clips in {time.time() - t0:.1f} s; mean cosine to the clips' "
        f"targets {trace}")
    ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    held = batches[0].to("cpu")
    out = {"target_cosine": trace}
    log("-- smooth objective <V,Rv> + <A,Ra> --")
    out["smooth"] = report(blocks=False, log=log, cfg=cfg, pair=(ref, net), data=held, threads=threads)
    with torch.no_grad():
        S = O.cosine_matrix(ref.encode_video(held.video), ref.encode_audio(held.audio))
        d = torch.diag(S)
        off = ~torch.eye(batch, dtype=torch.bool)
        active = (((cfg["margin"] + S - d.view(1, -1)) > 0)[off].float().mean().item()
                  + ((cfg["margin"] + S - d.view(-1, 1)) > 0)[off].float().mean().item()) / 2
    out["hinge_active"] = active
    log(f"-- the triplet loss itself: {100 * active:.0f} % of its off-diagonal hinges are active on this batch "
        f"(diagonal {d.mean().item():.3f}, off-diagonal {S[off].mean().item():.3f}) --")
    if 0.05 < active < 0.95:     # (all on: the loss is linear in S, i.e. one more smooth objective; all off: zero gradient)
        out["hinge"] = report(blocks=False, log=log, cfg=cfg, pair=(ref, net), data=held, threads=threads, hinge=True)
    return out


def forward_b64(batch=64, frames=16, size=112, samples=36800, log=print, threads=16):
    """BASELINE configs[1] at its TRUE batch (64 clips): both encoders + loss against the fp32 oracle (forward only:
    ~15 s of CPU, no graph kept)."""
    torch.set_num_threads(threads)
    cfg = make_cfg()
    ref, net = build_pair(cfg)
    data = synthetic_batch(batch, frames, size, samples)
    t0 = time.time()
    with torch.no_grad():
        V32 = ref.encode_video(data.video)
        A32 = ref.encode_audio(data.audio)
        loss32 = ref.loss(V32, A32).item()
    log(f"oracle forward at batch {batch}: {time.time() - t0:.1f} s, loss {loss32:.6f}")
    gb = data.to("cuda")
    with torch.no_grad():
        Vh, Ah = net.encode_pair(gb.video, gb.audio)
        loss = net.loss(Vh, Ah).item()
    Vh, Ah = Vh.cpu(), Ah.cpu()
    res = dict(video_cos=F.cosine_similarity(Vh, V32, dim=1).min().item(), video_maxabs=(Vh - V32).abs().max().item(),
               audio_cos=F.cosine_similarity(Ah, A32, dim=1).min().item(), audio_maxabs=(Ah - A32).abs().max().item(),
               loss=loss, loss_ref=loss32, dloss=abs(loss - loss32))
    log(f"batch {batch}: video min cos {res['video_cos']:.6f} max-abs {res['video_maxabs']:.5f}; audio min cos "
        f"{res['audio_cos']:.6f} max-abs {res['audio_maxabs']:.5f}; loss HIP {loss:.6f} oracle {loss32:.6f}")
    return res


def frozen_report(batch=8, frames=16, size=112, samples=36800, log=print, threads=16):
    """BASELINE configs[2] (hparams_freeze_wav2vec.yaml: feature extractor + 12 transformer layers frozen,
    pig/models.py:75-81) at the real clip geometry: the trainable audio parameters sit BEFORE the frozen layers
    (feature projection, positional conv, the encoder's LayerNorm), so their gradients cross all 12 frozen layers by data
    gradients only (T = 114 attention backward included).  Smooth objective, against the oracle with the same freezing."""
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "hparams_freeze_wav2vec.yaml")))
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    res = report(batch, frames, size, samples, blocks=False, autocast=False, log=log, cfg=cfg, threads=threads)
    ref, net = build_pair(cfg)
    res["frozen_names"] = [n for n, p in ref.named_parameters() if not p.requires_grad]
    res["hip_frozen_names"] = [n for n, p in net.named_parameters() if not p.requires_grad]
    return res


# ---- triplet accuracy within +-0.2 % ------------------------------------------------------------------
def _oracle_features(ref, data, bs):
    """Trunk features of the oracle, batch by batch (train-mode BatchNorm: batch statistics of each batch of `bs`)."""
    fv, fa = [], []
    with torch.no_grad():
        for i in range(0, data.video.shape[0], bs):
            v, a = data.video[i:i + bs], data.audio[i:i + bs]
            x = ref.video_encoder.video.trunk(O.normalize_video(v, ref.video_encoder.norm_kind))
            fv.append(x.mean(dim=(-1, -2)).permute(0, 2, 1))                       # (bs, T', 512): VideoAttention's input
            fa.append(ref.audio_encoder.audio(a.squeeze(1))[0])                    # (bs, T, 28)
    return torch.cat(fv), torch.cat(fa)


def _oracle_heads(ref, fv, fa):
    ve, ae = ref.video_encoder, ref.audio_encoder
    V = F.normalize(ve.project(ve.videopool.attn(fv)), p=2, dim=1)
    A = F.normalize(ae.project(ae.audiopool(fa)), p=2, dim=1)
    return V, A


def triplet_flips(clips=128, bs=8, frames=16, size=112, samples=36800, fit_steps=(0, 40, 400), n_samples=160, threads=16,
                  log=print):
    """Embeds `clips` structured synthetic clips with the oracle and with the HIP path (same weights, same batches) and
    scores >= 10 000 duration-matched triplets (pig/triplet.py:99-121 pairing, pig/metrics.py:45-52 accuracy) with
    both.  A random-init model embeds every clip almost identically, so its triplet decisions are coin flips decided by
    rounding; the check is therefore also made after the pooling / projection heads have been FITTED on the oracle's
    features for `fit_steps` Adam steps of the triplet loss (a partially trained and a trained-like model).  Returns
    {steps: (oracle accuracy, HIP accuracy, fraction of flipped decisions, number of triplets)}."""
    torch.set_num_threads(threads)
    cfg = make_cfg()
    ref, net = build_pair(cfg)
    data = synthetic_structured_batch(clips, frames, size, samples)
    t0 = time.time()
    fv, fa = _oracle_features(ref, data, bs)
    log(f"oracle features of {clips} clips: {time.time() - t0:.1f} s")
    dur = [2.0 + 0.1 * (i % 4) for i in range(clips)]                # four duration groups
    random.seed(0)
    trip = []
    for _ in range(n_samples):
        trip += list(pig.triplet._triplets(range(clips), lambda i: dur[i]))
    pos, neg = (torch.tensor(t) for t in zip(*trip))
    heads = [p for m in (ref.video_encoder.videopool, ref.video_encoder.project, ref.audio_encoder.audiopool,
                         ref.audio_encoder.project) for p in m.parameters()]
    optim = torch.optim.Adam(heads, lr=1e-3)
    out, done = {}, 0
    for steps in fit_steps:
        for _ in range(steps - done):
            optim.zero_grad()
            ref.loss(*_oracle_heads(ref, fv, fa)).backward()
            optim.step()
        done = steps
        with torch.no_grad():
            Vo, Ao = _oracle_heads(ref, fv, fa)
            fit_loss = ref.loss(Vo, Ao).item()
        net.load_state_dict(ref.state_dict())
        bn_state = copy.deepcopy(net.video_encoder.video.state_dict())
        Vh, Ah = [], []
        with torch.no_grad():
            for i in range(0, clips, bs):
                Vh.append(net.encode_video(data.video[i:i + bs].cuda()))
                Ah.append(net.encode_audio(data.audio[i:i + bs].cuda()))
                net.video_encoder.video.load_state_dict(bn_state)
        Vh, Ah = torch.cat(Vh), torch.cat(Ah)
        acc_o = O.triplet_accuracy(Ao[pos], Vo[pos], Vo[neg])
        acc_h = pig.metrics.triplet_accuracy(Ah[pos.cuda()], Vh[pos.cuda()], Vh[neg.cuda()]).cpu()
        flips = (acc_o != acc_h).float().mean().item()
        out[steps] = (acc_o.mean().item(), acc_h.mean().item(), flips, len(trip))
        cos = F.cosine_similarity(Vh.cpu(), Vo, dim=1).min().item()
        log(f"heads fitted {steps:4d} steps (oracle loss {fit_loss:.4f}): triplet accuracy oracle {out[steps][0]:.4f} HIP "
            f"{out[steps][1]:.4f}  flipped {100 * flips:.3f} % of {len(trip)} triplets; video min cos {cos:.5f}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=112)
    ap.add_argument("--samples", type=int, default=36800)
    ap.add_argument("--version", default="r2plus1d_18")
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--no-autocast", action="store_true")
    ap.add_argument("--no-blocks", action="store_true")
    ap.add_argument("--triplets", action="store_true")
    ap.add_argument("--conditioned", type=int, default=0, help="train the HIP model this many steps first")
    ap.add_argument("--b64", action="store_true")
    ap.add_argument("--frozen", action="store_true")
    ap.add_argument("--clips", type=int, default=128)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--precision", default="bf16")
    args = ap.parse_args()
    log = lambda *a: print(*a, flush=True)
    if args.conditioned:
        conditioned_report(args.conditioned, args.batch, args.frames, args.size, args.samples, log=log, threads=args.threads)
    elif args.b64:
        forward_b64(log=log, threads=args.threads)
    elif args.frozen:
        frozen_report(args.batch, args.frames, args.size, args.samples, log=log, threads=args.threads)
    elif args.triplets:
        triplet_flips(args.clips, frames=args.frames, size=args.size, samples=args.samples, threads=args.threads, log=log)
    else:
        report(args.batch, args.frames, args.size, args.samples, args.version, not args.no_bwd, not args.no_autocast,
               not args.no_blocks, args.threads, log, args.precision)


if __name__ == "__main__":
    main()
