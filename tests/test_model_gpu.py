"""Parity of the HIP training step (pig API -> peppa_amd) against the CPU oracle, same weights and
same synthetic clips.

What is asserted, and why these tolerances (DESIGN.md "Numerics"):
  * per-stage parity (stem, each residual block) with the oracle's own activation as input
    ("teacher forced"): forward <= 2.5 % relative L2 (measured 0.5-0.7 %, identical to torch's bf16
    autocast of the same block); backward <= 16-18 %: ReLU masks of near-zero bf16 activations flip
    and BatchNorm's backward subtracts two large projections, so torch's own bf16 autocast of one
    oracle block is 9-12 % off fp32 for dx and 10-13 % for dW (measured in the build container);
    the HIP path measures the same 9-12 %.  Each primitive is checked tightly in
    test_kernels_gpu.py; this test guards the composition (a dropped residual or a wrong operand
    shows up as >> 20 %);
  * end to end, a random-init r2plus1d_18 with train-mode BatchNorm amplifies any bf16 rounding by
    ~1.1x per conv unit (torch's own bf16 autocast of the ORACLE reaches cosine 0.984 / trunk error
    27 % on this input), so the full-depth check uses that autocast run as the yardstick: the HIP
    path must be as close to the fp32 oracle as torch's bf16 run is (x1.5 slack);
  * the audio tower (LayerNorm, no BatchNorm) is well conditioned: cosine >= 0.999, max-abs <= 2e-2.
"""
import copy
import warnings
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")

from oracle import model as O
import pig.models
import pig.optimization
import pig.metrics
from pig.execution import default_config
from peppa_amd.data import synthetic_batch
from peppa_amd import video as PV
from peppa_amd import layers as L
from peppa_amd import hip as H

DEV = "cuda"


def make_cfg(freeze=False):
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = False
    cfg["audio"]["pretrained"] = False
    if freeze:
        cfg["audio"]["freeze_feature_extractor"] = True
        cfg["audio"]["freeze_encoder_layers"] = 12
    return cfg


def build_pair(cfg, seed=0):
    torch.manual_seed(seed)
    ref = O.PeppaPigOracle(cfg, dropout=0.0, layer_drop=0.0)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # non-trivial affine parameters so their gradients are exercised
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.LayerNorm, torch.nn.GroupNorm)):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    net = pig.models.PeppaPig(cfg)
    missing, unexpected = net.load_state_dict(ref.state_dict(), strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    no_dropout(net)
    return ref, net.to(DEV)


def no_dropout(net):
    """Parity runs use p = 0 (RNG streams cannot match the reference's); throughput runs keep 0.1."""
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "layer_drop"):
            m.layer_drop = 0.0


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def rb(t):
    return t.to(torch.bfloat16).float()


def to_cl(x, cp):
    B, C = x.shape[:2]
    y = x.permute(0, 2, 3, 4, 1).reshape(-1, C)
    out = torch.zeros(y.shape[0], cp)
    out[:, :C] = y
    return out.to(torch.bfloat16).to(DEV)


def from_cl(y, B, thw, C):
    return y.float().cpu()[:, :C].reshape(B, *thw, C).permute(0, 4, 1, 2, 3)


def test_video_blocks_teacher_forced():
    """Every residual block of r2plus1d_18, forward and backward, fed the oracle's own input."""
    cfg = make_cfg()
    ref, net = build_pair(cfg)
    ref.train(); net.train()
    batch = synthetic_batch(4, 4, 32, 4000)
    rv, hv = ref.video_encoder.video, net.video_encoder.video
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        x = rv.stem(O.normalize_video(batch.video, "peppa"))
    worst_f = worst_b = 0.0
    for li, (rlayer, hlayer) in enumerate(zip((rv.layer1, rv.layer2, rv.layer3, rv.layer4),
                                              (hv.layer1, hv.layer2, hv.layer3, hv.layer4))):
        for bi, (rblk, hblk) in enumerate(zip(rlayer, hlayer)):
            xin = rb(x.detach()).requires_grad_()
            out = rblk(xin)
            dout = rb(torch.randn(out.shape, generator=g))
            for p in rblk.parameters():
                p.grad = None
            out.backward(dout)
            B, C = xin.shape[:2]
            thw = tuple(xin.shape[2:])
            with torch.no_grad():
                z, thw_o, tape = PV.run_plan(PV.VideoResNet.block_plan(hblk), to_cl(xin.detach(), L.cpad(C)), thw, B,
                                             True, True)
                grads = {}
                dx = PV.trunk_backward(tape, to_cl(dout, z.shape[1]), grads)
            torch.cuda.synchronize()
            ef = rel(from_cl(z, B, thw_o, out.shape[1]), out)
            eb = rel(from_cl(dx, B, thw, C), xin.grad)
            worst_f, worst_b = max(worst_f, ef), max(worst_b, eb)
            print(f"layer{li + 1}.{bi}: fwd {ef:.4f} dx {eb:.4f}", end="")
            assert ef < 0.025 and eb < 0.16, (li, bi, ef, eb)
            for (n, pr), ph in zip(rblk.named_parameters(), hblk.parameters()):
                e = rel(grads[ph], pr.grad)
                assert e < 0.18, (li, bi, n, e)
            print("  params ok")
            x = out.detach()
    print("worst fwd", worst_f, "worst dx", worst_b)


def test_video_stem_teacher_forced():
    cfg = make_cfg()
    ref, net = build_pair(cfg)
    ref.train(); net.train()
    batch = synthetic_batch(2, 4, 32, 4000)
    rv, hv = ref.video_encoder.video, net.video_encoder.video
    out = rv.stem(O.normalize_video(batch.video, "peppa"))
    dout = rb(torch.randn(out.shape, generator=torch.Generator().manual_seed(2)))
    out.backward(dout)
    with torch.no_grad():
        xg = batch.video.to(DEV)
        cur, thw_in, _ = PV.normalized_input(xg, "peppa", hv.stem_plan()[0][1])
        z, thw, tape = PV.run_plan(hv.stem_plan(), cur, thw_in, 2, True, True, first=True)
        grads = {}
        PV.trunk_backward(tape, to_cl(dout, 64), grads)
    torch.cuda.synchronize()
    assert rel(from_cl(z, 2, thw, 64), out) < 0.02
    for pr, ph in zip(rv.stem.parameters(), hv.stem.parameters()):
        assert rel(grads[ph], pr.grad) < 0.12


@pytest.mark.parametrize("freeze", [False, True])
def test_audio_tower_parity(freeze):
    """wav2vec2 + attention pooling + projection: forward and (smooth-objective) gradients."""
    cfg = make_cfg(freeze)
    ref, net = build_pair(cfg)
    ref.train(); net.train()
    batch = synthetic_batch(4, 4, 32, 4000)
    R = torch.randn(4, 512, generator=torch.Generator().manual_seed(1))
    Ar = ref.encode_audio(batch.audio)
    (Ar * R).sum().backward()
    A = net.encode_audio(batch.audio.to(DEV))
    (A * R.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    cos = F.cosine_similarity(A.detach().cpu(), Ar.detach(), dim=1).min().item()
    err = (A.detach().cpu() - Ar.detach()).abs().max().item()
    print(f"audio embeddings: min cosine {cos:.6f}, max abs err {err:.4g}")
    assert cos >= 0.999 and err <= 2e-2
    refp = dict(ref.audio_encoder.named_parameters())
    rows = []
    for name, p in net.audio_encoder.named_parameters():
        pr = refp[name]
        if pr.grad is None:
            assert p.grad is None, f"{name}: unexpected gradient"
            continue
        assert p.grad is not None and p.grad.shape == pr.grad.shape, name
        rows.append((rel(p.grad, pr.grad), pr.grad.norm().item(), name))
    rows.sort(reverse=True)
    for e, n, name in rows[:10]:
        print(f"  grad rel err {e:.4f} |g|={n:.2e} {name}")
    gmax = max(n for _, n, _ in rows)
    # tensors whose true gradient is (analytically) ~0, e.g. k_proj.bias, only need to be small
    bad = [(e, name) for e, n, name in rows if (n > 1e-4 * gmax and e > 0.08)]
    assert not bad, bad[:8]
    tiny = [(name, n) for e, n, name in rows if n <= 1e-4 * gmax]
    hp = dict(net.audio_encoder.named_parameters())
    for name, n in tiny:
        assert hp[name].grad.norm().item() < 1e-2 * gmax, name
    if freeze:
        fe = net.audio_encoder.audio.feature_extractor
        assert all(p.grad is None for p in fe.parameters())
        assert net.audio_encoder.audio.encoder.transformer.layers[0].attention.q_proj.weight.grad is None
        assert net.audio_encoder.audio.encoder.transformer.pos_conv_embed.conv.weight_v.grad is not None
        assert net.audio_encoder.audio.encoder.feature_projection.projection.weight.grad is not None


def test_training_step_end_to_end_with_bf16_yardstick():
    cfg = make_cfg()
    ref, net = build_pair(cfg)
    ref.train(); net.train()
    batch = synthetic_batch(4, 4, 32, 4000)
    with torch.no_grad():
        V32, A32 = ref.encode_video(batch.video), ref.encode_audio(batch.audio)
        sd = copy.deepcopy(ref.state_dict())  # autocast run must not see updated BN buffers
        with torch.autocast("cpu", dtype=torch.bfloat16):
            V16 = ref.encode_video(batch.video).float()
        ref.load_state_dict(sd)
        loss32 = ref.loss(V32, A32).item()
    gb = batch.to(DEV)
    loss = net.training_step(gb, 0)
    loss.backward()
    torch.cuda.synchronize()
    assert rel(net.video_encoder.video.stem[1].running_mean, ref.video_encoder.video.stem[1].running_mean) < 0.05
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    used = [n for n, p in net.named_parameters() if not n.startswith("video_encoder.video.fc")]
    assert all(dict(net.named_parameters())[n].grad is not None for n in used)
    assert net.video_encoder.video.fc.weight.grad is None  # unused parameter
    with torch.no_grad():
        Vh = net.encode_video(gb.video).cpu()
        Ah = net.encode_audio(gb.audio).cpu()
    yard = 1 - F.cosine_similarity(V16, V32, dim=1).min().item()
    ours = 1 - F.cosine_similarity(Vh, V32, dim=1).min().item()
    print(f"video 1-cos: HIP {ours:.5f} vs torch bf16 autocast of the oracle {yard:.5f}; loss {loss.item():.5f} / {loss32:.5f}")
    assert (Vh.norm(dim=1) - 1).abs().max() < 1e-4 and (Ah.norm(dim=1) - 1).abs().max() < 1e-4
    assert ours <= 1.5 * yard + 2e-3
    assert F.cosine_similarity(Ah, A32, dim=1).min().item() >= 0.999
    assert abs(loss.item() - loss32) <= 0.05  # hinge loss on N=4 noisy video embeddings


def test_optimizer_step_matches_reference_rule():
    cfg = make_cfg()
    ref, net = build_pair(cfg, seed=3)
    gb = synthetic_batch(2, 4, 32, 4000, seed=7).to(DEV)
    optim = net.configure_optimizers()
    assert isinstance(optim, pig.optimization.BertAdam)
    names = [n for n, _ in net.named_parameters()]
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    cpu_params = [before[n].cpu().clone() for n in names]
    st = {}
    for step in range(2):
        optim.zero_grad()
        loss = net.training_step(gb, step)
        loss.backward()
        grads = [p.grad.detach().cpu().clone() if p.grad is not None else None for p in net.parameters()]
        optim.step()
        O.bertadam_step(cpu_params, grads, st, **{k: cfg["optimizer"][k] for k in ("lr", "warmup", "t_total")})
        torch.cuda.synchronize()
        for n, p, pc in zip(names, net.parameters(), cpu_params):
            d = (p.detach().cpu() - pc).abs().max().item()
            assert d <= 1e-6 + 1e-5 * pc.abs().max().item(), (n, step, d)
        if step == 0:  # warmup_linear(0) == 0: the first step leaves the weights untouched
            assert all(torch.equal(p.detach(), before[n]) for n, p in net.named_parameters())
    moved = sum((p.detach() != before[n]).any().item() for n, p in net.named_parameters())
    assert moved > 100


def test_api_surface_and_errors():
    cfg = make_cfg()
    for key, field in (("video", "pooling"), ("audio", "pooling"), ("video", "version")):
        bad = copy.deepcopy(cfg)
        bad[key][field] = "nope"
        with pytest.raises(ValueError):
            pig.models.PeppaPig(bad)
    net = pig.models.PeppaPig(cfg)
    with pytest.raises(Exception):  # CPU tensors are rejected: no fallback
        net.encode_audio(torch.zeros(1, 1, 4000))
    net = net.to(DEV).eval()
    b = synthetic_batch(2, 4, 32, 4000).to(DEV)
    with torch.no_grad():
        out = net(b)
        out2 = net(b)
    assert out.video.shape == (2, 512) and out.audio.shape == (2, 512)
    assert torch.equal(out.video, out2.video)  # eval mode: running statistics, deterministic
    feats, _ = net.audio_encoder.audio.extract_features(b.audio.squeeze(1))
    assert feats.shape == (2, 12, 512)
    acc = pig.metrics.triplet_accuracy(out.audio, out.video, out.video.flip(0))
    ref_acc = O.triplet_accuracy(out.audio.cpu(), out.video.cpu(), out.video.flip(0).cpu())
    assert torch.equal(acc.cpu(), ref_acc)
    cm = pig.loss.cosine_matrix(out.video, out.audio).cpu()
    assert (cm - O.cosine_matrix(out.video.cpu(), out.audio.cpu())).abs().max() < 1e-5
    tri = pig.triplet.score_triplets(out.video.repeat(4, 1), out.audio.repeat(4, 1), torch.tensor([1., 1, 2, 2, 1, 1, 2, 2]), n_samples=3)
    assert tri.shape == (3,)


def test_static_image_encoder_c1_config():
    """BASELINE configs[0]: hparams_static.yaml, 8x64x64 frames + 1 s @ 16 kHz audio, batch 4 -- the reference's own
    CPU-runnable case, at its full size: embeddings and loss at SURVEY 8d's tolerances, every gradient present."""
    import os
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "hparams_static.yaml")))
    cfg["audio"]["pretrained"] = False
    ref, net = build_pair(cfg)
    ref.train(); net.train()
    batch = synthetic_batch(4, 8, 64, 16000)
    with torch.no_grad():
        V32, A32 = ref.encode_video(batch.video), ref.encode_audio(batch.audio)
        loss32 = ref.loss(V32, A32).item()
        sd = copy.deepcopy(ref.state_dict())
        with torch.autocast("cpu", dtype=torch.bfloat16):
            V16 = ref.encode_video(batch.video).float()
        ref.load_state_dict(sd)
    gb = batch.to(DEV)
    bn_state = copy.deepcopy(net.video_encoder.image.state_dict())
    loss = net.training_step(gb, 0)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    used = [n for n, p in net.named_parameters() if not n.startswith("video_encoder.image.fc")]
    named = dict(net.named_parameters())
    assert all(named[n].grad is not None and torch.isfinite(named[n].grad).all() for n in used)
    assert net.video_encoder.image.fc.weight.grad is None
    net.video_encoder.image.load_state_dict(bn_state)
    with torch.no_grad():
        Vh = net.encode_video(gb.video).cpu()
        Ah = net.encode_audio(gb.audio).cpu()
    cos_v = F.cosine_similarity(Vh, V32, dim=1).min().item()
    cos_a = F.cosine_similarity(Ah, A32, dim=1).min().item()
    yard = F.cosine_similarity(V16, V32, dim=1).min().item()
    print(f"static C1: video min cos {cos_v:.6f} (torch bf16 autocast of the oracle {yard:.6f}) max-abs "
          f"{(Vh - V32).abs().max().item():.5f}; audio min cos {cos_a:.6f} max-abs {(Ah - A32).abs().max().item():.5f}; "
          f"loss {loss.item():.6f} vs {loss32:.6f}")
    assert (Vh.norm(dim=1) - 1).abs().max() < 1e-4
    assert cos_v >= 0.999 and (Vh - V32).abs().max().item() <= 2e-2
    assert cos_a >= 0.999 and (Ah - A32).abs().max().item() <= 2e-2
    assert abs(loss.item() - loss32) <= 5e-3
    assert 1 - cos_v <= 1.5 * (1 - yard) + 2e-4


@pytest.mark.parametrize("pooling", ["attention", "average"])
def test_full_false_conv_features_only(pooling):
    """`audio.full: false` (pig/models.py:86,105): torchaudio 0.9.1's `extract_features` returns the CONV feature
    extractor's output (512-d, no transformer), pooled and projected; the transformer is an unused
    parameter set.  Values and gradients against the oracle."""
    cfg = make_cfg()
    cfg["audio"]["full"] = False
    cfg["audio"]["pooling"] = pooling
    ref, net = build_pair(cfg)
    ref.train(); net.train()
    assert net.audio_encoder.n_features == 512
    batch = synthetic_batch(4, 4, 32, 16000)
    R = torch.randn(4, 512, generator=torch.Generator().manual_seed(1))
    Ar = ref.encode_audio(batch.audio)
    (Ar * R).sum().backward()
    A = net.encode_audio(batch.audio.to(DEV))
    (A * R.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert F.cosine_similarity(A.detach().cpu(), Ar.detach(), dim=1).min().item() >= 0.999
    assert (A.detach().cpu() - Ar.detach()).abs().max().item() <= 2e-2
    refp = dict(ref.audio_encoder.named_parameters())
    checked = 0
    for name, p in net.audio_encoder.named_parameters():
        pr = refp[name]
        if pr.grad is None:
            assert p.grad is None, f"{name}: the transformer is unused with full: false"
            continue
        assert p.grad is not None, name
        if pr.grad.norm() > 1e-6:
            assert rel(p.grad, pr.grad) <= 0.08, (name, rel(p.grad, pr.grad))
            checked += 1
    assert checked >= 8 and net.audio_encoder.audio.encoder.transformer.layers[0].attention.q_proj.weight.grad is None


def test_audio_dropout_and_layerdrop_train_mode():
    """Default torchaudio regularisation (p = 0.1, LayerDrop 0.1): stochastic in train mode, finite
    gradients for every trainable tensor, deterministic in eval mode."""
    import random
    cfg = make_cfg()
    net = pig.models.PeppaPig(cfg).to(DEV)
    b = synthetic_batch(4, 4, 32, 4000).to(DEV)
    net.train()
    torch.manual_seed(0)
    A1 = net.encode_audio(b.audio)
    A2 = net.encode_audio(b.audio)
    assert not torch.equal(A1, A2)
    (A1 * torch.randn_like(A1)).sum().backward()
    torch.cuda.synchronize()
    got = [p.grad is not None for p in net.audio_encoder.parameters()]
    assert sum(got) >= len(got) - 16 * 2        # LayerDrop may skip a couple of layers (16 tensors each)
    assert all(torch.isfinite(p.grad).all() for p in net.audio_encoder.parameters() if p.grad is not None)
    net.eval()
    with torch.no_grad():
        E1, E2 = net.encode_audio(b.audio), net.encode_audio(b.audio)
    # eval mode has no dropout; the only run-to-run noise is the float-atomic GroupNorm statistic of conv0
    assert (E1 - E2).abs().max().item() < 2e-3
    # dropout is unbiased: the train-mode embedding stays close to the eval-mode one
    assert F.cosine_similarity(A1.detach(), E1, dim=1).min().item() > 0.5


@pytest.mark.parametrize("pooling,project,T,Fd", [("average", True, 49, 28), ("average", False, 114, 512), ("last", True, 7, 28),
                                                  ("last", False, 5, 512), ("vavg", True, 3, 512), ("vavg", False, 2, 512)])
def test_remaining_pooling_heads_match_reference_semantics(pooling, project, T, Fd):
    """`average` / `last` audio pooling, VideoAveragePool, and `project: false` (pig/models.py:45-61, 97-109, 204-211):
    the HIP heads against the reference expressions evaluated by torch on the CPU, values and gradients.
    (AveragePool is nn.AdaptiveAvgPool2d((size, 1)) on a 3-D tensor: it averages the FEATURE axis and resamples time.)"""
    from peppa_amd import models as PM
    g = torch.Generator().manual_seed(T * 31 + Fd)
    B = 5
    x = torch.randn(B, T, Fd, generator=g)
    lin = torch.nn.Linear(Fd, 512) if project else torch.nn.Identity()
    R = torch.randn(B, 512 if project else Fd, generator=g)
    # reference expressions
    xr = x.clone().requires_grad_()
    if pooling == "average":
        pooled = torch.nn.AdaptiveAvgPool2d((Fd, 1))(xr).squeeze(dim=2)
    elif pooling == "last":
        pooled = xr[:, -1, :]
    else:   # VideoAveragePool on (B, 512, T, H, W) == mean over T of the spatial means
        pooled = xr.mean(dim=1)
    out_ref = F.normalize(lin(pooled), p=2, dim=1)
    (out_ref * R).sum().backward()
    # HIP heads
    xd = x.clone().to(DEV).requires_grad_()
    lind = copy.deepcopy(lin).to(DEV)
    if pooling == "average":
        head = PM.AveragePool(size=Fd)
    elif pooling == "last":
        head = PM.LastStep()
    else:
        head = PM.VideoAveragePool()
    out = PM._project_normalize(head(xd), lind)
    (out * R.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert (out.detach().cpu() - out_ref.detach()).abs().max().item() <= 2e-5
    assert rel(xd.grad, xr.grad) <= 2e-4
    if project:
        assert rel(lind.weight.grad, lin.weight.grad) <= 2e-4 and rel(lind.bias.grad, lin.bias.grad) <= 2e-4


def test_data_parallel_machinery_single_rank():
    """Embedding all-gather, early gradient hand-off from the towers (dist.grad_dict), bucket packing on the producing
    streams and the RCCL all-reduce, on ONE rank (identity collectives): gradients must equal those of the plain step.
    A pack that ran ahead of its producer (missing event wait) shows up as garbage here."""
    import os
    import socket
    import torch.distributed as dist
    from peppa_amd.dist import default_buckets
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    cfg = make_cfg()
    _, net = build_pair(cfg)
    net.train()
    batch = synthetic_batch(4, 4, 32, 4000).to(DEV)
    dist.init_process_group("nccl")
    try:
        os.environ["PEPPA_FORCE_DIST"] = "1"
        buckets = default_buckets(net, torch.device(DEV))
        net.zero_grad(set_to_none=True)
        loss_dp = net.training_step(batch, 0)
        loss_dp.backward()
        pushed = sum(len(b["pushed"]) for b in buckets.buckets)
        buckets.finish()
        torch.cuda.synchronize()
        g_dp = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        assert pushed > 200, f"only {pushed} gradients were handed over early"
        assert all(p.grad.data_ptr() != 0 for p in net.parameters() if p.grad is not None)
        buckets.close()
        os.environ["PEPPA_FORCE_DIST"] = "0"
        net.zero_grad(set_to_none=True)
        loss = net.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - loss_dp.item()) <= 2e-4   # (float atomics in the audio GroupNorm statistics)
        gmax = max(p.grad.abs().max().item() for p in net.parameters() if p.grad is not None)
        for n, p in net.named_parameters():
            if p.grad is None:
                assert n not in g_dp, n
                continue
            assert n in g_dp, f"{n}: no gradient on the data-parallel path"
            err = (g_dp[n] - p.grad).abs().max().item()
            # Two plain steps of this tiny bf16 train-mode-BatchNorm net already differ by 3-4 % of a tensor's largest
            # gradient (float atomics in the audio statistics perturb dV in the last bits and the trunk amplifies
            # that, DESIGN.md "Numerics"; tools/probe/dp_dbg.py); a mis-ordered pack gives O(1) errors or zeros.
            assert err <= 0.12 * max(p.grad.abs().max().item(), 1e-2 * gmax), f"{n}: {err}"
        # Deterministic mode: nothing is left to summation order, so the data-parallel path (early hand-off, packing on the
        # producing streams, RCCL all-reduce over one rank) must reproduce the plain step BIT FOR BIT (the 12 % above is
        # what float atomics force on the default mode)
        prev_det = H.set_deterministic(True)
        try:
            net.zero_grad(set_to_none=True)
            loss_a = net.training_step(batch, 0)
            loss_a.backward()
            torch.cuda.synchronize()
            g_plain = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
            os.environ["PEPPA_FORCE_DIST"] = "1"
            buckets = default_buckets(net, torch.device(DEV))
            net.zero_grad(set_to_none=True)
            loss_b = net.training_step(batch, 0)
            loss_b.backward()
            buckets.finish()
            torch.cuda.synchronize()
            buckets.close()
            os.environ["PEPPA_FORCE_DIST"] = "0"
            assert torch.equal(loss_a, loss_b)
            bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.equal(p.grad, g_plain[n])]
            assert not bad, f"deterministic mode: {len(bad)} gradients differ between the plain and the data-parallel step: {bad[:5]}"
        finally:
            H.set_deterministic(prev_det)
        # SyncBN option: every BatchNorm layer all-reduces its statistics rows through RCCL (identity with one rank)
        from peppa_amd import layers as PL
        from peppa_amd.dist import enable_sync_bn
        enable_sync_bn(True)
        assert PL.SYNC_BN_REDUCE is not None and PL.SYNC_BN_WORLD == 1
        net.zero_grad(set_to_none=True)
        loss_sync = net.training_step(batch, 0)
        loss_sync.backward()
        torch.cuda.synchronize()
        # (the all-reduced row is an fp32 sum where the local path accumulates partial rows in fp64: means differ in the
        # last bits, and this tiny train-mode-BatchNorm net amplifies that; the arithmetic itself is checked to 1e-5 in
        # test_kernels_gpu.py::test_sync_bn_two_ranks_in_one_process_match_the_global_batch)
        assert abs(loss_sync.item() - loss.item()) <= 3e-3
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    finally:
        from peppa_amd.dist import enable_sync_bn
        enable_sync_bn(False)
        os.environ["PEPPA_FORCE_DIST"] = "0"
        dist.destroy_process_group()


def test_optimizer_keeps_a_step_count_per_tensor():
    """Tensors that miss steps (LayerDrop leaves a skipped layer without gradients) lag behind in the warm-up schedule,
    exactly as in the reference, although every tensor of a step goes through ONE fused launch."""
    g = torch.Generator().manual_seed(9)
    shapes = [(7,), (130, 70), (3, 5, 2), (70000,), (1,)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]
    cpu_params = [p.detach().cpu().clone() for p in params]
    kw = dict(lr=1e-2, warmup=0.3, t_total=10)
    optim = pig.optimization.BertAdam(params, **kw)
    st = {}
    skip = {1: {0, 3}, 2: {3}, 4: {1, 2, 4}}          # step -> tensors without a gradient in that step
    for step in range(6):
        grads = []
        for i, p in enumerate(params):
            if i in skip.get(step, ()):
                p.grad = None
                grads.append(None)
            else:
                gr = torch.randn(*shapes[i], generator=g) * (5.0 if i == 1 else 0.2)
                p.grad = gr.to(DEV)
                grads.append(gr)
        optim.step()
        O.bertadam_step(cpu_params, grads, st, **kw)
        torch.cuda.synchronize()
        for i, (p, pc) in enumerate(zip(params, cpu_params)):
            d = (p.detach().cpu() - pc).abs().max().item()
            assert d <= 1e-6 + 1e-5 * pc.abs().max().item(), (i, step, d)
    assert [optim.state[p]["step"] for p in params] == [5, 5, 5, 4, 5]
    assert optim.get_lr()[3] != optim.get_lr()[0]


def test_fused_batchnorm_apply_leaves_the_video_tower_unchanged():
    """video.FUSE_BN_APPLY (the layer-1 temporal convs and their weight gradients apply the preceding BatchNorm + ReLU on
    their own LDS windows, the activated mid tensors are never written): embeddings bit-identical to the unfused run,
    gradients equal up to the fp32 atomics of the weight-gradient kernels; and the fused run really skipped the passes."""
    cfg = make_cfg()
    torch.manual_seed(3)
    net = pig.models.PeppaPig(cfg).to(DEV).train()
    enc = net.video_encoder
    x = torch.rand(2, 3, 16, 64, 64, generator=torch.Generator().manual_seed(9)).to(DEV)     # layer 1: 32 x 32 positions
    calls = {"n": 0}
    real_apply = H.bn_apply

    def counting_apply(*a, **k):
        calls["n"] += 1
        return real_apply(*a, **k)

    outs = {}
    state = copy.deepcopy(net.state_dict())       # (BatchNorm running statistics move with every forward)
    try:
        H.bn_apply = counting_apply
        for fuse in (False, "again", True):      # ("again": a second unfused run, the yardstick for the float atomics)
            net.load_state_dict(state)
            PV.FUSE_BN_APPLY = fuse is True
            calls["n"] = 0
            net.zero_grad(set_to_none=True)
            v = enc(x)
            (v * torch.linspace(-1, 1, v.numel(), device=DEV).view_as(v)).sum().backward()
            torch.cuda.synchronize()
            outs[fuse] = (v.detach().clone(), {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None},
                          calls["n"])
    finally:
        H.bn_apply = real_apply
        PV.FUSE_BN_APPLY = True
    assert outs[True][2] == outs[False][2] - 5, (outs[True][2], outs[False][2])     # the stem's first unit + four mid units of layer 1
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs["again"][0], outs[False][0])
    for n, g0 in outs[False][1].items():
        g1, g2 = outs[True][1][n], outs["again"][1][n]
        scale = g0.abs().max().item() + 1e-12
        noise = (g2 - g0).abs().max().item()
        assert (g1 - g0).abs().max().item() <= 4 * noise + 1e-4 * scale, (n, (g1 - g0).abs().max().item(), noise, scale)
