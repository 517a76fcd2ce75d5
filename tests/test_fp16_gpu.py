"""The fp16 build of the HIP path (libpeppa_hip_f16.so: the same sources with IEEE-half operands) -- the reference's own
`precision: 16` (hparams_base.yaml:45) and BASELINE configs[4] -- with GradScaler-equivalent dynamic loss scaling
(peppa_amd/amp.py).  SURVEY 8d tolerances for fp16: embeddings max-abs <= 5e-3, cosine >= 0.999, loss |d| <= 1e-3."""
import copy
import math
import warnings
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")

from peppa_amd import hip as H
from peppa_amd import layers as L
from peppa_amd.amp import GradScaler

DEV = "cuda"


@pytest.fixture
def fp16():
    prev = H.set_precision("fp16")
    yield
    H.set_precision(prev)


def test_both_libraries_load_and_report_their_operand_type():
    from peppa_amd import _lib
    assert _lib.lib("bf16").pp_dtype() == 0 and _lib.lib("fp16").pp_dtype() == 1
    assert H.act16() == torch.bfloat16
    prev = H.set_precision("fp16")
    assert H.act16() == torch.float16 and prev == "bf16"
    H.set_precision(prev)
    with pytest.raises(ValueError):
        H.set_precision("fp8")


@pytest.mark.parametrize("case", [
    (64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2, 3, 60, 56),    # window kernel (spatial)
    (144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2, 8, 16, 16),    # temporal window kernel
    (64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), 2, 3, 10, 12),    # gather kernel, strided, 230 -> 240 channels
    (32, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, 4, 6, 6),
])
def test_conv_kernels_fp16(fp16, case):
    """Forward / data gradient / weight gradient on IEEE-half operands against fp32 torch on the same (fp16-rounded)
    operands: 3 more mantissa bits than bf16, so the tolerance is 8x tighter than in test_kernels_gpu.py."""
    Ci, Co, k, s, p, B, T, Hh, W = case
    rh = lambda t: t.to(torch.float16).float()
    g = torch.Generator().manual_seed(Ci * 7 + Co)
    x = rh(torch.randn(B, Ci, T, Hh, W, generator=g))
    w = rh(torch.randn(Co, Ci, *k, generator=g) / math.sqrt(Ci * k[0] * k[1] * k[2]))
    x.requires_grad_(); w.requires_grad_()
    y_ref = F.conv3d(x, w, stride=s, padding=p)
    dy = rh(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)

    def to_cl(t, cp):
        C = t.shape[1]
        v = t.detach().permute(0, 2, 3, 4, 1).reshape(-1, C)
        out = torch.zeros(v.shape[0], cp)
        out[:, :C] = v
        return out.to(torch.float16).to(DEV)

    def from_cl(y, thw, C):
        return y.float().cpu()[:, :C].reshape(B, *thw, C).permute(0, 4, 1, 2, 3)

    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    xc = to_cl(x, geom.in_cstride)
    wf, wd = L.prep_conv_weights(w.detach().to(DEV).contiguous(), geom)
    assert wf.dtype == torch.float16
    y, _ = L.conv_fwd(xc, geom, wf, stats=True)
    dyc = to_cl(dy, geom.out_cstride)
    dx = L.conv_dgrad(dyc, geom, wd)
    dw = L.conv_wgrad(xc, dyc, geom, w.shape)
    torch.cuda.synchronize()
    for name, got, want in (("fwd", from_cl(y, geom.out_thw, Co), y_ref.detach()), ("dgrad", from_cl(dx, (T, Hh, W), Ci), x.grad),
                            ("wgrad", dw.cpu(), w.grad)):
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err <= 3e-3, (name, err)


def test_fp16_parity_at_real_shape():
    """configs[1] geometry at batch 8 on the fp16 library against the fp32 oracle."""
    from parity_c2_report import report
    rep = report(precision="fp16", blocks=True)
    assert rep["video_cos"] >= 0.999 and rep["video_maxabs"] <= 5e-3, (rep["video_cos"], rep["video_maxabs"])
    assert rep["audio_cos"] >= 0.999 and rep["audio_maxabs"] <= 5e-3, (rep["audio_cos"], rep["audio_maxabs"])
    assert rep["dloss"] <= 1e-3
    for name, fwd, dx, dw in rep["blocks"]:
        assert fwd <= 2e-3 and dx <= 0.06 and dw <= 0.06, (name, fwd, dx, dw)
    assert rep["grads"]["audio"][0] <= 5e-3


def test_fp16_long_clips_config5():
    """BASELINE configs[4]: 32 frames of 112x112 + 73 600 audio samples (229 wav2vec2 frames), fp16, batch 4."""
    from parity_c2_report import report
    rep = report(batch=4, frames=32, size=112, samples=73600, precision="fp16", blocks=False)
    assert rep["video_cos"] >= 0.999 and rep["video_maxabs"] <= 5e-3, (rep["video_cos"], rep["video_maxabs"])
    assert rep["audio_cos"] >= 0.999 and rep["audio_maxabs"] <= 5e-3
    assert rep["dloss"] <= 1e-3


def test_grad_scaler_semantics():
    """torch.cuda.amp.GradScaler's contract: unscale in place, skip the step on inf / nan and back off, grow after
    `growth_interval` clean steps; checked against the same arithmetic done by hand."""
    ps = [torch.nn.Parameter(torch.randn(n, device=DEV)) for n in (5, 70000, 33)]
    opt = torch.optim.SGD(ps, lr=0.1)
    sc = GradScaler(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2)
    g = torch.Generator().manual_seed(0)
    true = [torch.randn(p.shape, generator=g).to(DEV) for p in ps]
    loss = sum((p * t).sum() for p, t in zip(ps, true))
    sc.scale(loss).backward()
    assert torch.allclose(ps[1].grad, true[1] * 1024.0)
    before = [p.detach().clone() for p in ps]
    sc.step(opt); sc.update()
    for p, b, t in zip(ps, before, true):
        assert torch.allclose(p.detach(), b - 0.1 * t, atol=1e-6)        # stepped with the UNSCALED gradient
    assert sc.get_scale() == 1024.0                                       # one clean step of two
    opt.zero_grad(set_to_none=True)
    sc.scale(sum((p * t).sum() for p, t in zip(ps, true))).backward()
    sc.step(opt); sc.update()
    assert sc.get_scale() == 2048.0                                       # grew after growth_interval clean steps
    # an overflow anywhere: the step is skipped for every tensor, the scale backs off
    opt.zero_grad(set_to_none=True)
    sc.scale(sum((p * t).sum() for p, t in zip(ps, true))).backward()
    ps[1].grad[12345] = float("inf")
    before = [p.detach().clone() for p in ps]
    sc.step(opt); sc.update()
    assert all(torch.equal(p.detach(), b) for p, b in zip(ps, before)) and sc.skipped_steps == 1
    assert sc.get_scale() == 1024.0
    opt.zero_grad(set_to_none=True)
    sc.scale(sum((p * t).sum() for p, t in zip(ps, true))).backward()
    ps[2].grad[3] = float("nan")
    sc.step(opt); sc.update()
    assert sc.skipped_steps == 2 and sc.get_scale() == 512.0
    sd = sc.state_dict()
    sc2 = GradScaler()
    sc2.load_state_dict(sd)
    assert sc2.get_scale() == 512.0


def test_grad_scaler_with_bertadam_skips_on_the_device_and_keeps_the_step_counters():
    """With BertAdam the overflow decision never visits the host on the step path: the fused launch is a no-op when the
    flag is set, and the per-tensor step counters (the warm-up schedule of pig/optimization.py:160-170) are corrected one
    step later.  Trajectory against the oracle's BertAdam fed only the clean steps."""
    import pig.optimization
    from oracle import model as O
    g = torch.Generator().manual_seed(4)
    shapes = [(9,), (300, 70), (4, 5, 6)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]
    cpu = [p.detach().cpu().clone() for p in params]
    kw = dict(lr=1e-2, warmup=0.3, t_total=10)
    opt = pig.optimization.BertAdam(params, **kw)
    sc = GradScaler(init_scale=256.0, growth_interval=1000)
    st = {}
    overflow_at = {2, 3}
    for step in range(6):
        grads = [torch.randn(*s, generator=g) * 0.3 for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = (gr * sc.get_scale()).to(DEV)            # what backward of the scaled loss leaves
        if step in overflow_at:
            params[1].grad[5, 7] = float("inf")
        before = [p.detach().clone() for p in params]
        sc.step(opt)
        sc.update()
        if step in overflow_at:
            assert all(torch.equal(p.detach(), b) for p, b in zip(params, before))
        else:
            O.bertadam_step(cpu, grads, st, **kw)
            for p, pc in zip(params, cpu):
                assert (p.detach().cpu() - pc).abs().max().item() <= 1e-6 + 1e-5 * pc.abs().max().item(), step
    assert sc.skipped_steps == 2 and sc.get_scale() == 64.0
    assert [opt.state[p]["step"] for p in params] == [4, 4, 4]


def test_overflow_on_the_final_step_is_settled_by_flush():
    """the step-counter correction of a device-skipped step used to wait for the NEXT scaler call; a run whose
    last step overflowed kept `step` one too high (warm-up schedule off by one on resume).  flush() / a checkpoint of the
    optimizer state with the scaler settle it."""
    import pig.optimization
    from peppa_amd.checkpoint import optimizer_state
    g = torch.Generator().manual_seed(5)
    params = [torch.nn.Parameter(torch.randn(40, 30, generator=g).to(DEV)), torch.nn.Parameter(torch.randn(7, generator=g).to(DEV))]
    for use_checkpoint in (False, True):
        opt = pig.optimization.BertAdam(params, lr=1e-2, warmup=0.3, t_total=10)
        sc = GradScaler(init_scale=128.0)
        for step in range(3):
            for p in params:
                p.grad = (torch.randn(p.shape, generator=g) * sc.get_scale()).to(DEV)
            if step == 2:
                params[0].grad[3, 3] = float("nan")          # the LAST step overflows
            sc.step(opt)
            sc.update()
        assert [opt.state[p]["step"] for p in params] == [3, 3]      # not settled yet: the flag is still in flight
        if use_checkpoint:
            state = optimizer_state(opt, sc)
            assert sorted(st["step"] for st in state["state"].values()) == [2, 2]
        else:
            sc.flush()
        assert [opt.state[p]["step"] for p in params] == [2, 2] and sc.skipped_steps == 1


def test_fp16_training_steps_with_loss_scaling():
    """A few optimizer steps of the whole model in fp16 under the built-in Trainer (precision="fp16"): BertAdam steps on
    unscaled gradients, nothing overflows into the weights, the loss stays finite and equals the bf16 run's within noise."""
    import pig.models
    from pig.execution import default_config
    from peppa_amd.trainer import Trainer
    from peppa_amd.data import synthetic_batch
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    batch = synthetic_batch(4, 4, 32, 4000).to(DEV)

    class Data:
        def train_dataloader(self):
            return iter([batch] * 6)

    losses = {}
    for prec in ("fp16", "bf16"):
        torch.manual_seed(0)
        net = pig.models.PeppaPig(cfg).to(DEV)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if hasattr(m, "layer_drop"):
                m.layer_drop = 0.0
        tr = Trainer(accumulate_grad_batches=2, precision=prec)
        seen = []
        orig = net.training_step
        net.training_step = lambda b, i, orig=orig, seen=seen: (lambda l: (seen.append(l.detach()), l)[1])(orig(b, i))
        tr.fit(net, Data())
        torch.cuda.synchronize()
        losses[prec] = [float(l) for l in seen]
        assert tr.global_step == 3 and all(math.isfinite(v) for v in losses[prec])
        assert all(torch.isfinite(p).all() for p in net.parameters())
        if prec == "fp16":
            assert net.precision == "fp16" and tr.scaler is not None
            assert tr.scaler.get_scale() > 0 and tr.scaler.skipped_steps <= 2
            assert next(iter(net.video_encoder.video.parameters())).dtype == torch.float32     # fp32 masters
        else:
            assert tr.scaler is None
    assert abs(losses["fp16"][0] - losses["bf16"][0]) <= 0.05
