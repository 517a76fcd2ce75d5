"""Deterministic mode (VERDICT r2 item 7 / missing 3): with pp_set_option("deterministic", 1) every sum that crosses
workgroups is ordered -- weight gradients through per-split slabs, the wav2vec2 conv0 statistics / weight-norm / column sums
by one workgroup per sum, BertAdam's norms and the loss over per-block partials, head GEMMs without split-K -- so two runs
of the same step agree BIT FOR BIT: loss, every gradient, every updated parameter.  Without the mode they do not (float
atomics), which is why the data-parallel comparisons of tests/test_model_gpu.py carry per-cent bounds."""
import copy
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")


def _net(cfg):
    import pig.models
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "layer_drop"):
            m.layer_drop = 0.0
    return net.cuda().train()


def _cfg():
    from pig.execution import default_config
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    return cfg


def _run(net, state, batch, steps=2):
    """`steps` optimizer steps from `state`; returns (losses, gradients of the last step, final parameters)."""
    net.load_state_dict(state)
    opt = net.configure_optimizers()
    losses = []
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = net.training_step(batch, i)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    return losses, grads, {k: v.detach().clone() for k, v in net.state_dict().items()}


@pytest.fixture()
def deterministic():
    from peppa_amd import hip as H
    prev = H.set_deterministic(True)
    yield
    H.set_deterministic(prev)


@pytest.mark.parametrize("shape", [(4, 4, 32, 4000), (2, 16, 112, 36800)])
def test_two_runs_of_the_same_steps_are_bitwise_identical(deterministic, shape):
    """Toy shape and the real clip geometry at batch 2 (window / sliding-window / temporal-window kernels, split weight
    gradients, T = 114 attention): two optimizer steps, run twice from one state."""
    from peppa_amd.data import synthetic_batch
    B, frames, size, samples = shape
    net = _net(_cfg())
    state = copy.deepcopy(net.state_dict())
    batch = synthetic_batch(B, frames, size, samples).to("cuda")
    a = _run(net, state, batch)
    b = _run(net, state, batch)
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0])), (a[0], b[0])
    bad = [n for n in a[1] if not torch.equal(a[1][n], b[1][n])]
    assert not bad, f"{len(bad)} of {len(a[1])} gradients differ between two runs, e.g. {bad[:5]}"
    bad = [n for n in a[2] if not torch.equal(a[2][n], b[2][n])]
    assert not bad, f"{len(bad)} parameters / buffers differ after two optimizer steps, e.g. {bad[:5]}"
    assert len(a[1]) > 300 and all(torch.isfinite(g).all() for g in a[1].values())


def test_the_mode_changes_results_only_within_rounding(deterministic):
    """Same step with and without the mode: the ordered sums are the same sums (differences at fp32 summation-order level
    for well-conditioned tensors)."""
    from peppa_amd import hip as H
    from peppa_amd.data import synthetic_batch
    net = _net(_cfg())
    state = copy.deepcopy(net.state_dict())
    batch = synthetic_batch(4, 4, 32, 4000).to("cuda")
    det = _run(net, state, batch, steps=1)
    H.set_deterministic(False)
    plain = _run(net, state, batch, steps=1)
    H.set_deterministic(True)
    # measured spread of this difference over runs of the plain (atomic) mode: 0.6e-4 ... 2.5e-4 (tools/probe/det_vs_plain.py;
    # bf16 activations turn a last-bit difference of an fp32 sum into 2^-9 relative steps downstream); SURVEY 8d's bf16
    # loss tolerance against the oracle is 5e-3
    assert abs(det[0][0].item() - plain[0][0].item()) <= 1e-3
    for n in ("audio_encoder.project.weight", "video_encoder.project.weight", "audio_encoder.audio.encoder.transformer.layers.11.feed_forward.output_dense.weight"):
        g0, g1 = det[1][n], plain[1][n]
        assert (g0 - g1).norm().item() <= 0.1 * g1.norm().item() + 1e-12, n     # (the hinge loss at random init amplifies last bits: test_model_gpu.py)


def test_weight_gradient_slabs_match_the_atomic_path(deterministic):
    """The three split weight-gradient kernels (generic gather, sliding window, temporal window) + the fused bias gradient:
    slabs + ordered pass against the atomic path (fp32 summation order only) and bitwise repeatable."""
    from peppa_amd import hip as H, layers as L
    g = torch.Generator().manual_seed(3)
    B = 2
    cases = [("spatial sw", 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 28, 28)),
             ("temporal tw", 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 28, 28)),
             ("strided generic", 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), (8, 28, 28))]
    for name, Ci, Co, k, s, p, thw in cases:
        geom = L.ConvGeom(B, thw, Ci, Co, k, s, p)
        x = torch.randn(geom.Min, geom.in_cstride, generator=g).to(torch.bfloat16).cuda()
        dy = torch.randn(geom.M, geom.out_cstride, generator=g).to(torch.bfloat16).cuda()
        d1 = L.conv_wgrad_raw(x, dy, geom).clone()
        d2 = L.conv_wgrad_raw(x, dy, geom).clone()
        H.set_deterministic(False)
        ref = L.conv_wgrad_raw(x, dy, geom).clone()
        H.set_deterministic(True)
        torch.cuda.synchronize()
        assert torch.equal(d1, d2), name
        assert (d1 - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-6, name
    M, N, K = 3000, 768, 512
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    a = [t.clone() for t in L.linear_wgrad(x, dy, M, N, K)]
    b = [t.clone() for t in L.linear_wgrad(x, dy, M, N, K)]
    H.set_deterministic(False)
    r = [t.clone() for t in L.linear_wgrad(x, dy, M, N, K)]
    H.set_deterministic(True)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert (a[0] - r[0]).abs().max().item() <= 1e-4 * r[0].abs().max().item() and (a[1] - r[1]).abs().max().item() <= 1e-4 * r[1].abs().max().item()


def test_layernorm_backward_is_reproducible_beside_the_register_staged_kernels(deterministic):
    """pp_layernorm_bwd on fixed inputs while a second stream runs the generic weight-gradient kernel.  The kernels that
    stage their operands through registers share CUs with it, and with such a neighbour about one launch in ten used to return
    a row of dx computed from slightly different sums -- as long as the library was built with packed-FP32 instructions
    (the SLP vectoriser's v_pk_fma_f32 ...); peppa_amd/build.py passes -fno-slp-vectorize since, and every launch returns
    the same bits (tools/probe/ln_vs_kernels.py, ln_variants.sh)."""
    from peppa_amd import layers as L
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    rows, D = 228, 768
    ln = torch.nn.LayerNorm(D).to(dev)
    x = torch.randn(rows, D, device=dev, generator=g).to(torch.bfloat16)
    dy = (torch.randn(rows, D, device=dev, generator=g) * 1e-4).to(torch.bfloat16)
    _, saved = L.layernorm_fwd(x, ln)
    ref = L.layernorm_bwd(dy, x, ln, saved)[0].clone()
    M = 64 * 114
    xa = torch.randn(M, 768, device=dev, generator=g).to(torch.bfloat16)
    da = torch.randn(M, 3072, device=dev, generator=g).to(torch.bfloat16)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    bad = torch.zeros((), device=dev, dtype=torch.int32)
    for _ in range(60):
        with torch.cuda.stream(side):
            L.linear_wgrad(xa, da, M, 3072, 768)
        for _ in range(8):
            bad += (L.layernorm_bwd(dy, x, ln, saved)[0] != ref).any().to(torch.int32)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, f"{int(bad.item())} of 480 launches returned a different dx"
