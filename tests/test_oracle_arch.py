"""Oracle architecture pins: parameter counts / stage shapes and an independent
wav2vec2 implementation (HF transformers, built from a local config; nothing is fetched)."""
import os
import pytest
import torch
from oracle import audio as OA, video as OV, model as OM

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")


def _count(m):
    return sum(p.numel() for p in m.parameters())


def test_r2plus1d_counts_and_shapes():
    torch.manual_seed(0)
    net = OV.VideoResNet18("r2plus1d_18")
    assert _count(net) == 31505325
    assert _count(net) - _count(net.fc) == 31300125
    mids = [net.layer1[0].conv1[0][0].out_channels, net.layer2[0].conv1[0][0].out_channels,
            net.layer2[1].conv1[0][0].out_channels, net.layer3[0].conv1[0][0].out_channels,
            net.layer3[1].conv1[0][0].out_channels, net.layer4[0].conv1[0][0].out_channels,
            net.layer4[1].conv1[0][0].out_channels]
    assert mids == [144, 230, 288, 460, 576, 921, 1152]
    net.eval()
    x = torch.rand(1, 3, 4, 32, 32)
    with torch.no_grad():
        s = net.stem(x); l1 = net.layer1(s); l2 = net.layer2(l1); l3 = net.layer3(l2); l4 = net.layer4(l3)
    assert s.shape == (1, 64, 4, 16, 16) and l1.shape == (1, 64, 4, 16, 16)
    assert l2.shape == (1, 128, 2, 8, 8) and l3.shape == (1, 256, 1, 4, 4) and l4.shape == (1, 512, 1, 2, 2)
    assert "layer1.0.conv1.0.0.weight" in net.state_dict()
    assert "layer2.0.downsample.1.running_var" in net.state_dict()


def test_other_trunks_build():
    assert _count(OV.VideoResNet18("r3d_18")) == 33371472
    assert _count(OV.VideoResNet18("mc3_18")) == 11695440
    assert _count(OV.ResNet18()) == 11689512


def test_wav2vec_counts_and_frames():
    torch.manual_seed(0)
    m = OA.wav2vec2_base(28)
    assert _count(m.feature_extractor) == 4200448
    assert _count(m.encoder.feature_projection) == 395008
    assert _count(m.encoder.transformer) == 89775488
    assert _count(m) == 94392476  # fe 4 200 448 + proj 395 008 + transformer 89 775 488 + readout 21 532
    assert [OA.n_frames(n) for n in (16000, 36800, 73600, 101429)] == [49, 114, 229, 316]
    sd = m.state_dict()
    assert "encoder.transformer.layers.0.attention.k_proj.weight" in sd
    assert "encoder.transformer.pos_conv_embed.conv.weight_g" in sd
    assert sd["encoder.transformer.pos_conv_embed.conv.weight_g"].shape == (1, 1, 128)


def test_wav2vec_matches_hf_implementation():
    tr = pytest.importorskip("transformers")
    from transformers import Wav2Vec2Config, Wav2Vec2Model
    torch.manual_seed(0)
    cfg = Wav2Vec2Config(feat_extract_norm="group", do_stable_layer_norm=False, conv_bias=False,
                         hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                         intermediate_size=3072, num_conv_pos_embeddings=128,
                         num_conv_pos_embedding_groups=16, vocab_size=32)
    hf = Wav2Vec2Model(cfg).eval()
    ours = OA.wav2vec2_base(28).eval()
    hsd = hf.state_dict()
    osd = ours.state_dict()
    new = {}
    for k in osd:
        if k.startswith("feature_extractor.conv_layers."):
            new[k] = hsd[k]
        elif k.startswith("encoder.feature_projection."):
            new[k] = hsd[k.replace("encoder.feature_projection.", "feature_projection.")]
        elif k.startswith("encoder.transformer.pos_conv_embed.conv."):
            leaf = k.split(".")[-1]
            cands = {"weight_g": ["encoder.pos_conv_embed.conv.weight_g",
                                  "encoder.pos_conv_embed.conv.parametrizations.weight.original0"],
                     "weight_v": ["encoder.pos_conv_embed.conv.weight_v",
                                  "encoder.pos_conv_embed.conv.parametrizations.weight.original1"],
                     "bias": ["encoder.pos_conv_embed.conv.bias"]}[leaf]
            new[k] = next(hsd[c] for c in cands if c in hsd)
        elif k.startswith("encoder.transformer.layer_norm."):
            new[k] = hsd[k.replace("encoder.transformer.layer_norm.", "encoder.layer_norm.")]
        elif k.startswith("encoder.transformer.layers."):
            h = k.replace("encoder.transformer.layers.", "encoder.layers.")
            h = h.replace("feed_forward.intermediate_dense", "feed_forward.intermediate_dense")
            new[k] = hsd[h]
        elif k.startswith("encoder.readout."):
            new[k] = osd[k]
        else:
            raise KeyError(k)
    ours.load_state_dict(new)
    g = torch.Generator().manual_seed(5)
    wave = 0.1 * torch.randn(2, 16000, generator=g)
    with torch.no_grad():
        ref = hf(wave).last_hidden_state
        feats, _ = ours.extract_features(wave)
        hid = ours.encoder.transformer(ours.encoder.feature_projection(feats))
        ref_feats = hf.feature_extractor(wave).transpose(1, 2)
    assert ref.shape == hid.shape == (2, 49, 768)
    assert (feats - ref_feats).abs().max().item() < 1e-5
    assert (hid - ref).abs().max().item() < 2e-4
    assert torch.nn.functional.cosine_similarity(hid.flatten(), ref.flatten(), dim=0).item() > 0.999999


def test_resnet18_matches_hf_implementation():
    """The static encoder's trunk (pig/models.py:156-200: torchvision resnet18) against an independent implementation,
    HF transformers' ResNetModel built from a local config (basic layers, depths 2-2-2-2, widths 64-512; nothing is fetched):
    same parameter count as torchvision's resnet18 without `fc` (11 176 512) and the same activations in eval AND train mode
    (batch statistics) once the restatement's weights are mapped onto it.  (No second implementation of the r2plus1d / r3d /
    mc3 video trunks exists in this container: those stay pinned by parameter counts, mid-plane widths and stage shapes.)"""
    pytest.importorskip("transformers")
    from transformers import ResNetConfig, ResNetModel
    torch.manual_seed(0)
    ours = OV.ResNet18()
    hf = ResNetModel(ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2],
                                  layer_type="basic", hidden_act="relu", downsample_in_first_stage=False))
    assert _count(hf) == _count(ours) - _count(ours.fc) == 11176512
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():   # non-trivial BatchNorm parameters and running statistics
        for m in ours.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.bias.shape, generator=g))
    osd, new = ours.state_dict(), {}

    def put(hf_prefix, conv, bn):
        new[hf_prefix + ".convolution.weight"] = osd[conv + ".weight"]
        for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            new[hf_prefix + ".normalization." + leaf] = osd[bn + "." + leaf]

    put("embedder.embedder", "conv1", "bn1")
    for si in range(4):
        for bi in range(2):
            o, h = f"layer{si + 1}.{bi}", f"encoder.stages.{si}.layers.{bi}"
            put(h + ".layer.0", o + ".conv1", o + ".bn1")
            put(h + ".layer.1", o + ".conv2", o + ".bn2")
            if f"{o}.downsample.0.weight" in osd:
                put(h + ".shortcut", o + ".downsample.0", o + ".downsample.1")
    missing, unexpected = hf.load_state_dict(new, strict=True)
    x = torch.rand(3, 3, 64, 64, generator=g)

    def trunk(net, x):
        y = net.maxpool(net.relu(net.bn1(net.conv1(x))))
        return net.layer4(net.layer3(net.layer2(net.layer1(y))))

    for mode in ("eval", "train"):
        getattr(ours, mode)(); getattr(hf, mode)()
        sd_o, sd_h = {k: v.clone() for k, v in ours.state_dict().items()}, {k: v.clone() for k, v in hf.state_dict().items()}
        with torch.no_grad():
            a = trunk(ours, x)
            b = hf(x).last_hidden_state
        ours.load_state_dict(sd_o); hf.load_state_dict(sd_h)    # (train mode moved the running statistics)
        assert a.shape == b.shape == (3, 512, 2, 2)
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item()), (mode, (a - b).abs().max().item())


def test_peppa_oracle_step_runs_c1_shape():
    cfg = {"margin": 0.2,
           "video": {"pretrained": False, "project": True, "version": "r2plus1d_18", "pooling": "attention"},
           "audio": {"path": None, "pretrained": False, "freeze_feature_extractor": False,
                     "freeze_encoder_layers": None, "pooling": "attention", "full": True}}
    torch.manual_seed(0)
    net = OM.PeppaPigOracle(cfg)
    v, a = OM.synthetic_batch(2, 4, 32, 4000)
    loss = net.training_loss(v, a)
    loss.backward()
    assert loss.ndim == 0 and torch.isfinite(loss)
    assert net.video_encoder.video.fc.weight.grad is None  # unused
    assert net.audio_encoder.project.weight.shape == (512, 28)
