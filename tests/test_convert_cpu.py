"""Pretrained-weight import (pig/models.py:70-74, 123-127): the fairseq-0.10.2-named wav2vec2 checkpoint
is renamed to the torchaudio-0.9.1 names the towers use, without fairseq or torchaudio.  Tested with a SYNTHETIC
fairseq-named file (random weights, the file layout fairseq writes: {"args": Namespace, "model": state_dict, ...}) round-
tripped through the oracle's forward: the imported tower must compute exactly what the tower the file was made from does,
the pre-training heads must be dropped and the 28-way readout must be the freshly initialised one."""
import argparse
import os

import pytest
import torch

from oracle import audio as OA
from peppa_amd import convert as CV


def _to_fairseq(name):
    """inverse of the import, written independently of convert._RULES (torchaudio 0.9.1 -> fairseq 0.10.2 names)"""
    n = name
    if n.startswith("feature_extractor.conv_layers."):
        idx, rest = n[len("feature_extractor.conv_layers."):].split(".", 1)
        return f"feature_extractor.conv_layers.{idx}." + rest.replace("conv.", "0.").replace("layer_norm.", "2.")
    table = [("encoder.feature_projection.projection.", "post_extract_proj."), ("encoder.feature_projection.layer_norm.", "layer_norm."),
             ("encoder.transformer.pos_conv_embed.conv.", "encoder.pos_conv.0."), ("encoder.transformer.layer_norm.", "encoder.layer_norm.")]
    for a, b in table:
        if n.startswith(a):
            return b + n[len(a):]
    if n.startswith("encoder.transformer.layers."):
        idx, rest = n[len("encoder.transformer.layers."):].split(".", 1)
        rest = (rest.replace("attention.", "self_attn.") if rest.startswith("attention.") else
                rest.replace("feed_forward.intermediate_dense.", "fc1.").replace("feed_forward.output_dense.", "fc2."))
        if rest.startswith("layer_norm."):
            rest = "self_attn_" + rest
        return f"encoder.layers.{idx}.{rest}"
    raise AssertionError(name)


def _fairseq_file(tmp_path, model, fused=False, **arg_overrides):
    sd = {}
    for k, v in model.state_dict().items():
        if k.startswith("encoder.readout."):
            continue                                   # a pre-training checkpoint has no output head
        sd[_to_fairseq(k)] = v.clone()
    if fused:                                          # fairseq < 0.9 stored q, k, v stacked
        for i in range(12):
            for wb in ("weight", "bias"):
                q, k, v = (sd.pop(f"encoder.layers.{i}.self_attn.{n}_proj.{wb}") for n in "qkv")
                sd[f"encoder.layers.{i}.self_attn.in_proj_{wb}"] = torch.cat([q, k, v], dim=0)
    g = torch.Generator().manual_seed(1)
    sd.update({"mask_emb": torch.randn(768, generator=g), "quantizer.vars": torch.randn(1, 640, 128, generator=g),
               "quantizer.weight_proj.weight": torch.randn(640, 512, generator=g), "quantizer.weight_proj.bias": torch.randn(640, generator=g),
               "project_q.weight": torch.randn(256, 256, generator=g), "project_q.bias": torch.randn(256, generator=g),
               "final_proj.weight": torch.randn(256, 768, generator=g), "final_proj.bias": torch.randn(256, generator=g)})
    args = dict(arch="wav2vec2", encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_attention_heads=12,
                conv_pos=128, conv_pos_groups=16, extractor_mode="default", conv_bias=False, layer_norm_first=False,
                conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] + [(512,2,2)]")
    args.update(arg_overrides)
    path = str(tmp_path / "wav2vec_small.pt")
    torch.save({"args": argparse.Namespace(**args), "model": sd, "optimizer_history": [{"num_updates": 400000}],
                "extra_state": {"epoch": 1}, "last_optimizer_state": None}, path)
    return path, sd


@pytest.fixture(scope="module")
def source():
    torch.manual_seed(3)
    return OA.wav2vec2_base(num_out=28, dropout=0.0, layer_drop=0.0).eval()


@pytest.mark.parametrize("fused", [False, True])
def test_synthetic_fairseq_checkpoint_round_trips_through_the_oracle(tmp_path, source, fused):
    path, fsd = _fairseq_file(tmp_path, source, fused=fused)
    from peppa_amd import audio as A
    torch.manual_seed(11)
    tower = A.wav2vec2_base(num_out=28)                        # the product's module tree (torchaudio names)
    fresh_readout = {k: v.clone() for k, v in tower.state_dict().items() if k.startswith("encoder.readout.")}
    dropped = CV.load_fairseq_wav2vec2(path, tower)
    assert sorted(dropped) == sorted(k for k in fsd if k.split(".")[0] in ("mask_emb", "quantizer", "project_q", "final_proj"))
    tsd = tower.state_dict()
    for k, v in source.state_dict().items():
        if k.startswith("encoder.readout."):
            assert torch.equal(tsd[k], fresh_readout[k]), k    # freshly initialised, as import_fairseq_model(num_out=28) leaves it
        else:
            assert torch.equal(tsd[k], v), k
    # through the oracle's forward: a second oracle tower loaded from the CONVERTED names computes the source's features
    twin = OA.wav2vec2_base(num_out=28, dropout=0.0, layer_drop=0.0).eval()
    twin.load_state_dict({**tsd, **{k: v for k, v in source.state_dict().items() if k.startswith("encoder.readout.")}})
    wave = 0.1 * torch.randn(2, 4000, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        a, _ = source(wave)
        b, _ = twin(wave)
    assert torch.equal(a, b)


def test_prefixed_finetuned_names_and_unknown_keys():
    sd = {"w2v_encoder.w2v_model.post_extract_proj.weight": torch.zeros(768, 512),
          "w2v_encoder.w2v_model.encoder.layers.3.fc1.bias": torch.zeros(3072),
          "w2v_encoder.proj.weight": torch.zeros(32, 768), "w2v_encoder.w2v_model.mask_emb": torch.zeros(768)}
    out, dropped = CV.fairseq_to_torchaudio(sd)
    assert set(out) == {"encoder.feature_projection.projection.weight",
                        "encoder.transformer.layers.3.feed_forward.intermediate_dense.bias", "encoder.readout.weight"}
    assert dropped == ["w2v_encoder.w2v_model.mask_emb"]
    with pytest.raises(CV.ConversionError, match="unexpected key"):
        CV.fairseq_to_torchaudio({"encoder.layers.0.some_new_module.weight": torch.zeros(1)})


def test_a_checkpoint_of_another_geometry_is_refused(tmp_path, source):
    from peppa_amd import audio as A
    path, _ = _fairseq_file(tmp_path, source, encoder_layers=24)
    with pytest.raises(CV.ConversionError, match="encoder_layers=24"):
        CV.load_fairseq_wav2vec2(path, A.wav2vec2_base(num_out=28))
    path, _ = _fairseq_file(tmp_path, source, extractor_mode="layer_norm")
    with pytest.raises(CV.ConversionError, match="extractor_mode"):
        CV.load_fairseq_wav2vec2(path, A.wav2vec2_base(num_out=28))


def test_encoder_reads_audio_path_like_the_reference(tmp_path, source):
    """`audio.path` + `pretrained: true` (every shipped yaml): the file at that path is imported; without it the error
    names it."""
    import pig.models
    path, _ = _fairseq_file(tmp_path, source)
    enc = pig.models.Wav2VecEncoder(path, pretrained=True, pooling="attention", full=True)
    assert torch.equal(enc.audio.state_dict()["encoder.transformer.layers.7.attention.k_proj.weight"],
                       source.state_dict()["encoder.transformer.layers.7.attention.k_proj.weight"])
    with pytest.raises(RuntimeError, match="no_such_file.pt"):
        pig.models.Wav2VecEncoder(str(tmp_path / "no_such_file.pt"), pretrained=True)


def test_torchvision_named_video_file_loads_into_the_trunk(tmp_path):
    """`mi355x: {video_weights}`: a file with torchvision's r2plus1d_18 names (as its model zoo ships it: no
    num_batches_tracked wrapper needed, optionally DataParallel-prefixed) loads into the trunk; a wrong shape is refused."""
    import pig.models
    from oracle import video as OV
    torch.manual_seed(2)
    zoo = OV.VideoResNet18("r2plus1d_18")                                   # torchvision's module tree restated (tests/test_oracle_arch.py)
    sd = {"module." + k: v for k, v in zoo.state_dict().items() if not k.endswith("num_batches_tracked")}
    path = str(tmp_path / "r2plus1d_18-91a641e6.pth")
    torch.save(sd, path)
    enc = pig.models.R3DEncoder(pretrained=True, version="r2plus1d_18", pooling="attention", weights=path)
    got = enc.video.state_dict()
    for k, v in zoo.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(got[k], v), k
    assert enc.norm_kind == "kinetics"
    sd["module.layer2.0.conv1.0.0.weight"] = torch.zeros(1, 2, 3)
    torch.save(sd, path)
    with pytest.raises(CV.ConversionError, match="layer2.0.conv1.0.0.weight"):
        pig.models.R3DEncoder(pretrained=True, version="r2plus1d_18", pooling="attention", weights=path)
