"""Pretrained-weight import without fairseq / torchaudio / torchvision (pig/models.py:70-74, 123-127).

The reference builds its audio tower as
    model, _, _ = fairseq.checkpoint_utils.load_model_ensemble_and_task([path])      # data/in/wav2vec/wav2vec_small.pt
    self.audio = torchaudio.models.wav2vec2.utils.import_fairseq_model(model[0], num_out=28)
i.e. fairseq 0.10.2 builds its `Wav2Vec2Model` from the checkpoint's `args`, and torchaudio 0.9.1 copies the weights
into its own module tree under its own names, DROPS the pre-training heads (quantizer, project_q, final_proj, mask_emb)
and leaves the `num_out`-way readout freshly initialised (a pre-training checkpoint has no `proj`).  Neither package is
installed here.  What that pair of calls does to the numbers is a pure renaming of a state dict, restated below from the
published layouts of the two releases; `fairseq_to_torchaudio` is host code and is tested with a synthetic fairseq-named
state dict against the oracle's forward (tests/test_convert_cpu.py).  The checkpoint file itself is read with the
restricted unpickler of peppa_amd.checkpoint (its `args` Namespace / `cfg` become inert stubs; nothing from the file runs).

Video: torchvision's `r2plus1d_18` / `r3d_18` / `mc3_18` / `resnet18` files already use the names of peppa_amd.video's
module tree; `video_state_dict` only strips wrappers ("module.", "state_dict") and checks shapes.
"""
import re

import torch

# fairseq Wav2Vec2Model keys that torchaudio's import drops (pre-training only)
_DROPPED = re.compile(r"^(mask_emb|quantizer\.|project_q\.|final_proj\.|target_glu\.|input_quantizer\.|project_inp\.|_float_tensor)")

# (fairseq pattern, torchaudio replacement); first match wins
_RULES = [
    # feature extractor.  extractor_mode "default": GroupNorm after conv 0 only, module index 2 of the block's Sequential
    (r"^feature_extractor\.conv_layers\.0\.2\.(weight|bias)$", r"feature_extractor.conv_layers.0.layer_norm.\1"),
    (r"^feature_extractor\.conv_layers\.(\d+)\.0\.(weight|bias)$", r"feature_extractor.conv_layers.\1.conv.\2"),
    # extractor_mode "layer_norm" (large models): Sequential(TransposeLast, LayerNorm, TransposeLast) at index 2
    (r"^feature_extractor\.conv_layers\.(\d+)\.2\.1\.(weight|bias)$", r"feature_extractor.conv_layers.\1.layer_norm.\2"),
    (r"^post_extract_proj\.(weight|bias)$", r"encoder.feature_projection.projection.\1"),
    (r"^layer_norm\.(weight|bias)$", r"encoder.feature_projection.layer_norm.\1"),
    (r"^encoder\.pos_conv\.0\.(bias|weight_g|weight_v)$", r"encoder.transformer.pos_conv_embed.conv.\1"),
    (r"^encoder\.layer_norm\.(weight|bias)$", r"encoder.transformer.layer_norm.\1"),
    (r"^encoder\.layers\.(\d+)\.self_attn\.((?:q|k|v|out)_proj)\.(weight|bias)$", r"encoder.transformer.layers.\1.attention.\2.\3"),
    (r"^encoder\.layers\.(\d+)\.self_attn_layer_norm\.(weight|bias)$", r"encoder.transformer.layers.\1.layer_norm.\2"),
    (r"^encoder\.layers\.(\d+)\.fc1\.(weight|bias)$", r"encoder.transformer.layers.\1.feed_forward.intermediate_dense.\2"),
    (r"^encoder\.layers\.(\d+)\.fc2\.(weight|bias)$", r"encoder.transformer.layers.\1.feed_forward.output_dense.\2"),
    (r"^encoder\.layers\.(\d+)\.final_layer_norm\.(weight|bias)$", r"encoder.transformer.layers.\1.final_layer_norm.\2"),
    (r"^proj\.(weight|bias)$", r"encoder.readout.\1"),      # fine-tuned (CTC) checkpoints only
]
_RULES = [(re.compile(p), r) for p, r in _RULES]
_FUSED = re.compile(r"^encoder\.layers\.(\d+)\.self_attn\.in_proj_(weight|bias)$")   # fairseq < 0.9: q, k, v stacked on dim 0


class ConversionError(ValueError):
    pass


def fairseq_to_torchaudio(state_dict, strict=True):
    """fairseq `Wav2Vec2Model` (or fine-tuned `w2v_encoder.w2v_model.`-prefixed) state dict -> torchaudio 0.9.1 names.
    Returns (converted, dropped_keys).  Tensors are passed through untouched (same storage): the import is a renaming,
    plus the split of a fused `in_proj_weight` into q / k / v thirds for checkpoints written by older fairseq."""
    out, dropped = {}, []
    for key, value in state_dict.items():
        k = key
        for prefix in ("w2v_encoder.w2v_model.", "w2v_model.", "module."):
            if k.startswith(prefix):
                k = k[len(prefix):]
        if k.startswith("w2v_encoder.proj."):          # the CTC head of a fine-tuned checkpoint
            k = k[len("w2v_encoder."):]
        if _DROPPED.match(k):
            dropped.append(key)
            continue
        m = _FUSED.match(k)
        if m:
            if value.shape[0] % 3:
                raise ConversionError(f"{key}: fused projection of {tuple(value.shape)} does not split into q, k, v")
            d = value.shape[0] // 3
            for i, name in enumerate(("q_proj", "k_proj", "v_proj")):
                out[f"encoder.transformer.layers.{m.group(1)}.attention.{name}.{m.group(2)}"] = value[i * d:(i + 1) * d]
            continue
        for pat, repl in _RULES:
            if pat.match(k):
                out[pat.sub(repl, k)] = value
                break
        else:
            if strict:
                raise ConversionError(f"unexpected key in the fairseq checkpoint: {key}")
            dropped.append(key)
    return out, dropped


def _get(obj, name, default=None):
    """Attribute of an argparse.Namespace / omegaconf node that the restricted unpickler turned into a stub or dict."""
    state = getattr(obj, "state", None)          # peppa_amd.checkpoint._Stub keeps the pickled __dict__ here
    if isinstance(state, dict) and name in state:
        return state[name]
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def check_base_architecture(ckpt):
    """The reference instantiates wav2vec2-BASE geometry through import_fairseq_model's config parsing; this build has
    that geometry fixed (peppa_amd.audio).  Refuse a checkpoint whose recorded arguments say otherwise."""
    args = ckpt.get("args")
    if args is None and ckpt.get("cfg") is not None:
        args = _get(ckpt["cfg"], "model")
    if args is None:
        return
    want = {"encoder_layers": 12, "encoder_embed_dim": 768, "encoder_ffn_embed_dim": 3072, "encoder_attention_heads": 12,
            "conv_pos": 128, "conv_pos_groups": 16, "extractor_mode": "default", "conv_bias": False,
            "layer_norm_first": False}
    for name, value in want.items():
        got = _get(args, name)
        if got is not None and got != value:
            raise ConversionError(f"fairseq checkpoint has {name}={got!r}; the wav2vec2-base tower needs {value!r}")


def load_fairseq_wav2vec2(path, model):
    """`import_fairseq_model(load_model_ensemble_and_task([path])[0][0], num_out)` for peppa_amd.audio.Wav2Vec2Model:
    reads `path`, renames, checks every shape against `model`, loads.  The readout keeps `model`'s fresh initialisation
    unless the checkpoint is a fine-tuned one with a head of the same width.  Returns the dropped keys."""
    from .checkpoint import load_checkpoint
    ckpt = load_checkpoint(path)
    if not isinstance(ckpt, dict) or "model" not in ckpt:
        raise ConversionError(f"{path}: not a fairseq checkpoint (no 'model' entry)")
    check_base_architecture(ckpt)
    converted, dropped = fairseq_to_torchaudio(ckpt["model"])
    target = model.state_dict()
    head = [k for k in converted if k.startswith("encoder.readout.")]
    if head and any(converted[k].shape != target[k].shape for k in head):
        for k in head:          # a CTC head of another vocabulary: torchaudio would build that width; pig asks for 28
            dropped.append(k)
            del converted[k]
    for k, v in converted.items():
        if k not in target:
            raise ConversionError(f"{path}: converted key {k} has no counterpart in the wav2vec2-base tower")
        if tuple(v.shape) != tuple(target[k].shape):
            raise ConversionError(f"{path}: {k} is {tuple(v.shape)}, the tower has {tuple(target[k].shape)}")
    missing = [k for k in target if k not in converted and not k.startswith("encoder.readout.")]
    if missing:
        raise ConversionError(f"{path}: the checkpoint lacks {missing[:5]}{' ...' if len(missing) > 5 else ''}")
    with torch.no_grad():
        model.load_state_dict({k: v.float() for k, v in converted.items()}, strict=False)
    return dropped


def video_state_dict(sd, trunk):
    """A torchvision video / image ResNet file (state dict, optionally wrapped or DataParallel-prefixed) checked against
    `trunk`'s own names and shapes; `num_batches_tracked` entries the file may lack are taken from `trunk`."""
    if isinstance(sd, dict) and "state_dict" in sd and not torch.is_tensor(sd["state_dict"]):
        sd = sd["state_dict"]
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    target = trunk.state_dict()
    for k, v in sd.items():
        if k not in target:
            raise ConversionError(f"video weights: unexpected key {k}")
        if tuple(v.shape) != tuple(target[k].shape):
            raise ConversionError(f"video weights: {k} is {tuple(v.shape)}, the trunk has {tuple(target[k].shape)}")
    missing = [k for k in target if k not in sd and not k.endswith("num_batches_tracked")]
    if missing:
        raise ConversionError(f"video weights: missing {missing[:5]}{' ...' if len(missing) > 5 else ''}")
    return {**{k: v for k, v in target.items() if k.endswith("num_batches_tracked")}, **sd}
