"""Lightning-format checkpoints: write / read / best-model selection, without Lightning installed.
Reference behaviour: run.py:32-55 (ModelCheckpoint callbacks), pig/evaluation.py:42-53 (load_best_model)."""
import copy
import os
import sys
import types

import pytest
import torch
import yaml
from torch import nn

from peppa_amd import checkpoint as C


class TinyPig(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.video_encoder = nn.Linear(config["video"]["width"], 4)
        self.audio_encoder = nn.BatchNorm1d(4)


CONFIG = {"video": {"width": 6, "pretrained": True}, "audio": {"pretrained": True, "path": "x.pt"}, "margin": 0.2}


def _tiny(seed):
    torch.manual_seed(seed)
    return TinyPig(copy.deepcopy(CONFIG))


def test_roundtrip_keeps_lightning_layout(tmp_path):
    net = _tiny(0)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    path = C.save_checkpoint(str(tmp_path / "a.ckpt"), net, opt, epoch=3, global_step=77,
                             callback_state={"monitor": "valnarr_triplet", "best_model_score": torch.tensor(0.71),
                                             "best_model_path": str(tmp_path / "a.ckpt"), "current_score": None,
                                             "dirpath": str(tmp_path)})
    cp = C.load_checkpoint(path)
    for key in ("epoch", "global_step", "pytorch-lightning_version", "state_dict", "callbacks", "optimizer_states",
                "lr_schedulers", "hparams_name", "hyper_parameters"):
        assert key in cp
    assert cp["epoch"] == 3 and cp["global_step"] == 77 and cp["hparams_name"] == "config"
    assert cp["hyper_parameters"] == CONFIG
    (state,) = C.callback_states(cp)
    assert state["monitor"] == "valnarr_triplet" and float(state["best_model_score"]) == pytest.approx(0.71)
    for k, v in net.state_dict().items():
        assert torch.equal(cp["state_dict"][k], v)
    again = C.load_model(TinyPig, path)
    assert again.config == CONFIG          # the stored config, including pretrained: true, survives
    for k, v in net.state_dict().items():
        assert torch.equal(again.state_dict()[k], v)


def test_strict_reports_mismatch(tmp_path):
    net = _tiny(0)
    path = C.save_checkpoint(str(tmp_path / "a.ckpt"), net)
    cp = C.load_checkpoint(path)
    del cp["state_dict"]["video_encoder.bias"]
    torch.save(cp, path)
    with pytest.raises(RuntimeError, match="video_encoder.bias"):
        C.load_model(TinyPig, path)
    C.load_model(TinyPig, path, strict=False)


def test_hparams_file_wins(tmp_path):
    net = _tiny(0)
    path = C.save_checkpoint(str(tmp_path / "a.ckpt"), net)
    other = copy.deepcopy(CONFIG)
    other["margin"] = 0.5
    hp = tmp_path / "hparams.yaml"
    hp.write_text(yaml.safe_dump(other))
    assert C.load_model(TinyPig, path, hparams_file=str(hp)).config["margin"] == 0.5


def test_reads_a_checkpoint_written_by_lightning_without_lightning(tmp_path):
    """Lightning 1.4.9 keys `callbacks` by the ModelCheckpoint CLASS and stores hyper-parameters as an AttributeDict;
    both live in modules that do not exist here.  Write such a file with stand-in modules, drop them, read it back."""
    names = ["pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.model_checkpoint",
             "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing"]
    if "pytorch_lightning" in sys.modules:
        pytest.skip("Lightning is installed: the real classes are used")
    mods = {n: types.ModuleType(n) for n in names}
    MC = type("ModelCheckpoint", (), {"__module__": names[2]})
    AD = type("AttributeDict", (dict,), {"__module__": names[4]})
    mods[names[2]].ModelCheckpoint, mods[names[4]].AttributeDict = MC, AD
    sys.modules.update(mods)
    try:
        net = _tiny(1)
        cp = {"epoch": 1, "global_step": 5, "pytorch-lightning_version": "1.4.9", "state_dict": net.state_dict(),
              "callbacks": {MC: {"monitor": "valnarr_rec_fixed", "best_model_score": torch.tensor(0.25),
                                 "best_model_path": "/home/u/peppa/lightning_logs/version_3/checkpoints/e.ckpt",
                                 "current_score": torch.tensor(0.25), "dirpath": "/x"}},
              "hparams_name": "config", "hyper_parameters": AD(copy.deepcopy(CONFIG))}
        path = str(tmp_path / "pl.ckpt")
        torch.save(cp, path)
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(Exception):
        torch.load(path, weights_only=False)             # the plain loader cannot resolve the classes
    cp = C.load_checkpoint(path)
    (state,) = C.callback_states(cp)
    assert state["monitor"] == "valnarr_rec_fixed"
    (key,) = cp["callbacks"].keys()
    assert C.dotted_name(key) == C.MODEL_CHECKPOINT
    assert C.config_from(cp) == CONFIG and type(C.config_from(cp)) is dict
    again = C.load_model(TinyPig, path)
    assert torch.equal(again.video_encoder.weight, net.video_encoder.weight)


def test_unpickler_does_not_run_foreign_globals(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    path = str(tmp_path / "evil.ckpt")
    torch.save({"x": Evil()}, path)
    cp = C.load_checkpoint(path)          # os.system became an inert stub: constructed, never called
    assert isinstance(cp["x"], C._Stub) and C.dotted_name(type(cp["x"])) in ("posix.system", "os.system", "nt.system")


@pytest.mark.parametrize("payload", ["torch_collect_env", "numpy_testing", "torch_load_bytes", "torch_hub"])
def test_unpickler_does_not_run_torch_or_numpy_rooted_callables(tmp_path, payload):
    """a root-module allowlist let `torch.utils.collect_env.run("...")` through.  The allowlist
    is exact now: a torch.* / numpy.* callable that is not a tensor rebuilder becomes a stub and never runs."""
    marker = tmp_path / "marker"
    cmd = f"touch {marker}"

    class Evil:
        def __reduce__(self):
            if payload == "torch_collect_env":
                import torch.utils.collect_env as ce
                return (ce.run, (cmd,))
            if payload == "numpy_testing":
                import numpy.testing._private.utils as u
                return (u.runstring, (f"import os; os.system({cmd!r})", {}))
            if payload == "torch_load_bytes":
                return (torch.storage._load_from_bytes, (b"not a pickle",))
            return (torch.hub.load, ("/nonexistent", "x"))
    path = str(tmp_path / "evil.ckpt")
    torch.save({"state_dict": {"w": torch.ones(2)}, "x": Evil()}, path)
    cp = C.load_checkpoint(path)
    assert not marker.exists()
    assert isinstance(cp["x"], C._Stub)
    assert torch.equal(cp["state_dict"]["w"], torch.ones(2))   # ordinary tensors still load


def test_unpickler_allowlist_covers_what_checkpoints_contain(tmp_path):
    import collections
    import numpy as np
    cp = {"state_dict": collections.OrderedDict(a=torch.arange(6, dtype=torch.bfloat16).view(2, 3), b=torch.nn.Parameter(torch.ones(2)),
                                                 c=torch.tensor(3, dtype=torch.int64), d=torch.zeros(2, dtype=torch.float16)),
          "np": np.arange(4, dtype=np.float32), "npscalar": np.float64(2.5), "size": torch.Size([2, 3]), "dtype": torch.float32,
          "dev": torch.device("cpu"), "set": {1, 2}, "nested": [(1, 2.0, "s", b"b", None, True)]}
    path = str(tmp_path / "plain.ckpt")
    torch.save(cp, path)
    got = C.load_checkpoint(path)
    assert torch.equal(got["state_dict"]["a"], cp["state_dict"]["a"]) and got["state_dict"]["a"].dtype == torch.bfloat16
    assert isinstance(got["state_dict"]["b"], torch.nn.Parameter)
    assert (got["np"] == cp["np"]).all() and got["npscalar"] == 2.5 and got["size"] == cp["size"]
    assert got["dtype"] is torch.float32 and got["dev"] == torch.device("cpu") and got["set"] == {1, 2}
    assert got["nested"] == cp["nested"]


def test_model_checkpoint_callback_and_load_best_model(tmp_path):
    root = tmp_path / "version_0"
    cb = C.ModelCheckpoint(monitor="valnarr_triplet", mode="max", save_last=True, save_top_k=1,
                           dirpath=str(root / "checkpoints"), filename="{epoch}-{valnarr_triplet:.2f}")
    (root).mkdir()
    (root / "hparams.yaml").write_text(yaml.safe_dump(CONFIG))
    nets = [_tiny(s) for s in range(3)]
    scores = [0.61, 0.74, 0.70]
    for epoch, (net, score) in enumerate(zip(nets, scores)):
        cb.on_validation_end(net, None, epoch, 10 * epoch, {"valnarr_triplet": torch.tensor(score), "val_loss": 1.0})
    files = sorted(os.listdir(root / "checkpoints"))
    assert files == ["epoch=1-valnarr_triplet=0.74.ckpt", "last.ckpt"]       # top-1 kept, older best removed
    best, recorded = C.load_best_model(str(root), cls=TinyPig)
    assert recorded.endswith("epoch=1-valnarr_triplet=0.74.ckpt")
    assert torch.equal(best.video_encoder.weight, nets[1].video_encoder.weight)
    last = C.load_checkpoint(str(root / "checkpoints" / "last.ckpt"))
    assert torch.equal(last["state_dict"]["video_encoder.weight"], nets[2].video_encoder.weight)
    assert float(C.callback_states(last)[0]["best_model_score"]) == pytest.approx(0.74)
    # lower-is-better monitors
    lo = C.ModelCheckpoint(monitor="val_loss", mode="min", save_last=False, dirpath=str(tmp_path / "lo"))
    for epoch, v in enumerate([0.9, 0.4, 0.6]):
        lo.on_validation_end(nets[epoch], None, epoch, epoch, {"val_loss": v})
    assert os.listdir(tmp_path / "lo") == ["epoch=1.ckpt"]
    with pytest.raises(ValueError):
        C.ModelCheckpoint(monitor="x", mode="best")


def test_moved_run_directory_is_resolved_by_file_name(tmp_path):
    root = tmp_path / "run"
    net = _tiny(4)
    elsewhere = "/data/someone/peppa/lightning_logs/version_9/checkpoints/epoch=7.ckpt"
    C.save_checkpoint(str(root / "checkpoints" / "epoch=7.ckpt"), net, callback_state={
        "monitor": "valnarr_triplet", "best_model_score": torch.tensor(0.8), "best_model_path": elsewhere,
        "current_score": None, "dirpath": os.path.dirname(elsewhere)})
    best, recorded = C.load_best_model(str(root), cls=TinyPig)
    assert recorded == elsewhere and torch.equal(best.video_encoder.weight, net.video_encoder.weight)
    with pytest.raises(FileNotFoundError):
        C.load_best_model(str(tmp_path / "empty"), cls=TinyPig)


def test_peppapig_state_dict_names_follow_torchvision_and_torchaudio():
    """A published checkpoint only loads if the parameter names are the reference's (pig/models.py:66-154 on top of
    torchvision r2plus1d_18 / torchaudio wav2vec2_base)."""
    import pig.models
    from pig.execution import default_config
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    keys = set(pig.models.PeppaPig(cfg).state_dict().keys())
    for name in ("video_encoder.video.stem.0.weight", "video_encoder.video.stem.1.running_var",
                 "video_encoder.video.layer1.0.conv1.0.0.weight", "video_encoder.video.layer1.0.conv1.0.3.weight",
                 "video_encoder.video.layer2.0.downsample.0.weight", "video_encoder.video.layer4.1.conv2.1.bias",
                 "video_encoder.project.weight",
                 "audio_encoder.audio.feature_extractor.conv_layers.0.conv.weight",
                 "audio_encoder.audio.feature_extractor.conv_layers.0.layer_norm.weight",
                 "audio_encoder.audio.encoder.feature_projection.projection.weight",
                 "audio_encoder.audio.encoder.transformer.pos_conv_embed.conv.weight_g",
                 "audio_encoder.audio.encoder.transformer.layers.11.attention.out_proj.bias",
                 "audio_encoder.audio.encoder.transformer.layers.0.feed_forward.intermediate_dense.weight",
                 "audio_encoder.audio.encoder.readout.weight", "audio_encoder.project.bias"):
        assert name in keys, name
    assert hasattr(pig.models.PeppaPig, "load_from_checkpoint")


def test_saving_leaves_the_live_optimizer_state_alone(tmp_path):
    net = _tiny(0)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    net.video_encoder(torch.ones(2, 6)).sum().backward()
    opt.step()
    before = {id(p): opt.state[p]["momentum_buffer"] for p in net.video_encoder.parameters()}
    C.save_checkpoint(str(tmp_path / "a.ckpt"), net, opt)
    for p in net.video_encoder.parameters():
        assert opt.state[p]["momentum_buffer"] is before[id(p)]
    cp = C.load_checkpoint(str(tmp_path / "a.ckpt"))
    assert torch.equal(cp["optimizer_states"][0]["state"][0]["momentum_buffer"],
                       opt.state[net.video_encoder.weight]["momentum_buffer"])


def test_loss_scaler_state_rides_where_lightning_puts_it(tmp_path):
    """An fp16 run's dynamic loss scale is saved under Lightning 1.4's "native_amp_scaling_state" and restored with its
    growth tracker, also into a scaler that has not touched a device yet."""
    from peppa_amd.amp import GradScaler
    net = _tiny(0)
    sc = GradScaler(init_scale=4096.0, growth_interval=2000)
    sc.load_state_dict({"scale": 512.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                        "_growth_tracker": 1234})
    path = C.save_checkpoint(str(tmp_path / "s.ckpt"), net, scaler=sc)
    cp = C.load_checkpoint(path)
    assert cp["native_amp_scaling_state"] == {"scale": 512.0, "growth_factor": 2.0, "backoff_factor": 0.5,
                                              "growth_interval": 2000, "_growth_tracker": 1234}
    again = GradScaler()
    again.load_state_dict(cp["native_amp_scaling_state"])
    assert again.get_scale() == 512.0 and again._init_tracker == 1234
    assert "native_amp_scaling_state" not in C.load_checkpoint(C.save_checkpoint(str(tmp_path / "n.ckpt"), net))
