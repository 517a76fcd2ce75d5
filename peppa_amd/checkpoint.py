"""Lightning-compatible checkpoint read/write for `PeppaPig`.

The reference trains under pytorch-lightning 1.4.9 (run.py:32-62) and scores published checkpoints through
`pig.evaluation.load_best_model` (pig/evaluation.py:42-53), which reads

    cp['callbacks'][pl.callbacks.model_checkpoint.ModelCheckpoint] -> {monitor, best_model_score, best_model_path, ...}

and then `PeppaPig.load_from_checkpoint(path, hparams_file=...)`.  A 1.4.9 checkpoint is a pickled dict with
`state_dict` (torchvision / torchaudio parameter names, which the HIP modules keep), `hyper_parameters` (the config
dict: `save_hyperparameters(config)`, pig/models.py:228), `hparams_name` = "config", `epoch`, `global_step`,
`optimizer_states`, `lr_schedulers`, `callbacks` (keyed by the callback CLASS) and `pytorch-lightning_version`.

Lightning is not installed here, so
 * reading goes through a restricted unpickler with an EXACT (module, name) allowlist (tensor / storage rebuilders,
   dtypes, OrderedDict, numpy array reconstruction, a short list of harmless builtins); every other global --
   Lightning's callback class used as a dict key, its enums, and equally `torch.utils.collect_env.run` or
   `numpy.testing.*` in a hostile file -- becomes an inert stub that remembers its dotted name and only stores the
   arguments it is "called" with.  No callable from the file is executed;
 * writing uses the real `ModelCheckpoint` class as the key when Lightning is importable and the dotted name otherwise;
   `callback_states` accepts both, so files written here load there and vice versa.
"""
import copy
import glob
import logging
import os
import pickle
import types

import torch
import yaml

MODEL_CHECKPOINT = "pytorch_lightning.callbacks.model_checkpoint.ModelCheckpoint"
LIGHTNING_VERSION = "1.4.9"          # requirements.txt:59 of the reference

def _allowed_globals():
    """Exact (module, name) allowlist, the way torch._weights_only_unpickler does it: what a tensor / optimizer-state /
    config pickle needs to rebuild itself, and nothing that takes a callable or a command.  A root-module allowlist
    ("anything under torch.*") is NOT safe: torch.utils.collect_env.run, torch.hub.*, numpy.testing.* are one REDUCE
    away from a shell."""
    import collections
    import copyreg
    import _codecs
    import numpy
    import torch._utils as tu
    ok = {
        ("collections", "OrderedDict"): collections.OrderedDict,
        ("_codecs", "encode"): _codecs.encode,
        ("copyreg", "_reconstructor"): copyreg._reconstructor,     # object.__new__(cls) for classes resolved HERE
        ("copyreg", "__newobj__"): copyreg.__newobj__,
        ("numpy", "dtype"): numpy.dtype,
        ("numpy", "ndarray"): numpy.ndarray,
        ("torch", "Size"): torch.Size,
        ("torch", "device"): torch.device,
        ("torch", "Tensor"): torch.Tensor,
        ("torch.nn.parameter", "Parameter"): torch.nn.Parameter,
        ("torch.serialization", "_get_layout"): torch.serialization._get_layout,
    }
    for name in ("_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_parameter_with_state",
                 "_rebuild_device_tensor_from_numpy"):
        if hasattr(tu, name):
            ok[("torch._utils", name)] = getattr(tu, name)
    try:
        from numpy._core import multiarray as ma
    except ImportError:  # numpy < 2
        from numpy.core import multiarray as ma
    for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
        ok[(mod, "_reconstruct")] = ma._reconstruct
        ok[(mod, "scalar")] = ma.scalar
    for name, obj in vars(torch).items():
        if isinstance(obj, torch.dtype):
            ok[("torch", name)] = obj
        elif name.endswith("Storage") and isinstance(obj, type):
            ok[("torch", name)] = obj
    for name in ("TypedStorage", "UntypedStorage"):
        ok[("torch.storage", name)] = getattr(torch.storage, name)
    return ok


_ALLOWED = None
_LIGHTNING_OK = {("pytorch_lightning.callbacks.model_checkpoint", "ModelCheckpoint"),
                 ("pytorch_lightning.utilities.parsing", "AttributeDict")}
_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
_STUBS = {}


class _Stub(dict):
    """Stand-in for a class that cannot be imported (state is kept, nothing runs).  A dict, so that pickled
    dict subclasses (Lightning's AttributeDict) keep their items."""
    _dotted = "?"

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.args, self.kwargs = args, kwargs

    def __hash__(self):
        return id(self)

    def __setstate__(self, state):
        self.state = state

    def __reduce_ex__(self, protocol):     # never re-pickle a stub as if it were the real object
        raise pickle.PicklingError(f"cannot pickle the stand-in for {self._dotted}")


def _stub(module, name):
    dotted = f"{module}.{name}"
    if dotted not in _STUBS:
        _STUBS[dotted] = type(name, (_Stub,), {"_dotted": dotted, "__module__": module})
    return _STUBS[dotted]


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        global _ALLOWED
        if _ALLOWED is None:
            _ALLOWED = _allowed_globals()
        hit = _ALLOWED.get((module, name))
        if hit is not None:
            return hit
        if module in ("builtins", "__builtin__") and name in _SAFE_BUILTINS:   # (protocol 2 writes the py2 module name)
            return super().find_class(module, name)
        if (module, name) in _LIGHTNING_OK:
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                pass
        return _stub(module, name)


_pickle_module = types.SimpleNamespace(
    __name__="peppa_amd.checkpoint", Unpickler=_Unpickler, Pickler=pickle.Pickler,
    load=lambda f, **kw: _Unpickler(f, **kw).load(), loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps)


def dotted_name(key):
    """'pkg.mod.Class' for a class object (real or stub); strings pass through."""
    if isinstance(key, str):
        return key
    if isinstance(key, type):
        return getattr(key, "_dotted", f"{key.__module__}.{key.__qualname__}")
    return f"{type(key).__module__}.{type(key).__qualname__}"


def load_checkpoint(path, map_location="cpu"):
    """torch.load for Lightning checkpoints that works without Lightning (see the module docstring)."""
    return torch.load(path, map_location=map_location, pickle_module=_pickle_module, weights_only=False)


def callback_states(checkpoint):
    """All `ModelCheckpoint` states of a checkpoint (1.4.9 keys them by class, so there is at most one: the second
    callback of run.py overwrites the first)."""
    return [state for key, state in checkpoint.get("callbacks", {}).items()
            if dotted_name(key).split("{")[0].endswith("ModelCheckpoint")]


def _callback_key():
    try:
        from pytorch_lightning.callbacks.model_checkpoint import ModelCheckpoint as Real
        return Real
    except Exception:
        return MODEL_CHECKPOINT


def _plain(obj):
    """Config as plain containers (an AttributeDict from Lightning pickles as its own class)."""
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    return obj


def optimizer_state(optimizer, scaler=None):
    """BertAdam keeps the reference's per-parameter `step` / `next_m` / `next_v` state (pig/optimization.py:120-128),
    so the plain `state_dict()` is already what Lightning stores under `optimizer_states`.  `scaler`: the fp16 run's
    GradScaler -- flushed first, so a step it skipped on the device does not stay counted in `step`."""
    if scaler is not None:
        scaler.flush()
    sd = optimizer.state_dict()
    # state_dict() hands out the optimizer's LIVE per-parameter dicts: build new ones, never move those in place
    state = {key: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
             for key, st in sd["state"].items()}
    return {"state": state, "param_groups": sd["param_groups"]}


def save_checkpoint(path, net, optimizer=None, epoch=0, global_step=0, callback_state=None, scaler=None):
    """Write a Lightning-1.4.9-shaped checkpoint.  `callback_state`: the ModelCheckpoint block
    ({monitor, best_model_score, best_model_path, current_score, dirpath}).  `scaler`: the fp16 run's GradScaler; its state
    goes where Lightning 1.4's native-AMP plugin puts it ("native_amp_scaling_state")."""
    amp_state = None if scaler is None else scaler.state_dict()    # first: settles the step counters of a skipped step
    cp = {
        "epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": LIGHTNING_VERSION,
        "state_dict": {k: v.detach().cpu() for k, v in net.state_dict().items()},
        "callbacks": {} if callback_state is None else {_callback_key(): dict(callback_state)},
        "optimizer_states": [] if optimizer is None else [optimizer_state(optimizer, scaler)],
        "lr_schedulers": [],
        "hparams_name": "config",
        "hyper_parameters": _plain({k: v for k, v in net.config.items()}),
    }
    if amp_state is not None:
        cp["native_amp_scaling_state"] = amp_state
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = f"{path}.tmp"
    torch.save(cp, tmp)
    os.replace(tmp, path)                 # a killed run never leaves a truncated .ckpt behind
    return path


def config_from(checkpoint, hparams_file=None):
    """The constructor argument: `hparams_file` (Lightning's hparams.yaml, the flattened config) wins over the
    checkpoint's own `hyper_parameters`, as in LightningModule.load_from_checkpoint."""
    if hparams_file is not None:
        with open(hparams_file) as f:
            config = yaml.safe_load(f)
    else:
        config = checkpoint.get("hyper_parameters")
        if config is None:
            raise KeyError("checkpoint has no 'hyper_parameters'; pass hparams_file=")
        config = _plain(getattr(config, "state", config) if isinstance(config, _Stub) else config)
    if "config" in config and "video" not in config:      # saved as save_hyperparameters() of the kwarg
        config = config["config"]
    return dict(config)


def load_model(cls, checkpoint_path, map_location=None, hparams_file=None, strict=True):
    """`PeppaPig.load_from_checkpoint` (pig/evaluation.py:52): rebuild from the stored config, then load the weights.
    The stored config may say `pretrained: true` (the run started from the fairseq / Kinetics weights); the
    state_dict replaces every weight, so the architecture is built from random init, the video encoder keeps the
    pretrained input normalisation, and the config is kept as stored."""
    cp = load_checkpoint(checkpoint_path, map_location="cpu")
    config = config_from(cp, hparams_file)
    build = copy.deepcopy(config)
    build.setdefault("audio", {})["pretrained"] = False
    video_pretrained = bool(build.setdefault("video", {}).get("pretrained", False))
    build["video"]["pretrained"] = False
    net = cls(build)
    if video_pretrained and hasattr(net.video_encoder, "mark_pretrained"):
        net.video_encoder.mark_pretrained()       # Kinetics / ImageNet input normalisation, as trained
    net.config = config
    missing, unexpected = net.load_state_dict(cp["state_dict"], strict=False)
    unexpected = [k for k in unexpected if not k.endswith("num_batches_tracked")]
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if strict and (missing or unexpected):
        raise RuntimeError(f"state_dict mismatch: missing {missing[:8]} unexpected {unexpected[:8]}")
    if map_location is not None:
        net = net.to(map_location)
    return net


def load_best_model(dirname, higher_better=True, cls=None):
    """pig/evaluation.py:42-53: scan `{dirname}/checkpoints/*.ckpt`, pick the best `best_model_score`, load that file
    with `{dirname}/hparams.yaml`.  The reference strips an absolute '/peppa/' prefix from the recorded path
    (:51); here a recorded path that no longer exists is resolved against `{dirname}/checkpoints/` by file name."""
    if cls is None:
        from .models import PeppaPig as cls
    info = []
    for path in sorted(glob.glob(f"{dirname}/checkpoints/*.ckpt")):
        for item in callback_states(load_checkpoint(path)):
            if item.get("best_model_score") is not None:
                info.append(item)
    if not info:
        raise FileNotFoundError(f"no checkpoint with a best_model_score under {dirname}/checkpoints")
    best = sorted(info, key=lambda x: float(x["best_model_score"]), reverse=higher_better)[0]
    logging.info(f"Best {best['monitor']}: {best['best_model_score']} at {best['best_model_path']}")
    recorded = best["best_model_path"]
    local = recorded.split("/peppa/")[1] if "/peppa/" in recorded else recorded
    if not os.path.exists(local):
        local = os.path.join(dirname, "checkpoints", os.path.basename(recorded))
    hparams = os.path.join(dirname, "hparams.yaml")
    net = load_model(cls, local, hparams_file=hparams if os.path.exists(hparams) else None)
    return net, recorded


class ModelCheckpoint:
    """The part of Lightning's callback run.py configures (run.py:32-55): keep the best epoch by `monitor`
    (`save_top_k=1`), and `last.ckpt` when `save_last`; file names follow `filename` with
    `auto_insert_metric_name` ("epoch=3-valnarr_triplet=0.71.ckpt")."""

    def __init__(self, monitor, mode="max", save_last=True, save_top_k=1, dirpath=None,
                 filename=None, auto_insert_metric_name=True, **ignored):
        if mode not in ("max", "min"):
            raise ValueError(f"mode must be 'max' or 'min', got {mode}")
        self.monitor, self.mode, self.save_last, self.save_top_k = monitor, mode, save_last, save_top_k
        self.dirpath, self.filename, self.auto_insert = dirpath, filename or "{epoch}", auto_insert_metric_name
        self.best_model_score, self.best_model_path, self.current_score = None, "", None

    def format_name(self, epoch, metrics):
        values = dict(metrics, epoch=epoch)
        name = self.filename
        if self.auto_insert:
            for key in values:
                name = name.replace("{" + key, key + "={" + key)
        return name.format(**{k: (float(v) if k != "epoch" else int(v)) for k, v in values.items()}) + ".ckpt"

    def state(self):
        return {"monitor": self.monitor, "best_model_score": self.best_model_score,
                "best_model_path": self.best_model_path, "current_score": self.current_score,
                "dirpath": self.dirpath}

    def load_state(self, state):
        self.best_model_score, self.best_model_path = state.get("best_model_score"), state.get("best_model_path", "")
        self.current_score = state.get("current_score")

    def better(self, score):
        if self.best_model_score is None:
            return True
        return score > self.best_model_score if self.mode == "max" else score < self.best_model_score

    def on_validation_end(self, net, optimizer, epoch, global_step, metrics, scaler=None):
        """Called by the trainer after `validation_epoch_end`; returns the paths written."""
        if self.dirpath is None or self.monitor not in metrics:
            return []
        score = torch.as_tensor(metrics[self.monitor]).detach().float().cpu()
        self.current_score = score
        written = []
        if self.save_top_k != 0 and self.better(score):
            path = os.path.join(self.dirpath, self.format_name(epoch, metrics))
            old, self.best_model_score, self.best_model_path = self.best_model_path, score, path
            written.append(save_checkpoint(path, net, optimizer, epoch, global_step, self.state(), scaler))
            if self.save_top_k == 1 and old and old != path and os.path.exists(old):
                os.remove(old)
        if self.save_last:
            written.append(save_checkpoint(os.path.join(self.dirpath, "last.ckpt"), net, optimizer, epoch,
                                           global_step, self.state(), scaler))
        return written
