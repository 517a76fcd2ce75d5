"""`pig.models` on MI355X: same classes, constructor arguments, attribute names and errors as
the reference (pig/models.py:30-342); every forward runs hand-written HIP kernels through
libpeppa_hip.so.  CPU tensors are rejected (PeppaHipError) -- there is no fallback path.
"""
import logging
import torch
from torch import nn

from . import hip as H
from . import layers as L
from . import audio as A
from . import video as V
from .hip import f32, act16
from .loss import TripletLoss
from . import metrics as _metrics
from . import optimization as opt
from .data import ClipBatch
from .dist import gather_embeddings, grad_dict
from .triplet import TripletBatch as EmbeddingTripletBatch, score_triplets
from .targeted_triplets import TripletBatch
from .transforms import SwapCT  # noqa: F401  (API surface)

try:  # the reference subclasses LightningModule; Lightning is optional here
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - Lightning absent in this image
    pl = None
    _Base = nn.Module


TargetedTripletBatch = TripletBatch     # the class `forward` dispatches on (pig/models.py:20,238)


class AttnPoolFn(torch.autograd.Function):
    """softmax-over-time attention pooling (+ optional Linear projection, + optional F.normalize)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, Wp, bp, normalize):
        if not x.is_cuda:
            raise H.PeppaHipError("peppa_amd pooling needs a CUDA/HIP tensor (no CPU fallback)")
        x = x.contiguous().float()
        B, T, Fd = x.shape
        Hd = W1.shape[0]
        E = Wp.shape[0] if Wp is not None else Fd
        e = lambda *s: torch.empty(*s, dtype=f32, device=x.device)
        hid, alpha, pooled, pre, out = e(B, T, Hd), e(B, T, Fd), e(B, Fd), e(B, E), e(B, E)
        H.attnpool_fwd(x, B, T, Fd, Hd, E, W1, b1, W2, b2, Wp, bp, hid, alpha, pooled, pre, out, normalize=normalize)
        ctx.save_for_backward(x, W1, W2, Wp, hid, alpha, pooled, pre, out)
        ctx.dims, ctx.normalize = (B, T, Fd, Hd, E), normalize
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W1, W2, Wp, hid, alpha, pooled, pre, out = ctx.saved_tensors
        B, T, Fd, Hd, E = ctx.dims
        e = lambda *s: torch.empty(*s, dtype=f32, device=x.device)
        dx, dW1, db1, dW2, db2 = e(B, T, Fd), e(Hd, Fd), e(Hd), e(Fd, Hd), e(Fd)
        dWp, dbp = (e(E, Fd), e(E)) if Wp is not None else (None, None)
        ws = e(H.attnpool_ws_floats(B, T, Fd, Hd, E))
        H.attnpool_bwd(dout.contiguous().float(), x, B, T, Fd, Hd, E, W1, W2, Wp, hid, alpha, pooled, pre, out, dx,
                       dW1, db1, dW2, db2, dWp, dbp, ws, normalize=ctx.normalize)
        return dx, dW1, db1, dW2, db2, dWp, dbp, None


class Attention(nn.Module):
    def __init__(self, in_size, hidden_size):
        super().__init__()
        self.hidden = nn.Linear(in_size, hidden_size)
        self.out = nn.Linear(hidden_size, in_size)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, input):
        return AttnPoolFn.apply(input, self.hidden.weight, self.hidden.bias, self.out.weight, self.out.bias, None,
                                None, False)

    def pooled_projected(self, input, project):
        """Fused pooling + Linear + F.normalize(p=2, dim=1) (the tail of both encoders)."""
        return AttnPoolFn.apply(input, self.hidden.weight, self.hidden.bias, self.out.weight, self.out.bias,
                                project.weight, project.bias, True)


class _AvgPoolTFFn(torch.autograd.Function):
    """pig/models.py:45-51: nn.AdaptiveAvgPool2d((size, 1)) on the 3-D (B, T, F) tensor, as torch evaluates it."""

    @staticmethod
    def forward(ctx, x, size):
        if not x.is_cuda:
            raise H.PeppaHipError("peppa_amd pooling needs a CUDA/HIP tensor (no CPU fallback)")
        x = x.contiguous().float()
        B, T, Fd = x.shape
        out = torch.empty(B, size, dtype=f32, device=x.device)
        H.avgpool_tf_fwd(x, out, B, T, Fd, size)
        ctx.dims = (B, T, Fd, size)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, Fd, size = ctx.dims
        dx = torch.empty(B, T, Fd, dtype=f32, device=dout.device)
        H.avgpool_tf_bwd(dout.contiguous().float(), dx, B, T, Fd, size)
        return dx, None


class _LastStepFn(torch.autograd.Function):
    """x[:, -1, :] (pig/models.py:54-61) as a strided copy; backward scatters into zeros."""

    @staticmethod
    def forward(ctx, x):
        if not x.is_cuda:
            raise H.PeppaHipError("peppa_amd pooling needs a CUDA/HIP tensor (no CPU fallback)")
        x = x.contiguous().float()
        B, T, Fd = x.shape
        out = torch.empty(B, Fd, dtype=f32, device=x.device)
        H.copy_2d_f32(x.view(B, T * Fd)[:, (T - 1) * Fd:], T * Fd, out, Fd, B, Fd)
        ctx.dims = (B, T, Fd)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, Fd = ctx.dims
        dx = torch.empty(B, T * Fd, dtype=f32, device=dout.device)
        H.fill_f32(dx, 0.0)
        H.copy_2d_f32(dout.contiguous().float(), Fd, dx[:, (T - 1) * Fd:], T * Fd, B, Fd)
        return dx.view(B, T, Fd)


def _project_normalize(pooled, project):
    """Tail shared by every encoder head: `project` (nn.Linear or nn.Identity), then F.normalize(p=2, dim=1).
    Runs through the attention-pooling entry with one time step (softmax over a single step is 1, so the pooling
    is the identity and its dummy attention weights receive zero gradient)."""
    B, Fd = pooled.shape
    z = lambda *shape: torch.zeros(*shape, dtype=f32, device=pooled.device)
    W, b = (project.weight, project.bias) if isinstance(project, nn.Linear) else (None, None)
    if W is not None and b is None:
        b = z(W.shape[0])
    return AttnPoolFn.apply(pooled.view(B, 1, Fd), z(8, Fd), z(8), z(Fd, 8), z(Fd), W, b, True)


def _time_mean(x):
    """Mean over dim 1 of (B, T, F): attention pooling with all-zero scores (softmax of zeros is exactly 1 / T)."""
    B, T, Fd = x.shape
    z = lambda *shape: torch.zeros(*shape, dtype=f32, device=x.device)
    return AttnPoolFn.apply(x, z(8, Fd), z(8), z(Fd, 8), z(Fd), None, None, False)


class AveragePool(nn.Module):
    def __init__(self, size=512):
        super().__init__()
        self.size = size
        self.pool = torch.nn.AdaptiveAvgPool2d((size, 1))   # (kept for state_dict / repr parity; has no parameters)

    def forward(self, x):
        return _AvgPoolTFFn.apply(x, self.size)


class LastStep(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x):
        return _LastStepFn.apply(x)


class Wav2VecEncoder(nn.Module):
    def __init__(self, path, pretrained=True, freeze_feature_extractor=False, freeze_encoder_layers=None,
                 pooling='average', project=True, full=False, weights=None):
        super().__init__()
        if pretrained:
            self.audio = _load_fairseq_checkpoint(path, weights)
        else:
            self.audio = A.wav2vec2_base(num_out=28)
        if freeze_feature_extractor:
            for param in self.audio.feature_extractor.parameters():
                param.requires_grad = False
        if freeze_encoder_layers is not None:
            for index in range(0, freeze_encoder_layers):
                for param in self.audio.encoder.transformer.layers[index].parameters():
                    param.requires_grad = False
        self.full = full
        self.n_features = 28 if self.full else 512
        if pooling == 'average':
            self.audiopool = AveragePool(size=self.n_features)
        elif pooling == 'attention':
            self.audiopool = Attention(self.n_features, 128)
        elif pooling == 'last':
            self.audiopool = LastStep()
        else:
            raise ValueError(f"Invalid pooling: {pooling}")
        self.project = nn.Linear(self.n_features, 512) if project else nn.Identity()

    def forward(self, x):
        wave = x.squeeze(dim=1)
        if self.full:
            features, _ = self.audio(wave)
        else:
            features, _ = self.audio.extract_features(wave)
        if isinstance(self.audiopool, Attention) and isinstance(self.project, nn.Linear):
            return self.audiopool.pooled_projected(features, self.project)    # one fused tail
        return _project_normalize(self.audiopool(features), self.project)


def _load_weights_file(path):
    """A state_dict file (tensors only) through the restricted unpickler of peppa_amd.checkpoint."""
    from .checkpoint import load_checkpoint
    sd = load_checkpoint(path)
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


def _load_fairseq_checkpoint(path, weights=None):
    """pig/models.py:70-72: `load_model_ensemble_and_task([path])` + torchaudio's `import_fairseq_model(model, num_out=28)`.
    Neither fairseq nor torchaudio exists here; what the two calls do to the numbers is a renaming of the checkpoint's
    state dict (pre-training heads dropped, the 28-way readout left freshly initialised), which peppa_amd.convert does
    on the file itself.  `weights` (yaml: `mi355x: {audio_weights: file}`) is the alternative: the same model already
    converted, a state_dict with torchaudio parameter names (what `import_fairseq_model(...).state_dict()` saves)."""
    import os
    audio = A.wav2vec2_base(num_out=28)
    if weights is not None:
        audio.load_state_dict(_load_weights_file(weights))
        return audio
    if path is not None and os.path.isfile(path):
        from .convert import load_fairseq_wav2vec2
        load_fairseq_wav2vec2(path, audio)
        return audio
    raise RuntimeError(f"audio.pretrained=true needs the fairseq checkpoint {path!r} (pig/models.py:70-72, README.md:13: a "
                       "download), which is not there. Put the file in place, use pretrained: false (run.py "
                       "--random_init), or give `mi355x: {audio_weights: <state_dict with torchaudio names>}`.")


def _pretrained_trunk(trunk, kind, weights):
    """`pretrained=True` downloads Kinetics / ImageNet weights in the reference (pig/models.py:123-127,164).  Offline
    that is impossible, and silently training from random init while the config (and every checkpoint's
    hyper_parameters) says `pretrained: true` is worse than failing: require the weights as a local state_dict with
    torchvision parameter names (yaml: `mi355x: {video_weights: file}`) -- the file torchvision itself would have
    downloaded loads as it is (peppa_amd.convert.video_state_dict checks names and shapes)."""
    if weights is None:
        raise RuntimeError(f"video.pretrained=true needs the {kind} weights torchvision would download; they cannot be "
                           "fetched offline. Use pretrained: false (run.py --random_init), or give "
                           "`mi355x: {video_weights: <state_dict with torchvision names>}`.")
    from .convert import video_state_dict
    trunk.load_state_dict(video_state_dict(_load_weights_file(weights), trunk))
    return trunk


def _video_trunk_launch(enc, x, save):
    """Issue the trunk forward + spatial mean; returns (out (B,T',512) fp32, tape, dims)."""
    net = enc.video
    if save and not net.training:
        raise RuntimeError("backward through eval-mode BatchNorm is not supported on the HIP path")
    with torch.no_grad():
        z, thw, tape = V.trunk_forward(net, x, enc.norm_kind, net.training, save)
        B = x.shape[0]
        Tn, HW = thw[0], thw[1] * thw[2]
        out = torch.empty(B, Tn, 512, dtype=f32, device=x.device)
        H.spatial_mean_fwd(z, out, B, Tn, HW, 512, z.shape[1])
    return out, tape, (B, Tn, HW, z.shape[1])


def _trunk_input(x):
    """fp32 (B,3,T,H,W) in [0,1] (the reference's batch), or the decoder's uint8 frames (B,T,H,W,3) as
    data.collate_device(video_dtype=torch.uint8) leaves them: the stem's input kernel scales those itself."""
    if x.dtype == torch.uint8:
        if x.dim() != 5 or x.shape[-1] != 3:
            raise ValueError(f"uint8 video batches are (B,T,H,W,3), got {tuple(x.shape)}")
        return x.contiguous()
    return x.contiguous().float()


def prelaunch_video_trunk(enc, x):
    """Issue the trunk forward of `enc` (R3DEncoder / ImageEncoder) NOW, ahead of the autograd node that will own it.

    The next `enc(x)` with this very tensor adopts the result instead of launching again.  PeppaPig.encode_pair uses
    this to put the long video kernels on the GPU before the host spends milliseconds issuing the audio tower, while
    the video autograd node is still created after the audio one -- so the backward pass also starts with video."""
    if not x.is_cuda:
        raise H.PeppaHipError("peppa_amd.video needs a CUDA/HIP tensor (no CPU fallback)")
    xc = _trunk_input(x)
    params = enc.video.trunk_parameters()
    save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    enc._prelaunched = (xc, save) + _video_trunk_launch(enc, xc, save)
    return xc


FORCE_WGRAD_SIDE = False   # A/B switch (tools/ab_step.py): weight-gradient side stream even when the towers are paired


class VideoTrunkFn(torch.autograd.Function):
    """(B,3,T,H,W) fp32 in [0,1] -> spatial means of the trunk output, (B,T',512) fp32."""

    @staticmethod
    def forward(ctx, x, enc, want_grad, *params):
        if not x.is_cuda:
            raise H.PeppaHipError("peppa_amd.video needs a CUDA/HIP tensor (no CPU fallback)")
        x = _trunk_input(x)
        save = want_grad and any(ctx.needs_input_grad)  # grad mode is always off inside Function.forward
        pre, enc._prelaunched = getattr(enc, "_prelaunched", None), None
        if pre is not None and pre[0] is x and pre[1] == save:
            out, tape, dims = pre[2:]       # kernels already in flight (PeppaPig.encode_pair): only adopt the tape
        else:
            out, tape, dims = _video_trunk_launch(enc, x, save)
        ctx.tape, ctx.params, ctx.dims = tape, params, dims
        ctx.precision = H.precision()     # the backward pass runs on the library that produced the tape
        # With the audio tower running beside this trunk (PeppaPig.encode_pair) the weight gradients stay on the trunk's
        # own stream: the audio kernels already fill the GPU's idle corners, and a third stream only made every kernel of
        # the critical chain slower (A/B on one box, full step: 49.7 ms without vs 49.9-50.6 ms with the weight-gradient
        # stream; the trunk alone: 40.0 vs 38.3 ms, so a lone trunk keeps it)
        ctx.overlap_wgrad = FORCE_WGRAD_SIDE or not getattr(enc, "_paired", False)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Tn, HW, Cp = ctx.dims
        grads = grad_dict()
        H.set_precision(ctx.precision)
        with torch.no_grad():
            dz = torch.empty(B * Tn * HW, Cp, dtype=act16(), device=dout.device)
            H.spatial_mean_bwd(dout.contiguous().float(), dz, B, Tn, HW, 512, Cp)
            with L.ZeroPool("video", dout.device):
                V.trunk_backward(ctx.tape, dz, grads, overlap_wgrad=ctx.overlap_wgrad)
        ctx.tape = None
        return (None, None, None) + tuple(grads.get(p) for p in ctx.params)


class R3DEncoder(nn.Module):
    def __init__(self, pretrained=True, project=True, version='r3d_18', pooling='average', weights=None):
        super().__init__()
        self.pretrained = pretrained
        if version not in ('r3d_18', 'mc3_18', 'r2plus1d_18'):
            raise ValueError(f"Invalid version {version}")
        self.video = V.VideoResNet(version)
        if pretrained:
            _pretrained_trunk(self.video, "Kinetics", weights)
        self.project = nn.Linear(512, 512) if project else nn.Identity()
        if pooling == 'attention':
            self.videopool = VideoAttention(512, 128)
        elif pooling == 'average':
            self.videopool = VideoAveragePool()
        else:
            raise ValueError(f"Invalid pooling {pooling}")
        self.norm_kind = "kinetics" if self.pretrained else "peppa"
        self.transform = build_transform(self.norm_kind)

    def mark_pretrained(self):
        """The weights come from a checkpoint of a `pretrained: true` run (checkpoint.load_model): built from random
        init, then overwritten; the input normalisation is the pretrained one (pig/models.py:140)."""
        self.pretrained, self.norm_kind = True, "kinetics"
        self.transform = build_transform(self.norm_kind)

    def forward(self, x):
        feats = VideoTrunkFn.apply(x, self, torch.is_grad_enabled(), *self.video.trunk_parameters())
        if isinstance(self.videopool, VideoAttention) and isinstance(self.project, nn.Linear):
            return self.videopool.attn.pooled_projected(feats, self.project)
        return _project_normalize(self.videopool(feats), self.project)


class ImageEncoder(nn.Module):
    """Static (per-frame) encoder of hparams_static.yaml: resnet18 trunk on every frame, pooled over time."""

    def __init__(self, pretrained=True, project=True, pooling='average', weights=None):
        super().__init__()
        self.pretrained = pretrained
        self.image = V.ResNet18()
        if pretrained:
            _pretrained_trunk(self.image, "ImageNet", weights)
        for param in self.image.fc.parameters():
            param.requires_grad = False
        self.project = nn.Linear(512, 512) if project else nn.Identity()
        self.norm_kind = "imagenet" if self.pretrained else "peppa"
        self.transform = build_transform(self.norm_kind)
        if pooling == 'attention':
            self.pool = Attention(512, 128)
        elif pooling == 'average':
            self.pool = _time_mean                    # x.mean(dim=1) over the frames (pig/models.py:173)
        else:
            raise ValueError(f"Invalid pooling {pooling}")

    @property
    def video(self):   # VideoTrunkFn reads `.video` (the trunk) and `.norm_kind`
        return self.image

    def mark_pretrained(self):
        self.pretrained, self.norm_kind = True, "imagenet"
        self.transform = build_transform(self.norm_kind)

    def forward(self, x):
        feats = VideoTrunkFn.apply(x, self, torch.is_grad_enabled(), *self.image.trunk_parameters())
        if isinstance(self.pool, Attention) and isinstance(self.project, nn.Linear):
            return self.pool.pooled_projected(feats, self.project)
        return _project_normalize(self.pool(feats), self.project)


class VideoAveragePool(nn.Module):
    def __init__(self):
        super().__init__()
        self.pool = torch.nn.AdaptiveAvgPool3d(output_size=(1, 1, 1))

    def forward(self, x):
        """x: spatial means (B,T',512) as produced by VideoTrunkFn; AdaptiveAvgPool3d((1,1,1)) = their mean over T'."""
        return _time_mean(x)


class VideoAttention(nn.Module):
    def __init__(self, in_size=512, hidden_size=128):
        super().__init__()
        self.spatial_avg = torch.nn.AdaptiveAvgPool2d(output_size=(1, 1))
        self.attn = Attention(in_size, hidden_size)

    def forward(self, x):
        """x: spatial means (B,T',512) as produced by VideoTrunkFn."""
        return self.attn(x)


class PeppaPig(_Base):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if pl is not None:
            self.save_hyperparameters(config)
        # (`mi355x: {hardest_negatives: true}`: opt-in hardest-negative mining, an extension; the reference sums all negatives)
        self.loss = TripletLoss(margin=self.config['margin'],
                                hardest=bool((config.get('mi355x', {}) or {}).get('hardest_negatives', False)))
        static = self.config['video'].get('static', False)
        video_config = {key: value for key, value in self.config['video'].items() if key != 'static'}
        extra = config.get('mi355x', {}) or {}     # optional block the reference ignores: local pretrained weights
        if static:
            self.video_encoder = ImageEncoder(**video_config, weights=extra.get('video_weights'))
        else:
            self.video_encoder = R3DEncoder(**video_config, weights=extra.get('video_weights'))
        self.audio_encoder = Wav2VecEncoder(**config['audio'], weights=extra.get('audio_weights'))
        self._logged = {}
        # optional `mi355x:` block (ignored by the reference): run the two encoders on separate HIP streams
        self._overlap = bool(extra.get('overlap_encoders', True))
        if extra.get('deterministic'):      # bitwise-reproducible steps: ordered reductions instead of fp32 atomics
            H.set_deterministic(True)
        self._side_stream = None
        # 16-bit operand type of the towers: "bf16" (default; BASELINE configs[1]) or "fp16" (the reference's own
        # `precision: 16` AMP, hparams_base.yaml:45; BASELINE configs[4]) -- needs peppa_amd.amp.GradScaler around the
        # optimizer like any fp16 AMP run.  `Trainer(precision=...)` / run.py --precision set it too.
        self.precision = "bf16"
        self.set_precision(extra.get('dtype', 'bf16'))

    if pl is None:
        def log(self, name, value, **kwargs):
            self._logged[name] = value

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, hparams_file=None, strict=True, **kwargs):
            """LightningModule.load_from_checkpoint for Lightning-1.4.9 files (pig/evaluation.py:52), without Lightning."""
            from .checkpoint import load_model
            return load_model(cls, checkpoint_path, map_location=map_location, hparams_file=hparams_file, strict=strict)

    def forward(self, batch):
        if isinstance(batch, TargetedTripletBatch):
            a = self.encode_audio(batch.anchor)
            p = self.encode_video(batch.positive)
            n = self.encode_video(batch.negative)
            return EmbeddingTripletBatch(anchor=a, positive=p, negative=n)
        V_ = self.encode_video(batch.video)
        A_ = self.encode_audio(batch.audio)
        return ClipBatch(video=V_, audio=A_, video_duration=batch.video_duration,
                         audio_duration=batch.audio_duration)

    def set_precision(self, precision):
        """"bf16" or "fp16": which build of the HIP library the towers run on from now on."""
        prev = H.set_precision(precision)
        self.precision = H.precision()
        H.set_precision(prev)
        return self

    def encode_video(self, x):
        H.set_precision(self.precision)
        return self.video_encoder(x)

    def encode_audio(self, x):
        H.set_precision(self.precision)
        return self.audio_encoder(x)

    def encode_pair(self, video, audio):
        """Both encoders; they are independent until the loss, so the audio tower (many small
        kernels) runs on a side stream under the video trunk's large ones.  Autograd replays each
        backward on its forward stream, so the overlap also holds for the backward pass."""
        H.set_precision(self.precision)
        if not (self._overlap and video.is_cuda):
            return self.encode_video(video), self.encode_audio(audio)
        main = torch.cuda.current_stream()
        side = self._side_stream = V.tower_stream(video.device)
        side.wait_stream(main)                                   # (the inputs; not the video kernels issued next)
        video = prelaunch_video_trunk(self.video_encoder, video)  # long kernels first: the host runs ahead of them
        with torch.cuda.stream(side):
            A_ = self.encode_audio(audio)
        self.video_encoder._paired = True
        try:
            V_ = self.encode_video(video)                        # adopts the launched trunk; node created after audio
        finally:
            self.video_encoder._paired = False
        main.wait_stream(side)
        A_.record_stream(main)
        return V_, A_

    def training_step(self, batch, batch_idx):
        V_, A_ = self.encode_pair(batch.video, batch.audio)
        V_, A_ = gather_embeddings(V_, A_)  # data-parallel: global negative pool (identity on one GPU)
        loss = self.loss(V_, A_)
        # the reference logs loss.item() (a host sync per step, pig/models.py:264); log the tensor instead
        self.log("train_loss", loss.detach(), prog_bar=True)
        return loss

    def validation_step(self, batch, batch_idx, dataloader_idx=None):
        V_ = self.encode_video(batch.video)
        A_ = self.encode_audio(batch.audio)
        if dataloader_idx == 0:
            self.log("val_loss", self.loss(V_, A_).detach(), prog_bar=True)
            return (V_, A_)
        elif dataloader_idx == 1:
            self.log("valnarr_loss", self.loss(V_, A_).detach(), prog_bar=False)
            return (V_, A_)
        elif dataloader_idx in [2, 3]:
            return (V_, A_, batch.audio_duration)
        raise ValueError(f"Invalid dataloader index {dataloader_idx}")

    def validation_epoch_end(self, outputs):
        out_main, out_narr, out_dia3, out_narr3 = outputs
        for name, out in (("val_rec_fixed", out_main), ("valnarr_rec_fixed", out_narr)):
            Vs, As = zip(*out)
            rec = _metrics.resampled_recall(torch.cat(Vs, dim=0), torch.cat(As, dim=0), size=100, n_samples=500, n=10)
            self.log(name, rec.mean(), prog_bar=True)
        for name, out in (("val_triplet", out_dia3), ("valnarr_triplet", out_narr3)):
            Vs, As, Ds = zip(*out)
            tri = score_triplets(torch.cat(Vs, dim=0), torch.cat(As, dim=0), torch.cat(Ds, dim=0), n_samples=500)
            self.log(name, tri.mean(), prog_bar=True)

    def configure_optimizers(self):
        return opt.BertAdam(self.parameters(), **self.config['optimizer'])


def build_transform(normalization):
    """Returns the (mean, std) the HIP input kernel applies; the reference's permute -> Normalize ->
    permute (pig/models.py:327-342) is fused into the NCDHW->NDHWC load of the stem."""
    if normalization not in V.VIDEO_STATS:
        raise ValueError(f"Unsupported normalization type {normalization}")
    return V.VIDEO_STATS[normalization]
