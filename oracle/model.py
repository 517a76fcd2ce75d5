"""Oracle (test infrastructure): fp32 CPU restatement of the PeppaPig training step.

Follows `pig/models.py:30-61` (pooling heads), `:66-109` (Wav2VecEncoder),
`:113-154` (R3DEncoder), `:156-200` (ImageEncoder), `:204-221` (video pooling),
`:223-265` (PeppaPig.training_step), `:327-342` (build_transform), `pig/loss.py:28-55`
(TripletLoss / contrastive / cosine_matrix), `pig/metrics.py:42-52`
(triplet_accuracy) and `pig/optimization.py:38-43,101-179` (warmup_linear, BertAdam).
Loss / metrics / optimizer are pinned against the live reference by
`oracle/make_golden.py`; see `oracle/__init__.py` for the pinning status.
"""
import math
import torch
from torch import nn
import torch.nn.functional as F

from . import audio as A
from . import video as V

# Normalisation constants: data/out/stats.pt ("peppa"), data/out/kinetics-stats.pt
# ("kinetics") as recovered in SURVEY.md section 0; "imagenet" from pig/models.py:335-336.
VIDEO_STATS = {
    "peppa": ((0.62745821, 0.66273642, 0.66865104), (0.24167268, 0.20884572, 0.27490067)),
    "kinetics": ((0.43216, 0.394666, 0.37645), (0.22803, 0.22145, 0.216989)),
    "imagenet": ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
}


def normalize_video(x, kind):
    """(B,3,T,H,W) per-channel (x-mean)/std; the reference does it in place."""
    mean, std = VIDEO_STATS[kind]
    m = torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1, 1)
    s = torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1, 1)
    return (x - m) / s


class Attention(nn.Module):
    """alpha = softmax_time(out(tanh(hidden(x)))), per feature; sum_t alpha*x."""

    def __init__(self, in_size, hidden_size):
        super().__init__()
        self.hidden = nn.Linear(in_size, hidden_size)
        self.out = nn.Linear(hidden_size, in_size)

    def forward(self, x):
        alpha = torch.softmax(self.out(torch.tanh(self.hidden(x))), dim=1)
        return (alpha * x).sum(dim=1)


class VideoAttention(nn.Module):
    def __init__(self, in_size=512, hidden_size=128):
        super().__init__()
        self.attn = Attention(in_size, hidden_size)

    def forward(self, x):  # (B,C,T,H,W)
        return self.attn(x.mean(dim=(-1, -2)).permute(0, 2, 1))


class VideoAveragePool(nn.Module):
    def forward(self, x):
        return x.mean(dim=(2, 3, 4))


class AveragePool(nn.Module):
    """Quirk kept: AdaptiveAvgPool2d((size,1)) on a (B,T,F) tensor."""

    def __init__(self, size=512):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d((size, 1))

    def forward(self, x):
        return self.pool(x).squeeze(dim=2)


class LastStep(nn.Module):
    def forward(self, x):
        return x[:, -1, :]


class Wav2VecEncoder(nn.Module):
    def __init__(self, path=None, pretrained=False, freeze_feature_extractor=False,
                 freeze_encoder_layers=None, pooling="average", project=True, full=False,
                 dropout=0.1, layer_drop=0.1):
        super().__init__()
        if pretrained:
            raise RuntimeError("oracle: pretrained wav2vec checkpoint is not available offline")
        self.audio = A.wav2vec2_base(28, dropout, layer_drop)
        if freeze_feature_extractor:
            for p in self.audio.feature_extractor.parameters():
                p.requires_grad = False
        if freeze_encoder_layers is not None:
            for i in range(freeze_encoder_layers):
                for p in self.audio.encoder.transformer.layers[i].parameters():
                    p.requires_grad = False
        self.full = full
        self.n_features = 28 if full else 512
        if pooling == "average":
            self.audiopool = AveragePool(self.n_features)
        elif pooling == "attention":
            self.audiopool = Attention(self.n_features, 128)
        elif pooling == "last":
            self.audiopool = LastStep()
        else:
            raise ValueError(f"Invalid pooling: {pooling}")
        self.project = nn.Linear(self.n_features, 512) if project else nn.Identity()

    def forward(self, x):
        wave = x.squeeze(dim=1)
        feats, _ = self.audio(wave) if self.full else self.audio.extract_features(wave)
        return F.normalize(self.project(self.audiopool(feats)), p=2, dim=1)


class R3DEncoder(nn.Module):
    def __init__(self, pretrained=False, project=True, version="r3d_18", pooling="average"):
        super().__init__()
        self.pretrained = pretrained
        self.video = V.VideoResNet18(version)
        self.project = nn.Linear(512, 512) if project else nn.Identity()
        if pooling == "attention":
            self.videopool = VideoAttention(512, 128)
        elif pooling == "average":
            self.videopool = VideoAveragePool()
        else:
            raise ValueError(f"Invalid pooling {pooling}")
        self.norm_kind = "kinetics" if pretrained else "peppa"

    def forward(self, x):
        x = self.video.trunk(normalize_video(x, self.norm_kind))
        return F.normalize(self.project(self.videopool(x)), p=2, dim=1)


class ImageEncoder(nn.Module):
    def __init__(self, pretrained=False, project=True, pooling="average"):
        super().__init__()
        self.pretrained = pretrained
        self.image = V.ResNet18()
        for p in self.image.fc.parameters():
            p.requires_grad = False
        self.project = nn.Linear(512, 512) if project else nn.Identity()
        self.norm_kind = "imagenet" if pretrained else "peppa"
        if pooling == "attention":
            self.pool = Attention(512, 128)
        elif pooling == "average":
            self.pool = lambda x: x.mean(dim=1)
        else:
            raise ValueError(f"Invalid pooling {pooling}")

    def forward(self, x):
        x = normalize_video(x, self.norm_kind).permute(0, 2, 1, 3, 4)
        b, t, c, h, w = x.shape
        im = self.image
        y = x.reshape(b * t, c, h, w)
        y = im.maxpool(im.relu(im.bn1(im.conv1(y))))
        y = im.avgpool(im.layer4(im.layer3(im.layer2(im.layer1(y))))).flatten(1)
        y = self.project(self.pool(y.reshape(b, t, -1)))
        return F.normalize(y, p=2, dim=1)


def cosine_matrix(U, Vv):
    Un = U / U.norm(2, dim=1, keepdim=True)
    Vn = Vv / Vv.norm(2, dim=1, keepdim=True)
    return Un @ Vn.t()


def contrastive(S, margin=0.2):
    """SURVEY 0.2 closed form: (1/N^2) sum_{i!=j} hinge(m+S_ij-S_jj) + hinge(m+S_ij-S_ii)."""
    n = S.size(0)
    d = torch.diag(S)
    col = torch.clamp(margin + S - d.view(1, -1), min=0)
    row = torch.clamp(margin + S - d.view(-1, 1), min=0)
    off = 1.0 - torch.eye(n, dtype=S.dtype)
    return ((col + row) * off).sum() / n ** 2


def contrastive_hardest(S, margin=0.2):
    """Opt-in extension (NOT pig/loss.py): per anchor only the hardest in-batch negative, both directions, mean over N.
    Checker for pp_triplet_loss_hardest_fwd."""
    N = S.shape[0]
    d = torch.diag(S)
    if N == 1:
        return S.sum() * 0.0
    off = S.masked_fill(torch.eye(N, dtype=torch.bool, device=S.device), float("-inf"))
    hr = off.max(dim=1).values          # video i against its hardest audio negative
    hc = off.max(dim=0).values          # audio j against its hardest video negative
    return (torch.relu(margin + hr - d) + torch.relu(margin + hc - d)).sum() / N


class TripletLoss(nn.Module):
    def __init__(self, margin, hardest=False):
        super().__init__()
        self.margin, self.hardest = margin, hardest

    def forward(self, Vv, Aa):
        S = cosine_matrix(Vv, Aa)
        return contrastive_hardest(S, self.margin) if self.hardest else contrastive(S, self.margin)


def triplet_accuracy(anchor, positive, negative, dim=1, discrete=True):
    diff = F.cosine_similarity(anchor, positive, dim=dim) - F.cosine_similarity(anchor, negative, dim=dim)
    return (torch.sign(diff) + 1) / 2 if discrete else diff


def warmup_linear(x, warmup=0.002):
    return x / warmup if x < warmup else max((x - 1.0) / (warmup - 1.0), 0)


def bertadam_step(params, grads, state, lr, warmup=-1, t_total=-1, b1=0.9, b2=0.999, e=1e-6,
                  weight_decay=0.01, max_grad_norm=1.0):
    """One BertAdam step over parallel lists (functional restatement of
    pig/optimization.py:101-179): per-tensor clip, no bias correction, eps outside
    sqrt, decoupled decay on every tensor, schedule multiplier from the per-tensor step."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        st = state.setdefault(i, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        if max_grad_norm > 0:
            coef = max_grad_norm / (g.norm(2) + 1e-6)
            g = g * torch.clamp(coef, max=1.0)
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        upd = st["m"] / (st["v"].sqrt() + e)
        if weight_decay > 0:
            upd = upd + weight_decay * p
        lr_t = lr * warmup_linear(st["step"] / t_total, warmup) if t_total != -1 else lr
        p.sub_(lr_t * upd)
        st["step"] += 1


class PeppaPigOracle(nn.Module):
    """Same construction rule as pig/models.py:224-236 (static -> ImageEncoder)."""

    def __init__(self, config, dropout=0.0, layer_drop=0.0):
        super().__init__()
        self.config = config
        self.loss = TripletLoss(margin=config["margin"])
        vcfg = {k: v for k, v in config["video"].items() if k != "static"}
        self.video_encoder = ImageEncoder(**vcfg) if config["video"].get("static", False) \
            else R3DEncoder(**vcfg)
        self.audio_encoder = Wav2VecEncoder(**config["audio"], dropout=dropout, layer_drop=layer_drop)

    def encode_video(self, x):
        return self.video_encoder(x)

    def encode_audio(self, x):
        return self.audio_encoder(x)

    def training_loss(self, video, audio):
        return self.loss(self.encode_video(video), self.encode_audio(audio))


def synthetic_batch(batch, frames, size, samples, seed=1234):
    """SURVEY 8d synthetic inputs: video U[0,1), audio 0.1*N(0,1), CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    H, W = (size, size) if isinstance(size, int) else size
    video = torch.rand(batch, 3, frames, H, W, generator=g)
    audio = 0.1 * torch.randn(batch, 1, samples, generator=g)
    return video, audio
