"""The part of `pig.data` the training step touches: the Clip / ClipBatch containers and `collate`
(pig/data.py:49-65).  Decoding real clips (moviepy/ffmpeg, the private dataset) is out of scope."""
from dataclasses import dataclass
import torch

from .util import pad_audio_batch, pad_video_batch


@dataclass
class Clip:
    """Video clip with associated audio (pig/data.py:28-37): the durations are SECONDS (`clip.duration` /
    `clip.audio.duration` of the moviepy clip, pig/data.py:72-73), not frame or sample counts."""
    video: torch.Tensor
    audio: torch.Tensor
    video_duration: float = None
    audio_duration: float = None
    filename: str = None
    offset: float = None
    index: int = None

    @property
    def duration(self):
        """`pig.triplet.triplets` groups clips by `x.duration` (pig/triplet.py:110); the validation path matches by
        audio duration (pig/models.py:288, pig/triplet.py:44)."""
        return self.audio_duration


@dataclass
class ClipBatch:
    video: torch.Tensor
    audio: torch.Tensor
    video_duration: torch.Tensor = None
    audio_duration: torch.Tensor = None

    def to(self, device, non_blocking=False):
        mv = lambda t: t if t is None else t.to(device, non_blocking=non_blocking)
        return ClipBatch(mv(self.video), mv(self.audio), mv(self.video_duration), mv(self.audio_duration))


def collate(data):
    """pig/data.py:60-65: zero-pad clips to the longest in the batch along time (video) / samples (audio); the batch
    durations are the clips' durations in seconds (validation matches triplets on `audio_duration`)."""
    video, audio, vlen, alen = zip(*[(x.video, x.audio, x.video_duration, x.audio_duration) for x in data])
    return ClipBatch(video=pad_video_batch(video), audio=pad_audio_batch(audio),
                     video_duration=torch.tensor(vlen), audio_duration=torch.tensor(alen))


@dataclass
class RawClip:
    """A decoded clip as the decoder hands it over, before `featurize` (pig/data.py:66-77) touches it:
    `frames` uint8 (T,H,W,3) (moviepy's `iter_frames()` stacked), `audio` float32 (1,L) (mono samples)."""
    frames: torch.Tensor
    audio: torch.Tensor
    video_duration: float = None
    audio_duration: float = None


def _ragged_table(tensors, lengths, device, align=16):
    """Pack host tensors into ONE pinned buffer (each at a 16-byte offset), copy it with one H2D transfer and return
    (device buffer, int64 device table [n][2] = {device pointer, length})."""
    sizes = [t.numel() * t.element_size() for t in tensors]
    offs, total = [], 0
    for sz in sizes:
        offs.append(total)
        total += (sz + align - 1) // align * align
    host = torch.empty(max(total, align), dtype=torch.uint8).pin_memory()
    for t, off, sz in zip(tensors, offs, sizes):
        host[off:off + sz] = t.contiguous().reshape(-1).view(torch.uint8)
    buf = host.to(device, non_blocking=True)
    base = buf.data_ptr()
    table = torch.tensor([[base + off, n] for off, n in zip(offs, lengths)], dtype=torch.int64).pin_memory()
    return buf, table.to(device, non_blocking=True)


def collate_device(clips, device="cuda", video_dtype=torch.float32, fps=10, audio_sample_rate=44100):
    """`collate` (pig/data.py:60-65) for RawClips, on the GPU: one byte per sample crosses PCIe, the /255 scaling,
    (T,H,W,C)->(C,T,H,W) transpose, zero-padding and stacking run in HIP kernels (csrc/collate.hip).

    video_dtype=torch.float32: `ClipBatch.video` is the reference's fp32 (B,3,Tmax,H,W) batch in [0,1], bit-identical
    to featurize + pad_video_batch.  video_dtype=torch.uint8: it stays a padded uint8 (B,Tmax,H,W,3) batch, which
    `encode_video` accepts directly (the stem's input kernel scales and normalises it) -- 4x less HBM traffic and no
    fp32 copy; the embeddings are bit-identical to the fp32 route.

    Batch durations are SECONDS like the reference's (pig/data.py:64-65, 72-73); a RawClip without them gets
    frames / `fps` and samples / `audio_sample_rate` (10 fps snippets, pig/preprocess.py:45-47; 44.1 kHz,
    pig/data.py:26)."""
    from . import hip as H
    if len(clips) == 0:
        raise ValueError("collate_device: empty batch")
    shapes = {tuple(c.frames.shape[1:]) for c in clips}
    if len(shapes) != 1 or next(iter(shapes))[-1] != 3:
        raise ValueError(f"collate_device: frames must all be (T,H,W,3) with one frame size, got {sorted(shapes)}")
    for c in clips:
        if c.frames.dtype != torch.uint8 or c.audio.dtype != torch.float32 or c.audio.dim() != 2 or c.audio.shape[0] != 1:
            raise ValueError("collate_device: frames must be uint8 (T,H,W,3) and audio float32 (1,L)")
        if c.frames.shape[0] == 0:
            raise ValueError("Clip has zero frames.")                         # pig/data.py:77
    Hh, W, _ = next(iter(shapes))
    n = len(clips)
    T = [c.frames.shape[0] for c in clips]
    Ls = [c.audio.shape[1] for c in clips]
    Tmax, Lmax = max(T), max(Ls)
    dev = torch.device(device)
    if video_dtype == torch.float32:
        vbuf, vtab = _ragged_table([c.frames for c in clips], T, dev)
        video = torch.empty(n, 3, Tmax, Hh, W, dtype=torch.float32, device=dev)
        H.collate_video_u8(vtab, n, Tmax, Hh, W, video)
    elif video_dtype == torch.uint8:
        vbuf, vtab = _ragged_table([c.frames for c in clips], [t * Hh * W * 3 for t in T], dev)
        video = torch.empty(n, Tmax, Hh, W, 3, dtype=torch.uint8, device=dev)
        H.collate_rows(vtab, n, Tmax * Hh * W * 3, video)
    else:
        raise ValueError(f"video_dtype must be torch.float32 or torch.uint8, got {video_dtype}")
    abuf, atab = _ragged_table([c.audio for c in clips], [4 * l for l in Ls], dev)
    audio = torch.empty(n, 1, Lmax, dtype=torch.float32, device=dev)
    H.collate_rows(atab, n, 4 * Lmax, audio)
    for t in (vbuf, vtab, abuf, atab):
        t.record_stream(torch.cuda.current_stream(dev))
    dur = lambda xs, fallback: torch.tensor([f if x is None else x for x, f in zip(xs, fallback)])
    return ClipBatch(video=video, audio=audio,
                     video_duration=dur([c.video_duration for c in clips], [t / fps for t in T]),
                     audio_duration=dur([c.audio_duration for c in clips], [l / audio_sample_rate for l in Ls]))


def synthetic_batch(batch, frames, size, samples, seed=1234, device="cpu"):
    """SURVEY 8d synthetic clips: video U[0,1) (frame/255), audio 0.1*N(0,1), CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    H, W = (size, size) if isinstance(size, int) else size     # (H, W): the reference's own frames are 100 x 180
    video = torch.rand(batch, 3, frames, H, W, generator=g)
    audio = 0.1 * torch.randn(batch, 1, samples, generator=g)
    # seconds, as in the reference's batches: 10 fps video snippets, 16 kHz synthetic audio
    return ClipBatch(video.to(device), audio.to(device), torch.full((batch,), frames / 10.0),
                     torch.full((batch,), samples / 16000.0))


def synthetic_structured_batch(batch, frames, size, samples, seed=4321, device="cpu"):
    """Synthetic clips that differ from each other the way real clips do (a colour cast, smooth spatio-temporal
    texture, a moving blob; audio = a few amplitude-modulated tones): iid-noise clips (`synthetic_batch`) are
    statistically identical, so a random-init model maps them all to almost the same embedding and any ranking of
    them (triplet accuracy, recall) is decided by rounding noise.  Same value ranges as `synthetic_batch`."""
    import math
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    base = 0.2 + 0.6 * r(batch, 3, 1, 1, 1)
    coarse = n(batch, 3, max(2, frames // 4 + 1), 7, 7)
    tex = F.interpolate(coarse, size=(frames, size, size), mode="trilinear", align_corners=True)
    contrast = 0.1 + 0.3 * r(batch, 1, 1, 1, 1)
    tt = torch.linspace(0, 1, frames).view(1, frames, 1, 1)
    yy = torch.linspace(0, 1, size).view(1, 1, size, 1)
    xx = torch.linspace(0, 1, size).view(1, 1, 1, size)
    p0, vel = r(batch, 2, 1, 1, 1), 0.6 * (r(batch, 2, 1, 1, 1) - 0.5)
    rad = 0.05 + 0.15 * r(batch, 1, 1, 1)
    blob = torch.exp(-((yy - (p0[:, 0] + vel[:, 0] * tt)) ** 2 + (xx - (p0[:, 1] + vel[:, 1] * tt)) ** 2) / (2 * rad ** 2))
    colour = r(batch, 3, 1, 1, 1) - 0.5
    video = (base + contrast * tex + colour * blob.unsqueeze(1) + 0.02 * n(batch, 3, frames, size, size)).clamp_(0, 1)
    t = torch.arange(samples).view(1, 1, samples) / 16000.0
    freq = 100.0 * torch.exp(math.log(40.0) * r(batch, 3, 1))            # 100 Hz .. 4 kHz
    mod = 0.5 + 0.5 * torch.sin(2 * math.pi * (0.5 + 4 * r(batch, 3, 1)) * t + 6.28 * r(batch, 3, 1))
    amp = 0.02 + 0.08 * r(batch, 3, 1)
    audio = (amp * mod * torch.sin(2 * math.pi * freq * t + 6.28 * r(batch, 3, 1))).sum(dim=1, keepdim=True)
    audio = audio + 0.01 * n(batch, 1, samples)
    return ClipBatch(video.to(device), audio.float().to(device), torch.full((batch,), frames / 10.0),
                     torch.full((batch,), samples / 16000.0))
