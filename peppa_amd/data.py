"""The part of `pig.data` the training step touches: the Clip / ClipBatch containers and `collate`
(pig/data.py:49-65).  Decoding real clips (moviepy/ffmpeg, the private dataset) is out of scope."""
from dataclasses import dataclass
import torch

from .util import pad_audio_batch, pad_video_batch


@dataclass
class Clip:
    video: torch.Tensor
    audio: torch.Tensor
    duration: float = None
    filename: str = None
    offset: float = None
    index: int = None


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
    """Zero-pad clips to the longest in the batch along time (video) / samples (audio)."""
    video, audio = zip(*[(x.video, x.audio) for x in data])
    return ClipBatch(video=pad_video_batch(video), audio=pad_audio_batch(audio),
                     video_duration=torch.tensor([x.video.shape[1] for x in data]),
                     audio_duration=torch.tensor([x.audio.shape[1] for x in data]))


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


def collate_device(clips, device="cuda", video_dtype=torch.float32):
    """`collate` (pig/data.py:60-65) for RawClips, on the GPU: one byte per sample crosses PCIe, the /255 scaling,
    (T,H,W,C)->(C,T,H,W) transpose, zero-padding and stacking run in HIP kernels (csrc/collate.hip).

    video_dtype=torch.float32: `ClipBatch.video` is the reference's fp32 (B,3,Tmax,H,W) batch in [0,1], bit-identical
    to featurize + pad_video_batch.  video_dtype=torch.uint8: it stays a padded uint8 (B,Tmax,H,W,3) batch, which
    `encode_video` accepts directly (the stem's input kernel scales and normalises it) -- 4x less HBM traffic and no
    fp32 copy; the embeddings are bit-identical to the fp32 route."""
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
                     video_duration=dur([c.video_duration for c in clips], T),
                     audio_duration=dur([c.audio_duration for c in clips], Ls))


def synthetic_batch(batch, frames, size, samples, seed=1234, device="cpu"):
    """SURVEY 8d synthetic clips: video U[0,1) (frame/255), audio 0.1*N(0,1), CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    video = torch.rand(batch, 3, frames, size, size, generator=g)
    audio = 0.1 * torch.randn(batch, 1, samples, generator=g)
    dur = torch.full((batch,), float(samples))
    return ClipBatch(video.to(device), audio.to(device), torch.full((batch,), float(frames)), dur)
