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


def synthetic_batch(batch, frames, size, samples, seed=1234, device="cpu"):
    """SURVEY 8d synthetic clips: video U[0,1) (frame/255), audio 0.1*N(0,1), CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    video = torch.rand(batch, 3, frames, size, size, generator=g)
    audio = 0.1 * torch.randn(batch, 1, samples, generator=g)
    dur = torch.full((batch,), float(samples))
    return ClipBatch(video.to(device), audio.to(device), torch.full((batch,), float(frames)), dur)
