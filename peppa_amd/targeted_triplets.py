"""The part of `pig.targeted_triplets` the model touches (pig/targeted_triplets.py:19-33,162-166): the `Triplet` /
`TripletBatch` containers `PeppaPig.forward` dispatches on (pig/models.py:238-242) and `collate_triplets`.
The minimal-pair datasets themselves need moviepy, spaCy and the private dataset and are out of scope."""
from dataclasses import dataclass
import torch

from .util import pad_audio_batch, pad_video_batch

FPS = 10


@dataclass
class Triplet:
    anchor: torch.Tensor
    positive: torch.Tensor
    negative: torch.Tensor
    video_duration: float = None
    audio_duration: float = None


@dataclass
class TripletBatch:
    anchor: torch.Tensor
    positive: torch.Tensor
    negative: torch.Tensor

    def to(self, device, non_blocking=False):
        return TripletBatch(*(t.to(device, non_blocking=non_blocking) for t in (self.anchor, self.positive, self.negative)))


def collate_triplets(data):
    """pig/targeted_triplets.py:162-166: anchor audio zero-padded along samples, both videos along time."""
    anchor, pos, neg = zip(*[(x.anchor, x.positive, x.negative) for x in data])
    return TripletBatch(anchor=pad_audio_batch(anchor), positive=pad_video_batch(pos), negative=pad_video_batch(neg))


class PeppaTargetedTripletDataset:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the targeted-triplet eval sets need moviepy and the private Peppa dataset")


PeppaTargetedTripletCachedDataset = PeppaTargetedTripletDataset
