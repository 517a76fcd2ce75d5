"""`pig.util` tensor helpers (pig/util.py:9-35).  cosine_matrix runs on the HIP path; the pad/crop
helpers are host-side collation; shuffled/grouped drive the duration-matched triplet pairing."""
import random
from itertools import groupby
import torch
import torch.nn.functional as F

from .loss import cosine_matrix  # noqa: F401  (same function, pig/util.py:9-13 duplicates pig/loss.py)


def identity(x):
    return x


def crop_audio_batch(audio):
    size = min(x.shape[1] for x in audio)
    return torch.stack([x[:, :size] for x in audio])


def pad_audio_batch(audio):
    size = max(x.shape[1] for x in audio)
    return torch.stack([F.pad(x, (0, size - x.shape[1]), 'constant', 0) for x in audio])


def crop_video_batch(video):
    size = min(x.shape[1] for x in video)
    return torch.stack([x[:, :size, :, :] for x in video])


def pad_video_batch(video):
    size = max(x.shape[1] for x in video)
    return torch.stack([F.pad(x, (0, 0, 0, 0, 0, size - x.shape[1]), 'constant', 0) for x in video])


def shuffled(xs):
    return sorted(xs, key=lambda _: random.random())


def grouped(xs, key=lambda x: x):
    return groupby(sorted(xs, key=key), key=key)
