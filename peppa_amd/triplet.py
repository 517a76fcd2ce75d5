"""`pig.triplet` (pig/triplet.py:17-121): duration-matched triplet pairing and scoring.

Deviation: the reference's `score_triplets` raises NameError at HEAD
(`success.append(success)`) and returns a dict where its callers expect a tensor; this module
implements the contract the callers use (pig/models.py:311-312, pig/evaluation.py:59-60): a
length-`n_samples` tensor of mean triplet accuracies.  The dataset-backed `TripletScorer` needs
moviepy and the private dataset and is out of scope."""
import random
from dataclasses import dataclass
import torch

from .metrics import triplet_accuracy
from .util import grouped, shuffled


@dataclass
class Triplet:
    anchor: ...
    positive: ...
    negative: ...


@dataclass
class TripletBatch:
    anchor: ...
    positive: ...
    negative: ...


class TripletScorer:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("TripletScorer needs pig.data.PeppaPigDataset (moviepy + the private dataset)")


def pairs(xs):
    return [xs[i:i + 2] for i in range(0, len(xs), 2) if len(xs[i:i + 2]) == 2]


def _triplets(clips, criterion):
    for _, items in grouped(clips, key=criterion):
        for pair in pairs(shuffled(list(items))):
            target, distractor = random.sample(pair, 2)
            yield (target, distractor)


def triplets(clips):
    """(audio, matching video, distractor video) triplets matched by duration."""
    for target, distractor in _triplets(clips, lambda x: x.duration):
        yield Triplet(anchor=target.audio, positive=target.video, negative=distractor.video)


def _sample_indices(duration):
    dur = duration.tolist() if torch.is_tensor(duration) else list(duration)
    idx = list(_triplets(range(len(dur)), lambda i: dur[i]))
    if not idx:
        raise ValueError("no two clips share a duration: cannot form triplets")
    pos, neg = zip(*idx)
    return torch.tensor(pos), torch.tensor(neg)


def score_triplets(video, audio, duration, n_samples=100):
    accuracy = []
    for _ in range(n_samples):
        pos_idx, neg_idx = _sample_indices(duration)
        pos_idx, neg_idx = pos_idx.to(video.device), neg_idx.to(video.device)
        acc = triplet_accuracy(anchor=audio[pos_idx], positive=video[pos_idx], negative=video[neg_idx])
        accuracy.append(acc.mean())
    return torch.stack(accuracy)


def comparative_score_triplets(video_set, audio_set, duration, n_samples=100):
    success = [[] for _ in range(len(video_set))]
    length = []
    for _ in range(n_samples):
        pos_idx, neg_idx = _sample_indices(duration)
        for i in range(len(video_set)):
            dev = video_set[i].device
            p, n = pos_idx.to(dev), neg_idx.to(dev)
            success[i].append(triplet_accuracy(anchor=audio_set[i][p], positive=video_set[i][p],
                                               negative=video_set[i][n], discrete=False))
        length.append(duration[pos_idx.to(duration.device)] if torch.is_tensor(duration) else
                      torch.tensor([duration[j] for j in pos_idx.tolist()]))
    return {'success': [torch.cat(s) for s in success], 'duration': torch.cat(length)}
