"""`pig.metrics` (pig/metrics.py:7-81).  triplet_accuracy and the similarity matrix run on the HIP
path; the rank-based recall loops stay host-side like the reference's (validation only; a batched
device version is listed under "next" in DESIGN.md)."""
import torch

from . import hip as H
from .hip import f32
from .loss import cosine_matrix


def triplet_accuracy(anchor, positive, negative, dim=1, discrete=True):
    """(sign(cos(a,p) - cos(a,n)) + 1) / 2 per row (ties -> 0.5), or the raw difference."""
    if dim != 1 or anchor.dim() != 2:
        raise NotImplementedError("triplet_accuracy on the HIP path expects (M, D) inputs and dim=1")
    a, p, n = (t.detach().contiguous().float() for t in (anchor, positive, negative))
    out = torch.empty(a.shape[0], dtype=f32, device=a.device)
    H.triplet_accuracy(a, p, n, discrete, out)
    return out


def batch_triplet_accuracy(batch):
    return triplet_accuracy(batch.anchor, batch.positive, batch.negative)


def _ranked(candidates, references):
    distances = 1 - cosine_matrix(references, candidates)
    return distances.cpu()


def recall_at_n(candidates, references, correct, n=1):
    distances = _ranked(candidates, references)
    correct = correct.cpu()
    recall = []
    for j, row in enumerate(distances):
        topn = row.argsort()[:n]
        target = torch.nonzero(correct[j])[:, 0]
        overlap = (topn.unsqueeze(dim=0) == target.unsqueeze(dim=1)).sum().item()
        recall.append(overlap / len(target))
    return torch.tensor(recall)


def recall_at_1_to_n(candidates, references, correct, N=1):
    distances = _ranked(candidates, references)
    correct = correct.cpu()
    recall = [[] for _ in range(0, N + 1)]
    recall[0] = [0 for _ in distances]
    for j, row in enumerate(distances):
        ranked = row.argsort()
        target = torch.nonzero(correct[j])[:, 0]
        for n in range(1, N + 1):
            overlap = (ranked[:n].unsqueeze(dim=0) == target.unsqueeze(dim=1)).sum().item()
            recall[n].append(overlap / len(target))
    return torch.tensor(recall)


def sample_indices(x, size):
    return torch.randperm(x.size(0))[:size]


def resampled_recall(candidates, references, size=100, n_samples=100, n=1):
    assert len(candidates) == len(references)
    assert len(candidates) >= size
    result = []
    for _ in range(n_samples):
        ix = sample_indices(candidates, size).to(candidates.device)
        X, Y = candidates[ix], references[ix]
        result.append(recall_at_n(X, Y, torch.eye(X.shape[0]), n=n))
    return torch.stack(result)


def resampled_recall_at_1_to_n(candidates, references, size=100, n_samples=100, N=1):
    assert len(candidates) == len(references)
    assert len(candidates) >= size
    result = []
    for _ in range(n_samples):
        ix = sample_indices(candidates, size).to(candidates.device)
        X, Y = candidates[ix], references[ix]
        result.append(recall_at_1_to_n(X, Y, torch.eye(X.shape[0]), N=N))
    return torch.stack(result)
