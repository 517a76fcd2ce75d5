"""`pig.metrics` (pig/metrics.py:7-81) on the HIP path: triplet_accuracy, the similarity matrix and the rank-based
recalls (one launch for all index sets of resampled_recall instead of n_samples x size host-side argsorts)."""
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


def _recall_table(candidates, references, correct, Nmax, idx=None):
    """recall@1..Nmax per query row, evaluated on the device: (nsets, Nmax, rows) fp32 on the CPU.

    Positions are those of the reference's `row.argsort()` over 1 - cosine (ties by index); instead of sorting,
    every target counts the candidates ahead of it (pp_recall_at_n)."""
    S = cosine_matrix(references, candidates)                       # (len(references), len(candidates)) on the device
    if correct is not None:
        correct = (correct != 0).to(device=S.device, dtype=torch.uint8).contiguous()
        if correct.shape != S.shape:
            raise ValueError(f"correct has shape {tuple(correct.shape)}, expected {tuple(S.shape)}")
    if idx is not None:
        idx = idx.to(device=S.device, dtype=torch.int32).contiguous()
    nsets, rows = (idx.shape if idx is not None else (1, S.shape[0]))
    out = torch.empty(nsets, Nmax, rows, dtype=f32, device=S.device)
    H.recall_at_n(S, idx, correct, Nmax, out)
    return out.cpu()


def recall_at_n(candidates, references, correct, n=1):
    return _recall_table(candidates, references, correct, n)[0, n - 1]


def recall_at_1_to_n(candidates, references, correct, N=1):
    table = _recall_table(candidates, references, correct, N)[0]   # (N, rows)
    return torch.cat([torch.zeros(1, table.shape[1]), table], dim=0)   # recall at 0 is always zero


def sample_indices(x, size):
    return torch.randperm(x.size(0))[:size]


def _index_sets(x, size, n_samples):
    # one torch.randperm per sample from the global CPU generator, exactly the draws the reference makes
    return torch.stack([sample_indices(x, size) for _ in range(n_samples)])


def resampled_recall(candidates, references, size=100, n_samples=100, n=1):
    """(n_samples, size): all index sets are ranked by ONE launch on one similarity matrix (the reference does
    n_samples x size host-side argsorts); targets are the diagonal (torch.eye in the reference)."""
    assert len(candidates) == len(references)
    assert len(candidates) >= size
    ix = _index_sets(candidates, size, n_samples)
    return _recall_table(candidates, references, None, n, idx=ix)[:, n - 1, :]


def resampled_recall_at_1_to_n(candidates, references, size=100, n_samples=100, N=1):
    assert len(candidates) == len(references)
    assert len(candidates) >= size
    ix = _index_sets(candidates, size, n_samples)
    table = _recall_table(candidates, references, None, N, idx=ix)   # (n_samples, N, size)
    return torch.cat([torch.zeros(n_samples, 1, size), table], dim=1)
