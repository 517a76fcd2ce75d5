"""`pig.loss` on MI355X (pig/loss.py:28-55): TripletLoss = contrastive(cosine_matrix(V, A), margin),
the all-negatives hinge summed in both directions and divided by N^2 (there is no
hardest-negative mining in the reference).  Forward and backward run in libpeppa_hip.so."""
import torch

from . import hip as H
from .hip import f32


class TripletLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, V, A, margin, hardest=False):
        if not (V.is_cuda and A.is_cuda):
            raise H.PeppaHipError("peppa_amd.loss needs CUDA/HIP tensors (no CPU fallback)")
        V, A = V.contiguous().float(), A.contiguous().float()
        N, D = V.shape
        ws = torch.empty(H.triplet_workspace_bytes(N, D) // 4, dtype=f32, device=V.device)
        loss = torch.empty(1, dtype=f32, device=V.device)
        (H.triplet_loss_hardest_fwd if hardest else H.triplet_loss_fwd)(V, A, float(margin), loss, ws)
        ctx.save_for_backward(V, A, ws)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        V, A, ws = ctx.saved_tensors
        dV, dA = torch.empty_like(V), torch.empty_like(A)
        H.triplet_loss_bwd(V, A, dloss.reshape(1).contiguous().float(), ws, dV, dA)
        return dV, dA, None, None


class TripletLoss(torch.nn.Module):
    """`TripletLoss(margin)` is the reference's loss (pig/loss.py:28-39): the hinge summed over ALL in-batch negatives.
    `hardest=True` is an opt-in extension that is NOT in the reference (yaml `mi355x: {hardest_negatives:
    true}`): per anchor only its hardest in-batch negative counts, found by a wavefront-64 arg-max
    (pp_triplet_loss_hardest_fwd): (1/N) sum_i [relu(m + max_{j != i} S_ij - S_ii) + relu(m + max_{j != i} S_ji - S_ii)]."""

    def __init__(self, margin, hardest=False):
        super(TripletLoss, self).__init__()
        self.margin = margin
        self.hardest = bool(hardest)

    def forward(self, V, A):
        """V, A: (N, D) embeddings (video, audio) -> scalar loss."""
        return TripletLossFn.apply(V, A, self.margin, self.hardest)


class CosineMatrixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, V):
        if not (U.is_cuda and V.is_cuda):
            raise H.PeppaHipError("peppa_amd.loss needs CUDA/HIP tensors (no CPU fallback)")
        U, V = U.contiguous().float(), V.contiguous().float()
        out = torch.empty(U.shape[0], V.shape[0], dtype=f32, device=U.device)
        H.cosine_matrix(U, V, out)
        ctx.save_for_backward(U, V)
        return out

    @staticmethod
    def backward(ctx, dS):
        U, V = ctx.saved_tensors
        dU, dV = torch.empty_like(U), torch.empty_like(V)
        H.cosine_matrix_bwd(U, V, dS.contiguous().float(), dU, dV)
        return dU, dV


class ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M, margin):
        if not M.is_cuda:
            raise H.PeppaHipError("peppa_amd.loss needs CUDA/HIP tensors (no CPU fallback)")
        if M.dim() != 2 or M.shape[0] != M.shape[1]:
            raise ValueError(f"contrastive: square similarity matrix expected, got {tuple(M.shape)}")
        M = M.contiguous().float()
        loss = torch.empty(1, dtype=f32, device=M.device)
        H.contrastive_fwd(M, float(margin), loss)
        ctx.save_for_backward(M)
        ctx.margin = float(margin)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (M,) = ctx.saved_tensors
        dM = torch.empty_like(M)
        H.contrastive_bwd(M, ctx.margin, dloss.reshape(1).contiguous().float(), dM)
        return dM, None


def cosine_matrix(U, V):
    "Returns the matrix of cosine similarity between each row of U and each row of V (differentiable, pig/loss.py:51-55)."
    return CosineMatrixFn.apply(U, V)


def contrastive(M, margin=0.2):
    "Returns contrastive margin loss over similarity matrix M (differentiable, pig/loss.py:41-48)."
    return ContrastiveFn.apply(M, margin)


class MILNCELoss(torch.nn.Module):
    """Present in the reference (pig/loss.py:5-26) but never instantiated by any config."""

    def forward(self, V, A):
        raise NotImplementedError("MILNCELoss is dead code in the reference and is not on the HIP path")
