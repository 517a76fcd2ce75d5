"""`pig.loss` on MI355X (pig/loss.py:28-55): TripletLoss = contrastive(cosine_matrix(V, A), margin),
the all-negatives hinge summed in both directions and divided by N^2 (there is no
hardest-negative mining in the reference).  Forward and backward run in libpeppa_hip.so."""
import torch

from . import hip as H
from .hip import f32


class TripletLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, V, A, margin):
        if not (V.is_cuda and A.is_cuda):
            raise H.PeppaHipError("peppa_amd.loss needs CUDA/HIP tensors (no CPU fallback)")
        V, A = V.contiguous().float(), A.contiguous().float()
        N, D = V.shape
        ws = torch.empty(H.triplet_workspace_bytes(N, D) // 4, dtype=f32, device=V.device)
        loss = torch.empty(1, dtype=f32, device=V.device)
        H.triplet_loss_fwd(V, A, float(margin), loss, ws)
        ctx.save_for_backward(V, A, ws)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        V, A, ws = ctx.saved_tensors
        dV, dA = torch.empty_like(V), torch.empty_like(A)
        H.triplet_loss_bwd(V, A, dloss.reshape(1).contiguous().float(), ws, dV, dA)
        return dV, dA, None


class TripletLoss(torch.nn.Module):
    def __init__(self, margin):
        super(TripletLoss, self).__init__()
        self.margin = margin

    def forward(self, V, A):
        """V, A: (N, D) embeddings (video, audio) -> scalar loss."""
        return TripletLossFn.apply(V, A, self.margin)


def cosine_matrix(U, V):
    "Matrix of cosine similarities between the rows of U and the rows of V (no gradient)."
    U, V = U.detach().contiguous().float(), V.detach().contiguous().float()
    out = torch.empty(U.shape[0], V.shape[0], dtype=f32, device=U.device)
    H.cosine_matrix(U, V, out)
    return out


def contrastive(M, margin=0.2):
    "Contrastive margin loss over a similarity matrix M (forward value only)."
    if M.requires_grad:
        raise NotImplementedError("contrastive(M) on a pre-computed matrix has no backward on the HIP path; "
                                  "use TripletLoss(margin)(V, A)")
    loss = torch.empty(1, dtype=f32, device=M.device)
    H.contrastive_fwd(M.contiguous().float(), float(margin), loss)
    return loss.reshape(())


class MILNCELoss(torch.nn.Module):
    """Present in the reference (pig/loss.py:5-26) but never instantiated by any config."""

    def forward(self, V, A):
        raise NotImplementedError("MILNCELoss is dead code in the reference and is not on the HIP path")
