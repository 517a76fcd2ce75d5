"""`pig.transforms` (pig/transforms.py:5-8).  On the HIP path the channel/time swap is fused into the
stem's NCDHW -> NDHWC load, so this module only keeps the API."""
from torch import nn


class SwapCT(nn.Module):
    def forward(self, vid):
        return vid.permute(0, 2, 1, 3, 4)
