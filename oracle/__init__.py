"""CPU oracle for the PeppaPig training-step hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 CPU restatement of
the reference algorithm (gchrupala/peppa `pig/models.py`, `pig/loss.py`,
`pig/metrics.py`, `pig/optimization.py` plus the pinned third-party backbones
torchvision 0.10.1 `video.resnet` / torchaudio 0.9.1 `wav2vec2`).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The product (`peppa_amd`) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * loss / metrics / BertAdam: pinned against the reference itself, imported live
    in the build container (`oracle/make_golden.py` -> `tests/golden/ref_*.npz`).
  * wav2vec2-base: architecture pinned against an independent implementation
    (HF `transformers` Wav2Vec2Model built from a local config, eval mode).
  * resnet18 (static image encoder): architecture pinned against an independent
    implementation (HF `transformers` ResNetModel from a local config, eval and train mode).
  * r2plus1d_18 / r3d_18 / mc3_18: the reference's own tests pin nothing (it has
    none), torchvision is not installed and no second implementation exists in the
    container -> PARITY UNPINNED for the video trunk beyond parameter counts, mid-plane
    widths, stage shapes and GMAC totals.
"""
