"""Parity at the REFERENCE's own clip geometry (/root/reference/hparams_base.yaml:9,14-16,23,
pig/preprocess.py:45-47): target_size [180, 100] = frames of H 100 x W 180, 2.3 s at 10 fps = 23 frames,
44.1 kHz audio fed unresampled = 101 429 samples -> 316 wav2vec2 frames, micro-batch 8.  Not a BASELINE config, but the
shape `run.py --config_file hparams_base.yaml` sees on real Peppa clips: non-square frames through every stage
(50x90 -> 25x45 -> 13x23 -> 7x12), an odd frame count (23 -> 12 -> 6 -> 3) and T = 316 > 256 in the attention.

Same measurement and the same SURVEY 8d tolerances as tests/test_parity_c2_gpu.py (fp32 CPU oracle, same weights and
clips): embeddings cosine >= 0.999 / max-abs <= 2e-2, loss |d| <= 5e-3, every residual block teacher-forced, trunk
activations and full-depth gradients against torch's own bf16 autocast of the oracle."""
import pytest

pytestmark = pytest.mark.gpu

REF_FRAMES, REF_HW, REF_SAMPLES = 23, (100, 180), 101429


@pytest.fixture(scope="module")
def rep():
    from parity_c2_report import report
    return report(batch=8, frames=REF_FRAMES, size=REF_HW, samples=REF_SAMPLES, log=lambda *a: print(*a, flush=True))


def test_reference_geometry_embeddings_and_loss(rep):
    assert rep["video_cos"] >= 0.999 and rep["video_maxabs"] <= 2e-2, (rep["video_cos"], rep["video_maxabs"])
    assert rep["audio_cos"] >= 0.999 and rep["audio_maxabs"] <= 2e-2, (rep["audio_cos"], rep["audio_maxabs"])
    assert rep["dloss"] <= 5e-3, (rep["loss"], rep["loss_ref"])


def test_reference_geometry_blocks_and_stages(rep):
    for stage, (ours, yard) in rep["stages"].items():
        assert ours <= 1.05 * yard + 1e-3, (stage, ours, yard)
    for name, fwd, dx, dw in rep["blocks"]:
        assert fwd <= 0.01 and dx <= 0.13 and dw <= 0.15, (name, fwd, dx, dw)


def test_reference_geometry_gradients(rep):
    from test_parity_c2_gpu import _assert_full_depth_gradients
    g = rep["grads"]
    assert g["audio"][0] <= 0.02 and g["audio"][1] <= 0.06, g["audio"]      # T = 316: the attention's long-sequence path
    assert g["video_encoder.project"][0] <= 0.06, g["video_encoder.project"]
    assert not rep.get("missing_grads") and not rep.get("extra_grads"), (rep.get("missing_grads"), rep.get("extra_grads"))
    _assert_full_depth_gradients(g, rep["gstats"])
