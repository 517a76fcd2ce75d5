"""Parity at the REAL geometry of BASELINE configs[1] (16x112x112 video + 36 800 audio samples) with SURVEY 8d's
tolerances, against the fp32 CPU oracle at batch 8 (the oracle's forward + backward take ~6 s on the box's 16 cores).

Why this shape: tests/test_model_gpu.py runs a 4x4x32x32 toy whose layer 4 normalises over 16 samples -- that measures
BatchNorm ill-conditioning, not the kernels.  Here layer 4 sees 784 rows per channel and the kernel variants are the ones
bench.py runs (window / temporal-window / sliding-window kernels).

What bf16 activations cost, measured here with torch's own bf16 autocast of the ORACLE as the yardstick:
  * forward: the free-running activation error grows ~2.5x per stage (0.5 % after the stem, 25 % after layer 4, the same
    for torch-bf16) yet the pooled, projected, normalised EMBEDDING keeps cosine >= 0.999 / max-abs <= 2e-2 and the
    loss |d| <= 5e-3 -- asserted;
  * backward, one block teacher-forced: dx ~10 %, dW ~11 % (ReLU masks of near-zero activations flip for ~0.5-1 % of
    the elements once the activation is rounded to bf16: relative L2 ~ sqrt(fraction)); asserted <= 13 / 15 %;
  * backward, full depth: a random-init train-mode-BatchNorm trunk is chaotic, torch-bf16's own trunk gradients are
    ~100 % off fp32 (stem 1.04, layer4 0.83); the HIP path must not be worse than that yardstick, and the
    well-conditioned parts (audio tower, projection) are asserted tightly.
"""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rep():
    from parity_c2_report import report
    lines = []
    res = report(log=lambda *a: (lines.append(" ".join(str(x) for x in a)), print(*a, flush=True)))
    res["log"] = lines
    return res


def test_embeddings_and_loss_meet_survey_8d_at_real_shape(rep):
    assert rep["video_cos"] >= 0.999 and rep["video_maxabs"] <= 2e-2, (rep["video_cos"], rep["video_maxabs"])
    assert rep["audio_cos"] >= 0.999 and rep["audio_maxabs"] <= 2e-2, (rep["audio_cos"], rep["audio_maxabs"])
    assert rep["dloss"] <= 5e-3, (rep["loss"], rep["loss_ref"])


def test_trunk_activations_track_the_bf16_yardstick_stage_by_stage(rep):
    for stage, (ours, yard) in rep["stages"].items():
        assert ours <= 1.05 * yard + 1e-3, (stage, ours, yard)
    assert rep["stages"]["stem"][0] <= 0.01


def test_every_block_teacher_forced_at_real_shape(rep):
    for name, fwd, dx, dw in rep["blocks"]:
        assert fwd <= 0.01 and dx <= 0.13 and dw <= 0.15, (name, fwd, dx, dw)


def test_gradients_under_a_smooth_objective(rep):
    g = rep["grads"]
    assert g["audio"][0] <= 0.02 and g["audio"][1] <= 0.06, g["audio"]            # LayerNorm tower: well conditioned
    assert g["video_encoder.project"][0] <= 0.06, g["video_encoder.project"]
    for key, (ours, worst, yard) in g.items():
        if key.startswith("video_encoder.video.") or key == "video_encoder.videopool":
            # chaotic regime (see the module docstring): never worse than torch's own bf16 run of the oracle
            if yard == yard and yard != float("inf"):
                assert ours <= 1.1 * yard + 0.02, (key, ours, yard)
            assert ours <= 1.3, (key, ours)


def test_triplet_accuracy_within_0p2_percent_on_10k_triplets():
    """"within +-0.2 %" = fraction of >= 10 000 duration-matched triplets whose decision flips between the
    oracle's and the HIP path's embeddings (128 structured synthetic clips at the configs[1] geometry, pairing as
    pig/triplet.py:99-121, 160 resamplings).  Measured (gpurun_out r02): random init 8.7 % flips at accuracy 0.50 --
    every clip embeds almost identically, each decision is a coin flip settled by rounding, reported only; heads fitted
    40 steps (accuracy 0.999): 0.17 %; 400 steps: 0.00 %.  Asserted for the fitted (trained-like) models."""
    from parity_c2_report import triplet_flips
    out = triplet_flips(fit_steps=(0, 10, 40, 400))
    assert all(n >= 10000 for _, _, _, n in out.values())
    acc_o, acc_h, flips, _ = out[400]
    assert flips <= 0.002 and abs(acc_o - acc_h) <= 0.002, out
    acc_o, acc_h, flips, _ = out[40]
    assert flips <= 0.005 and abs(acc_o - acc_h) <= 0.002, out
    for steps in (0, 10):
        acc_o, acc_h, flips, _ = out[steps]
        assert abs(acc_o - acc_h) <= 0.01, out


@pytest.mark.parametrize("version", ["r3d_18", "mc3_18"])
def test_other_backbones_model_level(version):
    """`video.version: r3d_18 | mc3_18` (pig/models.py:122-129), selectable from the yaml: the whole model on the GPU
    against the oracle -- BasicStem (3,7,7), 3x3x3 residual blocks (mc3: (1,3,3) from layer 2 on, (1,s,s) downsamples).
    8 frames of 64x64 at batch 4 (layer 4 normalises over 4*1*4*4 = 64 rows: a small, noisier shape than configs[1]; the
    yardstick is torch's bf16 autocast of the oracle on the same input)."""
    from parity_c2_report import report
    rep = report(batch=4, frames=8, size=64, samples=16000, version=version, blocks=True)
    for stage, (ours, yard) in rep["stages"].items():
        assert ours <= 1.1 * yard + 2e-3, (stage, ours, yard)
    for name, fwd, dx, dw in rep["blocks"]:
        assert fwd <= 0.02 and dx <= 0.16 and dw <= 0.18, (name, fwd, dx, dw)
    assert rep["audio_cos"] >= 0.999 and rep["audio_maxabs"] <= 2e-2
    assert 1 - rep["video_cos"] <= 1.5 * (1 - rep["video_cos_bf16"]) + 1e-3, (rep["video_cos"], rep["video_cos_bf16"])
    assert rep["grads"]["audio"][0] <= 0.02


def test_long_clips_config5_shape_bf16():
    """BASELINE configs[4] geometry (32 frames of 112x112, 4.6 s = 73 600 samples -> 229 wav2vec2 frames) in bf16 at
    batch 4: the attention core beyond the 128-frame fused kernel, temporal convolutions over 32 frames."""
    from parity_c2_report import report
    rep = report(batch=4, frames=32, size=112, samples=73600, blocks=False)
    assert rep["video_cos"] >= 0.999 and rep["video_maxabs"] <= 2e-2, (rep["video_cos"], rep["video_maxabs"])
    assert rep["audio_cos"] >= 0.999 and rep["audio_maxabs"] <= 2e-2, (rep["audio_cos"], rep["audio_maxabs"])
    assert rep["dloss"] <= 5e-3
    assert rep["grads"]["audio"][0] <= 0.02 and rep["grads"]["video_encoder.project"][0] <= 0.06
    for stage, (ours, yard) in rep["stages"].items():
        assert ours <= 1.05 * yard + 1e-3, (stage, ours, yard)
