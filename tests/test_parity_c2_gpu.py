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
    ~100 % off fp32 in relative L2 (stem 1.04, layer4 0.83) -- a figure an all-zero gradient also reaches, so the
    assertion is on quantities a wrong gradient cannot meet: per stage the NORM ratio against the fp32 oracle within
    0.9-1.1 and the COSINE not below the yardstick's (0.46-0.53) minus 0.05; the well-conditioned parts (audio tower,
    projection) are asserted tightly;
  * the same for a CONDITIONED model (300 optimizer steps on structured clips towards nearly orthogonal targets, state
    loaded into the oracle): the full-depth error leaves the ~1.0 regime (0.02-0.20) and is asserted against torch's own bf16
    run (<= 1.15 x + 0.02), by norm ratio 0.9-1.1 and cosine, and with absolute bounds 1.4x above the largest of four measured
    runs (the training is not reproducible);
  * configs[1] at its true batch 64 (forward + loss) and configs[2] (frozen wav2vec2) at the real geometry.
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


def _assert_full_depth_gradients(g, stats, trunk_rel_bound=None, stem_ratio=(0.9, 1.1)):
    """A stage's gradient (all its tensors pooled) against the fp32 oracle: the NORM ratio within 0.9-1.1 and the COSINE at
    least the torch-bf16 yardstick's minus 0.05 -- neither a zero, a mis-scaled, a sign-flipped nor a mis-routed gradient
    can satisfy both (an all-zero gradient has relative L2 error 1.0, which the round-2 form of this test accepted)."""
    for key, st in stats.items():
        print(f"  {key:34s} |g|/|ref| {st['ratio']:.4f} cosine {st['cos']:.4f}   (torch bf16: {st['ratio16']:.4f} {st['cos16']:.4f})")
    for key, st in stats.items():
        trunk = key.startswith("video_encoder.video.")
        lo, hi = stem_ratio if key.endswith(".stem") else (0.9, 1.1)
        assert lo <= st["ratio"] <= hi, (key, st)
        yard = st["cos16"]
        floor = yard - 0.05 if yard == yard and yard > 0 else (0.55 if trunk else 0.97)    # (layer4: the yardstick run keeps no fp32 copy)
        assert st["cos"] >= floor, (key, st, floor)
        if not trunk:
            assert st["cos"] >= 0.97, (key, st)
        if trunk and trunk_rel_bound is not None:
            assert g[key][0] <= trunk_rel_bound[key.split(".")[-1]], (key, g[key])
            if g[key][2] == g[key][2]:
                assert g[key][0] <= 1.15 * g[key][2] + 0.02, (key, g[key])      # never worse than torch's own bf16 run


def test_gradients_under_a_smooth_objective(rep):
    g = rep["grads"]
    assert g["audio"][0] <= 0.02 and g["audio"][1] <= 0.06, g["audio"]            # LayerNorm tower: well conditioned
    assert g["video_encoder.project"][0] <= 0.06, g["video_encoder.project"]
    assert not rep.get("missing_grads") and not rep.get("extra_grads"), (rep.get("missing_grads"), rep.get("extra_grads"))
    # random-init train-mode-BatchNorm trunk: chaotic (module docstring) -- measured r03: |g|/|ref| 0.998-1.018, cosine
    # 0.465-0.656 against the yardstick's 0.461-0.525
    _assert_full_depth_gradients(g, rep["gstats"])


def test_full_depth_gradients_of_a_conditioned_model():
    """the HIP model is trained 300 optimizer steps on structured clips, its state is loaded into the
    oracle and the full-depth comparison is repeated.

    Round 4: the clips' targets are nearly ORTHOGONAL (common_weight 0.3: pairwise cosine 0.08; rounds 2-3 used 0.5).  The
    video tower then has to separate the clips (it reaches cosine 1.00 to every target) and ends in a state whose backward
    pass is well conditioned in bf16: over four runs (tools/probe/cond_spread.py, profiles/r04_probe_cond_spread.log) the
    trunk's relative L2 error against the fp32 oracle is stem 0.14-0.20, layer1 0.18-0.21, layer2 0.15-0.18, layer3
    0.07-0.09, layer4 0.022 -- within 0.01 of the torch-bf16 yardstick's every time -- with norm ratios 0.97-1.01 and cosines
    0.977-0.9998.  With the old targets the stem sat at the edge of its bounds (ratio 0.87 ... 1.24, cosine 0.84-0.87, error
    0.49-0.75), which had them widened twice after failing runs; they are back to 0.9-1.1 and
    yardstick - 0.05, the absolute bounds are 1.4x the largest of the four measurements.  (The training is not bitwise
    reproducible -- fp32 atomics -- so every run ends in a slightly different state.)

    The TRIPLET LOSS's own gradient ("has never actually run"): the audio tower does not separate the clips
    in 300 steps at any learning rate (tools/probe/cond_dbg.py: its embeddings stay on one point a, cos(a, target_i) =
    0.23 ... 0.36), so with the configured margin 0.2 every hinge is active and the loss is 2 m (N-1)/N exactly -- one more smooth
    objective.  With margin 0.03 the column hinges m + (V_i - V_j).a of the pairs more than 0.03 apart switch off (about 15 %
    of all hinges), and the comparison runs on a partly active hinge pattern: measured at margin 0.05 (90 % active) loss 0.093868
    vs the oracle's 0.093882, trunk error 0.23 / 0.25 / 0.21 / 0.10 / 0.023 against the yardstick's 0.21 / 0.25 / 0.21 / 0.10, audio
    0.0055 (profiles/r04_probe_cond_spread.log)."""
    from parity_c2_report import conditioned_report
    out = conditioned_report(steps=300, common_weight=0.3, margin=0.03)
    assert out["target_cosine"][-1] >= 0.6, out["target_cosine"]        # it did train
    rep = out["smooth"]
    assert rep["video_cos"] >= 0.9995 and rep["audio_cos"] >= 0.9995 and rep["dloss"] <= 1e-3
    for stage, (ours, yard) in rep["stages"].items():
        assert ours <= 1.05 * yard + 1e-3 and ours <= 0.03, (stage, ours, yard)
    bound = {"stem": 0.28, "layer1": 0.30, "layer2": 0.26, "layer3": 0.13, "layer4": 0.035}
    _assert_full_depth_gradients(rep["grads"], rep["gstats"], trunk_rel_bound=bound)
    assert rep["grads"]["audio"][0] <= 0.01 and rep["grads"]["video_encoder.project"][0] <= 0.02
    assert "hinge" in out and 0.05 < out["hinge_active"] < 0.95, out["hinge_active"]   # neither all off nor all on: it RAN
    hinge = out["hinge"]
    assert hinge["dloss"] <= 1e-3
    _assert_full_depth_gradients(hinge["grads"], hinge["gstats"], trunk_rel_bound={"stem": 0.35, "layer1": 0.36, "layer2": 0.30, "layer3": 0.15, "layer4": 0.04})
    assert hinge["grads"]["audio"][0] <= 0.02, hinge["grads"]["audio"]


def test_configs1_at_its_true_batch_64():
    """BASELINE configs[1] at batch 64 (not the batch 8 the other tests run): both towers + loss against the fp32 oracle
    (forward only; the oracle takes ~17 s).  Measured r03: video min cosine 0.999017 / max-abs 7.7e-3, audio 0.999980 /
    8.8e-4, loss |d| 2.4e-5.  (torch's own bf16 autocast of the oracle sits at 0.9992 at batch 8: the headroom to 0.999 is
    a property of bf16 operands)"""
    from parity_c2_report import forward_b64
    r = forward_b64()
    assert r["video_cos"] >= 0.999 and r["video_maxabs"] <= 2e-2, r
    assert r["audio_cos"] >= 0.999 and r["audio_maxabs"] <= 2e-2, r
    assert r["dloss"] <= 5e-3, r


def test_configs2_frozen_wav2vec_at_real_geometry():
    """BASELINE configs[2] (hparams_freeze_wav2vec.yaml) at 16x112x112 + 36 800 samples, batch 8: the trainable audio
    parameters sit before the 12 frozen transformer layers (pig/models.py:75-81), so their gradients cross all of them by
    data gradients alone, T = 114 attention backward included.  Measured r03: audio pooled relative L2 0.6 %, worst tensor
    1.4 %; exactly the oracle's set of tensors has a gradient."""
    from parity_c2_report import frozen_report
    r = frozen_report()
    assert sorted(r["frozen_names"]) == sorted(r["hip_frozen_names"]) and len(r["frozen_names"]) > 190
    assert not r.get("missing_grads") and not r.get("extra_grads"), (r.get("missing_grads"), r.get("extra_grads"))
    assert r["grads"]["audio"][0] <= 0.02 and r["grads"]["audio"][1] <= 0.05, r["grads"]["audio"]
    st = r["gstats"]["audio"]
    assert 0.98 <= st["ratio"] <= 1.02 and st["cos"] >= 0.999, st
    assert r["audio_cos"] >= 0.999 and r["video_cos"] >= 0.999 and r["dloss"] <= 5e-3


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
