"""run.py's loop end to end on the GPU: train a few steps, validate (recall@10 + triplet accuracy on the device),
write Lightning-format checkpoints through the two ModelCheckpoint callbacks of run.py:32-55, reload the best one
(pig/evaluation.py:42-53) and check it reproduces the embeddings bit for bit."""
import copy
import os

import pytest
import torch


@pytest.mark.gpu
def test_fit_validate_checkpoint_reload(tmp_path):
    import pig.models
    import pig.evaluation
    from pig.execution import default_config
    from peppa_amd.checkpoint import ModelCheckpoint, load_checkpoint, callback_states
    from peppa_amd.trainer import SyntheticPigData, Trainer
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    cfg["data"]["train"]["batch_size"] = 26
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg).cuda()
    data = SyntheticPigData(cfg["data"], frames=8, size=64, samples=16000, steps_per_epoch=2, val_batches=4)
    root = str(tmp_path / "version_0")
    callbacks = [ModelCheckpoint(monitor=m, mode="max", save_last=True, save_top_k=1, filename="{epoch}-{" + m + ":.2f}")
                 for m in ("valnarr_rec_fixed", "valnarr_triplet")]
    trainer = Trainer(max_epochs=2, callbacks=callbacks, default_root_dir=root)
    trainer.fit(net, data)
    assert trainer.global_step == 4
    for name in ("val_loss", "valnarr_loss", "val_rec_fixed", "valnarr_rec_fixed", "val_triplet", "valnarr_triplet"):
        assert name in trainer.callback_metrics, name
    assert 0.0 <= float(trainer.callback_metrics["valnarr_rec_fixed"]) <= 1.0
    assert 0.0 <= float(trainer.callback_metrics["valnarr_triplet"]) <= 1.0
    files = sorted(os.listdir(os.path.join(root, "checkpoints")))
    assert "last.ckpt" in files and os.path.exists(os.path.join(root, "hparams.yaml"))
    assert any(f.startswith("epoch=") and "valnarr_rec_fixed=" in f for f in files)
    assert any(f.startswith("epoch=") and "valnarr_triplet=" in f for f in files)
    last = load_checkpoint(os.path.join(root, "checkpoints", "last.ckpt"))
    assert last["global_step"] == 4 and last["epoch"] == 1 and callback_states(last)
    opt_state = last["optimizer_states"][0]["state"]
    assert all(set(st) >= {"step", "next_m", "next_v"} for st in opt_state.values()) and len(opt_state) > 100

    best, path = pig.evaluation.load_best_model(root)
    stored = load_checkpoint(path)["state_dict"]
    for k, v in best.state_dict().items():
        assert torch.equal(v.cpu(), stored[k]), k
    best = best.cuda().eval()
    twin = pig.models.PeppaPig.load_from_checkpoint(path).cuda().eval()
    with torch.no_grad():
        a = best.encode_video(data.batch.video[:4])
        b = twin.encode_video(data.batch.video[:4])
        c = best.encode_audio(data.batch.audio[:4])
        d = twin.encode_audio(data.batch.audio[:4])
    assert torch.equal(a, b) and torch.isfinite(a).all() and torch.isfinite(c).all()
    assert torch.allclose(c, d, atol=1e-3)        # the audio tower's GroupNorm statistics use float atomics
    # saving must not disturb the live optimizer state (it is copied to the host, never moved there)
    for st in trainer.optimizer.state.values():
        assert st["next_m"].is_cuda and st["next_v"].is_cuda


@pytest.mark.gpu
def test_resume_continues_where_the_checkpoint_stopped(tmp_path):
    """Two epochs in one run == one epoch, checkpoint, resume for the second: same step counters, same optimizer
    schedule position, and the restored run starts from the stored weights bit for bit."""
    import pig.models
    from pig.execution import default_config
    from peppa_amd.checkpoint import ModelCheckpoint, load_checkpoint
    from peppa_amd.trainer import SyntheticPigData, Trainer
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    cfg["video"]["static"], cfg["video"]["version"] = False, "r2plus1d_18"
    cfg["data"]["train"]["batch_size"] = 26
    data = SyntheticPigData(cfg["data"], frames=4, size=32, samples=8000, steps_per_epoch=2, val_batches=4)
    root = str(tmp_path / "run")
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg).cuda()
    cb = ModelCheckpoint(monitor="valnarr_triplet", mode="max", save_last=True)
    Trainer(max_epochs=1, callbacks=[cb], default_root_dir=root).fit(net, data)
    last = os.path.join(root, "checkpoints", "last.ckpt")
    stored = load_checkpoint(last)
    assert stored["epoch"] == 0 and stored["global_step"] == 2

    torch.manual_seed(1)                                  # different init: everything must come from the file
    net2 = pig.models.PeppaPig(cfg).cuda()
    cb2 = ModelCheckpoint(monitor="valnarr_triplet", mode="max", save_last=True)
    trainer = Trainer(max_epochs=2, callbacks=[cb2], default_root_dir=root, resume_from_checkpoint=last)
    optim = net2.configure_optimizers()
    assert trainer._restore(net2, optim, last) == 1 and trainer.global_step == 2
    for k, v in net2.state_dict().items():
        assert torch.equal(v.cpu(), stored["state_dict"][k]), k
    for st in optim.state.values():
        assert st["step"] == 2 and st["next_m"].is_cuda
    assert float(cb2.best_model_score) == pytest.approx(float(cb.best_model_score))
    trainer.fit(net2, data)
    assert trainer.global_step == 4 and trainer.current_epoch == 1
    assert load_checkpoint(last)["global_step"] == 4


@pytest.mark.gpu
def test_a_step_limit_inside_an_epoch_still_writes_last_ckpt_and_a_finished_run_does_not_step_again(tmp_path, caplog):
    """`--max_steps N` shorter than an epoch wrote no checkpoint at all, and resuming a checkpoint that had
    reached max_steps took one more optimizer step.  Also: `precision: 16` announces that it means bf16 here."""
    import logging
    import pig.models
    from pig.execution import default_config
    from peppa_amd.checkpoint import ModelCheckpoint, load_checkpoint
    from peppa_amd.trainer import SyntheticPigData, Trainer
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    cfg["data"]["train"]["batch_size"] = 26
    data = SyntheticPigData(cfg["data"], frames=4, size=32, samples=8000, steps_per_epoch=5, val_batches=4)
    root = str(tmp_path / "run")
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg).cuda()
    cb = ModelCheckpoint(monitor="valnarr_triplet", mode="max", save_last=True)
    with caplog.at_level(logging.WARNING, logger="peppa_amd.trainer"):
        tr = Trainer(max_epochs=3, max_steps=2, callbacks=[cb], default_root_dir=root, precision=16)
        tr.fit(net, data)
    assert tr.global_step == 2 and tr.current_epoch == 0
    assert any("precision 16 -> bf16" in r.getMessage() for r in caplog.records)
    last = os.path.join(root, "checkpoints", "last.ckpt")
    assert load_checkpoint(last)["global_step"] == 2
    before = {k: v.clone() for k, v in net.state_dict().items()}
    tr2 = Trainer(max_epochs=3, max_steps=2, callbacks=[ModelCheckpoint(monitor="valnarr_triplet", mode="max")],
                  default_root_dir=root, resume_from_checkpoint=last)
    tr2.fit(net, data)
    assert tr2.global_step == 2                       # nothing left to do
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k


@pytest.mark.gpu
def test_weight_operands_are_cached_between_optimizer_steps_and_never_stale():
    """layers.cached_operands: with gradient accumulation (hparams_base.yaml:42 accumulates 8 micro-batches) and in
    validation the 16-bit operand layouts are built once per optimizer step, not once per forward pass; they must be
    rebuilt after BertAdam.step() (raw-pointer update) and after load_state_dict (torch update)."""
    import copy
    import pig.models
    from pig.execution import default_config
    from peppa_amd import layers as L
    from peppa_amd.data import synthetic_batch
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg).cuda().train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "layer_drop"):
            m.layer_drop = 0.0
    conv = net.video_encoder.video.layer1[0].conv1[0][0]
    geom = L.ConvGeom(2, (4, 32, 32), conv.in_channels, conv.out_channels, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    b = synthetic_batch(2, 4, 32, 4000).to("cuda")
    opt = net.configure_optimizers()
    try:
        L.CACHE_OPERANDS = True
        L.weights_changed()
        fresh = lambda: L._prep_conv_weights(conv.weight, geom, True)
        w1 = L.prep_conv_weights(conv.weight, geom)
        assert L.prep_conv_weights(conv.weight, geom)[0] is w1[0]                 # second request: the same tensors
        for step in range(3):        # step 0 leaves the weights alone (warm-up multiplier 0), steps 1-2 move them
            opt.zero_grad(set_to_none=True)
            for micro in range(2):   # two micro-batches per optimizer step: operands built once
                n_before = len(L._OPERANDS)
                (net.training_step(b, micro) / 2).backward()
                if micro == 1:
                    assert len(L._OPERANDS) == n_before
            opt.step()
            w2 = L.prep_conv_weights(conv.weight, geom)
            assert w2[0] is not w1[0]
            assert torch.equal(w2[0], fresh()[0]) and torch.equal(w2[1], fresh()[1])     # rebuilt from the moved masters
            w1 = w2
        sd = {k: v * 1.5 if v.is_floating_point() else v for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        w3 = L.prep_conv_weights(conv.weight, geom)
        assert w3[0] is not w1[0] and torch.equal(w3[0], fresh()[0])
    finally:
        L.CACHE_OPERANDS = False
        L.weights_changed()


@pytest.mark.gpu
@pytest.mark.parametrize("flags,ok", [(["--random_init"], True), (["--random_init", "--precision", "fp16"], True), ([], False)])
def test_run_py_cli(tmp_path, flags, ok):
    """The reference's entry point (run.py:64-71) with the shipped yaml: `pretrained: true` cannot be honoured offline and
    must stop the run with the reason (not silently train from random init); `--random_init` opts in; `--precision fp16`
    trains on the half library with the loss scaler."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "run.py"), "--config_file", os.path.join(root, "hparams_base.yaml"),
           "--limit_train_batches", "3", "--frames", "4", "--size", "32", "--samples", "4000", "--max_epochs", "1",
           "--accumulate_grad_batches", "2", "--default_root_dir", str(tmp_path / "run")] + flags
    env = dict(os.environ, PYTHONPATH=root)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    if ok:
        assert res.returncode == 0, res.stderr[-2000:]
        assert os.path.exists(tmp_path / "run" / "hparams.yaml")
        import yaml
        saved = yaml.safe_load(open(tmp_path / "run" / "hparams.yaml"))
        assert saved["video"]["pretrained"] is False and saved["audio"]["pretrained"] is False   # the EFFECTIVE setting
        assert os.path.exists(tmp_path / "run" / "checkpoints" / "last.ckpt")
    else:
        assert res.returncode != 0 and "Kinetics" in res.stderr and "--random_init" in res.stderr
