"""The subset of `pytorch_lightning.Trainer` that the reference's run.py uses (run.py:56-62): fit(),
accumulate_grad_batches, limit_train_batches / limit_val_batches, max_steps, max_time, precision, the
ModelCheckpoint callbacks, resume_from_checkpoint, per-step logging.  run.py ALWAYS uses this loop
(Lightning is not installed here and the HIP towers own their streams / loss scaling); `pl.Trainer` flags
outside this subset are reported by run.py and dropped.  One process per GPU; under torch.distributed the
gradients go through peppa_amd.dist.GradBuckets (RCCL all-reduce, SUM)."""
import time
import logging
import torch

from .dist import default_buckets, is_dist

log = logging.getLogger(__name__)


class SyntheticPigData:
    """Synthetic ClipBatch stream with the shapes of the configured clips (the Peppa dataset and
    moviepy are not available offline)."""

    def __init__(self, config, frames=16, size=112, samples=36800, steps_per_epoch=100, device="cuda", val_batches=2):
        from .data import synthetic_batch
        self.batch = synthetic_batch(config["train"]["batch_size"], frames, size, samples).to(device)
        self.steps_per_epoch, self.val_batches = steps_per_epoch, val_batches

    def train_dataloader(self):
        for _ in range(self.steps_per_epoch):
            yield self.batch

    def val_dataloader(self, batches=None):
        """The reference's four validation loaders (pig/data.py PigData.val_dataloader: main, narration, and the
        two duration-matched triplet sets), as `batches` synthetic batches each; the triplet sets carry
        durations drawn from three values so that `score_triplets` finds same-duration pairs."""
        b = self.batch
        batches = self.val_batches if batches is None else batches
        n = b.video.shape[0]
        dur = torch.tensor([2.0, 2.5, 3.0])[torch.arange(n) % 3]
        trip = type(b)(b.video, b.audio, dur, dur)
        return [[b] * batches, [b] * batches, [trip] * batches, [trip] * batches]


class Trainer:
    def __init__(self, accumulate_grad_batches=1, limit_train_batches=None, max_steps=None, max_time_s=None,
                 log_every=10, max_epochs=1, limit_val_batches=None, callbacks=(), default_root_dir=None,
                 resume_from_checkpoint=None, precision=None, **ignored):
        # precision: None / 16 / "bf16" -> the model's own setting (bf16 unless `mi355x: {dtype: fp16}`);
        # "fp16" -> IEEE half like the reference's AMP runs, with dynamic loss scaling (peppa_amd.amp.GradScaler)
        self.precision = None if precision in (None, 16, "16") else str(precision)
        self._precision_arg = precision
        self.scaler = None
        self.accumulate = max(1, int(accumulate_grad_batches))
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.max_steps, self.max_time_s, self.log_every = max_steps, max_time_s, log_every
        self.max_epochs, self.callbacks = max_epochs, list(callbacks)
        self.global_step = self.current_epoch = 0
        self.callback_metrics = {}
        self.resume_from_checkpoint = resume_from_checkpoint      # Lightning 1.4 Trainer argument of the same name
        self.default_root_dir = default_root_dir
        if default_root_dir is not None:      # Lightning: {root}/lightning_logs/version_N/checkpoints
            for cb in self.callbacks:
                if getattr(cb, "dirpath", None) is None:
                    cb.dirpath = f"{default_root_dir}/checkpoints"

    def validate(self, net, data):
        """One pass over the validation loaders -> `validation_epoch_end` -> logged metrics (pig/models.py:266-318)."""
        was_training = net.training
        net.eval()
        outputs = []
        with torch.no_grad():
            for idx, loader in enumerate(data.val_dataloader()):
                outs = []
                for i, batch in enumerate(loader):
                    if self.limit_val_batches is not None and i >= self.limit_val_batches:
                        break
                    outs.append(net.validation_step(batch, i, dataloader_idx=idx))
                outputs.append(outs)
            net.validation_epoch_end(outputs)
        net.train(was_training)
        self.callback_metrics = dict(getattr(net, "_logged", {}))
        return self.callback_metrics

    def fit(self, net, data):
        from . import layers as L
        prev_cache, L.CACHE_OPERANDS = L.CACHE_OPERANDS, True    # 16-bit weight operands live until optimizer.step()
        try:
            return self._fit(net, data)
        finally:
            L.CACHE_OPERANDS = prev_cache
            L.weights_changed()

    def _fit(self, net, data):
        optim = self.optimizer = net.configure_optimizers()
        if self.precision is not None and hasattr(net, "set_precision"):
            net.set_precision(self.precision)
        if getattr(net, "precision", "bf16") == "fp16":
            from .amp import GradScaler
            self.scaler = GradScaler()
        elif self._precision_arg in (16, "16"):
            # the reference's `precision: 16` (hparams_base.yaml:45) is fp16 native AMP + GradScaler; here 16 keeps the
            # model's own 16-bit type.  Say so once, loudly: nobody should find out from a diverging loss curve.
            log.warning("precision 16 -> %s operands, no loss scaler; pass precision='fp16' (run.py --precision fp16, or "
                        "`mi355x: {dtype: fp16}`) for the reference's fp16 AMP semantics", getattr(net, "precision", "bf16"))
        buckets = None
        if is_dist():
            buckets = default_buckets(net, next(net.parameters()).device)
        net.train()
        if self.default_root_dir is not None and (not is_dist() or torch.distributed.get_rank() == 0):
            import os
            import yaml
            os.makedirs(self.default_root_dir, exist_ok=True)     # Lightning's logger: hparams.yaml beside checkpoints/
            with open(os.path.join(self.default_root_dir, "hparams.yaml"), "w") as f:
                yaml.safe_dump(dict(net.config), f)
        first_epoch = 0
        if self.resume_from_checkpoint is not None:
            first_epoch = self._restore(net, optim, self.resume_from_checkpoint)
        t0 = time.time()
        optim.zero_grad(set_to_none=True)
        for epoch in range(first_epoch, self.max_epochs):
            if self.max_steps is not None and self.global_step >= self.max_steps:
                break       # e.g. resumed from a checkpoint that had already reached max_steps: not one step more
            self.current_epoch = epoch
            limit_hit = self._train_epoch(net, data, optim, buckets, t0)
            if self.scaler is not None:
                self.scaler.flush()     # a device-skipped last step must not stay counted in BertAdam's `step`
            # also when a step / time limit ended the epoch early: validate and let the callbacks write (at least)
            # last.ckpt, so a `--max_steps N` run shorter than an epoch still leaves a checkpoint behind
            if hasattr(data, "val_dataloader") and self.callbacks:
                metrics = self.validate(net, data)
                rank0 = not is_dist() or torch.distributed.get_rank() == 0
                for cb in self.callbacks:
                    if rank0:
                        cb.on_validation_end(net, optim, epoch, self.global_step, metrics, scaler=self.scaler)
            if limit_hit:
                break
        if self.scaler is not None:
            self.scaler.flush()
        return net

    def _restore(self, net, optim, path):
        """Weights, optimizer state (per-parameter step / next_m / next_v), counters and the callbacks' best score
        from a checkpoint written after an epoch's validation; training continues with the next epoch."""
        from .checkpoint import callback_states, load_checkpoint
        cp = load_checkpoint(path)
        net.load_state_dict(cp["state_dict"])
        if cp.get("optimizer_states"):
            optim.load_state_dict(cp["optimizer_states"][0])     # torch moves the state to each parameter's device
        self.global_step = int(cp.get("global_step", 0))
        if self.scaler is not None and cp.get("native_amp_scaling_state"):
            self.scaler.load_state_dict(cp["native_amp_scaling_state"])
        elif cp.get("native_amp_scaling_state"):
            log.warning("%s carries an fp16 loss-scaler state (native_amp_scaling_state) but this run has no scaler "
                        "(precision %s): the state is dropped; resume with precision='fp16' to keep it", path,
                        getattr(net, "precision", "bf16"))
        for state in callback_states(cp):
            for cb in self.callbacks:
                if getattr(cb, "monitor", None) == state.get("monitor"):
                    cb.load_state(state)
        log.info("resumed from %s (epoch %d, step %d)", path, cp.get("epoch", 0), self.global_step)
        return int(cp.get("epoch", -1)) + 1

    def _train_epoch(self, net, data, optim, buckets, t0):
        """True when a step / time limit ended training.  The optimizer steps every `accumulate` micro-batches and, like
        Lightning, on the last batch of the epoch (leftover micro-batches are applied, not carried into the next epoch).
        Under data parallelism only the micro-batch that completes an optimizer step is all-reduced."""
        def limited(loader):
            for i, batch in enumerate(loader):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    return
                yield batch

        it = iter(limited(data.train_dataloader()))
        nxt = next(it, None)
        i = 0
        while nxt is not None:
            batch, nxt = nxt, next(it, None)
            stepping = (i + 1) % self.accumulate == 0 or nxt is None
            if buckets is not None:
                buckets.sync = stepping
            loss = net.training_step(batch, i)
            if self.scaler is not None:
                self.scaler.scale(loss / self.accumulate).backward()
            else:
                (loss / self.accumulate).backward()
            i += 1
            if stepping:
                if buckets is not None:
                    buckets.finish()
                if self.scaler is not None:     # unscale (after the all-reduce: every rank sees the same overflow), step or skip
                    self.scaler.step(optim)
                    self.scaler.update()
                else:
                    optim.step()
                optim.zero_grad(set_to_none=True)
                self.global_step += 1
                if self.global_step % self.log_every == 0:
                    log.info("step %d loss %.5f (%.1f s)", self.global_step, float(loss.detach()), time.time() - t0)
                if self.max_steps is not None and self.global_step >= self.max_steps:
                    return True
                if self.max_time_s is not None and time.time() - t0 > self.max_time_s:
                    return True
        return False
