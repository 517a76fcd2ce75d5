"""Minimal stand-in for the subset of `pytorch_lightning.Trainer` that run.py uses
(run.py:56-62 of the reference): fit(), accumulate_grad_batches, limit_train_batches, max_time,
per-step logging.  Used when Lightning is not installed; with Lightning present run.py hands the
same module to the real Trainer.  One process per GPU; under torch.distributed the gradients go
through peppa_amd.dist.GradBuckets (RCCL all-reduce, SUM)."""
import time
import logging
import torch

from .dist import default_buckets, is_dist

log = logging.getLogger(__name__)


class SyntheticPigData:
    """Synthetic ClipBatch stream with the shapes of the configured clips (the Peppa dataset and
    moviepy are not available offline)."""

    def __init__(self, config, frames=16, size=112, samples=36800, steps_per_epoch=100, device="cuda"):
        from .data import synthetic_batch
        self.batch = synthetic_batch(config["train"]["batch_size"], frames, size, samples).to(device)
        self.steps_per_epoch = steps_per_epoch

    def train_dataloader(self):
        for _ in range(self.steps_per_epoch):
            yield self.batch


class Trainer:
    def __init__(self, accumulate_grad_batches=1, limit_train_batches=None, max_steps=None, max_time_s=None,
                 log_every=10, **ignored):
        self.accumulate = max(1, int(accumulate_grad_batches))
        self.limit_train_batches = limit_train_batches
        self.max_steps, self.max_time_s, self.log_every = max_steps, max_time_s, log_every
        self.global_step = 0

    def fit(self, net, data):
        optim = net.configure_optimizers()
        buckets = None
        if is_dist():
            buckets = default_buckets(net, next(net.parameters()).device)
        net.train()
        t0 = time.time()
        optim.zero_grad(set_to_none=True)
        for i, batch in enumerate(data.train_dataloader()):
            if self.limit_train_batches is not None and i >= self.limit_train_batches:
                break
            loss = net.training_step(batch, i)
            (loss / self.accumulate).backward()
            if (i + 1) % self.accumulate == 0:
                if buckets is not None:
                    buckets.finish()
                optim.step()
                optim.zero_grad(set_to_none=True)
                self.global_step += 1
                if self.global_step % self.log_every == 0:
                    log.info("step %d loss %.5f (%.1f s)", self.global_step, float(loss), time.time() - t0)
            if self.max_steps is not None and self.global_step >= self.max_steps:
                break
            if self.max_time_s is not None and time.time() - t0 > self.max_time_s:
                break
        return net
