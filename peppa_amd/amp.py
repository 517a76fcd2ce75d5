"""Dynamic loss scaling for the fp16 build of the HIP path.

The reference trains with `precision: 16` (hparams_base.yaml:45): Lightning's native AMP wraps every step in
`torch.cuda.amp.GradScaler` -- scale the loss, unscale the gradients before `BertAdam.step` (its per-tensor
`clip_grad_norm_`, pig/optimization.py:136-137, must see true gradients), skip the step when a gradient overflowed,
and grow / back off the scale.  `GradScaler` below keeps that API and those semantics
(init 65536, growth 2 every 2000 clean steps, backoff 0.5) with the unscale + non-finite check and the scale update in
libpeppa_hip (`pp_grad_unscale_check`, `pp_amp_update_scale`).  torch's `step()` reads the found-inf flag on the host, a
full device sync per optimizer step (measured here: +2 ms on a 49-ms step).  With `BertAdam` (which takes `skip_flag`) the
decision stays on the device -- the fused launch is a no-op when the flag is set -- and the flag travels to the host
asynchronously; the only host-side consequence of a skipped step, the per-tensor step counters of the warm-up schedule, is
corrected when the NEXT step begins (the flag of step k is waited for at step k + 1, when it has long arrived).  Other
optimizers get torch's behaviour (host read, step skipped).  bf16 runs need no scaler.
"""
import numpy as np
import torch

from . import hip as H
from .hip import f32

_CHUNK = 65536


class GradScaler:
    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        if growth_factor <= 1.0 or not 0.0 < backoff_factor < 1.0 or growth_interval < 1:
            raise ValueError("GradScaler: growth_factor > 1, 0 < backoff_factor < 1, growth_interval >= 1")
        self._enabled = enabled
        self._init_scale, self._growth, self._backoff, self._interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self._scale = self._tracker = self._found = None
        self._init_tracker = 0
        self._unscaled = False
        self._tables = {}
        self._skipped = 0
        self._pending = None      # (event, pinned flag copy, parameters stepped) of the previous BertAdam step

    def is_enabled(self):
        return self._enabled

    @property
    def skipped_steps(self):
        self._reconcile()
        return self._skipped

    def flush(self):
        """Settle the bookkeeping of the last optimizer step NOW (one event wait): after it `optimizer.state[p]['step']`
        is what the reference's skipped-step semantics give, also when the very last step of a run overflowed.  The
        trainer calls it when fit() ends or stops on a limit; checkpoint.optimizer_state calls it for a known scaler."""
        self._reconcile()

    def _reconcile(self, optimizer=None):
        """Wait for the previous step's found-inf flag (recorded long ago) and, if that step was skipped on the device,
        take back the step-counter increment BertAdam made for it."""
        if self._pending is None:
            return
        event, flag, opt, params = self._pending
        self._pending = None
        event.synchronize()
        if flag.item() != 0.0:
            self._skipped += 1
            for p in params:
                opt.state[p]['step'] -= 1

    def _lazy_init(self, device):
        if self._scale is None:
            self._scale = torch.full((1,), self._init_scale, dtype=f32, device=device)
            self._tracker = torch.full((1,), self._init_tracker, dtype=torch.int32, device=device)
            self._found = torch.zeros(1, dtype=f32, device=device)

    def scale(self, loss):
        """loss * scale (the backward pass then carries the scale into every gradient)."""
        if not self._enabled:
            return loss
        self._lazy_init(loss.device)
        return loss * self._scale.to(loss.dtype).reshape(())

    def get_scale(self):
        return self._init_scale if self._scale is None else float(self._scale.item())

    def _grads(self, optimizer):
        return [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]

    def unscale_(self, optimizer):
        """Divide every gradient by the scale in place and record whether any of them is inf / nan."""
        if not self._enabled or self._unscaled:
            return
        grads = self._grads(optimizer)
        if not grads:
            return
        device = grads[0].device
        self._lazy_init(device)
        for g in grads:
            if not g.is_cuda or g.dtype != f32:
                raise H.PeppaHipError("GradScaler: gradients must be fp32 tensors on the GPU")
        if not all(g.is_contiguous() for g in grads):
            raise H.PeppaHipError("GradScaler: gradients must be contiguous (they are unscaled in place)")
        self._found.zero_()
        numels = tuple(g.numel() for g in grads)
        if numels not in self._tables:
            counts = (np.asarray(numels, dtype=np.int64) + _CHUNK - 1) // _CHUNK
            ct = np.repeat(np.arange(len(numels), dtype=np.int32), counts)
            starts = np.cumsum(counts) - counts
            co = (np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(starts, counts)) * _CHUNK
            up = lambda a: torch.from_numpy(a).pin_memory().to(device, non_blocking=True)
            if len(self._tables) > 64:
                self._tables.clear()
            self._tables[numels] = (up(ct), up(co), int(counts.sum()))
        ct, co, n_chunks = self._tables[numels]
        tl, keep = H.make_tensor_list(grads, grads, grads, grads, device)
        H.grad_unscale_check(tl, ct, co, n_chunks, _CHUNK, self._scale, self._found)
        self._unscaled = True

    def step(self, optimizer, *args, **kwargs):
        """`optimizer.step()` unless a gradient overflowed (then the step is skipped and `update()` backs the scale off)."""
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        self._reconcile()
        self.unscale_(optimizer)
        if self._found is None:
            return optimizer.step(*args, **kwargs)
        from .optimization import BertAdam
        if isinstance(optimizer, BertAdam):       # decision on the device, bookkeeping one step later
            out = optimizer.step(*args, skip_flag=self._found, **kwargs)
            host = torch.empty(1, dtype=f32).pin_memory()
            host.copy_(self._found, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            self._pending = (event, host, optimizer, list(optimizer.last_stepped))
            return out
        if self._found.item() != 0.0:             # host sync, as in torch.cuda.amp.GradScaler.step
            self._skipped += 1
            return None
        return optimizer.step(*args, **kwargs)

    def update(self):
        if not self._enabled or self._scale is None:
            return
        H.amp_update_scale(self._scale, self._tracker, self._found, self._growth, self._backoff, self._interval)
        self._found.zero_()
        self._unscaled = False

    def state_dict(self):
        self._reconcile()
        return {"scale": self.get_scale(), "growth_factor": self._growth, "backoff_factor": self._backoff,
                "growth_interval": self._interval, "_growth_tracker": self._init_tracker if self._tracker is None else int(self._tracker.item())}

    def load_state_dict(self, sd):
        self._init_scale = float(sd["scale"])
        self._init_tracker = int(sd.get("_growth_tracker", 0))
        self._growth, self._backoff, self._interval = sd["growth_factor"], sd["backoff_factor"], sd["growth_interval"]
        if self._scale is not None:
            self._scale.fill_(self._init_scale)
            self._tracker.fill_(self._init_tracker)
