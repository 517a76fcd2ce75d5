"""`pig.optimization` (pig/optimization.py:26-179): BertAdam and its schedules.

Same constructor, defaults, state names (`step`, `next_m`, `next_v`) and update rule as the
reference: per-tensor gradient clipping, no bias correction, epsilon outside the square root,
decoupled weight decay on every tensor, schedule multiplier from the per-tensor step count
(0 at step 0).  The update itself is one fused multi-tensor HIP launch pair instead of ~10
elementwise kernels per parameter tensor."""
import math
import logging
import numpy as np
import torch
from torch.optim import Optimizer
from torch.optim.optimizer import required

from . import hip as H
from .hip import f32

logger = logging.getLogger(__name__)


def warmup_cosine(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 1.0


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {
    'warmup_cosine': warmup_cosine,
    'warmup_constant': warmup_constant,
    'warmup_linear': warmup_linear,
}

_CHUNK = 65536


class BertAdam(Optimizer):
    def __init__(self, params, lr=required, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999,
                 e=1e-6, weight_decay=0.01, max_grad_norm=1.0):
        if lr is not required and lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super(BertAdam, self).__init__(params, defaults)
        self._chunk_cache = {}

    def _lr(self, group, step):
        if group['t_total'] != -1:
            return group['lr'] * SCHEDULES[group['schedule']](step / group['t_total'], group['warmup'])
        return group['lr']

    def get_lr(self):
        lr = []
        for group in self.param_groups:
            for p in group['params']:
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                lr.append(self._lr(group, state['step']))
        return lr

    def _chunks(self, numels, device):
        """Chunk tables of one launch: chunk c covers tensor ct[c], elements [co[c], co[c] + _CHUNK).  Built with numpy
        and uploaded from pinned memory without blocking: which tensors take part changes from step to step (LayerDrop
        leaves the skipped layers without gradients), and a pageable `torch.tensor(..., device=)` here made the host
        wait for the whole backward pass every time the pattern was new."""
        key = (tuple(numels), str(device))
        if key not in self._chunk_cache:
            counts = (np.asarray(numels, dtype=np.int64) + _CHUNK - 1) // _CHUNK
            ct = np.repeat(np.arange(len(numels), dtype=np.int32), counts)
            starts = np.cumsum(counts) - counts
            co = (np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(starts, counts)) * _CHUNK
            up = lambda a: torch.from_numpy(a).pin_memory().to(device, non_blocking=True)
            if len(self._chunk_cache) > 256:      # (patterns recur; the cap only guards against unbounded growth)
                self._chunk_cache.clear()
            self._chunk_cache[key] = (up(ct), up(co), int(counts.sum()), torch.empty(len(numels) + int(counts.sum()), dtype=f32, device=device))   # norms [tensors] + [chunks] (deterministic mode)
        return self._chunk_cache[key]

    def step(self, closure=None, skip_flag=None):
        """skip_flag (fp16 runs, peppa_amd.amp.GradScaler): a device float; non-zero = a gradient overflowed and the
        launch leaves parameters and moments untouched.  The per-tensor step counters below are advanced here on the host
        either way; `amp.GradScaler` takes the increment back one step later, when the flag has arrived, so the schedule
        sees exactly the steps the reference's skipped `optimizer.step()` calls would have left (pig/optimization.py:160-170)."""
        loss = None
        if closure is not None:
            loss = closure()
        warned = False
        self.last_stepped = []
        for group in self.param_groups:
            per_device = {}  # every tensor of the group with a gradient, per device: ONE fused launch pair
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                if not p.is_cuda:
                    raise H.PeppaHipError("BertAdam on the HIP path needs CUDA/HIP parameters (no CPU fallback)")
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['next_m'] = torch.zeros_like(p.data)
                    state['next_v'] = torch.zeros_like(p.data)
                per_device.setdefault(p.device, []).append(p)
            for device, ps in per_device.items():
                # the reference schedules every tensor by its OWN step count (pig/optimization.py:160-170): tensors that
                # skipped steps lag behind, so the scheduled learning rate is per tensor
                lrs = []
                for p in ps:
                    step = self.state[p]['step']
                    if (group['t_total'] != -1 and group['schedule'] == "warmup_linear" and
                            step / group['t_total'] > 1. and not warned):
                        logger.warning("Training beyond specified 't_total' steps with schedule '{}'.".format(group['schedule']))
                        warned = True
                    lrs.append(self._lr(group, step))
                gs = [p.grad.data if p.grad.is_contiguous() else p.grad.data.contiguous() for p in ps]
                ms = [self.state[p]['next_m'] for p in ps]
                vs = [self.state[p]['next_v'] for p in ps]
                tl, keep = H.make_tensor_list([p.data for p in ps], gs, ms, vs, device)
                lr_t = torch.tensor(lrs, dtype=f32).pin_memory().to(device, non_blocking=True)
                ct, co, n_chunks, norms = self._chunks([p.numel() for p in ps], device)
                H.bertadam_step(tl, ct, co, n_chunks, _CHUNK, norms, float(lrs[0]), group['b1'], group['b2'], group['e'],
                                group['weight_decay'], group['max_grad_norm'], lr_t=lr_t, skip=skip_flag)
                for p in ps:
                    self.state[p]['step'] += 1
                self.last_stepped += ps
        from . import layers as L
        L.weights_changed()     # the masters moved through raw pointers: cached 16-bit operand layouts are stale
        return loss
