"""Data-parallel pieces (new work: the reference is single-GPU).

One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI) or "gloo" (CPU tests):
  * `gather_embeddings`: ONE all-gather of the concatenated [V_local; A_local] (2 x B x 512 fp32,
    256 KiB per rank -- latency bound) so every rank evaluates the triplet loss on the global
    N = world x B negative pool.  Backward needs no collective: every rank holds the same global
    dV/dA and keeps its own rows.
  * `GradBuckets`: gradients are copied into a few large flat fp32 buckets (one per encoder, sized
    for 7 x 153 GB/s point-to-point xGMI links rather than for many small NVSwitch-style calls) and
    all-reduced with SUM as soon as a bucket is complete (asynchronously, on RCCL's stream), overlapping
    the other encoder's backward.  SUM, not mean: each rank's loss is already the global mean and its backward
    only covers the local clips ("all-gather gradient routing").
BatchNorm uses per-rank statistics (documented deviation from a single-process N=512 batch).
"""
import os
import torch
import torch.distributed as dist


def is_dist():
    """True under torch.distributed with more than one rank (PEPPA_FORCE_DIST=1 also exercises the collective
    path with a single rank, which is how the RCCL plumbing is smoke-tested on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PEPPA_FORCE_DIST") == "1"


class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, V, A):
        world, rank = dist.get_world_size(), dist.get_rank()
        B, D = V.shape
        local = torch.cat([V, A], dim=0).contiguous()            # plumbing: pack for one collective
        out = torch.empty(world * 2 * B, D, dtype=V.dtype, device=V.device)
        dist.all_gather_into_tensor(out, local)
        out = out.view(world, 2, B, D)
        ctx.rank, ctx.B = rank, B
        return out[:, 0].reshape(world * B, D), out[:, 1].reshape(world * B, D)

    @staticmethod
    def backward(ctx, dVg, dAg):
        r, B = ctx.rank, ctx.B
        return dVg[r * B:(r + 1) * B].contiguous(), dAg[r * B:(r + 1) * B].contiguous()


def gather_embeddings(V, A):
    """(B,D),(B,D) local -> (world*B,D),(world*B,D) global, rank-major; identity when not distributed."""
    if not is_dist():
        return V, A
    return _GatherFn.apply(V, A)


_ACTIVE = None   # the GradBuckets instance the towers hand their gradients to as they are produced


class _PushDict(dict):
    """`grads[param] = tensor` inside a tower's backward also hands the gradient to the bucket manager."""

    def __init__(self, mgr):
        super().__init__()
        self._mgr = mgr

    def __setitem__(self, k, v):
        # a gradient the bucket manager TOOK is not handed to autograd as well: the tower returns None for it.  (Returned too,
        # it has two owners, so autograd cannot adopt the tensor and clones it into p.grad -- one hipMemcpyAsync per
        # parameter, 340 per step = 1.4 ms on the towers' streams -- only for finish() to replace p.grad by the bucket view.)
        if not self._mgr.push(k, v):
            super().__setitem__(k, v)


def grad_dict():
    """Dictionary the towers' backward passes fill (param -> gradient).  Under data parallelism every assignment is an
    early hand-off: a tower is ONE autograd node, so autograd itself would deliver all of its gradients in a burst at
    the very end of the backward pass and no all-reduce could overlap it."""
    if _ACTIVE is not None and is_dist():
        return _PushDict(_ACTIVE)
    return {}


class GradBuckets:
    """Flat gradient buckets with per-bucket asynchronous all-reduce (SUM), ONE reduction per optimizer step.

    A bucket is packed and reduced as soon as all of its gradients exist.  They arrive either early, from inside a
    tower's backward (`push`, via grad_dict()), or through autograd's post-accumulate hooks (the heads).

    Gradient accumulation (`accumulate_grad_batches`, hparams_base.yaml:42): only the LAST micro-batch of an optimizer
    step is reduced.  With `sync = False` (or inside `no_sync()`) a backward pass leaves the buckets alone and
    autograd accumulates into `p.grad` as usual; the backward pass that runs with `sync = True` packs
    `p.grad (earlier micro-batches) + this pass's gradient` and all-reduces that once."""

    def __init__(self, named_groups, device):
        """named_groups: [(name, [params])]; params without grad at step time are skipped (zeros)."""
        global _ACTIVE
        from .video import ensure_streams
        ensure_streams(device)   # the towers' side streams must exist before RCCL creates its own (see there)
        self.buckets = []
        self.cuda = torch.device(device).type == "cuda"
        self.sync = True
        for name, params in named_groups:
            params = [p for p in params if p.requires_grad]
            total = sum(p.numel() for p in params)
            flat = torch.zeros(total, dtype=torch.float32, device=device)
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.buckets.append(dict(name=name, params=params, flat=flat, views=views, pending=0, work=None,
                                     pushed={}, events={}))
        self._hooks = []
        self._by_param = {}
        for b in self.buckets:
            for p in b["params"]:
                self._by_param[p] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()
        _ACTIVE = self

    def close(self):
        global _ACTIVE
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if _ACTIVE is self:
            _ACTIVE = None

    def no_sync(self):
        """Context manager for the micro-batches of an optimizer step that must not be reduced (all but the last)."""
        mgr = self

        class _NoSync:
            def __enter__(self):
                self.prev, mgr.sync = mgr.sync, False

            def __exit__(self, *exc):
                mgr.sync = self.prev
                return False
        return _NoSync()

    def reset(self):
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["work"] = None
            b["pushed"] = {}
            b["events"] = {}

    def push(self, p, g):
        """Gradient `g` of parameter `p` is final for this backward pass (called on the stream that produces it).  True: the
        manager took it -- the caller must not return it to autograd as well; `p.grad`, if it exists, holds the earlier
        micro-batches only and is added when the bucket is packed."""
        b = self._by_param.get(p)
        if not self.sync or b is None or g is None or p in b["pushed"] or b["work"] is not None:
            return False   # unsynced micro-batch / not ours / already handed over: autograd's accumulation takes it
        b["pushed"][p] = g
        if self.cuda:
            b["events"][torch.cuda.current_stream()] = True   # (the set of producing streams)
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)
        return True

    def _on_grad(self, p):
        if not self.sync or p.grad is None:   # autograd also runs the hook when a Function returned None for this
            return                            # parameter (e.g. a layer skipped by LayerDrop): nothing arrived
        b = self._by_param[p]
        if p in b["pushed"]:
            return                    # handed over early by its tower (which returned None for it: p.grad is untouched)
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _pack(self, b):
        """flat view <- this optimizer step's gradient of every parameter: the early hand-off plus what earlier
        micro-batches left in p.grad (a gradient handed over early is never given to autograd as well), or p.grad itself for
        the parameters that arrived through autograd; zeros for none."""
        cp_dst, cp_src, add_dst, add_src, zero = [], [], [], [], []
        for p, v in zip(b["params"], b["views"]):
            g = b["pushed"].get(p)
            # (p.grad can BE the view: zero_grad(set_to_none=False) keeps last step's tensor and accumulates into it)
            alias = p.grad is not None and p.grad.data_ptr() == v.data_ptr()
            if g is not None:
                if alias:
                    add_dst.append(v), add_src.append(g)
                    continue
                cp_dst.append(v), cp_src.append(g)
                if p.grad is not None:
                    add_dst.append(v), add_src.append(p.grad)
            elif alias:
                continue
            elif p.grad is not None:
                cp_dst.append(v), cp_src.append(p.grad)
            else:
                zero.append(v)
        if cp_dst:
            # pack: one launch per bucket (torch._foreach_copy_ issues one hipMemcpyAsync per tensor on this stack: ~380 copies
            # of 4 us per step, 1.6 ms on the towers' streams -- profiles/r04_probe_dist_overhead.log)
            if self.cuda and all(s_.is_cuda and s_.dtype == torch.float32 and s_.is_contiguous() for s_ in cp_src):
                from . import hip as H
                H.copy_f32_multi(cp_dst, cp_src)
            else:
                torch._foreach_copy_(cp_dst, cp_src)
        if add_dst:
            torch._foreach_add_(add_dst, add_src)
        if zero:
            torch._foreach_zero_(zero)
        return bool(cp_dst)

    def _launch(self, b):
        # On the stream of the last arrival (a tower's own stream): RCCL orders the collective after the work queued
        # there and runs it on its internal stream, so the remaining kernels of both towers overlap it.  (A dedicated
        # packing stream cost 4 ms per step on one GPU -- see video.ensure_streams.)
        if self.cuda:   # gradients produced on other streams: everything queued there so far comes first (one event
            cur = torch.cuda.current_stream()   # per stream and bucket; per-gradient events cost 4 ms per step)
            for st in b["events"]:
                if st != cur:
                    cur.wait_stream(st)
        self._pack(b)
        if is_dist():
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)
        else:
            b["work"] = True

    def finish(self):
        """Wait for the collectives and point every p.grad at its reduced bucket view.  Buckets that were not complete
        in the last backward pass (a layer skipped by LayerDrop in that micro-batch, unused parameters) are reduced
        here from p.grad.  Parameters without any gradient this step (the LayerDrop decision is shared by the ranks)
        contribute zeros to the sum and keep p.grad = None, so the optimizer skips them like the reference."""
        for b in self.buckets:
            had = [p.grad is not None or p in b["pushed"] for p in b["params"]]
            b["had_grad"] = had
            if b["work"] is None and any(had):
                self._pack(b)
                if is_dist():
                    b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)
        for b in self.buckets:
            if b["work"] is not None and b["work"] is not True:
                b["work"].wait()
            for p, v, had in zip(b["params"], b["views"], b["had_grad"]):
                if had:
                    p.grad = v
        self.reset()


def enable_sync_bn(on=True):
    """BatchNorm statistics over the global batch (SURVEY 8e `SyncBN` option; yaml: `mi355x: {sync_bn: true}`): every
    BN layer all-reduces its (sum, sum of squares) row in the forward pass and its (sum g, sum g*xhat) row in the
    backward pass -- two [2][C] fp32 RCCL all-reduces per layer and step.  Off by default: the reference trains under
    Lightning DDP without `sync_batchnorm`, i.e. with per-rank statistics."""
    from . import layers as L
    if on and dist.is_available() and dist.is_initialized():
        L.SYNC_BN_REDUCE = lambda t: dist.all_reduce(t)
        L.SYNC_BN_GATHER = lambda out, t: dist.all_gather_into_tensor(out, t)     # deterministic mode: partial rows, not sums
        L.SYNC_BN_WORLD = dist.get_world_size()
    else:
        L.SYNC_BN_REDUCE, L.SYNC_BN_GATHER, L.SYNC_BN_WORLD = None, None, 1


def default_buckets(net, device):
    """Bucket layout for PeppaPig: the video tower by stage, the wav2vec2 feature extractor, one bucket per
    transformer layer (so LayerDrop leaves whole buckets empty instead of forcing a late, unoverlapped reduce) and the
    rest of the audio tower.  Mostly 17-100 MB: large enough for RCCL's multi-ring bandwidth over the 7 xGMI links."""
    audio = net.audio_encoder
    groups = []
    enc = getattr(audio.audio, "encoder", None)
    layer_params = set()
    if enc is not None:
        for i, layer in enumerate(enc.transformer.layers):
            ps = list(layer.parameters())
            layer_params.update(ps)
            groups.append((f"audio.layer{i}", ps))
    fe = list(audio.audio.feature_extractor.parameters())
    groups.append(("audio.feature_extractor", fe))
    seen = layer_params | set(fe)
    groups.append(("audio.rest", [p for p in audio.parameters() if p not in seen]))
    # video tower by stage, last stage first in the backward pass: layer4 holds three quarters of its parameters and
    # its gradients exist after the first couple of milliseconds of the backward pass
    trunk = "image" if hasattr(net.video_encoder, "image") else "video"
    stages = {}
    for n, p in net.video_encoder.named_parameters():
        if n.startswith(trunk + ".fc"):
            continue   # torchvision's classifier: frozen / unused
        parts = n.split(".")
        key = parts[1] if parts[0] == trunk and parts[1].startswith("layer") else ("stem" if parts[0] == trunk else "head")
        key = "layer1" if key == "stem" else key           # (stem + layer1: 0.7 M parameters together)
        stages.setdefault(key, []).append(p)
    for key in sorted(stages):
        groups.append((f"video.{key}", stages[key]))
    enable_sync_bn(bool(getattr(net, "config", {}).get("mi355x", {}).get("sync_bn", False)))
    return GradBuckets([(n, ps) for n, ps in groups if any(p.requires_grad for p in ps)], device)
