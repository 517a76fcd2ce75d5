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


class GradBuckets:
    """Flat gradient buckets with per-bucket asynchronous all-reduce (SUM)."""

    def __init__(self, named_groups, device):
        """named_groups: [(name, [params])]; params without grad at step time are skipped (zeros)."""
        from .video import ensure_streams
        ensure_streams(device)   # the towers' side streams must exist before RCCL creates its own (see there)
        self.buckets = []
        for name, params in named_groups:
            params = [p for p in params if p.requires_grad]
            total = sum(p.numel() for p in params)
            flat = torch.zeros(total, dtype=torch.float32, device=device)
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.buckets.append(dict(name=name, params=params, flat=flat, views=views, pending=0, work=None))
        self._hooks = []
        self._by_param = {}
        for b in self.buckets:
            for p in b["params"]:
                self._by_param[p] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def reset(self):
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["work"] = None

    def _on_grad(self, p):
        if p.grad is None:   # autograd also runs the hook when a Function returned None for this parameter
            return           # (e.g. a layer skipped by LayerDrop): nothing arrived
        b = self._by_param[p]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        # On the stream the hook runs on (the tower's own): RCCL orders the collective after the work queued there and
        # runs it on its internal stream, so the tower's remaining kernels and the other tower overlap it.  A dedicated
        # stream for the packing cost 4 ms per step on one GPU: with the video, audio and weight-gradient streams it
        # is the fifth, and HIP maps streams onto four hardware queues -- two streams then share one, in order.
        torch._foreach_copy_(b["views"], [p.grad for p in b["params"]])          # plumbing: pack
        if is_dist():
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """Wait for the collectives and point every p.grad at its reduced bucket view.  Parameters that got no
        gradient this step (unused, or a layer skipped by LayerDrop on every rank -- the decision is shared)
        contribute zeros to the sum and keep p.grad = None, so the optimizer skips them like the reference."""
        for b in self.buckets:
            b["had_grad"] = [p.grad is not None for p in b["params"]]
            if b["pending"] == len(b["params"]):
                continue   # nothing arrived (e.g. a layer dropped by LayerDrop on every rank): no collective
            if b["pending"] > 0:
                if sum(not h for h in b["had_grad"]) != b["pending"]:
                    raise RuntimeError(f"bucket {b['name']}: inconsistent gradient arrival")
                for p, v in zip(b["params"], b["views"]):
                    if p.grad is None:
                        v.zero_()
                    else:
                        v.copy_(p.grad)
                if is_dist():
                    b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)
            if b["work"] is not None:
                b["work"].wait()
        for b in self.buckets:
            for p, v, had in zip(b["params"], b["views"], b["had_grad"]):
                if had:
                    p.grad = v
        self.reset()


def default_buckets(net, device):
    """Bucket layout for PeppaPig: the video tower, the wav2vec2 feature extractor, one bucket per transformer
    layer (so LayerDrop leaves whole buckets empty instead of forcing a late, unoverlapped reduce) and the rest of
    the audio tower.  24-130 MB each: large enough for RCCL's multi-ring bandwidth over the 7 xGMI links."""
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
    vname = "image.fc" if hasattr(net.video_encoder, "image") else "video.fc"
    groups.append(("video", [p for n, p in net.video_encoder.named_parameters() if not n.startswith(vname)]))
    return GradBuckets([(n, ps) for n, ps in groups if any(p.requires_grad for p in ps)], device)
