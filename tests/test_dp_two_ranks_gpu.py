"""Two data-parallel ranks driving the REAL PeppaPig on one GPU (no 8-GPU node is available to the
builder): two processes share cuda:0 and talk through gloo (RCCL refuses two ranks on one device), which exercises
everything but the transport -- the embedding all-gather with its local-rows backward, `default_buckets` with the towers'
early gradient hand-off, LayerDrop with rank-shared decisions (whole buckets stay empty), SyncBN statistics over the
global batch, gradient accumulation with one reduction per optimizer step -- against ONE process running the global batch.
"""
import copy
import os
import socket
import warnings
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore")

B, FRAMES, SIZE, SAMPLES = 4, 16, 64, 16000   # (16 frames: layer 4 keeps 2 x 4 x 4 = 32 rows per clip, 128 per rank = whole statistics blocks)


def _cfg(sync_bn):
    from pig.execution import default_config
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    cfg["mi355x"] = {"sync_bn": sync_bn}
    return cfg


def _net(cfg, layer_drop):
    import pig.models
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                      # element masks are indexed by the LOCAL tensor: not comparable across layouts
        if hasattr(m, "layer_drop"):
            m.layer_drop = layer_drop      # decisions come from a generator seeded identically in every process
    return net.cuda().train()


def _batches(world, accumulate):
    from peppa_amd.data import synthetic_batch
    return [synthetic_batch(world * B, FRAMES, SIZE, SAMPLES, seed=50 + k) for k in range(accumulate)]


def _objective(net, batch, k, world):
    """The step's loss (value only) and a SMOOTH objective <V_global, Rv> + <A_global, Ra> through the same all-gather:
    the hinge loss of near-identical random-init embeddings amplifies last-bit differences of V ~100x (tests/
    test_parity_c2_gpu.py), which would bury a routing error of a few per cent; its own data-parallel arithmetic is
    checked exactly on the CPU (tests/test_dist_cpu.py)."""
    from peppa_amd.dist import gather_embeddings
    g = torch.Generator().manual_seed(900 + k)
    Rv, Ra = (torch.randn(world * B, 512, generator=g).cuda() for _ in range(2))
    V, A = net.encode_pair(batch.video, batch.audio)
    Vg, Ag = gather_embeddings(V, A)
    with torch.no_grad():
        loss = net.loss(Vg, Ag).item()
    return loss, (Vg * Rv).sum() + (Ag * Ra).sum()


def _worker(rank, world, port, out, layer_drop, accumulate, det=False):
    warnings.filterwarnings("ignore")
    if det:
        from peppa_amd import hip as H
        H.set_deterministic(True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from peppa_amd.dist import default_buckets
    net = _net(_cfg(True), layer_drop)
    buckets = default_buckets(net, torch.device("cuda", 0))
    losses = []
    for k, gb in enumerate(_batches(world, accumulate)):
        sl = slice(rank * B, (rank + 1) * B)
        local = type(gb)(gb.video[sl].cuda(), gb.audio[sl].cuda(), gb.video_duration[sl], gb.audio_duration[sl])
        buckets.sync = k == accumulate - 1
        loss, obj = _objective(net, local, k, world)
        (obj / accumulate).backward()
        losses.append(loss)
    pushed = sum(len(b["pushed"]) for b in buckets.buckets)
    buckets.finish()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"loss": losses, "pushed": pushed,
                    "grads": {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("layer_drop,accumulate,det", [(0.0, 1, False), (0.3, 2, False), (0.0, 1, True), (0.3, 2, True)])
def test_two_ranks_match_one_process_on_the_global_batch(tmp_path, layer_drop, accumulate, det):
    """det: the deterministic mode (pp_set_option("deterministic", 1)) in the ranks and in the reference process.  Every
    reduction is then ordered and SyncBN exchanges the ranks' partial rows instead of their sums, so the two ranks compute
    bit for bit the activations and data gradients of the one process; what is left is the fp32 order in which a weight
    gradient's rows are added (per rank, then across ranks): every group agrees to <= 1e-3 instead of the per-cent /
    cosine >= 0.5 bounds the atomics force on the default mode."""
    from peppa_amd import hip as H
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(world, port, out, layer_drop, accumulate, det), nprocs=world, join=True)
    got = torch.load(out)
    prev = H.set_deterministic(det)
    try:
        # one process, the global batch, plain BatchNorm over all 2B clips (= what SyncBN computes across the two ranks)
        net = _net(_cfg(False), layer_drop)
        losses = []
        for k, gb in enumerate(_batches(world, accumulate)):
            loss, obj = _objective(net, gb.to("cuda"), k, world)
            (obj / accumulate).backward()
            losses.append(loss)
        torch.cuda.synchronize()
        print("losses", losses, got["loss"], "early hand-offs", got["pushed"])
        assert got["pushed"] > 150
        for a, b in zip(losses, got["loss"]):
            assert abs(a - b) <= (1e-6 if det else 2e-3), (losses, got["loss"])
        ref = {n: p.grad.cpu() for n, p in net.named_parameters() if p.grad is not None}
        assert set(ref) == set(got["grads"]), set(ref) ^ set(got["grads"])      # same tensors skipped by LayerDrop / unused
        if layer_drop > 0:
            assert len(ref) < sum(1 for _ in net.parameters()) - 2              # (some layer was dropped in both micro-batches or fc)
        # per group: pooled relative error, ratio of the pooled norms, pooled cosine (tensors whose true gradient is ~0,
        # e.g. k_proj.bias, only count through the pooled figures)
        acc = {}
        for n, g in ref.items():
            d = got["grads"][n]
            parts = n.split(".")
            key = ".".join(parts[:3]) if parts[1] in ("video", "audio") else ".".join(parts[:2])
            a = acc.setdefault(key, [0.0, 0.0, 0.0, 0.0])
            a[0] += (d - g).pow(2).sum().item(); a[1] += g.pow(2).sum().item(); a[2] += d.pow(2).sum().item(); a[3] += (d * g).sum().item()
        stats = {k: ((e / r) ** 0.5, (dd / r) ** 0.5, dg / (r * dd) ** 0.5) for k, (e, r, dd, dg) in acc.items()}
        for k in sorted(stats):
            print(f"  {k:40s} rel-err {stats[k][0]:.4f}  norm ratio {stats[k][1]:.4f}  cosine {stats[k][2]:.4f}")
        # The two runs differ only by summation order (SyncBN's all-reduced fp32 sums vs one process's block sums, float
        # atomics).  Audio tower and heads are well conditioned: a few 1e-3..1e-2.  The random-init train-mode-BatchNorm video
        # trunk at this small shape is chaotic in its backward pass (two identical plain steps differ by several per cent,
        # DESIGN.md "Numerics"): for it the check is the one a routing error cannot pass -- a factor of `world`, a rank's
        # rows lost, a bucket reduced twice or never change the NORM of a stage's gradient by >= 30 % or decorrelate it.
        # (measured: audio 0.3-1.1 %, video projection 3.5 %; trunk stages norm ratio 1.002-1.007, cosine 0.78-0.86)
    finally:
        H.set_deterministic(prev)
    gmax = max(r for _, r, _, _ in acc.values())
    for k, (err, ratio, cos) in stats.items():
        if acc[k][1] < 1e-8 * gmax:
            continue      # analytically zero here (the video attention pooling over a single output frame)
        if det:
            assert err <= 1e-3 and abs(ratio - 1) <= 1e-3 and cos >= 0.999999, (k, err, ratio, cos)
        elif k.startswith("video_encoder.video."):
            assert 0.9 <= ratio <= 1.1 and cos >= 0.5, (k, err, ratio, cos)
        elif k == "video_encoder.videopool":
            # the attention pooling over the trunk's two output frames sits right behind the chaotic trunk (measured 0.11)
            assert err <= 0.25 and 0.97 <= ratio <= 1.03 and cos >= 0.97, (k, err, ratio, cos)
        else:
            assert err <= 0.06 and 0.97 <= ratio <= 1.03, (k, err, ratio, cos)
