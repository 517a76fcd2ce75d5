"""Data-parallel logic on CPU (gloo, world_size 2): the embedding all-gather with its local-rows
backward and the SUM-reduced gradient buckets reproduce the single-process global-batch gradient
of the reference loss (oracle)."""
import os
import socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from peppa_amd.dist import gather_embeddings, GradBuckets
    from oracle import model as O
    torch.manual_seed(0)
    B, D = 3, 16
    enc_v, enc_a = torch.nn.Linear(8, D), torch.nn.Linear(6, D)        # identical replicas on both ranks
    g = torch.Generator().manual_seed(100)
    xv, xa = torch.randn(world * B, 8, generator=g), torch.randn(world * B, 6, generator=g)
    buckets = GradBuckets([("a", list(enc_a.parameters())), ("v", list(enc_v.parameters()))], "cpu")
    V = enc_v(xv[rank * B:(rank + 1) * B])
    A = enc_a(xa[rank * B:(rank + 1) * B])
    Vg, Ag = gather_embeddings(V, A)
    assert Vg.shape == (world * B, D)
    loss = O.TripletLoss(0.2)(Vg, Ag)
    # the audio "tower" hands its gradients to the buckets itself, before autograd delivers them (early hand-off as
    # in Wav2Vec2Fn.backward via dist.grad_dict()); the video encoder's arrive through the post-accumulate hooks
    from peppa_amd.dist import grad_dict
    os.environ["PEPPA_FORCE_DIST"] = "1"
    gd = grad_dict()
    assert type(gd) is not dict
    ga_w, ga_b = torch.autograd.grad(loss, [enc_a.weight, enc_a.bias], retain_graph=True)
    gd[enc_a.weight], gd[enc_a.bias] = ga_w, ga_b
    assert buckets.buckets[0]["work"] is not None            # bucket "a" was complete and went out at once
    loss.backward()
    buckets.finish()
    if rank == 0:
        torch.save({"loss": loss.detach(), "gv": enc_v.weight.grad.clone(), "ga": enc_a.bias.grad.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_sum_buckets_match_global_batch(tmp_path):
    from oracle import model as O
    world, port, out = 2, _free_port(), str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    B, D = 3, 16
    enc_v, enc_a = torch.nn.Linear(8, D), torch.nn.Linear(6, D)
    g = torch.Generator().manual_seed(100)
    xv, xa = torch.randn(world * B, 8, generator=g), torch.randn(world * B, 6, generator=g)
    loss = O.TripletLoss(0.2)(enc_v(xv), enc_a(xa))
    loss.backward()
    assert abs(loss.item() - got["loss"].item()) < 1e-6
    assert (enc_v.weight.grad - got["gv"]).abs().max() < 1e-6   # SUM over ranks == single-process gradient
    assert (enc_a.bias.grad - got["ga"]).abs().max() < 1e-6


def test_single_process_is_identity():
    from peppa_amd.dist import gather_embeddings, is_dist
    V, A = torch.randn(2, 4), torch.randn(2, 4)
    assert not is_dist()
    V2, A2 = gather_embeddings(V, A)
    assert V2 is V and A2 is A


def test_default_buckets_cover_every_trainable_parameter_once():
    import copy
    import warnings
    warnings.filterwarnings("ignore")
    import pig.models
    from pig.execution import default_config
    from peppa_amd.dist import default_buckets
    cfg = copy.deepcopy(default_config)
    cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
    cfg["audio"]["freeze_feature_extractor"] = True
    net = pig.models.PeppaPig(cfg)
    gb = default_buckets(net, "cpu")
    in_buckets = [p for b in gb.buckets for p in b["params"]]
    assert len(in_buckets) == len(set(in_buckets))
    want = {p for n, p in net.named_parameters() if p.requires_grad and "video.fc" not in n}
    assert set(in_buckets) == want
    names = [b["name"] for b in gb.buckets]
    assert "audio.layer11" in names and "video.layer4" in names and "video.layer1" in names and "video.head" in names
    assert "audio.feature_extractor" not in names
    # a bucket whose parameters got no gradient at all is skipped; a complete one is reduced (no-op on 1 process)
    layer0 = next(b for b in gb.buckets if b["name"] == "audio.layer0")
    layer1 = next(b for b in gb.buckets if b["name"] == "audio.layer1")
    for p in layer1["params"]:      # autograd also fires the hook when a Function returned None (LayerDrop)
        gb._on_grad(p)
    assert layer1["pending"] == len(layer1["params"])
    for p in layer0["params"]:
        p.grad = torch.ones_like(p)
        gb._on_grad(p)
    gb.finish()
    assert all(p.grad is not None for p in layer0["params"])
    assert all(p.grad is None for b in gb.buckets if b["name"] == "audio.layer1" for p in b["params"])
