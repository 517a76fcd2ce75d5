import os, sys, time, warnings
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch, yaml
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ["PEPPA_FORCE_DIST"] = "1"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if os.environ.get("PREALLOC"):   # reserve allocator segments before the communicator exists
    blk = torch.empty(int(float(os.environ["PREALLOC"]) * 2**30), dtype=torch.uint8, device=dev)
    del blk
if os.environ.get("PRESTREAM"):   # create the tower side streams before RCCL creates its own
    from peppa_amd import video as PV
    ws = PV._wgrad_stream(dev)
    with torch.cuda.stream(ws):
        torch.zeros(8, device=dev).add_(1)
    torch.cuda.synchronize()
dist.init_process_group("nccl")
import pig.models
from peppa_amd.data import synthetic_batch
from peppa_amd.dist import default_buckets
cfg = yaml.safe_load(open(os.path.join(ROOT, "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).to(dev).train()
if os.environ.get("PRESTREAM"):
    net._side_stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(net._side_stream):
        torch.zeros(8, device=dev).add_(1)
    torch.cuda.synchronize()
    x = torch.ones(8, device=dev); dist.all_reduce(x); torch.cuda.synchronize()   # RCCL's stream comes last
optim = net.configure_optimizers()
batch = synthetic_batch(64, 16, 112, 36800, seed=1234).to(dev)
buckets = default_buckets(net, dev) if os.environ.get("BUCKETS", "1") == "1" else None
def step(i):
    optim.zero_grad(set_to_none=True)
    loss = net.training_step(batch, i)
    loss.backward()
    if buckets is not None:
        buckets.finish()
    optim.step()
    return loss
if os.environ.get("LOCAL_WARM"):   # allocator warm-up without collectives first
    os.environ["PEPPA_FORCE_DIST"] = "0"
    for i in range(3):
        optim.zero_grad(set_to_none=True); net.training_step(batch, i).backward(); optim.step()
    torch.cuda.synchronize()
    os.environ["PEPPA_FORCE_DIST"] = "1"
for i in range(5): step(i)
torch.cuda.synchronize()
ts = []
for i in range(12):
    t0 = time.perf_counter(); step(5 + i); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step (sync each):", " ".join(f"{t:.1f}" for t in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(12): step(20 + i)
torch.cuda.synchronize(); print(f"free-running: {(time.perf_counter() - t0) / 12 * 1e3:.2f} ms  (peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB)")
dist.destroy_process_group()
