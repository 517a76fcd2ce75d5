"""Where the single-rank cost of the data-parallel machinery goes (PEPPA_FORCE_DIST=1, world 1)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, yaml
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", **({"device_id": torch.device("cuda", 0)} if os.environ.get("EAGER") else {}))
import pig.models
from peppa_amd.data import synthetic_batch
from peppa_amd import dist as PD
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
opt = net.configure_optimizers()
b = synthetic_batch(64, 16, 112, 36800).to("cuda")


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def plain():
    opt.zero_grad(set_to_none=True); net.training_step(b, 1).backward(); opt.step()


print(f"no DP machinery                      {timeit(plain):.2f} ms")
os.environ["PEPPA_FORCE_DIST"] = "1"
buckets = PD.default_buckets(net, torch.device("cuda", 0))


def dp():
    opt.zero_grad(set_to_none=True); net.training_step(b, 1).backward(); buckets.finish(); opt.step()


print(f"gather + buckets + all-reduce        {timeit(dp):.2f} ms   ({len(buckets.buckets)} buckets)")
dist.barrier()
print(f"  ... after a dist.barrier()         {timeit(dp):.2f} ms")
real = dist.all_reduce
dist.all_reduce = lambda *a, **k: None
print(f"gather + buckets, all-reduce skipped {timeit(dp):.2f} ms")
dist.all_reduce = real


def gather_only():   # all-gather of the embeddings, no gradient buckets (hooks stay registered but finish() is not called)
    opt.zero_grad(set_to_none=True); net.training_step(b, 1).backward(); opt.step()


import types
buckets._launch = types.MethodType(lambda self, b: None, buckets)   # hooks still count, nothing is copied or reduced


def hooks_only():
    opt.zero_grad(set_to_none=True); net.training_step(b, 1).backward(); buckets.reset(); opt.step()


print(f"hooks only (no copy, no reduce)      {timeit(hooks_only):.2f} ms")
for h in buckets._hooks:
    h.remove()
print(f"gather only (bucket hooks removed)   {timeit(gather_only):.2f} ms")
os.environ["PEPPA_FORCE_DIST"] = "0"
print(f"nothing (communicator still alive)   {timeit(plain):.2f} ms")
os.environ["PEPPA_FORCE_DIST"] = "1"
for bk in buckets.buckets:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): dist.all_reduce(bk["flat"])
    torch.cuda.synchronize()
    print(f"  all_reduce {bk['name']:24s} {bk['flat'].numel()*4/1e6:7.1f} MB: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
dist.destroy_process_group()
