import os, sys, time, warnings
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
warnings.filterwarnings("ignore")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, yaml
import torch.distributed as dist
torch.cuda.set_device(0)
import pig.models
from peppa_amd.data import synthetic_batch
cfg = yaml.safe_load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "hparams_base.yaml")))
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
print(f"before init_process_group             {timeit(plain):.2f} ms", flush=True)
dist.init_process_group("nccl")   # lazy: no communicator yet
print(f"after lazy init (no collective yet)   {timeit(plain):.2f} ms", flush=True)
x = torch.ones(1024, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
print(f"after the first all_reduce            {timeit(plain):.2f} ms", flush=True)
print("env:", {k: v for k, v in os.environ.items() if "NCCL" in k or "RCCL" in k or "HSA" in k}, flush=True)
dist.destroy_process_group()
print(f"after destroy_process_group           {timeit(plain):.2f} ms", flush=True)
