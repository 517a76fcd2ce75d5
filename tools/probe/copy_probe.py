import os, sys, time, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
warnings.filterwarnings("ignore")
import torch, yaml
import pig.models
from peppa_amd import dist as PD
cfg = yaml.safe_load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
net = pig.models.PeppaPig(cfg).cuda()
for p in net.parameters():
    p.grad = torch.randn_like(p)
buckets = PD.default_buckets(net, torch.device("cuda", 0))
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tot = 0
for b in buckets.buckets:
    grads = [p.grad for p in b["params"]]
    t = timeit(lambda: torch._foreach_copy_(b["views"], grads))
    tot += t
    print(f"{b['name']:26s} {len(grads):4d} tensors {b['flat'].numel()*4/1e6:7.1f} MB  foreach_copy {t:.3f} ms")
print("total", tot)
flat_all = torch.empty(sum(b["flat"].numel() for b in buckets.buckets), device="cuda")
src = torch.randn_like(flat_all)
print("one flat copy of the same bytes:", timeit(lambda: flat_all.copy_(src)))
