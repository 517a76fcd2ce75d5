import os, sys, copy, warnings
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
import test_model_gpu as T
from peppa_amd.data import synthetic_batch
from peppa_amd.dist import default_buckets
cfg = T.make_cfg()
_, net = T.build_pair(cfg)
net.train()
batch = synthetic_batch(4, 4, 32, 4000).to("cuda")
def grads_plain():
    net.zero_grad(set_to_none=True); net.training_step(batch, 0).backward(); torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
def cmp(a, b, tag):
    gmax = max(v.abs().max().item() for v in b.values())
    worst = sorted(((( a[n] - b[n]).abs().max().item() / max(b[n].abs().max().item(), 1e-2 * gmax), n) for n in b), reverse=True)[:5]
    print(tag, f"gmax {gmax:.3g}", [(f"{e:.2e}", n[-48:]) for e, n in worst])
g1 = grads_plain(); g2 = grads_plain()
cmp(g1, g2, "plain vs plain:")
dist.init_process_group("nccl")
os.environ["PEPPA_FORCE_DIST"] = "1"
buckets = default_buckets(net, torch.device("cuda"))
net.zero_grad(set_to_none=True); net.training_step(batch, 0).backward(); buckets.finish(); torch.cuda.synchronize()
g3 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
cmp(g3, g1, "dp vs plain:   ")
os.environ["PEPPA_FORCE_DIST"] = "0"; buckets.close()
g4 = grads_plain()
cmp(g4, g1, "plain again:   ")
dist.destroy_process_group()
