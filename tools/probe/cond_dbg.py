import os, sys, warnings
warnings.filterwarnings("ignore")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, torch.nn.functional as F
import parity_c2_report as R
from peppa_amd.data import synthetic_structured_batch
import pig.optimization
cw = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
cfg = R.make_cfg(); ref, net = R.build_pair(cfg)
pool, batch, steps = 6, 8, int(sys.argv[2]) if len(sys.argv) > 2 else 300
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-4
batches = [synthetic_structured_batch(batch, 16, 112, 36800, seed=100 + k).to("cuda") for k in range(pool)]
g = torch.Generator().manual_seed(7)
common = torch.randn(1, 1, 512, generator=g)
targets = F.normalize(cw * common + torch.randn(pool, batch, 512, generator=g), dim=-1).cuda()
optim = pig.optimization.BertAdam(net.parameters(), lr=lr, warmup=0.05, t_total=2 * steps)
for i in range(steps):
    optim.zero_grad(set_to_none=True)
    V, A = net.encode_pair(batches[i % pool].video, batches[i % pool].audio)
    obj = -((V * targets[i % pool]).sum() + (A * targets[i % pool]).sum()) / (2 * batch)
    obj.backward(); optim.step()
with torch.no_grad():
    print("steps", steps, "lr", lr)
    for k in (0,):
        V, A = net.encode_pair(batches[k].video, batches[k].audio)
        S = (F.normalize(V, dim=1) @ F.normalize(A, dim=1).t()).cpu()
        off = ~torch.eye(batch, dtype=torch.bool)
        print(f"batch {k}: cos(V, target) {[round(x, 2) for x in (V * targets[k]).sum(1).tolist()]}")
        print(f"          cos(A, target) {[round(x, 2) for x in (A * targets[k]).sum(1).tolist()]}")
        print(f"          S diag {S.diag().mean():.3f} off {S[off].mean():.3f};  VV off-diag {(F.normalize(V,dim=1)@F.normalize(V,dim=1).t()).cpu()[off].mean():.3f}  AA off-diag {(F.normalize(A,dim=1)@F.normalize(A,dim=1).t()).cpu()[off].mean():.3f}")
