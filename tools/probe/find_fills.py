"""Who launches torch's fill / copy kernels inside one training step?  torch.profiler with Python stacks, grouped by the
innermost frame of this repository.     python tools/probe/find_fills.py"""
import collections
import os
import sys
import warnings

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import yaml
from torch.profiler import ProfilerActivity, profile

import pig.models
from peppa_amd.data import synthetic_batch

cfg = yaml.safe_load(open(os.path.join(ROOT, "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
opt = net.configure_optimizers()
b = synthetic_batch(64, 16, 112, 36800).to("cuda")


def step(i):
    opt.zero_grad(set_to_none=True)
    loss = net.training_step(b, i)
    loss.backward()
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(4)
torch.cuda.synchronize()
names = collections.Counter(ev.name for ev in prof.events())
print("all CPU-side events:", [(k, v) for k, v in names.most_common(40)])
count = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous", "aten::to"):
        where = "?"
        for fr in ev.stack:
            if "/peppa_amd/" in fr or "/pig/" in fr or "bench.py" in fr or "find_fills" in fr:
                where = fr.replace(ROOT + "/", "")
                break
        count[(ev.name, where)] += 1
for (name, where), n in sorted(count.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:5d}  {name:18s} {where}")

kern = collections.Counter(ev.name[:90] for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
print("GPU kernels of the step:", sum(kern.values()))
for k, v in kern.most_common(12):
    print(f"{v:5d}  {k}")
