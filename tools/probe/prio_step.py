"""Does a HIGH-priority HIP stream for the video trunk (the step's critical chain) keep the audio tower's kernels from
delaying it?  The step runs under `with torch.cuda.stream(s)`: s = a priority -1 stream (trunk high, audio tower's side
stream normal) against s = a priority 0 stream, alternating inside one process.     python tools/probe/prio_step.py"""
import os, sys, time, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, yaml
import pig.models
from peppa_amd.data import synthetic_batch

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = yaml.safe_load(open(os.path.join(root, "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
opt = net.configure_optimizers()
b = synthetic_batch(64, 16, 112, 36800).to("cuda")
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
streams = {"normal (0)": torch.cuda.Stream(priority=0), "high (-1)": torch.cuda.Stream(priority=-1)}


def step(i):
    opt.zero_grad(set_to_none=True)
    net.training_step(b, i).backward()
    opt.step()


def timed(st, n=12):
    with torch.cuda.stream(st):
        step(0); step(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(2 + i)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step(0)
torch.cuda.synchronize()
for rep in range(3):
    for name, st in streams.items():
        print(f"trunk stream priority {name}: {timed(st):.2f} ms/step", flush=True)
