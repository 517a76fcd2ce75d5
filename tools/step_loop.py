"""The bench step in a bare loop, for rocprofv3 (no CPU baseline, no forward-only pass, no event timing).

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/step_loop.py --steps 6 [--isolated]

--isolated: the two towers and the weight-gradient kernels run on ONE stream, one kernel at a time, so the profile's
average durations are those of each kernel ALONE on the GPU (tools/prof_families.py puts the two runs side by side)."""
import argparse
import os
import sys
import time
import warnings

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yaml

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--samples", type=int, default=36800)
ap.add_argument("--size", default="112", help="frame size H or HxW")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--isolated", action="store_true")
ap.add_argument("--no-overlap-audio", action="store_true")
ap.add_argument("--no-overlap-wgrad", action="store_true")
ap.add_argument("--config", default="hparams_base.yaml")
ap.add_argument("--launch-log", default=None, help="write every pp_igemm / pp_wgrad call (shape, mode) of the run, in launch "
                                                   "order, as JSON: tools/prof_shapes.py joins it with the kernel trace")
args = ap.parse_args()

import pig.models
from peppa_amd import hip as H
from peppa_amd import video as PV
from peppa_amd.data import synthetic_batch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml.safe_load(open(os.path.join(root, args.config)))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
net.set_precision(args.dtype)
if args.isolated:
    net._overlap, PV.OVERLAP_WGRAD = False, False
if args.no_overlap_audio:
    net._overlap = False
if args.no_overlap_wgrad:
    PV.OVERLAP_WGRAD = False
opt = net.configure_optimizers()
scaler = None
if args.dtype == "fp16":
    from peppa_amd.amp import GradScaler
    scaler = GradScaler()
hw = [int(v) for v in args.size.lower().split("x")]
b = synthetic_batch(args.batch, args.frames, hw[0] if len(hw) == 1 else (hw[0], hw[1]), args.samples).to("cuda")


def step(i):
    opt.zero_grad(set_to_none=True)
    loss = net.training_step(b, i)
    if scaler is not None:
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
    else:
        loss.backward()
        opt.step()


if args.launch_log:
    H.LAUNCH_LOG = []
for i in range(args.warmup):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    step(args.warmup + i)
torch.cuda.synchronize()
if args.launch_log:
    import json
    with open(args.launch_log, "w") as f:
        json.dump({"steps": args.steps, "warmup": args.warmup, "batch": args.batch, "frames": args.frames,
                   "samples": args.samples, "dtype": args.dtype, "isolated": args.isolated, "launches": H.LAUNCH_LOG}, f)
print(f"{(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms/step over {args.steps} steps ({args.warmup} warm-up), "
      f"isolated={args.isolated}, dtype={args.dtype}, frames={args.frames}, samples={args.samples}, audio overlap "
      f"{net._overlap}, wgrad overlap {PV.OVERLAP_WGRAD}")
