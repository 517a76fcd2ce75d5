"""When is each gradient bucket complete during the backward pass, and what would its all-reduce cost on eight GPUs?

One rank with every collective live (RCCL, world = 1, like PEPPA_FORCE_DIST=1 in bench.py): GradBuckets._launch records a
HIP event when a bucket has been packed and handed to RCCL; the step records events at the start / end of the backward pass
and at the end of the optimizer.  The table sets the measured hand-off times against a MODELLED eight-GPU ring all-reduce
(this pool gives a builder one GPU per call; the 1 -> 8 curve itself is unmeasured):

    t(S) = latency + 2 (N - 1) / N * S / BW,   N = 8,  BW = --bw GB/s (default 320),  latency = 30 us

BW: xGMI is point to point, 7 links x ~153 GB/s per GPU (task statement) ~ 535 GB/s out of every GPU when all seven rings
run; RCCL's large-message all-reduce reaches roughly 60 % of that on this class of node -> 320 GB/s of "bus bandwidth".

    python tools/bucket_timeline.py [--steps 12] [--bw 320] [--out profiles/r04_bucket_timeline.md]"""
import argparse
import os
import sys
import warnings

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import yaml

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--bw", type=float, default=320.0, help="modelled all-reduce bus bandwidth, GB/s")
ap.add_argument("--latency-us", type=float, default=30.0)
ap.add_argument("--world", type=int, default=8, help="ranks of the MODELLED ring")
ap.add_argument("--out", default=None)
args = ap.parse_args()

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ["PEPPA_FORCE_DIST"] = "1"      # one rank, every collective live (peppa_amd.dist.is_dist)
torch.cuda.set_device(0)
dist.init_process_group("nccl")

import pig.models
from peppa_amd import dist as PD
from peppa_amd.data import synthetic_batch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml.safe_load(open(os.path.join(root, "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
opt = net.configure_optimizers()
batch = synthetic_batch(64, 16, 112, 36800).to("cuda")
buckets = PD.default_buckets(net, torch.device("cuda", 0))

marks = {}          # bucket name -> [event per step]
orig_launch = PD.GradBuckets._launch


STEP = [0]


def launch(self, b):
    orig_launch(self, b)
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()      # on the stream of the last arrival, behind the packing copies and the collective's enqueue
    marks.setdefault(b["name"], []).append((STEP[0], ev, len(b["pushed"]), len(b["params"])))


PD.GradBuckets._launch = launch
steps = []
for i in range(args.warmup + args.steps):
    if i == args.warmup:
        marks.clear()
    opt.zero_grad(set_to_none=True)
    STEP[0] = i - args.warmup
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    loss = net.training_step(batch, i)
    e[1].record()
    loss.backward()
    e[2].record()
    buckets.finish()
    opt.step()
    e[3].record()
    if i >= args.warmup:
        steps.append(e)
torch.cuda.synchronize()
n = len(steps)
fwd = sum(e[0].elapsed_time(e[1]) for e in steps) / n
bwd = sum(e[1].elapsed_time(e[2]) for e in steps) / n
tot = sum(e[0].elapsed_time(e[3]) for e in steps) / n
rows, early = [], {}
for b in buckets.buckets:
    evs = [m for m in marks.get(b["name"], []) if 0 <= m[0] < n]
    mb = b["flat"].numel() * 4 / 1e6
    # (LayerDrop: a layer's bucket is reduced in finish() in the steps that skip it -- those steps carry no mark)
    when = sum(steps[k][1].elapsed_time(ev) for k, ev, _, _ in evs) / len(evs) if evs else float("nan")
    early[b["name"]] = (sum(m[2] for m in evs) / max(1, len(evs)), len(b["params"]), len(evs))
    t_ar = args.latency_us * 1e-3 + 2.0 * (args.world - 1) / args.world * mb * 1e6 / (args.bw * 1e9) * 1e3
    rows.append((when, b["name"], mb, t_ar))
rows.sort(key=lambda r: (r[0] != r[0], r[0]))
lines = ["# Gradient buckets: when each is complete in the backward pass, and its modelled 8-GPU all-reduce\n",
         f"`python tools/bucket_timeline.py` on ONE MI355X with the collectives live (RCCL, world = 1): hparams_base, batch 64, bf16, "
         f"{n} steps after {args.warmup} warm-up.  Forward {fwd:.2f} ms, backward {bwd:.2f} ms, step {tot:.2f} ms (dropout and LayerDrop on: "
         "a transformer layer that is skipped in a step hands its bucket over in `finish()`).  'complete' = ms after the start of the "
         "backward pass at which the bucket's last gradient existed and it was packed and given to RCCL (mean over the steps in which it "
         f"was).  Modelled all-reduce: {args.latency_us:.0f} us + 2 x 7/8 x bytes / {args.bw:.0f} GB/s (ring over xGMI, {args.world} ranks; "
         "assumption, see the tool's docstring) -- **no 1 -> 8 curve was measured**.  The collectives run on RCCL's own stream, one after "
         "the other: 'done' = max(complete, previous done) + its time; slack = end of backward - done.\n",
         "| bucket | MB | tensors handed over early / all | steps it was complete in backward | complete (ms) | modelled all-reduce (ms) | modelled done (ms) | slack to the end of backward (ms) |", "|---|---|---|---|---|---|---|---|"]
done = 0.0
total_mb, exposed = 0.0, 0.0
for when, name, mb, t_ar in rows:
    total_mb += mb
    start = max(done, when) if when == when else done
    done = start + t_ar
    lines.append(f"| {name} | {mb:.1f} | {early[name][0]:.0f} / {early[name][1]} | {early[name][2]} / {n} | {when:.2f} | {t_ar:.3f} | {done:.2f} | {bwd - done:.2f} |")
exposed = max(0.0, done - bwd)
lines.append(f"| **all** | **{total_mb:.0f}** | | | | **{sum(r[3] for r in rows):.2f}** | {done:.2f} | {bwd - done:.2f} |\n")
lines.append(f"Reading: {total_mb:.0f} MB of gradients per rank take {sum(r[3] for r in rows):.1f} ms of modelled all-reduce inside a {bwd:.1f} ms backward pass; "
             f"the last bucket is done {('%.2f ms AFTER' % exposed) if exposed > 0 else ('%.2f ms before' % (bwd - done))} the backward pass ends"
             f"{'' if exposed > 0 else ' (nothing exposed)'}.  Predicted 8-GPU step = this one-rank step with the collectives live ({tot:.2f} ms) "
             f"+ exposed all-reduce ({exposed:.2f} ms) + the embedding all-gather's latency (~0.05 ms) = **{tot + exposed + 0.05:.1f} ms** -> "
             f"{8 * 64 / (tot + exposed + 0.05) * 1e3:.0f} clip-pairs/s on 8 GPUs, if RCCL's kernels cost the towers no more than the "
             "world = 1 collectives do here (they occupy CUs and HBM beside the trunk; unmeasured).\n")
text = "\n".join(lines)
print(text)
if args.out:
    with open(args.out, "w") as f:
        f.write(text)
dist.destroy_process_group()
