"""Phase timeline of the training step WITHOUT a profiler (rocprofv3 adds ~15 us to every launch, which shifts where the
host-bound audio tower starts relative to the trunk: its kernel-trace timeline is distorted; tools/prof_timeline.py).
HIP events are recorded on the stream each phase runs on -- at the entry / exit of the two encoders' forward passes, in
autograd pre / post hooks of the two towers' backward nodes (they fire on the node's own stream), around the loss and the
optimizer -- and read after the run.  ~14 events per step; the step time with and without them is printed.

    python tools/step_timeline.py [--steps 20] [--out profiles/r03_step_timeline.md]
"""
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
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=8)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--out", default=None)
ap.add_argument("--video-only", action="store_true", help="the video tower alone (audio embeddings = constants): its forward / backward durations without the audio tower beside them")
args = ap.parse_args()

import pig.models
from peppa_amd.data import synthetic_batch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = yaml.safe_load(open(os.path.join(root, "hparams_base.yaml")))
cfg["video"]["pretrained"] = cfg["audio"]["pretrained"] = False
torch.manual_seed(0)
net = pig.models.PeppaPig(cfg).cuda().train()
opt = net.configure_optimizers()
b = synthetic_batch(args.batch, 16, 112, 36800).to("cuda")

marks = {}          # name -> [events]
ON = False


def mark(name):
    if ON:
        e = torch.cuda.Event(enable_timing=True)
        e.record()          # on the CURRENT stream (the phase's own)
        marks.setdefault(name, []).append(e)


def find_node(fn, name, seen=None):
    """the autograd node of a tower (VideoTrunkFnBackward / Wav2Vec2FnBackward) below an embedding's grad_fn"""
    seen = set() if seen is None else seen
    if fn is None or fn in seen:
        return None
    seen.add(fn)
    if type(fn).__name__.startswith(name):
        return fn
    for nxt, _ in fn.next_functions:
        hit = find_node(nxt, name, seen)
        if hit is not None:
            return hit
    return None


ep = net.encode_pair


def encode_pair(video, audio):
    """events around the two forward passes: the trunk is launched first (prelaunch), the audio tower on its own stream"""
    mark("video_fwd_begin")
    from peppa_amd import video as PV
    side = PV.tower_stream(video.device)
    V, A = ep(video, audio)
    mark("fwd_joined")                       # trunk stream, after it has waited for the audio stream
    if ON:
        with torch.cuda.stream(side):
            mark("audio_fwd_end")            # audio stream: end of its forward work
        for emb, name, tag in ((V, "VideoTrunkFn", "video"), (A, "Wav2Vec2Fn", "audio")):
            node = find_node(emb.grad_fn, name)
            assert node is not None, name
            node.register_prehook(lambda g, tag=tag: mark(tag + "_bwd_begin"))
            node.register_hook(lambda gi, go, tag=tag: mark(tag + "_bwd_end"))
    return V, A


net.encode_pair = encode_pair
# the trunk's own forward ends where its pooling tail begins: mark inside encode_video (called by encode_pair after the
# audio tower has been launched; the mark lands on the trunk stream behind the prelaunched trunk)
ev = net.encode_video


def encode_video(x):
    mark("video_trunk_fwd_end")
    return ev(x)


net.encode_video = encode_video


if args.video_only:
    A_const = torch.nn.functional.normalize(torch.randn(args.batch, 512, device="cuda"), dim=1)

    def encode_pair_v(video, audio):
        mark("video_fwd_begin")
        from peppa_amd import video as PV
        video2 = pig.models.prelaunch_video_trunk(net.video_encoder, video)
        net.video_encoder._paired = True
        try:
            V = encode_video(video2)
        finally:
            net.video_encoder._paired = False
        mark("fwd_joined")
        if ON:
            mark("audio_fwd_end")
            node = find_node(V.grad_fn, "VideoTrunkFn")
            def pre(g):
                mark("video_bwd_begin"); mark("audio_bwd_begin"); mark("audio_bwd_end")
            node.register_prehook(pre)
            node.register_hook(lambda gi, go: mark("video_bwd_end"))
        return V, A_const

    net.encode_pair = encode_pair_v


def step(i):
    mark("step_begin")
    opt.zero_grad(set_to_none=True)
    loss = net.training_step(b, i)
    mark("loss_fwd_end")
    loss.backward()
    mark("backward_end")          # main stream: everything the optimizer waits for
    opt.step()
    mark("step_end")


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(args.warmup)
plain = run(args.steps)
ON = True
timed = run(args.steps)
ON = False
n = args.steps
names = ["step_begin", "video_fwd_begin", "video_trunk_fwd_end", "audio_fwd_end", "fwd_joined", "loss_fwd_end", "video_bwd_begin",
         "audio_bwd_begin", "audio_bwd_end", "video_bwd_end", "backward_end", "step_end"]
for k in names:
    assert len(marks.get(k, [])) == n, (k, len(marks.get(k, [])))
rel = {k: sum(marks["step_begin"][i].elapsed_time(marks[k][i]) for i in range(n)) / n for k in names}
lines = [f"# Phase timeline of one training step, HIP events on the phases' own streams (no profiler)", "",
         f"hparams_base, batch {args.batch}, bf16, {n} steps after {args.warmup} warm-up: {plain:.2f} ms/step without the events, "
         f"{timed:.2f} ms/step with them.  Times in ms after the step's first launch (mean over the steps).", "",
         "| event (stream) | ms |", "|---|---|"]
stream_of = {"video": "trunk stream", "audio": "audio stream", "loss": "trunk stream", "backward": "trunk stream", "step": "trunk stream",
             "fwd": "trunk stream"}
for k in names:
    lines.append(f"| {k} ({stream_of[k.split('_')[0]]}) | {rel[k]:.2f} |")
d = lambda a, b: rel[b] - rel[a]
lines += ["", "| phase | ms | reading |", "|---|---|---|",
          f"| video trunk forward (trunk stream) | {d('video_fwd_begin', 'video_trunk_fwd_end'):.2f} | |",
          f"| audio forward (audio stream, beside it) | {rel['audio_fwd_end'] - rel['video_fwd_begin']:.2f} | ends {rel['audio_fwd_end'] - rel['video_trunk_fwd_end']:+.2f} ms relative to the trunk's forward |",
          f"| heads, join of the streams, loss forward | {d('video_trunk_fwd_end', 'loss_fwd_end'):.2f} | |",
          f"| loss backward + heads backward until the towers' backward passes start | {max(rel['video_bwd_begin'], rel['audio_bwd_begin']) - rel['loss_fwd_end']:.2f} | video at {rel['video_bwd_begin']:.2f}, audio at {rel['audio_bwd_begin']:.2f} |",
          f"| video backward (trunk stream; weight gradients included) | {d('video_bwd_begin', 'video_bwd_end'):.2f} | |",
          f"| audio backward (audio stream) | {d('audio_bwd_begin', 'audio_bwd_end'):.2f} | ends {rel['audio_bwd_end'] - rel['video_bwd_end']:+.2f} ms relative to the video backward |",
          f"| end of both backward passes -> optimizer done | {rel['step_end'] - max(rel['video_bwd_end'], rel['audio_bwd_end']):.2f} | BertAdam + its norms |",
          f"| **step** | {rel['step_end']:.2f} | |"]
text = "\n".join(lines)
print(text)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write(text + "\n")
