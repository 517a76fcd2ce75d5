#!/usr/bin/env python
"""Headline benchmark: clip-pairs/sec of the PeppaPig training step (video fwd + audio fwd + triplet
loss + backward + gradient all-reduce + BertAdam) on hparams_base shapes (BASELINE.json configs[1]):
B=64 clips per GPU of 16x112x112 video + 2.3 s @ 16 kHz audio, bf16 compute, synthetic data,
random-init weights.  One process per GPU (torch.distributed / RCCL); weak scaling.

    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import copy
import json
import os
import sys
import time
import warnings

warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

GF_TRAIN_PER_PAIR = 340.0e9      # SURVEY 8d / BASELINE.md: 3 x (81.04 + 32.30) GFLOP at 16x112x112 + 36 800 samples


def train_flops_per_pair(frames, size, samples, frozen_audio=False):
    """Algorithmic FLOPs of one training step per clip pair: r2plus1d_18 scales with
    frames x pixels, wav2vec2 with its frame count (229 frames: 65.76 GF; the quadratic attention term is < 2 %)."""
    T = (samples - 400) // 320 + 1
    H, W = (size, size) if isinstance(size, int) else size
    video = 81.04e9 * (frames / 16.0) * (H * W / 112.0 ** 2)
    audio = {114: 32.30e9, 229: 65.76e9, 49: 13.82e9}.get(T, 32.30e9 * T / 114.0)
    if frozen_audio:      # hparams_freeze_wav2vec: forward + data gradient through the frozen layers (~298 GF at C2)
        return 3 * video + audio + 22.56e9 * (audio / 32.30e9)
    return 3 * (video + audio)


MFMA_BF16_PEAK = 2.5e15          # dense bf16 / fp16 matrix peak, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", default="112", help="frame size H or HxW (the reference's own clips: 100x180, 23 frames, "
                    "101429 audio samples = 2.3 s at 44.1 kHz fed unresampled: hparams_base.yaml:9,14-16, SURVEY 0.8)")
    ap.add_argument("--samples", type=int, default=36800)
    ap.add_argument("--audio-rate", type=int, default=16000, help="only labels the workload (the model sees samples)")
    ap.add_argument("--config", default=os.path.join(ROOT, "hparams_base.yaml"))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit operand type: bf16 (BASELINE configs[1]) or fp16 + dynamic loss scaling (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="no HIP events inside the timed region (roofline: null)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    args = ap.parse_args()
    hw = [int(v) for v in str(args.size).lower().split("x")]
    args.size = hw[0] if len(hw) == 1 or hw[0] == hw[1] else (hw[0], hw[1])
    args.size_name = f"{hw[0]}x{hw[-1]}"
    return args


def cpu_baseline(cfg, args):
    """The CPU oracle (fp32, torch CPU threads) on a bounded sample of the same workload."""
    from oracle import model as O
    # threads actually available to this process: the affinity mask can list every core of the host while the cgroup
    # grants a one-GPU box 16 of them -- more threads than that only oversubscribes (minutes instead of seconds)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, ncpu)))
    print(f"[bench] cpu_baseline: one oracle step at batch {args.cpu_batch} on {torch.get_num_threads()} threads ...",
          file=sys.stderr, flush=True)
    torch.manual_seed(0)
    net = O.PeppaPigOracle(cfg, dropout=0.0, layer_drop=0.0).train()
    v, a = O.synthetic_batch(args.cpu_batch, args.frames, args.size, args.samples)
    params = [p for p in net.parameters()]
    state = {}
    nsteps, t0 = 0, time.perf_counter()
    while True:   # whole training steps until ~10 s of CPU work have been sampled (at most 6 steps)
        for p in params:
            p.grad = None
        loss = net.training_loss(v, a)
        loss.backward()
        with torch.no_grad():
            O.bertadam_step(params, [p.grad for p in params], state, **{k: cfg["optimizer"][k] for k in ("lr", "warmup", "t_total")})
        nsteps += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or nsteps >= 6:
            break
    return {"value": args.cpu_batch * nsteps / dt, "unit": "clip-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nsteps} full training steps (fwd+loss+bwd+BertAdam) of the fp32 CPU oracle at batch {args.cpu_batch}, "
                      f"{args.frames}x{args.size_name} video + {args.samples} audio samples, {dt:.1f} s"}


def main():
    args = parse()
    share_gpu = os.environ.get("PEPPA_BENCH_SHARE_GPU") == "1"     # one-GPU rehearsal of the N-rank path (gloo, all ranks on cuda:0)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called without a launcher: start the ranks ourselves, as CHILD processes, before this process touches the GPU
        # (never a re-exec), and leave with the launcher's exit code.  Fewer visible GPUs than ranks -> refuse.
        from peppa_amd.launch import require_devices, spawn_ranks
        require_devices(args.gpus)
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    # stdout carries exactly ONE line, the JSON record: libraries that print banners there (RCCL's version block when
    # NCCL_DEBUG=VERSION is set in the environment) go to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the record's n_gpus would be wrong")
    if not share_gpu and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("PEPPA_FORCE_DIST") == "1"
    backend = "gloo" if share_gpu and world > 1 else "nccl"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend)   # (no device_id: the eager communicator made every kernel of the step ~10 % slower)

    import yaml
    import pig.models
    from peppa_amd import hip as H
    from peppa_amd.data import synthetic_batch
    from peppa_amd.dist import default_buckets
    cfg = yaml.safe_load(open(args.config))
    cfg["video"]["pretrained"] = False   # weights cannot be downloaded offline: random init, same architecture
    cfg["audio"]["pretrained"] = False
    torch.manual_seed(0)
    net = pig.models.PeppaPig(cfg).to(dev).train()
    net.set_precision(args.dtype)
    scaler = None
    if args.dtype == "fp16":      # what Lightning's native AMP does around BertAdam under the reference's `precision: 16`
        from peppa_amd.amp import GradScaler
        scaler = GradScaler()
    optim = net.configure_optimizers()
    batch = synthetic_batch(args.batch, args.frames, args.size, args.samples, seed=1234 + rank).to(dev)
    buckets = None
    if use_dist:
        buckets = default_buckets(net, dev)

    def step(i):
        optim.zero_grad(set_to_none=True)
        loss = net.training_step(batch, i)
        if scaler is not None:
            scaler.scale(loss).backward()
        else:
            loss.backward()
        if buckets is not None:
            buckets.finish()
        if scaler is not None:
            scaler.step(optim)
            scaler.update()
        else:
            optim.step()
        return loss

    # The last warm-up step times every large GEMM launch to find the dominant kernel family; in the timed region only
    # that family carries HIP events (each timed launch costs two event records on its stream).
    H.PROFILE_STREAM = torch.cuda.current_stream()   # the video trunk's stream; the audio tower overlaps on a side stream
    for i in range(args.warmup):
        H.PROFILE.clear()
        H.PROFILE_ONLY, H.PROFILE_ON = None, (i == args.warmup - 1 and not args.no_roofline)
        step(i)
    dom = H.profile_summary()
    H.PROFILE.clear()
    H.PROFILE_ONLY = dom[3] if dom else None
    H.PROFILE_ON = not args.no_roofline
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    trace = os.environ.get("PEPPA_BENCH_TRACE")     # per-step GPU times on stderr (events; no extra syncs)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if trace else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        if trace:
            marks[i].record()
        loss = step(args.warmup + i)
    if trace:
        marks[args.steps].record()
    torch.cuda.synchronize()
    if trace and rank == 0:
        print("per-step ms:", " ".join(f"{marks[i].elapsed_time(marks[i + 1]):.1f}" for i in range(args.steps)), file=sys.stderr)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    H.PROFILE_ON = False
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    if rank != 0:
        dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt
    cfg_base = os.path.basename(args.config) == "hparams_base.yaml"
    # dominant kernel: HIP events recorded around its launches inside the timed region
    roof = H.profile_summary()
    roof_obj = None
    if roof:
        flops, secs, n, name = roof
        ach = flops / secs / 1e12
        # HBM bytes from the COMMITTED PMC passes under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in their own runs;
        # not measured inside this run -- `traffic_source` says which table): `traffic` = this kernel shape's bytes per
        # launch, `step_hbm_bytes` = every launch of one training step (tools/prof_step_traffic.py), only for the workload
        # the tables were taken on (configs[1], batch 64)
        traffic = step_bytes = src = None
        try:
            for fn in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
                path = os.path.join(ROOT, "profiles", fn)
                if traffic is None and os.path.exists(path):
                    traffic = json.load(open(path)).get(name)
                    src = "profiles/" + fn if traffic is not None else None
            acct = os.path.join(ROOT, "profiles", "r04_step_hbm_account.json")
            if os.path.exists(acct) and (args.frames, args.size, args.samples, args.batch) == (16, 112, 36800, 64) and cfg_base:
                step_bytes = json.load(open(acct))["step_bytes"]
        except Exception:
            pass
        roof_obj = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                    "frac": round(ach * 1e12 / MFMA_BF16_PEAK, 4), "traffic": traffic, "kernel": name,
                    "launches": n, "avg_us": round(secs / n * 1e6, 1)}
        if src:
            roof_obj["traffic_source"] = src + " (committed rocprofv3 PMC table, per launch; not collected in this run)"
        if step_bytes:
            roof_obj["step_hbm_bytes"] = step_bytes
            roof_obj["step_hbm_frac_of_step_at_6.3TBps"] = round(step_bytes / 6.3e12 / (ms * 1e-3), 3)
        # In the timed region this kernel shares the chip with the audio tower and the weight-gradient stream, which
        # stretches its launches.  One extra untimed step with the side streams off gives its duration alone.
        iso = None
        if world == 1:   # (the other ranks of a multi-GPU run have already left)
            from peppa_amd import video as PV
            H.PROFILE.clear()
            H.PROFILE_ON, net._overlap, PV.OVERLAP_WGRAD = True, False, False   # (PROFILE_ONLY still set)
            step(args.warmup + args.steps)
            H.PROFILE_ON, net._overlap, PV.OVERLAP_WGRAD = False, True, True
            iso = H.profile_summary(name)
        if iso:
            roof_obj["isolated"] = {"achieved": round(iso[0] / iso[1] / 1e12, 2), "avg_us": round(iso[1] / iso[2] * 1e6, 1),
                                    "frac": round(iso[0] / iso[1] / MFMA_BF16_PEAK, 4),
                                    "note": "same kernel family, one extra step with stream overlap off"}
    # forward-only throughput (SURVEY 8d asks for it next to the training number): both encoders + loss, no graph, the
    # same batch; rank 0's rate x world (the forward pass has no collective)
    fwd_pairs = None
    if True:
        with torch.no_grad():
            for _ in range(2):
                net.loss(*net.encode_pair(batch.video, batch.audio))
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            nf = max(5, args.steps // 2)
            for _ in range(nf):
                net.loss(*net.encode_pair(batch.video, batch.audio))
            torch.cuda.synchronize()
            fwd_pairs = world * args.batch * nf / (time.perf_counter() - tf0)
    cfg_name = os.path.basename(args.config)
    frozen = bool(cfg["audio"].get("freeze_feature_extractor")) and cfg["audio"].get("freeze_encoder_layers") == 12
    flops_pair = train_flops_per_pair(args.frames, args.size, args.samples, frozen)
    which = {("hparams_base.yaml", 16, 36800): " (BASELINE configs[1])", ("hparams_freeze_wav2vec.yaml", 16, 36800): " (BASELINE configs[2])",
             ("hparams_jitter.yaml", 32, 73600): " (BASELINE configs[4] geometry)"}.get((cfg_name, args.frames, args.samples), "")
    out = {
        "metric": f"clip-pairs/sec (A+V encode + triplet loss), {cfg_name[:-5] if cfg_name.endswith('.yaml') else cfg_name}",
        "value": round(value, 2),
        "unit": "clip-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{cfg_name} training step, {args.frames}x{args.size_name} video + "
                               f"{args.samples / args.audio_rate:.1f} s@{args.audio_rate / 1000:g}kHz audio, batch {args.batch}/GPU{which}",
                   "global_batch": world * args.batch,
                   "parallelism": f"dp{world}" + ("+rccl" if use_dist and world == 1 else "")
                                  + (" (gloo, ranks share cuda:0: rehearsal of the launch path, not a scaling number)" if share_gpu and world > 1 else ""),
                   "gflop_per_pair": round(flops_pair / 1e9, 1),
                   "step_tflops": round(flops_pair * world * args.batch / (ms * 1e-3) / 1e12, 1),
                   "loss": round(float(loss.item()), 5),
                   "forward_only_pairs_per_s": None if fwd_pairs is None else round(fwd_pairs, 1)},
        "roofline": roof_obj,
    }
    if use_dist:
        dist.destroy_process_group()
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, args)
        except Exception as e:  # report, never hide
            out["cpu_baseline"] = {"error": repr(e)}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
