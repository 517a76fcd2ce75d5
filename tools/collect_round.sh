#!/bin/bash
# Everything the round's profiles/ directory is made from, on the GPU box:   bash tools/collect_round.sh gpurun_out/r4/final r04 "Round 4"
# (bench lines, rocprofv3 --stats of bench.py, the per-shape / traffic / family / timeline tables, the step's HBM account, the
# reference-geometry bench + per-shape table).  Copy what should be judged from D into profiles/ afterwards.
set -e
D=$1; R=${2:-r04}; TITLE=${3:-Round 4}
mkdir -p $D
export TMPDIR=/tmp
python bench.py > $D/${R}_bench_default.json 2> $D/bench_default.err
echo "bench: $(cut -c1-160 $D/${R}_bench_default.json)"
STEP_MS=$(python -c "import json,sys; print(json.load(open('$D/${R}_bench_default.json'))['ms_per_step'])")
rocprofv3 --kernel-trace --stats --output-format csv -d $D/bench_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $D/bench_prof.json 2> $D/bench_prof.err
cp $(ls $D/bench_stats/*/*_kernel_stats.csv | head -1) $D/${R}_bench_kernel_stats.csv
python tools/prof_summary.py $D/${R}_bench_kernel_stats.csv 8 $D/${R}_bench_kernel_stats.md "$TITLE: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline (2 warm-up + 5 timed + 1 isolated step = 8 training steps, plus the forward-only passes; durations include cross-stream overlap; rocprofv3 aggregates by kernel TEMPLATE -- per-shape rows: ${R}_shapes.md; the same box ran bench.py without the profiler: ${R}_bench_default.json)"
TRAFFIC=1 bash tools/collect_profiles.sh $D/prof
T_IN=$(ls $D/prof/instep/*/*_kernel_trace.csv | head -1); T_AL=$(ls $D/prof/alone/*/*_kernel_trace.csv | head -1)
python tools/prof_shapes.py $T_IN $D/prof/instep_launch.json $D/${R}_shapes.md "$TITLE: every matrix-core launch of the hparams_base step by SHAPE (B = 64, bf16)" --alone $T_AL $D/prof/alone_launch.json --pmc $D/prof/pmc_sq --pmc-log $D/prof/pmc_launch.json
python tools/prof_traffic.py $D/prof/pmc_fetch $D/prof/pmc_write $D/prof/traffic_launch.json $D/${R}_pmc_traffic.md $D/${R}_traffic.json
python tools/prof_step_traffic.py $D/prof/pmc_fetch $D/prof/pmc_write 3 $STEP_MS $D/${R}_step_hbm_account.md $D/${R}_step_hbm_account.json
python tools/prof_families.py $T_IN $T_AL 9 $D/${R}_families_alone_vs_instep.md "$TITLE: kernel families alone vs in-step (hparams_base, B=64, bf16; 3 warm-up + 6 timed steps, the 6 timed ones counted)" 3
python tools/prof_timeline.py $T_IN $D/${R}_rocprof_stream_timeline.md "$TITLE: per-stream timeline of the step UNDER rocprofv3 (the profiler adds ~15 us per launch; see ${R}_step_timeline.md for the undistorted phases)" --steps 4
python tools/step_timeline.py --out $D/${R}_step_timeline.md > $D/step_timeline.log 2>&1
python tools/step_timeline.py --video-only --out $D/${R}_step_timeline_video_only.md > $D/step_timeline_vo.log 2>&1
[ -n "$SKIP_CONFIGS" ] && { ls $D; exit 0; }
# the reference's own clip geometry (hparams_base.yaml:9,14-16,23: 23 x 100 x 180 frames, 101 429 samples = 316 frames): batch 64 and its micro-batch 8
python bench.py --frames 23 --size 100x180 --samples 101429 --audio-rate 44100 --no-cpu-baseline > $D/${R}_bench_refshape.json 2> $D/bench_refshape.err || true
python bench.py --batch 8 --frames 23 --size 100x180 --samples 101429 --audio-rate 44100 --no-cpu-baseline > $D/${R}_bench_refshape_b8.json 2> $D/bench_refshape_b8.err || true
SKIP_PMC=1 bash tools/collect_profiles.sh $D/prof_ref --frames 23 --size 100x180 --samples 101429 || true
T_INR=$(ls $D/prof_ref/instep/*/*_kernel_trace.csv | head -1); T_ALR=$(ls $D/prof_ref/alone/*/*_kernel_trace.csv | head -1)
python tools/prof_shapes.py $T_INR $D/prof_ref/instep_launch.json $D/${R}_shapes_refshape.md "$TITLE: every matrix-core launch of the step at the REFERENCE's own clip geometry (23 x 100 x 180 frames, 316 audio frames; B = 64, bf16)" --alone $T_ALR $D/prof_ref/alone_launch.json || true
python bench.py --config hparams_jitter.yaml --frames 32 --samples 73600 --dtype fp16 --no-cpu-baseline > $D/${R}_bench_configs4.json 2> $D/bench_c4.err || true
python bench.py --config hparams_freeze_wav2vec.yaml --no-cpu-baseline > $D/${R}_bench_configs2.json 2> $D/bench_c2.err || true
python bench.py --dtype fp16 --no-cpu-baseline > $D/${R}_bench_fp16.json 2> $D/bench_fp16.err || true
ls $D
