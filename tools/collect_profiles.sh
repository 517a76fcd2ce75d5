#!/bin/bash
# Profiles of one round, on the GPU box (run from the repo root):  bash tools/collect_profiles.sh gpurun_out/r3/prof [extra step_loop flags]
# 1. kernel trace of the step as bench.py runs it (streams overlapped) + launch log  -> per-shape table, timeline
# 2. kernel trace of the same step on one stream (every kernel alone)                -> "alone" columns
# 3. PMC passes on the one-stream step (counters in their own runs, --kernel-trace only): SQ wave-cycle split + MFMA busy
set -e
D=$1; shift
mkdir -p $D
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D/instep -- python3 tools/step_loop.py --steps 6 --launch-log $D/instep_launch.json "$@" > $D/instep.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $D/alone -- python3 tools/step_loop.py --steps 6 --isolated --launch-log $D/alone_launch.json "$@" > $D/alone.log 2>&1
if [ -n "$SKIP_PMC" ]; then tail -n 1 $D/instep.log $D/alone.log; exit 0; fi
rocprofv3 -L > $D/counters.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/pmc_sq -- python3 tools/step_loop.py --steps 2 --warmup 1 --isolated --launch-log $D/pmc_launch.json "$@" > $D/pmc_sq.log 2>&1
if [ -n "$TRAFFIC" ]; then   # HBM bytes: FETCH_SIZE and WRITE_SIZE in separate passes (together they exceed the TCC slots)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc_fetch -- python3 tools/step_loop.py --steps 2 --warmup 1 --isolated --launch-log $D/traffic_launch.json "$@" > $D/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc_write -- python3 tools/step_loop.py --steps 2 --warmup 1 --isolated "$@" > $D/pmc_write.log 2>&1
fi
tail -n 1 $D/instep.log $D/alone.log $D/pmc_sq.log
