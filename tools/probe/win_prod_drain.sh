#!/bin/bash
# temporal window kernels, producer form: do the producer waves wait for EVERY window in flight at a tile's first phase
# (PP_WIN_PROD_DRAIN=1, rounds 2-3: "a tile's first waits also cover the previous epilogue's stores" -- but producers store
# nothing) or keep the window of the phase after next in flight across the tile boundary (0)?  Variant library in /tmp.
set -e
cd "$(dirname "$0")/../.."
source tools/probe/variant_lib.sh
for rep in 1 2; do
for v in 1 0; do
  variant_lib igemm_win -DPP_WIN_PROD_DRAIN=$v
  echo "== PP_WIN_PROD_DRAIN=$v"
  CASE="temporal" python tools/bench_gemm.py "fwd dgrad" 2>/dev/null | grep "l1 "
  CASE="stem2" python tools/bench_gemm.py "fwd dgrad" 2>/dev/null
  python tools/probe/bna_tw.py 2>/dev/null | tail -1
done
done
