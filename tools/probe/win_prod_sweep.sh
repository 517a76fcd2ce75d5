#!/bin/bash
# producer-wave split of the window kernel (weight waves of the four producers: PP_WIN_PROD_B) on the spatial shapes
# (variant library in /tmp; the shipped library is not touched)
set -e
cd "$(dirname "$0")/../.."
source tools/probe/variant_lib.sh
for nb in ${SPLITS:-3 2}; do
  variant_lib igemm_win -DPP_WIN_PROD_B=$nb
  for wp in 0 2; do
    echo "== weight producers $nb, win_producers $wp"
    WIN_PRODUCERS=$wp CASE="spatial" python tools/bench_gemm.py "fwd dgrad" 2>/dev/null | grep -v "s2"
  done
done
