#!/bin/bash
# temporal form of the window kernel: do the two weight producers (idle after the resident weights have landed) also issue window pieces (PP_WIN_DUAL)
# -- variant library in /tmp, the shipped library is not touched
set -e
cd "$(dirname "$0")/../.."
source tools/probe/variant_lib.sh
for nb in ${SPLITS:-0 1}; do
  variant_lib igemm_win -DPP_WIN_DUAL=$nb
  echo "== PP_WIN_DUAL=$nb (1: all four producers issue windows in the temporal form)"
  CASE="temporal" python tools/bench_gemm.py "fwd dgrad" 2>/dev/null | grep -v "s2"
  CASE="stem2" python tools/bench_gemm.py "fwd dgrad" 2>/dev/null
done
