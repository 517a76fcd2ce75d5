#!/bin/bash
# look-ahead (LA=..) / ablation (ABLS="0 1 ..": bits in wgrad_tw.hip) sweep of the narrow temporal sliding-window weight
# gradient (rebuilds wgrad_tw.o on the GPU box into a VARIANT library in /tmp; the shipped library is not touched)
set -e
cd "$(dirname "$0")/../.."
source tools/probe/variant_lib.sh
for np in ${NPRODS:-2}; do
for abl in ${ABLS:-0 1 2 3 4 8 15}; do
  la=${LA:-5}
  variant_lib wgrad_tw -DPP_TW_LA=$la -DPP_TW_ABLATE=$abl -DPP_TW_NPROD=$np
  echo "== look-ahead $la ablate $abl producers $np"
  python tools/probe/tw_narrow.py ${CI:-45} 2>&1 | grep "tw_narrow=1 tw_producers=1" | tail -2
done
done
