#!/bin/bash
# look-ahead (LA=..) / ablation (ABLS="0 1 ..": bits in wgrad_tw.hip) sweep of the narrow temporal sliding-window weight gradient (rebuilds wgrad_tw.o on the GPU box)
set -e
cd "$(dirname "$0")/../.."
objs=$(ls peppa_amd/build/*.o | grep -v wgrad_tw.o)
for np in ${NPRODS:-2}; do
for abl in ${ABLS:-0 1 2 3 4 8 15}; do
  la=${LA:-5}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPP_TW_LA=$la -DPP_TW_ABLATE=$abl -DPP_TW_NPROD=$np -c peppa_amd/csrc/wgrad_tw.hip -o /tmp/wgrad_tw_la.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o peppa_amd/libpeppa_hip.so $objs /tmp/wgrad_tw_la.o
  echo "== look-ahead $la ablate $abl producers $np"
  python tools/probe/tw_narrow.py ${CI:-45} 2>&1 | grep "tw_narrow=1 tw_producers=1" | tail -2
done
done
