#!/bin/bash
# producer-wave split of the window kernel (weight waves of the four producers: PP_WIN_PROD_B) on the spatial shapes
set -e
cd "$(dirname "$0")/../.."
objs=$(ls peppa_amd/build/*.o | grep -v "/igemm_win.o")
cp peppa_amd/libpeppa_hip.so /tmp/peppa_keep.so
for nb in ${SPLITS:-3 2}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPP_WIN_PROD_B=$nb -c peppa_amd/csrc/igemm_win.hip -o /tmp/igemm_win_v.o 2>&1 | grep -i " error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o peppa_amd/libpeppa_hip.so $objs /tmp/igemm_win_v.o
  for wp in 0 2; do
    echo "== weight producers $nb, win_producers $wp"
    WIN_PRODUCERS=$wp CASE="spatial" python tools/bench_gemm.py "fwd dgrad" 2>/dev/null | grep -v "s2"
  done
done
cp /tmp/peppa_keep.so peppa_amd/libpeppa_hip.so
