#!/bin/bash
# which ingredient of ln_bwd_kernel makes its dx differ between launches inside the step?  (rebuilds norm.o per variant)
set -e
cd "$(dirname "$0")/../.."
objs=$(ls peppa_amd/build/*.o | grep -v "/norm.o")
cp peppa_amd/libpeppa_hip.so /tmp/peppa_keep.so
for v in ${VARIANTS:-0 1 2 4 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPP_LN_VARIANT=$v $EXTRA -c peppa_amd/csrc/norm.hip -o /tmp/norm_v.o 2>&1 | grep -i " error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o peppa_amd/libpeppa_hip.so $objs /tmp/norm_v.o
  echo "== variant $v: concurrent dense wgrad: $(LN_BWD_ALONE=0 CASE='dense wgrad' ITERS=200 python tools/probe/ln_vs_kernels.py 2>/dev/null | grep 'dense wgrad')"
  [ -n "$NO_STEP" ] && continue
  echo "== variant $v: reps with a differing launch: $(REPS=${REPS:-20} python tools/probe/det_ln_twice.py 2>/dev/null | grep "^rep" | grep -vc ': \[\]') of ${REPS:-20}"
done
cp /tmp/peppa_keep.so peppa_amd/libpeppa_hip.so
