#!/bin/bash
# which ingredient of ln_bwd_kernel makes its dx differ between launches inside the step?  (rebuilds norm.o per variant
# into a VARIANT library in /tmp; EXTRA="-fslp-vectorize" brings the packed-FP32 forms back for the A/B)
set -e
cd "$(dirname "$0")/../.."
source tools/probe/variant_lib.sh
for v in ${VARIANTS:-0 1 2 4 7}; do
  variant_lib norm -DPP_LN_VARIANT=$v $EXTRA
  echo "== variant $v: concurrent dense wgrad: $(LN_BWD_ALONE=0 CASE='dense wgrad' ITERS=200 python tools/probe/ln_vs_kernels.py 2>/dev/null | grep 'dense wgrad')"
  [ -n "$NO_STEP" ] && continue
  echo "== variant $v: reps with a differing launch: $(REPS=${REPS:-20} python tools/probe/det_ln_twice.py 2>/dev/null | grep "^rep" | grep -vc ': \[\]') of ${REPS:-20}"
done
