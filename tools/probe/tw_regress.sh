#!/bin/bash
# which commit slowed the 144-column temporal data gradient?  libraries with igemm_win.o from older commits (tools/probe/prevlib)
cd "$(dirname "$0")/../.."
for r in 1 2; do
  for tag in f057706 433e0d7 head; do
    if [ $tag = head ]; then unset PEPPA_HIP_LIB; else export PEPPA_HIP_LIB=$PWD/tools/probe/prevlib/libpeppa_$tag.so PEPPA_ALLOW_EXPERIMENTAL=1; fi
    echo -n "$tag: "; CASE="l1 temporal" python tools/bench_gemm.py 2>/dev/null | grep -v amdgpu
  done
done
