#!/bin/bash
# A/B of two BUILDS of libpeppa_hip.so on one box: bench.py alternately with tools/probe/prevlib/libpeppa_hip.so (built from
# an earlier commit's sources) and the current library.     [CMD="python tools/probe/x.py"] bash tools/probe/ab_lib.sh [rounds]
# The other build is selected through PEPPA_HIP_LIB: the shipped library is never overwritten.
set -e
cd "$(dirname "$0")/../.."
# build the other library first, e.g.:  git stash; python -c "import __graft_entry__ as g; g.build()"; mkdir -p tools/probe/prevlib;
#   cp peppa_amd/libpeppa_hip.so tools/probe/prevlib/; git stash pop; python -c "import __graft_entry__ as g; g.build()"
[ -f tools/probe/prevlib/libpeppa_hip.so ] || { echo "tools/probe/prevlib/libpeppa_hip.so is missing (see the comment above)"; exit 1; }
for r in $(seq 1 ${1:-3}); do
  for which in prev new; do
    if [ $which = prev ]; then export PEPPA_HIP_LIB=$PWD/tools/probe/prevlib/libpeppa_hip.so PEPPA_ALLOW_EXPERIMENTAL=1; else unset PEPPA_HIP_LIB; fi
    echo -n "$which: "
    if [ -n "$CMD" ]; then $CMD 2>/dev/null | grep -v amdgpu; else
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])"; fi
  done
done
