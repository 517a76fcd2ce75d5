#!/bin/bash
# A/B of two TREES on one box: bench.py alternately in tools/probe/wt_head (a worktree of the last commit, built there:
#   git worktree add -f tools/probe/wt_head HEAD && (cd tools/probe/wt_head && python -m peppa_amd.build))
# and in the working tree.      bash tools/probe/ab_tree.sh [rounds]
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${1:-3}); do
  for which in head work; do
    d=$PWD; [ $which = head ] && d=$PWD/tools/probe/wt_head
    echo -n "$which: "
    (cd $d && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step')")
  done
done
