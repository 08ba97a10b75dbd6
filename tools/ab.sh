#!/bin/bash
# Same-box A/B of bench.py variants (box-to-box spread is +-2-3 %, far above most single changes): runs every variant
# ROUNDS times, interleaved, and prints ms_per_step per run.  GPU box, repo root:
#   tools/ab.sh 3 "" "--set fuse_stage_io=0" "--no-pack-cache"
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    ms=$(python3 bench.py --no-cpu-baseline --no-kernel-timing --no-f32-exact $v 2>>gpurun_out/ab.err | tail -1 | python3 -c "import sys,json; print(json.load(sys.stdin)['ms_per_step'])")
    echo "round $r  [${v:-default}]  $ms ms"
  done
done
