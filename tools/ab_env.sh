#!/bin/bash
# Same-box A/B of environment settings (e.g. HIP runtime flags) for bench.py: tools/ab_env.sh ROUNDS "" "VAR=1" "VAR=2 OTHER=3" ...
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    ms=$(env $v python3 bench.py --no-cpu-baseline --no-kernel-timing --no-f32-exact --mode graph 2>>gpurun_out/ab.err | tail -1 | python3 -c "import sys,json; print(json.load(sys.stdin)['ms_per_step'])")
    echo "round $r  [${v:-default}]  $ms ms"
  done
done
