#!/bin/bash
# Regenerates the round's judged artifacts under gpurun_out/ (copy to profiles/ afterwards).  GPU box, repo root, ~6 min:
#   tools/round_artifacts.sh r02
set -e
tag=$1
python3 bench.py --dump-launches gpurun_out/${tag}_bench_launch_shapes.txt 2> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_line.json
python3 bench.py --precision f32 --no-cpu-baseline 2>> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_line_f32.json
python3 bench.py --workload config5 --no-cpu-baseline 2>> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_line_config5.json
python3 bench.py --workload frontend --no-cpu-baseline 2>> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_line_frontend.json
echo "bench lines done"
tools/profile_step.sh ${tag}_bf16x6
tools/profile_step.sh ${tag}_f32 --precision f32
echo "kernel traces done"
tools/pmc_passes.sh ${tag}
