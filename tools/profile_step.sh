#!/bin/bash
# rocprofv3 kernel trace + stats of a few bench steps, reduced to the two summaries that get committed under profiles/:
#   <tag>_kernel_stats.csv (rocprofv3 --stats) and <tag>_timeline.txt (tools/trace_report.py over one step).
# usage (on the GPU box, from the repo root): tools/profile_step.sh <tag> [bench.py arguments...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --no-cpu-baseline --no-kernel-timing --no-f32-exact --steps 8 --warmup 3 "$@" > gpurun_out/${tag}_bench.log 2>&1
trace=$(find $out -name "*kernel_trace.csv" | head -1)
stats=$(find $out -name "*kernel_stats.csv" | head -1)
python3 tools/trace_report.py "$trace" median 16 > gpurun_out/${tag}_timeline.txt
cp "$stats" gpurun_out/${tag}_kernel_stats.csv
rm -rf $out
tail -1 gpurun_out/${tag}_bench.log | cut -c1-200
