#!/bin/bash
# The round's PMC evidence, regenerated from scratch (run on the GPU box from the repo root; needs ~3 min):
#   profiles/<tag>_pmc_fetch_write_per_kernel.csv   FETCH_SIZE (calibrated) / WRITE_SIZE per kernel of one eager bench step
#   profiles/<tag>_pmc_calibration.txt              the calibration on kernels of known traffic (tools/pmc_calibrate.py)
#   profiles/<tag>_pmc_mfma_busy_per_kernel.csv     MFMA-pipe utilisation and wave-stall split per kernel
# Counters are collected in their own passes, kernel trace only (no other trace domains).
# usage: tools/pmc_passes.sh <tag> [bench.py arguments...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/pmc_$tag
rm -rf $o; mkdir -p $o
B="python3 bench.py --no-cpu-baseline --no-kernel-timing --no-f32-exact --mode eager --single-stream --steps 1 --warmup 1 --coin-patterns 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/sf -o r -- $B "$@" > $o/sf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/sw -o r -- $B "$@" > $o/sw.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/cf -o r -- python3 tools/pmc_calibrate.py $o/known.json > $o/cf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/cw -o r -- python3 tools/pmc_calibrate.py $o/known.json > $o/cw.log 2>&1
python3 tools/pmc_report.py $o/sf $o/sw $o/cf $o/cw $o/known.json gpurun_out/${tag}_pmc_fetch_write_per_kernel.csv gpurun_out/${tag}_pmc_calibration.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $o/mf -o r -- $B "$@" > $o/mf.log 2>&1
python3 tools/pmc_mfma_report.py $o/mf gpurun_out/${tag}_pmc_mfma_busy_per_kernel.csv
cat gpurun_out/${tag}_pmc_calibration.txt
rm -rf $o/sf $o/sw $o/cf $o/cw $o/mf
