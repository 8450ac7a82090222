#!/bin/bash
# GPU box: kernel-trace summaries of the BASELINE configs -> gpurun_out/trace_summary.txt
set -u
REPO=$(pwd); export TMPDIR=/tmp
: > $REPO/gpurun_out/trace_summary.txt
for CFG in multiscale online batched gaussian gaussian_ms energy; do
  echo "=== config: $CFG  (rocprofv3 --kernel-trace --stats -- python tools/run_config.py $CFG 2)" >> $REPO/gpurun_out/trace_summary.txt
  bash tools/trace_config.sh $CFG 2>&1 | grep -v "^W2026\|^E2026" >> $REPO/gpurun_out/trace_summary.txt
done
cat $REPO/gpurun_out/trace_summary.txt | cut -c1-200
