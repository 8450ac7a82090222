#!/bin/bash
# GPU box: the measurement artefacts of a round in one call (~10 GPU-minutes).  usage: bash tools/final_measure.sh <tag>
# Copy what should be judged from gpurun_out/ into profiles/ afterwards.
set -u
TAG=${1:-r02}
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
bash tools/profile_gpu.sh $TAG > /dev/null 2>&1                       # kernel trace + PMC passes of the headline command
python tools/summarize_profile.py gpurun_out/prof_$TAG $TAG > /dev/null 2>&1
bash tools/profile_kernels.sh $TAG > /dev/null 2>&1                   # kernel trace + PMC of every reduction at 1e6
bash tools/trace_all.sh > /dev/null 2>&1                              # per-config kernel traces
timeout 300 python tools/accuracy_report.py > gpurun_out/accuracy_report.txt 2>&1
timeout 200 python tools/fuzz_kernels.py 600 3 > gpurun_out/fuzz.txt 2>&1
timeout 200 python tools/small_probe.py > gpurun_out/small_probe.txt 2>&1
timeout 100 python tools/probe_batch.py > gpurun_out/probe_batch.txt 2>&1
timeout 100 python tools/first_call.py > gpurun_out/first_call.txt 2>&1
timeout 500 python tools/reference_protocol_bench.py --quick > gpurun_out/reference_protocol.log 2>&1
tail -c 400 gpurun_out/bench_$TAG.json
