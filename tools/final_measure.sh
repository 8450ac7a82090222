#!/bin/bash
# GPU box: every measurement artefact of a round in one call.  usage: bash tools/final_measure.sh <tag>
set -u
TAG=${1:-r01}
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
bash tools/profile_gpu.sh $TAG > /dev/null 2>&1
bash tools/trace_all.sh > /dev/null 2>&1
timeout 300 python tools/accuracy_report.py > gpurun_out/accuracy_report.txt 2>&1
timeout 300 python tools/fuzz_kernels.py 2000 3 > gpurun_out/fuzz.txt 2>&1
timeout 200 python tools/small_probe.py > gpurun_out/small_probe.txt 2>&1
timeout 200 python tools/microbench.py --sizes 100000,1000000 > gpurun_out/microbench.txt 2>&1
timeout 100 python tools/microbench.py --what batched > gpurun_out/microbench_batched.txt 2>&1
timeout 500 python tools/reference_protocol_bench.py > gpurun_out/reference_protocol.txt 2>&1
tail -c 400 gpurun_out/bench_$TAG.json
