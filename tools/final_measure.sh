#!/bin/bash
# GPU box: the measurement artefacts of a round in one call (~10 GPU-minutes).  usage: bash tools/final_measure.sh <tag>
# Copy what should be judged from gpurun_out/ into profiles/ afterwards.
# Every leg's exit status is checked: a leg that fails (a traceback, a timeout) is named on stdout and in
# gpurun_out/final_measure_<tag>.status, its output file is renamed *.FAILED so that it cannot be mistaken for a report,
# and the script exits non-zero.
set -u
TAG=${1:-r02}
mkdir -p gpurun_out
STATUS=gpurun_out/final_measure_$TAG.status
: > $STATUS
FAIL=0

leg() {   # leg <name> <output file | -> <timeout s> <command...>     (SKIP="name name ..." in the environment leaves legs out)
    local name=$1 out=$2 limit=$3
    shift 3
    case " ${SKIP:-} " in *" $name "*) echo "skipped leg '$name'" >> $STATUS; return;; esac
    if [ "$out" = "-" ]; then
        timeout "$limit" "$@" > /dev/null 2> gpurun_out/$name.err
    else
        timeout "$limit" "$@" > "$out" 2> gpurun_out/$name.err
    fi
    local rc=$?
    if [ $rc -ne 0 ]; then
        echo "FAILED leg '$name' (exit $rc): $*" | tee -a $STATUS
        tail -n 5 gpurun_out/$name.err | sed 's/^/    /' | tee -a $STATUS
        [ "$out" != "-" ] && [ -e "$out" ] && mv "$out" "$out.FAILED"
        FAIL=1
    else
        echo "ok     leg '$name'" >> $STATUS
    fi
}

leg bench gpurun_out/bench_$TAG.json 900 python bench.py --steps 20 --warmup 5
leg profile_gpu - 900 bash tools/profile_gpu.sh $TAG                    # kernel trace + PMC passes of the headline command
leg summarize_profile - 200 python tools/summarize_profile.py gpurun_out/prof_$TAG $TAG
cp profiles/${TAG}_kernel_trace_stats.txt profiles/${TAG}_pmc_softmin.json gpurun_out/ 2>/dev/null      # (written under profiles/ on the box: only gpurun_out/ comes back)
leg profile_kernels - 900 bash tools/profile_kernels.sh $TAG            # kernel trace + PMC of every reduction at 1e6
leg trace_all - 900 bash tools/trace_all.sh                             # per-config kernel traces
leg accuracy_report gpurun_out/accuracy_report.txt 300 python tools/accuracy_report.py
leg fuzz gpurun_out/fuzz.txt 200 python tools/fuzz_kernels.py 600 3
leg fuzz_highd gpurun_out/fuzz_highd.txt 200 python tools/fuzz_highd.py 300 3
leg fuzz_sparse gpurun_out/fuzz_sparse.txt 300 python tools/fuzz_sparse.py 300 3          # block-sparse forward (carried tiles, both layouts) and gradient
leg small_probe gpurun_out/small_probe.txt 200 python tools/small_probe.py
leg multiscale_pmc gpurun_out/multiscale_pmc.txt 600 env MIN_NS=5e6 PAIRS=2.1e11 bash tools/profile_kernels.sh ${TAG}_ms multiscale_1e6.py   # the block-sparse kernels of config 3
leg raw_p1 gpurun_out/raw_p1_1e6.txt 200 python tools/raw_p1_1e6.py             # the self-sorting distance reductions through the raw C-ABI
leg bench_f64 gpurun_out/bench_f64.txt 200 python tools/bench_f64.py
leg sparse_ideal gpurun_out/sparse_ideal.txt 200 python tools/probe_sparse_ideal.py
leg first_call gpurun_out/first_call.txt 100 python tools/first_call.py
leg reference_protocol gpurun_out/reference_protocol.log 500 python tools/reference_protocol_bench.py --quick
cat $STATUS
[ -e gpurun_out/bench_$TAG.json ] && tail -c 400 gpurun_out/bench_$TAG.json
exit $FAIL
