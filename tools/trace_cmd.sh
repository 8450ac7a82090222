#!/bin/bash
# usage (GPU box): bash tools/trace_cmd.sh <tag> <command ...> ; rocprofv3 --kernel-trace --stats over any command: top kernels
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o t -- "$@" > $OUT/log.txt 2>&1
cd $REPO
python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot/1e3:.2f} ms over {sum(r[1] for r in rows)} launches")      # the view's durations are in microseconds
    for r in rows[:int("${TOP:-12}")]:
        print(f"{r[4]:6.2f}%  calls {r[1]:5d}  avg {r[3]:10.2f} us  {r[0][:150]}")
PY
find $OUT -size +4M -delete
