#!/bin/bash
# usage (GPU box): bash tools/trace_config.sh <config> ; prints the top kernels of one end-to-end run
set -u
CFG=$1; REPO=$(pwd); OUT=$REPO/gpurun_out/trace_$CFG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $REPO/tools/run_config.py $CFG 2 > $OUT/log.txt 2>&1
grep -E "rep [0-9]|clusters|Keep|Jump|scales" $OUT/log.txt
cd $REPO
python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot/1e3:.1f} ms over {sum(r[1] for r in rows)} launches")
    for r in rows[:14]:
        print(f"{r[4]:6.2f}%  calls {r[1]:5d}  avg {r[3]:10.1f} us  {r[0][:110]}")
PY
find $OUT -size +4M -delete
