#!/bin/bash
# usage (GPU box): bash tools/trace_seq.sh <tag> <reps> <command ...> ; the ORDERED kernel sequence of the last of <reps> equal
# repetitions of a command's main loop (start offset, duration, gap to the previous kernel's end, short name)
set -u
TAG=$1; REPS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/seq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT -o t -- "$@" > $OUT/log.txt 2>&1
cd $REPO
python - <<PY
import sqlite3, glob, re
for db in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select name, start, end from kernels order by start"))
    except Exception as e:
        print("no kernels view:", e, [r[0] for r in c.execute("select name from sqlite_master")][:40]); continue
    n = len(rows) // int("$REPS")
    last = rows[-n:]
    t0, prev = last[0][1], last[0][1]
    def short(s):
        s = re.sub(r"\(anonymous namespace\)::|glhip::|at::native::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", s)
        return s[:110]
    busy = 0
    for name, s, e in last:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}  {short(name)}")
        prev = e; busy += e - s
    print(f"{n} launches, span {(last[-1][2] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
PY
find $OUT -size +4M -delete
