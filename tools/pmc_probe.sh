#!/bin/bash
# usage (on the GPU box): bash tools/pmc_probe.sh <tag> "<counter list>" [extra bench args]
set -u
TAG=$1; CTRS=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-extras "$@" > $OUT/log.txt 2>&1
cd $REPO
python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/*.db"):
    c = sqlite3.connect(db)
    for r in c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if 'glhip' in r[0]:
            print(r[0][:70].replace('void glhip::',''), r[1], r[2], f"{r[3]:.4g}", f"{r[4]/1e3:.1f}us")
PY
find $OUT -size +4M -delete
