#!/bin/bash
# one-off: which runtime knob moves the one-time host stall of the 10-call loop
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python tools/scratch/stall_hunt.py sync 32 10 2>&1 | grep -v amdgpu.ids | tail -1 | tr ' ' '\n' | awk -F/ 'NF==3 && $1>3.0 {printf "  stall block: %s\n", $0; n++} END {if(!n) print "  no stall"}'; }
run A=1
run A=1
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=0
run HSA_KERNARG_POOL_SIZE=33554432
run HSA_KERNARG_POOL_SIZE=33554432
run ROC_SIGNAL_POOL_SIZE=4096
run ROC_SIGNAL_POOL_SIZE=4096
run DEBUG_CLR_MAX_BATCH_SIZE=1000000
run GPU_MAX_HW_QUEUES=1
run HIP_FORCE_DEV_KERNARG=0
run ROC_AQL_QUEUE_SIZE=65536
AMD_LOG_LEVEL=3 python tools/scratch/stall_hunt.py sync 32 10 > /tmp/log.txt 2>&1
python - <<'PY'
import re
prev=None; rows=[]
for line in open('/tmp/log.txt', errors='ignore'):
    m=re.search(r'ts:\s*(\d+)', line) or re.search(r'\[(\d+\.\d+) us\]', line)
    if not m: continue
    t=float(m.group(1))
    if prev is not None and t-prev[0] > 5000: rows.append((t-prev[0], prev[1][:300], line[:300]))
    prev=(t,line)
print("log lines", sum(1 for _ in open('/tmp/log.txt', errors='ignore')))
for d,a,b in rows[-12:]: print(f"gap {d:.0f} us\n   BEFORE: {a.strip()}\n   AFTER:  {b.strip()}")
PY
head -c 3000 /tmp/log.txt | tail -c 1500
