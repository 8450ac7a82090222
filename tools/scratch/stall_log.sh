#!/bin/bash
cd $GRAFT_REPO_ROOT
AMD_LOG_LEVEL=3 python tools/scratch/stall_hunt.py sync 32 10 > /tmp/log.txt 2>&1
python - <<'PY'
import re
prev=None; rows=[]; lines=open('/tmp/log.txt', errors='ignore').read().split('\n')
ts=[]
for i,line in enumerate(lines):
    m=re.search(r': (\d{9,}) us:', line)
    ts.append(int(m.group(1)) if m else None)
last=None
for i,t in enumerate(ts):
    if t is None: continue
    if last is not None and t-ts[last] > 8000: rows.append((t-ts[last], last, i))
    last=i
print("log lines", len(lines), "gaps", len(rows))
for d,a,b in rows[-6:]:
    print(f"=== gap {d} us between line {a} and {b}")
    for l in lines[max(0,a-6):b+4]: print("   ", re.sub(r'\x1b\[[0-9;]*m','',l)[:260])
PY
