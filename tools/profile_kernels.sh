#!/bin/bash
# GPU box: kernel-trace stats + PMC passes (one counter group per pass, no other trace domain) over tools/kernels_1e6.py.
# Output: gpurun_out/prof_kernels_<tag>/summary.txt ; copy into profiles/.
set -u
# usage: tools/profile_kernels.sh <tag> [script under tools/, default kernels_1e6.py]
TAG=${1:-r03}; SCRIPT=${2:-kernels_1e6.py}; REPO=$(pwd); OUT=$REPO/gpurun_out/prof_kernels_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $REPO/tools/$SCRIPT"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc VALUBusy MfmaUtil --kernel-trace -d $OUT/pmc_util -o pmc -- $CMD > $OUT/pmc_util.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_inst -o pmc -- $CMD > $OUT/pmc_inst.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_wait -o pmc -- $CMD > $OUT/pmc_wait.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_pipe -o pmc -- $CMD > $OUT/pmc_pipe.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
cd $REPO
python - <<PY > $OUT/summary.txt
import sqlite3, glob, collections
print("# rocprofv3 over tools/$SCRIPT (N = M = 1e6, fp32; 2 launches of each reduction); MI355X")
for db in glob.glob("$OUT/trace/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    print("## --kernel-trace --stats (top_kernels): calls, average us, % of GPU time")
    for r in list(c.execute("select name, total_calls, average, percentage from top_kernels"))[:16]:
        print(f"{r[1]:5d}  {r[2]:12.1f}  {r[3]:6.2f}%  {r[0][:140]}")
vals = collections.defaultdict(dict)
for sub in ("pmc_util", "pmc_inst", "pmc_wait", "pmc_pipe", "pmc_fetch", "pmc_write"):
    for db in glob.glob("$OUT/" + sub + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        for k, cn, n, avg, dur in c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "glhip::" in k and dur > float("${MIN_NS:-5e6}"):
                vals[k][cn] = avg; vals[k]["_ns"] = dur
print("## PMC per launch (kernels longer than ${MIN_NS:-5e6} ns); pairs per launch = ${PAIRS:-1e12}")
for k, d in sorted(vals.items(), key=lambda kv: -kv[1]["_ns"]):
    ms = d["_ns"] / 1e6
    line = f"{ms:8.2f} ms  " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(d.items()) if n != "_ns")
    if "SQ_INSTS_VALU" in d:
        line += f"  | VALU wave-instr per 64 pairs = {d['SQ_INSTS_VALU'] / (${PAIRS:-1e12} / 64):.2f}"
    if "GRBM_GUI_ACTIVE" in d:
        line += f"  clock = {d['GRBM_GUI_ACTIVE'] / 8.0 / d['_ns']:.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        line += f"  MFMA-busy cycles that co-execute with VALU = {d['SQ_VALU_MFMA_COEXEC_CYCLES'] / d['SQ_VALU_MFMA_BUSY_CYCLES']:.2f}"
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        line += f"  HBM bytes <= {(2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024 / 1e6:.0f} MB"
    print(line); print("            ", k[:160])
PY
cat $OUT/summary.txt | cut -c1-260
find $OUT -size +4M -delete
