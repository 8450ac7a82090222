"""Turns the rocprofv3 SQLite outputs written by tools/profile_gpu.sh (under gpurun_out/prof_<tag>/) into the
small text / JSON summaries committed under profiles/.

usage: python tools/summarize_profile.py gpurun_out/prof_r01a r01
"""
import json
import os
import sqlite3
import sys


def rows(db, q):
    c = sqlite3.connect(db)
    cur = c.execute(q)
    cols = [d[0] for d in cur.description]
    return cols, cur.fetchall()


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)

    cols, top = rows(os.path.join(src, "trace", "trace_results.db"), "select * from top_kernels")
    with open(os.path.join(out_dir, f"{tag}_kernel_trace_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-extras   (MI355X, 1 GPU)\n")
        f.write("# durations in microseconds; source: top_kernels view of the rocprofv3 results database\n")
        f.write(" | ".join(cols) + "\n")
        for r in top:
            f.write(" | ".join(str(v) for v in r) + "\n")
    print(open(os.path.join(out_dir, f"{tag}_kernel_trace_stats.txt")).read())

    counters = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc", "pmc_pipe", "pmc_f16", "pmc_util"):
        db = os.path.join(src, sub, "pmc_results.db")
        if not os.path.exists(db):
            continue
        _, rs = rows(db, "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                         "group by kernel_name, counter_name")
        for k, cn, n, avg, dur in rs:
            if "glhip::" in k:
                counters.setdefault(k, {})[cn] = {"launches": n, "avg_per_launch": avg, "avg_kernel_ns": dur}
    summary = {"source": "rocprofv3 --pmc <counter> --kernel-trace (one pass per counter group), bench.py workload",
               "kernels": counters}
    for k, c in sorted(counters.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {}).get("avg_kernel_ns", 0)):
        if "summary_done" in summary:
            break
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            summary["summary_done"] = True
            fetch_kb, write_kb = c["FETCH_SIZE"]["avg_per_launch"], c["WRITE_SIZE"]["avg_per_launch"]
            # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of a
            # wide coalesced stream.  Our reads are 4-12 B per lane (uncalibrated width), so both the raw and the doubled
            # figure are kept; the doubled one is the conservative (upper) estimate used as `traffic`.
            summary["hbm_bytes_per_launch"] = (2 * fetch_kb + write_kb) * 1024
            summary["hbm_bytes_per_launch_uncorrected"] = (fetch_kb + write_kb) * 1024
            summary["kernel"] = k
        if "TCC_HIT_sum" in c:
            h, m = c["TCC_HIT_sum"]["avg_per_launch"], c["TCC_MISS_sum"]["avg_per_launch"]
            summary["l2_hit_rate"] = h / (h + m)
        if "SQ_INSTS_VALU" in c:
            summary["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"]["avg_per_launch"]
        if "GRBM_GUI_ACTIVE" in c:
            g = c["GRBM_GUI_ACTIVE"]
            # GRBM_GUI_ACTIVE counts busy cycles of the graphics clock, summed over the 8 XCDs of the part:
            # cycles / 8 / kernel time = effective clock under this load
            summary["effective_clock_GHz"] = g["avg_per_launch"] / 8.0 / g["avg_kernel_ns"]
        for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_MFMA_BF16",
                     "SQ_INSTS_VALU_MFMA_F16", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "VALUBusy", "MfmaUtil"):
            if name in c:
                summary[name + "_per_launch"] = c[name]["avg_per_launch"]
    path = os.path.join(out_dir, f"{tag}_pmc_softmin.json")
    json.dump(summary, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
