"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or a log file).

usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> build.log; python tools/kernel_resources.py build.log [filter]
"""
import re
import subprocess
import sys


def main():
    log = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = log.split("Function Name: ")[1:]
    names = [b.split("\n")[0].strip().split(" ")[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':130s} vgpr sgpr scratch occ lds")
    for b, n in zip(blocks, dem):
        def g(k):
            m = re.search(k + r":\s*(\d+)", b)
            return int(m.group(1)) if m else -1
        n = n.replace("glhip::", "").replace("void ", "")
        if flt and flt not in n:
            continue
        vals = (g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                g(r"LDS Size \[bytes/block\]"))
        print("%-130s %4d %4d %5d %3d %6d" % ((n[:130],) + vals))


if __name__ == "__main__":
    main()
