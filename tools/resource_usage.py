#!/usr/bin/env python3
"""Register / scratch / spill use of every kernel of a .hip file, one line each (developer tool).
usage: tools/resource_usage.py bf_wavefront.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beifong_amd", "csrc")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
           *sys.argv[2:], "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", sys.argv[1], "-o", "/dev/null"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for l in out.split("\n"):
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", l)
        if not m:
            if "error" in l:
                print(l)
            continue
        k, v = m.group(1).split(" [")[0], m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        print("%-58s vgpr %3s sgpr %3s scratch %4s occ %s sgpr-spill %3s vgpr-spill %3s lds %s" % (
            n, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size")))


if __name__ == "__main__":
    main()
