#!/usr/bin/env python3
"""Instruction-class counts and register use of one kernel in a hipcc -S listing (developer tool).
usage: asm_stats.py listing.s <substring of the mangled kernel name> ..."""
import re
import sys

CLASSES = ["v_", "s_load", "global_load", "global_store", "global_atomic", "scratch_", "ds_", "v_readlane", "v_writelane",
           "v_div_scale", "v_div_fixup", "v_rcp", "v_sqrt", "s_waitcnt", "s_waitcnt vmcnt", "s_waitcnt lgkmcnt", "s_cbranch"]


def main():
    lines = open(sys.argv[1]).read().split("\n")
    for key in sys.argv[2:]:
        start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and "; @" in l))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = [l.strip() for l in lines[start:end]]
        meta = {}
        for l in lines[end:end + 80]:
            m = re.search(r"\.(num_vgpr|numbered_sgpr|private_seg_size), (\d+)", l)
            if m:
                meta[m.group(1)] = int(m.group(2))
            m = re.search(r"; (ScratchSize|Occupancy|NumVgprs|SGPRBlocks|LDSByteSize|codeLenInByte).*?: (\d+)", l)
            if m:
                meta[m.group(1)] = int(m.group(2))
        print(key, meta)
        print("  " + "  ".join(f"{c}={sum(1 for l in body if l.startswith(c))}" for c in CLASSES))


if __name__ == "__main__":
    main()
