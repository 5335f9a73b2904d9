"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: sum of each
counter over dispatches, dispatch count, total duration."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        dur = collections.defaultdict(float)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:
                disp[k].add(r["Dispatch_Id"])
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        print("#", f)
        for k in agg:
            if "bfd" not in k and "wf_" not in k:
                continue
            print(f"{k:42s} dispatches {len(disp[k]):5d}  total_ms {dur[k]:10.3f}  " +
                  "  ".join(f"{c}={v:.6g}" for c, v in sorted(agg[k].items())))
