"""Summaries of rocprofv3's rocpd SQLite output (ROCm 7.2 default format).
  python tools/rocpd_summary.py kernels  <results.db>   -> per-kernel stats CSV (as --stats would print)
  python tools/rocpd_summary.py counters <results.db>.. -> per-kernel sums of the collected PMC counters
"""
import collections
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def kernels(path):
    cur = sqlite3.connect(path).cursor()
    rows = collections.defaultdict(list)
    for name, dur in cur.execute("select name, duration from kernels"):
        rows[short(name)].append(dur)
    total = sum(sum(v) for v in rows.values())
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print(f'"{k}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)},{100.0 * sum(v) / total:.2f}')


def counters(paths):
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(dict)
        for name, did, cname, val, dur in cur.execute(
                "select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
            k = short(name)
            agg[k][cname] += val
            disp[k][did] = dur
        print("#", path)
        for k in agg:
            if "bfd::" not in k:
                continue
            print(f"{k:46s} dispatches {len(disp[k]):5d}  total_ms {sum(disp[k].values()) / 1e6:10.3f}  " +
                  "  ".join(f"{c}={v:.6g}" for c, v in sorted(agg[k].items())))


if __name__ == "__main__":
    if sys.argv[1] == "kernels":
        kernels(sys.argv[2])
    else:
        counters(sys.argv[2:])
