"""Per-kernel HBM traffic of the bench configs from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
tools/profile_r04.sh -> the JSON bench.py reads as `roofline.kernels[].traffic` (profiles/r04_pmc_traffic.json), stamped
with a hash of beifong_amd/csrc (bench.py: csrc_hash) so that a kernel change makes it stale instead of silently wrong.
Units and corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB and sit on the L2's fabric side (Infinity-Cache hits
are counted: this is L2-miss traffic, an upper bound of HBM traffic).  FETCH_SIZE = read requests x 64 B while every request
fills a 128-byte line: x 2.  Calibrated in round 4 for THIS engine's access pattern (tools/fetch_probe.py, profiles/
r04_fetch_size_calibration.txt): a streaming read, a scattered 16-byte-per-lane gather of every row of a 64 MiB table and
one row per line all report exactly one request (64 B counted, no 32-byte requests) per L2 miss, so the factor of the
streaming case applies to the scattered gathers of wf_trace / wf_shade too.  WRITE_SIZE is exact."""
import collections
import glob
import json
import os
import sqlite3
import sys

root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
CLASS = (("wf_trace", "wf_trace"), ("wf_shade", "wf_shade"), ("bf_render_kernel", "tail"))
out = {}
for cfg in ("c2", "c3", "c4shard", "c4", "c5"):
    per = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {}})
    found = False
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for db in glob.glob(os.path.join(root, "pmc_%s_%s" % (cfg, ctr), "**", "*.db"), recursive=True):
            found = True
            cur = sqlite3.connect(db).cursor()
            for name, did, cname, val in cur.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
                if "wf_trace<true" in name or "bf_render_kernel<true" in name:
                    continue                      # the BF_FLAG_STATS variants of bench.py's counter pass: not the product kernels
                for key, cls in CLASS:
                    if key in name and cname == ctr:
                        per[cls][ctr] += val
                        per[cls]["n"].setdefault(ctr, set()).add(did)
    if not found:
        continue
    try:
        with open(os.path.join(root, "pmc_%s_FETCH_SIZE.json" % cfg)) as f:
            line = json.loads(f.read().strip().splitlines()[-1])
        paths = line["config"]["paths_per_gpu_per_step"] // (64 if cfg == "c5" else 1)
        rolling = bool(line["config"].get("rolling"))
        steps = int(line["steps"])
    except Exception:
        paths, rolling, steps = None, False, 0
    e = {"paths": paths, "rolling": rolling, "steps": steps, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --config %s --steps 4 --warmup 0 --no-cpu --streams 1" % cfg,
         "fetch_correction": 2.0, "traffic_bytes_per_launch": {}, "dispatches": {}, "fetch_kib_sum": {}, "write_kib_sum": {}}
    for cls, d in per.items():
        nf, nw = len(d["n"].get("FETCH_SIZE", ())), len(d["n"].get("WRITE_SIZE", ()))
        if not nf or not nw:
            continue
        e["traffic_bytes_per_launch"][cls] = round(2.0 * d["FETCH_SIZE"] * 1024 / nf + d["WRITE_SIZE"] * 1024 / nw)
        e["dispatches"][cls] = nf
        e["fetch_kib_sum"][cls] = d["FETCH_SIZE"]
        e["write_kib_sum"][cls] = d["WRITE_SIZE"]
    out[cfg] = e
out["csrc_sha16"] = bench.csrc_hash()
print(json.dumps(out, indent=1))
