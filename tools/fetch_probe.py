"""What does rocprofv3's FETCH_SIZE count for THIS engine's access patterns?  (MI355X_MICROARCH.md: the x2
correction is measured for wide coalesced streaming reads only.)  Three kernels read a 64 MiB table of 16-byte rows exactly
once per launch (bf_kernels.hip: bf_gather_probe<MODE>): 0 streaming, 1 scattered rows (every row once, a line's eight rows by
eight different waves), 2 one row per 128-byte line.  Run it directly under the profiler, once per counter set:

  rocprofv3 --pmc FETCH_SIZE -d OUT/f -o c -- python3 tools/fetch_probe.py
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d OUT/r -o c -- python3 tools/fetch_probe.py
  python3 tools/fetch_probe.py --report OUT      -> per pattern: counter value per launch against the known byte counts

Without a profiler it prints the launch times (the implied gather rates)."""
import ctypes as C
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LOG2_ROWS, REPS = 22, 5          # 4 Mi rows x 16 B = 64 MiB
NAMES = {0: "streaming (row g)", 1: "scattered rows (every row once)", 2: "one row per 128-B line"}


def report(root):
    rows = 1 << LOG2_ROWS
    useful = {0: rows * 16, 1: rows * 16, 2: rows // 8 * 16}
    lines = {0: rows // 8, 1: rows // 8, 2: rows // 8}
    agg = {}
    for db in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        for name, did, cname, val in cur.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            if "bf_gather_probe" not in name:
                continue
            mode = int(name.split("bf_gather_probe<")[1][0])
            a = agg.setdefault((mode, cname), [0.0, set()])
            a[0] += val
            a[1].add(did)
    print("table: 2^%d rows x 16 B = %d MiB; per launch:" % (LOG2_ROWS, rows * 16 >> 20))
    for mode in (0, 1, 2):
        print("  %-34s useful %6.1f MiB, %d lines of 128 B touched" % (NAMES[mode], useful[mode] / 2 ** 20, lines[mode]))
        for (m, cname), (v, dids) in sorted(agg.items()):
            if m != mode:
                continue
            per = v / max(1, len(dids))
            if cname == "FETCH_SIZE":
                b = per * 1024.0
                print("      FETCH_SIZE %10.1f KiB = %7.2f MiB raw: %.3f x the useful bytes, %6.1f B per line touched" % (per, b / 2 ** 20, b / useful[mode], b / lines[mode]))
            else:
                print("      %-24s %12.0f per launch = %.3f per line touched" % (cname, per, per / lines[mode]))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    from beifong_amd import capi
    lib = capi.load_library()
    lib.bfdbg_gather_probe.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
    lib.bfdbg_gather_probe.restype = C.c_int
    for mode in (0, 1, 2):
        ms = C.c_float(0)
        capi.check(lib, lib.bfdbg_gather_probe(mode, LOG2_ROWS, REPS, C.byref(ms)), "bfdbg_gather_probe")
        rows = (1 << LOG2_ROWS) // (8 if mode == 2 else 1)
        print("%-34s %8.4f ms per launch: %7.1f GB/s of useful bytes, %7.1f GB/s of touched 128-B lines" % (
            NAMES[mode], ms.value, rows * 16 / ms.value / 1e6, (1 << LOG2_ROWS) * 16 / ms.value / 1e6), flush=True)


if __name__ == "__main__":
    main()
