"""How well do the launches of several handles overlap?  Reads the kernel trace (rocprofv3 --kernel-trace, rocpd sqlite) of a bench run
and prints, for the timed region (the last dense run of launches), the GPU-busy time, the sum of kernel durations and
their ratio (average number of kernels in flight), plus the share of wall time with 0 / 1 / 2 / 3+ kernels in flight and the gaps.
usage: python tools/concurrency_probe.py <results.db>"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name,start,end from kernels order by start"))
rows = [r for r in rows if "bfd::" in r[0]]
# the timed region is the last dense run of launches (the instrumented passes before it run one handle, serially; gaps between
# the passes are host work): walk back from the end while consecutive launches start within 300 us of the previous one's end
i = len(rows) - 1
while i > 0 and rows[i][1] - rows[i - 1][2] < 300e3:
    i -= 1
rows = rows[i:]
ev = []
for n, s, e in rows:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
    depth += d
    last = t
wall = rows[-1][2] - rows[0][1]
busy = wall - hist.get(0, 0)
tot = sum(e - s for _, s, e in rows)
print("kernels %d  wall %.3f ms  busy %.3f ms (%.1f %%)  sum of durations %.3f ms  -> %.2f kernels in flight while busy" % (
    len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, tot / 1e6, tot / busy))
print("share of wall time with 0 / 1 / 2 / 3+ kernels in flight: " + " / ".join("%.1f %%" % (100.0 * hist.get(k, 0) / wall) for k in range(4)))
by = {}
for n, s, e in rows:
    k = n.split("(")[0].replace("void ", "").replace("bfd::", "")[:28]
    a = by.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
for k, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-30s %5d launches  avg %8.1f us" % (k, c, d / c / 1e3))
