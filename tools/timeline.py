"""Print the kernel timeline of the last render recorded by tools/timeline.sh."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/seq/q_results.db").cursor()
rows = list(cur.execute("select name,start,end,duration,grid_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "wf_shade<true" in r[0]][-1]
t0 = rows[idx][1]
for r in rows[idx:]:
    n = r[0].split("(")[0].replace("void bfd::", "")
    if "rocclr" in n:
        continue
    print(f"{(r[1] - t0) / 1e3:9.1f} us  +{r[3] / 1e3:8.1f} us  grid {r[4]:8d}  {n[:44]}")
