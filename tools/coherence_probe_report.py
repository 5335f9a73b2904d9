import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bf_trace_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
names = ["primary random", "primary sorted by direction", "secondary random", "secondary 15-bit origin + octant", "secondary 30-bit origin",
         "secondary direction cell major, origin minor"]
for kind in ("primary", "secondary", "shadow-like (closest-hit kernel)"):
    names += [kind + " random"] + [kind + " 18-bit " + v for v in ("dir6_pos12", "oct3_pos15", "pos15_oct3")]
for (t, ms), n in zip(rows, names):
    print(f"{n:48s} {ms:8.3f} ms")
