"""Summarise a rocprofv3 kernel trace: per-kernel median duration and median gap before each launch."""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
acc = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-24:]
    if prev is not None:
        acc.setdefault(name, []).append((s - prev, e - s))
    prev = e
for name, g in acc.items():
    if len(g) >= 5:
        print("%-26s n=%5d  median_gap_before=%7.0f ns  median_dur=%8.0f ns  p90_dur=%8.0f" % (
            name, len(g), statistics.median(x[0] for x in g), statistics.median(x[1] for x in g),
            sorted(x[1] for x in g)[int(len(g) * 0.9)]))
