"""Per-kernel summary of a rocprofv3 kernel trace: calls, total, mean / median / p90 duration, median gap before the launch.
usage: python scripts/kstats.py <rocprof output dir> [substring ...]"""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
want = sys.argv[2:]
acc, prev = {}, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mvx::", "")
    acc.setdefault(name, []).append(((s - prev) if prev is not None else 0, e - s))
    prev = e
tot = sum(d for g in acc.values() for _, d in g)
print("%-34s %7s %10s %6s %9s %9s %9s %9s" % ("kernel", "calls", "total_us", "%", "mean_us", "med_us", "p90_us", "gap_med"))
for name, g in sorted(acc.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    if want and not any(w in name for w in want):
        continue
    d = sorted(x[1] for x in g)
    print("%-34s %7d %10.1f %6.1f %9.2f %9.2f %9.2f %9.2f" % (name[:34], len(g), sum(d) / 1e3, 100.0 * sum(d) / tot, sum(d) / len(d) / 1e3,
                                                      statistics.median(d) / 1e3, d[int(len(d) * 0.9)] / 1e3, statistics.median(x[0] for x in g) / 1e3))
