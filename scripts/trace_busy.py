"""GPU busy time per kernel from a rocprofv3 kernel trace (sum of durations, union of intervals)."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:]) for r in csv.DictReader(open(f))))
tot = {}
for s, e, n in rows:
    d = tot.setdefault(n, [0, 0]); d[0] += 1; d[1] += e - s
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
for n, (k, d) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-30s calls=%7d total=%9.2f ms avg=%8.2f us" % (n, k, d / 1e6, d / k / 1e3))
print("union busy %.2f ms of span %.2f ms" % (busy / 1e6, span / 1e6))
