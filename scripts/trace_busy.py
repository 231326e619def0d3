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

# the timed run alone: from the last root solve on (scripts/bnbtrace.py runs a 64-node warm-up first; the root LP of
# each run is one k_persist / k_fb sequence, the B&B proper follows it)
marks = [s for s, e, n in rows if "k_persist" in n] or [s for s, e, n in rows if "k_fboot" in n]
if marks:
    sel = [r for r in rows if r[0] >= marks[-1]]
    busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
    for s, e, n in sel[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = sel[-1][1] - sel[0][0]
    print("timed run only (from its root solve on): union busy %.2f ms of span %.2f ms = %.2f" % (busy / 1e6, span / 1e6, busy / span))
