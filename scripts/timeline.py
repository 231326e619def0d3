"""Merged timeline (kernels + memory copies) of the last solve call in a rocprofv3 trace directory.
usage: timeline.py DIR [N_EVENTS]   (DIR from: rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- ...)"""
import csv, glob, sys
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 70
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
ev.sort()
ev = ev[-last:]
t0 = ev[0][0]
prev = None
for s, e, name in ev:
    print("%10.1f us  +%7.1f gap  %8.1f dur  %s" % ((s - t0) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, (e - s) / 1e3, name))
    prev = e
