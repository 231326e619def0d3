"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per k_fb launch.

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
(16 B per lane), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
usage: pmc_traffic.py <fetch_dir> <write_dir> <m> <n> <out.json> [kernel-name substring, default k_fb]
"""
import csv, glob, json, statistics, sys

KERNEL = sys.argv[6] if len(sys.argv) > 6 else "k_fb"


def per_launch(d, counter):
    f = (glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    vals = []
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    # steady state: drop the bootstrap (gather-only) launches, which move almost nothing
    big = [v for v in vals if v > 0.5 * max(vals)]
    return statistics.median(big), len(big)

fetch_kb, nf = per_launch(sys.argv[1], "FETCH_SIZE")
write_kb, nw = per_launch(sys.argv[2], "WRITE_SIZE")
m, n = int(sys.argv[3]), int(sys.argv[4])
alg = 16 * (m + 1) * (n + 1)
out = {
    "m": m, "n": n, "kernel": KERNEL,
    "FETCH_SIZE_KiB_median": fetch_kb, "WRITE_SIZE_KiB_median": write_kb, "launches": [nf, nw],
    "read_bytes": 2.0 * fetch_kb * 1024.0, "write_bytes": write_kb * 1024.0,
    "bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0,
    "algorithmic_bytes": alg,
    "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), both counters in KiB; separate --pmc passes",
}
out["traffic_over_algorithmic"] = out["bytes_per_launch"] / alg
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps(out))
