"""BASELINE config 5 instance: first 10k nodes of the FIFO tree of the 512x1024 ILP (seed 12345, U=3).
Writes tests/golden/config5.json (node/pivot counts, incumbent, sha256 of the event stream) from a GPU run;
the first 2000 nodes of the same tree are checked against node-at-a-time in scripts/bnbtime.py and the first
nodes against the CPU oracle in tests/test_gpu_bnb.py."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
from tests import lpgen

def digest(r):
    h = hashlib.sha256()
    for e in r["events"]:
        h.update(repr((e[0], e[1], e[2], e[3], float(e[4]).hex(), float(e[5]).hex(), e[6], e[7])).encode())
    h.update(repr(r["prune"]).encode())
    return h.hexdigest()

if __name__ == "__main__":
    api = mvolps_amd.api()
    m, n, seed, U, nodes = 512, 1024, 12345, 3, 10000
    A, b, c, U = synth.dense_ilp(m, n, seed, U)
    bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=64)
    out = {"m": m, "n": n, "seed": seed, "U": U, "cap": 0.4, "max_nodes": nodes, "order": "FIFO", "var_strat": "VO", "reference_quirks": 0}
    r3 = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=3000, window=1)
    out["prefix_3000"] = {"nodes": r3["count"], "pivots": r3["total_pivots"], "sha256": digest(r3), "driver": "node at a time"}
    r3w = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=3000, window=64)
    assert digest(r3w) == out["prefix_3000"]["sha256"]
    for window in (64,):
        t = time.perf_counter()
        r = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, window=window)
        dt = time.perf_counter() - t
        d = digest(r)
        print(json.dumps({"window": window, "nodes": r["count"], "n_oids": r["n_nodes"], "pivots": r["total_pivots"], "secs": dt,
                          "nodes_per_s": r["count"] / dt, "best_lower": r["best_lower"], "sha256": d}), flush=True)
        out.update({"nodes": r["count"], "n_oids": r["n_nodes"], "pivots": r["total_pivots"], "has_incumbent": r["has_incumbent"],
                    "best_lower": r["best_lower"] if r["has_incumbent"] else None, "events": len(r["events"]), "sha256": d})
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "config5.json"), "w"), indent=1)
