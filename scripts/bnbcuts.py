"""B&B with GMI cuts on the 512x1024 ILP: node-at-a-time against window mode (same tree), bug-compatible and repaired cuts."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
from tests import lpgen
m, n, nodes = 512, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 600
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(m, n, 12345, 3)
bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=20)
for kw in (dict(quirks=0, cut_strat=1), dict(quirks=0, cut_strat=1, cut_select=1, cut_chance=0.05), dict(quirks=1, cut_strat=1)):
    ref = None
    for window in (1, 64):
        t = time.perf_counter()
        r = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), max_nodes=nodes, window=window, **kw)
        dt = time.perf_counter() - t
        if ref is None: ref = r
        same = all(repr(r[k]) == repr(ref[k]) for k in ("events", "prune", "parent", "count", "total_pivots", "node_bound"))
        print(json.dumps({"mode": kw, "window": window, "nodes": r["count"], "pivots": r["total_pivots"], "ms": dt * 1e3, "nodes_per_s": r["count"] / dt, "same_tree": same}), flush=True)
