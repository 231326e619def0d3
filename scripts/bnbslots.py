"""Window B&B throughput against (window, batch slots): how many node LPs share one launch."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
from tests import lpgen
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(m, n, 12345, 3)
bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=20)
ref = None
for window, slots in ((64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512)):
    api.set_batch_slots(slots)
    best = None
    for rep in range(2):
        t = time.perf_counter()
        r = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, window=window)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    if ref is None:
        ref = r
    print(json.dumps({"window": window, "slots": slots, "nodes": r["count"], "pivots": r["total_pivots"], "ms": best * 1e3,
                      "nodes_per_s": r["count"] / best, "pivots_per_s": r["total_pivots"] / best,
                      "same_tree": r["events"] == ref["events"] and r["prune"] == ref["prune"]}), flush=True)
