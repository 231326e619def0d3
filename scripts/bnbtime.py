"""B&B node throughput on one GPU: serial C++ driver vs the window coordinator (batched node solves)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, dist_bnb, synth
from tests import lpgen

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 400
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(m, n, 12345, 3)
root = lpgen.load_ilp(api, A, b, c, U)
bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=20)  # warm-up
ref = None
for window in (1, 8, 32, 64, 128):
    t = time.perf_counter()
    r = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, window=window)
    dt = time.perf_counter() - t
    if ref is None:
        ref = r
    print(json.dumps({"driver": "mvx_branchAndBound (C++)", "window": window, "nodes": r["count"], "pivots": r["total_pivots"],
                      "ms": dt * 1e3, "nodes_per_s": r["count"] / dt, "pivots_per_s": r["total_pivots"] / dt,
                      "same_tree_as_window1": r["events"] == ref["events"] and r["prune"] == ref["prune"]}), flush=True)
eng = dist_bnb.HipNodeEngine(0)
for per_rank in (32, 64):
    t = time.perf_counter()
    got = dist_bnb.branch_and_bound(eng, lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, per_rank=per_rank)
    dt = time.perf_counter() - t
    same = json.loads(json.dumps(got["events"])) == json.loads(json.dumps(ref["events"])) and got["prune"] == ref["prune"]
    print(json.dumps({"driver": "window coordinator", "per_rank": per_rank, "nodes": got["count"], "pivots": got["total_pivots"],
                      "ms": dt * 1e3, "nodes_per_s": got["count"] / dt, "pivots_per_s": got["total_pivots"] / dt,
                      "same_tree_as_serial": same}), flush=True)
