"""Window B&B on large dense ILPs (tableaux of 67 MB and 268 MB): nodes/s, pivots, device memory in use.
usage: bnbbig.py M N NODES [WINDOW]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mvolps_amd
from mvolps_amd import bnb, synth
from tests import lpgen
m, n, nodes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
window = int(sys.argv[4]) if len(sys.argv) > 4 else 16
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(m, n, 12345, 3)
P = lpgen.load_ilp(api, A, b, c, U)
t = time.perf_counter(); P.simplex(); api.sync(); t_root = time.perf_counter() - t
root_piv = P.it_cnt
t = time.perf_counter()
r = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, window=window)
dt = time.perf_counter() - t
free, total = torch.cuda.mem_get_info()
print(json.dumps({"m": m, "n": n, "window": window, "root_pivots": root_piv, "root_s": t_root, "nodes": r["count"], "pivots": r["total_pivots"],
                  "secs_incl_root": dt, "nodes_per_s_after_root": r["count"] / max(1e-9, dt - t_root), "open_nodes": r["n_nodes"] - r["count"],
                  "device_GiB_in_use": (total - free) / 2**30}), flush=True)
