"""First B&B of a process on the wide tree (N nodes): pivots and the engine's debug counters (see MVX_SLAB_FILL / MVX_BIND_FILL)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth, treedigest
api = mvolps_amd.api()
lib = mvolps_amd.load_library()
A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t = time.perf_counter()
r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=n, window=64)
dbg = (C.c_longlong * 8)()
lib.mvx_debug_counters(dbg, 1)
print(os.environ.get("MVX_SLAB_FILL"), os.environ.get("MVX_BIND_FILL"), "nodes", r["count"], "pivots", r["total_pivots"], treedigest.digest(r)[:12],
      "%.0f ms" % ((time.perf_counter() - t) * 1e3), "refreshes/looks/singles/batches/fallbacks", list(dbg)[:5], flush=True)
