"""Host + device cost of the per-node operations of the window B&B on a 512x1024 ILP: clone, bound edit,
classification queries (bs.cpp:116,274,282; util.cpp:414-473)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth, capi, bnb
from tests import lpgen
api = mvolps_amd.api()
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
A, b, c, U = synth.dense_ilp(m, n, 12345, 3)
P = lpgen.load_ilp(api, A, b, c, U)
P.simplex(); api.sync()
N = 500
warm = [P.copy() for _ in range(64)]; del warm; api.sync()  # slabs now come from the free list, as in a running B&B
t = time.perf_counter()
for _ in range(N):
    q = P.copy(); del q
api.sync(); t_steady = (time.perf_counter() - t) / N
t = time.perf_counter(); kids = [P.copy() for _ in range(N)]; api.sync(); t_clone = (time.perf_counter() - t) / N
t = time.perf_counter(); kids2 = [P.copy(capi.OFF) for _ in range(N)]; api.sync(); t_clone_nonames = (time.perf_counter() - t) / N
x = P.col_prim()
j = next(j + 1 for j in range(n) if abs(x[j] - round(x[j])) > 1e-9)
t = time.perf_counter()
for k in kids: api.set_col_bnds(k.h, j, capi.DB, 0.0, float(int(x[j - 1])))
api.sync(); t_bnd = (time.perf_counter() - t) / N
t = time.perf_counter()
for k in kids[:200]: bnb.print_info(k, quirks=0)
t_info = (time.perf_counter() - t) / 200
t = time.perf_counter(); del kids; del kids2; api.sync(); t_del = (time.perf_counter() - t) / (2 * N)
print(json.dumps({"m": m, "n": n, "clone_delete_recycled_slab_us": t_steady * 1e6, "clone_us": t_clone * 1e6, "clone_no_names_us": t_clone_nonames * 1e6, "set_col_bnds_us": t_bnd * 1e6,
                  "print_info_us": t_info * 1e6, "delete_us": t_del * 1e6}))
