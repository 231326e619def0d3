"""Determinism soak of the window driver: the wide 512x1024 tree (first N nodes) and the config-5 tree, again and again in
one process under several MVX_BNB_* settings; every run must give the same pivots and digest.  Prints the odd ones."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth, treedigest
api = mvolps_amd.api()
lib = mvolps_amd.load_library()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = json.load(open(os.path.join(root, "tests", "golden", "config5.json")))
wide = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
c5 = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ref = {"wide": (2000, 85493, "39f32327363e")}
bad = 0
for cfg in ({"MVX_BNB_PREFIX": "0", "MVX_BNB_CHUNK": "1000"}, {"MVX_BNB_CHUNK": "2"}, {"MVX_BNB_CHUNK": "8"}, {"MVX_BNB_CHUNK": "4", "MVX_BNB_DEPTH": "16"}):
    for k in ("MVX_BNB_PREFIX", "MVX_BNB_CHUNK", "MVX_BNB_DEPTH"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    for name, inst, kw, n in (("wide", wide, dict(quirks=0, max_nodes=2000, window=64), reps), ("config5", c5, dict(quirks=0, window=64), max(1, reps // 5))):
        ts = []
        for rep in range(n):
            t = time.perf_counter()
            r = bnb.branch_and_bound(synth.load_ilp(api, *inst), **kw)
            ts.append(time.perf_counter() - t)
            key = (r["count"], r["total_pivots"], treedigest.digest(r)[:12])
            dbg = (C.c_longlong * 8)()
            lib.mvx_debug_counters(dbg, 1)
            if name not in ref:
                ref[name] = key
            if key != ref[name]:
                bad += 1
                print("ODD", cfg, name, rep, key, "expected", ref[name], "%.1f ms" % (ts[-1] * 1e3), "refreshes/looks/singles/batches/fallbacks", list(dbg)[:5], flush=True)
            elif rep == 0:
                print("first", cfg, name, key, "refreshes/looks/singles/batches/fallbacks", list(dbg)[:5], flush=True)
        ts.sort()
        print(cfg, name, "runs", n, "median %.1f ms  max %.1f ms" % (ts[len(ts) // 2] * 1e3, ts[-1] * 1e3), ref[name], flush=True)
l, a = C.c_longlong(), C.c_longlong()
lib.mvx_cluster_stats(C.byref(l), C.byref(a))
print("odd runs", bad, "cluster launches", l.value, "aborts", a.value)
