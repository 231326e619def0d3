"""Dual simplex on a large LP (set-cover relaxation: min c x, A x >= 1, 0 <= x <= 1 -- dual feasible at the slack
basis): generic path (k_select + k_update per pivot) against the fused dual pair (k_da + k_fb<DUAL>).
usage: MVX_DUAL_FUSED=0|1 dualtime.py M N [PIVOTS]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import capi
from tests import lpgen
m, n = int(sys.argv[1]), int(sys.argv[2])
piv = int(sys.argv[3]) if len(sys.argv) > 3 else 400
api = mvolps_amd.api()
A, c = lpgen.setcover_ilp(m, n, 7, dens=0.05)
P = lpgen.load_setcover(api, A, c)
P.simplex(it_lim=20)
api.sync()
t = time.perf_counter()
P.simplex(it_lim=piv)
api.sync()
el = time.perf_counter() - t
done = P.it_cnt - 20
print(json.dumps({"m": m, "n": n, "fused": os.environ.get("MVX_DUAL_FUSED", "1"), "pivots": done, "us_per_pivot": el / max(1, done) * 1e6,
                  "status": P.status, "obj": P.obj.hex()}), flush=True)
