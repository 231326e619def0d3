"""Host cost of one B&B clone (mvx_create_prob + mvx_copy_prob) and of its release, at 512x1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
from tests import lpgen
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3)
P = lpgen.load_ilp(api, A, b, c, U)
P.simplex()
N = 1000
for names in (1, 0):
    for rep in range(2):
        t = time.perf_counter()
        kids = []
        for i in range(N):
            Q = api.create_prob()
            api.copy_prob(Q, P.h, names)
            kids.append(Q)
        t1 = time.perf_counter()
        api.get_obj_val(kids[-1])
        for Q in kids:
            api.delete_prob(Q)
        t2 = time.perf_counter()
        print(f"names={names} clone {1e6*(t1-t)/N:.1f} us  delete {1e6*(t2-t1)/N:.1f} us")
