"""Why is the B&B leg slower inside bench.py than on its own?  Same 2000-node run after (a) nothing, (b) torch.cuda
initialised, (c) the 4096x8192 headline LP solved first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3)
def run(tag):
    bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=64)
    t = time.perf_counter()
    r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=2000)
    dt = time.perf_counter() - t
    print(tag, "nodes/s %.0f" % (r["count"] / dt), flush=True)
mode = sys.argv[1]
if mode in ("b", "c", "d"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
if mode in ("c",):
    A2, b2, c2 = synth.dense_lp(4096, 8192, 12345)
    P = api.create(); P.load_dense(A2, b2, c2); P.simplex(it_lim=300)
    Q = P.copy(); Q.simplex(it_lim=100)
if mode in ("e", "f", "g"):
    import torch
    torch.cuda.set_device(0)
    A2, b2, c2 = synth.dense_lp(4096, 8192, 12345)
    P0 = api.create(); P0.load_dense(A2, b2, c2); del A2
    P0.simplex(it_lim=0)
    P = P0.copy()
    P.simplex(it_lim=50)
    torch.cuda.synchronize()
    P.simplex(it_lim=600)
    torch.cuda.synchronize()
    if mode in ("f", "g"):
        api.profile_reset(); api.profile_enable(1); P.simplex(it_lim=200); api.profile_enable(0)
    if mode == "g":
        A3, b3, c3 = synth.dense_lp(1024, 2048, 12345)
        Q = api.create(); Q.load_dense(A3, b3, c3); Q.simplex(it_lim=50); api.sync(); Q.simplex(it_lim=600); api.sync()
if mode == "d":
    A2, b2, c2 = synth.dense_lp(1024, 2048, 12345)
    P = api.create(); P.load_dense(A2, b2, c2); P.simplex(it_lim=300)
run(mode); run(mode)
