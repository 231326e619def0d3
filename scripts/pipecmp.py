"""A/B of the two primal pipelines (k_fs single launch vs k_fa/k_fb pair), interleaved in one process."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
api = mvolps_amd.api()
for (m, n, steps) in [(4096, 8192, 400), (2048, 4096, 400), (1024, 2048, 600), (512, 1024, 300)]:
    A, b, c = synth.dense_lp(m, n, 12345)
    res = {0: [], 1: []}
    ref = None
    for rep in range(3):
        for pipe in (0, 1):
            api.use_pipeline(pipe)
            P = api.create(); P.load_dense(A, b, c); P.simplex(it_lim=40); api.sync()
            t = time.perf_counter(); P.simplex(it_lim=steps); api.sync(); el = time.perf_counter() - t
            res[pipe].append(el / steps * 1e6)
            if ref is None: ref = P.obj
            assert P.obj == ref
            del P
    api.use_pipeline(0)
    print(json.dumps({"m": m, "n": n, "two_kernel_us": min(res[0]), "single_launch_us": min(res[1]),
                      "speedup": min(res[0]) / min(res[1]), "roofline_frac_single": 16 * (m + 1) * (n + 1) / (min(res[1]) * 1e-6) / 8e12}), flush=True)
