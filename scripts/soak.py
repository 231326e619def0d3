"""Run-to-run determinism soak of the fused primal path: repeated full solves must give identical bits; solves
cut into pieces by iteration limits (kernel boundaries at other pivots, devex weights restarted per call, so
another path) must reach the same optimum to 1e-9."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth
api = mvolps_amd.api()
bad = 0
for (m, n, reps) in [(512, 1024, 30), (1024, 2048, 20), (2048, 4096, 10), (4096, 8192, 4), (777, 3001, 10)]:
    A, b, c = synth.dense_lp(m, n, 99)
    ref = None
    t0 = time.time()
    for r in range(reps):
        P = api.create(); P.load_dense(A, b, c)
        # vary the batching so that kernel boundaries fall at different pivots
        if r % 3 == 1:
            P.simplex(it_lim=17 + r)
        if r % 3 == 2:
            P.simplex(it_lim=100 + 7 * r); P.simplex(it_lim=3)
        P.simplex()
        h = hashlib.sha256(P.tableau().tobytes()).hexdigest()
        sig = (P.status, P.it_cnt, P.obj, h)
        if ref is None:
            ref = sig
        elif r % 3 == 0 and sig != ref:
            bad += 1
            print("MISMATCH", m, n, r, sig, ref, flush=True)
        elif r % 3 and (sig[0] != ref[0] or abs(sig[2] - ref[2]) > 1e-9 * abs(ref[2])):
            bad += 1
            print("MISMATCH (split solve)", m, n, r, sig[:3], ref[:3], flush=True)
        del P
    print("%dx%d: %d solves identical=%s pivots=%d obj=%.12g (%.1fs)" % (m, n, reps, bad == 0, ref[1], ref[2], time.time() - t0), flush=True)
print("bad", bad)
