"""Kernel-trace / PMC input: M N [PIVOTS] pivots of a dense LP on the fused path."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
m, n = int(sys.argv[1]), int(sys.argv[2])
api = mvolps_amd.api()
A, b, c = synth.dense_lp(m, n, 12345)
P = api.create(); P.load_dense(A, b, c)
P.simplex(it_lim=30)
P.simplex(it_lim=int(sys.argv[3]) if len(sys.argv) > 3 else 200)
print(P.it_cnt)
