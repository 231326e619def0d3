"""One primal run of a chosen pipeline, for rocprofv3 --kernel-trace --stats.  usage: pipeprof.py PIPE M N STEPS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
pipe, m, n, steps = (int(x) for x in sys.argv[1:5])
api = mvolps_amd.api()
api.use_pipeline(pipe)
A, b, c = synth.dense_lp(m, n, 12345)
P = api.create(); P.load_dense(A, b, c); P.simplex(it_lim=steps); api.sync()
print(P.status, P.obj, P.it_cnt)
