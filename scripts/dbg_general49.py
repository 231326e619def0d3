"""Diagnostic: trial 49 of tests/test_gpu_chain.py::test_dual_chains_on_general_lps_with_warm_starts, first solves of a process.
usage: dbg_general49.py FIRST_TRIAL   (env: MVX_CLUSTER, MVX_ZC_STAGE, MVX_PERSIST)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import capi
from oracle import oracle
from tests import lpgen
gpu, orc = mvolps_amd.api(), oracle.api()
first = int(sys.argv[1])
tag = "first=%d cl=%s zc=%s" % (first, os.environ.get("MVX_CLUSTER"), os.environ.get("MVX_ZC_STAGE"))
rng = np.random.default_rng(11)
for trial in range(50):
    A, row_b, col_b, c, direction = lpgen.random_general_lp(rng, mmax=60, nmax=90)
    if trial >= first:
        o = orc.create(); o.load_general(A, row_b, col_b, c, direction=direction); o.rc = o.simplex()
        for rep in range(2):
            g = gpu.create(); g.load_general(A, row_b, col_b, c, direction=direction); g.rc = g.simplex()
            fg, fo = g.basis()[2], o.basis()[2]
            bad = np.nonzero(fg != fo)[0]
            if trial == 49 or len(bad):
                print(tag, "trial", trial, A.shape, "rep", rep, "status", g.status, o.status, "it", g.it_cnt, o.it_cnt, "flagdiff", bad.tolist(), "tableau_equal", np.array_equal(g.tableau(), o.tableau()), flush=True)
    j = int(rng.integers(1, A.shape[1] + 1)); v = float(rng.integers(-2, 4))
