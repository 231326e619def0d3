"""GPU: GMI cut generation on the device (mvx_gmi_cuts: tableau rows, the coefficient formula of gmi.cpp:41-74 and the
back-substitution of gmi.cpp:81-89 in one pass for all columns) against the oracle's one-column restatements
(orc_generateCut3, orc_generateCutGMI), bitwise -- NaN and infinities of the bug-compatible formula included."""
import ctypes as C

import numpy as np
import pytest

from mvolps_amd import bnb, capi, synth

from . import lpgen

pytestmark = pytest.mark.gpu


def device_cuts(gpu, P, cols, repaired):
    lib = gpu.lib
    n = P.n
    k = len(cols)
    vals = np.zeros((k, n + 1))
    rhs = np.zeros(k)
    ok = np.zeros(k, dtype=np.int32)
    arr = np.asarray(cols, dtype=np.int32)
    lib.mvx_gmi_cuts.restype = C.c_int
    lib.mvx_gmi_cuts.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.mvx_gmi_cuts(P.h, repaired, arr.ctypes.data, k, vals.ctypes.data, rhs.ctypes.data, ok.ctypes.data)
    assert rc == 0
    return vals, rhs, ok


def oracle_cut(orc, Q, j, repaired):
    n = Q.n
    inds = np.zeros(n + 1, dtype=np.int32)
    vals = np.zeros(n + 1)
    lb = C.c_double(0.0)
    ip, vp = inds.ctypes.data_as(C.POINTER(C.c_int)), vals.ctypes.data_as(C.POINTER(C.c_double))
    if repaired:
        eff = C.c_double(0.0)
        rc = orc.generateCutGMI(Q.h, j, ip, vp, C.byref(lb), C.byref(eff))
    else:
        rc = orc.generateCut3(Q.h, j, ip, vp, C.byref(lb))
    return rc, vals, lb.value


def same_bits(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64), np.asarray(b, dtype=np.float64).view(np.uint64))


@pytest.mark.parametrize("case", [(24, 48, 6, 2), (64, 128, 3, 3), (512, 1024, 12345, 3)], ids=lambda c: "%dx%d" % (c[0], c[1]))
def test_device_cuts_equal_the_one_column_restatements(gpu, orc, case):
    m, n, seed, U = case
    A, b, c, U = synth.dense_ilp(m, n, seed, U)
    P, Q = lpgen.load_ilp(gpu, A, b, c, U), lpgen.load_ilp(orc, A, b, c, U)
    for H in (P, Q):
        H.simplex()
    for rounds in range(3):  # the second and third pass see appended cut rows (positional back-substitution over them)
        stat = Q.col_stat()
        basic = [j + 1 for j in range(n) if stat[j] == capi.BS]
        assert len(basic) > 3
        for repaired in (0, 1):
            vals, rhs, ok = device_cuts(gpu, P, basic, repaired)
            for t, j in enumerate(basic):
                rc, rv, rl = oracle_cut(orc, Q, j, repaired)
                if repaired and rc != 0:
                    # rejected for its fractional part or its norm on the host side of the driver; the engine only
                    # reports the free-non-basic rejection
                    continue
                assert rc == 0 and ok[t] == 1
                assert same_bits(rhs[t], rl), (j, rhs[t], rl)
                assert same_bits(vals[t, 1:], rv[1:]), j
        # append the last column's bug-compatible cut to both and re-solve, as bs.cpp:249-258 + cut.cpp:23-43 do
        j = basic[-1]
        rc, rv, rl = oracle_cut(orc, Q, j, 0)
        ind = np.arange(n + 1, dtype=np.int32)
        for api, H in ((gpu, P), (orc, Q)):
            r = api.add_rows(H.h, 1)
            H.set_mat_row(r, ind, rv)
            api.set_row_bnds(H.h, r, capi.LO, float(rl), 0.0)
            H.simplex()
        assert np.array_equal(P.tableau(), Q.tableau())


def test_driver_uses_the_device_entry_and_matches_the_host_path(gpu):
    """The same driver with and without mvx_lp_api.gmi_cuts (host loop per cut) builds the same tree."""
    A, b, c, U = synth.dense_ilp(64, 128, 3, 3)
    tab = bnb.table_from(gpu)
    assert tab.gmi_cuts
    host = bnb.table_from(gpu)
    host.gmi_cuts = None
    for kw in (dict(quirks=1, cut_strat=1, lazy_pool=0), dict(quirks=0, cut_strat=1, cut_select=1, cut_chance=0.2), dict(quirks=0, cut_strat=1)):
        r_dev = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), max_nodes=200, table=tab, **kw)
        r_host = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), max_nodes=200, table=host, **kw)
        for key in ("events", "prune", "parent", "node_bound", "total_pivots", "count"):
            assert repr(r_dev[key]) == repr(r_host[key]), (kw, key)
