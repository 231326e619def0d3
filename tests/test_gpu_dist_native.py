"""GPU: the C++ multi-rank entry (mvx_branchAndBound_dist) over the gfx950 engine, and its RCCL transport
(libmvolps_rccl.so).  The box has one GPU: RCCL runs with one rank (the collectives and a send/receive to itself go
through the library's own code path), two ranks share the card over gloo with host copies of the device images."""
import ctypes as C
import json

import numpy as np
import pytest

from mvolps_amd import bnb, capi, dist_native, synth

from . import dist_helpers, lpgen

pytestmark = pytest.mark.gpu
KEYS = ("n_nodes", "parent", "prune", "count", "events", "node_bound", "x", "total_pivots", "incumbent_oid", "best_lower", "has_incumbent", "hit_limit")


def canon(r):
    return json.loads(json.dumps(r))


def test_rccl_transport_with_one_rank(gpu):
    """The three operations of mvx_comm as libmvolps_rccl.so implements them: MAX all-reduce and broadcast of host
    doubles through the device, and a node image sent to and received from the own rank device to device."""
    comm = dist_native.RcclComm(0, 1)
    try:
        c = comm.c
        allreduce = dist_native.ALLREDUCE_FN(c.allreduce_max)
        bcast = dist_native.BCAST_FN(c.bcast)
        exchange = dist_native.EXCHANGE_FN(c.exchange)
        v = np.array([1.5, -np.inf, 3.0, -7.25] * 300, dtype=np.float64)  # more than the first scratch size
        w = v.copy()
        assert allreduce(c.ctx, w.ctypes.data_as(C.POINTER(C.c_double)), w.size) == 0
        assert np.array_equal(v, w)
        assert bcast(c.ctx, w.ctypes.data_as(C.POINTER(C.c_double)), w.size, 0) == 0
        assert np.array_equal(v, w)
        A, b, cc, U = synth.dense_ilp(24, 48, 6, 2)
        root = lpgen.load_ilp(gpu, A, b, cc, U)
        P = root.copy()
        P.simplex()
        x = P.col_prim()
        j = [k + 1 for k in range(48) if np.trunc(x[k]) != x[k]][0]
        gpu.set_col_bnds(P.h, j, capi.LO, float(np.ceil(x[j - 1])), 0.0)
        n = gpu.pack_size_from(P.h, root.h)
        L = dist_native._lib()
        L.mvx_image_alloc.restype = C.c_void_p
        L.mvx_image_alloc.argtypes = [C.c_size_t]
        L.mvx_image_free.argtypes = [C.c_void_p]
        src, dst = L.mvx_image_alloc(n), L.mvx_image_alloc(n)
        assert src and dst
        assert gpu.pack_from(P.h, root.h, src) == 0
        s = (dist_native.Xfer * 1)(dist_native.Xfer(src, n, 0))
        r = (dist_native.Xfer * 1)(dist_native.Xfer(dst, n, 0))
        assert exchange(c.ctx, s, 1, r, 1) == 0
        Q = gpu.create()
        assert gpu.unpack(Q.h, root.h, dst) == 0
        L.mvx_image_free(src)
        L.mvx_image_free(dst)
        assert np.array_equal(P.tableau(), Q.tableau())
        P.simplex()
        Q.simplex()
        assert P.it_cnt == Q.it_cnt and P.obj == Q.obj and np.array_equal(P.tableau(), Q.tableau())
    finally:
        comm.close()


@pytest.mark.parametrize("kw", [dict(quirks=0), dict(quirks=1, cut_strat=1, max_nodes=200), dict(quirks=0, cut_strat=1, cut_select=1, cut_chance=0.4)],
                         ids=["plain", "bugcompat-cuts", "efficacy-cuts"])
def test_native_coordinator_one_rank_matches_driver(gpu, kw):
    A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
    comm = dist_native.RcclComm(0, 1)
    try:
        got = canon(dist_native.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), comm=comm, per_rank=8, **kw))
    finally:
        comm.close()
    ref = canon(bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw))
    for k in KEYS:
        assert got[k] == ref[k], k
    assert ref["count"] > 20


def test_native_coordinator_512x1024_prefix_matches_driver(gpu):
    """BASELINE config 5's shape through the C++ entry: 600 nodes of the wide 512x1024 tree, 64 node LPs per launch."""
    from mvolps_amd import treedigest

    A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
    got = dist_native.branch_and_bound(synth.load_ilp(gpu, A, b, c, U), quirks=0, max_nodes=600, per_rank=64)
    ref = bnb.branch_and_bound(synth.load_ilp(gpu, A, b, c, U), quirks=0, max_nodes=600)
    assert treedigest.digest(got) == treedigest.digest(ref)
    assert got["total_pivots"] == ref["total_pivots"] and got["count"] == 600


def test_two_ranks_one_gpu_match_serial(gpu, tmp_path):
    case = (8, 16, 3, 2)
    A, b, c, U = synth.dense_ilp(*case)
    serial = canon(bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0))
    res = dist_helpers.run_world_native(2, case, dict(quirks=0, per_rank=2), str(tmp_path), use_gpu=True)
    for r in res:
        for k in KEYS:
            assert r[k] == serial[k], k
    assert res[0]["dist"]["migrated"] > 0
