"""GPU: node migration images in device memory, and the coordinator over the gfx950 engine.
Two ranks share the box's single GPU here (gloo moves host copies of the images); on a real
8-GPU node the same code runs one rank per GPU with RCCL."""
import json

import numpy as np
import pytest
import torch

from mvolps_amd import bnb, capi, dist_bnb, synth

from . import dist_helpers, lpgen

pytestmark = pytest.mark.gpu


def test_device_pack_unpack_roundtrip(gpu):
    eng = dist_bnb.HipNodeEngine(0)
    A, b, c, U = synth.dense_ilp(24, 48, 6, 2)
    root = lpgen.load_ilp(gpu, A, b, c, U)
    P = root.copy()
    P.simplex()
    x = P.col_prim()
    j = [k + 1 for k in range(48) if np.trunc(x[k]) != x[k]][0]
    gpu.set_col_bnds(P.h, j, capi.LO, float(np.ceil(x[j - 1])), 0.0)
    img = eng.pack(P, root)
    assert img.is_cuda and img.numel() == gpu.pack_size(P.h)
    Q = eng.unpack(root, img)
    assert np.array_equal(P.tableau(), Q.tableau())
    for u, v in zip(P.basis(), Q.basis()):
        assert np.array_equal(u, v)
    P.simplex()
    Q.simplex()
    assert P.it_cnt == Q.it_cnt and P.obj == Q.obj and np.array_equal(P.tableau(), Q.tableau())


def test_coordinator_world1_matches_driver(gpu):
    eng = dist_bnb.HipNodeEngine(0)
    A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
    got = json.loads(json.dumps(dist_bnb.branch_and_bound(eng, lpgen.load_ilp(gpu, A, b, c, U), quirks=0)))
    ref = json.loads(json.dumps(bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0)))
    for k in ("n_nodes", "parent", "prune", "count", "events", "node_bound", "x", "total_pivots", "incumbent_oid"):
        assert got[k] == ref[k], k


def test_two_ranks_one_gpu_match_serial(gpu, tmp_path):
    case = (8, 16, 3, 2)
    A, b, c, U = synth.dense_ilp(*case)
    serial = json.loads(json.dumps(bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0)))
    res = dist_helpers.run_world(2, case, dict(quirks=0, per_rank=2), str(tmp_path), use_gpu=True)
    for r in res:
        for k in ("n_nodes", "parent", "prune", "count", "events", "node_bound", "x", "total_pivots", "incumbent_oid", "best_lower"):
            assert r[k] == serial[k], k


def test_coordinator_over_rccl_world1(gpu, tmp_path):
    """backend "nccl" (= RCCL) with one rank on the one GPU: the MAX all-reduces and the incumbent
    broadcast run on device tensors through RCCL; result equals the serial driver's."""
    case = (8, 16, 3, 2)
    A, b, c, U = synth.dense_ilp(*case)
    serial = json.loads(json.dumps(bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0, window=1)))
    r = dist_helpers.run_nccl_world1(case, dict(quirks=0, per_rank=4), str(tmp_path))
    for k in ("n_nodes", "parent", "prune", "count", "events", "node_bound", "x", "total_pivots", "incumbent_oid", "best_lower"):
        assert r[k] == serial[k], k
