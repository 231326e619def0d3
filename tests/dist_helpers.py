"""Test-side adapters for the distributed coordinator (tests only: this is where the ORACLE is plugged in)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleNodeEngine:
    """CPU stand-in for HipNodeEngine built on the oracle library; migration images are host tensors."""

    def __init__(self):
        from mvolps_amd import bnb
        from oracle import oracle

        self.api = oracle.api()
        self.table = bnb.table_from(self.api)
        self._bnb = bnb
        self.comm_device = torch.device("cpu")

    def print_info(self, prob, quirks):
        return self._bnb.print_info(prob, quirks=quirks, table=self.table)

    def classify(self, prob, root, quirks, var_strat):
        return self._bnb.classify(prob, root, quirks, var_strat, table=self.table)

    def make_children(self, a, pick, quirks):
        return self._bnb.make_children(a, pick, quirks, table=self.table)

    def solve_many(self, probs):
        for p in probs:
            p.simplex()

    def node_cuts(self, a, params):
        return self._bnb.node_cuts(a, params, table=self.table)

    def pack_size(self, prob, base):
        return self.api.pack_size_from(prob.h, base.h)

    def pack(self, prob, base):
        n = self.api.pack_size_from(prob.h, base.h)
        buf = np.zeros(n, dtype=np.uint8)
        assert self.api.pack_from(prob.h, base.h, buf.ctypes.data) == 0
        return torch.from_numpy(buf)

    def recv_buffer(self, nbytes):
        return torch.empty(nbytes, dtype=torch.uint8)

    def unpack(self, base, t):
        buf = np.ascontiguousarray(t.numpy())
        q = self.api.create()
        assert self.api.unpack(q.h, base.h, buf.ctypes.data) == 0
        return q


def _worker(rank, world, port, outdir, case, kw, use_gpu):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import torch.distributed as dist

    from mvolps_amd import dist_bnb, synth
    from tests import lpgen

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        if use_gpu:
            eng = dist_bnb.HipNodeEngine(0, comm_device="cpu")  # both ranks share cuda:0; gloo moves host copies
        else:
            eng = OracleNodeEngine()
        root = lpgen.load_case(eng.api, tuple(case))
        res = dist_bnb.branch_and_bound(eng, root, **kw)
        with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
            json.dump(res, f)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def run_world(world, case, kw, outdir, use_gpu=False, port=None):
    import torch.multiprocessing as mp

    port = port or (29500 + (os.getpid() % 2000))
    mp.spawn(_worker, args=(world, port, outdir, case, kw, use_gpu), nprocs=world, join=True)
    return [json.load(open(os.path.join(outdir, "rank%d.json" % r))) for r in range(world)]


def _nccl_world1(rank, port, outdir, case, kw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from mvolps_amd import dist_bnb, synth
    from tests import lpgen

    eng = dist_bnb.HipNodeEngine(0)  # device tensors: the collectives below run through RCCL
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        A, b, c, U = synth.dense_ilp(*case)
        res = dist_bnb.branch_and_bound(eng, lpgen.load_ilp(eng.api, A, b, c, U), **kw)
        with open(os.path.join(outdir, "nccl.json"), "w") as f:
            json.dump(res, f)
    finally:
        dist.destroy_process_group()


def run_nccl_world1(case, kw, outdir, port=None):
    import torch.multiprocessing as mp

    port = port or (31500 + (os.getpid() % 2000))
    mp.spawn(_nccl_world1, args=(port, outdir, case, kw), nprocs=1, join=True)
    return json.load(open(os.path.join(outdir, "nccl.json")))


# ---- the C++ coordinator (mvx_branchAndBound_dist) behind the same harness
def oracle_tables():
    """(api, lp table, image table) of the oracle library: host buffers from libc."""
    from mvolps_amd import bnb, dist_native
    from oracle import oracle

    api = oracle.api()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    return api, bnb.table_from(api), dist_native.image_api_from(api, libc.malloc, libc.free)


def _native_worker(rank, world, port, outdir, case, kw, use_gpu):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import torch.distributed as dist

    from mvolps_amd import dist_native
    from tests import lpgen

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        if use_gpu:
            import mvolps_amd

            mvolps_amd.require_device()
            api, table, image = mvolps_amd.api(), None, None  # the engine's own tables; both ranks share cuda:0
            api.set_device(0)
            torch.cuda.set_device(0)
            comm = dist_native.TorchComm(device_buffers=True)  # gloo moves host copies of the device images
        else:
            api, table, image = oracle_tables()
            comm = dist_native.TorchComm()
        root = lpgen.load_case(api, tuple(case))
        res = dist_native.branch_and_bound(root, comm=comm, table=table, image=image, **kw)
        with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
            json.dump(res, f)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def run_world_native(world, case, kw, outdir, use_gpu=False, port=None):
    import torch.multiprocessing as mp

    port = port or (33500 + (os.getpid() % 2000))
    mp.spawn(_native_worker, args=(world, port, outdir, case, kw, use_gpu), nprocs=world, join=True)
    return [json.load(open(os.path.join(outdir, "rank%d.json" % r))) for r in range(world)]
