"""Python binding of the multi-GPU branch-and-bound entry (include/mvx_dist.h, csrc/bnb_dist.cpp).

The coordinator itself is C++ (mvx_branchAndBound_dist); it reaches its peers through the three operations of
`mvx_comm`.  Two tables are offered here:

* RcclComm  -- libmvolps_rccl.so: RCCL called directly on device buffers (ncclAllReduce / ncclSend / ncclRecv /
  ncclBroadcast); the id of rank 0 reaches the other ranks through torch.distributed's store (any launcher's channel
  would do);
* TorchComm -- the same three operations over an initialised torch.distributed process group through ctypes
  callbacks: what the world-2 gloo tests run on the CPU, and a second way to carry the images on the GPUs.

mvolps_amd.dist_bnb is the earlier pure-Python coordinator of the same algorithm; both produce the serial driver's
tree (tests/test_dist_bnb.py, tests/test_dist_native.py).
"""
import ctypes as C
import os

import numpy as np

from . import bnb


class ImageApi(C.Structure):
    """struct mvx_image_api"""

    _fields_ = [(name, C.c_void_p) for name in ("pack_size", "pack", "unpack", "buf_alloc", "buf_free")]


class Xfer(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("bytes", C.c_size_t), ("peer", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer), C.c_int, C.POINTER(Xfer), C.c_int)
BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_int)


class Comm(C.Structure):
    """struct mvx_comm"""

    _fields_ = [
        ("ctx", C.c_void_p),
        ("rank", C.c_int),
        ("size", C.c_int),
        ("allreduce_max", C.c_void_p),
        ("exchange", C.c_void_p),
        ("bcast", C.c_void_p),
    ]


class DistParams(C.Structure):
    _fields_ = [("per_rank", C.c_int), ("slack", C.c_int), ("roundrobin", C.c_int)]


class DistStats(C.Structure):
    _fields_ = [("children", C.c_longlong), ("migrated", C.c_longlong), ("migrated_bytes", C.c_longlong), ("rounds", C.c_longlong), ("allreduces", C.c_longlong)]


def _lib():
    L = bnb.lib()
    if not getattr(L, "_dist_bound", False):
        L.mvx_branchAndBound_dist.restype = C.c_int
        L.mvx_branchAndBound_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(bnb.BnbParams), C.POINTER(DistParams),
                                              C.c_void_p, C.POINTER(bnb.BnbResult), C.POINTER(DistStats)]
        L.mvx_dist_default_params.argtypes = [C.POINTER(DistParams)]
        L.mvx_hip_image_api.restype = C.c_void_p
        L._dist_bound = True
    return L


def image_api_from(api, alloc, free):
    """mvx_image_api out of a library that exports pack_size_from / pack_from / unpack under api.prefix and a pair of
    buffer functions (ctypes function objects)."""
    t = ImageApi()
    t.pack_size = C.cast(getattr(api.lib, api.prefix + "pack_size_from"), C.c_void_p).value
    t.pack = C.cast(getattr(api.lib, api.prefix + "pack_from"), C.c_void_p).value
    t.unpack = C.cast(getattr(api.lib, api.prefix + "unpack"), C.c_void_p).value
    t.buf_alloc = C.cast(alloc, C.c_void_p).value
    t.buf_free = C.cast(free, C.c_void_p).value
    return t


class _DevView:
    """A raw device pointer dressed as a __cuda_array_interface__ object so that torch can alias it."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchComm:
    """mvx_comm over torch.distributed.  device_buffers=True: the image buffers are device memory (the gfx950 engine):
    they are aliased as CUDA tensors and handed to the process group as they are (backend nccl = RCCL), or staged
    through the host when the group is gloo (rehearsal of several ranks on one GPU)."""

    def __init__(self, group=None, device_buffers=False, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.device_buffers = device_buffers
        self.on_gpu = dist.get_backend(group) == "nccl"
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if (device_buffers or self.on_gpu) else None)
        self.error = None
        self._cb = (ALLREDUCE_FN(self._allreduce), EXCHANGE_FN(self._exchange), BCAST_FN(self._bcast))  # kept alive here
        self.c = Comm(None, self.rank, self.size, C.cast(self._cb[0], C.c_void_p).value, C.cast(self._cb[1], C.c_void_p).value,
                      C.cast(self._cb[2], C.c_void_p).value)

    def _host(self, ptr, n):
        return self.torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(n,)))

    def _allreduce(self, ctx, v, n):
        try:
            t = self._host(v, n)
            if self.on_gpu:
                d = t.to(self.device)
                self.dist.all_reduce(d, op=self.dist.ReduceOp.MAX, group=self.group)
                t.copy_(d.cpu())
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
            return 0
        except Exception as e:  # an exception must not cross the C frame
            self.error = e
            return 1

    def _bcast(self, ctx, v, n, root):
        try:
            src = root if self.group is None else self.dist.get_global_rank(self.group, root)
            t = self._host(v, n)
            if self.on_gpu:
                d = t.to(self.device)
                self.dist.broadcast(d, src=src, group=self.group)
                t.copy_(d.cpu())
            else:
                self.dist.broadcast(t, src=src, group=self.group)
            return 0
        except Exception as e:
            self.error = e
            return 1

    def _view(self, x):
        if self.device_buffers:
            return self.torch.as_tensor(_DevView(x.buf, x.bytes), device=self.device)
        return self.torch.from_numpy(np.ctypeslib.as_array(C.cast(x.buf, C.POINTER(C.c_ubyte)), shape=(x.bytes,)))

    def _exchange(self, ctx, sends, ns, recvs, nr):
        try:
            torch, dist = self.torch, self.dist
            stage = self.device_buffers and not self.on_gpu  # device images over a host transport
            ops, later = [], []
            for k in range(ns):
                t = self._view(sends[k])
                if stage:
                    t = t.cpu()
                elif self.on_gpu and not self.device_buffers:
                    t = t.to(self.device)
                ops.append(dist.P2POp(dist.isend, t, sends[k].peer, group=self.group))
            for k in range(nr):
                dst = self._view(recvs[k])
                t = dst
                if stage:
                    t = torch.empty(recvs[k].bytes, dtype=torch.uint8)
                    later.append((dst, t))
                elif self.on_gpu and not self.device_buffers:
                    t = torch.empty(recvs[k].bytes, dtype=torch.uint8, device=self.device)
                    later.append((dst, t))
                ops.append(dist.P2POp(dist.irecv, t, recvs[k].peer, group=self.group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for dst, t in later:
                dst.copy_(t)
            if self.device_buffers:
                torch.cuda.synchronize()
            return 0
        except Exception as e:
            self.error = e
            return 1

    def pointer(self):
        return C.cast(C.pointer(self.c), C.c_void_p)

    def close(self):
        pass


_rccl = None


def rccl_library():
    """libmvolps_rccl.so (built by mvolps_amd.build next to the engine library); raises when it is missing."""
    global _rccl
    if _rccl is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmvolps_rccl.so")
        if not os.path.exists(path):
            raise RuntimeError("libmvolps_rccl.so has not been built (python __graft_entry__.py)")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.mvx_rccl_unique_id.restype = C.c_int
        L.mvx_rccl_unique_id.argtypes = [C.c_void_p]
        L.mvx_rccl_comm_create.restype = C.c_int
        L.mvx_rccl_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Comm)]
        L.mvx_rccl_comm_destroy.argtypes = [C.POINTER(Comm)]
        _rccl = L
    return _rccl


class RcclComm:
    """mvx_comm over RCCL, called from C++ (libmvolps_rccl.so).  The unique id is made on rank 0 and published through
    an initialised torch.distributed group of any backend (its store is only the bootstrap channel)."""

    ID_BYTES = 128

    def __init__(self, rank, size, id_bytes=None, group=None):
        L = rccl_library()
        if id_bytes is None:
            import torch.distributed as dist

            box = [None]
            if rank == 0:
                buf = C.create_string_buffer(self.ID_BYTES)
                if L.mvx_rccl_unique_id(buf) != 0:
                    raise RuntimeError("ncclGetUniqueId failed")
                box[0] = bytes(buf.raw)
            if size > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            id_bytes = box[0]
        self.c = Comm()
        idb = C.create_string_buffer(id_bytes, self.ID_BYTES)
        rc = L.mvx_rccl_comm_create(idb, rank, size, C.byref(self.c))
        if rc != 0:
            raise RuntimeError("mvx_rccl_comm_create failed (%d)" % rc)
        self.rank, self.size, self.error = rank, size, None

    def pointer(self):
        return C.cast(C.pointer(self.c), C.c_void_p)

    def close(self):
        if self.c.ctx:
            rccl_library().mvx_rccl_comm_destroy(C.byref(self.c))


def branch_and_bound(root, comm=None, table=None, image=None, per_rank=64, slack=None, deal="owner", var_strat=0, cut_strat=0, max_nodes=0,
                     quirks=1, lazy_pool=1, cut_select=0, cut_chance=1.0):
    """mvx_branchAndBound_dist on this rank's handle `root` of the root problem.  table / image None: the gfx950
    engine's own tables.  Returns mvolps_amd.bnb.branch_and_bound's dictionary (identical on every rank) plus `dist`."""
    L = _lib()
    pr = bnb.make_params(var_strat, 0, cut_strat, max_nodes, quirks, lazy_pool, None, cut_select, cut_chance)
    dp = DistParams()
    L.mvx_dist_default_params(C.byref(dp))
    dp.per_rank = per_rank
    dp.slack = -1 if slack is None else slack
    dp.roundrobin = 1 if deal == "roundrobin" else 0
    res, st = bnb.BnbResult(), DistStats()
    tptr = C.cast(C.pointer(table), C.c_void_p) if table is not None else None
    iptr = C.cast(C.pointer(image), C.c_void_p) if image is not None else None
    rc = L.mvx_branchAndBound_dist(tptr, iptr, root.h, C.byref(pr), C.byref(dp), comm.pointer() if comm is not None else None, C.byref(res), C.byref(st))
    if rc != 0:
        err = getattr(comm, "error", None)
        raise RuntimeError("mvx_branchAndBound_dist failed (%#x)%s" % (rc, ": %r" % (err,) if err else ""))
    out = bnb.result_to_dict(res)
    L.mvx_bnb_free_result(C.byref(res))
    world = comm.size if comm is not None else 1
    out["dist"] = {"world": world, "per_rank": per_rank, "deal": deal, "slack": dp.slack if dp.slack >= 0 else max(1, per_rank // 4),
                   "children": st.children, "migrated": st.migrated, "migrated_bytes": st.migrated_bytes, "rounds": st.rounds, "allreduces": st.allreduces,
                   "coordinator": "mvx_branchAndBound_dist"}
    return out
