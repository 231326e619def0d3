"""Multi-GPU branch-and-bound: node LPs farmed over the ranks of one node, serial-equivalent.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the
CPU tests).  What is sharded is the unit the reference itself treats as independent: one B&B node
= one `glp_prob` clone + its `glp_simplex` calls (/root/reference/bs.cpp:114-117,269-288).  The only
shared state is the incumbent `bestLower` (bs.cpp:90,172-174) and the child bounds
(`NodeData::upperBound`, bs.cpp:280,288); both travel in small MAX all-reduces.

Serial equivalence.  With the reference's default FIFO node order
(`problems.front()`, util.cpp:165-166) children go to the back of the deque (bs.cpp:297-298) and no
node is discarded before it has been solved, so the next W nodes the serial loop would pop are
exactly the front W of the deque whatever their outcome.  Each round therefore
  A. solves the front window in parallel (owner ranks), MAX-all-reduces (status, objective, ...);
  B. replays the serial decisions in queue order on every rank (incumbent updated in that order),
     which fixes the branch list and the child oids;
  C. lets the owners append their nodes' GMI cut rows (bs.cpp:249-258), create and solve the children
     (bs.cpp:269-288), MAX-all-reduces their bounds;
  D. gives every child an owner.  Ownership does not affect the result, so a child STAYS on its parent's
     rank -- its tableau is already there -- unless that rank's share of the window the child will be popped
     in is full: every future window (W = world x per_rank consecutive queue positions, known exactly under
     FIFO order) takes at most per_rank nodes per rank, and while the whole queue still fits one window the
     nodes are spread evenly.  Only a child that has to change ranks is migrated, as a device image of its
     bounds + basis + tableau + appended cut rows (mvx_pack_from / mvx_unpack) through an RCCL send/recv -- a
     bitwise copy, so its later arithmetic is the serial run's.  `result["dist"]` reports children, migrated
     images and bytes (dealing children round-robin, as round 1 did, migrates (N-1)/N of all children).
Tree, oids, prune labels, events and the incumbent come out identical to the serial driver
(mvx_branchAndBound); tests/test_dist_bnb.py asserts that with world_size 2.

Best-bound order (util.cpp:170-186) picks by fresh child bounds and is not window-batchable; it stays on the
single-GPU driver.  Bug-compatible cuts: bs.cpp:73's pool persists across nodes, but the only cut ever appended
is the pool's last (cut.cpp:20), which is the branching node's own whenever it generates one -- and a branched
node always does (its fractional integer column is basic); the coordinator raises if that ever fails.
"""
from collections import deque

import numpy as np
import torch

from . import capi

NEG_INF = float("-inf")
EV_PREGNANT, EV_INTEGER, EV_INFEASIBLE, EV_FATHOMED, EV_BRANCHED, EV_CANDIDATE = range(6)
INTG, FEAS, BNDS, NONE = 0, 1, 3, 4


def branch_direction(oid):
    """bs.cpp:43-52"""
    if oid <= 1:
        return 0
    return 1 if oid % 2 == 0 else 2


class HipNodeEngine:
    """The gfx950 engine as seen by the coordinator: handles, printInfo, and migration images that
    live in device memory (torch CUDA tensors) so that RCCL moves them GPU to GPU."""

    def __init__(self, local_rank=0, comm_device=None):
        import mvolps_amd
        from . import bnb

        mvolps_amd.require_device()
        self.api = mvolps_amd.api()
        if self.api.set_device(local_rank) != 0:
            raise RuntimeError("cannot bind device %d" % local_rank)
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        # tensors handed to the process group: on the GPU for RCCL, on the host for gloo
        self.comm_device = torch.device(comm_device) if comm_device else self.device
        self._bnb = bnb
        self.table = None

    def print_info(self, prob, quirks):
        return self._bnb.print_info(prob, quirks=quirks, table=self.table)

    def classify(self, prob, root, quirks, var_strat):
        return self._bnb.classify(prob, root, quirks, var_strat, table=self.table)

    def make_children(self, a, pick, quirks):
        return self._bnb.make_children(a, pick, quirks, table=self.table)

    def solve_many(self, probs):
        """Independent handles that share every launch (mvx_simplex_batch: one grid.z slot each)."""
        if not probs:
            return
        import ctypes as C

        arr = (C.c_void_p * len(probs))(*[p.h for p in probs])
        self.api.simplex_batch(arr, len(probs), None, None)

    def node_cuts(self, a, params):
        return self._bnb.node_cuts(a, params, table=self.table)

    def pack_size(self, prob, base):
        return self.api.pack_size_from(prob.h, base.h)

    def pack(self, prob, base):
        n = self.api.pack_size_from(prob.h, base.h)
        t = torch.empty(n, dtype=torch.uint8, device=self.device)
        if self.api.pack_from(prob.h, base.h, t.data_ptr()) != 0:
            raise RuntimeError("mvx_pack_from failed")
        return t.to(self.comm_device)

    def recv_buffer(self, nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device=self.comm_device)

    def unpack(self, base, t):
        t = t.to(self.device)
        torch.cuda.synchronize()
        q = self.api.create()
        if self.api.unpack(q.h, base.h, t.data_ptr()) != 0:
            raise RuntimeError("mvx_unpack failed")
        return q


class _Node:
    # cls: what the window step (bs.cpp:114-117 + printInfo) will find when the node is popped -- (status, objective,
    # violated columns, their fractional sum, pick, pivots of the pop-time re-solve) -- worked out by the rank that
    # created the node right after solving it (nothing touches a node's problem between its creation and its pop) and
    # shipped in the round's ONE all-reduce next to the child's bound: popping a window needs no collective.
    # solver: the rank that solved it (holds its solution if it turns out integral).
    __slots__ = ("oid", "owner", "upper", "inital", "cls", "solver")

    def __init__(self, oid, owner, upper, inital=False, cls=None, solver=None):
        self.oid, self.owner, self.upper, self.inital = oid, owner, upper, inital
        self.cls, self.solver = cls, owner if solver is None else solver


def branch_and_bound(engine, root, var_strat=0, quirks=1, max_nodes=0, loop_limit=200000, per_rank=1, group=None,
                     cut_strat=0, lazy_pool=1, cut_select=0, cut_chance=1.0, deal="owner", slack=None):
    """Serial-equivalent FIFO branch-and-bound over all ranks of `group`.

    `root` is this rank's handle of the (identical) root problem.  Returns the same dictionary as
    mvolps_amd.bnb.branch_and_bound, identical on every rank, plus `dist` (migration statistics).
    deal="owner": children stay on their parent's rank unless its share of their window is full (default);
    deal="roundrobin": round 1's dealing, kept to measure the migrated-bytes ratio against.
    slack: how many nodes beyond per_rank a rank may hold in one window before a child is sent away (default
    per_rank // 4, at least 1): moving a tableau costs about as much as solving the node, so a little imbalance
    is cheaper than the traffic that would remove it.
    """
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    api = engine.api
    cdev = engine.comm_device

    def allreduce_max(t):
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return t.cpu().numpy()

    parent, prune, bound = {1: 0}, {1: NONE}, {1: float("inf")}
    events = []
    queue = deque([_Node(1, 0, float("inf"), True)])
    local = {}
    if rank == 0:
        local[1] = root.copy(capi.ON)  # S1 = NodeData(prob), bs.cpp:80
    next_id = 2
    child_seq = 0
    # bs.cpp:172,210 compare as a maximiser; the repaired mode turns the compares round for a minimisation problem
    sg = -1.0 if (not quirks and api.get_obj_dir(root.h) == capi.MIN) else 1.0
    best_lower, has_inc, inc_oid, inc_owner = -sg * float("inf"), 0, 0, 0
    x_keep = {}
    count, hit_limit, total_pivots = 0, 0, 0
    n0 = root.n
    stop_all = False
    W_full = world * per_rank
    if slack is None:
        slack = max(1, per_rank // 4)
    round_no = 0
    win_load = {}  # absolute window number -> nodes per rank already placed in it
    stats = {"world": world, "per_rank": per_rank, "deal": deal, "slack": slack, "children": 0, "migrated": 0, "migrated_bytes": 0, "rounds": 0, "allreduces": 0}
    cut_params = None
    if cut_strat:
        cut_params = dict(var_strat=var_strat, cut_strat=cut_strat, quirks=quirks, lazy_pool=lazy_pool, cut_select=cut_select,
                          cut_chance=cut_chance)
    widths = []  # nodes per round: how much of the farm a tree can keep busy (a narrow tree measures the all-reduce latency)

    while queue and not stop_all:
        if max_nodes > 0 and count >= max_nodes:
            hit_limit = 1
            break
        W = min(len(queue), world * per_rank)
        widths.append(W)
        window = [queue[i] for i in range(W)]

        # ---- A. the window (bs.cpp:114-117, printInfo bs.cpp:135|151).  Every node but the root was classified by the
        # rank that created it (section C) and carries the result; only a window with a node not yet classified is
        # solved here and agreed on by an all-reduce (every rank sees the same queue: every rank takes the same branch)
        if all(nd.cls is not None for nd in window):
            A = [nd.cls for nd in window]
        else:
            A = torch.full((W, 6), NEG_INF, dtype=torch.float64)
            mine = []
            for w, nd in enumerate(window):
                if nd.owner != rank:
                    continue
                # bs.cpp:114-116 copies the node's problem into a scratch and solves the copy; the node is
                # discarded after this round either way, so its own clone is solved in place
                a = local[nd.oid]
                mine.append((w, nd, a, a.it_cnt))
            engine.solve_many([a for (_, _, a, _) in mine])
            for w, nd, a, before in mine:
                st, obj, nviol, acc, pick = engine.classify(a, root, quirks, var_strat)
                A[w] = torch.tensor([float(st), obj, float(nviol), acc, float(pick), float(a.it_cnt - before)], dtype=torch.float64)
                if st == 1:
                    x_keep[nd.oid] = a.col_prim()
            for nd in window:
                nd.solver = nd.owner
            A = allreduce_max(A.to(cdev))
            stats["allreduces"] += 1

        # ---- B. replay the serial decisions in queue order
        per_node_events = []
        branch_list = []
        processed = 0
        for w, nd in enumerate(window):
            if max_nodes > 0 and count >= max_nodes:
                hit_limit = 1
                stop_all = True
                break
            st, obj, nv, acc, pick, piv = int(A[w][0]), float(A[w][1]), int(A[w][2]), float(A[w][3]), int(A[w][4]), int(A[w][5])
            total_pivots += piv
            ev = [(EV_PREGNANT, nd.oid, obj, 0.0, 0, 0)]
            per_node_events.append(ev)
            processed += 1
            if nd.inital:
                if st == -1:  # bs.cpp:139-143
                    prune[nd.oid] = FEAS
                    stop_all = True
                    break
                if st == 1:  # bs.cpp:144-149 (leaves without recording the solution; repaired mode keeps it)
                    nd.upper = obj
                    bound[nd.oid] = obj
                    prune[nd.oid] = INTG
                    if not quirks:
                        best_lower, has_inc, inc_oid, inc_owner = obj, 1, nd.oid, nd.solver
                    stop_all = True
                    break
            nd.upper = obj
            bound[nd.oid] = obj
            if st == 1:
                prune[nd.oid] = INTG
                ev.append((EV_INTEGER, nd.oid, obj, 0.0, 0, 0))
                if sg * obj > sg * best_lower:
                    best_lower, has_inc, inc_oid, inc_owner = obj, 1, nd.oid, nd.solver
            elif st == -1:
                prune[nd.oid] = FEAS
                ev.append((EV_INFEASIBLE, nd.oid, 0.0, 0.0, 0, 0))
            elif sg * obj <= sg * best_lower:
                prune[nd.oid] = BNDS
                ev.append((EV_FATHOMED, nd.oid, 0.0, 0.0, 0, 0))
            else:
                s2, s3 = next_id, next_id + 1
                next_id += 2
                for c in (s2, s3):
                    parent[c], prune[c], bound[c] = nd.oid, NONE, float("inf")
                ev.append((EV_BRANCHED, nd.oid, obj, acc, nv, pick))
                branch_list.append((nd, s2, s3, pick, ev))
                if count > loop_limit:  # bs.cpp:320-323
                    hit_limit = 1
                    count += 1
                    stop_all = True
                    break
            count += 1

        # ---- C. owners create and solve the children (bs.cpp:269-288)
        # one row per branching: both bounds, pivots, image bytes, then 6 + 6 for the two children's own window step
        C = torch.full((max(1, len(branch_list)), 16), NEG_INF, dtype=torch.float64)
        fresh = {}
        made = []
        for k, (nd, s2, s3, pick, ev) in enumerate(branch_list):
            if nd.owner != rank:
                continue
            a = local[nd.oid]
            if cut_params is not None and engine.node_cuts(a, cut_params) < 0:  # bs.cpp:249-258
                raise RuntimeError("node %d generated no cut: bs.cpp would re-add a cut pooled by an earlier node "
                                   "(cut.cpp:16-21), which the coordinator does not carry between ranks" % nd.oid)
            S2, S3 = engine.make_children(a, pick, quirks)  # bs.cpp:261-282
            made.append((k, s2, s3, S2, S3, S2.it_cnt, S3.it_cnt))
        # every child of this round is an independent LP (bs.cpp:279,287): solve them together
        kids = [p for (_, _, _, S2, S3, _, _) in made for p in (S2, S3)]
        engine.solve_many(kids)
        first = [(S2.obj, S3.obj, float((S2.it_cnt - b2) + (S3.it_cnt - b3))) for (_, _, _, S2, S3, b2, b3) in made]
        # ... and their own window step right away: the solve bs.cpp:117 repeats when a node is popped (no pivots unless
        # the first one ended infeasible or unbounded) and printInfo -- all that a later round needs to know of them
        before2 = [p.it_cnt for p in kids]
        engine.solve_many(kids)
        for i, (k, s2, s3, S2, S3, b2, b3) in enumerate(made):
            row = list(first[i]) + [0.0]
            for side, (oid, S) in enumerate(((s2, S2), (s3, S3))):
                st, obj, nviol, acc, pick = engine.classify(S, root, quirks, var_strat)
                row += [float(st), obj, float(nviol), acc, float(pick), float(S.it_cnt - before2[2 * i + side])]
                if st == 1:
                    x_keep[oid] = S.col_prim()
            row[3] = float(engine.pack_size(S2, root))  # after the window step: what travels is the node as popped
            C[k] = torch.tensor(row, dtype=torch.float64)
            fresh[s2], fresh[s3] = S2, S3
        C = allreduce_max(C.to(cdev))
        stats["allreduces"] += 1

        # ---- D. publish the children, give each an owner, migrate the ones that change ranks
        sends, recvs = [], []
        base = len(queue) - processed  # nodes that stay queued after this round's pops
        n_kids = 2 * len(branch_list)
        spread_cap = -(-(base + n_kids) // world) if base + n_kids <= W_full else None  # whole queue fits one window
        # Which children leave their parent's rank.  Per future window (rounds pop W_full nodes each from the front,
        # so a child's window follows from its queue position) and per parent rank: the children beyond that rank's
        # remaining share are taken EVENLY out of the run, not off its tail.  Under FIFO order a node's children
        # sit side by side, so ranks own runs of consecutive queue positions that double every level; thinning a
        # run evenly interleaves the ranks again, and the doubled runs of the next levels fit their windows.
        kid_owner = []
        if deal != "roundrobin":
            groups = {}
            for k, (nd, s2, s3, pick, ev) in enumerate(branch_list):
                for t in range(2):
                    kn = 2 * k + t
                    wno = round_no + 1 + (base + kn) // W_full
                    groups.setdefault((wno, nd.owner), []).append(kn)
                    kid_owner.append(nd.owner)
            movers = []
            for (wno, r), kids in sorted(groups.items()):
                load = win_load.setdefault(wno, [0] * world)
                cap = spread_cap if spread_cap is not None else per_rank + slack
                keep = max(0, min(len(kids), cap - load[r]))
                move = len(kids) - keep
                load[r] += keep
                for t in range(move):  # evenly spaced positions of the run
                    movers.append((kids[(2 * t + 1) * len(kids) // (2 * move)], wno))
            for kn, wno in sorted(movers):
                load = win_load[wno]
                dst = min(range(world), key=lambda r: (load[r], r))
                load[dst] += 1
                kid_owner[kn] = dst
        kid_no = 0
        for k, (nd, s2, s3, pick, ev) in enumerate(branch_list):
            ub2, ub3, piv, nbytes = float(C[k][0]), float(C[k][1]), int(C[k][2]), int(C[k][3])
            total_pivots += piv
            for side, (oid, ub) in enumerate(((s2, ub2), (s3, ub3))):
                owner = child_seq % world if deal == "roundrobin" else kid_owner[kid_no]
                child_seq += 1
                kid_no += 1
                stats["children"] += 1
                bound[oid] = ub
                queue.append(_Node(oid, owner, ub, cls=[float(v) for v in C[k][4 + 6 * side:10 + 6 * side]], solver=nd.owner))
                ev.append((EV_CANDIDATE, oid, ub, 0.0, 0, 0))
                if owner != nd.owner:
                    stats["migrated"] += 1
                    stats["migrated_bytes"] += nbytes
                if nd.owner == rank and owner == rank:
                    local[oid] = fresh[oid]
                elif nd.owner == rank:
                    sends.append((oid, owner, engine.pack(fresh[oid], root)))
                elif owner == rank:
                    recvs.append((oid, nd.owner, engine.recv_buffer(nbytes)))
        if world > 1 and (sends or recvs):
            ops = [dist.P2POp(dist.isend, t, dst, group=group) for (_, dst, t) in sends]
            ops += [dist.P2POp(dist.irecv, t, src, group=group) for (_, src, t) in recvs]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            for oid, _, t in recvs:
                local[oid] = engine.unpack(root, t)
        win_load.pop(round_no, None)
        round_no += 1
        stats["rounds"] += 1
        for ev in per_node_events:
            events.extend(ev)
        for _ in range(processed):
            nd = queue.popleft()
            local.pop(nd.oid, None)
        # solutions kept for integral nodes: the incumbent's, and those of nodes still waiting in the queue
        if len(x_keep) > (1 if has_inc else 0):
            waiting = {nd.oid for nd in queue}
            for oid in list(x_keep):
                if oid != inc_oid and oid not in waiting:
                    del x_keep[oid]

    # incumbent solution (bs.cpp:181-187) from the rank that solved it
    x = torch.zeros(n0, dtype=torch.float64)
    if has_inc and rank == inc_owner:
        x = torch.from_numpy(np.ascontiguousarray(x_keep[inc_oid][:n0]))
    if world > 1 and has_inc:
        xt = x.to(cdev)
        dist.broadcast(xt, src=inc_owner if group is None else dist.get_global_rank(group, inc_owner), group=group)
        x = xt.cpu()
    if widths:
        ws = sorted(widths)
        stats["nodes_per_round"] = {"min": ws[0], "median": ws[len(ws) // 2], "max": ws[-1], "window_capacity": world * per_rank}
    nn = next_id - 1
    out_events = [(t, oid, parent[oid], branch_direction(oid), f6, f7, f8, pk) for (t, oid, f6, f7, f8, pk) in events]
    return {
        "n_nodes": nn,
        "parent": [parent[i] for i in range(1, nn + 1)],
        "prune": [prune[i] for i in range(1, nn + 1)],
        "node_bound": [bound[i] for i in range(1, nn + 1)],
        "events": out_events,
        "count": count,
        "has_incumbent": has_inc,
        "best_lower": best_lower,
        "incumbent_oid": inc_oid,
        "x": [float(v) for v in x.tolist()],
        "total_pivots": total_pivots,
        "hit_limit": hit_limit,
        "dist": stats,
    }
