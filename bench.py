#!/usr/bin/env python3
"""bench.py -- simplex pivots/s on a dense fp64 tableau, with the HBM roofline of the rank-1
update kernel and a same-host CPU baseline (BASELINE.json metric).

One "step" = one simplex pivot on the synthetic dense LP of BASELINE.md config 4: m=4096, n=8192, fp64, splitmix64 seed
12345 (+rank).  The pivots are chained: k_chain chooses up to 32 of them from O(m+n) slices of the tableau as it stands,
then ONE pass over the tableau (k_fbc3, the streamed Gauss-Jordan update: HBM-bound) applies them all, k_fpatch writes
the chain's pivot rows / columns.  The tableau is resident in HBM before the timed region starts; a run longer than the
LP's ~3.4k pivots to optimality carries on with a device-to-device clone of the initial tableau.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N>1: every rank solves its own LP of the same shape (seed 12345 + rank) -- independent
subproblems, no data-path collective (weak scaling).  Rank 0 prints ONE JSON line.  `python bench.py --gpus N`
without a launcher starts the N ranks itself (torch.distributed.run as a child process, before anything touches
the GPU); a WORLD_SIZE that disagrees with --gpus is an error -- the line never reports another n_gpus than asked.
At N>1 the line also carries `secondary.bnb_ilp_512x1024_dist`: the B&B node farm (mvolps_amd/dist_bnb.py,
/root/reference/bs.cpp:96-327 sharded over the ranks) on the BASELINE config-5 instance over RCCL.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (spec 8.0 TB/s)


def bytes_per_pivot(m, n):
    """Algorithmic bytes of one rank-1 update: every tableau entry read once + written once (fp64)."""
    return 16 * (m + 1) * (n + 1)


def pmc_traffic(m, n):
    """HBM bytes per bulk launch from the committed rocprofv3 PMC passes (profiles/*traffic*.json, written by
    scripts/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for wide coalesced reads on gfx950).  A constant shipped with the repo, NOT measured in this run:
    returns (bytes, source file) of the newest matching record, (None, None) when there is none."""
    import glob

    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("m") == m and d.get("n") == n and d.get("bytes_per_launch"):
            best, src = d["bytes_per_launch"], os.path.relpath(f, ROOT)
    return best, src


def glpk_probe(A, b, c, budget_pivots=400):
    """Opportunistic true-reference line (CPU-baseline plan 2): if a libglpk happens to be
    installed on this host, time real glp_simplex on the same LP.  Never assumed; any failure -> absent."""
    import ctypes as C
    import ctypes.util

    try:
        name = ctypes.util.find_library("glpk")
        if not name:
            return {"glpk": "absent"}
        g = C.CDLL(name)
        g.glp_create_prob.restype = C.c_void_p
        g.glp_get_obj_val.restype = C.c_double
        g.glp_version.restype = C.c_char_p
        m, n = A.shape
        P = C.c_void_p(g.glp_create_prob())
        g.glp_term_out(0)
        g.glp_set_obj_dir(P, 2)
        g.glp_add_rows(P, m)
        g.glp_add_cols(P, n)
        for j in range(n):
            g.glp_set_col_bnds(P, j + 1, 2, C.c_double(0.0), C.c_double(0.0))
            g.glp_set_obj_coef(P, j + 1, C.c_double(float(c[j])))
        ind = (C.c_int * (n + 1))(*range(n + 1))
        for i in range(m):
            g.glp_set_row_bnds(P, i + 1, 3, C.c_double(0.0), C.c_double(float(b[i])))
            val = (C.c_double * (n + 1))(0.0, *A[i].tolist())
            g.glp_set_mat_row(P, i + 1, n, ind, val)
        t0 = time.perf_counter()
        g.glp_simplex(P, None)
        el = time.perf_counter() - t0
        it = getattr(g, "glp_get_it_cnt", None)
        iters = int(it(P)) if it else None
        out = {"glpk": g.glp_version().decode(), "seconds": el, "objective": float(g.glp_get_obj_val(P)), "iterations": iters}
        g.glp_delete_prob(P)
        return out
    except Exception as e:  # pragma: no cover - depends on the host
        return {"glpk": "probe failed: %s" % e}


def cpu_baseline(m, n, seed, budget_s=9.0, max_pivots=400):
    """Oracle (CPU restatement) on a bounded sample of the same LP: all host threads this job may use
    (OpenMP row-parallel update) and, beside it, one thread."""
    import ctypes as C

    threads = min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from mvolps_amd import synth
    from oracle import oracle

    A, b, c = synth.dense_lp(m, n, seed)
    P = oracle.api().create()
    P.load_dense(A, b, c)
    P.simplex(it_lim=2)  # builds the tableau + touches the pages
    try:
        gomp = C.CDLL("libgomp.so.1")
    except OSError:
        gomp = None

    def run(nthreads, budget):
        if gomp is not None:
            gomp.omp_set_num_threads(nthreads)
        piv0 = P.it_cnt
        t0 = time.perf_counter()
        while True:
            P.simplex(it_lim=10)
            el = time.perf_counter() - t0
            if el >= budget or P.it_cnt - piv0 >= max_pivots or P.status == 5:
                break
        return (P.it_cnt - piv0), el

    done, el = run(threads, budget_s)
    done1, el1 = run(1, budget_s * 0.6) if gomp is not None else (0, 1.0)
    out = {
        "value": done / el,
        "unit": "pivots/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d pivots of the same %dx%d LP (seed %d) right after its first 2, oracle/mvolps_oracle.c, %d OpenMP threads"
        % (done, m, n, seed, threads),
        "single_thread": {"value": done1 / el1, "pivots": done1} if done1 else None,
    }
    out["reference_glpk"] = glpk_probe(A, b, c) if m * n <= 2048 * 4096 else glpk_probe(A[:1024, :2048].copy(), b[:1024], c[:2048])
    return out


def config5_fixture():
    """tests/golden/config5.json: the calibrated config-5 instance and the oracle's digests of its tree (data only;
    None when the file is not there)."""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "config5.json")))
    except Exception:
        return None


def config5_instance():
    from mvolps_amd import synth

    fx = config5_fixture() or {"m": 512, "n": 1024, "seed": 12345, "U": 1.0, "cap": 0.002, "reference_quirks": 0}
    A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx.get("cap", 0.4))
    return fx, (A, b, c, U)


def bnb_cpu_baseline(inst, quirks, budget_nodes=150):
    """Oracle restatement of bs.cpp on the oracle's LP engine, same instance, first `budget_nodes` nodes."""
    from mvolps_amd import synth
    from oracle import oracle

    threads = min(os.cpu_count() or 1, 16)
    orc = oracle.api()
    A, b, c, U = inst
    t0 = time.perf_counter()
    r = oracle.branch_and_bound(synth.load_ilp(orc, A, b, c, U), quirks=quirks, max_nodes=budget_nodes)
    el = time.perf_counter() - t0
    return {"value": r["count"] / el, "unit": "nodes/s", "cores": threads, "kind": "port", "pivots_per_s": r["total_pivots"] / el,
            "sample": "first %d nodes of the same FIFO tree, oracle/mvolps_oracle_bnb.c (one node at a time; OpenMP row-parallel "
                      "pivots, %d threads)" % (r["count"], threads)}


def secondary(api, with_cpu=True):
    """Other BASELINE configs on the same GPU, reported beside the headline (not part of `value`)."""
    from mvolps_amd import bnb, dist_bnb, synth, treedigest

    out = {}
    # config 2: dense LP 1024x2048 -- cache-resident, latency-bound (report, do not headline)
    A, b, c = synth.dense_lp(1024, 2048, 12345)
    P = api.create()
    P.load_dense(A, b, c)
    P.simplex(it_lim=50)
    api.sync()
    t0 = time.perf_counter()
    P.simplex(it_lim=600)
    api.sync()
    el = time.perf_counter() - t0
    out["dense_lp_1024x2048"] = {"pivots_per_s": 600 / el, "us_per_pivot": el / 600 * 1e6,
                                 "frac_of_hbm_roofline": 600 / el * bytes_per_pivot(1024, 2048) / 1e9 / HBM_PEAK_GBS}
    # configs 3/5 shape: ILP 512x1024, FIFO B&B through mvx_branchAndBound in window mode (64 node LPs share each launch)
    # and through the multi-rank coordinator on one rank (same tree, Python replay of the decisions).  Two instances:
    # the WIDE one of round 1 (seed 12345, U = 3, cap = 0.4: the breadth-first tree doubles every level, thousands of
    # open nodes -- throughput with every batch slot busy) and the CALIBRATED config-5 one (tests/golden/config5.json:
    # cap = 0.002, a tree that closes after ~15.7k nodes but is only tens of nodes wide).
    A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
    nodes = 2000
    # warm-up = the run itself once: the frontier of this tree is a thousand 4 MB node tableaux, and the first run of a
    # process allocates them (slab arenas, batch contexts); the timed run is the steady state a long B&B job is in
    bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes)
    t0 = time.perf_counter()
    r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes)
    el = time.perf_counter() - t0
    out["bnb_ilp_512x1024"] = {"driver": "mvx_branchAndBound", "instance": "wide (cap 0.4, U 3)", "nodes": r["count"], "nodes_per_s": r["count"] / el,
                               "pivots": r["total_pivots"], "pivots_per_s": r["total_pivots"] / el, "window": 64}
    eng = dist_bnb.HipNodeEngine(0)
    t0 = time.perf_counter()
    r2 = dist_bnb.branch_and_bound(eng, synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, per_rank=64)
    el = time.perf_counter() - t0
    out["bnb_ilp_512x1024_coordinator"] = {"driver": "mvolps_amd.dist_bnb (1 rank)", "nodes": r2["count"], "nodes_per_s": r2["count"] / el,
                                           "pivots": r2["total_pivots"], "same_tree": treedigest.digest(r2) == treedigest.digest(r), "per_rank": 64}
    from mvolps_amd import dist_native

    t0 = time.perf_counter()
    r3 = dist_native.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, per_rank=64)
    el = time.perf_counter() - t0
    out["bnb_ilp_512x1024_native_coordinator"] = {"driver": "mvx_branchAndBound_dist (C++, 1 rank, no communicator)", "nodes": r3["count"],
                                                  "nodes_per_s": r3["count"] / el, "pivots": r3["total_pivots"],
                                                  "same_tree": treedigest.digest(r3) == treedigest.digest(r), "per_rank": 64}
    if with_cpu:
        out["bnb_ilp_512x1024"]["cpu_baseline"] = bnb_cpu_baseline((A, b, c, U), 0)
    # config 3: the same ILP with GMI cuts (gmi.cpp / cut.cpp path): bug-compatible (bs.cpp as written: the pool's last cut
    # at every branching node) and repaired; one cut per branching node, made for the whole window in one device pass
    cuts = {}
    for name, kw in (("bug_compatible", dict(quirks=1, cut_strat=1)), ("repaired", dict(quirks=0, cut_strat=1))):
        bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), max_nodes=64, **kw)  # warm-up
        t0 = time.perf_counter()
        rc = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), max_nodes=600, **kw)
        el = time.perf_counter() - t0
        cuts[name] = {"nodes": rc["count"], "nodes_per_s": rc["count"] / el, "pivots": rc["total_pivots"], "window": 64}
    out["bnb_ilp_512x1024_cuts"] = {"driver": "mvx_branchAndBound", "instance": "wide (cap 0.4, U 3), first 600 nodes, one GMI cut per branching node", **cuts}
    if with_cpu:
        from oracle import oracle

        t0 = time.perf_counter()
        ro = oracle.branch_and_bound(synth.load_ilp(oracle.api(), A, b, c, U), quirks=1, cut_strat=1, max_nodes=40)
        el = time.perf_counter() - t0
        out["bnb_ilp_512x1024_cuts"]["cpu_baseline"] = {"value": ro["count"] / el, "unit": "nodes/s", "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                                                        "sample": "first %d nodes of the bug-compatible tree, oracle/mvolps_oracle_bnb.c" % ro["count"]}
    fx, inst = config5_instance()
    if fx.get("full"):
        A5, b5, c5, U5 = inst
        q = int(fx.get("reference_quirks", 0))
        t0 = time.perf_counter()
        r5 = bnb.branch_and_bound(synth.load_ilp(api, A5, b5, c5, U5), quirks=q)
        el = time.perf_counter() - t0
        out["bnb_config5_tree"] = {"driver": "mvx_branchAndBound", "instance": "calibrated config 5 (cap %g, U %g): the whole tree" % (fx["cap"], fx["U"]),
                                   "nodes": r5["count"], "nodes_per_s": r5["count"] / el, "pivots": r5["total_pivots"], "best": r5["best_lower"],
                                   "incumbent_updates": treedigest.incumbent_updates(r5),
                                   "same_tree_as_oracle_fixture": treedigest.digest(r5) == fx["full"]["sha256"]}
    return out


def dist_bnb_leg(api, dist, rank, world, dev_index, rehearsal, nodes, publish=lambda res: None, note=lambda phase: None):
    """secondary.bnb_ilp_512x1024_dist (every rank calls this): the serial-equivalent node farm over the process
    group -- node LPs sharded over the ranks (bs.cpp:96-327), MAX all-reduces for the incumbent and the child bounds,
    RCCL send/recv for the children that change ranks.  Two instances, as in secondary(): the wide 512x1024 tree
    (first `nodes` nodes; children dealt to their parent's rank, and round-robin as round 1 did for comparison) and the
    calibrated config-5 tree run to its end and checked against the oracle's record."""
    import torch

    from mvolps_amd import bnb, dist_bnb, synth, treedigest

    eng = dist_bnb.HipNodeEngine(dev_index, comm_device="cpu" if rehearsal else None)
    A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
    note("warm-up")
    dist_bnb.branch_and_bound(eng, synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=4 * world * 16, per_rank=16)  # warm-up

    def timed(inst, quirks, max_nodes, deal):
        Ai, bi, ci, Ui = inst
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = dist_bnb.branch_and_bound(eng, synth.load_ilp(api, Ai, bi, ci, Ui), quirks=quirks, max_nodes=max_nodes, per_rank=64, deal=deal)
        dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        st = r["dist"]
        return r, {"nodes": r["count"], "nodes_per_s": r["count"] / el, "pivots": r["total_pivots"], "seconds": el, "children": st["children"],
                   "migrated_images": st["migrated"], "migrated_share": st["migrated"] / max(1, st["children"]),
                   "migrated_bytes_per_node": st["migrated_bytes"] / max(1, r["count"]), "rounds": st["rounds"], "allreduces": st.get("allreduces"),
                   "nodes_per_round": st.get("nodes_per_round")}

    note("wide tree, children on their parent's rank")
    r_own, own = timed((A, b, c, U), 0, nodes, "owner")
    note("wide tree, children dealt round-robin")
    r_rr, rr = timed((A, b, c, U), 0, nodes, "roundrobin")
    res = dict(own)
    res["instance"] = "wide (cap 0.4, U 3), first %d nodes" % nodes
    res["same_tree_both_dealings"] = treedigest.digest(r_own) == treedigest.digest(r_rr)
    if rank == 0:  # the single-GPU driver on the same prefix, for the tree only
        r1 = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes)
        res["same_tree_as_single_gpu_driver"] = treedigest.digest(r1) == treedigest.digest(r_own)
    res["roundrobin_dealing"] = {k: rr[k] for k in ("nodes_per_s", "migrated_images", "migrated_share", "migrated_bytes_per_node")}
    fx, inst5 = config5_instance()
    if fx.get("full"):
        note("config-5 tree")
        r5, c5 = timed(inst5, int(fx.get("reference_quirks", 0)), 0, "owner")
        c5["instance"] = "calibrated config 5 (cap %g, U %g): the whole tree" % (fx["cap"], fx["U"])
        c5["same_tree_as_oracle_fixture"] = treedigest.digest(r5) == fx["full"]["sha256"]
        c5["best"] = r5["best_lower"]
        res["config5_tree"] = c5
    res["driver"] = "mvolps_amd.dist_bnb"
    res["ranks"] = world
    res["per_rank"] = 64
    res["collective"] = {"backend": "gloo (rehearsal on one GPU)" if rehearsal else "nccl (RCCL over xGMI)", "ranks": world,
                         "ops_per_round": "1 MAX all-reduce (children's bounds and their own window step together) + send/recv of migrated node images"}
    # The same farm through the C++ entry (mvx_branchAndBound_dist, include/mvx_dist.h) with RCCL called directly from
    # libmvolps_rccl.so (its own communicator; torch.distributed only hands the id round).  What has been measured so
    # far is published first: should this part stall on a node it has not met, the line still carries the rest.
    res["native_coordinator"] = {"error": "did not finish"}
    publish(dict(res))
    if os.environ.get("MVX_BENCH_NO_NATIVE_DIST") != "1":
        note("native coordinator (mvx_branchAndBound_dist)")
        try:
            from mvolps_amd import dist_native

            comm = dist_native.TorchComm(device_buffers=True) if rehearsal else dist_native.RcclComm(rank, world)
            try:
                dist_native.branch_and_bound(synth.load_ilp(api, A, b, c, U), comm=comm, quirks=0, max_nodes=4 * world * 16, per_rank=16)  # warm-up
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rn = dist_native.branch_and_bound(synth.load_ilp(api, A, b, c, U), comm=comm, quirks=0, max_nodes=nodes, per_rank=64)
                dist.barrier()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                t = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
                st = rn["dist"]
                res["native_coordinator"] = {
                    "driver": "mvx_branchAndBound_dist", "transport": "torch.distributed gloo through callbacks (rehearsal)" if rehearsal else "RCCL from C++ (libmvolps_rccl.so)",
                    "nodes": rn["count"], "nodes_per_s": rn["count"] / el, "seconds": el, "migrated_images": st["migrated"],
                    "migrated_share": st["migrated"] / max(1, st["children"]), "same_tree_as_python_coordinator": treedigest.digest(rn) == treedigest.digest(r_own),
                    "same_dealing": st["migrated"] == own["migrated_images"]}
            finally:
                comm.close()
        except Exception as e:
            res["native_coordinator"] = {"error": repr(e)}
    else:
        res["native_coordinator"] = {"skipped": "MVX_BENCH_NO_NATIVE_DIST=1"}
    return res


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a child process tree
    (torch.distributed.run), before this process has touched the GPU, and leave with its exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MVX_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--rows", type=int, default=4096, help="constraints m of the dense LP")
    ap.add_argument("--cols", type=int, default=8192, help="columns n of the dense LP")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the side measurements of configs 2 and 3/5")
    ap.add_argument("--profile-steps", type=int, default=200, help="extra pivots timed per-kernel with HIP events")
    ap.add_argument("--bnb-nodes", type=int, default=3000, help="nodes of the wide 512x1024 tree the distributed B&B leg runs (N>1)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # rehearsal on a one-GPU box only: MVX_BENCH_REHEARSAL=1 puts every rank on device 0 and uses
    # gloo for the barrier / MAX-reduce, to exercise the multi-rank code path without N GPUs
    rehearsal = os.environ.get("MVX_BENCH_REHEARSAL") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one.  Nothing here initialises the GPU (torch.cuda.device_count() only counts devices).
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus and not rehearsal:
            raise SystemExit("--gpus %d asked for, %d HIP device(s) visible (MVX_BENCH_REHEARSAL=1 rehearses the %d-rank path "
                             "on one GPU over gloo)" % (args.gpus, have, args.gpus))
        raise SystemExit(self_launch(args))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch with `python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...`, or run `python bench.py --gpus %d` alone"
                         % (world, args.gpus, args.gpus, args.gpus, args.gpus))
    import mvolps_amd

    mvolps_amd.require_device()
    api = mvolps_amd.api()
    dev_index = 0 if rehearsal else local_rank
    if api.set_device(dev_index) != 0:
        raise SystemExit("cannot bind device %d" % dev_index)
    torch.cuda.set_device(dev_index)
    dist = None
    # MVX_BENCH_FORCE_DIST=1: take the process-group path at world size 1 too (checks the RCCL barrier /
    # all-reduce plumbing on a one-GPU box; run it under torch.distributed.run --nproc-per-node 1)
    if world > 1 or os.environ.get("MVX_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))

    from mvolps_amd import synth

    m, n = args.rows, args.cols
    A, b, c = synth.dense_lp(m, n, args.seed + rank)
    P0 = api.create()
    P0.load_dense(A, b, c)
    del A
    P0.simplex(it_lim=0)  # builds the tableau in HBM (host build + one upload), no pivots
    state = {"P": P0.copy(), "device_ms": 0.0}

    def run_pivots(k):
        """Exactly k pivots of the workload, HBM-resident: when the LP reaches its optimum first (devex pricing
        needs ~3.4k pivots at 4096x8192) the run carries on with a fresh device-to-device clone of the
        initial tableau."""
        left = k
        state["device_ms"] = 0.0
        while left > 0:
            P = state["P"]
            before = P.it_cnt
            P.simplex(it_lim=left)
            state["device_ms"] += api.last_solve_ms(P.h)
            left -= P.it_cnt - before
            if left > 0:
                if P.status != 5:
                    raise SystemExit("rank %d: LP ended with status %d" % (rank, P.status))
                state["P"] = P0.copy()
        return k

    # warmup: W untimed pivots
    run_pivots(args.warmup)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    steps_done = run_pivots(args.steps)
    barrier()
    el = time.perf_counter() - t0
    device_ms = state["device_ms"]

    # max over ranks
    el_max = el
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_max = float(t.item())

    # per-kernel pass: HIP events around every k_fb launch, on the engine's own stream.  It runs on the
    # same handle right after the timed region (the next pivots of the same LP) rather than inside it:
    # two event records per pivot cost ~6 us/pivot (scripts/evcost.py), which would distort `value`.
    roof = None
    if rank == 0:
        def timed_passes(pivots_wanted):
            api.profile_reset()
            api.profile_enable(1)
            done = run_pivots(pivots_wanted)
            api.profile_enable(0)
            # events were recorded around every queued bulk launch (k_fbc3 alone); only launches that stepped count
            return done, api.profile_update_ms(), api.profile_update_launches()

        pivots, k_ms, k_n = timed_passes(args.profile_steps)
        if pivots > 0 and k_ms > 0 and k_n > 0:
            # one bulk launch reads and writes the tableau once, whether it applies one pivot or a chain of them
            avg_ms = k_ms / k_n
            achieved = bytes_per_pivot(m, n) / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(m, n)
            roof = {
                "bound": "hbm",
                "kernel": "k_fbc3 (streamed Gauss-Jordan update: one pass over the tableau per launch, every step of the chain applied in registers)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": ("%s (committed rocprofv3 PMC pass, not measured in this run)" % traffic_src) if traffic_src else None,
                "avg_launch_us": avg_ms * 1e3,
                "launches_timed": k_n,
                "pivots_timed": pivots,
                "pivots_per_launch": pivots / k_n,
                "bytes_per_launch": bytes_per_pivot(m, n),
                "clock": "HIP events around each launch.  rocprofv3 --kernel-trace reads ~13 us more for the same chained launches "
                         "(profiles/r03_bench_4096x8192_kernel_stats.csv: k_fbc3<8,2> 117 us): with launches queued back to back every "
                         "kernel's start stamp there is its predecessor's end stamp (all gaps 0.0 in the trace), so the hand-over between "
                         "kernels is inside the duration; with an idle queue in front (the rank1 leg) the two clocks agree (76.5 / 78.5 us)",
            }
            # the same kernel with ONE pivot per pass (the rank-1 update the north star prices: MVX_CHAIN=1), timed in
            # this run over >= 50 launches
            try:
                api.set_chain(1)
                run_pivots(5)
                p1, ms1, n1 = timed_passes(60)
                if p1 > 0 and ms1 > 0 and n1 > 0:
                    a1 = bytes_per_pivot(m, n) / (ms1 / n1 * 1e-3) / 1e9
                    roof["rank1"] = {"achieved": a1, "frac": a1 / HBM_PEAK_GBS, "avg_launch_us": ms1 / n1 * 1e3, "launches_timed": n1,
                                     "what": "k_fbc3 with one pivot per pass (mvx_set_chain(1)): the plain rank-1 update"}
            finally:
                api.set_chain(0)

    dist_leg = None
    leg_hung = False
    if dist is not None and world > 1 and os.environ.get("MVX_BENCH_NO_DIST_BNB") != "1":
        # The B&B farm is a side measurement: it must never cost the headline line.  It runs on a helper thread with a
        # deadline; if a rank stalls in a collective (a transport problem on a node this code has not met), every rank
        # gives up at its own deadline, rank 0 prints the line with the leg marked as timed out, and the process leaves
        # without waiting for the stuck collective.
        import threading

        box = {}

        def leg():
            try:
                torch.cuda.set_device(dev_index)
                api.bind_thread()
                box["res"] = dist_bnb_leg(api, dist, rank, world, dev_index, rehearsal, args.bnb_nodes, lambda part: box.__setitem__("part", part), lambda ph: box.__setitem__("phase", ph))
            except Exception as e:  # the failure modes that raise are deterministic (fixture, engine): every rank raises
                box["res"] = {"error": repr(e)}

        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(float(os.environ.get("MVX_BENCH_DIST_DEADLINE", "240")))
        if th.is_alive():
            leg_hung = True
            dist_leg = dict(box.get("part") or {})
            dist_leg["error"] = "timed out after %s s (a rank stalled in a collective); headline unaffected" % os.environ.get("MVX_BENCH_DIST_DEADLINE", "240")
        else:
            dist_leg = box.get("res")

    out = None
    if rank == 0:
        value = world * args.steps / el_max
        ppl = roof["pivots_per_launch"] if roof else 1.0
        out = {
            "metric": "simplex_pivots_per_s",
            "value": value,
            "unit": "pivots/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "dense LP m=%d n=%d fp64, splitmix64 seed %d+rank, primal simplex pivots on the (m+1)x(n+1) tableau" % (m, n, args.seed),
                "m": m,
                "n": n,
                # one pass over the tableau (read + write every entry) applies `pivots_per_pass` pivots: the chained
                # path chooses the pivots after the first from O(m+n) slices, so the bytes a pivot needs are the pass
                # divided by the chain length the run actually reached
                "bytes_per_pass": bytes_per_pivot(m, n),
                "pivots_per_pass": ppl,
                "bytes_per_pivot": bytes_per_pivot(m, n) / ppl,
                "parallelism": "independent LP per GPU (x%d)" % world,
            },
            "pivot_roofline_frac": (args.steps / el_max) * (bytes_per_pivot(m, n) / ppl) / 1e9 / HBM_PEAK_GBS,
            "device_ms_per_step": device_ms / args.steps,
            "roofline": roof,
            "collective": None if dist is None else {"backend": "gloo (rehearsal on one GPU)" if rehearsal else "nccl (RCCL over xGMI)",
                                                     "ranks": world, "data_path": "none: independent LPs (barrier + MAX of the wall time only)"},
        }
        if world == 1 and not args.no_secondary:
            try:
                out["secondary"] = secondary(api, with_cpu=not args.no_cpu_baseline)
            except Exception as e:  # the headline line must not depend on the side measurements
                out["secondary"] = {"error": str(e)}
        if dist_leg is not None:
            out["secondary"] = {"bnb_ilp_512x1024_dist": dist_leg}
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(m, n, args.seed)
        print(json.dumps(out), flush=True)
    sys.stdout.flush()
    if leg_hung:
        sys.stderr.write("bench.py rank %d: the distributed B&B leg did not finish before its deadline (stuck in: %s); the headline line was printed first\n"
                         % (rank, (box.get("phase") or "unknown phase")))
        sys.stderr.flush()
        os._exit(3)  # a collective is stuck on the helper thread: leave without joining it, and say so in the exit code
    if dist is not None:
        # leave together, but never wait for a peer that gave up on the side leg
        import threading

        def bye():
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:
                pass

        th = threading.Thread(target=bye, daemon=True)
        th.start()
        th.join(60.0)
        if th.is_alive():
            sys.stderr.write("bench.py rank %d: the farewell barrier timed out\n" % rank)
            sys.stderr.flush()
            os._exit(4)


if __name__ == "__main__":
    main()
