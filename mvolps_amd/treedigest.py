"""Digest of a branch-and-bound result (oracle's or the product's): every event field (doubles as hex strings,
so bitwise), the prune labels and the parent array.  Two runs with the same digest took the same decisions in the
same order with the same LP bounds."""
import hashlib


def digest(r):
    h = hashlib.sha256()
    for e in r["events"]:
        h.update(repr((e[0], e[1], e[2], e[3], float(e[4]).hex(), float(e[5]).hex(), e[6], e[7])).encode())
    h.update(repr(list(r["prune"])).encode())
    h.update(repr(list(r["parent"])).encode())
    return h.hexdigest()


def summary(r):
    """What a fixture records about a run besides its digest."""
    return {
        "count": r["count"],
        "n_nodes": r["n_nodes"],
        "pivots": r["total_pivots"],
        "events": len(r["events"]),
        "hit_limit": r["hit_limit"],
        "has_incumbent": r["has_incumbent"],
        "incumbent_oid": r["incumbent_oid"],
        "best_lower": float(r["best_lower"]).hex() if r["has_incumbent"] else None,
        "incumbent_updates": incumbent_updates(r),
        "prune_counts": {str(k): list(r["prune"]).count(k) for k in (0, 1, 3, 4)},
        "depth": depth(r),
        "sha256": digest(r),
    }


def incumbent_updates(r):
    """Number of times bs.cpp:172-174 replaced the incumbent (integer events whose bound beat every earlier one)."""
    best, k = None, 0
    for e in r["events"]:
        if e[0] == 1 and (best is None or e[4] > best):
            best, k = e[4], k + 1
    return k


def depth(r):
    d = [0] * (r["n_nodes"] + 1)
    for oid in range(2, r["n_nodes"] + 1):
        d[oid] = d[r["parent"][oid - 1]] + 1
    return max(d) if d else 0
