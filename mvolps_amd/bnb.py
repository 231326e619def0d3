"""Python binding of the branch-and-bound driver (include/mvx_bnb.h).

branch_and_bound(prob, ...) mirrors `int branchAndBound(glp_prob*, MVOLP::ParameterObj&)`
(/root/reference/bs.h:7); the strategy arguments are ParameterObj's (-vs / -bs / -cm flags,
/root/reference/2test.cpp:85-149).
"""
import ctypes as C

from . import capi

_FIELDS = [
    "create_prob", "erase_prob", "delete_prob", "copy_prob", "add_rows", "set_mat_row", "set_row_bnds", "set_col_bnds",
    "simplex", "get_status", "get_obj_val", "get_obj_coef", "get_col_prim", "get_num_rows", "get_num_cols", "get_col_kind",
    "get_col_stat", "get_row_stat", "get_row_ub", "get_row_lb", "get_col_ub", "get_col_lb", "get_col_type", "get_mat_row", "eval_tab_row",
    "get_it_cnt",
]
_OPTIONAL = ["simplex_batch", "get_obj_dir", "gmi_cuts", "gmi_cuts_many", "get_col_prim_all"]


class LpApiTable(C.Structure):
    """struct mvx_lp_api: the GLPK-shaped function table the driver calls through."""

    _fields_ = [(name, C.c_void_p) for name in _FIELDS + _OPTIONAL]


class BnbParams(C.Structure):
    _fields_ = [
        ("var_strat", C.c_int),
        ("node_strat", C.c_int),
        ("cut_strat", C.c_int),
        ("cut_chance", C.c_double),
        ("loop_limit", C.c_int),
        ("max_nodes", C.c_int),
        ("reference_quirks", C.c_int),
        ("lazy_pool", C.c_int),
        ("cut_select", C.c_int),
        ("window", C.c_int),
    ]


class BnbEvent(C.Structure):
    _fields_ = [
        ("type", C.c_int),
        ("oid", C.c_int),
        ("pid", C.c_int),
        ("direction", C.c_int),
        ("lp_bound", C.c_double),
        ("sum_infeas", C.c_double),
        ("n_violated", C.c_int),
        ("pick", C.c_int),
    ]


class BnbResult(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int),
        ("parent", C.POINTER(C.c_int)),
        ("prune", C.POINTER(C.c_int)),
        ("node_bound", C.POINTER(C.c_double)),
        ("n_events", C.c_int),
        ("events", C.POINTER(BnbEvent)),
        ("count", C.c_int),
        ("has_incumbent", C.c_int),
        ("best_lower", C.c_double),
        ("incumbent_oid", C.c_int),
        ("n", C.c_int),
        ("x", C.POINTER(C.c_double)),
        ("total_pivots", C.c_longlong),
        ("hit_limit", C.c_int),
    ]


def table_from(api):
    """Build an mvx_lp_api table out of any library that exports the ABI under api.prefix."""
    t = LpApiTable()
    for name in _FIELDS:
        fn = getattr(api.lib, api.prefix + name)
        setattr(t, name, C.cast(fn, C.c_void_p).value)
    for name in _OPTIONAL:  # optional entries stay NULL when the library does not export them
        fn = getattr(api.lib, api.prefix + name, None) if hasattr(api.lib, api.prefix + name) else None
        setattr(t, name, C.cast(fn, C.c_void_p).value if fn is not None else None)
    return t


def _bind(lib):
    lib.mvx_hip_lp_api.restype = C.c_void_p
    lib.mvx_bnb_default_params.argtypes = [C.POINTER(BnbParams)]
    lib.mvx_branchAndBound.restype = C.c_int
    lib.mvx_branchAndBound.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BnbParams), C.POINTER(BnbResult)]
    lib.mvx_bnb_free_result.argtypes = [C.POINTER(BnbResult)]
    lib.mvx_getFract.restype = C.c_double
    lib.mvx_getFract.argtypes = [C.c_double]
    lib.mvx_printInfo.restype = C.c_int
    lib.mvx_printInfo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mvx_bnb_classify.restype = C.c_int
    lib.mvx_bnb_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.mvx_bnb_make_children.restype = C.c_int
    lib.mvx_bnb_make_children.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.mvx_bnb_node_cuts.restype = C.c_int
    lib.mvx_bnb_node_cuts.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BnbParams)]
    lib.mvx_generateCut3.restype = C.c_int
    lib.mvx_generateCut3.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        from . import load_library

        _lib = _bind(load_library())
    return _lib


def result_to_dict(res):
    nn = res.n_nodes
    return {
        "n_nodes": nn,
        "parent": [res.parent[i] for i in range(1, nn + 1)],
        "prune": [res.prune[i] for i in range(1, nn + 1)],
        "node_bound": [res.node_bound[i] for i in range(1, nn + 1)],
        "events": [
            (e.type, e.oid, e.pid, e.direction, e.lp_bound, e.sum_infeas, e.n_violated, e.pick)
            for e in (res.events[k] for k in range(res.n_events))
        ],
        "count": res.count,
        "has_incumbent": res.has_incumbent,
        "best_lower": res.best_lower,
        "incumbent_oid": res.incumbent_oid,
        "x": [res.x[j] for j in range(1, res.n + 1)],
        "total_pivots": res.total_pivots,
        "hit_limit": res.hit_limit,
    }


def make_params(var_strat=0, node_strat=0, cut_strat=0, max_nodes=0, quirks=1, lazy_pool=1, window=None, cut_select=0, cut_chance=1.0):
    """mvx_bnb_params with ParameterObj's defaults (util.h:65-67) overridden by the arguments."""
    pr = BnbParams()
    lib().mvx_bnb_default_params(C.byref(pr))
    pr.var_strat, pr.node_strat, pr.cut_strat, pr.max_nodes = var_strat, node_strat, cut_strat, max_nodes
    pr.reference_quirks, pr.lazy_pool = quirks, lazy_pool
    pr.cut_select, pr.cut_chance = cut_select, cut_chance
    if window is not None:
        pr.window = window
    return pr


def branch_and_bound(prob, var_strat=0, node_strat=0, cut_strat=0, max_nodes=0, quirks=1, lazy_pool=1, table=None, window=None,
                     cut_select=0, cut_chance=1.0):
    """Run the driver on `prob` (a capi.Prob).  table=None uses the gfx950 engine's own table."""
    L = lib()
    pr = make_params(var_strat, node_strat, cut_strat, max_nodes, quirks, lazy_pool, window, cut_select, cut_chance)
    res = BnbResult()
    tptr = C.cast(C.pointer(table), C.c_void_p) if table is not None else None
    L.mvx_branchAndBound(tptr, prob.h, C.byref(pr), C.byref(res))
    out = result_to_dict(res)
    L.mvx_bnb_free_result(C.byref(res))
    return out


def print_info(prob, quirks=1, table=None):
    import numpy as np

    L = lib()
    n = prob.n
    buf = np.zeros(n + 1, dtype=np.int32)
    cnt = C.c_int(0)
    tptr = C.cast(C.pointer(table), C.c_void_p) if table is not None else None
    st = L.mvx_printInfo(tptr, prob.h, quirks, buf.ctypes.data_as(C.POINTER(C.c_int)), C.byref(cnt))
    return st, buf[: cnt.value].tolist()


def generate_cut3(prob, j, table=None):
    import numpy as np

    L = lib()
    n = prob.n
    inds = np.zeros(n + 1, dtype=np.int32)
    vals = np.zeros(n + 1, dtype=np.float64)
    lb = C.c_double(0.0)
    tptr = C.cast(C.pointer(table), C.c_void_p) if table is not None else None
    rc = L.mvx_generateCut3(tptr, prob.h, j, inds.ctypes.data_as(C.POINTER(C.c_int)), vals.ctypes.data_as(C.POINTER(C.c_double)), C.byref(lb))
    if rc != 0:
        return None
    return inds, vals, lb.value


def node_cuts(a, params, table=None):
    """bs.cpp:249-258 on one solved node about to be branched: append its GMI cut row(s); returns their number
    (-1: bug-compatible mode and the node generated none).  `params`: keyword arguments of make_params."""
    pr = make_params(**params)
    tptr = C.cast(C.pointer(table), C.c_void_p) if table is not None else None
    return lib().mvx_bnb_node_cuts(tptr, a.h, C.byref(pr))


def classify(prob, root, quirks=1, var_strat=0, table=None):
    """(status, objective, n_violated, sum_fract, pick) of a solved node in one call."""
    out = (C.c_double * 5)()
    tptr = C.cast(C.pointer(table), C.c_void_p) if table is not None else None
    lib().mvx_bnb_classify(tptr, prob.h, root.h, quirks, var_strat, out)
    return int(out[0]), out[1], int(out[2]), out[3], int(out[4])


def make_children(a, pick, quirks=1, table=None):
    """The two branching clones of bs.cpp:269-282 (bounds set, not yet solved)."""
    S2, S3 = a.api.create(), a.api.create()
    tptr = C.cast(C.pointer(table), C.c_void_p) if table is not None else None
    lib().mvx_bnb_make_children(tptr, a.h, pick, quirks, S2.h, S3.h)
    return S2, S3
