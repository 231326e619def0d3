"""ctypes binding for a GLPK-shaped LP-engine C ABI (include/mvx.h).

The same signature table serves any library that exports the ABI under a prefix
(`mvx_` for the HIP engine).  The surface is exactly what MVOLPS binds from GLPK
(/root/reference/bs.cpp:89,114-117,274-288; cut.cpp:23,40,43;
gmi.cpp:15-52,84; util.cpp:33-41,423-455).
"""
import ctypes as C

import numpy as np

MIN, MAX = 1, 2
CV, IV, BV = 1, 2, 3
FR, LO, UP, DB, FX = 1, 2, 3, 4, 5
BS, NL, NU, NF, NS = 1, 2, 3, 4, 5
UNDEF, FEAS, INFEAS, NOFEAS, OPT, UNBND = 1, 2, 3, 4, 5, 6
OFF, ON = 0, 1
EFAIL, EITLIM = 5, 8


class Smcp(C.Structure):
    _fields_ = [
        ("msg_lev", C.c_int),
        ("meth", C.c_int),
        ("it_lim", C.c_int),
        ("tol_bnd", C.c_double),
        ("tol_dj", C.c_double),
        ("tol_piv", C.c_double),
    ]


_P = C.c_void_p
_I = C.c_int
_D = C.c_double
_IP = C.POINTER(C.c_int)
_DP = C.POINTER(C.c_double)

# name -> (restype, argtypes)
SIGNATURES = {
    "create_prob": (_P, []),
    "erase_prob": (None, [_P]),
    "delete_prob": (None, [_P]),
    "copy_prob": (None, [_P, _P, _I]),
    "set_obj_dir": (None, [_P, _I]),
    "add_rows": (_I, [_P, _I]),
    "add_cols": (_I, [_P, _I]),
    "set_row_bnds": (None, [_P, _I, _I, _D, _D]),
    "set_col_bnds": (None, [_P, _I, _I, _D, _D]),
    "set_obj_coef": (None, [_P, _I, _D]),
    "set_mat_row": (None, [_P, _I, _I, _IP, _DP]),
    "set_col_kind": (None, [_P, _I, _I]),
    "set_col_name": (None, [_P, _I, C.c_char_p]),
    "load_dense": (_I, [_P, _I, _I, _DP, _DP, _DP]),
    "init_smcp": (None, [C.POINTER(Smcp)]),
    "set_default_tolerances": (None, [_D, _D, _D]),
    "simplex": (_I, [_P, C.POINTER(Smcp)]),
    "get_obj_dir": (_I, [_P]),
    "get_num_rows": (_I, [_P]),
    "get_num_cols": (_I, [_P]),
    "get_num_int": (_I, [_P]),
    "get_status": (_I, [_P]),
    "get_obj_val": (_D, [_P]),
    "get_obj_coef": (_D, [_P, _I]),
    "get_col_prim": (_D, [_P, _I]),
    "get_row_prim": (_D, [_P, _I]),
    "get_col_dual": (_D, [_P, _I]),
    "get_row_dual": (_D, [_P, _I]),
    "get_col_stat": (_I, [_P, _I]),
    "get_row_stat": (_I, [_P, _I]),
    "get_col_kind": (_I, [_P, _I]),
    "get_row_type": (_I, [_P, _I]),
    "get_row_lb": (_D, [_P, _I]),
    "get_row_ub": (_D, [_P, _I]),
    "get_col_type": (_I, [_P, _I]),
    "get_col_lb": (_D, [_P, _I]),
    "get_col_ub": (_D, [_P, _I]),
    "get_col_name": (C.c_char_p, [_P, _I]),
    "get_mat_row": (_I, [_P, _I, _IP, _DP]),
    "eval_tab_row": (_I, [_P, _I, _IP, _DP]),
    "get_it_cnt": (_I, [_P]),
    "get_bland_cnt": (_I, [_P]),
    "get_pert_cnt": (_I, [_P]),
    "set_stall_limit": (None, [_I]),
    "term_out": (_I, [_I]),
    "version": (C.c_char_p, []),
    "get_tableau_ld": (_I, [_P]),
    "get_tableau": (_I, [_P, _DP]),
    "get_basis": (_I, [_P, _IP, _IP, _IP]),
}


def _dp(a):
    return a.ctypes.data_as(_DP)


def _ip(a):
    return a.ctypes.data_as(_IP)


class LpApi:
    """Function table of one engine library."""

    def __init__(self, lib, prefix, extra=None):
        self.lib = lib
        self.prefix = prefix
        sigs = dict(SIGNATURES)
        if extra:
            sigs.update(extra)
        for name, (res, args) in sigs.items():
            fn = getattr(lib, prefix + name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def create(self):
        return Prob(self)


class Prob:
    """Thin object wrapper; index conventions stay GLPK's (1-based)."""

    def __init__(self, api, handle=None):
        self.api = api
        self.h = handle if handle is not None else api.create_prob()
        self._own = True

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            try:
                self.api.delete_prob(self.h)
            except Exception:
                pass
            self.h = None

    def copy(self, names=ON):
        q = Prob(self.api)
        self.api.copy_prob(q.h, self.h, names)
        return q

    # --- build
    def load_dense(self, A, b, c):
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        m, n = A.shape
        assert b.shape == (m,) and c.shape == (n,)
        rc = self.api.load_dense(self.h, m, n, _dp(A), _dp(b), _dp(c))
        if rc != 0:
            raise RuntimeError("load_dense failed rc=%d" % rc)

    def load_general(self, A, row_bnds, col_bnds, c, c0=0.0, kinds=None, direction=MAX):
        """row_bnds / col_bnds: lists of (type, lb, ub)."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        m, n = A.shape
        api, h = self.api, self.h
        api.erase_prob(h)
        api.set_obj_dir(h, direction)
        api.add_cols(h, n)
        api.add_rows(h, m)
        api.set_obj_coef(h, 0, float(c0))
        for j in range(n):
            api.set_obj_coef(h, j + 1, float(c[j]))
            t, lb, ub = col_bnds[j]
            api.set_col_bnds(h, j + 1, t, float(lb), float(ub))
            if kinds is not None:
                api.set_col_kind(h, j + 1, int(kinds[j]))
        ind = np.arange(n + 1, dtype=np.int32)
        for i in range(m):
            val = np.concatenate([[0.0], A[i]])
            nz = np.nonzero(val)[0]
            ii = np.concatenate([[0], ind[nz]]).astype(np.int32)
            vv = np.concatenate([[0.0], val[nz]])
            api.set_mat_row(h, i + 1, len(nz), _ip(ii), _dp(vv))
            t, lb, ub = row_bnds[i]
            api.set_row_bnds(h, i + 1, t, float(lb), float(ub))

    def set_mat_row(self, i, ind, val):
        ind = np.ascontiguousarray(ind, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        self.api.set_mat_row(self.h, i, len(ind) - 1, _ip(ind), _dp(val))

    # --- solve
    def simplex(self, it_lim=None, meth=None, tol=None):
        if it_lim is None and meth is None and tol is None:
            return self.api.simplex(self.h, None)
        parm = Smcp()
        self.api.init_smcp(C.byref(parm))
        if it_lim is not None:
            parm.it_lim = it_lim
        if meth is not None:
            parm.meth = meth
        if tol is not None:
            parm.tol_bnd, parm.tol_dj, parm.tol_piv = tol
        return self.api.simplex(self.h, C.byref(parm))

    # --- query
    @property
    def m(self):
        return self.api.get_num_rows(self.h)

    @property
    def n(self):
        return self.api.get_num_cols(self.h)

    @property
    def status(self):
        return self.api.get_status(self.h)

    @property
    def obj(self):
        return self.api.get_obj_val(self.h)

    @property
    def it_cnt(self):
        return self.api.get_it_cnt(self.h)

    @property
    def bland_cnt(self):
        return self.api.get_bland_cnt(self.h)

    @property
    def pert_cnt(self):
        return self.api.get_pert_cnt(self.h)

    def col_prim(self):
        return np.array([self.api.get_col_prim(self.h, j) for j in range(1, self.n + 1)])

    def row_prim(self):
        return np.array([self.api.get_row_prim(self.h, i) for i in range(1, self.m + 1)])

    def col_stat(self):
        return np.array([self.api.get_col_stat(self.h, j) for j in range(1, self.n + 1)], dtype=np.int32)

    def row_stat(self):
        return np.array([self.api.get_row_stat(self.h, i) for i in range(1, self.m + 1)], dtype=np.int32)

    def tableau(self):
        m, n = self.m, self.n
        out = np.empty((m + 1, n + 1), dtype=np.float64)
        rc = self.api.get_tableau(self.h, _dp(out))
        if rc != 0:
            raise RuntimeError("no tableau")
        return out

    def basis(self):
        m, n = self.m, self.n
        head = np.zeros(m + 1, dtype=np.int32)
        nb = np.zeros(n + 1, dtype=np.int32)
        flag = np.zeros(n + 1, dtype=np.int32)
        rc = self.api.get_basis(self.h, _ip(head), _ip(nb), _ip(flag))
        if rc != 0:
            raise RuntimeError("no basis")
        return head, nb, flag

    def eval_tab_row(self, k):
        n = self.n
        ind = np.zeros(n + 1, dtype=np.int32)
        val = np.zeros(n + 1, dtype=np.float64)
        ln = self.api.eval_tab_row(self.h, k, _ip(ind), _dp(val))
        return ind[1 : ln + 1].copy(), val[1 : ln + 1].copy()

    def get_mat_row(self, i):
        n = self.n
        ind = np.zeros(n + 1, dtype=np.int32)
        val = np.zeros(n + 1, dtype=np.float64)
        ln = self.api.get_mat_row(self.h, i, _ip(ind), _dp(val))
        return ind[1 : ln + 1].copy(), val[1 : ln + 1].copy()
