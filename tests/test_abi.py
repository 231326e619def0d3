"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, and its model layer
(no engine call) behaves like GLPK's; engine entry points refuse to run without a device."""
import ctypes
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import mvolps_amd
from mvolps_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(mvx_[a-zA-Z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = mvolps_amd.load_library()
    syms = declared_symbols()
    assert len(syms) >= 50
    # mvx_rccl_* (include/mvx_dist.h) live in libmvolps_rccl.so, so that the engine library does not depend on librccl
    rccl = ctypes.CDLL(os.path.join(ROOT, "mvolps_amd", "lib", "libmvolps_rccl.so"))
    missing = [s for s in syms if not hasattr(rccl if s.startswith("mvx_rccl_") else lib, s)]
    assert not missing, missing
    assert [s for s in syms if s.startswith("mvx_rccl_")] == ["mvx_rccl_comm_create", "mvx_rccl_comm_destroy", "mvx_rccl_unique_id"]


def test_model_layer_without_engine():
    api = mvolps_amd.api()
    A, b, c = synth.dense_lp(5, 7, 3)
    P = api.create()
    P.load_dense(A, b, c)
    assert (P.m, P.n) == (5, 7)
    assert api.get_obj_dir(P.h) == capi.MAX
    ind, val = P.get_mat_row(3)
    assert list(ind) == list(range(1, 8)) and np.array_equal(val, A[2])
    assert api.get_row_type(P.h, 1) == capi.UP and api.get_row_ub(P.h, 1) == b[0]
    assert api.get_row_lb(P.h, 1) == -sys.float_info.max  # absent bound reads back as -DBL_MAX
    assert api.get_col_ub(P.h, 2) == sys.float_info.max
    assert api.get_col_stat(P.h, 1) == capi.NL and api.get_row_stat(P.h, 1) == capi.BS
    assert api.get_status(P.h) == capi.UNDEF
    assert api.get_col_prim(P.h, 1) == 0.0
    # GLPK's BV convention: an integer column with bounds [0,1] reports GLP_BV
    api.set_col_kind(P.h, 2, capi.IV)
    assert api.get_col_kind(P.h, 2) == capi.IV and api.get_num_int(P.h) == 1
    api.set_col_bnds(P.h, 2, capi.DB, 0.0, 1.0)
    assert api.get_col_kind(P.h, 2) == capi.BV
    # rows share storage between clones until one of them is rewritten
    Q = P.copy()
    r = api.add_rows(Q.h, 1)
    assert r == 6 and Q.m == 6 and P.m == 5
    Q.set_mat_row(2, np.array([0, 1, 4], dtype=np.int32), np.array([0.0, 2.0, -1.0]))
    assert list(Q.get_mat_row(2)[0]) == [1, 4]
    assert np.array_equal(P.get_mat_row(2)[1], A[1])
    api.erase_prob(Q.h)
    assert Q.m == 0 and Q.n == 0
    # the row list and the column names are shared between clones too (copy-on-write): writes on either side part them
    api.set_col_name(P.h, 3, b"x3")
    R = P.copy()
    S = P.copy(capi.OFF)
    assert api.get_col_name(R.h, 3) == b"x3" and api.get_col_name(S.h, 3) is None
    api.set_col_name(R.h, 3, b"y3")
    api.set_col_name(P.h, 4, b"x4")
    assert api.get_col_name(P.h, 3) == b"x3" and api.get_col_name(R.h, 3) == b"y3"
    assert api.get_col_name(R.h, 4) is None and api.get_col_name(P.h, 4) == b"x4"
    assert api.add_rows(P.h, 2) == 6 and (P.m, R.m, S.m) == (7, 5, 5)
    R.set_mat_row(5, np.array([0, 2], dtype=np.int32), np.array([0.0, 9.0]))
    assert np.array_equal(P.get_mat_row(5)[1], A[4]) and np.array_equal(S.get_mat_row(5)[1], A[4])
    assert list(R.get_mat_row(5)[0]) == [2] and list(P.get_mat_row(7)[0]) == []
    del R
    assert np.array_equal(P.get_mat_row(1)[1], A[0]) and np.array_equal(S.get_mat_row(1)[1], A[0])


def test_engine_refuses_without_device():
    """No GPU here: the product must fail loudly instead of computing on the CPU."""
    if mvolps_amd.api().device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(RuntimeError):
        mvolps_amd.require_device()
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, mvolps_amd\n"
        "from mvolps_amd import synth\n"
        "P = mvolps_amd.api().create(); P.load_dense(*synth.dense_lp(3, 4, 1)); P.simplex()\n" % ROOT
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


def test_product_never_references_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    bad = []
    for path in glob.glob(os.path.join(ROOT, "mvolps_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
            txt = open(path, errors="ignore").read()
            # comments may mention the oracle; linking, loading, importing or calling it may not happen
            if re.search(r"liboracle|orc_\w+\s*\(|#include\s*\".*oracle|import\s+oracle|from\s+oracle|oracle\.py", txt):
                bad.append(path)
    assert not bad, bad
