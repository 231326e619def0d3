// capi.cpp -- the C ABI (include/mvx.h): problem model on the host, GLPK-shaped edit and
// query semantics, dispatch into the device engine.  Replaces the glp_* surface MVOLPS
// binds; each function cites its reference call site in mvx.h.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>

#include "engine.hpp"
#include "../../include/mvx_dist.h"

using mvx::RowPtr;

static const double INF = HUGE_VAL;
static int g_term_out = 1;

static void fault(const char *msg) {
  // GLPK aborts on invalid arguments [GLPK-recalled]; same contract here
  std::fprintf(stderr, "mvx: %s\n", msg);
  std::abort();
}

static void norm_bounds(int type, double lb, double ub, double *olb, double *oub) {
  switch (type) {
    case MVX_FR: *olb = -INF; *oub = INF; break;
    case MVX_LO: *olb = lb; *oub = INF; break;
    case MVX_UP: *olb = -INF; *oub = ub; break;
    case MVX_DB: *olb = lb; *oub = ub; break;
    case MVX_FX: *olb = lb; *oub = lb; break;
    default: fault("invalid bound type");
  }
}
static int std_flag(int type) {
  switch (type) {
    case MVX_FR: return MVX_NF;
    case MVX_LO: return MVX_NL;
    case MVX_UP: return MVX_NU;
    case MVX_DB: return MVX_NL;
    default: return MVX_NS;
  }
}
static double nb_value(int flag, double lb, double ub) {
  switch (flag) {
    case MVX_NL: return lb;
    case MVX_NU: return ub;
    case MVX_NS: return lb;
    default: return 0.0;
  }
}

static void reset_model(mvx_prob *P) {
  P->m = P->n = 0;
  P->dir = MVX_MIN; // GLPK default [GLPK-recalled]
  P->A.reset(1);
  P->c.assign(1, 0.0);
  P->kind.assign(1, 0);
  P->cname.mut().clear();
  P->rtype.assign(1, 0); P->rlb.assign(1, 0.0); P->rub.assign(1, 0.0);
  P->ctype.assign(1, 0); P->clb.assign(1, 0.0); P->cub.assign(1, 0.0);
  P->valid = false;
  P->status = MVX_UNDEF;
  P->it_cnt = 0;
  P->piv_since_check = 0;
  P->refresh_cnt = 0;
  P->bland_cnt = 0;
  P->pert_cnt = 0;
  P->last_ms = 0.0;
  P->bvar.clear(); P->nvar.clear(); P->nflag.clear(); P->pos.clear();
  P->pending.clear();
  P->dmat.reset();
  P->sol_fresh = false;
  P->fresh_rows = -1;
  P->beta.clear(); P->dj.clear();
}

extern "C" {

mvx_prob *mvx_create_prob(void) {
  mvx_prob *P = new mvx_prob();
  reset_model(P);
  return P;
}

void mvx_erase_prob(mvx_prob *P) {
  mvx::release_device(P);
  reset_model(P);
}

void mvx_delete_prob(mvx_prob *P) {
  if (!P) return;
  mvx::release_device(P);
  delete P;
}

namespace {
// MVX_API_TIMING=1: host time of the row-edit entry points a cut goes through (printed when the library unloads)
struct ApiTiming {
  bool on = std::getenv("MVX_API_TIMING") != nullptr;
  double t[4] = {0, 0, 0, 0};
  long calls[4] = {0, 0, 0, 0};
  ~ApiTiming() {
    static const char *names[4] = {"add_rows", "set_mat_row", "set_row_bnds", "eval_tab_row"};
    if (on)
      for (int k = 0; k < 4; k++)
        if (calls[k]) std::fprintf(stderr, "mvx_%s: %ld calls, %.1f us each\n", names[k], calls[k], 1e6 * t[k] / calls[k]);
  }
} g_api_timing;
struct ApiTimer {
  int k;
  double t0;
  explicit ApiTimer(int kk) : k(kk), t0(g_api_timing.on ? std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0.0) {}
  ~ApiTimer() {
    if (!g_api_timing.on) return;
    g_api_timing.t[k] += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
    g_api_timing.calls[k]++;
  }
};
} // namespace

namespace {
// MVX_COPY_TIMING=1: where a clone's host time goes (printed when the library unloads)
struct CopyTiming {
  bool on = std::getenv("MVX_COPY_TIMING") != nullptr;
  double host = 0, engine = 0;
  long calls = 0;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  ~CopyTiming() {
    if (on && calls) std::fprintf(stderr, "mvx_copy_prob: %ld calls, host fields %.1f us, engine_copy %.1f us each\n", calls, 1e6 * host / calls, 1e6 * engine / calls);
  }
} g_copy_timing;
} // namespace

void mvx_copy_prob(mvx_prob *dst, const mvx_prob *src, int names) {
  if (dst == src) fault("copy_prob: dst == src");
  const double t0 = g_copy_timing.on ? CopyTiming::now() : 0.0;
  mvx::release_device(dst);
  // host model: matrix rows are shared (immutable, copy-on-write in set_mat_row)
  dst->m = src->m; dst->n = src->n; dst->dir = src->dir;
  dst->A = src->A;
  dst->c = src->c; dst->kind = src->kind;
  if (names) dst->cname = src->cname; else dst->cname = mvx::CowVec<std::string>();
  dst->rtype = src->rtype; dst->rlb = src->rlb; dst->rub = src->rub;
  dst->ctype = src->ctype; dst->clb = src->clb; dst->cub = src->cub;
  dst->status = src->status; dst->it_cnt = src->it_cnt; dst->bland_cnt = src->bland_cnt; dst->pert_cnt = src->pert_cnt; dst->last_ms = 0.0;
  dst->hint_dual = src->hint_dual;
  dst->piv_since_check = src->piv_since_check;
  dst->refresh_cnt = src->refresh_cnt;
  dst->pending = src->pending;
  dst->dmat = src->dmat;
  std::memcpy(dst->last_tol, src->last_tol, sizeof(dst->last_tol));
  dst->bvar = src->bvar; dst->nvar = src->nvar; dst->nflag = src->nflag; dst->pos = src->pos;
  dst->sol_fresh = src->sol_fresh; dst->fresh_rows = src->fresh_rows; dst->beta = src->beta; dst->dj = src->dj;
  const double t1 = g_copy_timing.on ? CopyTiming::now() : 0.0;
  mvx::engine_copy(dst, src);
  if (g_copy_timing.on) {
    g_copy_timing.host += t1 - t0;
    g_copy_timing.engine += CopyTiming::now() - t1;
    g_copy_timing.calls++;
  }
}

void mvx_set_obj_dir(mvx_prob *P, int dir) {
  if (dir != MVX_MIN && dir != MVX_MAX) fault("set_obj_dir: invalid direction");
  P->dir = dir;
  P->status = MVX_UNDEF;
}

int mvx_add_rows(mvx_prob *P, int nrs) {
  ApiTimer timer_(0);
  if (nrs < 1) fault("add_rows: invalid count");
  const int first = P->m + 1;
  for (int r = 0; r < nrs; r++) {
    P->A.push_back(std::make_shared<std::vector<double>>((size_t)P->n + 1, 0.0));
    P->rtype.push_back(MVX_FR);
    P->rlb.push_back(-INF);
    P->rub.push_back(INF);
  }
  P->m += nrs;
  if (P->valid) mvx::engine_add_rows(P, first, nrs);
  P->status = MVX_UNDEF;
  return first;
}

int mvx_add_cols(mvx_prob *P, int ncs) {
  if (ncs < 1) fault("add_cols: invalid count");
  const int first = P->n + 1;
  P->n += ncs;
  P->c.resize((size_t)P->n + 1, 0.0);
  P->kind.resize((size_t)P->n + 1, MVX_CV);
  P->ctype.resize((size_t)P->n + 1, MVX_FX); // GLPK default column: fixed at 0 [GLPK-recalled]
  P->clb.resize((size_t)P->n + 1, 0.0);
  P->cub.resize((size_t)P->n + 1, 0.0);
  if (!P->cname.empty()) P->cname.mut().resize((size_t)P->n + 1);
  for (int i = 1; i <= P->m; i++) {
    auto row = std::make_shared<std::vector<double>>(*P->A[i]);
    row->resize((size_t)P->n + 1, 0.0);
    P->A.set((size_t)i, row);
  }
  if (P->valid) mvx::engine_invalidate(P);
  P->status = MVX_UNDEF;
  return first;
}

void mvx_set_row_bnds(mvx_prob *P, int i, int type, double lb, double ub) {
  ApiTimer timer_(2);
  if (i < 1 || i > P->m) fault("set_row_bnds: row out of range");
  const double olb = P->rlb[i], oub = P->rub[i];
  P->rtype[i] = type;
  norm_bounds(type, lb, ub, &P->rlb[i], &P->rub[i]);
  if (P->valid) mvx::engine_apply_bounds(P, i, type, olb, oub, P->rlb[i], P->rub[i]);
  P->status = MVX_UNDEF;
}

void mvx_set_col_bnds(mvx_prob *P, int j, int type, double lb, double ub) {
  if (j < 1 || j > P->n) fault("set_col_bnds: column out of range");
  const double olb = P->clb[j], oub = P->cub[j];
  P->ctype[j] = type;
  norm_bounds(type, lb, ub, &P->clb[j], &P->cub[j]);
  if (P->valid) mvx::engine_apply_bounds(P, P->m + j, type, olb, oub, P->clb[j], P->cub[j]);
  P->status = MVX_UNDEF;
}

void mvx_set_obj_coef(mvx_prob *P, int j, double coef) {
  if (j < 0 || j > P->n) fault("set_obj_coef: column out of range");
  P->c[j] = coef;
  if (P->valid) mvx::engine_recompute_cost_row(P);
  P->status = MVX_UNDEF;
}

void mvx_set_mat_row(mvx_prob *P, int i, int len, const int *ind, const double *val) {
  ApiTimer timer_(1);
  if (i < 1 || i > P->m) fault("set_mat_row: row out of range");
  if (len < 0 || len > P->n) fault("set_mat_row: invalid length");
  auto row = std::make_shared<std::vector<double>>((size_t)P->n + 1, 0.0);
  for (int k = 1; k <= len; k++) {
    if (ind[k] < 1 || ind[k] > P->n) fault("set_mat_row: column index out of range");
    (*row)[ind[k]] = val[k];
  }
  P->A.set((size_t)i, row);
  if (P->valid) {
    if (P->pos[i] <= 0) mvx::engine_invalidate(P); // row of a non-basic auxiliary changed
    else mvx::engine_row_from_model(P, i);
  }
  P->status = MVX_UNDEF;
}

void mvx_set_col_kind(mvx_prob *P, int j, int kind) {
  if (j < 1 || j > P->n) fault("set_col_kind: column out of range");
  if (kind == MVX_BV) {
    P->kind[j] = MVX_IV;
    mvx_set_col_bnds(P, j, MVX_DB, 0.0, 1.0);
  } else if (kind == MVX_CV || kind == MVX_IV) {
    P->kind[j] = kind;
  } else
    fault("set_col_kind: invalid kind");
}

void mvx_set_col_name(mvx_prob *P, int j, const char *name) {
  if (j < 1 || j > P->n) fault("set_col_name: column out of range");
  if (P->cname.size() < (size_t)P->n + 1) P->cname.mut().resize((size_t)P->n + 1);
  P->cname.mut()[j] = name ? name : "";
}

const char *mvx_get_col_name(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_name: column out of range");
  if (P->cname.size() < (size_t)P->n + 1 || P->cname[j].empty()) return nullptr;
  return P->cname[j].c_str();
}

int mvx_load_dense(mvx_prob *P, int m, int n, const double *A, const double *b, const double *c) {
  if (m < 1 || n < 1 || !A || !b || !c) return -1;
  mvx_erase_prob(P);
  P->dir = MVX_MAX;
  P->n = n;
  P->c.assign((size_t)n + 1, 0.0);
  std::memcpy(&P->c[1], c, (size_t)n * sizeof(double));
  P->kind.assign((size_t)n + 1, MVX_CV);
  P->ctype.assign((size_t)n + 1, MVX_LO);
  P->clb.assign((size_t)n + 1, 0.0);
  P->cub.assign((size_t)n + 1, INF);
  P->m = m;
  P->A.resize((size_t)m + 1);
  P->rtype.assign((size_t)m + 1, MVX_UP);
  P->rlb.assign((size_t)m + 1, -INF);
  P->rub.assign((size_t)m + 1, 0.0);
  for (int i = 1; i <= m; i++) {
    auto row = std::make_shared<std::vector<double>>((size_t)n + 1, 0.0);
    std::memcpy(row->data() + 1, A + (size_t)(i - 1) * n, (size_t)n * sizeof(double));
    P->A.set((size_t)i, row);
    P->rub[i] = b[i - 1];
  }
  return 0;
}

// Tolerances behind `parm == NULL` / mvx_init_smcp.  GLPK's own defaults are 1e-7 / 1e-7 / 1e-9 [GLPK-recalled]; this
// engine's are 1e-9 throughout so that the objective of the dense LPs meets 1e-9 relative against the HiGHS goldens
// (a reduced cost of 5e-8 left un-entered moves the optimum by more than that).  mvx_set_default_tolerances puts
// GLPK's values (or any others) behind every NULL-parameter call, which is all MVOLPS ever makes (bs.cpp:117,279,287).
static double g_tol_bnd = 1e-9, g_tol_dj = 1e-9, g_tol_piv = 1e-9;

void mvx_set_default_tolerances(double tol_bnd, double tol_dj, double tol_piv) {
  if (!(tol_bnd > 0.0) || !(tol_dj > 0.0) || !(tol_piv > 0.0)) fault("set_default_tolerances: tolerances must be positive");
  g_tol_bnd = tol_bnd;
  g_tol_dj = tol_dj;
  g_tol_piv = tol_piv;
}

void mvx_init_smcp(mvx_smcp *parm) {
  parm->msg_lev = 0;
  parm->meth = 1;
  parm->it_lim = -1;
  parm->tol_bnd = g_tol_bnd;
  parm->tol_dj = g_tol_dj;
  parm->tol_piv = g_tol_piv;
}

// a model that has just been loaded (or a lineage that has collected many cut rows) holds its rows in its own tail:
// fold them into a shared head before the handle starts being cloned (mvx_internal.hpp: RowList)
static inline void settle_rows(mvx_prob *P) {
  if (P->A.tail_size() > 16) P->A.freeze();
}
int mvx_simplex(mvx_prob *P, const mvx_smcp *parm) {
  settle_rows(P);
  return mvx::engine_simplex(P, parm);
}
int mvx_simplex_batch(mvx_prob **probs, int count, const mvx_smcp *parm, int *rcs) {
  for (int k = 0; k < count; k++)
    if (probs && probs[k]) settle_rows(probs[k]);
  return mvx::engine_simplex_batch(probs, count, parm, rcs);
}

int mvx_get_obj_dir(const mvx_prob *P) { return P->dir; }
int mvx_get_num_rows(const mvx_prob *P) { return P->m; }
int mvx_get_num_cols(const mvx_prob *P) { return P->n; }
int mvx_get_num_int(const mvx_prob *P) {
  int k = 0;
  for (int j = 1; j <= P->n; j++) k += (P->kind[j] == MVX_IV);
  return k;
}
int mvx_get_status(const mvx_prob *P) { return P->status; }

double mvx_get_obj_val(const mvx_prob *P) {
  if (!P->valid) return P->c[0];
  if (!P->sol_fresh && P->fresh_rows >= 0) return P->beta[0]; // only rows behind row 0 have been appended / rewritten since the export
  mvx::refresh_solution(P);
  return P->beta[0];
}
double mvx_get_obj_coef(const mvx_prob *P, int j) {
  if (j < 0 || j > P->n) fault("get_obj_coef: column out of range");
  return P->c[j];
}

static double var_prim(const mvx_prob *P, int k) {
  if (!P->valid) return 0.0;
  // A variable that is basic in a row the edits since the last export have not touched (cut rows are appended behind
  // it: bs.cpp:249-261 adds a cut and then reads the branching variable's value) is read from the mirror as it stands;
  // anything else exports the solution first.
  if (!P->sol_fresh && !(P->fresh_rows >= 0 && P->pos[k] > 0 && P->pos[k] <= P->fresh_rows)) mvx::refresh_solution(P);
  const int pos = P->pos[k];
  if (pos > 0) return P->beta[pos];
  const double lb = (k <= P->m) ? P->rlb[k] : P->clb[k - P->m];
  const double ub = (k <= P->m) ? P->rub[k] : P->cub[k - P->m];
  return nb_value(P->nflag[-pos], lb, ub);
}
static double var_dual(const mvx_prob *P, int k) {
  if (!P->valid) return 0.0;
  mvx::refresh_solution(P);
  const int pos = P->pos[k];
  return pos > 0 ? 0.0 : P->dj[-pos];
}
static int var_stat(const mvx_prob *P, int k, int type) {
  if (!P->valid) return (k <= P->m) ? MVX_BS : std_flag(type);
  const int pos = P->pos[k];
  return pos > 0 ? MVX_BS : P->nflag[-pos];
}

double mvx_get_col_prim(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_prim: column out of range");
  return var_prim(P, P->m + j);
}
void mvx_get_col_prim_all(const mvx_prob *P, double *x) { // x[1..n], x[0] untouched: glp_get_col_prim for every column at once
  for (int j = 1; j <= P->n; j++) x[j] = var_prim(P, P->m + j);
}
double mvx_get_row_prim(const mvx_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_prim: row out of range");
  return var_prim(P, i);
}
double mvx_get_col_dual(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_dual: column out of range");
  return var_dual(P, P->m + j);
}
double mvx_get_row_dual(const mvx_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_dual: row out of range");
  return var_dual(P, i);
}
int mvx_get_col_stat(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_stat: column out of range");
  return var_stat(P, P->m + j, P->ctype[j]);
}
int mvx_get_row_stat(const mvx_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_stat: row out of range");
  return var_stat(P, i, P->rtype[i]);
}
int mvx_get_col_kind(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_kind: column out of range");
  // GLPK reports an integer column with bounds [0,1] as GLP_BV [GLPK-recalled]
  if (P->kind[j] == MVX_IV && P->ctype[j] == MVX_DB && P->clb[j] == 0.0 && P->cub[j] == 1.0) return MVX_BV;
  return P->kind[j];
}
int mvx_get_row_type(const mvx_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_type: row out of range");
  return P->rtype[i];
}
// absent bounds read back as -/+DBL_MAX [GLPK-recalled]; consumed arithmetically at gmi.cpp:73
double mvx_get_row_lb(const mvx_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_lb: row out of range");
  return P->rlb[i] == -INF ? -DBL_MAX : P->rlb[i];
}
double mvx_get_row_ub(const mvx_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_ub: row out of range");
  return P->rub[i] == INF ? DBL_MAX : P->rub[i];
}
int mvx_get_col_type(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_type: column out of range");
  return P->ctype[j];
}
double mvx_get_col_lb(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_lb: column out of range");
  return P->clb[j] == -INF ? -DBL_MAX : P->clb[j];
}
double mvx_get_col_ub(const mvx_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_ub: column out of range");
  return P->cub[j] == INF ? DBL_MAX : P->cub[j];
}

int mvx_get_mat_row(const mvx_prob *P, int i, int *ind, double *val) {
  // non-zeros in ascending column order (GLPK's own order cannot be observed here)
  if (i < 1 || i > P->m) fault("get_mat_row: row out of range");
  const double *a = P->A[i]->data();
  int len = 0;
  for (int j = 1; j <= P->n; j++) {
    if (a[j] != 0.0) {
      len++;
      if (ind) ind[len] = j;
      if (val) val[len] = a[j];
    }
  }
  return len;
}

int mvx_eval_tab_row(const mvx_prob *P, int k, int *ind, double *val) {
  ApiTimer timer_(3);
  if (!P->valid) fault("eval_tab_row: basis does not exist");
  if (k < 1 || k > P->m + P->n) fault("eval_tab_row: variable out of range");
  const int pos = P->pos[k];
  if (pos <= 0) fault("eval_tab_row: variable must be basic");
  std::vector<double> row((size_t)P->n + 1);
  if (mvx::engine_get_row(P, pos, row.data()) != 0) fault("eval_tab_row: tableau unavailable");
  int len = 0;
  for (int j = 1; j <= P->n; j++) {
    if (row[j] != 0.0) {
      len++;
      ind[len] = P->nvar[j];
      val[len] = row[j];
    }
  }
  return len;
}

int mvx_get_it_cnt(const mvx_prob *P) { return P->it_cnt; }
int mvx_get_bland_cnt(const mvx_prob *P) { return P->bland_cnt; }
int mvx_get_pert_cnt(const mvx_prob *P) { return P->pert_cnt; }
int mvx_term_out(int flag) {
  int old = g_term_out;
  g_term_out = flag;
  return old;
}
const char *mvx_version(void) { return "mvolps-amd 0.1 (gfx950 dense simplex; GLPK-shaped API)"; }

int mvx_get_tableau_ld(const mvx_prob *P) { return P->ld; }
int mvx_get_tableau(const mvx_prob *P, double *out) { return mvx::engine_get_tableau(P, out); }
int mvx_get_basis(const mvx_prob *P, int *head, int *nb, int *flag) {
  if (!P->valid) return -1;
  head[0] = 0;
  for (int i = 1; i <= P->m; i++) head[i] = P->bvar[i];
  nb[0] = 0;
  flag[0] = 0;
  for (int j = 1; j <= P->n; j++) {
    nb[j] = P->nvar[j];
    flag[j] = P->nflag[j];
  }
  return 0;
}

int mvx_gmi_cuts(const mvx_prob *P, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok) {
  return mvx::engine_gmi_cuts(P, repaired ? 1 : 0, cols, count, vals, rhs, ok);
}
int mvx_gmi_cuts_many(const mvx_prob *const *Ps, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok) {
  return mvx::engine_gmi_cuts_many(Ps, repaired ? 1 : 0, cols, count, vals, rhs, ok);
}

int mvx_device_count(void) { return mvx::device_count(); }
int mvx_set_device(int dev) { return mvx::set_device(dev); }
long long mvx_pack_size(const mvx_prob *P) { return mvx::engine_pack_size(P, P->m); }
int mvx_pack(const mvx_prob *P, void *dev_buf) { return mvx::engine_pack(P, P->m, dev_buf); }
long long mvx_pack_size_from(const mvx_prob *P, const mvx_prob *base) { return mvx::engine_pack_size(P, base ? base->m : P->m); }
int mvx_pack_from(const mvx_prob *P, const mvx_prob *base, void *dev_buf) { return mvx::engine_pack(P, base ? base->m : P->m, dev_buf); }
int mvx_unpack(mvx_prob *dst, const mvx_prob *base, const void *dev_buf) {
  if (dst == base) return -1;
  // model rows / objective / kinds come from the receiver's own copy of the root problem
  mvx::release_device(dst);
  dst->m = base->m; dst->n = base->n; dst->dir = base->dir;
  dst->A = base->A; dst->c = base->c; dst->kind = base->kind; dst->cname = base->cname;
  dst->rtype = base->rtype; dst->rlb = base->rlb; dst->rub = base->rub;
  dst->ctype = base->ctype; dst->clb = base->clb; dst->cub = base->cub;
  dst->valid = false; dst->sol_fresh = false; dst->fresh_rows = -1;
  return mvx::engine_unpack(dst, dev_buf);
}
void mvx_set_tuning(int tr, int hot, int nt) { mvx::tuning(tr, hot, nt); }
void mvx_set_stall_limit(int limit) { mvx::set_stall_limit(limit); }
int mvx_fcs_debug_stamps(unsigned long long *out) { return mvx::fcs_debug_stamps(out); }
void mvx_set_refresh(int check_every, double tol) { mvx::set_refresh(check_every, tol); }
int mvx_get_refresh_cnt(const mvx_prob *P) { return P->refresh_cnt; }
double mvx_row_residual(const mvx_prob *P) { return mvx::row_residual(P); }
void mvx_set_persist(int mode) { mvx::set_persist(mode); }
void mvx_set_chain(int len) { mvx::set_chain(len); }
void mvx_set_cluster(int on) { mvx::set_cluster(on); }
void mvx_cluster_stats(long long *launches, long long *aborts) { mvx::cluster_stats(launches, aborts); }
void mvx_set_dual_chain(int len) { mvx::set_dual_chain(len); }
void mvx_persist_stats(long long *launches, long long *aborts) { mvx::persist_stats(launches, aborts); }
void mvx_persist_cycles(unsigned long long *out5) { mvx::persist_cycles(out5); }
void mvx_set_batch_slots(int slots) { mvx::set_batch_slots(slots); }
void mvx_profile_enable(int on) { mvx::profile_enable(on); }
void mvx_profile_reset(void) { mvx::profile_reset(); }
double mvx_profile_update_ms(void) { return mvx::profile_update_ms(); }
long long mvx_profile_update_launches(void) { return mvx::profile_update_launches(); }
double mvx_last_solve_ms(const mvx_prob *P) { return P->last_ms; }
void mvx_sync(void) { mvx::sync_stream(); }
int mvx_last_error(void) { return mvx::take_last_error(); }
int mvx_bind_thread(void) { return mvx::bind_thread(); }

// ---- include/mvx_dist.h: how a node travels between ranks, for the gfx950 engine
void *mvx_image_alloc(size_t bytes) { return mvx::engine_image_alloc(bytes); }
void mvx_image_free(void *buf) { mvx::engine_image_free(buf); }
static long long img_pack_size(const void *P, const void *base) { return mvx_pack_size_from((const mvx_prob *)P, (const mvx_prob *)base); }
static int img_pack(const void *P, const void *base, void *buf) { return mvx_pack_from((const mvx_prob *)P, (const mvx_prob *)base, buf); }
static int img_unpack(void *dst, const void *base, const void *buf) { return mvx_unpack((mvx_prob *)dst, (const mvx_prob *)base, buf); }
const mvx_image_api *mvx_hip_image_api(void) {
  static const mvx_image_api t = {img_pack_size, img_pack, img_unpack, mvx_image_alloc, mvx_image_free};
  return &t;
}

} // extern "C"
