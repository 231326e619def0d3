/*
 * mvolps_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See the header.
 *
 * LP core: dense condensed-tableau bounded-variable simplex standing in for
 * glp_simplex (/root/reference/bs.cpp:117,279,287).  GLPK itself is not in the image;
 * results are pinned against scipy/HiGHS fixtures (tests/golden).  "parity unpinned"
 * with respect to GLPK's own pivot sequence.
 *
 * Pivot rules (each one mirrored by mvolps_amd/csrc/kernels.hip): primal and dual devex pricing, bounded
 * ratio tests with bound flips, a phase 1 that carries its infeasibility-sum row through the pivots, bound
 * perturbation and Bland's rule against stalling.
 *
 * Determinism contract (mirrored by the HIP engine, bit for bit):
 *   - every product-sum that feeds the tableau uses fma() exactly where written here,
 *     and nowhere else (build with -ffp-contract=off);
 *   - every arg-min / arg-max has a total order (value, then magnitude, then index).
 */
#include "mvolps_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INF HUGE_VAL
#define ROWCOMB_CHUNK 64

struct orc_prob {
  int m, n, m_cap, n_cap;
  int dir;
  double **A;    /* A[i] (i=1..m) -> n_cap+1 doubles, 1-based columns */
  double *c;     /* c[0..n] */
  int *kind;     /* kind[1..n] */
  char **cname;  /* cname[1..n] or NULL */
  int *rtype;
  double *rlb, *rub; /* normalised: -INF / +INF when absent */
  int *ctype;
  double *clb, *cub;
  /* engine state */
  int valid;
  int ld;
  double *T;     /* (m_cap+1) x ld */
  int *bvar;     /* bvar[1..m] */
  double *blb, *bub;
  int *nvar;     /* nvar[1..n] */
  int *nflag;    /* ORC_NL / NU / NF / NS */
  double *nlb, *nub;
  int status;
  int it_cnt;
  int bland_cnt; /* pivots chosen by Bland's rule (diagnostic) */
  int pert_cnt;  /* bound perturbations applied (diagnostic) */
  int piv_since_check; /* pivots since the residual of A x = x_R was last looked at (clones inherit it) */
  int refresh_cnt;     /* tableau refreshes (diagnostic) */
};

/* a / b, correctly rounded -- the same function as xdiv() in mvolps_amd/csrc/kernels.hip.  On this side the
   native quotient is already the nearest double and comes back unchanged; the device's fp64 division can be one
   ulp off when the quotient lies very close to the midpoint of two doubles, and the fix-up (exact residual of q and of its neighbour, keep the
   smaller) makes both sides return the correctly rounded quotient. */
static inline double xdiv(double a, double b) {
  const double q = a / b;
  const double aq = fabs(q);
  if (!(aq > 1e-290 && aq < 1e290)) return q; /* zero, subnormal range, inf, nan */
  const double r = fma(-q, b, a);
  if (r == 0.0) return q;
  const int up = (r > 0.0) == (b > 0.0);
  long long bits;
  memcpy(&bits, &q, sizeof bits);
  bits += ((q > 0.0) == up) ? 1 : -1;
  double q1;
  memcpy(&q1, &bits, sizeof q1);
  const double r1 = fma(-q1, b, a);
  return (fabs(r1) < fabs(r)) ? q1 : q;
}

static int g_term_out = 1;
static int g_stall_limit = 0; /* > 0: overrides 64 + (m+n)/8 (tests drive the anti-stalling rules with it) */
void orc_set_stall_limit(int limit) { g_stall_limit = limit; }

/* ------------------------------------------------------------------ helpers */
static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz);
  if (!p) {
    fprintf(stderr, "orc: out of memory\n");
    abort();
  }
  return p;
}
static void *xrealloc(void *p, size_t sz) {
  p = realloc(p, sz ? sz : 1);
  if (!p) {
    fprintf(stderr, "orc: out of memory\n");
    abort();
  }
  return p;
}
static void fault(const char *msg) {
  /* GLPK aborts on invalid arguments [GLPK-recalled]; so does the oracle. */
  fprintf(stderr, "orc: %s\n", msg);
  abort();
}

static void norm_bounds(int type, double lb, double ub, double *olb, double *oub) {
  switch (type) {
    case ORC_FR: *olb = -INF; *oub = INF; break;
    case ORC_LO: *olb = lb; *oub = INF; break;
    case ORC_UP: *olb = -INF; *oub = ub; break;
    case ORC_DB: *olb = lb; *oub = ub; break;
    case ORC_FX: *olb = lb; *oub = lb; break;
    default: fault("invalid bound type");
  }
}

static int std_flag(int type) {
  /* glp_std_basis convention [GLPK-recalled]: NL if a lower bound exists, else NU,
     free -> NF, fixed -> NS */
  switch (type) {
    case ORC_FR: return ORC_NF;
    case ORC_LO: return ORC_NL;
    case ORC_UP: return ORC_NU;
    case ORC_DB: return ORC_NL;
    default: return ORC_NS;
  }
}

static double nb_value(int flag, double lb, double ub) {
  switch (flag) {
    case ORC_NL: return lb;
    case ORC_NU: return ub;
    case ORC_NS: return lb;
    default: return 0.0;
  }
}

/* ---------------------------------------------------------------- lifecycle */
orc_prob *orc_create_prob(void) {
  orc_prob *P = (orc_prob *)xcalloc(1, sizeof(orc_prob));
  P->dir = ORC_MIN; /* GLPK default [GLPK-recalled] */
  P->c = (double *)xcalloc(1, sizeof(double));
  P->status = ORC_UNDEF;
  return P;
}

static void free_contents(orc_prob *P) {
  for (int i = 1; i <= P->m; i++) free(P->A[i]);
  free(P->A);
  free(P->c);
  free(P->kind);
  if (P->cname) {
    for (int j = 1; j <= P->n; j++) free(P->cname[j]);
    free(P->cname);
  }
  free(P->rtype); free(P->rlb); free(P->rub);
  free(P->ctype); free(P->clb); free(P->cub);
  free(P->T);
  free(P->bvar); free(P->blb); free(P->bub);
  free(P->nvar); free(P->nflag); free(P->nlb); free(P->nub);
}

void orc_erase_prob(orc_prob *P) {
  free_contents(P);
  memset(P, 0, sizeof(*P));
  P->dir = ORC_MIN;
  P->c = (double *)xcalloc(1, sizeof(double));
  P->status = ORC_UNDEF;
}

void orc_delete_prob(orc_prob *P) {
  if (!P) return;
  free_contents(P);
  free(P);
}

static void ensure_rows(orc_prob *P, int m_new) {
  if (m_new <= P->m_cap) return;
  int cap = P->m_cap ? P->m_cap : 8;
  while (cap < m_new) cap *= 2;
  P->A = (double **)xrealloc(P->A, (size_t)(cap + 1) * sizeof(double *));
  P->rtype = (int *)xrealloc(P->rtype, (size_t)(cap + 1) * sizeof(int));
  P->rlb = (double *)xrealloc(P->rlb, (size_t)(cap + 1) * sizeof(double));
  P->rub = (double *)xrealloc(P->rub, (size_t)(cap + 1) * sizeof(double));
  P->bvar = (int *)xrealloc(P->bvar, (size_t)(cap + 1) * sizeof(int));
  P->blb = (double *)xrealloc(P->blb, (size_t)(cap + 1) * sizeof(double));
  P->bub = (double *)xrealloc(P->bub, (size_t)(cap + 1) * sizeof(double));
  if (P->T) P->T = (double *)xrealloc(P->T, (size_t)(cap + 1) * P->ld * sizeof(double));
  P->m_cap = cap;
}

static void ensure_cols(orc_prob *P, int n_new) {
  if (n_new <= P->n_cap) return;
  int cap = P->n_cap ? P->n_cap : 8;
  while (cap < n_new) cap *= 2;
  for (int i = 1; i <= P->m; i++) {
    P->A[i] = (double *)xrealloc(P->A[i], (size_t)(cap + 1) * sizeof(double));
    for (int j = P->n_cap + 1; j <= cap; j++) P->A[i][j] = 0.0;
  }
  P->c = (double *)xrealloc(P->c, (size_t)(cap + 1) * sizeof(double));
  P->kind = (int *)xrealloc(P->kind, (size_t)(cap + 1) * sizeof(int));
  P->ctype = (int *)xrealloc(P->ctype, (size_t)(cap + 1) * sizeof(int));
  P->clb = (double *)xrealloc(P->clb, (size_t)(cap + 1) * sizeof(double));
  P->cub = (double *)xrealloc(P->cub, (size_t)(cap + 1) * sizeof(double));
  P->nvar = (int *)xrealloc(P->nvar, (size_t)(cap + 1) * sizeof(int));
  P->nflag = (int *)xrealloc(P->nflag, (size_t)(cap + 1) * sizeof(int));
  P->nlb = (double *)xrealloc(P->nlb, (size_t)(cap + 1) * sizeof(double));
  P->nub = (double *)xrealloc(P->nub, (size_t)(cap + 1) * sizeof(double));
  if (P->cname) {
    P->cname = (char **)xrealloc(P->cname, (size_t)(cap + 1) * sizeof(char *));
    for (int j = P->n_cap + 1; j <= cap; j++) P->cname[j] = NULL;
  }
  P->n_cap = cap;
}

void orc_copy_prob(orc_prob *dst, const orc_prob *src, int names) {
  /* glp_copy_prob (bs.cpp:116, util.cpp:34): deep copy incl. basis and last solution
     [GLPK-recalled]; names iff `names` is ON. */
  if (dst == src) fault("copy_prob: dst == src");
  orc_erase_prob(dst);
  ensure_rows(dst, src->m);
  ensure_cols(dst, src->n);
  dst->m = src->m;
  dst->n = src->n;
  dst->dir = src->dir;
  for (int i = 1; i <= src->m; i++) {
    dst->A[i] = (double *)xcalloc((size_t)dst->n_cap + 1, sizeof(double));
    memcpy(dst->A[i], src->A[i], (size_t)(src->n + 1) * sizeof(double));
    dst->rtype[i] = src->rtype[i];
    dst->rlb[i] = src->rlb[i];
    dst->rub[i] = src->rub[i];
  }
  memcpy(dst->c, src->c, (size_t)(src->n + 1) * sizeof(double));
  for (int j = 1; j <= src->n; j++) {
    dst->kind[j] = src->kind[j];
    dst->ctype[j] = src->ctype[j];
    dst->clb[j] = src->clb[j];
    dst->cub[j] = src->cub[j];
  }
  if (names && src->cname) {
    dst->cname = (char **)xcalloc((size_t)dst->n_cap + 1, sizeof(char *));
    for (int j = 1; j <= src->n; j++)
      if (src->cname[j]) dst->cname[j] = strdup(src->cname[j]);
  }
  dst->valid = src->valid;
  dst->status = src->status;
  dst->it_cnt = src->it_cnt;
  dst->piv_since_check = src->piv_since_check;
  dst->refresh_cnt = src->refresh_cnt;
  dst->bland_cnt = src->bland_cnt;
  dst->pert_cnt = src->pert_cnt;
  if (src->valid) {
    dst->ld = src->ld;
    dst->T = (double *)xcalloc((size_t)(dst->m_cap + 1) * dst->ld, sizeof(double));
    memcpy(dst->T, src->T, (size_t)(src->m + 1) * src->ld * sizeof(double));
    for (int i = 1; i <= src->m; i++) {
      dst->bvar[i] = src->bvar[i];
      dst->blb[i] = src->blb[i];
      dst->bub[i] = src->bub[i];
    }
    for (int j = 1; j <= src->n; j++) {
      dst->nvar[j] = src->nvar[j];
      dst->nflag[j] = src->nflag[j];
      dst->nlb[j] = src->nlb[j];
      dst->nub[j] = src->nub[j];
    }
  }
}

/* -------------------------------------------------------- tableau utilities */
#define TT(P, i, j) ((P)->T[(size_t)(i) * (P)->ld + (j)])

/* out[j] (j=0..n) = base[j] + sum over 64-row chunks (in order) of the fma-chain
   sum_{i in chunk, w[i] != 0} w[i]*T[i][j].  Same order in the HIP kernel k_rowcomb. */
static void rowcomb(const orc_prob *P, const double *w, const double *base, double *out) {
  int m = P->m, n = P->n;
  for (int j = 0; j <= n; j++) out[j] = base ? base[j] : 0.0;
  for (int c0 = 1; c0 <= m; c0 += ROWCOMB_CHUNK) {
    int c1 = c0 + ROWCOMB_CHUNK - 1;
    if (c1 > m) c1 = m;
    for (int j = 0; j <= n; j++) {
      double acc = 0.0;
      for (int i = c0; i <= c1; i++)
        if (w[i] != 0.0) acc = fma(w[i], TT(P, i, j), acc);
      out[j] = out[j] + acc;
    }
  }
}

/* position of variable k: +i if basic in row i, -j if non-basic in column j */
static int var_pos(const orc_prob *P, int k) {
  for (int i = 1; i <= P->m; i++)
    if (P->bvar[i] == k) return i;
  for (int j = 1; j <= P->n; j++)
    if (P->nvar[j] == k) return -j;
  fault("var_pos: variable not found");
  return 0;
}

static void build_slack_tableau_flags(orc_prob *P, const int *sflag);
static void build_slack_tableau(orc_prob *P) { build_slack_tableau_flags(P, NULL); }

/* sflag (nullable): non-basic status per structural column instead of the standard one (tableau refresh) */
static void build_slack_tableau_flags(orc_prob *P, const int *sflag) {
  int m = P->m, n = P->n;
  P->ld = ((n + 1 + 7) / 8) * 8;
  free(P->T);
  P->T = (double *)xcalloc((size_t)(P->m_cap + 1) * P->ld, sizeof(double));
  for (int j = 1; j <= n; j++) {
    P->nvar[j] = m + j;
    P->nflag[j] = sflag ? sflag[j] : std_flag(P->ctype[j]);
    P->nlb[j] = P->clb[j];
    P->nub[j] = P->cub[j];
  }
  for (int i = 1; i <= m; i++) {
    P->bvar[i] = i;
    P->blb[i] = P->rlb[i];
    P->bub[i] = P->rub[i];
  }
  double z = P->c[0];
  for (int j = 1; j <= n; j++) {
    double x = nb_value(P->nflag[j], P->nlb[j], P->nub[j]);
    TT(P, 0, j) = P->c[j];
    if (x != 0.0) z = fma(P->c[j], x, z);
  }
  TT(P, 0, 0) = z;
  for (int i = 1; i <= m; i++) {
    double acc = 0.0;
    for (int j = 1; j <= n; j++) {
      double a = P->A[i][j];
      double x = nb_value(P->nflag[j], P->nlb[j], P->nub[j]);
      TT(P, i, j) = a;
      if (x != 0.0) acc = fma(a, x, acc);
    }
    TT(P, i, 0) = acc;
  }
  P->valid = 1;
  P->status = ORC_UNDEF;
}

/* a non-basic variable in column jj changes value by delta: beta += T[:,jj]*delta */
static void shift_nonbasic(orc_prob *P, int jj, double delta) {
  for (int i = 0; i <= P->m; i++) TT(P, i, 0) = fma(TT(P, i, jj), delta, TT(P, i, 0));
}

static void apply_bounds_to_engine(orc_prob *P, int k, int type, double lb, double ub) {
  if (!P->valid) return;
  int pos = var_pos(P, k);
  if (pos > 0) {
    P->blb[pos] = lb;
    P->bub[pos] = ub;
  } else {
    int jj = -pos;
    double xo = nb_value(P->nflag[jj], P->nlb[jj], P->nub[jj]);
    int flag;
    switch (type) {
      case ORC_FR: flag = ORC_NF; break;
      case ORC_LO: flag = ORC_NL; break;
      case ORC_UP: flag = ORC_NU; break;
      case ORC_DB: flag = (P->nflag[jj] == ORC_NU) ? ORC_NU : ORC_NL; break;
      default: flag = ORC_NS; break;
    }
    P->nlb[jj] = lb;
    P->nub[jj] = ub;
    P->nflag[jj] = flag;
    double xn = nb_value(flag, lb, ub);
    if (xn != xo) shift_nonbasic(P, jj, xn - xo);
  }
  P->status = ORC_UNDEF;
}

/* ----------------------------------------------------------- build / modify */
void orc_set_obj_dir(orc_prob *P, int dir) {
  if (dir != ORC_MIN && dir != ORC_MAX) fault("set_obj_dir: invalid direction");
  P->dir = dir;
  P->status = ORC_UNDEF;
}

int orc_add_rows(orc_prob *P, int nrs) {
  if (nrs < 1) fault("add_rows: invalid count");
  int first = P->m + 1;
  ensure_rows(P, P->m + nrs);
  for (int i = first; i < first + nrs; i++) {
    P->A[i] = (double *)xcalloc((size_t)P->n_cap + 1, sizeof(double));
    P->rtype[i] = ORC_FR;
    P->rlb[i] = -INF;
    P->rub[i] = INF;
    if (P->valid) {
      /* new auxiliary variable enters the basis in its own new row [GLPK-recalled:
         glp_add_rows creates rows with status GLP_BS]; an empty row is identically 0 */
      P->bvar[i] = i;
      P->blb[i] = -INF;
      P->bub[i] = INF;
      for (int j = 0; j <= P->n; j++) TT(P, i, j) = 0.0;
    }
  }
  P->m += nrs;
  if (P->valid) {
    /* structural variable numbers shift by nrs */
    for (int i = 1; i < first; i++)
      if (P->bvar[i] >= first) P->bvar[i] += nrs;
    for (int j = 1; j <= P->n; j++)
      if (P->nvar[j] >= first) P->nvar[j] += nrs;
  }
  P->status = ORC_UNDEF;
  return first;
}

int orc_add_cols(orc_prob *P, int ncs) {
  if (ncs < 1) fault("add_cols: invalid count");
  int first = P->n + 1;
  ensure_cols(P, P->n + ncs);
  for (int j = first; j < first + ncs; j++) {
    P->c[j] = 0.0;
    P->kind[j] = ORC_CV;
    P->ctype[j] = ORC_FX; /* GLPK default column: fixed at 0 [GLPK-recalled] */
    P->clb[j] = 0.0;
    P->cub[j] = 0.0;
    for (int i = 1; i <= P->m; i++) P->A[i][j] = 0.0;
  }
  P->n += ncs;
  P->valid = 0; /* engine state is rebuilt from the slack basis */
  P->status = ORC_UNDEF;
  return first;
}

void orc_set_row_bnds(orc_prob *P, int i, int type, double lb, double ub) {
  if (i < 1 || i > P->m) fault("set_row_bnds: row out of range");
  P->rtype[i] = type;
  norm_bounds(type, lb, ub, &P->rlb[i], &P->rub[i]);
  apply_bounds_to_engine(P, i, type, P->rlb[i], P->rub[i]);
}

void orc_set_col_bnds(orc_prob *P, int j, int type, double lb, double ub) {
  if (j < 1 || j > P->n) fault("set_col_bnds: column out of range");
  P->ctype[j] = type;
  norm_bounds(type, lb, ub, &P->clb[j], &P->cub[j]);
  apply_bounds_to_engine(P, P->m + j, type, P->clb[j], P->cub[j]);
}

static void recompute_cost_row(orc_prob *P) {
  /* d = c_N + c_B^T T ; z = c0 + c_B^T beta + c_N^T x_N */
  int m = P->m, n = P->n;
  double *w = (double *)xcalloc((size_t)m + 1, sizeof(double));
  double *base = (double *)xcalloc((size_t)n + 1, sizeof(double));
  double *out = (double *)xcalloc((size_t)n + 1, sizeof(double));
  for (int i = 1; i <= m; i++) w[i] = (P->bvar[i] > m) ? P->c[P->bvar[i] - m] : 0.0;
  double z = P->c[0];
  for (int j = 1; j <= n; j++) {
    double cj = (P->nvar[j] > m) ? P->c[P->nvar[j] - m] : 0.0;
    double x = nb_value(P->nflag[j], P->nlb[j], P->nub[j]);
    base[j] = cj;
    if (x != 0.0 && cj != 0.0) z = fma(cj, x, z);
  }
  base[0] = z;
  rowcomb(P, w, base, out);
  for (int j = 0; j <= n; j++) TT(P, 0, j) = out[j];
  free(w); free(base); free(out);
}

void orc_set_obj_coef(orc_prob *P, int j, double coef) {
  if (j < 0 || j > P->n) fault("set_obj_coef: column out of range");
  P->c[j] = coef;
  if (P->valid) recompute_cost_row(P);
  P->status = ORC_UNDEF;
}

void orc_set_mat_row(orc_prob *P, int i, int len, const int *ind, const double *val) {
  /* cut.cpp:40: 1-based ind/val, element 0 ignored */
  if (i < 1 || i > P->m) fault("set_mat_row: row out of range");
  if (len < 0 || len > P->n) fault("set_mat_row: invalid length");
  for (int j = 1; j <= P->n; j++) P->A[i][j] = 0.0;
  for (int k = 1; k <= len; k++) {
    if (ind[k] < 1 || ind[k] > P->n) fault("set_mat_row: column index out of range");
    P->A[i][ind[k]] = val[k];
  }
  if (P->valid) {
    int pos = var_pos(P, i);
    if (pos <= 0) {
      P->valid = 0; /* row of a non-basic auxiliary changed: rebuild from the slack basis */
    } else {
      /* x_i = sum_j v_j x_(m+j): substitute the basic structurals by their tableau rows */
      int m = P->m, n = P->n;
      double *w = (double *)xcalloc((size_t)m + 1, sizeof(double));
      double *base = (double *)xcalloc((size_t)n + 1, sizeof(double));
      double *out = (double *)xcalloc((size_t)n + 1, sizeof(double));
      for (int r = 1; r <= m; r++)
        if (r != pos && P->bvar[r] > m) w[r] = P->A[i][P->bvar[r] - m];
      double b0 = 0.0;
      for (int jj = 1; jj <= n; jj++) {
        if (P->nvar[jj] > m) {
          double v = P->A[i][P->nvar[jj] - m];
          double x = nb_value(P->nflag[jj], P->nlb[jj], P->nub[jj]);
          base[jj] = v;
          if (x != 0.0 && v != 0.0) b0 = fma(v, x, b0);
        }
      }
      base[0] = b0;
      rowcomb(P, w, base, out);
      for (int jj = 0; jj <= n; jj++) TT(P, pos, jj) = out[jj];
      free(w); free(base); free(out);
    }
  }
  P->status = ORC_UNDEF;
}

void orc_set_col_kind(orc_prob *P, int j, int kind) {
  if (j < 1 || j > P->n) fault("set_col_kind: column out of range");
  if (kind == ORC_BV) {
    /* GLPK: BV = IV with bounds [0,1] [GLPK-recalled] */
    P->kind[j] = ORC_IV;
    orc_set_col_bnds(P, j, ORC_DB, 0.0, 1.0);
  } else {
    P->kind[j] = kind;
  }
}

void orc_set_col_name(orc_prob *P, int j, const char *name) {
  if (j < 1 || j > P->n) fault("set_col_name: column out of range");
  if (!P->cname) P->cname = (char **)xcalloc((size_t)P->n_cap + 1, sizeof(char *));
  free(P->cname[j]);
  P->cname[j] = name ? strdup(name) : NULL;
}

const char *orc_get_col_name(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_name: column out of range");
  return P->cname ? P->cname[j] : NULL;
}

int orc_load_dense(orc_prob *P, int m, int n, const double *A, const double *b, const double *c) {
  /* fast-path loader: max c'x, Ax <= b, x >= 0 (SURVEY.md section 8(d) generator) */
  orc_erase_prob(P);
  orc_set_obj_dir(P, ORC_MAX);
  orc_add_cols(P, n);
  orc_add_rows(P, m);
  for (int j = 1; j <= n; j++) {
    P->c[j] = c[j - 1];
    P->ctype[j] = ORC_LO;
    P->clb[j] = 0.0;
    P->cub[j] = INF;
  }
  for (int i = 1; i <= m; i++) {
    memcpy(&P->A[i][1], &A[(size_t)(i - 1) * n], (size_t)n * sizeof(double));
    P->rtype[i] = ORC_UP;
    P->rlb[i] = -INF;
    P->rub[i] = b[i - 1];
  }
  return 0;
}

/* ------------------------------------------------------------ simplex steps */
typedef struct {
  double tol_bnd, tol_dj, tol_piv;
  int budget; /* remaining pivots, <0 = unlimited */
  /* anti-cycling: `stall` counts consecutive degenerate pivots (step length at most DEGEN_TOL).  From
     `stall_limit` on, every choice is made by Bland's smallest-subscript rule (entering and leaving
     variable by lowest variable number) until a pivot moves again; a cycle consists of degenerate
     pivots only, so this ends it.  The HIP engine keeps the same two counters in its control block. */
  int stall, stall_limit;
  /* first line of defence against stalling, primal phase 2 only: when `stall` first reaches the limit
     the bounds of all basic variables are pushed outwards by tiny, distinct amounts (PERT_EPS), which
     breaks the ties of a degenerate vertex; the true bounds come back when phase 2 ends (the dual
     simplex then removes what infeasibility is left).  Used once per solve; Bland's rule covers a
     second stall, the dual simplex and phase 1. */
  int perturbed, pert_used;
  /* primal devex reference weights by non-basic position (phase 2 only; NULL elsewhere): entering column
     = argmax d_j^2 / pw[j]; after a pivot (p,q): pw[j] = max(pw[j], (a_pj/a_pq)^2 pw[q]) for j != q,
     pw[q] = max(pw[q] / a_pq^2, 1).  Reset on entering the phase.  Halves the pivots of large LPs
     (1024x2048 seed 12345: 1439 -> 754) */
  double *pw;
  /* primal phase 1 only: the infeasibility-sum cost row sum_i g[i]*T[i][:] is carried through the pivots
     like one more tableau row (xrow) instead of being recomputed; xg = the signs g it is built from */
  double *xrow;
  int *xg;
} ctl_t;

#define PERT_EPS 1e-6
#define DEGEN_TOL 1e-9 /* a step (primal) or dual ratio no longer than this counts as degenerate */

static int bland_on(const ctl_t *ctl) { return ctl->stall >= ctl->stall_limit; }

/* Dantzig pricing on row `cost` (length n+1, entries 1..n), maximisation sense already
   folded in through sgn.  Returns column q (0 = none), *sdir = +1 (increase) / -1. */
static int price(const orc_prob *P, const double *cost, double sgn, double tol, int *sdir, int bland, const double *pw) {
  int q = 0;
  double best = 0.0;
  for (int j = 1; j <= P->n; j++) {
    int f = P->nflag[j];
    if (f == ORC_NS) continue;
    double dj = sgn * cost[j];
    int up = (f == ORC_NL || f == ORC_NF) && dj > tol;
    int dn = (f == ORC_NU || f == ORC_NF) && dj < -tol;
    if (!up && !dn) continue;
    /* Bland: lowest variable number wins; devex: d^2 / weight; else Dantzig */
    double sc = bland ? -(double)P->nvar[j] : (pw ? xdiv(dj * dj, pw[j]) : fabs(dj));
    if (q == 0 || sc > best) { /* strict > keeps the lowest j on ties */
      best = sc;
      q = j;
      *sdir = up ? +1 : -1;
    }
  }
  return q;
}

/* lexicographic candidate compare: smaller t, then larger |a|, then smaller index */
static int better(double t, double mag, int idx, double bt, double bmag, int bidx) {
  if (bidx == 0) return 1;
  if (t < bt) return 1;
  if (t > bt) return 0;
  if (mag > bmag) return 1;
  if (mag < bmag) return 0;
  return idx < bidx;
}

/* leaving row of the dual simplex (0 = primal feasible); *to_upper = 1 if above ub.  Score of an
   infeasible row: viol^2 / w[i] with the dual devex reference weights w (NULL = all ones), lowest row on
   ties; under Bland's rule the lowest variable number. */
static int select_infeasible_row(const orc_prob *P, double tol_bnd, int *to_upper, int bland, const double *w) {
  int p = 0;
  double best = 0.0;
  for (int i = 1; i <= P->m; i++) {
    double beta = TT(P, i, 0);
    double lb = P->blb[i], ub = P->bub[i];
    double viol = 0.0;
    int up = 0;
    if (lb > -INF && beta < lb - tol_bnd * (1.0 + fabs(lb))) viol = lb - beta;
    if (ub < INF && beta > ub + tol_bnd * (1.0 + fabs(ub))) {
      viol = beta - ub;
      up = 1;
    }
    double sc = bland ? -(double)P->bvar[i] : xdiv(viol * viol, w ? w[i] : 1.0);
    if (viol > 0.0 && (p == 0 || sc > best)) {
      best = sc;
      p = i;
      *to_upper = up;
    }
  }
  return p;
}

/* Gauss-Jordan pivot on (p,q).  bound = value the leaving variable lands on,
   leave_flag = its non-basic status afterwards. */
static void pivot(orc_prob *P, int p, int q, double bound, int leave_flag, double *xrow) {
  int m = P->m, n = P->n, ld = P->ld;
  double *rowp = &TT(P, p, 0);
  double piv = rowp[q];
  double xq = nb_value(P->nflag[q], P->nlb[q], P->nub[q]);
  double *s = (double *)xcalloc((size_t)n + 1, sizeof(double));
  s[0] = xdiv(rowp[0] - bound, piv);
  for (int j = 1; j <= n; j++) s[j] = xdiv(rowp[j], piv);
#pragma omp parallel for schedule(static) if ((long)m * n >= 262144)
  for (int i = 0; i <= m; i++) {
    if (i == p) continue;
    double *row = &P->T[(size_t)i * ld];
    double ci = row[q];
    double nci = -ci;
    for (int j = 0; j <= n; j++) row[j] = fma(nci, s[j], row[j]);
    row[q] = xdiv(ci, piv);
  }
  if (xrow) { /* one more row i != p (the device keeps it in tableau row m+1) */
    double ci = xrow[q];
    double nci = -ci;
    for (int j = 0; j <= n; j++) xrow[j] = fma(nci, s[j], xrow[j]);
    xrow[q] = xdiv(ci, piv);
  }
  for (int j = 1; j <= n; j++) rowp[j] = -s[j];
  rowp[q] = xdiv(1.0, piv);
  rowp[0] = xq - s[0];
  free(s);
  /* swap basis bookkeeping */
  int kv = P->bvar[p];
  double klb = P->blb[p], kub = P->bub[p];
  P->bvar[p] = P->nvar[q];
  P->blb[p] = P->nlb[q];
  P->bub[p] = P->nub[q];
  P->nvar[q] = kv;
  P->nlb[q] = klb;
  P->nub[q] = kub;
  P->nflag[q] = leave_flag;
  P->it_cnt++;
}

static int leave_flag_for(double lb, double ub, int to_upper) {
  if (lb == ub) return ORC_NS;
  return to_upper ? ORC_NU : ORC_NL;
}

enum { R_OPT = 1, R_UNBND, R_NOFEAS, R_ITLIM, R_PFEAS, R_FAIL };

/* Primal ratio test for entering column q moving in direction sdir (nothing is changed).
   phase1: g[i] != 0 marks infeasible basics (+1 below lb, -1 above ub).
   Returns RT_PIVOT (*p, *p_up, *bt = leaving row, the bound it lands on, the step), RT_FLIP (the entering variable
   reaches its own other bound first; *bt = 0) or RT_NONE (no blocking row). */
enum { RT_PIVOT = 0, RT_FLIP = 1, RT_NONE = 2 };
static int primal_ratio(const orc_prob *P, const ctl_t *ctl, int q, int sdir, const int *g, int *pp, int *pp_up, double *pbt) {
  int m = P->m;
  int p = 0, p_up = 0;
  const int bland = bland_on(ctl);
  double bt = 0.0, bmag = 0.0;
  for (int i = 1; i <= m; i++) {
    double a = TT(P, i, q);
    double aa = (sdir > 0) ? a : -a;
    double beta = TT(P, i, 0);
    double t;
    int up;
    if (aa > ctl->tol_piv) { /* basic variable increases */
      if (g && g[i] < 0) continue;               /* above ub, moving further away */
      if (g && g[i] > 0) { t = xdiv(P->blb[i] - beta, aa); up = 0; } /* reaches lb from below */
      else {
        if (!(P->bub[i] < INF)) continue;
        t = xdiv(P->bub[i] - beta, aa);
        up = 1;
      }
    } else if (aa < -ctl->tol_piv) { /* decreases */
      if (g && g[i] > 0) continue;
      if (g && g[i] < 0) { t = xdiv(beta - P->bub[i], -aa); up = 1; }
      else {
        if (!(P->blb[i] > -INF)) continue;
        t = xdiv(beta - P->blb[i], -aa);
        up = 0;
      }
    } else
      continue;
    if (t < 0.0) t = 0.0;
    double mag = bland ? -(double)P->bvar[i] : fabs(a); /* tie-break among equal steps */
    if (better(t, mag, i, bt, bmag, p)) {
      bt = t;
      bmag = mag;
      p = i;
      p_up = up;
    }
  }
  *pp = p;
  *pp_up = p_up;
  *pbt = bt;
  /* bound flip of the entering variable itself */
  if (P->nlb[q] > -INF && P->nub[q] < INF && P->nflag[q] != ORC_NF) {
    double tf = P->nub[q] - P->nlb[q];
    if (p == 0 || tf <= bt) return RT_FLIP;
  }
  return p == 0 ? RT_NONE : RT_PIVOT;
}

/* Carries out what primal_ratio found: the bound flip, or the pivot (p, q) with the devex weight update. */
static void primal_apply(orc_prob *P, ctl_t *ctl, int kind, int q, int sdir, int p, int p_up, double bt) {
  const int bland = bland_on(ctl);
  if (kind == RT_FLIP) {
    double tf = P->nub[q] - P->nlb[q];
    double delta = (sdir > 0) ? tf : -tf;
    shift_nonbasic(P, q, delta);
    P->nflag[q] = (sdir > 0) ? ORC_NU : ORC_NL;
    ctl->stall = 0;
    return;
  }
  double bound = p_up ? P->bub[p] : P->blb[p];
  if (ctl->pw) {
    double *w = ctl->pw;
    const double apq = TT(P, p, q), wq = w[q];
    for (int j = 1; j <= P->n; j++) {
      if (j == q) continue;
      double r = xdiv(TT(P, p, j), apq);
      double c = r * r * wq;
      if (c > w[j]) w[j] = c;
    }
    double c = xdiv(wq, apq * apq);
    w[q] = c > 1.0 ? c : 1.0;
  }
  pivot(P, p, q, bound, leave_flag_for(P->blb[p], P->bub[p], p_up), ctl->xrow);
  if (ctl->xrow) {
    /* xrow is now sum_{i != p} g[i]*T'[i][:] + g[p]*e_q: the leaving variable sits on the bound it was
       violating (or was feasible, g[p] = 0), so its term goes; row p holds the entering variable, no term yet */
    ctl->xrow[q] = ctl->xrow[q] - (double)ctl->xg[p];
    ctl->xg[p] = 0;
  }
  if (ctl->budget > 0) ctl->budget--;
  if (bland) P->bland_cnt++;
  ctl->stall = (bt <= DEGEN_TOL) ? ctl->stall + 1 : 0;
}

/* One primal iteration for entering column q moving in direction sdir.
   Returns 0 = pivot/flip done, R_UNBND = no blocking row. */
static int primal_step(orc_prob *P, ctl_t *ctl, int q, int sdir, const int *g) {
  int p, p_up;
  double bt;
  const int kind = primal_ratio(P, ctl, q, sdir, g, &p, &p_up, &bt);
  if (kind == RT_NONE) return R_UNBND;
  primal_apply(P, ctl, kind, q, sdir, p, p_up, bt);
  return 0;
}

/* splitmix64 of the variable number -> [0,1): the same amount for a variable wherever it sits */
static double pert_unit(int var) {
  unsigned long long z = (unsigned long long)var * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

static void var_true_bounds(const orc_prob *P, int k, double *lb, double *ub) {
  if (k <= P->m) { *lb = P->rlb[k]; *ub = P->rub[k]; }
  else { *lb = P->clb[k - P->m]; *ub = P->cub[k - P->m]; }
}

static void perturb_basis(orc_prob *P, ctl_t *ctl) {
  for (int i = 1; i <= P->m; i++) {
    double u = 1.0 + pert_unit(P->bvar[i]);
    if (P->blb[i] > -INF) {
      double d = PERT_EPS * (1.0 + fabs(P->blb[i]));
      d = d * u;
      P->blb[i] = P->blb[i] - d;
    }
    if (P->bub[i] < INF) {
      double d = PERT_EPS * (1.0 + fabs(P->bub[i]));
      d = d * u;
      P->bub[i] = P->bub[i] + d;
    }
  }
  ctl->perturbed = 1;
  ctl->pert_used = 1;
  ctl->stall = 0;
  P->pert_cnt++;
}

/* true bounds back on every position; non-basic variables parked on a perturbed bound move to the true
   one (columns in ascending order, one shift_nonbasic each) */
static void restore_bounds(orc_prob *P, ctl_t *ctl) {
  for (int i = 1; i <= P->m; i++) var_true_bounds(P, P->bvar[i], &P->blb[i], &P->bub[i]);
  for (int j = 1; j <= P->n; j++) {
    double lb, ub;
    var_true_bounds(P, P->nvar[j], &lb, &ub);
    int f = P->nflag[j];
    double xold = nb_value(f, P->nlb[j], P->nub[j]);
    if (lb == ub) f = ORC_NS;
    P->nlb[j] = lb;
    P->nub[j] = ub;
    P->nflag[j] = f;
    double xnew = nb_value(f, lb, ub);
    if (xnew != xold) shift_nonbasic(P, j, xnew - xold);
  }
  ctl->perturbed = 0;
}

/* Score of column j as an entering candidate (devex: d^2 / weight); 0 = not eligible. */
static int price_col(const orc_prob *P, const double *cost, double sgn, double tol, const double *pw, int j, double *sc, int *sdir) {
  int f = P->nflag[j];
  if (f == ORC_NS) return 0;
  double dj = sgn * cost[j];
  int up = (f == ORC_NL || f == ORC_NF) && dj > tol;
  int dn = (f == ORC_NU || f == ORC_NF) && dj < -tol;
  if (!up && !dn) return 0;
  *sc = xdiv(dj * dj, pw[j]);
  *sdir = up ? +1 : -1;
  return 1;
}

static int g_mp_cand = 0;
long long g_mp_major = 0, g_mp_minor = 0;
void orc_set_mp(int k) { g_mp_cand = k; }
void orc_mp_stats(long long *out) { out[0] = g_mp_major; out[1] = g_mp_minor; g_mp_major = g_mp_minor = 0; }
#define MP_MAX 64

static int primal_phase2(orc_prob *P, ctl_t *ctl) {
  double sgn = (P->dir == ORC_MAX) ? 1.0 : -1.0;
  double *w = (double *)xcalloc((size_t)P->n + 1, sizeof(double));
  for (int j = 0; j <= P->n; j++) w[j] = 1.0;
  ctl->pw = w;
  for (;;) {
    int sdir = 0;
    if (bland_on(ctl) && !ctl->pert_used) perturb_basis(P, ctl);
    int r = 0;
    const int K = (bland_on(ctl) || g_mp_cand <= 1) ? 1 : (g_mp_cand > MP_MAX ? MP_MAX : g_mp_cand);
    if (K == 1) {
      int q = price(P, &TT(P, 0, 0), sgn, ctl->tol_dj, &sdir, bland_on(ctl), w);
      if (q == 0) r = R_OPT;
      else if (ctl->budget == 0) r = R_ITLIM;
      else r = primal_step(P, ctl, q, sdir, NULL);
    } else {
      /* multiple pricing: the K best columns of a full pricing pass are the candidates of one major iteration; the
         minor iterations that follow choose among them only (current reduced costs and weights) */
      int cand[MP_MAX], nc = 0;
      double csc[MP_MAX];
      const double *cost = &TT(P, 0, 0);
      for (int j = 1; j <= P->n; j++) {
        double sc;
        int sd;
        if (!price_col(P, cost, sgn, ctl->tol_dj, w, j, &sc, &sd)) continue;
        /* sorted by (score desc, j asc); j ascends, so an equal score goes behind */
        int pos = nc;
        while (pos > 0 && sc > csc[pos - 1]) pos--;
        if (pos >= K) continue;
        int last = (nc < K) ? nc : K - 1;
        for (int t = last; t > pos; t--) { cand[t] = cand[t - 1]; csc[t] = csc[t - 1]; }
        cand[pos] = j;
        csc[pos] = sc;
        if (nc < K) nc++;
      }
      if (nc == 0) r = R_OPT;
      else if (ctl->budget == 0) r = R_ITLIM;
      else {
        int steps = 0;
        g_mp_major++;
        for (;;) {
          int q = 0, qs = 0, qc = -1;
          double best = 0.0;
          for (int t = 0; t < nc; t++) {
            double sc;
            int sd;
            if (cand[t] == 0 || !price_col(P, cost, sgn, ctl->tol_dj, w, cand[t], &sc, &sd)) continue;
            if (q == 0 || sc > best || (sc == best && cand[t] < q)) { best = sc; q = cand[t]; qs = sd; qc = t; }
          }
          if (q == 0) break;
          if (steps > 0 && (ctl->budget == 0 || bland_on(ctl))) break;
          int p, p_up;
          double bt;
          const int kind = primal_ratio(P, ctl, q, qs, NULL, &p, &p_up, &bt);
          if (kind == RT_NONE) { if (steps == 0) r = R_UNBND; break; }
          if (kind == RT_FLIP && steps > 0) break;
          primal_apply(P, ctl, kind, q, qs, p, p_up, bt);
          cand[qc] = 0;
          steps++;
          g_mp_minor++;
          if (kind == RT_FLIP || steps == K) break;
        }
      }
    }
    if (r) {
      if (ctl->perturbed) restore_bounds(P, ctl);
      ctl->pw = NULL;
      free(w);
      return r;
    }
  }
}

/* Primal phase 1: minimise the sum of infeasibilities.  The cost row sum_i g[i]*T[i][:] (g = +1 below the
   lower bound, -1 above the upper one) is kept up to date instead of recomputed: rows whose sign changed
   since the last iteration are added / removed in ascending row order, the pivot carries the row along. */
static int primal_phase1(orc_prob *P, ctl_t *ctl) {
  int m = P->m, n = P->n;
  int *g = (int *)xcalloc((size_t)m + 2, sizeof(int));
  double *cost = (double *)xcalloc((size_t)n + 1, sizeof(double));
  int ret;
  ctl->xrow = cost;
  ctl->xg = g;
  for (;;) {
    int ninf = 0;
    for (int i = 1; i <= m; i++) {
      double beta = TT(P, i, 0), lb = P->blb[i], ub = P->bub[i];
      int gn = 0;
      if (lb > -INF && beta < lb - ctl->tol_bnd * (1.0 + fabs(lb))) gn = 1;
      if (ub < INF && beta > ub + ctl->tol_bnd * (1.0 + fabs(ub))) gn = -1;
      if (gn != g[i]) {
        const double d = (double)(gn - g[i]);
        const double *row = &TT(P, i, 0);
        for (int j = 0; j <= n; j++) cost[j] = fma(d, row[j], cost[j]);
        g[i] = gn;
      }
      if (gn) ninf++;
    }
    if (ninf == 0) { ret = R_PFEAS; break; }
    int sdir = 0;
    int q = price(P, cost, 1.0, ctl->tol_dj, &sdir, bland_on(ctl), NULL);
    if (q == 0) { ret = R_NOFEAS; break; }
    if (ctl->budget == 0) { ret = R_ITLIM; break; }
    int r = primal_step(P, ctl, q, sdir, g);
    if (r == R_UNBND) { ret = R_FAIL; break; }
  }
  ctl->xrow = NULL;
  ctl->xg = NULL;
  free(g); free(cost);
  return ret;
}

/* Dual simplex with devex pricing (Forrest & Goldfarb's reference framework, reset on entry): row i is
   scored viol_i^2 / w_i; after the pivot (p,q) is chosen, w_i = max(w_i, (a_iq/a_pq)^2 w_p) for i != p and
   w_p = max(w_p / a_pq^2, 1).  Cuts the pivots of a warm-started child solve by a quarter to a half against
   "largest infeasibility" (B&B on dense_ilp 128x256: 13587 -> 9171 pivots for 400 nodes). */
static int dual_simplex(orc_prob *P, ctl_t *ctl) {
  int n = P->n, m = P->m;
  double sgn = (P->dir == ORC_MAX) ? 1.0 : -1.0;
  double *w = (double *)xcalloc((size_t)m + 1, sizeof(double));
  int ret;
  for (int i = 0; i <= m; i++) w[i] = 1.0;
  for (;;) {
    int to_upper = 0;
    const int bland = bland_on(ctl);
    int p = select_infeasible_row(P, ctl->tol_bnd, &to_upper, bland, w);
    if (p == 0) { ret = R_PFEAS; break; }
    if (ctl->budget == 0) { ret = R_ITLIM; break; }
    int need_inc = !to_upper; /* below lb: the basic variable must increase */
    int q = 0;
    double br = 0.0, bmag = 0.0;
    for (int j = 1; j <= n; j++) {
      int f = P->nflag[j];
      if (f == ORC_NS) continue;
      double a = TT(P, p, j);
      double aa = need_inc ? a : -a; /* > 0: raising x_j helps; < 0: lowering x_j helps */
      double d = sgn * TT(P, 0, j);
      double r;
      if (aa > ctl->tol_piv && (f == ORC_NL || f == ORC_NF)) {
        r = (f == ORC_NF) ? fabs(d) : (d < 0.0 ? -d : 0.0);
      } else if (aa < -ctl->tol_piv && (f == ORC_NU || f == ORC_NF)) {
        r = (f == ORC_NF) ? fabs(d) : (d > 0.0 ? d : 0.0);
      } else
        continue;
      double mag = fabs(a);
      r = xdiv(r, mag);
      if (bland) mag = -(double)P->nvar[j]; /* tie-break among equal ratios */
      if (better(r, mag, j, br, bmag, q)) {
        br = r;
        bmag = mag;
        q = j;
      }
    }
    if (q == 0) { ret = R_NOFEAS; break; }
    {
      const double apq = TT(P, p, q), wp = w[p];
      for (int i = 1; i <= m; i++) {
        if (i == p) continue;
        double r = xdiv(TT(P, i, q), apq);
        double c = r * r * wp;
        if (c > w[i]) w[i] = c;
      }
      double c = xdiv(wp, apq * apq);
      w[p] = c > 1.0 ? c : 1.0;
    }
    double bound = to_upper ? P->bub[p] : P->blb[p];
    pivot(P, p, q, bound, leave_flag_for(P->blb[p], P->bub[p], to_upper), NULL);
    if (ctl->budget > 0) ctl->budget--;
    if (bland) P->bland_cnt++;
    ctl->stall = (br <= DEGEN_TOL) ? ctl->stall + 1 : 0;
  }
  free(w);
  return ret;
}

/* ------------------------------------------------------------ tableau refresh
   A dense tableau carries the rounding of every pivot it has been through (GLPK refactorises its basis behind
   glp_simplex; a Gauss-Jordan tableau has nothing to refactorise).  Rule, identical in the HIP engine: a solve that
   ends OPTIMAL on a handle with at least `check_every` pivots since the last look computes the residual of the row
   equations, max_i |sum_j A_ij x_j - x_Ri| / (1 + |x_Ri|), over 32 rows picked from the pivot count; above `tol` the tableau is REBUILT from the model for the
   same basis -- slack tableau with every non-basic variable on the bound it sits at, then the basic structural
   variables pivoted back in, in ascending variable number, each on the row of largest |entry| (lowest row on ties)
   among the rows whose auxiliary has to leave -- and the simplex carries on from there. */
static int g_check_every = 1024;
static double g_refresh_tol = 1e-9;
void orc_set_refresh(int check_every, double tol) {
  g_check_every = check_every > 0 ? check_every : 1024;
  g_refresh_tol = tol >= 0.0 ? tol : 1e-9;
}
int orc_get_refresh_cnt(const orc_prob *P) { return P->refresh_cnt; }

static double var_value(const orc_prob *P, const int *pos, int k) {
  int q = pos[k];
  if (q > 0) return TT(P, q, 0);
  double lb, ub;
  if (k <= P->m) { lb = P->rlb[k]; ub = P->rub[k]; }
  else { lb = P->clb[k - P->m]; ub = P->cub[k - P->m]; }
  return nb_value(P->nflag[-q], lb, ub);
}

/* residual over `rows` rows picked from the pivot count (rows <= 0 or >= m: every row) */
static double row_residual_sample(const orc_prob *P, int rows);
double orc_row_residual(const orc_prob *P) { return row_residual_sample(P, 0); }

#define REFRESH_SAMPLE_ROWS 32 /* the look that decides on a refresh reads this many rows, not all m (one look costs
                                  O(rows * n) on the host; a lineage of B&B nodes takes one every check_every pivots) */
static double row_residual_sample(const orc_prob *P, int rows) {
  if (!P->valid) return 0.0;
  int m = P->m, n = P->n;
  int *pos = (int *)xcalloc((size_t)m + n + 1, sizeof(int));
  for (int i = 1; i <= m; i++) pos[P->bvar[i]] = i;
  for (int j = 1; j <= n; j++) pos[P->nvar[j]] = -j;
  double *x = (double *)xcalloc((size_t)n + 1, sizeof(double));
  for (int j = 1; j <= n; j++) x[j] = var_value(P, pos, m + j);
  double worst = 0.0;
  int all = rows <= 0 || rows >= m;
  int cnt = all ? m : rows;
  for (int k = 0; k < cnt; k++) {
    int i = all ? k + 1 : 1 + (int)(((unsigned)P->it_cnt * 2654435761u + (unsigned)k * 0x9E3779B1u) % (unsigned)m);
    double acc = 0.0;
    const double *a = P->A[i];
    for (int j = 1; j <= n; j++) acc = acc + a[j] * x[j];
    double xr = var_value(P, pos, i);
    double r = fabs(acc - xr) / (1.0 + fabs(xr));
    if (r > worst) worst = r;
  }
  free(pos); free(x);
  return worst;
}

static void refresh_tableau(orc_prob *P) {
  int m = P->m, n = P->n;
  /* target: non-basic status by variable number, 0 = basic */
  int *tflag = (int *)xcalloc((size_t)m + n + 1, sizeof(int));
  for (int j = 1; j <= n; j++) tflag[P->nvar[j]] = P->nflag[j];
  int *sflag = (int *)xcalloc((size_t)n + 1, sizeof(int));
  for (int j = 1; j <= n; j++) sflag[j] = tflag[m + j] ? tflag[m + j] : std_flag(P->ctype[j]);
  int it_keep = P->it_cnt, status = P->status;
  build_slack_tableau_flags(P, sflag);
  for (int k = m + 1; k <= m + n; k++) {
    if (tflag[k]) continue; /* non-basic in the target basis */
    int q = 0;
    for (int j = 1; j <= n; j++)
      if (P->nvar[j] == k) { q = j; break; }
    if (!q) continue;
    int p = 0;
    double best = 0.0;
    for (int i = 1; i <= m; i++) {
      int v = P->bvar[i];
      if (v > m || !tflag[v]) continue; /* only rows whose auxiliary has to leave */
      double mag = fabs(TT(P, i, q));
      if (mag > best) { best = mag; p = i; } /* strict >: lowest row on ties */
    }
    if (!p) continue; /* numerically singular for this column: the simplex run that follows sorts it out */
    int tf = tflag[P->bvar[p]];
    double bound = (tf == ORC_NU) ? P->bub[p] : (tf == ORC_NF ? 0.0 : P->blb[p]);
    pivot(P, p, q, bound, tf, NULL);
  }
  P->it_cnt = it_keep;
  P->status = status;
  P->refresh_cnt++;
  free(tflag); free(sflag);
}

static double g_tol_bnd = 1e-9, g_tol_dj = 1e-9, g_tol_piv = 1e-9; /* behind parm == NULL; see mvx_set_default_tolerances */
void orc_set_default_tolerances(double tol_bnd, double tol_dj, double tol_piv) {
  g_tol_bnd = tol_bnd;
  g_tol_dj = tol_dj;
  g_tol_piv = tol_piv;
}

void orc_init_smcp(orc_smcp *parm) {
  parm->msg_lev = 0;
  parm->meth = 1;
  parm->it_lim = -1;
  parm->tol_bnd = g_tol_bnd;
  parm->tol_dj = g_tol_dj;
  parm->tol_piv = g_tol_piv;
}

static int simplex_once(orc_prob *P, const orc_smcp *parm);

int orc_simplex(orc_prob *P, const orc_smcp *parm) {
  int before = P->it_cnt;
  int rc = simplex_once(P, parm);
  P->piv_since_check += P->it_cnt - before;
  if (rc == 0 && P->status == ORC_OPT && P->piv_since_check >= g_check_every) {
    P->piv_since_check = 0;
    if (row_residual_sample(P, REFRESH_SAMPLE_ROWS) > g_refresh_tol) {
      refresh_tableau(P);
      before = P->it_cnt;
      rc = simplex_once(P, parm); /* the pivot limit of the call, if any, applies to this leg afresh */
      P->piv_since_check += P->it_cnt - before;
    }
  }
  return rc;
}

static int simplex_once(orc_prob *P, const orc_smcp *parm) {
  orc_smcp dflt;
  if (!parm) {
    orc_init_smcp(&dflt);
    parm = &dflt;
  }
  if (P->m < 1 || P->n < 1) {
    P->status = ORC_UNDEF;
    return ORC_EFAIL;
  }
  if (!P->valid) build_slack_tableau(P);
  /* no limit asked for: a safety cap stands in (belt and braces behind the anti-cycling rule: a solve
     must end with EITLIM rather than spin).  Same formula in the HIP engine. */
  int budget = parm->it_lim >= 0 ? parm->it_lim : 200 * (P->m + P->n) + 100000;
  int stall_limit = g_stall_limit > 0 ? g_stall_limit : 64 + (P->m + P->n) / 8;
  ctl_t ctl = {parm->tol_bnd, parm->tol_dj, parm->tol_piv, budget, 0, stall_limit, 0, 0, NULL, NULL, NULL};
  double sgn = (P->dir == ORC_MAX) ? 1.0 : -1.0;
  for (int round = 0; round < 64; round++) {
    int to_upper = 0, sdir = 0;
    int p = select_infeasible_row(P, ctl.tol_bnd, &to_upper, 0, NULL); /* existence only */
    int r;
    if (p == 0) {
      r = primal_phase2(P, &ctl);
      if (r == R_OPT) {
        if (select_infeasible_row(P, ctl.tol_bnd, &to_upper, 0, NULL) == 0) {
          P->status = ORC_OPT;
          return 0;
        }
        continue;
      }
      if (r == R_UNBND) { P->status = ORC_UNBND; return 0; }
      P->status = ORC_FEAS;
      return ORC_EITLIM;
    }
    int q = price(P, &TT(P, 0, 0), sgn, ctl.tol_dj, &sdir, 0, NULL); /* existence only */
    r = (q == 0) ? dual_simplex(P, &ctl) : primal_phase1(P, &ctl);
    if (r == R_PFEAS) continue;
    if (r == R_NOFEAS) { P->status = ORC_NOFEAS; return 0; }
    if (r == R_ITLIM) { P->status = ORC_INFEAS; return ORC_EITLIM; }
    P->status = ORC_UNDEF;
    return ORC_EFAIL;
  }
  P->status = ORC_UNDEF;
  return ORC_EFAIL;
}

int orc_simplex_batch(orc_prob **probs, int count, const orc_smcp *parm, int *rcs) {
  for (int i = 0; i < count; i++) {
    int rc = orc_simplex(probs[i], parm);
    if (rcs) rcs[i] = rc;
  }
  return 0;
}

/* -------------------------------------------------------------------- query */
int orc_get_obj_dir(const orc_prob *P) { return P->dir; }
int orc_get_num_rows(const orc_prob *P) { return P->m; }
int orc_get_num_cols(const orc_prob *P) { return P->n; }
int orc_get_num_int(const orc_prob *P) {
  int k = 0;
  for (int j = 1; j <= P->n; j++) k += (P->kind[j] == ORC_IV);
  return k;
}
int orc_get_status(const orc_prob *P) { return P->status; }
double orc_get_obj_val(const orc_prob *P) { return P->valid ? TT(P, 0, 0) : P->c[0]; }
double orc_get_obj_coef(const orc_prob *P, int j) {
  if (j < 0 || j > P->n) fault("get_obj_coef: column out of range");
  return P->c[j];
}

static double var_prim(const orc_prob *P, int k) {
  if (!P->valid) return 0.0;
  int pos = var_pos(P, k);
  if (pos > 0) return TT(P, pos, 0);
  return nb_value(P->nflag[-pos], P->nlb[-pos], P->nub[-pos]);
}
static double var_dual(const orc_prob *P, int k) {
  if (!P->valid) return 0.0;
  int pos = var_pos(P, k);
  return pos > 0 ? 0.0 : TT(P, 0, -pos);
}
static int var_stat(const orc_prob *P, int k, int type) {
  if (!P->valid) return (k <= P->m) ? ORC_BS : std_flag(type);
  int pos = var_pos(P, k);
  return pos > 0 ? ORC_BS : P->nflag[-pos];
}

double orc_get_col_prim(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_prim: column out of range");
  return var_prim(P, P->m + j);
}
double orc_get_row_prim(const orc_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_prim: row out of range");
  return var_prim(P, i);
}
double orc_get_col_dual(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_dual: column out of range");
  return var_dual(P, P->m + j);
}
double orc_get_row_dual(const orc_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_dual: row out of range");
  return var_dual(P, i);
}
int orc_get_col_stat(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_stat: column out of range");
  return var_stat(P, P->m + j, P->ctype[j]);
}
int orc_get_row_stat(const orc_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_stat: row out of range");
  return var_stat(P, i, P->rtype[i]);
}
int orc_get_col_kind(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_kind: column out of range");
  /* GLPK reports an integer column with bounds [0,1] as GLP_BV [GLPK-recalled]; this is
     what makes gmi.cpp:18 reject binaries while util.cpp:444 still branches on them */
  if (P->kind[j] == ORC_IV && P->ctype[j] == ORC_DB && P->clb[j] == 0.0 && P->cub[j] == 1.0)
    return ORC_BV;
  return P->kind[j];
}
int orc_get_row_type(const orc_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_type: row out of range");
  return P->rtype[i];
}
/* absent bounds read back as -/+DBL_MAX [GLPK-recalled]; consumed arithmetically at
   gmi.cpp:73 */
double orc_get_row_lb(const orc_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_lb: row out of range");
  return P->rlb[i] == -INF ? -DBL_MAX : P->rlb[i];
}
double orc_get_row_ub(const orc_prob *P, int i) {
  if (i < 1 || i > P->m) fault("get_row_ub: row out of range");
  return P->rub[i] == INF ? DBL_MAX : P->rub[i];
}
int orc_get_col_type(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_type: column out of range");
  return P->ctype[j];
}
double orc_get_col_lb(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_lb: column out of range");
  return P->clb[j] == -INF ? -DBL_MAX : P->clb[j];
}
double orc_get_col_ub(const orc_prob *P, int j) {
  if (j < 1 || j > P->n) fault("get_col_ub: column out of range");
  return P->cub[j] == INF ? DBL_MAX : P->cub[j];
}

int orc_get_mat_row(const orc_prob *P, int i, int *ind, double *val) {
  /* non-zeros in ascending column order (GLPK's own order is unobservable here,
     SURVEY.md section 8(c) "unverifiable hazards") */
  if (i < 1 || i > P->m) fault("get_mat_row: row out of range");
  int len = 0;
  for (int j = 1; j <= P->n; j++) {
    if (P->A[i][j] != 0.0) {
      len++;
      if (ind) ind[len] = j;
      if (val) val[len] = P->A[i][j];
    }
  }
  return len;
}

int orc_eval_tab_row(const orc_prob *P, int k, int *ind, double *val) {
  /* gmi.cpp:36: row of the simplex tableau for basic variable k over the non-basic
     variables; non-zeros only, in ascending non-basic column position */
  if (!P->valid) fault("eval_tab_row: basis does not exist");
  if (k < 1 || k > P->m + P->n) fault("eval_tab_row: variable out of range");
  int pos = var_pos(P, k);
  if (pos <= 0) fault("eval_tab_row: variable must be basic");
  int len = 0;
  for (int j = 1; j <= P->n; j++) {
    double a = TT(P, pos, j);
    if (a != 0.0) {
      len++;
      ind[len] = P->nvar[j];
      val[len] = a;
    }
  }
  return len;
}

int orc_get_it_cnt(const orc_prob *P) { return P->it_cnt; }
int orc_get_bland_cnt(const orc_prob *P) { return P->bland_cnt; }
int orc_get_pert_cnt(const orc_prob *P) { return P->pert_cnt; }
int orc_term_out(int flag) {
  int old = g_term_out;
  g_term_out = flag;
  return old;
}
const char *orc_version(void) { return "mvolps-oracle 1.0 (dense tableau; GLPK-shaped API)"; }

int orc_get_tableau_ld(const orc_prob *P) { return P->ld; }
int orc_get_tableau(const orc_prob *P, double *out) {
  if (!P->valid) return -1;
  for (int i = 0; i <= P->m; i++)
    memcpy(&out[(size_t)i * (P->n + 1)], &TT(P, i, 0), (size_t)(P->n + 1) * sizeof(double));
  return 0;
}
int orc_get_basis(const orc_prob *P, int *head, int *nb, int *flag) {
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

/* ------------------------------------------------------------- pack / unpack */
/* Host-memory counterpart of mvx_pack_from / mvx_unpack (world_size-2 gloo tests run the coordinator over this
   library): rows 1..m_base of the model are the receiver's own, the rows appended since (cut rows) travel. */
typedef struct {
  long long magic, m, n, ld, status, it_cnt, valid, m_base, piv_since_check, reserved;
} pack_hdr;
#define PACK_MAGIC 0x4d56584f5244ll

static long long pack_size_from(const orc_prob *P, int m_base) {
  long long m = P->m, n = P->n;
  long long sz = sizeof(pack_hdr);
  sz += (long long)sizeof(int) * ((n + 1) + (m + 1) + (m + 1) + 2 * (n + 1));
  sz = (sz + 7) / 8 * 8;
  sz += (long long)sizeof(double) * (2 * (n + 1) + 2 * (m + 1));
  sz += (long long)sizeof(double) * (m - m_base) * (n + 1);
  if (P->valid) sz += (long long)sizeof(double) * ((m + 1) * (long long)P->ld + 2 * (m + 1) + 2 * (n + 1));
  return sz;
}

static int pack_from(const orc_prob *P, int m_base, void *buf) {
  int m = P->m, n = P->n;
  if (m_base < 0 || m_base > m) return -1;
  unsigned char *b = (unsigned char *)buf;
  pack_hdr h = {PACK_MAGIC, m, n, P->ld, P->status, P->it_cnt, P->valid, m_base, P->piv_since_check, 0};
  memcpy(b, &h, sizeof(h)); b += sizeof(h);
  unsigned char *b0 = b;
#define PUT(ptr, cnt, T) do { memcpy(b, (ptr), (size_t)(cnt) * sizeof(T)); b += (size_t)(cnt) * sizeof(T); } while (0)
  PUT(P->ctype, n + 1, int);
  PUT(P->rtype, m + 1, int);
  if (P->valid) { PUT(P->bvar, m + 1, int); PUT(P->nvar, n + 1, int); PUT(P->nflag, n + 1, int); }
  else b += sizeof(int) * ((size_t)(m + 1) + 2 * (size_t)(n + 1));
  b = b0 + ((size_t)(b - b0) + 7) / 8 * 8;
  PUT(P->clb, n + 1, double); PUT(P->cub, n + 1, double);
  PUT(P->rlb, m + 1, double); PUT(P->rub, m + 1, double);
  for (int i = m_base + 1; i <= m; i++) PUT(P->A[i], n + 1, double);
  if (P->valid) {
    PUT(P->T, (size_t)(m + 1) * P->ld, double);
    PUT(P->blb, m + 1, double); PUT(P->bub, m + 1, double);
    PUT(P->nlb, n + 1, double); PUT(P->nub, n + 1, double);
  }
#undef PUT
  return 0;
}

long long orc_pack_size(const orc_prob *P) { return pack_size_from(P, P->m); }
int orc_pack(const orc_prob *P, void *buf) { return pack_from(P, P->m, buf); }
long long orc_pack_size_from(const orc_prob *P, const orc_prob *base) { return pack_size_from(P, base ? base->m : P->m); }
int orc_pack_from(const orc_prob *P, const orc_prob *base, void *buf) { return pack_from(P, base ? base->m : P->m, buf); }

int orc_unpack(orc_prob *dst, const orc_prob *base, const void *buf) {
  const unsigned char *b = (const unsigned char *)buf;
  pack_hdr h;
  memcpy(&h, b, sizeof(h)); b += sizeof(h);
  if (h.magic != PACK_MAGIC || h.m_base != base->m || h.n != base->n || h.m < h.m_base) return -1;
  int m = (int)h.m, n = (int)h.n, m_base = (int)h.m_base;
  /* model rows 1..m_base / objective / kinds come from the receiver's copy of the root problem */
  {
    int was_valid = base->valid;
    ((orc_prob *)base)->valid = 0; /* copy the model only */
    orc_copy_prob(dst, base, ORC_ON);
    ((orc_prob *)base)->valid = was_valid;
  }
  if (m > m_base) orc_add_rows(dst, m - m_base);
  const unsigned char *b0 = b;
#define GET(ptr, cnt, T) do { memcpy((ptr), b, (size_t)(cnt) * sizeof(T)); b += (size_t)(cnt) * sizeof(T); } while (0)
  GET(dst->ctype, n + 1, int);
  GET(dst->rtype, m + 1, int);
  if (h.valid) { GET(dst->bvar, m + 1, int); GET(dst->nvar, n + 1, int); GET(dst->nflag, n + 1, int); }
  else b += sizeof(int) * ((size_t)(m + 1) + 2 * (size_t)(n + 1));
  b = b0 + ((size_t)(b - b0) + 7) / 8 * 8;
  GET(dst->clb, n + 1, double); GET(dst->cub, n + 1, double);
  GET(dst->rlb, m + 1, double); GET(dst->rub, m + 1, double);
  for (int i = m_base + 1; i <= m; i++) GET(dst->A[i], n + 1, double);
  dst->status = (int)h.status;
  dst->it_cnt = (int)h.it_cnt;
  dst->piv_since_check = (int)h.piv_since_check;
  dst->valid = (int)h.valid;
  if (h.valid) {
    dst->ld = (int)h.ld;
    free(dst->T);
    dst->T = (double *)xcalloc((size_t)(dst->m_cap + 1) * dst->ld, sizeof(double));
    GET(dst->T, (size_t)(m + 1) * dst->ld, double);
    GET(dst->blb, m + 1, double); GET(dst->bub, m + 1, double);
    GET(dst->nlb, n + 1, double); GET(dst->nub, n + 1, double);
  }
#undef GET
  return 0;
}
