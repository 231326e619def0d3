/*
 * mvolps_oracle_bnb.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the MVOLPS-owned decisions around the LP engine, written against the
 * public orc_* API exactly where the reference calls glp_*:
 *   getFract           /root/reference/util.cpp:11-23
 *   printInfo          /root/reference/util.cpp:414-473
 *   pickNode / pickVar /root/reference/util.cpp:154-230
 *   generateCut3       /root/reference/gmi.cpp:11-117
 *   CutPool            /root/reference/cut.cpp:6-46
 *   branchAndBound     /root/reference/bs.cpp:54-348
 * Reference quirks (SURVEY.md section 3.2 notes A-G) are reproduced, not repaired.
 */
#include "mvolps_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz);
  if (!p) abort();
  return p;
}
static void *xrealloc(void *p, size_t sz) {
  p = realloc(p, sz ? sz : 1);
  if (!p) abort();
  return p;
}

/* util.cpp:11-23 */
double orc_getFract(double x) {
  double ip;
  double f = modf(x, &ip);
  if (f < 0.0) f += 1;
  return f;
}

/* util.cpp:414-473 */
int orc_printInfo(const orc_prob *P, int *violated, int *nviolated) {
  return orc_printInfo_ex(P, 1, violated, nviolated);
}

int orc_printInfo_ex(const orc_prob *P, int quirks, int *violated, int *nviolated) {
  int cols = orc_get_num_cols(P);
  *nviolated = 0;
  int status = orc_get_status(P);
  if (status == ORC_NOFEAS || status == ORC_INFEAS || status == ORC_UNBND) return -1; /* :424 */
  for (int i = 1; i <= cols; i++) {
    double v = orc_get_col_prim(P, i);
    if (!quirks) {
      if (fabs(v - round(v)) > 1e-9 && orc_get_col_kind(P, i) != ORC_CV) violated[(*nviolated)++] = i;
      continue;
    }
    if (v != 0 && orc_get_obj_coef(P, i) != 0) {                  /* :437 */
      if (trunc(v) != v && orc_get_col_kind(P, i) != ORC_CV) {    /* :443-444 */
        violated[(*nviolated)++] = i;
      }
    }
  }
  return (*nviolated == 0) ? 1 : 0;
}

/* gmi.cpp:11-117 */
int orc_generateCut3(const orc_prob *P, int j, int *inds, double *vals, double *lb) {
  double temp = 0.0; /* uninitialised in the reference (gmi.cpp:13); 0 here */
  int m = orc_get_num_rows(P);
  int n = orc_get_num_cols(P);
  if (orc_get_col_kind(P, j) != ORC_IV) return -1; /* :18 */
  if (orc_get_col_stat(P, j) != ORC_BS) return -1; /* :23 */

  double *work = (double *)xcalloc((size_t)m + n + 1, sizeof(double));
  double *val2 = (double *)xcalloc((size_t)n + 1, sizeof(double));
  int *ind2 = (int *)xcalloc((size_t)n + 1, sizeof(int));
  int len = orc_eval_tab_row(P, m + j, ind2, val2); /* :36 */
  double rhs = orc_get_col_prim(P, j);              /* :37 */
  for (int i = 1; i <= len; i++) {
    double val = val2[i];
    int kind;
    double ub;
    if (ind2[i] <= m) { /* :42-47 */
      kind = ORC_CV;
      ub = orc_get_row_ub(P, ind2[i]);
    } else { /* :48-53 */
      int cc = ind2[i] - m;
      kind = orc_get_col_kind(P, cc);
      ub = orc_get_col_ub(P, cc);
    }
    double fRhs = orc_getFract(rhs); /* rhs is the RUNNING value (:55,:73) */
    double fVal = orc_getFract(val);
    if (kind == ORC_IV) {
      if (fRhs >= fVal) temp = fVal;
      else temp = (fRhs / (1.0 - fRhs)) * (1.0 - fVal);
    }
    if (kind == ORC_CV) {
      if (val >= 0.0) temp = val;
      else temp = (fRhs / (1.0 - fRhs)) * (-1.0 * val);
    }
    work[ind2[i]] = -1.0 * temp; /* :72 */
    rhs -= temp * ub;            /* :73 */
  }
  /* back-substitution, :81-89 -- uses POSITION k of the row's non-zero list, not ind[k] */
  double *rv = (double *)xcalloc((size_t)n + 1, sizeof(double));
  int *ri = (int *)xcalloc((size_t)n + 1, sizeof(int));
  for (int i = 1; i <= m; i++) {
    int len2 = orc_get_mat_row(P, i, ri, rv);
    for (int k = 1; k <= len2; k++) work[m + k] += work[i] * rv[k];
  }
  inds[0] = 0;
  vals[0] = rhs;
  for (int i = 1; i <= n; i++) {
    inds[i] = i;
    vals[i] = work[m + i];
  }
  *lb = rhs;
  free(work); free(val2); free(ind2); free(rv); free(ri);
  return 0;
}

/* Repaired GMI (not the reference's formula).  Tableau row of the basic
   column: x_B = sum_j alpha_j x_Nj.  With y_j = x_j - l_j (at lower) or u_j - x_j (at upper) >= 0:
   x_B + sum_j abar_j y_j = beta, abar_j = -alpha_j (lower) / +alpha_j (upper).  f0 = frac(beta);
   integer y_j: g_j = f_j/f0 if f_j <= f0 else (1-f_j)/(1-f0), f_j = frac(abar_j);
   continuous:  g_j = abar_j/f0 if abar_j >= 0 else -abar_j/(1-f0).   Cut: sum_j g_j y_j >= 1. */
int orc_generateCutGMI(const orc_prob *P, int j, int *inds, double *vals, double *lb, double *efficacy) {
  int m = orc_get_num_rows(P), n = orc_get_num_cols(P);
  if (orc_get_col_kind(P, j) == ORC_CV) return -1;
  if (orc_get_col_stat(P, j) != ORC_BS) return -1;
  double beta = orc_get_col_prim(P, j);
  double f0 = orc_getFract(beta);
  if (f0 < 1e-6 || f0 > 1.0 - 1e-6) return -1;
  double *val2 = (double *)xcalloc((size_t)n + 1, sizeof(double));
  int *ind2 = (int *)xcalloc((size_t)n + 1, sizeof(int));
  double *work = (double *)xcalloc((size_t)m + n + 1, sizeof(double)); /* coefficients on x (rows 1..m, cols m+1..) */
  int len = orc_eval_tab_row(P, m + j, ind2, val2);
  double rhs = 1.0;
  int ok = 1;
  for (int t = 1; t <= len && ok; t++) {
    int k = ind2[t];
    double alpha = val2[t];
    int stat, isint;
    double lo, up;
    if (k <= m) {
      stat = orc_get_row_stat(P, k);
      isint = 0;
      lo = orc_get_row_lb(P, k);
      up = orc_get_row_ub(P, k);
    } else {
      stat = orc_get_col_stat(P, k - m);
      isint = orc_get_col_kind(P, k - m) != ORC_CV;
      lo = orc_get_col_lb(P, k - m);
      up = orc_get_col_ub(P, k - m);
    }
    if (stat == ORC_NS) continue;             /* fixed: y_j = 0 */
    if (stat == ORC_NF) { ok = 0; break; }     /* free non-basic with a non-zero entry: no valid cut */
    double abar = (stat == ORC_NL) ? -alpha : alpha;
    double g;
    if (isint) {
      double fj = orc_getFract(abar);
      g = (fj <= f0) ? fj / f0 : (1.0 - fj) / (1.0 - f0);
    } else {
      g = (abar >= 0.0) ? abar / f0 : -abar / (1.0 - f0);
    }
    /* g*y_j with y_j = x - lo  or  up - x */
    if (stat == ORC_NL) { work[k] += g; rhs += g * lo; }
    else { work[k] -= g; rhs -= g * up; }
  }
  if (ok) {
    /* auxiliary variables are row activities: x_i = sum_k A_ik x_(m+k) */
    double *rv = (double *)xcalloc((size_t)n + 1, sizeof(double));
    int *ri = (int *)xcalloc((size_t)n + 1, sizeof(int));
    for (int i = 1; i <= m; i++) {
      if (work[i] == 0.0) continue;
      int len2 = orc_get_mat_row(P, i, ri, rv);
      for (int t = 1; t <= len2; t++) work[m + ri[t]] += work[i] * rv[t];
    }
    free(rv); free(ri);
    double dot = 0.0, nrm = 0.0;
    inds[0] = 0;
    vals[0] = rhs;
    for (int k = 1; k <= n; k++) {
      inds[k] = k;
      vals[k] = work[m + k];
      dot += vals[k] * orc_get_col_prim(P, k);
      nrm += vals[k] * vals[k];
    }
    *lb = rhs;
    *efficacy = (nrm > 0.0) ? (rhs - dot) / sqrt(nrm) : 0.0;
    if (!(nrm > 0.0)) ok = 0;
  }
  free(val2); free(ind2); free(work);
  return ok ? 0 : -1;
}

/* ----------------------------------------------------------------- cut pool */
typedef struct {
  int *inds;
  double *vals;
  double lb;
  int len; /* n + 1 */
} cut_t;
typedef struct {
  cut_t *cuts;
  int n, cap;
} pool_t;

static void pool_add(pool_t *pl, cut_t c) { /* cut.cpp:6-9 */
  if (pl->n == pl->cap) {
    pl->cap = pl->cap ? pl->cap * 2 : 16;
    pl->cuts = (cut_t *)xrealloc(pl->cuts, (size_t)pl->cap * sizeof(cut_t));
  }
  pl->cuts[pl->n++] = c;
}
static int pool_add_cut_constraint(pool_t *pl, orc_prob *in) { /* cut.cpp:11-46 */
  if (pl->n == 0) return -1;
  int cID = pl->n - 1; /* :20 "last cut" */
  int index = orc_add_rows(in, 1);
  cut_t *c = &pl->cuts[cID];
  orc_set_mat_row(in, index, c->len - 1, c->inds, c->vals); /* :40 */
  orc_set_row_bnds(in, index, ORC_LO, c->lb, 0);             /* :43 */
  return cID;
}
static void pool_free(pool_t *pl) {
  for (int i = 0; i < pl->n; i++) {
    free(pl->cuts[i].inds);
    free(pl->cuts[i].vals);
  }
  free(pl->cuts);
}

/* -------------------------------------------------------------- B&B driver */
typedef struct {
  int oid;
  double upperBound, lowerBound;
  orc_prob *prob;
  int inital;
} node_t;

typedef struct {
  orc_bnb_result *res;
  int ev_cap, node_cap;
  int next_id; /* util.h:17 `static int id = 1` */
} ctx_t;

static void ensure_node(ctx_t *cx, int oid) {
  orc_bnb_result *r = cx->res;
  if (oid >= cx->node_cap) {
    int cap = cx->node_cap ? cx->node_cap : 64;
    while (cap <= oid) cap *= 2;
    r->parent = (int *)xrealloc(r->parent, (size_t)cap * sizeof(int));
    r->prune = (int *)xrealloc(r->prune, (size_t)cap * sizeof(int));
    r->node_bound = (double *)xrealloc(r->node_bound, (size_t)cap * sizeof(double));
    cx->node_cap = cap;
  }
}

static node_t *node_new(ctx_t *cx, const orc_prob *parent, int parent_oid) { /* util.cpp:25-37 */
  node_t *nd = (node_t *)xcalloc(1, sizeof(node_t));
  nd->oid = cx->next_id++;
  nd->lowerBound = -HUGE_VAL;
  nd->upperBound = HUGE_VAL;
  nd->prob = orc_create_prob();
  orc_copy_prob(nd->prob, parent, ORC_ON);
  nd->inital = 0;
  ensure_node(cx, nd->oid);
  cx->res->parent[nd->oid] = parent_oid;
  cx->res->prune[nd->oid] = 4; /* NONE */
  cx->res->node_bound[nd->oid] = nd->upperBound;
  if (nd->oid > cx->res->n_nodes) cx->res->n_nodes = nd->oid;
  return nd;
}
static void node_free(node_t *nd) { /* util.cpp:39-42 */
  orc_delete_prob(nd->prob);
  free(nd);
}

static int branch_direction(int oid) { /* bs.cpp:43-52 */
  if (oid <= 1) return 0;
  return (oid % 2 == 0) ? 1 : 2;
}

static void emit(ctx_t *cx, int type, int oid, double f6, double f7, int f8, int pick) {
  orc_bnb_result *r = cx->res;
  if (r->n_events == cx->ev_cap) {
    cx->ev_cap = cx->ev_cap ? cx->ev_cap * 2 : 256;
    r->events = (orc_bnb_event *)xrealloc(r->events, (size_t)cx->ev_cap * sizeof(orc_bnb_event));
  }
  orc_bnb_event *e = &r->events[r->n_events++];
  e->type = type;
  e->oid = oid;
  e->pid = r->parent[oid];
  e->direction = branch_direction(oid);
  e->lp_bound = f6;
  e->sum_infeas = f7;
  e->n_violated = f8;
  e->pick = pick;
}

/* util.cpp:154-188 */
static int pick_node(const orc_bnb_params *pr, node_t **q, int nq, double sg) {
  if (pr->node_strat == 0) return 0; /* "DFS" is problems.front(), util.cpp:165 */
  int best = 0;                      /* std::max_element: first maximum */
  for (int i = 1; i < nq; i++)
    if (sg * q[best]->upperBound < sg * q[i]->upperBound) best = i;
  return best;
}

/* util.cpp:190-230; `root` is ParameterObj::_prob, the never-solved root problem */
static int pick_var(const orc_bnb_params *pr, const orc_prob *root, const int *vars, int nv) {
  if (pr->var_strat == 0) return vars[0];
  if (pr->var_strat == 1) {
    double curBest = fabs(orc_getFract(orc_get_col_prim(root, vars[0])) - 0.5);
    int index = vars[0];
    for (int k = 0; k < nv; k++) {
      double cur = fabs(orc_getFract(orc_get_col_prim(root, vars[k])) - 0.5);
      if (cur < curBest) {
        curBest = cur;
        index = vars[k];
      }
    }
    return index;
  }
  double bestCoef = 0.0;
  int index = vars[0]; /* uninitialised in the reference when no coefficient is > 0 */
  for (int k = 0; k < nv; k++) {
    double cur = orc_get_obj_coef(root, vars[k]);
    if (cur > bestCoef) {
      bestCoef = cur;
      index = vars[k];
    }
  }
  return index;
}

void orc_bnb_default_params(orc_bnb_params *p) {
  p->var_strat = 0;  /* util.h:65 */
  p->node_strat = 0; /* util.h:66 */
  p->cut_strat = 0;  /* util.h:67 */
  p->cut_chance = 0.0;
  p->loop_limit = 200000; /* bs.cpp:320 */
  p->max_nodes = 0;
  p->reference_quirks = 1;
  p->cut_select = 0;
}

static int solve(ctx_t *cx, orc_prob *p) {
  int before = orc_get_it_cnt(p);
  int rc = orc_simplex(p, NULL); /* return code ignored by the reference */
  cx->res->total_pivots += orc_get_it_cnt(p) - before;
  return rc;
}

int orc_branchAndBound(orc_prob *prob, const orc_bnb_params *params, orc_bnb_result *res) {
  memset(res, 0, sizeof(*res));
  ctx_t cx;
  memset(&cx, 0, sizeof(cx));
  cx.res = res;
  cx.next_id = 1;
  pool_t pool;
  memset(&pool, 0, sizeof(pool));

  int n0 = orc_get_num_cols(prob);
  res->n = n0;
  res->x = (double *)xcalloc((size_t)n0 + 1, sizeof(double));
  int *vars = (int *)xcalloc((size_t)n0 + 1, sizeof(int));

  int qcap = 64, nq = 0;
  node_t **leaf = (node_t **)xcalloc((size_t)qcap, sizeof(node_t *));
  node_t *S1 = node_new(&cx, prob, 0); /* bs.cpp:80 */
  S1->inital = 1;
  leaf[nq++] = S1;

  orc_prob *a = orc_create_prob(); /* bs.cpp:89 */
  /* bs.cpp:172,210 compare as a maximiser whatever the direction; the repaired mode turns the compares
     round for a minimisation problem (sg = -1) */
  const double sg = (!params->reference_quirks && orc_get_obj_dir(prob) == ORC_MIN) ? -1.0 : 1.0;
  double bestLower = -sg * HUGE_VAL; /* bs.cpp:90 */
  int count = 0;

  while (nq > 0) { /* bs.cpp:96 */
    if (params->max_nodes > 0 && count >= params->max_nodes) {
      res->hit_limit = 1;
      break;
    }
    int index = pick_node(params, leaf, nq, sg);
    node_t *node = leaf[index];
    orc_erase_prob(a);                       /* bs.cpp:114-115 */
    orc_copy_prob(a, node->prob, ORC_OFF);   /* bs.cpp:116 */
    solve(&cx, a);                           /* bs.cpp:117 */
    emit(&cx, ORC_EV_PREGNANT, node->oid, orc_get_obj_val(a), 0.0, 0, 0); /* bs.cpp:119-129 */

    int nv = 0;
    int status = orc_printInfo_ex(a, params->reference_quirks, vars, &nv); /* bs.cpp:135|151 */
    if (node->inital) {
      if (status == -1) { /* bs.cpp:139-143 */
        res->prune[node->oid] = 1;
        break;
      }
      if (status == 1) { /* bs.cpp:144-149: leaves without recording the solution; repaired mode keeps it */
        node->upperBound = orc_get_obj_val(a);
        res->node_bound[node->oid] = node->upperBound;
        res->prune[node->oid] = 0;
        if (!params->reference_quirks) {
          bestLower = node->upperBound;
          res->has_incumbent = 1;
          res->incumbent_oid = node->oid;
          int na = orc_get_num_cols(a);
          for (int i = 1; i <= na && i <= n0; i++) res->x[i] = orc_get_col_prim(a, i);
        }
        break;
      }
    }
    node->upperBound = orc_get_obj_val(a); /* bs.cpp:156 */
    res->node_bound[node->oid] = node->upperBound;

    int erase = 1;
    if (status == 1) { /* bs.cpp:158-193 */
      res->prune[node->oid] = 0;
      emit(&cx, ORC_EV_INTEGER, node->oid, node->upperBound, 0.0, 0, 0);
      if (sg * node->upperBound > sg * bestLower) {
        bestLower = node->upperBound;
        res->has_incumbent = 1;
        res->incumbent_oid = node->oid;
        int na = orc_get_num_cols(a);
        for (int i = 1; i <= na && i <= n0; i++) res->x[i] = orc_get_col_prim(a, i);
      }
    } else if (status == -1) { /* bs.cpp:194-209 */
      res->prune[node->oid] = 1;
      emit(&cx, ORC_EV_INFEASIBLE, node->oid, 0.0, 0.0, 0, 0);
    } else if (sg * orc_get_obj_val(a) <= sg * bestLower) { /* bs.cpp:210-223 */
      res->prune[node->oid] = 3;
      emit(&cx, ORC_EV_FATHOMED, node->oid, 0.0, 0.0, 0, 0);
    } else { /* bs.cpp:224-324 */
      double acc = 0;
      for (int k = 0; k < nv; k++)
        if (vars[k] != 0) acc += orc_getFract(orc_get_col_prim(a, vars[k])); /* bs.cpp:229-233 */

      if (params->cut_strat != 0 && params->reference_quirks) { /* bs.cpp:249-258 */
        int na = orc_get_num_cols(a);
        for (int j = 1; j <= na; j++) {
          cut_t c;
          c.len = na + 1;
          c.inds = (int *)xcalloc((size_t)na + 1, sizeof(int));
          c.vals = (double *)xcalloc((size_t)na + 1, sizeof(double));
          if (orc_generateCut3(a, j, c.inds, c.vals, &c.lb) != -1) {
            pool_add(&pool, c);
          } else {
            free(c.inds);
            free(c.vals);
          }
        }
        pool_add_cut_constraint(&pool, a);
      } else if (params->cut_strat != 0) {
        /* repaired cuts: this node's own cuts only, chosen by cut_select / cut_chance */
        int na = orc_get_num_cols(a);
        pool_t local;
        memset(&local, 0, sizeof(local));
        double *eff = (double *)xcalloc((size_t)na + 1, sizeof(double));
        for (int j = 1; j <= na; j++) {
          cut_t c;
          c.len = na + 1;
          c.inds = (int *)xcalloc((size_t)na + 1, sizeof(int));
          c.vals = (double *)xcalloc((size_t)na + 1, sizeof(double));
          double e = 0.0;
          if (orc_generateCutGMI(a, j, c.inds, c.vals, &c.lb, &e) != -1) {
            eff[local.n] = e;
            pool_add(&local, c);
          } else {
            free(c.inds);
            free(c.vals);
          }
        }
        if (local.n > 0) {
          int take = 1;
          if (params->cut_select == 1) {
            take = (int)ceil(params->cut_chance * local.n);
            if (take < 1) take = 1;
            if (take > local.n) take = local.n;
          }
          char *used = (char *)xcalloc((size_t)local.n, 1);
          for (int t = 0; t < take; t++) {
            int best = -1;
            if (params->cut_select == 0) best = local.n - 1; /* cut.cpp:20: the last one */
            else
              for (int q = 0; q < local.n; q++)
                if (!used[q] && (best < 0 || eff[q] > eff[best])) best = q; /* ties: first generated */
            used[best] = 1;
            cut_t *cc = &local.cuts[best];
            int index = orc_add_rows(a, 1);
            orc_set_mat_row(a, index, cc->len - 1, cc->inds, cc->vals);
            orc_set_row_bnds(a, index, ORC_LO, cc->lb, 0);
          }
          free(used);
        }
        free(eff);
        pool_free(&local);
      }
      int pick = pick_var(params, prob, vars, nv);  /* bs.cpp:260 */
      double bound = orc_get_col_prim(a, pick);      /* bs.cpp:261 */
      emit(&cx, ORC_EV_BRANCHED, node->oid, node->upperBound, acc, nv, pick);

      node_t *S2 = node_new(&cx, a, node->oid); /* bs.cpp:269-273 */
      node_t *S3 = node_new(&cx, a, node->oid);
      if (params->reference_quirks) {
        orc_set_col_bnds(S2->prob, pick, ORC_UP, 0, floor(bound)); /* bs.cpp:274 */
      } else {
        int t = orc_get_col_type(a, pick);
        double l = orc_get_col_lb(a, pick);
        if (t == ORC_LO || t == ORC_DB || t == ORC_FX)
          orc_set_col_bnds(S2->prob, pick, (l == floor(bound)) ? ORC_FX : ORC_DB, l, floor(bound));
        else
          orc_set_col_bnds(S2->prob, pick, ORC_UP, 0, floor(bound));
      }
      solve(&cx, S2->prob);                                        /* bs.cpp:279 */
      S2->upperBound = orc_get_obj_val(S2->prob);
      if (params->reference_quirks) {
        orc_set_col_bnds(S3->prob, pick, ORC_LO, ceil(bound), 0); /* bs.cpp:282 */
      } else {
        int t = orc_get_col_type(a, pick);
        double u = orc_get_col_ub(a, pick);
        if (t == ORC_UP || t == ORC_DB || t == ORC_FX)
          orc_set_col_bnds(S3->prob, pick, (u == ceil(bound)) ? ORC_FX : ORC_DB, ceil(bound), u);
        else
          orc_set_col_bnds(S3->prob, pick, ORC_LO, ceil(bound), 0);
      }
      solve(&cx, S3->prob);                                        /* bs.cpp:287 */
      S3->upperBound = orc_get_obj_val(S3->prob);
      res->node_bound[S2->oid] = S2->upperBound;
      res->node_bound[S3->oid] = S3->upperBound;

      /* erase the parent (bs.cpp:247) then push the children to the back (bs.cpp:297-298) */
      node_free(node);
      memmove(&leaf[index], &leaf[index + 1], (size_t)(nq - index - 1) * sizeof(node_t *));
      nq--;
      erase = 0;
      if (nq + 2 > qcap) {
        qcap *= 2;
        leaf = (node_t **)xrealloc(leaf, (size_t)qcap * sizeof(node_t *));
      }
      leaf[nq++] = S2;
      leaf[nq++] = S3;
      emit(&cx, ORC_EV_CANDIDATE, S2->oid, S2->upperBound, 0.0, 0, 0); /* bs.cpp:300-318 */
      emit(&cx, ORC_EV_CANDIDATE, S3->oid, S3->upperBound, 0.0, 0, 0);
      if (count > params->loop_limit) { /* bs.cpp:320-323 (exit(-1) there) */
        res->hit_limit = 1;
        count++;
        break;
      }
    }
    if (erase) {
      node_free(node);
      memmove(&leaf[index], &leaf[index + 1], (size_t)(nq - index - 1) * sizeof(node_t *));
      nq--;
    }
    count++; /* bs.cpp:326 */
  }

  for (int i = 0; i < nq; i++) node_free(leaf[i]);
  free(leaf);
  free(vars);
  orc_delete_prob(a);
  pool_free(&pool);
  res->count = count;
  res->best_lower = bestLower;
  return 0;
}

void orc_bnb_free_result(orc_bnb_result *res) {
  free(res->parent);
  free(res->prune);
  free(res->node_bound);
  free(res->events);
  free(res->x);
  memset(res, 0, sizeof(*res));
}
