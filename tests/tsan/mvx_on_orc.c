/* Test infrastructure: the mvx_* names bnb.cpp's built-in engine table refers to, forwarded to the CPU oracle, so that
   the branch-and-bound driver (mvolps_amd/csrc/bnb.cpp) can be built WITHOUT the HIP engine and run under
   ThreadSanitizer (tests/test_tsan.py).  Not product code; never linked into libmvolps_amd.so. */
#include "../../include/mvx.h"
#include "../../oracle/mvolps_oracle.h"

#define O(p) ((orc_prob *)(p))
#define CO(p) ((const orc_prob *)(p))
mvx_prob *mvx_create_prob(void) { return (mvx_prob *)orc_create_prob(); }
void mvx_erase_prob(mvx_prob *P) { orc_erase_prob(O(P)); }
void mvx_delete_prob(mvx_prob *P) { orc_delete_prob(O(P)); }
void mvx_copy_prob(mvx_prob *d, const mvx_prob *s, int names) { orc_copy_prob(O(d), CO(s), names); }
int mvx_add_rows(mvx_prob *P, int nrs) { return orc_add_rows(O(P), nrs); }
void mvx_set_mat_row(mvx_prob *P, int i, int len, const int *ind, const double *val) { orc_set_mat_row(O(P), i, len, ind, val); }
void mvx_set_row_bnds(mvx_prob *P, int i, int t, double lb, double ub) { orc_set_row_bnds(O(P), i, t, lb, ub); }
void mvx_set_col_bnds(mvx_prob *P, int j, int t, double lb, double ub) { orc_set_col_bnds(O(P), j, t, lb, ub); }
int mvx_simplex(mvx_prob *P, const mvx_smcp *parm) { return orc_simplex(O(P), (const orc_smcp *)parm); }
int mvx_simplex_batch(mvx_prob **probs, int count, const mvx_smcp *parm, int *rcs) {
  return orc_simplex_batch((orc_prob **)probs, count, (const orc_smcp *)parm, rcs);
}
int mvx_get_status(const mvx_prob *P) { return orc_get_status(CO(P)); }
double mvx_get_obj_val(const mvx_prob *P) { return orc_get_obj_val(CO(P)); }
double mvx_get_obj_coef(const mvx_prob *P, int j) { return orc_get_obj_coef(CO(P), j); }
double mvx_get_col_prim(const mvx_prob *P, int j) { return orc_get_col_prim(CO(P), j); }
int mvx_get_num_rows(const mvx_prob *P) { return orc_get_num_rows(CO(P)); }
int mvx_get_num_cols(const mvx_prob *P) { return orc_get_num_cols(CO(P)); }
int mvx_get_col_kind(const mvx_prob *P, int j) { return orc_get_col_kind(CO(P), j); }
int mvx_get_col_stat(const mvx_prob *P, int j) { return orc_get_col_stat(CO(P), j); }
int mvx_get_row_stat(const mvx_prob *P, int i) { return orc_get_row_stat(CO(P), i); }
double mvx_get_row_ub(const mvx_prob *P, int i) { return orc_get_row_ub(CO(P), i); }
double mvx_get_row_lb(const mvx_prob *P, int i) { return orc_get_row_lb(CO(P), i); }
double mvx_get_col_ub(const mvx_prob *P, int j) { return orc_get_col_ub(CO(P), j); }
double mvx_get_col_lb(const mvx_prob *P, int j) { return orc_get_col_lb(CO(P), j); }
int mvx_get_col_type(const mvx_prob *P, int j) { return orc_get_col_type(CO(P), j); }
int mvx_get_mat_row(const mvx_prob *P, int i, int *ind, double *val) { return orc_get_mat_row(CO(P), i, ind, val); }
int mvx_eval_tab_row(const mvx_prob *P, int k, int *ind, double *val) { return orc_eval_tab_row(CO(P), k, ind, val); }
int mvx_get_it_cnt(const mvx_prob *P) { return orc_get_it_cnt(CO(P)); }
int mvx_get_obj_dir(const mvx_prob *P) { return orc_get_obj_dir(CO(P)); }
int mvx_gmi_cuts(const mvx_prob *P, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok) {
  (void)P; (void)repaired; (void)cols; (void)count; (void)vals; (void)rhs; (void)ok;
  return -1; /* no batch entry on this side: the harness runs without cuts */
}
int mvx_gmi_cuts_many(const mvx_prob *const *Ps, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok) {
  (void)Ps; (void)repaired; (void)cols; (void)count; (void)vals; (void)rhs; (void)ok;
  return -1;
}
/* the bulk read printInfo uses: read by several host threads at once in the window driver */
void mvx_get_col_prim_all(const mvx_prob *P, double *x) {
  const int n = orc_get_num_cols(CO(P));
  for (int j = 1; j <= n; j++) x[j] = orc_get_col_prim(CO(P), j);
}
/* model construction for the harness */
void mvx_set_obj_dir(mvx_prob *P, int dir) { orc_set_obj_dir(O(P), dir); }
int mvx_add_cols(mvx_prob *P, int ncs) { return orc_add_cols(O(P), ncs); }
void mvx_set_obj_coef(mvx_prob *P, int j, double v) { orc_set_obj_coef(O(P), j, v); }
void mvx_set_col_kind(mvx_prob *P, int j, int kind) { orc_set_col_kind(O(P), j, kind); }
