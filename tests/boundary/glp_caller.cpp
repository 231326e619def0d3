// glp_caller.cpp -- a caller written in GLPK spelling against include/glpk_on_mvx.h (test infrastructure).
//
// It is NOT one of the reference's files: it is a small program of this repo that makes every call of SURVEY.md
// section 8(b) -- the calls MVOLPS makes on its LP engine -- in the order and shape MVOLPS makes them (node solve,
// clone, branch bounds, child solves, tableau row, cut row append, the queries of printInfo / generateCut3), so
// that the binding header is compiled and linked, not only described.
//
//   glp_caller model            no engine call: builds a model through the glp_* edit calls and reads it back
//   glp_caller solve FILE       reads FILE (.lp / .mps), then one branching step the way bs.cpp:114-288 does it
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "glpk_on_mvx.h"

static int fail(const char *what) {
  std::printf("FAIL %s\n", what);
  return 1;
}

// F1 through the edit calls
static glp_prob *build_f1() {
  static const double A[3][5] = {{2, 3, 1, 4, 2}, {4, 1, 2, 3, 5}, {3, 4, 2, 1, 3}};
  static const double b[3] = {15, 23, 17}, c[5] = {5, 4, 3, 7, 6};
  glp_prob *P = glp_create_prob();
  glp_set_obj_dir(P, GLP_MAX);
  glp_add_cols(P, 5);
  glp_add_rows(P, 3);
  for (int j = 1; j <= 5; j++) {
    glp_set_obj_coef(P, j, c[j - 1]);
    glp_set_col_bnds(P, j, GLP_LO, 0.0, 0.0);
    glp_set_col_kind(P, j, GLP_IV);
  }
  int ind[6] = {0, 1, 2, 3, 4, 5};
  for (int i = 1; i <= 3; i++) {
    double val[6] = {0, A[i - 1][0], A[i - 1][1], A[i - 1][2], A[i - 1][3], A[i - 1][4]};
    glp_set_mat_row(P, i, 5, ind, val);
    glp_set_row_bnds(P, i, GLP_UP, 0.0, b[i - 1]);
  }
  return P;
}

static int mode_model() {
  std::printf("version %s\n", glp_version());
  glp_term_out(GLP_OFF);
  glp_prob *P = build_f1();
  if (glp_get_num_rows(P) != 3 || glp_get_num_cols(P) != 5 || glp_get_num_int(P) != 5) return fail("sizes");
  if (glp_get_obj_dir(P) != GLP_MAX || glp_get_obj_coef(P, 4) != 7.0 || glp_get_obj_coef(P, 0) != 0.0) return fail("objective");
  if (glp_get_col_kind(P, 2) != GLP_IV || glp_get_col_type(P, 2) != GLP_LO || glp_get_col_lb(P, 2) != 0.0) return fail("columns");
  if (glp_get_col_ub(P, 2) != DBL_MAX) return fail("absent bound sentinel"); // consumed arithmetically at gmi.cpp:73
  if (glp_get_row_type(P, 2) != GLP_UP || glp_get_row_ub(P, 2) != 23.0 || glp_get_row_lb(P, 2) != -DBL_MAX) return fail("rows");
  int ind[6];
  double val[6];
  if (glp_get_mat_row(P, 3, ind, val) != 5 || ind[4] != 4 || val[4] != 1.0) return fail("get_mat_row");
  if (glp_get_status(P) != GLP_UNDEF) return fail("status before a solve");
  if (glp_get_col_stat(P, 1) != GLP_NL || glp_get_row_stat(P, 1) != GLP_BS) return fail("standard basis statuses");
  // clone + branching-style bound edits (bs.cpp:116,274,282) on the model alone
  glp_prob *Q = glp_create_prob();
  glp_copy_prob(Q, P, GLP_ON);
  glp_set_col_bnds(Q, 3, GLP_UP, 0, 5.0);
  if (glp_get_col_type(Q, 3) != GLP_UP || glp_get_col_ub(Q, 3) != 5.0 || glp_get_col_lb(Q, 3) != -DBL_MAX) return fail("GLP_UP drops the lower bound");
  if (glp_get_col_type(P, 3) != GLP_LO) return fail("deep copy");
  // a cut-shaped row append (cut.cpp:23,40,43)
  int r = glp_add_rows(Q, 1);
  int cind[6] = {0, 1, 2, 3, 4, 5};
  double cval[6] = {0, 1, 1, 1, 1, 1};
  glp_set_mat_row(Q, r, 5, cind, cval);
  glp_set_row_bnds(Q, r, GLP_LO, 2.0, 0);
  if (r != 4 || glp_get_num_rows(Q) != 4 || glp_get_row_type(Q, 4) != GLP_LO || glp_get_row_lb(Q, 4) != 2.0) return fail("row append");
  glp_erase_prob(Q);
  if (glp_get_num_rows(Q) != 0 || glp_get_num_cols(Q) != 0) return fail("erase");
  glp_delete_prob(Q);
  glp_delete_prob(P);
  glp_smcp parm;
  glp_init_smcp(&parm);
  std::printf("smcp tol_bnd %.1e tol_dj %.1e tol_piv %.1e it_lim %d\n", parm.tol_bnd, parm.tol_dj, parm.tol_piv, parm.it_lim);
  std::printf("MODEL OK\n");
  return 0;
}

// printInfo's loop (util.cpp:436-451): first column that reads as fractional, 0 when none
static int first_fractional(glp_prob *a) {
  for (int i = 1; i <= glp_get_num_cols(a); i++) {
    const double v = glp_get_col_prim(a, i);
    if (v != 0 && glp_get_obj_coef(a, i) != 0 && std::trunc(v) != v && glp_get_col_kind(a, i) != GLP_CV) return i;
  }
  return 0;
}

static int mode_solve(const char *file) {
  glp_term_out(GLP_OFF);
  glp_prob *prob = glp_create_prob();
  const std::string f(file);
  int rc;
  if (f.size() > 4 && f.substr(f.size() - 4) == ".mps") rc = glp_read_mps(prob, GLP_MPS_FILE, NULL, file); // util.cpp:290
  else rc = glp_read_lp(prob, NULL, file);                                                                   // util.cpp:284
  if (rc != 0) return fail("reader");
  std::printf("read %d rows %d cols %d int dir %d\n", glp_get_num_rows(prob), glp_get_num_cols(prob), glp_get_num_int(prob), glp_get_obj_dir(prob));
  // node solve on a scratch copy (bs.cpp:114-117)
  glp_prob *a = glp_create_prob();
  glp_erase_prob(a);
  glp_copy_prob(a, prob, GLP_OFF);
  glp_simplex(a, NULL);
  if (glp_get_status(a) != GLP_OPT) return fail("root status");
  std::printf("root status %d obj %.15g\n", glp_get_status(a), glp_get_obj_val(a));
  const int m = glp_get_num_rows(a), n = glp_get_num_cols(a);
  const int pick = first_fractional(a);
  if (!pick) return fail("no fractional column");
  const double bound = glp_get_col_prim(a, pick); // bs.cpp:261
  std::printf("pick %d value %.15g stat %d\n", pick, bound, glp_get_col_stat(a, pick));
  // tableau row of the branching column + the bound queries of generateCut3 (gmi.cpp:36-53)
  std::vector<int> ind((size_t)n + 1);
  std::vector<double> val((size_t)n + 1);
  const int len = glp_eval_tab_row(a, m + pick, ind.data(), val.data());
  double acc = 0.0;
  for (int t = 1; t <= len; t++) {
    const int k = ind[(size_t)t];
    const double ub = k <= m ? glp_get_row_ub(a, k) : glp_get_col_ub(a, k - m);
    const int st = k <= m ? glp_get_row_stat(a, k) : glp_get_col_stat(a, k - m);
    acc += val[(size_t)t] * (st == GLP_NU ? ub : 0.0);
  }
  std::printf("tab row len %d\n", len);
  // a cut-shaped row on the solved node (cut.cpp:23-43): sum x >= 0, never binding
  const int r = glp_add_rows(a, 1);
  std::vector<int> cind((size_t)n + 1);
  std::vector<double> cval((size_t)n + 1, 1.0);
  for (int j = 0; j <= n; j++) cind[(size_t)j] = j;
  glp_set_mat_row(a, r, n, cind.data(), cval.data());
  glp_set_row_bnds(a, r, GLP_LO, 0.0, 0);
  // the two children (bs.cpp:269-288)
  glp_prob *S2 = glp_create_prob(), *S3 = glp_create_prob();
  glp_copy_prob(S2, a, GLP_ON);
  glp_copy_prob(S3, a, GLP_ON);
  glp_set_col_bnds(S2, pick, GLP_UP, 0, std::floor(bound));
  glp_simplex(S2, NULL);
  glp_set_col_bnds(S3, pick, GLP_LO, std::ceil(bound), 0);
  glp_simplex(S3, NULL);
  std::printf("child down status %d obj %.15g\n", glp_get_status(S2), glp_get_obj_val(S2));
  std::printf("child up status %d obj %.15g\n", glp_get_status(S3), glp_get_obj_val(S3));
  if (glp_get_status(S2) == GLP_OPT && !(glp_get_obj_val(S2) <= glp_get_obj_val(a) + 1e-9)) return fail("child bound above parent");
  // the solution line of bs.cpp:180-190
  std::string sol;
  for (int i = 1; i <= glp_get_num_cols(S2); i++)
    if (glp_get_col_prim(S2, i) != 0 && glp_get_obj_coef(S2, i) != 0) {
      char buf[96];
      std::snprintf(buf, sizeof buf, "%g*(x[%d] = %g) + ", glp_get_obj_coef(S2, i), i, glp_get_col_prim(S2, i));
      sol += buf;
    }
  std::printf("%s%g = %.15g\n", sol.c_str(), glp_get_obj_coef(S2, 0), glp_get_obj_val(S2));
  glp_delete_prob(S2);
  glp_delete_prob(S3);
  glp_delete_prob(a);
  glp_delete_prob(prob);
  (void)acc;
  std::printf("SOLVE OK\n");
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 2 && !std::strcmp(argv[1], "model")) return mode_model();
  if (argc >= 3 && !std::strcmp(argv[1], "solve")) return mode_solve(argv[2]);
  std::printf("usage: glp_caller model | solve FILE\n");
  return 2;
}
