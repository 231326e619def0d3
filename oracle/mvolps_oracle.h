/*
 * mvolps_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the LP-relaxation path that MVOLPS delegates to GLPK
 * (/root/reference/bs.cpp:114-117,279,287; BranchAndBound.cpp:52,134,141) plus the
 * MVOLPS-owned decisions around it (util.cpp:11-23,154-230,414-473; gmi.cpp:11-117;
 * cut.cpp:6-46; bs.cpp:54-348).
 *
 * PARITY STATUS: the arithmetic behind glp_simplex is GLPK's (linked as -lglpk,
 * /root/reference/Makefile:2, un-vendored and un-pinned; GLPK 4.65 was current at
 * the reference's last commit).  GLPK is absent from this image and the reference has
 * no tests, fixtures or golden vectors, so this oracle is pinned
 * against an INDEPENDENT solver instead: scipy 1.15.3 HiGHS, through the committed
 * fixtures under tests/golden/ (generator: tests/golden/make_golden.py).  It is pinned
 * at the API boundary (status / objective / primal values), NOT on GLPK's pivot
 * sequence -> "parity unpinned" with respect to GLPK internals.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * The algorithm is a dense condensed-tableau bounded-variable simplex:
 *   T is (m+1) x (n+1), row-major, leading dimension ld.
 *   T[0][0] = objective value          T[0][j] = reduced cost of non-basic column j
 *   T[i][0] = value of basic var i     T[i][j] = tableau entry  (x_B = T x_N)
 * Variables are numbered GLPK-style: 1..m auxiliary (rows), m+1..m+n structural.
 * Every arithmetic step here is mirrored operation-for-operation by the HIP engine
 * (mvolps_amd/csrc) so that results are BIT-EXACT between the two.
 */
#ifndef MVOLPS_ORACLE_H
#define MVOLPS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* GLPK public enum values [GLPK-recalled, SURVEY.md section 8(b)] */
#define ORC_MIN 1
#define ORC_MAX 2
#define ORC_CV 1
#define ORC_IV 2
#define ORC_BV 3
#define ORC_FR 1
#define ORC_LO 2
#define ORC_UP 3
#define ORC_DB 4
#define ORC_FX 5
#define ORC_BS 1
#define ORC_NL 2
#define ORC_NU 3
#define ORC_NF 4
#define ORC_NS 5
#define ORC_UNDEF 1
#define ORC_FEAS 2
#define ORC_INFEAS 3
#define ORC_NOFEAS 4
#define ORC_OPT 5
#define ORC_UNBND 6
#define ORC_OFF 0
#define ORC_ON 1
#define ORC_EFAIL 5
#define ORC_EITLIM 8

typedef struct orc_prob orc_prob;

typedef struct {
  int msg_lev;
  int meth;       /* 1 = automatic (primal / dual / phase 1 chosen from the basis); nothing else is implemented */
  int it_lim;     /* pivot limit for THIS call; <0 = none */
  double tol_bnd; /* primal feasibility tolerance (relative: tol*(1+|bound|)) */
  double tol_dj;  /* dual feasibility tolerance (absolute) */
  double tol_piv; /* pivot magnitude tolerance (absolute) */
} orc_smcp;

/* lifecycle (bs.cpp:89,114-116; util.cpp:33-34,41) */
orc_prob *orc_create_prob(void);
void orc_erase_prob(orc_prob *P);
void orc_delete_prob(orc_prob *P);
void orc_copy_prob(orc_prob *dst, const orc_prob *src, int names);

/* build / modify (cut.cpp:23,40,43; bs.cpp:274,282; util.cpp:55,58) */
void orc_set_obj_dir(orc_prob *P, int dir);
int orc_add_rows(orc_prob *P, int nrs);
int orc_add_cols(orc_prob *P, int ncs);
void orc_set_row_bnds(orc_prob *P, int i, int type, double lb, double ub);
void orc_set_col_bnds(orc_prob *P, int j, int type, double lb, double ub);
void orc_set_obj_coef(orc_prob *P, int j, double coef);
void orc_set_mat_row(orc_prob *P, int i, int len, const int *ind, const double *val);
void orc_set_col_kind(orc_prob *P, int j, int kind);
void orc_set_col_name(orc_prob *P, int j, const char *name);
int orc_load_dense(orc_prob *P, int m, int n, const double *A, const double *b, const double *c);

/* solve (bs.cpp:117,279,287) */
void orc_init_smcp(orc_smcp *parm);
void orc_set_default_tolerances(double tol_bnd, double tol_dj, double tol_piv);
/* tableau refresh (see mvolps_oracle.c): residual look every `check_every` pivots of a lineage, rebuild above `tol` */
void orc_set_refresh(int check_every, double tol);
int orc_get_refresh_cnt(const orc_prob *P);
double orc_row_residual(const orc_prob *P);
int orc_simplex(orc_prob *P, const orc_smcp *parm);
/* same contract as mvx_simplex_batch; here simply one after the other */
int orc_simplex_batch(orc_prob **probs, int count, const orc_smcp *parm, int *rcs);

/* query */
int orc_get_obj_dir(const orc_prob *P);
int orc_get_num_rows(const orc_prob *P);
int orc_get_num_cols(const orc_prob *P);
int orc_get_num_int(const orc_prob *P);
int orc_get_status(const orc_prob *P);
double orc_get_obj_val(const orc_prob *P);
double orc_get_obj_coef(const orc_prob *P, int j);
double orc_get_col_prim(const orc_prob *P, int j);
double orc_get_row_prim(const orc_prob *P, int i);
double orc_get_col_dual(const orc_prob *P, int j);
double orc_get_row_dual(const orc_prob *P, int i);
int orc_get_col_stat(const orc_prob *P, int j);
int orc_get_row_stat(const orc_prob *P, int i);
int orc_get_col_kind(const orc_prob *P, int j);
int orc_get_row_type(const orc_prob *P, int i);
double orc_get_row_lb(const orc_prob *P, int i);
double orc_get_row_ub(const orc_prob *P, int i);
int orc_get_col_type(const orc_prob *P, int j);
double orc_get_col_lb(const orc_prob *P, int j);
double orc_get_col_ub(const orc_prob *P, int j);
const char *orc_get_col_name(const orc_prob *P, int j);
int orc_get_mat_row(const orc_prob *P, int i, int *ind, double *val);
int orc_eval_tab_row(const orc_prob *P, int k, int *ind, double *val);
int orc_get_it_cnt(const orc_prob *P);
int orc_get_bland_cnt(const orc_prob *P); /* pivots chosen under the anti-cycling (Bland) rule */
void orc_set_stall_limit(int limit); /* > 0 overrides the 64 + (m+n)/8 degenerate pivots that arm the rules; 0 = default */
int orc_get_pert_cnt(const orc_prob *P);  /* bound perturbations applied against stalling */
int orc_term_out(int flag);
const char *orc_version(void);

/* parity hooks: raw engine state */
int orc_get_tableau_ld(const orc_prob *P);
/* copies (m+1) x (n+1) entries, packed row-major, into out */
int orc_get_tableau(const orc_prob *P, double *out);
/* head[0..m] (head[0] unused) basic variable per row; nb[0..n] non-basic variable per column;
   flag[0..n] non-basic status (ORC_NL/NU/NF/NS) */
int orc_get_basis(const orc_prob *P, int *head, int *nb, int *flag);

/* node migration between ranks: serialise bounds + basis + tableau of P (same model rows as
   the receiver's base problem; appended cut rows are not carried).  Host memory. */
long long orc_pack_size(const orc_prob *P);
int orc_pack(const orc_prob *P, void *buf);
long long orc_pack_size_from(const orc_prob *P, const orc_prob *base);
int orc_pack_from(const orc_prob *P, const orc_prob *base, void *buf);
int orc_unpack(orc_prob *dst, const orc_prob *base, const void *buf);

/* ---- MVOLPS-owned pieces restated on top of the API above ---- */
double orc_getFract(double x); /* util.cpp:11-23 */

/* printInfo (util.cpp:414-473): returns status -1/0/1, fills violated[] (1-based col
   indices, ascending), *nviolated. */
int orc_printInfo(const orc_prob *P, int *violated, int *nviolated);
/* quirks != 0: exactly the reference rule.  quirks == 0: repaired rule (integrality within
   1e-9, no dependence on the objective coefficient). */
int orc_printInfo_ex(const orc_prob *P, int quirks, int *violated, int *nviolated);

/* generateCut3 (gmi.cpp:11-117).  inds/vals have n+1 entries (element 0: ind 0, val rhs).
   Returns -1 when rejected (non-integer or non-basic column), 0 on success. */
int orc_generateCut3(const orc_prob *P, int j, int *inds, double *vals, double *lb);
/* Repaired Gomory mixed-integer cut for basic integer column j (reference_quirks = 0): non-basic
   variables measured from the bound they sit at, f0 from the row's own value, back-substitution by
   column index.  Same output layout as orc_generateCut3; also returns the cut's efficacy
   (violation / 2-norm) at the current vertex.  -1 when no cut can be derived. */
int orc_generateCutGMI(const orc_prob *P, int j, int *inds, double *vals, double *lb, double *efficacy);

typedef struct {
  int var_strat;  /* 0 VO, 1 VFP, 2 VGO   (util.h:30) */
  int node_strat; /* 0 DFS (= FIFO, util.cpp:165), 1 BEST (util.cpp:170-186) */
  int cut_strat;  /* 0 NONE, 1 GMI         (util.h:32) */
  double cut_chance; /* stored, never read (util.cpp:259-261) */
  int loop_limit; /* bs.cpp:320: 200000 */
  int max_nodes;  /* safety cap for tests: stop after this many loop iterations (<=0: none) */
  int reference_quirks; /* 1 (default): bug-compatible with bs.cpp/util.cpp;
                           0: child bounds keep the opposite bound (bs.cpp:274,282 drop it),
                           printInfo uses the repaired integrality rule, cuts use orc_generateCutGMI
                           and are not carried from node to node */
  int cut_select;       /* reference_quirks = 0 only.  0: add the last generated cut (cut.cpp:20);
                           1: add the ceil(cut_chance * k) most effective of the node's k cuts */
} orc_bnb_params;

/* event types follow message.h EventType order used at the bs.cpp emit points */
#define ORC_EV_PREGNANT 0
#define ORC_EV_INTEGER 1
#define ORC_EV_INFEASIBLE 2
#define ORC_EV_FATHOMED 3
#define ORC_EV_BRANCHED 4
#define ORC_EV_CANDIDATE 5

typedef struct {
  int type, oid, pid, direction; /* direction: 0 M, 1 R, 2 L (bs.cpp:43-52) */
  double lp_bound;               /* field6 */
  double sum_infeas;             /* field7 */
  int n_violated;                /* field8 */
  int pick;                      /* branching variable (branched events), else 0 */
} orc_bnb_event;

typedef struct {
  int n_nodes;        /* number of oids created (oids are 1..n_nodes) */
  int *parent;        /* parent[oid], 0 for the root            (bs.cpp:26-33) */
  int *prune;         /* prune[oid]: 0 INTG, 1 FEAS, 3 BNDS, 4 NONE (util.h:27) */
  double *node_bound; /* NodeData::upperBound per oid */
  int n_events;
  orc_bnb_event *events;
  int count;          /* loop iterations (bs.cpp:326) */
  int has_incumbent;
  double best_lower;  /* bs.cpp:90,172-174 */
  int incumbent_oid;
  int n;              /* columns */
  double *x;          /* x[1..n] of the incumbent */
  long long total_pivots;
  int hit_limit;
} orc_bnb_result;

void orc_bnb_default_params(orc_bnb_params *p);
int orc_branchAndBound(orc_prob *P, const orc_bnb_params *params, orc_bnb_result *res);
void orc_bnb_free_result(orc_bnb_result *res);

#ifdef __cplusplus
}
#endif
#endif
