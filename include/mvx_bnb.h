/*
 * mvx_bnb.h -- C ABI of the branch-and-bound driver that sits on top of the LP engine.
 *
 * The driver is the MI355X-side counterpart of MVOLPS's own code around glp_simplex:
 *   branchAndBound      /root/reference/bs.cpp:54-348      (bs.h:7)
 *   printInfo           /root/reference/util.cpp:414-473
 *   pickNode / pickVar  /root/reference/util.cpp:154-230   (ParameterObj, util.h:61-99)
 *   getFract            /root/reference/util.cpp:11-23
 *   generateCut3        /root/reference/gmi.cpp:11-117     (gmi.h:7)
 *   CutPool             /root/reference/cut.cpp:6-46       (cut.h:15-23)
 *
 * It talks to its LP engine only through `mvx_lp_api`, a table of exactly the GLPK-shaped
 * entry points MVOLPS binds -- that table IS the drop-in boundary.
 * mvx_hip_lp_api() returns the gfx950 engine's table (the only engine this library ships;
 * passing NULL selects it).
 */
#ifndef MVX_BNB_H
#define MVX_BNB_H

#include "mvx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvx_lp_api {
  void *(*create_prob)(void);
  void (*erase_prob)(void *P);
  void (*delete_prob)(void *P);
  void (*copy_prob)(void *dst, const void *src, int names);
  int (*add_rows)(void *P, int nrs);
  void (*set_mat_row)(void *P, int i, int len, const int *ind, const double *val);
  void (*set_row_bnds)(void *P, int i, int type, double lb, double ub);
  void (*set_col_bnds)(void *P, int j, int type, double lb, double ub);
  int (*simplex)(void *P, const void *parm);
  int (*get_status)(const void *P);
  double (*get_obj_val)(const void *P);
  double (*get_obj_coef)(const void *P, int j);
  double (*get_col_prim)(const void *P, int j);
  int (*get_num_rows)(const void *P);
  int (*get_num_cols)(const void *P);
  int (*get_col_kind)(const void *P, int j);
  int (*get_col_stat)(const void *P, int j);
  int (*get_row_stat)(const void *P, int i);
  double (*get_row_ub)(const void *P, int i);
  double (*get_row_lb)(const void *P, int i);
  double (*get_col_ub)(const void *P, int j);
  double (*get_col_lb)(const void *P, int j);
  int (*get_col_type)(const void *P, int j);
  int (*get_mat_row)(const void *P, int i, int *ind, double *val);
  int (*eval_tab_row)(const void *P, int k, int *ind, double *val);
  int (*get_it_cnt)(const void *P);
  /* optional (may be NULL): solve `count` independent handles concurrently, same results as
     `count` simplex calls -- used for the two children of a branch (bs.cpp:279,287) */
  int (*simplex_batch)(void **probs, int count, const void *parm, int *rcs);
  /* optional (may be NULL = maximisation): GLP_MIN / GLP_MAX.  bs.cpp compares bounds as a maximiser
     (bs.cpp:172,210) whatever the direction; with reference_quirks = 0 the driver turns the compares
     round for a minimisation problem */
  int (*get_obj_dir)(const void *P);
  /* optional (may be NULL): generateCut3 / the repaired formula for `count` basic integer columns of a solved node in
     one call (mvx_gmi_cuts: tableau rows, coefficient formula and back-substitution on the device); the driver then
     generates a node's cuts through it instead of one eval_tab_row + m get_mat_row calls per cut */
  int (*gmi_cuts)(const void *P, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok);
  /* optional (may be NULL): one cut from each of `count` different solved handles (mvx_gmi_cuts_many): the window driver
     generates the cuts of a whole round through it */
  int (*gmi_cuts_many)(const void *const *Ps, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok);
  /* optional (may be NULL): get_col_prim for every column at once, x[1..n] -- printInfo (util.cpp:414-473) reads all n
     values of every node */
  void (*get_col_prim_all)(const void *P, double *x);
} mvx_lp_api;

const mvx_lp_api *mvx_hip_lp_api(void);

/* ParameterObj (util.h:61-99); defaults VO / DFS(=FIFO) / no cuts (util.h:65-67) */
typedef struct {
  int var_strat;        /* 0 VO, 1 VFP, 2 VGO   (util.h:30) */
  int node_strat;       /* 0 DFS (problems.front(), util.cpp:165), 1 BEST (util.cpp:170-186) */
  int cut_strat;        /* 0 NONE, 1 GMI         (util.h:32) */
  double cut_chance;    /* -cf: stored, never read (util.cpp:259-261) */
  int loop_limit;       /* bs.cpp:320: 200000 branchings */
  int max_nodes;        /* stop after this many loop iterations (<= 0: none) */
  int reference_quirks; /* 1 (default): bug-compatible with bs.cpp / util.cpp;
                           0: children keep the opposite bound (bs.cpp:274,282 drop it), the
                           integrality test has a 1e-9 tolerance, cuts are the repaired GMI of
                           mvx_generateCutGMI and are not carried from node to node, and a
                           minimisation problem is bounded and pruned as one (best_lower is then the
                           best upper bound) */
  int lazy_pool;        /* 1 (default): generate only the cut cut.cpp:20 will actually add (the last
                           eligible column's; with reference_quirks = 0 and cut_select = 0 the last column
                           that yields a cut); 0: generate every cut like bs.cpp:250-255 */
  int cut_select;       /* reference_quirks = 0 only (changes results, hence not the
                           default path).  0: add the last generated cut (cut.cpp:20); 1: add the
                           ceil(cut_chance * k) most effective of the node's k cuts (-cf honoured) */
  int window;           /* FIFO order, engine with a batch entry: solve the front `window` nodes of the deque
                           together and replay bs.cpp's decisions (cuts included: the replay is in queue
                           order, so the persistent pool sees the nodes as bs.cpp does) -- same tree, oids,
                           events and incumbent as node-at-a-time; default 64, 1 = node
                           at a time */
} mvx_bnb_params;

/* B&B events at the emit points of bs.cpp (message.h EventType) */
#define MVX_EV_PREGNANT 0   /* bs.cpp:119-129 */
#define MVX_EV_INTEGER 1    /* bs.cpp:163-166 */
#define MVX_EV_INFEASIBLE 2 /* bs.cpp:199-203 */
#define MVX_EV_FATHOMED 3   /* bs.cpp:215-217 */
#define MVX_EV_BRANCHED 4   /* bs.cpp:225-244 */
#define MVX_EV_CANDIDATE 5  /* bs.cpp:300-318 */

typedef struct {
  int type, oid, pid, direction; /* direction 0 M, 1 R, 2 L (bs.cpp:43-52) */
  double lp_bound;               /* field6 */
  double sum_infeas;             /* field7 (bs.cpp:227-241) */
  int n_violated;                /* field8 */
  int pick;                      /* branching variable of a branched event, else 0 */
} mvx_bnb_event;

typedef struct {
  int n_nodes;        /* oids are 1..n_nodes (util.h:17, util.cpp:29-30) */
  int *parent;        /* parent[oid]; 0 for the root (bs.cpp:26-33) */
  int *prune;         /* prune[oid]: 0 INTG, 1 FEAS, 3 BNDS, 4 NONE (util.h:27) */
  double *node_bound; /* NodeData::upperBound */
  int n_events;
  mvx_bnb_event *events;
  int count;          /* loop iterations (bs.cpp:326) */
  int has_incumbent;
  double best_lower;  /* bs.cpp:90,172-174 */
  int incumbent_oid;
  int n;
  double *x;          /* x[1..n] of the incumbent (bs.cpp:181-187) */
  long long total_pivots;
  int hit_limit;
} mvx_bnb_result;

void mvx_bnb_default_params(mvx_bnb_params *p);
/* int branchAndBound(glp_prob*, MVOLP::ParameterObj&)  bs.h:7 */
int mvx_branchAndBound(const mvx_lp_api *api, void *prob, const mvx_bnb_params *params, mvx_bnb_result *res);
void mvx_bnb_free_result(mvx_bnb_result *res);

double mvx_getFract(double x); /* util.cpp:11-23 */
/* std::pair<int, std::vector<int>> printInfo(glp_prob*, bool)  util.cpp:414 */
int mvx_printInfo(const mvx_lp_api *api, const void *prob, int quirks, int *violated, int *nviolated);
/* CutContainer generateCut3(glp_prob*, int j)  gmi.h:7; inds/vals hold n+1 entries, returns -1 when rejected */
int mvx_generateCut3(const mvx_lp_api *api, const void *prob, int j, int *inds, double *vals, double *lb);

/* Repaired Gomory mixed-integer cut (used when reference_quirks = 0): non-basic variables measured
   from the bound they sit at, f0 from the row's own value, back-substitution by column index.
   Same output layout as mvx_generateCut3; *efficacy = violation / 2-norm at the current vertex */
int mvx_generateCutGMI(const mvx_lp_api *api, const void *prob, int j, int *inds, double *vals, double *lb, double *efficacy);

/* ---- node-level helpers for window drivers (mvolps_amd/dist_bnb.py): one call per node instead of
   one per query.  Same arithmetic, same order as the loop body of bs.cpp. ---- */
/* classification of a solved node (bs.cpp:135-156,227-241,260): out[0] status -1/0/1 (printInfo),
   out[1] objective, out[2] number of violated columns, out[3] sum of their fractional parts,
   out[4] pickVar's choice (0 when none), `root` = ParameterObj::_prob */
int mvx_bnb_classify(const mvx_lp_api *api, const void *prob, const void *root, int quirks, int var_strat, double *out);
/* bs.cpp:261-282: bound = col_prim(a, pick); S2/S3 = clones of `a` (created by the caller with
   create_prob) with the branching bounds set; they are NOT solved here (the caller batches them) */
int mvx_bnb_make_children(const mvx_lp_api *api, const void *a, int pick, int quirks, void *S2, void *S3);

/* bs.cpp:249-258 on one solved node `a` that is about to be branched: generate its GMI cut(s) and append the
   row(s) (cut_strat / reference_quirks / lazy_pool / cut_select / cut_chance of `params`).  Returns the number
   of rows appended.  The pool here holds this node's cuts only; -1 = bug-compatible mode and the node generated
   no cut, where bs.cpp would re-add the last cut pooled by an earlier node (cut.cpp:16-21) --
   a branched node has a fractional, hence basic, integer column and therefore always generates one */
int mvx_bnb_node_cuts(const mvx_lp_api *api, void *a, const mvx_bnb_params *params);

/* ---- callers and data formats either side of the path ---- */
/* glp_read_lp(prob, NULL, fname) util.cpp:284 -- CPLEX LP format; 0 on success */
int mvx_read_lp(mvx_prob *P, const void *parm, const char *fname);
/* glp_read_mps(prob, GLP_MPS_FILE, NULL, fname) util.cpp:290 -- free/fixed MPS; 0 on success */
int mvx_read_mps(mvx_prob *P, int fmt, const void *parm, const char *fname);
/* the B&B event stream of IPCDispatch::write (message.cpp:32-191), one text line per event in the
   format of message.h:141-226, to `path` (NULL = stdout) instead of a ZeroMQ socket; the first field
   (timeSpan) is the event's sequence number so that two runs can be diffed */
int mvx_bnb_write_events(const mvx_bnb_result *res, const char *path);
/* the tree report of bs.cpp:329-343 (PrettyPrintTree, tree_print.h:12-22) */
int mvx_bnb_print_tree(const mvx_bnb_result *res, const char *path);
/* the solution line of bs.cpp:176-191; returns 0, or the needed capacity when `cap` is too small */
int mvx_bnb_solution_string(const mvx_lp_api *api, const void *root, const mvx_bnb_result *res, char *buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
