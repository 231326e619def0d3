/*
 * mvx.h -- C ABI of the MI355X dense-simplex LP-relaxation engine (libmvolps_amd.so).
 *
 * Drop-in boundary: MVOLPS talks to its LP engine only through the GLPK C API
 * (SURVEY.md section 8(b)).  Every entry point below replaces the glp_* call named next
 * to it (file:line under /root/reference); semantics, 1-based indexing, "element 0
 * ignored" array convention, ownership (opaque handle created/destroyed by the caller,
 * deep copy) and error behaviour (int return codes, never throws; invalid arguments
 * abort, as GLPK does) follow GLPK's.  Enum VALUES are GLPK's public ones so that a
 * caller's GLP_* constants can be passed through unchanged.
 *
 * Plain C: pointers, ints and doubles only; no C++ or torch types cross this boundary.
 * The engine needs a gfx950 device: without one every entry point that touches the
 * engine (mvx_simplex and anything after it) fails loudly (message on stderr + abort).
 * There is no CPU fallback in this library.
 */
#ifndef MVX_H
#define MVX_H

#ifdef __cplusplus
extern "C" {
#endif

/* optimisation direction (glp_set_obj_dir / glp_get_obj_dir) */
#define MVX_MIN 1
#define MVX_MAX 2
/* column kind (glp_get_col_kind) */
#define MVX_CV 1
#define MVX_IV 2
#define MVX_BV 3
/* bound type (glp_set_row_bnds / glp_set_col_bnds) */
#define MVX_FR 1
#define MVX_LO 2
#define MVX_UP 3
#define MVX_DB 4
#define MVX_FX 5
/* basis status (glp_get_row_stat / glp_get_col_stat) */
#define MVX_BS 1
#define MVX_NL 2
#define MVX_NU 3
#define MVX_NF 4
#define MVX_NS 5
/* solution status (glp_get_status) */
#define MVX_UNDEF 1
#define MVX_FEAS 2
#define MVX_INFEAS 3
#define MVX_NOFEAS 4
#define MVX_OPT 5
#define MVX_UNBND 6
#define MVX_OFF 0
#define MVX_ON 1
/* mvx_simplex return codes (glp_simplex) */
#define MVX_EFAIL 5
#define MVX_EITLIM 8
/* mvx_last_error codes */
#define MVX_ENOMEM 0x101 /* the device ran out of memory for a tableau slab */

typedef struct mvx_prob mvx_prob; /* replaces glp_prob */

/* replaces glp_smcp (only the controls this engine honours) */
typedef struct {
  int msg_lev;
  int meth;       /* 1 = automatic (primal / dual / phase 1 from the basis at hand) */
  int it_lim;     /* pivot limit for this call, < 0 = none */
  double tol_bnd; /* primal feasibility tolerance, relative: tol * (1 + |bound|) */
  double tol_dj;  /* dual feasibility tolerance, absolute */
  double tol_piv; /* pivot magnitude tolerance, absolute */
} mvx_smcp;

/* ---- lifecycle ---------------------------------------------------------------- */
mvx_prob *mvx_create_prob(void);   /* glp_create_prob  bs.cpp:89,115; util.cpp:33,281 */
void mvx_erase_prob(mvx_prob *P);  /* glp_erase_prob   bs.cpp:114 */
void mvx_delete_prob(mvx_prob *P); /* glp_delete_prob  util.cpp:41 */
/* glp_copy_prob bs.cpp:116; util.cpp:34 -- deep copy incl. basis, tableau and solution;
   the device tableau is cloned device-to-device */
void mvx_copy_prob(mvx_prob *dst, const mvx_prob *src, int names);

/* ---- build / modify ----------------------------------------------------------- */
void mvx_set_obj_dir(mvx_prob *P, int dir);                                  /* util.cpp:58 */
int mvx_add_rows(mvx_prob *P, int nrs);                                      /* cut.cpp:23 */
int mvx_add_cols(mvx_prob *P, int ncs);                                      /* (readers) */
void mvx_set_row_bnds(mvx_prob *P, int i, int type, double lb, double ub);   /* cut.cpp:43 */
void mvx_set_col_bnds(mvx_prob *P, int j, int type, double lb, double ub);   /* bs.cpp:274,282 */
void mvx_set_obj_coef(mvx_prob *P, int j, double coef);                      /* util.cpp:55 */
void mvx_set_mat_row(mvx_prob *P, int i, int len, const int *ind, const double *val); /* cut.cpp:40 */
void mvx_set_col_kind(mvx_prob *P, int j, int kind);                         /* (readers) */
void mvx_set_col_name(mvx_prob *P, int j, const char *name);                 /* (readers) */
/* dense fast path: max c'x, Ax <= b, x >= 0; A row-major m x n */
int mvx_load_dense(mvx_prob *P, int m, int n, const double *A, const double *b, const double *c);

/* ---- solve -------------------------------------------------------------------- */
void mvx_init_smcp(mvx_smcp *parm);                /* glp_init_smcp */
/* Tolerances used when `parm` is NULL (every glp_simplex call of MVOLPS passes NULL: bs.cpp:117,279,287) and filled in
   by mvx_init_smcp.  Engine defaults: tol_bnd = tol_dj = tol_piv = 1e-9.  GLPK's defaults are tol_bnd = tol_dj = 1e-7,
   tol_piv = 1e-9 [GLPK-recalled]; they are tighter here so that objectives meet 1e-9 relative against independent
   solvers.  Where it matters to MVOLPS: util.cpp:443 tests integrality with no tolerance, so a primal value the
   tolerance lets stop short of its bound by 1e-8 reads as fractional -- the tighter value makes that rarer, not
   different in kind.  A caller that wants GLPK's numbers calls mvx_set_default_tolerances(1e-7, 1e-7, 1e-9) once. */
void mvx_set_default_tolerances(double tol_bnd, double tol_dj, double tol_piv);
int mvx_simplex(mvx_prob *P, const mvx_smcp *parm); /* glp_simplex bs.cpp:117,279,287;
                                                       BranchAndBound.cpp:52,134,141 */

/* batch entry (SURVEY.md section 8(b) "a batch entry for config 5"): `count` independent handles --
   e.g. the two children of a branch (bs.cpp:279,287) or a window of open nodes -- solved
   concurrently, one HIP stream each; results are identical to `count` calls of mvx_simplex.
   rcs[i] (nullable) receives each handle's return code */
int mvx_simplex_batch(mvx_prob **probs, int count, const mvx_smcp *parm, int *rcs);

/* ---- query -------------------------------------------------------------------- */
int mvx_get_obj_dir(const mvx_prob *P);               /* util.cpp:51 */
int mvx_get_num_rows(const mvx_prob *P);              /* gmi.cpp:15 */
int mvx_get_num_cols(const mvx_prob *P);              /* gmi.cpp:16; bs.cpp:181,250 */
int mvx_get_num_int(const mvx_prob *P);               /* util.cpp:299 */
int mvx_get_status(const mvx_prob *P);                /* util.cpp:423 */
double mvx_get_obj_val(const mvx_prob *P);            /* bs.cpp:125,156,190,210,280,288 */
double mvx_get_obj_coef(const mvx_prob *P, int j);    /* bs.cpp:182,190; util.cpp:437,455 */
double mvx_get_col_prim(const mvx_prob *P, int j);    /* bs.cpp:182,232,261; gmi.cpp:37 */
/* extension: glp_get_col_prim for every column in one call, x[1..n] (util.cpp:414-473 reads them all for every node) */
void mvx_get_col_prim_all(const mvx_prob *P, double *x);
double mvx_get_row_prim(const mvx_prob *P, int i);
double mvx_get_col_dual(const mvx_prob *P, int j);
double mvx_get_row_dual(const mvx_prob *P, int i);
int mvx_get_col_stat(const mvx_prob *P, int j);       /* gmi.cpp:23,50 */
int mvx_get_row_stat(const mvx_prob *P, int i);       /* gmi.cpp:45 */
int mvx_get_col_kind(const mvx_prob *P, int j);       /* gmi.cpp:18,51; util.cpp:444 */
int mvx_get_row_type(const mvx_prob *P, int i);       /* util.cpp:377 */
double mvx_get_row_lb(const mvx_prob *P, int i);      /* util.cpp:378 */
double mvx_get_row_ub(const mvx_prob *P, int i);      /* gmi.cpp:47 */
int mvx_get_col_type(const mvx_prob *P, int j);       /* util.cpp:319 */
double mvx_get_col_lb(const mvx_prob *P, int j);      /* util.cpp:320 */
double mvx_get_col_ub(const mvx_prob *P, int j);      /* gmi.cpp:52 */
const char *mvx_get_col_name(const mvx_prob *P, int j);
int mvx_get_mat_row(const mvx_prob *P, int i, int *ind, double *val);  /* gmi.cpp:84 */
int mvx_eval_tab_row(const mvx_prob *P, int k, int *ind, double *val); /* gmi.cpp:36 */
int mvx_get_it_cnt(const mvx_prob *P);
/* diagnostics of the anti-stalling rules.  After 64 + (m+n)/8 consecutive degenerate pivots, primal
   phase 2 first perturbs the bounds of the basic variables (once per solve; true bounds return when the
   phase ends); a second stall, the dual simplex and phase 1 fall back on Bland's smallest-subscript
   rule until a pivot moves again.  Counts: perturbations applied / pivots chosen by Bland's rule */
int mvx_get_pert_cnt(const mvx_prob *P);
/* > 0 overrides the number of consecutive degenerate pivots that arms the rules; 0 = default */
void mvx_set_stall_limit(int limit);
/* diagnostic (MVX_FCS_DBG=1): phase stamps of the step kernel's lead workgroup, 8 per chain position; returns rows */
int mvx_fcs_debug_stamps(unsigned long long *out);
int mvx_get_bland_cnt(const mvx_prob *P);
int mvx_term_out(int flag);      /* glp_term_out 2test.cpp:45,53,62 */
const char *mvx_version(void);   /* glp_version  util.cpp:278 */

/* generateCut3 (gmi.cpp:11-117) for `count` basic integer columns cols[0..count-1] (1-based) of a solved node in one
   device pass: tableau rows, the coefficient formula and the back-substitution over the model rows run on the GPU.
   vals is count x (n+1), row-major: vals[t][1..n] the cut coefficients, vals[t][0] = rhs[t] = its lower bound
   (the CutContainer of gmi.cpp:91-109); ok[t] = 0 where no cut exists.  repaired = 0: the formula as written, running
   right-hand side and positional back-substitution included; 1: mvx_generateCutGMI's.  Bit-identical to the
   one-column host functions of mvx_bnb.h.  Returns 0, -1 on bad arguments, -2 when the device is out of memory */
int mvx_gmi_cuts(const mvx_prob *P, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok);
/* The same for cuts taken from DIFFERENT solved handles in one call: cut t comes from Ps[t], column cols[t].  The handles
   must share their columns and their first model rows (clones of one root, each with its own bounds, basis and appended
   cut rows): a round of a B&B window takes one cut from each of its branching nodes (bs.cpp:249-258 run for 64 nodes at
   once).  vals is count x (n+1).  Returns -3 when the handles do not share a root (call mvx_gmi_cuts per handle then). */
int mvx_gmi_cuts_many(const mvx_prob *const *Ps, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok);

/* ---- engine-state access (parity tests, visualisers) --------------------------- */
int mvx_get_tableau_ld(const mvx_prob *P);
int mvx_get_tableau(const mvx_prob *P, double *out); /* (m+1) x (n+1), packed row-major */
int mvx_get_basis(const mvx_prob *P, int *head, int *nb, int *flag);

/* ---- device / measurement ------------------------------------------------------ */
/* number of visible HIP devices (0 when none); never aborts */
int mvx_device_count(void);
/* bind this process to device `dev` (one process per GPU); 0 on success */
int mvx_set_device(int dev);
/* HIP-event timing on the engine's own stream: accumulated milliseconds and launch count
   of the rank-1 update kernel since the last reset (only collected while enabled) */
void mvx_profile_enable(int on);
void mvx_profile_reset(void);
double mvx_profile_update_ms(void);
long long mvx_profile_update_launches(void);
/* wall time (ms, HIP events on the engine stream) of the last mvx_simplex call's device work */
double mvx_last_solve_ms(const mvx_prob *P);
/* node migration between ranks: image of bounds + basis + tableau in DEVICE
   memory of this process's GPU, ready for an RCCL send; the receiver rebuilds the handle on top of
   its own copy `base` of the root model.  The `_from` forms also carry the model rows appended since
   `base` (rows base.m+1 .. m: GMI cut rows, cut.cpp:23-43), so that a node with cuts can migrate; the plain
   forms carry none and need a receiver `base` with the same number of rows.  The image also carries the
   tolerances of the solve that produced the status, so a migrated optimal node is not solved again. */
long long mvx_pack_size(const mvx_prob *P);
int mvx_pack(const mvx_prob *P, void *dev_buf);
long long mvx_pack_size_from(const mvx_prob *P, const mvx_prob *base);
int mvx_pack_from(const mvx_prob *P, const mvx_prob *base, void *dev_buf);
int mvx_unpack(mvx_prob *dst, const mvx_prob *base, const void *dev_buf);
/* streamed-update tuning (row-block depth 8/16/32, batched-load hot loop, non-temporal access);
   for measurement sweeps -- results are identical for every setting */
void mvx_set_tuning(int tr, int hot, int nt);
/* Tableau refresh.  A dense Gauss-Jordan tableau carries the rounding of every pivot it has been through (GLPK
   refactorises its basis behind glp_simplex, bs.cpp:117).  A solve that ends OPTIMAL on a handle with at least
   `check_every` pivots since the last look (default 1024; clones inherit the count) computes the residual of the row
   equations, max_i |sum_j a_ij x_j - x_Ri| / (1 + |x_Ri|), over 32 rows picked from the pivot count (mvx_row_residual
   gives it over all rows); above `tol` (default 1e-9) the tableau
   is rebuilt from the model for the same basis -- slack tableau with the non-basic variables on their bounds, then
   the basic structural variables pivoted back in by ascending variable number, each on the row of largest |entry|
   among the rows whose auxiliary has to leave -- and the simplex carries on.  Same rule, same arithmetic, in the
   CPU oracle.  mvx_get_refresh_cnt: refreshes on this handle's lineage */
void mvx_set_refresh(int check_every, double tol);
int mvx_get_refresh_cnt(const mvx_prob *P);
double mvx_row_residual(const mvx_prob *P);
/* resident-tableau path of primal phase 2 (cache-resident sizes: one launch keeps the tableau in LDS for the whole
   run of pivots): 1 on, 0 off, -1 back to the default (on unless the environment has MVX_PERSIST=0).  Results are
   identical either way.  mvx_persist_stats: launches made / launches that aborted and were redone by the two-kernel path */
void mvx_set_persist(int mode);
/* Pivots one pass over the tableau applies on the fused primal path (the chained selection):
   0 = by tableau size (default), 1 = one pivot per pass, 2..32 = chains of that length.  Results do not depend on it. */
void mvx_set_chain(int len);
/* Who chooses the steps of a chain: 1 (default) one launch per chain -- up to 32 workgroups that stay resident and
   exchange their candidates through the L2 of the XCD they share (k_chain); 0 two launches per step (k_pc / k_pr), which
   also take over by themselves if a cluster launch ever gives up waiting for a peer.  Results do not depend on it.
   mvx_cluster_stats: cluster launches made / launches that gave up */
void mvx_set_cluster(int on);
void mvx_cluster_stats(long long *launches, long long *aborts);
/* the same for the dual simplex steps of the generic path (every warm-started B&B child): 0 = default (8 for launches
   shared by 32 or more node LPs and for tableaux of 16 MB and more, else 4), 1 = off, 2..8 = dual pivots one update
   pass applies */
void mvx_set_dual_chain(int len);
void mvx_persist_stats(long long *launches, long long *aborts);
/* shader-clock cycles workgroup 0 spent per phase of the resident-tableau loop, summed over launches: propose, gather,
   read, apply; out5[4] = pivots */
void mvx_persist_cycles(unsigned long long *out5);
/* number of handles that share one launch in mvx_simplex_batch (default 64, 2..256) */
void mvx_set_batch_slots(int slots);
/* block until all work queued on the engine stream has finished */
void mvx_sync(void);
/* Device out-of-memory never aborts: a solve that cannot get its tableau returns MVX_EFAIL (status MVX_UNDEF), a clone
   that cannot get one keeps the model only, a row append that cannot grow its slab drops the tableau (the next solve
   restarts from the slack basis) -- and this call reads MVX_ENOMEM once (0 = nothing happened since the last call).
   Threads: callers use one host thread at a time per handle (GLPK's contract); the engine itself is safe for the one
   split mvx_branchAndBound makes -- batch solves on a worker thread while the calling thread clones, edits, queries
   and deletes OTHER handles. */
int mvx_last_error(void);
/* HIP's current device is per host thread: a thread other than the one that made the first engine call calls this
   once before it uses handles (0 on success) */
int mvx_bind_thread(void);

#ifdef __cplusplus
}
#endif
#endif
