// mvx_internal.hpp -- host-side model + device control block of the MI355X LP engine.
//
// Layout in HBM (one slab per problem handle, see DESIGN.md "Data layout"):
//   T      (m_cap+1) x ld doubles, row-major, ld a multiple of 32 doubles (256 B rows)
//          T[0][0] objective, T[0][j] reduced costs, T[i][0] basic values, T[i][j] body
//   bvar/blb/bub   per tableau row   (basic variable, its bounds)
//   nvar/nflag/nlb/nub per tableau column (non-basic variable, status, bounds)
// Everything position-indexed so that every kernel access is contiguous.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mvx.h"

struct DevMatrix;
namespace mvx {

constexpr int ROWCOMB_CHUNK = 64; // rows per partial sum of k_rowcomb (fixed summation order)
constexpr int LD_ALIGN = 32;      // doubles; 256-byte rows
constexpr int ROW_SLACK = 64;     // spare tableau rows per handle for cut appends (cut.cpp:23)
constexpr double DEGEN_TOL = 1e-9; // a step / dual ratio no longer than this counts as degenerate (oracle: DEGEN_TOL)
constexpr double PERT_EPS = 1e-6; // relative size of the anti-stalling bound perturbation (oracle: PERT_EPS)
constexpr size_t NT_THRESHOLD_BYTES = (size_t)320 << 20; // tableaux larger than this stream with non-temporal access (pick_nt)
constexpr size_t WT_MIN_BYTES = (size_t)96 << 20, WT_MAX_BYTES = (size_t)272 << 20; // write-through stores in this band (pick_nt)
constexpr int KCH = 32;          // most steps one bulk launch of the chained primal path applies (k_fcs / k_fbc3)
constexpr int DCH_MAX = 8;  // most dual pivots one k_update applies (dual_chain)
constexpr int DA_THREADS = 1024; // k_dboot / k_da workgroup size: the O(m) leaving-row pass is redundant per block
constexpr int MAX_EDITS = 8;      // pending bound edits a control block carries (more are flushed by launches)
constexpr int ROW_SPARE = 32;     // rows behind row m that always exist: k_fb streams whole row tiles

// state-machine phases (device-driven; mirrors orc_simplex's round loop)
enum : int { PH_START = 0, PH_PRIMAL2 = 1, PH_DUAL = 2, PH_PHASE1 = 3 };
// done codes
enum : int { D_RUN = 0, D_OPT = 1, D_UNBND = 2, D_NOFEAS = 3, D_ITLIM = 4, D_PFEAS = 5, D_FAIL = 6, D_NEED_PHASE1 = 7 };
// step kinds
enum : int { ST_NONE = 0, ST_PIVOT = 1, ST_FLIP = 2, ST_STOP = 3 };
// fused primal fast path state
enum : int { F_OFF = 0, F_RUN = 1, F_STOP = 2, F_RUN_DUAL = 3 };

// reduction candidate: (k1, k2, idx) is a strict total order, aux rides along
struct Cand {
  double k1, k2;
  int idx, aux;
};


// Device-resident control block.  Kernels take only a pointer to it, so one launch
// sequence serves every problem handle and pivots can be queued ahead of the host.
struct Ctl {
  // geometry + pointers (written by the host before a solve)
  double *T;
  int *bvar; double *blb; double *bub;
  int *nvar; int *nflag; double *nlb; double *nub;
  double *colq;   // [m_cap+1] copy of the pivot column (old values)
  double *srow;   // [ld]      pivot row / pivot
  double *cost1;  // [ld]      phase-1 cost row
  double *wts;    // [m_cap+1] row weights for k_rowcomb
  int *gflag;     // [m_cap+1] phase-1 infeasibility signs
  double *part;   // [nchunks x ld] partial sums of k_rowcomb
  double *rc_base; // [ld] base vector for k_rowcomb (nullable)
  double *rc_out;  // destination of k_rowcomb
  int m, n, ld, m_cap;
  // parameters
  double sgn, tol_bnd, tol_dj, tol_piv;
  // running state
  int phase, done, rounds, budget;
  int it_cnt, n_flips, n_bland;
  int step, p, q, sdir, p_up, leave_flag;
  double piv, bound, xq, delta;
  // fused primal fast path (k_fboot / k_fa / k_fb): ping-pong buffers indexed by parity
  double *colqx[2]; // [m_cap+1] contiguous copy of the NEXT pivot column, exported by k_fb
  double *betac[2]; // [m_cap+1] contiguous copy of column 0 (basic values), exported by k_fb
  Cand *pp[2];      // [npb] pricing partials (one per k_fa block)
  Cand *rp;         // [nrb] ratio-test partials (one per k_fb row block)
  int npb, nrb;
  int fstate, curA, curB, flipflag;
  // anti-cycling (oracle: ctl_t.stall): consecutive degenerate pivots; from stall_limit on every choice
  // follows Bland's smallest-subscript rule until a pivot moves again
  int stall, stall_limit;
  // bound perturbation against stalling (oracle: perturb_basis / restore_bounds): original bounds by
  // variable number, saved when the perturbation is applied
  double *olb, *oub;
  int perturbed, pert_used, n_pert;
  // primal phase 1: the infeasibility-sum cost row lives in tableau row m+1 (first spare row) and is carried
  // through the pivots by k_update; k_p1_head lists the rows whose sign changed (p1_list / wts), k_p1_fix adds them
  int *p1_list;
  int p1_init, p1_nchg, p1_fix_q, p1_fix_g;
  double *pw[2]; // [ld] primal devex reference weights by non-basic position (oracle: ctl_t.pw); the current set
                 // is pw[curA]: the fused path writes the other one and k_fb flips curA, the generic path updates in place
  double *dw; // [m_cap+1] dual devex reference weights by row (oracle: dual_simplex's w), reset on entering the dual phase
  int stall_new; // fused path: k_fa's verdict on the step it prepared, committed by k_fb (k_fa workgroups read `stall`)
  double ent_lb, ent_ub;
  // bound edits of basic variables made since the last solve (glp_set_col_bnds on a branching child, bs.cpp:274,282;
  // glp_set_row_bnds on a fresh cut row, cut.cpp:43): applied by the first k_select of the solve instead of one
  // launch per edit
  int n_edits, edit_row[MAX_EDITS];
  double edit_lb[MAX_EDITS], edit_ub[MAX_EDITS];
  int job; // batched solve: which job of the queue this slot is working on (-1: none)
  // fused dual path (k_dboot / k_da / k_fb<DUAL>): dual devex weights in two sets (the current one is dwx[curA & 1];
  // the generic path updates it in place, k_da writes the other and k_fb flips curA), and the leaving row of the
  // NEXT pivot, chosen by k_da from column 0 before the bulk update (ping-pong like the other fused buffers)
  double *dwx[2];
  int p_nextx[2], p_up_nextx[2];
  int npbd; // number of dual-ratio partials (k_da blocks of DA_THREADS columns)
  // Chained primal path (k_fcc / k_fcr / k_fbc): the pivots after the one k_fa prepared are chosen from O(m + n) slices of the
  // tableau as it stands -- column q_k and row p_k, carried through the earlier pivots of the chain entry by entry --
  // and ONE bulk launch applies the whole chain (each entry read and written once for up to KCH pivots).  Step 0 of a
  // chain is the step k_fa left in the fields above; ch_*[l] describes step l >= 1 (index 0 is filled for uniform loops).
  int chain_max, nch; // chain length allowed by the host (1 = off) / prepared for the coming bulk launch
  int dchain_max;     // the same for the dual steps of the generic path (dual_chain in k_select)
  int n_bulk;         // bulk launches that stepped so far in this solve (one per pivot or flip; one per chain)
  int ch_alive, ch_nrpc; // the chain may still grow / ratio-test partials k_fcc left
  int ch_p[KCH], ch_q[KCH], ch_pup[KCH], ch_lf[KCH], ch_stall[KCH], ch_sdir[KCH], ch_fq[KCH];
  double ch_piv[KCH], ch_bound[KCH], ch_xq[KCH], ch_s0[KCH], ch_dq[KCH], ch_wq[KCH];
  Cand *rpc; // [ceil(m_cap / 256)] ratio-test partials of k_fcc
  double ch_elb[KCH], ch_eub[KCH]; // bounds of the entering variable (become row p's)
  double ch_llb[KCH], ch_lub[KCH]; // bounds of the leaving variable (become column q's)
  double *srowk[KCH], *colqk[KCH]; // scaled pivot row / pivot column of step l
  // Chained primal path (k_pboot / k_pc / k_pr / k_fbc3): what the device decides about the pending chain
  int pc_n;        // steps recorded in the pending chain (k_fbc3 applies them and resets it)
  int pc_epoch;    // chains applied so far + 1
  unsigned pc_arrive; // k_fbc3: workgroups that have finished (the last one commits the chain's bookkeeping)
  int ch_kind[KCH], ch_cnt[KCH], ch_ok[KCH]; // ST_PIVOT / ST_FLIP; pivots among steps 0..l; == pc_epoch once step l is recorded
  int ch_okc[KCH];  // == pc_epoch once k_pc has chosen step l's entering column
  int hd_q[KCH], hd_sdir[KCH], hd_fq[KCH]; // the entering column of step l as k_pc found it: column, direction, status,
  double hd_dq[KCH], hd_wq[KCH], hd_lbq[KCH], hd_ubq[KCH]; // reduced cost, weight, bounds
  double ch_delta[KCH]; // bound flips: the entering variable's move
  int pc_itlim;         // k_chain: the pivot limit is reached once the pending chain is applied and an entering column is still
                        // on offer: the generic step that closes the batch reports it without pricing again
  int dsel;             // k_dsel has prepared the coming step (a dual phase carrying on): the k_select that follows returns at once
  int cl_abort;         // k_chain gave up waiting for its peer workgroups: the chain was dropped, nothing was changed
  unsigned long long *dbg; // diagnostic phase stamps of k_fcs (MVX_FCS_DBG=1), nullptr otherwise
};

// What the kernels of the chained primal path (k_pboot / k_pc / k_pr / k_fbc3) need that the host knows -- pointers,
// geometry, tolerances: passed by value, so that no dependent load stands between a launch and its data; what the
// device decides -- the chain, the counters, the verdicts -- stays in the control block.
// State that a step changes is kept in two sets that alternate with the step index (a launch never writes what a
// workgroup of the same launch may still read); the chain starts from the handle's own arrays ("base").
struct ChainArgs {
  Ctl *c;
  double *T;
  double *blb, *bub, *nlb, *nub; // base: the handle's own arrays (the bulk launch's commit brings them up to date)
  int *nflag;
  double *pw[2];                 // base weights: both sets hold them between chains (the generic step reads pw[curA & 1])
  double *betab;                 // base basic values: column 0, exported contiguously by k_pboot / k_fbc3
  // column side, as of step g in set g & 1: objective row, devex weights, statuses and bounds of the non-basic variables
  double *drowk[2], *pwk[2], *nlbk[2], *nubk[2];
  int *nflagk[2];
  // row side, as of step g in set g & 1: basic values and bounds of the basic variables
  double *betak[2], *blbk[2], *bubk[2];
  double *pp; // pricing partials of k_pr / k_pboot: 8 fields x ppstride (score, q, sdir, d_q, w_q, lb_q, ub_q, flag_q)
  double *rp; // ratio-test partials of k_pc: 4 fields x rpstride (step, |a|, row, to_upper)
  size_t ppstride, rpstride;
  double *srow0, *colq0; // scaled pivot rows / pivot columns of the chain's steps, `sstride` / `cstride` doubles apart
  size_t sstride, cstride;
  int m, n, ld, mcap1, ncb, nrb; // ncb column blocks (k_pr), nrb row blocks (k_pc)
  double tol_dj, tol_piv, tol_bnd, sgn;
  int stall_limit;
  const double *zeros; // max(ld, m_cap + 1) zeros: the operands of a bound flip in the bulk pass
  // one-XCD cluster selection (k_chain): exchange area, tags, participants, steps this launch may take
  unsigned *xg;
  int xg_bytes;
  unsigned tagbase;
  int nw, kmax;
  int boot; // first chain launch of a batch: k_chain does what k_pboot does for k_pc / k_pr (feasibility of the start, fresh weights)
  int *xabort;
};

// Work queue of a batched solve (mvx_simplex_batch): the host uploads one control block per handle (`jobs`), the slots
// of the launch pull them -- a slot whose solve has ended exports its mirrors into the job's staging area and takes the
// next job in the very launch that finds it idle, instead of idling until the host's next synchronisation point.
struct SlotScratch { // per-slot scratch the pulled control block is pointed at
  double *colq, *srow, *olb, *oub, *dw, *pw;
  double *chain; // DCH_MAX - 1 x (pivot column [m_cap+1 rounded], scaled pivot row [ld]) for dual chains
  size_t chain_col, chain_stride; // bytes: offset of the row part inside one step's pair, size of a pair
};
struct BatchQueue {
  const Ctl *jobs;
  const SlotScratch *scratch;
  int *counters; // [0] next job to hand out, [1] jobs finished and exported, [2] rounds (k_select launches) so far, [3] round of the last job's end
  unsigned char *stage;
  size_t stage_stride;
  int count;
};

// one launch of k_copy_many: up to COPY_BATCH byte ranges (16-byte aligned, sizes multiples of 16)
constexpr int COPY_BATCH = 32;
struct CopyJob {
  const void *src;
  void *dst;
  size_t bytes;
};
struct CopyBatch {
  int count;
  CopyJob jobs[COPY_BATCH];
};

// one cut of a GMI launch: the solved handle it is taken from (a round of a B&B window takes one cut from each of up to
// 64 node LPs: same columns and same first m0 model rows, their own tableaux, bases and appended cut rows)
struct GmiNode {
  const double *T; // tableau
  const int *nvar, *nflag;
  const double *nlb, *nub;
  int m, ld, pos, pad; // rows of this handle, its row stride, the tableau row of the cut's basic column
};

// arguments of the GMI cut kernels (k_gmi_work / k_gmi_backsub): `count` cuts, each with its node
struct GmiArgs {
  const GmiNode *nodes; // [count]
  const int *kind;     // [n+1] column kinds of the model (MVX_CV / MVX_IV), device copy
  double *work;        // [count][wld] coefficients by variable number 0..m+n (gmi.cpp:29-32 `work`)
  double *rhs;         // [count]
  int *ok;             // [count] 0 = no valid cut (repaired mode: free non-basic with a non-zero entry)
  const double *A;     // [m0+1][lda] model rows 1..m0: packed non-zeros (bug-compatible) or by column (repaired)
  const int *len;      // [m0+1] non-zeros per row (nullptr: every row dense)
  double *out;         // [count][old] cut coefficients by structural column 1..n after rows 1..m0
  int n, wld, lda, m0, old, count, mode; // mode 0 bug-compatible (gmi.cpp:41-89), 1 repaired
};

// shared immutable matrix row (1-based, n+1 doubles)
using RowPtr = std::shared_ptr<std::vector<double>>;

// A vector a clone shares with its source until one of them writes: B&B clones copy a handle thousands of times a second
// and never touch the row list (a cut appends to it: that node's list parts from its siblings' then) or the names.
// Reads go through operator[] / get(); every write goes through mut(), which parts from the sharers first.
template <class T>
class CowVec {
  std::shared_ptr<std::vector<T>> p_ = std::make_shared<std::vector<T>>();

public:
  const T &operator[](size_t i) const { return (*p_)[i]; }
  size_t size() const { return p_->size(); }
  bool empty() const { return p_->empty(); }
  const std::vector<T> &get() const { return *p_; }
  std::vector<T> &mut() {
    if (p_.use_count() != 1) p_ = std::make_shared<std::vector<T>>(*p_);
    return *p_;
  }
};

// The rows of a model: a frozen head shared by every clone (one reference count) and this handle's own tail.  B&B clones
// copy a handle thousands of times a second; what a node adds to its parent's model is a cut row or two, which goes to
// the tail -- a clone costs one count plus the tail, an append touches nothing shared.  (A plain shared list of 513 row
// pointers cost 513 counts per clone; a copy-on-write list moved the same cost to every node's first cut.)  freeze()
// folds the tail into a new head; the solve entry calls it when the tail has grown long (a freshly loaded model).
class RowList {
  std::shared_ptr<std::vector<RowPtr>> head_ = std::make_shared<std::vector<RowPtr>>();
  std::vector<RowPtr> tail_;

public:
  size_t size() const { return head_->size() + tail_.size(); }
  const RowPtr &operator[](size_t i) const { return i < head_->size() ? (*head_)[i] : tail_[i - head_->size()]; }
  size_t tail_size() const { return tail_.size(); }
  void push_back(RowPtr r) { tail_.push_back(std::move(r)); }
  void set(size_t i, RowPtr r) {
    if (i >= head_->size()) {
      tail_[i - head_->size()] = std::move(r);
      return;
    }
    if (head_.use_count() != 1) head_ = std::make_shared<std::vector<RowPtr>>(*head_);
    (*head_)[i] = std::move(r);
  }
  void resize(size_t n) {
    if (n >= head_->size()) {
      tail_.resize(n - head_->size());
      return;
    }
    if (head_.use_count() != 1) head_ = std::make_shared<std::vector<RowPtr>>(head_->begin(), head_->begin() + (long)n);
    else head_->resize(n);
    tail_.clear();
  }
  void reset(size_t n) { // n empty rows
    head_ = std::make_shared<std::vector<RowPtr>>();
    tail_.assign(n, RowPtr());
  }
  void freeze() {
    if (tail_.empty()) return;
    auto h = std::make_shared<std::vector<RowPtr>>();
    h->reserve(size());
    h->insert(h->end(), head_->begin(), head_->end());
    h->insert(h->end(), tail_.begin(), tail_.end());
    head_ = std::move(h);
    tail_.clear();
  }
};

} // namespace mvx

struct mvx_prob {
  // ---- model (host) ----
  int m = 0, n = 0;
  int dir = MVX_MIN;
  mvx::RowList A; // A[i], i=1..m; rows are shared between clones (copy-on-write), and so is the list's frozen head
  std::vector<double> c;      // c[0..n]
  std::vector<int> kind;      // kind[1..n]
  mvx::CowVec<std::string> cname;
  std::vector<int> rtype;
  std::vector<double> rlb, rub; // normalised (+-inf when absent)
  std::vector<int> ctype;
  std::vector<double> clb, cub;
  // ---- engine state ----
  bool valid = false;   // device tableau + basis exist
  int status = MVX_UNDEF;
  int it_cnt = 0;
  int pert_cnt = 0;  // bound perturbations applied against stalling, diagnostic
  int bland_cnt = 0; // pivots chosen under the anti-cycling (Bland) rule, diagnostic
  double last_ms = 0.0;
  double last_tol[3] = {0.0, 0.0, 0.0}; // tolerances of the solve that produced `status`
  int piv_since_check = 0; // pivots since the residual of the row equations was last looked at (clones inherit it)
  int refresh_cnt = 0;     // tableau refreshes, diagnostic
  bool hint_dual = false; // last edit made a basic variable infeasible: start in the dual simplex
  // bound edits of basic variables not yet on the device: (row position, lb, ub); the next solve's control block carries them
  struct Edit {
    int row;
    double lb, ub;
  };
  std::vector<Edit> pending;
  // device copy of model rows 1..m0 for the GMI back-substitution (gmi.cpp:81-89), shared by every clone whose first
  // m0 rows are the same objects (B&B nodes share their root's rows); built on first use
  std::shared_ptr<struct DevMatrix> dmat;
  // host mirrors of the basis (always in sync while valid)
  std::vector<int> bvar, nvar, nflag; // [m+1], [n+1], [n+1]
  std::vector<int> pos;               // pos[k], k=1..m+n: +row or -column
  // cached solution vectors (refreshed by export after each solve / modification)
  mutable bool sol_fresh = false;
  // while !sol_fresh: rows 0..fresh_rows of `beta` (and all of `dj`) are still the tableau's -- the only edits since
  // the last export appended rows behind them or rewrote a row behind them (a cut: cut.cpp:40) -- so reading the value
  // of a variable that is basic in one of those rows needs no device export (-1: nothing is known to be current)
  mutable int fresh_rows = -1;
  mutable std::vector<double> beta; // [m+1], beta[0] = objective
  mutable std::vector<double> dj;   // [n+1]
  // ---- device slab ----
  void *slab = nullptr;
  size_t slab_bytes = 0;
  int m_cap = 0, ld = 0;
  double *d_T = nullptr;
  int *d_bvar = nullptr; double *d_blb = nullptr; double *d_bub = nullptr;
  int *d_nvar = nullptr; int *d_nflag = nullptr; double *d_nlb = nullptr; double *d_nub = nullptr;
};
