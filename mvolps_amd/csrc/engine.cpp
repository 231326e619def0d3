// engine.cpp -- host side of the MI355X dense-simplex engine: device context, per-handle
// HBM slabs, the queued pivot loop, tableau maintenance under model edits, and the export
// of basis / solution mirrors.  Counterpart of glp_simplex and the glp_* edit calls MVOLPS
// makes (/root/reference/bs.cpp:114-117,274-288; cut.cpp:23-43).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "engine.hpp"

// Device copy of model rows 1..m0 (see mvx_prob::dmat).  `plain` is indexed by column; `packed` holds each row's
// non-zeros side by side (what position k of glp_get_mat_row's list means, gmi.cpp:84-87) and is the same buffer
// when every row is dense.
struct DevMatrix {
  int m0 = 0, n = 0, lda = 0;
  double *plain = nullptr, *packed = nullptr;
  int *len = nullptr; // nullptr: every row has n non-zeros
  std::vector<const void *> rows; // identity of the host rows this was built from
  ~DevMatrix() {
    if (packed && packed != plain) (void)hipFree(packed);
    if (plain) (void)hipFree(plain);
    if (len) (void)hipFree(len);
  }
};

namespace mvx {

#define HIPCHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      std::fprintf(stderr, "mvx: HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
      std::abort();                                                                            \
    }                                                                                          \
  } while (0)

// kernel launch wrappers (kernels.hip)
void set_tuning(int tr, int hot, int nt);
int fused_npb(int n);
int fused_nrb_max(int m);
int chain_ncb(int n);
int chain_nrb(int m);
void launch_pboot(const ChainArgs &, hipStream_t);
void launch_pstep(const ChainArgs &, int g, hipStream_t);
void launch_pc(const ChainArgs &, int g, hipStream_t);
void launch_fbc3(const ChainArgs &, int steps, hipStream_t);
void launch_fpatch(const ChainArgs &, int steps, hipStream_t);
int launch_chain(const ChainArgs &, hipStream_t);
int chain_cluster_nw(int m, int n);
int chain_cluster_kmax(int m, int n);
void launch_dboot(Ctl *, int n, hipStream_t);
void launch_da(Ctl *, int n, hipStream_t);
void launch_db(Ctl *, int m, int n, hipStream_t);
void launch_select(Ctl *, hipStream_t, int slots = 1);
int launch_dsel(Ctl *, int m, int n, hipStream_t, int slots = 1);
void launch_select_queue(Ctl *, const BatchQueue &q, hipStream_t, int slots);
void launch_update(Ctl *, int m, int n, hipStream_t, int slots = 1, int chained = 0, int busy_slots = 0);
void launch_p1_head(Ctl *, hipStream_t);
void launch_p1_select(Ctl *, hipStream_t);
void launch_p1_fix(Ctl *, int n, hipStream_t);
void launch_scatter_ctl(Ctl *dst, const Ctl *src, const int *idx, int count, hipStream_t);
void launch_copy_many(const CopyBatch &b, hipStream_t);
void launch_gmi(const GmiArgs &a, hipStream_t);
void launch_refresh_select(Ctl *, const int *tflag, int var, hipStream_t);
size_t persist_lds_bytes(int m, int cpw);
int persist_max_cpw();
int persist_slot_words(int m_cap);
int launch_persist(Ctl *, unsigned long long *head, unsigned long long *slot, int *abort_flag, unsigned long long *dbg, int m, int cpw, int nw,
                   int slot_stride, int max_steps, int head_stride, hipStream_t);
void launch_rowcomb(Ctl *, int m, int n, int respect_done, hipStream_t);
void launch_shift_nonbasic(double *T, int ld, int m, int jj, double delta, hipStream_t);
void launch_set_basic_bounds(double *blb, double *bub, int i, double lb, double ub, hipStream_t);
void launch_set_nonbasic(double *nlb, double *nub, int *nflag, int j, double lb, double ub, int flag, hipStream_t);
void launch_add_rows(double *T, int ld, int n, int *bvar, double *blb, double *bub, int *nvar, int first, int nrs, int m_new,
                     hipStream_t);
void launch_export(Ctl *, unsigned char *stage, int m, int n, int force, hipStream_t, int slots = 1, size_t slot_stride = 0);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------ context
// One solve context = one HIP stream with its own control block, scratch and staging.  The main
// context serves every single-handle call; mvx_simplex_batch has its own (BatchCtx below).
constexpr size_t XG_BYTES = 4 * 8 * 64 * 16; // k_chain's exchange area: 4 regions x 8 fields x 64 records x 16 bytes
struct SolveCtx {
  hipStream_t stream = nullptr;
  Ctl *d_ctl = nullptr;
  Ctl *h_ctl = nullptr; // pinned
  // scratch sized for the largest problem seen
  int sc_m_cap = 0, sc_ld = 0;
  void *scratch = nullptr;
  double *d_colq = nullptr, *d_srow = nullptr, *d_cost1 = nullptr, *d_wts = nullptr, *d_part = nullptr, *d_rcbase = nullptr;
  int *d_gflag = nullptr;
  double *d_colqx[2] = {nullptr, nullptr}, *d_betac[2] = {nullptr, nullptr};
  double *d_dw = nullptr, *d_dw2 = nullptr; // dual devex weights by row, two sets (fused dual path ping-pong)
  int *d_p1list = nullptr; // phase 1: rows whose infeasibility sign changed
  int *d_tflag = nullptr;  // tableau refresh: target non-basic status by variable number
  double *d_pw[2] = {nullptr, nullptr}; // primal devex weights by column, two sets (fused path ping-pong)
  double *d_srowk[KCH] = {}, *d_colqk[KCH] = {}; // chained primal path: scaled pivot rows / pivot columns of the steps
  // chained primal path: what a step changes besides the tableau, in two alternating sets (ChainArgs), and the partials
  double *d_drowk[2] = {}, *d_pwk[2] = {}, *d_nlbk[2] = {}, *d_nubk[2] = {}, *d_betak[2] = {}, *d_blbk[2] = {}, *d_bubk[2] = {};
  int *d_nflagk[2] = {};
  double *d_betab = nullptr, *d_ppart = nullptr, *d_rpart = nullptr;
  double *d_zeros = nullptr; // chained primal path: zeros (bound-flip operands of the bulk pass)
  // row combinations queued without a host round trip each (cut rows of a B&B round): a ring of pinned staging slots
  // (control block, row weights, base row); the stream is synchronised only when the ring wraps
  unsigned char *rc_ring = nullptr;
  size_t rc_slot_bytes = 0;
  int rc_next = 0, rc_m_cap = 0, rc_ld = 0;
  // cluster selection (k_chain): exchange area, abort flag, tag of the next launch's first exchange
  unsigned *d_xg = nullptr;
  int *d_xabort = nullptr;
  unsigned xtag = 0;
  size_t pp_stride = 0, rp_stride = 0;
  size_t sk_stride = 0, ck_stride = 0; // doubles between the chain's consecutive scaled pivot rows / pivot columns
  Cand *d_rpc = nullptr;
  double *d_olb = nullptr, *d_oub = nullptr; // bounds by variable number, saved by the anti-stalling perturbation
  Cand *d_pp[2] = {nullptr, nullptr}, *d_rp = nullptr;
  // staging area of k_export.  Default: ONE pinned host buffer that the kernel writes straight over PCIe (d_stage is its
  // device-side address): no device-to-host copy behind the last kernel of a call (~16 us: 12 us to start the copy
  // engine, 4 us of copy).  MVX_ZC_STAGE=0: a device buffer and a copy, as before.
  unsigned char *d_stage = nullptr, *h_stage = nullptr;
  bool stage_zc = false;
  size_t stage_bytes = 0;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  // resident-tableau path (k_persist): candidate granules, pivot messages, abort flag; a backup of the slab, the
  // control block and the devex weights taken in front of every launch, put back if the launch aborts
  unsigned long long *d_pcand = nullptr, *d_pmsg = nullptr;
  int *d_pabort = nullptr, *h_pabort = nullptr;
  int p_msg_words = 0;
  void *p_backup = nullptr;
  size_t p_backup_bytes = 0;
  Ctl *d_pctl = nullptr;
  double *d_ppw = nullptr;
  int p_pw_ld = 0;
};

// Slabs carved out of one multi-slab allocation (a growing B&B tree asks for one slab per open node: hipMalloc
// costs ~60 us a call, a chunk of up to 32 amortises it).  An arena goes back to the driver only whole, when every
// slab of it is idle.
struct Arena {
  void *base = nullptr;
  size_t slab_bytes = 0;
  int count = 0, idle = 0;
};

// Slab recycling (B&B clones come and go at one size).  Shared by the calling thread and the B&B driver's worker
// thread (bnb.cpp: child solves run on a worker while the caller clones and deletes other handles), hence the lock.
struct SlabCache {
  std::mutex mu;
  std::multimap<size_t, void *> free_slabs;
  std::unordered_map<void *, int> arena_of; // slab -> index into arenas
  std::vector<Arena> arenas;
  size_t idle_bytes = 0; // bytes held by idle slabs, arena slabs included
};

struct Context {
  int dev = -1;
  bool aux_ready = false;
  // every entry point that works on the main context (its stream, control block, scratch, staging buffer) holds
  // this lock: single-handle solves, model edits, clones, queries that refresh the mirrors
  std::recursive_mutex main_mu;
  SolveCtx main;
  // single-handle solves issued from inside a batch call (a lone pending handle, the phase-1 fallback) run on a
  // context of their own: the batch call may come from the B&B driver's worker thread while the calling thread
  // uses `main` for cut rows and solution queries
  SolveCtx aux;
  std::mutex aux_mu; // one single-handle solve at a time on `aux` (two batch calls may run on two threads)
  SlabCache slabs;
  // clones recorded by engine_copy and not yet launched (main_mu held): a round of B&B branchings makes one clone per
  // branching with only host-side work in between, so they leave as ONE k_copy_many launch when the next entry point
  // that touches device data arrives (flush_copies)
  std::vector<CopyJob> copies;
  // GMI cut generation (engine_gmi_cuts): device buffers for `work`, the cuts, right-hand sides, flags, row positions
  // and column kinds, and the pinned host side of the transfers
  void *gmi_dev = nullptr, *gmi_host = nullptr;
  size_t gmi_bytes = 0;
  // profiling (main context only)
  bool prof = false;
  double prof_update_ms = 0.0;
  long long prof_update_n = 0;
  std::vector<hipEvent_t> ev_pool;
};
#define MAIN_LOCK(c) std::lock_guard<std::recursive_mutex> main_lock_((c).main_mu)

static Context *g_ctx = nullptr;
static std::mutex g_ctx_mu;
static std::atomic<int> g_last_error{0}; // MVX_ENOMEM ...: read and cleared by mvx_last_error()
static int g_stall_limit = 0;      // > 0: overrides 64 + (m+n)/8 (tests drive the anti-stalling rules with it)
static int g_requested_dev = -1;

int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int set_device(int dev) {
  if (g_ctx && g_ctx->dev != dev) return -1; // one device per process
  g_requested_dev = dev;
  return 0;
}

static void init_solve_ctx(SolveCtx &sc) {
  HIPCHECK(hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking));
  HIPCHECK(hipMalloc((void **)&sc.d_ctl, sizeof(Ctl)));
  HIPCHECK(hipHostMalloc((void **)&sc.h_ctl, sizeof(Ctl)));
  HIPCHECK(hipEventCreate(&sc.ev_a));
  HIPCHECK(hipEventCreate(&sc.ev_b));
  HIPCHECK(hipMalloc((void **)&sc.d_xg, XG_BYTES + 256));
  HIPCHECK(hipMemsetAsync(sc.d_xg, 0, XG_BYTES + 256, sc.stream)); // ordered with the launches that use it (a null-stream memset is not)
  sc.d_xabort = (int *)((unsigned char *)sc.d_xg + XG_BYTES);
}

int take_last_error() { return g_last_error.exchange(0); }

// the current device is per host thread: a thread other than the one that made the first engine call binds itself
int bind_thread() {
  if (!g_ctx) return -1;
  return hipSetDevice(g_ctx->dev) == hipSuccess ? 0 : -1;
}

static Context &ctx() {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (g_ctx) return *g_ctx;
  int n = device_count();
  if (n <= 0) {
    std::fprintf(stderr,
                 "mvx: no HIP device visible -- the MI355X (gfx950) engine cannot run and this library has no CPU "
                 "fallback\n");
    std::abort();
  }
  Context *c = new Context();
  c->dev = g_requested_dev >= 0 ? g_requested_dev : 0;
  HIPCHECK(hipSetDevice(c->dev));
  init_solve_ctx(c->main);
  g_ctx = c;
  return *c;
}

static void sync_batch_stream();
static void flush_copies(Context &c);
void sync_stream() {
  if (!g_ctx) return;
  {
    MAIN_LOCK(*g_ctx);
    flush_copies(*g_ctx);
  }
  HIPCHECK(hipStreamSynchronize(g_ctx->main.stream));
  sync_batch_stream();
}

// caller holds main_mu.  Launch the recorded clones on the main stream, COPY_BATCH ranges per launch.
static void flush_copies(Context &c) {
  if (c.copies.empty()) return;
  for (size_t k0 = 0; k0 < c.copies.size(); k0 += COPY_BATCH) {
    CopyBatch b;
    b.count = (int)std::min<size_t>(COPY_BATCH, c.copies.size() - k0);
    for (int k = 0; k < b.count; k++) b.jobs[k] = c.copies[k0 + (size_t)k];
    launch_copy_many(b, c.main.stream);
  }
  c.copies.clear();
}

static size_t stage_size(int m_cap, int ld) {
  return align_up(sizeof(Ctl) + (size_t)(m_cap + 1) * 8 + (size_t)ld * 8 + (size_t)(m_cap + 1) * 4 + (size_t)ld * 8, 256);
}

static void ensure_scratch(SolveCtx &sc, int m_cap, int ld) {
  if (m_cap <= sc.sc_m_cap && ld <= sc.sc_ld) return;
  HIPCHECK(hipStreamSynchronize(sc.stream));
  int mc = m_cap > sc.sc_m_cap ? m_cap : sc.sc_m_cap;
  int l = ld > sc.sc_ld ? ld : sc.sc_ld;
  if (sc.scratch) HIPCHECK(hipFree(sc.scratch));
  if (sc.d_stage && !sc.stage_zc) HIPCHECK(hipFree(sc.d_stage));
  if (sc.h_stage) HIPCHECK(hipHostFree(sc.h_stage));
  sc.d_stage = nullptr;
  const int nchunks = (mc + ROWCOMB_CHUNK - 1) / ROWCOMB_CHUNK + 1;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  size_t o_colq = carve((size_t)(mc + 1) * 8), o_srow = carve((size_t)l * 8), o_cost1 = carve((size_t)l * 8);
  size_t o_wts = carve((size_t)(mc + 1) * 8), o_rcb = carve((size_t)l * 8), o_g = carve((size_t)(mc + 1) * 4);
  size_t o_part = carve((size_t)nchunks * l * 8);
  size_t o_cx0 = carve((size_t)(mc + 1) * 8), o_cx1 = carve((size_t)(mc + 1) * 8);
  size_t o_bc0 = carve((size_t)(mc + 1) * 8), o_bc1 = carve((size_t)(mc + 1) * 8);
  size_t o_pp0 = carve((size_t)fused_npb(l) * sizeof(Cand)), o_pp1 = carve((size_t)fused_npb(l) * sizeof(Cand));
  size_t o_rp = carve((size_t)fused_nrb_max(mc) * sizeof(Cand));
  size_t o_olb = carve((size_t)(mc + l + 1) * 8), o_oub = carve((size_t)(mc + l + 1) * 8);
  size_t o_dw = carve((size_t)(mc + 1) * 8), o_dw2 = carve((size_t)(mc + 1) * 8);
  size_t o_p1l = carve((size_t)(mc + 2) * 4);
  size_t o_pw0 = carve((size_t)l * 8), o_pw1 = carve((size_t)l * 8);
  size_t o_tf = carve((size_t)(mc + l + 1) * 4);
  size_t o_sk[KCH], o_ck[KCH];
  size_t o_rpc = carve((size_t)((mc + 255) / 256 + 1) * sizeof(Cand));
  for (int k = 0; k < KCH; k++) o_sk[k] = carve((size_t)l * 8);
  for (int k = 0; k < KCH; k++) o_ck[k] = carve((size_t)(mc + 1) * 8);
  size_t o_colset[2][5], o_rowset[2][3];
  for (int k = 0; k < 2; k++) {
    for (int f = 0; f < 5; f++) o_colset[k][f] = carve((size_t)l * 8);
    for (int f = 0; f < 3; f++) o_rowset[k][f] = carve((size_t)(mc + 1) * 8);
  }
  const size_t pps = align_up((size_t)chain_ncb(l) + 1, 32), rps = align_up((size_t)chain_nrb(mc) + 1, 32);
  const size_t o_betab = carve((size_t)(mc + 1) * 8), o_ppart = carve(8 * pps * 8), o_rpart = carve(4 * rps * 8);
  const size_t o_zeros = carve((size_t)std::max(l, mc + 1) * 8 + 256);
  HIPCHECK(hipMalloc(&sc.scratch, off));
  HIPCHECK(hipMemsetAsync(sc.scratch, 0, off, sc.stream));
  unsigned char *b = (unsigned char *)sc.scratch;
  sc.d_colq = (double *)(b + o_colq);
  sc.d_srow = (double *)(b + o_srow);
  sc.d_cost1 = (double *)(b + o_cost1);
  sc.d_wts = (double *)(b + o_wts);
  sc.d_rcbase = (double *)(b + o_rcb);
  sc.d_gflag = (int *)(b + o_g);
  sc.d_part = (double *)(b + o_part);
  sc.d_colqx[0] = (double *)(b + o_cx0);
  sc.d_colqx[1] = (double *)(b + o_cx1);
  sc.d_betac[0] = (double *)(b + o_bc0);
  sc.d_betac[1] = (double *)(b + o_bc1);
  sc.d_pp[0] = (Cand *)(b + o_pp0);
  sc.d_pp[1] = (Cand *)(b + o_pp1);
  sc.d_rp = (Cand *)(b + o_rp);
  sc.d_olb = (double *)(b + o_olb);
  sc.d_oub = (double *)(b + o_oub);
  sc.d_dw = (double *)(b + o_dw);
  sc.d_dw2 = (double *)(b + o_dw2);
  sc.d_p1list = (int *)(b + o_p1l);
  sc.d_pw[0] = (double *)(b + o_pw0);
  sc.d_pw[1] = (double *)(b + o_pw1);
  sc.d_tflag = (int *)(b + o_tf);
  sc.d_rpc = (Cand *)(b + o_rpc);
  for (int k = 0; k < KCH; k++) {
    sc.d_srowk[k] = (double *)(b + o_sk[k]);
    sc.d_colqk[k] = (double *)(b + o_ck[k]);
  }
  for (int k = 0; k < 2; k++) {
    sc.d_drowk[k] = (double *)(b + o_colset[k][0]);
    sc.d_pwk[k] = (double *)(b + o_colset[k][1]);
    sc.d_nlbk[k] = (double *)(b + o_colset[k][2]);
    sc.d_nubk[k] = (double *)(b + o_colset[k][3]);
    sc.d_nflagk[k] = (int *)(b + o_colset[k][4]);
    sc.d_betak[k] = (double *)(b + o_rowset[k][0]);
    sc.d_blbk[k] = (double *)(b + o_rowset[k][1]);
    sc.d_bubk[k] = (double *)(b + o_rowset[k][2]);
  }
  sc.d_betab = (double *)(b + o_betab);
  sc.d_ppart = (double *)(b + o_ppart);
  sc.d_rpart = (double *)(b + o_rpart);
  sc.d_zeros = (double *)(b + o_zeros);
  sc.pp_stride = pps;
  sc.rp_stride = rps;
  sc.sk_stride = (o_sk[1] - o_sk[0]) / 8;
  sc.ck_stride = (o_ck[1] - o_ck[0]) / 8;
  sc.stage_bytes = stage_size(mc, l);
  HIPCHECK(hipHostMalloc((void **)&sc.h_stage, sc.stage_bytes));
  {
    static int zc = -1;
    if (zc < 0) {
      const char *e = std::getenv("MVX_ZC_STAGE");
      zc = e ? (std::atoi(e) != 0) : 1;
    }
    void *dp = nullptr;
    sc.stage_zc = zc && hipHostGetDevicePointer(&dp, sc.h_stage, 0) == hipSuccess && dp;
    if (sc.stage_zc) sc.d_stage = (unsigned char *)dp;
    else {
      (void)hipGetLastError();
      HIPCHECK(hipMalloc((void **)&sc.d_stage, sc.stage_bytes));
    }
  }
  sc.sc_m_cap = mc;
  sc.sc_ld = l;
}

// ------------------------------------------------------------------------------- slabs
struct SlabLayout {
  size_t o_T, o_bvar, o_blb, o_bub, o_nvar, o_nflag, o_nlb, o_nub, total;
};
static SlabLayout slab_layout(int m_cap, int ld) {
  SlabLayout L;
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  L.o_T = carve((size_t)(m_cap + 1) * ld * 8);
  L.o_bvar = carve((size_t)(m_cap + 1) * 4);
  L.o_blb = carve((size_t)(m_cap + 1) * 8);
  L.o_bub = carve((size_t)(m_cap + 1) * 8);
  L.o_nvar = carve((size_t)ld * 4);
  L.o_nflag = carve((size_t)ld * 4);
  L.o_nlb = carve((size_t)ld * 8);
  L.o_nub = carve((size_t)ld * 8);
  L.total = off;
  return L;
}

static void bind_slab(mvx_prob *P, void *slab, int m_cap, int ld) {
  SlabLayout L = slab_layout(m_cap, ld);
  unsigned char *b = (unsigned char *)slab;
  P->slab = slab;
  P->slab_bytes = L.total;
  P->m_cap = m_cap;
  P->ld = ld;
  P->d_T = (double *)(b + L.o_T);
  P->d_bvar = (int *)(b + L.o_bvar);
  P->d_blb = (double *)(b + L.o_blb);
  P->d_bub = (double *)(b + L.o_bub);
  P->d_nvar = (int *)(b + L.o_nvar);
  P->d_nflag = (int *)(b + L.o_nflag);
  P->d_nlb = (double *)(b + L.o_nlb);
  P->d_nub = (double *)(b + L.o_nub);
  // debugging aid: MVX_BIND_FILL="t,a" fills the tableau region with byte t and the small arrays with byte a at every
  // binding (-1: leave), on the main stream (the caller's upload or clone follows on it)
  static const char *bf = std::getenv("MVX_BIND_FILL");
  if (bf) {
    int t = -1, a = -1;
    std::sscanf(bf, "%d,%d", &t, &a);
    hipStream_t st = ctx().main.stream;
    if (t >= 0) HIPCHECK(hipMemsetAsync(b + L.o_T, t & 255, L.o_bvar - L.o_T, st));
    if (a >= 0) HIPCHECK(hipMemsetAsync(b + L.o_bvar, a & 255, L.total - L.o_bvar, st));
  }
}

static const size_t SLAB_CACHE_LIMIT = (size_t)32 << 30; // idle bytes kept for reuse (of 288 GB)

// caller holds the cache lock.  Give idle memory back to the driver: every idle slab that is not part of an arena, and
// every arena whose slabs are all idle; stops once the idle bytes are at or below `keep`.
static void slab_trim(SlabCache &sc, size_t keep) {
  if (sc.idle_bytes <= keep) return;
  (void)hipDeviceSynchronize(); // queued work may still read a slab that was recycled a moment ago
  for (auto it = sc.free_slabs.begin(); it != sc.free_slabs.end() && sc.idle_bytes > keep;) {
    if (sc.arena_of.count(it->second)) {
      ++it;
      continue;
    }
    (void)hipFree(it->second);
    sc.idle_bytes -= it->first;
    it = sc.free_slabs.erase(it);
  }
  for (size_t a = 0; a < sc.arenas.size() && sc.idle_bytes > keep; a++) {
    Arena &ar = sc.arenas[a];
    if (!ar.base || ar.idle != ar.count) continue;
    auto range = sc.free_slabs.equal_range(ar.slab_bytes);
    for (auto it = range.first; it != range.second;) {
      auto f = sc.arena_of.find(it->second);
      if (f != sc.arena_of.end() && f->second == (int)a) {
        sc.arena_of.erase(f);
        it = sc.free_slabs.erase(it);
      } else
        ++it;
    }
    (void)hipFree(ar.base);
    sc.idle_bytes -= ar.slab_bytes * (size_t)ar.count;
    ar.base = nullptr;
  }
}

// Fresh device memory is not zeroed by the driver (it holds whatever the last process left there), and a slab has
// regions no kernel of a solve writes before the first whole-slab clone reads them (spare rows for cuts, the padding of
// a row, array tails): they are never part of a result, but they travel with every clone, so they start as zeros.
// MVX_SLAB_FILL=<byte> fills with that byte instead (255: NaN patterns, to flush out a read of such a region).
static void slab_fill(void *p, size_t bytes) {
  static hipStream_t fill_stream = nullptr;
  static const int fill = std::getenv("MVX_SLAB_FILL") ? std::atoi(std::getenv("MVX_SLAB_FILL")) : 0;
  if (fill < 0) return;
  if (!fill_stream) HIPCHECK(hipStreamCreateWithFlags(&fill_stream, hipStreamNonBlocking));
  HIPCHECK(hipMemsetAsync(p, fill & 255, bytes, fill_stream));
  HIPCHECK(hipStreamSynchronize(fill_stream));
}

// nullptr when the device is out of memory even after the idle slabs have been given back (mvx_last_error() then
// reads MVX_ENOMEM); never aborts
static void *slab_alloc(Context &c, size_t bytes) {
  SlabCache &sc = c.slabs;
  std::lock_guard<std::mutex> lk(sc.mu);
  auto it = sc.free_slabs.find(bytes);
  if (it != sc.free_slabs.end()) {
    void *p = it->second;
    sc.free_slabs.erase(it);
    sc.idle_bytes -= bytes;
    auto f = sc.arena_of.find(p);
    if (f != sc.arena_of.end()) sc.arenas[(size_t)f->second].idle--;
    return p;
  }
  void *p = nullptr;
  if (bytes <= ((size_t)64 << 20)) {
    // A size class doubles each time it runs dry (32, 32, 64, 128, ... slabs, at most 16 GB at once): a B&B frontier of
    // a thousand 4 MB node tableaux is six allocations, not thirty.  hipMalloc beside two threads that launch and wait
    // was measured at 60-130 us per CLONE in arenas of 32 (milliseconds per call), against 12 us when the process is idle.
    const size_t least = std::min<size_t>(32, std::max<size_t>(2, ((size_t)256 << 20) / bytes));
    size_t have = 0;
    for (const Arena &ar : sc.arenas)
      if (ar.base && ar.slab_bytes == bytes) have += (size_t)ar.count;
    size_t count = std::min(std::max(least, have), std::max<size_t>(least, ((size_t)16 << 30) / bytes));
    if (hipMalloc(&p, bytes * count) != hipSuccess) {
      (void)hipGetLastError();
      count = least;
      p = nullptr;
      if (hipMalloc(&p, bytes * count) != hipSuccess) p = nullptr;
    }
    if (p) {
      slab_fill(p, bytes * count);
      Arena ar;
      ar.base = p;
      ar.slab_bytes = bytes;
      ar.count = (int)count;
      ar.idle = (int)count - 1;
      const int a = (int)sc.arenas.size();
      sc.arenas.push_back(ar);
      for (size_t k = 0; k < count; k++) {
        void *q = (unsigned char *)p + k * bytes;
        sc.arena_of.emplace(q, a);
        if (k) sc.free_slabs.emplace(bytes, q);
      }
      sc.idle_bytes += bytes * (count - 1);
      return p;
    }
    (void)hipGetLastError();
    p = nullptr;
  }
  if (hipMalloc(&p, bytes) == hipSuccess) {
    slab_fill(p, bytes);
    return p;
  }
  (void)hipGetLastError();
  slab_trim(sc, 0); // give everything idle back and retry once
  if (hipMalloc(&p, bytes) == hipSuccess) {
    slab_fill(p, bytes);
    return p;
  }
  (void)hipGetLastError();
  g_last_error.store(MVX_ENOMEM);
  return nullptr;
}

// back to the cache (the streams are in order: work already queued on the slab finishes before any reuse)
static void slab_recycle(Context &c, void *slab, size_t bytes) {
  SlabCache &sc = c.slabs;
  std::lock_guard<std::mutex> lk(sc.mu);
  auto f = sc.arena_of.find(slab);
  if (f != sc.arena_of.end()) sc.arenas[(size_t)f->second].idle++;
  sc.free_slabs.emplace(bytes, slab);
  sc.idle_bytes += bytes;
  if (sc.idle_bytes > SLAB_CACHE_LIMIT) slab_trim(sc, SLAB_CACHE_LIMIT / 2);
}

void release_device(mvx_prob *P) {
  if (!P->slab) return;
  Context &c = ctx();
  {
    MAIN_LOCK(c);
    // a recorded clone still reads from / writes to this slab: launch it before the slab can be handed out again
    const unsigned char *lo = (const unsigned char *)P->slab, *hi = lo + P->slab_bytes;
    for (const CopyJob &j : c.copies)
      if (((const unsigned char *)j.src >= lo && (const unsigned char *)j.src < hi) ||
          ((const unsigned char *)j.dst >= lo && (const unsigned char *)j.dst < hi)) {
        flush_copies(c);
        break;
      }
  }
  slab_recycle(c, P->slab, P->slab_bytes);
  P->slab = nullptr;
  P->slab_bytes = 0;
  P->d_T = nullptr;
  P->valid = false;
}

static bool alloc_device(mvx_prob *P, int m_cap, int ld) {
  Context &c = ctx();
  SlabLayout L = slab_layout(m_cap, ld);
  void *slab = slab_alloc(c, L.total);
  if (!slab) return false;
  bind_slab(P, slab, m_cap, ld);
  return true;
}

static int ld_for(int n) { return (int)align_up((size_t)n + 1, LD_ALIGN); }

// grow row capacity, preserving contents; false (handle left as it was) when the device is out of memory
static bool grow_rows(mvx_prob *P, int m_new) {
  if (m_new + ROW_SPARE <= P->m_cap) return true;
  Context &c = ctx();
  MAIN_LOCK(c);
  flush_copies(c);
  SolveCtx &sc = c.main;
  const int cap = m_new + ROW_SLACK;
  void *o_slab = P->slab;
  const size_t o_bytes = P->slab_bytes;
  const int o_rows = P->m_cap + 1, ld = P->ld;
  double *oT = P->d_T, *oblb = P->d_blb, *obub = P->d_bub, *onlb = P->d_nlb, *onub = P->d_nub;
  int *obvar = P->d_bvar, *onvar = P->d_nvar, *onflag = P->d_nflag;
  SlabLayout Ln = slab_layout(cap, ld);
  void *slab = slab_alloc(c, Ln.total);
  if (!slab) return false;
  bind_slab(P, slab, cap, ld);
  HIPCHECK(hipMemcpyAsync(P->d_T, oT, (size_t)o_rows * ld * 8, hipMemcpyDeviceToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_bvar, obvar, (size_t)o_rows * 4, hipMemcpyDeviceToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_blb, oblb, (size_t)o_rows * 8, hipMemcpyDeviceToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_bub, obub, (size_t)o_rows * 8, hipMemcpyDeviceToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nvar, onvar, (size_t)ld * 4, hipMemcpyDeviceToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nflag, onflag, (size_t)ld * 4, hipMemcpyDeviceToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nlb, onlb, (size_t)ld * 8, hipMemcpyDeviceToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nub, onub, (size_t)ld * 8, hipMemcpyDeviceToDevice, sc.stream));
  // recycle the old slab (in-order stream: the copies above complete before any reuse)
  slab_recycle(c, o_slab, o_bytes);
  return true;
}

// --------------------------------------------------------------------------- helpers
static inline double nb_value(int flag, double lb, double ub) {
  switch (flag) {
    case MVX_NL: return lb;
    case MVX_NU: return ub;
    case MVX_NS: return lb;
    default: return 0.0;
  }
}
static inline int std_flag(int type) {
  switch (type) {
    case MVX_FR: return MVX_NF;
    case MVX_LO: return MVX_NL;
    case MVX_UP: return MVX_NU;
    case MVX_DB: return MVX_NL;
    default: return MVX_NS;
  }
}

static void rebuild_pos(mvx_prob *P) {
  P->pos.assign((size_t)P->m + P->n + 1, 0);
  for (int i = 1; i <= P->m; i++) P->pos[P->bvar[i]] = i;
  for (int j = 1; j <= P->n; j++) P->pos[P->nvar[j]] = -j;
}

// variable-indexed bounds from the model
static inline void var_bounds(const mvx_prob *P, int k, double *lb, double *ub) {
  if (k <= P->m) {
    *lb = P->rlb[k];
    *ub = P->rub[k];
  } else {
    *lb = P->clb[k - P->m];
    *ub = P->cub[k - P->m];
  }
}

// Cluster selection (k_chain): one launch chooses a whole chain.  MVX_CLUSTER=0 keeps the two launches per step
// (k_pc / k_pr), which also take over for the rest of the process when a cluster launch ever gives up waiting for its
// peer workgroups (cl_abort), and serve the geometries k_chain does not cover (more than 16384 rows or columns).
static std::atomic<int> g_cluster{-1};
static std::atomic<bool> g_cluster_broken{false};
static std::atomic<long long> g_cluster_launches{0}, g_cluster_aborts{0};
static bool cluster_wanted() {
  int v = g_cluster.load();
  if (v < 0) {
    const char *e = std::getenv("MVX_CLUSTER");
    v = e ? (std::atoi(e) != 0) : 1;
    g_cluster.store(v);
  }
  return v != 0 && !g_cluster_broken.load();
}

// Steps per bulk launch of the chained primal path.  A step costs one small launch (k_fcs), a pass over the tableau
// costs its bytes: chains pay at every size, the longer the pass the longer the chain (scripts/chainsweep.py).
// MVX_CHAIN=1 turns the chaining off, 2..KCH fixes the length.
static int g_chain = -1;
static int chain_length(const mvx_prob *P) {
  if (g_chain < 0) {
    const char *e = std::getenv("MVX_CHAIN");
    g_chain = e ? std::max(1, std::min(KCH, std::atoi(e))) : 0;
  }
  if (g_chain > 0) return g_chain;
  const size_t bytes = (size_t)(P->m + 1) * (size_t)P->ld * 8;
  if (cluster_wanted() && chain_cluster_kmax(P->m, P->n) > 0) return bytes < ((size_t)24 << 20) ? 16 : 32; // a step costs no launch there
  if (bytes < ((size_t)24 << 20)) return 8;
  return 16;
}

// Dual pivots one k_update applies (dual_chain; MVX_DCHAIN=1 turns it off, 2..8 fixes the length).  The chain is built
// inside k_select at a few microseconds per step; what it saves is the k_update pass and the two launch boundaries of
// every step it absorbs.  Where the pass is the cost -- a launch shared by many node LPs, or one large tableau -- the
// longest chain pays (wide 512x1024 tree: 9.1 k nodes/s unchained, 14.8 k at 4, 18.9 k at 8); where a few small
// tableaux wait on each other's longest solve, 4 is the optimum (config-5 tree: 3.30 k, 4.40 k at 4, 4.22 k at 8).
static int g_dchain = -1;
static int dual_chain_length(bool wide, bool on_chip = false) {
  if (g_dchain < 0) {
    const char *e = std::getenv("MVX_DCHAIN");
    g_dchain = e ? std::max(1, std::min(DCH_MAX, std::atoi(e))) : 0;
  }
  if (g_dchain > 0) return g_dchain;
  // node LPs that k_dsel takes (up to 1024 x 1024): a chained step costs a few microseconds there, the longest chain pays
  // (config-5 tree: 5.5 k nodes/s at 4, 5.8 k at 6 and 8; wide tree 158 -> 140 ms per 2000 nodes)
  if (on_chip) return 8;
  return wide ? 8 : 4;
}

static void fill_ctl(SolveCtx &sc, mvx_prob *P, Ctl *h) {
  std::memset(h, 0, sizeof(Ctl));
  h->T = P->d_T;
  h->bvar = P->d_bvar; h->blb = P->d_blb; h->bub = P->d_bub;
  h->nvar = P->d_nvar; h->nflag = P->d_nflag; h->nlb = P->d_nlb; h->nub = P->d_nub;
  h->colq = sc.d_colq; h->srow = sc.d_srow; h->cost1 = sc.d_cost1; h->wts = sc.d_wts;
  h->gflag = sc.d_gflag; h->part = sc.d_part; h->rc_base = nullptr; h->rc_out = sc.d_cost1;
  h->m = P->m; h->n = P->n; h->ld = P->ld; h->m_cap = P->m_cap;
  h->sgn = (P->dir == MVX_MAX) ? 1.0 : -1.0;
  h->tol_bnd = 1e-9; h->tol_dj = 1e-9; h->tol_piv = 1e-9;
  h->phase = PH_START; h->done = D_RUN; h->budget = -1;
  h->stall = 0; h->stall_limit = g_stall_limit > 0 ? g_stall_limit : 64 + (P->m + P->n) / 8;
  h->olb = sc.d_olb; h->oub = sc.d_oub; h->dw = sc.d_dw;
  h->dwx[0] = sc.d_dw; h->dwx[1] = sc.d_dw2;
  h->pw[0] = sc.d_pw[0]; h->pw[1] = sc.d_pw[1];
  h->p1_list = sc.d_p1list;
  h->colqx[0] = sc.d_colqx[0]; h->colqx[1] = sc.d_colqx[1];
  h->betac[0] = sc.d_betac[0]; h->betac[1] = sc.d_betac[1];
  h->pp[0] = sc.d_pp[0]; h->pp[1] = sc.d_pp[1]; h->rp = sc.d_rp;
  h->npb = fused_npb(P->n); h->nrb = 0; // nrb is published by k_fb (its grid height)
  h->fstate = F_OFF;
  for (int k = 0; k < KCH; k++) {
    h->srowk[k] = sc.d_srowk[k];
    h->colqk[k] = sc.d_colqk[k];
  }
  h->pc_epoch = 1;
  {
    static unsigned long long *dbg = nullptr;
    static bool looked = false;
    if (!looked) {
      looked = true;
      if (std::getenv("MVX_FCS_DBG")) {
        HIPCHECK(hipMalloc((void **)&dbg, (size_t)KCH * 16 * 8));
        HIPCHECK(hipMemset(dbg, 0, (size_t)KCH * 16 * 8));
        HIPCHECK(hipDeviceSynchronize());
      }
    }
    h->dbg = dbg;
  }
  h->rpc = sc.d_rpc;
  h->chain_max = chain_length(P);
  h->dchain_max = dual_chain_length((size_t)(P->m + 1) * (size_t)P->ld * 8 >= ((size_t)16 << 20), P->m <= 1024 && P->n <= 1024);
  h->nch = 1;
}

// the handle's pending bound edits ride in the control block of the solve that is about to start
static void take_edits(mvx_prob *P, Ctl *h) {
  h->n_edits = (int)P->pending.size();
  for (int k = 0; k < h->n_edits; k++) {
    h->edit_row[k] = P->pending[(size_t)k].row;
    h->edit_lb[k] = P->pending[(size_t)k].lb;
    h->edit_ub[k] = P->pending[(size_t)k].ub;
  }
  P->pending.clear();
}

// ... or go to the device now (more than MAX_EDITS of them, or something other than a solve needs the bounds there)
static void flush_edits(SolveCtx &sc, mvx_prob *P) {
  for (const auto &e : P->pending) launch_set_basic_bounds(P->d_blb, P->d_bub, e.row, e.lb, e.ub, sc.stream);
  P->pending.clear();
}

static void upload_ctl(SolveCtx &sc) {
  HIPCHECK(hipMemcpyAsync(sc.d_ctl, sc.h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, sc.stream));
}

// copy the staging buffer back and refresh the host mirrors
static void pull_stage(SolveCtx &sc, mvx_prob *P, bool mirrors) {
  if (!sc.stage_zc) HIPCHECK(hipMemcpyAsync(sc.h_stage, sc.d_stage, stage_size(P->m_cap, P->ld), hipMemcpyDeviceToHost, sc.stream));
  HIPCHECK(hipStreamSynchronize(sc.stream));
  if (!mirrors) return;
  const unsigned char *s = sc.h_stage;
  const double *beta = (const double *)(s + sizeof(Ctl));
  const double *dj = beta + (P->m_cap + 1);
  const int *bv = (const int *)(dj + P->ld);
  const int *nv = bv + (P->m_cap + 1);
  const int *nf = nv + P->ld;
  P->beta.assign(beta, beta + P->m + 1);
  P->dj.assign(dj, dj + P->n + 1);
  P->bvar.assign(bv, bv + P->m + 1);
  P->nvar.assign(nv, nv + P->n + 1);
  P->nflag.assign(nf, nf + P->n + 1);
  rebuild_pos(P);
  P->sol_fresh = true;
  P->fresh_rows = -1;
}

void refresh_solution(const mvx_prob *Pc) {
  mvx_prob *P = const_cast<mvx_prob *>(Pc);
  if (!P->valid || P->sol_fresh) return;
  Context &c = ctx();
  MAIN_LOCK(c);
  flush_copies(c);
  SolveCtx &sc = c.main;
  ensure_scratch(sc, P->m_cap, P->ld);
  fill_ctl(sc, P, sc.h_ctl);
  upload_ctl(sc);
  launch_export(sc.d_ctl, sc.d_stage, P->m, P->n, 1, sc.stream);
  pull_stage(sc, P, true);
}

// ---------------------------------------------------------------------- tableau build
// false: the device is out of memory (mvx_last_error() reads MVX_ENOMEM); the handle stays without a tableau
static bool build_slack_tableau(mvx_prob *P, const int *sflag = nullptr) { // sflag: non-basic status per structural column (refresh)
  Context &c = ctx();
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  const int m = P->m, n = P->n;
  const int ld = ld_for(n);
  if (P->slab && (P->ld != ld || P->m_cap < m + ROW_SPARE)) release_device(P);
  if (!P->slab && !alloc_device(P, m + ROW_SLACK, ld)) {
    P->status = MVX_UNDEF;
    return false;
  }
  ensure_scratch(sc, P->m_cap, P->ld);
  P->bvar.assign((size_t)m + 1, 0);
  P->nvar.assign((size_t)n + 1, 0);
  P->nflag.assign((size_t)n + 1, 0);
  std::vector<double> nlb((size_t)ld, 0.0), nub((size_t)ld, 0.0), blb((size_t)m + 1, 0.0), bub((size_t)m + 1, 0.0);
  std::vector<double> xn((size_t)n + 1, 0.0);
  bool any_x = false;
  for (int j = 1; j <= n; j++) {
    P->nvar[j] = m + j;
    P->nflag[j] = sflag ? sflag[j] : std_flag(P->ctype[j]);
    nlb[j] = P->clb[j];
    nub[j] = P->cub[j];
    xn[j] = nb_value(P->nflag[j], nlb[j], nub[j]);
    any_x = any_x || xn[j] != 0.0;
  }
  for (int i = 1; i <= m; i++) {
    P->bvar[i] = i;
    blb[i] = P->rlb[i];
    bub[i] = P->rub[i];
  }
  // host tableau, uploaded row by row through a pinned bounce buffer
  const size_t rows_per_chunk = std::max<size_t>(1, ((size_t)32 << 20) / ((size_t)ld * 8));
  double *bounce = nullptr;
  HIPCHECK(hipHostMalloc((void **)&bounce, rows_per_chunk * ld * 8));
  for (size_t r0 = 0; r0 <= (size_t)m; r0 += rows_per_chunk) {
    size_t r1 = std::min<size_t>((size_t)m + 1, r0 + rows_per_chunk);
    for (size_t i = r0; i < r1; i++) {
      double *row = bounce + (i - r0) * ld;
      std::memset(row, 0, (size_t)ld * 8);
      if (i == 0) {
        double z = P->c[0];
        for (int j = 1; j <= n; j++) {
          row[j] = P->c[j];
          if (xn[j] != 0.0) z = std::fma(P->c[j], xn[j], z);
        }
        row[0] = z;
      } else {
        const double *a = P->A[i]->data();
        std::memcpy(row + 1, a + 1, (size_t)n * 8);
        double acc = 0.0;
        if (any_x)
          for (int j = 1; j <= n; j++)
            if (xn[j] != 0.0) acc = std::fma(a[j], xn[j], acc);
        row[0] = acc;
      }
    }
    HIPCHECK(hipMemcpyAsync(P->d_T + r0 * ld, bounce, (r1 - r0) * ld * 8, hipMemcpyHostToDevice, sc.stream));
    HIPCHECK(hipStreamSynchronize(sc.stream));
  }
  HIPCHECK(hipHostFree(bounce));
  HIPCHECK(hipMemcpyAsync(P->d_bvar, P->bvar.data(), (size_t)(m + 1) * 4, hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_blb, blb.data(), (size_t)(m + 1) * 8, hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_bub, bub.data(), (size_t)(m + 1) * 8, hipMemcpyHostToDevice, sc.stream));
  std::vector<int> nv((size_t)ld, 0), nf((size_t)ld, MVX_NS);
  for (int j = 1; j <= n; j++) {
    nv[j] = P->nvar[j];
    nf[j] = P->nflag[j];
  }
  HIPCHECK(hipMemcpyAsync(P->d_nvar, nv.data(), (size_t)ld * 4, hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nflag, nf.data(), (size_t)ld * 4, hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nlb, nlb.data(), (size_t)ld * 8, hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nub, nub.data(), (size_t)ld * 8, hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipStreamSynchronize(sc.stream));
  rebuild_pos(P);
  P->pending.clear(); // the bounds just uploaded are the model's
  P->valid = true;
  P->sol_fresh = false;
  P->fresh_rows = -1;
  P->status = MVX_UNDEF;
  return true;
}

// ------------------------------------------------------------------------------ simplex
static void flush_update_events(Context &c, size_t used) {
  for (size_t k = 0; k + 1 < used; k += 2) {
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, c.ev_pool[k], c.ev_pool[k + 1]));
    c.prof_update_ms += ms;
    c.prof_update_n++;
  }
}

// A solve is a small host-side state machine around queued launches, so that several of them
// can be in flight on different streams (engine_simplex_batch): begin -> {enqueue, sync, collect}*.
// ---- resident-tableau path (k_persist): which problems take it, its buffers, backup and restore
constexpr int PERSIST_HEAD_STRIDE_MAX = 4; // granules between two strips' heads
static std::atomic<int> g_persist_mode{-1}; // -1: environment MVX_PERSIST (default on), 0 off, 1 on
static std::atomic<bool> g_persist_broken{false}; // a launch aborted (its workgroups were not co-resident in time): off for good
static std::atomic<long long> g_persist_launches{0}, g_persist_aborts{0};
struct PersistPlan {
  int cpw = 0, nw = 0;
  size_t lds = 0;
};
static bool persist_plan(Context &c, const mvx_prob *P, PersistPlan *pl) {
  if (g_persist_broken) return false;
  if (g_persist_mode < 0) {
    const char *e = std::getenv("MVX_PERSIST");
    g_persist_mode = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1);
  }
  if (!g_persist_mode) return false;
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.dev) != hipSuccess) return false;
    cus = prop.multiProcessorCount;
  }
  const int m = P->m, n = P->n;
  if ((long)(m + 1) * (n + 1) < 32768) return false; // a handful of pivots: the launch is not worth its set-up
  // Measured (profiles/r02_persist_ab.jsonl): 256x512 29.6 -> 10.6 us/pivot, 512x1024 16.0 -> 12.1, 1024x2048 15.9 -> 15.6,
  // 1024x4096 20.3 -> 20.2: the per-pivot gather of the 256 proposals costs ~6.5 us whatever the size (3.6 polling sweeps
  // of ~1.8 us each), the strip work grows with m.  MVX_PERSIST=2 lifts the size cap for experiments.
  if (g_persist_mode < 2 && (long)(m + 1) * (n + 1) > 700000) return false;
  const int cpw = (n + cus - 1) / cus;
  if (cpw > persist_max_cpw()) return false;
  pl->cpw = cpw;
  pl->nw = (n + cpw - 1) / cpw; // one workgroup per CU at most: every workgroup must be resident at once
  if (pl->nw > 256) return false; // the exchange buffers are sized for 256 workgroups
  pl->lds = persist_lds_bytes(m, cpw);
  return pl->lds <= (size_t)150 * 1024;
}

static bool ensure_persist(Context &c, SolveCtx &sc, const mvx_prob *P, const PersistPlan &pl) {
  const int words = persist_slot_words(P->m_cap);
  if (!sc.d_pabort) {
    HIPCHECK(hipMalloc((void **)&sc.d_pabort, 1024));
    HIPCHECK(hipMemsetAsync(sc.d_pabort, 0, 1024, sc.stream));
    HIPCHECK(hipHostMalloc((void **)&sc.h_pabort, 1024));
    sc.h_pabort[0] = 0;
    HIPCHECK(hipMalloc((void **)&sc.d_pcand, (size_t)2 * 256 * PERSIST_HEAD_STRIDE_MAX * 8));
    HIPCHECK(hipMalloc((void **)&sc.d_pctl, sizeof(Ctl)));
  }
  if (words > sc.p_msg_words) {
    HIPCHECK(hipStreamSynchronize(sc.stream));
    if (sc.d_pmsg) HIPCHECK(hipFree(sc.d_pmsg));
    HIPCHECK(hipMalloc((void **)&sc.d_pmsg, (size_t)2 * 256 * words * 8)); // a slot per workgroup and parity
    sc.p_msg_words = words;
  }
  if (P->ld > sc.p_pw_ld) {
    HIPCHECK(hipStreamSynchronize(sc.stream));
    if (sc.d_ppw) HIPCHECK(hipFree(sc.d_ppw));
    HIPCHECK(hipMalloc((void **)&sc.d_ppw, (size_t)2 * P->ld * 8));
    sc.p_pw_ld = P->ld;
  }
  if (sc.p_backup_bytes != P->slab_bytes) {
    if (sc.p_backup) slab_recycle(c, sc.p_backup, sc.p_backup_bytes);
    sc.p_backup = slab_alloc(c, P->slab_bytes);
    sc.p_backup_bytes = sc.p_backup ? P->slab_bytes : 0;
    if (!sc.p_backup) {
      (void)g_last_error.exchange(0); // not an error of the caller's: the two-kernel path serves
      return false;
    }
  }
  (void)pl;
  return true;
}

struct SolveJob {
  mvx_prob *P = nullptr;
  SolveCtx *sc = nullptr;
  mvx_smcp parm;
  enum { MAIN, PHASE1, FINAL } mode = MAIN;
  int batch = 8, pb = 4;
  int done = D_RUN;
  bool try_fused = false;
  bool try_dfused = false;    // dual phase on a large tableau: k_dboot / k_da / k_fb<DUAL>
  bool persist_queued = false; // this batch of launches contains a k_persist launch (its abort flag is copied back)
  int seen_steps = 0, seen_pivots = 0, seen_bulk = 0;
  int chain = 1, chain0 = 1; // pivots per bulk launch the next batch is queued for / the size rule's choice
  bool resident = false;     // this solve holds the token that lets it launch kernels whose workgroups must be resident together
  bool cluster = false;      // the chains of this solve are chosen by k_chain (one launch) instead of k_pc / k_pr per step
  ChainArgs cargs{};         // what the kernels of the chained primal path take by value
  size_t ev_used = 0;
  bool profiled = false;
  int rc = 0;
  Ctl snap;
};

// No limit asked for: a safety cap stands in (belt and braces behind the anti-cycling rule; a stalled
// degenerate LP must end with EITLIM rather than spin on the device).  Same formula as the oracle.
static int pivot_budget(const mvx_prob *P, const mvx_smcp &parm) {
  return parm.it_lim >= 0 ? parm.it_lim : 200 * (P->m + P->n) + 100000;
}

static void stage_copy_async(SolveCtx &sc, const mvx_prob *P) {
  HIPCHECK(hipEventRecord(sc.ev_b, sc.stream)); // end of the device work queued so far (last_solve_ms)
  if (!sc.stage_zc) HIPCHECK(hipMemcpyAsync(sc.h_stage, sc.d_stage, stage_size(P->m_cap, P->ld), hipMemcpyDeviceToHost, sc.stream));
  if (sc.d_pabort) HIPCHECK(hipMemcpyAsync(sc.h_pabort, sc.d_pabort, 512, hipMemcpyDeviceToHost, sc.stream)); // abort flag + phase cycle totals
}

// The fused dual pipeline replaces k_select's four dependent stages (34 us at 4096x8192) by k_da (~8 us); below about
// two million tableau entries the update itself is so short that the two extra bootstrap launches do not pay.
static bool dual_fused_worth_it(const mvx_prob *P) {
  static int mode = -1; // MVX_DUAL_FUSED=0 off, 1 size rule (default), 2 always
  if (mode < 0) {
    const char *e = std::getenv("MVX_DUAL_FUSED");
    mode = e ? std::atoi(e) : 1;
  }
  if (mode == 0) return false;
  if (mode >= 2) return true;
  // ... and from about twelve million entries on the generic path with its dual pivots chained eight to a pass is the
  // faster one (4096x8192: 67.7 against 96.5 us per dual pivot; 2048x4096: 40.4 against 39.9, scripts/dualtime.py)
  const long entries = (long)(P->m + 1) * (P->n + 1);
  return entries >= 2000000 && entries < 12000000;
}

static void job_begin(Context &c, SolveJob &J) {
  mvx_prob *P = J.P;
  SolveCtx &sc = *J.sc;
  ensure_scratch(sc, P->m_cap, P->ld);
  Ctl *h = sc.h_ctl;
  fill_ctl(sc, P, h);
  h->tol_bnd = J.parm.tol_bnd;
  h->tol_dj = J.parm.tol_dj;
  h->tol_piv = J.parm.tol_piv;
  h->budget = pivot_budget(P, J.parm);
  take_edits(P, h);
  upload_ctl(sc);
  HIPCHECK(hipEventRecord(sc.ev_a, sc.stream));
  {
    ChainArgs &a = J.cargs;
    a.c = sc.d_ctl;
    a.T = P->d_T;
    a.blb = P->d_blb; a.bub = P->d_bub; a.nlb = P->d_nlb; a.nub = P->d_nub; a.nflag = P->d_nflag;
    a.betab = sc.d_betab;
    for (int k = 0; k < 2; k++) {
      a.pw[k] = sc.d_pw[k];
      a.drowk[k] = sc.d_drowk[k]; a.pwk[k] = sc.d_pwk[k]; a.nlbk[k] = sc.d_nlbk[k]; a.nubk[k] = sc.d_nubk[k]; a.nflagk[k] = sc.d_nflagk[k];
      a.betak[k] = sc.d_betak[k]; a.blbk[k] = sc.d_blbk[k]; a.bubk[k] = sc.d_bubk[k];
    }
    a.pp = sc.d_ppart; a.rp = sc.d_rpart; a.ppstride = sc.pp_stride; a.rpstride = sc.rp_stride;
    a.srow0 = sc.d_srowk[0]; a.colq0 = sc.d_colqk[0];
    a.sstride = sc.sk_stride; a.cstride = sc.ck_stride;
    a.m = P->m; a.n = P->n; a.ld = P->ld; a.mcap1 = P->m_cap + 1;
    a.ncb = chain_ncb(P->n); a.nrb = chain_nrb(P->m);
    a.tol_dj = h->tol_dj; a.tol_piv = h->tol_piv; a.tol_bnd = h->tol_bnd; a.sgn = h->sgn;
    a.stall_limit = h->stall_limit;
    a.zeros = sc.d_zeros;
    a.xg = sc.d_xg; a.xg_bytes = (int)XG_BYTES; a.xabort = sc.d_xabort;
    a.nw = chain_cluster_nw(P->m, P->n); a.kmax = 0; a.tagbase = 0; a.boot = 0;
  }
  // k_chain and k_persist need their workgroups resident together: two such launches from two host threads (the main
  // context and a B&B worker's) could each hold half of the CUs they both need and wait for the rest until they time
  // out.  One solve at a time may use them (solve_once holds the token for the length of the call); a solve that finds
  // the token taken runs on the plain kernels.
  J.cluster = J.resident && cluster_wanted() && chain_cluster_kmax(P->m, P->n) > 0;
  J.try_fused = !P->hint_dual; // dual-phase warm starts (B&B children) skip the primal fast path
  J.chain = J.chain0 = h->chain_max;
  J.try_dfused = P->hint_dual && dual_fused_worth_it(P);
  J.profiled = c.prof && J.sc == &c.main;
  // With a pivot limit the number of pivots wanted is known: queue them in one go (up to 256) instead of
  // growing the batch 8, 16, 32, ... -- every batch boundary costs a host round trip and a generic
  // (one-workgroup) selection step.  A primal fast-path solve that ends before the limit leaves no-op
  // launches of ~2 us behind; dual warm starts keep the cautious growth (they usually end after a few pivots).
  if (J.parm.it_lim > 8 && J.try_fused) J.batch = std::min(J.parm.it_lim, 256);
}

static void job_enqueue(Context &c, SolveJob &J) {
  mvx_prob *P = J.P;
  SolveCtx &sc = *J.sc;
  const int m = P->m, n = P->n;
  J.ev_used = 0;
  if (J.mode == SolveJob::FINAL) {
    launch_export(sc.d_ctl, sc.d_stage, m, n, 1, sc.stream); // forced export (state left by a host-side decision)
    stage_copy_async(sc, P);
    return;
  }
  if (J.mode == SolveJob::PHASE1) {
    // host-driven phase 1: per iteration head (signs, changed rows) -> cost-row fix -> select -> update;
    // the cost row is tableau row m+1, so the update grid reaches one row further
    for (int k = 0; k < J.pb; k++) {
      launch_p1_head(sc.d_ctl, sc.stream);
      launch_p1_fix(sc.d_ctl, n, sc.stream);
      launch_p1_select(sc.d_ctl, sc.stream);
      launch_update(sc.d_ctl, m + 1, n, sc.stream);
    }
    launch_export(sc.d_ctl, sc.d_stage, m, n, 0, sc.stream);
    stage_copy_async(sc, P);
    return;
  }
  const int batch = J.batch;
  if (J.profiled && c.ev_pool.size() < (size_t)2 * batch + 4) { // sized for the largest batch
    size_t old = c.ev_pool.size();
    c.ev_pool.resize((size_t)2 * batch + 4);
    for (size_t k = old; k < c.ev_pool.size(); k++) HIPCHECK(hipEventCreate(&c.ev_pool[k]));
  }
  // with a pivot limit, never queue more pivots than the limit still allows (+1 launch so that
  // k_select can observe the exhausted budget): keeps no-op launches out of profiles
  const int remaining = (J.parm.it_lim >= 0) ? std::max(0, J.parm.it_lim - J.seen_pivots) : (1 << 30);
  auto ev = [&]() {
    if (J.profiled) HIPCHECK(hipEventRecord(c.ev_pool[J.ev_used++], sc.stream));
  };
  // (replaying each batch as one captured hipGraph was measured: no gain -- the dispatch of small kernels is
  // bound by the command processor, not by the host -- and removed)
  int depth;
  if (J.try_fused) depth = std::max(0, std::min(batch, remaining)); // + the generic step that closes the batch, if the limit leaves it a pivot
  else depth = std::max(1, std::min(batch, remaining));
  auto body = [&](int m_grid) {
    if (J.try_fused) {
      PersistPlan pl;
      // the resident-tableau kernel serves where the cluster chain is not on offer (mvx_set_cluster(0), a cluster launch
      // that gave up): since round 3 the chain is the faster one at every size that fits k_persist (scripts/smalltime.py:
      // 512x1024 9.1 against 12.3 us per pivot, 128x256 9.7 against 10.8)
      if (depth > 0 && !J.profiled && J.resident && !(J.cluster && !g_cluster_broken.load()) && persist_plan(c, P, &pl) && ensure_persist(c, sc, P, pl)) {
        // cache-resident size: one generic step settles the phase, then the whole run of primal pivots in ONE launch
        // with the tableau held in LDS.  In front of it a backup (slab, control block, both devex weight sets): a
        // launch whose workgroups do not all become resident in time aborts mid-step, and the backup is what the
        // two-kernel path then carries on from.
        launch_select(sc.d_ctl, sc.stream);
        launch_update(sc.d_ctl, m_grid, n, sc.stream, 1, 1);
        HIPCHECK(hipMemcpyAsync(sc.p_backup, P->slab, P->slab_bytes, hipMemcpyDeviceToDevice, sc.stream));
        HIPCHECK(hipMemcpyAsync(sc.d_pctl, sc.d_ctl, sizeof(Ctl), hipMemcpyDeviceToDevice, sc.stream));
        HIPCHECK(hipMemcpyAsync(sc.d_ppw, sc.d_pw[0], (size_t)P->ld * 8, hipMemcpyDeviceToDevice, sc.stream));
        HIPCHECK(hipMemcpyAsync(sc.d_ppw + P->ld, sc.d_pw[1], (size_t)P->ld * 8, hipMemcpyDeviceToDevice, sc.stream));
        const int head_stride = 4; // 4 B .. 4 KB between heads measured the same
        HIPCHECK(hipMemsetAsync(sc.d_pcand, 0, (size_t)2 * pl.nw * head_stride * 8, sc.stream));
        const int steps = 1 << 24; // the pivot limit is the control block's `budget`, which the kernel counts down
        if (launch_persist(sc.d_ctl, sc.d_pcand, sc.d_pmsg, sc.d_pabort, (unsigned long long *)(sc.d_pabort + 16), P->m, pl.cpw, pl.nw, sc.p_msg_words, steps, head_stride, sc.stream) == 0) {
          J.persist_queued = true;
          g_persist_launches++;
          // what the run ended on (optimum, unbounded ray, stall, pivot limit) is settled by one generic step
          launch_select(sc.d_ctl, sc.stream);
          launch_update(sc.d_ctl, m_grid, n, sc.stream, 1, 1);
        } else {
          g_persist_broken = true;
        }
      } else {
        // The fused two-kernel pipeline (k_fa / k_fb) first: k_fboot takes a call from its very first pivot when the
        // basis it starts from is primal feasible, and otherwise (dual simplex, phase 1, Bland's rule in force) turns
        // the pipeline off, so that its launches return at once.  One generic step closes the batch: it settles what
        // a stopped run ended on (optimum, unbounded ray, stall) in the same host round trip, and is an ordinary
        // pivot when nothing stopped.  Its events come first in the pool: the leading pairs are the ones that stepped.
        const size_t e_generic = J.ev_used;
        if (J.profiled) J.ev_used += 2;
        if (depth > 0) {
          // chained path: k_pboot once, then per chain of up to `kc` steps two small launches per step (k_pc / k_pr)
          // and one bulk launch (k_fbc3), so `depth` pivots take depth / kc passes over the tableau when every chain
          // fills (a chain that ends early leaves pivots for the next batch)
          const bool cl = J.cluster && !g_cluster_broken.load();
          if (!cl) launch_pboot(J.cargs, sc.stream); // k_chain does the start-of-batch checks itself (first launch: boot)
          bool first = true;
          const int kc = cl ? std::max(1, std::min(J.chain0, chain_cluster_kmax(m, n))) : std::max(1, J.chain);
          auto chain_launch = [&](int steps) { // k_chain: the whole selection of a chain of up to `steps` steps
            J.cargs.kmax = steps;
            J.cargs.boot = first ? 1 : 0;
            first = false;
            J.cargs.tagbase = sc.xtag;
            sc.xtag += 128;
            if (launch_chain(J.cargs, sc.stream) != 0) g_cluster_broken.store(true);
            g_cluster_launches++;
          };
          for (int left = depth; left > 0;) {
            const int steps = std::min(kc, left); // the last pass of a limited run chains only what the limit leaves
            if (cl) chain_launch(steps);
            else
              for (int t = 0; t < steps; t++) launch_pstep(J.cargs, t, sc.stream);
            ev(); // the profiled pair of events brackets the bulk pass alone
            launch_fbc3(J.cargs, steps, sc.stream);
            ev();
            launch_fpatch(J.cargs, steps, sc.stream);
            left -= steps;
          }
          // a run that may end on the pivot limit: one more selection, which finds the limit and reports it
          // (k_chain notes it itself when the limit falls on the end of its last chain: pc_itlim)
          if (J.parm.it_lim >= 0 && depth >= remaining && !cl) launch_pc(J.cargs, 0, sc.stream);
        }
        launch_select(sc.d_ctl, sc.stream);
        if (J.profiled) HIPCHECK(hipEventRecord(c.ev_pool[e_generic], sc.stream));
        launch_update(sc.d_ctl, m_grid, n, sc.stream, 1, 1);
        if (J.profiled) HIPCHECK(hipEventRecord(c.ev_pool[e_generic + 1], sc.stream));
      }
    } else if (J.try_dfused) {
      // one generic step settles the phase (and restarts the dual devex weights when the phase has just been
      // entered); if it is the dual simplex, the fused pair k_da / k_fb<DUAL> takes over, otherwise its launches
      // return at once
      launch_select(sc.d_ctl, sc.stream);
      launch_update(sc.d_ctl, m_grid, n, sc.stream, 1, 1);
      if (depth > 1) {
        launch_dboot(sc.d_ctl, n, sc.stream);
        launch_db(sc.d_ctl, m_grid, n, sc.stream);
        for (int k = 0; k < depth - 1; k++) {
          launch_da(sc.d_ctl, n, sc.stream);
          launch_db(sc.d_ctl, m_grid, n, sc.stream);
        }
      }
    } else {
      for (int k = 0; k < depth; k++) {
        launch_dsel(sc.d_ctl, m, n, sc.stream); // a dual phase carrying on (warm starts): the chain on chip, k_select then returns
        launch_select(sc.d_ctl, sc.stream);
        ev();
        launch_update(sc.d_ctl, m_grid, n, sc.stream, 1, 1);
        ev();
      }
    }
    launch_export(sc.d_ctl, sc.d_stage, m_grid, n, 0, sc.stream);
    stage_copy_async(sc, P);
  };
  body(m);
}

// The staging buffer holds the control block and, because the solve has ended (k_export packs the
// mirrors whenever done != RUN), the basis / solution mirrors: publish them on the handle.
static bool job_finalize(SolveJob &J) {
  mvx_prob *P = J.P;
  SolveCtx &sc = *J.sc;
  const Ctl &snap = J.snap;
  const unsigned char *s = sc.h_stage;
  const double *beta = (const double *)(s + sizeof(Ctl));
  const double *dj = beta + (P->m_cap + 1);
  const int *bv = (const int *)(dj + P->ld);
  const int *nv = bv + (P->m_cap + 1);
  const int *nf = nv + P->ld;
  P->beta.assign(beta, beta + P->m + 1);
  P->dj.assign(dj, dj + P->n + 1);
  P->bvar.assign(bv, bv + P->m + 1);
  P->nvar.assign(nv, nv + P->n + 1);
  P->nflag.assign(nf, nf + P->n + 1);
  rebuild_pos(P);
  P->sol_fresh = true;
  P->fresh_rows = -1;
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, sc.ev_a, sc.ev_b));
  P->last_ms = ms;
  P->it_cnt += snap.it_cnt;
  P->bland_cnt += snap.n_bland;
  P->pert_cnt += snap.n_pert;
  P->hint_dual = false;
  switch (J.done) {
    case D_OPT: P->status = MVX_OPT; J.rc = 0; break;
    case D_UNBND: P->status = MVX_UNBND; J.rc = 0; break;
    case D_NOFEAS: P->status = MVX_NOFEAS; J.rc = 0; break;
    case D_ITLIM:
      P->status = (snap.phase == PH_PRIMAL2) ? MVX_FEAS : MVX_INFEAS;
      J.rc = MVX_EITLIM;
      break;
    default: P->status = MVX_UNDEF; J.rc = MVX_EFAIL; break;
  }
  return true;
}

// the job's stream has been synchronised; returns true when the solve is complete
static bool job_collect(Context &c, SolveJob &J) {
  SolveCtx &sc = *J.sc;
  Ctl &snap = J.snap;
  if (J.persist_queued) {
    J.persist_queued = false;
    if (sc.h_pabort[0]) {
      // the resident-tableau launch gave up waiting for its peers: its workgroups may have stopped one step apart,
      // so the tableau is put back as it was in front of the launch and the two-kernel path carries on from there
      mvx_prob *P = J.P;
      g_persist_broken = true;
      g_persist_aborts++;
      std::fprintf(stderr, "mvx: resident-tableau launch aborted (workgroups not co-resident in time); using the two-kernel path from now on\n");
      HIPCHECK(hipMemcpyAsync(P->slab, sc.p_backup, P->slab_bytes, hipMemcpyDeviceToDevice, sc.stream));
      HIPCHECK(hipMemcpyAsync(sc.d_ctl, sc.d_pctl, sizeof(Ctl), hipMemcpyDeviceToDevice, sc.stream));
      HIPCHECK(hipMemcpyAsync(sc.d_pw[0], sc.d_ppw, (size_t)P->ld * 8, hipMemcpyDeviceToDevice, sc.stream));
      HIPCHECK(hipMemcpyAsync(sc.d_pw[1], sc.d_ppw + P->ld, (size_t)P->ld * 8, hipMemcpyDeviceToDevice, sc.stream));
      HIPCHECK(hipMemsetAsync(sc.d_pabort, 0, sizeof(int), sc.stream));
      sc.h_pabort[0] = 0;
      J.try_fused = true; // the backup was taken right after a generic step in primal phase 2
      return false;
    }
  }
  std::memcpy(&snap, sc.h_stage, sizeof(Ctl));
  if (J.mode == SolveJob::FINAL) return job_finalize(J);
  if (J.mode == SolveJob::PHASE1) {
    if (snap.done == D_RUN) {
      J.pb = std::min(J.pb * 2, 64);
      return false;
    }
    if (snap.done == D_PFEAS) {
      snap.rounds++;
      if (snap.rounds >= 64) {
        J.done = D_FAIL;
        J.mode = SolveJob::FINAL;
        return false;
      }
      snap.done = D_RUN;
      snap.phase = PH_START;
      *sc.h_ctl = snap;
      upload_ctl(sc);
      J.mode = SolveJob::MAIN;
      J.batch = 8;
      J.try_fused = false;
      J.try_dfused = false;
      return false;
    }
    J.done = snap.done;
    return job_finalize(J); // the same batch's export already packed the mirrors
  }
  if (J.profiled) {
    // once the solve finishes inside a batch the queued-ahead launches are no-ops; only the
    // leading launches that really stepped are timed
    const int steps_now = snap.n_bulk; // launches that stepped: one per pivot or flip, one per chain of pivots
    flush_update_events(c, std::min(J.ev_used, (size_t)2 * (size_t)(steps_now - J.seen_steps)));
    J.seen_steps = steps_now;
  }
  if (snap.cl_abort) {
    // a cluster launch gave up waiting for its peers: it changed nothing, the rest of the batch returned at once and
    // the closing generic step carried on; from here on the chains are chosen by k_pc / k_pr
    g_cluster_broken.store(true);
    g_cluster_aborts++;
    std::fprintf(stderr, "mvx: cluster selection launch aborted (peer workgroups not reachable in time); using two launches per chained step from now on\n");
    HIPCHECK(hipMemsetAsync(sc.d_xabort, 0, sizeof(int), sc.stream));
  }
  J.done = snap.done;
  {
    // chains that keep ending early (bound flips, degenerate stretches, the end of the solve) leave their queued
    // selection launches as no-ops: follow the length the last batch actually reached
    const int dp = snap.it_cnt - J.seen_pivots, db = snap.n_bulk - J.seen_bulk;
    if (J.chain0 > 1 && db >= 4) {
      if (2 * dp < J.chain * db) J.chain = std::max(1, J.chain / 2);
      else if (10 * dp >= 9 * J.chain * db) J.chain = std::min(J.chain0, J.chain * 2);
    }
    J.seen_bulk = snap.n_bulk;
  }
  J.seen_pivots = snap.it_cnt;
  J.try_fused = (snap.phase == PH_PRIMAL2) && snap.stall < snap.stall_limit; // the fused path prices by Dantzig only
  J.try_dfused = (snap.phase == PH_DUAL) && snap.stall < snap.stall_limit && dual_fused_worth_it(J.P);
  if (J.done == D_NEED_PHASE1) {
    snap.done = D_RUN;
    snap.phase = PH_PHASE1;
    snap.p1_init = 1; // k_p1_head zeroes the cost row (tableau row m+1) and the signs
    snap.p1_fix_q = 0;
    *sc.h_ctl = snap;
    upload_ctl(sc);
    J.mode = SolveJob::PHASE1;
    J.pb = 4;
    return false;
  }
  if (J.done != D_RUN) return job_finalize(J); // the same batch's export already packed the mirrors
  J.batch = std::min(J.batch * 2, 256);
  return false;
}

// A handle whose last solve ended OPT and that has not been edited since (every edit resets `status`)
// would go through zero pivots and come out unchanged: bs.cpp:116-117 re-solves exactly such clones at
// every pop.  Answer from the state at hand instead of queueing launches.  NOFEAS / UNBND are not in the
// list: a new call restarts the devex weights, so the re-solve may pick another infeasible row (another
// entering column) than the one that proved infeasibility (unboundedness) and pivot on before it ends there again.
static bool already_solved(const mvx_prob *P, const mvx_smcp &parm) {
  if (!P->valid || parm.it_lim == 0) return false;
  if (P->status != MVX_OPT) return false;
  return P->last_tol[0] == parm.tol_bnd && P->last_tol[1] == parm.tol_dj && P->last_tol[2] == parm.tol_piv;
}
static void remember_tolerances(mvx_prob *P, const mvx_smcp &parm) {
  P->last_tol[0] = parm.tol_bnd;
  P->last_tol[1] = parm.tol_dj;
  P->last_tol[2] = parm.tol_piv;
}

static bool job_prepare(SolveJob &J, mvx_prob *P, const mvx_smcp *parm) {
  J.P = P;
  if (parm) J.parm = *parm;
  else mvx_init_smcp(&J.parm);
  if (P->m < 1 || P->n < 1) {
    P->status = MVX_UNDEF;
    J.rc = MVX_EFAIL;
    return false;
  }
  if (already_solved(P, J.parm)) {
    P->last_ms = 0.0;
    J.rc = 0;
    return false;
  }
  if (!P->valid && !build_slack_tableau(P)) {
    J.rc = MVX_EFAIL;
    return false;
  }
  remember_tolerances(P, J.parm);
  return true;
}

static int solve_once(mvx_prob *P, const mvx_smcp *parm, bool aux) {
  SolveJob J;
  if (!job_prepare(J, P, parm)) return J.rc;
  P->fresh_rows = -1; // the solve rewrites the tableau: whatever ends it exports the mirrors anew, or leaves them stale
  Context &c = ctx();
  HIPCHECK(hipSetDevice(c.dev)); // the current device is per host thread (the B&B driver solves from a worker thread)
  std::unique_lock<std::recursive_mutex> main_lock(c.main_mu);
  flush_copies(c);
  if (aux) main_lock.unlock();
  std::unique_lock<std::mutex> aux_lock(c.aux_mu, std::defer_lock);
  if (aux) aux_lock.lock();
  if (aux && !c.aux_ready) {
    init_solve_ctx(c.aux);
    c.aux_ready = true;
  }
  J.sc = aux ? &c.aux : &c.main;
  static std::atomic<bool> g_resident_token{false};
  bool expected = false;
  J.resident = g_resident_token.compare_exchange_strong(expected, true);
  job_begin(c, J);
  for (;;) {
    job_enqueue(c, J);
    HIPCHECK(hipStreamSynchronize(J.sc->stream));
    if (job_collect(c, J)) break;
  }
  if (J.resident) g_resident_token.store(false);
  return J.rc;
}

// ---- tableau refresh (same rule and arithmetic as the oracle's refresh_tableau / orc_row_residual)
static int g_check_every = 1024;
static double g_refresh_tol = 1e-9;
void set_refresh(int check_every, double tol) {
  g_check_every = check_every > 0 ? check_every : 1024;
  g_refresh_tol = tol >= 0.0 ? tol : 1e-9;
}

constexpr int REFRESH_SAMPLE_ROWS = 32; // rows the look that decides on a refresh reads (oracle: REFRESH_SAMPLE_ROWS)
static double row_residual_sample(const mvx_prob *P, int rows);
double row_residual(const mvx_prob *P) { return row_residual_sample(P, 0); }

// residual over `rows` rows picked from the pivot count (rows <= 0 or >= m: every row)
static double row_residual_sample(const mvx_prob *P, int rows) {
  if (!P->valid) return 0.0;
  refresh_solution(P);
  const int m = P->m, n = P->n;
  auto value = [&](int k) {
    const int pos = P->pos[(size_t)k];
    if (pos > 0) return P->beta[(size_t)pos];
    double lb, ub;
    var_bounds(P, k, &lb, &ub);
    return nb_value(P->nflag[(size_t)-pos], lb, ub);
  };
  std::vector<double> x((size_t)n + 1, 0.0);
  for (int j = 1; j <= n; j++) x[(size_t)j] = value(m + j);
  double worst = 0.0;
  const bool all = rows <= 0 || rows >= m;
  const int cnt = all ? m : rows;
  for (int k = 0; k < cnt; k++) {
    const int i = all ? k + 1 : 1 + (int)(((unsigned)P->it_cnt * 2654435761u + (unsigned)k * 0x9E3779B1u) % (unsigned)m);
    const double *a = P->A[(size_t)i]->data();
    double acc = 0.0;
    for (int j = 1; j <= n; j++) acc = acc + a[j] * x[(size_t)j];
    const double xr = value(i);
    const double r = std::fabs(acc - xr) / (1.0 + std::fabs(xr));
    if (r > worst) worst = r;
  }
  return worst;
}

static bool refresh_tableau(mvx_prob *P) {
  Context &c = ctx();
  MAIN_LOCK(c);
  flush_copies(c);
  SolveCtx &sc = c.main;
  const int m = P->m, n = P->n;
  std::vector<int> tflag((size_t)m + n + 1, 0), sflag((size_t)n + 1, 0);
  for (int j = 1; j <= n; j++) tflag[(size_t)P->nvar[(size_t)j]] = P->nflag[(size_t)j];
  for (int j = 1; j <= n; j++) sflag[(size_t)j] = tflag[(size_t)m + j] ? tflag[(size_t)m + j] : std_flag(P->ctype[(size_t)j]);
  const int status = P->status;
  if (!build_slack_tableau(P, sflag.data())) return false;
  ensure_scratch(sc, P->m_cap, P->ld);
  HIPCHECK(hipMemcpyAsync(sc.d_tflag, tflag.data(), tflag.size() * 4, hipMemcpyHostToDevice, sc.stream));
  fill_ctl(sc, P, sc.h_ctl);
  upload_ctl(sc);
  for (int k = m + 1; k <= m + n; k++) {
    if (tflag[(size_t)k]) continue; // non-basic in the target basis
    launch_refresh_select(sc.d_ctl, sc.d_tflag, k, sc.stream);
    launch_update(sc.d_ctl, m, n, sc.stream);
  }
  launch_export(sc.d_ctl, sc.d_stage, m, n, 1, sc.stream);
  pull_stage(sc, P, true); // synchronises: tflag / the control block may go out of scope
  P->status = status;
  P->refresh_cnt++;
  return true;
}

// the counter / residual / refresh step that follows every solve (oracle: orc_simplex)
// scripts/bnbrepeat.py: [0] tableau refreshes, [1] residual looks, [2] single-handle solves, [3] batch calls, [4] jobs a batch
// handed to the single-handle path (neither primal nor dual feasible)
static std::atomic<long long> g_dbg[8];
extern "C" void mvx_debug_counters(long long *out, int reset) {
  for (int k = 0; k < 8; k++) {
    out[k] = g_dbg[k].load();
    if (reset) g_dbg[k].store(0);
  }
}

static int after_solve(mvx_prob *P, const mvx_smcp *parm, int rc, int pivots) {
  P->piv_since_check += pivots;
  if (rc == 0 && P->status == MVX_OPT && P->piv_since_check >= g_check_every) {
    P->piv_since_check = 0;
    g_dbg[1]++;
    if (row_residual_sample(P, REFRESH_SAMPLE_ROWS) > g_refresh_tol && refresh_tableau(P)) {
      g_dbg[0]++;
      const int before = P->it_cnt;
      // the rebuilt tableau has not been looked at by anything yet: the simplex runs on it whatever the status says (the
      // oracle's simplex_once has no "already solved" shortcut) -- it may be infeasible or non-optimal beyond the
      // tolerance, or a numerically singular column may have been skipped
      P->status = MVX_UNDEF;
      rc = solve_once(P, parm, true); // the pivot limit of the call, if any, applies to this leg afresh
      P->piv_since_check += P->it_cnt - before;
    }
  }
  return rc;
}

static int engine_simplex_on(mvx_prob *P, const mvx_smcp *parm, bool aux) {
  const int before = P->it_cnt;
  g_dbg[2]++;
  const int rc = solve_once(P, parm, aux);
  return after_solve(P, parm, rc, P->it_cnt - before);
}

int engine_simplex(mvx_prob *P, const mvx_smcp *parm) { return engine_simplex_on(P, parm, false); }

// ------------------------------------------------------------------------ batched solves
// Independent node LPs (the two children of a branch, a window of open B&B nodes) advance together:
// ONE launch of k_select / k_update carries every slot of the batch (grid.z = slot), each slot
// with its own control block and scratch.  Small tableaux are bound by the dispatch rate of tiny
// kernels (measured: 8 streams give only ~1.7x), so the slots share launches instead of competing
// for them.  The handles form a work queue on the device: the host uploads one control block per handle
// and the slots pull them (k_select: a slot whose solve has ended exports its mirrors into the job's
// staging area and takes the next job in the very launch that finds it idle), so slots do not sit
// idle until the host's next synchronisation point -- the host only polls two counters.  Results are
// bit-identical to one mvx_simplex call per handle: every job runs the same generic state machine
// (k_select) on its own data.
struct BatchCtx {
  int slots = 0, jobs = 0;
  int m_cap = 0, ld = 0; // per-slot scratch capacity
  hipStream_t stream = nullptr;
  Ctl *d_ctl = nullptr, *h_ctl = nullptr;   // slot control blocks (host copy: the idle pattern)
  Ctl *d_jobs = nullptr, *h_jobs = nullptr; // job control blocks
  Ctl *h_fill = nullptr;                    // slot control blocks read back for the tail compaction
  hipEvent_t poll_ev = nullptr;             // follows each poll's copies on `stream`
  SlotScratch *d_sp = nullptr, *h_sp = nullptr;
  int *d_cnt = nullptr, *h_cnt = nullptr; // [0] next job, [1] finished jobs, [2] rounds launched, [3] round of the last job's end
  int pred_rounds = 16;                   // rounds the previous batch on this context needed
  unsigned char *scratch = nullptr;
  size_t scratch_stride = 0;
  unsigned char *d_stage = nullptr, *h_stage = nullptr; // one staging area per JOB
  size_t stage_stride = 0;
};
// Two batch contexts (stream, slot control blocks, scratch, staging): two host threads can each drive a batched solve
// at the same time -- the B&B driver splits a round's children over two workers, so that the host side of one batch
// (uploads, polls, result mirrors) overlaps the kernels of the other.
constexpr int N_BATCH_CTX = 8;
static BatchCtx g_batch[N_BATCH_CTX];
static std::mutex g_batch_mu[N_BATCH_CTX];
static void sync_batch_stream() {
  for (int k = 0; k < N_BATCH_CTX; k++)
    if (g_batch[k].stream) HIPCHECK(hipStreamSynchronize(g_batch[k].stream));
}
static int g_batch_slots = 64;

static void ensure_batch(BatchCtx &bc, int slots, int jobs, int m_cap, int ld) {
  if (!bc.stream) HIPCHECK(hipStreamCreateWithFlags(&bc.stream, hipStreamNonBlocking));
  if (!bc.poll_ev) HIPCHECK(hipEventCreateWithFlags(&bc.poll_ev, hipEventDisableTiming));
  if (slots <= bc.slots && jobs <= bc.jobs && m_cap <= bc.m_cap && ld <= bc.ld) return;
  HIPCHECK(hipStreamSynchronize(bc.stream));
  if (bc.d_ctl) HIPCHECK(hipFree(bc.d_ctl));
  if (bc.h_ctl) HIPCHECK(hipHostFree(bc.h_ctl));
  if (bc.d_jobs) HIPCHECK(hipFree(bc.d_jobs));
  if (bc.h_jobs) HIPCHECK(hipHostFree(bc.h_jobs));
  if (bc.h_fill) HIPCHECK(hipHostFree(bc.h_fill));
  if (bc.d_sp) HIPCHECK(hipFree(bc.d_sp));
  if (bc.h_sp) HIPCHECK(hipHostFree(bc.h_sp));
  if (bc.d_cnt) HIPCHECK(hipFree(bc.d_cnt));
  if (bc.h_cnt) HIPCHECK(hipHostFree(bc.h_cnt));
  if (bc.scratch) HIPCHECK(hipFree(bc.scratch));
  if (bc.d_stage) HIPCHECK(hipFree(bc.d_stage));
  if (bc.h_stage) HIPCHECK(hipHostFree(bc.h_stage));
  bc.slots = std::max(slots, bc.slots);
  bc.jobs = std::max(jobs + jobs / 2, bc.jobs);
  bc.m_cap = std::max(m_cap, bc.m_cap);
  bc.ld = std::max(ld, bc.ld);
  HIPCHECK(hipMalloc((void **)&bc.d_ctl, sizeof(Ctl) * bc.slots));
  HIPCHECK(hipHostMalloc((void **)&bc.h_ctl, sizeof(Ctl) * bc.slots));
  HIPCHECK(hipMalloc((void **)&bc.d_jobs, sizeof(Ctl) * bc.jobs));
  HIPCHECK(hipHostMalloc((void **)&bc.h_jobs, sizeof(Ctl) * bc.jobs));
  HIPCHECK(hipHostMalloc((void **)&bc.h_fill, sizeof(Ctl) * bc.slots));
  HIPCHECK(hipMalloc((void **)&bc.d_sp, sizeof(SlotScratch) * bc.slots));
  HIPCHECK(hipHostMalloc((void **)&bc.h_sp, sizeof(SlotScratch) * bc.slots));
  HIPCHECK(hipMalloc((void **)&bc.d_cnt, sizeof(int) * 4));
  HIPCHECK(hipMemsetAsync(bc.d_cnt, 0, sizeof(int) * 4, bc.stream)); // on the batch's own stream: a null-stream memset is not ordered with it
  HIPCHECK(hipHostMalloc((void **)&bc.h_cnt, sizeof(int) * 8));
  const size_t s_row = align_up((size_t)(bc.m_cap + 1) * 8, 256), s_col = align_up((size_t)bc.ld * 8, 256);
  const size_t s_var = align_up((size_t)(bc.m_cap + bc.ld + 1) * 8, 256);
  const size_t s_chain = (size_t)(DCH_MAX - 1) * (s_row + s_col);
  bc.scratch_stride = s_row + s_col + 2 * s_var + s_row + s_col + s_chain;
  HIPCHECK(hipMalloc((void **)&bc.scratch, bc.scratch_stride * bc.slots));
  HIPCHECK(hipMemsetAsync(bc.scratch, 0, bc.scratch_stride * bc.slots, bc.stream));
  for (int k = 0; k < bc.slots; k++) {
    unsigned char *sb = bc.scratch + (size_t)k * bc.scratch_stride;
    SlotScratch &sp = bc.h_sp[k];
    sp.colq = (double *)sb;
    sp.srow = (double *)(sb + s_row);
    sp.olb = (double *)(sb + s_row + s_col);
    sp.oub = (double *)(sb + s_row + s_col + s_var);
    sp.dw = (double *)(sb + s_row + s_col + 2 * s_var);
    sp.pw = (double *)(sb + s_row + s_col + 2 * s_var + s_row); // generic path only: one set of primal weights
    sp.chain = (double *)(sb + s_row + s_col + 2 * s_var + s_row + s_col);
    sp.chain_col = s_row;
    sp.chain_stride = s_row + s_col;
  }
  HIPCHECK(hipMemcpyAsync(bc.d_sp, bc.h_sp, sizeof(SlotScratch) * bc.slots, hipMemcpyHostToDevice, bc.stream));
  bc.stage_stride = stage_size(bc.m_cap, bc.ld);
  HIPCHECK(hipMalloc((void **)&bc.d_stage, bc.stage_stride * bc.jobs));
  HIPCHECK(hipHostMalloc((void **)&bc.h_stage, bc.stage_stride * bc.jobs));
  // an idle slot reads as finished and holds no job
  std::memset(bc.h_ctl, 0, sizeof(Ctl) * bc.slots);
  for (int k = 0; k < bc.slots; k++) {
    bc.h_ctl[k].done = D_FAIL;
    bc.h_ctl[k].job = -1;
  }
}

// control block of one job; the slot that pulls it points it at its own scratch (k_select)
static void batch_fill_job(Ctl *h, mvx_prob *P, const mvx_smcp &parm, int njobs) {
  std::memset(h, 0, sizeof(Ctl));
  h->T = P->d_T;
  h->bvar = P->d_bvar; h->blb = P->d_blb; h->bub = P->d_bub;
  h->nvar = P->d_nvar; h->nflag = P->d_nflag; h->nlb = P->d_nlb; h->nub = P->d_nub;
  h->m = P->m; h->n = P->n; h->ld = P->ld; h->m_cap = P->m_cap;
  h->sgn = (P->dir == MVX_MAX) ? 1.0 : -1.0;
  h->tol_bnd = parm.tol_bnd; h->tol_dj = parm.tol_dj; h->tol_piv = parm.tol_piv;
  h->phase = PH_START; h->done = D_RUN; h->budget = pivot_budget(P, parm);
  h->stall = 0; h->stall_limit = g_stall_limit > 0 ? g_stall_limit : 64 + (P->m + P->n) / 8;
  h->fstate = F_OFF;
  h->job = -1;
  h->nch = 1;
  h->dchain_max = dual_chain_length(njobs >= 32, P->m <= 1024 && P->n <= 1024);
  take_edits(P, h);
}

// mirrors + status of a finished job; layout inside the staging area follows the handle's own m_cap / ld
static int batch_finish_job(BatchCtx &bc, int j, mvx_prob *P, int *done_code, int *pivots) {
  const unsigned char *s = bc.h_stage + (size_t)j * bc.stage_stride;
  Ctl snap;
  std::memcpy(&snap, s, sizeof(Ctl));
  *done_code = snap.done;
  *pivots = snap.it_cnt;
  P->it_cnt += snap.it_cnt;
  P->bland_cnt += snap.n_bland;
  P->pert_cnt += snap.n_pert;
  if (snap.done == D_NEED_PHASE1) return 0; // neither primal nor dual feasible: the caller finishes it on the single-handle path
  const double *beta = (const double *)(s + sizeof(Ctl));
  const double *dj = beta + (P->m_cap + 1);
  const int *bv = (const int *)(dj + P->ld);
  const int *nv = bv + (P->m_cap + 1);
  const int *nf = nv + P->ld;
  P->beta.assign(beta, beta + P->m + 1);
  P->dj.assign(dj, dj + P->n + 1);
  P->bvar.assign(bv, bv + P->m + 1);
  P->nvar.assign(nv, nv + P->n + 1);
  P->nflag.assign(nf, nf + P->n + 1);
  rebuild_pos(P);
  P->sol_fresh = true;
  P->fresh_rows = -1;
  P->last_ms = 0.0;
  P->hint_dual = false;
  switch (snap.done) {
    case D_OPT: P->status = MVX_OPT; return 0;
    case D_UNBND: P->status = MVX_UNBND; return 0;
    case D_NOFEAS: P->status = MVX_NOFEAS; return 0;
    case D_ITLIM: P->status = (snap.phase == PH_PRIMAL2) ? MVX_FEAS : MVX_INFEAS; return MVX_EITLIM;
    default: P->status = MVX_UNDEF; return MVX_EFAIL;
  }
}

int engine_simplex_batch(mvx_prob **probs, int count, const mvx_smcp *parm_in, int *rcs) {
  if (count <= 0) return 0;
  Context &c = ctx();
  HIPCHECK(hipSetDevice(c.dev)); // per host thread, see engine_simplex
  mvx_smcp parm;
  if (parm_in) parm = *parm_in;
  else mvx_init_smcp(&parm);
  std::vector<int> pending, fallback;
  int m_cap = 0, ld = 0, m_max = 0, n_max = 0;
  for (int i = 0; i < count; i++) {
    mvx_prob *P = probs[i];
    if (P->m < 1 || P->n < 1) {
      P->status = MVX_UNDEF;
      if (rcs) rcs[i] = MVX_EFAIL;
      continue;
    }
    if (already_solved(P, parm)) {
      P->last_ms = 0.0;
      if (rcs) rcs[i] = 0;
      continue;
    }
    if (!P->valid && !build_slack_tableau(P)) {
      if (rcs) rcs[i] = MVX_EFAIL;
      continue;
    }
    remember_tolerances(P, parm);
    P->fresh_rows = -1; // as in solve_once
    pending.push_back(i);
    m_cap = std::max(m_cap, P->m_cap);
    ld = std::max(ld, P->ld);
    m_max = std::max(m_max, P->m);
    n_max = std::max(n_max, P->n);
  }
  {
    MAIN_LOCK(c);
    flush_copies(c); // clones recorded since the last device call: one launch for all of them
  }
  if (pending.size() == 1) { // nothing to share a launch with
    const int i = pending[0];
    HIPCHECK(hipStreamSynchronize(c.main.stream)); // edits queued on the main stream first
    const int rc = engine_simplex_on(probs[i], &parm, true);
    if (rcs) rcs[i] = rc;
    return 0;
  }
  if (pending.empty()) return 0;
  // whichever batch context is free; all busy: wait for the first
  std::unique_lock<std::mutex> batch_lock;
  int which = -1;
  for (int k = 0; k < N_BATCH_CTX && which < 0; k++) {
    batch_lock = std::unique_lock<std::mutex>(g_batch_mu[k], std::try_to_lock);
    if (batch_lock.owns_lock()) which = k;
  }
  if (which < 0) {
    batch_lock = std::unique_lock<std::mutex>(g_batch_mu[0]);
    which = 0;
  }
  BatchCtx &bc = g_batch[which];
  g_dbg[3]++;
  const int njobs = (int)pending.size();
  const int K = std::min(njobs, g_batch_slots);
  ensure_batch(bc, K, njobs, m_cap, ld);
  // edits queued on the main stream (bound changes, clones) must be visible to the batch stream
  HIPCHECK(hipStreamSynchronize(c.main.stream));
  for (int j = 0; j < njobs; j++) batch_fill_job(&bc.h_jobs[j], probs[pending[(size_t)j]], parm, njobs);
  bc.h_cnt[0] = bc.h_cnt[1] = bc.h_cnt[2] = bc.h_cnt[3] = 0;
  HIPCHECK(hipMemcpyAsync(bc.d_jobs, bc.h_jobs, sizeof(Ctl) * (size_t)njobs, hipMemcpyHostToDevice, bc.stream));
  HIPCHECK(hipMemcpyAsync(bc.d_ctl, bc.h_ctl, sizeof(Ctl) * (size_t)K, hipMemcpyHostToDevice, bc.stream)); // every slot idle
  HIPCHECK(hipMemcpyAsync(bc.d_cnt, bc.h_cnt, sizeof(int) * 4, hipMemcpyHostToDevice, bc.stream));
  BatchQueue q;
  q.jobs = bc.d_jobs;
  q.scratch = bc.d_sp;
  q.counters = bc.d_cnt;
  q.stage = bc.d_stage;
  q.stage_stride = bc.stage_stride;
  q.count = njobs;
  // Rounds queued ahead of the host.  The slots refill themselves, so a poll only has to notice the end; every round
  // queued past the end is three launches that start only to leave (~12 us), and every poll the stream waits for is a
  // host round trip with the GPU idle (~20 us).  So the polls are pipelined: burst k is queued BEFORE the host looks at
  // the poll that followed burst k-1 -- the stream never runs dry, at the price of at most one short burst of empty
  // rounds.  MVX_BATCH_FIRST / MVX_BATCH_BURST: rounds in the first / in every later burst (8 / 4; sweeps of 4..16 and
  // 2..8 are within the noise of one another).  MVX_BATCH_PRED=1 sizes the first burst by the previous batch on this
  // context (the device records the round in which the last job ended): measured no better, a poll early on lets the
  // tail shrink sooner.
  static const bool predict = std::getenv("MVX_BATCH_PRED") && std::atoi(std::getenv("MVX_BATCH_PRED")) != 0;
  static const int later = std::getenv("MVX_BATCH_BURST") ? std::max(1, std::min(64, std::atoi(std::getenv("MVX_BATCH_BURST")))) : 4;
  int Kact = K; // slots launched: all of them while the queue has work, the occupied ones once it is drained
  // slots that still hold a running solve, as of the last poll (an over-estimate by then): the update's tile depth goes
  // by the work in the launch, not by its slots -- behind the slowest LPs of a batch a launch is a few tableaux deep
  int busy = K;
  static const bool busy_tiles = !(std::getenv("MVX_BUSY_TILES") && std::atoi(std::getenv("MVX_BUSY_TILES")) == 0);
  auto launch_rounds = [&](int rounds) {
    for (int d = 0; d < rounds; d++) {
      launch_dsel(bc.d_ctl, m_max, n_max, bc.stream, Kact); // every slot whose dual phase is carrying on: its chain on chip
      launch_select_queue(bc.d_ctl, q, bc.stream, Kact);
      launch_update(bc.d_ctl, m_max, n_max, bc.stream, Kact, 1, busy_tiles ? busy : 0);
    }
  };
  bool with_fill = false; // the poll in flight carries the slots' control blocks (the queue was drained when it was queued)
  auto queue_poll = [&](bool fill) {
    HIPCHECK(hipMemcpyAsync(bc.h_cnt + 4, bc.d_cnt, sizeof(int) * 4, hipMemcpyDeviceToHost, bc.stream));
    with_fill = fill && Kact > 1;
    if (with_fill) HIPCHECK(hipMemcpyAsync(bc.h_fill, bc.d_ctl, sizeof(Ctl) * (size_t)Kact, hipMemcpyDeviceToHost, bc.stream));
    HIPCHECK(hipEventRecord(bc.poll_ev, bc.stream));
  };
  static const int first = std::getenv("MVX_BATCH_FIRST") ? std::max(1, std::min(64, std::atoi(std::getenv("MVX_BATCH_FIRST")))) : 8;
  launch_rounds(predict ? std::max(2, std::min(bc.pred_rounds, 256) - 2) : (njobs <= 2 ? std::min(8, first) : first));
  queue_poll(false);
  for (;;) {
    launch_rounds(later); // runs while the host waits for the poll queued before it
    HIPCHECK(hipEventSynchronize(bc.poll_ev));
    if (bc.h_cnt[5] >= njobs) break;
    const bool drained = bc.h_cnt[4] >= njobs; // no slot can pull a job any more
    busy = std::max(1, std::min(Kact, njobs - bc.h_cnt[5]));
    if (with_fill) {
      // Tail of the batch: the queue is empty and the slots finish one by one.  An idle slot of a launch still costs
      // its share of workgroups that start only to leave (512x1024: 132 per slot), so the control blocks that still
      // hold a job move to the front and the launches shrink.  Their pointers keep addressing their own scratch.  (The
      // snapshot is one burst old: a slot idle then is idle for good, one busy then is moved whether it has ended since
      // or not.)
      int w = 0;
      for (int k = 0; k < Kact; k++) {
        if (bc.h_fill[k].job < 0) continue;
        if (k != w) HIPCHECK(hipMemcpyAsync(&bc.d_ctl[w], &bc.d_ctl[k], sizeof(Ctl), hipMemcpyDeviceToDevice, bc.stream));
        w++;
      }
      if (w >= 1 && w < Kact) Kact = w;
    }
    queue_poll(drained);
  }
  bc.pred_rounds = std::max(1, bc.h_cnt[7]);
  HIPCHECK(hipMemcpyAsync(bc.h_stage, bc.d_stage, bc.stage_stride * (size_t)njobs, hipMemcpyDeviceToHost, bc.stream));
  HIPCHECK(hipStreamSynchronize(bc.stream));
  for (int j = 0; j < njobs; j++) {
    const int i = pending[(size_t)j];
    int code = 0, piv = 0;
    int rc = batch_finish_job(bc, j, probs[i], &code, &piv);
    if (code == D_NEED_PHASE1) {
      g_dbg[4]++;
      fallback.push_back(i);
      probs[i]->piv_since_check += piv; // the single-handle leg below adds its own and applies the refresh rule
    } else {
      rc = after_solve(probs[i], &parm, rc, piv); // residual look / tableau refresh, as after any solve
      if (rcs) rcs[i] = rc;
    }
  }
  for (int i : fallback) {
    // the batch left this handle untouched apart from zero or more completed pivots
    probs[i]->sol_fresh = false;
    probs[i]->fresh_rows = -1;
    const int rc = engine_simplex_on(probs[i], &parm, true);
    if (rcs) rcs[i] = rc;
  }
  return 0;
}

// -------------------------------------------------------------------------- model edits
void engine_apply_bounds(mvx_prob *P, int k, int type, double old_lb, double old_ub, double lb, double ub) {
  if (!P->valid) return;
  Context &c = ctx();
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  const int pos = P->pos[k];
  if (pos > 0) {
    // no launch: the edit waits on the handle and rides in the control block of the next solve (k_select applies it)
    bool merged = false;
    for (auto &e : P->pending)
      if (e.row == pos) {
        e.lb = lb;
        e.ub = ub;
        merged = true;
      }
    if (!merged) {
      if ((int)P->pending.size() == MAX_EDITS) {
        flush_copies(c); // a recorded clone INTO this handle's slab must land before the edits do
        flush_edits(sc, P);
      }
      P->pending.push_back({pos, lb, ub});
    }
    P->hint_dual = true; // a basic variable's bound moved: the warm start is a dual one (bs.cpp:274,282)
  } else {
    const int jj = -pos;
    const double xo = nb_value(P->nflag[jj], old_lb, old_ub);
    int flag;
    switch (type) {
      case MVX_FR: flag = MVX_NF; break;
      case MVX_LO: flag = MVX_NL; break;
      case MVX_UP: flag = MVX_NU; break;
      case MVX_DB: flag = (P->nflag[jj] == MVX_NU) ? MVX_NU : MVX_NL; break;
      default: flag = MVX_NS; break;
    }
    P->nflag[jj] = flag;
    flush_copies(c);
    launch_set_nonbasic(P->d_nlb, P->d_nub, P->d_nflag, jj, lb, ub, flag, sc.stream);
    const double xn = nb_value(flag, lb, ub);
    if (xn != xo) launch_shift_nonbasic(P->d_T, P->ld, P->m, jj, xn - xo, sc.stream);
  }
  P->sol_fresh = false;
  P->fresh_rows = -1;
  P->status = MVX_UNDEF;
}

void engine_add_rows(mvx_prob *P, int first, int nrs) {
  if (!P->valid) return;
  Context &c = ctx();
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  flush_copies(c);
  if (!grow_rows(P, P->m)) { // out of memory: the tableau is given up, the next solve restarts from the slack basis
    release_device(P);
    engine_invalidate(P);
    return;
  }
  launch_add_rows(P->d_T, P->ld, P->n, P->d_bvar, P->d_blb, P->d_bub, P->d_nvar, first, nrs, P->m, sc.stream);
  // host mirrors
  for (int i = 1; i < first; i++)
    if (P->bvar[i] >= first) P->bvar[i] += nrs;
  for (int j = 1; j <= P->n; j++)
    if (P->nvar[j] >= first) P->nvar[j] += nrs;
  P->bvar.resize((size_t)P->m + 1);
  for (int r = 0; r < nrs; r++) P->bvar[first + r] = first + r;
  rebuild_pos(P);
  // the new rows sit behind the old ones: what the mirrors hold of those stays current
  P->fresh_rows = P->sol_fresh ? first - 1 : std::min(P->fresh_rows, first - 1);
  P->sol_fresh = false;
  P->status = MVX_UNDEF;
}

// run k_rowcomb with host-provided weights / base, writing row `dst_row` of the tableau
constexpr int RC_SLOTS = 32;
static void rowcomb_into_row(mvx_prob *P, const std::vector<double> &w, const std::vector<double> &base, int dst_row) {
  Context &c = ctx();
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  flush_copies(c);
  ensure_scratch(sc, P->m_cap, P->ld);
  // The operands go through a pinned slot of their own, so nothing here waits for the device: a B&B round appends one
  // cut row to each of its branching nodes, and with a synchronisation in front of and behind every one of them the
  // host paid ~100 us per node for 10 us of kernels.  Whoever reads the row next does so on this stream or
  // synchronises it first (eval_tab_row, the clone launches, the batch entry).
  if (!sc.rc_ring || P->m_cap > sc.rc_m_cap || P->ld > sc.rc_ld) {
    HIPCHECK(hipStreamSynchronize(sc.stream));
    if (sc.rc_ring) HIPCHECK(hipHostFree(sc.rc_ring));
    sc.rc_m_cap = std::max(P->m_cap, sc.rc_m_cap);
    sc.rc_ld = std::max(P->ld, sc.rc_ld);
    sc.rc_slot_bytes = align_up(sizeof(Ctl), 256) + align_up((size_t)(sc.rc_m_cap + 1) * 8, 256) + align_up((size_t)sc.rc_ld * 8, 256);
    HIPCHECK(hipHostMalloc((void **)&sc.rc_ring, sc.rc_slot_bytes * RC_SLOTS));
    sc.rc_next = 0;
  }
  if (sc.rc_next == RC_SLOTS) { // every slot may still be in flight
    HIPCHECK(hipStreamSynchronize(sc.stream));
    sc.rc_next = 0;
  }
  unsigned char *slot = sc.rc_ring + sc.rc_slot_bytes * (size_t)sc.rc_next++;
  Ctl *hc = (Ctl *)slot;
  double *hw = (double *)(slot + align_up(sizeof(Ctl), 256));
  double *hb = (double *)((unsigned char *)hw + align_up((size_t)(sc.rc_m_cap + 1) * 8, 256));
  fill_ctl(sc, P, hc);
  hc->rc_base = sc.d_rcbase;
  hc->rc_out = P->d_T + (size_t)dst_row * P->ld;
  std::memcpy(hw, w.data(), (size_t)(P->m + 1) * 8);
  std::memcpy(hb, base.data(), (size_t)(P->n + 1) * 8);
  HIPCHECK(hipMemcpyAsync(sc.d_ctl, hc, sizeof(Ctl), hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(sc.d_wts, hw, (size_t)(P->m + 1) * 8, hipMemcpyHostToDevice, sc.stream));
  HIPCHECK(hipMemcpyAsync(sc.d_rcbase, hb, (size_t)(P->n + 1) * 8, hipMemcpyHostToDevice, sc.stream));
  launch_rowcomb(sc.d_ctl, P->m, P->n, 0, sc.stream);
}

void engine_row_from_model(mvx_prob *P, int i) {
  // x_i = sum_j v_j x_(m+j): substitute the basic structurals by their tableau rows
  const int m = P->m, n = P->n;
  const int pos = P->pos[i];
  const double *a = P->A[i]->data();
  std::vector<double> w((size_t)m + 1, 0.0), base((size_t)n + 1, 0.0);
  for (int r = 1; r <= m; r++)
    if (r != pos && P->bvar[r] > m) w[r] = a[P->bvar[r] - m];
  double b0 = 0.0;
  for (int jj = 1; jj <= n; jj++) {
    if (P->nvar[jj] > m) {
      const int col = P->nvar[jj] - m;
      const double v = a[col];
      const double x = nb_value(P->nflag[jj], P->clb[col], P->cub[col]);
      base[jj] = v;
      if (x != 0.0 && v != 0.0) b0 = std::fma(v, x, b0);
    }
  }
  base[0] = b0;
  rowcomb_into_row(P, w, base, pos);
  P->hint_dual = true; // appended cut rows (cut.cpp:40) leave the basis dual feasible
  P->fresh_rows = P->sol_fresh ? pos - 1 : std::min(P->fresh_rows, pos - 1); // only tableau row `pos` has been rewritten
  P->sol_fresh = false;
  P->status = MVX_UNDEF;
}

void engine_recompute_cost_row(mvx_prob *P) {
  if (!P->valid) return;
  const int m = P->m, n = P->n;
  std::vector<double> w((size_t)m + 1, 0.0), base((size_t)n + 1, 0.0);
  for (int i = 1; i <= m; i++) w[i] = (P->bvar[i] > m) ? P->c[P->bvar[i] - m] : 0.0;
  double z = P->c[0];
  for (int j = 1; j <= n; j++) {
    const int k = P->nvar[j];
    const double cj = (k > m) ? P->c[k - m] : 0.0;
    double lb, ub;
    var_bounds(P, k, &lb, &ub);
    const double x = nb_value(P->nflag[j], lb, ub);
    base[j] = cj;
    if (x != 0.0 && cj != 0.0) z = std::fma(cj, x, z);
  }
  base[0] = z;
  rowcomb_into_row(P, w, base, 0);
  P->sol_fresh = false;
  P->fresh_rows = -1;
  P->status = MVX_UNDEF;
}

void engine_invalidate(mvx_prob *P) {
  P->pending.clear();
  P->valid = false;
  P->sol_fresh = false;
  P->fresh_rows = -1;
  P->status = MVX_UNDEF;
}

void engine_copy(mvx_prob *dst, const mvx_prob *src) {
  // host fields were copied by the caller; clone the slab device-to-device
  dst->slab = nullptr;
  dst->slab_bytes = 0;
  dst->d_T = nullptr;
  if (!src->valid) {
    dst->valid = false;
    return;
  }
  Context &c = ctx();
  {
    static thread_local bool bound = false; // clones are made from the B&B driver's helper threads too
    if (!bound) {
      HIPCHECK(hipSetDevice(c.dev));
      bound = true;
    }
  }
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  void *slab = slab_alloc(c, src->slab_bytes);
  if (!slab) { // out of memory: the clone keeps the model only (mvx_last_error() reads MVX_ENOMEM)
    dst->valid = false;
    dst->sol_fresh = false;
    dst->fresh_rows = -1;
    dst->status = MVX_UNDEF;
    return;
  }
  bind_slab(dst, slab, src->m_cap, src->ld);
  // only the live rows of T need to travel; the small arrays follow T in one contiguous tail.  When the
  // spare rows in between are few (B&B clones of a small tableau) one range over the whole slab is cheaper
  // than two
  SlabLayout L = slab_layout(src->m_cap, src->ld);
  const size_t live = (size_t)(src->m + 1) * src->ld * 8;
  const bool whole = L.o_bvar - live <= (size_t)1 << 20;
  if (L.total <= ((size_t)64 << 20)) {
    // small tableau: record the clone; it leaves with the others of this round in one k_copy_many launch.  A source
    // that is itself waiting to be written by a recorded clone goes first.
    const unsigned char *lo = (const unsigned char *)src->slab, *hi = lo + src->slab_bytes;
    for (const CopyJob &j : c.copies)
      if ((const unsigned char *)j.dst >= lo && (const unsigned char *)j.dst < hi) {
        flush_copies(c);
        break;
      }
    if (whole) {
      c.copies.push_back({src->slab, slab, L.total});
    } else {
      c.copies.push_back({src->d_T, dst->d_T, live});
      c.copies.push_back({(const unsigned char *)src->slab + L.o_bvar, (unsigned char *)slab + L.o_bvar, L.total - L.o_bvar});
    }
    if (c.copies.size() >= 4 * COPY_BATCH) flush_copies(c);
  } else if (whole) {
    HIPCHECK(hipMemcpyAsync(slab, src->slab, L.total, hipMemcpyDeviceToDevice, sc.stream));
  } else {
    HIPCHECK(hipMemcpyAsync(dst->d_T, src->d_T, live, hipMemcpyDeviceToDevice, sc.stream));
    HIPCHECK(hipMemcpyAsync((unsigned char *)slab + L.o_bvar, (const unsigned char *)src->slab + L.o_bvar, L.total - L.o_bvar,
                            hipMemcpyDeviceToDevice, sc.stream));
  }
  dst->valid = true;
}

int engine_get_tableau(const mvx_prob *P, double *out) {
  if (!P->valid) return -1;
  Context &c = ctx();
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  flush_copies(c);
  HIPCHECK(hipMemcpy2DAsync(out, (size_t)(P->n + 1) * 8, P->d_T, (size_t)P->ld * 8, (size_t)(P->n + 1) * 8, (size_t)P->m + 1,
                            hipMemcpyDeviceToHost, sc.stream));
  HIPCHECK(hipStreamSynchronize(sc.stream));
  return 0;
}

int engine_get_row(const mvx_prob *P, int row, double *out) {
  if (!P->valid) return -1;
  Context &c = ctx();
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  flush_copies(c);
  HIPCHECK(hipMemcpyAsync(out, P->d_T + (size_t)row * P->ld, (size_t)(P->n + 1) * 8, hipMemcpyDeviceToHost, sc.stream));
  HIPCHECK(hipStreamSynchronize(sc.stream));
  return 0;
}

// ------------------------------------------------------------------------ GMI cuts (gmi.cpp:11-117)
static std::shared_ptr<DevMatrix> build_dev_matrix(Context &c, const mvx_prob *P, int rows = -1) { // rows: only the first `rows` (default: all)
  SolveCtx &sc = c.main;
  auto D = std::make_shared<DevMatrix>();
  const int m0 = (rows >= 0 && rows < P->m) ? rows : P->m, n = P->n;
  const int lda = (int)align_up((size_t)n + 1, LD_ALIGN);
  D->m0 = m0;
  D->n = n;
  D->lda = lda;
  std::vector<double> plain((size_t)(m0 + 1) * lda, 0.0), packed;
  std::vector<int> len((size_t)m0 + 1, 0);
  bool dense = true;
  for (int i = 1; i <= m0; i++) {
    const double *a = P->A[(size_t)i]->data();
    D->rows.push_back(P->A[(size_t)i].get());
    std::memcpy(&plain[(size_t)i * lda + 1], a + 1, (size_t)n * 8);
    int l = 0;
    for (int j = 1; j <= n; j++) l += (a[j] != 0.0);
    len[(size_t)i] = l;
    dense = dense && l == n;
  }
  const size_t bytes = plain.size() * 8;
  if (hipMalloc((void **)&D->plain, bytes) != hipSuccess) {
    (void)hipGetLastError();
    g_last_error.store(MVX_ENOMEM);
    D->plain = nullptr;
    return nullptr;
  }
  HIPCHECK(hipMemcpyAsync(D->plain, plain.data(), bytes, hipMemcpyHostToDevice, sc.stream));
  if (dense) {
    D->packed = D->plain;
  } else {
    packed.assign(plain.size(), 0.0);
    for (int i = 1; i <= m0; i++) {
      const double *a = P->A[(size_t)i]->data();
      int l = 0;
      for (int j = 1; j <= n; j++)
        if (a[j] != 0.0) packed[(size_t)i * lda + (size_t)(++l)] = a[j];
    }
    if (hipMalloc((void **)&D->packed, bytes) != hipSuccess || hipMalloc((void **)&D->len, len.size() * 4) != hipSuccess) {
      (void)hipGetLastError();
      g_last_error.store(MVX_ENOMEM);
      return nullptr;
    }
    HIPCHECK(hipMemcpyAsync(D->packed, packed.data(), bytes, hipMemcpyHostToDevice, sc.stream));
    HIPCHECK(hipMemcpyAsync(D->len, len.data(), len.size() * 4, hipMemcpyHostToDevice, sc.stream));
  }
  HIPCHECK(hipStreamSynchronize(sc.stream)); // the pageable sources above go out of scope
  return D;
}

// the handle's device matrix still describes its first m0 model rows?
static bool dev_matrix_current(const mvx_prob *P) {
  const DevMatrix *D = P->dmat.get();
  if (!D || D->n != P->n || D->m0 > P->m) return false;
  for (int i = 1; i <= D->m0; i++)
    if (D->rows[(size_t)i - 1] != P->A[(size_t)i].get()) return false;
  return true;
}

// `count` cuts, cut t taken from the solved handle Ps[t] for its structural column cols[t] (1-based, basic):
// vals[t][0..n] with vals[t][0] = rhs[t] = the cut's lower bound (gmi.cpp:91-109), ok[t] = 0 where no valid cut exists.
// mode 0 = generateCut3 as written (gmi.cpp:11-117), 1 = the repaired formula (mvx_generateCutGMI).  The handles share
// their columns and their first m0 model rows (B&B nodes of one tree: the root's rows are the same objects in every
// clone); each has its own tableau, basis and appended cut rows.  One launch pair for all of them: a round of a B&B
// window takes one cut from each of its branching nodes (bs.cpp:249-258 with cut.cpp:20's "last cut only").
static int gmi_core(mvx_prob *const *Ps, int mode, const int *cols, int count, double *vals, double *rhs, int *ok) {
  if (count < 1) return -1;
  Context &c = ctx();
  MAIN_LOCK(c);
  flush_copies(c);
  SolveCtx &sc = c.main;
  const int n = Ps[0]->n;
  int mmax = 0;
  for (int t = 0; t < count; t++) {
    if (!Ps[t]->valid || Ps[t]->n != n) return -1;
    mmax = std::max(mmax, Ps[t]->m);
  }
  // One device copy of the model rows the handles have in common serves them all: the rows every handle shares, object
  // for object, from row 1 on (the root's rows, and the cut rows of the ancestors the whole window descends from); what
  // a handle has beyond that -- its own lineage's cut rows -- is added on the host below, as for rows appended later.
  int common = Ps[0]->m;
  for (int t = 1; t < count; t++) {
    const mvx_prob *P = Ps[t];
    common = std::min(common, P->m);
    int i = 1;
    while (i <= common && P->A[(size_t)i].get() == Ps[0]->A[(size_t)i].get()) i++;
    common = i - 1;
  }
  if (common < 1) return -3; // not clones of one root
  std::shared_ptr<DevMatrix> Dp;
  for (int t = 0; t < count; t++) { // the deepest copy at hand that covers no more than the common rows
    const mvx_prob *P = Ps[t];
    if (P->dmat && P->dmat->m0 <= common && dev_matrix_current(P) && (!Dp || P->dmat->m0 > Dp->m0)) Dp = P->dmat;
  }
  if (!Dp) {
    Dp = build_dev_matrix(c, Ps[0], common);
    if (!Dp) return -2;
  }
  for (int t = 0; t < count; t++) Ps[t]->dmat = Dp; // (their children inherit it)
  const DevMatrix &D = *Dp;
  const size_t wld = align_up((size_t)mmax + n + 1, 32), old = align_up((size_t)n + 1, 32);
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  // [node descriptors x count][kind i32 x (n+1)] go up; [rhs][ok][out][work] come back (work: the auxiliaries' part only)
  const size_t o_nodes = carve((size_t)count * sizeof(GmiNode)), o_kind = carve((size_t)(n + 1) * 4), up_bytes = off;
  const size_t o_rhs = carve((size_t)count * 8), o_ok = carve((size_t)count * 4);
  const size_t o_out = carve((size_t)count * old * 8), o_work = carve((size_t)count * wld * 8);
  if (off > c.gmi_bytes) {
    HIPCHECK(hipStreamSynchronize(sc.stream));
    if (c.gmi_dev) HIPCHECK(hipFree(c.gmi_dev));
    if (c.gmi_host) HIPCHECK(hipHostFree(c.gmi_host));
    c.gmi_dev = c.gmi_host = nullptr;
    c.gmi_bytes = 0;
    const size_t want = off + off / 2;
    if (hipMalloc(&c.gmi_dev, want) != hipSuccess) {
      (void)hipGetLastError();
      g_last_error.store(MVX_ENOMEM);
      return -2;
    }
    HIPCHECK(hipHostMalloc(&c.gmi_host, want));
    c.gmi_bytes = want;
  }
  unsigned char *hb = (unsigned char *)c.gmi_host, *db = (unsigned char *)c.gmi_dev;
  GmiNode *h_nodes = (GmiNode *)(hb + o_nodes);
  int *h_kind = (int *)(hb + o_kind);
  bool own_rows = false;
  for (int t = 0; t < count; t++) {
    const mvx_prob *P = Ps[t];
    const int j = cols[t];
    if (j < 1 || j > n || P->pos[(size_t)P->m + j] <= 0) return -1; // the column must be basic (gmi.cpp:23)
    GmiNode &nd = h_nodes[t];
    nd.T = P->d_T; nd.nvar = P->d_nvar; nd.nflag = P->d_nflag; nd.nlb = P->d_nlb; nd.nub = P->d_nub;
    nd.m = P->m; nd.ld = P->ld; nd.pos = P->pos[(size_t)P->m + j]; nd.pad = 0;
    own_rows = own_rows || P->m > D.m0;
  }
  h_kind[0] = 0;
  for (int j = 1; j <= n; j++) h_kind[j] = Ps[0]->kind[(size_t)j];
  HIPCHECK(hipMemcpyAsync(db, hb, up_bytes, hipMemcpyHostToDevice, sc.stream));
  GmiArgs a;
  a.nodes = (const GmiNode *)(db + o_nodes);
  a.kind = (const int *)(db + o_kind);
  a.work = (double *)(db + o_work); a.rhs = (double *)(db + o_rhs); a.ok = (int *)(db + o_ok);
  a.A = mode == 0 ? D.packed : D.plain; a.len = mode == 0 ? D.len : nullptr; a.out = (double *)(db + o_out);
  a.n = n; a.wld = (int)wld; a.lda = D.lda; a.m0 = D.m0; a.old = (int)old; a.count = count; a.mode = mode;
  launch_gmi(a, sc.stream);
  HIPCHECK(hipMemcpyAsync(hb + o_rhs, db + o_rhs, o_work - o_rhs, hipMemcpyDeviceToHost, sc.stream)); // rhs, ok, out
  if (own_rows) // the auxiliaries of the rows appended since: their terms are added below
    HIPCHECK(hipMemcpy2DAsync(hb + o_work, wld * 8, db + o_work, wld * 8, (size_t)(mmax + 1) * 8, (size_t)count, hipMemcpyDeviceToHost, sc.stream));
  HIPCHECK(hipStreamSynchronize(sc.stream));
  const double *h_rhs = (const double *)(hb + o_rhs), *h_out = (const double *)(hb + o_out), *h_work = (const double *)(hb + o_work);
  const int *h_ok = (const int *)(hb + o_ok);
  for (int t = 0; t < count; t++) {
    const mvx_prob *P = Ps[t];
    double *v = vals + (size_t)t * (n + 1);
    std::memcpy(v + 1, h_out + (size_t)t * old + 1, (size_t)n * 8);
    // rows m0+1..m (this node's own cut rows) in order, the way gmi.cpp:81-89 / the repaired loop continue
    for (int i = D.m0 + 1; i <= P->m; i++) {
      const double wi = h_work[(size_t)t * wld + i];
      const double *ai = P->A[(size_t)i]->data();
      if (mode == 0) {
        int l = 0;
        for (int j = 1; j <= n; j++)
          if (ai[j] != 0.0) {
            ++l;
            v[l] += wi * ai[j]; // position l, not column j (gmi.cpp:87)
          }
      } else {
        if (wi == 0.0) continue;
        for (int j = 1; j <= n; j++)
          if (ai[j] != 0.0) v[j] += wi * ai[j];
      }
    }
    v[0] = h_rhs[t];
    rhs[t] = h_rhs[t];
    ok[t] = h_ok[t];
  }
  return 0;
}

int engine_gmi_cuts(const mvx_prob *Pc, int mode, const int *cols, int count, double *vals, double *rhs, int *ok) {
  if (count < 1) return -1;
  std::vector<mvx_prob *> Ps((size_t)count, const_cast<mvx_prob *>(Pc));
  return gmi_core(Ps.data(), mode, cols, count, vals, rhs, ok);
}

int engine_gmi_cuts_many(const mvx_prob *const *Ps, int mode, const int *cols, int count, double *vals, double *rhs, int *ok) {
  return gmi_core(const_cast<mvx_prob *const *>(Ps), mode, cols, count, vals, rhs, ok);
}

// ------------------------------------------------------------------ pack / unpack (migration)
// Device-side image of a handle for node migration between ranks:
//   [PackHdr][ctype,rtype,bvar,nvar,nflag i32][clb,cub,rlb,rub f64][model rows m_base+1..m, n+1 f64 each] pad 256
//   | T live rows | slab tail
// Rows 1..m_base of the model are the receiver's own (its copy of the root problem); the rows appended since
// (GMI cut rows, cut.cpp:23-43) travel densely with the image.
struct PackHdr {
  long long magic, m, n, ld, m_cap, status, it_cnt, valid, hint_dual, host_bytes, m_base, reserved;
  double last_tol[3];
  double pad_;
};
static const long long PACK_MAGIC = 0x4d56584849504cll;

static size_t pack_host_bytes(int m, int n, int m_base) {
  size_t sz = sizeof(PackHdr);
  sz += sizeof(int) * ((size_t)(n + 1) + 2 * (size_t)(m + 1) + 2 * (size_t)(n + 1));
  sz = align_up(sz, 8);
  sz += sizeof(double) * (2 * (size_t)(n + 1) + 2 * (size_t)(m + 1));
  sz += sizeof(double) * (size_t)(m - m_base) * (size_t)(n + 1);
  return align_up(sz, 256);
}

long long engine_pack_size(const mvx_prob *P, int m_base) {
  if (m_base < 0 || m_base > P->m) return -1;
  size_t sz = pack_host_bytes(P->m, P->n, m_base);
  if (P->valid) {
    SlabLayout L = slab_layout(P->m_cap, P->ld);
    sz += align_up((size_t)(P->m + 1) * P->ld * 8, 256) + (L.total - L.o_bvar);
  }
  return (long long)sz;
}

int engine_pack(const mvx_prob *P, int m_base, void *dev_buf) {
  if (m_base < 0 || m_base > P->m) return -1;
  Context &c = ctx();
  MAIN_LOCK(c);
  SolveCtx &sc = c.main;
  flush_copies(c);
  if (P->valid) flush_edits(sc, const_cast<mvx_prob *>(P)); // the image carries the device-side bounds
  const int m = P->m, n = P->n;
  const size_t hb = pack_host_bytes(m, n, m_base);
  std::vector<unsigned char> host(hb, 0);
  unsigned char *b = host.data();
  PackHdr h{};
  h.magic = PACK_MAGIC; h.m = m; h.n = n; h.ld = P->ld; h.m_cap = P->m_cap; h.status = P->status;
  h.it_cnt = P->it_cnt; h.valid = P->valid; h.hint_dual = P->hint_dual; h.host_bytes = (long long)hb; h.m_base = m_base; h.reserved = P->piv_since_check;
  std::memcpy(h.last_tol, P->last_tol, sizeof(h.last_tol));
  std::memcpy(b, &h, sizeof(h));
  b += sizeof(h);
  unsigned char *b0 = b;
  auto put = [&](const void *src, size_t bytes) {
    if (src) std::memcpy(b, src, bytes);
    b += bytes;
  };
  put(P->ctype.data(), sizeof(int) * (n + 1));
  put(P->rtype.data(), sizeof(int) * (m + 1));
  put(P->valid ? P->bvar.data() : nullptr, sizeof(int) * (m + 1));
  put(P->valid ? P->nvar.data() : nullptr, sizeof(int) * (n + 1));
  put(P->valid ? P->nflag.data() : nullptr, sizeof(int) * (n + 1));
  b = b0 + align_up((size_t)(b - b0), 8);
  put(P->clb.data(), 8 * (size_t)(n + 1));
  put(P->cub.data(), 8 * (size_t)(n + 1));
  put(P->rlb.data(), 8 * (size_t)(m + 1));
  put(P->rub.data(), 8 * (size_t)(m + 1));
  for (int i = m_base + 1; i <= m; i++) put(P->A[(size_t)i]->data(), 8 * (size_t)(n + 1));
  unsigned char *d = (unsigned char *)dev_buf;
  HIPCHECK(hipMemcpyAsync(d, host.data(), hb, hipMemcpyHostToDevice, sc.stream));
  if (P->valid) {
    SlabLayout L = slab_layout(P->m_cap, P->ld);
    const size_t tb = (size_t)(m + 1) * P->ld * 8;
    HIPCHECK(hipMemcpyAsync(d + hb, P->d_T, tb, hipMemcpyDeviceToDevice, sc.stream));
    HIPCHECK(hipMemcpyAsync(d + hb + align_up(tb, 256), (const unsigned char *)P->slab + L.o_bvar, L.total - L.o_bvar,
                            hipMemcpyDeviceToDevice, sc.stream));
  }
  HIPCHECK(hipStreamSynchronize(sc.stream));
  return 0;
}

// buffers for node images (mvx_image_api): plain device allocations on the bound device
void *engine_image_alloc(size_t bytes) {
  Context &c = ctx();
  if (hipSetDevice(c.dev) != hipSuccess) return nullptr;
  void *p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void engine_image_free(void *p) {
  if (p) (void)hipFree(p);
}

// dst already holds a copy of the receiver's root MODEL (rows 1..m_base, objective, kinds); the appended rows,
// bounds, basis and tableau come from the image
int engine_unpack(mvx_prob *dst, const void *dev_buf) {
  Context &c = ctx();
  MAIN_LOCK(c);
  flush_copies(c);
  SolveCtx &sc = c.main;
  const unsigned char *d = (const unsigned char *)dev_buf;
  PackHdr h;
  HIPCHECK(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, sc.stream));
  HIPCHECK(hipStreamSynchronize(sc.stream));
  if (h.magic != PACK_MAGIC || h.m_base != dst->m || h.n != dst->n || h.m < h.m_base) return -1;
  const int m = (int)h.m, n = (int)h.n, m_base = (int)h.m_base;
  std::vector<unsigned char> host((size_t)h.host_bytes);
  HIPCHECK(hipMemcpyAsync(host.data(), d, (size_t)h.host_bytes, hipMemcpyDeviceToHost, sc.stream));
  HIPCHECK(hipStreamSynchronize(sc.stream));
  const unsigned char *b = host.data() + sizeof(PackHdr);
  const unsigned char *b0 = b;
  auto get = [&](void *dstp, size_t bytes) {
    if (dstp) std::memcpy(dstp, b, bytes);
    b += bytes;
  };
  dst->m = m;
  dst->A.resize((size_t)m + 1);
  dst->rtype.resize((size_t)m + 1);
  dst->rlb.resize((size_t)m + 1);
  dst->rub.resize((size_t)m + 1);
  get(dst->ctype.data(), sizeof(int) * (n + 1));
  get(dst->rtype.data(), sizeof(int) * (m + 1));
  if (h.valid) {
    dst->bvar.assign((size_t)m + 1, 0);
    dst->nvar.assign((size_t)n + 1, 0);
    dst->nflag.assign((size_t)n + 1, 0);
  }
  get(h.valid ? dst->bvar.data() : nullptr, sizeof(int) * (m + 1));
  get(h.valid ? dst->nvar.data() : nullptr, sizeof(int) * (n + 1));
  get(h.valid ? dst->nflag.data() : nullptr, sizeof(int) * (n + 1));
  b = b0 + align_up((size_t)(b - b0), 8);
  get(dst->clb.data(), 8 * (size_t)(n + 1));
  get(dst->cub.data(), 8 * (size_t)(n + 1));
  get(dst->rlb.data(), 8 * (size_t)(m + 1));
  get(dst->rub.data(), 8 * (size_t)(m + 1));
  for (int i = m_base + 1; i <= m; i++) {
    auto row = std::make_shared<std::vector<double>>((size_t)n + 1, 0.0);
    get(row->data(), 8 * (size_t)(n + 1));
    dst->A.set((size_t)i, row);
  }
  dst->status = (int)h.status;
  dst->it_cnt = (int)h.it_cnt;
  dst->hint_dual = h.hint_dual != 0;
  dst->piv_since_check = (int)h.reserved;
  std::memcpy(dst->last_tol, h.last_tol, sizeof(dst->last_tol));
  dst->sol_fresh = false;
  dst->fresh_rows = -1;
  release_device(dst);
  dst->pending.clear();
  dst->valid = false;
  if (h.valid) {
    SlabLayout L = slab_layout((int)h.m_cap, (int)h.ld);
    void *slab = slab_alloc(c, L.total);
    if (!slab) return -2;
    bind_slab(dst, slab, (int)h.m_cap, (int)h.ld);
    const size_t tb = (size_t)(m + 1) * dst->ld * 8;
    HIPCHECK(hipMemcpyAsync(dst->d_T, d + h.host_bytes, tb, hipMemcpyDeviceToDevice, sc.stream));
    HIPCHECK(hipMemcpyAsync((unsigned char *)slab + L.o_bvar, d + h.host_bytes + align_up(tb, 256), L.total - L.o_bvar,
                            hipMemcpyDeviceToDevice, sc.stream));
    HIPCHECK(hipStreamSynchronize(sc.stream));
    rebuild_pos(dst);
    dst->valid = true;
  }
  return 0;
}

void tuning(int tr, int hot, int nt) {
  sync_stream();
  set_tuning(tr, hot, nt);
  if (g_ctx) { // partial-buffer sizes depend on the row-block depth
    g_ctx->main.sc_m_cap = 0;
    g_ctx->main.sc_ld = 0;
  }
}

void set_dual_chain(int len) { g_dchain = (len <= 0) ? 0 : std::min(DCH_MAX, len); } // <= 0: by batch width / size
void set_chain(int len) { g_chain = (len <= 0) ? 0 : std::min(KCH, len); } // 0: by tableau size
void set_cluster(int on) { // 0: k_pc / k_pr per step; 1: k_chain (default); also forgets an earlier abort
  g_cluster.store(on != 0);
  g_cluster_broken.store(false);
}
void cluster_stats(long long *launches, long long *aborts) {
  *launches = g_cluster_launches.load();
  *aborts = g_cluster_aborts.load();
}
void set_persist(int mode) {
  g_persist_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode); // 2: no size cap
  g_persist_broken = false;
}
void persist_stats(long long *launches, long long *aborts) {
  *launches = g_persist_launches;
  *aborts = g_persist_aborts;
}
// cycle totals of workgroup 0 per phase since the context was created: propose, gather, read, apply, pivots
void persist_cycles(unsigned long long *out5) {
  for (int k = 0; k < 48; k++) out5[k] = 0;
  if (!g_ctx || !g_ctx->main.h_pabort) return;
  const unsigned long long *d = (const unsigned long long *)(g_ctx->main.h_pabort + 16);
  for (int k = 0; k < 48; k++) out5[k] = d[k];
}
// diagnostic (MVX_FCS_DBG=1): the last phase stamps of k_fcs, 8 per chain position (+1 row for the boot launch), 10 ns ticks
int fcs_debug_stamps(unsigned long long *out) {
  Context &c = ctx();
  if (!c.main.h_ctl || !c.main.h_ctl->dbg) return 0;
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(out, c.main.h_ctl->dbg, (size_t)KCH * 16 * 8, hipMemcpyDeviceToHost));
  return KCH;
}
void set_stall_limit(int limit) { g_stall_limit = limit > 0 ? limit : 0; }
void set_batch_slots(int k) { g_batch_slots = k < 2 ? 2 : (k > 256 ? 256 : k); }
void profile_enable(int on) { ctx().prof = on != 0; }
void profile_reset() {
  Context &c = ctx();
  MAIN_LOCK(c);
  c.prof_update_ms = 0.0;
  c.prof_update_n = 0;
}
double profile_update_ms() { return g_ctx ? g_ctx->prof_update_ms : 0.0; }
long long profile_update_launches() { return g_ctx ? g_ctx->prof_update_n : 0; }

} // namespace mvx
