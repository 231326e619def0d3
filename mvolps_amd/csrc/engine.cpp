// engine.cpp -- host side of the MI355X dense-simplex engine: device context, per-handle
// HBM slabs, the queued pivot loop, tableau maintenance under model edits, and the export
// of basis / solution mirrors.  Counterpart of glp_simplex and the glp_* edit calls MVOLPS
// makes (/root/reference/bs.cpp:114-117,274-288; cut.cpp:23-43).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "engine.hpp"

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
void launch_fboot(Ctl *, int n, hipStream_t);
void launch_fa(Ctl *, int n, hipStream_t);
void launch_fb(Ctl *, int m, int n, hipStream_t);
void launch_select(Ctl *, hipStream_t);
void launch_update(Ctl *, int m, int n, hipStream_t);
void launch_p1_head(Ctl *, hipStream_t);
void launch_p1_select(Ctl *, hipStream_t);
void launch_rowcomb(Ctl *, int m, int n, int respect_done, hipStream_t);
void launch_shift_nonbasic(double *T, int ld, int m, int jj, double delta, hipStream_t);
void launch_set_basic_bounds(double *blb, double *bub, int i, double lb, double ub, hipStream_t);
void launch_set_nonbasic(double *nlb, double *nub, int *nflag, int j, double lb, double ub, int flag, hipStream_t);
void launch_add_rows(double *T, int ld, int n, int *bvar, double *blb, double *bub, int *nvar, int first, int nrs, int m_new,
                     hipStream_t);
void launch_export(Ctl *, unsigned char *stage, int m, int n, int force, hipStream_t);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------ context
struct Context {
  int dev = -1;
  hipStream_t stream = nullptr;
  Ctl *d_ctl = nullptr;
  Ctl *h_ctl = nullptr; // pinned
  // scratch sized for the largest problem seen
  int sc_m_cap = 0, sc_ld = 0;
  void *scratch = nullptr;
  double *d_colq = nullptr, *d_srow = nullptr, *d_cost1 = nullptr, *d_wts = nullptr, *d_part = nullptr, *d_rcbase = nullptr;
  int *d_gflag = nullptr;
  double *d_colqx[2] = {nullptr, nullptr}, *d_betac[2] = {nullptr, nullptr};
  Cand *d_pp[2] = {nullptr, nullptr}, *d_rp = nullptr;
  unsigned char *d_stage = nullptr, *h_stage = nullptr;
  size_t stage_bytes = 0;
  // slab recycling (B&B clones come and go at one size)
  std::multimap<size_t, void *> free_slabs;
  size_t cached_bytes = 0;
  // profiling
  bool prof = false;
  double prof_update_ms = 0.0;
  long long prof_update_n = 0;
  std::vector<hipEvent_t> ev_pool;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
};

static Context *g_ctx = nullptr;
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

static Context &ctx() {
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
  HIPCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHECK(hipMalloc((void **)&c->d_ctl, sizeof(Ctl)));
  HIPCHECK(hipHostMalloc((void **)&c->h_ctl, sizeof(Ctl)));
  HIPCHECK(hipEventCreate(&c->ev_a));
  HIPCHECK(hipEventCreate(&c->ev_b));
  g_ctx = c;
  return *c;
}

void sync_stream() {
  if (g_ctx) HIPCHECK(hipStreamSynchronize(g_ctx->stream));
}

static size_t stage_size(int m_cap, int ld) {
  return align_up(sizeof(Ctl) + (size_t)(m_cap + 1) * 8 + (size_t)ld * 8 + (size_t)(m_cap + 1) * 4 + (size_t)ld * 8, 256);
}

static void ensure_scratch(Context &c, int m_cap, int ld) {
  if (m_cap <= c.sc_m_cap && ld <= c.sc_ld) return;
  HIPCHECK(hipStreamSynchronize(c.stream));
  int mc = m_cap > c.sc_m_cap ? m_cap : c.sc_m_cap;
  int l = ld > c.sc_ld ? ld : c.sc_ld;
  if (c.scratch) HIPCHECK(hipFree(c.scratch));
  if (c.d_stage) HIPCHECK(hipFree(c.d_stage));
  if (c.h_stage) HIPCHECK(hipHostFree(c.h_stage));
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
  HIPCHECK(hipMalloc(&c.scratch, off));
  HIPCHECK(hipMemsetAsync(c.scratch, 0, off, c.stream));
  unsigned char *b = (unsigned char *)c.scratch;
  c.d_colq = (double *)(b + o_colq);
  c.d_srow = (double *)(b + o_srow);
  c.d_cost1 = (double *)(b + o_cost1);
  c.d_wts = (double *)(b + o_wts);
  c.d_rcbase = (double *)(b + o_rcb);
  c.d_gflag = (int *)(b + o_g);
  c.d_part = (double *)(b + o_part);
  c.d_colqx[0] = (double *)(b + o_cx0);
  c.d_colqx[1] = (double *)(b + o_cx1);
  c.d_betac[0] = (double *)(b + o_bc0);
  c.d_betac[1] = (double *)(b + o_bc1);
  c.d_pp[0] = (Cand *)(b + o_pp0);
  c.d_pp[1] = (Cand *)(b + o_pp1);
  c.d_rp = (Cand *)(b + o_rp);
  c.stage_bytes = stage_size(mc, l);
  HIPCHECK(hipMalloc((void **)&c.d_stage, c.stage_bytes));
  HIPCHECK(hipHostMalloc((void **)&c.h_stage, c.stage_bytes));
  c.sc_m_cap = mc;
  c.sc_ld = l;
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
}

static void *slab_alloc(Context &c, size_t bytes) {
  auto it = c.free_slabs.find(bytes);
  if (it != c.free_slabs.end()) {
    void *p = it->second;
    c.free_slabs.erase(it);
    c.cached_bytes -= bytes;
    return p;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    // drop the cache and retry once
    HIPCHECK(hipStreamSynchronize(c.stream));
    for (auto &kv : c.free_slabs) (void)hipFree(kv.second);
    c.free_slabs.clear();
    c.cached_bytes = 0;
    HIPCHECK(hipMalloc(&p, bytes));
  }
  return p;
}

void release_device(mvx_prob *P) {
  if (!P->slab) return;
  Context &c = ctx();
  // the stream is in-order: work already queued on the slab finishes before any reuse
  const size_t cache_limit = (size_t)8 << 30;
  if (c.cached_bytes + P->slab_bytes <= cache_limit) {
    c.free_slabs.emplace(P->slab_bytes, P->slab);
    c.cached_bytes += P->slab_bytes;
  } else {
    HIPCHECK(hipStreamSynchronize(c.stream));
    HIPCHECK(hipFree(P->slab));
  }
  P->slab = nullptr;
  P->slab_bytes = 0;
  P->d_T = nullptr;
  P->valid = false;
}

static void alloc_device(mvx_prob *P, int m_cap, int ld) {
  Context &c = ctx();
  SlabLayout L = slab_layout(m_cap, ld);
  void *slab = slab_alloc(c, L.total);
  bind_slab(P, slab, m_cap, ld);
}

static int ld_for(int n) { return (int)align_up((size_t)n + 1, LD_ALIGN); }

// grow row capacity, preserving contents
static void grow_rows(mvx_prob *P, int m_new) {
  if (m_new + ROW_SPARE <= P->m_cap) return;
  Context &c = ctx();
  const int cap = m_new + ROW_SLACK;
  void *o_slab = P->slab;
  const size_t o_bytes = P->slab_bytes;
  const int o_rows = P->m_cap + 1, ld = P->ld;
  double *oT = P->d_T, *oblb = P->d_blb, *obub = P->d_bub, *onlb = P->d_nlb, *onub = P->d_nub;
  int *obvar = P->d_bvar, *onvar = P->d_nvar, *onflag = P->d_nflag;
  SlabLayout Ln = slab_layout(cap, ld);
  void *slab = slab_alloc(c, Ln.total);
  bind_slab(P, slab, cap, ld);
  HIPCHECK(hipMemcpyAsync(P->d_T, oT, (size_t)o_rows * ld * 8, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_bvar, obvar, (size_t)o_rows * 4, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_blb, oblb, (size_t)o_rows * 8, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_bub, obub, (size_t)o_rows * 8, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nvar, onvar, (size_t)ld * 4, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nflag, onflag, (size_t)ld * 4, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nlb, onlb, (size_t)ld * 8, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nub, onub, (size_t)ld * 8, hipMemcpyDeviceToDevice, c.stream));
  // recycle the old slab (in-order stream: the copies above complete before any reuse)
  c.free_slabs.emplace(o_bytes, o_slab);
  c.cached_bytes += o_bytes;
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

static void fill_ctl(Context &c, mvx_prob *P, Ctl *h) {
  std::memset(h, 0, sizeof(Ctl));
  h->T = P->d_T;
  h->bvar = P->d_bvar; h->blb = P->d_blb; h->bub = P->d_bub;
  h->nvar = P->d_nvar; h->nflag = P->d_nflag; h->nlb = P->d_nlb; h->nub = P->d_nub;
  h->colq = c.d_colq; h->srow = c.d_srow; h->cost1 = c.d_cost1; h->wts = c.d_wts;
  h->gflag = c.d_gflag; h->part = c.d_part; h->rc_base = nullptr; h->rc_out = c.d_cost1;
  h->m = P->m; h->n = P->n; h->ld = P->ld; h->m_cap = P->m_cap;
  h->sgn = (P->dir == MVX_MAX) ? 1.0 : -1.0;
  h->tol_bnd = 1e-9; h->tol_dj = 1e-9; h->tol_piv = 1e-9;
  h->phase = PH_START; h->done = D_RUN; h->budget = -1;
  h->colqx[0] = c.d_colqx[0]; h->colqx[1] = c.d_colqx[1];
  h->betac[0] = c.d_betac[0]; h->betac[1] = c.d_betac[1];
  h->pp[0] = c.d_pp[0]; h->pp[1] = c.d_pp[1]; h->rp = c.d_rp;
  h->npb = fused_npb(P->n); h->nrb = 0; // nrb is published by k_fb (its grid height)
  h->fstate = F_OFF;
}

static void upload_ctl(Context &c) {
  HIPCHECK(hipMemcpyAsync(c.d_ctl, c.h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, c.stream));
}

// copy the staging buffer back and refresh the host mirrors
static void pull_stage(Context &c, mvx_prob *P, bool mirrors) {
  HIPCHECK(hipMemcpyAsync(c.h_stage, c.d_stage, stage_size(P->m_cap, P->ld), hipMemcpyDeviceToHost, c.stream));
  HIPCHECK(hipStreamSynchronize(c.stream));
  if (!mirrors) return;
  const unsigned char *s = c.h_stage;
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
}

void refresh_solution(const mvx_prob *Pc) {
  mvx_prob *P = const_cast<mvx_prob *>(Pc);
  if (!P->valid || P->sol_fresh) return;
  Context &c = ctx();
  ensure_scratch(c, P->m_cap, P->ld);
  fill_ctl(c, P, c.h_ctl);
  upload_ctl(c);
  launch_export(c.d_ctl, c.d_stage, P->m, P->n, 1, c.stream);
  pull_stage(c, P, true);
}

// ---------------------------------------------------------------------- tableau build
static void build_slack_tableau(mvx_prob *P) {
  Context &c = ctx();
  const int m = P->m, n = P->n;
  const int ld = ld_for(n);
  if (P->slab && (P->ld != ld || P->m_cap < m + ROW_SPARE)) release_device(P);
  if (!P->slab) alloc_device(P, m + ROW_SLACK, ld);
  ensure_scratch(c, P->m_cap, P->ld);
  P->bvar.assign((size_t)m + 1, 0);
  P->nvar.assign((size_t)n + 1, 0);
  P->nflag.assign((size_t)n + 1, 0);
  std::vector<double> nlb((size_t)ld, 0.0), nub((size_t)ld, 0.0), blb((size_t)m + 1, 0.0), bub((size_t)m + 1, 0.0);
  std::vector<double> xn((size_t)n + 1, 0.0);
  bool any_x = false;
  for (int j = 1; j <= n; j++) {
    P->nvar[j] = m + j;
    P->nflag[j] = std_flag(P->ctype[j]);
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
    HIPCHECK(hipMemcpyAsync(P->d_T + r0 * ld, bounce, (r1 - r0) * ld * 8, hipMemcpyHostToDevice, c.stream));
    HIPCHECK(hipStreamSynchronize(c.stream));
  }
  HIPCHECK(hipHostFree(bounce));
  HIPCHECK(hipMemcpyAsync(P->d_bvar, P->bvar.data(), (size_t)(m + 1) * 4, hipMemcpyHostToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_blb, blb.data(), (size_t)(m + 1) * 8, hipMemcpyHostToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_bub, bub.data(), (size_t)(m + 1) * 8, hipMemcpyHostToDevice, c.stream));
  std::vector<int> nv((size_t)ld, 0), nf((size_t)ld, MVX_NS);
  for (int j = 1; j <= n; j++) {
    nv[j] = P->nvar[j];
    nf[j] = P->nflag[j];
  }
  HIPCHECK(hipMemcpyAsync(P->d_nvar, nv.data(), (size_t)ld * 4, hipMemcpyHostToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nflag, nf.data(), (size_t)ld * 4, hipMemcpyHostToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nlb, nlb.data(), (size_t)ld * 8, hipMemcpyHostToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(P->d_nub, nub.data(), (size_t)ld * 8, hipMemcpyHostToDevice, c.stream));
  HIPCHECK(hipStreamSynchronize(c.stream));
  rebuild_pos(P);
  P->valid = true;
  P->sol_fresh = false;
  P->status = MVX_UNDEF;
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

int engine_simplex(mvx_prob *P, const mvx_smcp *parm) {
  mvx_smcp dflt;
  if (!parm) {
    mvx_init_smcp(&dflt);
    parm = &dflt;
  }
  if (P->m < 1 || P->n < 1) {
    P->status = MVX_UNDEF;
    return MVX_EFAIL;
  }
  Context &c = ctx();
  if (!P->valid) build_slack_tableau(P);
  ensure_scratch(c, P->m_cap, P->ld);
  Ctl *h = c.h_ctl;
  fill_ctl(c, P, h);
  h->tol_bnd = parm->tol_bnd;
  h->tol_dj = parm->tol_dj;
  h->tol_piv = parm->tol_piv;
  h->budget = parm->it_lim;
  upload_ctl(c);
  HIPCHECK(hipEventRecord(c.ev_a, c.stream));

  const int m = P->m, n = P->n;
  int batch = 8;
  int done = D_RUN;
  int seen_steps = 0; // pivots + flips already accounted to the profile
  int seen_pivots = 0;
  bool try_fused = !P->hint_dual; // dual-phase warm starts (B&B children) skip the primal fast path
  Ctl snap;
  for (;;) {
    size_t ev_used = 0;
    if (c.prof && c.ev_pool.size() < (size_t)2 * batch + 2) { // sized for the largest batch
      size_t old = c.ev_pool.size();
      c.ev_pool.resize((size_t)2 * batch + 2);
      for (size_t k = old; k < c.ev_pool.size(); k++) HIPCHECK(hipEventCreate(&c.ev_pool[k]));
    }
    // with a pivot limit, never queue more pivots than the limit still allows (+1 launch so that
    // k_select can observe the exhausted budget): keeps no-op launches out of profiles
    const int remaining = (parm->it_lim >= 0) ? std::max(0, parm->it_lim - seen_pivots) : (1 << 30);
    auto ev = [&]() {
      if (c.prof) HIPCHECK(hipEventRecord(c.ev_pool[ev_used++], c.stream));
    };
    if (try_fused) {
      // one generic step settles the phase; if it is primal phase 2 the fused two-kernel pipeline
      // (k_fa / k_fb) takes over, otherwise its launches return at once
      launch_select(c.d_ctl, c.stream);
      ev();
      launch_update(c.d_ctl, m, n, c.stream);
      ev();
      const int nf = std::min(batch - 1, remaining - 1);
      if (nf > 0) {
        launch_fboot(c.d_ctl, n, c.stream);
        launch_fb(c.d_ctl, m, n, c.stream);
        for (int k = 0; k < nf; k++) {
          launch_fa(c.d_ctl, n, c.stream);
          ev();
          launch_fb(c.d_ctl, m, n, c.stream);
          ev();
        }
      }
    } else {
      const int nb = std::max(1, std::min(batch, remaining));
      for (int k = 0; k < nb; k++) {
        launch_select(c.d_ctl, c.stream);
        ev();
        launch_update(c.d_ctl, m, n, c.stream);
        ev();
      }
    }
    launch_export(c.d_ctl, c.d_stage, m, n, 0, c.stream);
    pull_stage(c, P, false);
    std::memcpy(&snap, c.h_stage, sizeof(Ctl));
    if (c.prof) {
      // once the solve finishes inside a batch the queued-ahead launches are no-ops; only the
      // leading launches that really stepped are timed
      const int steps_now = snap.it_cnt + snap.n_flips;
      flush_update_events(c, std::min(ev_used, (size_t)2 * (size_t)(steps_now - seen_steps)));
      seen_steps = steps_now;
    }
    done = snap.done;
    seen_pivots = snap.it_cnt;
    try_fused = (snap.phase == PH_PRIMAL2);
    if (done == D_NEED_PHASE1) {
      // host-driven phase 1: per iteration head -> cost row (rowcomb) -> select -> update
      snap.done = D_RUN;
      snap.phase = PH_PHASE1;
      snap.rc_base = nullptr;
      snap.rc_out = c.d_cost1;
      *h = snap;
      upload_ctl(c);
      int pb = 4;
      for (;;) {
        for (int k = 0; k < pb; k++) {
          launch_p1_head(c.d_ctl, c.stream);
          launch_rowcomb(c.d_ctl, m, n, 1, c.stream);
          launch_p1_select(c.d_ctl, c.stream);
          launch_update(c.d_ctl, m, n, c.stream);
        }
        launch_export(c.d_ctl, c.d_stage, m, n, 0, c.stream);
        pull_stage(c, P, false);
        std::memcpy(&snap, c.h_stage, sizeof(Ctl));
        if (snap.done != D_RUN) break;
        pb = std::min(pb * 2, 64);
      }
      if (snap.done == D_PFEAS) {
        snap.rounds++;
        if (snap.rounds >= 64) {
          done = D_FAIL;
          break;
        }
        snap.done = D_RUN;
        snap.phase = PH_START;
        *h = snap;
        upload_ctl(c);
        batch = 8;
        continue;
      }
      done = snap.done;
      break;
    }
    if (done != D_RUN) break;
    batch = std::min(batch * 2, 256);
  }
  HIPCHECK(hipEventRecord(c.ev_b, c.stream));
  // final export with mirrors (forced: phase-1 exits and FAIL paths included)
  launch_export(c.d_ctl, c.d_stage, m, n, 1, c.stream);
  pull_stage(c, P, true);
  std::memcpy(&snap, c.h_stage, sizeof(Ctl));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, c.ev_a, c.ev_b));
  P->last_ms = ms;
  P->it_cnt += snap.it_cnt;
  P->hint_dual = false;
  switch (done) {
    case D_OPT: P->status = MVX_OPT; return 0;
    case D_UNBND: P->status = MVX_UNBND; return 0;
    case D_NOFEAS: P->status = MVX_NOFEAS; return 0;
    case D_ITLIM:
      P->status = (snap.phase == PH_PRIMAL2) ? MVX_FEAS : MVX_INFEAS;
      return MVX_EITLIM;
    default: P->status = MVX_UNDEF; return MVX_EFAIL;
  }
}

// -------------------------------------------------------------------------- model edits
void engine_apply_bounds(mvx_prob *P, int k, int type, double old_lb, double old_ub, double lb, double ub) {
  if (!P->valid) return;
  Context &c = ctx();
  const int pos = P->pos[k];
  if (pos > 0) {
    launch_set_basic_bounds(P->d_blb, P->d_bub, pos, lb, ub, c.stream);
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
    launch_set_nonbasic(P->d_nlb, P->d_nub, P->d_nflag, jj, lb, ub, flag, c.stream);
    const double xn = nb_value(flag, lb, ub);
    if (xn != xo) launch_shift_nonbasic(P->d_T, P->ld, P->m, jj, xn - xo, c.stream);
  }
  P->sol_fresh = false;
  P->status = MVX_UNDEF;
}

void engine_add_rows(mvx_prob *P, int first, int nrs) {
  if (!P->valid) return;
  Context &c = ctx();
  grow_rows(P, P->m);
  launch_add_rows(P->d_T, P->ld, P->n, P->d_bvar, P->d_blb, P->d_bub, P->d_nvar, first, nrs, P->m, c.stream);
  // host mirrors
  for (int i = 1; i < first; i++)
    if (P->bvar[i] >= first) P->bvar[i] += nrs;
  for (int j = 1; j <= P->n; j++)
    if (P->nvar[j] >= first) P->nvar[j] += nrs;
  P->bvar.resize((size_t)P->m + 1);
  for (int r = 0; r < nrs; r++) P->bvar[first + r] = first + r;
  rebuild_pos(P);
  P->sol_fresh = false;
  P->status = MVX_UNDEF;
}

// run k_rowcomb with host-provided weights / base, writing row `dst_row` of the tableau
static void rowcomb_into_row(mvx_prob *P, const std::vector<double> &w, const std::vector<double> &base, int dst_row) {
  Context &c = ctx();
  ensure_scratch(c, P->m_cap, P->ld);
  HIPCHECK(hipStreamSynchronize(c.stream)); // h_ctl / pageable sources below must not be in flight
  fill_ctl(c, P, c.h_ctl);
  c.h_ctl->rc_base = c.d_rcbase;
  c.h_ctl->rc_out = P->d_T + (size_t)dst_row * P->ld;
  upload_ctl(c);
  HIPCHECK(hipMemcpyAsync(c.d_wts, w.data(), (size_t)(P->m + 1) * 8, hipMemcpyHostToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync(c.d_rcbase, base.data(), (size_t)(P->n + 1) * 8, hipMemcpyHostToDevice, c.stream));
  launch_rowcomb(c.d_ctl, P->m, P->n, 0, c.stream);
  HIPCHECK(hipStreamSynchronize(c.stream));
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
  P->status = MVX_UNDEF;
}

void engine_invalidate(mvx_prob *P) {
  P->valid = false;
  P->sol_fresh = false;
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
  void *slab = slab_alloc(c, src->slab_bytes);
  bind_slab(dst, slab, src->m_cap, src->ld);
  // only the live rows of T need to travel; the small arrays follow T in one contiguous tail
  SlabLayout L = slab_layout(src->m_cap, src->ld);
  HIPCHECK(hipMemcpyAsync(dst->d_T, src->d_T, (size_t)(src->m + 1) * src->ld * 8, hipMemcpyDeviceToDevice, c.stream));
  HIPCHECK(hipMemcpyAsync((unsigned char *)slab + L.o_bvar, (const unsigned char *)src->slab + L.o_bvar, L.total - L.o_bvar,
                          hipMemcpyDeviceToDevice, c.stream));
  dst->valid = true;
}

int engine_get_tableau(const mvx_prob *P, double *out) {
  if (!P->valid) return -1;
  Context &c = ctx();
  HIPCHECK(hipMemcpy2DAsync(out, (size_t)(P->n + 1) * 8, P->d_T, (size_t)P->ld * 8, (size_t)(P->n + 1) * 8, (size_t)P->m + 1,
                            hipMemcpyDeviceToHost, c.stream));
  HIPCHECK(hipStreamSynchronize(c.stream));
  return 0;
}

int engine_get_row(const mvx_prob *P, int row, double *out) {
  if (!P->valid) return -1;
  Context &c = ctx();
  HIPCHECK(hipMemcpyAsync(out, P->d_T + (size_t)row * P->ld, (size_t)(P->n + 1) * 8, hipMemcpyDeviceToHost, c.stream));
  HIPCHECK(hipStreamSynchronize(c.stream));
  return 0;
}

// ------------------------------------------------------------------ pack / unpack (migration)
// Device-side image of a handle for node migration between ranks:
//   [PackHdr][ctype,rtype,bvar,nvar,nflag i32][clb,cub,rlb,rub f64] pad 256 | T live rows | slab tail
struct PackHdr {
  long long magic, m, n, ld, m_cap, status, it_cnt, valid, hint_dual, host_bytes, reserved[2];
};
static const long long PACK_MAGIC = 0x4d56584849504bll;

static size_t pack_host_bytes(int m, int n) {
  size_t sz = sizeof(PackHdr);
  sz += sizeof(int) * ((size_t)(n + 1) + 2 * (size_t)(m + 1) + 2 * (size_t)(n + 1));
  sz = align_up(sz, 8);
  sz += sizeof(double) * (2 * (size_t)(n + 1) + 2 * (size_t)(m + 1));
  return align_up(sz, 256);
}

long long engine_pack_size(const mvx_prob *P) {
  size_t sz = pack_host_bytes(P->m, P->n);
  if (P->valid) {
    SlabLayout L = slab_layout(P->m_cap, P->ld);
    sz += align_up((size_t)(P->m + 1) * P->ld * 8, 256) + (L.total - L.o_bvar);
  }
  return (long long)sz;
}

int engine_pack(const mvx_prob *P, void *dev_buf) {
  Context &c = ctx();
  const int m = P->m, n = P->n;
  const size_t hb = pack_host_bytes(m, n);
  std::vector<unsigned char> host(hb, 0);
  unsigned char *b = host.data();
  PackHdr h{};
  h.magic = PACK_MAGIC; h.m = m; h.n = n; h.ld = P->ld; h.m_cap = P->m_cap; h.status = P->status;
  h.it_cnt = P->it_cnt; h.valid = P->valid; h.hint_dual = P->hint_dual; h.host_bytes = (long long)hb;
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
  unsigned char *d = (unsigned char *)dev_buf;
  HIPCHECK(hipMemcpyAsync(d, host.data(), hb, hipMemcpyHostToDevice, c.stream));
  if (P->valid) {
    SlabLayout L = slab_layout(P->m_cap, P->ld);
    const size_t tb = (size_t)(m + 1) * P->ld * 8;
    HIPCHECK(hipMemcpyAsync(d + hb, P->d_T, tb, hipMemcpyDeviceToDevice, c.stream));
    HIPCHECK(hipMemcpyAsync(d + hb + align_up(tb, 256), (const unsigned char *)P->slab + L.o_bvar, L.total - L.o_bvar,
                            hipMemcpyDeviceToDevice, c.stream));
  }
  HIPCHECK(hipStreamSynchronize(c.stream));
  return 0;
}

// dst must already hold a copy of the receiver's root MODEL (rows, objective, kinds)
int engine_unpack(mvx_prob *dst, const void *dev_buf) {
  Context &c = ctx();
  const unsigned char *d = (const unsigned char *)dev_buf;
  PackHdr h;
  HIPCHECK(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c.stream));
  HIPCHECK(hipStreamSynchronize(c.stream));
  if (h.magic != PACK_MAGIC || h.m != dst->m || h.n != dst->n) return -1;
  const int m = (int)h.m, n = (int)h.n;
  std::vector<unsigned char> host((size_t)h.host_bytes);
  HIPCHECK(hipMemcpyAsync(host.data(), d, (size_t)h.host_bytes, hipMemcpyDeviceToHost, c.stream));
  HIPCHECK(hipStreamSynchronize(c.stream));
  const unsigned char *b = host.data() + sizeof(PackHdr);
  const unsigned char *b0 = b;
  auto get = [&](void *dstp, size_t bytes) {
    if (dstp) std::memcpy(dstp, b, bytes);
    b += bytes;
  };
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
  dst->status = (int)h.status;
  dst->it_cnt = (int)h.it_cnt;
  dst->hint_dual = h.hint_dual != 0;
  dst->sol_fresh = false;
  release_device(dst);
  dst->valid = false;
  if (h.valid) {
    SlabLayout L = slab_layout((int)h.m_cap, (int)h.ld);
    void *slab = slab_alloc(c, L.total);
    bind_slab(dst, slab, (int)h.m_cap, (int)h.ld);
    const size_t tb = (size_t)(m + 1) * dst->ld * 8;
    HIPCHECK(hipMemcpyAsync(dst->d_T, d + h.host_bytes, tb, hipMemcpyDeviceToDevice, c.stream));
    HIPCHECK(hipMemcpyAsync((unsigned char *)slab + L.o_bvar, d + h.host_bytes + align_up(tb, 256), L.total - L.o_bvar,
                            hipMemcpyDeviceToDevice, c.stream));
    HIPCHECK(hipStreamSynchronize(c.stream));
    rebuild_pos(dst);
    dst->valid = true;
  }
  return 0;
}

void tuning(int tr, int hot, int nt) {
  sync_stream();
  set_tuning(tr, hot, nt);
  if (g_ctx) { // partial-buffer sizes depend on the row-block depth
    g_ctx->sc_m_cap = 0;
    g_ctx->sc_ld = 0;
  }
}

void profile_enable(int on) { ctx().prof = on != 0; }
void profile_reset() {
  Context &c = ctx();
  c.prof_update_ms = 0.0;
  c.prof_update_n = 0;
}
double profile_update_ms() { return g_ctx ? g_ctx->prof_update_ms : 0.0; }
long long profile_update_launches() { return g_ctx ? g_ctx->prof_update_n : 0; }

} // namespace mvx
