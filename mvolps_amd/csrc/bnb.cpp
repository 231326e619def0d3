// bnb.cpp -- branch-and-bound driver over the GLPK-shaped LP-engine table (include/mvx_bnb.h).
//
// Host-side mirror of MVOLPS's own control flow, same names and argument meaning:
//   MVOLP::NodeData        util.cpp:25-42     MVOLP::ParameterObj::pickNode/pickVar  util.cpp:154-230
//   printInfo              util.cpp:414-473   getFract                               util.cpp:11-23
//   generateCut3           gmi.cpp:11-117     CutPool::addToPool/addCutConstraint    cut.cpp:6-46
//   branchAndBound         bs.cpp:54-348      getParentOid / getBranchDirection      bs.cpp:26-52
// Every LP call goes through `mvx_lp_api`; with the default table that is the gfx950 engine.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <limits>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/mvx_bnb.h"

namespace {

struct CutContainer { // cut.h:7-13
  std::vector<int> inds;
  std::vector<double> vals;
  double lb = 0.0;
  int oid = 0;
};

class CutPool { // cut.h:15-23
public:
  explicit CutPool(const mvx_lp_api *api) : _api(api) {}
  int addToPool(CutContainer cut) { // cut.cpp:6-9
    _cuts.push_back(std::move(cut));
    return (int)_cuts.size();
  }
  // the pool keeps only what addCutConstraint can ever read: its last element (cut.cpp:20)
  void replaceLast(CutContainer cut) {
    if (_cuts.empty()) _cuts.push_back(std::move(cut));
    else _cuts.back() = std::move(cut);
  }
  int addCutConstraint(void *in, int cID = -1) { // cut.cpp:11-46
    if (_cuts.empty()) return -1;
    if (cID < 0) cID = (int)_cuts.size() - 1; // cut.cpp:20
    const int index = _api->add_rows(in, 1);
    const CutContainer &sel = _cuts.at((size_t)cID);
    _api->set_mat_row(in, index, (int)sel.inds.size() - 1, sel.inds.data(), sel.vals.data()); // cut.cpp:40
    _api->set_row_bnds(in, index, MVX_LO, sel.lb, 0);                                          // cut.cpp:43
    return cID;
  }

private:
  const mvx_lp_api *_api;
  std::vector<CutContainer> _cuts;
};

double getFract(double x) { // util.cpp:11-23
  double intPart;
  double fractPart = std::modf(x, &intPart);
  if (fractPart < 0.0) fractPart += 1;
  return fractPart;
}

std::pair<int, std::vector<int>> printInfo(const mvx_lp_api *api, const void *prob, bool quirks) { // util.cpp:414-473
  const int cols = api->get_num_cols(prob);
  std::vector<int> violated;
  const int status = api->get_status(prob);
  if (status == MVX_NOFEAS || status == MVX_INFEAS || status == MVX_UNBND) return {-1, violated}; // util.cpp:424
  std::vector<double> xs;
  if (api->get_col_prim_all) { // one call instead of n through the table
    xs.resize((size_t)cols + 1);
    api->get_col_prim_all(prob, xs.data());
  }
  // this loop runs over every column of every node: trunc / round without a libm call per column (values beyond 2^52
  // are integers; |v - round(v)| = min(|f|, 1 - |f|) with f = v - trunc(v), exact wherever it is near the tolerance)
  auto trunc_of = [](double v) { return std::fabs(v) < 4503599627370496.0 ? (double)(long long)v : v; };
  for (int i = 1; i <= cols; i++) {
    const double v = xs.empty() ? api->get_col_prim(prob, i) : xs[(size_t)i];
    const double t = trunc_of(v);
    if (!quirks) {
      const double f = std::fabs(v - t);
      if ((f < 1.0 - f ? f : 1.0 - f) > 1e-9 && api->get_col_kind(prob, i) != MVX_CV) violated.push_back(i);
      continue;
    }
    if (v != 0 && api->get_obj_coef(prob, i) != 0) {            // util.cpp:437
      if (t != v && api->get_col_kind(prob, i) != MVX_CV) {     // util.cpp:443-444
        violated.push_back(i);
      }
    }
  }
  return {violated.empty() ? 1 : 0, violated};
}

// gmi.cpp:11-117.  oid stays -1 on rejection (gmi.cpp:20,25); on success the reference leaves it
// indeterminate and bs.cpp:252 treats it as "not -1", which is what 0 reproduces.
CutContainer generateCut3(const mvx_lp_api *api, const void *in, int j) {
  CutContainer result;
  double temp = 0.0; // uninitialised at gmi.cpp:13
  const int m = api->get_num_rows(in);
  const int n = api->get_num_cols(in);
  if (api->get_col_kind(in, j) != MVX_IV || api->get_col_stat(in, j) != MVX_BS) {
    result.oid = -1;
    return result;
  }
  std::vector<double> work((size_t)m + n + 1, 0.0), val2((size_t)n + 1, 0.0);
  std::vector<int> ind2((size_t)n + 1, 0);
  const int len = api->eval_tab_row(in, m + j, ind2.data(), val2.data()); // gmi.cpp:36
  double rhs = api->get_col_prim(in, j);                                   // gmi.cpp:37
  for (int i = 1; i <= len; i++) {
    const double val = val2[i];
    int kind;
    double ub;
    if (ind2[i] <= m) { // gmi.cpp:42-47
      kind = MVX_CV;
      ub = api->get_row_ub(in, ind2[i]);
    } else { // gmi.cpp:48-53
      const int curCol = ind2[i] - m;
      kind = api->get_col_kind(in, curCol);
      ub = api->get_col_ub(in, curCol);
    }
    const double fRhs = getFract(rhs); // the RUNNING rhs (gmi.cpp:55,73)
    const double fVal = getFract(val);
    if (kind == MVX_IV) temp = (fRhs >= fVal) ? fVal : (fRhs / (1.0 - fRhs)) * (1.0 - fVal);
    if (kind == MVX_CV) temp = (val >= 0.0) ? val : (fRhs / (1.0 - fRhs)) * (-1.0 * val);
    work[ind2[i]] = -1.0 * temp; // gmi.cpp:72
    rhs -= temp * ub;            // gmi.cpp:73
  }
  // gmi.cpp:81-89: back-substitution indexed by POSITION k in the row's non-zero list
  std::vector<double> rv((size_t)n + 1);
  std::vector<int> ri((size_t)n + 1);
  for (int i = 1; i <= m; i++) {
    // no shortcut for work[i] == 0: an earlier cut of this formula may have left inf / NaN coefficients in the
    // model (gmi.cpp:73 with absent bounds), and 0 * inf is NaN -- bs.cpp goes through every row, so does this
    const int len2 = api->get_mat_row(in, i, ri.data(), rv.data());
    const double wi = work[i];
    for (int k = 1; k <= len2; k++) work[m + k] += wi * rv[k];
  }
  result.inds.resize((size_t)n + 1);
  result.vals.resize((size_t)n + 1);
  result.inds[0] = 0;
  result.vals[0] = rhs;
  for (int i = 1; i <= n; i++) {
    result.inds[i] = i;
    result.vals[i] = work[m + i];
  }
  result.lb = rhs;
  result.oid = 0;
  return result;
}

// Repaired GMI -- see the derivation next to orc_generateCutGMI.
CutContainer generateCutGMI(const mvx_lp_api *api, const void *in, int j, double *efficacy) {
  CutContainer result;
  result.oid = -1;
  const int m = api->get_num_rows(in), n = api->get_num_cols(in);
  if (api->get_col_kind(in, j) == MVX_CV) return result;
  if (api->get_col_stat(in, j) != MVX_BS) return result;
  const double beta = api->get_col_prim(in, j);
  const double f0 = getFract(beta);
  if (f0 < 1e-6 || f0 > 1.0 - 1e-6) return result;
  std::vector<double> val2((size_t)n + 1, 0.0), work((size_t)m + n + 1, 0.0);
  std::vector<int> ind2((size_t)n + 1, 0);
  const int len = api->eval_tab_row(in, m + j, ind2.data(), val2.data());
  double rhs = 1.0;
  for (int t = 1; t <= len; t++) {
    const int k = ind2[t];
    const double alpha = val2[t];
    int stat;
    bool isint;
    double lo, up;
    if (k <= m) {
      stat = api->get_row_stat(in, k);
      isint = false;
      lo = api->get_row_lb(in, k);
      up = api->get_row_ub(in, k);
    } else {
      stat = api->get_col_stat(in, k - m);
      isint = api->get_col_kind(in, k - m) != MVX_CV;
      lo = api->get_col_lb(in, k - m);
      up = api->get_col_ub(in, k - m);
    }
    if (stat == MVX_NS) continue;      // fixed: y_j = 0
    if (stat == MVX_NF) return result; // free non-basic with a non-zero entry: no valid cut
    const double abar = (stat == MVX_NL) ? -alpha : alpha;
    double g;
    if (isint) {
      const double fj = getFract(abar);
      g = (fj <= f0) ? fj / f0 : (1.0 - fj) / (1.0 - f0);
    } else {
      g = (abar >= 0.0) ? abar / f0 : -abar / (1.0 - f0);
    }
    if (stat == MVX_NL) {
      work[k] += g;
      rhs += g * lo;
    } else {
      work[k] -= g;
      rhs -= g * up;
    }
  }
  std::vector<double> rv((size_t)n + 1);
  std::vector<int> ri((size_t)n + 1);
  for (int i = 1; i <= m; i++) {
    if (work[i] == 0.0) continue;
    const int len2 = api->get_mat_row(in, i, ri.data(), rv.data());
    for (int t = 1; t <= len2; t++) work[m + ri[t]] += work[i] * rv[t];
  }
  result.inds.resize((size_t)n + 1);
  result.vals.resize((size_t)n + 1);
  result.inds[0] = 0;
  result.vals[0] = rhs;
  double dot = 0.0, nrm = 0.0;
  for (int k = 1; k <= n; k++) {
    result.inds[k] = k;
    result.vals[k] = work[m + k];
    dot += result.vals[k] * api->get_col_prim(in, k);
    nrm += result.vals[k] * result.vals[k];
  }
  result.lb = rhs;
  if (!(nrm > 0.0)) return result;
  *efficacy = (rhs - dot) / std::sqrt(nrm);
  result.oid = 0;
  return result;
}

namespace MVOLP {
enum PruneType { INTG = 0, FEAS = 1, BNDS = 3, NONE = 4 }; // util.h:27

struct NodeData { // util.h:35-54
  NodeData(const mvx_lp_api *api, const void *parent, int &idCounter) : _api(api) { // util.cpp:25-37
    oid = idCounter;
    idCounter += 1;
    lowerBound = -std::numeric_limits<double>::infinity();
    upperBound = std::numeric_limits<double>::infinity();
    prob = _api->create_prob();
    _api->copy_prob(prob, parent, MVX_ON);
    inital = false;
  }
  // adopts `owned` instead of cloning it: the state a clone would have, without the device-to-device copy
  NodeData(const mvx_lp_api *api, void *owned, int &idCounter, bool /*adopt*/) : _api(api) {
    oid = idCounter;
    idCounter += 1;
    lowerBound = -std::numeric_limits<double>::infinity();
    upperBound = std::numeric_limits<double>::infinity();
    prob = owned;
    inital = false;
  }
  ~NodeData() {
    if (prob) _api->delete_prob(prob); // util.cpp:39-42
  }
  NodeData(const NodeData &) = delete;
  NodeData &operator=(const NodeData &) = delete;
  double upperBound, lowerBound;
  void *prob;
  bool inital;
  int oid;
  int repiv = -1; // window driver: pivots of the pop-time re-solve (bs.cpp:117) when it was done ahead, else -1

private:
  const mvx_lp_api *_api;
};

// +1: compare bounds as a maximiser (bs.cpp:172,210 always do); -1: repaired mode on a minimisation problem
inline double sense_of(const mvx_lp_api *api, const void *prob, const mvx_bnb_params &p) {
  return (p.reference_quirks == 0 && api->get_obj_dir && api->get_obj_dir(prob) == MVX_MIN) ? -1.0 : 1.0;
}

class ParameterObj { // util.h:61-99
public:
  ParameterObj(const mvx_lp_api *api, const void *prob, const mvx_bnb_params &p)
      : _api(api), _prob(prob), _p(p), _sg(sense_of(api, prob, p)) {}
  double sense() const { return _sg; }
  bool IsCutEnabled() const { return _p.cut_strat != 0; }

  std::shared_ptr<NodeData> pickNode(const std::deque<std::shared_ptr<NodeData>> &problems, int &index) const { // util.cpp:154-188
    if (_p.node_strat == 0) { // "DFS" == FIFO
      index = 0;
      return problems.front();
    }
    index = 0;
    for (int i = 1; i < (int)problems.size(); i++)
      if (_sg * problems[(size_t)index]->upperBound < _sg * problems[(size_t)i]->upperBound) index = i; // first maximum
    return problems.at((size_t)index);
  }

  int pickVar(const std::vector<int> &vars) const { // util.cpp:190-230
    if (_p.var_strat == 0) return vars.front();
    if (_p.var_strat == 1) {
      // util.cpp:200-203 query _prob, the ROOT problem that is never solved
      double curBest = std::fabs(getFract(_api->get_col_prim(_prob, vars.front())) - 0.5);
      int index = vars.front();
      for (int i : vars) {
        const double cur = std::fabs(getFract(_api->get_col_prim(_prob, i)) - 0.5);
        if (cur < curBest) {
          curBest = cur;
          index = i;
        }
      }
      return index;
    }
    double bestCoef = 0.0;
    int index = vars.front(); // uninitialised in the reference when no coefficient is > 0
    for (int i : vars) {
      const double cur = _api->get_obj_coef(_prob, i);
      if (cur > bestCoef) {
        bestCoef = cur;
        index = i;
      }
    }
    return index;
  }

private:
  const mvx_lp_api *_api;
  const void *_prob;
  mvx_bnb_params _p;
  double _sg;
};
} // namespace MVOLP

int getBranchDirection(int oid) { // bs.cpp:43-52
  if (oid <= 1) return 0;
  return (oid % 2 == 0) ? 1 : 2;
}

struct Recorder {
  std::vector<int> parent, prune;
  std::vector<double> bound;
  std::vector<mvx_bnb_event> events;
  std::vector<mvx_bnb_event> *sink = nullptr; // window mode buffers events per node
  long long pivots = 0;
  void node(int oid, int pid) {
    if ((int)parent.size() <= oid) {
      parent.resize((size_t)oid + 1, 0);
      prune.resize((size_t)oid + 1, MVOLP::NONE);
      bound.resize((size_t)oid + 1, std::numeric_limits<double>::infinity());
    }
    parent[(size_t)oid] = pid;
  }
  void emit(int type, int oid, double f6, double f7, int f8, int pick) {
    mvx_bnb_event e;
    e.type = type;
    e.oid = oid;
    e.pid = parent[(size_t)oid]; // getParentOid, bs.cpp:26-33
    e.direction = getBranchDirection(oid);
    e.lp_bound = f6;
    e.sum_infeas = f7;
    e.n_violated = f8;
    e.pick = pick;
    (sink ? *sink : events).push_back(e);
  }
};

int solve(const mvx_lp_api *api, void *p, Recorder &rec) {
  const int before = api->get_it_cnt(p);
  const int rc = api->simplex(p, nullptr); // the reference ignores the return code
  rec.pivots += api->get_it_cnt(p) - before;
  return rc;
}

template <typename T>
T *dup(const std::vector<T> &v) {
  T *p = (T *)std::malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
  if (!v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

void pack_result(mvx_bnb_result *res, const Recorder &rec, int id, int count, int has_incumbent, double bestLower, int incumbent_oid,
                 int n0, const std::vector<double> &xbest, int hit_limit) {
  std::memset(res, 0, sizeof(*res));
  res->n_nodes = id - 1;
  res->parent = dup(rec.parent);
  res->prune = dup(rec.prune);
  res->node_bound = dup(rec.bound);
  res->n_events = (int)rec.events.size();
  res->events = dup(rec.events);
  res->count = count;
  res->has_incumbent = has_incumbent;
  res->best_lower = bestLower;
  res->incumbent_oid = incumbent_oid;
  res->n = n0;
  res->x = dup(xbest);
  res->total_pivots = rec.pivots;
  res->hit_limit = hit_limit;
}

// A node's cuts through the engine's batch entry (mvx_lp_api.gmi_cuts): `cols` are the candidate columns (each one
// basic and integer, checked by the caller the way gmi.cpp:18-27 / the repaired filter do); returns one container per
// column, oid -1 where the engine found no cut.
static std::vector<CutContainer> cuts_via_engine(const mvx_lp_api *api, const void *a, bool repaired, const std::vector<int> &cols,
                                                 std::vector<double> *eff) {
  const int n = api->get_num_cols(a), k = (int)cols.size();
  std::vector<double> vals((size_t)k * (n + 1)), rhs((size_t)k);
  std::vector<int> ok((size_t)k, 0);
  std::vector<CutContainer> out((size_t)k);
  if (api->gmi_cuts(a, repaired ? 1 : 0, cols.data(), k, vals.data(), rhs.data(), ok.data()) != 0) {
    for (auto &c : out) c.oid = -1;
    return out;
  }
  std::vector<double> x;
  if (repaired) {
    x.resize((size_t)n + 1);
    for (int j = 1; j <= n; j++) x[(size_t)j] = api->get_col_prim(a, j);
  }
  for (int t = 0; t < k; t++) {
    CutContainer &c = out[(size_t)t];
    if (!ok[(size_t)t]) {
      c.oid = -1;
      continue;
    }
    const double *v = &vals[(size_t)t * (n + 1)];
    c.inds.resize((size_t)n + 1);
    c.vals.assign(v, v + n + 1);
    for (int j = 0; j <= n; j++) c.inds[(size_t)j] = j;
    c.lb = rhs[(size_t)t];
    c.oid = 0;
    if (repaired) { // efficacy as generateCutGMI computes it
      double dot = 0.0, nrm = 0.0;
      for (int j = 1; j <= n; j++) {
        dot += c.vals[(size_t)j] * x[(size_t)j];
        nrm += c.vals[(size_t)j] * c.vals[(size_t)j];
      }
      if (!(nrm > 0.0)) {
        c.oid = -1;
        continue;
      }
      if (eff) (*eff)[(size_t)t] = (c.lb - dot) / std::sqrt(nrm);
    }
  }
  return out;
}

// generateCutGMI's own rejections that need no tableau row (the rest is the engine's ok flag)
static bool gmi_candidate(const mvx_lp_api *api, const void *a, int j) {
  if (api->get_col_kind(a, j) == MVX_CV || api->get_col_stat(a, j) != MVX_BS) return false;
  const double f0 = getFract(api->get_col_prim(a, j));
  return !(f0 < 1e-6 || f0 > 1.0 - 1e-6);
}

// The lazy cut modes generate ONE cut per branching node (cut.cpp:20 only ever appends the last one): a window of 64
// nodes means 64 tableau-row reads and 64 O(m n) back-substitutions on the host, one node after the other, while the
// device idles -- that loop, not the LP solves, bounded the cut path (1.7 k nodes/s against 15 k without cuts).  Here the
// cuts of a whole round are made in one device pass (mvx_lp_api.gmi_cuts_many): for every node of the window that
// may branch, the column the host loop would settle on -- bug-compatible: the last basic integer column (gmi.cpp:18-27);
// repaired: the last column generateCutGMI would not reject out of hand -- and one launch pair for all of them.  Same
// bits as generateCut3 / generateCutGMI (tests/test_gpu_gmi.py); a node whose cut the engine rejects (ok = 0, zero
// norm) gets nullptr and goes through the host loop as before.
static std::vector<std::unique_ptr<CutContainer>> round_cuts(const mvx_lp_api *api, const std::vector<void *> &nodes, const std::vector<char> &wanted,
                                                              const mvx_bnb_params &prm, bool quirks) {
  std::vector<std::unique_ptr<CutContainer>> out(nodes.size());
  if (!api->gmi_cuts_many || prm.cut_strat == 0 || !prm.lazy_pool || (!quirks && prm.cut_select != 0)) return out;
  std::vector<const void *> ps;
  std::vector<int> cols;
  std::vector<size_t> slot;
  for (size_t w = 0; w < nodes.size(); w++) {
    if (!wanted[w]) continue;
    const void *a = nodes[w];
    const int na = api->get_num_cols(a);
    for (int j = na; j >= 1; j--) {
      const bool cand = quirks ? (api->get_col_kind(a, j) == MVX_IV && api->get_col_stat(a, j) == MVX_BS) : gmi_candidate(api, a, j);
      if (cand) {
        ps.push_back(a);
        cols.push_back(j);
        slot.push_back(w);
        break;
      }
    }
  }
  const int k = (int)ps.size();
  if (k < 2) return out; // a single cut: the host loop is as fast
  const int n = api->get_num_cols(ps[0]);
  std::vector<double> vals((size_t)k * (n + 1)), rhs((size_t)k);
  std::vector<int> ok((size_t)k, 0);
  const int grc = api->gmi_cuts_many(ps.data(), quirks ? 0 : 1, cols.data(), k, vals.data(), rhs.data(), ok.data());
  if (grc != 0) {
    if (std::getenv("MVX_BNB_TIMING")) std::fprintf(stderr, "round_cuts: gmi_cuts_many returned %d for %d cuts\n", grc, k);
    return out;
  }
  for (int t = 0; t < k; t++) {
    if (!ok[(size_t)t]) continue;
    const double *v = &vals[(size_t)t * (n + 1)];
    if (!quirks) { // generateCutGMI rejects a cut without coefficients
      double nrm = 0.0;
      for (int j = 1; j <= n; j++) nrm += v[j] * v[j];
      if (!(nrm > 0.0)) continue;
    }
    auto c = std::make_unique<CutContainer>();
    c->inds.resize((size_t)n + 1);
    c->vals.assign(v, v + n + 1);
    for (int j = 0; j <= n; j++) c->inds[(size_t)j] = j;
    c->lb = rhs[(size_t)t];
    c->oid = 0;
    out[slot[(size_t)t]] = std::move(c);
  }
  return out;
}

// Cut step of a branching node (bs.cpp:249-258), shared by both drivers: bug-compatible mode feeds the
// persistent pool and appends its last cut (cut.cpp:16-21); repaired mode appends this node's own GMI cuts,
// chosen by cut_select / -cf.
// Returns the number of rows appended; -1 when the bug-compatible path found its pool empty (nothing generated yet).
// `pre`: the one cut the lazy modes would generate for this node, already made by round_cuts (nullptr: make it here).
static double g_t_cut_scan = 0, g_t_cut_copy = 0, g_t_cut_add = 0; // MVX_BNB_TIMING: parts of add_node_cuts (window driver's thread only)
static double cut_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int add_node_cuts(const mvx_lp_api *api, void *a, const mvx_bnb_params &prm, bool quirks, CutPool &pool, const CutContainer *pre = nullptr) {
  if (prm.cut_strat == 0) return 0;
  const int na = api->get_num_cols(a);
  const bool dev = api->gmi_cuts != nullptr;
  if (quirks) {
    if (prm.lazy_pool) {
      // only the pool's LAST cut is ever added (cut.cpp:20): generate just that one
      const double t0 = cut_now();
      for (int j = na; j >= 1; j--) {
        if (api->get_col_kind(a, j) == MVX_IV && api->get_col_stat(a, j) == MVX_BS) {
          // one cut: the host loop (one row read, one back-substitution) is as fast as a device pass with its set-up
          const double t1 = cut_now();
          g_t_cut_scan += t1 - t0;
          pool.replaceLast(pre ? *pre : generateCut3(api, a, j));
          g_t_cut_copy += cut_now() - t1;
          break;
        }
      }
    } else if (dev) {
      std::vector<int> cols;
      for (int j = 1; j <= na; j++)
        if (api->get_col_kind(a, j) == MVX_IV && api->get_col_stat(a, j) == MVX_BS) cols.push_back(j); // gmi.cpp:18-27
      if (!cols.empty())
        for (auto &c : cuts_via_engine(api, a, false, cols, nullptr))
          if (c.oid != -1) pool.addToPool(std::move(c));
    } else {
      for (int j = 1; j <= na; j++) {
        CutContainer result = generateCut3(api, a, j);
        if (result.oid != -1) pool.addToPool(std::move(result));
      }
    }
    const double t2 = cut_now();
    const int rc = pool.addCutConstraint(a) < 0 ? -1 : 1;
    g_t_cut_add += cut_now() - t2;
    return rc;
  }
  std::vector<CutContainer> local;
  std::vector<double> eff;
  if (prm.cut_select == 0 && prm.lazy_pool && pre) {
    const double t1 = cut_now();
    local.push_back(*pre);
    eff.push_back(0.0);
    g_t_cut_copy += cut_now() - t1;
  } else if (prm.cut_select == 0 && prm.lazy_pool) {
    // only the last cut generated would be appended (cut.cpp:20): find it from the far end instead of
    // generating one cut per basic integer column (each is a tableau row read + an O(m n) back-substitution)
    for (int j = na; j >= 1 && local.empty(); j--) {
      double e = 0.0;
      CutContainer c = generateCutGMI(api, a, j, &e); // one cut: host loop (2 437 against 1 880 nodes/s through the device pass)
      if (c.oid != -1) {
        local.push_back(std::move(c));
        eff.push_back(e);
      }
    }
  } else if (dev) {
    std::vector<int> cols;
    for (int j = 1; j <= na; j++)
      if (gmi_candidate(api, a, j)) cols.push_back(j);
    if (!cols.empty()) {
      std::vector<double> e(cols.size(), 0.0);
      std::vector<CutContainer> all = cuts_via_engine(api, a, true, cols, &e);
      for (size_t t = 0; t < all.size(); t++)
        if (all[t].oid != -1) {
          local.push_back(std::move(all[t]));
          eff.push_back(e[t]);
        }
    }
  } else {
    for (int j = 1; j <= na; j++) {
      double e = 0.0;
      CutContainer c = generateCutGMI(api, a, j, &e);
      if (c.oid != -1) {
        local.push_back(std::move(c));
        eff.push_back(e);
      }
    }
  }
  if (local.empty()) return 0;
  int take = 1;
  if (prm.cut_select == 1) {
    take = (int)std::ceil(prm.cut_chance * (double)local.size());
    take = std::max(1, std::min(take, (int)local.size()));
  }
  std::vector<char> used(local.size(), 0);
  for (int t = 0; t < take; t++) {
    int best = -1;
    if (prm.cut_select == 0) best = (int)local.size() - 1; // cut.cpp:20: the last one
    else
      for (int q = 0; q < (int)local.size(); q++)
        if (!used[(size_t)q] && (best < 0 || eff[(size_t)q] > eff[(size_t)best])) best = q; // ties: first generated
    used[(size_t)best] = 1;
    const CutContainer &cc = local[(size_t)best];
    const double t2 = cut_now();
    const int index = api->add_rows(a, 1);
    api->set_mat_row(a, index, (int)cc.inds.size() - 1, cc.inds.data(), cc.vals.data());
    api->set_row_bnds(a, index, MVX_LO, cc.lb, 0);
    g_t_cut_add += cut_now() - t2;
  }
  return take;
}

int branchAndBound(const mvx_lp_api *api, void *prob, const mvx_bnb_params &prm, mvx_bnb_result *res) { // bs.cpp:54
  MVOLP::ParameterObj params(api, prob, prm);
  CutPool pool(api);
  Recorder rec;
  int id = 1; // util.h:17
  const bool quirks = prm.reference_quirks != 0;

  std::deque<std::shared_ptr<MVOLP::NodeData>> leafContainer;
  auto S1 = std::make_shared<MVOLP::NodeData>(api, prob, id); // bs.cpp:80
  S1->inital = true;
  rec.node(S1->oid, 0);
  leafContainer.push_back(S1);

  void *a = api->create_prob();                                 // bs.cpp:89
  const double sg = params.sense();
  double bestLower = -sg * std::numeric_limits<double>::infinity(); // bs.cpp:90
  const int n0 = api->get_num_cols(prob);
  std::vector<double> xbest((size_t)n0 + 1, 0.0);
  int incumbent_oid = 0, has_incumbent = 0, hit_limit = 0;
  int count = 0;

  while (!leafContainer.empty()) { // bs.cpp:96
    if (prm.max_nodes > 0 && count >= prm.max_nodes) {
      hit_limit = 1;
      break;
    }
    int index;
    std::shared_ptr<MVOLP::NodeData> node = params.pickNode(leafContainer, index); // bs.cpp:101
    api->erase_prob(a);                       // bs.cpp:114-115
    api->copy_prob(a, node->prob, MVX_OFF);   // bs.cpp:116
    solve(api, a, rec);                       // bs.cpp:117
    rec.emit(MVX_EV_PREGNANT, node->oid, api->get_obj_val(a), 0.0, 0, 0); // bs.cpp:119-129

    auto ret = printInfo(api, a, quirks); // bs.cpp:135|151
    const int status = ret.first;
    const std::vector<int> &vars = ret.second;
    if (node->inital) {
      if (status == -1) { // bs.cpp:139-143
        rec.prune[(size_t)node->oid] = MVOLP::FEAS;
        break;
      }
      if (status == 1) { // bs.cpp:144-149: leaves without recording the solution; repaired mode keeps it
        node->upperBound = api->get_obj_val(a);
        rec.bound[(size_t)node->oid] = node->upperBound;
        rec.prune[(size_t)node->oid] = MVOLP::INTG;
        if (!quirks) {
          bestLower = node->upperBound;
          has_incumbent = 1;
          incumbent_oid = node->oid;
          const int na = api->get_num_cols(a);
          for (int i = 1; i <= na && i <= n0; i++) xbest[(size_t)i] = api->get_col_prim(a, i);
        }
        break;
      }
    }
    node->upperBound = api->get_obj_val(a); // bs.cpp:156
    rec.bound[(size_t)node->oid] = node->upperBound;

    if (status == 1) { // prune by integrality, bs.cpp:158-193
      rec.prune[(size_t)node->oid] = MVOLP::INTG;
      rec.emit(MVX_EV_INTEGER, node->oid, node->upperBound, 0.0, 0, 0);
      if (sg * node->upperBound > sg * bestLower) {
        bestLower = node->upperBound;
        has_incumbent = 1;
        incumbent_oid = node->oid;
        const int na = api->get_num_cols(a);
        for (int i = 1; i <= na && i <= n0; i++) xbest[(size_t)i] = api->get_col_prim(a, i);
      }
      leafContainer.erase(leafContainer.begin() + index);
    } else if (status == -1) { // bs.cpp:194-209
      rec.prune[(size_t)node->oid] = MVOLP::FEAS;
      rec.emit(MVX_EV_INFEASIBLE, node->oid, 0.0, 0.0, 0, 0);
      leafContainer.erase(leafContainer.begin() + index);
    } else if (sg * api->get_obj_val(a) <= sg * bestLower) { // bs.cpp:210-223 (ties pruned)
      rec.prune[(size_t)node->oid] = MVOLP::BNDS;
      rec.emit(MVX_EV_FATHOMED, node->oid, 0.0, 0.0, 0, 0);
      leafContainer.erase(leafContainer.begin() + index);
    } else { // branch, bs.cpp:224-324
      double acc = 0;
      for (int i : vars)
        if (i != 0) acc += getFract(api->get_col_prim(a, i)); // bs.cpp:229-233
      leafContainer.erase(leafContainer.begin() + index);       // bs.cpp:247

      add_node_cuts(api, a, prm, quirks, pool); // bs.cpp:249-258
      const int pick = params.pickVar(vars);         // bs.cpp:260
      const double bound = api->get_col_prim(a, pick); // bs.cpp:261
      rec.emit(MVX_EV_BRANCHED, node->oid, node->upperBound, acc, (int)vars.size(), pick);

      auto S2 = std::make_shared<MVOLP::NodeData>(api, a, id); // bs.cpp:269-273
      auto S3 = std::make_shared<MVOLP::NodeData>(api, a, id);
      rec.node(S2->oid, node->oid);
      rec.node(S3->oid, node->oid);
      if (quirks) {
        api->set_col_bnds(S2->prob, pick, MVX_UP, 0, std::floor(bound)); // bs.cpp:274
      } else {
        const int t = api->get_col_type(a, pick);
        const double l = api->get_col_lb(a, pick);
        if (t == MVX_LO || t == MVX_DB || t == MVX_FX)
          api->set_col_bnds(S2->prob, pick, (l == std::floor(bound)) ? MVX_FX : MVX_DB, l, std::floor(bound));
        else
          api->set_col_bnds(S2->prob, pick, MVX_UP, 0, std::floor(bound));
      }
      if (quirks) {
        api->set_col_bnds(S3->prob, pick, MVX_LO, std::ceil(bound), 0); // bs.cpp:282
      } else {
        const int t = api->get_col_type(a, pick);
        const double u = api->get_col_ub(a, pick);
        if (t == MVX_UP || t == MVX_DB || t == MVX_FX)
          api->set_col_bnds(S3->prob, pick, (u == std::ceil(bound)) ? MVX_FX : MVX_DB, std::ceil(bound), u);
        else
          api->set_col_bnds(S3->prob, pick, MVX_LO, std::ceil(bound), 0);
      }
      // bs.cpp:279 and :287 solve two independent clones; an engine with a batch entry runs them
      // side by side (identical results), otherwise one after the other as the reference does
      if (api->simplex_batch) {
        void *pair[2] = {S2->prob, S3->prob};
        const int before = api->get_it_cnt(S2->prob) + api->get_it_cnt(S3->prob);
        api->simplex_batch(pair, 2, nullptr, nullptr);
        rec.pivots += api->get_it_cnt(S2->prob) + api->get_it_cnt(S3->prob) - before;
      } else {
        solve(api, S2->prob, rec);
        solve(api, S3->prob, rec);
      }
      S2->upperBound = api->get_obj_val(S2->prob);
      S3->upperBound = api->get_obj_val(S3->prob);
      rec.bound[(size_t)S2->oid] = S2->upperBound;
      rec.bound[(size_t)S3->oid] = S3->upperBound;

      leafContainer.push_back(S2); // bs.cpp:297-298
      leafContainer.push_back(S3);
      rec.emit(MVX_EV_CANDIDATE, S2->oid, S2->upperBound, 0.0, 0, 0); // bs.cpp:300-318
      rec.emit(MVX_EV_CANDIDATE, S3->oid, S3->upperBound, 0.0, 0, 0);
      if (count > prm.loop_limit) { // bs.cpp:320-323 calls std::exit(-1); here the run stops
        hit_limit = 1;
        count++;
        break;
      }
    }
    count++; // bs.cpp:326
  }
  leafContainer.clear();
  api->delete_prob(a);
  pack_result(res, rec, id, count, has_incumbent, bestLower, incumbent_oid, n0, xbest, hit_limit);
  return 0;
}

// Window mode: the front W nodes of the FIFO deque are solved together (one batched launch carries
// all of them), then bs.cpp's decisions are replayed in queue order.  With FIFO order children go to
// the back (bs.cpp:297-298) and no node is discarded unsolved, so the next W nodes the serial loop
// would pop are exactly the front W whatever their outcome: the tree, oids, events, pivot counts and
// incumbent are those of the node-at-a-time loop above.
int branchAndBoundWindow(const mvx_lp_api *api, void *prob, const mvx_bnb_params &prm, mvx_bnb_result *res) {
  MVOLP::ParameterObj params(api, prob, prm);
  Recorder rec;
  int id = 1;
  const bool quirks = prm.reference_quirks != 0;
  std::deque<std::shared_ptr<MVOLP::NodeData>> leafContainer;
  auto S1 = std::make_shared<MVOLP::NodeData>(api, prob, id);
  S1->inital = true;
  rec.node(S1->oid, 0);
  leafContainer.push_back(S1);
  const double sg = params.sense();
  double bestLower = -sg * std::numeric_limits<double>::infinity();
  const int n0 = api->get_num_cols(prob);
  std::vector<double> xbest((size_t)n0 + 1, 0.0);
  int incumbent_oid = 0, has_incumbent = 0, hit_limit = 0, count = 0;
  bool stop = false;

  struct Branch {
    size_t slot;
    std::shared_ptr<MVOLP::NodeData> S2, S3;
    int before2, before3;
  };
  // One round of the loop.  Its child solves (phase C) run on a worker thread while the main thread replays
  // the next window, whose nodes were solved rounds ago: a breadth-first queue is longer than the window
  // almost all the time.  Results do not depend on the overlap -- every LP is solved by the same calls on the
  // same state -- and the event stream is flushed round by round, in order.
  struct Round {
    std::vector<std::vector<mvx_bnb_event>> node_events;
    std::vector<Branch> branches;
    std::vector<void *> kids;
    std::vector<int> after1, repiv; // per kid: pivot count after its first solve; pivots of the re-solve done ahead (-1: none)
    std::vector<double> obj1;       // per kid: objective after its first solve (bs.cpp:280,288)
    std::future<void> fut;
    bool active = false;
  };
  // Rounds whose children are still being solved, oldest first.  Up to `depth` of them: while the slowest children of
  // one round finish (few batch slots busy), the next round's batch is already running on another batch context.
  std::deque<std::unique_ptr<Round>> flight;
  std::unordered_map<const void *, const Round *> inflight; // child handle -> the round that is solving it
  // Experiment knobs (off by default).  MVX_BNB_CHUNK=<c>: a replay's branchings leave as up to four rounds (chunks of at
  // least c branchings, in queue order), each one batched solve on its own worker; MVX_BNB_PREFIX=1: the next window
  // takes the queue's nodes only as far as their solves have ended, so one slow child holds back the nodes behind it in
  // its chunk and nothing else.  On the config-5 tree (a few tens of nodes wide, two thirds of a whole-round batch's
  // slot-launches idle behind its slowest LP: scripts/roundstats.py) this was measured SLOWER -- 2.86 s whole rounds,
  // 3.35 s with chunks of 8, 3.43 s with chunks of 2: more and smaller batches pay more fixed cost per batch, and the
  // dispatch of their tiny launches is what the GPU runs out of, not slots.  The wide tree does not care (104-117 ms).
  size_t depth = 2, chunk = (size_t)1 << 30;
  if (const char *e = std::getenv("MVX_BNB_DEPTH")) depth = (size_t)std::max(1, std::min(32, std::atoi(e)));
  if (const char *e = std::getenv("MVX_BNB_CHUNK")) chunk = (size_t)std::max(1, std::atoi(e));
  const bool prefix = std::getenv("MVX_BNB_PREFIX") && std::atoi(std::getenv("MVX_BNB_PREFIX")) != 0;
  CutPool pool(api); // persistent across nodes in bug-compatible mode (cut.h:15-23)
  const bool timing = std::getenv("MVX_BNB_TIMING") != nullptr;
  double tA = 0, tB = 0, tB_info = 0, tB_clone = 0, tB_cuts = 0, tB_rcuts = 0, tWait = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

  // wait for a round's children, book their results, flush the round's events
  auto finalize = [&](Round &R) {
    if (!R.active) return;
    const double t0 = now();
    if (R.fut.valid()) R.fut.get();
    tWait += now() - t0;
    for (size_t b = 0; b < R.branches.size(); b++) {
      Branch &br = R.branches[b];
      const size_t k2 = 2 * b, k3 = 2 * b + 1;
      rec.pivots += (R.after1[k2] - br.before2) + (R.after1[k3] - br.before3);
      br.S2->upperBound = R.obj1[k2];
      br.S3->upperBound = R.obj1[k3];
      br.S2->repiv = R.repiv[k2];
      br.S3->repiv = R.repiv[k3];
      rec.bound[(size_t)br.S2->oid] = br.S2->upperBound;
      rec.bound[(size_t)br.S3->oid] = br.S3->upperBound;
      rec.sink = &R.node_events[br.slot];
      rec.emit(MVX_EV_CANDIDATE, br.S2->oid, br.S2->upperBound, 0.0, 0, 0);
      rec.emit(MVX_EV_CANDIDATE, br.S3->oid, br.S3->upperBound, 0.0, 0, 0);
    }
    rec.sink = nullptr;
    for (auto &ne : R.node_events) rec.events.insert(rec.events.end(), ne.begin(), ne.end());
    for (void *k : R.kids) inflight.erase(k);
  };
  // rounds finish in the order they were started: the event stream is the serial one
  auto finalize_oldest = [&]() {
    if (flight.empty()) return;
    finalize(*flight.front());
    flight.pop_front();
  };
  auto finalize_through = [&](const Round *upto) {
    while (!flight.empty()) {
      const bool last = flight.front().get() == upto;
      finalize_oldest();
      if (last) break;
    }
  };

  while (!leafContainer.empty() && !stop) {
    if (prm.max_nodes > 0 && count >= prm.max_nodes) {
      hit_limit = 1;
      break;
    }
    double t0 = now();
    size_t W = std::min(leafContainer.size(), (size_t)prm.window);
    // The window may reach into the children that are still being solved: it ends in front of the first node whose
    // round has not finished (rounds are in queue order, so every older one has been booked by then), and waits
    // only when that node is the queue's head.  Same nodes in the same order as one by one -- only the grouping moves.
    for (size_t w = 0; w < W; w++) {
      auto it = inflight.find(leafContainer[w]->prob);
      if (it == inflight.end()) continue;
      const Round *R = it->second;
      if (prefix && w > 0 && R->fut.valid() && R->fut.wait_for(std::chrono::seconds(0)) != std::future_status::ready) {
        W = w;
        break;
      }
      finalize_through(R);
    }
    // A. solve the window (bs.cpp:114-117).  The reference copies the node's problem into the scratch
    // `a` and solves the copy; the node is discarded after this step either way, so its own clone is
    // solved in place here -- same state, one device-to-device clone fewer per node.  A node whose last
    // solve ended OPT, or whose re-solve was done ahead, goes through zero pivots and is not passed on.
    std::vector<void *> a(W), need;
    std::vector<int> before(W);
    for (size_t w = 0; w < W; w++) {
      a[w] = leafContainer[w]->prob;
      before[w] = api->get_it_cnt(a[w]);
      if (leafContainer[w]->repiv >= 0) continue;
      const int st = api->get_status(a[w]);
      if (st != MVX_OPT) need.push_back(a[w]);
    }
    if (!need.empty()) api->simplex_batch(need.data(), (int)need.size(), nullptr, nullptr); // none of them is in flight
    tA += now() - t0; t0 = now();
    // the lazy cut modes: the one cut of every node of the window that may branch, in one device pass (round_cuts);
    // printInfo of the window is taken ahead for that and reused by the replay
    std::vector<std::pair<int, std::vector<int>>> info;
    std::vector<std::unique_ptr<CutContainer>> pre;
    if (W >= 8) {
      // printInfo (util.cpp:414-473) of the whole window ahead of the replay, on a few threads: it reads every column
      // value of a node out of host mirrors that no cache holds yet (a window's nodes were solved rounds ago), ~15 us a
      // node when done one after the other inside the replay; the nodes are independent, the replay reuses the results
      info.resize(W);
      double ti = now();
      const size_t parts = std::min<size_t>(4, W / 4);
      std::vector<std::future<void>> fs;
      for (size_t part = 1; part < parts; part++)
        fs.push_back(std::async(std::launch::async, [&, part]() {
          for (size_t w = part * W / parts; w < (part + 1) * W / parts; w++) info[w] = printInfo(api, a[w], quirks);
        }));
      for (size_t w = 0; w < W / parts; w++) info[w] = printInfo(api, a[w], quirks);
      for (auto &f : fs) f.get();
      tB_info += now() - ti;
    }
    if (prm.cut_strat != 0 && prm.lazy_pool && api->gmi_cuts_many && !info.empty()) {
      std::vector<char> wanted(W, 0);
      for (size_t w = 0; w < W; w++) wanted[w] = info[w].first == 0; // neither infeasible nor integral: branches unless its bound prunes it
      const double ti = now();
      pre = round_cuts(api, a, wanted, prm, quirks);
      tB_cuts += now() - ti;
      tB_rcuts += now() - ti;
    }
    // B. replay in queue order
    Round cur;
    cur.node_events.resize(W);
    std::vector<Branch> &branches = cur.branches;
    size_t processed = 0;
    for (size_t w = 0; w < W; w++) {
      if (prm.max_nodes > 0 && count >= prm.max_nodes) {
        hit_limit = 1;
        stop = true;
        break;
      }
      std::shared_ptr<MVOLP::NodeData> node = leafContainer[w];
      void *aw = a[w];
      rec.sink = &cur.node_events[w];
      rec.pivots += (node->repiv >= 0) ? node->repiv : api->get_it_cnt(aw) - before[w];
      processed++;
      rec.emit(MVX_EV_PREGNANT, node->oid, api->get_obj_val(aw), 0.0, 0, 0);
      double ti = now();
      auto ret = info.empty() ? printInfo(api, aw, quirks) : std::move(info[w]);
      tB_info += now() - ti;
      const int status = ret.first;
      const std::vector<int> &vars = ret.second;
      if (node->inital) {
        if (status == -1) {
          rec.prune[(size_t)node->oid] = MVOLP::FEAS;
          stop = true;
          break;
        }
        if (status == 1) {
          node->upperBound = api->get_obj_val(aw);
          rec.bound[(size_t)node->oid] = node->upperBound;
          rec.prune[(size_t)node->oid] = MVOLP::INTG;
          if (!quirks) { // bs.cpp:144-149 leaves without recording the solution; repaired mode keeps it
            bestLower = node->upperBound;
            has_incumbent = 1;
            incumbent_oid = node->oid;
            for (int i = 1; i <= n0; i++) xbest[(size_t)i] = api->get_col_prim(aw, i);
          }
          stop = true;
          break;
        }
      }
      node->upperBound = api->get_obj_val(aw);
      rec.bound[(size_t)node->oid] = node->upperBound;
      if (status == 1) {
        rec.prune[(size_t)node->oid] = MVOLP::INTG;
        rec.emit(MVX_EV_INTEGER, node->oid, node->upperBound, 0.0, 0, 0);
        if (sg * node->upperBound > sg * bestLower) {
          bestLower = node->upperBound;
          has_incumbent = 1;
          incumbent_oid = node->oid;
          for (int i = 1; i <= n0; i++) xbest[(size_t)i] = api->get_col_prim(aw, i);
        }
      } else if (status == -1) {
        rec.prune[(size_t)node->oid] = MVOLP::FEAS;
        rec.emit(MVX_EV_INFEASIBLE, node->oid, 0.0, 0.0, 0, 0);
      } else if (sg * api->get_obj_val(aw) <= sg * bestLower) {
        rec.prune[(size_t)node->oid] = MVOLP::BNDS;
        rec.emit(MVX_EV_FATHOMED, node->oid, 0.0, 0.0, 0, 0);
      } else {
        double acc = 0;
        for (int i : vars)
          if (i != 0) acc += getFract(api->get_col_prim(aw, i));
        // cuts go onto the node's own problem (the serial driver's scratch copy `a`): both children inherit them.
        // The replay runs in queue order, so the persistent pool sees the nodes in bs.cpp's order.
        // bs.cpp:260-261 pick the variable and read its value behind the cut step; both are taken in front of it here: the
        // pick looks at the violated list and the root problem only, and appending a row leaves every other row's value
        // as it is -- but it marks the handle's solution mirrors stale, and reading one value afterwards is a device
        // export and a host round trip per branching node (~40 us, a tenth of the cut modes' run)
        const int pick = params.pickVar(vars);
        const double bound = api->get_col_prim(aw, pick);
        ti = now();
        add_node_cuts(api, aw, prm, quirks, pool, pre.empty() ? nullptr : pre[w].get());
        tB_cuts += now() - ti;
        rec.emit(MVX_EV_BRANCHED, node->oid, node->upperBound, acc, (int)vars.size(), pick);
        Branch br;
        br.slot = w;
        // bs.cpp:269-273 clones the solved node twice.  The node itself is dropped at the end of this
        // round, so the second child takes over its problem object instead of cloning it (same state,
        // one device-to-device tableau copy fewer per branching).
        const int t = api->get_col_type(aw, pick);
        const double l = api->get_col_lb(aw, pick), u = api->get_col_ub(aw, pick);
        double tc = now();
        br.S2 = std::make_shared<MVOLP::NodeData>(api, aw, id); // even oid (R), then odd (L): bs.cpp:43-52
        tB_clone += now() - tc;
        br.S3 = std::make_shared<MVOLP::NodeData>(api, aw, id, true);
        node->prob = nullptr;
        rec.node(br.S2->oid, node->oid);
        rec.node(br.S3->oid, node->oid);
        // same bounds as mvx_bnb_make_children
        if (quirks) {
          api->set_col_bnds(br.S2->prob, pick, MVX_UP, 0, std::floor(bound));
          api->set_col_bnds(br.S3->prob, pick, MVX_LO, std::ceil(bound), 0);
        } else {
          if (t == MVX_LO || t == MVX_DB || t == MVX_FX)
            api->set_col_bnds(br.S2->prob, pick, (l == std::floor(bound)) ? MVX_FX : MVX_DB, l, std::floor(bound));
          else
            api->set_col_bnds(br.S2->prob, pick, MVX_UP, 0, std::floor(bound));
          if (t == MVX_UP || t == MVX_DB || t == MVX_FX)
            api->set_col_bnds(br.S3->prob, pick, (u == std::ceil(bound)) ? MVX_FX : MVX_DB, std::ceil(bound), u);
          else
            api->set_col_bnds(br.S3->prob, pick, MVX_LO, std::ceil(bound), 0);
        }
        br.before2 = api->get_it_cnt(br.S2->prob);
        br.before3 = api->get_it_cnt(br.S3->prob);
        branches.push_back(br);
        leafContainer.push_back(br.S2); // the queue order bs.cpp:297-298 gives them
        leafContainer.push_back(br.S3);
        if (count > prm.loop_limit) { // bs.cpp:320-323
          hit_limit = 1;
          count++;
          stop = true;
          break;
        }
      }
      count++;
    }
    rec.sink = nullptr;
    tB += now() - t0;
    // C. every child of this round is an independent LP (bs.cpp:279,287): batched solves on worker threads; then, for
    // the children found infeasible (or unbounded), the re-solve bs.cpp:117 will ask for when they are
    // popped (it depends on nothing that happens in between), so that popping never has to solve.
    // The replay is cut into rounds of whole nodes: chunk k holds the events of its nodes and their branchings.
    {
      const size_t nb = branches.size();
      const size_t per = std::max(chunk, (nb + 3) / 4);
      const size_t nchunks = nb == 0 ? 1 : (nb + per - 1) / per;
      size_t ev_lo = 0;
      for (size_t ck = 0; ck < nchunks; ck++) {
        const size_t b_lo = ck * per, b_hi = std::min(nb, b_lo + per);
        const size_t ev_hi = (ck + 1 == nchunks) ? cur.node_events.size() : branches[b_hi - 1].slot + 1;
        auto part = std::make_unique<Round>();
        for (size_t e = ev_lo; e < ev_hi; e++) part->node_events.push_back(std::move(cur.node_events[e]));
        for (size_t bi = b_lo; bi < b_hi; bi++) {
          Branch br = branches[bi];
          br.slot -= ev_lo;
          part->kids.push_back(br.S2->prob);
          part->kids.push_back(br.S3->prob);
          part->branches.push_back(std::move(br));
        }
        ev_lo = ev_hi;
        part->active = true;
        while (flight.size() >= depth) finalize_oldest(); // at most `depth` rounds in flight; events stay in round order
        flight.push_back(std::move(part));
        Round *R = flight.back().get();
        if (R->kids.empty()) continue;
        for (void *k : R->kids) inflight.emplace(k, R);
        const size_t nk = R->kids.size();
        R->after1.assign(nk, 0);
        R->repiv.assign(nk, -1);
        R->obj1.assign(nk, 0.0);
        auto solve_range = [api, R](size_t lo, size_t hi) {
          if (hi <= lo) return;
          api->simplex_batch(R->kids.data() + lo, (int)(hi - lo), nullptr, nullptr);
          std::vector<void *> again;
          std::vector<size_t> idx;
          for (size_t k = lo; k < hi; k++) {
            R->after1[k] = api->get_it_cnt(R->kids[k]);
            R->obj1[k] = api->get_obj_val(R->kids[k]);
            const int st1 = api->get_status(R->kids[k]);
            if (st1 == MVX_NOFEAS || st1 == MVX_UNBND) { // a re-solve of these may pivot on (fresh devex weights)
              again.push_back(R->kids[k]);
              idx.push_back(k);
            }
          }
          if (!again.empty()) {
            api->simplex_batch(again.data(), (int)again.size(), nullptr, nullptr);
            for (size_t t = 0; t < idx.size(); t++) R->repiv[idx[t]] = api->get_it_cnt(again[t]) - R->after1[idx[t]];
          }
        };
        // a wide chunk goes through the engine's batch entry in two halves on two threads: the host side of one batch
        // (control-block uploads, polls, result mirrors) overlaps the kernels of the other
        const bool two = nk >= 32 && !std::getenv("MVX_BNB_ONE_WORKER");
        auto work = [solve_range, nk, two]() {
          if (!two) {
            solve_range(0, nk);
            return;
          }
          const size_t half = (nk / 2 + 1) & ~(size_t)1; // siblings stay together
          auto other = std::async(std::launch::async, solve_range, half, nk);
          solve_range(0, half);
          other.get();
        };
        if (prm.window > 1 && !std::getenv("MVX_BNB_SYNC")) R->fut = std::async(std::launch::async, work);
        else work();
      }
    }
    leafContainer.erase(leafContainer.begin(), leafContainer.begin() + (long)processed);
  }
  while (!flight.empty()) finalize_oldest();
  if (timing) {
    std::fprintf(stderr, "add_node_cuts: column scan %.1f ms, cut copy/generation %.1f ms, row append %.1f ms\n", g_t_cut_scan * 1e3, g_t_cut_copy * 1e3, g_t_cut_add * 1e3);
    g_t_cut_scan = g_t_cut_copy = g_t_cut_add = 0;
  }
  if (timing)
    std::fprintf(stderr, "bnb window timing: A %.1f ms  B %.1f ms (printInfo %.1f, clone %.1f, cuts %.1f of which the round's device pass %.1f)  waiting for child solves %.1f ms\n", tA * 1e3,
                 tB * 1e3, tB_info * 1e3, tB_clone * 1e3, tB_cuts * 1e3, tB_rcuts * 1e3, tWait * 1e3);
  leafContainer.clear();
  pack_result(res, rec, id, count, has_incumbent, bestLower, incumbent_oid, n0, xbest, hit_limit);
  return 0;
}

// ---- the gfx950 engine's table ----
const mvx_lp_api g_hip_api = {
    []() -> void * { return mvx_create_prob(); },
    [](void *P) { mvx_erase_prob((mvx_prob *)P); },
    [](void *P) { mvx_delete_prob((mvx_prob *)P); },
    [](void *d, const void *s, int names) { mvx_copy_prob((mvx_prob *)d, (const mvx_prob *)s, names); },
    [](void *P, int nrs) { return mvx_add_rows((mvx_prob *)P, nrs); },
    [](void *P, int i, int len, const int *ind, const double *val) { mvx_set_mat_row((mvx_prob *)P, i, len, ind, val); },
    [](void *P, int i, int t, double lb, double ub) { mvx_set_row_bnds((mvx_prob *)P, i, t, lb, ub); },
    [](void *P, int j, int t, double lb, double ub) { mvx_set_col_bnds((mvx_prob *)P, j, t, lb, ub); },
    [](void *P, const void *parm) { return mvx_simplex((mvx_prob *)P, (const mvx_smcp *)parm); },
    [](const void *P) { return mvx_get_status((const mvx_prob *)P); },
    [](const void *P) { return mvx_get_obj_val((const mvx_prob *)P); },
    [](const void *P, int j) { return mvx_get_obj_coef((const mvx_prob *)P, j); },
    [](const void *P, int j) { return mvx_get_col_prim((const mvx_prob *)P, j); },
    [](const void *P) { return mvx_get_num_rows((const mvx_prob *)P); },
    [](const void *P) { return mvx_get_num_cols((const mvx_prob *)P); },
    [](const void *P, int j) { return mvx_get_col_kind((const mvx_prob *)P, j); },
    [](const void *P, int j) { return mvx_get_col_stat((const mvx_prob *)P, j); },
    [](const void *P, int i) { return mvx_get_row_stat((const mvx_prob *)P, i); },
    [](const void *P, int i) { return mvx_get_row_ub((const mvx_prob *)P, i); },
    [](const void *P, int i) { return mvx_get_row_lb((const mvx_prob *)P, i); },
    [](const void *P, int j) { return mvx_get_col_ub((const mvx_prob *)P, j); },
    [](const void *P, int j) { return mvx_get_col_lb((const mvx_prob *)P, j); },
    [](const void *P, int j) { return mvx_get_col_type((const mvx_prob *)P, j); },
    [](const void *P, int i, int *ind, double *val) { return mvx_get_mat_row((const mvx_prob *)P, i, ind, val); },
    [](const void *P, int k, int *ind, double *val) { return mvx_eval_tab_row((const mvx_prob *)P, k, ind, val); },
    [](const void *P) { return mvx_get_it_cnt((const mvx_prob *)P); },
    [](void **probs, int count, const void *parm, int *rcs) {
      return mvx_simplex_batch((mvx_prob **)probs, count, (const mvx_smcp *)parm, rcs);
    },
    [](const void *P) { return mvx_get_obj_dir((const mvx_prob *)P); },
    [](const void *P, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok) {
      return mvx_gmi_cuts((const mvx_prob *)P, repaired, cols, count, vals, rhs, ok);
    },
    [](const void *const *Ps, int repaired, const int *cols, int count, double *vals, double *rhs, int *ok) {
      return mvx_gmi_cuts_many((const mvx_prob *const *)Ps, repaired, cols, count, vals, rhs, ok);
    },
    [](const void *P, double *x) { mvx_get_col_prim_all((const mvx_prob *)P, x); },
};

} // namespace

extern "C" {

const mvx_lp_api *mvx_hip_lp_api(void) { return &g_hip_api; }

void mvx_bnb_default_params(mvx_bnb_params *p) {
  p->var_strat = 0;  // util.h:65
  p->node_strat = 0; // util.h:66
  p->cut_strat = 0;  // util.h:67
  p->cut_chance = 0.0;
  p->loop_limit = 200000; // bs.cpp:320
  p->max_nodes = 0;
  p->reference_quirks = 1;
  p->lazy_pool = 1;
  p->cut_select = 0;
  p->window = 64;
}

int mvx_branchAndBound(const mvx_lp_api *api, void *prob, const mvx_bnb_params *params, mvx_bnb_result *res) {
  mvx_bnb_params dflt;
  if (!params) {
    mvx_bnb_default_params(&dflt);
    params = &dflt;
  }
  if (!api) api = &g_hip_api;
  if (api->simplex_batch && params->node_strat == 0 && params->window > 1)
    return branchAndBoundWindow(api, prob, *params, res);
  return branchAndBound(api, prob, *params, res);
}

void mvx_bnb_free_result(mvx_bnb_result *res) {
  std::free(res->parent);
  std::free(res->prune);
  std::free(res->node_bound);
  std::free(res->events);
  std::free(res->x);
  std::memset(res, 0, sizeof(*res));
}

double mvx_getFract(double x) { return getFract(x); }

int mvx_generateCutGMI(const mvx_lp_api *api, const void *prob, int j, int *inds, double *vals, double *lb, double *efficacy) {
  double e = 0.0;
  CutContainer c = generateCutGMI(api ? api : &g_hip_api, prob, j, &e);
  if (c.oid == -1) return -1;
  std::memcpy(inds, c.inds.data(), c.inds.size() * sizeof(int));
  std::memcpy(vals, c.vals.data(), c.vals.size() * sizeof(double));
  *lb = c.lb;
  *efficacy = e;
  return 0;
}

int mvx_bnb_classify(const mvx_lp_api *api, const void *prob, const void *root, int quirks, int var_strat, double *out) {
  if (!api) api = &g_hip_api;
  auto ret = printInfo(api, prob, quirks != 0);
  mvx_bnb_params p;
  mvx_bnb_default_params(&p);
  p.var_strat = var_strat;
  MVOLP::ParameterObj params(api, root, p);
  double acc = 0;
  for (int i : ret.second)
    if (i != 0) acc += getFract(api->get_col_prim(prob, i)); // bs.cpp:229-233
  out[0] = (double)ret.first;
  out[1] = api->get_obj_val(prob);
  out[2] = (double)ret.second.size();
  out[3] = acc;
  out[4] = ret.second.empty() ? 0.0 : (double)params.pickVar(ret.second);
  return 0;
}

int mvx_bnb_make_children(const mvx_lp_api *api, const void *a, int pick, int quirks, void *S2, void *S3) {
  if (!api) api = &g_hip_api;
  const double bound = api->get_col_prim(a, pick); // bs.cpp:261
  api->copy_prob(S2, a, MVX_ON);                   // NodeData(a), util.cpp:33-34
  api->copy_prob(S3, a, MVX_ON);
  if (quirks) {
    api->set_col_bnds(S2, pick, MVX_UP, 0, std::floor(bound)); // bs.cpp:274
    api->set_col_bnds(S3, pick, MVX_LO, std::ceil(bound), 0);  // bs.cpp:282
  } else {
    const int t = api->get_col_type(a, pick);
    const double l = api->get_col_lb(a, pick), u = api->get_col_ub(a, pick);
    if (t == MVX_LO || t == MVX_DB || t == MVX_FX)
      api->set_col_bnds(S2, pick, (l == std::floor(bound)) ? MVX_FX : MVX_DB, l, std::floor(bound));
    else
      api->set_col_bnds(S2, pick, MVX_UP, 0, std::floor(bound));
    if (t == MVX_UP || t == MVX_DB || t == MVX_FX)
      api->set_col_bnds(S3, pick, (u == std::ceil(bound)) ? MVX_FX : MVX_DB, std::ceil(bound), u);
    else
      api->set_col_bnds(S3, pick, MVX_LO, std::ceil(bound), 0);
  }
  return 0;
}

int mvx_bnb_node_cuts(const mvx_lp_api *api, void *a, const mvx_bnb_params *params) {
  if (!api) api = &g_hip_api;
  CutPool pool(api); // this node's cuts only: a caller that needs bs.cpp:73's pool across nodes keeps the order itself
  return add_node_cuts(api, a, *params, params->reference_quirks != 0, pool);
}

int mvx_printInfo(const mvx_lp_api *api, const void *prob, int quirks, int *violated, int *nviolated) {
  auto r = printInfo(api ? api : &g_hip_api, prob, quirks != 0);
  *nviolated = (int)r.second.size();
  for (size_t k = 0; k < r.second.size(); k++) violated[k] = r.second[k];
  return r.first;
}

int mvx_generateCut3(const mvx_lp_api *api, const void *prob, int j, int *inds, double *vals, double *lb) {
  CutContainer c = generateCut3(api ? api : &g_hip_api, prob, j);
  if (c.oid == -1) return -1;
  std::memcpy(inds, c.inds.data(), c.inds.size() * sizeof(int));
  std::memcpy(vals, c.vals.data(), c.vals.size() * sizeof(double));
  *lb = c.lb;
  return 0;
}

} // extern "C"
