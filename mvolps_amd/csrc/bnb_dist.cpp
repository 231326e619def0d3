// bnb_dist.cpp -- mvx_branchAndBound_dist: the loop of /root/reference/bs.cpp:96-327 run in lock-step on every rank,
// node LPs farmed over the ranks (include/mvx_dist.h).  One round =
//   A. the owners solve the front window of the FIFO deque together (bs.cpp:114-117) and classify their nodes
//      (printInfo / pickVar, bs.cpp:135-156,260); one MAX all-reduce publishes (status, objective, ...);
//   B. every rank replays bs.cpp's decisions in queue order (incumbent bs.cpp:172-174, pruning bs.cpp:199-217,
//      branching bs.cpp:225-244), which fixes the branch list and the child oids;
//   C. the owners append their nodes' GMI rows (bs.cpp:249-258), create and solve the children (bs.cpp:261-288);
//      one MAX all-reduce publishes their bounds;
//   D. every child gets an owner.  It stays on its parent's rank -- its tableau is already there -- unless that rank's
//      share of the window the child will be popped in is full; only such a child travels (mvx_image_api).
// Nothing here touches a device: the LP engine is behind mvx_lp_api, the transport behind mvx_comm.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/mvx_dist.h"

namespace {

constexpr double NEG_INF = -std::numeric_limits<double>::infinity();
constexpr double POS_INF = std::numeric_limits<double>::infinity();
enum { INTG = 0, FEAS = 1, BNDS = 3, NONE = 4 }; // util.h:27

struct Node {
  int oid, owner;
  double upper;
  bool inital;
  // what the window step (bs.cpp:114-117 + printInfo) will find when this node is popped: status, objective, number of
  // violated columns, their fractional sum, the pick, and the pivots of the pop-time re-solve.  A child's creator works it
  // out right after solving the child -- nothing touches a node's problem between its creation and its pop -- and it
  // travels in the round's ONE all-reduce next to the child's bound, so popping a window needs no collective at all.
  bool known = false;
  double cls[6] = {0, 0, 0, 0, 0, 0};
  int solver = 0; // the rank that solved it (holds its solution if it turns out integral)
};

struct Branch {
  int oid, owner, s2, s3, pick;
  size_t ev_at; // where this node's candidate events go (after its branched event)
};

int branch_direction(int oid) { // bs.cpp:43-52
  if (oid <= 1) return 0;
  return oid % 2 == 0 ? 1 : 2;
}

template <typename T>
T *dup(const std::vector<T> &v) {
  T *p = (T *)std::malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
  if (!v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

struct Tree {
  std::vector<int> parent{0, 0}, prune{NONE, NONE};
  std::vector<double> bound{POS_INF, POS_INF};
  void add(int oid, int pid) {
    if ((int)parent.size() <= oid) {
      parent.resize((size_t)oid + 1, 0);
      prune.resize((size_t)oid + 1, NONE);
      bound.resize((size_t)oid + 1, POS_INF);
    }
    parent[(size_t)oid] = pid;
  }
};

struct Ev {
  int type, oid;
  double f6, f7;
  int f8, pick;
};

int solve_many(const mvx_lp_api *api, std::vector<void *> &hs) {
  if (hs.empty()) return 0;
  if (api->simplex_batch && hs.size() > 1) return api->simplex_batch(hs.data(), (int)hs.size(), nullptr, nullptr);
  for (void *h : hs) api->simplex(h, nullptr); // the reference ignores the return code
  return 0;
}

} // namespace

extern "C" void mvx_dist_default_params(mvx_dist_params *p) {
  p->per_rank = 64;
  p->slack = -1;
  p->roundrobin = 0;
}

extern "C" int mvx_branchAndBound_dist(const mvx_lp_api *api, const mvx_image_api *img, void *root, const mvx_bnb_params *params_in,
                                       const mvx_dist_params *dist_in, const mvx_comm *comm, mvx_bnb_result *res, mvx_dist_stats *stats_out) {
  if (!api) api = mvx_hip_lp_api();
  if (!img) img = mvx_hip_image_api();
  mvx_bnb_params prm;
  if (params_in) prm = *params_in;
  else mvx_bnb_default_params(&prm);
  mvx_dist_params dp;
  if (dist_in) dp = *dist_in;
  else mvx_dist_default_params(&dp);
  if (prm.node_strat != 0 || dp.per_rank < 1) return MVX_EFAIL;
  const int rank = comm ? comm->rank : 0, world = comm ? comm->size : 1;
  const int per_rank = dp.per_rank;
  const int slack = dp.slack >= 0 ? dp.slack : std::max(1, per_rank / 4);
  const int quirks = prm.reference_quirks, var_strat = prm.var_strat, max_nodes = prm.max_nodes, loop_limit = prm.loop_limit;
  const bool cuts = prm.cut_strat != 0;
  long long n_allreduce = 0;
  auto allreduce_max = [&](std::vector<double> &v) -> int {
    if (!(comm && world > 1)) return 0;
    n_allreduce++;
    return comm->allreduce_max(comm->ctx, v.data(), v.size());
  };

  Tree tree;
  std::vector<Ev> events;
  std::deque<Node> queue;
  queue.push_back(Node{1, 0, POS_INF, true});
  std::unordered_map<int, void *> local; // nodes this rank holds
  if (rank == 0) {
    void *s1 = api->create_prob();
    api->copy_prob(s1, root, MVX_ON); // S1 = NodeData(prob), bs.cpp:80
    local[1] = s1;
  }
  int next_id = 2;
  long long child_seq = 0;
  // bs.cpp:172,210 compare as a maximiser; the repaired mode turns the compares round for a minimisation problem
  const double sg = (!quirks && api->get_obj_dir && api->get_obj_dir(root) == MVX_MIN) ? -1.0 : 1.0;
  double best_lower = -sg * POS_INF;
  int has_inc = 0, inc_oid = 0, inc_owner = 0;
  std::map<int, std::vector<double>> x_keep;
  int count = 0, hit_limit = 0;
  long long total_pivots = 0;
  const int n0 = api->get_num_cols(root);
  bool stop_all = false;
  const long long W_full = (long long)world * per_rank;
  long long round_no = 0;
  std::map<long long, std::vector<int>> win_load; // absolute window number -> nodes per rank already placed in it
  mvx_dist_stats stats{0, 0, 0, 0, 0};
  int rc = 0;
  int late_err = 0; // a failure one rank met AFTER an exchange (unpacking an image): it rides in the next round's first
                    // all-reduce (or in one of its own behind the loop), so that every rank stops in the same place

  auto drop = [&](int oid) {
    auto it = local.find(oid);
    if (it != local.end()) {
      api->delete_prob(it->second);
      local.erase(it);
    }
  };

  while (!queue.empty() && !stop_all && rc == 0) {
    if (max_nodes > 0 && count >= max_nodes) {
      hit_limit = 1;
      break;
    }
    const int W = (int)std::min<long long>((long long)queue.size(), W_full);

    // ---- A. the window (bs.cpp:114-117, printInfo bs.cpp:135|151).  Every node but the root was classified by the rank
    // that created it (section C) and carries the result; only a window that holds a node not yet classified is solved
    // here and agreed on by an all-reduce -- every rank sees the same queue, so every rank takes the same branch.
    std::vector<double> A((size_t)W * 6 + 1, NEG_INF);
    bool all_known = true;
    for (int w = 0; w < W; w++) all_known = all_known && queue[(size_t)w].known;
    if (all_known) {
      for (int w = 0; w < W; w++)
        for (int t = 0; t < 6; t++) A[(size_t)w * 6 + (size_t)t] = queue[(size_t)w].cls[t];
    } else {
      A[(size_t)W * 6] = (double)late_err; // last entry: a rank carries a failure from the last round
      if (!late_err) {
        // bs.cpp:114-116 copies the node's problem into a scratch and solves the copy; the node is discarded after
        // this round either way, so its own clone is solved in place
        std::vector<void *> hs;
        std::vector<int> ws, before;
        for (int w = 0; w < W; w++) {
          const Node &nd = queue[(size_t)w];
          if (nd.owner != rank) continue;
          void *a = local.at(nd.oid);
          hs.push_back(a);
          ws.push_back(w);
          before.push_back(api->get_it_cnt(a));
        }
        solve_many(api, hs);
        for (size_t k = 0; k < hs.size(); k++) {
          double out[5];
          mvx_bnb_classify(api, hs[k], root, quirks, var_strat, out);
          double *row = &A[(size_t)ws[k] * 6];
          for (int t = 0; t < 5; t++) row[t] = out[t];
          row[5] = (double)(api->get_it_cnt(hs[k]) - before[k]);
          if (out[0] == 1.0) {
            std::vector<double> x((size_t)n0);
            for (int j = 1; j <= n0; j++) x[(size_t)j - 1] = api->get_col_prim(hs[k], j);
            x_keep[queue[(size_t)ws[k]].oid] = std::move(x);
          }
        }
      }
      for (int w = 0; w < W; w++) queue[(size_t)w].solver = queue[(size_t)w].owner;
      if ((rc = allreduce_max(A)) != 0) break;
      if (A[(size_t)W * 6] > 0.0) {
        rc = MVX_EFAIL;
        break;
      }
    }

    // ---- B. replay the serial decisions in queue order
    std::vector<Branch> branch_list;
    int processed = 0;
    for (int w = 0; w < W; w++) {
      if (max_nodes > 0 && count >= max_nodes) {
        hit_limit = 1;
        stop_all = true;
        break;
      }
      Node &nd = queue[(size_t)w];
      const double *row = &A[(size_t)w * 6];
      const int st = (int)row[0], nv = (int)row[2], pick = (int)row[4];
      const double obj = row[1], acc = row[3];
      total_pivots += (long long)row[5];
      events.push_back(Ev{MVX_EV_PREGNANT, nd.oid, obj, 0.0, 0, 0});
      processed++;
      if (nd.inital) {
        if (st == -1) { // bs.cpp:139-143
          tree.prune[(size_t)nd.oid] = FEAS;
          stop_all = true;
          break;
        }
        if (st == 1) { // bs.cpp:144-149 (leaves without recording the solution; repaired mode keeps it)
          nd.upper = obj;
          tree.bound[(size_t)nd.oid] = obj;
          tree.prune[(size_t)nd.oid] = INTG;
          if (!quirks) {
            best_lower = obj;
            has_inc = 1;
            inc_oid = nd.oid;
            inc_owner = nd.solver;
          }
          stop_all = true;
          break;
        }
      }
      nd.upper = obj;
      tree.bound[(size_t)nd.oid] = obj;
      if (st == 1) {
        tree.prune[(size_t)nd.oid] = INTG;
        events.push_back(Ev{MVX_EV_INTEGER, nd.oid, obj, 0.0, 0, 0});
        if (sg * obj > sg * best_lower) {
          best_lower = obj;
          has_inc = 1;
          inc_oid = nd.oid;
          inc_owner = nd.solver;
        }
      } else if (st == -1) {
        tree.prune[(size_t)nd.oid] = FEAS;
        events.push_back(Ev{MVX_EV_INFEASIBLE, nd.oid, 0.0, 0.0, 0, 0});
      } else if (sg * obj <= sg * best_lower) {
        tree.prune[(size_t)nd.oid] = BNDS;
        events.push_back(Ev{MVX_EV_FATHOMED, nd.oid, 0.0, 0.0, 0, 0});
      } else {
        const int s2 = next_id, s3 = next_id + 1;
        next_id += 2;
        tree.add(s2, nd.oid);
        tree.add(s3, nd.oid);
        events.push_back(Ev{MVX_EV_BRANCHED, nd.oid, obj, acc, nv, pick});
        branch_list.push_back(Branch{nd.oid, nd.owner, s2, s3, pick, events.size()});
        if (count > loop_limit) { // bs.cpp:320-323
          hit_limit = 1;
          count++;
          stop_all = true;
          break;
        }
      }
      count++;
    }

    // ---- C. owners create and solve the children (bs.cpp:269-288)
    const size_t nb = branch_list.size();
    constexpr size_t CW = 16; // per branching: both bounds, pivots, image bytes, then 6 + 6 for the two children's own window step
    std::vector<double> Cv((nb + 1) * CW, NEG_INF); // last row: [0] = a rank met a node without a cut, [1] = a rank carries a failure
    std::unordered_map<int, void *> fresh;
    {
      std::vector<void *> kids;
      std::vector<size_t> ks;
      std::vector<int> before;
      bool nocut = false;
      for (size_t k = 0; k < nb && !nocut && !late_err; k++) {
        const Branch &b = branch_list[k];
        if (b.owner != rank) continue;
        void *a = local.at(b.oid);
        if (cuts && mvx_bnb_node_cuts(api, a, &prm) < 0) { // bs.cpp:249-258
          nocut = true;
          break;
        }
        void *S2 = api->create_prob(), *S3 = api->create_prob();
        mvx_bnb_make_children(api, a, b.pick, quirks, S2, S3); // bs.cpp:261-282
        fresh[b.s2] = S2;
        fresh[b.s3] = S3;
        kids.push_back(S2);
        kids.push_back(S3);
        ks.push_back(k);
        before.push_back(api->get_it_cnt(S2) + api->get_it_cnt(S3));
      }
      if (late_err) Cv[nb * CW + 1] = 1.0;
      if (nocut) Cv[nb * CW] = 1.0;
      else if (!late_err) {
        // every child of this round is an independent LP (bs.cpp:279,287): solve them together
        solve_many(api, kids);
        for (size_t t = 0; t < ks.size(); t++) {
          void *S2 = kids[2 * t], *S3 = kids[2 * t + 1];
          double *row = &Cv[ks[t] * CW];
          row[0] = api->get_obj_val(S2);
          row[1] = api->get_obj_val(S3);
          row[2] = (double)(api->get_it_cnt(S2) + api->get_it_cnt(S3) - before[t]);
        }
        // ... and their own window step right away: the solve bs.cpp:117 repeats when a node is popped (no pivots unless
        // the first one ended infeasible or unbounded) and printInfo, which is all a later round needs to know of them
        std::vector<int> before2(kids.size());
        for (size_t i = 0; i < kids.size(); i++) before2[i] = api->get_it_cnt(kids[i]);
        solve_many(api, kids);
        for (size_t t = 0; t < ks.size(); t++) {
          double *row = &Cv[ks[t] * CW];
          for (int side = 0; side < 2; side++) {
            void *S = kids[2 * t + (size_t)side];
            double out[5];
            mvx_bnb_classify(api, S, root, quirks, var_strat, out);
            double *cl = row + 4 + 6 * side;
            for (int u = 0; u < 5; u++) cl[u] = out[u];
            cl[5] = (double)(api->get_it_cnt(S) - before2[2 * t + (size_t)side]);
            if (out[0] == 1.0) {
              std::vector<double> x((size_t)n0);
              for (int j = 1; j <= n0; j++) x[(size_t)j - 1] = api->get_col_prim(S, j);
              x_keep[side == 0 ? branch_list[ks[t]].s2 : branch_list[ks[t]].s3] = std::move(x);
            }
          }
          row[3] = (double)img->pack_size(kids[2 * t], root); // after the window step: what travels is the node as popped
        }
      }
    }
    if ((rc = allreduce_max(Cv)) != 0) {
      for (auto &kv : fresh) api->delete_prob(kv.second);
      break;
    }
    if (Cv[nb * CW + 1] == 1.0 || Cv[nb * CW] == 1.0) {
      for (auto &kv : fresh) api->delete_prob(kv.second);
      rc = Cv[nb * CW + 1] == 1.0 ? MVX_EFAIL : MVX_EDIST_NOCUT;
      break;
    }

    // ---- D. publish the children, give each an owner, migrate the ones that change ranks
    const long long base = (long long)queue.size() - processed; // nodes that stay queued after this round's pops
    const long long n_kids = 2 * (long long)nb;
    const long long spread_cap = (base + n_kids <= W_full) ? (base + n_kids + world - 1) / world : -1; // whole queue fits one window
    // Which children leave their parent's rank.  Per future window (rounds pop W_full nodes each from the front, so a
    // child's window follows from its queue position) and per parent rank: the children beyond that rank's remaining
    // share are taken EVENLY out of the run, not off its tail.  Under FIFO order a node's children sit side by side,
    // so ranks own runs of consecutive queue positions that double every level; thinning a run evenly interleaves the
    // ranks again, and the doubled runs of the next levels fit their windows.
    std::vector<int> kid_owner((size_t)n_kids, 0);
    if (!dp.roundrobin) {
      std::map<std::pair<long long, int>, std::vector<long long>> groups;
      for (size_t k = 0; k < nb; k++)
        for (int t = 0; t < 2; t++) {
          const long long kn = 2 * (long long)k + t;
          const long long wno = round_no + 1 + (base + kn) / W_full;
          groups[{wno, branch_list[k].owner}].push_back(kn);
          kid_owner[(size_t)kn] = branch_list[k].owner;
        }
      std::vector<std::pair<long long, long long>> movers; // (child number, window)
      for (auto &g : groups) {
        const long long wno = g.first.first;
        const int r = g.first.second;
        std::vector<long long> &kids = g.second;
        std::vector<int> &load = win_load[wno];
        if (load.empty()) load.assign((size_t)world, 0);
        const long long cap = spread_cap >= 0 ? spread_cap : per_rank + slack;
        const long long len = (long long)kids.size();
        const long long keep = std::max<long long>(0, std::min<long long>(len, cap - load[(size_t)r]));
        const long long move = len - keep;
        load[(size_t)r] += (int)keep;
        for (long long t = 0; t < move; t++) movers.push_back({kids[(size_t)((2 * t + 1) * len / (2 * move))], wno}); // evenly spaced
      }
      std::sort(movers.begin(), movers.end());
      for (auto &mv : movers) {
        std::vector<int> &load = win_load[mv.second];
        int dst = 0;
        for (int r = 1; r < world; r++)
          if (load[(size_t)r] < load[(size_t)dst]) dst = r;
        load[(size_t)dst]++;
        kid_owner[(size_t)mv.first] = dst;
      }
    }
    std::vector<mvx_xfer> sends, recvs;
    std::vector<int> recv_oid, send_oid;
    std::vector<std::pair<size_t, Ev>> cand;
    long long kid_no = 0, moved_this_round = 0;
    int lerr = 0; // a failure that only this rank sees (device memory for an image, pack): agreed on below, before any
                  // point-to-point transfer is posted -- a rank that walked away alone would leave its peers waiting
    for (size_t k = 0; k < nb; k++) {
      const Branch &b = branch_list[k];
      const double *row = &Cv[k * CW];
      const double ub[2] = {row[0], row[1]};
      total_pivots += (long long)row[2];
      const size_t nbytes = (size_t)row[3];
      const int oids[2] = {b.s2, b.s3};
      for (int t = 0; t < 2; t++) {
        const int oid = oids[t];
        const int owner = dp.roundrobin ? (int)(child_seq % world) : kid_owner[(size_t)kid_no];
        child_seq++;
        kid_no++;
        stats.children++;
        tree.bound[(size_t)oid] = ub[t];
        Node kid{oid, owner, ub[t], false};
        kid.known = true;
        for (int u = 0; u < 6; u++) kid.cls[u] = row[4 + 6 * t + u];
        kid.solver = b.owner;
        queue.push_back(kid);
        cand.push_back({b.ev_at, Ev{MVX_EV_CANDIDATE, oid, ub[t], 0.0, 0, 0}});
        if (owner != b.owner) {
          stats.migrated++;
          stats.migrated_bytes += (long long)nbytes;
          moved_this_round++;
        }
        if (b.owner == rank && owner == rank) {
          local[oid] = fresh.at(oid);
          fresh.erase(oid);
        } else if (b.owner == rank) {
          void *buf = lerr ? nullptr : img->buf_alloc(nbytes);
          if (!buf || img->pack(fresh.at(oid), root, buf) != 0) lerr = 1;
          sends.push_back(mvx_xfer{buf, nbytes, owner});
          send_oid.push_back(oid);
        } else if (owner == rank) {
          void *buf = lerr ? nullptr : img->buf_alloc(nbytes);
          if (!buf) lerr = 1;
          recvs.push_back(mvx_xfer{buf, nbytes, b.owner});
          recv_oid.push_back(oid);
        }
      }
    }
    // every rank knows the whole dealing, so every rank knows whether this round moves a child at all: only then is
    // there anything to agree on
    if (world > 1 && moved_this_round > 0) {
      std::vector<double> E(1, (double)lerr);
      const int arc = allreduce_max(E);
      if (arc != 0) rc = arc;
      else if (E[0] != 0.0) rc = MVX_EFAIL; // the same code on every rank, and nobody enters the exchange
    } else if (lerr)
      rc = MVX_EFAIL;
    if (rc != 0) {
      for (auto &x : sends) if (x.buf) img->buf_free(x.buf);
      for (auto &x : recvs) if (x.buf) img->buf_free(x.buf);
      for (auto &kv : fresh) api->delete_prob(kv.second); // the children not handed to `local` yet
      break;
    }
    for (int oid : send_oid) { // packed: the image travels, the handle goes
      api->delete_prob(fresh.at(oid));
      fresh.erase(oid);
    }
    // candidate events go right behind their node's branched event (bs.cpp:300-318 emits them inside the node's
    // iteration): splice from the back so that earlier positions stay valid
    for (size_t i = cand.size(); i-- > 0;) events.insert(events.begin() + (std::ptrdiff_t)cand[i].first, cand[i].second);
    if (rc == 0 && world > 1 && (!sends.empty() || !recvs.empty()))
      rc = comm->exchange(comm->ctx, sends.data(), (int)sends.size(), recvs.data(), (int)recvs.size());
    for (size_t i = 0; i < recvs.size(); i++) {
      if (rc == 0 && !late_err) {
        void *q = api->create_prob();
        if (img->unpack(q, root, recvs[i].buf) != 0) late_err = 1; // this rank only: agreed on at the next all-reduce
        local[recv_oid[i]] = q;
      }
      img->buf_free(recvs[i].buf);
    }
    for (auto &s : sends) img->buf_free(s.buf);
    win_load.erase(round_no);
    round_no++;
    stats.rounds++;
    for (int i = 0; i < processed; i++) {
      drop(queue.front().oid);
      queue.pop_front();
    }
    // solutions kept for integral nodes: the incumbent's, and those of nodes still waiting in the queue (classified at
    // their creation, they become incumbents -- or not -- when they are popped)
    if (x_keep.size() > (size_t)(has_inc ? 1 : 0)) {
      std::unordered_set<int> waiting;
      for (const Node &nd : queue) waiting.insert(nd.oid);
      for (auto it = x_keep.begin(); it != x_keep.end();) it = (it->first != inc_oid && !waiting.count(it->first)) ? x_keep.erase(it) : std::next(it);
    }
  }

  if (rc == 0 && world > 1) { // a failure behind the last exchange has had no all-reduce to ride on yet
    std::vector<double> E(1, (double)late_err);
    rc = allreduce_max(E);
    if (rc == 0 && E[0] > 0.0) rc = MVX_EFAIL;
  } else if (rc == 0 && late_err)
    rc = MVX_EFAIL;
  for (auto &kv : local) api->delete_prob(kv.second);
  local.clear();
  if (rc != 0) return rc;

  // incumbent solution (bs.cpp:181-187) from the rank that solved it
  std::vector<double> x((size_t)n0 + 1, 0.0); // x[1..n]
  if (has_inc && rank == inc_owner) {
    const std::vector<double> &xs = x_keep.at(inc_oid);
    for (int j = 1; j <= n0; j++) x[(size_t)j] = xs[(size_t)j - 1];
  }
  if (comm && world > 1 && has_inc && (rc = comm->bcast(comm->ctx, x.data(), x.size(), inc_owner)) != 0) return rc;

  const int nn = next_id - 1;
  tree.parent.resize((size_t)nn + 1);
  tree.prune.resize((size_t)nn + 1);
  tree.bound.resize((size_t)nn + 1);
  std::vector<mvx_bnb_event> ev(events.size());
  for (size_t i = 0; i < events.size(); i++) {
    const Ev &e = events[i];
    ev[i].type = e.type;
    ev[i].oid = e.oid;
    ev[i].pid = tree.parent[(size_t)e.oid];
    ev[i].direction = branch_direction(e.oid);
    ev[i].lp_bound = e.f6;
    ev[i].sum_infeas = e.f7;
    ev[i].n_violated = e.f8;
    ev[i].pick = e.pick;
  }
  std::memset(res, 0, sizeof(*res));
  res->n_nodes = nn;
  res->parent = dup(tree.parent);
  res->prune = dup(tree.prune);
  res->node_bound = dup(tree.bound);
  res->n_events = (int)ev.size();
  res->events = dup(ev);
  res->count = count;
  res->has_incumbent = has_inc;
  res->best_lower = best_lower;
  res->incumbent_oid = inc_oid;
  res->n = n0;
  res->x = dup(x);
  res->total_pivots = total_pivots;
  res->hit_limit = hit_limit;
  stats.allreduces = n_allreduce;
  if (stats_out) *stats_out = stats;
  return 0;
}
