// Test infrastructure: the window driver of mvolps_amd/csrc/bnb.cpp (child solves on a std::async worker while the
// calling thread replays the next window, clones and deletes handles) built against the CPU oracle and run under
// ThreadSanitizer.  Prints the node count and the incumbent; TSan reports go to stderr and fail the test.
#include <cstdio>
#include <vector>

#include "../../include/mvx_bnb.h"

static unsigned long long sm_state = 5;
static double u01() {
  sm_state += 0x9E3779B97F4A7C15ull;
  unsigned long long z = sm_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int main() {
  const int m = 16, n = 32;
  mvx_prob *P = mvx_create_prob();
  mvx_set_obj_dir(P, MVX_MAX);
  mvx_add_cols(P, n);
  mvx_add_rows(P, m);
  std::vector<std::vector<double>> A((size_t)m, std::vector<double>((size_t)n + 1, 0.0));
  for (int i = 0; i < m; i++)
    for (int j = 1; j <= n; j++) A[(size_t)i][(size_t)j] = 1.0 + (double)(int)(u01() * 20.0);
  std::vector<int> ind((size_t)n + 1);
  for (int j = 0; j <= n; j++) ind[(size_t)j] = j;
  for (int j = 1; j <= n; j++) {
    mvx_set_obj_coef(P, j, 1.0 + (double)(int)(u01() * 20.0));
    mvx_set_col_bnds(P, j, MVX_DB, 0.0, 2.0);
    mvx_set_col_kind(P, j, MVX_IV);
  }
  for (int i = 0; i < m; i++) {
    double sum = 0.0;
    for (int j = 1; j <= n; j++) sum += A[(size_t)i][(size_t)j];
    mvx_set_mat_row(P, i + 1, n, ind.data(), A[(size_t)i].data());
    mvx_set_row_bnds(P, i + 1, MVX_UP, 0.0, (double)(long)(0.4 * sum));
  }
  mvx_bnb_params prm;
  mvx_bnb_default_params(&prm);
  prm.reference_quirks = 0;
  prm.window = 8;
  prm.max_nodes = 1500;
  mvx_bnb_result res;
  mvx_branchAndBound(nullptr, P, &prm, &res);
  std::printf("nodes %d incumbent %d best %.12g pivots %lld\n", res.count, res.has_incumbent, res.best_lower, res.total_pivots);
  mvx_bnb_free_result(&res);
  mvx_delete_prob(P);
  return 0;
}
