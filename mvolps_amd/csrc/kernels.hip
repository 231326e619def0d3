// kernels.hip -- gfx950 kernels of the dense-simplex engine.
//
// Every kernel takes only the device control block (mvx::Ctl), reads its pointers and
// geometry from there, and returns at once when the solve has already finished
// (ctl->done), so the host can queue pivots ahead without a round trip per pivot.
//
// Two pivot pipelines share the arithmetic:
//   generic   k_select (one 1024-thread workgroup: the driver's state machine, pricing, ratio tests,
//             pivot-row scaling; wave64 shuffle reductions + LDS across the 16 waves) then k_update
//             (the HBM-bound Gauss-Jordan rank-1 update, 16 bytes per lane).  Serves the dual simplex,
//             phase 1, the first step of every call, and -- launched once with grid.z = slots -- all
//             handles of a batched solve (B&B children / windows).
//   fused     k_fa / k_fb: primal phase 2 with no single-workgroup stage (large dense LPs).
//
// Arithmetic is mirrored operation-for-operation by the CPU checker (oracle/mvolps_oracle.c, test
// infrastructure); compiled with -ffp-contract=off so fma() appears exactly where written.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "mvx_internal.hpp"

namespace mvx {

#define TIDX ((int)threadIdx.x)

// MODE 0: larger k1, then smaller idx.   MODE 1: smaller k1, then larger k2, then smaller idx.
// a / b, correctly rounded.  The fp64 division sequence of gfx950 is almost, not exactly, IEEE: a quotient that lies
// very close to the midpoint of two doubles can come out one ulp off (-0x1.6666666666663p-1 / -0x1.ffffffffffffbp-1
// gives ...666p-1, the nearest double is ...667p-1).  Random operands never hit it (scripts/divcheck.py: 0 of 3e8), the
// near-rational entries of a tableau do: one such quotient parted a 641-node B&B run from the oracle.  The residual a - q*b of a quotient that is
// within one ulp is exact in one fma, so the better of q and its neighbour on the side the residual points to is
// the correctly rounded quotient, whatever the native division returned.  The oracle runs the same function
// (there the native quotient is already the nearest and comes back unchanged).
__device__ __forceinline__ double xdiv(double a, double b) {
  const double q = a / b;
  const double aq = fabs(q);
  if (!(aq > 1e-290 && aq < 1e290)) return q; // zero, subnormal range, inf, nan
  const double r = fma(-q, b, a);
  if (r == 0.0) return q;
  const bool up = (r > 0.0) == (b > 0.0); // the true quotient lies above q
  long long bits = __double_as_longlong(q);
  bits += ((q > 0.0) == up) ? 1 : -1;
  const double q1 = __longlong_as_double(bits);
  const double r1 = fma(-q1, b, a);
  return (fabs(r1) < fabs(r)) ? q1 : q;
}

template <int MODE>
__device__ __forceinline__ bool cand_better(const Cand &a, const Cand &b) {
  if (a.idx == 0) return false;
  if (b.idx == 0) return true;
  if (MODE == 0) {
    if (a.k1 > b.k1) return true;
    if (a.k1 < b.k1) return false;
    return a.idx < b.idx;
  } else {
    if (a.k1 < b.k1) return true;
    if (a.k1 > b.k1) return false;
    if (a.k2 > b.k2) return true;
    if (a.k2 < b.k2) return false;
    return a.idx < b.idx;
  }
}

template <int MODE>
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Cand o;
    o.k1 = __shfl_down(c.k1, off, 64);
    o.k2 = __shfl_down(c.k2, off, 64);
    o.idx = __shfl_down(c.idx, off, 64);
    o.aux = __shfl_down(c.aux, off, 64);
    if (cand_better<MODE>(o, c)) c = o;
  }
  return c;
}

// Block-wide arg-best; result broadcast to every thread.  The key is a strict total order
// (index included), so the winner does not depend on the reduction tree.
template <int MODE>
__device__ Cand block_best(Cand c, Cand *lds) {
  const int lane = TIDX & 63, wid = TIDX >> 6, nw = (int)blockDim.x >> 6;
  c = wave_best<MODE>(c);
  __syncthreads();
  if (lane == 0) lds[wid] = c;
  __syncthreads();
  if (wid == 0) {
    Cand r = (lane < nw) ? lds[lane] : Cand{0.0, 0.0, 0, 0};
    r = wave_best<MODE>(r);
    if (lane == 0) lds[16] = r;
  }
  __syncthreads();
  return lds[16];
}

// wave-level arg-best over values already in registers, broadcast to all lanes
template <int MODE>
__device__ __forceinline__ Cand wave_bcast_best(Cand b) {
  b = wave_best<MODE>(b);
  Cand r;
  r.k1 = __shfl(b.k1, 0, 64);
  r.k2 = __shfl(b.k2, 0, 64);
  r.idx = __shfl(b.idx, 0, 64);
  r.aux = __shfl(b.aux, 0, 64);
  return r;
}

// value of lane `l` (wave-uniform l) in every lane, through the scalar unit: no LDS, no barrier
__device__ __forceinline__ int rl_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ double rl_d(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
// Wave-wide maximum of a double through DPP row shifts / row broadcasts (six dependent steps of a few cycles each; the
// shuffle form is six ds_bpermute round trips per value).  Every lane must be active.
__device__ __forceinline__ double wave_max_f64(double v) {
  const int ilo = __double2loint(-INFINITY), ihi = __double2hiint(-INFINITY);
#define WMAX_STEP(CTRL, RMASK)                                                                            \
  {                                                                                                       \
    const int lo = __builtin_amdgcn_update_dpp(ilo, __double2loint(v), CTRL, RMASK, 0xf, false);          \
    const int hi = __builtin_amdgcn_update_dpp(ihi, __double2hiint(v), CTRL, RMASK, 0xf, false);          \
    v = fmax(v, __hiloint2double(hi, lo));                                                                \
  }
  WMAX_STEP(0x111, 0xf) // row_shr:1
  WMAX_STEP(0x112, 0xf) // row_shr:2
  WMAX_STEP(0x114, 0xf) // row_shr:4
  WMAX_STEP(0x118, 0xf) // row_shr:8 -> lane 15 of every row holds the row's maximum
  WMAX_STEP(0x142, 0xa) // row_bcast:15 into rows 1 and 3
  WMAX_STEP(0x143, 0xc) // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef WMAX_STEP
  return rl_d(v, 63);
}
// Wave-level arg-best, broadcast, with the lane that holds it.  The first key decides almost always: its maximum
// (MODE 0) / minimum (MODE 1) by DPP, a ballot finds who holds it; only when several lanes tie on it (or none is
// valid) the full keys go through the shuffle reduction.  Same winner as wave_bcast_best: the key is a total order.
template <int MODE>
__device__ __forceinline__ Cand wave_argbest(const Cand &x, int *who) {
  const bool valid = (x.idx != 0);
  const double key = valid ? (MODE == 0 ? x.k1 : -x.k1) : -INFINITY;
  const double best = wave_max_f64(key);
  const unsigned long long tied = __ballot(valid && key == best);
  Cand r;
  if (__builtin_popcountll(tied) == 1) {
    const int l = (int)__builtin_ctzll(tied);
    r.k1 = rl_d(x.k1, l);
    r.k2 = rl_d(x.k2, l);
    r.idx = rl_i(x.idx, l);
    r.aux = rl_i(x.aux, l);
    *who = l;
    return r;
  }
  r = wave_bcast_best<MODE>(x);
  const unsigned long long mk = __ballot(x.idx == r.idx && r.idx != 0);
  *who = mk ? (int)__builtin_ctzll(mk) : 0;
  return r;
}
// The same where the candidates' indices rise with the lane (one candidate per thread in thread order, or one per wave in
// wave order): a tie on the first key is settled by the second key's maximum among the tied lanes (MODE 1) and then by
// the lowest lane, which is the lowest index -- no shuffle reduction unless no lane is valid or a key is not a number.
// Degenerate node LPs tie on a zero ratio in most dual ratio tests.
template <int MODE>
__device__ __forceinline__ Cand wave_argbest_rising(const Cand &x, int *who) {
  const bool valid = (x.idx != 0);
  const double key = valid ? (MODE == 0 ? x.k1 : -x.k1) : -INFINITY;
  const double best = wave_max_f64(key);
  unsigned long long tied = __ballot(valid && key == best);
  if (MODE == 1 && __builtin_popcountll(tied) > 1) {
    const bool mine = valid && key == best;
    const double best2 = wave_max_f64(mine ? x.k2 : -INFINITY);
    tied = __ballot(mine && x.k2 == best2);
  }
  if (tied != 0) {
    const int l = (int)__builtin_ctzll(tied);
    Cand r;
    r.k1 = rl_d(x.k1, l);
    r.k2 = rl_d(x.k2, l);
    r.idx = rl_i(x.idx, l);
    r.aux = rl_i(x.aux, l);
    *who = l;
    return r;
  }
  return wave_argbest<MODE>(x, who);
}

// Kernel constants of one step: pointers, geometry and tolerances copied out of the control block
// at kernel entry (one burst of scalar loads) instead of being re-fetched, dependently, inside
// every helper.
struct KC {
  double *T;
  double *blb, *bub, *nlb, *nub;
  int *nflag;
  double *colq, *srow;
  int *bvar, *nvar;
  double *olb, *oub;
  double *dw; // dual devex weights by row
  double *pw; // primal devex weights by column (current set)
  int bland; // Bland's rule in force (stall >= stall_limit)
  int m, n, ld;
  double tol_bnd, tol_dj, tol_piv, sgn;
};
__device__ __forceinline__ KC load_kc(const Ctl *c) {
  KC k;
  k.T = c->T;
  k.blb = c->blb; k.bub = c->bub; k.nlb = c->nlb; k.nub = c->nub;
  k.nflag = c->nflag;
  k.colq = c->colq; k.srow = c->srow;
  k.bvar = c->bvar; k.nvar = c->nvar;
  k.olb = c->olb; k.oub = c->oub;
  k.dw = c->dwx[c->curA & 1];
  k.pw = c->pw[c->curA & 1];
  k.bland = c->stall >= c->stall_limit;
  k.m = c->m; k.n = c->n; k.ld = c->ld;
  k.tol_bnd = c->tol_bnd; k.tol_dj = c->tol_dj; k.tol_piv = c->tol_piv; k.sgn = c->sgn;
  return k;
}

// Leaving row of the dual simplex / primal feasibility check.  Score of an infeasible row: viol^2 / w[i]
// (dual devex weights; w == nullptr: all ones), lowest row on ties; under Bland's rule the lowest
// variable number (oracle: select_infeasible_row).
__device__ Cand dev_infeas_row(const KC &k, Cand *lds, const double *w) {
  Cand best{0.0, 0.0, 0, 0};
  const size_t ld = (size_t)k.ld;
  const double tol = k.tol_bnd;
  for (int i = 1 + TIDX; i <= k.m; i += (int)blockDim.x) {
    const double beta = k.T[(size_t)i * ld];
    const double lb = k.blb[i], ub = k.bub[i];
    double viol = 0.0;
    int up = 0;
    if (lb > -INFINITY && beta < lb - tol * (1.0 + fabs(lb))) viol = lb - beta;
    if (ub < INFINITY && beta > ub + tol * (1.0 + fabs(ub))) {
      viol = beta - ub;
      up = 1;
    }
    if (viol > 0.0) {
      Cand x{k.bland ? -(double)k.bvar[i] : xdiv(viol * viol, w ? w[i] : 1.0), 0.0, i, up}; // Bland: lowest variable number wins
      if (cand_better<0>(x, best)) best = x;
    }
  }
  return block_best<0>(best, lds);
}

// Entering column.  mode 0: Dantzig (largest |d_j|; phase 1 and existence checks), 1: devex with the
// current weights (d_j^2 / pw[j]), 2: devex with all weights one.  Bland's rule overrides both.
__device__ Cand dev_price(const KC &k, const double *cost, double sgn, Cand *lds, int mode = 0) {
  Cand best{0.0, 0.0, 0, 0};
  const double tol = k.tol_dj;
  for (int j = 1 + TIDX; j <= k.n; j += (int)blockDim.x) {
    const int f = k.nflag[j];
    if (f == MVX_NS) continue;
    const double dj = sgn * cost[j];
    const bool up = (f == MVX_NL || f == MVX_NF) && dj > tol;
    const bool dn = (f == MVX_NU || f == MVX_NF) && dj < -tol;
    if (!up && !dn) continue;
    const double sc = k.bland ? -(double)k.nvar[j] : (mode == 0 ? fabs(dj) : xdiv(dj * dj, mode == 1 ? k.pw[j] : 1.0));
    Cand x{sc, 0.0, j, up ? 1 : -1};
    if (cand_better<0>(x, best)) best = x;
  }
  return block_best<0>(best, lds);
}

// One row of the primal ratio test.  a = T[i][q], beta = T[i][0], gi = phase-1 sign (0 in phase 2).
// Returns false when row i does not block.
__device__ __forceinline__ bool ratio_row(double a, int sdir, double beta, double lb, double ub, int gi, double tp, int i,
                                          Cand &x) {
  const double aa = (sdir > 0) ? a : -a;
  double t = 0.0;
  int up = 0;
  if (aa > tp) {
    if (gi < 0) return false;
    if (gi > 0) {
      t = xdiv(lb - beta, aa);
      up = 0;
    } else {
      if (!(ub < INFINITY)) return false;
      t = xdiv(ub - beta, aa);
      up = 1;
    }
  } else if (aa < -tp) {
    if (gi > 0) return false;
    if (gi < 0) {
      t = xdiv(beta - ub, -aa);
      up = 1;
    } else {
      if (!(lb > -INFINITY)) return false;
      t = xdiv(beta - lb, -aa);
      up = 0;
    }
  } else
    return false;
  if (t < 0.0) t = 0.0;
  x = Cand{t, fabs(a), i, up};
  return true;
}

// Primal ratio test for entering column q moving in direction sdir; also copies the pivot
// column into colq[0..m].  g (nullable) = phase-1 infeasibility signs.
__device__ Cand dev_primal_ratio(const KC &k, int q, int sdir, const int *g, Cand *lds) {
  Cand best{0.0, 0.0, 0, 0};
  const size_t ld = (size_t)k.ld;
  const double tp = k.tol_piv;
  for (int i = TIDX; i <= k.m; i += (int)blockDim.x) {
    const double a = k.T[(size_t)i * ld + q];
    k.colq[i] = a;
    if (i == 0) continue;
    Cand x{0.0, 0.0, 0, 0};
    if (ratio_row(a, sdir, k.T[(size_t)i * ld], k.blb[i], k.bub[i], g ? g[i] : 0, tp, i, x)) {
      if (k.bland) x.k2 = -(double)k.bvar[i]; // tie-break among equal steps
      if (cand_better<1>(x, best)) best = x;
    }
  }
  return block_best<1>(best, lds);
}

__device__ Cand dev_dual_ratio(const KC &k, int p, int to_upper, Cand *lds) {
  Cand best{0.0, 0.0, 0, 0};
  const double *rowp = k.T + (size_t)p * k.ld;
  const double *row0 = k.T;
  const double tp = k.tol_piv, sgn = k.sgn;
  const bool need_inc = !to_upper;
  for (int j = 1 + TIDX; j <= k.n; j += (int)blockDim.x) {
    const int f = k.nflag[j];
    if (f == MVX_NS) continue;
    const double a = rowp[j];
    const double aa = need_inc ? a : -a;
    const double d = sgn * row0[j];
    double r;
    if (aa > tp && (f == MVX_NL || f == MVX_NF)) {
      r = (f == MVX_NF) ? fabs(d) : (d < 0.0 ? -d : 0.0);
    } else if (aa < -tp && (f == MVX_NU || f == MVX_NF)) {
      r = (f == MVX_NF) ? fabs(d) : (d > 0.0 ? d : 0.0);
    } else
      continue;
    const double mag = fabs(a);
    r = xdiv(r, mag);
    Cand x{r, k.bland ? -(double)k.nvar[j] : mag, j, 0};
    if (cand_better<1>(x, best)) best = x;
  }
  return block_best<1>(best, lds);
}

__device__ __forceinline__ double dev_nb_value(int flag, double lb, double ub) {
  return flag == MVX_NL ? lb : flag == MVX_NU ? ub : flag == MVX_NS ? lb : 0.0;
}
__device__ __forceinline__ int dev_leave_flag(double lb, double ub, int to_upper) {
  if (lb == ub) return MVX_NS;
  return to_upper ? MVX_NU : MVX_NL;
}

// Scale the pivot row into srow and publish the pivot description.  All threads call.
// wmode 0: leave the primal devex weights alone (dual and phase-1 pivots); 1: update them from the pivot
// row with pw[q] = wq (read by the caller before any lane can overwrite it); 2: same from all-one weights.
__device__ void dev_prepare_pivot(const KC &k, Ctl *c, int p, int q, int p_up, int wmode = 0, double wq = 1.0) {
  const double *rowp = k.T + (size_t)p * k.ld;
  const double piv = rowp[q];
  const double bound = p_up ? k.bub[p] : k.blb[p];
  for (int j = TIDX; j <= k.n; j += (int)blockDim.x) {
    const double v = rowp[j];
    const double sj = (j == 0) ? xdiv(v - bound, piv) : xdiv(v, piv);
    k.srow[j] = sj;
    if (wmode && j >= 1) {
      if (j == q) {
        const double cc = xdiv(wq, piv * piv);
        k.pw[j] = cc > 1.0 ? cc : 1.0;
      } else {
        const double cc = sj * sj * wq;
        double wj = (wmode == 1) ? k.pw[j] : 1.0;
        if (cc > wj) wj = cc;
        k.pw[j] = wj;
      }
    }
  }
  if (TIDX == 0) {
    c->step = ST_PIVOT;
    c->p = p;
    c->q = q;
    c->p_up = p_up;
    c->piv = piv;
    c->bound = bound;
    c->xq = dev_nb_value(k.nflag[q], k.nlb[q], k.nub[q]);
    c->leave_flag = dev_leave_flag(k.blb[p], k.bub[p], p_up);
  }
}

// Entering column chosen: ratio test, then bound flip (done here) or pivot preparation.
// Returns false when no row blocks (unbounded ray).
__device__ bool dev_primal_step(const KC &k, Ctl *c, int q, int sdir, const int *g, Cand *lds, int wmode = 0) {
  const double wq = (wmode == 1) ? k.pw[q] : 1.0; // every lane reads it ahead of the barriers inside the ratio test
  Cand r = dev_primal_ratio(k, q, sdir, g, lds);
  const double lbq = k.nlb[q], ubq = k.nub[q];
  const int fq = k.nflag[q];
  if (lbq > -INFINITY && ubq < INFINITY && fq != MVX_NF) {
    const double tf = ubq - lbq;
    if (r.idx == 0 || tf <= r.k1) {
      const double delta = (sdir > 0) ? tf : -tf;
      __syncthreads(); // colq complete
      const size_t ld = (size_t)k.ld;
      for (int i = TIDX; i <= k.m; i += (int)blockDim.x)
        k.T[(size_t)i * ld] = fma(k.colq[i], delta, k.T[(size_t)i * ld]);
      if (wmode == 2) // a flip on the first step of the phase: the weights still have to start from one
        for (int j = TIDX; j <= k.n; j += (int)blockDim.x) k.pw[j] = 1.0;
      if (TIDX == 0) {
        k.nflag[q] = (sdir > 0) ? MVX_NU : MVX_NL;
        c->step = ST_FLIP;
        c->n_flips++;
        c->n_bulk++;
        c->stall = 0;
      }
      return true;
    }
  }
  if (r.idx == 0) return false;
  dev_prepare_pivot(k, c, r.idx, q, r.aux, wmode, wq);
  if (TIDX == 0) {
    if (k.bland) c->n_bland++;
    c->stall = (r.k1 <= DEGEN_TOL) ? c->stall + 1 : 0;
  }
  return true;
}

// splitmix64 of the variable number -> [0,1) (oracle: pert_unit)
__device__ __forceinline__ double dev_pert_unit(int var) {
  unsigned long long z = (unsigned long long)var * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// Anti-stalling perturbation (oracle: perturb_basis): save every variable's bounds by variable number,
// then push the bounds of the basic variables outwards by tiny distinct amounts.  All threads call.
__device__ void dev_perturb(const KC &k, Ctl *c) {
  const int bs = (int)blockDim.x;
  for (int j = 1 + TIDX; j <= k.n; j += bs) {
    const int var = k.nvar[j];
    k.olb[var] = k.nlb[j];
    k.oub[var] = k.nub[j];
  }
  for (int i = 1 + TIDX; i <= k.m; i += bs) {
    const int var = k.bvar[i];
    const double lb = k.blb[i], ub = k.bub[i];
    k.olb[var] = lb;
    k.oub[var] = ub;
    const double u = 1.0 + dev_pert_unit(var);
    if (lb > -INFINITY) {
      double d = PERT_EPS * (1.0 + fabs(lb));
      d = d * u;
      k.blb[i] = lb - d;
    }
    if (ub < INFINITY) {
      double d = PERT_EPS * (1.0 + fabs(ub));
      d = d * u;
      k.bub[i] = ub + d;
    }
  }
  if (TIDX == 0) {
    c->perturbed = 1;
    c->pert_used = 1;
    c->n_pert++;
    c->stall = 0;
  }
}

// True bounds back on every position (oracle: restore_bounds); non-basic variables parked on a
// perturbed bound move to the true one: column 0 takes the shifts column by column in ascending
// order, each row its own fma chain -- the order shift_nonbasic is applied in by the oracle.
__device__ void dev_restore(const KC &k, Ctl *c) {
  const int bs = (int)blockDim.x;
  const size_t ld = (size_t)k.ld;
  for (int i = 1 + TIDX; i <= k.m; i += bs) {
    const int var = k.bvar[i];
    k.blb[i] = k.olb[var];
    k.bub[i] = k.oub[var];
  }
  for (int j = 1 + TIDX; j <= k.n; j += bs) {
    const int var = k.nvar[j];
    const double lb = k.olb[var], ub = k.oub[var];
    int f = k.nflag[j];
    const double xold = dev_nb_value(f, k.nlb[j], k.nub[j]);
    if (lb == ub) f = MVX_NS;
    k.nlb[j] = lb;
    k.nub[j] = ub;
    k.nflag[j] = f;
    const double xnew = dev_nb_value(f, lb, ub);
    k.srow[j] = (xnew != xold) ? xnew - xold : 0.0; // srow is free between pivots
  }
  __syncthreads();
  for (int i = TIDX; i <= k.m; i += bs) {
    const double *row = k.T + (size_t)i * ld;
    double beta = row[0];
    for (int j = 1; j <= k.n; j++) {
      const double d = k.srow[j];
      if (d != 0.0) beta = fma(row[j], d, beta);
    }
    k.T[(size_t)i * ld] = beta;
  }
  __syncthreads();
  if (TIDX == 0) c->perturbed = 0;
}

__device__ __forceinline__ void dev_finish(Ctl *c, int code, int phase, int rounds) {
  if (TIDX == 0) {
    c->done = code;
    c->phase = phase;
    c->rounds = rounds;
    c->step = ST_NONE;
  }
}

struct ChainStep {
  int p, q, lf;
  double piv, xq, s0; // s0 = srow_l[0]
};

// entry (i, j) with value v before step l -> after it; ci = column q_l entry of row i, sj = scaled pivot row entry of
// column j (both as of step l)
__device__ __forceinline__ double chain_apply(const ChainStep &st, int i, int j, double v, double ci, double sj) {
  if (i == st.p) return (j == st.q) ? xdiv(1.0, st.piv) : ((j == 0) ? st.xq - sj : -sj);
  if (j == st.q) return xdiv(ci, st.piv);
  return fma(-ci, sj, v);
}

__device__ __forceinline__ const double *uniform_ptr(const double *p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const double *)(((unsigned long long)hi << 32) | lo);
}
struct ChainView { // steps 0..k-1 as every lane needs them (one copy per block, in LDS)
  ChainStep st[KCH];
  double elb[KCH], eub[KCH], llb[KCH], lub[KCH];
  const double *cq[KCH], *sr[KCH]; // pivot column / scaled pivot row of each step
};
// One lane's share of the chain description: fetched from the control block together with everything else the kernel
// reads there (the same dependent-load level), handed to LDS once the kernel knows it has work.
struct ChainRec {
  ChainStep st;
  double elb, eub, llb, lub;
  const double *cq, *sr;
};
__device__ __forceinline__ ChainRec chain_fetch(const Ctl *c, int k) {
  ChainRec r{};
  const int l = TIDX;
  if (l < k) {
    r.st = ChainStep{c->ch_p[l], c->ch_q[l], c->ch_lf[l], c->ch_piv[l], c->ch_xq[l], c->ch_s0[l]};
    r.elb = c->ch_elb[l];
    r.eub = c->ch_eub[l];
    r.llb = c->ch_llb[l];
    r.lub = c->ch_lub[l];
    r.cq = (l == 0) ? c->colqx[c->curB] : c->colqk[l];
    r.sr = (l == 0) ? c->srow : c->srowk[l];
  }
  return r;
}
// all threads of the block call; ends with a barrier
__device__ __forceinline__ void chain_store(const ChainRec &r, int k, ChainView &v) {
  const int l = TIDX;
  if (l < k) {
    v.st[l] = r.st;
    v.elb[l] = r.elb;
    v.eub[l] = r.eub;
    v.llb[l] = r.llb;
    v.lub[l] = r.lub;
    v.cq[l] = r.cq;
    v.sr[l] = r.sr;
  }
  __syncthreads();
}
__device__ __forceinline__ void chain_load(const Ctl *c, int k, ChainView &v) { chain_store(chain_fetch(c, k), k, v); }

// ------------------------------------------------------------------ chained dual steps
// The dual simplex counterpart of the chained primal path (see "chained primal path" below for the idea): after
// select_step has prepared a dual pivot (step 0), the following dual pivots of the same solve are chosen by the same
// workgroup from slices of the tableau as it stands -- column 0, the leaving row, the objective row, the entering
// column -- carried through the earlier steps of the chain with chain_apply, and k_update then applies the whole chain
// in one pass.  This is what every warm-started B&B child runs (bs.cpp:279,287): with 64-128 node LPs per launch the
// pass over their tableaux is the cost of a step, and a chain divides it.  A chain ends (the next k_select carries on
// from the updated tableau) on: pivot limit, stall limit, no infeasible row left, no entering column.  Bland's rule, a
// dual phase that has just begun (weights restart) and the other phases never chain.
__device__ int dual_chain(const KC &k, Ctl *c, Cand *lds, int kmax) {
  __shared__ ChainView v;
  __shared__ double s_val[2];
  const int m = k.m, n = k.n, bs = (int)blockDim.x;
  const size_t ld = (size_t)k.ld;
  const double *const T = k.T;
  const int budget = c->budget, stall_limit = c->stall_limit;
  if (TIDX == 0) {
    const int p0 = c->p, q0 = c->q;
    v.st[0] = ChainStep{p0, q0, c->leave_flag, c->piv, c->xq, k.srow[0]};
    v.elb[0] = k.nlb[q0];
    v.eub[0] = k.nub[q0];
    v.llb[0] = k.blb[p0];
    v.lub[0] = k.bub[p0];
    v.cq[0] = k.colq;
    v.sr[0] = k.srow;
    for (int l = 1; l < kmax; l++) {
      v.cq[l] = c->colqk[l];
      v.sr[l] = c->srowk[l];
    }
  }
  __syncthreads();
  int stall = c->stall; // select_step has just set it for step 0
  int nch = 1;
  for (int kk = 1; kk < kmax; kk++) {
    if ((budget >= 0 && budget < kk + 1) || stall >= stall_limit) break;
    // ---- leaving row: basic values and bounds as of step kk (dev_infeas_row on carried values)
    Cand rbest{0.0, 0.0, 0, 0};
    const double tolb = k.tol_bnd;
    for (int i = 1 + TIDX; i <= m; i += bs) {
      double beta = T[(size_t)i * ld];
      double lb = k.blb[i], ub = k.bub[i];
      for (int l = 0; l < kk; l++) {
        beta = chain_apply(v.st[l], i, 0, beta, v.cq[l][i], v.st[l].s0);
        if (v.st[l].p == i) {
          lb = v.elb[l];
          ub = v.eub[l];
        }
      }
      double viol = 0.0;
      int up = 0;
      if (lb > -INFINITY && beta < lb - tolb * (1.0 + fabs(lb))) viol = lb - beta;
      if (ub < INFINITY && beta > ub + tolb * (1.0 + fabs(ub))) {
        viol = beta - ub;
        up = 1;
      }
      if (viol > 0.0) {
        Cand x{xdiv(viol * viol, k.dw[i]), 0.0, i, up};
        if (cand_better<0>(x, rbest)) rbest = x;
      }
    }
    rbest = block_best<0>(rbest, lds);
    if (rbest.idx == 0) break; // primal feasible: the next k_select changes phase
    const int p = rbest.idx, p_up = rbest.aux;
    const double wp = k.dw[p]; // every lane reads it before its owner rewrites it below
    // ---- row p and the objective row as of step kk: dual ratio test (dev_dual_ratio on carried values)
    double *const sk = c->srowk[kk];
    Cand best{0.0, 0.0, 0, 0};
    const double tp = k.tol_piv, sgn = k.sgn;
    const bool need_inc = !p_up;
    for (int j = TIDX; j <= n; j += bs) {
      double a = T[(size_t)p * ld + j];
      double d0 = T[j];
      int f = (j >= 1) ? k.nflag[j] : MVX_NS;
      for (int l = 0; l < kk; l++) {
        const double sj = v.sr[l][j];
        a = chain_apply(v.st[l], p, j, a, v.cq[l][p], sj);
        d0 = chain_apply(v.st[l], 0, j, d0, v.cq[l][0], sj);
        if (v.st[l].q == j) f = v.st[l].lf;
      }
      sk[j] = a; // row p as of step kk; scaled in place once the pivot element is known
      if (j == 0 || f == MVX_NS) continue;
      const double aa = need_inc ? a : -a;
      const double d = sgn * d0;
      double r;
      if (aa > tp && (f == MVX_NL || f == MVX_NF)) {
        r = (f == MVX_NF) ? fabs(d) : (d < 0.0 ? -d : 0.0);
      } else if (aa < -tp && (f == MVX_NU || f == MVX_NF)) {
        r = (f == MVX_NF) ? fabs(d) : (d > 0.0 ? d : 0.0);
      } else
        continue;
      const double mag = fabs(a);
      r = xdiv(r, mag);
      Cand x{r, mag, j, 0};
      if (cand_better<1>(x, best)) best = x;
    }
    best = block_best<1>(best, lds);
    if (best.idx == 0) break; // no entering column: the generic step reports it
    const int q = best.idx;
    // the entering variable and the leaving row as the earlier steps left them
    double lbq = k.nlb[q], ubq = k.nub[q];
    int fq = k.nflag[q];
    double plb = k.blb[p], pub = k.bub[p];
    for (int l = 0; l < kk; l++) {
      if (v.st[l].q == q) {
        lbq = v.llb[l];
        ubq = v.lub[l];
        fq = v.st[l].lf;
      }
      if (v.st[l].p == p) {
        plb = v.elb[l];
        pub = v.eub[l];
      }
    }
    const double apq = sk[q]; // written above by the lane that owns column q (barriers inside block_best)
    __syncthreads(); // ... and rescaled in place further down: every lane has its copy before any lane gets there
    const double bound = p_up ? pub : plb;
    const int lf = dev_leave_flag(plb, pub, p_up);
    // ---- column q as of step kk + dual devex weights (select_step's pass over the rows)
    double *const ck = c->colqk[kk];
    for (int i = TIDX; i <= m; i += bs) {
      double a = T[(size_t)i * ld + q];
      for (int l = 0; l < kk; l++) a = chain_apply(v.st[l], i, q, a, v.cq[l][i], v.sr[l][q]);
      ck[i] = a;
      if (i == 0) continue;
      if (i == p) {
        const double cc = xdiv(wp, apq * apq);
        k.dw[i] = cc > 1.0 ? cc : 1.0;
      } else {
        const double r = xdiv(a, apq);
        const double cc = r * r * wp;
        double wi = k.dw[i];
        if (cc > wi) wi = cc;
        k.dw[i] = wi;
      }
    }
    // ---- scaled pivot row (dev_prepare_pivot)
    for (int j = TIDX; j <= n; j += bs) {
      const double a = sk[j];
      const double sj = (j == 0) ? xdiv(a - bound, apq) : xdiv(a, apq);
      sk[j] = sj;
      if (j == 0) s_val[0] = sj;
    }
    stall = (best.k1 <= DEGEN_TOL) ? stall + 1 : 0;
    __syncthreads();
    if (TIDX == 0) {
      v.st[kk] = ChainStep{p, q, lf, apq, dev_nb_value(fq, lbq, ubq), s_val[0]};
      v.elb[kk] = lbq;
      v.eub[kk] = ubq;
      v.llb[kk] = plb;
      v.lub[kk] = pub;
      c->ch_p[kk] = p;
      c->ch_q[kk] = q;
      c->ch_lf[kk] = lf;
      c->ch_piv[kk] = apq;
      c->ch_xq[kk] = v.st[kk].xq;
      c->ch_s0[kk] = s_val[0];
    }
    __syncthreads();
    nch = kk + 1;
  }
  if (TIDX == 0) {
    c->ch_p[0] = v.st[0].p;
    c->ch_q[0] = v.st[0].q;
    c->ch_lf[0] = v.st[0].lf;
    c->ch_piv[0] = v.st[0].piv;
    c->ch_xq[0] = v.st[0].xq;
    c->ch_s0[0] = v.st[0].s0;
    c->stall = stall;
  }
  return nch;
}

struct DcSlot {
  Cand c;
  double pay[4];
};
// Workgroup barrier for exchanges that go through LDS only: waits for this wave's LDS traffic, not for its global stores
// (__syncthreads() carries a release fence that also waits for every outstanding global store to be acknowledged).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Arg-best over the 1024 threads of a block with four doubles of the winner's, in every thread.  slots: 17 entries.
// The sixteen wave winners are settled by wave 0 alone: k_dsel is bound by instruction issue (sixteen waves on four
// SIMDs, ~16 k cycles a step), and the same second stage run by all sixteen waves costs four times what one wave
// running it alone does, against one more LDS read per thread here.
template <int MODE>
__device__ __forceinline__ Cand block_argbest16(Cand mine, const double (&pin)[4], double (&pout)[4], DcSlot *slots) {
  const int lane = TIDX & 63, wave = TIDX >> 6;
  int ol;
  const Cand wb = wave_argbest_rising<MODE>(mine, &ol); // thread t holds index t + 1
  if (lane == 0) slots[wave].c = wb;
  if (lane == ol) {
#pragma unroll
    for (int k = 0; k < 4; k++) slots[wave].pay[k] = pin[k];
  }
  lds_barrier();
  if (wave == 0) {
    Cand rc{0.0, 0.0, 0, 0};
    if (lane < 16) rc = slots[lane].c;
    int wl;
    const Cand win = wave_argbest_rising<MODE>(rc, &wl); // lane w holds wave w's best
    if (lane == 0) slots[16].c = win;
    if (lane < 4) slots[16].pay[lane] = slots[wl].pay[lane];
  }
  lds_barrier(); // slot 16 is rewritten only after the next reduction's first barrier: every wave has read it by then
#pragma unroll
  for (int k = 0; k < 4; k++) pout[k] = slots[16].pay[k];
  return slots[16].c;
}

// ------------------------------------------------------------------ k_dsel: a dual phase that carries on, on chip
// Every warm-started B&B child runs the dual simplex (bs.cpp:279,287), and k_select's generic stage plus dual_chain
// re-derive everything from memory at every step -- column 0, the bounds, the objective row, the statuses carried through
// the chain with a global load per earlier step: five dependent stages of memory round trips, 11.5 us per step; the
// calibrated config-5 tree spends 77 % of its GPU time there (profiles/r03_config5_kernel_stats.txt).  k_dsel is launched
// in front of k_select and takes the step over when the solve is simply carrying on in the dual phase on a node LP of up
// to 1024 rows and 1024 columns (one row and one column per thread of the 1024): what a step changes lives in registers
// -- basic values, bounds and dual devex weights of the rows; reduced costs, statuses and bounds of the columns --, every
// thread's own entries of the chain's pivot columns and scaled pivot rows in LDS, so a step is two memory round trips
// (row p, column q) and a handful of LDS exchanges, and the whole chain of up to 8 dual pivots, the first included,
// starts from one load of the state.  Same arithmetic in the same order per entry as select_step + dual_chain: the
// result is theirs bit for bit.  Whenever the first step is not a plain dual pivot (no infeasible row left: the phase
// changes; no entering column: the LP is infeasible; Bland's rule in force; bound edits waiting; the pivot limit) it
// touches nothing and k_select decides; otherwise it leaves `dsel` set and k_select returns at once.
// what the batched rounds do (scripts/roundstats.py): [0] k_dsel not applicable, [1..DCH_MAX] chains of that length,
// [DCH_MAX+1] applicable but no plain dual pivot; then k_select: [+2] passthrough, [+3] idle slot, [+4] primal phase 2 step,
// [+5] dual step, [+6] other phase, [+7] solve ended in this call
__device__ int g_stats_on; // mvx_debug_stats(1): the counters below are kept (a stamp costs a clock read and a wait)
__device__ unsigned long long g_round_hist[DCH_MAX + 10];
// shader-clock cycles of k_dsel by part (thread 0, summed): [0] entry + state loads, [1] leaving row, [2] row p + ratio test,
// [3] column q, [4] bookkeeping of the step, [5] exit; [6] steps, [7] launches that ran a chain
__device__ unsigned long long g_dsel_cycles[8];
constexpr size_t DSEL_LDS = (size_t)2 * DCH_MAX * 1024 * 8; // the chain's history: one slot per thread, step and side
__global__ __launch_bounds__(1024) void k_dsel(Ctl *c) {
  extern __shared__ double dc_hist[]; // [2][DCH_MAX][1024]
  __shared__ DcSlot s_slots[17];
  __shared__ double s_hp[DCH_MAX], s_hq[DCH_MAX], s_misc[4];
  __shared__ ChainStep sts[DCH_MAX]; // the chain so far (every thread holds the same values; thread 0 writes them down)
  c += blockIdx.z;
  if (c->done != D_RUN || c->phase != PH_DUAL || c->n_edits != 0 || c->dchain_max <= 1 || c->budget == 0 || c->pc_itlim ||
      c->stall >= c->stall_limit || c->m > 1024 || c->n > 1024 || c->T == nullptr) {
    if (g_stats_on && TIDX == 0 && c->done == D_RUN) atomicAdd(&g_round_hist[0], 1ull);
    return;
  }
  const bool stats = g_stats_on != 0;
  unsigned long long tk = stats ? __builtin_readcyclecounter() : 0ull;
#define DS_STAMP(K)                                                \
  if (stats && TIDX == 0) {                                        \
    const unsigned long long now_ = __builtin_readcyclecounter();  \
    atomicAdd(&g_dsel_cycles[K], now_ - tk);                       \
    tk = now_;                                                     \
  }
  const KC k = load_kc(c);
  const int m = k.m, n = k.n, t = TIDX, kmax = c->dchain_max;
  const size_t ld = (size_t)k.ld;
  const double *const T = k.T;
  const int budget = c->budget, stall_limit = c->stall_limit;
  const int i = 1 + t, j = 1 + t;
  const bool ract = (i <= m), cact = (j <= n);
  const int ic = ract ? i : m, jc = cact ? j : n;
#define DC_HC(L) dc_hist[(size_t)(L) * 1024 + t]
#define DC_HS(L) dc_hist[(size_t)(DCH_MAX + (L)) * 1024 + t]
  // ---- the state as it stands
  double be = T[(size_t)ic * ld], d = T[jc];
  double lb = k.blb[ic], ub = k.bub[ic], dw = k.dw[ic];
  int f = cact ? k.nflag[jc] : MVX_NS;
  double nlbj = k.nlb[jc], nubj = k.nub[jc]; // bounds of the columns as the chain leaves them (x_q of a later step)
  int stall = c->stall;
  int nch = 0, p0 = 0, q0 = 0, lf0 = 0, p_up0 = 0;
  double piv0 = 1.0, xq0 = 0.0, bound0 = 0.0;
  for (int kk = 0; kk < kmax; kk++) {
    if (kk >= 1 && ((budget >= 0 && budget < kk + 1) || stall >= stall_limit)) break;
    // ---- leaving row (dev_infeas_row on the carried values)
    if (kk == 0) DS_STAMP(0);
    Cand rb{0.0, 0.0, 0, 0};
    const double tolb = k.tol_bnd;
    if (ract) {
      double viol = 0.0;
      int up = 0;
      if (lb > -INFINITY && be < lb - tolb * (1.0 + fabs(lb))) viol = lb - be;
      if (ub < INFINITY && be > ub + tolb * (1.0 + fabs(ub))) {
        viol = be - ub;
        up = 1;
      }
      if (viol > 0.0) rb = Cand{xdiv(viol * viol, dw), 0.0, i, up};
    }
    double rp[4] = {be, lb, ub, dw}, ro[4];
    const Cand rw = block_argbest16<0>(rb, rp, ro, s_slots);
    if (rw.idx == 0) break; // primal feasible: k_select changes phase
    const int p = rw.idx, p_up = rw.aux;
    const double bp = ro[0], plb = ro[1], pub = ro[2], wp = ro[3];
    DS_STAMP(1);
    // ---- row p as of step kk: its entries of the earlier pivot columns come from the thread that owns the row
    double a = T[(size_t)p * ld + jc];
    if (i == p)
      for (int l = 0; l < kk; l++) s_hp[l] = DC_HC(l);
    lds_barrier();
    for (int l = 0; l < kk; l++) a = chain_apply(sts[l], p, jc, a, s_hp[l], DC_HS(l));
    // dual ratio test (dev_dual_ratio on the carried values)
    Cand best{0.0, 0.0, 0, 0};
    const double tp = k.tol_piv, sgn = k.sgn;
    if (cact && f != MVX_NS) {
      const double aa = p_up ? -a : a;
      const double dd = sgn * d;
      double r = 0.0;
      bool ok = false;
      if (aa > tp && (f == MVX_NL || f == MVX_NF)) {
        r = (f == MVX_NF) ? fabs(dd) : (dd < 0.0 ? -dd : 0.0);
        ok = true;
      } else if (aa < -tp && (f == MVX_NU || f == MVX_NF)) {
        r = (f == MVX_NF) ? fabs(dd) : (dd > 0.0 ? dd : 0.0);
        ok = true;
      }
      if (ok) {
        const double mag = fabs(a);
        best = Cand{xdiv(r, mag), mag, j, 0};
      }
    }
    double qp[4] = {a, d, nlbj, nubj}, qo[4];
    const Cand qw = block_argbest16<1>(best, qp, qo, s_slots);
    if (qw.idx == 0) break; // no entering column: k_select reports it
    const int q = qw.idx;
    const double apq = qo[0], dq = qo[1], lbq = qo[2], ubq = qo[3];
    DS_STAMP(2);
    // ---- column q as of step kk: its entries of the earlier scaled rows (and its status) come from its owner
    double cq = T[(size_t)ic * ld + q];
    if (j == q) {
      for (int l = 0; l < kk; l++) s_hq[l] = DC_HS(l);
      s_misc[0] = (double)f;
    }
    lds_barrier();
    const int fq = (int)s_misc[0];
    for (int l = 0; l < kk; l++) cq = chain_apply(sts[l], ic, q, cq, DC_HC(l), s_hq[l]);
    if (stats && TIDX == 0) asm volatile("" ::"v"(cq)); // the stamp that follows waits for the column
    DS_STAMP(3);
    const double bound = p_up ? pub : plb;
    const int lf = dev_leave_flag(plb, pub, p_up);
    const double s0 = xdiv(bp - bound, apq);
    const double xq = dev_nb_value(fq, lbq, ubq);
    const ChainStep stk{p, q, lf, apq, xq, s0};
    // step 0 goes where the generic stage puts its step (colq / srow), the following ones where dual_chain puts them
    double *const ck = (kk == 0) ? k.colq : c->colqk[kk], *const sk = (kk == 0) ? k.srow : c->srowk[kk];
    // the rows: pivot column out, dual devex weights (select_step's pass), basic values and bounds after the step
    if (ract) {
      ck[i] = cq;
      if (i == p) {
        const double cc = xdiv(wp, apq * apq);
        dw = cc > 1.0 ? cc : 1.0;
      } else {
        const double r = xdiv(cq, apq);
        const double cc = r * r * wp;
        if (cc > dw) dw = cc;
      }
      be = chain_apply(stk, i, 0, be, cq, s0);
      if (i == p) {
        lb = lbq;
        ub = ubq;
      }
    }
    DC_HC(kk) = cq;
    // the columns: scaled pivot row out, reduced costs and statuses after the step
    const double sj = xdiv(a, apq);
    if (cact) sk[j] = sj;
    d = chain_apply(stk, 0, jc, d, dq, sj);
    if (cact && j == q) {
      f = lf;
      nlbj = plb;
      nubj = pub;
    }
    DC_HS(kk) = sj;
    stall = (qw.k1 <= DEGEN_TOL) ? stall + 1 : 0;
    if (kk == 0) {
      p0 = p; q0 = q; lf0 = lf; piv0 = apq; xq0 = xq; p_up0 = p_up; bound0 = bound;
    }
    if (t == 0) {
      sts[kk] = stk;
      ck[0] = dq;
      sk[0] = s0;
      c->ch_p[kk] = p;
      c->ch_q[kk] = q;
      c->ch_lf[kk] = lf;
      c->ch_piv[kk] = apq;
      c->ch_xq[kk] = xq;
      c->ch_s0[kk] = s0;
    }
    nch = kk + 1;
    DS_STAMP(4);
    if (stats && TIDX == 0) atomicAdd(&g_dsel_cycles[6], 1ull);
  }
#undef DC_HC
#undef DC_HS
  if (stats && TIDX == 0) atomicAdd(&g_round_hist[nch ? nch : DCH_MAX + 1], 1ull);
  if (nch == 0) return; // not a plain dual pivot: nothing has been touched, k_select decides
  if (ract) k.dw[i] = dw; // the weights as the chain leaves them, where the next step reads them
  if (t == 0) {
    c->stall = stall;
    c->nch = nch;
    c->dsel = 1; // k_select has nothing to do for this step
    // what dev_prepare_pivot publishes
    c->step = ST_PIVOT;
    c->p = p0;
    c->q = q0;
    c->p_up = p_up0;
    c->piv = piv0;
    c->bound = bound0;
    c->xq = xq0;
    c->leave_flag = lf0;
  }
  DS_STAMP(5);
  if (stats && TIDX == 0) atomicAdd(&g_dsel_cycles[7], 1ull);
#undef DS_STAMP
}

// ---------------------------------------------------------------------------- k_select
// Device-side restatement of orc_simplex's round loop + one pricing / ratio-test step.
__device__ void select_step(Ctl *c, Cand *lds) {
  const KC k = load_kc(c); // every pointer / constant the step needs, fetched in one burst
  if (c->done != D_RUN) return;
  if (c->pc_itlim) { // k_chain saw the pivot limit fall on the end of its chain, with an entering column still on offer
    __syncthreads();
    if (TIDX == 0) {
      c->pc_itlim = 0;
      c->fstate = F_STOP;
    }
    dev_finish(c, D_ITLIM, PH_PRIMAL2, c->rounds);
    return;
  }
  if (TIDX == 0) c->nch = 1; // whatever step this call prepares is a single one unless a dual chain says otherwise below
  const int ne = c->n_edits;
  if (ne) { // first step of a solve: the bound edits made since the last one (one lane each)
    if (TIDX < ne) {
      const int r = c->edit_row[TIDX];
      k.blb[r] = c->edit_lb[TIDX];
      k.bub[r] = c->edit_ub[TIDX];
    }
    __syncthreads();
    if (TIDX == 0) c->n_edits = 0;
  }
  int phase = c->phase, rounds = c->rounds;
  int nch = 1; // pivots the coming k_update applies (dual chains)
  int p = 0, p_up = 0, q = 0, sdir = 0, kind = 0; // kind 1 primal, 2 dual
  bool fresh_dual = false; // the dual phase starts with this step: devex weights restart from one
  bool fresh_primal = false; // primal phase 2 starts with this step: likewise
  for (;;) {
    if (phase == PH_START) {
      Cand r = dev_infeas_row(k, lds, nullptr);
      if (r.idx == 0) {
        phase = PH_PRIMAL2;
        fresh_primal = true;
      } else {
        Cand pr = dev_price(k, k.T, k.sgn, lds);
        if (pr.idx != 0) {
          dev_finish(c, D_NEED_PHASE1, PH_PHASE1, rounds);
          return;
        }
        phase = PH_DUAL;
        p = r.idx;
        p_up = r.aux;
        kind = 2;
        fresh_dual = true;
        break;
      }
    }
    if (phase == PH_PRIMAL2) {
      if (k.bland && !c->pert_used) {
        // stalled for the first time: perturb instead of pivoting; the next launch prices again
        dev_perturb(k, c);
        if (fresh_primal)
          for (int j = TIDX; j <= k.n; j += (int)blockDim.x) k.pw[j] = 1.0;
        if (TIDX == 0) {
          c->step = ST_NONE;
          c->phase = phase;
          c->rounds = rounds;
        }
        return;
      }
      Cand pr = dev_price(k, k.T, k.sgn, lds, fresh_primal ? 2 : 1);
      if (pr.idx != 0) {
        q = pr.idx;
        sdir = pr.aux;
        kind = 1;
        break;
      }
      if (c->perturbed) {
        // optimal for the perturbed bounds: true bounds back, then look again (the dual simplex
        // removes what infeasibility the shift leaves)
        dev_restore(k, c);
        if (TIDX == 0) {
          c->step = ST_NONE;
          c->phase = phase;
          c->rounds = rounds;
        }
        return;
      }
      Cand r = dev_infeas_row(k, lds, nullptr);
      if (r.idx == 0) {
        dev_finish(c, D_OPT, phase, rounds);
        return;
      }
      if (++rounds >= 64) {
        dev_finish(c, D_FAIL, phase, rounds);
        return;
      }
      phase = PH_DUAL;
      p = r.idx;
      p_up = r.aux;
      kind = 2;
      fresh_dual = true;
      break;
    }
    if (phase == PH_DUAL) {
      Cand r = dev_infeas_row(k, lds, k.dw);
      if (r.idx != 0) {
        p = r.idx;
        p_up = r.aux;
        kind = 2;
        break;
      }
      if (++rounds >= 64) {
        dev_finish(c, D_FAIL, phase, rounds);
        return;
      }
      phase = PH_PRIMAL2;
      fresh_primal = true;
    }
  }
  if (c->budget == 0) {
    if (c->perturbed) dev_restore(k, c);
    dev_finish(c, D_ITLIM, phase, rounds);
    return;
  }
  if (kind == 1) {
    if (!dev_primal_step(k, c, q, sdir, nullptr, lds, fresh_primal ? 2 : 1)) {
      if (c->perturbed) dev_restore(k, c);
      dev_finish(c, D_UNBND, phase, rounds);
      return;
    }
  } else {
    const double wp = fresh_dual ? 1.0 : k.dw[p]; // read by every lane before its owner rewrites it below
    Cand dr = dev_dual_ratio(k, p, p_up, lds);
    if (dr.idx == 0) {
      dev_finish(c, D_NOFEAS, phase, rounds);
      return;
    }
    q = dr.idx;
    const size_t ld = (size_t)k.ld;
    const double apq = k.T[(size_t)p * ld + q];
    // pivot column copy + devex weight update (oracle: dual_simplex), one pass over the rows
    for (int i = TIDX; i <= k.m; i += (int)blockDim.x) {
      const double a = k.T[(size_t)i * ld + q];
      k.colq[i] = a;
      if (i == 0) continue;
      if (i == p) {
        const double cc = xdiv(wp, apq * apq);
        k.dw[i] = cc > 1.0 ? cc : 1.0;
      } else {
        const double r = xdiv(a, apq);
        const double cc = r * r * wp;
        double wi = fresh_dual ? 1.0 : k.dw[i];
        if (cc > wi) wi = cc;
        k.dw[i] = wi;
      }
    }
    dev_prepare_pivot(k, c, p, q, p_up);
    if (TIDX == 0) {
      if (k.bland) c->n_bland++;
      c->stall = (dr.k1 <= DEGEN_TOL) ? c->stall + 1 : 0;
    }
    if (c->dchain_max > 1 && !fresh_dual && !k.bland && phase == PH_DUAL) {
      __syncthreads(); // step 0 is complete: srow, colq, weights, the control block's pivot description
      nch = dual_chain(k, c, lds, c->dchain_max);
    }
  }
  if (TIDX == 0) {
    c->phase = phase;
    c->rounds = rounds;
    c->nch = nch;
  }
}

// What the host needs after a solve, packed into one staging area (all threads of the block call):
//   [Ctl][beta (m_cap+1) f64][d (ld) f64][bvar (m_cap+1) i32][nvar (ld) i32][nflag (ld) i32]
__device__ void pack_mirrors(const Ctl *c, unsigned char *stage, int t0, int step) {
  const int m = c->m, n = c->n, ld = c->ld, mc = c->m_cap;
  double *beta = reinterpret_cast<double *>(stage + sizeof(Ctl));
  double *dj = beta + (mc + 1);
  int *bv = reinterpret_cast<int *>(dj + ld);
  int *nv = bv + (mc + 1);
  int *nf = nv + ld;
  for (int t = t0; t <= m || t <= n; t += step) {
    if (t <= m) {
      beta[t] = c->T[(size_t)t * ld];
      bv[t] = c->bvar[t];
    }
    if (t <= n) {
      dj[t] = c->T[t];
      nv[t] = c->nvar[t];
      nf[t] = c->nflag[t];
    }
  }
}

__global__ __launch_bounds__(1024) void k_select(Ctl *c, BatchQueue q) {
  __shared__ Cand lds[17];
  __shared__ int s_job;
  if (q.jobs && blockIdx.z == 0 && TIDX == 0) q.counters[2]++; // rounds of this batch so far (one writer per launch)
  c += blockIdx.z; // slot of a batched launch (mvx_simplex_batch); 0 for single solves
  if (c->dsel) { // k_dsel has prepared this step (a dual phase carrying on): nothing to select
    __syncthreads();
    if (TIDX == 0) {
      c->dsel = 0;
      if (g_stats_on) atomicAdd(&g_round_hist[DCH_MAX + 2], 1ull);
    }
    return;
  }
  if (q.jobs && c->done != D_RUN) {
    // this slot's solve has ended (or it never had one): hand the finished job over and pull the next
    const int old = c->job;
    if (old >= 0) {
      unsigned char *st = q.stage + (size_t)old * q.stage_stride;
      pack_mirrors(c, st, TIDX, 1024);
      const unsigned *src = reinterpret_cast<const unsigned *>(c);
      unsigned *dst = reinterpret_cast<unsigned *>(st);
      for (int w = TIDX; w < (int)(sizeof(Ctl) / 4); w += 1024) dst[w] = src[w];
    }
    __syncthreads();
    if (TIDX == 0) {
      if (old >= 0) {
        __threadfence(); // the staging area is complete before the job counts as finished
        // the round in which the last job ended (give or take one): the host sizes the next batch's first burst by it
        if (atomicAdd(&q.counters[1], 1) + 1 == q.count) q.counters[3] = q.counters[2];
      }
      s_job = (q.counters[0] < q.count) ? atomicAdd(&q.counters[0], 1) : q.count;
    }
    __syncthreads();
    const int j = s_job;
    if (j >= q.count) {
      if (TIDX == 0) {
        c->job = -1;
        if (g_stats_on) atomicAdd(&g_round_hist[DCH_MAX + 3], 1ull);
      }
      return;
    }
    const unsigned *src = reinterpret_cast<const unsigned *>(&q.jobs[j]);
    unsigned *dst = reinterpret_cast<unsigned *>(c);
    for (int w = TIDX; w < (int)(sizeof(Ctl) / 4); w += 1024) dst[w] = src[w];
    __syncthreads();
    if (TIDX == 0) {
      const SlotScratch sp = q.scratch[blockIdx.z];
      c->colq = sp.colq; c->srow = sp.srow; c->olb = sp.olb; c->oub = sp.oub; c->dw = sp.dw; c->dwx[0] = c->dwx[1] = sp.dw;
      c->pw[0] = c->pw[1] = sp.pw;
      for (int l = 1; l < DCH_MAX; l++) { // dual chains: this slot's per-step buffers
        unsigned char *b = reinterpret_cast<unsigned char *>(sp.chain) + (size_t)(l - 1) * sp.chain_stride;
        c->colqk[l] = reinterpret_cast<double *>(b);
        c->srowk[l] = reinterpret_cast<double *>(b + sp.chain_col);
      }
      c->job = j;
    }
    __syncthreads();
  }
  const int ph0 = c->phase;
  select_step(c, lds);
  if (TIDX == 0 && q.jobs && g_stats_on) // thread 0 wrote `done` itself: no barrier
    atomicAdd(&g_round_hist[DCH_MAX + (c->done != D_RUN ? 7 : ph0 == PH_PRIMAL2 ? 4 : (ph0 == PH_DUAL || ph0 == PH_START) ? 5 : 6)], 1ull);
}

extern "C" void mvx_debug_stats(int on) { // switch the round / cycle counters on or off (off at start)
  (void)hipDeviceSynchronize();
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats_on), &on, sizeof(int));
}
extern "C" void mvx_debug_round_hist(unsigned long long *out, int reset) { // DCH_MAX + 10 counters (g_round_hist)
  static const unsigned long long zeros[DCH_MAX + 10] = {};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_round_hist), sizeof(zeros));
  if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_round_hist), zeros, sizeof(zeros));
}
extern "C" void mvx_debug_dsel_cycles(unsigned long long *out, int reset) { // 8 counters (g_dsel_cycles)
  static const unsigned long long zeros[8] = {};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dsel_cycles), sizeof(zeros));
  if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dsel_cycles), zeros, sizeof(zeros));
}

// ------------------------------------------------------------------------ phase-1 kernels
// Phase 1 keeps its cost row  sum_i g[i]*T[i][:]  (g = +1 below the lower bound, -1 above the upper one) in
// tableau row m+1 and carries it through the pivots (k_update treats it like any other row) instead of
// recomputing it: k_p1_head lists, in ascending order, the rows whose sign changed since the last
// iteration, k_p1_fix adds them with the signed difference (oracle: primal_phase1).
__global__ __launch_bounds__(1024) void k_p1_head(Ctl *c) {
  __shared__ int s_scan[1024];
  __shared__ int s_inf[16];
  if (c->done != D_RUN) return;
  const int m = c->m, n = c->n, bs = (int)blockDim.x;
  const size_t ld = (size_t)c->ld;
  const double tol = c->tol_bnd;
  double *cost = c->T + (size_t)(m + 1) * ld;
  const int init = c->p1_init;
  if (init) {
    for (int j = TIDX; j <= n; j += bs) cost[j] = 0.0;
    for (int i = TIDX; i <= m + 1; i += bs) c->gflag[i] = 0;
  } else if (TIDX == 0 && c->p1_fix_q) {
    // the last pivot left g[p]*e_q in the row: the leaving variable now sits on the bound it violated
    cost[c->p1_fix_q] = cost[c->p1_fix_q] - (double)c->p1_fix_g;
  }
  __syncthreads();
  // this lane's rows: a contiguous chunk, so that the list comes out in ascending row order
  const int R = (m + bs - 1) / bs;
  const int i0 = 1 + TIDX * R;
  int nchg = 0, ninf = 0;
  for (int r = 0; r < R; r++) {
    const int i = i0 + r;
    if (i > m) break;
    const double beta = c->T[(size_t)i * ld];
    const double lb = c->blb[i], ub = c->bub[i];
    int g = 0;
    if (lb > -INFINITY && beta < lb - tol * (1.0 + fabs(lb))) g = 1;
    if (ub < INFINITY && beta > ub + tol * (1.0 + fabs(ub))) g = -1;
    nchg += (g != c->gflag[i]);
    ninf += (g != 0);
  }
  // exclusive scan of the change counts over the lanes (Hillis-Steele in LDS)
  s_scan[TIDX] = nchg;
  __syncthreads();
  for (int off = 1; off < bs; off <<= 1) {
    const int v = (TIDX >= off) ? s_scan[TIDX - off] : 0;
    __syncthreads();
    s_scan[TIDX] += v;
    __syncthreads();
  }
  int pos = s_scan[TIDX] - nchg;
  const int total = s_scan[bs - 1];
  for (int r = 0; r < R; r++) {
    const int i = i0 + r;
    if (i > m) break;
    const double beta = c->T[(size_t)i * ld];
    const double lb = c->blb[i], ub = c->bub[i];
    int g = 0;
    if (lb > -INFINITY && beta < lb - tol * (1.0 + fabs(lb))) g = 1;
    if (ub < INFINITY && beta > ub + tol * (1.0 + fabs(ub))) g = -1;
    const int go = c->gflag[i];
    if (g != go) {
      c->p1_list[pos] = i;
      c->wts[pos] = (double)(g - go);
      pos++;
      c->gflag[i] = g;
    }
  }
  for (int off = 32; off > 0; off >>= 1) ninf += __shfl_down(ninf, off, 64);
  if ((TIDX & 63) == 0) s_inf[TIDX >> 6] = ninf;
  __syncthreads();
  if (TIDX == 0) {
    int tot = 0;
    for (int w = 0; w < (bs >> 6); w++) tot += s_inf[w];
    c->p1_nchg = total;
    c->p1_init = 0;
    c->p1_fix_q = 0;
    if (tot == 0) {
      c->done = D_PFEAS;
      c->step = ST_NONE;
    }
  }
}

// cost[j] = fma(d_k, T[i_k][j], cost[j]) over the listed rows, in list order; one lane per column
__global__ __launch_bounds__(256) void k_p1_fix(Ctl *c) {
  if (c->done != D_RUN) return;
  const int nchg = c->p1_nchg;
  if (nchg == 0) return;
  const int j = (int)blockIdx.x * 256 + TIDX;
  if (j > c->n) return;
  const size_t ld = (size_t)c->ld;
  const double *T = c->T;
  double *cost = c->T + (size_t)(c->m + 1) * ld;
  const int *list = c->p1_list;
  const double *dv = c->wts;
  double v = cost[j];
  constexpr int U = 8;
  int k = 0;
  for (; k + U <= nchg; k += U) {
    double t[U], d[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      t[u] = T[(size_t)list[k + u] * ld + j];
      d[u] = dv[k + u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) v = fma(d[u], t[u], v);
  }
  for (; k < nchg; k++) v = fma(dv[k], T[(size_t)list[k] * ld + j], v);
  cost[j] = v;
}

__global__ __launch_bounds__(1024) void k_p1_select(Ctl *c) {
  __shared__ Cand lds[17];
  const KC k = load_kc(c);
  if (c->done != D_RUN) return;
  const double *cost = k.T + (size_t)(k.m + 1) * k.ld;
  Cand pr = dev_price(k, cost, 1.0, lds);
  if (pr.idx == 0) {
    dev_finish(c, D_NOFEAS, PH_PHASE1, c->rounds);
    return;
  }
  if (c->budget == 0) {
    dev_finish(c, D_ITLIM, PH_PHASE1, c->rounds);
    return;
  }
  if (!dev_primal_step(k, c, pr.idx, pr.aux, c->gflag, lds)) {
    dev_finish(c, D_FAIL, PH_PHASE1, c->rounds);
    return;
  }
  if (TIDX == 0 && c->step == ST_PIVOT) {
    // the cost row rides along as row m+1 of the update; its leaving-variable term is dropped by the next head
    const int p = c->p, q = c->q;
    k.colq[k.m + 1] = cost[q];
    c->p1_fix_q = q;
    c->p1_fix_g = c->gflag[p];
    c->gflag[p] = 0;
  }
}

// ---------------------------------------------------------------------------- k_refresh_select
// Tableau refresh (engine.cpp refresh_tableau; oracle: refresh_tableau): structural variable `var` has to be basic.
// Its column is wherever the slack tableau / the pivots so far left it; it enters on the row of largest |entry|
// (lowest row on ties) among the rows whose basic auxiliary is non-basic in the target basis, which leaves to the
// bound `tflag` names.  Prepares the pivot for k_update exactly as k_select does.
__global__ __launch_bounds__(1024) void k_refresh_select(Ctl *c, const int *tflag, int var) {
  __shared__ Cand lds[17];
  __shared__ int s_q;
  const KC k = load_kc(c);
  if (TIDX == 0) {
    s_q = 0;
    c->step = ST_NONE;
  }
  __syncthreads();
  for (int j = 1 + TIDX; j <= k.n; j += (int)blockDim.x)
    if (k.nvar[j] == var) s_q = j;
  __syncthreads();
  const int q = s_q;
  if (q == 0) return;
  const size_t ld = (size_t)k.ld;
  Cand best{0.0, 0.0, 0, 0};
  for (int i = TIDX; i <= k.m; i += (int)blockDim.x) {
    const double a = k.T[(size_t)i * ld + q];
    k.colq[i] = a;
    if (i == 0) continue;
    const int v = k.bvar[i];
    if (v > k.m || tflag[v] == 0) continue;
    const double mag = fabs(a);
    Cand x{mag, 0.0, i, tflag[v]};
    if (mag > 0.0 && cand_better<0>(x, best)) best = x;
  }
  best = block_best<0>(best, lds);
  if (best.idx == 0) return;
  const int p = best.idx, tf = best.aux;
  const double *rowp = k.T + (size_t)p * ld;
  const double piv = rowp[q];
  const double bound = (tf == MVX_NU) ? k.bub[p] : (tf == MVX_NF ? 0.0 : k.blb[p]);
  for (int j = TIDX; j <= k.n; j += (int)blockDim.x) {
    const double v = rowp[j];
    k.srow[j] = (j == 0) ? xdiv(v - bound, piv) : xdiv(v, piv);
  }
  if (TIDX == 0) {
    c->step = ST_PIVOT;
    c->p = p;
    c->q = q;
    c->p_up = (tf == MVX_NU);
    c->piv = piv;
    c->bound = bound;
    c->xq = dev_nb_value(k.nflag[q], k.nlb[q], k.nub[q]);
    c->leave_flag = tf;
  }
}
void launch_refresh_select(Ctl *d_ctl, const int *tflag, int var, hipStream_t s) {
  hipLaunchKernelGGL(k_refresh_select, dim3(1), dim3(1024), 0, s, d_ctl, tflag, var);
}

// ---------------------------------------------------------------------------- k_update
template <int NT>
__device__ __forceinline__ double2 ld2(const double2 *p) {
  if (NT == 1) {
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  }
  return *p;
}
template <int NT>
__device__ __forceinline__ void st2(double2 *p, double2 v) {
  if (NT == 2) {
    // write-through store (sc1): the line goes to memory as it is written instead of waiting in L2 as a dirty line for
    // its eviction.  Measured on tableaux around the Infinity Cache size (scripts/ntsweep.py); the s_nop covers the
    // wait state a >8-byte store needs before its data registers may be rewritten (inline asm is not hazard-checked).
    typedef double d2v __attribute__((ext_vector_type(2)));
    d2v w = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
  } else if (NT == 1) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else
    *p = v;
}

// Gauss-Jordan rank-1 update, streamed: every tableau entry read once and written once.
//   T[i][j] = fma(-colq[i], srow[j], T[i][j])   (i != p, j != q)
//   T[i][q] = colq[i] / piv                      (i != p)
//   T[p][j] = -srow[j], T[p][q] = 1/piv, T[p][0] = xq - srow[0]
// Block = 256 lanes x 2 columns (16 B per lane, 4 KB per row segment), TR rows deep.
// k_update for a chain of dual pivots (dual_chain): the tile is loaded once, the steps are applied in registers in
// order, stored once; the basis swaps of the chain in order.  Rows 0..m (the objective row is an ordinary row here).
template <int TR, int NT>
__device__ __forceinline__ void update_chain(Ctl *c, int nch) {
  __shared__ ChainStep s_st[KCH];
  __shared__ const double *s_cq[KCH], *s_sr[KCH];
  const int m = c->m, n = c->n;
  const size_t ld = (size_t)c->ld;
  if (TIDX < nch) {
    const int l = TIDX;
    s_st[l] = ChainStep{c->ch_p[l], c->ch_q[l], c->ch_lf[l], c->ch_piv[l], c->ch_xq[l], c->ch_s0[l]};
    s_cq[l] = (l == 0) ? c->colq : c->colqk[l];
    s_sr[l] = (l == 0) ? c->srow : c->srowk[l];
  }
  __syncthreads();
  if (blockIdx.x == 0 && blockIdx.y == 0 && TIDX == 0) {
    for (int l = 0; l < nch; l++) {
      const int p = s_st[l].p, q = s_st[l].q;
      const int kv = c->bvar[p];
      const double klb = c->blb[p], kub = c->bub[p];
      c->bvar[p] = c->nvar[q];
      c->blb[p] = c->nlb[q];
      c->bub[p] = c->nub[q];
      c->nvar[q] = kv;
      c->nlb[q] = klb;
      c->nub[q] = kub;
      c->nflag[q] = s_st[l].lf;
    }
    c->it_cnt += nch;
    c->n_bulk++;
    if (c->budget > 0) c->budget -= nch;
  }
  const int j0 = 2 * ((int)blockIdx.x * 256 + TIDX);
  const int i0 = (int)blockIdx.y * TR;
  if (i0 > m || 2 * (int)blockIdx.x * 256 > n) return; // block-uniform: the whole block is outside this slot's tableau
  // the pivot-column entries of the tile for every step, one load per lane (DCH_MAX * TR <= 256)
  __shared__ double s_ci[DCH_MAX][TR];
  if (TIDX < nch * TR) s_ci[TIDX / TR][TIDX % TR] = s_cq[TIDX / TR][i0 + TIDX % TR];
  __syncthreads();
  if (j0 > n) return;
  double *base = c->T + (size_t)i0 * ld + j0;
  double2 v[TR];
#pragma unroll
  for (int r = 0; r < TR; r++) v[r] = ld2<NT>(reinterpret_cast<const double2 *>(base + (size_t)r * ld));
  double2 s_n = *reinterpret_cast<const double2 *>(uniform_ptr(s_sr[0]) + j0);
  for (int l = 0; l < nch; l++) {
    const ChainStep st = s_st[l];
    const double2 s = s_n;
    if (l + 1 < nch) s_n = *reinterpret_cast<const double2 *>(uniform_ptr(s_sr[l + 1]) + j0);
    double ci[TR];
#pragma unroll
    for (int r = 0; r < TR; r++) ci[r] = s_ci[l][r];
    const bool q0 = (j0 == st.q), q1 = (j0 + 1 == st.q);
#pragma unroll
    for (int r = 0; r < TR; r++) {
      v[r].x = fma(-ci[r], s.x, v[r].x);
      v[r].y = fma(-ci[r], s.y, v[r].y);
    }
    if (q0 || q1) {
#pragma unroll
      for (int r = 0; r < TR; r++) {
        const double qv = xdiv(ci[r], st.piv);
        if (q0) v[r].x = qv;
        if (q1) v[r].y = qv;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (st.p >= i0 && st.p < i0 + TR) {
#pragma unroll
      for (int r = 0; r < TR; r++) {
        if (i0 + r == st.p) {
          v[r].x = q0 ? xdiv(1.0, st.piv) : -s.x;
          v[r].y = q1 ? xdiv(1.0, st.piv) : -s.y;
          if (j0 == 0) v[r].x = st.xq - s.x;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < TR; r++) st2<NT>(reinterpret_cast<double2 *>(base + (size_t)r * ld), v[r]);
}

template <int TR, int NT>
__global__ __launch_bounds__(256) void k_update(Ctl *c, int chained) {
  c += blockIdx.z;
  if (c->done != D_RUN || c->step != ST_PIVOT) return;
  if (chained) {
    const int nch = c->nch;
    if (nch > 1) {
      update_chain<TR, NT>(c, nch);
      return;
    }
  }
  const int m = c->m, n = c->n, p = c->p, q = c->q;
  const size_t ld = (size_t)c->ld;
  const double piv = c->piv;
  if (blockIdx.x == 0 && blockIdx.y == 0 && TIDX == 0) {
    // basis bookkeeping: entering variable takes row p, leaving variable takes column q
    const int kv = c->bvar[p];
    const double klb = c->blb[p], kub = c->bub[p];
    c->bvar[p] = c->nvar[q];
    c->blb[p] = c->nlb[q];
    c->bub[p] = c->nub[q];
    c->nvar[q] = kv;
    c->nlb[q] = klb;
    c->nub[q] = kub;
    c->nflag[q] = c->leave_flag;
    c->it_cnt++;
    c->n_bulk++;
    if (c->budget > 0) c->budget--;
  }
  const int j0 = 2 * ((int)blockIdx.x * 256 + TIDX);
  const int i0 = (int)blockIdx.y * TR;
  // a batched launch is sized for its largest slot: blocks wholly outside this slot's tableau leave;
  // a block that straddles row m streams on into the spare rows every slab keeps behind it (ROW_SPARE)
  if (j0 > n || i0 > m + (c->phase == PH_PHASE1 ? 1 : 0)) return; // phase 1: row m+1 is its cost row
  const double2 s = *reinterpret_cast<const double2 *>(c->srow + j0);
  const bool q0 = (j0 == q), q1 = (j0 + 1 == q);
  double *base = c->T + (size_t)i0 * ld + j0;
  const double *colq = c->colq + i0;
  // straight-line stream (same shape as k_fb's hot path): every load issued before the first use
  double2 v[TR];
  double ci[TR];
#pragma unroll
  for (int r = 0; r < TR; r++) v[r] = ld2<NT>(reinterpret_cast<const double2 *>(base + (size_t)r * ld));
#pragma unroll
  for (int r = 0; r < TR; r++) ci[r] = colq[r];
#pragma unroll
  for (int r = 0; r < TR; r++) {
    v[r].x = fma(-ci[r], s.x, v[r].x);
    v[r].y = fma(-ci[r], s.y, v[r].y);
  }
  if (q0 || q1) {
#pragma unroll
    for (int r = 0; r < TR; r++) {
      const double qv = xdiv(ci[r], piv);
      if (q0) v[r].x = qv;
      if (q1) v[r].y = qv;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (p >= i0 && p < i0 + TR) {
#pragma unroll
    for (int r = 0; r < TR; r++) {
      if (i0 + r == p) {
        v[r].x = q0 ? xdiv(1.0, piv) : -s.x;
        v[r].y = q1 ? xdiv(1.0, piv) : -s.y;
        if (j0 == 0) v[r].x = c->xq - s.x;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < TR; r++) st2<NT>(reinterpret_cast<double2 *>(base + (size_t)r * ld), v[r]);
}

// ---------------------------------------------------------------------------- k_rowcomb
// out[j] = base[j] + sum over 64-row chunks (in order) of fma-chain sum_i w[i]*T[i][j].
__global__ __launch_bounds__(256) void k_rowcomb_partial(Ctl *c, int respect_done) {
  if (respect_done && c->done != D_RUN) return;
  const int j = (int)blockIdx.x * 256 + TIDX;
  if (j > c->n) return;
  const int chunk = (int)blockIdx.y;
  const int i0 = 1 + chunk * ROWCOMB_CHUNK;
  int i1 = i0 + ROWCOMB_CHUNK - 1;
  if (i1 > c->m) i1 = c->m;
  const size_t ld = (size_t)c->ld;
  double acc = 0.0;
  for (int i = i0; i <= i1; i++) {
    const double w = c->wts[i];
    if (w != 0.0) acc = fma(w, c->T[(size_t)i * ld + j], acc);
  }
  c->part[(size_t)chunk * ld + j] = acc;
}

__global__ __launch_bounds__(256) void k_rowcomb_final(Ctl *c, int respect_done, int nchunks) {
  if (respect_done && c->done != D_RUN) return;
  const int j = (int)blockIdx.x * 256 + TIDX;
  if (j > c->n) return;
  const size_t ld = (size_t)c->ld;
  double out = c->rc_base ? c->rc_base[j] : 0.0;
  for (int ch = 0; ch < nchunks; ch++) out = out + c->part[(size_t)ch * ld + j];
  c->rc_out[j] = out;
}

// beta += T[:, jj] * delta  (a non-basic variable moved by delta)
__global__ __launch_bounds__(256) void k_shift_nonbasic(double *T, int ld, int m, int jj, double delta) {
  const int i = (int)blockIdx.x * 256 + TIDX;
  if (i > m) return;
  double *row = T + (size_t)i * ld;
  row[0] = fma(row[jj], delta, row[0]);
}

__global__ __launch_bounds__(64) void k_set_basic_bounds(double *blb, double *bub, int i, double lb, double ub) {
  if (TIDX == 0) {
    blb[i] = lb;
    bub[i] = ub;
  }
}
__global__ __launch_bounds__(64) void k_set_nonbasic(double *nlb, double *nub, int *nflag, int j, double lb, double ub, int flag) {
  if (TIDX == 0) {
    nlb[j] = lb;
    nub[j] = ub;
    nflag[j] = flag;
  }
}

// new empty rows first..last: zero body, basic auxiliary first.., free bounds; shift the
// structural variable numbers by nrs
__global__ __launch_bounds__(256) void k_add_rows(double *T, int ld, int n, int *bvar, double *blb, double *bub, int *nvar,
                                                  int first, int nrs) {
  const int t = (int)blockIdx.x * 256 + TIDX;
  for (int r = 0; r < nrs; r++) {
    if (t <= n) T[(size_t)(first + r) * ld + t] = 0.0;
  }
  if (t < nrs) {
    bvar[first + t] = first + t;
    blb[first + t] = -INFINITY;
    bub[first + t] = INFINITY;
  }
  if (t >= 1 && t < first && bvar[t] >= first) bvar[t] += nrs;
  if (t >= 1 && t <= n && nvar[t] >= first) nvar[t] += nrs;
}

// ---------------------------------------------------------------------------- k_export
// Pack what the host needs after a solve into one staging buffer:
//   [Ctl][beta (m_cap+1) f64][d (ld) f64][bvar (m_cap+1) i32][nvar (ld) i32][nflag (ld) i32]
__global__ __launch_bounds__(256) void k_export(Ctl *c, unsigned char *stage, int force, size_t slot_stride) {
  c += blockIdx.z;
  stage += (size_t)blockIdx.z * slot_stride;
  const int t = (int)blockIdx.x * 256 + TIDX;
  if (blockIdx.x == 0) { // the control block, a word per thread (one thread copying its 6 KB alone took 12 us)
    const unsigned *src = reinterpret_cast<const unsigned *>(c);
    unsigned *dst = reinterpret_cast<unsigned *>(stage);
    for (int w = TIDX; w < (int)(sizeof(Ctl) / 4); w += 256) dst[w] = src[w];
  }
  if (c->T == nullptr) return; // idle slot of a batched launch
  if (!force && c->done == D_RUN) return;
  pack_mirrors(c, stage, t, (int)gridDim.x * 256);
}

// ======================================================================= fused primal path
// Two multi-workgroup kernels per primal phase-2 pivot, no single-CU stage:
//   k_fa  (one block per 256 columns): reduces the partials left by the previous step to the
//         entering column q and leaving row p, scales the pivot row into srow, updates the
//         objective row (so the NEXT entering column can be priced before the bulk update runs)
//         and leaves new pricing partials.
//   k_fb  (2-D grid over the tableau): the streamed rank-1 update; the lanes that own column 0
//         and the next entering column also export them contiguously (betac / colqx) and leave
//         per-row-block ratio-test partials for the next k_fa.
// Every reduction key is a strict total order, so the redundant per-block reductions agree.
// Arithmetic per entry is identical to k_select/k_update (and to the oracle).

// devex score of column j: d_j^2 / w (same expression as dev_price modes 1 / 2)
__device__ __forceinline__ bool price_col(int f, double dj, double tol, int j, double w, Cand &x) {
  if (f == MVX_NS) return false;
  const bool up = (f == MVX_NL || f == MVX_NF) && dj > tol;
  const bool dn = (f == MVX_NU || f == MVX_NF) && dj < -tol;
  if (!up && !dn) return false;
  x = Cand{xdiv(dj * dj, w), 0.0, j, up ? 1 : -1};
  return true;
}

// Row block rb covers rows 1 + rb*TR .. (rb+1)*TR; row 0 (objective) belongs to k_fa.  The slab
// always has at least 32 spare rows behind row m (mvx::ROW_SPARE), so the last block streams
// whole tiles too: spare rows are never read by anything else and are re-zeroed when a cut row
// is appended (k_add_rows).
template <int TR, int HOT, int NT, int DUAL>
__device__ __forceinline__ void fb_body(Ctl *c) {
  const int cur = c->curB, nxt = cur ^ 1;
  const int step = c->step;
  const int m = c->m, n = c->n, p = c->p, q = c->q;
  const size_t ld = (size_t)c->ld;
  const double piv = c->piv;
  const int j0 = 2 * ((int)blockIdx.x * 256 + TIDX);
  const bool active = (j0 <= n);
  const int i0 = 1 + (int)blockIdx.y * TR;
  const bool has0 = (j0 == 0);
  const bool tile0 = (blockIdx.x == 0);
  const double *colq = c->colqx[cur];
  double *bnew = c->betac[nxt];
  // pricing partial of the NEXT step: requested now, reduced after the stream has been issued
  const int npb = DUAL ? c->npbd : c->npb;
  const Cand *ppn = c->pp[nxt];
  Cand ncv = ((TIDX & 63) < npb) ? ppn[TIDX & 63] : Cand{0.0, 0.0, 0, 0};
  if (step == ST_PIVOT) {
    if (active) {
      const double2 s = *reinterpret_cast<const double2 *>(c->srow + j0);
      const bool q0 = (j0 == q), q1 = (j0 + 1 == q);
      double *base = c->T + (size_t)i0 * ld + j0;
      if (HOT == 1) {
        // straight-line stream: every load issued before the first use
        double2 v[TR];
        double ci[TR];
#pragma unroll
        for (int r = 0; r < TR; r++) v[r] = ld2<NT>(reinterpret_cast<const double2 *>(base + (size_t)r * ld));
#pragma unroll
        for (int r = 0; r < TR; r++) ci[r] = colq[i0 + r];
#pragma unroll
        for (int r = 0; r < TR; r++) {
          v[r].x = fma(-ci[r], s.x, v[r].x);
          v[r].y = fma(-ci[r], s.y, v[r].y);
        }
        if (q0 || q1) {
          // only the wave that owns column q divides; the barrier keeps the division sequences
          // (about 10 temporaries each) from being interleaved, so the kernel keeps its occupancy
#pragma unroll
          for (int r = 0; r < TR; r++) {
            const double qv = xdiv(ci[r], piv);
            if (q0) v[r].x = qv;
            if (q1) v[r].y = qv;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (p >= i0 && p < i0 + TR) {
#pragma unroll
          for (int r = 0; r < TR; r++) {
            if (i0 + r == p) {
              v[r].x = q0 ? xdiv(1.0, piv) : -s.x;
              v[r].y = q1 ? xdiv(1.0, piv) : -s.y;
              if (has0) v[r].x = c->xq - s.x;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < TR; r++) st2<NT>(reinterpret_cast<double2 *>(base + (size_t)r * ld), v[r]);
        if (has0) {
#pragma unroll
          for (int r = 0; r < TR; r++) bnew[i0 + r] = v[r].x;
        }
      } else {
        for (int r = 0; r < TR; r++) {
          const int i = i0 + r;
          if (i > m) break;
          double2 *ptr = reinterpret_cast<double2 *>(base + (size_t)r * ld);
          double2 v = *ptr;
          const double ci = colq[i];
          if (i == p) {
            v.x = q0 ? xdiv(1.0, piv) : -s.x;
            v.y = q1 ? xdiv(1.0, piv) : -s.y;
            if (has0) v.x = c->xq - s.x;
          } else {
            v.x = fma(-ci, s.x, v.x);
            v.y = fma(-ci, s.y, v.y);
            if (q0) v.x = xdiv(ci, piv);
            if (q1) v.y = xdiv(ci, piv);
          }
          *ptr = v;
          if (has0) bnew[i] = v.x;
        }
      }
    }
  } else if (tile0 && TIDX < TR && i0 + TIDX <= m) {
    // bound flip: only column 0 moves; bootstrap: nothing moves.  One lane per row.
    const int i = i0 + TIDX;
    double *b0 = c->T + (size_t)i * ld;
    double beta = *b0;
    if (step == ST_FLIP) {
      beta = fma(colq[i], c->delta, beta);
      *b0 = beta;
    }
    bnew[i] = beta;
  }
  // next entering column (0 = none: k_fa will stop); reduced after the stream has been issued
  for (int k = (TIDX & 63) + 64; k < npb; k += 64) {
    Cand x = ppn[k];
    if (cand_better<DUAL ? 1 : 0>(x, ncv)) ncv = x; // primal: best pricing score; dual: smallest dual ratio
  }
  const Cand nc = wave_bcast_best<DUAL ? 1 : 0>(ncv);
  const int qn = nc.idx, sdn = nc.aux;
  // 512 columns per tile; the bootstrap launch (nothing to stream) has one block per row block, which stands in for
  // whichever tile owns the column
  const bool tilen = (qn != 0 && ((qn >> 9) == (int)blockIdx.x || (step == ST_NONE && gridDim.x == 1)));
  if (tilen) {
    // export the next entering column contiguously and leave this row block's ratio-test partial;
    // one lane per row, values re-read after the block's own stores (the write-through stores are inline asm, which
    // the compiler's wait-count bookkeeping does not see: wait for them by hand before the barrier)
    if (NT == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (TIDX < 64) {
      Cand best{0.0, 0.0, 0, 0};
      for (int r = TIDX; r < TR; r += 64) {
        const int i = i0 + r;
        if (i > m) break;
        const double a = c->T[(size_t)i * ld + qn];
        c->colqx[nxt][i] = a;
        if (DUAL) continue; // the dual side needs the column only: its leaving row is chosen from column 0 by k_da
        double beta, lb = c->blb[i], ub = c->bub[i];
        if (step == ST_PIVOT) {
          if (i == p) {
            beta = c->xq - c->srow[0];
            lb = c->ent_lb;
            ub = c->ent_ub;
          } else
            beta = fma(-colq[i], c->srow[0], c->betac[cur][i]);
        } else if (step == ST_FLIP) {
          beta = fma(colq[i], c->delta, c->betac[cur][i]);
        } else {
          beta = c->T[(size_t)i * ld];
        }
        Cand x{0.0, 0.0, 0, 0};
        if (ratio_row(a, sdn, beta, lb, ub, 0, c->tol_piv, i, x) && cand_better<1>(x, best)) best = x;
      }
      best = wave_best<1>(best);
      if (TIDX == 0 && !DUAL) c->rp[blockIdx.y] = best;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && TIDX == 0) {
    // objective row exports (row 0 is outside the row blocks)
    double z = c->T[0];
    if (step == ST_FLIP) {
      z = fma(colq[0], c->delta, z);
      c->T[0] = z;
    }
    bnew[0] = z;
    if (qn != 0) c->colqx[nxt][0] = c->T[qn];
    if (step == ST_PIVOT) {
      const int kv = c->bvar[p];
      const double klb = c->blb[p], kub = c->bub[p];
      c->bvar[p] = c->nvar[q];
      c->blb[p] = c->nlb[q];
      c->bub[p] = c->nub[q];
      c->nvar[q] = kv;
      c->nlb[q] = klb;
      c->nub[q] = kub;
      c->nflag[q] = c->leave_flag;
      c->it_cnt++;
      c->n_bulk++;
      if (c->budget > 0) c->budget--;
      c->stall = c->stall_new;
    } else if (step == ST_FLIP) {
      c->nflag[q] = c->flipflag;
      c->n_flips++;
      c->n_bulk++;
      c->stall = c->stall_new;
    }
    if (!DUAL && step == ST_NONE) c->phase = PH_PRIMAL2; // a run taken over at PH_START by k_fboot
    c->curA = nxt;
    c->nrb = (int)gridDim.y; // number of ratio-test partials this launch leaves for k_fa
  }
}
template <int TR, int HOT, int NT, int DUAL>
__global__ __launch_bounds__(256) void k_fb(Ctl *c) {
  if (c->done != D_RUN || c->fstate != (DUAL ? F_RUN_DUAL : F_RUN)) return;
  fb_body<TR, HOT, NT, DUAL>(c);
}

// ================================================================= chained primal path
// Primal phase 2 of large tableaux.  The selection of a pivot reads O(m + n) entries of the tableau -- the objective
// row, the entering column, the leaving row -- while the update touches all (m+1)(n+1).  So the pivots that follow one
// another are chosen before any bulk update has run: column q_g and row p_g are read from the tableau as it stands in
// memory and carried through steps 0..g-1 of the pending chain entry by entry, with exactly the operations the bulk
// update would have applied; ONE bulk launch then streams the tableau once and applies the whole chain.  Same pivots,
// same bits as one pass per pivot -- the traffic per pivot is the pass divided by the chain length.
//   k_pboot      start of a call: is the basis primal feasible; devex weights; pricing partials of the objective row;
//                column 0 exported contiguously
//   k_pc(g)      column phase of step g (one lane per row): entering column q_g = reduction of the pricing partials;
//                column q_g carried through the chain; ratio-test partial per block
//   k_pr(g)      row phase of step g (one lane per column): leaving row p_g = reduction of those partials (or the bound
//                flip of the entering variable: a step of its own kind); row p_g carried through the chain -> scaled
//                pivot row; objective row, devex weights, statuses and bounds of the columns as the step leaves them;
//                pricing partials for step g+1
//   k_fbc3       the bulk pass: the plain multiply-adds of every step
//   k_fpatch     the chain's pivot rows, pivot columns and column 0; the chain's bookkeeping
// What two rounds of in-kernel stamps taught (scripts/fcsdbg.py): a step is a chain of dependent
// memory round trips, so (1) pointers, geometry and tolerances come by value (ChainArgs): every load whose address is
// known on entry is requested on entry; (2) loads are unconditional -- indices are clamped, results masked -- because a
// load behind a run-time condition makes the compiler wait for it on the spot; (3) the description of the pending
// chain sits in the registers of lanes 0..g-1 of every wave and is broadcast with v_readlane instead of being staged in
// LDS; (4) what a step changes besides the tableau (basic values, bounds, statuses, objective row, weights) is carried
// from step to step in two alternating sets of small arrays, so no launch re-derives it from the chain; (5) one
// barrier per launch; every store at the end.  A chain ends early on: no entering column, the pivot limit, the stall
// limit (Bland's rule and the perturbation stay with the generic step), an unbounded ray.

// block-wide arg-best with ONE barrier: every wave leaves its best in its own slot, every wave reduces the four slots;
// returns the winning wave's number in *bw
template <int MODE>
__device__ __forceinline__ Cand block_best1(Cand x, Cand *slots, int *bw) {
  if ((TIDX & 63) == 0) slots[TIDX >> 6] = x; // x: the wave's best, already uniform within the wave
  __syncthreads();
  Cand r = slots[0];
  int w0 = 0;
#pragma unroll
  for (int w = 1; w < 4; w++) {
    const Cand y = slots[w];
    if (cand_better<MODE>(y, r)) {
      r = y;
      w0 = w;
    }
  }
  *bw = w0;
  return r;
}

enum : int { PPF_SCORE = 0, PPF_Q, PPF_SDIR, PPF_DQ, PPF_WQ, PPF_LB, PPF_UB, PPF_FQ };
enum : int { RPF_K1 = 0, RPF_K2, RPF_IDX, RPF_AUX };

// a column block's best entering column and what the column phase needs of it, taken from the lane that owns it;
// thread 0 writes the block's pricing partial.  Ends with the launch's one barrier.
__device__ __forceinline__ void publish_pricing(const ChainArgs &A, Cand best, int b, double d, double w, double lbj, double ubj, int f,
                                                Cand *slots, double (*pay)[5]) {
  const int lane = TIDX & 63, wave = TIDX >> 6;
  const Cand wb = wave_bcast_best<0>(best);
  const int ol = wb.idx ? ((wb.idx - b * 256) & 63) : 0;
  const double p0 = rl_d(d, ol), p1 = rl_d(w, ol), p2 = rl_d(lbj, ol), p3 = rl_d(ubj, ol);
  const int p4 = rl_i(f, ol);
  if (lane == 0) {
    pay[wave][0] = p0;
    pay[wave][1] = p1;
    pay[wave][2] = p2;
    pay[wave][3] = p3;
    pay[wave][4] = (double)p4;
  }
  int bw;
  const Cand bb = block_best1<0>(wb, slots, &bw);
  if (TIDX == 0) {
    double *pp = A.pp + b;
    const size_t st = A.ppstride;
    pp[PPF_SCORE * st] = bb.k1;
    pp[PPF_Q * st] = (double)bb.idx;
    pp[PPF_SDIR * st] = (double)bb.aux;
    pp[PPF_DQ * st] = pay[bw][0];
    pp[PPF_WQ * st] = pay[bw][1];
    pp[PPF_LB * st] = pay[bw][2];
    pp[PPF_UB * st] = pay[bw][3];
    pp[PPF_FQ * st] = pay[bw][4];
  }
}

// Start of a call.  A call that has just begun (phase PH_START, no bound edits waiting) is taken from its first pivot:
// every block repeats select_step's opening check -- no basic variable outside its bounds by more than the tolerance --
// and, if that holds, the primal devex weights restart from one exactly as the generic step's `fresh_primal` does.
// Otherwise (dual simplex, phase 1, Bland's rule in force) the pipeline is turned off and its launches return at once.
__global__ __launch_bounds__(256) void k_pboot(const ChainArgs A) {
  __shared__ Cand s_slots[4];
  __shared__ double s_pay[4][5];
  Ctl *const c = A.c;
  const int b = (int)blockIdx.x;
  const bool lead = (b == 0 && TIDX == 0);
  const int m = A.m, n = A.n;
  const size_t ld = (size_t)A.ld;
  const double *const T = A.T;
  const int phase = c->phase;
  const bool fresh = (phase == PH_START);
  if (c->done != D_RUN || c->stall >= A.stall_limit || !(phase == PH_PRIMAL2 || (fresh && c->n_edits == 0))) { // no Bland pricing here
    if (lead) c->fstate = F_OFF;
    return;
  }
  const int src = c->curA & 1;
  const int i = 1 + b * 256 + TIDX, ic = (i <= m) ? i : m;
  const double bi = T[(size_t)ic * ld];
  if (fresh) {
    const double tolb = A.tol_bnd;
    int bad = 0;
    for (int i0 = 1 + TIDX; i0 <= m; i0 += 256 * 8) {
      double xb[8], xl[8], xu[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int r = i0 + 256 * u;
        const int rc = (r <= m) ? r : m;
        xb[u] = T[(size_t)rc * ld];
        xl[u] = A.blb[rc];
        xu[u] = A.bub[rc];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (xl[u] > -INFINITY && xb[u] < xl[u] - tolb * (1.0 + fabs(xl[u]))) bad = 1;
        if (xu[u] < INFINITY && xb[u] > xu[u] + tolb * (1.0 + fabs(xu[u]))) bad = 1;
      }
    }
    if (__syncthreads_or(bad)) { // not primal feasible: the generic step decides between the dual simplex and phase 1
      if (lead) c->fstate = F_OFF;
      return;
    }
  }
  if (i <= m) A.betab[i] = bi;
  const int j = b * 256 + TIDX;
  if (b < A.ncb) {
    const bool act = (j <= n);
    const int jc = act ? j : n;
    const double d = T[jc];
    const double w = fresh ? 1.0 : A.pw[src][jc];
    const int f = (act && j >= 1) ? A.nflag[jc] : MVX_NS;
    const double lbj = A.nlb[jc], ubj = A.nub[jc];
    if (act) {
      A.pw[0][j] = w; // both sets current: the generic step reads pw[curA & 1]
      A.pw[1][j] = w;
    }
    Cand best{0.0, 0.0, 0, 0};
    if (f != MVX_NS) {
      Cand x{0.0, 0.0, 0, 0};
      if (price_col(f, A.sgn * d, A.tol_dj, j, w, x)) best = x;
    }
    publish_pricing(A, best, b, d, w, lbj, ubj, f, s_slots, s_pay);
  }
  if (lead) {
    c->fstate = F_RUN;
    c->step = ST_NONE;
    c->pc_n = 0;
  }
}

// the chain so far, one step per lane (lanes 0..g-1 of every wave; lanes past it re-read its last step: no guard)
struct LaneRec {
  int kind, p, q;
  double piv, xq, ip; // ip = 1 / piv
};
__device__ __forceinline__ LaneRec lane_rec(const Ctl *c, int g) {
  const int lane = TIDX & 63;
  const int rl = (lane < g) ? lane : (g > 0 ? g - 1 : 0);
  LaneRec r;
  r.kind = c->ch_kind[rl];
  r.p = c->ch_p[rl];
  r.q = c->ch_q[rl];
  r.piv = c->ch_piv[rl];
  r.xq = c->ch_xq[rl];
  r.ip = 1.0;
  return r;
}

// Column phase of step g.
__global__ __launch_bounds__(256) void k_pc(const ChainArgs A, int g) {
  __shared__ Cand s_slots[4];
  Ctl *const c = A.c;
  const int b = (int)blockIdx.x;
  const bool lead = (b == 0 && TIDX == 0);
  const int lane = TIDX & 63;
  const int m = A.m, ncb = A.ncb;
  const size_t ld = (size_t)A.ld;
  const double *const T = A.T;
  unsigned long long *const dbg = c->dbg;
#define PC_STAMP(k) do { if (dbg && lead) dbg[(size_t)g * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  PC_STAMP(0);
  // ---- entry: everything whose address is known
  const int done = c->done, fstate = c->fstate, epoch = c->pc_epoch, budget = c->budget;
  const int okprev = g ? c->ch_ok[g - 1] : 0;
  const int stall = g ? c->ch_stall[g - 1] : c->stall;
  const int used = g ? c->ch_cnt[g - 1] : 0;
  const int i = 1 + b * 256 + TIDX;
  const bool act = (i <= m);
  const int ic = act ? i : m;
  // the row side as of step g-1 (set (g-1) & 1; the handle's own arrays while the chain has no step) ...
  const int xs = (g - 1) & 1;
  const double *bsrc = (g <= 1) ? A.betab : A.betak[xs], *lsrc = (g <= 1) ? A.blb : A.blbk[xs], *usrc = (g <= 1) ? A.bub : A.bubk[xs];
  double be = bsrc[ic], lb = lsrc[ic], ub = usrc[ic];
  // ... step g-1 itself, which this launch applies to it ...
  const int gp = g ? g - 1 : 0;
  const int pk = c->ch_kind[gp], pp_ = c->ch_p[gp];
  const double pxq = c->ch_xq[gp], ps0 = c->ch_s0[gp], pdelta = c->ch_delta[gp], pelb = c->ch_elb[gp], peub = c->ch_eub[gp];
  // ... this lane's row of the chain's pivot columns ...
  double ca[16], cb[16];
#pragma unroll
  for (int s = 0; s < 16; s++) ca[s] = A.colq0[(size_t)s * A.cstride + ic];
  if (g > 16) {
#pragma unroll
    for (int s = 0; s < 16; s++) cb[s] = A.colq0[(size_t)(16 + s) * A.cstride + ic];
  } else {
#pragma unroll
    for (int s = 0; s < 16; s++) cb[s] = 0.0;
  }
  LaneRec rec = lane_rec(c, g);
  // ... and the pricing partials: every wave reduces them on its own; a lane keeps the whole record of its best one
  double m_sc = 0.0, m_dq = 0.0, m_wq = 1.0, m_lb = 0.0, m_ub = 0.0;
  int m_q = 0, m_sd = 0, m_fq = 0;
  for (int t = lane; t < ncb; t += 64) {
    const double *pp = A.pp + t;
    const size_t st = A.ppstride;
    const double sc = pp[PPF_SCORE * st];
    const int q = (int)pp[PPF_Q * st];
    const int sd = (int)pp[PPF_SDIR * st];
    const double dq = pp[PPF_DQ * st], wq = pp[PPF_WQ * st], l0 = pp[PPF_LB * st], u0 = pp[PPF_UB * st];
    const int fq = (int)pp[PPF_FQ * st];
    if (q != 0 && (m_q == 0 || sc > m_sc || (sc == m_sc && q < m_q))) {
      m_sc = sc; m_q = q; m_sd = sd; m_dq = dq; m_wq = wq; m_lb = l0; m_ub = u0; m_fq = fq;
    }
  }
  if (done != D_RUN || fstate != F_RUN) return;
  if (lead && g == 0) c->pc_n = 0; // a new chain: nothing recorded yet (the bulk pass and k_fpatch apply pc_n steps)
  if (g > 0 && okprev != epoch) return; // the chain ended before this step
  Cand key{m_q ? m_sc : 0.0, 0.0, m_q, lane};
  key = wave_bcast_best<0>(key);
  const int q = key.idx, wl = key.aux;
  const bool none = (q == 0);
  const int left = (budget < 0) ? 1 : budget - used;
  if (none || left <= 0 || stall >= A.stall_limit) {
    if (lead && g == 0) {
      c->fstate = F_STOP;
      c->phase = PH_PRIMAL2;
      // the pivot limit with an entering column still on offer and nothing perturbed is what the generic step would
      // report as it stands (select_step: price, then `budget == 0` -> D_ITLIM)
      if (!none && left <= 0 && stall < A.stall_limit && !c->perturbed) {
        c->done = D_ITLIM;
        c->step = ST_NONE;
      }
    }
    return;
  }
  const int sdir = rl_i(m_sd, wl);
  PC_STAMP(1);
  // ---- what depends on the entering column: its entries of this lane's row and of the chain's scaled pivot rows
  double a = T[(size_t)ic * ld + q];
  const double sqv = A.srow0[(size_t)((lane < g) ? lane : 0) * A.sstride + q];
  rec.ip = xdiv(1.0, rec.piv);
  // the row side as of step g
  double cg1 = 0.0; // this lane's entry of step g-1's pivot column
#pragma unroll
  for (int s = 0; s < 16; s++) {
    if (s == g - 1) cg1 = ca[s];
    if (16 + s == g - 1) cg1 = cb[s];
  }
  if (g > 0) {
    if (pk == ST_FLIP) be = fma(cg1, pdelta, be);
    else if (i == pp_) {
      be = pxq - ps0;
      lb = pelb;
      ub = peub;
    } else
      be = fma(-cg1, ps0, be);
  }
  PC_STAMP(2);
  // column q carried through steps 0..g-1
#define PC_COLSTEP(S, CV)                                                          \
  if ((S) < g) {                                                                   \
    const int lk = rl_i(rec.kind, (S));                                            \
    if (lk == ST_PIVOT) {                                                          \
      const int lp = rl_i(rec.p, (S)), lq = rl_i(rec.q, (S));                      \
      const double sq = rl_d(sqv, (S));                                            \
      if (lq == q) a = (i == lp) ? rl_d(rec.ip, (S)) : xdiv((CV), rl_d(rec.piv, (S))); \
      else a = (i == lp) ? -sq : fma(-(CV), sq, a);                                \
    }                                                                              \
  }
#pragma unroll
  for (int s = 0; s < 16; s++) { PC_COLSTEP(s, ca[s]) }
  if (g > 16) {
#pragma unroll
    for (int s = 0; s < 16; s++) { PC_COLSTEP(16 + s, cb[s]) }
  }
#undef PC_COLSTEP
  PC_STAMP(3);
  Cand rb{0.0, 0.0, 0, 0};
  if (act) {
    Cand x{0.0, 0.0, 0, 0};
    if (ratio_row(a, sdir, be, lb, ub, 0, A.tol_piv, i, x)) rb = x;
  }
  int bw;
  rb = block_best1<1>(wave_bcast_best<1>(rb), s_slots, &bw);
  PC_STAMP(4);
  // ---- every store of the launch
  if (act) {
    A.colq0[(size_t)g * A.cstride + i] = a;
    if (g > 0) {
      A.betak[g & 1][i] = be;
      A.blbk[g & 1][i] = lb;
      A.bubk[g & 1][i] = ub;
    }
  }
  if (TIDX == 0) {
    double *rp = A.rp + b;
    const size_t st = A.rpstride;
    rp[RPF_K1 * st] = rb.k1;
    rp[RPF_K2 * st] = rb.k2;
    rp[RPF_IDX * st] = (double)rb.idx;
    rp[RPF_AUX * st] = (double)rb.aux;
  }
  if (lead) {
    c->hd_q[g] = q;
    c->hd_sdir[g] = sdir;
    c->hd_fq[g] = rl_i(m_fq, wl);
    c->hd_dq[g] = rl_d(m_dq, wl);
    c->hd_wq[g] = rl_d(m_wq, wl);
    c->hd_lbq[g] = rl_d(m_lb, wl);
    c->hd_ubq[g] = rl_d(m_ub, wl);
    c->ch_okc[g] = epoch;
    c->phase = PH_PRIMAL2;
  }
  PC_STAMP(5);
#undef PC_STAMP
}

// Row phase of step g.
__global__ __launch_bounds__(256) void k_pr(const ChainArgs A, int g) {
  __shared__ Cand s_slots[4];
  __shared__ double s_pay[4][5];
  Ctl *const c = A.c;
  const int b = (int)blockIdx.x;
  const bool lead = (b == 0 && TIDX == 0);
  const int lane = TIDX & 63;
  const int n = A.n, nrb = A.nrb;
  const size_t ld = (size_t)A.ld;
  const double *const T = A.T;
  unsigned long long *const dbg = c->dbg;
#define PR_STAMP(k) do { if (dbg && lead) dbg[(size_t)g * 16 + 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  PR_STAMP(0);
  // ---- entry
  const int done = c->done, fstate = c->fstate, epoch = c->pc_epoch;
  const int okc = c->ch_okc[g];
  const int stall = g ? c->ch_stall[g - 1] : c->stall;
  const int used = g ? c->ch_cnt[g - 1] : 0;
  const int q = c->hd_q[g], sdir = c->hd_sdir[g], fq = c->hd_fq[g];
  const double dq = c->hd_dq[g], wq = c->hd_wq[g], lbq = c->hd_lbq[g], ubq = c->hd_ubq[g];
  const int j = b * 256 + TIDX;
  const bool act = (j <= n);
  const int jc = act ? j : n;
  // the column side as of step g (set g & 1; the handle's own arrays while the chain has no step)
  const int xr = g & 1, xw = xr ^ 1;
  const double dj = (g == 0) ? T[jc] : A.drowk[xr][jc];
  const double wj = (g == 0) ? A.pw[0][jc] : A.pwk[xr][jc];
  int fj = (g == 0) ? A.nflag[jc] : A.nflagk[xr][jc];
  double lbj = (g == 0) ? A.nlb[jc] : A.nlbk[xr][jc], ubj = (g == 0) ? A.nub[jc] : A.nubk[xr][jc];
  if (!act || j == 0) fj = MVX_NS;
  // this lane's entries of the chain's scaled pivot rows (all KCH rows of that buffer exist: no guard)
  double ta[16], tb[16];
#pragma unroll
  for (int s = 0; s < 16; s++) ta[s] = A.srow0[(size_t)s * A.sstride + jc];
  if (g > 16) {
#pragma unroll
    for (int s = 0; s < 16; s++) tb[s] = A.srow0[(size_t)(16 + s) * A.sstride + jc];
  } else {
#pragma unroll
    for (int s = 0; s < 16; s++) tb[s] = 0.0;
  }
  LaneRec rec = lane_rec(c, g);
  const int rl = (lane < g) ? lane : (g > 0 ? g - 1 : 0);
  // ratio-test partials: every wave reduces them on its own
  Cand rc{0.0, 0.0, 0, 0};
  for (int t = lane; t < nrb; t += 64) {
    const double *rp = A.rp + t;
    const size_t st = A.rpstride;
    Cand x{rp[RPF_K1 * st], rp[RPF_K2 * st], (int)rp[RPF_IDX * st], (int)rp[RPF_AUX * st]};
    if (cand_better<1>(x, rc)) rc = x;
  }
  if (done != D_RUN || fstate != F_RUN || okc != epoch) return;
  rc = wave_bcast_best<1>(rc);
  const int p = rc.idx, p_up = rc.aux;
  const double tstep = rc.k1;
  bool flip = false;
  double tf = 0.0;
  if (lbq > -INFINITY && ubq < INFINITY && fq != MVX_NF) {
    tf = ubq - lbq;
    if (p == 0 || tf <= tstep) flip = true;
  }
  if (!flip && p == 0) { // unbounded ray: the generic path reports it
    if (lead && g == 0) c->fstate = F_STOP;
    return;
  }
  PR_STAMP(1);
  // ---- what depends on the leaving row
  const int pr = flip ? 1 : p;
  const double val0 = T[(size_t)pr * ld + jc];
  const double piv = A.colq0[(size_t)g * A.cstride + pr];
  const double bp = (g == 0) ? A.betab[pr] : A.betak[xr][pr];
  const double plb = (g == 0) ? A.blb[pr] : A.blbk[xr][pr], pub = (g == 0) ? A.bub[pr] : A.bubk[xr][pr];
  const double rcp = A.colq0[(size_t)rl * A.cstride + pr]; // step l's pivot-column entry of the leaving row
  rec.ip = xdiv(1.0, rec.piv);
  const double rcd = xdiv(rcp, rec.piv);
  PR_STAMP(2);
  double sj = 0.0, dnew, wnew;
  int fnew, kind, lf = 0, stall_new;
  double delta = 0.0, bound = 0.0, s0 = 0.0, xq = 0.0;
  if (flip) {
    const int nf = (sdir > 0) ? MVX_NU : MVX_NL;
    delta = (sdir > 0) ? tf : -tf;
    kind = ST_FLIP;
    lf = nf;
    dnew = (j == 0) ? fma(dq, delta, dj) : dj; // the objective value moves with the flipped variable
    wnew = wj;
    fnew = (j == q) ? nf : fj;
    stall_new = 0;
  } else {
    // the leaving row's entry in this lane's column, carried through the chain so far
    double val = val0;
#define PR_ROWSTEP(S, TV)                                                                              \
  if ((S) < g) {                                                                                       \
    const int lk = rl_i(rec.kind, (S));                                                                \
    if (lk == ST_PIVOT) {                                                                              \
      const int lp = rl_i(rec.p, (S)), lq = rl_i(rec.q, (S));                                          \
      if (lp == p) val = (j == lq) ? rl_d(rec.ip, (S)) : -(TV);                                        \
      else val = (j == lq) ? rl_d(rcd, (S)) : fma(-rl_d(rcp, (S)), (TV), val);                         \
    }                                                                                                  \
  }
#pragma unroll
    for (int s = 0; s < 16; s++) { PR_ROWSTEP(s, ta[s]) }
    if (g > 16) {
#pragma unroll
      for (int s = 0; s < 16; s++) { PR_ROWSTEP(16 + s, tb[s]) }
    }
#undef PR_ROWSTEP
    bound = p_up ? pub : plb;
    lf = dev_leave_flag(plb, pub, p_up);
    s0 = xdiv(bp - bound, piv); // what lane 0 of block 0 gets for column 0
    sj = (j == 0) ? s0 : xdiv(val, piv);
    dnew = (j == q) ? xdiv(dq, piv) : fma(-dq, sj, dj);
    if (j == q) {
      const double cc = xdiv(wq, piv * piv);
      wnew = cc > 1.0 ? cc : 1.0;
      lbj = plb; // the leaving variable comes to sit in column q
      ubj = pub;
    } else {
      const double cc = sj * sj * wq;
      wnew = cc > wj ? cc : wj;
    }
    fnew = (j == q) ? lf : fj;
    kind = ST_PIVOT;
    xq = dev_nb_value(fq, lbq, ubq);
    stall_new = (tstep <= DEGEN_TOL) ? stall + 1 : 0;
  }
  if (!act || j == 0) fnew = MVX_NS;
  PR_STAMP(3);
  // pricing partial for step g+1
  Cand best{0.0, 0.0, 0, 0};
  if (fnew != MVX_NS) {
    Cand x{0.0, 0.0, 0, 0};
    if (price_col(fnew, A.sgn * dnew, A.tol_dj, j, wnew, x)) best = x;
  }
  publish_pricing(A, best, b, dnew, wnew, lbj, ubj, fnew, s_slots, s_pay);
  PR_STAMP(4);
  // ---- every store of the launch
  if (act) {
    A.srow0[(size_t)g * A.sstride + j] = sj;
    A.drowk[xw][j] = dnew;
    A.pwk[xw][j] = (j >= 1) ? wnew : 1.0;
    A.nflagk[xw][j] = (j >= 1) ? fnew : MVX_NS;
    A.nlbk[xw][j] = lbj;
    A.nubk[xw][j] = ubj;
  }
  if (lead) {
    c->ch_kind[g] = kind;
    c->ch_p[g] = flip ? 0 : p;
    c->ch_q[g] = q;
    c->ch_lf[g] = lf;
    c->ch_piv[g] = flip ? 1.0 : piv;
    c->ch_xq[g] = xq;
    c->ch_s0[g] = s0;
    c->ch_delta[g] = delta;
    c->ch_elb[g] = lbq;
    c->ch_eub[g] = ubq;
    c->ch_llb[g] = plb;
    c->ch_lub[g] = pub;
    c->ch_bound[g] = bound;
    c->ch_pup[g] = p_up;
    c->ch_stall[g] = stall_new;
    c->ch_cnt[g] = used + (flip ? 0 : 1);
    c->ch_ok[g] = epoch;
    c->pc_n = g + 1;
  }
  PR_STAMP(5);
#undef PR_STAMP
}

// ================================================================= cluster selection (k_chain)
// The whole selection of a chain in ONE launch.  k_pc / k_pr pay two launch boundaries per step, and behind each
// boundary every first load goes to memory; here up to 32 workgroups stay resident for the chain, each owning 256 rows
// and 256 columns (x RPT / CPT), and what a step changes besides the tableau lives in their registers: reduced costs,
// devex weights, statuses and bounds of the columns; values and bounds of the rows.  The pivot columns / scaled pivot
// rows of the earlier steps, which the carries need entry by entry, sit in LDS (one slot per thread and step).  A step
// is two phases, each closed by an all-to-all exchange among the workgroups:
//   column phase  gather column q (strided, from the tableau as it stands), carry it through the chain, ratio test
//                 -> exchange: leaving row p with its pivot element, value and bounds
//   row phase     load row p, carry it through the chain -> scaled pivot row; objective row, weights, statuses; pricing
//                 -> exchange: entering column q of the NEXT step with its reduced cost, weight, bounds and its entry in
//                 the scaled row just made
// The exchange: every workgroup reduces its candidates (one barrier), publishes ONE record of eight self-tagged 16-byte
// fields {lo, tag, hi, tag} with a single store instruction, and every wave sweeps all records (field-major: one
// 16-byte load per field, lane = workgroup) until every tag is the exchange's own, then reduces them with shuffles:
// no flag, no fence, no atomic; every reduction key is a strict total order, so all waves agree.  Placement: the launch
// has 8 x NW workgroups and only those with blockIdx % 8 == 0 take part -- under the round-robin dispatch of this
// device they share one XCD, whose L2 then serves the exchange (sc0 stores stay in that L2, sc1 loads bypass the L1:
// 1.7 us per exchange against 3.3 us across the XCDs, scripts/probe/cluster_probe.hip).  Correctness does not rest on
// that placement or on the workgroups being co-resident: a record that never shows up ends in a bounded wait, the
// abort flag, and a launch that has changed nothing (the chain is recorded by workgroup 0 only at the very end); the
// host then falls back on k_pc / k_pr.  What other workgroups read of a workgroup's stores (pivot columns, scaled rows
// of earlier steps) they read with sc1 loads at least one full exchange after the store was issued in front of a sweep:
// the sweep's loads are behind it in the wave's memory queue, so it has reached the L2 by then.
constexpr int XNW = 64; // records per field row of the exchange area
constexpr int XNF = 8;  // fields per record
enum : int { XAUX_SC0 = 1, XAUX_SC1 = 16 };
typedef unsigned xu4 __attribute__((ext_vector_type(4)));

struct XCtx {
  __amdgpu_buffer_rsrc_t rs;
  int w, nw;
  unsigned tag; // tag of the next exchange
  int xno;      // exchanges so far
  int *xabort;
  bool dead;
  bool flag; // in: this thread's contribution to the OR the next exchange carries; out: the OR over the whole cluster
};

__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// One exchange: `mine` is this thread's candidate with NP payload doubles; returns the winner among all workgroups and
// its payload, the same in every thread of every workgroup.  One barrier.
template <int MODE, int NP>
__device__ __forceinline__ Cand xchg(XCtx &X, Cand mine, const double (&pin)[NP], double (&pout)[NP], Cand (*s_slots)[4], double (*s_pay)[4][5]) {
  const int lane = TIDX & 63, wave = TIDX >> 6, b = X.xno & 1;
  // workgroup level
  int ol;
  Cand wb = wave_argbest<MODE>(mine, &ol);
  wb.aux = (wb.aux & 0xffff) | (__ballot(X.flag) ? 0x10000 : 0); // the flag rides in the record's aux word
  if (lane == 0) s_slots[b][wave] = wb;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    const double v = rl_d(pin[k], ol);
    if (lane == 0) s_pay[b][wave][k] = v;
  }
  __syncthreads();
  Cand bb = s_slots[b][0];
  int bw = 0, fl = bb.aux & 0x10000;
#pragma unroll
  for (int k = 1; k < 4; k++) {
    const Cand y = s_slots[b][k];
    fl |= y.aux & 0x10000;
    if (cand_better<MODE>(y, bb)) {
      bb = y;
      bw = k;
    }
  }
  bb.aux = (bb.aux & 0xffff) | fl;
  const unsigned tag = X.tag;
  const unsigned reg = (unsigned)(X.xno & 3) * (unsigned)(XNF * XNW * 16);
  if (TIDX < 3 + NP) {
    double v;
    if (TIDX == 0) v = bb.k1;
    else if (TIDX == 1) v = bb.k2;
    else if (TIDX == 2) v = __hiloint2double(bb.aux, bb.idx);
    else v = s_pay[b][bw][TIDX - 3];
    const xu4 x = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
    __builtin_amdgcn_raw_buffer_store_b128(x, X.rs, reg + (unsigned)(TIDX * XNW + X.w) * 16, 0, XAUX_SC0);
  }
  // every wave sweeps: lane r < nw takes workgroup r's record
  double rec[3 + NP];
  const int lr = lane < X.nw ? lane : 0;
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int f = 0; f < 3 + NP; f++) {
      const xu4 x = __builtin_amdgcn_raw_buffer_load_b128(X.rs, reg + (unsigned)(f * XNW + lr) * 16, 0, XAUX_SC1);
      ok &= (x.y == tag) & (x.w == tag);
      rec[f] = __hiloint2double((int)x.z, (int)x.x);
    }
    if (__all(ok)) break;
    ++spins;
    if ((spins & 255u) == 0) {
      if (spins > 2000000u) __hip_atomic_store(X.xabort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (spins > 2000000u || __hip_atomic_load(X.xabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        X.dead = true;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  X.tag++;
  X.xno++;
  Cand rc{0.0, 0.0, 0, 0};
  if (lane < X.nw && !X.dead) rc = Cand{rec[0], rec[1], __double2loint(rec[2]), __double2hiint(rec[2])};
  int wl;
  Cand win = wave_argbest<MODE>(rc, &wl);
  X.flag = __ballot((rc.aux & 0x10000) != 0) != 0ull;
  win.aux = (int)(short)(win.aux & 0xffff);
#pragma unroll
  for (int k = 0; k < NP; k++) pout[k] = rl_d(rec[3 + k], wl);
  return win;
}

template <int CPT, int RPT>
__global__ __launch_bounds__(256) void k_chain(const ChainArgs A) {
  extern __shared__ double hist[]; // [kmax][CPT + RPT][256]: scaled-row entries of this thread's columns, pivot-column entries of its rows
  __shared__ Cand s_slots[2][4];
  __shared__ double s_pay[2][4][5];
  // the chain so far (every thread of the workgroup writes the same values) and, per wave, the operands of the carries
  __shared__ int s_kind[KCH], s_p[KCH], s_q[KCH];
  __shared__ double s_piv[KCH], s_ip[KCH], s_wv[4][2][KCH];
  if (blockIdx.x & 7) return; // the workgroups that share an XCD with workgroup 0
  const int w = (int)blockIdx.x >> 3, t = TIDX, lane = t & 63, wave = t >> 6;
  Ctl *const c = A.c;
  const int nw = A.nw, TT = nw * 256, gt = w * 256 + t;
  const int m = A.m, n = A.n, kmax = A.kmax;
  const size_t ld = (size_t)A.ld;
  const double *const T = A.T;
  const bool lead = (gt == 0);
  unsigned long long *const dbg = c->dbg;
#define CH_STAMP(K) do { if (dbg && lead) dbg[(size_t)g * 16 + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  // ---- entry
  const int done = c->done, fstate0 = c->fstate, budget = c->budget, perturbed = c->perturbed, aborted = c->cl_abort, epoch = c->pc_epoch;
  const int phase0 = c->phase, n_edits = c->n_edits, cur0 = c->curA & 1;
  int stall = c->stall;
  // The first chain launch of a batch decides what k_pboot decides for k_pc / k_pr: a call that has just begun (phase
  // PH_START, no bound edits waiting) is taken from its first pivot if no basic variable lies outside its bounds by more
  // than the tolerance -- every thread looks at its own rows, the verdict rides on the first exchange -- and then the
  // devex weights restart from one, as the generic step's `fresh_primal` does; a run already in primal phase 2 carries
  // on; anything else (dual simplex, phase 1, Bland's rule in force) turns the pipeline off.
  const bool boot = (A.boot != 0);
  const bool fresh = boot && (phase0 == PH_START);
  const int fstate = boot ? F_RUN : fstate0;
  const bool boot_off = boot && (stall >= A.stall_limit || !(phase0 == PH_PRIMAL2 || (fresh && n_edits == 0)));
  double d[CPT], wgt[CPT], lbj[CPT], ubj[CPT], sqn[CPT];
  int f[CPT], jj[CPT];
#pragma unroll
  for (int u = 0; u < CPT; u++) {
    const int j = 1 + gt + u * TT;
    const bool act = (j <= n);
    const int jc = act ? j : n;
    jj[u] = j;
    d[u] = T[jc];
    wgt[u] = fresh ? 1.0 : A.pw[boot ? cur0 : 0][jc];
    f[u] = act ? A.nflag[jc] : MVX_NS;
    lbj[u] = A.nlb[jc];
    ubj[u] = A.nub[jc];
    sqn[u] = 0.0;
  }
  double z = T[0];
  double be[RPT], lb[RPT], ub[RPT];
  int ii[RPT];
#pragma unroll
  for (int u = 0; u < RPT; u++) {
    const int i = 1 + gt + u * TT;
    const int ic = (i <= m) ? i : m;
    ii[u] = i;
    be[u] = boot ? T[(size_t)ic * ld] : A.betab[ic];
    lb[u] = A.blb[ic];
    ub[u] = A.bub[ic];
  }
  if (done != D_RUN || fstate != F_RUN || aborted) return;
  if (boot_off) {
    if (lead) c->fstate = F_OFF;
    return;
  }
  XCtx X;
  X.rs = __builtin_amdgcn_make_buffer_rsrc(A.xg, 0, A.xg_bytes, 0x00027000);
  X.w = w;
  X.nw = nw;
  X.tag = A.tagbase + 1;
  X.xno = 0;
  X.xabort = A.xabort;
  X.dead = false;
  X.flag = false;
  if (fresh) { // is the start primal feasible?
    const double tolb = A.tol_bnd;
#pragma unroll
    for (int u = 0; u < RPT; u++) {
      if (ii[u] <= m) {
        if (lb[u] > -INFINITY && be[u] < lb[u] - tolb * (1.0 + fabs(lb[u]))) X.flag = true;
        if (ub[u] < INFINITY && be[u] > ub[u] + tolb * (1.0 + fabs(ub[u]))) X.flag = true;
      }
    }
  }
#define HS(S, U) hist[((size_t)(S) * (CPT + RPT) + (U)) * 256 + t]
#define HC(S, U) hist[((size_t)(S) * (CPT + RPT) + CPT + (U)) * 256 + t]
  int g = 0, used = 0;
  // pricing of this thread's columns -> candidate with payload {d, w, lb, ub, entry in the newest scaled row}
  Cand pc;
  double pp[5], po[5];
#define PRICE()                                                                                          \
  do {                                                                                                   \
    pc = Cand{0.0, 0.0, 0, 0};                                                                           \
    pp[0] = d[0]; pp[1] = wgt[0]; pp[2] = lbj[0]; pp[3] = ubj[0]; pp[4] = sqn[0];                        \
    _Pragma("unroll") for (int u = 0; u < CPT; u++) {                                                    \
      Cand x{0.0, 0.0, 0, 0};                                                                            \
      if (f[u] != MVX_NS && price_col(f[u], A.sgn * d[u], A.tol_dj, jj[u], wgt[u], x) && cand_better<0>(x, pc)) { \
        pc = x;                                                                                          \
        pc.aux = (x.aux > 0 ? 1 : 0) | (f[u] << 1);                                                      \
        pp[0] = d[u]; pp[1] = wgt[u]; pp[2] = lbj[u]; pp[3] = ubj[u]; pp[4] = sqn[u];                    \
      }                                                                                                  \
    }                                                                                                    \
  } while (0)
  PRICE();
  Cand qc = xchg<0, 5>(X, pc, pp, po, s_slots, s_pay);
  if (boot && !X.dead) {
    if (X.flag) { // not primal feasible: the generic step decides between the dual simplex and phase 1
      if (lead) c->fstate = F_OFF;
      return;
    }
    if (lead) {
      c->fstate = F_RUN;
      c->step = ST_NONE;
    }
  }
  X.flag = false;
  bool end_itlim = false;
  for (;;) {
    if (X.dead) break;
    const int q = qc.idx;
    const bool none = (q == 0);
    const int left = (budget < 0) ? 1 : budget - used;
    if (none || left <= 0 || stall >= A.stall_limit) {
      // the pivot limit with an entering column still on offer and nothing perturbed is what the generic step would
      // report as it stands (select_step: price, then `budget == 0` -> D_ITLIM)
      const bool itlim = (!none && left <= 0 && stall < A.stall_limit && !perturbed);
      if (g == 0) {
        if (lead) {
          c->fstate = F_STOP;
          c->phase = PH_PRIMAL2;
          if (itlim) {
            c->done = D_ITLIM;
            c->step = ST_NONE;
          }
        }
      } else
        end_itlim = itlim; // once the chain is applied: the generic step that closes the batch says so (pc_itlim)
      break;
    }
    if (g >= kmax) break;
    CH_STAMP(0);
    const int sdir = (qc.aux & 1) ? 1 : -1, fq = qc.aux >> 1;
    const double dq = po[0], wq = po[1], lbq = po[2], ubq = po[3];
    // ---- column phase: column q of this thread's rows, the entries of the earlier scaled rows in column q
    double a[RPT];
#pragma unroll
    for (int u = 0; u < RPT; u++) a[u] = T[(size_t)((ii[u] <= m) ? ii[u] : m) * ld + q];
    {
      double sqv = po[4]; // lane g-1: from the exchange; lanes below: from memory
      if (lane < g - 1) sqv = ld_sc1(A.srow0 + (size_t)lane * A.sstride + q);
      if (lane < g) s_wv[wave][0][lane] = sqv;
    }
    // the carry, eight steps to a block: the operands of a block are requested together (LDS), then the block's
    // dependent multiply-adds run; which steps are pivots and which of them are the rare case -- the column that
    // entered at step s enters again, c_i / piv of that step -- is known to the scalar unit beforehand
    const int gu = __builtin_amdgcn_readfirstlane(g);
    const bool ispc = (lane < gu) && (s_kind[lane < KCH ? lane : 0] == ST_PIVOT);
    const unsigned pivm = (unsigned)__ballot(ispc);
    const unsigned spc = (unsigned)__ballot(ispc && s_q[lane < KCH ? lane : 0] == q);
    // (each SIMD runs one wave of this kernel, so every branch and every wait is paid in full: a block whose steps are
    // all ordinary pivots -- nearly every block -- runs without a single branch; blocks of 8, 4, 2, 1 steps)
    {
      int s0 = 0;
#define COL_BLOCK(N)                                                                                                     \
  if (gu - s0 >= N) {                                                                                                    \
    const unsigned bm = ((1u << N) - 1u) << s0;                                                                          \
    double hc8[RPT][N], sq8[N];                                                                                          \
    int lp8[N];                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < N; k++) {                                                                      \
      sq8[k] = s_wv[wave][0][s0 + k];                                                                                    \
      lp8[k] = s_p[s0 + k];                                                                                              \
      _Pragma("unroll") for (int u = 0; u < RPT; u++) hc8[u][k] = HC(s0 + k, u);                                         \
    }                                                                                                                    \
    if ((pivm & bm) == bm && (spc & bm) == 0u) {                                                                         \
      _Pragma("unroll") for (int k = 0; k < N; k++) {                                                                    \
        _Pragma("unroll") for (int u = 0; u < RPT; u++) a[u] = (ii[u] == lp8[k]) ? -sq8[k] : fma(-hc8[u][k], sq8[k], a[u]); \
      }                                                                                                                  \
    } else {                                                                                                             \
      _Pragma("unroll") for (int k = 0; k < N; k++) {                                                                    \
        const int sx = s0 + k;                                                                                           \
        if ((pivm >> sx) & 1u) {                                                                                         \
          if ((spc >> sx) & 1u) {                                                                                        \
            const double ips = s_ip[sx], pvs = s_piv[sx];                                                                \
            _Pragma("unroll") for (int u = 0; u < RPT; u++) a[u] = (ii[u] == lp8[k]) ? ips : xdiv(hc8[u][k], pvs);       \
          } else {                                                                                                       \
            _Pragma("unroll") for (int u = 0; u < RPT; u++) a[u] = (ii[u] == lp8[k]) ? -sq8[k] : fma(-hc8[u][k], sq8[k], a[u]); \
          }                                                                                                              \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
    s0 += N;                                                                                                             \
  }
      while (gu - s0 >= 8) { COL_BLOCK(8) }
      COL_BLOCK(4)
      COL_BLOCK(2)
      COL_BLOCK(1)
#undef COL_BLOCK
    }
    CH_STAMP(1);
    Cand rb{0.0, 0.0, 0, 0};
    double rp[4] = {a[0], be[0], lb[0], ub[0]}, ro[4];
#pragma unroll
    for (int u = 0; u < RPT; u++) {
      Cand x{0.0, 0.0, 0, 0};
      if (ii[u] <= m) {
        A.colq0[(size_t)g * A.cstride + ii[u]] = a[u];
        if (ratio_row(a[u], sdir, be[u], lb[u], ub[u], 0, A.tol_piv, ii[u], x) && cand_better<1>(x, rb)) {
          rb = x;
          rp[0] = a[u]; rp[1] = be[u]; rp[2] = lb[u]; rp[3] = ub[u];
        }
      }
      HC(g, u) = a[u];
    }
    CH_STAMP(2);
    const Cand pw = xchg<1, 4>(X, rb, rp, ro, s_slots, s_pay);
    if (X.dead) break;
    CH_STAMP(3);
    const int p = pw.idx, p_up = pw.aux;
    const double tstep = pw.k1, piv = ro[0], bp = ro[1], plb = ro[2], pub = ro[3];
    // row p of this thread's columns and the entries of the earlier pivot columns in row p: requested as soon as p is
    // known, ahead of the arithmetic of the decision (a bound flip does not use them)
    double val[CPT];
#pragma unroll
    for (int u = 0; u < CPT; u++) val[u] = T[(size_t)p * ld + ((jj[u] <= n) ? jj[u] : n)];
    double rcp_l = 0.0;
    if (lane < g) rcp_l = ld_sc1(A.colq0 + (size_t)lane * A.cstride + p);
    bool flip = false;
    double tf = 0.0;
    if (lbq > -INFINITY && ubq < INFINITY && fq != MVX_NF) {
      tf = ubq - lbq;
      if (p == 0 || tf <= tstep) flip = true;
    }
    if (!flip && p == 0) { // unbounded ray: the generic path reports it
      if (lead && g == 0) {
        c->fstate = F_STOP;
        c->phase = PH_PRIMAL2;
      }
      break;
    }
    int kind, lf, stall_new;
    double delta = 0.0, bound = 0.0, s0 = 0.0, xq = 0.0, ip = 1.0;
    if (flip) {
      kind = ST_FLIP;
      lf = (sdir > 0) ? MVX_NU : MVX_NL;
      delta = (sdir > 0) ? tf : -tf;
      stall_new = 0;
#pragma unroll
      for (int u = 0; u < RPT; u++) be[u] = fma(a[u], delta, be[u]);
      z = fma(dq, delta, z); // the objective value moves with the flipped variable
#pragma unroll
      for (int u = 0; u < CPT; u++) {
        if (jj[u] == q) f[u] = lf;
        sqn[u] = 0.0;
      }
    } else {
      kind = ST_PIVOT;
      bound = p_up ? pub : plb;
      lf = dev_leave_flag(plb, pub, p_up);
      s0 = xdiv(bp - bound, piv);
      xq = dev_nb_value(fq, lbq, ubq);
      ip = xdiv(1.0, piv);
      stall_new = (tstep <= DEGEN_TOL) ? stall + 1 : 0;
#pragma unroll
      for (int u = 0; u < RPT; u++) {
        if (ii[u] == p) {
          be[u] = xq - s0;
          lb[u] = lbq; // the entering variable comes to sit in row p
          ub[u] = ubq;
        } else
          be[u] = fma(-a[u], s0, be[u]);
      }
      // ---- row phase
      CH_STAMP(4);
      if (lane < g) {
        const double rcp = rcp_l;
        s_wv[wave][0][lane] = rcp;
        s_wv[wave][1][lane] = xdiv(rcp, s_piv[lane]);
      }
      // the rare case here: the row that left at step s leaves again, -s_j of that step
      const unsigned spr = (unsigned)__ballot(ispc && s_p[lane < KCH ? lane : 0] == p);
      {
        int s0 = 0;
#define ROW_BLOCK(N)                                                                                                     \
  if (gu - s0 >= N) {                                                                                                    \
    const unsigned bm = ((1u << N) - 1u) << s0;                                                                          \
    double hs8[CPT][N], cp8[N], cd8[N];                                                                                  \
    int lq8[N];                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < N; k++) {                                                                      \
      cp8[k] = s_wv[wave][0][s0 + k];                                                                                    \
      cd8[k] = s_wv[wave][1][s0 + k];                                                                                    \
      lq8[k] = s_q[s0 + k];                                                                                              \
      _Pragma("unroll") for (int u = 0; u < CPT; u++) hs8[u][k] = HS(s0 + k, u);                                         \
    }                                                                                                                    \
    if ((pivm & bm) == bm && (spr & bm) == 0u) {                                                                         \
      _Pragma("unroll") for (int k = 0; k < N; k++) {                                                                    \
        _Pragma("unroll") for (int u = 0; u < CPT; u++) val[u] = (jj[u] == lq8[k]) ? cd8[k] : fma(-cp8[k], hs8[u][k], val[u]); \
      }                                                                                                                  \
    } else {                                                                                                             \
      _Pragma("unroll") for (int k = 0; k < N; k++) {                                                                    \
        const int sx = s0 + k;                                                                                           \
        if ((pivm >> sx) & 1u) {                                                                                         \
          if ((spr >> sx) & 1u) {                                                                                        \
            const double ips = s_ip[sx];                                                                                 \
            _Pragma("unroll") for (int u = 0; u < CPT; u++) val[u] = (jj[u] == lq8[k]) ? ips : -hs8[u][k];               \
          } else {                                                                                                       \
            _Pragma("unroll") for (int u = 0; u < CPT; u++) val[u] = (jj[u] == lq8[k]) ? cd8[k] : fma(-cp8[k], hs8[u][k], val[u]); \
          }                                                                                                              \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
    s0 += N;                                                                                                             \
  }
        while (gu - s0 >= 8) { ROW_BLOCK(8) }
        ROW_BLOCK(4)
        ROW_BLOCK(2)
        ROW_BLOCK(1)
#undef ROW_BLOCK
      }
      CH_STAMP(5);
#pragma unroll
      for (int u = 0; u < CPT; u++) {
        const int j = jj[u];
        const double sj = xdiv(val[u], piv);
        if (j == q) {
          d[u] = xdiv(dq, piv);
          const double cc = xdiv(wq, piv * piv);
          wgt[u] = cc > 1.0 ? cc : 1.0;
          lbj[u] = plb; // the leaving variable comes to sit in column q
          ubj[u] = pub;
          f[u] = lf;
        } else {
          d[u] = fma(-dq, sj, d[u]);
          const double cc = sj * sj * wq;
          wgt[u] = cc > wgt[u] ? cc : wgt[u];
        }
        if (j <= n) A.srow0[(size_t)g * A.sstride + j] = sj;
        HS(g, u) = sj;
        sqn[u] = sj;
      }
      z = fma(-dq, s0, z);
      if (lead) A.srow0[(size_t)g * A.sstride] = s0;
    }
    // ---- the step is chosen: every wave notes it in lane g, workgroup 0 writes it down for the bulk pass
    s_kind[g] = kind; // read from the next step on, behind the barrier of the exchange that follows
    s_p[g] = flip ? 0 : p;
    s_q[g] = q;
    s_piv[g] = flip ? 1.0 : piv;
    s_ip[g] = ip;
    used += flip ? 0 : 1;
    if (lead) {
      c->ch_kind[g] = kind;
      c->ch_p[g] = flip ? 0 : p;
      c->ch_q[g] = q;
      c->ch_lf[g] = lf;
      c->ch_piv[g] = flip ? 1.0 : piv;
      c->ch_xq[g] = xq;
      c->ch_s0[g] = s0;
      c->ch_delta[g] = delta;
      c->ch_elb[g] = lbq;
      c->ch_eub[g] = ubq;
      c->ch_llb[g] = plb;
      c->ch_lub[g] = pub;
      c->ch_bound[g] = bound;
      c->ch_pup[g] = p_up;
      c->ch_stall[g] = stall_new;
      c->ch_cnt[g] = used;
      c->ch_ok[g] = epoch;
    }
    CH_STAMP(6);
    PRICE();
    CH_STAMP(7);
    stall = stall_new;
    g++;
    qc = xchg<0, 5>(X, pc, pp, po, s_slots, s_pay);
    if (dbg && lead) dbg[(size_t)(g - 1) * 16 + 8] = __builtin_amdgcn_s_memrealtime();
  }
#undef PRICE
#undef CH_STAMP
#undef HS
#undef HC
  if (X.dead) { // a peer never showed up: nothing has been changed, the host falls back on k_pc / k_pr
    if (lead) {
      c->pc_n = 0;
      c->cl_abort = 1;
    }
    return;
  }
  // ---- the chain is complete: the column side as the last step leaves it, where the bulk pass looks for it
  if (g == 0 && fresh) { // no step (optimum, pivot limit, unbounded ray): nothing will write the restarted weights back,
                         // and the generic step that settles the matter prices with them
#pragma unroll
    for (int u = 0; u < CPT; u++) {
      if (jj[u] <= n) {
        A.pw[0][jj[u]] = 1.0;
        A.pw[1][jj[u]] = 1.0;
      }
    }
  }
  if (g > 0) {
    const int xf = g & 1;
#pragma unroll
    for (int u = 0; u < CPT; u++) {
      if (jj[u] <= n) {
        A.drowk[xf][jj[u]] = d[u];
        A.pwk[xf][jj[u]] = wgt[u];
      }
    }
    if (lead) {
      A.drowk[xf][0] = z;
      A.pwk[xf][0] = 1.0;
    }
  }
  if (lead) {
    c->pc_n = g;
    if (g > 0) {
      c->phase = PH_PRIMAL2;
      c->pc_itlim = end_itlim ? 1 : 0;
    }
  }
}

// The chain's bookkeeping (one workgroup, every thread calls): the basis swaps of the chain in order -- a swap exchanges
// (variable, bounds) between row p_t and column q_t.  The bounds need no exchange: the selection recorded what enters
// row p_t (ch_elb / ch_eub) and what comes to sit in column q_t (ch_llb / ch_lub), so the last step that touches a row
// or a column writes them; the variable numbers go through the swaps in LDS (one lane, no memory round trip per step),
// lane t holding step t.  Nothing the bulk pass or k_fpatch reads is touched (they read the chain's record and pc_n).
__device__ void chain_commit(const ChainArgs &A, Ctl *c, int nch) {
  __shared__ int s_cp[KCH], s_cq[KCH], s_ck[KCH], s_rf[KCH], s_cf[KCH], s_R[KCH], s_C[KCH];
  const int t = TIDX;
  const bool on = (t < nch);
  int kind = 0, p = 0, q = 0;
  if (on) {
    kind = c->ch_kind[t];
    p = c->ch_p[t];
    q = c->ch_q[t];
    s_ck[t] = kind;
    s_cp[t] = p;
    s_cq[t] = q;
    s_R[t] = (kind == ST_PIVOT) ? c->bvar[p] : 0;
    s_C[t] = (kind == ST_PIVOT) ? c->nvar[q] : 0;
  }
  __syncthreads();
  bool lastrow = true, lastcol = true, lastflag = true;
  if (on) {
    int rf = t, cf = t;
    for (int u = t - 1; u >= 0; u--) {
      if (s_ck[u] != ST_PIVOT) continue;
      if (s_cp[u] == p) rf = u;
      if (s_cq[u] == q) cf = u;
    }
    s_rf[t] = rf;
    s_cf[t] = cf;
    for (int u = t + 1; u < nch; u++) {
      if (s_cq[u] == q) lastflag = false;
      if (s_ck[u] != ST_PIVOT) continue;
      if (s_cp[u] == p) lastrow = false;
      if (s_cq[u] == q) lastcol = false;
    }
  }
  __syncthreads();
  if (t == 0) {
    for (int u = 0; u < nch; u++) {
      if (s_ck[u] != ST_PIVOT) continue;
      const int a = s_rf[u], b = s_cf[u];
      const int kv = s_R[a];
      s_R[a] = s_C[b];
      s_C[b] = kv;
    }
  }
  __syncthreads();
  if (on) {
    if (kind == ST_PIVOT) {
      if (lastrow) {
        c->bvar[p] = s_R[s_rf[t]];
        A.blb[p] = c->ch_elb[t];
        A.bub[p] = c->ch_eub[t];
      }
      if (lastcol) {
        c->nvar[q] = s_C[s_cf[t]];
        A.nlb[q] = c->ch_llb[t];
        A.nub[q] = c->ch_lub[t];
      }
    }
    if (lastflag) A.nflag[q] = c->ch_lf[t];
  }
  if (t == 0) {
    const int piv_n = c->ch_cnt[nch - 1];
    c->it_cnt += piv_n;
    c->n_flips += nch - piv_n;
    c->n_bulk++;
    if (c->budget > 0) c->budget -= piv_n;
    c->stall = c->ch_stall[nch - 1];
    c->pc_epoch++;
  }
}

// The bulk pass of a chain.  Every entry is loaded once, goes through the chain's steps in registers and is stored
// once.  What a step needs besides the entry -- the pivot-column entries of the tile's rows -- is the same for every
// lane: it comes in through scalar loads (constant address space: nothing in this launch writes it) and sits in SGPRs,
// so there is no LDS staging and no barrier.  The loop body is 2 TR multiply-adds per lane and nothing else: a bound
// flip runs them on zeros (fma(-0, 0, v) = v bit for bit), and the entries a step treats differently -- the pivot row
// (becomes -s_j), the pivot column (c_i / piv), column 0 under a bound flip -- are not this kernel's business: what they
// hold after their step does not depend on what they held before, so k_fpatch writes the chain's pivot rows, pivot
// columns and column 0 afterwards from the chain's own data.  (With the exceptions inside this loop the compiler kept
// two or three copies of the tile in registers and moved the tile between them every step: 210 VGPRs, 140 us for 16
// steps; profiles/r03_*.)  Column 0 is never stored here: k_fpatch needs it as it was.
typedef const double __attribute__((address_space(4))) *kconst_f64;
template <int TR, int NT>
__global__ __launch_bounds__(256) void k_fbc3(const ChainArgs A) {
  __shared__ double s_cq[KCH][TR]; // the tile's rows of every step's pivot column (zeros for a bound flip)
  Ctl *const c = A.c;
  if (c->done != D_RUN || c->fstate != F_RUN) return;
  const int nch = c->pc_n;
  if (nch == 0) return;
  const int n = A.n;
  const size_t ld = (size_t)A.ld;
  const int j0 = 2 * ((int)blockIdx.x * 256 + TIDX);
  const int i0 = 1 + (int)blockIdx.y * TR;
  const int xf = nch & 1; // the set the chain's last step wrote
  const bool act = (j0 <= n);
  const int jc = act ? j0 : 0;
  double *base = A.T + (size_t)i0 * ld + jc;
  double2 v[TR];
#pragma unroll
  for (int r = 0; r < TR; r++) v[r] = ld2<NT>(reinterpret_cast<const double2 *>(base + (size_t)r * ld));
  unsigned flips = 0; // which steps are bound flips: known before the loop, so no step waits for its own description
  for (int l = 0; l < nch; l++) flips |= (c->ch_kind[l] == ST_FLIP) ? (1u << l) : 0u;
  for (int e = TIDX; e < nch * TR; e += 256) {
    const int l = e / TR, r = e % TR;
    s_cq[l][r] = ((flips >> l) & 1u) ? 0.0 : A.colq0[(size_t)l * A.cstride + i0 + r];
  }
  __syncthreads();
  const double *sp = A.srow0 + jc;
  const double *zp = A.zeros + jc;
  // four steps to a group: the group's pairs of the scaled pivot rows are requested together, so that one memory round
  // trip is paid per group and not per step
  for (int l0 = 0; l0 < nch; l0 += 4) {
    double2 s4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int lk = (l0 + k < nch) ? l0 + k : nch - 1;
      s4[k] = *reinterpret_cast<const double2 *>(((flips >> lk) & 1u) ? zp : sp + (size_t)lk * A.sstride);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (l0 + k < nch) {
#pragma unroll
        for (int r = 0; r < TR; r++) {
          // v = fma(-c_i, s_j, v) in place (an instruction with a tied operand: left to itself the compiler gives each
          // step's results new registers and copies the whole tile back)
          const double ci = s_cq[l0 + k][r];
          asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(v[r].x) : "v"(ci), "v"(s4[k].x));
          asm("v_fma_f64 %0, -%1, %2, %0" : "+v"(v[r].y) : "v"(ci), "v"(s4[k].y));
        }
      }
    }
  }
  if (act) {
    if (j0 == 0) {
#pragma unroll
      for (int r = 0; r < TR; r++) base[(size_t)r * ld + 1] = v[r].y;
    } else {
#pragma unroll
      for (int r = 0; r < TR; r++) st2<NT>(reinterpret_cast<double2 *>(base + (size_t)r * ld), v[r]);
    }
    if (blockIdx.y == 0) { // the objective row and the weights live outside the row blocks: back to where the other paths read them
      *reinterpret_cast<double2 *>(A.T + j0) = *reinterpret_cast<const double2 *>(A.drowk[xf] + j0);
      const double2 w = *reinterpret_cast<const double2 *>(A.pwk[xf] + j0);
      *reinterpret_cast<double2 *>(A.pw[0] + j0) = w;
      *reinterpret_cast<double2 *>(A.pw[1] + j0) = w;
    }
  }
  // the first workgroup to be dispatched also commits the chain's bookkeeping, beside the rest of the pass
  if (blockIdx.x == 0 && blockIdx.y == 0) chain_commit(A, c, nch);
}

// After the bulk pass: the chain's pivot rows and pivot columns (what they hold after their own step does not depend
// on what they held before it: written from the chain's data and carried through the steps that follow), column 0 (from
// its old values, which the bulk pass leaves alone: bound flips move it) with its contiguous copy.  Jobs (blockIdx.y): 0 .. K-1 the pivot row of step y, K .. 2K-1
// the pivot column of step y - K, 2K column 0.  An entry that two jobs cover gets the same value from both.  The rules
// of the bulk update in full: the pivot row becomes -s_j (1/piv at the pivot, x_q - s_0 in column 0), the pivot column
// c_i / piv, a bound flip moves column 0 only, everything else fma(-c_i, s_j, v).  Every operand of every step is
// requested up front (the steps' descriptions and the operand that is the same for the whole job into LDS, the other
// one into registers): two memory round trips per launch, not two per step.
__global__ __launch_bounds__(256) void k_fpatch(const ChainArgs A, int kmax) {
  __shared__ int s_kind[KCH], s_p[KCH], s_q[KCH];
  __shared__ double s_piv[KCH], s_ip[KCH], s_xq[KCH], s_delta[KCH], s_u[KCH];
  Ctl *const c = A.c;
  if (c->done != D_RUN || c->fstate != F_RUN) return;
  const int nch = c->pc_n;
  if (nch == 0) return;
  const int m = A.m, n = A.n;
  const size_t ld = (size_t)A.ld;
  const int job = (int)blockIdx.y, x = (int)blockIdx.x * 256 + TIDX;
  // job kind 0: row p_l (x = column), 1: column q_l (x = row), 2: column 0 (x = row)
  const int jk = (job < kmax) ? 0 : (job < 2 * kmax) ? 1 : 2;
  const int l = (jk == 0) ? job : (jk == 1) ? job - kmax : -1;
  bool live = (jk == 2) || (l < nch && c->ch_kind[l < nch ? l : 0] == ST_PIVOT);
  const int span = (jk == 0) ? n : (jk == 1) ? m : A.mcap1 - 1;
  if ((int)blockIdx.x * 256 > span) live = false; // nothing of this job in this block
  if (live) {
    const int lp = (l >= 0) ? c->ch_p[l] : 0, lq = (l >= 0) ? c->ch_q[l] : 0;
    if (TIDX < KCH) {
      const int t = (TIDX < nch) ? TIDX : nch - 1;
      const int kind = c->ch_kind[t];
      const double piv = c->ch_piv[t];
      s_kind[TIDX] = kind;
      s_p[TIDX] = c->ch_p[t];
      s_q[TIDX] = c->ch_q[t];
      s_piv[TIDX] = piv;
      s_ip[TIDX] = (kind == ST_PIVOT) ? xdiv(1.0, piv) : 0.0;
      s_xq[TIDX] = c->ch_xq[t];
      s_delta[TIDX] = c->ch_delta[t];
      // the operand every entry of the job shares: row job c_t[p_l]; column job s_t[q_l]; column 0 s_t[0]
      s_u[TIDX] = (jk == 0) ? A.colq0[(size_t)t * A.cstride + lp] : A.srow0[(size_t)t * A.sstride + (jk == 1 ? lq : 0)];
    }
    const int xc = (x >= 1 && x <= span) ? x : 1;
    double op[KCH]; // the other operand, per entry: row job s_t[x]; column jobs c_t[x]
#pragma unroll
    for (int t = 0; t < KCH; t++) {
      const int tc = (t < nch) ? t : nch - 1;
      op[t] = (jk == 0) ? A.srow0[(size_t)tc * A.sstride + xc] : A.colq0[(size_t)tc * A.cstride + xc];
    }
    double *dst = (jk == 0) ? A.T + (size_t)lp * ld + xc : A.T + (size_t)xc * ld + (jk == 1 ? lq : 0);
    double val = (jk == 2) ? *dst : 0.0;
    __syncthreads();
    const int i = (jk == 0) ? lp : xc, j = (jk == 0) ? xc : (jk == 1) ? lq : 0;
    const int l0 = (jk == 2) ? 0 : l;
#pragma unroll
    for (int t = 0; t < KCH; t++) {
      if (t >= l0 && t < nch) {
        const double ci = (jk == 0) ? s_u[t] : op[t], sj = (jk == 0) ? op[t] : s_u[t];
        if (s_kind[t] == ST_FLIP) {
          if (j == 0) val = fma(ci, s_delta[t], val);
        } else if (i == s_p[t]) {
          val = (j == s_q[t]) ? s_ip[t] : (j == 0) ? s_xq[t] - sj : -sj;
        } else if (j == s_q[t]) {
          val = xdiv(ci, s_piv[t]);
        } else {
          val = fma(-ci, sj, val);
        }
      }
    }
    if (x >= 1 && x <= span) {
      if (jk != 2 || x <= m) *dst = val; // spare rows behind row m: their pivot-column entries are zero, the value stays
      if (jk == 2) A.betab[x] = val;
    }
  }
}

// ======================================================================= fused dual path
// The dual simplex (every warm start of a B&B child, bs.cpp:279,287) as the same two multi-workgroup kernels per pivot:
//   k_da  (one block per 256 columns): the entering column q is the reduction of the dual-ratio partials left by the
//         previous step; scales the pivot row into srow, updates the objective row, and -- before the bulk update runs
//         -- chooses the NEXT leaving row from column 0 alone (beta' = fma(-colq, s0, beta), weights updated from the
//         exported pivot column; every block does this O(m) pass redundantly: the keys are a total order), then
//         prices that row as it will be after the update (fma(-colq[p'], s_j, T[p'][j])) and leaves new partials.
//   k_fb<DUAL>: the streamed rank-1 update; the tile that owns the next entering column exports it contiguously.
// Same arithmetic per entry as k_select / k_update in the dual phase (and the oracle's dual_simplex).
__device__ __forceinline__ bool row_violation(double beta, double lb, double ub, double tol, double &viol, int &up) {
  viol = 0.0;
  up = 0;
  if (lb > -INFINITY && beta < lb - tol * (1.0 + fabs(lb))) viol = lb - beta;
  if (ub < INFINITY && beta > ub + tol * (1.0 + fabs(ub))) {
    viol = beta - ub;
    up = 1;
  }
  return viol > 0.0;
}

// one column's candidate in the dual ratio test of a row (dev_dual_ratio's body)
__device__ __forceinline__ bool dual_ratio_col(double a, double d, int f, int to_upper, double tp, int j, Cand &x) {
  if (f == MVX_NS) return false;
  const double aa = to_upper ? -a : a;
  double r;
  if (aa > tp && (f == MVX_NL || f == MVX_NF)) r = (f == MVX_NF) ? fabs(d) : (d < 0.0 ? -d : 0.0);
  else if (aa < -tp && (f == MVX_NU || f == MVX_NF)) r = (f == MVX_NF) ? fabs(d) : (d > 0.0 ? d : 0.0);
  else return false;
  const double mag = fabs(a);
  x = Cand{xdiv(r, mag), mag, j, 0};
  return true;
}

// bootstrap after a generic dual step: leaving row and dual-ratio partials from the tableau as it stands
__global__ __launch_bounds__(DA_THREADS) void k_dboot(Ctl *c) {
  __shared__ Cand lds[17];
  const bool lead = (blockIdx.x == 0 && TIDX == 0);
  if (c->done != D_RUN || c->phase != PH_DUAL || c->stall >= c->stall_limit || c->perturbed) {
    if (lead) c->fstate = F_OFF;
    return;
  }
  const int a = c->curA & 1, m = c->m, n = c->n;
  const size_t ld = (size_t)c->ld;
  const double *T = c->T, *w = c->dwx[a];
  const double tol = c->tol_bnd;
  Cand rb{0.0, 0.0, 0, 0};
  for (int i = 1 + TIDX; i <= m; i += DA_THREADS) {
    double viol;
    int up;
    if (row_violation(T[(size_t)i * ld], c->blb[i], c->bub[i], tol, viol, up)) {
      Cand x{xdiv(viol * viol, w[i]), 0.0, i, up};
      if (cand_better<0>(x, rb)) rb = x;
    }
  }
  rb = block_best<0>(rb, lds);
  const int p2 = rb.idx, p2_up = rb.aux;
  const int j = (int)blockIdx.x * DA_THREADS + TIDX;
  Cand best{0.0, 0.0, 0, 0};
  if (p2 && j >= 1 && j <= n) {
    Cand x{0.0, 0.0, 0, 0};
    if (dual_ratio_col(T[(size_t)p2 * ld + j], c->sgn * T[j], c->nflag[j], p2_up, c->tol_piv, j, x)) best = x;
  }
  __syncthreads();
  best = block_best<1>(best, lds);
  if (TIDX == 0) c->pp[a][blockIdx.x] = best;
  if (lead) {
    c->p_nextx[a] = p2;
    c->p_up_nextx[a] = p2_up;
    c->fstate = F_RUN_DUAL;
    c->step = ST_NONE;
    c->curB = a ^ 1;
    c->npbd = (int)gridDim.x;
  }
}

__global__ __launch_bounds__(DA_THREADS) void k_da(Ctl *c) {
  __shared__ Cand lds[17];
  // level 1
  const int done = c->done, fstate = c->fstate, cur = c->curA, budget = c->budget;
  const int stall = c->stall, stall_limit = c->stall_limit;
  const int npb = c->npbd, m = c->m, n = c->n;
  const size_t ld = (size_t)c->ld;
  double *const T = c->T;
  double *const srow = c->srow;
  const int *const nflag = c->nflag;
  const double *const colq = c->colqx[cur];
  const double *const betac = c->betac[cur];
  const Cand *const pp = c->pp[cur];
  Cand *const ppn = c->pp[cur ^ 1];
  const double *const dwc = c->dwx[cur & 1];
  double *const dwn = c->dwx[(cur & 1) ^ 1];
  const double sgn = c->sgn, tol_bnd = c->tol_bnd, tol_piv = c->tol_piv;
  const int p = c->p_nextx[cur & 1], p_up = c->p_up_nextx[cur & 1];
  if (done != D_RUN || fstate != F_RUN_DUAL) return;
  const bool lead = (blockIdx.x == 0 && TIDX == 0);
  const int lane = TIDX & 63;
  const int j = (int)blockIdx.x * DA_THREADS + TIDX;
  const bool act = (j <= n);
  // level 2: the entering column (smallest dual ratio), this lane's own entries
  Cand pc = (lane < npb) ? pp[lane] : Cand{0.0, 0.0, 0, 0};
  const double dold = act ? T[j] : 0.0;
  const int fj = (act && j >= 1) ? nflag[j] : MVX_NS;
  for (int k = lane + 64; k < npb; k += 64) {
    Cand x = pp[k];
    if (cand_better<1>(x, pc)) pc = x;
  }
  pc = wave_bcast_best<1>(pc);
  // no leaving row (primal feasible: the phase ends), no entering column (dual unbounded), no budget, or a stalled run
  // that Bland's rule must take over: the generic path continues
  if (p == 0 || pc.idx == 0 || budget == 0 || stall >= stall_limit) {
    if (lead) c->fstate = F_STOP;
    return;
  }
  const int q = pc.idx;
  // level 3
  const double piv = colq[p], dq = colq[0], wp = dwc[p];
  const double plb = c->blb[p], pub = c->bub[p];
  const double lbq = c->nlb[q], ubq = c->nub[q];
  const int fq = nflag[q];
  const double v = act ? T[(size_t)p * ld + j] : 0.0;
  const double bound = p_up ? pub : plb;
  const int lf = dev_leave_flag(plb, pub, p_up);
  const double xq = dev_nb_value(fq, lbq, ubq);
  const double s0 = xdiv(betac[p] - bound, piv);
  double sj = 0.0, dnew = 0.0;
  if (act) {
    sj = (j == 0) ? s0 : xdiv(v, piv);
    srow[j] = sj;
    dnew = (j == q) ? xdiv(dq, piv) : fma(-dq, sj, dold);
    T[j] = dnew;
  }
  // level 4: the next leaving row, from column 0 as the update will leave it, with the updated dual devex weights
  Cand rb{0.0, 0.0, 0, 0};
  for (int i = 1 + TIDX; i <= m; i += DA_THREADS) {
    const double ci = colq[i];
    double beta, lb, ub, w;
    if (i == p) {
      beta = xq - s0;
      lb = lbq;
      ub = ubq;
      const double cc = xdiv(wp, piv * piv);
      w = cc > 1.0 ? cc : 1.0;
    } else {
      beta = fma(-ci, s0, betac[i]);
      lb = c->blb[i];
      ub = c->bub[i];
      const double r = xdiv(ci, piv);
      const double cc = r * r * wp;
      w = dwc[i];
      if (cc > w) w = cc;
    }
    if (blockIdx.x == 0) dwn[i] = w;
    double viol;
    int up;
    if (row_violation(beta, lb, ub, tol_bnd, viol, up)) {
      Cand x{xdiv(viol * viol, w), 0.0, i, up};
      if (cand_better<0>(x, rb)) rb = x;
    }
  }
  rb = block_best<0>(rb, lds);
  const int p2 = rb.idx, p2_up = rb.aux;
  // level 5: that row as the update will leave it, priced against the updated objective row
  Cand best{0.0, 0.0, 0, 0};
  if (p2 && act && j >= 1) {
    double a2;
    if (p2 == p) a2 = (j == q) ? xdiv(1.0, piv) : -sj;
    else a2 = (j == q) ? xdiv(colq[p2], piv) : fma(-colq[p2], sj, T[(size_t)p2 * ld + j]);
    Cand x{0.0, 0.0, 0, 0};
    if (dual_ratio_col(a2, sgn * dnew, (j == q) ? lf : fj, p2_up, tol_piv, j, x)) best = x;
  }
  __syncthreads();
  best = block_best<1>(best, lds);
  if (TIDX == 0) ppn[blockIdx.x] = best;
  if (lead) {
    c->step = ST_PIVOT;
    c->p = p;
    c->q = q;
    c->p_up = p_up;
    c->piv = piv;
    c->bound = bound;
    c->xq = xq;
    c->leave_flag = lf;
    c->ent_lb = lbq;
    c->ent_ub = ubq;
    c->curB = cur;
    c->stall_new = (pc.k1 <= DEGEN_TOL) ? stall + 1 : 0;
    c->p_nextx[(cur & 1) ^ 1] = p2;
    c->p_up_nextx[(cur & 1) ^ 1] = p2_up;
  }
}

// ======================================================================= resident-tableau primal path
// Cache-resident sizes (1024x2048: 16.8 MB; every 512x1024 node LP) are bound by launch latency, not by
// bandwidth: two dependent launches per pivot cost ~14 us where the bytes would take ~4.  k_persist keeps
// the whole tableau ON CHIP for hundreds of pivots: one launch, one workgroup per CU, each owning a strip
// of CPW consecutive columns in its LDS (column-major, so that a lane walks rows conflict-free), plus its
// own replica of column 0 (basic values) and of the per-row metadata (basic variable, bounds).
// One exchange per pivot: after applying a step, EVERY workgroup prices its strip, runs the ratio test on its
// own best column as if that column were going to enter, writes the column and the pivot description into its
// slot of a global buffer (8-byte agent-scope stores, drained), and only then publishes a two-word head
// {score, column} as self-tagged granules.  Every workgroup gathers the 256 heads (all loads of a sweep in
// flight), reduces them to the winner, and reads the winner's slot -- which is complete, because its head was
// published after the drain (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload, every storing wave's
// vmcnt(0), workgroup barrier, then the tagged word).  Everything else -- scaled pivot-row entries, the rank-1
// update of the strip, devex weights -- is local.  Arithmetic per entry is that of k_fa / k_fb / the oracle;
// every wait is bounded and ends in a shared abort flag, so the grid always drains (the host then restores its
// backup and falls back on k_fa / k_fb for good).
struct PersistArgs {
  Ctl *ctl;
  unsigned long long *head; // [2][nw][2 words x 2 granules] tagged {score}, {column, direction}
  unsigned long long *slot; // [2][nw][slot_stride] pivot description (PMSG_HDR words) + the candidate column
  int *abort_flag;
  unsigned long long *dbg; // [8] cycle totals of workgroup 0 per phase (propose, gather, read, apply), pivots
  int cpw, nw, slot_stride, max_steps;
  int head_stride; // granules between two strips' heads (>= 4): spreads the heads over memory channels
};
constexpr int PMSG_HDR = 16; // 64-bit words in front of the column in a slot
constexpr int PERSIST_SPIN = 1 << 20;
constexpr int PERSIST_MAX_CPW = 16;

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long d2u(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double u2d(unsigned long long x) { return __longlong_as_double((long long)x); }
// a 64-bit word as two self-validating granules {tag : 32 | half : 32}: the reader needs no flag and no fence,
// a granule is one naturally aligned 8-byte store (MI355X_MICROARCH.md: data-tagged granules)
__device__ __forceinline__ void put_word(unsigned long long *g, unsigned long long v, unsigned tag) {
  const unsigned long long t = (unsigned long long)tag << 32;
  st_agent(g, t | (v >> 32));
  st_agent(g + 1, t | (v & 0xffffffffull));
}
__device__ __forceinline__ bool persist_aborted(const int *abort_flag) {
  return __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// NW words (2 granules each) at g + 2 * (first + k * stride), k = 0..NW-1, all loads of a sweep in flight together;
// re-reads only what has not arrived.  false = gave up (abort)
template <int NW>
__device__ __forceinline__ bool get_words(const unsigned long long *g, int first, int stride, int count, unsigned tag, const int *abort_flag,
                                          unsigned long long (&out)[NW]) {
  unsigned pending = 0;
#pragma unroll
  for (int k = 0; k < NW; k++)
    if (first + k * stride < count) pending |= 1u << k;
  for (int spin = 0; pending && spin < PERSIST_SPIN; spin++) {
    unsigned long long a[NW], b[NW];
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (pending & (1u << k)) {
        const unsigned long long *p = g + 2 * (size_t)(first + k * stride);
        a[k] = ld_agent(p);
        b[k] = ld_agent(p + 1);
      }
#pragma unroll
    for (int k = 0; k < NW; k++)
      if ((pending & (1u << k)) && (unsigned)(a[k] >> 32) == tag && (unsigned)(b[k] >> 32) == tag) {
        out[k] = (a[k] << 32) | (b[k] & 0xffffffffull);
        pending &= ~(1u << k);
      }
    if (pending && (spin & 63) == 63) {
      if (persist_aborted(abort_flag)) return false;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  return pending == 0;
}

__global__ __launch_bounds__(256) void k_persist(PersistArgs a) {
  extern __shared__ double smem[];
  __shared__ Cand lds[17];
  __shared__ unsigned long long s_hdr[PMSG_HDR];
  __shared__ unsigned s_gran[1024]; // low halves of the gathered head granules
  __shared__ double s_sc[PERSIST_MAX_CPW]; // scaled pivot-row entries of the strip
  __shared__ int s_ok;
  Ctl *c = a.ctl;
  if (c->done != D_RUN || c->phase != PH_PRIMAL2 || c->stall >= c->stall_limit || c->budget == 0 || c->perturbed) return;
  if (persist_aborted(a.abort_flag)) return;
  const int m = c->m, n = c->n, cpw = a.cpw, w = (int)blockIdx.x;
  const size_t ld = (size_t)c->ld;
  const int R = m + 1; // rows 0..m
  const int j0 = 1 + w * cpw;
  const int nc = (j0 > n) ? 0 : ((j0 + cpw - 1 <= n) ? cpw : n - j0 + 1); // columns this workgroup owns
  // LDS carve-up (doubles): tile[cpw][R] | beta[R] | colq[R] | blb[R] | bub[R] | bvar (ints)[R]
  double *tile = smem;
  double *beta = tile + (size_t)cpw * R;
  double *colq = beta + R;
  double *blb = colq + R;
  double *bub = blb + R;
  int *bvar = reinterpret_cast<int *>(bub + R);
  // per-column state of the strip: lanes 0..nc-1 of wave 0 own one column each
  const double tol_dj = c->tol_dj, tol_piv = c->tol_piv, sgn = c->sgn;
  const int cur = c->curA & 1;
  double *const pw_g = c->pw[cur];
  int my_nvar = 0, my_nflag = MVX_NS;
  double my_nlb = 0.0, my_nub = 0.0, my_w = 1.0;
  if (TIDX < nc) {
    my_nvar = c->nvar[j0 + TIDX];
    my_nflag = c->nflag[j0 + TIDX];
    my_nlb = c->nlb[j0 + TIDX];
    my_nub = c->nub[j0 + TIDX];
    my_w = pw_g[j0 + TIDX];
  }
  // load the strip (rows 0..m of the owned columns), column 0 and the row metadata
  for (int i = TIDX; i < R; i += 256) {
    const double *row = c->T + (size_t)i * ld;
    for (int cc = 0; cc < nc; cc++) tile[(size_t)cc * R + i] = row[j0 + cc];
    beta[i] = row[0];
    blb[i] = i ? c->blb[i] : 0.0;
    bub[i] = i ? c->bub[i] : 0.0;
    bvar[i] = i ? c->bvar[i] : 0;
  }
  int it_cnt = c->it_cnt, n_flips = c->n_flips, stall = c->stall, budget = c->budget;
  const int stall_limit = c->stall_limit;
  __syncthreads();
  bool ok = true;
  int steps = 0;
  unsigned it = 0; // iteration number; tags are it + 1
  long long t_prop = 0, t_gath = 0, t_read = 0, t_appl = 0, n_sweeps = 0, t_first = 0;
  while (ok) {
    const long long c0 = clock64();
    const unsigned tag = it + 1;
    unsigned long long *myslot = a.slot + ((size_t)(tag & 1) * a.nw + w) * a.slot_stride;
    // ---- this strip's proposal: best column of the strip, and the whole step it would make
    {
      Cand best{0.0, 0.0, 0, 0};
      if (TIDX < 64) {
        if (TIDX < nc) {
          Cand x{0.0, 0.0, 0, 0};
          if (price_col(my_nflag, sgn * tile[(size_t)TIDX * R], tol_dj, j0 + TIDX, my_w, x)) best = x;
        }
        best = wave_best<0>(best);
        if (TIDX == 0) lds[16] = best;
      }
      __syncthreads();
      best = lds[16];
      __syncthreads(); // lds is reused by the ratio test
      const int q = best.idx, sdir = best.aux;
      if (q) {
        const int cq = q - j0;
        const double *col = tile + (size_t)cq * R;
        Cand rb{0.0, 0.0, 0, 0};
        for (int i = 1 + TIDX; i < R; i += 256) {
          Cand x{0.0, 0.0, 0, 0};
          if (ratio_row(col[i], sdir, beta[i], blb[i], bub[i], 0, tol_piv, i, x)) {
            if (cand_better<1>(x, rb)) rb = x;
          }
        }
        const Cand r = block_best<1>(rb, lds);
        // the column's own bounds / status live in lane cq of wave 0
        const double lbq = __shfl(my_nlb, cq, 64), ubq = __shfl(my_nub, cq, 64), wq = __shfl(my_w, cq, 64);
        const int fq = __shfl(my_nflag, cq, 64), vq = __shfl(my_nvar, cq, 64);
        if (TIDX == 0) {
          int kind = ST_PIVOT; // ST_PIVOT / ST_FLIP / ST_STOP
          double delta = 0.0;
          if (lbq > -INFINITY && ubq < INFINITY && fq != MVX_NF) {
            const double tf = ubq - lbq;
            if (r.idx == 0 || tf <= r.k1) {
              kind = ST_FLIP;
              delta = (sdir > 0) ? tf : -tf;
            }
          }
          if (kind == ST_PIVOT && r.idx == 0) kind = ST_STOP; // unbounded ray: the generic path reports it
          const int p = r.idx, p_up = r.aux;
          st_agent(myslot + 1, (unsigned long long)kind);
          st_agent(myslot + 2, (unsigned long long)(unsigned)p | ((unsigned long long)(unsigned)p_up << 32));
          st_agent(myslot + 3, d2u(p ? col[p] : 1.0));                          // piv
          st_agent(myslot + 4, d2u(p ? (p_up ? bub[p] : blb[p]) : 0.0));          // bound
          st_agent(myslot + 5, d2u(dev_nb_value(fq, lbq, ubq)));                 // xq
          st_agent(myslot + 6, d2u(lbq));
          st_agent(myslot + 7, d2u(ubq));
          st_agent(myslot + 8, d2u(delta));
          st_agent(myslot + 9, d2u(wq));
          st_agent(myslot + 10, (unsigned long long)(unsigned)vq);
          st_agent(myslot + 11, d2u(r.k1));
          st_agent(myslot + 12, (unsigned long long)(unsigned)((kind == ST_FLIP) ? ((sdir > 0) ? MVX_NU : MVX_NL) : (p ? dev_leave_flag(blb[p], bub[p], p_up) : 0)));
        }
        for (int i = TIDX; i < R; i += 256) st_agent(myslot + PMSG_HDR + i, d2u(col[i]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the head goes out
      }
      __syncthreads();
      if (TIDX == 0) {
        unsigned long long *g = a.head + ((size_t)(tag & 1) * a.nw + w) * a.head_stride;
        put_word(g, d2u(best.k1), tag);
        put_word(g + 2, ((unsigned long long)(unsigned)best.idx << 1) | (best.aux > 0 ? 1ull : 0ull), tag);
      }
    }
    const long long c1 = clock64();
    t_prop += c1 - c0;
    // ---- gather every strip's head: wave 0, lane l taking strips l, l + 64, ... (four granules each, all loads of a
    // sweep in flight), re-reading only what has not arrived
    if (TIDX < 64) {
      const unsigned long long *g = a.head + (size_t)(tag & 1) * a.nw * a.head_stride;
      unsigned pend = 0;
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (TIDX + 64 * (k >> 2) < a.nw) pend |= 1u << k;
      bool good = true;
      for (int spin = 0; spin < PERSIST_SPIN; spin++) {
        unsigned long long v[16];
        n_sweeps++;
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (pend & (1u << k)) v[k] = ld_agent(g + (size_t)(TIDX + 64 * (k >> 2)) * a.head_stride + (k & 3));
#pragma unroll
        for (int k = 0; k < 16; k++)
          if ((pend & (1u << k)) && (unsigned)(v[k] >> 32) == tag) {
            s_gran[4 * (TIDX + 64 * (k >> 2)) + (k & 3)] = (unsigned)v[k];
            pend &= ~(1u << k);
          }
        if (spin == 0) t_first += clock64() - c1;
        if (__all(pend == 0)) break;
        if ((spin & 63) == 63) {
          if (persist_aborted(a.abort_flag)) {
            good = false;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (spin == PERSIST_SPIN - 1) good = false;
      }
      const int allgood = __all(good ? 1 : 0);
      Cand best{0.0, 0.0, 0, 0};
      if (allgood) {
        for (int st = TIDX; st < a.nw; st += 64) {
          const unsigned long long sc = ((unsigned long long)s_gran[4 * st] << 32) | s_gran[4 * st + 1];
          const unsigned long long mt = ((unsigned long long)s_gran[4 * st + 2] << 32) | s_gran[4 * st + 3];
          const int idx = (int)(mt >> 1);
          if (idx) {
            Cand x{u2d(sc), 0.0, idx, (mt & 1ull) ? 1 : -1};
            if (cand_better<0>(x, best)) best = x;
          }
        }
      }
      best = wave_best<0>(best);
      if (TIDX == 0) {
        lds[16] = best;
        s_ok = allgood;
      }
    }
    __syncthreads();
    if (!s_ok) {
      ok = false;
      break;
    }
    const long long c2 = clock64();
    t_gath += c2 - c1;
    const Cand ent = lds[16];
    const int q = ent.idx;
    if (q == 0 || budget == 0 || stall >= stall_limit || steps >= a.max_steps) break; // the generic path takes over
    const int owner = (q - 1) / cpw;
    // ---- the winner's slot: complete since its head was published after the drain
    {
      const unsigned long long *ws = a.slot + ((size_t)(tag & 1) * a.nw + owner) * a.slot_stride;
      if (TIDX >= 1 && TIDX <= 12) s_hdr[TIDX] = ld_agent(ws + TIDX);
      if (owner == w) {
        const double *col = tile + (size_t)(q - j0) * R;
        for (int i = TIDX; i < R; i += 256) colq[i] = col[i];
      } else {
        for (int i0 = 0; i0 < R; i0 += 1024) {
          unsigned long long v[4];
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (i0 + TIDX + 256 * k < R) v[k] = ld_agent(ws + PMSG_HDR + i0 + TIDX + 256 * k);
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (i0 + TIDX + 256 * k < R) colq[i0 + TIDX + 256 * k] = u2d(v[k]);
        }
      }
      __syncthreads();
    }
    const long long c3 = clock64();
    t_read += c3 - c2;
    // ---- apply the step to the strip
    const int kind = (int)s_hdr[1];
    if (kind == ST_STOP) break;
    const int p = (int)(unsigned)(s_hdr[2] & 0xffffffffull);
    const double piv = u2d(s_hdr[3]), bound = u2d(s_hdr[4]), xq = u2d(s_hdr[5]), ent_lb = u2d(s_hdr[6]), ent_ub = u2d(s_hdr[7]);
    const double delta = u2d(s_hdr[8]), wq = u2d(s_hdr[9]), step_len = u2d(s_hdr[11]);
    const int ent_var = (int)(unsigned)s_hdr[10], newflag = (int)(unsigned)s_hdr[12];
    if (kind == ST_FLIP) {
      // bound flip: only column 0 moves (rows 0..m), column q changes status
      for (int i = TIDX; i < R; i += 256) beta[i] = fma(colq[i], delta, beta[i]);
      if (owner == w && TIDX == q - j0) my_nflag = newflag;
      n_flips++;
      stall = 0;
      it++;
      __syncthreads();
      continue;
    }
    // pivot: scaled pivot-row entries of the strip, objective row, devex weights
    const double s0 = xdiv(beta[p] - bound, piv);
    if (TIDX < nc) {
      const int j = j0 + TIDX;
      double *col = tile + (size_t)TIDX * R;
      const double s_own = xdiv(col[p], piv);
      s_sc[TIDX] = s_own;
      const double dq = colq[0];
      col[0] = (j == q) ? xdiv(dq, piv) : fma(-dq, s_own, col[0]);
      if (j == q) {
        const double cc = xdiv(wq, piv * piv);
        my_w = cc > 1.0 ? cc : 1.0;
      } else {
        const double cc = s_own * s_own * wq;
        my_w = cc > my_w ? cc : my_w;
      }
    }
    // bookkeeping of the swap: the leaving variable takes column q, the entering one takes row p
    const int lv_var = bvar[p];
    const double lv_lb = blb[p], lv_ub = bub[p];
    if (owner == w && TIDX == q - j0) {
      my_nvar = lv_var;
      my_nlb = lv_lb;
      my_nub = lv_ub;
      my_nflag = newflag;
    }
    __syncthreads(); // s_sc is complete; everyone has read row p's metadata and beta[p]
    double sc[PERSIST_MAX_CPW];
    for (int cc = 0; cc < nc; cc++) sc[cc] = s_sc[cc];
    const int cq = (owner == w) ? q - j0 : -1; // the entering column, when it lives in this strip
    // bulk update of the strip (rows 1..m; row 0 is done), column 0, metadata of row p
    for (int i = 1 + TIDX; i < R; i += 256) {
      const double ci = colq[i];
      if (i == p) {
        for (int cc = 0; cc < nc; cc++) tile[(size_t)cc * R + i] = -sc[cc];
        beta[i] = xq - s0;
      } else {
        for (int cc = 0; cc < nc; cc++) {
          double *e = tile + (size_t)cc * R + i;
          *e = fma(-ci, sc[cc], *e);
        }
        beta[i] = fma(-ci, s0, beta[i]);
      }
      if (cq >= 0) tile[(size_t)cq * R + i] = (i == p) ? xdiv(1.0, piv) : xdiv(ci, piv); // column q: T[i][q] = old / piv
    }
    if (TIDX == 0) {
      beta[0] = fma(-colq[0], s0, beta[0]);
      bvar[p] = ent_var;
      blb[p] = ent_lb;
      bub[p] = ent_ub;
    }
    it_cnt++;
    if (budget > 0) budget--;
    stall = (step_len <= DEGEN_TOL) ? stall + 1 : 0;
    steps++;
    it++;
    __syncthreads();
    t_appl += clock64() - c3;
  }
  if (TIDX == 0 && a.dbg && steps > 0) { // spread of the per-workgroup work (propose + read + apply) and gather, cycles per pivot
    const unsigned long long work = (unsigned long long)((t_prop + t_read + t_appl) / steps), gat = (unsigned long long)(t_gath / steps);
    atomicMax(&a.dbg[8], work);
    atomicMax(&a.dbg[9], ~work); // min via max of the complement
    atomicMax(&a.dbg[10], gat);
    atomicMax(&a.dbg[11], ~gat);
  }
  if (w == 0 && TIDX == 0 && a.dbg) {
    a.dbg[0] += (unsigned long long)t_prop; a.dbg[1] += (unsigned long long)t_gath; a.dbg[2] += (unsigned long long)t_read;
    a.dbg[3] += (unsigned long long)t_appl; a.dbg[4] += (unsigned long long)steps;
    a.dbg[5] += (unsigned long long)n_sweeps; a.dbg[6] += (unsigned long long)t_first;
  }
  if (!ok && TIDX == 0) __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- write the strip, column 0 and the metadata back (the host restores its backup after an abort)
  __syncthreads();
  for (int i = TIDX; i < R; i += 256) {
    double *row = c->T + (size_t)i * ld;
    for (int cc = 0; cc < nc; cc++) row[j0 + cc] = tile[(size_t)cc * R + i];
    if (w == 0) {
      row[0] = beta[i];
      if (i) {
        c->bvar[i] = bvar[i];
        c->blb[i] = blb[i];
        c->bub[i] = bub[i];
      }
    }
  }
  if (TIDX < nc) {
    c->nvar[j0 + TIDX] = my_nvar;
    c->nflag[j0 + TIDX] = my_nflag;
    c->nlb[j0 + TIDX] = my_nlb;
    c->nub[j0 + TIDX] = my_nub;
    pw_g[j0 + TIDX] = my_w;
  }
  if (w == 0 && TIDX == 0) {
    c->it_cnt = it_cnt;
    c->n_flips = n_flips;
    c->stall = stall;
    c->budget = budget;
    c->step = ST_NONE;
  }
}

// ======================================================================= GMI cuts on the device
// generateCut3 (/root/reference/gmi.cpp:11-117) and its repaired variant, for `count` basic integer columns of one
// solved node at once.  k_gmi_work turns the tableau row of each column into the coefficient vector `work` by
// variable number (gmi.cpp:41-74): the bug-compatible formula threads a RUNNING right-hand side through the
// non-basic columns in ascending position order (gmi.cpp:55,73), so one lane walks the row; everything around
// it (row loads, bound / kind look-ups, the repaired formula's per-column terms) is done by the whole workgroup.
// k_gmi_backsub is gmi.cpp:81-89: out[col] = work[m+col] + sum over model rows i = 1..m0, in that order, of
// work[i] * A[i][col] -- multiply and add rounded separately, as the host loop does (-ffp-contract=off).
// The caller finishes rows m0+1..m (the node's own appended cut rows) on the host, in the same order.
__device__ __forceinline__ double dev_fract(double x) { // util.cpp:11-23
  double ip;
  double f = modf(x, &ip);
  if (f < 0.0) f += 1;
  return f;
}
__device__ __forceinline__ double api_ub(double ub) { return ub == INFINITY ? 1.79769313486231570815e+308 : ub; }
__device__ __forceinline__ double api_lb(double lb) { return lb == -INFINITY ? -1.79769313486231570815e+308 : lb; }

constexpr int GMI_CH = 1024; // non-basic positions staged per pass

__global__ __launch_bounds__(256) void k_gmi_work(GmiArgs a) {
  __shared__ double s_val[GMI_CH], s_aux[GMI_CH];
  __shared__ int s_var[GMI_CH], s_kind[GMI_CH];
  __shared__ double s_rhs, s_temp;
  __shared__ int s_bad;
  const int c = (int)blockIdx.x;
  const GmiNode nd = a.nodes[c];
  const int m = nd.m, n = a.n;
  const double *row = nd.T + (size_t)nd.pos * nd.ld;
  double *work = a.work + (size_t)c * a.wld;
  for (int v = TIDX; v <= m + n; v += 256) work[v] = 0.0;
  const double beta = row[0];
  const double f0 = dev_fract(beta);
  if (TIDX == 0) {
    s_rhs = a.mode == 0 ? beta : 1.0; // gmi.cpp:37 / the repaired cut's right-hand side starts at 1
    s_temp = 0.0;                     // `temp` is uninitialised at gmi.cpp:13; 0 here and in the oracle
    s_bad = 0;
  }
  __syncthreads();
  for (int base = 1; base <= n; base += GMI_CH) {
    const int cnt = (n - base + 1 < GMI_CH) ? n - base + 1 : GMI_CH;
    for (int t = TIDX; t < cnt; t += 256) {
      const int jj = base + t;
      const double val = row[jj];
      const int var = nd.nvar[jj];
      // glp_get_col_kind: an integer column with bounds [0,1] reads as GLP_BV; auxiliaries are continuous
      int kind = MVX_CV;
      const double lb = nd.nlb[jj], ub = nd.nub[jj];
      if (var > m) {
        kind = a.kind[var - m];
        if (kind == MVX_IV && lb == 0.0 && ub == 1.0) kind = MVX_BV;
      }
      s_val[t] = val;
      s_var[t] = var;
      if (a.mode == 0) {
        s_kind[t] = kind;
        s_aux[t] = api_ub(ub); // gmi.cpp:47,52
      } else {
        // repaired: this column's term of the cut and of its right-hand side
        const int stat = nd.nflag[jj];
        int code = 0; // 0 skip, 1 at lower, 2 at upper
        double g = 0.0, term = 0.0;
        if (val != 0.0 && stat != MVX_NS) {
          if (stat == MVX_NF) {
            code = 3;
          } else {
            const double abar = (stat == MVX_NL) ? -val : val;
            if (kind != MVX_CV) {
              const double fj = dev_fract(abar);
              g = (fj <= f0) ? xdiv(fj, f0) : xdiv(1.0 - fj, 1.0 - f0);
            } else {
              g = (abar >= 0.0) ? xdiv(abar, f0) : xdiv(-abar, 1.0 - f0);
            }
            if (stat == MVX_NL) {
              code = 1;
              term = g * api_lb(lb);
            } else {
              code = 2;
              term = -(g * api_ub(ub));
            }
          }
        }
        s_kind[t] = code;
        s_aux[t] = term;
        if (code == 1) work[var] = 0.0 + g;
        if (code == 2) work[var] = 0.0 - g;
      }
    }
    __syncthreads();
    if (TIDX == 0) {
      double rhs = s_rhs;
      if (a.mode == 0) {
        double temp = s_temp;
        for (int t = 0; t < cnt; t++) {
          const double val = s_val[t];
          if (val == 0.0) continue; // glp_eval_tab_row returns the non-zeros only
          const int kind = s_kind[t];
          const double fRhs = dev_fract(rhs); // the RUNNING rhs (gmi.cpp:55,73)
          const double fVal = dev_fract(val);
          if (kind == MVX_IV) temp = (fRhs >= fVal) ? fVal : xdiv(fRhs, 1.0 - fRhs) * (1.0 - fVal);
          if (kind == MVX_CV) temp = (val >= 0.0) ? val : xdiv(fRhs, 1.0 - fRhs) * (-1.0 * val);
          work[s_var[t]] = -1.0 * temp; // gmi.cpp:72
          rhs = rhs - temp * s_aux[t];  // gmi.cpp:73
        }
        s_temp = temp;
      } else {
        int bad = s_bad;
        for (int t = 0; t < cnt; t++) {
          const int code = s_kind[t];
          if (code == 3) bad = 1;
          if (code == 1 || code == 2) rhs = rhs + s_aux[t];
        }
        s_bad = bad;
      }
      s_rhs = rhs;
    }
    __syncthreads();
  }
  if (TIDX == 0) {
    a.rhs[c] = s_rhs;
    a.ok[c] = s_bad ? 0 : 1;
  }
}

constexpr int GMI_CT = 4; // cuts per lane in the back-substitution (each loaded matrix entry serves four cuts)

__global__ __launch_bounds__(256) void k_gmi_backsub(GmiArgs a) {
  __shared__ double s_w[64][GMI_CT];
  const int col = 1 + (int)blockIdx.x * 256 + TIDX;
  const int c0 = (int)blockIdx.y * GMI_CT;
  const bool act = col <= a.n;
  double acc[GMI_CT];
#pragma unroll
  for (int u = 0; u < GMI_CT; u++) acc[u] = (act && c0 + u < a.count) ? a.work[(size_t)(c0 + u) * a.wld + a.nodes[c0 + u].m + col] : 0.0;
  for (int i0 = 1; i0 <= a.m0; i0 += 64) {
    __syncthreads();
    {
      const int r = TIDX >> 2, u = TIDX & 3; // 64 rows x 4 cuts
      const int i = i0 + r;
      s_w[r][u] = (i <= a.m0 && c0 + u < a.count) ? a.work[(size_t)(c0 + u) * a.wld + i] : 0.0;
    }
    __syncthreads();
    const int cnt = (a.m0 - i0 + 1 < 64) ? a.m0 - i0 + 1 : 64;
    if (act) {
      for (int r = 0; r < cnt; r++) {
        const int i = i0 + r;
        const double av = a.A[(size_t)i * a.lda + col];
        if (a.mode == 0) {
          // position `col` of row i's non-zero list (gmi.cpp:87 indexes by position, not by column)
          if (a.len && col > a.len[i]) continue;
#pragma unroll
          for (int u = 0; u < GMI_CT; u++) acc[u] = acc[u] + s_w[r][u] * av;
        } else {
          if (av == 0.0) continue;
#pragma unroll
          for (int u = 0; u < GMI_CT; u++)
            if (s_w[r][u] != 0.0) acc[u] = acc[u] + s_w[r][u] * av;
        }
      }
    }
  }
  if (act) {
#pragma unroll
    for (int u = 0; u < GMI_CT; u++)
      if (c0 + u < a.count) a.out[(size_t)(c0 + u) * a.old + col] = acc[u];
  }
}

// LDS bytes of one k_persist workgroup: the strip, column 0, the pivot column, row bounds (f64) and basic variables (i32)
size_t persist_lds_bytes(int m, int cpw) { return ((size_t)(m + 1) * (size_t)(cpw + 4)) * 8 + (size_t)(m + 1) * 4 + 64; }
int persist_max_cpw() { return PERSIST_MAX_CPW; }
int persist_slot_words(int m_cap) { return (PMSG_HDR + m_cap + 1 + 31) / 32 * 32; }
int launch_persist(Ctl *d_ctl, unsigned long long *head, unsigned long long *slot, int *abort_flag, unsigned long long *dbg, int m, int cpw, int nw,
                   int slot_stride, int max_steps, int head_stride, hipStream_t s) {
  static size_t attr_bytes = 0;
  const size_t lds = persist_lds_bytes(m, cpw);
  if (lds > attr_bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    attr_bytes = lds;
  }
  PersistArgs a;
  a.ctl = d_ctl; a.head = head; a.slot = slot; a.abort_flag = abort_flag; a.dbg = dbg;
  a.cpw = cpw; a.nw = nw; a.slot_stride = slot_stride; a.max_steps = max_steps;
  a.head_stride = head_stride;
  hipLaunchKernelGGL(k_persist, dim3((unsigned)nw), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

void launch_gmi(const GmiArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_gmi_work, dim3((unsigned)a.count), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_gmi_backsub, dim3((unsigned)((a.n + 255) / 256), (unsigned)((a.count + GMI_CT - 1) / GMI_CT)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ launch wrappers

// tuning knobs of the streamed update (mvx_set_tuning; defaults are the measured best)
static int g_tr = 0, g_hot = 1, g_nt = -1; // g_tr 0 = pick from the grid size; g_nt -1 = pick from the tableau size
void set_tuning(int tr, int hot, int nt) {
  g_tr = (tr == 4 || tr == 8 || tr == 16 || tr == 32) ? tr : 0;
  g_hot = hot ? 1 : 0;
  g_nt = (nt >= 0 && nt <= 2) ? nt : -1; // 0 plain, 1 non-temporal, 2 write-through stores; anything else: by size
}
// How the update kernels touch the tableau, by its size (scripts/ntsweep.py, profiles/r02_ntsweep_sizes.jsonl):
//   1  non-temporal loads and stores once the tableau no longer fits the 256 MiB Infinity Cache: every entry is touched
//      once per pivot, so lines kept for reuse only evict each other -- 8192x8192 (537 MB): 5.34 -> 6.13 TB/s;
//   2  write-through stores (sc1) from about 100 MiB up to the cache size: the cache still serves the loads, and a line
//      written through does not sit in L2 as a dirty line until evicted -- 3-4 % less time per launch at 145-257 MiB;
//   0  plain access below that (a tie at 65 MiB) and in the narrow band between the two (289 MiB: plain is fastest).
static int pick_nt(int m, int n) {
  if (g_nt >= 0) return g_nt;
  const size_t bytes = (size_t)(m + 1) * (size_t)((n + 1 + LD_ALIGN - 1) / LD_ALIGN * LD_ALIGN) * 8;
  if (bytes > NT_THRESHOLD_BYTES) return 1;
  return (bytes >= WT_MIN_BYTES && bytes <= WT_MAX_BYTES) ? 2 : 0;
}
// row-block depth: 16 rows per block once that still gives every CU several blocks, else 8
static int pick_tr(int m, int n) {
  if (g_tr) return g_tr;
  const long tiles = ((n + 2) / 2 + 255) / 256;
  if ((long)((m + 15) / 16) * tiles >= 2048) return 16;
  if ((long)((m + 7) / 8) * tiles >= 2048) return 8;
  return 4;
}
int fused_npb(int n) { return (n + 1 + 255) / 256; }
int fused_nrb_max(int m) { return (m + 3) / 4; }
int chain_ncb(int n) { return fused_npb(n); }
int chain_nrb(int m) { return (m + 255) / 256; }
void launch_pboot(const ChainArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(k_pboot, dim3(a.ncb > a.nrb ? a.ncb : a.nrb), dim3(256), 0, s, a);
}
void launch_pstep(const ChainArgs &a, int g, hipStream_t s) {
  hipLaunchKernelGGL(k_pc, dim3(a.nrb), dim3(256), 0, s, a, g);
  hipLaunchKernelGGL(k_pr, dim3(a.ncb), dim3(256), 0, s, a, g);
}
void launch_pc(const ChainArgs &a, int g, hipStream_t s) { hipLaunchKernelGGL(k_pc, dim3(a.nrb), dim3(256), 0, s, a, g); }
// k_chain: nw workgroups of 256 threads cover the rows and the columns CPT / RPT to a thread; LDS = the chain's history
int chain_cluster_nw(int m, int n) {
  const int span = m > n ? m : n;
  int nw = (span + 255) / 256;
  if (nw > 32) nw = 32;
  return nw < 1 ? 1 : nw;
}
// longest chain k_chain can keep in LDS for this geometry (0: the geometry is not covered)
int chain_cluster_kmax(int m, int n) {
  const int nw = chain_cluster_nw(m, n), tt = nw * 256;
  const int cpt = (n + tt - 1) / tt, rpt = (m + tt - 1) / tt;
  if (cpt > 2 || rpt > 2) return 0;
  const int per_step = (cpt > 1 || rpt > 1 ? 4 : 2) * 256 * 8;
  const int k = (140 * 1024) / per_step;
  return k > KCH ? KCH : k;
}
int launch_chain(const ChainArgs &a, hipStream_t s) {
  const int tt = a.nw * 256;
  const int cpt = (a.n + tt - 1) / tt, rpt = (a.m + tt - 1) / tt;
  const bool wide = (cpt > 1 || rpt > 1);
  const size_t lds = (size_t)(a.kmax > 0 ? a.kmax : 1) * (wide ? 4 : 2) * 256 * 8;
  static size_t attr1 = 0, attr2 = 0;
  size_t &attr = wide ? attr2 : attr1;
  const void *fn = wide ? reinterpret_cast<const void *>(k_chain<2, 2>) : reinterpret_cast<const void *>(k_chain<1, 1>);
  if (lds > attr) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return -1;
    }
    attr = lds;
  }
  if (wide) hipLaunchKernelGGL((k_chain<2, 2>), dim3(8 * a.nw), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((k_chain<1, 1>), dim3(8 * a.nw), dim3(256), lds, s, a);
  return 0;
}
void launch_fpatch(const ChainArgs &a, int steps, hipStream_t s) {
  const int span = a.mcap1 > a.n + 1 ? a.mcap1 : a.n + 1;
  hipLaunchKernelGGL(k_fpatch, dim3((span + 255) / 256, 2 * steps + 1), dim3(256), 0, s, a, steps);
}
void launch_fbc3(const ChainArgs &a, int steps, hipStream_t s) {
  const int m = a.m, n = a.n;
  const int pairs = (n + 2) / 2;
  int tr = pick_tr(m, n);
  if (tr > 16) tr = 16;
  // long chains: shallower tiles, more waves to a SIMD to cover the per-step operand fetches (scripts/bulktime.py:
  // 4096x8192, 32 steps: 117.7 us with 16-row tiles, 104.8 with 8; 10 steps: 82.3 / 81.6; one step: 76.1 / 79.6)
  if (!g_tr && tr == 16 && steps >= 12) tr = 8;
  const int nt = pick_nt(m, n);
  dim3 grid((pairs + 255) / 256, (m + tr - 1) / tr);
#define FBC2_CASE(TR_, NT_) \
  if (tr == TR_ && nt == NT_) { hipLaunchKernelGGL((k_fbc3<TR_, NT_>), grid, dim3(256), 0, s, a); return; }
  FBC2_CASE(16, 0) FBC2_CASE(16, 1) FBC2_CASE(16, 2) FBC2_CASE(8, 0) FBC2_CASE(8, 1) FBC2_CASE(8, 2) FBC2_CASE(4, 0) FBC2_CASE(4, 1) FBC2_CASE(4, 2)
#undef FBC2_CASE
  std::abort(); // unreachable
}
void launch_dboot(Ctl *d_ctl, int n, hipStream_t s) { hipLaunchKernelGGL(k_dboot, dim3((n + DA_THREADS) / DA_THREADS), dim3(DA_THREADS), 0, s, d_ctl); }
void launch_da(Ctl *d_ctl, int n, hipStream_t s) { hipLaunchKernelGGL(k_da, dim3((n + DA_THREADS) / DA_THREADS), dim3(DA_THREADS), 0, s, d_ctl); }
void launch_db(Ctl *d_ctl, int m, int n, hipStream_t s) {
  const int pairs = (n + 2) / 2;
  const int tr = pick_tr(m, n);
  const int nt = pick_nt(m, n);
  dim3 grid((pairs + 255) / 256, (m + tr - 1) / tr);
#define DB_CASE(TR_, NT_) \
  if (tr == TR_ && nt == NT_) { hipLaunchKernelGGL((k_fb<TR_, 1, NT_, 1>), grid, dim3(256), 0, s, d_ctl); return; }
  DB_CASE(16, 0) DB_CASE(16, 1) DB_CASE(8, 0) DB_CASE(8, 1) DB_CASE(4, 0) DB_CASE(4, 1) DB_CASE(32, 0) DB_CASE(32, 1)
  DB_CASE(16, 2) DB_CASE(8, 2) DB_CASE(4, 2) DB_CASE(32, 2)
#undef DB_CASE
  std::abort(); // unreachable
}
// k_dsel in front of k_select (node LPs of up to 1024 rows and columns); MVX_DSEL=0 or a refused LDS attribute: never
int launch_dsel(Ctl *d_ctl, int m, int n, hipStream_t s, int slots) {
  static int state = -1; // -1 not tried, 0 off, 1 on
  if (state < 0) {
    const char *e = std::getenv("MVX_DSEL");
    if (e && e[0] == '0') state = 0;
    else if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dsel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DSEL_LDS) == hipSuccess) state = 1;
    else {
      (void)hipGetLastError();
      state = 0;
    }
  }
  if (state != 1 || m > 1024 || n > 1024) return 0;
  hipLaunchKernelGGL(k_dsel, dim3(1, 1, slots), dim3(1024), DSEL_LDS, s, d_ctl);
  return 1;
}
void launch_select(Ctl *d_ctl, hipStream_t s, int slots) {
  BatchQueue q{};
  hipLaunchKernelGGL(k_select, dim3(1, 1, slots), dim3(1024), 0, s, d_ctl, q);
}
void launch_select_queue(Ctl *d_ctl, const BatchQueue &q, hipStream_t s, int slots) {
  hipLaunchKernelGGL(k_select, dim3(1, 1, slots), dim3(1024), 0, s, d_ctl, q);
}
void launch_update(Ctl *d_ctl, int m, int n, hipStream_t s, int slots, int chained, int busy_slots) {
  const int pairs = (n + 2) / 2;
  const long tiles = (pairs + 255) / 256;
  // 16-row tiles once the launch still has >= 2048 workgroups with work in them, else 8, else 4 (latency-bound sizes);
  // busy_slots: the slots of a batched launch that still hold a running solve (0: all of them)
  const long work = busy_slots > 0 ? std::min(busy_slots, slots) : slots;
  int tr = ((long)((m + 16) / 16) * tiles * work >= 2048) ? 16 : ((long)((m + 8) / 8) * tiles * work >= 2048) ? 8 : 4;
  // batched node LPs: the chained update's 16-row tile is 168 VGPRs / 3 waves per SIMD (the exceptions of the chain's
  // steps sit in its loop), the 4-row one 56 / 8: 2000 nodes of the wide tree 113 ms with 16-row tiles, 111 with 8, 109
  // with 4 (MVX_UPD_TR=16 / 8 bring the deeper tiles back)
  static const int tr_cap = std::getenv("MVX_UPD_TR") ? std::atoi(std::getenv("MVX_UPD_TR")) : 4;
  if (slots > 1 && chained && tr > tr_cap) tr = tr_cap >= 16 ? 16 : tr_cap >= 8 ? 8 : 4;
  dim3 grid((unsigned)tiles, (m + tr) / tr, slots);
  const int nt = (tr == 16) ? pick_nt(m, n) : 0;
  if (nt == 1) hipLaunchKernelGGL((k_update<16, 1>), grid, dim3(256), 0, s, d_ctl, chained);
  else if (nt == 2) hipLaunchKernelGGL((k_update<16, 2>), grid, dim3(256), 0, s, d_ctl, chained);
  else if (tr == 16) hipLaunchKernelGGL((k_update<16, 0>), grid, dim3(256), 0, s, d_ctl, chained);
  else if (tr == 8) hipLaunchKernelGGL((k_update<8, 0>), grid, dim3(256), 0, s, d_ctl, chained);
  else hipLaunchKernelGGL((k_update<4, 0>), grid, dim3(256), 0, s, d_ctl, chained);
}
void launch_p1_head(Ctl *d_ctl, hipStream_t s) { hipLaunchKernelGGL(k_p1_head, dim3(1), dim3(1024), 0, s, d_ctl); }
void launch_p1_select(Ctl *d_ctl, hipStream_t s) { hipLaunchKernelGGL(k_p1_select, dim3(1), dim3(1024), 0, s, d_ctl); }
// control blocks of freshly filled batch slots: one upload of the packed blocks + this scatter instead of one
// small host-to-device copy per slot (each of those is a blit kernel of ~5 us on the device)
__global__ __launch_bounds__(64) void k_scatter_ctl(Ctl *dst, const Ctl *src, const int *idx, int count) {
  const int t = (int)blockIdx.x;
  if (t >= count) return;
  const unsigned *s = reinterpret_cast<const unsigned *>(&src[t]);
  unsigned *d = reinterpret_cast<unsigned *>(&dst[idx[t]]);
  for (int w = TIDX; w < (int)(sizeof(Ctl) / 4); w += 64) d[w] = s[w];
}
void launch_scatter_ctl(Ctl *dst, const Ctl *src, const int *idx, int count, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter_ctl, dim3(count), dim3(64), 0, s, dst, src, idx, count);
}
// Device-to-device clones of a round of B&B branchings (glp_copy_prob, bs.cpp:269-273) in ONE launch: a 4 MB copy
// on its own cannot fill the chip (5.9 us each as separate copyBuffer launches), 64 of them together stream at HBM rate.
__global__ __launch_bounds__(256) void k_copy_many(CopyBatch b) {
  const CopyJob j = b.jobs[blockIdx.y];
  const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(j.src);
  uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(j.dst);
  const size_t n = j.bytes >> 4;
  constexpr int U = 8;
  for (size_t base = (size_t)blockIdx.x * (256 * U); base < n; base += (size_t)gridDim.x * (256 * U)) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t i = base + (size_t)u * 256 + TIDX;
      if (i < n) v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t i = base + (size_t)u * 256 + TIDX;
      if (i < n) dst[i] = v[u];
    }
  }
}
void launch_copy_many(const CopyBatch &b, hipStream_t s) {
  size_t mx = 0;
  for (int k = 0; k < b.count; k++) mx = b.jobs[k].bytes > mx ? b.jobs[k].bytes : mx;
  const size_t per_block = (size_t)256 * 8 * 16;
  unsigned gx = (unsigned)((mx + per_block - 1) / per_block);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_copy_many, dim3(gx, (unsigned)b.count), dim3(256), 0, s, b);
}
void launch_p1_fix(Ctl *d_ctl, int n, hipStream_t s) { hipLaunchKernelGGL(k_p1_fix, dim3((n + 1 + 255) / 256), dim3(256), 0, s, d_ctl); }
void launch_rowcomb(Ctl *d_ctl, int m, int n, int respect_done, hipStream_t s) {
  const int nchunks = (m + ROWCOMB_CHUNK - 1) / ROWCOMB_CHUNK;
  dim3 grid((n + 1 + 255) / 256, nchunks > 0 ? nchunks : 1);
  hipLaunchKernelGGL(k_rowcomb_partial, grid, dim3(256), 0, s, d_ctl, respect_done);
  hipLaunchKernelGGL(k_rowcomb_final, dim3((n + 1 + 255) / 256), dim3(256), 0, s, d_ctl, respect_done, nchunks);
}
void launch_shift_nonbasic(double *T, int ld, int m, int jj, double delta, hipStream_t s) {
  hipLaunchKernelGGL(k_shift_nonbasic, dim3((m + 1 + 255) / 256), dim3(256), 0, s, T, ld, m, jj, delta);
}
void launch_set_basic_bounds(double *blb, double *bub, int i, double lb, double ub, hipStream_t s) {
  hipLaunchKernelGGL(k_set_basic_bounds, dim3(1), dim3(64), 0, s, blb, bub, i, lb, ub);
}
void launch_set_nonbasic(double *nlb, double *nub, int *nflag, int j, double lb, double ub, int flag, hipStream_t s) {
  hipLaunchKernelGGL(k_set_nonbasic, dim3(1), dim3(64), 0, s, nlb, nub, nflag, j, lb, ub, flag);
}
void launch_add_rows(double *T, int ld, int n, int *bvar, double *blb, double *bub, int *nvar, int first, int nrs, int m_new,
                     hipStream_t s) {
  int span = (n > m_new ? n : m_new) + 1;
  hipLaunchKernelGGL(k_add_rows, dim3((span + 255) / 256), dim3(256), 0, s, T, ld, n, bvar, blb, bub, nvar, first, nrs);
}
void launch_export(Ctl *d_ctl, unsigned char *stage, int m, int n, int force, hipStream_t s, int slots, size_t slot_stride) {
  int span = (n > m ? n : m) + 1;
  hipLaunchKernelGGL(k_export, dim3((span + 255) / 256, 1, slots), dim3(256), 0, s, d_ctl, stage, force, slot_stride);
}

} // namespace mvx
