// Host-side set-up of a smoothed-aggregation multigrid hierarchy for the SPD operator
// A_hat = M + dt K (Dirichlet rows = identity).  Plain C++ (no HIP): strength graph,
// greedy aggregation, Jacobi-smoothed prolongator P = (I - w D^-1 A) T, Galerkin
// A_c = P^T A P.  The dense inverse of the coarsest operator and the V-cycle itself run on the GPU
// (heatflow_hip.hip) with the same CSR SpMV kernels as the PCG loop.
//
// This has no counterpart in the reference (it factors with MUMPS, run_with_diamond.py:389-394);
// it is the preconditioner that lets PCG reach the reference's answer in ~15 instead of ~800
// iterations on the 1M-DOF mesh.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace amg {

struct Csr {
  int nrow = 0, ncol = 0;
  std::vector<int> ptr, idx;
  std::vector<double> val;
  int64_t nnz() const { return static_cast<int64_t>(idx.size()); }
};

inline Csr transpose(const Csr& A) {
  Csr T;
  T.nrow = A.ncol;
  T.ncol = A.nrow;
  T.ptr.assign(static_cast<size_t>(A.ncol) + 1, 0);
  for (int c : A.idx) T.ptr[c + 1]++;
  for (int i = 0; i < A.ncol; ++i) T.ptr[i + 1] += T.ptr[i];
  T.idx.resize(A.idx.size());
  T.val.resize(A.val.size());
  std::vector<int> cur(T.ptr.begin(), T.ptr.end() - 1);
  for (int i = 0; i < A.nrow; ++i)
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const int q = cur[A.idx[k]]++;
      T.idx[q] = i;  // rows visited in increasing order -> sorted columns
      T.val[q] = A.val[k];
    }
  return T;
}

// C = A * B (Gustavson, dense accumulator), columns of every row sorted.  Serial on purpose: row ranges on host
// threads were measured slower than one thread on the GPU boxes (0.05 s serial, 0.11-0.29 s with 2-16 threads for
// the 1M-row A*P; scripts/micro notes) - their cores are a share of a large NUMA machine.
inline Csr spgemm(const Csr& A, const Csr& B) {
  Csr C;
  C.nrow = A.nrow;
  C.ncol = B.ncol;
  C.ptr.assign(static_cast<size_t>(A.nrow) + 1, 0);
  std::vector<double> acc(B.ncol, 0.0);
  std::vector<int> mark(B.ncol, -1), cols;
  C.idx.reserve(A.idx.size() * 2);
  C.val.reserve(A.idx.size() * 2);
  for (int i = 0; i < A.nrow; ++i) {
    cols.clear();
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const int a = A.idx[k];
      const double av = A.val[k];
      for (int q = B.ptr[a]; q < B.ptr[a + 1]; ++q) {
        const int j = B.idx[q];
        if (mark[j] != i) { mark[j] = i; acc[j] = 0.0; cols.push_back(j); }
        acc[j] += av * B.val[q];
      }
    }
    std::sort(cols.begin(), cols.end());
    for (int j : cols) { C.idx.push_back(j); C.val.push_back(acc[j]); }
    C.ptr[i + 1] = static_cast<int>(C.idx.size());
  }
  return C;
}

inline std::vector<double> diagonal(const Csr& A) {
  std::vector<double> d(A.nrow, 0.0);
  for (int i = 0; i < A.nrow; ++i)
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
      if (A.idx[k] == i) d[i] = A.val[k];
  return d;
}

// Greedy aggregation on the strength graph |a_ij| >= theta sqrt(a_ii a_jj) (Vanek et al.):
// pass 1 roots whose whole strong neighbourhood is free, pass 2 attaches leftovers to the
// aggregate of their strongest aggregated neighbour (pass-1 state), pass 3 groups the rest.
// Nodes without strong connections (Dirichlet rows, diagonally dominant rows) get agg = -1:
// they take no coarse correction, the Jacobi smoother treats them.
inline int aggregate(const Csr& A, const std::vector<double>& d, double theta, std::vector<int>& agg, bool attach_weak = false) {
  const int n = A.nrow;
  std::vector<int> sptr(static_cast<size_t>(n) + 1, 0), sidx;
  std::vector<double> sval;
  sidx.reserve(A.idx.size());
  sval.reserve(A.idx.size());
  const double t2 = theta * theta;
  for (int i = 0; i < n; ++i) {
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const int j = A.idx[k];
      const double v = A.val[k];
      if (j != i && v != 0.0 && v * v >= t2 * d[i] * d[j]) { sidx.push_back(j); sval.push_back(std::fabs(v)); }
    }
    sptr[i + 1] = static_cast<int>(sidx.size());
  }
  agg.assign(n, -1);
  int na = 0;
  for (int i = 0; i < n; ++i) {                       // pass 1
    if (agg[i] != -1 || sptr[i] == sptr[i + 1]) continue;
    bool free_nb = true;
    for (int k = sptr[i]; k < sptr[i + 1] && free_nb; ++k) free_nb = agg[sidx[k]] == -1;
    if (!free_nb) continue;
    agg[i] = na;
    for (int k = sptr[i]; k < sptr[i + 1]; ++k) agg[sidx[k]] = na;
    ++na;
  }
  std::vector<int> agg1(agg);
  for (int i = 0; i < n; ++i) {                       // pass 2
    if (agg1[i] != -1 || sptr[i] == sptr[i + 1]) continue;
    double best = -1.0;
    int who = -1;
    for (int k = sptr[i]; k < sptr[i + 1]; ++k)
      if (agg1[sidx[k]] != -1 && sval[k] > best) { best = sval[k]; who = agg1[sidx[k]]; }
    if (who >= 0) agg[i] = who;
  }
  for (int i = 0; i < n; ++i) {                       // pass 3
    if (agg[i] != -1 || sptr[i] == sptr[i + 1]) continue;
    agg[i] = na;
    for (int k = sptr[i]; k < sptr[i + 1]; ++k)
      if (agg[sidx[k]] == -1) agg[sidx[k]] = na;
    ++na;
  }
  if (attach_weak) {
    // pass 4: rows whose couplings all fall below the threshold still take a coarse correction - through the aggregate
    // of their largest coupling (a new aggregate if that neighbour has none); only uncoupled rows (Dirichlet) stay out
    for (int i = 0; i < n; ++i) {
      if (agg[i] != -1) continue;
      double best = 0.0;
      int who = -1;
      for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
        const int j = A.idx[k];
        if (j != i && std::fabs(A.val[k]) > best) { best = std::fabs(A.val[k]); who = j; }
      }
      if (who < 0) continue;
      if (agg[who] == -1) agg[who] = na++;
      agg[i] = agg[who];
    }
  }
  return na;
}

// P = (I - w D^-1 A) T with T the piecewise-constant aggregate indicator.
inline Csr smoothed_prolongator(const Csr& A, const std::vector<double>& d, const std::vector<int>& agg, int na,
                                double w) {
  Csr P;
  P.nrow = A.nrow;
  P.ncol = na;
  P.ptr.assign(static_cast<size_t>(A.nrow) + 1, 0);
  std::vector<double> acc(na, 0.0);
  std::vector<int> mark(na, -1), cols;
  for (int i = 0; i < A.nrow; ++i) {
    cols.clear();
    if (agg[i] >= 0) { mark[agg[i]] = i; acc[agg[i]] = 1.0; cols.push_back(agg[i]); }
    // rows without an aggregate stay zero: (I - w D^-1 A) T has the row -w/d_i sum_j a_ij T_j, which is
    // dropped on purpose so that isolated / Dirichlet rows receive no coarse correction
    if (agg[i] >= 0) {
      const double s = -w / d[i];
      for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
        const int c = agg[A.idx[k]];
        if (c < 0) continue;
        if (mark[c] != i) { mark[c] = i; acc[c] = 0.0; cols.push_back(c); }
        acc[c] += s * A.val[k];
      }
    }
    std::sort(cols.begin(), cols.end());
    for (int c : cols) { P.idx.push_back(c); P.val.push_back(acc[c]); }
    P.ptr[i + 1] = static_cast<int>(P.idx.size());
  }
  return P;
}

inline double gershgorin_rho(const Csr& A, const std::vector<double>& d) {  // bound on lambda_max(D^-1 A)
  double rho = 0.0;
  for (int i = 0; i < A.nrow; ++i) {
    double s = 0.0;
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) s += std::fabs(A.val[k]);
    rho = std::max(rho, s / d[i]);
  }
  return rho;
}

struct Level {
  Csr A;                    // operator of this level (level 0: not stored here, the caller owns it)
  Csr P, R;                 // to / from the next coarser level (empty on the coarsest)
  // Intermediate levels (neither finest nor coarsest) also carry the V(1,1) cycle's two legs as single
  // operators, so that a level costs two SpMV launches instead of four (same cycle in exact arithmetic):
  //   down:  b_{l+1} = R (b - A w D^-1 b)                       = Rt b,           Rt = ((I - w D^-1 A) P)^T
  //   up:    x = (I - w D^-1 A)(w D^-1 b + P e) + w D^-1 b       = GP [b; e],      GP = [2wD^-1 - w^2 D^-1 A D^-1 | (I - w D^-1 A) P]
  Csr Rt, GP;
  std::vector<double> dinv;
  double omega = 0.0;       // Jacobi damping 4 / (3 rho)
};

// Pt = (I - w D^-1 A) P = P - w D^-1 (A P), from the product A P the Galerkin step has anyway: per row a merge of two
// sorted column lists (rows of P without an aggregate stay empty in P; their A P part is kept: they are smoothed
// against their neighbours' corrections like every other row)
inline Csr smoothed_by_product(const Csr& P, const Csr& AP, const std::vector<double>& dinv, double w) {
  Csr C;
  C.nrow = P.nrow;
  C.ncol = P.ncol;
  C.ptr.assign(static_cast<size_t>(P.nrow) + 1, 0);
  C.idx.reserve(AP.idx.size());
  C.val.reserve(AP.idx.size());
  for (int i = 0; i < P.nrow; ++i) {
    int a = P.ptr[i], b = AP.ptr[i];
    const int a1 = P.ptr[i + 1], b1 = AP.ptr[i + 1];
    const double s = -w * dinv[i];
    while (a < a1 || b < b1) {
      const int ca = a < a1 ? P.idx[a] : INT32_MAX, cb = b < b1 ? AP.idx[b] : INT32_MAX;
      if (ca == cb) { C.idx.push_back(ca); C.val.push_back(P.val[a] + s * AP.val[b]); ++a; ++b; }
      else if (ca < cb) { C.idx.push_back(ca); C.val.push_back(P.val[a]); ++a; }
      else { C.idx.push_back(cb); C.val.push_back(s * AP.val[b]); ++b; }
    }
    C.ptr[i + 1] = static_cast<int>(C.idx.size());
  }
  return C;
}

// [G | Pt] with G = 2 w D^-1 - w^2 D^-1 A D^-1 (pattern of A) and the columns of Pt shifted by A.ncol
inline Csr fused_up_leg(const Csr& A, const std::vector<double>& dinv, double w, const Csr& Pt) {
  Csr C;
  C.nrow = A.nrow;
  C.ncol = A.ncol + Pt.ncol;
  C.ptr.assign(static_cast<size_t>(A.nrow) + 1, 0);
  C.idx.reserve(A.idx.size() + Pt.idx.size());
  C.val.reserve(A.idx.size() + Pt.idx.size());
  for (int i = 0; i < A.nrow; ++i) {
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const int j = A.idx[k];
      C.idx.push_back(j);
      C.val.push_back((j == i ? 2.0 * w * dinv[i] : 0.0) - w * w * dinv[i] * A.val[k] * dinv[j]);
    }
    for (int k = Pt.ptr[i]; k < Pt.ptr[i + 1]; ++k) {
      C.idx.push_back(A.ncol + Pt.idx[k]);
      C.val.push_back(Pt.val[k]);
    }
    C.ptr[i + 1] = static_cast<int>(C.idx.size());
  }
  return C;
}

struct Hierarchy {
  std::vector<Level> levels;          // levels[0].A is left empty (the fine operator lives on the device)
  int coarse_n = 0;                   // rows of the coarsest operator (its dense inverse is formed on the device)
  double op_complexity = 0.0;
};

struct Params {
  double theta = 0.08;      // strength threshold on level 0, halved per level
  int coarse_size = 2500;   // stop when a level has at most this many rows (dense inverse on the GPU)
  int max_levels = 12;
  // Smoother damping w = smooth_scale * 4/(3 rho) with rho the Gershgorin bound on lambda_max(D^-1 A).
  // The bound overestimates lambda_max, so scale 1 under-relaxes; any scale < 1.5 keeps w*lambda_max < 2
  // (the damped-Jacobi sweep, and with it the V-cycle, stays SPD).  Measured at 1M DOF: 1.0 -> 16, 1.2 -> 14,
  // 1.4 -> 13.5 PCG iterations per step.  The prolongator keeps the classical 4/(3 rho).
  double smooth_scale = 1.4;
  // Also fold the FINEST level's two smoothing sweeps into the transfer operators (Rt, GP as on the intermediate
  // levels): the preconditioner then never passes over the fine operator itself, only over (I - w D^-1 A) P and G.
  bool fuse_fine = false;
  bool fuse_fine_down_only = false;   // with fuse_fine: only the down leg Rt_0 (the up leg stays explicit: P_0, post-smoothing sweep)
  // study knobs (scripts/micro/amg_study.cpp); the defaults are the production values
  double prolong_scale = 1.0;   // factor on the prolongator's damping 4/(3 rho)
  int prolong_steps = 1;        // Jacobi steps applied to the tentative prolongator
  double theta_decay = 0.5;     // strength threshold of level l = theta * theta_decay^l
  bool verbose = false;
  double theta_level[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // > 0: threshold of that level (study)
  double theta_coarse = 0.0;    // > 0: strength threshold of every level below the finest
  bool attach_weak = false;     // aggregate(): rows without a strong coupling join the aggregate of their largest coupling
  // A0 is the operator of global level `level_offset` of a hierarchy whose finer levels were built elsewhere (on the GPU,
  // hf_amg_gpu.hpp): thresholds, fused legs and the ownership of levels[0].A follow the global level index
  int level_offset = 0;
};

inline double level_theta(const Params& prm, int glev) {
  double t = (glev > 0 && prm.theta_coarse > 0.0) ? prm.theta_coarse : prm.theta * std::pow(prm.theta_decay, glev);
  if (glev < 8 && prm.theta_level[glev] > 0.0) t = prm.theta_level[glev];
  return t;
}

// Build from the fine operator A0 (moved in; released after the first Galerkin product).
inline bool build(Csr&& A0, const Params& prm, Hierarchy& H) {
  H.levels.clear();
  Csr A = std::move(A0);
  const double nnz0 = static_cast<double>(A.nnz());
  double nnz_sum = nnz0;
  for (int lev = 0;; ++lev) {
    const int glev = lev + prm.level_offset;
    Level L;
    std::vector<double> d = diagonal(A);
    for (double v : d)
      if (!(v > 0.0)) return false;
    L.dinv.resize(d.size());
    for (size_t i = 0; i < d.size(); ++i) L.dinv[i] = 1.0 / d[i];
    const double rho = gershgorin_rho(A, d);
    L.omega = prm.smooth_scale * 4.0 / (3.0 * rho);
    const bool last = A.nrow <= prm.coarse_size || glev + 1 >= prm.max_levels;
    if (!last) {
      std::vector<int> agg;
      const double theta_l = level_theta(prm, glev);
      const int na = aggregate(A, d, theta_l, agg, prm.attach_weak);
      if (prm.verbose) {
        int64_t none = 0, trivial = 0;
        for (int i = 0; i < A.nrow; ++i) {
          if (agg[i] < 0) { ++none; if (A.ptr[i + 1] - A.ptr[i] <= 1) ++trivial; }
        }
        std::fprintf(stderr, "[amg setup] level %d rows %d aggregates %d (ratio %.2f) without aggregate %lld (of which identity rows %lld) theta %.4f\n", glev, A.nrow, na,
                     static_cast<double>(A.nrow) / std::max(na, 1), static_cast<long long>(none), static_cast<long long>(trivial), theta_l);
      }
      if (na == 0 || na > 0.8 * A.nrow) {            // coarsening stalled: finish here
        L.A = std::move(A);
        H.levels.push_back(std::move(L));
        break;
      }
      const auto tp0 = std::chrono::steady_clock::now();
      auto lap = [&](const char* what) {
        if (prm.verbose) std::fprintf(stderr, "[amg setup] level %d %-22s %.3f s\n", glev, what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count());
      };
      L.P = smoothed_prolongator(A, d, agg, na, prm.prolong_scale * 4.0 / (3.0 * rho));
      for (int s = 1; s < prm.prolong_steps; ++s) L.P = smoothed_by_product(L.P, spgemm(A, L.P), L.dinv, prm.prolong_scale * 4.0 / (3.0 * rho));
      lap("prolongator");
      L.R = transpose(L.P);
      lap("+ transpose");
      Csr AP = spgemm(A, L.P);
      lap("+ A P");
      Csr Ac = spgemm(L.R, AP);
      lap("+ R (A P)");
      if (glev > 0 || prm.fuse_fine) {               // fused legs of the cycle (intermediate levels; the finest on request)
        const Csr Pt = smoothed_by_product(L.P, AP, L.dinv, L.omega);
        L.Rt = transpose(Pt);
        if (glev > 0 || !prm.fuse_fine_down_only) L.GP = fused_up_leg(A, L.dinv, L.omega, Pt);
      }
      lap("+ fused legs");
      if (glev > 0) L.A = std::move(A);              // level 0's operator stays with the caller
      H.levels.push_back(std::move(L));
      A = std::move(Ac);
      nnz_sum += static_cast<double>(A.nnz());
    } else {
      L.A = std::move(A);
      H.levels.push_back(std::move(L));
      break;
    }
  }
  const Csr& Ac = H.levels.back().A;
  H.coarse_n = Ac.nrow;
  H.op_complexity = nnz_sum / nnz0;
  if (H.levels.size() == 1 && prm.level_offset == 0) H.coarse_n = 0;  // no coarsening possible: plain Jacobi
  return true;                               // the dense inverse of the coarsest operator is formed on the device
}

}  // namespace amg
