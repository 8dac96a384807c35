// Study tool (CPU): convergence of PCG preconditioned with the V(1,1) cycle of amg_host.hpp on a dumped fine operator
// (tests/tools/dump_system.py), for parameter and smoother studies without a GPU.
//   g++ -O2 -std=c++17 -I heatflow_amd/csrc scripts/micro/amg_study.cpp -o /tmp/amg_study
//   /tmp/amg_study A.bin [key=value ...]   keys: theta sscale pscale coarse smoother(0 jacobi,1 l1,2 cheb2,3 jacobi x2) nu lam(0 gershgorin,1 power) over
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "amg_host.hpp"

using amg::Csr;
using Vec = std::vector<double>;

static void matvec(const Csr& A, const Vec& x, Vec& y) {
  y.assign(A.nrow, 0.0);
  for (int i = 0; i < A.nrow; ++i) {
    double s = 0.0;
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) s += A.val[k] * x[A.idx[k]];
    y[i] = s;
  }
}
static double dot(const Vec& a, const Vec& b) { double s = 0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }

struct Lev { const Csr* A; const Csr* P; const Csr* R; Vec dinv, l1inv; double omega, lam; };
static std::vector<Lev> L;
static std::vector<double> chol;  // dense Cholesky factor of the coarsest operator
static int nc = 0;
static int smoother = 0, nu = 1;
static double over = 1.0;

static int smoother_c = -1, nu_c = -1, nu_l1 = -1, nu_l2 = -1, gamma_c = 1, gamma_from = 0;
static void smooth(const Lev& l, const Vec& b, Vec& x, bool zero_guess) {
  const int n = l.A->nrow;
  const bool fine = &l == &L[0];
  const int smoother = (!fine && smoother_c >= 0) ? smoother_c : ::smoother;
  const size_t li = &l - &L[0];
  int nu = (!fine && nu_c >= 0) ? nu_c : ::nu;
  if (li == 1 && nu_l1 >= 0) nu = nu_l1;
  if (li >= 2 && nu_l2 >= 0) nu = nu_l2;
  Vec r(n), t;
  for (int s = 0; s < nu; ++s) {
    if (smoother == 2) {  // Chebyshev degree 2 on [lam/4 .. lam] hmm: interval [a, b] = [lam * 0.25, lam * 1.05]
      const double a = 0.25 * l.lam, bb = 1.05 * l.lam, theta = 0.5 * (a + bb), delta = 0.5 * (bb - a);
      // standard 3-term recurrence, 2 steps
      if (zero_guess && s == 0) { for (int i = 0; i < n; ++i) r[i] = b[i]; } else { matvec(*l.A, x, t); for (int i = 0; i < n; ++i) r[i] = b[i] - t[i]; }
      Vec d(n);
      double sigma = theta / delta, rho = 1.0 / sigma;
      for (int i = 0; i < n; ++i) d[i] = l.dinv[i] * r[i] / theta;
      for (int i = 0; i < n; ++i) x[i] = ((zero_guess && s == 0) ? 0.0 : x[i]) + d[i];
      for (int k = 1; k < 2; ++k) {
        matvec(*l.A, d, t);
        for (int i = 0; i < n; ++i) r[i] -= t[i];
        const double rho1 = 1.0 / (2.0 * sigma - rho);
        for (int i = 0; i < n; ++i) d[i] = rho1 * rho * d[i] + 2.0 * rho1 / delta * l.dinv[i] * r[i];
        rho = rho1;
        for (int i = 0; i < n; ++i) x[i] += d[i];
      }
      continue;
    }
    const Vec& di = smoother == 1 ? l.l1inv : l.dinv;
    const double w = smoother == 1 ? 1.0 : l.omega;
    if (zero_guess && s == 0) { for (int i = 0; i < n; ++i) x[i] = w * di[i] * b[i]; continue; }
    matvec(*l.A, x, t);
    for (int i = 0; i < n; ++i) x[i] += w * di[i] * (b[i] - t[i]);
  }
}

static void coarse_solve(const Vec& b, Vec& x) {
  x = b;
  for (int i = 0; i < nc; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= chol[(size_t)i * nc + k] * x[k]; x[i] = s / chol[(size_t)i * nc + i]; }
  for (int i = nc - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < nc; ++k) s -= chol[(size_t)k * nc + i] * x[k]; x[i] = s / chol[(size_t)i * nc + i]; }
}

static void vcycle(size_t lev, const Vec& b, Vec& x) {
  if (lev + 1 == L.size()) { if (nc > 0) coarse_solve(b, x); else { x.assign(b.size(), 0.0); smooth(L[lev], b, x, true); } return; }
  const Lev& l = L[lev];
  const int n = l.A->nrow;
  x.assign(n, 0.0);
  smooth(l, b, x, true);
  Vec t, r(n), bc, ec, e;
  matvec(*l.A, x, t);
  for (int i = 0; i < n; ++i) r[i] = b[i] - t[i];
  matvec(*l.R, r, bc);
  vcycle(lev + 1, bc, ec);
  if (gamma_c > 1 && lev + 2 < L.size() && static_cast<int>(lev) >= gamma_from) {   // W-cycle: second visit with the residual of the first
    Vec t2, bc2(bc.size()), ec2;
    matvec(*L[lev + 1].A, ec, t2);
    for (size_t i = 0; i < bc.size(); ++i) bc2[i] = bc[i] - t2[i];
    vcycle(lev + 1, bc2, ec2);
    for (size_t i = 0; i < ec.size(); ++i) ec[i] += ec2[i];
  }
  matvec(*l.P, ec, e);
  for (int i = 0; i < n; ++i) x[i] += over * e[i];
  smooth(l, b, x, false);
}

static double power_lambda(const Csr& A, const Vec& dinv) {
  std::mt19937_64 rng(3); std::uniform_real_distribution<double> U(-1, 1);
  Vec v(A.nrow), t; for (auto& q : v) q = U(rng);
  double lam = 0;
  for (int it = 0; it < 30; ++it) {
    matvec(A, v, t);
    for (int i = 0; i < A.nrow; ++i) t[i] *= dinv[i];
    lam = std::sqrt(dot(t, t) / dot(v, v));
    const double s = 1.0 / std::sqrt(dot(t, t));
    for (int i = 0; i < A.nrow; ++i) v[i] = t[i] * s;
  }
  return lam;
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  std::map<std::string, double> kv{{"theta", 0.08}, {"sscale", 1.4}, {"pscale", 1.0}, {"coarse", 2500}, {"smoother", 0}, {"nu", 1}, {"lam", 0}, {"over", 1.0}, {"tol", 1e-10}, {"psteps", 1}, {"thdecay", 0.5}};
  for (int a = 2; a < argc; ++a) { char* eq = std::strchr(argv[a], '='); if (eq) kv[std::string(argv[a], eq - argv[a])] = std::atof(eq + 1); }
  FILE* f = std::fopen(argv[1], "rb");
  int32_t n; int64_t nnz;
  if (!f || std::fread(&n, 4, 1, f) != 1 || std::fread(&nnz, 8, 1, f) != 1) return 2;
  Csr A0; A0.nrow = A0.ncol = n; A0.ptr.resize(n + 1); A0.idx.resize(nnz); A0.val.resize(nnz);
  if (std::fread(A0.ptr.data(), 4, n + 1, f) != (size_t)n + 1 || std::fread(A0.idx.data(), 4, nnz, f) != (size_t)nnz || std::fread(A0.val.data(), 8, nnz, f) != (size_t)nnz) return 3;
  std::fclose(f);
  amg::Params prm; prm.theta = kv["theta"]; prm.smooth_scale = kv["sscale"]; prm.coarse_size = (int)kv["coarse"];
  prm.prolong_scale = kv["pscale"]; prm.prolong_steps = (int)kv["psteps"]; prm.theta_decay = kv["thdecay"]; prm.verbose = true; for (int l = 0; l < 8; ++l) { const std::string k = "th" + std::to_string(l); if (kv.count(k)) prm.theta_level[l] = kv[k]; } if (kv.count("thetac")) prm.theta_coarse = kv["thetac"]; prm.attach_weak = kv.count("weak") && kv["weak"] > 0;
  smoother = (int)kv["smoother"]; nu = (int)kv["nu"]; over = kv["over"];
  if (kv.count("sc")) smoother_c = (int)kv["sc"];
  if (kv.count("nuc")) nu_c = (int)kv["nuc"];
  if (kv.count("nul1")) nu_l1 = (int)kv["nul1"];
  if (kv.count("nul2")) nu_l2 = (int)kv["nul2"];
  if (kv.count("gamma")) gamma_c = (int)kv["gamma"];
  if (kv.count("gfrom")) gamma_from = (int)kv["gfrom"];
  amg::Hierarchy H;
  Csr Acopy = A0;
  auto t0 = std::chrono::steady_clock::now();
  if (!amg::build(std::move(Acopy), prm, H)) { std::printf("build failed\n"); return 4; }
  const double ts = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("levels:");
  for (auto& l : H.levels) std::printf(" %zu", l.dinv.size());
  std::printf("  opc %.3f  setup %.2fs  nnz(P0) %lld nnz(A1) %lld\n", H.op_complexity, ts, (long long)H.levels[0].P.nnz(), H.levels.size() > 1 ? (long long)H.levels[1].A.nnz() : 0LL);
  for (size_t l = 0; l < H.levels.size(); ++l) {
    Lev q; q.A = l == 0 ? &A0 : &H.levels[l].A; q.P = &H.levels[l].P; q.R = &H.levels[l].R; q.dinv = H.levels[l].dinv; q.omega = H.levels[l].omega;
    q.l1inv.resize(q.A->nrow);
    for (int i = 0; i < q.A->nrow; ++i) { double s = 0; for (int k = q.A->ptr[i]; k < q.A->ptr[i + 1]; ++k) s += std::fabs(q.A->val[k]); q.l1inv[i] = 1.0 / s; }
    q.lam = 0;
    if (kv["lam"] > 0 || smoother == 2 || smoother_c == 2) { q.lam = power_lambda(*q.A, q.dinv); if (kv["lam"] > 0) q.omega = kv["sscale"] / 1.4 * 4.0 / (3.0 * q.lam) * 1.0; std::printf("  level %zu lambda_max(D^-1 A) ~ %.4f  omega %.4f (gershgorin-based %.4f)\n", l, q.lam, q.omega, H.levels[l].omega); }
    L.push_back(std::move(q));
  }
  const Csr& Ac = *L.back().A;
  nc = H.coarse_n > 0 ? Ac.nrow : 0;
  if (nc > 0) {
    chol.assign((size_t)nc * nc, 0.0);
    for (int i = 0; i < nc; ++i) for (int k = Ac.ptr[i]; k < Ac.ptr[i + 1]; ++k) if (Ac.idx[k] <= i) chol[(size_t)i * nc + Ac.idx[k]] = Ac.val[k];
    for (int j = 0; j < nc; ++j) {
      double d = chol[(size_t)j * nc + j];
      for (int k = 0; k < j; ++k) d -= chol[(size_t)j * nc + k] * chol[(size_t)j * nc + k];
      if (!(d > 0)) { std::printf("coarse operator not SPD\n"); return 5; }
      d = std::sqrt(d); chol[(size_t)j * nc + j] = d;
      for (int i = j + 1; i < nc; ++i) { double s = chol[(size_t)i * nc + j]; for (int k = 0; k < j; ++k) s -= chol[(size_t)i * nc + k] * chol[(size_t)j * nc + k]; chol[(size_t)i * nc + j] = s / d; }
    }
  }
  // PCG, random right-hand side, x0 = 0, stop on ||D^-1 r|| <= tol ||D^-1 b||
  std::mt19937_64 rng(7); std::uniform_real_distribution<double> U(-1, 1);
  Vec b(n), x(n, 0.0), r, z, p, Ap;
  for (auto& q : b) q = U(rng);
  r = b;
  auto dnorm = [&](const Vec& v) { double s = 0; for (int i = 0; i < n; ++i) { const double q = L[0].dinv[i] * v[i]; s += q * q; } return std::sqrt(s); };
  const double bn = dnorm(b);
  vcycle(0, r, z); p = z;
  double rz = dot(r, z);
  std::vector<double> hist{1.0};
  int it = 0;
  for (; it < 100; ++it) {
    matvec(A0, p, Ap);
    const double alpha = rz / dot(p, Ap);
    for (int i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; }
    const double rel = dnorm(r) / bn;
    hist.push_back(rel);
    if (rel <= kv["tol"]) { ++it; break; }
    vcycle(0, r, z);
    const double rz1 = dot(r, z);
    const double beta = rz1 / rz; rz = rz1;
    for (int i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
  }
  const size_t m = hist.size();
  const double fac = m > 6 ? std::pow(hist[m - 1] / hist[m - 6], 0.2) : 0.0;
  std::printf("iterations %d to %.0e   asymptotic factor %.3f   (rel after 5: %.2e, after 10: %.2e)\n", it, kv["tol"], fac, m > 5 ? hist[5] : 0.0, m > 10 ? hist[10] : 0.0);
  return 0;
}
