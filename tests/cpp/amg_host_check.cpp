// CPU check of the multigrid set-up algebra in heatflow_amd/csrc/amg_host.hpp (built and run by
// tests/test_amg_host_cpu.py): Galerkin products, transposes, and the fused legs of the intermediate levels
// against the explicit V(1,1) steps they replace, on an SPD model operator (5-point Laplacian + mass, one
// Dirichlet-like row with unit diagonal and no couplings).
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "amg_host.hpp"

using amg::Csr;

static std::vector<double> matvec(const Csr& A, const std::vector<double>& x) {
  std::vector<double> y(A.nrow, 0.0);
  for (int i = 0; i < A.nrow; ++i)
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) y[i] += A.val[k] * x[A.idx[k]];
  return y;
}

static double maxdiff(const std::vector<double>& a, const std::vector<double>& b) {
  double m = 0.0, s = 0.0;
  for (size_t i = 0; i < a.size(); ++i) { m = std::fmax(m, std::fabs(a[i] - b[i])); s = std::fmax(s, std::fabs(a[i])); }
  return m / std::fmax(s, 1e-300);
}

int main(int argc, char** argv) {
  const int nx = argc > 1 ? std::atoi(argv[1]) : 96, ny = argc > 2 ? std::atoi(argv[2]) : 80;
  const int n = nx * ny;
  Csr A;
  A.nrow = A.ncol = n;
  A.ptr.assign(n + 1, 0);
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      const int r = j * nx + i;
      if (r == 7) { A.idx.push_back(r); A.val.push_back(1.0); A.ptr[r + 1] = (int)A.idx.size(); continue; }   // eliminated row
      const double kx = (i < nx / 2) ? 1.0 : 40.0;           // coefficient jump
      auto add = [&](int c, double v) { if (c != 7) { A.idx.push_back(c); A.val.push_back(v); } };
      double diag = 0.3;
      if (j > 0) { add(r - nx, -1.0); }
      if (i > 0) { add(r - 1, -kx); }
      diag += (j > 0 ? 1.0 : 0.0) + (j < ny - 1 ? 1.0 : 0.0) + (i > 0 ? kx : 0.0) + (i < nx - 1 ? ((i + 1 < nx / 2) ? 1.0 : 40.0) : 0.0);
      // keep symmetry: coupling (i,i+1) uses the coefficient of the right cell
      add(r, diag);
      if (i < nx - 1) { add(r + 1, -(((i + 1) < nx / 2) ? 1.0 : 40.0)); }
      if (j < ny - 1) { add(r + nx, -1.0); }
      A.ptr[r + 1] = (int)A.idx.size();
    }
  const Csr A0 = A;
  amg::Hierarchy H;
  amg::Params prm;
  prm.coarse_size = 40;
  if (!amg::build(std::move(A), prm, H)) { std::printf("FAIL build\n"); return 1; }
  std::printf("levels %zu:", H.levels.size());
  for (auto& L : H.levels) std::printf(" %zu", L.dinv.size());
  std::printf("  opc %.3f\n", H.op_complexity);
  if (H.levels.size() < 3) { std::printf("FAIL: want at least one intermediate level\n"); return 1; }
  std::mt19937_64 rng(11);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  double worst = 0.0;
  for (size_t l = 0; l + 1 < H.levels.size(); ++l) {
    const amg::Level& L = H.levels[l];
    const Csr& Al = l == 0 ? A0 : L.A;
    const Csr& Ac = H.levels[l + 1].A;
    const int nl = Al.nrow, nc = Ac.nrow;
    // R = P^T
    const Csr Pt = amg::transpose(L.P);
    if (Pt.idx != L.R.idx || Pt.val != L.R.val) { std::printf("FAIL: R != P^T on level %zu\n", l); return 1; }
    // Galerkin: Ac x = R A P x
    std::vector<double> xc(nc);
    for (auto& v : xc) v = U(rng);
    const double g = maxdiff(matvec(Ac, xc), matvec(L.R, matvec(Al, matvec(L.P, xc))));
    worst = std::fmax(worst, g);
    std::printf("level %zu: Galerkin %.2e", l, g);
    if (l >= 1) {
      // down leg: Rt b == R (b - A w D^-1 b)
      std::vector<double> b(nl), e(nc);
      for (auto& v : b) v = U(rng);
      for (auto& v : e) v = U(rng);
      std::vector<double> x0(nl);
      for (int i = 0; i < nl; ++i) x0[i] = L.omega * L.dinv[i] * b[i];
      std::vector<double> Ax0 = matvec(Al, x0), res(nl);
      for (int i = 0; i < nl; ++i) res[i] = b[i] - Ax0[i];
      const double d1 = maxdiff(matvec(L.Rt, b), matvec(L.R, res));
      // up leg: GP [b; e] == x1 + w D^-1 (b - A x1),  x1 = x0 + P e
      std::vector<double> x1 = matvec(L.P, e);
      for (int i = 0; i < nl; ++i) x1[i] += x0[i];
      std::vector<double> Ax1 = matvec(Al, x1), x2(nl), cat(b);
      for (int i = 0; i < nl; ++i) x2[i] = x1[i] + L.omega * L.dinv[i] * (b[i] - Ax1[i]);
      cat.insert(cat.end(), e.begin(), e.end());
      const double d2 = maxdiff(matvec(L.GP, cat), x2);
      worst = std::fmax(worst, std::fmax(d1, d2));
      std::printf("  down leg %.2e  up leg %.2e", d1, d2);
      if (L.GP.ncol != nl + nc || L.Rt.nrow != nc || L.Rt.ncol != nl) { std::printf("\nFAIL: leg shapes\n"); return 1; }
    } else if (L.Rt.nrow != 0 || L.GP.nrow != 0) {
      std::printf("\nFAIL: the finest level must not carry fused legs\n");
      return 1;
    }
    std::printf("\n");
    // rows without an aggregate (the eliminated row) get no coarse correction
    if (l == 0 && L.P.ptr[8] != L.P.ptr[7]) { std::printf("FAIL: eliminated row has prolongator entries\n"); return 1; }
    // smoother stays convergent: w * lambda_max(D^-1 A) < 2 by Gershgorin
    if (!(L.omega * amg::gershgorin_rho(Al, amg::diagonal(Al)) < 2.0)) { std::printf("FAIL: damping too large\n"); return 1; }
  }
  if (!(worst < 1e-12)) { std::printf("FAIL: worst relative deviation %.3e\n", worst); return 1; }
  std::printf("OK worst %.2e\n", worst);
  return 0;
}
