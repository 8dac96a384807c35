// Micro-benchmark: phases of the multigrid set-up (heatflow_amd/csrc/amg_host.hpp) on a 1M-row model operator.
//   g++ -O3 -std=c++17 -I ../../heatflow_amd/csrc amg_setup_time.cpp -o amg_setup_time && ./amg_setup_time
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double T0;
#define TICK(name) do { double t_ = now(); std::printf("  %-28s %.3f\n", name, t_ - T0); T0 = t_; } while (0)
#include "amg_host.hpp"
using amg::Csr;
int main() {
  const int nx = 1020, ny = 1020, n = nx * ny;
  Csr A; A.nrow = A.ncol = n; A.ptr.assign(n + 1, 0);
  for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
    const int r = j * nx + i;
    auto add = [&](int c, double v) { A.idx.push_back(c); A.val.push_back(v); };
    if (j > 0) { if (i > 0) add(r - nx - 1, -0.5); add(r - nx, -1.0); }
    if (i > 0) add(r - 1, -1.0);
    add(r, 6.3);
    if (i < nx - 1) add(r + 1, -1.0);
    if (j < ny - 1) { add(r + nx, -1.0); if (i < nx - 1) add(r + nx + 1, -0.5); }
    A.ptr[r + 1] = (int)A.idx.size();
  }
  // replicate build() with timers
  amg::Params prm; amg::Hierarchy H;
  for (int lev = 0;; ++lev) {
    T0 = now();
    amg::Level L;
    std::vector<double> d = amg::diagonal(A);
    L.dinv.resize(d.size()); for (size_t i = 0; i < d.size(); ++i) L.dinv[i] = 1.0 / d[i];
    const double rho = amg::gershgorin_rho(A, d);
    L.omega = prm.smooth_scale * 4.0 / (3.0 * rho);
    std::printf("level %d rows %d nnz %lld\n", lev, A.nrow, (long long)A.nnz());
    TICK("diag+rho");
    if (A.nrow <= prm.coarse_size) break;
    std::vector<int> agg; const int na = amg::aggregate(A, d, prm.theta * std::pow(0.5, lev), agg); TICK("aggregate");
    L.P = amg::smoothed_prolongator(A, d, agg, na, 4.0 / (3.0 * rho)); TICK("prolongator");
    L.R = amg::transpose(L.P); TICK("transpose");
    Csr AP = amg::spgemm(A, L.P); TICK("A*P");
    Csr Ac = amg::spgemm(L.R, AP); TICK("R*AP");
    if (lev > 0) {
       
      const Csr Pt = amg::smoothed_by_product(L.P, AP, L.dinv, L.omega); TICK("Pt merge");
      L.Rt = amg::transpose(Pt); TICK("transpose Pt");
      L.GP = amg::fused_up_leg(A, L.dinv, L.omega, Pt); TICK("GP");
    }
    A = std::move(Ac); TICK("move");
  }
}
