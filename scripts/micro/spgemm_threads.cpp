// Micro-benchmark: Gustavson SpGEMM A*P of the multigrid set-up on T host threads (row ranges, private accumulators,
// concatenated afterwards) against the serial routine of amg_host.hpp, on a 1M-row model operator.
//   g++ -O3 -std=c++17 -pthread -I ../../heatflow_amd/csrc spgemm_threads.cpp -o spgemm_threads && ./spgemm_threads
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include "amg_host.hpp"
using amg::Csr;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static Csr spgemm_mt(const Csr& A, const Csr& B, int T) {
  std::vector<Csr> part(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      const int r0 = static_cast<int>(static_cast<int64_t>(A.nrow) * t / T), r1 = static_cast<int>(static_cast<int64_t>(A.nrow) * (t + 1) / T);
      Csr& C = part[t];
      C.ptr.assign(1, 0);
      std::vector<double> acc(B.ncol, 0.0);
      std::vector<int> mark(B.ncol, -1), cols;
      C.idx.reserve(static_cast<size_t>(A.ptr[r1] - A.ptr[r0]) * 2);
      C.val.reserve(static_cast<size_t>(A.ptr[r1] - A.ptr[r0]) * 2);
      for (int i = r0; i < r1; ++i) {
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
        C.ptr.push_back(static_cast<int>(C.idx.size()));
      }
    });
  for (auto& x : th) x.join();
  Csr C; C.nrow = A.nrow; C.ncol = B.ncol; C.ptr.assign(1, 0);
  size_t tot = 0; for (auto& p : part) tot += p.idx.size();
  C.idx.reserve(tot); C.val.reserve(tot); C.ptr.reserve(A.nrow + 1);
  for (auto& p : part) {
    const int off = static_cast<int>(C.idx.size());
    C.idx.insert(C.idx.end(), p.idx.begin(), p.idx.end());
    C.val.insert(C.val.end(), p.val.begin(), p.val.end());
    for (size_t r = 1; r < p.ptr.size(); ++r) C.ptr.push_back(off + p.ptr[r]);
  }
  return C;
}

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
  std::vector<double> d = amg::diagonal(A);
  std::vector<int> agg; const int na = amg::aggregate(A, d, 0.08, agg);
  Csr P = amg::smoothed_prolongator(A, d, agg, na, 4.0 / (3.0 * amg::gershgorin_rho(A, d)));
  std::printf("hardware threads %u, n %d, aggregates %d\n", std::thread::hardware_concurrency(), n, na);
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); Csr C0 = amg::spgemm(A, P); double t1 = now();
    std::printf("serial          %.3f s  nnz %lld\n", t1 - t0, (long long)C0.nnz());
    for (int T : {2, 4, 8, 16}) {
      t0 = now(); Csr C = spgemm_mt(A, P, T); t1 = now();
      bool same = C.idx == C0.idx && C.val == C0.val && C.ptr == C0.ptr;
      std::printf("%2d threads      %.3f s  identical %d\n", T, t1 - t0, (int)same);
    }
  }
  return 0;
}
