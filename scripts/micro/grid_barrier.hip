// Micro-benchmark: cost of a software grid barrier (atomic counter + bounded spin + agent-scope fences)
// between dependent stages inside one kernel, against the launch-to-launch cost of separate kernels.
//   hipcc -O3 --offload-arch=gfx950 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    int good = 0;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
    if (!good) *err = 1;
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

// each stage: y[i] = x[(i*7+1) % n] + 1 over n entries (reads what other workgroups wrote in the previous stage)
__global__ __launch_bounds__(256) void k_staged(int n, int stages, double* a, double* b, unsigned* counter, int* err) {
  double* x = a; double* y = b;
  for (int s = 0; s < stages; ++s) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = x[(i * 7 + 1) % n] + 1.0;
    if (s + 1 < stages) { if (!grid_barrier(counter, (s + 1) * gridDim.x, err)) return; }
    double* t = x; x = y; y = t;
  }
}

__global__ __launch_bounds__(256) void k_one(int n, const double* x, double* y) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = x[(i * 7 + 1) % n] + 1.0;
}

int main() {
  const int n = 16384, stages = 8, reps = 200;
  double *a, *b; unsigned* counter; int* err;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(err, 0, 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {16, 32, 64, 128, 256}) {
    // warm-up + correctness
    CK(hipMemsetAsync(counter, 0, 4, st));
    hipLaunchKernelGGL(k_staged, dim3(grid), dim3(256), 0, st, n, stages, a, b, counter, err);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(counter, 0, 4, st));
      hipLaunchKernelGGL(k_staged, dim3(grid), dim3(256), 0, st, n, stages, a, b, counter, err);
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
      for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, st, n, (s & 1) ? b : a, (s & 1) ? a : b);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms2 = 0; CK(hipEventElapsedTime(&ms2, e0, e1));
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<double> h(n); CK(hipMemcpy(h.data(), (stages & 1) ? b : a, n * 8, hipMemcpyDeviceToHost));
    printf("grid %3d: fused kernel (8 stages, 7 barriers, + memset) %7.2f us  | 8 separate launches %7.2f us | err %d  check %.0f\n", grid,
           1e3 * ms / reps, 1e3 * ms2 / reps, herr, h[5]);
  }
  return 0;
}
