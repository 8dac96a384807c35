// Micro-benchmark: read bandwidth of a buffer the size of the C3 operator (12 B x 7.27 M entries = 87 MB, fits the
// 256 MiB Infinity Cache) and of larger ones, with 16-byte loads, 512-thread workgroups, 1024 / 2048 workgroups.
//   hipcc -O3 --offload-arch=gfx950 -o stream_read stream_read.hip && ./stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_read(size_t n16, const double2* __restrict__ a, double* __restrict__ out) {
  double s = 0.0;
  size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 512;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
    s += (v0.x + v0.y) + (v1.x + v1.y) + (v2.x + v2.y) + (v3.x + v3.y);
  }
  for (; i < n16; i += stride) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double* out; CK(hipMalloc(&out, 8));
  for (size_t mb : {32, 87, 133, 200, 350, 1000}) {
    const size_t bytes = mb * 1000000, n16 = bytes / 16;
    double2* a; CK(hipMalloc(&a, n16 * 16)); CK(hipMemset(a, 0, n16 * 16));
    for (int grid : {1024, 2048}) {
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(512), 0, st, n16, a, out);
      CK(hipEventRecord(e0, st));
      const int reps = 100;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(512), 0, st, n16, a, out);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%5zu MB grid %4d: %7.2f us per pass  %6.2f TB/s\n", mb, grid, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e12);
    }
    CK(hipFree(a));
  }
  return 0;
}
