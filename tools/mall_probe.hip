// Does a working set that fits the 256 MiB Infinity Cache run faster when re-used back to back?
// hipcc -O3 --offload-arch=gfx950 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <bool NT> __global__ __launch_bounds__(256) void k_scale(double2* p, u64 n) {
  const u64 i0 = ((u64)blockIdx.x * 2) * 256 + threadIdx.x;
  double2 v[2];
  for (int r = 0; r < 2; ++r) {
    const double2* q = p + i0 + r * 256;
    if (NT) { v[r].x = __builtin_nontemporal_load(&q->x); v[r].y = __builtin_nontemporal_load(&q->y); } else v[r] = *q;
  }
  for (int r = 0; r < 2; ++r) {
    double2 w = make_double2(v[r].x * 1.0000001, v[r].y * 0.9999999);
    double2* q = p + i0 + r * 256;
    if (NT) { __builtin_nontemporal_store(w.x, &q->x); __builtin_nontemporal_store(w.y, &q->y); } else *q = w;
  }
}
int main() {
  const u64 N = 1ull << 28;   // 4 GiB
  double2* p; CK(hipMalloc(&p, N * 16)); CK(hipMemset(p, 0, N * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nt = 0; nt < 2; ++nt)
    for (int lg = 20; lg <= 28; lg += 1) {      // region of 2^lg amplitudes (16 MiB .. 4 GiB)
      const u64 n = 1ull << lg; const int reps = (int)((1ull << 30) / n) + 4;
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) {
        if (nt) hipLaunchKernelGGL(k_scale<true>, dim3(n / 512), dim3(256), 0, 0, p, n);
        else hipLaunchKernelGGL(k_scale<false>, dim3(n / 512), dim3(256), 0, 0, p, n);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s region %5llu MiB x %3d reps: %.3f ms/pass  %.0f GB/s (r+w)\n", nt ? "NT " : "def", (n * 16) >> 20, reps, ms / reps, 32.0 * n / (ms / reps * 1e-3) / 1e9);
    }
  return 0;
}
