// What does __builtin_readcyclecounter (s_memtime) count on gfx950?  One wave spins for 2 ms of the constant 100 MHz
// real-time counter (s_memrealtime) -- alone, and again while a second kernel loads the other CUs with f64 work --
// and reports s_memtime ticks per microsecond.   hipcc --offload-arch=gfx950 -O3 tools/cycle_calib.hip -o tools/cycle_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, unsigned long long real_ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  unsigned long long r = r0;
  while (r - r0 < real_ticks) r = __builtin_amdgcn_s_memrealtime();
  out[0] = __builtin_readcyclecounter() - c0;
  out[1] = r - r0;
}
__global__ void burn(double* p, int iters) {
  double a = p[threadIdx.x], b = 1.000001;
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, b, 1e-9);
  p[threadIdx.x + blockIdx.x * blockDim.x] = a;
}
int main() {
  unsigned long long* d; double* w;
  hipMalloc(&d, 16); hipMalloc(&w, sizeof(double) * 256 * 4096);
  hipMemset(w, 0, sizeof(double) * 256 * 4096);
  unsigned long long h[2];
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  for (int load = 0; load < 2; ++load) {
    if (load) burn<<<4096, 256, 0, s2>>>(w, 4000000);
    spin<<<1, 64, 0, s1>>>(d, 200000);
    hipStreamSynchronize(s1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%s: %llu s_memtime ticks in %llu real ticks (100 MHz) -> %.1f ticks/us\n", load ? "loaded" : "idle", h[0], h[1], h[0] / (h[1] / 100.0));
    hipDeviceSynchronize();
  }
  return 0;
}
