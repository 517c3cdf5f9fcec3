// How many 256-thread workgroups with L bytes of LDS are resident on one CU of gfx950 at once?  Every workgroup
// bumps a per-CU counter (CU identity from HW_ID / XCC_ID), keeps the maximum, lingers 30 us, and leaves.
//   hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o tools/occupancy_probe && tools/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
extern __shared__ char dyn[];
__global__ void probe(int* cnt, int* mx, int lds_bytes) {
  if (threadIdx.x == 0) {
    dyn[lds_bytes - 1] = 1;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = ((xcc & 15) << 8) | (se << 5) | (sh << 4) | cu;
    const int now = atomicAdd(&cnt[key], 1) + 1;
    atomicMax(&mx[key], now);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 3000) {}
    atomicSub(&cnt[key], 1);
  }
  __syncthreads();
}
int main() {
  int *cnt, *mx;
  (void)hipMalloc(&cnt, 4096 * 4); (void)hipMalloc(&mx, 4096 * 4);
  static int h[4096];
  for (int lds : {16384, 31744, 32000, 32256, 32768, 33280, 40960}) {
    (void)hipMemset(cnt, 0, 4096 * 4); (void)hipMemset(mx, 0, 4096 * 4);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    probe<<<256 * 16, 256, lds>>>(cnt, mx, lds);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, mx, sizeof h, hipMemcpyDeviceToHost);
    std::map<int, int> hist; int cus = 0;
    for (int i = 0; i < 4096; ++i) if (h[i]) { ++hist[h[i]]; ++cus; }
    printf("LDS %6d B, 256 threads: %d CUs seen; max resident workgroups per CU:", lds, cus);
    for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
    printf("\n");
  }
  return 0;
}
