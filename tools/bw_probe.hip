// Bandwidth probe: variants of the in-place 1-qubit butterfly and a plain copy, one process,
// interleaved rounds (guide 5.4 rule 24).   hipcc -O3 --offload-arch=gfx950 -o bw_probe bw_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>

typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x); }
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 c) {
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}
struct U2 { double2 u[4]; };

template <bool NT> __device__ __forceinline__ double2 ld(const double2* p) {
  if (NT) {
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
  return *p;
}
template <bool NT> __device__ __forceinline__ void st(double2* p, double2 v) {
  if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
  else *p = v;
}

// one-shot: block handles ITEMS*BLOCK consecutive pairs
template <int ITEMS, int BLOCK, bool NT, int SWZ, bool SNT = NT>
__global__ __launch_bounds__(BLOCK) void k_1q(double2* psi, int q, u64 npairs, U2 U) {
  u64 bid = blockIdx.x;
  if (SWZ) { // XCD-aware: blocks b, b+8, .. share an XCD; give each XCD a contiguous range
    const u64 nb = gridDim.x; const u64 per = nb / 8;
    bid = (bid % 8) * per + bid / 8;
  }
  const u64 first = bid * ITEMS * BLOCK + threadIdx.x;
  u64 i0[ITEMS]; double2 a[ITEMS], b[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    u64 c = first + (u64)r * BLOCK;
    i0[r] = ((c >> q) << (q + 1)) | (c & ((1ull << q) - 1));
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) { a[r] = ld<NT>(psi + i0[r]); b[r] = ld<NT>(psi + i0[r] + (1ull << q)); }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    st<SNT>(psi + i0[r], cfma(U.u[1], b[r], cmul(U.u[0], a[r])));
    st<SNT>(psi + i0[r] + (1ull << q), cfma(U.u[3], b[r], cmul(U.u[2], a[r])));
  }
}

// persistent grid-stride
template <int ITEMS, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_1q_persist(double2* psi, int q, u64 npairs, U2 U) {
  const u64 per_iter = (u64)ITEMS * BLOCK;
  for (u64 base = (u64)blockIdx.x * per_iter; base < npairs; base += (u64)gridDim.x * per_iter) {
    u64 i0[ITEMS]; double2 a[ITEMS], b[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      u64 c = base + threadIdx.x + (u64)r * BLOCK;
      i0[r] = ((c >> q) << (q + 1)) | (c & ((1ull << q) - 1));
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) { a[r] = ld<NT>(psi + i0[r]); b[r] = ld<NT>(psi + i0[r] + (1ull << q)); }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      st<NT>(psi + i0[r], cfma(U.u[1], b[r], cmul(U.u[0], a[r])));
      st<NT>(psi + i0[r] + (1ull << q), cfma(U.u[3], b[r], cmul(U.u[2], a[r])));
    }
  }
}

// shuffle variant for q < 6: lane owns one amplitude per item, partner by ds_bpermute
template <int ITEMS, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_1q_shfl(double2* psi, int q, u64 namps, U2 U) {
  const u64 first = (u64)blockIdx.x * ITEMS * BLOCK + threadIdx.x;
  const int bit = (threadIdx.x >> q) & 1;
  const double2 cs = bit ? U.u[3] : U.u[0];
  const double2 co = bit ? U.u[2] : U.u[1];
  double2 x[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) x[r] = ld<NT>(psi + first + (u64)r * BLOCK);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    double2 p;
    p.x = __shfl_xor(x[r].x, 1 << q, 64);
    p.y = __shfl_xor(x[r].y, 1 << q, 64);
    st<NT>(psi + first + (u64)r * BLOCK, cfma(co, p, cmul(cs, x[r])));
  }
}

template <int ITEMS, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_copy(double2* dst, const double2* src, u64 n) {
  const u64 first = (u64)blockIdx.x * ITEMS * BLOCK + threadIdx.x;
  double2 x[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) x[r] = ld<NT>(src + first + (u64)r * BLOCK);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) st<NT>(dst + first + (u64)r * BLOCK, x[r]);
}

// in-place scale (read+write same address, all amplitudes)
template <int ITEMS, int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_scale(double2* p, u64 n, double2 s) {
  const u64 first = (u64)blockIdx.x * ITEMS * BLOCK + threadIdx.x;
  double2 x[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) x[r] = ld<NT>(p + first + (u64)r * BLOCK);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) st<NT>(p + first + (u64)r * BLOCK, cmul(s, x[r]));
}

__global__ void k_init(double2* p, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
    p[i] = make_double2(1e-5 * (double)(i & 1023), -1e-5 * (double)((i >> 3) & 511));
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 30;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const u64 N = 1ull << n;
  double2 *psi, *tmp;
  CK(hipMalloc(&psi, N * 16));
  CK(hipMalloc(&tmp, N * 8));  // half-size copy target
  hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, psi, N);
  CK(hipDeviceSynchronize());
  const double s = 0.70710678118654752440;
  U2 H = {{{s, 0}, {s, 0}, {s, 0}, {-s, 0}}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  struct Var { std::string name; double bytes; std::function<void()> run; std::vector<float> t; };
  std::vector<Var> vars;
  const u64 P = N / 2;
  auto add = [&](std::string name, double bytes, std::function<void()> f) { vars.push_back({name, bytes, f, {}}); };

  for (int q : {3, 6, 8, 11, 14, 20, 25, 29}) {
    char nm[64];
#define ADD1Q(I, LNT, SW, SNT, tag) snprintf(nm, 64, "1q q=%d " tag, q); \
    add(nm, 32.0 * N, [=] { hipLaunchKernelGGL((k_1q<I, 256, LNT, SW, SNT>), dim3(P / (256 * I)), dim3(256), 0, 0, psi, q, P, H); });
    ADD1Q(4, false, 0, false, "I4")
    ADD1Q(2, false, 0, false, "I2")
    ADD1Q(4, true, 0, true, "I4 NT")
    ADD1Q(2, true, 0, true, "I2 NT")
    ADD1Q(1, true, 0, true, "I1 NT")
    ADD1Q(4, true, 0, false, "I4 ldNT")
    ADD1Q(4, false, 0, true, "I4 stNT")
    ADD1Q(4, true, 1, true, "I4 NT SWZ")
    ADD1Q(2, true, 1, true, "I2 NT SWZ")
    ADD1Q(8, true, 0, true, "I8 NT")
  }
  for (int q : {0, 1, 2, 5}) {
    char nm[64];
#define ADDSH(I, NT, tag) snprintf(nm, 64, "1q q=%d shfl " tag, q); \
    add(nm, 32.0 * N, [=] { hipLaunchKernelGGL((k_1q_shfl<I, 256, NT>), dim3(N / (256 * I)), dim3(256), 0, 0, psi, q, N, H); });
    ADDSH(2, false, "I2")
    ADDSH(4, false, "I4")
    ADDSH(2, true, "I2 NT")
    ADDSH(4, true, "I4 NT")
    ADDSH(8, true, "I8 NT")
  }
  add("copy half->tmp I2 NT", 16.0 * N, [=] { hipLaunchKernelGGL((k_copy<2, 256, true>), dim3(P / 512), dim3(256), 0, 0, tmp, psi, P); });
  add("scale inplace I4 NT", 32.0 * N, [=] { hipLaunchKernelGGL((k_scale<4, 256, true>), dim3(N / 1024), dim3(256), 0, 0, psi, N, make_double2(1.0, 0.0)); });
  add("scale inplace I2 NT", 32.0 * N, [=] { hipLaunchKernelGGL((k_scale<2, 256, true>), dim3(N / 512), dim3(256), 0, 0, psi, N, make_double2(1.0, 0.0)); });
  add("copy half->tmp I4", 16.0 * N, [=] { hipLaunchKernelGGL((k_copy<4, 256, false>), dim3(P / 1024), dim3(256), 0, 0, tmp, psi, P); });
  add("copy half->tmp I8", 16.0 * N, [=] { hipLaunchKernelGGL((k_copy<8, 256, false>), dim3(P / 2048), dim3(256), 0, 0, tmp, psi, P); });
  add("copy half->tmp I4 NT", 16.0 * N, [=] { hipLaunchKernelGGL((k_copy<4, 256, true>), dim3(P / 1024), dim3(256), 0, 0, tmp, psi, P); });
  add("hipMemcpyDtoD half", 16.0 * N, [=] { CK(hipMemcpyAsync(tmp, psi, P * 16, hipMemcpyDeviceToDevice, 0)); });
  add("scale inplace I4", 32.0 * N, [=] { hipLaunchKernelGGL((k_scale<4, 256, false>), dim3(N / 1024), dim3(256), 0, 0, psi, N, make_double2(1.0, 0.0)); });
  add("scale inplace I8", 32.0 * N, [=] { hipLaunchKernelGGL((k_scale<8, 256, false>), dim3(N / 2048), dim3(256), 0, 0, psi, N, make_double2(1.0, 0.0)); });
  add("scale inplace I8 NT", 32.0 * N, [=] { hipLaunchKernelGGL((k_scale<8, 256, true>), dim3(N / 2048), dim3(256), 0, 0, psi, N, make_double2(1.0, 0.0)); });

  for (auto& v : vars) { v.run(); }  // warm
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars) {
      CK(hipEventRecord(e0, 0)); v.run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.t.push_back(ms);
    }
  printf("n=%d rounds=%d\n%-28s %9s %9s %9s %7s\n", n, rounds, "variant", "med ms", "min ms", "GB/s", "frac8T");
  for (auto& v : vars) {
    std::sort(v.t.begin(), v.t.end());
    float med = v.t[v.t.size() / 2], mn = v.t[0];
    double gbs = v.bytes / (med * 1e-3) / 1e9;
    printf("%-28s %9.3f %9.3f %9.1f %7.3f\n", v.name.c_str(), med, mn, gbs, gbs / 8000.0);
  }
  return 0;
}
