// Memory-pattern probe of a tile pass with DIFFERENT read and write layouts (round 3).
//   hipcc -O3 --offload-arch=gfx950 -o perm_probe perm_probe.hip
//   ./perm_probe <n_qubits> "<spec>" ["<spec>" ...]
// One workgroup moves tiles of 2^11 amplitudes exactly like a gate-less k_tile pass (256 threads, 8 x 16 B
// per thread, whole 128-B lines, non-temporal, TPW consecutive tiles per workgroup with the second tile's
// loads issued before the first is stored), but the store address is a BIT PERMUTATION of the load address:
//   out_index = sum_b bit(in_index, b) << pi[b]
// A spec is   R=b0,...,b7[;P=a>b,c>d,...][;inplace][;tpw=1|2][;lds=0|1][;order=0|1|2][;name=...]
//   R      the eight tile bits above the three line bits (read layout)
//   P      moves of the permutation (a>b: input bit a lands on output bit b); the rest is the identity and the
//          moves must form a permutation of the bits they name;  bits 0..2 stay (whole lines)
//   inplace  store into the source buffer (P must be empty)
//   lds    1: the tile goes registers -> LDS -> barrier -> registers before the store (as a gate-less pass does);
//          2: the same in two halves (real, imaginary) through 16 KiB; 0: no LDS.  occ=4|5|8: waves per SIMD (lds 0 / 2);
//   pad    bytes of unused dynamic LDS per workgroup (caps the workgroups per CU: 160 KiB in 1280-byte granules)
//   plain  1: plain loads, 2: plain stores, 3: both (default: non-temporal)
//   lay    which tile bit (index into the sorted R) each in-tile position takes: 5 thread bits (3 lane, 2 wave), then the 3 bits of
//          a thread's 8 accesses; default 0,1,...,7
//   swz    software XOR swizzle of the addresses (both sides): a>b,... = physical bit b ^= index bit a (a > b, b >= 3)
//   ro / wo  reads only / writes only (16 B per amplitude)
//   order  0 consecutive tiles in flight, 1 XCD-contiguous (each XCD walks one eighth), 2 bit-reversed,
//          3 consecutive in the OUTPUT: the tile number counts the non-tile bits in the order of their output positions
// Prints milliseconds per pass (median of 7) and TB/s moved (32 B per amplitude), and checks the result
// against the permutation on sampled amplitudes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args {
  double2* src;
  double2* dst;
  unsigned ntiles;
  int order;
  int rw;                   // 0 read + write, 1 reads only, 2 writes only
  int plain;                // bit 0: plain (cached) loads instead of non-temporal, bit 1: plain stores
  int n;
  unsigned char R[8];       // ascending tile bits (read layout)
  unsigned char lay[8];     // in-tile bit i (5 thread bits, then 3 element bits) -> index into R
  unsigned char pi[40];     // input bit -> output bit
  unsigned char swz_src[8], swz_dst[8];   // software address swizzle: physical bit dst ^= index bit src (both sides)
  int nswz;
  unsigned char ord[40];    // order 3: the j-th bit of the tile number is input bit ord[j] (non-tile bits sorted by OUTPUT position)
};

// LDS: 0 none, 1 whole amplitudes (32 KiB: four workgroups per CU), 2 real and imaginary parts one after the other
// through a 16 KiB buffer (eight workgroups per CU fit); OCC: waves per SIMD the register allocator is told
template <int TPW, int LDS, int OCC>
__global__ __launch_bounds__(256, OCC) void k_move(const Args a) {
  __shared__ double lds[LDS == 1 ? 4096 : (LDS == 2 ? 2048 : 1)];
  const int tid = threadIdx.x;
  typedef double amp_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) amp_t gamp_t;
  auto tile_base = [&](unsigned i, u64* out) -> u64 {
    unsigned tile = blockIdx.x * TPW + i;
    if (a.order == 1) { const unsigned per = a.ntiles >> 3; const unsigned wg = blockIdx.x; tile = ((wg & 7) * (per / TPW) + (wg >> 3)) * TPW + i; }
    if (a.order == 2) tile = __brev(tile) >> (__clz(a.ntiles) + 1);
    u64 base = (u64)tile << 3;
    if (a.order == 3) {
      base = 0;
      for (int j = 0; j < a.n - 11; ++j) base |= (u64)((tile >> j) & 1u) << a.ord[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = a.R[j];
        base = ((base >> p) << (p + 1)) | (base & ((1ull << p) - 1));
      }
    }
    u64 o = 0;
    for (int b = 3; b < a.n; ++b) o |= ((base >> b) & 1ull) << a.pi[b];
    *out = o;
    return base;
  };
  u64 tin = tid & 7, tout = tid & 7;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int p = a.R[a.lay[i]];
    tin |= (u64)((tid >> (3 + i)) & 1) << p;
    tout |= (u64)((tid >> (3 + i)) & 1) << a.pi[p];
  }
  u64 jin[8], jout[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    jin[j] = jout[j] = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int p = a.R[a.lay[5 + b]];
      jin[j] |= (u64)((j >> b) & 1) << p;
      jout[j] |= (u64)((j >> b) & 1) << a.pi[p];
    }
  }
  auto swz = [&](u64 i) -> u64 {
    u64 x = 0;
    for (int t = 0; t < a.nswz; ++t) x |= ((i >> a.swz_src[t]) & 1ull) << a.swz_dst[t];
    return i ^ x;
  };
  // (the swizzle is linear over XOR and the three parts of an index have disjoint bits: swizzle them separately)
  const u64 stin = swz(tin), stout = swz(tout);
  u64 sjin[8], sjout[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sjin[j] = swz(jin[j]); sjout[j] = swz(jout[j]); }
  u64 obase;
  u64 base = tile_base(0, &obase);
  u64 sb = swz(base);
  amp_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a.rw == 2 ? amp_t{1.0, 2.0} : ((a.plain & 1) ? *(gamp_t*)(a.src + (sb ^ stin ^ sjin[j])) : __builtin_nontemporal_load((gamp_t*)(a.src + (sb ^ stin ^ sjin[j]))));
#pragma unroll 1
  for (int it = 0; it < TPW; ++it) {
    amp_t x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[j];
    const u64 cur_out = obase;
    const u64 scur = swz(cur_out);
    if (it + 1 < TPW) {
      base = tile_base(it + 1, &obase);
      sb = swz(base);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = a.rw == 2 ? amp_t{1.0, 2.0} : ((a.plain & 1) ? *(gamp_t*)(a.src + (sb ^ stin ^ sjin[j])) : __builtin_nontemporal_load((gamp_t*)(a.src + (sb ^ stin ^ sjin[j]))));
    }
    if (LDS == 1) {
      amp_t* t = reinterpret_cast<amp_t*>(lds);
#pragma unroll
      for (int j = 0; j < 8; ++j) t[(tid ^ ((tid >> 4) & 15)) + 256 * j] = x[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = t[(tid ^ ((tid >> 4) & 15)) + 256 * j];
    }
    if (LDS == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) lds[(tid ^ ((tid >> 4) & 15)) + 256 * j] = x[j].x;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j].x = lds[(tid ^ ((tid >> 4) & 15)) + 256 * j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) lds[(tid ^ ((tid >> 4) & 15)) + 256 * j] = x[j].y;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j].y = lds[(tid ^ ((tid >> 4) & 15)) + 256 * j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (a.rw != 1 || x[j].x == -12345.678) {
        if (a.plain & 2) *(gamp_t*)(a.dst + (scur ^ stout ^ sjout[j])) = x[j];
        else __builtin_nontemporal_store(x[j], (gamp_t*)(a.dst + (scur ^ stout ^ sjout[j])));
      }
    if (LDS != 0 && it + 1 < TPW) __syncthreads();
  }
}

__global__ void k_fill(double2* p, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
    p[i] = make_double2((double)i, -(double)(i & 0xFFFF));
}

static std::vector<std::string> split(const std::string& s, char c) {
  std::vector<std::string> out;
  size_t at = 0;
  while (at <= s.size()) {
    size_t e = s.find(c, at);
    if (e == std::string::npos) e = s.size();
    if (e > at) out.push_back(s.substr(at, e - at));
    at = e + 1;
  }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: perm_probe n spec...\n"); return 2; }
  const int n = atoi(argv[1]);
  const u64 N = 1ull << n;
  double2 *A = nullptr, *B = nullptr;
  CK(hipMalloc((void**)&A, N * sizeof(double2)));
  CK(hipMalloc((void**)&B, N * sizeof(double2)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int s = 2; s < argc; ++s) {
    Args a;
    std::memset(&a, 0, sizeof a);
    a.n = n;
    for (int b = 0; b < 40; ++b) a.pi[b] = (unsigned char)b;
    bool inplace = false;
    int tpw = 2, lds = 1, occ = 4, pad = 0;
    for (int i = 0; i < 8; ++i) a.lay[i] = (unsigned char)i;
    std::string name = argv[s];
    bool ok = true;
    for (const std::string& part : split(argv[s], ';')) {
      if (part.rfind("R=", 0) == 0) {
        auto v = split(part.substr(2), ',');
        if (v.size() != 8) { ok = false; break; }
        for (int i = 0; i < 8; ++i) a.R[i] = (unsigned char)atoi(v[i].c_str());
        std::sort(a.R, a.R + 8);
      } else if (part.rfind("P=", 0) == 0) {
        for (const std::string& mv : split(part.substr(2), ',')) {
          auto ab = split(mv, '>');
          if (ab.size() != 2) { ok = false; break; }
          a.pi[atoi(ab[0].c_str())] = (unsigned char)atoi(ab[1].c_str());
        }
      } else if (part == "inplace") inplace = true;
      else if (part.rfind("tpw=", 0) == 0) tpw = atoi(part.c_str() + 4);
      else if (part.rfind("lds=", 0) == 0) lds = atoi(part.c_str() + 4);
      else if (part.rfind("swz=", 0) == 0) {
        for (const std::string& mv : split(part.substr(4), ',')) {
          auto ab = split(mv, '>');
          if (ab.size() != 2 || a.nswz >= 8) { ok = false; break; }
          a.swz_src[a.nswz] = (unsigned char)atoi(ab[0].c_str());
          a.swz_dst[a.nswz] = (unsigned char)atoi(ab[1].c_str());
          ++a.nswz;
        }
      }
      else if (part.rfind("lay=", 0) == 0) {
        auto v = split(part.substr(4), ',');
        if (v.size() != 8) { ok = false; break; }
        for (int i = 0; i < 8; ++i) a.lay[i] = (unsigned char)atoi(v[i].c_str());
      }
      else if (part.rfind("pad=", 0) == 0) pad = atoi(part.c_str() + 4);
      else if (part.rfind("occ=", 0) == 0) occ = atoi(part.c_str() + 4);
      else if (part.rfind("order=", 0) == 0) a.order = atoi(part.c_str() + 6);
      else if (part.rfind("plain=", 0) == 0) a.plain = atoi(part.c_str() + 6);
      else if (part == "ro") a.rw = 1;
      else if (part == "wo") a.rw = 2;
      else if (part.rfind("name=", 0) == 0) name = part.substr(5);
      else ok = false;
    }
    u64 seen = 0;
    for (int b = 0; b < n; ++b) { if (a.pi[b] >= n || ((seen >> a.pi[b]) & 1)) ok = false; seen |= 1ull << a.pi[b]; }
    for (int b = 0; b < 3; ++b) if (a.pi[b] != b) ok = false;
    for (int i = 0; i < 8; ++i) if (a.R[i] < 3 || a.R[i] >= n || (i && a.R[i] == a.R[i - 1])) ok = false;
    if (!ok) { printf("bad spec: %s\n", argv[s]); continue; }
    {   // order 3: non-tile input bits sorted by where they land in the output
      std::vector<int> free_bits;
      for (int b = 3; b < n; ++b) { bool t = false; for (int i = 0; i < 8; ++i) t = t || a.R[i] == b; if (!t) free_bits.push_back(b); }
      std::sort(free_bits.begin(), free_bits.end(), [&](int x, int y) { return a.pi[x] < a.pi[y]; });
      for (size_t j = 0; j < free_bits.size(); ++j) a.ord[j] = (unsigned char)free_bits[j];
    }
    a.src = A;
    a.dst = inplace ? A : B;
    a.ntiles = (unsigned)(N >> 11);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, N);
    if (!inplace) CK(hipMemsetAsync(B, 0xFF, N * sizeof(double2), 0));
    auto launch = [&]() {
      const unsigned grid = a.ntiles / (unsigned)tpw;
#define GO(T, L, O) hipLaunchKernelGGL((k_move<T, L, O>), dim3(grid), dim3(256), (size_t)pad, 0, a)
      if (lds == 1) { if (tpw == 2) GO(2, 1, 4); else GO(1, 1, 4); }
      else if (lds == 2) {
        if (tpw == 2) { if (occ >= 8) GO(2, 2, 8); else if (occ >= 5) GO(2, 2, 5); else GO(2, 2, 4); }
        else { if (occ >= 8) GO(1, 2, 8); else if (occ >= 5) GO(1, 2, 5); else GO(1, 2, 4); }
      } else {
        if (tpw == 2) { if (occ >= 8) GO(2, 0, 8); else if (occ >= 5) GO(2, 0, 5); else GO(2, 0, 4); }
        else { if (occ >= 8) GO(1, 0, 8); else if (occ >= 5) GO(1, 0, 5); else GO(1, 0, 4); }
      }
#undef GO
    };
    launch();
    CK(hipDeviceSynchronize());
    // check sampled amplitudes: dst[pi(i)] == value of i
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
      u64 i = ((u64)rand() << 20 ^ (u64)rand()) & (N - 1), o = 0;
      for (int b = 0; b < n; ++b) o |= ((i >> b) & 1ull) << a.pi[b];
      double2 got;
      { u64 x = 0; for (int t = 0; t < a.nswz; ++t) x |= ((o >> a.swz_src[t]) & 1ull) << a.swz_dst[t]; o ^= x; }
      CK(hipMemcpy(&got, a.dst + o, sizeof got, hipMemcpyDeviceToHost));
      u64 si = i; { u64 x = 0; for (int t = 0; t < a.nswz; ++t) x |= ((i >> a.swz_src[t]) & 1ull) << a.swz_dst[t]; si ^= x; }
      if (got.x != (double)si) ++bad;
    }
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) {
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t = 0; CK(hipEventElapsedTime(&t, e0, e1));
      ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-64s %7.4f ms  %6.3f TB/s  min %.4f%s\n", name.c_str(), ms[3], (a.rw ? 16.0 : 32.0) * (double)N / (ms[3] * 1e-3) / 1e12, ms[0],
           bad && !inplace && !a.rw ? "  CHECK FAILED" : "");
    fflush(stdout);
  }
  return 0;
}
