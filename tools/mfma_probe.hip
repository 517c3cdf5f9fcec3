// mfma_probe.hip -- does fp64 MFMA pay for a fused multi-qubit block?  (SURVEY 8d / VERDICT r01 "no MFMA
// variant was ever measured".)   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_probe tools/mfma_probe.hip
//
// A 3-qubit block on index bits 0..2 (one "register group" of the fused pass: 8 complex amplitudes = 16 reals
// per octet) is applied R times to every octet of a 2^27-amplitude state, three ways:
//   V  k = 3 butterflies: three general complex 2x2 gates, one per bit (what a tensor-product block
//      U2 x U1 x U0 costs as butterflies: 3 x 4 pairs x 16 = 192 f64 vector ops per octet and round);
//   D  the same block as ONE dense 8x8 complex matrix on the vector ALU (256 FMA per octet and round);
//   M  the same dense block as a real 16x16 matrix on the matrix cores: v_mfma_f64_16x16x4_f64, 4 per 16
//      octets and round; the accumulator layout (col = lane & 15, row = (lane >> 4) + 4 reg) is also the B
//      layout of the next round, so rounds chain without moving data.
// R = 1 is the HBM-bound single application; R = 16 keeps the octet in registers (the fused-pass regime,
// where the gate phase is bound by vector issue).  All three must agree to 1e-12.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef std::complex<double> cd;
typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Gates { double2 u[3][4]; };           // three 2x2 gates (bit 0, 1, 2), row-major
struct Dense { double2 m[64]; };             // 8x8 complex, row-major
struct Real16 { double m[256]; };            // 16x16 real, row-major: [[Re, -Im], [Im, Re]] interleaved per amplitude

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 c) {
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}

template <int R>
__global__ __launch_bounds__(256) void k_butterflies(double2* amp, const Gates g) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  double2 x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = amp[o * 8 + i];
#pragma unroll 1
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (!((i >> b) & 1)) {
          const double2 lo = x[i], hi = x[i | (1 << b)];
          x[i] = cfma(g.u[b][1], hi, cmul(g.u[b][0], lo));
          x[i | (1 << b)] = cfma(g.u[b][3], hi, cmul(g.u[b][2], lo));
        }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) amp[o * 8 + i] = x[i];
}

template <int R>
__global__ __launch_bounds__(256) void k_dense_valu(double2* amp, const Dense d) {
  const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
  double2 x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = amp[o * 8 + i];
#pragma unroll 1
  for (int r = 0; r < R; ++r) {
    double2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      double2 acc = cmul(d.m[8 * i], x[0]);
#pragma unroll
      for (int j = 1; j < 8; ++j) acc = cfma(d.m[8 * i + j], x[j], acc);
      y[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = y[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) amp[o * 8 + i] = x[i];
}

// One wave = 64 octets = four 16-column blocks; lane (j = l & 15, q = l >> 4) holds the reals q, 4+q, 8+q, 12+q
// of columns j, 16+j, 32+j, 48+j.  A operand of k-step s: Mr[row = l & 15][4 s + q].
template <int R>
__global__ __launch_bounds__(256) void k_dense_mfma(double* amp, const double* __restrict__ mr) {
  const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  double* base = amp + wave * 64 * 16;                 // 64 octets x 16 reals
  double a[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) a[s] = mr[16 * j + 4 * s + q];
  double4_t x[4];                                      // x[c][s] = real 4 s + q of column 16 c + j
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) x[c][s] = base[(16 * c + j) * 16 + 4 * s + q];
#pragma unroll 1
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], x[c][s], acc, 0, 0, 0);
      x[c] = acc;                                      // D[row = q + 4 i][col = j] in element i: the next round's B
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s) base[(16 * c + j) * 16 + 4 * s + q] = x[c][s];
}

static double splitmix(unsigned long long& s) {
  s += 0x9E3779B97F4A7C15ull;
  unsigned long long z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (double)((z ^ (z >> 31)) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

static void random_unitary2(unsigned long long& seed, cd u[4]) {
  const double t = 3.0 * splitmix(seed), a = 6.0 * splitmix(seed), b = 6.0 * splitmix(seed), c = 6.0 * splitmix(seed);
  u[0] = std::polar(std::cos(t), a);
  u[1] = -std::polar(std::sin(t), b);
  u[2] = std::polar(std::sin(t), c);
  u[3] = std::polar(std::cos(t), b + c - a);
}

template <class F>
static float timed(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return best;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 27;
  const size_t N = 1ull << n, octets = N / 8;
  unsigned long long seed = 7;
  cd u[3][4];
  for (int b = 0; b < 3; ++b) random_unitary2(seed, u[b]);
  Gates g;
  for (int b = 0; b < 3; ++b)
    for (int e = 0; e < 4; ++e) g.u[b][e] = make_double2(u[b][e].real(), u[b][e].imag());
  Dense d;
  Real16 m16;
  for (int i = 0; i < 8; ++i)
    for (int k = 0; k < 8; ++k) {
      cd v = 1.0;
      for (int b = 0; b < 3; ++b) v *= u[b][2 * ((i >> b) & 1) + ((k >> b) & 1)];
      d.m[8 * i + k] = make_double2(v.real(), v.imag());
      m16.m[16 * (2 * i) + 2 * k] = v.real();      m16.m[16 * (2 * i) + 2 * k + 1] = -v.imag();
      m16.m[16 * (2 * i + 1) + 2 * k] = v.imag();  m16.m[16 * (2 * i + 1) + 2 * k + 1] = v.real();
    }
  std::vector<cd> host(N);
  for (size_t i = 0; i < N; ++i) host[i] = cd(splitmix(seed), splitmix(seed));
  double2 *s0, *sv, *sd, *sm;
  double* dmr;
  CHECK(hipMalloc(&s0, N * 16)); CHECK(hipMalloc(&sv, N * 16)); CHECK(hipMalloc(&sd, N * 16)); CHECK(hipMalloc(&sm, N * 16));
  CHECK(hipMalloc(&dmr, sizeof m16));
  CHECK(hipMemcpy(s0, host.data(), N * 16, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dmr, m16.m, sizeof m16, hipMemcpyHostToDevice));
  const unsigned grid = (unsigned)(octets / 256);
  // ---- agreement after one application ----
  CHECK(hipMemcpy(sv, s0, N * 16, hipMemcpyDeviceToDevice));
  CHECK(hipMemcpy(sd, s0, N * 16, hipMemcpyDeviceToDevice));
  CHECK(hipMemcpy(sm, s0, N * 16, hipMemcpyDeviceToDevice));
  hipLaunchKernelGGL(k_butterflies<1>, dim3(grid), dim3(256), 0, 0, sv, g);
  hipLaunchKernelGGL(k_dense_valu<1>, dim3(grid), dim3(256), 0, 0, sd, d);
  hipLaunchKernelGGL(k_dense_mfma<1>, dim3(grid), dim3(256), 0, 0, (double*)sm, dmr);
  CHECK(hipDeviceSynchronize());
  std::vector<cd> hv(1 << 16), hd(1 << 16), hm(1 << 16);
  CHECK(hipMemcpy(hv.data(), sv + (N / 2), hv.size() * 16, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hd.data(), sd + (N / 2), hd.size() * 16, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hm.data(), sm + (N / 2), hm.size() * 16, hipMemcpyDeviceToHost));
  double e_d = 0, e_m = 0;
  for (size_t i = 0; i < hv.size(); ++i) { e_d = std::fmax(e_d, std::abs(hv[i] - hd[i])); e_m = std::fmax(e_m, std::abs(hv[i] - hm[i])); }
  std::printf("n=%d: max |butterflies - dense VALU| = %.2e, max |butterflies - MFMA| = %.2e\n", n, e_d, e_m);
  if (!(e_d < 1e-12 && e_m < 1e-12)) { std::printf("MISMATCH\n"); return 1; }
  // ---- timing ----
  const double gb = 32.0 * (double)N / 1e9;
  auto row = [&](const char* name, int R, float ms, double ops_per_octet) {
    std::printf("%-34s R=%2d  %8.3f ms  %7.1f GB/s moved  %7.2f Tflop/s-equivalent (%4.0f f64 flop per octet and round)\n", name, R, ms,
                gb / (ms * 1e-3), 2.0 * ops_per_octet * R * (double)octets / (ms * 1e-3) / 1e12, 2.0 * ops_per_octet);
  };
  row("V  3 butterflies (VALU)", 1, timed([&] { hipLaunchKernelGGL(k_butterflies<1>, dim3(grid), dim3(256), 0, 0, sv, g); }, 5), 192);
  row("D  dense 8x8 complex (VALU)", 1, timed([&] { hipLaunchKernelGGL(k_dense_valu<1>, dim3(grid), dim3(256), 0, 0, sd, d); }, 5), 256);
  row("M  dense 16x16 real (MFMA f64)", 1, timed([&] { hipLaunchKernelGGL(k_dense_mfma<1>, dim3(grid), dim3(256), 0, 0, (double*)sm, dmr); }, 5), 256);
  row("V  3 butterflies (VALU)", 16, timed([&] { hipLaunchKernelGGL(k_butterflies<16>, dim3(grid), dim3(256), 0, 0, sv, g); }, 5), 192);
  row("D  dense 8x8 complex (VALU)", 16, timed([&] { hipLaunchKernelGGL(k_dense_valu<16>, dim3(grid), dim3(256), 0, 0, sd, d); }, 5), 256);
  row("M  dense 16x16 real (MFMA f64)", 16, timed([&] { hipLaunchKernelGGL(k_dense_mfma<16>, dim3(grid), dim3(256), 0, 0, (double*)sm, dmr); }, 5), 256);
  row("V  3 butterflies (VALU)", 64, timed([&] { hipLaunchKernelGGL(k_butterflies<64>, dim3(grid), dim3(256), 0, 0, sv, g); }, 3), 192);
  row("D  dense 8x8 complex (VALU)", 64, timed([&] { hipLaunchKernelGGL(k_dense_valu<64>, dim3(grid), dim3(256), 0, 0, sd, d); }, 3), 256);
  row("M  dense 16x16 real (MFMA f64)", 64, timed([&] { hipLaunchKernelGGL(k_dense_mfma<64>, dim3(grid), dim3(256), 0, 0, (double*)sm, dmr); }, 3), 256);
  return 0;
}
