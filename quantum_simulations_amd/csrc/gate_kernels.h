// gate_kernels.h -- device helpers and the per-gate kernels k_gate / k_gate_shuffle.
// Part of the single translation unit qsim_hip.hip (included there, in order; not a standalone header).
// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 acc) {
  // acc + a*b
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, acc.x)), fma(a.x, b.y, fma(a.y, b.x, acc.y)));
}

// Streaming (non-temporal) 16-byte accesses: `global_load/store_dwordx4 ... nt`.  Measured on
// MI355X (profiles/r01_bw_probe.txt): +6..15 % on the in-place butterfly when every wave
// instruction covers whole 128-B lines; harmful when a line is shared by two instructions,
// so the launcher only selects NT when the lowest removed index bit is >= 3.
template <bool NT>
__device__ __forceinline__ double2 ld_amp(const double2* p) {
  if (NT) {
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st_amp(double2* p, double2 v) {
  if (NT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else {
    *p = v;
  }
}

// Re-insert zero bits at ascending positions pos[0..npos) of a compressed index.
__device__ __forceinline__ u64 expand_index(u64 c, int npos, int p0, int p1, int p2) {
  if (npos > 0) c = ((c >> p0) << (p0 + 1)) | (c & ((1ull << p0) - 1));
  if (npos > 1) c = ((c >> p1) << (p1 + 1)) | (c & ((1ull << p1) - 1));
  if (npos > 2) c = ((c >> p2) << (p2 + 1)) | (c & ((1ull << p2) - 1));
  return c;
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one).  With SWZ
// each XCD walks one contiguous eighth of the work-item space instead of every eighth block.
// (2-D grids only carry block counts whose x extent would overflow the 2^32 work-item limit of one
// grid dimension -- 33-qubit chunks; the linear id is y * gridDim.x + x.)
template <bool SWZ>
__device__ __forceinline__ u64 logical_block() {
  const u64 bid = (u64)blockIdx.y * gridDim.x + blockIdx.x;
  if (SWZ) {
    const u64 per_xcd = ((u64)gridDim.x * gridDim.y) >> 3;
    return (bid & 7) * per_xcd + (bid >> 3);
  }
  return bid;
}

constexpr unsigned kMaxGridX = 1u << 22;   // x * 256 threads stays below 2^32 work-items
static dim3 grid_for(u64 blocks) {
  if (blocks <= kMaxGridX) return dim3((unsigned)blocks);
  return dim3(kMaxGridX, (unsigned)((blocks + kMaxGridX - 1) / kMaxGridX));
}

template <int NM>
struct GateArgs {
  double2* member[NM];  // base pointer of each member (offsets folded in)
  u64 count;            // work items
  int npos;
  int pos[3];
  double2 u[NM * NM];   // row-major NM x NM
};

constexpr int kBlock = 256;

// All target bits resolved in registers: a work item owns NM amplitudes.
template <int NM, int ITEMS, bool NT, bool SWZ>
__global__ __launch_bounds__(kBlock) void k_gate(const GateArgs<NM> a) {
  const u64 first = (logical_block<SWZ>() * ITEMS) * kBlock + threadIdx.x;
  u64 idx[ITEMS];
  bool live[ITEMS];
  double2 x[ITEMS][NM];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const u64 c = first + (u64)r * kBlock;
    live[r] = c < a.count;
    idx[r] = expand_index(c, a.npos, a.pos[0], a.pos[1], a.pos[2]);
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (live[r]) {
#pragma unroll
      for (int m = 0; m < NM; ++m) x[r][m] = ld_amp<NT>(a.member[m] + idx[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (live[r]) {
#pragma unroll
      for (int row = 0; row < NM; ++row) {
        double2 acc = cmul(a.u[row * NM], x[r][0]);
#pragma unroll
        for (int col = 1; col < NM; ++col) acc = cfma(a.u[row * NM + col], x[r][col], acc);
        st_amp<NT>(a.member[row] + idx[r], acc);
      }
    }
  }
}

// Low target bits (index bit < 3: partners share a 128-B line) are resolved across lanes:
// every lane loads its own amplitude(s) with whole-line coalescing, fetches the partner
// values with ds_bpermute (`__shfl_xor`) and computes only its own output row.
//   NMR register members x 2^NSH lane states; canonical matrix index = (r << NSH) | s,
//   s bit b <-> lane bit lane_bit[b].
template <int NMR, int NSH>
struct ShuffleArgs {
  double2* member[NMR];
  u64 count;
  int npos;
  int pos[3];
  int lane_bit[2];
  double2 u[(NMR << NSH) * (NMR << NSH)];
};

template <int NMR, int NSH, int ITEMS, bool NT>
__global__ __launch_bounds__(kBlock) void k_gate_shuffle(const ShuffleArgs<NMR, NSH> a) {
  constexpr int NL = 1 << NSH;
  constexpr int DIM = NMR << NSH;
  const int lane = threadIdx.x & 63;
  int s = 0;
#pragma unroll
  for (int b = 0; b < NSH; ++b) s |= ((lane >> a.lane_bit[b]) & 1) << b;
  // per-lane coefficients: coef[r][r2][d] = U[(r<<NSH)|s][(r2<<NSH)|(s^d)]
  double2 coef[NMR][NMR][NL];
  int xmask[NL];
#pragma unroll
  for (int d = 0; d < NL; ++d) {
    xmask[d] = 0;
#pragma unroll
    for (int b = 0; b < NSH; ++b) xmask[d] |= ((d >> b) & 1) << a.lane_bit[b];
#pragma unroll
    for (int r = 0; r < NMR; ++r)
#pragma unroll
      for (int r2 = 0; r2 < NMR; ++r2) {
        double2 c = a.u[((r << NSH) | 0) * DIM + ((r2 << NSH) | (0 ^ d))];
#pragma unroll
        for (int sv = 1; sv < NL; ++sv) {
          const double2 alt = a.u[((r << NSH) | sv) * DIM + ((r2 << NSH) | (sv ^ d))];
          c.x = (s == sv) ? alt.x : c.x;
          c.y = (s == sv) ? alt.y : c.y;
        }
        coef[r][r2][d] = c;
      }
  }
  const u64 first = (logical_block<false>() * ITEMS) * kBlock + threadIdx.x;
  u64 idx[ITEMS];
  bool live[ITEMS];
  double2 x[ITEMS][NMR];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const u64 c = first + (u64)it * kBlock;
    live[it] = c < a.count;
    idx[it] = expand_index(c, a.npos, a.pos[0], a.pos[1], a.pos[2]);
#pragma unroll
    for (int r = 0; r < NMR; ++r) x[it][r] = make_double2(0.0, 0.0);
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (live[it]) {
#pragma unroll
      for (int r = 0; r < NMR; ++r) x[it][r] = ld_amp<NT>(a.member[r] + idx[it]);
    }
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    double2 out[NMR];
#pragma unroll
    for (int r = 0; r < NMR; ++r) out[r] = make_double2(0.0, 0.0);
#pragma unroll
    for (int r2 = 0; r2 < NMR; ++r2) {
#pragma unroll
      for (int d = 0; d < NL; ++d) {
        double2 v = x[it][r2];
        if (d != 0) {  // all 64 lanes take part (inactive tail lanes hold zeros)
          v.x = __shfl_xor(x[it][r2].x, xmask[d], 64);
          v.y = __shfl_xor(x[it][r2].y, xmask[d], 64);
        }
#pragma unroll
        for (int r = 0; r < NMR; ++r) out[r] = cfma(coef[r][r2][d], v, out[r]);
      }
    }
    if (live[it]) {
#pragma unroll
      for (int r = 0; r < NMR; ++r) st_amp<NT>(a.member[r] + idx[it], out[r]);
    }
  }
}
