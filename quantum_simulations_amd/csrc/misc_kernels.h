// misc_kernels.h -- argument checks and the non-gate kernels (fill, norm, pack, closed-form checkers).
// Part of the single translation unit qsim_hip.hip (included there, in order; not a standalone header).
// ------------------------------------------------------------------ argument checks
static int check_chunk(const qsim_chunk* c, const char* what) {
  if (!c || !c->amp) return fail(QSIM_ERR_INVALID, "%s: null chunk", what);
  return QSIM_OK;
}

static int check_local_qubit(const qsim_chunk* c, int q) {
  if (q < 0) return fail(QSIM_ERR_INVALID, "qubit %d is negative", q);
  if (q >= c->k)
    return fail(QSIM_ERR_NONLOCAL,
                "qubit %d >= log2(chunk_size)=%d: non-local gate requires layout/collect step",
                q, c->k);
  return QSIM_OK;
}

static int check_group(qsim_chunk* const* cs, int n, const char* what) {
  for (int i = 0; i < n; ++i) {
    int rc = check_chunk(cs[i], what);
    if (rc) return rc;
    if (cs[i]->k != cs[0]->k) return fail(QSIM_ERR_INVALID, "%s: chunks differ in size", what);
    if (cs[i]->device != cs[0]->device)
      return fail(QSIM_ERR_INVALID, "%s: chunks live on different devices", what);
    for (int j = 0; j < i; ++j)
      if (cs[i]->amp == cs[j]->amp) return fail(QSIM_ERR_INVALID, "%s: the same chunk twice", what);
  }
  return QSIM_OK;
}

// ------------------------------------------------------------------ misc kernels
__global__ void k_fill_zero(double2* p, u64 n, int set0) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = make_double2((i == 0 && set0) ? 1.0 : 0.0, 0.0);
}

__device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// amplitude i = (u(2i), u(2i+1)), u(j) = (splitmix64(seed + j) >> 11) * 2^-52 - 1 in [-1, 1)
__global__ void k_fill_random(double2* p, u64 n, u64 seed) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double re = (double)(splitmix64(seed + 2 * i) >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    const double im = (double)(splitmix64(seed + 2 * i + 1) >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    p[i] = make_double2(re, im);
  }
}

__global__ void k_scale(double2* p, u64 n, double s) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double2 v = p[i];
    p[i] = make_double2(v.x * s, v.y * s);
  }
}

// NT: streaming (non-temporal) accesses for copies larger than the Infinity Cache -- the same cache
// policy the gate kernels use; this copy is also bench.py's same-run device-to-device ceiling.
// One-shot grid, ITEMS 16-byte elements per thread, XCD-contiguous block order (the shape of the gate
// kernels: grid-stride loops are 5-8 % slower, tools/bw_probe.hip).
template <bool NT, int ITEMS>
__global__ __launch_bounds__(kBlock) void k_copy(double2* __restrict__ dst, const double2* __restrict__ src, u64 n) {
  const u64 first = (logical_block<true>() * ITEMS) * kBlock + threadIdx.x;
  double2 v[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) if (first + (u64)r * kBlock < n) v[r] = ld_amp<NT>(src + first + (u64)r * kBlock);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) if (first + (u64)r * kBlock < n) st_amp<NT>(dst + first + (u64)r * kBlock, v[r]);
}

// dst[j] = src[insert(j, bit, value)]  /  inverse
__global__ void k_pack_half(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_half,
                            int bit, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_half; j += stride) {
    const u64 i = (((j >> bit) << (bit + 1)) | (j & ((1ull << bit) - 1))) | value_off;
    dst[j] = src[i];
  }
}
__global__ void k_unpack_half(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_half,
                              int bit, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_half; j += stride) {
    const u64 i = (((j >> bit) << (bit + 1)) | (j & ((1ull << bit) - 1))) | value_off;
    dst[i] = src[j];
  }
}

// slab gather / scatter for the all-to-all re-layout: j runs over the 2^(k-m) amplitudes whose
// bits `pos` equal the pattern folded into value_off
__global__ void k_pack_bits(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_slab,
                            int npos, int p0, int p1, int p2, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_slab; j += stride)
    dst[j] = ld_amp<true>(src + (expand_index(j, npos, p0, p1, p2) | value_off));
}
__global__ void k_unpack_bits(double2* __restrict__ dst, const double2* __restrict__ src, u64 n_slab,
                              int npos, int p0, int p1, int p2, u64 value_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_slab; j += stride)
    dst[expand_index(j, npos, p0, p1, p2) | value_off] = ld_amp<true>(src + j);
}

// All slabs in one pass: amplitude i goes to slab pattern(i) at its compressed index (and back).
// Consecutive lanes read consecutive amplitudes, and the <= 8 slabs a wave feeds each receive a
// contiguous run, so reads and writes are whole 128-B lines for ANY choice of bits -- the
// per-pattern kernels above touch 16 B of every line when a selected bit is below 3
// (tools/relayout_probe.py: 0.7-1.0 TB/s instead of 4.5-5; the gather side of unpack with lanes of
// one line 8 apart still runs at 2.6-3.5 TB/s).  `skip` = the pattern that stays.
__device__ __forceinline__ u64 drop_bit(u64 x, int p) { return ((x >> (p + 1)) << p) | (x & ((1ull << p) - 1)); }
template <bool PACK>
__global__ void k_slabs_all(double2* __restrict__ state, double2* __restrict__ buf, u64 n, int m,
                            int b0, int b1, int b2, int s_hi, int s_mid, int s_lo, int slab_bits, int skip,
                            int n_piece_bits, int pb0, int pb1, int pb2, int piece) {
  // A piece = the amplitudes whose top `n_piece_bits` non-selected index bits equal `piece`: the
  // same contiguous sub-range of every slab, so the exchange can be pipelined piece by piece.
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    u64 i = t;
    if (n_piece_bits > 0) i = (((i >> pb0) << (pb0 + 1)) | (i & ((1ull << pb0) - 1))) | ((u64)(piece & 1) << pb0);
    if (n_piece_bits > 1) i = (((i >> pb1) << (pb1 + 1)) | (i & ((1ull << pb1) - 1))) | ((u64)((piece >> 1) & 1) << pb1);
    if (n_piece_bits > 2) i = (((i >> pb2) << (pb2 + 1)) | (i & ((1ull << pb2) - 1))) | ((u64)((piece >> 2) & 1) << pb2);
    int p = (int)((i >> b0) & 1);
    if (m > 1) p |= (int)((i >> b1) & 1) << 1;
    if (m > 2) p |= (int)((i >> b2) & 1) << 2;
    if (p == skip) continue;
    u64 j = i;
    if (m > 2) j = drop_bit(j, s_hi);
    if (m > 1) j = drop_bit(j, s_mid);
    j = drop_bit(j, s_lo);
    double2* const slot = buf + (((u64)p << slab_bits) | j);
    if (PACK) st_amp<true>(slot, ld_amp<true>(state + i));
    else st_amp<true>(state + i, ld_amp<true>(slot));
  }
}

// exchange slab (bits pos == a_off pattern) of chunk A with slab (bits pos == b_off pattern) of chunk B
__global__ void k_swap_slabs(double2* __restrict__ a, double2* __restrict__ b, u64 n_slab,
                             int npos, int p0, int p1, int p2, u64 a_off, u64 b_off) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_slab; j += stride) {
    const u64 e = expand_index(j, npos, p0, p1, p2);
    const double2 x = a[e | a_off], y = b[e | b_off];
    a[e | a_off] = y;
    b[e | b_off] = x;
  }
}

constexpr int kReduceBlocks = 2048;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

template <bool MAX>
__device__ __forceinline__ void block_reduce_store(double v, double* out) {
  __shared__ double part[kBlock / 64];
  v = MAX ? wave_max(v) : wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = part[0];
    for (int w = 1; w < kBlock / 64; ++w) r = MAX ? fmax(r, part[w]) : r + part[w];
    out[blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(kBlock) void k_norm2_partial(const double2* p, u64 n, double* partial) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double2 v = p[i];
    acc = fma(v.x, v.x, fma(v.y, v.y, acc));
  }
  block_reduce_store<false>(acc, partial);
}

// ---- complex64 <-> complex128 on the device: the reference's chunk files are complex64 (storage/block_store.py:11),
// so an export rounds on the GPU and moves 8 B per amplitude over PCIe instead of 16 (round to nearest even, what
// numpy's astype(complex64) does)
__global__ __launch_bounds__(kBlock) void k_to_c64(float2* __restrict__ dst, const double2* __restrict__ src, u64 n) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double2 v = src[i];
    dst[i] = make_float2(__double2float_rn(v.x), __double2float_rn(v.y));
  }
}
__global__ __launch_bounds__(kBlock) void k_from_c64(double2* __restrict__ dst, const float2* __restrict__ src, u64 n) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float2 v = src[i];
    dst[i] = make_double2((double)v.x, (double)v.y);
  }
}

// ---- sparse view of the state: the v3 worker's rows (idx, re, im) with its pruning rule |re| > eps or |im| > eps
// (parallel_gate_applicator.py:372-374, state_manager.py:95-106), made on the device: a GHZ state of 30 qubits has two
// rows, not 16 GiB.  One pass counts, a second one appends the kept amplitudes (one atomic per wave reserves the slots;
// the rows arrive unordered and are sorted by index on the host).
__device__ __forceinline__ bool kept_amp(double2 v, double eps) { return fabs(v.x) > eps || fabs(v.y) > eps; }

__global__ __launch_bounds__(kBlock) void k_count_kept(const double2* p, u64 n, double eps, unsigned long long* count) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) mine += kept_amp(p[i], eps) ? 1u : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

__global__ __launch_bounds__(kBlock) void k_append_kept(const double2* p, u64 n, double eps, unsigned long long* cursor,
                                                        u64 capacity, u64* out_idx, double2* out_amp) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 rounds = (n + stride - 1) / stride;        // every lane runs every round (the ballot needs whole waves)
  for (u64 r = 0; r < rounds; ++r) {
    const u64 i = r * stride + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const double2 v = i < n ? p[i] : make_double2(0.0, 0.0);
    const bool keep = i < n && kept_amp(v, eps);
    const unsigned long long mask = __ballot(keep);
    if (!mask) continue;
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(mask));
    base = __shfl(base, 0, 64);
    if (keep) {
      const u64 at = base + (u64)__popcll(mask & ((1ull << lane) - 1));
      if (at < capacity) { out_idx[at] = i; out_amp[at] = v; }
    }
  }
}

// kind 0: GHZ, kind 1: GHZ+QFT closed form.  `base` = global index of amp 0.
struct BitPerm { unsigned char to_logical[64]; int active; };   // physical index bit -> logical qubit

__global__ __launch_bounds__(kBlock) void k_closed_form_err(const double2* p, u64 n, int kind,
                                                            int n_total, u64 base, double* partial,
                                                            const BitPerm perm) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const double inv_n = exp2(-(double)n_total);
  const double amp = exp2(-0.5 * (double)(n_total + 1));
  const u64 last = (n_total >= 64) ? ~0ull : ((1ull << n_total) - 1);
  double worst = 0.0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u64 y = base + i;
    if (perm.active) {                       // staged layout: logical index from the physical one
      const u64 x = y;
      y = 0;
      for (int b = 0; b < n_total; ++b) y |= ((x >> b) & 1ull) << perm.to_logical[b];
    }
    double er, ei;
    if (kind == 0) {
      er = (y == 0 || y == last) ? 0.70710678118654752440 : 0.0;
      ei = 0.0;
    } else {
      // exp(-2 pi i y / 2^n): y * 2^-n is exact in double for n <= 52
      double s, c;
      sincospi(-2.0 * ((double)y * inv_n), &s, &c);
      er = amp * (1.0 + c);
      ei = amp * s;
    }
    const double2 v = p[i];
    worst = fmax(worst, hypot(v.x - er, v.y - ei));
  }
  block_reduce_store<true>(worst, partial);
}

// ---- layout-aware fingerprint of a (partitioned, staged) state: sum_i amp_i * w(y_i) over the chunk's amplitudes whose
// LOGICAL index y_i passes the filter (y & sel_mask) == sel_value, with counter-based pseudo-random complex weights of y
// (two rounds of splitmix64 over y ^ mix(seed); real and imaginary part uniform in [-1, 1), exactly representable).
// A state that is right up to rounding gives the same value wherever its amplitudes live: a shard of a multi-GPU run in
// its staged layout, or the same index set of a one-GPU run (ref_dense.simulate semantics, ref_dense.py:44-57; layout
// permute_state, staging.py:639-658).  Slabs that trade places, a wrong permutation or a lost phase change it at O(1).
__host__ __device__ __forceinline__ u64 fp_mix(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(kBlock) void k_fingerprint(const double2* p, u64 n, int n_total, u64 base, u64 seed_mix,
                                                        u64 sel_mask, u64 sel_value, double* partial, const BitPerm perm) {
  const u64 stride = (u64)gridDim.x * blockDim.x;
  double sr = 0.0, si = 0.0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u64 y = base + i;
    if (perm.active) {
      const u64 x = y;
      y = 0;
      for (int b = 0; b < n_total; ++b) y |= ((x >> b) & 1ull) << perm.to_logical[b];
    }
    if ((y & sel_mask) != sel_value) continue;
    const u64 a = fp_mix(y ^ seed_mix), b2 = fp_mix(a);
    const double wr = (double)(a >> 11) * 0x1p-52 - 1.0, wi = (double)(b2 >> 11) * 0x1p-52 - 1.0;
    const double2 v = p[i];
    sr += v.x * wr - v.y * wi;
    si += v.x * wi + v.y * wr;
  }
  __shared__ double part[2 * (kBlock / 64)];
  sr = wave_sum(sr);
  si = wave_sum(si);
  if ((threadIdx.x & 63) == 0) { part[2 * (threadIdx.x >> 6)] = sr; part[2 * (threadIdx.x >> 6) + 1] = si; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0, im = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) { r += part[2 * w]; im += part[2 * w + 1]; }
    partial[2 * blockIdx.x] = r;
    partial[2 * blockIdx.x + 1] = im;
  }
}

// ---- dense k-qubit block (v3's fused block as a genuine 2^k x 2^k contraction: parallel_gate_applicator.py:315-385) ------
// A work item owns the 2^K amplitudes that differ in the block's K index bits; new = M old with M row-major in device
// memory (wave-uniform addresses: scalar loads).  HBM-bound like every gate here (8 * 2^K flop per amplitude over the
// same 32 B: 2 flop / B at K = 3, 4 at K = 4 -- far under the fp64 ridge); the matrix cores would pay from K ~ 5-6 on
// (tools/mfma_probe.hip), which no caller of the reference's gate set produces.
struct DenseArgs {
  double2* amp;
  const double2* mat;       // 2^K x 2^K, row-major: M[out][in]
  u64 count;                // work items = 2^(k - K)
  int pos[4];               // the block's index bits, ascending (zeros are inserted there)
  int bit[4];               // pattern bit i <-> index bit bit[i] (the caller's qubit order)
};
template <int K, bool NT>
__global__ __launch_bounds__(kBlock) void k_dense(const DenseArgs a) {
  constexpr int N = 1 << K;
  u64 c = logical_block<true>() * kBlock + threadIdx.x;
  if (c >= a.count) return;
#pragma unroll
  for (int i = 0; i < K; ++i) { const int p = a.pos[i]; c = ((c >> p) << (p + 1)) | (c & ((1ull << p) - 1)); }
  double2 x[N];
#pragma unroll
  for (int s = 0; s < N; ++s) {
    u64 off = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) off |= (u64)((s >> i) & 1) << a.bit[i];
    x[s] = ld_amp<NT>(a.amp + (c | off));
  }
#pragma unroll
  for (int r = 0; r < N; ++r) {
    double2 acc = cmul(a.mat[r * N], x[0]);
#pragma unroll
    for (int col = 1; col < N; ++col) acc = cfma(a.mat[r * N + col], x[col], acc);
    u64 off = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) off |= (u64)((r >> i) & 1) << a.bit[i];
    st_amp<NT>(a.amp + (c | off), acc);
  }
}

// Dense 3- and 4-qubit blocks on the MATRIX cores (round 4): a genuine 2^K x 2^K complex contraction per block of 2^K
// amplitudes = a real 2^(K+1) x 2^(K+1) matrix [[Re, -Im], [Im, Re]] (rows / columns 2 p + c: pattern p, c = 0 real / 1
// imaginary -- the order the amplitudes lie in memory) times the 2^(K+1) reals of the block, as v_mfma_f64_16x16x4_f64: a
// wave takes 16 COLUMNS (= 16 blocks: consecutive values of the index bits that are not block bits) at a time; K = 4: two
// 16-row output tiles x eight k-steps = 16 MFMAs per 256 amplitudes (128 flop per amplitude: 1.75 ms of the matrix cores'
// 78 Tflop/s at 30 qubits, under the 4.3 ms of HBM); K = 3: one tile x four k-steps per 128 amplitudes.  Operand layouts
// (lane l: j = l & 15, g = l >> 4; measured with tools/mfma_probe.hip in round 2):
//   A of k-step s, row tile t:  Mr[16 t + j][4 s + g]            (formed once per wave from the caller's complex matrix)
//   B of k-step s:              X[row 4 s + g][column j]  = component g & 1 of pattern 2 s + (g >> 1) of block j
//   D element i of row tile t:  Y[row 16 t + 4 i + g][column j] = component g & 1 of pattern 2 (4 t + i) + (g >> 1)
// so a lane reads the doubles { pattern 2 m + (g >> 1), m = 0 .. 2^(K-1) - 1 } x { component g & 1 } of its block and writes
// its results back to the same places: in place, no shuffles, no LDS.  Lane pairs (g & 1) cover one amplitude (16 B), 32
// lanes 256 contiguous bytes when the block bits lie above index bit 3.  Measured at 30 qubits (profiles/r04x_*): K = 4
// 0.68-0.73 of the HBM peak against 0.43 for the vector-ALU form (k_dense<4>: 256 complex multiply-adds per work item at 128+
// VGPRs) -- the one place of this path where the work IS a matrix product (north star: "MFMA only for fused
// multi-qubit dense blocks where it is a real 2^k x 2^k contraction").
typedef double qs_double4_t __attribute__((ext_vector_type(4)));
#ifdef QSIM_PROBES          // (the round-4 form: kept in the probe build as the A/B partner of k_dense_mfma2 below)
struct DenseMfmaArgs {
  double* amp;              // the chunk as reals
  const double2* mat;       // the caller's 2^K x 2^K complex matrix, row-major M[out][in] (its real image is formed in registers)
  u64 col_blocks;           // groups of 16 columns: 2^(k - K - 4)
  int pos[4];               // the block's index bits, ascending (zeros are inserted there)
  int bit[4];               // pattern bit i <-> index bit bit[i] (the caller's qubit order)
};
constexpr int kDenseMfmaColBlocksPerWave = 4;
template <int K, bool NT>
__global__ __launch_bounds__(kBlock) void k_dense_mfma(const DenseMfmaArgs a) {
  static_assert(K == 3 || K == 4, "16-row MFMA tiles: 16 or 32 reals per block");
  constexpr int TT = 1 << (K - 3);          // 16-row output tiles
  constexpr int S = 1 << (K - 1);           // k-steps of 4 rows = patterns per lane
  constexpr int DIM = 1 << K;
  const int l = threadIdx.x & 63, j = l & 15, g = l >> 4;
  double A[TT][S];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int s = 0; s < S; ++s) {
      // Mr[2 po + co][2 pi + ci] = Re M[po][pi] if co == ci, Im if (co, ci) = (1, 0), -Im if (0, 1)
      const int r = 16 * t + j, col = 4 * s + g;
      const double2 z = a.mat[(r >> 1) * DIM + (col >> 1)];
      A[t][s] = (r & 1) == (col & 1) ? z.x : ((r & 1) ? z.y : -z.y);
    }
  u64 off[S];               // (wave-uniform) offsets of the patterns' upper K - 1 bits, in reals
#pragma unroll
  for (int m = 0; m < S; ++m) {
    u64 o = 0;
#pragma unroll
    for (int i = 0; i < K - 1; ++i) o |= (u64)((m >> i) & 1) << a.bit[i + 1];
    off[m] = 2 * o;
  }
  const u64 lane_part = 2 * ((u64)(g >> 1) << a.bit[0]) + (u64)(g & 1);
  const u64 wave = logical_block<true>() * (kBlock / 64) + (threadIdx.x >> 6);
#pragma unroll
  for (int it = 0; it < kDenseMfmaColBlocksPerWave; ++it) {
    const u64 cb = wave * kDenseMfmaColBlocksPerWave + it;
    if (cb >= a.col_blocks) break;                    // (wave-uniform)
    u64 c = cb * 16 + (u64)j;
#pragma unroll
    for (int i = 0; i < K; ++i) { const int p = a.pos[i]; c = ((c >> p) << (p + 1)) | (c & ((1ull << p) - 1)); }
    double* const p0 = a.amp + 2 * c + lane_part;
    double x[S];
#pragma unroll
    for (int m = 0; m < S; ++m) x[m] = NT ? __builtin_nontemporal_load(p0 + off[m]) : p0[off[m]];
    qs_double4_t acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = qs_double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t][s], x[s], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (NT) __builtin_nontemporal_store(acc[t][i], p0 + off[4 * t + i]);
        else p0[off[4 * t + i]] = acc[t][i];
      }
  }
}

#endif  // QSIM_PROBES

// Dense K-qubit blocks, K = 3 .. 6, on the matrix cores -- second form (round 5).  Same product as k_dense_mfma (a real
// 2^(K+1) x 2^(K+1) matrix times the reals of 16 blocks per wave and step, v_mfma_f64_16x16x4_f64), with two changes:
//   * the ROWS are ordered so that a lane owns WHOLE amplitudes: row r = 4 s + g with s = 2 mu + c -- g = the two LOWEST
//     pattern bits (the block's two lowest index bits), c = component (0 real, 1 imaginary), mu = the upper K - 2 pattern
//     bits.  Lane (j = l & 15, g = l >> 4) then holds the 2^(K-2) amplitudes { pattern 4 mu + g } of column j as double2 and
//     gets them back at the same places (D element i of row tile t is row 16 t + 4 i + g: s' = 4 t + i): 16-byte accesses,
//     in place, no LDS for the state, no shuffles.  With the block's lowest bits inside a 128-byte line the four g lanes of
//     a column cover 64 contiguous bytes and two neighbouring columns the rest of the line (r04's form touched every line
//     with four 32-byte pieces: 0.40 of peak on blocks over index bits 0-2);
//   * K = 5 and 6: the matrix image no longer fits registers (16 x 2^(K-1) / 64 ... = 64 / 256 doubles per lane), so every
//     workgroup keeps it in LDS in A-OPERAND layout -- tab[(t S + s) 64 + lane] = Mr[16 t + (lane & 15)][4 s + (lane >> 4)],
//     32 / 128 KiB -- and each MFMA takes its A operand with one conflict-free ds_read_b64.  K = 6 is the first kernel of
//     this path that is NOT bound by HBM: 512 flop per amplitude = 7.0 ms of the 78.6 Tflop/s at 30 qubits against 4.3 ms
//     of HBM at peak (K = 5: 3.5 ms against 4.3: HBM still).
// Mr[(mu_o, c_o, g_o)][(mu_i, c_i, g_i)] = Re M[p_o][p_i] if c_o == c_i, Im if (c_o, c_i) = (1, 0), -Im if (0, 1), with
// p = 4 mu + g in SORTED block-bit order; `to_caller[i]` = the caller's pattern bit of the i-th lowest block bit.
struct DenseMfma2Args {
  double2* amp;
  const double2* mat;       // the caller's 2^K x 2^K complex matrix, row-major M[out][in]
  u64 col_blocks;           // groups of 16 columns: 2^(k - K - 4)
  int pos[6];               // the block's index bits, ascending
  int to_caller[6];         // pattern bit i (sorted order) -> bit of the caller's pattern
  int consec_log2;          // a wave takes runs of 2^consec_log2 CONSECUTIVE column groups (its accesses to one pattern then
                            // cover 2^consec_log2 x 256 contiguous bytes over as many steps), run after run strided through its XCD's region
  u64 skew;                 // (probe knob) XCD x starts x * skew column groups into its region (wrapping): the eight streams out of step
};
template <int K>
__device__ __forceinline__ double dense_mr_entry(const DenseMfma2Args& a, int t, int s, int lane) {
  const int j = lane & 15, g = lane >> 4;
  const int so = 4 * t + (j >> 2), go = j & 3;
  const int po = 4 * (so >> 1) + go, co = so & 1, pi = 4 * (s >> 1) + g, ci = s & 1;
  int ro = 0, ri = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) { ro |= ((po >> i) & 1) << a.to_caller[i]; ri |= ((pi >> i) & 1) << a.to_caller[i]; }
  const double2 z = a.mat[ro * (1 << K) + ri];
  return co == ci ? z.x : (co ? z.y : -z.y);
}
// PF: how the NEXT column group's amplitudes are requested.  1: before this group's products, if there is a next group --
// the compiler's waits must then cover the case without one, and in the steady state the first MFMA waits for the loads
// just requested: the overlap is left to the other waves of the SIMD.  Measured best for K <= 5: what bounds them is the
// memory system's rate for this access shape, and MORE bytes in flight lower it (a true prefetch costs K = 3, 4 3-15 %).
// 2: requested UNCONDITIONALLY (a wave's last group asks for itself again, unused) after one explicit wait for the first
// group's loads: no wait in the MFMA chain, the loads land behind it (K = 6, bound by the matrix cores with 2 waves per
// SIMD: -3 %).  0 (probe build): not ahead at all.  profiles/r05q_dense_knob_scans.txt
template <int K, bool NT, int THREADS, int PF>
__global__ __launch_bounds__(THREADS) void k_dense_mfma2(const DenseMfma2Args a) {
  static_assert(K >= 3 && K <= 6, "dense blocks of 3 .. 6 qubits");
  constexpr int TT = 1 << (K - 3);          // 16-row output tiles
  constexpr int S = 1 << (K - 1);           // k-steps of 4 rows
  constexpr int MU = 1 << (K - 2);          // amplitudes per lane and column
  constexpr bool IN_LDS = K >= 5;
  __shared__ double tab[IN_LDS ? TT * S * 64 : 1];
  const int l = threadIdx.x & 63, j = l & 15, g = l >> 4;
  double A[IN_LDS ? 1 : TT][IN_LDS ? 1 : S];
  if constexpr (IN_LDS) {
    for (int idx = threadIdx.x; idx < TT * S * 64; idx += THREADS) tab[idx] = dense_mr_entry<K>(a, idx / (S * 64), (idx >> 6) % S, idx & 63);
    __syncthreads();
  } else {
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int s = 0; s < S; ++s) A[t][s] = dense_mr_entry<K>(a, t, s, l);
  }
  u64 off[MU];              // (wave-uniform) offsets of the upper pattern bits, in amplitudes
#pragma unroll
  for (int m = 0; m < MU; ++m) {
    u64 o = 0;
#pragma unroll
    for (int i = 0; i < K - 2; ++i) o |= (u64)((m >> i) & 1) << a.pos[i + 2];
    off[m] = o;
  }
  const u64 lane_part = ((u64)(g & 1) << a.pos[0]) | ((u64)(g >> 1) << a.pos[1]);
  // Column groups of a wave: the workgroups are dealt round-robin over the 8 XCDs, so workgroup b works in the b % 8-th
  // contiguous eighth of the column groups (one XCD's L2 sees one region) and the waves of an XCD stride through it.
  const u64 bid = (u64)blockIdx.y * gridDim.x + blockIdx.x, n_blocks = (u64)gridDim.x * gridDim.y;
  const bool split = (n_blocks & 7) == 0 && (a.col_blocks & 7) == 0;
  const u64 region = split ? a.col_blocks >> 3 : a.col_blocks;
  const u64 region_base = split ? (bid & 7) * region : 0;
  const u64 wave_in_region = (split ? bid >> 3 : bid) * (THREADS / 64) + (threadIdx.x >> 6);          // (wave-uniform)
  const u64 run = 1ull << a.consec_log2;
  const u64 run_stride = ((split ? n_blocks >> 3 : n_blocks) * (THREADS / 64) - 1) << a.consec_log2;   // from a run's end to the wave's next run
  u64 cb = wave_in_region << a.consec_log2;
  const u64 rot = split ? ((bid & 7) * a.skew) % region : 0;
  auto column_ptr = [&](u64 col_block) -> double2* {
    u64 cr = col_block + rot;
    if (cr >= region) cr -= region;
    u64 c = (region_base + cr) * 16 + (u64)j;
#pragma unroll
    for (int i = 0; i < K; ++i) { const int p = a.pos[i]; c = ((c >> p) << (p + 1)) | (c & ((1ull << p) - 1)); }
    return a.amp + (c | lane_part);
  };
  if (cb >= region) return;
  double2* p0 = column_ptr(cb);
  double2 x[MU];
#pragma unroll
  for (int m = 0; m < MU; ++m) x[m] = ld_amp<NT>(p0 + off[m]);
  if constexpr (PF == 2) __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0), the other counters left alone (once per wave)
  for (;;) {
    const u64 cb_next = ((cb + 1) & (run - 1)) ? cb + 1 : cb + 1 + run_stride;
    const bool more = cb_next < region;               // (wave-uniform)
    double2* const p1 = more ? column_ptr(cb_next) : p0;
    double2 xn[MU];
    if (PF == 2 || (PF == 1 && more)) {
#pragma unroll
      for (int m = 0; m < MU; ++m) xn[m] = ld_amp<NT>(p1 + off[m]);
    }
    qs_double4_t acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = qs_double4_t{0.0, 0.0, 0.0, 0.0};
    if constexpr (IN_LDS) {
      // A operands from LDS, one k-step ahead of the MFMAs that use them; the scheduling barrier keeps the compiler from
      // hoisting all TT x S reads to the top (it did: 256 VGPRs and 368 spilled at K = 6)
      double cur[TT], nxt[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) cur[t] = tab[(t * S) * 64 + l];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (s + 1 < S) {
#pragma unroll
          for (int t = 0; t < TT; ++t) nxt[t] = tab[(t * S + s + 1) * 64 + l];
        }
        const double b = (s & 1) ? x[s >> 1].y : x[s >> 1].x;
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[t], b, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TT; ++t) cur[t] = nxt[t];
      }
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const double b = (s & 1) ? x[s >> 1].y : x[s >> 1].x;
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[IN_LDS ? 0 : t][IN_LDS ? 0 : s], b, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) st_amp<NT>(p0 + off[2 * t + h], make_double2(acc[t][2 * h], acc[t][2 * h + 1]));
    if (!more) break;
    cb = cb_next;
    p0 = p1;
    if constexpr (PF == 0) {
#pragma unroll
      for (int m = 0; m < MU; ++m) x[m] = ld_amp<NT>(p0 + off[m]);
    } else {
#pragma unroll
      for (int m = 0; m < MU; ++m) x[m] = xn[m];
    }
  }
}

template <int K> constexpr int kDensePf = K == 6 ? 2 : 1;       // the product's choice per K (see PF above)
template <int K, bool NT, int THREADS>
static void launch_dense_mfma2(const DenseMfma2Args& d, unsigned grid, hipStream_t stream, int pf) {
#ifdef QSIM_PROBES
  if (pf == 0) { hipLaunchKernelGGL((k_dense_mfma2<K, NT, THREADS, 0>), dim3(grid), dim3(THREADS), 0, stream, d); return; }
  if (pf == 1) { hipLaunchKernelGGL((k_dense_mfma2<K, NT, THREADS, 1>), dim3(grid), dim3(THREADS), 0, stream, d); return; }
  if (pf == 2) { hipLaunchKernelGGL((k_dense_mfma2<K, NT, THREADS, 2>), dim3(grid), dim3(THREADS), 0, stream, d); return; }
#endif
  (void)pf;
  hipLaunchKernelGGL((k_dense_mfma2<K, NT, THREADS, kDensePf<K>>), dim3(grid), dim3(THREADS), 0, stream, d);
}

// Chunks too small for 16 columns per wave (fewer than 2^(K+4) amplitudes): one 64-thread workgroup per block, the block's
// 2^K amplitudes through LDS.  Correctness path for tiny chunks (tests, chunked runners with small chunk_size).
struct DenseSmallArgs {
  double2* amp;
  const double2* mat;
  int k;
  int pos[6];
  int bit[6];
};
__global__ __launch_bounds__(64) void k_dense_small(const DenseSmallArgs a) {
  __shared__ double2 x[64];
  const int N = 1 << a.k;
  u64 c = blockIdx.x;
  for (int i = 0; i < a.k; ++i) { const int p = a.pos[i]; c = ((c >> p) << (p + 1)) | (c & ((1ull << p) - 1)); }
  const int r = threadIdx.x;
  u64 off = 0;
  for (int i = 0; i < a.k; ++i) off |= (u64)((r >> i) & 1) << a.bit[i];
  if (r < N) x[r] = a.amp[c | off];
  __syncthreads();
  if (r < N) {
    double2 acc = cmul(a.mat[r * N], x[0]);
    for (int col = 1; col < N; ++col) acc = cfma(a.mat[r * N + col], x[col], acc);
    a.amp[c | off] = acc;
  }
}

constexpr int kScratchDoubles = 8192;      // >= kReduceBlocks reduction partials, and a 64 x 64 complex matrix (dense blocks)
static_assert(kScratchDoubles >= kReduceBlocks, "the scratch holds the reduction partials");
static int ensure_scratch(qsim_chunk* c) {
  if (!c->scratch) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc((void**)&c->scratch, sizeof(double) * kScratchDoubles));
  }
  return QSIM_OK;
}

static unsigned stream_grid(u64 n) {
  const u64 want = (n + kBlock - 1) / kBlock;
  return (unsigned)std::min<u64>(std::max<u64>(want, 1), 8192);
}
