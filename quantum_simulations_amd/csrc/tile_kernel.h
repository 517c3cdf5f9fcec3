// tile_kernel.h -- the fused LDS-tile kernel k_tile and its opcode table.
// Part of the single translation unit qsim_hip.hip (included there, in order; not a standalone header).
// ================================================================== fused tile passes
// One HBM round trip applies MANY gates (the GPU form of the reference's level batching,
// wenbo_engine/circuit/fusion.py:86-142, and of v3's fused independent-gate block,
// parallel_gate_applicator.py:169-204): a workgroup loads a *tile* of 2^T amplitudes into LDS,
// applies every gate of the pass whose target bits are tile bits, and stores the tile back.
//   tile bits = the kTileLow lowest index bits (every global access is a whole 128-B line, NT)
//   + T - kTileLow arbitrary higher bits chosen by the gates of the pass.
//   Control bits and diagonal bits may lie OUTSIDE the tile: they become a per-tile predicate.
// Inside the tile gates are applied in *register groups*: a group owns kGroupBits tile bits;
// each thread pulls the 2^kGroupBits amplitudes that differ in those bits from LDS into eight
// NAMED registers, applies every gate of the group on them, and writes them back once -- LDS
// traffic is paid per group, not per gate.
// The gate loop is instruction-issue bound (rocprofv3: SALU ~ VALU, one scalar unit per CU), so
// the host pre-decodes every gate into ONE opcode byte selecting a straight-line case (every
// kind x register target x register control combination, plus special cases for real matrices,
// Y-like gates and -1 / +-i phases) and ready-made predicate masks.
// Algorithmic bytes per pass: 32 B x 2^k (every amplitude read and written once), for g gates.
#ifndef QSIM_TILE_LOW
#define QSIM_TILE_LOW 3
#endif
constexpr int kTileLow = QSIM_TILE_LOW;
#ifndef QSIM_TILE_BITS_MAX
#define QSIM_TILE_BITS_MAX 11
#endif
constexpr int kTileBitsMax = QSIM_TILE_BITS_MAX;       // 2^11 amplitudes = 32 KiB of LDS: 4-5 workgroups per CU
constexpr int kGroupBits = 3;
#ifndef QSIM_TILE_THREADS
#define QSIM_TILE_THREADS 256
#endif
constexpr int kTileThreads = QSIM_TILE_THREADS;
constexpr int kTileThreadBits = kTileThreads == 64 ? 6 : kTileThreads == 128 ? 7 : kTileThreads == 256 ? 8 : kTileThreads == 512 ? 9 : 10;
constexpr int kTileGateSlots = 144;    // descriptor array                  (2304 B of kernel arguments)
constexpr int kTileMaxGates = 143;     // usable entries incl. group headers: the device reads one entry ahead
constexpr int kTileMaxMat = 104;       // complex matrix pool          (1664 B)

enum : uint8_t {
  OPC_NOP = 0,
  OPC_DENSE1 = 1,     // +variant 0..8: general 2x2                           (pool: 4)
  OPC_SWAP1 = 10,     // +variant: a <-> b                  X, CNOT           (pool: 0)
  OPC_ANTI1 = 19,     // +variant: a' = u01 b, b' = u10 a                     (pool: 4)
  OPC_PHASE = 28,     // +register mask 0..7: x *= m[0]     T, R, CR          (pool: 1)
  OPC_DENSE2 = 36,    // +3*JA + JB: general 4x4, SWAP                        (pool: 16)
  OPC_REAL1 = 45,     // +variant: 2x2 with real entries    H, RY, G          (pool: 2, packed)
  OPC_YLIKE1 = 54,    // +variant: [[0,-i],[i,0]]           Y, CY             (pool: 0)
  OPC_PHASE_NEG = 63, // +mask: x = -x                      Z, CZ             (pool: 0)
  OPC_PHASE_I = 71,   // +mask: x = i x                     S                 (pool: 0)
  OPC_PHASE_NI = 79,  // +mask: x = -i x                                      (pool: 0)
  OPC_DIAGR = 87,     // +{0: bits 0,1; 1: bits 0,2; 2: bits 1,2; 3: bits 0,1,2}: several phase gates
                      // that share their predicate, one per register bit, merged by the host:
                      // x_i *= prod of the listed bits' phases that are set in i   (pool: 2 or 3)
  OPC_GROUP = 0xFE    // group header
};
// 1q variant: 0..2 = target register bit J, no register control; 3 + 2*J + k = control on the
// k-th of the two other register bits (ascending)
static inline int opc_1q_variant(int J, int C) {
  return C < 0 ? J : 3 + 2 * J + ((C > J ? C - 1 : C) == 0 ? 0 : 1);
}

struct alignas(16) TileGate {   // 16 bytes: one s_load_dwordx4
  uint8_t opcode;
  uint8_t count;           // group header: entries in the group
  uint16_t blk_mask;       // gate: tile bits OUTSIDE the group that must be 1;
                           // group header: s0 | s1 << 4 | s2 << 8 (ascending tile bits)
  uint16_t mat;            // gate: first TileArgs::mat entry of its matrix (0 when it has none)
  uint16_t pad;            // 0: the device reads mat | pad << 16 as one dword
  uint64_t outer_mask;     // absolute index bits outside the tile that must be 1
};

// The device reads a descriptor as ONE 128-bit scalar load (a struct copy is split by the
// compiler into per-field loads that each wait for scalar memory: four round trips per gate).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static_assert(sizeof(TileGate) == sizeof(u32x4), "descriptor = one dwordx4");

struct TileArgs {
  double2* amp;
  int ngates;
  int pad;
  uint8_t h[16];           // ascending absolute positions of the tile's high bits
  u32x4 g[kTileGateSlots]; // TileGate images
  double2 mat[kTileMaxMat];
};
static_assert(sizeof(TileArgs) <= 4096, "kernel arguments must fit 4 KiB");

static inline void put_gate(TileArgs* a, int i, const TileGate& g) { std::memcpy(&a->g[i], &g, sizeof g); }
static inline TileGate get_gate(const TileArgs* a, int i) { TileGate g; std::memcpy(&g, &a->g[i], sizeof g); return g; }

__device__ __forceinline__ unsigned insert_zero(unsigned c, int p) {
  return ((c >> p) << (p + 1)) | (c & ((1u << p) - 1));
}
// XOR-swizzled LDS slot (measured: within 1 % of five other swizzles and of none -- bank
// conflicts are not what limits the gate phase)
__device__ __forceinline__ unsigned lds_slot(unsigned t) { return t ^ ((t >> 4) & 15u); }

// Every case of the gate switch must leave x0..x7 in the registers it found them in: when a case
// defines a new value while the old one is still live the register allocator gives the whole PHI
// web of the switch a second register set and EVERY gate pays 32 v_mov_b64 (2/3 of the VALU work of
// a pass, r01f ISA audit).  So the last instruction of each output is inline asm whose destination
// is tied ("+v") to the old register; partial sums live in ordinary temporaries.
#define QS_IP_FMA(D, S, V, C)  asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(D) : "s"(S), "v"(V), "v"(C))    /* D = S*V + C  */
#define QS_IP_FNMA(D, S, V, C) asm volatile("v_fma_f64 %0, -%1, %2, %3" : "+v"(D) : "s"(S), "v"(V), "v"(C))   /* D = -S*V + C */
#define QS_IP_MOV(D, V)        asm volatile("v_mov_b64 %0, %1" : "+v"(D) : "v"(V))                              /* D = V        */
#define QS_IP_NEG(D, V)        asm volatile("v_mul_f64 %0, -1.0, %1" : "+v"(D) : "v"(V))                        /* D = -V       */
#if defined(QSIM_PLAIN_ALL)
#define QS_D1(A, B) { const double2 a_ = A, b_ = B; A = cfma(u01, b_, cmul(u00, a_)); B = cfma(u11, b_, cmul(u10, a_)); }
#else
#define QS_D1(A, B) {                                                                               \
    double t0_ = fma(-u00.y, A.y, u00.x * A.x), t1_ = fma(u00.y, A.x, u00.x * A.y);                 \
    double t2_ = fma(-u10.y, A.y, u10.x * A.x), t3_ = fma(u10.y, A.x, u10.x * A.y);                 \
    t0_ = fma(u01.x, B.x, t0_); t1_ = fma(u01.x, B.y, t1_);                                         \
    t2_ = fma(-u11.y, B.y, fma(u11.x, B.x, t2_)); t3_ = fma(u11.x, B.y, t3_);                       \
    QS_IP_FNMA(A.x, u01.y, B.y, t0_); QS_IP_FMA(A.y, u01.y, B.x, t1_);                              \
    QS_IP_FMA(B.y, u11.y, B.x, t3_); QS_IP_MOV(B.x, t2_); }
#endif
#if defined(QSIM_PLAIN_ALL)
#define QS_AN(A, B) { const double2 a_ = A, b_ = B; A = cmul(u01, b_); B = cmul(u10, a_); }
#else
#define QS_AN(A, B) {                                                                               \
    const double t2_ = fma(-u10.y, A.y, u10.x * A.x), t3_ = fma(u10.y, A.x, u10.x * A.y);           \
    const double t0_ = u01.x * B.x, t1_ = u01.x * B.y;                                              \
    QS_IP_FNMA(A.x, u01.y, B.y, t0_); QS_IP_FMA(A.y, u01.y, B.x, t1_);                              \
    QS_IP_MOV(B.x, t2_); QS_IP_MOV(B.y, t3_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_SW(A, B) { const double2 t_ = A; A = B; B = t_; }
#else
#if defined(QSIM_SWAP_MOV)
#define QS_SW(A, B) { const double t0_ = A.x, t1_ = A.y;                                            \
    QS_IP_MOV(A.x, B.x); QS_IP_MOV(A.y, B.y); QS_IP_MOV(B.x, t0_); QS_IP_MOV(B.y, t1_); }
#else
#define QS_SWAP64(P, Q) {                                                                           \
    unsigned pl_ = __double2loint(P), ph_ = __double2hiint(P), ql_ = __double2loint(Q), qh_ = __double2hiint(Q); \
    asm volatile("v_swap_b32 %0, %1" : "+v"(pl_), "+v"(ql_));                                       \
    asm volatile("v_swap_b32 %0, %1" : "+v"(ph_), "+v"(qh_));                                       \
    P = __hiloint2double(ph_, pl_); Q = __hiloint2double(qh_, ql_); }
#define QS_SW(A, B) { QS_SWAP64(A.x, B.x) QS_SWAP64(A.y, B.y) }
#endif
#endif
#if defined(QSIM_PLAIN_ALL)
#define QS_DR(A, B) { const double2 a_ = A, b_ = B;                                               \
    A = make_double2(fma(u00.y, b_.x, u00.x * a_.x), fma(u00.y, b_.y, u00.x * a_.y));               \
    B = make_double2(fma(u01.y, b_.x, u01.x * a_.x), fma(u01.y, b_.y, u01.x * a_.y)); }
#else
// real 2x2, packed by the host into two pool entries: u00 = (r00, r01), u01 = (r10, r11)
#define QS_DR(A, B) {                                                                               \
    const double tx_ = u00.x * A.x, ty_ = u00.x * A.y, sx_ = u01.x * A.x, sy_ = u01.x * A.y;        \
    QS_IP_FMA(A.x, u00.y, B.x, tx_); QS_IP_FMA(A.y, u00.y, B.y, ty_);                               \
    QS_IP_FMA(B.x, u01.y, B.x, sx_); QS_IP_FMA(B.y, u01.y, B.y, sy_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_YL(A, B) { const double2 a_ = A, b_ = B; A = make_double2(b_.y, -b_.x); B = make_double2(-a_.y, a_.x); }
#else
#define QS_YL(A, B) { const double t0_ = A.x, t1_ = A.y;                                            \
    QS_IP_MOV(A.x, B.y); QS_IP_NEG(A.y, B.x); QS_IP_NEG(B.x, t1_); QS_IP_MOV(B.y, t0_); }
#endif
#if defined(QSIM_PLAIN_ALL)
#define QS_PH(A) { A = cmul(u00, A); }
#else
#define QS_PH(A) { const double p_ = u00.y * A.x, t_ = u00.x * A.x;                                 \
    QS_IP_FNMA(A.x, u00.y, A.y, t_); QS_IP_FMA(A.y, u00.x, A.y, p_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_PN(A) { A = make_double2(-A.x, -A.y); }
#else
#define QS_PN(A) { QS_IP_NEG(A.x, A.x); QS_IP_NEG(A.y, A.y); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_PI(A) { A = make_double2(-A.y, A.x); }
#else
#define QS_PI(A) { const double t_ = A.x; QS_IP_NEG(A.x, A.y); QS_IP_MOV(A.y, t_); }
#endif
#if defined(QSIM_PLAIN_PERM) || defined(QSIM_PLAIN_ALL)
#define QS_PM(A) { A = make_double2(A.y, -A.x); }
#else
#define QS_PM(A) { const double t_ = A.x; QS_IP_MOV(A.x, A.y); QS_IP_NEG(A.y, t_); }
#endif
// A *= W with W in vector registers (a product of two pool entries)
#define QS_IP_FMA_V(D, S, V, C)  asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(D) : "v"(S), "v"(V), "v"(C))
#define QS_IP_FNMA_V(D, S, V, C) asm volatile("v_fma_f64 %0, -%1, %2, %3" : "+v"(D) : "v"(S), "v"(V), "v"(C))
#define QS_PHS(A, U) { const double p_ = U.y * A.x, t_ = U.x * A.x;                                  \
    QS_IP_FNMA(A.x, U.y, A.y, t_); QS_IP_FMA(A.y, U.x, A.y, p_); }
#define QS_PHV(A, W) { const double p_ = W.y * A.x, t_ = W.x * A.x;                                  \
    QS_IP_FNMA_V(A.x, W.y, A.y, t_); QS_IP_FMA_V(A.y, W.x, A.y, p_); }
// register pairs (bit J clear / set) of each 1q variant
#define QS_PAIRS_0(OP) OP(x0, x1) OP(x2, x3) OP(x4, x5) OP(x6, x7)
#define QS_PAIRS_1(OP) OP(x0, x2) OP(x1, x3) OP(x4, x6) OP(x5, x7)
#define QS_PAIRS_2(OP) OP(x0, x4) OP(x1, x5) OP(x2, x6) OP(x3, x7)
#define QS_PAIRS_3(OP) OP(x2, x3) OP(x6, x7)
#define QS_PAIRS_4(OP) OP(x4, x5) OP(x6, x7)
#define QS_PAIRS_5(OP) OP(x1, x3) OP(x5, x7)
#define QS_PAIRS_6(OP) OP(x4, x6) OP(x5, x7)
#define QS_PAIRS_7(OP) OP(x1, x5) OP(x3, x7)
#define QS_PAIRS_8(OP) OP(x2, x6) OP(x3, x7)
#define QS_CASES_1Q(BASE, OP)                                                             \
  case BASE + 0: QS_PAIRS_0(OP) break;  case BASE + 1: QS_PAIRS_1(OP) break;             \
  case BASE + 2: QS_PAIRS_2(OP) break;  case BASE + 3: QS_PAIRS_3(OP) break;             \
  case BASE + 4: QS_PAIRS_4(OP) break;  case BASE + 5: QS_PAIRS_5(OP) break;             \
  case BASE + 6: QS_PAIRS_6(OP) break;  case BASE + 7: QS_PAIRS_7(OP) break;             \
  case BASE + 8: QS_PAIRS_8(OP) break;
#define QS_CASES_PHASE(BASE, OP)                                                                  \
  case BASE + 0: OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7) break;                   \
  case BASE + 1: OP(x1) OP(x3) OP(x5) OP(x7) break;  case BASE + 2: OP(x2) OP(x3) OP(x6) OP(x7) break; \
  case BASE + 3: OP(x3) OP(x7) break;                case BASE + 4: OP(x4) OP(x5) OP(x6) OP(x7) break; \
  case BASE + 5: OP(x5) OP(x7) break;                case BASE + 6: OP(x6) OP(x7) break;          \
  case BASE + 7: OP(x7) break;
// 4x4 on (v00, v01 = qb set, v10 = qa set, v11); the 16 entries are read where they are used
#define QS_D2(V0, V1, V2, V3) {                                                                         \
    const double2 a_ = V0, b_ = V1, c_ = V2, d_ = V3;                                                   \
    const double2 r0_ = cfma(a.mat[mq + 3], d_, cfma(a.mat[mq + 2], c_, cfma(a.mat[mq + 1], b_, cmul(a.mat[mq + 0], a_))));     \
    const double2 r1_ = cfma(a.mat[mq + 7], d_, cfma(a.mat[mq + 6], c_, cfma(a.mat[mq + 5], b_, cmul(a.mat[mq + 4], a_))));     \
    const double2 r2_ = cfma(a.mat[mq + 11], d_, cfma(a.mat[mq + 10], c_, cfma(a.mat[mq + 9], b_, cmul(a.mat[mq + 8], a_))));   \
    const double2 r3_ = cfma(a.mat[mq + 15], d_, cfma(a.mat[mq + 14], c_, cfma(a.mat[mq + 13], b_, cmul(a.mat[mq + 12], a_)))); \
    QS_IP_MOV(V0.x, r0_.x); QS_IP_MOV(V0.y, r0_.y); QS_IP_MOV(V1.x, r1_.x); QS_IP_MOV(V1.y, r1_.y);     \
    QS_IP_MOV(V2.x, r2_.x); QS_IP_MOV(V2.y, r2_.y); QS_IP_MOV(V3.x, r3_.x); QS_IP_MOV(V3.y, r3_.y); }

// min waves per SIMD asked of the register allocator: what the LDS footprint admits, capped at 4
// (5 forces spills at T = 11 and measured slower)
constexpr int tile_waves(int T) {
  return (160 * 1024) / ((1 << T) * 16) > 4 ? 4 : (160 * 1024) / ((1 << T) * 16);
}

// One workgroup per tile (a resident grid that prefetched the next tile into registers during the
// gate phase was measured twice and was never faster: 163 VGPRs -> 3 workgroups per CU).
template <int T, bool NT>
__global__ __launch_bounds__(kTileThreads, tile_waves(T)) void k_tile(const TileArgs a) {
  constexpr int N = 1 << T;
  constexpr int LOW = kTileLow;
  constexpr int NH = T - LOW;                         // tile high bits
  constexpr int BLOCK = kTileThreads;
  constexpr int TB = kTileThreadBits;                 // thread id bits: LOW element bits + row bits
  constexpr int PER = N / BLOCK > 0 ? N / BLOCK : 1;  // tile elements per thread
  constexpr int NBLK = N >> kGroupBits;               // register blocks per tile (<= BLOCK)
  static_assert(NBLK <= BLOCK, "one register block per thread");
  __shared__ double2 lds[N];
  const int tid = threadIdx.x;
  const bool elem_ok = N >= BLOCK || tid < N;         // tiny tiles: surplus threads idle
  // global index of a tile's element 0: the tile number enumerates the non-tile bits
  auto tile_base = [&](unsigned tile_no) -> u64 {
    u64 base = (u64)tile_no << LOW;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int p = a.h[j];
      base = ((base >> p) << (p + 1)) | (base & ((1ull << p) - 1));
    }
    return base;
  };
  // element t = tid + BLOCK * j -> row = (tid >> LOW) | (j << (TB - LOW)): the thread part of the
  // offset is computed once, the j part is wave-uniform (scalar registers)
  u64 off_tid = tid & ((1 << LOW) - 1);
#pragma unroll
  for (int i = 0; i < TB - LOW && i < NH; ++i) off_tid |= (u64)((tid >> (LOW + i)) & 1) << a.h[i];
  auto off_j = [&](int j) -> u64 {
    u64 o = 0;
#pragma unroll
    for (int i = TB - LOW; i < NH; ++i) o |= (u64)((j >> (i - (TB - LOW))) & 1) << a.h[i];
    return o;
  };
  const u64 base = tile_base(blockIdx.x);
  {
    double2 v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) if (elem_ok) v[j] = ld_amp<NT>(a.amp + base + off_tid + off_j(j));
#pragma unroll
    for (int j = 0; j < PER; ++j) if (elem_ok) lds[lds_slot(tid + BLOCK * j)] = v[j];
  }
  __syncthreads();

  const bool live = NBLK == BLOCK || tid < NBLK;
  // Descriptors are read ONE ENTRY AHEAD: a gate's chain was s_load descriptor -> wait -> s_load
  // matrix -> wait -> dispatch, and above ~32 gates a pass is bound by that per-gate latency
  // (0.03 ms per descriptor at 28 qubits); with the next descriptor already in flight the matrix
  // load and the descriptor load of the following gate share one wait behind the dispatch.
  int gi = 0;
  u32x4 nxt = a.g[0];
  while (gi < a.ngates) {
    gi = __builtin_amdgcn_readfirstlane(gi);          // keep the descriptor reads scalar
    const u32x4 hd = nxt;                             // group header: opcode | count << 8 | bits << 16
    nxt = a.g[(unsigned)(gi + 1) & 0xFF];
    ++gi;
    const unsigned hb = hd.x >> 16;
    const int s0 = hb & 15, s1 = (hb >> 4) & 15, s2 = (hb >> 8) & 15;
    const int ge = gi + ((hd.x >> 8) & 0xFF);
    const unsigned tb = insert_zero(insert_zero(insert_zero((unsigned)tid, s0), s1), s2);
    const unsigned b0 = 1u << s0, b1 = 1u << s1, b2 = 1u << s2;
    // (zero for the surplus threads of tiny tiles: they run the gate cases too, but never write back)
    double2 x0 = make_double2(0.0, 0.0), x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0;
    if (live) {
      x0 = lds[lds_slot(tb)];            x1 = lds[lds_slot(tb | b0)];
      x2 = lds[lds_slot(tb | b1)];       x3 = lds[lds_slot(tb | b1 | b0)];
      x4 = lds[lds_slot(tb | b2)];       x5 = lds[lds_slot(tb | b2 | b0)];
      x6 = lds[lds_slot(tb | b2 | b1)];  x7 = lds[lds_slot(tb | b2 | b1 | b0)];
    }
    // The loop below runs on the CU's single scalar unit for every wave (SALU ~ 2x VALU per gate,
    // r01f ISA audit), so the bookkeeping is kept to: one add + one 128-bit load for the
    // descriptor (unsigned index), and the lane predicate evaluated only for gates that have one.
    for (unsigned q0 = (unsigned)gi, qe = (unsigned)ge; q0 != qe; ++q0) {
      const unsigned q = __builtin_amdgcn_readfirstlane(q0);
      const u32x4 g = nxt;                            // loaded one iteration ago
      nxt = a.g[(q + 1) & 0xFF];                      // one s_load_dwordx4 (index the kernarg arrays directly:
                                                      // a pointer formed into them turns the loads into vector loads)
      const int mq = g.y & 0xFFFF;                    // (known-small indices fold into the load's offset); pool keeps 3 spare entries
      const double2 u00 = a.mat[mq], u01 = a.mat[mq + 1], u10 = a.mat[mq + 2], u11 = a.mat[mq + 3];
      const u64 outer = (u64)g.z | ((u64)g.w << 32);
      if ((base & outer) != outer) continue;
      // Lane predicate (control / diagonal bits that are tile bits outside the group).  EXEC is
      // narrowed by hand: a compiler-managed divergent region makes StructurizeCFG rewrite the
      // (uniform) opcode switch into flow blocks whose PHIs double-buffer x0..x7 (see QS_IP_*).
      // Everything up to the restore is VALU on x0..x7 / case-local temporaries + scalar branches.
      // (Threads that are not `live` -- tiles smaller than 8 x blockDim -- compute on registers
      // they never write back.)
      const unsigned bm = g.x >> 16;
      if (bm) {
        const u64 act = __builtin_amdgcn_ballot_w64((tb & bm) == bm);
        if (act == 0) continue;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_mov_b64 exec, %0" : : "s"(act) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      switch (g.x & 0xFF) {
        QS_CASES_1Q(OPC_DENSE1, QS_D1)
        QS_CASES_1Q(OPC_SWAP1, QS_SW)
        QS_CASES_1Q(OPC_ANTI1, QS_AN)
        QS_CASES_1Q(OPC_REAL1, QS_DR)
        QS_CASES_1Q(OPC_YLIKE1, QS_YL)
        QS_CASES_PHASE(OPC_PHASE, QS_PH)
        QS_CASES_PHASE(OPC_PHASE_NEG, QS_PN)
        QS_CASES_PHASE(OPC_PHASE_I, QS_PI)
        QS_CASES_PHASE(OPC_PHASE_NI, QS_PM)
        case OPC_DIAGR + 0: { const double2 w_ = cmul(u00, u01);     // bits 0 (u00) and 1 (u01)
          QS_PHS(x1, u00) QS_PHS(x5, u00) QS_PHS(x2, u01) QS_PHS(x6, u01) QS_PHV(x3, w_) QS_PHV(x7, w_) } break;
        case OPC_DIAGR + 1: { const double2 w_ = cmul(u00, u01);     // bits 0 (u00) and 2 (u01)
          QS_PHS(x1, u00) QS_PHS(x3, u00) QS_PHS(x4, u01) QS_PHS(x6, u01) QS_PHV(x5, w_) QS_PHV(x7, w_) } break;
        case OPC_DIAGR + 2: { const double2 w_ = cmul(u00, u01);     // bits 1 (u00) and 2 (u01)
          QS_PHS(x2, u00) QS_PHS(x3, u00) QS_PHS(x4, u01) QS_PHS(x5, u01) QS_PHV(x6, w_) QS_PHV(x7, w_) } break;
        case OPC_DIAGR + 3: {                                        // bits 0 (u00), 1 (u01), 2 (u10)
          const double2 w01_ = cmul(u00, u01), w02_ = cmul(u00, u10), w12_ = cmul(u01, u10), w012_ = cmul(w01_, u10);
          QS_PHS(x1, u00) QS_PHS(x2, u01) QS_PHS(x4, u10) QS_PHV(x3, w01_) QS_PHV(x5, w02_) QS_PHV(x6, w12_)
          QS_PHV(x7, w012_) } break;
        case OPC_DENSE2 + 1: QS_D2(x0, x2, x1, x3) QS_D2(x4, x6, x5, x7) break;   // qa = bit 0, qb = bit 1
        case OPC_DENSE2 + 2: QS_D2(x0, x4, x1, x5) QS_D2(x2, x6, x3, x7) break;   // qa = bit 0, qb = bit 2
        case OPC_DENSE2 + 3: QS_D2(x0, x1, x2, x3) QS_D2(x4, x5, x6, x7) break;   // qa = bit 1, qb = bit 0
        case OPC_DENSE2 + 5: QS_D2(x0, x4, x2, x6) QS_D2(x1, x5, x3, x7) break;   // qa = bit 1, qb = bit 2
        case OPC_DENSE2 + 6: QS_D2(x0, x1, x4, x5) QS_D2(x2, x3, x6, x7) break;   // qa = bit 2, qb = bit 0
        case OPC_DENSE2 + 7: QS_D2(x0, x2, x4, x6) QS_D2(x1, x3, x5, x7) break;   // qa = bit 2, qb = bit 1
        default: break;
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_mov_b64 exec, -1" : : : "memory");   // whole waves: blockDim is a multiple of 64
      __builtin_amdgcn_sched_barrier(0);
    }
    if (live) {
      lds[lds_slot(tb)] = x0;            lds[lds_slot(tb | b0)] = x1;
      lds[lds_slot(tb | b1)] = x2;       lds[lds_slot(tb | b1 | b0)] = x3;
      lds[lds_slot(tb | b2)] = x4;       lds[lds_slot(tb | b2 | b0)] = x5;
      lds[lds_slot(tb | b2 | b1)] = x6;  lds[lds_slot(tb | b2 | b1 | b0)] = x7;
    }
    __syncthreads();
    gi = ge;
  }
  {
    double2 w[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) if (elem_ok) w[j] = lds[lds_slot(tid + BLOCK * j)];
#pragma unroll
    for (int j = 0; j < PER; ++j) if (elem_ok) st_amp<NT>(a.amp + base + off_tid + off_j(j), w[j]);
  }
}
