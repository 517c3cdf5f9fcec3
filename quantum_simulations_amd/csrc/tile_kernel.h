// tile_kernel.h -- the fused LDS-tile kernel k_tile and the record stream it interprets.
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
// fixed registers, applies every gate of the group on them, and writes them back once -- LDS
// traffic is paid per group, not per gate.
// The compute phase (everything between "tile in LDS" and "tile final in LDS") is the gate engine
// of tile_engine_gen.h: a banked, direct-threaded interpreter of the pass's record stream written
// in gfx950 assembly (generator: gen_tile_engine.py, which documents the design).  The r01 C++
// form of this loop was bound by instruction issue: ~27 scalar instructions per gate for decode and
// a 7-level compare/branch tree, ~45 per group change for LDS addresses.
// Algorithmic bytes per pass: 32 B x 2^k (every amplitude read and written once), for g gates.
#include "tile_engine_gen.h"

#ifndef QSIM_TILE_LOW
#define QSIM_TILE_LOW 3
#endif
constexpr int kTileLow = QSIM_TILE_LOW;
#ifndef QSIM_TILE_BITS_MAX
#define QSIM_TILE_BITS_MAX 11
#endif
constexpr int kTileBitsMax = QSIM_TILE_BITS_MAX;       // 2^11 amplitudes = 32 KiB of LDS: 4 workgroups per CU (1280-byte LDS granules, see tile_waves)
constexpr int kGroupBits = 3;
#ifndef QSIM_TILE_THREAD_BITS
#define QSIM_TILE_THREAD_BITS 8                        // (9 with QSIM_TILE_BITS_MAX = 12: 512 threads on a 64 KiB tile, probe)
#endif
constexpr int kTileThreadBits = QSIM_TILE_THREAD_BITS;
constexpr int kTileThreads = 1 << kTileThreadBits;
constexpr int kTileMaxQubits = 35;     // outer predicates are 32-bit masks over index bits 3..34

// Entries of the engine's branch table (header dword 0 of a record = 4 x entry).
enum : uint8_t {
  OPC_NOP = QS_ENT_NOP,
  OPC_DENSE1 = QS_ENT_DENSE1,         // +variant 0..8: general 2x2                  (8 doubles: u00 u01 u10 | u11)
  OPC_SWAP1 = QS_ENT_SWAP1,           // +variant: a <-> b                  X, CNOT  (0)
  OPC_ANTI1 = QS_ENT_ANTI1,           // +variant: a' = u01 b, b' = u10 a            (4: u01 u10)
  OPC_PHASE = QS_ENT_PHASE,           // +register mask 0..7: x *= u        T, R, CR (2)
  OPC_DENSE2 = QS_ENT_DENSE2,         // +3*JA + JB: general 4x4, SWAP               (32, none in the record's first 64 bytes)
  OPC_REAL1 = QS_ENT_REAL1,           // +variant: 2x2 with real entries    H, RY, G (4: r00 r01 r10 r11)
  OPC_YLIKE1 = QS_ENT_YLIKE1,         // +variant: [[0,-i],[i,0]]           Y, CY    (0)
  OPC_PHASE_NEG = QS_ENT_PHASE_NEG,   // +mask: x = -x                      Z, CZ    (0)
  OPC_PHASE_I = QS_ENT_PHASE_I,       // +mask: x = i x                     S        (0)
  OPC_PHASE_NI = QS_ENT_PHASE_NI,     // +mask: x = -i x                             (0)
  OPC_DIAGR = QS_ENT_DIAGR,           // +{0: bits 0,1; 1: bits 0,2; 2: bits 1,2; 3: bits 0,1,2}: several phase gates
                                      // that share their predicate, one per register bit, merged by the host:
                                      // x_i *= prod of the listed bits' phases that are set in i
                                      // (6: u_p u_q u_p*u_q; three bits 14: a b c | ab ac bc abc)
  OPC_PRED_OUTER = QS_ENT_PRED_OUTER, // gate with index bits outside the tile that must be 1
  OPC_PRED_LANE = QS_ENT_PRED_LANE,   // gate with tile bits outside the register group that must be 1 (+ outer bits)
  OPC_GROUP = QS_ENT_GROUP,           // register-group change
  OPC_GROUP_FIRST = QS_ENT_GROUP_FIRST,
  OPC_END = QS_ENT_END,
  OPC_HAD1 = QS_ENT_HAD1,             // +variant 0..2 (no control): a' = a + b, b' = a - b          H     (0)
                                      // the factor c of c [[1,1],[1,-1]] is collected over the pass and applied by
  OPC_SCALE = QS_ENT_SCALE,           // every amplitude times a real factor (1 double), last record of the pass
  OPC_ASWAP1 = QS_ENT_ASWAP1,         // +variant: OPC_SWAP1 deferred to the group's write-back -- the LDS addresses of the
                                      // pair trade places (1 instruction per pair instead of 4); only when nothing later
                                      // in the group touches the target or a register control          (0)
  OPC_GROUP_DIRECT = QS_ENT_GROUP_DIRECT,   // first group = the layout the kernel loaded the tile in (TileArgs::lay_in):
                                      // x0..x7 arrive in registers, no LDS read
  OPC_END_DIRECT = QS_ENT_END_DIRECT, // last group = the layout the kernel stores in (lay_out): no LDS write-back
  OPC_PRED_OUTER_ZERO = QS_ENT_PRED_OUTER_ZERO   // gate with index bits outside the tile that must all be 0
};
// 1q variant: 0..2 = target register bit J, no register control; 3 + 2*J + k = control on the
// k-th of the two other register bits (ascending)
static inline int opc_1q_variant(int J, int C) {
  return C < 0 ? J : 3 + 2 * J + ((C > J ? C - 1 : C) == 0 ? 0 : 1);
}

// Record stream.  A record is a 16-byte header followed by its matrix doubles (padded to 16 bytes);
// the engine fetches the first 64 bytes of every record with one s_load_dwordx16, one record ahead.
//   gate:   d0 = 4 x entry (the case itself, or OPC_PRED_OUTER(_ZERO) / OPC_PRED_LANE),
//           d1 = byte offset of the next record from the start of the kernel arguments,
//           d2 = outer predicate (absolute index bits >> kTileLow that must be 1),
//           d3 = lane predicate (tile bits, low half) | 4 x case entry << 16 (second dispatch of a predicated gate)
//   group:  d0 = 4 x OPC_GROUP(_FIRST / _DIRECT), d1 = next, d2..d4 = ~0 << s_i for the group's tile bits s_0 < s_1 < s_2,
//           d5..d11 = LDS byte-address XOR constants of registers x1..x7 (lds_slot is linear over GF(2))
//   end:    d0 = 4 x OPC_END(_DIRECT), d1 = its own offset
constexpr int kTileArgBytes = 4096;
constexpr int kTileStreamOff = 192;                      // byte offset of the first record
constexpr int kTileStreamBytes = kTileArgBytes - kTileStreamOff;
constexpr int kTileStreamSlack = 48;                     // the 64-byte fetch of the END record stays inside the block

// Slab layout as index arithmetic: the removed bits split the index into at most four fields that shift down by
// 0..3 places; the removed bit `bit[j]` lands on physical bit top + j (unused entries: bit 63, which is never set).
struct TileSlab {
  u64 field[4];
  uint8_t bit[3];
  uint8_t top;
  uint8_t pad[4];
};
struct TileArgs {
  double2* amp;            // the state the pass reads (and writes: amp_out == amp unless the pass re-lays the state out)
  int nrec;                // records in the stream incl. END (host bookkeeping; the device follows the stream)
  int T;                   // tile size of the pass (read by the pass-image consumers; the kernel is a template)
  uint8_t h[11];           // ascending absolute positions of the tile's high bits (LOGICAL index bits)
  uint8_t order;           // bits 0-1: tile order of the launch: 0 consecutive, 1 hashed, 2 bit-reversed (see k_tile);
                           // kTileDirectIn / kTileDirectOut: the tile goes global <-> registers without LDS
  uint32_t ntiles;         // 2^(k - T)
  // Thread layout of the global loads / stores: element j of thread tid is index bits
  //   (tid & 7) | sum_i bit(tid, 3 + i) << lay[i]  (i < 5)  |  sum_b bit(j, b) << lay[5 + b].
  // Through LDS (no direct flag) both are h[0..7] in order (element tid + 256 j of the tile).  Direct: lay_in[5..7] are
  // the bits of the FIRST register group (all above the line bits), lay_in[0..4] the other tile bits ascending --
  // what the thread then holds IS the group's x0..x7; lay_out likewise for the LAST group.  PHYSICAL bit positions
  // (= the logical ones unless the pass reads / writes a re-laid-out buffer, below).
  uint8_t lay_in[12];
  uint8_t lay_out[12];
  double2* amp_out;        // where the tiles are stored (== amp: in place)
  // ---- re-layout fused into the pass (round 3; all zero = none) -------------------------------------------------
  // The buffer a pass reads (writes) may hold the state in the SLAB layout of an all-to-all re-layout (qsim_pack_all:
  // slab d = the amplitudes whose m chosen local bits have the pattern d, sent to rank d): a bit permutation of the
  // index -- the m bits move to the top, the others close ranks -- that keeps bits 0..2 (whole 128-B lines).  The
  // LAST local pass before the exchange writes it and the FIRST one after it reads it, instead of two extra HBM
  // passes of the shard (pack, unpack).
  double2* amp_out_own;    // tiles whose logical base has (base & own_mask) == own_value are stored HERE (same layout):
  u64 own_mask, own_value; // the slab that stays on this rank goes straight into the receive buffer
  TileSlab slab_in, slab_out;   // where the tile with logical base b starts in amp / amp_out (kTilePermIn / kTilePermOut)
  uint8_t nbits;           // k: index bits of the chunk
  uint8_t perm;            // kTilePermIn | kTilePermOut | kTileOwnOut
  // ---- a launch over PART of the tiles (round 4; nfix = 0: all tiles) ---------------------------------------------
  // The slab-storing pass of a fused re-layout is launched once per destination slab, so that the exchange of slab d can
  // be posted while the launches of the other slabs still run: the tiles of ONE launch are those whose logical base has
  // the nfix index bits of the slab pattern fixed (fix_or = their values in place); the tile number then enumerates the
  // remaining non-tile bits.  fix_pos[j] (ascending) = position of the j-th fixed bit MINUS the number of tile high bits
  // below it: the zeros of the fixed bits are inserted first, into the number that still lacks the tile bits.
  uint8_t nfix;
  uint8_t fix_pos[3];
  uint8_t reserved0[2];
  u64 fix_or;
  uint8_t reserved[8];
  uint32_t stream[kTileStreamBytes / 4];
};
constexpr uint8_t kTilePermIn = 1, kTilePermOut = 2, kTileOwnOut = 4;
constexpr uint8_t kTileDirectIn = 0x10, kTileDirectOut = 0x20, kTileOrderMask = 0x03;
static_assert(sizeof(TileArgs) == kTileArgBytes, "kernel arguments are one 4 KiB block");
static_assert(offsetof(TileArgs, nbits) % 8 == 0 && offsetof(TileArgs, nfix) == offsetof(TileArgs, nbits) + 2 &&
              offsetof(TileArgs, fix_pos) == offsetof(TileArgs, nbits) + 3 && offsetof(TileArgs, fix_or) == offsetof(TileArgs, nbits) + 8,
              "k_tile<PART> reads {nbits .. fix_pos} and fix_or as two aligned 8-byte words");
static_assert(offsetof(TileArgs, stream) == kTileStreamOff, "record offsets are relative to the argument block");

// XOR-swizzled LDS slot (measured: within 1 % of five other swizzles and of none -- bank
// conflicts are not what limits the gate phase).  Linear over GF(2): the engine relies on it.
__host__ __device__ __forceinline__ unsigned lds_slot(unsigned t) { return t ^ ((t >> 4) & 15u); }

// Waves per SIMD the register allocator may count on = workgroups per CU that the LDS footprint admits (one wave per
// SIMD each).  gfx950 hands out its 160 KiB of LDS in 1280-byte granules, so the 32 KiB of a 2^11 tile takes 26 of the
// 128 and FOUR workgroups fit, not five (tools/occupancy_probe.hip, profiles/r02x_occupancy_probe.txt: 5 at <= 32000 B,
// 4 at 32768 B) -- which leaves 128 VGPRs per wave.  Residency is not what limits the pass: 3 -> 4 workgroups per CU
// gain 2 % (profiles/r02x_ab_residency.txt).  Tried with the room (profiles/r02z_ab_paired_stores.txt): the first
// tile's result kept in registers and both tiles of a workgroup stored back to back -- 1.1 % SLOWER (the stores start
// later); tiles of a workgroup half the state apart instead of adjacent -- 4.7 % slower (adjacent tiles share DRAM pages).
constexpr int tile_waves(int T, int tiles_per_wg = 1) {
  const int granules = ((1 << T) * 16 + 1279) / 1280;
  const int per_simd = (128 / granules) * (kTileThreads / 256);   // workgroups per CU x waves per SIMD of one workgroup
  return per_simd > 5 ? 5 : (per_simd < 1 ? 1 : per_simd);      // (small tiles: 5 as before, their kernels use < 96 VGPRs anyway)
}

// TPW tiles per workgroup (1 or 2): with 2, the second tile's global loads are issued BEFORE the gate engine
// runs on the first and land while it works (the engine issues no vector-memory operation), which hides one
// of the two load latencies of the pair: -3.9 % per pass (profiles/r02r_ab_prefetch_before_engine.txt) at 95
// VGPRs (the engine's registers were packed into v4..v61 and the C++ part keeps no address VGPRs for that).
// Other shapes measured on the same device and session (profiles/r02b_ab_persist.txt, r02d_ab_tiles_per_wg.txt):
// a resident grid walking all tiles is 29 % SLOWER (2.47 vs 1.92 ms per pass: its workgroups run their load /
// compute / store phases in lockstep, so HBM idles while they compute; freshly dispatched workgroups stagger
// by themselves); 4 / 8 tiles per workgroup give the gain of 2 back; 2 tiles WITHOUT the early loads: +2 %; the
// next tile's loads issued only before the finished tile is stored: no gain;
// s_setprio 3 around the load and store phases (and 2 around register-group changes): 0.0 % (r02o_ab_prio.txt).
//
// WIDE = false: the thread part of an element's byte offset fits 32 bits (its highest index bit is below 28,
// always true for states of up to 31 qubits): every access is `global_* v, voffset, s[base]` with a scalar
// 64-bit base per element row -- no 64-bit vector arithmetic (the r01 form spent ~35 quarter-rate
// v_lshl_add_u64 / v_lshlrev_b64 per wave and pass on addresses).
// PART = true: a launch over the tiles of ONE destination slab (TileArgs::nfix; a separate instantiation, so the
// all-tiles kernel keeps its register allocation: with the fixed-bit code inside it the compiler spilled 56 scalar
// registers to vector lanes instead of 22).
template <int T, bool NT, bool WIDE, int TPW, bool PART = false>
__global__ __launch_bounds__(kTileThreads, tile_waves(T, TPW)) void k_tile(const TileArgs a) {
  constexpr int N = 1 << T;
  constexpr int LOW = kTileLow;
  constexpr int NH = T - LOW;                         // tile high bits
  constexpr int BLOCK = kTileThreads;
  constexpr int TB = kTileThreadBits;                 // thread id bits: LOW element bits + row bits
  constexpr int PER = N / BLOCK;                      // tile elements per thread
  constexpr int NBLK = N >> kGroupBits;               // register blocks per tile (<= BLOCK)
  static_assert(N >= BLOCK && NBLK <= BLOCK, "one register block per thread, at least one element per thread");
#ifndef QSIM_LDS_PAD_ELEMS
#define QSIM_LDS_PAD_ELEMS 0                          // (probe: extra LDS per workgroup lowers the residency)
#endif
  __shared__ double2 lds[N + QSIM_LDS_PAD_ELEMS];     // the only LDS object: the engine addresses it from 0
#ifdef QSIM_PROBES
  // in-kernel stamps (probe build): entry / tile loaded / engine done / stores issued, per sampled workgroup
  unsigned long long* const stamps = *reinterpret_cast<unsigned long long* const*>(&a.stream[(kTileArgBytes - 8 - kTileStreamOff) / 4]);
  const unsigned long long t_entry = __builtin_readcyclecounter();
  unsigned long long t_loaded = 0, t_engine = 0;
#endif
  const int tid = threadIdx.x;
  int hs[NH];                                         // the tile's high bits, pinned to scalar registers
#pragma unroll
  for (int j = 0; j < NH; ++j) hs[j] = __builtin_amdgcn_readfirstlane((int)a.h[j]);
  using off_t = typename std::conditional<WIDE, u64, unsigned>::type;   // !WIDE: every thread-part bit is < 28
  constexpr int NTB = (TB - LOW) < NH ? (TB - LOW) : NH;                // tile high bits that are thread bits
  const bool din = (a.order & kTileDirectIn) != 0, dout = (a.order & kTileDirectOut) != 0;   // (uniform)
  // element j of thread tid in a layout `lay` (TileArgs::lay_in / lay_out): the thread part of the offset is
  // computed once per kernel, the j part is wave-uniform (scalar registers)
  auto thread_off = [&](const uint8_t* lay) -> off_t {
    off_t o = tid & ((1 << LOW) - 1);
#pragma unroll
    for (int i = 0; i < NTB; ++i) o |= (off_t)((tid >> (LOW + i)) & 1) << __builtin_amdgcn_readfirstlane((int)lay[i]);
    return o;
  };
  const off_t tin = thread_off(a.lay_in), tout = thread_off(a.lay_out);
  int rin[NH - NTB + 1], rout[NH - NTB + 1];          // (+1: no zero-length arrays)
#pragma unroll
  for (int i = NTB; i < NH; ++i) {
    rin[i - NTB] = __builtin_amdgcn_readfirstlane((int)a.lay_in[i]);
    rout[i - NTB] = __builtin_amdgcn_readfirstlane((int)a.lay_out[i]);
  }
  // Which tile a workgroup takes: TPW consecutive tiles per workgroup; order 0 = consecutive tiles in flight,
  // 1 = hashed, 2 = bit-reversed (probe build only, see tile_order_for).  ntiles is a power of two.
  const unsigned nfix = PART ? __builtin_amdgcn_readfirstlane((unsigned)a.nfix) : 0u;   // fixed index bits of a partial launch
  auto tile_base = [&](unsigned i) -> u64 {
    // (consecutive tiles per workgroup, workgroups dealt round-robin over the XCDs; measured against it,
    // profiles/r02z_*: one contiguous eighth of the tiles per XCD +1.1 %, the two tiles half the state apart +4.7 %,
    // a staggered start of the first round of workgroups 0.0 %, the first two tiles' loads issued together / two tiles
    // ahead with four per workgroup 0.0 % / still +1 %, tiles stored into a second buffer instead of in place +6 %)
    unsigned tile = blockIdx.x * TPW + i;
#ifdef QSIM_PROBES
    if ((a.order & kTileOrderMask) == 1) tile = (tile * 0x9E3779B1u) & (a.ntiles - 1);
    if ((a.order & kTileOrderMask) == 2) tile = a.ntiles > 1 ? __brev(tile) >> (__clz(a.ntiles) + 1) : 0;
#endif
    u64 base = (u64)tile << LOW;                      // the tile number enumerates the non-tile bits
    u64 fix_or = 0;
    if (PART && nfix) {                               // (uniform; a launch over the tiles of one piece)
      // two aligned 8-byte scalar loads from the kernel-argument segment: {nbits, perm, nfix, fix_pos[3], -, -} and fix_or
      // (read where they are used: nothing of them lives across the gate engine; scalar loads ignore the low address
      // bits, so the fields are never addressed byte-wise)
      typedef __attribute__((address_space(4))) const u64 cu64_t;
      cu64_t* fp = (cu64_t*)((__attribute__((address_space(4))) const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TileArgs, nbits));
      asm volatile("" : "+s"(fp));
      const u64 w0 = fp[0];
      fix_or = fp[1];
      for (unsigned j = 0; j < nfix; ++j) {
        const int p = (int)((w0 >> (8 * (3 + j))) & 0xffu);
        base = ((base >> p) << (p + 1)) | (base & ((1ull << p) - 1));
      }
    }
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int p = hs[j];
      base = ((base >> p) << (p + 1)) | (base & ((1ull << p) - 1));
    }
    return base | fix_or;                             // LOGICAL index of the tile's first amplitude
  };
  // Re-laid-out buffers (TileArgs::slab_in / slab_out): the tile's position from its logical base -- wave-uniform
  // scalar work (~25 instructions), only in the passes that carry a layout.  (The pointer is made opaque so that the
  // loads of the masks stay where they are used: hoisted above the gate engine, which clobbers most scalar registers,
  // they would be spilled.)
  const unsigned perm = __builtin_amdgcn_readfirstlane((unsigned)a.perm);
  typedef __attribute__((address_space(4))) const TileSlab cslab_t;   // read through the kernel-argument segment pointer:
  auto slab_base = [&](u64 lbase, unsigned off) -> u64 {               // taking &a.slab_in would copy the 4 KiB block to scratch
    cslab_t* sp = (cslab_t*)((__attribute__((address_space(4))) const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
    asm volatile("" : "+s"(sp));
    u64 o = (lbase & sp->field[0]) | ((lbase & sp->field[1]) >> 1) | ((lbase & sp->field[2]) >> 2) | ((lbase & sp->field[3]) >> 3);
#pragma unroll
    for (int j = 0; j < 3; ++j) o |= ((lbase >> sp->bit[j]) & 1ull) << (sp->top + j);
    return o;
  };
  auto base_in = [&](u64 lbase) -> u64 { return (perm & kTilePermIn) ? slab_base(lbase, (unsigned)offsetof(TileArgs, slab_in)) : lbase; };
  auto base_out = [&](u64 lbase) -> u64 { return (perm & kTilePermOut) ? slab_base(lbase, (unsigned)offsetof(TileArgs, slab_out)) : lbase; };
  typedef double amp_t __attribute__((ext_vector_type(2)));   // one amplitude as a 128-bit register operand of the engine
  typedef __attribute__((address_space(1))) amp_t gamp_t;     // ... in global memory: global_*, not flat_* (a flat access
                                                              // also counts on lgkmcnt, which the engine waits on per record)
  auto element = [&](u64 base, int j, off_t toff, const int* rbits, double2* amp) -> gamp_t* {
    u64 oj = 0;
#pragma unroll
    for (int i = 0; i < NH - NTB; ++i) oj |= (u64)((j >> i) & 1) << rbits[i];
    if constexpr (WIDE) {
      return (gamp_t*)(u64)(amp + base + toff + oj);
    } else {
      // wave-uniform row base pinned to scalar registers + 32-bit thread offset: `global_* v, voff, s[row]`,
      // no 64-bit vector address is formed (or kept alive across the engine)
      const u64 r = reinterpret_cast<u64>(amp + base + oj);
      const u64 row = ((u64)__builtin_amdgcn_readfirstlane((unsigned)(r >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)r);
      unsigned tb = toff << 4;
      asm volatile("" : "+v"(tb));   // (opaque: otherwise the zero-extended 64-bit form is kept alive across the engine and spilled)
      return (gamp_t*)((__attribute__((address_space(1))) char*)row + tb);
    }
  };
#ifndef QSIM_TILE_LOAD_NT
#define QSIM_TILE_LOAD_NT 1      // (probe: 0 = plain loads resp. stores even for states beyond the Infinity Cache: +0.8 % / +2.3 %, profiles/r02z_ab_paired_stores.txt)
#endif
#ifndef QSIM_TILE_STORE_NT
#define QSIM_TILE_STORE_NT 1
#endif
  auto load = [&](gamp_t* p) -> amp_t { return (NT && QSIM_TILE_LOAD_NT) ? __builtin_nontemporal_load(p) : *p; };
  auto store = [&](gamp_t* p, amp_t v) { if (NT && QSIM_TILE_STORE_NT) __builtin_nontemporal_store(v, p); else *p = v; };
  auto store_to = [&](u64 lbase) -> double2* {        // (uniform) which buffer this tile is stored in
    return ((perm & kTileOwnOut) && (lbase & a.own_mask) == a.own_value) ? a.amp_out_own : a.amp_out;
  };
  const unsigned slot0 = lds_slot(tid);               // element tid + BLOCK * j sits BLOCK * j slots further (the swizzle uses bits 4-7 only)
  static_assert(BLOCK >= 256, "lds_slot(tid + BLOCK * j) = lds_slot(tid) + BLOCK * j needs BLOCK to be a multiple of 256");
  amp_t* const tile = reinterpret_cast<amp_t*>(lds);
  u64 base = tile_base(0);
  amp_t v[PER];
  {
    const u64 pb = base_in(base);
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = load(element(pb, j, tin, rin, a.amp));
  }
#pragma unroll 1
  for (int it = 0; it < TPW; ++it) {                  // (rolled: one copy of the 92 KiB engine)
    // the engine's x0..x7 (pinned to v[4:35]): the loaded tile itself when the first register group is the load
    // layout (din: the stream starts with OPC_GROUP_DIRECT), otherwise overwritten by the first group's LDS read
    amp_t x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = j < PER ? v[j < PER ? j : 0] : amp_t{0.0, 0.0};
    if (!din) {
#pragma unroll
      for (int j = 0; j < PER; ++j) tile[slot0 + BLOCK * j] = v[j];
      __syncthreads();
    }
    // TPW > 1: the NEXT tile's loads are issued before the gate engine runs and land while it works (the
    // engine issues no vector-memory operation and waits for none)
    const u64 cur = base;
    if (it + 1 < TPW) {
      base = tile_base(it + 1);
      const u64 pb = base_in(base);
#pragma unroll
      for (int j = 0; j < PER; ++j) v[j] = load(element(pb, j, tin, rin, a.amp));
    }
#ifdef QSIM_PROBES
    if (it == 0) t_loaded = __builtin_readcyclecounter();
#endif
    // ---- gate engine: interprets a.stream on the tile (registers and / or LDS); returns after its last barrier ----
    {
      const auto karg = __builtin_amdgcn_kernarg_segment_ptr();
      const unsigned baseh = __builtin_amdgcn_readfirstlane((unsigned)(cur >> LOW));
      if constexpr (NBLK == BLOCK) {
        asm volatile(QS_ENGINE_ASM_FULL
                     : "+{v[4:7]}"(x[0]), "+{v[8:11]}"(x[1]), "+{v[12:15]}"(x[2]), "+{v[16:19]}"(x[3]),
                       "+{v[20:23]}"(x[4]), "+{v[24:27]}"(x[5]), "+{v[28:31]}"(x[6]), "+{v[32:35]}"(x[7])
                     : [tid] "v"(tid), [karg] "s"(karg), [baseh] "s"(baseh), [first] "i"(kTileStreamOff)
                     : QS_ENGINE_CLOBBERS);
      } else {
        asm volatile(QS_ENGINE_ASM_PARTIAL
                     : "+{v[4:7]}"(x[0]), "+{v[8:11]}"(x[1]), "+{v[12:15]}"(x[2]), "+{v[16:19]}"(x[3]),
                       "+{v[20:23]}"(x[4]), "+{v[24:27]}"(x[5]), "+{v[28:31]}"(x[6]), "+{v[32:35]}"(x[7])
                     : [tid] "v"(tid), [karg] "s"(karg), [baseh] "s"(baseh), [first] "i"(kTileStreamOff), [nblk] "s"(NBLK)
                     : QS_ENGINE_CLOBBERS);
      }
    }
#ifdef QSIM_PROBES
    if (it == 0) t_engine = __builtin_readcyclecounter();
#endif
    double2* const dst = store_to(cur);
    const u64 pcur = base_out(cur);
    if (dout) {                                       // (host: only full tiles, PER == 8)
#pragma unroll
      for (int j = 0; j < PER; ++j) store(element(pcur, j, tout, rout, dst), x[j < 8 ? j : 0]);
    } else {
      amp_t w[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) w[j] = tile[slot0 + BLOCK * j];
#pragma unroll
      for (int j = 0; j < PER; ++j) store(element(pcur, j, tout, rout, dst), w[j]);
    }
    // the next tile's first LDS write (by the kernel or by the engine's first group change) must not overtake a
    // wave still reading this tile's last group / result
    if (it + 1 < TPW) __syncthreads();
  }
#ifdef QSIM_PROBES
  if (stamps && tid == 0 && (blockIdx.x & 63) == 0) {
    unsigned long long* o = stamps + 4 * (blockIdx.x >> 6);
    o[0] = t_entry; o[1] = t_loaded; o[2] = t_engine; o[3] = __builtin_readcyclecounter();
  }
#endif
}

// ---- host: logical descriptors of a pass and their serialisation -----------------------------------
struct TileDesc {          // one gate of a register group, before serialisation
  uint8_t opcode;          // case entry
  uint16_t blk_mask;       // tile bits OUTSIDE the group that must be 1
  u64 outer_mask;          // absolute index bits outside the tile that must be 1 ...
  bool outer_zero;         // ... or, when set, that must all be 0 (no lane predicate then)
  int nd;                  // matrix doubles
  double m[32];
};
struct TileGroup {
  int s[3];                // ascending tile bits of the register group
  u64 qmask = 0;           // every qubit its gates touch (planner bookkeeping: groups on disjoint qubits commute)
  std::vector<TileDesc> gates;
};

static inline int desc_bytes(int nd) { return 16 + ((8 * nd + 15) & ~15); }
static inline int desc_bytes(const TileDesc& d) { return desc_bytes(d.nd); }
constexpr int kGroupRecordBytes = 48;
constexpr int kEndRecordBytes = 16;
// bytes available to group and gate records (END and the over-read slack are set aside)
constexpr int kScaleRecordBytes = 32;
constexpr int kTileRecordBudget = kTileStreamBytes - kEndRecordBytes - kTileStreamSlack - kScaleRecordBytes;

// Write the record stream of a pass.  The caller has kept the total inside kTileRecordBudget.
static int serialize_pass(const std::vector<TileGroup>& groups, TileArgs* a) {
  unsigned char* const base = reinterpret_cast<unsigned char*>(a);
  int off = kTileStreamOff;
  int nrec = 0;
  auto put32 = [&](int at, uint32_t v) { std::memcpy(base + at, &v, 4); };
  auto room = [&](int bytes) { return off + bytes + kEndRecordBytes + kTileStreamSlack <= kTileArgBytes; };   // (the SCALE record is inside kTileRecordBudget's reserve)
  bool first = true;
  if (groups.empty()) return fail(QSIM_ERR_INVALID, "internal: a pass needs at least one register group");
  // Global <-> register layouts (TileArgs::lay_in / lay_out).  A full tile whose first (last) register group lies
  // above the line bits is loaded (stored) in that group's layout, skipping one LDS round trip + barrier each;
  // tuning().tile_direct = 0 keeps everything through LDS.
  const int nh = a->T - kTileLow;
  const bool full = a->T == kTileBitsMax && kTileBitsMax - kTileLow == kTileThreadBits;   // 8 amplitudes per thread, 5 + 3 high bits
  auto layout = [&](const TileGroup& g, uint8_t* lay) -> bool {
    const bool direct = full && tuning().tile_direct && g.s[0] >= kTileLow;
    if (!direct) {
      for (int i = 0; i < nh; ++i) lay[i] = a->h[i];
      return false;
    }
    int nt = 0;
    for (int b = kTileLow; b < a->T; ++b)
      if (b != g.s[0] && b != g.s[1] && b != g.s[2]) lay[nt++] = a->h[b - kTileLow];
    for (int i = 0; i < 3; ++i) lay[nt + i] = a->h[g.s[i] - kTileLow];
    return true;
  };
  const bool din = layout(groups.front(), a->lay_in), dout = layout(groups.back(), a->lay_out);
  a->order = (uint8_t)((a->order & kTileOrderMask) | (din ? kTileDirectIn : 0) | (dout ? kTileDirectOut : 0));
  for (const TileGroup& g : groups) {
    if (!room(kGroupRecordBytes)) return fail(QSIM_ERR_INVALID, "internal: pass record stream overflow");
    put32(off + 0, 4u * (first ? (din ? OPC_GROUP_DIRECT : OPC_GROUP_FIRST) : OPC_GROUP));
    put32(off + 4, (uint32_t)(off + kGroupRecordBytes));
    for (int i = 0; i < 3; ++i) put32(off + 8 + 4 * i, ~0u << g.s[i]);
    for (int r = 1; r < 8; ++r) {
      unsigned t = 0;
      for (int i = 0; i < 3; ++i) if ((r >> i) & 1) t |= 1u << g.s[i];
      put32(off + 16 + 4 * r, lds_slot(t) << 4);
    }
    off += kGroupRecordBytes;
    ++nrec;
    first = false;
    for (TileDesc d : g.gates) {
      // no write-back after the last group of a direct-out pass: a sunk swap (OPC_ASWAP1, which acts by swapping the
      // write-back addresses) is applied in place instead -- it commutes with everything after it in its group
      if (dout && &g == &groups.back() && d.opcode >= OPC_ASWAP1 && d.opcode < OPC_ASWAP1 + 9)
        d.opcode = (uint8_t)(OPC_SWAP1 + (d.opcode - OPC_ASWAP1));
      const int bytes = desc_bytes(d);
      if (!room(bytes)) return fail(QSIM_ERR_INVALID, "internal: pass record stream overflow");
      if ((d.outer_mask & ((1ull << kTileLow) - 1)) || (d.outer_mask >> kTileLow) > 0xFFFFFFFFull)
        return fail(QSIM_ERR_INVALID, "internal: outer predicate outside index bits %d..%d", kTileLow, kTileMaxQubits - 1);
      if (d.outer_zero && (d.blk_mask || !d.outer_mask))
        return fail(QSIM_ERR_INVALID, "internal: a zero-predicate needs outer bits and no lane bits");
      const uint32_t entry = d.blk_mask ? OPC_PRED_LANE : (d.outer_mask ? (d.outer_zero ? OPC_PRED_OUTER_ZERO : OPC_PRED_OUTER) : d.opcode);
      put32(off + 0, 4u * entry);
      put32(off + 4, (uint32_t)(off + bytes));
      put32(off + 8, (uint32_t)(d.outer_mask >> kTileLow));
      put32(off + 12, (uint32_t)d.blk_mask | (4u * d.opcode) << 16);
      // matrix layout: the first 6 doubles follow the header; anything beyond them is placed so that it
      // ENDS at the next record (the engine addresses it as next - size): dense 2x2 -> u11, three-phase
      // run -> the 4 products, dense 4x4 -> all 16 entries
      if (d.opcode >= OPC_DENSE2 && d.opcode < OPC_DENSE2 + 9) {
        std::memcpy(base + off + bytes - 256, d.m, 256);
      } else {
        const int head = d.nd < 6 ? d.nd : 6;
        std::memcpy(base + off + 16, d.m, 8 * (size_t)head);
        if (d.nd > 6) std::memcpy(base + off + bytes - 8 * (d.nd - 6), d.m + 6, 8 * (size_t)(d.nd - 6));
      }
      off += bytes;
      ++nrec;
    }
  }
  put32(off + 0, 4u * (dout ? OPC_END_DIRECT : OPC_END));
  put32(off + 4, (uint32_t)off);
  ++nrec;
  a->nrec = nrec;
  return QSIM_OK;
}
