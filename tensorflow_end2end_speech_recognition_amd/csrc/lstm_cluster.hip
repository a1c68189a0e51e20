// LSTM recurrence, multi-CU form: the hidden units of one direction are split over a CLUSTER of
// G = H/64 workgroups (one per CU), so that each CU's slice of W_h is resident in its LDS for the
// whole launch and nothing is re-streamed from L2 per step.
//
// Why (measured): the single-CU kernels of lstm.hip spend 4.2-6.6k of their
// 10.6k cycles/step pulling the 384-448 KB of W_h that does not fit on chip through the one CU's
// L2 port (58 B/clk).  With G CUs the slice is H x 4*64 bf16 = 128 KB (H = 256) -> LDS.
// The price is one all-gather of h (forward) / one reduce-scatter of dh (backward) per step
// between the G CUs.  It is done with the placement-independent hand-off of
// cdna_hip_programming.md G16 "R2": the data IS the flag -- 8-byte {tag = step+1, payload}
// granules written with ONE relaxed agent-scope atomic store and polled with relaxed agent-scope
// atomic loads; no fence, no separate flag, correct for any workgroup->XCD placement.
// Slots are double-buffered by step parity (a producer can only reach step s+1 after every
// consumer has consumed step s-1, see the comment at the poll).  Every spin is bounded: on
// timeout the workgroup raises the error word and keeps going (garbage, but no hang).
//
// Same semantics / data layouts as lstm.hip (gates interleaved [T,B,dir,H,4], etc.).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) __bf16 cbf16x4_t;

constexpr int HS = 64;                 // units per CU
constexpr unsigned SPIN_LIMIT = 4000000u;

__device__ __forceinline__ float cfsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float cftanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  return __builtin_bit_cast(unsigned, (bf2){(__bf16)lo, (__bf16)hi});
}
__device__ __forceinline__ void gstore(u64* p, unsigned epoch, unsigned payload) {
  __hip_atomic_store(p, ((u64)epoch << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gload(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ unsigned long long* g_cdbg = nullptr;   // debug phase timers (env ASR_LSTM_DBG=1)

// 16-byte exchange store of self-tagged words.  Same-XCD cluster: plain store (stays in the shared
// L2); otherwise sc1 (write-through), as the agent-scope atomics lower to.  Inline asm pins the store
// in program order (a plain C++ store could legally sink below the spin loop that follows).
__device__ __forceinline__ void xstore16(f32x4_t* ubase, unsigned voff, f32x4_t v, bool fast) {
  // uniform base in SGPRs + 32-bit per-lane byte offset: no 64-bit pointer registers per slot
  // (s_nop 4 in front: a base restored from a spill lane by v_readlane is a VALU-written SGPR, which a VMEM address may
  // only use 5 wait states later -- the compiler cannot see that hazard inside the block; see the XP publish)
  if (fast) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(ubase) : "memory");
  else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(ubase) : "memory");
}
// ---- forward exchange with 4-byte SELF-TAGGED words (XW) ----
// h = o * tanh(c) lies in [-1, 1], so the top exponent bit of its bf16 form (bit 14) is always 0: the two free bits of a
// packed {bf16, bf16} word (bits 14 and 30) carry a 2-bit step tag and the DATA IS THE FLAG at 4-byte granularity -- no
// separate tag word (half of every polled byte in the 8-byte {tag, payload} granules), tear-proof for any load / store
// width because every 4-byte word validates itself.  tag(s) = ((s >> 1) + 1) & 3 for the slot of parity s & 1: 1 on the
// first write into the zeroed area, a different value on each of the next three writes to the same word (a consumer is
// never more than one write behind, see the poll).  The publisher FORCES the two bits, so a NaN (exponent all ones)
// cannot forge a tag and stall the cluster.
constexpr unsigned XW_MASK = 0x40004000u;
__device__ __forceinline__ unsigned xw_tag(int s) {
  const unsigned t = (((unsigned)s >> 1) + 1u) & 3u;
  return ((t & 1u) << 14) | ((t & 2u) << 29);
}
typedef __attribute__((ext_vector_type(4))) unsigned xw4_t;
// uniform base + per-lane element offset (lets the backend use the SGPR-base addressing mode)
template <typename T>
__device__ __forceinline__ T* uoff(T* ubase, unsigned elem) {
  return ubase + elem;   // plain pointer arithmetic: an integer round trip would demote it to a flat pointer
}

// The exchange area is double-buffered ACROSS launches: launch n runs on area n % 2 and, as its first act, zeroes the
// part of the other area the previous launch dirtied (all workgroups of the grid help: < 3 stores per thread), so the
// next launch finds clean tags without a memset launch in front of every recurrence kernel (10 per training step,
// each ~6 us on the critical path).  The host side tracks the dirty extent of both areas (asr_handle::xch_dirty).
__device__ __forceinline__ void zero_next_area(u64* __restrict__ znext, unsigned zwords) {
  const unsigned nthreads = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < zwords; i += nthreads) znext[i] = 0;
}

// ---------------------------------------------------------------- forward, 8 waves
// Same cluster / exchange as above with TWO waves per SIMD: the 4-wave form spends ~2.2k of its
// 5.2k cycles/step in the gate math of 4 (row, unit) pairs per lane with nothing to hide the
// v_exp/v_rcp latency chains behind.  Here a wave owns 8 units: tile 0 = columns [i x 8 | ci x 8],
// tile 1 = [f x 8 | o x 8]; after the MFMAs lanes n and n^8 swap halves over DPP row_ror:8 so each
// lane ends up with all four gates of TWO (row, unit) pairs; the second wave of the SIMD fills
// the other's stalls.  LDS A-fragment reads double (8 waves x 8 KB) but hide behind the MFMAs.
constexpr int CT8 = 512;
// LDS images [16 rows][...] read as MFMA A fragments with ds_read_b128: row stride = 33 x 16 B.  The
// hardware services a wave in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (rows 0-3,12-15
// of k-block q together with rows 4-11 of k-block q^1), so a plain row-major image always has one
// 2-way bank conflict per group; swapping the two 16-byte halves of every 32 B for rows 4-11 makes
// each group hit 16 distinct 16-byte slots.  Byte offset XOR applied by writers and readers alike.
__device__ __forceinline__ unsigned lds_swz(int row) { return (((row + 4) >> 3) & 1) ? 16u : 0u; }

__device__ __forceinline__ float dpp_ror8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_xor1(float v) {   // quad_perm [1,0,3,2]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

// ---------------------------------------------------------------- cluster placement
// Cluster c = tile * ndir + dir.  Block b of the 1-D grid: XCD slot x = b % 8, q = b / 8,
// member g = q % G, round = q / G, cluster c = round * 8 + x.  grid = 8 * G * ceil(nclusters / 8).
struct ClusterId { int c, g, d, tile; bool valid; };
template <int G>
__device__ __forceinline__ ClusterId cluster_id(int ndir, int ntiles) {
  const int b = blockIdx.x, x = b & 7, q = b >> 3;
  ClusterId r;
  r.g = q % G;
  r.c = (q / G) * 8 + x;
  r.valid = r.c < ndir * ntiles;
  r.d = r.c % ndir;
  r.tile = r.c / ndir;
  return r;
}
static inline unsigned cluster_grid(int G, int nclusters) { return 8u * G * ((nclusters + 7) / 8); }

constexpr int XHDR = 16;                  // header granules per cluster (XCC ids of the members)
constexpr unsigned XCC_EPOCH = 0x80000001u;

// Every member publishes the XCC (= XCD) id it runs on with the placement-independent granule
// protocol and reads all G of them: true iff the whole cluster shares one XCD, hence one L2.
// Same inputs -> same answer on every member.
template <int G>
__device__ __forceinline__ bool same_xcd(u64* hdr, int g, bool& timed_out) {
  const unsigned mine = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;   // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 0) gstore(hdr + g, XCC_EPOCH, mine);
  bool same = true;
#pragma unroll
  for (int i = 0; i < G; ++i) {
    u64 v = gload(hdr + i);
    unsigned spins = 0;
    while ((unsigned)(v >> 32) != XCC_EPOCH) {
      if (++spins > SPIN_LIMIT) { timed_out = true; break; }
      v = gload(hdr + i);
    }
    same = same && ((unsigned)v == mine);
  }
  return same;
}
// Granule publish.  Same-XCD cluster: a PLAIN 8-byte store stays in the shared L2, where the
// consumers' L1-bypassing (sc1) polls find it after ~an L2 round trip; an sc1 store would drop the
// line from L2 and make every poll a fabric round trip (MI355X_MICROARCH.md, store-flavour row).
// Otherwise: the write-through agent-scope store, correct for any placement.
__device__ __forceinline__ void gpublish(u64* p, unsigned epoch, unsigned payload, bool fast) {
  const u64 v = ((u64)epoch << 32) | payload;
  if (fast) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store, no wait
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// masked DPP: lanes whose bank (lane & 15) >> 2 is in `BANKS` take `src` of lane (i + 8) % 16 of
// their row, the others keep `old` -- the halves swap without any v_cndmask
template <int BANKS>
__device__ __forceinline__ float dpp_ror8_into(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                 __builtin_bit_cast(int, src), 0x128, 0xF, BANKS, false));
}

// The kernel is instruction-issue bound (a wave issues ~1 instruction per 4 cycles; measured
// 435 instructions/step -> 3.5k cycles), so the step body is written for instruction count:
// exchange-buffer parity is a compile-time constant (loop unrolled by two), every address is an
// incrementally advanced register, the phase timers (DBG) are a template parameter, nothing in
// the loop is exec-masked.
// EARLY (default at H = 256; ASR_LSTM_DFLAGS bit 5 inverts the default): the k-chunks of this CU's OWN slice of h -- which
// never leave the CU -- are multiplied for step s+1 right after they are written, i.e. before the wave starts polling
// for the peers' slices, so that part of the LDS-read + MFMA phase runs under the L2 hop.  The chunk order in the
// registers is rotated by the CU index so that the own chunks are always register chunks 0 .. KO-1.
// HSU = hidden units per CU (64: eight waves per CU, two per SIMD; 32: four waves per CU, ONE per SIMD -- twice the CUs
// per cluster, each wave alone on its SIMD's VALU / MFMA pipes and with half the LDS fragment traffic per CU).
// XW: the all-gather uses 4-byte self-tagged words (see xw_tag) -- a lane polls with 16-byte loads, each holding one row of
// the 8 units of a peer's wave (ceil((G-1)/4) loads per round instead of G-1 8-byte ones, half the bytes) and stages it
// with one ds_write_b128.  XW = false: 8-byte {step, payload} granules (ASR_LSTM_XW=0 / ASR_LSTM_DFLAGS bit 10).
// ABL (builds with -DASR_LSTM_ABLATE only, scripts/probe_lstm_ablate.py): bit k set = piece k of the step is left out (the
// results are garbage; the launch time against ABL = 0 is what that piece costs ON the serial chain).  0: the poll loop does
// not wait for valid tags, 1: no A-fragment reads from LDS, 2: no MFMAs, 3: no gate math, 4: no saved-activation stores,
// 5: no barriers, 6: no x-projection fetch, 7: no publish, 8: no LDS staging of the polled slices, 9: no poll loads at all.
template <int H, bool DBG, bool EARLY = false, int HSU = 64, int FPIN = 0, bool XW = true, int ABL = 0>
__global__ __launch_bounds__(HSU * 8, 1) void lstm_fwd_cluster8_kernel(
    int T_, int B_, int ndir, const f32x4_t* __restrict__ xg, const bf16_t* __restrict__ whp,
    const float* __restrict__ peep, const int32_t* __restrict__ seq_len, float forget_bias,
    float cell_clip, cbf16x4_t* __restrict__ gates, bf16_t* __restrict__ hout, float* __restrict__ cs,
    float* __restrict__ c_final, float* __restrict__ h_final, u64* __restrict__ xch,
    unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords) {
  zero_next_area(znext, zwords);
  constexpr int G = H / HSU;
  constexpr int CTW = HSU * 8;                             // threads per workgroup: a wave owns 8 units
  constexpr int KS = H / 32;
  constexpr int LDH = H + 8;
  constexpr int SLICE = 16 * (HSU / 2);                     // granules one CU publishes per step
  constexpr int KO = HSU / 32;                              // k-chunks of one CU's own slice
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* hs = reinterpret_cast<bf16_t*>(smem);            // [2][16][LDH]

  // 1-D grid laid out so that the G workgroups of a cluster get block ids congruent mod 8: the
  // dispatcher is observed to put block b on XCD b % 8, so a cluster shares ONE L2 (speed only:
  // the placement is verified below and the exchange falls back to write-through stores)
  // EARLY: k-chunk index modulo KS (a mask where KS is a power of two; H = 320 has 10 chunks: x < 2 KS there)
  auto krot = [](int x) -> int { return ((KS & (KS - 1)) == 0) ? (x & (KS - 1)) : (x >= KS ? x - KS : x); };
  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool lo = col < 8, odd = (col & 1) != 0;
  const bool rev = (d == 1);
  const bf16_t* wp = whp + (size_t)d * H * 4 * H;
  const int ul = wave * 8 + (col & 7);                     // unit inside this CU's slice
  const unsigned jw = g * HSU + ul;                         // global unit of this lane
  const int rbase = rg * 4 + (lo ? 0 : 2);                 // first of this lane's two batch rows

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = seq_len[b0 + rbase + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  float c[2] = {0.f, 0.f}, hr[2] = {0.f, 0.f};
  const float wci = peep ? peep[(d * 3 + 0) * H + jw] : 0.f;
  const float wcf = peep ? peep[(d * 3 + 1) * H + jw] : 0.f;
  const float wco = peep ? peep[(d * 3 + 2) * H + jw] : 0.f;

  for (int i = threadIdx.x; i < 2 * 16 * LDH; i += CTW) hs[i] = 0;
  // B fragments straight out of the standard forward packing (tile = (unit/16)*4 + gate, column
  // unit%16): this lane's column of tile p is (gate p*2 + (col>>3), unit jw)
  bf16x8_t wreg[2][KS];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int tile = (jw >> 4) * 4 + p * 2 + (col >> 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (EARLY) {
        const int kk = krot(ks + g * KO);                  // register chunk ks holds k-chunk kk
        wreg[p][ks] = *reinterpret_cast<const bf16x8_t*>(wp + (((size_t)tile * KS + kk) * 64 + rg * 16 + (jw & 15)) * 8);
      } else {
        wreg[p][ks] = *reinterpret_cast<const bf16x8_t*>(wp + (((size_t)tile * KS + ks) * 64 + rg * 16 + (jw & 15)) * 8);
      }
    }
  }

  u64* xhdr = xch + (size_t)cid.c * (XHDR + 2 * G * SLICE);
  u64* xbase = xhdr + XHDR;                                // [2 parity][G][8 waves][16 rows][4]
  bool timed_out = false;
  const bool colocated = same_xcd<G>(xhdr, g, timed_out);
  const bool fast = colocated && !(kflags & 1);            // same decision on every member
  // after the first timeout the limit drops to 0: the launch drains quickly instead of spinning T times
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY (ASR_LSTM_DFLAGS bit 6): a member goes missing
  __syncthreads();

  // element offsets into the [T,B,ndir,H] arrays: os = frame s (padded rows write their zeros
  // there), oa = frame of the running step of an active row (s or len-1-s); both just advance
  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? 0u - stride : stride;
  unsigned oa[2], os[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    os[r] = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    oa[r] = os[r] + (rev ? (unsigned)max(len[r] - 1, 0) * stride : 0u);
  }
  // per-lane exchange addresses (both parities): publish slot, the three foreign slices this
  // wave polls (wave w polls the granules wave w of the peers publishes), LDS staging targets
  // Rows past their length: their saved activations are never read and their x-projection rows are not used, so both go
  // to ONE parked position per row -- the row's own frame tmax - 1, padding for any row that is ever inactive -- that
  // stays in the L2 instead of streaming the padded part of the batch through HBM (round 4: the "1.87x the algorithmic
  // bytes" of profiles/r03_pmc_hbm.md was exactly this: the batch is 49 % padding).  Only hout keeps its own frames:
  // the next layer reads zeros there.
  unsigned opark[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) opark[r] = os[r] + (unsigned)max(tmax - 1, 0) * stride;
  const int prow = rbase + (odd ? 1 : 0);
  const unsigned pofs = (unsigned)(wave * 64 + prow * 4 + ((col & 7) >> 1));   // publish slot in the slice
  const unsigned lofs = threadIdx.x;                                           // polled granule in a slice
  unsigned ldst[G - 1];                                    // byte offset inside one h buffer
  auto slice = [&](int P, int gg) -> u64* { return xbase + ((size_t)P * G + gg) * SLICE; };   // uniform
#pragma unroll
  for (int k = 0; k < G - 1; ++k) {
    const int gsrc = k + (k >= g ? 1 : 0);
    ldst[k] = ((unsigned)(((lane >> 2) & 15) * LDH + gsrc * HSU + (wave * 4 + (lane & 3)) * 2) * 2u) ^
              lds_swz((lane >> 2) & 15);
  }
  // XW: load j of a lane covers peer k = 4 j + lane / 16 (clamped: the spare lanes of the last load repeat its last peer --
  // same bytes, same LDS target, nothing exec-masked), row lane % 16 of that peer's wave `wave`: 16 bytes = 8 units
  constexpr int NL = XW ? ((G - 1) * 16 + 63) / 64 : 1;
  unsigned* xw = reinterpret_cast<unsigned*>(xbase);       // [2 parity][G][SLICE] words
  const __amdgpu_buffer_rsrc_t xwrs = __builtin_amdgcn_make_buffer_rsrc(xw, 0, 0x7fffffff, 0x00020000);
  unsigned xvoff[NL], xldst[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int k = min(j * 4 + (lane >> 4), G - 2);
    const int gsrc = k + (k >= g ? 1 : 0);
    xvoff[j] = (unsigned)(gsrc * SLICE + wave * 64 + (lane & 15) * 4) * 4u;
    xldst[j] = ((unsigned)((lane & 15) * LDH + gsrc * HSU + wave * 8) * 2u) ^ lds_swz(lane & 15);
  }
  const unsigned lown = ((unsigned)(prow * LDH + g * HSU + (ul & ~1)) * 2u) ^ lds_swz(prow);
  const unsigned lrd = ((unsigned)(col * LDH + rg * 8) * 2u) ^ lds_swz(col);

  // xproj rows of the running step, requested TWO steps ahead (slot = step parity): one step (~1.2 us) covers the
  // idle HBM latency but not always the latency beside the side streams' traffic (measured: 945 -> 931 us per launch)
  // (H = 512 keeps the one-step queue: the second slot costs 8 VGPRs the 8-CU form does not have -- 4 -> 12 spills)
  constexpr int XD = (H <= 320 || HSU == 32) ? 2 : 1;      // (one wave per SIMD may use the whole 512-register file)
  f32x4_t xq[XD][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    xq[0][r] = xg[(0 < len[r]) ? oa[r] : opark[r]];
    if constexpr (XD == 2) xq[1][r] = (tmax > 1) ? xg[(1 < len[r]) ? oa[r] + dstep : opark[r]] : xq[0][r];
  }
  f32x4_t accn0 = {0.f, 0.f, 0.f, 0.f}, accn1 = {0.f, 0.f, 0.f, 0.f};   // EARLY: own-slice part of the next step

  unsigned long long* dbg = g_cdbg;
  unsigned long long ph[4] = {0, 0, 0, 0};
  unsigned long long ph4 = 0;
  unsigned nspin = 0;
  unsigned nonfin = 0;                                     // XW: OR of every published pair (one v_or per step)
#define C8_T() (DBG ? (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime()) : 0ull)
#define C8_FPIN(k) do { if constexpr (!DBG && ((FPIN >> (k)) & 1)) __builtin_amdgcn_sched_barrier(0); } while (0)

  auto step = [&](int s, auto PAR) {
    constexpr int P = decltype(PAR)::value;                // parity of step s
    C8_FPIN(0);
    const unsigned long long t0 = C8_T();
    const char* hcur = smem + P * 16 * LDH * 2;
    char* hnxt = smem + (1 - P) * 16 * LDH * 2;
    const f32x4_t x0 = xq[P % XD][0], x1 = xq[P % XD][1];
    // (requested here, at the top of the step: behind the poll loop at its end -- where the BPTT kernel's fetch belongs,
    // see there -- the forward kernel measured 1.21 ms per launch instead of 0.92)
    if (s + XD < tmax && !(ABL & 64)) {                    // lands during the next step(s)
#pragma unroll
      for (int r = 0; r < 2; ++r)
        xq[P % XD][r] = xg[(s + XD < len[r]) ? oa[r] + (unsigned)XD * dstep : opark[r]];
    }
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EARLY) { acc0 = accn0; acc1 = accn1; }    // own-slice chunks: done at the end of the last step
    // 8 A fragments in flight at once (left alone the scheduler recycles ONE register and exposes
    // the LDS latency KS times: 870 cycles for 16 MFMAs); H = 512 takes two such batches
#pragma unroll
    for (int kb = 0; kb < KS; kb += 8) {
      bf16x8_t afr[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (kb + ks >= KS) continue;                       // H = 320: 10 chunks = 8 + 2
        if constexpr ((ABL & 2) != 0) {
          afr[ks] = wreg[1][kb + ks];
        } else if constexpr (EARLY) {
          if (kb + ks >= KO) afr[ks] = *reinterpret_cast<const bf16x8_t*>(hcur + lrd + krot(kb + ks + g * KO) * 64);
        } else {
          afr[ks] = *reinterpret_cast<const bf16x8_t*>(hcur + lrd + (kb + ks) * 64);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (DBG && kb == 0) {                                // sub-phase: first A fragments landed
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(afr[ks]));
        ph4 += C8_T() - t0;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (EARLY && kb + ks < KO) continue;
        if (kb + ks >= KS) continue;
        if constexpr ((ABL & 4) != 0) {                     // keeps the dependence on every fragment, nothing else
          const xw4_t fb = __builtin_bit_cast(xw4_t, afr[ks]);
          acc0[0] = __uint_as_float(__float_as_uint(acc0[0]) ^ (fb[0] & 1u));
          acc1[0] = __uint_as_float(__float_as_uint(acc1[0]) ^ (fb[3] & 1u));
          continue;
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], wreg[0][kb + ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], wreg[1][kb + ks], acc1, 0, 0, 0);
      }
    }
    if (DBG) { asm volatile("" : "+v"(acc0)); asm volatile("" : "+v"(acc1)); }
    C8_FPIN(1);
    const unsigned long long t1 = C8_T();
    // lanes n / n^8 swap halves: lo lanes (banks 0,1) keep rows 0,1 (own i,f + the partner's ci,o),
    // hi lanes (banks 2,3) rows 2,3 (own ci,o + the partner's i,f)
    float pi[2], pq[2], pf[2], po[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      pi[r] = dpp_ror8_into<0xC>(acc0[r], acc0[2 + r]);    // hi lanes <- partner's i of rows 2,3
      pq[r] = dpp_ror8_into<0x3>(acc0[2 + r], acc0[r]);    // lo lanes <- partner's ci of rows 0,1
      pf[r] = dpp_ror8_into<0xC>(acc1[r], acc1[2 + r]);
      po[r] = dpp_ror8_into<0x3>(acc1[2 + r], acc1[r]);
    }
    C8_FPIN(4);                                            // behind the DPP half swap
    const unsigned epoch = (unsigned)s + 1u;
    bool act[2];
    float ig[2], gg[2], fg[2], og[2], cn[2], hn[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) act[r] = s < len[r];
    const f32x4_t xr[2] = {x0, x1};
    if constexpr ((ABL & 8) != 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        ig[r] = pi[r] + xr[r][0]; gg[r] = pq[r] + xr[r][1]; fg[r] = pf[r] + xr[r][2]; og[r] = po[r] + xr[r][3];
        cn[r] = c[r];
        hn[r] = fminf(fmaxf((ig[r] + gg[r]) + (fg[r] + og[r]), -1.f), 1.f);
      }
    } else {
#pragma unroll
    for (int r = 0; r < 2; ++r) ig[r] = cfsig(pi[r] + xr[r][0] + wci * c[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) gg[r] = cftanh(pq[r] + xr[r][1]);
#pragma unroll
    for (int r = 0; r < 2; ++r) fg[r] = cfsig(pf[r] + xr[r][2] + forget_bias + wcf * c[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      cn[r] = gg[r] * ig[r] + c[r] * fg[r];
      if (cell_clip > 0.f) cn[r] = fminf(fmaxf(cn[r], -cell_clip), cell_clip);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) og[r] = cfsig(po[r] + xr[r][3] + wco * cn[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) hn[r] = cftanh(cn[r]) * og[r];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      c[r] = act[r] ? cn[r] : c[r];
      hr[r] = act[r] ? hn[r] : hr[r];
    }
    C8_FPIN(5);                                            // behind the gate math
    // publish: even lanes the (unit, unit+1) granule of row 0, odd lanes (unit-1, unit) of row 1
    const float nb = dpp_xor1(odd ? hr[0] : hr[1]);
    const unsigned pk = odd ? pack_bf16x2(nb, hr[1]) : pack_bf16x2(hr[0], nb);
    const unsigned etag = xw_tag(s);
    if constexpr ((ABL & 128) != 0) {
    } else if constexpr (XW) {
      unsigned* pw = uoff(xw + (size_t)(P * G + g) * SLICE, pofs);
      nonfin |= pk;                                        // bit 14 / 30 of a FINITE |h| <= 1 is clear (see the tail)
      const unsigned tv = (pk & ~XW_MASK) | etag;
      if (fast) __hip_atomic_store(pw, tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store, stays in the L2
      else __hip_atomic_store(pw, tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      gpublish(uoff(slice(P, g), pofs), epoch, pk, fast);
    }
    C8_FPIN(6);                                            // behind the publish
    *reinterpret_cast<unsigned*>(hnxt + lown) = pk;
    // saved activations; rows past their length write frame s of the padding (hout: zeros,
    // gates / cs: never read there), so nothing is predicated
    unsigned off[2], offg[2];                              // hout position / saved-activation position
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      off[r] = act[r] ? oa[r] : os[r];
      offg[r] = act[r] ? oa[r] : opark[r];
    }
    // H = 256: the saved activations are stored BEHIND the poll loop, see there.  H = 512 stores them here: keeping the
    // values alive across the loop costs registers that form has not got (4 -> 12 spills, 43.3 -> 47.7 ms at cfg D).
    constexpr bool LATE_STORE = (H <= 320 || HSU == 32);
    unsigned offs[2] = {off[0], off[1]}, offgs[2] = {offg[0], offg[1]};
    if constexpr (!LATE_STORE) {
      const bool pact = odd ? act[1] : act[0];
      const unsigned poff = (odd ? off[1] : off[0]) - (odd ? 1u : 0u);
      *reinterpret_cast<unsigned*>(hout + poff) = pact ? pk : 0u;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        gates[offg[r]] = (cbf16x4_t){(__bf16)ig[r], (__bf16)gg[r], (__bf16)fg[r], (__bf16)og[r]};
        cs[offg[r]] = cn[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      oa[r] += dstep;
      os[r] += stride;
    }
    C8_FPIN(2);
    const unsigned long long t2 = C8_T();
    if constexpr (EARLY) {
      if (s + 1 < tmax) {                                  // block-uniform
        if constexpr (!(ABL & 32)) __syncthreads();        // the CU's own slice of h(s) is complete in hnxt
        bf16x8_t ao[KO];
#pragma unroll
        for (int k = 0; k < KO; ++k) ao[k] = *reinterpret_cast<const bf16x8_t*>(hnxt + lrd + krot(k + g * KO) * 64);
        accn0 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        accn1 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KO; ++k) {
          accn0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao[k], wreg[0][k], accn0, 0, 0, 0);
          accn1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ao[k], wreg[1][k], accn1, 0, 0, 0);
        }
      }
    }
    {
      // The first poll round goes out BEHIND the own-slice MFMAs above, not in front of them (left alone the scheduler
      // hoists the poll loads to right behind the barrier, in front of the LDS read and the MFMAs): ~150 cycles later --
      // forward launch 840 -> 805 us at H = 256, 1400 -> 1358 at H = 512, 975 -> 967 at H = 320.  (Later still is worse again: + 128 cycles 818 us,
      // + 256 860, + 384 882; the fp32 kernel, whose own-slice MFMAs take three times as long, gains nothing from it.)
      // (Round 5, measured: TWO poll rounds in flight -- the second issued s_sleep(2 / 4 / 6) behind the first, before it
      // returns -- is slower by exactly the sleep: 693 -> 722 / 765 / 838 us per launch, same bits.  The L2 round trip of a
      // poll is ~200 cycles, so the one-round loop already samples every ~200 cycles; the 0.7 - 1.0 repolls per step of the
      // phase probe cost what they look like, not a fabric trip each.)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (XW) {
        constexpr unsigned pbytes = (unsigned)P * G * SLICE * 4u;   // this parity's slices
        xw4_t v[NL];
        unsigned spins = 0;
#pragma unroll 1
        for (;;) {                                         // wave-uniform loop: no exec masking
          asm volatile("" ::: "memory");                   // every round is a fresh set of loads
          if constexpr ((ABL & 512) != 0) {
#pragma unroll
            for (int j = 0; j < NL; ++j) v[j] = (xw4_t){pk, pk, pk, pk};
            break;
          }
#pragma unroll
          for (int j = 0; j < NL; ++j)                      // L1-bypassing (sc1) 16-byte loads, counted on vmcnt by the compiler
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(xwrs, xvoff[j], pbytes, 16);
          // v ^ tag has the two tag bits clear exactly where the word is this step's -- and is then the clean payload
          unsigned bad = 0;
#pragma unroll
          for (int j = 0; j < NL; ++j) {
            v[j] ^= etag;
            bad |= v[j][0] | v[j][1] | v[j][2] | v[j][3];
          }
          if (__all((bad & XW_MASK) == 0u)) break;
          if constexpr ((ABL & 1) != 0) break;              // one round, whatever it carries
          if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
        }
        if (DBG) nspin += spins;
        C8_FPIN(7);                                        // behind the poll loop
        if constexpr ((ABL & 256) != 0) {
#pragma unroll
          for (int j = 0; j < NL; ++j) asm volatile("" ::"v"(v[j]));
        } else {
#pragma unroll
        for (int j = 0; j < NL; ++j) *reinterpret_cast<xw4_t*>(hnxt + xldst[j]) = v[j];
        }
        C8_FPIN(8);                                        // behind the LDS staging
      } else {
      u64 v[G - 1];
#pragma unroll
      for (int k = 0; k < G - 1; ++k) v[k] = gload(uoff(slice(P, k + (k >= g ? 1 : 0)), lofs));
      unsigned spins = 0;
#pragma unroll 1
      for (;;) {                                           // wave-uniform loop: no exec masking
        bool ok = true;
#pragma unroll
        for (int k = 0; k < G - 1; ++k) ok = ok && ((unsigned)(v[k] >> 32) == epoch);
        if (__all(ok)) break;
        if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
        for (int k = 0; k < G - 1; ++k) v[k] = gload(uoff(slice(P, k + (k >= g ? 1 : 0)), lofs));
      }
      if (DBG) nspin += spins;
      C8_FPIN(7);                                          // behind the poll loop
#pragma unroll
      for (int k = 0; k < G - 1; ++k) *reinterpret_cast<unsigned*>(hnxt + ldst[k]) = (unsigned)v[k];
      C8_FPIN(8);                                          // behind the LDS staging
      }
    }
    // saved activations: stored behind the poll loop -- on gfx950 loads and stores share one in-order counter, so a poll
    // issued after these stores waits for their acknowledgements too (0.921 -> 0.908 ms per launch)
    if constexpr (LATE_STORE && !(ABL & 16)) {
      const bool pact = odd ? act[1] : act[0];
      const unsigned poff = (odd ? offs[1] : offs[0]) - (odd ? 1u : 0u);
      *reinterpret_cast<unsigned*>(hout + poff) = pact ? pk : 0u;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        gates[offgs[r]] = (cbf16x4_t){(__bf16)ig[r], (__bf16)gg[r], (__bf16)fg[r], (__bf16)og[r]};
        cs[offgs[r]] = cn[r];
      }
    }
    C8_FPIN(3);
    const unsigned long long t3 = C8_T();
    if constexpr (!(ABL & 32)) __syncthreads();
    if (DBG) {
      const unsigned long long t4 = C8_T();
      ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;
    }
  };
  int s = 0;
  for (; s + 1 < tmax; s += 2) {
    step(s, std::integral_constant<int, 0>{});
    step(s + 1, std::integral_constant<int, 1>{});
  }
  if (s < tmax) step(s, std::integral_constant<int, 0>{});
#undef C8_T
#undef C8_FPIN

  if (DBG && dbg && lane == 0 && cid.tile == 0 && g < 4) {
    unsigned long long* o = dbg + 256 + ((size_t)(d * 4 + g) * 8 + wave) * 8;   // [256, 768): 8-wave kernels
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = ph[k];
    o[4] = nspin;
    o[5] = tmax;
    o[6] = fast ? 1 : 0;
    o[7] = ph4;
  }
  if (timed_out) atomicOr(err, 1u);
  // XW forces bits 14 / 30 of a published word (the step tag), so a NaN / Inf h (exponent all ones) would reach the
  // peers as a finite value while its owner keeps the NaN: report it instead of laundering it.  h =
  // o * tanh(c) is in [-1, 1], whose bf16 forms all have bit 14 clear; any set bit 14 is a non-finite (or > 1) value.
  if (XW && (nonfin & XW_MASK)) atomicOr(err, 4u);
  // zero-fill the common padded tail [tmax, T): os already points at frame tmax
  for (int t = tmax; t < T_; ++t)
#pragma unroll
    for (int r = 0; r < 2; ++r) { hout[os[r]] = 0; os[r] += stride; }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t o = ((size_t)d * B_ + b0 + rbase + r) * H + jw;
    if (c_final) c_final[o] = c[r];
    if (h_final) h_final[o] = hr[r];
  }
}

// (Round 6, profiles/r06_lstm_ablation.md -- the step with one piece compiled out at a time, ABL above: free-running (no
// poll at all) the forward step is 0.753 of its 0.929 us, i.e. the cross-CU exchange is 19 % of the step; the A-fragment
// reads cost 0.016 - 0.05 us (the "800 exposed LDS cycles" of the round-5 s_memtime table were the timers: s_memtime returns
// on the counter the LDS reads use), the MFMAs 0.10, gate math 0.11 - 0.13, the two barriers 0.11, staging 0.03 - 0.06,
// stores 0.044, x fetch 0.033, and 0.23 is publish / own-slice hop / DPP / addresses.  No piece is above 14 %.)
// (Round 6, built and measured, profiles/r06_helper_wave.md: a FIFTH wave per workgroup that owns the step's big arrays -- it
// fetched the x rows ahead into an LDS image and copied the saved activations from an LDS image to memory, so that the chain
// waves issued no global load / store at all.  Bit-identical.  An idle fifth wave costs nothing (690 vs 694 us per launch);
// the stores through it gain 1.2 % at H = 256 (686 us) and LOSE 18 % at H = 512 (2 004 -> 2 380 us); x through LDS loses
// 12 % whatever the prefetch depth (2 or 4 steps) and wherever the helper writes it (775 - 809 us).  Removed again.)
// (Round 5, built and measured: a forward kernel WITHOUT the LDS image -- the slice layout [wave][row][8 units] is already the
// A operand of v_mfma_f32_16x16x32_bf16, so every wave polled the 16 bytes per lane of ALL G slices and multiplied them as
// they came: no staging, no barrier, no LDS read on the chain.  Correct (same parity figures) and SLOWER: 695 -> 805 - 813 us
// per launch at H = 256.  A poll round is then 4 waves x 8 KB = 32 KB per CU through a 64 B/clk L2 port (~500 cycles)
// instead of 8 KB shared through the LDS (128 B/clk): the staging is what keeps the hop short.  Removed again.)

// ---------------------------------------------------------------- backward
// K-partition: CU g owns the gate gradients of its 64 units (k' = unit*4+q in its slice) and the
// rows k' of W_h^T; each step it multiplies its dG slice by W_h^T[k' slice, all H] -> a partial
// dh_prev for ALL units, keeps the part for its own units and publishes the rest; the partials
// for its own units from the other CUs are gathered at the next step (reduce-scatter).
// ---------------------------------------------------------------- backward, 8 waves
// Built around what bounds these kernels (instruction issue + the cross-CU hop):
//  * a thread owns TWO (row, unit) pairs in the MFMA C layout: wave w = (hh = w >> 2, wt = w & 3),
//    lane (col, rg) -> unit wt*16 + col of the slice, rows rg*4 + hh*2 + {0,1};
//  * the own-unit tile wt is computed by BOTH waves wt and wt+4 (8 extra MFMAs per wave), so the
//    own partial is already in the right registers: no LDS hand-over, ONE barrier per step;
//  * foreign tiles (12 per CU) are spread 2/1 over the waves of each SIMD and published with one
//    16-byte store per lane: 4 rows of one unit, every fp32 word carrying the step's parity tag in
//    its mantissa LSB (the data is the flag, per word, so any store width is tear-proof);
//  * everything that does not depend on dh (tanh(c), the gate-derivative factors) is computed
//    while the polls for the peers' partials are in flight.
// xch per cluster: header + [2 parity][G dst][G src][4 tiles][64 lanes][4 words].
// HSU = hidden units per CU: 64 (eight waves: 4 unit tiles x 2 row halves) or 32 (four waves, ONE per SIMD: 2 unit tiles x
// 2 row halves, twice the CUs per cluster, the whole 512-entry register file per wave).
// PIN: bit k set = a scheduling barrier at phase boundary k of the step (0: top, 1: behind the fetch, 2: in front of the
// step's barrier, 3: end) -- the places where the timer build has its s_memtime reads.  -1: the default for the width.
// Measured, BPTT launch at H = 256 (four waves): no pins 888 us; {1,2} 888; {0,3} 859; {0,1,2} 859; {1,2,3} 859; {0,1,3} 855; all
// four 855 -- what matters is that nothing moves across the STEP boundary (the scheduler otherwise interleaves the next
// iteration's polls and address arithmetic with this iteration's MFMAs and exchange stores).  H = 512: none 1537, {0} 1538,
// {1} 1530, {2} 1529, {1,2} 1528, {3} 1605, {0,3} 1570, all four 1567 -- there the pins stay off.  Further points on top of the
// four at H = 256 (bits 4..8: behind the poll issue / the dh-independent math / the poll loop / the barrier / the gradient
// sums): 858 / 858 / 849 / 856 / 857 against 855 -- bit 6 is kept.  (H = 512 with bit 6 / {1,2,6} / bit 5: 1541 / 1535 / 1536
// against 1530; finer points -- between the foreign-tile chains and their stores / between those stores and the own-tile
// chain / in front of the A-fragment reads: 1575 / 1525 / 1523 against 1519; forward, between the two A-fragment batches:
// 1364 against 1353.)
// XP (default): the reduce-scatter slots are laid out for the CONSUMER -- a destination wave (hh, wt) finds the two rows it
// owns of TWO sources side by side in one 16-byte word per lane, [dst][tile][hh][source pair][lane]: ceil((G-1)/2)
// full-line 16-byte polls per round instead of G-1 8-byte ones that each use half of the 16 bytes per lane they touch
// (8-byte L2 accesses run at 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md).  The producers pay with two 8-byte
// stores per lane and tile (rows 0,1 -> the hh = 0 word, rows 2,3 -> hh = 1) instead of one 16-byte store.  Same words,
// same tags, same summation order: bit-identical to XP = false.  Which form runs: bwd_xp_enabled() (a win only on the
// eight-wave clusters, measured there).
// ABL (builds with -DASR_LSTM_ABLATE only): bit k set = piece k of the step is left out, as in the forward kernel.  0: the
// poll loop does not wait for valid tags, 1: no A-fragment reads from LDS, 2: no MFMAs, 3: no gate-gradient math, 4: no
// dgates store to memory, 5: no barrier, 6: no fetch of the saved activations, 7: no publish stores, 8: no dG write to
// LDS, 9: no poll loads at all.
// CLIPZ (asr_lstm_bwd_ex with bf16 operands, the projected LSTMCell layers): a state whose saved cs has |c| >= clipz passes no
// gradient to its gates or to c_prev (tf.clip_by_value); its own instantiations -- the default kernels' code does not change.
template <int H, bool DBG, int HSU = 64, int PIN = -1, bool XP = true, int ABL = 0, bool CLIPZ = false>
__global__ __launch_bounds__(HSU * 8, 1) void lstm_bwd_cluster8_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dhout, const cbf16x4_t* __restrict__ gates,
    const float* __restrict__ cs, const bf16_t* __restrict__ whpb, const float* __restrict__ peep,
    const int32_t* __restrict__ seq_len, const float* __restrict__ d_c_final,
    const float* __restrict__ d_h_final, cbf16x4_t* __restrict__ dgates, float* __restrict__ dpeep_part,
    u64* __restrict__ xch, unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords, float clipz) {
  zero_next_area(znext, zwords);
  constexpr int PINM = PIN >= 0 ? PIN : ((HSU == 32 && H <= 320) ? (15 | 64) : 0);   // (H = 320 on eight waves: 1089 -> 1216 us with pins)
  constexpr int G = H / HSU;
  constexpr int TPC = HSU / 16;              // 16-unit output tiles per CU (4 / 2)
  constexpr int NWAVES = 2 * TPC;            // waves per workgroup
  constexpr int KC = 4 * HSU / 32;           // k-chunks of this CU's slice (8 / 4)
  constexpr int KSF = 4 * H / 32;            // k-chunks of the full packing
  constexpr int LDG = 4 * HSU + 8;
  constexpr size_t CL_U64 = XHDR + (size_t)2 * G * G * TPC * 64 * 2;   // u64 words per cluster
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // smem: [2][16][LDG] bf16 -- the own dG slice (MFMA A operand), double-buffered by iteration
  // parity so that ONE barrier per step orders writers and readers
  constexpr int DGB = 16 * LDG * 2;                        // bytes per buffer

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, rg = lane >> 4;
  const int hh = wave / TPC, wt = wave % TPC;
  const bool rev = (d == 1);
  const bf16_t* wp = whpb + (size_t)d * H * 4 * H;
  const int ul = wt * 16 + col;
  const unsigned jw = g * HSU + ul;
  const int rbase = rg * 4 + hh * 2;

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = seq_len[b0 + rbase + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  const float wci = peep ? peep[(d * 3 + 0) * H + jw] : 0.f;
  const float wcf = peep ? peep[(d * 3 + 1) * H + jw] : 0.f;
  const float wco = peep ? peep[(d * 3 + 2) * H + jw] : 0.f;

  // element offsets: os = frame s, oa = frame of step s for an active row (s or len-1-s)
  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? stride : 0u - stride;      // oa moves with decreasing s
  unsigned oa[2], os[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned base = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    os[r] = base + (unsigned)(tmax - 1) * stride;
    oa[r] = base + (unsigned)(rev ? len[r] - tmax : tmax - 1) * stride;
  }
  // inactive rows read nothing that is used: their fetches go to one parked position per row (frame tmax - 1, padding
  // for any row that is ever inactive) that stays in the L2 -- see the forward kernel
  unsigned opark[2] = {os[0], os[1]};
  // zero the gate gradients of the common padded tail [tmax, T)
  {
    const cbf16x4_t gzero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    for (int t = tmax; t < T_; ++t)
#pragma unroll
      for (int r = 0; r < 2; ++r) dgates[os[r] + (unsigned)(t - tmax + 1) * stride] = gzero;
  }

  float dhr[2], dcr[2], cc[2];
  float sums[7] = {0, 0, 0, 0, 0, 0, 0};     // dwci, dwcf, dwco, db_i, db_g, db_f, db_o
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t o = ((size_t)d * B_ + b0 + rbase + r) * H + jw;
    dhr[r] = d_h_final ? d_h_final[o] : 0.f;
    dcr[r] = d_c_final ? d_c_final[o] : 0.f;
    cc[r] = (tmax > 0 && tmax - 1 < len[r]) ? cs[oa[r]] : 0.f;
  }
  // W_h^T fragments: own tile (4g + wt) and this wave's foreign tiles f = wave, 8 + wave, ...
  // foreign index f in [0, 4(G-1)): destination CU = (f >> 2) skipping g, tile inside it = f & 3
  constexpr int NFT = TPC * (G - 1);                       // foreign tiles per CU (12 / 28; HSU = 32: 14 / 30)
  // G = 4: every wave also computes its own-unit tile (both row-half waves redundantly: no hand-over, one
  //        barrier per step) next to NF = 2 / 1 foreign tiles.
  // G = 8: that would be 5 tiles x 32 fragment registers per wave and spills (measured: 8 k of the 10 k
  //        cycles per step); instead the 4 own + 28 foreign tiles are dealt 4 per wave, the own tile is
  //        computed once by the hh = 0 wave, which hands rows 2,3 to its hh = 1 partner through LDS (a
  //        second barrier per step, cheap next to 32 MFMAs per wave).
  // HSU = 32 (one wave per SIMD, 512 registers): the redundant own tile again, NF = 4 (H = 256) / 8 (H = 512) foreign sets.
  constexpr bool OWN_ONCE = (G == 8 && HSU == 64);
  constexpr int NF = OWN_ONCE ? 4 : (NFT + NWAVES - 1) / NWAVES;   // fragment sets per wave (last one: see below)
  auto ftile = [&](int f) { const int q = f / TPC; return ((q + (q >= g ? 1 : 0)) * TPC) | (f % TPC); };
  const int nt_own = g * TPC + wt;
  // OWN_ONCE: slot NF-1 is the own tile on hh = 0 waves and foreign tile 24 + wt on hh = 1 waves
  const bool last_f = OWN_ONCE ? (hh == 1) : (wave < NFT - NWAVES * (NF - 1));
  int nt_f[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    if (OWN_ONCE) nt_f[i] = (i < NF - 1) ? ftile(wave + 8 * i) : (hh == 1 ? ftile(24 + wt) : nt_own);
    else nt_f[i] = ftile((i < NF - 1 || last_f) ? wave + NWAVES * i : wave);
  }
  bf16x8_t wo[OWN_ONCE ? 1 : KC], wf[NF][KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    if (!OWN_ONCE)
      wo[kc] = *reinterpret_cast<const bf16x8_t*>(wp + (((size_t)nt_own * KSF + g * KC + kc) * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < NF; ++i)
      wf[i][kc] = *reinterpret_cast<const bf16x8_t*>(wp + (((size_t)nt_f[i] * KSF + g * KC + kc) * 64 + lane) * 8);
  }
  // OWN_ONCE hand-over buffer behind the two dG images: [2 parity][4 tiles][64 lanes] x (row 2, row 3)
  float* ownx = reinterpret_cast<float*>(smem + 2 * DGB);

  u64* xhdr = xch + (size_t)cid.c * CL_U64;
  bool timed_out = false;
  const bool fast = same_xcd<G>(xhdr, g, timed_out) && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;  // drops to 0 after the first timeout
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY (ASR_LSTM_DFLAGS bit 6): a member goes missing
  f32x4_t* xs = reinterpret_cast<f32x4_t*>(xhdr + XHDR);   // [2][G dst][G src][4][64] x 16 B
  auto uslot = [&](int par, int dst, int src_, int tile) -> f32x4_t* {       // uniform part of a slot
    return xs + ((((size_t)par * G + dst) * G + src_) * TPC + tile) * 64;
  };
  const unsigned voff16 = (unsigned)lane * 16u;            // byte offset of this lane's 16-byte word group
  const unsigned pofs = (unsigned)lane * 2u + hh;          // u64 index of this lane's two rows in a slot
  // XP: 16-byte units [2 par][G dst][TPC][2 hh][NPAIR][64 lanes]; source g sits in pair k/2, half k%2 of destination
  // dst, k = g - (g > dst) its index among dst's peers
  constexpr int NPAIR = G / 2;                             // = ceil((G - 1) / 2)
  auto pslot = [&](int par, int dst, int tile, int h2, int pair) -> unsigned {   // uniform 16-byte index
    return (unsigned)((((par * G + dst) * TPC + tile) * 2 + h2) * NPAIR + pair) * 64u;
  };
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xs, 0, 0x7fffffff, 0x00020000);
  // own dG rows -> LDS (bytes); rows rbase, rbase+1 are in the same swizzle class (rbase is even)
  const unsigned lwr[2] = {((unsigned)(rbase * LDG + ul * 4) * 2u) ^ lds_swz(rbase),
                           ((unsigned)((rbase + 1) * LDG + ul * 4) * 2u) ^ lds_swz(rbase + 1)};
  const unsigned lrd = ((unsigned)(col * LDG + rg * 8) * 2u) ^ lds_swz(col);   // A fragment reads

  // saved activations of iteration s, fetched one iteration ahead
  // saved activations of iteration s, fetched one iteration ahead (two ahead was measured: 1.34 -> 1.37 ms)
  cbf16x4_t pg[2];
  float pcp[2], pdh[2];
  auto prefetch = [&](int s_) {                            // oa/os already hold iteration s_
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool act = s_ < len[r];
      const bool ldp = (s_ > 0) && (s_ - 1 < len[r]);
      const unsigned offl = act ? oa[r] : opark[r];
      const unsigned offn = ldp ? oa[r] + dstep : opark[r];
      pg[r] = gates[offl];
      pcp[r] = cs[offn];
      pdh[r] = dhout[offl];
    }
  };
  if (tmax > 0) prefetch(tmax - 1);
  __syncthreads();

  unsigned long long* dbg = g_cdbg;
  unsigned long long ph[4] = {0, 0, 0, 0};
  unsigned nspin = 0;
#define C8_T() (DBG ? (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime()) : 0ull)
#define C8_PIN(k) do { if constexpr (!DBG && ((PINM >> (k)) & 1)) __builtin_amdgcn_sched_barrier(0); } while (0)

  auto step = [&](int s, auto PAR) {
    constexpr int P = decltype(PAR)::value;                // parity of THIS iteration's publish
    const int it = tmax - 1 - s;                           // 0, 1, ...
    C8_PIN(0);
    const unsigned long long t0 = C8_T();
    // ---- 1. polls for the partials the peers published at the previous iteration (parity 1-P)
    u64 pv[XP ? 1 : G - 1];
    xw4_t pq[XP ? NPAIR : 1];
    if (it > 0) {
      if constexpr (XP) {
#pragma unroll
        for (int p = 0; p < NPAIR; ++p)                     // L1-bypassing (sc1) 16-byte loads, counted on vmcnt by the compiler
          pq[p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff16, pslot(1 - P, g, wt, hh, p) * 16u, 16);
      } else if constexpr ((ABL & 512) != 0) {
#pragma unroll
        for (int k = 0; k < G - 1; ++k) pv[k] = (u64)(unsigned)lane;
      } else {
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
          pv[k] = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, k + (k >= g ? 1 : 0), wt)), pofs));
      }
    }
    C8_PIN(4);                                             // behind the poll issue
    // ---- 2. everything that does not need dh
    bool act[2], ldp[2];
    float gi[2], gq[2], gf[2], go[2], cprev[2], a_o[2], b_c[2], c_g[2], c_i[2], c_f[2];
    unsigned off[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      act[r] = s < len[r];
      ldp[r] = (s > 0) && (s - 1 < len[r]);
      off[r] = act[r] ? oa[r] : os[r];
      gi[r] = (float)pg[r][0]; gq[r] = (float)pg[r][1]; gf[r] = (float)pg[r][2]; go[r] = (float)pg[r][3];
      cprev[r] = (act[r] && s > 0) ? pcp[r] : 0.f;
    }
    float tc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) tc[r] = cftanh(cc[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      a_o[r] = tc[r] * go[r] * (1.f - go[r]);
      b_c[r] = go[r] * (1.f - tc[r] * tc[r]);
      c_g[r] = gi[r] * (1.f - gq[r] * gq[r]);
      c_i[r] = gq[r] * gi[r] * (1.f - gi[r]);
      c_f[r] = cprev[r] * gf[r] * (1.f - gf[r]);
    }
    float pdh0 = pdh[0], pdh1 = pdh[1], pcp0 = pcp[0], pcp1 = pcp[1];
    const float cur0 = cc[0], cur1 = cc[1];
    if constexpr (HSU == 32) {
      // Everything that reads the values fetched one iteration ago is pinned HERE, ahead of the next fetch.  With 512
      // registers the allocator alternates the fetch's destination registers between the two unrolled iterations and
      // lets the last read of the old ones (cprev) sink below the new fetch; at the join behind the conditional fetch it
      // then has to wait with vmcnt(0) -- for the NEW fetch, i.e. for HBM, on every step (1.25 instead of 0.95 ms/launch).
      asm volatile("" : "+v"(pdh0), "+v"(pdh1), "+v"(pcp0), "+v"(pcp1), "+v"(cprev[0]), "+v"(cprev[1]));
      asm volatile("" : "+v"(c_g[0]), "+v"(c_g[1]), "+v"(c_i[0]), "+v"(c_i[1]), "+v"(c_f[0]), "+v"(c_f[1]));
      asm volatile("" : "+v"(a_o[0]), "+v"(a_o[1]), "+v"(b_c[0]), "+v"(b_c[1]), "+v"(gf[0]), "+v"(gf[1]));
    }
    // next iteration's inputs (independent of everything below)
#pragma unroll
    for (int r = 0; r < 2; ++r) { oa[r] += dstep; os[r] -= stride; }
    // The fetch of the next iteration's saved activations used to go out here, ahead of the poll loop; kflags bit 2
    // (ASR_LSTM_DFLAGS bit 7) puts it back for A-B runs.  It now goes out BEHIND the loop, see there.
    const bool early_fetch = (kflags & 4) != 0;
    if (s > 0 && early_fetch) prefetch(s - 1);
    C8_PIN(5);                                             // behind the dh-independent math
    // ---- 3. finish the polls: every word must carry the previous iteration's tag
    if (it > 0) {
      const unsigned want = (((unsigned)(it - 1) >> 1) + 1u) & 1u;
      const u64 wmask = 0x0000000100000001ull, wtag = want ? wmask : 0ull;
      unsigned spins = 0;
      float add0 = 0.f, add1 = 0.f;
      if constexpr (XP) {
#pragma unroll 1
        for (;;) {
          unsigned bad = 0;                                 // LSB of every live word must be the tag
#pragma unroll
          for (int p = 0; p < NPAIR; ++p) {
            bad |= (pq[p][0] ^ want) | (pq[p][1] ^ want);
            if (2 * p + 1 <= G - 2) bad |= (pq[p][2] ^ want) | (pq[p][3] ^ want);   // (G even: the last pair has one source)
          }
          if (__all((bad & 1u) == 0u)) break;
          if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
          asm volatile("" ::: "memory");                    // a fresh round of loads, not the old values
#pragma unroll
          for (int p = 0; p < NPAIR; ++p)
            pq[p] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff16, pslot(1 - P, g, wt, hh, p) * 16u, 16);
        }
#pragma unroll
        for (int k = 0; k < G - 1; ++k) {                   // fixed order: source index ascending, as XP = false
          add0 += __uint_as_float(pq[k >> 1][(k & 1) * 2 + 0] & ~1u);
          add1 += __uint_as_float(pq[k >> 1][(k & 1) * 2 + 1] & ~1u);
        }
      } else {
#pragma unroll 1
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int k = 0; k < G - 1; ++k) ok = ok && ((pv[k] & wmask) == wtag);
          if (__all(ok)) break;
          if constexpr ((ABL & (1 | 512)) != 0) break;       // whatever the first round carried
          if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
          for (int k = 0; k < G - 1; ++k)
            pv[k] = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, k + (k >= g ? 1 : 0), wt)), pofs));
        }
#pragma unroll
        for (int k = 0; k < G - 1; ++k) {                   // fixed order
          add0 += __uint_as_float((unsigned)pv[k] & ~1u);
          add1 += __uint_as_float((unsigned)(pv[k] >> 32) & ~1u);
        }
      }
      if (DBG) nspin += spins;
      dhr[0] += add0;
      dhr[1] += add1;
    }
    // The fetch of the next iteration's saved activations goes out BEHIND the poll loop.  Vector loads return in order: a
    // re-poll issued after a fetch that has to come from HBM cannot land before that fetch does, so with the fetch in
    // front of the loop the hand-off waited for DRAM on every step whose first poll came too early -- 1.22 ms per
    // launch against 0.94 ms behind it (11.93 -> 10.52 ms per headline step); the fetch still has the rest of the
    // step to arrive.  (Written as the second arm of a run-time switch on purpose: as an unconditional statement the
    // compiler schedules these independent loads differently and the launch takes 1.06 ms.)
    C8_PIN(6);                                             // behind the poll loop, in front of the fetch
    if constexpr ((ABL & 64) != 0) {
    } else if constexpr (HSU == 32) {
      // unconditional (the last iteration re-fetches its own rows: harmless) and pinned behind the loop by a compiler
      // barrier: behind a branch the wait-count pass waits with vmcnt(0) -- for THIS fetch -- at the join
      asm volatile("" ::: "memory");
      const bool more = s > 0;                             // block-uniform
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool actn = s - 1 < len[r];
        const bool ldpn = (s - 1 > 0) && (s - 2 < len[r]);
        const unsigned offl = more ? (actn ? oa[r] : opark[r]) : off[r];   // (oa / os already hold iteration s - 1)
        const unsigned offn = more ? (ldpn ? oa[r] + dstep : opark[r]) : off[r];
        pg[r] = gates[offl];
        pcp[r] = cs[offn];
        pdh[r] = dhout[offl];
      }
    } else {
      if (s > 0 && !early_fetch) prefetch(s - 1);
    }
    C8_PIN(1);
    const unsigned long long t1 = C8_T();
    // ---- 4. gate gradients of the own pairs
    const float pdhv[2] = {pdh0, pdh1}, pcpv[2] = {pcp0, pcp1}, curv[2] = {cur0, cur1};
    float zi[2], zg[2], zf[2], zo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float dh = pdhv[r] + dhr[r];
      if constexpr ((ABL & 8) != 0) {
        zi[r] = zg[r] = zf[r] = zo[r] = dh;
        dhr[r] = 0.f;
      } else {
      const float d_o = dh * a_o[r];
      const float dct = dcr[r] + dh * b_c[r] + d_o * wco;
      const float dc = (CLIPZ && fabsf(curv[r]) >= clipz) ? 0.f : dct;
      const float d_g = dc * c_g[r], d_i = dc * c_i[r], d_f = dc * c_f[r];
      dcr[r] = act[r] ? (dc * gf[r] + d_i * wci + d_f * wcf) : dcr[r];
      dhr[r] = act[r] ? 0.f : dhr[r];
      zi[r] = act[r] ? d_i : 0.f; zg[r] = act[r] ? d_g : 0.f;
      zf[r] = act[r] ? d_f : 0.f; zo[r] = act[r] ? d_o : 0.f;
      }
      cc[r] = ldp[r] ? pcpv[r] : 0.f;
      const cbf16x4_t pk = {(__bf16)zi[r], (__bf16)zg[r], (__bf16)zf[r], (__bf16)zo[r]};
      if constexpr (!(ABL & 256)) *reinterpret_cast<cbf16x4_t*>(smem + P * DGB + lwr[r]) = pk;
      if constexpr (!(ABL & 16)) dgates[off[r]] = pk;
      if constexpr ((ABL & (256 | 16)) == (256 | 16)) asm volatile("" ::"v"(pk));
    }
    C8_PIN(2);
    const unsigned long long t2 = C8_T();
    if constexpr (!(ABL & 32)) __syncthreads();
    C8_PIN(7);                                             // behind the step's barrier
    // peephole / bias gradient sums: off the critical path, behind the barrier
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      sums[0] += zi[r] * cprev[r]; sums[1] += zf[r] * cprev[r]; sums[2] += zo[r] * curv[r];
      sums[3] += zi[r]; sums[4] += zg[r]; sums[5] += zf[r]; sums[6] += zo[r];
    }
    C8_PIN(8);                                             // behind the gradient sums
    // ---- 5. partial dh_prev from the own dG slice; own tile stays, foreign tiles are published
    if (s > 0) {
      bf16x8_t afr[KC];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        if constexpr ((ABL & 2) != 0) afr[kc] = wf[0][kc];
        else afr[kc] = *reinterpret_cast<const bf16x8_t*>(smem + P * DGB + lrd + kc * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned tag = (((unsigned)it >> 1) + 1u) & 1u;
      auto tagged = [&](const f32x4_t& a) {
        f32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __uint_as_float((__float_as_uint(a[i]) & ~1u) | tag);
        return o;
      };
      // one foreign tile's partial (4 rows of one unit per lane) to its destination
      auto publish = [&](int nt, const f32x4_t& tv, auto FAST) {
        constexpr bool F = decltype(FAST)::value;
        const int dst = nt / TPC, tile = nt % TPC;
        if constexpr ((ABL & 128) != 0) {
          asm volatile("" ::"v"(tv));
        } else if constexpr (XP) {
          // Compiler-issued buffer stores, NOT inline asm with an SGPR base: with 16 slot bases per step (H = 512) the
          // bases are spilled to VGPR lanes and restored by v_readlane right in front of the store, and a VALU-written
          // SGPR needs 5 wait states before a VMEM instruction may use it as its address -- a hazard the compiler
          // handles for its own instructions but cannot see inside an asm block (measured: wild stores, memory fault).
          typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
          const int k = g - (g > dst ? 1 : 0);
          const unsigned so0 = pslot(P, dst, tile, 0, k >> 1) * 16u + (unsigned)(k & 1) * 8u;
          const unsigned so1 = pslot(P, dst, tile, 1, k >> 1) * 16u + (unsigned)(k & 1) * 8u;
          const u32x2_t lo2 = {__float_as_uint(tv[0]), __float_as_uint(tv[1])};
          const u32x2_t hi2 = {__float_as_uint(tv[2]), __float_as_uint(tv[3])};
          __builtin_amdgcn_raw_buffer_store_b64(lo2, xrs, voff16, so0, F ? 0 : 16);   // plain (stays in the L2) / sc1
          __builtin_amdgcn_raw_buffer_store_b64(hi2, xrs, voff16, so1, F ? 0 : 16);
        } else {
          xstore16(uslot(P, dst, g, tile), voff16, tv, F);
        }
      };
      // the tiles other CUs wait for go first; while they issue, this wave outranks its SIMD sibling's
      // own-tile MFMAs (nobody waits for those before the next step's gate math)
      __builtin_amdgcn_s_setprio(2);
      if constexpr (HSU == 32) {
        // One wave per SIMD: nothing fills the gaps of a tile-by-tile schedule (four DEPENDENT MFMAs into one
        // accumulator, drain, tag, a branch on the store flavour, store: ~170 cycles per tile, 9 tiles at H = 512), and
        // every destination waits for the LAST of its partials anyway.  So: all NF chains advance together (k-chunk
        // outer, tile inner: independent accumulators back to back), then the stores go out in one straight run per
        // flavour.  (The last tile of a wave that has none is computed and not stored.)
        f32x4_t af[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) af[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            if constexpr ((ABL & 4) != 0) af[i][0] = __uint_as_float(__float_as_uint(af[i][0]) ^ (__builtin_bit_cast(xw4_t, afr[kc])[i & 3] & 1u));
            else af[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], wf[i][kc], af[i], 0, 0, 0);
          }
        if (fast) {
#pragma unroll
          for (int i = 0; i < NF; ++i)
            if (i < NF - 1 || last_f) publish(nt_f[i], tagged(af[i]), std::integral_constant<bool, true>{});
        } else {
#pragma unroll
          for (int i = 0; i < NF; ++i)
            if (i < NF - 1 || last_f) publish(nt_f[i], tagged(af[i]), std::integral_constant<bool, false>{});
        }
      } else {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          if (OWN_ONCE ? (i < NF - 1 || last_f) : (i < NF - 1 || last_f)) {      // wave-uniform: a foreign tile
            f32x4_t af = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) af = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], wf[i][kc], af, 0, 0, 0);
            if (fast) publish(nt_f[i], tagged(af), std::integral_constant<bool, true>{});
            else publish(nt_f[i], tagged(af), std::integral_constant<bool, false>{});
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (OWN_ONCE) {
        if (hh == 0) {                                     // own tile, once: rows 0,1 stay, rows 2,3 -> partner
          f32x4_t ao = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) ao = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], wf[NF - 1][kc], ao, 0, 0, 0);
          dhr[0] += ao[0];
          dhr[1] += ao[1];
          float* o = ownx + ((P * 4 + wt) * 64 + lane) * 2;
          o[0] = ao[2];
          o[1] = ao[3];
        }
      } else {
        f32x4_t ao = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          if constexpr ((ABL & 4) != 0) ao[hh ? 2 : 0] = __uint_as_float(__float_as_uint(ao[hh ? 2 : 0]) ^ (__builtin_bit_cast(xw4_t, afr[kc])[0] & 1u));
          else ao = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kc], wo[kc], ao, 0, 0, 0);
        }
        dhr[0] += hh ? ao[2] : ao[0];
        dhr[1] += hh ? ao[3] : ao[1];
      }
    }
    if constexpr (OWN_ONCE) {
      __syncthreads();                                     // hand-over visible before the partner's next step
      if (s > 0 && hh == 1) {
        const float* o = ownx + ((P * 4 + wt) * 64 + lane) * 2;
        dhr[0] += o[0];
        dhr[1] += o[1];
      }
    }
    C8_PIN(3);
    const unsigned long long t3 = C8_T();
    if (DBG) { ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; }
  };
  (void)dbg;
  // parity of iteration `it` is it & 1; iterations come in pairs from it = 0
  int s = tmax - 1;
  for (; s >= 1; s -= 2) {
    step(s, std::integral_constant<int, 0>{});
    step(s - 1, std::integral_constant<int, 1>{});
  }
  if (s == 0) step(0, std::integral_constant<int, 0>{});
#undef C8_T
#undef C8_PIN

  if (DBG && dbg && lane == 0 && cid.tile == 0 && g < 4) {
    unsigned long long* o = dbg + 768 + ((size_t)(d * 4 + g) * 8 + wave) * 8;   // [768, 1280)
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = ph[k];
    o[4] = nspin;
    o[5] = tmax;
    o[6] = fast ? 1 : 0;
    o[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;
  }
  if (timed_out) atomicOr(err, 2u);
  if (dpeep_part) {
    // reduce over the 4 row groups of the wave, then over the two waves (hh) that share a unit
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      sums[k] += __shfl_xor(sums[k], 16, 64);
      sums[k] += __shfl_xor(sums[k], 32, 64);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [7][HSU]
    if (hh == 1 && rg == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) red[k * HSU + ul] = sums[k];
    }
    __syncthreads();
    if (hh == 0 && rg == 0) {
      float* p = dpeep_part + ((size_t)cid.tile * ndir + d) * 7 * H;
#pragma unroll
      for (int k = 0; k < 7; ++k) p[k * H + jw] = sums[k] + red[k * HSU + ul];
    }
  }
}

// ---------------------------------------------------------------- fp32 operands (exact fp32 MFMA path), H = 128
// BASELINE configs[0] (2 x 128 BLSTM-CTC, fp32) ran on the single-CU kernels of lstm.hip at 5.1 / 6.7 us per
// recurrence step: with fp32 operands the step is bound by the CU's fp32 matrix rate (v_mfma_f32_16x16x4_f32:
// 256 flop/clk/CU; h W_h of one step = 2.1 Mflop = 8.2 k cycles), not by streaming W_h.  The same cluster scheme
// halves that: G = H/64 = 2 CUs per (direction, tile), 64 units per CU, and the byte layouts of the bf16 H = 256
// form carry over unchanged (a row of h is 128 fp32 = 512 B, a fragment 16 B per lane, 8 fragments per tile, W_h
// slice in 64 VGPRs per wave).  What differs: four K = 4 MFMAs per fragment, one {tag, fp32} granule per (row,
// unit) pair in the all-gather (two per lane) instead of one {tag, 2 x bf16}, saved activations in fp32.
__device__ __forceinline__ void mma_f32x2(const f32x4_t& a, const f32x4_t& b0, const f32x4_t& b1, f32x4_t& c0,
                                          f32x4_t& c1) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], c1, 0, 0, 0);
  }
}
__device__ __forceinline__ f32x4_t mma_f32(const f32x4_t& a, const f32x4_t& b, f32x4_t c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
  return c;
}

// EARLY: as in the bf16 kernel, the fragments of the CU's OWN slice of h (half of K with two CUs) are multiplied for
// step s+1 right after they are written, before the wave starts polling for the peer's half.
// HSU = units per CU: 64 (eight waves) or 32 (four waves, one per SIMD: twice the CUs, each wave alone on its SIMD's fp32
// matrix pipe -- the step is bound by that pipe: 2 waves x 64 K=4 MFMAs x 32 cycles per SIMD in the 8-wave form).
template <int H, bool EARLY, int HSU = 64>
__global__ __launch_bounds__(HSU * 8, 1) void lstm_fwd_cluster8_f32_kernel(
    int T_, int B_, int ndir, const f32x4_t* __restrict__ xg, const float* __restrict__ whp,
    const float* __restrict__ peep, const int32_t* __restrict__ seq_len, float forget_bias,
    float cell_clip, f32x4_t* __restrict__ gates, float* __restrict__ hout, float* __restrict__ cs,
    float* __restrict__ c_final, float* __restrict__ h_final, u64* __restrict__ xch,
    unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords) {
  zero_next_area(znext, zwords);
  constexpr int G = H / HSU;
  constexpr int CTW = HSU * 8;
  constexpr int KS = H / 16;                               // fragments of k = 16 (four K = 4 MFMAs each)
  constexpr int LDH = H + 4;                               // floats: 33 x 16 B at H = 128, like the bf16 image
  constexpr int SLICE = 16 * HSU;                           // granules one CU publishes per step
  constexpr int KO = HSU / 16;                              // fragments of one CU's own slice
  // fragment index modulo KS (a mask where KS is a power of two; H = 320 has 20 fragments: x < 2 KS there)
  auto krotf = [](int x) -> int { return ((KS & (KS - 1)) == 0) ? (x & (KS - 1)) : (x >= KS ? x - KS : x); };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* hs = reinterpret_cast<float*>(smem);              // [2][16][LDH]

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool lo = col < 8;
  const bool rev = (d == 1);
  const float* wp = whp + (size_t)d * H * 4 * H;
  const int ul = wave * 8 + (col & 7);                     // unit inside this CU's slice
  const unsigned jw = g * HSU + ul;                         // global unit of this lane
  const int rbase = rg * 4 + (lo ? 0 : 2);                 // first of this lane's two batch rows

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = seq_len[b0 + rbase + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  float c[2] = {0.f, 0.f}, hr[2] = {0.f, 0.f};
  const float wci = peep ? peep[(d * 3 + 0) * H + jw] : 0.f;
  const float wcf = peep ? peep[(d * 3 + 1) * H + jw] : 0.f;
  const float wco = peep ? peep[(d * 3 + 2) * H + jw] : 0.f;

  for (int i = threadIdx.x; i < 2 * 16 * LDH; i += CTW) hs[i] = 0.f;
  // B fragments out of the standard forward packing (lstm.hip prep: tile = (unit/16)*4 + gate, fragment ks, lane
  // (n, rg) holds k = ks*16 + rg*4 + e): this lane's column of tile p is (gate p*2 + (col>>3), unit jw)
  f32x4_t wreg[2][KS];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int tile = (jw >> 4) * 4 + p * 2 + (col >> 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = EARLY ? krotf(ks + g * KO) : ks;            // EARLY: register fragment ks holds k-fragment kk
      wreg[p][ks] = *reinterpret_cast<const f32x4_t*>(wp + (((size_t)tile * KS + kk) * 64 + rg * 16 + (jw & 15)) * 4);
    }
  }

  u64* xhdr = xch + (size_t)cid.c * (XHDR + 2 * G * SLICE);
  u64* xbase = xhdr + XHDR;                                // [2 parity][G][8 waves][16 rows][8 units]
  bool timed_out = false;
  const bool colocated = same_xcd<G>(xhdr, g, timed_out);
  const bool fast = colocated && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY: a member goes missing
  __syncthreads();

  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? 0u - stride : stride;
  unsigned oa[2], os[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    os[r] = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    oa[r] = os[r] + (rev ? (unsigned)max(len[r] - 1, 0) * stride : 0u);
  }
  // granule i of a wave's 128: row i >> 3, unit i & 7.  A lane publishes its two (row, unit) pairs and polls
  // granules lane and 64 + lane of wave `wave` of every peer.
  const unsigned pofs = (unsigned)(wave * 128 + rbase * 8 + (col & 7));
  const unsigned lofs = (unsigned)(wave * 128 + lane);
  unsigned ldst[G - 1][2];                                 // byte offsets inside one h buffer
  auto slice = [&](int P, int gg) -> u64* { return xbase + ((size_t)P * G + gg) * SLICE; };
#pragma unroll
  for (int k = 0; k < G - 1; ++k) {
    const int gsrc = k + (k >= g ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 8 + (lane >> 3);
      ldst[k][j] = ((unsigned)(row * LDH + gsrc * HSU + wave * 8 + (lane & 7)) * 4u) ^ lds_swz(row);
    }
  }
  unsigned lown[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) lown[r] = ((unsigned)((rbase + r) * LDH + g * HSU + ul) * 4u) ^ lds_swz(rbase + r);
  const unsigned lrd = ((unsigned)(col * LDH + rg * 4) * 4u) ^ lds_swz(col);

  f32x4_t xq[2][2];                                        // x projection rows, requested two steps ahead
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    xq[0][r] = xg[(0 < len[r]) ? oa[r] : os[r]];
    xq[1][r] = (tmax > 1) ? xg[(1 < len[r]) ? oa[r] + dstep : os[r] + stride] : xq[0][r];
  }
  f32x4_t accn0 = {0.f, 0.f, 0.f, 0.f}, accn1 = {0.f, 0.f, 0.f, 0.f};   // EARLY: own-slice part of the next step

  auto step = [&](int s, auto PAR) {
    constexpr int P = decltype(PAR)::value;
    const char* hcur = smem + P * 16 * LDH * 4;
    char* hnxt = smem + (1 - P) * 16 * LDH * 4;
    const f32x4_t x0 = xq[P][0], x1 = xq[P][1];
    if (s + 2 < tmax) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        xq[P][r] = xg[(s + 2 < len[r]) ? oa[r] + 2u * dstep : os[r] + 2u * stride];
    }
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EARLY) { acc0 = accn0; acc1 = accn1; }    // own-slice fragments: done at the end of the last step
    {
      constexpr int K0 = EARLY ? KO : 0;
      if constexpr (KS <= 16) {
        f32x4_t afr[KS];
#pragma unroll
        for (int ks = K0; ks < KS; ++ks)
          afr[ks] = *reinterpret_cast<const f32x4_t*>(hcur + lrd + (EARLY ? krotf(ks + g * KO) : ks) * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = K0; ks < KS; ++ks) mma_f32x2(afr[ks], wreg[0][ks], wreg[1][ks], acc0, acc1);
      } else {                                             // H >= 320: the weights hold 160+ registers; A in runs of 8
#pragma unroll
        for (int kb = K0; kb < KS; kb += 8) {
          f32x4_t afr[8];
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
            if (kb + ks < KS)
              afr[ks] = *reinterpret_cast<const f32x4_t*>(hcur + lrd + (EARLY ? krotf(kb + ks + g * KO) : kb + ks) * 64);
#pragma unroll
          for (int ks = 0; ks < 8; ++ks)
            if (kb + ks < KS) mma_f32x2(afr[ks], wreg[0][kb + ks], wreg[1][kb + ks], acc0, acc1);
        }
      }
    }
    float pi[2], pq[2], pf[2], po[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      pi[r] = dpp_ror8_into<0xC>(acc0[r], acc0[2 + r]);
      pq[r] = dpp_ror8_into<0x3>(acc0[2 + r], acc0[r]);
      pf[r] = dpp_ror8_into<0xC>(acc1[r], acc1[2 + r]);
      po[r] = dpp_ror8_into<0x3>(acc1[2 + r], acc1[r]);
    }
    const unsigned epoch = (unsigned)s + 1u;
    bool act[2];
    float ig[2], gg[2], fg[2], og[2], cn[2], hn[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) act[r] = s < len[r];
    const f32x4_t xr[2] = {x0, x1};
#pragma unroll
    for (int r = 0; r < 2; ++r) ig[r] = cfsig(pi[r] + xr[r][0] + wci * c[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) gg[r] = cftanh(pq[r] + xr[r][1]);
#pragma unroll
    for (int r = 0; r < 2; ++r) fg[r] = cfsig(pf[r] + xr[r][2] + forget_bias + wcf * c[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      cn[r] = gg[r] * ig[r] + c[r] * fg[r];
      if (cell_clip > 0.f) cn[r] = fminf(fmaxf(cn[r], -cell_clip), cell_clip);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) og[r] = cfsig(po[r] + xr[r][3] + wco * cn[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) hn[r] = cftanh(cn[r]) * og[r];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      c[r] = act[r] ? cn[r] : c[r];
      hr[r] = act[r] ? hn[r] : hr[r];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      gpublish(uoff(slice(P, g), pofs + 8u * r), epoch, __float_as_uint(hr[r]), fast);
      *reinterpret_cast<float*>(hnxt + lown[r]) = hr[r];
    }
    unsigned offs[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      offs[r] = act[r] ? oa[r] : os[r];
      oa[r] += dstep;
      os[r] += stride;
    }
    if constexpr (EARLY) {
      if (s + 1 < tmax) {                                  // block-uniform
        __syncthreads();                                   // the CU's own slice of h(s) is complete in hnxt
        f32x4_t ao[KO];
#pragma unroll
        for (int k = 0; k < KO; ++k) ao[k] = *reinterpret_cast<const f32x4_t*>(hnxt + lrd + krotf(k + g * KO) * 64);
        accn0 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        accn1 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KO; ++k) mma_f32x2(ao[k], wreg[0][k], wreg[1][k], accn0, accn1);
      }
    }
    {
      u64 v[G - 1][2];
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(slice(P, k + (k >= g ? 1 : 0)), lofs + 64u * j));
      unsigned spins = 0;
#pragma unroll 1
      for (;;) {                                           // wave-uniform loop: no exec masking
        bool ok = true;
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
#pragma unroll
          for (int j = 0; j < 2; ++j) ok = ok && ((unsigned)(v[k][j] >> 32) == epoch);
        if (__all(ok)) break;
        if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
#pragma unroll
          for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(slice(P, k + (k >= g ? 1 : 0)), lofs + 64u * j));
      }
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<unsigned*>(hnxt + ldst[k][j]) = (unsigned)v[k][j];
    }
    // saved activations behind the poll loop (loads and stores share the wave's in-order counter); rows past their
    // length write frame s of the padding (hout: zeros; gates / cs: never read there)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      hout[offs[r]] = act[r] ? hr[r] : 0.f;
      gates[offs[r]] = (f32x4_t){ig[r], gg[r], fg[r], og[r]};
      cs[offs[r]] = cn[r];
    }
    __syncthreads();
  };
  int s = 0;
  for (; s + 1 < tmax; s += 2) {
    step(s, std::integral_constant<int, 0>{});
    step(s + 1, std::integral_constant<int, 1>{});
  }
  if (s < tmax) step(s, std::integral_constant<int, 0>{});

  if (timed_out) atomicOr(err, 1u);
  for (int t = tmax; t < T_; ++t)
#pragma unroll
    for (int r = 0; r < 2; ++r) { hout[os[r]] = 0.f; os[r] += stride; }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t o = ((size_t)d * B_ + b0 + rbase + r) * H + jw;
    if (c_final) c_final[o] = c[r];
    if (h_final) h_final[o] = hr[r];
  }
}

// ---------------------------------------------------------------- fp32 operands on the bf16 matrix pipe (round 5)
// The fp32 cluster kernels above are bound by v_mfma_f32_16x16x4_f32: 32 cycles per SIMD for K = 4, i.e. 256 cycles per
// K = 32 against 17 for one v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md) -- 2 048 of the ~4 000 cycles of a step at
// H = 128.  An fp32 value IS the sum of three bf16 values (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8
// significand bits; both subtractions are exact in fp32), so x.w = sum of the nine term products, of which the six with
// weight >= 2^-16 are kept: lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi -- every product of two bf16 values is exact
// in the MFMA's fp32 accumulator; the dropped terms (mid.lo, lo.mid, lo.lo) are below 2^-24 of the result.  6 bf16 MFMAs x 17
// cycles = 102 per K = 32: the matrix phase of a step drops from 2 048 to ~820 cycles.  W_h is split once per launch into
// registers; h (forward) / the gate gradients (BPTT) are split by the lane that writes them into three bf16 LDS images.
// Same exchange, same saved activations, same lane <-> (row, unit) mapping as the fp32 kernels above; results differ from
// theirs by accumulation rounding only (fp32 parity bounds unchanged, tests/test_gpu_ops.py).  ASR_LSTM_F32_SPLIT=0 keeps
// the exact-fp32 MFMA kernels.
__device__ __forceinline__ void split3(float x, unsigned short (&t)[3]) {
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;
  const __bf16 m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  t[0] = __builtin_bit_cast(unsigned short, h);
  t[1] = __builtin_bit_cast(unsigned short, m);
  t[2] = __builtin_bit_cast(unsigned short, (__bf16)r2);
}
// one K = 32 chunk of a 16 x 16 tile: a[term], b[term] bf16 fragments; smallest terms first
__device__ __forceinline__ f32x4_t mma_s3(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x4_t c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}
// B fragment of chunk c (k = 32 c + 8 rg + j, j < 8) of one 16-column tile out of an fp32 fragment packing
// [frag16][64 lanes][4] (lane (n, rg') holds k = 16 frag16 + 4 rg' + e): two 16-byte loads, split into three bf16 terms
__device__ __forceinline__ void load_split_frag(const float* tile_base, int c, int rg, int n, bf16x8_t (&out)[3]) {
  const float* p = tile_base + (((size_t)(2 * c + (rg >> 1)) * 64 + ((rg & 1) * 2) * 16 + n) * 4);
  const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(p);
  const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(p + 16 * 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned short t[3];
    split3(j < 4 ? v0[j] : v1[j - 4], t);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k][j] = (short)t[k];
  }
}

template <int H, bool EARLY, int HSU = 32>
__global__ __launch_bounds__(HSU * 8, 1) void lstm_fwd_cluster_f32s_kernel(
    int T_, int B_, int ndir, const f32x4_t* __restrict__ xg, const float* __restrict__ whp,
    const float* __restrict__ peep, const int32_t* __restrict__ seq_len, float forget_bias,
    float cell_clip, f32x4_t* __restrict__ gates, float* __restrict__ hout, float* __restrict__ cs,
    float* __restrict__ c_final, float* __restrict__ h_final, u64* __restrict__ xch,
    unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords) {
  zero_next_area(znext, zwords);
  constexpr int G = H / HSU;
  constexpr int CTW = HSU * 8;
  static_assert(HSU == 32 && (H & (H - 1)) == 0, "four waves per CU, power-of-two width");
  constexpr int KS = H / 32;                               // chunks of k = 32 (six bf16 MFMAs each per tile)
  constexpr int KS16 = H / 16;                             // fragments of the fp32 weight packing
  constexpr int LDH = H + 8;                               // bf16 elements per image row: 17 / 33 x 16 B
  constexpr int PLB = 16 * LDH * 2;                        // bytes of one term image
  constexpr int SLICE = 16 * HSU;                           // granules one CU publishes per step
  constexpr int KO = HSU / 32;                              // chunks of one CU's own slice
  auto krotf = [](int x) -> int { return x & (KS - 1); };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* hs = reinterpret_cast<unsigned short*>(smem);   // [2 parity][3 terms][16][LDH]

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool lo = col < 8;
  const bool rev = (d == 1);
  const float* wp = whp + (size_t)d * H * 4 * H;
  const int ul = wave * 8 + (col & 7);                     // unit inside this CU's slice
  const unsigned jw = g * HSU + ul;                         // global unit of this lane
  const int rbase = rg * 4 + (lo ? 0 : 2);                 // first of this lane's two batch rows

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = seq_len[b0 + rbase + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  float c[2] = {0.f, 0.f}, hr[2] = {0.f, 0.f};
  const float wci = peep ? peep[(d * 3 + 0) * H + jw] : 0.f;
  const float wcf = peep ? peep[(d * 3 + 1) * H + jw] : 0.f;
  const float wco = peep ? peep[(d * 3 + 2) * H + jw] : 0.f;

  for (int i = threadIdx.x; i < 2 * 3 * 16 * LDH; i += CTW) hs[i] = 0;
  // B fragments out of the standard fp32 forward packing (lstm.hip prep: tile = (unit/16)*4 + gate, fragment of k = 16, lane
  // (n, rg) holds k = 16 frag + 4 rg + e), re-cut into k = 32 chunks and split: this lane's column of tile p is
  // (gate p*2 + (col>>3), unit jw)
  bf16x8_t wreg[2][KS][3];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int tile = (jw >> 4) * 4 + p * 2 + (col >> 3);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = EARLY ? krotf(ks + g * KO) : ks;            // EARLY: register chunk ks holds k-chunk kk
      load_split_frag(wp + (size_t)tile * KS16 * 256, kk, rg, (int)(jw & 15), wreg[p][ks]);
    }
  }

  u64* xhdr = xch + (size_t)cid.c * (XHDR + 2 * G * SLICE);
  u64* xbase = xhdr + XHDR;                                // [2 parity][G][8 waves][16 rows][8 units]
  bool timed_out = false;
  const bool colocated = same_xcd<G>(xhdr, g, timed_out);
  const bool fast = colocated && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY: a member goes missing
  __syncthreads();

  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? 0u - stride : stride;
  unsigned oa[2], os[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    os[r] = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    oa[r] = os[r] + (rev ? (unsigned)max(len[r] - 1, 0) * stride : 0u);
  }
  // granule i of a wave's 128: row i >> 3, unit i & 7.  A lane publishes its two (row, unit) pairs and polls
  // granules lane and 64 + lane of wave `wave` of every peer.
  const unsigned pofs = (unsigned)(wave * 128 + rbase * 8 + (col & 7));
  const unsigned lofs = (unsigned)(wave * 128 + lane);
  unsigned ldst[G - 1][2];                                 // byte offsets inside one h buffer
  auto slice = [&](int P, int gg) -> u64* { return xbase + ((size_t)P * G + gg) * SLICE; };
#pragma unroll
  for (int k = 0; k < G - 1; ++k) {
    const int gsrc = k + (k >= g ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 8 + (lane >> 3);
      ldst[k][j] = ((unsigned)(row * LDH + gsrc * HSU + wave * 8 + (lane & 7)) * 2u) ^ lds_swz(row);
    }
  }
  unsigned lown[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) lown[r] = ((unsigned)((rbase + r) * LDH + g * HSU + ul) * 2u) ^ lds_swz(rbase + r);
  const unsigned lrd = ((unsigned)(col * LDH + rg * 8) * 2u) ^ lds_swz(col);
  // The all-gather carries the TERMS: granule = {hi | mid << 16, lo | tag16 << 16} (one 8-byte store / load per (row, unit)
  // pair as in the fp32 kernels, tag = (step + 1) & 0xffff: T_ < 65536, checked by the launcher) -- a value is split once, by
  // the lane that produced it, and a consumer stages what it polled with three 2-byte LDS stores and no arithmetic.
  auto put3w = [&](char* img, unsigned off, unsigned w0, unsigned w1) {
    *reinterpret_cast<unsigned short*>(img + off) = (unsigned short)w0;
    *reinterpret_cast<unsigned short*>(img + PLB + off) = (unsigned short)(w0 >> 16);
    *reinterpret_cast<unsigned short*>(img + 2 * PLB + off) = (unsigned short)w1;
  };

  // rows past their length: saved activations never read, x-projection rows not used -> ONE parked position per row (its
  // frame tmax - 1) that stays in the L2 instead of streaming the padded part of the batch through HBM (see the bf16 kernel)
  unsigned opark[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) opark[r] = os[r] + (unsigned)max(tmax - 1, 0) * stride;
  f32x4_t xq[2][2];                                        // x projection rows, requested two steps ahead
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    xq[0][r] = xg[(0 < len[r]) ? oa[r] : opark[r]];
    xq[1][r] = (tmax > 1) ? xg[(1 < len[r]) ? oa[r] + dstep : opark[r]] : xq[0][r];
  }
  f32x4_t accn0 = {0.f, 0.f, 0.f, 0.f}, accn1 = {0.f, 0.f, 0.f, 0.f};   // EARLY: own-slice part of the next step

  auto step = [&](int s, auto PAR) {
    constexpr int P = decltype(PAR)::value;
    const char* hcur = smem + P * 3 * PLB;
    char* hnxt = smem + (1 - P) * 3 * PLB;
    const f32x4_t x0 = xq[P][0], x1 = xq[P][1];
    if (s + 2 < tmax) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        xq[P][r] = xg[(s + 2 < len[r]) ? oa[r] + 2u * dstep : opark[r]];
    }
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EARLY) { acc0 = accn0; acc1 = accn1; }    // own-slice fragments: done at the end of the last step
    {
      constexpr int K0 = EARLY ? KO : 0;
      bf16x8_t afr[KS][3];
#pragma unroll
      for (int ks = K0; ks < KS; ++ks)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          afr[ks][k] = *reinterpret_cast<const bf16x8_t*>(hcur + k * PLB + lrd + (EARLY ? krotf(ks + g * KO) : ks) * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = K0; ks < KS; ++ks) {
        acc0 = mma_s3(afr[ks], wreg[0][ks], acc0);
        acc1 = mma_s3(afr[ks], wreg[1][ks], acc1);
      }
    }
    float pi[2], pq[2], pf[2], po[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      pi[r] = dpp_ror8_into<0xC>(acc0[r], acc0[2 + r]);
      pq[r] = dpp_ror8_into<0x3>(acc0[2 + r], acc0[r]);
      pf[r] = dpp_ror8_into<0xC>(acc1[r], acc1[2 + r]);
      po[r] = dpp_ror8_into<0x3>(acc1[2 + r], acc1[r]);
    }
    const unsigned epoch = (unsigned)s + 1u;
    bool act[2];
    float ig[2], gg[2], fg[2], og[2], cn[2], hn[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) act[r] = s < len[r];
    const f32x4_t xr[2] = {x0, x1};
#pragma unroll
    for (int r = 0; r < 2; ++r) ig[r] = cfsig(pi[r] + xr[r][0] + wci * c[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) gg[r] = cftanh(pq[r] + xr[r][1]);
#pragma unroll
    for (int r = 0; r < 2; ++r) fg[r] = cfsig(pf[r] + xr[r][2] + forget_bias + wcf * c[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      cn[r] = gg[r] * ig[r] + c[r] * fg[r];
      if (cell_clip > 0.f) cn[r] = fminf(fmaxf(cn[r], -cell_clip), cell_clip);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) og[r] = cfsig(po[r] + xr[r][3] + wco * cn[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) hn[r] = cftanh(cn[r]) * og[r];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      c[r] = act[r] ? cn[r] : c[r];
      hr[r] = act[r] ? hn[r] : hr[r];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned short t3[3];
      split3(hr[r], t3);
      const unsigned w0 = (unsigned)t3[0] | ((unsigned)t3[1] << 16);
      gpublish(uoff(slice(P, g), pofs + 8u * r), ((unsigned)t3[2]) | (epoch << 16), w0, fast);
      put3w(hnxt, lown[r], w0, (unsigned)t3[2]);
    }
    unsigned offs[2], offgs[2];                            // hout position / saved-activation position
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      offs[r] = act[r] ? oa[r] : os[r];
      offgs[r] = act[r] ? oa[r] : opark[r];
      oa[r] += dstep;
      os[r] += stride;
    }
    if constexpr (EARLY) {
      if (s + 1 < tmax) {                                  // block-uniform
        __syncthreads();                                   // the CU's own slice of h(s) is complete in hnxt
        bf16x8_t ao[KO][3];
#pragma unroll
        for (int k = 0; k < KO; ++k)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            ao[k][q] = *reinterpret_cast<const bf16x8_t*>(hnxt + q * PLB + lrd + krotf(k + g * KO) * 64);
        accn0 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        accn1 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KO; ++k) {
          accn0 = mma_s3(ao[k], wreg[0][k], accn0);
          accn1 = mma_s3(ao[k], wreg[1][k], accn1);
        }
      }
    }
    {
      u64 v[G - 1][2];
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(slice(P, k + (k >= g ? 1 : 0)), lofs + 64u * j));
      unsigned spins = 0;
#pragma unroll 1
      for (;;) {                                           // wave-uniform loop: no exec masking
        bool ok = true;
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
#pragma unroll
          for (int j = 0; j < 2; ++j) ok = ok && ((unsigned)(v[k][j] >> 48) == (epoch & 0xffffu));
        if (__all(ok)) break;
        if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
#pragma unroll
          for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(slice(P, k + (k >= g ? 1 : 0)), lofs + 64u * j));
      }
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) put3w(hnxt, ldst[k][j], (unsigned)v[k][j], (unsigned)(v[k][j] >> 32));
    }
    // saved activations behind the poll loop (loads and stores share the wave's in-order counter); rows past their
    // length write frame s of the padding (hout: zeros; gates / cs: never read there)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      hout[offs[r]] = act[r] ? hr[r] : 0.f;
      gates[offgs[r]] = (f32x4_t){ig[r], gg[r], fg[r], og[r]};
      cs[offgs[r]] = cn[r];
    }
    __syncthreads();
  };
  int s = 0;
  for (; s + 1 < tmax; s += 2) {
    step(s, std::integral_constant<int, 0>{});
    step(s + 1, std::integral_constant<int, 1>{});
  }
  if (s < tmax) step(s, std::integral_constant<int, 0>{});

  if (timed_out) atomicOr(err, 1u);
  for (int t = tmax; t < T_; ++t)
#pragma unroll
    for (int r = 0; r < 2; ++r) { hout[os[r]] = 0.f; os[r] += stride; }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t o = ((size_t)d * B_ + b0 + rbase + r) * H + jw;
    if (c_final) c_final[o] = c[r];
    if (h_final) h_final[o] = hr[r];
  }
}

// ---------------------------------------------------------------- GRU forward on a cluster (round 6)
// tf.contrib.rnn.GRUCell (models/encoders/core/gru.py:58-76):  [r | u] = sigmoid(xg + h W_gh),  c = tanh(xc + (r * h) W_ch),
// h' = u h + (1 - u) c -- TWO dependent products per step, so two all-gathers: r * h between them, h' behind the second.
// The single-CU persistent kernel (gru.hip) streams both recurrent blocks from L2 every step and multiplies on the fp32 MFMA:
// 11.7 us of matrix work + the weight trips per step at H = 256.  Here the cluster machinery of lstm_fwd_cluster_f32s_kernel:
// H / 32 CUs per (direction, 16-utterance tile), four waves x 8 units, the CU's slices of W_gh ([H] x [r | u of its units])
// and W_ch as three-bf16-term fragments in registers for the whole launch, h and r * h as three-term LDS images, both
// exchanges in 8-byte self-tagged granules {hi | mid << 16, lo | tag << 16}.  Phase 1 is one 16-column tile per wave
// ([r x 8 | u x 8], halves swapped over DPP as in the LSTM kernels), phase 2 one tile whose upper eight columns repeat the
// lower ones (a 16-column MFMA for 8 units: the lanes col >= 8 read rows 2, 3 of the same unit from their own copy).
// Saved activations r, u, c, r * h at the frame a row worked on; rows past their length store nothing but hout's zero
// and keep h (their x loads are parked on a valid frame).  Same fp32-level arithmetic as the f32s LSTM kernels (six exact bf16 products per k = 32 chunk).
template <int H>
__global__ __launch_bounds__(256, 1) void gru_fwd_cluster_kernel(
    int T_, int B_, int ndir, const float* __restrict__ xg, const float* __restrict__ xc, const float* __restrict__ wgh,
    const float* __restrict__ wch, const int32_t* __restrict__ seq_len, float* __restrict__ r_out, float* __restrict__ u_out,
    float* __restrict__ c_out, float* __restrict__ rh_out, float* __restrict__ hout, float* __restrict__ h_final,
    u64* __restrict__ xch, unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords) {
  zero_next_area(znext, zwords);
  constexpr int HSU = 32, G = H / HSU;
  constexpr int KS = H / 32;
  constexpr int LDH = H + 8;                               // bf16 elements per image row
  constexpr int PLB = 16 * LDH * 2;                        // bytes of one term image
  constexpr int SLICE = 16 * HSU;                           // granules one CU publishes per exchange and step
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [h parity 0][h parity 1][r * h], three term images each

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool lo = col < 8;
  const bool rev = (d == 1);
  const int ul = wave * 8 + (col & 7);                     // unit inside this CU's slice
  const unsigned jw = g * HSU + ul;                         // global unit of this lane
  const int rbase = rg * 4 + (lo ? 0 : 2);                 // first of this lane's two batch rows

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = min(max(seq_len[b0 + rbase + r], 0), T_);
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  for (int i = threadIdx.x; i < 3 * 3 * 16 * LDH; i += 256) reinterpret_cast<unsigned short*>(smem)[i] = 0;
  // B fragments straight from the row-major recurrent blocks: lane (column col, k-group rg) of chunk ks holds
  // W[32 ks + 8 rg + j][column], j < 8, split into three bf16 terms.  Phase 1: column = gate (col >> 3) of unit jw;
  // phase 2: unit jw (both halves of the tile).
  bf16x8_t w1[KS][3], w2[KS][3];
  {
    const float* wg = wgh + (size_t)d * H * 2 * H + (size_t)(col >> 3) * H + jw;
    const float* wc = wch + (size_t)d * H * H + jw;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * ks + 8 * rg + j;
        unsigned short t1[3], t2[3];
        split3(wg[(size_t)k * 2 * H], t1);
        split3(wc[(size_t)k * H], t2);
#pragma unroll
        for (int q = 0; q < 3; ++q) { w1[ks][q][j] = (short)t1[q]; w2[ks][q][j] = (short)t2[q]; }
      }
  }

  u64* xhdr = xch + (size_t)cid.c * (XHDR + 4 * G * SLICE);
  u64* xa = xhdr + XHDR;                                   // r * h: [2 parity][G][SLICE]
  u64* xb = xa + 2 * G * SLICE;                            // h':    [2 parity][G][SLICE]
  bool timed_out = false;
  const bool colocated = same_xcd<G>(xhdr, g, timed_out);
  const bool fast = colocated && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY: a member goes missing
  __syncthreads();

  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? 0u - stride : stride;
  unsigned oa[2], os[2], os0[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    os[r] = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    os0[r] = os[r];
    oa[r] = os[r] + (rev ? (unsigned)max(len[r] - 1, 0) * stride : 0u);
  }
  const unsigned pofs = (unsigned)(wave * 128 + rbase * 8 + (col & 7));
  const unsigned lofs = (unsigned)(wave * 128 + lane);
  unsigned ldst[G - 1][2];
#pragma unroll
  for (int k = 0; k < G - 1; ++k) {
    const int gsrc = k + (k >= g ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 8 + (lane >> 3);
      ldst[k][j] = ((unsigned)(row * LDH + gsrc * HSU + wave * 8 + (lane & 7)) * 2u) ^ lds_swz(row);
    }
  }
  unsigned lown[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) lown[r] = ((unsigned)((rbase + r) * LDH + g * HSU + ul) * 2u) ^ lds_swz(rbase + r);
  const unsigned lrd = ((unsigned)(col * LDH + rg * 8) * 2u) ^ lds_swz(col);
  auto put3w = [&](char* img, unsigned off, unsigned w0, unsigned w1v) {
    *reinterpret_cast<unsigned short*>(img + off) = (unsigned short)w0;
    *reinterpret_cast<unsigned short*>(img + PLB + off) = (unsigned short)(w0 >> 16);
    *reinterpret_cast<unsigned short*>(img + 2 * PLB + off) = (unsigned short)w1v;
  };
  auto product = [&](const char* img, const bf16x8_t (&w)[KS][3]) -> f32x4_t {
    bf16x8_t afr[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int q = 0; q < 3; ++q) afr[ks][q] = *reinterpret_cast<const bf16x8_t*>(img + q * PLB + lrd + ks * 64);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = mma_s3(afr[ks], w[ks], acc);
    return acc;
  };
  // publish this lane's two (row, unit) values into slice g of `area`, stage them into `img`; then collect the peers' slices
  auto exchange = [&](u64* area, int P, unsigned epoch, const float (&val)[2], char* img) {
    u64* mine = area + ((size_t)P * G + g) * SLICE;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned short t3[3];
      split3(val[r], t3);
      const unsigned w0 = (unsigned)t3[0] | ((unsigned)t3[1] << 16);
      gpublish(uoff(mine, pofs + 8u * r), ((unsigned)t3[2]) | (epoch << 16), w0, fast);
      put3w(img, lown[r], w0, (unsigned)t3[2]);
    }
    u64 v[G - 1][2];
#pragma unroll
    for (int k = 0; k < G - 1; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(area + ((size_t)P * G + k + (k >= g ? 1 : 0)) * SLICE, lofs + 64u * j));
    unsigned spins = 0;
#pragma unroll 1
    for (;;) {                                             // wave-uniform loop: no exec masking
      bool ok = true;
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) ok = ok && ((unsigned)(v[k][j] >> 48) == (epoch & 0xffffu));
      if (__all(ok)) break;
      if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(area + ((size_t)P * G + k + (k >= g ? 1 : 0)) * SLICE, lofs + 64u * j));
    }
#pragma unroll
    for (int k = 0; k < G - 1; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) put3w(img, ldst[k][j], (unsigned)v[k][j], (unsigned)(v[k][j] >> 32));
  };

  float hr[2] = {0.f, 0.f};
  char* rhimg = smem + 2 * 3 * PLB;
  // x-projections of a step, requested one step ahead (a row past its length: the parked frame, unused)
  float nxr[2], nxu[2], nxc[2];
  auto xfetch = [&](int s) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool a = s < len[r];
      const int fr = a ? (rev ? len[r] - 1 - s : s) : max(tmax - 1, 0);
      const unsigned o = os0[r] + (unsigned)fr * stride;
      const unsigned og = (o - jw) * 2u + jw;              // the same (frame, row, direction) in the [.., 2H] gate tensor
      nxr[r] = xg[og];
      nxu[r] = xg[og + H];
      nxc[r] = xc[o];
    }
  };
  xfetch(0);
  for (int s = 0; s < tmax; ++s) {
    const int P = s & 1;
    const char* hcur = smem + P * 3 * PLB;
    char* hnxt = smem + (1 - P) * 3 * PLB;
    const unsigned epoch = (unsigned)s + 1u;
    bool act[2];
    float xr[2], xu[2], xcv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      act[r] = s < len[r];
      xr[r] = nxr[r]; xu[r] = nxu[r]; xcv[r] = nxc[r];
    }
    if (s + 1 < tmax) xfetch(s + 1);                       // block-uniform
    // ---- phase 1: [r | u] = sigmoid(xg + h W_gh)
    const f32x4_t a1 = product(hcur, w1);
    float rv[2], uv[2], rhv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float pr = dpp_ror8_into<0xC>(a1[r], a1[2 + r]);
      const float pu = dpp_ror8_into<0x3>(a1[2 + r], a1[r]);
      rv[r] = cfsig(pr + xr[r]);
      uv[r] = cfsig(pu + xu[r]);
      rhv[r] = rv[r] * hr[r];
    }
    exchange(xa, P, epoch, rhv, rhimg);
    __syncthreads();                                       // r * h complete in LDS
    // ---- phase 2: c = tanh(xc + (r h) W_ch), h' = u h + (1 - u) c
    const f32x4_t a2 = product(rhimg, w2);
    float cv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float pc = lo ? a2[r] : a2[2 + r];
      cv[r] = cftanh(pc + xcv[r]);
      const float hn = uv[r] * hr[r] + (1.f - uv[r]) * cv[r];
      hr[r] = act[r] ? hn : hr[r];
    }
    exchange(xb, P, epoch, hr, hnxt);
    // saved activations behind the poll loop; a row past its length writes a zero into frame s of hout's padding and
    // nothing else (r * h is contracted over ALL T * B rows by the weight-gradient GEMM: the caller's zeros must stay)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      hout[act[r] ? oa[r] : os[r]] = act[r] ? hr[r] : 0.f;
      if (act[r]) {
        const unsigned o = oa[r];
        r_out[o] = rv[r];
        u_out[o] = uv[r];
        c_out[o] = cv[r];
        rh_out[o] = rhv[r];
      }
      oa[r] += dstep;
      os[r] += stride;
    }
    __syncthreads();                                       // h' complete in LDS; r * h free again
  }
  if (timed_out) atomicOr(err, 1u);
  for (int t = tmax; t < T_; ++t)
#pragma unroll
    for (int r = 0; r < 2; ++r) { hout[os[r]] = 0.f; os[r] += stride; }
#pragma unroll
  for (int r = 0; r < 2; ++r) h_final[((size_t)d * B_ + b0 + rbase + r) * H + jw] = hr[r];
}

// GRU backward through time on the same clusters, as two ALL-GATHERS per step (not the LSTM BPTT's reduce-scatter of fp32
// partials): a CU owns the dh of its 32 units, so it needs every unit's pre-activation gradients and the rows of W^T that
// lead to its units --
//     dh = dout[frame] + dh_rec;  du_pre = dh (hp - c) u (1 - u);  dc_pre = dh (1 - u)(1 - c^2);  acc = dh u      (own units)
//     gather dc_pre, du_pre;      d(rh) = dc_pre W_c^T [own units];  dr_pre = d(rh) hp r (1 - r);  acc += d(rh) r
//     gather dr_pre;              dh_rec' = acc + [dr_pre | du_pre] W_g^T [own units]
// with W_c^T / W_g^T columns of the CU's units as three-term fragments in registers (three sets), the gathered gradients
// as three-term LDS images (dc_pre and du_pre double-buffered by step parity: two barriers per step), saved activations and
// dout requested one step ahead.  dgate / dcand are written for active rows only (the caller zeroes them).
template <int H>
__global__ __launch_bounds__(256, 1) void gru_bwd_cluster_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dout, const float* __restrict__ d_h_final,
    const float* __restrict__ hout, const float* __restrict__ r_in, const float* __restrict__ u_in,
    const float* __restrict__ c_in, const float* __restrict__ wghT, const float* __restrict__ wchT,
    const int32_t* __restrict__ seq_len, float* __restrict__ dgate, float* __restrict__ dcand, u64* __restrict__ xch,
    unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords) {
  zero_next_area(znext, zwords);
  constexpr int HSU = 32, G = H / HSU;
  constexpr int KS = H / 32;
  constexpr int LDH = H + 8;
  constexpr int PLB = 16 * LDH * 2;
  constexpr int IMG = 3 * PLB;                             // one gathered vector: three term images
  constexpr int SLICE = 16 * HSU;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [dc_pre P0][dc_pre P1][du_pre P0][du_pre P1][dr_pre]

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool lo = col < 8;
  const bool rev = (d == 1);
  const int ul = wave * 8 + (col & 7);
  const unsigned jw = g * HSU + ul;
  const int rbase = rg * 4 + (lo ? 0 : 2);

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = min(max(seq_len[b0 + rbase + r], 0), T_);
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  for (int i = threadIdx.x; i < 5 * 3 * 16 * LDH; i += 256) reinterpret_cast<unsigned short*>(smem)[i] = 0;
  // B fragments: lane (column = unit jw, both halves of the tile; k-group rg) of chunk ks holds W^T[32 ks + 8 rg + j][jw]
  bf16x8_t wc[KS][3], wr[KS][3], wu[KS][3];
  {
    const float* pc = wchT + (size_t)d * H * H + jw;
    const float* pg = wghT + (size_t)d * 2 * H * H + jw;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * ks + 8 * rg + j;
        unsigned short t1[3], t2[3], t3[3];
        split3(pc[(size_t)k * H], t1);
        split3(pg[(size_t)k * H], t2);
        split3(pg[(size_t)(H + k) * H], t3);
#pragma unroll
        for (int q = 0; q < 3; ++q) { wc[ks][q][j] = (short)t1[q]; wr[ks][q][j] = (short)t2[q]; wu[ks][q][j] = (short)t3[q]; }
      }
  }

  u64* xhdr = xch + (size_t)cid.c * (XHDR + 6 * G * SLICE);
  u64* xc_ = xhdr + XHDR;                                  // dc_pre: [2 parity][G][SLICE]
  u64* xu_ = xc_ + 2 * G * SLICE;                          // du_pre
  u64* xr_ = xu_ + 2 * G * SLICE;                          // dr_pre
  bool timed_out = false;
  const bool colocated = same_xcd<G>(xhdr, g, timed_out);
  const bool fast = colocated && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY: a member goes missing
  __syncthreads();

  const unsigned stride = (unsigned)B_ * ndir * H;
  unsigned ob[2];                                          // frame 0 of this lane's (row, direction, unit)
#pragma unroll
  for (int r = 0; r < 2; ++r) ob[r] = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
  const unsigned pofs = (unsigned)(wave * 128 + rbase * 8 + (col & 7));
  const unsigned lofs = (unsigned)(wave * 128 + lane);
  unsigned ldst[G - 1][2];
#pragma unroll
  for (int k = 0; k < G - 1; ++k) {
    const int gsrc = k + (k >= g ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 8 + (lane >> 3);
      ldst[k][j] = ((unsigned)(row * LDH + gsrc * HSU + wave * 8 + (lane & 7)) * 2u) ^ lds_swz(row);
    }
  }
  unsigned lown[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) lown[r] = ((unsigned)((rbase + r) * LDH + g * HSU + ul) * 2u) ^ lds_swz(rbase + r);
  const unsigned lrd = ((unsigned)(col * LDH + rg * 8) * 2u) ^ lds_swz(col);
  auto put3w = [&](char* img, unsigned off, unsigned w0, unsigned w1v) {
    *reinterpret_cast<unsigned short*>(img + off) = (unsigned short)w0;
    *reinterpret_cast<unsigned short*>(img + PLB + off) = (unsigned short)(w0 >> 16);
    *reinterpret_cast<unsigned short*>(img + 2 * PLB + off) = (unsigned short)w1v;
  };
  auto product = [&](const char* img, const bf16x8_t (&w)[KS][3], f32x4_t acc) -> f32x4_t {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t afr[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) afr[q] = *reinterpret_cast<const bf16x8_t*>(img + q * PLB + lrd + ks * 64);
      acc = mma_s3(afr, w[ks], acc);
    }
    return acc;
  };
  auto publish = [&](u64* area, int P, unsigned epoch, const float (&val)[2], char* img) {
    u64* mine = area + ((size_t)P * G + g) * SLICE;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned short t3[3];
      split3(val[r], t3);
      const unsigned w0 = (unsigned)t3[0] | ((unsigned)t3[1] << 16);
      gpublish(uoff(mine, pofs + 8u * r), ((unsigned)t3[2]) | (epoch << 16), w0, fast);
      put3w(img, lown[r], w0, (unsigned)t3[2]);
    }
  };
  auto collect = [&](u64* area, int P, unsigned epoch, char* img) {
    u64 v[G - 1][2];
#pragma unroll
    for (int k = 0; k < G - 1; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(area + ((size_t)P * G + k + (k >= g ? 1 : 0)) * SLICE, lofs + 64u * j));
    unsigned spins = 0;
#pragma unroll 1
    for (;;) {                                             // wave-uniform loop: no exec masking
      bool ok = true;
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) ok = ok && ((unsigned)(v[k][j] >> 48) == (epoch & 0xffffu));
      if (__all(ok)) break;
      if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) v[k][j] = gload(uoff(area + ((size_t)P * G + k + (k >= g ? 1 : 0)) * SLICE, lofs + 64u * j));
    }
#pragma unroll
    for (int k = 0; k < G - 1; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) put3w(img, ldst[k][j], (unsigned)v[k][j], (unsigned)(v[k][j] >> 32));
  };

  // saved activations + dout of step s, requested one step ahead (rows past their length: a valid position, unused)
  struct Saved { float dout, u, c, r, hp; };
  auto fetch = [&](int s, Saved (&sv)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool act = s >= 0 && s < len[r];
      const int fr = act ? (rev ? len[r] - 1 - s : s) : 0;
      const int fp = act ? (rev ? len[r] - s : s - 1) : 0;   // the frame of the previous step's h (s > 0)
      const unsigned o = ob[r] + (unsigned)fr * stride;
      sv[r].dout = dout[o];
      sv[r].u = u_in[o];
      sv[r].c = c_in[o];
      sv[r].r = r_in[o];
      sv[r].hp = (act && s > 0) ? hout[ob[r] + (unsigned)fp * stride] : 0.f;
    }
  };

  float dhc[2];                                            // dh_rec of this lane's two (row, unit) pairs
#pragma unroll
  for (int r = 0; r < 2; ++r) dhc[r] = d_h_final ? d_h_final[((size_t)d * B_ + b0 + rbase + r) * H + jw] : 0.f;
  char* drimg = smem + 4 * IMG;
  Saved cur[2], nxt[2];
  fetch(tmax - 1, cur);
  unsigned epoch = 0;
  for (int s = tmax - 1; s >= 0; --s) {
    ++epoch;
    const int P = (int)(epoch & 1u);
    char* dcimg = smem + P * IMG;
    char* duimg = smem + (2 + P) * IMG;
    fetch(s - 1, nxt);
    bool act[2];
    float dcp[2], dup[2], da[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      act[r] = s < len[r];
      const float dh = cur[r].dout + dhc[r];
      const float u = cur[r].u, c = cur[r].c;
      dup[r] = act[r] ? dh * (cur[r].hp - c) * u * (1.f - u) : 0.f;
      dcp[r] = act[r] ? dh * (1.f - u) * (1.f - c * c) : 0.f;
      da[r] = act[r] ? dh * u : dhc[r];
    }
    publish(xc_, P, epoch, dcp, dcimg);
    publish(xu_, P, epoch, dup, duimg);
    collect(xc_, P, epoch, dcimg);
    collect(xu_, P, epoch, duimg);
    __syncthreads();                                       // dc_pre, du_pre complete in LDS
    // ---- d(rh) = dc_pre W_c^T for the own units
    const f32x4_t a1 = product(dcimg, wc, (f32x4_t){0.f, 0.f, 0.f, 0.f});
    float drp[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float drh = lo ? a1[r] : a1[2 + r];
      const float rr = cur[r].r;
      drp[r] = act[r] ? drh * cur[r].hp * rr * (1.f - rr) : 0.f;
      da[r] += act[r] ? drh * rr : 0.f;
    }
    publish(xr_, P, epoch, drp, drimg);
    // the du_pre half of the second product does not wait for dr_pre: its multiplies run under the poll
    f32x4_t a2 = product(duimg, wu, (f32x4_t){0.f, 0.f, 0.f, 0.f});
    collect(xr_, P, epoch, drimg);
    // gradients of the pre-activations, behind the poll loop (active rows only: the caller zeroed the tensors)
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (act[r]) {
        const int fr = rev ? len[r] - 1 - s : s;
        const unsigned o = ob[r] + (unsigned)fr * stride;
        dcand[o] = dcp[r];
        const unsigned og = (o - jw) * 2u + jw;
        dgate[og] = drp[r];
        dgate[og + H] = dup[r];
      }
    __syncthreads();                                       // dr_pre complete in LDS
    // ---- dh_rec' = acc + [dr_pre | du_pre] W_g^T for the own units
    a2 = product(drimg, wr, a2);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      dhc[r] = da[r] + (lo ? a2[r] : a2[2 + r]);
      cur[r] = nxt[r];
    }
  }
  if (timed_out) atomicOr(err, 2u);
}

// BPTT, fp32 operands: the H = 512 arrangement of the bf16 kernel (each tile computed ONCE: the hh = 0 waves take the
// four own-unit tiles and hand rows 2,3 to their hh = 1 partners through LDS, the hh = 1 waves take the four tiles of
// the peer's units and publish them) -- 64 K = 4 MFMAs per wave and step instead of the 128 + 64 of the redundant form.
template <int H>
__global__ __launch_bounds__(CT8, 1) void lstm_bwd_cluster8_f32_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dhout, const f32x4_t* __restrict__ gates,
    const float* __restrict__ cs, const float* __restrict__ whpb, const float* __restrict__ peep,
    const int32_t* __restrict__ seq_len, const float* __restrict__ d_c_final,
    const float* __restrict__ d_h_final, f32x4_t* __restrict__ dgates, float* __restrict__ dpeep_part,
    u64* __restrict__ xch, unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords, float clipz) {
  static_assert(H / HS == 2, "one foreign tile per hh = 1 wave: two CUs per direction");
  zero_next_area(znext, zwords);
  constexpr int G = H / HS;
  constexpr int KC = 4 * HS / 16;            // fragments of this CU's slice of k' (16)
  constexpr int KSF = 4 * H / 16;            // fragments of the full packing
  constexpr int LDG = 4 * HS + 4;            // floats: 65 x 16 B per row
  constexpr size_t CL_U64 = XHDR + (size_t)2 * G * G * 4 * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DGB = 16 * LDG * 4;                        // bytes per dG image

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, rg = lane >> 4;
  const int hh = wave >> 2, wt = wave & 3;
  const bool rev = (d == 1);
  const float* wp = whpb + (size_t)d * H * 4 * H;
  const int ul = wt * 16 + col;
  const unsigned jw = g * HS + ul;
  const int rbase = rg * 4 + hh * 2;

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = seq_len[b0 + rbase + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  const float wci = peep ? peep[(d * 3 + 0) * H + jw] : 0.f;
  const float wcf = peep ? peep[(d * 3 + 1) * H + jw] : 0.f;
  const float wco = peep ? peep[(d * 3 + 2) * H + jw] : 0.f;

  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? stride : 0u - stride;
  unsigned oa[2], os[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned base = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    os[r] = base + (unsigned)(tmax - 1) * stride;
    oa[r] = base + (unsigned)(rev ? len[r] - tmax : tmax - 1) * stride;
  }
  {
    const f32x4_t gzero = {0.f, 0.f, 0.f, 0.f};
    for (int t = tmax; t < T_; ++t)
#pragma unroll
      for (int r = 0; r < 2; ++r) dgates[os[r] + (unsigned)(t - tmax + 1) * stride] = gzero;
  }

  float dhr[2], dcr[2], cc[2];
  float sums[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t o = ((size_t)d * B_ + b0 + rbase + r) * H + jw;
    dhr[r] = d_h_final ? d_h_final[o] : 0.f;
    dcr[r] = d_c_final ? d_c_final[o] : 0.f;
    cc[r] = (tmax > 0 && tmax - 1 < len[r]) ? cs[oa[r]] : 0.f;
  }
  // this wave's tile of W_h^T (rows k' of the CU's slice): hh = 0 the own-unit tile 4g + wt, hh = 1 tile wt of the peer
  const int peer = 1 - g;
  const int nt = (hh == 0 ? g : peer) * 4 + wt;
  f32x4_t wf[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
    wf[kc] = *reinterpret_cast<const f32x4_t*>(wp + (((size_t)nt * KSF + g * KC + kc) * 64 + lane) * 4);
  float* ownx = reinterpret_cast<float*>(smem + 2 * DGB);  // [2 parity][4 tiles][64 lanes] x (row 2, row 3)

  u64* xhdr = xch + (size_t)cid.c * CL_U64;
  bool timed_out = false;
  const bool fast = same_xcd<G>(xhdr, g, timed_out) && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY: a member goes missing
  f32x4_t* xs = reinterpret_cast<f32x4_t*>(xhdr + XHDR);   // [2][G dst][G src][4][64] x 16 B
  auto uslot = [&](int par, int dst, int src_, int tile) -> f32x4_t* {
    return xs + ((((size_t)par * G + dst) * G + src_) * 4 + tile) * 64;
  };
  const unsigned voff16 = (unsigned)lane * 16u;
  const unsigned pofs = (unsigned)lane * 2u + hh;          // u64 index of this lane's two rows in a slot
  const unsigned lwr[2] = {((unsigned)(rbase * LDG + ul * 4) * 4u) ^ lds_swz(rbase),
                           ((unsigned)((rbase + 1) * LDG + ul * 4) * 4u) ^ lds_swz(rbase + 1)};
  const unsigned lrd = ((unsigned)(col * LDG + rg * 4) * 4u) ^ lds_swz(col);

  f32x4_t pg[2];
  float pcp[2], pdh[2];
  auto prefetch = [&](int s_) {                            // oa/os already hold iteration s_
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool act = s_ < len[r];
      const bool ldp = (s_ > 0) && (s_ - 1 < len[r]);
      const unsigned offl = act ? oa[r] : os[r];
      const unsigned offn = ldp ? oa[r] + dstep : os[r];
      pg[r] = gates[offl];
      pcp[r] = cs[offn];
      pdh[r] = dhout[offl];
    }
  };
  if (tmax > 0) prefetch(tmax - 1);
  __syncthreads();

  auto step = [&](int s, auto PAR) {
    constexpr int P = decltype(PAR)::value;                // parity of THIS iteration's publish
    const int it = tmax - 1 - s;
    // ---- 1. poll for the partial the peer published at the previous iteration (parity 1-P)
    u64 pv = 0;
    if (it > 0) pv = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, peer, wt)), pofs));
    // ---- 2. everything that does not need dh
    bool act[2], ldp[2];
    float gi[2], gq[2], gf[2], go[2], cprev[2], a_o[2], b_c[2], c_g[2], c_i[2], c_f[2];
    unsigned off[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      act[r] = s < len[r];
      ldp[r] = (s > 0) && (s - 1 < len[r]);
      off[r] = act[r] ? oa[r] : os[r];
      gi[r] = pg[r][0]; gq[r] = pg[r][1]; gf[r] = pg[r][2]; go[r] = pg[r][3];
      cprev[r] = (act[r] && s > 0) ? pcp[r] : 0.f;
    }
    float tc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) tc[r] = cftanh(cc[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      a_o[r] = tc[r] * go[r] * (1.f - go[r]);
      b_c[r] = go[r] * (1.f - tc[r] * tc[r]);
      c_g[r] = gi[r] * (1.f - gq[r] * gq[r]);
      c_i[r] = gq[r] * gi[r] * (1.f - gi[r]);
      c_f[r] = cprev[r] * gf[r] * (1.f - gf[r]);
    }
    const float pdhv[2] = {pdh[0], pdh[1]}, pcpv[2] = {pcp[0], pcp[1]}, curv[2] = {cc[0], cc[1]};
#pragma unroll
    for (int r = 0; r < 2; ++r) { oa[r] += dstep; os[r] -= stride; }
    // ---- 3. finish the poll: both words must carry the previous iteration's tag
    if (it > 0) {
      const unsigned want = (((unsigned)(it - 1) >> 1) + 1u) & 1u;
      const u64 wmask = 0x0000000100000001ull, wtag = want ? wmask : 0ull;
      unsigned spins = 0;
#pragma unroll 1
      for (;;) {
        if (__all((pv & wmask) == wtag)) break;
        if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
        pv = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, peer, wt)), pofs));
      }
      dhr[0] += __uint_as_float((unsigned)pv & ~1u);
      dhr[1] += __uint_as_float((unsigned)(pv >> 32) & ~1u);
    }
    // next iteration's saved activations: requested BEHIND the poll loop (see the bf16 kernel)
    if (s > 0) prefetch(s - 1);
    // ---- 4. gate gradients of the own pairs
    float zi[2], zg[2], zf[2], zo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float dh = pdhv[r] + dhr[r];
      const float d_o = dh * a_o[r];
      const float dct = dcr[r] + dh * b_c[r] + d_o * wco;
      const float dc = (clipz > 0.f && fabsf(curv[r]) >= clipz) ? 0.f : dct;   // asr_lstm_bwd_ex
      const float d_g = dc * c_g[r], d_i = dc * c_i[r], d_f = dc * c_f[r];
      dcr[r] = act[r] ? (dc * gf[r] + d_i * wci + d_f * wcf) : dcr[r];
      dhr[r] = act[r] ? 0.f : dhr[r];
      zi[r] = act[r] ? d_i : 0.f; zg[r] = act[r] ? d_g : 0.f;
      zf[r] = act[r] ? d_f : 0.f; zo[r] = act[r] ? d_o : 0.f;
      cc[r] = ldp[r] ? pcpv[r] : 0.f;
      const f32x4_t pk = {zi[r], zg[r], zf[r], zo[r]};
      *reinterpret_cast<f32x4_t*>(smem + P * DGB + lwr[r]) = pk;
      dgates[off[r]] = pk;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      sums[0] += zi[r] * cprev[r]; sums[1] += zf[r] * cprev[r]; sums[2] += zo[r] * curv[r];
      sums[3] += zi[r]; sums[4] += zg[r]; sums[5] += zf[r]; sums[6] += zo[r];
    }
    // ---- 5. partial dh_prev of this wave's tile from the own dG slice
    if (s > 0) {
      f32x4_t ac = {0.f, 0.f, 0.f, 0.f};
      // the tile the peer waits for outranks its SIMD sibling's own-unit tile (nobody waits for that one before the
      // next step's gate math): it is published after ~half of the SIMD's MFMA time instead of all of it
      if (hh == 1 && !(kflags & 8)) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int kb = 0; kb < KC; kb += 8) {
        f32x4_t afr[8];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) afr[kc] = *reinterpret_cast<const f32x4_t*>(smem + P * DGB + lrd + (kb + kc) * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) ac = mma_f32(afr[kc], wf[kb + kc], ac);
      }
      if (hh == 1) {                                       // the peer's units: publish, every word tagged
        const unsigned tag = (((unsigned)it >> 1) + 1u) & 1u;
        f32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __uint_as_float((__float_as_uint(ac[i]) & ~1u) | tag);
        xstore16(uslot(P, peer, g, wt), voff16, o, fast);
        __builtin_amdgcn_s_setprio(0);
      } else {                                             // own units: rows 0,1 stay, rows 2,3 -> partner wave
        dhr[0] += ac[0];
        dhr[1] += ac[1];
        float* o = ownx + ((P * 4 + wt) * 64 + lane) * 2;
        o[0] = ac[2];
        o[1] = ac[3];
      }
    }
    __syncthreads();                                       // hand-over visible before the partner's next step
    if (s > 0 && hh == 1) {
      const float* o = ownx + ((P * 4 + wt) * 64 + lane) * 2;
      dhr[0] += o[0];
      dhr[1] += o[1];
    }
  };
  int s = tmax - 1;
  for (; s >= 1; s -= 2) {
    step(s, std::integral_constant<int, 0>{});
    step(s - 1, std::integral_constant<int, 1>{});
  }
  if (s == 0) step(0, std::integral_constant<int, 0>{});

  if (timed_out) atomicOr(err, 2u);
  if (dpeep_part) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      sums[k] += __shfl_xor(sums[k], 16, 64);
      sums[k] += __shfl_xor(sums[k], 32, 64);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [7][HS]
    if (hh == 1 && rg == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) red[k * HS + ul] = sums[k];
    }
    __syncthreads();
    if (hh == 0 && rg == 0) {
      float* p = dpeep_part + ((size_t)cid.tile * ndir + d) * 7 * H;
#pragma unroll
      for (int k = 0; k < 7; ++k) p[k * H + jw] = sums[k] + red[k * HS + ul];
    }
  }
}

// BPTT, fp32 operands, GENERAL form: clusters of G = H / HSU CUs (any even G), four waves per CU at HSU = 32 (one per SIMD:
// the fp32 step is bound by the SIMD's matrix pipe -- 32 cycles per K = 4 MFMA -- so a wave alone on its SIMD is what pays).
// Tile deal as in the bf16 kernel's each-tile-once arrangement: a CU owns TPC = HSU / 16 unit tiles and multiplies its dG
// slice by W_h^T for ALL H / 16 tiles; the TPC own + TPC (G - 1) foreign tiles are dealt G / 2 per wave -- an hh = 0 wave
// takes its own-unit tile (rows 2,3 handed to the hh = 1 partner through LDS) and G/2 - 1 foreign ones, an hh = 1 wave
// G / 2 foreign ones; foreign partials go out as 16-byte stores of self-tagged fp32 words, the chains of a wave's
// foreign tiles advance together and the stores follow in one run.  H = 128 runs on four CUs (the two-CU kernel above:
// 2011 us per launch at T = 778), H = 256 / 320 / 512 -- which had no fp32 cluster kernel -- on 8 / 10 / 16.
template <int H, int HSU>
__global__ __launch_bounds__(HSU * 8, 1) void lstm_bwd_cluster_f32_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dhout, const f32x4_t* __restrict__ gates,
    const float* __restrict__ cs, const float* __restrict__ whpb, const float* __restrict__ peep,
    const int32_t* __restrict__ seq_len, const float* __restrict__ d_c_final,
    const float* __restrict__ d_h_final, f32x4_t* __restrict__ dgates, float* __restrict__ dpeep_part,
    u64* __restrict__ xch, unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords, float clipz) {
  zero_next_area(znext, zwords);
  constexpr int G = H / HSU;
  static_assert(G % 2 == 0 && G >= 2 && G <= XHDR, "even number of CUs per cluster");
  constexpr int TPC = HSU / 16, NWAVES = 2 * TPC;
  constexpr int KC = 4 * HSU / 16;           // fragments (k = 16) of this CU's slice of k'
  constexpr int KSF = 4 * H / 16;            // fragments of the full packing
  constexpr int LDG = 4 * HSU + 4;           // floats per row of the dG image
  constexpr int NF = G / 2;                  // tile slots per wave
  constexpr size_t CL_U64 = XHDR + (size_t)2 * G * G * TPC * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DGB = 16 * LDG * 4;                        // bytes per dG image

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, rg = lane >> 4;
  const int hh = wave / TPC, wt = wave % TPC;
  const bool rev = (d == 1);
  const float* wp = whpb + (size_t)d * H * 4 * H;
  const int ul = wt * 16 + col;
  const unsigned jw = g * HSU + ul;
  const int rbase = rg * 4 + hh * 2;

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = seq_len[b0 + rbase + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  const float wci = peep ? peep[(d * 3 + 0) * H + jw] : 0.f;
  const float wcf = peep ? peep[(d * 3 + 1) * H + jw] : 0.f;
  const float wco = peep ? peep[(d * 3 + 2) * H + jw] : 0.f;

  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? stride : 0u - stride;
  unsigned oa[2], os[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned base = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    os[r] = base + (unsigned)(tmax - 1) * stride;
    oa[r] = base + (unsigned)(rev ? len[r] - tmax : tmax - 1) * stride;
  }
  {
    const f32x4_t gzero = {0.f, 0.f, 0.f, 0.f};
    for (int t = tmax; t < T_; ++t)
#pragma unroll
      for (int r = 0; r < 2; ++r) dgates[os[r] + (unsigned)(t - tmax + 1) * stride] = gzero;
  }

  float dhr[2], dcr[2], cc[2];
  float sums[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t o = ((size_t)d * B_ + b0 + rbase + r) * H + jw;
    dhr[r] = d_h_final ? d_h_final[o] : 0.f;
    dcr[r] = d_c_final ? d_c_final[o] : 0.f;
    cc[r] = (tmax > 0 && tmax - 1 < len[r]) ? cs[oa[r]] : 0.f;
  }
  // tile slots: i < NF - 1: foreign tile f = wave + NWAVES i;  slot NF - 1: hh = 0 the own tile, hh = 1 foreign tile
  // f = NWAVES (NF - 1) + wt.  Foreign index f -> destination CU (f / TPC, skipping g), its tile f % TPC.
  auto ftile = [&](int f) { const int q = f / TPC; return ((q + (q >= g ? 1 : 0)) * TPC) | (f % TPC); };
  int nt_f[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
    nt_f[i] = (i < NF - 1) ? ftile(wave + NWAVES * i) : (hh == 1 ? ftile(NWAVES * (NF - 1) + wt) : g * TPC + wt);
  f32x4_t wf[NF][KC];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      wf[i][kc] = *reinterpret_cast<const f32x4_t*>(wp + (((size_t)nt_f[i] * KSF + g * KC + kc) * 64 + lane) * 4);
  float* ownx = reinterpret_cast<float*>(smem + 2 * DGB);  // [2 parity][TPC tiles][64 lanes] x (row 2, row 3)

  u64* xhdr = xch + (size_t)cid.c * CL_U64;
  bool timed_out = false;
  const bool fast = same_xcd<G>(xhdr, g, timed_out) && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY: a member goes missing
  f32x4_t* xs = reinterpret_cast<f32x4_t*>(xhdr + XHDR);   // [2][G dst][G src][TPC][64] x 16 B
  auto uslot = [&](int par, int dst, int src_, int tile) -> f32x4_t* {
    return xs + ((((size_t)par * G + dst) * G + src_) * TPC + tile) * 64;
  };
  const unsigned voff16 = (unsigned)lane * 16u;
  const unsigned pofs = (unsigned)lane * 2u + hh;          // u64 index of this lane's two rows in a slot
  const unsigned lwr[2] = {((unsigned)(rbase * LDG + ul * 4) * 4u) ^ lds_swz(rbase),
                           ((unsigned)((rbase + 1) * LDG + ul * 4) * 4u) ^ lds_swz(rbase + 1)};
  const unsigned lrd = ((unsigned)(col * LDG + rg * 4) * 4u) ^ lds_swz(col);

  f32x4_t pg[2];
  float pcp[2], pdh[2];
  if (tmax > 0) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int s_ = tmax - 1;
      const bool act = s_ < len[r];
      const bool ldp = (s_ > 0) && (s_ - 1 < len[r]);
      pg[r] = gates[act ? oa[r] : os[r]];
      pcp[r] = cs[ldp ? oa[r] + dstep : os[r]];
      pdh[r] = dhout[act ? oa[r] : os[r]];
    }
  }
  __syncthreads();

  auto step = [&](int s, auto PAR) {
    // step boundary pinned (see the bf16 BPTT kernel): neutral at H = 128 / 256, 7.6 -> 4.4 ms per launch at H = 512, B = 32
    __builtin_amdgcn_sched_barrier(0);
    constexpr int P = decltype(PAR)::value;                // parity of THIS iteration's publish
    const int it = tmax - 1 - s;
    // ---- 1. polls for the partials the peers published at the previous iteration (parity 1-P)
    u64 pv[G - 1];
    if (it > 0) {
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
        pv[k] = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, k + (k >= g ? 1 : 0), wt)), pofs));
    }
    // ---- 2. everything that does not need dh
    bool act[2], ldp[2];
    float gi[2], gq[2], gf[2], go[2], cprev[2], a_o[2], b_c[2], c_g[2], c_i[2], c_f[2];
    unsigned off[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      act[r] = s < len[r];
      ldp[r] = (s > 0) && (s - 1 < len[r]);
      off[r] = act[r] ? oa[r] : os[r];
      gi[r] = pg[r][0]; gq[r] = pg[r][1]; gf[r] = pg[r][2]; go[r] = pg[r][3];
      cprev[r] = (act[r] && s > 0) ? pcp[r] : 0.f;
    }
    float tc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) tc[r] = cftanh(cc[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      a_o[r] = tc[r] * go[r] * (1.f - go[r]);
      b_c[r] = go[r] * (1.f - tc[r] * tc[r]);
      c_g[r] = gi[r] * (1.f - gq[r] * gq[r]);
      c_i[r] = gq[r] * gi[r] * (1.f - gi[r]);
      c_f[r] = cprev[r] * gf[r] * (1.f - gf[r]);
    }
    float pdh0 = pdh[0], pdh1 = pdh[1], pcp0 = pcp[0], pcp1 = pcp[1];
    const float cur0 = cc[0], cur1 = cc[1];
    // everything that reads the values fetched one iteration ago is pinned here, ahead of the next fetch (see the bf16 kernel)
    asm volatile("" : "+v"(pdh0), "+v"(pdh1), "+v"(pcp0), "+v"(pcp1), "+v"(cprev[0]), "+v"(cprev[1]));
    asm volatile("" : "+v"(c_g[0]), "+v"(c_g[1]), "+v"(c_i[0]), "+v"(c_i[1]), "+v"(c_f[0]), "+v"(c_f[1]));
    asm volatile("" : "+v"(a_o[0]), "+v"(a_o[1]), "+v"(b_c[0]), "+v"(b_c[1]), "+v"(gf[0]), "+v"(gf[1]));
#pragma unroll
    for (int r = 0; r < 2; ++r) { oa[r] += dstep; os[r] -= stride; }
    // ---- 3. finish the polls: every word must carry the previous iteration's tag
    if (it > 0) {
      const unsigned want = (((unsigned)(it - 1) >> 1) + 1u) & 1u;
      const u64 wmask = 0x0000000100000001ull, wtag = want ? wmask : 0ull;
      unsigned spins = 0;
#pragma unroll 1
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < G - 1; ++k) ok = ok && ((pv[k] & wmask) == wtag);
        if (__all(ok)) break;
        if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
          pv[k] = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, k + (k >= g ? 1 : 0), wt)), pofs));
      }
      float add0 = 0.f, add1 = 0.f;
#pragma unroll
      for (int k = 0; k < G - 1; ++k) {                     // fixed order
        add0 += __uint_as_float((unsigned)pv[k] & ~1u);
        add1 += __uint_as_float((unsigned)(pv[k] >> 32) & ~1u);
      }
      dhr[0] += add0;
      dhr[1] += add1;
    }
    // next iteration's saved activations: unconditional (the last iteration re-fetches its own rows) and pinned BEHIND
    // the poll loop by a compiler barrier
    {
      asm volatile("" ::: "memory");
      const bool more = s > 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool actn = s - 1 < len[r];
        const bool ldpn = (s - 1 > 0) && (s - 2 < len[r]);
        const unsigned offl = more ? (actn ? oa[r] : os[r]) : off[r];
        const unsigned offn = more ? (ldpn ? oa[r] + dstep : os[r]) : off[r];
        pg[r] = gates[offl];
        pcp[r] = cs[offn];
        pdh[r] = dhout[offl];
      }
    }
    // ---- 4. gate gradients of the own pairs
    const float pdhv[2] = {pdh0, pdh1}, pcpv[2] = {pcp0, pcp1}, curv[2] = {cur0, cur1};
    float zi[2], zg[2], zf[2], zo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float dh = pdhv[r] + dhr[r];
      const float d_o = dh * a_o[r];
      const float dct = dcr[r] + dh * b_c[r] + d_o * wco;
      const float dc = (clipz > 0.f && fabsf(curv[r]) >= clipz) ? 0.f : dct;   // asr_lstm_bwd_ex
      const float d_g = dc * c_g[r], d_i = dc * c_i[r], d_f = dc * c_f[r];
      dcr[r] = act[r] ? (dc * gf[r] + d_i * wci + d_f * wcf) : dcr[r];
      dhr[r] = act[r] ? 0.f : dhr[r];
      zi[r] = act[r] ? d_i : 0.f; zg[r] = act[r] ? d_g : 0.f;
      zf[r] = act[r] ? d_f : 0.f; zo[r] = act[r] ? d_o : 0.f;
      cc[r] = ldp[r] ? pcpv[r] : 0.f;
      const f32x4_t pk = {zi[r], zg[r], zf[r], zo[r]};
      *reinterpret_cast<f32x4_t*>(smem + P * DGB + lwr[r]) = pk;
      dgates[off[r]] = pk;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      sums[0] += zi[r] * cprev[r]; sums[1] += zf[r] * cprev[r]; sums[2] += zo[r] * curv[r];
      sums[3] += zi[r]; sums[4] += zg[r]; sums[5] += zf[r]; sums[6] += zo[r];
    }
    // ---- 5. partial dh_prev of this wave's tiles from the own dG slice
    if (s > 0) {
      f32x4_t afr[KC];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) afr[kc] = *reinterpret_cast<const f32x4_t*>(smem + P * DGB + lrd + kc * 64);
      __builtin_amdgcn_sched_barrier(0);
      f32x4_t ac[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) ac[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int i = 0; i < NF; ++i) ac[i] = mma_f32(afr[kc], wf[i][kc], ac[i]);
      const unsigned tag = (((unsigned)it >> 1) + 1u) & 1u;
      auto tagged = [&](const f32x4_t& a) {
        f32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __uint_as_float((__float_as_uint(a[i]) & ~1u) | tag);
        return o;
      };
      if (fast) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
          if (i < NF - 1 || hh == 1) xstore16(uslot(P, nt_f[i] / TPC, g, nt_f[i] % TPC), voff16, tagged(ac[i]), true);
      } else {
#pragma unroll
        for (int i = 0; i < NF; ++i)
          if (i < NF - 1 || hh == 1) xstore16(uslot(P, nt_f[i] / TPC, g, nt_f[i] % TPC), voff16, tagged(ac[i]), false);
      }
      if (hh == 0) {                                       // own units: rows 0,1 stay, rows 2,3 -> partner wave
        dhr[0] += ac[NF - 1][0];
        dhr[1] += ac[NF - 1][1];
        float* o = ownx + ((P * TPC + wt) * 64 + lane) * 2;
        o[0] = ac[NF - 1][2];
        o[1] = ac[NF - 1][3];
      }
    }
    __syncthreads();                                       // hand-over visible before the partner's next step
    if (s > 0 && hh == 1) {
      const float* o = ownx + ((P * TPC + wt) * 64 + lane) * 2;
      dhr[0] += o[0];
      dhr[1] += o[1];
    }
  };
  int s = tmax - 1;
  for (; s >= 1; s -= 2) {
    step(s, std::integral_constant<int, 0>{});
    step(s - 1, std::integral_constant<int, 1>{});
  }
  if (s == 0) step(0, std::integral_constant<int, 0>{});

  if (timed_out) atomicOr(err, 2u);
  if (dpeep_part) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      sums[k] += __shfl_xor(sums[k], 16, 64);
      sums[k] += __shfl_xor(sums[k], 32, 64);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [7][HSU]
    if (hh == 1 && rg == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) red[k * HSU + ul] = sums[k];
    }
    __syncthreads();
    if (hh == 0 && rg == 0) {
      float* p = dpeep_part + ((size_t)cid.tile * ndir + d) * 7 * H;
#pragma unroll
      for (int k = 0; k < 7; ++k) p[k * H + jw] = sums[k] + red[k * HSU + ul];
    }
  }
}

// BPTT of the fp32-operand clusters on the bf16 matrix pipe (three-term split, see lstm_fwd_cluster_f32s_kernel)
template <int H, int HSU>
__global__ __launch_bounds__(HSU * 8, 1) void lstm_bwd_cluster_f32s_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dhout, const f32x4_t* __restrict__ gates,
    const float* __restrict__ cs, const float* __restrict__ whpb, const float* __restrict__ peep,
    const int32_t* __restrict__ seq_len, const float* __restrict__ d_c_final,
    const float* __restrict__ d_h_final, f32x4_t* __restrict__ dgates, float* __restrict__ dpeep_part,
    u64* __restrict__ xch, unsigned* __restrict__ err, int kflags, u64* __restrict__ znext, unsigned zwords, float clipz) {
  zero_next_area(znext, zwords);
  constexpr int G = H / HSU;
  static_assert(G % 2 == 0 && G >= 2 && G <= XHDR, "even number of CUs per cluster");
  constexpr int TPC = HSU / 16, NWAVES = 2 * TPC;
  constexpr int KC = 4 * HSU / 32;           // chunks (k = 32) of this CU's slice of k'
  constexpr int KC16 = 4 * HSU / 16;         // ... in fragments of the fp32 weight packing
  constexpr int KSF = 4 * H / 16;            // fragments of the full packing
  constexpr int LDG = 4 * HSU + 8;           // bf16 elements per row of a dG term image (17 x 16 B)
  constexpr int NF = G / 2;                  // tile slots per wave
  constexpr size_t CL_U64 = XHDR + (size_t)2 * G * G * TPC * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PLB = 16 * LDG * 2;                        // bytes of one term image
  constexpr int DGB = 3 * PLB;                             // bytes per dG image (three bf16 terms)

  const ClusterId cid = cluster_id<G>(ndir, B_ / 16);
  if (!cid.valid) return;
  const int g = cid.g, d = cid.d, b0 = cid.tile * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, rg = lane >> 4;
  const int hh = wave / TPC, wt = wave % TPC;
  const bool rev = (d == 1);
  const float* wp = whpb + (size_t)d * H * 4 * H;
  const int ul = wt * 16 + col;
  const unsigned jw = g * HSU + ul;
  const int rbase = rg * 4 + hh * 2;

  int len[2];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) len[r] = seq_len[b0 + rbase + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  const float wci = peep ? peep[(d * 3 + 0) * H + jw] : 0.f;
  const float wcf = peep ? peep[(d * 3 + 1) * H + jw] : 0.f;
  const float wco = peep ? peep[(d * 3 + 2) * H + jw] : 0.f;

  const unsigned stride = (unsigned)B_ * ndir * H;
  const unsigned dstep = rev ? stride : 0u - stride;
  unsigned oa[2], os[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned base = ((unsigned)(b0 + rbase + r) * ndir + d) * H + jw;
    os[r] = base + (unsigned)(tmax - 1) * stride;
    oa[r] = base + (unsigned)(rev ? len[r] - tmax : tmax - 1) * stride;
  }
  // inactive rows read nothing that is used: their fetches go to one parked position per row (frame tmax - 1)
  const unsigned opark[2] = {os[0], os[1]};
  {
    const f32x4_t gzero = {0.f, 0.f, 0.f, 0.f};
    for (int t = tmax; t < T_; ++t)
#pragma unroll
      for (int r = 0; r < 2; ++r) dgates[os[r] + (unsigned)(t - tmax + 1) * stride] = gzero;
  }

  float dhr[2], dcr[2], cc[2];
  float sums[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t o = ((size_t)d * B_ + b0 + rbase + r) * H + jw;
    dhr[r] = d_h_final ? d_h_final[o] : 0.f;
    dcr[r] = d_c_final ? d_c_final[o] : 0.f;
    cc[r] = (tmax > 0 && tmax - 1 < len[r]) ? cs[oa[r]] : 0.f;
  }
  // tile slots: i < NF - 1: foreign tile f = wave + NWAVES i;  slot NF - 1: hh = 0 the own tile, hh = 1 foreign tile
  // f = NWAVES (NF - 1) + wt.  Foreign index f -> destination CU (f / TPC, skipping g), its tile f % TPC.
  auto ftile = [&](int f) { const int q = f / TPC; return ((q + (q >= g ? 1 : 0)) * TPC) | (f % TPC); };
  int nt_f[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
    nt_f[i] = (i < NF - 1) ? ftile(wave + NWAVES * i) : (hh == 1 ? ftile(NWAVES * (NF - 1) + wt) : g * TPC + wt);
  bf16x8_t wf[NF][KC][3];                  // W_h^T fragments, re-cut into k = 32 chunks and split into three bf16 terms
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      load_split_frag(wp + ((size_t)nt_f[i] * KSF + g * KC16) * 256, kc, rg, col, wf[i][kc]);
  float* ownx = reinterpret_cast<float*>(smem + 2 * DGB);  // [2 parity][TPC tiles][64 lanes] x (row 2, row 3)

  u64* xhdr = xch + (size_t)cid.c * CL_U64;
  bool timed_out = false;
  const bool fast = same_xcd<G>(xhdr, g, timed_out) && !(kflags & 1);
  unsigned spin_limit = (kflags & 2) ? 2000u : SPIN_LIMIT;
  if ((kflags & 2) && g == G - 1) return;                  // TEST ONLY: a member goes missing
  f32x4_t* xs = reinterpret_cast<f32x4_t*>(xhdr + XHDR);   // [2][G dst][G src][TPC][64] x 16 B
  auto uslot = [&](int par, int dst, int src_, int tile) -> f32x4_t* {
    return xs + ((((size_t)par * G + dst) * G + src_) * TPC + tile) * 64;
  };
  const unsigned voff16 = (unsigned)lane * 16u;
  const unsigned pofs = (unsigned)lane * 2u + hh;          // u64 index of this lane's two rows in a slot
  const unsigned lwr[2] = {((unsigned)(rbase * LDG + ul * 4) * 2u) ^ lds_swz(rbase),
                           ((unsigned)((rbase + 1) * LDG + ul * 4) * 2u) ^ lds_swz(rbase + 1)};
  const unsigned lrd = ((unsigned)(col * LDG + rg * 8) * 2u) ^ lds_swz(col);

  f32x4_t pg[2];
  float pcp[2], pdh[2];
  if (tmax > 0) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int s_ = tmax - 1;
      const bool act = s_ < len[r];
      const bool ldp = (s_ > 0) && (s_ - 1 < len[r]);
      pg[r] = gates[act ? oa[r] : opark[r]];
      pcp[r] = cs[ldp ? oa[r] + dstep : opark[r]];
      pdh[r] = dhout[act ? oa[r] : opark[r]];
    }
  }
  __syncthreads();

  auto step = [&](int s, auto PAR) {
    // step boundary pinned (see the bf16 BPTT kernel): neutral at H = 128 / 256, 7.6 -> 4.4 ms per launch at H = 512, B = 32
    __builtin_amdgcn_sched_barrier(0);
    constexpr int P = decltype(PAR)::value;                // parity of THIS iteration's publish
    const int it = tmax - 1 - s;
    // ---- 1. polls for the partials the peers published at the previous iteration (parity 1-P)
    u64 pv[G - 1];
    if (it > 0) {
#pragma unroll
      for (int k = 0; k < G - 1; ++k)
        pv[k] = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, k + (k >= g ? 1 : 0), wt)), pofs));
    }
    // ---- 2. everything that does not need dh
    bool act[2], ldp[2];
    float gi[2], gq[2], gf[2], go[2], cprev[2], a_o[2], b_c[2], c_g[2], c_i[2], c_f[2];
    unsigned off[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      act[r] = s < len[r];
      ldp[r] = (s > 0) && (s - 1 < len[r]);
      off[r] = act[r] ? oa[r] : os[r];
      gi[r] = pg[r][0]; gq[r] = pg[r][1]; gf[r] = pg[r][2]; go[r] = pg[r][3];
      cprev[r] = (act[r] && s > 0) ? pcp[r] : 0.f;
    }
    float tc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) tc[r] = cftanh(cc[r]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      a_o[r] = tc[r] * go[r] * (1.f - go[r]);
      b_c[r] = go[r] * (1.f - tc[r] * tc[r]);
      c_g[r] = gi[r] * (1.f - gq[r] * gq[r]);
      c_i[r] = gq[r] * gi[r] * (1.f - gi[r]);
      c_f[r] = cprev[r] * gf[r] * (1.f - gf[r]);
    }
    float pdh0 = pdh[0], pdh1 = pdh[1], pcp0 = pcp[0], pcp1 = pcp[1];
    const float cur0 = cc[0], cur1 = cc[1];
    // everything that reads the values fetched one iteration ago is pinned here, ahead of the next fetch (see the bf16 kernel)
    asm volatile("" : "+v"(pdh0), "+v"(pdh1), "+v"(pcp0), "+v"(pcp1), "+v"(cprev[0]), "+v"(cprev[1]));
    asm volatile("" : "+v"(c_g[0]), "+v"(c_g[1]), "+v"(c_i[0]), "+v"(c_i[1]), "+v"(c_f[0]), "+v"(c_f[1]));
    asm volatile("" : "+v"(a_o[0]), "+v"(a_o[1]), "+v"(b_c[0]), "+v"(b_c[1]), "+v"(gf[0]), "+v"(gf[1]));
#pragma unroll
    for (int r = 0; r < 2; ++r) { oa[r] += dstep; os[r] -= stride; }
    // ---- 3. finish the polls: every word must carry the previous iteration's tag
    if (it > 0) {
      const unsigned want = (((unsigned)(it - 1) >> 1) + 1u) & 1u;
      const u64 wmask = 0x0000000100000001ull, wtag = want ? wmask : 0ull;
      unsigned spins = 0;
#pragma unroll 1
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < G - 1; ++k) ok = ok && ((pv[k] & wmask) == wtag);
        if (__all(ok)) break;
        if (++spins > spin_limit) { timed_out = true; spin_limit = 0; break; }
#pragma unroll
        for (int k = 0; k < G - 1; ++k)
          pv[k] = gload(uoff(reinterpret_cast<const u64*>(uslot(1 - P, g, k + (k >= g ? 1 : 0), wt)), pofs));
      }
      float add0 = 0.f, add1 = 0.f;
#pragma unroll
      for (int k = 0; k < G - 1; ++k) {                     // fixed order
        add0 += __uint_as_float((unsigned)pv[k] & ~1u);
        add1 += __uint_as_float((unsigned)(pv[k] >> 32) & ~1u);
      }
      dhr[0] += add0;
      dhr[1] += add1;
    }
    // next iteration's saved activations: unconditional (the last iteration re-fetches its own rows) and pinned BEHIND
    // the poll loop by a compiler barrier
    {
      asm volatile("" ::: "memory");
      const bool more = s > 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool actn = s - 1 < len[r];
        const bool ldpn = (s - 1 > 0) && (s - 2 < len[r]);
        const unsigned offl = more ? (actn ? oa[r] : opark[r]) : off[r];
        const unsigned offn = more ? (ldpn ? oa[r] + dstep : opark[r]) : off[r];
        pg[r] = gates[offl];
        pcp[r] = cs[offn];
        pdh[r] = dhout[offl];
      }
    }
    // ---- 4. gate gradients of the own pairs
    const float pdhv[2] = {pdh0, pdh1}, pcpv[2] = {pcp0, pcp1}, curv[2] = {cur0, cur1};
    float zi[2], zg[2], zf[2], zo[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float dh = pdhv[r] + dhr[r];
      const float d_o = dh * a_o[r];
      const float dct = dcr[r] + dh * b_c[r] + d_o * wco;
      const float dc = (clipz > 0.f && fabsf(curv[r]) >= clipz) ? 0.f : dct;   // asr_lstm_bwd_ex
      const float d_g = dc * c_g[r], d_i = dc * c_i[r], d_f = dc * c_f[r];
      dcr[r] = act[r] ? (dc * gf[r] + d_i * wci + d_f * wcf) : dcr[r];
      dhr[r] = act[r] ? 0.f : dhr[r];
      zi[r] = act[r] ? d_i : 0.f; zg[r] = act[r] ? d_g : 0.f;
      zf[r] = act[r] ? d_f : 0.f; zo[r] = act[r] ? d_o : 0.f;
      cc[r] = ldp[r] ? pcpv[r] : 0.f;
      const f32x4_t pk = {zi[r], zg[r], zf[r], zo[r]};
      typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
      us4_t tq[3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned short t[3];
        split3(pk[q], t);
#pragma unroll
        for (int k = 0; k < 3; ++k) tq[k][q] = t[k];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<us4_t*>(smem + P * DGB + k * PLB + lwr[r]) = tq[k];
      dgates[off[r]] = pk;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      sums[0] += zi[r] * cprev[r]; sums[1] += zf[r] * cprev[r]; sums[2] += zo[r] * curv[r];
      sums[3] += zi[r]; sums[4] += zg[r]; sums[5] += zf[r]; sums[6] += zo[r];
    }
    // ---- 5. partial dh_prev of this wave's tiles from the own dG slice
    if (s > 0) {
      bf16x8_t afr[KC][3];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          afr[kc][k] = *reinterpret_cast<const bf16x8_t*>(smem + P * DGB + k * PLB + lrd + kc * 64);
      __builtin_amdgcn_sched_barrier(0);
      f32x4_t ac[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) ac[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int i = 0; i < NF; ++i) ac[i] = mma_s3(afr[kc], wf[i][kc], ac[i]);
      const unsigned tag = (((unsigned)it >> 1) + 1u) & 1u;
      auto tagged = [&](const f32x4_t& a) {
        f32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __uint_as_float((__float_as_uint(a[i]) & ~1u) | tag);
        return o;
      };
      if (fast) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
          if (i < NF - 1 || hh == 1) xstore16(uslot(P, nt_f[i] / TPC, g, nt_f[i] % TPC), voff16, tagged(ac[i]), true);
      } else {
#pragma unroll
        for (int i = 0; i < NF; ++i)
          if (i < NF - 1 || hh == 1) xstore16(uslot(P, nt_f[i] / TPC, g, nt_f[i] % TPC), voff16, tagged(ac[i]), false);
      }
      if (hh == 0) {                                       // own units: rows 0,1 stay, rows 2,3 -> partner wave
        dhr[0] += ac[NF - 1][0];
        dhr[1] += ac[NF - 1][1];
        float* o = ownx + ((P * TPC + wt) * 64 + lane) * 2;
        o[0] = ac[NF - 1][2];
        o[1] = ac[NF - 1][3];
      }
    }
    __syncthreads();                                       // hand-over visible before the partner's next step
    if (s > 0 && hh == 1) {
      const float* o = ownx + ((P * TPC + wt) * 64 + lane) * 2;
      dhr[0] += o[0];
      dhr[1] += o[1];
    }
  };
  int s = tmax - 1;
  for (; s >= 1; s -= 2) {
    step(s, std::integral_constant<int, 0>{});
    step(s - 1, std::integral_constant<int, 1>{});
  }
  if (s == 0) step(0, std::integral_constant<int, 0>{});

  if (timed_out) atomicOr(err, 2u);
  if (dpeep_part) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      sums[k] += __shfl_xor(sums[k], 16, 64);
      sums[k] += __shfl_xor(sums[k], 32, 64);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [7][HSU]
    if (hh == 1 && rg == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) red[k * HSU + ul] = sums[k];
    }
    __syncthreads();
    if (hh == 0 && rg == 0) {
      float* p = dpeep_part + ((size_t)cid.tile * ndir + d) * 7 * H;
#pragma unroll
      for (int k = 0; k < 7; ++k) p[k * H + jw] = sums[k] + red[k * HSU + ul];
    }
  }
}

static unsigned long long* g_cdbg_host = nullptr;
static void cdbg_setup() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* e = getenv("ASR_LSTM_DBG");
  if (!(e && e[0] == '1')) return;
  (void)hipMalloc(&g_cdbg_host, 1280 * sizeof(unsigned long long));
  (void)hipMemset(g_cdbg_host, 0, 1280 * sizeof(unsigned long long));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cdbg), &g_cdbg_host, sizeof(g_cdbg_host));
}
static int g_dflags = -1;
// bit 4 (16): force the placement-independent write-through exchange
// bit 5 (32): invert the default choice of the EARLY forward variant (A/B measurements)
// bit 7 (128): BPTT kernel requests the next iteration's saved activations ahead of the poll loop (the old place; A/B)
// bit 8 (256): fp32 BPTT kernel without the priority of the published tile's MFMAs (A/B)
// bit 9 (512): H = 256 / 512 clusters of H/64 CUs x eight waves instead of H/32 CUs x four waves
// bit 10 (1024): forward all-gather with 8-byte {step, payload} granules instead of 4-byte self-tagged words (A/B, tests)
// bit 11 (2048): inverts the default choice of the BPTT reduce-scatter slot layout (consumer-major source pairs, XP) (A/B, tests)
// bit 12 (4096): fp32 operands on the exact-fp32 MFMA kernels instead of the three-term bf16 split (A/B, tests)
// bit 6 (64): TEST ONLY -- the last member of every cluster leaves right after the placement handshake and the
//             spin limit drops to 2000 polls, so every hand-off times out (tests/test_gpu_ops.py checks that the
//             error word is raised and surfaces as an exception)
static int dbg_flags() {
  if (g_dflags < 0) { const char* e = getenv("ASR_LSTM_DFLAGS"); g_dflags = e ? atoi(e) : 0; }
  return g_dflags;
}
static int kernel_flags() {
  return ((dbg_flags() & 16) ? 1 : 0) | ((dbg_flags() & 64) ? 2 : 0) | ((dbg_flags() & 128) ? 4 : 0) |
         ((dbg_flags() & 256) ? 8 : 0);
}
static bool cluster_enabled() {
  static const bool on = [] { const char* e = getenv("ASR_LSTM_CLUSTER"); return !(e && e[0] == '0'); }();
  return on;
}
// H = 320 (the width of most of the reference's recipes): five CUs per (direction, tile).  ASR_LSTM_CLUSTER_320=0
// keeps the single-CU kernels for that width (A/B).
static bool cluster_320_enabled() {
  static const bool on = [] { const char* e = getenv("ASR_LSTM_CLUSTER_320"); return !(e && e[0] == '0'); }();
  return on;
}
}  // namespace

static constexpr size_t XCH_BYTES = ASR_XCH_BYTES;
static constexpr size_t XCH_HALF = ((XCH_BYTES - 256) / 2) & ~(size_t)255;   // one of the two exchange areas

// The area this launch runs on (clean by construction; the memset is the fallback for a handle whose bookkeeping was
// reset) and the stretch of the other one it has to zero for its successor (zero_next_area).
struct XchAreas { u64* area; u64* znext; unsigned zwords; };
static XchAreas xch_take(asr_handle* h, char* base, size_t need, hipStream_t st) {
  const int a = h->xch_next & 1, o = a ^ 1;
  char* area = base + 256 + (size_t)a * XCH_HALF;
  char* other = base + 256 + (size_t)o * XCH_HALF;
  if (h->xch_dirty[a]) {
    (void)hipMemsetAsync(area, 0, h->xch_dirty[a], st);
    h->xch_dirty[a] = 0;
  }
  XchAreas x;
  x.area = (u64*)area;
  x.znext = (u64*)other;
  x.zwords = (unsigned)(h->xch_dirty[o] / sizeof(u64));
  h->xch_dirty[o] = 0;
  h->xch_dirty[a] = (need + 7) & ~(size_t)7;
  h->xch_next = o;
  return x;
}

// ASR_LSTM_FWD_HS=32: forward recurrence on clusters of H/32 CUs with four waves each (one per SIMD) instead of H/64
// CUs with eight (A/B switch)
// Units per CU of the H = 256 / 512 clusters: 32 (default since round 3: H/32 CUs with four waves each, one per SIMD) or
// 64 (H/64 CUs with eight waves; ASR_LSTM_HS=64, ASR_LSTM_FWD_HS / ASR_LSTM_BWD_HS for one pass only, or
// ASR_LSTM_DFLAGS bit 9 at run time -- the tests run both).  Measured (profiles/r03_cluster_hs32.md): forward
// 907 -> 866 us and BPTT 953 -> 929 us per launch at H = 256 (T = 778), 1791 -> 1374 and 2354 -> 1833 us at H = 512.
static int units_per_cu(const char* specific) {
  if (dbg_flags() & 512) return 64;
  const char* e = getenv(specific);
  if (!e) e = getenv("ASR_LSTM_HS");
  return (e && atoi(e) == 64) ? 64 : 32;
}
static int fwd_units_per_cu() { return units_per_cu("ASR_LSTM_FWD_HS"); }
// forward all-gather word format: 4-byte self-tagged words unless ASR_LSTM_XW=0 or ASR_LSTM_DFLAGS bit 10 (run time, tests)
static bool fwd_xw_enabled() {
  static const bool on = [] { const char* e = getenv("ASR_LSTM_XW"); return !(e && e[0] == '0'); }();
  return on && !(dbg_flags() & 1024);
}
static int bwd_units_per_cu() { return units_per_cu("ASR_LSTM_BWD_HS"); }
// BPTT reduce-scatter slot layout: consumer-major source pairs (XP) or one 16-byte slot per (source, tile).  Measured
// (round 4, us per BPTT launch, T = 778): H = 320 on five CUs x eight waves 1107 -> 1064 with XP, but H = 256 on eight CUs x
// four waves 851 -> 940 and H = 512 on sixteen 1520 -> 1790: with one wave per SIMD the doubled store count of the
// producers (two 8-byte stores per tile instead of one 16-byte store) sits on the chain between the MFMAs and the peers'
// polls and costs more than the halved poll count saves.  Default: XP only on the eight-wave clusters (units per CU 64);
// ASR_LSTM_XP=1 / 0 forces it on / off everywhere, ASR_LSTM_DFLAGS bit 11 inverts the choice at run time (tests, A/B).
static bool bwd_xp_enabled(int hsu) {
  static const int env = [] { const char* e = getenv("ASR_LSTM_XP"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  const bool def = env >= 0 ? env == 1 : hsu == 64;
  return def != ((dbg_flags() & 2048) != 0);
}

template <int H, int HSU = 64>
static bool cluster_fwd_launch(asr_handle* h, int T, int B, int ndir, const float* xproj, const void* whp,
                               const float* peep, const int32_t* seq_len, float fb, float clip, void* gates,
                               void* hout, float* cs, float* cf, float* hf, hipStream_t st) {
  constexpr int G = H / HSU;
  static_assert(G <= XHDR, "placement header too small");
  const int ncl = (B / 16) * ndir;
  const size_t need = (size_t)ncl * (XHDR + 2 * G * 16 * (HSU / 2)) * sizeof(u64);
  if ((size_t)T * B * ndir * H >= (1ull << 31) || h->scratch_bytes < XCH_BYTES || need > XCH_HALF ||
      (int)cluster_grid(G, ncl) > h->num_cu)   // every member must be resident at once: 1 workgroup per CU
    return false;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  const XchAreas xa = xch_take(h, base, need, st);
  // EARLY (own-slice k-chunks multiplied under the L2 hop): measured at H = 256 (round 2, cfg B): 970 -> 938 us per
  // launch, at H = 320: 1069 -> 969; default at both.  ASR_LSTM_DFLAGS bit 5 (32) inverts the default for A/B measurements.
  const bool early = (H == 256 || H == 320 || HSU == 32) != ((dbg_flags() & 32) != 0);   // H = 320: 1069 -> 969 us per launch
  // forward all-gather in 4-byte self-tagged words (default) or 8-byte {step, payload} granules (ASR_LSTM_XW=0, flag bit 10)
  const bool xw = fwd_xw_enabled();
  auto k = g_cdbg_host ? (xw ? lstm_fwd_cluster8_kernel<H, true, false, HSU, 0, true> : lstm_fwd_cluster8_kernel<H, true, false, HSU, 0, false>)
                       : (early ? (xw ? lstm_fwd_cluster8_kernel<H, false, true, HSU, 0, true> : lstm_fwd_cluster8_kernel<H, false, true, HSU, 0, false>)
                                : (xw ? lstm_fwd_cluster8_kernel<H, false, false, HSU, 0, true> : lstm_fwd_cluster8_kernel<H, false, false, HSU, 0, false>));
  // scheduling barrier between the MFMA phase and the gate math (FPIN bit 1) at H = 256 on four waves; masks measured
  // there (us per launch): none 806, {0} 808, {1} 777, {2} 855, {3} 808, {0,1} 779, {1,2} 799, all 808; at H = 512 every mask
  // is slower than none (1350: 1355 .. 1399), and so is this one at H = 320 on eight waves (968 -> 984).  On top of {1}:
  // a second barrier behind the gate math (bit 5) 781 -> 776; behind the DPP half swap (bit 4) 850, behind the publish
  // (bit 6) 799, behind the poll loop (bit 7) 797, behind the LDS staging (bit 8) 797.  H = 512 with bit 5 / 6 / 7 alone:
  // 1365 / 1369 / 1391 against 1356
  if constexpr (HSU == 32 && H == 256) {
    if (early && !g_cdbg_host)
      k = xw ? lstm_fwd_cluster8_kernel<H, false, true, HSU, 2 | 32, true> : lstm_fwd_cluster8_kernel<H, false, true, HSU, 2 | 32, false>;
#ifdef ASR_LSTM_ABLATE
    // scripts/probe_lstm_ablate.py: ASR_LSTM_ABL_FWD selects an ablated build of the headline kernel (garbage results)
    if (const char* e = getenv("ASR_LSTM_ABL_FWD")) {
      switch (atoi(e)) {
#define ABLF(n) case n: k = lstm_fwd_cluster8_kernel<H, false, true, HSU, 2 | 32, true, n>; break;
        ABLF(1) ABLF(2) ABLF(4) ABLF(8) ABLF(16) ABLF(32) ABLF(64) ABLF(128) ABLF(256) ABLF(512)
        ABLF(513) ABLF(515) ABLF(519) ABLF(527) ABLF(545) ABLF(769) ABLF(1023) ABLF(6) ABLF(14) ABLF(48) ABLF(800)
#undef ABLF
        default: break;
      }
    }
#endif
  }
#ifdef ASR_LSTM_ABLATE
  if constexpr (HSU == 32 && H == 512) {                   // the width of cfg C / D / E: a shorter list
    if (const char* e = getenv("ASR_LSTM_ABL_FWD")) {
      if (early && xw && !g_cdbg_host) {
        switch (atoi(e)) {
#define ABLF(n) case n: k = lstm_fwd_cluster8_kernel<H, false, true, HSU, 0, true, n>; break;
          ABLF(1) ABLF(8) ABLF(16) ABLF(64) ABLF(512) ABLF(6) ABLF(513) ABLF(519) ABLF(527) ABLF(545) ABLF(1023)
#undef ABLF
          default: break;
        }
      }
    }
  }
#endif
  // (padding the LDS request past half a CU so that two 4-wave members can never share one was measured: no
  // difference, 866.7 vs 867.6 us -- the dispatcher spreads the members over the CUs by itself)
  const size_t lds = (size_t)2 * 16 * (H + 8) * 2;
  hipLaunchKernelGGL(k, dim3(cluster_grid(G, ncl)), dim3(HSU * 8), lds, st, T, B, ndir,
                     (const f32x4_t*)xproj, (const bf16_t*)whp, peep, seq_len, fb, clip, (cbf16x4_t*)gates,
                     (bf16_t*)hout, cs, cf, hf, xa.area, (unsigned*)base, kernel_flags(), xa.znext, xa.zwords);
  return true;
}

bool asr_cluster_fwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* xproj,
                         const void* whp, const float* peep, const int32_t* seq_len, float fb,
                         float clip, void* gates, void* hout, float* cs, float* cf, float* hf,
                         hipStream_t st) {
  if (!cluster_enabled() || (H != 256 && H != 512 && !(H == 320 && cluster_320_enabled()))) return false;
  cdbg_setup();
  if (H == 320) return cluster_fwd_launch<320>(h, T, B, ndir, xproj, whp, peep, seq_len, fb, clip, gates, hout, cs, cf, hf, st);
  if (fwd_units_per_cu() == 32) {
    const bool ok = H == 512
        ? cluster_fwd_launch<512, 32>(h, T, B, ndir, xproj, whp, peep, seq_len, fb, clip, gates, hout, cs, cf, hf, st)
        : cluster_fwd_launch<256, 32>(h, T, B, ndir, xproj, whp, peep, seq_len, fb, clip, gates, hout, cs, cf, hf, st);
    if (ok) return true;
  }
  return H == 512 ? cluster_fwd_launch<512>(h, T, B, ndir, xproj, whp, peep, seq_len, fb, clip, gates, hout, cs, cf, hf, st)
                  : cluster_fwd_launch<256>(h, T, B, ndir, xproj, whp, peep, seq_len, fb, clip, gates, hout, cs, cf, hf, st);
}

template <int H, int HSU = 64>
static bool cluster_bwd_launch(asr_handle* h, int T, int B, int ndir, const float* dhout, const void* gates,
                               const float* cs, const void* whpb, const float* peep, const int32_t* seq_len,
                               const float* dcf, const float* dhf, void* dgates, float* dpeep_part,
                               hipStream_t st) {
  constexpr int G = H / HSU;
  static_assert(G <= XHDR, "placement header too small");
  const int ncl = (B / 16) * ndir;
  const size_t need = (size_t)ncl * (XHDR + (size_t)2 * G * G * (HSU / 16) * 64 * 2) * sizeof(u64);
  if ((size_t)T * B * ndir * H >= (1ull << 31) || h->scratch_bytes < XCH_BYTES || need > XCH_HALF ||
      (int)cluster_grid(G, ncl) > h->num_cu)   // every member must be resident at once: 1 workgroup per CU
    return false;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  const XchAreas xa = xch_take(h, base, need, st);
  // two dG images; the H = 512 form adds the own-tile hand-over buffer behind them
  const size_t lds = (size_t)2 * 16 * (4 * HSU + 8) * 2 + ((G == 8 && HSU == 64) ? 2 * 4 * 64 * 8 : 0);
  // reduce-scatter slots in the consumer-major paired layout (default) or one 16-byte slot per (source, tile) (ASR_LSTM_XP=0)
  const bool xp = bwd_xp_enabled(HSU);
  auto k = g_cdbg_host ? (xp ? lstm_bwd_cluster8_kernel<H, true, HSU, -1, true> : lstm_bwd_cluster8_kernel<H, true, HSU, -1, false>)
                       : (xp ? lstm_bwd_cluster8_kernel<H, false, HSU, -1, true> : lstm_bwd_cluster8_kernel<H, false, HSU, -1, false>);
  if (h->bptt_clip > 0.f)   // asr_lstm_bwd_ex: the gradient-blocking clip of the projected LSTMCell layers
    k = xp ? lstm_bwd_cluster8_kernel<H, false, HSU, -1, true, 0, true> : lstm_bwd_cluster8_kernel<H, false, HSU, -1, false, 0, true>;
#ifdef ASR_LSTM_ABLATE
  if constexpr (HSU == 32 && H == 512) {
    if (const char* e = getenv("ASR_LSTM_ABL_BWD")) {
      switch (atoi(e)) {
#define ABLB(n) case n: k = lstm_bwd_cluster8_kernel<H, false, HSU, -1, false, n>; break;
        ABLB(1) ABLB(8) ABLB(64) ABLB(128) ABLB(512) ABLB(6) ABLB(513) ABLB(519) ABLB(527) ABLB(545) ABLB(641) ABLB(1023)
#undef ABLB
        default: break;
      }
    }
  }
  if constexpr (HSU == 32 && H == 256) {
    if (const char* e = getenv("ASR_LSTM_ABL_BWD")) {
      switch (atoi(e)) {
#define ABLB(n) case n: k = lstm_bwd_cluster8_kernel<H, false, HSU, -1, false, n>; break;
        ABLB(1) ABLB(2) ABLB(4) ABLB(8) ABLB(16) ABLB(32) ABLB(64) ABLB(128) ABLB(256) ABLB(512)
        ABLB(513) ABLB(515) ABLB(519) ABLB(527) ABLB(545) ABLB(641) ABLB(1023) ABLB(6) ABLB(14) ABLB(144) ABLB(80)
#undef ABLB
        default: break;
      }
    }
  }
#endif
  hipLaunchKernelGGL(k, dim3(cluster_grid(G, ncl)), dim3(HSU * 8), lds, st, T, B, ndir, dhout,
                     (const cbf16x4_t*)gates, cs, (const bf16_t*)whpb, peep, seq_len, dcf, dhf,
                     (cbf16x4_t*)dgates, dpeep_part, xa.area, (unsigned*)base, kernel_flags(), xa.znext, xa.zwords,
                     h->bptt_clip);
  return true;
}

bool asr_cluster_bwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* dhout,
                         const void* gates, const float* cs, const void* whpb, const float* peep,
                         const int32_t* seq_len, const float* dcf, const float* dhf, void* dgates,
                         float* dpeep_part, hipStream_t st) {
  if (!cluster_enabled() || (H != 256 && H != 512 && !(H == 320 && cluster_320_enabled()))) return false;
  cdbg_setup();
  if (H == 320)
    return cluster_bwd_launch<320>(h, T, B, ndir, dhout, gates, cs, whpb, peep, seq_len, dcf, dhf, dgates, dpeep_part, st);
  if (bwd_units_per_cu() == 32) {
    // (falls through to the 64-unit form when the exchange area of the wider clusters does not fit)
    const bool ok = H == 512
        ? cluster_bwd_launch<512, 32>(h, T, B, ndir, dhout, gates, cs, whpb, peep, seq_len, dcf, dhf, dgates, dpeep_part, st)
        : cluster_bwd_launch<256, 32>(h, T, B, ndir, dhout, gates, cs, whpb, peep, seq_len, dcf, dhf, dgates, dpeep_part, st);
    if (ok) return true;
  }
  return H == 512 ? cluster_bwd_launch<512>(h, T, B, ndir, dhout, gates, cs, whpb, peep, seq_len, dcf, dhf, dgates, dpeep_part, st)
                  : cluster_bwd_launch<256>(h, T, B, ndir, dhout, gates, cs, whpb, peep, seq_len, dcf, dhf, dgates, dpeep_part, st);
}

// fp32 operands: H = 128 on two CUs per (direction, tile).  ASR_LSTM_CLUSTER_F32=0 keeps the single-CU kernels (A/B).
static bool cluster_f32_enabled() {
  static const bool on = [] { const char* e = getenv("ASR_LSTM_CLUSTER_F32"); return !(e && e[0] == '0'); }();
  return on && cluster_enabled();
}

// fp32 operands as three bf16 terms on the bf16 matrix pipe (H = 128 / 256, four waves per CU); ASR_LSTM_F32_SPLIT=0 or
// ASR_LSTM_DFLAGS bit 12 (run time, tests) keeps the exact-fp32 MFMA kernels
static bool cluster_f32_split_enabled() {
  static const bool on = [] { const char* e = getenv("ASR_LSTM_F32_SPLIT"); return !(e && e[0] == '0'); }();
  return on && !(dbg_flags() & 4096);
}

template <int HH, int HSU>
static bool cluster_fwd_f32_launch(asr_handle* h, int T, int B, int ndir, const float* xproj, const void* whp,
                                   const float* peep, const int32_t* seq_len, float fb, float clip, void* gates,
                                   void* hout, float* cs, float* cf, float* hf, hipStream_t st) {
  constexpr int G = HH / HSU;
  static_assert(G <= XHDR, "placement header too small");
  const int ncl = (B / 16) * ndir;
  const size_t need = (size_t)ncl * (XHDR + 2 * G * 16 * HSU) * sizeof(u64);
  if ((size_t)T * B * ndir * HH >= (1ull << 31) || h->scratch_bytes < XCH_BYTES || need > XCH_HALF ||
      (int)cluster_grid(G, ncl) > h->num_cu)
    return false;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  const XchAreas xa = xch_take(h, base, need, st);
  // EARLY own-slice products by default, except H = 512 where their extra live accumulators push the 256 weight
  // registers into scratch (measured 5.93 vs 5.36 ms per 778-step launch); ASR_LSTM_DFLAGS bit 5 inverts (A/B)
  const bool early = ((dbg_flags() & 32) == 0) != (HH >= 512);
  if constexpr (HSU == 32 && (HH == 128 || HH == 256)) {
    if (cluster_f32_split_enabled() && T < 65536) {   // round 5: the same recurrence on the bf16 matrix pipe (three-term split; 16-bit step tags)
      auto ks = early ? lstm_fwd_cluster_f32s_kernel<HH, true, HSU> : lstm_fwd_cluster_f32s_kernel<HH, false, HSU>;
      const size_t lds_s = (size_t)2 * 3 * 16 * (HH + 8) * 2;
      hipLaunchKernelGGL(ks, dim3(cluster_grid(G, ncl)), dim3(HSU * 8), lds_s, st, T, B, ndir, (const f32x4_t*)xproj,
                         (const float*)whp, peep, seq_len, fb, clip, (f32x4_t*)gates, (float*)hout, cs, cf, hf, xa.area,
                         (unsigned*)base, kernel_flags(), xa.znext, xa.zwords);
      return true;
    }
  }
  auto k = early ? lstm_fwd_cluster8_f32_kernel<HH, true, HSU> : lstm_fwd_cluster8_f32_kernel<HH, false, HSU>;
  const size_t lds = (size_t)2 * 16 * (HH + 4) * 4;
  if (lds > ((size_t)64 << 10))
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(cluster_grid(G, ncl)), dim3(HSU * 8), lds, st, T, B, ndir, (const f32x4_t*)xproj,
                     (const float*)whp, peep, seq_len, fb, clip, (f32x4_t*)gates, (float*)hout, cs, cf, hf, xa.area,
                     (unsigned*)base, kernel_flags(), xa.znext, xa.zwords);
  return true;
}

// fp32 operands: H = 128 / 256 / 320 / 512 on clusters of H/32 CUs x four waves (default), H = 128 also on two CUs x eight
// waves (ASR_LSTM_HS=64 / flag bit 9).  ASR_LSTM_CLUSTER_F32=0 keeps the single-CU kernels (A/B);
// ASR_LSTM_CLUSTER_F32_WIDE=0 keeps them for H > 128 only.
static bool cluster_f32_wide_enabled() {
  static const bool on = [] { const char* e = getenv("ASR_LSTM_CLUSTER_F32_WIDE"); return !(e && e[0] == '0'); }();
  return on;
}
#define ASR_F32_ARGS_F h, T, B, ndir, xproj, whp, peep, seq_len, fb, clip, gates, hout, cs, cf, hf, st
bool asr_cluster_fwd_f32_try(asr_handle* h, int T, int B, int H, int ndir, const float* xproj,
                             const void* whp, const float* peep, const int32_t* seq_len, float fb,
                             float clip, void* gates, void* hout, float* cs, float* cf, float* hf,
                             hipStream_t st) {
  if (!cluster_f32_enabled()) return false;
  if (fwd_units_per_cu() == 32) {
    if (H == 128) return cluster_fwd_f32_launch<128, 32>(ASR_F32_ARGS_F);
    if (!cluster_f32_wide_enabled()) return false;
    if (H == 256) return cluster_fwd_f32_launch<256, 32>(ASR_F32_ARGS_F);
    if (H == 320) return cluster_fwd_f32_launch<320, 32>(ASR_F32_ARGS_F);
    if (H == 512) return cluster_fwd_f32_launch<512, 32>(ASR_F32_ARGS_F);
    return false;
  }
  return H == 128 ? cluster_fwd_f32_launch<128, 64>(ASR_F32_ARGS_F) : false;
}
#undef ASR_F32_ARGS_F

template <int HH, int HSU>
static bool cluster_bwd_f32_launch(asr_handle* h, int T, int B, int ndir, const float* dhout, const void* gates,
                                   const float* cs, const void* whpb, const float* peep, const int32_t* seq_len,
                                   const float* dcf, const float* dhf, void* dgates, float* dpeep_part, hipStream_t st) {
  constexpr int G = HH / HSU, TPC = HSU / 16;
  const int ncl = (B / 16) * ndir;
  const size_t need = (size_t)ncl * (XHDR + (size_t)2 * G * G * TPC * 64 * 2) * sizeof(u64);
  if ((size_t)T * B * ndir * HH >= (1ull << 31) || h->scratch_bytes < XCH_BYTES || need > XCH_HALF ||
      (int)cluster_grid(G, ncl) > h->num_cu)
    return false;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  const XchAreas xa = xch_take(h, base, need, st);
  if constexpr (HSU == 32 && (HH == 128 || HH == 256)) {
    if (cluster_f32_split_enabled()) {
      const size_t lds_s = (size_t)2 * 3 * 16 * (4 * HSU + 8) * 2 + (size_t)2 * TPC * 64 * 8;   // two x three term images + hand-over
      hipLaunchKernelGGL((lstm_bwd_cluster_f32s_kernel<HH, HSU>), dim3(cluster_grid(G, ncl)), dim3(HSU * 8), lds_s, st, T,
                         B, ndir, dhout, (const f32x4_t*)gates, cs, (const float*)whpb, peep, seq_len, dcf, dhf,
                         (f32x4_t*)dgates, dpeep_part, xa.area, (unsigned*)base, kernel_flags(), xa.znext, xa.zwords, h->bptt_clip);
      return true;
    }
  }
  const size_t lds = (size_t)2 * 16 * (4 * HSU + 4) * 4 + (size_t)2 * TPC * 64 * 8;   // two dG images + the hand-over buffer
  hipLaunchKernelGGL((lstm_bwd_cluster_f32_kernel<HH, HSU>), dim3(cluster_grid(G, ncl)), dim3(HSU * 8), lds, st, T, B, ndir,
                     dhout, (const f32x4_t*)gates, cs, (const float*)whpb, peep, seq_len, dcf, dhf, (f32x4_t*)dgates,
                     dpeep_part, xa.area, (unsigned*)base, kernel_flags(), xa.znext, xa.zwords, h->bptt_clip);
  return true;
}

// GRU forward on clusters of H / 32 CUs (H = 64 / 128 / 256 / 320; B a multiple of 16).  ASR_GRU_CLUSTER=0 keeps the single-CU
// persistent kernel (A/B, and the tests run both).  false = not applicable, nothing launched.
bool asr_cluster_gru_fwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* xg, const float* xc,
                             const float* wgh, const float* wch, const int32_t* seq_len, float* r, float* u, float* c,
                             float* rh, float* hout, float* h_final, hipStream_t st) {
  const char* env_c = getenv("ASR_GRU_CLUSTER");          // (read per call: the A-B test flips it inside one process)
  if ((env_c && env_c[0] == '0') || !cluster_f32_enabled() || (H != 64 && H != 128 && H != 256 && H != 320) || T < 1 || T >= 65536) return false;
  const int G = H / 32, ncl = (B / 16) * ndir;
  const size_t need = (size_t)ncl * (XHDR + (size_t)4 * G * 16 * 32) * sizeof(u64);
  if ((size_t)T * B * ndir * 2 * H >= (1ull << 31) || h->scratch_bytes < XCH_BYTES || need > XCH_HALF ||
      (int)cluster_grid(G, ncl) > h->num_cu)
    return false;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  const XchAreas xa = xch_take(h, base, need, st);
  const size_t lds = (size_t)3 * 3 * 16 * (H + 8) * 2;
#define ASR_GRU_CL(HH)                                                                                              \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)gru_fwd_cluster_kernel<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(gru_fwd_cluster_kernel<HH>, dim3(cluster_grid(HH / 32, ncl)), dim3(256), lds, st, T, B, ndir, xg, xc, \
                       wgh, wch, seq_len, r, u, c, rh, hout, h_final, xa.area, (unsigned*)base, kernel_flags(), xa.znext, \
                       xa.zwords);                                                                                   \
  } while (0)
  if (H == 64) ASR_GRU_CL(64); else if (H == 128) ASR_GRU_CL(128); else if (H == 256) ASR_GRU_CL(256); else ASR_GRU_CL(320);
#undef ASR_GRU_CL
  return true;
}

bool asr_cluster_gru_bwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* dout, const float* d_h_final,
                             const float* hout, const float* r, const float* u, const float* c, const float* wghT,
                             const float* wchT, const int32_t* seq_len, float* dgate, float* dcand, hipStream_t st) {
  const char* env_c = getenv("ASR_GRU_CLUSTER");          // (read per call: the A-B test flips it inside one process)
  if ((env_c && env_c[0] == '0') || !cluster_f32_enabled() || (H != 64 && H != 128 && H != 256 && H != 320) || T < 1 || T >= 65536) return false;
  const int G = H / 32, ncl = (B / 16) * ndir;
  const size_t need = (size_t)ncl * (XHDR + (size_t)6 * G * 16 * 32) * sizeof(u64);
  if ((size_t)T * B * ndir * 2 * H >= (1ull << 31) || h->scratch_bytes < XCH_BYTES || need > XCH_HALF ||
      (int)cluster_grid(G, ncl) > h->num_cu)
    return false;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  const XchAreas xa = xch_take(h, base, need, st);
  const size_t lds = (size_t)5 * 3 * 16 * (H + 8) * 2;
#define ASR_GRU_CLB(HH)                                                                                             \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)gru_bwd_cluster_kernel<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(gru_bwd_cluster_kernel<HH>, dim3(cluster_grid(HH / 32, ncl)), dim3(256), lds, st, T, B, ndir, dout,   \
                       d_h_final, hout, r, u, c, wghT, wchT, seq_len, dgate, dcand, xa.area, (unsigned*)base,         \
                       kernel_flags(), xa.znext, xa.zwords);                                                         \
  } while (0)
  if (H == 64) ASR_GRU_CLB(64); else if (H == 128) ASR_GRU_CLB(128); else if (H == 256) ASR_GRU_CLB(256); else ASR_GRU_CLB(320);
#undef ASR_GRU_CLB
  return true;
}

#define ASR_F32_ARGS_B h, T, B, ndir, dhout, gates, cs, whpb, peep, seq_len, dcf, dhf, dgates, dpeep_part, st
bool asr_cluster_bwd_f32_try(asr_handle* h, int T, int B, int H, int ndir, const float* dhout,
                             const void* gates, const float* cs, const void* whpb, const float* peep,
                             const int32_t* seq_len, const float* dcf, const float* dhf, void* dgates,
                             float* dpeep_part, hipStream_t st) {
  if (!cluster_f32_enabled()) return false;
  if (bwd_units_per_cu() == 32) {
    if (H == 128) return cluster_bwd_f32_launch<128, 32>(ASR_F32_ARGS_B);
    if (!cluster_f32_wide_enabled()) return false;
    if (H == 256) return cluster_bwd_f32_launch<256, 32>(ASR_F32_ARGS_B);
    if (H == 320) return cluster_bwd_f32_launch<320, 32>(ASR_F32_ARGS_B);
    if (H == 512) return cluster_bwd_f32_launch<512, 32>(ASR_F32_ARGS_B);
    return false;
  }
  constexpr int HH = 128, G = HH / HS;
  if (H != HH) return false;
  const int ncl = (B / 16) * ndir;
  const size_t need = (size_t)ncl * (XHDR + (size_t)2 * G * G * 4 * 64 * 2) * sizeof(u64);
  if ((size_t)T * B * ndir * H >= (1ull << 31) || h->scratch_bytes < XCH_BYTES || need > XCH_HALF ||
      (int)cluster_grid(G, ncl) > h->num_cu)
    return false;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  const XchAreas xa = xch_take(h, base, need, st);
  const size_t lds = (size_t)2 * 16 * (4 * HS + 4) * 4 + 2 * 4 * 64 * 8;   // two dG images + the hand-over buffer
  hipLaunchKernelGGL(lstm_bwd_cluster8_f32_kernel<HH>, dim3(cluster_grid(G, ncl)), dim3(CT8), lds, st, T, B, ndir,
                     dhout, (const f32x4_t*)gates, cs, (const float*)whpb, peep, seq_len, dcf, dhf, (f32x4_t*)dgates,
                     dpeep_part, xa.area, (unsigned*)base, kernel_flags(), xa.znext, xa.zwords, h->bptt_clip);
  return true;
}
#undef ASR_F32_ARGS_B

// ---------------------------------------------------------------- debug: 16-byte exchange words, tear probe
// Would a lane's 16-byte store be seen whole by a 16-byte load of another CU?  The clusters' exchange uses 8-byte granules
// (one 64-bit store / load per lane: single-copy atomic by construction) or per-word tags; a 16-byte self-tagged word would
// halve the poll instructions but is only correct if the four dwords of one
// global_store_dwordx4 never appear torn.  Workgroup 0 writes {i, i, i, i} for i = 1 .. iters into one word per lane
// (plain stores when `wt` is 0, as a co-located cluster does, write-through sc1 stores otherwise); workgroup `peer`
// (8: same XCD as workgroup 0, 1: another XCD) polls the same words with L1-bypassing 16-byte loads and counts the loads
// whose four dwords differ.  out[3 lane .. + 2] = {loads, torn loads, distinct values seen}.
namespace {
__global__ __launch_bounds__(64) void tear_probe_kernel(unsigned* __restrict__ buf, unsigned iters, int peer, int wt,
                                                        unsigned long long* __restrict__ out) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const int lane = threadIdx.x;
  u32x4_t* word = reinterpret_cast<u32x4_t*>(buf) + lane;
  if (blockIdx.x == 0) {
    for (unsigned i = 1; i <= iters; ++i) {
      const u32x4_t v = {i, i, i, i};
      if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(word), "v"(v) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(word), "v"(v) : "memory");
      if ((i & 63u) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // bounded queue, stores in order
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if ((int)blockIdx.x != peer) return;
  unsigned long long loads = 0, torn = 0, seen = 0;
  unsigned last = 0;
  for (unsigned long long spin = 0; spin < (1ull << 24); ++spin) {          // bounded: ~15 s even if nothing ever arrives
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(word) : "memory");
    ++loads;
    if (!(v[0] == v[1] && v[1] == v[2] && v[2] == v[3])) ++torn;
    if (v[0] != last) { ++seen; last = v[0]; }
    if (__all(v[0] >= iters && v[1] >= iters && v[2] >= iters && v[3] >= iters)) break;
  }
  out[3 * lane + 0] = loads;
  out[3 * lane + 1] = torn;
  out[3 * lane + 2] = seen;
}
}  // namespace
extern "C" int asr_debug_tear_probe(asr_handle* h, unsigned* buf, unsigned iters, int peer, int write_through,
                                    unsigned long long* out, asr_stream s) {
  if (!h || !buf || !out || iters < 1 || (peer != 1 && peer != 8) || ((uintptr_t)buf) % 16 != 0) return ASR_ERR_INVALID_ARG;
  if (hipMemsetAsync(buf, 0, 64 * 16, (hipStream_t)s) != hipSuccess) return ASR_ERR_HIP;
  hipLaunchKernelGGL(tear_probe_kernel, dim3(9), dim3(64), 0, (hipStream_t)s, buf, iters, peer, write_through, out);
  ASR_CHECK_LAUNCH(h, "asr_debug_tear_probe");
  return ASR_OK;
}

extern "C" int asr_debug_set_lstm_flags(int flags) { g_dflags = flags; return 0; }

// Asynchronous read of the sticky error word: one 4-byte device->host copy on `st`, no synchronisation.
extern "C" int asr_peek_async_errors(asr_handle* h, unsigned* host_flags, asr_stream s) {
  hipStream_t st = (hipStream_t)s;
  if (!h || !host_flags) return ASR_ERR_INVALID_ARG;
  if (h->scratch_bytes < XCH_BYTES) { *host_flags = 0; return ASR_OK; }
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  if (hipMemcpyAsync(host_flags, base, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_peek_async_errors: copy failed");
  return ASR_OK;
}
// Clears the sticky error word (after it has been reported) -- and, since a launch whose hand-off timed out leaves its
// exchange area in a state no later launch was written for (members that gave up at different steps, a header of a cluster
// that never completed its placement handshake), puts BOTH exchange areas and their bookkeeping back to what a fresh handle
// has: everything zero, nothing owed.  64 MiB of memset on the error path only.
extern "C" int asr_clear_async_errors(asr_handle* h, asr_stream s) {
  hipStream_t st = (hipStream_t)s;
  if (!h) return ASR_ERR_INVALID_ARG;
  if (h->scratch_bytes < XCH_BYTES) return ASR_OK;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  if (hipMemsetAsync(base, 0, XCH_BYTES, st) != hipSuccess) ASR_FAIL(h, ASR_ERR_HIP, "asr_clear_async_errors");
  h->xch_dirty[0] = h->xch_dirty[1] = 0;
  h->xch_next = 0;
  return ASR_OK;
}
extern "C" int asr_debug_cluster_cycles(unsigned long long* out, int n) {
  if (!g_cdbg_host || n > 1280) return -1;
  return hipMemcpy(out, g_cdbg_host, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

// Synchronises the device and returns the sticky error word of the cluster kernels
// (bit 0: forward hand-off timed out, bit 1: backward); 0 = fine.
extern "C" int asr_check_async_errors(asr_handle* h, unsigned* flags_out) {
  if (!h) return ASR_ERR_INVALID_ARG;
  unsigned v = 0;
  char* base = (char*)h->scratch + (h->scratch_bytes - XCH_BYTES);
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(&v, base, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_check_async_errors: device error");
  if (flags_out) *flags_out = v;
  if (v & 3u) ASR_FAIL(h, ASR_ERR_HIP, "LSTM cluster hand-off timed out (flags 0x%x)", v);
  if (v) ASR_FAIL(h, ASR_ERR_HIP, "LSTM recurrence produced a non-finite hidden state (flags 0x%x)", v);
  return ASR_OK;
}
