// CTC prefix beam search on gfx950.
//
// Restates models/ctc/decoders/beam_search_decoder.py:53-152 (the reference's numpy decoder,
// which is also what pins this kernel: tests/golden/decoders_v1.npz) and stands in for
// tf.nn.ctc_beam_search_decoder at models/ctc/ctc.py:344-346.
//
// Reference semantics kept exactly:
//   * log-space (p_blank, p_non_blank) per prefix, float64;
//   * per frame, for every class c and every beam entry: blank -> stays; c != last -> extend
//     with (p_b + p_nb); c == last -> extend with p_b only AND the unchanged prefix collects
//     p_nb (the "merging case", :120-139);
//   * next_beam is a dict keyed by the prefix tuple: an extension that spells a prefix already
//     in the beam merges into it;
//   * sorted(..., key=logsumexp(p_b,p_nb), reverse=True)[:beam_width] is a STABLE sort, so ties
//     keep dict insertion order (vocab-major: for c: for entry: new_prefix, then prefix).
//
// Mapping: one workgroup per utterance (utterances are independent -> B workgroups; the T
// frames are serial).  Prefixes live in a per-utterance trie (parent, label) in the workspace;
// prefix identity is a 64-bit rolling hash + length (a dropped-and-recreated prefix gets a
// new trie node but the same hash, so dict merging stays correct).  Per frame:
//   1. lp = log_softmax(logits[t]) in fp64 (LDS),
//   2. W stay candidates (with the merge from the parent entry, if that is in the beam),
//   3. W x C extension totals to scratch (pairs that merged are masked out),
//   4. exact top-W by a byte-wise radix select over the 12-byte composite key
//      (orderable fp64 total, ~insertion index)  -> ties resolved as the reference's stable sort,
//   5. winners ranked (W^2 compares) into the next beam, new trie nodes appended.
// Exact class pruning (step 2b): an extension (entry j, class c) can only reach the top W if fewer
// than W classes c' != last_j have a strictly larger log-probability (each of those gives a distinct
// candidate of the same parent with a larger total), so only the classes at or above the (W+1)-th
// largest log-probability of the frame (minus a margin far above fp64 rounding, so "larger" survives
// the addition) are enumerated: W*(W+1) candidates instead of W*C (10 k vs 339 k for C = 3387, W = 100).
// Merged pairs are handled in the stay candidates and do not depend on the pruning; insertion
// indices still use the real class id, so the reference's tie order is unchanged.
// HBM/latency-bound integer+fp64 work; nothing here is GEMM-shaped.
//
// Round 4: the candidate keys of a frame live in LDS (orderable u64 of the fp64 total; 0 = excluded) whenever they fit --
// with the class pruning they are W (W + 2) ~ 10 k keys = 80 KB -- instead of fp64 totals in global memory re-read by every
// pass of the radix select; the digit of a radix pass is chosen by one wave with a suffix scan over the 256 bins instead of
// thread 0 walking them; the ordered compaction of the kept classes is one pass with a block-wide prefix sum instead of
// C / 256 rounds of two barriers each; and the two radix selects NARROW: a thread keeps its share of the keys in
// registers, a key that falls out of (or is decided by) a pass is dropped from the later ones, and the select stops at
// the first pass whose chosen bin holds exactly the number of keys still wanted (usually the third or fourth of twelve).
// Same candidates, same composite keys, same winners: the result is bit-identical (49 + cfg-E goldens of the reference's
// decoder, TensorFlow's testCTCDecoderBeamSearch).  Measured on MI355X (bench.py decode leg, us per frame of one
// utterance's workgroup): C = 3387, W = 100: 366 -> 69; C = 62, W = 20: 90 -> 17.  What a frame costs now, in cycles
// (ASR_BEAM_DBG=1): fp64 log-softmax 16 k, stay candidates 27 k, class select 22 k, compaction 8 k, keys 14 k, top-W
// select 52 k, ranking + trie 22 k.
//
// Round 6: 512 threads per utterance instead of 256 (template parameter NT; the phases that stride over classes or
// candidates scale, the register budget still holds 24 keys per thread); the parent lookup and the ranking of the
// winners run NT / 128 threads per entry, branch-free (as loops that break / short-circuit they waited for one LDS read
// per iteration: 13 k + 15 k cycles of a W = 100 frame); trie node ids are base + rank instead of one atomic per new
// prefix; the frame's logits are requested one frame ahead; both radix selects skip the leading bytes all keys share
// (one AND / OR reduction); a radix pass is ONE barrier (three rotating histograms, every wave walks the bins itself,
// on a DPP scan); the class select stops at a bin that holds a few keys more than wanted (any threshold at or below the
// exact one prunes correctly); the top-W select only narrows during the passes -- the winners are collected by one pass
// against the threshold the passes spell.  Same candidates up to classes kept in vain, same composite keys, same winners
// (goldens, and ASR_BEAM_THREADS=256 against 512 in tests/test_gpu_ops.py).  And the candidates themselves are pruned by
// (entry, class) PAIR, not only by class (step 2c in the kernel: entry j x class of rank r has (j + 1)(r - 1) candidates
// strictly above it): ~W (ln W + 2) keys per frame instead of W (W + 1), 0.7 k instead of 10 k at W = 100 -- the select's
// first counting pass had been 10 k LDS atomics on a handful of bins.  us per frame (scripts/probe_beam.py),
// C = 3387 / W = 100: 69 -> 26 (flat and peaked posteriors); C = 62 / W = 20: 16.7 -> 12.0 / 11.3.  Cycles of a
// C = 3387 / W = 100 frame now: log-softmax 8 k, stay candidates 8 k, class select 12 k, compaction 5 k, class ranks +
// keys 15 k, top-W select 7 k, ranking + trie 5 k.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BEAM_MAX = 128;
constexpr double DNEG = -INFINITY;

__device__ __forceinline__ double lse2d(double a, double b) {
  const double m = fmax(a, b);
  if (m == DNEG) return DNEG;
  return m + log(exp(a - m) + exp(b - m));
}
__device__ __forceinline__ double lse3d(double a, double b, double c) {
  const double m = fmax(fmax(a, b), c);
  if (m == DNEG) return DNEG;
  return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}
// order-preserving map double -> uint64 (ascending)
__device__ __forceinline__ unsigned long long okey(double x) {
  unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unokey(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ unsigned long long hmix(unsigned long long h, int c) {
  unsigned long long x = h * 0x9E3779B97F4A7C15ull + (unsigned long long)(c + 1);
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29;
  return x;
}

// Inclusive prefix sum over the 64 lanes of a wave on the DPP path (row shifts by 1 / 2 / 4 / 8 with zero fill, then lane 15
// of a row added to the next row, then lane 31 to the upper half): eight VALU operations instead of six LDS permutes.
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// One radix pass' decision, by one wave (all 64 lanes active): the reference walk
//     for (d = 255; d > 0; --d) { if (hist[d] >= rem) break; rem -= hist[d]; }
// as a scan -- lane l owns bins 4 q .. 4 q + 3 of q = 63 - l (the HIGHEST bins in lane 0, so that the inclusive prefix over
// the lanes is the suffix sum over the bins; bin 0 never counts: the walk stops at d = 1 and falls through to 0).
// Also returns the population of the chosen bin.
__device__ __forceinline__ void pick_digit(const unsigned* hist, unsigned rem0, int lane, unsigned& digit, unsigned& rem,
                                           unsigned& pop) {
  const int q = 63 - lane;
  const unsigned h0 = hist[4 * q], h1 = hist[4 * q + 1], h2 = hist[4 * q + 2], h3 = hist[4 * q + 3];
  const unsigned own = h1 + h2 + h3 + (q ? h0 : 0u);
  const unsigned suf = wave_incl_scan_u32(own);            // bins 4 q and above
  const unsigned above = suf - own;
  const bool here = above < rem0 && (q == 0 || rem0 <= suf);
  unsigned d = 0, r = 0, p = 0;
  if (here) {
    unsigned acc = above;
    if (acc + h3 >= rem0) { d = 4 * q + 3; r = rem0 - acc; p = h3; }
    else {
      acc += h3;
      if (acc + h2 >= rem0) { d = 4 * q + 2; r = rem0 - acc; p = h2; }
      else {
        acc += h2;
        if (acc + h1 >= rem0) { d = 4 * q + 1; r = rem0 - acc; p = h1; }
        else { acc += h1; d = 4 * q; r = rem0 - acc; p = h0; }   // q = 0: digit 0, the walk's fall-through
      }
    }
  }
  const int src = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)__ballot(here)) - 1);
  digit = (unsigned)__builtin_amdgcn_readlane((int)d, src);
  rem = (unsigned)__builtin_amdgcn_readlane((int)r, src);
  pop = (unsigned)__builtin_amdgcn_readlane((int)p, src);
}

// The keys a thread holds in registers (0 = dead) towards the histogram of byte `byte` (7 .. 0 of the 64-bit key; all
// arithmetic on the 32-bit half the byte lives in).  The high-order bytes are mostly the same for every key (sign and
// exponent of values a few units apart): a thread whose keys agree adds its count with ONE LDS atomic.
template <int NK>
__device__ __forceinline__ void hist_regs(unsigned* hist, const unsigned long long (&rk)[NK], int byte) {
  const bool upper = byte >= 4;
  const int ds = (byte & 3) * 8;
  unsigned d0 = 0;
  int n = 0;
  bool uni = true;
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const unsigned long long k = rk[i];
    const unsigned d = ((upper ? (unsigned)(k >> 32) : (unsigned)k) >> ds) & 0xffu;
    if (k != 0ull) {
      if (n == 0) d0 = d;
      uni = uni && d == d0;
      ++n;
    }
  }
  if (n == 0) return;
  if (uni) { atomicAdd(&hist[d0], (unsigned)n); return; }
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const unsigned long long k = rk[i];
    if (k != 0ull) atomicAdd(&hist[((upper ? (unsigned)(k >> 32) : (unsigned)k) >> ds) & 0xffu], 1u);
  }
}

struct Entry {            // one beam entry
  double pb, pnb;
  unsigned long long hash, phash;   // hash of the prefix / of its parent prefix
  int node, len, last;
};

// byte `byte` (11 = most significant) of the 12-byte composite (key64 : ~idx32)
__device__ __forceinline__ unsigned digit_of(unsigned long long key, unsigned nidx, int byte) {
  return byte >= 4 ? (unsigned)((key >> ((byte - 4) * 8)) & 0xff) : ((nidx >> (byte * 8)) & 0xff);
}

template <int NT>
__global__ __launch_bounds__(NT) void ctc_beam_kernel(
    const float* __restrict__ logits, int T, int B, int C, const int32_t* __restrict__ seq_len,
    int blank, int W, double* __restrict__ tot_ws, int2* __restrict__ node_ws,
    int32_t* __restrict__ out_labels, int32_t* __restrict__ out_len, double* __restrict__ out_score, int lds_keys,
    unsigned long long* __restrict__ dbg) {
  // phase timers (ASR_BEAM_DBG=1): cycles of utterance 0's workgroup per phase, summed over the frames
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define BEAM_T(k) do { if (dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[k] += now_ - tprev; tprev = now_; } } while (0)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lp = reinterpret_cast<double*>(smem);      // [C]
  int* kc = reinterpret_cast<int*>(lp + C);          // [C] kept classes of the frame, ascending
  int* kpos = kc + C;                                // [C] position of a class in kc, -1 if pruned
  unsigned long long* lkeys = reinterpret_cast<unsigned long long*>(smem + (((size_t)C * 16 + 15) & ~(size_t)15));   // [lds_keys]
  __shared__ unsigned tcnt[NT / 64];
  __shared__ unsigned long long s_min[NT / 64], s_or[NT / 64];
  __shared__ double s_L[BEAM_MAX];                   // logsumexp(p_b, p_nb) of each beam entry
  __shared__ unsigned wcnt[NT / 64];
  __shared__ int s_K;
  __shared__ double s_thr;
  __shared__ Entry beam[BEAM_MAX];
  __shared__ Entry nbeam[BEAM_MAX];
  __shared__ double s_pb[BEAM_MAX], s_pnb[BEAM_MAX];  // stay candidates
  __shared__ unsigned long long s_key[BEAM_MAX];
  __shared__ unsigned s_idx[BEAM_MAX];
  __shared__ int s_parent[BEAM_MAX];
  __shared__ unsigned hist[256];                     // select_mem only
  __shared__ unsigned hist3[3][256];                 // the register selects' histograms, rotating (radix_pass)
  __shared__ unsigned long long s_mn2[NT / 64];
  __shared__ unsigned long long w_key[BEAM_MAX];
  __shared__ unsigned w_nidx[BEAM_MAX];
  __shared__ int w_src[BEAM_MAX];                    // candidate id of each winner
  __shared__ double red[NT / 64];
  __shared__ unsigned long long sel_key;              // threshold prefix being built
  __shared__ unsigned sel_nidx;
  __shared__ unsigned sel_remaining;
  __shared__ int s_nw, s_nb, s_nodes;
  __shared__ int s_rank[BEAM_MAX];
  __shared__ int cJ[NT], coff[NT + 1];               // entries enumerated per kept class / their first candidate (pruned frames)
  __shared__ int s_mext;

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Tb = min(max(seq_len[b], 0), T);
  unsigned long long* gkeys = reinterpret_cast<unsigned long long*>(tot_ws) + (size_t)b * ((size_t)W * C + W);
  int2* nodes = node_ws + (size_t)b * ((size_t)T * W + 1);

  for (int i = tid; i < 3 * 256; i += NT) hist3[0][i] = 0;
  int hrot = 0;                                            // the histogram the next radix pass counts into (block-uniform)
  if (tid == 0) {
    beam[0].pb = 0.0; beam[0].pnb = DNEG; beam[0].hash = 0x1234567ull; beam[0].phash = 0;
    beam[0].node = 0; beam[0].len = 0; beam[0].last = -1;
    nodes[0] = make_int2(-1, -1);
    s_nb = 1; s_nodes = 1;
  }
  __syncthreads();

  // a thread's share of a frame's logits (classes tid, tid + NT, ...; C <= 6144: the launcher's LDS limit), requested one
  // frame ahead: the row of frame t + 1 travels while frame t is searched
  constexpr int RP = (6144 + NT - 1) / NT;
  float pre[RP];
  auto fetch_row = [&](int t) {
    const float* row = logits + ((size_t)t * B + b) * C;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const int c = i * NT + tid;
      pre[i] = c < C ? row[c] : -INFINITY;
    }
  };
  if (Tb > 0) fetch_row(0);

  for (int t = 0; t < Tb; ++t) {
    const int nb = s_nb;
    if (dbg) tprev = __builtin_amdgcn_s_memtime();
    // ---- 1. fp64 log-softmax of the frame
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < RP; ++i) {                         // (a thread revisits its own lp[c])
      const int c = i * NT + tid;
      if (c < C) {
        lp[c] = (double)pre[i];
        m = fmaxf(m, pre[i]);
      }
    }
    if (t + 1 < Tb) fetch_row(t + 1);
    m = wave_reduce_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    double mm = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) mm = fmax(mm, red[w]);
    __syncthreads();
    double ssum = 0.0;
    for (int c = tid; c < C; c += NT) ssum += exp(lp[c] - mm);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
    if (lane == 0) red[wave] = ssum;
    __syncthreads();
    double zs = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) zs += red[w];
    const double z = mm + log(zs);
    for (int c = tid; c < C; c += NT) lp[c] = lp[c] - z;
    BEAM_T(0);
    // ---- 2. parent lookup + stay candidates
    // the first (lowest) entry that spells this entry's parent prefix: NT / 128 threads per entry walk the beam in strides,
    // branch-free (a loop that breaks at the hit waits for one LDS read per iteration: 13 k cycles of a W = 100 frame)
    if (tid < BEAM_MAX) s_parent[tid] = 0x7fffffff;
    __syncthreads();
    {
      constexpr int PARTS = NT / BEAM_MAX;
      const int e = tid & (BEAM_MAX - 1), part = tid / BEAM_MAX;
      if (e < nb) {
        const unsigned long long eph = beam[e].phash;
        const int elen = beam[e].len;                        // (len 0, the empty prefix: no entry has len + 1 == 0)
        int cand = 0x7fffffff;
#pragma unroll 4
        for (int j = part; j < nb; j += PARTS) {
          const bool hit = (beam[j].hash == eph) & (beam[j].len + 1 == elen);
          cand = hit ? min(cand, j) : cand;
        }
        if (cand != 0x7fffffff) atomicMin(&s_parent[e], cand);
      }
    }
    __syncthreads();
    if (tid < nb && s_parent[tid] == 0x7fffffff) s_parent[tid] = -1;   // (every later reader of s_parent[i] is thread i)
    if (tid < nb) {
      const Entry e = beam[tid];
      const double lpb = lp[blank];
      double pb = lse3d(DNEG, e.pb + lpb, e.pnb + lpb);
      double pnb = DNEG;
      unsigned first = (unsigned)blank * (2u * W) + 2u * tid;       // touched when c == blank
      const int par = s_parent[tid];
      if (e.len > 0) {
        const double lpl = lp[e.last];
        // insertion order inside class c = last: parent's extension (2*par) comes at its own
        // position, the entry's own merging-case touch at 2*tid+1
        double from_parent = DNEG;
        if (par >= 0) {
          const Entry p = beam[par];
          from_parent = (e.last == p.last) ? p.pb + lpl : lse2d(p.pb, p.pnb) + lpl;
        }
        const double own = e.pnb + lpl;
        if (par >= 0 && par < tid) pnb = lse2d(lse2d(DNEG, from_parent), own);   // dict order
        else if (par >= 0) pnb = lse2d(lse2d(DNEG, own), from_parent);
        else pnb = own;
        if (e.last != blank) {
          unsigned a = (unsigned)e.last * (2u * W) + 2u * tid + 1u;
          if (par >= 0) a = min(a, (unsigned)e.last * (2u * W) + 2u * par);
          first = min(first, a);
        }
      }
      s_pb[tid] = pb; s_pnb[tid] = pnb;
      s_key[tid] = okey(lse2d(pb, pnb));
      s_idx[tid] = ~first;
    }
    __syncthreads();
    BEAM_T(1);
    // ---- 2b. classes that can still reach the top W (see the header): threshold = (W+1)-th largest
    //      non-blank log-probability, found with a byte-wise radix select over the frame's C values
    if (tid < nb) s_L[tid] = lse2d(beam[tid].pb, beam[tid].pnb);
    // Bytes 7 .. first + 1 are the same in every live key a thread holds in rk (first = -1: all keys are equal); `common`
    // = those bytes.  Log-probabilities a few units apart share sign, exponent and often the top of the significand: two
    // or three of the eight radix passes would find every key in one bin.  One barrier.
    auto lead_byte = [&](const auto& rk, unsigned long long& common) __attribute__((always_inline)) -> int {
      unsigned long long ka = ~0ull, ko = 0ull;
#pragma unroll
      for (int i = 0; i < (int)(sizeof(rk) / sizeof(rk[0])); ++i)
        if (rk[i] != 0ull) { ka &= rk[i]; ko |= rk[i]; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ka &= __shfl_xor(ka, o, 64);
        ko |= __shfl_xor(ko, o, 64);
      }
      if (lane == 0) { s_min[wave] = ka; s_or[wave] = ko; }
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) { ka &= s_min[w]; ko |= s_or[w]; }
      const unsigned long long diff = ka ^ ko;
      if (ko == 0ull) { common = 0ull; return -1; }          // no live key at all
      const int first = diff ? 7 - (__clzll((long long)diff) >> 3) : -1;
      common = first >= 7 ? 0ull : (first < 0 ? ka : (ka & ~((1ull << ((first + 1) * 8)) - 1ull)));
      return first;
    };
    // One pass of a register-resident radix select, ONE barrier: count byte `byte` of the live keys into the current
    // histogram (zero since the pass before last) while the next one is cleared, then every wave walks the bins itself
    // (the same arithmetic on the same counts: block-uniform results in registers, nothing to publish).  rem: keys still
    // wanted, counted down by the bins above the chosen one; exact: the chosen bin holds exactly rem keys.
    // Why three histograms: the readers of the one cleared here finished before the barrier of the previous pass.
    auto radix_pass = [&](const auto& rk, int byte, unsigned& rem, unsigned& dgt, bool& exact, unsigned* pop_out = nullptr)
        __attribute__((always_inline)) {
      unsigned* hc = hist3[hrot];
      hrot = hrot == 2 ? 0 : hrot + 1;
      if (tid < 256) hist3[hrot][tid] = 0;
      hist_regs(hc, rk, byte);
      __syncthreads();
      unsigned r2, pop;
      pick_digit(hc, rem, lane, dgt, r2, pop);
      exact = pop == r2;
      rem = r2;
      if (pop_out) *pop_out = pop;
    };
    const int R = W + 1;
    if (C - 1 > 4 * R) {   // small vocabularies: the select + compaction passes cost more than they prune
      // R-th largest non-blank log-probability by a NARROWING radix select: a thread holds its classes' keys in registers;
      // after a pass the keys outside the chosen bin are dropped (the ones above it are simply counted off), and the first
      // pass whose bin holds exactly the keys still wanted ends the search -- the answer is that bin's smallest key
      auto class_select = [&](auto NKC) __attribute__((always_inline)) {
        constexpr int NK = decltype(NKC)::value;
        unsigned long long rc[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          const int c = i * NT + tid;
          rc[i] = (c < C && c != blank) ? okey(lp[c]) : 0ull;   // (okey of a log-probability is never 0)
        }
        unsigned long long common_c = 0;
        const int first_c = lead_byte(rc, common_c);         // the passes start at the first byte the keys differ in
        // The threshold only has to be AT OR BELOW the R-th largest value (a class kept in vain costs W more candidates, a
        // class dropped in error would cost the result): the passes stop as soon as the bin the R-th largest falls into
        // holds at most W / 8 keys more than are still wanted -- the smallest key of that bin is the threshold
        // (... as long as the kept classes still fit the pruned enumeration below, K <= NT, or else the dense one the LDS)
        const int fit = lds_keys / nb - 1 - R;
        const unsigned slack = (unsigned)((R + (W >> 3) <= NT) ? (W >> 3) : max(0, min(W >> 3, fit)));
        unsigned rem = (unsigned)R;
        for (int byte = first_c; byte >= 0; --byte) {
          unsigned dgt, pop;
          bool ex;
          radix_pass(rc, byte, rem, dgt, ex, &pop);
          const bool upper = byte >= 4;
          const int ds = (byte & 3) * 8;
#pragma unroll
          for (int i = 0; i < NK; ++i) {
            const unsigned long long k = rc[i];
            const unsigned d = ((upper ? (unsigned)(k >> 32) : (unsigned)k) >> ds) & 0xffu;
            if (d != dgt) rc[i] = 0ull;
          }
          if (pop - rem <= slack) break;                      // block-uniform (pop >= rem; equal: the pass ended exactly)
        }
        // the threshold = the smallest key still alive (the R-th largest itself when a pass ended exactly)
        unsigned long long mn = ~0ull;
#pragma unroll
        for (int i = 0; i < NK; ++i)
          if (rc[i] != 0ull && rc[i] < mn) mn = rc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long v = __shfl_xor(mn, o, 64);
          if (v < mn) mn = v;
        }
        if (lane == 0) s_mn2[wave] = mn;
        __syncthreads();
        if (tid == 0) {
          unsigned long long m4 = s_mn2[0];
          for (int w = 1; w < NT / 64; ++w) if (s_mn2[w] < m4) m4 = s_mn2[w];
          const double th = unokey(m4);
          s_thr = th - 1e-9 * (1.0 + fabs(th));
        }
      };
      if (NT > 256 && C <= 2 * NT) class_select(std::integral_constant<int, 2>{});
      else if (NT > 256 && C <= 4 * NT) class_select(std::integral_constant<int, 4>{});
      else if (C <= 8 * NT) class_select(std::integral_constant<int, 8>{});
      else class_select(std::integral_constant<int, NT == 256 ? 24 : 8>{});   // (NT = 256) C <= 6144: the LDS limit of the launcher
    } else if (tid == 0) {
      s_thr = DNEG;
    }
    __syncthreads();
    BEAM_T(2);
    {   // ordered compaction of the kept classes: thread t owns the contiguous classes [t cpt, (t + 1) cpt)
      const double thr = s_thr;
      const int cpt = (C + NT - 1) / NT;
      const int c_lo = min(C, tid * cpt), c_hi = min(C, c_lo + cpt);
      int mine = 0;
      for (int c = c_lo; c < c_hi; ++c) mine += (c != blank && lp[c] >= thr) ? 1 : 0;
      int incl = mine;                                      // inclusive prefix over the wave's lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wcnt[wave] = (unsigned)incl;
      __syncthreads();
      int pos = incl - mine;
      for (int w = 0; w < wave; ++w) pos += (int)wcnt[w];
      for (int c = c_lo; c < c_hi; ++c) {
        const bool keep = c != blank && lp[c] >= thr;
        kpos[c] = keep ? pos : -1;
        if (keep) kc[pos++] = c;
      }
      if (tid == 0) {
        unsigned kk = 0;
        for (int w = 0; w < NT / 64; ++w) kk += wcnt[w];
        s_K = (int)kk;
      }
    }
    __syncthreads();
    const int K = s_K;
    BEAM_T(3);
    // ---- 2c. which (entry, class) pairs can still reach the top W.  The beam is sorted by L = logsumexp(p_b, p_nb)
    //      (the rank order of the previous frame's select), so the extension of entry j by class c has at least
    //      (j + 1) (r_c - 1) candidates STRICTLY above it, r_c = the kept classes whose log-probability exceeds lp_c by a
    //      margin far above fp64 rounding: entries j' <= j (L_j' >= L_j) x those classes, less the one class per entry that
    //      repeats its last label (its total uses p_b alone); a pair that merged into a beam prefix counts through that
    //      prefix's stay candidate, whose total is at least the pair's.  With W or more above it a candidate cannot be among
    //      the top W whatever the tie order, so class c is enumerated for the entries j < J_c only, J_c = nb for r_c <= 1 and
    //      min(nb, (W - 1) / (r_c - 1)) otherwise: ~W (ln W + 2) candidates instead of W (W + 1) -- 0.7 k instead of 10 k at
    //      W = 100.  (j = 0 gives back the class pruning above: r_c <= W.)  Frames that keep more classes than the
    //      workgroup has threads (ties at the threshold) enumerate every pair: J_c = nb.
    // (frames with at most four candidates per thread are not worth the three barriers of the bookkeeping)
    const bool pruned = K > 0 && K <= NT && K <= lds_keys && nb * K > 4 * NT;
    if (pruned) {
      double* klp = reinterpret_cast<double*>(lkeys);      // (the key area is free until the keys are written)
      if (tid < K) klp[tid] = lp[kc[tid]];
      __syncthreads();
      int mine = 0;
      if (tid < K) {
        const double x = klp[tid];
        const double xm = x + 1e-9 * (1.0 + fabs(x));
        int r = 0;
#pragma unroll 8
        for (int o = 0; o < K; ++o) r += klp[o] > xm ? 1 : 0;
        mine = r <= 1 ? nb : min(nb, (W - 1) / (r - 1));
        cJ[tid] = mine;
      }
      const int incl = (int)wave_incl_scan_u32((unsigned)mine);
      if (lane == 63) wcnt[wave] = (unsigned)incl;
      __syncthreads();                                      // (also: every klp read is done before the keys land there)
      int pos = incl - mine;
      for (int w = 0; w < wave; ++w) pos += (int)wcnt[w];
      if (tid < K) coff[tid] = pos;
      if (tid == K - 1) { coff[K] = pos + mine; s_mext = pos + mine; }
      __syncthreads();
    }
    // ---- 3. candidate keys: [0, nb) the stay candidates, then class-major the extensions: nb + base(ci) + j = entry j
    //      extended by kept class ci (p_b = -inf, so total = p_nb contribution); 0 = excluded.  In LDS when they fit, else
    //      in the workspace.
    const int M = nb + (pruned ? s_mext : nb * K);
    auto cand_base = [&](int ci) -> int { return pruned ? coff[ci] : ci * nb; };
    auto cand_cnt = [&](int ci) -> int { return pruned ? cJ[ci] : nb; };
    auto cand_decode = [&](int e, int& ci, int& j) {
      if (!pruned) { ci = e / nb; j = e - ci * nb; return; }
      int lo = 0, hi = K;                                    // the last class whose first candidate is at or before e
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (coff[mid] <= e) lo = mid; else hi = mid;
      }
      ci = lo; j = e - coff[lo];
    };
    // tie-break index of candidate id (needed for the winners, and where keys tie at the threshold)
    auto nidx_of = [&](int id) -> unsigned {
      if (id < nb) return s_idx[id];
      int ci, j;
      cand_decode(id - nb, ci, j);
      return ~((unsigned)kc[ci] * (2u * W) + 2u * j);
    };
    // (the tie index of a winner is computed by its own thread in the ranking step -- one division per winner in
    // parallel instead of one inside every divergent push)
    auto push_winner = [&](unsigned long long k, int id) {
      const int pos = atomicAdd(&s_nw, 1);
      if (pos < BEAM_MAX) { w_key[pos] = k; w_src[pos] = id; }
    };
    auto frame_tail = [&](unsigned long long* ck) __attribute__((always_inline)) {
      if (tid < nb) ck[tid] = s_key[tid];
      if (!pruned) {   // every pair: candidate e = ci nb + j, walked in strides of NT (K == 0: one class, or every
                       // non-blank probability NaN -- the loop is empty)
        int ci = tid / nb, j = tid - ci * nb;
        const int dci = NT / nb, dj = NT - dci * nb;
        for (int e = tid; e < nb * K; e += NT) {
          const int c = kc[ci];
          ck[nb + e] = okey(((c == beam[j].last) ? beam[j].pb : s_L[j]) + lp[c]);
          ci += dci; j += dj;
          if (j >= nb) { j -= nb; ++ci; }
        }
      } else {   // NT / 128 classes at a time, one thread per entry
        const int j = tid & (BEAM_MAX - 1);
        const int jlast = j < nb ? beam[j].last : -2;
        const double jpb = j < nb ? beam[j].pb : 0.0, jL = j < nb ? s_L[j] : 0.0;
#pragma unroll 2
        for (int ci = tid / BEAM_MAX; ci < K; ci += NT / BEAM_MAX) {
          const int c = kc[ci], cnt = cand_cnt(ci), base = cand_base(ci);
          const double lpc = lp[c];
          if (j < cnt) ck[nb + base + j] = okey((c == jlast ? jpb : jL) + lpc);
        }
      }
      __syncthreads();
      if (tid < nb && s_parent[tid] >= 0 && beam[tid].last != blank) {   // merged pairs are not new prefixes
        const int kp = kpos[beam[tid].last];
        if (kp >= 0 && s_parent[tid] < cand_cnt(kp)) ck[nb + cand_base(kp) + s_parent[tid]] = 0ull;
      }
      if (tid == 0) { s_nw = 0; sel_key = 0; sel_nidx = 0; }
      __syncthreads();
      BEAM_T(4);
      // ---- 4. exact top-W over the valid candidates, ordered by (key, tie index): a NARROWING radix select on the keys a
      //      thread holds in registers.  After each pass a key above the chosen bin is a winner (pushed at once), one
      //      below it is out, one inside it stays; when the bin holds exactly the keys still wanted they all win and
      //      the select is over.  Keys that survive all eight bytes are equal: the tie index decides, four more bytes.
      // frames with more candidates than a thread can hold (ties at the pruning threshold, flat posteriors): the plain
      // twelve-pass select over the key array
      auto select_mem = [&](const unsigned long long* ckm) -> int {
        int cnt = 0;
        for (int id = tid; id < M; id += NT) cnt += ckm[id] != 0ull ? 1 : 0;
        cnt = (int)wave_reduce_sum((float)cnt);              // <= 64 * ceil(M / NT): exact in fp32
        if (lane == 0) tcnt[wave] = (unsigned)cnt;
        __syncthreads();
        int valid = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) valid += (int)tcnt[w];
        const int want = min(W, valid);
        if (tid == 0) sel_remaining = want;
        __syncthreads();
        if (valid > want) {
          for (int byte = 11; byte >= 0; --byte) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned long long pk = sel_key;
            const unsigned pn = sel_nidx;
            if (byte >= 4) {
              const int sh = (byte - 4 + 1) * 8, ds = (byte - 4) * 8;
              for (int id = tid; id < M; id += NT) {
                const unsigned long long k = ckm[id];
                if (k != 0ull && (sh >= 64 || (k >> sh) == (pk >> sh))) atomicAdd(&hist[(unsigned)((k >> ds) & 0xff)], 1u);
              }
            } else {
              const int sh = (byte + 1) * 8;
              for (int id = tid; id < M; id += NT) {
                if (ckm[id] != pk) continue;                 // (pk != 0: the threshold is a valid candidate's key)
                const unsigned n = nidx_of(id);
                if (sh >= 32 || (n >> sh) == (pn >> sh)) atomicAdd(&hist[(n >> (byte * 8)) & 0xff], 1u);
              }
            }
            __syncthreads();
            if (tid < 64) {
              unsigned d, rem, pop;
              pick_digit(hist, sel_remaining, lane, d, rem, pop);
              if (tid == 0) {
                sel_remaining = rem;
                if (byte >= 4) sel_key |= ((unsigned long long)d) << ((byte - 4) * 8);
                else sel_nidx |= d << (byte * 8);
              }
            }
            __syncthreads();
          }
        }
        const unsigned long long tk = sel_key;
        const unsigned tn = sel_nidx;
        for (int id = tid; id < M; id += NT) {    // winners: composite >= threshold (all valid ones when valid <= W)
          const unsigned long long k = ckm[id];
          if (k == 0ull) continue;
          bool win = (valid <= want) || (k > tk);
          if (!win && k == tk) win = nidx_of(id) >= tn;
          if (win) push_winner(k, id);
        }
        return want;
      };
      auto select = [&](auto NKC) __attribute__((always_inline)) {
        constexpr int NK = decltype(NKC)::value;
        unsigned long long rk[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          const int id = i * NT + tid;
          rk[i] = id < M ? ck[id] : 0ull;
        }
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NK; ++i) cnt += rk[i] != 0ull ? 1 : 0;
        cnt = (int)wave_reduce_sum((float)cnt);              // <= 64 NK: exact in fp32
        if (lane == 0) tcnt[wave] = (unsigned)cnt;
        unsigned long long common = 0;
        const int first = lead_byte(rk, common);             // (its barrier also publishes tcnt)
        int valid = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) valid += (int)tcnt[w];
        const int want = min(W, valid);
        if (valid <= want) {                                  // everything wins
#pragma unroll
          for (int i = 0; i < NK; ++i)
            if (rk[i] != 0ull) push_winner(rk[i], i * NT + tid);
          return want;
        }
        unsigned rem = (unsigned)want;
        // The passes only NARROW (branch-free: a key outside the chosen bin leaves the registers, the ones above the bin
        // are counted off by pick_digit); `thr` collects the chosen digits under the bytes every key shares.  When a bin
        // holds exactly the keys still wanted the search is over: the winners are the keys >= thr.  Keys that survive all
        // eight bytes are equal (= thr): everything above wins, and the tie index decides among them, four more bytes.
        unsigned long long thr = common;
        bool key_exact = false;
        BEAM_T(5);
        for (int byte = first; byte >= 0; --byte) {
          unsigned dgt;
          radix_pass(rk, byte, rem, dgt, key_exact);
          thr |= (unsigned long long)dgt << (byte * 8);
          const bool upper = byte >= 4;
          const int ds = (byte & 3) * 8;
#pragma unroll
          for (int i = 0; i < NK; ++i) {
            const unsigned long long k = rk[i];
            const unsigned d = ((upper ? (unsigned)(k >> 32) : (unsigned)k) >> ds) & 0xffu;
            rk[i] = (d == dgt) ? k : 0ull;                    // (a dead slot stays dead)
          }
          if (key_exact) break;                               // block-uniform
        }
        if (!key_exact) {
          // the survivors share one key; sel_remaining of them win, those with the largest tie index (= the earliest in
          // the reference's insertion order).  The same select on the 32-bit index, kept in the register's low word
          // (bit 32 marks the slot as alive: an index can be 0)
#pragma unroll
          for (int i = 0; i < NK; ++i)
            if (rk[i] != 0ull) rk[i] = (1ull << 32) | nidx_of(i * NT + tid);
          bool exact = false;
          for (int byte = 3; byte >= 0; --byte) {
            unsigned dgt;
            radix_pass(rk, byte, rem, dgt, exact);
#pragma unroll
            for (int i = 0; i < NK; ++i) {
              const unsigned long long k = rk[i];
              if (k == 0ull) continue;
              const unsigned n = (unsigned)k, d = (n >> (byte * 8)) & 0xffu;
              if (d > dgt || (exact && d == dgt)) {
                push_winner(thr, i * NT + tid);
                rk[i] = 0ull;
              } else if (d < dgt) {
                rk[i] = 0ull;
              }
            }
            if (exact) break;
          }
          // (tie indices are distinct, so the last byte's bin holds one key: the select always ends exactly)
        }
        // the winners decided by their keys alone: one pass over the frame's keys
        #pragma unroll
        for (int i = 0; i < NK; ++i) {
          const int id = i * NT + tid;
          const unsigned long long k = id < M ? ck[id] : 0ull;
          if (k != 0ull && (key_exact ? k >= thr : k > thr)) push_winner(k, id);
        }
        return want;
      };
      int want;
      // keys per thread: what M needs, from a ladder that fits the register budget of NT / 64 waves (512 / 256 VGPRs)
      if (NT > 256 && M <= 2 * NT) want = select(std::integral_constant<int, 2>{});
      else if (M <= 4 * NT) want = select(std::integral_constant<int, 4>{});
      else if (M <= 8 * NT) want = select(std::integral_constant<int, 8>{});
      else if (M <= 16 * NT) want = select(std::integral_constant<int, 16>{});
      else if (NT == 512 && M <= 24 * NT) want = select(std::integral_constant<int, NT == 512 ? 24 : 16>{});
      else if (NT == 256 && M <= 40 * NT) want = select(std::integral_constant<int, NT == 256 ? 40 : 16>{});
      else if (NT == 256 && M <= 48 * NT) want = select(std::integral_constant<int, NT == 256 ? 48 : 16>{});
      else want = select_mem(ck);
      __syncthreads();
      BEAM_T(6);
      const int nw = min(s_nw, want);
      // ---- 5. rank winners (composite keys are distinct) and build the next beam
      if (tid < nw) w_nidx[tid] = nidx_of(w_src[tid]);
      if (tid < BEAM_MAX) s_rank[tid] = 0;
      __syncthreads();
      {   // rank of winner i = winners above it: NT / 128 threads per winner, branch-free partial counts
        constexpr int PARTS = NT / BEAM_MAX;
        const int i = tid & (BEAM_MAX - 1), part = tid / BEAM_MAX;
        if (i < nw) {
          const unsigned long long k = w_key[i];
          const unsigned n = w_nidx[i];
          int r = 0;
#pragma unroll 4
          for (int o = part; o < nw; o += PARTS) {
            const unsigned long long ko = w_key[o];
            const unsigned no = w_nidx[o];
            r += (int)((ko > k) | ((ko == k) & (no > n)));
          }
          if (r) atomicAdd(&s_rank[i], r);
        }
      }
      __syncthreads();
      if (tid < nw) {
        const unsigned long long k = w_key[tid];
        const int rank = s_rank[tid];
        const int id = w_src[tid];
        Entry ne;
        if (id < nb) {
          ne = beam[id];
          ne.pb = s_pb[id]; ne.pnb = s_pnb[id];
        } else {
          int ci, j;
          cand_decode(id - nb, ci, j);
          const int c = kc[ci];
          const Entry p = beam[j];
          ne.pb = DNEG; ne.pnb = unokey(k);                  // the key IS the total (order-preserving bijection)
          ne.phash = p.hash; ne.hash = hmix(p.hash, c);
          ne.len = p.len + 1; ne.last = c;
          const int node = s_nodes + rank;                   // ids of a frame: s_nodes .. s_nodes + nw - 1 (stay winners leave gaps)
          nodes[node] = make_int2(p.node, c);
          ne.node = node;
        }
        nbeam[rank] = ne;
      }
      BEAM_T(7);
      return nw;
    };
    const int nw = (M <= lds_keys) ? frame_tail(lkeys) : frame_tail(gkeys);
    __syncthreads();
    if (tid < nw) beam[tid] = nbeam[tid];
    if (tid == 0) { s_nb = nw; s_nodes += nw; }
    __syncthreads();
  }

  if (dbg && b == 0 && tid == 0)
    for (int k = 0; k < 8; ++k) dbg[k] = ph[k];
#undef BEAM_T
  // best hypothesis = beam[0]; walk the trie back
  if (tid == 0) {
    const Entry e = beam[0];
    const int n = e.len;
    int node = e.node;
    for (int i = n - 1; i >= 0; --i) {
      const int2 nd = nodes[node];
      out_labels[(size_t)b * T + i] = nd.y;
      node = nd.x;
    }
    for (int i = n; i < T; ++i) out_labels[(size_t)b * T + i] = -1;
    out_len[b] = n;
    out_score[b] = -lse2d(e.pb, e.pnb);
  }
}

struct BeamWs { size_t tot, nodes, total; };
inline BeamWs beam_ws_layout(int T, int B, int C, int W) {
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  BeamWs w;
  size_t o = 0;
  w.tot = o;   o += al((size_t)B * ((size_t)W * C + W) * sizeof(double));   // candidate keys when they do not fit in LDS
  w.nodes = o; o += al((size_t)B * ((size_t)T * W + 1) * sizeof(int2));
  w.total = o;
  return w;
}

}  // namespace

extern "C" size_t asr_ctc_beam_workspace_bytes(int T, int B, int C, int beam_width) {
  if (T < 0 || B < 0 || C < 1 || beam_width < 1) return 0;
  return beam_ws_layout(T, B, C, beam_width).total;
}

extern "C" int asr_ctc_beam_decode(asr_handle* h, const float* logits, int T, int B, int C,
                                   const int32_t* seq_len, int blank, int beam_width,
                                   int32_t* out_labels, int32_t* out_len, double* out_score,
                                   void* workspace, size_t workspace_bytes, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!logits || !seq_len || !out_labels || !out_len || !out_score || T <= 0 || B <= 0 || C < 2 ||
      blank < 0 || blank >= C || beam_width < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_ctc_beam_decode: bad args T=%d B=%d C=%d beam=%d", T, B, C, beam_width);
  if (beam_width > BEAM_MAX)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_beam_decode: beam_width %d > %d", beam_width, BEAM_MAX);
  if ((double)C * 2.0 * beam_width >= 4294967295.0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_beam_decode: C*beam too large");
  const BeamWs w = beam_ws_layout(T, B, C, beam_width);
  if (!workspace || workspace_bytes < w.total)
    ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_ctc_beam_decode: workspace %zu < %zu bytes", workspace_bytes, w.total);
  const size_t lds_c = (((size_t)C * (sizeof(double) + 2 * sizeof(int))) + 15) & ~(size_t)15;
  if (lds_c > 96 * 1024) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_beam_decode: C=%d too large for LDS", C);
  // candidate keys in LDS: all of W (C - 1) + W when that fits, else what the pruned search needs in the common case
  // (W (W + 2) + slack) up to the room left beside ~29 KB of static arrays; a frame with more candidates (ties at the
  // pruning threshold, flat posteriors) keeps that frame's keys in the workspace
  const size_t room = (size_t)160 * 1024 - 30 * 1024 - lds_c;
  size_t nkeys = (size_t)beam_width * (C - 1) + beam_width;
  if (nkeys * 8 > room) nkeys = room / 8;
  const size_t lds = lds_c + nkeys * 8;
  // 512 threads per utterance (round 6; 256 before): every phase that strides over a frame's classes / candidates gets
  // twice the lanes and the kernel still has 256 registers per thread (1024 threads: 128 registers, ~100 spilled, no faster
  // at any shape measured).  ASR_BEAM_THREADS=256 selects the four-wave form (A/B, the bit-identity test).
  const char* env_t = getenv("ASR_BEAM_THREADS");          // (read per call: the test flips it inside one process)
  const int nt = (env_t && atoi(env_t) == 256) ? 256 : 512;
  const void* kfn = nt == 512 ? (const void*)ctc_beam_kernel<512> : (const void*)ctc_beam_kernel<256>;
  (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  char* ws = (char*)workspace;
  static unsigned long long* dbgbuf = [] {
    const char* e = getenv("ASR_BEAM_DBG");
    unsigned long long* p = nullptr;
    if (e && e[0] == '1') (void)hipMalloc(&p, 8 * sizeof(unsigned long long));
    return p;
  }();
#define ASR_BEAM_LAUNCH(NT_)                                                                                              \
  hipLaunchKernelGGL(ctc_beam_kernel<NT_>, dim3(B), dim3(NT_), lds, (hipStream_t)s, logits, T, B, C, seq_len, blank,         \
                     beam_width, (double*)(ws + w.tot), (int2*)(ws + w.nodes), out_labels, out_len, out_score, (int)nkeys, \
                     dbgbuf)
  if (nt == 512) ASR_BEAM_LAUNCH(512);
  else ASR_BEAM_LAUNCH(256);
#undef ASR_BEAM_LAUNCH
  ASR_CHECK_LAUNCH(h, "asr_ctc_beam_decode");
  if (dbgbuf) {
    unsigned long long v[8];
    if (hipMemcpy(v, dbgbuf, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[beam dbg] T=%d C=%d W=%d cycles/frame: softmax %llu stay %llu class-select %llu compact %llu keys %llu "
              "select-setup %llu select %llu rank+build %llu\n", T, C, beam_width, v[0] / T, v[1] / T, v[2] / T, v[3] / T, v[4] / T, v[5] / T, v[6] / T, v[7] / T);
  }
  return ASR_OK;
}
