// CTC loss (alpha/beta DP + gradient), greedy decode, row softmax for gfx950.
//
// Replaces tf.nn.ctc_loss (CPU-only kernel in TF1 -> a device->host->device hop every
// step in the reference: models/ctc/ctc.py:289-297, joint_ctc_attention.py:308-316),
// tf.nn.ctc_greedy_decoder (ctc.py:341-342 == models/ctc/decoders/greedy_decoder.py:19-50)
// and tf.nn.softmax in CTC.posteriors (ctc.py:354-380).  blank = C-1.
//
// These are HBM/latency-bound passes over logits[T,B,C]; nothing here is GEMM-shaped:
//   1. row_lse:      one wave per (t,b) row: ln sum exp, and the emission probabilities of the row along the
//                    utterance's blank-extended label sequence (yext)          (1 read of logits)
//   2. alpha_beta:   ONE WAVE per recursion (alpha and beta of an utterance in two single-wave workgroups), LINEAR
//                    domain with a private exponent per state ({fp32 mantissa, int32 exponent}: the range of the
//                    log domain without its exp / log), K consecutive states per lane in registers, lane-boundary
//                    values over wave_shr / wave_shl DPP: no transcendental, no LDS, no barrier on the per-frame chain
//   3. grad:         one wave per (t,b): posterior occupations gamma_t(s) =
//                    alpha beta / (y p) are summed per class in a fixed order
//                    (blank by a wave tree reduction, labels by rank rounds) so the result
//                    is deterministic; writes softmax - occupation   (1 read + 1 write)
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr float NEG_INF = -INFINITY;

constexpr double DNEG_INF = -INFINITY;

// ---- 1. row log-sum-exp + emissions of the extended label sequence ----------
// One wave per (t,b) row: lse[row] = ln sum_k exp(x_k) (fp64), then -- when yext is given -- the emission
// probabilities of the row's utterance along its blank-extended label sequence l' (s even: blank, s odd: label s>>1),
//   yext[b][t][s] = {m, e}: y_t(l'_s) = exp(x[l'_s] - lse) = m * 2^e, m in [0.5, 1] fp32, e int32   for s < S,
//   {0, ME_ZERO} for S <= s < SP  (SP = 64 * K, K states per lane of kernel 2)
// i.e. already in the form the recursion multiplies with.  This is the only thing the serial kernel reads per frame:
// contiguous, prefetchable, no gather and no exp on the chain.  yrev holds the same rows in reversed time order
// (row Tb-1-t): the beta recursion then walks memory forwards exactly like alpha, and the two share one loop body.
constexpr int ME_ZERO = -(1 << 24);      // exponent of the value 0 (any real exponent stays above -126 * T)
struct __attribute__((aligned(8))) MantExp { float m; int e; };

__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ x, int rows, int C,
                                                      double* __restrict__ lse, int T, int B,
                                                      const int32_t* __restrict__ labels_flat,
                                                      const int32_t* __restrict__ label_offsets,
                                                      const int32_t* __restrict__ seq_len, int SP,
                                                      MantExp* __restrict__ yext, MantExp* __restrict__ yrev,
                                                      int32_t* __restrict__ zero_word = nullptr) {
  // (the counter the recursion kernel adds to is cleared here: as a memset it was one more packet -- 5 us and a queue gap --
  // between the output layer and the recursions)
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + (size_t)row * C;
  float m = NEG_INF;
  for (int k = lane; k < C; k += 64) m = fmaxf(m, p[k]);
  m = wave_reduce_max(m);
  double s = 0.0;
  for (int k = lane; k < C; k += 64) s += exp((double)p[k] - (double)m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const double z = (double)m + log(s);
  if (lane == 0) lse[row] = z;
  if (!yext) return;
  const int t = row / B, b = row - t * B;
  const int Tb = min(seq_len[b], T);
  if (t >= Tb) return;                                    // frames past the utterance are never read
  const int lo = label_offsets[b];
  const int S = 2 * (label_offsets[b + 1] - lo) + 1;
  const int32_t* lab = labels_flat + lo;
  MantExp* out = yext + ((size_t)b * T + t) * SP;
  MantExp* outr = yrev + ((size_t)b * T + (Tb - 1 - t)) * SP;   // the same frame where the beta recursion reads it
  const int blank = C - 1;
  for (int st = lane; st < SP; st += 64) {
    MantExp v = {0.f, ME_ZERO};
    if (st < S) {
      const int cls = (st & 1) ? lab[st >> 1] : blank;
      const double l2 = ((double)p[cls] - z) * 1.4426950408889634074;      // log2 y  (<= 0)
      if (l2 > -1.0e7) {                                                    // a logit of -inf: y = 0
        const double fl = floor(l2);
        v.e = (int)fl + 1;
        v.m = (float)exp2(l2 - fl - 1.0);                                   // 2^[-1, 0) = [0.5, 1)
      }
    }
    out[st] = v;
    outr[st] = v;
  }
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + (size_t)row * C;
  float m = NEG_INF;
  for (int k = lane; k < C; k += 64) m = fmaxf(m, p[k]);
  m = wave_reduce_max(m);
  float s = 0.f;
  for (int k = lane; k < C; k += 64) s += expf(p[k] - m);
  s = wave_reduce_sum(s);
  const float inv = 1.f / s;
  for (int k = lane; k < C; k += 64) y[(size_t)row * C + k] = expf(p[k] - m) * inv;
}

// ---- 2. alpha / beta recursions --------------------------------------------
// ONE WAVE per recursion (grid = 2 x B workgroups of 64 threads: alpha of utterance b, beta of utterance b), in the
// LINEAR domain:
//   alpha_t(s) = (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [skip] alpha_{t-1}(s-2)) * y_t(l'_s)
// with every state carried as {m, e} = m * 2^e (fp32 mantissa in [0.5, 1], private int32 exponent; 0 = {0, ME_ZERO}).
// That is the range of the log domain (the states of one frame of a peaked model differ by far more than fp64's
// 2^2046: a state that is negligible now can carry the whole likelihood a hundred frames later) without its cost:
// the sum aligns the three terms to their largest exponent (v_max3, 3 v_sub, 3 v_ldexp), the product adds exponents,
// v_frexp_* renormalises -- ~16 full-rate instructions per state and frame, no exp / log.  (The log-domain form this
// replaces spent ~1.1k cycles per frame in three exp + one log behind an LDS round trip and a 512-thread barrier.)
// Lane i owns the K consecutive states K*i .. K*i+K-1 in registers; the two states that cross a lane boundary move
// with wave_shr:1 / wave_shl:1 DPP -- no LDS, no barrier.  The recursion is started from a virtual frame
// (alpha_{-1} = indicator of state 0, beta_T = indicator of state S-1), which yields the usual initial rows.
// Emissions come from yext (kernel 1) through a register ring D frames deep; the main loop body is D frames of
// straight-line code (static ring registers, exact s_waitcnt counts), the < D remaining frames run from the ring.
// Per-frame precision: ~4 roundings of 2^-24 -> ~1e-5 relative on p(l|x) after 1000 frames, i.e. ~1e-8 relative on
// the loss; -ln p = -(ln(m1 2^(e1-E) + m2 2^(e2-E)) + E ln 2) over the two final states, in fp64.
__device__ __forceinline__ float dpp_shift_f(float v, bool right) {
  return __builtin_bit_cast(float, right ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false)
                                         : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ int dpp_shift_e(int v, bool right) {   // the lane without a source gets the exponent of 0
  return right ? __builtin_amdgcn_update_dpp(ME_ZERO, v, 0x138, 0xF, 0xF, false)
               : __builtin_amdgcn_update_dpp(ME_ZERO, v, 0x130, 0xF, 0xF, false);
}

// Round 6: NW waves per recursion.  With long label sequences (S = 2 L + 1 up to 801 states at cfg D: K = 16 states per lane) a
// frame is K x ~16 dependent-ish instructions of ONE wave: 0.7 us at K = 8 (cfg C), 1.2 ms of a 58 ms step on the critical
// path between the forward and the backward pass.  The states are dealt over NW x 64 lanes instead (K / NW per lane, lane
// index = wave * 64 + lane); the two states that cross a WAVE boundary go through LDS (double-buffered by frame parity) behind
// ONE workgroup barrier per frame.  Same operands, same operations per state: the values are bit-identical to NW = 1.
struct CtcEdge { float m1, m2; int e1, e2; };
// ybase / ws: rows in RECURSION order (step 0, 1, ...): for beta that is reversed time (row i = frame Tb-1-i).
// `lane` is the lane index over all NW waves (0 .. 64 NW - 1); SPW = states per frame in memory (64 NW K).
template <int K, int D, bool BETA, int NW>
__device__ __forceinline__ void ctc_recursion(const MantExp* __restrict__ ybase, MantExp* __restrict__ ws, int Tb,
                                              int S, int lane, const int32_t* __restrict__ lab, float (&am)[K],
                                              int (&ae)[K], CtcEdge (*edge)[NW]) {
  constexpr int SP = 64 * K * NW;
  const int wave = lane >> 6, wl = lane & 63;
  // exponent offset of the skip term: 0 where the transition s-2 -> s (beta: s+2 -> s) exists, else "times 0"
  int skoff[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int s = K * lane + k;
    bool ok = false;
    if ((s & 1) && s < S) {
      if (!BETA) ok = s >= 3 && lab[s >> 1] != lab[(s >> 1) - 1];
      else ok = s + 2 < S && lab[s >> 1] != lab[(s >> 1) + 1];
    }
    skoff[k] = ok ? 0 : ME_ZERO;
  }
  // virtual start frame
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int s = K * lane + k;
    const bool on = BETA ? (s == S - 1) : (s == 0);
    am[k] = on ? 0.5f : 0.f;
    ae[k] = on ? 1 : ME_ZERO;
  }
  float rm[D][K];
  int re[D][K];
  auto fetch = [&](int slot, int step) {                   // emissions of recursion step `step` (clamped: no branch)
    const MantExp* src = ybase + (size_t)min(step, Tb - 1) * SP;
#pragma unroll
    for (int k = 0; k < K; ++k) { const MantExp v = src[k]; rm[slot][k] = v.m; re[slot][k] = v.e; }
  };
  auto frame = [&](int slot, int step) {
    float nm[K];
    int ne[K];
    // (K = 1: the second neighbour state is the first state of the lane TWO away; not instantiated, K >= 2 here)
    float p1m = dpp_shift_f(BETA ? am[0] : am[K - 1], !BETA), p2m = dpp_shift_f(BETA ? am[1] : am[K - 2], !BETA);
    int p1e = dpp_shift_e(BETA ? ae[0] : ae[K - 1], !BETA), p2e = dpp_shift_e(BETA ? ae[1] : ae[K - 2], !BETA);
    if constexpr (NW > 1) {
      // the lane at the wave's edge publishes what its neighbour WAVE needs, the lane at the other edge picks it up.
      // (Measured: computing the frame's states under the barrier and redoing the two edge states behind it is slower --
      // 1 219 against 1 177 us per call at cfg C: the frame is the LDS -> barrier -> LDS chain, not the state arithmetic.)
      CtcEdge* row = edge[step & 1];
      if (wl == (BETA ? 0 : 63)) row[wave] = (CtcEdge){BETA ? am[0] : am[K - 1], BETA ? am[1] : am[K - 2],
                                                        BETA ? ae[0] : ae[K - 1], BETA ? ae[1] : ae[K - 2]};
      __syncthreads();
      const int nbw = BETA ? wave + 1 : wave - 1;
      if (nbw >= 0 && nbw < NW) {                         // wave-uniform
        const CtcEdge nb = row[nbw];
        const bool at = wl == (BETA ? 63 : 0);
        p1m = at ? nb.m1 : p1m;  p2m = at ? nb.m2 : p2m;
        p1e = at ? nb.e1 : p1e;  p2e = at ? nb.e2 : p2e;
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float m1, m2;
      int e1, e2;
      if (!BETA) {
        m1 = k >= 1 ? am[k - 1] : p1m;  e1 = k >= 1 ? ae[k - 1] : p1e;
        m2 = k >= 2 ? am[k - 2] : (k == 1 ? p1m : p2m);  e2 = k >= 2 ? ae[k - 2] : (k == 1 ? p1e : p2e);
      } else {
        m1 = k + 1 < K ? am[k + 1] : p1m;  e1 = k + 1 < K ? ae[k + 1] : p1e;
        m2 = k + 2 < K ? am[k + 2] : (k + 2 == K ? p1m : p2m);  e2 = k + 2 < K ? ae[k + 2] : (k + 2 == K ? p1e : p2e);
      }
      e2 += skoff[k];
      const int E = max(max(ae[k], e1), e2);
      const float sum = ldexpf(am[k], ae[k] - E) + ldexpf(m1, e1 - E) + ldexpf(m2, e2 - E);
      const float pr = sum * rm[slot][k];
      nm[k] = __builtin_amdgcn_frexp_mantf(pr);
      ne[k] = pr > 0.f ? E + re[slot][k] + __builtin_amdgcn_frexp_expf(pr) : ME_ZERO;
    }
    MantExp* dst = ws + (size_t)step * SP;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      am[k] = nm[k];
      ae[k] = ne[k];
      dst[k] = (MantExp){nm[k], ne[k]};
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j) fetch(j, j);
  int base = 0;
  for (; base + D <= Tb; base += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      frame(j, base + j);
      fetch(j, base + j + D);
    }
  }
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (base + j < Tb) frame(j, base + j);                 // wave-uniform; slot j holds frame base + j
}

template <int K, int D, int NW = 1>
__global__ __launch_bounds__(64 * NW) void ctc_alpha_beta_kernel(
    const MantExp* __restrict__ yext, const MantExp* __restrict__ yrev, int T, int B, int C,
    const int32_t* __restrict__ labels_flat, const int32_t* __restrict__ label_offsets,
    const int32_t* __restrict__ seq_len, int Lcap, MantExp* __restrict__ alpha_ws, MantExp* __restrict__ beta_ws, int32_t* __restrict__ rank_ws,
    double* __restrict__ ll_ws, float* __restrict__ loss, int32_t* __restrict__ num_infeasible) {
  constexpr int SP = 64 * K * NW;
  static_assert(K >= 2, "two states cross a lane boundary");
  __shared__ CtcEdge edge[2][NW];
  __shared__ CtcEdge fin[NW];
  const int b = blockIdx.x >> 1;
  const bool is_beta = blockIdx.x & 1;
  const int lane = threadIdx.x;                            // over all NW waves
  const int lo = label_offsets[b];
  const int L = label_offsets[b + 1] - lo;
  const int S = 2 * L + 1;
  const int Tb = min(seq_len[b], T);
  const int32_t* lab = labels_flat + lo;

  if (!is_beta) {
    // rank of each label among equal earlier labels (fixed summation order in the grad kernel) and the position of
    // the first label of its class (where the grad kernel accumulates that class), packed rank | first << 16
    for (int i = lane; i < L && i < Lcap; i += 64 * NW) {
      const int li = lab[i];
      int r = 0, first = i;
      for (int k = i - 1; k >= 0; --k)
        if (lab[k] == li) { ++r; first = k; }
      rank_ws[(size_t)b * (Lcap > 0 ? Lcap : 1) + i] = r | (first << 16);
    }
  }
  if (Tb <= 0 || L > Lcap || S > SP) {
    if (!is_beta && lane == 0) {
      loss[b] = 0.f;
      ll_ws[b] = DNEG_INF;
      if (num_infeasible && Tb > 0) atomicAdd(num_infeasible, 1);
    }
    return;
  }
  const MantExp* ybase = (is_beta ? yrev : yext) + (size_t)b * T * SP + K * lane;
  float am[K];
  int ae[K];
  if (is_beta) {
    ctc_recursion<K, D, true, NW>(ybase, beta_ws + (size_t)b * T * SP + K * lane, Tb, S, lane, lab, am, ae, edge);
    return;
  }
  ctc_recursion<K, D, false, NW>(ybase, alpha_ws + (size_t)b * T * SP + K * lane, Tb, S, lane, lab, am, ae, edge);
  // p = alpha_T(S-1) + alpha_T(S-2): the two states live in lanes (S-1)/K and (S-2)/K
  float m1 = 0.f, m2 = 0.f;
  int e1 = ME_ZERO, e2 = ME_ZERO;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int s = K * lane + k;
    if (s == S - 1) { m1 = am[k]; e1 = ae[k]; }
    if (s == S - 2) { m2 = am[k]; e2 = ae[k]; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {                       // every lane but the owners holds {0, ME_ZERO}
    const float om1 = __shfl_xor(m1, o, 64), om2 = __shfl_xor(m2, o, 64);
    const int oe1 = __shfl_xor(e1, o, 64), oe2 = __shfl_xor(e2, o, 64);
    if (oe1 > e1) { m1 = om1; e1 = oe1; }
    if (oe2 > e2) { m2 = om2; e2 = oe2; }
  }
  if constexpr (NW > 1) {                                  // the owners of the two final states may sit in different waves
    if ((lane & 63) == 0) fin[lane >> 6] = (CtcEdge){m1, m2, e1, e2};
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        if (fin[w].e1 > e1) { m1 = fin[w].m1; e1 = fin[w].e1; }
        if (fin[w].e2 > e2) { m2 = fin[w].m2; e2 = fin[w].e2; }
      }
    }
  }
  if (lane == 0) {
    const int E = max(e1, e2);
    const bool ok = E > ME_ZERO;
    double ll = DNEG_INF;
    if (ok) ll = log(ldexp((double)m1, e1 - E) + ldexp((double)m2, e2 - E)) + (double)E * 0.6931471805599453094;
    loss[b] = ok ? (float)(-ll) : 0.f;
    ll_ws[b] = ll;
    if (!ok && num_infeasible) atomicAdd(num_infeasible, 1);
  }
}

// ---- 3. gradient -------------------------------------------------------------
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logits, const double* __restrict__ lse, int T, int B, int C,
    const int32_t* __restrict__ labels_flat, const int32_t* __restrict__ label_offsets,
    const int32_t* __restrict__ seq_len, int SW, int SP, const MantExp* __restrict__ yext,
    const MantExp* __restrict__ alpha_ws, const MantExp* __restrict__ beta_ws,
    const int32_t* __restrict__ rank_ws, const double* __restrict__ ll_ws, float grad_scale,
    float* __restrict__ grad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Per wave: gam[SW] (state occupations), acc[(SW-1)/2] (occupation of a class, kept at the position of its FIRST
  // label) and a C-bit membership map of the label classes.  Nothing is sized by C x 4 bytes, so word-level
  // vocabularies (C = 18-27 k classes, examples/librispeech) fit: 4 x (SW + Lmax + C/32) words of LDS.
  const int LW = (SW - 1) / 2, BW = (C + 31) / 32;
  float* gam = reinterpret_cast<float*>(smem) + (size_t)wave * (SW + LW + BW);
  float* acc = gam + SW;
  unsigned* bits = reinterpret_cast<unsigned*>(acc + LW);
  const int b = blockIdx.y;
  const int t = blockIdx.x * 4 + wave;
  if (t >= T) return;
  float* g = grad + ((size_t)t * B + b) * C;
  const int Tb = min(seq_len[b], T);
  const double ll = ll_ws[b];
  if (t >= Tb || !(ll > DNEG_INF)) {
    for (int k = lane; k < C; k += 64) g[k] = 0.f;
    return;
  }
  const int lo = label_offsets[b];
  const int L = label_offsets[b + 1] - lo;
  const int S = 2 * L + 1;
  const int blank = C - 1;
  const int32_t* lab = labels_flat + lo;
  const int32_t* rk = rank_ws + (size_t)b * (LW > 0 ? LW : 1);
  const float* row = logits + ((size_t)t * B + b) * C;
  const double z = lse[(size_t)t * B + b];
  const MantExp* al = alpha_ws + ((size_t)b * T + t) * SP;
  const MantExp* be = beta_ws + ((size_t)b * T + (Tb - 1 - t)) * SP;   // stored in recursion order
  const MantExp* ye = yext + ((size_t)b * T + t) * SP;

  for (int k = lane; k < BW; k += 64) bits[k] = 0u;
  for (int i = lane; i < L; i += 64) acc[i] = 0.f;
  float blank_sum = 0.f;
  int maxrank = 0;
  for (int s = lane; s < S; s += 64) {
    // gamma_t(s) = alpha_t(s) beta_t(s) / (y_t(s) p): mantissas in fp64, exponents as integers, ln p = ll
    const MantExp av = al[s], bv = be[s], yv = ye[s];
    float gm = 0.f;
    if (av.m > 0.f && bv.m > 0.f)
      gm = (float)((double)av.m * (double)bv.m / (double)yv.m *
                   exp((double)(av.e + bv.e - yv.e) * 0.6931471805599453094 - ll));
    gam[s] = gm;
    if (!(s & 1)) blank_sum += gm;
    else maxrank = max(maxrank, rk[s >> 1] & 0xffff);
  }
  blank_sum = wave_reduce_sum(blank_sum);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxrank = max(maxrank, __shfl_xor(maxrank, o, 64));
  __builtin_amdgcn_wave_barrier();
  // labels: round r adds the r-th occurrence of every class to the slot of its first occurrence -> no two lanes
  // touch one slot in a round, and the summation order is fixed (deterministic)
  for (int r = 0; r <= maxrank; ++r) {
    for (int i = lane; i < L; i += 64) {
      const int pk = rk[i];
      if ((pk & 0xffff) == r) {
        acc[pk >> 16] += gam[2 * i + 1];
        if (r == 0) atomicOr(&bits[lab[i] >> 5], 1u << (lab[i] & 31));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // classes that are not labels (and not blank): softmax only; every address below is written exactly once
  for (int k = lane; k < C; k += 64) {
    if (k == blank) g[k] = ((float)exp((double)row[k] - z) - blank_sum) * grad_scale;
    else if (!((bits[k >> 5] >> (k & 31)) & 1u)) g[k] = (float)exp((double)row[k] - z) * grad_scale;
  }
  for (int i = lane; i < L; i += 64) {
    if ((rk[i] & 0xffff) == 0) {
      const int k = lab[i];
      g[k] = ((float)exp((double)row[k] - z) - acc[i]) * grad_scale;
    }
  }
}

// ---- greedy decode -------------------------------------------------------------
// tf.nn.ctc_greedy_decoder (ctc.py:341-342) / GreedyDecoder (greedy_decoder.py:19-50): per-frame argmax (first maximum
// wins), collapse repeats, drop blanks.  Two launches:
//  1. ctc_argmax_kernel -- one WAVE per valid (t, b) row over the whole grid: the only pass over the logits, HBM-bound
//     (a 3 387-class row is 13.5 KB: with one workgroup per utterance -- the first version -- 8 utterances kept 8 CUs busy
//     and the pass ran at 24 GB/s); writes the frame's argmax to out_labels[b, t], which serves as the scratch row;
//  2. ctc_collapse_kernel -- one workgroup per utterance: its row of argmaxes -> LDS, ordered compaction back into
//     out_labels[b, :n], -1 behind it.
__global__ __launch_bounds__(256) void ctc_argmax_kernel(const float* __restrict__ logits, int T, int B, int C,
                                                         const int32_t* __restrict__ seq_len,
                                                         int32_t* __restrict__ out_labels) {
  const int lane = threadIdx.x & 63;
  const size_t nrows = (size_t)T * B;
  for (size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += (size_t)gridDim.x * 4) {
    const int t = (int)(row / B), b = (int)(row % B);
    if (t >= min(max(seq_len[b], 0), T)) continue;       // wave-uniform
    const float* p = logits + row * C;
    float best = NEG_INF;
    int bi = 0x7fffffff;
    for (int k = lane; k < C; k += 64) {
      const float v = p[k];
      if (v > best || (v == best && k < bi)) { best = v; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) out_labels[(size_t)b * T + t] = (bi == 0x7fffffff) ? 0 : bi;
  }
}

__global__ __launch_bounds__(256) void ctc_collapse_kernel(int T, const int32_t* __restrict__ seq_len, int blank,
                                                           int32_t* __restrict__ out_labels,
                                                           int32_t* __restrict__ out_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* am = reinterpret_cast<int32_t*>(smem);  // [T] per-frame argmax
  __shared__ int32_t wsum[4];
  __shared__ int32_t carry;
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Tb = min(max(seq_len[b], 0), T);
  for (int t = threadIdx.x; t < Tb; t += 256) am[t] = out_labels[(size_t)b * T + t];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  // ordered compaction, 256 frames per pass
  for (int base = 0; base < Tb; base += 256) {
    const int t = base + threadIdx.x;
    int keep = 0, a = 0;
    if (t < Tb) {
      a = am[t];
      keep = (a != blank) && (t == 0 || a != am[t - 1]);
    }
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (keep) out_labels[(size_t)b * T + off + before] = a;
    __syncthreads();
    if (threadIdx.x == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int n = carry;
  for (int i = n + threadIdx.x; i < T; i += 256) out_labels[(size_t)b * T + i] = -1;
  if (threadIdx.x == 0) out_len[b] = n;
}

// states per lane of the recursion kernel for an extended label sequence of SW = 2 Lmax + 1 states (0: too long)
inline int ctc_states_per_lane(int SW) {
  const int ks[7] = {3, 6, 8, 12, 16, 24, 32};
  for (int i = 0; i < 7; ++i)
    if (64 * ks[i] >= SW) return ks[i];
  return 0;
}
struct CtcWs {
  size_t lse, yext, yrev, alpha, beta, rank, ll, total;
  int K;
};
inline CtcWs ctc_ws_layout(int T, int B, int Lmax) {
  const int SW = 2 * Lmax + 1;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  CtcWs w;
  w.K = ctc_states_per_lane(SW);
  const size_t SP = 64 * (size_t)(w.K ? w.K : 32);
  size_t o = 0;
  w.lse = o;   o += al((size_t)T * B * 8);
  w.yext = o;  o += al((size_t)B * T * SP * sizeof(MantExp));
  w.yrev = o;  o += al((size_t)B * T * SP * sizeof(MantExp));
  w.alpha = o; o += al((size_t)B * T * SP * sizeof(MantExp));
  w.beta = o;  o += al((size_t)B * T * SP * sizeof(MantExp));
  w.rank = o;  o += al((size_t)B * (Lmax > 0 ? Lmax : 1) * 4);
  w.ll = o;    o += al((size_t)B * 8);
  w.total = o;
  return w;
}

}  // namespace

extern "C" size_t asr_ctc_workspace_bytes(int T, int B, int max_label_len) {
  if (T < 0 || B < 0 || max_label_len < 0) return 0;
  return ctc_ws_layout(T, B, max_label_len).total;
}

extern "C" int asr_ctc_loss(asr_handle* h, const float* logits, int T, int B, int C,
                            const int32_t* labels_flat, const int32_t* label_offsets,
                            const int32_t* seq_len, int max_label_len, float grad_scale, float* loss,
                            float* grad, int32_t* num_infeasible, void* workspace,
                            size_t workspace_bytes, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!logits || !label_offsets || !seq_len || !loss || T <= 0 || B <= 0 || C < 2 ||
      max_label_len < 0 || (max_label_len > 0 && !labels_flat))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_ctc_loss: bad args T=%d B=%d C=%d Lmax=%d", T, B, C, max_label_len);
  const int SW = 2 * max_label_len + 1;
  const CtcWs w = ctc_ws_layout(T, B, max_label_len);
  if (!w.K) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_loss: label length %d > %d", max_label_len, (64 * 32 - 1) / 2);
  if (!workspace || workspace_bytes < w.total)
    ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_ctc_loss: workspace %zu < %zu bytes", workspace_bytes, w.total);
  char* ws = (char*)workspace;
  double* lse = (double*)(ws + w.lse);
  MantExp* yext = (MantExp*)(ws + w.yext);
  MantExp* yrev = (MantExp*)(ws + w.yrev);
  MantExp* alpha = (MantExp*)(ws + w.alpha);
  MantExp* beta = (MantExp*)(ws + w.beta);
  int32_t* rank = (int32_t*)(ws + w.rank);
  double* ll = (double*)(ws + w.ll);
  hipStream_t st = (hipStream_t)s;
  const int rows = T * B;
  const int SP = 64 * w.K;
  hipLaunchKernelGGL(row_lse_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, rows, C, lse, T, B, labels_flat,
                     label_offsets, seq_len, SP, yext, yrev, num_infeasible);
  ASR_CHECK_LAUNCH(h, "asr_ctc_loss(row_lse)");
  {
#define ASR_AB(KV, DV, NWV)                                                                              \
  hipLaunchKernelGGL((ctc_alpha_beta_kernel<KV, DV, NWV>), dim3(2 * B), dim3(64 * NWV), 0, st, yext, yrev, T, B, C, labels_flat, \
                     label_offsets, seq_len, max_label_len, alpha, beta, rank, ll, loss, num_infeasible)
    // ASR_CTC_WAVES=1: one wave per recursion whatever the label length (round 5; A/B and the bit-identity test)
    const char* env_w = getenv("ASR_CTC_WAVES");            // (read per call: the bit-identity test flips it inside one process)
    const bool multi = !(env_w && env_w[0] == '1');
    if (multi && w.K >= 6) {
      switch (w.K) {                          // K states per lane of ONE wave = K / NW per lane of NW waves
        case 6: ASR_AB(3, 8, 2); break;
        case 8: ASR_AB(2, 8, 4); break;
        case 12: ASR_AB(3, 8, 4); break;
        case 16: ASR_AB(4, 8, 4); break;
        case 24: ASR_AB(6, 4, 4); break;
        default: ASR_AB(8, 4, 4); break;
      }
    } else {
      switch (w.K) {
        case 3: ASR_AB(3, 8, 1); break;       // D frames of emissions in flight: D * K * 2 ring registers
        case 6: ASR_AB(6, 4, 1); break;
        case 8: ASR_AB(8, 4, 1); break;
        case 12: ASR_AB(12, 4, 1); break;
        case 16: ASR_AB(16, 3, 1); break;
        case 24: ASR_AB(24, 2, 1); break;
        default: ASR_AB(32, 2, 1); break;
      }
    }
#undef ASR_AB
  }
  ASR_CHECK_LAUNCH(h, "asr_ctc_loss(alpha_beta)");
  if (grad) {
    const size_t lds_g = (size_t)4 * (SW + (SW - 1) / 2 + (C + 31) / 32) * sizeof(float);
    if (lds_g > 160 * 1024)
      ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_loss: C=%d, Lmax=%d need %zu B of LDS", C, max_label_len, lds_g);
    (void)hipFuncSetAttribute((const void*)ctc_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((T + 3) / 4, B), dim3(256), lds_g, st, logits, lse, T, B, C,
                       labels_flat, label_offsets, seq_len, SW, SP, yext, alpha, beta, rank, ll, grad_scale, grad);
    ASR_CHECK_LAUNCH(h, "asr_ctc_loss(grad)");
  }
  return ASR_OK;
}

extern "C" int asr_ctc_greedy_decode(asr_handle* h, const float* logits, int T, int B, int C,
                                     const int32_t* seq_len, int blank, int32_t* out_labels,
                                     int32_t* out_len, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!logits || !seq_len || !out_labels || !out_len || T <= 0 || B <= 0 || C < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_ctc_greedy_decode: bad args T=%d B=%d C=%d", T, B, C);
  const size_t lds = (size_t)T * sizeof(int32_t);
  if (lds > 150 * 1024) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_greedy_decode: T=%d too long", T);
  (void)hipFuncSetAttribute((const void*)ctc_collapse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const size_t nrows = (size_t)T * B;
  const unsigned ablocks = (unsigned)((nrows + 3) / 4 < 16384 ? (nrows + 3) / 4 : 16384);
  hipLaunchKernelGGL(ctc_argmax_kernel, dim3(ablocks), dim3(256), 0, (hipStream_t)s, logits, T, B, C, seq_len, out_labels);
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(256), lds, (hipStream_t)s, T, seq_len, blank, out_labels,
                     out_len);
  ASR_CHECK_LAUNCH(h, "asr_ctc_greedy_decode");
  return ASR_OK;
}

extern "C" int asr_softmax_rows(asr_handle* h, const float* in, float* out, int rows, int C,
                                asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!in || !out || rows < 0 || C < 1) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_softmax_rows: bad args");
  if (rows == 0) return ASR_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, in, out, rows, C);
  ASR_CHECK_LAUNCH(h, "asr_softmax_rows");
  return ASR_OK;
}
