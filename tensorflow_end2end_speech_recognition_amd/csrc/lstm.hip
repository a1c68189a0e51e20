// LSTM recurrence for gfx950: the serial hot loop of the encoder.
//
// Replaces  tf.contrib.rnn.LSTMBlockCell + tf.nn.(bidirectional_)dynamic_rnn(sequence_length)
// (models/encoders/core/blstm.py:286-323, lstm.py:253-285; cell equations as in
// models/recurrent/layers/lstm.py:142-170 and SURVEY.md Appendix B):
//   icfo = x W_x + b (hoisted, one GEMM over all T)  +  h_{t-1} W_h   (this kernel)
//   i = sig(i + wci*c_prev)  ci = tanh(ci)  f = sig(f + fb + wcf*c_prev)
//   c = ci*i + c_prev*f ; clip ; o = sig(o + wco*c) ; h = tanh(c)*o
//   t >= seq_len[b]: output 0, state carried; the backward direction walks frames
//   len-1 .. 0 (reverse_sequence semantics).
//
// Mapping (MI355X-first, not a port of TF's per-step op):
//   * one workgroup (up to 16 waves = 4 per SIMD) per (direction, 16-utterance batch tile)
//     runs ALL T steps; the 16 utterances are the M dimension of a 16x16 MFMA tile, so a lane
//     of the C/D fragment permanently owns (utterance b = (lane>>4)*4+r, unit j = ub*16+(lane&15)):
//     c, h, the peepholes and the four gate pre-activations of that (b,j) never leave its
//     registers -> the gate math needs no cross-lane traffic.
//   * wave w owns unit blocks ub = w, w+NW, ...; for each it accumulates the four gate tiles
//     (i, ci, f, o) so one lane ends a step with all four gates of its (b,j).
//   * h_{t-1} (16 x H) lives in LDS (double buffered, one barrier per step) as the MFMA A
//     operand; W_h is pre-packed in B-fragment order so every wave-load is one contiguous
//     1 KiB line; as many k-chunks as fit are parked in LDS for the whole launch, the rest is
//     streamed from L2 each step.
//   * The loop is VALU-issue bound, so the data layout is chosen to minimise
//     instructions per (b,j) pair: the four gates of a unit are INTERLEAVED in memory
//     ([T,B,dir,H,4]: x W_x + b, saved gates, gate gradients), i.e. one 16-B load and one
//     8/16-B store per pair per step, 32-bit element offsets, v_cvt_pk_bf16_f32 packing and
//     v_exp/v_rcp gate functions.
#include "common.h"
#include <stdlib.h>

namespace {

template <typename T> struct LT;
template <> struct LT<float> {
  static constexpr int KV = 16;  // k covered by one packed 16-B fragment load (4 MFMAs of K=4)
  static constexpr int PAD = 4;
  typedef f32x4_t g4_t;          // four gates of one unit
  static __device__ __forceinline__ g4_t pack(float a, float b, float c, float d) { return (g4_t){a, b, c, d}; }
  static __device__ __forceinline__ void unpack(const g4_t& v, float& a, float& b, float& c, float& d) {
    a = v[0]; b = v[1]; c = v[2]; d = v[3];
  }
  static __device__ __forceinline__ float cvt(float v) { return v; }
};
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
template <> struct LT<bf16_t> {
  static constexpr int KV = 32;  // one 16x16x32 MFMA
  static constexpr int PAD = 8;
  typedef bf16x4_t g4_t;
  static __device__ __forceinline__ g4_t pack(float a, float b, float c, float d) {
    return (g4_t){(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};   // 2 x v_cvt_pk_bf16_f32
  }
  static __device__ __forceinline__ void unpack(const g4_t& v, float& a, float& b, float& c, float& d) {
    a = (float)v[0]; b = (float)v[1]; c = (float)v[2]; d = (float)v[3];
  }
  static __device__ __forceinline__ bf16_t cvt(float v) { return __builtin_bit_cast(bf16_t, (__bf16)v); }
};

// ---------------------------------------------------------------- weight preparation
// One pass per (layer, direction) and step, over kernel [Din+H, 4H] fp32 (TF layout, gate-major
// columns q*H + j) and bias [4H]:
//   wx_il  [Din,4H] T   : W_x with INTERLEAVED columns j*4+q  (B operand of the hoisted GEMM)
//   bias_il[4H]     f32 : same permutation
//   pf              T   : W_h packed as MFMA B fragments for  h[16,H] x W_h        (tiles (ub,q))
//   pb              T   : W_h packed as MFMA B fragments for dG[16,4H'] x W_h^T    (k' interleaved)
template <typename T>
__global__ void prep_weights_kernel(const float* __restrict__ kernel, const float* __restrict__ bias,
                                    int Din, int H, T* __restrict__ wx_il, float* __restrict__ bias_il,
                                    T* __restrict__ pf, T* __restrict__ pb) {
  constexpr int KV = LT<T>::KV;
  constexpr int E = 16 / sizeof(T);
  const int G = 4 * H;
  const size_t n_wx = (size_t)Din * G, n_wh = (size_t)H * G;
  const size_t total = n_wx + n_wh + G;
  const float* wh = kernel + n_wx;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    if (idx < n_wx) {
      const int r = idx / G, cp = idx % G;        // cp = j*4 + q
      wx_il[idx] = Elem<T>::from_f32(kernel[(size_t)r * G + (cp & 3) * H + (cp >> 2)]);
    } else if (idx < n_wx + n_wh) {
      const size_t i2 = idx - n_wx;
      const int e = i2 % E;
      const int lane = (i2 / E) % 64;
      const size_t frag = i2 / (E * 64);
      const int n = lane & 15, rg = lane >> 4;
      const int kin = rg * E + e;   // element e of lane (n, rg): bf16 k = rg*8+e ; fp32 k = rg*4+e
      {  // forward: B[k][n] = Wh[k][q*H + ub*16 + n]
        const int KS = H / KV;
        const int ks = frag % KS;
        const int tile = frag / KS;  // ub*4 + q
        const int q = tile & 3, ub = tile >> 2;
        const int k = ks * KV + kin;
        pf[i2] = Elem<T>::from_f32(wh[(size_t)k * G + q * H + ub * 16 + n]);
      }
      {  // backward: B[k'][n] = Wh[ub*16 + n][(k'&3)*H + (k'>>2)],  k' = j*4 + q
        const int KS = G / KV;
        const int ks = frag % KS;
        const int ub = frag / KS;
        const int kp = ks * KV + kin;
        pb[i2] = Elem<T>::from_f32(wh[(size_t)(ub * 16 + n) * G + (kp & 3) * H + (kp >> 2)]);
      }
    } else {
      const int cp = idx - n_wx - n_wh;
      bias_il[cp] = bias[(cp & 3) * H + (cp >> 2)];
    }
  }
}

// rows of [R, 4H]: interleaved columns (j*4+q) -> gate-major (q*H+j); used for dW after the GEMMs
__global__ void deinterleave_cols_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                         int ld_out, int R, int H) {
  const int G = 4 * H;
  const size_t total = (size_t)R * G;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int r = idx / G, c = idx % G;   // c = q*H + j (output, coalesced)
    const int q = c / H, j = c % H;
    out[(size_t)r * ld_out + c] = in[(size_t)r * ld_in + j * 4 + q];
  }
}

// Everything a layer's launches consume that depends only on its variables, for BOTH directions in one pass (the
// per-direction prep_weights + transpose + concatenations it replaces were ~9 launches per layer and step):
//   wxT    [ndir*4H, ldk]      T   W_x^T with interleaved rows j*4+q, directions stacked, k padded with zeros to ldk
//                                  (both operands of the hoisted GEMM x W_x reduction-contiguous)
//   wx_cat [Din, ndir*4H]      T   W_x with interleaved columns, directions side by side (dx = dG wx_cat^T: one GEMM)
//   bias   [ndir*4H]         f32   interleaved
//   pf, pb [ndir][H*4H]        T   the two MFMA packings of W_h (see prep_weights_kernel)
//   peep   [ndir][3][H]      f32   w_i_diag, w_f_diag, w_o_diag (only if the layer has peepholes)
struct PrepVars { const float* kernel[2]; const float* bias[2]; const float* peep[2][3]; };
template <typename T>
__global__ void prep_layer_kernel(PrepVars v, int ndir, int Din, int ldk, int H, T* __restrict__ wxT,
                                  T* __restrict__ wx_cat, float* __restrict__ bias_cat, T* __restrict__ pf,
                                  T* __restrict__ pb, float* __restrict__ peep_out) {
  constexpr int KV = LT<T>::KV;
  constexpr int E = 16 / sizeof(T);
  const int G = 4 * H;
  const size_t n_t = (size_t)G * ldk, n_wx = (size_t)Din * G, n_wh = (size_t)H * G;
  const size_t n_peep = peep_out ? (size_t)3 * H : 0;
  const size_t per_dir = n_t + n_wx + n_wh + G + n_peep;
  for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < per_dir * ndir;
       gidx += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(gidx / per_dir);
    size_t idx = gidx - (size_t)d * per_dir;
    const float* kernel = v.kernel[d];
    const float* wh = kernel + n_wx;
    if (idx < n_t) {                                       // wxT: row cp = j*4+q, column r (k), coalesced writes
      const int cp = (int)(idx / ldk), r = (int)(idx % ldk);
      const float w = r < Din ? kernel[(size_t)r * G + (cp & 3) * H + (cp >> 2)] : 0.f;
      wxT[((size_t)d * G + cp) * ldk + r] = Elem<T>::from_f32(w);
      continue;
    }
    idx -= n_t;
    if (idx < n_wx) {
      const int r = (int)(idx / G), cp = (int)(idx % G);
      wx_cat[(size_t)r * ndir * G + (size_t)d * G + cp] = Elem<T>::from_f32(kernel[(size_t)r * G + (cp & 3) * H + (cp >> 2)]);
      continue;
    }
    idx -= n_wx;
    if (idx < n_wh) {
      const size_t i2 = idx;
      const int e = i2 % E;
      const int lane = (i2 / E) % 64;
      const size_t frag = i2 / (E * 64);
      const int n = lane & 15, rg = lane >> 4;
      const int kin = rg * E + e;
      {
        const int KS = H / KV;
        const int ks = frag % KS;
        const int tile = frag / KS;
        const int q = tile & 3, ub = tile >> 2;
        const int k = ks * KV + kin;
        pf[(size_t)d * n_wh + i2] = Elem<T>::from_f32(wh[(size_t)k * G + q * H + ub * 16 + n]);
      }
      {
        const int KS = G / KV;
        const int ks = frag % KS;
        const int ub = frag / KS;
        const int kp = ks * KV + kin;
        pb[(size_t)d * n_wh + i2] = Elem<T>::from_f32(wh[(size_t)(ub * 16 + n) * G + (kp & 3) * H + (kp >> 2)]);
      }
      continue;
    }
    idx -= n_wh;
    if (idx < (size_t)G) {
      const int cp = (int)idx;
      bias_cat[(size_t)d * G + cp] = v.bias[d][(cp & 3) * H + (cp >> 2)];
      continue;
    }
    idx -= G;
    peep_out[(size_t)d * 3 * H + idx] = v.peep[d][idx / H][idx % H];
  }
}

// The inverse trip for the gradients of one layer (both directions, one launch): dW [ndir][Din+H][4H] with interleaved
// columns (what the weight-gradient GEMMs write) -> kernel gradient in TF's gate-major layout; rows 3..6 of
// dpeep [ndir][7][H] (bias gradient by gate, accumulated inside the BPTT kernel) -> bias gradient; rows 0..2 -> the
// three peephole gradients.
struct GradVars { float* kernel[2]; float* bias[2]; float* peep[2][3]; };
__global__ void grad_finish_kernel(GradVars v, int ndir, int R, int H, const float* __restrict__ dw_il,
                                   const float* __restrict__ dpeep, int has_peep) {
  const int G = 4 * H;
  const size_t n_w = (size_t)R * G;
  const size_t per_dir = n_w + G + (has_peep ? (size_t)3 * H : 0);
  for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < per_dir * ndir;
       gidx += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(gidx / per_dir);
    size_t idx = gidx - (size_t)d * per_dir;
    if (idx < n_w) {
      const int r = (int)(idx / G), c = (int)(idx % G);    // c = q*H + j (output, coalesced)
      const int q = c / H, j = c % H;
      v.kernel[d][idx] = dw_il[(size_t)d * n_w + (size_t)r * G + j * 4 + q];
      continue;
    }
    idx -= n_w;
    if (idx < (size_t)G) {
      v.bias[d][idx] = dpeep[((size_t)d * 7 + 3) * H + idx];
      continue;
    }
    idx -= G;
    v.peep[d][idx / H][idx % H] = dpeep[(size_t)d * 7 * H + idx];
  }
}

// one k-chunk of MFMA work: acc += A(16 x KV) * B(KV x 16)
__device__ __forceinline__ f32x4_t mma_chunk(const bf16x8_t& a, const bf16x8_t& b, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mma_chunk(const f32x4_t& a, const f32x4_t& b, f32x4_t acc) {
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  return acc;
}
template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4_t type; };
template <> struct Frag<bf16_t> { typedef bf16x8_t type; };

// ---------------------------------------------------------------- gate math
// v_exp_f32 / v_rcp_f32 forms (1 ulp each): a sigmoid is 4 VALU ops instead of the ~25 of
// expf + IEEE division.
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

// Waves that share a SIMD (w, w+4, w+8, w+12) contend for one MFMA pipe.  Giving them distinct
// static priorities while they issue MFMAs makes them finish the matrix phase one after the
// other, so the early ones run their (VALU/transcendental-bound) gate math while the late ones
// still own the matrix pipe -- instead of all waves doing MFMA, then all doing VALU.
__device__ __forceinline__ void mfma_phase_prio(int wave) {
  const int p = __builtin_amdgcn_readfirstlane(wave >> 2);
  if (p == 0) __builtin_amdgcn_s_setprio(3);
  else if (p == 1) __builtin_amdgcn_s_setprio(2);
  else if (p == 2) __builtin_amdgcn_s_setprio(1);
}

// waves per workgroup: the largest divisor of H/16 (unit blocks) that is <= 16, so that every
// wave owns NUB = H/(16*NW) unit blocks and the CU runs up to 4 waves per SIMD.
constexpr int pick_nw(int H) {
  int nb = H / 16, best = 1;
  for (int w = 1; w <= 16; ++w)
    if (nb % w == 0) best = w;
  return best;
}
// k-chunks of every tile of W_h parked in LDS for the whole launch (the rest is streamed from
// L2 every step); sized to leave room for the h / dG buffers; even (keeps the streamed loop's
// trip count a multiple of its unroll).
template <typename T> constexpr int fwd_ksl(int H) {
  const int tiles = (H / 16) * 4, ks = H / LT<T>::KV;
  const long hbuf = 2L * 16 * (H + LT<T>::PAD) * (long)sizeof(T);
  long k = (150L * 1024 - hbuf) / (tiles * 1024L);
  if (k < 0) k = 0;
  if (k > ks) k = ks;
  return (int)(k & ~1L);
}
template <typename T> constexpr int bwd_ksl(int H, bool db) {
  const int tiles = H / 16, ks = 4 * H / LT<T>::KV;
  const long gbuf = (db ? 2L : 1L) * 16 * (4 * H + LT<T>::PAD) * (long)sizeof(T);
  long k = (150L * 1024 - gbuf) / (tiles * 1024L);
  if (k < 0) k = 0;
  if (k > ks) k = ks;
  return (int)(k & ~1L);
}

// Debug phase timers (env ASR_LSTM_DBG=1): per (direction, wave) sums of s_memtime deltas
// [top, lds-mfma, stream-mfma, gate-math, barrier-wait] over all steps of the forward kernel.
__device__ unsigned long long* g_dbg = nullptr;
#define DBG_T() (dbg ? (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime()) : 0ull)

// ---------------------------------------------------------------- forward
template <typename T, int H, int NW, int KSL, int KSR, bool PF, bool PRIO>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_kernel(
    int T_, int B_, int ndir, const f32x4_t* __restrict__ xg, const T* __restrict__ whp,
    const float* __restrict__ peep, const int32_t* __restrict__ seq_len, float forget_bias,
    float cell_clip, typename LT<T>::g4_t* __restrict__ gates, T* __restrict__ hout,
    float* __restrict__ cs, float* __restrict__ c_final, float* __restrict__ h_final) {
  constexpr int NUB = H / (16 * NW);
  constexpr int KV = LT<T>::KV, KS = H / KV;
  constexpr int LDH = H + LT<T>::PAD;
  constexpr int E = 16 / (int)sizeof(T);
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* hs = reinterpret_cast<T*>(smem);                       // [2][16][LDH]
  T* wl = hs + 2 * 16 * LDH;                                // [tile][KSL][64][E]  (LDS-resident W_h)

  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool rev = (d == 1);
  const T* wp = whp + (size_t)d * H * 4 * H;
  const unsigned jw = wave * 16 + col;                      // unit of u = 0; u adds NW*16

  int len[4];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) len[r] = seq_len[b0 + rg * 4 + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  float c[NUB][4], hr[NUB][4], wci[NUB], wcf[NUB], wco[NUB];
#pragma unroll
  for (int u = 0; u < NUB; ++u) {
    const int j = jw + NW * 16 * u;
    wci[u] = peep ? peep[(d * 3 + 0) * H + j] : 0.f;
    wcf[u] = peep ? peep[(d * 3 + 1) * H + j] : 0.f;
    wco[u] = peep ? peep[(d * 3 + 2) * H + j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[u][r] = hr[u][r] = 0.f;
  }
  for (int i = threadIdx.x; i < 2 * 16 * LDH; i += NW * 64) hs[i] = T(0);
  if (KSL > 0) {   // park the first KSL k-chunks of every tile in LDS (fragment order)
    constexpr int NT = (H / 16) * 4;
    for (int f = threadIdx.x; f < NT * KSL * 64; f += NW * 64) {
      const int l = f & 63, fk = (f >> 6) % KSL, tile = (f >> 6) / KSL;
      const frag_t v = *reinterpret_cast<const frag_t*>(wp + (((size_t)tile * KS + fk) * 64 + l) * E);
      *reinterpret_cast<frag_t*>(wl + (size_t)f * E) = v;
    }
  }
  __syncthreads();

  // element offset (in units) of row (t, b) of this direction: ((t*B + b)*ndir + d)*H
  auto row_off = [&](int t, int brow) -> unsigned {
    return ((unsigned)(t * B_ + b0 + brow) * ndir + d) * H + jw;
  };
  // x W_x + b of a step (one 16-B load per pair) goes straight into the MFMA accumulators.  The
  // loads for step s+1 are issued at the END of step s's gate math (the accumulators are dead
  // by then), so their latency hides under the barrier wait and the next step's LDS reads.
  f32x4_t acc[NUB][4];
  auto load_x = [&](int s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = s < len[r];
      const int t = act ? (rev ? len[r] - 1 - s : s) : 0;
      const unsigned off = row_off(t, rg * 4 + r);
#pragma unroll
      for (int u = 0; u < NUB; ++u) {
        // no select on the loaded value (it would force the wait here): rows that are inactive
        // at step s read some valid row and their results are discarded by the act-selects below
        const f32x4_t v = xg[off + NW * 16 * u];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][q][r] = v[q];
      }
    }
  };
  if (tmax > 0) load_x(0);
  // k-chunks [KSL, KSL+KSR) of this wave's tiles stay in REGISTERS for the whole launch
  frag_t breg[NUB][4][KSR > 0 ? KSR : 1];
#pragma unroll
  for (int u = 0; u < NUB; ++u)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < KSR; ++k) {
        const int tile = (wave + NW * u) * 4 + q;
        breg[u][q][k] = *reinterpret_cast<const frag_t*>(wp + (((size_t)tile * KS + KSL + k) * 64 + lane) * E);
      }
  constexpr int KSS = KS - KSL - KSR;          // streamed chunks
  constexpr int KSP = (PF && KSS > 0) ? ((KSS < 2 || KSR > 0) ? 1 : 2) : 0;   // of which issued at the top of the step
  unsigned long long* dbg = g_dbg;
  unsigned long long ph[5] = {0, 0, 0, 0, 0};

  for (int s = 0; s < tmax; ++s) {
    const unsigned long long t0 = DBG_T();
    const T* hcur = hs + (s & 1) * 16 * LDH;
    T* hnxt = hs + ((s + 1) & 1) * 16 * LDH;
    // streamed chunks: the first KSP are requested now and consumed after the resident chunks
    frag_t bst[NUB][4][KSP > 0 ? KSP : 1];
#pragma unroll
    for (int k = 0; k < KSP; ++k)
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int tile = (wave + NW * u) * 4 + q;
          bst[u][q][k] = *reinterpret_cast<const frag_t*>(wp + (((size_t)tile * KS + KSL + KSR + k) * 64 + lane) * E);
        }

    if (PRIO) mfma_phase_prio(wave);
    const unsigned long long t1 = DBG_T();
    // LDS-resident k-chunks
#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) {
      const frag_t a = *reinterpret_cast<const frag_t*>(hcur + col * LDH + ks * KV + rg * E);
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int tile = (wave + NW * u) * 4 + q;
          const frag_t b = *reinterpret_cast<const frag_t*>(wl + (((size_t)tile * KSL + ks) * 64 + lane) * E);
          acc[u][q] = mma_chunk(a, b, acc[u][q]);
        }
    }
    // register-resident k-chunks
#pragma unroll
    for (int k = 0; k < KSR; ++k) {
      const frag_t a = *reinterpret_cast<const frag_t*>(hcur + col * LDH + (KSL + k) * KV + rg * E);
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][q] = mma_chunk(a, breg[u][q][k], acc[u][q]);
    }
    if (dbg) {
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[u][q]));
    }
    const unsigned long long t2 = DBG_T();
    // streamed k-chunks requested at the top of the step
#pragma unroll
    for (int k = 0; k < KSP; ++k) {
      const frag_t a = *reinterpret_cast<const frag_t*>(hcur + col * LDH + (KSL + KSR + k) * KV + rg * E);
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][q] = mma_chunk(a, bst[u][q][k], acc[u][q]);
    }
    // remaining streamed k-chunks (limited unroll: each chunk keeps 4*NUB 16-B loads in flight)
#pragma unroll 2
    for (int ks = KSL + KSR + KSP; ks < KS; ++ks) {
      const frag_t a = *reinterpret_cast<const frag_t*>(hcur + col * LDH + ks * KV + rg * E);
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int tile = (wave + NW * u) * 4 + q;
          const frag_t b = *reinterpret_cast<const frag_t*>(wp + (((size_t)tile * KS + ks) * 64 + lane) * E);
          acc[u][q] = mma_chunk(a, b, acc[u][q]);
        }
    }

    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (dbg) {
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[u][q]));
    }
    const unsigned long long t3 = DBG_T();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int brow = rg * 4 + r;
      const bool act = s < len[r];
      const int t = act ? (rev ? len[r] - 1 - s : s) : s;   // inactive: frame s is a padded frame
      const unsigned off = row_off(t, brow);
#pragma unroll
      for (int u = 0; u < NUB; ++u) {
        const float cprev = c[u][r];
        const float ig = fsig(acc[u][0][r] + wci[u] * cprev);
        const float gg = ftanh(acc[u][1][r]);
        const float fg = fsig(acc[u][2][r] + forget_bias + wcf[u] * cprev);
        float cn = gg * ig + cprev * fg;
        if (cell_clip > 0.f) cn = fminf(fmaxf(cn, -cell_clip), cell_clip);
        const float og = fsig(acc[u][3][r] + wco[u] * cn);
        const float hn = ftanh(cn) * og;
        c[u][r] = act ? cn : cprev;
        hr[u][r] = act ? hn : hr[u][r];
        const unsigned o = off + NW * 16 * u;
        if (act) {
          gates[o] = LT<T>::pack(ig, gg, fg, og);
          cs[o] = cn;
        }
        hout[o] = LT<T>::cvt(act ? hn : 0.f);
        hnxt[brow * LDH + jw + NW * 16 * u] = LT<T>::cvt(hr[u][r]);
      }
    }
    if (s + 1 < tmax) load_x(s + 1);
    const unsigned long long t4 = DBG_T();
    __syncthreads();
    if (dbg) {
      const unsigned long long t5 = DBG_T();
      ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4;
    }
  }
  if (dbg && lane == 0 && blockIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) dbg[((size_t)d * 16 + wave) * 8 + k] = ph[k];
    dbg[((size_t)d * 16 + wave) * 8 + 5] = tmax;
  }
  // zero-fill the common padded tail [tmax, T)
  for (int s = tmax; s < T_; ++s)
#pragma unroll
    for (int u = 0; u < NUB; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) hout[row_off(s, rg * 4 + r) + NW * 16 * u] = T(0);
#pragma unroll
  for (int u = 0; u < NUB; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = ((size_t)d * B_ + b0 + rg * 4 + r) * H + jw + NW * 16 * u;
      if (c_final) c_final[o] = c[u][r];
      if (h_final) h_final[o] = hr[u][r];
    }
}

// ---------------------------------------------------------------- backward (BPTT)
// TF's LSTMBlockCellGrad restated (cell_clip is not part of the gradient op):
//   do = dh*tanh(c)*o(1-o);  dc = dc_rec + dh*o*(1-tanh(c)^2) + do*wco
//   dci = dc*i*(1-ci^2); di = dc*ci*i(1-i); df = dc*c_prev*f(1-f)
//   dc_prev = dc*f + di*wci + df*wcf;  dh_prev = [di dci df do] W_h^T
template <typename T, int H, bool DB, int NW, int KSL, bool PRIO>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dhout,
    const typename LT<T>::g4_t* __restrict__ gates, const float* __restrict__ cs,
    const T* __restrict__ whpb, const float* __restrict__ peep, const int32_t* __restrict__ seq_len,
    const float* __restrict__ d_c_final, const float* __restrict__ d_h_final,
    typename LT<T>::g4_t* __restrict__ dgates, float* __restrict__ dpeep_part, float clipz) {
  constexpr int NUB = H / (16 * NW);
  constexpr int KV = LT<T>::KV, KS = 4 * H / KV;
  constexpr int LDG = 4 * H + LT<T>::PAD;
  constexpr int E = 16 / (int)sizeof(T);
  typedef typename Frag<T>::type frag_t;
  typedef typename LT<T>::g4_t g4_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* dgs = reinterpret_cast<T*>(smem);                      // [DB?2:1][16][LDG], k' = j*4+q
  T* wl = dgs + (DB ? 2 : 1) * 16 * LDG;                    // [ub][KSL][64][E]

  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool rev = (d == 1);
  const T* wp = whpb + (size_t)d * H * 4 * H;
  const unsigned jw = wave * 16 + col;

  int len[4];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) len[r] = seq_len[b0 + rg * 4 + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  auto row_off = [&](int t, int brow) -> unsigned {
    return ((unsigned)(t * B_ + b0 + brow) * ndir + d) * H + jw;
  };
  auto frame = [&](int s, int r) -> int { return rev ? len[r] - 1 - s : s; };

  float dhr[NUB][4], dcr[NUB][4], cc[NUB][4], wci[NUB], wcf[NUB], wco[NUB];
  float pwi[NUB], pwf[NUB], pwo[NUB];
  float sbi[NUB], sbg[NUB], sbf[NUB], sbo[NUB];   // bias gradient = column sums of dgates
#pragma unroll
  for (int u = 0; u < NUB; ++u) {
    const int j = jw + NW * 16 * u;
    wci[u] = peep ? peep[(d * 3 + 0) * H + j] : 0.f;
    wcf[u] = peep ? peep[(d * 3 + 1) * H + j] : 0.f;
    wco[u] = peep ? peep[(d * 3 + 2) * H + j] : 0.f;
    pwi[u] = pwf[u] = pwo[u] = 0.f;
    sbi[u] = sbg[u] = sbf[u] = sbo[u] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = ((size_t)d * B_ + b0 + rg * 4 + r) * H + j;
      dhr[u][r] = d_h_final ? d_h_final[o] : 0.f;
      dcr[u][r] = d_c_final ? d_c_final[o] : 0.f;
      // c of the first processed step (s = tmax-1) for rows active there
      const bool a0 = tmax > 0 && tmax - 1 < len[r];
      cc[u][r] = a0 ? cs[row_off(frame(tmax - 1, r), rg * 4 + r) + NW * 16 * u] : 0.f;
    }
  }
  if (KSL > 0) {
    constexpr int NT = H / 16;
    for (int f = threadIdx.x; f < NT * KSL * 64; f += NW * 64) {
      const int l = f & 63, fk = (f >> 6) % KSL, tile = (f >> 6) / KSL;
      const frag_t v = *reinterpret_cast<const frag_t*>(wp + (((size_t)tile * KS + fk) * 64 + l) * E);
      *reinterpret_cast<frag_t*>(wl + (size_t)f * E) = v;
    }
  }
  const g4_t gzero = LT<T>::pack(0.f, 0.f, 0.f, 0.f);
  // zero-fill the common padded tail frames [tmax, T)
  for (int s = T_ - 1; s >= tmax; --s)
#pragma unroll
    for (int u = 0; u < NUB; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) dgates[row_off(s, rg * 4 + r) + NW * 16 * u] = gzero;
  __syncthreads();

  for (int s = tmax - 1; s >= 0; --s) {
    T* dcur = dgs + (DB ? (s & 1) : 0) * 16 * LDG;
    // ---- dh_rec of this step = dG(step s+1) W_h^T, carried through inactive rows
    if (s != tmax - 1) {
      const T* dprev = dgs + (DB ? ((s + 1) & 1) : 0) * 16 * LDG;
      f32x4_t acc[NUB];
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[u][r] = dhr[u][r];  // carry (inactive) or 0 (active)
      if (PRIO) mfma_phase_prio(wave);
#pragma unroll
      for (int ks = 0; ks < KSL; ++ks) {
        const frag_t a = *reinterpret_cast<const frag_t*>(dprev + col * LDG + ks * KV + rg * E);
#pragma unroll
        for (int u = 0; u < NUB; ++u) {
          const int tile = wave + NW * u;
          const frag_t b = *reinterpret_cast<const frag_t*>(wl + (((size_t)tile * KSL + ks) * 64 + lane) * E);
          acc[u] = mma_chunk(a, b, acc[u]);
        }
      }
#pragma unroll 2
      for (int ks = KSL; ks < KS; ++ks) {
        const frag_t a = *reinterpret_cast<const frag_t*>(dprev + col * LDG + ks * KV + rg * E);
#pragma unroll
        for (int u = 0; u < NUB; ++u) {
          const int tile = wave + NW * u;
          const frag_t b = *reinterpret_cast<const frag_t*>(wp + (((size_t)tile * KS + ks) * 64 + lane) * E);
          acc[u] = mma_chunk(a, b, acc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) dhr[u][r] = acc[u][r];
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      if (!DB) __syncthreads();  // all reads of dgs done before it is overwritten
    }
    // ---- gate gradients (predicated, no divergent control flow)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int brow = rg * 4 + r;
      const bool act = s < len[r];
      const bool hasp = act && s > 0;
      // the row's c(s-1): needed as c_prev now (if active) and as "current c" next iteration
      const bool ldp = (s > 0) && (s - 1 < len[r]);
      const unsigned offn = row_off(ldp ? frame(s - 1, r) : 0, brow);
      // inactive rows: frame s is a padded frame -> zero gradient there; loads use any valid row
      const unsigned off = act ? row_off(frame(s, r), brow) : row_off(s, brow);
      const unsigned offl = act ? off : offn;
#pragma unroll
      for (int u = 0; u < NUB; ++u) {
        const unsigned o = off + NW * 16 * u;
        float i, g, f, oo;
        LT<T>::unpack(gates[offl + NW * 16 * u], i, g, f, oo);
        const float cpv = cs[offn + NW * 16 * u];
        const float cprev = hasp ? cpv : 0.f;
        const float dho = dhout[offl + NW * 16 * u];
        const float cur = cc[u][r];
        const float dh = dho + dhr[u][r];
        const float tc = ftanh(cur);
        const float d_o = dh * tc * oo * (1.f - oo);
        const float dct = dcr[u][r] + dh * oo * (1.f - tc * tc) + d_o * wco[u];
        // clipz > 0 (asr_lstm_bwd_ex): cs holds the CLAMPED state, |c| >= clip marks a state that passes nothing back
        const float dc = (clipz > 0.f && fabsf(cur) >= clipz) ? 0.f : dct;
        const float d_g = dc * i * (1.f - g * g);
        const float d_i = dc * g * i * (1.f - i);
        const float d_f = dc * cprev * f * (1.f - f);
        dcr[u][r] = act ? (dc * f + d_i * wci[u] + d_f * wcf[u]) : dcr[u][r];
        dhr[u][r] = act ? 0.f : dhr[u][r];   // active: the next MFMA supplies dh_prev
        const float zi = act ? d_i : 0.f, zg = act ? d_g : 0.f, zf = act ? d_f : 0.f, zo = act ? d_o : 0.f;
        pwi[u] += zi * cprev;
        pwf[u] += zf * cprev;
        pwo[u] += zo * cur;
        sbi[u] += zi; sbg[u] += zg; sbf[u] += zf; sbo[u] += zo;
        cc[u][r] = ldp ? cpv : 0.f;          // c(s-1): this row's "current c" in the next iteration
        const g4_t pk = LT<T>::pack(zi, zg, zf, zo);
        dgates[o] = pk;
        *reinterpret_cast<g4_t*>(dcur + brow * LDG + (jw + NW * 16 * u) * 4) = pk;
      }
    }
    __syncthreads();
  }

  // per-tile partials of the peephole gradients (3 x H) and the bias gradient (4 x H):
  // sum over the 4 row groups sharing a unit column, one row of 7*H floats per (tile, direction)
  if (dpeep_part) {
#pragma unroll
    for (int u = 0; u < NUB; ++u) {
      float v[7] = {pwi[u], pwf[u], pwo[u], sbi[u], sbg[u], sbf[u], sbo[u]};
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        v[k] += __shfl_xor(v[k], 16, 64);
        v[k] += __shfl_xor(v[k], 32, 64);
      }
      if (rg == 0) {
        const int j = jw + NW * 16 * u;
        float* p = dpeep_part + ((size_t)blockIdx.x * ndir + d) * 7 * H;
#pragma unroll
        for (int k = 0; k < 7; ++k) p[k * H + j] = v[k];
      }
    }
  }
}

__global__ void reduce_tiles_kernel(const float* __restrict__ part, int ntiles, int n,
                                    float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int t = 0; t < ntiles; ++t) a += part[(size_t)t * n + i];
  out[i] = a;
}

static unsigned long long* g_dbg_host_ptr = nullptr;
static void dbg_setup() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* e = getenv("ASR_LSTM_DBG");
  if (!(e && e[0] == '1')) return;
  (void)hipMalloc(&g_dbg_host_ptr, 2 * 16 * 8 * sizeof(unsigned long long));
  (void)hipMemset(g_dbg_host_ptr, 0, 2 * 16 * 8 * sizeof(unsigned long long));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &g_dbg_host_ptr, sizeof(g_dbg_host_ptr));
}
// experiment switch (env ASR_LSTM_PRIO=0 disables the staggered MFMA-phase priorities)
static const bool g_lstm_prio = [] { const char* e = getenv("ASR_LSTM_PRIO"); return (e && e[0] == '1'); }();

// Per (dtype, H) geometry of the forward kernel: waves per workgroup, k-chunks of W_h resident
// in LDS (KSL) and in registers (KSR); the remaining KS-KSL-KSR chunks are streamed every step.
template <typename T, int H> struct FwdCfg {
  static constexpr int NW = pick_nw(H);
  static constexpr int KSL = fwd_ksl<T>(H);
  static constexpr int KSR = 0;
};
// headline shape (5x256 bf16).  Measured with the in-kernel phase timers (scripts/probe_lstm_phases.py,
// cycles per step): 16 waves, 2 chunks in LDS, 6 streamed = 11.7k (streamed phase 6.6k = 58 B/clk,
// the per-CU L2 limit); 8 waves x 256 VGPRs with 4 more chunks in registers = 12.5k-15k (two waves
// per SIMD cannot hide the gate-math latency and the allocation spills) -> 16 waves it is.
template <> struct FwdCfg<bf16_t, 256> {
  static constexpr int NW = 16;
  static constexpr int KSL = 2;
  static constexpr int KSR = 0;
};

template <typename T, int H>
int launch_fwd(int T_, int B, int ndir, const float* xg, const void* whp, const float* peep,
               const int32_t* seq_len, float fb, float clip, void* gates, void* hout, float* cs,
               float* cf, float* hf, hipStream_t st) {
  dbg_setup();
  constexpr int NW = FwdCfg<T, H>::NW;
  constexpr int KSL = FwdCfg<T, H>::KSL;
  constexpr int KSR = FwdCfg<T, H>::KSR;
  const size_t lds = (size_t)2 * 16 * (H + LT<T>::PAD) * sizeof(T) + (size_t)(H / 16) * 4 * KSL * 1024;
  constexpr bool PF = true;   // request the first streamed chunks at the top of the step
  auto k = g_lstm_prio ? lstm_fwd_kernel<T, H, NW, KSL, KSR, PF, true> : lstm_fwd_kernel<T, H, NW, KSL, KSR, PF, false>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(B / 16, ndir), dim3(NW * 64), lds, st, T_, B, ndir, (const f32x4_t*)xg,
                     (const T*)whp, peep, seq_len, fb, clip, (typename LT<T>::g4_t*)gates, (T*)hout, cs, cf, hf);
  return 0;
}

template <typename T, int H>
int launch_bwd(int T_, int B, int ndir, const float* dhout, const void* gates, const float* cs,
               const void* whpb, const float* peep, const int32_t* seq_len, const float* dcf,
               const float* dhf, void* dgates, float* dpeep_part, float clipz, hipStream_t st) {
  constexpr size_t one = (size_t)16 * (4 * H + LT<T>::PAD) * sizeof(T);
  constexpr bool DB = (2 * one <= 72 * 1024);
  constexpr int NW = pick_nw(H);
  constexpr int KSL = bwd_ksl<T>(H, DB);
  const size_t lds = (DB ? 2 * one : one) + (size_t)(H / 16) * KSL * 1024;
  auto k = g_lstm_prio ? lstm_bwd_kernel<T, H, DB, NW, KSL, true> : lstm_bwd_kernel<T, H, DB, NW, KSL, false>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(B / 16, ndir), dim3(NW * 64), lds, st, T_, B, ndir, dhout,
                     (const typename LT<T>::g4_t*)gates, cs, (const T*)whpb, peep, seq_len, dcf, dhf,
                     (typename LT<T>::g4_t*)dgates, dpeep_part, clipz);
  return 0;
}

}  // namespace

#define ASR_H_DISPATCH(H_, T_, CALL)            \
  switch (H_) {                                 \
    case 64:  { constexpr int HH = 64;  CALL; } break;  \
    case 128: { constexpr int HH = 128; CALL; } break;  \
    case 192: { constexpr int HH = 192; CALL; } break;  \
    case 256: { constexpr int HH = 256; CALL; } break;  \
    case 320: { constexpr int HH = 320; CALL; } break;  \
    case 512: { constexpr int HH = 512; CALL; } break;  \
    default: ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "lstm: num_units %d not in {64,128,192,256,320,512}", H_); \
  }

// debug only (not part of the public header): copy the forward phase timers to the host
extern "C" int asr_debug_lstm_cycles(unsigned long long* out, int n) {
  if (!g_dbg_host_ptr || n > 2 * 16 * 8) return -1;
  return hipMemcpy(out, g_dbg_host_ptr, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

extern "C" int asr_lstm_prep_weights(asr_handle* h, int dtype, const float* kernel, const float* bias,
                                     int Din, int H, void* wx_il, float* bias_il, void* packed_fwd,
                                     void* packed_bwd, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || !kernel || !bias || !wx_il || !bias_il || !packed_fwd || !packed_bwd ||
      Din <= 0 || H <= 0 || H % 64)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_prep_weights: bad args (H=%d must be a multiple of 64)", H);
  const size_t total = (size_t)(Din + H + 1) * 4 * H;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  if (dtype == ASR_F32)
    hipLaunchKernelGGL(prep_weights_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, kernel, bias,
                       Din, H, (float*)wx_il, bias_il, (float*)packed_fwd, (float*)packed_bwd);
  else
    hipLaunchKernelGGL(prep_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, kernel, bias,
                       Din, H, (bf16_t*)wx_il, bias_il, (bf16_t*)packed_fwd, (bf16_t*)packed_bwd);
  ASR_CHECK_LAUNCH(h, "asr_lstm_prep_weights");
  return ASR_OK;
}

extern "C" int asr_gate_deinterleave(asr_handle* h, const float* in, int ld_in, float* out, int ld_out,
                                     int rows, int H, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!in || !out || rows < 0 || H <= 0 || ld_in < 4 * H || ld_out < 4 * H)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_gate_deinterleave: bad args");
  const size_t total = (size_t)rows * 4 * H;
  if (!total) return ASR_OK;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(deinterleave_cols_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, in, ld_in, out,
                     ld_out, rows, H);
  ASR_CHECK_LAUNCH(h, "asr_gate_deinterleave");
  return ASR_OK;
}

extern "C" int asr_lstm_prep_layer(asr_handle* h, int dtype, int ndir, const float* const* vars, int Din, int ldk,
                                   int H, void* wxT, void* wx_cat, float* bias_cat, void* packed_fwd,
                                   void* packed_bwd, float* peep_out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || (ndir != 1 && ndir != 2) || !vars || !wxT || !wx_cat || !bias_cat || !packed_fwd ||
      !packed_bwd || Din <= 0 || ldk < Din || H <= 0 || H % 64)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_prep_layer: bad args (H=%d must be a multiple of 64, ldk=%d >= Din=%d)", H, ldk, Din);
  PrepVars v;
  memset(&v, 0, sizeof(v));
  for (int d = 0; d < ndir; ++d) {
    v.kernel[d] = vars[d * 5 + 0];
    v.bias[d] = vars[d * 5 + 1];
    for (int k = 0; k < 3; ++k) v.peep[d][k] = vars[d * 5 + 2 + k];
    if (!v.kernel[d] || !v.bias[d]) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_prep_layer: kernel / bias of direction %d missing", d);
    if (peep_out && (!v.peep[d][0] || !v.peep[d][1] || !v.peep[d][2]))
      ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_prep_layer: peep_out given but a peephole vector of direction %d is missing", d);
  }
  const size_t total = (size_t)ndir * ((size_t)4 * H * ldk + (size_t)(Din + H + 1) * 4 * H + (peep_out ? 3 * H : 0));
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == ASR_F32)
    hipLaunchKernelGGL(prep_layer_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, v, ndir, Din, ldk, H,
                       (float*)wxT, (float*)wx_cat, bias_cat, (float*)packed_fwd, (float*)packed_bwd, peep_out);
  else
    hipLaunchKernelGGL(prep_layer_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, v, ndir, Din, ldk, H,
                       (bf16_t*)wxT, (bf16_t*)wx_cat, bias_cat, (bf16_t*)packed_fwd, (bf16_t*)packed_bwd, peep_out);
  ASR_CHECK_LAUNCH(h, "asr_lstm_prep_layer");
  return ASR_OK;
}

extern "C" int asr_lstm_grad_finish(asr_handle* h, int ndir, float* const* grads, int rows, int H,
                                    const float* dw_il, const float* dpeep_dbias, int has_peep, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if ((ndir != 1 && ndir != 2) || !grads || !dw_il || !dpeep_dbias || rows <= 0 || H <= 0)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_grad_finish: bad args");
  GradVars v;
  memset(&v, 0, sizeof(v));
  for (int d = 0; d < ndir; ++d) {
    v.kernel[d] = grads[d * 5 + 0];
    v.bias[d] = grads[d * 5 + 1];
    for (int k = 0; k < 3; ++k) v.peep[d][k] = grads[d * 5 + 2 + k];
    if (!v.kernel[d] || !v.bias[d] || (has_peep && (!v.peep[d][0] || !v.peep[d][1] || !v.peep[d][2])))
      ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_grad_finish: a gradient pointer of direction %d is missing", d);
  }
  const size_t total = (size_t)ndir * ((size_t)(rows + 1) * 4 * H + (has_peep ? 3 * H : 0));
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, v, ndir, rows, H, dw_il,
                     dpeep_dbias, has_peep);
  ASR_CHECK_LAUNCH(h, "asr_lstm_grad_finish");
  return ASR_OK;
}

extern "C" int asr_lstm_fwd(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                            const float* xproj, const void* wh_packed, const float* peep,
                            const int32_t* seq_len, float forget_bias, float cell_clip, void* gates,
                            void* hout, float* cs, float* c_final, float* h_final, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || T < 0 || B <= 0 || B % 16 || (ndir != 1 && ndir != 2) || !xproj ||
      !wh_packed || !seq_len || !gates || !hout || !cs)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_fwd: bad args (B=%d must be a multiple of 16, ndir=%d)", B, ndir);
  if ((double)T * B * ndir * H >= 4294967295.0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_lstm_fwd: T*B*ndir*H exceeds 32-bit element offsets");
  if (T == 0) return ASR_OK;
  hipStream_t st = (hipStream_t)s;
  // multi-CU form (W_h slices LDS-resident, per-step h all-gather between 4 CUs): lstm_cluster.hip
  if (dtype == ASR_BF16 && asr_cluster_fwd_try(h, T, B, H, ndir, xproj, wh_packed, peep, seq_len, forget_bias,
                                               cell_clip, gates, hout, cs, c_final, h_final, st)) {
    ASR_CHECK_LAUNCH(h, "asr_lstm_fwd(cluster)");
    return ASR_OK;
  }
  if (dtype == ASR_F32 && asr_cluster_fwd_f32_try(h, T, B, H, ndir, xproj, wh_packed, peep, seq_len, forget_bias,
                                                  cell_clip, gates, hout, cs, c_final, h_final, st)) {
    ASR_CHECK_LAUNCH(h, "asr_lstm_fwd(cluster, fp32)");
    return ASR_OK;
  }
  if (dtype == ASR_F32) {
    ASR_H_DISPATCH(H, T, (launch_fwd<float, HH>(T, B, ndir, xproj, wh_packed, peep, seq_len, forget_bias,
                                                cell_clip, gates, hout, cs, c_final, h_final, st)));
  } else {
    ASR_H_DISPATCH(H, T, (launch_fwd<bf16_t, HH>(T, B, ndir, xproj, wh_packed, peep, seq_len, forget_bias,
                                                 cell_clip, gates, hout, cs, c_final, h_final, st)));
  }
  ASR_CHECK_LAUNCH(h, "asr_lstm_fwd");
  return ASR_OK;
}

static int lstm_bwd_impl(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                         const float* dhout, const void* gates, const float* cs,
                         const void* wh_packed_bwd, const float* peep, const int32_t* seq_len,
                         const float* d_c_final, const float* d_h_final, float clipz, void* dgates,
                         float* dpeep, float* dpeep_workspace, asr_stream s) {
  if (!asr_dtype_ok(dtype) || T < 0 || B <= 0 || B % 16 || (ndir != 1 && ndir != 2) || !dhout ||
      !gates || !cs || !wh_packed_bwd || !seq_len || !dgates)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_bwd: bad args (B=%d must be a multiple of 16, ndir=%d)", B, ndir);
  if (dpeep && !dpeep_workspace)
    ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_lstm_bwd: dpeep_dbias needs a workspace of (B/16)*ndir*7*H floats");
  if ((double)T * B * ndir * H >= 4294967295.0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_lstm_bwd: T*B*ndir*H exceeds 32-bit element offsets");
  hipStream_t st = (hipStream_t)s;
  if (T == 0) {
    if (dpeep) (void)hipMemsetAsync(dpeep, 0, sizeof(float) * ndir * 7 * H, st);
    return ASR_OK;
  }
  // one 16-utterance tile: the tile partials [tile][ndir][7][H] ARE the result, no reduction launch
  const bool single_tile = (B == 16);
  float* part = dpeep ? (single_tile ? dpeep : dpeep_workspace) : nullptr;
  bool launched = false;
  if (dtype == ASR_BF16 && asr_cluster_bwd_try(h, T, B, H, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len,
                                               d_c_final, d_h_final, dgates, part, st)) {
    launched = true;
  } else
  if (dtype == ASR_F32 && asr_cluster_bwd_f32_try(h, T, B, H, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len,
                                                  d_c_final, d_h_final, dgates, part, st)) {
    launched = true;
  } else
  if (dtype == ASR_F32) {
    ASR_H_DISPATCH(H, T, (launch_bwd<float, HH>(T, B, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len,
                                                d_c_final, d_h_final, dgates, part, clipz, st)));
  } else {
    ASR_H_DISPATCH(H, T, (launch_bwd<bf16_t, HH>(T, B, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len,
                                                 d_c_final, d_h_final, dgates, part, clipz, st)));
  }
  ASR_CHECK_LAUNCH(h, "asr_lstm_bwd");
  if (dpeep && !single_tile) {
    const int n = ndir * 7 * H;
    hipLaunchKernelGGL(reduce_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, st, part, B / 16, n, dpeep);
    ASR_CHECK_LAUNCH(h, "asr_lstm_bwd(reduce)");
  }
  return ASR_OK;
}

// clip_no_grad > 0: the forward ran with cell_clip = clip_no_grad and the cell is tf.contrib.rnn.LSTMCell, whose
// tf.clip_by_value passes no gradient through a clamped state (LSTMBlockCell's fused gradient op does not know the
// clip: asr_lstm_bwd, clip_no_grad = 0).
extern "C" int asr_lstm_bwd_ex(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                               const float* dhout, const void* gates, const float* cs,
                               const void* wh_packed_bwd, const float* peep, const int32_t* seq_len,
                               const float* d_c_final, const float* d_h_final, float clip_no_grad, void* dgates,
                               float* dpeep, float* dpeep_workspace, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!(clip_no_grad >= 0.f)) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_bwd_ex: clip_no_grad must be >= 0");
  h->bptt_clip = clip_no_grad;   // the cluster launchers (lstm_cluster.hip) read it
  const int rc = lstm_bwd_impl(h, dtype, T, B, H, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len, d_c_final,
                               d_h_final, clip_no_grad, dgates, dpeep, dpeep_workspace, s);
  h->bptt_clip = 0.f;
  return rc;
}

extern "C" int asr_lstm_bwd(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                            const float* dhout, const void* gates, const float* cs,
                            const void* wh_packed_bwd, const float* peep, const int32_t* seq_len,
                            const float* d_c_final, const float* d_h_final, void* dgates,
                            float* dpeep, float* dpeep_workspace, asr_stream s) {
  return asr_lstm_bwd_ex(h, dtype, T, B, H, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len, d_c_final, d_h_final,
                         0.f, dgates, dpeep, dpeep_workspace, s);
}
