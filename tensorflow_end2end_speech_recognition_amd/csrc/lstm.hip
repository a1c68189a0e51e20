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
//     runs ALL T steps;
//     the 16 utterances are the M dimension of a 16x16 MFMA tile, so a lane of the C/D
//     fragment permanently owns (utterance b = (lane>>4)*4+r, unit j = ub*16+(lane&15)):
//     c, h, the peepholes and the four gate pre-activations of that (b,j) never leave
//     its registers -> the gate math needs no cross-lane traffic.
//   * wave w owns unit blocks ub = w, w+NW, ...; for each it accumulates the four gate
//     tiles (i, ci, f, o) so one lane ends a step with all four gates of its (b,j).
//   * h_{t-1} (16 x H) lives in LDS (double buffered, one barrier per step) as the MFMA
//     A operand; W_h is pre-packed in B-fragment order so every wave-load is one
//     contiguous 1 KiB line; as many k-chunks as fit are parked in LDS for the whole
//     launch, the rest is streamed from L2 each step.
//   * x W_x + b for step s+1 is prefetched into registers while step s computes; the
//     same buffer is overwritten in place with the post-activation gates for BPTT.
#include "common.h"

namespace {

template <typename T> struct LT;
template <> struct LT<float> {
  static constexpr int KV = 16;  // k covered by one packed 16-B fragment load (4 MFMAs of K=4)
  static constexpr int PAD = 4;
};
template <> struct LT<bf16_t> {
  static constexpr int KV = 32;  // one 16x16x32 MFMA
  static constexpr int PAD = 8;
};

// ---------------------------------------------------------------- weight packing
// fwd: B[k][n] = Wh[k][q*H + ub*16 + n]            tiles (ub,q), k-chunks of KV
// bwd: B[k][n] = Wh[ub*16 + n][k]   (k over 4H)    tiles ub,     k-chunks of KV
template <typename T>
__global__ void pack_wh_kernel(const float* __restrict__ wh, int H, T* __restrict__ pf,
                               T* __restrict__ pb) {
  constexpr int KV = LT<T>::KV;
  constexpr int E = 16 / sizeof(T);      // elements per lane per fragment load
  const size_t total = (size_t)H * 4 * H;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int e = idx % E;
    const int lane = (idx / E) % 64;
    const size_t frag = idx / (E * 64);
    const int n = lane & 15, rg = lane >> 4;
    // element e of lane (n, rg) in a k-chunk: bf16 k = rg*8+e ; fp32 k = rg*4+e
    const int kin = rg * E + e;
    {  // forward
      const int KS = H / KV;
      const int ks = frag % KS;
      const int tile = frag / KS;  // ub*4 + q
      const int q = tile & 3, ub = tile >> 2;
      const int k = ks * KV + kin;
      pf[idx] = Elem<T>::from_f32(wh[(size_t)k * 4 * H + q * H + ub * 16 + n]);
    }
    {  // backward
      const int KS = 4 * H / KV;
      const int ks = frag % KS;
      const int ub = frag / KS;
      const int k = ks * KV + kin;
      pb[idx] = Elem<T>::from_f32(wh[(size_t)(ub * 16 + n) * 4 * H + k]);
    }
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
// expf + IEEE division.  The recurrence is VALU-issue/latency bound, so this
// and the wave count below are what set the step time, not HBM or the MFMA pipe.
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

// waves per workgroup: the largest divisor of H/16 (unit blocks) that is <= 16, so that every
// wave owns NUB = H/(16*NW) unit blocks and the CU runs up to 4 waves per SIMD.
constexpr int pick_nw(int H) {
  int nb = H / 16, best = 1;
  for (int w = 1; w <= 16; ++w)
    if (nb % w == 0) best = w;
  return best;
}
// k-chunks of every (unit block, gate) tile of W_h parked in LDS for the whole launch
// (the rest is streamed from L2 every step); sized to leave room for the h / dG buffers.
template <typename T> constexpr int fwd_ksl(int H) {
  const int tiles = (H / 16) * 4, ks = H / LT<T>::KV;
  const long hbuf = 2L * 16 * (H + LT<T>::PAD) * (long)sizeof(T);
  long k = (150L * 1024 - hbuf) / (tiles * 1024L);
  if (k < 0) k = 0;
  if (k > ks) k = ks;
  return (int)(k & ~1L);   // even: keeps the streamed loop's trip count a multiple of its unroll
}
template <typename T> constexpr int bwd_ksl(int H, bool db) {
  const int tiles = H / 16, ks = 4 * H / LT<T>::KV;
  const long gbuf = (db ? 2L : 1L) * 16 * (4 * H + LT<T>::PAD) * (long)sizeof(T);
  long k = (150L * 1024 - gbuf) / (tiles * 1024L);
  if (k < 0) k = 0;
  if (k > ks) k = ks;
  return (int)(k & ~1L);   // even: keeps the streamed loop's trip count a multiple of its unroll
}

// ---------------------------------------------------------------- forward
template <typename T, int H, int NW, int KSL, bool PF>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_kernel(
    int T_, int B_, int ndir, float* __restrict__ xg, const T* __restrict__ whp,
    const float* __restrict__ peep, const int32_t* __restrict__ seq_len, float forget_bias,
    float cell_clip, T* __restrict__ hout, float* __restrict__ cs, float* __restrict__ c_final,
    float* __restrict__ h_final) {
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
  const int G4 = ndir * 4 * H, G1 = ndir * H;
  const T* wp = whp + (size_t)d * H * 4 * H;

  int len[4];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) len[r] = seq_len[b0 + rg * 4 + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  float c[NUB][4], hr[NUB][4], wci[NUB], wcf[NUB], wco[NUB];
#pragma unroll
  for (int u = 0; u < NUB; ++u) {
    const int j = (wave + NW * u) * 16 + col;
    wci[u] = peep ? peep[(d * 3 + 0) * H + j] : 0.f;
    wcf[u] = peep ? peep[(d * 3 + 1) * H + j] : 0.f;
    wco[u] = peep ? peep[(d * 3 + 2) * H + j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[u][r] = hr[u][r] = 0.f;
  }
  for (int i = threadIdx.x; i < 2 * 16 * LDH; i += NW * 64) hs[i] = T(0);
  // park the first KSL k-chunks of every tile in LDS (fragment order, 16 B per lane)
  if (KSL > 0) {
    constexpr int NT = (H / 16) * 4;
    for (int f = threadIdx.x; f < NT * KSL * 64; f += NW * 64) {
      const int l = f & 63, fk = (f >> 6) % KSL, tile = (f >> 6) / KSL;
      const frag_t v = *reinterpret_cast<const frag_t*>(wp + (((size_t)tile * KS + fk) * 64 + l) * E);
      *reinterpret_cast<frag_t*>(wl + (size_t)f * E) = v;
    }
  }
  __syncthreads();

  // x W_x + b of the coming step, prefetched while the current step computes
  float xn[NUB][4][4];
  auto prefetch = [&](int s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = s < len[r];
      const int t = act ? (rev ? len[r] - 1 - s : s) : 0;
      const float* row = xg + ((size_t)t * B_ + b0 + rg * 4 + r) * G4 + d * 4 * H + wave * 16 + col;
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = row[q * H + NW * 16 * u];     // padded frames hold finite junk: harmless
          xn[u][q][r] = act ? v : 0.f;
        }
    }
  };
  if (PF && tmax > 0) prefetch(0);

  for (int s = 0; s < tmax; ++s) {
    const T* hcur = hs + (s & 1) * 16 * LDH;
    T* hnxt = hs + ((s + 1) & 1) * 16 * LDH;
    f32x4_t acc[NUB][4];
    if (!PF) prefetch(s);
#pragma unroll
    for (int u = 0; u < NUB; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[u][q][r] = xn[u][q][r];
    if (PF && s + 1 < tmax) prefetch(s + 1);

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
    // k-chunks streamed from L2 (limited unroll: each chunk keeps 4*NUB 16-B loads in flight)
#pragma unroll 2
    for (int ks = KSL; ks < KS; ++ks) {
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

#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int brow = rg * 4 + r;
      const bool act = s < len[r];
      const int t = act ? (rev ? len[r] - 1 - s : s) : s;   // inactive: frame s is a padded frame
      const size_t rowi = (size_t)t * B_ + b0 + brow;
      float* gp = xg + rowi * G4 + d * 4 * H + wave * 16 + col;
      float* cp_ = cs + rowi * G1 + d * H + wave * 16 + col;
      T* hp = hout + rowi * G1 + d * H + wave * 16 + col;
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
        const int o = NW * 16 * u;
        if (act) {
          gp[o] = ig; gp[H + o] = gg; gp[2 * H + o] = fg; gp[3 * H + o] = og;
          cp_[o] = cn;
        }
        hp[o] = Elem<T>::from_f32(act ? hn : 0.f);
        hnxt[brow * LDH + (wave + NW * u) * 16 + col] = Elem<T>::from_f32(hr[u][r]);
      }
    }
    __syncthreads();
  }
  // zero-fill the common padded tail [tmax, T)
  for (int s = tmax; s < T_; ++s)
#pragma unroll
    for (int u = 0; u < NUB; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        hout[((size_t)s * B_ + b0 + rg * 4 + r) * G1 + d * H + (wave + NW * u) * 16 + col] = T(0);
#pragma unroll
  for (int u = 0; u < NUB; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = ((size_t)d * B_ + b0 + rg * 4 + r) * H + (wave + NW * u) * 16 + col;
      if (c_final) c_final[o] = c[u][r];
      if (h_final) h_final[o] = hr[u][r];
    }
}

// ---------------------------------------------------------------- backward (BPTT)
// TF's LSTMBlockCellGrad restated (cell_clip is not part of the gradient op):
//   do = dh*tanh(c)*o(1-o);  dc = dc_rec + dh*o*(1-tanh(c)^2) + do*wco
//   dci = dc*i*(1-ci^2); di = dc*ci*i(1-i); df = dc*c_prev*f(1-f)
//   dc_prev = dc*f + di*wci + df*wcf;  dh_prev = [di dci df do] W_h^T
template <typename T, int H, bool DB, int NW, int KSL>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dhout, const float* __restrict__ gates,
    const float* __restrict__ cs, const T* __restrict__ whpb, const float* __restrict__ peep,
    const int32_t* __restrict__ seq_len, const float* __restrict__ d_c_final,
    const float* __restrict__ d_h_final, T* __restrict__ dgates, float* __restrict__ dpeep_part) {
  constexpr int NUB = H / (16 * NW);
  constexpr int KV = LT<T>::KV, KS = 4 * H / KV;
  constexpr int LDG = 4 * H + LT<T>::PAD;
  constexpr int E = 16 / (int)sizeof(T);
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* dgs = reinterpret_cast<T*>(smem);                      // [DB?2:1][16][LDG]
  T* wl = dgs + (DB ? 2 : 1) * 16 * LDG;                    // [ub][KSL][64][E]

  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool rev = (d == 1);
  const int G4 = ndir * 4 * H, G1 = ndir * H;
  const T* wp = whpb + (size_t)d * H * 4 * H;

  int len[4];
  int tmax = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) len[r] = seq_len[b0 + rg * 4 + r];
  for (int i = 0; i < 16; ++i) tmax = max(tmax, seq_len[b0 + i]);
  tmax = min(tmax, T_);

  float dhr[NUB][4], dcr[NUB][4], wci[NUB], wcf[NUB], wco[NUB];
  float pwi[NUB], pwf[NUB], pwo[NUB];
#pragma unroll
  for (int u = 0; u < NUB; ++u) {
    const int j = (wave + NW * u) * 16 + col;
    wci[u] = peep ? peep[(d * 3 + 0) * H + j] : 0.f;
    wcf[u] = peep ? peep[(d * 3 + 1) * H + j] : 0.f;
    wco[u] = peep ? peep[(d * 3 + 2) * H + j] : 0.f;
    pwi[u] = pwf[u] = pwo[u] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = ((size_t)d * B_ + b0 + rg * 4 + r) * H + j;
      dhr[u][r] = d_h_final ? d_h_final[o] : 0.f;
      dcr[u][r] = d_c_final ? d_c_final[o] : 0.f;
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

  // zero-fill the common padded tail frames [tmax, T)
  for (int s = T_ - 1; s >= tmax; --s)
#pragma unroll
    for (int u = 0; u < NUB; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T* gp = dgates + ((size_t)s * B_ + b0 + rg * 4 + r) * G4 + d * 4 * H + (wave + NW * u) * 16 + col;
        gp[0] = T(0); gp[H] = T(0); gp[2 * H] = T(0); gp[3 * H] = T(0);
      }
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
      if (!DB) __syncthreads();  // all reads of dgs done before it is overwritten
    }
    // ---- loads that do not depend on the recurrence (after the MFMA chain: with 4 waves/SIMD other waves cover
    //      the latency, and the two phases do not add up their register pressure)
    float gi[NUB][4], gg[NUB][4], gf[NUB][4], go[NUB][4], cc[NUB][4], cp[NUB][4], dho[NUB][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = s < len[r];
      const int t = act ? (rev ? len[r] - 1 - s : s) : 0;
      const bool hasp = act && s > 0;
      const int tp = hasp ? (rev ? t + 1 : t - 1) : 0;
      const size_t rowi = (size_t)t * B_ + b0 + rg * 4 + r;
      const size_t rowp = (size_t)tp * B_ + b0 + rg * 4 + r;
      const float* gp = gates + rowi * G4 + d * 4 * H + wave * 16 + col;
      const float* cq = cs + rowi * G1 + d * H + wave * 16 + col;
      const float* cpq = cs + rowp * G1 + d * H + wave * 16 + col;
      const float* dq = dhout + rowi * G1 + d * H + wave * 16 + col;
#pragma unroll
      for (int u = 0; u < NUB; ++u) {
        const int o = NW * 16 * u;
        gi[u][r] = gp[o];
        gg[u][r] = gp[H + o];
        gf[u][r] = gp[2 * H + o];
        go[u][r] = gp[3 * H + o];
        cc[u][r] = cq[o];
        const float cpv = cpq[o];
        cp[u][r] = hasp ? cpv : 0.f;
        dho[u][r] = dq[o];
      }
    }
    // ---- gate gradients (predicated, no divergent control flow)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int brow = rg * 4 + r;
      const bool act = s < len[r];
      const int t = act ? (rev ? len[r] - 1 - s : s) : s;    // inactive: frame s is padded -> zeros
      T* gp = dgates + ((size_t)t * B_ + b0 + brow) * G4 + d * 4 * H + wave * 16 + col;
      T* ls = dcur + brow * LDG + wave * 16 + col;
#pragma unroll
      for (int u = 0; u < NUB; ++u) {
        const float dh = dho[u][r] + dhr[u][r];
        const float tc = ftanh(cc[u][r]);
        const float o = go[u][r], i = gi[u][r], g = gg[u][r], f = gf[u][r];
        const float d_o = dh * tc * o * (1.f - o);
        const float dc = dcr[u][r] + dh * o * (1.f - tc * tc) + d_o * wco[u];
        const float d_g = dc * i * (1.f - g * g);
        const float d_i = dc * g * i * (1.f - i);
        const float d_f = dc * cp[u][r] * f * (1.f - f);
        dcr[u][r] = act ? (dc * f + d_i * wci[u] + d_f * wcf[u]) : dcr[u][r];
        dhr[u][r] = act ? 0.f : dhr[u][r];   // active: the next MFMA supplies dh_prev
        pwi[u] += act ? d_i * cp[u][r] : 0.f;
        pwf[u] += act ? d_f * cp[u][r] : 0.f;
        pwo[u] += act ? d_o * cc[u][r] : 0.f;
        const T ti = Elem<T>::from_f32(act ? d_i : 0.f), tg = Elem<T>::from_f32(act ? d_g : 0.f),
                tf = Elem<T>::from_f32(act ? d_f : 0.f), to = Elem<T>::from_f32(act ? d_o : 0.f);
        const int oo = NW * 16 * u;
        gp[oo] = ti; gp[H + oo] = tg; gp[2 * H + oo] = tf; gp[3 * H + oo] = to;
        ls[oo] = ti; ls[H + oo] = tg; ls[2 * H + oo] = tf; ls[3 * H + oo] = to;
      }
    }
    __syncthreads();
  }

  // peephole gradient partials: sum over the 4 row groups sharing a unit column
  if (dpeep_part) {
#pragma unroll
    for (int u = 0; u < NUB; ++u) {
      float a = pwi[u], b = pwf[u], cpo = pwo[u];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
      cpo += __shfl_xor(cpo, 16, 64); cpo += __shfl_xor(cpo, 32, 64);
      if (rg == 0) {
        const int j = (wave + NW * u) * 16 + col;
        float* p = dpeep_part + ((size_t)blockIdx.x * ndir + d) * 3 * H;
        p[j] = a; p[H + j] = b; p[2 * H + j] = cpo;
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

template <typename T, int H>
int launch_fwd(int T_, int B, int ndir, float* xg, const void* whp, const float* peep,
               const int32_t* seq_len, float fb, float clip, void* hout, float* cs, float* cf,
               float* hf, hipStream_t st) {
  constexpr int NW = pick_nw(H);
  constexpr int KSL = fwd_ksl<T>(H);
  const size_t lds = (size_t)2 * 16 * (H + LT<T>::PAD) * sizeof(T) + (size_t)(H / 16) * 4 * KSL * 1024;
  constexpr bool PF = (H / (16 * NW)) * NW <= 16;   // prefetch x W_x only when 128 VGPRs/wave can hold it
  auto k = lstm_fwd_kernel<T, H, NW, KSL, PF>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(B / 16, ndir), dim3(NW * 64), lds, st, T_, B, ndir, xg, (const T*)whp, peep,
                     seq_len, fb, clip, (T*)hout, cs, cf, hf);
  return 0;
}

template <typename T, int H>
int launch_bwd(int T_, int B, int ndir, const float* dhout, const float* gates, const float* cs,
               const void* whpb, const float* peep, const int32_t* seq_len, const float* dcf,
               const float* dhf, void* dgates, float* dpeep_part, hipStream_t st) {
  constexpr size_t one = (size_t)16 * (4 * H + LT<T>::PAD) * sizeof(T);
  constexpr bool DB = (2 * one <= 72 * 1024);
  constexpr int NW = pick_nw(H);
  constexpr int KSL = bwd_ksl<T>(H, DB);
  const size_t lds = (DB ? 2 * one : one) + (size_t)(H / 16) * KSL * 1024;
  auto k = lstm_bwd_kernel<T, H, DB, NW, KSL>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(B / 16, ndir), dim3(NW * 64), lds, st, T_, B, ndir, dhout, gates, cs,
                     (const T*)whpb, peep, seq_len, dcf, dhf, (T*)dgates, dpeep_part);
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

extern "C" int asr_lstm_pack_wh(asr_handle* h, int dtype, const float* wh, int H, void* packed_fwd,
                                void* packed_bwd, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || !wh || !packed_fwd || !packed_bwd || H <= 0 || H % 64)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_pack_wh: bad args (H=%d must be a multiple of 64)", H);
  const size_t total = (size_t)H * 4 * H;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  if (dtype == ASR_F32)
    hipLaunchKernelGGL(pack_wh_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, wh, H,
                       (float*)packed_fwd, (float*)packed_bwd);
  else
    hipLaunchKernelGGL(pack_wh_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, wh, H,
                       (bf16_t*)packed_fwd, (bf16_t*)packed_bwd);
  ASR_CHECK_LAUNCH(h, "asr_lstm_pack_wh");
  return ASR_OK;
}

extern "C" int asr_lstm_fwd(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                            float* xproj_gates, const void* wh_packed, const float* peep,
                            const int32_t* seq_len, float forget_bias, float cell_clip, void* hout,
                            float* cs, float* c_final, float* h_final, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || T < 0 || B <= 0 || B % 16 || (ndir != 1 && ndir != 2) ||
      !xproj_gates || !wh_packed || !seq_len || !hout || !cs)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_fwd: bad args (B=%d must be a multiple of 16, ndir=%d)", B, ndir);
  if (T == 0) return ASR_OK;
  hipStream_t st = (hipStream_t)s;
  if (dtype == ASR_F32) {
    ASR_H_DISPATCH(H, T, (launch_fwd<float, HH>(T, B, ndir, xproj_gates, wh_packed, peep, seq_len,
                                                forget_bias, cell_clip, hout, cs, c_final, h_final, st)));
  } else {
    ASR_H_DISPATCH(H, T, (launch_fwd<bf16_t, HH>(T, B, ndir, xproj_gates, wh_packed, peep, seq_len,
                                                 forget_bias, cell_clip, hout, cs, c_final, h_final, st)));
  }
  ASR_CHECK_LAUNCH(h, "asr_lstm_fwd");
  return ASR_OK;
}

extern "C" int asr_lstm_bwd(asr_handle* h, int dtype, int T, int B, int H, int ndir,
                            const float* dhout, const float* gates, const float* cs,
                            const void* wh_packed_bwd, const float* peep, const int32_t* seq_len,
                            const float* d_c_final, const float* d_h_final, void* dgates,
                            float* dpeep, float* dpeep_workspace, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || T < 0 || B <= 0 || B % 16 || (ndir != 1 && ndir != 2) || !dhout ||
      !gates || !cs || !wh_packed_bwd || !seq_len || !dgates)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_bwd: bad args (B=%d must be a multiple of 16, ndir=%d)", B, ndir);
  if (dpeep && !dpeep_workspace)
    ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_lstm_bwd: dpeep needs a workspace of (B/16)*ndir*3*H floats");
  hipStream_t st = (hipStream_t)s;
  if (T == 0) {
    if (dpeep) (void)hipMemsetAsync(dpeep, 0, sizeof(float) * ndir * 3 * H, st);
    return ASR_OK;
  }
  float* part = dpeep ? dpeep_workspace : nullptr;
  if (dtype == ASR_F32) {
    ASR_H_DISPATCH(H, T, (launch_bwd<float, HH>(T, B, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len,
                                                d_c_final, d_h_final, dgates, part, st)));
  } else {
    ASR_H_DISPATCH(H, T, (launch_bwd<bf16_t, HH>(T, B, ndir, dhout, gates, cs, wh_packed_bwd, peep, seq_len,
                                                 d_c_final, d_h_final, dgates, part, st)));
  }
  ASR_CHECK_LAUNCH(h, "asr_lstm_bwd");
  if (dpeep) {
    const int n = ndir * 3 * H;
    hipLaunchKernelGGL(reduce_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, st, part, B / 16, n, dpeep);
    ASR_CHECK_LAUNCH(h, "asr_lstm_bwd(reduce)");
  }
  return ASR_OK;
}
