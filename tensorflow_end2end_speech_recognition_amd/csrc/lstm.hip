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
//   * one workgroup (4 waves) per (direction, 16-utterance batch tile) runs ALL T steps;
//     the 16 utterances are the M dimension of a 16x16 MFMA tile, so a lane of the C/D
//     fragment permanently owns (utterance b = (lane>>4)*4+r, unit j = ub*16+(lane&15)):
//     c, h, the peepholes and the four gate pre-activations of that (b,j) never leave
//     its registers -> the gate math needs no cross-lane traffic.
//   * wave w owns unit blocks ub = w, w+4, ...; for each it accumulates the four gate
//     tiles (i, ci, f, o) so one lane ends a step with all four gates of its (b,j).
//   * h_{t-1} (16 x H) lives in LDS (double buffered, one barrier per step) as the MFMA
//     A operand; W_h is pre-packed in B-fragment order so every wave-load is one
//     contiguous 1 KiB (bf16) / 1 KiB (4 k-steps of fp32) line, streamed from L2 each step.
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

// ---------------------------------------------------------------- forward
template <typename T, int H, int NW, bool PF>
__global__ __launch_bounds__(NW * 64, NW / 4) void lstm_fwd_kernel(
    int T_, int B_, int ndir, float* __restrict__ xg, const T* __restrict__ whp,
    const float* __restrict__ peep, const int32_t* __restrict__ seq_len, float forget_bias,
    float cell_clip, T* __restrict__ hout, float* __restrict__ cs, float* __restrict__ c_final,
    float* __restrict__ h_final) {
  constexpr int NUB = H / (16 * NW);
  constexpr int KV = LT<T>::KV, KS = H / KV;
  constexpr int LDH = H + LT<T>::PAD;
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* hs = reinterpret_cast<T*>(smem);  // [2][16][LDH]

  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool rev = (d == 1);
  const int G4 = ndir * 4 * H, G1 = ndir * H;

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
  __syncthreads();

  const T* wp = whp + (size_t)d * H * 4 * H;

  // prefetch registers for x W_x + b of the coming step
  // (PF = false for H >= 512: 16xH (b,j) pairs x {4 acc, 4 prefetch, c, h} would not fit the
  // 512 KB register file of one CU; there the loads are issued just before the MFMA chain.)
  float xn[NUB][4][4];
  auto prefetch = [&](int s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = s < len[r];
      const int t = rev ? len[r] - 1 - s : s;
      const float* row = xg + ((size_t)(act ? t : 0) * B_ + b0 + rg * 4 + r) * G4 + d * 4 * H;
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          xn[u][q][r] = act ? row[q * H + (wave + NW * u) * 16 + col] : 0.f;
    }
  };
  if (PF && tmax > 0) prefetch(0);

  for (int s = 0; s < T_; ++s) {
    if (s < tmax) {
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

#pragma unroll 2
      for (int ks = 0; ks < KS; ++ks) {
        const frag_t a = *reinterpret_cast<const frag_t*>(hcur + col * LDH + ks * KV + rg * (16 / (int)sizeof(T)));
#pragma unroll
        for (int u = 0; u < NUB; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const size_t fi = ((size_t)((wave + NW * u) * 4 + q) * KS + ks) * 64 + lane;
            const frag_t b = *reinterpret_cast<const frag_t*>(wp + fi * (16 / sizeof(T)));
            acc[u][q] = mma_chunk(a, b, acc[u][q]);
          }
      }

#pragma unroll
      for (int u = 0; u < NUB; ++u) {
        const int j = (wave + NW * u) * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int brow = rg * 4 + r, b = b0 + brow;
          const bool act = s < len[r];
          const float cp = c[u][r];
          const float ig = sigmoidf_(acc[u][0][r] + wci[u] * cp);
          const float gg = tanhf_(acc[u][1][r]);
          const float fg = sigmoidf_(acc[u][2][r] + forget_bias + wcf[u] * cp);
          float cn = gg * ig + cp * fg;
          if (cell_clip > 0.f) cn = fminf(fmaxf(cn, -cell_clip), cell_clip);
          const float og = sigmoidf_(acc[u][3][r] + wco[u] * cn);
          const float hn = tanhf_(cn) * og;
          if (act) {
            const int t = rev ? len[r] - 1 - s : s;
            const size_t rowi = (size_t)t * B_ + b;
            float* gp = xg + rowi * G4 + d * 4 * H + j;
            gp[0] = ig; gp[H] = gg; gp[2 * H] = fg; gp[3 * H] = og;
            cs[rowi * G1 + d * H + j] = cn;
            hout[rowi * G1 + d * H + j] = Elem<T>::from_f32(hn);
            c[u][r] = cn;
            hr[u][r] = hn;
          } else {
            // frame s is a padded frame of utterance b in either direction
            hout[((size_t)s * B_ + b) * G1 + d * H + j] = T(0);
          }
          hnxt[brow * LDH + j] = Elem<T>::from_f32(hr[u][r]);
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hout[((size_t)s * B_ + b0 + rg * 4 + r) * G1 + d * H + (wave + NW * u) * 16 + col] = T(0);
    }
  }
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
template <typename T, int H, bool DB, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void lstm_bwd_kernel(
    int T_, int B_, int ndir, const float* __restrict__ dhout, const float* __restrict__ gates,
    const float* __restrict__ cs, const T* __restrict__ whpb, const float* __restrict__ peep,
    const int32_t* __restrict__ seq_len, const float* __restrict__ d_c_final,
    const float* __restrict__ d_h_final, T* __restrict__ dgates, float* __restrict__ dpeep_part) {
  constexpr int NUB = H / (16 * NW);
  constexpr int KV = LT<T>::KV, KS = 4 * H / KV;
  constexpr int LDG = 4 * H + LT<T>::PAD;
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* dgs = reinterpret_cast<T*>(smem);  // [DB?2:1][16][LDG]

  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const bool rev = (d == 1);
  const int G4 = ndir * 4 * H, G1 = ndir * H;

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
  const T* wp = whpb + (size_t)d * H * 4 * H;

  // zero-fill the padded tail frames [tmax, T)
  for (int s = T_ - 1; s >= tmax; --s)
#pragma unroll
    for (int u = 0; u < NUB; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T* gp = dgates + ((size_t)s * B_ + b0 + rg * 4 + r) * G4 + d * 4 * H + (wave + NW * u) * 16 + col;
        gp[0] = T(0); gp[H] = T(0); gp[2 * H] = T(0); gp[3 * H] = T(0);
      }

  for (int s = tmax - 1; s >= 0; --s) {
    T* dcur = dgs + (DB ? (s & 1) : 0) * 16 * LDG;
    // ---- loads that do not depend on the recurrence (issued before the MFMA chain)
    float gi[NUB][4], gg[NUB][4], gf[NUB][4], go[NUB][4], cc[NUB][4], cp[NUB][4], dho[NUB][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = s < len[r];
      const int t = rev ? len[r] - 1 - s : s;
      const int tp = rev ? t + 1 : t - 1;
      const size_t rowi = (size_t)(act ? t : 0) * B_ + b0 + rg * 4 + r;
      const size_t rowp = (size_t)((act && s > 0) ? tp : 0) * B_ + b0 + rg * 4 + r;
#pragma unroll
      for (int u = 0; u < NUB; ++u) {
        const int j = (wave + NW * u) * 16 + col;
        const float* gp = gates + rowi * G4 + d * 4 * H + j;
        gi[u][r] = act ? gp[0] : 0.f;
        gg[u][r] = act ? gp[H] : 0.f;
        gf[u][r] = act ? gp[2 * H] : 0.f;
        go[u][r] = act ? gp[3 * H] : 0.f;
        cc[u][r] = act ? cs[rowi * G1 + d * H + j] : 0.f;
        cp[u][r] = (act && s > 0) ? cs[rowp * G1 + d * H + j] : 0.f;
        dho[u][r] = act ? dhout[rowi * G1 + d * H + j] : 0.f;
      }
    }
    // ---- dh_rec of this step = dG(step s+1) W_h^T, carried through inactive rows
    if (s != tmax - 1) {
      const T* dprev = dgs + (DB ? ((s + 1) & 1) : 0) * 16 * LDG;
      f32x4_t acc[NUB];
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[u][r] = dhr[u][r];  // holds carry (inactive) or 0 (active)
#pragma unroll 2
      for (int ks = 0; ks < KS; ++ks) {
        const frag_t a = *reinterpret_cast<const frag_t*>(dprev + col * LDG + ks * KV + rg * (16 / (int)sizeof(T)));
#pragma unroll
        for (int u = 0; u < NUB; ++u) {
          const size_t fi = ((size_t)(wave + NW * u) * KS + ks) * 64 + lane;
          const frag_t b = *reinterpret_cast<const frag_t*>(wp + fi * (16 / sizeof(T)));
          acc[u] = mma_chunk(a, b, acc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < NUB; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) dhr[u][r] = acc[u][r];
      if (!DB) __syncthreads();  // all reads of dgs done before it is overwritten
    }
    // ---- gate gradients
#pragma unroll
    for (int u = 0; u < NUB; ++u) {
      const int j = (wave + NW * u) * 16 + col;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int brow = rg * 4 + r, b = b0 + brow;
        const bool act = s < len[r];
        T* ls = dcur + brow * LDG + j;
        if (act) {
          const int t = rev ? len[r] - 1 - s : s;
          const float dh = dho[u][r] + dhr[u][r];
          const float tc = tanhf_(cc[u][r]);
          const float o = go[u][r], i = gi[u][r], g = gg[u][r], f = gf[u][r];
          const float d_o = dh * tc * o * (1.f - o);
          const float dc = dcr[u][r] + dh * o * (1.f - tc * tc) + d_o * wco[u];
          const float d_g = dc * i * (1.f - g * g);
          const float d_i = dc * g * i * (1.f - i);
          const float d_f = dc * cp[u][r] * f * (1.f - f);
          dcr[u][r] = dc * f + d_i * wci[u] + d_f * wcf[u];
          dhr[u][r] = 0.f;  // the MFMA of the next iteration supplies dh_prev
          pwi[u] += d_i * cp[u][r];
          pwf[u] += d_f * cp[u][r];
          pwo[u] += d_o * cc[u][r];
          const T ti = Elem<T>::from_f32(d_i), tg = Elem<T>::from_f32(d_g),
                  tf = Elem<T>::from_f32(d_f), to = Elem<T>::from_f32(d_o);
          T* gp = dgates + ((size_t)t * B_ + b) * G4 + d * 4 * H + j;
          gp[0] = ti; gp[H] = tg; gp[2 * H] = tf; gp[3 * H] = to;
          ls[0] = ti; ls[H] = tg; ls[2 * H] = tf; ls[3 * H] = to;
        } else {
          // padded frame s: zero gradient; (dh, dc) carried unchanged
          T* gp = dgates + ((size_t)s * B_ + b) * G4 + d * 4 * H + j;
          gp[0] = T(0); gp[H] = T(0); gp[2 * H] = T(0); gp[3 * H] = T(0);
          ls[0] = T(0); ls[H] = T(0); ls[2 * H] = T(0); ls[3 * H] = T(0);
        }
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
  const size_t lds = (size_t)2 * 16 * (H + LT<T>::PAD) * sizeof(T);
  constexpr int NW = (H >= 512) ? 8 : 4;
  auto k = lstm_fwd_kernel<T, H, NW, (H < 512)>;
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
  constexpr bool DB = (2 * one <= 150 * 1024);
  const size_t lds = DB ? 2 * one : one;
  constexpr int NW = (H >= 512) ? 8 : 4;
  auto k = lstm_bwd_kernel<T, H, DB, NW>;
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
