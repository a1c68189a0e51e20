// VGG front-end helpers for gfx950 (row a5): the per-frame convolution stack of
// models/encoders/core/vgg_blstm.py:107-177 over windows [F=channels(40), W=splice*stack, 3] NHWC
// (layout of utils/io/inputs/splicing.py:60-73), conv_layer / max_pool of cnn_util.py:13-84:
//   conv3x3 SAME + bias + relu (x4), max_pool 2x2/2 SAME (x2), flatten, bridge FC 256 + relu.
//
// Round-1 form: the 3x3 SAME convolutions run as  im2col (this file)  ->  asr_gemm with a fused
// bias+ReLU epilogue (MFMA)  -- chunked over frames so the patch matrix stays a bounded scratch.
// (An implicit-GEMM A-operand gather that never materialises the patches is the planned upgrade;
// DESIGN.md.)  Everything in this file is HBM-bound data movement.
#include "common.h"

namespace {

// patches[m, tap*Cin + ci] = in[n, h+dh, w+dw, ci] (0 outside); m = (n*H + h)*W + w; row stride ldp
template <typename T>
__global__ void im2col3x3_kernel(const T* __restrict__ in, int N, int H, int W, int Cin, int ldp,
                                 T* __restrict__ patches) {
  const size_t total = (size_t)N * H * W * 9 * Cin;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int ci = idx % Cin;
    const int tap = (idx / Cin) % 9;
    const size_t m = idx / ((size_t)9 * Cin);
    const int w = m % W, h = (m / W) % H;
    const size_t n = m / ((size_t)W * H);
    const int hs = h + tap / 3 - 1, ws = w + tap % 3 - 1;
    T v = T(0);
    if (hs >= 0 && hs < H && ws >= 0 && ws < W) v = in[((n * H + hs) * W + ws) * Cin + ci];
    patches[m * ldp + tap * Cin + ci] = v;
  }
}

// din[n,h,w,ci] = sum_taps dpatches[pixel(h-dh, w-dw), tap*Cin+ci]   (gather: no atomics)
__global__ void col2im3x3_kernel(const float* __restrict__ dpatches, int N, int H, int W, int Cin, int ldp,
                                 float* __restrict__ din) {
  const size_t total = (size_t)N * H * W * Cin;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int ci = idx % Cin;
    const size_t m = idx / Cin;
    const int w = m % W, h = (m / W) % H;
    const size_t n = m / ((size_t)W * H);
    float s = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hd = h - (tap / 3 - 1), wd = w - (tap % 3 - 1);
      if (hd >= 0 && hd < H && wd >= 0 && wd < W)
        s += dpatches[((n * H + hd) * W + wd) * ldp + tap * Cin + ci];
    }
    din[idx] = s;
  }
}

// General SAME convolution as im2col + GEMM (the CLDNN front-end, cldnn_wang.py:141-177: 11x21 stride (3,2),
// 11x11 stride (1,2), 3x3): patches[p, (ky*kw + kx)*Cin + ci] = in[n, yo*sh + ky - pt, xo*sw + kx - pl, ci] (0 outside),
// p = (n*Ho + yo)*Wo + xo, Ho = ceil(H/sh), pt = max((Ho-1)*sh + kh - H, 0) / 2 (TF puts the odd cell after).
struct ConvGeo { int N, H, W, Cin, kh, kw, sh, sw, Ho, Wo, pt, pl; };
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ in, ConvGeo g, int ldp, T* __restrict__ patches) {
  const int K = g.kh * g.kw * g.Cin;
  const size_t total = (size_t)g.N * g.Ho * g.Wo * K;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = idx % K;
    const size_t p = idx / K;
    const int ci = kk % g.Cin, tap = kk / g.Cin, kx = tap % g.kw, ky = tap / g.kw;
    const int xo = p % g.Wo, yo = (p / g.Wo) % g.Ho;
    const size_t n = p / ((size_t)g.Wo * g.Ho);
    const int y = yo * g.sh + ky - g.pt, x = xo * g.sw + kx - g.pl;
    T v = T(0);
    if (y >= 0 && y < g.H && x >= 0 && x < g.W) v = in[((n * g.H + y) * g.W + x) * g.Cin + ci];
    patches[p * ldp + kk] = v;
  }
}
// din[n,y,x,ci] = sum over the (ky,kx) whose output pixel exists of dpatches[...]   (gather: no atomics, fixed order)
__global__ void col2im_kernel(const float* __restrict__ dpatches, ConvGeo g, int ldp, float* __restrict__ din) {
  const size_t total = (size_t)g.N * g.H * g.W * g.Cin;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int ci = idx % g.Cin;
    const size_t m = idx / g.Cin;
    const int x = m % g.W, y = (m / g.W) % g.H;
    const size_t n = m / ((size_t)g.W * g.H);
    float s = 0.f;
    for (int ky = 0; ky < g.kh; ++ky) {
      const int yy = y + g.pt - ky;
      if (yy < 0 || yy % g.sh) continue;
      const int yo = yy / g.sh;
      if (yo >= g.Ho) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int xx = x + g.pl - kx;
        if (xx < 0 || xx % g.sw) continue;
        const int xo = xx / g.sw;
        if (xo >= g.Wo) continue;
        s += dpatches[((n * g.Ho + yo) * g.Wo + xo) * ldp + (ky * g.kw + kx) * g.Cin + ci];
      }
    }
    din[idx] = s;
  }
}

// max_pool 2x2 stride 2 SAME (cnn_util.py:13-28): Ho = ceil(H/2); padding goes AFTER (-inf)
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ in, int N, int H, int W, int C, T* __restrict__ out,
                                   uint8_t* __restrict__ arg) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)N * Ho * Wo * C;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c = idx % C;
    const int wo = (idx / C) % Wo, ho = (idx / ((size_t)C * Wo)) % Ho;
    const size_t n = idx / ((size_t)C * Wo * Ho);
    float best = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int h = 2 * ho + (k >> 1), w = 2 * wo + (k & 1);
      if (h < H && w < W) {
        const float v = Elem<T>::to_f32(in[((n * H + h) * W + w) * C + c]);
        if (v > best) { best = v; bi = k; }
      }
    }
    out[idx] = Elem<T>::from_f32(best);
    arg[idx] = (uint8_t)bi;
  }
}
// four channels per thread (C % 4 == 0): one 8 / 16-byte load per window cell instead of four scalar ones
// use_drop: tf.nn.dropout on the (rounded) pooled output -- the stored pooled activation is the dropped one
// (n, h, w, c4) of a flat index over [N][H][W][C4]: 32-bit divisions when the whole range fits (three udiv of ~20
// instructions each); the size_t form -- five 64-bit divisions by run-time values, ~150 instructions each -- was what these
// "memory-bound" pooling passes actually spent their time on (round 4: un-pool pass 1.61 ms at 2.2 TB/s)
__device__ __forceinline__ void nhwc4_split(size_t idx, bool i32, int C4, int W, int H, int& c4, int& w, int& h, size_t& n) {
  if (i32) {
    unsigned t = (unsigned)idx;
    c4 = (int)(t % (unsigned)C4); t /= (unsigned)C4;
    w = (int)(t % (unsigned)W); t /= (unsigned)W;
    h = (int)(t % (unsigned)H);
    n = t / (unsigned)H;
  } else {
    c4 = (int)(idx % C4);
    w = (int)((idx / C4) % W);
    h = (int)((idx / ((size_t)C4 * W)) % H);
    n = idx / ((size_t)C4 * W * H);
  }
}

template <typename T>
__global__ void maxpool_fwd_vec_kernel(const T* __restrict__ in, int N, int H, int W, int C, T* __restrict__ out,
                                       uint8_t* __restrict__ arg, float keep = 1.f, uint64_t seed = 0, uint64_t offset = 0,
                                       int use_drop = 0) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C / 4;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    int c4, wo, ho;
    size_t n;
    nhwc4_split(idx, total <= 0xFFFFFFFFull, C4, Wo, Ho, c4, wo, ho, n);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int h = 2 * ho + (k >> 1), w = 2 * wo + (k & 1);
      if (h < H && w < W) {
        const T* p = in + ((n * H + h) * W + w) * C + (size_t)c4 * 4;
        float v[4];
        if (sizeof(T) == 2) {
          typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
          const us4_t x = *reinterpret_cast<const us4_t*>(p);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = bf16_to_f32(x[j]);
        } else {
          const f32x4_t x = *reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = x[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > best[j]) { best[j] = v[j]; bi[j] = k; }
      }
    }
    const size_t e = idx * 4;
    if (use_drop) {                                         // e % 4 == 0: exactly one Philox block
      float mk[4];
      asr_dropout_words(offset + e / 4, seed, keep, 1.f / keep, mk);
#pragma unroll
      for (int j = 0; j < 4; ++j) best[j] = Elem<T>::to_f32(Elem<T>::from_f32(best[j])) * mk[j];
    }
    if (sizeof(T) == 2) {
      typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
      *reinterpret_cast<us4_t*>(out + e) = (us4_t){f32_to_bf16(best[0]), f32_to_bf16(best[1]), f32_to_bf16(best[2]), f32_to_bf16(best[3])};
    } else {
      *reinterpret_cast<f32x4_t*>(out + e) = (f32x4_t){best[0], best[1], best[2], best[3]};
    }
    *reinterpret_cast<uint32_t*>(arg + e) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
  }
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ arg, int N, int H,
                                   int W, int C, float* __restrict__ din) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)N * H * W * C;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c = idx % C;
    const int w = (idx / C) % W, h = (idx / ((size_t)C * W)) % H;
    const size_t n = idx / ((size_t)C * W * H);
    const size_t o = ((n * Ho + h / 2) * Wo + w / 2) * C + c;
    const int k = ((h & 1) << 1) | (w & 1);
    din[idx] = (arg[o] == k) ? dout[o] : 0.f;
  }
}

// ---- 3x3 SAME convolution with FEW input channels (9 Cin <= 32: the first VGG layer, Cin = 3), direct.
// Through im2col + GEMM the K = 27 layer writes and re-reads a [pixels x 32] patch matrix (1.6 GB at cfg C, twice:
// forward and weight gradient) to feed 27 multiply-adds per output; the layer is bound by writing its OUTPUT, so the
// patch is gathered in registers instead.  bf16 operands, fp32 accumulation (as the MFMA path).
//   forward: 4 threads per pixel, 16 output channels each; weights [9 Cin][64] as fp32 in LDS (broadcast reads).
constexpr int SC_K = 32, SC_CO = 64;
template <int CIN>
__device__ __forceinline__ float sc_tap(const bf16_t* __restrict__ x, size_t n, int h, int w, int H, int W, int k) {
  constexpr int Cin = CIN;
  const int tap = k / Cin, ci = k - tap * Cin;              // CIN is a compile-time constant: no integer division
  const int hs = h + tap / 3 - 1, ws = w + tap % 3 - 1;
  return (hs >= 0 && hs < H && ws >= 0 && ws < W) ? bf16_to_f32(x[((n * H + hs) * W + ws) * Cin + ci]) : 0.f;
}
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_smallc_fwd_kernel(const bf16_t* __restrict__ x, size_t Npix, int H, int W,
                                                                 const bf16_t* __restrict__ w2d,
                                                                 const float* __restrict__ bias, int relu,
                                                                 bf16_t* __restrict__ out) {
  __shared__ float ws[SC_K][SC_CO];
  constexpr int K = 9 * CIN;
  for (int i = threadIdx.x; i < SC_K * SC_CO; i += 256) ws[i / SC_CO][i % SC_CO] = (i / SC_CO < K) ? bf16_to_f32(w2d[i]) : 0.f;
  __syncthreads();
  const size_t p = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (p >= Npix) return;
  const int cg = (threadIdx.x & 3) * 16;
  const int w = p % W, h = (p / W) % H;
  const size_t n = p / ((size_t)W * H);
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[cg + j] : 0.f;
  // by tap (9 trips), the CIN channels of a tap unrolled: full unrolling of all 27 terms hoists every load and spills
  // (12 ms instead of 3.8)
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) {
    const int k = tap * CIN + ci;
    const float xv = sc_tap<CIN>(x, n, h, w, H, W, k);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(&ws[k][cg + q * 4]);
      acc[q * 4 + 0] += xv * wv[0]; acc[q * 4 + 1] += xv * wv[1]; acc[q * 4 + 2] += xv * wv[2]; acc[q * 4 + 3] += xv * wv[3];
    }
  }
  typedef __attribute__((ext_vector_type(8))) unsigned short us8_t;
  bf16_t* o = out + p * SC_CO + cg;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    us8_t y;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = relu ? fmaxf(acc[q * 8 + j], 0.f) : acc[q * 8 + j];
      y[j] = f32_to_bf16(v);
    }
    *reinterpret_cast<us8_t*>(o + q * 8) = y;
  }
}
//   weight gradient dW[k][co] = sum_p patch[p][k] dpre[p][co]: a workgroup walks its slice of the pixels in tiles of
//   64, stages the tile's dpre rows and gathered patches in LDS and keeps its [32][64] sums in registers (thread: one
//   output channel x 8 consecutive k); the per-workgroup sums are added in a fixed order by a second kernel.
// The same layer on the matrix cores (round 3).  The VALU form above is bound by its 27 x 64 multiply-adds per pixel
// (87 GFLOP at cfg C = a third of the chip's fp32 vector rate: 3.4 ms), not by the 3.2 GB it writes.  As a GEMM the layer
// is [pixels x 32] x [32 x 64] with ONE k-step: a wave gathers the (zero-padded) 27-value patches of 16 pixels straight
// into an MFMA operand (lane = (pixel, 8 consecutive k)), multiplies them with the four 16-channel weight fragments it
// keeps in registers, and stages its 64 pixels x 64 channels through LDS so that the result leaves as 1 KB stores -- the
// output rows of consecutive pixels are contiguous.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_smallc_fwd_mfma_kernel(const bf16_t* __restrict__ x, size_t Npix, int H, int W,
                                                                      const bf16_t* __restrict__ w2d,
                                                                      const float* __restrict__ bias, int relu,
                                                                      bf16_t* __restrict__ out, float keep, uint64_t seed,
                                                                      uint64_t offset, int use_drop) {
  // use_drop: tf.nn.dropout on the (rounded) ReLU output in the epilogue -- the stored activation is the dropped one
  constexpr int K = 9 * CIN, LDT = SC_CO + 8;
  __shared__ __attribute__((aligned(16))) bf16_t tile[4][64][LDT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8_t wb[4];
  f32x4_t bv[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = fq * 8 + j;
      wb[nt][j] = k < K ? (short)w2d[k * SC_CO + nt * 16 + fr] : (short)0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[nt][r] = bias ? bias[nt * 16 + fq * 4 + r] : 0.f;
  }
  const size_t pbase = (size_t)blockIdx.x * 256 + (size_t)wave * 64;
  const int HW = H * W;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const size_t p = pbase + mt * 16 + fr;
    const bool valid = p < Npix;
    // (a 32-bit division when the pixel count allows: the size_t form is ~150 instructions per pixel)
    const size_t n = !valid ? 0 : (Npix <= 0xFFFFFFFFull ? (size_t)((unsigned)p / (unsigned)HW) : p / (size_t)HW);
    const int rem = valid ? (int)(p - n * HW) : 0;
    const int h = rem / W, w = rem - h * W;
    const bf16_t* xn = x + n * (size_t)HW * CIN;
    bf16x8_t a;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = fq * 8 + j;
      const int tap = k / CIN, ci = k - tap * CIN;
      const int hs = h + tap / 3 - 1, ws = w + tap % 3 - 1;
      const bool ok = valid && k < K && hs >= 0 && hs < H && ws >= 0 && ws < W;
      a[j] = ok ? (short)xn[(hs * W + ws) * CIN + ci] : (short)0;
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4_t acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[nt], a, bv[nt], 0, 0, 0);   // lane: pixel fr, channels fq*4..+3
      typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
      us4_t y;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = f32_to_bf16(relu ? fmaxf(acc[r], 0.f) : acc[r]);
      if (use_drop) {                                       // element p * 64 + nt * 16 + fq * 4: one Philox block of four
        float mk[4];
        asr_dropout_words(offset + (p * SC_CO + nt * 16 + fq * 4) / 4, seed, keep, 1.f / keep, mk);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = f32_to_bf16(bf16_to_f32(y[r]) * mk[r]);
      }
      *reinterpret_cast<us4_t*>(&tile[wave][mt * 16 + fr][nt * 16 + fq * 4]) = y;
    }
  }
  __syncthreads();
  typedef __attribute__((ext_vector_type(8))) unsigned short us8_t;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
    const size_t p = pbase + row;
    if (p < Npix) *reinterpret_cast<us8_t*>(out + p * SC_CO + c8) = *reinterpret_cast<const us8_t*>(&tile[wave][row][c8]);
  }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_smallc_wgrad_kernel(const bf16_t* __restrict__ x,
                                                                   const bf16_t* __restrict__ dpre, size_t Npix, int H,
                                                                   int W, size_t per_blk,
                                                                   float* __restrict__ partial) {
  __shared__ float dp[64][SC_CO + 1];
  __shared__ float pt[64][SC_K + 4];
  constexpr int K = 9 * CIN;
  const int co = threadIdx.x & 63, kg = (threadIdx.x >> 6) * 8;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const size_t p_beg = (size_t)blockIdx.x * per_blk, p_end = min(Npix, p_beg + per_blk);
  for (size_t p0 = p_beg; p0 < p_end; p0 += 64) {
    // stage: dpre rows (16 elements per thread), patches (8 per thread)
    {
      const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
      const size_t p = p0 + r;
      typedef __attribute__((ext_vector_type(8))) unsigned short us8_t;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        us8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p < p_end) v = *reinterpret_cast<const us8_t*>(dpre + p * SC_CO + c0 + q * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dp[r][c0 + q * 8 + j] = bf16_to_f32(v[j]);
      }
      const int k0 = (threadIdx.x & 3) * 8;
      const int w = p % W, h = (p / W) % H;
      const size_t n = p / ((size_t)W * H);
#pragma unroll
      for (int j = 0; j < 8; ++j) pt[r][k0 + j] = (p < p_end && k0 + j < K) ? sc_tap<CIN>(x, n, h, w, H, W, k0 + j) : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
      const float d = dp[r][co];
      const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(&pt[r][kg]), a1 = *reinterpret_cast<const f32x4_t*>(&pt[r][kg + 4]);
      acc[0] += a0[0] * d; acc[1] += a0[1] * d; acc[2] += a0[2] * d; acc[3] += a0[3] * d;
      acc[4] += a1[0] * d; acc[5] += a1[1] * d; acc[6] += a1[2] * d; acc[7] += a1[3] * d;
    }
    __syncthreads();
  }
  float* o = partial + (size_t)blockIdx.x * SC_K * SC_CO;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[(kg + i) * SC_CO + co] = acc[i];
}
// Weight gradient of the same layer on the matrix cores: dW[k][co] = sum_p patch[p][k] dpre[p][co] is a reduction-major
// product with M = 32 (27 used), N = 64 and K = pixels.  A wave stages 32 pixels at a time: each lane gathers 16 patch
// values of ONE pixel (one coordinate computation) and loads 64 bytes of that pixel's dpre row, both written as whole
// 32-byte rows into [16-column subtile][pixel] images -- the layout gemm.hip's reduction-major GEMM uses -- from which
// the transposing LDS reads (ds_read_b64_tr_b16) deliver the MFMA operands: 8 MFMAs per 32 pixels instead of 27 x 64
// vector multiply-adds per pixel.  Partials per workgroup as before.
constexpr int SCW_SUB = 32 * 32 + 32;                     // bytes from one [32 pixels][16 columns] subtile to the next
__device__ __forceinline__ bf16x8_t scw_frag(const char* p) {
  typedef __attribute__((ext_vector_type(4))) short s4_t;
  typedef __attribute__((address_space(3))) s4_t lds4_t;
  const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t*)(p));
  const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t*)(p + 128));
  return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_smallc_wgrad_mfma_kernel(const bf16_t* __restrict__ x,
                                                                        const bf16_t* __restrict__ dpre, size_t Npix, int H,
                                                                        int W, size_t per_blk,
                                                                        float* __restrict__ partial) {
  constexpr int K = 9 * CIN;
  // per wave: 2 patch + 4 dpre subtiles; the same memory holds the four waves' accumulators for the final sum
  constexpr int IMG_BYTES = 4 * 6 * SCW_SUB, RED_BYTES = 4 * SC_K * (SC_CO + 1) * 4;
  __shared__ __attribute__((aligned(16))) char lds[IMG_BYTES > RED_BYTES ? IMG_BYTES : RED_BYTES];
  char (*img)[6 * SCW_SUB] = reinterpret_cast<char (*)[6 * SCW_SUB]>(lds);
  float (*red)[SC_K][SC_CO + 1] = reinterpret_cast<float (*)[SC_K][SC_CO + 1]>(lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int px = lane & 31, half = lane >> 5;              // staging: pixel of the group, which half of its row
  const int HW = H * W;
  f32x4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  char* my = img[wave];
  const unsigned piece = (unsigned)(8 * fq + (fr >> 2)) * 32u + (unsigned)(fr & 3) * 8u;
  const size_t p_beg = (size_t)blockIdx.x * per_blk, p_end = min(Npix, p_beg + per_blk);
  typedef __attribute__((ext_vector_type(8))) unsigned short us8_t;
  for (size_t g0 = p_beg; g0 < p_end; g0 += 128) {          // block-uniform trip count
    const size_t p = g0 + (size_t)wave * 32 + px;
    const bool valid = p < p_end;
    // ---- stage: 16 patch values (k = 16 half .. + 15) and 32 channels (32 half .. + 31) of pixel p
    {
      const size_t n = valid ? p / (size_t)HW : 0;
      const int rem = valid ? (int)(p - n * HW) : 0;
      const int h = rem / W, w = rem - h * W;
      const bf16_t* xn = x + n * (size_t)HW * CIN;
      us8_t pv[2];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = half * 16 + j;
        const int tap = k / CIN, ci = k - tap * CIN;
        const int hs = h + tap / 3 - 1, ws = w + tap % 3 - 1;
        const bool ok = valid && k < K && hs >= 0 && hs < H && ws >= 0 && ws < W;
        // (k == K < 32: a column of ONES -- row K of the product is the column sum of dpre, the layer's bias gradient)
        pv[j >> 3][j & 7] = ok ? xn[(hs * W + ws) * CIN + ci] : (bf16_t)((valid && k == K) ? 0x3F80 : 0);
      }
      char* prow = my + half * SCW_SUB + px * 32;
      *reinterpret_cast<us8_t*>(prow) = pv[0];
      *reinterpret_cast<us8_t*>(prow + 16) = pv[1];
      const us8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const us8_t dv = valid ? *reinterpret_cast<const us8_t*>(dpre + p * SC_CO + half * 32 + q * 8) : zero;
        // channels half*32 + q*8 .. +7 -> subtile 2 half + (q >> 1), half-row (q & 1)
        *reinterpret_cast<us8_t*>(my + (2 + 2 * half + (q >> 1)) * SCW_SUB + px * 32 + (q & 1) * 16) = dv;
      }
    }
    __syncthreads();
    {
      bf16x8_t a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = scw_frag(my + i * SCW_SUB + piece);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = scw_frag(my + (2 + j) * SCW_SUB + piece);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // C[k = 16 i + fr][co = 16 j + 4 fq ..]
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][i * 16 + fr][j * 16 + fq * 4 + r] = acc[i][j][r];
  __syncthreads();
  float* o = partial + (size_t)blockIdx.x * SC_K * SC_CO;
  for (int e = threadIdx.x; e < SC_K * SC_CO; e += 256) {
    const int k = e / SC_CO, co = e % SC_CO;
    o[e] = red[0][k][co] + red[1][k][co] + red[2][k][co] + red[3][k][co];
  }
}
__global__ void conv3x3_smallc_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int K,
                                                   float* __restrict__ dw, float* __restrict__ dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (K + (dbias ? 1 : 0)) * SC_CO) return;
  float s = 0.f;
  for (int b0 = 0; b0 < nblk; b0 += 16) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = (b0 + q < nblk) ? partial[(size_t)(b0 + q) * SC_K * SC_CO + i] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q];
  }
  if (i < K * SC_CO) dw[i] = s;
  else dbias[i - K * SC_CO] = s;                            // row K: the ones column of the matrix-core kernel
}

// First of two passes for many partials (round 6: 2048 workgroups leave 16.8 MB, and seven blocks walking all of them row by
// row took 0.11 ms alone and 2 ms beside the other weight-gradient kernels at the end of a cfg C step): block (x, seg) sums
// rows [seg SEGR, (seg + 1) SEGR) of 256 columns in row order -> stage[seg][2048]; the pass above then adds the segment sums
// in segment order.  Fixed order, deterministic (the grouping differs from one sequential sum).
constexpr int SC_SEGR = 64;
__global__ __launch_bounds__(256) void conv3x3_smallc_wgrad_stage_kernel(const float* __restrict__ partial, int nblk,
                                                                         int ncols, float* __restrict__ stage) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * SC_SEGR, r1 = min(nblk, r0 + SC_SEGR);
  if (i >= ncols) return;
  float s = 0.f;
  for (int b0 = r0; b0 < r1; b0 += 16) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = (b0 + q < r1) ? partial[(size_t)(b0 + q) * SC_K * SC_CO + i] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q];
  }
  stage[(size_t)blockIdx.y * SC_K * SC_CO + i] = s;
}

// Pooling backward, dropout on the pooled gradient and the ReLU backward of the convolution below the pool in ONE pass,
// written in the operand dtype:  dpre[n,h,w,c] = (act > 0 && arg[o] == k) ? dout[o] * mask(o) : 0,  o = the pooled cell.
// Separately (asr_dropout_apply -> asr_maxpool2x2_bwd -> asr_relu_bwd) the full-resolution fp32 gradient is written and
// read back: 21.6 GB against 8.6 GB at the first pool of cfg C, and the scalar kernels ran at 1-2 TB/s (9 ms -> ~2).
// Four channels per thread (C % 4 == 0): one 16-byte load of dout, 4 bytes of arg, 8 / 16 bytes of act.
__device__ __forceinline__ void vgg_philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
template <typename T>
__global__ void maxpool_relu_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ arg,
                                        const T* __restrict__ act, int N, int H, int W, int C, T* __restrict__ dpre,
                                        float keep, uint64_t seed, uint64_t offset, int use_drop) {
  // use_drop == 2: `act` is the POOLED activation AFTER its dropout, [N,Ho,Wo,C] -- it is > 0 exactly where the window's
  // maximum was active and the mask kept it, so the ReLU test, the mask and its 1 / keep come from one 8-byte read per
  // pooled cell (shared by the window's four positions) instead of an 8-byte read of the full-resolution ReLU output per
  // position plus a Philox block; keep == 1: no dropout was applied, the same test still holds
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = C / 4;
  const size_t total = (size_t)N * H * W * C4;
  const float inv = use_drop ? 1.f / keep : 1.f;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    int c4, w, h;
    size_t n;
    nhwc4_split(idx, total <= 0xFFFFFFFFull, C4, W, H, c4, w, h, n);
    const size_t o = ((n * Ho + h / 2) * Wo + w / 2) * C + (size_t)c4 * 4;
    const int k = ((h & 1) << 1) | (w & 1);
    const f32x4_t g = *reinterpret_cast<const f32x4_t*>(dout + o);
    const uint32_t a4 = *reinterpret_cast<const uint32_t*>(arg + o);
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (use_drop == 2) {
      float pv[4];
      if (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
        const us4_t x = *reinterpret_cast<const us4_t*>(act + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = bf16_to_f32(x[j]);
      } else {
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(act + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = x[j];
      }
      float r2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r2[j] = (pv[j] > 0.f && (int)((a4 >> (8 * j)) & 0xFFu) == k) ? g[j] * inv : 0.f;
      const size_t e2 = idx * 4;
      if (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
        *reinterpret_cast<us4_t*>(dpre + e2) = (us4_t){f32_to_bf16(r2[0]), f32_to_bf16(r2[1]), f32_to_bf16(r2[2]), f32_to_bf16(r2[3])};
      } else {
        *reinterpret_cast<f32x4_t*>(dpre + e2) = (f32x4_t){r2[0], r2[1], r2[2], r2[3]};
      }
      continue;
    }
    if (use_drop) {                                       // o % 4 == 0: exactly the Philox block of the pooled cell
      const uint64_t ctr = offset + o / 4;
      uint32_t cw[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
      vgg_philox(cw, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = ((cw[j] >> 8) * (1.0f / 16777216.0f) < keep) ? inv : 0.f;
    }
    const size_t e = idx * 4;
    float av[4], r[4];
    if (sizeof(T) == 2) {
      typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
      const us4_t x = *reinterpret_cast<const us4_t*>(act + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) av[j] = bf16_to_f32(x[j]);
    } else {
      const f32x4_t x = *reinterpret_cast<const f32x4_t*>(act + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) av[j] = x[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (av[j] > 0.f && (int)((a4 >> (8 * j)) & 0xFFu) == k) ? g[j] * m[j] : 0.f;
    if (sizeof(T) == 2) {
      typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
      us4_t y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = f32_to_bf16(r[j]);
      *reinterpret_cast<us4_t*>(dpre + e) = y;
    } else {
      *reinterpret_cast<f32x4_t*>(dpre + e) = (f32x4_t){r[0], r[1], r[2], r[3]};
    }
  }
}

// dpre = dout * (out > 0) (* mask), written in the MFMA operand dtype
template <typename T>
__global__ void relu_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ out,
                                const float* __restrict__ mask, size_t n, T* __restrict__ dpre) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float g = dout[i];
    if (mask) g *= mask[i];
    dpre[i] = Elem<T>::from_f32(Elem<T>::to_f32(out[i]) > 0.f ? g : 0.f);
  }
}

// dpre = (out > 0) ? dout * (1 / keep) : 0 -- `out` is a DROPPED ReLU output (> 0 exactly where active and kept); the scale is
// formed here as 1.f / keep like in every kernel that applies the mask, so the product is the same float
template <typename T>
__global__ void relu_bwd_scaled_kernel(const float* __restrict__ dout, const T* __restrict__ out, float keep, size_t n,
                                       T* __restrict__ dpre) {
  const float scale = 1.f / keep;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dpre[i] = Elem<T>::from_f32(Elem<T>::to_f32(out[i]) > 0.f ? dout[i] * scale : 0.f);
}

inline int gridv(size_t n) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

}  // namespace

#define VGG_NEED(cond, ...) do { if (!(cond)) ASR_FAIL(h, ASR_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

extern "C" int asr_im2col3x3(asr_handle* h, int dtype, const void* in, int N, int H, int W, int Cin, int ldp,
                             void* patches, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(asr_dtype_ok(dtype) && in && patches && N >= 0 && H > 0 && W > 0 && Cin > 0 && ldp >= 9 * Cin,
           "asr_im2col3x3: bad args");
  const size_t total = (size_t)N * H * W * 9 * Cin;
  if (!total) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL(im2col3x3_kernel<float>, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, (const float*)in, N, H, W, Cin, ldp, (float*)patches);
  else hipLaunchKernelGGL(im2col3x3_kernel<bf16_t>, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, N, H, W, Cin, ldp, (bf16_t*)patches);
  ASR_CHECK_LAUNCH(h, "asr_im2col3x3");
  return ASR_OK;
}
extern "C" int asr_col2im3x3(asr_handle* h, const float* dpatches, int N, int H, int W, int Cin, int ldp,
                             float* din, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(dpatches && din && N >= 0 && H > 0 && W > 0 && Cin > 0 && ldp >= 9 * Cin, "asr_col2im3x3: bad args");
  const size_t total = (size_t)N * H * W * Cin;
  if (!total) return ASR_OK;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, dpatches, N, H, W, Cin, ldp, din);
  ASR_CHECK_LAUNCH(h, "asr_col2im3x3");
  return ASR_OK;
}
extern "C" int asr_maxpool2x2_fwd(asr_handle* h, int dtype, const void* in, int N, int H, int W, int C, void* out,
                                  uint8_t* argmax, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(asr_dtype_ok(dtype) && in && out && argmax && N >= 0 && H > 0 && W > 0 && C > 0, "asr_maxpool2x2_fwd: bad args");
  const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
  if (!total) return ASR_OK;
  if (C % 4 == 0 && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)argmax) % 16) == 0) {
    if (dtype == ASR_F32) hipLaunchKernelGGL(maxpool_fwd_vec_kernel<float>, dim3(gridv(total / 4)), dim3(256), 0, (hipStream_t)s, (const float*)in, N, H, W, C, (float*)out, argmax);
    else hipLaunchKernelGGL(maxpool_fwd_vec_kernel<bf16_t>, dim3(gridv(total / 4)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, N, H, W, C, (bf16_t*)out, argmax);
    ASR_CHECK_LAUNCH(h, "asr_maxpool2x2_fwd");
    return ASR_OK;
  }
  if (dtype == ASR_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, (const float*)in, N, H, W, C, (float*)out, argmax);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, N, H, W, C, (bf16_t*)out, argmax);
  ASR_CHECK_LAUNCH(h, "asr_maxpool2x2_fwd");
  return ASR_OK;
}
extern "C" int asr_maxpool2x2_fwd_drop(asr_handle* h, int dtype, const void* in, int N, int H, int W, int C, void* out,
                                       uint8_t* argmax, float keep_prob, uint64_t seed, uint64_t offset, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(asr_dtype_ok(dtype) && in && out && argmax && N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 &&
               keep_prob > 0.f && keep_prob <= 1.f && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)argmax) % 16) == 0,
           "asr_maxpool2x2_fwd_drop: bad args (C %% 4 == 0, 16-byte aligned arrays)");
  const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
  if (!total) return ASR_OK;
  if (dtype == ASR_F32)
    hipLaunchKernelGGL(maxpool_fwd_vec_kernel<float>, dim3(gridv(total / 4)), dim3(256), 0, (hipStream_t)s, (const float*)in, N, H,
                       W, C, (float*)out, argmax, keep_prob, seed, offset, 1);
  else
    hipLaunchKernelGGL(maxpool_fwd_vec_kernel<bf16_t>, dim3(gridv(total / 4)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, N,
                       H, W, C, (bf16_t*)out, argmax, keep_prob, seed, offset, 1);
  ASR_CHECK_LAUNCH(h, "asr_maxpool2x2_fwd_drop");
  return ASR_OK;
}
extern "C" int asr_relu_bwd_scaled(asr_handle* h, int dtype, const float* dout, const void* out, size_t n, float keep,
                                   void* dpre, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(asr_dtype_ok(dtype) && dout && out && dpre && keep > 0.f && keep <= 1.f, "asr_relu_bwd_scaled: bad args");
  if (!n) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL(relu_bwd_scaled_kernel<float>, dim3(gridv(n)), dim3(256), 0, (hipStream_t)s, dout, (const float*)out, keep, n, (float*)dpre);
  else hipLaunchKernelGGL(relu_bwd_scaled_kernel<bf16_t>, dim3(gridv(n)), dim3(256), 0, (hipStream_t)s, dout, (const bf16_t*)out, keep, n, (bf16_t*)dpre);
  ASR_CHECK_LAUNCH(h, "asr_relu_bwd_scaled");
  return ASR_OK;
}

extern "C" int asr_maxpool2x2_bwd(asr_handle* h, const float* dout, const uint8_t* argmax, int N, int H, int W,
                                  int C, float* din, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(dout && argmax && din && N >= 0 && H > 0 && W > 0 && C > 0, "asr_maxpool2x2_bwd: bad args");
  const size_t total = (size_t)N * H * W * C;
  if (!total) return ASR_OK;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, dout, argmax, N, H, W, C, din);
  ASR_CHECK_LAUNCH(h, "asr_maxpool2x2_bwd");
  return ASR_OK;
}
static int smallc_fwd_launch(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* w2d, const float* bias,
                             int Cout, int relu, void* out, float keep, uint64_t seed, uint64_t offset, int use_drop,
                             asr_stream s);
extern "C" int asr_conv3x3_smallc_fwd(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* w2d,
                                     const float* bias, int Cout, int relu, void* out, asr_stream s) {
  return smallc_fwd_launch(h, x, N, H, W, Cin, w2d, bias, Cout, relu, out, 1.f, 0, 0, 0, s);
}
extern "C" int asr_conv3x3_smallc_fwd_drop(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* w2d,
                                          const float* bias, int Cout, float keep_prob, uint64_t seed, uint64_t offset,
                                          void* out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(keep_prob > 0.f && keep_prob <= 1.f, "asr_conv3x3_smallc_fwd_drop: keep_prob");
  return smallc_fwd_launch(h, x, N, H, W, Cin, w2d, bias, Cout, 1, out, keep_prob, seed, offset, 1, s);
}
static int smallc_fwd_launch(asr_handle* h, const void* x, int N, int H, int W, int Cin, const void* w2d, const float* bias,
                             int Cout, int relu, void* out, float keep, uint64_t seed, uint64_t offset, int use_drop,
                             asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(x && w2d && out && N >= 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 3 && Cout == SC_CO &&
               ((uintptr_t)out) % 16 == 0, "asr_conv3x3_smallc_fwd: needs Cin <= 3, Cout == 64");
  const size_t npix = (size_t)N * H * W;
  if (!npix) return ASR_OK;
#define ASR_SC_FWD(C_) \
  hipLaunchKernelGGL(conv3x3_smallc_fwd_kernel<C_>, dim3((unsigned)((npix + 63) / 64)), dim3(256), 0, (hipStream_t)s, \
                     (const bf16_t*)x, npix, H, W, (const bf16_t*)w2d, bias, relu, (bf16_t*)out)
  // matrix-core form (see the kernel); ASR_SMALLC_MFMA=0 keeps the vector-ALU kernels (A/B)
  static const bool mfma = [] { const char* e = getenv("ASR_SMALLC_MFMA"); return !(e && e[0] == '0'); }();
#define ASR_SC_FWD_M(C_) \
  hipLaunchKernelGGL(conv3x3_smallc_fwd_mfma_kernel<C_>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)s, \
                     (const bf16_t*)x, npix, H, W, (const bf16_t*)w2d, bias, relu, (bf16_t*)out, keep, seed, offset, use_drop)
  if (use_drop && !mfma) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3_smallc_fwd_drop needs the matrix-core kernel (ASR_SMALLC_MFMA)");
  if (mfma) { if (Cin == 1) ASR_SC_FWD_M(1); else if (Cin == 2) ASR_SC_FWD_M(2); else ASR_SC_FWD_M(3); }
  else if (Cin == 1) ASR_SC_FWD(1); else if (Cin == 2) ASR_SC_FWD(2); else ASR_SC_FWD(3);
#undef ASR_SC_FWD_M
#undef ASR_SC_FWD
  ASR_CHECK_LAUNCH(h, "asr_conv3x3_smallc_fwd");
  return ASR_OK;
}
extern "C" int asr_colsum(asr_handle* h, int dtype, const void* a, int M, int N, int lda, float* out, asr_stream s);
static int smallc_bwd_weight_impl(asr_handle* h, const void* x, const void* dpre, int N, int H, int W, int Cin, int Cout,
                                  float* dw, float* dbias, asr_stream s);
extern "C" int asr_conv3x3_smallc_bwd_weight(asr_handle* h, const void* x, const void* dpre, int N, int H, int W, int Cin,
                                            int Cout, float* dw, asr_stream s) {
  return smallc_bwd_weight_impl(h, x, dpre, N, H, W, Cin, Cout, dw, nullptr, s);
}
extern "C" int asr_conv3x3_smallc_bwd_weight_bias(asr_handle* h, const void* x, const void* dpre, int N, int H, int W,
                                                 int Cin, int Cout, float* dw, float* dbias, asr_stream s) {
  if (h && !dbias) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_smallc_bwd_weight_bias: dbias is NULL");
  return smallc_bwd_weight_impl(h, x, dpre, N, H, W, Cin, Cout, dw, dbias, s);
}
static int smallc_bwd_weight_impl(asr_handle* h, const void* x, const void* dpre, int N, int H, int W, int Cin, int Cout,
                                  float* dw, float* dbias, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(x && dpre && dw && N >= 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 3 && Cout == SC_CO &&
               ((uintptr_t)dpre) % 16 == 0, "asr_conv3x3_smallc_bwd_weight: needs Cin <= 3, Cout == 64");
  const size_t npix = (size_t)N * H * W;
  int nblk = (int)((npix + 2047) / 2048);
  if (nblk > 2048) nblk = 2048;
  if (nblk < 1) nblk = 1;
  size_t per = (npix + nblk - 1) / nblk;
  per = (per + 63) / 64 * 64;
  nblk = (int)((npix + per - 1) / per);
  if (nblk < 1) nblk = 1;
  const int nseg = (nblk + SC_SEGR - 1) / SC_SEGR;
  const size_t need = ((size_t)nblk + nseg) * SC_K * SC_CO * sizeof(float);
  if (need > h->scratch_bytes - ASR_XCH_BYTES) ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_conv3x3_smallc_bwd_weight: scratch too small");
  float* partial = (float*)h->scratch;
  float* stage = partial + (size_t)nblk * SC_K * SC_CO;
#define ASR_SC_WG(C_) \
  hipLaunchKernelGGL(conv3x3_smallc_wgrad_kernel<C_>, dim3(nblk), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, \
                     (const bf16_t*)dpre, npix, H, W, per, partial)
  static const bool mfma = [] { const char* e = getenv("ASR_SMALLC_MFMA"); return !(e && e[0] == '0'); }();
#define ASR_SC_WG_M(C_) \
  hipLaunchKernelGGL(conv3x3_smallc_wgrad_mfma_kernel<C_>, dim3(nblk), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, \
                     (const bf16_t*)dpre, npix, H, W, per, partial)
  if (mfma) { if (Cin == 1) ASR_SC_WG_M(1); else if (Cin == 2) ASR_SC_WG_M(2); else ASR_SC_WG_M(3); }
  else if (Cin == 1) ASR_SC_WG(1); else if (Cin == 2) ASR_SC_WG(2); else ASR_SC_WG(3);
#undef ASR_SC_WG_M
#undef ASR_SC_WG
  static const bool wbias_on = [] { const char* e = getenv("ASR_CONV_WGRAD_BIAS"); return !(e && e[0] == '0'); }();
  float* inb = (dbias && mfma && wbias_on) ? dbias : nullptr;      // bias gradient = row 9 Cin of the matrix-core kernel's product
  const int ncols = (9 * Cin + 1) * SC_CO;
  if (nblk > 2 * SC_SEGR) {                                 // many partials: segment sums first
    hipLaunchKernelGGL(conv3x3_smallc_wgrad_stage_kernel, dim3((ncols + 255) / 256, nseg), dim3(256), 0, (hipStream_t)s,
                       partial, nblk, ncols, stage);
    hipLaunchKernelGGL(conv3x3_smallc_wgrad_reduce_kernel, dim3((ncols + 255) / 256), dim3(256), 0, (hipStream_t)s, stage,
                       nseg, 9 * Cin, dw, inb);
  } else {
    hipLaunchKernelGGL(conv3x3_smallc_wgrad_reduce_kernel, dim3((ncols + 255) / 256), dim3(256), 0, (hipStream_t)s, partial,
                       nblk, 9 * Cin, dw, inb);
  }
  ASR_CHECK_LAUNCH(h, "asr_conv3x3_smallc_bwd_weight");
  if (dbias && !inb) return asr_colsum(h, ASR_BF16, dpre, (int)npix, Cout, Cout, dbias, s);
  return ASR_OK;
}
extern "C" int asr_maxpool2x2_relu_bwd(asr_handle* h, int dtype, const float* dout, const uint8_t* argmax,
                                      const void* act, int N, int H, int W, int C, void* dpre, float keep_prob,
                                      uint64_t seed, uint64_t offset, int use_drop, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(asr_dtype_ok(dtype) && dout && argmax && act && dpre && N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 &&
               (!use_drop || (keep_prob > 0.f && keep_prob <= 1.f)) && use_drop >= 0 && use_drop <= 2 &&
               (((uintptr_t)dout | (uintptr_t)argmax | (uintptr_t)act | (uintptr_t)dpre) % 16) == 0,
           "asr_maxpool2x2_relu_bwd: bad args (C %% 4 == 0, 16-byte aligned arrays)");
  const size_t total = (size_t)N * H * W * (C / 4);
  if (!total) return ASR_OK;
  if (dtype == ASR_F32)
    hipLaunchKernelGGL(maxpool_relu_bwd_kernel<float>, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, dout, argmax,
                       (const float*)act, N, H, W, C, (float*)dpre, keep_prob, seed, offset, use_drop);
  else
    hipLaunchKernelGGL(maxpool_relu_bwd_kernel<bf16_t>, dim3(gridv(total)), dim3(256), 0, (hipStream_t)s, dout, argmax,
                       (const bf16_t*)act, N, H, W, C, (bf16_t*)dpre, keep_prob, seed, offset, use_drop);
  ASR_CHECK_LAUNCH(h, "asr_maxpool2x2_relu_bwd");
  return ASR_OK;
}
extern "C" int asr_relu_bwd(asr_handle* h, int dtype, const float* dout, const void* out, const float* mask,
                            size_t n, void* dpre, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  VGG_NEED(asr_dtype_ok(dtype) && dout && out && dpre, "asr_relu_bwd: bad args");
  if (!n) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(gridv(n)), dim3(256), 0, (hipStream_t)s, dout, (const float*)out, mask, n, (float*)dpre);
  else hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(gridv(n)), dim3(256), 0, (hipStream_t)s, dout, (const bf16_t*)out, mask, n, (bf16_t*)dpre);
  ASR_CHECK_LAUNCH(h, "asr_relu_bwd");
  return ASR_OK;
}

static int conv_geo(asr_handle* h, int N, int H, int W, int Cin, int kh, int kw, int sh, int sw, ConvGeo* g) {
  if (N < 0 || H < 1 || W < 1 || Cin < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_im2col: bad geometry");
  g->N = N; g->H = H; g->W = W; g->Cin = Cin; g->kh = kh; g->kw = kw; g->sh = sh; g->sw = sw;
  g->Ho = (H + sh - 1) / sh;
  g->Wo = (W + sw - 1) / sw;
  const int ph = (g->Ho - 1) * sh + kh - H, pw = (g->Wo - 1) * sw + kw - W;
  g->pt = (ph > 0 ? ph : 0) / 2;
  g->pl = (pw > 0 ? pw : 0) / 2;
  return ASR_OK;
}
extern "C" int asr_im2col(asr_handle* h, int dtype, const void* in, int N, int H, int W, int Cin, int kh, int kw, int sh,
                          int sw, int ldp, void* patches, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ConvGeo g;
  const int rc = conv_geo(h, N, H, W, Cin, kh, kw, sh, sw, &g);
  if (rc != ASR_OK) return rc;
  if (!asr_dtype_ok(dtype) || !in || !patches || ldp < kh * kw * Cin) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_im2col: bad args");
  const size_t total = (size_t)N * g.Ho * g.Wo * kh * kw * Cin;
  if (!total) return ASR_OK;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == ASR_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)in, g, ldp, (float*)patches);
  else hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, g, ldp, (bf16_t*)patches);
  ASR_CHECK_LAUNCH(h, "asr_im2col");
  return ASR_OK;
}
extern "C" int asr_col2im(asr_handle* h, const float* dpatches, int N, int H, int W, int Cin, int kh, int kw, int sh,
                          int sw, int ldp, float* din, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ConvGeo g;
  const int rc = conv_geo(h, N, H, W, Cin, kh, kw, sh, sw, &g);
  if (rc != ASR_OK) return rc;
  if (!dpatches || !din || ldp < kh * kw * Cin) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_col2im: bad args");
  const size_t total = (size_t)N * H * W * Cin;
  if (!total) return ASR_OK;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(col2im_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, dpatches, g, ldp, din);
  ASR_CHECK_LAUNCH(h, "asr_col2im");
  return ASR_OK;
}

