// GRU recurrence for gfx950: tf.contrib.rnn.GRUCell under tf.nn.(bidirectional_)dynamic_rnn(sequence_length)
// (models/encoders/core/gru.py:58-76 GRUEncoder, :126-152 BGRUEncoder of the reference):
//     [r, u] = sigmoid([x, h] W_g + b_g)        W_g [(D+H), 2H], gate columns r | u, b_g initialised to 1
//     c      = tanh([x, r * h] W_c + b_c)       W_c [(D+H), H]
//     h'     = u * h + (1 - u) * c              state carried / output zeroed past seq_len
//
// The x-parts (x W_gx + b_g, x W_cx + b_c) are hoisted over all T into two GEMMs (asr_gemm), as for the LSTM.  What is
// left per step is two DEPENDENT small matrix products (h W_gh, then (r*h) W_ch): the reset gate sits between them, so
// one step is two grid-wide phases.  Each phase is one launch over (16-row tile of utterances) x (16-column tile of
// units) x direction: the 16 x H slice of h (or r*h) is staged in LDS, every thread owns one (utterance, unit) output
// and walks K = H with the weight column coalesced across the 16 unit-threads.  fp32 throughout (this encoder is not
// on a BASELINE configuration; the LSTM path is where the MFMA / multi-CU work went).  Launch-bound: ~2 x 5 us per
// step forward, 3 launches per step backward.
//
// Time indexing as dynamic_rnn: at recurrence step s direction 0 works on frame s, direction 1 on frame len_b-1-s
// (array_ops.reverse_sequence); rows with s >= len_b keep their state and emit zeros (written to frame s of the
// padding, as the LSTM kernels do).
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float gsig(float x) { return 1.0f / (1.0f + expf(-x)); }

// frame a row works on at step s; act = the row is still inside its utterance
__device__ __forceinline__ int gru_frame(int s, int len, int d, bool& act) {
  act = s < len;
  return act ? (d == 1 ? len - 1 - s : s) : s;
}

// ---- forward phase 1: r, u, r*h -------------------------------------------------------------------------------------
// grid (2H/16, B/16, ndir), block 256 = 16 rows x 16 columns.  xg [T,B,ndir,2H]; hstate [ndir,B,H] (h of step s-1);
// wgh [ndir][H][2H]; outputs r, u, rh at the row's frame: [T,B,ndir,H] each.
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(int s, int T, int B, int H, int ndir,
                                                            const float* __restrict__ xg,
                                                            const float* __restrict__ hstate,
                                                            const float* __restrict__ wgh,
                                                            const int32_t* __restrict__ seq_len,
                                                            float* __restrict__ r_out, float* __restrict__ u_out,
                                                            float* __restrict__ rh_out) {
  extern __shared__ float hs[];                            // [16][H + 1]
  const int d = blockIdx.z, b0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
  const int LD = H + 1;
  const float* hsrc = hstate + ((size_t)d * B + b0) * H;
  for (int i = threadIdx.x; i < 16 * H; i += 256) hs[(i / H) * LD + (i % H)] = hsrc[i];
  __syncthreads();
  const int b = b0 + row, j = c0 + col;                    // j in [0, 2H): gate column
  bool act;
  const int t = gru_frame(s, min(seq_len[b], T), d, act);
  if (!act) return;                                        // nothing of an inactive row is read later
  const float* w = wgh + (size_t)d * H * 2 * H + j;
  float acc = xg[(((size_t)t * B + b) * ndir + d) * 2 * H + j];
  const float* hr = hs + row * LD;
#pragma unroll 8
  for (int k = 0; k < H; ++k) acc = fmaf(hr[k], w[(size_t)k * 2 * H], acc);
  const float g = gsig(acc);
  const size_t o = (((size_t)t * B + b) * ndir + d) * H;
  if (j < H) {
    r_out[o + j] = g;
    rh_out[o + j] = g * hr[j];
  } else {
    u_out[o + j - H] = g;
  }
}

// ---- forward phase 2: candidate, new state, output ------------------------------------------------------------------
// grid (H/16, B/16, ndir).  xc [T,B,ndir,H]; rh, u [T,B,ndir,H] (phase 1); wch [ndir][H][H]; hstate -> hnext
// [ndir,B,H]; c_out [T,B,ndir,H]; hout [T,B,ndir*H] (fw | bw halves, zero for finished rows).
__global__ __launch_bounds__(256) void gru_cand_fwd_kernel(int s, int T, int B, int H, int ndir,
                                                           const float* __restrict__ xc,
                                                           const float* __restrict__ rh,
                                                           const float* __restrict__ u_in,
                                                           const float* __restrict__ wch,
                                                           const float* __restrict__ hstate,
                                                           const int32_t* __restrict__ seq_len,
                                                           float* __restrict__ hnext, float* __restrict__ c_out,
                                                           float* __restrict__ hout) {
  extern __shared__ float as[];                            // [16][H + 1]: r*h of the 16 rows at their own frames
  const int d = blockIdx.z, b0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
  const int LD = H + 1;
  for (int i = threadIdx.x; i < 16 * H; i += 256) {
    const int rr = i / H, k = i % H;
    bool a;
    const int tt = gru_frame(s, min(seq_len[b0 + rr], T), d, a);
    as[rr * LD + k] = a ? rh[(((size_t)tt * B + b0 + rr) * ndir + d) * H + k] : 0.f;
  }
  __syncthreads();
  const int b = b0 + row, j = c0 + col;
  bool act;
  const int t = gru_frame(s, min(seq_len[b], T), d, act);
  const size_t so = ((size_t)d * B + b) * H + j;
  const float hp = hstate[so];
  const size_t oo = ((size_t)t * B + b) * ndir * H + (size_t)d * H + j;
  if (!act) {
    hnext[so] = hp;
    hout[oo] = 0.f;
    return;
  }
  const size_t o = (((size_t)t * B + b) * ndir + d) * H + j;
  const float* w = wch + (size_t)d * H * H + j;
  float acc = xc[o];
  const float* ar = as + row * LD;
#pragma unroll 8
  for (int k = 0; k < H; ++k) acc = fmaf(ar[k], w[(size_t)k * H], acc);
  const float c = tanhf(acc), u = u_in[o];
  const float hn = u * hp + (1.f - u) * c;
  c_out[o] = c;
  hnext[so] = hn;
  hout[oo] = hn;
}

// ---- backward, per step (steps run tmax-1 .. 0) ----------------------------------------------------------------------
// B1 (elementwise over [ndir,B,H]): dh = dout(frame) + dh_rec;  du_pre, dc_pre to dgc / dcc at the row's frame,
//    dh_acc = dh * u.   h_prev of step s is hout at the frame of step s-1 (zero for s == 0).
__global__ void gru_bwd_elem_kernel(int s, int T, int B, int H, int ndir, const float* __restrict__ dout,
                                    const float* __restrict__ dh_rec, const float* __restrict__ hout,
                                    const float* __restrict__ u_in, const float* __restrict__ c_in,
                                    const int32_t* __restrict__ seq_len, float* __restrict__ dgate,
                                    float* __restrict__ dcand, float* __restrict__ dh_acc) {
  const size_t n = (size_t)ndir * B * H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i % H, b = (i / H) % B, d = i / ((size_t)H * B);
    const int len = min(seq_len[b], T);
    bool act;
    const int t = gru_frame(s, len, d, act);
    const float dr = dh_rec[i];
    if (!act) { dh_acc[i] = dr; continue; }
    const size_t o = (((size_t)t * B + b) * ndir + d) * H + j;
    float hp = 0.f;
    if (s > 0) {
      const int tp = d == 1 ? len - s : s - 1;             // frame of step s-1
      hp = hout[((size_t)tp * B + b) * ndir * H + (size_t)d * H + j];
    }
    const float dh = dout[((size_t)t * B + b) * ndir * H + (size_t)d * H + j] + dr;
    const float u = u_in[o], c = c_in[o];
    dgate[(((size_t)t * B + b) * ndir + d) * 2 * H + H + j] = dh * (hp - c) * u * (1.f - u);   // d u_pre
    dcand[o] = dh * (1.f - u) * (1.f - c * c);                                                   // d c_pre
    dh_acc[i] = dh * u;
  }
}

// B2: d(rh)[row][k] = sum_j dc_pre[row][j] W_c[k][j] (wchT [ndir][H j][H k]);  d r_pre -> dgate[..][k],
//     dh_acc[k] += d(rh) * r.   grid (H/16, B/16, ndir).
__global__ __launch_bounds__(256) void gru_bwd_reset_kernel(int s, int T, int B, int H, int ndir,
                                                            const float* __restrict__ dcand,
                                                            const float* __restrict__ wchT,
                                                            const float* __restrict__ hout,
                                                            const float* __restrict__ r_in,
                                                            const int32_t* __restrict__ seq_len,
                                                            float* __restrict__ dgate, float* __restrict__ dh_acc) {
  extern __shared__ float ds[];                            // [16][H + 1]: dc_pre of the rows' frames
  const int d = blockIdx.z, b0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
  const int LD = H + 1;
  for (int i = threadIdx.x; i < 16 * H; i += 256) {
    const int rr = i / H, k = i % H;
    bool a;
    const int tt = gru_frame(s, min(seq_len[b0 + rr], T), d, a);
    ds[rr * LD + k] = a ? dcand[(((size_t)tt * B + b0 + rr) * ndir + d) * H + k] : 0.f;
  }
  __syncthreads();
  const int b = b0 + row, k = c0 + col;
  const int len = min(seq_len[b], T);
  bool act;
  const int t = gru_frame(s, len, d, act);
  if (!act) return;
  const float* w = wchT + (size_t)d * H * H + k;
  const float* dr = ds + row * LD;
  float acc = 0.f;
#pragma unroll 8
  for (int j = 0; j < H; ++j) acc = fmaf(dr[j], w[(size_t)j * H], acc);
  float hp = 0.f;
  if (s > 0) {
    const int tp = d == 1 ? len - s : s - 1;
    hp = hout[((size_t)tp * B + b) * ndir * H + (size_t)d * H + k];
  }
  const size_t o = (((size_t)t * B + b) * ndir + d) * H + k;
  const float r = r_in[o];
  dgate[(((size_t)t * B + b) * ndir + d) * 2 * H + k] = acc * hp * r * (1.f - r);   // d r_pre
  dh_acc[((size_t)d * B + b) * H + k] += acc * r;
}

// B3: dh_rec'[row][k] = dh_acc[row][k] + sum_j dgate[row][j] W_g[k][j], j over 2H (wghT [ndir][2H j][H k]).
__global__ __launch_bounds__(256) void gru_bwd_state_kernel(int s, int T, int B, int H, int ndir,
                                                            const float* __restrict__ dgate,
                                                            const float* __restrict__ wghT,
                                                            const float* __restrict__ dh_acc,
                                                            const int32_t* __restrict__ seq_len,
                                                            float* __restrict__ dh_rec) {
  extern __shared__ float gs[];                            // [16][2H + 1]
  const int d = blockIdx.z, b0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
  const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
  const int G = 2 * H, LD = G + 1;
  for (int i = threadIdx.x; i < 16 * G; i += 256) {
    const int rr = i / G, j = i % G;
    bool a;
    const int tt = gru_frame(s, min(seq_len[b0 + rr], T), d, a);
    gs[rr * LD + j] = a ? dgate[(((size_t)tt * B + b0 + rr) * ndir + d) * G + j] : 0.f;
  }
  __syncthreads();
  const int b = b0 + row, k = c0 + col;
  const size_t so = ((size_t)d * B + b) * H + k;
  const float* w = wghT + (size_t)d * G * H + k;
  const float* gr = gs + row * LD;
  float acc = dh_acc[so];
#pragma unroll 8
  for (int j = 0; j < G; ++j) acc = fmaf(gr[j], w[(size_t)j * H], acc);
  dh_rec[so] = acc;
}


// =====================================================================================================================
// PERSISTENT form: ONE launch per layer -- one workgroup per (16-utterance tile, direction) walks all steps with the
// 16 x H state in LDS and both dependent products on the matrix cores (exact-fp32 MFMA, v_mfma_f32_16x16x4_f32):
//     phase 1   [r | u] = sigmoid(h W_gh + xg)          16 x 2H, K = H      -> r, u, r * h
//     phase 2   c = tanh((r * h) W_ch + xc),  h' = u h + (1 - u) c          16 x H, K = H
// with one barrier between the phases and one behind them.  The launch-per-step form above (2 / 3 launches per time
// step, scalar FMAs) spends its time in launch latency: ~10 us per step whatever H is; here a step costs its matrix
// work -- 16 x 3 H^2 MACs at the CU's 256 flop/clk fp32 matrix rate = 0.16 us (H = 64) / 0.64 us (H = 128) /
// 2.6 us (H = 256 per product pair streamed from L2) -- and the weights stream from L2 in MFMA-fragment order
// (gru_pack_kernel: a lane's operands of four consecutive k-steps are one 16-byte load).
// MFMA operand layouts (16x16x4 f32): A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// C[m = (lane >> 4) * 4 + i][n = lane & 15].  k is walked in groups of 16: group jj, MFMA j, lane part kq -> k =
// 16 jj + 4 j + kq; LDS rows are stored permuted ([jj][kq][j]) so that a lane's four A operands of a group are 16 bytes.
constexpr int GP_THREADS = 512, GP_WAVES = 8;

// src [K][N] row-major (row stride ld) -> dst [N/16][K/16][64 lanes][4]:  B fragments of column tile tn, k-group jj
__global__ void gru_pack_kernel(const float* __restrict__ src, int K, int N, int ld, float* __restrict__ dst, int nmat,
                                size_t src_stride, size_t dst_stride) {
  const size_t per = (size_t)K * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per * nmat; i += (size_t)gridDim.x * blockDim.x) {
    const int mat = (int)(i / per);
    const size_t e = i % per;
    const int j = e & 3, lane = (e >> 2) & 63;
    const size_t g = e >> 8;                                // tn * (K/16) + jj
    const int jj = (int)(g % (K / 16)), tn = (int)(g / (K / 16));
    const int k = 16 * jj + 4 * j + (lane >> 4), n = tn * 16 + (lane & 15);
    dst[mat * dst_stride + e] = src[mat * src_stride + (size_t)k * ld + n];
  }
}

// LDS position (in floats) of element (row m, k) of a permuted 16 x H image with row stride LD
__device__ __forceinline__ int gp_pos(int m, int k, int LD) {
  return m * LD + (k & ~15) + ((k & 3) << 2) + ((k >> 2) & 3);
}

// one 16 x 16 output tile: acc += A(LDS image, all K) * B(packed fragments of this column tile).  K is walked in chunks
// of 8 groups (128 k): the chunk's eight B fragments (8 x 16 bytes per lane, from L2) are all requested before its 32
// MFMAs -- with the loads issued one group ahead of their use a tile waited for one L2 round trip per four groups
// (55 us per step and layer at H = 256; the matrix work is 20).  The second wave of the SIMD fills what remains.
__device__ __forceinline__ f32x4_t gp_tile(const float* __restrict__ img, int LD, const float* __restrict__ wpk, int KG,
                                           int lane) {
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  const float* ap = img + (lane & 15) * LD + ((lane >> 4) << 2);
  const f32x4_t* bp = reinterpret_cast<const f32x4_t*>(wpk) + lane;
  int jj = 0;
  for (; jj + 8 <= KG; jj += 8) {
    f32x4_t b[8], a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = bp[(size_t)(jj + q) * 64];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const f32x4_t*>(ap + (jj + q) * 16);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], b[q][j], acc, 0, 0, 0);
  }
  for (; jj < KG; ++jj) {
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ap + jj * 16);
    const f32x4_t b = bp[(size_t)jj * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(GP_THREADS, 1) void gru_fwd_persistent_kernel(
    int T, int B, int H, int ndir, int tmax, const float* __restrict__ xg, const float* __restrict__ xc,
    const float* __restrict__ wg_pk, const float* __restrict__ wc_pk, const int32_t* __restrict__ seq_len,
    float* __restrict__ r_out, float* __restrict__ u_out, float* __restrict__ c_out, float* __restrict__ rh_out,
    float* __restrict__ hout, float* __restrict__ h_final) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const int LD = H + 4;
  float* hbuf = gsm;                        // [16][LD] permuted: h of the running step
  float* rhbuf = gsm + 16 * LD;             // [16][LD] permuted: r * h
  float* ubuf = gsm + 2 * 16 * LD;          // [16][LD] plain:    u
  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const int KG = H / 16, NT1 = 2 * H / 16, NT2 = H / 16;
  const float* wg = wg_pk + (size_t)d * 2 * H * H;
  const float* wc = wc_pk + (size_t)d * H * H;
  int len[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) len[i] = min(max(seq_len[b0 + rg * 4 + i], 0), T);
  for (int i = threadIdx.x; i < 16 * LD; i += GP_THREADS) hbuf[i] = 0.f;
  __syncthreads();
  for (int s = 0; s < tmax; ++s) {
    // ---- phase 1: gates
    for (int t = wave; t < NT1; t += GP_WAVES) {
      const int jcol = t * 16 + col;                       // gate column in [0, 2H)
      float xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {                        // requested ahead of the product
        const bool act = s < len[i];
        const int fr = act ? (d == 1 ? len[i] - 1 - s : s) : s;
        xv[i] = act ? xg[(((size_t)fr * B + b0 + rg * 4 + i) * ndir + d) * 2 * H + jcol] : 0.f;
      }
      const f32x4_t acc = gp_tile(hbuf, LD, wg + (size_t)t * KG * 256, KG, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rg * 4 + i, b = b0 + row;
        const bool act = s < len[i];
        const int fr = act ? (d == 1 ? len[i] - 1 - s : s) : s;
        const float g = act ? gsig(acc[i] + xv[i]) : 0.f;
        if (jcol < H) {
          const float hv = hbuf[gp_pos(row, jcol, LD)];
          rhbuf[gp_pos(row, jcol, LD)] = g * hv;
          if (act) {
            const size_t o = (((size_t)fr * B + b) * ndir + d) * H + jcol;
            r_out[o] = g;
            rh_out[o] = g * hv;
          }
        } else {
          ubuf[row * LD + jcol - H] = g;
          if (act) u_out[(((size_t)fr * B + b) * ndir + d) * H + jcol - H] = g;
        }
      }
    }
    __syncthreads();
    // ---- phase 2: candidate, new state, output.  h' is kept in registers until every wave has read h (barrier)
    float hn[(512 / 16 + GP_WAVES - 1) / GP_WAVES][4];
    int nt = 0;
    for (int t = wave; t < NT2; t += GP_WAVES, ++nt) {
      const int j = t * 16 + col;
      float xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool act = s < len[i];
        const int fr = act ? (d == 1 ? len[i] - 1 - s : s) : s;
        xv[i] = act ? xc[(((size_t)fr * B + b0 + rg * 4 + i) * ndir + d) * H + j] : 0.f;
      }
      const f32x4_t acc = gp_tile(rhbuf, LD, wc + (size_t)t * KG * 256, KG, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rg * 4 + i, b = b0 + row;
        const bool act = s < len[i];
        const int fr = act ? (d == 1 ? len[i] - 1 - s : s) : s;
        const float hp = hbuf[gp_pos(row, j, LD)];
        float hv = hp;
        if (act) {
          const size_t o = (((size_t)fr * B + b) * ndir + d) * H + j;
          const float c = tanhf(acc[i] + xv[i]);
          const float u = ubuf[row * LD + j];
          hv = u * hp + (1.f - u) * c;
          c_out[o] = c;
        }
        hn[nt][i] = hv;
        hout[((size_t)fr * B + b) * ndir * H + (size_t)d * H + j] = act ? hv : 0.f;
      }
    }
    __syncthreads();
    nt = 0;
    for (int t = wave; t < NT2; t += GP_WAVES, ++nt) {
      const int j = t * 16 + col;
#pragma unroll
      for (int i = 0; i < 4; ++i) hbuf[gp_pos(rg * 4 + i, j, LD)] = hn[nt][i];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 16 * H; i += GP_THREADS) {
    const int row = i / H, j = i % H;
    h_final[((size_t)d * B + b0 + row) * H + j] = hbuf[gp_pos(row, j, LD)];
  }
}

// Backward, persistent: per step (s = tmax-1 .. 0), for the rows still inside their utterance
//     dh = dout[frame] + dh_rec;   du_pre = dh (hp - c) u (1 - u);   dc_pre = dh (1 - u)(1 - c^2);   acc = dh u
//     d(rh) = dc_pre W_c^T  (MFMA, K = H);   dr_pre = d(rh) hp r (1 - r);   acc += d(rh) r
//     dh_rec' = acc + [dr_pre | du_pre] W_g^T  (MFMA, K = 2H)
// hp = h of the previous step = hout at the previous frame (zero at s = 0).  Rows past their length carry dh_rec.
__global__ __launch_bounds__(GP_THREADS, 1) void gru_bwd_persistent_kernel(
    int T, int B, int H, int ndir, int tmax, const float* __restrict__ dout, const float* __restrict__ d_h_final,
    const float* __restrict__ hout, const float* __restrict__ r_in, const float* __restrict__ u_in,
    const float* __restrict__ c_in, const float* __restrict__ wgT_pk, const float* __restrict__ wcT_pk,
    const int32_t* __restrict__ seq_len, float* __restrict__ dgate, float* __restrict__ dcand) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const int LD = H + 4, LD2 = 2 * H + 4;
  float* dcbuf = gsm;                       // [16][LD]  permuted: dc_pre
  float* dgbuf = gsm + 16 * LD;             // [16][LD2] permuted: [dr_pre | du_pre]
  float* drec = dgbuf + 16 * LD2;           // [16][LD]  plain:    dh_rec of the running step
  float* dacc = drec + 16 * LD;             // [16][LD]  plain:    dh u (+ d(rh) r)
  const int d = blockIdx.y, b0 = blockIdx.x * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const int KG = H / 16, NT = H / 16;
  const float* wgT = wgT_pk + (size_t)d * 2 * H * H;
  const float* wcT = wcT_pk + (size_t)d * H * H;
  int len[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) len[i] = min(max(seq_len[b0 + rg * 4 + i], 0), T);
  for (int i = threadIdx.x; i < 16 * H; i += GP_THREADS) {
    const int row = i / H, j = i % H;
    drec[row * LD + j] = d_h_final ? d_h_final[((size_t)d * B + b0 + row) * H + j] : 0.f;
  }
  __syncthreads();
  for (int s = tmax - 1; s >= 0; --s) {
    // ---- B1: elementwise, each wave the column tiles it owns
    for (int t = wave; t < NT; t += GP_WAVES) {
      const int j = t * 16 + col;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rg * 4 + i, b = b0 + row;
        const bool act = s < len[i];
        const float dr = drec[row * LD + j];
        float dcp = 0.f, dup = 0.f, da = dr;
        if (act) {
          const int fr = d == 1 ? len[i] - 1 - s : s;
          const size_t o = (((size_t)fr * B + b) * ndir + d) * H + j;
          float hp = 0.f;
          if (s > 0) hp = hout[((size_t)(d == 1 ? len[i] - s : s - 1) * B + b) * ndir * H + (size_t)d * H + j];
          const float dh = dout[((size_t)fr * B + b) * ndir * H + (size_t)d * H + j] + dr;
          const float u = u_in[o], c = c_in[o];
          dup = dh * (hp - c) * u * (1.f - u);
          dcp = dh * (1.f - u) * (1.f - c * c);
          da = dh * u;
          dcand[o] = dcp;
          dgate[(((size_t)fr * B + b) * ndir + d) * 2 * H + H + j] = dup;
        }
        dcbuf[gp_pos(row, j, LD)] = dcp;
        dgbuf[gp_pos(row, H + j, LD2)] = dup;
        dacc[row * LD + j] = da;
      }
    }
    __syncthreads();
    // ---- B2: d(rh) = dc_pre W_c^T
    for (int t = wave; t < NT; t += GP_WAVES) {
      const f32x4_t acc = gp_tile(dcbuf, LD, wcT + (size_t)t * KG * 256, KG, lane);
      const int k = t * 16 + col;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rg * 4 + i, b = b0 + row;
        const bool act = s < len[i];
        float drp = 0.f;
        if (act) {
          const int fr = d == 1 ? len[i] - 1 - s : s;
          float hp = 0.f;
          if (s > 0) hp = hout[((size_t)(d == 1 ? len[i] - s : s - 1) * B + b) * ndir * H + (size_t)d * H + k];
          const float r = r_in[(((size_t)fr * B + b) * ndir + d) * H + k];
          drp = acc[i] * hp * r * (1.f - r);
          dacc[row * LD + k] += acc[i] * r;
          dgate[(((size_t)fr * B + b) * ndir + d) * 2 * H + k] = drp;
        }
        dgbuf[gp_pos(row, k, LD2)] = drp;
      }
    }
    __syncthreads();
    // ---- B3: dh_rec' = acc + [dr_pre | du_pre] W_g^T
    for (int t = wave; t < NT; t += GP_WAVES) {
      const f32x4_t acc = gp_tile(dgbuf, LD2, wgT + (size_t)t * (2 * KG) * 256, 2 * KG, lane);
      const int k = t * 16 + col;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rg * 4 + i;
        drec[row * LD + k] = dacc[row * LD + k] + acc[i];
      }
    }
    __syncthreads();
  }
}

}  // namespace

#define GRU_NEED(cond, ...) do { if (!(cond)) ASR_FAIL(h, ASR_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

// The persistent kernels apply when the LDS images fit one CU (H <= 512 forward, H <= 480 backward) and the packed
// weights fit the handle's scratch; ASR_GRU_PERSISTENT=0 keeps the launch-per-step form (A/B, and the tests run both).
static int g_gru_persistent = -1;
extern "C" int asr_debug_set_gru_persistent(int on) { g_gru_persistent = on ? 1 : 0; return 0; }
static bool gru_persistent_ok(asr_handle* h, int H, int ndir, size_t lds) {
  if (g_gru_persistent < 0) { const char* e = getenv("ASR_GRU_PERSISTENT"); g_gru_persistent = (e && e[0] == '0') ? 0 : 1; }
  const size_t pk = (size_t)ndir * 3 * H * H * sizeof(float);
  return g_gru_persistent == 1 && H % 16 == 0 && H <= 512 && lds <= (size_t)160 * 1024 &&
         h->scratch_bytes > ASR_XCH_BYTES && pk <= h->scratch_bytes - ASR_XCH_BYTES;
}

// the clusters stand in for the persistent single-CU kernels: asr_debug_set_gru_persistent(0) / ASR_GRU_PERSISTENT=0 (the
// launch-per-step form, A/B and tests) switches them off as well
static bool gru_cluster_allowed() {
  if (g_gru_persistent < 0) { const char* e = getenv("ASR_GRU_PERSISTENT"); g_gru_persistent = (e && e[0] == '0') ? 0 : 1; }
  return g_gru_persistent == 1;
}

extern "C" int asr_gru_fwd(asr_handle* h, int T, int B, int H, int ndir, const float* xg, const float* xc,
                           const float* wgh, const float* wch, const int32_t* seq_len, int tmax, float* r, float* u,
                           float* c, float* rh, float* hout, float* hstate2, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  GRU_NEED(xg && xc && wgh && wch && seq_len && r && u && c && rh && hout && hstate2 && T >= 0 && B > 0 && B % 16 == 0 &&
               H > 0 && H % 16 == 0 && (ndir == 1 || ndir == 2) && tmax >= 0 && tmax <= T,
           "asr_gru_fwd: bad args (B=%d and H=%d must be multiples of 16, ndir=%d)", B, H, ndir);
  const size_t lds = (size_t)16 * (H + 1) * sizeof(float);
  if (lds > 160 * 1024) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_gru_fwd: num_units %d too large", H);
  hipStream_t st = (hipStream_t)s;
  (void)hipFuncSetAttribute((const void*)gru_gates_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)gru_cand_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const size_t sn = (size_t)ndir * B * H;
  float* hs[2] = {hstate2, hstate2 + sn};
  // H = 128 / 256: clusters of H / 32 CUs with the recurrent blocks in registers (lstm_cluster.hip), final state in hs[0]
  if (gru_cluster_allowed() &&
      asr_cluster_gru_fwd_try(h, T, B, H, ndir, xg, xc, wgh, wch, seq_len, r, u, c, rh, hout, hs[0], st)) {
    ASR_CHECK_LAUNCH(h, "asr_gru_fwd(cluster)");
    return ASR_OK;
  }
  if (gru_persistent_ok(h, H, ndir, (size_t)3 * 16 * (H + 4) * sizeof(float))) {
    // frames [tmax, T) of the output are beyond every utterance: zero
    if (T > tmax && hipMemsetAsync(hout + (size_t)tmax * B * ndir * H, 0, (size_t)(T - tmax) * B * ndir * H * sizeof(float), st) != hipSuccess)
      ASR_FAIL(h, ASR_ERR_HIP, "asr_gru_fwd: memset");
    float* wg_pk = (float*)h->scratch;                     // [ndir][2H/16][H/16][64][4]
    float* wc_pk = wg_pk + (size_t)ndir * 2 * H * H;       // [ndir][H/16][H/16][64][4]
    hipLaunchKernelGGL(gru_pack_kernel, dim3(256), dim3(256), 0, st, wgh, H, 2 * H, 2 * H, wg_pk, ndir, (size_t)2 * H * H,
                       (size_t)2 * H * H);
    hipLaunchKernelGGL(gru_pack_kernel, dim3(256), dim3(256), 0, st, wch, H, H, H, wc_pk, ndir, (size_t)H * H, (size_t)H * H);
    const size_t lds_p = (size_t)3 * 16 * (H + 4) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)gru_fwd_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
    hipLaunchKernelGGL(gru_fwd_persistent_kernel, dim3(B / 16, ndir), dim3(GP_THREADS), lds_p, st, T, B, H, ndir, tmax, xg, xc,
                       wg_pk, wc_pk, seq_len, r, u, c, rh, hout, hs[0]);
    ASR_CHECK_LAUNCH(h, "asr_gru_fwd");
    return ASR_OK;
  }
  if (hipMemsetAsync(hs[0], 0, sn * sizeof(float), st) != hipSuccess) ASR_FAIL(h, ASR_ERR_HIP, "asr_gru_fwd: memset");
  // frames [tmax, T) of the output are beyond every utterance: zero
  if (T > tmax && hipMemsetAsync(hout + (size_t)tmax * B * ndir * H, 0, (size_t)(T - tmax) * B * ndir * H * sizeof(float), st) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_gru_fwd: memset");
  for (int step = 0; step < tmax; ++step) {
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(2 * H / 16, B / 16, ndir), dim3(256), lds, st, step, T, B, H, ndir, xg,
                       hs[step & 1], wgh, seq_len, r, u, rh);
    hipLaunchKernelGGL(gru_cand_fwd_kernel, dim3(H / 16, B / 16, ndir), dim3(256), lds, st, step, T, B, H, ndir, xc, rh,
                       u, wch, hs[step & 1], seq_len, hs[(step + 1) & 1], c, hout);
  }
  ASR_CHECK_LAUNCH(h, "asr_gru_fwd");
  if ((tmax & 1) && tmax > 0 &&
      hipMemcpyAsync(hs[0], hs[1], sn * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_gru_fwd: copy of the final state");
  return ASR_OK;                                           // final state: hstate2[0 .. ndir*B*H)
}

extern "C" int asr_gru_bwd(asr_handle* h, int T, int B, int H, int ndir, const float* dout, const float* d_h_final,
                           const float* hout, const float* r, const float* u, const float* c, const float* wghT,
                           const float* wchT, const int32_t* seq_len, int tmax, float* dgate, float* dcand,
                           float* work2, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  GRU_NEED(dout && hout && r && u && c && wghT && wchT && seq_len && dgate && dcand && work2 && T >= 0 && B > 0 &&
               B % 16 == 0 && H > 0 && H % 16 == 0 && (ndir == 1 || ndir == 2) && tmax >= 0 && tmax <= T,
           "asr_gru_bwd: bad args (B=%d and H=%d must be multiples of 16, ndir=%d)", B, H, ndir);
  const size_t lds = (size_t)16 * (2 * H + 1) * sizeof(float);
  if (lds > 160 * 1024) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_gru_bwd: num_units %d too large", H);
  hipStream_t st = (hipStream_t)s;
  (void)hipFuncSetAttribute((const void*)gru_bwd_reset_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)gru_bwd_state_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const size_t sn = (size_t)ndir * B * H;
  float* dh_rec = work2;
  float* dh_acc = work2 + sn;
  // pre-activation gradients of frames no row reaches stay zero (the weight-gradient GEMMs run over all T*B rows)
  if (hipMemsetAsync(dgate, 0, (size_t)T * B * ndir * 2 * H * sizeof(float), st) != hipSuccess ||
      hipMemsetAsync(dcand, 0, (size_t)T * B * ndir * H * sizeof(float), st) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_gru_bwd: memset");
  // H = 128 / 256: clusters of H / 32 CUs (lstm_cluster.hip)
  if (gru_cluster_allowed() &&
      asr_cluster_gru_bwd_try(h, T, B, H, ndir, dout, d_h_final, hout, r, u, c, wghT, wchT, seq_len, dgate, dcand, st)) {
    ASR_CHECK_LAUNCH(h, "asr_gru_bwd(cluster)");
    return ASR_OK;
  }
  if (d_h_final) {
    if (hipMemcpyAsync(dh_rec, d_h_final, sn * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
      ASR_FAIL(h, ASR_ERR_HIP, "asr_gru_bwd: copy");
  } else if (hipMemsetAsync(dh_rec, 0, sn * sizeof(float), st) != hipSuccess) {
    ASR_FAIL(h, ASR_ERR_HIP, "asr_gru_bwd: memset");
  }
  const size_t lds_p = (size_t)16 * (3 * (H + 4) + 2 * H + 4) * sizeof(float);
  if (gru_persistent_ok(h, H, ndir, lds_p)) {
    float* wgT_pk = (float*)h->scratch;                    // [ndir][H/16][2H/16][64][4]  (K = 2H, N = H)
    float* wcT_pk = wgT_pk + (size_t)ndir * 2 * H * H;     // [ndir][H/16][H/16][64][4]
    hipLaunchKernelGGL(gru_pack_kernel, dim3(256), dim3(256), 0, st, wghT, 2 * H, H, H, wgT_pk, ndir, (size_t)2 * H * H,
                       (size_t)2 * H * H);
    hipLaunchKernelGGL(gru_pack_kernel, dim3(256), dim3(256), 0, st, wchT, H, H, H, wcT_pk, ndir, (size_t)H * H, (size_t)H * H);
    (void)hipFuncSetAttribute((const void*)gru_bwd_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
    hipLaunchKernelGGL(gru_bwd_persistent_kernel, dim3(B / 16, ndir), dim3(GP_THREADS), lds_p, st, T, B, H, ndir, tmax, dout,
                       d_h_final, hout, r, u, c, wgT_pk, wcT_pk, seq_len, dgate, dcand);
    ASR_CHECK_LAUNCH(h, "asr_gru_bwd");
    return ASR_OK;
  }
  const int eb = (int)((sn + 255) / 256 < 1024 ? (sn + 255) / 256 : 1024);
  const size_t lds1 = (size_t)16 * (H + 1) * sizeof(float);
  for (int step = tmax - 1; step >= 0; --step) {
    hipLaunchKernelGGL(gru_bwd_elem_kernel, dim3(eb), dim3(256), 0, st, step, T, B, H, ndir, dout, dh_rec, hout, u, c,
                       seq_len, dgate, dcand, dh_acc);
    hipLaunchKernelGGL(gru_bwd_reset_kernel, dim3(H / 16, B / 16, ndir), dim3(256), lds1, st, step, T, B, H, ndir, dcand,
                       wchT, hout, r, seq_len, dgate, dh_acc);
    hipLaunchKernelGGL(gru_bwd_state_kernel, dim3(H / 16, B / 16, ndir), dim3(256), lds, st, step, T, B, H, ndir, dgate,
                       wghT, dh_acc, seq_len, dh_rec);
  }
  ASR_CHECK_LAUNCH(h, "asr_gru_bwd");
  return ASR_OK;
}
