// Attention decoder kernels for gfx950 (rows a11-a15 of SURVEY.md section 8).
//
// Stand in for the per-step TF ops of the reference's decoder loop:
//   * LSTMBlockCell step of the decoder RNN           models/attention/attention_seq2seq.py:352-371
//   * AttentionLayer: energies, mask, sharpening, softmax, context
//                                                      models/attention/decoders/attention_layer.py:45-347
//   * tanh of the attentional vector, embedding lookup, masked sequence cross-entropy
//                                                      attention_decoder.py:189-209, attention_seq2seq.py:433-447,625-637
// Encoder outputs stay TIME-MAJOR [T,B,E] (what the encoder kernels write); keys [T,B,A] likewise.
// Everything here is HBM/latency-bound: per decoder step the energy pass reads keys once
// (B*T*A*4 B) and the context pass reads the encoder outputs once (B*T*E*4 B).  The GEMM-shaped
// parts (keys, query, cell input projection, attentional vector, output layer and all their
// gradients) run on asr_gemm, batched over all decoder steps wherever the recurrence allows.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float sigf(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- decoder LSTM cell, one step.  pre[B,4U] gate-major (i, ci, f, o) = [x,h]W + b.
// live[b] = 0 -> the row is finished: state copied through (dynamic_decode impute_finished).
__global__ void cell_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ c_prev,
                                const float* __restrict__ h_prev, const float* __restrict__ peep,
                                const float* __restrict__ live, int B, int U, float fb, float clip,
                                float* __restrict__ gates, float* __restrict__ c_raw,
                                float* __restrict__ c_out, float* __restrict__ h_out,
                                float* __restrict__ h_raw, const float* __restrict__ out_mask,
                                float* __restrict__ cell_out, float* __restrict__ h_out2, int ld_h2,
                                float* __restrict__ cell_out2, int ld_c2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * U) return;
  const int b = idx / U, j = idx % U;
  const float* p = pre + (size_t)b * 4 * U;
  const float cp = c_prev[idx];
  const float wci = peep ? peep[j] : 0.f, wcf = peep ? peep[U + j] : 0.f, wco = peep ? peep[2 * U + j] : 0.f;
  const float i = sigf(p[j] + wci * cp);
  const float g = tanhf(p[U + j]);
  const float f = sigf(p[2 * U + j] + fb + wcf * cp);
  float cn = g * i + cp * f;
  if (clip > 0.f) cn = fminf(fmaxf(cn, -clip), clip);
  const float o = sigf(p[3 * U + j] + wco * cn);
  const float hn = tanhf(cn) * o;
  float* gp = gates + (size_t)b * 4 * U;
  gp[j] = i; gp[U + j] = g; gp[2 * U + j] = f; gp[3 * U + j] = o;
  c_raw[idx] = cn;
  h_raw[idx] = hn;
  const float lv = live[b];
  c_out[idx] = lv > 0.f ? cn : cp;
  const float ho = lv > 0.f ? hn : h_prev[idx];
  h_out[idx] = ho;
  // the same values where their consumers read them (columns of the next step's cell input and of the attentional
  // vector's input): no copy launches between the step's kernels
  if (h_out2) h_out2[(size_t)b * ld_h2 + j] = ho;
  const float co = out_mask ? hn * out_mask[idx] : hn;    // DropoutWrapper(output_keep_prob) on the cell output
  if (cell_out) cell_out[idx] = co;
  if (cell_out2) cell_out2[(size_t)b * ld_c2 + j] = co;
}

// dh_raw: gradient w.r.t. the cell output h_new of LIVE rows (already includes everything that
// consumed it); dc_next / dh_next: gradient w.r.t. the carried state (c_out, h_out).
__global__ void cell_bwd_kernel(const float* __restrict__ dh_out_use, const float* __restrict__ dc_next,
                                const float* __restrict__ dh_next, const float* __restrict__ gates,
                                const float* __restrict__ c_raw, const float* __restrict__ c_prev,
                                const float* __restrict__ peep, const float* __restrict__ live, int B, int U,
                                float* __restrict__ dpre, float* __restrict__ dc_prev,
                                float* __restrict__ dh_prev_carry, float* __restrict__ dpeep_rows,
                                const float* __restrict__ dh_next2, int ld2, const float* __restrict__ use_mask,
                                float clip) {
  // clip (> 0: the forward clamped the new cell state to [-clip, clip], tf.clip_by_value in LSTMCell): no gradient
  // reaches the gates or c_prev through a state that was clamped (c_raw holds the CLAMPED value: |c| >= clip marks it).
  // dh_next2 (may be NULL, row stride ld2): a second addend of the carried-h gradient -- inside the decoder loop the
  // h-columns of the NEXT step's cell-input gradient; use_mask (may be NULL): the DropoutWrapper mask of the cell
  // output, applied to dh_out_use here instead of by a launch of its own
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * U) return;
  const int b = idx / U, j = idx % U;
  const float lv = live[b];
  float* dp = dpre + (size_t)b * 4 * U;
  const float dhn = dh_next2 ? dh_next[idx] + dh_next2[(size_t)b * ld2 + j] : dh_next[idx];
  if (!(lv > 0.f)) {   // finished row: state passes through, no parameter gradient
    dp[j] = 0.f; dp[U + j] = 0.f; dp[2 * U + j] = 0.f; dp[3 * U + j] = 0.f;
    dc_prev[idx] = dc_next[idx];
    dh_prev_carry[idx] = dhn;
    if (dpeep_rows) { dpeep_rows[(size_t)b * 3 * U + j] = 0.f; dpeep_rows[(size_t)b * 3 * U + U + j] = 0.f; dpeep_rows[(size_t)b * 3 * U + 2 * U + j] = 0.f; }
    return;
  }
  const float* gp = gates + (size_t)b * 4 * U;
  const float i = gp[j], g = gp[U + j], f = gp[2 * U + j], o = gp[3 * U + j];
  const float wci = peep ? peep[j] : 0.f, wcf = peep ? peep[U + j] : 0.f, wco = peep ? peep[2 * U + j] : 0.f;
  const float c = c_raw[idx], cp = c_prev[idx];
  const float dh = (use_mask ? dh_out_use[idx] * use_mask[idx] : dh_out_use[idx]) + dhn;
  const float tc = tanhf(c);
  const float d_o = dh * tc * o * (1.f - o);
  const float dct = dc_next[idx] + dh * o * (1.f - tc * tc) + d_o * wco;
  const float dc = (clip > 0.f && fabsf(c) >= clip) ? 0.f : dct;
  const float d_g = dc * i * (1.f - g * g);
  const float d_i = dc * g * i * (1.f - i);
  const float d_f = dc * cp * f * (1.f - f);
  dp[j] = d_i; dp[U + j] = d_g; dp[2 * U + j] = d_f; dp[3 * U + j] = d_o;
  dc_prev[idx] = dc * f + d_i * wci + d_f * wcf;
  dh_prev_carry[idx] = 0.f;   // h_prev enters only through the input projection (dpre W^T)
  if (dpeep_rows) {
    dpeep_rows[(size_t)b * 3 * U + j] = d_i * cp;
    dpeep_rows[(size_t)b * 3 * U + U + j] = d_f * cp;
    dpeep_rows[(size_t)b * 3 * U + 2 * U + j] = d_o * c;
  }
}

// ---- attention scoring / context.  All of these stream [T,B,*] tensors once per decoder step, so
// they are HBM-bound and laid out for that: grid = (frame chunks, utterances) -> hundreds of
// workgroups instead of one per utterance, float4 rows, per-chunk partial sums reduced in a fixed
// order (deterministic) through the handle's scratch.
constexpr int ATT_CH = 64;                      // frames per workgroup

// energies.  keys[T,B,A] (may be null), qz[B,A].  mode 0: sum_a v_a tanh(useK*K + qz); mode 1: sum_a K*qz.
__global__ __launch_bounds__(256) void att_energy_fwd_kernel(const float* __restrict__ keys,
                                                             const float* __restrict__ qz,
                                                             const float* __restrict__ v, int T, int B,
                                                             int A, int mode, float* __restrict__ energy,
                                                             const int32_t* __restrict__ seq_len) {
  // seq_len (may be NULL): frames at or past an utterance's length are masked by the softmax that follows and are
  // neither computed nor written
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tend = seq_len ? min(max(seq_len[b], 0), T) : T;
  const int t0 = blockIdx.x * ATT_CH, t1 = min(tend, t0 + ATT_CH);
  const float* q = qz + (size_t)b * A;
  for (int t = t0 + wave; t < t1; t += 4) {
    const float* k = keys ? keys + ((size_t)t * B + b) * A : nullptr;
    float s = 0.f;
    for (int a = lane; a < A; a += 64) {
      if (mode == 0) s += v[a] * tanhf((k ? k[a] : 0.f) + q[a]);
      else s += k[a] * q[a];
    }
    s = wave_reduce_sum(s);
    if (lane == 0) energy[(size_t)b * T + t] = s;
  }
}

// dkeys[T,B,A] += dZ ; per-chunk partials part[ch][b][2][A] = (sum_t dZ, sum_t denergy*tanh(Z))
__global__ __launch_bounds__(256) void att_energy_bwd_kernel(const float* __restrict__ denergy,
                                                             const float* __restrict__ keys,
                                                             const float* __restrict__ qz,
                                                             const float* __restrict__ v, int T, int B,
                                                             int A, int mode, float* __restrict__ dkeys,
                                                             float* __restrict__ part,
                                                             const int32_t* __restrict__ seq_len) {
  // seq_len (may be NULL): denergy is zero at and past an utterance's length, those frames are skipped
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc = reinterpret_cast<float*>(smem);   // [4 waves][2][A]
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tend = seq_len ? min(max(seq_len[b], 0), T) : T;
  const int t0 = blockIdx.x * ATT_CH, t1 = min(tend, t0 + ATT_CH);
  const float* q = qz + (size_t)b * A;
  for (int a = lane; a < 2 * A; a += 64) acc[wave * 2 * A + a] = 0.f;
  for (int t = t0 + wave; t < t1; t += 4) {
    const float de = denergy[(size_t)b * T + t];
    const size_t off = ((size_t)t * B + b) * A;
    for (int a = lane; a < A; a += 64) {
      if (mode == 0) {
        const float th = tanhf((keys ? keys[off + a] : 0.f) + q[a]);
        const float dz = de * v[a] * (1.f - th * th);
        if (dkeys) dkeys[off + a] += dz;
        acc[wave * 2 * A + a] += dz;
        acc[wave * 2 * A + A + a] += de * th;
      } else {
        const float kv = keys[off + a];
        if (dkeys) dkeys[off + a] += de * q[a];
        acc[wave * 2 * A + a] += de * kv;
      }
    }
  }
  __syncthreads();
  float* o = part + ((size_t)blockIdx.x * B + b) * 2 * A;
  for (int a = threadIdx.x; a < 2 * A; a += 256)
    o[a] = (acc[a] + acc[2 * A + a]) + (acc[4 * A + a] + acc[6 * A + a]);
}
// dqz[b,a] / dv_rows[b,a] = fixed-order sum of the chunk partials.  One thread per output, the loads of 32 chunks issued
// before the first add (the kernel is nothing but the latency of those loads)
__global__ void att_energy_bwd_reduce_kernel(const float* __restrict__ part, int nch, int B, int A,
                                             float* __restrict__ dqz, float* __restrict__ dv_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 2 * A) return;
  const int b = i / (2 * A), j = i % (2 * A);
  if (j >= A && !dv_rows) return;
  const float* p = part + (size_t)b * 2 * A + j;
  const size_t cs = (size_t)B * 2 * A;
  float s = 0.f;
  for (int c0 = 0; c0 < nch; c0 += 32) {
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = (c0 + c < nch) ? p[(size_t)(c0 + c) * cs] : 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) s += x[c];
  }
  if (j < A) dqz[(size_t)b * A + j] = s;
  else dv_rows[(size_t)b * A + j - A] = s;
}

struct SoftmaxBwdFold {        // see att_energy_bwd_vec_kernel
  const float* da;             // [B,T] d loss / d alpha  (NULL: not folded)
  const float* alpha;          // [B,T]
  const float* dotp;           // [ndot,B] partial sums of alpha . dalpha
  const float* norm;           // [B] sigmoid-smoothing normaliser or NULL
  int ndot;
  float sharp;
};

// The same two kernels for A % 4 == 0, A <= 512 (every configuration of the reference): a frame is handled by LPF lanes
// holding float4 slices of qz / v in registers (LPF = 16, 32 or 64 -> 4, 2 or 1 frames per wave and trip), tanh from
// the hardware exp2 / rcp, the per-frame sum by DPP adds, and -- backward -- the sums over frames in registers instead
// of read-modify-writes on LDS.  The one-element-per-lane forms above were ALU-bound at ~15 us per launch.
template <int LPF, int NV>
__global__ __launch_bounds__(256) void att_energy_fwd_vec_kernel(const float* __restrict__ keys,
                                                                 const float* __restrict__ qz,
                                                                 const float* __restrict__ v, int T, int B, int A,
                                                                 int mode, float* __restrict__ energy,
                                                                 const int32_t* __restrict__ seq_len) {
  constexpr int FPW = 64 / LPF;
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPF, l = lane % LPF, nvec = A >> 2;
  const int tend = seq_len ? min(max(seq_len[b], 0), T) : T;
  const int t0 = blockIdx.x * ATT_CH, t1 = min(tend, t0 + ATT_CH);
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  f32x4_t qv[NV], vv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int a4 = l + i * LPF;
    qv[i] = a4 < nvec ? *reinterpret_cast<const f32x4_t*>(qz + (size_t)b * A + a4 * 4) : zero;
    vv[i] = (a4 < nvec && mode == 0) ? *reinterpret_cast<const f32x4_t*>(v + a4 * 4) : zero;
  }
  for (int tb = t0 + wave * FPW; tb < t1; tb += 4 * FPW) {
    const int t = tb + sub;
    const bool valid = t < t1;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int a4 = l + i * LPF;
      f32x4_t kv = zero;
      if (keys && valid && a4 < nvec) kv = *reinterpret_cast<const f32x4_t*>(keys + ((size_t)t * B + b) * A + a4 * 4);
      if (mode == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s += vv[i][e] * fast_tanhf(kv[e] + qv[i][e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) s += kv[e] * qv[i][e];
      }
    }
    s = group_reduce_sum<LPF>(s);
    if (l == 0 && valid) energy[(size_t)b * T + t] = s;
  }
}
template <int LPF, int NV>
__global__ __launch_bounds__(256) void att_energy_bwd_vec_kernel(const float* __restrict__ denergy,
                                                                 const float* __restrict__ keys,
                                                                 const float* __restrict__ qz,
                                                                 const float* __restrict__ v, int T, int B, int A,
                                                                 int mode, float* __restrict__ dkeys,
                                                                 float* __restrict__ part,
                                                                 const int32_t* __restrict__ seq_len, int ech,
                                                                 SoftmaxBwdFold fold) {
  // ech: frames per workgroup.  fold.da != NULL: denergy is not read but formed here from the d-alpha values, the
  // attention weights and the per-chunk partial sums of alpha . dalpha the d-alpha kernel left (the softmax backward
  // kernel's arithmetic, without its launch)
  constexpr int FPW = 64 / LPF, AP = NV * LPF * 4;
  __shared__ float red[4][2][AP];
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPF, l = lane % LPF, nvec = A >> 2;
  const int tend = seq_len ? min(max(seq_len[b], 0), T) : T;
  const int t0 = blockIdx.x * ech, t1 = min(tend, t0 + ech);
  float dot = 0.f, nrm = 0.f;
  if (fold.da) {
    for (int c = 0; c < fold.ndot; ++c) dot += fold.dotp[(size_t)c * B + b];    // fixed order
    if (fold.norm) nrm = fold.norm[b];
  }
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  f32x4_t qv[NV], vv[NV], adq[NV], adv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int a4 = l + i * LPF;
    qv[i] = a4 < nvec ? *reinterpret_cast<const f32x4_t*>(qz + (size_t)b * A + a4 * 4) : zero;
    vv[i] = (a4 < nvec && mode == 0) ? *reinterpret_cast<const f32x4_t*>(v + a4 * 4) : zero;
    adq[i] = zero;
    adv[i] = zero;
  }
  for (int tb = t0 + wave * FPW; tb < t1; tb += 4 * FPW) {
    const int t = tb + sub;
    const bool valid = t < t1;
    float de = 0.f;
    if (valid) {
      if (fold.da) {
        const float al = fold.alpha[(size_t)b * T + t];
        de = fold.sharp * al * (fold.da[(size_t)b * T + t] - dot);
        if (fold.norm) de *= 1.f - al * nrm;             // sigmoid smoothing: s (1 - s) / sum, s = alpha * sum
      } else {
        de = denergy[(size_t)b * T + t];
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int a4 = l + i * LPF;
      if (!(valid && a4 < nvec)) continue;
      const size_t off = ((size_t)t * B + b) * A + a4 * 4;
      const f32x4_t kv = keys ? *reinterpret_cast<const f32x4_t*>(keys + off) : zero;
      f32x4_t dz;
      if (mode == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float th = fast_tanhf(kv[e] + qv[i][e]);
          dz[e] = de * vv[i][e] * (1.f - th * th);
          adq[i][e] += dz[e];
          adv[i][e] += de * th;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dz[e] = de * qv[i][e];
          adq[i][e] += de * kv[e];
        }
      }
      if (dkeys) {
        f32x4_t* dk = reinterpret_cast<f32x4_t*>(dkeys + off);
        f32x4_t o = *dk;
        o[0] += dz[0]; o[1] += dz[1]; o[2] += dz[2]; o[3] += dz[3];
        *dk = o;
      }
    }
  }
  // the FPW frame slots of a wave hold the same columns: fold them, then the four waves through LDS (fixed order)
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = LPF; o < 64; o <<= 1) {
        adq[i][e] += __shfl_xor(adq[i][e], o, 64);
        adv[i][e] += __shfl_xor(adv[i][e], o, 64);
      }
    }
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      *reinterpret_cast<f32x4_t*>(&red[wave][0][(l + i * LPF) * 4]) = adq[i];
      *reinterpret_cast<f32x4_t*>(&red[wave][1][(l + i * LPF) * 4]) = adv[i];
    }
  }
  __syncthreads();
  float* o = part + ((size_t)blockIdx.x * B + b) * 2 * A;
  for (int a = threadIdx.x; a < 2 * A; a += 256) {
    const int w = a >= A, c = a - w * A;
    o[a] = (red[0][w][c] + red[1][w][c]) + (red[2][w][c] + red[3][w][c]);
  }
}
// 0: not applicable, else LPF * 4 + NV  (A % 4 == 0, A <= 512, 16-byte aligned operands)
static inline int energy_vec_shape(int A, const void* keys, const void* qz, const void* v, const void* dkeys) {
  if (A % 4 != 0 || A > 512 || ((uintptr_t)keys | (uintptr_t)qz | (uintptr_t)v | (uintptr_t)dkeys) % 16 != 0) return 0;
  const int nvec = A / 4;
  return nvec <= 16 ? 16 * 4 + 1 : nvec <= 32 ? 32 * 4 + 1 : nvec <= 64 ? 64 * 4 + 1 : 64 * 4 + 2;
}

// masked softmax over t.  energy[B,T] -> alpha[B,T].
// mask: e*m + (1-m)*FLT_MIN(lowest), then *sharpening (attention_layer.py:76-89).
// norm != NULL selects the reference's sigmoid smoothing (attention_layer.py:92-96): alpha = sigmoid(e) / sum_t
// sigmoid(e); norm[b] receives the sum, which the backward needs (d sigmoid = s (1 - s), s = alpha * sum).
__global__ __launch_bounds__(256) void att_softmax_kernel(const float* __restrict__ energy,
                                                          const int32_t* __restrict__ seq_len, float sharp,
                                                          int T, float* __restrict__ alpha,
                                                          float* __restrict__ norm) {
  __shared__ float red[4];
  __shared__ float s_max, s_inv;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* al = reinterpret_cast<float*>(smem);   // [T]
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int len = min(max(seq_len[b], 0), T);
  const float lowest = -3.402823466e+38f;
  if (norm) {
    float sg = 0.f;
    for (int t = tid; t < T; t += 256) {
      const float e = fmaxf((t < len ? energy[(size_t)b * T + t] : lowest) * sharp, lowest);
      const float p = 1.f / (1.f + expf(-e));          // masked frames: exp(+3.4e38) = inf -> exactly 0
      al[t] = p;
      sg += p;
    }
    sg = wave_reduce_sum(sg);
    if (lane == 0) red[wave] = sg;
    __syncthreads();
    if (tid == 0) {
      const float tot = red[0] + red[1] + red[2] + red[3];
      norm[b] = tot;
      s_inv = tot > 0.f ? 1.f / tot : 0.f;
    }
    __syncthreads();
    const float inv = s_inv;
    // an all-masked row (batch padding) is 0/0 in the reference; uniform weights here, like the softmax branch
    for (int t = tid; t < T; t += 256) alpha[(size_t)b * T + t] = inv > 0.f ? al[t] * inv : 1.f / (float)T;
    return;
  }
  float m = -INFINITY;
  for (int t = tid; t < T; t += 256) {
    // float32.min * sharpening overflows to -inf for sharpening > 1 (reference quirk Q13); clamp so a
    // fully masked row (batch padding, len = 0) still yields finite, uniform weights
    const float e = fmaxf((t < len ? energy[(size_t)b * T + t] : lowest) * sharp, lowest);
    al[t] = e;
    m = fmaxf(m, e);
  }
  m = wave_reduce_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) s_max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float mx = s_max;
  float s = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float p = expf(al[t] - mx);
    al[t] = p;
    s += p;
  }
  s = wave_reduce_sum(s);
  __syncthreads();
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) s_inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  const float inv = s_inv;
  for (int t = tid; t < T; t += 256) alpha[(size_t)b * T + t] = al[t] * inv;
}

// encoder rows are read either as fp32 or from the bf16 operand copy the encoder already keeps (half the bytes
// of the two per-step streams over [T,B,2H])
// (Round 5, measured: the non-temporal hint on these streams -- to keep the decoder cell's 6.5 MB weight image in the L2s
// between two steps -- makes them SLOWER, att_fused_fwd 15.9 -> 19.8 us, att_dalpha_vec 15.1 -> 19.0, and the cell product
// no faster (11.4 us): the 69 MB of a step live in the 256 MB memory-side cache from one step to the next, which nt bypasses.)
__device__ __forceinline__ f32x4_t enc_ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ f32x4_t enc_ld4(const bf16_t* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
  const us4_t v = *reinterpret_cast<const us4_t*>(p);
  return (f32x4_t){bf16_to_f32(v[0]), bf16_to_f32(v[1]), bf16_to_f32(v[2]), bf16_to_f32(v[3])};
}
__device__ __forceinline__ float enc_ld1(const float* p) { return *p; }
__device__ __forceinline__ float enc_ld1(const bf16_t* p) { return bf16_to_f32(*p); }

// partial context of one frame chunk: part[ch][b][E] = sum_{t in chunk, t < len*} alpha[b,t] enc[t,b,:]
// (len* = len, or T for an all-masked row whose weights are uniform over T zero frames)
template <typename TE>
__global__ __launch_bounds__(256) void att_ctx_partial_kernel(const float* __restrict__ alpha,
                                                              const int32_t* __restrict__ seq_len,
                                                              const TE* __restrict__ enc, int T, int B,
                                                              int E, float* __restrict__ part) {
  __shared__ float al[ATT_CH];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int t0 = blockIdx.x * ATT_CH;
  int len = min(max(seq_len[b], 0), T);
  if (len == 0) len = T;
  const int n = max(0, min(ATT_CH, len - t0));
  if (tid < n) al[tid] = alpha[(size_t)b * T + t0 + tid];
  __syncthreads();
  float* o = part + ((size_t)blockIdx.x * B + b) * E;
  const size_t rs = (size_t)B * E;
  if ((E & 3) == 0) {
    for (int e4 = tid; e4 < E / 4; e4 += 256) {
      const TE* p = enc + ((size_t)t0 * B + b) * E + e4 * 4;
      f32x4_t c = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < n; ++i) {
        const f32x4_t x = enc_ld4(p + i * rs);
        const float w = al[i];
        c[0] += w * x[0]; c[1] += w * x[1]; c[2] += w * x[2]; c[3] += w * x[3];
      }
      *reinterpret_cast<f32x4_t*>(o + e4 * 4) = c;
    }
  } else {
    for (int e0 = tid; e0 < E; e0 += 256) {
      float c = 0.f;
      for (int i = 0; i < n; ++i) c += al[i] * enc_ld1(enc + ((size_t)(t0 + i) * B + b) * E + e0);
      o[e0] = c;
    }
  }
}
__global__ void att_ctx_reduce_kernel(const float* __restrict__ part, int nch, int BE, float* __restrict__ ctx, int E,
                                      float* __restrict__ ctx2, int ld2, float* __restrict__ ctx3, int ld3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BE) return;
  float c = 0.f;
#pragma unroll 8
  for (int k = 0; k < nch; ++k) c += part[(size_t)k * BE + i];   // fixed order
  ctx[i] = c;
  if (ctx2) ctx2[(size_t)(i / E) * ld2 + i % E] = c;
  if (ctx3) ctx3[(size_t)(i / E) * ld3 + i % E] = c;
}

// 16-byte vectors of an encoder row in either storage type
template <typename TE> struct EncVec;
template <> struct EncVec<float> {
  static constexpr int N = 4;
  typedef f32x4_t raw_t;
  static __device__ __forceinline__ void unpack(const raw_t& r, float* o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = r[i];
  }
};
template <> struct EncVec<bf16_t> {
  static constexpr int N = 8;
  typedef bf16x8_t raw_t;
  static __device__ __forceinline__ void unpack(const raw_t& r, float* o) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32((bf16_t)r[i]);
  }
};
// ---------------------------------------------------------------- energies + softmax + context in ONE pass (decoder loop)
// A decoder step used to be energies -> softmax -> partial context -> reduction: four dependent launches of 4-12 us each
// with ~4 us of queue latency between them (profiles/r03_cfgD_kernel_trace.md).  The softmax does not need a launch of
// its own: a (chunk, utterance) workgroup computes its 64 energies, their maximum m_c, p_t = exp(e_t - m_c), s_c = sum p_t
// and the UNNORMALISED partial context sum_t p_t enc_t in one go (the chunk's weights never leave LDS), and the
// combine kernel that sums the chunk partials anyway rescales them: with M = max_c m_c and S = sum_c s_c exp(m_c - M),
// ctx = sum_c part_c exp(m_c - M) / S and alpha_t = p_t exp(m_c(t) - M) / S (written for the backward pass).  Same
// masking as att_softmax_kernel: frames past the length get weight 0, an all-masked row (batch padding) uniform weights.
// Measured (cfg D / cfg E shards, ms per step): 83.84 / 48.91 against 83.85 / 49.57 for the four-launch form -- two launches
// and their queue gaps fewer, but the same dependent round trips inside one kernel: device time is unchanged (the
// round-2 finding again), host issue drops by 800 launches per step.
// (Round 4, measured: dropping the combine launch as well -- every workgroup publishes its partials with write-through
// stores and takes a ticket, the last arrival of an utterance combines after reading them back past its L2 -- is bit-identical
// and NOT faster: cfg D 77.7 ms with it against 77.4 ms with the combine launch, cfg E 44.95 / 44.92.  A 5 us launch and
// its gap buy the same as a device-scope ticket plus ~30 dependent sc1 round trips in one workgroup; left out.)
// FCH = frames per workgroup: 32 for T <= 2048 (twice the workgroups, half the serial frames of the context phase each)
template <int LPF, int NV, typename TE, int FCH>
__global__ __launch_bounds__(256) void att_fused_fwd_kernel(const float* __restrict__ keys, const float* __restrict__ qz,
                                                            const float* __restrict__ v, int T, int B, int A, int mode,
                                                            float sharp, const int32_t* __restrict__ seq_len,
                                                            const TE* __restrict__ enc, int E, float* __restrict__ alpha,
                                                            float* __restrict__ part, float* __restrict__ stat) {
  constexpr int FPW = 64 / LPF;
  __shared__ float el[FCH];
  __shared__ float s_m;
  const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sub = lane / LPF, l = lane % LPF, nvec = A >> 2;
  const int len = min(max(seq_len[b], 0), T);
  const int t0 = blockIdx.x * FCH;
  const float lowest = -3.402823466e+38f;
  const int n = max(0, min(FCH, (len == 0 ? T : len) - t0));      // frames of this chunk that carry weight
  float* st = stat + ((size_t)blockIdx.x * B + b) * 2;
  float* o = part + ((size_t)blockIdx.x * B + b) * E;
  if (n == 0) {                                                       // block-uniform
    if (tid == 0) { st[0] = lowest; st[1] = 0.f; }
    return;                                                           // (the combine kernel skips the partial)
  }
  if (len == 0) {
    if (tid < n) el[tid] = lowest;
  } else {
    const int t1 = t0 + n;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t qv[NV], vv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int a4 = l + i * LPF;
      qv[i] = a4 < nvec ? *reinterpret_cast<const f32x4_t*>(qz + (size_t)b * A + a4 * 4) : zero;
      vv[i] = (a4 < nvec && mode == 0) ? *reinterpret_cast<const f32x4_t*>(v + a4 * 4) : zero;
    }
    for (int tb = t0 + wave * FPW; tb < t1; tb += 4 * FPW) {
      const int t = tb + sub;
      const bool valid = t < t1;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int a4 = l + i * LPF;
        f32x4_t kv = zero;
        if (keys && valid && a4 < nvec) kv = *reinterpret_cast<const f32x4_t*>(keys + ((size_t)t * B + b) * A + a4 * 4);
        if (mode == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) sum += vv[i][e] * fast_tanhf(kv[e] + qv[i][e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) sum += kv[e] * qv[i][e];
        }
      }
      sum = group_reduce_sum<LPF>(sum);
      if (l == 0 && valid) el[t - t0] = fmaxf(sum * sharp, lowest);
    }
  }
  __syncthreads();
  if (wave == 0) {
    const float e = lane < n ? el[lane] : -INFINITY;
    const float m = wave_reduce_max(e);
    const float pe = lane < n ? expf(e - m) : 0.f;
    const float sm = wave_reduce_sum(pe);
    if (lane < n) {
      el[lane] = pe;
      alpha[(size_t)b * T + t0 + lane] = pe;                          // unnormalised; the combine kernel rescales it
    }
    if (lane == 0) { st[0] = m; st[1] = sm; s_m = m; }
  }
  __syncthreads();
  const size_t rs = (size_t)B * E;
  // (Measured, round 4: a form of this phase with one 16-byte vector per thread, the 256 threads split into frame groups
  // and 16 rows requested per trip -- 32 KB in flight per wave instead of a few rows -- takes 16.7 us per launch at
  // cfg D against 15.7 for this loop: the launch is not bound by the loads in flight per thread.)
  for (int e4 = tid; e4 < E / 4; e4 += 256) {
    const TE* pe = enc + ((size_t)t0 * B + b) * E + e4 * 4;
    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) {
      const f32x4_t x = enc_ld4(pe + i * rs);
      const float w = el[i];
      c[0] += w * x[0]; c[1] += w * x[1]; c[2] += w * x[2]; c[3] += w * x[3];
    }
    *reinterpret_cast<f32x4_t*>(o + e4 * 4) = c;
  }
}
// combine: a workgroup belongs to ONE utterance -- its first wave turns the <= 64 chunk statistics into the weights
// w_c = exp(m_c - M) / S once (LDS), then blocks [0, B * E/256) sum 256 context elements each over the chunk partials and the
// rest rescale 256 attention weights each
__global__ __launch_bounds__(256) void att_fused_combine_kernel(const float* __restrict__ part, const float* __restrict__ stat,
                                                                int nch, int fch, int B, int E, int T,
                                                                const int32_t* __restrict__ seq_len,
                                                                float* __restrict__ ctx, float* __restrict__ ctx2, int ld2,
                                                                float* __restrict__ ctx3, int ld3, float* __restrict__ alpha) {
  __shared__ float w[64];
  const int epb = E / 256, nb_ctx = B * epb, tpb = (T + 255) / 256;
  const bool is_ctx = (int)blockIdx.x < nb_ctx;
  const int b = is_ctx ? blockIdx.x / epb : (blockIdx.x - nb_ctx) / tpb;
  const int tid = threadIdx.x;
  if (tid < 64) {
    const float m = tid < nch ? stat[((size_t)tid * B + b) * 2] : -3.402823466e+38f;
    const float sk = tid < nch ? stat[((size_t)tid * B + b) * 2 + 1] : 0.f;
    const float M = wave_reduce_max(m);
    const float wk = sk > 0.f ? expf(m - M) : 0.f;
    const float S = wave_reduce_sum(sk * wk);
    w[tid] = wk / S;
  }
  __syncthreads();
  if (is_ctx) {
    const int e = (blockIdx.x % epb) * 256 + tid;
    const size_t BE = (size_t)B * E, i = (size_t)b * E + e;
    float c = 0.f;
    // (every partial is REQUESTED -- a load behind `if (w > 0)` cannot leave before the one in front of it has been
    // tested, and 25 dependent round trips were this kernel's 10 us -- but the partials of chunks past the length were
    // never written and are not used: a select, not a multiplication by zero)
#pragma unroll 8
    for (int k = 0; k < nch; ++k) {                                    // fixed order
      const float wk = w[k];
      const float pv = part[(size_t)k * BE + i];
      c += wk > 0.f ? pv * wk : 0.f;
    }
    ctx[i] = c;
    if (ctx2) ctx2[(size_t)b * ld2 + e] = c;
    if (ctx3) ctx3[(size_t)b * ld3 + e] = c;
  } else {
    const int t = ((blockIdx.x - nb_ctx) % tpb) * 256 + tid;
    if (t >= T) return;
    int len = min(max(seq_len[b], 0), T);
    if (len == 0) len = T;
    const size_t i = (size_t)b * T + t;
    alpha[i] = t < len ? alpha[i] * w[t / fch] : 0.f;
  }
}

// dalpha[b,t] = enc[t,b,:] . dctx[b,:]  for t < len (one wave per frame, float4 lanes)
template <typename TE>
__global__ __launch_bounds__(256) void att_dalpha_kernel(const float* __restrict__ dctx,
                                                         const int32_t* __restrict__ seq_len,
                                                         const TE* __restrict__ enc, int T, int B, int E,
                                                         float* __restrict__ da) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int len = min(max(seq_len[b], 0), T);
  const int t0 = blockIdx.x * ATT_CH, t1 = min(len, t0 + ATT_CH);
  const float* dc = dctx + (size_t)b * E;
  for (int t = t0 + wave; t < t1; t += 4) {
    const TE* er = enc + ((size_t)t * B + b) * E;
    float s = 0.f;
    if ((E & 3) == 0) {
      for (int e4 = lane; e4 < E / 4; e4 += 64) {
        const f32x4_t x = enc_ld4(er + e4 * 4);
        const f32x4_t d = *reinterpret_cast<const f32x4_t*>(dc + e4 * 4);
        s += x[0] * d[0] + x[1] * d[1] + x[2] * d[2] + x[3] * d[3];
      }
    } else {
      for (int e0 = lane; e0 < E; e0 += 64) s += enc_ld1(er + e0) * dc[e0];
    }
    s = wave_reduce_sum(s);
    if (lane == 0) da[(size_t)b * T + t] = s;
  }
}
// The same product for the rows the decoder loop streams every step (E a multiple of 64 sixteen-byte vectors): the
// lane keeps its slice of dctx in registers, a wave takes FOUR frames per trip and issues their 4 G sixteen-byte loads
// before the first multiply (the one-frame form above waits for each row in turn and ran at a third of the stream
// rate of att_ctx_partial_kernel over the same bytes).  dctx = dctx_a (+ dctx_b, row stride ldb): inside the loop the
// attentional-vector part plus the context columns of the next step's cell-input gradient; the sum is also written
// to dctx_out (by the first workgroup of each utterance) for the d_enc contraction after the loop.
// FPT = frames per trip of a wave (their FPT * G sixteen-byte loads are all requested before the first multiply).  Four:
// sixteen per trip (one trip per 64-frame chunk and wave) measured 16.6 us per launch at cfg D against 15.1 (round 4).
template <typename TE, int G, int FPT = 4>
__global__ __launch_bounds__(256) void att_dalpha_vec_kernel(const float* __restrict__ dctx_a,
                                                             const float* __restrict__ dctx_b, int ldb,
                                                             float* __restrict__ dctx_out,
                                                             const int32_t* __restrict__ seq_len,
                                                             const TE* __restrict__ enc, int T, int B, int E,
                                                             float* __restrict__ da,
                                                             const float* __restrict__ alpha,
                                                             float* __restrict__ dotp) {
  // dotp (may be NULL): dotp[chunk][b] = sum over the chunk's frames of alpha * dalpha -- what the softmax backward
  // needs from all of T, handed to the energy backward kernel so that no kernel of its own has to form it
  typedef EncVec<TE> V;
  constexpr int N = V::N;
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int len = min(max(seq_len[b], 0), T);
  const int t0 = blockIdx.x * ATT_CH, t1 = min(len, t0 + ATT_CH);
  float d[G][N];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int e0 = (g * 64 + lane) * N;
#pragma unroll
    for (int i = 0; i < N; i += 4) {
      f32x4_t x = *reinterpret_cast<const f32x4_t*>(dctx_a + (size_t)b * E + e0 + i);
      if (dctx_b) {
        const f32x4_t y = *reinterpret_cast<const f32x4_t*>(dctx_b + (size_t)b * ldb + e0 + i);
        x[0] += y[0]; x[1] += y[1]; x[2] += y[2]; x[3] += y[3];
      }
      if (dctx_out && blockIdx.x == 0 && wave == 0) *reinterpret_cast<f32x4_t*>(dctx_out + (size_t)b * E + e0 + i) = x;
      d[g][i] = x[0]; d[g][i + 1] = x[1]; d[g][i + 2] = x[2]; d[g][i + 3] = x[3];
    }
  }
  float wdot = 0.f;
  for (int tb = t0 + wave * FPT; tb < t1; tb += 4 * FPT) {
    typename V::raw_t x[FPT][G];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const int t = min(tb + f, t1 - 1);
#pragma unroll
      for (int g = 0; g < G; ++g)
        x[f][g] = *reinterpret_cast<const typename V::raw_t*>(enc + ((size_t)t * B + b) * E + (g * 64 + lane) * N);
    }
    float sum[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      float acc = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float xv[N];
        V::unpack(x[f][g], xv);
#pragma unroll
        for (int i = 0; i < N; ++i) acc += xv[i] * d[g][i];
      }
      sum[f] = wave_reduce_sum(acc);
    }
    if (lane == 0) {
#pragma unroll
      for (int f = 0; f < FPT; ++f)
        if (tb + f < t1) {
          da[(size_t)b * T + tb + f] = sum[f];
          if (dotp) wdot += alpha[(size_t)b * T + tb + f] * sum[f];
        }
    }
  }
  if (dotp) {
    __shared__ float wd[4];
    if (lane == 0) wd[wave] = wdot;
    __syncthreads();
    if (threadIdx.x == 0) dotp[(size_t)blockIdx.x * B + b] = (wd[0] + wd[1]) + (wd[2] + wd[3]);
  }
}
// denergy[b,t] = sharp * alpha * (dalpha - sum_t alpha dalpha)   (zero past len)
// dalpha_extra (may be NULL): a further gradient w.r.t. alpha -- the location features of the NEXT decoder step
// (carried-alpha location / hybrid attention)
__global__ __launch_bounds__(256) void att_softmax_bwd_kernel(float* __restrict__ da,
                                                              const float* __restrict__ alpha,
                                                              const int32_t* __restrict__ seq_len, float sharp,
                                                              int T, float* __restrict__ denergy,
                                                              const float* __restrict__ norm,
                                                              const float* __restrict__ dalpha_extra) {
  __shared__ float red[4];
  __shared__ float s_dot;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int len = min(max(seq_len[b], 0), T);
  float dot = 0.f;
  if (dalpha_extra)      // each thread revisits exactly the elements it updates here
    for (int t = tid; t < len; t += 256) da[(size_t)b * T + t] += dalpha_extra[(size_t)b * T + t];
  for (int t = tid; t < len; t += 256) dot += alpha[(size_t)b * T + t] * da[(size_t)b * T + t];
  dot = wave_reduce_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  if (tid == 0) s_dot = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  const float dsum = s_dot;
  for (int t = tid; t < T; t += 256) {
    float de = 0.f;
    if (t < len) {
      const float a = alpha[(size_t)b * T + t];
      de = sharp * a * (da[(size_t)b * T + t] - dsum);
      if (norm) de *= 1.f - a * norm[b];               // sigmoid smoothing: s (1 - s) / sum, s = a * sum
    }
    denergy[(size_t)b * T + t] = de;
  }
}
// denc[t,b,:] += alpha[b,t] * dctx[b,:]  (per-step form; the model defers this to one GEMM per utterance)
__global__ __launch_bounds__(256) void att_denc_kernel(const float* __restrict__ dctx,
                                                       const float* __restrict__ alpha,
                                                       const int32_t* __restrict__ seq_len, int T, int B, int E,
                                                       float* __restrict__ denc) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int len = min(max(seq_len[b], 0), T);
  const int t0 = blockIdx.x * ATT_CH, t1 = min(len, t0 + ATT_CH);
  const float* dc = dctx + (size_t)b * E;
  for (int t = t0 + wave; t < t1; t += 4) {
    const float a = alpha[(size_t)b * T + t];
    float* dr = denc + ((size_t)t * B + b) * E;
    for (int e0 = lane; e0 < E; e0 += 64) dr[e0] += a * dc[e0];
  }
}

// ---- location / hybrid attention with the CARRIED previous weights (attention_layer.py:191-265) -------------------
// f[b,t,c] = sum_j alpha_prev[b, t + j - P] F[j,c]   (tf.nn.conv1d(alpha [B,T,1], filter [taps,1,10], 'SAME'):
//            cross-correlation, P = (taps-1)/2 zero frames before, taps-1-P after; taps = 201 location / 200 hybrid)
// z[b,t,a] = (keys[t,b,a]) + qz[b,a] + sum_c f[b,t,c] W_filter[c,a]      (qz carries W_query s + b_filter)
// energy[b,t] = sum_a v_a tanh(z)
// The [T,B,A] location term is never materialised: a workgroup owns 64 frames of one utterance, stages the
// alpha window (64 + taps - 1 values), the filter (8 KB) and W_filter (5 KB) in LDS, forms its 64 x 10 features
// there and folds them into the energy pass that streams the keys once.
constexpr int LOC_C = 10;                       // feature channels of the reference's filter

__device__ __forceinline__ void loc_stage(const float* __restrict__ alpha_prev_b, const float* __restrict__ filt,
                                          int T, int taps, int t0, float* aw, float* fl, float* ff) {
  const int P = (taps - 1) / 2;
  for (int i = threadIdx.x; i < ATT_CH + taps - 1; i += 256) {
    const int u = t0 + i - P;
    aw[i] = (u >= 0 && u < T) ? alpha_prev_b[u] : 0.f;
  }
  for (int i = threadIdx.x; i < taps * LOC_C; i += 256) fl[i] = filt[i];
  __syncthreads();
  for (int o = threadIdx.x; o < ATT_CH * LOC_C; o += 256) {
    const int i = o / LOC_C, c = o % LOC_C;
    float s = 0.f;
    for (int j = 0; j < taps; ++j) s += aw[i + j] * fl[j * LOC_C + c];
    ff[o] = s;
  }
}
static inline size_t loc_lds_floats(int taps, int A) {   // aw | filter | features | W_filter
  return (size_t)(ATT_CH + taps - 1) + (size_t)taps * LOC_C + (size_t)ATT_CH * LOC_C + (size_t)LOC_C * A;
}

__global__ __launch_bounds__(256) void att_loc_energy_fwd_kernel(
    const float* __restrict__ alpha_prev, const float* __restrict__ filt, const float* __restrict__ wfil,
    const float* __restrict__ keys, const float* __restrict__ qz, const float* __restrict__ v, int T, int B, int A,
    int taps, float* __restrict__ energy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* aw = reinterpret_cast<float*>(smem);
  float* fl = aw + (ATT_CH + taps - 1);
  float* ff = fl + taps * LOC_C;
  float* ws = ff + ATT_CH * LOC_C;
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t0 = blockIdx.x * ATT_CH, t1 = min(T, t0 + ATT_CH);
  for (int i = threadIdx.x; i < LOC_C * A; i += 256) ws[i] = wfil[i];
  loc_stage(alpha_prev + (size_t)b * T, filt, T, taps, t0, aw, fl, ff);
  __syncthreads();
  const float* q = qz + (size_t)b * A;
  for (int t = t0 + wave; t < t1; t += 4) {
    const float* k = keys ? keys + ((size_t)t * B + b) * A : nullptr;
    const float* f = ff + (t - t0) * LOC_C;
    float s = 0.f;
    for (int a = lane; a < A; a += 64) {
      float z = (k ? k[a] : 0.f) + q[a];
#pragma unroll
      for (int c = 0; c < LOC_C; ++c) z += f[c] * ws[c * A + a];
      s += v[a] * tanhf(z);
    }
    s = wave_reduce_sum(s);
    if (lane == 0) energy[(size_t)b * T + t] = s;
  }
}

// Backward, pass A: dZ -> dkeys (+=), per-chunk partials part[ch][b][(2 + LOC_C) * A] = (sum_t dZ, sum_t de tanh(z),
// sum_t f[t,c] dZ[t,a]) and the feature gradients dfeat[b,t,c] = sum_a dZ[t,a] W_filter[c,a] for pass B.
__global__ __launch_bounds__(256) void att_loc_energy_bwd_kernel(
    const float* __restrict__ denergy, const float* __restrict__ alpha_prev, const float* __restrict__ filt,
    const float* __restrict__ wfil, const float* __restrict__ keys, const float* __restrict__ qz,
    const float* __restrict__ v, int T, int B, int A, int taps, float* __restrict__ dkeys, float* __restrict__ part,
    float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* aw = reinterpret_cast<float*>(smem);
  float* fl = aw + (ATT_CH + taps - 1);
  float* ff = fl + taps * LOC_C;
  float* ws = ff + ATT_CH * LOC_C;
  float* acc = ws + LOC_C * A;                   // [4 waves][(2 + LOC_C) * A]
  const int NP = (2 + LOC_C) * A;
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t0 = blockIdx.x * ATT_CH, t1 = min(T, t0 + ATT_CH);
  for (int i = threadIdx.x; i < LOC_C * A; i += 256) ws[i] = wfil[i];
  for (int i = threadIdx.x; i < 4 * NP; i += 256) acc[i] = 0.f;
  loc_stage(alpha_prev + (size_t)b * T, filt, T, taps, t0, aw, fl, ff);
  __syncthreads();
  const float* q = qz + (size_t)b * A;
  float* my = acc + wave * NP;
  for (int t = t0 + wave; t < t1; t += 4) {
    const float de = denergy[(size_t)b * T + t];
    const size_t off = ((size_t)t * B + b) * A;
    const float* f = ff + (t - t0) * LOC_C;
    float df[LOC_C];
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) df[c] = 0.f;
    for (int a = lane; a < A; a += 64) {
      float z = (keys ? keys[off + a] : 0.f) + q[a];
#pragma unroll
      for (int c = 0; c < LOC_C; ++c) z += f[c] * ws[c * A + a];
      const float th = tanhf(z);
      const float dz = de * v[a] * (1.f - th * th);
      if (dkeys) dkeys[off + a] += dz;
      my[a] += dz;
      my[A + a] += de * th;
#pragma unroll
      for (int c = 0; c < LOC_C; ++c) {
        my[(2 + c) * A + a] += f[c] * dz;
        df[c] += dz * ws[c * A + a];
      }
    }
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) df[c] = wave_reduce_sum(df[c]);
    if (lane < LOC_C) {
      float mine = 0.f;
#pragma unroll
      for (int c = 0; c < LOC_C; ++c) mine = (lane == c) ? df[c] : mine;
      dfeat[((size_t)b * T + t) * LOC_C + lane] = mine;
    }
  }
  __syncthreads();
  float* o = part + ((size_t)blockIdx.x * B + b) * NP;
  for (int i = threadIdx.x; i < NP; i += 256) o[i] = (acc[i] + acc[NP + i]) + (acc[2 * NP + i] + acc[3 * NP + i]);
}
// dqz, dv_rows [B,A] and dwfil_rows [B,LOC_C,A] (+= when `accumulate`): fixed-order sums of the chunk partials
__global__ void att_loc_bwd_reduce_kernel(const float* __restrict__ part, int nch, int B, int A,
                                          float* __restrict__ dqz, float* __restrict__ dv_rows,
                                          float* __restrict__ dwfil_rows, int accumulate) {
  const int NP = (2 + LOC_C) * A;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * NP) return;
  const int b = i / NP, r = i % NP;
  float s = 0.f;
  for (int c = 0; c < nch; ++c) s += part[((size_t)c * B + b) * NP + r];
  if (r < A) dqz[(size_t)b * A + r] = s;
  else if (r < 2 * A) { if (dv_rows) dv_rows[(size_t)b * A + r - A] = s; }
  else {
    float* o = dwfil_rows + (size_t)b * LOC_C * A + (r - 2 * A);
    *o = accumulate ? *o + s : s;
  }
}
// Backward, pass B: gradients through the convolution.
//   dalpha_prev[b,u] = sum_j sum_c dfeat[b, u - j + P, c] F[j,c]
//   dfilt partial of the chunk: sum_{t in chunk} alpha_prev[b, t + j - P] dfeat[b,t,c]  -> fpart[ch][b][taps*LOC_C]
__global__ __launch_bounds__(256) void att_loc_conv_bwd_kernel(
    const float* __restrict__ dfeat, const float* __restrict__ alpha_prev, const float* __restrict__ filt, int T,
    int B, int taps, float* __restrict__ dalpha_prev, float* __restrict__ fpart) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = ATT_CH + taps - 1;
  float* aw = reinterpret_cast<float*>(smem);   // alpha window  [W]
  float* fl = aw + W;                             // filter        [taps*LOC_C]
  float* dw = fl + taps * LOC_C;                  // dfeat window  [W*LOC_C]: frames t0 + P - (taps-1) ...
  const int P = (taps - 1) / 2;
  const int b = blockIdx.y, t0 = blockIdx.x * ATT_CH, n = min(ATT_CH, T - t0);
  for (int i = threadIdx.x; i < W; i += 256) {
    const int u = t0 + i - P;
    aw[i] = (u >= 0 && u < T) ? alpha_prev[(size_t)b * T + u] : 0.f;
  }
  for (int i = threadIdx.x; i < taps * LOC_C; i += 256) fl[i] = filt[i];
  const int tb = t0 + P - (taps - 1);
  for (int i = threadIdx.x; i < W * LOC_C; i += 256) {
    const int t = tb + i / LOC_C;
    dw[i] = (t >= 0 && t < T) ? dfeat[((size_t)b * T + t) * LOC_C + i % LOC_C] : 0.f;
  }
  __syncthreads();
  // dalpha: 4 lanes per output frame, each a quarter of the taps, combined with two DPP-free shuffles
  {
    const int i = threadIdx.x >> 2, r = threadIdx.x & 3;
    float s = 0.f;
    for (int j = r; j < taps; j += 4) {
      const float* d = dw + (i + (taps - 1) - j) * LOC_C;
      const float* f = fl + j * LOC_C;
#pragma unroll
      for (int c = 0; c < LOC_C; ++c) s += d[c] * f[c];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (r == 0 && i < n) dalpha_prev[(size_t)b * T + t0 + i] = s;
  }
  // dfilt partial: this chunk's own dfeat rows sit at window rows (taps-1-P) + i
  float* o = fpart + ((size_t)blockIdx.x * B + b) * taps * LOC_C;
  const int own = (taps - 1) - P;
  for (int q = threadIdx.x; q < taps * LOC_C; q += 256) {
    const int j = q / LOC_C, c = q % LOC_C;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += aw[i + j] * dw[(own + i) * LOC_C + c];
    o[q] = s;
  }
}
__global__ void att_loc_filt_reduce_kernel(const float* __restrict__ fpart, int nch, int B, int n,
                                           float* __restrict__ dfilt_rows, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const int b = i / n, r = i % n;
  float s = 0.f;
  for (int c = 0; c < nch; ++c) s += fpart[((size_t)c * B + b) * n + r];
  dfilt_rows[i] = accumulate ? dfilt_rows[i] + s : s;
}

// ---- the same three passes, MI355X-shaped (A % 4 == 0, A <= 256; the forms above stay as the general fallback and
// ran at 38 / 116 / 37 us per decoder step at T = 1600, A = 128, 201 taps -- LDS-bound: two LDS reads per multiply-add in
// the convolutions, twelve LDS read-modify-writes per (frame, column) in the backward):
//  * the convolution  f[64 x 10] = Toeplitz(alpha window)[64 x taps] . F[taps x 10]  and the filter gradient
//    dF[taps x 10] = Toeplitz^T[taps x 64] . dfeat[64 x 10]  are exact-fp32 MFMA products (16x16x4) whose A operand is
//    read straight out of the alpha window in LDS (one ds_read per MFMA and lane);
//  * the energy passes use the layout of att_energy_*_vec_kernel (LPF lanes x float4 per frame), W_filter in registers,
//    tanh from exp2 / rcp, and REGISTER accumulators for dq / dv / dW_filter (folded across lanes and waves once);
//  * only frames below the utterance's length are touched (everything past it is masked by the softmax: alpha, d-energy
//    and hence every gradient there are zero);
//  * the two partial-sum reductions are one launch.
constexpr int LOC_FS = 12;                      // floats per feature / filter row in LDS (16-byte aligned, 10 used)

// dst[i] = src(i), i < n, by the 256 threads of the workgroup, UNR loads in flight per thread before the first LDS write
// (a plain "load, write" loop is compiled to one global round trip per iteration: ~0.7 us each, 25 of them in pass B)
template <int UNR, typename F>
__device__ __forceinline__ void stage_lds(float* dst, int n, F src) {
  for (int base = 0; base < n; base += 256 * UNR) {
    float x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = base + u * 256 + (int)threadIdx.x;
      x[u] = i < n ? src(i) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = base + u * 256 + (int)threadIdx.x;
      if (i < n) dst[i] = x[u];
    }
  }
}
// ff[i][c] (row stride LOC_FS) = sum_j aw[i + j] F[j][c], i < 64.  aw is zero-padded to 64 + taps16 entries and fl is the
// filter as [taps16][16] with zero rows / columns past [taps][LOC_C] (taps16 = taps rounded up to 16), so that the loop
// is straight-line: with bounds checks on the operands the compiler wraps every LDS read in an exec-mask branch
__device__ __forceinline__ void loc_conv_mfma(const float* aw, const float* fl, int taps16, float* ff) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
  f32x4_t ac[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) ac[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float* arow = aw + wave * 16 + m + kg;
  const float* brow = fl + kg * 16 + m;
  for (int k0 = 0; k0 < taps16; k0 += 16) {
    float a[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = arow[k0 + 4 * u];
      bv[u] = brow[(k0 + 4 * u) * 16];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ac[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u], ac[u], 0, 0, 0);
  }
  f32x4_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r]);
  if (m < LOC_C) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ff[(wave * 16 + kg * 4 + r) * LOC_FS + m] = acc[r];
  }
}
__device__ __forceinline__ void loc_stage_window(const float* __restrict__ alpha_prev_b, const float* __restrict__ filt,
                                                 int T, int taps, int t0, float* aw, float* fl) {
  const int P = (taps - 1) / 2, taps16 = (taps + 15) & ~15;
  stage_lds<2>(aw, ATT_CH + taps16, [&](int i) {
    const int u = t0 + i - P;
    return (i < ATT_CH + taps - 1 && u >= 0 && u < T) ? alpha_prev_b[u] : 0.f;
  });
  stage_lds<8>(fl, taps16 * 16, [&](int i) {
    const int j = i >> 4, c = i & 15;
    return (j < taps && c < LOC_C) ? filt[j * LOC_C + c] : 0.f;
  });
}
static inline size_t loc_vec_lds_floats(int taps) {      // aw | filter | features
  const int taps16 = (taps + 15) & ~15;
  return (size_t)(ATT_CH + taps16) + (size_t)taps16 * 16 + (size_t)ATT_CH * LOC_FS;
}

template <int LPF, int NV>
__global__ __launch_bounds__(256) void att_loc_energy_fwd_vec_kernel(
    const float* __restrict__ alpha_prev, const float* __restrict__ filt, const float* __restrict__ wfil,
    const float* __restrict__ keys, const float* __restrict__ qz, const float* __restrict__ v, int T, int B, int A,
    int taps, float* __restrict__ energy, const int32_t* __restrict__ seq_len) {
  constexpr int FPW = 64 / LPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int taps16 = (taps + 15) & ~15;
  float* aw = reinterpret_cast<float*>(smem);
  float* fl = aw + ATT_CH + taps16;
  float* ff = fl + taps16 * 16;
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPF, l = lane % LPF, nvec = A >> 2;
  const int tend = seq_len ? min(max(seq_len[b], 0), T) : T;
  const int t0 = blockIdx.x * ATT_CH, t1 = min(tend, t0 + ATT_CH);
  if (t0 >= t1) return;                                    // uniform: the whole chunk is masked
  loc_stage_window(alpha_prev + (size_t)b * T, filt, T, taps, t0, aw, fl);
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  f32x4_t qv[NV], vv[NV], w[NV][LOC_C];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int a4 = l + i * LPF;
    const bool ok = a4 < nvec;
    qv[i] = ok ? *reinterpret_cast<const f32x4_t*>(qz + (size_t)b * A + a4 * 4) : zero;
    vv[i] = ok ? *reinterpret_cast<const f32x4_t*>(v + a4 * 4) : zero;
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) w[i][c] = ok ? *reinterpret_cast<const f32x4_t*>(wfil + (size_t)c * A + a4 * 4) : zero;
  }
  __syncthreads();
  loc_conv_mfma(aw, fl, taps16, ff);
  __syncthreads();
  for (int tb = t0 + wave * FPW; tb < t1; tb += 4 * FPW) {
    const int t = tb + sub;
    const bool valid = t < t1;
    const float* fr = ff + (valid ? t - t0 : 0) * LOC_FS;
    const f32x4_t f0 = *reinterpret_cast<const f32x4_t*>(fr), f1 = *reinterpret_cast<const f32x4_t*>(fr + 4),
                  f2 = *reinterpret_cast<const f32x4_t*>(fr + 8);
    const float f[LOC_C] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3], f2[0], f2[1]};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int a4 = l + i * LPF;
      f32x4_t z = qv[i];
      if (keys && valid && a4 < nvec) {
        const f32x4_t kv = *reinterpret_cast<const f32x4_t*>(keys + ((size_t)t * B + b) * A + a4 * 4);
        z[0] += kv[0]; z[1] += kv[1]; z[2] += kv[2]; z[3] += kv[3];
      }
#pragma unroll
      for (int c = 0; c < LOC_C; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] += f[c] * w[i][c][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) s += vv[i][e] * fast_tanhf(z[e]);
    }
    s = group_reduce_sum<LPF>(s);
    if (l == 0 && valid) energy[(size_t)b * T + t] = s;
  }
}

// pass A of the backward: part[ch][b][(2 + LOC_C) * A] and dfeat[b,t,c] as att_loc_energy_bwd_kernel
template <int LPF, int NV>
__global__ __launch_bounds__(256) void att_loc_energy_bwd_vec_kernel(
    const float* __restrict__ denergy, const float* __restrict__ alpha_prev, const float* __restrict__ filt,
    const float* __restrict__ wfil, const float* __restrict__ keys, const float* __restrict__ qz,
    const float* __restrict__ v, int T, int B, int A, int taps, float* __restrict__ dkeys, float* __restrict__ part,
    float* __restrict__ dfeat, const int32_t* __restrict__ seq_len) {
  constexpr int FPW = 64 / LPF, NR = 2 + LOC_C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int taps16 = (taps + 15) & ~15;
  float* aw = reinterpret_cast<float*>(smem);
  float* fl = aw + ATT_CH + taps16;
  float* ff = fl + taps16 * 16;
  float* red = ff + ATT_CH * LOC_FS;                       // [4 waves][NR * A]
  const int NP = NR * A;
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPF, l = lane % LPF, nvec = A >> 2;
  const int tend = seq_len ? min(max(seq_len[b], 0), T) : T;
  const int t0 = blockIdx.x * ATT_CH, t1 = min(tend, t0 + ATT_CH);
  float* o = part + ((size_t)blockIdx.x * B + b) * NP;
  if (t0 >= t1) {                                          // uniform: nothing but zero partials
    for (int i = threadIdx.x; i < NP; i += 256) o[i] = 0.f;
    return;
  }
  loc_stage_window(alpha_prev + (size_t)b * T, filt, T, taps, t0, aw, fl);
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  f32x4_t qv[NV], vv[NV], w[NV][LOC_C], acc[NV][NR];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int a4 = l + i * LPF;
    const bool ok = a4 < nvec;
    qv[i] = ok ? *reinterpret_cast<const f32x4_t*>(qz + (size_t)b * A + a4 * 4) : zero;
    vv[i] = ok ? *reinterpret_cast<const f32x4_t*>(v + a4 * 4) : zero;
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) w[i][c] = ok ? *reinterpret_cast<const f32x4_t*>(wfil + (size_t)c * A + a4 * 4) : zero;
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[i][r] = zero;
  }
  __syncthreads();
  loc_conv_mfma(aw, fl, taps16, ff);
  __syncthreads();
  for (int tb = t0 + wave * FPW; tb < t1; tb += 4 * FPW) {
    const int t = tb + sub;
    const bool valid = t < t1;
    const float de = valid ? denergy[(size_t)b * T + t] : 0.f;
    const float* fr = ff + (valid ? t - t0 : 0) * LOC_FS;
    const f32x4_t f0 = *reinterpret_cast<const f32x4_t*>(fr), f1 = *reinterpret_cast<const f32x4_t*>(fr + 4),
                  f2 = *reinterpret_cast<const f32x4_t*>(fr + 8);
    const float f[LOC_C] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3], f2[0], f2[1]};
    float df[LOC_C];
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) df[c] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int a4 = l + i * LPF;
      if (!(valid && a4 < nvec)) continue;
      const size_t off = ((size_t)t * B + b) * A + a4 * 4;
      f32x4_t z = qv[i];
      if (keys) {
        const f32x4_t kv = *reinterpret_cast<const f32x4_t*>(keys + off);
        z[0] += kv[0]; z[1] += kv[1]; z[2] += kv[2]; z[3] += kv[3];
      }
#pragma unroll
      for (int c = 0; c < LOC_C; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] += f[c] * w[i][c][e];
      f32x4_t dz;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float th = fast_tanhf(z[e]);
        dz[e] = de * vv[i][e] * (1.f - th * th);
        acc[i][0][e] += dz[e];
        acc[i][1][e] += de * th;
      }
#pragma unroll
      for (int c = 0; c < LOC_C; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[i][2 + c][e] += f[c] * dz[e];
          df[c] += dz[e] * w[i][c][e];
        }
      if (dkeys) {
        f32x4_t* dk = reinterpret_cast<f32x4_t*>(dkeys + off);
        f32x4_t x = *dk;
        x[0] += dz[0]; x[1] += dz[1]; x[2] += dz[2]; x[3] += dz[3];
        *dk = x;
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < LOC_C; ++c) {
      const float tot = group_reduce_sum<LPF>(df[c]);
      mine = (l == c) ? tot : mine;
    }
    if (l < LOC_C && valid) dfeat[((size_t)b * T + t) * LOC_C + l] = mine;
  }
  // fold the frame slots of the wave, then the four waves through LDS in a fixed order
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int x = LPF; x < 64; x <<= 1) acc[i][r][e] += __shfl_xor(acc[i][r][e], x, 64);
  if (sub == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int a4 = l + i * LPF;
      if (a4 < nvec) {
#pragma unroll
        for (int r = 0; r < NR; ++r) *reinterpret_cast<f32x4_t*>(&red[(size_t)wave * NP + r * A + a4 * 4]) = acc[i][r];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NP; i += 256) o[i] = (red[i] + red[NP + i]) + (red[2 * NP + i] + red[3 * NP + i]);
}

// pass B: dalpha_prev of the chunk's frames and the chunk's filter-gradient partial (as att_loc_conv_bwd_kernel);
// dfeat rows at or past the utterance's length are zero and are not read
__global__ __launch_bounds__(256) void att_loc_conv_bwd_vec_kernel(
    const float* __restrict__ dfeat, const float* __restrict__ alpha_prev, const float* __restrict__ filt, int T,
    int B, int taps, float* __restrict__ dalpha_prev, float* __restrict__ fpart, const int32_t* __restrict__ seq_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = ATT_CH + taps - 1, taps16 = (taps + 15) & ~15;
  float* aw = reinterpret_cast<float*>(smem);            // alpha window  [64 + taps16], zero past W
  float* fl = aw + ATT_CH + taps16;                        // filter        [taps][LOC_FS]
  float* dw = fl + (size_t)taps * LOC_FS;                  // dfeat window  [W][LOC_FS]: frames t0 + P - (taps-1) ...
  const int P = (taps - 1) / 2;
  const int b = blockIdx.y, t0 = blockIdx.x * ATT_CH, n = min(ATT_CH, T - t0);
  const int len = seq_len ? min(max(seq_len[b], 0), T) : T;
  const int tb = t0 + P - (taps - 1);
  float* o = fpart + ((size_t)blockIdx.x * B + b) * taps * LOC_C;
  if (tb >= len) {                                         // uniform: every dfeat row of the window is zero
    for (int i = threadIdx.x; i < taps * LOC_C; i += 256) o[i] = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) dalpha_prev[(size_t)b * T + t0 + i] = 0.f;
    return;
  }
  stage_lds<2>(aw, ATT_CH + taps16, [&](int i) {
    const int u = t0 + i - P;
    return (i < W && u >= 0 && u < T) ? alpha_prev[(size_t)b * T + u] : 0.f;
  });
  stage_lds<10>(fl, taps * LOC_FS, [&](int i) {
    const int j = i / LOC_FS, c = i % LOC_FS;
    return c < LOC_C ? filt[j * LOC_C + c] : 0.f;
  });
  stage_lds<13>(dw, W * LOC_FS, [&](int i) {
    const int t = tb + i / LOC_FS, c = i % LOC_FS;
    return (t >= 0 && t < len && c < LOC_C) ? dfeat[((size_t)b * T + t) * LOC_C + c] : 0.f;
  });
  __syncthreads();
  // dalpha[i] = sum_j dw[i + taps-1 - j] . fl[j]  (rows of LOC_FS floats, pad columns zero on both sides).  A thread
  // owns FOUR consecutive frames and one sixteenth of the taps: going up one tap slides its four dfeat rows down by
  // one, so a tap costs one new dfeat row and one filter row (6 ds_read_b128) for 40 multiply-adds; the sixteen tap
  // slices of a frame group are the sixteen lanes of a DPP row.
  {
    const int grp = threadIdx.x >> 4, part = threadIdx.x & 15;          // frames 4 grp .. 4 grp + 3
    const int per = (taps + 15) / 16, j0 = part * per, j1 = min(taps, j0 + per);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (j0 < j1) {
      f32x4_t row[4][3];                                                 // dw rows base + r, r = 0..3
      const int base0 = grp * 4 + (taps - 1) - j0;
#pragma unroll
      for (int r = 1; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) row[r][q] = reinterpret_cast<const f32x4_t*>(dw + (size_t)(base0 + r) * LOC_FS)[q];
      for (int j = j0; j < j1; ++j) {
        const f32x4_t* d = reinterpret_cast<const f32x4_t*>(dw + (size_t)(grp * 4 + (taps - 1) - j) * LOC_FS);
        const f32x4_t* f = reinterpret_cast<const f32x4_t*>(fl + (size_t)j * LOC_FS);
#pragma unroll
        for (int q = 0; q < 3; ++q) row[0][q] = d[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const f32x4_t y = f[q];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s[r] += row[r][q][0] * y[0] + row[r][q][1] * y[1] + row[r][q][2] * y[2] + row[r][q][3] * y[3];
        }
#pragma unroll
        for (int r = 3; r > 0; --r)
#pragma unroll
          for (int q = 0; q < 3; ++q) row[r][q] = row[r - 1][q];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = group_reduce_sum<16>(s[r]);
    if (part < 4 && grp * 4 + part < n) {
      const float mine = part == 0 ? s[0] : part == 1 ? s[1] : part == 2 ? s[2] : s[3];
      dalpha_prev[(size_t)b * T + t0 + grp * 4 + part] = mine;
    }
  }
  // filter-gradient partial  dF[j][c] = sum_i aw[i + j] dfeat[own + i][c]  as MFMA tiles of 16 taps
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
    const int own = (taps - 1) - P;
    for (int jt = wave; jt * 16 < taps; jt += 4) {
      // straight-line operands: rows j >= taps read the zero pad of aw, columns m >= LOC_C read whatever follows in the
      // dfeat row -- both only reach output rows / columns that are not stored
      const float* arow = aw + jt * 16 + m + kg;
      const float* brow = dw + (size_t)(own + kg) * LOC_FS + m;
      f32x4_t ac[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ac[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i0 = 0; i0 < ATT_CH; i0 += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          ac[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[i0 + 4 * u], brow[(size_t)(i0 + 4 * u) * LOC_FS], ac[u], 0, 0, 0);
      }
      f32x4_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = (ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r]);
      if (m < LOC_C) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jo = jt * 16 + kg * 4 + r;
          if (jo < taps) o[jo * LOC_C + m] = acc[r];
        }
      }
    }
  }
}
// both reductions of the backward in one launch: outputs [0, B*NP) are dqz / dv_rows / dwfil_rows, the rest dfilt_rows
__global__ void att_loc_reduce_both_kernel(const float* __restrict__ part, const float* __restrict__ fpart, int nch,
                                           int B, int A, int nf, float* __restrict__ dqz, float* __restrict__ dv_rows,
                                           float* __restrict__ dwfil_rows, float* __restrict__ dfilt_rows,
                                           int accumulate) {
  const int NP = (2 + LOC_C) * A;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool first = i < B * NP;
  const int k = first ? i : i - B * NP;
  if (!first && k >= B * nf) return;
  const int per = first ? NP : nf;
  const int b = k / per, r = k % per;
  const float* p = (first ? part : fpart) + (size_t)b * per + r;
  const size_t cs = (size_t)B * per;
  float s = 0.f;
  for (int c0 = 0; c0 < nch; c0 += 32) {
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = (c0 + c < nch) ? p[(size_t)(c0 + c) * cs] : 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) s += x[c];
  }
  if (!first) {
    dfilt_rows[k] = accumulate ? dfilt_rows[k] + s : s;
  } else if (r < A) {
    dqz[(size_t)b * A + r] = s;
  } else if (r < 2 * A) {
    if (dv_rows) dv_rows[(size_t)b * A + r - A] = s;
  } else {
    float* o = dwfil_rows + (size_t)b * LOC_C * A + (r - 2 * A);
    *o = accumulate ? *o + s : s;
  }
}
// 0: the general kernels, else LPF * 4 + NV
static inline int loc_vec_shape(int A, const void* keys, const void* qz, const void* v, const void* wfil, const void* dkeys) {
  if (A % 4 != 0 || A > 256 ||
      ((uintptr_t)keys | (uintptr_t)qz | (uintptr_t)v | (uintptr_t)wfil | (uintptr_t)dkeys) % 16 != 0)
    return 0;
  const int nvec = A / 4;
  return nvec <= 16 ? 16 * 4 + 1 : nvec <= 32 ? 32 * 4 + 1 : 64 * 4 + 1;
}

__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                float* __restrict__ dx, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

// out[r,:] = W[ids[r],:]
__global__ void emb_gather_kernel(const float* __restrict__ W, const int32_t* __restrict__ ids, int R, int E,
                                  float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)R * E; i += (size_t)gridDim.x * blockDim.x)
    out[i] = W[(size_t)ids[i / E] * E + i % E];
}
// dW[v,:] = sum_{r: ids[r]==v} dout[r,:]   (one block per vocabulary row, fixed order: deterministic)
__global__ __launch_bounds__(256) void emb_scatter_kernel(const float* __restrict__ dout,
                                                          const int32_t* __restrict__ ids, int R, int E,
                                                          float* __restrict__ dW) {
  const int v = blockIdx.x;
  for (int e0 = threadIdx.x; e0 < E; e0 += 256) {
    float s = 0.f;
    for (int r = 0; r < R; ++r)
      if (ids[r] == v) s += dout[(size_t)r * E + e0];
    dW[(size_t)v * E + e0] = s;
  }
}

// same, E % 4 == 0 and E <= 1024: a thread owns 4 consecutive embedding columns, the 256 / (E/4) row groups of a block
// scan interleaved rows and are combined in a fixed order through LDS (deterministic; 16x fewer serial rows per thread
// at E = 64: 1.8 ms -> ~0.1 ms for the 12.8 k decoder inputs of a cfg-D step)
constexpr int EMB_TILE = 4096;
__global__ __launch_bounds__(256) void emb_scatter_v4_kernel(const float* __restrict__ dout,
                                                             const int32_t* __restrict__ ids, int R, int E,
                                                             float* __restrict__ dW) {
  extern __shared__ float part[];                          // [groups][E], then EMB_TILE ids
  const int v = blockIdx.x;
  const int lpr = E >> 2, groups = 256 / lpr;
  const int g = threadIdx.x / lpr, l = threadIdx.x % lpr;
  int* tile = reinterpret_cast<int*>(part + (size_t)groups * E);
  int* list = tile + EMB_TILE;                             // [groups][per]: matching rows of the tile, per group
  const int per = (EMB_TILE + groups - 1) / groups;
  __shared__ int cnts[256];
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  // (round 6: the ids come through LDS, EMB_TILE at a time, loaded by the whole block, and a group's matching rows are
  // listed first and then fetched eight at a time -- as `if (ids[r] == v) acc += dout[r]` on global memory every group waited
  // for one load per row and one more per match, and the padding id matches half the 12.8 k decoder inputs of a cfg-D step:
  // 322 us on 30 of 256 CUs.  Same rows per group in the same order.)
  for (int t0 = 0; t0 < R; t0 += EMB_TILE) {               // block-uniform
    const int n = min(EMB_TILE, R - t0);
    for (int i = threadIdx.x; i < n; i += 256) tile[i] = ids[t0 + i];
    __syncthreads();
    if (g < groups && l == 0) {                            // one lane per group lists the group's matching rows of the tile
      int i = g - t0 % groups;                             // first row of this tile with (t0 + i) % groups == g
      if (i < 0) i += groups;
      int c = 0;
      int* my = list + g * per;
      for (; i < n; i += groups)
        if (tile[i] == v) my[c++] = i;
      cnts[g] = c;
    }
    __syncthreads();
    if (g < groups) {                                      // eight rows in flight, added in row order
      const int c = cnts[g];
      const int* my = list + g * per;
      const float* base = dout + (size_t)t0 * E + l * 4;
      int k = 0;
      for (; k + 8 <= c; k += 8) {
        f32x4_t d[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = *reinterpret_cast<const f32x4_t*>(base + (size_t)my[k + q] * E);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += d[q];
      }
      for (; k < c; ++k) acc += *reinterpret_cast<const f32x4_t*>(base + (size_t)my[k] * E);
    }
    __syncthreads();
  }
  if (g < groups) *reinterpret_cast<f32x4_t*>(part + (size_t)g * E + l * 4) = acc;
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) {
    float sum = 0.f;
    for (int k = 0; k < groups; ++k) sum += part[(size_t)k * E + e];
    dW[(size_t)v * E + e] = sum;
  }
}

// masked sequence cross-entropy: rows = B*To; per-row loss*w and dlogits = (softmax-onehot)*w*scale
__global__ __launch_bounds__(256) void seq_xent_kernel(const float* __restrict__ logits,
                                                       const int32_t* __restrict__ targets,
                                                       const float* __restrict__ weights, int rows, int Cc,
                                                       float eps, float dscale, float* __restrict__ row_loss,
                                                       float* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = logits + (size_t)row * Cc;
  const float w = weights[row];
  if (w == 0.f) {   // masked position: exact zero (never 0 * NaN)
    if (lane == 0) row_loss[row] = 0.f;
    if (dlogits)
      for (int k = lane; k < Cc; k += 64) dlogits[(size_t)row * Cc + k] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int k = lane; k < Cc; k += 64) m = fmaxf(m, p[k] + eps);
  m = wave_reduce_max(m);
  float s = 0.f;
  for (int k = lane; k < Cc; k += 64) s += expf(p[k] + eps - m);
  s = wave_reduce_sum(s);
  const float lse = m + logf(s);
  const int tg = targets[row];
  if (lane == 0) row_loss[row] = w * (lse - (p[tg] + eps));
  if (dlogits) {
    const float sc = w * dscale;
    for (int k = lane; k < Cc; k += 64)
      dlogits[(size_t)row * Cc + k] = (expf(p[k] + eps - lse) - (k == tg ? 1.f : 0.f)) * sc;
  }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int rows, int Cc,
                                                          int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + (size_t)row * Cc;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int k = lane; k < Cc; k += 64) {
    const float v = p[k];
    if (v > best || (v == best && k < bi)) { best = v; bi = k; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

inline int gridn(size_t n) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

}  // namespace

namespace {
// ---- query-FC backward + cell backward of one decoder step, one launch (the reverse loop's third and fourth kernels
// and the partial-sum reduction before them):
//   dqz[b,:]  = fixed-order sum over the frame chunks of the energy backward's partials      (also dv_rows)
//   dcell     = dcell_in + dqz . W_q^T            [B,U]   exact-fp32 MFMA, K = A
//   cell backward on dcell (cell_bwd_kernel's arithmetic) in the epilogue
// One workgroup = 16 utterances x 16 units; wave w multiplies the 64-wide slice w of A (A <= 512), its A fragment
// being the SUM of the chunk partials read straight from the energy kernel's scratch; the tiles meet in LDS in a fixed
// order.  Column-tile 0 also writes dqz / dv_rows (the weight gradients after the loop need them).
struct CellBwdArgs {
  const float *dc_next, *dh_next, *dh_next2, *gates, *c_raw, *c_prev, *peep, *live, *use_mask;
  float *dpre, *dc_prev, *dh_prev_carry, *dpeep_rows;
  int ld2;
  float clip;
};
// NTL MFMA column tiles (16 NTL units) per workgroup.  Measured on the cfg D shard (A = 128, U = 512, 25 chunks): NTL = 1
// (34 workgroups x 2 row groups) 54.3 ms for the reverse pass, NTL = 4 (10 x 2) 55.4, the three separate launches 55.0.
constexpr int DQ_NTL = 1, DQ_CT = 16 * DQ_NTL;
__global__ __launch_bounds__(512) void att_dq_cell_bwd_kernel(const float* __restrict__ part, int nch, int B, int A, int U,
                                                              const float* __restrict__ Wq, int ldw,
                                                              const float* __restrict__ dcell_in,
                                                              float* __restrict__ dqz_out, float* __restrict__ dv_out,
                                                              CellBwdArgs c) {
  // grid.x = U / DQ_CT column blocks + 2 "writer" workgroups (dqz, dv_rows); A / 64 = units in {1, 2, 4, 8}.  Every
  // column block re-reads the chunk partials of its 16 utterances (the price of not having a reduction launch).
  __shared__ f32x4_t asum[8][4][64];            // per wave: its share of a slice's chunk sum, in MFMA A-fragment layout
  __shared__ float red[8][16][DQ_CT + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const int nblk = U / DQ_CT;
  const int writer = (int)blockIdx.x - nblk;    // < 0: a column block; 0: writes dqz; 1: writes dv_rows
  const int n0 = blockIdx.x * DQ_CT, m0 = blockIdx.y * 16;
  const int units = A / 64, wpu = 8 / units;     // waves per 64-wide slice
  const int row = min(m0 + col, B - 1);
  const size_t cs = (size_t)B * 2 * A;
  if (writer == 1 && !dv_out) return;
  // wave w: slice u = w / wpu, share si = w % wpu = the chunks si, si + wpu, ... of that slice.  One load phase for the
  // whole workgroup, the shares meet in LDS and the first wave of each slice adds them in a fixed order.
  const int u = wave / wpu, si = wave % wpu;
  const int kb = u * 64 + rg * 4;
  // everything the multiply and the epilogue read from global memory is requested up front, so that the chunk
  // partials are the only round trip on the kernel's chain (each of these was one more ~1 us hop behind a barrier)
  f32x4_t bw[DQ_NTL][4];
#pragma unroll
  for (int t = 0; t < DQ_NTL; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bw[t][j] = (writer < 0 && si == 0)
                     ? *reinterpret_cast<const f32x4_t*>(Wq + (size_t)(n0 + t * 16 + col) * ldw + kb + j * 16)
                     : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int idx = threadIdx.x;
  // epilogue: output o = idx + 512 h2 of the [16][DQ_CT] tile
  constexpr int EPT = (16 * DQ_CT + 511) / 512;
  float e_dcell[EPT], e_dhn[EPT], e_dcn[EPT], e_g[EPT][4], e_cr[EPT], e_cp[EPT], e_w[EPT][3], e_mask[EPT], e_lv[EPT];
  if (writer < 0) {
#pragma unroll
    for (int h2 = 0; h2 < EPT; ++h2) {
      const int o = min(idx + 512 * h2, 16 * DQ_CT - 1);
      const int eb = min(m0 + o / DQ_CT, B - 1), ej = n0 + o % DQ_CT;
      const size_t bu = (size_t)eb * U + ej;
      e_lv[h2] = c.live[eb];
      e_dcell[h2] = dcell_in[bu];
      e_dhn[h2] = c.dh_next[bu];
      if (c.dh_next2) e_dhn[h2] += c.dh_next2[(size_t)eb * c.ld2 + ej];
      e_dcn[h2] = c.dc_next[bu];
#pragma unroll
      for (int q = 0; q < 4; ++q) e_g[h2][q] = c.gates[(size_t)eb * 4 * U + q * U + ej];
      e_cr[h2] = c.c_raw[bu];
      e_cp[h2] = c.c_prev[bu];
#pragma unroll
      for (int q = 0; q < 3; ++q) e_w[h2][q] = c.peep ? c.peep[q * U + ej] : 0.f;
      e_mask[h2] = c.use_mask ? c.use_mask[bu] : 1.f;
    }
  }
  {
    const float* p = part + (size_t)row * 2 * A + (writer == 1 ? A : 0) + kb;
    f32x4_t sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int c0 = si; c0 < nch; c0 += 8 * wpu) {
      f32x4_t x[8][4];
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          x[q][j] = (c0 + q * wpu < nch) ? *reinterpret_cast<const f32x4_t*>(p + (size_t)(c0 + q * wpu) * cs + j * 16)
                                         : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sh[j][0] += x[q][j][0]; sh[j][1] += x[q][j][1]; sh[j][2] += x[q][j][2]; sh[j][3] += x[q][j][3];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) asum[wave][j][lane] = sh[j];
  }
  __syncthreads();
  f32x4_t acc[DQ_NTL];
#pragma unroll
  for (int t = 0; t < DQ_NTL; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if (si == 0) {
    f32x4_t a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4_t t = asum[wave][j][lane];
      for (int w = 1; w < wpu; ++w) {
        const f32x4_t y = asum[wave + w][j][lane];
        t[0] += y[0]; t[1] += y[1]; t[2] += y[2]; t[3] += y[3];
      }
      a[j] = t;
    }
    if (writer >= 0) {
      if (m0 + col < B) {
        float* o = (writer == 1 ? dv_out : dqz_out) + (size_t)row * A + kb;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(o + j * 16) = a[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < DQ_NTL; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], bw[t][j][e], acc[t], 0, 0, 0);
    }
  }
  if (writer >= 0) return;
#pragma unroll
  for (int t = 0; t < DQ_NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][rg * 4 + r][t * 16 + col] = acc[t][r];
  __syncthreads();
#pragma unroll
  for (int h2 = 0; h2 < EPT; ++h2) {
    const int o = idx + 512 * h2;
    if (o >= 16 * DQ_CT) continue;
    const int em = o / DQ_CT, en = o % DQ_CT, b = m0 + em, j = n0 + en;
    if (b >= B) continue;
    const size_t bu = (size_t)b * U + j;
    float dq = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) dq += red[w][em][en];
    const float dcell = e_dcell[h2] + dq;
    // ---- cell_bwd_kernel's arithmetic on (b, j)
    float* dp = c.dpre + (size_t)b * 4 * U;
    if (!(e_lv[h2] > 0.f)) {
      dp[j] = 0.f; dp[U + j] = 0.f; dp[2 * U + j] = 0.f; dp[3 * U + j] = 0.f;
      c.dc_prev[bu] = e_dcn[h2];
      c.dh_prev_carry[bu] = e_dhn[h2];
      if (c.dpeep_rows) {
        c.dpeep_rows[(size_t)b * 3 * U + j] = 0.f; c.dpeep_rows[(size_t)b * 3 * U + U + j] = 0.f;
        c.dpeep_rows[(size_t)b * 3 * U + 2 * U + j] = 0.f;
      }
      continue;
    }
    const float gi = e_g[h2][0], gg = e_g[h2][1], gf = e_g[h2][2], go = e_g[h2][3];
    const float dh = dcell * e_mask[h2] + e_dhn[h2];
    const float tc = tanhf(e_cr[h2]);
    const float d_o = dh * tc * go * (1.f - go);
    const float dct = e_dcn[h2] + dh * go * (1.f - tc * tc) + d_o * e_w[h2][2];
    const float dc = (c.clip > 0.f && fabsf(e_cr[h2]) >= c.clip) ? 0.f : dct;     // a clamped state passes nothing back
    const float d_g = dc * gi * (1.f - gg * gg);
    const float d_i = dc * gg * gi * (1.f - gi);
    const float d_f = dc * e_cp[h2] * gf * (1.f - gf);
    dp[j] = d_i; dp[U + j] = d_g; dp[2 * U + j] = d_f; dp[3 * U + j] = d_o;
    c.dc_prev[bu] = dc * gf + d_i * e_w[h2][0] + d_f * e_w[h2][1];
    c.dh_prev_carry[bu] = 0.f;
    if (c.dpeep_rows) {
      c.dpeep_rows[(size_t)b * 3 * U + j] = d_i * e_cp[h2];
      c.dpeep_rows[(size_t)b * 3 * U + U + j] = d_f * e_cp[h2];
      c.dpeep_rows[(size_t)b * 3 * U + 2 * U + j] = d_o * e_cr[h2];
    }
  }
}
}  // namespace

#define ATT_NEED(cond, ...) do { if (!(cond)) ASR_FAIL(h, ASR_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

extern "C" int asr_lstm_cell_fwd_ex(asr_handle* h, const float* pre, const float* c_prev, const float* h_prev,
                                    const float* peep, const float* live, int B, int U, float forget_bias,
                                    float cell_clip, float* gates, float* c_raw, float* c_out, float* h_out,
                                    float* h_raw, const float* out_mask, float* cell_out, float* h_out2, int ld_h2,
                                    float* cell_out2, int ld_c2, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(pre && c_prev && h_prev && live && gates && c_raw && c_out && h_out && h_raw && B > 0 && U > 0 &&
               (!h_out2 || ld_h2 >= U) && (!cell_out2 || ld_c2 >= U), "asr_lstm_cell_fwd: bad args");
  hipLaunchKernelGGL(cell_fwd_kernel, dim3((B * U + 255) / 256), dim3(256), 0, (hipStream_t)s, pre, c_prev, h_prev,
                     peep, live, B, U, forget_bias, cell_clip, gates, c_raw, c_out, h_out, h_raw, out_mask, cell_out,
                     h_out2, ld_h2, cell_out2, ld_c2);
  ASR_CHECK_LAUNCH(h, "asr_lstm_cell_fwd");
  return ASR_OK;
}
extern "C" int asr_lstm_cell_fwd(asr_handle* h, const float* pre, const float* c_prev, const float* h_prev,
                                 const float* peep, const float* live, int B, int U, float forget_bias,
                                 float cell_clip, float* gates, float* c_raw, float* c_out, float* h_out,
                                 float* h_raw, asr_stream s) {
  return asr_lstm_cell_fwd_ex(h, pre, c_prev, h_prev, peep, live, B, U, forget_bias, cell_clip, gates, c_raw, c_out,
                              h_out, h_raw, nullptr, nullptr, nullptr, 0, nullptr, 0, s);
}

static int cell_bwd_launch(asr_handle* h, const float* dh_use, const float* dc_next, const float* dh_next,
                           const float* gates, const float* c_raw, const float* c_prev, const float* peep,
                           const float* live, int B, int U, float* dpre, float* dc_prev, float* dh_prev_carry,
                           float* dpeep_rows, const float* dh_next2, int ld2, const float* use_mask, float clip,
                           asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(dh_use && dc_next && dh_next && gates && c_raw && c_prev && live && dpre && dc_prev && dh_prev_carry &&
               B > 0 && U > 0 && (!dh_next2 || ld2 >= U), "asr_lstm_cell_bwd: bad args");
  hipLaunchKernelGGL(cell_bwd_kernel, dim3((B * U + 255) / 256), dim3(256), 0, (hipStream_t)s, dh_use, dc_next,
                     dh_next, gates, c_raw, c_prev, peep, live, B, U, dpre, dc_prev, dh_prev_carry, dpeep_rows,
                     dh_next2, ld2, use_mask, clip);
  ASR_CHECK_LAUNCH(h, "asr_lstm_cell_bwd");
  return ASR_OK;
}
extern "C" int asr_lstm_cell_bwd(asr_handle* h, const float* dh_use, const float* dc_next, const float* dh_next,
                                 const float* gates, const float* c_raw, const float* c_prev, const float* peep,
                                 const float* live, int B, int U, float* dpre, float* dc_prev,
                                 float* dh_prev_carry, float* dpeep_rows, asr_stream s) {
  return cell_bwd_launch(h, dh_use, dc_next, dh_next, gates, c_raw, c_prev, peep, live, B, U, dpre, dc_prev,
                         dh_prev_carry, dpeep_rows, nullptr, 0, nullptr, 0.f, s);
}
extern "C" int asr_lstm_cell_bwd_ex(asr_handle* h, const float* dh_use, const float* dc_next, const float* dh_next,
                                    const float* gates, const float* c_raw, const float* c_prev, const float* peep,
                                    const float* live, int B, int U, float cell_clip, float* dpre, float* dc_prev,
                                    float* dh_prev_carry, float* dpeep_rows, asr_stream s) {
  return cell_bwd_launch(h, dh_use, dc_next, dh_next, gates, c_raw, c_prev, peep, live, B, U, dpre, dc_prev,
                         dh_prev_carry, dpeep_rows, nullptr, 0, nullptr, cell_clip, s);
}

static int energy_fwd_launch(asr_handle* h, const float* keys, const float* qz, const float* v, int T, int B, int A,
                             int mode, float* energy, const int32_t* seq_len, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(qz && energy && T > 0 && B > 0 && A > 0 && (mode == 0 ? v != nullptr : keys != nullptr),
           "asr_att_energy_fwd: bad args");
  const dim3 grid((T + ATT_CH - 1) / ATT_CH, B);
#define ASR_EFWD(L, NV_) \
  hipLaunchKernelGGL((att_energy_fwd_vec_kernel<L, NV_>), grid, dim3(256), 0, (hipStream_t)s, keys, qz, v, T, B, A, mode, energy, seq_len)
  switch (energy_vec_shape(A, keys, qz, v, nullptr)) {
    case 16 * 4 + 1: ASR_EFWD(16, 1); break;
    case 32 * 4 + 1: ASR_EFWD(32, 1); break;
    case 64 * 4 + 1: ASR_EFWD(64, 1); break;
    case 64 * 4 + 2: ASR_EFWD(64, 2); break;
    default:
      hipLaunchKernelGGL(att_energy_fwd_kernel, grid, dim3(256), 0, (hipStream_t)s, keys, qz, v, T, B, A, mode, energy,
                         seq_len);
  }
#undef ASR_EFWD
  ASR_CHECK_LAUNCH(h, "asr_att_energy_fwd");
  return ASR_OK;
}
extern "C" int asr_att_energy_fwd(asr_handle* h, const float* keys, const float* qz, const float* v, int T,
                                  int B, int A, int mode, float* energy, asr_stream s) {
  return energy_fwd_launch(h, keys, qz, v, T, B, A, mode, energy, nullptr, s);
}

static inline float* att_scratch(asr_handle* h, size_t bytes) {
  return (bytes <= h->scratch_bytes - ASR_XCH_BYTES) ? (float*)h->scratch : nullptr;
}

static int energy_bwd_launch(asr_handle* h, const float* denergy, const float* keys, const float* qz, const float* v,
                             int T, int B, int A, int mode, float* dkeys, float* dqz, float* dv_rows,
                             const int32_t* seq_len, const SoftmaxBwdFold* fold, asr_stream s,
                             const float** part_out = nullptr, int* nch_out = nullptr) {
  // part_out / nch_out: the caller sums the chunk partials itself (att_dq_cell_bwd_kernel does it on its operand
  // load) -- no reduction launch here, dqz / dv_rows are not written
  if (!h) return ASR_ERR_INVALID_ARG;
  const int shape = energy_vec_shape(A, keys, qz, v, dkeys);
  ATT_NEED((denergy || (fold && fold->da && shape)) && qz && dqz && T > 0 && B > 0 && A > 0,
           "asr_att_energy_bwd: bad args");
  // frames per workgroup: 64 (measured at T = 1600, A = 128 without keys: 256-frame workgroups -- a 4x shorter
  // reduction -- run 20 us instead of 7, one wave per SIMD cannot hide the exp / rcp latency)
  const int ech = ATT_CH;
  const int nch = (T + ech - 1) / ech;
  // a folded softmax backward reads d-alpha and the partial dots from the head of the scratch: the partials go behind
  const size_t skip = (fold && fold->da) ? (((size_t)B * T + (size_t)fold->ndot * B + 3) & ~(size_t)3) : 0;
  float* base = att_scratch(h, (skip + (size_t)nch * B * 2 * A) * sizeof(float));
  if (!base) ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_att_energy_bwd: scratch too small");
  float* part = base + skip;
  const size_t lds = (size_t)8 * A * sizeof(float);
  const dim3 grid(nch, B);
  SoftmaxBwdFold f = {nullptr, nullptr, nullptr, nullptr, 0, 0.f};
  if (fold && fold->da) f = *fold;
#define ASR_EBWD(L, NV_) \
  hipLaunchKernelGGL((att_energy_bwd_vec_kernel<L, NV_>), grid, dim3(256), 0, (hipStream_t)s, denergy, keys, qz, v, T, B, A, mode, dkeys, part, seq_len, ech, f)
  switch (shape) {
    case 16 * 4 + 1: ASR_EBWD(16, 1); break;
    case 32 * 4 + 1: ASR_EBWD(32, 1); break;
    case 64 * 4 + 1: ASR_EBWD(64, 1); break;
    case 64 * 4 + 2: ASR_EBWD(64, 2); break;
    default:
      hipLaunchKernelGGL(att_energy_bwd_kernel, grid, dim3(256), lds, (hipStream_t)s, denergy, keys, qz, v, T, B, A, mode,
                         dkeys, part, seq_len);
  }
#undef ASR_EBWD
  if (part_out) {
    *part_out = part;
    *nch_out = nch;
  } else {
    hipLaunchKernelGGL(att_energy_bwd_reduce_kernel, dim3((B * 2 * A + 255) / 256), dim3(256), 0, (hipStream_t)s, part,
                       nch, B, A, dqz, dv_rows);
  }
  ASR_CHECK_LAUNCH(h, "asr_att_energy_bwd");
  return ASR_OK;
}
extern "C" int asr_att_energy_bwd(asr_handle* h, const float* denergy, const float* keys, const float* qz,
                                  const float* v, int T, int B, int A, int mode, float* dkeys, float* dqz,
                                  float* dv_rows, asr_stream s) {
  return energy_bwd_launch(h, denergy, keys, qz, v, T, B, A, mode, dkeys, dqz, dv_rows, nullptr, nullptr, s);
}

static int loc_energy_fwd_launch(asr_handle* h, const float* alpha_prev, const float* filt, const float* wfil,
                                 const float* keys, const float* qz, const float* v, int T, int B, int A, int taps,
                                 float* energy, const int32_t* seq_len, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(alpha_prev && filt && wfil && qz && v && energy && T > 0 && B > 0 && A > 0 && taps > 0,
           "asr_att_loc_energy_fwd: bad args");
  const dim3 grid((T + ATT_CH - 1) / ATT_CH, B);
  const int shape = loc_vec_shape(A, keys, qz, v, wfil, nullptr);
  const size_t ldsv = loc_vec_lds_floats(taps) * sizeof(float);
  if (shape && ldsv <= 64 * 1024) {
#define ASR_LFWD(L) \
  hipLaunchKernelGGL((att_loc_energy_fwd_vec_kernel<L, 1>), grid, dim3(256), ldsv, (hipStream_t)s, alpha_prev, filt, wfil, keys, qz, v, T, B, A, taps, energy, seq_len)
    if (shape == 16 * 4 + 1) ASR_LFWD(16); else if (shape == 32 * 4 + 1) ASR_LFWD(32); else ASR_LFWD(64);
#undef ASR_LFWD
    ASR_CHECK_LAUNCH(h, "asr_att_loc_energy_fwd");
    return ASR_OK;
  }
  const size_t lds = loc_lds_floats(taps, A) * sizeof(float);
  if (lds > 64 * 1024) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_att_loc_energy_fwd: taps=%d / A=%d need %zu B of LDS", taps, A, lds);
  hipLaunchKernelGGL(att_loc_energy_fwd_kernel, grid, dim3(256), lds, (hipStream_t)s, alpha_prev, filt, wfil, keys, qz, v,
                     T, B, A, taps, energy);
  ASR_CHECK_LAUNCH(h, "asr_att_loc_energy_fwd");
  return ASR_OK;
}
extern "C" int asr_att_loc_energy_fwd(asr_handle* h, const float* alpha_prev, const float* filt, const float* wfil,
                                      const float* keys, const float* qz, const float* v, int T, int B, int A,
                                      int taps, float* energy, asr_stream s) {
  return loc_energy_fwd_launch(h, alpha_prev, filt, wfil, keys, qz, v, T, B, A, taps, energy, nullptr, s);
}

static int loc_energy_bwd_launch(asr_handle* h, const float* denergy, const float* alpha_prev, const float* filt,
                                 const float* wfil, const float* keys, const float* qz, const float* v, int T, int B,
                                 int A, int taps, float* dkeys, float* dqz, float* dv_rows, float* dwfil_rows,
                                 float* dfilt_rows, float* dalpha_prev, int accumulate, const int32_t* seq_len,
                                 asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(denergy && alpha_prev && filt && wfil && qz && v && dqz && dwfil_rows && dfilt_rows && dalpha_prev &&
               T > 0 && B > 0 && A > 0 && taps > 0, "asr_att_loc_energy_bwd: bad args");
  const int nch = (T + ATT_CH - 1) / ATT_CH;
  const size_t NP = (size_t)(2 + LOC_C) * A;
  const size_t n_part = (size_t)nch * B * NP, n_feat = (size_t)B * T * LOC_C, n_fp = (size_t)nch * B * taps * LOC_C;
  float* part = att_scratch(h, (n_part + n_feat + n_fp) * sizeof(float));
  if (!part) ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_att_loc_energy_bwd: scratch too small");
  float* dfeat = part + n_part;
  float* fpart = dfeat + n_feat;
  const dim3 grid(nch, B);
  const int shape = loc_vec_shape(A, keys, qz, v, wfil, dkeys);
  const size_t ldsAv = (loc_vec_lds_floats(taps) + 4 * NP) * sizeof(float);
  const size_t ldsBv = ((size_t)(ATT_CH + ((taps + 15) & ~15)) + (size_t)taps * LOC_FS +
                        (size_t)(ATT_CH + taps - 1) * LOC_FS + 16) * sizeof(float);
  if (shape && ldsAv <= 64 * 1024 && ldsBv <= 64 * 1024) {
#define ASR_LBWD(L) \
  hipLaunchKernelGGL((att_loc_energy_bwd_vec_kernel<L, 1>), grid, dim3(256), ldsAv, (hipStream_t)s, denergy, alpha_prev, filt, wfil, keys, qz, v, T, B, A, taps, dkeys, part, dfeat, seq_len)
    if (shape == 16 * 4 + 1) ASR_LBWD(16); else if (shape == 32 * 4 + 1) ASR_LBWD(32); else ASR_LBWD(64);
#undef ASR_LBWD
    hipLaunchKernelGGL(att_loc_conv_bwd_vec_kernel, grid, dim3(256), ldsBv, (hipStream_t)s, dfeat, alpha_prev, filt, T, B,
                       taps, dalpha_prev, fpart, seq_len);
    const size_t nout = (size_t)B * NP + (size_t)B * taps * LOC_C;
    hipLaunchKernelGGL(att_loc_reduce_both_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       part, fpart, nch, B, A, taps * LOC_C, dqz, dv_rows, dwfil_rows, dfilt_rows, accumulate);
    ASR_CHECK_LAUNCH(h, "asr_att_loc_energy_bwd");
    return ASR_OK;
  }
  const size_t ldsA = (loc_lds_floats(taps, A) + 4 * NP) * sizeof(float);
  const size_t ldsB = ((size_t)(ATT_CH + taps - 1) * (1 + LOC_C) + (size_t)taps * LOC_C) * sizeof(float);
  if (ldsA > 64 * 1024 || ldsB > 64 * 1024)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_att_loc_energy_bwd: taps=%d / A=%d need %zu / %zu B of LDS", taps, A, ldsA, ldsB);
  hipLaunchKernelGGL(att_loc_energy_bwd_kernel, grid, dim3(256), ldsA, (hipStream_t)s, denergy, alpha_prev, filt,
                     wfil, keys, qz, v, T, B, A, taps, dkeys, part, dfeat);
  hipLaunchKernelGGL(att_loc_bwd_reduce_kernel, dim3((unsigned)((B * NP + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                     part, nch, B, A, dqz, dv_rows, dwfil_rows, accumulate);
  hipLaunchKernelGGL(att_loc_conv_bwd_kernel, grid, dim3(256), ldsB, (hipStream_t)s, dfeat, alpha_prev, filt, T,
                     B, taps, dalpha_prev, fpart);
  hipLaunchKernelGGL(att_loc_filt_reduce_kernel, dim3((B * taps * LOC_C + 255) / 256), dim3(256), 0, (hipStream_t)s,
                     fpart, nch, B, taps * LOC_C, dfilt_rows, accumulate);
  ASR_CHECK_LAUNCH(h, "asr_att_loc_energy_bwd");
  return ASR_OK;
}
extern "C" int asr_att_loc_energy_bwd(asr_handle* h, const float* denergy, const float* alpha_prev,
                                      const float* filt, const float* wfil, const float* keys, const float* qz,
                                      const float* v, int T, int B, int A, int taps, float* dkeys, float* dqz,
                                      float* dv_rows, float* dwfil_rows, float* dfilt_rows, float* dalpha_prev,
                                      int accumulate, asr_stream s) {
  return loc_energy_bwd_launch(h, denergy, alpha_prev, filt, wfil, keys, qz, v, T, B, A, taps, dkeys, dqz, dv_rows,
                               dwfil_rows, dfilt_rows, dalpha_prev, accumulate, nullptr, s);
}

extern "C" int asr_att_softmax_ctx_fwd_ex(asr_handle* h, const float* energy, const int32_t* seq_len,
                                          float sharpening, const void* enc, int enc_dtype, int T, int B, int E,
                                          float* alpha, float* ctx, float* sigmoid_norm, float* ctx2, int ld2,
                                          float* ctx3, int ld3, asr_stream s);
extern "C" int asr_att_softmax_ctx_fwd(asr_handle* h, const float* energy, const int32_t* seq_len,
                                       float sharpening, const void* enc, int enc_dtype, int T, int B, int E,
                                       float* alpha, float* ctx, float* sigmoid_norm, asr_stream s) {
  return asr_att_softmax_ctx_fwd_ex(h, energy, seq_len, sharpening, enc, enc_dtype, T, B, E, alpha, ctx, sigmoid_norm,
                                    nullptr, 0, nullptr, 0, s);
}
extern "C" int asr_att_softmax_ctx_fwd_ex(asr_handle* h, const float* energy, const int32_t* seq_len,
                                          float sharpening, const void* enc, int enc_dtype, int T, int B, int E,
                                          float* alpha, float* ctx, float* sigmoid_norm, float* ctx2, int ld2,
                                          float* ctx3, int ld3, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(energy && seq_len && enc && alpha && ctx && T > 0 && B > 0 && E > 0 && asr_dtype_ok(enc_dtype) &&
               (!ctx2 || ld2 >= E) && (!ctx3 || ld3 >= E), "asr_att_softmax_ctx_fwd: bad args");
  const size_t lds = (size_t)T * sizeof(float);
  if (lds > 64 * 1024) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_att_softmax_ctx_fwd: T=%d too long", T);
  const int nch = (T + ATT_CH - 1) / ATT_CH;
  float* part = att_scratch(h, (size_t)nch * B * E * sizeof(float));
  if (!part) ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_att_softmax_ctx_fwd: scratch too small");
  hipLaunchKernelGGL(att_softmax_kernel, dim3(B), dim3(256), lds, (hipStream_t)s, energy, seq_len, sharpening, T, alpha,
                     sigmoid_norm);
  if (enc_dtype == ASR_F32)
    hipLaunchKernelGGL(att_ctx_partial_kernel<float>, dim3(nch, B), dim3(256), 0, (hipStream_t)s, alpha, seq_len,
                       (const float*)enc, T, B, E, part);
  else
    hipLaunchKernelGGL(att_ctx_partial_kernel<bf16_t>, dim3(nch, B), dim3(256), 0, (hipStream_t)s, alpha, seq_len,
                       (const bf16_t*)enc, T, B, E, part);
  hipLaunchKernelGGL(att_ctx_reduce_kernel, dim3((B * E + 255) / 256), dim3(256), 0, (hipStream_t)s, part, nch, B * E,
                     ctx, E, ctx2, ld2, ctx3, ld3);
  ASR_CHECK_LAUNCH(h, "asr_att_softmax_ctx_fwd");
  return ASR_OK;
}

namespace {
__global__ void add_cols_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                float* __restrict__ out, int ldo, int B, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, j = i % W;
  out[(size_t)b * ldo + j] = x[(size_t)b * ldx + j] + y[(size_t)b * ldy + j];
}
template <typename TE>
bool dalpha_vec_launch(const float* da_, const float* db_, int ldb, float* dout, const int32_t* seq_len, const TE* enc,
                       int T, int B, int E, float* da, const float* alpha, float* dotp, hipStream_t st) {
  constexpr int N = EncVec<TE>::N;
  if (E % (64 * N) != 0 || ((uintptr_t)enc) % 16 != 0 || ((uintptr_t)da_) % 16 != 0 ||
      (db_ && (((uintptr_t)db_) % 16 != 0 || ldb % 4 != 0)) || (dout && ((uintptr_t)dout) % 16 != 0))
    return false;
  const dim3 grid((T + ATT_CH - 1) / ATT_CH, B);
#define ASR_DALPHA(G_) \
  hipLaunchKernelGGL((att_dalpha_vec_kernel<TE, G_>), grid, dim3(256), 0, st, da_, db_, ldb, dout, seq_len, enc, T, B, E, da, alpha, dotp)
  switch (E / (64 * N)) {
    case 1: ASR_DALPHA(1); return true;
    case 2: ASR_DALPHA(2); return true;
    case 4: ASR_DALPHA(4); return true;
    default: return false;
  }
#undef ASR_DALPHA
}
}  // namespace

// dctx = dctx_a (+ dctx_b with row stride ldb, may be NULL); the sum goes to dctx_out when that is given.
// fold (may be NULL): the caller's energy backward can take the softmax backward in (vectorised kernels on both sides,
// no carried-alpha gradient); when that applies, *fold is filled, denergy is NOT written and no softmax backward
// kernel is launched -- otherwise fold->da is left NULL and denergy holds the result as usual.
static int softmax_ctx_bwd_launch(asr_handle* h, const float* dctx_a, const float* dctx_b, int ldb, float* dctx_out,
                                  const float* alpha, const int32_t* seq_len, float sharpening, const void* enc,
                                  int enc_dtype, int T, int B, int E, float* denergy, float* denc,
                                  const float* sigmoid_norm, const float* dalpha_extra, SoftmaxBwdFold* fold,
                                  asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(dctx_a && alpha && seq_len && enc && denergy && T > 0 && B > 0 && E > 0 && asr_dtype_ok(enc_dtype) &&
               (!dctx_b || (dctx_out && ldb >= E)), "asr_att_softmax_ctx_bwd: bad args");
  hipStream_t st = (hipStream_t)s;
  const int nch = (T + ATT_CH - 1) / ATT_CH;
  float* da = att_scratch(h, ((size_t)B * T + (size_t)nch * B) * sizeof(float));
  if (!da) ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_att_softmax_ctx_bwd: scratch too small");
  const bool want_fold = fold && !dalpha_extra && !denc;
  float* dotp = want_fold ? da + (size_t)B * T : nullptr;
  if (fold) fold->da = nullptr;
  const bool fast = enc_dtype == ASR_F32
                        ? dalpha_vec_launch<float>(dctx_a, dctx_b, ldb, dctx_out, seq_len, (const float*)enc, T, B, E, da, alpha, dotp, st)
                        : dalpha_vec_launch<bf16_t>(dctx_a, dctx_b, ldb, dctx_out, seq_len, (const bf16_t*)enc, T, B, E, da, alpha, dotp, st);
  const float* dctx = dctx_a;
  if (!fast) {
    if (dctx_b) {
      hipLaunchKernelGGL(add_cols_kernel, dim3((B * E + 255) / 256), dim3(256), 0, st, dctx_a, E, dctx_b, ldb, dctx_out,
                         E, B, E);
      dctx = dctx_out;
    }
    if (enc_dtype == ASR_F32)
      hipLaunchKernelGGL(att_dalpha_kernel<float>, dim3(nch, B), dim3(256), 0, st, dctx, seq_len, (const float*)enc, T,
                         B, E, da);
    else
      hipLaunchKernelGGL(att_dalpha_kernel<bf16_t>, dim3(nch, B), dim3(256), 0, st, dctx, seq_len, (const bf16_t*)enc,
                         T, B, E, da);
  } else if (dctx_b) {
    dctx = dctx_out;
  }
  if (fast && want_fold) {
    fold->da = da;
    fold->alpha = alpha;
    fold->dotp = dotp;
    fold->norm = sigmoid_norm;
    fold->ndot = nch;
    fold->sharp = sharpening;
    ASR_CHECK_LAUNCH(h, "asr_att_softmax_ctx_bwd");
    return ASR_OK;
  }
  hipLaunchKernelGGL(att_softmax_bwd_kernel, dim3(B), dim3(256), 0, st, da, alpha, seq_len, sharpening, T, denergy,
                     sigmoid_norm, dalpha_extra);
  if (denc)   // NULL: the caller accumulates d_enc = sum_steps alpha (x) dctx itself (one GEMM per utterance)
    hipLaunchKernelGGL(att_denc_kernel, dim3(nch, B), dim3(256), 0, st, dctx, alpha, seq_len, T, B, E, denc);
  ASR_CHECK_LAUNCH(h, "asr_att_softmax_ctx_bwd");
  return ASR_OK;
}
extern "C" int asr_att_softmax_ctx_bwd(asr_handle* h, const float* dctx, const float* alpha,
                                       const int32_t* seq_len, float sharpening, const void* enc, int enc_dtype,
                                       int T, int B, int E, float* denergy, float* denc,
                                       const float* sigmoid_norm, const float* dalpha_extra, asr_stream s) {
  return softmax_ctx_bwd_launch(h, dctx, nullptr, 0, nullptr, alpha, seq_len, sharpening, enc, enc_dtype, T, B, E,
                                denergy, denc, sigmoid_norm, dalpha_extra, nullptr, s);
}

extern "C" int asr_tanh_fwd(asr_handle* h, const float* x, float* y, size_t n, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(x && y, "asr_tanh_fwd: null");
  if (!n) return ASR_OK;
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(gridn(n)), dim3(256), 0, (hipStream_t)s, x, y, n);
  ASR_CHECK_LAUNCH(h, "asr_tanh_fwd");
  return ASR_OK;
}
extern "C" int asr_tanh_bwd(asr_handle* h, const float* dy, const float* y, float* dx, size_t n, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(dy && y && dx, "asr_tanh_bwd: null");
  if (!n) return ASR_OK;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(gridn(n)), dim3(256), 0, (hipStream_t)s, dy, y, dx, n);
  ASR_CHECK_LAUNCH(h, "asr_tanh_bwd");
  return ASR_OK;
}
extern "C" int asr_embedding_gather(asr_handle* h, const float* W, const int32_t* ids, int rows, int E,
                                    float* out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(W && ids && out && rows >= 0 && E > 0, "asr_embedding_gather: bad args");
  if (!rows) return ASR_OK;
  hipLaunchKernelGGL(emb_gather_kernel, dim3(gridn((size_t)rows * E)), dim3(256), 0, (hipStream_t)s, W, ids, rows, E, out);
  ASR_CHECK_LAUNCH(h, "asr_embedding_gather");
  return ASR_OK;
}
extern "C" int asr_embedding_scatter(asr_handle* h, const float* dout, const int32_t* ids, int rows, int E,
                                     int vocab, float* dW, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(dout && ids && dW && rows >= 0 && E > 0 && vocab > 0, "asr_embedding_scatter: bad args");
  if (E % 4 == 0 && E <= 1024 && ((uintptr_t)dout) % 16 == 0) {
    const int groups = 256 / (E / 4);
    hipLaunchKernelGGL(emb_scatter_v4_kernel, dim3(vocab), dim3(256), (size_t)groups * E * sizeof(float) + (size_t)(2 * EMB_TILE + groups) * sizeof(int), (hipStream_t)s,
                       dout, ids, rows, E, dW);
  } else {
    hipLaunchKernelGGL(emb_scatter_kernel, dim3(vocab), dim3(256), 0, (hipStream_t)s, dout, ids, rows, E, dW);
  }
  ASR_CHECK_LAUNCH(h, "asr_embedding_scatter");
  return ASR_OK;
}
extern "C" int asr_seq_xent(asr_handle* h, const float* logits, const int32_t* targets, const float* weights,
                            int rows, int C, float eps, float dscale, float* row_loss, float* dlogits,
                            asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(logits && targets && weights && row_loss && rows >= 0 && C > 0, "asr_seq_xent: bad args");
  if (!rows) return ASR_OK;
  hipLaunchKernelGGL(seq_xent_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, logits, targets, weights,
                     rows, C, eps, dscale, row_loss, dlogits);
  ASR_CHECK_LAUNCH(h, "asr_seq_xent");
  return ASR_OK;
}
extern "C" int asr_argmax_rows(asr_handle* h, const float* x, int rows, int C, int32_t* out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(x && out && rows >= 0 && C > 0, "asr_argmax_rows: bad args");
  if (!rows) return ASR_OK;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, x, rows, C, out);
  ASR_CHECK_LAUNCH(h, "asr_argmax_rows");
  return ASR_OK;
}

// ---------------------------------------------------------------- the decoder loop, native
extern "C" int asr_add_cols(asr_handle* h, const float* x, int ldx, const float* y, int ldy, float* out, int ldo, int B,
                            int W, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ATT_NEED(x && y && out && B > 0 && W > 0 && ldx >= W && ldy >= W && ldo >= W, "asr_add_cols: bad args");
  hipLaunchKernelGGL(add_cols_kernel, dim3((B * W + 255) / 256), dim3(256), 0, (hipStream_t)s, x, ldx, y, ldy, out, ldo,
                     B, W);
  ASR_CHECK_LAUNCH(h, "asr_add_cols");
  return ASR_OK;
}

#define DEC_TRY(call) do { const int rc_ = (call); if (rc_ != ASR_OK) return rc_; } while (0)

static int dec_check(asr_handle* h, const asr_att_decoder* a, bool bwd) {
  if (!a || a->To < 1 || a->B < 1 || a->T < 1 || a->U < 1 || a->Em < 0 || a->E2 < 1 || a->A < 1 || !a->W_cell ||
      !a->b_cell || !a->enc || !a->seq_len || !a->live || !a->dec_in || !a->av_in || !a->alpha_all || !a->gates_all ||
      !a->craw_all || !a->c_all || !a->h_all || !a->qz_all || !a->work || (a->att_mode != 0 && a->att_mode != 1) ||
      (a->has_query_fc && !a->W_q) || (!a->has_query_fc && a->A != a->U) || (a->att_mode == 0 && !a->v) ||
      (a->att_mode == 1 && !a->keys) || (a->carry_alpha && (!a->filt || !a->wfil || !a->alpha_zero || a->taps < 1)))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_att_decoder: bad arguments");
  if (bwd && (!a->dav_cell || !a->dav_ctx || !a->dctx_all || !a->dpre_all || !a->dqz_all || !a->d_in_all || !a->dc0 ||
              !a->dh0 || (a->att_mode == 0 && !a->dv_all) || (a->peep && !a->dpeep_all) ||
              (a->carry_alpha && (!a->dwfil_rows || !a->dfilt_rows))))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_att_decoder_bwd: bad arguments");
  return ASR_OK;
}

// energies + softmax + context of one decoder step as two launches (att_fused_fwd_kernel + combine); false = the shape is
// not covered (the caller takes the four-launch path).  ASR_ATT_FUSED=0 switches it off (A/B).
static bool att_fused_step(asr_handle* h, const asr_att_decoder* a, const float* qz, float* alpha, float* ctx, float* ctx2,
                           int ld2, float* ctx3, int ld3, asr_stream s) {
  static const bool on = [] { const char* e = getenv("ASR_ATT_FUSED"); return !(e && e[0] == '0'); }();
  const int B = a->B, T = a->T, E = a->E2, A = a->A;
  if (!on || a->carry_alpha || a->snorm_all || E % 256 != 0 || T > 64 * ATT_CH || ((uintptr_t)a->enc) % 16 != 0) return false;
  const int shape = energy_vec_shape(A, a->keys, qz, a->v, nullptr);
  if (!shape) return false;
  // 64-frame chunks; 32 (twice the workgroups, half the serial frames of the context phase) measured 86.5 vs 83.6 ms
  // per cfg D step (ASR_ATT_FUSED_CH=32 selects it)
  static const int fch_env = [] { const char* e = getenv("ASR_ATT_FUSED_CH"); return e ? atoi(e) : 64; }();
  const int fch = (fch_env == 32 && T <= 2048) ? 32 : 64;
  const int nch = (T + fch - 1) / fch;
  float* part = att_scratch(h, ((size_t)nch * B * E + (size_t)nch * B * 2) * sizeof(float));
  if (!part) return false;
  float* stat = part + (size_t)nch * B * E;
  const dim3 grid(nch, B);
  hipStream_t st = (hipStream_t)s;
#define ASR_FUSED(L, NV_, TE_, F_) \
  hipLaunchKernelGGL((att_fused_fwd_kernel<L, NV_, TE_, F_>), grid, dim3(256), 0, st, a->keys, qz, a->v, T, B, A, a->att_mode, \
                     a->sharpening, a->seq_len, (const TE_*)a->enc, E, alpha, part, stat)
#define ASR_FUSED_T(L, NV_) do { \
    if (a->enc_dtype == ASR_F32) { if (fch == 32) ASR_FUSED(L, NV_, float, 32); else ASR_FUSED(L, NV_, float, 64); } \
    else { if (fch == 32) ASR_FUSED(L, NV_, bf16_t, 32); else ASR_FUSED(L, NV_, bf16_t, 64); } } while (0)
  switch (shape) {
    case 16 * 4 + 1: ASR_FUSED_T(16, 1); break;
    case 32 * 4 + 1: ASR_FUSED_T(32, 1); break;
    case 64 * 4 + 1: ASR_FUSED_T(64, 1); break;
    default: ASR_FUSED_T(64, 2); break;
  }
#undef ASR_FUSED_T
#undef ASR_FUSED
  const int nb_ctx = B * (E / 256), nb_al = B * ((T + 255) / 256);
  hipLaunchKernelGGL(att_fused_combine_kernel, dim3(nb_ctx + nb_al), dim3(256), 0, st, part, stat, nch, fch, B, E, T, a->seq_len,
                     ctx, ctx2, ld2, ctx3, ld3, alpha);
  return true;
}

// The fused cell launch of a step (asr_lstm_cell_gemm_fwd) needs the caller's work space for the interleaved weight image
static inline bool dec_cell_gemm(const asr_att_decoder* a) {
  const int Din = a->Em + a->E2 + a->U;
  return a->W_cell_il && asr_lstm_cell_gemm_ok(a->B, Din, a->U, Din) && ((uintptr_t)a->dec_in) % 16 == 0 &&
         ((uintptr_t)a->W_cell_il) % 16 == 0;
}
// ... or for the bf16 weight images (both loops)
static inline bool dec_cell_gemm_h(const asr_att_decoder* a) {
  const int Din = a->Em + a->E2 + a->U;
  return a->W_cell_h && asr_lstm_cell_gemm_ok(a->B, Din, a->U, Din) && a->U % 16 == 0 && ((uintptr_t)a->dec_in) % 16 == 0 &&
         ((uintptr_t)a->W_cell_h) % 16 == 0;
}
static int dec_cell_image(asr_handle* h, const asr_att_decoder* a, asr_stream s) {
  if (dec_cell_gemm_h(a)) return asr_lstm_cell_gemm_prep_h(h, a->W_cell, a->b_cell, a->Em + a->E2 + a->U, a->U, a->W_cell_h, s);
  if (!dec_cell_gemm(a)) return ASR_OK;
  return asr_lstm_cell_gemm_prep(h, a->W_cell, a->b_cell, a->Em + a->E2 + a->U, a->U, a->W_cell_il, s);
}

// One forward step of the decoder loop: cell-input GEMM -> cell -> query FC -> energies -> softmax + context.  k indexes
// the step arrays that carry the recurrence (dec_in, av_in, c / h, alpha, live); ks the saved activations only the
// backward reads (gates, raw cell, query: the inference loop reuses row 0).
static int dec_fwd_step(asr_handle* h, const asr_att_decoder* a, int k, int ks, bool more, asr_stream s) {
  const int B = a->B, U = a->U, T = a->T, E2 = a->E2, Em = a->Em, A = a->A;
  const int Din = Em + E2 + U, Dav = U + E2;
  float* pre = a->work;                                    // [B,4U]
  float* hraw = pre + (size_t)B * 4 * U;                   // [B,U]
  float* energy = hraw + (size_t)B * U;                    // [B,T]
  float* ctx = energy + (size_t)B * T;                     // [B,E2]
  float* din = a->dec_in + (size_t)k * B * Din;
  float* dnext = more ? din + (size_t)B * Din : nullptr;
  float* av = a->av_in + (size_t)k * B * Dav;
  float* qz = a->qz_all + (size_t)ks * B * A;
  // the cell output (times its dropout mask) lands in av[:, :U]; without a query FC it IS the query
  if (dec_cell_gemm_h(a)) {  // product + cell as one launch on the bf16 fragment image (dec_cell_image)
    DEC_TRY(asr_lstm_cell_gemm_fwd_h(h, din, Din, Din, a->W_cell_h, a->c_all + (size_t)k * B * U,
                                     a->h_all + (size_t)k * B * U, a->peep, a->live + (size_t)k * B, B, U, a->forget_bias,
                                     a->cell_clip, a->gates_all + (size_t)ks * B * 4 * U, a->craw_all + (size_t)ks * B * U,
                                     a->c_all + (size_t)(k + 1) * B * U, a->h_all + (size_t)(k + 1) * B * U, hraw,
                                     a->dmask ? a->dmask + (size_t)k * B * U : nullptr, a->has_query_fc ? nullptr : qz,
                                     dnext ? dnext + Em + E2 : nullptr, Din, av, Dav, s));
  } else if (dec_cell_gemm(a)) {    // ... on the fp32 interleaved weight image
    DEC_TRY(asr_lstm_cell_gemm_fwd(h, din, Din, Din, a->W_cell_il, a->b_cell ? 1 : 0, a->c_all + (size_t)k * B * U,
                                   a->h_all + (size_t)k * B * U, a->peep, a->live + (size_t)k * B, B, U, a->forget_bias,
                                   a->cell_clip, a->gates_all + (size_t)ks * B * 4 * U, a->craw_all + (size_t)ks * B * U,
                                   a->c_all + (size_t)(k + 1) * B * U, a->h_all + (size_t)(k + 1) * B * U, hraw,
                                   a->dmask ? a->dmask + (size_t)k * B * U : nullptr, a->has_query_fc ? nullptr : qz,
                                   dnext ? dnext + Em + E2 : nullptr, Din, av, Dav, s));
  } else {
    DEC_TRY(asr_gemm_act(h, ASR_F32, ASR_F32, 0, 0, B, 4 * U, Din, din, Din, a->W_cell, 4 * U, pre, 4 * U, a->b_cell, 0, 0, s));
    DEC_TRY(asr_lstm_cell_fwd_ex(h, pre, a->c_all + (size_t)k * B * U, a->h_all + (size_t)k * B * U, a->peep,
                                 a->live + (size_t)k * B, B, U, a->forget_bias, a->cell_clip,
                                 a->gates_all + (size_t)ks * B * 4 * U, a->craw_all + (size_t)ks * B * U,
                                 a->c_all + (size_t)(k + 1) * B * U, a->h_all + (size_t)(k + 1) * B * U, hraw,
                                 a->dmask ? a->dmask + (size_t)k * B * U : nullptr, a->has_query_fc ? nullptr : qz,
                                 dnext ? dnext + Em + E2 : nullptr, Din, av, Dav, s));
  }
  if (a->has_query_fc)
    DEC_TRY(asr_gemm_act(h, ASR_F32, ASR_F32, 0, 0, B, A, U, av, Dav, a->W_q, a->ld_wq, qz, A, a->b_q, 0, 0, s));
  if (att_fused_step(h, a, qz, a->alpha_all + (size_t)k * B * T, ctx, av + U, Dav, dnext ? dnext + Em : nullptr, Din, s)) {
    ASR_CHECK_LAUNCH(h, "asr_att_decoder_fwd(fused step)");
    return ASR_OK;
  }
  if (a->carry_alpha)
    DEC_TRY(loc_energy_fwd_launch(h, k > 0 ? a->alpha_all + (size_t)(k - 1) * B * T : a->alpha_zero, a->filt, a->wfil,
                                  a->keys, qz, a->v, T, B, A, a->taps, energy, a->seq_len, s));
  else
    DEC_TRY(energy_fwd_launch(h, a->keys, qz, a->v, T, B, A, a->att_mode, energy, a->seq_len, s));
  DEC_TRY(asr_att_softmax_ctx_fwd_ex(h, energy, a->seq_len, a->sharpening, a->enc, a->enc_dtype, T, B, E2,
                                     a->alpha_all + (size_t)k * B * T, ctx,
                                     a->snorm_all ? a->snorm_all + (size_t)k * B : nullptr, av + U, Dav,
                                     dnext ? dnext + Em : nullptr, Din, s));
  return ASR_OK;
}

extern "C" int asr_att_decoder_fwd(asr_handle* h, const asr_att_decoder* a, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  DEC_TRY(dec_check(h, a, false));
  const int B = a->B, U = a->U, T = a->T, E2 = a->E2, Em = a->Em, A = a->A;
  const int Din = Em + E2 + U, Dav = U + E2;
  DEC_TRY(dec_cell_image(h, a, s));
  for (int k = 0; k < a->To; ++k) DEC_TRY(dec_fwd_step(h, a, k, k, k + 1 < a->To, s));
  (void)B; (void)U; (void)T; (void)E2; (void)Em; (void)A; (void)Din; (void)Dav;
  return ASR_OK;
}

// ---------------------------------------------------------------- greedy inference loop, native
namespace {
// One workgroup per batch row, behind the output layer of decoder step k: id = argmax(logits) (first maximum wins, as
// asr_argmax_rows), emitted id = id for a row that was live at the start of the step and 0 otherwise (impute_finished),
// live[k+1] = live[k] && id != eos, live_count[k+1] += live[k+1]; the NEXT step's input row gets the embedding of id
// (whatever the row's state: GreedyEmbeddingHelper.next_inputs does not look at `finished`) and its context columns
// are zeroed for rows that had finished before this step (dynamic_decode zeroes next_inputs' attention part with the
// imputed outputs, dynamic_decoder.py:172-190).
__global__ __launch_bounds__(256) void att_infer_select_kernel(const float* __restrict__ logits, int C2, int eos,
                                                               const float* __restrict__ live_k, float* __restrict__ live_n,
                                                               int32_t* __restrict__ count_n, int32_t* __restrict__ ids,
                                                               const float* __restrict__ emb, int Em, float* __restrict__ dnext,
                                                               int Din, int E2) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* p = logits + (size_t)b * C2;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int k = threadIdx.x; k < C2; k += 256) {
    const float v = p[k];
    if (v > best || (v == best && k < bi)) { best = v; bi = k; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { sv[w] = best; si[w] = bi; }
  __syncthreads();
  best = sv[0]; bi = si[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (sv[i] > best || (sv[i] == best && si[i] < bi)) { best = sv[i]; bi = si[i]; }
  const int id = bi == 0x7fffffff ? 0 : bi;
  const float lv = live_k[b];
  if (threadIdx.x == 0) {
    ids[b] = lv != 0.f ? id : 0;
    const float ln = (lv != 0.f && id != eos) ? 1.f : 0.f;
    live_n[b] = ln;
    if (ln != 0.f) atomicAdd(count_n, 1);
  }
  if (dnext) {
    float* d = dnext + (size_t)b * Din;
    const float* e = emb + (size_t)id * Em;
    for (int j = threadIdx.x; j < Em; j += 256) d[j] = e[j];
    if (lv == 0.f)
      for (int j = threadIdx.x; j < E2; j += 256) d[Em + j] = 0.f;
  }
}
}  // namespace

extern "C" int asr_att_decoder_infer(asr_handle* h, const asr_att_decoder* a, const asr_att_infer* f, int* steps_issued,
                                     asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  DEC_TRY(dec_check(h, a, false));
  if (!f || !f->W_av || !f->W_out || !f->embedding || !f->live || !f->av_all || !f->logits_all || !f->ids_all ||
      !f->live_count || f->C2 < 1 || f->eos < 0 || f->live != a->live || a->dmask)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_att_decoder_infer: bad arguments");
  const int B = a->B, U = a->U, E2 = a->E2, Em = a->Em, To = a->To, C2 = f->C2;
  const int Din = Em + E2 + U, Dav = U + E2;
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(f->live_count + 1, 0, (size_t)To * sizeof(int32_t), st) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_att_decoder_infer: memset");
  // early exit without draining the pipeline: every `check_every` steps the count of live rows at that step is copied to
  // the caller's pinned words behind the step's kernels.  A check point looks at the newest copy that has ALREADY landed
  // and, so that the host cannot run arbitrarily far past the end of the decode (it enqueues a step in a fraction of the
  // time the device takes to run one), waits for the copy of two check points ago: the issue loop stays 2 .. 3 intervals
  // ahead of the device, never idle, and at most that many surplus steps are enqueued.  The result does not depend on how
  // many were: a step with no live row changes nothing but its own (imputed: zero) outputs, and the caller trims at the
  // first such step.
  constexpr int NEV = 3;
  hipEvent_t ev[NEV] = {nullptr, nullptr, nullptr};
  const int every = (f->host_live_count && f->check_every > 0) ? f->check_every : 0;
  if (every) {
    for (int i = 0; i < NEV; ++i)
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
        for (int j = 0; j < i; ++j) (void)hipEventDestroy(ev[j]);          // the ones already made
        ASR_FAIL(h, ASR_ERR_HIP, "asr_att_decoder_infer: event");
      }
  }
  int rc = dec_cell_image(h, a, s), k = 0;
  for (; rc == ASR_OK && k < To; ++k) {
    if (every && k % every == 0 && k > 0) {
      const int c = k / every;                             // this check point; c - 1 was recorded one interval ago
      bool done = false;
      if (c >= 3) {
        (void)hipEventSynchronize(ev[(c - 2) % NEV]);
        done = f->host_live_count[(c - 2) * every] == 0;
      }
      if (!done && c >= 2 && hipEventQuery(ev[(c - 1) % NEV]) == hipSuccess) done = f->host_live_count[(c - 1) * every] == 0;
      if (done) break;
      if (hipMemcpyAsync(f->host_live_count + k, f->live_count + k, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipEventRecord(ev[c % NEV], st) != hipSuccess) { rc = ASR_ERR_HIP; break; }
    }
    const bool more = k + 1 < To;
    // (saved-activation arrays the backward would need are reused every step: index 0)
    if ((rc = dec_fwd_step(h, a, k, 0, more, s)) != ASR_OK) break;
    float* av = f->av_all + (size_t)k * B * U;
    float* lg = f->logits_all + (size_t)k * B * C2;
    if ((rc = asr_gemm_act(h, ASR_F32, ASR_F32, 0, 0, B, U, Dav, a->av_in + (size_t)k * B * Dav, Dav, f->W_av, U, av, U, nullptr,
                           0, 0, s)) != ASR_OK) break;
    if ((rc = asr_tanh_fwd(h, av, av, (size_t)B * U, s)) != ASR_OK) break;
    if ((rc = asr_gemm_act(h, ASR_F32, ASR_F32, 0, 0, B, C2, U, av, U, f->W_out, C2, lg, C2, f->b_out, 0, 0, s)) != ASR_OK) break;
    hipLaunchKernelGGL(att_infer_select_kernel, dim3(B), dim3(256), 0, st, lg, C2, f->eos, f->live + (size_t)k * B,
                       f->live + (size_t)(k + 1) * B, f->live_count + k + 1, f->ids_all + (size_t)k * B, f->embedding, Em,
                       more ? a->dec_in + (size_t)(k + 1) * B * Din : nullptr, Din, E2);
  }
  if (every)
    for (int i = 0; i < NEV; ++i) (void)hipEventDestroy(ev[i]);
  if (rc != ASR_OK) ASR_FAIL(h, rc, "asr_att_decoder_infer: step %d failed", k);
  ASR_CHECK_LAUNCH(h, "asr_att_decoder_infer");
  if (steps_issued) *steps_issued = k;
  return ASR_OK;
}

extern "C" int asr_att_decoder_bwd(asr_handle* h, const asr_att_decoder* a, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  DEC_TRY(dec_check(h, a, true));
  const int B = a->B, U = a->U, T = a->T, E2 = a->E2, Em = a->Em, A = a->A, To = a->To;
  const int Din = Em + E2 + U;
  hipStream_t st = (hipStream_t)s;
  // work: dc / carried-dh state (two buffers each, ping-pong), denergy, dalpha_prev x2
  float* dcs[2] = {a->work, a->work + (size_t)B * U};
  float* dhc[2] = {a->work + (size_t)2 * B * U, a->work + (size_t)3 * B * U};
  float* dalp[2] = {a->work + (size_t)4 * B * U, a->work + (size_t)4 * B * U + (size_t)B * T};
  float* denergy = dalp[1] + (size_t)B * T;
  if (hipMemsetAsync(dcs[0], 0, (size_t)B * U * sizeof(float), st) != hipSuccess ||
      hipMemsetAsync(dhc[0], 0, (size_t)B * U * sizeof(float), st) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_att_decoder_bwd: memset");
  // the bf16 images of W_cell are written again here (one launch per loop): the call does not depend on the forward
  // loop having run with the same work space
  const bool cell_h = dec_cell_gemm_h(a);
  if (cell_h) DEC_TRY(asr_lstm_cell_gemm_prep_h(h, a->W_cell, a->b_cell, Din, U, a->W_cell_h, s));
  int cur = 0;
  const float* dalpha_next = nullptr;
  for (int k = To - 1; k >= 0; --k) {
    // d loss / d ctx_k = attentional-vector part + the context columns of step k+1's cell-input gradient; the sum is
    // formed inside the d-alpha kernel (and kept in dctx_all for the d_enc contraction after the loop)
    float* dctx = a->dctx_all + (size_t)k * B * E2;
    const float* dav_ctx = a->dav_ctx + (size_t)k * B * E2;
    const float* d_in_next = (k + 1 < To) ? a->d_in_all + (size_t)(k + 1) * B * Din : nullptr;
    const float* alpha_k = a->alpha_all + (size_t)k * B * T;
    const float* qz = a->qz_all + (size_t)k * B * A;
    float* dqz = a->dqz_all + (size_t)k * B * A;
    float* dv = a->dv_all ? a->dv_all + (size_t)k * B * A : nullptr;
    // the softmax backward is folded into the energy backward when both sides run their vectorised kernels
    SoftmaxBwdFold fold = {nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    SoftmaxBwdFold* fp = (!a->carry_alpha && energy_vec_shape(A, a->keys, qz, a->v, a->dkeys)) ? &fold : nullptr;
    const float* snorm = a->snorm_all ? a->snorm_all + (size_t)k * B : nullptr;
    if (d_in_next) {
      DEC_TRY(softmax_ctx_bwd_launch(h, dav_ctx, d_in_next + Em, Din, dctx, alpha_k, a->seq_len, a->sharpening, a->enc,
                                     a->enc_dtype, T, B, E2, denergy, nullptr, snorm, dalpha_next, fp, s));
    } else {
      if (hipMemcpyAsync(dctx, dav_ctx, (size_t)B * E2 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        ASR_FAIL(h, ASR_ERR_HIP, "asr_att_decoder_bwd: copy");
      DEC_TRY(softmax_ctx_bwd_launch(h, dctx, nullptr, 0, nullptr, alpha_k, a->seq_len, a->sharpening, a->enc,
                                     a->enc_dtype, T, B, E2, denergy, nullptr, snorm, dalpha_next, fp, s));
    }
    if (a->carry_alpha) {
      float* dap = dalp[k & 1];
      DEC_TRY(loc_energy_bwd_launch(h, denergy, k > 0 ? a->alpha_all + (size_t)(k - 1) * B * T : a->alpha_zero, a->filt,
                                    a->wfil, a->keys, qz, a->v, T, B, A, a->taps, a->dkeys, dqz, dv, a->dwfil_rows,
                                    a->dfilt_rows, dap, k != To - 1, a->seq_len, s));
      dalpha_next = dap;
    }
    float* dcell = const_cast<float*>(a->dav_cell) + (size_t)k * B * U;
    float* dpre = a->dpre_all + (size_t)k * B * 4 * U;
    float* dpeep = a->dpeep_all ? a->dpeep_all + (size_t)k * B * 3 * U : nullptr;
    const float* dh2 = d_in_next ? d_in_next + Em + E2 : nullptr;
    const float* dmask = a->dmask ? a->dmask + (size_t)k * B * U : nullptr;
    // query-FC backward + cell backward in ONE launch that also sums the energy backward's chunk partials, when the
    // shapes allow (a query FC over A = 64 / 128 / 256 / 512 columns, U % 16 == 0, 16-byte aligned rows, no carried alpha)
    const bool fused_q = !a->carry_alpha && a->has_query_fc && (A == 64 || A == 128 || A == 256 || A == 512) && U % DQ_CT == 0 &&
                         a->ld_wq % 4 == 0 && ((uintptr_t)a->W_q) % 16 == 0 && ((uintptr_t)dqz) % 16 == 0 &&
                         (!dv || ((uintptr_t)dv) % 16 == 0);
    if (fused_q) {
      const float* part = nullptr;
      int nchq = 0;
      DEC_TRY(energy_bwd_launch(h, denergy, a->keys, qz, a->v, T, B, A, a->att_mode, a->dkeys, dqz, dv, a->seq_len, fp, s,
                                &part, &nchq));
      CellBwdArgs ca = {dcs[cur], dhc[cur], dh2, a->gates_all + (size_t)k * B * 4 * U, a->craw_all + (size_t)k * B * U,
                        a->c_all + (size_t)k * B * U, a->peep, a->live + (size_t)k * B, dmask,
                        dpre, dcs[cur ^ 1], dhc[cur ^ 1], dpeep, Din, 0.f};   // (clip: see the unfused call below)
      hipLaunchKernelGGL(att_dq_cell_bwd_kernel, dim3(U / DQ_CT + 2, (B + 15) / 16), dim3(512), 0, st, part, nchq, B, A, U,
                         a->W_q, a->ld_wq, dcell, dqz, dv, ca);
      ASR_CHECK_LAUNCH(h, "asr_att_decoder_bwd");
    } else {
      if (!a->carry_alpha)
        DEC_TRY(energy_bwd_launch(h, denergy, a->keys, qz, a->v, T, B, A, a->att_mode, a->dkeys, dqz, dv, a->seq_len, fp, s));
      if (a->has_query_fc)
        DEC_TRY(asr_gemm_act(h, ASR_F32, ASR_F32, 0, 1, B, U, A, dqz, A, a->W_q, a->ld_wq, dcell, U, nullptr, 1, 0, s));
      else
        DEC_TRY(asr_add_cols(h, dcell, U, dqz, A, dcell, U, B, U, s));
      // the cell kernel applies the output dropout mask to dcell and adds the h-columns of step k+1's cell-input
      // gradient to the carried dh itself
      DEC_TRY(cell_bwd_launch(h, dcell, dcs[cur], dhc[cur], a->gates_all + (size_t)k * B * 4 * U,
                              a->craw_all + (size_t)k * B * U, a->c_all + (size_t)k * B * U, a->peep,
                              a->live + (size_t)k * B, B, U, dpre, dcs[cur ^ 1], dhc[cur ^ 1], dpeep, dh2, Din, dmask,
                              0.f, s));   // the decoder cell is tf.contrib.rnn.LSTMBlockCell (attention_seq2seq.py:354-365),
                                          // whose gradient op ignores cell_clip: the clamp is straight-through here
    }
    float* d_in = a->d_in_all + (size_t)k * B * Din;
    if (cell_h) DEC_TRY(asr_lstm_cell_gemm_bwd_h(h, dpre, B, Din, U, a->W_cell_h, d_in, Din, s));   // bf16 weight rows
    else DEC_TRY(asr_gemm_act(h, ASR_F32, ASR_F32, 0, 1, B, Din, 4 * U, dpre, 4 * U, a->W_cell, 4 * U, d_in, Din, nullptr, 0, 0, s));
    cur ^= 1;
  }
  DEC_TRY(asr_add_cols(h, dhc[cur], U, a->d_in_all + Em + E2, Din, a->dh0, U, B, U, s));
  if (hipMemcpyAsync(a->dc0, dcs[cur], (size_t)B * U * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
    ASR_FAIL(h, ASR_ERR_HIP, "asr_att_decoder_bwd: copy");
  return ASR_OK;
}

