// CTC loss (alpha/beta DP + gradient), greedy decode, row softmax for gfx950.
//
// Replaces tf.nn.ctc_loss (CPU-only kernel in TF1 -> a device->host->device hop every
// step in the reference: models/ctc/ctc.py:289-297, joint_ctc_attention.py:308-316),
// tf.nn.ctc_greedy_decoder (ctc.py:341-342 == models/ctc/decoders/greedy_decoder.py:19-50)
// and tf.nn.softmax in CTC.posteriors (ctc.py:354-380).  blank = C-1.
//
// These are HBM/latency-bound passes over logits[T,B,C]; nothing here is GEMM-shaped:
//   1. row_lse:      one wave per (t,b) row, ln sum exp           (1 read of logits)
//   2. alpha_beta:   one workgroup per utterance, 4 waves run the alpha recursion while 4
//                    run beta; previous row in LDS (double buffered), one barrier per frame,
//                    the emission ln y_t(l'_s) of the NEXT frame is fetched before the barrier
//   3. grad:         one wave per (t,b): posterior occupations gamma_t(s) =
//                    exp(alpha+beta-ln y-ln p) are summed per class in a fixed order
//                    (blank by a wave tree reduction, labels by rank rounds) so the result
//                    is deterministic; writes softmax - occupation   (1 read + 1 write)
#include "common.h"
#include <math.h>

namespace {

constexpr float NEG_INF = -INFINITY;

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}
// The alpha/beta recursions run in fp64: for T of several hundred frames the log-domain values
// reach -1e3 and fp32 (what TF's CPU kernel uses) leaves ~1e-3 absolute error in the
// posteriors.  The DP is latency-bound, MI355X has full-rate fp64 VALU, so this costs little.
constexpr double DNEG_INF = -INFINITY;
__device__ __forceinline__ double dlse2(double a, double b) {
  const double m = fmax(a, b);
  if (m == DNEG_INF) return DNEG_INF;
  return m + log(exp(a - m) + exp(b - m));
}
__device__ __forceinline__ double dlse3(double a, double b, double c) {
  const double m = fmax(fmax(a, b), c);
  if (m == DNEG_INF) return DNEG_INF;
  return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}

// Recursion form: the running maximum and the result stay fp64; only the three differences
// (<= 0) go through the fp32 v_exp / v_log units.  Per-frame error ~1e-7 absolute in the log
// domain (vs 6e-5 when alpha itself is fp32), ~30 instructions instead of ~400 of fp64 libm.
__device__ __forceinline__ double dlse3_mixed(double a, double b, double c) {
  const double m = fmax(fmax(a, b), c);
  if (m == DNEG_INF) return DNEG_INF;
  const float e = __expf((float)(a - m)) + __expf((float)(b - m)) + __expf((float)(c - m));
  return m + (double)__logf(e);
}

// ---- 1. row log-sum-exp ---------------------------------------------------
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ x, int rows, int C,
                                                      double* __restrict__ lse) {
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
  if (lane == 0) lse[row] = (double)m + log(s);
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
// workspace per utterance: alpha[T][SW], beta[T][SW] (SW = 2*Lmax+1), rank[Lmax], ll
constexpr int AB_THREADS = 512;  // waves 0-3 alpha, 4-7 beta
constexpr int AB_HALF = 256;
constexpr int MAX_NS = 8;        // S <= 2048

template <int NS>
__global__ __launch_bounds__(AB_THREADS) void ctc_alpha_beta_kernel(
    const float* __restrict__ logits, const double* __restrict__ lse, int T, int B, int C,
    const int32_t* __restrict__ labels_flat, const int32_t* __restrict__ label_offsets,
    const int32_t* __restrict__ seq_len, int SW, double* __restrict__ alpha_ws,
    double* __restrict__ beta_ws, int32_t* __restrict__ rank_ws, double* __restrict__ ll_ws,
    float* __restrict__ loss, int32_t* __restrict__ num_infeasible) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* rowbuf = reinterpret_cast<double*>(smem);  // [2 (alpha/beta)][2 (ping/pong)][SW]
  const int b = blockIdx.x;
  const int blank = C - 1;
  const int lo = label_offsets[b];
  const int L = label_offsets[b + 1] - lo;
  const int S = 2 * L + 1;
  const int Tb = min(seq_len[b], T);
  const int half = threadIdx.x / AB_HALF;  // 0 alpha, 1 beta
  const int tid = threadIdx.x % AB_HALF;
  const int32_t* lab = labels_flat + lo;
  double* ws = (half == 0 ? alpha_ws : beta_ws) + (size_t)b * T * SW;
  double* buf = rowbuf + half * 2 * SW;

  // rank of each label among equal earlier labels (fixed summation order in the grad kernel) and the position of
  // the first label of its class (where the grad kernel accumulates that class), packed rank | first << 16
  for (int i = threadIdx.x; i < L; i += AB_THREADS) {
    const int li = lab[i];
    int r = 0, first = i;
    for (int k = i - 1; k >= 0; --k)
      if (lab[k] == li) { ++r; first = k; }
    rank_ws[(size_t)b * ((SW - 1) / 2) + i] = r | (first << 16);
  }
  if (Tb <= 0 || L > (SW - 1) / 2) {
    if (threadIdx.x == 0) {
      loss[b] = 0.f;
      ll_ws[b] = DNEG_INF;
      if (num_infeasible && Tb > 0) atomicAdd(num_infeasible, 1);
    }
    return;
  }

  // per-thread extended-label info for s = tid + n*256
  int ext[NS];
  bool skp[NS];  // alpha: may come from s-2 ; beta: may go to s+2
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const int s = tid + n * AB_HALF;
    ext[n] = blank; skp[n] = false;
    if (s < S) {
      ext[n] = (s & 1) ? lab[s >> 1] : blank;
      if (half == 0) skp[n] = (s & 1) && s >= 2 && lab[s >> 1] != lab[(s >> 1) - 1];
      else skp[n] = (s & 1) && s + 2 < S && lab[s >> 1] != lab[(s >> 1) + 1];
    }
  }
  const int t0 = half == 0 ? 0 : Tb - 1;
  const int dt = half == 0 ? 1 : -1;
  // emissions of the first frame
  double lp[NS];
  {
    const float* row = logits + ((size_t)t0 * B + b) * C;
    const double z = lse[(size_t)t0 * B + b];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int s = tid + n * AB_HALF;
      lp[n] = (s < S) ? (double)row[ext[n]] - z : DNEG_INF;
    }
  }
  // init row
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const int s = tid + n * AB_HALF;
    if (s < S) {
      double v = DNEG_INF;
      if (half == 0) { if (s <= 1) v = lp[n]; }
      else { if (s >= S - 2) v = lp[n]; }
      buf[s] = v;
      ws[(size_t)t0 * SW + s] = v;
    }
  }
  __syncthreads();
  // emissions of the next frame are requested BEFORE this frame's recursion and converted after it
  const size_t rstride = (size_t)B * C;
  const float* row = logits + ((size_t)t0 * B + b) * C;
  const double* zp = lse + (size_t)t0 * B + b;
  for (int step = 1; step < Tb; ++step) {
    const int t = t0 + dt * step;
    const double* prev = buf + ((step - 1) & 1) * SW;
    double* cur = buf + (step & 1) * SW;
    // lp currently holds frame t's emissions (loaded one iteration ahead, or below for step 1)
    if (step == 1) {
      row += dt * (ptrdiff_t)rstride; zp += dt * (ptrdiff_t)B;
      const double z = *zp;
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        const int s = tid + n * AB_HALF;
        if (s < S) lp[n] = (double)row[ext[n]] - z;
      }
    }
    float nf[NS];
    double nz = 0.0;
    const bool more = step + 1 < Tb;
    if (more) {
      row += dt * (ptrdiff_t)rstride; zp += dt * (ptrdiff_t)B;
      nz = *zp;
#pragma unroll
      for (int n = 0; n < NS; ++n) {
        const int s = tid + n * AB_HALF;
        nf[n] = (s < S) ? row[ext[n]] : 0.f;
      }
    }
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int s = tid + n * AB_HALF;
      if (s < S) {
        double a0 = prev[s], a1, a2;
        if (half == 0) {
          a1 = s >= 1 ? prev[s - 1] : DNEG_INF;
          a2 = skp[n] ? prev[s - 2] : DNEG_INF;
        } else {
          a1 = s + 1 < S ? prev[s + 1] : DNEG_INF;
          a2 = skp[n] ? prev[s + 2] : DNEG_INF;
        }
        const double v = dlse3_mixed(a0, a1, a2) + lp[n];
        cur[s] = v;
        ws[(size_t)t * SW + s] = v;
      }
    }
    if (more) {
#pragma unroll
      for (int n = 0; n < NS; ++n) lp[n] = (double)nf[n] - nz;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // alpha half, thread 0
    const double* last = buf + ((Tb - 1) & 1) * SW;
    const double ll = dlse2(last[S - 1], S >= 2 ? last[S - 2] : DNEG_INF);
    const bool ok = ll > DNEG_INF;
    loss[b] = ok ? (float)(-ll) : 0.f;
    ll_ws[b] = ok ? ll : DNEG_INF;
    if (!ok && num_infeasible) atomicAdd(num_infeasible, 1);
  }
}

// ---- 3. gradient -------------------------------------------------------------
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logits, const double* __restrict__ lse, int T, int B, int C,
    const int32_t* __restrict__ labels_flat, const int32_t* __restrict__ label_offsets,
    const int32_t* __restrict__ seq_len, int SW, const double* __restrict__ alpha_ws,
    const double* __restrict__ beta_ws, const int32_t* __restrict__ rank_ws,
    const double* __restrict__ ll_ws, float grad_scale, float* __restrict__ grad) {
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
  const int32_t* rk = rank_ws + (size_t)b * LW;
  const float* row = logits + ((size_t)t * B + b) * C;
  const double z = lse[(size_t)t * B + b];
  const double* al = alpha_ws + ((size_t)b * T + t) * SW;
  const double* be = beta_ws + ((size_t)b * T + t) * SW;

  for (int k = lane; k < BW; k += 64) bits[k] = 0u;
  for (int i = lane; i < L; i += 64) acc[i] = 0.f;
  float blank_sum = 0.f;
  int maxrank = 0;
  for (int s = lane; s < S; s += 64) {
    const int e = (s & 1) ? lab[s >> 1] : blank;
    const double lpv = (double)row[e] - z;
    const double v = al[s] + be[s] - lpv - ll;   // ln gamma_t(s); -inf/NaN-safe below
    const float gm = (al[s] > DNEG_INF && be[s] > DNEG_INF) ? (float)exp(v) : 0.f;
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
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, int T,
                                                         int B, int C,
                                                         const int32_t* __restrict__ seq_len,
                                                         int blank, int32_t* __restrict__ out_labels,
                                                         int32_t* __restrict__ out_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* am = reinterpret_cast<int32_t*>(smem);  // [T] per-frame argmax
  __shared__ int32_t wsum[4];
  __shared__ int32_t carry;
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Tb = min(max(seq_len[b], 0), T);
  for (int t = wave; t < Tb; t += 4) {
    const float* row = logits + ((size_t)t * B + b) * C;
    float best = NEG_INF;
    int bi = 0x7fffffff;
    for (int k = lane; k < C; k += 64) {
      const float v = row[k];
      if (v > best || (v == best && k < bi)) { best = v; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) am[t] = (bi == 0x7fffffff) ? 0 : bi;
  }
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

struct CtcWs {
  size_t lse, alpha, beta, rank, ll, total;
};
inline CtcWs ctc_ws_layout(int T, int B, int Lmax) {
  const size_t SW = 2 * (size_t)Lmax + 1;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  CtcWs w;
  size_t o = 0;
  w.lse = o;   o += al((size_t)T * B * 8);
  w.alpha = o; o += al((size_t)B * T * SW * 8);
  w.beta = o;  o += al((size_t)B * T * SW * 8);
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
  if (SW > MAX_NS * AB_HALF)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_loss: label length %d > %d", max_label_len, (MAX_NS * AB_HALF - 1) / 2);
  const CtcWs w = ctc_ws_layout(T, B, max_label_len);
  if (!workspace || workspace_bytes < w.total)
    ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_ctc_loss: workspace %zu < %zu bytes", workspace_bytes, w.total);
  char* ws = (char*)workspace;
  double* lse = (double*)(ws + w.lse);
  double* alpha = (double*)(ws + w.alpha);
  double* beta = (double*)(ws + w.beta);
  int32_t* rank = (int32_t*)(ws + w.rank);
  double* ll = (double*)(ws + w.ll);
  hipStream_t st = (hipStream_t)s;
  if (num_infeasible) (void)hipMemsetAsync(num_infeasible, 0, sizeof(int32_t), st);
  const int rows = T * B;
  hipLaunchKernelGGL(row_lse_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, logits, rows, C, lse);
  ASR_CHECK_LAUNCH(h, "asr_ctc_loss(row_lse)");
  const size_t lds_ab = (size_t)4 * SW * sizeof(double);
  {
    const int ns = (SW + AB_HALF - 1) / AB_HALF;
#define ASR_AB(NSV)                                                                                    \
  hipLaunchKernelGGL(ctc_alpha_beta_kernel<NSV>, dim3(B), dim3(AB_THREADS), lds_ab, st, logits, lse, T, B, \
                     C, labels_flat, label_offsets, seq_len, SW, alpha, beta, rank, ll, loss, num_infeasible)
    if (ns <= 1) ASR_AB(1);
    else if (ns <= 2) ASR_AB(2);
    else if (ns <= 4) ASR_AB(4);
    else ASR_AB(8);
#undef ASR_AB
  }
  ASR_CHECK_LAUNCH(h, "asr_ctc_loss(alpha_beta)");
  if (grad) {
    const size_t lds_g = (size_t)4 * (SW + (SW - 1) / 2 + (C + 31) / 32) * sizeof(float);
    if (lds_g > 160 * 1024)
      ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_loss: C=%d, Lmax=%d need %zu B of LDS", C, max_label_len, lds_g);
    (void)hipFuncSetAttribute((const void*)ctc_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((T + 3) / 4, B), dim3(256), lds_g, st, logits, lse, T, B, C,
                       labels_flat, label_offsets, seq_len, SW, alpha, beta, rank, ll, grad_scale, grad);
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
  (void)hipFuncSetAttribute((const void*)ctc_greedy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(256), lds, (hipStream_t)s, logits, T, B, C, seq_len,
                     blank, out_labels, out_len);
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
