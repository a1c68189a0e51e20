// Handle lifetime + the HBM-bound elementwise passes of the training step:
// layout change, casts, dropout masks, bias (column) sums, per-variable clip_by_norm,
// weight decay and the seven TF1 optimizers of models/model_base.py:12-20.
// All are grid-stride, 16 B per lane where the layout allows (cdna guide G13).
#include "common.h"
#include <math.h>

// 2 (round 4): asr_create_ex's minimum scratch is 96 MiB (was 32), asr_ctc_beam_workspace_bytes asks for W more doubles per
// utterance, new entry points asr_att_decoder_infer / asr_lstm_cell_bwd_ex; nothing was removed or re-typed.
// 3 (round 4): asr_lstm_cell_gemm_prep / _fwd and their _h forms (decoder cell product + cell in one launch).
// 4 (rounds 5-6, additive): asr_conv3x3_bwd_weight_bias, asr_conv3x3_smallc_bwd_weight_bias (bias gradient out of the
// weight-gradient kernels), the asr_debug_* hooks; nothing removed or re-typed.
extern "C" int asr_abi_version(void) { return 5; }

extern "C" int asr_create(asr_handle** out, int device) { return asr_create_ex(out, device, (size_t)192 << 20); }
extern "C" size_t asr_scratch_bytes(asr_handle* h) { return h ? h->scratch_bytes : 0; }

extern "C" int asr_create_ex(asr_handle** out, int device, size_t scratch_bytes) {
  if (!out || scratch_bytes < ((size_t)96 << 20)) return ASR_ERR_INVALID_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return ASR_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return ASR_ERR_HIP;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return ASR_ERR_HIP;
  asr_handle* h = new asr_handle();
  h->device = device;
  h->num_cu = p.multiProcessorCount;
  snprintf(h->name, sizeof(h->name), "%s (%s)", p.name, p.gcnArchName);
  h->err[0] = 0;
  h->scratch = nullptr;
  h->scratch_bytes = scratch_bytes;
  h->xch_dirty[0] = h->xch_dirty[1] = 0;
  h->xch_next = 0;
  h->xcd_skip = 0;
  h->tn_wgs = 0;
  if (hipMalloc(&h->scratch, h->scratch_bytes) != hipSuccess) {
    delete h;
    return ASR_ERR_HIP;
  }
  // ASR_POISON_SCRATCH=1 (debug / test aid): the work part of the arena starts as NaN bit patterns (0xFF bytes) instead of
  // whatever the device memory held -- on a fresh box that is usually zeros, which hides a kernel that reads a partial
  // nobody wrote (a 1e-3 error once in a few cold starts instead of a NaN every time)
  if (const char* e = getenv("ASR_POISON_SCRATCH"))
    if (e[0] == '1') (void)hipMemset(h->scratch, 0xFF, h->scratch_bytes - ASR_XCH_BYTES);
  (void)hipMemset((char*)h->scratch + h->scratch_bytes - ASR_XCH_BYTES, 0, ASR_XCH_BYTES);
  *out = h;
  return ASR_OK;
}
extern "C" int asr_destroy(asr_handle* h) {
  if (h && h->scratch) (void)hipFree(h->scratch);
  delete h;
  return ASR_OK;
}
extern "C" int asr_set_xcd_skip(asr_handle* h, int n) {
  if (!h || n < 0 || n > 6) return ASR_ERR_INVALID_ARG;
  h->xcd_skip = n;
  return ASR_OK;
}
extern "C" int asr_set_gemm_tn_workgroups(asr_handle* h, int n) {
  if (!h || n < 0) return ASR_ERR_INVALID_ARG;
  h->tn_wgs = n;
  return ASR_OK;
}
extern "C" const char* asr_last_error_string(asr_handle* h) { return h ? h->err : "null handle"; }
extern "C" int asr_device_info(asr_handle* h, int* num_cu, char* name, int name_len) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (num_cu) *num_cu = h->num_cu;
  if (name && name_len > 0) snprintf(name, name_len, "%s", h->name);
  return ASR_OK;
}

// ---- placement probe ----------------------------------------------------------------------------
// Records where the workgroups of a grid run ({XCC id, HW_ID} per block).  Used to establish that (i) block b of a
// 1-D grid lands on XCD b % 8 (what the cluster kernels' layout relies on for speed, never for correctness) and
// (ii) queue CU masks (hipExtStreamCreateWithCUMask) are NOT honoured on this stack: a masked stream's grid still
// spreads over all 8 XCDs / 256 CUs (profiles/r02_cumask_probe.json), so side work cannot be fenced off that way.
namespace {
__global__ void placement_kernel(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;      // HW_REG_XCC_ID[3:0]
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);         // HW_REG_HW_ID
  }
  // keep the workgroup resident for a while so that a grid spreads over CUs instead of reusing the first one
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
}
}  // namespace
extern "C" int asr_debug_placement(asr_handle* h, unsigned* out, int nblocks, int spin_cycles, asr_stream s) {
  if (!h || !out || nblocks < 1) return ASR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(placement_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)s, out, spin_cycles);
  ASR_CHECK_LAUNCH(h, "placement_kernel");
  return ASR_OK;
}

namespace {

inline int grid_for(size_t n, int per_thread = 1) {
  size_t blocks = (n + 256 * (size_t)per_thread - 1) / (256 * (size_t)per_thread);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  return (int)blocks;
}

// [B,T,D] -> [T,B,D]: one row of D per wave-slice, rows of the OUTPUT enumerated in order
template <typename T>
__global__ void bt_to_tb_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int T_, int D, int ld) {
  const size_t total = (size_t)B * T_ * ld;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int dd = i % ld;
    const size_t r = i / ld;
    const int b = r % B;
    const size_t t = r / B;
    out[i] = dd < D ? Elem<T>::from_f32(in[((size_t)b * T_ + t) * D + dd]) : Elem<T>::from_f32(0.f);
  }
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = Elem<T>::from_f32(in[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = Elem<T>::to_f32(in[i]);
}
template <typename T>
__global__ void apply_mask_kernel(const T* __restrict__ in, const float* __restrict__ mask,
                                  T* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = Elem<T>::from_f32(Elem<T>::to_f32(in[i]) * mask[i]);
}

// Philox4x32-10 (Salmon et al. 2011): counter-based, one call gives 4 uniforms
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
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
__global__ void dropout_mask_kernel(float* __restrict__ mask, size_t n, float keep, uint64_t seed,
                                    uint64_t offset) {
  const float inv = 1.f / keep;
  const size_t n4 = (n + 3) / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t ctr = offset + i;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t e = i * 4 + j;
      if (e < n) {
        const float u = (c[j] >> 8) * (1.0f / 16777216.0f);  // [0,1)
        mask[e] = (u < keep) ? inv : 0.f;
      }
    }
  }
}

// The mask of dropout_mask_kernel formed where it is used instead of being stored: element e takes word e % 4 of
// the Philox block at counter offset + e / 4 -- bit-identical to asr_dropout_mask followed by asr_apply_mask.  For the
// VGG front-end, whose activation tensors are gigabytes (cfg C: 1.6 G elements after the first block): an fp32 mask
// written once and read twice is 12 bytes per element against 2-4 for the activation itself.
__device__ __forceinline__ void dropout_words(uint64_t ctr, uint64_t seed, float keep, float inv, float m[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = ((c[j] >> 8) * (1.0f / 16777216.0f) < keep) ? inv : 0.f;
}
// out = in * mask;  relu_out != NULL: out = (relu_out > 0 ? in * mask : 0)  (in fp32 gradient, out in TO).
// Four elements per thread and trip as ONE vector load / store per array (8 bytes of bf16, 16 of fp32): with 2-byte
// scalar accesses the kernel ran at 2.5 TB/s.
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  typedef f32x4_t raw_t;
  static __device__ __forceinline__ void load(const float* p, float v[4]) {
    const f32x4_t r = *reinterpret_cast<const f32x4_t*>(p);
    v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
  }
  static __device__ __forceinline__ void store(float* p, const float v[4]) {
    *reinterpret_cast<f32x4_t*>(p) = (f32x4_t){v[0], v[1], v[2], v[3]};
  }
};
template <> struct Vec4<bf16_t> {
  typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
  static __device__ __forceinline__ void load(const bf16_t* p, float v[4]) {
    const us4_t r = *reinterpret_cast<const us4_t*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = bf16_to_f32(r[j]);
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float v[4]) {
    us4_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = f32_to_bf16(v[j]);
    *reinterpret_cast<us4_t*>(p) = r;
  }
};
template <typename TI, typename TO>
__global__ void dropout_apply_kernel(const TI* __restrict__ in, const TO* __restrict__ relu_out, TO* __restrict__ out,
                                     size_t n, float keep, uint64_t seed, uint64_t offset, int vec_ok) {
  const float inv = 1.f / keep;
  const size_t n4 = (n + 3) / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float m[4];
    dropout_words(offset + i, seed, keep, inv, m);
    const size_t e0 = i * 4;
    if (vec_ok && e0 + 4 <= n) {
      float x[4], r[4];
      Vec4<TI>::load(in + e0, x);
      if (relu_out) Vec4<TO>::load(relu_out + e0, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] *= m[j];
        if (relu_out) x[j] = r[j] > 0.f ? x[j] : 0.f;
      }
      Vec4<TO>::store(out + e0, x);
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t e = e0 + j;
      if (e < n) {
        float g = Elem<TI>::to_f32(in[e]) * m[j];
        if (relu_out) g = Elem<TO>::to_f32(relu_out[e]) > 0.f ? g : 0.f;
        out[e] = Elem<TO>::from_f32(g);
      }
    }
  }
}

// Read pass over a buffer (16-byte loads, nothing written): pulls it into the memory-side cache.  The BPTT kernels fetch
// their saved activations one step ahead, which covers the cache's latency but not HBM's -- a layer's BPTT launch takes
// 1.15 ms with gates / cell states resident and 1.29 - 1.36 ms without (scripts/probe_bptt_cache.py) -- so the layer
// below is touched while the dx product of the layer above runs (the product is bound by its writes).
__global__ void touch_kernel(const f32x4_t* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4_t v = p[i];
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-33f) *sink = 1u;      // never true in practice: keeps the loads alive
}

// column sums, two deterministic stages: grid (col blocks, row blocks) -> partial[rb][N] -> out[N]
constexpr int COLSUM_ROWS = 512;
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ a, int M, int N, int lda,
                                                     float* __restrict__ partial) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rgp = threadIdx.x >> 6;
  const int r0 = blockIdx.y * COLSUM_ROWS, r1 = min(M, r0 + COLSUM_ROWS);
  float s = 0.f;
  if (col < N) {
#pragma unroll 8                                           // eight rows requested ahead of the (ordered) additions
    for (int m = r0 + rgp; m < r1; m += 4) s += Elem<T>::to_f32(a[(size_t)m * lda + col]);
  }
  part[rgp][threadIdx.x & 63] = s;
  __syncthreads();
  if (rgp == 0 && col < N)
    partial[(size_t)blockIdx.y * N + col] =
        part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
// bf16, contiguous rows of N = 8 * NV columns (the pre-activation gradients of the VGG stack: [26 M pixels x 64] ...
// [7 M x 128], 2 - 3 GB each): a thread owns EIGHT columns (one 16-byte load per row) of every (256 / NV)-th row of the
// block's row range, so a wave's load covers whole rows and 1 KB instead of 128 bytes.  Sums in a fixed order:
// thread-serial over its rows, then over the block's row-threads through LDS.  partial[block][N].
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const bf16_t* __restrict__ a, long long M, int N,
                                                              long long rows_per_block, float* __restrict__ partial) {
  extern __shared__ float cred[];                          // [256 / NV][N]
  const int NV = N / 8, RT = 256 / NV;                     // column groups, row-threads
  const int cg = threadIdx.x % NV, rt = threadIdx.x / NV;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rt < RT) {
    for (long long m = r0 + rt; m < r1; m += RT) {
      const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(a + (size_t)m * N + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += bf16_to_f32((bf16_t)v[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) cred[rt * N + cg * 8 + i] = s[i];
  }
  __syncthreads();
  for (int col = threadIdx.x; col < N; col += 256) {
    float t = 0.f;
    for (int r = 0; r < RT; ++r) t += cred[r * N + col];
    partial[(size_t)blockIdx.x * N + col] = t;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int RB, int N, float* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= N) return;
  float s = 0.f;
#pragma unroll 8
  for (int r = 0; r < RB; ++r) s += partial[(size_t)r * N + col];
  out[col] = s;
}

// The same final pass for MANY partial rows (the vector kernel leaves up to 2048): one thread per column walks them with
// a handful of loads in flight -- 335 us for 2048 rows of 64 columns (`profiles/r03_cfgC_kernel_trace.md`, six calls
// per cfg C step).  Here 16 columns x 16 row groups per workgroup: a thread sums every 16th row in row order, the 16
// group sums are added in group order (fixed order: deterministic).
__global__ __launch_bounds__(256) void colsum_final_wide_kernel(const float* __restrict__ partial, int RB, int N,
                                                                float* __restrict__ out) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + c;
  float s = 0.f;
  if (col < N)
    for (int r = grp; r < RB; r += 16) s += partial[(size_t)r * N + col];
  red[grp][c] = s;
  __syncthreads();
  if (grp == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    out[col] = t;
  }
}

// ---- per-tensor L2 norms (two-stage, deterministic) + clip ----
constexpr int NORM_CHUNK = 4096;  // elements per partial
__global__ __launch_bounds__(256) void sqsum_partial_kernel(const float* __restrict__ g,
                                                            const int64_t* __restrict__ offsets,
                                                            const int64_t* __restrict__ chunk_start,
                                                            int num_tensors, float* __restrict__ partial) {
  // blockIdx.x = global chunk id; find its tensor by binary search on chunk_start
  __shared__ float red[4];
  const int64_t cid = blockIdx.x;
  int lo = 0, hi = num_tensors - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunk_start[mid] <= cid) lo = mid; else hi = mid - 1;
  }
  const int64_t beg = offsets[lo] + (cid - chunk_start[lo]) * NORM_CHUNK;
  const int64_t end = min(beg + (int64_t)NORM_CHUNK, offsets[lo + 1]);
  float s = 0.f;
#pragma unroll 8                                           // (NORM_CHUNK / 256 = 16 elements per thread: the loads ahead of the ordered sum)
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) { const float v = g[i]; s += v * v; }
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[cid] = red[0] + red[1] + red[2] + red[3];
}
__global__ void clip_scale_kernel(float* __restrict__ g, const int64_t* __restrict__ offsets,
                                  const int64_t* __restrict__ chunk_start, int num_tensors,
                                  const float* __restrict__ partial, float clip) {
  const int64_t cid = blockIdx.x;
  int lo = 0, hi = num_tensors - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunk_start[mid] <= cid) lo = mid; else hi = mid - 1;
  }
  // every block of a tensor recomputes the same ordered sum of that tensor's partials -- staged through LDS 1024 at a time
  // (round 6: as a loop over global loads the 768 partials of a 5 x 512 layer's kernel gradient were a 224 us chain in
  // every one of its blocks; same additions in the same order)
  __shared__ float sp[1024];
  float ss = 0.f;
  const int64_t cend = chunk_start[lo + 1];
  for (int64_t c0 = chunk_start[lo]; c0 < cend; c0 += 1024) {
    const int n = (int)min((int64_t)1024, cend - c0);
    for (int i = threadIdx.x; i < n; i += blockDim.x) sp[i] = partial[c0 + i];
    __syncthreads();
    for (int i = 0; i < n; ++i) ss += sp[i];
    __syncthreads();
  }
  const float norm = sqrtf(ss);
  const float scale = clip / fmaxf(norm, clip);  // tf.clip_by_norm: t * clip / max(||t||, clip)
  if (scale == 1.f) return;
  const int64_t beg = offsets[lo] + (cid - chunk_start[lo]) * NORM_CHUNK;
  const int64_t end = min(beg + (int64_t)NORM_CHUNK, offsets[lo + 1]);
  // 16-byte accesses on the aligned middle of the chunk (a variable starts wherever the one before it ends): the scalar
  // loop ran at 1.1 TB/s -- 224 us for the 31 M gradients of cfg D
  const int64_t a0 = min(end, (beg + 3) & ~(int64_t)3), a1 = a0 + ((end - a0) & ~(int64_t)3);
  for (int64_t i = beg + threadIdx.x; i < a0; i += blockDim.x) g[i] *= scale;
  for (int64_t i = a0 + 4 * (int64_t)threadIdx.x; i < a1; i += 4 * (int64_t)blockDim.x) {
    f32x4_t v = *reinterpret_cast<f32x4_t*>(g + i);
    v[0] *= scale; v[1] *= scale; v[2] *= scale; v[3] *= scale;
    *reinterpret_cast<f32x4_t*>(g + i) = v;
  }
  for (int64_t i = a1 + threadIdx.x; i < end; i += blockDim.x) g[i] *= scale;
}

__global__ void weight_decay_kernel(float* __restrict__ g, const float* __restrict__ p,
                                    const int64_t* __restrict__ offsets,
                                    const uint8_t* __restrict__ decay_mask, int num_tensors, float wd) {
  const int t = blockIdx.y;
  if (!decay_mask[t]) return;
  const int64_t beg = offsets[t], end = offsets[t + 1];
  for (int64_t i = beg + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < end;
       i += (int64_t)gridDim.x * blockDim.x)
    g[i] += wd * p[i];
}
__global__ __launch_bounds__(256) void l2_loss_kernel(const float* __restrict__ p,
                                                      const int64_t* __restrict__ offsets,
                                                      const uint8_t* __restrict__ decay_mask,
                                                      int num_tensors, float wd, float* __restrict__ out) {
  // single block, fixed order: sum_t mask_t * 0.5*||p_t||^2 * wd
  __shared__ float red[4];
  float s = 0.f;
  for (int t = 0; t < num_tensors; ++t) {
    if (!decay_mask[t]) continue;
    for (int64_t i = offsets[t] + threadIdx.x; i < offsets[t + 1]; i += 256) { const float v = p[i]; s += v * v; }
  }
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = 0.5f * wd * (red[0] + red[1] + red[2] + red[3]);
}

// TF1 optimizer update rules with default hyper-parameters
template <int OPT>
__global__ void optimizer_kernel(float* __restrict__ p, const float* __restrict__ g,
                                 float* __restrict__ s0, float* __restrict__ s1, size_t n, float lr,
                                 float adam_lr_t) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i];
    if (OPT == ASR_OPT_SGD) {
      pi -= lr * gi;
    } else if (OPT == ASR_OPT_MOMENTUM) {
      const float a = 0.9f * s0[i] + gi;
      s0[i] = a;
      pi -= lr * a;
    } else if (OPT == ASR_OPT_NESTEROV) {
      const float a = 0.9f * s0[i] + gi;
      s0[i] = a;
      pi -= lr * gi + lr * 0.9f * a;
    } else if (OPT == ASR_OPT_ADAGRAD) {
      const float a = s0[i] + gi * gi;  // slot initialised to 0.1 by the host
      s0[i] = a;
      pi -= lr * gi / sqrtf(a);
    } else if (OPT == ASR_OPT_ADADELTA) {
      const float rho = 0.95f, eps = 1e-8f;
      const float a = rho * s0[i] + (1.f - rho) * gi * gi;
      const float upd = sqrtf(s1[i] + eps) / sqrtf(a + eps) * gi;
      s0[i] = a;
      s1[i] = rho * s1[i] + (1.f - rho) * upd * upd;
      pi -= lr * upd;
    } else if (OPT == ASR_OPT_RMSPROP) {
      const float decay = 0.9f, eps = 1e-10f;
      const float ms = decay * s0[i] + (1.f - decay) * gi * gi;  // slot initialised to 1 by the host
      s0[i] = ms;
      const float mom = 0.f * s1[i] + lr * gi / sqrtf(ms + eps);
      s1[i] = mom;
      pi -= mom;
    } else if (OPT == ASR_OPT_ADAM) {
      const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
      const float m = b1 * s0[i] + (1.f - b1) * gi;
      const float v = b2 * s1[i] + (1.f - b2) * gi * gi;
      s0[i] = m; s1[i] = v;
      pi -= adam_lr_t * m / (sqrtf(v) + eps);
    }
    p[i] = pi;
  }
}

__global__ void scale_kernel(float* __restrict__ x, size_t n, float sc) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] *= sc;
}

}  // namespace

#define ASR_NEED(cond, ...) do { if (!(cond)) ASR_FAIL(h, ASR_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* __restrict__ in, int rows, int cols, int ld_in,
                                                          T* __restrict__ out, int ld_out) {
  __shared__ T tile[64][64 + 2];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * ld_in + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * ld_out + r0 + tx] = tile[tx][i];
}

extern "C" int asr_transpose2d(asr_handle* h, int dtype, const void* in, int rows, int cols, int ld_in,
                               void* out, int ld_out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && in && out && rows >= 0 && cols >= 0 && ld_in >= cols && ld_out >= rows,
           "asr_transpose2d: bad args");
  if (rows == 0 || cols == 0) return ASR_OK;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype == ASR_F32)
    hipLaunchKernelGGL(transpose2d_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)in, rows, cols,
                       ld_in, (float*)out, ld_out);
  else
    hipLaunchKernelGGL(transpose2d_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, rows, cols,
                       ld_in, (bf16_t*)out, ld_out);
  ASR_CHECK_LAUNCH(h, "asr_transpose2d");
  return ASR_OK;
}

namespace {
// Frame stacking / skipping on the padded batch (utils/io/inputs/frame_stacking.py:14-85): output frame k of
// utterance b holds input frames k*skip .. k*skip+stack-1 side by side; slots past the utterance's last frame
// (and whole frames past ceil(len/skip)) are zero.
__global__ void stack_frames_kernel(const float* __restrict__ x, const int32_t* __restrict__ seq_len, int B, int T,
                                    int F, int num_stack, int num_skip, int Tn, float* __restrict__ out,
                                    int32_t* __restrict__ out_len) {
  const size_t row = (size_t)F * num_stack;
  const size_t n = (size_t)B * Tn * row;
  const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (out_len && gid < (size_t)B) {
    const int len = min(max(seq_len[gid], 0), T);
    out_len[gid] = (len + num_skip - 1) / num_skip;
  }
  for (size_t i = gid; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % row);
    const size_t bk = i / row;
    const int k = (int)(bk % Tn), b = (int)(bk / Tn);
    const int len = min(max(seq_len[b], 0), T);
    const int src = k * num_skip + col / F;
    out[i] = src < len ? x[((size_t)b * T + src) * F + col % F] : 0.f;
  }
}

// Context splicing (utils/io/inputs/splicing.py:9-73, behaviour of the code, quirk Q9): per utterance, over its
// own len frames, frame t takes frames t-splice .. t-1 with the edge rules of :42-57; an input frame
// [C*3*num_stack] is read as (C, 3, num_stack) and the result is laid out [C][splice*num_stack][3].
// The reference writes slots i .. i+num_stack-1 for i = 0 .. splice-1 in order, so slot j keeps the write of
// i = min(j, splice-1) and slots j >= splice-1+num_stack stay zero.
__global__ void splice_kernel(const float* __restrict__ x, const int32_t* __restrict__ seq_len, int B, int T, int D,
                              int splice, int num_stack, float* __restrict__ out) {
  const int C = D / (3 * num_stack);
  const int SJ = splice * num_stack;
  const size_t row = (size_t)C * SJ * 3;
  const size_t n = (size_t)B * T * row;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % row);
    const size_t bt = i / row;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const int len = min(max(seq_len[b], 0), T);
    const int k3 = col % 3, j = (col / 3) % SJ, c = col / (3 * SJ);
    const int is = min(j, splice - 1), sidx = j - is;
    float v = 0.f;
    if (t < len && sidx < num_stack) {
      int src = t + is - splice;
      const bool left = t <= splice - 1 && is < splice - t;
      if (left) src = 0;
      else if (len - splice <= t && src > len - 1) src = len - 1;
      v = x[((size_t)b * T + src) * D + (size_t)c * 3 * num_stack + k3 * num_stack + sidx];
    }
    out[i] = v;
  }
}

// Same result, laid out for the write stream that dominates (the output is splice x the input): the column ->
// (window slot, offset inside the source frame) map is identical for every frame, so a workgroup builds it once
// in LDS (the only integer divisions) and then emits SPLICE_FPB frames as 16-byte stores, four gathers each.
constexpr int SPLICE_FPB = 8;
__global__ __launch_bounds__(256) void splice_rows_kernel(const float* __restrict__ x,
                                                          const int32_t* __restrict__ seq_len, int B, int T, int D,
                                                          int splice, int num_stack, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char splice_smem[];
  uint32_t* tab = reinterpret_cast<uint32_t*>(splice_smem);       // [row]: slot << 24 | offset, ~0 = stays zero
  const int C = D / (3 * num_stack);
  const int SJ = splice * num_stack;
  const int row = C * SJ * 3;
  for (int col = threadIdx.x; col < row; col += 256) {
    const int k3 = col % 3, j = (col / 3) % SJ, c = col / (3 * SJ);
    const int is = min(j, splice - 1), sidx = j - is;
    tab[col] = sidx < num_stack ? ((uint32_t)is << 24) | (uint32_t)(c * 3 * num_stack + k3 * num_stack + sidx)
                                : 0xffffffffu;
  }
  __syncthreads();
  const size_t nf = (size_t)B * T;
  const size_t f0 = (size_t)blockIdx.x * SPLICE_FPB;
  for (size_t f = f0; f < f0 + SPLICE_FPB && f < nf; ++f) {
    const int b = (int)(f / T), t = (int)(f % T);
    const int len = min(max(seq_len[b], 0), T);
    const float* xb = x + (size_t)b * T * D;
    float* o = out + f * row;
    const bool live = t < len;
    for (int c4 = threadIdx.x * 4; c4 < row; c4 += 1024) {
      const uint4 u = *reinterpret_cast<const uint4*>(tab + c4);
      const uint32_t uu[4] = {u.x, u.y, u.z, u.w};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = 0.f;
        if (live && uu[e] != 0xffffffffu) {
          const int is = (int)(uu[e] >> 24);
          int src = t + is - splice;
          if (t <= splice - 1 && is < splice - t) src = 0;
          else if (len - splice <= t && src > len - 1) src = len - 1;
          v[e] = xb[(size_t)src * D + (uu[e] & 0xffffffu)];
        }
      }
      *reinterpret_cast<float4*>(o + c4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
}  // namespace

extern "C" int asr_stack_frames(asr_handle* h, const float* x, const int32_t* seq_len, int B, int T, int F,
                                int num_stack, int num_skip, float* out, int32_t* out_len, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(x && seq_len && out && B >= 0 && T >= 0 && F > 0 && num_stack >= 1 && num_skip >= 1,
           "asr_stack_frames: bad args");
  // frame_stacking.py:30-31 raises ValueError for num_stack < num_skip
  ASR_NEED(num_stack >= num_skip, "asr_stack_frames: num_skip must not exceed num_stack");
  const int Tn = (T + num_skip - 1) / num_skip;
  const size_t n = (size_t)B * Tn * F * num_stack;
  if (!n) return ASR_OK;
  hipLaunchKernelGGL(stack_frames_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, x, seq_len, B, T, F,
                     num_stack, num_skip, Tn, out, out_len);
  ASR_CHECK_LAUNCH(h, "asr_stack_frames");
  return ASR_OK;
}

extern "C" int asr_splice(asr_handle* h, const float* x, const int32_t* seq_len, int B, int T, int D, int splice,
                          int num_stack, float* out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(x && seq_len && out && B >= 0 && T >= 0 && D > 0 && splice >= 1 && num_stack >= 1,
           "asr_splice: bad args");
  ASR_NEED(D % (3 * num_stack) == 0, "asr_splice: frame width %d is not channels*3*num_stack", D);
  const size_t n = (size_t)B * T * D * splice;
  if (!n) return ASR_OK;
  const size_t row = (size_t)D * splice;
  if (row % 4 == 0 && row * sizeof(uint32_t) <= 48 * 1024 && splice < 256 && D < (1 << 24) &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const size_t nblk = ((size_t)B * T + SPLICE_FPB - 1) / SPLICE_FPB;
    hipLaunchKernelGGL(splice_rows_kernel, dim3((unsigned)nblk), dim3(256), row * sizeof(uint32_t), (hipStream_t)s, x,
                       seq_len, B, T, D, splice, num_stack, out);
  } else {
    hipLaunchKernelGGL(splice_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, x, seq_len, B, T, D, splice,
                       num_stack, out);
  }
  ASR_CHECK_LAUNCH(h, "asr_splice");
  return ASR_OK;
}

extern "C" int asr_bt_to_tb_ld(asr_handle* h, int dtype, const float* in, void* out, int B, int T, int D, int ld_out,
                               asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && in && out && B >= 0 && T >= 0 && D >= 0 && ld_out >= D, "asr_bt_to_tb: bad args");
  const size_t n = (size_t)B * T * ld_out;
  if (!n) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL(bt_to_tb_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, in, (float*)out, B, T, D, ld_out);
  else hipLaunchKernelGGL(bt_to_tb_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, in, (bf16_t*)out, B, T, D, ld_out);
  ASR_CHECK_LAUNCH(h, "asr_bt_to_tb");
  return ASR_OK;
}
extern "C" int asr_bt_to_tb(asr_handle* h, int dtype, const float* in, void* out, int B, int T, int D, asr_stream s) {
  return asr_bt_to_tb_ld(h, dtype, in, out, B, T, D, D, s);
}
extern "C" int asr_cast_from_f32(asr_handle* h, int dtype, const float* in, void* out, size_t n, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && in && out, "asr_cast_from_f32: bad args");
  if (!n) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, in, (float*)out, n);
  else hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, in, (bf16_t*)out, n);
  ASR_CHECK_LAUNCH(h, "asr_cast_from_f32");
  return ASR_OK;
}
extern "C" int asr_cast_to_f32(asr_handle* h, int dtype, const void* in, float* out, size_t n, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && in && out, "asr_cast_to_f32: bad args");
  if (!n) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, (const float*)in, out, n);
  else hipLaunchKernelGGL(cast_to_f32_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, out, n);
  ASR_CHECK_LAUNCH(h, "asr_cast_to_f32");
  return ASR_OK;
}
extern "C" int asr_apply_mask(asr_handle* h, int dtype, const void* in, const float* mask, void* out, size_t n, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && in && mask && out, "asr_apply_mask: bad args");
  if (!n) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL(apply_mask_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, (const float*)in, mask, (float*)out, n);
  else hipLaunchKernelGGL(apply_mask_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, mask, (bf16_t*)out, n);
  ASR_CHECK_LAUNCH(h, "asr_apply_mask");
  return ASR_OK;
}
extern "C" int asr_dropout_mask(asr_handle* h, float* mask, size_t n, float keep_prob, uint64_t seed, uint64_t offset, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(mask && keep_prob > 0.f && keep_prob <= 1.f, "asr_dropout_mask: keep_prob must be in (0,1]");
  if (!n) return ASR_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)s, mask, n, keep_prob, seed, offset);
  ASR_CHECK_LAUNCH(h, "asr_dropout_mask");
  return ASR_OK;
}
extern "C" int asr_dropout_apply(asr_handle* h, int dtype, const void* in, void* out, size_t n, float keep_prob,
                                 uint64_t seed, uint64_t offset, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && in && out && keep_prob > 0.f && keep_prob <= 1.f, "asr_dropout_apply: bad args");
  if (!n) return ASR_OK;
  const dim3 grid(grid_for((n + 3) / 4));
  const int vec_ok = (((uintptr_t)in | (uintptr_t)out) % 16) == 0;
  if (dtype == ASR_F32)
    hipLaunchKernelGGL((dropout_apply_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)s, (const float*)in,
                       (const float*)nullptr, (float*)out, n, keep_prob, seed, offset, vec_ok);
  else
    hipLaunchKernelGGL((dropout_apply_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)in,
                       (const bf16_t*)nullptr, (bf16_t*)out, n, keep_prob, seed, offset, vec_ok);
  ASR_CHECK_LAUNCH(h, "asr_dropout_apply");
  return ASR_OK;
}
extern "C" int asr_relu_bwd_drop(asr_handle* h, int dtype, const float* dout, const void* out, size_t n, float keep_prob,
                                 uint64_t seed, uint64_t offset, void* dpre, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && dout && out && dpre && keep_prob > 0.f && keep_prob <= 1.f, "asr_relu_bwd_drop: bad args");
  if (!n) return ASR_OK;
  const dim3 grid(grid_for((n + 3) / 4));
  const int vec_ok = (((uintptr_t)dout | (uintptr_t)out | (uintptr_t)dpre) % 16) == 0;
  if (dtype == ASR_F32)
    hipLaunchKernelGGL((dropout_apply_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)s, dout, (const float*)out,
                       (float*)dpre, n, keep_prob, seed, offset, vec_ok);
  else
    hipLaunchKernelGGL((dropout_apply_kernel<float, bf16_t>), grid, dim3(256), 0, (hipStream_t)s, dout,
                       (const bf16_t*)out, (bf16_t*)dpre, n, keep_prob, seed, offset, vec_ok);
  ASR_CHECK_LAUNCH(h, "asr_relu_bwd_drop");
  return ASR_OK;
}
extern "C" int asr_touch(asr_handle* h, const void* p, size_t bytes, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(p && ((uintptr_t)p) % 16 == 0, "asr_touch: 16-byte aligned buffer");
  const size_t n16 = bytes / 16;
  if (!n16) return ASR_OK;
  int blocks = (int)((n16 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  // the sink word: the handle's error word region is not ours to write; the first scratch word is (never written)
  hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const f32x4_t*)p, n16,
                     (unsigned*)h->scratch);
  ASR_CHECK_LAUNCH(h, "asr_touch");
  return ASR_OK;
}
extern "C" int asr_colsum(asr_handle* h, int dtype, const void* a, int M, int N, int lda, float* out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(asr_dtype_ok(dtype) && a && out && M >= 0 && N > 0 && lda >= N, "asr_colsum: bad args");
  if (dtype == ASR_BF16 && lda == N && N % 8 == 0 && 256 % (N / 8) == 0 && N <= 512 && M >= 4096 &&
      ((uintptr_t)a & 15) == 0) {
    // tall bf16 inputs with whole-row vector loads: at most 2048 blocks, one partial row each, then the final pass
    long long rpb = ((long long)M + 2047) / 2048;
    if (rpb < 256) rpb = 256;
    const int nb = (int)(((long long)M + rpb - 1) / rpb);
    float* pv = (float*)h->scratch;
    hipLaunchKernelGGL(colsum_bf16_vec_kernel, dim3(nb), dim3(256), (size_t)(256 / (N / 8)) * N * sizeof(float),
                       (hipStream_t)s, (const bf16_t*)a, (long long)M, N, rpb, pv);
    if (nb > 64)
      hipLaunchKernelGGL(colsum_final_wide_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)s, pv, nb, N, out);
    else
      hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)s, pv, nb, N, out);
    ASR_CHECK_LAUNCH(h, "asr_colsum");
    return ASR_OK;
  }
  int RB = (M + COLSUM_ROWS - 1) / COLSUM_ROWS > 0 ? (M + COLSUM_ROWS - 1) / COLSUM_ROWS : 1;
  // partial[RB][N], then (for tall inputs) the partials themselves are reduced 512 rows at a time, ping-pong
  // inside the scratch, until few enough remain for the one-thread-per-column final pass
  const size_t need = ((size_t)RB + (size_t)(RB + COLSUM_ROWS - 1) / COLSUM_ROWS + 1) * N * sizeof(float);
  if (need > h->scratch_bytes - ASR_XCH_BYTES) ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_colsum: scratch too small");
  float* partial = (float*)h->scratch;
  {
    const dim3 grid((N + 63) / 64, RB);
    if (dtype == ASR_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)a, M, N, lda, partial);
    else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)a, M, N, lda, partial);
  }
  float* other = partial + (size_t)RB * N;
  while (RB > 64) {
    const int RB2 = (RB + COLSUM_ROWS - 1) / COLSUM_ROWS;
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((N + 63) / 64, RB2), dim3(256), 0, (hipStream_t)s, partial, RB, N, N, other);
    float* t = partial; partial = other; other = t;
    RB = RB2;
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)s, partial, RB, N, out);
  ASR_CHECK_LAUNCH(h, "asr_colsum");
  return ASR_OK;
}

extern "C" int asr_clip_plan(asr_handle* h, const int64_t* offsets_host, int num_tensors,
                             int64_t* chunk_start_host) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(offsets_host && chunk_start_host && num_tensors >= 0, "asr_clip_plan: bad args");
  int64_t c = 0;
  for (int t = 0; t < num_tensors; ++t) {
    chunk_start_host[t] = c;
    c += (offsets_host[t + 1] - offsets_host[t] + NORM_CHUNK - 1) / NORM_CHUNK;
  }
  chunk_start_host[num_tensors] = c;
  return ASR_OK;
}
extern "C" int asr_clip_by_norm_multi(asr_handle* h, float* grads, const int64_t* offsets,
                                      const int64_t* chunk_start, int num_tensors, int64_t total_chunks,
                                      float clip_norm, float* partial_ws, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(grads && offsets && chunk_start && partial_ws && num_tensors >= 0 && clip_norm > 0.f,
           "asr_clip_by_norm_multi: bad args (clip_norm must be > 0)");
  if (num_tensors == 0 || total_chunks == 0) return ASR_OK;
  hipLaunchKernelGGL(sqsum_partial_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)s, grads,
                     offsets, chunk_start, num_tensors, partial_ws);
  ASR_CHECK_LAUNCH(h, "asr_clip_by_norm_multi(sqsum)");
  hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)s, grads,
                     offsets, chunk_start, num_tensors, partial_ws, clip_norm);
  ASR_CHECK_LAUNCH(h, "asr_clip_by_norm_multi(scale)");
  return ASR_OK;
}
extern "C" int asr_weight_decay(asr_handle* h, float* grads, const float* params, const int64_t* offsets,
                                const uint8_t* decay_mask, int num_tensors, float wd, float* l2_out,
                                asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(params && offsets && decay_mask && num_tensors >= 0 && wd >= 0.f, "asr_weight_decay: bad args");
  if (num_tensors == 0) return ASR_OK;
  if (grads) {
    hipLaunchKernelGGL(weight_decay_kernel, dim3(64, num_tensors), dim3(256), 0, (hipStream_t)s, grads, params,
                       offsets, decay_mask, num_tensors, wd);
    ASR_CHECK_LAUNCH(h, "asr_weight_decay");
  }
  if (l2_out) {
    hipLaunchKernelGGL(l2_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, params, offsets, decay_mask,
                       num_tensors, wd, l2_out);
    ASR_CHECK_LAUNCH(h, "asr_weight_decay(l2)");
  }
  return ASR_OK;
}
extern "C" int asr_optimizer_step(asr_handle* h, int optimizer, float* params, const float* grads,
                                  float* slot0, float* slot1, size_t n, float lr, int64_t step,
                                  asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(params && grads, "asr_optimizer_step: null params/grads");
  if (!n) return ASR_OK;
  const int needs0 = optimizer != ASR_OPT_SGD;
  const int needs1 = optimizer == ASR_OPT_ADADELTA || optimizer == ASR_OPT_RMSPROP || optimizer == ASR_OPT_ADAM;
  ASR_NEED(!needs0 || slot0, "asr_optimizer_step: optimizer %d needs slot0", optimizer);
  ASR_NEED(!needs1 || slot1, "asr_optimizer_step: optimizer %d needs slot1", optimizer);
  float adam_lr_t = lr;
  if (optimizer == ASR_OPT_ADAM) {
    ASR_NEED(step >= 1, "asr_optimizer_step: adam needs step >= 1");
    adam_lr_t = (float)(lr * sqrt(1.0 - pow(0.999, (double)step)) / (1.0 - pow(0.9, (double)step)));
  }
  const dim3 g(grid_for(n)), b(256);
  hipStream_t st = (hipStream_t)s;
  switch (optimizer) {
#define ASR_OPT_CASE(O) case O: hipLaunchKernelGGL(optimizer_kernel<O>, g, b, 0, st, params, grads, slot0, slot1, n, lr, adam_lr_t); break
    ASR_OPT_CASE(ASR_OPT_SGD);
    ASR_OPT_CASE(ASR_OPT_MOMENTUM);
    ASR_OPT_CASE(ASR_OPT_NESTEROV);
    ASR_OPT_CASE(ASR_OPT_ADAGRAD);
    ASR_OPT_CASE(ASR_OPT_ADADELTA);
    ASR_OPT_CASE(ASR_OPT_RMSPROP);
    ASR_OPT_CASE(ASR_OPT_ADAM);
#undef ASR_OPT_CASE
    default: ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_optimizer_step: unknown optimizer %d", optimizer);
  }
  ASR_CHECK_LAUNCH(h, "asr_optimizer_step");
  return ASR_OK;
}
extern "C" int asr_scale(asr_handle* h, float* x, size_t n, float scale, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  ASR_NEED(x, "asr_scale: null");
  if (!n) return ASR_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)s, x, n, scale);
  ASR_CHECK_LAUNCH(h, "asr_scale");
  return ASR_OK;
}

// ---------------------------------------------------------------- debug: LDS poisoning
// Every CU's LDS filled with NaN bit patterns (twice as many workgroups as CUs, each claiming the whole 160 KB, so that
// the dispatcher has to walk over every CU).  A kernel that reads LDS words nobody wrote -- the tail of a tile, the unused
// lanes of a reduction array -- normally sees what the previous kernel left there, i.e. the same thing on every run; on a
// freshly booted box it sees something else.  tests/conftest.py runs this in front of every test under ASR_POISON_LDS=1.
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned words, unsigned* sink) {
  extern __shared__ unsigned lds_words[];
  for (unsigned i = threadIdx.x; i < words; i += 256) lds_words[i] = pattern;
  __syncthreads();
  if (sink && lds_words[(threadIdx.x * 97u) % words] != pattern) *sink = 1u;   // keeps the stores alive
  __builtin_amdgcn_s_sleep(64);
}
}  // namespace
extern "C" int asr_debug_poison_lds(asr_handle* h, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  const size_t bytes = (size_t)160 << 10;
  (void)hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  hipLaunchKernelGGL(poison_lds_kernel, dim3((unsigned)(2 * h->num_cu)), dim3(256), bytes, (hipStream_t)s, 0xFFFFFFFFu,
                     (unsigned)(bytes / 4), (unsigned*)nullptr);
  ASR_CHECK_LAUNCH(h, "asr_debug_poison_lds");
  return ASR_OK;
}
