// Shared helpers for the gfx950 kernels of libasr_hip.so.  CDNA4 only: wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/asr_hip.h"

struct asr_handle {
  int device;
  int num_cu;
  char name[128];
  char err[512];
  // device scratch owned by the handle (allocated once in asr_create, never in a hot call):
  // split-K slabs of asr_gemm.  Calls on one handle are single-stream by contract.
  void* scratch;
  size_t scratch_bytes;
  // exchange areas of the multi-CU LSTM kernels (lstm_cluster.hip): bytes of each area that hold stale tags, and the
  // area the next launch runs on
  size_t xch_dirty[2];
  int xch_next;
  // side-stream handles: number of leading XCDs the lean weight-gradient GEMM leaves to the recurrence clusters
  int xcd_skip;
  // workgroups the lean reduction-major GEMM aims at (0 = default 512); see asr_set_gemm_tn_workgroups
  int tn_wgs;
  // asr_lstm_bwd_ex: the clip the forward applied to the cell state when a clamped state must pass no gradient
  // (tf.clip_by_value of LSTMCell); 0 = the straight-through clip of LSTMBlockCell.  Set around the kernel dispatch only.
  float bptt_clip;
};

#define ASR_FAIL(h, code, ...)                                  \
  do {                                                          \
    if (h) snprintf((h)->err, sizeof((h)->err), __VA_ARGS__);   \
    return (code);                                              \
  } while (0)

#define ASR_CHECK_LAUNCH(h, what)                                                   \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) ASR_FAIL(h, ASR_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;  // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;   // MFMA 16x16 C/D fragment

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN stays NaN: gfx950's own conversion (v_cvt_pk_bf16_f32, two values per instruction) -- the
// integer form of rounds 1-3 (compare, select, two adds, two shifts per value) was a third of the VALU work of every
// bf16 epilogue; same result for every finite input and infinities
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

// Gate nonlinearities.  Precise libm forms (expf/tanhf, <= 1-2 ulp): the fp32 path
// is the parity path (loss within 1e-4 relative of the oracle).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sum over aligned groups of W lanes (W = 16, 32 or 64), result in every lane of the group.  The steps inside a
// 16-lane row are DPP modifiers on the add (quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8), not LDS permutes.
template <int CTRL> __device__ __forceinline__ float dpp_mov_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int W> __device__ __forceinline__ float group_reduce_sum(float v) {
  v += dpp_mov_f32<0xB1>(v);
  v += dpp_mov_f32<0x4E>(v);
  v += dpp_mov_f32<0x124>(v);
  v += dpp_mov_f32<0x128>(v);
  if (W >= 32) v += __shfl_xor(v, 16, 64);
  if (W >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
// tanh from the hardware exp2 / reciprocal: 1 - 2 / (exp(2x) + 1), absolute error < 2e-7 (each of v_exp_f32 and
// v_rcp_f32 is good to 1 ulp), saturating correctly at +-1.  For the attention energies -- sum_a v_a tanh(.) over every
// encoder frame and decoder step, which the libm form (~40 instructions) makes ALU-bound; NOT for the cell
// nonlinearities, which stay on tanhf_.
__device__ __forceinline__ float fast_tanhf(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// Philox4x32-10 block -> the four dropout multipliers of elements 4 ctr' .. 4 ctr' + 3 (ctr = offset + element / 4);
// the generator of asr_dropout_mask (elementwise.hip), for the kernels that form the mask in their epilogue.
__device__ __forceinline__ void asr_dropout_words(uint64_t ctr, uint64_t seed, float keep, float inv, float m[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
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
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = ((c[j] >> 8) * (1.0f / 16777216.0f) < keep) ? inv : 0.f;
}

// top ASR_XCH_BYTES of the scratch: granule exchange + error word of the multi-CU LSTM kernels
// (64 MB: two areas; the BPTT kernel's H = 512 clusters of 16 CUs take 1 MB of slots each, 16 clusters at B = 128)
static constexpr size_t ASR_XCH_BYTES = (size_t)64 << 20;
bool asr_cluster_fwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* xproj,
                         const void* whp, const float* peep, const int32_t* seq_len, float fb,
                         float clip, void* gates, void* hout, float* cs, float* cf, float* hf,
                         hipStream_t st);
bool asr_cluster_bwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* dhout,
                         const void* gates, const float* cs, const void* whpb, const float* peep,
                         const int32_t* seq_len, const float* dcf, const float* dhf, void* dgates,
                         float* dpeep_part, hipStream_t st);
// fp32 operands (H = 128 on two CUs per direction); same contract: false = not applicable, nothing launched
bool asr_cluster_fwd_f32_try(asr_handle* h, int T, int B, int H, int ndir, const float* xproj,
                             const void* whp, const float* peep, const int32_t* seq_len, float fb,
                             float clip, void* gates, void* hout, float* cs, float* cf, float* hf,
                             hipStream_t st);
bool asr_cluster_bwd_f32_try(asr_handle* h, int T, int B, int H, int ndir, const float* dhout,
                             const void* gates, const float* cs, const void* whpb, const float* peep,
                             const int32_t* seq_len, const float* dcf, const float* dhf, void* dgates,
                             float* dpeep_part, hipStream_t st);

// GRU forward on clusters (lstm_cluster.hip: the exchange machinery lives there); false = not applicable
bool asr_cluster_gru_fwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* xg, const float* xc,
                             const float* wgh, const float* wch, const int32_t* seq_len, float* r, float* u, float* c,
                             float* rh, float* hout, float* h_final, hipStream_t st);

bool asr_cluster_gru_bwd_try(asr_handle* h, int T, int B, int H, int ndir, const float* dout, const float* d_h_final,
                             const float* hout, const float* r, const float* u, const float* c, const float* wghT,
                             const float* wchT, const int32_t* seq_len, float* dgate, float* dcand, hipStream_t st);

static inline int asr_dtype_ok(int dt) { return dt == ASR_F32 || dt == ASR_BF16; }
static inline size_t asr_dtype_size(int dt) { return dt == ASR_BF16 ? 2 : 4; }
