// LDS-tiled MFMA GEMM for gfx950:  C[M,N] = op(A) * op(B) (+bias) (+C).
//   fp32 operands -> v_mfma_f32_16x16x4_f32   (exact fp32, the parity path)
//   bf16 operands -> v_mfma_f32_16x16x32_bf16 (fp32 accumulate)
// Used for every contraction that is not on the serial recurrence: the LSTM input
// projections x*W_x hoisted over all T (models/encoders/core/blstm.py:286-320), the
// bottleneck/output FC (models/ctc/ctc.py:198-233) and all dW / dX products of backward.
//
// Tile BM x BN x 128 bytes-of-K, 256 threads = 4 waves in a 2x2 grid, each wave owns a
// (BM/2)x(BN/2) block of 16x16 MFMA tiles.  Both operands are staged K-contiguous in
// LDS (As[m][k], Bs[n][k], row padded by 16 B so the 16 rows of a fragment read hit
// distinct banks); global loads are 16 B per lane along whichever dimension is
// contiguous in memory, register-staged so the next tile's loads fly during the MFMAs.
#include "common.h"
#include <stdlib.h>

namespace {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: "set once per process" leaves a
// second GPU of the same process without the LDS opt-in.  One bit per device ordinal.
static inline bool first_on_device(unsigned long long& mask) {
  int d = 0;
  (void)hipGetDevice(&d);
  const unsigned long long bit = 1ull << (d & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

template <typename T> struct GT;
template <> struct GT<float> {
  static constexpr int BK = 32;   // 128 B of K per row
  static constexpr int VEC = 4;   // elements per 16 B
};
template <> struct GT<bf16_t> {
  static constexpr int BK = 64;
  static constexpr int VEC = 8;
};

template <typename T> struct Vec16 { T v[GT<T>::VEC]; } __attribute__((aligned(16)));

template <typename TO> __device__ __forceinline__ void store_out(TO* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }
template <typename TO> __device__ __forceinline__ float load_out(const TO* p);
template <> __device__ __forceinline__ float load_out<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_out<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

// Stage one operand tile [ROWS x BK] (logical: row r, reduction index k) into registers.
// KCONT: memory is k-contiguous (elem(r,k) = p[r*ld + k]); else r-contiguous (p[k*ld + r]).
template <typename T, int ROWS, bool KCONT>
struct TileLoader {
  static constexpr int BK = GT<T>::BK, VEC = GT<T>::VEC;
  static constexpr int NVEC = ROWS * BK / VEC;
  static constexpr int PER_THREAD = NVEC / 256;
  static_assert(NVEC % 256 == 0, "tile must divide over 256 threads");
  Vec16<T> reg[PER_THREAD];

  __device__ __forceinline__ void load(const T* __restrict__ p, int ld, int r0, int k0, int R,
                                       int K, bool aligned) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int v = tid + i * 256;
      int r, k;
      size_t off;
      int nr, nk;  // extent of the vector in r and k
      if (KCONT) {
        r = v / (BK / VEC); k = (v % (BK / VEC)) * VEC;
        off = (size_t)(r0 + r) * ld + (k0 + k);
        nr = 1; nk = VEC;
      } else {
        k = v / (ROWS / VEC); r = (v % (ROWS / VEC)) * VEC;
        off = (size_t)(k0 + k) * ld + (r0 + r);
        nr = VEC; nk = 1;
      }
      const bool full = (r0 + r + nr <= R) && (k0 + k + nk <= K);
      if (full && aligned) {
        reg[i] = *reinterpret_cast<const Vec16<T>*>(p + off);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int rr = r0 + r + (KCONT ? 0 : j), kk = k0 + k + (KCONT ? j : 0);
          T val = T(0);
          if (rr < R && kk < K) val = p[KCONT ? ((size_t)rr * ld + kk) : ((size_t)kk * ld + rr)];
          reg[i].v[j] = val;
        }
      }
    }
  }
  // LDS image: s[r][k], row stride LDS_LD elements
  __device__ __forceinline__ void store(T* s, int lds_ld) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int v = tid + i * 256;
      if (KCONT) {
        const int r = v / (BK / VEC), k = (v % (BK / VEC)) * VEC;
        *reinterpret_cast<Vec16<T>*>(s + r * lds_ld + k) = reg[i];
      } else {
        const int k = v / (ROWS / VEC), r = (v % (ROWS / VEC)) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[(r + j) * lds_ld + k] = reg[i].v[j];
      }
    }
  }
};

template <typename T, typename TO, int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(int M, int N, int K, const T* __restrict__ A,
                                                   int lda, const T* __restrict__ B, int ldb,
                                                   TO* __restrict__ C, int ldc,
                                                   const float* __restrict__ bias, int accumulate,
                                                   int a_aligned, int b_aligned, int kchunk,
                                                   float* __restrict__ partial, int act, int c_vec) {
  constexpr int BK = GT<T>::BK, VEC = GT<T>::VEC;
  constexpr int LDS_LD = BK + VEC;  // +16 B pad
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + BM * LDS_LD;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // A tile: rows = m. memory k-contiguous iff !TA.  B tile: rows = n. k-contiguous iff TB.
  TileLoader<T, BM, !TA> la;
  TileLoader<T, BN, TB> lb;

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // split-K: blockIdx.z owns the reduction range [kbeg, kend)
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(K, kbeg + kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  // LDS is double-buffered: tile kt+1 is stored into the other buffer while nobody reads it, so
  // one barrier per k-tile orders everything (writers of buffer b at kt+1 have passed the
  // barrier of kt, which every reader of b reached only after finishing tile kt-1).
  constexpr int STAGE = (BM + BN) * LDS_LD;
  la.load(A, lda, m0, kbeg, M, kend, a_aligned);
  lb.load(B, ldb, n0, kbeg, N, kend, b_aligned);
  la.store(As, LDS_LD);
  lb.store(Bs, LDS_LD);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const T* Ac = As + (kt & 1) * STAGE;
    const T* Bc = Bs + (kt & 1) * STAGE;
    if (kt + 1 < nkt) {
      la.load(A, lda, m0, kbeg + (kt + 1) * BK, M, kend, a_aligned);
      lb.load(B, ldb, n0, kbeg + (kt + 1) * BK, N, kend, b_aligned);
    }
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8_t a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const bf16x8_t*>(Ac + (wm * WM + i * 16 + fr) * LDS_LD + ks * 32 + fq * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const bf16x8_t*>(Bc + (wn * WN + j * 16 + fr) * LDS_LD + ks * 32 + fq * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = Ac[(wm * WM + i * 16 + fr) * LDS_LD + ks * 4 + fq];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bc[(wn * WN + j * 16 + fr) * LDS_LD + ks * 4 + fq];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) {
      la.store(As + ((kt + 1) & 1) * STAGE, LDS_LD);
      lb.store(Bs + ((kt + 1) & 1) * STAGE, LDS_LD);
    }
    __syncthreads();
  }

  // epilogue.  The MFMAs above take (B fragment, A fragment), i.e. they produce the TRANSPOSED
  // tile: a lane holds C[m = lane&15][n = (lane>>4)*4 + 0..3] -- four consecutive columns of one
  // row -> one 16-byte (fp32) / 8-byte (bf16) store per tile instead of four scalar ones (the
  // scalar-store epilogue was store-issue bound and as long as the K loop at K = 512).
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int m = m0 + wm * WM + i * 16 + (lane & 15);
      const int nb = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
      if (m >= M || nb >= N) continue;
      const bool full = nb + 3 < N;
      if (partial) {  // split-K partial slab [z][M][N]; bias / accumulate applied by the reducer
        float* pp = partial + ((size_t)blockIdx.z * M + m) * N + nb;
        if (full && (N & 3) == 0) *reinterpret_cast<f32x4_t*>(pp) = acc[i][j];
        else
          for (int r = 0; r < 4 && nb + r < N; ++r) pp[r] = acc[i][j][r];
        continue;
      }
      TO* cp = C + (size_t)m * ldc + nb;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + ((bias && nb + r < N) ? bias[nb + r] : 0.f);
      if (full && c_vec) {
        if constexpr (sizeof(TO) == 4) {
          if (accumulate) {
            const f32x4_t o = *reinterpret_cast<const f32x4_t*>(cp);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
          }
          if (act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
        } else {
          typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
          if (accumulate) {
            const us4_t o = *reinterpret_cast<const us4_t*>(cp);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bf16_to_f32(o[r]);
          }
          if (act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          *reinterpret_cast<us4_t*>(cp) = (us4_t){f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
        }
      } else {
        for (int r = 0; r < 4 && nb + r < N; ++r) {
          float w = v[r];
          if (accumulate) w += load_out<TO>(cp + r);
          if (act == 1) w = fmaxf(w, 0.f);
          store_out<TO>(cp + r, w);
        }
      }
    }
}

// LDS images [rows][64 + 8 bf16] read as MFMA fragments with ds_read_b128 (16 lanes = 16 rows of one 16-byte k-slot): the
// hardware serves a wave in lane groups that pair rows 0-3, 12-15 of k-slot q with rows 4-11 of k-slot q ^ 1
// (lstm_cluster.hip, lds_swz), so with plain padding those two sets collide on the 16 sixteen-byte bank groups --
// SQ_LDS_BANK_CONFLICT on 11-13 % of the CU cycles of these kernels (profiles/r03_pmc_util.md).  Swapping the two halves of
// every 32 bytes OF A ROW for rows 4-11 (mod 16), by writers and readers alike, makes every group hit 16 distinct slots
// (row stride 9 slots: 9 r + q mod 16 is a permutation of the 16 rows).
__device__ __forceinline__ unsigned nt_swz(int row) { return (((row + 4) >> 3) & 1) ? 8u : 0u; }   // in bf16 elements

struct GemmDrop {               // dropout multiplier of the OUTPUT formed in the epilogue (element m*N + n of a contiguous
  float keep;                   // [M,N] tensor -> word (m*N + n) % 4 of Philox block offset + (m*N + n) / 4): asr_dropout_mask's
  uint64_t seed, offset;        // values without the mask tensor
  int use;
};

// ---------------------------------------------------------------- lean NT kernel (bf16)
// C[M,N] = A[M,K] * Bt[N,K]^T (+bias)(+C)(relu): both operands reduction-contiguous, K % 64 == 0,
// N % 128 == 0, 16-byte aligned rows.  This is the shape of every GEMM on the critical path of
// the BLSTM step (x W_x with W_x pre-transposed, dG W_x^T).  No bounds code in the loop: rows of
// the last M tile are clamped for the loads and masked at the store.  128x128x64 tiles, register
// staged global->LDS with the next tile's loads in flight during the MFMAs, double-buffered LDS
// (one barrier per k-tile), transposed accumulators (16-byte stores).  Tiles are mapped to blocks
// so that the N-tiles sharing an A row-panel run on ONE XCD (block b lands on XCD b % 8): the
// panel is then fetched into one L2 instead of eight.
template <typename TO>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                           int lda, const bf16_t* __restrict__ Bt, int ldb,
                                                           TO* __restrict__ C, int ldc,
                                                           const float* __restrict__ bias, int accumulate,
                                                           int act, const float* __restrict__ mul, int ldm,
                                                           GemmDrop drop) {
  constexpr int BM = 128, BN = 128, BK = 64, LD = BK + 8;
  constexpr int STAGE = (BM + BN) * LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* S = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN, ntm = (M + BM - 1) / BM, total = ntn * ntm;
  int t = blockIdx.x;
  if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);   // XCD x gets a contiguous run of tiles
  const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;

  // 4 A vectors + 4 B vectors of 16 B per thread per k-tile: vector v -> row v>>3, k-offset (v&7)*8
  const bf16_t* pa[4];
  const bf16_t* pb[4];
  unsigned so[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = tid + i * 256, r = v >> 3, kv = (v & 7) * 8;
    pa[i] = A + (size_t)min(m0 + r, M - 1) * lda + kv;
    pb[i] = Bt + (size_t)(n0 + r) * ldb + kv;
    so[i] = (unsigned)(r * LD) + ((unsigned)kv ^ nt_swz(r));
  }
  bf16x8_t ra[4], rb[4];
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const bf16x8_t*>(pa[i]);
      rb[i] = *reinterpret_cast<const bf16x8_t*>(pb[i]);
      pa[i] += BK;
      pb[i] += BK;
    }
  };
  auto sstore = [&](bf16_t* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<bf16x8_t*>(st + so[i]) = ra[i];
      *reinterpret_cast<bf16x8_t*>(st + BM * LD + so[i]) = rb[i];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  gload();
  sstore(S);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned aoff = (unsigned)((wm * 64 + fr) * LD) + ((unsigned)(fq * 8) ^ nt_swz(fr));
  const unsigned boff = (unsigned)(BM * LD + (wn * 64 + fr) * LD) + ((unsigned)(fq * 8) ^ nt_swz(fr));
  for (int kt = 0; kt < nkt; ++kt) {
    const bf16_t* cur = S + (kt & 1) * STAGE;
    if (kt + 1 < nkt) gload();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(cur + aoff + i * 16 * LD + ks * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(cur + boff + j * 16 * LD + ks * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(S + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }
  // lane holds C[m = fr][n = fq*4 .. +3] of each 16x16 tile
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 64 + j * 16 + fq * 4;
      TO* cp = C + (size_t)m * ldc + nb;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (bias) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + nb);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if constexpr (sizeof(TO) == 4) {
        if (accumulate) {
          const f32x4_t o = *reinterpret_cast<const f32x4_t*>(cp);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += o[r];
        }
        if (act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (mul) {                                         // e.g. the dropout mask of the tensor this gradient is of
          const f32x4_t mv = *reinterpret_cast<const f32x4_t*>(mul + (size_t)m * ldm + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= mv[r];
        }
        if (drop.use) {                                    // the same mask from its Philox counter (N % 4 == 0)
          float mk[4];
          asr_dropout_words(drop.offset + ((size_t)m * N + nb) / 4, drop.seed, drop.keep, 1.f / drop.keep, mk);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= mk[r];
        }
        *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
      } else {
        typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
        if (accumulate) {
          const us4_t o = *reinterpret_cast<const us4_t*>(cp);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bf16_to_f32(o[r]);
        }
        if (act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        *reinterpret_cast<us4_t*>(cp) = (us4_t){f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
      }
    }
  }
}

// ---------------------------------------------------------------- NT kernel, large tiles (bf16)
// The 128 x 128 kernel above is bound by LDS traffic, not by the matrix pipes: a wave's 64 x 64 tile reads 8 KB of
// fragments per 16 MFMAs, eight waves per CU -> 128 KB read + 64 KB written per k-tile round = 1 536 LDS clocks against
// 1 024 MFMA clocks per SIMD (measured 0.62-0.64 PFLOP/s on the [105 600 x 1 024] x [1 024 x 4 096] projection of cfg C).
// This one gives a wave a (BM / WM) x (BN / WN) tile of a BM x BN block: 128 x 64 wave tiles read 12 KB per 32 MFMAs,
// 128 x 128 ones 16 KB per 64.  Measured on one MI355X (scripts/probe_matmul.py, fp32 output, TFLOP/s on the
// [12 448 x 512] x [512 x 2 048] / [51 136 x 1 024] x [1 024 x 4 096] / [105 600 x 1 024] x [1 024 x 4 096] projections):
//   128 x 128 block, four waves of 64 x 64, two blocks per CU (the kernel above)      509 / 578 / 643
//   256 x 256 block, EIGHT waves of 128 x 64 (two per SIMD, 244 VGPRs)                496 / 721 / 842   <- used from 512 tiles on
//   256 x 128 block, four waves of 128 x 64 (one per SIMD: nothing hides a stall)     321 / 491 / 512
//   256 x 256 block, four waves of 128 x 128: the 256 accumulators ARE the AGPR file; the allocator keeps half of them
//   in VGPRs and copies (433 v_accvgpr moves per k-tile, 94 spilled registers)        169 / 200 / 205
// (torch.matmul -> hipBLASLt with a bf16 result reaches 420 / 1 029 / 1 237 on the same operands.)  Same staging scheme (next k-tile's global loads in flight during the MFMAs, one barrier
// per k-tile), same transposed accumulators and epilogue; the staging loads are buffer loads (descriptor in SGPRs, a
// per-lane 32-bit byte offset that never changes, the k-tile's offset in one SGPR).
// (Round 5, measured: a PERSISTENT form -- 256 / 512 workgroups walking the tiles, the next tile's first k-tile requested
// before the epilogue's stores -- is 2 % SLOWER on every cfg C / D projection (1178 -> 1206 us, 493 -> 516 us): the
// dispatcher already replaces a finished workgroup faster than the epilogue drains.  scripts/bench_nt.py.)
template <typename TO, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 1) void gemm_nt_bf16_big_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                                           int lda, const bf16_t* __restrict__ Bt, int ldb,
                                                                           TO* __restrict__ C, int ldc,
                                                                           const float* __restrict__ bias, int accumulate,
                                                                           int act, const float* __restrict__ mul, int ldm,
                                                                           GemmDrop drop) {
  constexpr int BK = 64, LD = BK + 8, NT = WM * WN * 64;
  constexpr int STAGE = (BM + BN) * LD;
  constexpr int TM = BM / WM, TN = BN / WN, TI = TM / 16, TJ = TN / 16;
  constexpr int VA = BM * 8 / NT, VB = BN * 8 / NT, RSTEP = NT / 8;   // staging vectors per thread, rows between them
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* S = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = N / BN, ntm = (M + BM - 1) / BM, total = ntn * ntm;
  int t = blockIdx.x;
  if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);   // XCD x gets a contiguous run of tiles
  const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;

  // staging vector i of a thread: row (tid >> 3) + RSTEP i, k-offset (tid & 7) * 8; rows of A past M (last tile) are
  // clamped to row M - 1 and masked at the store
  const int r0 = tid >> 3, kv = (tid & 7) * 8;
  unsigned oa[VA], ob[VB];
#pragma unroll
  for (int i = 0; i < VA; ++i) oa[i] = (unsigned)(((size_t)min(m0 + r0 + RSTEP * i, M - 1) * lda + kv) * sizeof(bf16_t));
#pragma unroll
  for (int i = 0; i < VB; ++i) ob[i] = (unsigned)(((size_t)(n0 + r0 + RSTEP * i) * ldb + kv) * sizeof(bf16_t));
  const unsigned so0 = (unsigned)(r0 * LD) + ((unsigned)kv ^ nt_swz(r0));   // (a thread's rows are RSTEP = 32 k apart: same class)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(A), 0, (int)min((size_t)0xFFFFFFFFu, ((size_t)(M - 1) * lda + K) * sizeof(bf16_t)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(Bt), 0, (int)min((size_t)0xFFFFFFFFu, ((size_t)(N - 1) * ldb + K) * sizeof(bf16_t)), 0x00020000);
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  u32x4_t ra[VA], rb[VB];
  unsigned koff = 0;                                       // byte offset of the k-tile being fetched (uniform)
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < VA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa[i], koff, 0);
#pragma unroll
    for (int i = 0; i < VB; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ob[i], koff, 0);
    koff += BK * sizeof(bf16_t);
  };
  auto sstore = [&](bf16_t* st) {
#pragma unroll
    for (int i = 0; i < VA; ++i) *reinterpret_cast<u32x4_t*>(st + so0 + i * RSTEP * LD) = ra[i];
#pragma unroll
    for (int i = 0; i < VB; ++i) *reinterpret_cast<u32x4_t*>(st + BM * LD + so0 + i * RSTEP * LD) = rb[i];
  };

  f32x4_t acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  gload();
  sstore(S);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned aoff = (unsigned)((wm * TM + fr) * LD) + ((unsigned)(fq * 8) ^ nt_swz(fr));
  const unsigned boff = (unsigned)(BM * LD + (wn * TN + fr) * LD) + ((unsigned)(fq * 8) ^ nt_swz(fr));
  for (int kt = 0; kt < nkt; ++kt) {
    const bf16_t* cur = S + (kt & 1) * STAGE;
    if (kt + 1 < nkt) gload();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[TI];
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(cur + aoff + i * 16 * LD + ks * 32);
#pragma unroll
      for (int jh = 0; jh < TJ; jh += 4) {                  // B fragments in runs of four (registers)
        bf16x8_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(cur + boff + (jh + j) * 16 * LD + ks * 32);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][jh + j], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) sstore(S + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }
  // lane holds C[m = fr][n = fq*4 .. +3] of each 16x16 tile
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = m0 + wm * TM + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int nb = n0 + wn * TN + j * 16 + fq * 4;
      TO* cp = C + (size_t)m * ldc + nb;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (bias) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + nb);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if constexpr (sizeof(TO) == 4) {
        if (accumulate) {
          const f32x4_t o = *reinterpret_cast<const f32x4_t*>(cp);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += o[r];
        }
        if (act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (mul) {
          const f32x4_t mv = *reinterpret_cast<const f32x4_t*>(mul + (size_t)m * ldm + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= mv[r];
        }
        if (drop.use) {
          float mk[4];
          asr_dropout_words(drop.offset + ((size_t)m * N + nb) / 4, drop.seed, drop.keep, 1.f / drop.keep, mk);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= mk[r];
        }
        *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
      } else {
        typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
        if (accumulate) {
          const us4_t o = *reinterpret_cast<const us4_t*>(cp);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bf16_to_f32(o[r]);
        }
        if (act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        *reinterpret_cast<us4_t*>(cp) = (us4_t){f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
      }
    }
  }
}

template <typename TO, int BM, int BN, int WM, int WN>
static void launch_gemm_nt_big(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                               const float* bias, int accumulate, hipStream_t st, int act, const float* mul, int ldm,
                               GemmDrop drop) {
  const size_t lds = (size_t)2 * (BM + BN) * (64 + 8) * sizeof(bf16_t);
  static unsigned long long attr_done = 0;
  auto k = gemm_nt_bf16_big_kernel<TO, BM, BN, WM, WN>;
  if (first_on_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long tiles = (long)(N / BN) * ((M + BM - 1) / BM);
  hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(WM * WN * 64), lds, st, M, N, K, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, (TO*)C, ldc, bias, accumulate, act, mul, ldm, drop);
}

template <typename TO>
bool try_gemm_nt_bf16(int transA, int transB, int M, int N, int K, const void* A, int lda, const void* B,
                      int ldb, void* C, int ldc, const float* bias, int accumulate, hipStream_t st, int act,
                      const float* mul, int ldm, GemmDrop drop = GemmDrop{1.f, 0, 0, 0}) {
  if (transA || !transB || K % 64 != 0 || N % 128 != 0 || M < 1024) return false;
  if (drop.use && sizeof(TO) != 4) return false;
  if (mul && (sizeof(TO) != 4 || ldm % 4 != 0 || ((uintptr_t)mul) % 16 != 0)) return false;
  if (lda % 8 != 0 || ldb % 8 != 0 || ((uintptr_t)A) % 16 != 0 || ((uintptr_t)B) % 16 != 0) return false;
  if (ldc % 4 != 0 || ((uintptr_t)C) % (4 * sizeof(TO)) != 0 || (bias && ((uintptr_t)bias) % 16 != 0)) return false;
  // large products (>= 512 tiles of 256 x 256, i.e. two full rounds of the chip): 256 x 256 blocks of eight waves, see
  // gemm_nt_bf16_big_kernel; ASR_GEMM_NT_BIG=0 keeps the 128 x 128 tiles (A/B), =2 takes the big ones from M >= 2048 on
  static const int big = [] { const char* e = getenv("ASR_GEMM_NT_BIG"); return e ? atoi(e) : 1; }();
  const bool fits32 = ((size_t)(M - 1) * lda + K) * sizeof(bf16_t) < 0xFFFFFFFFull && ((size_t)(N - 1) * ldb + K) * sizeof(bf16_t) < 0xFFFFFFFFull;
  const long tiles256 = (long)(N / 256) * ((M + 255) / 256);
  if (big && fits32 && N % 256 == 0 && (big == 2 ? M >= 2048 : tiles256 >= 512)) {
    launch_gemm_nt_big<TO, 256, 256, 2, 4>(M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act, mul, ldm, drop);
    return true;
  }
  const size_t lds = (size_t)2 * (128 + 128) * (64 + 8) * sizeof(bf16_t);
  static unsigned long long attr_done = 0;
  if (first_on_device(attr_done)) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int total = (N / 128) * ((M + 127) / 128);
  hipLaunchKernelGGL(gemm_nt_bf16_kernel<TO>, dim3(total), dim3(256), lds, st, M, N, K, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, (TO*)C, ldc, bias, accumulate, act, mul, ldm, drop);
  return true;
}

// Workgroup b of a 1-D grid runs on XCD b % 8 (asr_debug_placement).  Side-stream kernels leave the first `skip` XCDs
// -- where the recurrence clusters live -- alone: their workgroups there retire at once and the work is dealt over
// the others.  Returns the dense work index of this workgroup, -1 for a workgroup that has nothing to do.
__device__ __forceinline__ int xcd_work_index(int skip) {
  const int b = blockIdx.x, x = b & 7;
  if (x < skip) return -1;
  return (b >> 3) * (8 - skip) + (x - skip);
}
static inline unsigned xcd_grid(long work, int skip) {     // workgroups to launch for `work` work items
  const long per = 8 - skip;
  return (unsigned)((work + per - 1) / per * 8);
}

// ---------------------------------------------------------------- skinny fp32 kernel (M <= 32)
// C[M,N] = A[M,K] op(B) (+bias)(+C)(relu), fp32, M <= 32: the per-step products of the attention decoder (cell input
// 1600 -> 2048 and back, query 512 <-> 128 at B = 32 utterances, attention_decoder.py:142-229).  They are weight
// streaming -- 13 MB of W_cell per step against 0.2 GFLOP -- and sit on the decoder's serial chain, so what counts is
// how many CUs pull on the weights and how many bytes each has in flight.  One workgroup owns 16*NT output columns of
// ONE 16-row group (grid.y = row groups: the fp32 MFMA rate of the 64..100 column tiles alone is ~7 us per product,
// so the rows are spread over twice the CUs and B comes out of L2 for the second group); its eight waves split K in
// 64-element units (interleaved), and a wave issues the loads of its NEXT unit (4 float4 of A, NT x 4 float4 -- or
// NT x 16 row-segment words -- of B) before the MFMAs of the current one.  TRANSB reads a float4 of a B^T row per
// lane, the plain form reads full 128-byte lines (NT = 2).  The products are exact fp32 MFMAs (16x16x4) and the eight
// partial tiles meet in LDS in a fixed order (deterministic).
// Requires K % 64 == 0, N % (16 NT) == 0, 16-byte aligned rows.
constexpr int SK_WAVES = 8;
// CELL: the product is the decoder cell's pre-activation with the gate columns INTERLEAVED (column 4 u + g = gate g of unit
// u, asr_lstm_cell_gemm_prep) so that a workgroup's 32 columns are all four gates of 8 units, and the LSTM cell
// (cell_fwd_kernel of attention.hip, same arithmetic in the same order) runs in the epilogue on the reduced tile: one launch
// and one dependent round trip fewer per decoder step, the pre-activations never written.  The cell's own inputs are
// requested before the product's first load.
struct SkinnyCell {
  const float *c_prev, *h_prev, *peep, *live, *out_mask;
  float *gates, *c_raw, *c_out, *h_out, *h_raw, *cell_out, *h_out2, *cell_out2;
  int U, ld_h2, ld_c2;
  float fb, clip;
};
__device__ __forceinline__ float sk_sigf(float x) { return 1.0f / (1.0f + expf(-x)); }
// BH: the weights are bf16 (the activations stay fp32, the products are exact fp32 MFMAs of an fp32 value and a
// bf16-valued one): half the bytes of a launch that is weight streaming.  !TRANSB: Bm is the FRAGMENT image of
// asr_lstm_cell_gemm_prep_h -- per (32-column block, 64-row unit) the 64 lanes' 32 values each, contiguous, in the order
// the multiplies consume them: four 16-byte loads per lane and unit instead of 32 strided dword loads.  TRANSB: Bm is a
// bf16 [N, ldb] matrix of B^T rows; a lane takes 16 consecutive k of its row (two 16-byte loads) and the matching 16
// consecutive k of its A row -- MFMA slot rg carries k = kb + 16 rg + t at multiply t (any k-to-slot map is a valid dot
// product as long as both operands use it).
template <bool TRANSB, int NT, bool CELL = false, bool BH = false>
__global__ __launch_bounds__(64 * SK_WAVES) void gemm_skinny_f32_kernel(int M, int N, int K, const float* __restrict__ A,
                                                                        int lda, const float* __restrict__ Bm, int ldb,
                                                                        float* __restrict__ C, int ldc,
                                                                        const float* __restrict__ bias, int accumulate,
                                                                        int act, SkinnyCell cell) {
  constexpr int TN = 16 * NT;
  __shared__ float red[SK_WAVES][16][TN + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rg = lane >> 4;
  const int n0 = blockIdx.x * TN, m0 = blockIdx.y * 16;
  // CELL: thread idx < 16 * TN / 4 owns (row m0 + idx / (TN/4), unit n0/4 + idx % (TN/4))
  float cl_cp = 0.f, cl_hp = 0.f, cl_lv = 0.f, cl_wci = 0.f, cl_wcf = 0.f, cl_wco = 0.f, cl_om = 1.f;
  const int cl_m = m0 + (int)threadIdx.x / (TN / 4), cl_u = n0 / 4 + (int)threadIdx.x % (TN / 4);
  const bool cl_on = CELL && (int)threadIdx.x < 16 * (TN / 4) && cl_m < M;
  if (CELL && cl_on) {
    const size_t ci = (size_t)cl_m * cell.U + cl_u;
    cl_cp = cell.c_prev[ci];
    cl_hp = cell.h_prev[ci];
    cl_lv = cell.live[cl_m];
    if (cell.peep) { cl_wci = cell.peep[cl_u]; cl_wcf = cell.peep[cell.U + cl_u]; cl_wco = cell.peep[2 * cell.U + cl_u]; }
    if (cell.out_mask) cl_om = cell.out_mask[ci];
  }
  typedef __attribute__((ext_vector_type(4))) unsigned skw4_t;
  static_assert(!BH || (TRANSB ? NT == 1 : NT == 2), "bf16 weights: NT = 2 fragment image / NT = 1 transposed rows");
  struct Frag {
    f32x4_t a[4], b[BH ? 1 : NT][BH ? 1 : 4];
    skw4_t w[BH ? (TRANSB ? 2 : 4) : 1];
    float keep;                 // 1 for a real unit, 0 for a round past the end (A is zeroed at the multiply)
  };
  f32x4_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // rows past M read a valid row, masked at the store
  const float* ap = A + (size_t)min(m0 + col, M - 1) * lda + ((BH && TRANSB) ? rg * 16 : rg * 4);
  const int units = K / 64;
  const bf16_t* Bh = reinterpret_cast<const bf16_t*>(Bm);
  // round r of a wave is the 64-element unit wave + 8r; rounds past the end re-read the last unit with A zeroed, so
  // every wave runs the same (even) number of rounds and every prefetch below is consumed -- a prefetch whose use sits
  // behind a branch gets sunk below the multiplies by the compiler, which serialises load and multiply again
  auto load = [&](int round, Frag& f) {
    const int unit = wave + round * SK_WAVES;
    const bool valid = unit < units;
    const int kb = (valid ? unit : units - 1) * 64;
    f.keep = valid ? 1.f : 0.f;
    if constexpr (BH) {
      if constexpr (TRANSB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.a[j] = *reinterpret_cast<const f32x4_t*>(ap + kb + j * 4);   // A[row][kb + 16rg + 4j + e]
        const bf16_t* bp = Bh + (size_t)(n0 + col) * ldb + kb + rg * 16;                           // Bt[n][kb + 16rg + 0..15]
        f.w[0] = *reinterpret_cast<const skw4_t*>(bp);
        f.w[1] = *reinterpret_cast<const skw4_t*>(bp + 8);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.a[j] = *reinterpret_cast<const f32x4_t*>(ap + kb + j * 16);
        const skw4_t* bp = reinterpret_cast<const skw4_t*>(Bh + (((size_t)blockIdx.x * units + (kb >> 6)) * 64 + lane) * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) f.w[j] = bp[j];       // word e of w[j]: {column n0 + col, column n0 + 16 + col} at k = kb + 16j + 4rg + e
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.a[j] = *reinterpret_cast<const f32x4_t*>(ap + kb + j * 16);     // A[row][kb + 16j + 4rg + e]
#pragma unroll
      for (int n = 0; n < NT; ++n) {                                    // B[kb + 16j + 4rg + e][n0 + 16n + col]
        if (TRANSB) {
          f.b[n][j] = *reinterpret_cast<const f32x4_t*>(Bm + (size_t)(n0 + n * 16 + col) * ldb + kb + j * 16 + rg * 4);
        } else {
          const float* bp = Bm + (size_t)(kb + j * 16 + rg * 4) * ldb + n0 + n * 16 + col;
#pragma unroll
          for (int e = 0; e < 4; ++e) f.b[n][j][e] = bp[(size_t)e * ldb];
        }
      }
    }
  };
  auto mma = [&](const Frag& f) {
    if constexpr (BH) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (TRANSB) {                           // multiply t = 4j + e: k = kb + 16rg + t
            const int t = 4 * j + e;
            const unsigned wd = f.w[t >> 3][(t & 7) >> 1];
            const float bv = __uint_as_float((t & 1) ? (wd & 0xffff0000u) : (wd << 16));
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[j][e] * f.keep, bv, acc[0], 0, 0, 0);
          } else {
            const unsigned wd = f.w[j][e];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[j][e] * f.keep, __uint_as_float(wd << 16), acc[0], 0, 0, 0);
            acc[NT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[j][e] * f.keep, __uint_as_float(wd & 0xffff0000u), acc[NT - 1], 0, 0, 0);
          }
        }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[j][e] * f.keep, f.b[n][j][e], acc[n], 0, 0, 0);
  };
  // (Round 4, measured: with all four rounds of a K = 1600 / 2048 product requested before the first multiply -- four
  // Frags, 154 / 108 registers -- the cell launch takes 12.7 us against 11.4 and the 8-unit query product 5.9 against 4.9:
  // the launch is not a chain of weight trips, it is the fp32 matrix rate (128 16x16x4 multiplies of 32 cycles per wave,
  // two waves per SIMD: 3.4 us) plus launch and epilogue; bf16 weights alone therefore buy only ~1 %.)
  // (Round 5, measured: the activations split by the loading lane into hi + lo bf16 terms and multiplied as two
  // v_mfma_f32_16x16x32_bf16 per 32-wide chunk -- 8 x fewer matrix cycles, VERDICT r04 item 4(i) -- leaves both launches where
  // they were: cell product 11.3 -> 11.1 us, backward product 8.6 -> 8.6, cfg D step 69.0 -> 68.9 ms.  The fp32 multiplies
  // were already hidden under the weight trips: the attention kernels stream 69 MB between two steps, so the 6.5 MB of
  // W_cell come from the memory side every step and a launch is launch + two to four dependent ~2 us trips + the
  // eight-wave reduction.  Not kept: the exact products cost nothing.)
  Frag f0, f1;
  load(0, f0);
  if (units <= SK_WAVES) {                  // one round: nothing to overlap
    mma(f0);
  } else {
    const int rounds = ((units + SK_WAVES - 1) / SK_WAVES + 1) & ~1;
    for (int r = 0; r < rounds; r += 2) {   // the loads of one round are in flight under the MFMAs of the other
      load(r + 1, f1);
      __builtin_amdgcn_sched_barrier(0);
      mma(f0);
      __builtin_amdgcn_sched_barrier(0);
      load(r + 2, f0);
      __builtin_amdgcn_sched_barrier(0);
      mma(f1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // lane holds C[m0 + rg*4 + r][n0 + n*16 + col]
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][rg * 4 + r][n * 16 + col] = acc[n][r];
  __syncthreads();
  if constexpr (CELL) {
    if (!cl_on) return;
    const int m = (int)threadIdx.x / (TN / 4), ul = (int)threadIdx.x % (TN / 4), U = cell.U;
    float p[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {                          // the plain epilogue's sum, in its order
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += red[w][m][ul * 4 + g];
      if (bias) v += bias[n0 + ul * 4 + g];
      p[g] = v;
    }
    // cell_fwd_kernel (attention.hip), expression for expression
    const float cp = cl_cp;
    const float i = sk_sigf(p[0] + cl_wci * cp);
    const float g = tanhf(p[1]);
    const float f = sk_sigf(p[2] + cell.fb + cl_wcf * cp);
    float cn = g * i + cp * f;
    if (cell.clip > 0.f) cn = fminf(fmaxf(cn, -cell.clip), cell.clip);
    const float o = sk_sigf(p[3] + cl_wco * cn);
    const float hn = tanhf(cn) * o;
    const size_t ci = (size_t)cl_m * U + cl_u;
    float* gp = cell.gates + (size_t)cl_m * 4 * U;
    gp[cl_u] = i; gp[U + cl_u] = g; gp[2 * U + cl_u] = f; gp[3 * U + cl_u] = o;
    cell.c_raw[ci] = cn;
    cell.h_raw[ci] = hn;
    cell.c_out[ci] = cl_lv > 0.f ? cn : cp;
    const float ho = cl_lv > 0.f ? hn : cl_hp;
    cell.h_out[ci] = ho;
    if (cell.h_out2) cell.h_out2[(size_t)cl_m * cell.ld_h2 + cl_u] = ho;
    const float co = cell.out_mask ? hn * cl_om : hn;
    if (cell.cell_out) cell.cell_out[ci] = co;
    if (cell.cell_out2) cell.cell_out2[(size_t)cl_m * cell.ld_c2 + cl_u] = co;
    return;
  }
  for (int idx = threadIdx.x; idx < 16 * TN; idx += 64 * SK_WAVES) {
    const int m = idx / TN, n = idx % TN;
    if (m0 + m >= M) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) v += red[w][m][n];
    if (bias) v += bias[n0 + n];
    float* cp = C + (size_t)(m0 + m) * ldc + n0 + n;
    if (accumulate) v += *cp;
    if (act == 1) v = fmaxf(v, 0.f);
    *cp = v;
  }
}

static bool try_gemm_skinny_f32(int transA, int transB, int M, int N, int K, const void* A, int lda, const void* B,
                                int ldb, void* C, int ldc, const float* bias, int accumulate, hipStream_t st, int act) {
  if (transA || M < 1 || M > 32 || K % 64 != 0 || N % 16 != 0 || K < 64) return false;
  if (lda % 4 != 0 || ((uintptr_t)A) % 16 != 0 || ((uintptr_t)B) % 16 != 0) return false;
  if (transB && ldb % 4 != 0) return false;
  // full 128-byte lines of a row-major B need 32 columns per workgroup; B^T rows are contiguous in k, so the narrow
  // tile (twice the workgroups) is used until the launch is wide enough anyway
  const bool wide = (N % 32 == 0) && (!transB || N >= 32 * 192);
  const unsigned gy = (unsigned)((M + 15) / 16);
#define ASR_SKINNY(TB, NT_)                                                                                             \
  hipLaunchKernelGGL((gemm_skinny_f32_kernel<TB, NT_>), dim3(N / (16 * NT_), gy), dim3(64 * SK_WAVES), 0, st, M, N, K, \
                     (const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, bias, accumulate, act, SkinnyCell{})
  if (transB) {
    if (wide) ASR_SKINNY(true, 2); else ASR_SKINNY(true, 1);
  } else {
    if (wide) ASR_SKINNY(false, 2); else ASR_SKINNY(false, 1);
  }
#undef ASR_SKINNY
  return true;
}

// ---------------------------------------------------------------- lean TN kernel (bf16)
// C[M,N] = A^T B with A [K,M] and B [K,N] both REDUCTION-MAJOR (row = one k): the weight-gradient
// products X^T dG with K = T*B.  The MFMA fragments want 8 consecutive k per lane while memory has 8 consecutive m.
// gfx950 transposes on the LDS READ side: the tiles are copied into LDS as they come (one 16-byte vector = 8 columns
// of a k row per ds_write_b128), laid out as 16-column subtiles [128/16][64 k][16] so that the 4 x 16 block a
// 16-lane group needs is 128 contiguous bytes, and a fragment is two ds_read_b64_tr_b16 (k 0-3 and 4-7 of the lane's
// eight): each lane passes the address of one 8-byte piece of the block and receives its COLUMN.  Subtiles are
// 2080 bytes apart (2048 + 32) so the sixteen vectors of a k row land in sixteen different bank groups.
// (The first form of this kernel transposed on the WRITE side -- sixteen 32-bit LDS stores per thread and tile into
// a swizzled [m][k] image -- and ran the 5 x 512 weight gradients at ~200 TFLOP/s.)
// Split-K over slabs (summed in a fixed order by splitk_reduce_kernel), 128x128x64 tiles, double-buffered LDS.
// Requires M % 8 == 0, N % 8 == 0, lda/ldb % 8 == 0, 16-byte aligned bases.
constexpr int TN_SUB = 2048 + 32;                 // bytes from one 16-column subtile [64 k][16] to the next
constexpr int TN_OPER = 8 * TN_SUB;               // one operand tile: 128 columns
constexpr int TN_STAGE = 2 * TN_OPER;             // A tile | B tile
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
__device__ __forceinline__ bf16x8_t tn_frag(const char* p) {
  typedef __attribute__((address_space(3))) bf16x4_t lds4_t;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t*)(p));
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t*)(p + 128));
  return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                           int lda, const bf16_t* __restrict__ Bm, int ldb,
                                                           int kchunk, float* __restrict__ partial, int tn, int tm,
                                                           int nslab, int xcd_skip) {
  constexpr int BM = 128, BN = 128, BK = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // 1-D grid; workgroup b runs on XCD b % 8 and the first xcd_skip XCDs are left to the recurrence clusters
  const int widx = xcd_work_index(xcd_skip);
  if (widx < 0 || widx >= tn * tm * nslab) return;
  const int bx = widx % tn, by = (widx / tn) % tm, bz = widx / (tn * tm);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * kchunk, kend = min(K, kbeg + kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  // four (k row, 8-column vector) items per operand per thread: vector mvec = tid & 15 of rows (tid >> 4) + 16 q
  const int mvec = tid & 15, kr0 = tid >> 4;
  const bool a_ok = m0 + mvec * 8 + 8 <= M, b_ok = n0 + mvec * 8 + 8 <= N;
  const bf16_t* pa = A + (size_t)(kbeg + kr0) * lda + m0 + mvec * 8;
  const bf16_t* pb = Bm + (size_t)(kbeg + kr0) * ldb + n0 + mvec * 8;
  const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8_t ra[4], rb[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kbeg + kt * BK + kr0 + 16 * q;
      const size_t ro = (size_t)(kt * BK + 16 * q);
      ra[q] = (a_ok && k < kend) ? *reinterpret_cast<const bf16x8_t*>(pa + ro * lda) : zero;
      rb[q] = (b_ok && k < kend) ? *reinterpret_cast<const bf16x8_t*>(pb + ro * ldb) : zero;
    }
  };
  // image: subtile (mvec >> 1), row k (32 bytes), half (mvec & 1)
  const unsigned wbase = (unsigned)(mvec >> 1) * TN_SUB + (unsigned)kr0 * 32u + (unsigned)(mvec & 1) * 16u;
  auto sstore = [&](char* st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<bf16x8_t*>(st + wbase + q * 16 * 32) = ra[q];
      *reinterpret_cast<bf16x8_t*>(st + TN_OPER + wbase + q * 16 * 32) = rb[q];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (nkt > 0) {
    gload(0);
    sstore(smem);
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  // lane's 8-byte piece of a 16-lane group's [4 k][16 col] block: row (fr >> 2), columns 4 (fr & 3) .. + 3; the
  // group fq takes k rows 8 fq .. 8 fq + 7 of the 32-row k-block (two reads, 4 rows = 128 bytes apart)
  const unsigned piece = (unsigned)(8 * fq + (fr >> 2)) * 32u + (unsigned)(fr & 3) * 8u;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = (unsigned)(wm * 4 + i) * TN_SUB + piece;
    boff[i] = (unsigned)TN_OPER + (unsigned)(wn * 4 + i) * TN_SUB + piece;
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const char* cur = smem + (kt & 1) * TN_STAGE;
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tn_frag(cur + aoff[i] + ks * 32 * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = tn_frag(cur + boff[j] + ks * 32 * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(smem + ((kt + 1) & 1) * TN_STAGE);
    __syncthreads();
  }
  // slab [z][M][N]; lane holds C[m = fr][n = fq*4 .. +3] of each 16x16 tile
  float* slab = partial + (size_t)bz * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 64 + j * 16 + fq * 4;
      if (nb >= N) continue;                               // N % 8 == 0: a 4-group is all in or all out
      *reinterpret_cast<f32x4_t*>(slab + (size_t)m * N + nb) = acc[i][j];
    }
  }
}

struct ConvGate {               // epilogue operands of conv3x3_nt_bf16_kernel's act == 2 / act == 3
  const bf16_t* act;            // act == 2: ReLU output of the layer below, [pixels, Cout of this product]
  float keep;
  uint64_t seed, offset;        // dropout applied to that output (element e -> Philox block offset + e / 4)
  int use_drop;                 // act == 2: 1 = form the mask (Philox), 2 = `act` is the DROPPED output: it is > 0 exactly
};                              //           where the unit was active AND kept, the gradient is scaled by 1 / keep there
// act == 3 (forward): ReLU, round to the operand dtype, then tf.nn.dropout (mask from keep / seed / offset) -- the stored
// activation is the dropped one, bit for bit what asr_dropout_apply makes of the stored ReLU output, and the undropped
// one is never written (the backward needs only its sign where the mask kept it: gate mode 2)
typedef __attribute__((ext_vector_type(4))) unsigned short cg_us4_t;
// (the gate operand of act == 2 as a separate load: the image-resident kernel requests it at the top of a pixel tile, a
// thousand matrix cycles ahead of the epilogue that consumes it)
__device__ __forceinline__ cg_us4_t conv_gate_load(const ConvGate& gate, size_t e) {
  return *reinterpret_cast<const cg_us4_t*>(gate.act + e);
}
template <bool PRE = false>
__device__ __forceinline__ void conv_gate_apply(int act, const ConvGate& gate, size_t e, float (&v)[4],
                                                cg_us4_t pre = cg_us4_t{0, 0, 0, 0}) {
  typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
  if (act == 2) {
    const us4_t g = PRE ? pre : *reinterpret_cast<const us4_t*>(gate.act + e);
    float mk[4] = {1.f, 1.f, 1.f, 1.f};
    if (gate.use_drop == 1) asr_dropout_words(gate.offset + e / 4, gate.seed, gate.keep, 1.f / gate.keep, mk);
    else if (gate.use_drop == 2) { const float inv = 1.f / gate.keep; mk[0] = mk[1] = mk[2] = mk[3] = inv; }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = bf16_to_f32(g[r]) > 0.f ? v[r] * mk[r] : 0.f;
  } else if (act == 3) {
    float mk[4];
    asr_dropout_words(gate.offset + e / 4, gate.seed, gate.keep, 1.f / gate.keep, mk);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = bf16_to_f32(f32_to_bf16(fmaxf(v[r], 0.f))) * mk[r];
  }
}

// ---------------------------------------------------------------- implicit-GEMM 3x3 convolution (bf16)
// out[p, co] = act(sum_{tap, ci} x[p + s_tap, ci] * Wt[co][tap*Cin + ci] + bias[co]),  SAME padding,
// x NHWC [Nimg, H, W, Cin], p = flat pixel index, s_tap = (tap/3 - 1, tap%3 - 1).
// This is gemm_nt_bf16_kernel with the A tile gathered straight from the image: a 64-wide k-tile is
// (one tap, 64 consecutive input channels) = 128 contiguous bytes of the shifted pixel, zero when the
// shifted pixel falls off the frame -- no im2col patch matrix (9x the activation bytes) is ever written.
// The data gradient is the same kernel on dOut with the flipped-tap weight image (conv3x3_prep_kernel).
// Requires Cin % 64 == 0; BN in {64, 128} = Cout tile.
template <typename TO, int BN>
__global__ __launch_bounds__(256) void conv3x3_nt_bf16_kernel(int Mpix, int H, int W, int Cin, int Cout,
                                                              const bf16_t* __restrict__ X,
                                                              const bf16_t* __restrict__ Wt,
                                                              TO* __restrict__ Out, const float* __restrict__ bias,
                                                              int act, ConvGate gate) {
  // act == 2 (data gradient): the ReLU backward of the layer BELOW in the epilogue -- Out (operand dtype) =
  // (gate.act[p, c] > 0) ? value * dropout mask(element) : 0, instead of an fp32 gradient that asr_relu_bwd(_drop)
  // would read back
  constexpr int BM = 128, BK = 64, LD = BK + 8;
  constexpr int STAGE = (BM + BN) * LD;
  constexpr int WN = BN / 2, TN = WN / 16;
  constexpr int NB = BN * 8 / 256;                         // B vectors per thread per k-tile (4 / 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* S = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = Cout / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const int K = 9 * Cin, nkt = K / BK, kpt = Cin / BK;     // k-tiles per tap
  const int HW = H * W;

  // A: 4 vectors per thread: pixel row r = v >> 3 (tile-local), channel offset (v & 7) * 8
  int py[4], px[4];
  const bf16_t* pa[4];
  unsigned so[4];
  bool mok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = tid + i * 256, r = v >> 3, kv = (v & 7) * 8;
    const int m = m0 + r;
    mok[i] = m < Mpix;
    const int mm = mok[i] ? m : 0;
    const int rem = mm % HW;
    py[i] = rem / W;
    px[i] = rem % W;
    pa[i] = X + (size_t)mm * Cin + kv;
    so[i] = (unsigned)(r * LD + kv);
  }
  const bf16_t* pb[NB];
  unsigned sob[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int v = tid + i * 256, r = v >> 3, kv = (v & 7) * 8;
    pb[i] = Wt + (size_t)(n0 + r) * K + kv;
    sob[i] = (unsigned)(BM * LD + r * LD + kv);
  }
  const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8_t ra[4], rb[NB];
  auto gload = [&](int kt) {
    const int tap = kt / kpt, ci0 = (kt - tap * kpt) * BK;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const ptrdiff_t sh = ((ptrdiff_t)dy * W + dx) * Cin + ci0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = mok[i] && (unsigned)(py[i] + dy) < (unsigned)H && (unsigned)(px[i] + dx) < (unsigned)W;
      ra[i] = ok ? *reinterpret_cast<const bf16x8_t*>(pa[i] + sh) : zero;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const bf16x8_t*>(pb[i] + (size_t)kt * BK);
  };
  auto sstore = [&](bf16_t* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16x8_t*>(st + so[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<bf16x8_t*>(st + sob[i]) = rb[i];
  };

  f32x4_t acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  gload(0);
  sstore(S);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned aoff = (unsigned)((wm * 64 + fr) * LD + fq * 8);
  const unsigned boff = (unsigned)(BM * LD + (wn * WN + fr) * LD + fq * 8);
  for (int kt = 0; kt < nkt; ++kt) {
    const bf16_t* cur = S + (kt & 1) * STAGE;
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4], b[TN];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(cur + aoff + i * 16 * LD + ks * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(cur + boff + j * 16 * LD + ks * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(S + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= Mpix) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + wn * WN + j * 16 + fq * 4;
      TO* cp = Out + (size_t)m * Cout + nb;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (bias) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + nb);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
      }
      if (act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (act >= 2) conv_gate_apply(act, gate, (size_t)m * Cout + nb, v);
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<f32x4_t*>(cp) = (f32x4_t){v[0], v[1], v[2], v[3]};
      } else {
        typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
        *reinterpret_cast<us4_t*>(cp) = (us4_t){f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])};
      }
    }
  }
}

// ---------------------------------------------------------------- 3x3 convolution, image-resident form (bf16)
// The per-frame images of the VGG front-end are small (40 x 11 x 64 ch = 56 KB, 20 x 6 x 128 ch = 31 KB), so a whole
// image WITH its zero border fits the LDS of one CU.  conv3x3_nt_bf16_kernel above re-gathers every shifted pixel row from
// L2 for each of the nine taps and is bound by its CU's L2 port and by re-staging A through LDS (330 - 540 TFLOP/s on
// these shapes); here an image is staged ONCE (coalesced 16-byte copies, the next image's loads in flight under this
// image's products), the A fragments of all nine taps are 16-byte LDS reads at constant offsets from the centre pixel,
// and the weights -- the B operand -- sit in REGISTERS for the whole launch: a wave owns NTW 16-channel output tiles x all
// of K (288 VGPRs; one wave per SIMD, 512-entry register file), so nothing but A fragments moves per MFMA.
//   CIN = 64:  NTW = 4 -> a wave covers 64 output channels; waves split the image's 16-pixel tiles
//   CIN = 128: NTW = 2 -> waves split the output channels (and the pixel tiles when COUT = 64)
// One workgroup walks images blockIdx.x, + gridDim.x, ...  Same operands / epilogues / results as the kernel above
// (out[p, co] = act(sum_{tap, ci} x[p + s_tap, ci] Wt[co][tap * CIN + ci] + bias[co])).
// phase timers of the image loop (ASR_CONV_DBG=1; scripts/probe_conv_phases.py): per workgroup and wave, cycles summed over
// its images: [0] tile loop (of which [1] multiplies incl. their LDS reads, [2] epilogues), [3] wait for + LDS store of the
// next image, [4] the image's closing barrier, [5] images, [6] prefetch issue
__device__ unsigned long long* g_convdbg = nullptr;
// MAXV = staged 16-byte vectors per thread: ceil(H W CIN / 8 / 256) rounded up to an instantiated value (4 / 8 / 14 / 16) --
// the staging registers are what the CIN = 64 forms are short of (288 weight registers)
// ACT (compile time since round 5: a run-time `act` put branches into every epilogue piece, and a piece has to be straight-line
// code to be scheduled between the multiplies): 0 none, 1 ReLU, 2 data gradient gated by the sign of gate.act with a uniform
// scale (use_drop 0: 1, use_drop 2: 1 / keep), 4 the same with the Philox mask (use_drop 1), 3 forward ReLU + dropout.
// NW = waves per workgroup: 4 (one per SIMD) or, CIN = 64 only, 8 (two per SIMD, each wave two 16-channel output tiles = 144
// weight registers: the sibling wave fills the LDS latency and the epilogue's VALU work; round 5)
// STREAM (round 5, two LDS images): the next image is not held in MAXV staging registers for the whole image and written to LDS in
// one phase at its end (14 loads issued at once: 1.7 k cycles of issue stall, then 0.9 k cycles of LDS stores with nothing
// beside them, per 22.5 k-cycle image at 40 x 11 x 64) but STREAMED: every pixel tile requests two vectors at its top and stores
// the two of the tile before into the other image buffer -- eight staging registers instead of 16 - 64.
template <typename TO, int CIN, int COUT, int MAXV = 16, int ACT = 1, bool DBG = false, int NW = 4, bool STREAM = false>
__global__ __launch_bounds__(NW * 64, 1) void conv3x3_img_kernel(int Nimg, int H, int W, const bf16_t* __restrict__ X,
                                                             const bf16_t* __restrict__ Wt, TO* __restrict__ Out,
                                                             const float* __restrict__ bias, ConvGate gate,
                                                             int nbuf) {
  constexpr int KS = 9 * CIN / 32;                         // k-steps of 32
  constexpr int KPT = CIN / 32;                            // k-steps per tap
  constexpr int NTW = (CIN == 64 && NW == 4) ? 4 : 2;      // output tiles per wave
  constexpr int NTHR = NW * 64;
  constexpr int NGROUPS = (COUT / 16) / NTW;               // wave groups over the output channels
  constexpr int MPARTS = NW / NGROUPS;                     // waves sharing the pixel tiles of one channel group
  constexpr int PST = CIN * 2 + 16;                        // bytes per pixel in LDS (16-byte pad: conflict-free b128 reads)
  // the deferred epilogue keeps a second set of accumulators + gate operands alive: CIN = 64 with 14+ staged vectors (the
  // 40 x 11 images) has no registers for it (measured with it: spills in the tile loop, 2.87 -> 4.13 ms) and keeps the
  // epilogue behind its own tile; every other form defers (20 x 6 x 64 -> 128: 1.59 -> 1.23 ms, 128 -> 128: 2.44 -> 1.90 ms)
  constexpr bool DEFER = true;
  static_assert(NGROUPS >= 1 && NGROUPS <= NW && NW % NGROUPS == 0, "wave split");
  extern __shared__ __attribute__((aligned(16))) char csm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = wave % NGROUPS, mp = wave / NGROUPS;
  const int HW = H * W, WP = W + 2;
  const int img_bytes = (H + 2) * WP * PST;
  const int nvec = HW * CIN / 8;                           // 16-byte vectors of one image
  const int fr = lane & 15, fq = lane >> 4;

  // weights -> registers: B fragment (as first MFMA operand: rows = output channel) of tile nt, k-step ks:
  // lane (channel fr, k-group fq) holds Wt[n0 + fr][ks * 32 + fq * 8 .. + 7]
  bf16x8_t breg[NTW][KS];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const bf16_t* wp = Wt + (size_t)((ng * NTW + j) * 16 + fr) * (9 * CIN) + fq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) breg[j][ks] = *reinterpret_cast<const bf16x8_t*>(wp + ks * 32);
  }
  // zero both images (borders stay zero for the whole launch)
  for (int i = tid * 16; i < nbuf * img_bytes; i += NTHR * 16) *reinterpret_cast<bf16x8_t*>(csm + i) = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();

  constexpr int SPT = 2;                                   // STREAM: vectors requested per pixel tile
  bf16x8_t stage[STREAM ? SPT : MAXV];                     // staged vectors per thread (nvec <= 256 MAXV)
  const int nvt = (nvec + NTHR - 1) / NTHR;                // vectors per thread and image
  auto gfetch = [&](int img) {
    const bf16x8_t* src = reinterpret_cast<const bf16x8_t*>(X + (size_t)img * HW * CIN);
    if constexpr (!STREAM) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int v = tid + i * NTHR;
        if (v < nvec) stage[i] = src[v];
      }
    }
  };
  const float invW = 1.0f / (float)W;
  // LDS position of vector v of an image (round 5: the quotient by the run-time W through the reciprocal -- exact for p < 2^22
  // -- instead of an integer division per staged vector: the 14 divisions were most of the 1.3 k cycles of this phase per image)
  auto vofs = [&](int v) -> int {
    const int p = v / (CIN / 8), cv = v % (CIN / 8);
    const int y = (int)(((float)p + 0.5f) * invW), x = p - y * W;
    return ((y + 1) * WP + x + 1) * PST + cv * 16;
  };
  auto lstore = [&](char* buf) {
    if constexpr (!STREAM) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int v = tid + i * NTHR;
        if (v < nvec) *reinterpret_cast<bf16x8_t*>(buf + vofs(v)) = stage[i];
      }
    }
  };
  // STREAM: vectors [i0, i0 + SPT) of this thread: request / store
  auto sfetch = [&](const bf16x8_t* src, int i0) {
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
      const int v = tid + (i0 + q) * NTHR;
      if (i0 + q < nvt && v < nvec) stage[q] = src[v];
    }
  };
  auto sstore = [&](char* buf, int i0) {
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
      const int v = tid + (i0 + q) * NTHR;
      if (i0 + q < nvt && v < nvec) *reinterpret_cast<bf16x8_t*>(buf + vofs(v)) = stage[q];
    }
  };
  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = ((t / 3 - 1) * WP + (t % 3 - 1)) * PST;
  const int ntm = (HW + 15) / 16;

  int img = blockIdx.x;
  if (img >= Nimg) return;
  if constexpr (STREAM) {                                  // the first image: straight through, once per launch
    const bf16x8_t* src0 = reinterpret_cast<const bf16x8_t*>(X + (size_t)img * HW * CIN);
    for (int i0 = 0; i0 < nvt; i0 += SPT) { sfetch(src0, i0); sstore(csm, i0); }
  } else {
    gfetch(img);
    lstore(csm);
  }
  __syncthreads();
  // Round 4: one wave per SIMD means every dependent trip of the epilogue was exposed -- the bias vector and (act == 2)
  // the gate operand were requested inside the epilogue, 4 + 4 global round trips per 16-pixel tile against 1 152 matrix
  // cycles (MfmaUtil 18 % at 64 -> 64, profiles/r03_pmc_util.md).  The bias now lives in registers for the launch, the gate
  // operand of a tile is requested at its top, and the first fragment group of the NEXT tile is read from LDS under the
  // last multiplies of this one.  Same arithmetic in the same order.
  // (round 5: the bias sits in LDS behind the images -- 16 registers the deferred epilogue needs; its read is one more
  // ds_read_b128 per piece beside the multiplies)
  float* bsm = reinterpret_cast<float*>(csm + nbuf * img_bytes);
  if (tid < COUT) bsm[tid] = bias ? bias[tid] : 0.f;
  __syncthreads();
  // taps per fragment group: CIN = 64 with the big staging set reads ONE tap (two fragments, 8 multiplies) ahead instead of
  // three -- 32 fragment registers that pay for the pending tile's accumulators
  constexpr int TG = (CIN == 64 && MAXV <= 8) ? 3 : 1, NG = 9 / TG, GF = TG * KPT;
  auto tile_ptr = [&](const char* base, int mt) -> const char* {
    const int p = mt * 16 + fr;
    const int pc = p < HW ? p : 0;
    const int y = (int)(((float)pc + 0.5f) * invW), x = pc - y * W;
    return base + ((y + 1) * WP + x + 1) * PST + fq * 16;
  };
  // Round 5 (phase timers, scripts/probe_conv_phases.py: of 24.6 k cycles per image and wave at 64 -> 64 the epilogues took
  // 5.5 k with nothing beside them -- one wave per SIMD, and the compiler does not pipeline across loop iterations): the
  // epilogue of a pixel tile is DEFERRED by one tile and issued in pieces (one 16-channel output tile each) between the
  // fragment groups of the NEXT tile's multiplies, across image boundaries too; the last tile of a workgroup is flushed
  // behind the image loop.  The pending tile's stores are raw buffer stores on a per-image descriptor: a lane whose pixel
  // lies past the image (the ragged last tile) or the very first "pending tile" of the workgroup (a zero-length
  // descriptor) is dropped by the bounds check -- no exec masking, no branch in the multiply stream.  Same arithmetic,
  // same values.
  const unsigned img_out_bytes = (unsigned)HW * COUT * (unsigned)sizeof(TO);
  f32x4_t accp[NTW];                                       // the pending tile: accumulators, gate operand, where it goes
  cg_us4_t gprep[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) { accp[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; gprep[j] = cg_us4_t{0, 0, 0, 0}; }
  __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(Out, 0, 0, 0x00020000);   // nothing pending: zero length
  unsigned offp = 0;                                       // byte offset of the pending pixel's channel 0 in its image
  size_t mp_elem = 0;                                      // element index of that pixel's channel 0 (gate / dropout counters)
  const float gscale = (ACT == 2 && gate.use_drop == 2) ? 1.f / gate.keep : 1.f;
  auto epilogue_piece = [&](int j) {
    const int nb = (ng * NTW + j) * 16 + fq * 4;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = accp[j][r];
    if constexpr (ACT == 0 || ACT == 1 || ACT == 3) {       // the forward products carry a bias (zeros in LDS without one)
      const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bsm + nb);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bv[r];
    }
    if constexpr (ACT == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    // conv_gate_apply, branch-free (same expressions in the same order)
    if constexpr (ACT == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = bf16_to_f32(gprep[j][r]) > 0.f ? v[r] * gscale : 0.f;
    }
    if constexpr (ACT == 4) {
      float mk[4];
      asr_dropout_words(gate.offset + (mp_elem + nb) / 4, gate.seed, gate.keep, 1.f / gate.keep, mk);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = bf16_to_f32(gprep[j][r]) > 0.f ? v[r] * mk[r] : 0.f;
    }
    if constexpr (ACT == 3) {
      float mk[4];
      asr_dropout_words(gate.offset + (mp_elem + nb) / 4, gate.seed, gate.keep, 1.f / gate.keep, mk);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = bf16_to_f32(f32_to_bf16(fmaxf(v[r], 0.f))) * mk[r];
    }
    typedef __attribute__((ext_vector_type(2))) unsigned cv_u2_t;
    typedef __attribute__((ext_vector_type(4))) unsigned cv_u4_t;
    if constexpr (sizeof(TO) == 4) {
      const cv_u4_t w = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      __builtin_amdgcn_raw_buffer_store_b128(w, rsp, offp + (unsigned)nb * 4u, 0, 0);
    } else {
      const cv_u2_t w = {(unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16),
                         (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16)};
      __builtin_amdgcn_raw_buffer_store_b64(w, rsp, offp + (unsigned)nb * 2u, 0, 0);
    }
  };
  unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
#define CV_T() (DBG ? (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime()) : 0ull)
  for (int it = 0; img < Nimg; img += gridDim.x, ++it) {
    char* cur = csm + (nbuf == 2 ? (it & 1) * img_bytes : 0);
    const int nxt = img + gridDim.x;
    const unsigned long long tq0 = CV_T();
    if (nxt < Nimg) gfetch(nxt);                           // lands under this image's products
    const unsigned long long tq1 = CV_T();
    const bool morei = nxt < Nimg;                         // block-uniform
    const bf16x8_t* srcn = reinterpret_cast<const bf16x8_t*>(X + (size_t)(morei ? nxt : img) * HW * CIN);
    char* bufn = csm + ((it + 1) & 1) * img_bytes;         // STREAM implies two buffers
    int sq = 0;                                            // STREAM: this thread's next vector index
    const __amdgpu_buffer_rsrc_t rsc =
        __builtin_amdgcn_make_buffer_rsrc(Out + (size_t)img * HW * COUT, 0, img_out_bytes, 0x00020000);
    // (CIN = 64 keeps 288 weight registers + 64 staging registers: the 24 of the look-ahead group would spill)
    constexpr bool NEXTPF = CIN == 128;
    bf16x8_t anx[NEXTPF ? GF : 1];                         // group 0 of the tile about to start
    if (NEXTPF && mp < ntm) {
      const char* ap0 = tile_ptr(cur, mp);
#pragma unroll
      for (int q = 0; q < GF; ++q) anx[q] = *reinterpret_cast<const bf16x8_t*>(ap0 + tapoff[q / KPT] + (q % KPT) * 64);
    }
    // Two tiles per loop trip where the registers allow it (the pending / current accumulator sets then swap roles without
    // copies and the scheduler sees both tiles: 20 x 6 x 64 -> 128 ReLU 1.48 -> 1.21 ms, 128 -> 128 2.30 -> 2.04 ms); the gated
    // data gradient (ACT 2: two sets of gate operands) and the 40 x 11 x 64 form spill 13 - 31 registers that way and stay at one
    constexpr int UNR = (ACT >= 2 || MAXV > 8) ? 1 : 2;        // (the Philox epilogues measured slower unrolled: 1.96 -> 2.03 ms)
#pragma unroll UNR
    for (int mt = mp; mt < ntm; mt += MPARTS) {
      if constexpr (STREAM) {
        if (morei) {
          if (sq > 0) sstore(bufn, sq - SPT);              // what the tile before requested has landed (a tile of multiplies ago)
          sfetch(srcn, sq);
          sq += SPT;
        }
      }
      const int p = mt * 16 + fr;
      const char* ap = tile_ptr(cur, mt);
      const bool more = mt + MPARTS < ntm;                 // wave-uniform
      const char* apn = tile_ptr(cur, more ? mt + MPARTS : mt);
      const size_t m = (size_t)img * HW + (p < HW ? p : 0);
      cg_us4_t gpre[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) gpre[j] = cg_us4_t{0, 0, 0, 0};
      if constexpr (ACT == 2 || ACT == 4) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) gpre[j] = conv_gate_load(gate, m * COUT + (ng * NTW + j) * 16 + fq * 4);
      }
      const unsigned long long tt0 = CV_T();
      f32x4_t acc[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      // A fragments in groups of TG taps, the next group's LDS reads issued ahead of this group's MFMAs (one wave per
      // SIMD: nothing else hides the LDS latency -- read-wait-multiply per fragment ran the matrix cores at ~15 %)
      bf16x8_t a[2][GF];
#pragma unroll
      for (int q = 0; q < GF; ++q) {
        if constexpr (NEXTPF) a[0][q] = anx[q];
        else a[0][q] = *reinterpret_cast<const bf16x8_t*>(ap + tapoff[q / KPT] + (q % KPT) * 64);
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
          for (int q = 0; q < GF; ++q)
            a[(g + 1) & 1][q] = *reinterpret_cast<const bf16x8_t*>(ap + tapoff[(g + 1) * TG + q / KPT] + (q % KPT) * 64);
        } else if constexpr (NEXTPF) {
#pragma unroll
          for (int q = 0; q < GF; ++q)                      // the next tile's first group (this tile's again if it is the last)
            anx[q] = *reinterpret_cast<const bf16x8_t*>(apn + tapoff[q / KPT] + (q % KPT) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);                 // (left alone the scheduler recycles ONE register quad)
#pragma unroll
        for (int q = 0; q < GF; ++q)
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[j][g * GF + q], a[g & 1][q], acc[j], 0, 0, 0);
        // the PENDING tile's epilogue, one output tile per piece, beside this group's multiplies
        if constexpr (DEFER) {
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            if ((j * NG) / NTW == g) epilogue_piece(j);
        }
      }
      // this tile becomes the pending one
#pragma unroll
      for (int j = 0; j < NTW; ++j) { accp[j] = acc[j]; gprep[j] = gpre[j]; }
      rsp = rsc;
      offp = (unsigned)p * COUT * (unsigned)sizeof(TO);    // p >= HW: past the descriptor's length, the stores are dropped
      mp_elem = m * COUT;
      if constexpr (!DEFER) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) epilogue_piece(j);
      }
      if (DBG) {
        const unsigned long long tt2 = CV_T();
        ph[1] += tt2 - tt0;
      }
    }
    // (Measured, round 4: barriers that wait for the LDS counter only -- s_waitcnt lgkmcnt(0) + s_barrier instead of
    // __syncthreads(), whose release fence also waits for the epilogue's global stores -- change nothing: 2.80 vs 2.76 ms.)
    const unsigned long long tq2 = CV_T();
    if constexpr (STREAM) {
      if (morei) {
        if (sq > 0) sstore(bufn, sq - SPT);
        for (; sq < nvt; sq += SPT) { sfetch(srcn, sq); sstore(bufn, sq); }   // more vectors than tiles x SPT: the rest, exposed
      }
    } else if (nxt < Nimg) {
      if (nbuf == 1) __syncthreads();                      // every wave is done with the only buffer
      lstore(csm + (nbuf == 2 ? ((it + 1) & 1) * img_bytes : 0));
    }
    const unsigned long long tq3 = CV_T();
    __syncthreads();
    if (DBG) {
      const unsigned long long tq4 = CV_T();
      ph[0] += tq2 - tq1; ph[3] += tq3 - tq2; ph[4] += tq4 - tq3; ph[5] += 1; ph[6] += tq1 - tq0;
    }
  }
  // flush: the last tile of this workgroup
  if constexpr (DEFER) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) epilogue_piece(j);
  }
  if (DBG && g_convdbg && lane == 0 && blockIdx.x < 64) {
    unsigned long long* o = g_convdbg + ((size_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
    for (int k = 0; k < 7; ++k) o[k] = ph[k];
  }
#undef CV_T
}

// weight images for the two implicit GEMMs from the HWIO fp32 master [9][Cin][Cout]:
//   wf[co][tap*Cin + ci] = w[tap][ci][co]          (forward:   B^T of x * W)
//   wb[ci][tap*Cout + co] = w[8 - tap][ci][co]     (data grad: B^T of dOut * flipped W)
__global__ void conv3x3_prep_kernel(const float* __restrict__ w, int Cin, int Cout, bf16_t* __restrict__ wf,
                                    bf16_t* __restrict__ wb) {
  const size_t total = (size_t)9 * Cin * Cout;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = i % Cout, ci = (i / Cout) % Cin, tap = i / ((size_t)Cin * Cout);
    const bf16_t v = f32_to_bf16(w[i]);
    wf[(size_t)co * 9 * Cin + (size_t)tap * Cin + ci] = v;
    wb[(size_t)ci * 9 * Cout + (size_t)(8 - tap) * Cout + co] = v;
  }
}

// Weight gradient: dW[tap*Cin + ci][co] = sum_p x[p + s_tap, ci] * dOut[p, co].  gemm_tn_bf16_kernel with the
// A operand (reduction index = pixel, column = (tap, ci)) gathered from the image; B = dOut is plain.
__global__ __launch_bounds__(256) void conv3x3_wgrad_tn_kernel(int Mpix, int H, int W, int Cin, int Cout,
                                                               const bf16_t* __restrict__ X,
                                                               const bf16_t* __restrict__ dY, int kchunk,
                                                               float* __restrict__ partial) {
  constexpr int BM = 128, BN = 128, BK = 64, LD = BK + 8;
  constexpr int STAGE = (BM + BN) * LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* S = reinterpret_cast<bf16_t*>(smem);
  const int M = 9 * Cin, N = Cout, K = Mpix, HW = H * W;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  const int mvec = tid & 15;
  const int kp0 = tid >> 4;
  const int mcol = m0 + mvec * 8;                          // first of this thread's 8 virtual columns
  const bool a_ok = mcol + 8 <= M, b_ok = n0 + mvec * 8 + 8 <= N;
  const int tap = a_ok ? mcol / Cin : 0, ci = a_ok ? mcol - tap * Cin : 0;
  const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
  const ptrdiff_t sh = ((ptrdiff_t)dy * W + dx) * Cin + ci;
  const float invW = 1.0f / (float)W;
  const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8_t ra[2][2], rb[2][2];
  auto gload = [&](int kt) {
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int p = kbeg + kt * BK + 2 * (kp0 + 16 * it) + h;              // pixel = reduction index
        const bool kin = p < kend;
        const int pp = kin ? p : 0;
        const int rem = pp % HW;
        const int y = (int)(((float)rem + 0.5f) * invW), x = rem - y * W;    // exact for rem < 2^22
        const bool ok = a_ok && kin && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
        ra[it][h] = ok ? *reinterpret_cast<const bf16x8_t*>(X + (ptrdiff_t)((size_t)pp * Cin) + sh) : zero;
        rb[it][h] = (b_ok && kin) ? *reinterpret_cast<const bf16x8_t*>(dY + (size_t)pp * Cout + n0 + mvec * 8) : zero;
      }
  };
  const unsigned sw = (unsigned)(mvec & 7);
  auto sstore = [&](bf16_t* st) {
    char* base = reinterpret_cast<char*>(st);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const unsigned k2 = (unsigned)(2 * (kp0 + 16 * it));
      const unsigned inrow = (((k2 >> 3) ^ sw) << 4) + (k2 & 7u) * 2u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned off = (unsigned)(mvec * 8 + j) * LD * 2u + inrow;
        *reinterpret_cast<unsigned*>(base + off) =
            (unsigned)(unsigned short)ra[it][0][j] | ((unsigned)(unsigned short)ra[it][1][j] << 16);
        *reinterpret_cast<unsigned*>(base + BM * LD * 2 + off) =
            (unsigned)(unsigned short)rb[it][0][j] | ((unsigned)(unsigned short)rb[it][1][j] << 16);
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if (nkt > 0) {
    gload(0);
    sstore(S);
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  unsigned arow[4], brow[4], au[4], bu[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned ra_ = (unsigned)(wm * 64 + i * 16 + fr), rb_ = (unsigned)(wn * 64 + i * 16 + fr);
    arow[i] = ra_ * LD * 2u;
    brow[i] = (unsigned)BM * LD * 2u + rb_ * LD * 2u;
    au[i] = (unsigned)fq ^ ((ra_ >> 3) & 7u);
    bu[i] = (unsigned)fq ^ ((rb_ >> 3) & 7u);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const char* cur = reinterpret_cast<const char*>(S + (kt & 1) * STAGE);
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(cur + arow[i] + ((au[i] ^ (unsigned)(ks * 4)) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(cur + brow[j] + ((bu[j] ^ (unsigned)(ks * 4)) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(S + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }
  float* slab = partial + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 64 + j * 16 + fq * 4;
      if (nb >= N) continue;
      *reinterpret_cast<f32x4_t*>(slab + (size_t)m * N + nb) = acc[i][j];
    }
  }
}

// The same weight gradient on gemm_tn_bf16_kernel's LDS image: a thread's gathered 16 bytes (8 channels of one tap at one
// pixel) are ONE ds_write_b128 into [subtile][k row][16 columns], and the MFMA fragments come out of it through the
// transposing reads (ds_read_b64_tr_b16) -- the kernel above interleaves pixel pairs into a [column][k] image with 64
// four-byte LDS writes per thread per k-tile (twice the instructions of its 32 MFMAs).  ASR_CONV_WGRAD_TR=0 keeps it (A/B).
template <int BN>
__global__ __launch_bounds__(256) void conv3x3_wgrad_tr_kernel(int Mpix, int H, int W, int Cin, int Cout,
                                                               const bf16_t* __restrict__ X,
                                                               const bf16_t* __restrict__ dY, int kchunk,
                                                               float* __restrict__ partial) {
  constexpr int BM = 128, BK = 64;
  constexpr int WN = BN / 64, WM = 4 / WN, TI = BM / WM / 16;   // BN = 64: four waves of 32 x 64, BN = 128: 2 x 2 of 64 x 64
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int M = 9 * Cin, N = Cout, K = Mpix, HW = H * W;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  // four (pixel, 8-column vector) items per operand per thread: vector mvec = tid & 15 of pixels (tid >> 4) + 16 q
  const int mvec = tid & 15, kr0 = tid >> 4;
  const int mcol = m0 + mvec * 8;                          // first of this thread's 8 virtual columns (tap, ci)
  const bool a_ok = mcol + 8 <= M, b_ok = mvec * 8 < BN && n0 + mvec * 8 + 8 <= N;
  const int tap = a_ok ? mcol / Cin : 0, ci = a_ok ? mcol - tap * Cin : 0;
  const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
  const ptrdiff_t sh = ((ptrdiff_t)dy * W + dx) * Cin + ci;
  const float invW = 1.0f / (float)W;
  const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8_t ra[4], rb[4];
  // position of this thread's four pixels inside their images, carried from k-tile to k-tile (gload runs once per k-tile, in
  // order): the modulo by a run-time H W per pixel and k-tile -- ~60 instructions, four times -- was more VALU work than the
  // 32 MFMAs of the k-tile it feeds (MfmaUtil 27 %, profiles/r03_pmc_util.md)
  int remq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) remq[q] = (kbeg + kr0 + 16 * q) % HW;
  auto gload = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = kbeg + kt * BK + kr0 + 16 * q;         // pixel = reduction index
      const bool kin = p < kend;
      const int pp = kin ? p : 0;
      const int rem = remq[q];
      remq[q] += BK;
      while (remq[q] >= HW) remq[q] -= HW;
      const int y = (int)(((float)rem + 0.5f) * invW), x = rem - y * W;      // exact for rem < 2^22
      const bool ok = a_ok && kin && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
      ra[q] = ok ? *reinterpret_cast<const bf16x8_t*>(X + (ptrdiff_t)((size_t)pp * Cin) + sh) : zero;
      rb[q] = (b_ok && kin) ? *reinterpret_cast<const bf16x8_t*>(dY + (size_t)pp * Cout + n0 + mvec * 8) : zero;
    }
  };
  const unsigned wbase = (unsigned)(mvec >> 1) * TN_SUB + (unsigned)kr0 * 32u + (unsigned)(mvec & 1) * 16u;
  auto sstore = [&](char* st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<bf16x8_t*>(st + wbase + q * 16 * 32) = ra[q];
      if (mvec * 8 < BN) *reinterpret_cast<bf16x8_t*>(st + TN_OPER + wbase + q * 16 * 32) = rb[q];
    }
  };

  f32x4_t acc[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if (nkt > 0) {
    gload(0);
    sstore(smem);
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned piece = (unsigned)(8 * fq + (fr >> 2)) * 32u + (unsigned)(fr & 3) * 8u;
  unsigned aoff[TI], boff[4];
#pragma unroll
  for (int i = 0; i < TI; ++i) aoff[i] = (unsigned)(wm * TI + i) * TN_SUB + piece;
#pragma unroll
  for (int i = 0; i < 4; ++i) boff[i] = (unsigned)TN_OPER + (unsigned)(wn * 4 + i) * TN_SUB + piece;
  for (int kt = 0; kt < nkt; ++kt) {
    const char* cur = smem + (kt & 1) * TN_STAGE;
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[TI], b[4];
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = tn_frag(cur + aoff[i] + ks * 32 * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = tn_frag(cur + boff[j] + ks * 32 * 32);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(smem + ((kt + 1) & 1) * TN_STAGE);
    __syncthreads();
  }
  float* slab = partial + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = m0 + wm * (16 * TI) + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 64 + j * 16 + fq * 4;
      if (nb >= N) continue;
      *reinterpret_cast<f32x4_t*>(slab + (size_t)m * N + nb) = acc[i][j];
    }
  }
}

// fixed-order sum of the split-K slabs -> deterministic
template <typename TO>
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int S, int M, int N,
                                     TO* __restrict__ C, int ldc, const float* __restrict__ bias,
                                     int accumulate, int act, int xcd_skip) {
  const size_t total = (size_t)M * N;
  const int widx = xcd_work_index(xcd_skip);
  if (widx < 0) return;
  const size_t nwork = (size_t)(gridDim.x >> 3) * (8 - xcd_skip) + max(0, (int)(gridDim.x & 7) - xcd_skip);
  for (size_t i = widx * (size_t)blockDim.x + threadIdx.x; i < total; i += nwork * blockDim.x) {
    const int m = i / N, n = i % N;
    float v = bias ? bias[n] : 0.f;
    // the slabs' values are requested eight at a time and added in slab order (round 6: one dependent load per addition
    // left this pass at the memory LATENCY per slab -- ~1 ms for 128 slabs where the bytes take 25 us; same sums, same order)
    int z = 0;
    for (; z + 8 <= S; z += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(z + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; z < S; ++z) v += partial[(size_t)z * total + i];
    TO* cp = C + (size_t)m * ldc + n;
    if (accumulate) v += load_out<TO>(cp);
    if (act == 1) v = fmaxf(v, 0.f);
    store_out<TO>(cp, v);
  }
}

template <typename T, typename TO, int BM, int BN>
int launch_layout(int transA, int transB, dim3 grid, size_t lds, hipStream_t st, int M, int N, int K,
                  const T* A, int lda, const T* B, int ldb, TO* C, int ldc, const float* bias,
                  int accumulate, int aa, int ba, int kchunk, float* partial, int act) {
  // 4-column vector stores need the row starts of C aligned to 4 elements
  const int c_vec = (((uintptr_t)C) % (4 * sizeof(TO)) == 0) && (ldc % 4 == 0);
#define ASR_GEMM_LAUNCH(TA_, TB_)                                                             \
  hipLaunchKernelGGL((gemm_kernel<T, TO, BM, BN, TA_, TB_>), grid, dim3(256), lds, st, M, N, K, A, \
                     lda, B, ldb, C, ldc, bias, accumulate, aa, ba, kchunk, partial, act, c_vec)
  if (!transA && !transB) ASR_GEMM_LAUNCH(false, false);
  else if (!transA && transB) ASR_GEMM_LAUNCH(false, true);
  else if (transA && !transB) ASR_GEMM_LAUNCH(true, false);
  else ASR_GEMM_LAUNCH(true, true);
#undef ASR_GEMM_LAUNCH
  return 0;
}

template <typename T, typename TO>
int launch_gemm(asr_handle* h, int transA, int transB, int M, int N, int K, const void* A, int lda,
                const void* B, int ldb, void* C, int ldc, const float* bias, int accumulate,
                hipStream_t st, int act, const float* mul = nullptr, int ldm = 0, bool* mul_done = nullptr,
                GemmDrop drop = GemmDrop{1.f, 0, 0, 0}) {
  constexpr int VEC = GT<T>::VEC, BK = GT<T>::BK;
  if constexpr (sizeof(T) == 4 && sizeof(TO) == 4) {
    if (try_gemm_skinny_f32(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act)) return 0;
  }
  if constexpr (sizeof(T) == 2) {
    if (try_gemm_nt_bf16<TO>(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act, mul, ldm, drop)) {
      if (mul_done) *mul_done = true;
      return 0;
    }
    // reduction-major operands (X^T dG): lean TN kernel, always through split-K slabs
    if (transA && !transB && K >= 2048 && M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
        ((uintptr_t)A) % 16 == 0 && ((uintptr_t)B) % 16 == 0) {
      const int tm = (M + 127) / 128, tn = (N + 127) / 128;
      // workgroups (output tiles x split-K slabs) to aim at: ~2 per CU by default; a handle whose GEMMs run BESIDE a
      // recurrence kernel asks for few (asr_set_gemm_tn_workgroups): the slabs' write + re-read traffic is what stretches
      // the recurrence (BPTT launch 1340 / 1307 / 1285 / 1256 us with 1024 / 512 / 128 / 32 workgroups per GEMM)
      const int tn_wgs = h->tn_wgs > 0 ? h->tn_wgs : 512;
      int S = (tn_wgs + tm * tn - 1) / (tm * tn);
      const int maxS = (K + 255) / 256;
      if (S > maxS) S = maxS;
      if (S > 256) S = 256;
      while (S > 1 && (size_t)S * M * N * sizeof(float) > h->scratch_bytes - ASR_XCH_BYTES) --S;
      if ((size_t)S * M * N * sizeof(float) <= h->scratch_bytes - ASR_XCH_BYTES) {
        int kchunk = (K + S - 1) / S;
        kchunk = (kchunk + 63) / 64 * 64;
        S = (K + kchunk - 1) / kchunk;
        const size_t lds = (size_t)2 * TN_STAGE;
        static unsigned long long attr_done = 0;
        if (first_on_device(attr_done)) {
          (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        float* partial = (float*)h->scratch;
        const int skip = h->xcd_skip;
        hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(xcd_grid((long)tn * tm * S, skip)), dim3(256), lds, st, M, N, K,
                           (const bf16_t*)A, lda, (const bf16_t*)B, ldb, kchunk, partial, tn, tm, S, skip);
        const size_t total = (size_t)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel<TO>, dim3(xcd_grid(blocks, skip)), dim3(256), 0, st, partial, S, M, N,
                           (TO*)C, ldc, bias, accumulate, act, skip);
        return 0;
      }
    }
  }
  const int aa = (((uintptr_t)A) % 16 == 0) && (lda % VEC == 0);
  const int ba = (((uintptr_t)B) % 16 == 0) && (ldb % VEC == 0);
  // big tiles only when they still fill the 256 CUs
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles128 >= 320) {
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    size_t lds = (size_t)2 * (128 + 128) * (BK + VEC) * sizeof(T);
    static unsigned long long attr_done = 0;   // 72 KB of dynamic LDS needs the opt-in
    if (first_on_device(attr_done)) {
      (void)hipFuncSetAttribute((const void*)gemm_kernel<T, TO, 128, 128, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_kernel<T, TO, 128, 128, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_kernel<T, TO, 128, 128, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)gemm_kernel<T, TO, 128, 128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    return launch_layout<T, TO, 128, 128>(transA, transB, grid, lds, st, M, N, K, (const T*)A, lda,
                                          (const T*)B, ldb, (TO*)C, ldc, bias, accumulate, aa, ba, K, nullptr, act);
  }
  const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  // small output, long reduction (the dW = X^T dG products: K = T*B): split K over blockIdx.z so
  // the launch has >= ~2 workgroups per CU; slabs go to the handle's scratch and are summed in
  // a fixed order (deterministic, unlike fp32 atomics).
  int S = 1;
  if (tiles64 < 256 && K >= 16 * BK) {
    S = (int)((512 + tiles64 - 1) / tiles64);
    const int maxS = K / (4 * BK);
    if (S > maxS) S = maxS;
    if (S > 64) S = 64;
    while (S > 1 && (size_t)S * M * N * sizeof(float) > h->scratch_bytes - ASR_XCH_BYTES) --S;
  }
  size_t lds = (size_t)2 * (64 + 64) * (BK + VEC) * sizeof(T);
  if (S <= 1) {
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    return launch_layout<T, TO, 64, 64>(transA, transB, grid, lds, st, M, N, K, (const T*)A, lda,
                                        (const T*)B, ldb, (TO*)C, ldc, bias, accumulate, aa, ba, K, nullptr, act);
  }
  int kchunk = (K + S - 1) / S;
  kchunk = (kchunk + BK - 1) / BK * BK;
  S = (K + kchunk - 1) / kchunk;
  dim3 grid((N + 63) / 64, (M + 63) / 64, S);
  float* partial = (float*)h->scratch;
  launch_layout<T, TO, 64, 64>(transA, transB, grid, lds, st, M, N, K, (const T*)A, lda, (const T*)B, ldb,
                               (TO*)C, ldc, nullptr, 0, aa, ba, kchunk, partial, 0);
  const size_t total = (size_t)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel<TO>, dim3(blocks), dim3(256), 0, st, partial, S, M, N, (TO*)C, ldc,
                     bias, accumulate, act, 0);
  return 0;
}

// rows x [4, U] gate-major -> rows x [U, 4] (column 4 u + g <- column g U + u): the weight image of the CELL form of the
// skinny kernel; the bias is row `rows - 1` of the image when one is given
__global__ void interleave_gates_kernel(const float* __restrict__ W, const float* __restrict__ bias, int K, int U,
                                        float* __restrict__ out) {
  const size_t n = (size_t)(K + (bias ? 1 : 0)) * U;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t k = i / U;
    const int u = (int)(i % U);
    const float* src = k < (size_t)K ? W + k * 4 * U : bias;
    *reinterpret_cast<f32x4_t*>(out + k * 4 * U + (size_t)u * 4) = (f32x4_t){src[u], src[U + u], src[2 * U + u], src[3 * U + u]};
  }
}

// The bf16 images of a decoder cell's kernel W [K, 4U] (gate-major columns) for the BH forms of the skinny kernel:
//   frag [4U/32][K/64][64 lanes][32] bf16 -- gate-interleaved columns (4 u + g), fragment order (see the kernel): value
//        idx = (4j + e) * 2 + n of lane rg * 16 + col is W[unit * 64 + 16j + 4rg + e][column 32 nb + 16 n + col];
//   bias fp32 [4U], interleaved;   plain [K, 4U] bf16, W as it is (the backward product's B^T rows).
__global__ void cell_images_h_kernel(const float* __restrict__ W, const float* __restrict__ bias, int K, int U,
                                     bf16_t* __restrict__ frag, float* __restrict__ bias_il, bf16_t* __restrict__ plain) {
  const size_t n = (size_t)K * 4 * U;
  const int units = K / 64;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    plain[i] = f32_to_bf16(W[i]);
    // frag element i: decode (nb, unit, lane, idx)
    const int idx = (int)(i & 31), lane = (int)((i >> 5) & 63);
    const size_t bu = i >> 11;
    const int unit = (int)(bu % units), nb = (int)(bu / units);
    const int nn = idx & 1, je = idx >> 1, j = je >> 2, e = je & 3, rg = lane >> 4, col = lane & 15;
    const int k = unit * 64 + 16 * j + 4 * rg + e, c = nb * 32 + 16 * nn + col;      // interleaved column c = 4 u + g
    frag[i] = f32_to_bf16(W[(size_t)k * 4 * U + (size_t)(c & 3) * U + (c >> 2)]);
    if (i < (size_t)4 * U) bias_il[i] = bias ? bias[(i & 3) * U + (i >> 2)] : 0.f;
  }
}

}  // namespace

extern "C" int asr_gemm_act(asr_handle* h, int dtype, int out_dtype, int transA, int transB, int M,
                            int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                            int ldc, const float* bias, int accumulate, int act, asr_stream s);
extern "C" int asr_gemm(asr_handle* h, int dtype, int out_dtype, int transA, int transB, int M,
                        int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                        int ldc, const float* bias, int accumulate, asr_stream s) {
  return asr_gemm_act(h, dtype, out_dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, 0, s);
}
extern "C" int asr_gemm_act(asr_handle* h, int dtype, int out_dtype, int transA, int transB, int M,
                            int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                            int ldc, const float* bias, int accumulate, int act, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || !asr_dtype_ok(out_dtype))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_gemm: bad dtype %d/%d", dtype, out_dtype);
  if (act != 0 && act != 1) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_gemm_act: act must be 0 (none) or 1 (relu)");
  if (M < 0 || N < 0 || K < 0 || !A || !B || !C)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_gemm: bad shape/pointer M=%d N=%d K=%d", M, N, K);
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_gemm: leading dimension too small (lda=%d ldb=%d ldc=%d)",
             lda, ldb, ldc);
  if (M == 0 || N == 0) return ASR_OK;
  hipStream_t st = (hipStream_t)s;
  if (dtype == ASR_F32 && out_dtype == ASR_F32)
    launch_gemm<float, float>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act);
  else if (dtype == ASR_F32 && out_dtype == ASR_BF16)
    launch_gemm<float, bf16_t>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act);
  else if (dtype == ASR_BF16 && out_dtype == ASR_F32)
    launch_gemm<bf16_t, float>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act);
  else
    launch_gemm<bf16_t, bf16_t>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act);
  ASR_CHECK_LAUNCH(h, "asr_gemm");
  return ASR_OK;
}

namespace {
__global__ void mul_rows_kernel(float* __restrict__ C, int ldc, const float* __restrict__ mul, int ldm, int M, int N) {
  const size_t total = (size_t)M * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / N, n = i % N;
    C[m * ldc + n] *= mul[m * ldm + n];
  }
}
}  // namespace

// C = (op(A) op(B) + bias [+ C]) [relu] * mul, mul [M, N] fp32 with row stride ldmul: the multiplier is applied in the
// epilogue of the lean NT kernel (no second pass over C); other shapes run the plain GEMM and one multiply pass.
extern "C" int asr_gemm_mul(asr_handle* h, int dtype, int transA, int transB, int M, int N, int K, const void* A,
                            int lda, const void* B, int ldb, float* C, int ldc, const float* bias, int accumulate,
                            int act, const float* mul, int ldmul, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!mul) return asr_gemm_act(h, dtype, ASR_F32, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, act, s);
  if (!asr_dtype_ok(dtype) || (act != 0 && act != 1) || M < 0 || N < 0 || K < 0 || !A || !B || !C ||
      lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N || ldmul < N)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_gemm_mul: bad args M=%d N=%d K=%d lda=%d ldb=%d ldc=%d ldmul=%d", M, N, K, lda,
             ldb, ldc, ldmul);
  if (M == 0 || N == 0) return ASR_OK;
  hipStream_t st = (hipStream_t)s;
  bool fused = false;
  if (dtype == ASR_F32)
    launch_gemm<float, float>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act, mul, ldmul, &fused);
  else
    launch_gemm<bf16_t, float>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act, mul, ldmul, &fused);
  ASR_CHECK_LAUNCH(h, "asr_gemm_mul");
  if (!fused) {
    const size_t total = (size_t)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mul_rows_kernel, dim3(blocks), dim3(256), 0, st, C, ldc, mul, ldmul, M, N);
    ASR_CHECK_LAUNCH(h, "asr_gemm_mul(multiply)");
  }
  return ASR_OK;
}

// C (fp32, CONTIGUOUS rows: ldc == N) = (op(A) op(B) + bias [+ C]) [relu] * dropout mask(seed, offset) of a [M,N] tensor
// -- asr_gemm_mul with the mask of asr_dropout_mask(M*N, keep, seed, offset) formed in the epilogue instead of read
// from memory (the lean NT kernel); other shapes run the plain GEMM and asr_dropout_apply in place.
extern "C" int asr_dropout_apply(asr_handle* h, int dtype, const void* in, void* out, size_t n, float keep_prob,
                                 uint64_t seed, uint64_t offset, asr_stream s);
extern "C" int asr_gemm_drop(asr_handle* h, int dtype, int transA, int transB, int M, int N, int K, const void* A,
                             int lda, const void* B, int ldb, float* C, int ldc, const float* bias, int accumulate,
                             int act, float keep_prob, uint64_t seed, uint64_t offset, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!asr_dtype_ok(dtype) || (act != 0 && act != 1) || M < 0 || N < 0 || K < 0 || !A || !B || !C ||
      lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc != N || N % 4 != 0 || !(keep_prob > 0.f && keep_prob <= 1.f))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_gemm_drop: bad args M=%d N=%d K=%d ldc=%d (ldc == N, N %% 4 == 0)", M, N, K, ldc);
  if (M == 0 || N == 0) return ASR_OK;
  hipStream_t st = (hipStream_t)s;
  bool fused = false;
  const GemmDrop drop = {keep_prob, seed, offset, 1};
  if (dtype == ASR_F32)
    launch_gemm<float, float>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act, nullptr, 0, &fused, drop);
  else
    launch_gemm<bf16_t, float>(h, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, st, act, nullptr, 0, &fused, drop);
  ASR_CHECK_LAUNCH(h, "asr_gemm_drop");
  if (!fused) return asr_dropout_apply(h, ASR_F32, C, C, (size_t)M * N, keep_prob, seed, offset, s);
  return ASR_OK;
}

// ---------------------------------------------------------------- implicit-GEMM conv entry points
extern "C" int asr_conv3x3_prep_weights(asr_handle* h, const float* w_hwio, int Cin, int Cout, void* wt_fwd,
                                        void* wt_bwd, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!w_hwio || !wt_fwd || !wt_bwd || Cin < 1 || Cout < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_prep_weights: bad args");
  const size_t total = (size_t)9 * Cin * Cout;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(conv3x3_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, w_hwio, Cin, Cout,
                     (bf16_t*)wt_fwd, (bf16_t*)wt_bwd);
  ASR_CHECK_LAUNCH(h, "asr_conv3x3_prep_weights");
  return ASR_OK;
}

static unsigned long long* g_convdbg_host = nullptr;
extern "C" int asr_debug_conv_cycles(unsigned long long* out, int n) {
  if (!g_convdbg_host || n > 64 * 4 * 8) return -1;
  return hipMemcpy(out, g_convdbg_host, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
template <typename TO>
static int conv3x3_launch(asr_handle* h, const void* x, int Nimg, int H, int W, int Cin, const void* wt,
                          const float* bias, int Cout, int act, void* out, hipStream_t st,
                          ConvGate gate = ConvGate{nullptr, 1.f, 0, 0, 0}) {
  const long long mp = (long long)Nimg * H * W;
  if (mp <= 0 || mp >= (1ll << 31)) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3: %lld pixels", mp);
  const int Mpix = (int)mp;
  {
    // image-resident form: the image with its border fits one CU's LDS (twice: the next image is staged under the
    // products of this one) and there are enough images to fill the chip; ASR_CONV_IMG=0 keeps the tiled kernel (A/B)
    static const bool img_on = [] { const char* e = getenv("ASR_CONV_IMG"); return !(e && e[0] == '0'); }();
    const size_t img_bytes = (size_t)(H + 2) * (W + 2) * (Cin * 2 + 16);
    const int nvec = H * W * Cin / 8;
    const bool shape = (Cin == 64 || Cin == 128) && (Cout == 64 || Cout == 128);
    if (img_on && shape && nvec <= 16 * 256 && img_bytes <= (size_t)156 * 1024 && Nimg >= 64) {
      const int nbuf = 2 * img_bytes <= (size_t)158 * 1024 ? 2 : 1;
      const size_t lds = nbuf * img_bytes + 128 * sizeof(float);             // images + the bias vector
      const int mv = (nvec + 255) / 256;
      const unsigned grid = (unsigned)(Nimg < h->num_cu ? Nimg : h->num_cu);
      static unsigned long long* dbg_host = nullptr;
      static const bool dbg_on = [] { const char* e = getenv("ASR_CONV_DBG"); return e && e[0] == '1'; }();
      if (dbg_on && !dbg_host) {
        (void)hipMalloc(&dbg_host, 64 * 4 * 8 * sizeof(unsigned long long));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_convdbg), &dbg_host, sizeof(dbg_host));
        g_convdbg_host = dbg_host;
      }
      // compile-time epilogue: 0 none, 1 ReLU, 2 gated data gradient with a uniform scale, 4 with the Philox mask, 3 ReLU + dropout
      const int actc = act == 2 ? (gate.use_drop == 1 ? 4 : 2) : act;
      static const bool stream_on = [] { const char* e = getenv("ASR_CONV_STREAM"); return !(e && e[0] == '0'); }();
      const bool strm = stream_on && nbuf == 2;
#define ASR_CONV_IMG_A(CI, CO, MV, AC)                                                                               \
  do {                                                                                                               \
    /* measured (profiles/r05_conv_stream.md): streaming wins 5 % for 64 -> 64 without the gate loads, loses 3 - 12 % elsewhere */ \
    constexpr bool SOK = CI == 64 && CO == 64 && (AC == 1 || AC == 3);                                               \
    auto k = conv3x3_img_kernel<TO, CI, CO, MV, AC, false>;                                                          \
    if constexpr (SOK) { if (strm) k = conv3x3_img_kernel<TO, CI, CO, MV, AC, false, 4, true>; }                     \
    if constexpr (AC == 1) { if (dbg_on) k = conv3x3_img_kernel<TO, CI, CO, MV, AC, true>; }                         \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, Nimg, H, W, (const bf16_t*)x, (const bf16_t*)wt, (TO*)out, \
                       bias, gate, nbuf);                                                                            \
  } while (0)
#define ASR_CONV_IMG8_A(CI, CO, MV, AC)                                                                              \
  do {                                                                                                               \
    auto k = conv3x3_img_kernel<TO, CI, CO, MV, AC, false, 8>;                                                       \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, Nimg, H, W, (const bf16_t*)x, (const bf16_t*)wt, (TO*)out, \
                       bias, gate, nbuf);                                                                            \
  } while (0)
#define ASR_CONV_IMG(CI, CO, MV)                                                                                     \
  do {                                                                                                               \
    if constexpr (sizeof(TO) == 4) { ASR_CONV_IMG_A(CI, CO, MV, 0); }                                                \
    else {                                                                                                           \
      if (actc == 1) ASR_CONV_IMG_A(CI, CO, MV, 1);                                                                  \
      else if (actc == 2) ASR_CONV_IMG_A(CI, CO, MV, 2);                                                             \
      else if (actc == 3) ASR_CONV_IMG_A(CI, CO, MV, 3);                                                             \
      else if (actc == 4) ASR_CONV_IMG_A(CI, CO, MV, 4);                                                             \
      else ASR_CONV_IMG_A(CI, CO, MV, 0);                                                                            \
    }                                                                                                                \
  } while (0)
      // (the VGG front-end's images: 40 x 11 x 64 -> 14 staged vectors per thread, 20 x 6 x 64 -> 4, 20 x 6 x 128 -> 8)
      // eight waves (two per SIMD, two output tiles each) for the forward ReLU + dropout of small images: the sibling wave
      // hides the Philox rounds of the epilogue (20 x 6 x 64 -> 128: 1.96 -> 1.70 ms).  Measured, not used elsewhere: the
      // 40 x 11 forms spill 16 - 46 registers at 256 per wave (ACT 3: 3.54 -> 4.15 ms), the ReLU forms are unchanged (1.17 ms).
      static const bool w8_on = [] { const char* e = getenv("ASR_CONV_IMG_W8"); return !(e && e[0] == '0'); }();
      bool done8 = false;
      if constexpr (sizeof(TO) == 2) {
        if (Cin == 64 && w8_on && mv <= 4 && actc == 3) {
          if (Cout == 64) ASR_CONV_IMG8_A(64, 64, 2, 3); else ASR_CONV_IMG8_A(64, 128, 2, 3);
          done8 = true;
        }
      }
      if (done8) {}
      else if (Cin == 64 && Cout == 64) { if (mv <= 14) ASR_CONV_IMG(64, 64, 14); else ASR_CONV_IMG(64, 64, 16); }
      else if (Cin == 64 && Cout == 128) { if (mv <= 4) ASR_CONV_IMG(64, 128, 4); else ASR_CONV_IMG(64, 128, 16); }
      else if (Cin == 128 && Cout == 128) { if (mv <= 8) ASR_CONV_IMG(128, 128, 8); else ASR_CONV_IMG(128, 128, 16); }
      else { if (mv <= 8) ASR_CONV_IMG(128, 64, 8); else ASR_CONV_IMG(128, 64, 16); }
#undef ASR_CONV_IMG
#undef ASR_CONV_IMG_A
#undef ASR_CONV_IMG8_A
      ASR_CHECK_LAUNCH(h, "asr_conv3x3(image-resident)");
      return ASR_OK;
    }
  }
  const int tm = (Mpix + 127) / 128;
  if (Cout % 128 == 0) {
    const size_t lds = (size_t)2 * (128 + 128) * 72 * sizeof(bf16_t);
    auto k = conv3x3_nt_bf16_kernel<TO, 128>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)tm * (Cout / 128)), dim3(256), lds, st, Mpix, H, W, Cin, Cout,
                       (const bf16_t*)x, (const bf16_t*)wt, (TO*)out, bias, act, gate);
  } else {
    const size_t lds = (size_t)2 * (128 + 64) * 72 * sizeof(bf16_t);
    auto k = conv3x3_nt_bf16_kernel<TO, 64>;
    hipLaunchKernelGGL(k, dim3((unsigned)tm * (Cout / 64)), dim3(256), lds, st, Mpix, H, W, Cin, Cout,
                       (const bf16_t*)x, (const bf16_t*)wt, (TO*)out, bias, act, gate);
  }
  ASR_CHECK_LAUNCH(h, "asr_conv3x3");
  return ASR_OK;
}

extern "C" int asr_conv3x3_fwd(asr_handle* h, const void* x, int Nimg, int H, int W, int Cin, const void* wt_fwd,
                               const float* bias, int Cout, int relu, void* out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!x || !wt_fwd || !out || Nimg < 1 || H < 1 || W < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_fwd: bad args");
  if (Cin % 64 != 0 || Cout % 64 != 0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3_fwd: Cin=%d, Cout=%d must be multiples of 64", Cin, Cout);
  return conv3x3_launch<bf16_t>(h, x, Nimg, H, W, Cin, wt_fwd, bias, Cout, relu ? 1 : 0, out, (hipStream_t)s);
}

extern "C" int asr_conv3x3_fwd_drop(asr_handle* h, const void* x, int Nimg, int H, int W, int Cin, const void* wt_fwd,
                                    const float* bias, int Cout, float keep_prob, uint64_t seed, uint64_t offset,
                                    void* out, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!x || !wt_fwd || !out || Nimg < 1 || H < 1 || W < 1 || !(keep_prob > 0.f && keep_prob <= 1.f))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_fwd_drop: bad args");
  if (Cin % 64 != 0 || Cout % 64 != 0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3_fwd_drop: Cin=%d, Cout=%d must be multiples of 64", Cin, Cout);
  const ConvGate gate = {nullptr, keep_prob, seed, offset, 1};
  return conv3x3_launch<bf16_t>(h, x, Nimg, H, W, Cin, wt_fwd, bias, Cout, 3, out, (hipStream_t)s, gate);
}

extern "C" int asr_conv3x3_bwd_data(asr_handle* h, const void* dy, int Nimg, int H, int W, int Cout,
                                    const void* wt_bwd, int Cin, float* dx, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!dy || !wt_bwd || !dx || Nimg < 1 || H < 1 || W < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_bwd_data: bad args");
  if (Cin % 64 != 0 || Cout % 64 != 0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3_bwd_data: Cin=%d, Cout=%d must be multiples of 64", Cin, Cout);
  // the data gradient is a convolution of dOut (Cout channels) with the flipped-tap image -> Cin channels
  return conv3x3_launch<float>(h, dy, Nimg, H, W, Cout, wt_bwd, nullptr, Cin, 0, dx, (hipStream_t)s);
}

extern "C" int asr_conv3x3_bwd_data_relu(asr_handle* h, const void* dy, int Nimg, int H, int W, int Cout,
                                         const void* wt_bwd, int Cin, const void* act_below, float keep_prob,
                                         uint64_t seed, uint64_t offset, int use_drop, void* dpre_below, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!dy || !wt_bwd || !act_below || !dpre_below || Nimg < 1 || H < 1 || W < 1 ||
      (use_drop && !(keep_prob > 0.f && keep_prob <= 1.f)) || use_drop < 0 || use_drop > 2)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_bwd_data_relu: bad args");
  if (Cin % 64 != 0 || Cout % 64 != 0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3_bwd_data_relu: Cin=%d, Cout=%d must be multiples of 64", Cin, Cout);
  const ConvGate gate = {(const bf16_t*)act_below, keep_prob, seed, offset, use_drop};
  return conv3x3_launch<bf16_t>(h, dy, Nimg, H, W, Cout, wt_bwd, nullptr, Cin, 2, dpre_below, (hipStream_t)s, gate);
}

// ---------------------------------------------------------------- 3x3 weight gradient, image-resident form (round 5)
// dW[tap * CIN + ci][co] = sum over images and pixels p of x[p + s_tap, ci] * dY[p, co].  The tiled kernel above gathers every
// shifted pixel row from L2 once per tap (nine times) and, at COUT = 64, multiplies half-empty 128-column tiles: 6.4 ms for
// the 40 x 11 x 64 -> 64 layer of cfg C (299 TFLOP/s).  Here a workgroup stages a whole frame image of x (with its zero
// border) and 64 output channels of dY into LDS ONCE per image, in the pixel-major layout they have in memory, and takes both
// MFMA operands out of them with the transposing LDS read (ds_read_b64_tr_b16: each lane passes the address of 4 channels of
// ONE pixel and receives ONE channel at 4 consecutive pixels -- the row stride is free, so the padded image works as it lies
// and a tap is a constant byte offset).  Wave w owns the 16 input channels 16 w .. of all nine taps x the 64 output channels of
// the workgroup's column block (blockIdx.y): 36 accumulator tiles in registers for the whole launch; CIN / 16 waves.  The
// reduction runs over the pixels of an image in chunks of 32 and over the images blockIdx.x, + gridDim.x, ...; the next
// image's vectors are requested a few per chunk under the multiplies.  Each workgroup row writes ONE slab [9 CIN][COUT] (its
// 64 columns); splitk_reduce_kernel sums the slabs in a fixed order (deterministic).  HBM traffic: dY once, x once per
// 64-column block.  40 x 11 x 64 -> 64: 6.40 -> 2.59 ms (738 TFLOP/s) on the first build.
// NSPLIT: the workgroup's 64 output columns dealt over NSPLIT waves per input-channel group (CIN = 64: 2 -> eight waves, two per
// SIMD, each 9 x 2 accumulator tiles: the second wave of a SIMD fills the LDS latency of the first; 2.66 -> see the launcher)
// BIAS: the bias gradient (column sums of dY over all pixels) from the staged dY as well: the wave of input-channel group g
// multiplies a fragment of ONES with its dY fragments in the chunks c % (CIN / 16) == g (one more MFMA per NT tiles in a
// quarter / an eighth of the chunks, instead of a second pass over dY: 0.30 - 0.54 ms per layer of the cfg C step), and
// writes its partial sums as row 9 CIN + g of the slab; wgrad_img_reduce_kernel adds those rows over slabs and groups.
template <int CIN, int MAXVX, int MAXVY, int NSPLIT, bool BIAS = false>
__global__ __launch_bounds__(CIN * 4 * NSPLIT, 1) void conv3x3_wgrad_img_kernel(int Nimg, int H, int W, int Cout,
                                                                       const bf16_t* __restrict__ X,
                                                                       const bf16_t* __restrict__ dY,
                                                                       float* __restrict__ partial) {
  static_assert(CIN == 64 || CIN == 128, "one 16-channel group of x per wave, 4 or 8 waves");
  constexpr int NTH = CIN * 4 * NSPLIT;
  constexpr int CB = 64, NT = 4 / NSPLIT;                    // columns of one workgroup / output tiles of one wave
  constexpr int PSX = CIN * 2 + 16, PSY = CB * 2 + 16;       // bytes per pixel in LDS (16-byte pad: distinct banks per block row)
  constexpr int MAXV = MAXVX > MAXVY ? MAXVX : MAXVY;
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) % (CIN / 16), nh = (tid >> 6) / (CIN / 16);   // input-channel group / column part of this wave
  const int HW = H * W, WP = W + 2;
  const int n0 = blockIdx.y * CB;
  const int nchunk = (HW + 31) / 32;
  const int ximg_bytes = (H + 2) * WP * PSX;
  char* xs = wsm;                                            // x image with its border
  char* ys = wsm + ximg_bytes;                               // dY image (64 channels), nchunk * 32 pixel rows (tail rows stay zero)
  const int yimg_bytes = nchunk * 32 * PSY;
  const int nvx = HW * CIN / 8, nvy = HW * CB / 8;           // 16-byte vectors of one image
  const float invW = 1.0f / (float)W;

  for (int i = tid * 16; i < ximg_bytes + yimg_bytes; i += NTH * 16)
    *reinterpret_cast<bf16x8_t*>(wsm + i) = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();

  bf16x8_t stx[MAXVX], sty[MAXVY];
  auto fetch1 = [&](int img, int i) {                        // vector i of this thread, both operands
    const int v = tid + i * NTH;
    if (i < MAXVX && v < nvx) stx[i < MAXVX ? i : 0] = reinterpret_cast<const bf16x8_t*>(X + (size_t)img * HW * CIN)[v];
    if (i < MAXVY && v < nvy) {
      const int p = v / (CB / 8), cv = v % (CB / 8);
      sty[i < MAXVY ? i : 0] = *reinterpret_cast<const bf16x8_t*>(dY + ((size_t)img * HW + p) * Cout + n0 + cv * 8);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < MAXVX; ++i) {
      const int v = tid + i * NTH;
      if (v < nvx) {
        const int p = v / (CIN / 8), cv = v % (CIN / 8);
        const int y = (int)(((float)p + 0.5f) * invW), x = p - y * W;
        *reinterpret_cast<bf16x8_t*>(xs + ((y + 1) * WP + x + 1) * PSX + cv * 16) = stx[i];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXVY; ++i) {
      const int v = tid + i * NTH;
      if (v < nvy) {
        const int p = v / (CB / 8), cv = v % (CB / 8);
        *reinterpret_cast<bf16x8_t*>(ys + p * PSY + cv * 16) = sty[i];
      }
    }
  };
  // lane -> its piece of a 4-pixel x 16-channel block: pixel row (lane & 15) >> 2 of k-group lane >> 4, channels 4 (lane & 3)..
  const int kg = lane >> 4, prow = (lane & 15) >> 2, c4 = lane & 3;
  const int xch = (wave * 16 + c4 * 4) * 2;                  // byte offset of this lane's 4 input channels
  const int ych = (nh * NT * 16 + c4 * 4) * 2;
  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = ((t / 3 - 1) * WP + (t % 3 - 1)) * PSX;
  // LDS byte offset of a pixel's centre tap in the x image.  A pixel past the image reads the LAST pixel of x (finite)
  // against a zero row of dY.
  auto xofs = [&](int p) -> int {
    const int pc = p < HW ? p : HW - 1;
    const int y = (int)(((float)pc + 0.5f) * invW), x = pc - y * W;
    return ((y + 1) * WP + x + 1) * PSX + xch;
  };
  typedef __attribute__((address_space(3))) bf16x4_t wl4_t;
  auto trfrag = [&](const char* lo, const char* hi) -> bf16x8_t {
    const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wl4_t*)(lo));
    const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wl4_t*)(hi));
    return (bf16x8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  };

  f32x4_t acc[9][NT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t accb[BIAS ? NT : 1];
#pragma unroll
  for (int j = 0; j < (BIAS ? NT : 1); ++j) accb[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16x8_t ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  int img = blockIdx.x;
  if (img < Nimg) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) fetch1(img, i);
    lstore();
  }
  __syncthreads();
  for (; img < Nimg; img += gridDim.x) {
    const int nxt = img + gridDim.x;
    const bool more = nxt < Nimg;                            // block-uniform
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
      // the next image: vector pairs i = 2c, 2c + 1 in chunk c (2 nchunk >= MAXV for every supported image: the launcher checks)
      if (more) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
          if (i / 2 == c) fetch1(nxt, i);
      }
      const int p0 = c * 32 + kg * 8 + prow;
      const int xlo = xofs(p0), xhi = xofs(p0 + 4);
      const int ylo = p0 * PSY + ych, yhi = (p0 + 4) * PSY + ych;
      bf16x8_t b[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = trfrag(ys + ylo + j * 32, ys + yhi + j * 32);
      if constexpr (BIAS) {
        if (c % (CIN / 16) == wave) {                        // wave-uniform
#pragma unroll
          for (int j = 0; j < NT; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b[j], accb[j], 0, 0, 0);
        }
      }
      bf16x8_t a[3];
      a[0] = trfrag(xs + xlo + tapoff[0], xs + xhi + tapoff[0]);
      a[1] = trfrag(xs + xlo + tapoff[1], xs + xhi + tapoff[1]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 2 < 9) a[(t + 2) % 3] = trfrag(xs + xlo + tapoff[t + 2], xs + xhi + tapoff[t + 2]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t % 3], b[j], acc[t][j], 0, 0, 0);
      }
    }
    __syncthreads();                                         // every wave is done with this image
    if (more) lstore();
    __syncthreads();
  }
  // slab of this workgroup row: lane holds rows rg*4 + r (input channel within the wave's group), column lane & 15 of tile j
  float* slab = partial + (size_t)blockIdx.x * (9 * CIN + (BIAS ? CIN / 16 : 0)) * Cout;
  const int col = lane & 15, rg = lane >> 4;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[(size_t)(t * CIN + wave * 16 + rg * 4 + r) * Cout + n0 + (nh * NT + j) * 16 + col] = acc[t][j][r];
  if constexpr (BIAS) {                                      // all 16 rows of accb are the same column sums: row 0
    if (rg == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) slab[(size_t)(9 * CIN + wave) * Cout + n0 + (nh * NT + j) * 16 + col] = accb[j][0];
    }
  }
}
// fixed-order sum of the slabs of conv3x3_wgrad_img_kernel<..., BIAS = true>: rows [0, Mw) -> dw, rows Mw .. Mw + G - 1 -> dbias
// Round 6: four columns per thread and eight slabs requested at a time, added in slab order -- the same sums in the same order
// as the one-load-per-addition loop this replaces, which ran at the memory latency per slab (0.98 ms for the 128 slabs of
// the 128 -> 128 layer at cfg C, three such passes on the lane that ends the step; the bytes take ~25 us).  N % 4 == 0 (the
// image-resident kernel wants Cout % 64 == 0); VEC = 0: dw is not 16-byte aligned, scalar stores.
template <int VEC>
__global__ void wgrad_img_reduce_kernel(const float* __restrict__ partial, int S, int Mw, int G, int N, float* __restrict__ dw,
                                        float* __restrict__ dbias, int accumulate) {
  const size_t slab = (size_t)(Mw + G) * N, wn = (size_t)Mw * N;
  const size_t nv = wn / 4, slab4 = slab / 4;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x, tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const f32x4_t* p4 = reinterpret_cast<const f32x4_t*>(partial);
  for (size_t q = tid; q < nv; q += nthreads) {
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= S; z += 8) {
      f32x4_t t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p4[(size_t)(z + u) * slab4 + q];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; z < S; ++z) v += p4[(size_t)z * slab4 + q];
    if (VEC) {
      f32x4_t* d = reinterpret_cast<f32x4_t*>(dw) + q;
      *d = accumulate ? *d + v : v;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) dw[4 * q + k] = accumulate ? dw[4 * q + k] + v[k] : v[k];
    }
  }
  if (dbias) {
    for (size_t n = tid; n < (size_t)N; n += nthreads) {
      float v = 0.f;
      for (int z = 0; z < S; ++z)
        for (int g = 0; g < G; ++g) v += partial[(size_t)z * slab + wn + (size_t)g * N + n];
      dbias[n] = accumulate ? dbias[n] + v : v;
    }
  }
}

extern "C" int asr_colsum(asr_handle* h, int dtype, const void* a, int M, int N, int lda, float* out, asr_stream s);
static int conv3x3_bwd_weight_impl(asr_handle* h, const void* x, const void* dy, int Nimg, int H, int W,
                                   int Cin, int Cout, float* dw, float* dbias, int accumulate, asr_stream s);
extern "C" int asr_conv3x3_bwd_weight(asr_handle* h, const void* x, const void* dy, int Nimg, int H, int W,
                                      int Cin, int Cout, float* dw, int accumulate, asr_stream s) {
  return conv3x3_bwd_weight_impl(h, x, dy, Nimg, H, W, Cin, Cout, dw, nullptr, accumulate, s);
}
extern "C" int asr_conv3x3_bwd_weight_bias(asr_handle* h, const void* x, const void* dy, int Nimg, int H, int W,
                                           int Cin, int Cout, float* dw, float* dbias, asr_stream s) {
  if (h && !dbias) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_bwd_weight_bias: dbias is NULL");
  return conv3x3_bwd_weight_impl(h, x, dy, Nimg, H, W, Cin, Cout, dw, dbias, 0, s);
}
static int conv3x3_bwd_weight_impl(asr_handle* h, const void* x, const void* dy, int Nimg, int H, int W,
                                   int Cin, int Cout, float* dw, float* dbias, int accumulate, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!x || !dy || !dw || Nimg < 1 || H < 1 || W < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_conv3x3_bwd_weight: bad args");
  if (Cin % 8 != 0 || Cout % 8 != 0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3_bwd_weight: Cin=%d, Cout=%d must be multiples of 8", Cin, Cout);
  const long long mp = (long long)Nimg * H * W;
  if (mp >= (1ll << 31)) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_conv3x3_bwd_weight: %lld pixels", mp);
  const int Mpix = (int)mp, M = 9 * Cin, N = Cout;
  static const bool wgrad_tr = [] { const char* e = getenv("ASR_CONV_WGRAD_TR"); return !(e && e[0] == '0'); }();
  // COUT = 64 (or a last 64-column tile): 128 x 64 tiles, so that no MFMA multiplies zero columns -- measured 6.4 ms per
  // call against 6.0 for the 128 x 128 tiles at cfg C's second layer (the kernel is bound by gathering every pixel nine
  // times from L2, not by its MFMAs): OFF unless ASR_CONV_WGRAD_BN64=1
  static const bool bn64_on = [] { const char* e = getenv("ASR_CONV_WGRAD_BN64"); return e && e[0] == '1'; }();
  const bool bn64 = wgrad_tr && bn64_on && N % 128 == 64;
  const int tm = (M + 127) / 128, tn = bn64 ? (N + 63) / 64 : (N + 127) / 128;
  int S = (2048 + tm * tn - 1) / (tm * tn);
  const int maxS = (Mpix + 511) / 512;
  if (S > maxS) S = maxS;
  if (S > 512) S = 512;
  while (S > 1 && (size_t)S * M * N * sizeof(float) > h->scratch_bytes - ASR_XCH_BYTES) --S;
  if ((size_t)S * M * N * sizeof(float) > h->scratch_bytes - ASR_XCH_BYTES)
    ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_conv3x3_bwd_weight: scratch too small");
  int kchunk = (Mpix + S - 1) / S;
  kchunk = (kchunk + 63) / 64 * 64;
  S = (Mpix + kchunk - 1) / kchunk;
  float* partial = (float*)h->scratch;
  hipStream_t st = (hipStream_t)s;
  {
    // image-resident form (round 5): x and dY images in LDS, 9 taps from one staging; ASR_CONV_WGRAD_IMG=0 keeps the tiled kernel
    static const bool wimg_on = [] { const char* e = getenv("ASR_CONV_WGRAD_IMG"); return !(e && e[0] == '0'); }();
    const int HWp = H * W, nchunk = (HWp + 31) / 32;
    const size_t lds = (size_t)(H + 2) * (W + 2) * (Cin * 2 + 16) + (size_t)nchunk * 32 * (64 * 2 + 16);
    const int nth = Cin * 4;
    const int mvx = (HWp * Cin / 8 + nth - 1) / nth, mvy = (HWp * 8 + nth - 1) / nth;   // staged vectors per thread
    const int mv = mvx > mvy ? mvx : mvy;
    static const bool wbias_on = [] { const char* e = getenv("ASR_CONV_WGRAD_BIAS"); return !(e && e[0] == '0'); }();
    const bool inb = dbias && wbias_on;                      // bias gradient inside the weight-gradient kernel
    const int G = inb ? Cin / 16 : 0;
    const size_t slab = (size_t)(M + G) * N * sizeof(float);
    size_t wgs = (h->scratch_bytes - ASR_XCH_BYTES) / slab;
    const size_t cols = (size_t)(Cout / 64);
    if (wgs * cols > (size_t)h->num_cu) wgs = (size_t)h->num_cu / cols;
    if (wgs > (size_t)Nimg) wgs = (size_t)Nimg;
    // (Cin = 128 runs eight waves, two per SIMD: 256 registers, of which 144 are accumulators -- small images only)
    if (wimg_on && (Cin == 64 || (Cin == 128 && mvx <= 4 && mvy <= 2)) && Cout % 64 == 0 && Nimg >= 64 &&
        lds <= (size_t)158 * 1024 && mv <= 14 && nchunk * 2 >= mv && wgs >= 32) {
#define ASR_WGRAD_IMG(CI, VX, VY, NS)                                                                                  \
  do {                                                                                                                 \
    auto kern = inb ? conv3x3_wgrad_img_kernel<CI, VX, VY, NS, true> : conv3x3_wgrad_img_kernel<CI, VX, VY, NS, false>; \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs, (unsigned)cols), dim3(CI * 4 * NS), lds, st, Nimg, H, W, Cout,        \
                       (const bf16_t*)x, (const bf16_t*)dy, partial);                                                  \
  } while (0)
      // (the VGG front-end's images: 40 x 11 x 64 -> 7 + 7 staged vectors per thread of eight waves, 20 x 6 x 64 -> 2 + 2,
      // 20 x 6 x 128 -> 4 + 2)
      static const bool split_on = [] { const char* e = getenv("ASR_CONV_WGRAD_SPLIT"); return !(e && e[0] == '0'); }();
      if (Cin == 64 && split_on) { if (mv <= 4) ASR_WGRAD_IMG(64, 2, 2, 2); else ASR_WGRAD_IMG(64, 7, 7, 2); }
      else if (Cin == 64) { if (mv <= 4) ASR_WGRAD_IMG(64, 4, 4, 1); else ASR_WGRAD_IMG(64, 14, 14, 1); }
      else ASR_WGRAD_IMG(128, 4, 2, 1);
#undef ASR_WGRAD_IMG
      const size_t total = (size_t)M * N;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      if (inb) {
        const int rb = (int)((total / 4 + 255) / 256);
        if (((uintptr_t)dw) % 16 == 0)
          hipLaunchKernelGGL(wgrad_img_reduce_kernel<1>, dim3(rb), dim3(256), 0, st, partial, (int)wgs, M, G, N, dw, dbias, accumulate);
        else
          hipLaunchKernelGGL(wgrad_img_reduce_kernel<0>, dim3(rb), dim3(256), 0, st, partial, (int)wgs, M, G, N, dw, dbias, accumulate);
      } else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, partial, (int)wgs, M, N, dw, N,
                           nullptr, accumulate, 0, 0);
      ASR_CHECK_LAUNCH(h, "asr_conv3x3_bwd_weight(image-resident)");
      if (dbias && !inb) return asr_colsum(h, ASR_BF16, dy, Mpix, Cout, Cout, dbias, s);
      return ASR_OK;
    }
  }
  if (wgrad_tr) {
    const size_t lds = (size_t)2 * TN_STAGE;
    auto kern = bn64 ? conv3x3_wgrad_tr_kernel<64> : conv3x3_wgrad_tr_kernel<128>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(tn, tm, S), dim3(256), lds, st, Mpix, H, W, Cin, Cout,
                       (const bf16_t*)x, (const bf16_t*)dy, kchunk, partial);
  } else {
    const size_t lds = (size_t)2 * (128 + 128) * 72 * sizeof(bf16_t);
    (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv3x3_wgrad_tn_kernel, dim3(tn, tm, S), dim3(256), lds, st, Mpix, H, W, Cin, Cout,
                       (const bf16_t*)x, (const bf16_t*)dy, kchunk, partial);
  }
  const size_t total = (size_t)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, partial, S, M, N, dw, N, nullptr,
                     accumulate, 0, 0);
  ASR_CHECK_LAUNCH(h, "asr_conv3x3_bwd_weight");
  if (dbias) return asr_colsum(h, ASR_BF16, dy, Mpix, Cout, Cout, dbias, s);
  return ASR_OK;
}

// ---------------------------------------------------------------- decoder cell: product + LSTM cell in one launch
extern "C" int asr_lstm_cell_gemm_prep(asr_handle* h, const float* W, const float* bias, int K, int U, float* W_il,
                                       asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!W || !W_il || K < 1 || U < 1 || ((uintptr_t)W_il) % 16 != 0)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_cell_gemm_prep: bad args");
  const size_t n = (size_t)(K + (bias ? 1 : 0)) * U;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(interleave_gates_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, W, bias, K, U, W_il);
  ASR_CHECK_LAUNCH(h, "asr_lstm_cell_gemm_prep");
  return ASR_OK;
}
extern "C" int asr_lstm_cell_gemm_ok(int B, int K, int U, int ldx) {
  return B >= 1 && B <= 32 && K >= 64 && K % 64 == 0 && U >= 8 && U % 8 == 0 && ldx % 4 == 0 && ldx >= K;
}
extern "C" int asr_lstm_cell_gemm_fwd(asr_handle* h, const float* x, int ldx, int K, const float* W_il, int has_bias,
                                      const float* c_prev, const float* h_prev, const float* peep, const float* live,
                                      int B, int U, float forget_bias, float cell_clip, float* gates, float* c_raw,
                                      float* c_out, float* h_out, float* h_raw, const float* out_mask, float* cell_out,
                                      float* h_out2, int ld_h2, float* cell_out2, int ld_c2, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!x || !W_il || !c_prev || !h_prev || !live || !gates || !c_raw || !c_out || !h_out || !h_raw ||
      !asr_lstm_cell_gemm_ok(B, K, U, ldx) || ((uintptr_t)x) % 16 != 0 || ((uintptr_t)W_il) % 16 != 0 ||
      (h_out2 && ld_h2 < U) || (cell_out2 && ld_c2 < U))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_cell_gemm_fwd: bad args (B <= 32, K %% 64 == 0, U %% 8 == 0, 16-byte aligned rows)");
  SkinnyCell c;
  c.c_prev = c_prev; c.h_prev = h_prev; c.peep = peep; c.live = live; c.out_mask = out_mask;
  c.gates = gates; c.c_raw = c_raw; c.c_out = c_out; c.h_out = h_out; c.h_raw = h_raw;
  c.cell_out = cell_out; c.h_out2 = h_out2; c.cell_out2 = cell_out2;
  c.U = U; c.ld_h2 = ld_h2; c.ld_c2 = ld_c2; c.fb = forget_bias; c.clip = cell_clip;
  const int N = 4 * U;
  hipLaunchKernelGGL((gemm_skinny_f32_kernel<false, 2, true>), dim3(N / 32, (unsigned)((B + 15) / 16)), dim3(64 * SK_WAVES), 0,
                     (hipStream_t)s, B, N, K, x, ldx, W_il, N, (float*)nullptr, N, has_bias ? W_il + (size_t)K * N : nullptr,
                     0, 0, c);
  ASR_CHECK_LAUNCH(h, "asr_lstm_cell_gemm_fwd");
  return ASR_OK;
}

// ---- the same with bf16 weight images (bf16-operand models)
extern "C" size_t asr_lstm_cell_gemm_h_bytes(int K, int U) {
  return (size_t)K * 4 * U * 2 * 2 + (size_t)4 * U * 4;     // fragment image | interleaved fp32 bias | plain bf16 copy
}
static inline float* cell_h_bias(void* img, int K, int U) { return reinterpret_cast<float*>((char*)img + (size_t)K * 4 * U * 2); }
static inline bf16_t* cell_h_plain(void* img, int K, int U) {
  return reinterpret_cast<bf16_t*>((char*)img + (size_t)K * 4 * U * 2 + (size_t)4 * U * 4);
}
extern "C" int asr_lstm_cell_gemm_prep_h(asr_handle* h, const float* W, const float* bias, int K, int U, void* img,
                                         asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!W || !img || K < 64 || K % 64 != 0 || U < 8 || U % 8 != 0 || ((uintptr_t)img) % 16 != 0)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_cell_gemm_prep_h: bad args");
  const size_t n = (size_t)K * 4 * U;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cell_images_h_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, W, bias, K, U,
                     (bf16_t*)img, cell_h_bias(img, K, U), cell_h_plain(img, K, U));
  ASR_CHECK_LAUNCH(h, "asr_lstm_cell_gemm_prep_h");
  return ASR_OK;
}
extern "C" int asr_lstm_cell_gemm_fwd_h(asr_handle* h, const float* x, int ldx, int K, const void* img,
                                        const float* c_prev, const float* h_prev, const float* peep, const float* live,
                                        int B, int U, float forget_bias, float cell_clip, float* gates, float* c_raw,
                                        float* c_out, float* h_out, float* h_raw, const float* out_mask, float* cell_out,
                                        float* h_out2, int ld_h2, float* cell_out2, int ld_c2, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!x || !img || !c_prev || !h_prev || !live || !gates || !c_raw || !c_out || !h_out || !h_raw ||
      !asr_lstm_cell_gemm_ok(B, K, U, ldx) || ((uintptr_t)x) % 16 != 0 || ((uintptr_t)img) % 16 != 0 ||
      (h_out2 && ld_h2 < U) || (cell_out2 && ld_c2 < U))
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_cell_gemm_fwd_h: bad args (B <= 32, K %% 64 == 0, U %% 8 == 0, 16-byte aligned rows)");
  SkinnyCell c;
  c.c_prev = c_prev; c.h_prev = h_prev; c.peep = peep; c.live = live; c.out_mask = out_mask;
  c.gates = gates; c.c_raw = c_raw; c.c_out = c_out; c.h_out = h_out; c.h_raw = h_raw;
  c.cell_out = cell_out; c.h_out2 = h_out2; c.cell_out2 = cell_out2;
  c.U = U; c.ld_h2 = ld_h2; c.ld_c2 = ld_c2; c.fb = forget_bias; c.clip = cell_clip;
  const int N = 4 * U;
  hipLaunchKernelGGL((gemm_skinny_f32_kernel<false, 2, true, true>), dim3(N / 32, (unsigned)((B + 15) / 16)),
                     dim3(64 * SK_WAVES), 0, (hipStream_t)s, B, N, K, x, ldx, (const float*)img, N, (float*)nullptr, N,
                     cell_h_bias(const_cast<void*>(img), K, U), 0, 0, c);
  ASR_CHECK_LAUNCH(h, "asr_lstm_cell_gemm_fwd_h");
  return ASR_OK;
}
// dx [B, K] = dpre [B, 4U] W^T on the plain bf16 copy inside the image (the decoder step's backward product)
extern "C" int asr_lstm_cell_gemm_bwd_h(asr_handle* h, const float* dpre, int B, int K, int U, const void* img, float* dx,
                                        int lddx, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!dpre || !img || !dx || B < 1 || B > 32 || K < 16 || K % 16 != 0 || U < 16 || U % 16 != 0 || lddx < K ||
      ((uintptr_t)dpre) % 16 != 0 || ((uintptr_t)img) % 16 != 0)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_lstm_cell_gemm_bwd_h: bad args (B <= 32, K %% 16 == 0, U %% 16 == 0)");
  hipLaunchKernelGGL((gemm_skinny_f32_kernel<true, 1, false, true>), dim3(K / 16, (unsigned)((B + 15) / 16)),
                     dim3(64 * SK_WAVES), 0, (hipStream_t)s, B, K, 4 * U, dpre, 4 * U,
                     (const float*)cell_h_plain(const_cast<void*>(img), K, U), 4 * U, dx, lddx, (const float*)nullptr, 0, 0,
                     SkinnyCell{});
  ASR_CHECK_LAUNCH(h, "asr_lstm_cell_gemm_bwd_h");
  return ASR_OK;
}
