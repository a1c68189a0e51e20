// CTC prefix beam search on gfx950.
//
// Restates models/ctc/decoders/beam_search_decoder.py:53-152 (the reference's numpy decoder,
// which is also what pins this kernel: tests/golden/decoders_v1.npz) and stands in for
// tf.nn.ctc_beam_search_decoder at models/ctc/ctc.py:344-346.
//
// Reference semantics kept exactly:
//   * log-space (p_blank, p_non_blank) per prefix, float64;
//   * per frame, for every class c and every beam entry: blank -> stays; c != last -> extend
//     with (p_b + p_nb); c == last -> extend with p_b only AND the unchanged prefix collects
//     p_nb (the "merging case", :120-139);
//   * next_beam is a dict keyed by the prefix tuple: an extension that spells a prefix already
//     in the beam merges into it;
//   * sorted(..., key=logsumexp(p_b,p_nb), reverse=True)[:beam_width] is a STABLE sort, so ties
//     keep dict insertion order (vocab-major: for c: for entry: new_prefix, then prefix).
//
// Mapping: one workgroup per utterance (utterances are independent -> B workgroups; the T
// frames are serial).  Prefixes live in a per-utterance trie (parent, label) in the workspace;
// prefix identity is a 64-bit rolling hash + length (a dropped-and-recreated prefix gets a
// new trie node but the same hash, so dict merging stays correct).  Per frame:
//   1. lp = log_softmax(logits[t]) in fp64 (LDS),
//   2. W stay candidates (with the merge from the parent entry, if that is in the beam),
//   3. W x C extension totals to scratch (pairs that merged are masked out),
//   4. exact top-W by a byte-wise radix select over the 12-byte composite key
//      (orderable fp64 total, ~insertion index)  -> ties resolved as the reference's stable sort,
//   5. winners ranked (W^2 compares) into the next beam, new trie nodes appended.
// Exact class pruning (step 2b): an extension (entry j, class c) can only reach the top W if fewer
// than W classes c' != last_j have a strictly larger log-probability (each of those gives a distinct
// candidate of the same parent with a larger total), so only the classes at or above the (W+1)-th
// largest log-probability of the frame (minus a margin far above fp64 rounding, so "larger" survives
// the addition) are enumerated: W*(W+1) candidates instead of W*C (10 k vs 339 k for C = 3387, W = 100).
// Merged pairs are handled in the stay candidates and do not depend on the pruning; insertion
// indices still use the real class id, so the reference's tie order is unchanged.
// HBM/latency-bound integer+fp64 work; nothing here is GEMM-shaped.
#include "common.h"
#include <math.h>

namespace {

constexpr int BEAM_MAX = 128;
constexpr int BEAM_THREADS = 256;
constexpr double DNEG = -INFINITY;

__device__ __forceinline__ double lse2d(double a, double b) {
  const double m = fmax(a, b);
  if (m == DNEG) return DNEG;
  return m + log(exp(a - m) + exp(b - m));
}
__device__ __forceinline__ double lse3d(double a, double b, double c) {
  const double m = fmax(fmax(a, b), c);
  if (m == DNEG) return DNEG;
  return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}
// order-preserving map double -> uint64 (ascending)
__device__ __forceinline__ unsigned long long okey(double x) {
  unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unokey(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ unsigned long long hmix(unsigned long long h, int c) {
  unsigned long long x = h * 0x9E3779B97F4A7C15ull + (unsigned long long)(c + 1);
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29;
  return x;
}

struct Entry {            // one beam entry
  double pb, pnb;
  unsigned long long hash, phash;   // hash of the prefix / of its parent prefix
  int node, len, last;
};

// byte `byte` (11 = most significant) of the 12-byte composite (key64 : ~idx32)
__device__ __forceinline__ unsigned digit_of(unsigned long long key, unsigned nidx, int byte) {
  return byte >= 4 ? (unsigned)((key >> ((byte - 4) * 8)) & 0xff) : ((nidx >> (byte * 8)) & 0xff);
}

__global__ __launch_bounds__(BEAM_THREADS) void ctc_beam_kernel(
    const float* __restrict__ logits, int T, int B, int C, const int32_t* __restrict__ seq_len,
    int blank, int W, double* __restrict__ tot_ws, int2* __restrict__ node_ws,
    int32_t* __restrict__ out_labels, int32_t* __restrict__ out_len, double* __restrict__ out_score) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lp = reinterpret_cast<double*>(smem);      // [C]
  int* kc = reinterpret_cast<int*>(lp + C);          // [C] kept classes of the frame, ascending
  int* kpos = kc + C;                                // [C] position of a class in kc, -1 if pruned
  __shared__ double s_L[BEAM_MAX];                   // logsumexp(p_b, p_nb) of each beam entry
  __shared__ unsigned wcnt[BEAM_THREADS / 64];
  __shared__ int s_K;
  __shared__ double s_thr;
  __shared__ Entry beam[BEAM_MAX];
  __shared__ Entry nbeam[BEAM_MAX];
  __shared__ double s_pb[BEAM_MAX], s_pnb[BEAM_MAX];  // stay candidates
  __shared__ unsigned long long s_key[BEAM_MAX];
  __shared__ unsigned s_idx[BEAM_MAX];
  __shared__ int s_parent[BEAM_MAX];
  __shared__ unsigned hist[256];
  __shared__ unsigned long long w_key[BEAM_MAX];
  __shared__ unsigned w_nidx[BEAM_MAX];
  __shared__ int w_src[BEAM_MAX];                    // candidate id of each winner
  __shared__ double red[BEAM_THREADS / 64];
  __shared__ unsigned long long sel_key;              // threshold prefix being built
  __shared__ unsigned sel_nidx;
  __shared__ unsigned sel_remaining;
  __shared__ int s_nw, s_nb, s_nodes;

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Tb = min(max(seq_len[b], 0), T);
  double* tot = tot_ws + (size_t)b * W * C;
  int2* nodes = node_ws + (size_t)b * ((size_t)T * W + 1);

  if (tid == 0) {
    beam[0].pb = 0.0; beam[0].pnb = DNEG; beam[0].hash = 0x1234567ull; beam[0].phash = 0;
    beam[0].node = 0; beam[0].len = 0; beam[0].last = -1;
    nodes[0] = make_int2(-1, -1);
    s_nb = 1; s_nodes = 1;
  }
  __syncthreads();

  for (int t = 0; t < Tb; ++t) {
    const int nb = s_nb;
    // ---- 1. fp64 log-softmax of the frame
    const float* row = logits + ((size_t)t * B + b) * C;
    float m = -INFINITY;
    for (int c = tid; c < C; c += BEAM_THREADS) m = fmaxf(m, row[c]);
    m = wave_reduce_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    double mm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    double ssum = 0.0;
    for (int c = tid; c < C; c += BEAM_THREADS) ssum += exp((double)row[c] - mm);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
    if (lane == 0) red[wave] = ssum;
    __syncthreads();
    const double z = mm + log(red[0] + red[1] + red[2] + red[3]);
    for (int c = tid; c < C; c += BEAM_THREADS) lp[c] = (double)row[c] - z;
    // ---- 2. parent lookup + stay candidates
    if (tid < nb) {
      int par = -1;
      const Entry e = beam[tid];
      if (e.len > 0)
        for (int j = 0; j < nb; ++j)
          if (beam[j].hash == e.phash && beam[j].len + 1 == e.len) { par = j; break; }
      s_parent[tid] = par;
    }
    __syncthreads();
    if (tid < nb) {
      const Entry e = beam[tid];
      const double lpb = lp[blank];
      double pb = lse3d(DNEG, e.pb + lpb, e.pnb + lpb);
      double pnb = DNEG;
      unsigned first = (unsigned)blank * (2u * W) + 2u * tid;       // touched when c == blank
      const int par = s_parent[tid];
      if (e.len > 0) {
        const double lpl = lp[e.last];
        // insertion order inside class c = last: parent's extension (2*par) comes at its own
        // position, the entry's own merging-case touch at 2*tid+1
        double from_parent = DNEG;
        if (par >= 0) {
          const Entry p = beam[par];
          from_parent = (e.last == p.last) ? p.pb + lpl : lse2d(p.pb, p.pnb) + lpl;
        }
        const double own = e.pnb + lpl;
        if (par >= 0 && par < tid) pnb = lse2d(lse2d(DNEG, from_parent), own);   // dict order
        else if (par >= 0) pnb = lse2d(lse2d(DNEG, own), from_parent);
        else pnb = own;
        if (e.last != blank) {
          unsigned a = (unsigned)e.last * (2u * W) + 2u * tid + 1u;
          if (par >= 0) a = min(a, (unsigned)e.last * (2u * W) + 2u * par);
          first = min(first, a);
        }
      }
      s_pb[tid] = pb; s_pnb[tid] = pnb;
      s_key[tid] = okey(lse2d(pb, pnb));
      s_idx[tid] = ~first;
    }
    __syncthreads();
    // ---- 2b. classes that can still reach the top W (see the header): threshold = (W+1)-th largest
    //      non-blank log-probability, found with a byte-wise radix select over the frame's C values
    if (tid < nb) s_L[tid] = lse2d(beam[tid].pb, beam[tid].pnb);
    const int R = W + 1;
    if (C - 1 > 4 * R) {   // small vocabularies: the select + compaction passes cost more than they prune
      if (tid == 0) { sel_key = 0; sel_remaining = (unsigned)R; }
      __syncthreads();
      for (int byte = 7; byte >= 0; --byte) {
        hist[tid] = 0;
        __syncthreads();
        const unsigned long long pk = sel_key;
        const int sh = (byte + 1) * 8;
        for (int c = tid; c < C; c += BEAM_THREADS) {
          if (c == blank) continue;
          const unsigned long long k = okey(lp[c]);
          if (sh >= 64 || (k >> sh) == (pk >> sh)) atomicAdd(&hist[(unsigned)((k >> (byte * 8)) & 0xff)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned rem = sel_remaining;
          int dgt = 255;
          for (; dgt > 0; --dgt) {
            if (hist[dgt] >= rem) break;
            rem -= hist[dgt];
          }
          sel_remaining = rem;
          sel_key |= ((unsigned long long)dgt) << (byte * 8);
        }
        __syncthreads();
      }
      if (tid == 0) {
        const double th = unokey(sel_key);
        s_thr = th - 1e-9 * (1.0 + fabs(th));
      }
    } else if (tid == 0) {
      s_thr = DNEG;
    }
    __syncthreads();
    {   // ordered compaction of the kept classes
      const double thr = s_thr;
      int base = 0;
      for (int c0 = 0; c0 < C; c0 += BEAM_THREADS) {
        const int c = c0 + tid;
        const bool keep = c < C && c != blank && lp[c] >= thr;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wcnt[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += (int)wcnt[w];
        const int pos = off + (int)__popcll(bal & ((1ull << lane) - 1ull));
        if (c < C) kpos[c] = keep ? pos : -1;
        if (keep) kc[pos] = c;
        base += (int)(wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
        __syncthreads();
      }
      if (tid == 0) s_K = base;
    }
    __syncthreads();
    const int K = s_K;
    // ---- 3. extension totals over the kept classes (p_b = -inf, so total = p_nb contribution)
    for (int e = tid; e < nb * K; e += BEAM_THREADS) {
      const int j = e / K, c = kc[e % K];
      tot[e] = ((c == beam[j].last) ? beam[j].pb : s_L[j]) + lp[c];
    }
    __syncthreads();
    if (tid < nb && s_parent[tid] >= 0 && beam[tid].last != blank) {   // merged pairs are not new prefixes
      const int kp = kpos[beam[tid].last];
      if (kp >= 0) tot[(size_t)s_parent[tid] * K + kp] = NAN;           // NaN = excluded marker
    }
    __syncthreads();
    // ---- 4. exact top-W over nb stay + nb*K extension candidates (excluded ones skipped)
    const int M = nb + nb * K;
    auto cand = [&](int id, unsigned long long& key, unsigned& nidx) -> bool {
      if (id < nb) { key = s_key[id]; nidx = s_idx[id]; return true; }
      const int e = id - nb;
      const double v = tot[e];
      if (v != v) return false;
      const int j = e / K, c = kc[e % K];
      key = okey(v);
      nidx = ~((unsigned)c * (2u * W) + 2u * j);
      return true;
    };
    // count valid candidates
    if (tid == 0) { s_nw = 0; sel_key = 0; sel_nidx = 0; }
    __syncthreads();
    {
      int cnt = 0;
      for (int id = tid; id < M; id += BEAM_THREADS) { unsigned long long k; unsigned n; cnt += cand(id, k, n) ? 1 : 0; }
      atomicAdd(&s_nw, cnt);
    }
    __syncthreads();
    const int valid = s_nw;
    const int want = min(W, valid);
    __syncthreads();
    if (tid == 0) { sel_remaining = want; s_nw = 0; }
    __syncthreads();
    if (valid > want) {
      for (int byte = 11; byte >= 0; --byte) {
        hist[tid] = 0;   // BEAM_THREADS == 256
        __syncthreads();
        const unsigned long long pk = sel_key;
        const unsigned pn = sel_nidx;
        for (int id = tid; id < M; id += BEAM_THREADS) {
          unsigned long long k; unsigned n;
          if (!cand(id, k, n)) continue;
          // matches the already fixed more-significant bytes?
          bool ok;
          if (byte >= 4) {
            const int sh = (byte - 4 + 1) * 8;
            ok = (sh >= 64) ? true : ((k >> sh) == (pk >> sh));
          } else {
            const int sh = (byte + 1) * 8;
            ok = (k == pk) && ((sh >= 32) ? true : ((n >> sh) == (pn >> sh)));
          }
          if (ok) atomicAdd(&hist[digit_of(k, n, byte)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned rem = sel_remaining;
          int d = 255;
          for (; d > 0; --d) {
            if (hist[d] >= rem) break;
            rem -= hist[d];
          }
          sel_remaining = rem;
          if (byte >= 4) sel_key |= ((unsigned long long)d) << ((byte - 4) * 8);
          else sel_nidx |= ((unsigned)d) << (byte * 8);
        }
        __syncthreads();
      }
    }
    // collect winners: composite >= threshold (all valid ones when valid <= W)
    {
      const unsigned long long tk = sel_key;
      const unsigned tn = sel_nidx;
      for (int id = tid; id < M; id += BEAM_THREADS) {
        unsigned long long k; unsigned n;
        if (!cand(id, k, n)) continue;
        const bool win = (valid <= want) || (k > tk) || (k == tk && n >= tn);
        if (win) {
          const int pos = atomicAdd(&s_nw, 1);
          if (pos < BEAM_MAX) { w_key[pos] = k; w_nidx[pos] = n; w_src[pos] = id; }
        }
      }
    }
    __syncthreads();
    const int nw = min(s_nw, want);
    // ---- 5. rank winners (composite keys are distinct) and build the next beam
    if (tid < nw) {
      const unsigned long long k = w_key[tid];
      const unsigned n = w_nidx[tid];
      int rank = 0;
      for (int o = 0; o < nw; ++o) rank += (w_key[o] > k) || (w_key[o] == k && w_nidx[o] > n);
      const int id = w_src[tid];
      Entry ne;
      if (id < nb) {
        ne = beam[id];
        ne.pb = s_pb[id]; ne.pnb = s_pnb[id];
      } else {
        const int e = id - nb, j = e / K, c = kc[e % K];
        const Entry p = beam[j];
        ne.pb = DNEG; ne.pnb = tot[e];
        ne.phash = p.hash; ne.hash = hmix(p.hash, c);
        ne.len = p.len + 1; ne.last = c;
        const int node = atomicAdd(&s_nodes, 1);
        nodes[node] = make_int2(p.node, c);
        ne.node = node;
      }
      nbeam[rank] = ne;
    }
    __syncthreads();
    if (tid < nw) beam[tid] = nbeam[tid];
    if (tid == 0) s_nb = nw;
    __syncthreads();
  }

  // best hypothesis = beam[0]; walk the trie back
  if (tid == 0) {
    const Entry e = beam[0];
    const int n = e.len;
    int node = e.node;
    for (int i = n - 1; i >= 0; --i) {
      const int2 nd = nodes[node];
      out_labels[(size_t)b * T + i] = nd.y;
      node = nd.x;
    }
    for (int i = n; i < T; ++i) out_labels[(size_t)b * T + i] = -1;
    out_len[b] = n;
    out_score[b] = -lse2d(e.pb, e.pnb);
  }
}

struct BeamWs { size_t tot, nodes, total; };
inline BeamWs beam_ws_layout(int T, int B, int C, int W) {
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  BeamWs w;
  size_t o = 0;
  w.tot = o;   o += al((size_t)B * W * C * sizeof(double));
  w.nodes = o; o += al((size_t)B * ((size_t)T * W + 1) * sizeof(int2));
  w.total = o;
  return w;
}

}  // namespace

extern "C" size_t asr_ctc_beam_workspace_bytes(int T, int B, int C, int beam_width) {
  if (T < 0 || B < 0 || C < 1 || beam_width < 1) return 0;
  return beam_ws_layout(T, B, C, beam_width).total;
}

extern "C" int asr_ctc_beam_decode(asr_handle* h, const float* logits, int T, int B, int C,
                                   const int32_t* seq_len, int blank, int beam_width,
                                   int32_t* out_labels, int32_t* out_len, double* out_score,
                                   void* workspace, size_t workspace_bytes, asr_stream s) {
  if (!h) return ASR_ERR_INVALID_ARG;
  if (!logits || !seq_len || !out_labels || !out_len || !out_score || T <= 0 || B <= 0 || C < 2 ||
      blank < 0 || blank >= C || beam_width < 1)
    ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_ctc_beam_decode: bad args T=%d B=%d C=%d beam=%d", T, B, C, beam_width);
  if (beam_width > BEAM_MAX)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_beam_decode: beam_width %d > %d", beam_width, BEAM_MAX);
  if ((double)C * 2.0 * beam_width >= 4294967295.0)
    ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_beam_decode: C*beam too large");
  const BeamWs w = beam_ws_layout(T, B, C, beam_width);
  if (!workspace || workspace_bytes < w.total)
    ASR_FAIL(h, ASR_ERR_WORKSPACE, "asr_ctc_beam_decode: workspace %zu < %zu bytes", workspace_bytes, w.total);
  const size_t lds = (size_t)C * (sizeof(double) + 2 * sizeof(int));
  if (lds > 96 * 1024) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "asr_ctc_beam_decode: C=%d too large for LDS", C);
  (void)hipFuncSetAttribute((const void*)ctc_beam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  char* ws = (char*)workspace;
  hipLaunchKernelGGL(ctc_beam_kernel, dim3(B), dim3(BEAM_THREADS), lds, (hipStream_t)s, logits, T, B, C, seq_len,
                     blank, beam_width, (double*)(ws + w.tot), (int2*)(ws + w.nodes), out_labels, out_len,
                     out_score);
  ASR_CHECK_LAUNCH(h, "asr_ctc_beam_decode");
  return ASR_OK;
}
