// round 5 measurement (not part of the library): what does a chip-wide barrier between dependent phases of ONE persistent
// kernel cost on MI355X, against the boundary between two dependent kernel launches on an in-order stream?  This is the
// number the "one persistent kernel per attention decoder loop" question turns on: a decoder step has
// 4 (forward) / 4 (backward) dependent phases that each want the whole chip.
//   build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// flat: one monotonic counter, every workgroup adds 1 and spins until it reads it * nblk
__global__ __launch_bounds__(256) void flat_kernel(unsigned* ctr, int iters, float* sink, const float* src, int work) {
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    for (int w = 0; w < work; ++w) acc += src[(blockIdx.x * 256 + threadIdx.x + w * 65536) & 0xFFFFF];   // a phase's loads
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);      // agent scope by default for global atomics in HIP
      const unsigned want = (unsigned)it * gridDim.x;
      while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (acc == 12345.f) sink[0] = acc;
}
// two-level: workgroups b % 8 == x share XCD x (1-D grid, round-robin placement) and its L2; they arrive on a per-XCD counter,
// the last of an XCD arrives on the global counter, then releases its XCD through a per-XCD flag once the global count is full
__global__ __launch_bounds__(256) void tree_kernel(unsigned* loc, unsigned* glob, unsigned* rel, int iters, float* sink,
                                                   const float* src, int work) {
  const int x = blockIdx.x & 7;
  const unsigned nloc = (gridDim.x + 7 - x) / 8;
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    for (int w = 0; w < work; ++w) acc += src[(blockIdx.x * 256 + threadIdx.x + w * 65536) & 0xFFFFF];
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __atomic_fetch_add(loc + x * 32, 1u, __ATOMIC_ACQ_REL);
      if (old + 1 == (unsigned)it * nloc) {               // last of this XCD
        __atomic_fetch_add(glob, 1u, __ATOMIC_ACQ_REL);
        while (__atomic_load_n(glob, __ATOMIC_ACQUIRE) < (unsigned)it * 8u) __builtin_amdgcn_s_sleep(1);
        __atomic_store_n(rel + x * 32, (unsigned)it, __ATOMIC_RELEASE);
      } else {
        while (__atomic_load_n(rel + x * 32, __ATOMIC_ACQUIRE) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (acc == 12345.f) sink[0] = acc;
}
// flag array: workgroup b stores the phase number into ITS word (write-through sc1 store, no read-modify-write), one wave per
// workgroup polls all words with ONE 16-byte L1-bypassing load per lane (256 flags = 1 KB) -- the exchange idiom of the
// recurrence clusters, chip-wide
__global__ __launch_bounds__(256) void flags_kernel(unsigned* flags, int iters, float* sink, const float* src, int work) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  float acc = 0.f;
  const int lane = threadIdx.x;
  for (int it = 1; it <= iters; ++it) {
    for (int w = 0; w < work; ++w) acc += src[(blockIdx.x * 256 + threadIdx.x + w * 65536) & 0xFFFFF];
    __syncthreads();
    if (threadIdx.x < 64) {
      if (lane == 0) {
        unsigned* p = flags + blockIdx.x; unsigned v = (unsigned)it;
        asm volatile("s_waitcnt vmcnt(0)\n\tglobal_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
      }
      const unsigned* q = flags + lane * 4;
      bool ok;
      do {
        u4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
        ok = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) ok = ok && (lane * 4 + j >= (int)gridDim.x || v[j] >= (unsigned)it);
      } while (!__all(ok));
    }
    __syncthreads();
  }
  if (acc == 12345.f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void phase_kernel(float* sink, const float* src, int work) {
  float acc = 0.f;
  for (int w = 0; w < work; ++w) acc += src[(blockIdx.x * 256 + threadIdx.x + w * 65536) & 0xFFFFF];
  if (acc == 12345.f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int iters = 2000;
  unsigned* d; float *sink, *src;
  CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&src, (1 << 20) * 4));
  CK(hipMemset(src, 0, (1 << 20) * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int work : {0, 8}) for (int nblk : {64, 128, 256}) {
    float ms[4];
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(d, 0, 4096 * 4, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(flat_kernel, dim3(nblk), dim3(256), 0, st, d, iters, sink, src, work);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
      CK(hipMemsetAsync(d, 0, 4096 * 4, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(tree_kernel, dim3(nblk), dim3(256), 0, st, d, d + 1024, d + 2048, iters, sink, src, work);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
      CK(hipMemsetAsync(d, 0, 4096 * 4, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(flags_kernel, dim3(nblk), dim3(256), 0, st, d, iters, sink, src, work);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[3], e0, e1));
      CK(hipEventRecord(e0, st));
      for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(phase_kernel, dim3(nblk), dim3(256), 0, st, sink, src, work);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[2], e0, e1));
    }
    printf("work %d loads, %3d workgroups: atomic counter %.2f us / phase, two-level counters %.2f, flag array (sc1 store + "
           "one 16-byte poll per lane) %.2f, separate launches %.2f\n", work, nblk, ms[0] * 1e3 / iters, ms[1] * 1e3 / iters,
           ms[3] * 1e3 / iters, ms[2] * 1e3 / iters);
  }
  return 0;
}
