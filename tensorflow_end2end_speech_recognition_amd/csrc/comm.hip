// Data-parallel collective of the training step in the C ABI: the tower mean of
// utils/training/multi_gpu.py:13-48 (average_gradients) / examples/librispeech/training/train_ctc.py:112-147 as ONE
// RCCL all-reduce(sum) over the flat fp32 gradient buffer followed by x 1/N, one process per GPU over xGMI.
//
// librccl.so is opened at run time (dlopen) rather than linked: the host process usually has one loaded already
// (PyTorch ships its own), and both must be the SAME library instance to share a device context cleanly -- the
// caller names it (asr_comm_set_library; the Python front end passes torch's), with the ROCm one as the default.
// Bootstrap follows NCCL's contract: rank 0 obtains a 128-byte unique id (asr_comm_unique_id), the host program
// distributes it by whatever side channel it has (torch.distributed store, MPI, a file), every rank calls
// asr_comm_init(rank, world, id).
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>

namespace {

constexpr int kNcclUniqueIdBytes = 128;
struct nccl_uid { char internal[kNcclUniqueIdBytes]; };
typedef void* nccl_comm_t;
typedef int nccl_result_t;                         // ncclSuccess = 0
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

struct RcclApi {
  void* lib = nullptr;
  nccl_result_t (*GetUniqueId)(nccl_uid*) = nullptr;
  nccl_result_t (*CommInitRank)(nccl_comm_t*, int, nccl_uid, int) = nullptr;
  nccl_result_t (*CommDestroy)(nccl_comm_t) = nullptr;
  nccl_result_t (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(nccl_result_t) = nullptr;
  char path[512] = {0};
  char err[256] = {0};
};
RcclApi g_rccl;
char g_lib_override[512] = {0};

bool rccl_load() {
  if (g_rccl.lib) return true;
  const char* cands[4] = {nullptr, nullptr, nullptr, nullptr};
  int n = 0;
  if (g_lib_override[0]) cands[n++] = g_lib_override;
  if (const char* e = getenv("ASR_RCCL_PATH")) cands[n++] = e;
  cands[n++] = "librccl.so.1";
  cands[n++] = "librccl.so";
  for (int i = 0; i < n; ++i) {
    void* lib = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) continue;
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce) {
      snprintf(a.path, sizeof(a.path), "%s", cands[i]);
      g_rccl = a;
      return true;
    }
    dlclose(lib);
  }
  snprintf(g_rccl.err, sizeof(g_rccl.err), "librccl.so not loadable (tried the override, $ASR_RCCL_PATH, librccl.so.1): %s",
           dlerror() ? dlerror() : "symbols missing");
  return false;
}

// x[0..n) *= s.  Any 4-byte aligned buffer (an external C-ABI caller need not align to 16 bytes; ADVICE / VERDICT r04): a
// scalar head up to the first 16-byte boundary, 16-byte vectors, a scalar tail.
__global__ void comm_scale_kernel(float* __restrict__ x, size_t n, float s) {
  size_t head = ((16u - (unsigned)((uintptr_t)x & 15u)) & 15u) / 4u;
  if (head > n) head = n;
  const size_t n4 = (n - head) / 4;
  f32x4_t* x4 = reinterpret_cast<f32x4_t*>(x + head);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4_t v = x4[i];
    v[0] *= s; v[1] *= s; v[2] *= s; v[3] *= s;
    x4[i] = v;
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) x[threadIdx.x] *= s;
    const size_t tail0 = head + n4 * 4;
    if (threadIdx.x >= 32 && threadIdx.x - 32 < n - tail0) x[tail0 + threadIdx.x - 32] *= s;
  }
}

}  // namespace

struct asr_comm {
  asr_handle* h;
  nccl_comm_t comm;
  int rank, world;
};

extern "C" int asr_comm_set_library(const char* path) {
  if (!path || strlen(path) >= sizeof(g_lib_override)) return ASR_ERR_INVALID_ARG;
  if (g_rccl.lib) return strcmp(path, g_rccl.path) == 0 ? ASR_OK : ASR_ERR_UNSUPPORTED;   // already bound
  snprintf(g_lib_override, sizeof(g_lib_override), "%s", path);
  return ASR_OK;
}

extern "C" int asr_comm_unique_id(void* id128_host) {
  if (!id128_host) return ASR_ERR_INVALID_ARG;
  if (!rccl_load()) return ASR_ERR_UNSUPPORTED;
  nccl_uid id;
  if (g_rccl.GetUniqueId(&id) != 0) return ASR_ERR_HIP;
  memcpy(id128_host, id.internal, kNcclUniqueIdBytes);
  return ASR_OK;
}

extern "C" int asr_comm_init(asr_comm** out, asr_handle* h, int rank, int world, const void* id128_host) {
  if (!out || !h || !id128_host || world < 1 || rank < 0 || rank >= world) return ASR_ERR_INVALID_ARG;
  if (!rccl_load()) ASR_FAIL(h, ASR_ERR_UNSUPPORTED, "%s", g_rccl.err);
  if (hipSetDevice(h->device) != hipSuccess) ASR_FAIL(h, ASR_ERR_HIP, "asr_comm_init: hipSetDevice(%d)", h->device);
  nccl_uid id;
  memcpy(id.internal, id128_host, kNcclUniqueIdBytes);
  nccl_comm_t c = nullptr;
  const nccl_result_t r = g_rccl.CommInitRank(&c, world, id, rank);
  if (r != 0)
    ASR_FAIL(h, ASR_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world,
             g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
  asr_comm* cm = new asr_comm();
  cm->h = h;
  cm->comm = c;
  cm->rank = rank;
  cm->world = world;
  *out = cm;
  return ASR_OK;
}

extern "C" int asr_comm_destroy(asr_comm* c) {
  if (!c) return ASR_OK;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return ASR_OK;
}

extern "C" int asr_comm_info(asr_comm* c, int* rank, int* world) {
  if (!c) return ASR_ERR_INVALID_ARG;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return ASR_OK;
}

// test hook: the scaling pass of asr_allreduce_mean on its own (it only runs with world > 1, which a 1-GPU box cannot reach)
extern "C" int asr_debug_comm_scale(float* buf, size_t n, float s, asr_stream st) {
  if (!buf || !n) return ASR_ERR_INVALID_ARG;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(comm_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, buf, n, s);
  return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ERR_HIP;
}

// buf[i] <- (sum over ranks of buf[i]) / world, in place, enqueued on `s` (every rank, same n).
extern "C" int asr_allreduce_mean(asr_comm* c, float* buf, size_t n, asr_stream s) {
  if (!c || !c->h) return ASR_ERR_INVALID_ARG;
  asr_handle* h = c->h;
  if (!buf && n) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_allreduce_mean: null buffer");
  if (((uintptr_t)buf) & 3u) ASR_FAIL(h, ASR_ERR_INVALID_ARG, "asr_allreduce_mean: buffer not 4-byte aligned");
  if (!n) return ASR_OK;
  hipStream_t st = (hipStream_t)s;
  const nccl_result_t r = g_rccl.AllReduce(buf, buf, n, kNcclFloat32, kNcclSum, c->comm, st);
  if (r != 0)
    ASR_FAIL(h, ASR_ERR_HIP, "ncclAllReduce(%zu floats): %s", n, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
  if (c->world > 1) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(comm_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, buf, n, 1.0f / (float)c->world);
    ASR_CHECK_LAUNCH(h, "asr_allreduce_mean");
  }
  return ASR_OK;
}
