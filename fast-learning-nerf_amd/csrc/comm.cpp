// The exchange steps of the data-parallel path behind the C ABI: one all-reduce(SUM) of the flat fp32
// gradient buffer per optimiser step, one all-reduce(MAX) of the leaf-error table per subdivide epoch -- RCCL over xGMI, one
// communicator per process (= per GPU), collectives enqueued on the caller's HIP stream (no host synchronisation).
//
// The reference has no counterpart: its multi-GPU strategy is single-process nn.DataParallel around the MLP
// (nerf-ours/run_nerf.py:70,82,90), nerf++-ours uses torch DDP (ddp_train_nerf.py:150-184).  A PyTorch host reaches the same
// collectives through torch.distributed (backend "nccl" = RCCL; fast-learning-nerf_amd/parallel.py, the default); these entry
// points are the route for a host WITHOUT torch.distributed, and the standalone form of the same step.
//
// librccl is resolved at the first call (symbols already in the process -- PyTorch ships its own copy -- else librccl.so.1):
// libfastnerf.so itself has no link-time dependency on it and loads on a box without RCCL.
#include <dlfcn.h>
#include <link.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "../../include/fastnerf.h"
#include "common.h"

namespace {
struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

template <typename F>
bool sym(void* h, const char* name, F& f) {
  f = reinterpret_cast<F>(dlsym(h, name));
  return f != nullptr;
}
// an RCCL that is ALREADY in the process but not in the global symbol scope (PyTorch loads its bundled torch/lib/librccl.so as an
// RTLD_LOCAL dependency of libtorch_hip.so): found by walking the loaded objects, re-opened with RTLD_NOLOAD -- the same copy,
// never a second one next to it.  The callback ONLY COPIES NAMES: glibc holds its loader lock for the whole walk, and a dlopen from
// inside it orders the loader's locks opposite to a concurrent dlopen on another thread (torch and HIP load code objects lazily).
struct LoadedNames {
  static constexpr int kMax = 8;
  char path[kMax][1024];
  int n = 0;
};
int collect_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  auto* names = static_cast<LoadedNames*>(out);
  const char* name = info->dlpi_name;
  if (name && std::strstr(name, "librccl") && names->n < LoadedNames::kMax && std::strlen(name) < sizeof(names->path[0])) {
    std::strcpy(names->path[names->n++], name);
  }
  return 0;
}
void* open_loaded_rccl() {
  LoadedNames names;
  dl_iterate_phdr(collect_loaded_rccl, &names);
  for (int i = 0; i < names.n; ++i) {      // after the walk, outside the loader lock
    void* h = dlopen(names.path[i], RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    if (h == nullptr) continue;
    if (dlsym(h, "ncclCommInitRank")) return h;
    dlclose(h);                            // (RTLD_NOLOAD took a reference: give it back for an object that is not an RCCL)
  }
  return nullptr;
}
// Resolution order: (1) FASTNERF_RCCL_LIB=<path> if set; (2) the global symbol scope; (3) a librccl already loaded in the process
// (RTLD_NOLOAD); (4) only then the system librccl.so.1 -- a SECOND RCCL beside PyTorch's would come from another ROCm build than the
// process's HIP runtime, so it is the last resort and only reached when the process holds no RCCL at all.
void load_rccl() {
  void* h = nullptr;
  if (const char* p = getenv("FASTNERF_RCCL_LIB")) h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr && dlsym(RTLD_DEFAULT, "ncclCommInitRank") != nullptr) h = RTLD_DEFAULT;
  if (h == nullptr) h = open_loaded_rccl();
  if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) return;
  g_rccl.ok = sym(h, "ncclGetUniqueId", g_rccl.GetUniqueId) && sym(h, "ncclCommInitRank", g_rccl.CommInitRank) &&
              sym(h, "ncclCommDestroy", g_rccl.CommDestroy) && sym(h, "ncclAllReduce", g_rccl.AllReduce) &&
              sym(h, "ncclGetErrorString", g_rccl.GetErrorString);
}
bool rccl_ready(const char* who) {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.ok) fn::set_error("%s: librccl could not be resolved (neither in the process nor as librccl.so.1)", who);
  return g_rccl.ok;
}
#define FN_RCCL(expr)                                                                        \
  do {                                                                                       \
    ncclResult_t r_ = (expr);                                                                \
    if (r_ != ncclSuccess) {                                                                 \
      fn::set_error("%s: %s failed: %s", __func__, #expr, g_rccl.GetErrorString(r_));        \
      return -(int)r_ - 1000;                                                                \
    }                                                                                        \
  } while (0)

__global__ void scale_kernel(int64_t n, float* __restrict__ x, float s) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= s;
}
}  // namespace

struct fn_comm {
  ncclComm_t comm;
  int rank, world;
};

static_assert(FASTNERF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "fastnerf.h carries RCCL's unique-id size");

extern "C" int fastnerf_comm_unique_id(char* id) {
  FN_CHECK_ARG(id != nullptr, "id buffer of FASTNERF_COMM_ID_BYTES bytes");
  if (!rccl_ready(__func__)) return -2;
  ncclUniqueId u;
  FN_RCCL(g_rccl.GetUniqueId(&u));
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int fastnerf_comm_init(fn_comm** out, const char* id, int rank, int world) {
  FN_CHECK_ARG(out != nullptr && id != nullptr, "out / id");
  FN_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "0 <= rank < world");
  if (!rccl_ready(__func__)) return -2;
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c;
  FN_RCCL(g_rccl.CommInitRank(&c, world, u, rank));   // on the calling thread's current HIP device
  *out = new fn_comm{c, rank, world};
  return 0;
}

extern "C" int fastnerf_comm_destroy(fn_comm* c) {
  if (c == nullptr) return 0;
  if (!rccl_ready(__func__)) return -2;
  FN_RCCL(g_rccl.CommDestroy(c->comm));
  delete c;
  return 0;
}

extern "C" int fastnerf_allreduce_grads(fn_comm* c, float* grads, int64_t n, float scale, fn_stream_t stream) {
  FN_CHECK_ARG(c != nullptr && n >= 0 && (grads != nullptr || n == 0), "comm / grads");
  if (n == 0) return 0;
  if (!rccl_ready(__func__)) return -2;
  hipStream_t st = static_cast<hipStream_t>(stream);
  FN_RCCL(g_rccl.AllReduce(grads, grads, (size_t)n, ncclFloat32, ncclSum, c->comm, st));
  if (scale != 1.0f) {
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024)), dim3(256), 0, st, n, grads, scale);
    FN_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int fastnerf_allreduce_leaf_table(fn_comm* c, uint32_t* table, int64_t n, fn_stream_t stream) {
  FN_CHECK_ARG(c != nullptr && n >= 0 && (table != nullptr || n == 0), "comm / table");
  if (n == 0) return 0;
  if (!rccl_ready(__func__)) return -2;
  // bit patterns of non-negative floats are monotone in the float: MAX on them is exact and order independent
  FN_RCCL(g_rccl.AllReduce(table, table, (size_t)n, ncclUint32, ncclMax, c->comm, static_cast<hipStream_t>(stream)));
  return 0;
}

// nerf++ fork (MEAN rule, nerf++-ours/tree.py:609-632): the per-(image, leaf) fp64 sums and int32 ray counts of fastnerf_leaf_sumcount,
// summed over ranks.  The sums hold multiples of 2^-30 (train.hip): fp64 addition of them is exact, so whatever order the ring
// reduces in, every rank ends with the bit pattern a single rank would have accumulated over all rays.
extern "C" int fastnerf_allreduce_leaf_sumcount(fn_comm* c, double* sum, int32_t* count, int64_t n, fn_stream_t stream) {
  FN_CHECK_ARG(c != nullptr && n >= 0 && ((sum != nullptr && count != nullptr) || n == 0), "comm / sum / count");
  if (n == 0) return 0;
  if (!rccl_ready(__func__)) return -2;
  FN_RCCL(g_rccl.AllReduce(sum, sum, (size_t)n, ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(stream)));
  FN_RCCL(g_rccl.AllReduce(count, count, (size_t)n, ncclInt32, ncclSum, c->comm, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int fastnerf_leaf_table_reset(uint32_t* table, int64_t n, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && (table != nullptr || n == 0), "table");
  if (n) FN_HIP(hipMemsetAsync(table, 0, (size_t)n * sizeof(uint32_t), static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int fastnerf_leaf_table_read(const uint32_t* table, float* host_out, int64_t n, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && ((table != nullptr && host_out != nullptr) || n == 0), "table / host_out");
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  FN_HIP(hipMemcpyAsync(host_out, table, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  FN_HIP(hipStreamSynchronize(st));   // the reference reads the per-leaf maxima on the host right here (tree.py:629-652 walks them leaf by leaf)
  return 0;
}
