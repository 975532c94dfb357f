// vk_rccl.cpp - the multi-GPU layer of libvictor_hip.so (include/victor_hip.h: vk_comm_*): RCCL through dlopen, one
// ncclAllGather of log-likelihoods per batch on the context's stream.  Host code only (HIP runtime API, no device code);
// compiled by the host compiler.

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vk_host.h"

using vkh::fail;

size_t vkh::ctx_layout_rccl(size_t* last_offset) {
  if (last_offset) *last_offset = offsetof(vk_ctx, spin_timeouts);
  return sizeof(vk_ctx);
}

namespace {

// ---- RCCL via dlopen -------------------------------------------------------------------------------
typedef struct { char internal[VK_COMM_ID_BYTES]; } rccl_id_t;
typedef int (*fn_get_id)(rccl_id_t*);
typedef int (*fn_init_rank)(void**, int, rccl_id_t, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef int (*fn_init_all)(void**, int, const int*);
typedef int (*fn_group)(void);
typedef const char* (*fn_errstr)(int);

// Path of the shared object that defines `addr` (empty if unknown)
std::string object_of(const void* addr) {
  Dl_info info;
  if (addr && dladdr(addr, &info) && info.dli_fname) return info.dli_fname;
  return std::string();
}

std::string dir_of(const std::string& path) {
  const size_t cut = path.find_last_of('/');
  return cut == std::string::npos ? std::string() : path.substr(0, cut);
}

// The HIP runtime this library is actually running on.  libamdhip64 has one soname (libamdhip64.so.7) in every ROCm 7
// install, so whichever copy the process mapped first serves everybody: /opt/rocm's when this library is loaded into a
// fresh interpreter, PyTorch's bundled copy when torch was imported before (torch.distributed launchers).
std::string hip_runtime_path() { return object_of(reinterpret_cast<const void*>(&hipGetDeviceCount)); }

// RCCL must come from the same ROCm install as that runtime (its kernels and its HIP calls are built against it), so
// look next to the mapped libamdhip64 first and only then fall back to the loader's search order.
void* open_rccl(std::string* how = nullptr) {
  static void* lib = nullptr;
  static std::string chosen;
  if (!lib) {
    std::vector<std::string> names;
    // development override (tests/rccl_double): like every other VICTOR_HIP_* switch it is honoured only with VICTOR_HIP_DEV=1 -
    // a variable inherited from somebody's shell must never swap the collective library of a production run
    const char* dev = getenv("VICTOR_HIP_DEV");
    if (dev && strcmp(dev, "1") == 0)
      if (const char* env = getenv("VICTOR_HIP_RCCL_LIB")) names.push_back(env);
    const std::string dir = dir_of(hip_runtime_path());
    if (!dir.empty()) {
      names.push_back(dir + "/librccl.so.1");
      names.push_back(dir + "/librccl.so");
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& nm : names) {
      lib = dlopen(nm.c_str(), RTLD_NOW | RTLD_GLOBAL);
      if (lib) {
        chosen = nm;
        break;
      }
    }
  }
  if (how) *how = chosen;
  return lib;
}

}  // namespace

extern "C" {

// ---- RCCL -----------------------------------------------------------------------------------------
int vk_comm_unique_id(char* id_out) {
  void* lib = open_rccl();
  if (!lib || !id_out) return VK_E_RCCL;
  auto get = (fn_get_id)dlsym(lib, "ncclGetUniqueId");
  if (!get) return VK_E_RCCL;
  rccl_id_t id;
  if (get(&id) != 0) return VK_E_RCCL;
  memcpy(id_out, id.internal, VK_COMM_ID_BYTES);
  return VK_OK;
}

int vk_comm_init(vk_ctx* ctx, const char* id, int rank, int nranks) {
  if (!ctx || !id) return VK_E_ARG;
  void* lib = open_rccl();
  if (!lib) return fail(ctx, VK_E_RCCL, "cannot load librccl (looked next to %s first): %s", hip_runtime_path().c_str(), dlerror());
  auto init = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
  if (!init) return fail(ctx, VK_E_RCCL, "ncclCommInitRank not found");
  VK_HIP(ctx, hipSetDevice(ctx->device));
  rccl_id_t uid;
  memcpy(uid.internal, id, VK_COMM_ID_BYTES);
  int rc = init(&ctx->comm, nranks, uid, rank);
  if (rc != 0) {
    auto es = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    ctx->comm = nullptr;
    return fail(ctx, VK_E_RCCL, "ncclCommInitRank failed: %s", es ? es(rc) : "?");
  }
  ctx->rccl_lib = lib;
  ctx->comm_nranks = nranks;
  return VK_OK;
}

// An all-gather of host data that the caller collects LATER: the rows go into pinned memory, upload, ncclAllGather and download
// are enqueued on the context's stream, nothing waits.  vk_comm_allgather_host_finish waits for the download's event - by
// then, one block of walker steps later, long past - and hands the gathered rows over.
int vk_comm_allgather_host_begin(vk_ctx* ctx, const double* send, int64_t count) {
  if (!ctx || !ctx->comm) return fail(ctx, VK_E_RCCL, "communicator not initialised");
  if (!send || count < 1) return fail(ctx, VK_E_ARG, "vk_comm_allgather_host_begin: NULL buffer or count < 1");
  if (ctx->comm_begun != 0) return fail(ctx, VK_E_ARG, "vk_comm_allgather_host_begin: the previous gather has not been collected");
  VK_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t slots = 1 + (int64_t)ctx->comm_nranks;
  if (count > ctx->comm_cap) {
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
    if (ctx->d_comm) (void)hipFree(ctx->d_comm);
    ctx->h_comm = ctx->d_comm = nullptr;
    ctx->comm_cap = 0;
    VK_HIP(ctx, hipHostMalloc((void**)&ctx->h_comm, (size_t)slots * count * sizeof(double), hipHostMallocDefault));
    VK_HIP(ctx, hipMalloc((void**)&ctx->d_comm, (size_t)slots * count * sizeof(double)));
    ctx->comm_cap = count;
  }
  if (!ctx->ev_comm) VK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_comm, hipEventDisableTiming));
  memcpy(ctx->h_comm, send, (size_t)count * sizeof(double));
  VK_HIP(ctx, hipMemcpyAsync(ctx->d_comm, ctx->h_comm, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const int rc = vk_comm_allgather_async(ctx, ctx->d_comm, ctx->d_comm + ctx->comm_cap, count);
  if (rc) return rc;
  VK_HIP(ctx, hipMemcpyAsync(ctx->h_comm + ctx->comm_cap, ctx->d_comm + ctx->comm_cap, (size_t)ctx->comm_nranks * count * sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
  VK_HIP(ctx, hipEventRecord(ctx->ev_comm, ctx->stream));
  ctx->comm_begun = count;
  return VK_OK;
}

int vk_comm_allgather_host_finish(vk_ctx* ctx, double* recv) {
  if (!ctx) return VK_E_ARG;
  const int64_t count = ctx->comm_begun;
  if (count == 0) return fail(ctx, VK_E_ARG, "vk_comm_allgather_host_finish: nothing was begun on this context");
  if (!recv) return fail(ctx, VK_E_ARG, "vk_comm_allgather_host_finish: NULL buffer");
  ctx->comm_begun = 0;
  VK_HIP(ctx, hipSetDevice(ctx->device));
  VK_HIP(ctx, hipEventSynchronize(ctx->ev_comm));
  memcpy(recv, ctx->h_comm + ctx->comm_cap, (size_t)ctx->comm_nranks * count * sizeof(double));
  return VK_OK;
}

int vk_comm_allgather_async(vk_ctx* ctx, const double* d_send, double* d_recv, int64_t count) {
  if (!ctx || !ctx->comm) return fail(ctx, VK_E_RCCL, "communicator not initialised");
  if (!d_send || !d_recv || count < 0) return fail(ctx, VK_E_ARG, "vk_comm_allgather_async: NULL buffer or negative count");
  auto ag = (fn_allgather)dlsym(ctx->rccl_lib, "ncclAllGather");
  if (!ag) return fail(ctx, VK_E_RCCL, "ncclAllGather not found");
  const int kNcclDouble = 8;  // ncclFloat64 in rccl.h
  int rc = ag(d_send, d_recv, (size_t)count, kNcclDouble, ctx->comm, ctx->stream);
  if (rc != 0) return fail(ctx, VK_E_RCCL, "ncclAllGather failed (%d)", rc);
  return VK_OK;
}

int vk_device_bus_id(const vk_ctx* ctx, char* buf, size_t len) {
  if (!ctx || !buf || len < 16) return VK_E_ARG;
  if (hipDeviceGetPCIBusId(buf, (int)len, ctx->device) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(buf, len, "device%d", ctx->device);
  }
  return VK_OK;
}

int vk_comm_init_all(vk_ctx* const* ctxs, int32_t n) {
  if (!ctxs || n < 1 || !ctxs[0]) return VK_E_ARG;
  vk_ctx* lead = ctxs[0];
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return fail(lead, VK_E_ARG, "context %d is NULL", i);
    if (ctxs[i]->comm) return fail(lead, VK_E_ARG, "context %d already has a communicator", i);
    devs[i] = ctxs[i]->device;
    for (int j = 0; j < i; ++j)
      if (devs[j] == devs[i]) return fail(lead, VK_E_RCCL, "contexts %d and %d share device %d: RCCL needs one device per rank", j, i, devs[i]);
  }
  void* lib = open_rccl();
  if (!lib) return fail(lead, VK_E_RCCL, "cannot load librccl (looked next to %s first): %s", hip_runtime_path().c_str(), dlerror());
  auto init = (fn_init_all)dlsym(lib, "ncclCommInitAll");
  if (!init) return fail(lead, VK_E_RCCL, "ncclCommInitAll not found");
  std::vector<void*> comms(n, nullptr);
  const int rc = init(comms.data(), n, devs.data());
  if (rc != 0) {
    auto es = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    return fail(lead, VK_E_RCCL, "ncclCommInitAll failed: %s", es ? es(rc) : "?");
  }
  for (int i = 0; i < n; ++i) {
    ctxs[i]->comm = comms[i];
    ctxs[i]->rccl_lib = lib;
    ctxs[i]->comm_nranks = n;
  }
  return VK_OK;
}

int vk_comm_allgather_group_async(vk_ctx* const* ctxs, int32_t n, const double* const* d_send, double* const* d_recv,
                                  int64_t count) {
  if (!ctxs || n < 1 || !ctxs[0] || !d_send || !d_recv || count < 0) return VK_E_ARG;
  vk_ctx* lead = ctxs[0];
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i] || !ctxs[i]->comm) return fail(lead, VK_E_RCCL, "context %d has no communicator", i);
    if (!d_send[i] || !d_recv[i]) return fail(lead, VK_E_ARG, "vk_comm_allgather_group_async: NULL buffer for context %d", i);
  }
  auto ag = (fn_allgather)dlsym(lead->rccl_lib, "ncclAllGather");
  auto gs = (fn_group)dlsym(lead->rccl_lib, "ncclGroupStart");
  auto ge = (fn_group)dlsym(lead->rccl_lib, "ncclGroupEnd");
  if (!ag || !gs || !ge) return fail(lead, VK_E_RCCL, "ncclAllGather / ncclGroupStart / ncclGroupEnd not found");
  const int kNcclDouble = 8;  // ncclFloat64 in rccl.h
  int rc = gs();
  for (int i = 0; i < n && rc == 0; ++i) rc = ag(d_send[i], d_recv[i], (size_t)count, kNcclDouble, ctxs[i]->comm, ctxs[i]->stream);
  const int rc_end = ge();
  if (rc != 0 || rc_end != 0) return fail(lead, VK_E_RCCL, "grouped ncclAllGather failed (%d, %d)", rc, rc_end);
  return VK_OK;
}

// Which HIP runtime and which RCCL this process ended up with (a multi-GPU record must be diagnosable from its JSON line)
int vk_comm_info(char* buf, size_t len) {
  if (!buf || len == 0) return VK_E_ARG;
  int hip_rt = 0, hip_drv = 0, rccl_ver = 0;
  (void)hipRuntimeGetVersion(&hip_rt);
  (void)hipDriverGetVersion(&hip_drv);
  std::string asked, rccl_path;
  void* lib = open_rccl(&asked);
  if (lib) {
    typedef int (*fn_ver)(int*);
    if (auto ver = (fn_ver)dlsym(lib, "ncclGetVersion")) (void)ver(&rccl_ver);
    rccl_path = object_of(dlsym(lib, "ncclAllGather"));
  }
  const std::string hip_path = hip_runtime_path();
  const bool same_dir = lib && !rccl_path.empty() && dir_of(rccl_path) == dir_of(hip_path);
  snprintf(buf, len,
           "{\"hip_runtime\": \"%s\", \"hip_runtime_version\": %d, \"hip_driver_version\": %d, \"built_with_hip\": \"%d.%d.%d\", "
           "\"rccl\": \"%s\", \"rccl_opened_as\": \"%s\", \"rccl_version\": %d, \"rccl_next_to_hip_runtime\": %s}",
           hip_path.c_str(), hip_rt, hip_drv, HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH,
           lib ? rccl_path.c_str() : "", asked.c_str(), rccl_ver, same_dir ? "true" : "false");
  return lib ? VK_OK : VK_E_RCCL;
}

int vk_comm_rank_info(const vk_ctx* ctx, int32_t* count, int32_t* user_rank, int32_t* device) {
  if (count) *count = -1;
  if (user_rank) *user_rank = -1;
  if (device) *device = -1;
  if (!ctx || !ctx->comm || !ctx->rccl_lib) return VK_E_RCCL;
  typedef int (*fn_comm_int)(void*, int*);
  const struct { const char* name; int32_t* out; } asks[3] = {{"ncclCommCount", count}, {"ncclCommUserRank", user_rank}, {"ncclCommCuDevice", device}};
  for (const auto& ask : asks) {
    int v = -1;
    auto fn = (fn_comm_int)dlsym(ctx->rccl_lib, ask.name);
    if (ask.out && fn && fn(ctx->comm, &v) == 0) *ask.out = v;
  }
  return VK_OK;
}

int vk_comm_destroy(vk_ctx* ctx) {
  if (!ctx || !ctx->comm) return VK_OK;
  auto destroy = (fn_destroy)dlsym(ctx->rccl_lib, "ncclCommDestroy");
  if (destroy) destroy(ctx->comm);
  ctx->comm = nullptr;
  return VK_OK;
}

}  // extern "C"
