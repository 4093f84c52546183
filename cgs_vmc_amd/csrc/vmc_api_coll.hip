// Collectives: the RCCL binding, the host / device all-reduce hooks, the accumulator all-reduce.
// Split out of vmc_api.hip in round 6.
#include "vmc_ctx.hpp"

using namespace vmcapi;

namespace vmcapi {

// ---------------------------------------------------------------- collectives
// RCCL is resolved at first use with dlopen -- the copy already loaded into the process (torch's)
// if there is one -- so the library itself carries no link-time dependency on it.
struct RcclUniqueId { char internal[128]; };   // ncclUniqueId
struct Rccl {
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*get_unique_id)(RcclUniqueId*) = nullptr;
  int (*comm_init_rank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*get_error_string)(int) = nullptr;
};
std::string g_rccl_error;

const Rccl* rccl() {
  static Rccl r;
  static bool tried = false, ok = false;
  if (!tried) {
    tried = true;
    // The librccl that belongs to the HIP runtime THIS library is bound to: a process may hold two ROCm
    // stacks (torch bundles libamdhip64 / librccl next to the system's, same sonames), and a
    // communicator of the other stack's librccl would launch through the other runtime on this one's
    // streams and buffers.  So: the directory of the libamdhip64 behind our hip* symbols first.
    void* h = nullptr;
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash);
        names.push_back(dir + "/librccl.so.1");
        names.push_back(dir + "/librccl.so");
      }
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string& name : names)
      if (!h) h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h) {
      r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
      r.get_unique_id = (decltype(r.get_unique_id))dlsym(h, "ncclGetUniqueId");
      r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(h, "ncclCommInitRank");
      r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
      r.get_error_string = (decltype(r.get_error_string))dlsym(h, "ncclGetErrorString");
    }
    ok = r.all_reduce && r.get_unique_id && r.comm_init_rank && r.comm_destroy;
    if (!ok) g_rccl_error = "librccl.so / its nccl* entry points not found";
  }
  return ok ? &r : nullptr;
}

std::string rccl_error_string(const Rccl* r, int rc) {
  return (r && r->get_error_string) ? std::string(r->get_error_string(rc)) : "code " + std::to_string(rc);
}

// In-place all-reduce of n floats at device pointer buf, ordered on the ctx's stream.
//   comm != NULL                : ncclAllReduce on the stream (no host synchronisation)
//   comm == NULL, world <= 1    : nothing to do
//   comm == NULL, world  > 1    : the registered host hook, staged through pinned host memory
//   comm == NULL, world  > 1    : the device hook (the host's collective library reduces the device
//                                 buffer in stream order), else the host hook through pinned memory
// op == VMC_REDUCE_SUM_F64: buf holds n doubles.
int reduce_buffer(vmc_ctx* c, void* comm, int world, void* buf, long long n, int op) {
  const bool f64 = op == VMC_REDUCE_SUM_F64;
  if (comm) {
    const Rccl* r = rccl();
    if (!r) return fail(c, VMC_ERR_UNSUPPORTED, g_rccl_error);
    const int rc = r->all_reduce(buf, buf, (size_t)n, f64 ? /*ncclFloat64*/ 8 : /*ncclFloat32*/ 7,
                                 op == VMC_REDUCE_MAX ? /*ncclMax*/ 2 : /*ncclSum*/ 0, comm, c->stream);
    if (rc != 0) return fail(c, VMC_ERR_HIP, "ncclAllReduce: " + rccl_error_string(r, rc));
    return VMC_OK;
  }
  if (world <= 1) return VMC_OK;
  if (c->dev_reduce) {
    const int rc = c->dev_reduce(c->dev_reduce_user, buf, n, op, (void*)c->stream);
    if (rc != 0) return fail(c, VMC_ERR_HIP, "device all-reduce hook failed with code " + std::to_string(rc));
    return VMC_OK;
  }
  if (!c->host_reduce)
    return fail(c, VMC_ERR_STATE, "world_size > 1 needs an RCCL communicator, vmc_set_device_allreduce or vmc_set_host_allreduce");
  if (f64 && !(c->host_reduce_caps & VMC_HOST_REDUCE_CAP_F64))
    return fail(c, VMC_ERR_UNSUPPORTED, "the registered host all-reduce hook has not declared float64 support "
                "(vmc_set_host_allreduce_caps(ctx, VMC_HOST_REDUCE_CAP_F64)): it would be handed doubles");
  const long long nf = f64 ? 2 * n : n;          // staging size in floats
  if (nf > c->h_stage_n) {
    if (c->h_stage) hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_n = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, (size_t)nf * sizeof(float), hipHostMallocDefault));
    c->h_stage_n = nf;
  }
  HIPCHK(c, hipMemcpyAsync(c->h_stage, buf, nf * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int rc = c->host_reduce(c->host_reduce_user, c->h_stage, n, op);
  if (rc != 0) return fail(c, VMC_ERR_HIP, "host all-reduce hook failed with code " + std::to_string(rc));
  HIPCHK(c, hipMemcpyAsync(buf, c->h_stage, nf * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return VMC_OK;
}

bool sharded(void* comm, int world) { return comm != nullptr || world > 1; }

int reduce_accumulators(vmc_ctx* c, void* comm, int world) {
  if (!sharded(comm, world)) return VMC_OK;
  PROPAGATE(acc_zeros(c));
  PROPAGATE(reduce_buffer(c, comm, world, c->acc, 2 * c->P + 8, VMC_REDUCE_SUM));
  HIPCHK(c, launch_scale_one(c->stream, c->acc + 2 * c->P + 4, 1.f / (float)(world > 1 ? world : 1)));
  return VMC_OK;
}


}  // namespace vmcapi

extern "C" {

int vmc_accumulators_devptr(vmc_ctx* c, void** dev_ptr, int64_t* n_floats) {
  CHECK_CTX(c);
  PROPAGATE(acc_zeros(c));
  if (dev_ptr) *dev_ptr = c->acc;
  if (n_floats) *n_floats = 2 * c->P + 8;
  return VMC_OK;
}

// In-place SUM all-reduce of the accumulator buffer, stream-ordered on the ctx's stream (transport:
// see reduce_buffer).  g_count (number of accumulate calls, identical on every rank) is divided
// back by the world size so that sharded and unsharded gradients agree (cgs_vmc_amd/parallel.py).
int vmc_allreduce_accumulators(vmc_ctx* c, void* nccl_comm, int32_t world_size) {
  CHECK_CTX(c);
  PROPAGATE(acc_zeros(c));
  PROPAGATE(reduce_accumulators(c, nccl_comm, world_size));
  return VMC_OK;
}

int vmc_set_host_allreduce(vmc_ctx* c, vmc_host_allreduce_fn hook, void* user) {
  CHECK_CTX(c);
  c->host_reduce = hook;
  c->host_reduce_user = user;
  c->host_reduce_caps = 0;          // a new hook has declared nothing yet
  return VMC_OK;
}

int vmc_set_host_allreduce_caps(vmc_ctx* c, int32_t caps) {
  CHECK_CTX(c);
  if (caps & ~VMC_HOST_REDUCE_CAP_F64) return fail(c, VMC_ERR_INVALID, "unknown capability bits");
  c->host_reduce_caps = caps;
  return VMC_OK;
}

int vmc_set_device_allreduce(vmc_ctx* c, vmc_device_allreduce_fn hook, void* user) {
  CHECK_CTX(c);
  c->dev_reduce = hook;
  c->dev_reduce_user = user;
  return VMC_OK;
}

const char* vmc_rccl_last_error(void) { return g_rccl_error.c_str(); }

const char* vmc_rccl_library_path(void) {
  static std::string path;
  const Rccl* r = rccl();
  Dl_info info;
  if (r && dladdr((void*)r->all_reduce, &info) && info.dli_fname) path = info.dli_fname;
  return path.c_str();
}

int vmc_device_pci_bus_id(int32_t device, char* buf, int32_t len) {
  if (!buf || len < 16) return VMC_ERR_INVALID;
  buf[0] = 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return VMC_ERR_HIP;
  if (device < 0 || device >= n) return VMC_ERR_INVALID;
  return hipDeviceGetPCIBusId(buf, len, device) == hipSuccess ? VMC_OK : VMC_ERR_HIP;
}

const char* vmc_hip_runtime_path(void) {
  static std::string path;
  Dl_info info;
  if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) path = info.dli_fname;
  return path.c_str();
}

int vmc_rccl_unique_id(uint8_t id[128]) {
  if (!id) { g_rccl_error = "null id"; return VMC_ERR_INVALID; }
  const Rccl* r = rccl();
  if (!r) return VMC_ERR_UNSUPPORTED;
  RcclUniqueId u;
  const int rc = r->get_unique_id(&u);
  if (rc != 0) { g_rccl_error = std::string("ncclGetUniqueId: ") + rccl_error_string(r, rc); return VMC_ERR_HIP; }
  memcpy(id, u.internal, sizeof(u.internal));
  return VMC_OK;
}

int vmc_rccl_comm_create(const uint8_t id[128], int32_t world_size, int32_t rank, int32_t device,
                         void** nccl_comm) {
  if (!id || !nccl_comm || world_size < 1 || rank < 0 || rank >= world_size) {
    g_rccl_error = "bad communicator arguments";
    return VMC_ERR_INVALID;
  }
  *nccl_comm = nullptr;
  const Rccl* r = rccl();
  if (!r) return VMC_ERR_UNSUPPORTED;
  DeviceGuard guard(device);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device) { g_rccl_error = "cannot select the device"; return VMC_ERR_HIP; }
  RcclUniqueId u;
  memcpy(u.internal, id, sizeof(u.internal));
  const int rc = r->comm_init_rank(nccl_comm, world_size, u, rank);
  if (rc != 0) { g_rccl_error = std::string("ncclCommInitRank: ") + rccl_error_string(r, rc); *nccl_comm = nullptr; return VMC_ERR_HIP; }
  return VMC_OK;
}

int vmc_rccl_comm_destroy(void* nccl_comm) {
  if (!nccl_comm) return VMC_OK;
  const Rccl* r = rccl();
  if (!r) return VMC_ERR_UNSUPPORTED;
  const int rc = r->comm_destroy(nccl_comm);
  if (rc != 0) { g_rccl_error = std::string("ncclCommDestroy: ") + rccl_error_string(r, rc); return VMC_ERR_HIP; }
  return VMC_OK;
}

int vmc_debug_allreduce(vmc_ctx* c, void* nccl_comm, int32_t world_size, float* host, int64_t n, int32_t op) {
  ENTER(c);
  if (!host || n < 1 || (op != VMC_REDUCE_SUM && op != VMC_REDUCE_MAX)) return fail(c, VMC_ERR_INVALID, "bad arguments");
  if (n > c->d_stage_n) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_stage) hipFree(c->d_stage);
    c->d_stage = nullptr; c->d_stage_n = 0;
    HIPCHK(c, dalloc(&c->d_stage, n));
    c->d_stage_n = n;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_stage, host, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  PROPAGATE(reduce_buffer(c, nccl_comm, world_size, c->d_stage, n, op));
  HIPCHK(c, hipMemcpyAsync(host, c->d_stage, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return VMC_OK;
}


}  // extern "C"
