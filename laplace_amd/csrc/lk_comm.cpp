// The fit's ONE collective behind the C ABI (include/laplace_hip.h, SURVEY.md 8b: `lk_allreduce_sum_f32(comm, buf, count,
// stream)` wrapping ncclAllReduce): the epoch-end sum of the packed factors over the ranks of a node, for a host that is not
// PyTorch (the Python host of this repository uses torch.distributed's "nccl" backend — the same RCCL — and never calls
// these).  RCCL is bound at the FIRST call (dlopen), not at link time: the library loads on a box without RCCL, and inside a
// PyTorch process no second copy of RCCL is pulled in beside the one torch ships unless a caller asks for it.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include "laplace_hip.h"

namespace lk {
void set_error(const char* fmt, ...);
}

namespace {
// (the four entry points of rccl.h this file needs, by their C signatures: ncclResult_t is an int-sized enum with
//  ncclSuccess = 0; ncclUniqueId is 128 opaque bytes; ncclFloat = 7, ncclSum = 0 in RCCL 2.x)
struct UniqueId {
  char internal[128];
};
using GetUniqueId = int (*)(UniqueId*);
using CommInitRank = int (*)(void**, int, UniqueId, int);
using CommDestroy = int (*)(void*);
using AllReduce = int (*)(const void*, void*, size_t, int, int, void*, void*);
using GetErrorString = const char* (*)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueId get_unique_id = nullptr;
  CommInitRank comm_init_rank = nullptr;
  CommDestroy comm_destroy = nullptr;
  AllReduce all_reduce = nullptr;
  GetErrorString error_string = nullptr;
};

const Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    r.get_unique_id = (GetUniqueId)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRank)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (CommDestroy)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (AllReduce)dlsym(r.handle, "ncclAllReduce");
    r.error_string = (GetErrorString)dlsym(r.handle, "ncclGetErrorString");
  });
  if (!r.handle || !r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) {
    lk::set_error("RCCL (librccl.so.1) is not available on this host: %s", r.handle ? "missing symbols" : dlerror());
    return nullptr;
  }
  return &r;
}

int fail(const Rccl* r, const char* what, int rc) {
  lk::set_error("%s: %s (ncclResult %d)", what, r->error_string ? r->error_string(rc) : "RCCL error", rc);
  return LK_ELAUNCH;
}
}  // namespace

extern "C" int lk_comm_unique_id(void* id128) {
  if (!id128) {
    lk::set_error("lk_comm_unique_id: null pointer");
    return LK_EINVAL;
  }
  const Rccl* r = rccl();
  if (!r) return LK_ELAUNCH;
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return LK_OK;
}

extern "C" int lk_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) {
    lk::set_error("lk_comm_init_rank: bad arguments");
    return LK_EINVAL;
  }
  const Rccl* r = rccl();
  if (!r) return LK_ELAUNCH;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  const int rc = r->comm_init_rank(comm, nranks, id, rank);
  return rc ? fail(r, "ncclCommInitRank", rc) : LK_OK;
}

extern "C" int lk_comm_destroy(void* comm) {
  if (!comm) return LK_OK;
  const Rccl* r = rccl();
  if (!r) return LK_ELAUNCH;
  const int rc = r->comm_destroy(comm);
  return rc ? fail(r, "ncclCommDestroy", rc) : LK_OK;
}

extern "C" int lk_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream) {
  if (!comm || (!buf && count > 0) || count < 0) {
    lk::set_error("lk_allreduce_sum_f32: bad arguments");
    return LK_EINVAL;
  }
  if (count == 0) return LK_OK;
  const Rccl* r = rccl();
  if (!r) return LK_ELAUNCH;
  const int rc = r->all_reduce(buf, buf, (size_t)count, /* ncclFloat */ 7, /* ncclSum */ 0, comm, stream);
  return rc ? fail(r, "ncclAllReduce", rc) : LK_OK;
}
