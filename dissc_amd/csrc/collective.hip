// The path's one collective behind the C ABI (include/dissc_hip.h: dissc_comm_*, dissc_allgather_waves): a thin binding of the
// RCCL the process holds.  No link dependency: the four functions are looked up with dlsym at first use -- in the RCCL that is
// already loaded (PyTorch's librccl.so when torch was imported first: RTLD_NOLOAD finds it by soname), else a copy is loaded.
// Everything here is host code; the payload layout is dissc_pack_rows' (pipeline_glue.hip).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace dissc {
namespace {

struct RcclId { char internal[DISSC_COMM_ID_BYTES]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE
typedef int (*GetUniqueIdFn)(RcclId*);
typedef int (*CommInitRankFn)(void**, int, RcclId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*ErrStrFn)(int);
constexpr int kNcclFloat32 = 7;  // ncclDataType_t ncclFloat32 (rccl.h)

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init_rank = nullptr;
  CommDestroyFn destroy = nullptr;
  AllGatherFn all_gather = nullptr;
  ErrStrFn err = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("DISSC_RCCL_LIB");
    const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    // the copy the process already holds first (same communicators, same HIP runtime as the caller's other RCCL users)
    for (const char* n : names)
      if (n && *n && !r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (n && *n && !r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) {
      const char* e = dlerror();
      r.why = std::string("no RCCL library could be loaded (librccl.so / librccl.so.1 / $DISSC_RCCL_LIB): ") + (e ? e : "?");
      return;
    }
    r.get_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
    r.init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
    r.destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
    r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
    r.err = (ErrStrFn)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_id || !r.init_rank || !r.destroy || !r.all_gather) {
      r.why = "the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
      r.get_id = nullptr;
    }
  });
  return &r;
}

int need(Rccl*& r, const char* who) {
  r = rccl();
  if (!r->get_id) {
    set_error("%s: %s", who, r->why.c_str());
    return DISSC_ENOTSUP;
  }
  return DISSC_OK;
}

int fail(Rccl* r, const char* who, const char* call, int code) {
  set_error("%s: %s failed: %s (%d)", who, call, r->err ? r->err(code) : "?", code);
  return DISSC_ECOMM;
}

}  // namespace
}  // namespace dissc

using namespace dissc;

extern "C" int dissc_comm_unique_id(void* id_out) {
  Rccl* r;
  if (!id_out) {
    set_error("dissc_comm_unique_id: null id");
    return DISSC_EINVAL;
  }
  if (int rc = need(r, "dissc_comm_unique_id")) return rc;
  RcclId id;
  memset(&id, 0, sizeof id);
  if (int c = r->get_id(&id)) return fail(r, "dissc_comm_unique_id", "ncclGetUniqueId", c);
  memcpy(id_out, &id, sizeof id);
  return DISSC_OK;
}

extern "C" int dissc_comm_create(const void* id, int nranks, int rank, void** comm_out) {
  Rccl* r;
  if (!id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) {
    set_error("dissc_comm_create: bad argument (nranks %d, rank %d)", nranks, rank);
    return DISSC_EINVAL;
  }
  *comm_out = nullptr;
  if (int rc = need(r, "dissc_comm_create")) return rc;
  RcclId uid;
  memcpy(&uid, id, sizeof uid);
  void* comm = nullptr;
  if (int c = r->init_rank(&comm, nranks, uid, rank)) return fail(r, "dissc_comm_create", "ncclCommInitRank", c);
  *comm_out = comm;
  return DISSC_OK;
}

extern "C" int dissc_comm_destroy(void* comm) {
  Rccl* r;
  if (!comm) return DISSC_OK;
  if (int rc = need(r, "dissc_comm_destroy")) return rc;
  if (int c = r->destroy(comm)) return fail(r, "dissc_comm_destroy", "ncclCommDestroy", c);
  return DISSC_OK;
}

extern "C" int dissc_allgather_waves(void* nccl_comm, const float* send, size_t n_floats, float* recv, void* stream) {
  Rccl* r;
  if (!nccl_comm || (n_floats && (!send || !recv))) {
    set_error("dissc_allgather_waves: bad argument");
    return DISSC_EINVAL;
  }
  if (int rc = need(r, "dissc_allgather_waves")) return rc;
  if (n_floats == 0) return DISSC_OK;
  if (int c = r->all_gather(send, recv, n_floats, kNcclFloat32, nccl_comm, (hipStream_t)stream))
    return fail(r, "dissc_allgather_waves", "ncclAllGather", c);
  return DISSC_OK;
}
