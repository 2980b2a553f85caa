// Gradient all-reduce of data-parallel training on RCCL over xGMI, as C entry points: the exchange
// step tf.distribute.MultiWorkerMirroredStrategy performs for the reference at
// /root/reference model_train.py:272-301 (sum of the replicas' gradients, one replica per GPU).
// RCCL is bound at run time (dlopen) on the first miseg_comm_* call, so that loading libmiseg.so
// never depends on it and a process that has already loaded an RCCL (PyTorch's) shares that copy.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace miseg {

namespace {
typedef struct { char internal[128]; } rccl_unique_id;     // ncclUniqueId
typedef void* rccl_comm;                                    // ncclComm_t
typedef int (*get_unique_id_fn)(rccl_unique_id*);
typedef int (*comm_init_rank_fn)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*comm_destroy_fn)(rccl_comm);
typedef const char* (*get_error_string_fn)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;               // ncclDataType_t / ncclRedOp_t values

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  all_reduce_fn all_reduce = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  get_error_string_fn get_error_string = nullptr;
};
Rccl g_rccl;

int bind_rccl() {
  if (g_rccl.handle) return MISEG_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    set_error("comm: cannot load librccl.so (%s)", dlerror());
    return MISEG_E_HIP;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = (get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
  r.all_reduce = (all_reduce_fn)dlsym(h, "ncclAllReduce");
  r.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
  r.get_error_string = (get_error_string_fn)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy || !r.get_error_string) {
    set_error("comm: librccl.so lacks an expected symbol");
    dlclose(h);
    return MISEG_E_HIP;
  }
  g_rccl = r;
  return MISEG_OK;
}

int check_rccl(int rc, const char* what) {
  if (rc == 0) return MISEG_OK;
  set_error("comm: %s failed: %s", what, g_rccl.get_error_string(rc));
  return MISEG_E_HIP;
}
}  // namespace

}  // namespace miseg

struct miseg_comm {
  miseg::rccl_comm comm;
  int rank, world;
};

using namespace miseg;

extern "C" {

int miseg_comm_unique_id(void* id) {
  MISEG_CHECK_ARG(id, "comm: null id buffer");
  int rc = bind_rccl();
  if (rc != MISEG_OK) return rc;
  return check_rccl(g_rccl.get_unique_id((rccl_unique_id*)id), "ncclGetUniqueId");
}

int miseg_comm_init(miseg_comm** out, int32_t rank, int32_t world, const void* id) {
  MISEG_CHECK_ARG(out && id, "comm: null argument");
  MISEG_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "comm: rank %d outside world %d", rank, world);
  int rc = bind_rccl();
  if (rc != MISEG_OK) return rc;
  rccl_unique_id uid;
  memcpy(&uid, id, sizeof uid);
  rccl_comm c = nullptr;
  rc = check_rccl(g_rccl.comm_init_rank(&c, world, uid, rank), "ncclCommInitRank");
  if (rc != MISEG_OK) return rc;
  *out = new miseg_comm{c, rank, world};
  return MISEG_OK;
}

int miseg_allreduce_async(miseg_comm* comm, float* buf, int64_t count, void* stream) {
  MISEG_CHECK_ARG(comm && buf && count >= 0, "comm: bad all-reduce argument");
  if (count == 0) return MISEG_OK;
  return check_rccl(g_rccl.all_reduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, comm->comm, (hipStream_t)stream),
                    "ncclAllReduce");
}

int miseg_comm_destroy(miseg_comm* comm) {
  if (!comm) return MISEG_OK;
  const int rc = check_rccl(g_rccl.comm_destroy(comm->comm), "ncclCommDestroy");
  delete comm;
  return rc;
}

}  // extern "C"
