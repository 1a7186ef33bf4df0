// pd_comm_*: the data-parallel gradient exchange behind the C ABI (SURVEY.md 8(b) "later" set: pd_comm_init(unique_id, rank, world),
// pd_allreduce_bucket) -- what DistributedDataParallel's bucketed all-reduce does under `accelerator.backward(loss)`
// (train.py:311-326, utils_training.py:436).  RCCL is loaded at run time (dlopen: the library keeps loading on a box without it, and a
// process that never trains in data parallel never touches it); one communicator per process = per GPU, created on the CURRENT device.
//   pd_comm_unique_id   rank 0 draws the 128-byte id; the host program ships it to the other ranks (torch.distributed's store, a file, MPI)
//   pd_comm_init        ncclCommInitRank
//   pd_allreduce_bucket in-place fp32 sum (or mean: ncclAvg inside the collective when the library has it -- no extra pass over the bucket, 7 GB of
//                       traffic per step on the 3.46 GB SD gradient -- else sum + a scaling kernel) of one contiguous gradient bucket on the given stream; algo 0 = ncclAllReduce (RCCL picks
//                       ring / tree), algo 1 = reduce-scatter + all-gather over equal shards (the direct form SURVEY 5.8 asks for on the
//                       7 point-to-point xGMI links: each rank reduces 1 / world of the bucket and broadcasts it; needs count % world == 0)
//   pd_comm_query       rank / size of the communicator as RCCL reports them
//   pd_comm_destroy
// The Python trainers use torch.distributed (PyTorch is plumbing here); `phendiff_amd.comm.NativeComm` is the same exchange through this ABI.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "pd_common.h"

namespace pd {

typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(nccl_uid*) = nullptr;
  int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*CommCount)(const nccl_comm, int*) = nullptr;
  int (*CommUserRank)(const nccl_comm, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  bool has_avg = false;          // ncclAvg (NCCL >= 2.10): the mean comes out of the collective itself, no pass over the bucket afterwards
};
// Loaded once (thread-safe: function-local static initialiser); when the library or one of its symbols is missing the reason is kept
// for the error message and the handle is closed again.
struct RcclLoad {
  Rccl r;
  char why[256];
  RcclLoad() {
    why[0] = 0;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
      const char* e = dlerror();                       // (one call: dlerror() clears the message it returns)
      snprintf(why, sizeof(why), "%s", e ? e : "dlopen failed");
    }
    if (!r.h) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(r.h, "ncclReduceScatter");
    r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.h, "ncclCommUserRank");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.h, "ncclGetVersion");
    int ver = 0;
    r.has_avg = r.GetVersion != nullptr && r.GetVersion(&ver) == 0 && ver >= 21000;
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.ReduceScatter || !r.AllGather || !r.CommDestroy || !r.CommCount || !r.CommUserRank) {
      snprintf(why, sizeof(why), "librccl.so lacks one of the nccl* entry points this library binds");
      dlclose(r.h);
      r.h = nullptr;
    }
  }
};
static RcclLoad& rccl_load() { static RcclLoad l; return l; }
static Rccl* rccl() { RcclLoad& l = rccl_load(); return l.r.h ? &l.r : nullptr; }
static const char* rccl_why() { return rccl_load().why; }
struct Comm { nccl_comm c; int rank, world; };
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0, NCCL_AVG = 4;

__global__ __launch_bounds__(256) void scale_kernel(float* p, size_t n, float s) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] *= s;
}

#define PD_RCCL(call, what)                                                                                   \
  do {                                                                                                        \
    int rc_ = (call);                                                                                         \
    if (rc_ != 0) {                                                                                           \
      set_error("%s: RCCL error %d (%s)", what, rc_, R->GetErrorString ? R->GetErrorString(rc_) : "?");       \
      return PD_ERR_LAUNCH;                                                                                   \
    }                                                                                                         \
  } while (0)

}  // namespace pd

using namespace pd;

extern "C" int pd_comm_unique_id(pd_comm_id* out) {
  PD_CHECK(out != nullptr, PD_ERR_ARG, "pd_comm_unique_id: null output");
  Rccl* R = rccl();
  PD_CHECK(R != nullptr, PD_ERR_UNSUPPORTED, "pd_comm: librccl.so could not be loaded (%s)", rccl_why());
  static_assert(sizeof(pd_comm_id) == sizeof(nccl_uid), "id size");
  PD_RCCL(R->GetUniqueId((nccl_uid*)out), "pd_comm_unique_id");
  return PD_OK;
}

extern "C" int pd_comm_init(const pd_comm_id* id, int rank, int world, void** comm_out) {
  PD_CHECK(id != nullptr && comm_out != nullptr, PD_ERR_ARG, "pd_comm_init: null argument");
  PD_CHECK(world >= 1 && rank >= 0 && rank < world, PD_ERR_ARG, "pd_comm_init: rank %d of %d", rank, world);
  Rccl* R = rccl();
  PD_CHECK(R != nullptr, PD_ERR_UNSUPPORTED, "pd_comm: librccl.so could not be loaded (%s)", rccl_why());
  nccl_uid u;
  memcpy(&u, id, sizeof(u));
  nccl_comm c = nullptr;
  PD_RCCL(R->CommInitRank(&c, world, u, rank), "pd_comm_init");
  Comm* cm = new Comm{c, rank, world};
  *comm_out = cm;
  return PD_OK;
}

extern "C" int pd_allreduce_bucket(void* comm, float* buf, size_t count, int mean, int algo, void* stream) {
  PD_CHECK(comm != nullptr, PD_ERR_ARG, "pd_allreduce_bucket: null communicator (pd_comm_init first)");
  PD_CHECK(buf != nullptr && count > 0, PD_ERR_ARG, "pd_allreduce_bucket: empty bucket");
  PD_CHECK(algo == 0 || algo == 1, PD_ERR_ARG, "pd_allreduce_bucket: algo %d (0 = all-reduce, 1 = reduce-scatter + all-gather)", algo);
  Rccl* R = rccl();
  PD_CHECK(R != nullptr, PD_ERR_UNSUPPORTED, "pd_comm: librccl.so could not be loaded (%s)", rccl_why());
  Comm* cm = (Comm*)comm;
  hipStream_t st = (hipStream_t)stream;
  // (no `world > 1` guard: RCCL runs both collectives at one rank too, so a one-rank communicator exercises this branch and its in-place
  // pointer arithmetic -- ADVICE r3)
  // ncclAvg is a pre-scaled sum (x / W summed).  For a power-of-two world the scaling is exact, so the result is bit-identical to
  // sum-then-scale -- what the torch exchange of the trainers (all_reduce SUM, then div_) computes; for any other world size the two
  // round differently, so there the scaling stays a separate pass and the native exchange keeps matching the torch one bit for bit
  // (ADVICE r5).
  const bool avg_in_collective = mean && cm->world > 1 && R->has_avg && (cm->world & (cm->world - 1)) == 0;
  const int op = avg_in_collective ? NCCL_AVG : NCCL_SUM;
  if (algo == 1 && count % (size_t)cm->world == 0) {
    const size_t shard = count / cm->world;
    PD_RCCL(R->ReduceScatter(buf, buf + (size_t)cm->rank * shard, shard, NCCL_FLOAT32, op, cm->c, st), "pd_allreduce_bucket (reduce-scatter)");
    PD_RCCL(R->AllGather(buf + (size_t)cm->rank * shard, buf, shard, NCCL_FLOAT32, cm->c, st), "pd_allreduce_bucket (all-gather)");
  } else {
    PD_RCCL(R->AllReduce(buf, buf, count, NCCL_FLOAT32, op, cm->c, st), "pd_allreduce_bucket");
  }
  if (mean && cm->world > 1 && !avg_in_collective) {
    const unsigned blocks = (unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, st, buf, count, 1.0f / (float)cm->world);
    PD_LAUNCH_CHECK();
  }
  return PD_OK;
}

// rank / size as the COMMUNICATOR reports them (ncclCommUserRank / ncclCommCount), not as the caller passed them or the environment says
extern "C" int pd_comm_query(void* comm, int* rank_out, int* world_out) {
  PD_CHECK(comm != nullptr && rank_out != nullptr && world_out != nullptr, PD_ERR_ARG, "pd_comm_query: null argument");
  Rccl* R = rccl();
  PD_CHECK(R != nullptr, PD_ERR_UNSUPPORTED, "pd_comm: librccl.so could not be loaded (%s)", rccl_why());
  Comm* cm = (Comm*)comm;
  PD_RCCL(R->CommUserRank(cm->c, rank_out), "pd_comm_query (rank)");
  PD_RCCL(R->CommCount(cm->c, world_out), "pd_comm_query (count)");
  return PD_OK;
}

extern "C" int pd_comm_destroy(void* comm) {
  if (comm == nullptr) return PD_OK;
  Rccl* R = rccl();
  Comm* cm = (Comm*)comm;
  int rc = 0;
  if (R != nullptr) rc = R->CommDestroy(cm->c);
  delete cm;
  if (rc != 0) { set_error("pd_comm_destroy: RCCL error %d", rc); return PD_ERR_LAUNCH; }
  return PD_OK;
}
