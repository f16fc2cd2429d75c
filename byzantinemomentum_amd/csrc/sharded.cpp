// sharded.cpp — dim-sharded aggregation behind the C ABI: one call per aggregation, the path's one
// exchange (an all-reduce of the n x n fp64 squared-distance partials) issued on the caller's stream
// through RCCL, everything else local to the shard.
//
// the reference has no multi-GPU aggregation; the MI355X scaling axis is the
// coordinate dimension.  Every rank holds all n rows restricted to its slice of d_local coordinates:
//   distances (local partial) -> ncclAllReduce(sum, n*n doubles, <= 32 KB: latency-bound on xGMI, never
//   a d-sized collective) -> score / stable rank (identical bits on every rank) -> local average or
//   Bulyan pass 2 of the slice.
// One entry point replaces four Python-side calls plus a torch.distributed collective: at 8 ranks the
// per-rank kernels take 20-40 us each, comparable to the host cost of ONE ctypes call.
//
// RCCL is bound lazily (dlopen of the librccl.so.1 the process already maps — torch's copy — then
// dlsym): libbm_gar.so has no link-time dependency on it, loads on machines without RCCL, and
// shares the RCCL instance torch.distributed uses.  With comm == NULL no collective is issued and
// the same entry points are the single-GPU single-call forms.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>
#include "bm_common.h"

namespace bm {
int64_t pairwise_workspace_bytes(int n, int64_t d);
}

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy the process already uses (torch's)
      if (r.handle != nullptr) break;
    }
    if (r.handle == nullptr)
      for (const char* name : {"librccl.so.1", "librccl.so"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle != nullptr) break;
      }
    if (r.handle == nullptr) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather;
  });
  return r;
}

}  // namespace

struct bm_comm {
  ncclComm_t comm;
  int nranks;
  int rank;
};

extern "C" int bm_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int bm_comm_unique_id(void* id128) {
  if (id128 == nullptr) return BM_EINVAL;
  if (!rccl().ok) return BM_ENOCOMM;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return BM_ECOMM;
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int bm_comm_init(bm_comm** out, int nranks, int rank, const void* id128) {
  if (out == nullptr || id128 == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return BM_EINVAL;
  if (!rccl().ok) return BM_ENOCOMM;
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  if (rccl().CommInitRank(&c, nranks, id, rank) != ncclSuccess) return BM_ECOMM;
  *out = new bm_comm{c, nranks, rank};
  return 0;
}

extern "C" int bm_comm_destroy(bm_comm* comm) {
  if (comm == nullptr) return 0;
  const ncclResult_t r = rccl().ok ? rccl().CommDestroy(comm->comm) : ncclSuccess;
  delete comm;
  return r == ncclSuccess ? 0 : BM_ECOMM;
}

extern "C" int bm_comm_size(const bm_comm* comm) { return comm == nullptr ? 1 : comm->nranks; }

extern "C" int bm_allreduce_sum_f64(bm_comm* comm, double* buf, int64_t count, void* stream) {
  if (buf == nullptr || count < 0) return BM_EINVAL;
  if (comm == nullptr || count == 0) return 0;  // a one-rank communicator still goes through RCCL
  return rccl().AllReduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, comm->comm,
                          static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : BM_ECOMM;
}

extern "C" int bm_allgather_f32(bm_comm* comm, const float* mine, float* all, int64_t count_per_rank, void* stream) {
  if (mine == nullptr || all == nullptr || count_per_rank < 0) return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (comm == nullptr) {
    if (mine != all && count_per_rank > 0)
      return bm::hip_code(hipMemcpyAsync(all, mine, (size_t)count_per_rank * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
  }
  return rccl().AllGather(mine, all, (size_t)count_per_rank, ncclFloat32, comm->comm, s) == ncclSuccess ? 0 : BM_ECOMM;
}

// ws layout of the sharded rules: [sq n*n doubles][order 64 int32][pad][pairwise workspace]
namespace {
constexpr int64_t kShardHeader = BM_MAX_ROWS * BM_MAX_ROWS * 8 + BM_MAX_ROWS * 4 + 256;

// have_sq: the squared distances of the local shard are already at the head of ws (bm_momentum_stats_sqdist wrote them)
int sharded_rank(bm_comm* comm, const float* const* rows, int n, int64_t d_local, int64_t d_total, int f, int m, int mode,
                 void* ws, void* stream, double** sq_out, int32_t** order_out, bool have_sq = false) {
  char* base = static_cast<char*>(ws);
  double* sq = reinterpret_cast<double*>(base);
  int32_t* order = reinterpret_cast<int32_t*>(base + BM_MAX_ROWS * BM_MAX_ROWS * 8);
  void* pair_ws = base + kShardHeader;
  int rc = 0;
  if (!have_sq && comm == nullptr) {
    // one rank: nothing to exchange — distances, gate and ranking in the distance pass's own launches
    *sq_out = sq;
    *order_out = order;
    return bm_pairwise_rank(rows, n, d_local, d_total, f, m, mode, sq, order, nullptr, pair_ws, stream);
  }
  if (!have_sq) {
    // the precision plan of the distance pass follows the length of the WHOLE vector (all shards), which the caller
    // states: a short or empty trailing shard must plan exactly like its peers
    rc = bm_pairwise_sqdist_shard(rows, n, d_local, d_total, sq, pair_ws, stream);
    if (rc != 0) return rc;
  }
  rc = bm_allreduce_sum_f64(comm, sq, (int64_t)n * n, stream);
  if (rc != 0) return rc;
  rc = bm_krum_rank(sq, n, f, m, mode, order, nullptr, stream);
  *sq_out = sq;
  *order_out = order;
  return rc;
}
}  // namespace

extern "C" int64_t bm_sharded_workspace_bytes(int n, int64_t d_local) {
  if (n < 1 || n > BM_MAX_ROWS || d_local < 0) return BM_EINVAL;
  return kShardHeader + bm::pairwise_workspace_bytes(n, d_local);
}

extern "C" int bm_sharded_krum(bm_comm* comm, const float* const* rows, int n, int64_t d_local, int64_t d_total, int f, int m,
                               float* out_local, int32_t* order_out, void* ws, void* stream) {
  // an empty shard (d_local == 0) has no output buffer: torch.empty(0).data_ptr() is NULL
  if (rows == nullptr || (out_local == nullptr && d_local > 0) || ws == nullptr || n < 1 || n > BM_MAX_ROWS ||
      d_local < 0 || d_total < d_local || f < 0 || m < 1 || m > n)
    return BM_EINVAL;
  double* sq;
  int32_t* order;
  int rc = sharded_rank(comm, rows, n, d_local, d_total, f, m, BM_RANK_KRUM, ws, stream, &sq, &order);
  if (rc != 0) return rc;
  if (order_out != nullptr) {
    rc = bm::hip_code(hipMemcpyAsync(order_out, order, BM_MAX_ROWS * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                     static_cast<hipStream_t>(stream)));
    if (rc != 0) return rc;
  }
  return bm_selected_mean(rows, n, order, m, d_local, out_local, stream);
}

extern "C" int bm_sharded_bulyan(bm_comm* comm, const float* const* rows, int n, int64_t d_local, int64_t d_total, int f, int m,
                                 float* out_local, int32_t* order_out, void* ws, void* stream) {
  // an empty shard (d_local == 0) has no output buffer: torch.empty(0).data_ptr() is NULL
  if (rows == nullptr || (out_local == nullptr && d_local > 0) || ws == nullptr || n < 1 || n > BM_MAX_ROWS ||
      d_local < 0 || d_total < d_local || f < 0 || m < 1 || m > n)
    return BM_EINVAL;
  double* sq;
  int32_t* order;
  int rc = sharded_rank(comm, rows, n, d_local, d_total, f, m, BM_RANK_BULYAN, ws, stream, &sq, &order);
  if (rc != 0) return rc;
  if (order_out != nullptr) {
    rc = bm::hip_code(hipMemcpyAsync(order_out, order, BM_MAX_ROWS * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                     static_cast<hipStream_t>(stream)));
    if (rc != 0) return rc;
  }
  if (d_local == 0) return 0;
  return bm_bulyan_pass2(rows, n, order, f, m, d_local, out_local, stream);
}

// The same two rules when the squared distances of the local shard are already in the workspace
// (bm_sharded_sq_slot(ws), written by bm_momentum_stats_sqdist with ws_pair = bm_sharded_pair_workspace(ws)).
extern "C" double* bm_sharded_sq_slot(void* ws) { return static_cast<double*>(ws); }
extern "C" void* bm_sharded_pair_workspace(void* ws) { return static_cast<char*>(ws) + kShardHeader; }

extern "C" int bm_sharded_rule_from_sq(bm_comm* comm, int rule, const float* const* rows, int n, int64_t d_local, int f,
                                       int m, float* out_local, int32_t* order_out, void* ws, void* stream) {
  if (rows == nullptr || (out_local == nullptr && d_local > 0) || ws == nullptr || n < 1 || n > BM_MAX_ROWS ||
      d_local < 0 || f < 0 || m < 1 || m > n || (rule != BM_RULE_KRUM && rule != BM_RULE_BULYAN))
    return BM_EINVAL;
  double* sq;
  int32_t* order;
  int rc = sharded_rank(comm, rows, n, d_local, d_local, f, m, rule == BM_RULE_KRUM ? BM_RANK_KRUM : BM_RANK_BULYAN, ws, stream,
                        &sq, &order, true);
  if (rc != 0) return rc;
  if (order_out != nullptr) {
    rc = bm::hip_code(hipMemcpyAsync(order_out, order, BM_MAX_ROWS * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                     static_cast<hipStream_t>(stream)));
    if (rc != 0) return rc;
  }
  if (rule == BM_RULE_KRUM) return bm_selected_mean(rows, n, order, m, d_local, out_local, stream);
  if (d_local == 0) return 0;
  return bm_bulyan_pass2(rows, n, order, f, m, d_local, out_local, stream);
}
