// tf_comm.cpp -- multi-GPU boundary exchange inside the C ABI: one process per GPU,
// RCCL over xGMI.  The path has exactly one collective: an all-gather of the ghost-band chunks a rank updated
// since the previous exchange.  It is ONE fixed-capacity ncclAllGather of [count | records] blocks -- the record
// count travels in-band, so nothing comes back to the host and the exchange is just another operation on the
// handle's stream between the voxel update and the mesher.
//
// librccl is opened at run time (dlopen) when tf_comm_init is first called: single-GPU users never load it,
// and a host process that already carries an RCCL (e.g. PyTorch's) keeps exactly one copy.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "tf_volume.h"

namespace tf {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;

static int rccl_load() {
  if (g_rccl.lib) return TF_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) { set_error(std::string("cannot load librccl: ") + dlerror()); return TF_ERR_HIP; }
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.lib, "ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.lib, "ncclCommDestroy"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.lib, "ncclAllGather"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.lib, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) {
    set_error("librccl lacks an expected symbol");
    return TF_ERR_HIP;
  }
  return TF_OK;
}

#define TF_NCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) {                                                                            \
      set_error(std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?")); \
      return TF_ERR_HIP;                                                                                \
    }                                                                                                   \
  } while (0)

static int comm_buffers(tf_volume* v, int64_t cap_records) {
  CommState& c = v->comm;
  const size_t block = tf_boundary_block_bytes(cap_records);
  if (c.cap_records >= cap_records && c.d_send) return TF_OK;
  TF_HIP(hipStreamSynchronize(v->stream));
  if (c.d_send) hipFree(c.d_send);
  if (c.d_recv) hipFree(c.d_recv);
  c.d_send = c.d_recv = nullptr;
  TF_HIP(hipMalloc(&c.d_send, block));
  TF_HIP(hipMalloc(&c.d_recv, block * (size_t)(c.nranks > 0 ? c.nranks : 1)));
  c.cap_records = cap_records;
  return TF_OK;
}

// pack -> all-gather -> unpack on the handle's stream; dirty_par >= 0: the owned face neighbours of every ghost
// chunk that arrives join the fused flow's per-frame dirty list (their neighbour was updated on another rank)
int comm_exchange(tf_volume* v, int64_t cap_records, int dirty_par, uint32_t stamp) {
  CommState& c = v->comm;
  if (!c.comm) { set_error("tf_comm_init has not been called"); return TF_ERR_INVALID; }
  int rc = comm_buffers(v, cap_records);
  if (rc) return rc;
  const size_t block = tf_boundary_block_bytes(c.cap_records);
  rc = tf_boundary_pack_block(v, c.d_send, c.cap_records);
  if (rc) return rc;
  TF_NCCL(g_rccl.AllGather(c.d_send, c.d_recv, block, ncclUint8, reinterpret_cast<ncclComm_t>(c.comm), v->stream));
  VolumeDev d = v->dev;
  if (dirty_par >= 0) {
    d.work_ids = v->atlas.d_work_ids + (size_t)dirty_par * d.max_chunks;
    d.work_slot = v->atlas.d_work_slot + (size_t)dirty_par * d.max_chunks;
  }
  launch_boundary_unpack_blocks(d, reinterpret_cast<const uint8_t*>(c.d_recv), c.nranks, c.rank, (uint32_t)c.cap_records,
                                dirty_par, stamp, v->stream);
  TF_HIP(hipGetLastError());
  v->host_list_n = -1;
  return TF_OK;
}

void comm_destroy(tf_volume* v) {
  CommState& c = v->comm;
  if (c.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(c.comm));
  if (c.d_send) hipFree(c.d_send);
  if (c.d_recv) hipFree(c.d_recv);
  c = CommState();
}

}  // namespace tf

using namespace tf;

extern "C" {

int tf_comm_unique_id(void* out128) {
  if (!out128) { set_error("null argument"); return TF_ERR_INVALID; }
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  TF_NCCL(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, 128);
  return TF_OK;
}

int tf_comm_init(tf_volume* v, int rank, int nranks, const void* unique_id128) {
  if (!v || !unique_id128) { set_error("null argument"); return TF_ERR_INVALID; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { set_error("bad rank / nranks"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int rc = rccl_load();
  if (rc) return rc;
  comm_destroy(v);
  ncclUniqueId id;
  memcpy(&id, unique_id128, 128);
  ncclComm_t comm = nullptr;
  TF_NCCL(g_rccl.CommInitRank(&comm, nranks, id, rank));
  v->comm.comm = comm;
  v->comm.rank = rank;
  v->comm.nranks = nranks;
  return TF_OK;
}

int tf_comm_destroy(tf_volume* v) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  TF_HIP(hipStreamSynchronize(v->stream));
  comm_destroy(v);
  return TF_OK;
}

int tf_comm_exchange_every_frame(tf_volume* v, int64_t cap_records) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  if (cap_records > 0 && !v->comm.comm) { set_error("tf_comm_init has not been called"); return TF_ERR_INVALID; }
  v->comm_cap = cap_records > 0 ? cap_records : 0;
  return TF_OK;
}

int tf_exchange_boundary(tf_volume* v, int64_t cap_records) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  if (cap_records <= 0) { set_error("cap_records must be positive"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return comm_exchange(v, cap_records, -1, 0);
}

}  // extern "C"
