// tf_comm.cpp -- multi-GPU boundary exchange inside the C ABI: one process per GPU,
// RCCL over xGMI.  The path has exactly one exchange step: the ghost-band chunks a rank updated since the previous
// exchange travel to the ranks that read them.  Two forms, both fixed-capacity [count | records] blocks with the record
// count in-band, so nothing comes back to the host and the exchange is just another operation on the handle's stream
// between the voxel update and the mesher:
//   neighbours (default): slabs are contiguous key ranges, so a rank's ghost band is read by the rank below (keys
//     lo .. lo + a + b + c) and the rank above (key hi - 1) only -- one grouped ncclSend / ncclRecv pair per neighbour:
//     a rank receives TWO blocks whatever the number of ranks, and each block holds only that neighbour's band;
//   all-gather: ONE ncclAllGather of every rank's block to every rank (7 received blocks at 8 ranks); needed when a slab
//     is thinner than a + b + c + 1 keys, so that a band reaches beyond the adjacent slab.
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
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
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
  g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(dlsym(g_rccl.lib, "ncclSend"));
  g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(dlsym(g_rccl.lib, "ncclRecv"));
  g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(dlsym(g_rccl.lib, "ncclGroupStart"));
  g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(dlsym(g_rccl.lib, "ncclGroupEnd"));
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
  TF_HIP(hipMalloc(&c.d_send, block * 2));  // all-gather: one block; neighbours: the down block, then the up block
  TF_HIP(hipMalloc(&c.d_recv, block * (size_t)(c.nranks > 2 ? c.nranks : 2)));
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
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(c.comm);
  uint8_t* send = reinterpret_cast<uint8_t*>(c.d_send);
  uint8_t* recv = reinterpret_cast<uint8_t*>(c.d_recv);
  int nblocks = c.nranks, skip = c.rank;
  const bool neighbours = c.mode == TF_XCHG_NEIGHBOURS && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
  if (neighbours) {
    // down block -> rank - 1, up block -> rank + 1; from rank - 1 comes ITS up block, from rank + 1 its down block.
    // A missing neighbour's receive slot keeps a zero count.
    rc = tf_boundary_pack_bands(v, send, send + block, c.cap_records);
    if (rc) return rc;
    const bool lower = c.rank > 0, upper = c.rank + 1 < c.nranks;
    if (!lower) TF_HIP(hipMemsetAsync(recv, 0, 16, v->stream));
    if (!upper) TF_HIP(hipMemsetAsync(recv + block, 0, 16, v->stream));
    if (lower || upper) {
      TF_NCCL(g_rccl.GroupStart());
      if (lower) {
        TF_NCCL(g_rccl.Send(send, block, ncclUint8, c.rank - 1, comm, v->stream));
        TF_NCCL(g_rccl.Recv(recv, block, ncclUint8, c.rank - 1, comm, v->stream));
      }
      if (upper) {
        TF_NCCL(g_rccl.Send(send + block, block, ncclUint8, c.rank + 1, comm, v->stream));
        TF_NCCL(g_rccl.Recv(recv + block, block, ncclUint8, c.rank + 1, comm, v->stream));
      }
      TF_NCCL(g_rccl.GroupEnd());
    }
    nblocks = 2;
    skip = -1;
    c.bytes_received += (uint64_t)((lower ? 1 : 0) + (upper ? 1 : 0)) * block;
  } else {
    rc = tf_boundary_pack_block(v, send, c.cap_records);
    if (rc) return rc;
    TF_NCCL(g_rccl.AllGather(send, recv, block, ncclUint8, comm, v->stream));
    c.bytes_received += (uint64_t)(c.nranks > 1 ? c.nranks - 1 : 0) * block;
  }
  c.exchanges += 1;
  VolumeDev d = v->dev;
  if (dirty_par >= 0) {
    d.work_ids = v->atlas.d_work_ids + (size_t)dirty_par * d.max_chunks;
    d.work_slot = v->atlas.d_work_slot + (size_t)dirty_par * d.max_chunks;
  }
  launch_boundary_unpack_blocks(d, recv, nblocks, skip, (uint32_t)c.cap_records, dirty_par, stamp, v->stream);
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

int tf_comm_exchange_mode(tf_volume* v, int mode) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  if (mode != TF_XCHG_NEIGHBOURS && mode != TF_XCHG_ALLGATHER) { set_error("unknown exchange mode"); return TF_ERR_INVALID; }
  v->comm.mode = mode;
  return TF_OK;
}

int tf_comm_stats(tf_volume* v, int64_t* exchanges, int64_t* bytes_received) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  if (exchanges) *exchanges = (int64_t)v->comm.exchanges;
  if (bytes_received) *bytes_received = (int64_t)v->comm.bytes_received;
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
