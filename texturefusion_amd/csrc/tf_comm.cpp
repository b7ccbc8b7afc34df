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
// and a host process that already carries an RCCL (e.g. PyTorch's) keeps exactly one copy (RTLD_NOLOAD first).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <vector>

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
  // an RCCL the process already carries (PyTorch's, when the caller lives in a torch process) is reused: ONE library
  // instance, one set of its global state, whatever made the communicators -- RTLD_NOLOAD only succeeds for a library
  // that is mapped already
  for (const char* n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (g_rccl.lib) break;
  }
  for (const char* n : names) {
    if (g_rccl.lib) break;
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
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

// The neighbour form moves a band only to the adjacent rank, which is right when (1) rank r's slab sits directly above
// rank r - 1's (lo_r == hi_{r-1}, same key coefficients everywhere) and (2) every slab above the lowest is at least
// a + b + c + 1 keys wide, so that no band reaches past the adjacent slab.  One 32-byte all-gather of {lo, hi, a, b, c} decides it for all
// ranks alike; when it does not hold every rank uses the all-gather form (works for any partition).
static int comm_check_partition(tf_volume* v) {
  CommState& c = v->comm;
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(c.comm);
  const int n = c.nranks;
  int32_t mine[8] = {v->dev.part_lo, v->dev.part_hi, v->dev.part_a, v->dev.part_b, v->dev.part_c, 0, 0, 0};
  int32_t* d_buf = nullptr;
  TF_HIP(hipMalloc((void**)&d_buf, sizeof(mine) * (size_t)(n + 1)));
  struct Free { int32_t* p; ~Free() { if (p) hipFree(p); } } guard{d_buf};  // (also on the error returns below)
  TF_HIP(hipMemcpyAsync(d_buf, mine, sizeof(mine), hipMemcpyHostToDevice, v->stream));
  TF_NCCL(g_rccl.AllGather(d_buf, d_buf + 8, sizeof(mine), ncclUint8, comm, v->stream));
  std::vector<int32_t> all((size_t)8 * n);
  TF_HIP(hipMemcpyAsync(all.data(), d_buf + 8, sizeof(mine) * (size_t)n, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  bool ok = true;
  for (int r = 0; r < n; ++r) {
    const int32_t* p = &all[(size_t)8 * r];
    const long long w = (long long)p[2] + p[3] + p[4];
    if (p[2] != mine[2] || p[3] != mine[3] || p[4] != mine[4]) ok = false;
    if (r > 0 && p[0] != all[(size_t)8 * (r - 1) + 1]) ok = false;                       // contiguous, in rank order
    // a slab with a rank below it holds its whole down band (and stops the reads of the rank two below): >= w + 1 keys
    if ((long long)p[1] - (long long)p[0] < (r > 0 ? w + 1 : 1)) ok = false;
  }
  c.neighbours_ok = ok;
  c.checked = true;
  return TF_OK;
}

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
  // (a receive slot without a neighbour is never written: its count stays zero, no memset per exchange)
  TF_HIP(hipMemset(c.d_recv, 0, block * (size_t)(c.nranks > 2 ? c.nranks : 2)));
  c.cap_records = cap_records;
  return TF_OK;
}

// pack -> transport -> unpack on ONE stream (xs, or the handle's); dirty_par >= 0: the owned face neighbours of every ghost
// chunk that arrives join the fused flow's per-frame dirty list (their neighbour was updated on another rank).
// ctl != nullptr (the per-frame exchange of the fused flow, neighbour form): the blocks are SIZED by the frame's own
// selection -- FrameCtl::band_cnt holds, identically on every rank, how many selected chunks lie in this rank's two
// bands and in the two neighbouring bands it receives; selected is a superset of updated, so a block of
// xchg_bucket(count) records holds everything the frame flagged, and sender and receiver agree on every size without
// talking to each other.  The counts reach the host through a pinned word the previous exchange's unpack launch (or a
// launch of its own) publishes; nothing else synchronises.  Without ctl (tf_exchange_boundary on demand, all-gather
// form) the blocks have the caller's fixed capacity.
int comm_exchange(tf_volume* v, int64_t cap_records, int dirty_par, uint32_t stamp, const FrameCtl* ctl, uint32_t tag,
                  const FrameCtl* next_ctl, hipStream_t xs) {
  CommState& c = v->comm;
  const hipStream_t s = xs ? xs : v->stream;  // (the overlapped per-frame exchange runs on the handle's second stream)
  if (!c.comm) { set_error("tf_comm_init has not been called"); return TF_ERR_INVALID; }
  int rc = comm_buffers(v, cap_records);
  if (rc) return rc;
  const size_t block = tf_boundary_block_bytes(c.cap_records);
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(c.comm);
  uint8_t* send = reinterpret_cast<uint8_t*>(c.d_send);
  uint8_t* recv = reinterpret_cast<uint8_t*>(c.d_recv);
  int nblocks = c.nranks, skip = c.rank;
  bool neighbours = c.mode == TF_XCHG_NEIGHBOURS && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
  if (neighbours && !c.checked) {  // slabs wide enough and in rank order?  (one tiny all-gather, once per partition)
    rc = comm_check_partition(v);
    if (rc) return rc;
  }
  neighbours = neighbours && c.neighbours_ok;
  VolumeDev d = v->dev;
  if (dirty_par >= 0) {
    d.work_ids = v->atlas.d_work_ids + (size_t)dirty_par * d.max_chunks;
    d.work_slot = v->atlas.d_work_slot + (size_t)dirty_par * d.max_chunks;
  }
  struct ProfScope {  // HIP events around the whole exchange when tf_profile_enable asked for TF_PROF_XCHG
    tf_volume* v;
    hipStream_t s;
    ProfScope(tf_volume* vv, hipStream_t ss) : v(vv), s(ss) { prof_begin(v, TF_PROF_XCHG, s); }
    ~ProfScope() { prof_end(v, s); }
  } prof_scope(v, s);
  if (neighbours) {
    // down block -> rank - 1, up block -> rank + 1; from rank - 1 comes ITS up block, from rank + 1 its down block.
    // A missing neighbour's receive slot keeps a zero count.
    uint32_t cap_dn = (uint32_t)c.cap_records, cap_up = cap_dn, cap_lo = cap_dn, cap_hi = cap_dn;
    if (ctl) {
      uint32_t cnt[4];
      rc = xchg_band_counts(v, ctl, tag, cnt, s);
      if (rc) return rc;
      cap_dn = xchg_bucket(cnt[0], c.cap_records); cap_up = xchg_bucket(cnt[1], c.cap_records);
      cap_lo = xchg_bucket(cnt[2], c.cap_records); cap_hi = xchg_bucket(cnt[3], c.cap_records);
    }
    rc = boundary_pack_bands2_on(v, send, cap_dn, send + block, cap_up, s);
    if (rc) return rc;
    const bool lower = c.rank > 0, upper = c.rank + 1 < c.nranks;
    const size_t b_dn = tf_boundary_block_bytes(cap_dn), b_up = tf_boundary_block_bytes(cap_up);
    const size_t b_lo = tf_boundary_block_bytes(cap_lo), b_hi = tf_boundary_block_bytes(cap_hi);
    if (lower || upper) {
      TF_NCCL(g_rccl.GroupStart());
      if (lower) {
        TF_NCCL(g_rccl.Send(send, b_dn, ncclUint8, c.rank - 1, comm, s));
        TF_NCCL(g_rccl.Recv(recv, b_lo, ncclUint8, c.rank - 1, comm, s));
      }
      if (upper) {
        TF_NCCL(g_rccl.Send(send + block, b_up, ncclUint8, c.rank + 1, comm, s));
        TF_NCCL(g_rccl.Recv(recv + block, b_hi, ncclUint8, c.rank + 1, comm, s));
      }
      TF_NCCL(g_rccl.GroupEnd());
    }
    c.bytes_received += (uint64_t)(lower ? b_lo : 0) + (uint64_t)(upper ? b_hi : 0);
    c.bytes_sent += (uint64_t)(lower ? b_dn : 0) + (uint64_t)(upper ? b_up : 0);
    c.bound_records += (uint64_t)(lower ? cap_dn : 0) + (uint64_t)(upper ? cap_up : 0);
    c.exchanges += 1;
    // (the launch that stores the ghosts also tells the host the NEXT frame's band counts: that frame's selection ran
    // next to this frame's voxel update, so the next exchange finds its sizes waiting)
    const bool pub = ctl && next_ctl;
    if (pub && !v->h_xchg) {
      TF_HIP(hipHostMalloc((void**)&v->h_xchg, 64, hipHostMallocDefault));
      memset(v->h_xchg, 0, 64);
    }
    const uint32_t pub_seq = pub ? ++v->xchg_seq : 0u;  // (a sequence number of its own: see xchg_band_counts)
    launch_boundary_unpack_blocks(d, recv, 2, -1, cap_lo, dirty_par, stamp, s, recv + block, cap_hi,
                                  pub ? next_ctl : nullptr, pub ? v->h_xchg : nullptr, pub_seq);
    if (pub) { v->xchg_pub_enq = tag + 1u; v->xchg_pub_seq = pub_seq; }
  } else {
    rc = boundary_pack_block_on(v, send, c.cap_records, s);
    if (rc) return rc;
    TF_NCCL(g_rccl.AllGather(send, recv, block, ncclUint8, comm, s));
    c.bytes_received += (uint64_t)(c.nranks > 1 ? c.nranks - 1 : 0) * block;
    c.bytes_sent += (uint64_t)(c.nranks > 1 ? c.nranks - 1 : 0) * block;
    c.bound_records += (uint64_t)(c.nranks > 1 ? c.cap_records : 0);
    c.exchanges += 1;
    launch_boundary_unpack_blocks(d, recv, nblocks, skip, (uint32_t)c.cap_records, dirty_par, stamp, s);
  }
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

int tf_comm_stats_ex(tf_volume* v, int64_t out[8]) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  uint32_t rec[2] = {0, 0};
  TF_HIP(hipMemcpyAsync(rec, &v->dev.vctl->xchg_sent, 8, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  out[0] = (int64_t)v->comm.exchanges;
  out[1] = (int64_t)v->comm.bytes_sent;
  out[2] = (int64_t)v->comm.bytes_received;
  out[3] = (int64_t)rec[0];                      // ghost records written into blocks (all pack entry points)
  out[4] = (int64_t)rec[1];                      // ghost records read out of received blocks
  out[5] = (int64_t)v->comm.bound_records;       // record capacity of the blocks sent
  out[6] = v->comm.neighbours_ok ? TF_XCHG_NEIGHBOURS : TF_XCHG_ALLGATHER;  // the form in use (after the first exchange)
  out[7] = (v->comm.checked ? 1 : 0) | (int64_t)(v->xchg_overlapped << 1);  // bit 0: partition checked; >> 1: exchanges overlapped with an interior mesh pass
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
