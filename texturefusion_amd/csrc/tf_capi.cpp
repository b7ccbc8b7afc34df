// tf_capi.cpp -- C ABI (include/tf_fusion.h) over the gfx950 kernels: volume lifetime, the
// reference's prepare / integrate / finalize flow, the fused per-frame unit, state access.
// Host code only; compiled with hipcc for the runtime API.  No CPU compute fallback exists.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <sched.h>
#include <limits>
#include <map>
#include <memory>
#include <mutex>

#include "tf_host_math.h"
#include "tf_volume.h"

// Page-locking is a property of the PROCESS: two handles (two volumes on one GPU, or a volume per GPU) may be fed from the
// same caller buffers.  The ranges the library locked are counted here; the pages are locked by the first handle that
// registers a range and released by the last that lets go of it.
namespace {
struct LockedRange { size_t n; int refs; };
std::mutex g_locked_mu;
std::map<const uint8_t*, LockedRange> g_locked;
}  // namespace
static int host_range_release(const uint8_t* p) {
  using namespace tf;
  std::lock_guard<std::mutex> lk(g_locked_mu);
  auto it = g_locked.find(p);
  if (it == g_locked.end()) return TF_OK;
  if (--it->second.refs > 0) return TF_OK;
  g_locked.erase(it);
  TF_HIP(hipHostUnregister(const_cast<uint8_t*>(p)));
  return TF_OK;
}

namespace tf {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

static size_t pow2_at_least(size_t x) {
  size_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

template <typename T>
static int dev_alloc(tf_volume* v, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) {
    set_error(std::string("hipMalloc(") + std::to_string(count * sizeof(T)) + " B): " + hipGetErrorString(e));
    return TF_ERR_HIP;
  }
  v->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return TF_OK;
}

int ensure_tmp(tf_volume* v, size_t bytes) {
  if (bytes <= v->d_tmp_bytes) return TF_OK;
  if (v->d_tmp) {
    TF_HIP(hipStreamSynchronize(v->stream));
    TF_HIP(hipFree(v->d_tmp));
    v->d_tmp = nullptr;
    v->d_tmp_bytes = 0;
  }
  size_t want = pow2_at_least(bytes);
  TF_HIP(hipMalloc(&v->d_tmp, want));
  v->d_tmp_bytes = want;
  return TF_OK;
}

int ensure_pinned(tf_volume* v, size_t bytes) {
  if (bytes <= v->h_pinned_bytes) return TF_OK;
  if (v->h_pinned) {
    TF_HIP(hipStreamSynchronize(v->stream));
    TF_HIP(hipHostFree(v->h_pinned));
    v->h_pinned = nullptr;
    v->h_pinned_bytes = 0;
  }
  size_t want = pow2_at_least(bytes);
  TF_HIP(hipHostMalloc(&v->h_pinned, want, hipHostMallocDefault));
  v->h_pinned_bytes = want;
  return TF_OK;
}

void prof_begin(tf_volume* v, int kind, hipStream_t s) {
  v->prof_open = ((v->prof_mask >> kind) & 1u) != 0;
  if (!v->prof_open) return;
  if (!s) s = v->stream;
  ProfEvent pe;
  pe.kind = kind;
  auto get = [&](hipEvent_t* e) {
    if (!v->prof_pool.empty()) { *e = v->prof_pool.back(); v->prof_pool.pop_back(); }
    else hipEventCreate(e);
  };
  get(&pe.a);
  get(&pe.b);
  hipEventRecord(pe.a, s);
  v->prof_events.push_back(pe);
}
void prof_end(tf_volume* v, hipStream_t s) {
  if (!v->prof_open) return;
  v->prof_open = false;
  hipEventRecord(v->prof_events.back().b, s ? s : v->stream);
}

static void prof_collect(tf_volume* v) {
  for (auto& pe : v->prof_events) {
    hipEventSynchronize(pe.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
      v->prof_acc.ms[pe.kind] += ms;
      v->prof_acc.launches[pe.kind] += 1;
    }
    v->prof_pool.push_back(pe.a);
    v->prof_pool.push_back(pe.b);
  }
  v->prof_events.clear();
}

static void refresh_cam(tf_volume* v, int w, int h, float nearP, float farP) {
  v->cam.W = w;
  v->cam.H = h;
  v->cam.fxi = (float)(int)v->fx;  // PinholeCamera::GetFx() returns int (PinholeCamera.h:46-49)
  v->cam.fyi = (float)(int)v->fy;
  v->cam.cxi = (float)(int)v->cx;
  v->cam.cyi = (float)(int)v->cy;
  v->cam.nearP = nearP;
  v->cam.farP = farP;
}

static int status_to_error(uint32_t st) {
  if (!st) return TF_OK;
  std::string m = "device status:";
  if (st & kStPoolFull) m += " chunk pool full (raise tf_config.max_chunks)";
  if (st & kStListFull) m += " visible list full (raise tf_config.max_list)";
  if (st & kStCoarseFull) m += " candidate grid full (raise tf_config.max_coarse)";
  if (st & kStMissing) m += " list names a chunk that does not exist";
  if (st & kStHashFull) m += " hash table full";
  if (st & kStMeshFull) m += " the mesh store is exhausted: no block left for a chunk's first mesh (raise tf_config.mesh_blocks) or for a mesh beyond mesh_max_vertices / mesh_max_triangles (raise tf_config.mesh_overflow_blocks or those)";
  if (st & kStAtlasFull) m += " No enough space for texture storage.";  // std::overflow_error text, Atlas.cpp:53
  if (st & kStXchgFull) m += " a rank's ghost band did not fit the boundary exchange block (raise cap_records)";
  set_error(m);
  if (st & kStAtlasFull) return TF_ERR_ATLAS_FULL;
  if (st & kStMissing) return TF_ERR_MISSING_CHUNK;
  return TF_ERR_CAPACITY;
}

// D2H of the control blocks (synchronises the stream), status -> error code, status cleared.
struct CtlSnap {
  FrameCtl f;
  VolCtl vc;
};
static int fetch_ctl(tf_volume* v, CtlSnap* out) {
  if (!v->h_ctl) TF_HIP(hipHostMalloc((void**)&v->h_ctl, sizeof(FrameCtl) + sizeof(VolCtl), hipHostMallocDefault));
  launch_export_ctl(v->dev.sel.ctl, v->dev.vctl, v->h_ctl, v->stream);
  TF_HIP(hipGetLastError());
  TF_HIP(hipStreamSynchronize(v->stream));
  memcpy(&out->f, v->h_ctl, sizeof(FrameCtl));  // (without the pull counters)
  memcpy(&out->vc, reinterpret_cast<const uint8_t*>(v->h_ctl) + sizeof(FrameCtl), sizeof(VolCtl));
  if (out->vc.status) {
    TF_HIP(hipMemsetAsync(&v->dev.vctl->status, 0, sizeof(uint32_t), v->stream));
    return status_to_error(out->vc.status);
  }
  return TF_OK;
}

}  // namespace tf
int tf::sync_status(tf_volume* v, uint32_t* n_tmp) {
  CtlSnap ctl;
  const int rc = fetch_ctl(v, &ctl);
  if (n_tmp) *n_tmp = ctl.vc.n_tmp;
  return rc;
}
namespace tf {

static int init_device_state(tf_volume* v) {
  VolumeDev& d = v->dev;
  hipStream_t s = v->stream;
  TF_HIP(hipMemsetAsync(d.hent, 0xFF, ((size_t)d.hmask + 1) * sizeof(HEntry), s));  // key = empty
  TF_HIP(hipMemsetAsync(d.mark_epoch, 0, (size_t)d.max_chunks * 4, s));
  TF_HIP(hipMemsetAsync(d.erase_epoch, 0, (size_t)d.max_chunks * 4, s));
  TF_HIP(hipMemsetAsync(d.nbr, 0, (size_t)d.max_chunks * kNbrWords * 4, s));  // nothing known, nothing checked (create_seq = 0: k_reset_ctl)
  d.seq = 1;
  d.xl_par = 0;  // (the lists' counters: k_reset_ctl)
  TF_HIP(hipMemsetAsync(d.phase_buf, 0, (size_t)kPhaseWaves * 16 * 8, s));
  v->clear_floor = 0;
  for (int k = 0; k < tf_volume::kSelSets; ++k) {
    d.sel = v->selbuf[k];
    launch_reset_ctl(d, k == 0, s);
  }
  v->cur_sel = 0;
  d.sel = v->selbuf[0];
  launch_fill_pool(d, 0, d.max_chunks, s);
  TF_HIP(hipMemsetAsync(d.obs_key, 0xFF, sizeof(unsigned long long) * ((size_t)d.obs_mask + 1), s));  // no observations
  launch_init_meshes(d, s);  // chunkManager.Reset(): allMeshes.clear() (ChunkManager.cpp:272-275)
  TF_HIP(hipGetLastError());
  v->host_list_n = -1;
  v->epoch = 0;
  v->mesh_epoch = 0;
  v->mesh_par = 0;
  v->n_primed = 0;
  v->xchg_pub_enq = 0;  // (epochs start over; the publish sequence numbers the host waits for do not)
  return TF_OK;
}

// The kernels of Chisel::PrepareIntersectChunks (Structure/Chisel.h:103-140).  The fused per-frame
// unit skips the stand-alone slot lookup: k_integrate<FUSED> does it per chunk.
// Selections made ahead for frames of a previous streaming call (n_ahead) depend on the camera, the truncation
// model and the partition: a setter that changes one of these discards them, so that the next call selects again.
static void discard_primed(tf_volume* v) {
  for (int k = 0; k < v->n_primed; ++k) {
    VolumeDev d = v->dev;
    d.sel = v->selbuf[(v->cur_sel + 1 + k) % tf_volume::kSelSets];
    launch_reset_ctl(d, false, v->stream);
  }
  v->n_primed = 0;
  v->xchg_pub_enq = 0;  // (band counts published for a discarded selection are stale)
}

}  // namespace tf (the two functions below are shared with tf_unit.hip: declared in tf_volume.h)
int tf::xchg_band_counts(tf_volume* v, const tf::FrameCtl* ctl, uint32_t tag, uint32_t cnt[4], hipStream_t s) {
  using namespace tf;
  if (!s) s = v->stream;
  if (!v->h_xchg) {
    TF_HIP(hipHostMalloc((void**)&v->h_xchg, 64, hipHostMallocDefault));
    memset(v->h_xchg, 0, 64);
  }
  if (v->xchg_pub_enq != tag) {  // nobody published this frame's counts behind an earlier exchange: do it now
    // (the word the host waits for is a publish SEQUENCE number, never reused: a frame that replaces a discarded selection
    // has the same epoch tag as the frame it replaces, and the old publish may still sit in the pinned word)
    const uint32_t seq = ++v->xchg_seq;
    launch_xchg_publish(ctl, v->h_xchg, seq, s);
    TF_HIP(hipGetLastError());
    v->xchg_pub_enq = tag;
    v->xchg_pub_seq = seq;
  }
  // the publishing launch sits behind the selection on the stream; the words arrive with a system-scope release
  volatile uint32_t* w = v->h_xchg;
  const uint32_t want = v->xchg_pub_seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spin = 0; __atomic_load_n(&w[0], __ATOMIC_ACQUIRE) != want; ++spin) {
    if ((spin & 0xFFFu) == 0xFFFu) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
        set_error("band counts of the frame never arrived (device stalled?)");
        return TF_ERR_HIP;
      }
      std::this_thread::yield();
    }
  }
  for (int q = 0; q < 4; ++q) cnt[q] = w[1 + q];
  return TF_OK;
}
int tf::launch_prepare(tf_volume* v, const tf::Pose& pose, bool with_acquire, hipStream_t s) {
  using namespace tf;
  if (!s) s = v->stream;
  const SelectConsts sc = make_select_consts(pose.p, v->res);
  prof_begin(v, TF_PROF_BBOX, s);
  launch_bbox(v->dev, v->frame.depth, v->cam, pose, s);
  prof_end(v, s);
  prof_begin(v, TF_PROF_SELECT, s);
  launch_select(v->dev, v->frame.depth, v->cam, v->ig, pose, v->res, /*emit=*/!with_acquire, s);
  prof_end(v, s);
  if (with_acquire) {  // call-by-call flow: reference-ordered list, then slot lookup
    prof_begin(v, TF_PROF_SCAN, s);
    launch_scan(v->dev, sc.step, s);
    prof_end(v, s);
    prof_begin(v, TF_PROF_EMIT, s);
    launch_acquire(v->dev, s);
    prof_end(v, s);
  }
  return TF_OK;
}

namespace tf {
// Make the device-resident list equal to the caller's list (upload + slot lookup if it is not
// the list the last tf_prepare produced).
static int sync_list(tf_volume* v, const int32_t* ids, int64_t n) {
  if (n > (int64_t)v->dev.max_list) {
    set_error("list longer than tf_config.max_list");
    return TF_ERR_CAPACITY;
  }
  if (v->host_list_n == n && (n == 0 || memcmp(v->host_list.data(), ids, (size_t)n * 12) == 0))
    return TF_OK;
  int rc = ensure_pinned(v, (size_t)n * 16 + 16);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));  // pinned staging may still be in flight
  int32_t* st = reinterpret_cast<int32_t*>(v->h_pinned);
  for (int64_t i = 0; i < n; ++i) {
    st[4 * i] = ids[3 * i];
    st[4 * i + 1] = ids[3 * i + 1];
    st[4 * i + 2] = ids[3 * i + 2];
    st[4 * i + 3] = 0;
  }
  uint32_t* cnt = reinterpret_cast<uint32_t*>(st + 4 * n);
  cnt[0] = cnt[1] = (uint32_t)n;  // n_list and n_front: a plain list
  if (n) TF_HIP(hipMemcpyAsync(v->dev.sel.list_id, st, (size_t)n * 16, hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipMemcpyAsync(&v->dev.sel.ctl->n_list, cnt, 8, hipMemcpyHostToDevice, v->stream));
  if (n) {
    TF_HIP(hipMemsetAsync(v->dev.sel.list_new, 0, (size_t)n, v->stream));
    TF_HIP(hipMemsetAsync(v->dev.sel.list_needs, 0, (size_t)n, v->stream));
  }
  launch_lookup(v->dev, (uint32_t)n, v->stream);
  TF_HIP(hipGetLastError());
  TF_HIP(hipStreamSynchronize(v->stream));
  v->host_list.assign(ids, ids + 3 * n);
  v->host_list_n = n;
  v->host_needs.assign((size_t)n, 0);
  v->host_new.assign((size_t)n, 0);
  v->host_flags_n = n;
  return TF_OK;
}
// the caller's flags are what the device list already holds?
static bool flags_current(const tf_volume* v, const std::vector<uint8_t>& mirror, const uint8_t* flags, int64_t n) {
  return v->host_flags_n == n && v->host_list_n == n && (int64_t)mirror.size() == n && memcmp(mirror.data(), flags, (size_t)n) == 0;
}

}  // namespace tf

using namespace tf;

extern "C" {

const char* tf_last_error(void) { return g_err.c_str(); }

int tf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

#undef tf_volume_create
int tf_volume_create(const int32_t chunk_dim[3], float resolution, int use_color,
                     const tf_config* cfg, tf_volume** out) {
  return tf_volume_create_sized(chunk_dim, resolution, use_color, cfg, sizeof(tf_config), out);
}

int tf_volume_create_sized(const int32_t chunk_dim[3], float resolution, int use_color, const tf_config* cfg,
                           size_t cfg_bytes, tf_volume** out) {
  if (!out || !chunk_dim) { set_error("null argument"); return TF_ERR_INVALID; }
  *out = nullptr;
  if (chunk_dim[0] != 8 || chunk_dim[1] != 8 || chunk_dim[2] != 8) {
    set_error("chunk size must be 8x8x8 (the reference kernel hard-codes 512-voxel chunks)");
    return TF_ERR_INVALID;
  }
  if (!(resolution > 0.f)) { set_error("resolution must be positive"); return TF_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device visible: this library has no CPU fallback");
    return TF_ERR_NO_DEVICE;
  }
  tf_volume* v = new tf_volume();
  v->host_defer = tf::host_defer_default();
  memset(&v->cfg, 0, sizeof(v->cfg));
  if (cfg) memcpy(&v->cfg, cfg, std::min(cfg_bytes, sizeof(v->cfg)));  // (fields the caller's header lacks: defaults)
  if (v->cfg.max_chunks <= 0) v->cfg.max_chunks = 1ll << 20;
  v->cfg.max_chunks = (v->cfg.max_chunks + 63) & ~63ll;  // 64 allocation stripes
  if (v->cfg.max_chunks > (1ll << 27)) {
    // (pool slots are 32-bit everywhere; the hash table -- two entries per slot, indexed with 32 bits -- and the per-slot
    // tables bound the pool well before that.  2^27 chunks would be 1 TiB of voxels: HBM is the limit, not an encoding.)
    set_error("tf_config.max_chunks must not exceed 2^27");
    delete v;
    return TF_ERR_INVALID;
  }
  if (v->cfg.max_list <= 0) v->cfg.max_list = 1ll << 19;
  if (v->cfg.max_coarse <= 0) v->cfg.max_coarse = 1ll << 20;
  if (v->cfg.atlas_w <= 0) v->cfg.atlas_w = 13824;
  if (v->cfg.atlas_h <= 0) v->cfg.atlas_h = 13824;
  if (v->cfg.max_keyframes <= 0) v->cfg.max_keyframes = 64;
  if (v->cfg.mesh_max_vertices <= 0) v->cfg.mesh_max_vertices = 256;
  if (v->cfg.mesh_max_triangles <= 0) v->cfg.mesh_max_triangles = 512;
  v->cfg.mesh_max_vertices = std::min((v->cfg.mesh_max_vertices + 63) & ~63, 2240);    // whole wave rows
  v->cfg.mesh_max_triangles = std::min((v->cfg.mesh_max_triangles + 63) & ~63, 2560);
  if (v->cfg.device < 0 || v->cfg.device >= ndev) {
    set_error("tf_config.device out of range");
    delete v;
    return TF_ERR_INVALID;
  }
  v->device = v->cfg.device;
  v->res = resolution;
  v->use_color = use_color;
  // defaults of MobileFusion::initChiselMap (GCFusion/MobileFusion.h:205-258)
  v->ig = Integ{0.0019f, 0.00152f, 0.001504f, 6.0f, 1.0f};
  refresh_cam(v, 640, 480, 0.01f, 5.0f);

  auto fail = [&](int rc) { tf_volume_destroy(v); return rc; };
  if (hipSetDevice(v->device) != hipSuccess) { set_error("hipSetDevice failed"); return fail(TF_ERR_HIP); }
  if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("hipStreamCreate failed");
    return fail(TF_ERR_HIP);
  }
  v->own_stream = true;
  VolumeDev& d = v->dev;
  memset(&d, 0, sizeof(d));
  d.max_chunks = (uint32_t)v->cfg.max_chunks;
  d.max_list = (uint32_t)v->cfg.max_list;
  d.max_coarse = (uint32_t)v->cfg.max_coarse;
  d.part_lo = std::numeric_limits<int32_t>::min();
  d.part_hi = std::numeric_limits<int32_t>::max();
  d.part_a = 1; d.part_b = 0; d.part_c = 0;
  const size_t hcap = pow2_at_least((size_t)d.max_chunks * 2);
  d.hmask = (uint32_t)(hcap - 1);
  int rc;
  if ((rc = dev_alloc(v, &d.tsdf, (size_t)d.max_chunks * kChunkVoxels))) return fail(rc);
  if ((rc = dev_alloc(v, &d.color, (size_t)d.max_chunks * kChunkVoxels))) return fail(rc);
  if ((rc = dev_alloc(v, &d.hent, hcap))) return fail(rc);
  if ((rc = dev_alloc(v, &d.summ, (size_t)d.max_chunks))) return fail(rc);
  if ((rc = dev_alloc(v, &d.nbr, (size_t)d.max_chunks * kNbrWords))) return fail(rc);
  if ((rc = dev_alloc(v, &d.xl_ent, (size_t)d.max_chunks * 2))) return fail(rc);
  if ((rc = dev_alloc(v, &d.mark_epoch, (size_t)d.max_chunks * 2))) return fail(rc);  // mark | erase, one allocation
  d.erase_epoch = d.mark_epoch + d.max_chunks;
  if ((rc = dev_alloc(v, &d.phase_buf, (size_t)kPhaseWaves * 16))) return fail(rc);
  if ((rc = dev_alloc(v, &d.vctl, (size_t)1))) return fail(rc);
  {  // Chunk::observations on the device: four (chunk, keyframe) pairs per chunk of capacity, at least 64 k
    const size_t ocap = pow2_at_least(std::max<size_t>((size_t)d.max_chunks * 4, (size_t)1 << 16));
    d.obs_mask = (uint32_t)(ocap - 1);
    if ((rc = dev_alloc(v, &d.obs_key, ocap))) return fail(rc);
    if ((rc = dev_alloc(v, &d.obs_q, ocap))) return fail(rc);
  }
  d.mesh_cv = (uint32_t)v->cfg.mesh_max_vertices;
  d.mesh_ct = (uint32_t)v->cfg.mesh_max_triangles;
  {  // the mesh store's small pool: handed out block by block to the chunks that get a mesh (MeshRec::block)
    int64_t nb = v->cfg.mesh_blocks;
    if (nb <= 0) nb = v->cfg.max_chunks;  // every chunk can own a mesh (the reference's allMeshes never drops one)
    if (nb > v->cfg.max_chunks) nb = v->cfg.max_chunks;
    v->cfg.mesh_blocks = nb;
    d.mesh_blocks = (uint32_t)nb;
  }
  if ((rc = dev_alloc(v, &d.mesh_v, (size_t)d.mesh_blocks * kMeshPlanes * d.mesh_cv))) return fail(rc);
  if ((rc = dev_alloc(v, &d.mesh_t, (size_t)d.mesh_blocks * 3 * d.mesh_ct))) return fail(rc);
  {  // the large pool, for meshes beyond CV / CT
    int64_t nb = v->cfg.mesh_overflow_blocks;
    // (one full-size block per 64 pool slots: a scene with conflicting surfaces -- walls fused a few centimetres apart -- puts
    // meshes of a thousand vertices into hundreds of chunks, tests/test_gpu_neighbours.py; max_chunks / 256 lost meshes there)
    if (nb == 0) nb = std::max<int64_t>(256, v->cfg.max_chunks / 64);
    if (nb < 0) nb = 0;
    if (nb > 65535) nb = 65535;
    d.ovf_blocks = (uint32_t)nb;
    if (nb) {
      if ((rc = dev_alloc(v, &d.ovf_v, (size_t)nb * kMeshPlanes * kOvfCV))) return fail(rc);
      if ((rc = dev_alloc(v, &d.ovf_t, (size_t)nb * 3 * kOvfCT))) return fail(rc);
      if ((rc = dev_alloc(v, &d.ovf_vlist, (size_t)nb * kOvfCV))) return fail(rc);
    }
  }
  // the records, and behind them the free rings of the two pools (one word per block: blk_release / blk_take, tf_devfn.h)
  if ((rc = dev_alloc(v, &d.mesh_rec, (size_t)d.max_chunks + ((size_t)blk_ring_len(d.mesh_blocks) + blk_ring_len(d.ovf_blocks)) * 4 / sizeof(MeshRec) + 1))) return fail(rc);
  if ((rc = dev_alloc(v, &d.mesh_nbr, (size_t)kMeshShards * mesh_shard_rows(d.max_chunks) * 32))) return fail(rc);
  if ((rc = dev_alloc(v, &d.mesh_cnt, (size_t)2 * kMeshCntWords))) return fail(rc);
  if ((rc = dev_alloc(v, &d.reset_list, (size_t)kMeshShards * mesh_shard_rows(d.max_chunks)))) return fail(rc);
  for (int k = 0; k < tf_volume::kSelSets; ++k) {
    SelBuf& L = v->selbuf[k];
    if ((rc = dev_alloc(v, &L.masks, (size_t)d.max_coarse))) return fail(rc);
    if ((rc = dev_alloc(v, &L.offsets, (size_t)d.max_coarse))) return fail(rc);
    if (hipMemsetAsync(L.offsets, 0, sizeof(uint32_t) * (size_t)d.max_coarse, v->stream) != hipSuccess) { set_error("hipMemsetAsync failed"); return fail(TF_ERR_HIP); }  // k_scan's stamped tile words: no stamp yet
    if ((rc = dev_alloc(v, &L.list_id, (size_t)d.max_list))) return fail(rc);
    if ((rc = dev_alloc(v, &L.list_pre, (size_t)d.max_list * 4))) return fail(rc);
    if ((rc = dev_alloc(v, &L.list_slot, (size_t)d.max_list))) return fail(rc);
    if ((rc = dev_alloc(v, &L.list_ent, (size_t)d.max_list))) return fail(rc);
    if ((rc = dev_alloc(v, &L.list_new, (size_t)d.max_list))) return fail(rc);
    if ((rc = dev_alloc(v, &L.list_needs, (size_t)d.max_list + 64))) return fail(rc);  // (+64: kf_store_body reads the flags in 16-byte words, four per thread)
    if ((rc = dev_alloc(v, &L.list_quality, (size_t)d.max_list))) return fail(rc);
    if ((rc = dev_alloc(v, &L.list_rows, (size_t)d.max_list))) return fail(rc);
    if ((rc = dev_alloc(v, &L.cen, (size_t)3 * kChunkVoxels))) return fail(rc);
    if ((rc = dev_alloc(v, &L.ctl, (size_t)1))) return fail(rc);
  }
  if ((rc = init_device_state(v))) return fail(rc);
  if ((rc = atlas_init(v))) return fail(rc);
  if (hipStreamSynchronize(v->stream) != hipSuccess) { set_error("device init failed"); return fail(TF_ERR_HIP); }
  *out = v;
  return TF_OK;
}

int tf_volume_destroy(tf_volume* v) {
  if (!v) return TF_OK;
  hipSetDevice(v->device);
  if (v->stream) hipStreamSynchronize(v->stream);
  tf_keyframe_unit_release(v);
  prof_collect(v);
  for (hipEvent_t e : v->prof_pool) hipEventDestroy(e);
  atlas_destroy(v);
  comm_destroy(v);
  for (void* p : v->allocs) hipFree(p);
  if (v->d_depth) hipFree(v->d_depth);
  if (v->d_rgba) hipFree(v->d_rgba);
  if (v->d_quality) hipFree(v->d_quality);
  if (v->d_tmp) hipFree(v->d_tmp);
  if (v->d_group) hipFree(v->d_group);
  if (v->h_pinned) hipHostFree(v->h_pinned);
  if (v->h_ctl) hipHostFree(v->h_ctl);
  if (v->h_progress) hipHostFree(v->h_progress);
  v->h_progress = nullptr;
  if (v->h_xchg) hipHostFree(v->h_xchg);
  v->h_xchg = nullptr;
  if (v->xstream) { hipStreamSynchronize(v->xstream); hipStreamDestroy(v->xstream); v->xstream = nullptr; }
  if (v->read_stream) { hipStreamSynchronize(v->read_stream); hipStreamDestroy(v->read_stream); v->read_stream = nullptr; }
  if (v->read_ev) { hipEventDestroy(v->read_ev); v->read_ev = nullptr; }
  if (v->d_snap) { hipFree(v->d_snap); v->d_snap = nullptr; }
  if (v->ev_fork) { hipEventDestroy(v->ev_fork); v->ev_fork = nullptr; }
  if (v->ev_join) { hipEventDestroy(v->ev_join); v->ev_join = nullptr; }
  for (const tf_volume::HostRange& r : v->host_ranges) (void)host_range_release(r.locked);
  v->host_ranges.clear();
  for (int k = 0; k < tf_volume::kHostRing; ++k) {
    if (v->hslot[k].h) hipHostFree(v->hslot[k].h);
    if (v->hslot[k].d) hipFree(v->hslot[k].d);
    if (v->hslot[k].copied) hipEventDestroy(v->hslot[k].copied);
  }
  if (v->host_trace[5] > 0 && getenv("TF_HOST_TRACE") && atoi(getenv("TF_HOST_TRACE")))
    fprintf(stderr, "tf host frames: %.0f calls; per call us: wait kernels %.1f, wait upload %.1f, staging copy %.1f, "
                    "upload enqueue %.1f, launches %.1f; copies a launch waited for in the stream: %ld\n", v->host_trace[5],
            v->host_trace[0] / v->host_trace[5], v->host_trace[1] / v->host_trace[5], v->host_trace[2] / v->host_trace[5],
            v->host_trace[3] / v->host_trace[5], v->host_trace[4] / v->host_trace[5], v->host_waits);
  delete v->copy_pool;
  v->copy_pool = nullptr;
  if (v->copy_stream) hipStreamDestroy(v->copy_stream);
  if (v->copy_stream2) hipStreamDestroy(v->copy_stream2);
  if (v->copy_join) hipEventDestroy(v->copy_join);
  if (v->own_stream && v->stream) hipStreamDestroy(v->stream);
  delete v;
  return TF_OK;
}

int tf_volume_reset(tf_volume* v) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int rc = tf_keyframe_unit_release(v);  // Frame::validChunks of the keyframes integrated so far
  if (rc) return rc;
  rc = init_device_state(v);
  if (rc) return rc;
  return atlas_reset(v);
}

int tf_set_stream(tf_volume* v, void* hip_stream) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  TF_HIP(hipStreamSynchronize(v->stream));
  if (v->own_stream && v->stream) hipStreamDestroy(v->stream);
  v->stream = reinterpret_cast<hipStream_t>(hip_stream);
  v->own_stream = false;
  return TF_OK;
}

int tf_set_camera(tf_volume* v, float fx, float fy, float cx, float cy, int width, int height,
                  float near_plane, float far_plane) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (width <= 0 || height <= 0 || (width & 7)) {
    set_error("image width must be a positive multiple of 8 (the reference reads 8 pixels per step, ChunkManager.h:326)");
    return TF_ERR_INVALID;
  }
  discard_primed(v);
  v->fx = fx; v->fy = fy; v->cx = cx; v->cy = cy;
  refresh_cam(v, width, height, near_plane, far_plane);
  v->frame_bound = false;
  return TF_OK;
}

int tf_set_truncation(tf_volume* v, float q, float l, float c, float s) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  discard_primed(v);
  v->ig.quad = q; v->ig.lin = l; v->ig.cons = c; v->ig.scale = s;
  return TF_OK;
}

int tf_set_weight(tf_volume* v, float w) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  v->ig.weight = w;
  return TF_OK;
}

int tf_frame_upload(tf_volume* v, const float* depth, const uint8_t* rgba, const float* quality) {
  if (!v || !depth) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const size_t npix = (size_t)v->cam.W * v->cam.H;
  if (v->img_pixels != npix) {
    TF_HIP(hipStreamSynchronize(v->stream));
    if (v->d_depth) hipFree(v->d_depth);
    if (v->d_rgba) hipFree(v->d_rgba);
    if (v->d_quality) hipFree(v->d_quality);
    v->d_depth = nullptr; v->d_rgba = nullptr; v->d_quality = nullptr;
    TF_HIP(hipMalloc((void**)&v->d_depth, npix * 4));
    TF_HIP(hipMalloc((void**)&v->d_rgba, npix * 4));
    TF_HIP(hipMalloc((void**)&v->d_quality, npix * 4));
    v->img_pixels = npix;
  }
  int rc = ensure_pinned(v, npix * 12);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));  // previous use of the staging buffer
  uint8_t* st = reinterpret_cast<uint8_t*>(v->h_pinned);
  memcpy(st, depth, npix * 4);
  TF_HIP(hipMemcpyAsync(v->d_depth, st, npix * 4, hipMemcpyHostToDevice, v->stream));
  if (rgba) {
    memcpy(st + npix * 4, rgba, npix * 4);
    TF_HIP(hipMemcpyAsync(v->d_rgba, st + npix * 4, npix * 4, hipMemcpyHostToDevice, v->stream));
  }
  if (quality) {
    memcpy(st + npix * 8, quality, npix * 4);
    TF_HIP(hipMemcpyAsync(v->d_quality, st + npix * 8, npix * 4, hipMemcpyHostToDevice, v->stream));
  }
  v->frame.depth = v->d_depth;
  v->frame.rgba = rgba ? reinterpret_cast<const uchar4*>(v->d_rgba) : nullptr;
  v->frame.quality = quality ? v->d_quality : nullptr;
  v->frame_bound = true;
  return TF_OK;
}

// The caller-side RGBA staging loop of MobileFusion.cpp:144-163 on the device: the keyframe's RGB (3 bytes
// per pixel) and colorValidFlag travel as they are (4 B/pixel instead of the 4 B/pixel RGBA plus the host
// loop) and are packed into the RGBA image the path consumes: valid ? (r, g, b, 1) : 0.
int tf_frame_upload_rgb(tf_volume* v, const float* depth, const uint8_t* rgb, const uint8_t* color_valid,
                        const float* quality) {
  if (!v || !depth || !rgb || !color_valid) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int rc = tf_frame_upload(v, depth, nullptr, quality);
  if (rc) return rc;
  const size_t npix = (size_t)v->cam.W * v->cam.H;
  rc = ensure_tmp(v, npix * 4);
  if (rc) return rc;
  uint8_t* st = reinterpret_cast<uint8_t*>(v->h_pinned);  // [0, 12 npix): tf_frame_upload used depth (0..4) and quality (8..12)
  memcpy(st + npix * 4, rgb, npix * 3);
  memcpy(st + npix * 7, color_valid, npix);
  uint8_t* dt = reinterpret_cast<uint8_t*>(v->d_tmp);
  TF_HIP(hipMemcpyAsync(dt, st + npix * 4, npix * 4, hipMemcpyHostToDevice, v->stream));
  launch_pack_rgba(dt, dt + npix * 3, reinterpret_cast<uchar4*>(v->d_rgba), (uint32_t)npix, v->stream);
  TF_HIP(hipGetLastError());
  v->frame.rgba = reinterpret_cast<const uchar4*>(v->d_rgba);
  return TF_OK;
}

int tf_frame_bind_device(tf_volume* v, const float* d_depth, const uint8_t* d_rgba,
                         const float* d_quality) {
  if (!v || !d_depth) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if ((reinterpret_cast<uintptr_t>(d_depth) & 15) || (reinterpret_cast<uintptr_t>(d_rgba) & 3) ||
      (reinterpret_cast<uintptr_t>(d_quality) & 3)) {
    set_error("device images must be aligned (depth 16 B, rgba/quality 4 B)");
    return TF_ERR_INVALID;
  }
  v->frame.depth = d_depth;
  v->frame.rgba = reinterpret_cast<const uchar4*>(d_rgba);
  v->frame.quality = d_quality;
  v->frame_bound = true;
  return TF_OK;
}

int tf_prepare(tf_volume* v, const float pose[12], int32_t* out_ids, uint8_t* out_new, int64_t cap,
               int64_t* n) {
  if (!v || !pose || !n) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (!v->frame_bound) { set_error("no frame bound (tf_frame_upload / tf_frame_bind_device)"); return TF_ERR_INVALID; }
  Pose P;
  memcpy(P.p, pose, sizeof(P.p));
  v->host_list_n = -1;
  // the list goes to host-visible memory behind the kernels that make it: one synchronisation per call
  const size_t cap_list = (size_t)v->dev.max_list;
  int rc = ensure_pinned(v, cap_list * 17);
  if (rc) return rc;
  rc = launch_prepare(v, P, true);
  if (rc) return rc;
  TF_HIP(hipGetLastError());
  int32_t* st = reinterpret_cast<int32_t*>(v->h_pinned);
  uint8_t* stn = reinterpret_cast<uint8_t*>(st + 4 * cap_list);
  launch_export_list(v->dev, reinterpret_cast<int4*>(st), stn, (uint32_t)cap_list, v->stream);
  TF_HIP(hipGetLastError());
  CtlSnap ctl;
  rc = fetch_ctl(v, &ctl);
  if (rc) return rc;
  const int64_t cnt = ctl.f.n_list;
  *n = cnt;
  if (cnt > cap) { set_error("output capacity too small for the visible list"); return TF_ERR_CAPACITY; }
  if (cnt == 0) { v->host_list.clear(); v->host_list_n = 0; v->host_flags_n = -2; return TF_OK; }
  v->host_list.resize((size_t)cnt * 3);
  for (int64_t i = 0; i < cnt; ++i) {
    v->host_list[3 * i] = st[4 * i];
    v->host_list[3 * i + 1] = st[4 * i + 1];
    v->host_list[3 * i + 2] = st[4 * i + 2];
  }
  v->host_list_n = cnt;
  v->host_needs.assign((size_t)cnt, 0);  // (the selection leaves needsUpdate clear)
  v->host_new.assign(stn, stn + cnt);
  v->host_flags_n = cnt;
  if (out_ids) memcpy(out_ids, v->host_list.data(), (size_t)cnt * 12);
  if (out_new) memcpy(out_new, stn, (size_t)cnt);
  return TF_OK;
}

int tf_integrate(tf_volume* v, const float pose[12], const int32_t* ids, int64_t n,
                 int integrate_flag, int use_color, int use_quality, uint8_t* inout_needs_update,
                 float* out_quality) {
  if (!v || !pose || (n > 0 && (!ids || !inout_needs_update))) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (!v->frame_bound) { set_error("no frame bound"); return TF_ERR_INVALID; }
  if (n < 1) return TF_OK;  // Chisel.h:228
  if (use_color && !v->frame.rgba) { set_error("use_color set but the bound frame has no colour image"); return TF_ERR_INVALID; }
  if (use_quality && (!use_color || !v->frame.quality)) { set_error("use_quality needs colour and a quality image"); return TF_ERR_INVALID; }
  int rc = sync_list(v, ids, n);
  if (rc) return rc;
  const size_t npad = (size_t)((n + 3) & ~(int64_t)3);
  rc = ensure_pinned(v, 2 * npad + (size_t)n * 4 + 16);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* st = reinterpret_cast<uint8_t*>(v->h_pinned);           // flags going in
  uint8_t* st_out = st + npad;                                      // flags coming back (written by the device)
  float* stq = reinterpret_cast<float*>(st + 2 * npad);
  uint32_t* stat = reinterpret_cast<uint32_t*>(st + 2 * npad + (size_t)n * 4);
  if (!flags_current(v, v->host_needs, inout_needs_update, n)) {
    memcpy(st, inout_needs_update, (size_t)n);
    TF_HIP(hipMemcpyAsync(v->dev.sel.list_needs, st, (size_t)n, hipMemcpyHostToDevice, v->stream));
  }
  v->host_flags_n = -2;
  Pose P;
  memcpy(P.p, pose, sizeof(P.p));
  prof_begin(v, TF_PROF_INTEGRATE);
  launch_integrate(v->dev, v->frame, v->cam, v->ig, P, v->res, integrate_flag, use_color != 0,
                   use_quality != 0, 0, v->stream);
  prof_end(v);
  TF_HIP(hipGetLastError());
  // (a depth-only call has no quality sum: chunkObservationQuality stays 0, ProjectionIntegrator.cpp:212-238)
  const bool want_q = out_quality && use_color;
  *stat = 0;
  launch_export_integrate(v->dev, (uint32_t)n, st_out, want_q ? stq : nullptr, stat, v->stream);
  TF_HIP(hipGetLastError());
  TF_HIP(hipStreamSynchronize(v->stream));
  if (*stat) {  // (rare: the usual path reads, resets and translates the sticky bits)
    CtlSnap ctl;
    rc = fetch_ctl(v, &ctl);
    if (rc) return rc;
  }
  memcpy(inout_needs_update, st_out, (size_t)n);
  v->host_needs.assign(st_out, st_out + n);
  v->host_flags_n = n;
  if (want_q) memcpy(out_quality, stq, (size_t)n * 4);
  else if (out_quality) memset(out_quality, 0, (size_t)n * 4);
  return TF_OK;
}

int tf_integrate_depth_group(tf_volume* v, int32_t n_frames, const float* const* d_depth, const float* poses12,
                             const int32_t* ids, int64_t n, int integrate_flag, uint8_t* inout_needs_update) {
  if (!v || !d_depth || !poses12 || (n > 0 && (!ids || !inout_needs_update))) { set_error("null argument"); return TF_ERR_INVALID; }
  if (n_frames < 1 || n_frames > 6) { set_error("a keyframe group holds 1..6 local frames (GCFusion/MobileFusion.h: integrateLocalFrameNum)"); return TF_ERR_INVALID; }
  for (int f = 0; f < n_frames; ++f)
    if (!d_depth[f]) { set_error("null depth image"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n < 1) return TF_OK;  // Chisel.h:228
  int rc = sync_list(v, ids, n);
  if (rc) return rc;
  const size_t npad = (size_t)((n + 3) & ~(int64_t)3);
  rc = ensure_pinned(v, npad);
  if (rc) return rc;
  // scratch: per frame the list records (2 x float4 per entry, stride 4) and the centroid table
  const size_t pre_bytes = (size_t)n_frames * 4 * v->dev.max_list * sizeof(float4);
  const size_t cen_bytes = (size_t)n_frames * 3 * kChunkVoxels * sizeof(float);
  rc = ensure_tmp(v, pre_bytes + cen_bytes);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* st = reinterpret_cast<uint8_t*>(v->h_pinned);
  if (!flags_current(v, v->host_needs, inout_needs_update, n)) {
    memcpy(st, inout_needs_update, (size_t)n);
    TF_HIP(hipMemcpyAsync(v->dev.sel.list_needs, st, (size_t)n, hipMemcpyHostToDevice, v->stream));
  }
  v->host_flags_n = -2;
  float4* pre = reinterpret_cast<float4*>(v->d_tmp);
  float* cen = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(v->d_tmp) + pre_bytes);
  prof_begin(v, TF_PROF_INTEGRATE);
  launch_integrate_group(v->dev, n_frames, d_depth, poses12, pre, cen, v->cam, v->ig, v->res, integrate_flag, v->stream);
  prof_end(v);
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(st, v->dev.sel.list_needs, (size_t)n, hipMemcpyDeviceToHost, v->stream));
  CtlSnap ctl;
  rc = fetch_ctl(v, &ctl);
  if (rc) return rc;
  memcpy(inout_needs_update, st, (size_t)n);
  v->host_needs.assign(st, st + n);
  v->host_flags_n = n;
  return TF_OK;
}

int tf_integrate_depth_group_host(tf_volume* v, int32_t n_frames, const float* const* depth, const float* poses12,
                                  const int32_t* ids, int64_t n, int integrate_flag, uint8_t* inout_needs_update) {
  if (!v || !depth) { set_error("null argument"); return TF_ERR_INVALID; }
  if (n_frames < 1 || n_frames > 6) { set_error("a keyframe group holds 1..6 local frames"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const size_t npix = (size_t)v->cam.W * v->cam.H;
  if (v->d_group_pixels != npix) {
    if (v->d_group) hipFree(v->d_group);
    v->d_group = nullptr;
    TF_HIP(hipMalloc((void**)&v->d_group, 6 * npix * sizeof(float)));
    v->d_group_pixels = npix;
  }
  const float* dd[6];
  for (int f = 0; f < n_frames; ++f) {
    if (!depth[f]) { set_error("null depth image"); return TF_ERR_INVALID; }
    dd[f] = v->d_group + (size_t)f * npix;
    TF_HIP(hipMemcpyAsync(v->d_group + (size_t)f * npix, depth[f], npix * sizeof(float), hipMemcpyHostToDevice, v->stream));
  }
  return tf_integrate_depth_group(v, n_frames, dd, poses12, ids, n, integrate_flag, inout_needs_update);
}

int tf_finalize(tf_volume* v, const int32_t* ids, const uint8_t* needs_update, const uint8_t* is_new,
                int64_t n, int32_t* out_valid, int64_t* n_valid) {
  if (!v || (n > 0 && (!ids || !needs_update || !is_new))) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int64_t nv = 0;
  if (n > 0) {
    int rc = sync_list(v, ids, n);
    if (rc) return rc;
    rc = ensure_pinned(v, (size_t)n * 2);
    if (rc) return rc;
    TF_HIP(hipStreamSynchronize(v->stream));
    uint8_t* st = reinterpret_cast<uint8_t*>(v->h_pinned);
    if (!flags_current(v, v->host_needs, needs_update, n)) {
      memcpy(st, needs_update, (size_t)n);
      TF_HIP(hipMemcpyAsync(v->dev.sel.list_needs, st, (size_t)n, hipMemcpyHostToDevice, v->stream));
      v->host_needs.assign(needs_update, needs_update + n);
    }
    if (!flags_current(v, v->host_new, is_new, n)) {
      memcpy(st + n, is_new, (size_t)n);
      TF_HIP(hipMemcpyAsync(v->dev.sel.list_new, st + n, (size_t)n, hipMemcpyHostToDevice, v->stream));
      v->host_new.assign(is_new, is_new + n);
    }
    v->host_flags_n = n;
    discard_primed(v);  // (GarbageCollect parks chunks: selections made ahead for a stream may name them as permanently alive)
    prof_begin(v, TF_PROF_FINALIZE);
    launch_finalize(v->dev, v->epoch++, v->stream);
    prof_end(v);
    TF_HIP(hipGetLastError());
    // validChunks in list order (Chisel.h:204); pure host bookkeeping on the caller's flags
    for (int64_t i = 0; i < n; ++i)
      if (needs_update[i]) {
        if (out_valid) memcpy(out_valid + 3 * nv, ids + 3 * i, 12);
        ++nv;
      }
    CtlSnap ctl;
    rc = fetch_ctl(v, &ctl);
    if (rc == TF_ERR_MISSING_CHUNK) rc = TF_OK;  // finalize only flags/parks; absent chunks are skipped
    if (rc) return rc;
  }
  if (n_valid) *n_valid = nv;
  return TF_OK;
}

// Chisel::UpdateMeshes -> CompressMeshes -> GeneratePatches(label = this frame) -> UpdateAtlas over the dirty
// chunks of ONE integrated frame (GCFusion/MobileFusion.cpp:327-382 without the host-side view selection): `sel`
// holds the frame's visible list with its needsUpdate flags, `frame_epoch` the finalize epoch of that frame.
// The patch stage (adjacency exchange, slot hand-out, projection, blit) reads meshes and images only; it is left
// PENDING here and rides on the next frame's launch next to that frame's voxel update (AtlasState::pend_patch).
}  // extern "C" (C++ linkage for the helper below)
static int patch_launched(tf_volume* v);
int tf::fused_arm(tf_volume* v) {
  AtlasState& a = v->atlas;
  if (a.fused_armed) return TF_OK;
  // first textured frame after a reset / a call-by-call atlas call: empty work lists
  AtlasCtl::Set z[2];
  memset(z, 0, sizeof(z));
  TF_HIP(hipMemcpyAsync(&a.d_actl->set[0], z, sizeof(z), hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipMemsetAsync(a.d_patch_cnt, 0, sizeof(uint32_t) * 2 * kMeshShards * 16, v->stream));
  TF_HIP(hipMemsetAsync(a.d_wl_cnt, 0, sizeof(uint32_t) * 2 * kMeshShards * 16, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  a.fused_par = 0;
  a.fused_armed = true;
  return TF_OK;
}

// Neighbour table (VolumeDev::nbr): every filter launch carries a seq above that of every launch ahead of it on the
// stream, so that a row it checks outlives exactly the key insertions that come later (VolCtl::create_seq).  After 2^32
// launches the table starts over.
uint32_t tf::nbr_next_seq(tf_volume* v) {
  if (v->dev.seq >= 0xFFFFFFF0u) {
    (void)hipMemsetAsync(v->dev.nbr, 0, (size_t)v->dev.max_chunks * kNbrWords * 4, v->stream);
    (void)hipMemsetAsync(&v->dev.vctl->create_seq, 0, 4, v->stream);
    v->dev.seq = 1;
  }
  return ++v->dev.seq;
}

int tf::boundary_pack_block_on(tf_volume* v, void* d_block, int64_t cap_records, hipStream_t s) {
  uint8_t* blk = reinterpret_cast<uint8_t*>(d_block);
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, s));
  launch_boundary_pack(v->dev, blk + 16, (uint32_t)cap_records, s);
  v->dev.xl_par ^= 1u;  // (the voxel kernels list what they touch from now on into the other list)
  launch_boundary_headers(v->dev, reinterpret_cast<uint32_t*>(blk), (uint32_t)cap_records, nullptr, 0, s);  // the count travels in-band
  TF_HIP(hipGetLastError());
  return TF_OK;
}
int tf::boundary_pack_bands2_on(tf_volume* v, void* d_block_down, int64_t cap_down, void* d_block_up, int64_t cap_up, hipStream_t s) {
  // ONE launch: the counters live in VolCtl::xchg_cnt, the last workgroup writes the in-band counts and re-arms them
  // (round 4: two memsets + the pack + a header launch -- the exchange of a frame was seven stream operations at their
  // launch floor, 49 us with nothing on the wire; profiles/r5/README.md)
  launch_boundary_pack_bands(v->dev, reinterpret_cast<uint8_t*>(d_block_down), reinterpret_cast<uint8_t*>(d_block_up),
                             (uint32_t)cap_down, (uint32_t)cap_up, s);
  v->dev.xl_par ^= 1u;  // (the voxel kernels list what they touch from now on into the other list)
  TF_HIP(hipGetLastError());
  return TF_OK;
}

// claimed: the dirty set of this frame is already in the lists of the current parity -- K-A built it (FrameStage::claim_par
// = the parity used here), or the caller ran launch_dirty_frame over each of its lists (the keyframe unit)
int tf::texture_stage(tf_volume* v, const SelBuf& sel, const FrameImages& img, uint32_t frame_epoch,
                      const float* pose_inv16, int32_t frame_id, bool claimed, const FrameCtl* next_ctl, bool ride_filter,
                      bool sized_xchg, int phase, const KfStoreArgs* store) {
  AtlasState& a = v->atlas;
  if (phase == 2) {
    // second half of a stage whose first half (dirty set + interior meshes) ran before the caller's own exchange
    if (!a.phase1_on || a.phase1_epoch != frame_epoch) { set_error("tf_texture_frame_device_phase: phase 2 without phase 1 of the same frame"); return TF_ERR_INVALID; }
    a.phase1_on = false;
    const int par = a.fused_par;
    a.fused_par ^= 1;
    VolumeDev d = v->dev;
    d.sel = sel;
    d.work_ids = a.d_work_ids + (size_t)par * d.max_chunks;
    d.work_slot = a.d_work_slot + (size_t)par * d.max_chunks;
    prof_begin(v, TF_PROF_MESH);
    d.seq = nbr_next_seq(v);
    launch_mesh(d, v->mesh_par, d.work_ids, &d.actl->set[par].n_work, d.max_chunks, ++v->mesh_epoch, v->res, true, par ^ 1, 0u, nullptr, par,
                v->stream, nullptr, &v->cam, /*cls=*/2);
    v->mesh_par ^= 1;
    prof_end(v);
    return texture_stage_finish(v, img, frame_epoch, pose_inv16, frame_id, par);
  }
  if (a.phase1_on) { set_error("a texture stage is half done: call tf_texture_frame_device_phase(.., 2) first"); return TF_ERR_INVALID; }
  // A patch stage still pending here (the previous textured frame's) must read its meshes before this frame's mesher
  // rewrites them: it goes out on its own first -- or (ride_filter: the keyframe unit, which has no k_frame launch for it)
  // rides on this frame's FILTER launch (launch_mesh below).  In the per-frame stream the stage rides on k_frame instead:
  // next to the filter the two latency chains cost more than the stage costs K-A (profiles/r4/README.md, runs c2 / s1;
  // variants/r4_experiments.patch has the knob).
  const bool ride = a.pend_patch.on && a.fused_armed && ride_filter;
  int rc = TF_OK;
  if (!ride) { rc = patch_flush(v); if (rc) return rc; }
  rc = fused_arm(v);
  if (rc) return rc;
  const int par = a.fused_par;
  if (phase != 1) a.fused_par ^= 1;  // (phase 1: the caller's unpack still appends to this parity's list)
  VolumeDev d = v->dev;
  d.sel = sel;
  d.work_ids = a.d_work_ids + (size_t)par * d.max_chunks;
  d.work_slot = a.d_work_slot + (size_t)par * d.max_chunks;
  prof_begin(v, TF_PROF_DIRTY);
  // meshesToUpdate = everything marked since CompressMeshes last cleared it (Chisel.h:192-208, Chisel.cpp:146).  In a
  // textured stream that is this frame's chunks (stamps <= frame_epoch are cleared); after frames integrated without
  // the textured unit the older marks are still there and the general dirty list takes over for this frame.
  if (claimed) { /* nothing to launch: the shard lists of `par` hold the set */ }
  else if (v->clear_floor < frame_epoch) launch_dirty_backlog(d, par, v->clear_floor, v->stream);
  else launch_dirty_frame(d, par, frame_epoch + 1u, v->stream);
  prof_end(v);
  // (the filter's form follows the dirty-list length of an earlier frame: the kernel leaves it in host-visible memory,
  // read here without any synchronisation -- whatever value is there is good enough)
  const uint32_t len_guess = a.h_dirty_len ? *reinterpret_cast<volatile uint32_t*>(a.h_dirty_len) : 0u;
  const PatchStage prev = a.pend_patch.st;  // (copied: the pending record is overwritten below)
  // one filter + mesher pass over the frame's dirty set (cls: every chunk / interior chunks only / boundary chunks only);
  // the shard lists of this parity are walked in any case: empty when K-A did not claim -- the previous frame's mesher
  // re-armed them
  // (new_seq = false: the pass runs NEXT TO the unpack launch of an overlapped exchange, which carries the same seq -- a
  // key that launch inserts voids what this pass checks of the neighbour table)
  auto mesh_pass = [&](int cls, const uint32_t* flat_count, bool with_hint, bool with_ride, bool new_seq = true) -> bool {
    prof_begin(v, TF_PROF_MESH);
    d.seq = new_seq ? nbr_next_seq(v) : v->dev.seq;
    std::unique_ptr<AtlasWriteScope> aw;  // (a patch stage riding on the filter launch writes atlas texels)
    if (with_ride) aw.reset(new AtlasWriteScope(v, prev.kf.kf_id));
    const bool rode = launch_mesh(d, v->mesh_par, d.work_ids, flat_count, d.max_chunks, ++v->mesh_epoch, v->res, true, par ^ 1,
                                  len_guess, with_hint ? a.h_dirty_len : nullptr, par, v->stream, with_ride ? &prev : nullptr,
                                  &v->cam, cls, store);
    aw.reset();
    store = nullptr;  // (once)
    v->mesh_par ^= 1;
    prof_end(v);
    return rode;
  };
  bool rode = false;
  if (phase == 1) {
    // a caller with its own transport: the interior meshes now, the boundary ones behind its unpack (phase 2).  The flat
    // list of this parity holds the frame's own entries (stream order: the unpack has not run yet)
    rode = mesh_pass(1, &d.actl->set[par].n_work, true, false);
    a.phase1_on = true;
    a.phase1_epoch = frame_epoch;
    return TF_OK;
  }
  if (v->comm_cap > 0) {  // multi-GPU: ghost bands of this frame's updates, before the mesher reads them
    // (sized by this frame's selection when the fused stream ran it -- sel.ctl then holds the band counts, tagged with the
    // frame's epoch + 1; lists of the call-by-call flow and the keyframe unit carry no counts: fixed-capacity blocks)
    const FrameCtl* xc = sized_xchg ? sel.ctl : nullptr;
    const FrameCtl* xn = sized_xchg ? next_ctl : nullptr;
    // The exchange leaves the critical path: pack -> send / receive -> unpack run on a second stream while the main
    // stream filters and meshes the INTERIOR chunks of the dirty set -- those whose 27-chunk neighbourhood is owned, so that
    // nothing they read can arrive from another rank -- and only the boundary chunks (and what the arriving ghosts add to
    // the dirty set) wait for it.  Needs the dirty set in the shard lists K-A filled (claimed: the flat list, which the
    // unpack launch appends to, is then ignored by the interior pass) and no patch stage riding on the filter launch.
    // (sized_xchg = the stage comes from the fused stream: `claimed` then means K-A's shard lists; the keyframe unit also
    // passes claimed, with its dirty set in the FLAT list)
    const bool overlap = claimed && sized_xchg && !ride && v->xchg_overlap;
    if (overlap) {
      if (!v->xstream) {
        TF_HIP(hipStreamCreateWithFlags(&v->xstream, hipStreamNonBlocking));
        TF_HIP(hipEventCreateWithFlags(&v->ev_fork, hipEventDisableTiming));
        TF_HIP(hipEventCreateWithFlags(&v->ev_join, hipEventDisableTiming));
      }
      TF_HIP(hipEventRecord(v->ev_fork, v->stream));          // behind the voxel update of this frame
      (void)nbr_next_seq(v);   // the unpack launch and the interior pass next to it share this seq
      // the interior pass goes onto the main stream FIRST: the host then spends its time on the exchange's enqueue (band
      // counts, pack, transport, unpack: ~15 us) while the device already filters and meshes (a kernel trace of the other
      // order showed the main stream idle for 12 us behind every voxel update)
      rode = mesh_pass(1, &d.vctl->zero_word, true, false, /*new_seq=*/false);
      TF_HIP(hipStreamWaitEvent(v->xstream, v->ev_fork, 0));
      rc = comm_exchange(v, v->comm_cap, par, frame_epoch + 1u, xc, frame_epoch + 1u, xn, v->xstream);
      if (rc) return rc;
      TF_HIP(hipEventRecord(v->ev_join, v->xstream));
      prof_begin(v, TF_PROF_XCHG_WAIT);                        // what of the exchange is NOT hidden behind the interior pass
      TF_HIP(hipStreamWaitEvent(v->stream, v->ev_join, 0));
      prof_end(v);
      rode = mesh_pass(2, &d.actl->set[par].n_work, false, false);
      v->xchg_overlapped += 1;
    } else {
      rc = comm_exchange(v, v->comm_cap, par, frame_epoch + 1u, xc, frame_epoch + 1u, xn);
      if (rc) return rc;
      rode = mesh_pass(0, &d.actl->set[par].n_work, true, ride);
    }
  } else {
    rode = mesh_pass(0, &d.actl->set[par].n_work, true, ride);
  }
  if (ride) {
    if (!rode) { set_error("internal: the pending patch stage found no filter launch"); return TF_ERR_INVALID; }
    rc = patch_launched(v);
    if (rc) return rc;
  }
  return texture_stage_finish(v, img, frame_epoch, pose_inv16, frame_id, par);
}
// the tail of a texture stage: this frame's patch stage becomes the pending one
int tf::texture_stage_finish(tf_volume* v, const FrameImages& img, uint32_t frame_epoch, const float* pose_inv16, int32_t frame_id,
                             int par) {
  AtlasState& a = v->atlas;
  // (CompressMeshes' neighbour exchange, the list of chunks that own a mesh and the slot candidates are produced
  // by the mesher and consumed by the patch kernel: no kernel of their own in the fused flow)
  KfDev kf;
  memset(&kf, 0, sizeof(kf));
  kf.rgb = reinterpret_cast<const uint8_t*>(img.rgba);
  kf.depth = img.depth;
  kf.stride = 4;
  kf.kf_id = frame_id;
  static const int pdbg = getenv("TF_PATCH_DBG") ? atoi(getenv("TF_PATCH_DBG")) : 0;  // triage switch
  kf.pad[0] = pdbg;
  memcpy(kf.T, pose_inv16, 64);
  a.pend_patch.on = true;
  a.pend_patch.st.par = par;
  a.pend_patch.st.kf = kf;
  a.pend_patch.host_slot = -1;
  v->clear_floor = frame_epoch + 1u;  // CompressMeshes cleared meshesToUpdate
  return TF_OK;
}
// the pending patch stage has been put on the stream (as a role of a frame / filter launch or on its own)
static int patch_launched(tf_volume* v) {
  AtlasState& a = v->atlas;
  a.pend_patch.on = false;
  if (a.pend_patch.host_slot >= 0) {  // its frame's staging slot is free once this launch is through = the next one has started
    v->hslot[a.pend_patch.host_slot].free_when = v->progress_seq + 1u;
    a.pend_patch.host_slot = -1;
  }
  return TF_OK;
}
namespace tf {
int patch_flush(tf_volume* v) {  // the pending patch stage as a launch of its own
  AtlasState& a = v->atlas;
  if (!a.pend_patch.on) return TF_OK;
  {
    AtlasWriteScope aw(v, a.pend_patch.st.kf.kf_id);
    launch_patch_fused(v, v->dev, a.pend_patch.st.par, a.pend_patch.st.kf, v->stream);
  }
  TF_HIP(hipGetLastError());
  return patch_launched(v);
}
}  // namespace tf
extern "C" {

// Software-pipelined enqueue of n frames on the handle's stream: launch i carries K-A of frame i,
// K-C of frame i+1 and K-B of frame i+2 as independent block ranges of one kernel (launch_frame),
// so the only synchronisation is the kernel boundary.  The arrays may hold n_ahead (<= 2) frames more
// than the n that are integrated: their selection stages run in this call's last launches and stay
// valid ("primed") for the next call, which then starts with K-A at once -- a stream fed call by call
// costs one launch per frame, not n + 2 per call.  tex != nullptr: after K-A of frame i its dirty chunks
// are meshed and textured from the frame itself (the per-frame unit of BASELINE configs[2]).
struct TexturedArgs {
  const float* pose_inv16;  // per frame: f32(SE3d.inverse().matrix()) of the frame's pose
  int32_t first_frame_id;
};
static int enqueue_frames(tf_volume* v, int64_t n, int64_t n_ahead, const float* const* d_depth,
                          const uint8_t* const* d_rgba, const float* poses12, const TexturedArgs* tex) {
  const int NS = tf_volume::kSelSets;
  const int64_t n_all = n + n_ahead;
  auto stage = [&](int64_t f, FrameStage* st) {
    st->sel = v->selbuf[(v->cur_sel + 1 + f) % NS];
    st->img.depth = d_depth[f];
    st->img.rgba = (d_rgba && d_rgba[f]) ? reinterpret_cast<const uchar4*>(d_rgba[f]) : nullptr;
    st->img.quality = nullptr;
    memcpy(st->pose.p, poses12 + 12 * f, sizeof(st->pose.p));
    st->epoch = v->epoch + (uint32_t)f;
    st->coarse_summ = tex == nullptr;  // a TSDF-only stream: K-A skips the class ballots (tf_device.h)
    st->claim_par = -1;
  };
  if (tex) { int rc = fused_arm(v); if (rc) return rc; }  // (before the first launch appends to the shard lists)
  // how many leading frames already went through their selection stages in the previous call?
  int primed = 0;
  for (int k = 0; k < v->n_primed && k < n_all; ++k) {
    const tf_volume::Primed& p = v->primed[k];
    if (p.depth != d_depth[k] || memcmp(p.pose, poses12 + 12 * k, 48) != 0) break;
    primed = k + 1;
  }
  if (primed < v->n_primed) {  // the stream changed course: discard the selections made ahead
    for (int k = 0; k < v->n_primed; ++k) {
      VolumeDev d = v->dev;
      d.sel = v->selbuf[(v->cur_sel + 1 + k) % NS];
      launch_reset_ctl(d, false, v->stream);
    }
    primed = 0;
    v->xchg_pub_enq = 0;
  }
  v->n_primed = 0;
  // a primed frame 0 has its list (K-B and K-C ran), a primed frame 1 its bounding box (K-B ran)
  for (int64_t i = (primed == 2 ? 0 : (primed == 1 ? -1 : -2)); i < n; ++i) {
    FrameStage cur, nxt, nx2;
    const bool hc = i >= 0;
    bool hn = (i + 1 >= 0) && (i + 1 < n_all), h2 = (i + 2 < n_all);
    if (i + 2 < primed) h2 = false;               // K-B of that frame ran in the previous call
    if (primed >= 1 && i + 1 == 0) hn = false;    // K-C of frame 0 ran in the previous call
    if (hc) stage(i, &cur);
    // K-A builds the frame's dirty set itself when the frame's own chunks are all there is to mesh (marks of earlier,
    // untextured frames still waiting -> the general dirty list, texture_stage)
    // Both shortcuts pay for room-sized frames (a K-A wave has one or two chunks) and cost on hall-sized ones, where a
    // wave walks ten chunks and the claim's dependent hops add up behind each of them (hall: k_frame 250 -> 296 us for
    // 26 us of k_dirty_frame, profiles/r3): decided by the length of an earlier frame's dirty list, which the mesher's
    // filter leaves in host-visible memory (no synchronisation; any value gives correct results).
    const uint32_t dirty_hint = v->atlas.h_dirty_len ? *reinterpret_cast<volatile uint32_t*>(v->atlas.h_dirty_len) : 0u;
    static const uint32_t small_max = getenv("TF_SMALL_FRAME") ? (uint32_t)atoi(getenv("TF_SMALL_FRAME")) : 20000u;
    const bool small_frame = dirty_hint <= small_max;
    if (hc) cur.small_frame = small_frame;
    const bool claimed = hc && tex && small_frame && v->clear_floor >= cur.epoch;
    if (claimed) cur.claim_par = v->atlas.fused_par;
    if (hn) stage(i + 1, &nxt);
    if (h2) stage(i + 2, &nx2);
    if (!hc && !hn && !h2) continue;
    // the patch stage of the previous textured frame rides on this launch when it carries a colour voxel update
    // (the fused kernel has no depth-only instance with that role); otherwise it goes out on its own first
    AtlasState::PendPatch& pp = v->atlas.pend_patch;
    const bool carry = pp.on && hc && cur.img.rgba != nullptr;
    if (pp.on && hc && !carry) { int rc = patch_flush(v); if (rc) return rc; }
    if (hc) prof_begin(v, TF_PROF_INTEGRATE);
    if (carry) {  // (the launch writes atlas texels: ordered against tf_atlas_snapshot_rows)
      AtlasWriteScope aw(v, pp.st.kf.kf_id);
      launch_frame(v->dev, &cur, hn ? &nxt : nullptr, h2 ? &nx2 : nullptr, &pp.st, v->cam, v->ig, v->res, v->stream, v->h_progress,
                   &v->progress_seq);
    } else {
      launch_frame(v->dev, hc ? &cur : nullptr, hn ? &nxt : nullptr, h2 ? &nx2 : nullptr, nullptr, v->cam, v->ig, v->res, v->stream,
                   v->h_progress, &v->progress_seq);
    }
    if (hc) prof_end(v);
    if (carry) { int rc = patch_launched(v); if (rc) return rc; }
    if (hc && tex) {
      // (the next frame's selection role rode on this launch: its band counts can be published behind this frame's exchange)
      int rc = texture_stage(v, cur.sel, cur.img, cur.epoch, tex->pose_inv16 + 16 * i, tex->first_frame_id + (int32_t)i, claimed,
                             hn ? nxt.sel.ctl : nullptr, false, /*sized_xchg=*/true);
      if (rc) return rc;
    }
  }
  for (int64_t k = 0; k < n_ahead; ++k) {
    tf_volume::Primed& p = v->primed[k];
    p.depth = d_depth[n + k];
    memcpy(p.pose, poses12 + 12 * (n + k), 48);
  }
  v->n_primed = (int)n_ahead;
  v->epoch += (uint32_t)n;
  v->cur_sel = (int)((v->cur_sel + n) % NS);
  v->dev.sel = v->selbuf[v->cur_sel];
  v->host_list_n = -1;
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_integrate_frame(tf_volume* v, const float pose[12], int use_color) {
  if (!v || !pose) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (!v->frame_bound) { set_error("no frame bound"); return TF_ERR_INVALID; }
  const float* dd[1] = {v->frame.depth};
  const uint8_t* dc[1] = {use_color ? reinterpret_cast<const uint8_t*>(v->frame.rgba) : nullptr};
  return enqueue_frames(v, 1, 0, dd, dc, pose, nullptr);
}

static int bind_frame(tf_volume* v, const float* d_depth, const uint8_t* d_rgba) {  // tf_frame_bind_device without the entry checks
  v->frame.depth = d_depth;
  v->frame.rgba = reinterpret_cast<const uchar4*>(d_rgba);
  v->frame.quality = nullptr;
  v->frame_bound = true;
  return TF_OK;
}
// a frame of the host ring has been enqueued: its staging slot is free when the last launch that reads its device
// images is through -- the frame's own launch, or the one that carries its pending patch stage
static int host_slot_done(tf_volume* v, int slot) {
  AtlasState::PendPatch& pp = v->atlas.pend_patch;
  if (pp.on && pp.host_slot < 0) { pp.host_slot = slot; return TF_OK; }
  v->hslot[slot].free_when = v->progress_seq + 1u;  // through = a later frame launch has started
  return TF_OK;
}

// Blocks until the last launch that reads a staging slot's device images is through.  No stream event: the frame
// launches stamp tf_volume::h_progress when they start.  The launch that follows the slot's last reader is normally on
// the stream already (the entry point runs three frames behind); if none comes (the caller changed entry points), the
// stream is drained instead.
static int host_slot_wait(tf_volume* v, tf_volume::HostSlot& s) {
  if (!s.free_when) return TF_OK;
  volatile uint32_t* p = v->h_progress;
  if ((int32_t)(v->progress_seq - s.free_when) < 0) {  // no launch that would stamp it is on the stream
    TF_HIP(hipStreamSynchronize(v->stream));
    s.free_when = 0;
    return TF_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 0;; ++spin) {
    if ((int32_t)(*p - s.free_when) >= 0) break;
    static const int poll_sleep = getenv("TF_HOST_POLL_SLEEP_US") ? atoi(getenv("TF_HOST_POLL_SLEEP_US")) : 0;
    if (poll_sleep > 0) { std::this_thread::sleep_for(std::chrono::microseconds(poll_sleep)); continue; }
    __builtin_ia32_pause();
    if ((spin & 1023u) == 1023u) {
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us > 20000.0) { TF_HIP(hipStreamSynchronize(v->stream)); break; }  // (a stalled device: fail through the API)
      if (us > 200.0) sched_yield();
    }
  }
  s.free_when = 0;
  return TF_OK;
}

// the H2D copy of a staged host frame must be through before a launch reads its device images: nothing to do when the
// copy event is already complete (the usual case: the entry point runs behind), otherwise the stream waits for it
static int host_copy_ready(tf_volume* v, tf_volume::Pending* p) {
  if (p->copied) return TF_OK;
  hipEvent_t ev = v->hslot[p->slot].copied;
  const hipError_t q = hipEventQuery(ev);
  if (q == hipErrorNotReady) {
    TF_HIP(hipStreamWaitEvent(v->stream, ev, 0));
    if (v->host_trace[5] >= 0) v->host_waits += 1;
  } else if (q != hipSuccess) {
    TF_HIP(q);
  }
  p->copied = true;
  return TF_OK;
}

static int check_frames(int64_t n_all, const float* const* d_depth, const uint8_t* const* d_rgba) {
  for (int64_t f = 0; f < n_all; ++f)
    if ((reinterpret_cast<uintptr_t>(d_depth[f]) & 15) || !d_depth[f] ||
        (d_rgba && (reinterpret_cast<uintptr_t>(d_rgba[f]) & 3))) {
      set_error("device images must be aligned (depth 16 B, rgba 4 B)");
      return TF_ERR_INVALID;
    }
  return TF_OK;
}

int tf_integrate_frames_device(tf_volume* v, int64_t n_frames, const float* const* d_depth,
                               const uint8_t* const* d_rgba, const float* poses12) {
  return tf_stream_frames_device(v, n_frames, 0, d_depth, d_rgba, poses12);
}

int tf_stream_frames_device(tf_volume* v, int64_t n_frames, int64_t n_ahead, const float* const* d_depth,
                            const uint8_t* const* d_rgba, const float* poses12) {
  if (!v || !d_depth || !poses12) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV_STREAM(v);
  if (n_ahead < 0 || n_ahead > 2) { set_error("n_ahead must be 0, 1 or 2"); return TF_ERR_INVALID; }
  if (n_frames <= 0) return TF_OK;
  int rc = check_frames(n_frames + n_ahead, d_depth, d_rgba);
  if (rc) return rc;
  rc = enqueue_frames(v, n_frames, n_ahead, d_depth, d_rgba, poses12, nullptr);
  if (rc) return rc;
  // leave the last frame bound, like a sequence of tf_frame_bind_device calls would
  return bind_frame(v, d_depth[n_frames - 1], d_rgba ? d_rgba[n_frames - 1] : nullptr);
}

int tf_stream_frames_textured_device(tf_volume* v, int64_t n_frames, int64_t n_ahead, const float* const* d_depth,
                                     const uint8_t* const* d_rgba, const float* poses12, const float* pose_inv16,
                                     int32_t first_frame_id) {
  if (!v || !d_depth || !d_rgba || !poses12 || !pose_inv16) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV_STREAM(v);
  if (n_ahead < 0 || n_ahead > 2) { set_error("n_ahead must be 0, 1 or 2"); return TF_ERR_INVALID; }
  if (n_frames <= 0) return TF_OK;
  int rc = check_frames(n_frames + n_ahead, d_depth, d_rgba);
  if (rc) return rc;
  for (int64_t f = 0; f < n_frames; ++f)
    if (!d_rgba[f]) { set_error("the textured flow needs a colour image per frame"); return TF_ERR_INVALID; }
  TexturedArgs tex{pose_inv16, first_frame_id};
  rc = enqueue_frames(v, n_frames, n_ahead, d_depth, d_rgba, poses12, &tex);
  if (rc) return rc;
  // n_ahead > 0: the caller keeps feeding this stream (and keeps the images of the frames in flight alive), so the
  // patch stage of the last frame waits for the next call's first launch; otherwise it goes out now
  if (n_ahead == 0) { rc = patch_flush(v); if (rc) return rc; }
  return bind_frame(v, d_depth[n_frames - 1], d_rgba[n_frames - 1]);
}

// ring of staging slots of the per-frame host path, (re)sized to the camera

static int host_ring_prepare(tf_volume* v) {
  const size_t npix = (size_t)v->cam.W * v->cam.H;
  if (v->hslot_pixels == npix && v->copy_stream) return TF_OK;
  TF_HIP(hipStreamSynchronize(v->stream));
  if (!v->copy_stream) TF_HIP(hipStreamCreateWithFlags(&v->copy_stream, hipStreamNonBlocking));
  TF_HIP(hipStreamSynchronize(v->copy_stream));
  for (int k = 0; k < tf_volume::kHostRing; ++k) {
    tf_volume::HostSlot& s = v->hslot[k];
    if (s.h) hipHostFree(s.h);
    if (s.d) hipFree(s.d);
    s.h = nullptr; s.d = nullptr;
    TF_HIP(hipHostMalloc((void**)&s.h, npix * 8, hipHostMallocDefault));
    TF_HIP(hipMalloc((void**)&s.d, npix * 12));  // depth | colour as uploaded (RGBA, or RGB + valid flags) | RGBA packed from an RGB upload
    if (!s.copied) TF_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    s.free_when = 0;  // (both streams were drained above)
  }
  if (!v->h_progress) {
    TF_HIP(hipHostMalloc((void**)&v->h_progress, 64, hipHostMallocDefault));
    *v->h_progress = v->progress_seq;
  }
  v->hslot_pixels = npix;
  v->hslot_next = 0;
  return TF_OK;
}

}  // extern "C"
bool tf::host_defer_default() {
  static const bool on = !(getenv("TF_HOST_DEFER") && !atoi(getenv("TF_HOST_DEFER")));
  return on;
}
extern "C" {
int tf_host_frame_set_deferral(tf_volume* v, int on) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);  // (frames still in the pipeline go onto the stream under the old setting)
  v->host_defer = on != 0;
  return TF_OK;
}

int tf_host_frame_set_async(tf_volume* v, int on) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  v->host_async = on != 0;
  return TF_OK;
}
int tf_host_frame_fence(tf_volume* v) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV_NOFLUSH(v);
  if (v->last_upload) TF_HIP(hipEventSynchronize(v->last_upload));  // (uploads of one handle complete in order)
  return TF_OK;
}

int tf_host_frame_deferral(tf_volume* v, int32_t* frames_behind, int32_t* ring_slots) {
  // (a null handle answers for a handle as tf_volume_create makes it: TF_HOST_DEFER=0 in the environment turns the deferral
  // off for every new handle, tf_host_frame_set_deferral for one)
  const bool defer = v ? v->host_defer : tf::host_defer_default();
  if (frames_behind) *frames_behind = defer ? tf_volume::kHostDefer : 0;
  if (ring_slots) *ring_slots = tf_volume::kHostRing;
  return TF_OK;
}

}  // extern "C"
extern "C" {
int tf_host_register(tf_volume* v, const void* p, int64_t bytes) {
  if (!v || !p || bytes <= 0) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV_NOFLUSH(v);
  const uint8_t* b = static_cast<const uint8_t*>(p);
  for (const tf_volume::HostRange& r : v->host_ranges)
    if (b >= r.p && b + bytes <= r.p + r.n) return TF_OK;  // already inside a range of this handle
  const uint8_t* locked = b;  // base of the page-locked range that covers [b, b + bytes)
  {
    std::lock_guard<std::mutex> lk(g_locked_mu);
    // a range some handle of this process locked already may CONTAIN this one (depth / colour views inside one arena):
    // hipHostRegister on pages that are locked fails, so the containing range is shared instead
    auto it = g_locked.upper_bound(b);
    const bool have = it != g_locked.begin() && (--it, b < it->first + it->second.n);  // the range at or below b reaches b
    if (have && b + bytes <= it->first + it->second.n) {
      it->second.refs += 1;
      locked = it->first;
    } else {
      auto up = g_locked.lower_bound(b);  // the first range that starts at or above b
      if (have || (up != g_locked.end() && up->first < b + bytes)) {
        set_error("tf_host_register: the buffer overlaps a range that is page-locked already with a different extent "
                  "(register the whole arena once, or ranges that do not overlap)");
        return TF_ERR_INVALID;
      }
      TF_HIP(hipHostRegister(const_cast<uint8_t*>(b), (size_t)bytes, hipHostRegisterDefault));
      g_locked[b] = LockedRange{(size_t)bytes, 1};
    }
  }
  v->host_ranges.push_back({b, (size_t)bytes, locked});
  return TF_OK;
}
int tf_host_unregister(tf_volume* v, const void* p) {
  if (!v || !p) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);  // (frames still in the entry point's pipeline have been uploaded; their launches go out now)
  TF_HIP(hipStreamSynchronize(v->copy_stream ? v->copy_stream : v->stream));
  for (size_t i = 0; i < v->host_ranges.size(); ++i)
    if (v->host_ranges[i].p == static_cast<const uint8_t*>(p)) {
      const uint8_t* locked = v->host_ranges[i].locked;
      v->host_ranges.erase(v->host_ranges.begin() + (long)i);
      return host_range_release(locked);
    }
  set_error("not a registered buffer");
  return TF_ERR_INVALID;
}

int tf_host_frame_times(tf_volume* v, double out[7], int reset) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  out[0] = v->host_trace[5];  // calls that put a frame's launches on the stream
  for (int k = 0; k < 5; ++k) out[1 + k] = v->host_trace[k];  // us: waiting for the device to free a slot | waiting for the slot's last upload | staging copy | upload enqueue | launches
  out[6] = (double)v->host_waits;  // launches that had to wait in the stream for an upload
  if (reset) { for (int k = 0; k < 6; ++k) v->host_trace[k] = 0.0; v->host_waits = 0; }
  return TF_OK;
}

int tf_host_frame_buffers(tf_volume* v, float** depth, uint8_t** rgba) {
  if (!v || !depth || !rgba) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV_NOFLUSH(v);
  int rc = host_ring_prepare(v);
  if (rc) return rc;
  tf_volume::HostSlot& s = v->hslot[v->hslot_next];
  TF_HIP(hipEventSynchronize(s.copied));  // the previous upload out of this slot has left the host buffer
  *depth = reinterpret_cast<float*>(s.h);
  *rgba = s.h + v->hslot_pixels * 4;
  return TF_OK;
}

}  // extern "C"
// rgb != nullptr: the colour image comes as Frame::rgb (3 bytes per pixel) with Frame::colorValidFlag (or none: every pixel
// valid) -- the inputs of the caller's own RGBA staging loops (MobileFusion.cpp:144-163, :232-243), which then run on the
// device behind the upload, on the copy stream
static int integrate_frame_host_impl(tf_volume* v, const float* depth, const uint8_t* rgba, const uint8_t* rgb,
                                     const uint8_t* color_valid, const float pose[12], const float* pose_inv16, int32_t frame_id) {
  if (!v || !depth || !pose) { set_error("null argument"); return TF_ERR_INVALID; }
  if (pose_inv16 && !rgba && !rgb) { set_error("the textured unit needs a colour image"); return TF_ERR_INVALID; }
  TF_DEV_NOFLUSH(v);
  int rc = host_ring_prepare(v);
  if (rc) return rc;
  constexpr bool trace = true;  // per-phase host time (five clock reads per call): tf_host_frame_times; TF_HOST_TRACE=1 prints it at destroy
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto lap = [&](int k, std::chrono::steady_clock::time_point& t) {
    if (!trace) return;
    const auto t1 = now();
    v->host_trace[k] += std::chrono::duration<double, std::micro>(t1 - t).count();
    t = t1;
  };
  auto t = now();
  // The launch pipeline of the streaming entry points, kept alive across per-frame calls: this call integrates the
  // frame that arrived FOUR calls ago, and that launch carries the selection stages of the two frames behind it
  // (K-A(f-4) | K-C(f-3) | K-B(f-2)); frames f-1 and f are only staged and copied.  A copy therefore has a whole
  // call's time to finish before a launch needs it, so the host finds the copy event complete and no wait goes into
  // the stream (a cross-stream wait ahead of a launch costs ~7 us of idle device; with three frames of deferral the
  // newest image a launch reads was uploaded by the call before, and 6-20 % of the launches still waited).  The deferral cannot be
  // observed: every other entry point flushes first (TF_DEV).
  // The launches go out FIRST where that costs nothing -- they need nothing of the new frame -- so that an idle device is
  // at work while this call stages and uploads (it starts ~45 us earlier: 2 % of a 20-frame window).
  const bool defer = v->host_defer;
  constexpr int ND = tf_volume::kHostDefer;
  const float* bound_d = nullptr;
  const uint8_t* bound_c = nullptr;
  auto launch_oldest = [&]() -> int {
    if (!(defer && v->n_pend == ND)) return TF_OK;
    tf_volume::Pending all[ND];
    for (int k = 0; k < ND; ++k) all[k] = v->pend[k];
    tf_volume::Pending &p0 = all[0], &p1 = all[1], &p2 = all[2];  // the launch reads the oldest frame and the two behind it
    for (tf_volume::Pending* q : {&p0, &p1, &p2}) { rc = host_copy_ready(v, q); if (rc) return rc; }
    const float* dd[3] = {p0.d, p1.d, p2.d};
    const uint8_t* dc[3] = {p0.c, p1.c, p2.c};
    float poses[36];
    memcpy(poses, p0.pose, 48); memcpy(poses + 12, p1.pose, 48); memcpy(poses + 24, p2.pose, 48);
    TexturedArgs tex{p0.pinv, p0.fid};
    v->n_pend = 0;  // (helpers below enqueue_frames may pass through TF_DEV: nothing to flush while this call runs)
    rc = enqueue_frames(v, 1, 2, dd, dc, poses, p0.tex ? &tex : nullptr);
    for (int k = 1; k < ND; ++k) v->pend[k - 1] = all[k];
    v->n_pend = ND - 1;
    if (rc) return rc;
    rc = host_slot_done(v, p0.slot);
    if (rc) return rc;
    bound_d = p0.d;
    bound_c = p0.c;
    lap(4, t);
    if (trace) v->host_trace[5] += 1.0;
    return TF_OK;
  };
  // ... but only when they would not wait: the launch also reads the images of the two frames behind the oldest one
  // (selection roles), and the newest of those was uploaded by the PREVIOUS call.  On an idle device (the start of a
  // stream, a caller that paces its frames) that copy is through and the launches go out at once; in a saturated stream
  // it is a few microseconds old -- launching now would put a wait for it into the stream (7 us of idle device per
  // frame, run 46), launching behind the staging copy finds it complete.
  bool early = false;
  if (defer && v->n_pend == ND) {
    const tf_volume::Pending& newest = v->pend[2];  // (the newest frame the launch reads)
    early = newest.copied || hipEventQuery(v->hslot[newest.slot].copied) == hipSuccess;
  }
  if (early) { rc = launch_oldest(); if (rc) return rc; }
  const size_t npix = v->hslot_pixels;
  const int slot_index = v->hslot_next;
  tf_volume::HostSlot& s = v->hslot[slot_index];
  v->hslot_next = (v->hslot_next + 1) % tf_volume::kHostRing;
  // the kernels that read this slot's device images and the upload out of its pinned buffer have finished
  rc = host_slot_wait(v, s);
  if (rc) return rc;
  lap(0, t);
  TF_HIP(hipEventSynchronize(s.copied));
  lap(1, t);
  float* hd = reinterpret_cast<float*>(s.h);
  uint8_t* hc = s.h + npix * 4;
  // images inside registered caller buffers (tf_host_register) go up straight from there
  auto registered = [&](const void* q, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(q);
    for (const tf_volume::HostRange& r : v->host_ranges)
      if (b >= r.p && b + n <= r.p + r.n) return true;
    return false;
  };
  const bool direct = !v->host_ranges.empty() && registered(depth, npix * 4) && (!rgba || registered(rgba, npix * 4)) &&
                      (!rgb || (registered(rgb, npix * 3) && (!color_valid || registered(color_valid, npix))));
  if (!direct) {  // frames composed in tf_host_frame_buffers' slot skip the staging copy
    void* dst[2];
    const void* src[2];
    size_t nb[2];
    int nr = 0;
    if (depth != hd) { dst[nr] = hd; src[nr] = depth; nb[nr++] = npix * 4; }
    void* dst3[3];
    const void* src3[3];
    size_t nb3[3];
    if (rgba && rgba != hc) { dst[nr] = hc; src[nr] = rgba; nb[nr++] = npix * 4; }
    if (rgb) {  // RGB at hc, the valid flags behind it (composed in place by a caller of tf_host_frame_buffers: no copy)
      if (nr) { dst3[0] = dst[0]; src3[0] = src[0]; nb3[0] = nb[0]; }
      if (rgb != hc) { dst3[nr] = hc; src3[nr] = rgb; nb3[nr++] = npix * 3; }
      if (color_valid && color_valid != hc + npix * 3) { dst3[nr] = hc + npix * 3; src3[nr] = color_valid; nb3[nr++] = npix; }
    }
    if (nr) {
      if (!v->copy_pool) {
        const char* e = getenv("TF_COPY_THREADS");
        int helpers = e ? atoi(e) : 7;
        if (helpers < 0) helpers = 0;
        if (helpers > 15) helpers = 15;
        // helpers run where the scheduler puts them (on a shared host pinned helpers gave 1 run in 3 a 10-ms stall --
        // 100.8 us per frame at best, 150+ at worst, against a steady 102.6 unpinned; profiles/r3, run 36)
        const int pin = 0;
        static const int spin_us = getenv("TF_COPY_SPIN_US") ? atoi(getenv("TF_COPY_SPIN_US")) : 200;
        v->copy_pool = new CopyPool(helpers, pin, spin_us);
      }
      if (rgb) v->copy_pool->copy(dst3, src3, nb3, nr);
      else v->copy_pool->copy(dst, src, nb, nr);
    }
  }
  lap(2, t);
  // (depth and colour of a STAGED frame go up as one copy on one copy stream: two streams -- two SDMA queues -- helped a
  // TSDF-only stream in steady state on a quiet host, 62 -> 52-54 us per frame, and doubled the first window behind resident
  // frames on a shared one; profiles/r4/README.md, run s13)
  if (direct) {
    // (depth and colour are two caller arrays = two copies.  The link moves 2.46 MB as ONE copy in 53 us -- 46 GB/s,
    // page-locked by hipHostMalloc or in place alike, tools/h2d_probe.py --; as two copies of 1.2 MB it takes 60 us when
    // they travel side by side on two copy queues and 66 us one behind the other on one queue (profiles/r5/README.md).  A
    // kernel that fetches the images itself -- 16-byte loads out of the mapped pages -- was no faster than the DMA
    // transfers and slowed the step kernels it ran next to: 100 -> 125 us per frame, profiles/r4/README.md)
    if (rgba) {
      if (!v->copy_stream2) {
        TF_HIP(hipStreamCreateWithFlags(&v->copy_stream2, hipStreamNonBlocking));
        TF_HIP(hipEventCreateWithFlags(&v->copy_join, hipEventDisableTiming));
      }
      TF_HIP(hipMemcpyAsync(s.d + npix * 4, rgba, npix * 4, hipMemcpyHostToDevice, v->copy_stream2));
      TF_HIP(hipEventRecord(v->copy_join, v->copy_stream2));
      TF_HIP(hipMemcpyAsync(s.d, depth, npix * 4, hipMemcpyHostToDevice, v->copy_stream));
      TF_HIP(hipStreamWaitEvent(v->copy_stream, v->copy_join, 0));
    } else {
      TF_HIP(hipMemcpyAsync(s.d, depth, npix * 4, hipMemcpyHostToDevice, v->copy_stream));
    }
    if (rgb) {
      TF_HIP(hipMemcpyAsync(s.d + npix * 4, rgb, npix * 3, hipMemcpyHostToDevice, v->copy_stream));
      if (color_valid) TF_HIP(hipMemcpyAsync(s.d + npix * 7, color_valid, npix, hipMemcpyHostToDevice, v->copy_stream));
      launch_pack_rgba(s.d + npix * 4, color_valid ? s.d + npix * 7 : nullptr, reinterpret_cast<uchar4*>(s.d + npix * 8), (uint32_t)npix,
                       v->copy_stream);
      TF_HIP(hipGetLastError());
    }
  } else {
    {
      const size_t up = rgba ? npix * 8 : (rgb ? (color_valid ? npix * 8 : npix * 7) : npix * 4);
      TF_HIP(hipMemcpyAsync(s.d, s.h, up, hipMemcpyHostToDevice, v->copy_stream));
    }
    if (rgb) {  // rgba = valid ? (r, g, b, 1) : 0, behind the upload on the copy stream (null flags: every pixel valid)
      launch_pack_rgba(s.d + npix * 4, color_valid ? s.d + npix * 7 : nullptr, reinterpret_cast<uchar4*>(s.d + npix * 8), (uint32_t)npix,
                       v->copy_stream);
      TF_HIP(hipGetLastError());
    }
  }
  TF_HIP(hipEventRecord(s.copied, v->copy_stream));
  lap(3, t);
  tf_volume::Pending cur;
  cur.d = reinterpret_cast<const float*>(s.d);
  cur.c = rgba ? s.d + npix * 4 : (rgb ? s.d + npix * 8 : nullptr);
  memcpy(cur.pose, pose, sizeof(cur.pose));
  cur.tex = pose_inv16 != nullptr;
  if (pose_inv16) memcpy(cur.pinv, pose_inv16, sizeof(cur.pinv));
  cur.fid = frame_id;
  cur.slot = slot_index;
  cur.copied = false;
  // the caller's buffers are its own again when the call returns: an upload straight out of them must be through -- unless
  // the caller took that on itself (tf_host_frame_set_async: it calls tf_host_frame_fence before it touches a buffer again)
  v->last_upload = s.copied;
  auto wait_direct = [&]() {
    if (v->host_async) return;
    auto tw = now();
    // (TF_HOST_POLL_SLEEP_US > 0: sleep between polls instead of spinning -- several ranks under one CPU quota)
    static const int poll_sleep = getenv("TF_HOST_POLL_SLEEP_US") ? atoi(getenv("TF_HOST_POLL_SLEEP_US")) : 0;
    for (uint32_t spin = 0; hipEventQuery(s.copied) == hipErrorNotReady; ++spin) {
      if (poll_sleep > 0) std::this_thread::sleep_for(std::chrono::microseconds(poll_sleep));
      else if ((spin & 63u) == 63u) __builtin_ia32_pause();
    }
    lap(1, tw);
  };
  if (!defer) {  // integrate at once: two selection-only launches per frame, the stream waits for the copy
    if (direct) { wait_direct(); cur.copied = !v->host_async; }
    rc = host_copy_ready(v, &cur);
    if (rc) return rc;
    const float* dd[1] = {cur.d};
    const uint8_t* dc[1] = {cur.c};
    TexturedArgs tex{cur.pinv, cur.fid};
    rc = enqueue_frames(v, 1, 0, dd, dc, cur.pose, cur.tex ? &tex : nullptr);
    if (rc) return rc;
    rc = host_slot_done(v, slot_index);
    if (rc) return rc;
    return bind_frame(v, cur.d, cur.c);
  }
  if (!early) { rc = launch_oldest(); if (rc) return rc; }
  v->pend[v->n_pend++] = cur;
  if (direct) { wait_direct(); v->pend[v->n_pend - 1].copied = !v->host_async; }
  if (bound_d) return bind_frame(v, bound_d, bound_c);
  return TF_OK;
}
extern "C" {
int tf_integrate_frame_host(tf_volume* v, const float* depth, const uint8_t* rgba, const float pose[12],
                            const float* pose_inv16, int32_t frame_id) {
  return integrate_frame_host_impl(v, depth, rgba, nullptr, nullptr, pose, pose_inv16, frame_id);
}
int tf_integrate_frame_host_rgb(tf_volume* v, const float* depth, const uint8_t* rgb, const uint8_t* color_valid,
                                const float pose[12], const float* pose_inv16, int32_t frame_id) {
  if (!rgb) { set_error("null colour image (tf_integrate_frame_host takes depth-only frames)"); return TF_ERR_INVALID; }
  return integrate_frame_host_impl(v, depth, nullptr, rgb, color_valid, pose, pose_inv16, frame_id);
}

}  // extern "C" (C++ linkage for the helper below)
namespace tf {
// brings the deferred frames of tf_integrate_frame_host onto the stream, oldest first (each launch still carries
// the selection stages of the frames behind it)
int flush_deferred(tf_volume* v) {
  const int n = v->n_pend;
  if (!n) return TF_OK;
  constexpr int ND = tf_volume::kHostDefer;
  tf_volume::Pending p[ND];
  for (int k = 0; k < n; ++k) p[k] = v->pend[k];
  v->n_pend = 0;  // (enqueue_frames' helpers may pass through TF_DEV)
  for (int k = 0; k < n; ++k) {
    int rc = host_copy_ready(v, &p[k]);
    if (rc) return rc;
  }
  for (int k = 0; k < n; ++k) {
    const float* dd[ND];
    const uint8_t* dc[ND];
    float poses[12 * ND];
    for (int j = k; j < n; ++j) { dd[j - k] = p[j].d; dc[j - k] = p[j].c; memcpy(poses + 12 * (j - k), p[j].pose, 48); }
    TexturedArgs tex{p[k].pinv, p[k].fid};
    const int ahead = n - 1 - k < 2 ? n - 1 - k : 2;
    int rc = enqueue_frames(v, 1, ahead, dd, dc, poses, p[k].tex ? &tex : nullptr);
    if (rc) return rc;
    rc = host_slot_done(v, p[k].slot);
    if (rc) return rc;
  }
  return bind_frame(v, p[n - 1].d, p[n - 1].c);
}
}  // namespace tf
extern "C" {

int tf_texture_frame_device(tf_volume* v, const float pose_inv16[16], int32_t frame_id) {
  if (!v || !pose_inv16) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (!v->frame_bound || !v->frame.rgba) { set_error("no colour frame bound"); return TF_ERR_INVALID; }
  if (v->epoch == 0) { set_error("no frame has been integrated"); return TF_ERR_INVALID; }
  return texture_stage(v, v->dev.sel, v->frame, v->epoch - 1u, pose_inv16, frame_id);
}

int tf_texture_frame_device_phase(tf_volume* v, const float pose_inv16[16], int32_t frame_id, int phase) {
  if (!v || !pose_inv16) { set_error("null argument"); return TF_ERR_INVALID; }
  if (phase != 1 && phase != 2) { set_error("phase must be 1 or 2"); return TF_ERR_INVALID; }
  TF_DEV_STREAM(v);
  if (phase == 1 && v->atlas.pend_patch.on) { int rc = patch_flush(v); if (rc) return rc; }
  if (!v->frame_bound || !v->frame.rgba) { set_error("no colour frame bound"); return TF_ERR_INVALID; }
  if (v->epoch == 0) { set_error("no frame has been integrated"); return TF_ERR_INVALID; }
  return texture_stage(v, v->dev.sel, v->frame, v->epoch - 1u, pose_inv16, frame_id, false, nullptr, false, false, phase);
}

int tf_comm_exchange_overlap(tf_volume* v, int on) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  v->xchg_overlap = on != 0;
  return TF_OK;
}

int tf_get_texture_stats(tf_volume* v, tf_texture_stats* out) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  memset(out, 0, sizeof(*out));
  int rc = ensure_tmp(v, 64);
  if (rc) return rc;
  TF_HIP(hipMemsetAsync(v->d_tmp, 0, 48, v->stream));
  {
    const int par = v->atlas.fused_par ^ 1;
    VolumeDev d = v->dev;
    d.work_ids = v->atlas.d_work_ids + (size_t)par * d.max_chunks;
    d.work_slot = v->atlas.d_work_slot + (size_t)par * d.max_chunks;
    launch_texture_stats(d, par, reinterpret_cast<unsigned long long*>(v->d_tmp), v->stream);
  }
  TF_HIP(hipGetLastError());
  unsigned long long r[6];
  TF_HIP(hipMemcpyAsync(r, v->d_tmp, 48, hipMemcpyDeviceToHost, v->stream));
  AtlasCtl c;
  TF_HIP(hipMemcpyAsync(&c, v->dev.actl, sizeof(c), hipMemcpyDeviceToHost, v->stream));
  uint32_t mc[kMeshCntWords];  // the counters of the last mesher launch: rows per shard | {exact tests, rows with a surface cell} per shard
  TF_HIP(hipMemcpyAsync(mc, v->dev.mesh_cnt + (size_t)((v->mesh_par & 1) ^ 1) * kMeshCntWords, sizeof(mc),
                        hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  for (uint32_t k = 0; k < kMeshShards; ++k) {
    out->n_survivors += mc[k * 16] + mc[k * 16 + 1]; out->n_exact += mc[(kMeshShards + k) * 16]; out->n_surface += mc[(kMeshShards + k) * 16 + 1];
  }
  out->n_dirty = (int64_t)r[0]; out->n_meshes = (int64_t)r[1]; out->n_vertices = (int64_t)r[2];
  out->n_triangles = (int64_t)r[3]; out->roi_pixels = (int64_t)r[4]; out->n_patches = (int64_t)r[5];
  out->n_slots = (int64_t)c.n_slots;
  return TF_OK;
}

int tf_sync(tf_volume* v) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  CtlSnap ctl;
  return fetch_ctl(v, &ctl);
}

// ---- state access -------------------------------------------------------------------
int tf_chunks_download(tf_volume* v, const int32_t* ids, int64_t n, float* sdf, float* weight,
                       uint16_t* color) {
  if (!v || (n > 0 && !ids)) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  const size_t per = 16 + 2048 + 2048 + 4096 + 4;  // id, sdf, weight, colour, found
  int rc = ensure_tmp(v, (size_t)n * per);
  if (rc) return rc;
  rc = ensure_pinned(v, (size_t)n * per);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hb = reinterpret_cast<uint8_t*>(v->h_pinned);
  uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
  int32_t* hid = reinterpret_cast<int32_t*>(hb);
  for (int64_t i = 0; i < n; ++i) {
    hid[4 * i] = ids[3 * i]; hid[4 * i + 1] = ids[3 * i + 1]; hid[4 * i + 2] = ids[3 * i + 2]; hid[4 * i + 3] = 0;
  }
  const size_t o_sdf = (size_t)n * 16, o_w = o_sdf + (size_t)n * 2048, o_c = o_w + (size_t)n * 2048,
               o_f = o_c + (size_t)n * 4096;
  TF_HIP(hipMemcpyAsync(db, hb, (size_t)n * 16, hipMemcpyHostToDevice, v->stream));
  launch_gather_chunks(v->dev, reinterpret_cast<const int4*>(db), (uint32_t)n,
                       reinterpret_cast<float*>(db + o_sdf), reinterpret_cast<float*>(db + o_w),
                       reinterpret_cast<uint16_t*>(db + o_c), reinterpret_cast<uint32_t*>(db + o_f),
                       v->stream);
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(hb + o_sdf, db + o_sdf, (size_t)n * (per - 16), hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  const uint32_t* found = reinterpret_cast<const uint32_t*>(hb + o_f);
  for (int64_t i = 0; i < n; ++i)
    if (!found[i]) {
      set_error("chunk (" + std::to_string(ids[3 * i]) + "," + std::to_string(ids[3 * i + 1]) + "," +
                std::to_string(ids[3 * i + 2]) + ") does not exist");
      return TF_ERR_MISSING_CHUNK;
    }
  if (sdf) memcpy(sdf, hb + o_sdf, (size_t)n * 2048);
  if (weight) memcpy(weight, hb + o_w, (size_t)n * 2048);
  if (color) memcpy(color, hb + o_c, (size_t)n * 4096);
  return TF_OK;
}

int tf_chunk_download(tf_volume* v, const int32_t id[3], float* sdf, float* weight, uint16_t* color) {
  return tf_chunks_download(v, id, 1, sdf, weight, color);
}

int tf_has_chunk(tf_volume* v, const int32_t id[3], int* out) {
  if (!v || !id || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int rc = tf_chunks_download(v, id, 1, nullptr, nullptr, nullptr);
  if (rc == TF_ERR_MISSING_CHUNK) { *out = 0; return TF_OK; }
  if (rc) return rc;
  *out = 1;
  return TF_OK;
}

int tf_chunk_upload(tf_volume* v, const int32_t id[3], const float* sdf, const float* weight,
                    const uint16_t* color) {
  if (!v || !id) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if ((sdf == nullptr) != (weight == nullptr)) { set_error("sdf and weight must be given together"); return TF_ERR_INVALID; }
  int rc = ensure_tmp(v, 8192);
  if (rc) return rc;
  rc = ensure_pinned(v, 8192);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hb = reinterpret_cast<uint8_t*>(v->h_pinned);
  uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
  if (sdf) { memcpy(hb, sdf, 2048); memcpy(hb + 2048, weight, 2048); }
  if (color) memcpy(hb + 4096, color, 4096);
  TF_HIP(hipMemcpyAsync(db, hb, 8192, hipMemcpyHostToDevice, v->stream));
  int4 i4 = make_int4(id[0], id[1], id[2], 0);
  launch_scatter_chunk(v->dev, i4, sdf ? reinterpret_cast<float*>(db) : nullptr,
                       sdf ? reinterpret_cast<float*>(db + 2048) : nullptr,
                       color ? reinterpret_cast<uint16_t*>(db + 4096) : nullptr, v->stream);
  TF_HIP(hipGetLastError());
  v->host_list_n = -1;
  CtlSnap ctl;
  return fetch_ctl(v, &ctl);
}

static int list_common(tf_volume* v, bool dirty, int32_t* out_ids, int64_t cap, int64_t* n) {
  if (!v || !n) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (cap < 0) cap = 0;
  int rc = ensure_tmp(v, (size_t)cap * 16 + 16);
  if (rc) return rc;
  rc = ensure_pinned(v, (size_t)cap * 16 + 16);
  if (rc) return rc;
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
  if (dirty) launch_list_dirty(v->dev, reinterpret_cast<int4*>(v->d_tmp), (uint32_t)cap, v->clear_floor, v->stream);
  else launch_list_chunks(v->dev, reinterpret_cast<int4*>(v->d_tmp), (uint32_t)cap, v->stream);
  TF_HIP(hipGetLastError());
  CtlSnap ctl;
  rc = fetch_ctl(v, &ctl);
  if (rc) return rc;
  *n = ctl.vc.n_tmp;
  const int64_t m = std::min<int64_t>(cap, ctl.vc.n_tmp);
  if (m > 0 && out_ids) {
    TF_HIP(hipMemcpyAsync(v->h_pinned, v->d_tmp, (size_t)m * 16, hipMemcpyDeviceToHost, v->stream));
    TF_HIP(hipStreamSynchronize(v->stream));
    const int32_t* st = reinterpret_cast<const int32_t*>(v->h_pinned);
    for (int64_t i = 0; i < m; ++i) {
      out_ids[3 * i] = st[4 * i]; out_ids[3 * i + 1] = st[4 * i + 1]; out_ids[3 * i + 2] = st[4 * i + 2];
    }
  }
  if (ctl.vc.n_tmp > cap && out_ids) { set_error("output capacity too small"); return TF_ERR_CAPACITY; }
  return TF_OK;
}

int tf_list_chunks(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n) {
  return list_common(v, false, out_ids, cap, n);
}
int tf_list_dirty(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n) {
  return list_common(v, true, out_ids, cap, n);
}

int tf_clear_dirty(tf_volume* v) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  // chunksToUpdate.clear() (Chisel.cpp:146): every mark written so far carries an epoch stamp
  // <= the number of finalizes enqueued; raising the floor to it empties the set without touching HBM
  v->clear_floor = v->epoch;
  return TF_OK;
}

int tf_get_stats(tf_volume* v, tf_stats* out) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  memset(out, 0, sizeof(*out));
  int rc = ensure_tmp(v, 64);
  if (rc) return rc;
  TF_HIP(hipMemsetAsync(v->d_tmp, 0, 32, v->stream));
  launch_rowstats(v->dev, reinterpret_cast<unsigned long long*>(v->d_tmp), v->stream);
  TF_HIP(hipGetLastError());
  unsigned long long r3[4];
  TF_HIP(hipMemcpyAsync(r3, v->d_tmp, 32, hipMemcpyDeviceToHost, v->stream));
  CtlSnap ctl;
  rc = fetch_ctl(v, &ctl);
  if (rc) return rc;
  out->n_coarse = ctl.f.n_coarse;
  out->n_selected = ctl.f.n_list;
  out->n_listed = (int64_t)ctl.f.n_list;
  out->n_updated = (int64_t)r3[2];
  out->rows_tsdf = (int64_t)r3[0];
  out->rows_color = (int64_t)r3[1];
  out->n_slots = 0;
  for (int k = 0; k < kSlotStripes; ++k) out->n_slots += ctl.vc.slot_cnt[k];
  for (int a = 0; a < 3; ++a) { out->min_id[a] = ctl.f.min_id[a]; out->max_id[a] = ctl.f.max_id[a]; }
  int64_t nd = 0, na = 0;
  rc = list_common(v, true, nullptr, 0, &nd);
  if (rc) return rc;
  out->n_dirty = nd;
  rc = list_common(v, false, nullptr, 0, &na);  // alive chunks are counted on demand
  if (rc) return rc;
  out->n_chunks = na;
  return TF_OK;
}

// ---- measurement --------------------------------------------------------------------
int tf_profile_enable(tf_volume* v, uint32_t kind_mask) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  v->prof_mask = kind_mask;
  return TF_OK;
}

int tf_profile_get(tf_volume* v, tf_profile* out, int reset) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  TF_HIP(hipStreamSynchronize(v->stream));
  prof_collect(v);
  *out = v->prof_acc;
  if (reset) memset(&v->prof_acc, 0, sizeof(v->prof_acc));
  return TF_OK;
}

// ---- Chunk::observations on the device + the exports TexMap consumes -----------------
// ids (host, int32[3n]) -> int4 list at the start of d_tmp (which must hold 16 n + extra bytes); returns after the copy
// has been enqueued
static int ids_to_device(tf_volume* v, const int32_t* ids, int64_t n, size_t extra) {
  int rc = ensure_tmp(v, (size_t)n * 16 + extra + 64);
  if (rc) return rc;
  rc = ensure_pinned(v, (size_t)n * 16 + extra + 64);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));  // previous use of the staging buffer
  int32_t* h = reinterpret_cast<int32_t*>(v->h_pinned);
  for (int64_t i = 0; i < n; ++i) { h[4 * i] = ids[3 * i]; h[4 * i + 1] = ids[3 * i + 1]; h[4 * i + 2] = ids[3 * i + 2]; h[4 * i + 3] = 0; }
  TF_HIP(hipMemcpyAsync(v->d_tmp, h, (size_t)n * 16, hipMemcpyHostToDevice, v->stream));
  return TF_OK;
}

int tf_observations_record(tf_volume* v, int32_t keyframe_id) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (keyframe_id < 0) return TF_OK;  // Chisel.h:244: keyframeID >= 0
  launch_obs_record(v->dev, keyframe_id, v->stream);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_observations_retract(tf_volume* v, int32_t keyframe_id, const int32_t* ids, int64_t n) {
  if (!v || (n > 0 && !ids)) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  int rc = ids_to_device(v, ids, n, 0);
  if (rc) return rc;
  launch_obs_retract(v->dev, keyframe_id, reinterpret_cast<const int4*>(v->d_tmp), (uint32_t)n, v->stream);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_export_datacost(tf_volume* v, const int32_t* ids, int64_t n, int32_t frame_index, const int32_t* frames_to_update,
                       int32_t n_frames, float* out) {
  if (!v || (n > 0 && (!ids || !out)) || n_frames < 0 || (n_frames > 0 && !frames_to_update)) { set_error("invalid argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  const size_t cols = (size_t)1 + (size_t)n_frames;
  const size_t o_fr = (size_t)n * 16, o_out = (o_fr + (size_t)n_frames * 4 + 15) & ~(size_t)15;
  int rc = ids_to_device(v, ids, n, (size_t)n_frames * 4 + 16 + (size_t)n * cols * 4);
  if (rc) return rc;
  uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
  uint8_t* hb = reinterpret_cast<uint8_t*>(v->h_pinned);
  if (n_frames) {
    memcpy(hb + o_fr, frames_to_update, (size_t)n_frames * 4);
    TF_HIP(hipMemcpyAsync(db + o_fr, hb + o_fr, (size_t)n_frames * 4, hipMemcpyHostToDevice, v->stream));
  }
  launch_obs_export(v->dev, reinterpret_cast<const int4*>(db), (uint32_t)n, frame_index, reinterpret_cast<const int32_t*>(db + o_fr),
                    n_frames, reinterpret_cast<float*>(db + o_out), v->stream);
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(hb + o_out, db + o_out, (size_t)n * cols * 4, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  memcpy(out, hb + o_out, (size_t)n * cols * 4);
  return TF_OK;
}

int tf_export_adjacency(tf_volume* v, const int32_t* ids, int64_t n, int32_t* out_edges, int64_t cap_edges, int64_t* n_edges) {
  if (!v || !n_edges || (n > 0 && !ids) || cap_edges < 0 || (cap_edges > 0 && !out_edges)) { set_error("invalid argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  *n_edges = 0;
  if (n <= 0) return TF_OK;
  const size_t o_cnt = (size_t)n * 16, o_out = o_cnt + 16;
  int rc = ids_to_device(v, ids, n, 16 + (size_t)cap_edges * 16);
  if (rc) return rc;
  uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
  uint8_t* hb = reinterpret_cast<uint8_t*>(v->h_pinned);
  TF_HIP(hipMemsetAsync(db + o_cnt, 0, 16, v->stream));
  launch_adj_export(v->dev, reinterpret_cast<const int4*>(db), (uint32_t)n, reinterpret_cast<int4*>(db + o_out), (uint32_t)cap_edges,
                    reinterpret_cast<uint32_t*>(db + o_cnt), v->stream);
  TF_HIP(hipGetLastError());
  uint32_t cnt = 0;
  TF_HIP(hipMemcpyAsync(&cnt, db + o_cnt, 4, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  *n_edges = cnt;
  const int64_t m = (int64_t)cnt < cap_edges ? (int64_t)cnt : cap_edges;
  if (m > 0) {
    TF_HIP(hipMemcpyAsync(hb + o_out, db + o_out, (size_t)m * 16, hipMemcpyDeviceToHost, v->stream));
    TF_HIP(hipStreamSynchronize(v->stream));
    memcpy(out_edges, hb + o_out, (size_t)m * 16);
  }
  if ((int64_t)cnt > cap_edges && out_edges) { set_error("output capacity too small"); return TF_ERR_CAPACITY; }
  return TF_OK;
}

int tf_profile_calibrate(tf_volume* v, int32_t n_pairs, double* us_per_pair) {
  if (!v || !us_per_pair || n_pairs <= 0) { set_error("invalid argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  std::vector<hipEvent_t> ev((size_t)2 * n_pairs);
  for (auto& e : ev) TF_HIP(hipEventCreate(&e));
  TF_HIP(hipStreamSynchronize(v->stream));
  for (int i = 0; i < n_pairs; ++i) {
    TF_HIP(hipEventRecord(ev[2 * i], v->stream));
    launch_null(v->stream);
    TF_HIP(hipEventRecord(ev[2 * i + 1], v->stream));
  }
  TF_HIP(hipStreamSynchronize(v->stream));
  double sum = 0.0;
  for (int i = 0; i < n_pairs; ++i) {
    float ms = 0.f;
    TF_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    sum += ms;
  }
  for (auto& e : ev) hipEventDestroy(e);
  *us_per_pair = 1e3 * sum / n_pairs;
  return TF_OK;
}

int tf_debug_phase_raw(tf_volume* v, uint64_t* out, int64_t cap_words) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  size_t n = (size_t)kPhaseWaves * 16;
  if ((size_t)cap_words < n) n = (size_t)cap_words;
  TF_HIP(hipMemcpyAsync(out, v->dev.phase_buf, n * 8, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  return TF_OK;
}

// ---- multi-GPU partition ------------------------------------------------------------
int tf_set_partition_key(tf_volume* v, int32_t a, int32_t b, int32_t c, int32_t key_lo, int32_t key_hi) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (key_lo >= key_hi) { set_error("empty partition"); return TF_ERR_INVALID; }
  if (a < 0 || b < 0 || c < 0 || a > 1 || b > 1 || c > 1 || a + b + c == 0) {
    set_error("partition key coefficients must be 0 or 1, not all 0 (face chunks are found as key == lo / hi - 1)");
    return TF_ERR_INVALID;
  }
  discard_primed(v);  // (the fused selection drops chunks outside the slab)
  v->comm.checked = false;  // the neighbour form of the exchange is re-validated against the new slabs
  v->dev.part_lo = key_lo;
  v->dev.part_hi = key_hi;
  v->dev.part_a = a; v->dev.part_b = b; v->dev.part_c = c;
  return TF_OK;
}

int tf_set_partition(tf_volume* v, int32_t x_lo, int32_t x_hi) { return tf_set_partition_key(v, 1, 0, 0, x_lo, x_hi); }

int tf_boundary_pack(tf_volume* v, void* d_records, int64_t cap_records, int64_t* n) {
  if (!v || !d_records || !n) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
  launch_boundary_pack(v->dev, reinterpret_cast<uint8_t*>(d_records), (uint32_t)cap_records, v->stream);
  v->dev.xl_par ^= 1u;  // (the voxel kernels list what they touch from now on into the other list)
  TF_HIP(hipGetLastError());
  CtlSnap ctl;
  int rc = fetch_ctl(v, &ctl);
  if (rc) return rc;
  *n = ctl.vc.n_tmp;
  if ((int64_t)ctl.vc.n_tmp > cap_records) { set_error("boundary buffer too small"); return TF_ERR_CAPACITY; }
  return TF_OK;
}

int tf_boundary_pack_async(tf_volume* v, void* d_records, int64_t cap_records, uint32_t* d_count) {
  if (!v || !d_records || !d_count) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
  launch_boundary_pack(v->dev, reinterpret_cast<uint8_t*>(d_records), (uint32_t)cap_records, v->stream);
  v->dev.xl_par ^= 1u;  // (the voxel kernels list what they touch from now on into the other list)
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(d_count, &v->dev.vctl->n_tmp, 4, hipMemcpyDeviceToDevice, v->stream));
  return TF_OK;
}

size_t tf_boundary_block_bytes(int64_t cap_records) { return 16 + (size_t)cap_records * TF_BOUNDARY_RECORD_BYTES; }

int tf_boundary_pack_block(tf_volume* v, void* d_block, int64_t cap_records) {
  if (!v || !d_block) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return tf::boundary_pack_block_on(v, d_block, cap_records, v->stream);
}

int tf_boundary_pack_bands2(tf_volume* v, void* d_block_down, int64_t cap_down, void* d_block_up, int64_t cap_up) {
  if (!v || !d_block_down || !d_block_up) { set_error("null argument"); return TF_ERR_INVALID; }
  if (cap_down < 0 || cap_up < 0) { set_error("negative capacity"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return tf::boundary_pack_bands2_on(v, d_block_down, cap_down, d_block_up, cap_up, v->stream);
}

int tf_boundary_pack_bands(tf_volume* v, void* d_block_down, void* d_block_up, int64_t cap_records) {
  return tf_boundary_pack_bands2(v, d_block_down, cap_records, d_block_up, cap_records);
}

int tf_boundary_band_bounds(tf_volume* v, int64_t cap_records, int64_t bounds[4]) {
  if (!v || !bounds) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  // the frame integrated last by a streaming entry point: its selection set is the current one, its epoch v->epoch - 1
  uint32_t cnt[4];
  int rc = xchg_band_counts(v, v->dev.sel.ctl, v->epoch, cnt);
  if (rc) return rc;
  for (int q = 0; q < 4; ++q) bounds[q] = xchg_bucket(cnt[q], cap_records);
  return TF_OK;
}

static int unpack_blocks(tf_volume* v, const void* d_blocks, int32_t n_blocks, int32_t own_block, int64_t cap_records,
                         int join_dirty, const void* d_block_b, int64_t cap_b) {
  VolumeDev d = v->dev;
  int par = -1;
  if (join_dirty) {  // the ghosts belong to the frame integrated last; its texture stage has not run yet
    int rc = fused_arm(v);
    if (rc) return rc;
    par = v->atlas.fused_par;
    d.work_ids = v->atlas.d_work_ids + (size_t)par * d.max_chunks;
    d.work_slot = v->atlas.d_work_slot + (size_t)par * d.max_chunks;
  }
  launch_boundary_unpack_blocks(d, reinterpret_cast<const uint8_t*>(d_blocks), n_blocks, own_block,
                                (uint32_t)cap_records, par, v->epoch, v->stream, reinterpret_cast<const uint8_t*>(d_block_b),
                                (uint32_t)cap_b);
  TF_HIP(hipGetLastError());
  v->host_list_n = -1;
  return TF_OK;
}

int tf_boundary_unpack_blocks(tf_volume* v, const void* d_blocks, int32_t n_blocks, int32_t own_block,
                              int64_t cap_records, int join_dirty) {
  if (!v || !d_blocks) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return unpack_blocks(v, d_blocks, n_blocks, own_block, cap_records, join_dirty, nullptr, 0);
}

int tf_boundary_unpack_pair(tf_volume* v, const void* d_from_below, int64_t cap_below, const void* d_from_above,
                            int64_t cap_above, int join_dirty) {
  if (!v || !d_from_below || !d_from_above) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return unpack_blocks(v, d_from_below, 2, -1, cap_below, join_dirty, d_from_above, cap_above);
}

int tf_boundary_unpack(tf_volume* v, const void* d_records, int64_t n_records) {
  if (!v || (n_records > 0 && !d_records)) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  launch_boundary_unpack(v->dev, reinterpret_cast<const uint8_t*>(d_records), (uint32_t)n_records, v->stream);
  TF_HIP(hipGetLastError());
  v->host_list_n = -1;
  return TF_OK;
}

}  // extern "C"
