// tf_volume.h -- host-side state behind the opaque tf_volume handle.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tf_fusion.h"
#include "tf_copy_pool.h"
#include "tf_device.h"

namespace tf {

void set_error(const std::string& msg);

#define TF_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      ::tf::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
      return TF_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

// Every C-ABI entry point binds the calling thread to the handle's device first: scratch allocations,
// launches and copies must land on the GPU the volume lives on, whatever device the caller's thread
// had current (two volumes on different GPUs in one process, framework worker threads).
#define TF_DEV_NOFLUSH(v)                                                                   \
  do {                                                                                      \
    ++(v)->call_seq;                                                                        \
    hipError_t _e = hipSetDevice((v)->device);                                              \
    if (_e != hipSuccess) {                                                                 \
      ::tf::set_error(std::string("hipSetDevice: ") + hipGetErrorString(_e));               \
      return TF_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)
// Every entry point but tf_integrate_frame_host first brings the frames that entry point has deferred (its
// two-frame launch pipeline, see tf_capi.cpp) onto the stream, so that nothing can observe the deferral.
#define TF_DEV_STREAM(v)                                                                    \
  do {                                                                                      \
    TF_DEV_NOFLUSH(v);                                                                      \
    if ((v)->n_pend) {                                                                      \
      int _rc = ::tf::flush_deferred(v);                                                    \
      if (_rc) return _rc;                                                                  \
    }                                                                                       \
  } while (0)
// ... and, except for the streaming entry points (whose next launch carries it), the patch stage of the last
// textured frame (AtlasState::pend_patch).
#define TF_DEV(v)                                                                           \
  do {                                                                                      \
    TF_DEV_STREAM(v);                                                                       \
    if ((v)->atlas.pend_patch.on) {                                                         \
      int _rc = ::tf::patch_flush(v);                                                       \
      if (_rc) return _rc;                                                                  \
    }                                                                                       \
  } while (0)

struct ProfEvent {
  hipEvent_t a, b;
  int kind;
};

struct KeyframeSlot {
  uint8_t* rgb = nullptr;   // device, u8[H][W][stride]
  float* depth = nullptr;   // device, f32[H][W]
  bool owned = false;
  int slot = -1;            // entry of the device keyframe table
};

struct AtlasState {
  int32_t aw = 13824, ah = 13824;
  uint64_t pw = 0, ph = 0;
  uint8_t* buf = nullptr;  // device, u8[ah][aw][3]
  // keyframe table (device copy + host mirror)
  std::unordered_map<int32_t, KeyframeSlot> keyframes;
  std::vector<KfDev> h_kf;
  std::vector<uint8_t> kf_used;
  KfDev* d_kf = nullptr;
  int kf_cap = 0;
  AtlasCtl* d_actl = nullptr;
  int4* d_work_ids = nullptr;
  uint32_t* d_work_slot = nullptr;
  int4* d_wl_ids = nullptr;        // VolumeDev::wl_*: the dirty set K-A builds (shard lists, two parities)
  uint32_t* d_wl_slot = nullptr;
  uint32_t* d_wl_cnt = nullptr;
  int4* d_patch_list = nullptr;
  uint32_t* d_patch_cnt = nullptr;
  uint32_t* h_dirty_len = nullptr;  // pinned, device-visible: the mesher's filter leaves the length of a frame's dirty list here
  unsigned long long* d_cand = nullptr;
  int fused_par = 0;   // counter set of the next fused frame
  // fused flow: the patch stage of textured frame f (slot hand-out, CompressMeshes' exchange, CalculateTexCoords,
  // UpdateBuffer) is not launched behind the frame's mesher but rides on the NEXT frame's launch, next to its voxel
  // update (k_frame, tf_kernels.hip) -- it reads meshes, the frame's own images and the atlas, nothing K-A touches.
  // Anything that could observe the deferral flushes it first (TF_DEV -> patch_flush: the stage as a launch of its own).
  struct PendPatch {
    bool on = false;
    PatchStage st;
    int host_slot = -1;  // tf_integrate_frame_host: the staging slot whose device images the stage still reads
  } pend_patch;
  bool fused_armed = false;  // the counter sets are in the state the fused flow expects
  // tf_texture_frame_device_phase: phase 1 (dirty set + interior meshes) of the stage of frame `phase1_epoch` has run,
  // phase 2 (boundary meshes behind the caller's unpack, pending patch stage) has not
  bool phase1_on = false;
  uint32_t phase1_epoch = 0;
  // staging
  void* d_stage = nullptr;
  size_t d_stage_bytes = 0;
  void* h_stage = nullptr;
  size_t h_stage_bytes = 0;
};

// RCCL communicator of the handle (tf_comm_init) and the exchange buffers
struct CommState {
  void* comm = nullptr;  // ncclComm_t
  int rank = 0, nranks = 0;
  void* d_send = nullptr;
  void* d_recv = nullptr;
  int64_t cap_records = 0;
  int mode = TF_XCHG_NEIGHBOURS;
  uint64_t exchanges = 0, bytes_received = 0, bytes_sent = 0;  // tf_comm_stats / tf_comm_stats_ex
  uint64_t bound_records = 0;  // sum of the record capacities the sized exchanges were given (sent sides)
  bool neighbours_ok = false;  // tf_comm_check_partition found every slab wide enough and rank-ordered
  bool checked = false;
};
// the sized exchange's record capacity for a band with c selected chunks: multiples of 8 records, at least 8 (room
// for what an earlier, overflowing exchange left flagged), never more than the caller's cap
inline uint32_t xchg_bucket(uint32_t c, int64_t cap) {
  uint64_t b = ((uint64_t)c + 7u) & ~7ull;
  if (b < 8) b = 8;
  if (cap > 0 && b > (uint64_t)cap) b = (uint64_t)cap;
  return (uint32_t)b;
}

}  // namespace tf

struct tf_volume {
  static constexpr int kSelSets = 4;  // ring of selection sets: K-B / K-C run 2 / 1 frames ahead of K-A
  tf_config cfg;
  int device = 0;
  float res = 0.005f;
  int use_color = 1;
  // camera as given (floats) and as consumed (int-truncated)
  float fx = 525.f, fy = 525.f, cx = 319.5f, cy = 239.5f;
  tf::Cam cam;
  tf::Integ ig;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  tf::VolumeDev dev;
  tf::SelBuf selbuf[kSelSets];  // ring of selection scratch sets (dev.sel = the active set)
  int cur_sel = 0;
  // frames whose selection stages already ran at the end of the previous streaming call (n_ahead)
  struct Primed { const float* depth; float pose[12]; };
  Primed primed[2];
  int n_primed = 0;
  std::vector<void*> allocs;
  // frame images
  float* d_depth = nullptr;      // owned staging targets
  uint8_t* d_rgba = nullptr;
  float* d_quality = nullptr;
  size_t img_pixels = 0;
  // drop-in per-frame host path (tf_integrate_frame_host): ring of pinned staging + device image slots, H2D on
  // its own stream so that the copy of frame f+1 overlaps the kernels of frame f
  static constexpr int kHostRing = 8;  // four deferred frames + the one being staged + three whose kernels may still run
                                       // (a frame's images are read by its patch stage one launch behind its voxel update)
  struct HostSlot {
    uint8_t* h = nullptr;      // pinned: depth f32[npix] | rgba u8[4 npix]
    uint8_t* d = nullptr;      // device: same layout
    hipEvent_t copied = nullptr;
    uint32_t free_when = 0;    // 0: free; else the progress stamp (h_progress) at which the last launch that reads d is through
  };
  // Launch progress without stream events: every frame launch writes its sequence number into this pinned word when it
  // STARTS (= every launch ahead of it on the stream is through).  An event record between two launches of a frame cost
  // the host-frames path 6.8 us per frame of idle device time (profiles/r3, run 29).
  uint32_t* h_progress = nullptr;
  uint32_t progress_seq = 0;   // stamp of the last frame launch put on the stream
  HostSlot hslot[kHostRing];
  // caller buffers registered with tf_host_register (page-locked in place): host frames that lie inside one are uploaded
  // straight out of it -- no staging copy -- and the call returns when that upload is through
  struct HostRange { const uint8_t* p; size_t n; const uint8_t* locked; };  // locked: base of the process-wide page-locked range that covers it
  std::vector<HostRange> host_ranges;
  size_t hslot_pixels = 0;
  int hslot_next = 0;
  hipStream_t copy_stream = nullptr;
  hipStream_t copy_stream2 = nullptr;  // registered caller buffers: the colour image goes up next to the depth image (a second copy queue)
  hipEvent_t copy_join = nullptr;
  long host_waits = 0;  // copies a launch had to wait for in the stream (TF_HOST_TRACE prints it)
  double host_trace[6] = {0, 0, 0, 0, 0, 0};  // TF_HOST_TRACE=1: microseconds per phase of tf_integrate_frame_host, [5] = calls
  tf::CopyPool* copy_pool = nullptr;  // helper threads of the staging copy (TF_COPY_THREADS, default 3)
  void* h_pinned = nullptr;      // pinned host staging (uploads / downloads)
  size_t h_pinned_bytes = 0;
  tf::FrameImages frame{nullptr, nullptr, nullptr};
  bool frame_bound = false;
  // host shadow of the device-resident visible list (int32[3*n]); -1 = device list unknown
  std::vector<int32_t> host_list;
  int64_t host_list_n = -1;
  // what the device list's needsUpdate / isNew flags hold, as far as the call-by-call entry points know (valid while
  // host_flags_n == host_list_n): a caller that hands back the flags it was given -- the reference's loop over one
  // keyframe's frames does -- costs no upload
  std::vector<uint8_t> host_needs, host_new;
  int64_t host_flags_n = -2;
  // entry points counted (TF_DEV*): tf_compress_meshes reuses the dirty list tf_update_meshes left in d_tmp when that was
  // the call right before it (MobileFusion.cpp:327-345 calls them back to back; nothing in between can have marked a chunk)
  uint64_t call_seq = 0, dirty_list_seq = ~0ull;
  uint32_t dirty_list_n = 0;
  uint32_t* h_ctl = nullptr;  // pinned: FrameCtl head + VolCtl as fetch_ctl reads them
  uint32_t epoch = 0;        // finalize counter (mark / erase stamps are epoch + 1)
  uint32_t clear_floor = 0;  // stamps <= this were cleared (Chisel::CompressMeshes' chunksToUpdate.clear())
  uint32_t mesh_epoch = 0;   // meshing passes so far (MeshRec::epoch)
  int mesh_par = 0;          // parity of the next mesher launch (VolumeDev::mesh_cnt)
  // frames tf_integrate_frame_host has staged but not integrated yet (it runs three frames behind: K-A of frame f - 3
  // shares its launch with the selection stages of f - 2 and f - 1, like the streaming entry points; frame f itself is
  // only being copied, so that no launch ever has to wait for a copy in the stream)
  struct Pending {
    const float* d = nullptr;
    const uint8_t* c = nullptr;
    float pose[12];
    float pinv[16];
    bool tex = false;
    int32_t fid = 0;
    int slot = 0;
    bool copied = false;  // its H2D copy is known to be complete, or the handle's stream has been told to wait for it
  };
  static constexpr int kHostDefer = 4;  // frames tf_integrate_frame_host runs behind its caller (a launch reads the oldest three)
  Pending pend[kHostDefer];
  int n_pend = 0;
  bool host_defer = true;  // tf_integrate_frame_host runs kHostDefer frames behind its caller (tf_host_frame_set_deferral)
  bool host_async = false;           // tf_host_frame_set_async: a call out of registered buffers returns before its upload is through
  hipEvent_t last_upload = nullptr;  // the newest frame's upload (tf_host_frame_fence waits for it)
  float* d_group = nullptr;  // staging of tf_integrate_depth_group_host: six depth images
  size_t d_group_pixels = 0;
  // on-demand device scratch
  void* d_tmp = nullptr;
  size_t d_tmp_bytes = 0;
  // profiling
  bool prof_open = false;
  uint32_t prof_mask = 0;  // bit k = time kernels of kind TF_PROF_k
  std::vector<tf::ProfEvent> prof_events;
  std::vector<hipEvent_t> prof_pool;
  tf_profile prof_acc{};
  tf::AtlasState atlas;
  tf::CommState comm;
  int64_t comm_cap = 0;  // > 0: the fused textured flow exchanges the ghost band after every voxel update
  // band counts of a frame's selection as the host sees them: pinned words [0] tag (frame epoch + 1), [1..4] FrameCtl::band_cnt
  uint32_t* h_xchg = nullptr;
  uint32_t xchg_pub_enq = 0;  // frame tag of the publish that is already on the stream (0: none)
  uint32_t xchg_pub_seq = 0;  // ... the sequence number that publish writes into h_xchg[0] (what the host waits for)
  // the per-frame exchange overlapped with the interior meshes (texture_stage): second stream, fork / join events
  hipStream_t xstream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // tf_atlas_snapshot_rows: a second thread (the reference's GUI thread reads Atlas::texture_buffer while the map thread
  // writes it, GCFusion/MobileFusion.h:404-421) copies atlas rows out consistently.  atlas_mu orders that thread's
  // in-stream snapshot against the map thread's atlas-writing launches; atlas_seq counts those launches, atlas_frame is
  // the label of the newest one; snap_mu serialises readers (one snapshot buffer, one read stream).
  std::mutex atlas_mu, snap_mu;
  std::atomic<uint64_t> atlas_seq{0};
  std::atomic<int32_t> atlas_frame{-1};
  hipStream_t read_stream = nullptr;
  hipEvent_t read_ev = nullptr;
  uint8_t* d_snap = nullptr;
  size_t d_snap_bytes = 0;
  bool xchg_overlap = true;        // tf_comm_exchange_overlap
  uint64_t xchg_overlapped = 0;    // exchanges that ran next to an interior mesh pass (tf_comm_stats_ex)
  uint32_t xchg_seq = 0;      // publish sequence numbers handed out (monotonic over the handle's life: a stale word never matches)
};

namespace tf {
// Around the enqueue of a launch that writes atlas texels: nothing of tf_atlas_snapshot_rows can land between the launch
// and the bump of the write sequence (label: Patch::frameid of what it writes, INT32_MIN = keep the last one).
struct AtlasWriteScope {
  tf_volume* v;
  int32_t label;
  AtlasWriteScope(tf_volume* vv, int32_t l) : v(vv), label(l) { v->atlas_mu.lock(); }
  ~AtlasWriteScope() {
    if (label != INT32_MIN) v->atlas_frame.store(label, std::memory_order_relaxed);
    v->atlas_seq.fetch_add(1, std::memory_order_release);
    v->atlas_mu.unlock();
  }
  AtlasWriteScope(const AtlasWriteScope&) = delete;
  AtlasWriteScope& operator=(const AtlasWriteScope&) = delete;
};
// stream synchronisation + control blocks; VolCtl::n_tmp as read (may be NULL); sticky status -> error code
int sync_status(tf_volume* v, uint32_t* n_tmp);
int ensure_tmp(tf_volume* v, size_t bytes);
int launch_prepare(tf_volume* v, const Pose& pose, bool with_acquire, hipStream_t s = nullptr);  // tf_capi.cpp
struct KfStoreArgs;  // tf_kf_store.h
// ride_filter: a patch stage still pending when the stage starts rides on its filter launch (the keyframe unit: there is
// no k_frame launch for it to ride on) instead of going out as a launch of its own
int texture_stage(tf_volume* v, const SelBuf& sel, const FrameImages& img, uint32_t frame_epoch, const float* pose_inv16,
                  int32_t frame_id, bool claimed = false, const FrameCtl* next_ctl = nullptr, bool ride_filter = false,
                  bool sized_xchg = false, int phase = 0,   // sized_xchg: sel.ctl holds the frame's band counts (fused stream only)
                  const KfStoreArgs* store = nullptr);      // the keyframe unit: the list's validChunks store rides on the filter launch
int texture_stage_finish(tf_volume* v, const FrameImages& img, uint32_t frame_epoch, const float* pose_inv16, int32_t frame_id, int par);
// the four band counts of the frame whose selection wrote `ctl` (tag = its epoch + 1): waits for the device to publish them
int xchg_band_counts(tf_volume* v, const FrameCtl* ctl, uint32_t tag, uint32_t cnt[4], hipStream_t s = nullptr);
uint32_t nbr_next_seq(tf_volume* v);  // neighbour table: the seq of the filter launch about to go out (tf_capi.cpp)
int flush_deferred(tf_volume* v);
void launch_dirty_frame_store(const VolumeDev& v, int par, uint32_t stamp, const KfStoreArgs& a, hipStream_t s);  // tf_mesh.hip
int patch_flush(tf_volume* v);
bool host_defer_default();  // !(TF_HOST_DEFER=0 in the environment)
int fused_arm(tf_volume* v);  // the fused flow's counter sets in their start state (no-op once armed)
int ensure_pinned(tf_volume* v, size_t bytes);
void prof_begin(tf_volume* v, int kind, hipStream_t s = nullptr);
void prof_end(tf_volume* v, hipStream_t s = nullptr);
int atlas_init(tf_volume* v);
void atlas_destroy(tf_volume* v);
int atlas_reset(tf_volume* v);
int comm_exchange(tf_volume* v, int64_t cap_records, int dirty_par, uint32_t stamp, const FrameCtl* ctl = nullptr,
                  uint32_t tag = 0, const FrameCtl* next_ctl = nullptr, hipStream_t xs = nullptr);  // xs: the stream it runs on (null: the handle's)
// the pack launches of tf_boundary_pack_block / tf_boundary_pack_bands2 on a given stream (no deferred-frame flush: the caller did it)
int boundary_pack_block_on(tf_volume* v, void* d_block, int64_t cap_records, hipStream_t s);
int boundary_pack_bands2_on(tf_volume* v, void* d_block_down, int64_t cap_down, void* d_block_up, int64_t cap_up, hipStream_t s);
void comm_destroy(tf_volume* v);
int kf_push(tf_volume* v, int slot);
void launch_patch_fused(tf_volume* v, const VolumeDev& d, int par, const KfDev& kf, hipStream_t s);
inline uint64_t host_pack_id(const int32_t id[3]) {
  return ((uint64_t)((uint32_t)(id[0] + (1 << 20)) & 0x1FFFFFu) << 42) |
         ((uint64_t)((uint32_t)(id[1] + (1 << 20)) & 0x1FFFFFu) << 21) |
         (uint64_t)((uint32_t)(id[2] + (1 << 20)) & 0x1FFFFFu);
}
}  // namespace tf
