// tf_unit.hip -- the keyframe unit of MobileFusion::tsdfFusion (GCFusion/MobileFusion.cpp:274-406) as ONE asynchronous
// call: for every keyframe whose pose moved, RetractObservations (:252-272) + ReIntegrateKeyframe(flag 0) over the
// keyframe's stored validChunks at the OLD poses + ReIntegrateKeyframe(flag 1) at the NEW ones (:114-221: the
// keyframe's depth + colour + quality, then its <= 6 local frames depth-only over the same list); then the new keyframe
// group; then UpdateMeshes -> chunksToUpdate -> CompressMeshes (:327-355) and, optionally, GeneratePatches with the new
// keyframe as every chunk's label + UpdateAtlas (:374-382).  Everything stays on the handle's stream: the visible
// lists, the needsUpdate / new flags, Frame::validChunks of every keyframe and Chunk::observations live in HBM, nothing
// is copied back and the host never waits.  (The view selection of :357-366 is host code outside the path; a caller that
// runs it stops the unit behind CompressMeshes -- texture = 0 -- and continues with tf_generate_patches and its labels.)
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "tf_devfn.h"
#include "tf_kf_store.h"
#include "tf_volume.h"

namespace tf {

// Frame::validChunks of the keyframes (GCFusion/frame.h), device-resident: an arena of chunk ids and, per keyframe slot,
// its region (start, capacity) and the length of its list.  A keyframe that is integrated again writes into its region
// when the new list fits (regions are handed out with a quarter of slack), else it gets a new region at the top; when
// the top reaches the end of the arena the live regions are moved together first (k_kf_store, one workgroup, no host
// involvement).
struct UnitState {
  int4* arena = nullptr;
  uint32_t cap = 0;
  bool fixed_arena = false;  // TF_UNIT_ARENA: a test pins the arena (compaction is then the only way to make room)
  KfTab* tab = nullptr;
  uint32_t slots = 0;
  uint32_t* h_fill = nullptr;  // pinned: [0] top, [1] longest list, [2] live words -- as of the last store the device finished
  uint64_t grows = 0;
  std::unordered_map<int32_t, int> slot_of;
  float4* group_pre = nullptr;  // scratch of the keyframe-group kernel (list records + centroid tables of six frames)
  float* group_cen = nullptr;
  // The front end of a fresh group (k_bbox -> k_select -> k_pre_group: a pure function of the group's images and poses and of
  // which chunks exist / are parked) runs on a stream of its own behind the LAST launch that changes chunks -- ev_mut, recorded
  // at the end of every group -- and into the next selection set of the ring: when unit calls follow each other it overlaps
  // the previous call's filter || patch stage and mesher (60 us that read voxels, summaries and meshes, and whose kf_store
  // reads the previous set), instead of 26 us of three small dependent launches between two keyframes.
  hipStream_t front = nullptr;
  hipEvent_t ev_mut = nullptr, ev_front = nullptr;
  uint64_t mut_seq = 0;  // tf_volume::call_seq of the call that recorded ev_mut
};
static std::unordered_map<tf_volume*, UnitState> g_units;  // (one per handle; freed by tf_keyframe_unit_release)
static std::mutex g_units_mu;                              // (handles may live on different threads)

__global__ __launch_bounds__(1024) void k_kf_store(VolumeDev v, KfStoreArgs a) {
  kf_store_body(v, a.tab, a.slots, a.arena, a.cap, a.slot, a.slack, a.fill);
}

// localChunksIntersecting = kf.validChunks; needsUpdate = true, newChunk = false for every entry (MobileFusion.cpp:135-143);
// slots resolved like tf_integrate does for a caller's list (a missing chunk is an error: chunks.at() throws)
// ... and MobileFusion::RetractObservations' chunk side over the same entries: observations.erase(frame_id)
__global__ __launch_bounds__(256) void k_kf_load(VolumeDev v, const KfTab* tab, uint32_t slots, const int4* arena, int slot,
                                                 int32_t kf_id) {
  const SelBuf& L = v.sel;
  uint32_t n = kf_off(tab)[slots + slot];
  if (n > v.max_list) { n = 0; if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&v.vctl->status, kStListFull); }
  const uint32_t off = kf_off(tab)[slot];
  if (blockIdx.x == 0 && threadIdx.x == 0) { L.ctl->n_list = n; L.ctl->n_front = n; }
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int4 id = arena[off + e];
    L.list_id[e] = id;
    L.list_needs[e] = 1;
    L.list_new[e] = 0;
    L.list_quality[e] = 0.0f;
    uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
    uint32_t s = kInvalidSlot;
    if (ent != kInvalidSlot && v.hent[ent].alive) s = v.hent[ent].slot;
    if (s == kInvalidSlot) atomicOr(&v.vctl->status, kStMissing);
    L.list_slot[e] = s;
    L.list_ent[e] = ent == kInvalidSlot ? 0u : ent;
    if (s == kInvalidSlot) continue;
    const unsigned long long key = ((unsigned long long)s << 32) | (unsigned long long)(uint32_t)kf_id;
    uint32_t i = hash_key(key) & v.obs_mask;
    for (uint32_t probe = 0; probe <= v.obs_mask; ++probe) {
      const unsigned long long cur = v.obs_key[i];
      if (cur == key) { v.obs_q[i] = 0.0f; break; }
      if (cur == kEmptyKey) break;
      i = (i + 1) & v.obs_mask;
    }
  }
}

static int unit_state(tf_volume* v, UnitState** out) {
  std::unique_lock<std::mutex> lock(g_units_mu);
  UnitState& u = g_units[v];  // (references into an unordered_map stay valid when other handles are added)
  lock.unlock();
  if (!u.arena) {
    u.cap = (uint32_t)std::min<size_t>((size_t)v->dev.max_list * 16, (size_t)1 << 26);
    if (const char* e = getenv("TF_UNIT_ARENA")) { u.cap = (uint32_t)std::max(1024, atoi(e)); u.fixed_arena = true; }  // test knob: a small, pinned arena
    u.slots = kUnitSlots0;
    if (const char* e = getenv("TF_UNIT_SLOTS")) u.slots = (uint32_t)std::max(4, atoi(e));  // test knob: initial slots
    TF_HIP(hipMalloc((void**)&u.arena, sizeof(int4) * (size_t)u.cap));
    TF_HIP(hipMalloc((void**)&u.tab, kf_tab_bytes(u.slots)));
    TF_HIP(hipMemsetAsync(u.tab, 0, kf_tab_bytes(u.slots), v->stream));
    TF_HIP(hipHostMalloc((void**)&u.h_fill, 64, hipHostMallocDefault));
    memset(u.h_fill, 0, 64);
    TF_HIP(hipMalloc((void**)&u.group_pre, sizeof(float4) * (size_t)kGroupFrames * 4 * v->dev.max_list));
    TF_HIP(hipMalloc((void**)&u.group_cen, sizeof(float) * (size_t)kGroupFrames * 3 * kChunkVoxels));
    if (!(getenv("TF_UNIT_SERIAL_FRONT") && atoi(getenv("TF_UNIT_SERIAL_FRONT")))) {  // (A/B knob: everything on the handle's stream)
      TF_HIP(hipStreamCreateWithFlags(&u.front, hipStreamNonBlocking));
      TF_HIP(hipEventCreateWithFlags(&u.ev_mut, hipEventDisableTiming | hipEventDisableSystemFence));  // (both sides are this GPU: no system-scope writeback / invalidate at the edge)
      TF_HIP(hipEventCreateWithFlags(&u.ev_front, hipEventDisableTiming | hipEventDisableSystemFence));
    }
  }
  *out = &u;
  return TF_OK;
}

// May the front end of group g leave the handle's stream in this call?  Only when
//  - the previous entry point called on the handle was a unit call (ev_mut then stands for "every launch that changes chunks is
//    done" while that call's texture stage is still on the stream; anything else -> everything in stream order),
//  - the group has local frames (its acquire is the lazy one: it inserts keys but leaves parked chunks parked -- a chunk the
//    eager form revives is not the same as a parked one to the previous call's filter, a freshly inserted one is
//    indistinguishable from an absent one to every reader),
//  - the handle runs on a stream of its own (on a stream of the CALLER's, tf_set_stream, stream order towards whatever the
//    caller put there itself -- the producers of the group's images, say -- is kept),
//  - no selection made ahead by a streaming call sits in the ring of selection sets.
static bool can_fork(const tf_volume* v, const UnitState* u, const tf_unit_group* g) {
  return u->front && v->own_stream && v->n_primed == 0 && g->n_local > 0 && u->mut_seq + 1 == v->call_seq;
}
// The image part of a fresh group's front end -- bounding box and selection: a function of the keyframe's depth image and
// pose only -- on the unit's stream, into the NEXT selection set, while the handle's stream works through the call's moved
// keyframes (18 us of two small dependent launches off the critical path of such a call; the set's last users are at least
// two calls back, i.e. ahead of ev_mut).  integrate_group(front_mode = 2) joins.
static int front_ahead(tf_volume* v, UnitState* u, const tf_unit_group* g) {
  VolumeDev dn = v->dev;
  dn.sel = v->selbuf[(v->cur_sel + 1) % tf_volume::kSelSets];
  Pose P;
  memcpy(P.p, g->keyframe.pose, sizeof(P.p));
  TF_HIP(hipStreamWaitEvent(u->front, u->ev_mut, 0));
  launch_bbox(dn, g->keyframe.d_depth, v->cam, P, u->front);
  launch_select(dn, g->keyframe.d_depth, v->cam, v->ig, P, v->res, /*emit=*/true, u->front, /*plain=*/true);
  TF_HIP(hipGetLastError());
  TF_HIP(hipEventRecord(u->ev_front, u->front));
  return TF_OK;
}

// ReIntegrateKeyframe (MobileFusion.cpp:114-221) for one group with one flag
// dirty_par >= 0: the group's updated chunks and their face neighbours join the dirty set of that parity, stamped
// dirty_stamp, as they are finalized (the group kernel's claims into the shard lists; launch_dirty_frame into the flat list
// for a group without local frames) -- instead of a scan of every chunk's mark later
// ride_store != nullptr: the group's validChunks are not stored here -- *ride_store receives the arguments and the caller's
// texture stage takes them along on its filter launch
// front_mode (flag 1 only) -- 0: the front end on the handle's stream; 1: all of it on the unit's stream, now, behind ev_mut
// (the first pass of a call that follows another unit call); 2: its image part (bounding box, selection) is running on the
// unit's stream already, into the NEXT selection set (front_ahead below: the fresh group of a call with moved keyframes) --
// this pass switches to that set, joins, and runs the records / acquire launch on the handle's stream.
// last: the pass is the last one of its tf_keyframe_unit_device call (ev_mut is recorded behind it)
static int integrate_group(tf_volume* v, UnitState* u, const tf_unit_group* g, int flag, int kf_slot, int dirty_par = -1,
                           uint32_t dirty_stamp = 0, KfStoreArgs* ride_store = nullptr, int front_mode = 0, bool last = false) {
  hipStream_t s = v->stream;
  VolumeDev& d = v->dev;
  FrameImages img{g->keyframe.d_depth, reinterpret_cast<const uchar4*>(g->keyframe.d_rgba), g->keyframe.d_quality};
  const float* kpose = flag ? g->keyframe.pose : g->old_keyframe_pose;
  Pose P;
  memcpy(P.p, kpose, sizeof(P.p));
  hipStream_t fs = s;  // the stream of the front end
  // Worth it only for the FIRST pass of a call (a later pass of the same call has nothing of another call to overlap, and
  // the two event edges cost 3-4 us each, profiles/r6/README.md) -- or, its image part, for the fresh group behind moved ones.
  // (the conditions are can_fork()'s, checked by the caller)
  if (flag && front_mode) {
    // (no selection made ahead by a streaming call sits in the ring: the next set is free)
    v->cur_sel = (v->cur_sel + 1) % tf_volume::kSelSets;
    d.sel = v->selbuf[v->cur_sel];
    if (front_mode == 1) {
      TF_HIP(hipStreamWaitEvent(u->front, u->ev_mut, 0));
      fs = u->front;
    } else {
      TF_HIP(hipStreamWaitEvent(s, u->ev_front, 0));  // front_ahead()'s selection is in this set
    }
  }
  if (flag && front_mode == 2) {
    v->frame = img;
    v->frame_bound = true;
  } else if (flag) {
    // PrepareIntersectChunks at the keyframe's pose, without its list ORDER: validChunks = the finalized list in list order,
    // and no order of it is observable through this entry point -- the selection appends straight to a plain list, no scan /
    // write-out launch (k_scan: 17 us per keyframe); the slots, isNew and needsUpdate = false come with the records launch below
    v->frame = img;
    v->frame_bound = true;
    launch_bbox(d, img.depth, v->cam, P, fs);
    launch_select(d, img.depth, v->cam, v->ig, P, v->res, /*emit=*/true, fs, /*plain=*/true);
  } else {
    hipLaunchKernelGGL(k_kf_load, dim3(256), dim3(256), 0, s, d, u->tab, u->slots, u->arena, kf_slot, g->kf_id);
  }
  // the per-chunk records and centroid tables of all the group's frames in ONE launch (k_pre + k_pre_group were two)
  const float* dd[kGroupFrames];
  float poses[12 * kGroupFrames];
  for (int f = 0; f < g->n_local; ++f) {
    dd[f] = g->local[f].d_depth;
    memcpy(poses + 12 * f, flag ? g->local[f].pose : g->old_local_pose[f], 48);
  }
  launch_pre_frames(d, P, g->n_local, poses, u->group_pre, u->group_cen, v->ig, v->res, v->cam, fs,
                    /*acquire=*/flag ? (g->n_local > 0 ? 2 : 1) : 0,  // (lazily when the group kernel finalizes the list)
                    /*clear_word=*/flag ? nullptr : kf_off(u->tab) + u->slots + kf_slot);  // kf.validChunks.clear() (:217): the list is loaded
  if (fs != s) {  // join: the group's passes wait for its front end
    TF_HIP(hipEventRecord(u->ev_front, fs));
    TF_HIP(hipStreamWaitEvent(s, u->ev_front, 0));
  }
  // the keyframe's own depth + colour (+ quality) ...  With local frames behind it and no quality image the pass is the
  // first frame of the group kernel's visit (k_integrate_group<., KEY>), not a launch of its own.
  const bool color = img.rgba != nullptr, quality = color && img.quality != nullptr;
  const bool key_in_group = g->n_local > 0 && color;  // (round 6: with a quality image too -- k_integrate_group<., KEY, QUAL>)
  if (!key_in_group) launch_integrate(d, img, v->cam, v->ig, P, v->res, flag, color, quality, 0, s, true);
  // chunk->observations[keyframeID] for BOTH flags (Chisel.h:244-247): recorded by the group kernel's waves ahead of their
  // own work, by a launch of its own when the group has no local frame
  // ... then its local frames depth-only over the same list, one visit per chunk
  // FinalizeIntegrateChunks + GarbageCollect of the list, and (dirty_par >= 0) the dirty-set pass over it: the group
  // kernel's waves, each behind its entry's last frame (k_finalize + k_dirty_frame were two launches, 7 + 9 us)
  const bool folded = g->n_local > 0;
  if (folded) {
    launch_integrate_group(d, g->n_local, dd, poses, u->group_pre, u->group_cen, v->cam, v->ig, v->res, flag, s, true, g->kf_id,
                           /*fin=*/1, v->epoch++, dirty_par, dirty_stamp, key_in_group ? &img : nullptr, key_in_group ? &P : nullptr);
  } else {
    launch_obs_record(d, g->kf_id, s);
    launch_finalize(d, v->epoch++, s);
  }
  const int slack = !(getenv("TF_UNIT_NO_SLACK") && atoi(getenv("TF_UNIT_NO_SLACK")));  // test knob (read per call): exact-fit regions
  const KfStoreArgs sa{u->tab, u->slots, u->arena, u->cap, kf_slot, slack, u->h_fill};
  bool stored = false;
  if (dirty_par >= 0 && !folded) {
    VolumeDev dd = d;
    dd.work_ids = v->atlas.d_work_ids + (size_t)dirty_par * d.max_chunks;
    dd.work_slot = v->atlas.d_work_slot + (size_t)dirty_par * d.max_chunks;
    // (the group's validChunks are stored by one more workgroup of the same launch: k_kf_store alone is 11 us of launch floor)
    if (flag) { launch_dirty_frame_store(dd, dirty_par, dirty_stamp, sa, s); stored = true; }
    else launch_dirty_frame(dd, dirty_par, dirty_stamp, s);
  }
  if (flag && !stored) {
    if (ride_store) *ride_store = sa;  // (with the texture stage's filter launch)
    else hipLaunchKernelGGL(k_kf_store, dim3(1), dim3(1024), 0, s, d, sa);
  }
  TF_HIP(hipGetLastError());
  if (u->ev_mut && last) {  // the call's last launch that changes chunks is on the stream
    TF_HIP(hipEventRecord(u->ev_mut, s));
    u->mut_seq = v->call_seq;
  }
  v->host_list_n = -1;
  return TF_OK;
}

// One more keyframe than the table has slots: a table of twice the size (stream drained once per doubling).
static int grow_slots(tf_volume* v, UnitState* u) {
  TF_HIP(hipStreamSynchronize(v->stream));
  const uint32_t ns = u->slots * 2u;
  KfTab* nt = nullptr;
  TF_HIP(hipMalloc((void**)&nt, kf_tab_bytes(ns)));
  TF_HIP(hipMemset(nt, 0, kf_tab_bytes(ns)));
  TF_HIP(hipMemcpy(nt, u->tab, sizeof(KfTab), hipMemcpyDeviceToDevice));
  for (int a = 0; a < 3; ++a)  // off | n | capn (the order scratch carries nothing)
    TF_HIP(hipMemcpy(kf_off(nt) + (size_t)a * ns, kf_off(u->tab) + (size_t)a * u->slots, sizeof(uint32_t) * u->slots, hipMemcpyDeviceToDevice));
  TF_HIP(hipDeviceSynchronize());
  hipFree(u->tab);
  u->tab = nt;
  u->slots = ns;
  u->grows += 1;
  return TF_OK;
}
// Enough arena for the lists this call stores and the ones still in flight?  Decided on what the device last reported
// (h_fill, a few calls old at most -- hence the margin of eight further lists); an arena that runs full anyway compacts
// itself and, failing that, reports TF_ERR_CAPACITY as before.
static int grow_arena_if_needed(tf_volume* v, UnitState* u, int stores) {
  if (u->fixed_arena) return TF_OK;
  const uint64_t top = __atomic_load_n(&u->h_fill[0], __ATOMIC_ACQUIRE), longest = u->h_fill[1], live = u->h_fill[2];
  (void)top;
  const uint64_t per = std::max<uint64_t>(longest, 1024) * 5 / 4 + 64;
  const uint64_t margin = (uint64_t)(stores + 8) * per;
  if (live + margin <= (uint64_t)u->cap * 3 / 4) return TF_OK;
  uint64_t ncap = (uint64_t)u->cap * 2;
  while (live + margin > ncap * 3 / 4) ncap *= 2;
  if (ncap > 0xFFFFFFF0ull) { set_error("the keyframes' validChunks exceed 2^32 entries"); return TF_ERR_CAPACITY; }
  TF_HIP(hipStreamSynchronize(v->stream));
  int4* na = nullptr;
  if (hipMalloc((void**)&na, sizeof(int4) * (size_t)ncap) != hipSuccess) {
    set_error("cannot grow the keyframes' validChunks store (hipMalloc)");
    return TF_ERR_HIP;
  }
  TF_HIP(hipMemcpy(na, u->arena, sizeof(int4) * (size_t)u->cap, hipMemcpyDeviceToDevice));  // (regions keep their offsets)
  TF_HIP(hipDeviceSynchronize());
  hipFree(u->arena);
  u->arena = na;
  u->cap = (uint32_t)ncap;
  u->grows += 1;
  return TF_OK;
}

static int check_group(const tf_unit_group* g, bool moved) {
  if (!g->keyframe.d_depth) { set_error("a keyframe group needs the keyframe's depth image"); return TF_ERR_INVALID; }
  if (g->kf_id < 0) { set_error("keyframe id must be >= 0"); return TF_ERR_INVALID; }
  if (g->n_local < 0 || g->n_local > kGroupFrames) { set_error("a keyframe group holds at most 6 local frames"); return TF_ERR_INVALID; }
  if ((reinterpret_cast<uintptr_t>(g->keyframe.d_depth) & 15) || (reinterpret_cast<uintptr_t>(g->keyframe.d_rgba) & 3) ||
      (reinterpret_cast<uintptr_t>(g->keyframe.d_quality) & 3)) { set_error("device images must be aligned (depth 16 B, rgba / quality 4 B)"); return TF_ERR_INVALID; }
  for (int f = 0; f < g->n_local; ++f)
    if (!g->local[f].d_depth || (reinterpret_cast<uintptr_t>(g->local[f].d_depth) & 15)) { set_error("local frames need an aligned depth image"); return TF_ERR_INVALID; }
  (void)moved;
  return TF_OK;
}

}  // namespace tf

using namespace tf;

extern "C" {

int tf_keyframe_unit_device(tf_volume* v, const tf_unit_group* fresh, const tf_unit_group* moved, int32_t n_moved,
                            int32_t texture, const float* pose_inv16) {
  if (!v || (n_moved > 0 && !moved) || n_moved < 0) { set_error("invalid argument"); return TF_ERR_INVALID; }
  if (!fresh && n_moved == 0) return TF_OK;
  if (texture && (!fresh || !fresh->keyframe.d_rgba || !pose_inv16)) {
    set_error("the texture stage needs the new keyframe's colour image and inverse pose");
    return TF_ERR_INVALID;
  }
  TF_DEV_STREAM(v);
  // the patch stage of the previous unit call / textured frame, if still pending: with a texture stage in this call it
  // rides on that stage's filter launch (23.6 us as a launch of its own, profiles/r4/keyframe_unit_*); else it goes first
  if (!texture && v->atlas.pend_patch.on) { int rcf = patch_flush(v); if (rcf) return rcf; }
  UnitState* u = nullptr;
  int rc = unit_state(v, &u);
  if (rc) return rc;
  if (fresh && (rc = check_group(fresh, false))) return rc;
  for (int m = 0; m < n_moved; ++m)
    if ((rc = check_group(&moved[m], true))) return rc;
  auto slot_for = [&](int32_t kf, bool create) -> int {
    auto it = u->slot_of.find(kf);
    if (it != u->slot_of.end()) return it->second;
    if (!create) return -1;
    if (u->slot_of.size() >= (size_t)u->slots && grow_slots(v, u) != TF_OK) return -2;
    const int s = (int)u->slot_of.size();
    u->slot_of.emplace(kf, s);
    return s;
  };
  // meshesToUpdate for the texture stage.  When nothing older is waiting for a mesher (every mark up to now was cleared
  // by a CompressMeshes: the usual case, one unit call after the other), the set is exactly what this call's groups
  // update: each group adds its updated chunks and their face neighbours right behind its finalize (k_dirty_frame, 9 us)
  // and the stage starts with the filter -- instead of a counter reset, a scan of every chunk's marks and a list
  // adoption (12 + 36 + 5 us per call).  A neighbour that a LATER group of the call creates is not in an earlier
  // group's pass: it is then updated by that group (and joins through its pass) or parked again (does not exist).
  int dirty_par = -1;
  uint32_t dirty_stamp = 0;
  if (texture && v->clear_floor >= v->epoch) {
    // (a pending patch stage -- the previous call's -- reads the meshes this call's mesher will rewrite: it rides on
    // this call's filter launch, texture_stage below; nothing in between touches what it reads)
    if ((rc = fused_arm(v))) return rc;
    dirty_par = v->atlas.fused_par;
    dirty_stamp = v->epoch + (uint32_t)(2 * n_moved + (fresh ? 1 : 0));  // = the stage's frame_epoch + 1
  }
  if ((rc = grow_arena_if_needed(v, u, n_moved + (fresh ? 1 : 0)))) return rc;
  int fresh_front = 0;
  if (fresh && can_fork(v, u, fresh)) {
    fresh_front = n_moved == 0 ? 1 : 2;
    if (fresh_front == 2 && (rc = front_ahead(v, u, fresh))) return rc;
  }
  // tsdfFusion's loop over keyframesToUpdate (:296-312): retract, de-integrate at the old poses, integrate at the new
  for (int m = 0; m < n_moved; ++m) {
    const int slot = slot_for(moved[m].kf_id, false);
    if (slot < 0) { set_error("a moved keyframe was never integrated through this entry point"); return TF_ERR_INVALID; }
    if ((rc = integrate_group(v, u, &moved[m], 0, slot, dirty_par, dirty_stamp))) return rc;
    if ((rc = integrate_group(v, u, &moved[m], 1, slot, dirty_par, dirty_stamp, nullptr, 0, !fresh && m == n_moved - 1))) return rc;
  }
  KfStoreArgs ride_store{};
  bool have_ride = false;
  if (fresh) {  // :316-323
    const int slot = slot_for(fresh->kf_id, true);
    if (slot < 0) return TF_ERR_HIP;  // (the table could not grow: hipMalloc's message is in tf_last_error)
    if ((rc = integrate_group(v, u, fresh, 1, slot, dirty_par, dirty_stamp, texture ? &ride_store : nullptr, fresh_front, true))) return rc;
    have_ride = ride_store.tab != nullptr;  // (a group without local frames stored with its dirty-set launch)
  }
  if (texture) {
    // UpdateMeshes over everything marked since the last CompressMeshes, CompressMeshes, GeneratePatches with the new
    // keyframe as the label of every chunk of chunksToUpdate, UpdateAtlas (the fused texture stage; its patch stage stays
    // pending like a streamed frame's and goes out with the next launch or the next call that looks)
    FrameImages img{fresh->keyframe.d_depth, reinterpret_cast<const uchar4*>(fresh->keyframe.d_rgba), nullptr};
    // (dirty_par >= 0: the set is in the lists of that parity -- shard lists from the group kernels, the flat list from groups
    // without local frames; the filter walks both)
    return texture_stage(v, v->dev.sel, img, v->epoch - 1u, pose_inv16, fresh->kf_id, dirty_par >= 0, nullptr, true, false, 0,
                         have_ride ? &ride_store : nullptr);
  }
  // texture == 0: UpdateMeshes only (asynchronous).  The caller's tf_compress_meshes then returns chunksToUpdate, marks /
  // exchanges the adjacency flags and clears meshesToUpdate (MobileFusion.cpp:343-355), its view selection runs, and
  // tf_generate_patches / tf_update_atlas take its labels.
  const size_t cap = (size_t)v->dev.max_chunks;
  rc = ensure_tmp(v, cap * 16 + 16);
  if (rc) return rc;
  uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
  launch_list_dirty(v->dev, reinterpret_cast<int4*>(db + 16), (uint32_t)cap, v->clear_floor, v->stream);
  TF_HIP(hipMemcpyAsync(db, &v->dev.vctl->n_tmp, 4, hipMemcpyDeviceToDevice, v->stream));
  (void)nbr_next_seq(v);
  launch_mesh(v->dev, v->mesh_par, reinterpret_cast<const int4*>(db + 16), reinterpret_cast<const uint32_t*>(db), (uint32_t)cap,
              ++v->mesh_epoch, v->res, false, -1, 1u << 30, nullptr, -1, v->stream);
  v->mesh_par ^= 1;
  TF_HIP(hipGetLastError());
  // (d_tmp holds the dirty list: a tf_compress_meshes right behind this call takes it from there instead of scanning the
  // chunks' marks again; its length stays on the device -- this call does not wait)
  v->dirty_list_n = ~0u;
  v->dirty_list_seq = v->call_seq;
  return TF_OK;
}

int tf_keyframe_unit_stats(tf_volume* v, int64_t out[5]) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  for (int k = 0; k < 5; ++k) out[k] = 0;
  UnitState* u = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_units_mu);
    auto it = g_units.find(v);
    if (it == g_units.end() || !it->second.tab) return TF_OK;
    u = &it->second;
  }
  TF_DEV(v);
  uint32_t h[4];
  TF_HIP(hipMemcpyAsync(h, u->tab, sizeof(h), hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  out[0] = u->cap; out[1] = h[0]; out[2] = h[1]; out[3] = h[2]; out[4] = h[3];
  return TF_OK;
}

int tf_keyframe_unit_stats_ex(tf_volume* v, int64_t out[4]) {
  if (!v || !out) { set_error("null argument"); return TF_ERR_INVALID; }
  for (int k = 0; k < 4; ++k) out[k] = 0;
  std::lock_guard<std::mutex> lock(g_units_mu);
  auto it = g_units.find(v);
  if (it == g_units.end() || !it->second.tab) return TF_OK;
  out[0] = (int64_t)it->second.slots;           // slots of the region table
  out[1] = (int64_t)it->second.slot_of.size();  // keyframes that own one
  out[2] = (int64_t)it->second.grows;           // doublings (table + arena) so far
  out[3] = (int64_t)it->second.cap;             // arena entries
  return TF_OK;
}

int tf_keyframe_unit_release(tf_volume* v) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  {
    std::lock_guard<std::mutex> lock(g_units_mu);
    if (g_units.find(v) == g_units.end()) return TF_OK;
  }
  TF_DEV(v);
  TF_HIP(hipStreamSynchronize(v->stream));
  std::lock_guard<std::mutex> lock(g_units_mu);
  auto it = g_units.find(v);
  UnitState& u = it->second;
  if (u.arena) hipFree(u.arena);
  if (u.tab) hipFree(u.tab);
  if (u.h_fill) hipHostFree(u.h_fill);
  if (u.group_pre) hipFree(u.group_pre);
  if (u.group_cen) hipFree(u.group_cen);
  if (u.front) { hipStreamSynchronize(u.front); hipStreamDestroy(u.front); }
  if (u.ev_mut) hipEventDestroy(u.ev_mut);
  if (u.ev_front) hipEventDestroy(u.ev_front);
  g_units.erase(it);
  return TF_OK;
}

}  // extern "C"
