// tf_device.h -- device-side data layout and kernel launchers of the MI355X fusion path.
//
// HBM layout (one process per GPU, one arena per volume, allocated once at create):
//   tsdf  [max_chunks][512] float2 {sdf, weight}     4 KiB / chunk, voxel index (z*8+y)*8+x
//   color [max_chunks][512] ushort4 {R,G,B,count}    4 KiB / chunk
//     -> an 8-voxel row (the reference's AVX2 vector, ProjectionIntegrator.cpp:145-147) is one
//        64-B segment of either plane; a wave64 touches 8 rows = 512 contiguous bytes per plane.
//   chunk hash: HEntry[hcap] = {packed id u64, slot u32, alive u32}, 16 B, so ONE load answers
//     "does the chunk exist and where".  Entries are never removed: a chunk that is
//     garbage-collected (Chisel.h:472-477) is *parked* (alive = 0, storage reset to the fresh
//     state) and revived as "new" the next time it is selected.
//   meshesToUpdate (Chisel.h:489): mark_epoch / erase_epoch u32[max_chunks] by pool slot, written
//     with plain stores by the chunk's own wave; id is dirty iff the newest mark among id and its
//     six neighbours is newer than id's erase and than the last clear (expanded on demand).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tf {

constexpr int kChunkVoxels = 512;
constexpr int kGroupFrames = 6;  // local frames of a keyframe group (GCFusion/MobileFusion.h: integrateLocalFrameNum)
constexpr uint32_t kInvalidSlot = 0xFFFFFFFFu;
constexpr uint64_t kEmptyKey = ~0ull;

// status bits (VolumeDev::status)
constexpr uint32_t kStPoolFull = 1u;
constexpr uint32_t kStListFull = 2u;
constexpr uint32_t kStCoarseFull = 4u;
constexpr uint32_t kStMissing = 8u;
constexpr uint32_t kStHashFull = 16u;
constexpr uint32_t kStMeshFull = 32u;   // a mesh exceeds tf_config.mesh_max_vertices / mesh_max_triangles
constexpr uint32_t kStAtlasFull = 64u;
constexpr uint32_t kStXchgFull = 128u;  // a rank's ghost band did not fit the exchange block (raise cap_records)  // Atlas::AddPatch overflow (std::overflow_error, Atlas.cpp:52-53)

struct Cam {
  int W, H;
  float fxi, fyi, cxi, cyi;  // already truncated to int (PinholeCamera.h:46-49)
  float nearP, farP;
};

struct Integ {
  float quad, lin, cons, scale, weight;
};

struct Pose {
  float p[12];  // row-major 3x4 [R|t]
};

struct __attribute__((aligned(16))) HEntry {
  unsigned long long key;
  uint32_t slot;
  uint32_t alive;  // bit 0 alive, bit 1 touched since the last boundary exchange
};

// Device-resident control block of one selection: everything one frame's kernels hand to the
// next, so the fused per-frame unit never synchronises with the host.
struct FrameCtl {
  uint32_t bbox_key[6];  // ordered-uint keys of min xyz / max xyz (K-B reduction)
  int32_t min_id[3];
  int32_t max_id[3];
  int32_t dims[3];  // coarse-block grid
  uint32_t n_coarse;
  uint32_t n_list;   // entries of the visible list
  // Lists written by the fused selection role are TWO-ENDED: chunks the selection expects to be rewritten
  // ("expensive": some probe corner inside the truncation band) are appended from the front, the others from the
  // back of the arrays, so that logical entry e sits at list_phys(e).  K-A deals logical entries round-robin to
  // its resident waves; with ~1.5 entries per wave the second entry of a wave is then a cheap one.  Lists of the
  // call-by-call flow are plain (n_front = n_list).
  uint32_t n_front;
  uint32_t pad0;
  unsigned long long emit_pack;  // the selection role's append counters: low word front, high word back entries
  uint32_t pad[2];
  // multi-GPU, fused flow: selected chunks of this frame per ghost band -- [0] own down band (keys lo .. lo + a + b + c,
  // read by the rank below), [1] own up band (key hi - 1, read by the rank above), [2] the up band of the rank below
  // (key lo - 1), [3] the down band of the rank above (keys hi .. hi + a + b + c).  Every rank runs the same selection,
  // so sender and receiver of a block count the same number: the exchange of the frame is sized by it (selected is a
  // superset of updated).  Zeroed with emit_pack by the frame's K-B stage.
  uint32_t band_cnt[4];
};

// Volume-wide device words.
constexpr int kSlotStripes = 64;
struct VolCtl {
  uint32_t status;    // sticky error bits
  uint32_t n_tmp;     // scratch counter of the on-demand list/pack kernels
  uint32_t ovf_next;  // mesh store, large pool: blocks handed out (bump allocation, reset with the volume)
  uint32_t n_tmp2;    // second scratch counter (the "up" block of the two-band boundary pack)
  uint32_t xchg_sent, xchg_recv;  // ghost records written into / stored from exchange blocks since create (tf_comm_stats_ex)
  uint32_t zero_word;  // always 0: an empty list's count (the interior mesh pass of an overlapped exchange ignores the flat list)
  uint32_t blk_next;   // mesh store, small pool: blocks handed out (bump allocation, reset with the volume)
  // the two-band boundary pack (k_boundary_pack<true>): records counted per block, workgroups that are through; the LAST
  // workgroup writes the blocks' headers and re-arms the three words (no memset, no header launch per exchange)
  uint32_t xchg_cnt[2];
  uint32_t xchg_ticket;
  // multi-GPU: entries of the two lists of "touched" ghost-band chunks (VolumeDev::xl_ent): the voxel kernels append a chunk
  // when its touched bit goes 0 -> 1, the pack launch consumes the list of its parity and re-arms it
  uint32_t xl_n[2];
  // neighbour table (VolumeDev::nbr): VolumeDev::seq of the newest launch that INSERTED a key into the chunk hash.  A row's
  // "no chunk there" entries are trusted only if the row was checked by a launch with a larger seq (kNbrStamp).
  uint32_t create_seq;
  // mesh store: blocks that were given back (blk_release: a mesh that became empty, a small block whose mesh moved to the
  // large pool) wait in one ring per pool -- [0] small, [1] large -- behind VolumeDev::mesh_rec; blk_take hands them out
  // again before the bump counters above move.  Monotone tickets: head = taken, tail = released.
  uint32_t blk_head[2], blk_tail[2];
  // bit 0 / 1: the small / large pool has handed out half of its blocks at some point -- from then on blocks go back to the
  // rings (sticky until the volume is reset; before that a stream pays nothing for the rings: one read-mostly word)
  uint32_t blk_recycle;
  // Pool slots are handed out from 64 independent stripes (stripe s owns slots
  // [s*max_chunks/64, (s+1)*max_chunks/64)) so that the thousands of chunk creations of a
  // first-touch frame do not serialise on one atomic word.
  uint32_t slot_cnt[kSlotStripes];
};

// Per-frame selection scratch.  Two sets exist so that the selection of frame f+1 (a pure
// function of its depth image and pose) can run on a second stream while frame f integrates.
struct SelBuf {
  unsigned long long* masks;  // [max_coarse]
  uint32_t* offsets;          // [max_coarse] k_scan's tile words (64-bit: {launch stamp | kind | entries}, one per 256 coarse blocks)
  int4* list_id;              // [max_list]
  float4* list_pre;           // [4*max_list] 64-B records {o.x,o.y,o.z,wD}, {upper,id.x,id.y,id.z}, spare x2
  uint32_t* list_slot;        // [max_list]
  uint32_t* list_ent;         // [max_list] hash entry index of the chunk
  uint8_t* list_new;          // [max_list]
  uint8_t* list_needs;        // [max_list]
  float* list_quality;        // [max_list]
  uint16_t* list_rows;        // [max_list] low byte tsdf rows, high byte colour rows
  float* cen;                 // [3*512] centroid table of the frame (Chisel.cpp:52-110), c[a][voxel]
  FrameCtl* ctl;
};

// ---- device-resident meshes (ChunkManager::allMeshes) and their patches (Mesh::m_patch) ----------
// Blocks of the mesh store, planar (lane = vertex / triangle, every plane row is a coalesced
// segment):  mesh_v [block][17][CV] f32 -- Mesh::vertices x y z | normals x y z | colors r g b |
// Patch::texcoord u v | texcolor r g b | labs r g b;  mesh_t [block][3][CT] u16 -- Mesh::indices as
// triangle corner planes (a mesh has at most 3 * 729 vertices).  CV / CT come from tf_config
// (mesh_max_vertices / mesh_max_triangles); the theoretical maximum is 2187 / 2560.
constexpr int kMeshPlanes = 17;
constexpr int kMpPos = 0, kMpNrm = 3, kMpCol = 6, kMpTc = 9, kMpTcol = 11, kMpLabs = 14;
constexpr unsigned long long kNoTexloc = ~0ull;
constexpr uint32_t kMsInMap = 1u;       // the chunk has an entry in allMeshes (ChunkManager::HasMesh)
constexpr uint32_t kMsSimplified = 2u;  // Mesh::simplified
constexpr uint32_t kMsOverflow = 4u;    // no block of the mesh store was left for it: stored empty, kStMeshFull raised
constexpr int kMsAdjShift = 8;          // bits 8..13 = Mesh::adj[0..5]
// The mesh store is allocated ON DEMAND: a pool slot owns no mesh storage until its chunk first has a mesh with vertices.
// Then it is given a block (one bump allocation per chunk, ever; MeshRec::block = index + 1, 0 = none) of the SMALL
// pool -- CV vertices / CT triangles (tf_config.mesh_max_*), 20.4 KiB at the defaults -- or, for a mesh that does not fit
// that, of the LARGE pool, sized for the largest mesh a chunk can have (3 x 9^3 = 2187 vertices, 512 x 5 = 2560 triangles,
// Structure/ChunkManager.cpp:856-918 emits whatever a chunk produces): bit 31 of MeshRec::block.  A chunk keeps the block
// it was given (its later, smaller meshes stay there too; a chunk that outgrows a small block moves to a large one).
constexpr uint32_t kBlkLarge = 0x80000000u;
constexpr uint32_t kBlkNone = 0u;
constexpr int kOvfCV = 2240, kOvfCT = 2560;
constexpr uint32_t kPfHasPatch = 1u;    // Atlas::HasPatch
constexpr uint32_t kPfCaution = 2u;     // CalculateTexCoords returned -1
constexpr uint32_t kPfWrong = 4u;       // Patch::wrong_mapping
constexpr uint32_t kPfHasImage = 8u;    // Patch::has_image
constexpr uint32_t kPfAdjusted = 16u;   // Patch::has_adjusted
struct __attribute__((aligned(16))) MeshRec {  // 64 B per pool slot
  uint16_t nv, nt;            // (a mesh has at most 2187 vertices / 2560 triangles)
  uint32_t block;             // the mesh's storage: block index + 1 | kBlkLarge, kBlkNone = no storage yet
  uint32_t state;
  uint32_t epoch;             // the meshing pass that wrote the block (0 = never)
  unsigned long long texloc;  // Patch::texloc (linear texel index of the atlas slot), kNoTexloc = no slot yet
  int32_t frameid;            // Patch::frameid
  uint32_t pflags;
  int32_t bbox[4];            // Patch::boundingbox x y w h
  float ratio[2];             // Patch::ratio
  int32_t kf_slot;            // keyframe-table entry Patch::image views (-1 = released)
  uint32_t stamp;             // de-duplication stamp of the per-frame dirty list
};

// Keyframe images the patches are cut from (Frame::rgb / refined_depth, Patch::SetImage holds a
// non-owning ROI, Patch.cpp:172-175) and the keyframe's pose as CalculateTexCoords uses it.
struct __attribute__((aligned(16))) KfDev {
  const uint8_t* rgb;   // u8[H][W][stride], stride 3 (Frame::rgb) or 4 (the path's RGBA staging image)
  const float* depth;   // f32[H][W]
  float T[16];          // f32(SE3d.inverse().matrix()), row-major (Patch.cpp:51)
  int32_t stride;
  int32_t kf_id;        // Frame id = the label view selection hands out
  int32_t pad[2];
};
// Device words of the atlas: the slot allocator's position (Atlas::loc_next as a slot count), the hot
// range of the last GeneratePatches, the fused per-frame work lists.
struct AtlasCtl {
  uint32_t n_slots;              // slots handed out so far
  uint32_t n_done;               // patches projected by the last GeneratePatches
  unsigned long long loc_min, loc_max;  // min / max texloc of the last GeneratePatches (Chisel.cpp:153-186)
  // work-list counters, double-buffered by frame parity in the fused per-frame flow (the kernels of frame
  // f re-arm the set of frame f + 1; the call-by-call flow uses set 0)
  struct Set {
    uint32_t n_work;               // entries of the patch work list
    uint32_t n_cand;               // work entries that still need an atlas slot
    unsigned long long slots_base; // fused flow: AtlasCtl::n_slots before this frame's new patches (snapshot by k_compress_exchange)
    uint32_t n_patch;              // (unused)
    uint32_t pad;
  } set[2];
};

constexpr uint32_t kMeshShards = 32;  // survivor rows are appended shard by shard: 32 counters instead of one
constexpr uint32_t kMeshCntWords = kMeshShards * 48;  // VolumeDev::mesh_cnt per launch parity: row counters, statistics, deferred-reset counters
constexpr int kPhaseWaves = 16384;  // rows of the wave-timeline table (tuning aid)

struct VolumeDev {
  // pool
  float2* tsdf;
  ushort4* color;
  uint32_t max_chunks;
  VolCtl* vctl;
  // chunk hash
  HEntry* hent;
  uint32_t hmask;
  // Chisel::meshesToUpdate, kept implicitly: per pool slot the finalize epoch (+1) in which the
  // chunk was last updated (mark) / garbage-collected (erase).  The 6-neighbourhood closure of
  // Chisel.h:197-203 is expanded when the set is read (k_list_dirty), not on the per-frame path.
  uint32_t* mark_epoch;   // [max_chunks]
  uint32_t* erase_epoch;  // [max_chunks]
  // Per pool slot a conservative summary of the chunk's voxels for the mesher's filter (chunk_summary_bits in
  // tf_devfn.h): bits 0-3 = {some sdf <= 1, some 0 < sdf <= 1, some sdf < 0, some weight > 50} over the whole chunk,
  // bits 4-7 / 8-11 / 12-15 the same over its x = 0 / y = 0 / z = 0 face.  Every voxel writer ORs the class of what
  // it writes in, so a bit may outlive the voxel that set it (never the other way round); k_mesh_filter rewrites
  // the word exactly whenever it reads the chunk's voxels.
  uint32_t* summ;         // [max_chunks]
  // Neighbour table: per pool slot one 128-byte row.  Words 0..26 = pool slot + 1 of the chunk at id + (dx, dy, dz),
  // word index (dx + 1) + 3 (dy + 1) + 9 (dz + 1), 0 = no chunk known there (word 13, the chunk itself, is unused);
  // word kNbrStamp = VolumeDev::seq of the launch that last checked the row against the hash.  Pool slots never move and hash entries are never
  // removed (a garbage-collected chunk is parked: alive = 0, voxels back in the fresh state, summary 0, mesh out of the map
  // -- which is exactly what every reader by slot assumes of a chunk that does not exist), so a non-zero word is true for
  // the life of the volume; a zero word is true while no key has been inserted since the check (VolCtl::create_seq).  The
  // mesher's filter fills rows lazily: ONE coalesced load instead of 8 + 19 hash probes per dirty chunk and frame; the
  // patch stage reads the six face words for CompressMeshes' flag exchange.
  uint32_t* nbr;          // [max_chunks][kNbrWords]
  uint32_t seq;           // launch sequence of the table: the host bumps it ahead of every filter launch (launch_mesh)
  unsigned long long* phase_buf;  // [kPhaseWaves][16] tuning aid: per-wave {start, end, role, XCC} stamps (TF_KA_DBG bit 12)
  uint32_t max_list;
  uint32_t max_coarse;
  // partition (multi-GPU chunk-range ownership): lo <= id.x < hi
  // multi-GPU ownership: key(id) = part_a * x + part_b * y + part_c * z (coefficients >= 0), owned
  // iff part_lo <= key < part_hi; (1, 0, 0) = ChunkID.x slabs
  int32_t part_lo, part_hi;
  int32_t part_a, part_b, part_c;
  // meshes
  float* mesh_v;
  uint16_t* mesh_t;
  MeshRec* mesh_rec;
  uint32_t mesh_cv, mesh_ct;
  uint32_t mesh_blocks;  // blocks of the small pool (mesh_v / mesh_t); VolCtl::blk_next = blocks handed out
  float* ovf_v;         // [ovf_blocks][17][kOvfCV] large pool, same planar layout as mesh_v
  uint16_t* ovf_t;      // [ovf_blocks][3][kOvfCT]
  uint16_t* ovf_vlist;  // [ovf_blocks][kOvfCV] mesher scratch of a block (its used edge slots; the slot-sized list sits in LDS)
  uint32_t ovf_blocks;  // blocks of the pool; VolCtl::ovf_next = blocks handed out
  uint32_t* mesh_nbr;  // [kMeshShards][mesh_shard_rows(max_chunks)][32] mesher scratch, one row per SURVIVING work entry
                       // (k_mesh_filter): pool slots of its 27-chunk neighbourhood, [27] = the entry's list index
  // [2][kMeshCntWords], double-buffered by launch parity: [shard][16] rows used per shard (one counter per 64-B line), then
  // [shard][16] statistics {exact tests, rows with a surface cell} -- in lines of their own: every workgroup of the mesher
  // reads the row counters at its start, and atomics landing in those lines during the launch delayed that read (mesher
  // 36 -> 49 us with two statistics atomics per chunk next to the row counter)
  uint32_t* mesh_cnt;
  // Records the filter wants emptied ("cannot have a vertex": Mesh::Clear, ChunkManager.cpp:244-262) are not written by the
  // filter launch -- the patch stage of the PREVIOUS frame runs in that launch and still reads those records -- but listed
  // here {id, w = pool slot}, shard by shard (counters: third block of mesh_cnt), and applied by the first workgroups of
  // the mesher launch behind it.
  int4* reset_list;  // [kMeshShards][mesh_shard_rows(max_chunks)]
  // atlas (Structure/Atlas.h:43-75): u8 [atlas_h][atlas_w][3], slots of patch_w x patch_h texels
  uint8_t* atlas;
  int32_t atlas_w, atlas_h, patch_w, patch_h;
  AtlasCtl* actl;
  KfDev* kf_tab;      // [max_keyframes]
  int4* work_ids;       // [max_chunks] work list (dirty chunks of a frame / chunksToUpdate): id, w = keyframe-table entry
  uint32_t* work_slot;  // [max_chunks] pool slot of the entry, kInvalidSlot = not processed
  int4* patch_list;     // [2][kMeshShards][mesh_shard_rows] fused flow: {packed id lo, packed id hi, pool slot, mesh block} of the frame's
                        // dirty chunks that own a mesh, appended by the mesher (and by its filter for meshes that just became empty)
  uint32_t* patch_cnt;  // [2][kMeshShards][16] entries per shard (one counter per 64-B line), by frame parity
  unsigned long long* cand;  // [max_chunks] packed ids of the work entries that need an atlas slot
  // dirty set of a textured frame as K-A builds it (each updated chunk's wave claims the chunk and its six face
  // neighbours by stamp, Chisel.h:197-203): 32 shard lists by pool slot % 32, double-buffered by frame parity like the
  // flat work list, which other producers (backlog list, ghost arrivals) keep using -- the mesher's filter walks both
  int4* wl_ids;        // [2][kMeshShards][mesh_shard_rows] {id, w = hash entry index + 1 (0 = the slot is known alive)}
  uint32_t* wl_slot;   // [2][kMeshShards][mesh_shard_rows] pool slot
  uint32_t* wl_cnt;    // [2][kMeshShards][16] entries per shard (one counter per 64-B line)
  // Chunk::observations (Chunk.h:171) on the device: open-addressing table keyed by (pool slot << 32 | keyframe id) ->
  // quality; an erased observation keeps its key with quality 0 (recorded qualities are > 0, Chisel.h:244)
  unsigned long long* obs_key;  // [obs_mask + 1], kEmptyKey = free
  float* obs_q;
  uint32_t obs_mask;
  // multi-GPU: the chunks whose "touched" bit (HEntry::alive bit 1: a ghost-band chunk updated since it was last packed) is
  // set, as hash-entry indices -- two lists: the voxel kernels append to list xl_par, the boundary pack consumes list xl_par
  // (what does not fit its blocks goes to the other list, still flagged) and the host flips xl_par behind every pack launch.
  // The pack's cost follows the few hundred chunks a frame touches, not the size of the hash (a scan of the 2^20 entries of
  // the bench's pool took 35-49 us per frame, more than the slab's voxel update, filter and mesher together).
  uint32_t* xl_ent;  // [2][max_chunks]
  uint32_t xl_par;
  SelBuf sel;  // the selection set the launch works on
};
// planes of the mesh stored in block `blk` (MeshRec::block; only dereferenced for a mesh with vertices, i.e. blk != kBlkNone)
__host__ __device__ inline float* mesh_plane(const VolumeDev& v, uint32_t blk, int plane) {
  const size_t i = (size_t)((blk & ~kBlkLarge) - 1u);
  return (blk & kBlkLarge) ? v.ovf_v + (i * kMeshPlanes + plane) * kOvfCV : v.mesh_v + (i * kMeshPlanes + plane) * v.mesh_cv;
}
__host__ __device__ inline uint16_t* tri_plane(const VolumeDev& v, uint32_t blk, int corner) {
  const size_t i = (size_t)((blk & ~kBlkLarge) - 1u);
  return (blk & kBlkLarge) ? v.ovf_t + (i * 3 + corner) * kOvfCT : v.mesh_t + (i * 3 + corner) * v.mesh_ct;
}
// length of a pool's ring of released blocks (blk_release / blk_take, tf_devfn.h): the power of two >= blocks (>= 2)
__host__ __device__ inline uint32_t blk_ring_len(const uint32_t blocks) {
  uint32_t n = 2;
  while (n < blocks) n <<= 1;
  return n;
}
__host__ __device__ inline uint32_t mesh_cap_v(const VolumeDev& v, uint32_t blk) { return blk == kBlkNone ? 0u : ((blk & kBlkLarge) ? (uint32_t)kOvfCV : v.mesh_cv); }
__host__ __device__ inline uint32_t mesh_cap_t(const VolumeDev& v, uint32_t blk) { return blk == kBlkNone ? 0u : ((blk & kBlkLarge) ? (uint32_t)kOvfCT : v.mesh_ct); }

struct FrameImages {
  const float* depth;
  const uchar4* rgba;    // may be null
  const float* quality;  // may be null
};

// One frame of a stream as seen by the pipelined launcher.
struct FrameStage {
  SelBuf sel;
  FrameImages img;
  Pose pose;
  uint32_t epoch;
  int claim_par = -1;        // >= 0: a mesher follows this frame; K-A builds the frame's dirty set into the shard lists of this parity
  bool small_frame = true;   // the frame's lists are short (a room, not a hall): the patch / selection ranges are dispatched
                             // ahead of K-A (launch_frame)
  bool coarse_summ = false;  // no mesher follows this frame: an updated chunk's summary becomes "anything" (kSummAny)
                             // instead of the classes written; the filter makes it exact when it next reads the chunk
};
// The patch stage of the previous textured frame, carried by the next frame's launch (FrameLaunch::kf_patch).
struct PatchStage {
  int par;   // counter-set parity of that frame (AtlasCtl::set, patch_list, patch_cnt)
  KfDev kf;  // the frame itself as the keyframe its patches are cut from
};
constexpr int kNbrWords = 32, kNbrStamp = 27;  // VolumeDev::nbr
constexpr uint32_t kSummAny = 0xFFFFu;        // VolumeDev::summ: every class may occur
constexpr uint32_t kKaCoarseSumm = 16384u;    // IntegrateConsts::dbg bit: FrameStage::coarse_summ

// ---- launchers (tf_kernels.hip) ------------------------------------------------------
// patch != nullptr (and cur a colour frame): the launch also carries the patch stage of the previous frame
void launch_frame(const VolumeDev& v, const FrameStage* cur, const FrameStage* next,
                  const FrameStage* next2, const PatchStage* patch, const Cam& cam, const Integ& ig, float res, hipStream_t s,
                  uint32_t* progress = nullptr, uint32_t* progress_seq = nullptr);  // host-visible word a launch with a K-B role stamps with ++*progress_seq
void launch_reset_ctl(const VolumeDev& v, bool volume_too, hipStream_t s);
void launch_null(hipStream_t s);
void launch_bbox(const VolumeDev& v, const float* depth, const Cam& cam, const Pose& pose,
                 hipStream_t s);
void launch_select(const VolumeDev& v, const float* depth, const Cam& cam, const Integ& ig,
                   const Pose& pose, float res, bool emit, hipStream_t s, bool plain = false);
void launch_scan(const VolumeDev& v, int step, hipStream_t s);
void launch_acquire(const VolumeDev& v, hipStream_t s);
void launch_lookup(const VolumeDev& v, uint32_t n, hipStream_t s);
// have_pre: the list records / centroid table of this pose are already there (launch_pre_frames)
void launch_integrate(const VolumeDev& v, const FrameImages& img, const Cam& cam, const Integ& ig,
                      const Pose& pose, float res, int flag, bool use_color, bool use_quality,
                      uint32_t epoch, hipStream_t s, bool have_pre = false);
// list records + centroid tables of a keyframe (into the selection set, as k_pre leaves them) and of its n local frames
// (into the group scratch, as k_pre_group leaves them) in ONE launch
void launch_pre_frames(const VolumeDev& v, const Pose& keyframe, int n, const float* poses12, float4* pre_scratch,
                       float* cen_scratch, const Integ& ig, float res, const Cam& cam, hipStream_t s,
                       int acquire = 0,   // 1: the last row of workgroups resolves the slots of the list k_select<EMIT, plain> appended (acquire_emitted_body); 2: ... lazily
                       uint32_t* clear_word = nullptr);  // a device word the launch sets to 0 (the unit: kf.validChunks.clear())
void launch_finalize(const VolumeDev& v, uint32_t epoch, hipStream_t s);
int device_cus();  // compute units of the current device (tf_kernels.hip)
// the plain list's ids and isNew flags (first min(n_list, cap) entries) into host-visible memory
void launch_export_list(const VolumeDev& v, int4* h_ids, uint8_t* h_new, uint32_t cap, hipStream_t s);
// FrameCtl (without the pull counters) followed by VolCtl, as words, into host-visible memory
void launch_export_ctl(const FrameCtl* f, const VolCtl* vc, uint32_t* h, hipStream_t s);
// needs flags (+ quality sums) of the current list and VolCtl::status into host-visible pinned memory
void launch_export_integrate(const VolumeDev& v, uint32_t n, uint8_t* h_needs, float* h_quality, uint32_t* h_status, hipStream_t s);
void launch_pack_rgba(const uint8_t* rgb, const uint8_t* valid, uchar4* rgba, uint32_t npix, hipStream_t s);
void launch_fill_pool(const VolumeDev& v, uint32_t slot0, uint32_t nslots, hipStream_t s);
void launch_rowstats(const VolumeDev& v, unsigned long long* out3, hipStream_t s);
void launch_list_chunks(const VolumeDev& v, int4* out, uint32_t cap, hipStream_t s);
void launch_list_dirty(const VolumeDev& v, int4* out, uint32_t cap, uint32_t clear_floor, hipStream_t s);
void launch_gather_chunks(const VolumeDev& v, const int4* ids, uint32_t n, float* sdf, float* w,
                          uint16_t* col, uint32_t* found, hipStream_t s);
void launch_scatter_chunk(const VolumeDev& v, int4 id, const float* sdf, const float* w,
                          const uint16_t* col, hipStream_t s);
void launch_boundary_pack(const VolumeDev& v, uint8_t* records, uint32_t cap, hipStream_t s);
// two blocks: what the rank below / the rank above reads as ghosts (counts in VolCtl::n_tmp / n_tmp2)
// (the blocks' in-band counts -- first word of each 16-byte header ahead of the records -- are written by the launch itself)
void launch_boundary_pack_bands(const VolumeDev& v, uint8_t* block_down, uint8_t* block_up, uint32_t cap_down,
                                uint32_t cap_up, hipStream_t s);
// the blocks' in-band counts (first word of each header) from VolCtl::n_tmp / n_tmp2; hdr_b may be null
void launch_boundary_headers(const VolumeDev& v, uint32_t* hdr_a, uint32_t cap_a, uint32_t* hdr_b, uint32_t cap_b, hipStream_t s);
// FrameCtl::band_cnt -> host_words[1..4], then host_words[0] = tag (system-scope release)
void launch_xchg_publish(const FrameCtl* ctl, uint32_t* host_words, uint32_t tag, hipStream_t s);
void launch_boundary_unpack(const VolumeDev& v, const uint8_t* records, uint32_t n, hipStream_t s);
// blocks of [16-B header {count} | cap records] per rank, the block of `skip` is this rank's own
// (blocks_b: the second of exactly two separately placed blocks with its own capacity; pub_*: also publish the next
// frame's band counts like launch_xchg_publish)
void launch_boundary_unpack_blocks(const VolumeDev& v, const uint8_t* blocks, int nblocks, int skip, uint32_t cap,
                                   int dirty_par, uint32_t stamp, hipStream_t s, const uint8_t* blocks_b = nullptr,
                                   uint32_t cap_b = 0, const FrameCtl* pub_ctl = nullptr, uint32_t* pub_words = nullptr,
                                   uint32_t pub_tag = 0);
// Chunk::observations on the device (f-4: what TexMap::update_datacost reads, Structure/TexMap.cpp:64-105)
void launch_obs_record(const VolumeDev& v, int32_t kf_id, hipStream_t s);                   // Chisel.h:244-247 over the current list
void launch_obs_retract(const VolumeDev& v, int32_t kf_id, const int4* ids, uint32_t n, hipStream_t s);  // MobileFusion.cpp:252-272
// out[i * (1 + m) + 0] = observations[frame_index] of chunk ids[i], [1 + j] = observations[frames[j]]; 0 = none
void launch_obs_export(const VolumeDev& v, const int4* ids, uint32_t n, int32_t frame_index, const int32_t* frames, int32_t m,
                       float* out, hipStream_t s);
// edges of TexMap::update_chunkgraph (TexMap.cpp:50-62): {i, neighbour id} for every set Mesh::adj flag of ids[i] whose
// neighbour owns a mesh; *count edges (atomic append, order arbitrary)
void launch_adj_export(const VolumeDev& v, const int4* ids, uint32_t n, int4* out, uint32_t cap, uint32_t* count, hipStream_t s);
// ---- launchers (tf_mesh.hip) ---------------------------------------------------------
// dlist: int4 {id.x, id.y, id.z, -} per dirty chunk, *dcount entries
void launch_init_meshes(const VolumeDev& v, hipStream_t s);
// fused = the per-frame flow: the mesh is marked simplified at once (CompressMeshes follows in the same frame)
// keyframe group: n (<= 6) depth-only frames over the current list in one visit per chunk; scratch = n x (4 x max_list float4 + 1536 floats)
void launch_integrate_group(const VolumeDev& v, int n, const float* const* d_depth, const float* poses12, float4* pre_scratch,
                            float* cen_scratch, const Cam& cam, const Integ& ig, float res, int flag, hipStream_t s,
                            bool have_pre = false, int32_t obs_kf = -1,  // obs_kf >= 0: also tf_observations_record of the keyframe
                            int fin = 0, uint32_t fin_epoch = 0,         // fin: also launch_finalize(fin_epoch)
                            int claim_par = -1, uint32_t claim_stamp = 0,  // (with fin) claim_par >= 0: also the dirty-set pass, into the shard lists
                            const FrameImages* key_img = nullptr, const Pose* key_pose = nullptr);  // the keyframe's own depth + colour pass first
                                                                                                   // (its records: the list's, launch_pre_frames)
uint32_t mesh_shard_rows(uint32_t max_chunks);
// len_guess: the list length as far as the host knows (picks the filter's form); len_hint: host-visible word that
// receives the actual length (may be null)
// shards_par >= 0: the dirty set is the flat list PLUS the shard lists of that parity (VolumeDev::wl_*)
// patch != nullptr: the filter launch also carries the patch stage of the PREVIOUS frame (block range ahead of the
// filter's; tf_patch_body.h) -- returns true when it did (the fused-filter form of the mesher has no such launch)
// store != nullptr: one more workgroup of the filter launch stores the finalized list of v.sel as a keyframe's validChunks
// (kf_store_body, tf_kf_store.h: the keyframe unit -- a single-workgroup launch of its own is 11 us at its launch floor)
struct KfStoreArgs;
bool launch_mesh(const VolumeDev& v, int cnt_par, const int4* dlist, const uint32_t* dcount, uint32_t max_entries,
                 uint32_t epoch, float res, bool fused, int rearm_set, uint32_t len_guess, uint32_t* len_hint, int shards_par,
                 hipStream_t s, const PatchStage* patch = nullptr, const Cam* cam = nullptr, int cls = 0,  // cls: FilterPatch::cls
                 const KfStoreArgs* store = nullptr);
// per-frame dirty set of the fused flow -> work list of counter set `par` (when K-A did not build it: FrameStage::claim_par)
void launch_dirty_frame(const VolumeDev& v, int par, uint32_t stamp, hipStream_t s);
// ... when marks of earlier frames are still waiting for a mesher: everything marked since clear_floor
void launch_dirty_backlog(const VolumeDev& v, int par, uint32_t clear_floor, hipStream_t s);
void launch_compress(const VolumeDev& v, const int4* list, const uint32_t* count, uint32_t cap, bool mark,
                     hipStream_t s);
// sums over the work list of counter set `par`: {entries, with mesh, vertices, triangles, ROI pixels, patches}
void launch_texture_stats(const VolumeDev& v, int par, unsigned long long* out6, hipStream_t s);

}  // namespace tf
