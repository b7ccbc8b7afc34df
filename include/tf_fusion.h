/*
 * tf_fusion.h -- C ABI of the MI355X-native voxel-fusion + texture-atlas hot path.
 *
 * This is the drop-in boundary: the reference (THU-luvision/TextureFusion) has no FFI of its
 * own; the "operator API" of the path is the public C++ surface of chisel::Chisel /
 * ChunkManager / Atlas / Patch as called from GCFusion/MobileFusion.{h,cpp}.
 * Each entry point below names the reference interface it stands in for (paths relative to
 * the reference root).  The host-side C++ classes in texturefusion_amd/host/ keep the
 * reference's class/method names and forward to these functions; INTEGRATION.md shows the
 * binding a maintainer adds to the reference tree.
 *
 * Conventions
 *   - plain C, opaque handle, POD arguments only; no exceptions cross the boundary.
 *   - every function returns TF_OK (0) or a negative TF_ERR_* code; tf_last_error() gives the
 *     text of the most recent failure on the calling thread.
 *   - host pointers are borrowed for the duration of the call; outputs are caller-allocated
 *     with a capacity and a count-out.
 *   - one handle = one HIP stream; calls on a handle are serialised by the caller (the
 *     reference drives the path from a single map thread, GCFusion/MobileFusion.cpp:99-112).
 *   - poses are Eigen::Affine3f as 12 floats, row-major 3x4 [R | t] (camera-to-world).
 *   - chunk ids are int32[3]; lists are int32[3*n], order = the reference's list order.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     TF_ERR_NO_DEVICE.
 */
#ifndef TF_FUSION_H_
#define TF_FUSION_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define TF_API __attribute__((visibility("default")))
#else
#define TF_API
#endif

#define TF_OK 0
#define TF_ERR_ATLAS_FULL (-1)    /* == Chisel::GeneratePatches' -1 (Structure/Chisel.cpp:170-173) */
#define TF_ERR_INVALID (-2)       /* bad argument / call order */
#define TF_ERR_CAPACITY (-3)      /* chunk pool, list or scratch capacity exceeded */
#define TF_ERR_HIP (-4)           /* HIP runtime error */
#define TF_ERR_NO_DEVICE (-5)     /* no gfx950 device visible */
#define TF_ERR_MISSING_CHUNK (-6) /* list names a chunk that does not exist (chunks.at() would throw) */

typedef struct tf_volume tf_volume;

/* Sizing of the device-resident state (all allocated once, at create). 0 = default. */
typedef struct {
  int32_t device;        /* HIP device ordinal */
  int64_t max_chunks;    /* chunk pool capacity (8 KiB of HBM each); default 1<<20 */
  int64_t max_list;      /* max chunks in one visible-chunk list; default 1<<19 */
  int64_t max_coarse;    /* max 4x4x4 candidate blocks per frame; default 1<<20 */
  int32_t atlas_w;       /* default 13824 (Structure/Atlas.h:29) */
  int32_t atlas_h;       /* default 13824 (Structure/Atlas.h:30) */
  int32_t max_keyframes; /* keyframe image cache slots for the atlas; default 64 */
  /* blocks of the LARGE pool of the mesh store (167 KB each, room for the largest mesh a chunk can have); 0 = default
   * max(256, max_chunks / 64), at most 65535; -1 = none */
  int32_t mesh_overflow_blocks;
  /* device-resident meshes (ChunkManager::allMeshes) live in a store that is allocated ON DEMAND: a chunk-pool slot owns
   * no mesh storage until its chunk first has a mesh with vertices; then it is given a block of mesh_max_vertices /
   * mesh_max_triangles (0 = default 256 / 512: 20.4 KiB; a planar surface through a chunk has 81 / 128) out of
   * mesh_blocks blocks, or -- a mesh has at most 2187 vertices / 2560 triangles (9x9x9x3 edge grid, 512 cells x 5), and
   * the reference emits whatever a chunk produces (Structure/ChunkManager.cpp:856-918) -- a block of the large pool.
   * Once half of a pool has been handed out, a chunk whose mesh comes out empty (Mesh::Clear(), ChunkManager.cpp:254) or
   * outgrows its small block gives the block back and later meshes reuse it: a pool has to hold the meshes that have
   * vertices at one time (plus one generation of turnover), not every mesh there ever was.  Only when the pool a mesh needs
   * is exhausted is the mesh stored empty and reported as TF_ERR_CAPACITY at the next synchronising call. */
  int32_t mesh_max_vertices;
  int32_t mesh_max_triangles;
  /* blocks of the small pool; 0 = default max_chunks: every chunk can own a mesh, as in the reference, whose allMeshes never
   * drops one (ChunkManager.cpp:260-262) -- 20.4 KiB of HBM per pool slot at the default block size.  A caller that knows its
   * scene can take less (about 30 % of the chunks of a scanned room lie on a surface: DESIGN.md s.2); at most max_chunks */
  int64_t mesh_blocks;
} tf_config;

/* Counters of the most recent frame / list (device-side integers, read back on request). */
typedef struct {
  int64_t n_coarse;       /* candidate 4x4x4 blocks tested */
  int64_t n_selected;     /* chunks the reference's selection flags (GetChunkIDsObservedByCamera) */
  int64_t n_updated;      /* list entries whose needsUpdate flag is set */
  int64_t rows_tsdf;      /* 8-voxel rows whose sdf/weight were rewritten by the last integrate */
  int64_t rows_color;     /* 8-voxel rows whose colour was rewritten by the last integrate */
  int64_t n_chunks;       /* chunks alive in the volume */
  int64_t n_slots;        /* pool slots in use (alive + parked) */
  int64_t n_dirty;        /* entries of meshesToUpdate */
  int32_t min_id[3];      /* Chisel::minChunkID */
  int32_t max_id[3];      /* Chisel::maxChunkID */
  int64_t n_listed;       /* = n_selected (kept for layout: a build with depth-tile pruning of the selection subtracted the
                             pruned entries here; variants/r4_experiments.patch) */
} tf_stats;

/* What the textured per-frame unit did for its most recent frame (integers behind the byte counts). */
typedef struct {
  int64_t n_dirty;      /* chunks handed to the mesher */
  int64_t n_meshes;     /* of those, chunks in allMeshes (= patches projected) */
  int64_t n_vertices;   /* vertices of those meshes */
  int64_t n_triangles;
  int64_t roi_pixels;   /* sum of the patches' bounding-box areas */
  int64_t n_patches;    /* patches with an image */
  int64_t n_slots;      /* atlas slots handed out so far */
  int64_t n_exact;      /* dirty chunks whose own voxels were read for the "can it have a vertex" test (the class summaries ruled out the rest) */
  int64_t n_survivors;  /* of those, chunks handed to the marching-cubes kernel */
  int64_t n_surface;    /* of those, chunks in which marching cubes found a cell the surface passes through */
} tf_texture_stats;

/* Per-kernel timings collected with HIP events on the handle's stream (tf_profile_*). */
#define TF_PROF_BBOX 0
#define TF_PROF_SELECT 1
#define TF_PROF_SCAN 2
#define TF_PROF_EMIT 3
#define TF_PROF_INTEGRATE 4
#define TF_PROF_FINALIZE 5
#define TF_PROF_PATCH_PROJECT 6
#define TF_PROF_ATLAS_BLIT 7
#define TF_PROF_MESH 8
#define TF_PROF_DIRTY 9
#define TF_PROF_PATCH_RANK 10
#define TF_PROF_XCHG 11        /* N > 1: pack -> transport -> unpack of one boundary exchange (tf_comm.cpp) */
#define TF_PROF_XCHG_WAIT 12   /* N > 1, overlapped exchange: what the main stream still waits for it behind the interior mesh pass */
#define TF_PROF_COUNT 13
typedef struct {
  double ms[TF_PROF_COUNT];       /* summed elapsed time per kernel */
  int64_t launches[TF_PROF_COUNT];
} tf_profile;

TF_API const char* tf_last_error(void);
TF_API int tf_device_count(void);

/* ---- lifetime / parameters ---------------------------------------------------------
 * Chisel::Chisel(chunkSize, voxelResolution, useColor)   Structure/Chisel.cpp:38-41
 *   (+ ChunkManager ctor, Atlas ctor Structure/Atlas.cpp:32-41).  chunk_dim must be 8x8x8:
 *   the reference kernel hard-codes 512-voxel chunks (ColorVoxel.h:31, Chisel.cpp:81-109). */
TF_API int tf_volume_create(const int32_t chunk_dim[3], float resolution, int use_color,
                            const tf_config* cfg, tf_volume** out);
/* tf_config grows by appended fields and is handed over by pointer: the sized form copies cfg_bytes of it and takes the
 *   defaults (0) for whatever the caller's header did not know.  C / C++ callers get it through the macro below, so a
 *   program built against an older header keeps working against a newer library; tf_volume_create itself stays exported
 *   for bindings that fill the current struct (ctypes, texturefusion_amd/capi.py). */
TF_API int tf_volume_create_sized(const int32_t chunk_dim[3], float resolution, int use_color, const tf_config* cfg,
                                  size_t cfg_bytes, tf_volume** out);
#ifndef TF_NO_SIZED_CREATE
#define tf_volume_create(chunk_dim, resolution, use_color, cfg, out) \
  tf_volume_create_sized((chunk_dim), (resolution), (use_color), (cfg), sizeof(tf_config), (out))
#endif
TF_API int tf_volume_destroy(tf_volume* v);
/* Chisel::Reset   Structure/Chisel.cpp:47-50 */
TF_API int tf_volume_reset(tf_volume* v);
/* Run on an externally owned hipStream_t (e.g. the caller framework's current stream). */
TF_API int tf_set_stream(tf_volume* v, void* hip_stream);
/* PinholeCamera::SetIntrinsics/SetWidth/SetHeight/SetNearPlane/SetFarPlane
 *   3rd_party/open_chisel/camera/PinholeCamera.h:43-57; floats are stored as given and
 *   truncated to int where the reference's int-returning getters do (:46-49). */
TF_API int tf_set_camera(tf_volume* v, float fx, float fy, float cx, float cy, int width,
                         int height, float near_plane, float far_plane);
/* QuadraticTruncator(quadratic, linear, constant, scale)  truncation/QuadraticTruncator.h:33 */
TF_API int tf_set_truncation(tf_volume* v, float quadratic, float linear, float constant,
                             float scale);
/* ConstantWeighter(weight)  weighting/ConstantWeighter.h:34 */
TF_API int tf_set_weight(tf_volume* v, float weight);

/* ---- frame images ------------------------------------------------------------------
 * The reference passes raw cv::Mat data pointers into the path (MobileFusion.cpp:147-149).
 * tf_frame_upload copies host images to HBM; tf_frame_bind_device borrows images that are
 * already device-resident (valid until the next upload/bind).  rgba / quality may be NULL. */
TF_API int tf_frame_upload(tf_volume* v, const float* depth, const uint8_t* rgba,
                           const float* quality);
/* The caller's RGBA staging loop (GCFusion/MobileFusion.cpp:144-163) on the device: rgb = u8[H][W][3]
 * (Frame::rgb), color_valid = u8[H][W] (Frame::colorValidFlag); the path sees
 * rgba = color_valid > 0 ? (r, g, b, 1) : (0, 0, 0, 0). */
TF_API int tf_frame_upload_rgb(tf_volume* v, const float* depth, const uint8_t* rgb,
                               const uint8_t* color_valid, const float* quality);
TF_API int tf_frame_bind_device(tf_volume* v, const float* d_depth, const uint8_t* d_rgba,
                                const float* d_quality);

/* ---- the path, call by call (the reference's 10-argument flow) ----------------------
 * Chisel::PrepareIntersectChunks   Structure/Chisel.h:103-140
 *   -> chunksIntersecting (out_ids), newChunkFlag (out_new); needsUpdateFlag is all false. */
TF_API int tf_prepare(tf_volume* v, const float pose[12], int32_t* out_ids, uint8_t* out_new,
                      int64_t cap, int64_t* n);
/* Chisel::IntegrateDepthScanColor (10-arg)   Structure/Chisel.h:218-249
 *   integrate_flag 1 = integrate, 0 = de-integrate; keyframe_id / out_quality carry
 *   chunk->observations[keyframeID] (host keeps the map; out_quality may be NULL).
 *   use_color 0 = the call passes colorImage == NULL (depth-only local frames). */
TF_API int tf_integrate(tf_volume* v, const float pose[12], const int32_t* ids, int64_t n,
                        int integrate_flag, int use_color, int use_quality,
                        uint8_t* inout_needs_update, float* out_quality);
/* ---- the keyframe unit (what the product's map thread runs per keyframe) -----------------------------
 * MobileFusion::tsdfFusion (GCFusion/MobileFusion.cpp:274-406) as ONE asynchronous call on device-resident images:
 *   for every MOVED keyframe (GetMapDynamics' keyframesToUpdate, <= 12): RetractObservations (:252-272), then
 *     ReIntegrateKeyframe(flag 0) (:114-221) over the keyframe's stored Frame::validChunks at the OLD poses -- the
 *     keyframe's depth + colour + quality, then its <= 6 local frames depth-only over the same list -- and
 *     ReIntegrateKeyframe(flag 1) at the NEW poses (PrepareIntersectChunks, the same frames, FinalizeIntegrateChunks,
 *     validChunks stored again);
 *   the NEW keyframe group (`fresh`, may be NULL), flag 1;
 *   Chisel::UpdateMeshes over everything marked since the last CompressMeshes (:327);
 *   texture != 0: CompressMeshes, GeneratePatches with the new keyframe as the label of every chunk of chunksToUpdate,
 *     UpdateAtlas (:355-382; pose_inv16 = f32(SE3d.inverse().matrix()) of the new keyframe, Patch.cpp:51).  With
 *     texture == 0 the unit ends behind UpdateMeshes: the caller's tf_compress_meshes returns chunksToUpdate, its view
 *     selection (host code outside the path) runs, tf_generate_patches / tf_update_atlas take its labels.
 * Visible lists, needsUpdate / new flags, every keyframe's validChunks and Chunk::observations (tf_observations_*)
 * stay in HBM; nothing is copied back and the host does not wait.  All images are device pointers (depth f32 16-B
 * aligned; rgba u8[H][W][4] = valid ? (r, g, b, 1) : 0, MobileFusion.cpp:144-163; quality f32 or NULL) that must stay
 * valid until the next synchronising call.  A moved keyframe must have been integrated through this entry point before
 * (its validChunks live in the handle).  tf_keyframe_unit_release frees the per-handle store (also done by
 * tf_volume_reset / tf_volume_destroy). */
typedef struct {
  const float* d_depth;
  const uint8_t* d_rgba;     /* keyframe only; NULL for local frames */
  const float* d_quality;    /* keyframe only; may be NULL */
  float pose[12];            /* row-major [R|t], the pose to integrate with (pose_sophus[0]) */
} tf_unit_frame;
typedef struct {
  int32_t kf_id;             /* Frame::frame_index: key of Chunk::observations, label of its patches */
  int32_t n_local;           /* 0..6 */
  tf_unit_frame keyframe;
  tf_unit_frame local[6];    /* KeyFrameDatabase::corresponding_frames with a refined depth (:189-190) */
  float old_keyframe_pose[12];  /* moved keyframes: the poses of the previous integration (pose_sophus[1]) */
  float old_local_pose[6][12];
} tf_unit_group;
TF_API int tf_keyframe_unit_device(tf_volume* v, const tf_unit_group* fresh, const tf_unit_group* moved, int32_t n_moved,
                                   int32_t texture, const float* pose_inv16);
TF_API int tf_keyframe_unit_release(tf_volume* v);
/* the store of the keyframes' validChunks: out = {capacity, entries in use (top), compactions, stores into an existing
 * region, regions handed out}.  A re-integrated keyframe writes into its region when the list fits, else it gets a new
 * one; the live regions are moved together when the top reaches the capacity.  Synchronises. */
TF_API int tf_keyframe_unit_stats(tf_volume* v, int64_t out[5]);
/* The store has no fixed ceiling: the region table starts with 1024 keyframe slots and doubles when a new keyframe needs
 * one more (the reference sizes its keyframe database for 20 000 frames, main.cpp:81); the arena doubles when the live
 * regions pass three quarters of it (one stream synchronisation + one device copy per doubling).
 * out = {slots of the table, keyframes that own one, doublings so far (table + arena), arena entries}.  Does not synchronise. */
TF_API int tf_keyframe_unit_stats_ex(tf_volume* v, int64_t out[4]);

/* ---- view-selection bookkeeping on the device ------------------------------
 * Chunk::observations (3rd_party/open_chisel/geometry/Chunk.h:171) lives in HBM, keyed by (chunk, keyframe):
 * tf_observations_record   the write of Chisel::IntegrateDepthScanColor (Structure/Chisel.h:244-247) for the list the
 *                          last tf_integrate worked on: observations[keyframe_id] = quality where quality > 0 and the
 *                          chunk's needsUpdateFlag is set (keyframe_id < 0: nothing, as in the reference)
 * tf_observations_retract  MobileFusion::RetractObservations' chunk side (GCFusion/MobileFusion.cpp:252-260):
 *                          observations.erase(keyframe_id) for the listed chunks that exist
 * tf_export_datacost       what TexMap::update_datacost (Structure/TexMap.cpp:64-105) reads, as one table: for chunk i of
 *                          chunksToUpdate out[i * (1 + n_frames) + 0] = observations[frame_index], [1 + j] =
 *                          observations[frames_to_update[j]]; 0 = no observation (recorded qualities are > 0)
 * tf_export_adjacency      the edges TexMap::update_chunkgraph (Structure/TexMap.cpp:50-62) / UniGraph::add_edge_by_node
 *                          (uni_graph.cpp:41-49) add: one record int32[4] = {i, neighbour id} per set Mesh::adj flag of
 *                          chunk ids[i] whose face neighbour (chisel::neighbourhood, ChunkManager.h:55-57) owns a mesh
 *                          (the caller keeps the graph and drops neighbours that are not nodes); order arbitrary */
TF_API int tf_observations_record(tf_volume* v, int32_t keyframe_id);
TF_API int tf_observations_retract(tf_volume* v, int32_t keyframe_id, const int32_t* ids, int64_t n);
TF_API int tf_export_datacost(tf_volume* v, const int32_t* ids, int64_t n, int32_t frame_index,
                              const int32_t* frames_to_update, int32_t n_frames, float* out);
TF_API int tf_export_adjacency(tf_volume* v, const int32_t* ids, int64_t n, int32_t* out_edges, int64_t cap_edges,
                               int64_t* n_edges);
/* The local frames of a keyframe group in one visit per chunk (GCFusion/MobileFusion.cpp:187-203: after the keyframe's
 * own IntegrateDepthScanColor, its corresponding frames are integrated depth-only over the SAME chunk list, each with
 * its own pose).  Equivalent, bit for bit, to n_frames successive tf_integrate(use_color = 0) calls with these depth
 * images (device pointers, 16-B aligned) and poses (n_frames x 12 floats); the voxel rows are read and written once
 * instead of n_frames times.  n_frames <= 6 (integrateLocalFrameNum, main.cpp:91). */
TF_API int tf_integrate_depth_group(tf_volume* v, int32_t n_frames, const float* const* d_depth, const float* poses12,
                                    const int32_t* ids, int64_t n, int integrate_flag, uint8_t* inout_needs_update);
/* the same with host depth images (borrowed for the call; staged into a device buffer the handle keeps) */
TF_API int tf_integrate_depth_group_host(tf_volume* v, int32_t n_frames, const float* const* depth, const float* poses12,
                                         const int32_t* ids, int64_t n, int integrate_flag, uint8_t* inout_needs_update);
/* Chisel::FinalizeIntegrateChunks + GarbageCollect   Structure/Chisel.h:184-216,472-477
 *   out_valid receives validChunks (may be NULL). */
TF_API int tf_finalize(tf_volume* v, const int32_t* ids, const uint8_t* needs_update,
                       const uint8_t* is_new, int64_t n, int32_t* out_valid, int64_t* n_valid);

/* ---- the path, fused (the per-frame benchmark unit) ---------------------------------
 * Chisel::IntegrateDepthScanColor (5-arg)   Structure/Chisel.h:453-468, as driven by
 * MobileFusion::IntegrateFrame (GCFusion/MobileFusion.cpp:223-250): prepare -> integrate
 * (flag 1, no keyframe id, no quality) -> finalize on the bound frame.  Asynchronous: no
 * host synchronisation, nothing is copied back.  use_color 0 = rgb.empty() branch. */
TF_API int tf_integrate_frame(tf_volume* v, const float pose[12], int use_color);
/* Same, for a batch of device-resident frames (arrays of n device pointers / n poses). */
TF_API int tf_integrate_frames_device(tf_volume* v, int64_t n_frames, const float* const* d_depth,
                                      const uint8_t* const* d_rgba, const float* poses12);
/* The same for a stream that is fed call by call: the arrays hold n_ahead (0..2) frames more than the
 * n_frames that are integrated; those run through their selection stages (bounding box, visible-chunk
 * list) in this call's launches and the next call, if it starts with exactly these frames (same depth
 * pointer and pose), begins with the voxel update at once -- one launch per frame across calls. */
TF_API int tf_stream_frames_device(tf_volume* v, int64_t n_frames, int64_t n_ahead, const float* const* d_depth,
                                   const uint8_t* const* d_rgba, const float* poses12);
/* The per-frame unit with texturing (BASELINE configs[2]): after frame f is
 * integrated its dirty chunks -- the updated chunks and their six face neighbours, Chisel.h:192-208 -- run
 * through Chisel::UpdateMeshes, CompressMeshes, GeneratePatches with label = frame f and UpdateAtlas
 * (GCFusion/MobileFusion.cpp:327-382 without the host-side view selection), all on the device and without
 * a host synchronisation.  The keyframe of a patch is the frame itself: its RGBA image (alpha ignored) and
 * depth, pose_inv16[f] = f32(SE3d.inverse().matrix()) of its pose; Patch::frameid = first_frame_id + f.
 * New patches take their atlas slots in ascending (x, y, z) chunk-id order within a frame (the reference's
 * order is an unordered_map's).  Chisel::meshesToUpdate collects marks until CompressMeshes clears it: a frame
 * that follows frames integrated WITHOUT this unit (tf_stream_frames_device, tf_integrate ...) inherits their
 * marks and meshes / textures everything they touched as well.
 * IMAGE LIFETIME: with n_ahead == 0 everything that reads the images is on the stream when the call returns.  With
 * n_ahead > 0 the caller declares that it keeps feeding this stream, and the texture stage of the LAST integrated frame
 * (which reads d_rgba[n_frames - 1] and d_depth[n_frames - 1]) is not launched yet: it rides on the next call's first
 * launch.  Those two images must therefore stay valid and unmodified until the next tf_* call on the handle has
 * returned -- a device-wide synchronisation (hipDeviceSynchronize, torch.cuda.synchronize) does NOT cover it, tf_sync
 * does (every non-streaming entry point puts the pending stage on the stream first).  A ring of image buffers needs
 * n_ahead + 2 slots. */
TF_API int tf_stream_frames_textured_device(tf_volume* v, int64_t n_frames, int64_t n_ahead,
                                            const float* const* d_depth, const uint8_t* const* d_rgba,
                                            const float* poses12, const float* pose_inv16, int32_t first_frame_id);
/* MobileFusion::IntegrateFrame (GCFusion/MobileFusion.cpp:223-250) as the reference calls it: HOST images in,
 * one call per frame.  The images go through a ring of eight pinned staging / device slots and are uploaded
 * on a second stream, so the H2D of a frame overlaps the kernels of earlier ones and the call returns without
 * synchronising (tf_sync / any state access waits).  Internally the call for frame f launches the integration of
 * frame f - 4 together with the chunk selection of f - 3 and f - 2 (the launch pipeline of the streaming entry points,
 * kept alive across calls; frames f - 1 and f are only staged and copied, so that no launch ever waits for an upload);
 * every other entry point first integrates the frames still in that pipeline, so the deferral is not observable
 * through this API -- but it is LATENCY: a live caller sees frame f in the volume only after four more calls (about
 * 0.4 ms at the bench's rate) or after any synchronising call; the offline loop of main.cpp:272-277 does not care, a
 * caller that needs every frame at once switches the deferral off for its handle with tf_host_frame_set_deferral(v, 0)
 * (integrate in the call that brings the frame; TF_HOST_DEFER=0 in the environment makes that the default of new handles).
 * tf_host_frame_deferral reports both numbers (frames behind, ring slots; v may be NULL).  pose_inv16 != NULL runs the textured unit
 * (tf_stream_frames_textured_device's per-frame work) with Patch::frameid = frame_id; NULL = TSDF only.
 * tf_host_frame_buffers hands out the pinned slot the NEXT call will upload from: a caller that composes its
 * depth / RGBA images there (and passes these pointers) saves the staging copy. */
TF_API int tf_integrate_frame_host(tf_volume* v, const float* depth, const uint8_t* rgba, const float pose[12],
                                   const float* pose_inv16, int32_t frame_id);
/* The same with the colour image as the caller holds it: rgb = u8[H][W][3] (Frame::rgb) and color_valid = u8[H][W]
 * (Frame::colorValidFlag) or NULL (every pixel valid) -- the inputs of the RGBA staging loops the reference runs on the CPU
 * before it calls the path (MobileFusion::IntegrateFrame, GCFusion/MobileFusion.cpp:232-243: alpha = 1 everywhere;
 * :144-163: alpha = colorValid > 0).  7 (or 8) bytes per pixel travel instead of 8 and the loop runs on the device behind
 * the upload.  With tf_host_frame_buffers: rgb at the `rgba` pointer, the flags 3 * W * H bytes behind it. */
TF_API int tf_integrate_frame_host_rgb(tf_volume* v, const float* depth, const uint8_t* rgb, const uint8_t* color_valid,
                                       const float pose[12], const float* pose_inv16, int32_t frame_id);
TF_API int tf_host_frame_buffers(tf_volume* v, float** depth, uint8_t** rgba);
/* A caller that keeps its images in buffers of its own (cv::Mat data, a camera ring) registers them once: the pages are
 * locked in place (hipHostRegister) and host frames that lie inside a registered buffer are uploaded straight out of it --
 * no staging copy, no helper threads -- while the call puts the launches of the frames before on the stream; the call
 * returns when the upload is through, so the buffer is the caller's again on return, as with the staging path.  Registering
 * costs about a millisecond per megabyte: once per buffer, not per frame.  tf_volume_destroy unregisters what is left. */
TF_API int tf_host_register(tf_volume* v, const void* p, int64_t bytes);
TF_API int tf_host_unregister(tf_volume* v, const void* p);
TF_API int tf_host_frame_deferral(tf_volume* v, int32_t* frames_behind, int32_t* ring_slots);
TF_API int tf_host_frame_set_deferral(tf_volume* v, int on);
/* A caller with a RING of registered frame buffers need not wait for every upload: tf_host_frame_set_async(v, 1) lets a call
 * out of registered buffers return as soon as its upload is queued (the reference's contract -- "the buffer is mine again
 * when the call returns", MobileFusion.cpp:249 -- is then the caller's to keep); tf_host_frame_fence(v) returns when every
 * upload queued so far is through, i.e. every buffer handed over so far may be written again.  The upload of frame f then
 * overlaps the call for f + 1: a TSDF-only stream runs at the link's rate (2.46 MB per 640x480 frame at 46 GB/s = 53 us,
 * tools/h2d_probe.py) instead of upload + call time.  Default off. */
TF_API int tf_host_frame_set_async(tf_volume* v, int on);
TF_API int tf_host_frame_fence(tf_volume* v);
/* Where the host side of tf_integrate_frame_host(_rgb) spends its time, summed since create / the last reset:
 * out[0] = calls that put a frame's launches on the stream, out[1..5] = microseconds spent waiting for the device to free
 * a staging slot (back-pressure: the device is the bound), waiting for the slot's previous upload, in the staging copy, in
 * the upload enqueue, in the kernel launches; out[6] = launches that had to wait in the stream for an upload. */
TF_API int tf_host_frame_times(tf_volume* v, double out[7], int reset);
/* The texturing half of the per-frame unit on its own, for the frame integrated last (its images still bound):
 * UpdateMeshes -> CompressMeshes -> GeneratePatches(label = frame_id) -> UpdateAtlas over that frame's dirty
 * chunks.  tf_stream_frames_textured_device == per frame: voxel update, then this.  A multi-GPU host that
 * brings its own transport calls tf_boundary_pack_block / its all-gather / tf_boundary_unpack_blocks(join_dirty)
 * in between. */
TF_API int tf_texture_frame_device(tf_volume* v, const float pose_inv16[16], int32_t frame_id);
/* The same in two halves around the caller's own boundary exchange, so that the exchange overlaps work: phase 1 builds the
 * frame's dirty set and meshes its INTERIOR chunks (those whose 27-chunk neighbourhood this rank owns: nothing they read
 * comes from another rank); phase 2 meshes the boundary chunks and everything the arriving ghosts added, and leaves the
 * patch stage pending as tf_texture_frame_device does.  Order for the caller's transport:
 *   - tf_boundary_unpack_*(join_dirty) must run BEHIND phase 1 and ahead of phase 2: phase 1's filter walks the flat work
 *     list of the frame's parity, the list join_dirty appends to (the library's own overlapped exchange hands its interior
 *     pass an empty flat list instead; this entry point does not).  The tf_boundary_unpack_* entry points launch on the
 *     handle's stream, so calling them after phase 1 gives that order;
 *   - what may overlap phase 1 is the PACK and the TRANSPORT: issue tf_boundary_pack_* before phase 1 (it reads voxels the
 *     frame's update has finished writing and nothing phase 1 writes) and run the transport on the caller's own stream or
 *     thread while phase 1's launches execute; a pack issued after phase 1 sits behind it on the handle's stream and
 *     overlaps nothing.
 * Results equal the one-call form bit for bit.  (With tf_comm_exchange_every_frame the library does this itself, pack /
 * send / receive / unpack on a second stream next to the interior pass: tf_comm_exchange_overlap(v, 0) switches that off.) */
TF_API int tf_texture_frame_device_phase(tf_volume* v, const float pose_inv16[16], int32_t frame_id, int phase);
TF_API int tf_comm_exchange_overlap(tf_volume* v, int on);
TF_API int tf_sync(tf_volume* v);

/* ---- state access (host mirrors of Chunk::voxels / colors, ChunkManager queries) -----
 * ChunkManager::HasChunk   Structure/ChunkManager.h:133-135 */
TF_API int tf_has_chunk(tf_volume* v, const int32_t id[3], int* out);
/* Chunk::voxels.sdf / .weight (DistVoxel.h:102-103), Chunk::colors.colorData
 *   (ColorVoxel.h:66) in the reference's layouts: sdf[512], weight[512], color[512*4]. */
TF_API int tf_chunk_download(tf_volume* v, const int32_t id[3], float* sdf, float* weight,
                             uint16_t* color);
TF_API int tf_chunks_download(tf_volume* v, const int32_t* ids, int64_t n, float* sdf,
                              float* weight, uint16_t* color);
/* creates the chunk if needed (ChunkManager::CreateChunk, ChunkManager.cpp:266-270) */
TF_API int tf_chunk_upload(tf_volume* v, const int32_t id[3], const float* sdf,
                           const float* weight, const uint16_t* color);
/* ChunkManager::GetChunks() keys */
TF_API int tf_list_chunks(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n);
/* Chisel::meshesToUpdate (Structure/Chisel.h:489): keys with value true */
TF_API int tf_list_dirty(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n);
TF_API int tf_clear_dirty(tf_volume* v);
TF_API int tf_get_stats(tf_volume* v, tf_stats* out);
TF_API int tf_get_texture_stats(tf_volume* v, tf_texture_stats* out);

/* ---- meshing (the stage between the volume and the atlas) -----
 * Chisel::UpdateMeshes (Structure/Chisel.h:479-481) -> ChunkManager::RecomputeMeshes
 *   (Structure/ChunkManager.cpp:232-264): every chunk of meshesToUpdate that exists is re-meshed by
 *   GenerateMeshEfficient (:595-1002, gradient normals :277-455) on the device; meshes stay in HBM
 *   (ChunkManager::allMeshes).  *n_meshed = size of the dirty set handed to the mesher. */
TF_API int tf_update_meshes(tf_volume* v, int64_t* n_meshed);
/* Diagnostic of the filter ahead of the mesher.  The library keeps per chunk a 16-bit summary of the classes
 *   (observed / positive / negative / weight > 50, i.e. what GenerateMeshEfficient's per-cell tests at
 *   ChunkManager.cpp:669-722 and :776-777 ask for) that occur among its voxels and on its three low faces; a chunk
 *   whose summaries rule out a vertex is not read at all.  A summary must be a superset of the truth: *n_missing =
 *   chunks whose voxels hold a class the summary lacks (must be 0), *n_stale = chunks whose summary holds a class the
 *   voxels no longer do (allowed: costs time, never changes a result).  Any output may be NULL. */
TF_API int tf_check_summaries(tf_volume* v, int64_t* n_chunks, int64_t* n_missing, int64_t* n_stale);
/* Diagnostic of the neighbour table the filter and the patch stage read instead of probing the chunk hash (what
 *   ChunkManager::GenerateMeshEfficient resolves per call with GetChunk on the neighbour ids, Structure/ChunkManager.cpp:618-632,
 *   and Chisel::CompressMeshes with allMeshes.find, Structure/Chisel.cpp:134-137): per pool slot the pool slots of the 26
 *   chunks around it, filled lazily.  out6 = {rows, non-zero words, non-zero words that disagree with the hash (must be 0),
 *   rows whose "no chunk there" words are currently trusted, trusted "none" words whose chunk exists (must be 0), 0}. */
TF_API int tf_check_neighbours(tf_volume* v, int64_t out6[6]);
/* keys of ChunkManager::GetAllMeshes() (Structure/ChunkManager.h:714) */
TF_API int tf_list_meshes(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n);
/* Mesh::vertices.size() / indices.size() / adj[6] / simplified of listed chunks (Mesh.h:70-85);
 *   TF_ERR_MISSING_CHUNK when a chunk has no mesh (allMeshes.at() throws).  Outputs may be NULL. */
TF_API int tf_mesh_counts(tf_volume* v, const int32_t* ids, int64_t n, int32_t* n_vertices,
                          int32_t* n_indices, uint8_t* adj, uint8_t* simplified);
/* Mesh::vertices / normals / colors (Vec3List: 3 f32 per vertex) and Mesh::indices of listed chunks,
 *   packed by the caller's running offsets (vert_offsets[n+1], index_offsets[n+1] from tf_mesh_counts). */
TF_API int tf_meshes_download(tf_volume* v, const int32_t* ids, int64_t n, const int64_t* vert_offsets,
                              const int64_t* index_offsets, float* verts, float* normals, float* colors,
                              uint32_t* indices);
/* Chisel::CompressMeshes(meshesToUpdate) (Structure/Chisel.cpp:112-147): Mesh::SimplifyByClustering's
 *   adjacency flags (geometry/Mesh.cpp:39-83) exchanged with the face neighbours, meshesToUpdate
 *   cleared.  out_ids receives tsdfFusion's chunksToUpdate -- the dirty keys that have a mesh
 *   (GCFusion/MobileFusion.cpp:345-353) -- in ascending (x, y, z) order (the reference's order is its
 *   unordered_map's iteration order, i.e. unspecified). */
TF_API int tf_compress_meshes(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n);

/* ---- measurement ------------------------------------------------------------------- */
/* kind_mask: bit k set = bracket every launch of kernel kind TF_PROF_k with a pair of HIP events
 * on the handle's stream (0 = off).  Timing only the dominant kernel keeps the event overhead
 * out of the other launches. */
TF_API int tf_profile_enable(tf_volume* v, uint32_t kind_mask);
TF_API int tf_profile_get(tf_volume* v, tf_profile* out, int reset);
/* What a HIP-event pair around an EMPTY launch reads on the handle's stream (microseconds, mean of n_pairs): the
 *   floor contained in every per-kernel time of tf_profile_get (launch + marker latency, not kernel work). */
TF_API int tf_profile_calibrate(tf_volume* v, int32_t n_pairs, double* us_per_pair);
/* Tuning aid: the raw per-wave timeline table (16 words per wave; words 10..13 = {start, end, role+1,
 * XCC id}, 14..15 = K-A prologue end / first list record loaded, in 100 MHz ticks) of the last
 * fused launch that ran all three roles; filled only while the environment variable TF_KA_DBG has
 * bit 12 (4096) set. */
TF_API int tf_debug_phase_raw(tf_volume* v, uint64_t* out, int64_t cap_words);

/* ---- multi-GPU chunk-range partition --------------------------------
 * A rank owns chunks with lo <= id.x < hi; selection runs in full on every rank, integrate /
 * finalize touch only owned chunks.  The chunks of the slab's ghost band that were updated since
 * the previous tf_boundary_pack are packed into / unpacked from a device buffer that is
 * all-gathered over RCCL.  Ghost band = what another rank's mesher reads of this slab: the layer
 * key == hi-1, and the layers lo <= key <= lo + (a+b+c) -- GenerateMeshEfficient reads c + {0,1}^3
 * (Structure/ChunkManager.cpp:618-632) and extractGradientFromCubic the face neighbours of those
 * (:288-315).  Record = 16 B header {x,y,z,update epoch} + 4 KiB {sdf,weight}[512] + 4 KiB colour[512][4].
 * A record that does not fit the buffer stays flagged (TF_ERR_CAPACITY; call again with more room). */
#define TF_BOUNDARY_RECORD_BYTES (16 + 4096 + 4096)
TF_API int tf_set_partition(tf_volume* v, int32_t x_lo, int32_t x_hi);
/* The same with the ownership key a*x + b*y + c*z (a, b, c in {0, 1}) instead of x: a rank owns
 * key_lo <= key < key_hi.  (1, 1, 1) cuts axis-aligned walls and floors diagonally, so no single
 * rank holds a whole wall (its ghost band is four layers thick on the lower side). */
TF_API int tf_set_partition_key(tf_volume* v, int32_t a, int32_t b, int32_t c, int32_t key_lo, int32_t key_hi);
TF_API int tf_boundary_pack(tf_volume* v, void* d_records, int64_t cap_records, int64_t* n);
/* The same without a host round trip: the record count (which may exceed cap_records; only the first
 * cap_records were written) lands in the device word *d_count, ordered on the handle's stream, so
 * that the exchange of one frame batch can overlap the integration of the next. */
TF_API int tf_boundary_pack_async(tf_volume* v, void* d_records, int64_t cap_records, uint32_t* d_count);
TF_API int tf_boundary_unpack(tf_volume* v, const void* d_records, int64_t n_records);
/* Fixed-capacity form for a single collective without a host round trip: a block is
 *   [u32 record count, 12 B padding | cap_records records]   (tf_boundary_block_bytes(cap) bytes);
 * tf_boundary_pack_block fills this rank's block (the count may exceed cap_records: the surplus stays flagged
 * for the next exchange and the receivers report TF_ERR_CAPACITY at their next synchronising call);
 * tf_boundary_unpack_blocks consumes the all-gathered blocks of every rank but own_block; join_dirty != 0
 * (between the voxel update of a frame and tf_texture_frame_device): the owned face neighbours of every ghost
 * that arrives join that frame's dirty set -- their neighbour was updated on another rank.  Asynchronous. */
TF_API size_t tf_boundary_block_bytes(int64_t cap_records);
TF_API int tf_boundary_pack_block(tf_volume* v, void* d_block, int64_t cap_records);
TF_API int tf_boundary_unpack_blocks(tf_volume* v, const void* d_blocks, int32_t n_blocks, int32_t own_block,
                                     int64_t cap_records, int join_dirty);
/* Two blocks instead of one, for a neighbour-only exchange: slabs are contiguous key ranges, so the chunks the rank
 * BELOW reads as ghosts are those with key - key_lo <= a + b + c (d_block_down) and the rank ABOVE reads key == key_hi - 1
 * (d_block_up); a chunk of a thin slab goes to both.  Valid as the only exchange when every slab is at least
 * a + b + c + 1 keys wide.  Consume a neighbour's block with tf_boundary_unpack_blocks(..., own_block = -1, ...). */
TF_API int tf_boundary_pack_bands(tf_volume* v, void* d_block_down, void* d_block_up, int64_t cap_records);
/* The SIZED form of the neighbour exchange: blocks as large as what the frame can have changed, not a fixed capacity.
 * Every rank runs the same selection, so each one counts -- identically -- the selected chunks of its own two bands and of
 * the two neighbouring bands it receives (the selection role does it on the device for the frame integrated last by a
 * streaming entry point).  tf_boundary_band_bounds returns the four record capacities
 *     bounds = { send_down, send_up, recv_from_below, recv_from_above }
 * (counts rounded up to multiples of 8, at least 8, at most cap_records): rank r's send_down equals rank r - 1's
 * recv_from_above by construction, so both sides of a transfer post tf_boundary_block_bytes(bound) bytes without talking
 * to each other.  Selected chunks are a superset of updated ones, so nothing a frame flagged is left behind; whatever an
 * EARLIER overflow left flagged rides in the slack or waits (it stays flagged).  tf_boundary_band_bounds waits for the
 * device to publish the counts (it does not drain the stream).  tf_boundary_pack_bands2 / tf_boundary_unpack_pair are
 * pack_bands / unpack_blocks with one capacity per block and separately placed blocks.
 * Replaces nothing in the reference (single process); SURVEY.md s.8e. */
TF_API int tf_boundary_band_bounds(tf_volume* v, int64_t cap_records, int64_t bounds[4]);
TF_API int tf_boundary_pack_bands2(tf_volume* v, void* d_block_down, int64_t cap_down, void* d_block_up, int64_t cap_up);
TF_API int tf_boundary_unpack_pair(tf_volume* v, const void* d_from_below, int64_t cap_below, const void* d_from_above,
                                   int64_t cap_above, int join_dirty);
/* RCCL inside the library: one process per GPU.  tf_comm_unique_id on one rank, the 128 bytes
 * distributed by the host's own means, tf_comm_init(rank, nranks, id) on every rank (ncclCommInitRank);
 * tf_exchange_boundary = tf_boundary_pack_block -> ONE ncclAllGather over xGMI on the handle's stream ->
 * tf_boundary_unpack_blocks, nothing copied to the host.  tf_comm_exchange_every_frame(cap): the textured
 * per-frame flow (tf_stream_frames_textured_device / tf_integrate_frame_host) then exchanges after every
 * voxel update, ahead of the mesher; owned face neighbours of arriving ghost chunks join the frame's dirty set. */
TF_API int tf_comm_unique_id(void* out128);
TF_API int tf_comm_init(tf_volume* v, int rank, int nranks, const void* unique_id128);
TF_API int tf_comm_destroy(tf_volume* v);
TF_API int tf_exchange_boundary(tf_volume* v, int64_t cap_records);
TF_API int tf_comm_exchange_every_frame(tf_volume* v, int64_t cap_records);
/* How tf_exchange_boundary / the per-frame exchange move the blocks: TF_XCHG_NEIGHBOURS (default) = one grouped
 * ncclSend / ncclRecv pair with the rank below and the rank above (tf_boundary_pack_bands: a rank receives two blocks
 * whatever the number of ranks); TF_XCHG_ALLGATHER = one ncclAllGather of every rank's block.  The neighbour form
 * REQUIRES rank r's slab to sit directly above rank r - 1's (key_lo of r == key_hi of r - 1, the same key coefficients on
 * every rank) and every slab above the lowest to be at least a + b + c + 1 keys wide; the library checks this with one
 * 32-byte all-gather at the first exchange after tf_comm_init / tf_set_partition_key (a collective: every rank gets
 * there) and uses the all-gather form on ALL ranks when it does not hold.  The per-frame exchange of the neighbour form
 * is sized by the frame's selection (tf_boundary_band_bounds above); cap_records is then only the upper limit.
 * tf_comm_stats: exchanges run and bytes received by this rank so far.  tf_comm_stats_ex (synchronises): out =
 * { exchanges, bytes sent, bytes received, ghost records written into blocks, ghost records read from received blocks,
 *   record capacity of the blocks sent, the form in use (TF_XCHG_*), bit 0: the partition check has run | (exchanges that
 *   ran on the library's second stream next to an interior mesh pass) << 1 }. */
#define TF_XCHG_NEIGHBOURS 0
#define TF_XCHG_ALLGATHER 1
TF_API int tf_comm_exchange_mode(tf_volume* v, int mode);
TF_API int tf_comm_stats(tf_volume* v, int64_t* exchanges, int64_t* bytes_received);
TF_API int tf_comm_stats_ex(tf_volume* v, int64_t out[8]);

/* ---- texture atlas on device-resident meshes --------------------------------------------
 * Atlas / Patch (Structure/Atlas.{h,cpp}, Structure/Patch.{h,cpp}) as driven by Chisel::GeneratePatches /
 * CompensateColor / UpdateAtlas / DrawMeshes (Structure/Chisel.cpp:149-355).  Meshes (ChunkManager::allMeshes)
 * and their patches (Mesh::m_patch: texloc, frameid, boundingbox, ratio, texcoord, texcolor, labs) stay in
 * HBM; the host sees them through the *_download mirrors.
 *
 * tf_keyframe_cache: the reference keeps Frame::rgb / refined_depth alive and Patch::SetImage holds a
 *   non-owning ROI into it (Patch.cpp:172-175); here the keyframe's images are cached in HBM under kf_id
 *   (= the frame id view selection hands out as label).  rgb = u8[H][W][3], depth = f32[H][W].
 *   _device borrows images that are already resident (rgb_pixel_stride 3, or 4 for an RGBA image whose
 *   alpha is ignored).  tf_keyframe_set_pose: f32(SE3d.inverse().matrix()) of Frame::pose_sophus[0],
 *   16 floats row-major (Patch.cpp:51) -- poses move with every bundle adjustment, so it is set before
 *   GeneratePatches. */
TF_API int tf_keyframe_cache(tf_volume* v, int32_t kf_id, const uint8_t* rgb, const float* depth);
TF_API int tf_keyframe_cache_device(tf_volume* v, int32_t kf_id, const uint8_t* d_rgb, int32_t rgb_pixel_stride,
                                    const float* d_depth);
TF_API int tf_keyframe_set_pose(tf_volume* v, int32_t kf_id, const float pose_inv16[16]);
TF_API int tf_keyframe_release(tf_volume* v, int32_t kf_id);
/* Atlas::SetResolution  Structure/Atlas.h:62-65 */
TF_API int tf_atlas_patch_size(tf_volume* v, int32_t* patch_w, int32_t* patch_h);
/* Atlas::loc_next */
TF_API int tf_atlas_loc_next(tf_volume* v, uint64_t* loc_next);
/* MAX_PATCH_WIDTH x MAX_PATCH_HEIGHT of this volume's texture_buffer (Structure/Atlas.h:29-30; tf_config.atlas_w / _h) */
TF_API int tf_atlas_size(tf_volume* v, int32_t* atlas_w, int32_t* atlas_h);
/* ChunkManager::allMeshes[id] = a mesh built by the caller (a host that keeps its own mesher): Mesh::vertices /
 *   normals / colors as 3 f32 per vertex, Mesh::indices, packed by running offsets; creates missing chunks. */
TF_API int tf_meshes_upload(tf_volume* v, const int32_t* ids, int64_t n, const int64_t* vert_offsets,
                            const int64_t* index_offsets, const float* verts, const float* normals,
                            const float* colors, const uint32_t* indices);
/* Chisel::GeneratePatches(chunksToUpdate, labelset, frame_list, camera)  Structure/Chisel.cpp:149-189: for the
 *   listed chunks that have a mesh, in list order: Atlas::AddPatch (first call hands out the next slot,
 *   Atlas.cpp:43-64; later calls keep it, Patch::clear) -> Patch::CalculateTexCoords in keyframe labels[i]
 *   (Patch.cpp:40-108) -> SetFrameid / SetImage.  out_hot = atlas.hot_start / hot_end.
 *   Returns TF_ERR_ATLAS_FULL when the atlas overflows (GeneratePatches' -1): the entry that did not get a
 *   slot and everything behind it in the list stays unprocessed. */
TF_API int tf_generate_patches(tf_volume* v, const int32_t* ids, int64_t n, const int32_t* labels,
                               uint64_t out_hot[2]);
/* Chisel::CompensateColor()  Structure/Chisel.cpp:198-286 (+ computeMeanAndCov, Structure/Patch.cpp:342-348)
 *   over every mesh with a patch, in ascending chunk-id order (the reference iterates an unordered_map).
 *   Patches with has_adjusted are skipped; the rest is clustered by frame id (cluster order = first
 *   appearance).  Per cluster: mean / covariance of texcolor and of the mesh colours over the patches without
 *   wrong_mapping, the 3x3 transfer T, labs[k] = T (texcolor[k] - mean_src) + mean_tar, has_adjusted = 1.
 *   Reductions and the per-vertex transfer run on the device, the 3x3 eigen-decompositions on the host (f64
 *   Jacobi; Eigen's own iteration is not restated, the result agrees to float rounding). */
TF_API int tf_compensate_color(tf_volume* v, int64_t* out_n_clusters);
/* Chisel::UpdateAtlas(chunksToUpdate)  Structure/Chisel.cpp:191-196 -> Atlas::UpdateBuffer (Atlas.cpp:71-91):
 *   the keyframe ROI of every listed complete() patch is copied -- or cv::resize'd when it exceeds the slot --
 *   into the atlas. */
TF_API int tf_update_atlas(tf_volume* v, const int32_t* ids, int64_t n);
/* Chisel::DrawMeshes  Structure/Chisel.cpp:288-355: the interleaved vertex stream the renderer / exporter
 *   consumes, 12 f32 per vertex
 *     [x, y, z, 50, (float)(R<<16|G<<8|B), adj, u/atlas_w, v/atlas_h, nx, ny, nz, wrong_mapping]
 *   with (u,v) = texcoord * (ratio < 1 ? ratio : 1) + slot origin (Atlas::GetTexLoc) and
 *   adj = has_adjusted && !labs.empty() ? (float)(3 x 9 bit of int((labs - texcolor) * 255) + 255) : 0, plus
 *   the index stream rebased by the running vertex count, over every mesh whose patch is complete()
 *   (Patch.cpp:191-196), in ascending chunk-id order.  _device leaves both streams in HBM (e.g. a mapped GL
 *   buffer, MobileFusion.h:404-446). */
TF_API int tf_draw_meshes(tf_volume* v, float* vertices, uint32_t* indices, int64_t cap_vertices,
                          int64_t cap_indices, int64_t* n_vertices, int64_t* n_indices);
TF_API int tf_draw_meshes_device(tf_volume* v, float* d_vertices, uint32_t* d_indices, int64_t cap_vertices,
                                 int64_t cap_indices, int64_t* n_vertices, int64_t* n_indices);
/* Patch mirrors of listed chunks (Structure/Patch.h:51-94): texloc (~0 = no slot), frameid, boundingbox
 *   (x, y, w, h), flags (TF_PATCH_*), ratio; texcoord f32[2 nv], texcolor / labs f32[3 nv] packed by
 *   vert_offsets (from tf_mesh_counts).  Any output may be NULL. */
#define TF_PATCH_HAS_PATCH 1      /* Mesh::m_patch != nullptr */
#define TF_PATCH_CAUTION 2        /* CalculateTexCoords returned -1 */
#define TF_PATCH_WRONG_MAPPING 4  /* Patch::wrong_mapping */
#define TF_PATCH_HAS_IMAGE 8      /* Patch::has_image */
#define TF_PATCH_HAS_ADJUSTED 16  /* Patch::has_adjusted */
TF_API int tf_patches_download(tf_volume* v, const int32_t* ids, int64_t n, const int64_t* vert_offsets,
                               uint64_t* texloc, int32_t* frameid, int32_t* bbox, int32_t* flags, float* ratio,
                               float* texcoord, float* texcolor, float* labs);
/* Atlas::texture_buffer rows [row0,row1) (MobileFusion.h:406-421 uploads the hot rows) */
TF_API int tf_atlas_download_rows(tf_volume* v, int64_t row0, int64_t row1, uint8_t* dst);
/* The same rows for a thread OTHER than the one that drives the handle -- the reference's GUI thread reads
 *   atlas.texture_buffer while the map thread writes it (GCFusion/MobileFusion.h:404-421; SURVEY.md s.8(b) "Threading").
 *   tf_atlas_download_rows belongs to the driving thread: like every entry point it first brings the handle's deferred work
 *   onto the stream.  This one touches no state of the pipeline and may run at any time next to the driving thread's calls:
 *   the rows are copied device-to-device INSIDE the handle's stream, between two of the driving thread's launches, so they
 *   show ONE moment of the stream -- every atlas write enqueued before it, none enqueued after it -- and travel to dst on a
 *   stream of the reader's own.  *write_seq = atlas-writing launches ahead of the snapshot (monotonic), *frame_id = the
 *   label (Patch::frameid) of the newest fused frame among them, -1 = none yet; both may be NULL.  Concurrent readers are
 *   serialised. */
TF_API int tf_atlas_snapshot_rows(tf_volume* v, int64_t row0, int64_t row1, uint8_t* dst, int64_t* write_seq, int32_t* frame_id);

/* ---- frame pre-processing that feeds the path ---------------------------------
 * The per-frame image passes main.cpp:117-147 runs before a frame reaches the fusion path, on images resident in
 * device memory (row-major f32 depth / weight / quality, PLANAR f32 normal maps [3][H][W], packed u8 RGB), with the
 * untruncated intrinsics given to tf_set_camera (the reference passes camera.c_fx ...).  Asynchronous on the
 * handle's stream except tf_pre_refine_keyframe.  Pixels the reference leaves uninitialised are written as 0;
 * _mm256_rsqrt_ps is the correctly rounded 1 / sqrt (see oracle/tf_oracle.c).
 *   tf_pre_frame_depth           DatasetWrapper::framePreprocess     Tools/DatasetWrapper.hpp:186-263 (see below)
 *   tf_pre_normal_map            BasicAPI::extractNormalMapSIMD      BasicAPI.cpp:849-905
 *   tf_pre_refine_depth_normal   BasicAPI::refineDepthUseNormalSIMD  BasicAPI.cpp:728-781   (normal, depth in place)
 *   tf_pre_color_valid           BasicAPI::checkColorQuality         BasicAPI.cpp:783-806   (Frame::colorValidFlag)
 *   tf_pre_color_quality         BasicAPI::estimateColorQuality      BasicAPI.cpp:815-847   (Frame::observationQualityMap)
 *   tf_pre_refine_newframe       BasicAPI::refineNewframesSIMD       BasicAPI.cpp:378-443   (depth_new in place;
 *                                T = f32 of (pose_ref^-1 * pose_new).matrix()[3x4], row-major)
 *   tf_pre_refine_keyframe       BasicAPI::refineKeyframesSIMD       BasicAPI.cpp:506-636   (depth_ref, weight_ref in
 *                                place with the reference's sequential in-place semantics; T = f32 of
 *                                (pose_new^-1 * pose_ref).matrix()[3x4]; synchronises; *rounds = passes it took) */
/* DatasetWrapper::framePreprocess (Tools/DatasetWrapper.hpp:186-263, the loader's depth pass ahead of all of the
 *   above): raw u16 depth (device, updated in place like Frame::depth) -> readings above maximum_depth * depth_scale
 *   dropped -> metres -> cv::bilateralFilter(refined_depth, ., d = 9 (7 on MobileCPU builds), sigma_color = 0.03,
 *   sigma_space = 10) -> d_refined (Frame::refined_depth, f32, may be NULL) and written back to the u16 map.  The
 *   filter restates OpenCV's CV_32FC1 algorithm (4096-bin colour table over the image's own range, circular taps,
 *   BORDER_REFLECT_101); tap sums run in tap order, which OpenCV's vector builds do not pin (oracle/tf_oracle.c).
 *   Asynchronous on the handle's stream; d <= 15. */
TF_API int tf_pre_frame_depth(tf_volume* v, uint16_t* d_depth, float* d_refined, float maximum_depth, float depth_scale,
                              int d, double sigma_color, double sigma_space);
TF_API int tf_pre_normal_map(tf_volume* v, const float* d_depth, float* d_normal);
TF_API int tf_pre_refine_depth_normal(tf_volume* v, float* d_normal, float* d_depth);
TF_API int tf_pre_color_valid(tf_volume* v, const float* d_normal, uint8_t* d_flag);
TF_API int tf_pre_color_quality(tf_volume* v, const float* d_depth, const float* d_normal, const uint8_t* d_rgb,
                                float* d_quality);
TF_API int tf_pre_refine_newframe(tf_volume* v, const float* d_depth_ref, float* d_depth_new,
                                  const float T_new_to_ref[12]);
TF_API int tf_pre_refine_keyframe(tf_volume* v, float* d_depth_ref, float* d_weight_ref, const float* d_depth_new,
                                  const float T_ref_to_new[12], int32_t* rounds);

#ifdef __cplusplus
}
#endif
#endif /* TF_FUSION_H_ */
