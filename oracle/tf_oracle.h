/*
 * tf_oracle.h -- CPU restatement ("oracle") of TextureFusion's per-frame voxel-fusion +
 * texture-atlas-update hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (texturefusion_amd/, include/)
 * never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" except for the known-answer values recorded in
 * SURVEY.md App. A.1-9 (outputs the survey observed from the reference's own translation
 * unit) and analytic known-answer tests.  The reference has no tests / golden vectors of its
 * own and cannot be built in this image: every file on the path needs Eigen,
 * and the facade/atlas files also need OpenCV and Sophus, none of which are installed.  The
 * rules of this build forbid making a reference build out of stand-in headers, so no
 * oracle/_ref exists and no fixture in tests/golden/ was produced by reference code.
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root).
 *
 * All arithmetic is scalar IEEE binary32/binary64 with every operation rounded separately
 * (built with -ffp-contract=off, no fast-math), in the operation order of the reference
 * build (-O3 -mavx2, no -mfma, no -ffast-math; CMakeLists.txt:57-58).
 */
#ifndef TF_ORACLE_H_
#define TF_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFO_CHUNK_DIM 8
#define TFO_CHUNK_VOXELS 512
#define TFO_ATLAS_DIM 13824 /* Structure/Atlas.h:29-30: 96*72*2 */

/* PinholeCamera (3rd_party/open_chisel/camera/PinholeCamera.h:36-77).  The float intrinsics
 * are stored as given; every consumer sees them through the int-returning getters
 * (PinholeCamera.h:46-49), i.e. truncated toward zero. */
typedef struct {
  int width, height;
  float fx, fy, cx, cy;
  float near_plane, far_plane;
} tfo_camera;

/* QuadraticTruncator (truncation/QuadraticTruncator.h:33-48) + ConstantWeighter weight
 * (weighting/ConstantWeighter.h:34-46). */
typedef struct {
  float quad, lin, cons, scale;
  float weight;
} tfo_integrator;

/* Row statistics of one voxel-update call: the integers SURVEY.md s.8(d) builds the
 * algorithmic byte count from. */
typedef struct {
  int64_t rows_tsdf;  /* 8-voxel rows whose sdf/weight were rewritten */
  int64_t rows_color; /* 8-voxel rows whose colour was rewritten */
  int64_t chunks_updated;
} tfo_rowstats;

/* Pose = Eigen::Affine3f as 12 floats, row-major 3x4 [R | t]: R(i,j)=p[4*i+j], t(i)=p[4*i+3]. */

/* ---- scalar pieces ---------------------------------------------------------------- */
/* test aid: alternative summation orders at the sites where Eigen's order is unpinned (bit 0: fixed-size dots
 * sequential instead of a0b0 + (a1b1 + a2b2); bit 1: dynamic mat-vec in that tree order instead of sequential) */
void tfo_set_sum_order(int bits);
int tfo_get_sum_order(void);
float tfo_truncation(const tfo_integrator* ig, float z);
/* ConstantWeighter::GetWeight (weighting/ConstantWeighter.h:43-46) */
float tfo_weight(const tfo_integrator* ig, float truncation);
/* chisel::parallel_for's cut of n items (threading/Threading.h:36-54): stretch length to *group_out, returns the
 * number of threads that run items (spawned + caller).  Pinned against the compiled header by tests/test_ref_pin.py. */
int tfo_parallel_for_plan(int64_t n, int nthreads, int threshold, int64_t* group_out);
void tfo_centroids(const float pose[12], float res, float cen[3 * TFO_CHUNK_VOXELS]);
void tfo_chunk_scalars(const tfo_integrator* ig, const float pose[12], const int id[3], float res,
                       float origin_cam[3], float* truncation, float* weight);

/* ---- K-A: one chunk, one frame ---------------------------------------------------- */
int tfo_voxel_update(const float* depth, const uint8_t* rgba, const float* quality,
                     const tfo_camera* cam, const tfo_integrator* ig, const float pose[12],
                     int integrate_flag, const int id[3], float res,
                     const float cen[3 * TFO_CHUNK_VOXELS], float* sdf, float* weight,
                     uint16_t* color, float* quality_out, tfo_rowstats* stats);

/* kernel = 0: scalar restatement (the checker); 1: AVX2 row kernel, bit-identical, used for the
 * CPU baseline (the reference's kernel is AVX2 as well). */
int tfo_have_avx2(void);
int tfo_voxel_update_k(int kernel, const float* depth, const uint8_t* rgba, const float* quality,
                       const tfo_camera* cam, const tfo_integrator* ig, const float pose[12],
                       int integrate_flag, const int id[3], float res,
                       const float cen[3 * TFO_CHUNK_VOXELS], float* sdf, float* weight,
                       uint16_t* color, float* quality_out, tfo_rowstats* stats);

/* ---- K-B / K-C: visible-chunk selection ------------------------------------------- */
void tfo_bbox(const float* depth, const tfo_camera* cam, const float pose[12], float res,
              int min_id[3], int max_id[3]);
/* returns the number of selected chunks (may exceed cap; only the first cap are written) */
int64_t tfo_select(const float* depth, const tfo_camera* cam, const tfo_integrator* ig,
                   const float pose[12], float res, int32_t* ids, int64_t cap,
                   int64_t* n_coarse_tested);

/* ---- volume (Chisel + ChunkManager state) ------------------------------------------ */
typedef struct tfo_volume tfo_volume;
tfo_volume* tfo_volume_create(float res, int use_color);
void tfo_volume_destroy(tfo_volume* v);
void tfo_volume_reset(tfo_volume* v);
void tfo_volume_set_camera(tfo_volume* v, const tfo_camera* cam);
void tfo_volume_set_integrator(tfo_volume* v, const tfo_integrator* ig);
void tfo_volume_set_threads(tfo_volume* v, int nthreads);
void tfo_volume_set_kernel(tfo_volume* v, int kernel);
int64_t tfo_volume_num_chunks(const tfo_volume* v);
int64_t tfo_volume_list_chunks(const tfo_volume* v, int32_t* ids, int64_t cap);
int tfo_volume_has_chunk(const tfo_volume* v, const int id[3]);
int tfo_volume_get_chunk(const tfo_volume* v, const int id[3], float* sdf, float* weight,
                         uint16_t* color);
int tfo_volume_set_chunk(tfo_volume* v, const int id[3], const float* sdf, const float* weight,
                         const uint16_t* color);
int64_t tfo_volume_get_observations(const tfo_volume* v, const int id[3], int32_t* kf,
                                    float* q, int64_t cap);
int64_t tfo_volume_retract_observations(tfo_volume* v, int kf, const int32_t* ids, int64_t n);
/* selection (K-B / K-C): 0 = scalar checker, 1 = AVX2 forms with the reference's vector shapes (CPU baseline) */
void tfo_set_select_kernel(int kernel);
int64_t tfo_volume_num_dirty(const tfo_volume* v);
int64_t tfo_volume_list_dirty(const tfo_volume* v, int32_t* ids, int64_t cap);
void tfo_volume_clear_dirty(tfo_volume* v);
void tfo_volume_get_rowstats(const tfo_volume* v, tfo_rowstats* out);
void tfo_volume_clear_rowstats(tfo_volume* v);

/* Chisel::PrepareIntersectChunks (Structure/Chisel.h:103-140) */
int64_t tfo_prepare(tfo_volume* v, const float* depth, const float pose[12], int32_t* ids,
                    uint8_t* is_new, int64_t cap);
/* Chisel::IntegrateDepthScanColor, 10-arg (Structure/Chisel.h:218-249) */
int tfo_integrate(tfo_volume* v, const float* depth, const uint8_t* rgba, const float* quality,
                  const float pose[12], const int32_t* ids, int64_t n, int integrate_flag,
                  int keyframe_id, uint8_t* needs_update, float* quality_out);
/* Chisel::FinalizeIntegrateChunks (Structure/Chisel.h:184-216) */
int64_t tfo_finalize(tfo_volume* v, const int32_t* ids, const uint8_t* needs_update,
                     const uint8_t* is_new, int64_t n, int32_t* valid_ids);
/* Chisel::IntegrateDepthScanColor, 5-arg (Structure/Chisel.h:453-468): the per-frame unit */
int64_t tfo_integrate_frame(tfo_volume* v, const float* depth, const uint8_t* rgba,
                            const float pose[12], int64_t* n_selected);

/* ---- atlas (Structure/Atlas.{h,cpp}, Structure/Patch.cpp) -------------------------- */
typedef struct tfo_atlas tfo_atlas;
tfo_atlas* tfo_atlas_create(float res, int atlas_w, int atlas_h);
void tfo_atlas_destroy(tfo_atlas* a);
int tfo_atlas_patch_w(const tfo_atlas* a);
int tfo_atlas_patch_h(const tfo_atlas* a);
/* Atlas::AddPatch slot logic (Atlas.cpp:43-64): returns 0 and the texloc, or -1 on overflow */
int tfo_atlas_alloc(tfo_atlas* a, uint64_t* texloc);
uint64_t tfo_atlas_loc_next(const tfo_atlas* a);
uint8_t* tfo_atlas_buffer(tfo_atlas* a);
/* Patch::CalculateTexCoords (Patch.cpp:40-108) */
int tfo_patch_project(const float* verts, const float* colors, int64_t n_v, const float T[16],
                      const uint8_t* rgb, const float* depth, const tfo_camera* cam,
                      float* texcoord, float* texcolor, int32_t bbox[4], int* wrong_mapping,
                      int64_t* n_caution);
/* Atlas::UpdateBuffer (Atlas.cpp:71-91); ratio[2] in/out */
int tfo_atlas_blit(tfo_atlas* a, uint64_t texloc, const uint8_t* rgb, int img_w, int img_h,
                   const int32_t bbox[4], float ratio[2]);
/* Chisel::CompensateColor (Structure/Chisel.cpp:198-286) + computeMeanAndCov (Structure/Patch.cpp:342-348)
 * over a batch of patches in the reference's iteration order.  Patches with has_adjusted != 0 are
 * skipped; the others are clustered by frame id (cluster order = first appearance).  Per cluster:
 * mean / covariance (N-1) of texcolor and of the mesh colours over the patches without
 * wrong_mapping, the 3x3 transfer T, labs[k] = T (texcolor[k] - mean_src) + mean_tar for every
 * vertex of the cluster's patches that map correctly, has_adjusted := 1.  A cluster whose patches
 * are all wrong-mapped is left untouched (the reference `continue`s before the flag is set).
 * Symmetric eigen-decompositions: cyclic Jacobi in double on the f32 matrices -- Eigen's own
 * iteration is not restated ("parity unpinned", 1-ulp class); the transfer
 * matrix does not depend on eigenvector signs or order.  out_T (optional): 9 floats per cluster,
 * out_cluster (optional): cluster index per patch (-1 = skipped).  Returns the cluster count. */
/* the 3x3 transfer matrix alone (Chisel.cpp:247-266) */
void tfo_color_transfer(const float cov_src[9], const float cov_tar[9], float T[9]);
int64_t tfo_color_compensate(int64_t n_patches, const int32_t* frame_ids, const uint8_t* wrong_mapping,
                             uint8_t* has_adjusted, const int64_t* vert_offsets, const float* texcolor,
                             const float* meshcolor, float* labs, float* out_T, int32_t* out_cluster);
/* Chisel::DrawMeshes (Structure/Chisel.cpp:288-355): the interleaved
 * vertex stream the renderer / exporter consumes -- 12 f32 per vertex
 *   [x, y, z, 50, (float)(R<<16|G<<8|B), adj, u/atlas_w, v/atlas_h, nx, ny, nz, wrong_mapping]
 * with (u,v) = texcoord * (ratio < 1 ? ratio : 1) + slot origin (Atlas::GetTexLoc, Atlas.cpp:66-69),
 * adj = labs valid ? (float)(3 x 9 bit of int((labs - texcolor) * 255) + 255) : 0 -- and the index
 * stream rebased by the running vertex count, for the patches that are complete() (Patch.cpp:191-196),
 * in patch order.  Returns the vertex count; *n_indices = index count. */
int64_t tfo_pack_vertices(int64_t n_patches, const uint8_t* complete, const uint8_t* wrong_mapping,
                          const uint8_t* labs_valid, const uint64_t* texloc, const float* ratio,
                          int atlas_w, int atlas_h, const int64_t* vert_offsets, const float* verts,
                          const float* colors, const float* normals, const float* texcoord,
                          const float* texcolor, const float* labs, const int64_t* index_offsets,
                          const uint32_t* indices, float* out_vertices, uint32_t* out_indices,
                          int64_t* n_indices);
/* GeneratePatches + UpdateAtlas over a batch of patches of one keyframe (CPU-baseline helper) */
int tfo_patches_batch(tfo_atlas* a, int64_t n_patches, const uint64_t* texloc, const int64_t* voff,
                      const float* verts, const float* colors, const float* T16, const uint8_t* rgb,
                      const float* depth, const tfo_camera* cam, float* texcoord, float* texcolor);
/* hot row range, Chisel.cpp:153-186 */
void tfo_atlas_hot_range(const tfo_atlas* a, const uint64_t* texlocs, int64_t n,
                         uint64_t* hot_start, uint64_t* hot_end);

/* ---- meshing ------------------------------------------------
 * ChunkManager::GenerateMeshEfficient (Structure/ChunkManager.cpp:595-1002) incl.
 * extractGradientFromCubic (:277-455) for one chunk of the volume: marching cubes over the 512
 * cells (corners may live in the 7 chunks id + {0,1}^3, gradients read the face neighbours of
 * those), vertices kept when the corner's weight > 50 and its gradient is valid, de-duplicated on
 * the 9x9x9x3 edge grid (last writer in cell order wins), indices remapped.  Buffers (may be NULL):
 * verts / normals / colors f32[3 * 2187], indices u32[7680].  Returns the vertex count, -1 when
 * the chunk does not exist; *n_indices = index count. */
#define TFO_MESH_MAX_VERTS 2187
#define TFO_MESH_MAX_INDS 7680
int64_t tfo_mesh_chunk(const tfo_volume* v, const int id[3], float* verts, float* normals,
                       float* colors, uint32_t* indices, int64_t* n_indices);
/* Chisel::UpdateMeshes (Structure/Chisel.h:479-481) = ChunkManager::RecomputeMeshes
 * (Structure/ChunkManager.cpp:232-264) over meshesToUpdate; returns the number of chunks meshed */
int64_t tfo_update_meshes(tfo_volume* v);
int64_t tfo_volume_num_meshes(const tfo_volume* v);
int64_t tfo_volume_list_meshes(const tfo_volume* v, int32_t* ids, int64_t cap);
int tfo_volume_get_mesh(const tfo_volume* v, const int id[3], int64_t* nv, int64_t* ni, float* verts,
                        float* normals, float* colors, uint32_t* indices, uint8_t adj[6], int* simplified);
/* Mesh::SimplifyByClustering / GetIndice (3rd_party/open_chisel/geometry/Mesh.cpp:39-83) */
void tfo_mesh_adjacency(const float* verts, int64_t nv, const float origin[3], float grid, uint8_t adj[6]);
/* Chisel::CompressMeshes (Structure/Chisel.cpp:112-147) on meshesToUpdate (which it clears);
 * out_ids = tsdfFusion's chunksToUpdate (GCFusion/MobileFusion.cpp:345-353) in ascending id order */
int64_t tfo_compress_meshes(tfo_volume* v, int32_t* out_ids, int64_t cap);

/* ---- the atlas stage on the volume's meshes (Mesh::m_patch state kept with the mesh) -------- */
typedef struct {
  const uint8_t* rgb; /* u8[H][W][3] Frame::rgb */
  const float* depth; /* f32[H][W] Frame::refined_depth */
  float T[16];        /* f32(SE3d.inverse().matrix()) of the keyframe's pose, row-major */
  int kf_id;          /* the frame id (label) */
} tfo_keyframe;
int tfo_generate_patches(tfo_volume* v, tfo_atlas* a, const int32_t* ids, int64_t n, const int32_t* kf_index,
                         const tfo_keyframe* kfs, uint64_t hot[2]);
void tfo_update_atlas(tfo_volume* v, tfo_atlas* a, const int32_t* ids, int64_t n);
int64_t tfo_compensate_color_volume(tfo_volume* v);
int64_t tfo_draw_meshes(tfo_volume* v, const tfo_atlas* a, float* out_vertices, uint32_t* out_indices,
                        int64_t cap_v, int64_t cap_i, int64_t* n_indices);
int tfo_volume_get_patch(const tfo_volume* v, const int id[3], uint64_t* texloc, int* frameid, int32_t bbox[4],
                         int* flags, float ratio[2], int64_t* pnv, float* texcoord, float* texcolor, float* labs);
int64_t tfo_frame_textured(tfo_volume* v, tfo_atlas* a, const float* depth, const uint8_t* rgba, const float pose[12],
                           const float pose_inv16[16], int frame_id, uint8_t* rgb_scratch);

/* ---- frame pre-processing feeding the path (BasicAPI.cpp:378-443, 506-636, 728-905; see tf_oracle.c) ---- */
void tfo_pre_normal_map(const float* depth, int W, int H, float fx, float fy, float cx, float cy, float* normal);
void tfo_pre_refine_depth_normal(float* normal, float* depth, int W, int H, float fx, float fy, float cx, float cy);
void tfo_pre_color_valid(const float* normal, int W, int H, float fx, float fy, float cx, float cy, uint8_t* flag);
void tfo_pre_color_quality(const float* depth, const float* normal, const uint8_t* rgb, int W, int H, float fx,
                           float fy, float cx, float cy, float* quality);
void tfo_pre_refine_newframe(const float* depth_ref, float* depth_new, int W, int H, float fx, float fy, float cx,
                             float cy, const float T[12]);
void tfo_pre_refine_keyframe(float* depth_ref, float* weight_ref, const float* depth_new, int W, int H, float fx,
                             float fy, float cx, float cy, const float T[12]);
void tfo_pre_frame_depth(uint16_t* depth, int W, int H, float maximum_depth, float depth_scale, int d,
                         double sigma_color, double sigma_space, float* refined_out);

#ifdef __cplusplus
}
#endif
#endif /* TF_ORACLE_H_ */
