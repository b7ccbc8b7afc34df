// tf_chisel.hpp -- host-side C++14 mirror of the reference's operator API for the fusion path.
//
// Same class / method names, argument order and error behaviour as the reference's
// chisel::Chisel / ChunkManager / Atlas / ProjectionIntegrator / PinholeCamera
// (Structure/Chisel.h:46-493, Structure/ChunkManager.h:119-207, Structure/Atlas.h:43-75,
// 3rd_party/open_chisel/utils/ProjectionIntegrator.h:42-94, camera/PinholeCamera.h:36-77);
// every body forwards to the C ABI in include/tf_fusion.h, so GCFusion/MobileFusion.{h,cpp}
// style callers compile against it with only the math types swapped.
// The reference uses Eigen / OpenCV types in these signatures; neither library exists in this
// image, so the few value types the path needs (ChunkID, Transform, Vec3) are defined here
// with the same member names the callers use.
//
// Header-only; link with libtexfusion_hip.so.  No CPU fallback: without a GPU the Chisel
// constructor throws std::runtime_error carrying tf_last_error().
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tf_fusion.h"

namespace chisel {

// ---- value types (Eigen stand-ins with the members the path's callers touch) -------------
struct ChunkID {  // Eigen::Vector3i in the reference (geometry/Geometry.h)
  int v[3];
  ChunkID() : v{0, 0, 0} {}
  ChunkID(int x, int y, int z) : v{x, y, z} {}
  int operator()(int i) const { return v[i]; }
  int& operator()(int i) { return v[i]; }
  ChunkID operator+(const ChunkID& o) const { return ChunkID(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  bool operator==(const ChunkID& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};
typedef std::vector<ChunkID> ChunkIDList;

struct ChunkHasher {  // Structure/ChunkManager.h:44-53
  std::size_t operator()(const ChunkID& k) const {
    return (std::size_t)k(0) * 73856093u ^ (std::size_t)k(1) * 19349663u ^ (std::size_t)k(2) * 83492791u;
  }
};
typedef std::unordered_map<ChunkID, bool, ChunkHasher> ChunkSet;

struct Vec3 {
  float v[3];
  Vec3() : v{0, 0, 0} {}
  Vec3(float x, float y, float z) : v{x, y, z} {}
  float operator()(int i) const { return v[i]; }
};

// Eigen::Affine3f: camera-to-world [R | t], row-major 3x4.
struct Transform {
  float m[12];
  Transform() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = 1.0f; }
  float& operator()(int r, int c) { return m[4 * r + c]; }
  float operator()(int r, int c) const { return m[4 * r + c]; }
  const float* data() const { return m; }
};

// ---- parameter objects ------------------------------------------------------------------
class PinholeCamera {  // camera/PinholeCamera.h:36-77
 public:
  inline void SetIntrinsics(float ifx, float ify, float icx, float icy) { fx = ifx; fy = ify; cx = icx; cy = icy; }
  inline int GetWidth() const { return width; }
  inline int GetFx() const { return fx; }  // int-returning getters, as in the reference (:46-49)
  inline int GetFy() const { return fy; }
  inline int GetCx() const { return cx; }
  inline int GetCy() const { return cy; }
  inline int GetHeight() const { return height; }
  inline void SetWidth(int value) { width = value; }
  inline void SetHeight(int value) { height = value; }
  inline float GetNearPlane() const { return nearPlane; }
  inline float GetFarPlane() const { return farPlane; }
  inline void SetNearPlane(float value) { nearPlane = value; }
  inline void SetFarPlane(float value) { farPlane = value; }
  float fx = 525.f, fy = 525.f, cx = 319.5f, cy = 239.5f;
  int width = 640, height = 480;
  float nearPlane = 0.01f, farPlane = 5.0f;
};

class QuadraticTruncator {  // truncation/QuadraticTruncator.h:33-48
 public:
  QuadraticTruncator() = default;
  QuadraticTruncator(float quadratic, float linear, float constant, float scale)
      : quadraticTerm(quadratic), linearTerm(linear), constantTerm(constant), scalingFactor(scale) {}
  float GetTruncationDistance(float reading) const {
    return std::abs(quadraticTerm * std::pow((double)reading, 2) + linearTerm * reading + constantTerm) * scalingFactor;
  }
  float quadraticTerm = 0.0019f, linearTerm = 0.00152f, constantTerm = 0.001504f, scalingFactor = 6.0f;
};
typedef std::shared_ptr<QuadraticTruncator> TruncatorPtr;

class ConstantWeighter {  // weighting/ConstantWeighter.h:34-46
 public:
  ConstantWeighter() = default;
  explicit ConstantWeighter(float w) : weight(w) {}
  float GetWeight(float, float truncationDist) const { return weight / (2 * truncationDist); }
  float weight = 1.0f;
};
typedef std::shared_ptr<ConstantWeighter> WeighterPtr;

class ProjectionIntegrator {  // utils/ProjectionIntegrator.h:42-94 (parameters only; the kernel is on the GPU)
 public:
  inline const TruncatorPtr& GetTruncator() const { return truncator; }
  inline void SetTruncator(const TruncatorPtr& value) { truncator = value; }
  inline const WeighterPtr& GetWeighter() const { return weighter; }
  inline void SetWeighter(const WeighterPtr& value) { weighter = value; }
  inline float GetCarvingDist() const { return carvingDist; }
  inline bool IsCarvingEnabled() const { return enableVoxelCarving; }
  inline void SetCarvingDist(float dist) { carvingDist = dist; }       // unused by the reference kernel
  inline void SetCarvingEnabled(bool enabled) { enableVoxelCarving = enabled; }

 protected:
  TruncatorPtr truncator = std::make_shared<QuadraticTruncator>();
  WeighterPtr weighter = std::make_shared<ConstantWeighter>();
  float carvingDist = 0;
  bool enableVoxelCarving = false;
};

// ---- host mirror of one chunk (geometry/Chunk.h:48-184) ------------------------------------
struct DistVoxel {
  std::vector<float> sdf, weight;  // [512] each (DistVoxel.h:102-103)
};
struct ColorVoxel {
  std::vector<unsigned short> colorData;  // [512*4] R,G,B,count (ColorVoxel.h:66)
};
class Chunk {
 public:
  Chunk(const ChunkID& id, float res) : ID(id), voxelResolutionMeters(res) {
    voxels.sdf.assign(512, 999.0f);
    voxels.weight.assign(512, 0.0f);
    colors.colorData.assign(2048, 0);
  }
  inline const ChunkID& GetID() const { return ID; }
  inline Vec3 GetOrigin() const {  // Chunk.cpp:52
    return Vec3(8 * ID(0) * voxelResolutionMeters, 8 * ID(1) * voxelResolutionMeters, 8 * ID(2) * voxelResolutionMeters);
  }
  DistVoxel voxels;
  ColorVoxel colors;
  std::map<int, float> observations;  // Chunk.h:171

 protected:
  ChunkID ID;
  float voxelResolutionMeters;
};
typedef std::shared_ptr<Chunk> ChunkPtr;

inline void tf_check(int rc, const char* what) {
  if (rc != TF_OK) throw std::runtime_error(std::string(what) + ": " + tf_last_error());
}

// ---- ChunkManager: queries forward to the device volume; chunks are mirrored on demand -----
class ChunkManager {
 public:
  ChunkManager() = default;
  void Bind(tf_volume* v, float res) { vol = v; voxelResolutionMeters = res; }
  inline float GetResolution() const { return voxelResolutionMeters; }
  inline bool HasChunk(const ChunkID& chunk) const {  // ChunkManager.h:133-135
    int out = 0;
    tf_check(tf_has_chunk(vol, chunk.v, &out), "HasChunk");
    return out != 0;
  }
  // ChunkManager::GetChunk (:137-139): chunks.at() -> std::out_of_range when absent.  Returns a
  // host mirror refreshed from HBM; observations persist on the host side.
  inline ChunkPtr GetChunk(const ChunkID& chunk) {
    ChunkPtr c = Mirror(chunk);
    int rc = tf_chunk_download(vol, chunk.v, c->voxels.sdf.data(), c->voxels.weight.data(), c->colors.colorData.data());
    if (rc == TF_ERR_MISSING_CHUNK) throw std::out_of_range("ChunkManager::GetChunk: no such chunk");
    tf_check(rc, "GetChunk");
    return c;
  }
  ChunkIDList GetChunkIDs() const {
    int64_t n = 0;
    tf_check(tf_list_chunks(vol, nullptr, 0, &n), "GetChunks");
    std::vector<int32_t> ids((size_t)n * 3 + 3);
    tf_check(tf_list_chunks(vol, ids.data(), n, &n), "GetChunks");
    ChunkIDList out;
    for (int64_t i = 0; i < n; ++i) out.emplace_back(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
    return out;
  }
  // observation bookkeeping of Chisel::IntegrateDepthScanColor (Chisel.h:244-247)
  ChunkPtr Mirror(const ChunkID& id) {
    auto it = mirrors.find(id);
    if (it != mirrors.end()) return it->second;
    ChunkPtr c = std::make_shared<Chunk>(id, voxelResolutionMeters);
    mirrors.emplace(id, c);
    return c;
  }
  void DropMirror(const ChunkID& id) { mirrors.erase(id); }
  void Reset() { mirrors.clear(); }

 private:
  tf_volume* vol = nullptr;
  float voxelResolutionMeters = 0.005f;
  std::unordered_map<ChunkID, ChunkPtr, ChunkHasher> mirrors;
};

// ---- Atlas (Structure/Atlas.h:43-75) -------------------------------------------------------
class Atlas {
 public:
  std::size_t loc_next = 0, PATCH_WIDTH = 0, PATCH_HEIGHT = 0, hot_start = 0, hot_end = 0;
  void Bind(tf_volume* v) {
    vol = v;
    int32_t pw = 0, ph = 0;
    tf_check(tf_atlas_patch_size(vol, &pw, &ph), "Atlas");
    PATCH_WIDTH = pw;
    PATCH_HEIGHT = ph;
  }
  // texture_buffer rows [row0,row1) (the GUI uploads only the hot rows, MobileFusion.h:406-421)
  void DownloadRows(int64_t row0, int64_t row1, unsigned char* dst) {
    tf_check(tf_atlas_download_rows(vol, row0, row1, dst), "Atlas::DownloadRows");
  }
  void Refresh() {
    uint64_t l = 0;
    tf_check(tf_atlas_loc_next(vol, &l), "Atlas");
    loc_next = l;
  }

 private:
  tf_volume* vol = nullptr;
};

// Mesh input of the atlas stage (geometry/Mesh.h:70-85: vertices / colors of one chunk's mesh).
struct PatchMesh {
  ChunkID chunkID;
  std::vector<float> vertices;  // xyz per vertex
  std::vector<float> colors;    // rgb in [0,1] per vertex
  std::vector<float> normals;   // xyz per vertex (DrawMeshes only)
  std::vector<unsigned int> indices;  // (DrawMeshes only)
};
// Patch results (Structure/Patch.h:51-94 members the callers read).
struct PatchResult {
  std::size_t texloc = 0;
  int boundingbox[4] = {0, 0, 0, 0};  // x, y, width, height
  bool wrong_mapping = false;
  int flag = 0;  // CalculateTexCoords' return value (0 / -1)
  float ratio[2] = {1, 1};
  std::vector<float> texcoord, texcolor;
  int frameid = -1;           // Patch::frameid: source keyframe (the label)
  bool has_adjusted = false;  // Patch::has_adjusted
  std::vector<float> labs;    // Patch::labs: compensated colours (empty when wrong_mapping)
};

// ---- Chisel (Structure/Chisel.h:46-493) ----------------------------------------------------
class Chisel {
 public:
  Chisel(const int chunkSize[3], float voxelResolution, bool useColor, const tf_config* cfg = nullptr) {
    int32_t dims[3] = {chunkSize[0], chunkSize[1], chunkSize[2]};
    tf_check(tf_volume_create(dims, voxelResolution, useColor ? 1 : 0, cfg, &vol), "Chisel");
    res = voxelResolution;
    chunkManager.Bind(vol, res);
    atlas.Bind(vol);
  }
  virtual ~Chisel() { tf_volume_destroy(vol); }
  Chisel(const Chisel&) = delete;
  Chisel& operator=(const Chisel&) = delete;

  inline const ChunkManager& GetChunkManager() const { return chunkManager; }
  inline ChunkManager& GetMutableChunkManager() { return chunkManager; }
  tf_volume* Handle() { return vol; }

  void Reset() {  // Chisel.cpp:47-50
    tf_check(tf_volume_reset(vol), "Reset");
    chunkManager.Reset();
    meshesToUpdate.clear();
  }

  // Structure/Chisel.h:103-140.  depthImage is borrowed for the call (uploaded to HBM and kept
  // bound for the IntegrateDepthScanColor calls that follow, like the reference keeps the cv::Mat).
  void PrepareIntersectChunks(ProjectionIntegrator& integrator, float* depthImage,
                              const Transform& depthExtrinsic, const PinholeCamera& depthCamera,
                              ChunkIDList& chunksIntersecting, std::vector<bool>& needsUpdateFlag,
                              std::vector<bool>& newChunkFlag) {
    Configure(integrator, depthCamera);
    chunksIntersecting.clear();
    needsUpdateFlag.clear();
    newChunkFlag.clear();
    UploadFrame(depthImage, nullptr, nullptr, depthCamera);
    int64_t n = 0, cap = (int64_t)ids_buf.size() / 3;
    int rc = tf_prepare(vol, depthExtrinsic.data(), ids_buf.data(), new_buf.data(), cap, &n);
    if (rc == TF_ERR_CAPACITY && n > cap) {
      ids_buf.resize((size_t)n * 3);
      new_buf.resize((size_t)n);
      rc = tf_prepare(vol, depthExtrinsic.data(), ids_buf.data(), new_buf.data(), n, &n);
    }
    tf_check(rc, "PrepareIntersectChunks");
    for (int64_t i = 0; i < n; ++i) {
      chunksIntersecting.emplace_back(ids_buf[3 * i], ids_buf[3 * i + 1], ids_buf[3 * i + 2]);
      newChunkFlag.emplace_back(new_buf[i] != 0);
      needsUpdateFlag.emplace_back(false);
    }
    tf_stats st;
    if (tf_get_stats(vol, &st) == TF_OK) {
      minChunkID = ChunkID(st.min_id[0], st.min_id[1], st.min_id[2]);
      maxChunkID = ChunkID(st.max_id[0], st.max_id[1], st.max_id[2]);
    }
  }

  // Structure/Chisel.h:218-249 (10-argument overload).
  void IntegrateDepthScanColor(ProjectionIntegrator& integrator, float* depthImage, unsigned char* colorImage,
                               const Transform& depthExtrinsic, const PinholeCamera& depthCamera,
                               ChunkIDList& chunksIntersecting, std::vector<bool>& needsUpdateFlag,
                               int integrate_flag, int keyframeID = -1, float* observationQualityPointer = NULL) {
    Configure(integrator, depthCamera);
    if (chunksIntersecting.size() < 1) return;
    UploadFrame(depthImage, colorImage, observationQualityPointer, depthCamera);
    const size_t n = chunksIntersecting.size();
    Flatten(chunksIntersecting);
    needs_buf.resize(n);
    qual_buf.resize(n);
    for (size_t i = 0; i < n; ++i) needs_buf[i] = needsUpdateFlag[i] ? 1 : 0;
    tf_check(tf_integrate(vol, depthExtrinsic.data(), ids_buf.data(), (int64_t)n, integrate_flag,
                          colorImage != NULL, observationQualityPointer != NULL && colorImage != NULL,
                          needs_buf.data(), qual_buf.data()),
             "IntegrateDepthScanColor");
    for (size_t i = 0; i < n; ++i) {
      needsUpdateFlag[i] = needs_buf[i] != 0;
      if (keyframeID >= 0 && qual_buf[i] > 0 && needsUpdateFlag[i])  // Chisel.h:244-247
        chunkManager.Mirror(chunksIntersecting[i])->observations[keyframeID] = qual_buf[i];
    }
  }

  // Structure/Chisel.h:184-216.
  void FinalizeIntegrateChunks(ChunkIDList& chunksIntersecting, std::vector<bool>& needsUpdateFlag,
                               std::vector<bool>& newChunkFlag, ChunkIDList& validChunks) {
    validChunks.clear();
    const size_t n = chunksIntersecting.size();
    Flatten(chunksIntersecting);
    needs_buf.resize(n);
    new_buf.resize(std::max(n, new_buf.size()));
    for (size_t i = 0; i < n; ++i) {
      needs_buf[i] = needsUpdateFlag[i] ? 1 : 0;
      new_buf[i] = newChunkFlag[i] ? 1 : 0;
    }
    std::vector<int32_t> valid(n * 3 + 3);
    int64_t nv = 0;
    tf_check(tf_finalize(vol, ids_buf.data(), needs_buf.data(), new_buf.data(), (int64_t)n, valid.data(), &nv),
             "FinalizeIntegrateChunks");
    for (int64_t i = 0; i < nv; ++i) validChunks.emplace_back(valid[3 * i], valid[3 * i + 1], valid[3 * i + 2]);
    for (size_t i = 0; i < n; ++i)
      if (!needsUpdateFlag[i] && newChunkFlag[i]) chunkManager.DropMirror(chunksIntersecting[i]);  // GarbageCollect
    meshes_stale = true;
  }

  // Structure/Chisel.h:453-468 (5-argument overload): the per-frame unit, fused on the device.
  void IntegrateDepthScanColor(ProjectionIntegrator& integrator, float* depthImage, unsigned char* colorImage,
                               const Transform& depthExtrinsic, const PinholeCamera& depthCamera) {
    Configure(integrator, depthCamera);
    UploadFrame(depthImage, colorImage, nullptr, depthCamera);
    tf_check(tf_integrate_frame(vol, depthExtrinsic.data(), colorImage != NULL), "IntegrateDepthScanColor");
    meshes_stale = true;
  }

  // Chisel::meshesToUpdate (Chisel.h:489), refreshed from the device dirty set on access.
  const ChunkSet& GetMeshesToUpdate() {
    if (meshes_stale) {
      int64_t n = 0;
      tf_check(tf_list_dirty(vol, nullptr, 0, &n), "meshesToUpdate");
      std::vector<int32_t> ids((size_t)n * 3 + 3);
      tf_check(tf_list_dirty(vol, ids.data(), n, &n), "meshesToUpdate");
      meshesToUpdate.clear();
      for (int64_t i = 0; i < n; ++i) meshesToUpdate[ChunkID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2])] = true;
      meshes_stale = false;
    }
    return meshesToUpdate;
  }
  void ClearMeshesToUpdate() {  // chunksToUpdate.clear() at the end of CompressMeshes (Chisel.cpp:146)
    tf_check(tf_clear_dirty(vol), "meshesToUpdate.clear");
    meshesToUpdate.clear();
    meshes_stale = false;
  }

  // Frame::rgb / refined_depth of a keyframe, kept alive for Patch::SetImage (Patch.cpp:172-175).
  void CacheKeyframe(int frameid, const unsigned char* rgb, const float* depth) {
    tf_check(tf_keyframe_cache(vol, frameid, rgb, depth), "CacheKeyframe");
  }

  // Chisel::GeneratePatches + Chisel::UpdateAtlas (Chisel.cpp:149-196) over the chunks that have a
  // mesh.  labels[i] = source keyframe of chunk i (the MRF's output), pose_inv[i] = f32 of that
  // keyframe's SE3d inverse (16 floats row-major).  Returns 0, or -1 when the atlas is full.
  int GeneratePatchesAndUpdateAtlas(const std::vector<PatchMesh>& meshes, const std::vector<int>& labels,
                                    const std::vector<const float*>& pose_inv, const PinholeCamera& cameraModel,
                                    std::vector<PatchResult>& out) {
    ProjectionIntegrator dummy;
    Configure(dummy, cameraModel, false);
    const size_t np = meshes.size();
    std::vector<int32_t> ids(np * 3), kf(np), bbox(np * 4), flags(np);
    std::vector<float> T(np * 16), ratio(np * 2), verts, cols;
    std::vector<int64_t> voff(np + 1, 0);
    std::vector<uint64_t> texloc(np);
    for (size_t p = 0; p < np; ++p) {
      for (int a = 0; a < 3; ++a) ids[3 * p + a] = meshes[p].chunkID(a);
      kf[p] = labels[p];
      std::memcpy(&T[16 * p], pose_inv[p], 64);
      verts.insert(verts.end(), meshes[p].vertices.begin(), meshes[p].vertices.end());
      cols.insert(cols.end(), meshes[p].colors.begin(), meshes[p].colors.end());
      voff[p + 1] = (int64_t)verts.size() / 3;
    }
    std::vector<float> tc((size_t)voff[np] * 2 + 2), tcol((size_t)voff[np] * 3 + 3);
    uint64_t hot[2] = {0, 0};
    int rc = tf_patches_update(vol, (int64_t)np, ids.data(), kf.data(), T.data(), voff.data(), verts.data(),
                               cols.data(), tc.data(), tcol.data(), bbox.data(), flags.data(), ratio.data(),
                               texloc.data(), hot);
    if (rc == TF_ERR_ATLAS_FULL) return -1;  // Chisel.cpp:170-173
    tf_check(rc, "GeneratePatches");
    out.assign(np, PatchResult());
    for (size_t p = 0; p < np; ++p) {
      PatchResult& r = out[p];
      r.texloc = texloc[p];
      std::memcpy(r.boundingbox, &bbox[4 * p], 16);
      r.flag = (flags[p] & 1) ? -1 : 0;
      r.wrong_mapping = (flags[p] & 2) != 0;
      r.ratio[0] = ratio[2 * p];
      r.ratio[1] = ratio[2 * p + 1];
      r.texcoord.assign(tc.begin() + 2 * voff[p], tc.begin() + 2 * voff[p + 1]);
      r.texcolor.assign(tcol.begin() + 3 * voff[p], tcol.begin() + 3 * voff[p + 1]);
      r.frameid = labels[p];
    }
    atlas.hot_start = hot[0];
    atlas.hot_end = hot[1];
    atlas.Refresh();
    return 0;
  }

  // Chisel::CompensateColor (Chisel.cpp:198-286) over the patches GeneratePatches produced, in the
  // caller's (mesh map) order: clusters the not yet adjusted patches by source frame, learns one
  // colour transfer per cluster from the correctly mapped patches and fills PatchResult::labs.
  void CompensateColor(const std::vector<PatchMesh>& meshes, std::vector<PatchResult>& patches) {
    const size_t np = patches.size();
    if (!np) return;
    std::vector<int32_t> fid(np);
    std::vector<uint8_t> wrong(np), adj(np);
    std::vector<int64_t> voff(np + 1, 0);
    std::vector<float> tex, mesh;
    for (size_t p = 0; p < np; ++p) {
      fid[p] = patches[p].frameid;
      wrong[p] = patches[p].wrong_mapping ? 1 : 0;
      adj[p] = patches[p].has_adjusted ? 1 : 0;
      tex.insert(tex.end(), patches[p].texcolor.begin(), patches[p].texcolor.end());
      mesh.insert(mesh.end(), meshes[p].colors.begin(), meshes[p].colors.end());
      voff[p + 1] = (int64_t)tex.size() / 3;
    }
    std::vector<float> labs(tex.size() + 3, 0.0f);
    int64_t ncl = 0;
    tf_check(tf_color_compensate(vol, (int64_t)np, fid.data(), wrong.data(), adj.data(), voff.data(), tex.data(),
                                 mesh.data(), labs.data(), &ncl), "CompensateColor");
    for (size_t p = 0; p < np; ++p) {
      if (patches[p].has_adjusted || !adj[p]) continue;  // skipped, or its cluster had nothing to learn from
      patches[p].has_adjusted = true;
      if (patches[p].wrong_mapping) patches[p].labs.clear();  // :276-278
      else patches[p].labs.assign(labs.begin() + 3 * voff[p], labs.begin() + 3 * voff[p + 1]);
    }
  }

  // Chisel::DrawMeshes (Chisel.cpp:288-355): 12 floats per vertex + rebased indices for the meshes
  // whose patch is complete() (here: a patch result exists, has texcoords and a source frame).
  void DrawMeshes(const std::vector<PatchMesh>& meshes, const std::vector<PatchResult>& patches, float* vertices,
                  unsigned int* indices, unsigned int& tsdf_indice_num, unsigned int& tsdf_vertice_num) {
    const size_t np = patches.size();
    tsdf_indice_num = tsdf_vertice_num = 0;
    if (!np) return;
    std::vector<uint8_t> complete(np), wrong(np), lv(np);
    std::vector<uint64_t> texloc(np);
    std::vector<float> ratio(2 * np), verts, cols, nrm, tc, tcol, labs;
    std::vector<int64_t> voff(np + 1, 0), ioff(np + 1, 0);
    std::vector<uint32_t> idx;
    for (size_t p = 0; p < np; ++p) {
      const PatchResult& r = patches[p];
      const PatchMesh& m = meshes[p];
      const size_t nv = m.vertices.size() / 3;
      complete[p] = (nv > 0 && r.texcoord.size() == 2 * nv && r.frameid >= 0) ? 1 : 0;  // Patch::complete (Patch.cpp:191-196)
      wrong[p] = r.wrong_mapping ? 1 : 0;
      lv[p] = (r.has_adjusted && r.labs.size() == 3 * nv) ? 1 : 0;
      texloc[p] = r.texloc;
      ratio[2 * p] = r.ratio[0];
      ratio[2 * p + 1] = r.ratio[1];
      verts.insert(verts.end(), m.vertices.begin(), m.vertices.end());
      cols.insert(cols.end(), m.colors.begin(), m.colors.end());
      nrm.insert(nrm.end(), m.normals.begin(), m.normals.end());
      nrm.resize(verts.size(), 0.0f);
      tc.insert(tc.end(), r.texcoord.begin(), r.texcoord.end());
      tc.resize(verts.size() / 3 * 2, 0.0f);
      tcol.insert(tcol.end(), r.texcolor.begin(), r.texcolor.end());
      tcol.resize(verts.size(), 0.0f);
      if (lv[p]) labs.insert(labs.end(), r.labs.begin(), r.labs.end());
      labs.resize(verts.size(), 0.0f);
      idx.insert(idx.end(), m.indices.begin(), m.indices.end());
      voff[p + 1] = (int64_t)verts.size() / 3;
      ioff[p + 1] = (int64_t)idx.size();
    }
    idx.resize(idx.size() + 1);
    int64_t nv_out = 0, ni_out = 0;
    tf_check(tf_pack_vertices(vol, (int64_t)np, complete.data(), wrong.data(), lv.data(), texloc.data(), ratio.data(),
                              voff.data(), verts.data(), cols.data(), nrm.data(), tc.data(), tcol.data(), labs.data(),
                              ioff.data(), idx.data(), vertices, indices, &nv_out, &ni_out), "DrawMeshes");
    tsdf_vertice_num = (unsigned int)nv_out;
    tsdf_indice_num = (unsigned int)ni_out;
  }

  ChunkID maxChunkID, minChunkID;
  ChunkManager chunkManager;
  ChunkSet meshesToUpdate;
  Atlas atlas;

 protected:
  void Configure(const ProjectionIntegrator& integ, const PinholeCamera& cam, bool with_integrator = true) {
    tf_check(tf_set_camera(vol, cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height, cam.nearPlane, cam.farPlane),
             "camera");
    if (with_integrator) {
      const QuadraticTruncator& t = *integ.GetTruncator();
      tf_check(tf_set_truncation(vol, t.quadraticTerm, t.linearTerm, t.constantTerm, t.scalingFactor), "truncator");
      tf_check(tf_set_weight(vol, integ.GetWeighter()->weight), "weighter");
    }
    const size_t want = (size_t)1 << 16;
    if (ids_buf.size() < want * 3) { ids_buf.resize(want * 3); new_buf.resize(want); }
  }
  void UploadFrame(const float* depth, const unsigned char* rgba, const float* quality, const PinholeCamera&) {
    tf_check(tf_frame_upload(vol, depth, rgba, quality), "frame upload");
  }
  void Flatten(const ChunkIDList& l) {
    if (ids_buf.size() < l.size() * 3) ids_buf.resize(l.size() * 3);
    for (size_t i = 0; i < l.size(); ++i) { ids_buf[3 * i] = l[i](0); ids_buf[3 * i + 1] = l[i](1); ids_buf[3 * i + 2] = l[i](2); }
  }
  tf_volume* vol = nullptr;
  float res = 0.005f;
  bool meshes_stale = true;
  std::vector<int32_t> ids_buf;
  std::vector<uint8_t> new_buf, needs_buf;
  std::vector<float> qual_buf;
};
typedef std::shared_ptr<Chisel> ChiselPtr;

}  // namespace chisel
