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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <algorithm>
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
  float& operator()(int i) { return v[i]; }
};
struct Vec4 {  // Eigen::Vector4f (geometry/Geometry.h); initChiselMap packs the truncator's terms into one
  float v[4];
  Vec4() : v{0, 0, 0, 0} {}
  Vec4(float a, float b, float c, float d) : v{a, b, c, d} {}
  float operator()(int i) const { return v[i]; }
  float& operator()(int i) { return v[i]; }
};
typedef std::vector<Vec3> Vec3List;

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

class Truncator {  // truncation/Truncator.h:28-40
 public:
  Truncator() = default;
  virtual ~Truncator() {}
  virtual float GetTruncationDistance(float depthReading) const = 0;
};
typedef std::shared_ptr<Truncator> TruncatorPtr;
typedef std::shared_ptr<const Truncator> TruncatorConstPtr;

class QuadraticTruncator : public Truncator {  // truncation/QuadraticTruncator.h:33-48
 public:
  QuadraticTruncator() = default;
  QuadraticTruncator(float quadratic, float linear, float constant, float scale)
      : quadraticTerm(quadratic), linearTerm(linear), constantTerm(constant), scalingFactor(scale) {}
  float GetTruncationDistance(float reading) const override {
    return std::abs(quadraticTerm * std::pow((double)reading, 2) + linearTerm * reading + constantTerm) * scalingFactor;
  }
  inline float GetQuadraticTerm() const { return quadraticTerm; }
  inline float GetLinearTerm() const { return linearTerm; }
  inline float GetConstantTerm() const { return constantTerm; }
  inline float GetScalingFactor() const { return scalingFactor; }
  inline void SetQuadraticTerm(float value) { quadraticTerm = value; }
  inline void SetLinearTerm(float value) { linearTerm = value; }
  inline void SetConstantTerm(float value) { constantTerm = value; }
  inline void SetScalingFactor(float value) { scalingFactor = value; }
  float quadraticTerm = 0.0019f, linearTerm = 0.00152f, constantTerm = 0.001504f, scalingFactor = 6.0f;
};
typedef std::shared_ptr<QuadraticTruncator> QuadraticTruncatorPtr;

class ConstantTruncator : public Truncator {  // truncation/ConstantTruncator.h:31-56
 public:
  ConstantTruncator() = default;
  ConstantTruncator(float value) : truncationDistance(value) {}
  inline void SetTruncationDistance(float value) { truncationDistance = value; }
  float GetTruncationDistance(float) const override { return truncationDistance; }
  float truncationDistance = 0.0f;
};
typedef std::shared_ptr<ConstantTruncator> ConstantTruncatorPtr;

class Weighter {  // weighting/Weighter.h:28-40
 public:
  Weighter() = default;
  virtual ~Weighter() {}
  virtual float GetWeight(float surfaceDist, float truncationDist) const = 0;
};
typedef std::shared_ptr<Weighter> WeighterPtr;
typedef std::shared_ptr<const Weighter> WeighterConstPtr;

class ConstantWeighter : public Weighter {  // weighting/ConstantWeighter.h:34-46
 public:
  ConstantWeighter() = default;
  ConstantWeighter(float w) : weight(w) {}
  float GetWeight(float, float truncationDist) const override { return weight / (2 * truncationDist); }
  float weight = 1.0f;
};
typedef std::shared_ptr<ConstantWeighter> ConstantWeighterPtr;

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
  // ProjectionIntegrator.h:81.  The reference keeps the un-rotated voxel centres here and re-derives the per-pose
  // table from them on every call (Chisel.cpp:52-110); on the device the selection role builds that table itself
  // from (pose, resolution), so the list is only held for callers that read it back.
  inline void SetCentroids(const Vec3List& c) { centroids = c; }
  inline const Vec3List& GetCentroids() const { return centroids; }

 protected:
  Vec3List centroids;
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

// ---- Mesh / Patch mirrors (3rd_party/open_chisel/geometry/Mesh.h:36-89, Structure/Patch.h:51-94) ----------
// The data lives in HBM; these carry the members the path's callers read.  Counts and flags are refreshed
// by the Chisel methods that change them; the per-vertex arrays are filled on request
// (ChunkManager::FetchMeshData / Chisel::FetchPatchData) because nothing on the path reads them on the host.
struct Box {  // cv::Rect
  int x = 0, y = 0, width = 0, height = 0;
};
struct Vec2 {
  float v[2];
  Vec2() : v{0, 0} {}
  Vec2(float a, float b) : v{a, b} {}
  float operator()(int i) const { return v[i]; }
  float& operator()(int i) { return v[i]; }
};
typedef std::vector<Vec2> Vec2List;
typedef std::vector<unsigned int> VertIndexList;

class Mesh;
typedef std::shared_ptr<Mesh> MeshPtr;

class Patch {  // Structure/Patch.h:51-94
 public:
  explicit Patch(Mesh* meshit) : mesh(meshit) {}
  int frameid = -1;
  Mesh* mesh = nullptr;
  std::size_t texloc = 0;
  bool has_image = false, has_adjusted = false, has_updated = false, wrong_mapping = false;
  Box boundingbox;
  Vec2List texcoord;
  Vec3List texcolor;
  Vec2 ratio = Vec2(1, 1);
  Vec3List labs;
  int n_texcoord = 0;     // texcoord.size() of the device-resident patch
  bool labs_empty = true;  // labs.empty() of the device-resident patch
  inline void SetFrameid(int frameit) { frameid = frameit; }
  inline float GetWidth() const { return (float)boundingbox.width; }
  inline float GetHeight() const { return (float)boundingbox.height; }
  inline int GetCoordsNum() const { return n_texcoord; }
  bool complete() const;  // Patch.cpp:191-196
};
typedef std::shared_ptr<Patch> PatchPtr;

class Mesh {  // geometry/Mesh.h:36-89
 public:
  inline bool HasVertices() const { return n_vertices > 0; }
  inline bool HasIndices() const { return n_indices > 0; }
  Vec3List vertices, normals, colors;  // filled by ChunkManager::FetchMeshData
  VertIndexList indices;
  ChunkID chunkID;
  bool simplified = false;
  bool adj[6] = {false, false, false, false, false, false};
  int n_vertices = 0, n_indices = 0;  // vertices.size() / indices.size() of the device-resident mesh
  PatchPtr m_patch;
};
inline bool Patch::complete() const {
  if (mesh == nullptr || mesh->n_vertices == 0 || !mesh->simplified) return false;
  if (!has_image || n_texcoord == 0 || frameid < 0) return false;
  return true;
}
typedef std::unordered_map<ChunkID, MeshPtr, ChunkHasher> MeshMap;

// GCSLAM/frame.h: the members of a keyframe the atlas stage reads (Patch.cpp:51,69-70,172-175)
struct Frame {
  int frame_index = -1;
  const unsigned char* rgb = nullptr;    // cv::Mat rgb, 8UC3, H x W
  const float* refined_depth = nullptr;  // cv::Mat refined_depth, 32F, H x W
  float pose_inv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // f32(pose_sophus[0].inverse().matrix()), row-major
};
// Structure/uni_graph.{h,cpp}: the chunk graph of the view selection -- nodes are chunks in insertion order, an
// undirected edge joins two face neighbours whose meshes touch across the face.  GeneratePatches only asks it for
// a chunk's node index and label (Chisel.cpp:159-160); TexMap below builds it.
class UniGraph {
 public:
  std::unordered_map<ChunkID, std::size_t, ChunkHasher> chunks;
  std::vector<int> labels;

  explicit UniGraph(std::size_t nodes = 0) : labels(nodes, 0), nbrs(nodes), n_edges(0) {}
  inline int get_label(std::size_t n) const { return labels[n]; }
  inline void set_label(std::size_t n, std::size_t label) { labels[n] = (int)label; }
  inline std::size_t num_nodes() const { return nbrs.size(); }
  inline std::size_t num_edges() const { return n_edges; }
  inline const std::vector<std::size_t>& get_adj_nodes(std::size_t n) const { return nbrs[n]; }

  bool add_node(const ChunkID& id) {  // uni_graph.cpp:22-28: the new node's index is the node count so far
    if (chunks.count(id)) return false;
    chunks.emplace(id, nbrs.size());
    nbrs.emplace_back();
    if (labels.size() < nbrs.size()) labels.resize(nbrs.size(), 0);
    return true;
  }
  bool has_edge(std::size_t a, std::size_t b) const {
    for (std::size_t x : nbrs[a]) if (x == b) return true;
    return false;
  }
  void add_edge(std::size_t a, std::size_t b) {  // uni_graph.h:110-117: no loops, no duplicates
    if (a == b || has_edge(a, b)) return;
    nbrs[a].push_back(b);
    nbrs[b].push_back(a);
    ++n_edges;
  }
  // uni_graph.cpp:41-50: for every face whose flag is set and whose neighbour chunk is a node
  void add_edge_by_node(const ChunkID& id, const bool flag[6]) {
    auto self = chunks.find(id);
    if (self == chunks.end()) return;
    static const int d[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};  // ChunkManager.h:55-57
    for (int k = 0; k < 6; ++k) {
      if (!flag[k]) continue;
      auto other = chunks.find(ChunkID(id(0) + d[k][0], id(1) + d[k][1], id(2) + d[k][2]));
      if (other != chunks.end()) add_edge(self->second, other->second);
    }
  }
  void remove_node(const ChunkID& id) {  // uni_graph.cpp:91-110: the node keeps its index, it loses its edges
    auto self = chunks.find(id);
    if (self == chunks.end()) return;
    const std::size_t n = self->second;
    for (std::size_t m : nbrs[n]) {
      std::vector<std::size_t>& l = nbrs[m];
      l.erase(std::remove(l.begin(), l.end(), n), l.end());
    }
    nbrs[n].clear();  // (the edge counter is left alone, as in the reference)
  }
  void clear() { nbrs.clear(); labels.clear(); chunks.clear(); n_edges = 0; }

 private:
  std::vector<std::vector<std::size_t>> nbrs;
  std::size_t n_edges;
};

// Structure/sparse_matrix.{h,cpp}: data costs, one ordered column (frame row -> quality) per graph node
class SparseMat {
 public:
  typedef std::map<std::size_t, float> Column;
  inline std::size_t cols() const { return columns.size(); }
  inline std::size_t rows() const { return row_len; }
  inline std::size_t get_nnz() const { return nnz; }
  inline const Column& col(std::size_t c) const { return columns[c]; }
  inline void resize(std::size_t c) { columns.resize(c); }
  bool add_value(std::size_t c, std::size_t r, float v) {  // sparse_matrix.cpp:27-36: an existing entry is kept
    grow(c, r);
    ++nnz;  // counted even when nothing is inserted, as in the reference
    return columns[c].emplace(r, v).second;
  }
  void set_value(std::size_t c, std::size_t r, float v) { grow(c, r); columns[c][r] = v; }  // :38-43
  void remove_observation(std::size_t c, std::size_t r) { if (c < columns.size()) columns[c].erase(r); }  // :45-50
  void remove_node(std::size_t c) { if (c < columns.size()) columns[c].clear(); }
  void clear() { columns.clear(); nnz = 0; row_len = 0; }

 private:
  void grow(std::size_t c, std::size_t r) {
    if (c >= columns.size()) columns.resize(c + 1);
    if (r >= row_len) row_len = r + 1;
  }
  std::vector<Column> columns;
  std::size_t nnz = 0, row_len = 0;
};
typedef SparseMat DataCosts;

inline void tf_check(int rc, const char* what) {
  if (rc != TF_OK) throw std::runtime_error(std::string(what) + ": " + tf_last_error());
}

// ---- ChunkManager: queries forward to the device volume; chunks are mirrored on demand -----
class ChunkManager {
 public:
  ChunkManager() = default;
  void Bind(tf_volume* v, float res) { vol = v; voxelResolutionMeters = res; CacheCentroids(); }
  inline float GetResolution() const { return voxelResolutionMeters; }
  // ChunkManager::CacheCentroids / GetCentroids (ChunkManager.cpp:49-63, ChunkManager.h:735): voxel centres of a chunk
  // relative to its origin, x fastest.  MobileFusion::initChiselMap hands them to the integrator (MobileFusion.h:242-243).
  void CacheCentroids() {
    const float r = voxelResolutionMeters, half = r * 0.5f;
    centroids.clear();
    for (int z = 0; z < 8; z++)
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) centroids.emplace_back((float)x * r + half, (float)y * r + half, (float)z * r + half);
  }
  inline const Vec3List& GetCentroids() const { return centroids; }
  inline const ChunkID& GetChunkSize() const { return chunkSize; }  // ChunkManager.h:732 (8 x 8 x 8: the kernel's only shape)
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
  // RemoveChunk also drops the chunk's mesh (ChunkManager.h:151-161)
  void DropMirror(const ChunkID& id) { mirrors.erase(id); allMeshes.erase(id); }
  void Reset() { mirrors.clear(); allMeshes.clear(); }  // ChunkManager.cpp:272-275

  // ---- meshes (ChunkManager.h:714-724) ----
  inline const MeshMap& GetAllMeshes() const { return allMeshes; }
  inline MeshMap& GetAllMutableMeshes() { return allMeshes; }
  inline const MeshPtr& GetMesh(const ChunkID& chunkID) const { return allMeshes.at(chunkID); }
  inline MeshPtr& GetMutableMesh(const ChunkID& chunkID) { return allMeshes.at(chunkID); }
  inline bool HasMesh(const ChunkID& chunkID) const { return allMeshes.find(chunkID) != allMeshes.end(); }
  // refresh counts / flags of the listed chunks' meshes from the device; chunks without a mesh there are
  // left out (RecomputeMeshes only adds, :260-262)
  void RefreshMeshes(const std::vector<int32_t>& ids3) {
    const int64_t n = (int64_t)ids3.size() / 3;
    if (!n) return;
    std::vector<int32_t> nv((size_t)n), ni((size_t)n);
    std::vector<uint8_t> adj((size_t)n * 6), simp((size_t)n);
    for (int64_t i = 0; i < n; ++i) {  // per chunk: a chunk of the dirty set may have no mesh
      const int rc = tf_mesh_counts(vol, &ids3[3 * (size_t)i], 1, &nv[(size_t)i], &ni[(size_t)i], &adj[6 * (size_t)i], &simp[(size_t)i]);
      if (rc == TF_ERR_MISSING_CHUNK) continue;
      tf_check(rc, "RefreshMeshes");
      const ChunkID id(ids3[3 * (size_t)i], ids3[3 * (size_t)i + 1], ids3[3 * (size_t)i + 2]);
      MeshPtr& m = allMeshes[id];
      if (!m) { m = std::make_shared<Mesh>(); m->chunkID = id; }
      m->n_vertices = nv[(size_t)i]; m->n_indices = ni[(size_t)i];
      m->simplified = simp[(size_t)i] != 0;
      for (int k = 0; k < 6; ++k) m->adj[k] = adj[6 * (size_t)i + k] != 0;
      m->vertices.clear(); m->normals.clear(); m->colors.clear(); m->indices.clear();
    }
  }
  // Mesh::vertices / normals / colors / indices of one mesh into the mirror
  void FetchMeshData(const ChunkID& id) {
    MeshPtr& m = allMeshes.at(id);
    const int64_t voff[2] = {0, m->n_vertices}, ioff[2] = {0, m->n_indices};
    std::vector<float> v((size_t)m->n_vertices * 3 + 3), nr(v.size()), c(v.size());
    m->indices.assign((size_t)m->n_indices, 0u);
    std::vector<uint32_t> idx((size_t)m->n_indices + 1);
    tf_check(tf_meshes_download(vol, id.v, 1, voff, ioff, v.data(), nr.data(), c.data(), idx.data()), "FetchMeshData");
    m->vertices.clear(); m->normals.clear(); m->colors.clear();
    for (int i = 0; i < m->n_vertices; ++i) {
      m->vertices.emplace_back(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
      m->normals.emplace_back(nr[3 * i], nr[3 * i + 1], nr[3 * i + 2]);
      m->colors.emplace_back(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
    }
    for (int i = 0; i < m->n_indices; ++i) m->indices[(size_t)i] = idx[(size_t)i];
  }

 private:
  tf_volume* vol = nullptr;
  float voxelResolutionMeters = 0.005f;
  ChunkID chunkSize = ChunkID(8, 8, 8);
  Vec3List centroids;
  std::unordered_map<ChunkID, ChunkPtr, ChunkHasher> mirrors;
  MeshMap allMeshes;
};

// ASCII PLY of a triangle soup (what io/PLY.cpp:29-80 SaveMeshPLYASCII leaves on disk for main.cpp:266): a header that
// declares float x y z (+ uchar red green blue when colours are given) per vertex and one "vertex_index" list per face,
// then n_corners vertex lines and n_corners / 3 face lines "3 a b c ".  The file declares numPoints / 3 faces, so the soup
// is written corner by corner: xyz(i) / rgb(i) return the three floats of corner i (rgb may be null: no colour
// properties); numbers in "%g" = an ostream's default float formatting, colour channels truncated from [0, 1] * 255.
template <class XyzFn, class RgbFn>
inline bool WritePlySoup(const std::string& fileName, size_t n_corners, XyzFn xyz, RgbFn rgb, bool with_rgb,
                         const unsigned int* corner_index, size_t n_index) {
  std::FILE* f = std::fopen(fileName.c_str(), "w");
  if (!f) return false;
  std::fprintf(f, "ply\nformat ascii 1.0\nelement vertex %zu\n", n_corners);
  for (const char* axis : {"x", "y", "z"}) std::fprintf(f, "property float %s\n", axis);
  if (with_rgb)
    for (const char* ch : {"red", "green", "blue"}) std::fprintf(f, "property uchar %s\n", ch);
  std::fprintf(f, "element face %zu\nproperty list uchar int vertex_index\nend_header\n", n_corners / 3);
  for (size_t i = 0; i < n_corners; ++i) {
    const float* p = xyz(i);
    std::fprintf(f, "%g %g %g", p[0], p[1], p[2]);
    if (with_rgb) {
      const float* c = rgb(i);
      std::fprintf(f, " %d %d %d", (int)(c[0] * 255.0f), (int)(c[1] * 255.0f), (int)(c[2] * 255.0f));
    }
    std::fputc('\n', f);
  }
  for (size_t t = 0; t + 2 < n_index; t += 3)
    std::fprintf(f, "3 %u %u %u \n", corner_index[t], corner_index[t + 1], corner_index[t + 2]);
  return std::fclose(f) == 0;
}
// the reference's entry point (io/PLY.h): one Mesh, its vertices as they are
inline bool SaveMeshPLYASCII(const std::string& fileName, const MeshPtr& mesh) {
  const Mesh& m = *mesh;
  return WritePlySoup(fileName, m.vertices.size(), [&m](size_t i) { return m.vertices[i].v; },
                      [&m](size_t i) { return m.colors[i].v; }, !m.colors.empty(), m.indices.data(), m.indices.size());
}

// A PNG file of an RGB8 image written with stored (uncompressed) deflate blocks -- what cv::imwrite produces for
// Atlas::SaveTexturedModel, minus the compression (no zlib dependency here).  rows(y) returns row y's w * 3 bytes.
template <class RowFn>
inline bool WritePngRgb(const std::string& fileName, uint32_t w, uint32_t h, RowFn rows) {
  std::FILE* f = std::fopen(fileName.c_str(), "wb");
  if (!f) return false;
  static uint32_t table[256];
  static bool have = false;
  if (!have) {
    for (uint32_t n = 0; n < 256; n++) {
      uint32_t c = n;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[n] = c;
    }
    have = true;
  }
  uint32_t crc = 0;
  auto crc_feed = [&](const unsigned char* p, size_t n) { for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8); };
  auto be32 = [](unsigned char* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
  auto put = [&](const unsigned char* p, size_t n) { std::fwrite(p, 1, n, f); crc_feed(p, n); };
  auto begin = [&](const char* type, uint32_t len) {
    unsigned char b[4]; be32(b, len); std::fwrite(b, 1, 4, f);
    crc = 0xFFFFFFFFu; put((const unsigned char*)type, 4);
  };
  auto end = [&]() { unsigned char b[4]; be32(b, crc ^ 0xFFFFFFFFu); std::fwrite(b, 1, 4, f); };
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::fwrite(sig, 1, 8, f);
  unsigned char ihdr[13];
  be32(ihdr, w); be32(ihdr + 4, h); ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  begin("IHDR", 13); put(ihdr, 13); end();
  // one IDAT chunk per image row: a piece of the zlib stream = stored blocks of <= 65535 bytes holding filter byte + row
  const size_t line = (size_t)w * 3 + 1;
  std::vector<unsigned char> buf(line);
  uint32_t a1 = 1, a2 = 0;  // Adler-32 of the raw stream
  for (uint32_t y = 0; y < h; y++) {
    buf[0] = 0;
    std::memcpy(&buf[1], rows(y), (size_t)w * 3);
    for (size_t i = 0; i < line;) {  // 5552 bytes keep the sums below 2^32
      const size_t n = std::min<size_t>(5552, line - i);
      for (size_t k = 0; k < n; k++) { a1 += buf[i + k]; a2 += a1; }
      a1 %= 65521u; a2 %= 65521u; i += n;
    }
    const size_t nblk = (line + 65534) / 65535;
    const bool first = y == 0, last = y + 1 == h;
    begin("IDAT", (uint32_t)((first ? 2 : 0) + line + 5 * nblk + (last ? 4 : 0)));
    if (first) { const unsigned char z[2] = {0x78, 0x01}; put(z, 2); }
    for (size_t b = 0; b < nblk; b++) {
      const size_t off = b * 65535, n = std::min<size_t>(65535, line - off);
      const unsigned char hd[5] = {(unsigned char)((last && b + 1 == nblk) ? 1 : 0), (unsigned char)(n & 0xFF), (unsigned char)(n >> 8),
                                   (unsigned char)(~n & 0xFF), (unsigned char)((~n >> 8) & 0xFF)};
      put(hd, 5); put(&buf[off], n);
    }
    if (last) { unsigned char ad[4]; be32(ad, (a2 << 16) | a1); put(ad, 4); }
    end();
  }
  begin("IEND", 0); end();
  return std::fclose(f) == 0;
}

// ---- Atlas (Structure/Atlas.h:43-75) -------------------------------------------------------
// the atlas size as the reference's callers spell it (Atlas.h:29-30; the volume's own size is tf_config.atlas_w / _h)
#ifndef MAX_PATCH_WIDTH
#define MAX_PATCH_WIDTH (96 * 72 * 2)
#endif
#ifndef MAX_PATCH_HEIGHT
#define MAX_PATCH_HEIGHT (72 * 96 * 2)
#endif
class Atlas {
 public:
  std::size_t loc_next = 0, PATCH_WIDTH = 0, PATCH_HEIGHT = 0, hot_start = 0, hot_end = 0;
  // Atlas::texture_buffer (Atlas.h:45: a cv::Mat of atlas_h x atlas_w RGB texels that the GUI thread uploads its hot rows
  // from, GCFusion/MobileFusion.h:404-421: `&chiselMap->atlas.texture_buffer.data[chiselMap->atlas.hot_start * 3]`).  The
  // texels live in HBM; this is their host mirror with the member the caller dereferences.  Chisel::UpdateAtlas refreshes
  // the rows [hot_start, hot_end) behind its device call, on the map thread -- so the GUI thread reads plain host memory
  // under the reference's own protocol (its `vertex_data_updated` flag), with no library call of its own.  Allocated on
  // first use; pages no hot range ever covered are never touched.
  struct TextureBuffer {
    unsigned char* data = nullptr;  // cv::Mat::data
    int rows = 0, cols = 0;         // cv::Mat::rows / cols
    std::size_t step = 0;           // bytes per row
    bool empty() const { return data == nullptr; }
    ~TextureBuffer() { std::free(data); }
    TextureBuffer() = default;
    TextureBuffer(const TextureBuffer&) = delete;
    TextureBuffer& operator=(const TextureBuffer&) = delete;
  } texture_buffer;
  // rows [row0, row1) of the device atlas into texture_buffer (whole rows: hot_start / hot_end are multiples of the width)
  void RefreshTextureRows(std::size_t row0, std::size_t row1) {
    if (row1 > MAX_PATCH_HEIGHT_) row1 = MAX_PATCH_HEIGHT_;
    if (row0 >= row1) return;
    if (!texture_buffer.data) {
      texture_buffer.rows = (int)MAX_PATCH_HEIGHT_; texture_buffer.cols = (int)MAX_PATCH_WIDTH_;
      texture_buffer.step = MAX_PATCH_WIDTH_ * 3;
      texture_buffer.data = static_cast<unsigned char*>(std::calloc(MAX_PATCH_HEIGHT_ * MAX_PATCH_WIDTH_, 3));  // Atlas.cpp:34-36
      if (!texture_buffer.data) throw std::bad_alloc();
    }
    DownloadRows((int64_t)row0, (int64_t)row1, texture_buffer.data + row0 * texture_buffer.step);
  }
  void RefreshHotRows() { RefreshTextureRows(hot_start / MAX_PATCH_WIDTH_, hot_end / MAX_PATCH_WIDTH_); }
  void Bind(tf_volume* v, ChunkManager* m) {
    vol = v;
    manager = m;
    int32_t pw = 0, ph = 0;
    tf_check(tf_atlas_patch_size(vol, &pw, &ph), "Atlas");
    PATCH_WIDTH = pw;
    PATCH_HEIGHT = ph;
    int32_t aw = 0, ah = 0;
    tf_check(tf_atlas_size(vol, &aw, &ah), "Atlas");
    MAX_PATCH_WIDTH_ = (std::size_t)aw;
    MAX_PATCH_HEIGHT_ = (std::size_t)ah;
  }
  std::size_t MAX_PATCH_WIDTH_ = 13824, MAX_PATCH_HEIGHT_ = 13824;  // the macros of Atlas.h:29-30 (tf_config.atlas_w / _h)
  inline Vec2 GetTexLoc(const ChunkID& id) {  // Atlas.cpp:66-69
    const std::size_t k = GetPatch(id)->texloc;
    return Vec2((float)(k % MAX_PATCH_WIDTH_), (float)(k / MAX_PATCH_WIDTH_));
  }
  // Atlas::SaveTexturedModel (Atlas.cpp:93-179) leaves three files in basepath:
  //   texture_material.png  the whole texture_buffer (RGB, as cv::imwrite stores the BGR-converted buffer),
  //   texture_model.obj     Wavefront OBJ: "v" / "vt" / "vn" per vertex of every mesh whose patch is complete(), vt =
  //                         (slot origin + texcoord * ratio) / atlas size with the v axis flipped, one group + face per
  //                         triangle with 1-based v/vt/vn triples, numbers with six decimals,
  //   texture_model.mtl     one material, "demo_texture", that maps the PNG.
  // Here the geometry is not collected mesh by mesh on the host: it is the packed stream DrawMeshes hands the renderer
  // (tf_draw_meshes, 12 floats per vertex: position [0..2], atlas uv [6..7], normal [8..10]; triangle corners rebased over
  // the model; meshes in ascending chunk id where the reference follows its unordered_map).
  void SaveTexturedModel(std::string const& basepath) {
    SaveTexturePng(basepath + "/texture_material.png");
    int64_t cap_v = 0, cap_i = 0;
    for (const auto& kv : manager->GetAllMeshes())
      if (kv.second->m_patch && kv.second->m_patch->complete()) { cap_v += kv.second->n_vertices; cap_i += kv.second->n_indices; }
    std::vector<float> packed((size_t)cap_v * 12 + 12);
    std::vector<uint32_t> corners((size_t)cap_i + 3);
    int64_t n_v = 0, n_i = 0;
    tf_check(tf_draw_meshes(vol, packed.data(), corners.data(), cap_v, cap_i, &n_v, &n_i), "Atlas::SaveTexturedModel");
    std::FILE* obj = std::fopen((basepath + "/texture_model.obj").c_str(), "w");
    if (!obj) throw std::runtime_error("Atlas::SaveTexturedModel: cannot write " + basepath + "/texture_model.obj");
    std::fputs("mtllib texture_model.mtl\n", obj);
    struct Column { const char* tag; int first, count; };
    for (const Column& col : {Column{"v", 0, 3}, Column{"vt", 6, 2}, Column{"vn", 8, 3}}) {
      for (int64_t k = 0; k < n_v; ++k) {
        const float* rec = &packed[(size_t)k * 12 + col.first];
        if (col.count == 2) std::fprintf(obj, "%s %.6f %.6f\n", col.tag, rec[0], 1.0f - rec[1]);  // image rows run downwards
        else std::fprintf(obj, "%s %.6f %.6f %.6f\n", col.tag, rec[0], rec[1], rec[2]);
      }
    }
    std::fputs("s off\nusemtl demo_texture\n", obj);
    for (int64_t t = 0; 3 * t + 2 < n_i; ++t) {
      std::fprintf(obj, "g face%lld\nf", (long long)t);
      for (int c = 0; c < 3; ++c) {
        const unsigned long ref = (unsigned long)corners[(size_t)(3 * t + c)] + 1ul;  // OBJ counts from one
        std::fprintf(obj, " %lu/%lu/%lu", ref, ref, ref);
      }
      std::fputc('\n', obj);
    }
    std::fclose(obj);
    std::FILE* mtl = std::fopen((basepath + "/texture_model.mtl").c_str(), "w");
    if (!mtl) throw std::runtime_error("Atlas::SaveTexturedModel: cannot write " + basepath + "/texture_model.mtl");
    std::fputs("newmtl demo_texture\n", mtl);
    for (const char* reflectance : {"Ka 1.000000 1.000000 1.000000", "Kd 1.000000 1.000000 1.000000", "Ks 0.000000 0.000000 0.000000"})
      std::fprintf(mtl, "%s\n", reflectance);
    std::fputs("Tr 0.000000\nillum 1\nNs 1.000000\nmap_Kd texture_material.png\n", mtl);
    std::fclose(mtl);
  }
  // the atlas as a PNG, downloaded in bands of 64 rows
  void SaveTexturePng(const std::string& file) {
    const std::size_t band = 64;
    std::vector<unsigned char> rows(band * MAX_PATCH_WIDTH_ * 3);
    std::size_t have0 = 1, have1 = 0;
    WritePngRgb(file, (uint32_t)MAX_PATCH_WIDTH_, (uint32_t)MAX_PATCH_HEIGHT_, [&](uint32_t y) {
      if (y < have0 || y >= have1) {
        have0 = y;
        have1 = std::min<std::size_t>(MAX_PATCH_HEIGHT_, have0 + band);
        DownloadRows((int64_t)have0, (int64_t)have1, rows.data());
      }
      return (const unsigned char*)&rows[(y - have0) * MAX_PATCH_WIDTH_ * 3];
    });
  }
  void BindOwner(class Chisel* c) { owner = c; }
  inline bool HasPatch(const ChunkID& id) const { return manager->HasMesh(id); }            // Atlas.h:55
  inline PatchPtr GetPatch(const ChunkID& id) { return manager->GetMutableMesh(id)->m_patch; }  // :56-58
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
  ChunkManager* manager = nullptr;
  class Chisel* owner = nullptr;
};

// ---- Chisel (Structure/Chisel.h:46-493) ----------------------------------------------------
class Chisel {
 public:
  Chisel(const int chunkSize[3], float voxelResolution, bool useColor, const tf_config* cfg = nullptr) {
    Init(chunkSize[0], chunkSize[1], chunkSize[2], voxelResolution, useColor, cfg);
  }
  // Chisel(const Eigen::Vector3i& chunkSize, float voxelResolution, bool useColor) (Chisel.cpp:38-41): any vector
  // type indexed with (i) -- Eigen::Vector3i in a caller that has Eigen, ChunkID here.
  template <class V3, class = decltype(std::declval<const V3&>()(0))>
  Chisel(const V3& chunkSize, float voxelResolution, bool useColor, const tf_config* cfg = nullptr) {
    Init(chunkSize(0), chunkSize(1), chunkSize(2), voxelResolution, useColor, cfg);
  }
  // Device-side sizing (pool, lists, atlas) of the volumes the three-argument constructor makes; nullptr = library defaults.
  static const tf_config*& DefaultConfig() { static const tf_config* c = nullptr; return c; }
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

  // Structure/Chisel.h:74-99: the eight corners of the box [minChunkID, maxChunkID] * chunk edge (debug drawing,
  // MobileFusion.cpp:170); min / maxChunkID are those of the last PrepareIntersectChunks
  void GetSearchRegion(float* corners, const PinholeCamera& depthCamera, const Transform& depthExtrinsic) {
    (void)depthCamera; (void)depthExtrinsic;
    const float e = 8.0f * chunkManager.GetResolution();
    const float minX = minChunkID(0) * e, minY = minChunkID(1) * e, minZ = minChunkID(2) * e;
    const float maxX = maxChunkID(0) * e, maxY = maxChunkID(1) * e, maxZ = maxChunkID(2) * e;
    const float vertexList[24] = {minX, minY, minZ, maxX, minY, minZ, minX, maxY, minZ, maxX, maxY, minZ,
                                  minX, minY, maxZ, maxX, minY, maxZ, minX, maxY, maxZ, maxX, maxY, maxZ};
    std::memcpy(corners, vertexList, sizeof(vertexList));
  }
  // Structure/Chisel.h:344-375: 8 corners x 3 floats per listed chunk (MobileFusion.h:505 draws them)
  void GetChunkCubes(std::vector<float>& cubes, ChunkIDList& chunksIntersecting) {
    cubes.clear();
    static const int off[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    const float r = chunkManager.GetResolution();
    const float edge = r * 8.0f;
    for (size_t i = 0; i < chunksIntersecting.size(); i++) {
      const ChunkID& c = chunksIntersecting[i];
      const float origin[3] = {8 * c(0) * r, 8 * c(1) * r, 8 * c(2) * r};
      for (int j = 0; j < 8; j++)
        for (int a = 0; a < 3; a++) cubes.emplace_back(origin[a] + (float)off[j][a] * edge);
    }
  }
  std::vector<float> candidateCubes;  // Chisel.h:101, refreshed by PrepareIntersectChunks (:129)

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
    GetChunkCubes(candidateCubes, chunksIntersecting);  // :129
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
    if (keyframeID >= 0) tf_check(tf_observations_record(vol, keyframeID), "observations");  // Chisel.h:244-247, device copy
    for (size_t i = 0; i < n; ++i) {
      needsUpdateFlag[i] = needs_buf[i] != 0;
      if (keyframeID >= 0 && qual_buf[i] > 0 && needsUpdateFlag[i])  // Chisel.h:244-247
        chunkManager.Mirror(chunksIntersecting[i])->observations[keyframeID] = qual_buf[i];
    }
  }

  // The local-frame loop of MobileFusion::TSDFFusion (GCFusion/MobileFusion.cpp:187-203) as one call: the depth-only
  // IntegrateDepthScanColor of up to six frames over the keyframe's chunk list, each frame with its own pose, applied
  // in order in one visit per chunk (same results as calling the overload above once per frame with colorImage = NULL).
  void IntegrateDepthScanGroup(ProjectionIntegrator& integrator, const std::vector<float*>& depthImages,
                               const std::vector<Transform>& depthExtrinsics, const PinholeCamera& depthCamera,
                               ChunkIDList& chunksIntersecting, std::vector<bool>& needsUpdateFlag, int integrate_flag) {
    Configure(integrator, depthCamera);
    const size_t n = chunksIntersecting.size(), nf = depthImages.size();
    if (n < 1 || nf < 1) return;
    Flatten(chunksIntersecting);
    needs_buf.resize(n);
    for (size_t i = 0; i < n; ++i) needs_buf[i] = needsUpdateFlag[i] ? 1 : 0;
    std::vector<float> poses(12 * nf);
    for (size_t f = 0; f < nf; ++f) std::memcpy(&poses[12 * f], depthExtrinsics[f].data(), 48);
    std::vector<const float*> dp(depthImages.begin(), depthImages.end());
    tf_check(tf_integrate_depth_group_host(vol, (int32_t)nf, dp.data(), poses.data(), ids_buf.data(), (int64_t)n, integrate_flag,
                                           needs_buf.data()),
             "IntegrateDepthScanGroup");
    for (size_t i = 0; i < n; ++i) needsUpdateFlag[i] = needs_buf[i] != 0;
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
    // the host-frames entry point: the images are staged (or, if the caller registered its buffers with RegisterHostBuffer,
    // uploaded in place) and the frame joins a four-deep launch pipeline -- a stream of IntegrateFrame calls runs at the
    // device's rate instead of one upload + three launches + a wait per frame; every other call flushes the pipeline
    // first, so the deferral is not observable (tf_fusion.h: tf_integrate_frame_host)
    tf_check(tf_integrate_frame_host(vol, depthImage, colorImage, depthExtrinsic.data(), nullptr, 0), "IntegrateDepthScanColor");
    meshes_stale = true;
  }
  // a caller whose images live in a fixed set of buffers (cv::Mat data, a camera ring) registers them once: host frames
  // then go up straight out of them (tf_host_register)
  void RegisterHostBuffer(const void* p, size_t bytes) { tf_check(tf_host_register(vol, p, (int64_t)bytes), "RegisterHostBuffer"); }
  void UnregisterHostBuffer(const void* p) { tf_check(tf_host_unregister(vol, p), "UnregisterHostBuffer"); }

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

  // Structure/Chisel.h:479-481 -> ChunkManager::RecomputeMeshes (ChunkManager.cpp:232-264) on the device
  void UpdateMeshes(const PinholeCamera& camera) {
    (void)camera;
    const ChunkSet& dirty = GetMeshesToUpdate();
    std::vector<int32_t> ids;
    for (const auto& kv : dirty)
      if (kv.second) { ids.push_back(kv.first(0)); ids.push_back(kv.first(1)); ids.push_back(kv.first(2)); }
    int64_t n = 0;
    tf_check(tf_update_meshes(vol, &n), "UpdateMeshes");
    chunkManager.RefreshMeshes(ids);
  }

  // Structure/Chisel.cpp:112-147.  The reference is handed chiselMap->meshesToUpdate and clears it.
  void CompressMeshes(ChunkSet& chunksToUpdate) {
    std::vector<int32_t> ids((size_t)chunksToUpdate.size() * 3 + 3);
    int64_t n = 0;
    tf_check(tf_compress_meshes(vol, ids.data(), (int64_t)chunksToUpdate.size(), &n), "CompressMeshes");
    ids.resize((size_t)n * 3);
    // flags of the listed meshes and of their face neighbours changed
    std::vector<int32_t> touch;
    static const int nb[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (int64_t i = 0; i < n; ++i)
      for (int k = 0; k < 7; ++k) {
        const ChunkID q(ids[3 * (size_t)i] + nb[k][0], ids[3 * (size_t)i + 1] + nb[k][1], ids[3 * (size_t)i + 2] + nb[k][2]);
        if (k == 0 || chunkManager.HasMesh(q)) { touch.push_back(q(0)); touch.push_back(q(1)); touch.push_back(q(2)); }
      }
    std::vector<int> keep_patch_state;  // RefreshMeshes leaves m_patch alone
    chunkManager.RefreshMeshes(touch);
    chunksToUpdate.clear();
    meshesToUpdate.clear();
    meshes_stale = false;
  }

  // Structure/Chisel.cpp:149-189.  frame_list is indexed by label like in the reference; the frames named by
  // the labels are cached in HBM (Frame::rgb / refined_depth) and their poses refreshed.
  int GeneratePatches(ChunkIDList& chunksToUpdate, UniGraph& labelset, std::vector<Frame>& frame_list,
                      PinholeCamera& cameraModel) {
    ProjectionIntegrator dummy;
    Configure(dummy, cameraModel, false);
    const size_t n = chunksToUpdate.size();
    std::vector<int32_t> ids(n * 3 + 3), labels(n + 1);
    for (size_t i = 0; i < n; ++i) {
      for (int a = 0; a < 3; ++a) ids[3 * i + a] = chunksToUpdate[i](a);
      const int frameid = labelset.get_label(labelset.chunks.find(chunksToUpdate[i])->second);
      labels[i] = frameid;
      Frame& f = frame_list[(size_t)frameid];
      if (!cached_kf.count(frameid) || cached_kf[frameid] != f.rgb) {
        tf_check(tf_keyframe_cache(vol, frameid, f.rgb, f.refined_depth), "GeneratePatches: keyframe");
        cached_kf[frameid] = f.rgb;
      }
      if (!pose_set.count(frameid) || std::memcmp(pose_set[frameid].data(), f.pose_inv, 64) != 0) {
        tf_check(tf_keyframe_set_pose(vol, frameid, f.pose_inv), "GeneratePatches: pose");
        pose_set[frameid].assign(f.pose_inv, f.pose_inv + 16);
      }
    }
    uint64_t hot[2] = {0, 0};
    const int rc = tf_generate_patches(vol, ids.data(), (int64_t)n, labels.data(), hot);
    if (rc != TF_ERR_ATLAS_FULL) tf_check(rc, "GeneratePatches");
    RefreshPatches(ids, n);
    atlas.hot_start = hot[0];
    atlas.hot_end = hot[1];
    atlas.Refresh();
    return rc == TF_ERR_ATLAS_FULL ? -1 : 0;  // Chisel.cpp:170-173
  }

  // Structure/Chisel.cpp:198-286
  void CompensateColor() {
    int64_t ncl = 0;
    tf_check(tf_compensate_color(vol, &ncl), "CompensateColor");
    std::vector<int32_t> ids;
    for (const auto& kv : chunkManager.GetAllMeshes())
      if (kv.second->m_patch) { ids.push_back(kv.first(0)); ids.push_back(kv.first(1)); ids.push_back(kv.first(2)); }
    RefreshPatches(ids, ids.size() / 3);
  }

  // Structure/Chisel.cpp:191-196
  void UpdateAtlas(ChunkIDList& chunksToUpdate) {
    const size_t n = chunksToUpdate.size();
    std::vector<int32_t> ids(n * 3 + 3);
    for (size_t i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) ids[3 * i + a] = chunksToUpdate[i](a);
    tf_check(tf_update_atlas(vol, ids.data(), (int64_t)n), "UpdateAtlas");
    RefreshPatches(ids, n);  // Patch::ratio is written by Atlas::UpdateBuffer
    atlas.RefreshHotRows();  // Atlas::texture_buffer: what the GUI thread uploads next (MobileFusion.h:404-421)
  }

  // Structure/Chisel.cpp:288-355.  The caller's buffers must hold the whole model, as in the reference
  // (MobileFusion.h:109-178 reserves 30 M vertices).
  void DrawMeshes(float* vertices, unsigned int* indices, unsigned int& tsdf_indice_num, unsigned int& tsdf_vertice_num,
                  int64_t cap_vertices = (int64_t)1 << 40, int64_t cap_indices = (int64_t)1 << 40) {
    int64_t nv = 0, ni = 0;
    tf_check(tf_draw_meshes(vol, vertices, indices, cap_vertices, cap_indices, &nv, &ni), "DrawMeshes");
    tsdf_vertice_num = (unsigned int)nv;
    tsdf_indice_num = (unsigned int)ni;
    for (auto& kv : chunkManager.GetAllMutableMeshes())
      if (kv.second->m_patch && kv.second->m_patch->complete()) kv.second->m_patch->has_updated = true;  // :351
  }

  // Patch::texcoord / texcolor / labs of one chunk into the mirror
  void FetchPatchData(const ChunkID& id) {
    MeshPtr& m = chunkManager.GetMutableMesh(id);
    if (!m->m_patch) return;
    const int64_t voff[2] = {0, m->n_vertices};
    std::vector<float> tc((size_t)m->n_vertices * 2 + 2), tcol((size_t)m->n_vertices * 3 + 3), labs(tcol.size());
    tf_check(tf_patches_download(vol, id.v, 1, voff, nullptr, nullptr, nullptr, nullptr, nullptr, tc.data(), tcol.data(),
                                 labs.data()), "FetchPatchData");
    Patch& p = *m->m_patch;
    p.texcoord.clear(); p.texcolor.clear(); p.labs.clear();
    for (int i = 0; i < p.n_texcoord; ++i) {
      p.texcoord.emplace_back(tc[2 * i], tc[2 * i + 1]);
      p.texcolor.emplace_back(tcol[3 * i], tcol[3 * i + 1], tcol[3 * i + 2]);
      if (!p.labs_empty) p.labs.emplace_back(labs[3 * i], labs[3 * i + 1], labs[3 * i + 2]);
    }
  }

  // Chisel::SaveAllMeshesToPLY (Chisel.cpp:357-379; main.cpp:266): the whole map as ONE triangle soup -- every triangle
  // corner becomes a vertex of its own (position + colour), faces are consecutive triples -- in a PLY file.  Mesh order is
  // the map's iteration order, as in the reference.
  bool SaveAllMeshesToPLY(const std::string& filename) {
    std::printf("Saving all meshes to PLY file...\n");
    std::vector<float> soup_xyz, soup_rgb;
    for (const auto& kv : chunkManager.GetAllMeshes()) {
      chunkManager.FetchMeshData(kv.first);
      const Mesh& m = *kv.second;
      for (const unsigned int corner : m.indices) {
        soup_xyz.insert(soup_xyz.end(), m.vertices[corner].v, m.vertices[corner].v + 3);
        soup_rgb.insert(soup_rgb.end(), m.colors[corner].v, m.colors[corner].v + 3);
      }
    }
    const size_t n_corners = soup_xyz.size() / 3;
    std::vector<unsigned int> ident(n_corners);
    for (size_t i = 0; i < n_corners; ++i) ident[i] = (unsigned int)i;
    std::printf("Full mesh has %lu verts\n", (unsigned long)n_corners);
    const bool ok = WritePlySoup(filename, n_corners, [&](size_t i) { return &soup_xyz[3 * i]; },
                                 [&](size_t i) { return &soup_rgb[3 * i]; }, true, ident.data(), n_corners);
    if (!ok) std::printf("Saving failed!\n");
    return ok;
  }

  ChunkID maxChunkID, minChunkID;
  ChunkManager chunkManager;
  ChunkSet meshesToUpdate;
  Atlas atlas;

 protected:
  void Init(int sx, int sy, int sz, float voxelResolution, bool useColor, const tf_config* cfg) {
    int32_t dims[3] = {sx, sy, sz};
    tf_check(tf_volume_create(dims, voxelResolution, useColor ? 1 : 0, cfg ? cfg : DefaultConfig(), &vol), "Chisel");
    res = voxelResolution;
    chunkManager.Bind(vol, res);
    atlas.Bind(vol, &chunkManager);
    atlas.BindOwner(this);
  }
  void Configure(const ProjectionIntegrator& integ, const PinholeCamera& cam, bool with_integrator = true) {
    tf_check(tf_set_camera(vol, cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height, cam.nearPlane, cam.farPlane),
             "camera");
    if (with_integrator) {
      // the device evaluates the two classes the reference ships (truncation/*.h, weighting/*.h)
      const Truncator* tb = integ.GetTruncator().get();
      if (const QuadraticTruncator* t = dynamic_cast<const QuadraticTruncator*>(tb))
        tf_check(tf_set_truncation(vol, t->quadraticTerm, t->linearTerm, t->constantTerm, t->scalingFactor), "truncator");
      else if (const ConstantTruncator* c = dynamic_cast<const ConstantTruncator*>(tb))
        tf_check(tf_set_truncation(vol, 0.0f, 0.0f, c->truncationDistance, 1.0f), "truncator");  // |0 + 0 + c| * 1
      else
        throw std::invalid_argument("ProjectionIntegrator: only QuadraticTruncator / ConstantTruncator run on the device");
      const ConstantWeighter* w = dynamic_cast<const ConstantWeighter*>(integ.GetWeighter().get());
      if (!w) throw std::invalid_argument("ProjectionIntegrator: only ConstantWeighter runs on the device");
      tf_check(tf_set_weight(vol, w->weight), "weighter");
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
  // Patch members of the listed chunks' meshes from the device (Mesh::m_patch is created by Atlas::AddPatch)
  void RefreshPatches(const std::vector<int32_t>& ids3, size_t n) {
    if (!n) return;
    std::vector<uint64_t> texloc(n);
    std::vector<int32_t> frameid(n), bbox(4 * n), flags(n);
    std::vector<float> ratio(2 * n);
    for (size_t i = 0; i < n; ++i) {  // per chunk: listed chunks without a mesh are skipped by the path
      const ChunkID id(ids3[3 * i], ids3[3 * i + 1], ids3[3 * i + 2]);
      if (!chunkManager.HasMesh(id)) continue;
      const int rc = tf_patches_download(vol, &ids3[3 * i], 1, nullptr, &texloc[i], &frameid[i], &bbox[4 * i], &flags[i],
                                         &ratio[2 * i], nullptr, nullptr, nullptr);
      if (rc == TF_ERR_MISSING_CHUNK) continue;
      tf_check(rc, "RefreshPatches");
      if (!(flags[i] & TF_PATCH_HAS_PATCH)) continue;
      MeshPtr& m = chunkManager.GetMutableMesh(id);
      if (!m->m_patch) m->m_patch = std::make_shared<Patch>(m.get());
      Patch& p = *m->m_patch;
      p.texloc = (std::size_t)texloc[i];
      p.frameid = frameid[i];
      p.boundingbox.x = bbox[4 * i]; p.boundingbox.y = bbox[4 * i + 1];
      p.boundingbox.width = bbox[4 * i + 2]; p.boundingbox.height = bbox[4 * i + 3];
      p.wrong_mapping = (flags[i] & TF_PATCH_WRONG_MAPPING) != 0;
      p.has_image = (flags[i] & TF_PATCH_HAS_IMAGE) != 0;
      p.has_adjusted = (flags[i] & TF_PATCH_HAS_ADJUSTED) != 0;
      p.ratio = Vec2(ratio[2 * i], ratio[2 * i + 1]);
      p.n_texcoord = p.has_image ? m->n_vertices : 0;
      p.labs_empty = !(p.has_adjusted && !p.wrong_mapping && p.n_texcoord > 0);
      p.texcoord.clear(); p.texcolor.clear(); p.labs.clear();
    }
  }
  tf_volume* vol = nullptr;
  float res = 0.005f;
  bool meshes_stale = true;
  std::unordered_map<int, const unsigned char*> cached_kf;
  std::unordered_map<int, std::vector<float>> pose_set;
  std::vector<int32_t> ids_buf;
  std::vector<uint8_t> new_buf, needs_buf;
  std::vector<float> qual_buf;
};
typedef std::shared_ptr<Chisel> ChiselPtr;

// Structure/TexMap.{h,cpp}: the bookkeeping of the view selection that consumes this path's outputs -- the chunk
// graph from the meshes' adjacency flags (update_chunkgraph, TexMap.cpp:50-62) and the data costs from the
// chunks' observation qualities (update_datacost, :64-105; check_graph, :107-119).  The MRF solve itself
// (view_selection, mapMAP) is third-party host code outside the path and stays where it is.
class TexMap {
 public:
  float adjacent_cost = 0.5f;  // TexMap.h:53-54
  float pairwise_cost = 1.0f;
  UniGraph chunkGraph;
  DataCosts dataCost;
  std::vector<float> statistic;

  void update_chunkgraph(ChunkIDList& chunksToUpdate, ChunkManager& chunkManager) {
    MeshMap& allMeshes = chunkManager.GetAllMutableMeshes();
    for (const ChunkID& id : chunksToUpdate) chunkGraph.add_node(id);
    for (const ChunkID& id : chunksToUpdate) {
      auto m = allMeshes.find(id);
      if (m != allMeshes.end()) chunkGraph.add_edge_by_node(id, m->second->adj);
    }
  }

  // lookup[frame] = row of that keyframe in the cost table; frameindex = the keyframe just fused, framesToUpdate
  // = keyframes whose observations may have been retracted or re-integrated since
  void update_datacost(ChunkIDList& chunksToUpdate, ChunkManager& chunkManager, std::vector<int>& lookup, int frameindex,
                       std::vector<int>& framesToUpdate) {
    for (const ChunkID& id : chunksToUpdate) {
      // (the reference calls GetChunk; only the host-side observation map is read, so the mirror is not refreshed
      // from the device here)
      ChunkPtr chunk = chunkManager.Mirror(id);
      const std::size_t node = chunkGraph.chunks.find(id)->second;
      if (statistic.size() < node + 1) statistic.resize(node + 1, 1.0f);  // (vector::resize only ever grows here)
      float quality = 0.0f;
      auto now = chunk->observations.find(frameindex);
      if (now != chunk->observations.end()) quality = now->second;
      if (quality > statistic[node]) statistic[node] = quality;
      if (quality > 0.0f) dataCost.add_value(node, (std::size_t)lookup[frameindex], quality);
      if (dataCost.cols() <= node) dataCost.resize(node + 1);
      for (int f : framesToUpdate) {
        const std::size_t row = (std::size_t)lookup[f];
        auto seen = chunk->observations.find(f);
        if (seen == chunk->observations.end()) {
          dataCost.remove_observation(node, row);
        } else {
          quality = seen->second;
          if (quality > statistic[node]) statistic[node] = quality;
          if (quality > 0.0f) dataCost.set_value(node, row, quality);
        }
      }
    }
  }

  // ---- the same two updates fed by the device's exports instead of the host mirrors (tf_export_adjacency /
  // tf_export_datacost: Mesh::adj and Chunk::observations stay in HBM, tf_observations_record / _retract keep the latter)
  int update_chunkgraph_device(ChunkIDList& chunksToUpdate, tf_volume* vol) {
    for (const ChunkID& id : chunksToUpdate) chunkGraph.add_node(id);
    const int64_t n = (int64_t)chunksToUpdate.size();
    if (!n) return TF_OK;
    std::vector<int32_t> ids((std::size_t)n * 3), edges((std::size_t)n * 6 * 4);
    for (int64_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) ids[(std::size_t)(3 * i + a)] = chunksToUpdate[(std::size_t)i](a);
    int64_t ne = 0;
    const int rc = tf_export_adjacency(vol, ids.data(), n, edges.data(), n * 6, &ne);
    if (rc) return rc;
    for (int64_t e = 0; e < ne; ++e) {
      const int32_t* r = &edges[(std::size_t)e * 4];
      auto a = chunkGraph.chunks.find(chunksToUpdate[(std::size_t)r[0]]);
      auto b = chunkGraph.chunks.find(ChunkID(r[1], r[2], r[3]));
      if (a != chunkGraph.chunks.end() && b != chunkGraph.chunks.end()) chunkGraph.add_edge(a->second, b->second);
    }
    return TF_OK;
  }
  int update_datacost_device(ChunkIDList& chunksToUpdate, tf_volume* vol, std::vector<int>& lookup, int frameindex,
                             std::vector<int>& framesToUpdate) {
    const int64_t n = (int64_t)chunksToUpdate.size();
    if (!n) return TF_OK;
    const std::size_t m = framesToUpdate.size();
    std::vector<int32_t> ids((std::size_t)n * 3), fr(framesToUpdate.begin(), framesToUpdate.end());
    for (int64_t i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) ids[(std::size_t)(3 * i + a)] = chunksToUpdate[(std::size_t)i](a);
    std::vector<float> tab((std::size_t)n * (1 + m));
    const int rc = tf_export_datacost(vol, ids.data(), n, frameindex, fr.empty() ? nullptr : fr.data(), (int32_t)m, tab.data());
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i) {  // TexMap.cpp:67-104 with "observation present" == table entry > 0
      const float* row = &tab[(std::size_t)i * (1 + m)];
      const std::size_t node = chunkGraph.chunks.find(chunksToUpdate[(std::size_t)i])->second;
      if (statistic.size() < node + 1) statistic.resize(node + 1, 1.0f);
      float quality = row[0];
      if (quality > statistic[node]) statistic[node] = quality;
      if (quality > 0.0f) dataCost.add_value(node, (std::size_t)lookup[frameindex], quality);
      if (dataCost.cols() <= node) dataCost.resize(node + 1);
      for (std::size_t j = 0; j < m; ++j) {
        const std::size_t r = (std::size_t)lookup[framesToUpdate[j]];
        quality = row[1 + j];
        if (!(quality > 0.0f)) { dataCost.remove_observation(node, r); continue; }
        if (quality > statistic[node]) statistic[node] = quality;
        dataCost.set_value(node, r, quality);
      }
    }
    return TF_OK;
  }

  void check_graph(ChunkManager& chunkManager) {  // nodes whose mesh is gone lose their edges and costs
    const MeshMap& allMeshes = chunkManager.GetAllMeshes();
    for (const auto& it : chunkGraph.chunks) {
      if (allMeshes.find(it.first) != allMeshes.end()) continue;
      chunkGraph.remove_node(it.first);
      dataCost.remove_node(it.second);
    }
  }

  void clear() { chunkGraph.clear(); dataCost.clear(); statistic.clear(); }
};
typedef TexMap* TexPtr;

}  // namespace chisel

// Structure/uni_graph.h, sparse_matrix.h and TexMap.h declare these at global scope
using chisel::DataCosts;
using chisel::SparseMat;
using chisel::TexMap;
using chisel::UniGraph;

// ---------------------------------------------------------------------------------------------------------
// BasicAPI's per-frame image passes (BasicAPI.h:112-131, called from main.cpp:117-147) with the reference's
// names and argument order.  The reference hands over cv::Mat headers of host images; here the images stay in
// device memory (depth / weight / quality f32[H][W], normal maps planar f32[3][H][W], RGB u8[H][W][3]) and the
// intrinsics are the ones the volume's camera was given (PinholeCamera::SetIntrinsics).
// ---------------------------------------------------------------------------------------------------------
namespace BasicAPI {

struct DeviceFrame {  // the image members of ::Frame the passes touch (Frame.h), as device pointers
  float* refined_depth = nullptr;           // Frame::refined_depth
  float* weight = nullptr;                  // Frame::weight
  float* normal_map = nullptr;              // Frame::normal_map (planar)
  unsigned char* rgb = nullptr;             // Frame::rgb
  unsigned char* colorValidFlag = nullptr;  // Frame::colorValidFlag
  float* observationQualityMap = nullptr;   // Frame::observationQualityMap
};

inline void extractNormalMapSIMD(chisel::Chisel& map, const float* depthMap, float* normalMap) {
  chisel::tf_check(tf_pre_normal_map(map.Handle(), depthMap, normalMap), "extractNormalMapSIMD");
}
inline void refineDepthUseNormalSIMD(chisel::Chisel& map, float* normal, float* depth) {
  chisel::tf_check(tf_pre_refine_depth_normal(map.Handle(), normal, depth), "refineDepthUseNormalSIMD");
}
// T_ref_from_new = float((pose_ref^-1 * pose_new).matrix()) rows 0..2, row-major (BasicAPI.cpp:402-406)
inline void refineNewframesSIMD(chisel::Chisel& map, DeviceFrame& frame_ref, DeviceFrame& frame_new,
                                const float T_ref_from_new[12]) {
  chisel::tf_check(tf_pre_refine_newframe(map.Handle(), frame_ref.refined_depth, frame_new.refined_depth, T_ref_from_new),
                   "refineNewframesSIMD");
}
// T_new_from_ref = float((pose_new^-1 * pose_ref).matrix()) rows 0..2 (BasicAPI.cpp:528-533)
inline void refineKeyframesSIMD(chisel::Chisel& map, DeviceFrame& frame_ref, DeviceFrame& frame_new,
                                const float T_new_from_ref[12]) {
  chisel::tf_check(tf_pre_refine_keyframe(map.Handle(), frame_ref.refined_depth, frame_ref.weight, frame_new.refined_depth,
                                          T_new_from_ref, nullptr),
                   "refineKeyframesSIMD");
}
inline void checkColorQuality(chisel::Chisel& map, const float* normalMap, unsigned char* validColorFlag) {
  chisel::tf_check(tf_pre_color_valid(map.Handle(), normalMap, validColorFlag), "checkColorQuality");
}
inline void estimateColorQuality(chisel::Chisel& map, const float* depthMap, const float* normalMap, float* qualityMap,
                                 const unsigned char* rgb) {
  chisel::tf_check(tf_pre_color_quality(map.Handle(), depthMap, normalMap, rgb, qualityMap), "estimateColorQuality");
}

// DatasetWrapper::framePreprocess (Tools/DatasetWrapper.hpp:186-263): depth = Frame::depth (u16, device, in place),
// frame.refined_depth receives the filtered metres; bilateralFilterRange 9 (7 on MobileCPU builds), 0.03, 10 (:226-232).
// (Frame::weight = 0 of :220 is the caller's hipMemsetAsync.)
inline void framePreprocess(chisel::Chisel& map, unsigned short* depth, DeviceFrame& frame, float maximum_depth,
                            float depth_scale, int bilateralFilterRange = 9) {
  chisel::tf_check(tf_pre_frame_depth(map.Handle(), depth, frame.refined_depth, maximum_depth, depth_scale,
                                      bilateralFilterRange, 0.03, 10.0),
                   "framePreprocess");
}

}  // namespace BasicAPI
