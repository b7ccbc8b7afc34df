// Drives the host-side C++ mirror (texturefusion_amd/host/tf_chisel.hpp) the way
// GCFusion/MobileFusion.cpp drives the reference classes (ReIntegrateKeyframe :114-221,
// IntegrateFrame :223-250) and checks every result against the CPU oracle (test infrastructure).
// Plain C++14, no Eigen/OpenCV; built with g++ against libtexfusion_hip.so and libtf_oracle.so.
#include <algorithm>
#include <iostream>
#include <fstream>
#include <sstream>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <set>
#include <vector>

#include "../../oracle/tf_oracle.h"
#include "../../texturefusion_amd/host/tf_chisel.hpp"

#define CHECK(c)                                                                   \
  do {                                                                             \
    if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } \
  } while (0)

// The configuration MobileFusion::initChiselMap makes (GCFusion/MobileFusion.h:205-258), as a call sequence written for
// this test: same constructor, same setters, same constants.  That the reference's OWN text of that function compiles
// against tf_chisel.hpp is checked by tests/test_ref_pin.py::test_init_chisel_map_compiles_against_the_mirror, which cuts
// it out of /root/reference at test time (build container only); no reference text lives in this file.
struct MobileFusion {
  chisel::ChiselPtr chiselMap;
  chisel::ProjectionIntegrator projectionIntegrator;
  chisel::PinholeCamera cameraModel;

  void configure(float fx, float fy, float cx, float cy, int width, int height, float voxel, float far_plane) {
    const chisel::ChunkID chunk_dim(8, 8, 8);
    chiselMap = chisel::ChiselPtr(new chisel::Chisel(chunk_dim, voxel, /*useColor=*/true));
    projectionIntegrator.SetCentroids(chiselMap->GetChunkManager().GetCentroids());
    projectionIntegrator.SetTruncator(chisel::TruncatorPtr(new chisel::QuadraticTruncator(0.0019f, 0.00152f, 0.001504f, 6.0f)));
    projectionIntegrator.SetWeighter(chisel::WeighterPtr(new chisel::ConstantWeighter(1.0f)));
    projectionIntegrator.SetCarvingDist(0.05f);
    projectionIntegrator.SetCarvingEnabled(true);
    cameraModel.SetIntrinsics(fx, fy, cx, cy);
    cameraModel.SetNearPlane(0.01f);
    cameraModel.SetFarPlane(far_plane);
    cameraModel.SetWidth(width);
    cameraModel.SetHeight(height);
  }
};

static void make_frame(int W, int H, float z, int seed, std::vector<float>& depth,
                       std::vector<unsigned char>& rgba, std::vector<float>& quality) {
  depth.assign((size_t)W * H, z);
  rgba.resize((size_t)W * H * 4);
  quality.resize((size_t)W * H);
  for (int i = 0; i < W * H; ++i) {
    if ((i + seed) % 53 == 0) depth[i] = 0.0f;  // holes
    unsigned h = (unsigned)i * 2654435761u + (unsigned)seed * 97u;
    rgba[4 * i] = h & 0xFF; rgba[4 * i + 1] = (h >> 8) & 0xFF; rgba[4 * i + 2] = (h >> 16) & 0xFF; rgba[4 * i + 3] = 1;
    quality[i] = (float)((h >> 24) & 0xFF) / 256.0f;
  }
}

static void compare_chunk(chisel::Chisel& ch, tfo_volume* ov, const chisel::ChunkID& id) {
  float sdf[512], w[512];
  uint16_t col[2048];
  int cid[3] = {id(0), id(1), id(2)};
  CHECK(tfo_volume_get_chunk(ov, cid, sdf, w, col) == 0);
  chisel::ChunkPtr c = ch.GetMutableChunkManager().GetChunk(id);
  CHECK(std::memcmp(sdf, c->voxels.sdf.data(), sizeof(sdf)) == 0);
  CHECK(std::memcmp(w, c->voxels.weight.data(), sizeof(w)) == 0);
  CHECK(std::memcmp(col, c->colors.colorData.data(), sizeof(col)) == 0);
}

int main() {
  const int W = 640, H = 480;
  const float res = 0.005f;
  tf_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.max_chunks = 1 << 15;
  cfg.atlas_h = 72;
  chisel::Chisel::DefaultConfig() = &cfg;  // device-side sizing of the volume the reference's 3-argument constructor makes

  // what MobileFusion::initChiselMap does (GCFusion/MobileFusion.h:205-258)
  MobileFusion gcFusion;
  gcFusion.configure(525.0f, 525.0f, 319.5f, 239.5f, W, H, res, 5.0f);
  chisel::Chisel& chiselMap = *gcFusion.chiselMap;
  chisel::ProjectionIntegrator& projectionIntegrator = gcFusion.projectionIntegrator;
  chisel::PinholeCamera& cameraModel = gcFusion.cameraModel;
  CHECK(projectionIntegrator.GetCentroids().size() == 512);
  {
    float cen[3 * 512];
    const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    tfo_centroids(ident, res, cen);  // identity pose: the per-pose table IS ChunkManager::centroids
    for (int i = 0; i < 512; ++i)
      for (int a = 0; a < 3; ++a) CHECK(projectionIntegrator.GetCentroids()[(size_t)i](a) == cen[a * 512 + i]);
  }

  tfo_volume* ov = tfo_volume_create(res, 1);

  std::vector<float> depth, quality, depth2, q2;
  std::vector<unsigned char> rgba, rgba2;
  make_frame(W, H, 1.5f, 0, depth, rgba, quality);
  make_frame(W, H, 1.45f, 7, depth2, rgba2, q2);
  chisel::Transform lastPose;  // identity
  lastPose(0, 3) = 0.01f;
  const int kfIndex = 4;

  // ---- ReIntegrateKeyframe(..., integrateFlag = 1)
  chisel::ChunkIDList localChunksIntersecting, validChunks;
  std::vector<bool> localNeedsUpdateFlag, localNewChunkFlag;
  chiselMap.PrepareIntersectChunks(projectionIntegrator, depth.data(), lastPose, cameraModel,
                                   localChunksIntersecting, localNeedsUpdateFlag, localNewChunkFlag);
  const size_t n = localChunksIntersecting.size();
  CHECK(n > 1000);
  {  // Chisel::candidateCubes (Chisel.h:101,129) and GetSearchRegion (:74-99): debug geometry of the selection
    CHECK(chiselMap.candidateCubes.size() == 24 * n);
    const chisel::ChunkID c0 = localChunksIntersecting[0];
    CHECK(chiselMap.candidateCubes[0] == 8 * c0(0) * res && chiselMap.candidateCubes[21] == 8 * c0(0) * res + res * 8.0f);
    float corners[24];
    chiselMap.GetSearchRegion(corners, cameraModel, lastPose);
    CHECK(corners[0] == chiselMap.minChunkID(0) * 8.0f * res && corners[23] == chiselMap.maxChunkID(2) * 8.0f * res);
    CHECK(chiselMap.GetChunkManager().GetChunkSize()(0) == 8);
  }
  std::vector<int32_t> oids(n * 3 + 3);
  std::vector<uint8_t> onew(n + 1), oneeds(n + 1, 0);
  CHECK(tfo_prepare(ov, depth.data(), lastPose.data(), oids.data(), onew.data(), (int64_t)n) == (int64_t)n);
  for (size_t i = 0; i < n; ++i) {
    CHECK(oids[3 * i] == localChunksIntersecting[i](0) && oids[3 * i + 1] == localChunksIntersecting[i](1) &&
          oids[3 * i + 2] == localChunksIntersecting[i](2));
    CHECK((onew[i] != 0) == localNewChunkFlag[i]);
  }
  chiselMap.IntegrateDepthScanColor(projectionIntegrator, depth.data(), rgba.data(), lastPose, cameraModel,
                                    localChunksIntersecting, localNeedsUpdateFlag, 1, kfIndex, quality.data());
  std::vector<float> oq(n + 1);
  CHECK(tfo_integrate(ov, depth.data(), rgba.data(), quality.data(), lastPose.data(), oids.data(), (int64_t)n, 1,
                      kfIndex, oneeds.data(), oq.data()) == 0);
  // one depth-only local frame on the same list (MobileFusion.cpp:187-203)
  chiselMap.IntegrateDepthScanColor(projectionIntegrator, depth2.data(), NULL, lastPose, cameraModel,
                                    localChunksIntersecting, localNeedsUpdateFlag, 1);
  CHECK(tfo_integrate(ov, depth2.data(), NULL, NULL, lastPose.data(), oids.data(), (int64_t)n, 1, -1, oneeds.data(),
                      oq.data()) == 0);
  // two more local frames in ONE call (IntegrateDepthScanGroup = the same loop body, one visit per chunk)
  {
    std::vector<float> d3, d4, qx;
    std::vector<unsigned char> cx;
    make_frame(W, H, 1.47f, 11, d3, cx, qx);
    make_frame(W, H, 1.52f, 12, d4, cx, qx);
    chisel::Transform p3 = lastPose, p4 = lastPose;
    p3(0, 3) = 0.012f; p4(1, 3) = -0.004f;
    std::vector<float*> imgs{d3.data(), d4.data()};
    std::vector<chisel::Transform> poses{p3, p4};
    chiselMap.IntegrateDepthScanGroup(projectionIntegrator, imgs, poses, cameraModel, localChunksIntersecting,
                                      localNeedsUpdateFlag, 1);
    CHECK(tfo_integrate(ov, d3.data(), NULL, NULL, p3.data(), oids.data(), (int64_t)n, 1, -1, oneeds.data(), oq.data()) == 0);
    CHECK(tfo_integrate(ov, d4.data(), NULL, NULL, p4.data(), oids.data(), (int64_t)n, 1, -1, oneeds.data(), oq.data()) == 0);
  }
  for (size_t i = 0; i < n; ++i) CHECK((oneeds[i] != 0) == localNeedsUpdateFlag[i]);
  chiselMap.FinalizeIntegrateChunks(localChunksIntersecting, localNeedsUpdateFlag, localNewChunkFlag, validChunks);
  std::vector<int32_t> ovalid(n * 3 + 3);
  const int64_t nv = tfo_finalize(ov, oids.data(), oneeds.data(), onew.data(), (int64_t)n, ovalid.data());
  CHECK((size_t)nv == validChunks.size());
  size_t nobs = 0;
  for (int64_t i = 0; i < nv; ++i) {
    CHECK(ovalid[3 * i] == validChunks[i](0) && ovalid[3 * i + 1] == validChunks[i](1) && ovalid[3 * i + 2] == validChunks[i](2));
    if (i % 37 == 0) compare_chunk(chiselMap, ov, validChunks[i]);
    int32_t kf[8];
    float qq[8];
    int cid[3] = {validChunks[i](0), validChunks[i](1), validChunks[i](2)};
    int64_t no = tfo_volume_get_observations(ov, cid, kf, qq, 8);
    chisel::ChunkPtr m = chiselMap.GetMutableChunkManager().Mirror(validChunks[i]);
    CHECK((size_t)no == m->observations.size());
    if (no) { CHECK(m->observations.count(kfIndex) && m->observations[kfIndex] == qq[0]); ++nobs; }
  }
  CHECK(nobs > 100);
  CHECK((int64_t)chiselMap.GetMeshesToUpdate().size() == tfo_volume_num_dirty(ov));
  CHECK((int64_t)chiselMap.GetChunkManager().GetChunkIDs().size() == tfo_volume_num_chunks(ov));
  // a garbage-collected chunk is gone on both sides
  for (size_t i = 0; i < n; ++i)
    if (!localNeedsUpdateFlag[i] && localNewChunkFlag[i]) {
      CHECK(!chiselMap.GetChunkManager().HasChunk(localChunksIntersecting[i]));
      break;
    }

  // ---- MobileFusion::IntegrateFrame: the fused 5-argument unit
  chiselMap.IntegrateDepthScanColor(projectionIntegrator, depth2.data(), rgba2.data(), lastPose, cameraModel);
  int64_t nsel = 0;
  tfo_integrate_frame(ov, depth2.data(), rgba2.data(), lastPose.data(), &nsel);
  CHECK((int64_t)chiselMap.GetChunkManager().GetChunkIDs().size() == tfo_volume_num_chunks(ov));
  for (int64_t i = 0; i < nv; i += 53) compare_chunk(chiselMap, ov, validChunks[i]);

  // ---- ReIntegrateKeyframe(..., integrateFlag = 0): de-integrate over kf.validChunks
  std::vector<bool> deNeeds(validChunks.size(), true), deNew(validChunks.size(), false);
  chiselMap.IntegrateDepthScanColor(projectionIntegrator, depth.data(), rgba.data(), lastPose, cameraModel,
                                    validChunks, deNeeds, 0, kfIndex, quality.data());
  std::vector<uint8_t> dneeds((size_t)nv, 1);
  CHECK(tfo_integrate(ov, depth.data(), rgba.data(), quality.data(), lastPose.data(), ovalid.data(), nv, 0, kfIndex,
                      dneeds.data(), oq.data()) == 0);
  for (int64_t i = 0; i < nv; i += 29) compare_chunk(chiselMap, ov, validChunks[i]);

  // ---- GetChunk of an absent chunk throws like chunks.at()
  bool threw = false;
  try { chiselMap.GetMutableChunkManager().GetChunk(chisel::ChunkID(9999, 9999, 9999)); } catch (const std::out_of_range&) { threw = true; }
  CHECK(threw);

  // ---- the rest of MobileFusion::tsdfFusion (GCFusion/MobileFusion.cpp:327-384) on real meshes: a few more
  // frames so that voxel weights pass the mesher's threshold, then UpdateMeshes, chunksToUpdate, CompressMeshes,
  // GeneratePatches (labels from a stand-in for the MRF), CompensateColor, UpdateAtlas, DrawMeshes
  size_t n_patches = 0;
  {
    for (int k = 0; k < 7; ++k) {
      std::vector<float> d, q;
      std::vector<unsigned char> c;
      make_frame(W, H, 1.5f, 20 + k, d, c, q);
      chiselMap.IntegrateDepthScanColor(projectionIntegrator, d.data(), c.data(), lastPose, cameraModel);
      tfo_integrate_frame(ov, d.data(), c.data(), lastPose.data(), NULL);
    }
    chiselMap.UpdateMeshes(cameraModel);
    tfo_update_meshes(ov);
    const chisel::MeshMap& allMeshes = chiselMap.GetChunkManager().GetAllMeshes();
    CHECK((int64_t)allMeshes.size() == tfo_volume_num_meshes(ov));
    // chunksToUpdate (MobileFusion.cpp:345-353); the reference's order is its unordered_map's, the test sorts
    chisel::ChunkIDList chunksToUpdate;
    for (const auto& it : chiselMap.GetMeshesToUpdate()) {
      if (!it.second) continue;
      if (allMeshes.find(it.first) == allMeshes.end()) continue;
      chunksToUpdate.emplace_back(it.first);
    }
    std::sort(chunksToUpdate.begin(), chunksToUpdate.end(), [](const chisel::ChunkID& x, const chisel::ChunkID& y) {
      for (int k = 0; k < 3; ++k) if (x(k) != y(k)) return x(k) < y(k);
      return false;
    });
    chiselMap.CompressMeshes(chiselMap.meshesToUpdate);
    n_patches = chunksToUpdate.size();
    CHECK(n_patches > 300);
    std::vector<int32_t> oids2(n_patches * 3 + 3);
    CHECK(tfo_compress_meshes(ov, oids2.data(), (int64_t)n_patches) == (int64_t)n_patches);
    for (size_t i = 0; i < n_patches; ++i)
      CHECK(oids2[3 * i] == chunksToUpdate[i](0) && oids2[3 * i + 1] == chunksToUpdate[i](1) && oids2[3 * i + 2] == chunksToUpdate[i](2));
    CHECK(chiselMap.GetMeshesToUpdate().empty() && tfo_volume_num_dirty(ov) == 0);
    // mesh mirrors: counts, flags, and the arrays of a few meshes
    for (size_t i = 0; i < n_patches; i += 17) {
      const chisel::ChunkID id = chunksToUpdate[i];
      chiselMap.GetMutableChunkManager().FetchMeshData(id);
      const chisel::MeshPtr& m = chiselMap.GetChunkManager().GetMesh(id);
      int64_t onv = 0, oni = 0;
      uint8_t oadj[6];
      int osimp = 0;
      std::vector<float> ovx(3 * 2187), onr(3 * 2187), ocl(3 * 2187);
      std::vector<uint32_t> oix(7680);
      int cid[3] = {id(0), id(1), id(2)};
      CHECK(tfo_volume_get_mesh(ov, cid, &onv, &oni, ovx.data(), onr.data(), ocl.data(), oix.data(), oadj, &osimp) == 0);
      CHECK(m->n_vertices == onv && m->n_indices == oni && m->simplified == (osimp != 0));
      for (int k = 0; k < 6; ++k) CHECK(m->adj[k] == (oadj[k] != 0));
      CHECK(std::memcmp(m->vertices.data(), ovx.data(), (size_t)onv * 12) == 0);
      CHECK(std::memcmp(m->normals.data(), onr.data(), (size_t)onv * 12) == 0);
      CHECK(std::memcmp(m->colors.data(), ocl.data(), (size_t)onv * 12) == 0);
      CHECK(std::memcmp(m->indices.data(), oix.data(), (size_t)oni * 4) == 0);
    }
    // ---- TexMap's bookkeeping over the mirror (Structure/TexMap.cpp:50-119: chunk graph from Mesh::adj, data costs
    // from Chunk::observations) against the same facts read from the oracle and combined independently
    {
      chisel::TexMap texmap;
      chisel::ChunkManager& cm = chiselMap.GetMutableChunkManager();
      texmap.update_chunkgraph(chunksToUpdate, cm);
      CHECK(texmap.chunkGraph.num_nodes() == n_patches);
      std::map<std::array<int, 3>, size_t> index;
      for (size_t i = 0; i < n_patches; ++i) {
        index[{chunksToUpdate[i](0), chunksToUpdate[i](1), chunksToUpdate[i](2)}] = i;
        CHECK(texmap.chunkGraph.chunks.find(chunksToUpdate[i])->second == i);
      }
      static const int dd[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
      std::set<std::pair<size_t, size_t>> edges;
      std::vector<float> ovx(3 * 2187), onr(3 * 2187), ocl(3 * 2187);
      std::vector<uint32_t> oix(7680);
      for (size_t i = 0; i < n_patches; ++i) {
        int cid[3] = {chunksToUpdate[i](0), chunksToUpdate[i](1), chunksToUpdate[i](2)};
        int64_t onv = 0, oni = 0;
        uint8_t oadj[6];
        int osimp = 0;
        CHECK(tfo_volume_get_mesh(ov, cid, &onv, &oni, ovx.data(), onr.data(), ocl.data(), oix.data(), oadj, &osimp) == 0);
        for (int k = 0; k < 6; ++k) {
          if (!oadj[k]) continue;
          auto it = index.find({cid[0] + dd[k][0], cid[1] + dd[k][1], cid[2] + dd[k][2]});
          if (it != index.end()) edges.insert({std::min(i, it->second), std::max(i, it->second)});
        }
      }
      CHECK(edges.size() > 100 && texmap.chunkGraph.num_edges() == edges.size());
      size_t degree = 0;
      for (size_t i = 0; i < n_patches; ++i) degree += texmap.chunkGraph.get_adj_nodes(i).size();
      CHECK(degree == 2 * edges.size());
      for (const auto& e : edges) CHECK(texmap.chunkGraph.has_edge(e.first, e.second) && texmap.chunkGraph.has_edge(e.second, e.first));
      // data costs: the keyframe just fused has no observations (row 1 stays empty), keyframe kfIndex is refreshed
      std::vector<int> lookup(kfIndex + 2, -1), framesToUpdate(1, kfIndex);
      lookup[kfIndex] = 0;
      lookup[kfIndex + 1] = 1;
      texmap.dataCost.set_value(0, 0, 123.0f);  // a stale entry of node 0: overwritten or removed below
      texmap.update_datacost(chunksToUpdate, cm, lookup, kfIndex + 1, framesToUpdate);
      size_t entries = 0;
      for (size_t i = 0; i < n_patches; ++i) {
        int cid[3] = {chunksToUpdate[i](0), chunksToUpdate[i](1), chunksToUpdate[i](2)};
        int32_t kf[8];
        float qq[8];
        const int64_t no = tfo_volume_get_observations(ov, cid, kf, qq, 8);
        std::map<std::size_t, float> want;
        float stat = 1.0f;
        for (int64_t k = 0; k < no; ++k) {
          if (kf[k] != kfIndex && kf[k] != kfIndex + 1) continue;
          if (qq[k] > stat) stat = qq[k];
          if (qq[k] > 0.0f) want[(size_t)lookup[kf[k]]] = qq[k];
        }
        if (i == 0 && !want.count(0)) {
          bool seen = false;
          for (int64_t k = 0; k < no; ++k) seen = seen || kf[k] == kfIndex;
          if (seen) want[0] = 123.0f;  // observed with quality 0: set_value is skipped, the stale entry survives
        }
        CHECK(texmap.dataCost.col(i) == want);
        CHECK(texmap.statistic[i] == stat);
        entries += want.size();
      }
      CHECK(entries > 50);
      // ... and the same bookkeeping fed by the DEVICE's exports (tf_export_adjacency / tf_export_datacost over the
      // observation table tf_observations_record keeps in HBM): identical graph, costs and statistics
      {
        chisel::TexMap dev;
        CHECK(dev.update_chunkgraph_device(chunksToUpdate, chiselMap.Handle()) == TF_OK);
        CHECK(dev.chunkGraph.num_nodes() == n_patches && dev.chunkGraph.num_edges() == edges.size());
        for (const auto& e : edges) CHECK(dev.chunkGraph.has_edge(e.first, e.second) && dev.chunkGraph.has_edge(e.second, e.first));
        dev.dataCost.set_value(0, 0, 123.0f);
        CHECK(dev.update_datacost_device(chunksToUpdate, chiselMap.Handle(), lookup, kfIndex + 1, framesToUpdate) == TF_OK);
        for (size_t i = 0; i < n_patches; ++i) {
          CHECK(dev.dataCost.col(i) == texmap.dataCost.col(i));
          CHECK(dev.statistic[i] == texmap.statistic[i]);
        }
      }
      texmap.check_graph(cm);  // every node still has its mesh: nothing is removed
      CHECK(texmap.chunkGraph.num_edges() == edges.size() && texmap.chunkGraph.get_adj_nodes(edges.begin()->first).size() > 0);
    }
    // keyframes + labels (TexMap / mapMAP are host code outside the path: a fixed assignment stands in)
    std::vector<unsigned char> rgb((size_t)W * H * 3), rgbB((size_t)W * H * 3);
    for (int i = 0; i < W * H; ++i)
      for (int c = 0; c < 3; ++c) { rgb[3 * i + c] = rgba[4 * i + c]; rgbB[3 * i + c] = (unsigned char)(rgba2[4 * i + c] * 3 / 4); }
    std::vector<chisel::Frame> frame_list(kfIndex + 2);
    const float Tinv[16] = {1, 0, 0, -0.01f, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // inverse of lastPose
    for (int f = kfIndex; f < kfIndex + 2; ++f) {
      frame_list[f].frame_index = f;
      frame_list[f].rgb = f == kfIndex ? rgb.data() : rgbB.data();
      frame_list[f].refined_depth = f == kfIndex ? depth.data() : depth2.data();
      std::memcpy(frame_list[f].pose_inv, Tinv, 64);
    }
    chisel::UniGraph labelset;
    std::vector<int32_t> kfidx(n_patches);
    for (size_t i = 0; i < n_patches; ++i) {
      labelset.chunks[chunksToUpdate[i]] = i;
      labelset.labels.push_back(kfIndex + (int)(i % 2));
      kfidx[i] = (int32_t)(i % 2);
    }
    CHECK(chiselMap.GeneratePatches(chunksToUpdate, labelset, frame_list, cameraModel) == 0);
    tfo_camera ocam = {W, H, 525.0f, 525.0f, 319.5f, 239.5f, 0.01f, 5.0f};
    tfo_volume_set_camera(ov, &ocam);
    tfo_atlas* oa = tfo_atlas_create(res, 0, cfg.atlas_h);
    tfo_keyframe okf[2];
    for (int f = 0; f < 2; ++f) {
      okf[f].rgb = frame_list[kfIndex + f].rgb; okf[f].depth = frame_list[kfIndex + f].refined_depth;
      okf[f].kf_id = kfIndex + f;
      std::memcpy(okf[f].T, Tinv, 64);
    }
    uint64_t ohot[2];
    CHECK(tfo_generate_patches(ov, oa, oids2.data(), (int64_t)n_patches, kfidx.data(), okf, ohot) == 0);
    CHECK(chiselMap.atlas.hot_start == ohot[0] && chiselMap.atlas.hot_end == ohot[1]);
    CHECK(chiselMap.atlas.loc_next == tfo_atlas_loc_next(oa));
    chiselMap.CompensateColor();
    CHECK(tfo_compensate_color_volume(ov) == 2);
    chiselMap.UpdateAtlas(chunksToUpdate);
    tfo_update_atlas(ov, oa, oids2.data(), (int64_t)n_patches);
    for (size_t i = 0; i < n_patches; ++i) {
      const chisel::ChunkID id = chunksToUpdate[i];
      CHECK(chiselMap.atlas.HasPatch(id));
      chisel::PatchPtr p = chiselMap.atlas.GetPatch(id);
      CHECK(p != nullptr);
      uint64_t otl = 0;
      int ofid = 0, oflags = 0;
      int32_t obb[4];
      float orat[2];
      int64_t opnv = 0;
      int cid[3] = {id(0), id(1), id(2)};
      std::vector<float> otc(2 * 2187), otcol(3 * 2187), olabs(3 * 2187);
      CHECK(tfo_volume_get_patch(ov, cid, &otl, &ofid, obb, &oflags, orat, &opnv, otc.data(), otcol.data(), olabs.data()) == 0);
      CHECK(p->texloc == otl && p->frameid == ofid && p->frameid == labelset.labels[i]);
      CHECK(p->boundingbox.x == obb[0] && p->boundingbox.y == obb[1] && p->boundingbox.width == obb[2] && p->boundingbox.height == obb[3]);
      CHECK(p->wrong_mapping == ((oflags & 4) != 0) && p->has_image == ((oflags & 8) != 0) && p->has_adjusted == ((oflags & 16) != 0));
      CHECK(p->ratio(0) == orat[0] && p->ratio(1) == orat[1]);
      CHECK(p->complete());
      if (i % 23 == 0) {
        chiselMap.FetchPatchData(id);
        CHECK((int64_t)p->texcoord.size() == opnv);
        CHECK(std::memcmp(p->texcoord.data(), otc.data(), (size_t)opnv * 8) == 0);
        CHECK(std::memcmp(p->texcolor.data(), otcol.data(), (size_t)opnv * 12) == 0);
        if ((oflags & 16) && (oflags & 32))
          for (int64_t k = 0; k < 3 * opnv; ++k) {
            const float dlt = (&p->labs[0].v[0])[k] - olabs[(size_t)k];
            CHECK(dlt < 2e-5f && dlt > -2e-5f);  // colour compensation: the one stated tolerance
          }
      }
    }
    std::vector<unsigned char> rows((size_t)cfg.atlas_h * 13824 * 3);
    chiselMap.atlas.DownloadRows(0, cfg.atlas_h, rows.data());
    CHECK(std::memcmp(rows.data(), tfo_atlas_buffer(oa), rows.size()) == 0);
    {  // Atlas::texture_buffer: the hot range the GUI thread uploads (MobileFusion.h:404-421) holds the oracle's texels
      const chisel::Atlas& at = chiselMap.atlas;
      CHECK(at.hot_end > at.hot_start && !at.texture_buffer.empty() && at.texture_buffer.cols == 13824);
      CHECK(std::memcmp(&at.texture_buffer.data[at.hot_start * 3], tfo_atlas_buffer(oa) + at.hot_start * 3,
                        (at.hot_end - at.hot_start) * 3) == 0);
    }
    // DrawMeshes: counts, indices and every vertex column but the packed colour delta are identical
    int64_t oni = 0;
    const int64_t onv = tfo_draw_meshes(ov, oa, NULL, NULL, 0, 0, &oni);
    std::vector<float> gv(12 * (size_t)onv + 12), ovx(gv.size());
    std::vector<unsigned int> gi((size_t)oni + 1);
    std::vector<uint32_t> oi((size_t)oni + 1);
    unsigned int ni = 0, nvx = 0;
    chiselMap.DrawMeshes(gv.data(), gi.data(), ni, nvx, onv, oni);
    CHECK(tfo_draw_meshes(ov, oa, ovx.data(), oi.data(), onv, oni, &oni) == onv);
    CHECK((int64_t)nvx == onv && (int64_t)ni == oni && onv > 1000);
    CHECK(std::memcmp(gi.data(), oi.data(), (size_t)oni * 4) == 0);
    for (int64_t k = 0; k < onv; ++k)
      for (int c = 0; c < 12; ++c)
        if (c != 5) CHECK(std::memcmp(&gv[12 * (size_t)k + c], &ovx[12 * (size_t)k + c], 4) == 0);
    tfo_atlas_destroy(oa);

    // ---- main.cpp:262-270: SaveAllMeshesToPLY + atlas.SaveTexturedModel -- the files, read back and compared
    // with the mirrors they were written from (which the blocks above compared with the oracle)
    {
      const char* tmp = std::getenv("TF_TEST_TMP");
      const std::string base = tmp ? tmp : "/tmp";
      const std::string ply = base + "/OnlineModel_5mm.ply";
      CHECK(chiselMap.SaveAllMeshesToPLY(ply));
      size_t tri_verts = 0;
      for (const auto& it : chiselMap.GetChunkManager().GetAllMeshes()) tri_verts += (size_t)it.second->n_indices;
      std::ifstream in(ply.c_str());
      std::string line;
      size_t header_v = 0, header_f = 0, body = 0;
      bool in_body = false;
      while (std::getline(in, line)) {
        if (in_body) { ++body; continue; }
        if (line.compare(0, 15, "element vertex ") == 0) header_v = (size_t)std::atol(line.c_str() + 15);
        if (line.compare(0, 13, "element face ") == 0) header_f = (size_t)std::atol(line.c_str() + 13);
        if (line == "end_header") in_body = true;
      }
      CHECK(header_v == tri_verts && header_f == tri_verts / 3 && body == header_v + header_f && tri_verts > 3000);
      gcFusion.chiselMap->atlas.SaveTexturedModel(base);
      // texture_material.png: signature, IHDR of the whole texture_buffer, size of a stored-deflate stream
      std::ifstream png((base + "/texture_material.png").c_str(), std::ios::binary);
      unsigned char hd[24];
      png.read((char*)hd, 24);
      CHECK(png.gcount() == 24 && hd[1] == 'P' && hd[2] == 'N' && hd[3] == 'G');
      const uint32_t pw = (uint32_t)hd[16] << 24 | hd[17] << 16 | hd[18] << 8 | hd[19];
      const uint32_t ph = (uint32_t)hd[20] << 24 | hd[21] << 16 | hd[22] << 8 | hd[23];
      CHECK(pw == 13824 && ph == (uint32_t)cfg.atlas_h);
      // texture_model.obj: one v / vt / vn per vertex of every complete patch, one f per triangle; vt of the first
      // vertex of the first mesh = (texloc + texcoord * ratio) / atlas size, v flipped (Atlas.cpp:124-130,143)
      size_t want_v = 0, want_f = 0;
      for (const auto& it : chiselMap.GetChunkManager().GetAllMeshes())
        if (it.second->m_patch && it.second->m_patch->complete()) { want_v += (size_t)it.second->n_vertices; want_f += (size_t)it.second->n_indices / 3; }
      std::ifstream obj((base + "/texture_model.obj").c_str());
      size_t nv_ = 0, nvt = 0, nvn = 0, nf = 0;
      std::string first_vt;
      while (std::getline(obj, line)) {
        if (line.compare(0, 2, "v ") == 0) ++nv_;
        else if (line.compare(0, 3, "vt ") == 0) { if (!nvt) first_vt = line; ++nvt; }
        else if (line.compare(0, 3, "vn ") == 0) ++nvn;
        else if (line.compare(0, 2, "f ") == 0) ++nf;
      }
      CHECK(want_v > 1000 && nv_ == want_v && nvt == want_v && nvn == want_v && nf == want_f);
      {  // (the file lists the meshes in ascending chunk id, like DrawMeshes' stream)
        const chisel::Mesh* first_mesh = nullptr;
        for (const auto& it : chiselMap.GetChunkManager().GetAllMeshes()) {
          if (!(it.second->m_patch && it.second->m_patch->complete()) || it.second->n_vertices == 0) continue;
          const chisel::ChunkID& a = it.first;
          if (!first_mesh) { first_mesh = it.second.get(); continue; }
          const chisel::ChunkID& b = first_mesh->chunkID;
          if (a(0) < b(0) || (a(0) == b(0) && (a(1) < b(1) || (a(1) == b(1) && a(2) < b(2))))) first_mesh = it.second.get();
        }
        CHECK(first_mesh != nullptr);
        chiselMap.FetchPatchData(first_mesh->chunkID);
        const chisel::Patch& p = *first_mesh->m_patch;
        chisel::Vec2 tex((float)(p.texloc % 13824), (float)(p.texloc / 13824));
        tex(0) += p.texcoord[0](0) * p.ratio(0);
        tex(1) += p.texcoord[0](1) * p.ratio(1);
        tex(0) /= 13824;
        tex(1) /= (float)cfg.atlas_h;
        std::ostringstream want;
        want << std::fixed << std::setprecision(6) << "vt " << tex(0) << " " << 1.0f - tex(1);
        CHECK(first_vt == want.str());
      }
      std::ifstream mtl((base + "/texture_model.mtl").c_str());
      std::getline(mtl, line);
      CHECK(line == "newmtl demo_texture");
      // main.cpp:267: Reset() empties the map; the atlas keeps its texels (Chisel.cpp:47-50)
      gcFusion.chiselMap->Reset();
      CHECK(chiselMap.GetChunkManager().GetAllMeshes().empty() && chiselMap.GetChunkManager().GetChunkIDs().empty());
    }
  }

  tfo_volume_destroy(ov);
  std::printf("HOST MIRROR PARITY OK (%zu chunks in list, %lld valid, %zu observations, %zu patches)\n", n, (long long)nv, nobs, n_patches);
  return 0;
}
