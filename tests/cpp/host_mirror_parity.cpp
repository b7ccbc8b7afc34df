// Drives the host-side C++ mirror (texturefusion_amd/host/tf_chisel.hpp) the way
// GCFusion/MobileFusion.cpp drives the reference classes (ReIntegrateKeyframe :114-221,
// IntegrateFrame :223-250) and checks every result against the CPU oracle (test infrastructure).
// Plain C++14, no Eigen/OpenCV; built with g++ against libtexfusion_hip.so and libtf_oracle.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../oracle/tf_oracle.h"
#include "../../texturefusion_amd/host/tf_chisel.hpp"

#define CHECK(c)                                                                   \
  do {                                                                             \
    if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } \
  } while (0)

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
  const int chunkSize[3] = {8, 8, 8};
  tf_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.max_chunks = 1 << 15;
  cfg.atlas_h = 36;

  // MobileFusion::initChiselMap (GCFusion/MobileFusion.h:205-258)
  chisel::Chisel chiselMap(chunkSize, res, true, &cfg);
  chisel::ProjectionIntegrator projectionIntegrator;
  projectionIntegrator.SetTruncator(chisel::TruncatorPtr(new chisel::QuadraticTruncator(0.0019f, 0.00152f, 0.001504f, 6.0f)));
  projectionIntegrator.SetWeighter(chisel::WeighterPtr(new chisel::ConstantWeighter(1)));
  projectionIntegrator.SetCarvingDist(0.05f);
  projectionIntegrator.SetCarvingEnabled(true);
  chisel::PinholeCamera cameraModel;
  cameraModel.SetIntrinsics(525.0f, 525.0f, 319.5f, 239.5f);
  cameraModel.SetNearPlane(0.01f);
  cameraModel.SetFarPlane(5.0f);
  cameraModel.SetWidth(W);
  cameraModel.SetHeight(H);

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

  tfo_volume_destroy(ov);
  std::printf("HOST MIRROR PARITY OK (%zu chunks in list, %lld valid, %zu observations)\n", n, (long long)nv, nobs);
  return 0;
}
