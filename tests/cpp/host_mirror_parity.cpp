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

  // ---- atlas stage as MobileFusion.cpp:347-384 drives it: GeneratePatches + UpdateAtlas,
  // CompensateColor, DrawMeshes on per-chunk meshes (synthetic vertex clouds on the wall)
  size_t n_patches = 0;
  {
    std::vector<unsigned char> rgb((size_t)W * H * 3);
    for (int i = 0; i < W * H; ++i) { rgb[3 * i] = rgba[4 * i]; rgb[3 * i + 1] = rgba[4 * i + 1]; rgb[3 * i + 2] = rgba[4 * i + 2]; }
    chiselMap.CacheKeyframe(kfIndex, rgb.data(), depth.data());
    chiselMap.CacheKeyframe(kfIndex + 1, rgb.data(), depth2.data());
    std::vector<chisel::PatchMesh> meshes;
    std::vector<int> labels;
    std::vector<const float*> poseInv;
    float Tinv[16] = {1, 0, 0, -0.01f, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // inverse of lastPose
    const float edge = 8 * res;
    for (int64_t i = 0; i < nv && meshes.size() < 150; i += 7) {
      const chisel::ChunkID id = validChunks[i];
      if (!(id(2) * edge <= 1.5f && 1.5f < (id(2) + 1) * edge)) continue;
      chisel::PatchMesh m;
      m.chunkID = id;
      unsigned h = (unsigned)(id(0) * 73856093) ^ (unsigned)(id(1) * 19349663);
      const int nvert = 6 + (int)(h % 20u);
      for (int k = 0; k < nvert; ++k) {
        h = h * 1664525u + 1013904223u;
        const float fx = (float)((h >> 8) & 0xFFFF) / 65536.0f, fy = (float)((h >> 12) & 0xFFFF) / 65536.0f;
        m.vertices.push_back((id(0) + fx) * edge); m.vertices.push_back((id(1) + fy) * edge); m.vertices.push_back(1.5f);
        m.colors.push_back((float)((h >> 3) & 0xFF) / 255.0f); m.colors.push_back((float)((h >> 11) & 0xFF) / 255.0f);
        m.colors.push_back((float)((h >> 19) & 0xFF) / 255.0f);
        m.normals.push_back(0.0f); m.normals.push_back(0.0f); m.normals.push_back(-1.0f);
      }
      for (int k = 0; k + 2 < nvert; ++k) { m.indices.push_back(0); m.indices.push_back(k + 1); m.indices.push_back(k + 2); }
      meshes.push_back(m);
      labels.push_back(kfIndex + (int)(meshes.size() % 2));
      poseInv.push_back(Tinv);
    }
    n_patches = meshes.size();
    CHECK(n_patches > 40);
    std::vector<chisel::PatchResult> patches;
    CHECK(chiselMap.GeneratePatchesAndUpdateAtlas(meshes, labels, poseInv, cameraModel, patches) == 0);
    // oracle: slot, projection, blit per patch in the same order
    tfo_camera ocam = {W, H, 525.0f, 525.0f, 319.5f, 239.5f, 0.01f, 5.0f};
    tfo_atlas* oa = tfo_atlas_create(res, 0, cfg.atlas_h);
    std::vector<int32_t> fid(n_patches);
    std::vector<uint8_t> wrong(n_patches), adj(n_patches, 0), complete(n_patches, 1), labs_valid(n_patches);
    std::vector<int64_t> voff(n_patches + 1, 0), ioff(n_patches + 1, 0);
    std::vector<float> tex, meshc, verts, nrm, tcoord, ratio(2 * n_patches);
    std::vector<uint64_t> texlocs(n_patches);
    std::vector<uint32_t> idx;
    for (size_t p = 0; p < n_patches; ++p) {
      const size_t nvp = meshes[p].vertices.size() / 3;
      uint64_t tl = 0;
      CHECK(tfo_atlas_alloc(oa, &tl) == 0);
      CHECK(tl == patches[p].texloc);
      std::vector<float> tc(2 * nvp), tcol(3 * nvp);
      int32_t bbox[4];
      int wm = 0;
      int64_t caution = 0;
      const float* dimg = labels[p] == kfIndex ? depth.data() : depth2.data();
      const int flag = tfo_patch_project(meshes[p].vertices.data(), meshes[p].colors.data(), (int64_t)nvp, Tinv, rgb.data(),
                                         dimg, &ocam, tc.data(), tcol.data(), bbox, &wm, &caution);
      CHECK(flag == patches[p].flag && (wm != 0) == patches[p].wrong_mapping);
      CHECK(std::memcmp(bbox, patches[p].boundingbox, 16) == 0);
      CHECK(std::memcmp(tc.data(), patches[p].texcoord.data(), tc.size() * 4) == 0);
      CHECK(std::memcmp(tcol.data(), patches[p].texcolor.data(), tcol.size() * 4) == 0);
      float r2[2] = {1.0f, 1.0f};
      CHECK(tfo_atlas_blit(oa, tl, rgb.data(), W, H, bbox, r2) == 0);
      CHECK(r2[0] == patches[p].ratio[0] && r2[1] == patches[p].ratio[1]);
      fid[p] = labels[p]; wrong[p] = wm ? 1 : 0; texlocs[p] = tl;
      ratio[2 * p] = r2[0]; ratio[2 * p + 1] = r2[1];
      tex.insert(tex.end(), tcol.begin(), tcol.end());
      tcoord.insert(tcoord.end(), tc.begin(), tc.end());
      meshc.insert(meshc.end(), meshes[p].colors.begin(), meshes[p].colors.end());
      verts.insert(verts.end(), meshes[p].vertices.begin(), meshes[p].vertices.end());
      nrm.insert(nrm.end(), meshes[p].normals.begin(), meshes[p].normals.end());
      idx.insert(idx.end(), meshes[p].indices.begin(), meshes[p].indices.end());
      voff[p + 1] = (int64_t)verts.size() / 3;
      ioff[p + 1] = (int64_t)idx.size();
    }
    std::vector<unsigned char> rows((size_t)cfg.atlas_h * 13824 * 3);
    chiselMap.atlas.DownloadRows(0, cfg.atlas_h, rows.data());
    CHECK(std::memcmp(rows.data(), tfo_atlas_buffer(oa), rows.size()) == 0);
    // CompensateColor (Chisel.cpp:198-286): flags exact, colours within the stated tolerance
    chiselMap.CompensateColor(meshes, patches);
    std::vector<float> olabs(tex.size(), 0.0f);
    tfo_color_compensate((int64_t)n_patches, fid.data(), wrong.data(), adj.data(), voff.data(), tex.data(), meshc.data(),
                         olabs.data(), NULL, NULL);
    std::vector<float> mlabs(tex.size(), 0.0f);
    for (size_t p = 0; p < n_patches; ++p) {
      CHECK(patches[p].has_adjusted == (adj[p] != 0));
      const bool lv = patches[p].has_adjusted && !patches[p].labs.empty();
      CHECK(lv == (adj[p] != 0 && !wrong[p]));
      labs_valid[p] = lv ? 1 : 0;
      if (!lv) continue;
      for (size_t k = 0; k < patches[p].labs.size(); ++k) {
        const float d = patches[p].labs[k] - olabs[3 * voff[p] + k];
        CHECK(d < 2e-5f && d > -2e-5f);
        mlabs[3 * voff[p] + k] = patches[p].labs[k];
      }
    }
    // DrawMeshes (Chisel.cpp:288-355): bit-exact given the same compensated colours
    std::vector<float> gv(12 * (size_t)voff[n_patches] + 12), ovx(gv.size());
    std::vector<unsigned int> gi(idx.size() + 1);
    std::vector<uint32_t> oi(idx.size() + 1);
    unsigned int ni = 0, nvx = 0;
    chiselMap.DrawMeshes(meshes, patches, gv.data(), gi.data(), ni, nvx);
    int64_t oni = 0;
    const int64_t onv = tfo_pack_vertices((int64_t)n_patches, complete.data(), wrong.data(), labs_valid.data(), texlocs.data(),
                                          ratio.data(), 13824, cfg.atlas_h, voff.data(), verts.data(), meshc.data(), nrm.data(),
                                          tcoord.data(), tex.data(), mlabs.data(), ioff.data(), idx.data(), ovx.data(),
                                          oi.data(), &oni);
    CHECK((int64_t)nvx == onv && (int64_t)ni == oni && onv == voff[n_patches]);
    CHECK(std::memcmp(gv.data(), ovx.data(), (size_t)onv * 48) == 0);
    CHECK(std::memcmp(gi.data(), oi.data(), (size_t)oni * 4) == 0);
    tfo_atlas_destroy(oa);
  }

  tfo_volume_destroy(ov);
  std::printf("HOST MIRROR PARITY OK (%zu chunks in list, %lld valid, %zu observations, %zu patches)\n", n, (long long)nv, nobs, n_patches);
  return 0;
}
