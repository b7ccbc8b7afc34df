// tf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the TSDF fusion path.
//
// Compiled with -ffp-contract=off: every float operation is rounded separately, in the order
// the reference's AVX2 build performs it (no -mfma there, CMakeLists.txt:57-58).  f32 division
// is IEEE correctly rounded (hipcc default), f32 denormals are kept (gfx9 default), conversions
// reproduce _mm256_cvtps_epi32 (round-to-nearest-even, 0x80000000 for NaN / out of range).
//
// Kernel map (reference file:line each one replaces):
//   k_bbox      ChunkManager::findCubeCornerByMat / GetBoundaryChunkID  Structure/ChunkManager.h:303-378
//   k_select    ChunkManager::GetChunkIDsObservedByCamera + CheckCornerIntersectingSIMD  :380-636
//   k_scan      order-preserving compaction + write-out of the visible list (push_back order :544)
//   k_acquire   Chisel::PrepareIntersectChunks' HasChunk / CreateChunk  Structure/Chisel.h:130-138
//   k_integrate ProjectionIntegrator::voxelUpdateSIMD  3rd_party/open_chisel/utils/ProjectionIntegrator.cpp:67-426
//               + the per-chunk lambda of Chisel::IntegrateDepthScanColor  Structure/Chisel.h:234-248
//   k_finalize  Chisel::FinalizeIntegrateChunks + GarbageCollect  Structure/Chisel.h:184-216,472-477
//   k_frame     the 5-argument per-frame unit (Structure/Chisel.h:453-468) as ONE launch: K-A of frame f,
//               K-C (k_select's body) of frame f+1, K-B (k_bbox's body) of frame f+2 as block ranges
#include <stdlib.h>

#include <atomic>
#include "tf_device.h"
#include "tf_devfn.h"
#include "tf_host_math.h"
#include "tf_patch_body.h"
#include "tf_voxel_math.h"

#pragma clang fp contract(off)
#ifndef TF_KA_GP
#define TF_KA_GP 2  // z-slices per read-modify-write pass of K-A
#endif


namespace tf {

// ---------------------------------------------------------------------------------------
// control block reset (create / Reset only; per-frame re-arming rides on k_scan)
// ---------------------------------------------------------------------------------------
__global__ void k_reset_ctl(FrameCtl* ctl, VolCtl* vctl) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) {
      ctl->bbox_key[a] = f2key(1e8f);
      ctl->bbox_key[3 + a] = f2key(-1e8f);
      ctl->min_id[a] = ctl->max_id[a] = ctl->dims[a] = 0;
    }
    ctl->n_coarse = 0;
    ctl->n_list = 0;
    ctl->n_front = 0;
    ctl->emit_pack = 0ull;
    for (int k = 0; k < 4; ++k) ctl->band_cnt[k] = 0u;
    if (vctl) {
      vctl->status = 0; vctl->n_tmp = 0; vctl->n_tmp2 = 0; vctl->ovf_next = 0; vctl->xchg_sent = 0; vctl->xchg_recv = 0;
      vctl->zero_word = 0; vctl->blk_next = 0; vctl->xchg_cnt[0] = vctl->xchg_cnt[1] = 0; vctl->xchg_ticket = 0; vctl->create_seq = 0; vctl->xl_n[0] = vctl->xl_n[1] = 0;
      vctl->blk_head[0] = vctl->blk_head[1] = vctl->blk_tail[0] = vctl->blk_tail[1] = 0; vctl->blk_recycle = 0;
      for (int k = 0; k < kSlotStripes; ++k) vctl->slot_cnt[k] = 0;
    }
  }
}
void launch_reset_ctl(const VolumeDev& v, bool volume_too, hipStream_t s) {
  hipLaunchKernelGGL(k_reset_ctl, dim3(1), dim3(64), 0, s, v.sel.ctl, volume_too ? v.vctl : nullptr);
}

// an empty launch (tf_profile_calibrate: what a HIP-event pair around ANY launch reads at least)
__global__ void k_null() {}
void launch_null(hipStream_t s) { hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s); }

// fresh chunk state: sdf 999, weight 0 (Chunk.cpp:64-65), colour 0 (ColorVoxel.cpp:26-33)
__global__ __launch_bounds__(256) void k_fill_pool(float2* tsdf, ushort4* color, uint32_t* summ, size_t first,
                                                   size_t count) {
  const float4 f = make_float4(999.0f, 0.0f, 999.0f, 0.0f);
  const uint4 z = make_uint4(0, 0, 0, 0);
  float4* t4 = reinterpret_cast<float4*>(tsdf + first);
  uint4* c4 = reinterpret_cast<uint4*>(color + first);
  const size_t n4 = count / 2;  // two voxels per 16 B in either plane
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    t4[i] = f;
    c4[i] = z;
    if (i < count / kChunkVoxels) summ[first / kChunkVoxels + i] = 0u;  // nothing observed (VolumeDev::summ)
  }
}
void launch_fill_pool(const VolumeDev& v, uint32_t slot0, uint32_t nslots, hipStream_t s) {
  if (!nslots) return;
  size_t first = (size_t)slot0 * kChunkVoxels, count = (size_t)nslots * kChunkVoxels;
  size_t blocks = (count / 2 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_fill_pool, dim3((unsigned)blocks), dim3(256), 0, s, v.tsdf, v.color, v.summ, first, count);
}

// ---------------------------------------------------------------------------------------
// RGBA staging of the caller (GCFusion/MobileFusion.cpp:144-163): rgba = valid ? (r, g, b, 1) : 0
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_rgba(const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ valid,
                                                   uchar4* __restrict__ rgba, uint32_t npix) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const bool ok = !valid || valid[i] > 0;  // (no flags: MobileFusion::IntegrateFrame's loop, alpha = 1 everywhere)
    rgba[i] = ok ? make_uchar4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 1) : make_uchar4(0, 0, 0, 0);
  }
}
void launch_pack_rgba(const uint8_t* rgb, const uint8_t* valid, uchar4* rgba, uint32_t npix, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_rgba, dim3(1024), dim3(256), 0, s, rgb, valid, rgba, npix);
}

// ---------------------------------------------------------------------------------------
// K-B  world AABB of the back-projected (depth + 0.2) points
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void bbox_body(const float* __restrict__ depth, const Cam& cam, const Pose& P,
                                          FrameCtl* ctl, const uint32_t bid, const uint32_t nb) {
  if (bid == 0 && threadIdx.x == 0) { ctl->n_list = 0; ctl->n_front = 0; ctl->emit_pack = 0ull; }  // appended to by k_select<EMIT>
  if (bid == 0 && threadIdx.x < 4) ctl->band_cnt[threadIdx.x] = 0u;
  float mn[3] = {1e8f, 1e8f, 1e8f}, mx[3] = {-1e8f, -1e8f, -1e8f};
  const int W = cam.W;
  const int nvec = (cam.W * cam.H) >> 2;
  const float off = 0.2f;
  for (int q = (int)bid * 256 + threadIdx.x; q < nvec; q += (int)nb * 256) {
    const float4 d4 = reinterpret_cast<const float4*>(depth)[q];
    const int pix = q << 2;
    const int i = pix / W, j = pix - i * W;
    const float ly = ((float)i - cam.cyi) / cam.fyi;
    const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dz = dd[k] + off;
      const float lx = ((float)(j + k) - cam.cxi) / cam.fxi;
      const float vx = lx * dz, vy = ly * dz;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float p = P.p[4 * a] * vx;
        p = p + P.p[4 * a + 1] * vy;
        p = p + P.p[4 * a + 2] * dz;
        p = p + P.p[4 * a + 3];
        mx[a] = (p > mx[a]) ? p : mx[a];
        mn[a] = (p < mn[a]) ? p : mn[a];
      }
    }
  }
  // min/max are exact and order-independent: reduce across the wave, then the block.
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      float t = __shfl_xor(mn[a], o);
      mn[a] = (t < mn[a]) ? t : mn[a];
      t = __shfl_xor(mx[a], o);
      mx[a] = (t > mx[a]) ? t : mx[a];
    }
  }
  __shared__ float red[4][6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) { red[w][a] = mn[a]; red[w][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float r = red[0][threadIdx.x];
    for (int k = 1; k < 4; ++k) {
      float t = red[k][threadIdx.x];
      r = (threadIdx.x < 3) ? ((t < r) ? t : r) : ((t > r) ? t : r);
    }
    if (threadIdx.x < 3) atomicMin(&ctl->bbox_key[threadIdx.x], f2key(r));
    else atomicMax(&ctl->bbox_key[threadIdx.x], f2key(r));
  }
}
__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ depth, Cam cam, Pose P,
                                              FrameCtl* ctl) {
  bbox_body(depth, cam, P, ctl, blockIdx.x, gridDim.x);
}
void launch_bbox(const VolumeDev& v, const float* depth, const Cam& cam, const Pose& pose,
                 hipStream_t s) {
  int nvec = (cam.W * cam.H) >> 2;
  int blocks = (nvec + 255) / 256;
  if (blocks > 128) blocks = 128;  // 6 atomics per block on 6 words: keep the tail short
  hipLaunchKernelGGL(k_bbox, dim3(blocks), dim3(256), 0, s, depth, cam, pose, v.sel.ctl);
}

// ---------------------------------------------------------------------------------------
// K-C  coarse 4x4x4-block test, then per-chunk test; one 64-bit mask per coarse block
// ---------------------------------------------------------------------------------------
struct ProbeRes { bool valid; bool hit; float sd; };

// One lane = one of the 8 probe points of CheckCornerIntersectingSIMD (ChunkManager.h:561-636).
__device__ __forceinline__ ProbeRes probe(const float* __restrict__ depth, const Cam& cam,
                                          float ocx, float ocy, float ocz, float offx, float offy,
                                          float offz, float dtp, float ndtn) {
  const float px = ocx + offx, py = ocy + offy, pz = ocz + offz;
  const float u = (px / pz) * cam.fxi + cam.cxi;  // no +0.5 here (:584-593)
  const float w = (py / pz) * cam.fyi + cam.cyi;
  const int X = cvt_rne(u), Y = cvt_rne(w);
  ProbeRes r;
  r.valid = (X > 1) && (cam.W - 1 > X) && (Y > 1) && (cam.H - 1 > Y);
  float d = 0.0f;
  if (r.valid) d = depth[Y * cam.W + X];
  const float sd = d - pz;
  r.hit = r.valid && (sd > ndtn) && (dtp > sd);
  r.sd = sd;
  return r;
}

// EMIT = the fused per-frame unit: the list is consumed on the device only and its order is
// irrelevant there (chunks are independent), so each hit block appends its chunk ids straight to
// the list with one atomic and the ordered scan/write-out kernel is skipped.  The call-by-call
// flow (tf_prepare) needs the reference order and uses EMIT = false + k_scan.
template <bool EMIT>
__device__ __forceinline__ void select_body(const float* __restrict__ depth, const Cam& cam, const Integ& ig,
                                            const SelectConsts& sc, const VolumeDev& v, const uint32_t bid,
                                            const uint32_t nb) {
  FrameCtl* ctl = v.sel.ctl;
  // GetIDAt (ChunkManager.h:197-207) on the reduced corners; every block derives the same grid.
  int minI[3], maxI[3], dims[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    minI[a] = (int)floorf(key2f(ctl->bbox_key[a]) * sc.id_factor);
    maxI[a] = (int)floorf(key2f(ctl->bbox_key[3 + a]) * sc.id_factor);
    dims[a] = (maxI[a] + 1 - (minI[a] - 1)) / sc.step + 1;
    if (dims[a] < 0) dims[a] = 0;
  }
  const unsigned long long total = (unsigned long long)dims[0] * dims[1] * dims[2];
  const bool overflow = total > v.max_coarse;
  const uint32_t n_coarse = overflow ? 0u : (uint32_t)total;
  if (bid == 0 && threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) { ctl->min_id[a] = minI[a]; ctl->max_id[a] = maxI[a]; ctl->dims[a] = dims[a]; }
    ctl->n_coarse = n_coarse;
    if (overflow) atomicOr(&v.vctl->status, kStCoarseFull);
  }
  if (EMIT && bid == 0) centroid_table(sc.pose, sc.res, v.sel.cen);  // for K-A of this frame, one launch later
  const int lane = threadIdx.x & 63;
  // the wave id IS wave-uniform, but anything derived from threadIdx is divergent to the compiler;
  // readfirstlane makes the uniformity provable (scalar loads, no waterfall loops around buffer ops)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((bid * 256 + threadIdx.x) >> 6));
  const uint32_t nwaves = nb * 4;
  const int corner = lane & 7;
  const int step = sc.step;
  const uint32_t nzny = (uint32_t)dims[2] * (uint32_t)dims[1];
  const bool partitioned = EMIT && (v.part_lo != INT32_MIN || v.part_hi != INT32_MAX);
  const long long band_w = partitioned ? (long long)(v.part_a + v.part_b + v.part_c) : -1ll;

  // One wave per coarse block: the 8 corner probes run on lanes 0..7 (replicated 8x), the 64
  // per-chunk tests of a hit block on the 64 lanes (lane = (i-x)*16 + (j-y)*4 + (k-z), i.e. the
  // reference's i,j,k push_back order).  Every block is an independent short dependency chain.
  // Multi-GPU, fused flow: the list only has to hold this rank's slab, so coarse blocks whose key
  // range misses [part_lo, part_hi) are skipped before any probe and chunks outside the slab are
  // dropped from hit blocks.  The call-by-call flow keeps the full, reference-ordered list (every
  // rank returns the same list; K-A skips what it does not own).
  for (uint32_t cb = wave; cb < n_coarse; cb += nwaves) {
    const uint32_t bx = cb / nzny;
    const uint32_t brem = cb - bx * nzny;
    const uint32_t by = brem / (uint32_t)dims[2];
    const uint32_t bz = brem - by * (uint32_t)dims[2];
    const int x0 = minI[0] - 1 + (int)bx * step;
    const int y0 = minI[1] - 1 + (int)by * step;
    const int z0 = minI[2] - 1 + (int)bz * step;
    if (EMIT) {  // coefficients are >= 0: the block's smallest / largest key sit at opposite corners
      const int kmin = part_key(v, x0, y0, z0), kmax = part_key(v, x0 + step - 1, y0 + step - 1, z0 + step - 1);
      // (a partitioned volume also looks at the neighbours' ghost bands -- key lo - 1 below, hi .. hi + a + b + c above --
      // to count what they will send: FrameCtl::band_cnt)
      if ((long long)kmax < (long long)v.part_lo - 1ll || (long long)kmin > (long long)v.part_hi + band_w) continue;
    }
    float oc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // :473-479
      float ox = sc.r0[a] * (float)x0 - sc.tc[a];
      float oy = ox + sc.r1[a] * (float)y0;
      oc[a] = oy + (float)z0 * sc.r2[a];
    }
    const float trunc = truncation(ig, oc[2]);
    const float dtp = trunc + sc.diag_step;  // :489
    const float ndtn = -sc.dtn_coarse;       // :490, :623
    const bool depthValid = (oc[2] > cam.nearP) && (cam.farP > oc[2]);  // :598-602
    ProbeRes pr = probe(depth, cam, oc[0], oc[1], oc[2], sc.coarse[0][corner], sc.coarse[1][corner],
                        sc.coarse[2][corner], dtp, ndtn);
    const bool coarse_hit = __ballot(pr.hit && depthValid) != 0ull;
    unsigned long long m = 0ull;
    bool costly = false;
    if (coarse_hit) {
      bool flag = false;
      uint32_t band = 0u;
      if (lane < step * step * step) {
        const int di = (step == 4) ? (lane >> 4) : 0;
        const int dj = (step == 4) ? ((lane >> 2) & 3) : 0;
        const int dk = (step == 4) ? (lane & 3) : 0;
        const float org[3] = {(float)((x0 + di) * 8) * sc.res, (float)((y0 + dj) * 8) * sc.res,
                              (float)((z0 + dk) * 8) * sc.res};  // :521-523
        float of[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {  // rotation * origin - translation (:524)
          float sacc = sc.rot[a][0] * org[0];
          sacc = sacc + sc.rot[a][1] * org[1];
          sacc = sacc + sc.rot[a][2] * org[2];
          of[a] = sacc - sc.tc[a];
        }
        const float tr = truncation(ig, of[2]);
        const float fdtp = tr + sc.diag;   // :528
        const float fndtn = -sc.dtn_fine;  // :529
        const bool dv = (of[2] > cam.nearP) && (cam.farP > of[2]);
        bool anyhit = false;
        bool below = false, above = false;  // EMIT: some probed corner sits below the band's far edge / above its near edge
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          ProbeRes fr = probe(depth, cam, of[0], of[1], of[2], sc.fine[0][c], sc.fine[1][c],
                              sc.fine[2][c], fdtp, fndtn);
          anyhit |= fr.hit;
          if (EMIT) { below |= fr.valid && (tr > fr.sd); above |= fr.valid && (fr.sd > fndtn + sc.diag); }  // the band without the chunk-diagonal margin
        }
        flag = anyhit && dv;
        costly = (below && above) || sc.plain != 0;
        if (EMIT) {
          const long long k = part_key(v, x0 + di, y0 + dj, z0 + dk);
          const long long lo = v.part_lo, hi = v.part_hi;
          if (partitioned && flag)
            band = (k >= lo && k - lo <= band_w ? 1u : 0u) | (k == hi - 1 ? 2u : 0u) | (k == lo - 1 ? 4u : 0u) |
                   (k >= hi && k - hi <= band_w ? 8u : 0u);
          flag = flag && k >= lo && k < hi;
        }
      }
      m = __ballot(flag);
      if (partitioned) {  // (wave-uniform; the four counts of FrameCtl::band_cnt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t c = (uint32_t)__popcll(__ballot((band >> q) & 1u));
          if (c && lane == 0) atomicAdd(&ctl->band_cnt[q], c);
        }
      }
    }
    if (!EMIT) {
      if (lane == 0) v.sel.masks[cb] = m;
    } else if (m) {
      // two-ended append (FrameCtl::n_front): one 64-bit atomic hands out front and back positions together
      const unsigned long long mf = __ballot(((m >> lane) & 1ull) && costly);
      const uint32_t kf = (uint32_t)__popcll(mf), kb = (uint32_t)__popcll(m) - kf;
      unsigned long long pk = 0ull;
      if (lane == 0) pk = atomicAdd(&ctl->emit_pack, (unsigned long long)kf | ((unsigned long long)kb << 32));
      const uint32_t basef = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pk);
      const uint32_t baseb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pk >> 32));
      if ((unsigned long long)basef + kf + baseb + kb > (unsigned long long)v.max_list) {
        if (lane == 0) atomicOr(&v.vctl->status, kStListFull);
      } else if ((m >> lane) & 1ull) {
        int4 id;
        id.x = x0 + ((step == 4) ? (lane >> 4) : 0);
        id.y = y0 + ((step == 4) ? ((lane >> 2) & 3) : 0);
        id.z = z0 + ((step == 4) ? (lane & 3) : 0);
        id.w = 0;
        const unsigned long long lower = (1ull << lane) - 1ull;
        const uint32_t pos = costly ? basef + (uint32_t)__popcll(mf & lower)
                                    : v.max_list - 1u - (baseb + (uint32_t)__popcll(m & ~mf & lower));
        v.sel.list_id[pos] = id;
        const ChunkPre cp = chunk_pre(id, sc.pose, ig, sc.res, sc.resDiag);
        v.sel.list_pre[4 * pos] = make_float4(cp.a.x, cp.a.y, cp.a.z, cp.b.x);
        v.sel.list_pre[4 * pos + 1] = make_float4(cp.b.y, __int_as_float(id.x), __int_as_float(id.y), __int_as_float(id.z));
      }
    }
  }
}
template <bool EMIT>
__global__ __launch_bounds__(256) void k_select(const float* __restrict__ depth, Cam cam, Integ ig,
                                                SelectConsts sc, VolumeDev v) {
  select_body<EMIT>(depth, cam, ig, sc, v, blockIdx.x, gridDim.x);
}
void launch_select(const VolumeDev& v, const float* depth, const Cam& cam, const Integ& ig,
                   const Pose& pose, float res, bool emit, hipStream_t s, bool plain) {
  SelectConsts sc = make_select_consts(pose.p, res);
  sc.plain = plain ? 1 : 0;
  if (emit) hipLaunchKernelGGL(k_select<true>, dim3(1024), dim3(256), 0, s, depth, cam, ig, sc, v);
  else hipLaunchKernelGGL(k_select<false>, dim3(1024), dim3(256), 0, s, depth, cam, ig, sc, v);
}

// ---------------------------------------------------------------------------------------
// exclusive scan of popcount(mask) over the coarse blocks (x-outer .. z-inner order, then lane
// order i,j,k inside a block = the reference's push_back order) and list write-out.  Also re-arms
// the bbox keys for the next frame (they were consumed by k_select).
// Round 6: a single-pass scan with decoupled look-back over tiles of 256 coarse blocks (one
// workgroup of four waves per tile, 256 resident workgroups taking tiles in ascending order)
// instead of ONE workgroup that walked all of them: 17 us of every tf_prepare.  A tile publishes
// its aggregate, then its inclusive prefix, as one 64-bit word {launch stamp : 30 | kind : 2 |
// value : 32} in SelBuf::offsets -- stamped, so nothing is cleared between launches; wave 0 of a
// tile looks back 64 tiles at a time (lane l at tile - 1 - l) until it meets a prefix.  A tile
// only ever waits for lower tiles, and the lowest unfinished tile waits for nobody.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kScanTile = 256;  // coarse blocks per tile: one per thread
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += (uint32_t)__shfl_xor((int)x, o);
  return x;
}
__global__ __launch_bounds__(256) void k_scan(VolumeDev v, int step, uint32_t stamp) {
  FrameCtl* ctl = v.sel.ctl;
  const SelBuf& L = v.sel;
  const uint32_t n = ctl->n_coarse;
  const uint32_t n_tiles = (n + kScanTile - 1u) / kScanTile;
  unsigned long long* const state = reinterpret_cast<unsigned long long*>(L.offsets);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  auto finish = [&](uint32_t total) {  // one thread: the list's header, the bbox keys of the set's next frame
    if (total > v.max_list) {
      atomicOr(&v.vctl->status, kStListFull);
      total = 0;
    }
    ctl->n_list = total;
    ctl->n_front = total;  // a plain list
    for (int a = 0; a < 3; ++a) {
      ctl->bbox_key[a] = f2key(1e8f);
      ctl->bbox_key[3 + a] = f2key(-1e8f);
    }
  };
  if (n_tiles == 0u) {
    if (blockIdx.x == 0 && threadIdx.x == 0) finish(0u);
    return;
  }
  auto publish = [&](uint32_t tile, uint32_t kind, uint32_t value) {
    __hip_atomic_store(&state[tile], ((unsigned long long)stamp << 34) | ((unsigned long long)kind << 32) | value, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  };
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_base;
  const uint32_t dz = (uint32_t)ctl->dims[2], dy = (uint32_t)ctl->dims[1];
  const uint32_t nzny = dz * dy;
  const int mx = ctl->min_id[0] - 1, my = ctl->min_id[1] - 1, mz = ctl->min_id[2] - 1;
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // lane = coarse block: its mask, the entries ahead of it inside the wave (ex), and where its chunks' ids start
    const uint32_t cb = tile * kScanTile + (uint32_t)threadIdx.x;
    const unsigned long long m = cb < n ? L.masks[cb] : 0ull;
    const uint32_t c = (uint32_t)__popcll(m);
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
      if (lane >= o) inc += t;
    }
    const uint32_t ex = inc - c;
    if (lane == 63) wsum[w] = inc;
    // (the two divisions per block are done here, 64 blocks at a time, not per block inside the write-out loop)
    const uint32_t bx = cb / nzny;
    const uint32_t brem = cb - bx * nzny;
    const uint32_t by = brem / dz;
    const uint32_t bz = brem - by * dz;
    const int ox = mx + (int)bx * step, oy = my + (int)by * step, oz = mz + (int)bz * step;
    __syncthreads();
    uint32_t wbase = 0, agg = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < w) wbase += wsum[k];
      agg += wsum[k];
    }
    if (w == 0) {  // the tile's place in the list: entries of all lower tiles
      uint32_t base = 0;
      if (tile == 0u) {
        if (lane == 0) publish(0u, 2u, agg);
      } else {
        if (lane == 0) publish(tile, 1u, agg);
        int look = (int)tile - 1;
        for (;;) {
          const int tt = look - lane;
          uint32_t kind = 2u, val = 0u;  // (below tile 0: an empty prefix -- the walk always ends)
          if (tt >= 0) {
            unsigned long long sw;
            do {
              sw = __hip_atomic_load(&state[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((uint32_t)(sw >> 34) != stamp || ((uint32_t)(sw >> 32) & 3u) == 0u);
            kind = (uint32_t)(sw >> 32) & 3u;
            val = (uint32_t)sw;
          }
          const unsigned long long pm = __ballot(kind == 2u);
          if (pm) {  // the nearest prefix and the aggregates between it and this tile
            const int first = __builtin_ctzll(pm);
            base += wave_sum_u32(lane <= first ? val : 0u);
            break;
          }
          base += wave_sum_u32(val);
          look -= 64;
        }
        if (lane == 0) publish(tile, 2u, base + agg);
      }
      if (lane == 0) s_base = base;
    }
    __syncthreads();
    const uint32_t tile_base = s_base;
    const uint32_t tb = tile_base + wbase;
    // list write-out: the wave expands the non-empty ones of its 64 blocks, lane = bit index = (i-x)*16 + (j-y)*4 + (k-z), the
    // reference's inner-loop order (:508-548), so entry order == push_back order.  (A list that does not fit is reported by
    // the last tile; nothing is written past the arrays.)
    unsigned long long nonempty = __ballot(m != 0ull);
    while (nonempty) {
      const int src = __builtin_ctzll(nonempty);
      nonempty &= nonempty - 1;
      const unsigned long long mm =
          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(m >> 32), src) << 32) |
          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(m & 0xFFFFFFFFu), src);
      const uint32_t off = tb + (uint32_t)__builtin_amdgcn_readlane((int)ex, src);
      const int sx = __builtin_amdgcn_readlane(ox, src), sy = __builtin_amdgcn_readlane(oy, src), sz = __builtin_amdgcn_readlane(oz, src);
      if ((mm >> lane) & 1ull) {
        const uint32_t pos = off + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
        int4 id;
        id.x = sx + ((step == 4) ? (lane >> 4) : 0);
        id.y = sy + ((step == 4) ? ((lane >> 2) & 3) : 0);
        id.z = sz + ((step == 4) ? (lane & 3) : 0);
        id.w = 0;
        if (pos < v.max_list) L.list_id[pos] = id;
      }
    }
    if (tile == n_tiles - 1u && threadIdx.x == 0) finish(tile_base + agg);
    __syncthreads();  // (wsum / s_base are written again by the workgroup's next tile)
  }
}
void launch_scan(const VolumeDev& v, int step, hipStream_t s) {
  // (one stamp per launch, process-wide: every selection set has its own state words, all that matters is that a launch
  // never meets its own stamp from an earlier launch -- 2^30 - 1 launches apart)
  static std::atomic<uint32_t> g_stamp{0};
  uint32_t stamp;
  do { stamp = (g_stamp.fetch_add(1u) + 1u) & 0x3FFFFFFFu; } while (stamp == 0u);
  hipLaunchKernelGGL(k_scan, dim3(256), dim3(256), 0, s, v, step, stamp);
}

// ---------------------------------------------------------------------------------------
// chunk lookup/creation for a device-resident list (PrepareIntersectChunks' loop,
// Chisel.h:130-138): newChunkFlag, needsUpdateFlag = false.  One thread per entry.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_acquire(VolumeDev v) {
  const SelBuf& L = v.sel;
  const uint32_t n = L.ctl->n_list;
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int4 id = L.list_id[e];
    bool is_new = false;
    uint32_t ent = 0;
    const uint32_t slot = chunk_acquire(v, id, &is_new, &ent);
    L.list_slot[e] = slot;
    L.list_ent[e] = ent;
    L.list_new[e] = is_new ? 1 : 0;
    L.list_needs[e] = 0;
  }
}
void launch_acquire(const VolumeDev& v, hipStream_t s) {
  hipLaunchKernelGGL(k_acquire, dim3(512), dim3(256), 0, s, v);
}

// Per-chunk scalars for a list that is integrated with an explicit pose (call-by-call flow: every
// IntegrateDepthScanColor call brings its own pose, Chisel.h:226).  One thread per entry.
__global__ __launch_bounds__(256) void k_pre(VolumeDev v, Pose P, Integ ig, float res, float resDiag, float4* out_pre,
                                             float* out_cen) {
  const SelBuf& L = v.sel;
  float4* pre = out_pre ? out_pre : L.list_pre;  // (the keyframe-group kernel keeps one set of records per frame)
  if (blockIdx.x == 0) centroid_table(P.p, res, out_cen ? out_cen : L.cen);
  const uint32_t n = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int4 id = L.list_id[e];
    const ChunkPre cp = chunk_pre(id, P.p, ig, res, resDiag);
    pre[4 * e] = make_float4(cp.a.x, cp.a.y, cp.a.z, cp.b.x);
    pre[4 * e + 1] = make_float4(cp.b.y, __int_as_float(id.x), __int_as_float(id.y), __int_as_float(id.z));
  }
}

// Host-supplied list (the 10-argument flow / de-integration replays kf.validChunks):
// resolve ids to slots; a missing chunk is an error (reference: chunks.at() throws).
__global__ __launch_bounds__(256) void k_lookup(VolumeDev v, uint32_t n) {
  const SelBuf& L = v.sel;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 id = L.list_id[i];
  const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
  uint32_t slot = kInvalidSlot;
  if (ent != kInvalidSlot && v.hent[ent].alive) slot = v.hent[ent].slot;
  if (slot == kInvalidSlot) atomicOr(&v.vctl->status, kStMissing);
  L.list_slot[i] = slot;
  L.list_ent[i] = (ent == kInvalidSlot) ? 0u : ent;
}
void launch_lookup(const VolumeDev& v, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_lookup, dim3((n + 255) / 256), dim3(256), 0, s, v, n);
}

// ---------------------------------------------------------------------------------------
// K-A  voxel update.  One wave64 per chunk; lane = (y, x) of the voxel, z-slice g = 0..7, so one
// slice covers 8 consecutive 8-voxel rows of the reference (row = z*8 + y).  Row-granular
// predicates of the AVX2 code (`any lane of the row`) are bytes of a 64-bit wave ballot.
// Only rows that are actually rewritten are loaded/stored (exec-masked), so HBM traffic is the
// row-granular algorithmic traffic of SURVEY.md s.8(d).
//
// The kernel is bound by dependent memory round trips, so the order of issue is:
//   chunk id -> [hash entry load]  ||  geometry of all 8 slices -> 8 depth gathers -> predicates
//   -> voxel rows of slices 0-3 (tsdf, colour, rgba) -> update -> stores -> slices 4-7 -> dirty
//   stamps.  FUSED = the per-frame unit (Chisel.h:453-468) in one launch: slot lookup/creation in
//   front (PrepareIntersectChunks' loop), FinalizeIntegrateChunks + GarbageCollect behind.
// ---------------------------------------------------------------------------------------
#ifndef TF_KF_WAVES
#define TF_KF_WAVES 7  // resident waves per SIMD the fused kernel is compiled for (register budget)
#endif
#ifndef TF_KFP_EXTRA
#define TF_KFP_EXTRA 1  // K-A workgroups per CU above the residency of the patch-carrying instance (launch_frame)
#endif
#ifndef TF_KFP_WAVES
#define TF_KFP_WAVES 7  // ... the instance that also carries the patch stage of the previous frame.  Round 6: 72 VGPRs like the others
                        // (the patch stage shed ten registers: tf_patch_body.h) -- 7168 wave slots instead of 6144 for the patch
                        // stage's 4096 waves + K-A's 7-8 k: k_frame<true,true> 43.0 -> 41.9 us (two A/B pairs, profiles/r6).  Eight
                        // bytes of private memory remain: the thread index, stored once and reloaded where a non-K-A role starts.
#endif

// K-A runs per chunk as:   64-B list record + hash entry (scalar loads)  ->  geometry of all 8
// z-slices  ->  8 depth gathers in flight  ->  slot resolve  ->  RMW passes of GP slices (predicates
// -> voxel rows + colour pixels -> arithmetic -> stores).  A wave pays one scalar, one gather and
// 8/GP voxel round trips per chunk; the kernel is bound by these dependent round trips and by the
// launch/drain of a 30-us kernel, not by VALU (2x headroom), HBM (removing every voxel access saves
// 10 %) or the vector-memory front end (staging the image footprint in LDS removes all TA FIFO
// back-pressure and changes nothing) -- profiles/r1/README.md has the ablations.
//
// Predication is done the CDNA way: every per-lane predicate is folded into the byte offset of a
// buffer load/store (out-of-range offset = no memory access, loads return 0), so there is no
// exec-mask juggling and no lane mask has to live in SGPRs across phases.  Descriptors: the three
// frame images (kernel arguments) and the chunk's two 4-KiB voxel planes (wave-uniform slot).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* const_u32x4_ptr;
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8* const_u32x8_ptr;

// TL: the tuning instance (TF_KA_DBG bit 12, tools/timeline.py) -- per-wave time stamps and work counters into the debug table
template <bool COLOR, bool QUALITY, bool FUSED, bool FLAG, int GP, bool TL = false>
__device__ __forceinline__ void integrate_body(const VolumeDev& v, const FrameImages& img, const Cam& cam,
                                               const IntegrateConsts& kc, const uint32_t epoch,
                                               const uint32_t bid, const uint32_t nb, const int claim_par = -1) {
  const SelBuf& L = v.sel;
  const int lane = threadIdx.x & 63;
  // the wave id IS wave-uniform, but anything derived from threadIdx is divergent to the compiler;
  // readfirstlane makes the uniformity provable (scalar loads, no waterfall loops around buffer ops)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((bid * 256 + threadIdx.x) >> 6));
  const uint32_t nwaves = nb * 4;
  // a list that overflowed (kStListFull is raised by its producer) holds stale tail records of an
  // older frame: the frame is skipped as a whole instead of integrating them against this image
  uint32_t n = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  uint32_t nf = n;  // entries at the front of a two-ended list
  if (FUSED) {
    const unsigned long long pk = L.ctl->emit_pack;  // complete: the selection role ran one launch ago
    nf = (uint32_t)pk;
    const unsigned long long tot = (pk & 0xFFFFFFFFull) + (pk >> 32);
    n = tot <= (unsigned long long)v.max_list ? (uint32_t)tot : 0u;
    if (!n) nf = 0;
    if (bid == 0 && threadIdx.x == 0) { L.ctl->n_list = n; L.ctl->n_front = nf; }  // for the stages behind this launch
  }
  const uint32_t n_items = n;
  // where an entry's record (and its outputs) live
  auto item_pos = [&](const uint32_t e) -> uint32_t { return FUSED ? list_phys(v, e, nf) : e; };
  const int vy = lane >> 3;
  const int W = cam.W, H = cam.H;
  if (FUSED && bid == 0 && threadIdx.x == 0) {
    // re-arm the K-B reduction of this selection set for its next frame (k_scan does this in the
    // call-by-call flow); every k_select block of this frame has finished reading the keys
    for (int a = 0; a < 3; ++a) {
      L.ctl->bbox_key[a] = f2key(1e8f);
      L.ctl->bbox_key[3 + a] = f2key(-1e8f);
    }
  }

  // Which entries does this wave take?
  //  call-by-call flow: round-robin over the list (wave w: entries w, w + nwaves, ...).
  //  fused flow: the same, over the two-ended list (costly entries first).
  const uint32_t e_first = wave, e_step = nwaves;
  // the wave's first list record is requested before anything else: it arrives while the centroid
  // table is copied (64-B records {o.xyz, wD | upper, id.xyz | spare}; the slot exists for any index)
  u32x8 rec_next;
  {
    const uint32_t e0 = e_first < n_items ? e_first : 0u;
    const uint32_t p0 = n_items ? item_pos(e0) : 0u;
    rec_next = *(const_u32x8_ptr)(unsigned long long)(&L.list_pre[4 * (p0 < v.max_list ? p0 : 0u)]);
  }

  // centroid table (Chisel.cpp:52-110), computed once per frame ahead of this launch; copied into
  // LDS and shared by the four waves of the workgroup (6 KB).
  __shared__ float cenT[3][kChunkVoxels];
  {
    const float4* src = reinterpret_cast<const float4*>(L.cen);
    float4* dst = reinterpret_cast<float4*>(&cenT[0][0]);
    for (int i = threadIdx.x; i < 3 * kChunkVoxels / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_depth =
      __builtin_amdgcn_make_buffer_rsrc((void*)img.depth, 0, W * H * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_rgba =
      __builtin_amdgcn_make_buffer_rsrc((void*)img.rgba, 0, COLOR ? W * H * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_qual =
      __builtin_amdgcn_make_buffer_rsrc((void*)img.quality, 0, QUALITY ? W * H * 4 : 0, 0x00020000);

  // "any lane of my 8-voxel row" from a wave ballot: the row's byte of the mask, picked with two
  // loop-invariant per-lane byte masks (v_and + v_and_or + v_cmp, all full rate)
  const uint32_t row_lo = lane < 32 ? (0xFFu << (lane & 24)) : 0u;
  const uint32_t row_hi = lane < 32 ? 0u : (0xFFu << (lane & 24));
  auto row_any = [&](const unsigned long long m) -> bool {
    return ((((uint32_t)m) & row_lo) | (((uint32_t)(m >> 32)) & row_hi)) != 0u;
  };
  auto ballot = [](const bool b) -> unsigned long long { return __builtin_amdgcn_ballot_w64(b); };

  // Per-pass constants live in VGPRs on purpose: as SGPRs they lose the register allocation against
  // descriptors and lane masks and get re-read from the kernel-argument segment at the top of every
  // pass (an s_load + s_waitcnt on the critical path of each of the 2.6 passes of a chunk, and on
  // the scalar cache shared between CUs).
  float c_near = cam.nearP, c_far = cam.farP, c_thr = kc.thrCol, c_lower = kc.lower, c_sigma = kc.sigma;
  asm volatile("" : "+v"(c_near), "+v"(c_far), "+v"(c_thr), "+v"(c_lower), "+v"(c_sigma));

  if (TL && (kc.dbg & 8192u) && lane == 0 && wave < (uint32_t)kPhaseWaves)  // timeline aid: prologue end
  {
    v.phase_buf[wave * 16 + 14] = __builtin_amdgcn_s_memrealtime();
    v.phase_buf[wave * 16 + 8] = 0;
    v.phase_buf[wave * 16 + 9] = 0;
  }

  for (uint32_t e = e_first, e_next = e_first + e_step; e < n_items; e = e_next) {
    const uint32_t pe = item_pos(e);  // where the entry's record and outputs live
    // The list entry (per-chunk scalars + id) was written by the previous launch, so it is read
    // through the scalar cache: one 64-B record, one s_load, one wait, off the vector-memory queue
    // of the CU (a vector load here would wait behind every gather of the other waves).
    const u32x8 prw = rec_next;
    {
      // the next record of this wave travels while this chunk is processed (speculative: the slot
      // exists even when the index is past the list, it just holds an older frame's record)
      e_next = e + e_step;
      const uint32_t en = e_next < n_items ? e_next : e;
      rec_next = *(const_u32x8_ptr)(unsigned long long)(&L.list_pre[4 * item_pos(en)]);
    }
    const int4 id = make_int4((int)prw[5], (int)prw[6], (int)prw[7], 0);
    const bool owned = part_owned(v, id.x, id.y, id.z);
    if (!owned) {
      if (FUSED && lane == 0) {
        L.list_slot[pe] = kInvalidSlot; L.list_ent[pe] = 0; L.list_new[pe] = 0; L.list_needs[pe] = 0;
        L.list_quality[pe] = 0.0f; L.list_rows[pe] = 0;
      }
      continue;
    }
    // slot lookup: ONE 16-B scalar load of the chunk's home hash entry, consumed after the
    // geometry.  The scalar cache can only be stale for entries written during THIS launch, i.e.
    // for other chunks' keys; a miss of any kind goes through the atomic path (chunk_acquire).
    const unsigned long long key = pack_id(id.x, id.y, id.z);
    const uint32_t i0 = hash_key(key) & v.hmask;
    u32x4 h0 = {0u, 0u, 0u, 0u};
    uint32_t slot = kInvalidSlot;
    if (FUSED) h0 = *(const_u32x4_ptr)(unsigned long long)(&v.hent[i0]);
    else slot = L.list_slot[pe];
    bool is_new = false;
    uint32_t ent = i0;

    // per-chunk scalars (ProjectionIntegrator.cpp:74-101), precomputed per list entry
    const float o0 = __uint_as_float(prw[0]), o1 = __uint_as_float(prw[1]), o2 = __uint_as_float(prw[2]);
    const float pbx = __uint_as_float(prw[3]), pby = __uint_as_float(prw[4]);
    const f32x2 o01 = {o0, o1};
    const f32x2 fxy = {cam.fxi, cam.fyi}, cxy = {kc.cxs, kc.cys};
    const float wD = FLAG ? pbx : -pbx;  // depth_weight *= -1 when de-integrating (:95-99)
    const float upper = pby;
    // every voxel centre of the chunk is o + c with 0 < c < 16 * res * sqrt(3): if |o.z| clears that
    // band, p.z is far inside the normal exponent range; numerators o + c are exact zeros or at
    // least one ulp of c (> 2^-40), and |o| < 2^20 keeps quotients finite -> no scaling / fix-up
    // case of the division can occur in this chunk (wave-uniform)
    const float band = 32.0f * kc.res;
    const bool div_safe = (fabsf(o2) > band) && (fabsf(o2) < 1048576.0f) && (fabsf(o0) < 1048576.0f) &&
                          (fabsf(o1) < 1048576.0f) && (kc.res > 1e-6f) && (kc.res < 16.0f);
    if (TL && (kc.dbg & 8192u) && e == e_first && lane == 0 && wave < (uint32_t)kPhaseWaves) {
      asm volatile("" :: "s"(o0), "s"(o1), "s"(o2), "s"(id.x));
      v.phase_buf[wave * 16 + 15] = __builtin_amdgcn_s_memrealtime();  // timeline aid: first entry loaded
    }

    // ---- phase 1: geometry of the 8 z-slices.  Rows run in order until the first row with no
    // valid lane -- the reference's `continue` skips `pos++` (:176-178, :420), so that row and
    // every later row of the chunk is dead: R = number of processed rows.
    int off_d[8];             // image byte offset of the lane's pixel, kOOB when the gather is masked
    int oobl[QUALITY ? 8 : 1];
    int oob_any = 0;
    uint32_t R = 64;
    // All eight slices are computed in ONE straight-line block (eight independent dependency chains
    // the scheduler can interleave); the order-dependent part -- which row stalls the chunk -- is
    // resolved afterwards from the eight validity ballots, and only for chunks that do not project
    // entirely inside the image.
    unsigned long long all_valid = ~0ull;  // (the eight validity ballots are only kept as their AND: 14 scalar registers less
                                           // at the kernel's tightest spot; the rare chunk that needs them gets them back from off_d)
    uint32_t oob_bits = 0;  // bit j: the lane's pixel of slice j is off the image
    auto geometry = [&](auto safe_tag) {
      constexpr bool SAFE = decltype(safe_tag)::value;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = j * 64 + lane;
        // x and y run as the two halves of packed-f32 instructions (each half rounded on its own)
        const f32x2 pxy = o01 + (f32x2){cenT[0][k], cenT[1][k]};
        const float pzv = o2 + cenT[2][k];
        // px / pz and py / pz (:155-164), correctly rounded; fast path when every lane is in range
        f32x2 q;
        if (SAFE) {
          q = div2_by(pxy, recip_refined(pzv));
        } else {
          q.x = pxy.x / pzv;
          q.y = pxy.y / pzv;
        }
        const f32x2 uw = q * fxy + cxy;
        // in the SAFE range no quotient is NaN, so v_cvt's own saturation classifies like x86's
        const int X = SAFE ? cvt_rne_hw(uw.x) : cvt_sat_rne(uw.x);
        const int Y = SAFE ? cvt_rne_hw(uw.y) : cvt_sat_rne(uw.y);
        // 0 < X < W-1 and 0 < Y < H-1 (:167-173) as two unsigned range tests
        const bool valid = ((unsigned)(X - 1) < (unsigned)(W - 2)) && ((unsigned)(Y - 1) < (unsigned)(H - 2));
        all_valid &= ballot(valid);
        int od = (__mul24(Y, W) + X) * 4;  // valid => 0 < Y < H, exact in 24 bits
        asm volatile("" : "+v"(od));       // keep the select a v_cndmask (no exec-mask branch)
        off_d[j] = valid ? od : kOOB;
        // X < 0 || X > W-1 || Y < 0 || Y > H-1 (:212-220); implies !valid
        if (COLOR) oob_bits |= (((unsigned)X > (unsigned)(W - 1)) || ((unsigned)Y > (unsigned)(H - 1))) ? (1u << j) : 0u;
      }
    };
    if (div_safe) geometry(std::true_type{});
    else geometry(std::false_type{});
    if (QUALITY) {
#pragma unroll
      for (int j = 0; j < 8; ++j) oobl[j] = 0;
    }
    if (all_valid != ~0ull) {  // chunks that project entirely inside the image skip all of this
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (R == 64u) {
          const unsigned long long vmj = ballot(off_d[j] != kOOB);  // (off_d[j] == kOOB <=> !valid at this point)
          const unsigned long long dead = nonzero_bytes(vmj) ^ 0x0101010101010101ull;
          if (dead) R = (uint32_t)(j * 8) + ((uint32_t)__builtin_ctzll(dead) >> 3);
        }
        const bool live_lane = (uint32_t)(j * 8 + vy) < R;
        if (COLOR) {  // off-image lanes of processed rows only
          const bool oob = live_lane && ((oob_bits >> j) & 1u);
          oob_any |= oob ? 1 : 0;
          if (QUALITY) oobl[j] = oob ? 1 : 0;
        }
        off_d[j] = live_lane ? off_d[j] : kOOB;
      }
    }

    // ---- phase 2: depth gathers (masked lanes read 0, like the reference's masked gather)
    float dep[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      dep[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_depth, off_d[j], 0, 0));

    // ---- resolve the slot once.  Fast path: the home entry holds the key.  A parked chunk (alive
    // == 0: created by an earlier frame, never updated, garbage-collected) counts as new again; it is
    // revived only if this frame updates it, so the usual "selected, outside the band, parked again"
    // round trip of the chunks in front of the surface costs no hash traffic at all.
    bool lazy_revive = false;
    if (FUSED) {
      if ((((unsigned long long)h0.y << 32) | h0.x) == key && h0.z != kInvalidSlot) {
        slot = h0.z;
        is_new = lazy_revive = (h0.w == 0u);
      } else {
        uint32_t s0 = kInvalidSlot, nw = 1, en = 0;
        if (lane == 0) {
          bool bnew = true;
          s0 = chunk_acquire(v, id, &bnew, &en);
          nw = bnew ? 1u : 0u;
        }
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0);
        is_new = __builtin_amdgcn_readfirstlane((int)nw) != 0;
        ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)en);
      }
    }
    if (slot == kInvalidSlot) {
      if (FUSED && lane == 0) {
        L.list_slot[pe] = kInvalidSlot; L.list_ent[pe] = 0; L.list_new[pe] = 0; L.list_needs[pe] = 0;
        L.list_quality[pe] = 0.0f; L.list_rows[pe] = 0;
      }
      continue;
    }
    const __amdgpu_buffer_rsrc_t rs_T =
        __builtin_amdgcn_make_buffer_rsrc((void*)(v.tsdf + (size_t)slot * kChunkVoxels), 0, 4096, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_C =
        __builtin_amdgcn_make_buffer_rsrc((void*)(v.color + (size_t)slot * kChunkVoxels), 0, 4096, 0x00020000);

    float qsum = 0.0f;
    uint32_t lanes_t = 0, lanes_c = 0;  // lanes of rewritten rows (8 per row), wave-uniform
    // classes of the voxels written (VolumeDev::summ), reduced row by row into ONE scalar word: the kernel has no
    // VGPR to spare for per-lane classes and no SGPRs for four lane masks
    uint32_t sword = 0;
    unsigned long long m_ok = 0, m_pos = 0, m_neg = 0, m_hvy = 0;

#pragma unroll
    for (int p = 0; p < 8 / GP; ++p) {
      const int g0 = p * GP;
      if ((uint32_t)(g0 * 8) >= R) break;  // a stalled row ends the chunk
      // ---- phase 3: predicates -> offsets
      float nwv[GP], sd[GP];
      int off_t[GP], off_c[GP], off_i[GP];
      unsigned long long any = 0ull;
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int gj = g0 + j;
        const int kb = (gj * 64 + lane) * 8;
        const float d = dep[gj];
        const float s = d - (o2 + cenT[2][gj * 64 + lane]);  // p.z again: an LDS read is cheaper than 8 live VGPRs
        sd[j] = s;
        if (COLOR) {
          const bool upd = (off_d[gj] != kOOB) && (fabsf(s) < c_thr);  // -thr < sd < thr (:202-208)
          off_i[j] = upd ? off_d[gj] : kOOB;
          const bool ru_l = row_any(ballot(upd));
          const unsigned long long ru = ballot(ru_l);
          off_c[j] = ru_l ? kb : kOOB;
          lanes_c += (uint32_t)__popcll(ru);
          any |= ru;
        }
        const bool act = (uint32_t)(gj * 8 + vy) < R;
        const bool dv = (d > c_near) && (c_far > d);           // (:310-312)
        const bool inside = (s > c_lower) && (upper > s);      // (:313-316)
        const bool F = act && dv && inside;
        nwv[j] = F ? wD : 0.0f;
        const bool rf_l = row_any(ballot(F));
        const unsigned long long rf = ballot(rf_l);
        off_t[j] = rf_l ? kb : kOOB;
        lanes_t += (uint32_t)__popcll(rf);
        any |= rf;
      }
      // nothing of this pass is rewritten (chunk outside the band, or a hole): skip the RMW
      // phases for the whole wave -- about a third of the selected chunks never update a row
      const bool rmw = (any != 0ull) || (COLOR && QUALITY);
      // ---- phase 4: voxel rows that will be rewritten + their inputs
      u32x2 t[GP], c[GP];
      uint32_t in[GP];
      float qv[GP];
      if (rmw) {
#pragma unroll
        for (int j = 0; j < GP; ++j) t[j] = __builtin_amdgcn_raw_buffer_load_b64(rs_T, off_t[j], 0, 0);
        if (COLOR) {
#pragma unroll
          for (int j = 0; j < GP; ++j) {
            c[j] = __builtin_amdgcn_raw_buffer_load_b64(rs_C, off_c[j], 0, 0);
            in[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_rgba, off_i[j], 0, 0);
            if (QUALITY) qv[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_qual, off_i[j], 0, 0));
          }
        }
      }
      // ---- phase 5a: arithmetic on the loaded rows
      if (rmw) {
#pragma unroll
        for (int j = 0; j < GP; ++j) {
          if (COLOR) {
            // colour planes are 4 x u16 {r, g, b, count} per voxel = two packed-u16 dwords; the
            // image pixel is widened with two byte permutes ({r, g} and {b, a} as u16 pairs)
            // (__builtin_bit_cast needs plain scalars: applied to a vector element it reads element 0)
            const uint32_t p_rg = __builtin_amdgcn_perm(0u, in[j], 0x0c010c00u);
            const uint32_t p_ba = __builtin_amdgcn_perm(0u, in[j], 0x0c030c02u);
            const uint32_t w_rg = c[j].x, w_ba = c[j].y;
            const u16x2 in_rg = __builtin_bit_cast(u16x2, p_rg), in_ba = __builtin_bit_cast(u16x2, p_ba);
            u16x2 c_rg = __builtin_bit_cast(u16x2, w_rg), c_ba = __builtin_bit_cast(u16x2, w_ba);
            if (FLAG) {  // (:274-292)
              c_rg += in_rg;
              c_ba += in_ba;
              // (short)count > 120  <=>  the dword, read as signed, is >= 121 << 16
              const bool halve = (int)__builtin_bit_cast(uint32_t, c_ba) >= (121 << 16);
              const u16x2 h_rg = c_rg >> (unsigned short)2, h_ba = c_ba >> (unsigned short)2;
              c_rg = halve ? h_rg : c_rg;
              c_ba = halve ? h_ba : c_ba;
            } else {     // (:293-304)
              c_rg -= in_rg;
              c_ba -= in_ba;
            }
            c[j].x = __builtin_bit_cast(uint32_t, c_rg);
            c[j].y = __builtin_bit_cast(uint32_t, c_ba);
          }
          const float ts = __uint_as_float(t[j].x), tw = __uint_as_float(t[j].y);  // (:319-341)
          const float nw = nwv[j];
          const float num = ts * tw + sd[j] * nw;
          const float den = (tw + nw) + c_sigma;
          const float ns = num / den;
          const float nwt = tw + nw;
          const bool keep = nwt > 0.5f;
          t[j].x = __float_as_uint(keep ? ns : 999.0f);
          t[j].y = __float_as_uint(keep ? nwt : 0.0f);
          // (a row that is not rewritten was loaded as zeros and comes out as {999, 0}: class 0)
        }
      }
      // ---- phase 5b: write back the rewritten rows only (masked offsets drop the store)
      if (rmw) {
#pragma unroll
        for (int j = 0; j < GP; ++j) {
          if (COLOR) __builtin_amdgcn_raw_buffer_store_b64(c[j], rs_C, off_c[j], 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(t[j], rs_T, off_t[j], 0, 0);
          if (!(kc.dbg & kKaCoarseSumm)) {  // lane = x + 8 y of row z = g0 + j
            const float fs = __uint_as_float(t[j].x), fw = __uint_as_float(t[j].y);
            const unsigned long long b_ok = ballot(!(fs > 1.0f)), b_pos = ballot(fs > 0.0f) & b_ok;
            const unsigned long long b_neg = ballot(fs < 0.0f), b_hvy = ballot(fw > 50.0f);
            m_ok |= b_ok; m_pos |= b_pos; m_neg |= b_neg; m_hvy |= b_hvy;
            if (g0 + j == 0)
              sword = (b_ok ? 0x1000u : 0u) | (b_pos ? 0x2000u : 0u) | (b_neg ? 0x4000u : 0u) | (b_hvy ? 0x8000u : 0u);
          }
        }
      }
      // ---- phase 6: observationQualitySum bookkeeping in row order (:212-238)
      if (COLOR && QUALITY) {
        const int rowshift = lane & 56;
#pragma unroll
        for (int j = 0; j < GP; ++j) {
          const int gj = g0 + j;
          const unsigned long long mu = ballot(off_i[j] != kOOB);
          const unsigned long long mo = ballot(oobl[gj] != 0);
          if ((mu | mo) == 0ull) continue;
          float rowsum = 0.0f;
          if (mu) {  // sum += observationQuality[i], i = 0..7 (:233-236)
#pragma unroll
            for (int l = 0; l < 8; ++l) rowsum += __shfl(qv[j], rowshift + l);
          }
          const int left = (int)R - gj * 8;
          const int rmax = left < 8 ? left : 8;
          for (int r = 0; r < rmax; ++r) {
            if ((mo >> (8 * r)) & 0xFFull) qsum = kc.qoob;
            if ((mu >> (8 * r)) & 0xFFull)
              qsum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowsum), 8 * r));
          }
        }
      }
    }
    const uint32_t rows_t = lanes_t >> 3, rows_c = COLOR ? (lanes_c >> 3) : 0u;
    const bool updated = rows_t != 0;
    if (updated) {  // the classes of what was written join the chunk's summary: lane = x + 8 y of a row
      if (kc.dbg & kKaCoarseSumm) {
        if (lane == 0) v.summ[slot] = kSummAny;
      } else {
        const unsigned long long mm[4] = {m_ok, m_pos, m_neg, m_hvy};
#pragma unroll
        for (int b = 0; b < 4; ++b)
          sword |= (mm[b] ? (1u << b) : 0u) | ((mm[b] & 0x0101010101010101ull) ? (0x10u << b) : 0u) | ((mm[b] & 0xFFull) ? (0x100u << b) : 0u);
        if (lane == 0 && sword) atomicOr(&v.summ[slot], sword);
      }
    }
    if (TL && (kc.dbg & 8192u) && lane == 0 && wave < (uint32_t)kPhaseWaves) {  // timeline aid: work of this wave
      v.phase_buf[wave * 16 + 8] += 1;                 // chunks
      v.phase_buf[wave * 16 + 9] += rows_t + rows_c;   // rows rewritten
    }
    if (COLOR && !QUALITY) {
      // without a quality image nothing is ever added: the sum ends as the out-of-observation
      // constant iff any processed row had an off-image lane (:221-222)
      if (ballot(oob_any != 0) != 0ull) qsum = kc.qoob;
    }

    // multi-GPU: remember that this slab-face chunk changed since the last boundary exchange
    const bool face = part_band(v, id.x, id.y, id.z);
    if (updated && lane == 0 && (face || lazy_revive)) {
      const uint32_t en = FUSED ? ent : L.list_ent[pe];
      if (face) mark_touched(v, en);  // bit0 alive, bit1 touched (+ listed for the next boundary pack)
      else v.hent[en].alive = 1u;
    }
    if (FUSED) {
      // FinalizeIntegrateChunks (Chisel.h:192-208) + GarbageCollect (:472-477) for this entry
      // (erase_epoch == mark_epoch + max_chunks, one allocation: both stores go through ONE pointer member.  With
      // two, the compiler merges the stores into one through a phi of the members' addresses inside the
      // kernel-argument struct, which kept the depth-only instance's copy of that struct in private memory.)
      if (updated) {
        if (lane == 0) v.mark_epoch[slot] = epoch + 1u;  // meshesToUpdate[id and 6 nbrs] = true, expanded lazily
        if (claim_par >= 0) {
          // A mesher follows this frame: the frame's dirty set (this chunk and its six face neighbours, those that
          // exist, Chisel.h:197-203) is built here, behind the chunk's stores, instead of by a kernel of its own.
          // Lane k looks neighbour k up, the per-slot stamp de-duplicates, the winner appends {id, slot} to the shard
          // list of its pool slot.  A neighbour created or revived by another wave of THIS launch may be missed: it is
          // then updated in this frame and claims itself, or it is parked again and does not exist for the mesher.  A
          // neighbour that is alive now and parked later in this launch stays in the list: the entry carries its hash
          // entry, and the mesher's filter drops chunks that are not alive (RecomputeMeshes' !HasChunk).
          claim_dirty7(v, id, slot, ent, lane, epoch + 1u, claim_par);
        }
      } else if (is_new) {
        if (lane == 0) {
          if (!lazy_revive) v.hent[ent].alive = 0;
          v.mark_epoch[(size_t)v.max_chunks + slot] = epoch + 1u;  // erase_epoch[slot]: meshesToUpdate.erase(id)
        }
        if (rows_c) {  // parked storage returns to the fresh state (only colour can be dirty)
          uint4* c4 = reinterpret_cast<uint4*>(v.color + (size_t)slot * kChunkVoxels);
#pragma unroll
          for (int k = 0; k < 4; ++k) c4[k * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
      }
      if (lane == 0) {  // what later stages read of a fused frame: slot, needsUpdate, row counts
        L.list_slot[pe] = slot;
        L.list_needs[pe] = updated ? 1 : 0;
        L.list_rows[pe] = (uint16_t)(rows_t | (rows_c << 8));
      }
    } else if (lane == 0) {
      if (updated) L.list_needs[pe] = 1;  // needsUpdateFlag[i] |= needsUpdate (Chisel.h:241)
      L.list_quality[pe] = qsum;
      L.list_rows[pe] = (uint16_t)(rows_t | (rows_c << 8));
    }
  }
}

template <bool COLOR, bool QUALITY, bool FLAG>
__global__ __launch_bounds__(256) void k_integrate(VolumeDev v, FrameImages img, Cam cam, IntegrateConsts kc,
                                                   uint32_t epoch) {
  integrate_body<COLOR, QUALITY, false, FLAG, TF_KA_GP>(v, img, cam, kc, epoch, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------
// The per-frame unit as ONE launch: independent block ranges form a software pipeline over
// consecutive frames of a stream --
//     K-A   of frame f    (reads the list K-C left one launch ago)
//     patch stage of frame f-1 (textured stream: reads the meshes the mesher of f-1 left, the images of f-1 and
//           writes texcoords / atlas texels; K-A touches voxels, summaries and epochs only -- disjoint)
//     K-C   of frame f+1  (reads the keys K-B left one launch ago)
//     K-B   of frame f+2  (into selection set f+2)
// The kernel boundary is the only synchronisation: no events, no second stream, one dispatch per
// frame.  Selection is a pure function of (depth, pose), so running it ahead changes nothing; the
// patch stage reads nothing K-A writes, so running it one launch late changes nothing either (the mesher of
// frame f, which rewrites the mesh blocks it reads, runs behind this launch).
// ---------------------------------------------------------------------------------------
struct FrameLaunch {
  VolumeDev v;           // sel = set of frame f
  FrameImages img;       // frame f
  Pose P;                // frame f
  IntegrateConsts kc;
  Cam cam;
  Integ ig;
  uint32_t epoch;
  uint32_t n_ka, n_sel, n_bbox;
  uint32_t rot;          // dispatch-order rotation of the block ranges
  SelBuf sel1;           // set of frame f+1
  const float* depth1;
  SelectConsts sc1;
  FrameCtl* ctl2;        // set of frame f+2
  const float* depth2;
  Pose P2;
  int claim_par;         // FrameStage::claim_par of frame f
  uint32_t n_patch;      // patch stage of frame f-1: workgroups, counter-set parity, the frame as keyframe
  int patch_par;
  KfDev kf_patch;
  uint32_t* progress;    // host-visible word (or null): stamped with progress_val by a workgroup of the launch, i.e. when
  uint32_t progress_val; // every launch ahead of it on the stream is through (tf_volume::h_progress)
};

// TL: the tuning instances (TF_KA_DBG / TF_PATCH_DBG set): wave timelines, phase stamps and the triage cut-offs -- the
// product instances carry none of their branches
template <bool COLOR, bool PATCH, bool TL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PATCH ? TF_KFP_WAVES : TF_KF_WAVES, PATCH ? TF_KFP_WAVES : TF_KF_WAVES))) void k_frame(FrameLaunch a) {
  // Block ranges: K-A [0, n_ka), patches [n_ka, n_ka + n_patch), K-C, K-B behind them.
  // a.rot rotates the dispatch order: 0 = K-A blocks first, n_ka = the other roles first
  const uint32_t total = a.n_ka + a.n_patch + a.n_sel + a.n_bbox;
  const uint32_t b = blockIdx.x + a.rot < total ? blockIdx.x + a.rot : blockIdx.x + a.rot - total;
  // tuning aid (dbg bit 12): per-wave {start, end, role} stamps of the last launch -> phase_buf
  const bool timeline = TL && (a.kc.dbg & 4096u) != 0 && a.n_sel > 0 && a.n_bbox > 0;  // steady launches only
  const unsigned long long t0 = timeline ? __builtin_amdgcn_s_memrealtime() : 0ull;  // 100 MHz, chip-wide
  uint32_t role;
  if (b < a.n_ka) {
    role = 0;
    integrate_body<COLOR, false, true, true, TF_KA_GP, TL>(a.v, a.img, a.cam, a.kc, a.epoch, b, a.n_ka, a.claim_par);
  } else if (PATCH && b < a.n_ka + a.n_patch) {
    role = 3;
    patch_body<true, true, true, TL>(a.v, a.cam, a.patch_par, a.kf_patch, b - a.n_ka, a.n_patch);
  } else if (b < a.n_ka + a.n_patch + a.n_sel) {
    role = 1;
    if (!(TL && (a.kc.dbg & 512u))) {  // triage switch
      VolumeDev v1 = a.v;
      v1.sel = a.sel1;
      select_body<true>(a.depth1, a.cam, a.ig, a.sc1, v1, b - a.n_ka - a.n_patch, a.n_sel);
    }
  } else {
    role = 2;
    // the progress stamp lives in this (lightest) role: next to K-A it cost 36-212 B/lane of private memory
    if (a.progress && b + 1 == total && threadIdx.x == 0)
      __hip_atomic_store(a.progress, a.progress_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!(TL && (a.kc.dbg & 1024u))) bbox_body(a.depth2, a.cam, a.P2, a.ctl2, b - a.n_ka - a.n_patch - a.n_sel, a.n_bbox);
  }
  if (timeline) {
    const uint32_t gw = (b * 256 + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0 && gw < (uint32_t)kPhaseWaves) {
      a.v.phase_buf[gw * 16 + 10] = t0;
      a.v.phase_buf[gw * 16 + 11] = __builtin_amdgcn_s_memrealtime();
      a.v.phase_buf[gw * 16 + 12] = role + 1;
      a.v.phase_buf[gw * 16 + 13] = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;  // XCC_ID
    }
  }
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// K-A grid: exactly the resident capacity (TF_KF_WAVES waves per SIMD = that many 256-thread
// workgroups per CU), so every K-A wave starts at once and walks the list with a fixed stride.
int device_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
  }
  return cus;
}
static int ka_blocks_default() { return device_cus() * TF_KF_WAVES; }

void launch_integrate(const VolumeDev& v, const FrameImages& img, const Cam& cam, const Integ& ig,
                      const Pose& pose, float res, int flag, bool use_color, bool use_quality,
                      uint32_t epoch, hipStream_t s, bool have_pre) {
  IntegrateConsts kc = make_integrate_consts(cam.cxi, cam.cyi, res, flag);
  if (!have_pre) hipLaunchKernelGGL(k_pre, dim3(256), dim3(256), 0, s, v, pose, ig, res, kc.resDiag, (float4*)nullptr, (float*)nullptr);
  static const int nblocks = ka_blocks_default();
  const dim3 grid(nblocks > 0 ? nblocks : 2048), block(256);
#define TF_LAUNCH_KA(C, Q)                                                                           \
  do {                                                                                               \
    if (flag) hipLaunchKernelGGL((k_integrate<C, Q, true>), grid, block, 0, s, v, img, cam, kc, epoch);  \
    else hipLaunchKernelGGL((k_integrate<C, Q, false>), grid, block, 0, s, v, img, cam, kc, epoch);      \
  } while (0)
  if (use_color && use_quality) TF_LAUNCH_KA(true, true);
  else if (use_color) TF_LAUNCH_KA(true, false);
  else TF_LAUNCH_KA(false, false);
#undef TF_LAUNCH_KA
}

// One pipelined launch.  Any of the three stages may be absent (pipeline fill / drain):
// cur != nullptr -> K-A of *cur (its selection set must hold a finished list); next -> K-C of
// *next (its set must hold finished K-B keys); next2 -> K-B of *next2.
void launch_frame(const VolumeDev& v, const FrameStage* cur, const FrameStage* next,
                  const FrameStage* next2, const PatchStage* patch, const Cam& cam, const Integ& ig, float res, hipStream_t s,
                  uint32_t* progress, uint32_t* progress_seq) {
  const bool with_patch = patch != nullptr && cur != nullptr && cur->img.rgba != nullptr;
  static const int nblocks7 = ka_blocks_default();
  // with the patch stage on board K-A gets one workgroup per CU more than the instance's residency: the patch / selection
  // ranges are dispatched FIRST (below) and K-A's workgroups take the slots they leave as they finish
  static const int nblocksP = device_cus() * (TF_KFP_WAVES + TF_KFP_EXTRA);
  const int nblocks = with_patch ? nblocksP : nblocks7;
  FrameLaunch a;
  a.v = v;
  a.cam = cam;
  a.ig = ig;
  a.n_ka = a.n_sel = a.n_bbox = a.n_patch = 0;
  a.patch_par = 0;
  a.claim_par = -1;
  a.epoch = 0;
  a.progress = nullptr;
  a.progress_val = 0;
  if (with_patch) {
    a.n_patch = 1024u;  // one wave per patch: 4096 waves for a room frame's ~2.9 k patches (longer lists stride)
    a.patch_par = patch->par;
    a.kf_patch = patch->kf;
  }
  // workgroups of the selection role (run 24, profiles/r3/README.md): 256 where K-A is dispatched first and the role fills
  // its tail (TSDF-only 27.5 us at 256 and 512, 30.8 at 128; hall 225 / 237 / 232 us); 128 where the patch and selection
  // ranges go first and every wave they hold delays a K-A wave (textured room: k_frame 42.2 us at 512, 40.8 at 256, 39.8 at 128)
  a.kc = make_integrate_consts(cam.cxi, cam.cyi, res, 1);
  static const int dbg = env_int("TF_KA_DBG", 0);
  a.kc.dbg = (uint32_t)dbg;
  bool color = false;
  if (cur) {
    a.v.sel = cur->sel;
    a.img = cur->img;
    a.P = cur->pose;
    a.epoch = cur->epoch;
    a.n_ka = (uint32_t)(nblocks > 0 ? nblocks : 2048);
    color = cur->img.rgba != nullptr;
    a.claim_par = cur->claim_par;
    if (cur->coarse_summ) a.kc.dbg |= kKaCoarseSumm;
  }
  if (next) {
    a.sel1 = next->sel;
    a.depth1 = next->img.depth;
    a.sc1 = make_select_consts(next->pose.p, res);
    a.n_sel = 1u;  // (sized below, once the dispatch order is known)
  }
  if (next2) {
    a.ctl2 = next2->sel.ctl;
    a.depth2 = next2->img.depth;
    a.P2 = next2->pose;
    int nvec = (cam.W * cam.H) >> 2;
    int blocks = (nvec + 255) / 256;
    a.n_bbox = (uint32_t)(blocks > 128 ? 128 : blocks);
  }
  // (a hall-sized frame -- K-A waves with ten chunks each -- wants K-A first: the other ranges then fill its tail)
  const bool others_first = with_patch && cur->small_frame;
  if (a.n_sel) a.n_sel = others_first ? 128u : 256u;
  const uint32_t total = a.n_ka + a.n_patch + a.n_sel + a.n_bbox;
  if (!total) return;
  if (progress && progress_seq && a.n_bbox) {  // (the K-B role carries the stamp: steady launches of a stream have one)
    a.progress = progress;
    a.progress_val = ++*progress_seq;
  }
  if ((a.kc.dbg & 4096u) && a.n_sel && a.n_bbox) a.kc.dbg |= 8192u;  // timeline stamps: steady launches only
  // Dispatch order.  K-A alone fills the chip, so its workgroups go first and the selection roles take the slots it
  // frees.  With the patch stage on board that order leaves the stage's 15-us chains to start when K-A's waves end
  // (profiles/r3: span 48 us); patch + selection first, K-A behind them as their waves finish: 44 us.
  a.rot = others_first ? a.n_ka : 0u;
  const bool tuning = dbg != 0 || (with_patch && a.kf_patch.pad[0] != 0);  // TF_KA_DBG / TF_PATCH_DBG: the instances with the aids
  if (with_patch) {
    if (tuning) hipLaunchKernelGGL((k_frame<true, true, true>), dim3(total), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_frame<true, true>), dim3(total), dim3(256), 0, s, a);
  } else if (color) {
    if (tuning) hipLaunchKernelGGL((k_frame<true, false, true>), dim3(total), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_frame<true, false>), dim3(total), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((k_frame<false, false>), dim3(total), dim3(256), 0, s, a);
  }
}

// ---------------------------------------------------------------------------------------
// finalize (call-by-call flow): dirty marks for updated chunks (+6 neighbours), park
// new-but-untouched chunks
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(VolumeDev v, uint32_t epoch) {
  const SelBuf& L = v.sel;
  const uint32_t n = L.ctl->n_list;
  const int sub = threadIdx.x & 7;
  __shared__ uint32_t dead[32], dsumm[32];
  __shared__ uint32_t ndead;
  for (uint32_t base = blockIdx.x * 32; base < n; base += gridDim.x * 32) {
    if (threadIdx.x == 0) ndead = 0;
    __syncthreads();
    const uint32_t e = base + (threadIdx.x >> 3);
    if (e < n) {
      const int4 id = L.list_id[e];
      const bool owned = part_owned(v, id.x, id.y, id.z);
      const bool needs = L.list_needs[e] != 0;
      const bool isnew = L.list_new[e] != 0;
      if (owned && needs && sub == 0) {
        const uint32_t slot = L.list_slot[e];
        if (slot != kInvalidSlot) v.mark_epoch[slot] = epoch + 1u;  // expanded to the 6 nbrs on read
      }
      if (owned && sub == 7 && !needs && isnew) {
        // GarbageCollect (Chisel.h:472-477): RemoveChunk + meshesToUpdate.erase
        const uint32_t slot = L.list_slot[e];
        const uint32_t ent = L.list_ent[e];
        if (slot != kInvalidSlot && v.hent[ent].alive) {
          v.hent[ent].alive = 0;
          const uint32_t di = atomicAdd(&ndead, 1u);
          dead[di] = slot;
          dsumm[di] = v.summ[slot];  // (read here, by the parking threads side by side: not one load per chunk in the loop below)
        }
        if (slot != kInvalidSlot) v.erase_epoch[slot] = epoch + 1u;  // meshesToUpdate.erase(id)
      }
    }
    __syncthreads();
    // parked storage goes back to the fresh state.  In the reference's own flow TSDF rows of such a chunk were never
    // written (updated == false), only colour rows can be (colour band hit with depth outside [near, far]).  A caller
    // that hands in other flags than the calls before produced can remove a chunk WITH data (RemoveChunk erases chunk
    // and mesh, ChunkManager.h:151-161): its summary word tells (every voxel writer ORs into it) -- then the TSDF plane
    // is reset too and the chunk's mesh leaves allMeshes.
    const uint32_t nd = ndead;
    for (uint32_t k = 0; k < nd; ++k) {
      const uint32_t ds = dead[k];
      uint4* c4 = reinterpret_cast<uint4*>(v.color + (size_t)ds * kChunkVoxels);
      c4[threadIdx.x] = make_uint4(0, 0, 0, 0);
      if (dsumm[k] != 0u) {  // (workgroup-uniform)
        const uint32_t f999 = __float_as_uint(999.0f);
        uint4* t4 = reinterpret_cast<uint4*>(v.tsdf + (size_t)ds * kChunkVoxels);
        t4[threadIdx.x] = make_uint4(f999, 0u, f999, 0u);
      }
    }
    if (threadIdx.x < nd && dsumm[threadIdx.x] != 0u) {
      const uint32_t ds = dead[threadIdx.x];
      {
        v.summ[ds] = 0u;
        MeshRec* r = &v.mesh_rec[ds];
        if (r->state & kMsInMap) { r->state = 0u; r->nv = 0; r->nt = 0; }  // (the chunk's block of the mesh store stays its own)
      }
    }
    __syncthreads();
  }
}
void launch_finalize(const VolumeDev& v, uint32_t epoch, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize, dim3(512), dim3(256), 0, s, v, epoch);
}

// ---------------------------------------------------------------------------------------
// on-demand utilities (not on the per-frame hot path)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rowstats(VolumeDev v, unsigned long long* out3) {
  const SelBuf& L = v.sel;
  const uint32_t n = L.ctl->n_list, nf = L.ctl->n_front;
  unsigned long long rt = 0, rc = 0, nu = 0;
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const uint32_t i = list_phys(v, e, nf);
    const uint32_t r = L.list_rows[i];
    rt += r & 0xFFu;
    rc += r >> 8;
    nu += L.list_needs[i] ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    rt += __shfl_xor(rt, o);
    rc += __shfl_xor(rc, o);
    nu += __shfl_xor(nu, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out3[0], rt);
    atomicAdd(&out3[1], rc);
    atomicAdd(&out3[2], nu);
  }
}
// The outputs of a call-by-call tf_integrate -- needsUpdate flags, quality sums, the status word -- written straight into
// host-visible pinned memory: one synchronisation instead of three small device-to-host copies and theirs
// (94 -> ~70 us per call; a keyframe of the reference's own sequence makes seven, tools/call_by_call_times.py).
__global__ __launch_bounds__(256) void k_export_integrate(VolumeDev v, uint32_t n, uint8_t* __restrict__ h_needs,
                                                          float* __restrict__ h_quality, uint32_t* __restrict__ h_status) {
  const SelBuf& L = v.sel;
  const uint32_t i4 = (blockIdx.x * 256 + threadIdx.x) * 4u;
  if (i4 < n) {
    if (i4 + 4 <= n) *reinterpret_cast<uint32_t*>(h_needs + i4) = *reinterpret_cast<const uint32_t*>(L.list_needs + i4);
    else for (uint32_t k = i4; k < n; ++k) h_needs[k] = L.list_needs[k];
    if (h_quality) for (uint32_t k = i4; k < n && k < i4 + 4; ++k) h_quality[k] = L.list_quality[k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *h_status = v.vctl->status;
}
void launch_export_integrate(const VolumeDev& v, uint32_t n, uint8_t* h_needs, float* h_quality, uint32_t* h_status, hipStream_t s) {
  hipLaunchKernelGGL(k_export_integrate, dim3((n / 4 + 256) / 256), dim3(256), 0, s, v, n, h_needs, h_quality, h_status);
}

// The control blocks as the host reads them after a call-by-call entry point: into pinned memory by a launch (a copy
// into pageable host memory goes through the runtime's own staging: two of them cost more than this launch + the wait)
__global__ __launch_bounds__(128) void k_export_ctl(const uint32_t* __restrict__ f, uint32_t nf, const uint32_t* __restrict__ vc,
                                                    uint32_t nv, uint32_t* __restrict__ h) {
  const uint32_t t = threadIdx.x;
  if (t < nf) h[t] = f[t];
  if (t < nv) h[nf + t] = vc[t];
}
void launch_export_ctl(const FrameCtl* f, const VolCtl* vc, uint32_t* h, hipStream_t s) {
  constexpr uint32_t nf = sizeof(FrameCtl) / 4, nv = sizeof(VolCtl) / 4;
  static_assert(nf <= 128 && nv <= 128, "one workgroup exports the control blocks");
  hipLaunchKernelGGL(k_export_ctl, dim3(1), dim3(128), 0, s, reinterpret_cast<const uint32_t*>(f), nf,
                     reinterpret_cast<const uint32_t*>(vc), nv, h);
}

// tf_prepare's outputs -- the ordered list and its isNew flags -- into host-visible memory, behind the acquire launch
__global__ __launch_bounds__(256) void k_export_list(VolumeDev v, int4* __restrict__ h_ids, uint8_t* __restrict__ h_new, uint32_t cap) {
  const SelBuf& L = v.sel;
  uint32_t n = L.ctl->n_list;
  if (n > cap) n = cap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    h_ids[i] = L.list_id[i];
    h_new[i] = L.list_new[i];
  }
}
void launch_export_list(const VolumeDev& v, int4* h_ids, uint8_t* h_new, uint32_t cap, hipStream_t s) {
  hipLaunchKernelGGL(k_export_list, dim3(128), dim3(256), 0, s, v, h_ids, h_new, cap);
}

void launch_rowstats(const VolumeDev& v, unsigned long long* out3, hipStream_t s) {
  hipLaunchKernelGGL(k_rowstats, dim3(64), dim3(256), 0, s, v, out3);
}

// Compacting appends with ONE global atomic per workgroup: a thread keeps the outcome of its (up
// to 32) candidates as a bit mask, the workgroup scans the counts, reserves its output range once
// and writes.  Same-address device atomics retire at ~10 ns per wave instruction, so an atomic per
// wave costs 160 us on a 2^20-entry table and an atomic per workgroup of 8 passes 5 us.
__device__ __forceinline__ uint32_t block_reserve(uint32_t* counter, const uint32_t mine) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t gbase;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int k = 0; k < 4; ++k) {
    if (k < w) before += wsum[k];
    total += wsum[k];
  }
  if (threadIdx.x == 0) gbase = total ? atomicAdd(counter, total) : 0u;
  __syncthreads();
  return gbase + before + inc - mine;  // first output position of this thread
}

constexpr int kListPasses = 8;  // candidates per thread (<= 32)

__global__ __launch_bounds__(256) void k_list_chunks(VolumeDev v, int4* out, uint32_t cap) {
  const uint32_t n = v.hmask + 1u;
  const uint32_t first = blockIdx.x * (256u * kListPasses) + threadIdx.x;
  uint32_t bits = 0;
#pragma unroll
  for (int k = 0; k < kListPasses; ++k) {
    const uint32_t i = first + (uint32_t)k * 256u;
    if (i < n) {
      const HEntry h = v.hent[i];
      if (h.key != kEmptyKey && h.alive && h.slot != kInvalidSlot) bits |= 1u << k;
    }
  }
  uint32_t p = block_reserve(&v.vctl->n_tmp, (uint32_t)__popc(bits));
  while (bits) {
    const int k = __builtin_ctz(bits);
    bits &= bits - 1;
    if (p < cap) out[p] = unpack_id(v.hent[first + (uint32_t)k * 256u].key);
    ++p;
  }
}
void launch_list_chunks(const VolumeDev& v, int4* out, uint32_t cap, hipStream_t s) {
  const uint32_t per = 256u * kListPasses;
  hipLaunchKernelGGL(k_list_chunks, dim3((v.hmask + per) / per), dim3(256), 0, s, v, out, cap);
}

// meshesToUpdate on demand.  For every chunk n whose mark is newer than the last clear, its seven
// candidates id in {n, n+-x, n+-y, n+-z} are examined; id is dirty iff the newest mark among ITS
// seven neighbours is newer than max(clear floor, erase(id)) (marks of one finalize precede its
// erases, Chisel.h:192-214, so an equal epoch means erased).  id is emitted by the first marked
// neighbour in a fixed order, i.e. exactly once.
__device__ __forceinline__ bool dirty_candidate(const VolumeDev& v, const uint32_t t, const uint32_t total,
                                                const uint32_t floor_, int4* id_out) {
  const uint32_t i = t >> 3, k = t & 7u;
  if (k == 7u || t >= total) return false;
  const HEntry h = v.hent[i];
  if (h.key == kEmptyKey || h.slot == kInvalidSlot || v.mark_epoch[h.slot] <= floor_) return false;
  const int4 n = unpack_id(h.key);
  const int4 id = nbr7(n, (int)k);
  *id_out = id;
  uint32_t M = 0, erase = 0;
  int first = -1;
  for (int j = 0; j < 7; ++j) {
    const int4 q = nbr7(id, j);
    const uint32_t e = hash_find(v, pack_id(q.x, q.y, q.z));
    if (e == kInvalidSlot) continue;
    const uint32_t s = v.hent[e].slot;
    if (s == kInvalidSlot) continue;
    const uint32_t m = v.mark_epoch[s];
    if (j == 0) erase = v.erase_epoch[s];
    if (m > floor_) {
      if (first < 0) first = j;
      M = m > M ? m : M;
    }
  }
  if (first < 0) return false;
  const int4 f = nbr7(id, first);
  if (f.x != n.x || f.y != n.y || f.z != n.z) return false;  // another marked neighbour emits id
  const uint32_t lim = erase > floor_ ? erase : floor_;
  // multi-GPU: an id is reported by its owner only (who alone knows whether it was erased); marks of
  // the neighbours across the slab face arrive with the boundary records
  return M > lim && part_owned(v, id.x, id.y, id.z);
}
// One thread per hash entry finds the marked chunks of the workgroup's 2048 entries; eight threads per MARKED chunk then
// examine its candidates (the table is ~1 % marked: eight threads per entry spent 8.4 M threads on a 2^20-entry table).
// Emitted ids collect in LDS and leave with one counter atomic per 1792+ of them.
constexpr uint32_t kLdEntries = 2048;
__global__ __launch_bounds__(256) void k_list_dirty(VolumeDev v, int4* out, uint32_t cap, uint32_t floor_) {
  __shared__ uint32_t s_ent[kLdEntries];
  __shared__ int4 s_out[kLdEntries];
  __shared__ uint32_t s_n, s_fill, s_base;
  const uint32_t nent = v.hmask + 1u;
  const uint32_t total = nent * 8u;
  const uint32_t e0 = blockIdx.x * kLdEntries;
  if (threadIdx.x == 0) { s_n = 0; s_fill = 0; }
  __syncthreads();
  for (uint32_t k = 0; k < kLdEntries / 256u; ++k) {
    const uint32_t i = e0 + k * 256u + threadIdx.x;
    if (i < nent) {
      const HEntry h = v.hent[i];
      if (h.key != kEmptyKey && h.slot != kInvalidSlot && v.mark_epoch[h.slot] > floor_) s_ent[atomicAdd(&s_n, 1u)] = i;
    }
  }
  __syncthreads();
  const uint32_t npair = s_n * 8u;
  for (uint32_t r0 = 0; r0 <= npair; r0 += 256u) {  // (one round more than the pairs need: the last one only flushes)
    const uint32_t idx = r0 + threadIdx.x;
    int4 id = make_int4(0, 0, 0, 0);
    if (idx < npair && dirty_candidate(v, (s_ent[idx >> 3] << 3) | (idx & 7u), total, floor_, &id)) s_out[atomicAdd(&s_fill, 1u)] = id;
    __syncthreads();
    const uint32_t fill = s_fill;
    const bool last = r0 + 256u > npair;
    __syncthreads();
    if (fill && (last || fill + 256u > kLdEntries)) {  // (workgroup-uniform)
      if (threadIdx.x == 0) { s_base = atomicAdd(&v.vctl->n_tmp, fill); s_fill = 0; }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < fill; j += 256u)
        if (s_base + j < cap) out[s_base + j] = s_out[j];
      __syncthreads();
    }
  }
}
void launch_list_dirty(const VolumeDev& v, int4* out, uint32_t cap, uint32_t clear_floor, hipStream_t s) {
  const unsigned long long nent = (unsigned long long)v.hmask + 1ull;
  hipLaunchKernelGGL(k_list_dirty, dim3((unsigned)((nent + kLdEntries - 1) / kLdEntries)), dim3(256), 0, s, v, out, cap, clear_floor);
}

// De-interleave chunks into the reference's host layouts (sdf[512], weight[512], color[2048]).
__global__ __launch_bounds__(512) void k_gather_chunks(VolumeDev v, const int4* ids, uint32_t n,
                                                       float* sdf, float* w, uint16_t* col,
                                                       uint32_t* found) {
  const uint32_t c = blockIdx.x;
  if (c >= n) return;
  const int4 id = ids[c];
  const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
  uint32_t slot = kInvalidSlot;
  if (ent != kInvalidSlot && v.hent[ent].alive) slot = v.hent[ent].slot;
  if (threadIdx.x == 0) found[c] = (slot != kInvalidSlot);
  const uint32_t k = threadIdx.x;
  float2 t = make_float2(999.0f, 0.0f);
  ushort4 cc = make_ushort4(0, 0, 0, 0);
  if (slot != kInvalidSlot) {
    t = v.tsdf[(size_t)slot * kChunkVoxels + k];
    cc = v.color[(size_t)slot * kChunkVoxels + k];
  }
  sdf[(size_t)c * 512 + k] = t.x;
  w[(size_t)c * 512 + k] = t.y;
  reinterpret_cast<ushort4*>(col)[(size_t)c * 512 + k] = cc;
}
void launch_gather_chunks(const VolumeDev& v, const int4* ids, uint32_t n, float* sdf, float* w,
                          uint16_t* col, uint32_t* found, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_gather_chunks, dim3(n), dim3(512), 0, s, v, ids, n, sdf, w, col, found);
}

__global__ __launch_bounds__(512) void k_scatter_chunk(VolumeDev v, int4 id, const float* sdf,
                                                       const float* w, const uint16_t* col) {
  __shared__ uint32_t sslot;
  if (threadIdx.x == 0) {
    bool is_new;
    uint32_t ent;
    sslot = chunk_acquire(v, id, &is_new, &ent);
  }
  __syncthreads();
  const uint32_t slot = sslot;
  if (slot == kInvalidSlot) return;
  const uint32_t k = threadIdx.x;
  if (sdf && w) {
    v.tsdf[(size_t)slot * kChunkVoxels + k] = make_float2(sdf[k], w[k]);
    const uint32_t word = wave_or(chunk_summary_bits(sdf[k], w[k], (uint32_t)k));
    if ((threadIdx.x & 63) == 0 && word) atomicOr(&v.summ[slot], word);
  }
  if (col) v.color[(size_t)slot * kChunkVoxels + k] = reinterpret_cast<const ushort4*>(col)[k];
}
void launch_scatter_chunk(const VolumeDev& v, int4 id, const float* sdf, const float* w,
                          const uint16_t* col, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter_chunk, dim3(1), dim3(512), 0, s, v, id, sdf, w, col);
}

// ---------------------------------------------------------------------------------------
// Chunk::observations on the device (view-selection bookkeeping).  The reference keeps
// std::map<int, float> per chunk (Chunk.h:171), written by IntegrateDepthScanColor (Chisel.h:244-247), erased by
// MobileFusion::RetractObservations (GCFusion/MobileFusion.cpp:252-272), read by TexMap::update_datacost
// (Structure/TexMap.cpp:64-105).  Here: one table for the volume, key = (pool slot, keyframe id).
// ---------------------------------------------------------------------------------------
// chunk->observations[keyframeID] = chunkObservationQuality where keyframeID >= 0, quality > 0 and the chunk's
// needsUpdateFlag is set (Chisel.h:244-247), for every entry of the list the last integrate call worked on
__global__ __launch_bounds__(256) void k_obs_record(VolumeDev v, int32_t kf_id) {
  const SelBuf& L = v.sel;
  const uint32_t n = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const float q = L.list_quality[e];
    const uint32_t slot = L.list_slot[e];
    if (!(q > 0.0f) || !L.list_needs[e] || slot == kInvalidSlot) continue;
    const uint32_t at = obs_find(v, obs_pack(slot, kf_id), true);
    if (at != kInvalidSlot) v.obs_q[at] = q;
  }
}
void launch_obs_record(const VolumeDev& v, int32_t kf_id, hipStream_t s) {
  if (kf_id < 0) return;
  hipLaunchKernelGGL(k_obs_record, dim3(256), dim3(256), 0, s, v, kf_id);
}
__global__ __launch_bounds__(256) void k_obs_retract(VolumeDev v, int32_t kf_id, const int4* __restrict__ ids, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 id = ids[i];
  const uint32_t slot = hash_slot_alive(v, pack_id(id.x, id.y, id.z));  // !HasChunk -> continue (:258)
  if (slot == kInvalidSlot) return;
  const uint32_t at = obs_find(v, obs_pack(slot, kf_id), false);
  if (at != kInvalidSlot) v.obs_q[at] = 0.0f;  // observations.erase(frame_id)
}
void launch_obs_retract(const VolumeDev& v, int32_t kf_id, const int4* ids, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_obs_retract, dim3((n + 255) / 256), dim3(256), 0, s, v, kf_id, ids, n);
}
__global__ __launch_bounds__(256) void k_obs_export(VolumeDev v, const int4* __restrict__ ids, uint32_t n, int32_t frame_index,
                                                    const int32_t* __restrict__ frames, int32_t m, float* __restrict__ out) {
  const uint32_t total = n * (uint32_t)(1 + m);
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const uint32_t i = t / (uint32_t)(1 + m), j = t - i * (uint32_t)(1 + m);
    const int4 id = ids[i];
    const uint32_t slot = hash_slot_alive(v, pack_id(id.x, id.y, id.z));
    float q = 0.0f;
    if (slot != kInvalidSlot) {
      const int32_t kf = j == 0 ? frame_index : frames[j - 1];
      const uint32_t at = obs_find(v, obs_pack(slot, kf), false);
      if (at != kInvalidSlot) q = v.obs_q[at];
    }
    out[t] = q;
  }
}
void launch_obs_export(const VolumeDev& v, const int4* ids, uint32_t n, int32_t frame_index, const int32_t* frames, int32_t m,
                       float* out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_obs_export, dim3(512), dim3(256), 0, s, v, ids, n, frame_index, frames, m, out);
}
__global__ __launch_bounds__(256) void k_adj_export(VolumeDev v, const int4* __restrict__ ids, uint32_t n, int4* __restrict__ out,
                                                    uint32_t cap, uint32_t* __restrict__ count) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t i = t >> 3, k = t & 7u;
  if (i >= n || k >= 6u) return;
  const int4 id = ids[i];
  const uint32_t slot = hash_slot_alive(v, pack_id(id.x, id.y, id.z));
  if (slot == kInvalidSlot) return;
  const uint32_t st = v.mesh_rec[slot].state;
  if (!(st & kMsInMap) || !((st >> (kMsAdjShift + k)) & 1u)) return;  // allMeshes.find(id) / mesh->adj[k]
  // chisel::neighbourhood (Structure/ChunkManager.h:55-57): -x, +x, -y, +y, -z, +z, the order of Mesh::adj
  int4 q = id;
  if (k == 0) q.x -= 1; else if (k == 1) q.x += 1; else if (k == 2) q.y -= 1;
  else if (k == 3) q.y += 1; else if (k == 4) q.z -= 1; else q.z += 1;
  const uint32_t qs = hash_slot_alive(v, pack_id(q.x, q.y, q.z));
  if (qs == kInvalidSlot || !(v.mesh_rec[qs].state & kMsInMap)) return;  // a chunk that never owned a mesh is no graph node
  const uint32_t p = atomicAdd(count, 1u);
  if (p < cap) out[p] = make_int4((int)i, q.x, q.y, q.z);
}
void launch_adj_export(const VolumeDev& v, const int4* ids, uint32_t n, int4* out, uint32_t cap, uint32_t* count, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_adj_export, dim3((n * 8 + 255) / 256), dim3(256), 0, s, v, ids, n, out, cap, count);
}

}  // namespace tf
