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
//   k_scan      order-preserving compaction offsets of the visible list (push_back order :544)
//   k_emit      list write-out + Chisel::PrepareIntersectChunks' HasChunk/CreateChunk  Structure/Chisel.h:130-138
//   k_integrate ProjectionIntegrator::voxelUpdateSIMD  3rd_party/open_chisel/utils/ProjectionIntegrator.cpp:67-426
//               + the per-chunk lambda of Chisel::IntegrateDepthScanColor  Structure/Chisel.h:234-248
//   k_finalize  Chisel::FinalizeIntegrateChunks + GarbageCollect  Structure/Chisel.h:184-216,472-477
#include "tf_device.h"
#include "tf_host_math.h"

#pragma clang fp contract(off)

namespace tf {

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int cvt_rne(float x) {
  return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)rintf(x) : (int)0x80000000;
}

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}

// QuadraticTruncator::GetTruncationDistance (truncation/QuadraticTruncator.h:45-48):
// |q*pow(z,2) + l*z + c| * s with the pow/sum in double, l*z in float.
__device__ __forceinline__ float truncation(const Integ& ig, float z) {
  double zz = (double)z * (double)z;
  float lz = ig.lin * z;
  double v = (double)ig.quad * zz + (double)lz + (double)ig.cons;
  return (float)(fabs(v) * (double)ig.scale);
}

__device__ __forceinline__ unsigned long long pack_id(int x, int y, int z) {
  return ((unsigned long long)((uint32_t)(x + (1 << 20)) & 0x1FFFFFu) << 42) |
         ((unsigned long long)((uint32_t)(y + (1 << 20)) & 0x1FFFFFu) << 21) |
         (unsigned long long)((uint32_t)(z + (1 << 20)) & 0x1FFFFFu);
}
__device__ __forceinline__ int4 unpack_id(unsigned long long k) {
  int4 r;
  r.x = (int)((k >> 42) & 0x1FFFFFu) - (1 << 20);
  r.y = (int)((k >> 21) & 0x1FFFFFu) - (1 << 20);
  r.z = (int)(k & 0x1FFFFFu) - (1 << 20);
  r.w = 0;
  return r;
}
__device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
  return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 24);
}

// Lookup only.  Entries are never removed, so the probe sequence of a present key is stable.
__device__ __forceinline__ uint32_t hash_find(const VolumeDev& v, unsigned long long key) {
  uint32_t i = hash_key(key) & v.hmask;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    unsigned long long cur = v.hkeys[i];
    if (cur == key) return v.hvals[i];
    if (cur == kEmptyKey) return kInvalidSlot;
    i = (i + 1) & v.hmask;
  }
  return kInvalidSlot;
}

// Find or create the pool slot of a chunk id.  Within one launch every key is unique (the
// visible list has no duplicates), so the value of a freshly inserted key is only read by
// later launches.  *is_new = chunk did not exist (absent or parked).
__device__ __forceinline__ uint32_t chunk_acquire(const VolumeDev& v, int4 id, bool* is_new) {
  const unsigned long long key = pack_id(id.x, id.y, id.z);
  uint32_t i = hash_key(key) & v.hmask;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    unsigned long long cur = v.hkeys[i];
    if (cur == kEmptyKey) {
      cur = atomicCAS(&v.hkeys[i], kEmptyKey, key);
      if (cur == kEmptyKey) {  // inserted: allocate a fresh slot
        uint32_t slot = atomicAdd(&v.ctl->slot_top, 1u);
        if (slot >= v.max_chunks) {
          atomicOr(&v.ctl->status, kStPoolFull);
          v.hvals[i] = kInvalidSlot;
          *is_new = true;
          return kInvalidSlot;
        }
        v.hvals[i] = slot;
        v.slot_id[slot] = id;
        v.alive[slot] = 1;
        atomicAdd(&v.ctl->n_alive, 1u);
        *is_new = true;
        return slot;
      }
    }
    if (cur == key) {
      uint32_t slot = v.hvals[i];
      if (slot == kInvalidSlot) { *is_new = true; return slot; }
      if (!v.alive[slot]) {  // parked chunk: storage is in the fresh state, revive it
        v.alive[slot] = 1;
        atomicAdd(&v.ctl->n_alive, 1u);
        *is_new = true;
      } else {
        *is_new = false;
      }
      return slot;
    }
    i = (i + 1) & v.hmask;
  }
  atomicOr(&v.ctl->status, kStHashFull);
  *is_new = true;
  return kInvalidSlot;
}

// meshesToUpdate[id] = true / erase(id), order-independent within one finalize epoch.
__device__ __forceinline__ void dirty_stamp(const VolumeDev& v, int x, int y, int z, uint32_t stamp) {
  const unsigned long long key = pack_id(x, y, z);
  uint32_t i = hash_key(key) & v.dmask;
  for (uint32_t probe = 0; probe <= v.dmask; ++probe) {
    unsigned long long cur = v.dkeys[i];
    if (cur == kEmptyKey) cur = atomicCAS(&v.dkeys[i], kEmptyKey, key);
    if (cur == kEmptyKey || cur == key) {
      atomicMax(&v.dstamp[i], stamp);
      return;
    }
    i = (i + 1) & v.dmask;
  }
  atomicOr(&v.ctl->status, kStHashFull);
}

// ---------------------------------------------------------------------------------------
// control block reset (create / Reset only; per-frame resets ride on k_scan)
// ---------------------------------------------------------------------------------------
__global__ void k_reset_ctl(FrameCtl* ctl) {
  if (threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) {
      ctl->bbox_key[a] = f2key(1e8f);
      ctl->bbox_key[3 + a] = f2key(-1e8f);
      ctl->min_id[a] = ctl->max_id[a] = ctl->dims[a] = 0;
    }
    ctl->n_coarse = 0; ctl->n_list = 0; ctl->status = 0; ctl->slot_top = 0;
    ctl->n_alive = 0; ctl->fin_count = 0; ctl->n_tmp = 0;
  }
}
void launch_reset_ctl(const VolumeDev& v, hipStream_t s) {
  hipLaunchKernelGGL(k_reset_ctl, dim3(1), dim3(64), 0, s, v.ctl);
}

// fresh chunk state: sdf 999, weight 0 (Chunk.cpp:64-65), colour 0 (ColorVoxel.cpp:26-33)
__global__ __launch_bounds__(256) void k_fill_pool(float2* tsdf, ushort4* color, size_t first,
                                                   size_t count) {
  const float4 f = make_float4(999.0f, 0.0f, 999.0f, 0.0f);
  const uint4 z = make_uint4(0, 0, 0, 0);
  float4* t4 = reinterpret_cast<float4*>(tsdf + first);
  uint4* c4 = reinterpret_cast<uint4*>(color + first);
  const size_t n4 = count / 2;  // two voxels per 16 B in either plane
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    t4[i] = f;
    c4[i] = z;
  }
}
void launch_fill_pool(const VolumeDev& v, uint32_t slot0, uint32_t nslots, hipStream_t s) {
  if (!nslots) return;
  size_t first = (size_t)slot0 * kChunkVoxels, count = (size_t)nslots * kChunkVoxels;
  size_t blocks = (count / 2 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_fill_pool, dim3((unsigned)blocks), dim3(256), 0, s, v.tsdf, v.color, first, count);
}

// ---------------------------------------------------------------------------------------
// K-B  world AABB of the back-projected (depth + 0.2) points
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ depth, Cam cam, Pose P,
                                              FrameCtl* ctl) {
  float mn[3] = {1e8f, 1e8f, 1e8f}, mx[3] = {-1e8f, -1e8f, -1e8f};
  const int W = cam.W;
  const int nvec = (cam.W * cam.H) >> 2;
  const float off = 0.2f;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < nvec; q += gridDim.x * 256) {
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
void launch_bbox(const VolumeDev& v, const float* depth, const Cam& cam, const Pose& pose,
                 hipStream_t s) {
  int nvec = (cam.W * cam.H) >> 2;
  int blocks = (nvec + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(k_bbox, dim3(blocks), dim3(256), 0, s, depth, cam, pose, v.ctl);
}

// ---------------------------------------------------------------------------------------
// K-C  coarse 4x4x4-block test, then per-chunk test; one 64-bit mask per coarse block
// ---------------------------------------------------------------------------------------
struct ProbeRes { bool valid; bool hit; };

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
  return r;
}

__global__ __launch_bounds__(256) void k_select(const float* __restrict__ depth, Cam cam, Integ ig,
                                                SelectConsts sc, VolumeDev v) {
  FrameCtl* ctl = v.ctl;
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
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) { ctl->min_id[a] = minI[a]; ctl->max_id[a] = maxI[a]; ctl->dims[a] = dims[a]; }
    ctl->n_coarse = n_coarse;
    if (overflow) atomicOr(&ctl->status, kStCoarseFull);
  }
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const int corner = lane & 7;
  const int step = sc.step;
  const uint32_t ngroups = (n_coarse + 7) >> 3;
  const uint32_t nzny = (uint32_t)dims[2] * (uint32_t)dims[1];

  for (uint32_t G = wave; G < ngroups; G += nwaves) {
    const uint32_t cidx = G * 8 + (lane >> 3);
    const bool inrange = cidx < n_coarse;
    uint32_t ix = 0, iy = 0, iz = 0;
    if (inrange) {
      ix = cidx / nzny;
      const uint32_t rem = cidx - ix * nzny;
      iy = rem / (uint32_t)dims[2];
      iz = rem - iy * (uint32_t)dims[2];
    }
    const int x = minI[0] - 1 + (int)ix * step;
    const int y = minI[1] - 1 + (int)iy * step;
    const int z = minI[2] - 1 + (int)iz * step;
    float oc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // :473-479
      float ox = sc.r0[a] * (float)x - sc.tc[a];
      float oy = ox + sc.r1[a] * (float)y;
      oc[a] = oy + (float)z * sc.r2[a];
    }
    const float trunc = truncation(ig, oc[2]);
    const float dtp = trunc + sc.diag_step;  // :489
    const float ndtn = -sc.dtn_coarse;       // :490, :623
    const bool depthValid = (oc[2] > cam.nearP) && (cam.farP > oc[2]);  // :598-602
    ProbeRes pr = probe(depth, cam, oc[0], oc[1], oc[2], sc.coarse[0][corner], sc.coarse[1][corner],
                        sc.coarse[2][corner], dtp, ndtn);
    const unsigned long long mh = __ballot(inrange && pr.hit && depthValid);
    // blocks of this group that passed the coarse test
    unsigned hitbits = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) hitbits |= (((mh >> (8 * b)) & 0xFFull) ? 1u : 0u) << b;
    if (corner == 0 && inrange && !((hitbits >> (lane >> 3)) & 1u)) v.masks[cidx] = 0ull;

    while (hitbits) {  // wave-uniform loop over the hit blocks
      const int b = __builtin_ctz(hitbits);
      hitbits &= hitbits - 1;
      const uint32_t cb = G * 8 + b;
      const uint32_t bx = cb / nzny;
      const uint32_t brem = cb - bx * nzny;
      const uint32_t by = brem / (uint32_t)dims[2];
      const uint32_t bz = brem - by * (uint32_t)dims[2];
      const int x0 = minI[0] - 1 + (int)bx * step;
      const int y0 = minI[1] - 1 + (int)by * step;
      const int z0 = minI[2] - 1 + (int)bz * step;
      bool flag = false;
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
        const float fdtp = tr + sc.diag;  // :528
        const float fndtn = -sc.dtn_fine; // :529
        const bool dv = (of[2] > cam.nearP) && (cam.farP > of[2]);
        bool anyhit = false;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          ProbeRes fr = probe(depth, cam, of[0], of[1], of[2], sc.fine[0][c], sc.fine[1][c],
                              sc.fine[2][c], fdtp, fndtn);
          anyhit |= fr.hit;
        }
        flag = anyhit && dv;
      }
      const unsigned long long m = __ballot(flag);
      if (lane == 0) v.masks[cb] = m;
    }
  }
}
void launch_select(const VolumeDev& v, const float* depth, const Cam& cam, const Integ& ig,
                   const Pose& pose, float res, hipStream_t s) {
  SelectConsts sc = make_select_consts(pose.p, res);
  hipLaunchKernelGGL(k_select, dim3(512), dim3(256), 0, s, depth, cam, ig, sc, v);
}

// ---------------------------------------------------------------------------------------
// exclusive scan of popcount(mask) over the coarse blocks (x-outer .. z-inner order, then lane
// order i,j,k inside a block = the reference's push_back order); single workgroup.
// Also re-arms the bbox keys for the next frame (they were consumed by k_select).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(VolumeDev v) {
  FrameCtl* ctl = v.ctl;
  const uint32_t n = ctl->n_coarse;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t b = threadIdx.x * per;
  const uint32_t e = (b + per < n) ? b + per : n;
  uint32_t local = 0;
  for (uint32_t i = b; i < e; ++i) local += (uint32_t)__popcll(v.masks[i]);
  // block exclusive scan: wave scan + LDS
  __shared__ uint32_t wsum[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
  for (int k = 0; k < 16; ++k) {
    if (k < w) wbase += wsum[k];
    total += wsum[k];
  }
  uint32_t run = wbase + inc - local;
  for (uint32_t i = b; i < e; ++i) {
    v.offsets[i] = run;
    run += (uint32_t)__popcll(v.masks[i]);
  }
  if (threadIdx.x == 0) {
    if (total > v.max_list) {
      atomicOr(&ctl->status, kStListFull);
      total = 0;
      ctl->n_coarse = 0;
    }
    ctl->n_list = total;
    for (int a = 0; a < 3; ++a) {
      ctl->bbox_key[a] = f2key(1e8f);
      ctl->bbox_key[3 + a] = f2key(-1e8f);
    }
  }
}
void launch_scan(const VolumeDev& v, hipStream_t s) {
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, v);
}

// ---------------------------------------------------------------------------------------
// list write-out + chunk lookup/creation (PrepareIntersectChunks' loop, Chisel.h:130-138)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit(VolumeDev v, int step) {
  FrameCtl* ctl = v.ctl;
  const uint32_t n_coarse = ctl->n_coarse;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t dz = (uint32_t)ctl->dims[2], dy = (uint32_t)ctl->dims[1];
  const uint32_t nzny = dz * dy;
  // each lane first scans 64 coarse blocks for non-empty masks, then the wave expands them
  for (uint32_t base = wave * 64; base < n_coarse; base += nwaves * 64) {
    const uint32_t mine = base + lane;
    const unsigned long long mymask = (mine < n_coarse) ? v.masks[mine] : 0ull;
    unsigned long long nonempty = __ballot(mymask != 0ull);
    while (nonempty) {
      const int src = __builtin_ctzll(nonempty);
      nonempty &= nonempty - 1;
      const uint32_t cb = base + src;
      const unsigned long long m =
          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(mymask >> 32), src) << 32) |
          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(mymask & 0xFFFFFFFFu), src);
      if ((m >> lane) & 1ull) {
        const uint32_t pos = v.offsets[cb] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t bx = cb / nzny;
        const uint32_t brem = cb - bx * nzny;
        const uint32_t by = brem / dz;
        const uint32_t bz = brem - by * dz;
        int4 id;
        id.x = ctl->min_id[0] - 1 + (int)bx * step + ((step == 4) ? (lane >> 4) : 0);
        id.y = ctl->min_id[1] - 1 + (int)by * step + ((step == 4) ? ((lane >> 2) & 3) : 0);
        id.z = ctl->min_id[2] - 1 + (int)bz * step + ((step == 4) ? (lane & 3) : 0);
        id.w = 0;
        bool is_new = false;
        const uint32_t slot = chunk_acquire(v, id, &is_new);
        v.list_id[pos] = id;
        v.list_slot[pos] = slot;
        v.list_new[pos] = is_new ? 1 : 0;
        v.list_needs[pos] = 0;
      }
    }
  }
}
void launch_emit(const VolumeDev& v, int step, hipStream_t s) {
  hipLaunchKernelGGL(k_emit, dim3(256), dim3(256), 0, s, v, step);
}

// Host-supplied list (the 10-argument flow / de-integration replays kf.validChunks):
// resolve ids to slots; a missing chunk is an error (reference: chunks.at() throws).
__global__ __launch_bounds__(256) void k_lookup(VolumeDev v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 id = v.list_id[i];
  uint32_t slot = hash_find(v, pack_id(id.x, id.y, id.z));
  if (slot != kInvalidSlot && !v.alive[slot]) slot = kInvalidSlot;
  if (slot == kInvalidSlot) atomicOr(&v.ctl->status, kStMissing);
  v.list_slot[i] = slot;
}
void launch_lookup(const VolumeDev& v, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_lookup, dim3((n + 255) / 256), dim3(256), 0, s, v, n);
}

// ---------------------------------------------------------------------------------------
// K-A  voxel update.  One wave64 per chunk; lane = (y, x) of the voxel, 8 passes over z, so one
// pass covers 8 consecutive 8-voxel rows of the reference (row = z*8 + y).  Row-granular
// predicates of the AVX2 code (`any lane of the row`) are bytes of a 64-bit wave ballot.
// Only rows that are actually rewritten are loaded/stored (exec-masked), so HBM traffic is the
// row-granular algorithmic traffic of SURVEY.md s.8(d).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long nonzero_bytes(unsigned long long m) {
  unsigned long long t = m | (m >> 1);
  t |= t >> 2;
  t |= t >> 4;
  return t & 0x0101010101010101ull;
}

template <bool COLOR, bool QUALITY>
__global__ __launch_bounds__(256) void k_integrate(VolumeDev v, FrameImages img, Cam cam, Integ ig,
                                                   Pose P, IntegrateConsts kc,
                                                   const uint32_t* __restrict__ n_dev) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t n = *n_dev;
  const int vx = lane & 7, vy = lane >> 3;
  const int rowshift = lane & 56;
  // centroid table pieces (Chisel.cpp:67-69): (R^T (x,y,z)) summed as p0 + (p1 + p2)
  float p0[3], p1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p0[a] = P.p[a] * (float)vx;      // R(0,a) * x
    p1[a] = P.p[4 + a] * (float)vy;  // R(1,a) * y
  }
  const int W = cam.W, H = cam.H;

  for (uint32_t e = wave; e < n; e += nwaves) {
    const uint32_t slot = v.list_slot[e];
    const int4 id = v.list_id[e];
    if (slot == kInvalidSlot) continue;
    if (id.x < v.part_lo || id.x >= v.part_hi) continue;
    // per-chunk scalars (ProjectionIntegrator.cpp:74-101, Chunk.cpp:52)
    float dvec[3];
    dvec[0] = (float)(8 * id.x) * kc.res - P.p[3];
    dvec[1] = (float)(8 * id.y) * kc.res - P.p[7];
    dvec[2] = (float)(8 * id.z) * kc.res - P.p[11];
    float o[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float q0 = P.p[a] * dvec[0], q1 = P.p[4 + a] * dvec[1], q2 = P.p[8 + a] * dvec[2];
      const float s12 = q1 + q2;
      o[a] = q0 + s12;
    }
    const float trunc = truncation(ig, o[2]);
    float wD = ig.weight / (2.0f * trunc);
    if (!kc.flag) wD *= -1.0f;
    const float upper = trunc + kc.resDiag;

    float2* __restrict__ T = v.tsdf + (size_t)slot * kChunkVoxels;
    ushort4* __restrict__ Cc = v.color + (size_t)slot * kChunkVoxels;

    float qsum = 0.0f;
    bool updated = false;
    uint32_t rows_t = 0, rows_c = 0;

    for (int g = 0; g < 8; ++g) {
      const int k = g * 64 + lane;
      float cen[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float p2 = P.p[8 + a] * (float)g;  // R(2,a) * z
        const float s12 = p1[a] + p2;
        const float d = p0[a] + s12;
        cen[a] = d * kc.res + kc.half;
      }
      const float px = o[0] + cen[0], py = o[1] + cen[1], pz = o[2] + cen[2];
      const float u = (px / pz) * cam.fxi + kc.cxs;
      const float w = (py / pz) * cam.fyi + kc.cys;
      const int X = cvt_rne(u), Y = cvt_rne(w);
      const bool valid = (X > 0) && (W - 1 > X) && (Y > 0) && (H - 1 > Y);
      const unsigned long long mv = __ballot(valid);
      // rows are processed in order up to the first row with no valid lane; the reference's
      // `continue` skips `pos++` (:176-178, :420) so every later row of the chunk is dead.
      const unsigned long long zero = ~nonzero_bytes(mv) & 0x0101010101010101ull;
      const int firstzero = zero ? (__builtin_ctzll(zero) >> 3) : 8;
      if (firstzero == 0) break;
      const bool active = vy < firstzero;
      const int idx = Y * W + X;
      float d = 0.0f;
      if (active && valid) d = img.depth[idx];
      const float sd = d - pz;

      if (COLOR) {
        const bool upd = active && valid && (sd > kc.nthrCol) && (kc.thrCol > sd);
        const bool oob = active && ((0 > X) || (X > W - 1) || (0 > Y) || (Y > H - 1));
        const unsigned long long mu = __ballot(upd);
        const unsigned long long mo = __ballot(oob);
        float rowsum = 0.0f;
        if (QUALITY) {
          if (mu) {
            float qv = 0.0f;
            if (upd) qv = img.quality[idx];
            // sum += observationQuality[i], i = 0..7 (:233-236)
#pragma unroll
            for (int l = 0; l < 8; ++l) rowsum += __shfl(qv, rowshift + l);
          }
        }
        if ((mu >> rowshift) & 0xFFull) {  // the whole row is rewritten (:267-304)
          ushort4 c = Cc[k];
          uchar4 in = make_uchar4(0, 0, 0, 0);
          if (upd) in = img.rgba[idx];
          if (kc.flag) {
            c.x = (unsigned short)(c.x + in.x);
            c.y = (unsigned short)(c.y + in.y);
            c.z = (unsigned short)(c.z + in.z);
            c.w = (unsigned short)(c.w + in.w);
            if ((short)c.w > 120) { c.x >>= 2; c.y >>= 2; c.z >>= 2; c.w >>= 2; }
          } else {
            c.x = (unsigned short)(c.x - in.x);
            c.y = (unsigned short)(c.y - in.y);
            c.z = (unsigned short)(c.z - in.z);
            c.w = (unsigned short)(c.w - in.w);
          }
          Cc[k] = c;
        }
        // observationQualitySum bookkeeping in row order (:212-238)
        for (int r = 0; r < firstzero; ++r) {
          if ((mo >> (8 * r)) & 0xFFull) qsum = kc.qoob;
          if ((mu >> (8 * r)) & 0xFFull) {
            if (QUALITY)
              qsum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowsum), 8 * r));
          }
        }
        rows_c += (uint32_t)__popcll(nonzero_bytes(mu));
      }

      const bool dv = (d > cam.nearP) && (cam.farP > d);
      const bool inside = (sd > kc.lower) && (upper > sd);
      const bool F = active && dv && inside;
      const unsigned long long mf = __ballot(F);
      if ((mf >> rowshift) & 0xFFull) {  // the whole row is rewritten (:319-341)
        float2 t = T[k];
        const float nw = F ? wD : 0.0f;
        const float num = t.x * t.y + sd * nw;
        const float den = (t.y + nw) + kc.sigma;
        const float ns = num / den;
        const float nwt = t.y + nw;
        if (nwt > 0.5f) { t.x = ns; t.y = nwt; }
        else { t.x = 999.0f; t.y = 0.0f; }
        T[k] = t;
      }
      updated |= (mf != 0ull);
      rows_t += (uint32_t)__popcll(nonzero_bytes(mf));
      if (firstzero < 8) break;
    }
    if (lane == 0) {
      if (updated) v.list_needs[e] = 1;  // needsUpdateFlag[i] |= needsUpdate (Chisel.h:241)
      v.list_quality[e] = qsum;
      v.list_rows[e] = (uint16_t)(rows_t | (rows_c << 8));
    }
  }
}

void launch_integrate(const VolumeDev& v, const FrameImages& img, const Cam& cam, const Integ& ig,
                      const Pose& pose, float res, int flag, bool use_color, bool use_quality,
                      const uint32_t* n_dev, hipStream_t s) {
  IntegrateConsts kc = make_integrate_consts(cam.cxi, cam.cyi, res, flag);
  const dim3 grid(2048), block(256);
  if (use_color && use_quality)
    hipLaunchKernelGGL((k_integrate<true, true>), grid, block, 0, s, v, img, cam, ig, pose, kc, n_dev);
  else if (use_color)
    hipLaunchKernelGGL((k_integrate<true, false>), grid, block, 0, s, v, img, cam, ig, pose, kc, n_dev);
  else
    hipLaunchKernelGGL((k_integrate<false, false>), grid, block, 0, s, v, img, cam, ig, pose, kc, n_dev);
}

// ---------------------------------------------------------------------------------------
// finalize: dirty marks for updated chunks (+6 neighbours), park new-but-untouched chunks
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(VolumeDev v, const uint32_t* __restrict__ n_dev,
                                                  uint32_t epoch) {
  const uint32_t n = *n_dev;
  const int sub = threadIdx.x & 7;
  __shared__ uint32_t dead[32];
  __shared__ uint32_t ndead;
  for (uint32_t base = blockIdx.x * 32; base < n; base += gridDim.x * 32) {
    if (threadIdx.x == 0) ndead = 0;
    __syncthreads();
    const uint32_t e = base + (threadIdx.x >> 3);
    if (e < n) {
      const int4 id = v.list_id[e];
      const bool owned = (id.x >= v.part_lo) && (id.x < v.part_hi);
      const bool needs = v.list_needs[e] != 0;
      const bool isnew = v.list_new[e] != 0;
      if (owned && needs && sub < 7) {
        const int dx = (sub == 1) ? -1 : (sub == 2) ? 1 : 0;
        const int dy = (sub == 3) ? -1 : (sub == 4) ? 1 : 0;
        const int dz = (sub == 5) ? -1 : (sub == 6) ? 1 : 0;
        dirty_stamp(v, id.x + dx, id.y + dy, id.z + dz, 2u * epoch + 1u);
      }
      if (sub == 7 && !needs && isnew) {
        // GarbageCollect (Chisel.h:472-477): RemoveChunk + meshesToUpdate.erase
        const uint32_t slot = v.list_slot[e];
        if (slot != kInvalidSlot && v.alive[slot]) {
          v.alive[slot] = 0;
          atomicSub(&v.ctl->n_alive, 1u);
          dead[atomicAdd(&ndead, 1u)] = slot;
        }
        dirty_stamp(v, id.x, id.y, id.z, 2u * epoch + 2u);
      }
    }
    __syncthreads();
    // parked storage goes back to the fresh state: TSDF rows were never written (updated ==
    // false), only colour rows can be (colour band hit with depth outside [near, far]).
    const uint32_t nd = ndead;
    for (uint32_t k = 0; k < nd; ++k) {
      uint4* c4 = reinterpret_cast<uint4*>(v.color + (size_t)dead[k] * kChunkVoxels);
      c4[threadIdx.x] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
  }
}
void launch_finalize(const VolumeDev& v, const uint32_t* n_dev, uint32_t epoch, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize, dim3(512), dim3(256), 0, s, v, n_dev, epoch);
}

// ---------------------------------------------------------------------------------------
// on-demand utilities (not on the per-frame hot path)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rowstats(VolumeDev v, const uint32_t* __restrict__ n_dev,
                                                  unsigned long long* out3) {
  const uint32_t n = *n_dev;
  unsigned long long rt = 0, rc = 0, nu = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t r = v.list_rows[i];
    rt += r & 0xFFu;
    rc += r >> 8;
    nu += v.list_needs[i] ? 1 : 0;
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
void launch_rowstats(const VolumeDev& v, const uint32_t* n_dev, unsigned long long* out3,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_rowstats, dim3(64), dim3(256), 0, s, v, n_dev, out3);
}

__global__ __launch_bounds__(256) void k_list_chunks(VolumeDev v, int4* out, uint32_t cap) {
  const uint32_t top = v.ctl->slot_top < v.max_chunks ? v.ctl->slot_top : v.max_chunks;
  for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < top; s += gridDim.x * 256) {
    if (v.alive[s]) {
      const uint32_t p = atomicAdd(&v.ctl->n_tmp, 1u);
      if (p < cap) out[p] = v.slot_id[s];
    }
  }
}
void launch_list_chunks(const VolumeDev& v, int4* out, uint32_t cap, hipStream_t s) {
  hipLaunchKernelGGL(k_list_chunks, dim3(256), dim3(256), 0, s, v, out, cap);
}

__global__ __launch_bounds__(256) void k_list_dirty(VolumeDev v, int4* out, uint32_t cap) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i <= v.dmask; i += gridDim.x * 256) {
    if (v.dkeys[i] != kEmptyKey && (v.dstamp[i] & 1u)) {
      const uint32_t p = atomicAdd(&v.ctl->n_tmp, 1u);
      if (p < cap) out[p] = unpack_id(v.dkeys[i]);
    }
  }
}
void launch_list_dirty(const VolumeDev& v, int4* out, uint32_t cap, hipStream_t s) {
  hipLaunchKernelGGL(k_list_dirty, dim3(512), dim3(256), 0, s, v, out, cap);
}

// De-interleave chunks into the reference's host layouts (sdf[512], weight[512], color[2048]).
__global__ __launch_bounds__(512) void k_gather_chunks(VolumeDev v, const int4* ids, uint32_t n,
                                                       float* sdf, float* w, uint16_t* col,
                                                       uint32_t* found) {
  const uint32_t c = blockIdx.x;
  if (c >= n) return;
  const int4 id = ids[c];
  uint32_t slot = hash_find(v, pack_id(id.x, id.y, id.z));
  if (slot != kInvalidSlot && !v.alive[slot]) slot = kInvalidSlot;
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
    sslot = chunk_acquire(v, id, &is_new);
  }
  __syncthreads();
  const uint32_t slot = sslot;
  if (slot == kInvalidSlot) return;
  const uint32_t k = threadIdx.x;
  if (sdf && w) v.tsdf[(size_t)slot * kChunkVoxels + k] = make_float2(sdf[k], w[k]);
  if (col) v.color[(size_t)slot * kChunkVoxels + k] = reinterpret_cast<const ushort4*>(col)[k];
}
void launch_scatter_chunk(const VolumeDev& v, int4 id, const float* sdf, const float* w,
                          const uint16_t* col, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter_chunk, dim3(1), dim3(512), 0, s, v, id, sdf, w, col);
}

// ---- multi-GPU boundary exchange ------------------------------------------------------
// Pack the chunks of the current list that this rank owns, that were updated, and that sit
// on a partition face (x == lo or x == hi-1).  Record: int4 id | float2[512] | ushort4[512].
__global__ __launch_bounds__(512) void k_boundary_pack(VolumeDev v, const uint32_t* __restrict__ n_dev,
                                                       uint8_t* records, uint32_t cap) {
  const uint32_t n = *n_dev;
  __shared__ uint32_t spos;
  for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
    const int4 id = v.list_id[e];
    const bool face = (id.x == v.part_lo) || (id.x == v.part_hi - 1);
    const bool owned = (id.x >= v.part_lo) && (id.x < v.part_hi);
    const uint32_t slot = v.list_slot[e];
    if (!(face && owned && v.list_needs[e] && slot != kInvalidSlot)) continue;  // block-uniform
    if (threadIdx.x == 0) spos = atomicAdd(&v.ctl->n_tmp, 1u);
    __syncthreads();
    const uint32_t p = spos;
    if (p < cap) {
      uint8_t* rec = records + (size_t)p * (16 + 4096 + 4096);
      if (threadIdx.x == 0) *reinterpret_cast<int4*>(rec) = id;
      reinterpret_cast<float2*>(rec + 16)[threadIdx.x] = v.tsdf[(size_t)slot * kChunkVoxels + threadIdx.x];
      reinterpret_cast<ushort4*>(rec + 16 + 4096)[threadIdx.x] = v.color[(size_t)slot * kChunkVoxels + threadIdx.x];
    }
    __syncthreads();
  }
}
void launch_boundary_pack(const VolumeDev& v, const uint32_t* n_dev, uint8_t* records, uint32_t cap,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_boundary_pack, dim3(1024), dim3(512), 0, s, v, n_dev, records, cap);
}

// Store received records of chunks this rank does not own as ghost chunks.
__global__ __launch_bounds__(512) void k_boundary_unpack(VolumeDev v, const uint8_t* records, uint32_t n) {
  __shared__ uint32_t sslot;
  for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
    const uint8_t* rec = records + (size_t)r * (16 + 4096 + 4096);
    const int4 id = *reinterpret_cast<const int4*>(rec);
    const bool owned = (id.x >= v.part_lo) && (id.x < v.part_hi);
    if (owned) continue;  // block-uniform
    if (threadIdx.x == 0) {
      bool is_new;
      sslot = chunk_acquire(v, id, &is_new);
    }
    __syncthreads();
    const uint32_t slot = sslot;
    if (slot != kInvalidSlot) {
      v.tsdf[(size_t)slot * kChunkVoxels + threadIdx.x] = reinterpret_cast<const float2*>(rec + 16)[threadIdx.x];
      v.color[(size_t)slot * kChunkVoxels + threadIdx.x] = reinterpret_cast<const ushort4*>(rec + 16 + 4096)[threadIdx.x];
    }
    __syncthreads();
  }
}
void launch_boundary_unpack(const VolumeDev& v, const uint8_t* records, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_boundary_unpack, dim3(n < 1024 ? n : 1024), dim3(512), 0, s, v, records, n);
}

}  // namespace tf
