// tf_patch_body.h -- the per-patch stage of the atlas path as a device function, so that it can run as a kernel of
// its own (k_patch, tf_atlas.hip: the reference's separate GeneratePatches / UpdateAtlas entry points) and as a block
// range of the per-frame launch (k_frame, tf_kernels.hip: the patches of frame f - 1 next to the voxel update of
// frame f).  One WAVE per patch:
//   Patch::CalculateTexCoords + SetFrameid + SetImage   Structure/Patch.cpp:40-108,172-175 (GeneratePatches' loop body,
//                                                        Structure/Chisel.cpp:156-181)
//   Atlas::AddPatch for the fused flow's new patches     Structure/Atlas.cpp:43-64 (slot = rank in ascending chunk id)
//   Chisel::CompressMeshes' neighbour exchange           Structure/Chisel.cpp:127-145
//   Atlas::UpdateBuffer                                  Structure/Atlas.cpp:71-91
#pragma once

#include "tf_devfn.h"
#include "tf_device.h"

#pragma clang fp contract(off)

namespace tf {

// ---------------------------------------------------------------------------------------
// image access.  cv::Mat::at is unchecked pointer arithmetic: x == W lands on the next row.  Reads past
// the image (undefined in the reference) return 0.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void rgb_at(const uint8_t* rgb, int stride, int W, int H, int y, int x, float c[3]) {
  const long i = (long)y * W + x;
  if (i < 0 || i >= (long)W * H) { c[0] = c[1] = c[2] = 0.0f; return; }
  const uint8_t* p = rgb + (size_t)i * stride;
  c[0] = (float)p[0]; c[1] = (float)p[1]; c[2] = (float)p[2];
}
__device__ __forceinline__ float f_at(const float* img, int W, int H, int y, int x) {
  const long i = (long)y * W + x;
  if (i < 0 || i >= (long)W * H) return 0.0f;
  return img[i];
}

// Patch::bilinear (Patch.cpp:110-145) -- c2 stands where c4 belongs (:125-128).
__device__ __forceinline__ void bilinear_rgb(const uint8_t* rgb, int stride, int W, int H, float lx, float ly,
                                             float out[3]) {
  const int x = (int)floorf(lx), y = (int)floorf(ly);
  float c1[3], c2[3], c3[3];
  if (x < W - 1 && y < H - 1) {
    rgb_at(rgb, stride, W, H, y, x, c1); rgb_at(rgb, stride, W, H, y, x + 1, c2); rgb_at(rgb, stride, W, H, y + 1, x, c3);
    const float ax = (float)(x + 1) - lx, bx = lx - (float)x;
    const float ay = (float)(y + 1) - ly, by = ly - (float)y;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float t = (c1[k] * ax) * ay;
      t = t + (c2[k] * bx) * ay;
      t = t + (c3[k] * ax) * by;
      t = t + (c2[k] * bx) * by;
      out[k] = t;
    }
  } else if (x < W - 1 && y == H - 1) {
    rgb_at(rgb, stride, W, H, y, x, c1); rgb_at(rgb, stride, W, H, y, x + 1, c2);
    const float ax = (float)(x + 1) - lx, bx = lx - (float)x;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = c1[k] * ax + c2[k] * bx;
  } else if (x == W - 1 && y < H - 1) {
    rgb_at(rgb, stride, W, H, y, x, c1); rgb_at(rgb, stride, W, H, y + 1, x, c2);
    const float ay = (float)(y + 1) - ly, by = ly - (float)y;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = c1[k] * ay + c2[k] * by;
  } else {
    rgb_at(rgb, stride, W, H, y, x, out);
  }
}
// Patch::bilinear_depth (Patch.cpp:147-170)
__device__ __forceinline__ float bilinear_f(const float* img, int W, int H, float lx, float ly) {
  const int x = (int)floorf(lx), y = (int)floorf(ly);
  if (x < W - 1 && y < H - 1) {
    const float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y, x + 1), c3 = f_at(img, W, H, y + 1, x);
    const float ax = (float)(x + 1) - lx, bx = lx - (float)x;
    const float ay = (float)(y + 1) - ly, by = ly - (float)y;
    float t = (c1 * ax) * ay;
    t = t + (c2 * bx) * ay;
    t = t + (c3 * ax) * by;
    t = t + (c2 * bx) * by;
    return t;
  } else if (x < W - 1 && y == H - 1) {
    const float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y, x + 1);
    return c1 * ((float)(x + 1) - lx) + c2 * (lx - (float)x);
  } else if (x == W - 1 && y < H - 1) {
    const float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y + 1, x);
    return c1 * ((float)(y + 1) - ly) + c2 * (ly - (float)y);
  }
  return f_at(img, W, H, y, x);
}

// n-th slot the allocator hands out (Atlas.cpp:48-58): x advances by PW and wraps to the next band of PH
// rows once x + PW >= AW, so a band holds K = ceil(AW / PW) slots; the hand-out fails once y >= AH.
__host__ __device__ inline bool slot_texloc(int atlas_w, int atlas_h, int pw, int ph, unsigned long long n,
                                            unsigned long long* texloc) {
  const unsigned long long K = ((unsigned long long)atlas_w + pw - 1) / (unsigned long long)pw;
  const unsigned long long band = n / K, k = n - band * K;
  const unsigned long long y = band * (unsigned long long)ph;
  *texloc = k * (unsigned long long)pw + y * (unsigned long long)atlas_w;
  return y < (unsigned long long)atlas_h;
}
__device__ __forceinline__ uint32_t mesh_shard_rows_dev(uint32_t max_chunks) { return max_chunks / kMeshShards + 258u; }  // = mesh_shard_rows()
__device__ __forceinline__ bool slot_texloc(const VolumeDev& v, unsigned long long n, unsigned long long* texloc) {
  int aw = v.atlas_w, pw = v.patch_w;
  asm volatile("" : "+s"(aw), "+s"(pw));  // (divisors kept opaque: see the blit)
  return slot_texloc(aw, v.atlas_h, pw, v.patch_h, n, texloc);
}

// ---------------------------------------------------------------------------------------
// fused per-frame flow: the work list is the frame's dirty set, unordered.  k_compress_exchange (tf_mesh.hip) keeps the
// entries that have a mesh and lists the ones without an atlas slot ("candidates").  The slots go out in ascending
// chunk-id order (the harness' definition of chunksToUpdate's order): a candidate's slot is slots_base + its rank,
// rank = number of candidates with a smaller key, which the candidate's own wave counts in the patch kernel (a
// few hundred candidates in steady state, a few thousand on first touch).
// ---------------------------------------------------------------------------------------
// One wave per patch.  PROJECT = Patch::CalculateTexCoords + SetFrameid + SetImage (GeneratePatches' loop
// body), BLIT = Atlas::UpdateBuffer.  FUSED = the work list is the frame's dirty set of the fused flow
// (patch_begin here; slots of new patches by self-ranking; on atlas overflow the entries behind the first failing
// AddPatch in id order are skipped, Chisel.cpp:170-173).
// ---------------------------------------------------------------------------------------
typedef uint32_t u32_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ int cv_round_f(float x) { return (int)rintf(x); }

// ---- gather helpers of the batched projection: addresses first, all loads in flight, arithmetic last ----------
// Patch::bilinear / bilinear_depth (Patch.cpp:110-170) read up to three pixels: (y, x), then (y, x + 1) or
// (y + 1, x), then (y + 1, x).  kind: 0 = interior (c1, c2, c3; c2 stands where c4 belongs, :125-128),
// 1 = last row (c1, c2 along x), 2 = last column (c1, c2 along y), 3 = corner / beyond (c1 only).
struct Taps {
  int i1, i2, i3;  // linear pixel indices, -1 = outside the image (reads 0, see rgb_at)
};
// ... and the weights, recomputed from the position when the pixels have arrived (the same expressions give the same values;
// holding kind + four weights per vertex block across the gathers costs the fused kernel ten registers it does not have)
struct TapW {
  int kind;
  float ax, bx, ay, by;
};
__device__ __forceinline__ int pix_index(int W, int H, int y, int x) {
  const long i = (long)y * W + x;
  return (i < 0 || i >= (long)W * H) ? -1 : (int)i;
}
__device__ __forceinline__ int tap_kind(int W, int H, int x, int y) {
  return (x < W - 1 && y < H - 1) ? 0 : ((x < W - 1 && y == H - 1) ? 1 : ((x == W - 1 && y < H - 1) ? 2 : 3));
}
__device__ __forceinline__ Taps make_taps(int W, int H, float lx, float ly) {
  Taps t;
  const int x = (int)floorf(lx), y = (int)floorf(ly);
  const int kind = tap_kind(W, H, x, y);
  t.i1 = pix_index(W, H, y, x);
  t.i2 = t.i3 = -1;
  if (kind == 0) { t.i2 = pix_index(W, H, y, x + 1); t.i3 = pix_index(W, H, y + 1, x); }
  else if (kind == 1) t.i2 = pix_index(W, H, y, x + 1);
  else if (kind == 2) t.i2 = pix_index(W, H, y + 1, x);
  return t;
}
__device__ __forceinline__ TapW tap_weights(int W, int H, float lx, float ly) {
  TapW t;
  const int x = (int)floorf(lx), y = (int)floorf(ly);
  t.ax = (float)(x + 1) - lx; t.bx = lx - (float)x;
  t.ay = (float)(y + 1) - ly; t.by = ly - (float)y;
  t.kind = tap_kind(W, H, x, y);
  return t;
}
// one pixel as r | g << 8 | b << 16 (0 outside the image)
__device__ __forceinline__ uint32_t load_px(const uint8_t* rgb, int stride, int i) {
  if (i < 0) return 0u;
  if (stride == 4) return *reinterpret_cast<const uint32_t*>(rgb + 4 * (size_t)i) & 0xFFFFFFu;
  const uint8_t* p = rgb + 3 * (size_t)i;
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
__device__ __forceinline__ float blend(const TapW& t, float c1, float c2, float c3) {
  if (t.kind == 0) {
    float r = (c1 * t.ax) * t.ay;
    r = r + (c2 * t.bx) * t.ay;
    r = r + (c3 * t.ax) * t.by;
    r = r + (c2 * t.bx) * t.by;
    return r;
  }
  if (t.kind == 1) return c1 * t.ax + c2 * t.bx;
  if (t.kind == 2) return c1 * t.ay + c2 * t.by;
  return c1;
}

#ifndef TF_PATCH_KVB
#define TF_PATCH_KVB 2
#endif
#ifndef TF_PATCH_KB
#define TF_PATCH_KB 8
#endif
constexpr int kVB = TF_PATCH_KVB;  // 64-vertex blocks of a patch kept in registers: one sweep for meshes up to 128 vertices

// bid / nb: this workgroup's index among the nb 256-thread workgroups that run the stage
// TL: the tuning instance (TF_PATCH_DBG, tools/stamps.py patch): phase stamps (3) and the triage cut-offs (1, 2) of
// KfDev::pad[0]; the product instances carry none of it
template <bool PROJECT, bool BLIT, bool FUSED, bool TL = false>
__device__ __forceinline__ void patch_body(const VolumeDev& v, const Cam& cam, const int par, const KfDev& kf_fused,
                                           const uint32_t bid, const uint32_t nb) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((bid * 256 + threadIdx.x) >> 6));
  const uint32_t nwaves = nb * 4;
  AtlasCtl::Set* S = &v.actl->set[par];
  // tuning aid (TF_PATCH_DBG=3, tools/stamps.py): lane 0 of a wave stamps the phases of its patch into the debug table,
  // row = wave; a phase that ends in loads is closed with a wait so that the stamp means "data arrived"
  const bool tl = TL && FUSED && kf_fused.pad[0] == 3 && wave < (uint32_t)kPhaseWaves;
  auto stampw = [&](int k) {
    if (tl) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) v.phase_buf[wave * 16 + k] = __builtin_amdgcn_s_memrealtime();
    }
  };
  if (tl && lane == 0) v.phase_buf[wave * 16 + 0] = __builtin_amdgcn_s_memrealtime();
  // fused flow: the frame's dirty chunks that own a mesh sit in 32 shard lists (the mesher and its filter
  // appended them); wave w walks shard w % 32
  const uint32_t shard = wave & (kMeshShards - 1u);
  const uint32_t shard_rows = mesh_shard_rows_dev(v.max_chunks);
  const int4* plist = v.patch_list + ((size_t)(par & 1) * kMeshShards + shard) * shard_rows;
  // the wave's first list entry is requested TOGETHER with the shard's counter and the frame's slot counters (its address does
  // not depend on the count; a position beyond the count holds an older frame's entry, dropped when the count arrives): one
  // dependent round trip less per patch -- most waves have exactly one
  const uint32_t pe0 = FUSED ? wave / kMeshShards : wave;
  int4 e_first = make_int4(0, 0, 0, 0);
  if (FUSED) e_first = plist[pe0 < shard_rows ? pe0 : 0u];
  uint32_t n = S->n_work;
  if (FUSED) {
    n = v.patch_cnt[((par & 1) * kMeshShards + shard) * 16];
    if (n > shard_rows) n = shard_rows;
  }
  // FUSED: slot hand-out.  total = slots the atlas holds (slot n exists iff (n / K) * PH < AH), room = slots left
  // before this frame, n_cand = patches of this frame that need one.
  const uint32_t n_cand = FUSED ? S->n_cand : 0u;
  const unsigned long long slots_base = FUSED ? S->slots_base : 0ull;
  // (a function of the volume's sizes and the frame's first slot: recomputed by the few patches that need it -- as a
  // 64-bit per-lane value held across the patch loop it is two registers the fused kernel does not have at 72)
  auto slots_room = [&](const unsigned long long base) -> unsigned long long {
    int aw = v.atlas_w, ah = v.atlas_h, pw = v.patch_w, ph = v.patch_h;
    asm volatile("" : "+s"(aw), "+s"(ah), "+s"(pw), "+s"(ph));
    const unsigned long long K = ((unsigned long long)aw + pw - 1) / (unsigned long long)pw;
    const unsigned long long bands = ((unsigned long long)ah + ph - 1) / (unsigned long long)ph;
    const unsigned long long total = K * bands;
    return total > base ? total - base : 0ull;
  };
  const unsigned long long room = FUSED ? slots_room(slots_base) : 0ull;
  const bool overflow = FUSED && (unsigned long long)n_cand > room;  // some AddPatch of this frame throws
  const int W = cam.W, H = cam.H;
  const float Wf = (float)W, Hf = (float)H;
  if (FUSED && bid == 0 && threadIdx.x == 0) {
    // account for the slots this frame hands out (the waves of this kernel read the snapshot slots_base, the next
    // reader of n_slots is the next frame's k_compress_exchange, behind this kernel on the same stream)
    const unsigned long long got = (unsigned long long)n_cand < room ? (unsigned long long)n_cand : room;
    v.actl->n_slots = (uint32_t)(slots_base + got);
    if (overflow) atomicOr(&v.vctl->status, kStAtlasFull);
  }
  for (uint32_t pe = pe0; pe < n; pe += FUSED ? nwaves / kMeshShards : nwaves) {
    // fused flow: {packed id, pool slot, mesh block} straight from the list (one dependent load less per patch: the vertex
    // loads below need not wait for the record); the call-by-call flow reads the block from the record
    int4 id;
    uint32_t slot, lblk = kBlkNone;
    if (FUSED) {
      const int4 e = pe == pe0 ? e_first : plist[pe];
      id = unpack_id(((unsigned long long)(uint32_t)e.y << 32) | (uint32_t)e.x);
      slot = (uint32_t)e.z;
      lblk = (uint32_t)e.w;
    } else {
      id = v.work_ids[pe];
      slot = v.work_slot[pe];
    }
    if (slot == kInvalidSlot) continue;
    MeshRec* rec = &v.mesh_rec[slot];
    MeshRec R = *rec;  // one 64-B record: counts, flags, slot position, box
    const uint32_t mst = FUSED ? lblk : R.block;  // (the block of the mesh store that holds the mesh)
    stampw(1);
    // Chisel::CompressMeshes' neighbour exchange (Chisel.cpp:127-145) for this chunk: flag k of the mesh and flag k ^ 1 of its
    // k-th face neighbour's mesh become the OR of the two.  Every mesh of the frame is complete (the mesher ran before this
    // kernel); the pairwise updates are idempotent, so concurrent waves cannot disagree.
    // The neighbours' pool slots come from the chunk's row of the neighbour table (lanes 0..5: the face words, requested with
    // the record above) -- a mesh lives in the record of its chunk's pool slot, so nothing else of the neighbour is needed.
    // The row was checked in full by the filter launch ahead of the mesher that listed this chunk; a chunk inserted since
    // then has no mesh yet (only a mesher makes one, and this stage runs ahead of the next mesher).  The neighbour's state is
    // requested here, unconditionally (a lane without a neighbour reads its own record), and looked at behind the vertex
    // loads of the first sweep: the exchange adds no round trip to the patch's chain.
    int xlane = lane;
    asm volatile("" : "+v"(xlane));  // (or the lane's address into the table is hoisted out of the patch loop -- into private memory)
    const int xk = xlane < 6 ? xlane : 0, xm = xk ^ 1;
    uint32_t x_nw = 0u, x_bs = 0u;
    if (FUSED) {
      const int word = 13 + ((xk & 1) ? 1 : -1) * (xk < 2 ? 1 : (xk < 4 ? 3 : 9));
      x_nw = xlane < 6 ? v.nbr[(size_t)slot * kNbrWords + word] : 0u;
      x_bs = v.mesh_rec[x_nw ? x_nw - 1u : slot].state;
    }
    auto exchange_flags = [&]() {
      if (FUSED && x_nw != 0u && (x_bs & kMsInMap) && (x_bs & kMsSimplified)) {
        const uint32_t abit = 1u << (kMsAdjShift + xk), bbit = 1u << (kMsAdjShift + xm);
        const bool fa = (R.state & abit) != 0, fb = (x_bs & bbit) != 0;
        if (fa && !fb) atomicOr(&v.mesh_rec[x_nw - 1u].state, bbit);
        if (!fa && fb) atomicOr(&rec->state, abit);
      }
    };
    if (!PROJECT) exchange_flags();
    if (FUSED) {
      const bool cand = R.texloc == kNoTexloc;
      if (cand || overflow) {
        // c = candidates with a smaller key = this patch's rank if it is one itself
        const unsigned long long key = pack_id(id.x, id.y, id.z);
        uint32_t c = 0;
        int l3 = lane;
        asm volatile("" : "+v"(l3));  // (the lane's address into the candidate list would be hoisted out of the patch loop -- into private memory)
        for (uint32_t i0 = 0; i0 < n_cand; i0 += 256) {  // (four independent loads at a time: one at a time is a chain of round trips)
          unsigned long long ck[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t i = i0 + 64u * q + (uint32_t)l3;
            ck[q] = i < n_cand ? v.cand[i] : ~0ull;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) c += ck[q] < key ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
        // the first failing AddPatch is the candidate of rank `room`; it and every entry behind it are skipped
        const AtlasCtl::Set* S2 = S;
        asm volatile("" : "+v"(S2));  // (the frame's first slot is read again here: see slots_room)
        const unsigned long long base2 = S2->slots_base;
        if (overflow && (unsigned long long)c >= slots_room(base2) + (cand ? 0ull : 1ull)) { exchange_flags(); continue; }  // (CompressMeshes ran for it all the same)
        if (cand) {
          unsigned long long tl;
          if (!slot_texloc(v, base2 + c, &tl)) { exchange_flags(); continue; }  // (cannot happen: c < room)
          R.texloc = tl;
          if (lane == 0) rec->texloc = tl;
        }
      }
    }
    stampw(2);
    const uint32_t nv = R.nv;
    // fused flow: the keyframe is the frame itself, handed over by value; its image is not retained (kf_slot -1)
    const int kf_slot = FUSED ? -1 : (PROJECT ? id.w : R.kf_slot);
    if (!PROJECT && (!(R.pflags & kPfHasPatch) || kf_slot < 0)) continue;  // GetPatch == nullptr (Atlas.cpp:73-74)
    const KfDev kf = FUSED ? kf_fused : v.kf_tab[kf_slot];
    int bx = 0, by = 0, cols = 0, rows = 0;
    bool have_image = false;
    if (TL && FUSED && kf_fused.pad[0] == 2) { exchange_flags(); continue; }  // triage: list walk + record read only
    if (PROJECT) {
      float minX = Wf, maxX = 0.0f, minY = Hf, maxY = 0.0f;  // Patch.cpp:46-49
      uint32_t dcmp = 0, ccmp = 0, ncau = 0;
      float* tu = mesh_plane(v, mst, kMpTc);
      float* tv = mesh_plane(v, mst, kMpTc + 1);
      float keepX[kVB], keepY[kVB];  // texcoords of a one-sweep patch stay in registers until the box is known
      const bool one_sweep = nv <= 64u * kVB;
      // the first sweep does not wait for the record: its loads are clamped to the block, not to nv, and go out
      // together with the record's (lanes beyond nv are masked below)
      for (uint32_t base = 0; base == 0 || base < nv; base += 64u * kVB) {
        // ---- loads of the sweep: vertex positions and colours
        float px[kVB], py[kVB], pz[kVB], m0[kVB], m1[kVB], m2[kVB];
        const uint32_t lim = base == 0 ? mesh_cap_v(v, mst) : nv;
        // (a mesh that came out empty gave its block back, mesh_block_for: the unconditional first sweep then reads vertex 0
        // of block 1 -- any mapped address; every lane is masked by nv = 0 below)
        const uint32_t mld = mst != kBlkNone ? mst : 1u;
#pragma unroll
        for (int j = 0; j < kVB; ++j) {
          const uint32_t i = base + 64u * j + lane;
          const uint32_t ii = i < lim ? i : 0u;
          px[j] = mesh_plane(v, mld, kMpPos)[ii]; py[j] = mesh_plane(v, mld, kMpPos + 1)[ii];
          pz[j] = mesh_plane(v, mld, kMpPos + 2)[ii];
          m0[j] = mesh_plane(v, mld, kMpCol)[ii]; m1[j] = mesh_plane(v, mld, kMpCol + 1)[ii];
          m2[j] = mesh_plane(v, mld, kMpCol + 2)[ii];
        }
        if (base == 0) exchange_flags();  // (its loads went out with the record; the vertex loads are in flight now)
        stampw(3);
        // ---- projection (:52-66), then every image gather of the sweep in flight at once
        float cX[kVB], cY[kVB], dist[kVB];
        Taps tp[kVB];
        uint32_t q1[kVB], q2[kVB], q3[kVB];
        float d1[kVB], d2[kVB], d3[kVB];
        bool cau[kVB];
#pragma unroll
        for (int j = 0; j < kVB; ++j) {
          float vl[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) {  // T_g_l * (v, 1), accumulated column by column (:52-53)
            float s = kf.T[4 * r] * px[j];
            s = s + kf.T[4 * r + 1] * py[j];
            s = s + kf.T[4 * r + 2] * pz[j];
            s = s + kf.T[4 * r + 3] * 1.0f;
            vl[r] = s;
          }
          dist[j] = vl[2];
          const float x = vl[0] / vl[2], y = vl[1] / vl[2];
          float a = (float)((double)(x * cam.fxi + cam.cxi) + 0.5);  // :55-56
          float b = (float)((double)(y * cam.fyi + cam.cyi) + 0.5);
          cau[j] = (a < 0 || a >= Wf || b < 0 || b >= Hf);  // :58-62
          if (a < 0) a = 0;
          if (a >= Wf) a = Wf;
          if (b < 0) b = 0;
          if (b >= Hf) b = Hf;
          cX[j] = a; cY[j] = b;
          tp[j] = make_taps(W, H, a, b);
        }
#pragma unroll
        for (int j = 0; j < kVB; ++j) {
          q1[j] = load_px(kf.rgb, kf.stride, tp[j].i1);
          q2[j] = load_px(kf.rgb, kf.stride, tp[j].i2);
          q3[j] = load_px(kf.rgb, kf.stride, tp[j].i3);
          d1[j] = tp[j].i1 >= 0 ? kf.depth[tp[j].i1] : 0.0f;
          d2[j] = tp[j].i2 >= 0 ? kf.depth[tp[j].i2] : 0.0f;
          d3[j] = tp[j].i3 >= 0 ? kf.depth[tp[j].i3] : 0.0f;
        }
        stampw(4);
        // ---- arithmetic + stores
#pragma unroll
        for (int j = 0; j < kVB; ++j) {
          const uint32_t i = base + 64u * j + lane;
          const bool act = i < nv;
          bool cc = false, dc = false;
          if (act) {
            minX = minX < cX[j] ? minX : cX[j]; maxX = maxX > cX[j] ? maxX : cX[j];
            minY = minY < cY[j] ? minY : cY[j]; maxY = maxY > cY[j] ? maxY : cY[j];
            float tc[3];
            const TapW tw = tap_weights(W, H, cX[j], cY[j]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const float c1 = (float)((q1[j] >> (8 * k)) & 0xFFu), c2 = (float)((q2[j] >> (8 * k)) & 0xFFu),
                          c3 = (float)((q3[j] >> (8 * k)) & 0xFFu);
              tc[k] = blend(tw, c1, c2, c3) / 255.0f;
              mesh_plane(v, mst, kMpTcol + k)[i] = tc[k];
            }
            const float dpt = blend(tw, d1[j], d2[j], d3[j]);
            const float e0 = tc[0] - m0[j], e1 = tc[1] - m1[j], e2 = tc[2] - m2[j];
            const float s12 = e1 * e1 + e2 * e2;
            const float nrm = sqrtf(e0 * e0 + s12);
            cc = (double)nrm > 0.6;                    // :88
            dc = (double)fabsf(dist[j] - dpt) > 0.7;   // :89
            if (!one_sweep) { tu[i] = cX[j]; tv[i] = cY[j]; }
          }
          ncau += (uint32_t)__popcll(__ballot(act && cau[j]));
          ccmp += (uint32_t)__popcll(__ballot(cc));
          dcmp += (uint32_t)__popcll(__ballot(dc));
          if (one_sweep) { keepX[j] = cX[j]; keepY[j] = cY[j]; }
        }
      }
      // min / max are exact and order-free
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        float q = __shfl_xor(minX, o); minX = q < minX ? q : minX;
        q = __shfl_xor(maxX, o); maxX = q > maxX ? q : maxX;
        q = __shfl_xor(minY, o); minY = q < minY ? q : minY;
        q = __shfl_xor(maxY, o); maxY = q > maxY ? q : maxY;
      }
      const double nvd = (double)nv;
      const bool wrong = ((double)dcmp > 0.3 * nvd) || ((double)ccmp > 0.3 * nvd);  // :92-96
      float fbx = 0.0f, fby = 0.0f;
      bool have_box = false;
      if (maxX >= minX && maxY >= minY) {  // :98-99: cv::Rect(float..) truncates, & intersects
        const int ax = (int)(minX - 2.0f), ay = (int)(minY - 2.0f);
        const int aw = (int)(maxX - minX + 5.0f), ah = (int)(maxY - minY + 5.0f);
        bx = ax > 0 ? ax : 0; by = ay > 0 ? ay : 0;
        const int x2 = (ax + aw) < (W - 1) ? (ax + aw) : (W - 1);
        const int y2 = (ay + ah) < (H - 1) ? (ay + ah) : (H - 1);
        cols = x2 - bx; rows = y2 - by;
        if (cols <= 0 || rows <= 0) { bx = by = cols = rows = 0; }
        fbx = (float)bx; fby = (float)by;
        have_box = true;
      }
      if (one_sweep) {  // :100-102: texcoord -= box origin (or unshifted when there is no box)
#pragma unroll
        for (int j = 0; j < kVB; ++j) {
          const uint32_t i = 64u * j + lane;
          if (i < nv) { tu[i] = have_box ? keepX[j] - fbx : keepX[j]; tv[i] = have_box ? keepY[j] - fby : keepY[j]; }
        }
      } else if (have_box) {
        int l2 = lane;
        asm volatile("" : "+v"(l2));  // (or the lane's byte offset, hoisted out of the patch loop, is kept in private memory)
        for (uint32_t i = (uint32_t)l2; i < nv; i += 64) {  // a lane re-reads what it wrote
          tu[i] = tu[i] - fbx;
          tv[i] = tv[i] - fby;
        }
      }
      if (lane == 0) {
        rec->bbox[0] = bx; rec->bbox[1] = by; rec->bbox[2] = cols; rec->bbox[3] = rows;
        rec->pflags = kPfHasPatch | kPfHasImage | (ncau ? kPfCaution : 0u) | (wrong ? kPfWrong : 0u);
        rec->ratio[0] = 1.0f; rec->ratio[1] = 1.0f;
        if (FUSED) { rec->frameid = kf.kf_id; rec->kf_slot = -1; }  // Patch::clear + SetFrameid of the fused flow
      }
      have_image = true;
    } else {
      bx = R.bbox[0]; by = R.bbox[1]; cols = R.bbox[2]; rows = R.bbox[3];
      have_image = (R.pflags & kPfHasImage) != 0;
    }
    stampw(5);
    if (!BLIT) continue;
    if (TL && FUSED && kf_fused.pad[0] == 1) continue;  // triage: no blit
    // Patch::complete (Patch.cpp:191-196): vertices, simplified mesh, image, texcoords, frame id
    if (!(nv > 0 && (R.state & kMsSimplified) && have_image && kf.kf_id >= 0)) continue;
    if (cols <= 0 || rows <= 0) continue;  // empty ROI: nothing to copy
    // (opaque to the optimiser: float / double / reciprocal forms of the slot size hoisted out of the patch loop cost the
    // fused kernel 32 bytes of private memory per lane and a reload -- a dependent round trip -- at each of their uses)
    int PW = v.patch_w, PH = v.patch_h;
    asm volatile("" : "+s"(PW), "+s"(PH));
    float r0 = 1.0f, r1 = 1.0f;
    if (cols > PW) r0 = (float)PW / (float)cols;  // Atlas.cpp:77-80
    if (rows > PH) r1 = (float)PH / (float)rows;
    if (lane == 0) { rec->ratio[0] = r0; rec->ratio[1] = r1; }
    const unsigned long long tl = R.texloc;
    int aw_ = v.atlas_w;
    asm volatile("" : "+s"(aw_));  // (the reciprocal of a loop-invariant divisor is hoisted out of the patch loop -- into private memory)
    const unsigned long long ox = tl % (unsigned long long)aw_, oy = tl / (unsigned long long)aw_;
    const size_t astep = (size_t)v.atlas_w * 3;
    const int st = kf.stride;
    if (r0 < 1 || r1 < 1) {  // cv::resize(image, texroi, texroi.size()) into the FULL slot
      if (ox + PW > (unsigned long long)v.atlas_w || oy + PH > (unsigned long long)v.atlas_h) continue;
      const double scale_x = 1.0 / ((double)PW / cols), scale_y = 1.0 / ((double)PH / rows);
      for (int t = lane; t < PW * PH; t += 64) {
        const int dy = t / PW, dx = t - dy * PW;
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
        const int a0 = (short)cv_round_f((1.f - fx) * 2048.f), a1 = (short)cv_round_f(fx * 2048.f);
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        int sy0 = sy, sy1 = sy + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 > rows - 1 ? rows - 1 : sy0);
        sy1 = sy1 < 0 ? 0 : (sy1 > rows - 1 ? rows - 1 : sy1);
        const int b0 = (short)cv_round_f((1.f - fy) * 2048.f), b1 = (short)cv_round_f(fy * 2048.f);
        const int sx1 = sx + 1 < cols ? sx + 1 : sx;
        const uint32_t p00 = load_px(kf.rgb, st, (by + sy0) * W + bx + sx), p01 = load_px(kf.rgb, st, (by + sy0) * W + bx + sx1);
        const uint32_t p10 = load_px(kf.rgb, st, (by + sy1) * W + bx + sx), p11 = load_px(kf.rgb, st, (by + sy1) * W + bx + sx1);
        uint8_t* D = v.atlas + (oy + dy) * astep + (ox + dx) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int h0 = (int)((p00 >> (8 * k)) & 0xFFu) * a0 + (int)((p01 >> (8 * k)) & 0xFFu) * a1;
          const int h1 = (int)((p10 >> (8 * k)) & 0xFFu) * a0 + (int)((p11 >> (8 * k)) & 0xFFu) * a1;
          int val = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
          val = val < 0 ? 0 : (val > 255 ? 255 : val);
          D[k] = (uint8_t)val;
        }
      }
    } else {  // image.copyTo(texroi) at the slot origin
      if (ox + cols > (unsigned long long)v.atlas_w || oy + rows > (unsigned long long)v.atlas_h) continue;
      const int rowbytes = cols * 3;
      // the destination is written as aligned dwords; a destination dword gathers its (up to) four bytes
      // from the source row, whatever the source layout (3 or 4 bytes per pixel)
      uint8_t* d0 = v.atlas + oy * astep + ox * 3;
      const int mis = (int)((uintptr_t)d0 & 3u);  // the same for every row when the atlas row stride is a multiple of 4
      const bool rows_aligned = (astep & 3u) == 0;
      const int ndw = (rowbytes + mis + 3) >> 2;
      const float inv = 1.0f / (float)ndw;
      constexpr int kB = TF_PATCH_KB;  // 512 dwords in flight: a 24 x 18 slot has 18 x 18 = 324
      for (int t0 = 0; t0 < rows * ndw; t0 += 64 * kB) {
        uint32_t val[kB];
        int off[kB], rr[kB];
#pragma unroll
        for (int k = 0; k < kB; ++k) {
          const int t = t0 + 64 * k + lane;
          const int r = (int)(((float)t + 0.5f) * inv);  // t / ndw for t < 2^20 (never within 0.5 / ndw of an integer)
          const int j = t - r * ndw;
          rr[k] = t < rows * ndw ? r : -1;
          const int o = 4 * j - (rows_aligned ? mis : (int)((uintptr_t)(d0 + (size_t)r * astep) & 3u));
          off[k] = o;
          uint32_t x = 0;
          if (rr[k] >= 0) {
            const uint8_t* srow = kf.rgb + ((size_t)(by + r) * W + bx) * st;
            if (st == 4) {
              // RGBA source: the four bytes o .. o + 3 of the RGB row lie in at most two pixels, p0 and p0 + 1 --
              // two aligned dword loads instead of four byte loads
              const int p0 = (o > 0 ? o : 0) / 3;
              const uint32_t w0 = *reinterpret_cast<const uint32_t*>(srow + 4 * p0);
              const uint32_t w1 = (p0 + 1 < cols) ? *reinterpret_cast<const uint32_t*>(srow + 4 * (p0 + 1)) : 0u;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int bb = o + q;  // byte of the row
                if (bb >= 0 && bb < rowbytes) {
                  const int px = bb / 3, ch = bb - 3 * px;
                  x |= (((px == p0 ? w0 : w1) >> (8 * ch)) & 0xFFu) << (8 * q);
                }
              }
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int bb = o + q;  // byte of the row
                if (bb >= 0 && bb < rowbytes) {
                  const int px = bb / 3, ch = bb - 3 * px;
                  x |= (uint32_t)srow[px * st + ch] << (8 * q);
                }
              }
            }
          }
          val[k] = x;
        }
        stampw(6);
#pragma unroll
        for (int k = 0; k < kB; ++k) {
          if (rr[k] < 0) continue;
          uint8_t* drow = d0 + (size_t)rr[k] * astep;
          const int o = off[k];
          if (o >= 0 && o + 4 <= rowbytes) *reinterpret_cast<uint32_t*>(drow + o) = val[k];
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (o + q >= 0 && o + q < rowbytes) drow[o + q] = (uint8_t)(val[k] >> (8 * q));
          }
        }
      }
    }
  }
}

}  // namespace tf
