// tf_atlas.hip -- texture-atlas side of the path on device-resident meshes (ChunkManager::allMeshes
// lives in HBM as planar per-slot blocks, tf_device.h; Mesh::m_patch is part of the slot's MeshRec).
//
//   k_patch_assign     Atlas::AddPatch for an ordered chunk list (Structure/Atlas.cpp:43-64): the slot a
//                      new patch gets is loc_next at its turn = a prefix sum over the list
//   k_patch_rank       the same for the unordered per-frame dirty list of the fused flow:
//                      new patches take their slots in ascending chunk-id order (rank by comparison)
//   k_patch<P, B, F>   one WAVE per patch: Patch::CalculateTexCoords (Structure/Patch.cpp:40-108, lane =
//                      vertex, coalesced plane rows, bbox / vote by wave reductions) and / or
//                      Atlas::UpdateBuffer (Structure/Atlas.cpp:71-91, slot rows written as dwords)
//   k_cc_*             Chisel::CompensateColor reductions + transfer (Structure/Chisel.cpp:198-286)
//   k_draw             Chisel::DrawMeshes vertex / index packing (Structure/Chisel.cpp:288-355)
//
// cv::resize INTER_LINEAR for 8UC3 is restated in the same 11-bit fixed point (third-party arithmetic --
// parity unpinned); the copy branch is exact.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tf_devfn.h"
#include "tf_patch_body.h"
#include "tf_volume.h"

#pragma clang fp contract(off)

namespace tf {

// Patch::clear (Patch.cpp:177-189) + SetFrameid of a patch that is about to be re-projected
__device__ __forceinline__ void patch_begin(MeshRec* r, const KfDev& kf, int kf_slot) {
  r->frameid = kf.kf_id;
  r->kf_slot = kf_slot;
  r->pflags = kPfHasPatch;
  r->ratio[0] = 1.0f; r->ratio[1] = 1.0f;
}

// work-list entry -> pool slot of a chunk that has a mesh (Atlas::HasPatch == ChunkManager::HasMesh, Atlas.h:55)
__device__ __forceinline__ uint32_t mesh_slot_of(const VolumeDev& v, const int4 id) {
  const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
  uint32_t slot = kInvalidSlot;
  if (ent != kInvalidSlot && (v.hent[ent].alive & 1u)) slot = v.hent[ent].slot;
  if (slot != kInvalidSlot && !(v.mesh_rec[slot].state & kMsInMap)) slot = kInvalidSlot;
  return slot;
}
__global__ __launch_bounds__(256) void k_work_lookup(VolumeDev v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v.work_slot[i] = mesh_slot_of(v, v.work_ids[i]);
}

// ---------------------------------------------------------------------------------------
// Atlas::AddPatch over an ordered list (Chisel::GeneratePatches' loop, Chisel.cpp:156-181): single
// workgroup; entry j needs a slot iff its chunk has a mesh without one, the slot number is the number of
// such entries before it.  The first entry whose hand-out fails ends the call: it and everything behind it
// stays unprocessed (GeneratePatches returns -1 there).
// The pool slots of the entries come from k_work_lookup (a launch over the whole chip: the hash probes of one workgroup's
// threads, three entries each one after the other, were most of this kernel's 36 us).  Here thread t takes entries
// t, t + 1024, ...: a round of 1024 consecutive entries is one ballot scan, four rounds' loads are in flight together.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_patch_assign(VolumeDev v, uint32_t n) {
  __shared__ uint32_t wsum[16], ksum[16];
  __shared__ uint32_t first_fail;
  __shared__ unsigned long long smin, smax;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) { first_fail = 0xFFFFFFFFu; smin = ~0ull; smax = 0ull; }
  const unsigned long long base = v.actl->n_slots;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t run = 0;     // entries that needed a slot in the rounds before this one (block-uniform)
  uint32_t handed = 0;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n; i0 += 4096u) {
    uint32_t slot[4];
    unsigned long long tl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = i0 + 1024u * k + (uint32_t)t;
      slot[k] = i < n ? v.work_slot[i] : kInvalidSlot;  // !HasMesh -> continue (:157)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tl[k] = slot[k] != kInvalidSlot ? v.mesh_rec[slot[k]].texloc : 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + 1024u * k >= n) break;  // (block-uniform)
      const uint32_t i = i0 + 1024u * k + (uint32_t)t;
      const bool c = slot[k] != kInvalidSlot && tl[k] == kNoTexloc;
      const unsigned long long m = __ballot(c);
      __syncthreads();  // (wsum of the previous round has been read)
      if (lane == 0) wsum[w] = (uint32_t)__popcll(m);
      __syncthreads();
      uint32_t before = 0, tot = 0;
      for (int j = 0; j < 16; ++j) { if (j < w) before += wsum[j]; tot += wsum[j]; }
      if (c) {
        const uint32_t r = run + before + (uint32_t)__popcll(m & lt);
        unsigned long long ntl;
        if (slot_texloc(v, base + r, &ntl)) { v.mesh_rec[slot[k]].texloc = ntl; ++handed; }
        else atomicMin(&first_fail, i);
      }
      run += tot;
    }
  }
  __syncthreads();
  const uint32_t ff = first_fail;
  unsigned long long lmin = ~0ull, lmax = 0ull;
  uint32_t kept = 0;
  for (uint32_t i0 = 0; i0 < n; i0 += 4096u) {
    uint32_t slot[4];
    int kfs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = i0 + 1024u * k + (uint32_t)t;
      slot[k] = i < n ? v.work_slot[i] : kInvalidSlot;
      kfs[k] = i < n ? v.work_ids[i].w : 0;
    }
    unsigned long long tl[4];
    int32_t kid[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = i0 + 1024u * k + (uint32_t)t;
      const bool live = slot[k] != kInvalidSlot && i < ff;
      tl[k] = live ? v.mesh_rec[slot[k]].texloc : 0ull;
      kid[k] = live ? v.kf_tab[kfs[k]].kf_id : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = i0 + 1024u * k + (uint32_t)t;
      if (slot[k] == kInvalidSlot) continue;
      if (i >= ff) { v.work_slot[i] = kInvalidSlot; continue; }
      MeshRec* rec = &v.mesh_rec[slot[k]];
      ++kept;
      // Patch::clear (Patch.cpp:177-189) + SetFrameid: patch_begin with the keyframe's id fetched above
      rec->frameid = kid[k];
      rec->kf_slot = kfs[k];
      rec->pflags = kPfHasPatch;
      rec->ratio[0] = 1.0f; rec->ratio[1] = 1.0f;
      lmin = tl[k] < lmin ? tl[k] : lmin;
      lmax = tl[k] > lmax ? tl[k] : lmax;
    }
  }
  atomicMin(&smin, lmin);
  atomicMax(&smax, lmax);
  // every successful hand-out precedes the first failure (slot numbers grow with the list)
  uint32_t tot = handed, tk = kept;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { tot += __shfl_xor(tot, o); tk += __shfl_xor(tk, o); }
  __syncthreads();
  if (lane == 0) { wsum[w] = tot; ksum[w] = tk; }
  __syncthreads();
  if (t == 0) {
    uint32_t all = 0, allk = 0;
    for (int k = 0; k < 16; ++k) { all += wsum[k]; allk += ksum[k]; }
    v.actl->n_slots += all;
    v.actl->n_done = allk;
    v.actl->loc_min = smin;
    v.actl->loc_max = smax;
    v.actl->set[0].n_work = n;
    if (ff != 0xFFFFFFFFu) atomicOr(&v.vctl->status, kStAtlasFull);
  }
}

template <bool PROJECT, bool BLIT, bool FUSED>
__global__ __launch_bounds__(256) void k_patch(VolumeDev v, Cam cam, int par, KfDev kf_fused) {
  patch_body<PROJECT, BLIT, FUSED>(v, cam, par, kf_fused, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------
// Chisel::CompensateColor (Structure/Chisel.cpp:198-286).  The reductions over all vertices of a cluster and
// the per-vertex transfer run here; the two 3x3 eigen-decompositions per cluster are host work (a few
// hundred flops).  Reductions are fixed-shape trees (thread-strided partial sums in patch order, then an
// LDS tree), so results do not depend on timing.
// ---------------------------------------------------------------------------------------
struct CcPatch {
  uint32_t slot, nv;
  int32_t cluster;  // -1 = skipped (already adjusted)
  int32_t wrong;
};
// pass 0: sums of texcolor / mesh colour -> out[c][0..5], count -> out[c][6];
// pass 1: centred second moments (6 unique entries each) -> out[c][0..11] given mean[c][0..5]
template <int PASS>
__global__ __launch_bounds__(256) void k_cc_reduce(VolumeDev v, const CcPatch* __restrict__ pt, int64_t np,
                                                   const float* __restrict__ mean, float* __restrict__ out) {
  constexpr int NV = PASS == 0 ? 7 : 12;
  const int c = blockIdx.x;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
  float m[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (PASS == 1)
    for (int i = 0; i < 6; ++i) m[i] = mean[c * 6 + i];
  for (int64_t p = 0; p < np; ++p) {
    if (pt[p].cluster != c || pt[p].wrong) continue;
    const uint32_t slot = pt[p].slot;
    const uint32_t mst = v.mesh_rec[slot].block;  // (which block holds the mesh)
    for (uint32_t k = threadIdx.x; k < pt[p].nv; k += 256) {
      const float s0 = mesh_plane(v, mst, kMpTcol)[k], s1 = mesh_plane(v, mst, kMpTcol + 1)[k],
                  s2 = mesh_plane(v, mst, kMpTcol + 2)[k];
      const float t0 = mesh_plane(v, mst, kMpCol)[k], t1 = mesh_plane(v, mst, kMpCol + 1)[k],
                  t2 = mesh_plane(v, mst, kMpCol + 2)[k];
      if constexpr (PASS == 0) {
        acc[0] += s0; acc[1] += s1; acc[2] += s2;
        acc[3] += t0; acc[4] += t1; acc[5] += t2;
        acc[6] += 1.0f;
      } else {
        const float a = s0 - m[0], b = s1 - m[1], d = s2 - m[2];
        const float e = t0 - m[3], f = t1 - m[4], g = t2 - m[5];
        acc[0] += a * a; acc[1] += a * b; acc[2] += a * d; acc[3] += b * b; acc[4] += b * d; acc[5] += d * d;
        acc[6] += e * e; acc[7] += e * f; acc[8] += e * g; acc[9] += f * f; acc[10] += f * g; acc[11] += g * g;
      }
    }
  }
  __shared__ float red[NV][256];
#pragma unroll
  for (int i = 0; i < NV; ++i) red[i][threadIdx.x] = acc[i];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int i = 0; i < NV; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < NV) out[c * 12 + threadIdx.x] = red[threadIdx.x][0];
}
// labs[k] = T (texcolor[k] - mean_src) + mean_tar (Chisel.cpp:274); has_adjusted = true (:280)
__global__ __launch_bounds__(256) void k_cc_apply(VolumeDev v, const CcPatch* __restrict__ pt,
                                                  const float* __restrict__ xf /* per cluster: T[9], mean_src[3], mean_tar[3], ok */) {
  const CcPatch P = pt[blockIdx.x];
  if (P.cluster < 0) return;
  const float* X = xf + (size_t)P.cluster * 16;
  if (X[15] == 0.0f) return;  // nothing was learnt for this cluster (:242): has_adjusted stays false
  if (threadIdx.x == 0) v.mesh_rec[P.slot].pflags |= kPfAdjusted;
  if (P.wrong) return;  // labs cleared (:277-279)
  const uint32_t mst = v.mesh_rec[P.slot].block;
  for (uint32_t k = threadIdx.x; k < P.nv; k += 256) {
    const float d0 = mesh_plane(v, mst, kMpTcol)[k] - X[9], d1 = mesh_plane(v, mst, kMpTcol + 1)[k] - X[10],
                d2 = mesh_plane(v, mst, kMpTcol + 2)[k] - X[11];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float a = X[3 * i] * d0;
      a = a + X[3 * i + 1] * d1;
      a = a + X[3 * i + 2] * d2;
      mesh_plane(v, mst, kMpLabs + i)[k] = a + X[12 + i];
    }
  }
}

// Every mesh that has a patch -- the host sorts by id (the reference iterates allMeshes, an unordered_map:
// the harness order is ascending chunk id).
struct PatchRow {
  int32_t id[3];
  uint32_t slot;
  uint32_t nv, nt;
  int32_t frameid;
  uint32_t state, pflags;
  uint32_t pad;
};
__global__ __launch_bounds__(256) void k_list_patches(VolumeDev v, PatchRow* out, uint32_t cap) {
  const uint32_t nent = v.hmask + 1u;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nent; i += gridDim.x * 256) {
    const HEntry h = v.hent[i];
    if (h.key == kEmptyKey || !(h.alive & 1u) || h.slot == kInvalidSlot) continue;
    const MeshRec m = v.mesh_rec[h.slot];
    if (!(m.state & kMsInMap) || !(m.pflags & kPfHasPatch)) continue;
    const uint32_t p = atomicAdd(&v.vctl->n_tmp, 1u);
    if (p >= cap) continue;
    const int4 id = unpack_id(h.key);
    PatchRow r;
    r.id[0] = id.x; r.id[1] = id.y; r.id[2] = id.z;
    r.slot = h.slot; r.nv = m.nv; r.nt = m.nt; r.frameid = m.frameid; r.state = m.state; r.pflags = m.pflags; r.pad = 0;
    out[p] = r;
  }
}

// ---------------------------------------------------------------------------------------
// Chisel::DrawMeshes (Structure/Chisel.cpp:288-355): interleaved vertex stream + rebased indices.
// One workgroup per complete() patch; every output element is written once, 48 B per vertex.
// ---------------------------------------------------------------------------------------
struct DrawPatch {
  uint32_t slot, nv, nt, flags;  // flags: bit1 wrong_mapping, bit2 labs valid
  unsigned long long vout, iout;  // output positions (running counts over the patches before this one)
};
__global__ __launch_bounds__(256) void k_draw(VolumeDev v, const DrawPatch* __restrict__ pt, float* __restrict__ out_v,
                                              uint32_t* __restrict__ out_i) {
  const DrawPatch P = pt[blockIdx.x];
  const MeshRec rec = v.mesh_rec[P.slot];
  const float ox = (float)(rec.texloc % (unsigned long long)v.atlas_w);  // Atlas::GetTexLoc (Atlas.cpp:66-69)
  const float oy = (float)(rec.texloc / (unsigned long long)v.atlas_w);
  const float rx = rec.ratio[0], ry = rec.ratio[1];
  const float aw = (float)v.atlas_w, ah = (float)v.atlas_h;
  for (uint32_t j = threadIdx.x; j < P.nt; j += 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) out_i[P.iout + 3 * (size_t)j + a] = (uint32_t)tri_plane(v, rec.block, a)[j] + (uint32_t)P.vout;
  for (uint32_t k = threadIdx.x; k < P.nv; k += 256) {
    float* o = out_v + 12 * (P.vout + k);
    float tx = mesh_plane(v, rec.block, kMpTc)[k], ty = mesh_plane(v, rec.block, kMpTc + 1)[k];
    if (rx < 1.0f) tx = tx * rx;
    if (ry < 1.0f) ty = ty * ry;
    tx = tx + ox;
    ty = ty + oy;
    const float c0 = mesh_plane(v, rec.block, kMpCol)[k], c1 = mesh_plane(v, rec.block, kMpCol + 1)[k],
                c2 = mesh_plane(v, rec.block, kMpCol + 2)[k];
    int rgb = (int)(c0 * 255.0f);
    rgb = (rgb << 8) + (int)(c1 * 255.0f);
    rgb = (rgb << 8) + (int)(c2 * 255.0f);
    float adj = 0.0f;
    if (P.flags & 4u) {
      const float a0 = mesh_plane(v, rec.block, kMpLabs)[k] - mesh_plane(v, rec.block, kMpTcol)[k],
                  a1 = mesh_plane(v, rec.block, kMpLabs + 1)[k] - mesh_plane(v, rec.block, kMpTcol + 1)[k],
                  a2 = mesh_plane(v, rec.block, kMpLabs + 2)[k] - mesh_plane(v, rec.block, kMpTcol + 2)[k];
      int ad = (int)(a0 * 255.0f) + 255;
      ad = (ad << 9) + (int)(a1 * 255.0f) + 255;
      ad = (ad << 9) + (int)(a2 * 255.0f) + 255;
      adj = (float)ad;
    }
    const float4 q0 = make_float4(mesh_plane(v, rec.block, kMpPos)[k], mesh_plane(v, rec.block, kMpPos + 1)[k],
                                  mesh_plane(v, rec.block, kMpPos + 2)[k], 50.0f);
    const float4 q1 = make_float4((float)rgb, adj, tx / aw, ty / ah);
    const float4 q2 = make_float4(mesh_plane(v, rec.block, kMpNrm)[k], mesh_plane(v, rec.block, kMpNrm + 1)[k],
                                  mesh_plane(v, rec.block, kMpNrm + 2)[k], (P.flags & 2u) ? 1.0f : 0.0f);
    reinterpret_cast<float4*>(o)[0] = q0;
    reinterpret_cast<float4*>(o)[1] = q1;
    reinterpret_cast<float4*>(o)[2] = q2;
  }
}

// Patch mirrors of listed chunks: per patch record + texcoord / texcolor / labs in the reference's layouts
struct PatchHost {
  unsigned long long texloc;
  int32_t frameid;
  uint32_t pflags;
  int32_t bbox[4];
  float ratio[2];
  uint32_t nv, found;
};
__global__ __launch_bounds__(256) void k_patch_gather(VolumeDev v, const int4* __restrict__ ids, uint32_t n,
                                                      const long long* __restrict__ voff, PatchHost* __restrict__ ph,
                                                      float* texcoord, float* texcolor, float* labs) {
  const uint32_t c = blockIdx.x;
  if (c >= n) return;
  PatchHost h;
  memset(&h, 0, sizeof(h));
  h.texloc = kNoTexloc; h.frameid = -1;
  const uint32_t slot = mesh_slot_of(v, ids[c]);
  if (slot != kInvalidSlot) {
    const MeshRec m = v.mesh_rec[slot];
    h.texloc = m.texloc; h.frameid = m.frameid; h.pflags = m.pflags;
    for (int k = 0; k < 4; ++k) h.bbox[k] = m.bbox[k];
    h.ratio[0] = m.ratio[0]; h.ratio[1] = m.ratio[1];
    h.nv = m.nv; h.found = 1;
    if (voff) {
      const long long v0 = voff[c];
      const uint32_t nv = (uint32_t)(voff[c + 1] - v0) < m.nv ? (uint32_t)(voff[c + 1] - v0) : m.nv;
      for (uint32_t i = threadIdx.x; i < nv; i += 256) {
        if (texcoord) { texcoord[2 * (v0 + i)] = mesh_plane(v, m.block, kMpTc)[i]; texcoord[2 * (v0 + i) + 1] = mesh_plane(v, m.block, kMpTc + 1)[i]; }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (texcolor) texcolor[3 * (v0 + i) + a] = mesh_plane(v, m.block, kMpTcol + a)[i];
          if (labs) labs[3 * (v0 + i) + a] = mesh_plane(v, m.block, kMpLabs + a)[i];
        }
      }
    }
  }
  if (threadIdx.x == 0) ph[c] = h;
}

// allMeshes[id] = a host-built mesh (Mesh::vertices / normals / colors / indices): creates the chunk when
// it does not exist yet
__global__ __launch_bounds__(256) void k_mesh_scatter(VolumeDev v, const int4* __restrict__ ids, uint32_t n,
                                                      const long long* __restrict__ voff, const long long* __restrict__ ioff,
                                                      const float* __restrict__ verts, const float* __restrict__ normals,
                                                      const float* __restrict__ colors, const uint32_t* __restrict__ indices,
                                                      uint32_t epoch) {
  __shared__ uint32_t sslot;
  const uint32_t c = blockIdx.x;
  if (c >= n) return;
  const int4 id = ids[c];
  if (threadIdx.x == 0) {
    bool is_new;
    uint32_t ent;
    sslot = chunk_acquire(v, id, &is_new, &ent);
  }
  __syncthreads();
  const uint32_t slot = sslot;
  if (slot == kInvalidSlot) return;
  const long long v0 = voff[c], i0 = ioff[c];
  const uint32_t nv = (uint32_t)(voff[c + 1] - v0), nt = (uint32_t)(ioff[c + 1] - i0) / 3u;
  // the mesh goes into the block the chunk owns when it fits there, else into a block handed out now (small pool, or the
  // large one for a mesh beyond CV / CT)
  __shared__ uint32_t s_blk;
  if (threadIdx.x == 0) {
    const uint32_t was = v.mesh_rec[slot].block;
    s_blk = mesh_block_for(v, was, nv, nt, blk_pressure(v));
    if (s_blk != kBlkFail) mesh_block_settle(v, was, s_blk);
  }
  __syncthreads();
  const uint32_t mst = s_blk;
  if (mst == kBlkFail) {
    if (threadIdx.x == 0) atomicOr(&v.vctl->status, kStMeshFull);
    return;
  }
  for (uint32_t i = threadIdx.x; i < nv; i += 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mesh_plane(v, mst, kMpPos + a)[i] = verts[3 * (v0 + i) + a];
      mesh_plane(v, mst, kMpNrm + a)[i] = normals[3 * (v0 + i) + a];
      mesh_plane(v, mst, kMpCol + a)[i] = colors[3 * (v0 + i) + a];
    }
  for (uint32_t i = threadIdx.x; i < nt; i += 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) tri_plane(v, mst, a)[i] = (uint16_t)indices[i0 + 3 * (size_t)i + a];
  if (threadIdx.x == 0) {
    MeshRec* r = &v.mesh_rec[slot];
    r->nv = (uint16_t)nv; r->nt = (uint16_t)nt; r->epoch = epoch; r->block = mst;
    r->state = kMsInMap;  // Mesh::Clear: adj = false, simplified = false
  }
}

// measurement aid: what the fused flow did for the work list of counter set `par`
__global__ __launch_bounds__(256) void k_texture_stats(VolumeDev v, int par, unsigned long long* out) {
  const uint32_t n_flat = v.actl->set[par].n_work;
  const uint32_t rows = mesh_shard_rows_dev(v.max_chunks);
  unsigned long long a[6] = {0, 0, 0, 0, 0, 0};
  // the frame's dirty set: the flat work list, then the shard lists K-A filled (slot i of shard s = index n_flat + s * rows + i)
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_flat + kMeshShards * rows; i += gridDim.x * 256) {
    uint32_t slot;
    if (i < n_flat) {
      slot = v.work_slot[i];
    } else {
      const uint32_t sh = (i - n_flat) / rows, k = (i - n_flat) - sh * rows;
      if (k >= v.wl_cnt[((par & 1) * kMeshShards + sh) * 16]) continue;
      const size_t at = ((size_t)(par & 1) * kMeshShards + sh) * rows + k;
      slot = v.wl_slot[at];
      const int4 id = v.wl_ids[at];
      if (id.w > 0 && !(v.hent[(uint32_t)id.w - 1u].alive & 1u)) continue;  // (parked behind the claim: not a chunk)
    }
    a[0] += 1;
    if (slot == kInvalidSlot) continue;
    const MeshRec m = v.mesh_rec[slot];
    if (!(m.state & kMsInMap)) continue;
    a[1] += 1; a[2] += m.nv; a[3] += m.nt;
    if (m.pflags & kPfHasImage) { a[4] += (unsigned long long)(m.bbox[2] > 0 ? m.bbox[2] : 0) * (unsigned long long)(m.bbox[3] > 0 ? m.bbox[3] : 0); a[5] += 1; }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) a[k] += __shfl_xor(a[k], o);
    if ((threadIdx.x & 63) == 0 && a[k]) atomicAdd(&out[k], a[k]);
  }
}
void launch_texture_stats(const VolumeDev& v, int par, unsigned long long* out6, hipStream_t s) {
  hipLaunchKernelGGL(k_texture_stats, dim3(64), dim3(256), 0, s, v, par, out6);
}

// ---- host side ------------------------------------------------------------------------
int atlas_init(tf_volume* v) {
  AtlasState& a = v->atlas;
  VolumeDev& d = v->dev;
  a.aw = v->cfg.atlas_w;
  a.ah = v->cfg.atlas_h;
  a.pw = (uint64_t)floor((double)(4800.0f * v->res));  // Atlas::SetResolution, Atlas.h:62-65
  a.ph = (uint64_t)floor((double)(3600.0f * v->res));
  if (a.pw < 1 || a.ph < 1) { set_error("voxel resolution too small for an atlas patch"); return TF_ERR_INVALID; }
  const size_t bytes = (size_t)a.aw * a.ah * 3;
  hipError_t e = hipMalloc((void**)&a.buf, bytes);  // Atlas.cpp:34: 13824 x 13824 x RGB8 = 573 MB
  if (e != hipSuccess) { set_error("atlas hipMalloc failed"); return TF_ERR_HIP; }
  a.kf_cap = v->cfg.max_keyframes;
  TF_HIP(hipMalloc((void**)&a.d_kf, sizeof(KfDev) * (size_t)a.kf_cap));
  TF_HIP(hipMalloc((void**)&a.d_actl, sizeof(AtlasCtl)));
  // two work lists: the fused flow builds the list of frame f + 1 while the patches of frame f still read theirs
  TF_HIP(hipMalloc((void**)&a.d_work_ids, sizeof(int4) * (size_t)d.max_chunks * 2));
  TF_HIP(hipMalloc((void**)&a.d_work_slot, sizeof(uint32_t) * (size_t)d.max_chunks * 2));
  TF_HIP(hipMalloc((void**)&a.d_patch_list, sizeof(int4) * (size_t)2 * kMeshShards * mesh_shard_rows(d.max_chunks)));
  TF_HIP(hipMalloc((void**)&a.d_wl_ids, sizeof(int4) * (size_t)2 * kMeshShards * mesh_shard_rows(d.max_chunks)));
  TF_HIP(hipMalloc((void**)&a.d_wl_slot, sizeof(uint32_t) * (size_t)2 * kMeshShards * mesh_shard_rows(d.max_chunks)));
  TF_HIP(hipMalloc((void**)&a.d_wl_cnt, sizeof(uint32_t) * 2 * kMeshShards * 16));
  TF_HIP(hipMemset(a.d_wl_cnt, 0, sizeof(uint32_t) * 2 * kMeshShards * 16));
  TF_HIP(hipMalloc((void**)&a.d_patch_cnt, sizeof(uint32_t) * 2 * kMeshShards * 16));
  TF_HIP(hipMemset(a.d_patch_cnt, 0, sizeof(uint32_t) * 2 * kMeshShards * 16));
  TF_HIP(hipHostMalloc((void**)&a.h_dirty_len, 64, hipHostMallocDefault));
  *a.h_dirty_len = 0u;
  TF_HIP(hipMalloc((void**)&a.d_cand, sizeof(unsigned long long) * (size_t)d.max_chunks));
  KfDev blank;
  memset(&blank, 0, sizeof(blank));
  blank.kf_id = -1; blank.stride = 3;
  a.h_kf.assign((size_t)a.kf_cap, blank);
  a.kf_used.assign((size_t)a.kf_cap, 0);
  TF_HIP(hipMemcpy(a.d_kf, a.h_kf.data(), sizeof(KfDev) * (size_t)a.kf_cap, hipMemcpyHostToDevice));
  d.atlas = a.buf; d.atlas_w = a.aw; d.atlas_h = a.ah; d.patch_w = (int32_t)a.pw; d.patch_h = (int32_t)a.ph;
  d.actl = a.d_actl; d.kf_tab = a.d_kf; d.work_ids = a.d_work_ids; d.work_slot = a.d_work_slot; d.patch_list = a.d_patch_list; d.patch_cnt = a.d_patch_cnt; d.cand = a.d_cand;
  d.wl_ids = a.d_wl_ids; d.wl_slot = a.d_wl_slot; d.wl_cnt = a.d_wl_cnt;
  return atlas_reset(v);
}

void atlas_destroy(tf_volume* v) {
  AtlasState& a = v->atlas;
  for (auto& kv : a.keyframes) {
    KeyframeSlot& ks = kv.second;
    if (ks.owned) { hipFree(ks.rgb); hipFree(ks.depth); }
  }
  a.keyframes.clear();
  if (a.buf) hipFree(a.buf);
  if (a.d_kf) hipFree(a.d_kf);
  if (a.d_actl) hipFree(a.d_actl);
  if (a.d_work_ids) hipFree(a.d_work_ids);
  if (a.d_work_slot) hipFree(a.d_work_slot);
  if (a.d_wl_ids) hipFree(a.d_wl_ids);
  if (a.d_wl_slot) hipFree(a.d_wl_slot);
  if (a.d_wl_cnt) hipFree(a.d_wl_cnt);
  a.d_wl_ids = nullptr; a.d_wl_slot = nullptr; a.d_wl_cnt = nullptr;
  if (a.d_patch_list) hipFree(a.d_patch_list);
  if (a.d_patch_cnt) hipFree(a.d_patch_cnt);
  if (a.h_dirty_len) hipHostFree(a.h_dirty_len);
  a.h_dirty_len = nullptr;
  if (a.d_cand) hipFree(a.d_cand);
  if (a.d_stage) hipFree(a.d_stage);
  if (a.h_stage) hipHostFree(a.h_stage);
  a.buf = nullptr; a.d_stage = nullptr; a.h_stage = nullptr; a.d_kf = nullptr; a.d_actl = nullptr;
  a.d_work_ids = nullptr; a.d_work_slot = nullptr; a.d_patch_list = nullptr; a.d_patch_cnt = nullptr; a.d_cand = nullptr;
}

int atlas_reset(tf_volume* v) {
  AtlasState& a = v->atlas;
  {
    AtlasWriteScope aw(v, -1);
    TF_HIP(hipMemsetAsync(a.buf, 0, (size_t)a.aw * a.ah * 3, v->stream));  // Atlas.cpp:35-36
  }
  AtlasCtl c;
  memset(&c, 0, sizeof(c));
  c.loc_min = ~0ull;
  TF_HIP(hipMemcpyAsync(a.d_actl, &c, sizeof(c), hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipMemsetAsync(a.d_patch_cnt, 0, sizeof(uint32_t) * 2 * kMeshShards * 16, v->stream));
  TF_HIP(hipMemsetAsync(a.d_wl_cnt, 0, sizeof(uint32_t) * 2 * kMeshShards * 16, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  a.fused_par = 0;
  a.fused_armed = true;
  a.pend_patch.on = false;
  return TF_OK;
}

static int atlas_stage(tf_volume* v, size_t bytes) {
  AtlasState& a = v->atlas;
  if (bytes > a.d_stage_bytes) {
    TF_HIP(hipStreamSynchronize(v->stream));
    if (a.d_stage) hipFree(a.d_stage);
    if (a.h_stage) hipHostFree(a.h_stage);
    a.d_stage = nullptr; a.h_stage = nullptr;
    size_t want = 1;
    while (want < bytes) want <<= 1;
    TF_HIP(hipMalloc(&a.d_stage, want));
    TF_HIP(hipHostMalloc(&a.h_stage, want, hipHostMallocDefault));
    a.d_stage_bytes = a.h_stage_bytes = want;
  }
  return TF_OK;
}

// keyframe table: host mirror -> device (tiny)
// (the record travels as a kernel ARGUMENT: copied when the launch is enqueued, so the host copy may change right away and
// nothing waits -- a hipMemcpyAsync out of the pageable table needed a stream synchronisation per keyframe call)
__global__ void k_kf_set(KfDev* dst, KfDev val) {
  if (threadIdx.x == 0) *dst = val;
}
int kf_push(tf_volume* v, int slot) {
  AtlasState& a = v->atlas;
  hipLaunchKernelGGL(k_kf_set, dim3(1), dim3(64), 0, v->stream, a.d_kf + slot, a.h_kf[(size_t)slot]);
  TF_HIP(hipGetLastError());
  return TF_OK;
}
static int kf_slot_for(tf_volume* v, int32_t kf_id, bool create, int* out) {
  AtlasState& a = v->atlas;
  auto it = a.keyframes.find(kf_id);
  if (it != a.keyframes.end()) { *out = it->second.slot; return TF_OK; }
  if (!create) {
    set_error("keyframe " + std::to_string(kf_id) + " is not cached (tf_keyframe_cache)");
    return TF_ERR_INVALID;
  }
  int slot = -1;
  for (int s = 0; s < v->cfg.max_keyframes; ++s)
    if (!a.kf_used[(size_t)s]) { slot = s; break; }
  if (slot < 0) { set_error("keyframe cache full (tf_config.max_keyframes)"); return TF_ERR_CAPACITY; }
  KeyframeSlot ks;
  ks.slot = slot;
  a.keyframes[kf_id] = ks;
  a.kf_used[(size_t)slot] = 1;
  KfDev& k = a.h_kf[(size_t)slot];
  memset(&k, 0, sizeof(k));
  k.kf_id = kf_id; k.stride = 3;
  for (int i = 0; i < 4; ++i) k.T[5 * i] = 1.0f;
  *out = slot;
  return TF_OK;
}

// symmetric 3x3 eigen-decomposition, cyclic Jacobi in double: A = V diag(w) V^T
static void sym3_eig(const float A[9], double w[3], double V[9]) {
  double a[9];
  for (int i = 0; i < 9; i++) { a[i] = (double)A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    if (a[1] * a[1] + a[2] * a[2] + a[5] * a[5] < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        const double apq = a[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; k++) {
          const double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = cs * akp - sn * akq;
          a[3 * k + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; k++) {
          const double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = cs * apk - sn * aqk;
          a[3 * q + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = cs * vkp - sn * vkq;
          V[3 * k + q] = sn * vkp + cs * vkq;
        }
      }
  }
  for (int i = 0; i < 3; i++) w[i] = a[4 * i];
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// Chisel.cpp:247-266: T = U Ds' Um Dm Um^T Ds' U^T with media = Ds U^T Ct U Ds
static void color_transfer(const float cov_src[9], const float cov_tar[9], float T[9]) {
  double ws[3], U[9], Ut[9], ct[9], D[9] = {0}, M1[9], M2[9], media[9];
  sym3_eig(cov_src, ws, U);
  for (int i = 0; i < 3; i++) {
    D[4 * i] = (double)(float)sqrt(ws[i] > 0.0 ? ws[i] : 0.0);
    for (int j = 0; j < 3; j++) { Ut[3 * i + j] = U[3 * j + i]; ct[3 * i + j] = (double)cov_tar[3 * i + j]; }
  }
  mat3_mul(D, Ut, M1); mat3_mul(M1, ct, M2); mat3_mul(M2, U, M1); mat3_mul(M1, D, media);
  float mediaf[9];
  for (int i = 0; i < 9; i++) mediaf[i] = (float)media[i];
  for (int i = 0; i < 3; i++)
    for (int j = i + 1; j < 3; j++) mediaf[3 * j + i] = mediaf[3 * i + j];
  double wm[3], Um[9], Umt[9], Dm[9] = {0}, Di[9] = {0};
  sym3_eig(mediaf, wm, Um);
  for (int i = 0; i < 3; i++) {
    Dm[4 * i] = (double)(float)sqrt(wm[i] > 0.0 ? wm[i] : 0.0);
    Di[4 * i] = (double)(float)(1.0 / ((double)(float)D[4 * i] + 1e-2));  // 1 / (diag + 1e-2), double literal (:260-262)
    for (int j = 0; j < 3; j++) Umt[3 * i + j] = Um[3 * j + i];
  }
  double A1[9], A2[9];
  mat3_mul(U, Di, A1); mat3_mul(A1, Um, A2); mat3_mul(A2, Dm, A1); mat3_mul(A1, Umt, A2);
  mat3_mul(A2, Di, A1); mat3_mul(A1, Ut, A2);
  for (int i = 0; i < 9; i++) T[i] = (float)A2[i];
}

static bool row_less(const PatchRow& a, const PatchRow& b) {
  for (int k = 0; k < 3; ++k)
    if (a.id[k] != b.id[k]) return a.id[k] < b.id[k];
  return false;
}

// every mesh with a patch, ascending chunk id (host copy)
static int list_patches(tf_volume* v, std::vector<PatchRow>* rows) {
  const size_t cap = (size_t)v->dev.max_chunks;
  int rc = atlas_stage(v, cap * sizeof(PatchRow));
  if (rc) return rc;
  AtlasState& a = v->atlas;
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
  hipLaunchKernelGGL(k_list_patches, dim3(1024), dim3(256), 0, v->stream, v->dev,
                     reinterpret_cast<PatchRow*>(a.d_stage), (uint32_t)cap);
  TF_HIP(hipGetLastError());
  uint32_t n = 0;
  TF_HIP(hipMemcpyAsync(&n, &v->dev.vctl->n_tmp, 4, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  rows->resize(n);
  if (n) {
    TF_HIP(hipMemcpyAsync(a.h_stage, a.d_stage, (size_t)n * sizeof(PatchRow), hipMemcpyDeviceToHost, v->stream));
    TF_HIP(hipStreamSynchronize(v->stream));
    memcpy(rows->data(), a.h_stage, (size_t)n * sizeof(PatchRow));
    std::sort(rows->begin(), rows->end(), row_less);
  }
  return TF_OK;
}

// ids (host) -> device work list (w = keyframe-table entry per entry, or 0); the entry count goes to
// counter set 0
static int upload_work(tf_volume* v, const int32_t* ids, const int* kfslot, int64_t n) {
  AtlasState& a = v->atlas;
  if (n > (int64_t)v->dev.max_chunks) { set_error("chunk list longer than tf_config.max_chunks"); return TF_ERR_CAPACITY; }
  int rc = atlas_stage(v, (size_t)n * 16 + 16);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  int32_t* h = reinterpret_cast<int32_t*>(a.h_stage);
  for (int64_t i = 0; i < n; ++i) {
    h[4 * i] = ids[3 * i]; h[4 * i + 1] = ids[3 * i + 1]; h[4 * i + 2] = ids[3 * i + 2];
    h[4 * i + 3] = kfslot ? kfslot[i] : 0;
  }
  h[4 * n] = (int32_t)n;
  TF_HIP(hipMemcpyAsync(a.d_work_ids, h, (size_t)n * 16, hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipMemcpyAsync(&a.d_actl->set[0].n_work, h + 4 * n, 4, hipMemcpyHostToDevice, v->stream));
  return TF_OK;
}

// fused per-frame flow: AddPatch in ascending id order, CalculateTexCoords + UpdateBuffer for the frame's dirty set
void launch_patch_fused(tf_volume* v, const VolumeDev& d, int par, const KfDev& kf, hipStream_t s) {
  prof_begin(v, TF_PROF_PATCH_PROJECT, s);
  hipLaunchKernelGGL((k_patch<true, true, true>), dim3(1024), dim3(256), 0, s, d, v->cam, par, kf);
  prof_end(v, s);
}

}  // namespace tf

using namespace tf;

extern "C" {

int tf_keyframe_cache(tf_volume* v, int32_t kf_id, const uint8_t* rgb, const float* depth) {
  if (!v || !rgb || !depth) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  const size_t npix = (size_t)v->cam.W * v->cam.H;
  int slot = -1;
  int rc = kf_slot_for(v, kf_id, true, &slot);
  if (rc) return rc;
  KeyframeSlot& ks = a.keyframes[kf_id];
  if (!ks.owned) {
    ks.rgb = nullptr; ks.depth = nullptr;
    TF_HIP(hipMalloc((void**)&ks.rgb, npix * 3));
    TF_HIP(hipMalloc((void**)&ks.depth, npix * 4));
    ks.owned = true;
  }
  rc = atlas_stage(v, npix * 7);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  memcpy(hs, rgb, npix * 3);
  memcpy(hs + npix * 3, depth, npix * 4);
  TF_HIP(hipMemcpyAsync(ks.rgb, hs, npix * 3, hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipMemcpyAsync(ks.depth, hs + npix * 3, npix * 4, hipMemcpyHostToDevice, v->stream));
  a.h_kf[(size_t)slot].rgb = ks.rgb;
  a.h_kf[(size_t)slot].depth = ks.depth;
  a.h_kf[(size_t)slot].stride = 3;
  return kf_push(v, slot);
}

int tf_keyframe_cache_device(tf_volume* v, int32_t kf_id, const uint8_t* d_rgb, int32_t rgb_pixel_stride,
                             const float* d_depth) {
  if (!v || !d_rgb || !d_depth) { set_error("null argument"); return TF_ERR_INVALID; }
  if (rgb_pixel_stride != 3 && rgb_pixel_stride != 4) { set_error("rgb_pixel_stride must be 3 or 4"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  int slot = -1;
  int rc = kf_slot_for(v, kf_id, true, &slot);
  if (rc) return rc;
  KeyframeSlot& ks = a.keyframes[kf_id];
  if (ks.owned) { TF_HIP(hipStreamSynchronize(v->stream)); hipFree(ks.rgb); hipFree(ks.depth); ks.owned = false; }
  ks.rgb = const_cast<uint8_t*>(d_rgb);
  ks.depth = const_cast<float*>(d_depth);
  a.h_kf[(size_t)slot].rgb = d_rgb;
  a.h_kf[(size_t)slot].depth = d_depth;
  a.h_kf[(size_t)slot].stride = rgb_pixel_stride;
  return kf_push(v, slot);
}

int tf_keyframe_set_pose(tf_volume* v, int32_t kf_id, const float pose_inv16[16]) {
  if (!v || !pose_inv16) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int slot = -1;
  int rc = kf_slot_for(v, kf_id, false, &slot);
  if (rc) return rc;
  memcpy(v->atlas.h_kf[(size_t)slot].T, pose_inv16, 64);
  return kf_push(v, slot);
}

int tf_keyframe_release(tf_volume* v, int32_t kf_id) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  auto it = a.keyframes.find(kf_id);
  if (it == a.keyframes.end()) return TF_OK;
  TF_HIP(hipStreamSynchronize(v->stream));
  if (it->second.owned) { hipFree(it->second.rgb); hipFree(it->second.depth); }
  const int slot = it->second.slot;
  a.keyframes.erase(it);
  a.kf_used[(size_t)slot] = 0;
  KfDev& k = a.h_kf[(size_t)slot];
  memset(&k, 0, sizeof(k));
  k.kf_id = -1; k.stride = 3;  // patches that still view this keyframe are no longer complete()
  return kf_push(v, slot);
}

int tf_atlas_patch_size(tf_volume* v, int32_t* pw, int32_t* ph) {
  if (!v || !pw || !ph) { set_error("null argument"); return TF_ERR_INVALID; }
  *pw = (int32_t)v->atlas.pw;
  *ph = (int32_t)v->atlas.ph;
  return TF_OK;
}

int tf_atlas_size(tf_volume* v, int32_t* aw, int32_t* ah) {
  if (!v || !aw || !ah) { set_error("null argument"); return TF_ERR_INVALID; }
  *aw = v->atlas.aw;
  *ah = v->atlas.ah;
  return TF_OK;
}

int tf_atlas_loc_next(tf_volume* v, uint64_t* loc_next) {
  if (!v || !loc_next) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  AtlasCtl c;
  TF_HIP(hipMemcpyAsync(&c, a.d_actl, sizeof(c), hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  unsigned long long tl = 0;
  slot_texloc(a.aw, a.ah, (int)a.pw, (int)a.ph, c.n_slots, &tl);
  *loc_next = tl;
  return TF_OK;
}

int tf_meshes_upload(tf_volume* v, const int32_t* ids, int64_t n, const int64_t* vert_offsets,
                     const int64_t* index_offsets, const float* verts, const float* normals, const float* colors,
                     const uint32_t* indices) {
  if (!v || (n > 0 && (!ids || !vert_offsets || !index_offsets || !verts || !normals || !colors))) {
    set_error("null argument");
    return TF_ERR_INVALID;
  }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  const int64_t nv = vert_offsets[n], ni = index_offsets[n];
  if (ni > 0 && !indices) { set_error("null index argument"); return TF_ERR_INVALID; }
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
  const size_t o_ids = take((size_t)n * 16), o_vo = take((size_t)(n + 1) * 8), o_io = take((size_t)(n + 1) * 8);
  const size_t o_v = take((size_t)nv * 12), o_n = take((size_t)nv * 12), o_c = take((size_t)nv * 12), o_i = take((size_t)ni * 4);
  int rc = atlas_stage(v, o);
  if (rc) return rc;
  AtlasState& a = v->atlas;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  uint8_t* ds = reinterpret_cast<uint8_t*>(a.d_stage);
  int32_t* hid = reinterpret_cast<int32_t*>(hs + o_ids);
  for (int64_t i = 0; i < n; ++i) {
    hid[4 * i] = ids[3 * i]; hid[4 * i + 1] = ids[3 * i + 1]; hid[4 * i + 2] = ids[3 * i + 2]; hid[4 * i + 3] = 0;
  }
  memcpy(hs + o_vo, vert_offsets, (size_t)(n + 1) * 8);
  memcpy(hs + o_io, index_offsets, (size_t)(n + 1) * 8);
  memcpy(hs + o_v, verts, (size_t)nv * 12);
  memcpy(hs + o_n, normals, (size_t)nv * 12);
  memcpy(hs + o_c, colors, (size_t)nv * 12);
  if (ni) memcpy(hs + o_i, indices, (size_t)ni * 4);
  TF_HIP(hipMemcpyAsync(ds, hs, o, hipMemcpyHostToDevice, v->stream));
  hipLaunchKernelGGL(k_mesh_scatter, dim3((unsigned)n), dim3(256), 0, v->stream, v->dev,
                     reinterpret_cast<const int4*>(ds + o_ids), (uint32_t)n, reinterpret_cast<const long long*>(ds + o_vo),
                     reinterpret_cast<const long long*>(ds + o_io), reinterpret_cast<const float*>(ds + o_v),
                     reinterpret_cast<const float*>(ds + o_n), reinterpret_cast<const float*>(ds + o_c),
                     reinterpret_cast<const uint32_t*>(ds + o_i), ++v->mesh_epoch);
  TF_HIP(hipGetLastError());
  v->host_list_n = -1;
  return tf_sync(v);
}

int tf_generate_patches(tf_volume* v, const int32_t* ids, int64_t n, const int32_t* labels, uint64_t out_hot[2]) {
  if (!v || (n > 0 && (!ids || !labels))) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  if (out_hot) {  // Chisel.cpp:153-154,184-186 with an empty loop
    const uint64_t ls = (uint64_t)a.aw * (uint64_t)a.ah;
    out_hot[0] = (ls / a.aw) * a.aw;
    out_hot[1] = (0 / a.aw + a.ph) * a.aw;
  }
  if (n <= 0) return TF_OK;
  std::vector<int> kfs((size_t)n);
  int last_id = 0, last_slot = -1;
  for (int64_t i = 0; i < n; ++i) {
    if (last_slot < 0 || labels[i] != last_id) {
      int rc = kf_slot_for(v, labels[i], false, &last_slot);
      if (rc) return rc;
      last_id = labels[i];
    }
    kfs[(size_t)i] = last_slot;
  }
  // (room behind the work list for the counters coming back: pinned, so that the copy does not go through the runtime's staging)
  const size_t o_ctl = ((size_t)n * 16 + 16 + 63) & ~(size_t)63;
  int rc = atlas_stage(v, o_ctl + sizeof(AtlasCtl));
  if (rc) return rc;
  rc = upload_work(v, ids, kfs.data(), n);
  if (rc) return rc;
  a.fused_armed = false;
  hipLaunchKernelGGL(k_work_lookup, dim3(((uint32_t)n + 255u) / 256u), dim3(256), 0, v->stream, v->dev, (uint32_t)n);
  hipLaunchKernelGGL(k_patch_assign, dim3(1), dim3(1024), 0, v->stream, v->dev, (uint32_t)n);
  prof_begin(v, TF_PROF_PATCH_PROJECT);
  hipLaunchKernelGGL((k_patch<true, false, false>), dim3(1024), dim3(256), 0, v->stream, v->dev, v->cam, 0, KfDev{});
  prof_end(v);
  TF_HIP(hipGetLastError());
  AtlasCtl* hc = reinterpret_cast<AtlasCtl*>(reinterpret_cast<uint8_t*>(a.h_stage) + o_ctl);
  TF_HIP(hipMemcpyAsync(hc, a.d_actl, sizeof(AtlasCtl), hipMemcpyDeviceToHost, v->stream));
  rc = sync_status(v, nullptr);
  const AtlasCtl c = *hc;
  if (out_hot && c.n_done > 0) {  // Chisel.cpp:184-186
    out_hot[0] = (c.loc_min / (uint64_t)a.aw) * (uint64_t)a.aw;
    out_hot[1] = (c.loc_max / (uint64_t)a.aw + a.ph) * (uint64_t)a.aw;
  }
  return rc;  // TF_ERR_ATLAS_FULL = GeneratePatches' -1 (Chisel.cpp:170-173)
}

int tf_update_atlas(tf_volume* v, const int32_t* ids, int64_t n) {
  if (!v || (n > 0 && !ids)) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  int rc = upload_work(v, ids, nullptr, n);
  if (rc) return rc;
  v->atlas.fused_armed = false;
  const uint32_t n32 = (uint32_t)n;
  hipLaunchKernelGGL(k_work_lookup, dim3((n32 + 255) / 256), dim3(256), 0, v->stream, v->dev, n32);
  prof_begin(v, TF_PROF_ATLAS_BLIT);
  {
    AtlasWriteScope aw(v, INT32_MIN);  // (patches of several keyframes: the label of the last fused frame stays)
    hipLaunchKernelGGL((k_patch<false, true, false>), dim3(1024), dim3(256), 0, v->stream, v->dev, v->cam, 0, KfDev{});
  }
  prof_end(v);
  TF_HIP(hipGetLastError());
  return tf_sync(v);
}

int tf_compensate_color(tf_volume* v, int64_t* out_n_clusters) {
  if (out_n_clusters) *out_n_clusters = 0;
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  std::vector<PatchRow> rows;
  int rc = list_patches(v, &rows);
  if (rc) return rc;
  const int64_t np = (int64_t)rows.size();
  if (!np) return TF_OK;
  // clusters by source frame in order of first appearance (Chisel.cpp:199-214)
  std::vector<int32_t> cl((size_t)np, -1), first;
  for (int64_t p = 0; p < np; ++p) {
    if (rows[(size_t)p].pflags & kPfAdjusted) continue;
    size_t k = 0;
    for (; k < first.size(); ++k)
      if (rows[(size_t)first[k]].frameid == rows[(size_t)p].frameid) break;
    if (k == first.size()) first.push_back((int32_t)p);
    cl[(size_t)p] = (int32_t)k;
  }
  const size_t ncl = first.size();
  if (out_n_clusters) *out_n_clusters = (int64_t)ncl;
  if (!ncl) return TF_OK;
  AtlasState& a = v->atlas;
  const size_t o_pt = 0;
  const size_t o_red = (o_pt + sizeof(CcPatch) * (size_t)np + 15) & ~(size_t)15;  // [ncl][12] sums / moments
  const size_t o_mean = o_red + ncl * 48;                                             // [ncl][6]
  const size_t o_xf = o_mean + ncl * 24;                                              // [ncl][16]
  const size_t total = o_xf + ncl * 64;
  rc = atlas_stage(v, total);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  uint8_t* ds = reinterpret_cast<uint8_t*>(a.d_stage);
  CcPatch* hp = reinterpret_cast<CcPatch*>(hs + o_pt);
  for (int64_t p = 0; p < np; ++p) {
    hp[p].slot = rows[(size_t)p].slot; hp[p].nv = rows[(size_t)p].nv;
    hp[p].cluster = cl[(size_t)p]; hp[p].wrong = (rows[(size_t)p].pflags & kPfWrong) ? 1 : 0;
  }
  TF_HIP(hipMemcpyAsync(ds, hs, o_red, hipMemcpyHostToDevice, v->stream));
  const CcPatch* dp = reinterpret_cast<const CcPatch*>(ds + o_pt);
  float* dred = reinterpret_cast<float*>(ds + o_red);
  float* hred = reinterpret_cast<float*>(hs + o_red);
  float* hmean = reinterpret_cast<float*>(hs + o_mean);
  float* hxf = reinterpret_cast<float*>(hs + o_xf);
  // computeMeanAndCov (Patch.cpp:342-348): mean, then centred second moments / (N - 1)
  hipLaunchKernelGGL(k_cc_reduce<0>, dim3((unsigned)ncl), dim3(256), 0, v->stream, v->dev, dp, np, (const float*)nullptr, dred);
  TF_HIP(hipMemcpyAsync(hred, dred, ncl * 48, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  std::vector<float> cnt(ncl);
  for (size_t c = 0; c < ncl; ++c) {
    cnt[c] = hred[c * 12 + 6];
    for (int i = 0; i < 6; ++i) hmean[c * 6 + i] = cnt[c] > 0.0f ? hred[c * 12 + i] / cnt[c] : 0.0f;
  }
  TF_HIP(hipMemcpyAsync(ds + o_mean, hmean, ncl * 24, hipMemcpyHostToDevice, v->stream));
  hipLaunchKernelGGL(k_cc_reduce<1>, dim3((unsigned)ncl), dim3(256), 0, v->stream, v->dev, dp, np,
                     reinterpret_cast<const float*>(ds + o_mean), dred);
  TF_HIP(hipMemcpyAsync(hred, dred, ncl * 48, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  for (size_t c = 0; c < ncl; ++c) {
    float* X = hxf + c * 16;
    for (int i = 0; i < 16; ++i) X[i] = 0.0f;
    if (cnt[c] <= 0.0f) continue;  // Chisel.cpp:242: empty cluster, has_adjusted stays false
    const float nm1 = cnt[c] - 1.0f;
    float cs[9], ct[9];
    const int idx[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    for (int i = 0; i < 9; ++i) { cs[i] = hred[c * 12 + idx[i]] / nm1; ct[i] = hred[c * 12 + 6 + idx[i]] / nm1; }
    color_transfer(cs, ct, X);
    for (int i = 0; i < 3; ++i) { X[9 + i] = hmean[c * 6 + i]; X[12 + i] = hmean[c * 6 + 3 + i]; }
    X[15] = 1.0f;
  }
  TF_HIP(hipMemcpyAsync(ds + o_xf, hxf, ncl * 64, hipMemcpyHostToDevice, v->stream));
  hipLaunchKernelGGL(k_cc_apply, dim3((unsigned)np), dim3(256), 0, v->stream, v->dev, dp,
                     reinterpret_cast<const float*>(ds + o_xf));
  TF_HIP(hipGetLastError());
  return tf_sync(v);
}

// Patch::complete (Patch.cpp:191-196) from a listed row
static bool row_complete(const PatchRow& r) {
  return r.nv > 0 && (r.state & kMsSimplified) && (r.pflags & kPfHasImage) && r.frameid >= 0;
}

static int draw_common(tf_volume* v, float* d_vertices, uint32_t* d_indices, float* h_vertices, uint32_t* h_indices,
                       int64_t cap_v, int64_t cap_i, int64_t* out_nv, int64_t* out_ni) {
  if (out_nv) *out_nv = 0;
  if (out_ni) *out_ni = 0;
  std::vector<PatchRow> rows;
  int rc = list_patches(v, &rows);
  if (rc) return rc;
  std::vector<DrawPatch> dp;
  dp.reserve(rows.size());
  unsigned long long vout = 0, iout = 0;
  for (const PatchRow& r : rows) {
    if (!row_complete(r)) continue;
    DrawPatch P;
    P.slot = r.slot; P.nv = r.nv; P.nt = r.nt;
    const bool labs_valid = (r.pflags & kPfAdjusted) && !(r.pflags & kPfWrong);  // has_adjusted && !labs.empty()
    P.flags = 1u | ((r.pflags & kPfWrong) ? 2u : 0u) | (labs_valid ? 4u : 0u);
    P.vout = vout; P.iout = iout;
    vout += r.nv; iout += 3ull * r.nt;
    dp.push_back(P);
  }
  if (out_nv) *out_nv = (int64_t)vout;
  if (out_ni) *out_ni = (int64_t)iout;
  if ((int64_t)vout > cap_v || (int64_t)iout > cap_i) { set_error("vertex / index buffer too small"); return TF_ERR_CAPACITY; }
  if (dp.empty()) return TF_OK;
  AtlasState& a = v->atlas;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
  const size_t o_pt = take(sizeof(DrawPatch) * dp.size());
  const size_t o_v = d_vertices ? 0 : take((size_t)vout * 48), o_i = d_indices ? 0 : take((size_t)iout * 4);
  rc = atlas_stage(v, o);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  uint8_t* ds = reinterpret_cast<uint8_t*>(a.d_stage);
  memcpy(hs + o_pt, dp.data(), sizeof(DrawPatch) * dp.size());
  TF_HIP(hipMemcpyAsync(ds + o_pt, hs + o_pt, sizeof(DrawPatch) * dp.size(), hipMemcpyHostToDevice, v->stream));
  float* dv = d_vertices ? d_vertices : reinterpret_cast<float*>(ds + o_v);
  uint32_t* di = d_indices ? d_indices : reinterpret_cast<uint32_t*>(ds + o_i);
  hipLaunchKernelGGL(k_draw, dim3((unsigned)dp.size()), dim3(256), 0, v->stream, v->dev,
                     reinterpret_cast<const DrawPatch*>(ds + o_pt), dv, di);
  TF_HIP(hipGetLastError());
  if (!d_vertices) TF_HIP(hipMemcpyAsync(hs + o_v, ds + o_v, (size_t)vout * 48, hipMemcpyDeviceToHost, v->stream));
  if (!d_indices && iout) TF_HIP(hipMemcpyAsync(hs + o_i, ds + o_i, (size_t)iout * 4, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  if (!d_vertices && h_vertices) memcpy(h_vertices, hs + o_v, (size_t)vout * 48);
  if (!d_indices && h_indices && iout) memcpy(h_indices, hs + o_i, (size_t)iout * 4);
  return TF_OK;
}

int tf_draw_meshes(tf_volume* v, float* vertices, uint32_t* indices, int64_t cap_vertices, int64_t cap_indices,
                   int64_t* n_vertices, int64_t* n_indices) {
  if (!v || !vertices || !indices) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return draw_common(v, nullptr, nullptr, vertices, indices, cap_vertices, cap_indices, n_vertices, n_indices);
}
int tf_draw_meshes_device(tf_volume* v, float* d_vertices, uint32_t* d_indices, int64_t cap_vertices,
                          int64_t cap_indices, int64_t* n_vertices, int64_t* n_indices) {
  if (!v || !d_vertices || !d_indices) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return draw_common(v, d_vertices, d_indices, nullptr, nullptr, cap_vertices, cap_indices, n_vertices, n_indices);
}

int tf_patches_download(tf_volume* v, const int32_t* ids, int64_t n, const int64_t* vert_offsets, uint64_t* texloc,
                        int32_t* frameid, int32_t* bbox, int32_t* flags, float* ratio, float* texcoord,
                        float* texcolor, float* labs) {
  if (!v || (n > 0 && !ids)) { set_error("null argument"); return TF_ERR_INVALID; }
  if ((texcoord || texcolor || labs) && !vert_offsets) { set_error("per-vertex outputs need vert_offsets"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  const int64_t nv = vert_offsets ? vert_offsets[n] : 0;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
  const size_t o_ids = take((size_t)n * 16), o_vo = take((size_t)(n + 1) * 8);
  const size_t o_in_end = o;
  const size_t o_ph = take(sizeof(PatchHost) * (size_t)n), o_tc = take((size_t)nv * 8), o_tcol = take((size_t)nv * 12),
               o_labs = take((size_t)nv * 12);
  int rc = atlas_stage(v, o);
  if (rc) return rc;
  AtlasState& a = v->atlas;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  uint8_t* ds = reinterpret_cast<uint8_t*>(a.d_stage);
  int32_t* hid = reinterpret_cast<int32_t*>(hs + o_ids);
  for (int64_t i = 0; i < n; ++i) {
    hid[4 * i] = ids[3 * i]; hid[4 * i + 1] = ids[3 * i + 1]; hid[4 * i + 2] = ids[3 * i + 2]; hid[4 * i + 3] = 0;
  }
  if (vert_offsets) memcpy(hs + o_vo, vert_offsets, (size_t)(n + 1) * 8);
  TF_HIP(hipMemcpyAsync(ds, hs, o_in_end, hipMemcpyHostToDevice, v->stream));
  if (o > o_tc) TF_HIP(hipMemsetAsync(ds + o_tc, 0, o - o_tc, v->stream));
  hipLaunchKernelGGL(k_patch_gather, dim3((unsigned)n), dim3(256), 0, v->stream, v->dev,
                     reinterpret_cast<const int4*>(ds + o_ids), (uint32_t)n,
                     vert_offsets ? reinterpret_cast<const long long*>(ds + o_vo) : nullptr,
                     reinterpret_cast<PatchHost*>(ds + o_ph), texcoord ? reinterpret_cast<float*>(ds + o_tc) : nullptr,
                     texcolor ? reinterpret_cast<float*>(ds + o_tcol) : nullptr,
                     labs ? reinterpret_cast<float*>(ds + o_labs) : nullptr);
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(hs + o_ph, ds + o_ph, o - o_ph, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  const PatchHost* ph = reinterpret_cast<const PatchHost*>(hs + o_ph);
  for (int64_t i = 0; i < n; ++i) {
    if (!ph[i].found) {
      set_error("chunk (" + std::to_string(ids[3 * i]) + "," + std::to_string(ids[3 * i + 1]) + "," +
                std::to_string(ids[3 * i + 2]) + ") has no mesh");
      return TF_ERR_MISSING_CHUNK;
    }
    if (texloc) texloc[i] = ph[i].texloc;
    if (frameid) frameid[i] = ph[i].frameid;
    if (bbox) memcpy(bbox + 4 * i, ph[i].bbox, 16);
    if (flags) flags[i] = (int32_t)ph[i].pflags;
    if (ratio) { ratio[2 * i] = ph[i].ratio[0]; ratio[2 * i + 1] = ph[i].ratio[1]; }
  }
  if (texcoord) memcpy(texcoord, hs + o_tc, (size_t)nv * 8);
  if (texcolor) memcpy(texcolor, hs + o_tcol, (size_t)nv * 12);
  if (labs) memcpy(labs, hs + o_labs, (size_t)nv * 12);
  return TF_OK;
}

// Rows [row0, row1) of the atlas as they are behind every atlas-writing launch that is on the handle's stream NOW, for a
// thread OTHER than the one that drives the handle (no deferred work is flushed, no handle state is touched beyond the
// reader's own buffers): a device-to-device copy of the rows goes INTO the handle's stream under atlas_mu -- between two of
// the map thread's launches, never inside one's enqueue -- so the rows are those of ONE moment of the stream, whatever the
// map thread enqueues meanwhile; an event behind it releases the copy to the host on the reader's own stream.
int tf_atlas_snapshot_rows(tf_volume* v, int64_t row0, int64_t row1, uint8_t* dst, int64_t* write_seq, int32_t* frame_id) {
  if (!v || !dst) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_HIP(hipSetDevice(v->device));
  AtlasState& a = v->atlas;
  if (row0 < 0 || row1 > a.ah || row0 > row1) { set_error("row range outside the atlas"); return TF_ERR_INVALID; }
  const size_t step = (size_t)a.aw * 3, bytes = (size_t)(row1 - row0) * step;
  std::lock_guard<std::mutex> readers(v->snap_mu);
  if (!v->read_stream) {
    TF_HIP(hipStreamCreateWithFlags(&v->read_stream, hipStreamNonBlocking));
    TF_HIP(hipEventCreateWithFlags(&v->read_ev, hipEventDisableTiming));
  }
  if (bytes > v->d_snap_bytes) {  // (grows by doubling; the free waits for the device once per growth)
    size_t want = std::max<size_t>(64 * step, 1);
    while (want < bytes) want <<= 1;
    if (v->d_snap) { TF_HIP(hipStreamSynchronize(v->read_stream)); TF_HIP(hipFree(v->d_snap)); v->d_snap = nullptr; v->d_snap_bytes = 0; }
    TF_HIP(hipMalloc((void**)&v->d_snap, want));
    v->d_snap_bytes = want;
  }
  uint64_t seq = 0;
  int32_t fid = -1;
  {
    std::lock_guard<std::mutex> lk(v->atlas_mu);
    if (bytes) TF_HIP(hipMemcpyAsync(v->d_snap, a.buf + (size_t)row0 * step, bytes, hipMemcpyDeviceToDevice, v->stream));
    TF_HIP(hipEventRecord(v->read_ev, v->stream));
    seq = v->atlas_seq.load(std::memory_order_acquire);
    fid = v->atlas_frame.load(std::memory_order_relaxed);
  }
  TF_HIP(hipStreamWaitEvent(v->read_stream, v->read_ev, 0));
  if (bytes) TF_HIP(hipMemcpyAsync(dst, v->d_snap, bytes, hipMemcpyDeviceToHost, v->read_stream));
  TF_HIP(hipStreamSynchronize(v->read_stream));
  if (write_seq) *write_seq = (int64_t)seq;
  if (frame_id) *frame_id = fid;
  return TF_OK;
}

int tf_atlas_download_rows(tf_volume* v, int64_t row0, int64_t row1, uint8_t* dst) {
  if (!v || !dst) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  if (row0 < 0 || row1 > a.ah || row0 > row1) { set_error("row range outside the atlas"); return TF_ERR_INVALID; }
  const size_t step = (size_t)a.aw * 3;
  if (row1 == row0) return TF_OK;
  TF_HIP(hipMemcpyAsync(dst, a.buf + (size_t)row0 * step, (size_t)(row1 - row0) * step,
                        hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  return TF_OK;
}

}  // extern "C"
