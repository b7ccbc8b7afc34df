// tf_mesh.hip -- marching-cubes meshing of dirty chunks on the device.
//
//   k_mesh           ChunkManager::GenerateMeshEfficient + extractGradientFromCubic
//                    (Structure/ChunkManager.cpp:595-1002, :277-455), RecomputeMeshes' bookkeeping
//                    (:232-264) and Mesh::SimplifyByClustering's own adjacency flags
//                    (3rd_party/open_chisel/geometry/Mesh.cpp:39-83)
//   k_compress       Chisel::CompressMeshes' neighbour exchange of the flags (Structure/Chisel.cpp:127-145)
//   k_list_meshes / k_mesh_counts / k_mesh_gather   host mirrors of ChunkManager::allMeshes
//
// One workgroup per chunk.  The chunk's 8^3 cells read the sdf of the 9^3 cell corners and, for the
// gradient normals, one more voxel on every side: the whole 11^3 neighbourhood (27 chunks) is
// staged in LDS once, a chunk that does not exist reads as the fresh state (sdf 999, weight 0) --
// exactly what the reference's "chunk missing" / "sdf > 1" / "dd < 1" tests reduce to.  The
// reference de-duplicates vertices on a 9x9x9x3 edge grid where the LAST cell (z, y, x loop order)
// that emits a triangle on an edge leaves its own rounding of the vertex; here every emitting cell
// posts its index with an LDS atomicMax per edge slot, a block scan ranks the used slots (= the
// reference's ascending-slot vertex order) and the winners are evaluated once, lane = output vertex.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tf_devfn.h"
#include "tf_kf_store.h"
#include "tf_mc_table.h"
#include "tf_patch_body.h"
#include "tf_volume.h"

#pragma clang fp contract(off)

namespace tf {

__device__ const unsigned long long d_mc_tri[256] = TF_MC_TRI_TABLE_INIT;

constexpr int kR = 11;               // staged region: voxel coordinates -1 .. 9 per axis
constexpr int kRS = 11;              // row stride in words (12 / 13 measured the same: the waves do not wait for LDS issue, profiles/r4)
constexpr int kPS = kRS * kR;        // plane stride
constexpr int kRV = kPS * kR;        // 1331 words at the natural stride
constexpr int kEdgeSlots = 3 * 729;  // vertByEdge (ChunkManager.cpp:646-648)
__device__ __forceinline__ uint32_t mesh_shard_rows_d(uint32_t max_chunks) { return max_chunks / kMeshShards + 258u; }  // = mesh_shard_rows()

__device__ __forceinline__ int ridx(int x, int y, int z) { return (x + 1) + (y + 1) * kRS + (z + 1) * kPS; }

// Index tables of the staging passes (the kernel is VALU-bound: no div / mod by 11 or 9 per voxel).
//   halo: the region voxels that belong to neighbour chunks AND are read by somebody -- a cell corner (all coordinates in
//         0..8) or the partner of a corner's central difference (one coordinate -1 or 9, the other two in 0..8): 703 of
//         the 11^3 - 8^3 = 819 -- fetched as PAIRS of x-adjacent voxels (voxels 2k, 2k + 1 of a chunk are 16 contiguous
//         bytes): 406 loads of 16 B instead of 819 of 8 B.  The staging pass is bound by the number of its scattered
//         loads (TA / L1 tag rate: every one is its own cache line), not by their bytes.  One 64-bit entry per pair:
//         region index of the pair's FIRST voxel | neighbour (0..26) << 11 | pair index in that chunk (voxel >> 1) << 16
//         | use first << 24 | use second << 25 | corner index + 1 of the first (0 = not a corner) << 26 | ... of the
//         second << 36   (the second voxel's region index is the first's + 1: x-adjacent)
//   corner: region index of cell corner c = px + 9 py + 81 pz
constexpr bool halo_needed(int rx, int ry, int rz) {
  if (rx < -1 || rx > 9 || ry < -1 || ry > 9 || rz < -1 || rz > 9) return false;  // outside the staged region
  if (rx >= 0 && rx < 8 && ry >= 0 && ry < 8 && rz >= 0 && rz < 8) return false;  // the chunk's own voxels
  const int out = ((rx < 0 || rx > 8) ? 1 : 0) + ((ry < 0 || ry > 8) ? 1 : 0) + ((rz < 0 || rz > 8) ? 1 : 0);
  return out <= 1;
}
constexpr int count_halo_pairs() {
  int n = 0;
  for (int rz = -1; rz <= 9; ++rz)
    for (int ry = -1; ry <= 9; ++ry)
      for (int rx = -1; rx <= 9; ++rx) {
        if (!halo_needed(rx, ry, rz)) continue;
        const int vx = (rx + 8) & 7;
        // counted at the pair's first needed member
        if ((vx & 1) && halo_needed(rx - 1, ry, rz) && (((rx - 1 + 8) >> 3) == ((rx + 8) >> 3))) continue;
        ++n;
      }
  return n;
}
constexpr int kHalo = count_halo_pairs();
struct MeshTabs {
  unsigned long long halo[kHalo];
  uint16_t corner[729];
};
constexpr unsigned long long corner_of(int rx, int ry, int rz) {
  return (rx >= 0 && ry >= 0 && rz >= 0 && rx <= 8 && ry <= 8 && rz <= 8) ? (unsigned long long)(rx + ry * 9 + rz * 81 + 1) : 0ull;
}
constexpr MeshTabs make_mesh_tabs() {
  MeshTabs t{};
  int n = 0;
  for (int rz = -1; rz <= 9; ++rz)
    for (int ry = -1; ry <= 9; ++ry)
      for (int rx = -1; rx <= 9; ++rx) {
        if (!halo_needed(rx, ry, rz)) continue;
        const int vx = (rx + 8) & 7;
        const bool same_chunk_left = ((rx - 1 + 8) >> 3) == ((rx + 8) >> 3);
        if ((vx & 1) && halo_needed(rx - 1, ry, rz) && same_chunk_left) continue;  // second member of a pair already emitted
        // this voxel is the first needed member of its pair: rx0 = region x of the pair's first (even) voxel
        const int rx0 = (vx & 1) ? rx - 1 : rx;
        const bool use0 = !(vx & 1);
        const bool use1 = (vx & 1) ? true : (halo_needed(rx + 1, ry, rz) && (((rx + 1 + 8) >> 3) == ((rx + 8) >> 3)));
        const int cx = (rx + 8) >> 3, cy = (ry + 8) >> 3, cz = (rz + 8) >> 3;
        const unsigned long long nb = (unsigned long long)(cx + cy * 3 + cz * 9);
        const unsigned long long pair = (unsigned long long)((((rx0 + 8) & 7) + ((ry + 8) & 7) * 8 + ((rz + 8) & 7) * 64) >> 1);
        const unsigned long long r0 = (unsigned long long)((rx0 + 1) + (ry + 1) * kRS + (rz + 1) * kPS);  // (rx0 = -2 never indexes: use0 is false then)
        t.halo[n++] = (r0 & 2047ull) | (nb << 11) | (pair << 16) | ((use0 ? 1ull : 0ull) << 24) | ((use1 ? 1ull : 0ull) << 25) |
                      ((use0 ? corner_of(rx0, ry, rz) : 0ull) << 26) | ((use1 ? corner_of(rx0 + 1, ry, rz) : 0ull) << 36);
      }
  for (int c = 0; c < 729; ++c) {
    const int px = c % 9, py = (c / 9) % 9, pz = c / 81;
    t.corner[c] = (uint16_t)((px + 1) + (py + 1) * kRS + (pz + 1) * kPS);
  }
  return t;
}
__device__ const MeshTabs d_mesh_tabs = make_mesh_tabs();

// cubeIndexOffsets (ChunkManager.cpp:65-66): corner k = (ox, oy, oz)
__device__ __forceinline__ int cox(int k) { return (0x66 >> k) & 1; }
__device__ __forceinline__ int coy(int k) { return (0xCC >> k) & 1; }
__device__ __forceinline__ int coz(int k) { return (0xF0 >> k) & 1; }
// edgeIndexPairs (ChunkManager.cpp:68-75), nibble e
__device__ __forceinline__ int edge_c0(int e) { return (int)((0x321076543210ull >> (4 * e)) & 0xF); }
__device__ __forceinline__ int edge_c1(int e) { return (int)((0x765447650321ull >> (4 * e)) & 0xF); }

// edge grid slot of edge e of cell (x, y, z) (ChunkManager.cpp:856-885)
__device__ __forceinline__ int edge_slot(int x, int y, int z, int e) {
  const int bx = x + ((0x622 >> e) & 1);  // e in {1, 5, 9, 10}
  const int by = y + ((0xC44 >> e) & 1);  // e in {2, 6, 10, 11}
  const int bz = z + ((0x0F0 >> e) & 1);  // e in {4, 5, 6, 7}
  const int ax = e >= 8 ? 2 : (e & 1);    // 0,2,4,6 -> x edges; 1,3,5,7 -> y edges; 8..11 -> z edges
  return ax + (bx + by * 9 + bz * 81) * 3;
}
// inverse: which edge of cell (x, y, z) is slot (axis, bx, by, bz)
__device__ __forceinline__ int edge_of_slot(int ax, int dx, int dy, int dz) {
  if (ax == 0) return dz ? (dy ? 6 : 4) : (dy ? 2 : 0);
  if (ax == 1) return dz ? (dx ? 5 : 7) : (dx ? 1 : 3);
  return dy ? (dx ? 10 : 11) : (dx ? 9 : 8);
}

// extractGradientFromCubic (ChunkManager.cpp:277-455) at the cell corner with region coordinates
// (px, py, pz) in 0..8 whose cube offset is (ox, oy, oz).  Both partners of every central
// difference are the physical neighbour voxels (in-cube partner / own chunk / face-neighbour chunk,
// voxelNeighborIndex :108-157); the partner OUTSIDE the cube is the one GetNeighborSDF fetches and
// it must be < 1 (ChunkManager.h:790-823).  Reduction order of Eigen's fixed-size norm:
// x*x + (y*y + z*z); normalize() divides by the root when the squared norm is positive.
__device__ __forceinline__ bool gradient_at(const float* __restrict__ S, int px, int py, int pz, int ox, int oy,
                                            int oz, float res, float g[3]) {
  const int c = ridx(px, py, pz);
  const float xm = S[c - 1], xp = S[c + 1];
  const float ym = S[c - kRS], yp = S[c + kRS];
  const float zm = S[c - kPS], zp = S[c + kPS];
  const bool ok = ((ox ? xp : xm) < 1.0f) && ((oy ? yp : ym) < 1.0f) && ((oz ? zp : zm) < 1.0f);
  const float gx = xp - xm, gy = yp - ym, gz = zp - zm;
  const float yz = gy * gy + gz * gz;
  const float sq = gx * gx + yz;
  const float n = sqrtf(sq);
  g[0] = gx; g[1] = gy; g[2] = gz;
  if (sq > 0.0f) { g[0] = gx / n; g[1] = gy / n; g[2] = gz / n; }
  return ok && !(n > res * 100.0f);
}

// per-corner flags of the staged 9^3 cell corners
constexpr uint32_t kCfHeavy = 64u;   // weight > 50 (weight_threshold, ChunkManager.cpp:776-777)
constexpr uint32_t kCfGradOk = 128u; // |gradient| <= 100 * resolution (:449-452)
// bits 0..5: the voxel one step along -x, +x, -y, +y, -z, +z has sdf < 1 (GetNeighborSDF, ChunkManager.h:790-823)

template <int NT>
struct MeshSh {
  float S[kRV];               // sdf, region coordinates -1..9
  uint8_t cflag[732]; // per cell corner: kCf* | neighbour bits
  uint32_t nslot[27];         // pool slot of chunk id + (-1..1)^3, kInvalidSlot = missing
  // output vertex index of a used edge slot m = rbase[m / kEpt] + popcount(rmask[m / kEpt] below bit m % kEpt), kEpt =
  // 2304 / threads: one {base, mask} pair per THREAD of the ranking pass instead of 2187 16-bit entries (LDS per
  // workgroup decides how many chunks a CU holds)
  unsigned long long rmask[NT];
  uint16_t rbase[NT];
  uint32_t ownq[(kEdgeSlots + 7) / 8];  // per edge slot a nibble: bit q = the q-th cell around the edge emits on it
  uint16_t cinfo[512];        // edges used by the cell's emitted triangles | triangle count << 12
  uint8_t cidx[512];          // the cell's MC case
  uint16_t toff[512];         // first output triangle of the cell (a mesh that fits has < 2^16)
  uint32_t wsum[8];
  uint32_t nv, nt, adj, any;
  uint32_t ncell;             // cells the surface passes through
  uint16_t clist[512];        // ... in no particular order (what is computed per cell is stored per cell)
  uint32_t ovf;               // the chunk's block of the mesh store (MeshRec::block): owned before this pass or handed out in it
  uint32_t rblock;            // MeshRec::block as it was before this pass
  uint32_t rpress;            // blk_pressure() as read with the record
  uint32_t rstate;            // MeshRec::state as it was before this pass
  unsigned long long rtexloc; // MeshRec::texloc
};

// The (up to four) cells around edge slot (ax, bx, by, bz), numbered q = 0..3 in DESCENDING cell index (z, then
// y, then x): the reference's cell loop runs in ascending index and the last emitting cell overwrites
// vertByEdge (:872-885), so the emitter with the smallest q owns the vertex.  q = d0 + 2 d1 with
//   x edges: (d0, d1) = (by - y, bz - z);  y edges: (bx - x, bz - z);  z edges: (bx - x, by - y).
__device__ __forceinline__ int edge_q(int e) {  // q of edge e seen from its cell
  // e: 0 1 2 3 4 5 6 7 8 9 10 11 -> 0 1 1 0 2 3 3 2 0 1 3 2
  return (int)((0x231023320110ull >> (4 * e)) & 0x3);
}
__device__ __forceinline__ int owner_cell(int m, int q) {
  const int ax = m % 3, b = m / 3;
  const int bx = b % 9, by = (b / 9) % 9, bz = b / 81;
  const int d0 = q & 1, d1 = q >> 1;
  int x, y, z;
  if (ax == 0) { x = bx; y = by - d0; z = bz - d1; }
  else if (ax == 1) { x = bx - d0; y = by; z = bz - d1; }
  else { x = bx - d0; y = by - d1; z = bz; }
  return x + 8 * y + 64 * z;
}

// ---------------------------------------------------------------------------------------
// Filter, one WAVE per dirty chunk.  A vertex needs a cell whose 8 corners are all observed (sdf <= 1) with
// both signs among them, and a corner with weight > 50 (:669-722, :776-777): if the 9^3 corner voxels of
// the chunk do not hold a positive sdf, a negative sdf and a weight above 50, the mesh is empty.  Most dirty
// chunks (in front of / behind the surface inside the truncation band, or seen too few times) end here
// after reading their own 4 KiB, or that plus the 217 corner voxels the +x/+y/+z neighbours contribute;
// survivors are flagged for k_mesh: surv[32 * entry + 0..26] = pool slots of the survivor's 27-chunk
// neighbourhood (13 = the chunk itself), surv[32 * entry + 13] = kInvalidSlot for everything else.
// ---------------------------------------------------------------------------------------
// fused flow: a dirty chunk that owns a mesh (ChunkManager::HasMesh) goes to the patch list of its shard; one
// without an atlas slot is also a slot candidate (Atlas::AddPatch will be called for it, in ascending id order).
// Called by ONE thread per chunk.
// (entry = {packed id lo, packed id hi, pool slot, mesh block}: the patch stage reads the mesh planes without waiting for
// the record; ids are 21 bits per axis everywhere -- the hash keys are the same packing)
__device__ __forceinline__ void patch_list_append(const VolumeDev& v, int ppar, uint32_t shard, const int4 id,
                                                  uint32_t slot, unsigned long long texloc, uint32_t blk) {
  const uint32_t rows = mesh_shard_rows_d(v.max_chunks);
  const uint32_t p = atomicAdd(&v.patch_cnt[((ppar & 1) * kMeshShards + shard) * 16], 1u);
  const unsigned long long key = pack_id(id.x, id.y, id.z);
  if (p < rows) v.patch_list[((size_t)(ppar & 1) * kMeshShards + shard) * rows + p] = make_int4((int)(uint32_t)key, (int)(uint32_t)(key >> 32), (int)slot, (int)blk);
  else atomicOr(&v.vctl->status, kStMeshFull);
  if (texloc == kNoTexloc) {
    const uint32_t c = atomicAdd(&v.actl->set[ppar & 1].n_cand, 1u);
    if (c < v.max_chunks) v.cand[c] = pack_id(id.x, id.y, id.z);
  }
}

// Mesh::Clear + "stays in allMeshes if it was there" (:244-262) for a chunk the filter ruled out; one thread
__device__ __forceinline__ void filter_reset_record(const VolumeDev& v, uint32_t own, const int4 id, uint32_t epoch, int ppar) {
  MeshRec* rec = &v.mesh_rec[own];
  const uint32_t was = rec->state;
  const uint32_t inmap = was & kMsInMap;
  // fused flow (ppar >= 0): CompressMeshes follows in the same frame and its SimplifyByClustering marks EVERY dirty
  // mesh of allMeshes simplified, with or without vertices (Chisel.cpp:116-126, Mesh.cpp:39-48)
  rec->nv = 0; rec->nt = 0; rec->state = inmap | ((ppar >= 0 && inmap) ? kMsSimplified : 0u); rec->epoch = epoch;
  if (inmap) {  // (most chunks the filter rules out never had a mesh: they end with the stores above)
    uint32_t blk = rec->block;
    const unsigned long long texloc = rec->texloc;
    if (blk != kBlkNone && blk_pressed(blk_pressure(v), blk)) {  // Mesh::Clear(): the storage goes back to its pool (tf_devfn.h)
      blk_release(v, blk);
      rec->block = blk = kBlkNone;
    }
    if (ppar >= 0) patch_list_append(v, ppar, own & (kMeshShards - 1u), id, own, texloc, blk);  // an emptied mesh keeps its patch
  }
}

// The same, postponed to the mesher launch (VolumeDev::reset_list): one thread
__device__ __forceinline__ void filter_defer_reset(const VolumeDev& v, uint32_t* cnt, uint32_t cap_sh, uint32_t own, const int4 id) {
  const uint32_t shard = own & (kMeshShards - 1u);
  const uint32_t p = atomicAdd(&cnt[(2u * kMeshShards + shard) * 16u], 1u);
  if (p < cap_sh) v.reset_list[(size_t)shard * cap_sh + p] = make_int4(id.x, id.y, id.z, (int)own);
  // (p >= cap_sh cannot happen: at most max_chunks / 32 pool slots share a shard, each listed once per frame)
}

// the 19 words of a neighbour-table row that are neither the chunk itself (13) nor one of the eight the summary test reads
__device__ const uint8_t kFarWord[19] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15, 18, 19, 20, 21, 24};
// phase A of the filter for the 8-lane group a lane belongs to (k8 = its place in the group): lane k stands for chunk
// id + (k & 1, (k >> 1) & 1, k >> 2) -- the chunk itself and its seven +x / +y / +z neighbours -- whose pool slots come from
// the chunk's row of the neighbour table (VolumeDev::nbr: lane 0 fetches the stamp of the row's last full check, lanes 1..7 their words; a
// row whose "none" words may be stale re-probes ALL of them, the group's eight lanes sharing the 25 words), and reads that chunk's class summary.  The summaries are supersets
// of the classes that occur among a chunk's voxels / on the faces the neighbours contribute (an edge or the corner counts
// as the whole face it lies in), so a chunk they rule out is ruled out for good.  own_in = the chunk's pool slot as every
// lane of the group knows it (kInvalidSlot: none; listed: it came with the list entry, whose id.w may name a hash entry).  Returns the lane's pool slot; *own = the chunk's (kInvalidSlot: the
// chunk is parked), *maybe = the exact test is needed.
__device__ __forceinline__ uint32_t filter_near(const VolumeDev& v, const int4 id, int lane, int k8, uint32_t own_in, bool listed,
                                                bool use_summ, uint32_t create_seq, uint32_t* own, bool* maybe) {
  uint32_t nslot = kInvalidSlot;
  *maybe = true;
  if (own_in != kInvalidSlot) {
    const int word = k8 == 0 ? kNbrStamp : 13 + (k8 & 1) + 3 * ((k8 >> 1) & 1) + 9 * (k8 >> 2);
    uint32_t w = v.nbr[(size_t)own_in * kNbrWords + word];
    // an entry K-A claimed (id.w = hash entry + 1) may have been parked later in that launch: RecomputeMeshes skips a
    // chunk that does not exist (:240-242).  This load travels with the row's.
    uint32_t alive = 1u;
    if (k8 == 0 && listed && id.w > 0) alive = v.hent[(uint32_t)id.w - 1u].alive & 1u;
    const uint32_t st = (uint32_t)__shfl((int)w, lane & 56);
    if (!(st > create_seq)) {  // (group-uniform) the row's "none" words may lack a chunk inserted since the last check
      if (k8 != 0 && w == 0u) {
        w = nbr_probe(v, pack_id(id.x + (k8 & 1), id.y + ((k8 >> 1) & 1), id.z + (k8 >> 2)));
        if (w) v.nbr[(size_t)own_in * kNbrWords + word] = w;
      }
      // ... and the other 19 words, two or three per lane: whoever reads the row of a chunk that went through a
      // filter launch -- the exact test, the mesher, the patch stage's flag exchange -- finds it checked in full
      for (int q = k8; q < 19; q += 8) {
        const int fw = (int)kFarWord[q];
        if (v.nbr[(size_t)own_in * kNbrWords + fw] == 0u) {
          const uint32_t got = nbr_probe(v, pack_id(id.x + fw % 3 - 1, id.y + (fw / 3) % 3 - 1, id.z + fw / 9 - 1));
          if (got) v.nbr[(size_t)own_in * kNbrWords + fw] = got;
        }
      }
      if (k8 == 0) v.nbr[(size_t)own_in * kNbrWords + kNbrStamp] = v.seq;
    }
    nslot = k8 == 0 ? (alive ? own_in : kInvalidSlot) : (w ? w - 1u : kInvalidSlot);
  }
  *own = (uint32_t)__shfl((int)nslot, lane & 56);
  if (use_summ && *own != kInvalidSlot) {
    const uint32_t sm = nslot != kInvalidSlot ? v.summ[nslot] : 0u;
    // self: whole chunk; +x, +x+y, +x+z, +x+y+z: x = 0 face; +y, +y+z: y = 0 face; +z: z = 0 face
    const uint32_t sel = k8 == 0 ? 0u : ((k8 & 1) ? 4u : ((k8 & 2) ? 8u : 12u));
    uint32_t u = (sm >> sel) & 15u;
    u |= (uint32_t)__shfl_xor((int)u, 1); u |= (uint32_t)__shfl_xor((int)u, 2); u |= (uint32_t)__shfl_xor((int)u, 4);
    const uint32_t so = (uint32_t)__shfl((int)sm, lane & 56);
    *maybe = (so & 1u) && (u & 14u) == 14u;
  }
  return nslot;
}

// phase B, one wave per entry.  A vertex needs a cell whose 8 corners are all observed with both signs among them,
// and a corner with weight > 50 (:669-722, :776-777): the exact test reads the chunk's own 4 KiB and, if those do not
// decide, the 217 corner voxels the +x / +y / +z neighbours contribute.  nslot: lanes 0..26 = pool slot of neighbourhood
// index 13 + dx + 3 dy + 9 dz (kInvalidSlot: no chunk) -- HAVE_ROW: from the caller, else fetched here from the chunk's row
// of the neighbour table, next to the voxel reads (own: the chunk's pool slot).  A survivor gets a row of its shard for
// the mesher's staging.
template <bool HAVE_ROW>
__device__ __forceinline__ void filter_exact(const VolumeDev& v, const int4 id, uint32_t nslot, const uint32_t own, int lane,
                                             uint32_t epoch, uint32_t* __restrict__ surv, uint32_t* __restrict__ cnt,
                                             uint32_t cap_sh, int ppar, bool use_summ, bool defer, uint32_t create_seq) {
  // rows and patch entries go to shard own % 32: pool slots are unique, so a shard never holds more than
  // max_chunks / 32 of them whatever the order of the work
  const uint32_t shard = own & (kMeshShards - 1u);
  const float4* T4 = reinterpret_cast<const float4*>(v.tsdf + (size_t)own * kChunkVoxels);
  uint32_t roww = 0u;  // (!HAVE_ROW: the row's word of this lane, requested ahead of the voxels, looked at behind them)
  if (!HAVE_ROW && lane < kNbrWords) roww = v.nbr[(size_t)own * kNbrWords + lane];
  float4 qv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qv[j] = T4[j * 64 + lane];  // voxels 2 i and 2 i + 1 of the chunk, i = 64 j + lane: {sdf, w, sdf, w}
  const uint32_t had_mesh = v.mesh_rec[own].state & kMsInMap;  // (travels with the voxels: which end of the row list, below)
  uint32_t fl = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 q = qv[j];
    const uint32_t c0 = classify_voxel(q.x, q.y), c1 = classify_voxel(q.z, q.w);
    // x = 0: the first voxel of every fourth pair; y = 0: (i >> 2) & 7 == 0; z = 0: i < 32
    fl |= c0 | c1 | ((lane & 3) ? 0u : c0 << 4) | ((lane & 28) ? 0u : (c0 | c1) << 8) | ((j || lane >= 32) ? 0u : (c0 | c1) << 12);
  }
  fl = wave_or(fl);
  if (!HAVE_ROW) {
    roww = nbr_row_checked(v, own, id, lane, roww, create_seq);
    nslot = lane == 13 ? own : ((lane < 27 && roww) ? roww - 1u : kInvalidSlot);
  }
  if (use_summ && lane == 0) v.summ[own] = fl;  // the chunk's summary is exact again
  if (lane == 0) atomicAdd(&cnt[(kMeshShards + shard) * 16], 1u);  // statistic (tf_texture_stats::n_exact): chunks whose voxels the filter read
  bool empty = !(fl & 1u);
  if (!empty && (fl & 14u) != 14u) {
    uint32_t f2 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // corner voxels with a coordinate 8: 3 faces of 64, 3 edges of 8, 1 corner
      const int q = j * 64 + lane;
      int cx = 0, cy = 0, cz = 0;
      if (q < 64) { cx = 8; cy = q & 7; cz = q >> 3; }
      else if (q < 128) { cx = q & 7; cy = 8; cz = (q >> 3) & 7; }
      else if (q < 192) { cx = q & 7; cy = (q >> 3) & 7; cz = 8; }
      else if (q < 200) { cx = 8; cy = 8; cz = q & 7; }
      else if (q < 208) { cx = 8; cy = q & 7; cz = 8; }
      else if (q < 216) { cx = q & 7; cy = 8; cz = 8; }
      else { cx = 8; cy = 8; cz = 8; }
      const uint32_t s = (uint32_t)__shfl((int)nslot, 13 + (cx >> 3) + 3 * (cy >> 3) + 9 * (cz >> 3));
      if (q < 217 && s != kInvalidSlot) {
        const float2 val = v.tsdf[(size_t)s * kChunkVoxels + (cx & 7) + (cy & 7) * 8 + (cz & 7) * 64];
        f2 |= classify_voxel(val.x, val.y);
      }
    }
    f2 = wave_or(f2);
    empty = ((fl | f2) & 14u) != 14u;
  }
  if (empty) {
    if (lane == 0) { if (defer) filter_defer_reset(v, cnt, cap_sh, own, id); else filter_reset_record(v, own, id, epoch, ppar); }
    return;
  }
  // The shard's rows are a two-ended list.  About a quarter of the survivors turn out to have no surface cell (the class
  // test is necessary, not sufficient) and leave the mesher after its cell pass, at half a chunk's time.  Which ones is
  // known well enough from the last time they were meshed: a chunk that has a mesh takes a row from the front, one that
  // has none from the back.  The mesher walks all fronts, then all backs: when the survivors exceed its resident capacity,
  // the workgroups that start late are the short ones.  (Any order gives the same meshes.)
  const bool back = !had_mesh;
  uint32_t p = 0;
  if (lane == 0) p = atomicAdd(&cnt[shard * 16 + (back ? 1 : 0)], 1u);
  p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
  if (p >= cap_sh) {  // (cannot happen: at most max_chunks / 32 pool slots share a shard)
    if (lane == 0) atomicOr(&v.vctl->status, kStMeshFull);
    return;
  }
  if (back) p = cap_sh - 1u - p;
  // [27..29] = the chunk id (the mesher does not go back to the list)
  if (lane >= 27 && lane < 30) nslot = (uint32_t)(lane == 27 ? id.x : (lane == 28 ? id.y : id.z));
  if (lane < 30) surv[32 * ((size_t)shard * cap_sh + p) + lane] = nslot;
}

// neighbourhood index of a lane (0..26) -> its place k among the near eight (only for is_near lanes)
__device__ __forceinline__ int near_k(int lane) { return (lane % 3 - 1) + 2 * ((lane / 3) % 3 - 1) + 4 * (lane / 9 - 1); }

// Two forms of the filter, one kernel each (one kernel holding both needs 78 VGPRs).  The wave form is compiled for SEVEN
// waves per SIMD: 72 VGPRs and no private memory.  At eight -- the whole list of a 640x480 frame resident at once -- it
// fits 64 VGPRs only with 12 B / lane spilled, and waves that own private memory are dispatched more slowly than the
// second round of a 7 k-wave launch costs: 15.8 -> 15.4 us (profiles/r5/README.md).
//  WAVE_FORM: one entry per wave (strided when the list is longer than the grid): lanes 0..7 run phase A, the wave
//             phase B for the same entry -- nothing to share, no barrier;
//  batches:   per workgroup, batches of up to 32 entries; phase A with EIGHT LANES per entry decides "cannot have a
//             vertex" for most of them without touching a voxel, an entry that passes is parked in LDS; phase B takes
//             the parked entries one wave each.  For long lists (the 69 k dirty chunks of the 1280x960 hall).
// Either form is correct for any list; the host picks by the list length it last heard of (*len_hint, written here
// into host-visible memory for the NEXT frame's choice -- no synchronisation, a stale value only costs time).
// The dirty set = the flat list [0, *dcount) followed by the concatenation of the 32 shard lists K-A filled (shards_par >= 0,
// VolumeDev::wl_*): entry e >= n_flat is row e - n_flat of that concatenation, resolved with a 32-lane scan of the counters.
// PATCH: the launch also carries the patch stage of the PREVIOUS frame as its first n_patch workgroups (patch_body,
// tf_patch_body.h: one wave per patch).  That stage reads the meshes the previous mesher left, the previous frame's images
// and the atlas; the filter touches hash, summaries, voxels, survivor rows and the reset list -- disjoint (the records it
// wants emptied are written by the mesher launch: filter_defer_reset) -- and this frame's mesher, which rewrites mesh
// blocks, runs behind the launch.  OFF by default (TF_PATCH_IN_FILTER=1): the filter alone is a 15-us latency chain and so
// are the patch chains, but next to each other they take 26 us -- more than the 9 us the stage costs K-A when it rides on
// k_frame (profiles/r4/README.md).
#ifndef TF_FILTER_BATCH_WAVES
#define TF_FILTER_BATCH_WAVES 6  // ... of the workgroup-batch form
#endif
#ifndef TF_FILTER_WAVES
#define TF_FILTER_WAVES 7  // waves per SIMD of the plain wave-form filter
#endif
#ifndef TF_FILTER_PATCH_WAVES
#define TF_FILTER_PATCH_WAVES 7  // waves per SIMD of the wave-form filter that carries the patch stage (71 VGPRs since round 6; the keyframe
                                 // unit: 190.2 / 192.7 -> 188.8 / 189.2 us per keyframe against 6); the batch form with the stage: TF_FILTER_BATCH_WAVES
#endif
struct FilterPatch {
  uint32_t defer;    // the records the filter wants emptied go to the reset list (a patch stage shares the launch) instead of
                     // being written here (32 counters take one atomic per ruled-out chunk: 57 k of them in a hall frame
                     // cost the filter 17 us, 5 k in a room frame 2-4 us -- profiles/r4 -- so only when it is needed)
  uint32_t n_patch;  // workgroups of the patch range (0: none)
  uint32_t first;    // first workgroup of the patch range: 0 = dispatched ahead of the filter's, else behind them
  int par;
  // N > 1, exchange overlapped with the interior meshes: 0 = every entry, 1 = only chunks whose 27-neighbourhood is owned
  // (part_interior: nothing of their meshes waits for the ghosts), 2 = only the others
  uint32_t cls;
  // the keyframe unit: workgroup 0 of the launch is not the filter's, it stores the finalized list of v.sel (kf_store_body)
  uint32_t n_store;
  KfStoreArgs store;
  Cam cam;
  KfDev kf;
};
// TL: the tuning instance (TF_MESH_DBG=10, tools/stamps.py filter): lane 0 of a wave stamps s_memrealtime into the debug
// table at the phase boundaries of its FIRST entry, row = wave; a phase that ends in loads is closed with a wait
template <bool WAVE_FORM, bool PATCH, bool TL = false>
__global__ __launch_bounds__(256, PATCH ? (WAVE_FORM ? TF_FILTER_PATCH_WAVES : TF_FILTER_BATCH_WAVES) : (WAVE_FORM ? TF_FILTER_WAVES : TF_FILTER_BATCH_WAVES)) void k_mesh_filter(VolumeDev v, const int4* __restrict__ dlist,
                                                     const uint32_t* __restrict__ dslot,
                                                     const uint32_t* __restrict__ dcount, uint32_t max_entries,
                                                     uint32_t epoch, uint32_t* __restrict__ surv,
                                                     uint32_t* __restrict__ cnt, uint32_t cap_sh, int ppar, bool use_summ,
                                                     uint32_t* __restrict__ len_hint, int shards_par, FilterPatch fp) {
  if (fp.n_store && blockIdx.x == 0) {  // (first: its chain of barriers is the launch's longest)
    kf_store_body<256>(v, fp.store.tab, fp.store.slots, fp.store.arena, fp.store.cap, fp.store.slot, fp.store.slack, fp.store.fill);
    return;
  }
  const uint32_t bx = blockIdx.x - fp.n_store, nbx = gridDim.x - fp.n_store;
  if (PATCH && bx - fp.first < fp.n_patch) {
    patch_body<true, true, true>(v, fp.cam, fp.par, fp.kf, bx - fp.first, fp.n_patch);
    return;
  }
  const uint32_t bid = PATCH ? (fp.first ? bx : bx - fp.n_patch) : bx;   // the filter's own block index / grid
  const uint32_t nblk = PATCH ? nbx - fp.n_patch : nbx;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool is_near = lane < 27 && (lane % 3) >= 1 && ((lane / 3) % 3) >= 1 && lane / 9 >= 1;
  const uint32_t create_seq = v.vctl->create_seq;  // (no launch that inserts keys runs next to a filter over the same chunks)
  const uint32_t tl_wave = (bid * 256 + threadIdx.x) >> 6;
  bool tl_first = true;
  auto stamp = [&](int k, bool wait) {
    if (TL && tl_first && tl_wave < (uint32_t)kPhaseWaves) {
      if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) v.phase_buf[tl_wave * 16 + k] = __builtin_amdgcn_s_memrealtime();
    }
  };
  if (TL && lane == 0 && tl_wave < (uint32_t)kPhaseWaves) { for (int k = 1; k < 8; ++k) v.phase_buf[tl_wave * 16 + k] = 0; }
  stamp(0, false);
  uint32_t n_flat = *dcount;
  if (n_flat > max_entries) n_flat = max_entries;
  // shard lists: per-lane inclusive scan of the 32 counters
  const uint32_t wl_rows = mesh_shard_rows_d(v.max_chunks);
  const size_t wl_base = (size_t)(shards_par & 1) * kMeshShards * wl_rows;
  uint32_t sh_n = 0, sh_incl = 0;
  if (shards_par >= 0 && (!WAVE_FORM || bid == 0)) {  // (the wave form needs the sums for the length hint only)
    if (lane < (int)kMeshShards) { sh_n = v.wl_cnt[((shards_par & 1) * kMeshShards + lane) * 16]; if (sh_n > wl_rows) sh_n = wl_rows; }
    sh_incl = sh_n;
#pragma unroll
    for (int o = 1; o < (int)kMeshShards; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)sh_incl, o);
      if (lane >= o) sh_incl += u;
    }
  }
  const uint32_t n_sh = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)sh_incl, kMeshShards - 1));
  uint32_t n = n_flat + n_sh;
  if (n > max_entries) n = max_entries;
  if (len_hint && bid == 0 && threadIdx.x == 0) *len_hint = n;
  const uint32_t nwaves = nblk * 4;
  if (WAVE_FORM) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((bid * 256 + threadIdx.x) >> 6));
    auto process = [&](const int4 id, const uint32_t own_listed, const bool have_own) {
      if (fp.cls && ((fp.cls == 1u) != part_interior(v, id.x, id.y, id.z))) return;  // (the other pass of this frame takes it)
      // the chunk's pool slot: carried by the fused flow's lists, else one hash lookup (call-by-call flow).  RecomputeMeshes
      // skips a chunk that does not exist (:240-242).
      const uint32_t own = have_own ? own_listed : hash_slot_alive(v, pack_id(id.x, id.y, id.z));
      if (own == kInvalidSlot) return;
      // ONE hop: the chunk's row of the neighbour table (the 27 pool slots around it) and -- for an entry K-A claimed
      // (id.w = hash entry + 1), which may have been parked later in that launch -- the entry's alive word
      uint32_t w = lane < kNbrWords ? v.nbr[(size_t)own * kNbrWords + lane] : 0u;
      uint32_t alive = 1u;
      if (have_own && id.w > 0) alive = v.hent[(uint32_t)id.w - 1u].alive & 1u;
      stamp(2, true);
      if (!alive) return;
      w = nbr_row_checked(v, own, id, lane, w, create_seq);
      const uint32_t nslot = lane == 13 ? own : ((lane < 27 && w) ? w - 1u : kInvalidSlot);
      if (use_summ) {
        // class summaries of the chunk and its seven +x / +y / +z neighbours (lanes 13, 14, 16, 17, 22, 23, 25, 26): supersets
        // of the classes among a chunk's voxels / on the face it contributes, so a chunk they rule out is ruled out for good
        const uint32_t sm = (is_near && nslot != kInvalidSlot) ? v.summ[nslot] : 0u;
        // self: whole chunk; +x (with or without +y, +z): x = 0 face; +y, +y+z: y = 0 face; +z: z = 0 face
        const uint32_t sel = lane == 13 ? 0u : ((lane % 3) == 2 ? 4u : (((lane / 3) % 3) == 2 ? 8u : 12u));
        const uint32_t u = is_near ? (sm >> sel) & 15u : 0u;
        const bool so = (__shfl((int)sm, 13) & 1) != 0;
        stamp(3, true);
        const bool maybe = so && __ballot(u & 2u) != 0ull && __ballot(u & 4u) != 0ull && __ballot(u & 8u) != 0ull;
        if (!maybe) {
          if (lane == 0) { if (fp.defer) filter_defer_reset(v, cnt, cap_sh, own, id); else filter_reset_record(v, own, id, epoch, ppar); }
          stamp(4, true);
          return;
        }
      }
      filter_exact<true>(v, id, nslot, own, lane, epoch, surv, cnt, cap_sh, ppar, use_summ, fp.defer != 0, create_seq);
      stamp(5, true);
    };
    // the shard lists K-A filled: wave w walks shard w % 32 from position w / 32 on.  Its first entry is requested TOGETHER
    // with the shard's counter (the entry's address does not depend on the count; a position beyond the count holds an older
    // frame's entry, dropped when the count arrives): one dependent round trip less per entry than mapping the wave's
    // index through a scan of the 32 counters -- the filter is a chain of such round trips, ~2 us each under its own load.
    if (shards_par >= 0) {
      const uint32_t sd = wave & (kMeshShards - 1u);
      const uint32_t stride = (nwaves - sd + kMeshShards - 1u) / kMeshShards;  // waves that share this shard
      uint32_t i = wave / kMeshShards;
      size_t at = wl_base + (size_t)sd * wl_rows + (i < wl_rows ? i : 0u);
      int4 id = v.wl_ids[at];
      uint32_t own_listed = v.wl_slot[at];
      uint32_t cs = v.wl_cnt[((shards_par & 1) * kMeshShards + sd) * 16];
      if (cs > wl_rows) cs = wl_rows;
      stamp(1, true);
      while (i < cs) {
        process(id, own_listed, true);
        stamp(6, false);
        tl_first = false;
        i += stride;
        if (i < cs) { at = wl_base + (size_t)sd * wl_rows + i; id = v.wl_ids[at]; own_listed = v.wl_slot[at]; }
      }
    }
    // the flat list (ghost arrivals, a backlog of earlier frames' marks, the call-by-call flow)
    for (uint32_t entry = wave; entry < n_flat; entry += nwaves) {
      const int4 id = dlist[entry];
      uint32_t own_listed = kInvalidSlot;
      if (dslot) own_listed = dslot[entry];  // (under a wave-uniform test: see profiles/r2/README.md)
      process(id, own_listed, dslot != nullptr);
    }
    return;
  }
  __shared__ uint32_t s_n, s_ne;
  __shared__ uint32_t s_incl[kMeshShards], s_cnt[kMeshShards];
  if (threadIdx.x < kMeshShards) { s_incl[threadIdx.x] = sh_incl; s_cnt[threadIdx.x] = sh_n; }  // (wave 0's lanes 0..31)
  __shared__ uint32_t s_eown[32];  // entries the summaries ruled out: their records are reset behind the barrier, by the
  __shared__ int4 s_eid[32];       // last wave, so that no entry of phase B waits for those round trips
  __shared__ int4 s_id[32];
  __shared__ uint32_t s_own[32];
  const int grp = threadIdx.x >> 3, k8 = threadIdx.x & 7;
  const uint32_t per = (n + nblk - 1) / nblk;
  const uint32_t first = bid * per;
  const uint32_t last = first + per < n ? first + per : n;
  for (uint32_t base = first; base < last; base += 32u) {
    if (threadIdx.x == 0) { s_n = 0; s_ne = 0; }
    __syncthreads();
    const uint32_t entry = base + (uint32_t)grp;
    if (entry < last) {
      int4 id;
      uint32_t own_in = kInvalidSlot;  // (every lane of the group: the row loads of phase A hang on it)
      if (entry < n_flat) {
        id = dlist[entry];
        own_in = dslot ? dslot[entry] : hash_slot_alive(v, pack_id(id.x, id.y, id.z));
      } else {
        const uint32_t r = entry - n_flat;
        uint32_t shd = 0;
        for (uint32_t q = 0; q < kMeshShards; ++q) shd += s_incl[q] <= r ? 1u : 0u;  // (32 LDS words: a linear pass)
        shd &= kMeshShards - 1u;
        const size_t at = wl_base + (size_t)shd * wl_rows + (r - (s_incl[shd] - s_cnt[shd]));
        id = v.wl_ids[at];
        own_in = v.wl_slot[at];
      }
      uint32_t own;
      bool maybe;
      filter_near(v, id, lane, k8, own_in, dslot != nullptr || entry >= n_flat, use_summ, create_seq, &own, &maybe);
      if (fp.cls && ((fp.cls == 1u) != part_interior(v, id.x, id.y, id.z))) own = kInvalidSlot;  // (group-uniform: the other pass takes it)
      if (own != kInvalidSlot && k8 == 0) {  // RecomputeMeshes: !HasChunk -> skip (:240-242)
        if (maybe) {
          const uint32_t at = atomicAdd(&s_n, 1u);
          s_own[at] = own; s_id[at] = id;
        } else {
          const uint32_t at = atomicAdd(&s_ne, 1u);
          s_eown[at] = own; s_eid[at] = id;
        }
      }
    }
    __syncthreads();
    if (w == 3 && lane < (int)s_ne) {
      if (fp.defer) filter_defer_reset(v, cnt, cap_sh, s_eown[lane], s_eid[lane]);
      else filter_reset_record(v, s_eown[lane], s_eid[lane], epoch, ppar);
    }
    const uint32_t nm = s_n;
    for (uint32_t m = (uint32_t)w; m < nm; m += 4u)
      filter_exact<false>(v, s_id[m], kInvalidSlot, s_own[m], lane, epoch, surv, cnt, cap_sh, ppar, use_summ, fp.defer != 0, create_seq);
    __syncthreads();  // the parked entries are consumed before the next batch overwrites them
  }
}

// tuning aid (TF_MESH_DBG=9, tools/stamps.py): thread 0 of a workgroup stamps the phases of its FIRST chunk into the
// debug table, row = workgroup
__device__ __forceinline__ void mesh_stamp(const VolumeDev& v, uint32_t r, int k) {
  if (threadIdx.x == 0 && blockIdx.x < (uint32_t)kPhaseWaves && r == blockIdx.x)
    v.phase_buf[blockIdx.x * 16 + k] = __builtin_amdgcn_s_memrealtime();
}

#ifndef TF_MESH_WAVES
// waves per SIMD the 128-thread mesher is compiled for.  6 = 80 VGPRs (no private memory) and, with 12.4 KB of LDS per
// workgroup, TWELVE chunks per CU = 3072 resident workgroups: a room frame's ~2.9 k surviving chunks all start at once (time
// stamps, profiles/r5: last start 1.2 us instead of 20 us, first start to last end 28.3 instead of 35.2 us; a chunk then takes
// 20.9 instead of 19.4 us -- the CU is shared by more).  Over the orbit, whose frames reach 3.3 k chunks: 33.1 -> 31.7 us.
// (Round 4 measured no gain at 6: its kernel needed 88 VGPRs, and the 80 the compiler was forced to cost more than they gave.)
#define TF_MESH_WAVES 6
#endif
// DBG: the tuning instance (TF_MESH_DBG: triage cut-offs 1..4 -- results are WRONG with one --, phase stamps 9); the product
// instance carries none of their branches
template <int NT, bool DBG = false>  // NT: threads per chunk
__global__ __launch_bounds__(NT, TF_MESH_WAVES) void k_mesh(VolumeDev v, const uint32_t* __restrict__ surv, uint32_t* __restrict__ cnt,
                                                 uint32_t* __restrict__ cnt_next, uint32_t cap_sh,
                                                 uint32_t epoch, float res, uint32_t simplified, uint32_t dbg,
                                                 int rearm) {
  __shared__ MeshSh<NT> sh;
  // (LDS per workgroup decides how many chunks a CU holds at once: the triangle table is read from memory -- a few
  // dozen cached 8-byte reads per chunk --, the list of used edge slots is sized by the mesh capacity: dynamic LDS)
  extern __shared__ uint16_t vlist[];  // [mesh_cv] used edge slots in ascending order
  const int t = threadIdx.x, lane = t & 63;
  // The survivors sit in 32 shard lists of different lengths.  Workgroup b takes the b-th row of their CONCATENATION
  // (every wave reads the 32 counters and scans them): the workgroups that have a chunk are then exactly the first
  // N of the grid, and with N below the resident capacity (2560) all of them start at once.  With b -> (shard b % 32,
  // row b / 32) the busy workgroups reached up to 32 x the LONGEST list; the few beyond the resident capacity started
  // when the first round ended and set the kernel's time (time stamps: last start 18-20 us, last end 34 us of which a
  // chunk takes 20).
  // (the lists are two-ended, filter_exact: lanes 0..31 hold the shards' front counts, lanes 32..63 their back counts; the
  // concatenation is every front, then every back; list L = shard L % 32, back end iff L >= 32)
  uint32_t n_rows = 0, excl_l = 0, incl_l = 0;
  {
    uint32_t my_n = cnt[(lane & 31) * 16 + (lane >> 5)];
    if (my_n > cap_sh) my_n = cap_sh;
    uint32_t incl = my_n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += u;
    }
    n_rows = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)incl, 63));  // (wave-uniform: scalar registers)
    // this workgroup's first row (the common case: its only one), resolved here so that nothing of the scan stays live
    const uint32_t r0 = blockIdx.x;
    const uint32_t sh0 = (uint32_t)__popcll(__ballot(incl <= r0));
    incl_l = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)incl, (int)(sh0 & 63u)));
    excl_l = incl_l - (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)my_n, (int)(sh0 & 63u)));
    incl_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh0);  // (reused: the list of the first row)
  }
  if (blockIdx.x == 0 && t < (int)kMeshShards) { cnt_next[t * 16] = 0u; cnt_next[t * 16 + 1] = 0u; cnt_next[(kMeshShards + t) * 16] = 0u; cnt_next[(kMeshShards + t) * 16 + 1] = 0u; cnt_next[(2 * kMeshShards + t) * 16] = 0u; }  // the counters of the NEXT launch's filter
  if (blockIdx.x < kMeshShards) {
    // the records the filter wanted emptied (filter_defer_reset): shard b by workgroup b, one thread per record, ahead
    // of the workgroup's own chunk -- nothing in this launch reads another chunk's record
    uint32_t n_reset = cnt[(2u * kMeshShards + blockIdx.x) * 16u];
    if (n_reset > cap_sh) n_reset = cap_sh;
    for (uint32_t i = (uint32_t)t; i < n_reset; i += NT) {
      const int4 e = v.reset_list[(size_t)blockIdx.x * cap_sh + i];
      filter_reset_record(v, (uint32_t)e.w, make_int4(e.x, e.y, e.z, 0), epoch, rearm >= 0 ? (rearm ^ 1) : -1);
    }
  }
  const float half = res * 0.5f;
  if (rearm >= 0 && blockIdx.x == 0 && t == 0) {
    // fused flow: the patches of the previous frame are done (the main stream waited for them ahead of this
    // frame's filter), so the counter set the NEXT frame's dirty list will append to can be re-armed
    AtlasCtl::Set* O = &v.actl->set[rearm];
    O->n_work = 0; O->n_cand = 0; O->n_patch = 0; O->slots_base = 0ull;
    // ... and the slot allocator's position is final: the patch kernel of THIS frame ranks its new patches against it
    v.actl->set[rearm ^ 1].slots_base = v.actl->n_slots;
  }
  if (rearm >= 0 && blockIdx.x == 0 && t < (int)kMeshShards) {
    v.patch_cnt[((rearm & 1) * kMeshShards + t) * 16] = 0u;
    v.wl_cnt[((rearm & 1) * kMeshShards + t) * 16] = 0u;  // ... and the shard lists K-A of the next frame appends its dirty set to
  }
  if (DBG && dbg == 9) mesh_stamp(v, blockIdx.x, 0);
  for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    uint32_t shard, own;
    int4 id;
    float2 a[512 / NT];
    {
      shard = incl_l;
      uint32_t idx = r - excl_l;
      if (r != blockIdx.x) {  // a further row of this workgroup (lists longer than the grid): scan again
        uint32_t my_n = cnt[(lane & 31) * 16 + (lane >> 5)];
        if (my_n > cap_sh) my_n = cap_sh;
        uint32_t incl = my_n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t u = (uint32_t)__shfl_up((int)incl, o);
          if (lane >= o) incl += u;
        }
        shard = (uint32_t)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(incl <= r)));
        idx = r - (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)incl, (int)(shard & 63u)) - __shfl((int)my_n, (int)(shard & 63u)));
      }
      if (shard >= kMeshShards) { shard -= kMeshShards; idx = cap_sh - 1u - idx; }  // (a row of the list's back end)
      const size_t row = (size_t)shard * cap_sh + idx;
      own = surv[32 * row + 13];    // the chunk's pool slot
      id = make_int4((int)surv[32 * row + 27], (int)surv[32 * row + 28], (int)surv[32 * row + 29], 0);
#pragma unroll
      for (int j = 0; j < 512 / NT; ++j) a[j] = v.tsdf[(size_t)own * kChunkVoxels + j * NT + t];
      __syncthreads();  // the previous chunk of this workgroup is done with the shared tables
      if (t < 27) sh.nslot[t] = surv[32 * row + t];
    }
    // (what the rest of the chunk derives from the thread index is recomputed here, behind an opaque statement, instead
    // of living in registers across the row / voxel loads above: 88 instead of 96 VGPRs, no private memory)
    int t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));
    const int t = t_, lane = t & 63, w = t >> 6;
    MeshRec* rec = &v.mesh_rec[own];
    // the thread's entries of the halo table travel with the own voxels: the staging pass below is then ONE hop of
    // scattered loads instead of table -> voxel
    unsigned long long htab[(kHalo + NT - 1) / NT];
#pragma unroll
    for (int j = 0; j < (kHalo + NT - 1) / NT; ++j) htab[j] = (j * NT + t < kHalo) ? d_mesh_tabs.halo[j * NT + t] : ~0ull;
    if (DBG && dbg == 9) mesh_stamp(v, r, 1);
    if (t == 0) { sh.nv = 0; sh.nt = 0; sh.adj = 0; sh.ncell = 0; }
    // the record's previous state travels with the first batch of loads, so that the tail of the chunk is stores only
    if (t == NT - 64) { sh.rstate = rec->state; sh.rblock = rec->block; sh.rtexloc = rec->texloc; sh.rpress = blk_pressure(v); }
    if (DBG && dbg == 1) continue;  // triage: filter only
    // ---- stage the 11^3 voxels of the neighbourhood (own ones from registers)
#pragma unroll
    for (int j = 0; j < 512 / NT; ++j) {
      const int q = j * NT + t;
      const int x0 = q & 7, y0 = (q >> 3) & 7, z0 = q >> 6;
      sh.S[ridx(x0, y0, z0)] = a[j].x; sh.cflag[x0 + y0 * 9 + z0 * 81] = (a[j].y > 50.0f) ? kCfHeavy : 0u;
    }
    __syncthreads();
    // (measured: issuing the row, the table entries and then own + halo voxels as two batches of loads -- two
    // dependent hops instead of four -- is slower, 36 -> 46 us: the extra registers spill)
#pragma unroll
    for (int j = 0; j < (kHalo + NT - 1) / NT; ++j) {
      const unsigned long long e = htab[j];
      if (e == ~0ull) continue;
      const uint32_t s = sh.nslot[(uint32_t)(e >> 11) & 31u];
      float4 val = make_float4(999.0f, 0.0f, 999.0f, 0.0f);  // a missing chunk reads as the fresh state
      if (s != kInvalidSlot) val = reinterpret_cast<const float4*>(v.tsdf + (size_t)s * kChunkVoxels)[(uint32_t)(e >> 16) & 255u];
      const uint32_t r0 = (uint32_t)e & 2047u;
      if ((e >> 24) & 1ull) {
        sh.S[r0] = val.x;
        const uint32_t cf = (uint32_t)(e >> 26) & 1023u;
        if (cf) sh.cflag[cf - 1u] = (val.y > 50.0f) ? kCfHeavy : 0u;
      }
      if ((e >> 25) & 1ull) {
        sh.S[r0 + 1u] = val.z;
        const uint32_t cf = (uint32_t)(e >> 36) & 1023u;
        if (cf) sh.cflag[cf - 1u] = (val.w > 50.0f) ? kCfHeavy : 0u;
      }
    }
    __syncthreads();
    if (DBG && dbg == 9) mesh_stamp(v, r, 3);
    // ---- per corner: which of its six neighbours are below 1, is its gradient short enough.  A cell asks
    // for the three neighbours OUTSIDE its cube (extractGradientFromCubic fetches those through
    // GetNeighborSDF, :320-447), so the answer per (cell, corner) is three of these bits.
    for (int c = t; c < 729; c += NT) {
      const int r = d_mesh_tabs.corner[c];
      const float xm = sh.S[r - 1], xp = sh.S[r + 1], ym = sh.S[r - kRS], yp = sh.S[r + kRS];
      const float zm = sh.S[r - kPS], zp = sh.S[r + kPS];
      uint32_t f = (xm < 1.0f ? 1u : 0u) | (xp < 1.0f ? 2u : 0u) | (ym < 1.0f ? 4u : 0u) | (yp < 1.0f ? 8u : 0u) |
                   (zm < 1.0f ? 16u : 0u) | (zp < 1.0f ? 32u : 0u);
      const float gx = xp - xm, gy = yp - ym, gz = zp - zm;
      const float yz = gy * gy + gz * gz;
      const float nrm = sqrtf(gx * gx + yz);
      if (!(nrm > res * 100.0f)) f |= kCfGradOk;
      sh.cflag[c] |= (uint8_t)f;
    }
    for (int i = t; i < (kEdgeSlots + 7) / 8; i += NT) sh.ownq[i] = 0u;
    __syncthreads();

    if (DBG && dbg == 2) continue;  // triage: + staging and corner flags
    if (DBG && dbg == 9) mesh_stamp(v, r, 4);
    // ---- pass 1: per cell, the MC case, the edges its emitted triangles use, how many triangles.
    // 1a: every cell's case from its 8 corners; the few cells the surface passes through (64 of 512 for a plane) go
    // to a list, so that 1b -- edge validity, triangles, edge ownership: the long part -- runs on dense lanes instead
    // of being walked by every wave for a handful of active lanes each (per-workgroup time stamps: the cell pass took
    // 3.8 us median / 8.9 us worst of a chunk's 17 us).
    for (int cell = t; cell < 512; cell += NT) {
      const int x = cell & 7, y = (cell >> 3) & 7, z = cell >> 6;
      const int c0 = ridx(x, y, z);
      bool observed = true;
      int pos = 0, index = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = sh.S[c0 + cox(k) + coy(k) * kRS + coz(k) * kPS];
        observed = observed && !(s > 1.0f);  // :669-720
        pos += (s > 0.0f) ? 1 : 0;
        index |= (0.0f > s) ? (1 << k) : 0;  // :726-735
      }
      sh.cinfo[cell] = 0;
      if (observed && (pos % 8) > 0 && index != 0 && index != 255)  // :722; cases 0 and 255 are the ones without a triangle
        sh.clist[atomicAdd(&sh.ncell, 1u)] = (uint16_t)cell;
    }
    __syncthreads();
    if (t == 0 && sh.ncell) atomicAdd(&cnt[(kMeshShards + shard) * 16 + 1], 1u);  // statistic (tf_texture_stats::n_surface)
    // 1b (loops stay rolled and re-read LDS instead of keeping more in registers: occupancy matters more)
    for (uint32_t ci = t; ci < sh.ncell; ci += NT) {
      const int cell = sh.clist[ci];
      const int x = cell & 7, y = (cell >> 3) & 7, z = cell >> 6;
      const int c0 = ridx(x, y, z);
      float cube[8];
      int index = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float s = sh.S[c0 + cox(k) + coy(k) * kRS + coz(k) * kPS];
        cube[k] = s;
        index |= (0.0f > s) ? (1 << k) : 0;
      }
      const unsigned long long row = d_mc_tri[index];
      uint32_t valid = 0;
      const int cf0 = x + y * 9 + z * 81;
#pragma unroll
      for (int e = 0; e < 12; ++e) {  // :748-834 (e0 / e1 are compile-time constants here)
        const int e0 = (int)((0x321076543210ull >> (4 * e)) & 0xF), e1 = (int)((0x765447650321ull >> (4 * e)) & 0xF);
        const float s0 = cube[e0], s1 = cube[e1];
        const bool far = fabsf(s0) > fabsf(s1);  // the corner with the smaller |sdf| (:765)
        const int ko = far ? (cox(e1) + coy(e1) * 9 + coz(e1) * 81) : (cox(e0) + coy(e0) * 9 + coz(e0) * 81);
        const uint32_t need0 = kCfHeavy | kCfGradOk | (1u << cox(e0)) | (4u << coy(e0)) | (16u << coz(e0));
        const uint32_t need1 = kCfHeavy | kCfGradOk | (1u << cox(e1)) | (4u << coy(e1)) | (16u << coz(e1));
        const uint32_t need = far ? need1 : need0;
        if (s0 * s1 < 0.0f) {
          const uint32_t have = sh.cflag[cf0 + ko];
          if ((have & need) == need) valid |= 1u << e;
        }
      }
      uint32_t ntri = 0, used = 0;
#pragma unroll 1
      for (int col = 0; col < 15; col += 3) {  // :836-918
        const int s0 = (int)((row >> (4 * col)) & 0xF);
        if (s0 == 0xF) break;
        const int s1 = (int)((row >> (4 * col + 4)) & 0xF), s2 = (int)((row >> (4 * col + 8)) & 0xF);
        if (!((valid >> s0) & (valid >> s1) & (valid >> s2) & 1u)) continue;
        ++ntri;
        used |= (1u << s0) | (1u << s1) | (1u << s2);
      }
      for (uint32_t u = used; u; u &= u - 1) {  // this cell emits on these edges
        const int e = __builtin_ctz(u);
        const int m = edge_slot(x, y, z, e);
        atomicOr(&sh.ownq[m >> 3], 1u << (4 * (m & 7) + edge_q(e)));
      }
      sh.cinfo[cell] = (uint16_t)(used | (ntri << 12));
      sh.cidx[cell] = (uint8_t)index;
    }
    __syncthreads();

    if (DBG && dbg == 3) continue;  // triage: + cell pass
    if (DBG && dbg == 9) mesh_stamp(v, r, 5);
    // ---- ranks: used edge slots in ascending order (the reference's vertex order, :886-897) and the
    // cells' triangle offsets in cell order (the order of mesh->indices)
    {
      constexpr int kEpt = 2304 / NT, kCpt = 512 / NT;  // NT x kEpt = 2304 >= 2187 edge slots; kCpt cells per thread
      // the thread's kEpt slots are kEpt nibbles of the ownership stream: pull them out as one bit per slot
      // (nibble != 0) instead of walking them one by one
      const int first = t * kEpt;
      unsigned long long usedm = 0;  // bit j = slot first + j is used
      {
        constexpr int kWords = (kEpt * 4 + 31) / 32 + 1;
        const int w0 = first >> 3, sh0 = 4 * (first & 7);
#pragma unroll
        for (int q = 0; q < kWords; ++q) {
          const int wi = w0 + q;
          uint32_t x = wi < (kEdgeSlots + 7) / 8 ? sh.ownq[wi] : 0u;
          x |= x >> 1; x |= x >> 2; x &= 0x11111111u;          // bit 4k = nibble k is non-zero
          x = (x | (x >> 3)) & 0x03030303u;                    // gather: 2 bits per byte
          x = (x | (x >> 6)) & 0x000F000Fu;                    //         4 bits per half
          x = (x | (x >> 12)) & 0xFFu;                         //         8 bits: slot 8 wi + k -> bit k
          usedm |= (unsigned long long)x << (8 * q);
        }
        usedm >>= (sh0 >> 2);
        usedm &= (1ull << kEpt) - 1ull;
        const int left = kEdgeSlots - first;
        if (left < kEpt) usedm = left > 0 ? (usedm & ((1ull << left) - 1ull)) : 0ull;
      }
      const uint32_t cnt = (uint32_t)__popcll(usedm);
      uint32_t tc = 0;
#pragma unroll
      for (int j = 0; j < kCpt; ++j) tc += (uint32_t)sh.cinfo[kCpt * t + j] >> 12;
      uint32_t pk = cnt | (tc << 16);  // both counts scanned at once (each < 2^16)
      uint32_t inc = pk;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
      }
      if (lane == 63) sh.wsum[w] = inc;
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (int k = 0; k < NT / 64; ++k) {
        if (k < w) before += sh.wsum[k];
        total += sh.wsum[k];
      }
      // Where does the mesh go?  Into the block the chunk already owns when it fits there; else into a block handed out
      // now -- of the small pool (CV / CT), or of the large one for a mesh beyond that; a mesh without vertices needs none
      // and gives back what the chunk owned (mesh_block_for, tf_devfn.h).  kBlkFail: the pool is exhausted.  Block-uniform.
      if (t == 0) sh.ovf = mesh_block_for(v, sh.rblock, total & 0xFFFFu, total >> 16, sh.rpress);
      __syncthreads();
      const uint32_t ovf_blk = sh.ovf;
      uint16_t* const gvl = (ovf_blk != kBlkFail && (ovf_blk & kBlkLarge)) ? v.ovf_vlist + (size_t)((ovf_blk & ~kBlkLarge) - 1u) * kOvfCV : nullptr;
      const uint32_t excl = before + inc - pk;
      uint32_t r = excl & 0xFFFFu;
      sh.rbase[t] = (uint16_t)r;
      sh.rmask[t] = usedm;
      for (unsigned long long u = usedm; u; u &= u - 1ull) {
        const int m = first + (int)__builtin_ctzll(u);
        if (gvl) { if (r < (uint32_t)kOvfCV) gvl[r] = (uint16_t)m; }
        else if (r < v.mesh_cv) vlist[r] = (uint16_t)m;  // (a mesh with more went to the large pool, or is rejected below)
        ++r;
      }
      uint32_t t0 = excl >> 16;
#pragma unroll
      for (int j = 0; j < kCpt; ++j) {
        sh.toff[kCpt * t + j] = (uint16_t)t0;
        t0 += (uint32_t)sh.cinfo[kCpt * t + j] >> 12;
      }
      if (t == 0) { sh.nv = total & 0xFFFFu; sh.nt = total >> 16; }
    }
    __syncthreads();
    const uint32_t nv = sh.nv, nt = sh.nt;
    const uint32_t ovf = sh.ovf;
    if (ovf == kBlkFail) {  // no block left in the pool the mesh needs: reported, stored empty (the chunk keeps what it had)
      if (t == 0) {
        atomicOr(&v.vctl->status, kStMeshFull);
        const uint32_t was = sh.rstate & kMsInMap;
        rec->nv = 0; rec->nt = 0; rec->state = was | kMsOverflow; rec->epoch = epoch;
        if (rearm >= 0 && was) patch_list_append(v, rearm ^ 1, shard, id, own, sh.rtexloc, sh.rblock);
      }
      __syncthreads();
      continue;
    }
    const uint16_t* const gvlist = (ovf & kBlkLarge) ? v.ovf_vlist + (size_t)((ovf & ~kBlkLarge) - 1u) * kOvfCV : nullptr;

    if (DBG && dbg == 4) continue;  // triage: + ranking
    if (DBG && dbg == 9) mesh_stamp(v, r, 6);
    // a mesh enters allMeshes when it has vertices and stays there afterwards (:260-262).  The patch-list entry does
    // not depend on the vertices: the last wave (it rarely has vertex work) appends it now, so that the round trips
    // of the two counters overlap the vertex pass instead of ending the chunk.
    const uint32_t inmap = (sh.rstate & kMsInMap) | (nv ? kMsInMap : 0u);
    if (t == NT - 64) {
      mesh_block_settle(v, sh.rblock, ovf);  // (what the chunk owned and this generation does not use goes back to its pool)
      if (rearm >= 0 && inmap) patch_list_append(v, rearm ^ 1, shard, id, own, sh.rtexloc, ovf);
    }
    // ---- pass 2: the winning cell of every used slot evaluates the vertex; lane = output vertex
    const float org[3] = {(float)(8 * id.x) * res, (float)(8 * id.y) * res, (float)(8 * id.z) * res};  // Chunk.cpp:52
    uint32_t adj = 0;
    for (uint32_t i = t; i < nv; i += NT) {
      const int m = gvlist ? gvlist[i] : vlist[i];
      const int cell = owner_cell(m, __builtin_ctz((sh.ownq[m >> 3] >> (4 * (m & 7))) & 0xFu));
      const int x = cell & 7, y = (cell >> 3) & 7, z = cell >> 6;
      const int ax = m % 3, b = m / 3;
      const int bx = b % 9, by = (b / 9) % 9, bz = b / 81;
      const int e = edge_of_slot(ax, bx - x, by - y, bz - z);
      const int e0 = edge_c0(e), e1 = edge_c1(e);
      const float s0 = sh.S[ridx(x + cox(e0), y + coy(e0), z + coz(e0))];
      const float s1 = sh.S[ridx(x + cox(e1), y + coy(e1), z + coz(e1))];
      const float tt = s0 / (s0 - s1);  // :757-762
      const int o0[3] = {cox(e0), coy(e0), coz(e0)}, o1[3] = {cox(e1), coy(e1), coz(e1)};
      const int cc[3] = {x, y, z};
      float pv[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float p0 = (float)o0[a] * res, p1 = (float)o1[a] * res;  // cubeCoordOffsets (:635-636)
        const float ev = p0 + tt * (p1 - p0);
        const float origin = org[a] + ((float)cc[a] * res + half);     // Chunk origin + centroids[voxel] (:662)
        pv[a] = ev + origin;                                           // :872-877
      }
      const int k = (fabsf(s0) > fabsf(s1)) ? e1 : e0;
      const int px = x + cox(k), py = y + coy(k), pz = z + coz(k);
      float g[3];
      gradient_at(sh.S, px, py, pz, cox(k), coy(k), coz(k), res, g);
      // voxel colour of that corner (:808-824)
      const uint32_t cs = sh.nslot[((px + 8) >> 3) + ((py + 8) >> 3) * 3 + ((pz + 8) >> 3) * 9];
      const ushort4 c4 = v.color[(size_t)cs * kChunkVoxels + (px & 7) + (py & 7) * 8 + (pz & 7) * 64];
      float col[3] = {1.0f, 1.0f, 1.0f};
      const float cw = (float)c4.w;
      if (cw > 0.0f) {
        col[0] = ((float)c4.x / 255.0f) / cw;
        col[1] = ((float)c4.y / 255.0f) / cw;
        col[2] = ((float)c4.z / 255.0f) / cw;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        mesh_plane(v, ovf, kMpPos + a)[i] = pv[a];
        mesh_plane(v, ovf, kMpNrm + a)[i] = g[a];
        mesh_plane(v, ovf, kMpCol + a)[i] = col[a];
        // Mesh::GetIndice (Mesh.cpp:52-83) with grid_resolution = resolution * (8 / GRID_EACH_DIM)
        const int gp = (int)floorf((pv[a] - org[a]) / (res * 1.0f));
        if (gp >= 8) adj |= 1u << (2 * a + 1);
        if (gp <= 0) adj |= 1u << (2 * a);
      }
    }
    if (adj) atomicOr(&sh.adj, adj);
    if (DBG && dbg == 9) mesh_stamp(v, r, 7);
    // ---- triangles, in cell order; (s2, s1, s0) per triangle (:914-916)
    // (taken from the END of the workgroup: the vertices above keep the first waves busy -- a mesh has 78 of them --
    // while the last ones would otherwise have nothing to do)
    for (uint32_t ci = (uint32_t)(NT - 1 - t); ci < sh.ncell; ci += NT) {
      const int cell = sh.clist[ci];
      const uint32_t info = sh.cinfo[cell];
      if (!(info >> 12)) continue;
      const int x = cell & 7, y = (cell >> 3) & 7, z = cell >> 6;
      const unsigned long long row = d_mc_tri[sh.cidx[cell]];
      const uint32_t valid = info & 0xFFFu;
      uint32_t o = sh.toff[cell];
      for (int col = 0; col < 15; col += 3) {
        const int s0 = (int)((row >> (4 * col)) & 0xF);
        if (s0 == 0xF) break;
        const int s1 = (int)((row >> (4 * col + 4)) & 0xF), s2 = (int)((row >> (4 * col + 8)) & 0xF);
        if (!((valid >> s0) & (valid >> s1) & (valid >> s2) & 1u)) continue;
        constexpr int kEptT = 2304 / NT;
        auto ref_of = [&](int m) {
          const int tt = m / kEptT, j = m - tt * kEptT;
          return (uint16_t)(sh.rbase[tt] + (uint32_t)__popcll(sh.rmask[tt] & ((1ull << j) - 1ull)));
        };
        tri_plane(v, ovf, 0)[o] = ref_of(edge_slot(x, y, z, s2));
        tri_plane(v, ovf, 1)[o] = ref_of(edge_slot(x, y, z, s1));
        tri_plane(v, ovf, 2)[o] = ref_of(edge_slot(x, y, z, s0));
        ++o;
      }
    }
    __syncthreads();
    if (t == 0) {
      // its own adjacency flags are final now (SimplifyByClustering runs once per generation, Chisel.cpp:124)
      rec->nv = (uint16_t)nv; rec->nt = (uint16_t)nt; rec->epoch = epoch; rec->block = ovf;
      rec->state = inmap | (sh.adj << kMsAdjShift) | (inmap ? simplified : 0u);  // (simplified: every mesh of allMeshes, Chisel.cpp:116-126)
    }
    __syncthreads();
    if (DBG && dbg == 9) mesh_stamp(v, r, 8);
  }
}

__global__ __launch_bounds__(256) void k_init_mesh_rec(MeshRec* rec, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    MeshRec r;
    memset(&r, 0, sizeof(r));
    r.texloc = kNoTexloc;
    r.frameid = -1;
    r.ratio[0] = r.ratio[1] = 1.0f;
    rec[i] = r;
  }
}
void launch_init_meshes(const VolumeDev& v, hipStream_t s) {
  hipLaunchKernelGGL(k_init_mesh_rec, dim3(1024), dim3(256), 0, s, v.mesh_rec, v.max_chunks);
  (void)hipMemsetAsync(v.mesh_rec + v.max_chunks, 0, sizeof(uint32_t) * ((size_t)blk_ring_len(v.mesh_blocks) + blk_ring_len(v.ovf_blocks)), s);  // the free rings: vacant (blk_release)
  (void)hipMemsetAsync(v.mesh_cnt, 0, sizeof(uint32_t) * 2 * kMeshCntWords, s);
}

// rows per shard: a chunk's row and patch entry go to shard (pool slot % 32) and pool slots are unique, so a shard
// holds at most ceil(max_chunks / 32) of them (the margin is historical)
uint32_t mesh_shard_rows(uint32_t max_chunks) { return max_chunks / kMeshShards + 258u; }

static void launch_mesher(const VolumeDev& v, int cnt_par, uint32_t max_entries, uint32_t epoch, float res, bool fused,
                          int rearm_set, hipStream_t s) {
  static const uint32_t dbg = getenv("TF_MESH_DBG") ? (uint32_t)atoi(getenv("TF_MESH_DBG")) : 0u;  // triage switch / phase stamps
  uint32_t* surv = v.mesh_nbr;
  uint32_t* cnt = v.mesh_cnt + (size_t)(cnt_par & 1) * kMeshCntWords;
  uint32_t* cnt_next = v.mesh_cnt + (size_t)((cnt_par & 1) ^ 1) * kMeshCntWords;
  const uint32_t cap_sh = mesh_shard_rows(v.max_chunks);
  // workgroup b takes row b of the concatenated shard lists and strides by the grid.  The grid is the RESIDENT capacity
  // (128 threads per chunk: with 12.4 KB of LDS and 80 VGPRs -- TF_MESH_WAVES waves per SIMD -- twelve chunks per CU): every
  // workgroup starts at once and walks rows until the lists end.  (Up to round 5 the cap was 4096: the workgroups beyond the
  // capacity started when the first ones had finished ALL their rows and then walked theirs -- the hall's 12 k survivors
  // took 175.9 us of filter + mesher per frame instead of 157.3, a room frame's 3 k are the same either way.)
  uint32_t grid = ((max_entries + kMeshShards - 1) / kMeshShards + 1) * kMeshShards;
  static const uint32_t gmax = (uint32_t)device_cus() * (uint32_t)(TF_MESH_WAVES * 2);
  if (grid > gmax) grid = gmax;
  if (dbg)
    hipLaunchKernelGGL((k_mesh<128, true>), dim3(grid), dim3(128), v.mesh_cv * sizeof(uint16_t), s, v, surv, cnt, cnt_next, cap_sh, epoch, res,
                       fused ? kMsSimplified : 0u, dbg, rearm_set);
  else
    hipLaunchKernelGGL((k_mesh<128>), dim3(grid), dim3(128), v.mesh_cv * sizeof(uint16_t), s, v, surv, cnt, cnt_next, cap_sh, epoch, res,
                       fused ? kMsSimplified : 0u, 0u, rearm_set);
}

bool launch_mesh(const VolumeDev& v, int cnt_par, const int4* dlist, const uint32_t* dcount, uint32_t max_entries,
                 uint32_t epoch, float res, bool fused, int rearm_set, uint32_t len_guess, uint32_t* len_hint, int shards_par,
                 hipStream_t s, const PatchStage* patch, const Cam* cam, int cls, const KfStoreArgs* store) {
  if (!max_entries) return false;
  uint32_t* cnt = v.mesh_cnt + (size_t)(cnt_par & 1) * kMeshCntWords;
  const uint32_t cap_sh = mesh_shard_rows(v.max_chunks);
  if (max_entries > v.max_chunks) max_entries = v.max_chunks;
  // 2560 workgroups = 1.25 x the wave form's resident capacity: a list of up to 10 k entries runs one entry per wave
  const uint32_t fmax = 2560u;  // (the workgroup-batch form on the hall's 86 k entries: 1024 / 1536 / 2560 / 3072 / 4608 / 6144 workgroups = 59 / 50 / 43 / 42 / 42 / 44 us)
  const uint32_t fgrid = (max_entries + 3) / 4 < fmax ? (max_entries + 3) / 4 : fmax;
  const uint32_t* dslot = fused ? v.work_slot : nullptr;
  const int ppar = fused ? (rearm_set ^ 1) : -1;
  const bool use_summ = true;
  const bool wave_form = len_guess <= 2560u * 4u;  // (the wave form strides when its grid is smaller than the list)
  FilterPatch fp;
  memset(&fp, 0, sizeof(fp));
  fp.cls = (uint32_t)cls;
  if (store) { fp.n_store = 1u; fp.store = *store; }
  if (patch && cam) {
    // one wave per patch; the range is dispatched AHEAD of the filter's workgroups (its chains are the longer ones)
    fp.defer = 1u;
    fp.n_patch = 1024u;
    fp.first = 0u;
    fp.par = patch->par;
    fp.cam = *cam;
    fp.kf = patch->kf;
  }
#define TF_LAUNCH_FILTER(W, P)                                                                                       \
  hipLaunchKernelGGL((k_mesh_filter<W, P>), dim3(fgrid + fp.n_patch + fp.n_store), dim3(256), 0, s, v, dlist, dslot, dcount, max_entries, \
                     epoch, v.mesh_nbr, cnt, cap_sh, ppar, use_summ, len_hint, shards_par, fp)
  static const bool tl = getenv("TF_MESH_DBG") && atoi(getenv("TF_MESH_DBG")) == 10;
  if (wave_form && tl && !fp.n_patch)
    hipLaunchKernelGGL((k_mesh_filter<true, false, true>), dim3(fgrid + fp.n_store), dim3(256), 0, s, v, dlist, dslot, dcount, max_entries,
                       epoch, v.mesh_nbr, cnt, cap_sh, ppar, use_summ, len_hint, shards_par, fp);
  else if (wave_form) { if (fp.n_patch) TF_LAUNCH_FILTER(true, true); else TF_LAUNCH_FILTER(true, false); }
  else { if (fp.n_patch) TF_LAUNCH_FILTER(false, true); else TF_LAUNCH_FILTER(false, false); }
#undef TF_LAUNCH_FILTER
  launch_mesher(v, cnt_par, max_entries, epoch, res, fused, rearm_set, s);
  return fp.n_patch != 0;
}

// ---------------------------------------------------------------------------------------
// Per-frame dirty set of the fused flow = Chisel::meshesToUpdate after ONE FinalizeIntegrateChunks
// (Structure/Chisel.h:192-208): every updated chunk of the frame's list and its six face neighbours, those
// that exist, each once (a stamp per pool slot de-duplicates), appended to the work list.
// ---------------------------------------------------------------------------------------
constexpr uint32_t kDirtyBlocks = 512;  // two 1024-thread workgroups per CU: every block resident
__device__ __forceinline__ void dirty_frame_body(const VolumeDev& v, int par, uint32_t stamp) {
  const SelBuf& L = v.sel;
  const uint32_t nl = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  const uint32_t total = nl * 8u;  // 8 threads per entry: k = 0..6 self + neighbours, 7 idle
  AtlasCtl::Set* S = &v.actl->set[par];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ uint32_t wcnt[16];
  __shared__ uint32_t gbase;
  for (uint32_t b0 = blockIdx.x * 1024; b0 < total; b0 += kDirtyBlocks * 1024) {
    const uint32_t t = b0 + threadIdx.x;
    bool emit = false;
    int4 q = make_int4(0, 0, 0, 0);
    uint32_t slot = kInvalidSlot;
    if (t < total && (t & 7u) != 7u) {
      const uint32_t e = list_phys(v, t >> 3, L.ctl->n_front);  // (fused lists are two-ended, FrameCtl::n_front)
      const int k = (int)(t & 7u);
      const int4 id0 = L.list_id[e];           // (id and slot go out with the flag, not behind it)
      const uint32_t slot0 = L.list_slot[e];
      if (L.list_needs[e]) {
        q = nbr7(id0, k);
        q.w = 0;
        if (k == 0) slot = slot0;
        else slot = hash_slot_alive(v, pack_id(q.x, q.y, q.z));  // key, slot and state of an entry in one 16-byte load
        if (slot != kInvalidSlot && part_owned(v, q.x, q.y, q.z))
          emit = atomicMax(&v.mesh_rec[slot].stamp, stamp) < stamp;
      }
    }
    // one same-address atomic per workgroup of 1024 (they retire at ~10 ns each)
    const unsigned long long m = __ballot(emit);
    if (lane == 0) wcnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (int k = 0; k < 16; ++k) { const uint32_t c = wcnt[k]; wcnt[k] = tot; tot += c; }
      gbase = tot ? atomicAdd(&S->n_work, tot) : 0u;
    }
    __syncthreads();
    if (emit) {
      const uint32_t p = gbase + wcnt[w] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (p < v.max_chunks) { v.work_ids[p] = q; v.work_slot[p] = slot; }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(1024) void k_dirty_frame(VolumeDev v, int par, uint32_t stamp) { dirty_frame_body(v, par, stamp); }
// The keyframe unit: the same launch carries, as one more workgroup, the ordered store of the group's validChunks
// (kf_store_body, tf_kf_store.h) -- both only read the finalized list
__global__ __launch_bounds__(1024) void k_dirty_frame_store(VolumeDev v, int par, uint32_t stamp, KfStoreArgs a) {
  if (blockIdx.x == kDirtyBlocks) { kf_store_body(v, a.tab, a.slots, a.arena, a.cap, a.slot, a.slack, a.fill); return; }
  dirty_frame_body(v, par, stamp);
}
void launch_dirty_frame(const VolumeDev& v, int par, uint32_t stamp, hipStream_t s) {
  hipLaunchKernelGGL(k_dirty_frame, dim3(kDirtyBlocks), dim3(1024), 0, s, v, par, stamp);
}
void launch_dirty_frame_store(const VolumeDev& v, int par, uint32_t stamp, const KfStoreArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_dirty_frame_store, dim3(kDirtyBlocks + 1), dim3(1024), 0, s, v, par, stamp, a);
}

// A fused frame that finds marks of EARLIER frames still waiting for a mesher (frames integrated without the textured
// unit since the last CompressMeshes): Chisel::meshesToUpdate is everything marked since it was last cleared, not
// this frame's chunks alone.  The general dirty list (k_list_dirty) becomes the frame's work list.
__global__ __launch_bounds__(256) void k_adopt_dirty_list(VolumeDev v, int par) {
  uint32_t n = v.vctl->n_tmp;
  if (n > v.max_chunks) n = v.max_chunks;
  if (blockIdx.x == 0 && threadIdx.x == 0) v.actl->set[par].n_work = n;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int4 id = v.work_ids[i];
    const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
    v.work_slot[i] = (ent != kInvalidSlot && (v.hent[ent].alive & 1u)) ? v.hent[ent].slot : kInvalidSlot;
  }
}
void launch_dirty_backlog(const VolumeDev& v, int par, uint32_t clear_floor, hipStream_t s) {
  (void)hipMemsetAsync(&v.vctl->n_tmp, 0, 4, s);
  launch_list_dirty(v, v.work_ids, v.max_chunks, clear_floor, s);
  hipLaunchKernelGGL(k_adopt_dirty_list, dim3(512), dim3(256), 0, s, v, par);
}

// ---------------------------------------------------------------------------------------
// Chisel::CompressMeshes (Structure/Chisel.cpp:112-147) for a list of chunks: mark the meshes
// simplified (their own flags were computed with the mesh), then exchange the flags with the face
// neighbours' meshes: after the pass flag k of a mesh and flag k^1 of its k-th neighbour are the OR
// of the two (the reference's pairwise updates reach the same fixed point in any iteration order).
// Two kernels: the neighbour pass must see every listed mesh already marked simplified.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compress_mark(VolumeDev v, const int4* __restrict__ list,
                                                       const uint32_t* __restrict__ count, uint32_t cap) {
  uint32_t n = *count;
  if (n > cap) n = cap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int4 id = list[i];
    const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
    if (ent == kInvalidSlot || !(v.hent[ent].alive & 1u)) continue;
    const uint32_t slot = v.hent[ent].slot;
    if (slot == kInvalidSlot) continue;
    MeshRec* r = &v.mesh_rec[slot];
    if (r->state & kMsInMap) r->state |= kMsSimplified;
  }
}
// (the fused per-frame flow does this exchange inside the patch kernel, one wave per chunk)
__global__ __launch_bounds__(256) void k_compress_exchange(VolumeDev v, const int4* __restrict__ list,
                                                           const uint32_t* __restrict__ count, uint32_t cap) {
  uint32_t n = *count;
  if (n > cap) n = cap;
  const uint32_t total = n * 8u;  // 8 threads per entry: k = 0..5 neighbours, 6 and 7 idle
  for (uint32_t b0 = blockIdx.x * 256; b0 < total; b0 += gridDim.x * 256) {
    const uint32_t i = b0 + threadIdx.x;
    if (i >= total || (i & 7u) >= 6u) continue;
    const int4 id = list[i >> 3];
    const int k = (int)(i & 7u), m = k ^ 1;
    uint32_t slot = kInvalidSlot;
    const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
    if (ent != kInvalidSlot && (v.hent[ent].alive & 1u)) slot = v.hent[ent].slot;
    if (slot == kInvalidSlot || !(v.mesh_rec[slot].state & kMsInMap)) continue;
    MeshRec* a = &v.mesh_rec[slot];
    int4 q = id;
    if (k == 0) q.x -= 1; else if (k == 1) q.x += 1; else if (k == 2) q.y -= 1;
    else if (k == 3) q.y += 1; else if (k == 4) q.z -= 1; else q.z += 1;
    const uint32_t en = hash_find(v, pack_id(q.x, q.y, q.z));
    if (en == kInvalidSlot || !(v.hent[en].alive & 1u) || v.hent[en].slot == kInvalidSlot) continue;
    MeshRec* b = &v.mesh_rec[v.hent[en].slot];
    const uint32_t bs = b->state;
    if (!(bs & kMsInMap) || !(bs & kMsSimplified)) continue;
    const uint32_t abit = 1u << (kMsAdjShift + k), bbit = 1u << (kMsAdjShift + m);
    const bool fa = (a->state & abit) != 0, fb = (bs & bbit) != 0;
    if (fa && !fb) atomicOr(&b->state, bbit);
    if (!fa && fb) atomicOr(&a->state, abit);
  }
}

void launch_compress(const VolumeDev& v, const int4* list, const uint32_t* count, uint32_t cap, bool mark,
                     hipStream_t s) {
  if (mark) hipLaunchKernelGGL(k_compress_mark, dim3(256), dim3(256), 0, s, v, list, count, cap);
  hipLaunchKernelGGL(k_compress_exchange, dim3(512), dim3(256), 0, s, v, list, count, cap);
}

// keys of allMeshes
__global__ __launch_bounds__(256) void k_list_meshes(VolumeDev v, int4* out, uint32_t cap) {
  const uint32_t nent = v.hmask + 1u;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nent; i += gridDim.x * 256) {
    const HEntry h = v.hent[i];
    if (h.key == kEmptyKey || !(h.alive & 1u) || h.slot == kInvalidSlot) continue;
    if (!(v.mesh_rec[h.slot].state & kMsInMap)) continue;
    const uint32_t p = atomicAdd(&v.vctl->n_tmp, 1u);
    if (p < cap) out[p] = unpack_id(h.key);
  }
}

// per listed chunk: {nv, ni, state, found}
__global__ __launch_bounds__(256) void k_mesh_counts(VolumeDev v, const int4* __restrict__ ids, uint32_t n, int4* out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 id = ids[i];
  const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
  int4 r = make_int4(0, 0, 0, 0);
  if (ent != kInvalidSlot && (v.hent[ent].alive & 1u) && v.hent[ent].slot != kInvalidSlot) {
    const MeshRec m = v.mesh_rec[v.hent[ent].slot];
    if (m.state & kMsInMap) r = make_int4((int)m.nv, (int)(3u * m.nt), (int)m.state, 1);
  }
  out[i] = r;
}

// Mesh::vertices / normals / colors / indices in the reference's layouts (Vec3List = xyz per vertex)
__global__ __launch_bounds__(256) void k_mesh_gather(VolumeDev v, const int4* __restrict__ ids, uint32_t n,
                                                     const long long* __restrict__ voff, const long long* __restrict__ ioff,
                                                     float* verts, float* normals, float* colors, uint32_t* indices) {
  const uint32_t c = blockIdx.x;
  if (c >= n) return;
  const int4 id = ids[c];
  const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
  if (ent == kInvalidSlot || !(v.hent[ent].alive & 1u) || v.hent[ent].slot == kInvalidSlot) return;
  const uint32_t slot = v.hent[ent].slot;
  const MeshRec m = v.mesh_rec[slot];
  if (!(m.state & kMsInMap)) return;
  const long long v0 = voff[c], i0 = ioff[c];
  const uint32_t nv = (uint32_t)(voff[c + 1] - v0) < m.nv ? (uint32_t)(voff[c + 1] - v0) : m.nv;
  const uint32_t nt = (uint32_t)(ioff[c + 1] - i0) / 3u < m.nt ? (uint32_t)(ioff[c + 1] - i0) / 3u : m.nt;
  for (uint32_t i = threadIdx.x; i < nv; i += 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (verts) verts[3 * (v0 + i) + a] = mesh_plane(v, m.block, kMpPos + a)[i];
      if (normals) normals[3 * (v0 + i) + a] = mesh_plane(v, m.block, kMpNrm + a)[i];
      if (colors) colors[3 * (v0 + i) + a] = mesh_plane(v, m.block, kMpCol + a)[i];
    }
  if (indices)
    for (uint32_t i = threadIdx.x; i < nt; i += 256)
#pragma unroll
      for (int a = 0; a < 3; ++a) indices[i0 + 3 * i + a] = tri_plane(v, m.block, a)[i];
}

// ids (host, int32[3n]) -> device int4 list in d_tmp at byte offset `at`
static int upload_ids(tf_volume* v, const int32_t* ids, int64_t n, size_t at) {
  int rc = ensure_pinned(v, (size_t)n * 16 + 16);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  int32_t* hid = reinterpret_cast<int32_t*>(v->h_pinned);
  for (int64_t i = 0; i < n; ++i) {
    hid[4 * i] = ids[3 * i]; hid[4 * i + 1] = ids[3 * i + 1]; hid[4 * i + 2] = ids[3 * i + 2]; hid[4 * i + 3] = 0;
  }
  TF_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t*>(v->d_tmp) + at, hid, (size_t)n * 16, hipMemcpyHostToDevice, v->stream));
  return TF_OK;
}

// the dirty set (Chisel::meshesToUpdate) as a device list in d_tmp: [0,16) count word, ids from byte 16.  Enqueued only:
// the count is also in VolCtl::n_tmp, which sync_status reads behind whatever the caller launches on the list.
static int dirty_list_enqueue(tf_volume* v) {
  const size_t cap = (size_t)v->dev.max_chunks;
  int rc = ensure_tmp(v, cap * 16 + 16);
  if (rc) return rc;
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
  launch_list_dirty(v->dev, reinterpret_cast<int4*>(reinterpret_cast<uint8_t*>(v->d_tmp) + 16), (uint32_t)cap,
                    v->clear_floor, v->stream);
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(v->d_tmp, &v->dev.vctl->n_tmp, 4, hipMemcpyDeviceToDevice, v->stream));
  return TF_OK;
}

// diagnostic (tf_check_summaries): every alive chunk's VolumeDev::summ against the classes of its voxels
__global__ __launch_bounds__(256) void k_check_summaries(VolumeDev v, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t e = (blockIdx.x * 256 + threadIdx.x) >> 6; e <= v.hmask; e += nwaves) {
    const HEntry h = v.hent[e];
    if (h.slot == kInvalidSlot || !(h.alive & 1u)) continue;
    uint32_t w = 0;
    for (int j = 0; j < 8; ++j) {
      const float2 t = v.tsdf[(size_t)h.slot * kChunkVoxels + j * 64 + lane];
      w |= chunk_summary_bits(t.x, t.y, (uint32_t)(j * 64 + lane));
    }
    w = wave_or(w);
    const uint32_t have = v.summ[h.slot];
    if (lane == 0) {
      atomicAdd(&out[0], 1ull);
      if (w & ~have) atomicAdd(&out[1], 1ull);
      if (have & ~w) atomicAdd(&out[2], 1ull);
    }
  }
}

// diagnostic (tf_check_neighbours): every row of the neighbour table against the hash.  out[0] rows of chunks with a pool
// slot, [1] non-zero words, [2] non-zero words that do not name the pool slot the hash holds for that id (must be 0),
// [3] rows whose last check is newer than every key insertion, [4] zero words of such rows whose id the hash does hold
// (must be 0), [5] unused
__global__ __launch_bounds__(256) void k_check_neighbours(VolumeDev v, unsigned long long* out) {
  const int lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * 4;
  const uint32_t create_seq = v.vctl->create_seq;
  for (uint32_t e = (blockIdx.x * 256 + threadIdx.x) >> 6; e <= v.hmask; e += nwaves) {
    const HEntry h = v.hent[e];
    if (h.key == kEmptyKey || h.slot == kInvalidSlot) continue;
    const int4 id = unpack_id(h.key);
    const uint32_t w = lane < kNbrWords ? v.nbr[(size_t)h.slot * kNbrWords + lane] : 0u;
    const bool full = (uint32_t)__shfl((int)w, kNbrStamp) > create_seq;
    uint32_t truth = 0;
    const bool nb = lane < 27 && lane != 13;
    if (nb) truth = nbr_probe(v, pack_id(id.x + lane % 3 - 1, id.y + (lane / 3) % 3 - 1, id.z + lane / 9 - 1));
    const unsigned long long nz = __ballot(nb && w != 0u), bad = __ballot(nb && w != 0u && w != truth);
    const unsigned long long miss = __ballot(nb && full && w == 0u && truth != 0u);
    if (lane == 0) {
      atomicAdd(&out[0], 1ull);
      if (nz) atomicAdd(&out[1], (unsigned long long)__popcll(nz));
      if (bad) atomicAdd(&out[2], (unsigned long long)__popcll(bad));
      if (full) atomicAdd(&out[3], 1ull);
      if (miss) atomicAdd(&out[4], (unsigned long long)__popcll(miss));
    }
  }
}

}  // namespace tf

using namespace tf;

extern "C" {

int tf_update_meshes(tf_volume* v, int64_t* n_meshed) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int rc = dirty_list_enqueue(v);
  if (rc) return rc;
  // (the launches take the list's length from the device word: no synchronisation between the scan and the mesher; an
  // empty list costs two empty launches)
  const uint8_t* db = reinterpret_cast<const uint8_t*>(v->d_tmp);
  prof_begin(v, TF_PROF_MESH);
  (void)nbr_next_seq(v);
  launch_mesh(v->dev, v->mesh_par, reinterpret_cast<const int4*>(db + 16), reinterpret_cast<const uint32_t*>(db), v->dev.max_chunks,
              ++v->mesh_epoch, v->res, false, -1, v->dirty_list_n, nullptr, -1, v->stream);
  v->mesh_par ^= 1;
  prof_end(v);
  TF_HIP(hipGetLastError());
  uint32_t n = 0;
  rc = sync_status(v, &n);
  if (n > v->dev.max_chunks) n = v->dev.max_chunks;
  if (n_meshed) *n_meshed = n;
  if (rc) return rc;
  v->dirty_list_n = n;
  v->dirty_list_seq = v->call_seq;  // d_tmp holds the list: a tf_compress_meshes right behind this call takes it from there
  return TF_OK;
}

int tf_list_meshes(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n) {
  if (!v || !n) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (cap < 0) cap = 0;
  int rc = ensure_tmp(v, (size_t)cap * 16 + 16);
  if (rc) return rc;
  rc = ensure_pinned(v, (size_t)cap * 16 + 16);
  if (rc) return rc;
  TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
  hipLaunchKernelGGL(k_list_meshes, dim3(1024), dim3(256), 0, v->stream, v->dev, reinterpret_cast<int4*>(v->d_tmp),
                     (uint32_t)cap);
  TF_HIP(hipGetLastError());
  uint32_t cnt = 0;
  TF_HIP(hipMemcpyAsync(&cnt, &v->dev.vctl->n_tmp, 4, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  *n = cnt;
  const int64_t m = cnt < (uint64_t)cap ? cnt : cap;
  if (m > 0 && out_ids) {
    TF_HIP(hipMemcpyAsync(v->h_pinned, v->d_tmp, (size_t)m * 16, hipMemcpyDeviceToHost, v->stream));
    TF_HIP(hipStreamSynchronize(v->stream));
    const int32_t* st = reinterpret_cast<const int32_t*>(v->h_pinned);
    for (int64_t i = 0; i < m; ++i) {
      out_ids[3 * i] = st[4 * i]; out_ids[3 * i + 1] = st[4 * i + 1]; out_ids[3 * i + 2] = st[4 * i + 2];
    }
  }
  if ((int64_t)cnt > cap && out_ids) { set_error("output capacity too small"); return TF_ERR_CAPACITY; }
  return TF_OK;
}

int tf_mesh_counts(tf_volume* v, const int32_t* ids, int64_t n, int32_t* n_vertices, int32_t* n_indices,
                   uint8_t* adj, uint8_t* simplified) {
  if (!v || (n > 0 && !ids)) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  int rc = ensure_tmp(v, (size_t)n * 32);
  if (rc) return rc;
  rc = upload_ids(v, ids, n, 0);
  if (rc) return rc;
  uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
  hipLaunchKernelGGL(k_mesh_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v->stream, v->dev,
                     reinterpret_cast<const int4*>(db), (uint32_t)n, reinterpret_cast<int4*>(db + (size_t)n * 16));
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(v->h_pinned, db + (size_t)n * 16, (size_t)n * 16, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  const int32_t* r = reinterpret_cast<const int32_t*>(v->h_pinned);
  for (int64_t i = 0; i < n; ++i) {
    if (!r[4 * i + 3]) {
      set_error("chunk (" + std::to_string(ids[3 * i]) + "," + std::to_string(ids[3 * i + 1]) + "," +
                std::to_string(ids[3 * i + 2]) + ") has no mesh");
      return TF_ERR_MISSING_CHUNK;  // allMeshes.at() would throw
    }
    if (n_vertices) n_vertices[i] = r[4 * i];
    if (n_indices) n_indices[i] = r[4 * i + 1];
    if (adj)
      for (int k = 0; k < 6; ++k) adj[6 * i + k] = (uint8_t)(((uint32_t)r[4 * i + 2] >> (kMsAdjShift + k)) & 1u);
    if (simplified) simplified[i] = (uint8_t)(((uint32_t)r[4 * i + 2] & kMsSimplified) ? 1 : 0);
  }
  return TF_OK;
}

int tf_meshes_download(tf_volume* v, const int32_t* ids, int64_t n, const int64_t* vert_offsets,
                       const int64_t* index_offsets, float* verts, float* normals, float* colors,
                       uint32_t* indices) {
  if (!v || (n > 0 && (!ids || !vert_offsets || !index_offsets))) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  if (n <= 0) return TF_OK;
  const int64_t nv = vert_offsets[n], ni = index_offsets[n];
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
  const size_t o_ids = take((size_t)n * 16), o_vo = take((size_t)(n + 1) * 8), o_io = take((size_t)(n + 1) * 8);
  const size_t o_in_end = o;
  const size_t o_v = take((size_t)nv * 12), o_n = take((size_t)nv * 12), o_c = take((size_t)nv * 12), o_i = take((size_t)ni * 4);
  const size_t total = o;
  int rc = ensure_tmp(v, total);
  if (rc) return rc;
  rc = ensure_pinned(v, total);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hb = reinterpret_cast<uint8_t*>(v->h_pinned);
  uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
  int32_t* hid = reinterpret_cast<int32_t*>(hb + o_ids);
  for (int64_t i = 0; i < n; ++i) {
    hid[4 * i] = ids[3 * i]; hid[4 * i + 1] = ids[3 * i + 1]; hid[4 * i + 2] = ids[3 * i + 2]; hid[4 * i + 3] = 0;
  }
  memcpy(hb + o_vo, vert_offsets, (size_t)(n + 1) * 8);
  memcpy(hb + o_io, index_offsets, (size_t)(n + 1) * 8);
  TF_HIP(hipMemcpyAsync(db, hb, o_in_end, hipMemcpyHostToDevice, v->stream));
  hipLaunchKernelGGL(k_mesh_gather, dim3((unsigned)n), dim3(256), 0, v->stream, v->dev,
                     reinterpret_cast<const int4*>(db + o_ids), (uint32_t)n,
                     reinterpret_cast<const long long*>(db + o_vo), reinterpret_cast<const long long*>(db + o_io),
                     reinterpret_cast<float*>(db + o_v), reinterpret_cast<float*>(db + o_n),
                     reinterpret_cast<float*>(db + o_c), reinterpret_cast<uint32_t*>(db + o_i));
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(hb + o_v, db + o_v, total - o_v, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  if (verts) memcpy(verts, hb + o_v, (size_t)nv * 12);
  if (normals) memcpy(normals, hb + o_n, (size_t)nv * 12);
  if (colors) memcpy(colors, hb + o_c, (size_t)nv * 12);
  if (indices) memcpy(indices, hb + o_i, (size_t)ni * 4);
  return TF_OK;
}

// chunksToUpdate = the dirty keys that have a mesh (GCFusion/MobileFusion.cpp:345-353), written straight into host-visible
// memory (pool slot in w: the host sorts by id)
__global__ __launch_bounds__(256) void k_dirty_with_mesh(VolumeDev v, const int4* __restrict__ ids, const uint32_t* __restrict__ count,
                                                         uint32_t cap, int4* __restrict__ h_out, uint32_t cap_out) {
  uint32_t n = *count;
  if (n > cap) n = cap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int4 id = ids[i];
    const uint32_t ent = hash_find(v, pack_id(id.x, id.y, id.z));
    if (ent == kInvalidSlot || !(v.hent[ent].alive & 1u) || v.hent[ent].slot == kInvalidSlot) continue;
    if (!(v.mesh_rec[v.hent[ent].slot].state & kMsInMap)) continue;
    const uint32_t p = atomicAdd(&v.vctl->n_tmp, 1u);
    if (p < cap_out) h_out[p] = make_int4(id.x, id.y, id.z, 0);
  }
}

int tf_compress_meshes(tf_volume* v, int32_t* out_ids, int64_t cap, int64_t* n_out) {
  if (!v || !n_out) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  *n_out = 0;
  int rc = TF_OK;
  // (the call before this one was tf_update_meshes -- or a keyframe unit without its texture stage, which leaves the list in the
  // same place but, being asynchronous, not its length on the host: dirty_list_n == ~0u)
  const bool have_list = v->dirty_list_seq + 1 == v->call_seq && v->d_tmp;
  if (!have_list) { rc = dirty_list_enqueue(v); if (rc) return rc; }
  const bool known_n = have_list && v->dirty_list_n != ~0u;
  const uint32_t cap_list = v->dev.max_chunks;
  // room for every listed chunk: known when the list is update_meshes', the whole pool otherwise
  const size_t cap_host = known_n ? (size_t)v->dirty_list_n : (size_t)cap_list;
  int64_t m = 0;
  if (cap_host) {
    rc = ensure_pinned(v, cap_host * 16);
    if (rc) return rc;
    uint8_t* db = reinterpret_cast<uint8_t*>(v->d_tmp);
    const int4* list = reinterpret_cast<const int4*>(db + 16);
    const uint32_t* cnt = reinterpret_cast<const uint32_t*>(db);
    launch_compress(v->dev, list, cnt, cap_list, true, v->stream);
    TF_HIP(hipGetLastError());
    TF_HIP(hipMemsetAsync(&v->dev.vctl->n_tmp, 0, 4, v->stream));
    const uint32_t grid = known_n ? (v->dirty_list_n + 255u) / 256u : 1024u;
    hipLaunchKernelGGL(k_dirty_with_mesh, dim3(grid ? grid : 1u), dim3(256), 0, v->stream, v->dev, list, cnt, cap_list,
                       reinterpret_cast<int4*>(v->h_pinned), (uint32_t)cap_host);
    TF_HIP(hipGetLastError());
    uint32_t got = 0;
    rc = sync_status(v, &got);
    if (rc) return rc;
    m = got < cap_host ? got : (int64_t)cap_host;
    // ascending id (std::set<ChunkID> order): one 64-bit key per id, x most significant
    const int32_t* hid = reinterpret_cast<const int32_t*>(v->h_pinned);
    std::vector<unsigned long long> keys((size_t)m);
    bool wide = false;
    for (int64_t i = 0; i < m; ++i) {
      const int32_t x = hid[4 * i], y = hid[4 * i + 1], z = hid[4 * i + 2];
      if (x < -(1 << 20) || x >= (1 << 20) || y < -(1 << 20) || y >= (1 << 20) || z < -(1 << 20) || z >= (1 << 20)) { wide = true; break; }
      keys[(size_t)i] = ((unsigned long long)(uint32_t)(x + (1 << 20)) << 42) | ((unsigned long long)(uint32_t)(y + (1 << 20)) << 21) |
                        (unsigned long long)(uint32_t)(z + (1 << 20));
    }
    if (!wide) {
      std::sort(keys.begin(), keys.end());
      if (out_ids)
        for (int64_t i = 0; i < m && i < cap; ++i) {
          const unsigned long long k = keys[(size_t)i];
          out_ids[3 * i] = (int32_t)(k >> 42) - (1 << 20);
          out_ids[3 * i + 1] = (int32_t)((k >> 21) & 0x1FFFFFu) - (1 << 20);
          out_ids[3 * i + 2] = (int32_t)(k & 0x1FFFFFu) - (1 << 20);
        }
    } else {  // ids beyond 21 bits per axis: compare the triples
      std::vector<const int32_t*> keep((size_t)m);
      for (int64_t i = 0; i < m; ++i) keep[(size_t)i] = hid + 4 * i;
      std::sort(keep.begin(), keep.end(), [](const int32_t* a, const int32_t* b) {
        for (int k = 0; k < 3; ++k)
          if (a[k] != b[k]) return a[k] < b[k];
        return false;
      });
      if (out_ids)
        for (int64_t i = 0; i < m && i < cap; ++i) memcpy(out_ids + 3 * i, keep[(size_t)i], 12);
    }
  }
  *n_out = m;
  v->dirty_list_seq = ~0ull;
  rc = tf_clear_dirty(v);  // chunksToUpdate.clear() (Chisel.cpp:146)
  if (rc) return rc;
  if (out_ids && m > cap) { set_error("output capacity too small"); return TF_ERR_CAPACITY; }
  return TF_OK;
}

int tf_check_summaries(tf_volume* v, int64_t* n_chunks, int64_t* n_missing, int64_t* n_stale) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int rc = ensure_tmp(v, 32);
  if (rc) return rc;
  TF_HIP(hipMemsetAsync(v->d_tmp, 0, 32, v->stream));
  hipLaunchKernelGGL(k_check_summaries, dim3(1024), dim3(256), 0, v->stream, v->dev, reinterpret_cast<unsigned long long*>(v->d_tmp));
  TF_HIP(hipGetLastError());
  unsigned long long h[3] = {0, 0, 0};
  TF_HIP(hipMemcpyAsync(h, v->d_tmp, sizeof(h), hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  if (n_chunks) *n_chunks = (int64_t)h[0];
  if (n_missing) *n_missing = (int64_t)h[1];
  if (n_stale) *n_stale = (int64_t)h[2];
  return TF_OK;
}

int tf_check_neighbours(tf_volume* v, int64_t out6[6]) {
  if (!v || !out6) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  int rc = ensure_tmp(v, 64);
  if (rc) return rc;
  TF_HIP(hipMemsetAsync(v->d_tmp, 0, 64, v->stream));
  hipLaunchKernelGGL(k_check_neighbours, dim3(1024), dim3(256), 0, v->stream, v->dev, reinterpret_cast<unsigned long long*>(v->d_tmp));
  TF_HIP(hipGetLastError());
  unsigned long long h[6] = {0, 0, 0, 0, 0, 0};
  TF_HIP(hipMemcpyAsync(h, v->d_tmp, sizeof(h), hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  for (int k = 0; k < 6; ++k) out6[k] = (int64_t)h[k];
  return TF_OK;
}

}  // extern "C"
