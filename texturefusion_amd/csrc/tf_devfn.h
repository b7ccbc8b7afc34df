// tf_devfn.h -- device-side helpers shared by the kernel files: chunk-id packing, the chunk hash
// (lookup / find-or-create), the multi-GPU ownership key.
#pragma once

#include "tf_device.h"

namespace tf {

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
// multi-GPU ownership key of a chunk id (VolumeDev::part_*)
__device__ __forceinline__ int part_key(const VolumeDev& v, int x, int y, int z) {
  return v.part_a * x + v.part_b * y + v.part_c * z;
}
__device__ __forceinline__ bool part_owned(const VolumeDev& v, int x, int y, int z) {
  const int k = part_key(v, x, y, z);
  return k >= v.part_lo && k < v.part_hi;
}
// Chunks of this slab another rank reads as ghosts.  The mesher of a chunk c reads c + {0,1}^3
// (Structure/ChunkManager.cpp:618-632) and the six face neighbours of each of those (gradients,
// :288-315): key offsets -1 .. (a + b + c) + 1.  So the rank below needs this slab's keys
// lo .. lo + (a + b + c), the rank above needs key hi - 1 (which also covers the +-1 layer the
// dirty-mark closure of Chisel.h:197-203 looks at).
__device__ __forceinline__ bool part_band(const VolumeDev& v, int x, int y, int z) {
  const long long k = part_key(v, x, y, z);
  const long long s = v.part_a + v.part_b + v.part_c;
  return (k >= (long long)v.part_lo && k - (long long)v.part_lo <= s) || k == (long long)v.part_hi - 1;
}

// A chunk whose whole 27-chunk neighbourhood (what its mesh reads: keys within a + b + c of its own) is owned by this
// rank: its mesh does not depend on the boundary exchange of the frame.  Everything else this rank meshes is a
// "boundary" chunk and has to wait for the ghosts.
__device__ __forceinline__ bool part_interior(const VolumeDev& v, int x, int y, int z) {
  const long long k = part_key(v, x, y, z);
  const long long s = (long long)v.part_a + v.part_b + v.part_c;
  return k - s >= (long long)v.part_lo && k + s < (long long)v.part_hi;
}

// Fibonacci hashing folded to 32 bits: every bit of (x, y, z) reaches the index bits (the upper
// half of the product carries z, the lower half x and y), so columns of chunks do not share a home.
__device__ __forceinline__ uint32_t hash_key(unsigned long long k) {
  const unsigned long long h = k * 0x9E3779B97F4A7C15ull;
  return (uint32_t)(h >> 32) ^ (uint32_t)h;
}


// Lookup only.  Entries are never removed, so the probe sequence of a present key is stable.
// Returns the entry index or kInvalidSlot.
__device__ __forceinline__ uint32_t hash_find(const VolumeDev& v, unsigned long long key) {
  uint32_t i = hash_key(key) & v.hmask;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    const unsigned long long cur = v.hent[i].key;
    if (cur == key) return i;
    if (cur == kEmptyKey) return kInvalidSlot;
    i = (i + 1) & v.hmask;
  }
  return kInvalidSlot;
}

// Lookup of a chunk that must be alive (bit 0): its pool slot, or kInvalidSlot.  One 16-byte load per probe brings key,
// slot and state together (hash_find followed by reads of the entry's members is one more dependent round trip).
__device__ __forceinline__ uint32_t hash_slot_alive(const VolumeDev& v, unsigned long long key) {
  uint32_t i = hash_key(key) & v.hmask;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    const uint4 e = *reinterpret_cast<const uint4*>(&v.hent[i]);  // {key lo, key hi, slot, alive}
    const unsigned long long cur = ((unsigned long long)e.y << 32) | e.x;
    if (cur == key) return (e.w & 1u) ? e.z : kInvalidSlot;
    if (cur == kEmptyKey) return kInvalidSlot;
    i = (i + 1) & v.hmask;
  }
  return kInvalidSlot;
}

// ... the same, also returning the entry index
__device__ __forceinline__ uint32_t hash_slot_alive_ent(const VolumeDev& v, unsigned long long key, uint32_t* ent) {
  uint32_t i = hash_key(key) & v.hmask;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    const uint4 e = *reinterpret_cast<const uint4*>(&v.hent[i]);  // {key lo, key hi, slot, alive}
    const unsigned long long cur = ((unsigned long long)e.y << 32) | e.x;
    if (cur == key) { *ent = i; return (e.w & 1u) ? e.z : kInvalidSlot; }
    if (cur == kEmptyKey) return kInvalidSlot;
    i = (i + 1) & v.hmask;
  }
  return kInvalidSlot;
}

// Find or create the pool slot of a chunk id (general path: probing, insertion, revival).
// Within one launch every key is unique (the visible list has no duplicates), so the payload of
// a freshly inserted key is only read by later launches.  *is_new = chunk did not exist (absent
// or parked); *ent = hash entry index.
// lazy: a parked chunk is NOT revived here (it still counts as new); whoever finalizes the list revives it if it was updated
// -- so that a dirty-set claim running beside that finalize never sees a chunk alive that is about to be parked again.
__device__ __forceinline__ uint32_t chunk_acquire(const VolumeDev& v, int4 id, bool* is_new,
                                               uint32_t* ent, const bool lazy = false) {
  const unsigned long long key = pack_id(id.x, id.y, id.z);
  uint32_t i = hash_key(key) & v.hmask;
  *is_new = true;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    unsigned long long cur = v.hent[i].key;
    if (cur == kEmptyKey) {
      cur = atomicCAS(&v.hent[i].key, kEmptyKey, key);
      if (cur == kEmptyKey) {  // inserted: allocate a fresh slot (storage is in the fresh state)
        *ent = i;
        v.vctl->create_seq = v.seq;  // "no chunk there" words of the neighbour table checked before this launch are void
        const uint32_t stripe = (hash_key(key) >> 7) & (kSlotStripes - 1);
        const uint32_t per = v.max_chunks / kSlotStripes;
        const uint32_t k = atomicAdd(&v.vctl->slot_cnt[stripe], 1u);
        const uint32_t slot = stripe * per + k;
        if (k >= per) {
          atomicOr(&v.vctl->status, kStPoolFull);
          v.hent[i].slot = kInvalidSlot;
          v.hent[i].alive = 0;
          return kInvalidSlot;
        }
        v.hent[i].slot = slot;
        v.hent[i].alive = 1;
        return slot;
      }
    }
    if (cur == key) {
      *ent = i;
      const uint32_t slot = v.hent[i].slot;
      if (slot == kInvalidSlot) return slot;
      if (!v.hent[i].alive) { if (!lazy) v.hent[i].alive = 1; }  // parked chunk: revive, still "new"
      else *is_new = false;
      return slot;
    }
    i = (i + 1) & v.hmask;
  }
  atomicOr(&v.vctl->status, kStHashFull);
  *ent = 0;
  return kInvalidSlot;
}

// ---- neighbour table (VolumeDev::nbr) ------------------------------------------------------------------------------
// word of a row for the chunk at id + (dx, dy, dz)
__device__ __forceinline__ int nbr_word(int dx, int dy, int dz) { return (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1); }
// what a row stores for a key: pool slot + 1 of the chunk if the hash holds it with a pool slot (alive or parked), else 0.
// One 16-byte load per probe.  An entry that is being inserted by another wave of the same launch (key there, slot not
// yet) reads as "none" and is not cached.
__device__ __forceinline__ uint32_t nbr_probe(const VolumeDev& v, unsigned long long key) {
  uint32_t i = hash_key(key) & v.hmask;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    const uint4 e = *reinterpret_cast<const uint4*>(&v.hent[i]);  // {key lo, key hi, slot, alive}
    const unsigned long long cur = ((unsigned long long)e.y << 32) | e.x;
    if (cur == key) return e.z + 1u;  // (kInvalidSlot + 1 = 0: a chunk the pool had no slot for)
    if (cur == kEmptyKey) return 0u;
    i = (i + 1) & v.hmask;
  }
  return 0u;
}
// The row of pool slot `own` (chunk `id`), lane j < kNbrWords holding word j, with every "none" word made trustworthy: a row whose check is older than the newest key
// insertion re-probes its zero words, writes back what it finds and stamps the check (the caller's launch carries a seq
// above every insertion ahead of it on the stream, launch_mesh).  `w` = the lane's word as loaded (the caller issues the
// load, next to whatever else it wants in flight); create_seq = VolCtl::create_seq as this launch found it.  Wave-uniform
// control flow when the lanes of a wave share one row.
__device__ __forceinline__ uint32_t nbr_row_checked(const VolumeDev& v, const uint32_t own, const int4 id, const int lane,
                                                   uint32_t w, const uint32_t create_seq) {
  const uint32_t st = (uint32_t)__shfl((int)w, kNbrStamp);
  if (!(st > create_seq)) {
    if (lane < 27 && lane != 13 && w == 0u) {
      w = nbr_probe(v, pack_id(id.x + lane % 3 - 1, id.y + (lane / 3) % 3 - 1, id.z + lane / 9 - 1));
      if (w) v.nbr[(size_t)own * kNbrWords + lane] = w;
    }
    if (lane == kNbrStamp) v.nbr[(size_t)own * kNbrWords + lane] = v.seq;
  }
  return w;
}

// ---- the mesh store's blocks (MeshRec::block) ---------------------------------------------------------------------
// A chunk owns at most one block; the mesher (and tf_mesh_upload) decide per generation where the mesh goes: into the
// block the chunk owns when it fits there, else into one handed out now -- a released one first, then the pool's bump
// counter.  A mesh that came out EMPTY gives its block back (the reference's Mesh::Clear() frees the vectors; a surface
// that moves through a volume would otherwise keep one block per chunk it ever touched), and so does a small block
// whose mesh outgrew it.  Rings of block words (0 = vacant) behind the records, each a power of two long (blk_ring_len):
// small, then large.
// Which block a mesh sits in is not observable.  All of it runs on ONE thread of the workgroup that meshes the chunk.
constexpr uint32_t kBlkFail = 0x7FFFFFFFu;  // no block left in the pool the mesh needs
__device__ __forceinline__ uint32_t blk_ring_mask(const VolumeDev& v, const int large) {
  // (opaque scalars: the masks are loop-invariant in the mesher, and the compiler would otherwise keep them in registers --
  // spilled ones -- across the whole chunk loop for the sake of this rare path)
  uint32_t ns = v.mesh_blocks, nl = v.ovf_blocks;
  asm volatile("" : "+s"(ns), "+s"(nl));
  const uint32_t ms = ns > 2u ? (0xFFFFFFFFu >> __builtin_clz(ns - 1u)) : 1u;  // = blk_ring_len(blocks) - 1
  const uint32_t ml = nl > 2u ? (0xFFFFFFFFu >> __builtin_clz(nl - 1u)) : 1u;
  return large ? ml : ms;
}
__device__ __forceinline__ uint32_t* blk_ring(const VolumeDev& v, const int large) {
  uint32_t* const r = reinterpret_cast<uint32_t*>(v.mesh_rec + v.max_chunks);
  return large ? r + (blk_ring_mask(v, 0) + 1u) : r;
}
__device__ __forceinline__ void blk_release(const VolumeDev& v, const uint32_t b) {  // b != kBlkNone
  const int large = (b & kBlkLarge) ? 1 : 0;
  const uint32_t p = atomicAdd(&v.vctl->blk_tail[large], 1u);
  // (at most `blocks` blocks exist and each is released once while it is free: the ring position is vacant)
  __hip_atomic_store(&blk_ring(v, large)[p & blk_ring_mask(v, large)], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t blk_take(const VolumeDev& v, const int large) {  // a released block, or kBlkNone
  uint32_t* const head = &v.vctl->blk_head[large];
  uint32_t h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    const uint32_t tl = __hip_atomic_load(&v.vctl->blk_tail[large], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int32_t)(tl - h) <= 0) return kBlkNone;
    const uint32_t old = atomicCAS(head, h, h + 1u);
    if (old == h) break;
    h = old;
  }
  uint32_t* const e = &blk_ring(v, large)[h & blk_ring_mask(v, large)];
  // the ticket is ours; its releaser (another workgroup of this launch, at worst) may still be between its ticket and its
  // store -- a few instructions.  Bounded: giving up loses the block until the ring comes round, never correctness.
  for (int spin = 0; spin < (1 << 14); ++spin) {
    const uint32_t b = atomicExch(e, 0u);
    if (b) return b;
  }
  return kBlkNone;
}
// the block for a mesh of nv vertices / nt triangles of a chunk that owns `b`: kBlkNone for an empty mesh (under pressure,
// else the chunk keeps its block), kBlkFail when the
// pool it needs is exhausted (the chunk then keeps what it had).  The caller writes the mesh and then gives back what the
// chunk no longer uses: mesh_block_settle.
// Blocks only go back once their pool has come under pressure -- half of it handed out: a stream whose pool is large
// enough (the default gives every pool slot a block) never pays the ring's atomics, and a chunk whose surface flickers
// keeps its block.  Fresh blocks come from the bump counter while it lasts (one atomic, as before); the ring serves the rest.
// (VolCtl::blk_recycle -- bit 0: the small pool is under pressure, bit 1: the large one; read early, next to the chunk's
// record, so that the decision does not wait for it)
__device__ __forceinline__ uint32_t blk_pressure(const VolumeDev& v) { return v.vctl->blk_recycle; }
__device__ __forceinline__ bool blk_pressed(const uint32_t press, const uint32_t b) { return (press >> ((b & kBlkLarge) ? 1 : 0)) & 1u; }
__device__ __forceinline__ uint32_t blk_fresh(const VolumeDev& v, const int large) {  // kBlkFail: none left
  uint32_t* const next = large ? &v.vctl->ovf_next : &v.vctl->blk_next;
  const uint32_t n = large ? v.ovf_blocks : v.mesh_blocks;
  const uint32_t p = atomicAdd(next, 1u);
  if (p + 1u == ((n + 1u) >> 1)) atomicOr(&v.vctl->blk_recycle, large ? 2u : 1u);  // half of the pool is out: recycle from now on
  if (p < n) return (p + 1u) | (large ? kBlkLarge : 0u);
  atomicSub(next, 1u);  // (nobody waits for it: the counter stays at about n instead of running towards a wrap-around)
  const uint32_t b = n ? blk_take(v, large) : kBlkNone;
  return b != kBlkNone ? b : kBlkFail;
}
__device__ __forceinline__ uint32_t mesh_block_for(const VolumeDev& v, uint32_t b, const uint32_t nv, const uint32_t nt, const uint32_t press) {
  const bool big = nv > v.mesh_cv || nt > v.mesh_ct;
  if (nv == 0u && nt == 0u) {
    if (b != kBlkNone && blk_pressed(press, b)) b = kBlkNone;  // nothing to store
  } else if (big && !(b & kBlkLarge)) {
    b = blk_fresh(v, 1);
  } else if (!big && b == kBlkNone) {
    b = blk_fresh(v, 0);
  }
  return b;
}

// `was`: the block the chunk owned before this generation, `now`: where the generation went (not kBlkFail)
__device__ __forceinline__ void mesh_block_settle(const VolumeDev& v, const uint32_t was, const uint32_t now) {
  if (was != kBlkNone && was != now) blk_release(v, was);  // an emptied mesh's block, or the small block a mesh outgrew
}

// multi-GPU: an owned ghost-band chunk was updated -- flag it "touched" (HEntry::alive bit 1) and, if it was not flagged
// yet, list it for the next boundary pack (VolumeDev::xl_ent).  One thread.
__device__ __forceinline__ void mark_touched(const VolumeDev& v, const uint32_t ent) {
  const uint32_t old = atomicOr(&v.hent[ent].alive, 3u);
  if (!(old & 2u)) {
    const uint32_t p = atomicAdd(&v.vctl->xl_n[v.xl_par & 1u], 1u);
    if (p < v.max_chunks) v.xl_ent[(size_t)(v.xl_par & 1u) * v.max_chunks + p] = ent;  // (every flagged chunk is listed once: p < the chunks there are)
  }
}

// class of one voxel for the mesher's filter (ChunkManager.cpp:669-722, :776-777): observed (sdf <= 1), observed and
// positive, negative, weight above the vertex threshold
__device__ __forceinline__ uint32_t classify_voxel(const float sdf, const float w) {
  const bool ok = !(sdf > 1.0f);
  return (ok ? 1u : 0u) | ((ok && sdf > 0.0f) ? 2u : 0u) | ((sdf < 0.0f) ? 4u : 0u) | ((w > 50.0f) ? 8u : 0u);
}
// VolumeDev::summ contribution of the voxel with in-chunk index k = x + 8 y + 64 z
__device__ __forceinline__ uint32_t chunk_summary_bits(const float sdf, const float w, uint32_t k) {
  const uint32_t c = classify_voxel(sdf, w);
  return c | ((k & 7u) ? 0u : c << 4) | ((k & 56u) ? 0u : c << 8) | ((k & 448u) ? 0u : c << 12);
}
__device__ __forceinline__ uint32_t wave_or(uint32_t x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x |= (uint32_t)__shfl_xor((int)x, o);
  return x;
}


// id and its six face neighbours (Chisel.h:197-203): k = 0 self, 1 -x, 2 +x, 3 -y, 4 +y, 5 -z, 6 +z
__device__ __forceinline__ int4 nbr7(const int4 c, int k) {
  int4 r = c;
  if (k == 1) r.x -= 1; else if (k == 2) r.x += 1;
  else if (k == 3) r.y -= 1; else if (k == 4) r.y += 1;
  else if (k == 5) r.z -= 1; else if (k == 6) r.z += 1;
  return r;
}
// physical position of logical list entry e of a (possibly two-ended) visible list, see FrameCtl::n_front
__device__ __forceinline__ uint32_t list_phys(const VolumeDev& v, uint32_t e, uint32_t n_front) {
  return e < n_front ? e : v.max_list - 1u - (e - n_front);
}

}  // namespace tf
