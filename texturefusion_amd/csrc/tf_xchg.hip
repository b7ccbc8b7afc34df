// tf_xchg.hip -- the kernels of the multi-GPU boundary exchange: pack the ghost-band chunks a
// rank updated since the last exchange into [count | records] blocks, store received records as ghost chunks, publish the
// next frame's band counts to the host.  The transport (RCCL, or the caller's own) is tf_comm.cpp / the C ABI.
#include "tf_device.h"
#include "tf_devfn.h"

#pragma clang fp contract(off)

namespace tf {

// ---- multi-GPU boundary exchange ------------------------------------------------------
// Pack every chunk this rank owns whose "touched" bit is set (slab-face chunks updated since they were last packed) and
// clear the bit.  Record: int4 id | float2[512] | ushort4[512].  The flagged chunks are LISTED (VolumeDev::xl_ent, appended
// by the voxel kernels when the bit goes 0 -> 1): one wave per list entry copies the chunk; the cost of a pack follows what
// a frame touched, not the size of the hash.
// BANDS: two blocks instead of one -- `records` takes the chunks the rank BELOW reads as ghosts (key - lo <= a + b + c),
// `records_up` those the rank ABOVE reads (key == hi - 1); a chunk of a thin slab may go to both.  Slabs are contiguous key
// ranges, so with every slab at least a + b + c + 1 keys wide these two neighbours are the only readers (part_band).
template <bool BANDS>
__global__ __launch_bounds__(256) void k_boundary_pack(VolumeDev v, uint8_t* records, uint8_t* records_up, uint32_t cap,
                                                       uint32_t cap_up) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const uint32_t nwaves = gridDim.x * 4;
  const uint32_t par = v.xl_par & 1u;
  const uint32_t* list = v.xl_ent + (size_t)par * v.max_chunks;
  uint32_t* keep = v.xl_ent + (size_t)(par ^ 1u) * v.max_chunks;  // what does not fit stays flagged: listed for the next pack
  uint32_t n = v.vctl->xl_n[par];
  if (n > v.max_chunks) n = v.max_chunks;
  for (uint32_t e = wave; e < n; e += nwaves) {
    const uint32_t i = list[e];
    const HEntry h = v.hent[i];
    if (h.key == kEmptyKey || !(h.alive & 2u) || h.slot == kInvalidSlot) continue;  // (cannot happen: listed <=> flagged)
    const uint32_t slot = h.slot;
    int4 hd = unpack_id(h.key);
    bool down = true, up = false;
    if (BANDS) {
      const long long k = part_key(v, hd.x, hd.y, hd.z);
      down = k >= (long long)v.part_lo && k - (long long)v.part_lo <= (long long)(v.part_a + v.part_b + v.part_c);
      up = k == (long long)v.part_hi - 1;
      if (!down && !up) {  // the partition changed since the chunk was flagged: nobody reads it as a ghost any more
        if (lane == 0) v.hent[i].alive = h.alive & 5u;
        continue;
      }
    }
    uint32_t p = 0, q = 0;
    if (lane == 0) {
      if (down) p = atomicAdd(BANDS ? &v.vctl->xchg_cnt[0] : &v.vctl->n_tmp, 1u);
      if (up) q = atomicAdd(&v.vctl->xchg_cnt[1], 1u);
    }
    p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
    q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    // A side that does not fit is skipped on its own (the other block still gets its record); the chunk stays flagged --
    // and listed -- until every side that wants it has been written: a later exchange with room packs it again (the side
    // that already has it receives an identical or newer copy).
    const bool fit_down = down && p < cap, fit_up = up && q < cap_up;
    if (lane == 0) {
      if ((fit_down || !down) && (fit_up || !up)) v.hent[i].alive = h.alive & 5u;
      else {
        const uint32_t kp = atomicAdd(&v.vctl->xl_n[par ^ 1u], 1u);
        if (kp < v.max_chunks) keep[kp] = i;
      }
    }
    if (!fit_down && !fit_up) continue;
    hd.w = (int)v.mark_epoch[slot];  // header: id + the epoch of the chunk's last update (Chisel::meshesToUpdate travels with it)
    const uint4* st = reinterpret_cast<const uint4*>(v.tsdf + (size_t)slot * kChunkVoxels);
    const uint4* sc = reinterpret_cast<const uint4*>(v.color + (size_t)slot * kChunkVoxels);
    uint4 vt[4], vc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { vt[k] = st[k * 64 + lane]; vc[k] = sc[k * 64 + lane]; }
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      if (side == 0 ? !fit_down : !fit_up) continue;
      uint8_t* rec = (side == 0 ? records : records_up) + (size_t)(side == 0 ? p : q) * (16 + 4096 + 4096);
      if (lane == 0) *reinterpret_cast<int4*>(rec) = hd;
      uint4* dt = reinterpret_cast<uint4*>(rec + 16);
      uint4* dc = reinterpret_cast<uint4*>(rec + 16 + 4096);
#pragma unroll
      for (int k = 0; k < 4; ++k) { dt[k * 64 + lane] = vt[k]; dc[k * 64 + lane] = vc[k]; }
    }
  }
  // The last workgroup through re-arms the list it consumed and (BANDS) writes the two blocks' in-band counts (the first
  // word of the 16-byte header in front of the records), adds what fitted to the running total and re-arms the counters.
  // (Every count above is a RETURNING atomic -- complete at the L2 before its wave goes on --, and the barrier orders the
  // workgroup's waves ahead of its ticket; the records only have to be visible to the next launch.  Round 5's form -- a scan
  // of the whole hash by 1024 workgroups, each with a __threadfence() ahead of its ticket -- took 49 us per frame for ~190
  // records: the "host-bound exchange" of profiles/r5 was this kernel, tools/gaps.py on its kernel trace.)
  __shared__ uint32_t s_last;
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&v.vctl->xchg_ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    v.vctl->xl_n[par] = 0u;
    if (BANDS) {
      const uint32_t na = atomicExch(&v.vctl->xchg_cnt[0], 0u), nb = atomicExch(&v.vctl->xchg_cnt[1], 0u);
      *reinterpret_cast<uint32_t*>(records - 16) = na;
      *reinterpret_cast<uint32_t*>(records_up - 16) = nb;
      v.vctl->xchg_sent += (na < cap ? na : cap) + (nb < cap_up ? nb : cap_up);
    }
    v.vctl->xchg_ticket = 0u;
  }
}
// grid: 64 workgroups = 256 waves, one list entry each per round (a frame of the bench's 24-key slab flags ~190 chunks)
void launch_boundary_pack(const VolumeDev& v, uint8_t* records, uint32_t cap, hipStream_t s) {
  hipLaunchKernelGGL(k_boundary_pack<false>, dim3(64), dim3(256), 0, s, v, records, (uint8_t*)nullptr, cap, 0u);
}
void launch_boundary_pack_bands(const VolumeDev& v, uint8_t* block_down, uint8_t* block_up, uint32_t cap_down,
                                uint32_t cap_up, hipStream_t s) {
  hipLaunchKernelGGL(k_boundary_pack<true>, dim3(64), dim3(256), 0, s, v, block_down + 16, block_up + 16, cap_down, cap_up);
}
// The in-band counts of freshly packed blocks (VolCtl::n_tmp / n_tmp2 -> the first word of each block) and the running
// total of records written (what fitted), for tf_comm_stats_ex.
__global__ void k_boundary_headers(VolumeDev v, uint32_t* hdr_a, uint32_t cap_a, uint32_t* hdr_b, uint32_t cap_b) {
  if (threadIdx.x != 0) return;
  const uint32_t na = v.vctl->n_tmp, nb = hdr_b ? v.vctl->n_tmp2 : 0u;
  *hdr_a = na;
  if (hdr_b) *hdr_b = nb;
  v.vctl->xchg_sent += (na < cap_a ? na : cap_a) + (nb < cap_b ? nb : cap_b);
}
void launch_boundary_headers(const VolumeDev& v, uint32_t* hdr_a, uint32_t cap_a, uint32_t* hdr_b, uint32_t cap_b, hipStream_t s) {
  hipLaunchKernelGGL(k_boundary_headers, dim3(1), dim3(64), 0, s, v, hdr_a, cap_a, hdr_b, cap_b);
}
// FrameCtl::band_cnt of a frame whose selection is through -> host-visible memory, the tag last (the host polls it).
__global__ void k_xchg_publish(const FrameCtl* ctl, uint32_t* host_words, uint32_t tag) {
  if (threadIdx.x != 0) return;
  for (int q = 0; q < 4; ++q) __hip_atomic_store(&host_words[1 + q], ctl->band_cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&host_words[0], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_xchg_publish(const FrameCtl* ctl, uint32_t* host_words, uint32_t tag, hipStream_t s) {
  hipLaunchKernelGGL(k_xchg_publish, dim3(1), dim3(64), 0, s, ctl, host_words, tag);
}

// Store received records of chunks this rank does not own as ghost chunks.
__global__ __launch_bounds__(512) void k_boundary_unpack(VolumeDev v, const uint8_t* records, uint32_t n) {
  __shared__ uint32_t sslot;
  for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
    const uint8_t* rec = records + (size_t)r * (16 + 4096 + 4096);
    const int4 id = *reinterpret_cast<const int4*>(rec);
    const bool owned = part_owned(v, id.x, id.y, id.z);
    if (owned) continue;  // block-uniform
    if (threadIdx.x == 0) {
      bool is_new;
      uint32_t ent;
      sslot = chunk_acquire(v, id, &is_new, &ent);
      // the ghost carries its owner's update epoch: the owned neighbours of this chunk become dirty
      // exactly as they do in a single volume (same frame numbering on every rank)
      if (sslot != kInvalidSlot && (uint32_t)id.w > v.mark_epoch[sslot]) v.mark_epoch[sslot] = (uint32_t)id.w;
    }
    __syncthreads();
    const uint32_t slot = sslot;
    if (slot != kInvalidSlot) {
      const float2 tv = reinterpret_cast<const float2*>(rec + 16)[threadIdx.x];
      v.tsdf[(size_t)slot * kChunkVoxels + threadIdx.x] = tv;
      v.color[(size_t)slot * kChunkVoxels + threadIdx.x] = reinterpret_cast<const ushort4*>(rec + 16 + 4096)[threadIdx.x];
      const uint32_t word = wave_or(chunk_summary_bits(tv.x, tv.y, threadIdx.x));
      if ((threadIdx.x & 63) == 0 && word) atomicOr(&v.summ[slot], word);
    }
    __syncthreads();
  }
}
void launch_boundary_unpack(const VolumeDev& v, const uint8_t* records, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_boundary_unpack, dim3(n < 1024 ? n : 1024), dim3(512), 0, s, v, records, n);
}


// The same over the blocks of an all-gather: block b = [u32 count, 12 B pad | cap records]; the own block is
// skipped.  dirty_par >= 0 (fused textured flow, one exchange per frame): a ghost that arrives was updated on
// its owner's side in this frame, so its owned face neighbours belong to this frame's dirty set
// (Chisel.h:197-203) -- they join the work list, de-duplicated by the per-slot stamp like k_dirty_frame's.
// blocks_b != nullptr: exactly two blocks with their own addresses and capacities (the neighbour form: what came from the
// rank below, cap records, and from the rank above, cap_b records).  pub_ctl != nullptr: this launch also publishes the
// band counts of the NEXT frame's selection (already through: it ran next to this frame's voxel update) for the host.
__global__ __launch_bounds__(512) void k_boundary_unpack_blocks(VolumeDev v, const uint8_t* blocks, int nblocks, int skip,
                                                                uint32_t cap, int dirty_par, uint32_t stamp,
                                                                const uint8_t* blocks_b, uint32_t cap_b,
                                                                const FrameCtl* pub_ctl, uint32_t* pub_words, uint32_t pub_tag) {
  __shared__ uint32_t sslot;
  if (pub_ctl && blockIdx.x == 0 && threadIdx.x == 0) {
    for (int q = 0; q < 4; ++q) __hip_atomic_store(&pub_words[1 + q], pub_ctl->band_cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&pub_words[0], pub_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const size_t block_bytes = 16 + (size_t)cap * (16 + 4096 + 4096);
  for (int b = 0; b < nblocks; ++b) {
    if (b == skip) continue;
    const uint8_t* blk = (blocks_b && b == 1) ? blocks_b : blocks + (size_t)b * block_bytes;
    const uint32_t bcap = (blocks_b && b == 1) ? cap_b : cap;
    uint32_t n = *reinterpret_cast<const uint32_t*>(blk);
    if (n > bcap) {
      if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&v.vctl->status, kStXchgFull);
      n = bcap;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(&v.vctl->xchg_recv, n);
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
      const uint8_t* rec = blk + 16 + (size_t)r * (16 + 4096 + 4096);
      const int4 id = *reinterpret_cast<const int4*>(rec);
      if (part_owned(v, id.x, id.y, id.z)) continue;  // block-uniform
      if (threadIdx.x == 0) {
        bool is_new;
        uint32_t ent;
        sslot = chunk_acquire(v, id, &is_new, &ent);
        if (sslot != kInvalidSlot && (uint32_t)id.w > v.mark_epoch[sslot]) v.mark_epoch[sslot] = (uint32_t)id.w;
      }
      __syncthreads();
      const uint32_t slot = sslot;
      if (slot != kInvalidSlot) {
        const float2 tv = reinterpret_cast<const float2*>(rec + 16)[threadIdx.x];
        v.tsdf[(size_t)slot * kChunkVoxels + threadIdx.x] = tv;
        v.color[(size_t)slot * kChunkVoxels + threadIdx.x] = reinterpret_cast<const ushort4*>(rec + 16 + 4096)[threadIdx.x];
        const uint32_t word = wave_or(chunk_summary_bits(tv.x, tv.y, threadIdx.x));
        if ((threadIdx.x & 63) == 0 && word) atomicOr(&v.summ[slot], word);
      }
      if (dirty_par >= 0 && threadIdx.x >= 1 && threadIdx.x <= 6) {
        int4 q = nbr7(id, (int)threadIdx.x);
        q.w = 0;
        if (part_owned(v, q.x, q.y, q.z)) {
          const uint32_t ent = hash_find(v, pack_id(q.x, q.y, q.z));
          if (ent != kInvalidSlot && (v.hent[ent].alive & 1u) && v.hent[ent].slot != kInvalidSlot) {
            const uint32_t qs = v.hent[ent].slot;
            if (atomicMax(&v.mesh_rec[qs].stamp, stamp) < stamp) {
              const uint32_t p = atomicAdd(&v.actl->set[dirty_par].n_work, 1u);
              if (p < v.max_chunks) { v.work_ids[p] = q; v.work_slot[p] = qs; }
            }
          }
        }
      }
      __syncthreads();
    }
  }
}
void launch_boundary_unpack_blocks(const VolumeDev& v, const uint8_t* blocks, int nblocks, int skip, uint32_t cap,
                                   int dirty_par, uint32_t stamp, hipStream_t s, const uint8_t* blocks_b, uint32_t cap_b,
                                   const FrameCtl* pub_ctl, uint32_t* pub_words, uint32_t pub_tag) {
  hipLaunchKernelGGL(k_boundary_unpack_blocks, dim3(1024), dim3(512), 0, s, v, blocks, nblocks, skip, cap, dirty_par, stamp,
                     blocks_b, cap_b, pub_ctl, pub_words, pub_tag);
}

}  // namespace tf
