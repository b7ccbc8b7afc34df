// tf_kf_store.h -- Frame::validChunks of the keyframes on the device (the keyframe unit, tf_unit.hip): the region table and
// the ordered store of a finalized list, as a device function so that it can run as a kernel of its own (k_kf_store) and as
// one extra workgroup of the dirty-set launch that precedes it (k_dirty_frame_store, tf_mesh.hip: a single-workgroup
// kernel costs 11 us of a keyframe's 235 at its launch floor).
#pragma once

#include "tf_devfn.h"
#include "tf_device.h"

namespace tf {

// The table of regions has one slot per keyframe and GROWS (the reference sizes its keyframe database for 20 000 frames,
// main.cpp:81, GCSLAM/GCSLAM.h:24-26): header + four arrays of `slots` words behind it.  The arena grows too: the store
// kernel leaves {top, largest list, live words} in host-visible memory, and a call that finds the live regions above
// three quarters of the arena (with room for the lists still in flight) doubles it -- one stream synchronisation and one
// device copy per doubling; below that the device-side compaction reclaims dead regions on its own.
constexpr uint32_t kUnitSlots0 = 1024;  // initial slots (doubles on demand)
struct KfTab {
  uint32_t top;  // ids handed out so far
  uint32_t n_compact, n_reuse, n_regions;  // statistics: compactions, stores into an existing region, regions handed out
  uint32_t max_tot;  // longest list stored so far
  uint32_t live;     // sum of the live regions' sizes
  uint32_t pad[2];
  // uint32_t off[slots], n[slots], capn[slots] (size of the slot's region, 0: none), order[slots] (compaction scratch)
};
__host__ __device__ inline uint32_t* kf_off(KfTab* t) { return reinterpret_cast<uint32_t*>(t + 1); }
__host__ __device__ inline const uint32_t* kf_off(const KfTab* t) { return reinterpret_cast<const uint32_t*>(t + 1); }
__host__ inline size_t kf_tab_bytes(uint32_t slots) { return sizeof(KfTab) + (size_t)4 * slots * sizeof(uint32_t); }

// FinalizeIntegrateChunks' validChunks (Chisel.h:192-208): the entries of the current list whose needsUpdate flag is set,
// in list order, appended to the arena.  One workgroup: the order must be kept.
struct KfStoreArgs {
  KfTab* tab;
  uint32_t slots;
  int4* arena;
  uint32_t cap;
  int slot, slack;
  uint32_t* fill;
};
// (1024 threads)
__device__ __forceinline__ void kf_store_body(const VolumeDev& v, KfTab* tab, uint32_t slots, int4* arena, uint32_t cap, int slot,
                                              int slack, uint32_t* fill) {
  const SelBuf& L = v.sel;
  uint32_t* const t_off = kf_off(tab);
  uint32_t* const t_n = t_off + slots;
  uint32_t* const t_capn = t_n + slots;
  uint32_t* const t_order = t_capn + slots;
  const uint32_t n = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t base;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // The flags of the first 16 384 entries as ballots (wave w: the 1024 consecutive entries from 1024 w, sixteen coalesced
  // rounds) -- for a list that short (a room frame has 11 k) these ARE the count, and the compaction below reuses them;
  // a longer list is counted with a strided pass.  Thread 0 fetches the table's header and the slot's record meanwhile,
  // so that the serial part between the barriers is arithmetic on registers (it was a chain of ten dependent loads).
  unsigned long long masks0[16];
  uint32_t mine0 = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t e = (uint32_t)w * 1024u + (uint32_t)r * 64u + (uint32_t)lane;
    masks0[r] = __ballot(e < n && L.list_needs[e] != 0);
    mine0 += (uint32_t)__popcll(masks0[r]);
  }
  KfTab h = {};
  uint32_t capn_s = 0, off_s = 0;
  if (threadIdx.x == 0) { h = *tab; capn_s = t_capn[slot]; off_s = t_off[slot]; }
  uint32_t cnt = mine0;  // (wave-uniform)
  if (n > 16384u) {
    cnt = 0;
    for (uint32_t e = threadIdx.x; e < n; e += 1024) cnt += L.list_needs[e] ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
  }
  if (lane == 0) wsum[w] = cnt;
  __syncthreads();
  __shared__ uint32_t s_tot, s_need, s_compact;
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int k = 0; k < 16; ++k) tot += wsum[k];
    s_tot = tot;
    s_compact = 0;
    if (tot > h.max_tot) h.max_tot = tot;
    if (tot <= capn_s) {  // fits the region the keyframe already has
      s_need = 0;
      h.n_reuse += 1;
    } else {
      uint32_t need = tot + (slack ? tot / 4u + 64u : 0u);
      h.live -= capn_s;
      capn_s = 0;               // the old region (if any) is dead space from here on
      if (h.top + need > cap) {  // (rare) the compaction below works on the table in memory
        s_compact = 1;
        *tab = h;
        t_capn[slot] = 0;
        t_n[slot] = 0;
      }
      s_need = need;
    }
  }
  __syncthreads();
  if (s_compact) {
    // Move the live regions together, in the order they lie in the arena (a slot's thread counts the live regions
    // below its own -- quadratic in the number of slots, but a compaction is rare --; the moves go downwards one
    // region after the other, each by the whole workgroup).
    __shared__ uint32_t s_live;
    const uint32_t t = threadIdx.x;
    if (t == 0) s_live = 0;
    __syncthreads();
    for (uint32_t k = t; k < slots; k += 1024u) {
      if (!t_capn[k]) continue;
      const uint32_t my_off = t_off[k];
      uint32_t below = 0;
      for (uint32_t j = 0; j < slots; ++j) below += (t_capn[j] && t_off[j] < my_off) ? 1u : 0u;
      t_order[below] = k;
      atomicAdd(&s_live, 1u);
    }
    __syncthreads();
    uint32_t to = 0;
    for (uint32_t r = 0; r < s_live; ++r) {
      const uint32_t k = t_order[r];
      const uint32_t from = t_off[k], len = t_n[k], oldcap = t_capn[k];
      const uint32_t roomy = len + (slack ? len / 4u + 64u : 0u);
      const uint32_t newcap = roomy < oldcap ? roomy : oldcap;
      __syncthreads();  // (every thread has read the region's old record)
      if (from != to) {
        for (uint32_t b0 = 0; b0 < len; b0 += 1024u) {  // ascending, a block at a time: target <= source, they may overlap
          int4 val = make_int4(0, 0, 0, 0);
          if (b0 + t < len) val = arena[from + b0 + t];
          __syncthreads();
          if (b0 + t < len) arena[to + b0 + t] = val;
          __syncthreads();
        }
      }
      if (t == 0) { t_off[k] = to; t_capn[k] = newcap; }
      to += newcap;
      __syncthreads();
    }
    if (t == 0) { tab->top = to; tab->live = to; tab->n_compact += 1; h = *tab; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t tot = s_tot;
    uint32_t need = s_need;
    bool ok = true;
    if (need) {
      if (h.top + need > cap) need = tot;  // no slack left: an exact fit
      if (h.top + need > cap) { atomicOr(&v.vctl->status, kStListFull); ok = false; capn_s = 0; }
      else { off_s = h.top; capn_s = need; h.top += need; h.live += need; h.n_regions += 1; }
    }
    *tab = h;
    t_off[slot] = off_s;
    t_capn[slot] = capn_s;
    t_n[slot] = ok ? tot : 0u;
    base = (ok && tot) ? off_s : 0xFFFFFFFFu;
    if (fill) {  // what the host sizes the arena by (no synchronisation: whatever it reads is at most a few calls old)
      __hip_atomic_store(&fill[1], h.max_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&fill[2], h.live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&fill[0], h.top, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (base == 0xFFFFFFFFu) return;
  // pass 2: ordered compaction in stretches of 16 384 entries -- wave w takes the 1024 consecutive entries
  // [s0 + 1024 w, s0 + 1024 (w + 1)) as 16 coalesced rounds of 64 whose ballots stay in registers, ONE scan over the 16
  // wave counts places the waves, and every flagged entry is written behind the flagged entries before it.  (Rounds of
  // 1024 entries with three barriers each took 13.9 us per keyframe of the room stream, profiles/r4: a list of 11 k
  // entries is one stretch here.)
  __shared__ uint32_t wcnt[16];
  uint32_t run = 0;  // flagged entries of the stretches before this one (block-uniform)
  for (uint32_t s0 = 0; s0 < n; s0 += 16384u) {
    const uint32_t w0 = s0 + (uint32_t)w * 1024u;
    unsigned long long masks[16];
    uint32_t mine = mine0;
    if (s0 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) masks[r] = masks0[r];
    } else {
      mine = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t e = w0 + (uint32_t)r * 64u + (uint32_t)lane;
        masks[r] = __ballot(e < n && L.list_needs[e] != 0);
        mine += (uint32_t)__popcll(masks[r]);
      }
    }
    __syncthreads();  // (wcnt of the previous stretch has been read)
    if (lane == 0) wcnt[w] = mine;
    __syncthreads();
    uint32_t before = run, tot = 0;
    for (int k = 0; k < 16; ++k) { if (k < w) before += wcnt[k]; tot += wcnt[k]; }
    uint32_t at = base + before;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned long long m = masks[r];
      if ((m >> lane) & 1ull) arena[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = L.list_id[w0 + (uint32_t)r * 64u + (uint32_t)lane];
      at += (uint32_t)__popcll(m);
    }
    run += tot;
  }
}


}  // namespace tf
