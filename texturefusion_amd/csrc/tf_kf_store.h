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
// (one workgroup of NT threads, NT = 1024 or 256: the second form is the one that rides on the mesher's filter launch)
template <int NT = 1024>
__device__ __forceinline__ void kf_store_body(const VolumeDev& v, KfTab* tab, uint32_t slots, int4* arena, uint32_t cap, int slot,
                                              int slack, uint32_t* fill) {
  constexpr int NW = NT / 64;                        // waves
  constexpr int E = 16384 / NT;          // consecutive entries per thread and round (16 or 64)
  constexpr uint32_t kStretch = 16384u;  // entries per round of the workgroup (a room list -- 11 k -- is one round)
  const SelBuf& L = v.sel;
  uint32_t* const t_off = kf_off(tab);
  uint32_t* const t_n = t_off + slots;
  uint32_t* const t_capn = t_n + slots;
  uint32_t* const t_order = t_capn + slots;
  const uint32_t n = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  __shared__ uint32_t wsum[NW];
  __shared__ uint32_t base;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // The flags of a thread's E entries arrive as E / 16 independent 16-byte loads (list_needs is a byte per entry, allocated
  // with 64 bytes of slack) -- per-entry loads were a chain of round trips.  Thread 0 fetches the table's header and the
  // slot's record meanwhile, so that the serial part between the barriers is arithmetic on registers.
  const uint4* const flags4 = reinterpret_cast<const uint4*>(L.list_needs);
  auto nz4 = [](const uint32_t x) -> unsigned long long {
    return ((x & 0xFFu) ? 1ull : 0ull) | ((x & 0xFF00u) ? 2ull : 0ull) | ((x & 0xFF0000u) ? 4ull : 0ull) | ((x & 0xFF000000u) ? 8ull : 0ull);
  };
  auto mask_e = [&](const uint32_t s0) -> unsigned long long {  // bit k: entry s0 + E t + k is flagged
    const uint32_t e0 = s0 + threadIdx.x * (uint32_t)E;
    uint4 f[E / 16];
#pragma unroll
    for (int q = 0; q < E / 16; ++q) f[q] = (e0 + 16u * q < n) ? flags4[(e0 >> 4) + q] : make_uint4(0, 0, 0, 0);
    unsigned long long m = 0ull;
#pragma unroll
    for (int q = 0; q < E / 16; ++q)
      m |= (nz4(f[q].x) | (nz4(f[q].y) << 4) | (nz4(f[q].z) << 8) | (nz4(f[q].w) << 12)) << (16 * q);
    if (e0 < n && n - e0 < (uint32_t)E) m &= (1ull << (n - e0)) - 1ull;
    return e0 < n ? m : 0ull;
  };
  const unsigned long long m0 = mask_e(0u);
  uint32_t cnt = (uint32_t)__popcll(m0);
  for (uint32_t s0 = kStretch; s0 < n; s0 += kStretch) cnt += (uint32_t)__popcll(mask_e(s0));
  KfTab h = {};
  uint32_t capn_s = 0, off_s = 0;
  if (threadIdx.x == 0) { h = *tab; capn_s = t_capn[slot]; off_s = t_off[slot]; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
  if (lane == 0) wsum[w] = cnt;
  __syncthreads();
  __shared__ uint32_t s_tot, s_need, s_compact;
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int k = 0; k < NW; ++k) tot += wsum[k];
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
    for (uint32_t k = t; k < slots; k += (uint32_t)NT) {
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
        for (uint32_t b0 = 0; b0 < len; b0 += (uint32_t)NT) {  // ascending, a block at a time: target <= source, they may overlap
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
  // pass 2: ordered compaction, a stretch of kStretch entries at a time.  Every thread knows the flags of its E consecutive
  // entries; a wave owns 64 E consecutive entries and walks them in E rounds of 64 -- the round's 64 flags are the masks of
  // the 64 / E lanes that own its entries, fetched with lane reads; ids are loaded and stored coalesced, eight rounds in
  // flight at a time (a thread copying its own E entries one by one was E dependent round trips to 64 different lines).
  // The waves' totals go through LDS; every flagged entry is written behind the flagged entries before it.
  __shared__ uint32_t wcnt[NW];
  constexpr int LPR = 64 / E;  // lanes whose masks make up one round
  uint32_t run = 0;  // flagged entries of the stretches before this one (block-uniform)
  for (uint32_t s0 = 0; s0 < n; s0 += kStretch) {
    const unsigned long long m = s0 == 0u ? m0 : mask_e(s0);
    uint32_t c = (uint32_t)__popcll(m);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);  // the wave's total
    __syncthreads();  // (wcnt of the previous stretch has been read)
    if (lane == 0) wcnt[w] = c;
    __syncthreads();
    uint32_t at = base + run, tot = 0;
    for (int k = 0; k < NW; ++k) { if (k < w) at += wcnt[k]; tot += wcnt[k]; }
    const uint32_t wave_e0 = s0 + (uint32_t)w * 64u * (uint32_t)E;
    const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
    if (c) {  // (wave-uniform)
#pragma unroll
      for (int rb = 0; rb < E; rb += 8) {
        unsigned long long mr[8];
        int4 idv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          unsigned long long x = 0ull;
#pragma unroll
          for (int q = 0; q < LPR; ++q) {
            const int src = (rb + r) * LPR + q;
            const unsigned long long part = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)m_lo, src) |
                                            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)m_hi, src) << 32);
            x |= (E == 64 ? part : (part & ((1ull << (E & 63)) - 1ull))) << ((E * q) & 63);
          }
          mr[r] = x;
          const uint32_t e = wave_e0 + 64u * (uint32_t)(rb + r) + (uint32_t)lane;
          idv[r] = ((x >> lane) & 1ull) ? L.list_id[e] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if ((mr[r] >> lane) & 1ull) arena[at + (uint32_t)__popcll(mr[r] & ((1ull << lane) - 1ull))] = idv[r];
          at += (uint32_t)__popcll(mr[r]);
        }
      }
    }
    run += tot;
  }
}


}  // namespace tf
