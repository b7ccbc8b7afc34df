// tf_voxel_math.h -- device helpers shared by the voxel kernels (tf_kernels.hip: K-A, selection, finalize) and the keyframe
// group kernels (tf_group.hip): float keys, the truncation model, per-chunk records, centroid tables, the exact-division
// sequence of the projection, the dirty-set claim, Chunk::observations' table.
#pragma once

#include "tf_devfn.h"
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

// Per-chunk scalars of voxelUpdateSIMD (ProjectionIntegrator.cpp:74-101, Chunk.cpp:52) for one list
// entry.  Computed lane-per-entry where the list is produced (64 chunks per wave instruction
// instead of one redundant copy per lane inside k_integrate) and read back through scalar loads.
struct ChunkPre {
  float4 a;  // o.x, o.y, o.z (origin in camera), truncation
  float4 b;  // weight / (2 * truncation) (unsigned; the de-integration sign is applied in K-A), upper band
};
__device__ __forceinline__ ChunkPre chunk_pre(const int4 id, const float* __restrict__ Pp, const Integ& ig,
                                              float res, float resDiag) {
  float dvec[3];
  dvec[0] = (float)(8 * id.x) * res - Pp[3];
  dvec[1] = (float)(8 * id.y) * res - Pp[7];
  dvec[2] = (float)(8 * id.z) * res - Pp[11];
  float o[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float q0 = Pp[a] * dvec[0], q1 = Pp[4 + a] * dvec[1], q2 = Pp[8 + a] * dvec[2];
    const float s12 = q1 + q2;
    o[a] = q0 + s12;
  }
  const float trunc = truncation(ig, o[2]);
  ChunkPre r;
  r.a = make_float4(o[0], o[1], o[2], trunc);
  r.b = make_float4(ig.weight / (2.0f * trunc), trunc + resDiag, 0.0f, 0.0f);
  return r;
}


// Centroid table of a frame (Chisel::bufferIntegratorSIMDCentroids, Structure/Chisel.cpp:52-110):
// c[a][i] = (R^T (x,y,z))_a * res + res/2, i = (z*8+y)*8+x, summed p0 + (p1 + p2); a function of the
// pose only.  Written once per frame by one workgroup (ahead of K-A), read by every K-A workgroup.
__device__ __forceinline__ void centroid_table(const float* __restrict__ Pp, float res, float* __restrict__ cen) {
  const float half = res * 0.5f;
  for (int i = threadIdx.x; i < kChunkVoxels; i += 256) {
    const float fx = (float)(i & 7), fy = (float)((i >> 3) & 7), fz = (float)(i >> 6);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float q0 = Pp[a] * fx, q1 = Pp[4 + a] * fy, q2 = Pp[8 + a] * fz;
      const float s12 = q1 + q2;
      const float d = q0 + s12;
      cen[a * kChunkVoxels + i] = d * res + half;
    }
  }
}


// The same behind k_select<EMIT> with SelectConsts::plain: the list is the unordered one the selection appended (all of it
// from the front), its length still sits in the append counter -- this launch turns it into a finished plain list
// (n_list = n_front = length, bounding-box keys re-armed as k_scan does) while it resolves the slots.
__device__ __forceinline__ void acquire_emitted_body(const VolumeDev& v, const uint32_t bid, const uint32_t nb, const bool lazy = false) {
  const SelBuf& L = v.sel;
  const unsigned long long pk = L.ctl->emit_pack;
  uint32_t n = (uint32_t)pk;
  if ((pk >> 32) != 0ull || n > v.max_list) n = 0;  // (a list that did not fit was reported by the selection: kStListFull)
  if (bid == 0 && threadIdx.x == 0) {
    L.ctl->n_list = n;
    L.ctl->n_front = n;
    for (int a = 0; a < 3; ++a) { L.ctl->bbox_key[a] = f2key(1e8f); L.ctl->bbox_key[3 + a] = f2key(-1e8f); }
  }
  for (uint32_t e = bid * 256 + threadIdx.x; e < n; e += nb * 256) {
    const int4 id = L.list_id[e];
    bool is_new = false;
    uint32_t ent = 0;
    const uint32_t slot = chunk_acquire(v, id, &is_new, &ent, lazy);
    L.list_slot[e] = slot;
    L.list_ent[e] = ent;
    L.list_new[e] = is_new ? 1 : 0;
    L.list_needs[e] = 0;
  }
}

__device__ __forceinline__ unsigned long long nonzero_bytes(unsigned long long m) {
  unsigned long long t = m | (m >> 1);
  t |= t >> 2;
  t |= t >> 4;
  return t & 0x0101010101010101ull;
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kOOB = 0x7FFFFFF0;  // buffer byte offset that is out of range for every descriptor


// _mm256_cvtps_epi32 for the predicates that consume it: round-to-nearest-even; NaN -> INT_MIN
// (v_med3 returns the minimum when an operand is NaN); |x| >= 2^31 saturates, which every
// consumer (`valid`, `out of observation`) classifies exactly like x86's 0x80000000.
__device__ __forceinline__ int cvt_sat_rne(float x) {
  return (int)rintf(__builtin_amdgcn_fmed3f(x, -2147483648.0f, 2147483520.0f));
}

// The same for operands known not to be NaN: v_cvt_i32_f32 saturates by itself (a C++ cast of an
// out-of-range float would be undefined, hence the instruction is named explicitly).
__device__ __forceinline__ int cvt_rne_hw(float x) {
  int r;
  const float n = rintf(x);
  asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(n));
  return r;
}

// IEEE-correct f32 quotients with a shared denominator.  This is the instruction sequence hipcc
// emits for `a / b` (v_rcp, two FMA refinements of the reciprocal, product, three residual FMAs)
// without v_div_scale / v_div_fmas' scaling / v_div_fixup, which only act on operands or quotients
// outside the normal exponent range, zeros, infinities and NaNs.  `safe` (wave-uniform) tells
// whether every lane is inside that range; otherwise the generic division is used, so results are
// bit-identical to `/` in all cases.  Sharing the reciprocal saves one quarter-rate v_rcp_f32 and
// two FMAs per voxel in the projection (two quotients over p.z).
struct Recip { float d, r; };
__device__ __forceinline__ Recip recip_refined(float d) {
  Recip R;
  R.d = d;
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  R.r = __builtin_fmaf(e, r0, r0);
  return R;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// the same sequence for two numerators at once on the packed-f32 pipe (v_pk_mul / v_pk_fma)
__device__ __forceinline__ f32x2 div2_by(const f32x2 n, const Recip& R) {
  const f32x2 d = {-R.d, -R.d}, r = {R.r, R.r};
  const f32x2 q0 = n * r;
  const f32x2 e0 = __builtin_elementwise_fma(d, q0, n);
  const f32x2 q1 = __builtin_elementwise_fma(e0, r, q0);
  const f32x2 e1 = __builtin_elementwise_fma(d, q1, n);
  return __builtin_elementwise_fma(e1, r, q1);
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// The dirty-set claim of one updated chunk (Chisel.h:197-203: the chunk and its six face neighbours, those that exist): lane k
// < 7 of the wave looks neighbour k up, the per-slot stamp de-duplicates, the winner appends {id, entry, slot} to the shard
// list of its pool slot (counter set `par`).
__device__ __forceinline__ void claim_dirty7(const VolumeDev& v, const int4 id, const uint32_t slot, const uint32_t ent, const int lane,
                                             const uint32_t stamp, const int par) {
  uint32_t cs = kInvalidSlot, ce = 0;
  int4 q = id;
  if (lane < 7) {
    q = nbr7(id, lane);
    if (lane == 0) { cs = slot; ce = ent; }
    else if (part_owned(v, q.x, q.y, q.z)) cs = hash_slot_alive_ent(v, pack_id(q.x, q.y, q.z), &ce);
    if (cs != kInvalidSlot && !(atomicMax(&v.mesh_rec[cs].stamp, stamp) < stamp)) cs = kInvalidSlot;
    if (cs != kInvalidSlot) {
      const uint32_t rows = v.max_chunks / kMeshShards + 258u;  // = mesh_shard_rows()
      const uint32_t sh = cs & (kMeshShards - 1u);
      const uint32_t p = atomicAdd(&v.wl_cnt[((par & 1) * kMeshShards + sh) * 16], 1u);
      if (p < rows) {
        const size_t at = ((size_t)(par & 1) * kMeshShards + sh) * rows + p;
        v.wl_ids[at] = make_int4(q.x, q.y, q.z, (int)(ce + 1u));
        v.wl_slot[at] = cs;
      } else {
        atomicOr(&v.vctl->status, kStMeshFull);
      }
    }
  }
}

// (Chunk::observations on the device: the table is described with k_obs_record below)
__device__ __forceinline__ unsigned long long obs_pack(uint32_t slot, int32_t kf) {
  return ((unsigned long long)slot << 32) | (unsigned long long)(uint32_t)kf;
}
__device__ __forceinline__ uint32_t obs_find(const VolumeDev& v, unsigned long long key, bool insert) {
  uint32_t i = hash_key(key) & v.obs_mask;
  for (uint32_t probe = 0; probe <= v.obs_mask; ++probe) {
    unsigned long long cur = v.obs_key[i];
    if (cur == kEmptyKey) {
      if (!insert) return kInvalidSlot;
      cur = atomicCAS(&v.obs_key[i], kEmptyKey, key);
      if (cur == kEmptyKey) return i;
    }
    if (cur == key) return i;
    i = (i + 1) & v.obs_mask;
  }
  if (insert) atomicOr(&v.vctl->status, kStHashFull);
  return kInvalidSlot;
}

}  // namespace tf
