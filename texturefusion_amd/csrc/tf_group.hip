// tf_group.hip -- the keyframe group kernels (GCFusion/MobileFusion.cpp:165-217, TSDFFusion: the keyframe's depth + colour,
// then up to six local frames depth-only over the same chunk list): the records launch (k_pre_group) and the one visit per
// chunk (k_integrate_group), with what the keyframe unit folds into them (tf_unit.hip).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <type_traits>

#include "tf_device.h"
#include "tf_devfn.h"
#include "tf_host_math.h"
#include "tf_voxel_math.h"

#pragma clang fp contract(off)

namespace tf {

// ---------------------------------------------------------------------------------------
// The local frames of a keyframe group in ONE visit per chunk (GCFusion/MobileFusion.cpp:187-203: after the
// keyframe's own depth + colour, up to six depth-only frames are integrated over the SAME chunk list, each with
// its own pose).  Frame by frame that is six launches that each read and rewrite the same voxel rows; here a wave
// loads the chunk's 512 {sdf, weight} pairs once, applies the frames in order while they stay in registers -- the
// arithmetic of integrate_body<COLOR = false>, operation for operation, including the row-granular rewrite of lanes
// with weight 0 and the pos-stall of a fully off-image row -- and writes back the rows any frame rewrote.
// ---------------------------------------------------------------------------------------
constexpr int kGroupMax = 6;
struct GroupArgs {
  const float* depth[kGroupMax];   // device depth images
  const float4* pre[kGroupMax];    // per-frame list records (k_pre)
  const float* cen[kGroupMax];     // per-frame centroid tables
  int n;
};

struct GroupPoses {
  Pose P[kGroupMax];
};
// list records and centroid tables of all frames of a group in one launch (blockIdx.y = frame)
// (with_key: blockIdx.y == 0 is the KEYFRAME -- pose `key`, records into the selection set like k_pre's -- and the local
// frames follow at y = 1 + f: the keyframe unit computes all seven frames' records in one launch)
// (acquire: the list is the plain one k_select<EMIT> just appended, its length still in the append counter; the LAST row of
// blocks is acquire_emitted_body's work (tf_voxel_math.h) -- slots, isNew, the finished list header -- which the record rows do not read.
// acquire == 2: parked chunks stay parked, chunk_acquire's lazy form -- the list's finalize is k_integrate_group's)
__global__ __launch_bounds__(256) void k_pre_group(VolumeDev v, GroupPoses gp, Integ ig, float res, float resDiag,
                                                   float4* pre_scratch, float* cen_scratch, Pose key, int with_key, int acquire,
                                                   uint32_t* clear_word) {
  const SelBuf& L = v.sel;
  if (clear_word && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *clear_word = 0u;  // (the unit: kf.validChunks.clear(), :217)
  if (acquire && blockIdx.y == gridDim.y - 1) { acquire_emitted_body(v, blockIdx.x, gridDim.x, acquire == 2); return; }
  const bool is_key = with_key && blockIdx.y == 0;
  const int f = is_key ? 0 : (int)blockIdx.y - (with_key ? 1 : 0);
  const float* P = is_key ? key.p : gp.P[f].p;
  float4* pre = is_key ? L.list_pre : pre_scratch + (size_t)f * 4 * v.max_list;
  if (blockIdx.x == 0) centroid_table(P, res, is_key ? L.cen : cen_scratch + (size_t)f * 3 * kChunkVoxels);
  uint32_t n = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  if (acquire) {
    const unsigned long long pk = L.ctl->emit_pack;
    n = ((pk >> 32) != 0ull || (uint32_t)pk > v.max_list) ? 0u : (uint32_t)pk;
  }
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int4 id = L.list_id[e];
    const ChunkPre cp = chunk_pre(id, P, ig, res, resDiag);
    pre[4 * e] = make_float4(cp.a.x, cp.a.y, cp.a.z, cp.b.x);
    pre[4 * e + 1] = make_float4(cp.b.y, __int_as_float(id.x), __int_as_float(id.y), __int_as_float(id.z));
  }
}

// obs_kf >= 0: the launch also records chunk->observations[obs_kf] of the KEYFRAME's integration that ran just before it
// (Chisel.h:244-247: the list's quality sums and needsUpdate flags as that call left them -- read here before this
// kernel touches the entry's flag), instead of a launch of its own between the two (k_obs_record: 4.7 us of the unit)
// fin != 0: the launch is also the list's finalize (k_finalize: FinalizeIntegrateChunks + GarbageCollect, Chisel.h:192-208,
// :472-477, with epoch fin_epoch) -- each wave finishes its entry behind its last frame, when the entry's needsUpdate flag
// (the keyframe's call | this visit) is final, as K-A does in a stream (7 us of launch per group less); claim_par >= 0: and the
// dirty-set pass over the list (k_dirty_frame), into the shard lists of that parity
// KEY: the visit starts with the KEYFRAME's own depth + colour pass over the entry (k_integrate<COLOR, no quality image, FLAG>,
// operation for operation: ProjectionIntegrator.cpp:74-341 with the colour band of :202-304) -- its TSDF rows stay in the
// registers the local frames then work on, instead of a launch of its own that writes them and this one reading them
// back (31 us of a keyframe's 211).  Its centroids are computed per lane (centroid_table's expression; the workgroup's LDS
// holds the six local frames' tables), its records are the list's own (k_pre_group's keyframe row).
struct GroupKey {
  FrameImages img;  // the keyframe's depth + colour
  Pose P;           // its pose (centroids)
};
// QUAL (with KEY): the keyframe has a quality image -- observationQualitySum of the keyframe's pass is kept in row order as
// integrate_body<COLOR, QUALITY> keeps it (:212-238): reset to the out-of-observation constant by a processed row with an
// off-image lane, plus the eight qualities of a row that updates colour
template <bool FLAG, bool KEY, bool QUAL = false>
__global__ __launch_bounds__(256) void k_integrate_group(VolumeDev v, GroupArgs ga, Cam cam, IntegrateConsts kc, int32_t obs_kf,
                                                         int fin, uint32_t fin_epoch, int claim_par, uint32_t claim_stamp,
                                                         GroupKey key) {
  const SelBuf& L = v.sel;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const uint32_t nwaves = gridDim.x * 4;
  const uint32_t n = L.ctl->n_list <= v.max_list ? L.ctl->n_list : 0u;
  const int vy = lane >> 3;
  const int W = cam.W, H = cam.H;
  __shared__ float cenT[kGroupMax][3][kChunkVoxels];  // 6 KB per frame
  for (int f = 0; f < ga.n; ++f) {
    const float4* src = reinterpret_cast<const float4*>(ga.cen[f]);
    float4* dst = reinterpret_cast<float4*>(&cenT[f][0][0]);
    for (int i = threadIdx.x; i < 3 * kChunkVoxels / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const uint32_t row_lo = lane < 32 ? (0xFFu << (lane & 24)) : 0u;
  const uint32_t row_hi = lane < 32 ? 0u : (0xFFu << (lane & 24));
  auto row_any = [&](const unsigned long long m) -> bool {
    return ((((uint32_t)m) & row_lo) | (((uint32_t)(m >> 32)) & row_hi)) != 0u;
  };
  auto ballot = [](const bool b) -> unsigned long long { return __builtin_amdgcn_ballot_w64(b); };
  const f32x2 fxy = {cam.fxi, cam.fyi}, cxy = {kc.cxs, kc.cys};

  for (uint32_t e = wave; e < n; e += nwaves) {
    const int4 id = L.list_id[e];
    if (!part_owned(v, id.x, id.y, id.z)) continue;
    const uint32_t slot = L.list_slot[e];
    if (slot == kInvalidSlot) continue;
    if (!KEY && obs_kf >= 0 && lane == 0) {
      const float q = L.list_quality[e];
      if (q > 0.0f && L.list_needs[e]) {
        const uint32_t at = obs_find(v, obs_pack(slot, obs_kf), true);
        if (at != kInvalidSlot) v.obs_q[at] = q;
      }
    }
    const __amdgpu_buffer_rsrc_t rs_T =
        __builtin_amdgcn_make_buffer_rsrc((void*)(v.tsdf + (size_t)slot * kChunkVoxels), 0, 4096, 0x00020000);
    u32x2 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = __builtin_amdgcn_raw_buffer_load_b64(rs_T, (j * 64 + lane) * 8, 0, 0);
    uint32_t dirty_rows = 0;   // bit j: the lane's row of slice j was rewritten by some frame
    uint32_t rows_total = 0;
    uint32_t key_rows_t = 0, key_rows_c = 0;  // rows the keyframe's pass rewrote (wave-uniform)
    if (KEY) {
      const float4 r0 = L.list_pre[4 * e], r1 = L.list_pre[4 * e + 1];
      const float o0 = r0.x, o1 = r0.y, o2 = r0.z, pbx = r0.w, upper = r1.x;
      const f32x2 o01 = {o0, o1};
      const float wD = FLAG ? pbx : -pbx;
      const float band = 32.0f * kc.res;
      const bool div_safe = (fabsf(o2) > band) && (fabsf(o2) < 1048576.0f) && (fabsf(o0) < 1048576.0f) &&
                            (fabsf(o1) < 1048576.0f) && (kc.res > 1e-6f) && (kc.res < 16.0f);
      const __amdgpu_buffer_rsrc_t rs_depth =
          __builtin_amdgcn_make_buffer_rsrc((void*)key.img.depth, 0, W * H * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_rgba =
          __builtin_amdgcn_make_buffer_rsrc((void*)key.img.rgba, 0, W * H * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_C =
          __builtin_amdgcn_make_buffer_rsrc((void*)(v.color + (size_t)slot * kChunkVoxels), 0, 4096, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_qual =
          __builtin_amdgcn_make_buffer_rsrc((void*)key.img.quality, 0, QUAL ? W * H * 4 : 0, 0x00020000);
      // centroid of the lane's voxel of slice j, axis a: centroid_table's expression
      const float khalf = kc.res * 0.5f;
      const float fx = (float)(lane & 7), fy = (float)vy;
      float qx[3], qy[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) { qx[a] = key.P.p[a] * fx; qy[a] = key.P.p[4 + a] * fy; }
      auto cen = [&](const int a, const int j) -> float {
        const float q2 = key.P.p[8 + a] * (float)j;
        const float s12 = qy[a] + q2;
        const float d = qx[a] + s12;
        return d * kc.res + khalf;
      };
      int off_d[8];
      unsigned long long vm[8];
      uint32_t R = 64, oob_bits = 0;
      int oob_any = 0;
      auto geometry = [&](auto safe_tag) {
        constexpr bool SAFE = decltype(safe_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x2 pxy = o01 + (f32x2){cen(0, j), cen(1, j)};
          const float pzv = o2 + cen(2, j);
          f32x2 q;
          if (SAFE) {
            q = div2_by(pxy, recip_refined(pzv));
          } else {
            q.x = pxy.x / pzv;
            q.y = pxy.y / pzv;
          }
          const f32x2 uw = q * fxy + cxy;
          const int X = SAFE ? cvt_rne_hw(uw.x) : cvt_sat_rne(uw.x);
          const int Y = SAFE ? cvt_rne_hw(uw.y) : cvt_sat_rne(uw.y);
          const bool valid = ((unsigned)(X - 1) < (unsigned)(W - 2)) && ((unsigned)(Y - 1) < (unsigned)(H - 2));
          vm[j] = ballot(valid);
          int od = (__mul24(Y, W) + X) * 4;
          asm volatile("" : "+v"(od));
          off_d[j] = valid ? od : kOOB;
          // X < 0 || X > W-1 || Y < 0 || Y > H-1 (:212-220); implies !valid
          oob_bits |= (((unsigned)X > (unsigned)(W - 1)) || ((unsigned)Y > (unsigned)(H - 1))) ? (1u << j) : 0u;
        }
      };
      if (div_safe) geometry(std::true_type{});
      else geometry(std::false_type{});
      unsigned long long all_valid = ~0ull;
#pragma unroll
      for (int j = 0; j < 8; ++j) all_valid &= vm[j];
      if (all_valid != ~0ull) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (R == 64u) {
            const unsigned long long dead = nonzero_bytes(vm[j]) ^ 0x0101010101010101ull;
            if (dead) R = (uint32_t)(j * 8) + ((uint32_t)__builtin_ctzll(dead) >> 3);
          }
          const bool live_lane = (uint32_t)(j * 8 + vy) < R;
          oob_any |= (live_lane && ((oob_bits >> j) & 1u)) ? 1 : 0;  // off-image lanes of processed rows only
          off_d[j] = live_lane ? off_d[j] : kOOB;
        }
      }
      float dep[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        dep[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_depth, off_d[j], 0, 0));
      // TSDF rows, in the registers the local frames continue with
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((uint32_t)(j * 8) >= R) break;  // a stalled row ends the chunk
        const float d = dep[j];
        const float sd = d - (o2 + cen(2, j));
        const bool act = (uint32_t)(j * 8 + vy) < R;
        const bool dv = (d > cam.nearP) && (cam.farP > d);
        const bool inside = (sd > kc.lower) && (upper > sd);
        const bool F = act && dv && inside;
        const float nw = F ? wD : 0.0f;
        const bool rf_l = row_any(ballot(F));
        key_rows_t += (uint32_t)__popcll(ballot(rf_l)) >> 3;
        if (rf_l) {
          const float ts = __uint_as_float(t[j].x), tw = __uint_as_float(t[j].y);
          const float num = ts * tw + sd * nw;
          const float den = (tw + nw) + kc.sigma;
          const float ns = num / den;
          const float nwt = tw + nw;
          const bool keep = nwt > 0.5f;
          t[j].x = __float_as_uint(keep ? ns : 999.0f);
          t[j].y = __float_as_uint(keep ? nwt : 0.0f);
          dirty_rows |= 1u << j;
        }
      }
      // colour band (-thr < sd < thr, :202-208): which colour rows are rewritten, which pixels feed them -- four slices at
      // a time (eight would hold 40 registers across the loads)
      uint32_t lanes_c = 0;
      float qsum_rows = 0.0f;                       // QUAL: observationQualitySum, row by row
      const bool some_oob = all_valid != ~0ull;     // (a chunk that projects inside the image has no off-image lane)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int off_c[4], off_i[4];
        unsigned long long any_c = 0ull;
        unsigned long long mo[QUAL ? 4 : 1];        // QUAL: off-image lanes of the processed rows of the four slices
        unsigned long long any_o = 0ull;
        if (QUAL) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = h * 4 + jj;
            mo[jj] = some_oob ? ballot(((uint32_t)(j * 8 + vy) < R) && ((oob_bits >> j) & 1u)) : 0ull;
            any_o |= mo[jj];
          }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = h * 4 + jj;
          const float sd = dep[j] - (o2 + cen(2, j));
          const bool upd = (off_d[j] != kOOB) && (fabsf(sd) < kc.thrCol);  // (rows behind a stalled row: off_d is kOOB)
          off_i[jj] = upd ? off_d[j] : kOOB;
          const bool ru_l = row_any(ballot(upd));
          const unsigned long long ru = ballot(ru_l);
          off_c[jj] = ru_l ? (j * 64 + lane) * 8 : kOOB;
          lanes_c += (uint32_t)__popcll(ru);
          any_c |= ru;
        }
        if (any_c == 0ull) {
          if (QUAL && any_o) {  // no colour row, but rows with off-image lanes: the sum starts over at the constant
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              if (mo[jj]) qsum_rows = kc.qoob;
          }
          continue;
        }
        u32x2 c[4];
        uint32_t in[4];
        float qv[QUAL ? 4 : 1];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          c[jj] = __builtin_amdgcn_raw_buffer_load_b64(rs_C, off_c[jj], 0, 0);
          in[jj] = __builtin_amdgcn_raw_buffer_load_b32(rs_rgba, off_i[jj], 0, 0);
          if (QUAL) qv[jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_qual, off_i[jj], 0, 0));
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          // colour planes are 4 x u16 {r, g, b, count} per voxel = two packed-u16 dwords (integrate_body, phase 5a)
          const uint32_t p_rg = __builtin_amdgcn_perm(0u, in[jj], 0x0c010c00u);
          const uint32_t p_ba = __builtin_amdgcn_perm(0u, in[jj], 0x0c030c02u);
          const uint32_t w_rg = c[jj].x, w_ba = c[jj].y;
          const u16x2 in_rg = __builtin_bit_cast(u16x2, p_rg), in_ba = __builtin_bit_cast(u16x2, p_ba);
          u16x2 c_rg = __builtin_bit_cast(u16x2, w_rg), c_ba = __builtin_bit_cast(u16x2, w_ba);
          if (FLAG) {  // (:274-292)
            c_rg += in_rg;
            c_ba += in_ba;
            const bool halve = (int)__builtin_bit_cast(uint32_t, c_ba) >= (121 << 16);
            const u16x2 h_rg = c_rg >> (unsigned short)2, h_ba = c_ba >> (unsigned short)2;
            c_rg = halve ? h_rg : c_rg;
            c_ba = halve ? h_ba : c_ba;
          } else {     // (:293-304)
            c_rg -= in_rg;
            c_ba -= in_ba;
          }
          c[jj].x = __builtin_bit_cast(uint32_t, c_rg);
          c[jj].y = __builtin_bit_cast(uint32_t, c_ba);
          __builtin_amdgcn_raw_buffer_store_b64(c[jj], rs_C, off_c[jj], 0, 0);
        }
        if (QUAL) {  // observationQualitySum in row order (integrate_body, phase 6)
          const int rowshift = lane & 56;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = h * 4 + jj;
            const unsigned long long mu = ballot(off_i[jj] != kOOB);
            if ((mu | mo[jj]) == 0ull) continue;
            float rowsum = 0.0f;
            if (mu) {  // sum += observationQuality[i], i = 0..7 (:233-236)
#pragma unroll
              for (int l = 0; l < 8; ++l) rowsum += __shfl(qv[jj], rowshift + l);
            }
            const int left = (int)R - j * 8;
            const int rmax = left < 8 ? left : 8;
            for (int r = 0; r < rmax; ++r) {
              if ((mo[jj] >> (8 * r)) & 0xFFull) qsum_rows = kc.qoob;
              if ((mu >> (8 * r)) & 0xFFull)
                qsum_rows += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rowsum), 8 * r));
            }
          }
        }
      }
      key_rows_c = lanes_c >> 3;
      // without a quality image nothing is ever added to observationQualitySum: it ends as the out-of-observation
      // constant iff any processed row had an off-image lane (:221-222)
      const float qsum = QUAL ? qsum_rows : ((ballot(oob_any != 0) != 0ull) ? kc.qoob : 0.0f);
      if (lane == 0) {
        L.list_quality[e] = qsum;
        if (obs_kf >= 0 && qsum > 0.0f && key_rows_t != 0u) {  // chunk->observations[keyframe] (Chisel.h:244-247)
          const uint32_t at = obs_find(v, obs_pack(slot, obs_kf), true);
          if (at != kInvalidSlot) v.obs_q[at] = qsum;
        }
      }
    }
    for (int f = 0; f < ga.n; ++f) {
      const float4 r0 = ga.pre[f][4 * e], r1 = ga.pre[f][4 * e + 1];
      const float o0 = r0.x, o1 = r0.y, o2 = r0.z, pbx = r0.w, pby = r1.x;
      const f32x2 o01 = {o0, o1};
      const float wD = FLAG ? pbx : -pbx;
      const float upper = pby;
      const float band = 32.0f * kc.res;
      const bool div_safe = (fabsf(o2) > band) && (fabsf(o2) < 1048576.0f) && (fabsf(o0) < 1048576.0f) &&
                            (fabsf(o1) < 1048576.0f) && (kc.res > 1e-6f) && (kc.res < 16.0f);
      const __amdgpu_buffer_rsrc_t rs_depth =
          __builtin_amdgcn_make_buffer_rsrc((void*)ga.depth[f], 0, W * H * 4, 0x00020000);
      int off_d[8];
      unsigned long long vm[8];
      uint32_t R = 64;
      auto geometry = [&](auto safe_tag) {
        constexpr bool SAFE = decltype(safe_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = j * 64 + lane;
          const f32x2 pxy = o01 + (f32x2){cenT[f][0][k], cenT[f][1][k]};
          const float pzv = o2 + cenT[f][2][k];
          f32x2 q;
          if (SAFE) {
            q = div2_by(pxy, recip_refined(pzv));
          } else {
            q.x = pxy.x / pzv;
            q.y = pxy.y / pzv;
          }
          const f32x2 uw = q * fxy + cxy;
          const int X = SAFE ? cvt_rne_hw(uw.x) : cvt_sat_rne(uw.x);
          const int Y = SAFE ? cvt_rne_hw(uw.y) : cvt_sat_rne(uw.y);
          const bool valid = ((unsigned)(X - 1) < (unsigned)(W - 2)) && ((unsigned)(Y - 1) < (unsigned)(H - 2));
          vm[j] = ballot(valid);
          int od = (__mul24(Y, W) + X) * 4;
          asm volatile("" : "+v"(od));
          off_d[j] = valid ? od : kOOB;
        }
      };
      if (div_safe) geometry(std::true_type{});
      else geometry(std::false_type{});
      unsigned long long all_valid = ~0ull;
#pragma unroll
      for (int j = 0; j < 8; ++j) all_valid &= vm[j];
      if (all_valid != ~0ull) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (R == 64u) {
            const unsigned long long dead = nonzero_bytes(vm[j]) ^ 0x0101010101010101ull;
            if (dead) R = (uint32_t)(j * 8) + ((uint32_t)__builtin_ctzll(dead) >> 3);
          }
          const bool live_lane = (uint32_t)(j * 8 + vy) < R;
          off_d[j] = live_lane ? off_d[j] : kOOB;
        }
      }
      float dep[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        dep[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_depth, off_d[j], 0, 0));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((uint32_t)(j * 8) >= R) break;  // a stalled row ends the chunk
        const float d = dep[j];
        const float sd = d - (o2 + cenT[f][2][j * 64 + lane]);
        const bool act = (uint32_t)(j * 8 + vy) < R;
        const bool dv = (d > cam.nearP) && (cam.farP > d);
        const bool inside = (sd > kc.lower) && (upper > sd);
        const bool F = act && dv && inside;
        const float nw = F ? wD : 0.0f;
        const bool rf_l = row_any(ballot(F));
        rows_total += (uint32_t)__popcll(ballot(rf_l)) >> 3;
        if (rf_l) {  // every lane of a rewritten row is recomputed, with weight 0 where the voxel itself is not hit
          const float ts = __uint_as_float(t[j].x), tw = __uint_as_float(t[j].y);
          const float num = ts * tw + sd * nw;
          const float den = (tw + nw) + kc.sigma;
          const float ns = num / den;
          const float nwt = tw + nw;
          const bool keep = nwt > 0.5f;
          t[j].x = __float_as_uint(keep ? ns : 999.0f);
          t[j].y = __float_as_uint(keep ? nwt : 0.0f);
          dirty_rows |= 1u << j;
        }
      }
    }
    uint32_t word = 0;  // VolumeDev::summ: classes of the rewritten rows' final values
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool wr = ((dirty_rows >> j) & 1u) != 0;
      __builtin_amdgcn_raw_buffer_store_b64(t[j], rs_T, wr ? (j * 64 + lane) * 8 : kOOB, 0, 0);
      if (wr) word |= chunk_summary_bits(__uint_as_float(t[j].x), __uint_as_float(t[j].y), (uint32_t)(j * 64 + lane));
    }
    word = wave_or(word);
    if (lane == 0 && word) atomicOr(&v.summ[slot], word);
    const bool updated = (rows_total | key_rows_t) != 0;
    if (updated && lane == 0) {
      L.list_needs[e] = 1;  // needsUpdateFlag[i] |= needsUpdate (Chisel.h:241)
      if (part_band(v, id.x, id.y, id.z)) mark_touched(v, L.list_ent[e]);  // multi-GPU: touched since the last exchange
    }
    const uint32_t rows_key = KEY ? (key_rows_c << 8) : (fin ? (uint32_t)L.list_rows[e] : 0u);  // the keyframe's own pass: rows_t | rows_c << 8
    if (lane == 0) L.list_rows[e] = (uint16_t)(rows_total + key_rows_t < 255u ? rows_total + key_rows_t : 255u);  // (statistic; saturates)
    if (fin) {
      const bool isnew = L.list_new[e] != 0;
      if (updated || L.list_needs[e]) {
        if (lane == 0) {
          v.mark_epoch[slot] = fin_epoch + 1u;  // meshesToUpdate[id and 6 nbrs] = true, expanded on read
          // (the list was acquired lazily: a parked chunk that got data is revived here; band chunks carry bit 1 already)
          if (isnew && !part_band(v, id.x, id.y, id.z)) v.hent[L.list_ent[e]].alive = 1u;
        }
        // (a texture stage follows the call's groups: the chunk joins its dirty set here, as K-A's chunks do in a stream)
        if (claim_par >= 0) claim_dirty7(v, id, slot, L.list_ent[e], lane, claim_stamp, claim_par);
      } else if (isnew) {
        // GarbageCollect: RemoveChunk + meshesToUpdate.erase.  Parked storage goes back to the fresh state: colour rows
        // can have been written by the keyframe's pass (colour band hit with depth outside [near, far]: its row count says);
        // TSDF rows only when the caller's earlier flags removed a chunk WITH data -- the summary word tells (see k_finalize)
        const uint32_t ent = L.list_ent[e];
        if (rows_key >> 8) {
          uint4* c4 = reinterpret_cast<uint4*>(v.color + (size_t)slot * kChunkVoxels);
#pragma unroll
          for (int k = 0; k < 4; ++k) c4[k * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
        if (v.hent[ent].alive) {  // (created by this list's acquire: a chunk that was parked before stayed parked)
          const uint32_t ds = v.summ[slot];
          if (ds != 0u) {
            const uint32_t f999 = __float_as_uint(999.0f);
            uint4* t4 = reinterpret_cast<uint4*>(v.tsdf + (size_t)slot * kChunkVoxels);
#pragma unroll
            for (int k = 0; k < 4; ++k) t4[k * 64 + lane] = make_uint4(f999, 0u, f999, 0u);
          }
          if (lane == 0) {
            v.hent[ent].alive = 0;
            if (ds != 0u) {
              v.summ[slot] = 0u;
              MeshRec* r = &v.mesh_rec[slot];
              if (r->state & kMsInMap) { r->state = 0u; r->nv = 0; r->nt = 0; }
            }
          }
        }
        if (lane == 0) v.erase_epoch[slot] = fin_epoch + 1u;
      }
    }
  }
}


void launch_pre_frames(const VolumeDev& v, const Pose& keyframe, int n, const float* poses12, float4* pre_scratch,
                       float* cen_scratch, const Integ& ig, float res, const Cam& cam, hipStream_t s, int acquire,
                       uint32_t* clear_word) {
  const IntegrateConsts kc = make_integrate_consts(cam.cxi, cam.cyi, res, 1);  // (resDiag does not depend on the flag)
  GroupPoses gp = {};
  for (int f = 0; f < n; ++f)
    for (int q = 0; q < 12; ++q) gp.P[f].p[q] = poses12[12 * f + q];
  hipLaunchKernelGGL(k_pre_group, dim3(128, n + 1 + (acquire ? 1 : 0)), dim3(256), 0, s, v, gp, ig, res, kc.resDiag, pre_scratch,
                     cen_scratch, keyframe, 1, acquire, clear_word);
}

void launch_integrate_group(const VolumeDev& v, int n, const float* const* d_depth, const float* poses12, float4* pre_scratch,
                            float* cen_scratch, const Cam& cam, const Integ& ig, float res, int flag, hipStream_t s, bool have_pre,
                            int32_t obs_kf, int fin, uint32_t fin_epoch, int claim_par, uint32_t claim_stamp,
                            const FrameImages* key_img, const Pose* key_pose) {
  IntegrateConsts kc = make_integrate_consts(cam.cxi, cam.cyi, res, flag);
  GroupArgs ga = {};
  GroupPoses gp = {};
  ga.n = n;
  for (int f = 0; f < n; ++f) {
    for (int q = 0; q < 12; ++q) gp.P[f].p[q] = poses12[12 * f + q];
    ga.depth[f] = d_depth[f];
    ga.pre[f] = pre_scratch + (size_t)f * 4 * v.max_list;
    ga.cen[f] = cen_scratch + (size_t)f * 3 * kChunkVoxels;
  }
  if (!have_pre) hipLaunchKernelGGL(k_pre_group, dim3(128, n), dim3(256), 0, s, v, gp, ig, res, kc.resDiag, pre_scratch, cen_scratch, Pose{}, 0, 0, (uint32_t*)nullptr);
  static const int cus = device_cus();
  const dim3 grid(cus * 4), block(256);  // 36 KB of LDS per workgroup: four per CU
  GroupKey key = {};
  if (key_img) { key.img = *key_img; key.P = *key_pose; }
#define TF_LAUNCH_GROUP(F, K, Q) \
  hipLaunchKernelGGL((k_integrate_group<F, K, Q>), grid, block, 0, s, v, ga, cam, kc, obs_kf, fin, fin_epoch, claim_par, claim_stamp, key)
  if (key_img && key_img->quality) { if (flag) TF_LAUNCH_GROUP(true, true, true); else TF_LAUNCH_GROUP(false, true, true); }
  else if (key_img) { if (flag) TF_LAUNCH_GROUP(true, true, false); else TF_LAUNCH_GROUP(false, true, false); }
  else { if (flag) TF_LAUNCH_GROUP(true, false, false); else TF_LAUNCH_GROUP(false, false, false); }
#undef TF_LAUNCH_GROUP
}

}  // namespace tf
