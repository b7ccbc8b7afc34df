/*
 * tf_oracle.c -- CPU restatement of the TextureFusion voxel-fusion + atlas-update hot path.
 * TEST INFRASTRUCTURE ONLY; see tf_oracle.h for the parity status ("parity unpinned" apart
 * from the SURVEY.md App. A.1-9 known answers and analytic KATs).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math, scalar SSE2 float).
 */
#include "tf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* helpers                                                                              */
/* ------------------------------------------------------------------------------------ */

/* _mm256_cvtps_epi32 under default MXCSR: round-to-nearest-even; NaN / out of range ->
 * 0x80000000 ("integer indefinite"). */
static inline int32_t cvt_rne(float x) {
  if (!(x >= -2147483648.0f && x < 2147483648.0f)) return INT32_MIN;
  return (int32_t)lrintf(x);
}

#define R_(p, i, j) ((p)[4 * (i) + (j)])
#define T_(p, i) ((p)[4 * (i) + 3])

/* Fixed-size Eigen 3-vector dot in redux order a0*b0 + (a1*b1 + a2*b2) (Eigen's unrolled
 * redux halves the range; order itself is unpinned). */
static inline float dot3_tree(float a0, float a1, float a2, float b0, float b1, float b2) {
  float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  float s = p1 + p2;
  return p0 + s;
}
/* Dynamic (MatrixXf) product: sequential accumulation. */
static inline float dot3_seq(float a0, float a1, float a2, float b0, float b1, float b2) {
  float s = a0 * b0;
  s = s + a1 * b1;
  s = s + a2 * b2;
  return s;
}

/* The two orders above are this restatement's reading of Eigen's code paths; Eigen is not in the image, so
 * which one the reference's build uses is unpinned.  tools/oracle_sum_order.py re-runs the stream with the
 * alternatives to bound what the choice can change: bit 0 = fixed-size sites accumulate sequentially, bit 1 =
 * dynamic sites use the tree order, bit 2 = the gradient's squared norm as (x*x + y*y) + z*z, bit 3 = normalize()
 * multiplies by the reciprocal (Eigen 3.2) instead of dividing (Eigen >= 3.3). */
static int g_sum_order = 0;
void tfo_set_sum_order(int bits) { g_sum_order = bits; }
int tfo_get_sum_order(void) { return g_sum_order; }
static inline float dot3_fixed(float a0, float a1, float a2, float b0, float b1, float b2) {
  return (g_sum_order & 1) ? dot3_seq(a0, a1, a2, b0, b1, b2) : dot3_tree(a0, a1, a2, b0, b1, b2);
}
static inline float dot3_dyn(float a0, float a1, float a2, float b0, float b1, float b2) {
  return (g_sum_order & 2) ? dot3_tree(a0, a1, a2, b0, b1, b2) : dot3_seq(a0, a1, a2, b0, b1, b2);
}

/* QuadraticTruncator::GetTruncationDistance (truncation/QuadraticTruncator.h:45-48):
 * std::abs(q * pow(z, 2) + l * z + c) * s  with pow(float,int) -> double, l*z in float. */
float tfo_truncation(const tfo_integrator* ig, float z) {
  double zz = (double)z * (double)z;
  float lz = ig->lin * z;
  double v = (double)ig->quad * zz + (double)lz + (double)ig->cons;
  return (float)(fabs(v) * (double)ig->scale);
}

/* ConstantWeighter::GetWeight (weighting/ConstantWeighter.h:43-46): weight / (2 * truncationDist), all float */
float tfo_weight(const tfo_integrator* ig, float truncation) { return ig->weight / (2.0f * truncation); }

/* chisel::parallel_for (threading/Threading.h:36-54) as a plan: items are cut into stretches of
 * group = max(max(1, |threshold|), n / |nthreads|) (integer division); one std::thread per stretch that starts
 * below n - group, the calling thread runs what is left (at most one group).  Returns the number of threads that
 * run items (spawned + the caller), i.e. ceil(n / group) for n > 0.  nthreads is the reference's function-static
 * hardware_concurrency() - 2. */
int tfo_parallel_for_plan(int64_t n, int nthreads, int threshold, int64_t* group_out) {
  int64_t thr = threshold < 0 ? -(int64_t)threshold : threshold;
  int64_t nt = nthreads < 0 ? -(int64_t)nthreads : nthreads;
  int64_t group = thr > 1 ? thr : 1;
  if (nt > 0 && n / nt > group) group = n / nt;
  if (group_out) *group_out = group;
  int spawned = 0;
  for (int64_t it = 0; it < n - group; it += group) spawned++;
  return spawned + 1;
}

/* Chisel::bufferIntegratorSIMDCentroids (Structure/Chisel.cpp:52-110):
 * c[i] = (R^T * (x,y,z)) * res + res/2, i = (z*8+y)*8+x; half voxel added un-rotated. */
void tfo_centroids(const float pose[12], float res, float cen[3 * TFO_CHUNK_VOXELS]) {
  float half = res * 0.5f;
  int i = 0;
  for (int z = 0; z < 8; z++)
    for (int y = 0; y < 8; y++)
      for (int x = 0; x < 8; x++, i++) {
        float fx = (float)x, fy = (float)y, fz = (float)z;
        for (int a = 0; a < 3; a++) {
          /* row a of R^T = column a of R */
          float d = dot3_fixed(R_(pose, 0, a), R_(pose, 1, a), R_(pose, 2, a), fx, fy, fz);
          cen[a * TFO_CHUNK_VOXELS + i] = d * res + half;
        }
      }
}

/* Per-chunk scalars of voxelUpdateSIMD (utils/ProjectionIntegrator.cpp:74-101) and
 * Chunk origin (geometry/Chunk.cpp:52). */
void tfo_chunk_scalars(const tfo_integrator* ig, const float pose[12], const int id[3], float res,
                       float origin_cam[3], float* truncation, float* weight) {
  float o[3], d[3];
  for (int a = 0; a < 3; a++) {
    o[a] = (float)(8 * id[a]) * res;
    d[a] = o[a] - T_(pose, a);
  }
  for (int a = 0; a < 3; a++)
    origin_cam[a] = dot3_fixed(R_(pose, 0, a), R_(pose, 1, a), R_(pose, 2, a), d[0], d[1], d[2]);
  float tr = tfo_truncation(ig, origin_cam[2]);
  *truncation = tr;
  *weight = tfo_weight(ig, tr);
}

/* ------------------------------------------------------------------------------------ */
/* K-A  ProjectionIntegrator::voxelUpdateSIMD (utils/ProjectionIntegrator.cpp:67-426)   */
/* ------------------------------------------------------------------------------------ */
int tfo_voxel_update(const float* depth, const uint8_t* rgba, const float* quality,
                     const tfo_camera* cam, const tfo_integrator* ig, const float pose[12],
                     int integrate_flag, const int id[3], float res,
                     const float cen[3 * TFO_CHUNK_VOXELS], float* sdf, float* weight,
                     uint16_t* color, float* quality_out, tfo_rowstats* stats) {
  const float* c0 = cen;
  const float* c1 = cen + TFO_CHUNK_VOXELS;
  const float* c2 = cen + 2 * TFO_CHUNK_VOXELS;
  float qsum = 0.0f;
  int updated = 0;

  const float resDiag = (float)(sqrt(3.0) * (double)res); /* :77 (double sqrt) */
  const float fxi = (float)(int)cam->fx, fyi = (float)(int)cam->fy; /* :79-82, int getters */
  const float cxi = (float)(int)cam->cx, cyi = (float)(int)cam->cy;
  const int W = cam->width, H = cam->height;
  float o[3], trunc, wD;
  tfo_chunk_scalars(ig, pose, id, res, o, &trunc, &wD); /* :88-92 */
  if (!integrate_flag) wD *= -1.0f;                      /* :95-99 */
  const float thrCol = (float)((double)(resDiag / 2.0f) + 0.01); /* :101 */
  const float nthrCol = -thrCol;
  const float cxs = (float)((double)cxi + 0.5), cys = (float)((double)cyi + 0.5); /* :114-115 */
  const float lower = (float)(-0.03);   /* :314 */
  const float upper = trunc + resDiag;  /* :315 */
  const float nearP = cam->near_plane, farP = cam->far_plane;
  const float sigma = (float)1e-4;      /* :126 */

  int pos = 0;
  for (int it = 0; it < 64; it++) { /* z,y loops; x advances by 8 (:145-147) */
    float pz[8], sd[8], dval[8];
    int32_t X[8], Y[8], idx[8];
    int valid[8], anyvalid = 0;
    for (int l = 0; l < 8; l++) {
      int k = 8 * pos + l;
      float px = o[0] + c0[k];
      float py = o[1] + c1[k];
      pz[l] = o[2] + c2[k];
      float u = (px / pz[l]) * fxi + cxs; /* :155-164: div, mul, add each rounded */
      float v = (py / pz[l]) * fyi + cys;
      X[l] = cvt_rne(u);
      Y[l] = cvt_rne(v);
      valid[l] = (X[l] > 0) && (W - 1 > X[l]) && (Y[l] > 0) && (H - 1 > Y[l]); /* :167-173 */
      anyvalid |= valid[l];
    }
    if (!anyvalid) continue; /* :176-178: `continue` skips pos++ -> this row is re-tested forever */

    for (int l = 0; l < 8; l++) {
      idx[l] = (int32_t)((uint32_t)Y[l] * (uint32_t)W + (uint32_t)X[l]); /* :180-181 */
      dval[l] = valid[l] ? depth[idx[l]] : 0.0f;                          /* :182-183 */
      sd[l] = dval[l] - pz[l];                                            /* :191-192 */
    }

    if (rgba != NULL) { /* :201-306 */
      int upd[8], anyupd = 0, oob = 0;
      for (int l = 0; l < 8; l++) {
        upd[l] = valid[l] && (sd[l] > nthrCol) && (thrCol > sd[l]); /* :202-208 */
        anyupd |= upd[l];
        oob |= (0 > X[l]) || (X[l] > W - 1) || (0 > Y[l]) || (Y[l] > H - 1); /* :212-220 */
      }
      if (oob) qsum = (float)(-99999999999.0); /* :221-222 (assignment, not accumulation) */
      if (anyupd) {
        if (quality != NULL) { /* :227-238 */
          float sum = 0.0f;
          for (int l = 0; l < 8; l++) sum += (upd[l] ? quality[idx[l]] : 0.0f);
          qsum += sum;
        }
        for (int l = 0; l < 8; l++) { /* :250-304, all 8 voxels of the row are rewritten */
          uint16_t* c = color + (size_t)(8 * pos + l) * 4;
          uint16_t in[4] = {0, 0, 0, 0};
          if (upd[l]) {
            const uint8_t* p = rgba + (size_t)idx[l] * 4;
            in[0] = p[0]; in[1] = p[1]; in[2] = p[2]; in[3] = p[3];
          }
          if (integrate_flag) {
            uint16_t n[4];
            for (int k = 0; k < 4; k++) n[k] = (uint16_t)(c[k] + in[k]);
            if ((int16_t)n[3] > 120) /* :281-287: signed compare on the count channel */
              for (int k = 0; k < 4; k++) n[k] = (uint16_t)(n[k] >> 2);
            for (int k = 0; k < 4; k++) c[k] = n[k];
          } else {
            for (int k = 0; k < 4; k++) c[k] = (uint16_t)(c[k] - in[k]);
          }
        }
        if (stats) stats->rows_color++;
      }
    }

    int F[8], anyF = 0; /* :310-317 */
    for (int l = 0; l < 8; l++) {
      int dv = (dval[l] > nearP) && (farP > dval[l]);
      int inside = (sd[l] > lower) && (upper > sd[l]);
      F[l] = dv && inside;
      anyF |= F[l];
    }
    if (anyF) { /* :319-341: all 8 lanes of the row are rewritten */
      updated = 1;
      for (int l = 0; l < 8; l++) {
        int k = 8 * pos + l;
        float w = weight[k], s = sdf[k];
        float nw = F[l] ? wD : 0.0f;
        float num = s * w + sd[l] * nw;
        float den = (w + nw) + sigma;
        float ns = num / den;
        float nwt = w + nw;
        if (nwt > 0.5f) { sdf[k] = ns; weight[k] = nwt; }
        else { sdf[k] = 999.0f; weight[k] = 0.0f; }
      }
      if (stats) stats->rows_tsdf++;
    }
    pos++; /* :420 */
  }
  if (stats && updated) stats->chunks_updated++;
  *quality_out = qsum; /* :424 */
  return updated;
}

/* ------------------------------------------------------------------------------------ */
/* K-A, AVX2 form: the same row algorithm as tfo_voxel_update with one __m256 per 8-voxel   */
/* row, used for the CPU baseline (the reference's kernel is AVX2 too).  Must be            */
/* bit-identical to the scalar form above -- tests/test_oracle_kat.py checks that.          */
/* ------------------------------------------------------------------------------------ */
#if defined(__x86_64__)
#include <immintrin.h>
#define TFO_HAVE_AVX2_KERNEL 1

__attribute__((target("avx2"))) static inline __m256 ps_and3(__m256 a, __m256 b, __m256 c) {
  return _mm256_and_ps(_mm256_and_ps(a, b), c);
}

__attribute__((target("avx2"))) static int voxel_update_avx2(
    const float* depth, const uint8_t* rgba, const float* quality, const tfo_camera* cam,
    const tfo_integrator* ig, const float pose[12], int integrate_flag, const int id[3], float res,
    const float cen[3 * TFO_CHUNK_VOXELS], float* sdf, float* weight, uint16_t* color,
    float* quality_out, tfo_rowstats* stats) {
  const float* c0 = cen;
  const float* c1 = cen + TFO_CHUNK_VOXELS;
  const float* c2 = cen + 2 * TFO_CHUNK_VOXELS;
  float qsum = 0.0f;
  int updated = 0;
  const float resDiag = (float)(sqrt(3.0) * (double)res);
  const int W = cam->width, H = cam->height;
  float o[3], trunc, wD;
  tfo_chunk_scalars(ig, pose, id, res, o, &trunc, &wD);
  if (!integrate_flag) wD *= -1.0f;
  const float thrCol = (float)((double)(resDiag / 2.0f) + 0.01);

  const __m256 vo0 = _mm256_set1_ps(o[0]), vo1 = _mm256_set1_ps(o[1]), vo2 = _mm256_set1_ps(o[2]);
  const __m256 vfx = _mm256_set1_ps((float)(int)cam->fx), vfy = _mm256_set1_ps((float)(int)cam->fy);
  const __m256 vcx = _mm256_set1_ps((float)((double)(float)(int)cam->cx + 0.5));
  const __m256 vcy = _mm256_set1_ps((float)((double)(float)(int)cam->cy + 0.5));
  const __m256i izero = _mm256_setzero_si256();
  const __m256i iWm1 = _mm256_set1_epi32(W - 1), iHm1 = _mm256_set1_epi32(H - 1), iW = _mm256_set1_epi32(W);
  const __m256 vzero = _mm256_setzero_ps();
  const __m256 vthr = _mm256_set1_ps(thrCol), vnthr = _mm256_set1_ps(-thrCol);
  const __m256 vnear = _mm256_set1_ps(cam->near_plane), vfar = _mm256_set1_ps(cam->far_plane);
  const __m256 vlower = _mm256_set1_ps((float)(-0.03)), vupper = _mm256_set1_ps(trunc + resDiag);
  const __m256 vwD = _mm256_set1_ps(wD), vsigma = _mm256_set1_ps((float)1e-4);
  const __m256 vhalf = _mm256_set1_ps(0.5f), v999 = _mm256_set1_ps(999.0f);
  /* byte shuffle that copies the count channel (u16 #3 of each 4-u16 voxel) over the voxel */
  const __m256i cntsel = _mm256_setr_epi8(6, 7, 6, 7, 6, 7, 6, 7, 14, 15, 14, 15, 14, 15, 14, 15,
                                          6, 7, 6, 7, 6, 7, 6, 7, 14, 15, 14, 15, 14, 15, 14, 15);
  const __m256i i120 = _mm256_set1_epi16(120);

  int pos = 0;
  for (int it = 0; it < 64; it++) {
    const __m256 px = _mm256_add_ps(vo0, _mm256_loadu_ps(c0 + 8 * pos));
    const __m256 py = _mm256_add_ps(vo1, _mm256_loadu_ps(c1 + 8 * pos));
    const __m256 pz = _mm256_add_ps(vo2, _mm256_loadu_ps(c2 + 8 * pos));
    const __m256 u = _mm256_add_ps(_mm256_mul_ps(_mm256_div_ps(px, pz), vfx), vcx);
    const __m256 v = _mm256_add_ps(_mm256_mul_ps(_mm256_div_ps(py, pz), vfy), vcy);
    const __m256i X = _mm256_cvtps_epi32(u), Y = _mm256_cvtps_epi32(v);
    __m256i valid = _mm256_and_si256(_mm256_cmpgt_epi32(X, izero), _mm256_cmpgt_epi32(iWm1, X));
    valid = _mm256_and_si256(valid, _mm256_and_si256(_mm256_cmpgt_epi32(Y, izero), _mm256_cmpgt_epi32(iHm1, Y)));
    if (_mm256_testz_si256(valid, valid)) continue; /* `pos` is not advanced: the row stalls */
    const __m256 validf = _mm256_castsi256_ps(valid);
    const __m256i idx = _mm256_add_epi32(_mm256_mullo_epi32(Y, iW), X);
    const __m256 d = _mm256_mask_i32gather_ps(vzero, depth, idx, validf, 4);
    const __m256 sd = _mm256_sub_ps(d, pz);

    if (rgba != NULL) {
      const __m256 upd = ps_and3(validf, _mm256_cmp_ps(sd, vnthr, _CMP_GT_OS), _mm256_cmp_ps(vthr, sd, _CMP_GT_OS));
      __m256i oob = _mm256_or_si256(_mm256_cmpgt_epi32(izero, X), _mm256_cmpgt_epi32(X, iWm1));
      oob = _mm256_or_si256(oob, _mm256_or_si256(_mm256_cmpgt_epi32(izero, Y), _mm256_cmpgt_epi32(Y, iHm1)));
      if (!_mm256_testz_si256(oob, oob)) qsum = (float)(-99999999999.0);
      if (!_mm256_testz_ps(upd, upd)) {
        if (quality != NULL) {
          float qv[8];
          _mm256_storeu_ps(qv, _mm256_mask_i32gather_ps(vzero, quality, idx, upd, 4));
          float sum = 0.0f;
          for (int l = 0; l < 8; l++) sum += qv[l];
          qsum += sum;
        }
        const __m256i in = _mm256_mask_i32gather_epi32(izero, (const int*)rgba, idx, _mm256_castps_si256(upd), 4);
        __m256i* crow = (__m256i*)(color + (size_t)pos * 32);
        for (int hlf = 0; hlf < 2; hlf++) {
          const __m256i in16 = _mm256_cvtepu8_epi16(hlf ? _mm256_extracti128_si256(in, 1) : _mm256_castsi256_si128(in));
          __m256i c = _mm256_loadu_si256(crow + hlf);
          if (integrate_flag) {
            c = _mm256_add_epi16(c, in16);
            const __m256i over = _mm256_shuffle_epi8(_mm256_cmpgt_epi16(c, i120), cntsel);
            c = _mm256_blendv_epi8(c, _mm256_srli_epi16(c, 2), over);
          } else {
            c = _mm256_sub_epi16(c, in16);
          }
          _mm256_storeu_si256(crow + hlf, c);
        }
        if (stats) stats->rows_color++;
      }
    }

    const __m256 dv = _mm256_and_ps(_mm256_cmp_ps(d, vnear, _CMP_GT_OS), _mm256_cmp_ps(vfar, d, _CMP_GT_OS));
    const __m256 F = ps_and3(dv, _mm256_cmp_ps(sd, vlower, _CMP_GT_OS), _mm256_cmp_ps(vupper, sd, _CMP_GT_OS));
    if (!_mm256_testz_ps(F, F)) {
      updated = 1;
      const __m256 w = _mm256_loadu_ps(weight + 8 * pos), s = _mm256_loadu_ps(sdf + 8 * pos);
      const __m256 nw = _mm256_and_ps(F, vwD);
      const __m256 num = _mm256_add_ps(_mm256_mul_ps(s, w), _mm256_mul_ps(sd, nw));
      const __m256 den = _mm256_add_ps(_mm256_add_ps(w, nw), vsigma);
      const __m256 ns = _mm256_div_ps(num, den);
      const __m256 nwt = _mm256_add_ps(w, nw);
      const __m256 keep = _mm256_cmp_ps(nwt, vhalf, _CMP_GT_OS);
      _mm256_storeu_ps(sdf + 8 * pos, _mm256_blendv_ps(v999, ns, keep));
      _mm256_storeu_ps(weight + 8 * pos, _mm256_and_ps(nwt, keep));
      if (stats) stats->rows_tsdf++;
    }
    pos++;
  }
  if (stats && updated) stats->chunks_updated++;
  *quality_out = qsum;
  return updated;
}
#endif

int tfo_have_avx2(void) {
#ifdef TFO_HAVE_AVX2_KERNEL
  return __builtin_cpu_supports("avx2") ? 1 : 0;
#else
  return 0;
#endif
}

/* Same contract as tfo_voxel_update; kernel = 1 selects the AVX2 row kernel when the CPU has it. */
int tfo_voxel_update_k(int kernel, const float* depth, const uint8_t* rgba, const float* quality,
                       const tfo_camera* cam, const tfo_integrator* ig, const float pose[12],
                       int integrate_flag, const int id[3], float res,
                       const float cen[3 * TFO_CHUNK_VOXELS], float* sdf, float* weight,
                       uint16_t* color, float* quality_out, tfo_rowstats* stats) {
#ifdef TFO_HAVE_AVX2_KERNEL
  if (kernel == 1 && tfo_have_avx2())
    return voxel_update_avx2(depth, rgba, quality, cam, ig, pose, integrate_flag, id, res, cen, sdf,
                             weight, color, quality_out, stats);
#endif
  return tfo_voxel_update(depth, rgba, quality, cam, ig, pose, integrate_flag, id, res, cen, sdf,
                          weight, color, quality_out, stats);
}

/* ------------------------------------------------------------------------------------ */
/* K-B  ChunkManager::findCubeCornerByMat / GetBoundaryChunkID / GetIDAt                */
/*      (Structure/ChunkManager.h:303-378, 197-207)                                     */
/* ------------------------------------------------------------------------------------ */
void tfo_bbox(const float* depth, const tfo_camera* cam, const float pose[12], float res,
              int min_id[3], int max_id[3]) {
  const int W = cam->width, H = cam->height;
  const float fx = (float)(int)cam->fx, fy = (float)(int)cam->fy;
  const float cx = (float)(int)cam->cx, cy = (float)(int)cam->cy;
  float mx[3] = {-1e8f, -1e8f, -1e8f}, mn[3] = {1e8f, 1e8f, 1e8f};
  const float off = (float)0.2;
  for (int i = 0; i < H; i++) {
    float ly = ((float)i - cy) / fy;
    for (int j = 0; j < W; j++) {
      float dz = depth[(size_t)i * W + j] + off;
      float lx = ((float)j - cx) / fx;
      float vx = lx * dz, vy = ly * dz;
      for (int a = 0; a < 3; a++) {
        float p = R_(pose, a, 0) * vx;
        p = p + R_(pose, a, 1) * vy;
        p = p + R_(pose, a, 2) * dz;
        p = p + T_(pose, a);
        mx[a] = (p > mx[a]) ? p : mx[a]; /* _mm256_max_ps(p, max) */
        mn[a] = (p < mn[a]) ? p : mn[a];
      }
    }
  }
  /* GetIDAt: floor(pos * (1.0f / (chunkSize * res))) */
  const float f = 1.0f / (8.0f * res);
  for (int a = 0; a < 3; a++) {
    max_id[a] = (int)floorf(mx[a] * f);
    min_id[a] = (int)floorf(mn[a] * f);
  }
}

/* ------------------------------------------------------------------------------------ */
/* K-C  ChunkManager::GetChunkIDsObservedByCamera + CheckCornerIntersectingSIMD         */
/*      (Structure/ChunkManager.h:380-559, 561-636)                                     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  float fx, fy, cx, cy;
  int W, H;
  float nearP, farP;
  const float* depth;
} probe_ctx;

static int probe8(const probe_ctx* pc, const float oc[3], float dtp, float dtn,
                  const float off[3][8]) {
  int valid[8], anyvalid = 0;
  float pz[8];
  int32_t idx[8];
  const int depthValid = (oc[2] > pc->nearP) && (pc->farP > oc[2]); /* :598-602 */
  for (int l = 0; l < 8; l++) {
    float px = oc[0] + off[0][l];
    float py = oc[1] + off[1][l];
    pz[l] = oc[2] + off[2][l];
    float u = (px / pz[l]) * pc->fx + pc->cx; /* :584-593, no +0.5 here */
    float v = (py / pz[l]) * pc->fy + pc->cy;
    int32_t X = cvt_rne(u), Y = cvt_rne(v);
    valid[l] = (X > 1) && (pc->W - 1 > X) && (Y > 1) && (pc->H - 1 > Y); /* :603-609 */
    anyvalid |= valid[l];
    idx[l] = (int32_t)((uint32_t)Y * (uint32_t)pc->W + (uint32_t)X);
  }
  if (!anyvalid) return 0; /* :611-613 */
  const float ndtn = -dtn;
  int hit = 0;
  for (int l = 0; l < 8; l++) {
    float d = valid[l] ? pc->depth[idx[l]] : 0.0f;
    float sd = d - pz[l];
    int inside = (sd > ndtn) && (dtp > sd);
    hit |= (valid[l] && inside && depthValid);
  }
  return hit;
}

#ifdef TFO_HAVE_AVX2_KERNEL
/* The same two stages with the reference's own vector shapes (the CPU baseline's form; bit-identical to the scalar
 * checker above, tests/test_oracle_kat.py): findCubeCornerByMat reads 8 pixels per step and keeps 8-lane running
 * min / max (ChunkManager.h:326-363); CheckCornerIntersectingSIMD puts the 8 probe corners into one vector (:561-636).
 * Element arithmetic is unchanged (separate mul / add, IEEE div), min / max are exact and order-free. */
__attribute__((target("avx2"))) static void bbox_avx2(const float* depth, const tfo_camera* cam, const float pose[12],
                                                       float res, int min_id[3], int max_id[3]) {
  const int W = cam->width, H = cam->height;
  const float fx = (float)(int)cam->fx, fy = (float)(int)cam->fy;
  const float cx = (float)(int)cam->cx, cy = (float)(int)cam->cy;
  __m256 vmx[3], vmn[3];
  for (int a = 0; a < 3; a++) { vmx[a] = _mm256_set1_ps(-1e8f); vmn[a] = _mm256_set1_ps(1e8f); }
  const __m256 off = _mm256_set1_ps((float)0.2), vfx = _mm256_set1_ps(fx), vcx = _mm256_set1_ps(cx);
  const __m256 lane = _mm256_setr_ps(0, 1, 2, 3, 4, 5, 6, 7);
  float mx[3] = {-1e8f, -1e8f, -1e8f}, mn[3] = {1e8f, 1e8f, 1e8f};
  const int W8 = W & ~7;
  for (int i = 0; i < H; i++) {
    const float ly = ((float)i - cy) / fy;
    const __m256 vly = _mm256_set1_ps(ly);
    for (int j = 0; j < W8; j += 8) {
      const __m256 dz = _mm256_add_ps(_mm256_loadu_ps(depth + (size_t)i * W + j), off);
      const __m256 lx = _mm256_div_ps(_mm256_sub_ps(_mm256_add_ps(_mm256_set1_ps((float)j), lane), vcx), vfx);
      const __m256 vx = _mm256_mul_ps(lx, dz), vy = _mm256_mul_ps(vly, dz);
      for (int a = 0; a < 3; a++) {
        __m256 p = _mm256_mul_ps(_mm256_set1_ps(R_(pose, a, 0)), vx);
        p = _mm256_add_ps(p, _mm256_mul_ps(_mm256_set1_ps(R_(pose, a, 1)), vy));
        p = _mm256_add_ps(p, _mm256_mul_ps(_mm256_set1_ps(R_(pose, a, 2)), dz));
        p = _mm256_add_ps(p, _mm256_set1_ps(T_(pose, a)));
        vmx[a] = _mm256_max_ps(p, vmx[a]);
        vmn[a] = _mm256_min_ps(p, vmn[a]);
      }
    }
    for (int j = W8; j < W; j++) { /* (widths that are no multiple of 8: the scalar form) */
      float dz = depth[(size_t)i * W + j] + (float)0.2;
      float lx = ((float)j - cx) / fx;
      float vx = lx * dz, vy = ly * dz;
      for (int a = 0; a < 3; a++) {
        float p = R_(pose, a, 0) * vx;
        p = p + R_(pose, a, 1) * vy;
        p = p + R_(pose, a, 2) * dz;
        p = p + T_(pose, a);
        mx[a] = (p > mx[a]) ? p : mx[a];
        mn[a] = (p < mn[a]) ? p : mn[a];
      }
    }
  }
  for (int a = 0; a < 3; a++) {
    float tx[8], tn[8];
    _mm256_storeu_ps(tx, vmx[a]);
    _mm256_storeu_ps(tn, vmn[a]);
    for (int l = 0; l < 8; l++) { mx[a] = tx[l] > mx[a] ? tx[l] : mx[a]; mn[a] = tn[l] < mn[a] ? tn[l] : mn[a]; }
  }
  const float f = 1.0f / (8.0f * res);
  for (int a = 0; a < 3; a++) {
    max_id[a] = (int)floorf(mx[a] * f);
    min_id[a] = (int)floorf(mn[a] * f);
  }
}

__attribute__((target("avx2"))) static int probe8_avx2(const probe_ctx* pc, const float oc[3], float dtp, float dtn,
                                                        const float off[3][8]) {
  const __m256 px = _mm256_add_ps(_mm256_set1_ps(oc[0]), _mm256_loadu_ps(off[0]));
  const __m256 py = _mm256_add_ps(_mm256_set1_ps(oc[1]), _mm256_loadu_ps(off[1]));
  const __m256 pz = _mm256_add_ps(_mm256_set1_ps(oc[2]), _mm256_loadu_ps(off[2]));
  const __m256 u = _mm256_add_ps(_mm256_mul_ps(_mm256_div_ps(px, pz), _mm256_set1_ps(pc->fx)), _mm256_set1_ps(pc->cx));
  const __m256 v = _mm256_add_ps(_mm256_mul_ps(_mm256_div_ps(py, pz), _mm256_set1_ps(pc->fy)), _mm256_set1_ps(pc->cy));
  const __m256i X = _mm256_cvtps_epi32(u), Y = _mm256_cvtps_epi32(v); /* rne; NaN / out of range -> 0x80000000 */
  const __m256i one = _mm256_set1_epi32(1);
  __m256i valid = _mm256_and_si256(_mm256_cmpgt_epi32(X, one), _mm256_cmpgt_epi32(_mm256_set1_epi32(pc->W - 1), X));
  valid = _mm256_and_si256(valid, _mm256_and_si256(_mm256_cmpgt_epi32(Y, one), _mm256_cmpgt_epi32(_mm256_set1_epi32(pc->H - 1), Y)));
  if (_mm256_testz_si256(valid, valid)) return 0; /* :611-613 */
  const int depthValid = (oc[2] > pc->nearP) && (pc->farP > oc[2]);
  if (!depthValid) return 0; /* (every lane's hit is ANDed with it) */
  const __m256i idx = _mm256_add_epi32(_mm256_mullo_epi32(Y, _mm256_set1_epi32(pc->W)), X);
  const __m256 d = _mm256_mask_i32gather_ps(_mm256_setzero_ps(), pc->depth, idx, _mm256_castsi256_ps(valid), 4);
  const __m256 sd = _mm256_sub_ps(d, pz);
  __m256 inside = _mm256_and_ps(_mm256_cmp_ps(sd, _mm256_set1_ps(-dtn), _CMP_GT_OS), _mm256_cmp_ps(_mm256_set1_ps(dtp), sd, _CMP_GT_OS));
  inside = _mm256_and_ps(inside, _mm256_castsi256_ps(valid));
  return _mm256_movemask_ps(inside) != 0;
}
#endif

/* 0 = the scalar checker, 1 = the AVX2 forms above (the CPU baseline, when the CPU has AVX2) */
static int g_select_kernel = 0;
void tfo_set_select_kernel(int kernel) { g_select_kernel = kernel; }

int64_t tfo_select(const float* depth, const tfo_camera* cam, const tfo_integrator* ig,
                   const float pose[12], float res, int32_t* ids, int64_t cap,
                   int64_t* n_coarse_tested) {
  int minID[3], maxID[3];
  int (*probe)(const probe_ctx*, const float[3], float, float, const float[3][8]) = probe8;
#ifdef TFO_HAVE_AVX2_KERNEL
  const int vec = g_select_kernel == 1 && tfo_have_avx2();
  if (vec) { bbox_avx2(depth, cam, pose, res, minID, maxID); probe = probe8_avx2; }
  else
#endif
  tfo_bbox(depth, cam, pose, res, minID, maxID); /* :395-396 (second full-image pass) */

  float diag = 8.0f * res / 2.0f; /* :398 */
  int step = 4;
  float negTrunc = (float)0.03;
  if ((double)res > 0.01) { /* :401-405 (float vs double literal) */
    diag = (float)((double)(8.0f * res) * sqrt(3.0));
    step = 1;
    negTrunc = (float)(0.05 * (double)res / 0.005);
  }
  probe_ctx pc;
  pc.fx = (float)(int)cam->fx; pc.fy = (float)(int)cam->fy;
  pc.cx = (float)(int)cam->cx; pc.cy = (float)(int)cam->cy;
  pc.W = cam->width; pc.H = cam->height;
  pc.nearP = cam->near_plane; pc.farP = cam->far_plane;
  pc.depth = depth;

  /* rotation = R^T (MatrixXf), translation = rotation * t (:429-430) */
  float rot[3][3], tc[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) rot[i][j] = R_(pose, j, i);
  for (int i = 0; i < 3; i++)
    tc[i] = dot3_dyn(rot[i][0], rot[i][1], rot[i][2], T_(pose, 0), T_(pose, 1), T_(pose, 2));
  float r0[3], r1[3], r2[3]; /* :431-436: columns scaled by 8 then res */
  for (int i = 0; i < 3; i++) {
    r0[i] = (rot[i][0] * 8.0f) * res;
    r1[i] = (rot[i][1] * 8.0f) * res;
    r2[i] = (rot[i][2] * 8.0f) * res;
  }
  const float half = res * 0.5f;
  float coarse[3][8], fine[3][8]; /* :444-470 */
  for (int x = 0; x < 2; x++)
    for (int y = 0; y < 2; y++)
      for (int z = 0; z < 2; z++) {
        float cur[3] = {(float)(x * 8), (float)(y * 8), (float)(z * 8)};
        int k = x + y * 2 + z * 4;
        for (int a = 0; a < 3; a++) {
          float d = dot3_dyn(rot[a][0], rot[a][1], rot[a][2], cur[0], cur[1], cur[2]);
          coarse[a][k] = (d * res) * (float)step + half;
          fine[a][k] = (d * res) * 1.0f + half;
        }
      }

  int64_t n = 0, ncoarse = 0;
  for (int x = minID[0] - 1; x <= maxID[0] + 1; x += step) {
    float ox[3];
    for (int a = 0; a < 3; a++) ox[a] = r0[a] * (float)x - tc[a]; /* :473 */
    for (int y = minID[1] - 1; y <= maxID[1] + 1; y += step) {
      float oy[3];
      for (int a = 0; a < 3; a++) oy[a] = ox[a] + r1[a] * (float)y; /* :475 */
      for (int z = minID[2] - 1; z <= maxID[2] + 1; z += step) {
        ncoarse++;
        float oc[3];
        for (int a = 0; a < 3; a++) oc[a] = oy[a] + (float)z * r2[a]; /* :479 */
        float trunc = tfo_truncation(ig, oc[2]);
        float dtp = trunc + diag * (float)step; /* :489-490 */
        float dtn = negTrunc + diag * (float)step;
        if (!probe(&pc, oc, dtp, dtn, coarse)) continue;
        for (int i = x; i < x + step; i++)
          for (int j = y; j < y + step; j++)
            for (int k = z; k < z + step; k++) {
              float org[3] = {(float)(i * 8) * res, (float)(j * 8) * res, (float)(k * 8) * res};
              float of[3];
              for (int a = 0; a < 3; a++) /* :521-524 */
                of[a] = dot3_dyn(rot[a][0], rot[a][1], rot[a][2], org[0], org[1], org[2]) - tc[a];
              float tr = tfo_truncation(ig, of[2]);
              float fdtp = tr + diag;
              float fdtn = negTrunc + diag;
              if (probe(&pc, of, fdtp, fdtn, fine)) {
                if (n < cap) { ids[3 * n] = i; ids[3 * n + 1] = j; ids[3 * n + 2] = k; }
                n++;
              }
            }
      }
    }
  }
  if (n_coarse_tested) *n_coarse_tested = ncoarse;
  return n;
}

/* ------------------------------------------------------------------------------------ */
/* volume: ChunkManager's ChunkMap + Chisel::meshesToUpdate                              */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  int32_t kf;
  float q;
} tfo_obs;

typedef struct {
  int32_t id[3];
  float* sdf;      /* [512]  DistVoxel.h:102-103, init 999 (Chunk.cpp:64) */
  float* weight;   /* [512]  init 0 (Chunk.cpp:65) */
  uint16_t* color; /* [2048] ColorVoxel.cpp:26-33, init 0 */
  tfo_obs* obs;    /* Chunk::observations (std::map<int,float>), kept sorted by kf */
  int n_obs, cap_obs;
  int alive;
} tfo_chunk;

typedef struct {
  int32_t* keys;  /* 3 per entry */
  int64_t* vals;  /* -1 empty, -2 tombstone, else payload */
  int64_t cap, used, live;
} tfo_map;

struct tfo_volume {
  float res;
  int use_color;
  int nthreads;
  int kernel; /* 0 = scalar restatement (the checker), 1 = AVX2 row kernel (CPU baseline) */
  tfo_camera cam;
  tfo_integrator ig;
  tfo_chunk* chunks;
  int64_t n_chunks, cap_chunks;
  int64_t* free_list;
  int64_t n_free, cap_free;
  tfo_map map;   /* ChunkID -> index in chunks[] */
  tfo_map dirty; /* meshesToUpdate: ChunkID -> 1 */
  tfo_rowstats stats;
  /* ChunkManager::allMeshes (ChunkManager.h:1304): ChunkID -> index in meshes[] */
  struct tfo_mesh* meshes;
  int64_t n_meshes, cap_meshes;
  tfo_map mesh_map;
};
static void vol_erase_mesh(tfo_volume* v, const int32_t* id);
static void vol_meshes_clear(tfo_volume* v);

static uint64_t hash3(const int32_t* k) { /* Teschner hash, ChunkManager.h:44-53 */
  return ((uint64_t)(int64_t)k[0] * 73856093ull) ^ ((uint64_t)(int64_t)k[1] * 19349663ull) ^
         ((uint64_t)(int64_t)k[2] * 83492791ull);
}
static void map_init(tfo_map* m, int64_t cap) {
  m->cap = cap; m->used = 0; m->live = 0;
  m->keys = (int32_t*)malloc(sizeof(int32_t) * 3 * cap);
  m->vals = (int64_t*)malloc(sizeof(int64_t) * cap);
  for (int64_t i = 0; i < cap; i++) m->vals[i] = -1;
}
static void map_free(tfo_map* m) { free(m->keys); free(m->vals); m->keys = NULL; m->vals = NULL; }
static int64_t map_find(const tfo_map* m, const int32_t* k) {
  uint64_t h = hash3(k) * 0x9E3779B97F4A7C15ull;
  int64_t i = (int64_t)(h >> 17) & (m->cap - 1);
  for (;;) {
    if (m->vals[i] == -1) return -1;
    if (m->vals[i] != -2 && m->keys[3 * i] == k[0] && m->keys[3 * i + 1] == k[1] &&
        m->keys[3 * i + 2] == k[2])
      return i;
    i = (i + 1) & (m->cap - 1);
  }
}
static void map_put_raw(tfo_map* m, const int32_t* k, int64_t v) {
  uint64_t h = hash3(k) * 0x9E3779B97F4A7C15ull;
  int64_t i = (int64_t)(h >> 17) & (m->cap - 1);
  while (m->vals[i] >= 0) i = (i + 1) & (m->cap - 1);
  if (m->vals[i] == -1) m->used++;
  m->keys[3 * i] = k[0]; m->keys[3 * i + 1] = k[1]; m->keys[3 * i + 2] = k[2];
  m->vals[i] = v;
  m->live++;
}
static void map_grow(tfo_map* m) {
  tfo_map n;
  map_init(&n, (m->live * 4 > m->cap) ? m->cap * 2 : m->cap);
  for (int64_t i = 0; i < m->cap; i++)
    if (m->vals[i] >= 0) map_put_raw(&n, m->keys + 3 * i, m->vals[i]);
  map_free(m);
  *m = n;
}
static void map_put(tfo_map* m, const int32_t* k, int64_t v) {
  int64_t i = map_find(m, k);
  if (i >= 0) { m->vals[i] = v; return; }
  if ((m->used + 1) * 2 > m->cap) map_grow(m);
  map_put_raw(m, k, v);
}
static void map_erase(tfo_map* m, const int32_t* k) {
  int64_t i = map_find(m, k);
  if (i >= 0) { m->vals[i] = -2; m->live--; }
}

tfo_volume* tfo_volume_create(float res, int use_color) {
  tfo_volume* v = (tfo_volume*)calloc(1, sizeof(tfo_volume));
  v->res = res;
  v->use_color = use_color;
  v->nthreads = 1;
  map_init(&v->map, 1 << 12);
  map_init(&v->dirty, 1 << 12);
  map_init(&v->mesh_map, 1 << 12);
  /* defaults of MobileFusion::initChiselMap (GCFusion/MobileFusion.h:205-258) */
  v->ig.quad = 0.0019f; v->ig.lin = 0.00152f; v->ig.cons = 0.001504f; v->ig.scale = 6.0f;
  v->ig.weight = 1.0f;
  v->cam.width = 640; v->cam.height = 480;
  v->cam.fx = 525.0f; v->cam.fy = 525.0f; v->cam.cx = 319.5f; v->cam.cy = 239.5f;
  v->cam.near_plane = 0.01f; v->cam.far_plane = 5.0f;
  return v;
}
static void chunk_release(tfo_chunk* c) {
  free(c->sdf); free(c->weight); free(c->color); free(c->obs);
  memset(c, 0, sizeof(*c));
}
void tfo_volume_reset(tfo_volume* v) { /* Chisel::Reset (Chisel.cpp:47-50) */
  for (int64_t i = 0; i < v->n_chunks; i++)
    if (v->chunks[i].alive) chunk_release(&v->chunks[i]);
  v->n_chunks = 0; v->n_free = 0;
  map_free(&v->map); map_free(&v->dirty);
  map_init(&v->map, 1 << 12); map_init(&v->dirty, 1 << 12);
  vol_meshes_clear(v); /* chunkManager.Reset(): allMeshes.clear() (ChunkManager.cpp:272-275) */
  memset(&v->stats, 0, sizeof(v->stats));
}
void tfo_volume_destroy(tfo_volume* v) {
  if (!v) return;
  tfo_volume_reset(v);
  map_free(&v->map); map_free(&v->dirty); map_free(&v->mesh_map);
  free(v->chunks); free(v->free_list); free(v->meshes);
  free(v);
}
void tfo_volume_set_camera(tfo_volume* v, const tfo_camera* cam) { v->cam = *cam; }
void tfo_volume_set_integrator(tfo_volume* v, const tfo_integrator* ig) { v->ig = *ig; }
void tfo_volume_set_threads(tfo_volume* v, int n) { v->nthreads = n < 1 ? 1 : n; }
void tfo_volume_set_kernel(tfo_volume* v, int kernel) { v->kernel = kernel; }
int64_t tfo_volume_num_chunks(const tfo_volume* v) { return v->map.live; }
int64_t tfo_volume_list_chunks(const tfo_volume* v, int32_t* ids, int64_t cap) {
  int64_t n = 0;
  for (int64_t i = 0; i < v->n_chunks; i++)
    if (v->chunks[i].alive) {
      if (n < cap) memcpy(ids + 3 * n, v->chunks[i].id, 12);
      n++;
    }
  return n;
}
static tfo_chunk* vol_get(const tfo_volume* v, const int32_t* id) {
  int64_t i = map_find(&v->map, id);
  return i < 0 ? NULL : &v->chunks[v->map.vals[i]];
}
int tfo_volume_has_chunk(const tfo_volume* v, const int id[3]) { return vol_get(v, id) != NULL; }

/* ChunkManager::CreateChunk (ChunkManager.cpp:266-270) + Chunk ctor (Chunk.cpp:38-76) */
static tfo_chunk* vol_create_chunk(tfo_volume* v, const int32_t* id) {
  int64_t slot;
  if (v->n_free > 0) slot = v->free_list[--v->n_free];
  else {
    if (v->n_chunks == v->cap_chunks) {
      v->cap_chunks = v->cap_chunks ? v->cap_chunks * 2 : 4096;
      v->chunks = (tfo_chunk*)realloc(v->chunks, sizeof(tfo_chunk) * v->cap_chunks);
    }
    slot = v->n_chunks++;
  }
  tfo_chunk* c = &v->chunks[slot];
  memset(c, 0, sizeof(*c));
  memcpy(c->id, id, 12);
  c->sdf = (float*)malloc(sizeof(float) * 512);
  c->weight = (float*)malloc(sizeof(float) * 512);
  c->color = (uint16_t*)calloc(2048, sizeof(uint16_t));
  for (int i = 0; i < 512; i++) { c->sdf[i] = 999.0f; c->weight[i] = 0.0f; }
  c->alive = 1;
  map_put(&v->map, id, slot);
  return c;
}
/* ChunkManager::RemoveChunk (ChunkManager.h:151-161) */
static void vol_remove_chunk(tfo_volume* v, const int32_t* id) {
  int64_t i = map_find(&v->map, id);
  if (i < 0) return;
  int64_t slot = v->map.vals[i];
  chunk_release(&v->chunks[slot]);
  map_erase(&v->map, id);
  vol_erase_mesh(v, id); /* RemoveChunk also drops the chunk's mesh (ChunkManager.h:151-161) */
  if (v->n_free == v->cap_free) {
    v->cap_free = v->cap_free ? v->cap_free * 2 : 1024;
    v->free_list = (int64_t*)realloc(v->free_list, sizeof(int64_t) * v->cap_free);
  }
  v->free_list[v->n_free++] = slot;
}
int tfo_volume_get_chunk(const tfo_volume* v, const int id[3], float* sdf, float* weight,
                         uint16_t* color) {
  tfo_chunk* c = vol_get(v, id);
  if (!c) return -1;
  if (sdf) memcpy(sdf, c->sdf, 512 * 4);
  if (weight) memcpy(weight, c->weight, 512 * 4);
  if (color) memcpy(color, c->color, 2048 * 2);
  return 0;
}
int tfo_volume_set_chunk(tfo_volume* v, const int id[3], const float* sdf, const float* weight,
                         const uint16_t* color) {
  tfo_chunk* c = vol_get(v, id);
  if (!c) c = vol_create_chunk(v, id);
  if (sdf) memcpy(c->sdf, sdf, 512 * 4);
  if (weight) memcpy(c->weight, weight, 512 * 4);
  if (color) memcpy(c->color, color, 2048 * 2);
  return 0;
}
static void chunk_set_obs(tfo_chunk* c, int kf, float q) { /* observations[kf] = q */
  int i = 0;
  while (i < c->n_obs && c->obs[i].kf < kf) i++;
  if (i < c->n_obs && c->obs[i].kf == kf) { c->obs[i].q = q; return; }
  if (c->n_obs == c->cap_obs) {
    c->cap_obs = c->cap_obs ? c->cap_obs * 2 : 4;
    c->obs = (tfo_obs*)realloc(c->obs, sizeof(tfo_obs) * c->cap_obs);
  }
  memmove(c->obs + i + 1, c->obs + i, sizeof(tfo_obs) * (c->n_obs - i));
  c->obs[i].kf = kf; c->obs[i].q = q;
  c->n_obs++;
}
/* MobileFusion::RetractObservations, the chunk side (GCFusion/MobileFusion.cpp:252-260): for every listed chunk that
 * exists, chunk->observations.erase(frame_id).  Returns the number of observations erased. */
int64_t tfo_volume_retract_observations(tfo_volume* v, int kf, const int32_t* ids, int64_t n) {
  int64_t erased = 0;
  for (int64_t k = 0; k < n; k++) {
    tfo_chunk* c = vol_get(v, ids + 3 * k);
    if (!c) continue; /* !manager.HasChunk(kf.validChunks[i]) */
    for (int i = 0; i < c->n_obs; i++)
      if (c->obs[i].kf == kf) {
        memmove(c->obs + i, c->obs + i + 1, sizeof(tfo_obs) * (c->n_obs - i - 1));
        c->n_obs--;
        erased++;
        break;
      }
  }
  return erased;
}
int64_t tfo_volume_get_observations(const tfo_volume* v, const int id[3], int32_t* kf, float* q,
                                    int64_t cap) {
  tfo_chunk* c = vol_get(v, id);
  if (!c) return -1;
  for (int i = 0; i < c->n_obs && i < cap; i++) { kf[i] = c->obs[i].kf; q[i] = c->obs[i].q; }
  return c->n_obs;
}
int64_t tfo_volume_num_dirty(const tfo_volume* v) { return v->dirty.live; }
int64_t tfo_volume_list_dirty(const tfo_volume* v, int32_t* ids, int64_t cap) {
  int64_t n = 0;
  for (int64_t i = 0; i < v->dirty.cap; i++)
    if (v->dirty.vals[i] >= 0) {
      if (n < cap) memcpy(ids + 3 * n, v->dirty.keys + 3 * i, 12);
      n++;
    }
  return n;
}
void tfo_volume_clear_dirty(tfo_volume* v) {
  map_free(&v->dirty);
  map_init(&v->dirty, 1 << 12);
}
void tfo_volume_get_rowstats(const tfo_volume* v, tfo_rowstats* out) { *out = v->stats; }
void tfo_volume_clear_rowstats(tfo_volume* v) { memset(&v->stats, 0, sizeof(v->stats)); }

/* Chisel::PrepareIntersectChunks (Structure/Chisel.h:103-140). */
int64_t tfo_prepare(tfo_volume* v, const float* depth, const float pose[12], int32_t* ids,
                    uint8_t* is_new, int64_t cap) {
  /* :116 GetBoundaryChunkID fills min/maxChunkID; :124 selection recomputes the same bbox */
  int64_t n = tfo_select(depth, &v->cam, &v->ig, pose, v->res, ids, cap, NULL);
  if (n > cap) return -n;
  for (int64_t i = 0; i < n; i++) { /* :130-138 */
    int nw = 0;
    if (!vol_get(v, ids + 3 * i)) { nw = 1; vol_create_chunk(v, ids + 3 * i); }
    is_new[i] = (uint8_t)nw;
  }
  return n;
}

/* Chisel::IntegrateDepthScanColor 10-arg (Structure/Chisel.h:218-249).  The reference
 * runs the chunk loop through chisel::parallel_for (threading/Threading.h:36-54); chunks are
 * independent, so the thread count only affects timing (used for the CPU baseline). */
int tfo_integrate(tfo_volume* v, const float* depth, const uint8_t* rgba, const float* quality,
                  const float pose[12], const int32_t* ids, int64_t n, int integrate_flag,
                  int keyframe_id, uint8_t* needs_update, float* quality_out) {
  float cen[3 * TFO_CHUNK_VOXELS];
  tfo_centroids(pose, v->res, cen); /* :226 */
  if (n < 1) return 0;               /* :228 */
  int missing = 0;
  int64_t rt = 0, rc = 0, cu = 0;
  /* chisel::parallel_for (threading/Threading.h:36-54): stretches of max(1000, N / nthreads) items, one
   * thread per stretch -- with fewer than 1000 chunks the reference runs this loop on one thread */
  int64_t group = 1000;
  const int T = tfo_parallel_for_plan(n, v->nthreads, 1000, &group);
#pragma omp parallel for schedule(static, group) num_threads(T) reduction(+ : missing, rt, rc, cu)
  for (int64_t i = 0; i < n; i++) {
    tfo_chunk* c = vol_get(v, ids + 3 * i);
    if (!c) { missing++; continue; } /* reference: chunks.at() would throw */
    float q = 0.0f;
    tfo_rowstats st = {0, 0, 0};
    int id[3] = {ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]};
    int upd = tfo_voxel_update_k(v->kernel, depth, rgba, quality, &v->cam, &v->ig, pose, integrate_flag,
                                 id, v->res, cen, c->sdf, c->weight, c->color, &q, &st);
    needs_update[i] = (uint8_t)(needs_update[i] || upd); /* :241 */
    if (quality_out) quality_out[i] = q;
    if (keyframe_id >= 0 && q > 0.0f && needs_update[i]) chunk_set_obs(c, keyframe_id, q); /* :244-247 */
    rt += st.rows_tsdf; rc += st.rows_color; cu += st.chunks_updated;
  }
  v->stats.rows_tsdf += rt;
  v->stats.rows_color += rc;
  v->stats.chunks_updated += cu;
  return missing ? -missing : 0;
}

/* Chisel::FinalizeIntegrateChunks + GarbageCollect (Structure/Chisel.h:184-216, 472-477) */
int64_t tfo_finalize(tfo_volume* v, const int32_t* ids, const uint8_t* needs_update,
                     const uint8_t* is_new, int64_t n, int32_t* valid_ids) {
  static const int nb[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0},
                               {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
  int64_t nv = 0;
  for (int64_t i = 0; i < n; i++) {
    if (needs_update[i]) {
      for (int k = 0; k < 7; k++) {
        int32_t id[3] = {ids[3 * i] + nb[k][0], ids[3 * i + 1] + nb[k][1], ids[3 * i + 2] + nb[k][2]};
        map_put(&v->dirty, id, 1);
      }
      if (valid_ids) memcpy(valid_ids + 3 * nv, ids + 3 * i, 12);
      nv++;
    }
  }
  for (int64_t i = 0; i < n; i++) /* garbage = new and not updated; erased after all marks */
    if (!needs_update[i] && is_new[i]) {
      vol_remove_chunk(v, ids + 3 * i);
      map_erase(&v->dirty, ids + 3 * i);
    }
  return nv;
}

/* Chisel::IntegrateDepthScanColor 5-arg (Structure/Chisel.h:453-468) */
int64_t tfo_integrate_frame(tfo_volume* v, const float* depth, const uint8_t* rgba,
                            const float pose[12], int64_t* n_selected) {
  int64_t cap = 1 << 16, n;
  int32_t* ids = NULL;
  uint8_t* is_new = NULL;
  for (;;) {
    ids = (int32_t*)malloc(sizeof(int32_t) * 3 * cap);
    is_new = (uint8_t*)malloc(cap);
    n = tfo_prepare(v, depth, pose, ids, is_new, cap);
    if (n >= 0) break;
    free(ids); free(is_new);
    cap = -n + 16;
  }
  uint8_t* needs = (uint8_t*)calloc(n ? n : 1, 1);
  tfo_integrate(v, depth, rgba, NULL, pose, ids, n, 1, -1, needs, NULL);
  int64_t nv = tfo_finalize(v, ids, needs, is_new, n, NULL);
  if (n_selected) *n_selected = n;
  free(ids); free(is_new); free(needs);
  return nv;
}

/* ------------------------------------------------------------------------------------ */
/* atlas                                                                                */
/* ------------------------------------------------------------------------------------ */
struct tfo_atlas {
  int aw, ah;
  uint64_t pw, ph;
  uint64_t loc_next;
  uint8_t* buf;
};

tfo_atlas* tfo_atlas_create(float res, int atlas_w, int atlas_h) {
  tfo_atlas* a = (tfo_atlas*)calloc(1, sizeof(tfo_atlas));
  a->aw = atlas_w > 0 ? atlas_w : TFO_ATLAS_DIM;
  a->ah = atlas_h > 0 ? atlas_h : TFO_ATLAS_DIM;
  a->pw = (uint64_t)floor((double)(4800.0f * res)); /* Atlas.h:62-65: int*float, std::floor */
  a->ph = (uint64_t)floor((double)(3600.0f * res));
  a->buf = (uint8_t*)calloc((size_t)a->aw * a->ah, 3); /* Atlas.cpp:34-36 */
  return a;
}
void tfo_atlas_destroy(tfo_atlas* a) { if (a) { free(a->buf); free(a); } }
int tfo_atlas_patch_w(const tfo_atlas* a) { return (int)a->pw; }
int tfo_atlas_patch_h(const tfo_atlas* a) { return (int)a->ph; }
uint64_t tfo_atlas_loc_next(const tfo_atlas* a) { return a->loc_next; }
uint8_t* tfo_atlas_buffer(tfo_atlas* a) { return a->buf; }

/* Atlas::AddPatch, new-patch branch (Atlas.cpp:44-59).  texloc is assigned before the
 * overflow test; on overflow the reference throws and loc_next is unchanged. */
int tfo_atlas_alloc(tfo_atlas* a, uint64_t* texloc) {
  *texloc = a->loc_next;
  uint64_t x = a->loc_next % (uint64_t)a->aw;
  uint64_t y = a->loc_next / (uint64_t)a->aw;
  if (x >= (uint64_t)a->aw || y >= (uint64_t)a->ah) return -1;
  if (x + a->pw >= (uint64_t)a->aw) { x = 0; y += a->ph; }
  else x += a->pw;
  a->loc_next = x + y * (uint64_t)a->aw;
  return 0;
}

/* Patch::bilinear (Patch.cpp:110-145): note c2 reused where c4 belongs (:125-128). */
static void rgb_at(const uint8_t* rgb, int W, int H, int y, int x, float c[3]) {
  /* cv::Mat::at is unchecked pointer arithmetic; x==W reads the next row's first pixel.
   * Reads past the allocation (undefined in the reference) return 0 here. */
  int64_t i = (int64_t)y * W + x;
  if (i < 0 || i >= (int64_t)W * H) { c[0] = c[1] = c[2] = 0.0f; return; }
  c[0] = (float)rgb[3 * i]; c[1] = (float)rgb[3 * i + 1]; c[2] = (float)rgb[3 * i + 2];
}
static void bilinear_rgb(const uint8_t* rgb, int W, int H, float lx, float ly, float out[3]) {
  int x = (int)floorf(lx), y = (int)floorf(ly);
  float c1[3], c2[3], c3[3];
  if (x < W - 1 && y < H - 1) {
    rgb_at(rgb, W, H, y, x, c1); rgb_at(rgb, W, H, y, x + 1, c2); rgb_at(rgb, W, H, y + 1, x, c3);
    float ax = (float)(x + 1) - lx, bx = lx - (float)x;
    float ay = (float)(y + 1) - ly, by = ly - (float)y;
    for (int k = 0; k < 3; k++) {
      float t = (c1[k] * ax) * ay;
      t = t + (c2[k] * bx) * ay;
      t = t + (c3[k] * ax) * by;
      t = t + (c2[k] * bx) * by;
      out[k] = t;
    }
  } else if (x < W - 1 && y == H - 1) {
    rgb_at(rgb, W, H, y, x, c1); rgb_at(rgb, W, H, y, x + 1, c2);
    float ax = (float)(x + 1) - lx, bx = lx - (float)x;
    for (int k = 0; k < 3; k++) out[k] = c1[k] * ax + c2[k] * bx;
  } else if (x == W - 1 && y < H - 1) {
    rgb_at(rgb, W, H, y, x, c1); rgb_at(rgb, W, H, y + 1, x, c2);
    float ay = (float)(y + 1) - ly, by = ly - (float)y;
    for (int k = 0; k < 3; k++) out[k] = c1[k] * ay + c2[k] * by;
  } else {
    rgb_at(rgb, W, H, y, x, out);
  }
}
static float f_at(const float* img, int W, int H, int y, int x) {
  int64_t i = (int64_t)y * W + x;
  if (i < 0 || i >= (int64_t)W * H) return 0.0f;
  return img[i];
}
/* Patch::bilinear_depth (Patch.cpp:147-170) */
static float bilinear_f(const float* img, int W, int H, float lx, float ly) {
  int x = (int)floorf(lx), y = (int)floorf(ly);
  if (x < W - 1 && y < H - 1) {
    float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y, x + 1), c3 = f_at(img, W, H, y + 1, x);
    float ax = (float)(x + 1) - lx, bx = lx - (float)x;
    float ay = (float)(y + 1) - ly, by = ly - (float)y;
    float t = (c1 * ax) * ay;
    t = t + (c2 * bx) * ay;
    t = t + (c3 * ax) * by;
    t = t + (c2 * bx) * by;
    return t;
  } else if (x < W - 1 && y == H - 1) {
    float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y, x + 1);
    return c1 * ((float)(x + 1) - lx) + c2 * (lx - (float)x);
  } else if (x == W - 1 && y < H - 1) {
    float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y + 1, x);
    return c1 * ((float)(y + 1) - ly) + c2 * (ly - (float)y);
  }
  return f_at(img, W, H, y, x);
}

/* Patch::CalculateTexCoords (Patch.cpp:40-108).  T = f32(SE3d inverse) row-major 4x4. */
int tfo_patch_project(const float* verts, const float* colors, int64_t n_v, const float T[16],
                      const uint8_t* rgb, const float* depth, const tfo_camera* cam,
                      float* texcoord, float* texcolor, int32_t bbox[4], int* wrong_mapping,
                      int64_t* n_caution) {
  int flag = 0;
  const int W = cam->width, H = cam->height;
  const int fxi = (int)cam->fx, fyi = (int)cam->fy, cxi = (int)cam->cx, cyi = (int)cam->cy;
  float minX = (float)W, maxX = 0.0f, minY = (float)H, maxY = 0.0f;
  int64_t depth_cmp = 0, color_cmp = 0, ncau = 0;
  for (int64_t i = 0; i < n_v; i++) {
    const float* v = verts + 3 * i;
    float vl[3];
    for (int r = 0; r < 3; r++) { /* T * (v,1): column-sequential accumulation */
      float s = T[4 * r] * v[0];
      s = s + T[4 * r + 1] * v[1];
      s = s + T[4 * r + 2] * v[2];
      s = s + T[4 * r + 3] * 1.0f;
      vl[r] = s;
    }
    float dist = vl[2];
    float x = vl[0] / vl[2], y = vl[1] / vl[2];
    float cX = (float)((double)(x * (float)fxi + (float)cxi) + 0.5); /* :55-56 */
    float cY = (float)((double)(y * (float)fyi + (float)cyi) + 0.5);
    if (cX < 0 || cX >= (float)W || cY < 0 || cY >= (float)H) { flag = -1; ncau++; }
    if (cX < 0) cX = 0;
    if (cX >= (float)W) cX = (float)W;
    if (cY < 0) cY = 0;
    if (cY >= (float)H) cY = (float)H;
    texcoord[2 * i] = cX; texcoord[2 * i + 1] = cY;
    minX = minX < cX ? minX : cX; maxX = maxX > cX ? maxX : cX;
    minY = minY < cY ? minY : cY; maxY = maxY > cY ? maxY : cY;
    float tc[3];
    bilinear_rgb(rgb, W, H, cX, cY, tc);
    for (int k = 0; k < 3; k++) { tc[k] = tc[k] / 255.0f; texcolor[3 * i + k] = tc[k]; }
    float dpt = bilinear_f(depth, W, H, cX, cY);
    float d0 = tc[0] - colors[3 * i], d1 = tc[1] - colors[3 * i + 1], d2 = tc[2] - colors[3 * i + 2];
    float nrm = sqrtf(d0 * d0 + (d1 * d1 + d2 * d2));
    if ((double)nrm > 0.6) color_cmp++;
    if ((double)fabsf(dist - dpt) > 0.7) depth_cmp++;
  }
  *wrong_mapping = ((double)depth_cmp > 0.3 * (double)n_v) || ((double)color_cmp > 0.3 * (double)n_v);
  if (maxX >= minX && maxY >= minY) {
    /* cv::Rect(float...) truncates each argument; (a & b) intersection (:98-99) */
    int ax = (int)(minX - 2.0f), ay = (int)(minY - 2.0f);
    int aw = (int)(maxX - minX + 5.0f), ah = (int)(maxY - minY + 5.0f);
    int x1 = ax > 0 ? ax : 0, y1 = ay > 0 ? ay : 0;
    int x2 = (ax + aw) < (W - 1) ? (ax + aw) : (W - 1);
    int y2 = (ay + ah) < (H - 1) ? (ay + ah) : (H - 1);
    int w = x2 - x1, h = y2 - y1;
    if (w <= 0 || h <= 0) { x1 = y1 = w = h = 0; }
    bbox[0] = x1; bbox[1] = y1; bbox[2] = w; bbox[3] = h;
    for (int64_t i = 0; i < n_v; i++) {
      texcoord[2 * i] -= (float)x1;
      texcoord[2 * i + 1] -= (float)y1;
    }
  } else {
    bbox[0] = bbox[1] = bbox[2] = bbox[3] = 0;
  }
  if (n_caution) *n_caution = ncau;
  return flag;
}

/* cv::resize(src, dst, dst.size()) for CV_8UC3, INTER_LINEAR (third-party arithmetic, not in
 * the reference tree: OpenCV, README.md:87; restated from its published algorithm --
 * PARITY UNPINNED): coordinates (dx+0.5)*scale-0.5, 11-bit coefficients, horizontal pass in
 * int32, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2. */
static int cv_round(double v) { return (int)lrint(v); }
static void resize_linear_8uc3(const uint8_t* src, int sstep, int sw, int sh, uint8_t* dst,
                               int dstep, int dw, int dh) {
  double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
  double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
  int* rows0 = (int*)malloc(sizeof(int) * dw * 3);
  int* rows1 = (int*)malloc(sizeof(int) * dw * 3);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = (short)cv_round((double)((1.f - fx) * 2048.f));
    ialpha[2 * dx + 1] = (short)cv_round((double)(fx * 2048.f));
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    int sy0 = sy, sy1 = sy + 1;
    /* vertical border: indices clamped into the source, coefficients kept */
    if (sy0 < 0) sy0 = 0; if (sy0 > sh - 1) sy0 = sh - 1;
    if (sy1 < 0) sy1 = 0; if (sy1 > sh - 1) sy1 = sh - 1;
    short b0 = (short)cv_round((double)((1.f - fy) * 2048.f));
    short b1 = (short)cv_round((double)(fy * 2048.f));
    const uint8_t* S0 = src + (size_t)sy0 * sstep;
    const uint8_t* S1 = src + (size_t)sy1 * sstep;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < sw ? sx + 1 : sx;
      for (int k = 0; k < 3; k++) {
        rows0[3 * dx + k] = S0[3 * sx + k] * ialpha[2 * dx] + S0[3 * sx1 + k] * ialpha[2 * dx + 1];
        rows1[3 * dx + k] = S1[3 * sx + k] * ialpha[2 * dx] + S1[3 * sx1 + k] * ialpha[2 * dx + 1];
      }
    }
    uint8_t* D = dst + (size_t)dy * dstep;
    for (int x = 0; x < dw * 3; x++) {
      int val = (((b0 * (rows0[x] >> 4)) >> 16) + ((b1 * (rows1[x] >> 4)) >> 16) + 2) >> 2;
      D[x] = (uint8_t)(val < 0 ? 0 : (val > 255 ? 255 : val));
    }
  }
  free(xofs); free(ialpha); free(rows0); free(rows1);
}

/* Atlas::UpdateBuffer (Atlas.cpp:71-91) with Patch::SetImage's ROI (Patch.cpp:172-175). */
int tfo_atlas_blit(tfo_atlas* a, uint64_t texloc, const uint8_t* rgb, int img_w, int img_h,
                   const int32_t bbox[4], float ratio[2]) {
  (void)img_h;
  int cols = bbox[2], rows = bbox[3];
  uint64_t ox = texloc % (uint64_t)a->aw, oy = texloc / (uint64_t)a->aw;
  if ((uint64_t)cols > a->pw) ratio[0] = (float)a->pw / (float)cols;
  if ((uint64_t)rows > a->ph) ratio[1] = (float)a->ph / (float)rows;
  size_t astep = (size_t)a->aw * 3;
  const uint8_t* src = rgb + ((size_t)bbox[1] * img_w + bbox[0]) * 3;
  if (cols <= 0 || rows <= 0) return 0;
  if (ratio[0] < 1 || ratio[1] < 1) {
    if (ox + a->pw > (uint64_t)a->aw || oy + a->ph > (uint64_t)a->ah) return -1;
    resize_linear_8uc3(src, img_w * 3, cols, rows, a->buf + oy * astep + ox * 3, (int)astep,
                       (int)a->pw, (int)a->ph);
  } else {
    if (ox + cols > (uint64_t)a->aw || oy + rows > (uint64_t)a->ah) return -1;
    for (int r = 0; r < rows; r++)
      memcpy(a->buf + (oy + r) * astep + ox * 3, src + (size_t)r * img_w * 3, (size_t)cols * 3);
  }
  return 0;
}

/* Chisel::GeneratePatches hot range (Chisel.cpp:153-186) */
void tfo_atlas_hot_range(const tfo_atlas* a, const uint64_t* texlocs, int64_t n,
                         uint64_t* hot_start, uint64_t* hot_end) {
  uint64_t lo = (uint64_t)a->aw * (uint64_t)a->ah, hi = 0;
  for (int64_t i = 0; i < n; i++) {
    if (texlocs[i] < lo) lo = texlocs[i];
    if (texlocs[i] > hi) hi = texlocs[i];
  }
  *hot_start = (lo / (uint64_t)a->aw) * (uint64_t)a->aw;
  *hot_end = (hi / (uint64_t)a->aw + a->ph) * (uint64_t)a->aw;
}

/* ===================================================================================== */
/* a15  Chisel::CompensateColor (Structure/Chisel.cpp:198-286)                            */
/* ===================================================================================== */
/* eigen-decomposition of a symmetric 3x3 (cyclic Jacobi, double): A = V diag(w) V^T */
static void sym3_eig(const float A[9], double w[3], double V[9]) {
  double a[9];
  for (int i = 0; i < 9; i++) { a[i] = (double)A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = a[3 * p + q];
        if (apq == 0.0) continue;
        double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < 3; k++) { /* A <- A J */
          double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = cs * akp - sn * akq;
          a[3 * k + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; k++) { /* A <- J^T A */
          double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = cs * apk - sn * aqk;
          a[3 * q + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[3 * k + p], vkq = V[3 * k + q];
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
/* the transfer matrix of Chisel.cpp:247-268 from the two covariances */
void tfo_color_transfer(const float cov_src[9], const float cov_tar[9], float T[9]) {
  double ws[3], U[9], Ut[9], ct[9], D[9] = {0}, M1[9], M2[9], media[9];
  sym3_eig(cov_src, ws, U);
  for (int i = 0; i < 3; i++) {
    D[4 * i] = (double)(float)sqrt(ws[i] > 0.0 ? ws[i] : 0.0); /* eigenvalues().cwiseSqrt() in f32 */
    for (int j = 0; j < 3; j++) { Ut[3 * i + j] = U[3 * j + i]; ct[3 * i + j] = (double)cov_tar[3 * i + j]; }
  }
  /* media = diag_src * oth_src^T * cov_tar * oth_src * diag_src (:252-254) */
  mat3_mul(D, Ut, M1); mat3_mul(M1, ct, M2); mat3_mul(M2, U, M1); mat3_mul(M1, D, media);
  float mediaf[9];
  for (int i = 0; i < 9; i++) mediaf[i] = (float)media[i];
  for (int i = 0; i < 3; i++) /* symmetrise the rounding so that the solver sees a symmetric matrix */
    for (int j = i + 1; j < 3; j++) mediaf[3 * j + i] = mediaf[3 * i + j];
  double wm[3], Um[9], Umt[9], Dm[9] = {0}, Di[9] = {0};
  sym3_eig(mediaf, wm, Um);
  for (int i = 0; i < 3; i++) {
    Dm[4 * i] = (double)(float)sqrt(wm[i] > 0.0 ? wm[i] : 0.0);
    Di[4 * i] = (double)(float)(1.0 / ((double)(float)D[4 * i] + 1e-2)); /* 1 / (diag + 1e-2), double literal (:260-262) */
    for (int j = 0; j < 3; j++) Umt[3 * i + j] = Um[3 * j + i];
  }
  /* T = oth_src * diag_src * oth_media * diag_media * oth_media^T * diag_src * oth_src^T (:264-266) */
  double A1[9], A2[9];
  mat3_mul(U, Di, A1); mat3_mul(A1, Um, A2); mat3_mul(A2, Dm, A1); mat3_mul(A1, Umt, A2);
  mat3_mul(A2, Di, A1); mat3_mul(A1, Ut, A2);
  for (int i = 0; i < 9; i++) T[i] = (float)A2[i];
}

int64_t tfo_color_compensate(int64_t n_patches, const int32_t* frame_ids, const uint8_t* wrong_mapping,
                             uint8_t* has_adjusted, const int64_t* voff, const float* texcolor,
                             const float* meshcolor, float* labs, float* out_T, int32_t* out_cluster) {
  int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_patches > 0 ? n_patches : 1));
  int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_patches > 0 ? n_patches : 1));
  int64_t ncl = 0;
  for (int64_t p = 0; p < n_patches; p++) { /* :199-214 */
    cl[p] = -1;
    if (has_adjusted[p]) continue;
    int64_t k;
    for (k = 0; k < ncl; k++)
      if (frame_ids[first[k]] == frame_ids[p]) break;
    if (k == ncl) first[ncl++] = (int32_t)p;
    cl[p] = (int32_t)k;
  }
  for (int64_t c = 0; c < ncl; c++) {
    /* computeMeanAndCov (Patch.cpp:342-348): rowwise mean, centre, src * src^T / (N - 1); f32, in order */
    float mean[2][3] = {{0, 0, 0}, {0, 0, 0}}, cov[2][9];
    int64_t n = 0;
    for (int64_t p = 0; p < n_patches; p++) {
      if (cl[p] != c || wrong_mapping[p]) continue;
      for (int64_t k = voff[p]; k < voff[p + 1]; k++) {
        for (int a = 0; a < 3; a++) { mean[0][a] += texcolor[3 * k + a]; mean[1][a] += meshcolor[3 * k + a]; }
        n++;
      }
    }
    if (out_T) for (int i = 0; i < 9; i++) out_T[9 * c + i] = 0.0f;
    if (n == 0) continue; /* :242: nothing to learn from; has_adjusted stays false */
    for (int s = 0; s < 2; s++) {
      for (int a = 0; a < 3; a++) mean[s][a] = mean[s][a] / (float)n;
      for (int i = 0; i < 9; i++) cov[s][i] = 0.0f;
    }
    for (int64_t p = 0; p < n_patches; p++) {
      if (cl[p] != c || wrong_mapping[p]) continue;
      for (int64_t k = voff[p]; k < voff[p + 1]; k++)
        for (int s = 0; s < 2; s++) {
          const float* x = (s == 0 ? texcolor : meshcolor) + 3 * k;
          float d[3] = {x[0] - mean[s][0], x[1] - mean[s][1], x[2] - mean[s][2]};
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) cov[s][3 * i + j] += d[i] * d[j];
        }
    }
    for (int s = 0; s < 2; s++)
      for (int i = 0; i < 9; i++) cov[s][i] = cov[s][i] / (float)(n - 1);
    float T[9];
    tfo_color_transfer(cov[0], cov[1], T);
    if (out_T) memcpy(out_T + 9 * c, T, sizeof(T));
    for (int64_t p = 0; p < n_patches; p++) {
      if (cl[p] != c) continue;
      if (!wrong_mapping[p]) /* wrong-mapped patches end with labs cleared (:276-278) */
        for (int64_t k = voff[p]; k < voff[p + 1]; k++) {
          float d[3] = {texcolor[3 * k] - mean[0][0], texcolor[3 * k + 1] - mean[0][1], texcolor[3 * k + 2] - mean[0][2]};
          for (int i = 0; i < 3; i++) {
            float acc = T[3 * i] * d[0];
            acc = acc + T[3 * i + 1] * d[1];
            acc = acc + T[3 * i + 2] * d[2];
            labs[3 * k + i] = acc + mean[1][i];
          }
        }
      has_adjusted[p] = 1; /* :280 */
    }
  }
  if (out_cluster) memcpy(out_cluster, cl, sizeof(int32_t) * (size_t)n_patches);
  free(cl);
  free(first);
  return ncl;
}

/* ===================================================================================== */
/* f-2  Chisel::DrawMeshes (Structure/Chisel.cpp:288-355)                                 */
/* ===================================================================================== */
int64_t tfo_pack_vertices(int64_t n_patches, const uint8_t* complete, const uint8_t* wrong_mapping,
                          const uint8_t* labs_valid, const uint64_t* texloc, const float* ratio,
                          int atlas_w, int atlas_h, const int64_t* voff, const float* verts,
                          const float* colors, const float* normals, const float* texcoord,
                          const float* texcolor, const float* labs, const int64_t* ioff,
                          const uint32_t* indices, float* out_v, uint32_t* out_i, int64_t* n_indices) {
  int64_t vert_num = 0, index_num = 0;
  for (int64_t p = 0; p < n_patches; p++) {
    if (!complete[p]) continue; /* :298 */
    for (int64_t j = ioff[p]; j < ioff[p + 1]; j++) out_i[index_num++] = indices[j] + (uint32_t)vert_num; /* :300-305 */
    const float ox = (float)(texloc[p] % (uint64_t)atlas_w), oy = (float)(texloc[p] / (uint64_t)atlas_w); /* Atlas.cpp:66-69 */
    for (int64_t k = voff[p]; k < voff[p + 1]; k++) {
      float* o = out_v + 12 * vert_num;
      float tx = texcoord[2 * k], ty = texcoord[2 * k + 1];
      if (ratio[2 * p] < 1) tx = tx * ratio[2 * p];         /* :316-317 */
      if (ratio[2 * p + 1] < 1) ty = ty * ratio[2 * p + 1];
      tx = tx + ox;
      ty = ty + oy;
      o[0] = verts[3 * k]; o[1] = verts[3 * k + 1]; o[2] = verts[3 * k + 2];
      o[3] = 50.0f;
      int rgb = (int)(colors[3 * k] * 255.0f);               /* :325-328 */
      rgb = (rgb << 8) + (int)(colors[3 * k + 1] * 255.0f);
      rgb = (rgb << 8) + (int)(colors[3 * k + 2] * 255.0f);
      o[4] = (float)rgb;
      if (labs_valid[p]) {                                   /* :330-337 */
        const float a0 = labs[3 * k] - texcolor[3 * k], a1 = labs[3 * k + 1] - texcolor[3 * k + 1],
                    a2 = labs[3 * k + 2] - texcolor[3 * k + 2];
        int ad = (int)(a0 * 255.0f) + 255;
        ad = (ad << 9) + (int)(a1 * 255.0f) + 255;
        ad = (ad << 9) + (int)(a2 * 255.0f) + 255;
        o[5] = (float)ad;
      } else {
        o[5] = 0.0f;
      }
      o[6] = tx / (float)atlas_w;                            /* :340-341 */
      o[7] = ty / (float)atlas_h;
      o[8] = normals[3 * k]; o[9] = normals[3 * k + 1]; o[10] = normals[3 * k + 2];
      o[11] = wrong_mapping[p] ? 1.0f : 0.0f;
      vert_num++;
    }
  }
  if (n_indices) *n_indices = index_num;
  return vert_num;
}

/* GeneratePatches + UpdateAtlas for a batch of patches of one keyframe in one call (the loop of
 * Chisel.cpp:156-183,191-196), for the CPU baseline: no per-patch binding overhead. */
int tfo_patches_batch(tfo_atlas* a, int64_t n_patches, const uint64_t* texloc, const int64_t* voff,
                      const float* verts, const float* colors, const float* T16, const uint8_t* rgb,
                      const float* depth, const tfo_camera* cam, float* texcoord, float* texcolor) {
  for (int64_t p = 0; p < n_patches; p++) {
    int32_t bbox[4];
    int wrong = 0;
    int64_t caution = 0;
    float ratio[2] = {1.0f, 1.0f};
    tfo_patch_project(verts + 3 * voff[p], colors + 3 * voff[p], voff[p + 1] - voff[p], T16 + 16 * p, rgb, depth,
                      cam, texcoord + 2 * voff[p], texcolor + 3 * voff[p], bbox, &wrong, &caution);
    tfo_atlas_blit(a, texloc[p], rgb, cam->width, cam->height, bbox, ratio);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* meshing: ChunkManager::GenerateMeshEfficient,               */
/* extractGradientFromCubic, RecomputeMeshes; Mesh::SimplifyByClustering;                 */
/* Chisel::UpdateMeshes / CompressMeshes                                                 */
/* ------------------------------------------------------------------------------------ */
#include "mc_table.inc"

/* cubeIndexOffsets (ChunkManager.cpp:65-66): column k = corner k of a cell */
static const int kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0},
                                  {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
/* edgeIndexPairs (ChunkManager.cpp:68-75) */
static const int kEdgePair[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                     {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

struct tfo_mesh {
  int32_t id[3];
  int alive;
  int32_t nv, ni;
  float* verts;   /* [3*nv] Mesh::vertices */
  float* normals; /* [3*nv] Mesh::normals */
  float* colors;  /* [3*nv] Mesh::colors */
  uint32_t* indices; /* [ni] Mesh::indices */
  uint8_t adj[6];    /* Mesh::adj */
  int simplified;    /* Mesh::simplified */
  /* Mesh::m_patch (Structure/Patch.h:51-94) */
  int has_patch;     /* m_patch != nullptr */
  uint64_t texloc;
  int frameid, has_image, has_adjusted, wrong_mapping, caution;
  int32_t bbox[4];
  float ratio[2];
  int32_t pnv;       /* texcoord.size() */
  float* texcoord;   /* [2*pnv] */
  float* texcolor;   /* [3*pnv] */
  float* labs;       /* [3*pnv]; labs_n = 0 <=> labs.empty() */
  int32_t labs_n;
  const uint8_t* img_rgb; /* Patch::image = a non-owning ROI of the keyframe's rgb (Patch.cpp:172-175) */
  int img_w, img_h;
};

/* sdf of the voxel at chunk `cid`, voxel index `vi`; returns 0 when the chunk does not exist
 * (ChunkMap::find fails, ChunkManager.h:808-822) */
static int chunk_sdf(const tfo_volume* v, const int32_t cid[3], int vi, float* out) {
  const tfo_chunk* c = vol_get(v, cid);
  if (!c) return 0;
  *out = c->sdf[vi];
  return 1;
}

/* ChunkManager::extractGradientFromCubic (ChunkManager.cpp:277-455) with GetNeighborSDF
 * (ChunkManager.h:790-823).  cube[8] = corner sdfs of the cell, cell = (x,y,z) of the cell in its
 * chunk, k = corner of the cell whose gradient is wanted, corner_vi / corner_chunk = voxel index
 * and chunk id the corner lives in.  The three differences that stay inside the cube come from
 * cube[], the other three from the corner's own chunk or -- when the corner sits on a chunk face
 * (edgeFlag, :304-309, tested on the cell coordinate + corner offset, range 0..8) -- from the face
 * neighbour chunk (voxelNeighborIndex = the wrapped neighbour index, ChunkManager.cpp:108-157).
 * A fetched value must be < 1 (and its chunk must exist), else there is no normal.
 * grad.norm()/normalize(): Eigen fixed-size reduction x*x + (y*y + z*z); normalize() divides by
 * sqrt when the squared norm is > 0 (Eigen >= 3.3; 3.2 multiplies by the reciprocal and has no
 * zero test -- third-party arithmetic, parity unpinned). */
static int gradient_from_cubic(const tfo_volume* v, const float cube[8], const int cell[3], int k,
                               int corner_vi, const int32_t corner_chunk[3], float grad[3]) {
  float dd[6];
  const int vx = corner_vi & 7, vy = (corner_vi >> 3) & 7, vz = corner_vi >> 6;
  const int vc[3] = {vx, vy, vz};
  for (int a = 0; a < 3; a++) {
    const int near = cell[a] + kCorner[k][a];
    /* the in-cube partner: corner k with offset a flipped */
    int kk = -1;
    for (int q = 0; q < 8; q++) {
      int same = 1;
      for (int b = 0; b < 3; b++)
        if (kCorner[q][b] != (b == a ? 1 - kCorner[k][b] : kCorner[k][b])) same = 0;
      if (same) kk = q;
    }
    const int dir = kCorner[k][a] ? 1 : 0; /* 0: fetch the -a neighbour, 1: fetch the +a neighbour */
    dd[2 * a + (1 - dir)] = cube[kk];
    const int edge = dir ? (near == 7) : (near == 0);
    int nv[3] = {vc[0], vc[1], vc[2]};
    nv[a] = (vc[a] + (dir ? 1 : -1) + 8) % 8;
    const int nvi = nv[0] + nv[1] * 8 + nv[2] * 64;
    int32_t nc[3] = {corner_chunk[0], corner_chunk[1], corner_chunk[2]};
    if (edge) nc[a] += dir ? 1 : -1;
    float val;
    if (!chunk_sdf(v, nc, nvi, &val)) return 0;
    dd[2 * a + dir] = val;
    if (!(val < 1.0f)) return 0;
  }
  const float gx = dd[1] - dd[0], gy = dd[3] - dd[2], gz = dd[5] - dd[4];
  const float yz = gy * gy + gz * gz;
  float sq = gx * gx + yz;
  if (g_sum_order & 4) { const float xy = gx * gx + gy * gy; sq = xy + gz * gz; } /* alternative: sequential */
  const float g = sqrtf(sq);
  grad[0] = gx; grad[1] = gy; grad[2] = gz;
  if (sq > 0.0f) {
    if (g_sum_order & 8) { const float r = 1.0f / g; grad[0] = gx * r; grad[1] = gy * r; grad[2] = gz * r; } /* Eigen 3.2 */
    else { grad[0] = gx / g; grad[1] = gy / g; grad[2] = gz / g; }
  }
  if (g > v->res * 100.0f) return 0;
  return 1;
}

#define TFO_MESH_SLOTS (3 * 729)
#define TFO_MESH_MAX_INDICES (512 * 15)

/* ChunkManager::GenerateMeshEfficient (ChunkManager.cpp:595-1002) */
int64_t tfo_mesh_chunk(const tfo_volume* v, const int id[3], float* verts, float* normals,
                       float* colors, uint32_t* indices, int64_t* n_indices) {
  const tfo_chunk* c0 = vol_get(v, id);
  if (n_indices) *n_indices = 0;
  if (!c0) return -1;
  const float res = v->res;
  const tfo_chunk* nb[8];
  int32_t nbid[8][3];
  for (int i = 0; i < 8; i++) { /* :618-632 */
    nbid[i][0] = id[0] + (i % 2); nbid[i][1] = id[1] + (i % 4) / 2; nbid[i][2] = id[2] + i / 4;
    nb[i] = i ? vol_get(v, nbid[i]) : c0;
  }
  /* Chunk origin (Chunk.cpp:52) and CacheCentroids' table (ChunkManager.cpp:49-63) */
  const float org[3] = {(float)(8 * id[0]) * res, (float)(8 * id[1]) * res, (float)(8 * id[2]) * res};
  const float half = res * 0.5f;
  float* vbe = (float*)malloc(sizeof(float) * 3 * TFO_MESH_SLOTS * 3);
  float* cbe = vbe + 3 * TFO_MESH_SLOTS;
  float* nbe = cbe + 3 * TFO_MESH_SLOTS;
  uint8_t used[TFO_MESH_SLOTS];
  memset(used, 0, sizeof(used)); /* colorByEdge == (-1,-1,-1) <=> never written (:649,:890) */
  int64_t ni = 0;
  float cube[8]; /* cornerSDF lives across cells (:636); stale lanes are never consumed */
  for (int i = 0; i < 8; i++) cube[i] = 0.0f;
  for (int z = 0; z < 8; z++)
    for (int y = 0; y < 8; y++)
      for (int x = 0; x < 8; x++) {
        const int cell[3] = {x, y, z};
        float cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int observed = 1, pos = 0;
        int cvi[8], cch[8];
        for (int k = 0; k < 8; k++) { /* :669-720; neighbor*IndexChunkWise LUTs :77-107 */
          const int cx = x + kCorner[k][0], cy = y + kCorner[k][1], cz = z + kCorner[k][2];
          cch[k] = (cx == 8) + (cy == 8) * 2 + (cz == 8) * 4;
          cvi[k] = (cx & 7) + (cy & 7) * 8 + (cz & 7) * 64;
        }
        for (int k = 0; k < 8; k++) {
          if (!nb[cch[k]]) { observed = 0; break; }
          const float s = nb[cch[k]]->sdf[cvi[k]];
          cw[k] = nb[cch[k]]->weight[cvi[k]];
          if (s > 1.0f) { observed = 0; break; }
          cube[k] = s;
          pos += s > 0.0f;
        }
        if (!(observed && pos % 8 > 0)) continue; /* :722 */
        int index = 0;
        for (int k = 0; k < 8; k++) index |= (0.0f > cube[k]) << k; /* :726-735 */
        const unsigned long long row = tfo_mc_tri_table[index];
        if ((row & 0xF) == 0xF) continue; /* :747 */
        const float origin[3] = {org[0] + ((float)x * res + half), org[1] + ((float)y * res + half),
                                 org[2] + ((float)z * res + half)};
        float ev[12][3], en[12][3], ec[12][3];
        int nvalid[12];
        for (int e = 0; e < 12; e++) { /* :748-834 */
          nvalid[e] = 0;
          const int e0 = kEdgePair[e][0], e1 = kEdgePair[e][1];
          const float s0 = cube[e0], s1 = cube[e1];
          if (!(s0 * s1 < 0.0f)) continue;
          const float t = s0 / (s0 - s1);
          for (int a = 0; a < 3; a++) {
            const float p0 = (float)kCorner[e0][a] * res, p1 = (float)kCorner[e1][a] * res;
            ev[e][a] = p0 + t * (p1 - p0);
          }
          const int k = kEdgePair[e][fabsf(s0) > fabsf(s1)];
          float grad[3] = {0, 0, 0};
          const int ok = gradient_from_cubic(v, cube, cell, k, cvi[k], nbid[cch[k]], grad);
          if (cw[k] > 50.0f) { /* :776-827 */
            for (int a = 0; a < 3; a++) en[e][a] = grad[a];
            const uint16_t* col = nb[cch[k]]->color + cvi[k] * 4;
            const float cwt = (float)col[3];
            if (cwt > 0.0f) {
              for (int a = 0; a < 3; a++) ec[e][a] = ((float)col[a] / 255.0f) / cwt;
            } else {
              ec[e][0] = ec[e][1] = ec[e][2] = 1.0f;
            }
            nvalid[e] = ok;
          }
        }
        for (int col = 0; col < 15 && ((row >> (4 * col)) & 0xF) != 0xF; col += 3) { /* :836-918 */
          const int s[3] = {(int)((row >> (4 * col + 8)) & 0xF), (int)((row >> (4 * col + 4)) & 0xF),
                            (int)((row >> (4 * col)) & 0xF)}; /* s2, s1, s0 */
          if (!nvalid[s[0]] || !nvalid[s[1]] || !nvalid[s[2]]) continue;
          for (int q = 0; q < 3; q++) {
            const int e = s[q];
            const int bx = x + (e == 1 || e == 5 || e == 9 || e == 10);
            const int by = y + (e == 2 || e == 6 || e == 10 || e == 11);
            const int bz = z + (e == 4 || e == 5 || e == 6 || e == 7);
            const int ax = (e == 0 || e == 2 || e == 4 || e == 6) ? 0 : ((e == 1 || e == 3 || e == 5 || e == 7) ? 1 : 2);
            const int m = ax + (bx + by * 9 + bz * 81) * 3;
            for (int a = 0; a < 3; a++) {
              vbe[3 * m + a] = ev[e][a] + origin[a];
              cbe[3 * m + a] = ec[e][a];
              nbe[3 * m + a] = en[e][a];
            }
            used[m] = 1;
            if (indices && ni < TFO_MESH_MAX_INDICES) indices[ni] = (uint32_t)m;
            ni++;
          }
        }
      }
  /* de-duplication by edge slot (:886-897) */
  int ref[TFO_MESH_SLOTS];
  int64_t nv = 0;
  for (int m = 0; m < TFO_MESH_SLOTS; m++) {
    ref[m] = -1;
    if (!used[m]) continue;
    ref[m] = (int)nv;
    for (int a = 0; a < 3; a++) {
      if (verts) verts[3 * nv + a] = vbe[3 * m + a];
      if (colors) colors[3 * nv + a] = cbe[3 * m + a];
      if (normals) normals[3 * nv + a] = nbe[3 * m + a];
    }
    nv++;
  }
  if (indices)
    for (int64_t i = 0; i < ni; i++) indices[i] = (uint32_t)ref[indices[i]];
  if (n_indices) *n_indices = ni;
  free(vbe);
  return nv;
}

/* ---- allMeshes --------------------------------------------------------------------------- */
static struct tfo_mesh* vol_get_mesh(const tfo_volume* v, const int32_t* id) {
  int64_t i = map_find(&v->mesh_map, id);
  return i < 0 ? NULL : &v->meshes[v->mesh_map.vals[i]];
}
static void mesh_release(struct tfo_mesh* m) {
  free(m->verts); free(m->normals); free(m->colors); free(m->indices);
  free(m->texcoord); free(m->texcolor); free(m->labs);
  memset(m, 0, sizeof(*m));
}
static void vol_erase_mesh(tfo_volume* v, const int32_t* id) {
  int64_t i = map_find(&v->mesh_map, id);
  if (i < 0) return;
  mesh_release(&v->meshes[v->mesh_map.vals[i]]);
  map_erase(&v->mesh_map, id);
}
static void vol_meshes_clear(tfo_volume* v) {
  for (int64_t i = 0; i < v->n_meshes; i++)
    if (v->meshes[i].alive) mesh_release(&v->meshes[i]);
  v->n_meshes = 0;
  map_free(&v->mesh_map);
  map_init(&v->mesh_map, 1 << 12);
}

/* Chisel::UpdateMeshes (Chisel.h:479-481) -> ChunkManager::RecomputeMeshes (ChunkManager.cpp:232-264):
 * every chunk of meshesToUpdate that exists is re-meshed; a mesh enters allMeshes when it has
 * vertices, and a mesh that is already there stays (possibly empty).  Returns the number of chunks
 * meshed. */
int64_t tfo_update_meshes(tfo_volume* v) {
  /* the dirty chunks that exist (:236-250) */
  int64_t n = 0;
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * 3 * (v->dirty.live ? v->dirty.live : 1));
  for (int64_t i = 0; i < v->dirty.cap; i++) {
    if (v->dirty.vals[i] < 0) continue;
    const int32_t* id = v->dirty.keys + 3 * i;
    if (!vol_get(v, id)) continue;
    memcpy(ids + 3 * n, id, 12);
    n++;
  }
  /* parallel_for(meshes, GenerateMeshEfficient) (:256-259) with chisel::parallel_for's thread policy
   * (threading/Threading.h:36-54: groups of max(1000, N / nthreads) items) */
  typedef struct { int64_t nv, ni; float* vb; uint32_t* ib; } res_t;
  res_t* R = (res_t*)calloc(n ? n : 1, sizeof(res_t));
  int64_t group = 1000;
  const int T = tfo_parallel_for_plan(n, v->nthreads, 1000, &group);
#pragma omp parallel num_threads(T)
  {
    float* vb = (float*)malloc(sizeof(float) * 9 * TFO_MESH_SLOTS);
    uint32_t* ib = (uint32_t*)malloc(sizeof(uint32_t) * TFO_MESH_MAX_INDICES);
#pragma omp for schedule(static, group)
    for (int64_t i = 0; i < n; i++) {
      int64_t ni = 0;
      const int64_t nv = tfo_mesh_chunk(v, ids + 3 * i, vb, vb + 3 * TFO_MESH_SLOTS, vb + 6 * TFO_MESH_SLOTS, ib, &ni);
      R[i].nv = nv; R[i].ni = ni;
      if (nv > 0) {
        R[i].vb = (float*)malloc(sizeof(float) * 9 * nv);
        memcpy(R[i].vb, vb, sizeof(float) * 3 * nv);
        memcpy(R[i].vb + 3 * nv, vb + 3 * TFO_MESH_SLOTS, sizeof(float) * 3 * nv);
        memcpy(R[i].vb + 6 * nv, vb + 6 * TFO_MESH_SLOTS, sizeof(float) * 3 * nv);
        R[i].ib = (uint32_t*)malloc(sizeof(uint32_t) * (ni ? ni : 1));
        memcpy(R[i].ib, ib, sizeof(uint32_t) * ni);
      }
    }
    free(vb); free(ib);
  }
  for (int64_t i = 0; i < n; i++) { /* allMeshes[id] = mesh if it has vertices; an existing entry stays (:260-262) */
    const int32_t* id = ids + 3 * i;
    const int64_t nv = R[i].nv, ni = R[i].ni;
    struct tfo_mesh* m = vol_get_mesh(v, id);
    if (!m) {
      if (nv <= 0) continue;
      if (v->n_meshes == v->cap_meshes) {
        v->cap_meshes = v->cap_meshes ? v->cap_meshes * 2 : 4096;
        v->meshes = (struct tfo_mesh*)realloc(v->meshes, sizeof(struct tfo_mesh) * v->cap_meshes);
      }
      m = &v->meshes[v->n_meshes];
      memset(m, 0, sizeof(*m));
      memcpy(m->id, id, 12);
      m->alive = 1;
      map_put(&v->mesh_map, id, v->n_meshes++);
    }
    free(m->verts); free(m->normals); free(m->colors); free(m->indices);
    m->nv = (int32_t)nv; m->ni = (int32_t)ni;
    m->verts = (float*)malloc(sizeof(float) * 3 * (nv ? nv : 1));
    m->normals = (float*)malloc(sizeof(float) * 3 * (nv ? nv : 1));
    m->colors = (float*)malloc(sizeof(float) * 3 * (nv ? nv : 1));
    m->indices = (uint32_t*)malloc(sizeof(uint32_t) * (ni ? ni : 1));
    if (nv > 0) {
      memcpy(m->verts, R[i].vb, sizeof(float) * 3 * nv);
      memcpy(m->normals, R[i].vb + 3 * nv, sizeof(float) * 3 * nv);
      memcpy(m->colors, R[i].vb + 6 * nv, sizeof(float) * 3 * nv);
      memcpy(m->indices, R[i].ib, sizeof(uint32_t) * ni);
    }
    memset(m->adj, 0, 6); /* Mesh::Clear (Mesh.h:52-68) */
    m->simplified = 0;
    free(R[i].vb); free(R[i].ib);
  }
  free(R); free(ids);
  return n;
}

int64_t tfo_volume_num_meshes(const tfo_volume* v) { return v->mesh_map.live; }
int64_t tfo_volume_list_meshes(const tfo_volume* v, int32_t* ids, int64_t cap) {
  int64_t n = 0;
  for (int64_t i = 0; i < v->n_meshes; i++)
    if (v->meshes[i].alive) {
      if (n < cap) memcpy(ids + 3 * n, v->meshes[i].id, 12);
      n++;
    }
  return n;
}
/* counts first (buffers may be NULL), then the arrays */
int tfo_volume_get_mesh(const tfo_volume* v, const int id[3], int64_t* nv, int64_t* ni, float* verts,
                        float* normals, float* colors, uint32_t* indices, uint8_t adj[6], int* simplified) {
  const struct tfo_mesh* m = vol_get_mesh(v, id);
  if (!m) return -1;
  if (nv) *nv = m->nv;
  if (ni) *ni = m->ni;
  if (verts) memcpy(verts, m->verts, sizeof(float) * 3 * m->nv);
  if (normals) memcpy(normals, m->normals, sizeof(float) * 3 * m->nv);
  if (colors) memcpy(colors, m->colors, sizeof(float) * 3 * m->nv);
  if (indices) memcpy(indices, m->indices, sizeof(uint32_t) * m->ni);
  if (adj) memcpy(adj, m->adj, 6);
  if (simplified) *simplified = m->simplified;
  return 0;
}

/* Mesh::SimplifyByClustering / GetIndice (3rd_party/open_chisel/geometry/Mesh.cpp:39-83): only the
 * adjacency flags survive -- a vertex in grid cell <= 0 / >= GRID_EACH_DIM (8) of an axis marks that
 * face.  verts[3*nv], origin = Chunk::GetOrigin(), grid = resolution * (8 / GRID_EACH_DIM). */
void tfo_mesh_adjacency(const float* verts, int64_t nv, const float origin[3], float grid, uint8_t adj[6]) {
  for (int64_t i = 0; i < nv; i++)
    for (int j = 0; j < 3; j++) {
      const int pos = (int)floor((double)((verts[3 * i + j] - origin[j]) / grid));
      if (pos >= 8) adj[2 * j + 1] = 1;
      if (pos <= 0) adj[2 * j] = 1;
    }
}

/* Chisel::CompressMeshes (Structure/Chisel.cpp:112-147) on meshesToUpdate: adjacency flags of the
 * dirty chunks' meshes, exchanged with the face neighbours' meshes, then meshesToUpdate.clear().
 * out_ids (optional) receives tsdfFusion's chunksToUpdate -- the dirty keys that have a mesh
 * (GCFusion/MobileFusion.cpp:345-353) -- in ascending (x, y, z) order: the reference's order is its
 * unordered_map's iteration order, i.e. unspecified; the harness defines it. */
static int cmp_id3(const void* a, const void* b) {
  const int32_t* p = (const int32_t*)a; const int32_t* q = (const int32_t*)b;
  for (int k = 0; k < 3; k++) if (p[k] != q[k]) return p[k] < q[k] ? -1 : 1;
  return 0;
}
int64_t tfo_compress_meshes(tfo_volume* v, int32_t* out_ids, int64_t cap) {
  static const int nbh[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
  int64_t n = 0;
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * 3 * (v->dirty.live ? v->dirty.live : 1));
  for (int64_t i = 0; i < v->dirty.cap; i++) {
    if (v->dirty.vals[i] < 0) continue;
    const int32_t* id = v->dirty.keys + 3 * i;
    if (!vol_get_mesh(v, id)) continue;
    memcpy(ids + 3 * n, id, 12);
    n++;
  }
  qsort(ids, (size_t)n, 12, cmp_id3);
  const float grid = v->res * (float)(8 / 8);
  for (int64_t i = 0; i < n; i++) {
    struct tfo_mesh* m = vol_get_mesh(v, ids + 3 * i);
    if (m->simplified) continue;
    const float org[3] = {(float)(8 * m->id[0]) * v->res, (float)(8 * m->id[1]) * v->res,
                          (float)(8 * m->id[2]) * v->res};
    tfo_mesh_adjacency(m->verts, m->nv, org, grid, m->adj);
    m->simplified = 1;
  }
  for (int64_t i = 0; i < n; i++) {
    struct tfo_mesh* m = vol_get_mesh(v, ids + 3 * i);
    for (int k = 0; k < 6; k++) {
      const int32_t q[3] = {m->id[0] + nbh[k][0], m->id[1] + nbh[k][1], m->id[2] + nbh[k][2]};
      struct tfo_mesh* a = vol_get_mesh(v, q);
      if (!a || !a->simplified) continue;
      const int mm = (k % 2 == 0) ? k + 1 : k - 1;
      if (m->adj[k] && !a->adj[mm]) a->adj[mm] = 1;
      if (!m->adj[k] && a->adj[mm]) m->adj[k] = 1;
    }
  }
  if (out_ids) memcpy(out_ids, ids, sizeof(int32_t) * 3 * (size_t)(n < cap ? n : cap));
  free(ids);
  tfo_volume_clear_dirty(v);
  return n;
}

/* ------------------------------------------------------------------------------------ */
/* the atlas stage on the volume's meshes: Chisel::GeneratePatches / UpdateAtlas /         */
/* CompensateColor / DrawMeshes (Structure/Chisel.cpp:149-355) with Mesh::m_patch state    */
/* ------------------------------------------------------------------------------------ */
/* Chisel::GeneratePatches (Chisel.cpp:149-189) over `ids` in list order; kf_index[i] selects the
 * keyframe of entry i (the label view selection hands out).  Returns 0, or -1 when AddPatch
 * overflows (the entry and everything behind it stays unprocessed). */
int tfo_generate_patches(tfo_volume* v, tfo_atlas* a, const int32_t* ids, int64_t n, const int32_t* kf_index,
                         const tfo_keyframe* kfs, uint64_t hot[2]) {
  uint64_t loc_start = (uint64_t)a->aw * (uint64_t)a->ah, loc_end = 0;
  int rc = 0;
  for (int64_t i = 0; i < n; i++) {
    struct tfo_mesh* m = vol_get_mesh(v, ids + 3 * i);
    if (!m) continue; /* :157 */
    const tfo_keyframe* kf = &kfs[kf_index[i]];
    if (!m->has_patch) { /* Atlas::AddPatch (Atlas.cpp:43-64) */
      uint64_t tl;
      if (tfo_atlas_alloc(a, &tl) != 0) { rc = -1; break; }
      m->has_patch = 1;
      m->texloc = tl;
    }
    /* Patch::clear (Patch.cpp:177-189) */
    m->frameid = -1; m->has_image = 0; m->has_adjusted = 0; m->labs_n = 0;
    m->ratio[0] = m->ratio[1] = 1.0f;
    free(m->texcoord); free(m->texcolor); free(m->labs);
    m->pnv = m->nv;
    m->texcoord = (float*)malloc(sizeof(float) * 2 * (m->nv ? m->nv : 1));
    m->texcolor = (float*)malloc(sizeof(float) * 3 * (m->nv ? m->nv : 1));
    m->labs = (float*)malloc(sizeof(float) * 3 * (m->nv ? m->nv : 1));
    int64_t ncau = 0;
    int wrong = 0;
    const int flag = tfo_patch_project(m->verts, m->colors, m->nv, kf->T, kf->rgb, kf->depth, &v->cam,
                                       m->texcoord, m->texcolor, m->bbox, &wrong, &ncau);
    m->wrong_mapping = wrong;
    m->caution = flag < 0;
    m->frameid = kf->kf_id;                        /* SetFrameid */
    m->img_rgb = kf->rgb; m->img_w = v->cam.width; m->img_h = v->cam.height; m->has_image = 1; /* SetImage */
    if (m->texloc < loc_start) loc_start = m->texloc;
    if (m->texloc > loc_end) loc_end = m->texloc;
  }
  if (hot) {
    hot[0] = (loc_start / (uint64_t)a->aw) * (uint64_t)a->aw;
    hot[1] = (loc_end / (uint64_t)a->aw + a->ph) * (uint64_t)a->aw;
  }
  return rc;
}

/* Patch::complete (Patch.cpp:191-196) */
static int patch_complete(const struct tfo_mesh* m) {
  if (m->nv == 0 || !m->simplified) return 0;
  if (!m->has_patch || !m->has_image || m->pnv == 0 || m->frameid < 0) return 0;
  return 1;
}

/* Chisel::UpdateAtlas (Chisel.cpp:191-196) -> Atlas::UpdateBuffer (Atlas.cpp:71-91) */
void tfo_update_atlas(tfo_volume* v, tfo_atlas* a, const int32_t* ids, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    struct tfo_mesh* m = vol_get_mesh(v, ids + 3 * i);
    if (!m || !m->has_patch || !patch_complete(m)) continue;
    tfo_atlas_blit(a, m->texloc, m->img_rgb, m->img_w, m->img_h, m->bbox, m->ratio);
  }
}

/* every mesh in ascending (x, y, z) order (the reference iterates an unordered_map) */
static int64_t meshes_sorted(const tfo_volume* v, int32_t** out) {
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * 3 * (v->mesh_map.live ? v->mesh_map.live : 1));
  const int64_t n = tfo_volume_list_meshes(v, ids, v->mesh_map.live);
  qsort(ids, (size_t)n, 12, cmp_id3);
  *out = ids;
  return n;
}

/* Chisel::CompensateColor (Chisel.cpp:198-286) over allMeshes; returns the cluster count */
int64_t tfo_compensate_color_volume(tfo_volume* v) {
  int32_t* ids;
  const int64_t nm = meshes_sorted(v, &ids);
  int64_t np = 0, nvt = 0;
  struct tfo_mesh** pm = (struct tfo_mesh**)malloc(sizeof(void*) * (nm ? nm : 1));
  for (int64_t i = 0; i < nm; i++) {
    struct tfo_mesh* m = vol_get_mesh(v, ids + 3 * i);
    if (!m->has_patch) continue;
    pm[np++] = m;
    nvt += m->pnv;
  }
  int32_t* fid = (int32_t*)malloc(sizeof(int32_t) * (np ? np : 1));
  uint8_t* wrong = (uint8_t*)malloc(np ? np : 1);
  uint8_t* adj = (uint8_t*)malloc(np ? np : 1);
  int64_t* voff = (int64_t*)malloc(sizeof(int64_t) * (np + 1));
  float* tc = (float*)malloc(sizeof(float) * 3 * (nvt ? nvt : 1));
  float* mc = (float*)malloc(sizeof(float) * 3 * (nvt ? nvt : 1));
  float* labs = (float*)malloc(sizeof(float) * 3 * (nvt ? nvt : 1));
  voff[0] = 0;
  for (int64_t p = 0; p < np; p++) {
    fid[p] = pm[p]->frameid; wrong[p] = (uint8_t)pm[p]->wrong_mapping; adj[p] = (uint8_t)pm[p]->has_adjusted;
    voff[p + 1] = voff[p] + pm[p]->pnv;
    memcpy(tc + 3 * voff[p], pm[p]->texcolor, sizeof(float) * 3 * pm[p]->pnv);
    /* patch->mesh->colors: the mesh as it is now (same vertex count when the patch is current) */
    memcpy(mc + 3 * voff[p], pm[p]->colors, sizeof(float) * 3 * (pm[p]->pnv < pm[p]->nv ? pm[p]->pnv : pm[p]->nv));
  }
  uint8_t* adj0 = (uint8_t*)malloc(np ? np : 1);
  memcpy(adj0, adj, np);
  const int64_t ncl = tfo_color_compensate(np, fid, wrong, adj, voff, tc, mc, labs, NULL, NULL);
  for (int64_t p = 0; p < np; p++) {
    if (adj0[p]) continue; /* skipped: untouched */
    if (!adj[p]) { /* cluster without a correctly mapped vertex: labs = texcolor copy, cleared when wrong (:228-236) */
      memcpy(pm[p]->labs, pm[p]->texcolor, sizeof(float) * 3 * pm[p]->pnv);
      pm[p]->labs_n = pm[p]->wrong_mapping ? 0 : pm[p]->pnv;
      continue;
    }
    pm[p]->has_adjusted = 1;
    if (pm[p]->wrong_mapping) { pm[p]->labs_n = 0; continue; }
    memcpy(pm[p]->labs, labs + 3 * voff[p], sizeof(float) * 3 * pm[p]->pnv);
    pm[p]->labs_n = pm[p]->pnv;
  }
  free(ids); free(pm); free(fid); free(wrong); free(adj); free(adj0); free(voff); free(tc); free(mc); free(labs);
  return ncl;
}

/* Chisel::DrawMeshes (Chisel.cpp:288-355) over allMeshes (ascending id).  Buffers: out_vertices f32[12 * cap_v],
 * out_indices u32[cap_i]; returns the vertex count (even when it exceeds cap_v: nothing is written then). */
int64_t tfo_draw_meshes(tfo_volume* v, const tfo_atlas* a, float* out_vertices, uint32_t* out_indices,
                        int64_t cap_v, int64_t cap_i, int64_t* n_indices) {
  int32_t* ids;
  const int64_t nm = meshes_sorted(v, &ids);
  int64_t nv = 0, ni = 0;
  for (int64_t i = 0; i < nm; i++) {
    const struct tfo_mesh* m = vol_get_mesh(v, ids + 3 * i);
    if (!m->has_patch || !patch_complete(m)) continue;
    nv += m->nv; ni += m->ni;
  }
  if (n_indices) *n_indices = ni;
  if (nv > cap_v || ni > cap_i) { free(ids); return nv; }
  int64_t vout = 0, iout = 0;
  for (int64_t i = 0; i < nm; i++) {
    const struct tfo_mesh* m = vol_get_mesh(v, ids + 3 * i);
    if (!m->has_patch || !patch_complete(m)) continue;
    const uint8_t one = 1, wr = (uint8_t)m->wrong_mapping, lv = (uint8_t)(m->has_adjusted && m->labs_n > 0);
    const int64_t voff[2] = {0, m->nv}, ioff[2] = {0, m->ni};
    int64_t nidx = 0;
    uint32_t* tmp_i = out_indices + iout;
    tfo_pack_vertices(1, &one, &wr, &lv, &m->texloc, m->ratio, a->aw, a->ah, voff, m->verts, m->colors, m->normals,
                      m->texcoord, m->texcolor, m->labs, ioff, m->indices, out_vertices + 12 * vout, tmp_i, &nidx);
    for (int64_t k = 0; k < nidx; k++) tmp_i[k] += (uint32_t)vout; /* indices rebased by the running vertex count */
    vout += m->nv; iout += nidx;
  }
  free(ids);
  return nv;
}

/* Patch mirror of one chunk; buffers may be NULL */
int tfo_volume_get_patch(const tfo_volume* v, const int id[3], uint64_t* texloc, int* frameid, int32_t bbox[4],
                         int* flags, float ratio[2], int64_t* pnv, float* texcoord, float* texcolor, float* labs) {
  const struct tfo_mesh* m = vol_get_mesh(v, id);
  if (!m) return -1;
  if (texloc) *texloc = m->has_patch ? m->texloc : ~0ull;
  if (frameid) *frameid = m->has_patch ? m->frameid : -1;
  if (bbox) memcpy(bbox, m->bbox, 16);
  if (flags) *flags = (m->has_patch ? 1 : 0) | (m->caution ? 2 : 0) | (m->wrong_mapping ? 4 : 0) |
                      (m->has_image ? 8 : 0) | (m->has_adjusted ? 16 : 0) | ((m->labs_n > 0) ? 32 : 0);
  if (ratio) { ratio[0] = m->ratio[0]; ratio[1] = m->ratio[1]; }
  if (pnv) *pnv = m->has_patch ? m->pnv : 0;
  if (m->has_patch) {
    if (texcoord) memcpy(texcoord, m->texcoord, sizeof(float) * 2 * m->pnv);
    if (texcolor) memcpy(texcolor, m->texcolor, sizeof(float) * 3 * m->pnv);
    if (labs && m->labs_n > 0) memcpy(labs, m->labs, sizeof(float) * 3 * m->pnv);
  }
  return 0;
}

/* The textured per-frame unit (BASELINE configs[2]): IntegrateFrame, then over the
 * frame's dirty chunks UpdateMeshes -> CompressMeshes -> GeneratePatches with label = this frame ->
 * UpdateAtlas (GCFusion/MobileFusion.cpp:327-382 without the host-side view selection).  The keyframe is the
 * frame itself: rgb_scratch (u8[H*W*3], kept alive by the caller as long as the patches view it) receives the
 * RGB of its RGBA image.  chunksToUpdate is taken in ascending id order.  Returns the number of patches. */
int64_t tfo_frame_textured(tfo_volume* v, tfo_atlas* a, const float* depth, const uint8_t* rgba, const float pose[12],
                           const float pose_inv16[16], int frame_id, uint8_t* rgb_scratch) {
  tfo_integrate_frame(v, depth, rgba, pose, NULL);
  tfo_update_meshes(v);
  const int64_t cap = v->dirty.live;
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * 3 * (cap ? cap : 1));
  const int64_t n = tfo_compress_meshes(v, ids, cap);
  const size_t npix = (size_t)v->cam.width * v->cam.height;
  for (size_t i = 0; i < npix; i++) {
    rgb_scratch[3 * i] = rgba[4 * i]; rgb_scratch[3 * i + 1] = rgba[4 * i + 1]; rgb_scratch[3 * i + 2] = rgba[4 * i + 2];
  }
  tfo_keyframe kf;
  kf.rgb = rgb_scratch; kf.depth = depth; kf.kf_id = frame_id;
  memcpy(kf.T, pose_inv16, 64);
  int32_t* idx = (int32_t*)calloc(n ? n : 1, sizeof(int32_t));
  tfo_generate_patches(v, a, ids, n, idx, &kf, NULL);
  tfo_update_atlas(v, a, ids, n);
  free(ids); free(idx);
  return n;
}

/* =====================================================================================================
 * Frame pre-processing that feeds the path: BasicAPI::extractNormalMapSIMD,
 * refineDepthUseNormalSIMD, refineKeyframesSIMD, refineNewframesSIMD, checkColorQuality,
 * estimateColorQuality (BasicAPI.cpp:378-443, 506-636, 728-905), called per frame from main.cpp:117-147.
 * Images are row-major; normal maps are PLANAR (x plane, y plane, z plane) as the reference stores them.
 *
 * Harness definitions where the reference leaves a value undefined:
 *   - cv::Mat::create does not clear: the border of the normal map (row 0, the last row, column 0, the columns
 *     behind the last full 8-wide group) and the cleared bits of colorValidFlag are garbage there, ZERO here;
 *   - _mm256_rsqrt_ps is an implementation-specific approximation (relative error <= 1.5 * 2^-12, different on
 *     Intel and AMD cores): restated as the correctly rounded 1 / sqrt(x) -- results agree with any x86 run of
 *     the reference within that relative error, not bit for bit ("parity unpinned" for these two functions);
 *   - cv::cvtColor(CV_RGB2GRAY) and cv::Sobel(dx = 1, dy = 1, ksize 3, BORDER_REFLECT_101) follow OpenCV's
 *     published 8-bit algorithms (gray = (4899 R + 9617 G + 1868 B + 8192) >> 14; the mixed 3x3 derivative
 *     I(x+1,y+1) - I(x-1,y+1) - I(x+1,y-1) + I(x-1,y-1)), OpenCV itself is not in the image.
 * No FMA anywhere (operations in the order the vec8 operators apply them).
 * ===================================================================================================== */
static inline float pre_rsqrt(float x) { return 1.0f / sqrtf(x); }

/* BasicAPI::extractNormalMapSIMD (BasicAPI.cpp:849-905) */
void tfo_pre_normal_map(const float* depth, int W, int H, float fx, float fy, float cx, float cy, float* normal) {
  const size_t np = (size_t)W * H;
  memset(normal, 0, 3 * np * sizeof(float));
  const float thr = 0.3f;
  const unsigned width_dst = (unsigned)(W - 1), height_dst = (unsigned)(H - 1);
  for (unsigned i = 1; i < height_dst; i++) {
    for (unsigned j = 1; (int)j < (int)width_dst - 9; j += 8) { /* j < width_dst - 9 (unsigned vs int: values stay small) */
      for (unsigned l = 0; l < 8; l++) {
        const size_t p = (size_t)i * W + j + l;
        const float dr = depth[p + 1], db = depth[p + W], dl = depth[p - 1], dt = depth[p - W];
        const float xs = ((float)l + (float)j) - cx;      /* inc + vec8(j) - vec8(cx) */
        const float ys = (float)((float)i - cy);          /* vec8(i - cy): unsigned i converted to float */
        const float u3 = dr - dl, v3 = db - dt;
        const float u1 = ((xs * u3 + dr) + dl) / fx;
        const float u2 = (ys * u3) / fy;
        const float v1 = (xs * v3) / fx;
        const float v2 = ((ys * v3 + db) + dt) / fy;
        float nX = u2 * v3 - u3 * v2;
        float nY = u3 * v1 - u1 * v3;
        float nZ = u1 * v2 - u2 * v1;
        const float nsq = (nX * nX + nY * nY) + nZ * nZ;
        const int valid = (u3 < thr) && (u3 > -thr) && (v3 < thr) && (v3 > -thr) && (nsq > 1e-24f);
        const float r = pre_rsqrt(nsq);
        nX = nX * r; nY = nY * r; nZ = nZ * r;
        normal[p] = valid ? nX : 0.0f;
        normal[p + np] = valid ? nY : 0.0f;
        normal[p + 2 * np] = valid ? nZ : 0.0f;
      }
    }
  }
}

/* BasicAPI::refineDepthUseNormalSIMD (BasicAPI.cpp:728-781), in place */
void tfo_pre_refine_depth_normal(float* normal, float* depth, int W, int H, float fx, float fy, float cx, float cy) {
  const size_t np = (size_t)W * H;
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++) {
      const size_t p = (size_t)i * W + j;
      float vX = ((float)j - cx) / fx, vY = ((float)i - cy) / fy, vZ = 1.0f;
      const float r = pre_rsqrt((vX * vX + vY * vY) + vZ * vZ);
      vX = vX * r; vY = vY * r; vZ = vZ * r;
      const float q = (vX * normal[p] + vY * normal[p + np]) + vZ * normal[p + 2 * np];
      if (q > -0.1f && q < 0.1f) { depth[p] = 0.0f; normal[p] = 0.0f; normal[p + np] = 0.0f; normal[p + 2 * np] = 0.0f; }
    }
}

static inline void pre_view_angle(int i, int j, float fx, float fy, float cx, float cy, float v[3]) {
  /* Eigen::Vector3f((j - cx) / fx, (i - cy) / fy, 1).normalize(): squaredNorm in the fixed-size redux order
   * x*x + (y*y + z*z), division by the root (Eigen >= 3.3) -- the conventions of the rest of this file */
  const float x = ((float)j - cx) / fx, y = ((float)i - cy) / fy, z = 1.0f;
  const float yz = y * y + z * z;
  const float sq = x * x + yz;
  v[0] = x; v[1] = y; v[2] = z;
  if (sq > 0.0f) { const float n = sqrtf(sq); v[0] = x / n; v[1] = y / n; v[2] = z / n; }
}

/* BasicAPI::checkColorQuality (BasicAPI.cpp:783-806): flag = 1 where |view . normal| >= 0.2 */
void tfo_pre_color_valid(const float* normal, int W, int H, float fx, float fy, float cx, float cy, uint8_t* flag) {
  const size_t np = (size_t)W * H;
  memset(flag, 0, np);
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++) {
      const size_t p = (size_t)i * W + j;
      float v[3];
      pre_view_angle(i, j, fx, fy, cx, cy, v);
      const float q = dot3_tree(v[0], v[1], v[2], normal[p], normal[p + np], normal[p + 2 * np]);
      if ((double)fabsf(q) >= 0.2) flag[p] = 1;
    }
}

/* BasicAPI::estimateColorQuality (BasicAPI.cpp:815-847): |Sobel_xy(gray)| * |view . normal| where depth > 0,
 * the raw mixed derivative elsewhere */
void tfo_pre_color_quality(const float* depth, const float* normal, const uint8_t* rgb, int W, int H, float fx,
                           float fy, float cx, float cy, float* quality) {
  const size_t np = (size_t)W * H;
  uint8_t* gray = (uint8_t*)malloc(np);
  for (size_t p = 0; p < np; p++)
    gray[p] = (uint8_t)((4899 * (int)rgb[3 * p] + 9617 * (int)rgb[3 * p + 1] + 1868 * (int)rgb[3 * p + 2] + 8192) >> 14);
  for (int i = 0; i < H; i++) {
    const int im = i == 0 ? 1 : i - 1, ip = i == H - 1 ? H - 2 : i + 1; /* BORDER_REFLECT_101 */
    for (int j = 0; j < W; j++) {
      const int jm = j == 0 ? 1 : j - 1, jp = j == W - 1 ? W - 2 : j + 1;
      const int s = (int)gray[(size_t)ip * W + jp] - (int)gray[(size_t)ip * W + jm] - (int)gray[(size_t)im * W + jp] +
                    (int)gray[(size_t)im * W + jm];
      const size_t p = (size_t)i * W + j;
      float qv = (float)s;
      if (depth[p] > 0) {
        float v[3];
        pre_view_angle(i, j, fx, fy, cx, cy, v);
        const float vq = fabsf(dot3_tree(v[0], v[1], v[2], normal[p], normal[p + np], normal[p + 2 * np]));
        qv = fabsf(qv) * vq;
      }
      quality[p] = qv;
    }
  }
  free(gray);
}

/* the projection both refinement passes share: the pixel's vertex (x - cx) / fx * d, ..., moved by [R | t] */
static inline void pre_project(const float T[12], float cx, float cy, float fx, float fy, int i, int j, float d,
                               float V[3]) {
  const float lx = (((float)j - cx) / fx) * d, ly = (((float)i - cy) / fy) * d;
  for (int r = 0; r < 3; r++) V[r] = ((T[4 * r] * lx + T[4 * r + 1] * ly) + T[4 * r + 2] * d) + T[4 * r + 3];
}

/* BasicAPI::refineNewframesSIMD (BasicAPI.cpp:378-443): depth_new (in place) keeps the pixels whose projection
 * into the keyframe finds a depth within 5 % of their own; T = f32 of (pose_ref^-1 * pose_new)[3x4]. */
void tfo_pre_refine_newframe(const float* depth_ref, float* depth_new, int W, int H, float fx, float fy, float cx,
                             float cy, const float T[12]) {
  const float thr = 0.05f;
  const float cxh = cx + 0.5f, cyh = cy + 0.5f; /* vec8(cx + 0.5): float + double literal -> double -> float */
  const float cxh2 = (float)((double)cx + 0.5), cyh2 = (float)((double)cy + 0.5);
  (void)cxh; (void)cyh;
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++) {
      const size_t p = (size_t)i * W + j;
      const float d = depth_new[p];
      float V[3];
      pre_project(T, cx, cy, fx, fy, i, j, d, V);
      const float rx = (V[0] / V[2]) * fx + cxh2, ry = (V[1] / V[2]) * fy + cyh2;
      const int valid = (rx > 1.0f) && (rx < (float)(W - 1)) && (ry > 1.0f) && (ry < (float)(H - 1));
      float nd = 0.0f;
      if (valid) nd = depth_ref[(size_t)cvt_rne(floorf(rx) + floorf(ry) * (float)W)];
      const float diff = nd - V[2];
      const int keep = (diff > (-thr) * V[2]) && (diff < thr * V[2]);
      depth_new[p] = keep ? d : 0.0f;
    }
}

/* BasicAPI::refineKeyframesSIMD (BasicAPI.cpp:506-636): running weighted mean of the keyframe's depth with the new
 * frame's, per pixel; depth_ref and weight_ref are updated IN PLACE in row-major order, 8 pixels at a time, and
 * the nearest-neighbour fallback gathers from depth_ref itself -- at a position an earlier group may already have
 * rewritten.  T = f32 of (pose_new^-1 * pose_ref)[3x4]. */
void tfo_pre_refine_keyframe(float* depth_ref, float* weight_ref, const float* depth_new, int W, int H, float fx,
                             float fy, float cx, float cy, const float T[12]) {
  const float thr = 0.05f;
  for (int i = 0; i < H; i++)
    for (int j0 = 0; j0 < W; j0 += 8) {
      float outd[8], outw[8];
      for (int l = 0; l < 8; l++) {
        const int j = j0 + l;
        const size_t p = (size_t)i * W + j;
        const float d = depth_ref[p];
        float V[3];
        pre_project(T, cx, cy, fx, fy, i, j, d, V);
        const float rx = (V[0] / V[2]) * fx + cx, ry = (V[1] / V[2]) * fy + cy;
        const int valid = (rx > 2.0f) && (rx < (float)(W - 2)) && (ry > 2.0f) && (ry < (float)(H - 2));
        float ul = 0, ur = 0, bl = 0, br = 0, nn = 0;
        const float fxr = floorf(rx), fyr = floorf(ry);
        if (valid) {
          const int32_t q = cvt_rne(fxr + fyr * (float)W);
          ul = depth_new[q]; ur = depth_new[q + 1]; bl = depth_new[q + W]; br = depth_new[q + W + 1];
          const int32_t qn = cvt_rne(floorf(rx + 0.5f) + floorf(ry + 0.5f) * (float)W);
          nn = depth_ref[qn]; /* the keyframe's own map, possibly already refined at qn (groups before this one) */
        }
        const float dx = rx - fxr, dy = ry - fyr;
        const int smooth = ((ul - ur) < 0.1f) && ((ul - ur) > -0.1f) && ((ul - bl) < 0.1f) && ((ul - bl) > -0.1f) &&
                           ((ul - br) < 0.1f) && ((ul - br) > -0.1f);
        float bil = ((((1.0f - dx) * (1.0f - dy)) * ul + ((1.0f - dx) * dy) * ur) + (dx * (1.0f - dy)) * bl) +
                    (dx * dy) * br;
        if (!smooth) bil = nn;
        const float diff = bil - V[2];
        const int ok = (diff > (-thr) * V[2]) && (diff < thr * V[2]);
        const float scale = bil / V[2];
        const float X = V[0] * scale - T[3], Y = V[1] * scale - T[7], Z = V[2] * scale - T[11];
        const float vZ = (T[2] * X + T[6] * Y) + T[10] * Z; /* row 2 of R^T = column 2 of R */
        const float w = weight_ref[p];
        outd[l] = ok ? (d * w + vZ) / (w + 1.0f) : d;
        outw[l] = ok ? w + 1.0f : w;
      }
      for (int l = 0; l < 8; l++) {
        depth_ref[(size_t)i * W + j0 + l] = outd[l];
        weight_ref[(size_t)i * W + j0 + l] = outw[l];
      }
    }
}

/* DatasetWrapper::framePreprocess (Tools/DatasetWrapper.hpp:186-263 with UNDISTORTION / DEVIGNETTING off): the
 * loader's depth pass.  raw u16 depth -> readings beyond maximum_depth dropped (:211-221) -> metres ->
 * cv::bilateralFilter(refined_depth, filtered, 9 (7 on MobileCPU builds), 0.03, 10) (:223-233) -> written back to
 * the u16 map (:249-252).
 *
 * cv::bilateralFilter is OpenCV (README.md:87 pins commit 8f1356c), absent from /root/reference: what follows
 * restates the published CV_32FC1 algorithm of imgproc (bilateralFilter_32f + BilateralFilter_32f_Invoker):
 *   radius = d / 2, taps (i, j) with sqrt(i^2 + j^2) <= radius in row-major order, space weight
 *   (float)exp(r * r * -0.5 / sigma_space^2) in double; BORDER_REFLECT_101; the colour weight is a 4096-bin lookup
 *   table of (float)exp(v * v * -0.5 / sigma_color^2), v = i / scale_index, scale_index = 4096 / (max - min) of
 *   the image (both f32), read with linear interpolation at alpha = |val - val0| * scale_index;
 *   out = sum(val * w) / sum(w), w = space * colour; an image with max - min < FLT_EPSILON is copied.
 * PARITY UNPINNED at the rounding level: OpenCV's vector builds accumulate the tap sums in 4 / 8 partial lanes,
 * this restatement (and the device kernel, bit for bit) in tap order; no reference test holds a filtered image.
 * patchNaNs is skipped: a u16 reading cannot be NaN.  refined_out may be NULL. */
void tfo_pre_frame_depth(uint16_t* depth, int W, int H, float maximum_depth, float depth_scale, int d,
                         double sigma_color, double sigma_space, float* refined_out) {
  const size_t np = (size_t)W * H;
  float* src = (float*)malloc(np * sizeof(float));
  float* dst = (float*)malloc(np * sizeof(float));
  for (size_t p = 0; p < np; p++) {
    if ((float)depth[p] > maximum_depth * depth_scale) depth[p] = 0;
    src[p] = (float)depth[p] / depth_scale;
  }
  if (sigma_color <= 0) sigma_color = 1;
  if (sigma_space <= 0) sigma_space = 1;
  const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
  const double gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
  int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
  if (radius < 1) radius = 1;
  float mn = src[0], mx = src[0];
  for (size_t p = 1; p < np; p++) {
    if (src[p] < mn) mn = src[p];
    if (src[p] > mx) mx = src[p];
  }
  if (fabs((double)mn - (double)mx) < FLT_EPSILON) {
    memcpy(dst, src, np * sizeof(float));
  } else {
    enum { kBins = 1 << 12 };
    float* lut = (float*)malloc((kBins + 2) * sizeof(float));
    const float len = (float)((double)mx - (double)mn);
    const float scale_index = (float)kBins / len;
    float last = 1.0f;
    for (int i = 0; i < kBins + 2; i++) {
      if (last > 0.0f) {
        const double val = (double)((float)i / scale_index);
        lut[i] = (float)exp(val * val * gauss_color_coeff);
        last = lut[i];
      } else {
        lut[i] = 0.0f;
      }
    }
    const int dd = 2 * radius + 1;
    float* sw = (float*)malloc((size_t)dd * dd * sizeof(float));
    int* oi = (int*)malloc((size_t)dd * dd * sizeof(int));
    int* oj = (int*)malloc((size_t)dd * dd * sizeof(int));
    int maxk = 0;
    for (int i = -radius; i <= radius; i++)
      for (int j = -radius; j <= radius; j++) {
        const double r = sqrt((double)i * i + (double)j * j);
        if (r > radius) continue;
        sw[maxk] = (float)exp(r * r * gauss_space_coeff);
        oi[maxk] = i; oj[maxk] = j;
        maxk++;
      }
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const float val0 = src[(size_t)y * W + x];
        float sum = 0.0f, wsum = 0.0f;
        for (int k = 0; k < maxk; k++) {
          int yy = y + oi[k], xx = x + oj[k];
          /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
          if (H == 1) yy = 0; else { while (yy < 0 || yy >= H) yy = yy < 0 ? -yy : 2 * (H - 1) - yy; }
          if (W == 1) xx = 0; else { while (xx < 0 || xx >= W) xx = xx < 0 ? -xx : 2 * (W - 1) - xx; }
          const float val = src[(size_t)yy * W + xx];
          float alpha = fabsf(val - val0) * scale_index;
          int idx = (int)floorf(alpha);
          alpha -= (float)idx;
          const float w = sw[k] * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]));
          sum += val * w;
          wsum += w;
        }
        dst[(size_t)y * W + x] = sum / wsum;
      }
    free(lut); free(sw); free(oi); free(oj);
  }
  for (size_t p = 0; p < np; p++) {
    depth[p] = (uint16_t)(dst[p] * depth_scale); /* float -> unsigned short (:250-251) */
    if (refined_out) refined_out[p] = dst[p];
  }
  free(src); free(dst);
}
