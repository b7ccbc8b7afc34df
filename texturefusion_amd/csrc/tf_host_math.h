// tf_host_math.h -- per-frame scalar constants computed once on the host (C++, x86 scalar SSE,
// -ffp-contract=off) and passed to the kernels by value.  They reproduce, operation by
// operation, the scalar prologues of the reference functions cited at each line.
#pragma once

#include <math.h>
#include <stdint.h>

namespace tf {

// Prologue of ChunkManager::GetChunkIDsObservedByCamera (Structure/ChunkManager.h:398-470)
// and GetIDAt's rounding factor (:197-207).
struct SelectConsts {
  float res;
  float id_factor;   // 1.0f / (chunkSize * res)
  int step;          // stepSize (:399-403)
  float diag;        // resolutionDiagonal (:398,402)
  float diag_step;   // resolutionDiagonal * stepSize (:489)
  float dtn_coarse;  // negativeTruncation + resolutionDiagonal * stepSize (:490)
  float dtn_fine;    // negativeTruncation + resolutionDiagonal (:529)
  float rot[3][3];   // cameraPose.linear().transpose() (:429)
  float tc[3];       // rotation * cameraPose.translation() (:430)
  float r0[3], r1[3], r2[3];  // (:431-436)
  float coarse[3][8];         // diffCentroidCoarse (:444-456)
  float fine[3][8];           // diffCentroidRefine
  float pose[12];             // the frame pose (per-chunk scalars of the emitted list entries)
  float resDiag;              // sqrt(3.0f) * resolution (ProjectionIntegrator.cpp:77)
  int plain;                  // EMIT: append every entry from the front (a plain, unordered list: n_front = n_list)
};

inline SelectConsts make_select_consts(const float* p /*pose[12]*/, float res) {
  SelectConsts sc;
  sc.res = res;
  for (int i = 0; i < 12; ++i) sc.pose[i] = p[i];
  sc.resDiag = (float)(sqrt(3.0) * (double)res);
  sc.plain = 0;
  sc.id_factor = 1.0f / (8.0f * res);
  float diag = 8.0f * res / 2.0f;
  int step = 4;
  float negTrunc = (float)0.03;
  if ((double)res > 0.01) {
    diag = (float)((double)(8.0f * res) * sqrt(3.0));
    step = 1;
    negTrunc = (float)(0.05 * (double)res / 0.005);
  }
  sc.step = step;
  sc.diag = diag;
  sc.diag_step = diag * (float)step;
  sc.dtn_coarse = negTrunc + diag * (float)step;
  sc.dtn_fine = negTrunc + diag;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) sc.rot[i][j] = p[4 * j + i];
  for (int i = 0; i < 3; ++i) {  // dynamic-size product: sequential accumulation
    float s = sc.rot[i][0] * p[3];
    s = s + sc.rot[i][1] * p[7];
    s = s + sc.rot[i][2] * p[11];
    sc.tc[i] = s;
  }
  for (int i = 0; i < 3; ++i) {
    sc.r0[i] = (sc.rot[i][0] * 8.0f) * res;
    sc.r1[i] = (sc.rot[i][1] * 8.0f) * res;
    sc.r2[i] = (sc.rot[i][2] * 8.0f) * res;
  }
  const float half = res * 0.5f;
  for (int x = 0; x < 2; ++x)
    for (int y = 0; y < 2; ++y)
      for (int z = 0; z < 2; ++z) {
        const float cur[3] = {(float)(x * 8), (float)(y * 8), (float)(z * 8)};
        const int k = x + y * 2 + z * 4;
        for (int a = 0; a < 3; ++a) {
          float d = sc.rot[a][0] * cur[0];
          d = d + sc.rot[a][1] * cur[1];
          d = d + sc.rot[a][2] * cur[2];
          sc.coarse[a][k] = (d * res) * (float)step + half;
          sc.fine[a][k] = (d * res) * 1.0f + half;
        }
      }
  return sc;
}

// Scalar prologue of ProjectionIntegrator::voxelUpdateSIMD that does not depend on the chunk
// (3rd_party/open_chisel/utils/ProjectionIntegrator.cpp:74-130).
struct IntegrateConsts {
  float res;
  float half;     // halfVoxel (Chisel.cpp:55-58)
  float resDiag;  // sqrt(3.0f) * resolution, double sqrt (:77)
  float thrCol;   // resolutionDiagonal / 2 + 0.01 (:101)
  float nthrCol;
  float cxs, cys; // cx + 0.5, cy + 0.5 (:114-115)
  float lower;    // -0.03 (:314)
  float sigma;    // 1e-4 (:126)
  float qoob;     // -99999999999 (:222)
  int flag;       // integrateFlag
  uint32_t dbg;   // ablation switches for performance triage (0 in normal operation)
};

inline IntegrateConsts make_integrate_consts(float cxi, float cyi, float res, int flag) {
  IntegrateConsts kc;
  kc.res = res;
  kc.half = res * 0.5f;
  kc.resDiag = (float)(sqrt(3.0) * (double)res);
  kc.thrCol = (float)((double)(kc.resDiag / 2.0f) + 0.01);
  kc.nthrCol = -kc.thrCol;
  kc.cxs = (float)((double)cxi + 0.5);
  kc.cys = (float)((double)cyi + 0.5);
  kc.lower = (float)(-0.03);
  kc.sigma = (float)1e-4;
  kc.qoob = (float)(-99999999999.0);
  kc.flag = flag;
  kc.dbg = 0;
  return kc;
}

}  // namespace tf
