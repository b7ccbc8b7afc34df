// tf_atlas.hip -- texture-atlas side of the path: slot allocator, keyframe cache, per-patch
// vertex projection (Patch::CalculateTexCoords, Structure/Patch.cpp:40-108) and the patch
// blit / resample into the device-resident atlas (Atlas::UpdateBuffer, Structure/Atlas.cpp:71-91).
//
// One workgroup per patch.  k_patch_project streams the patch's vertices (coalesced 12-B
// reads), gathers the keyframe image bilinearly, reduces the bounding box in LDS.
// k_atlas_blit stages the keyframe ROI in LDS (the ROI is tens of pixels on a side) and writes
// the atlas slot rows; the resize branch restates cv::resize INTER_LINEAR for 8UC3 in the
// same 11-bit fixed point (third-party arithmetic -- parity unpinned).
#include <math.h>
#include <string.h>
#include <vector>

#include "tf_volume.h"

#pragma clang fp contract(off)

namespace tf {

struct PatchIn {       // one per patch, uploaded
  float T[16];         // f32(SE3d.inverse().matrix()), row-major
  const uint8_t* rgb;  // keyframe rgb u8[H][W][3]
  const float* depth;  // keyframe depth f32[H][W]
  int64_t v0, v1;      // vertex range
  uint64_t texloc;
};

struct PatchOut {  // one per patch, downloaded
  int32_t bbox[4];
  int32_t flags;  // bit0: CalculateTexCoords returned -1; bit1: wrong_mapping
  float ratio[2];
  int32_t n_caution;
};

// cv::Mat::at is unchecked pointer arithmetic: x == W lands on the next row.  Reads past the
// image (undefined in the reference) return 0.
__device__ __forceinline__ void rgb_at(const uint8_t* rgb, int W, int H, int y, int x, float c[3]) {
  const long i = (long)y * W + x;
  if (i < 0 || i >= (long)W * H) { c[0] = c[1] = c[2] = 0.0f; return; }
  c[0] = (float)rgb[3 * i]; c[1] = (float)rgb[3 * i + 1]; c[2] = (float)rgb[3 * i + 2];
}
__device__ __forceinline__ float f_at(const float* img, int W, int H, int y, int x) {
  const long i = (long)y * W + x;
  if (i < 0 || i >= (long)W * H) return 0.0f;
  return img[i];
}

// Patch::bilinear (Patch.cpp:110-145) -- c2 stands where c4 belongs (:125-128).
__device__ __forceinline__ void bilinear_rgb(const uint8_t* rgb, int W, int H, float lx, float ly,
                                             float out[3]) {
  const int x = (int)floorf(lx), y = (int)floorf(ly);
  float c1[3], c2[3], c3[3];
  if (x < W - 1 && y < H - 1) {
    rgb_at(rgb, W, H, y, x, c1); rgb_at(rgb, W, H, y, x + 1, c2); rgb_at(rgb, W, H, y + 1, x, c3);
    const float ax = (float)(x + 1) - lx, bx = lx - (float)x;
    const float ay = (float)(y + 1) - ly, by = ly - (float)y;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float t = (c1[k] * ax) * ay;
      t = t + (c2[k] * bx) * ay;
      t = t + (c3[k] * ax) * by;
      t = t + (c2[k] * bx) * by;
      out[k] = t;
    }
  } else if (x < W - 1 && y == H - 1) {
    rgb_at(rgb, W, H, y, x, c1); rgb_at(rgb, W, H, y, x + 1, c2);
    const float ax = (float)(x + 1) - lx, bx = lx - (float)x;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = c1[k] * ax + c2[k] * bx;
  } else if (x == W - 1 && y < H - 1) {
    rgb_at(rgb, W, H, y, x, c1); rgb_at(rgb, W, H, y + 1, x, c2);
    const float ay = (float)(y + 1) - ly, by = ly - (float)y;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = c1[k] * ay + c2[k] * by;
  } else {
    rgb_at(rgb, W, H, y, x, out);
  }
}
// Patch::bilinear_depth (Patch.cpp:147-170)
__device__ __forceinline__ float bilinear_f(const float* img, int W, int H, float lx, float ly) {
  const int x = (int)floorf(lx), y = (int)floorf(ly);
  if (x < W - 1 && y < H - 1) {
    const float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y, x + 1), c3 = f_at(img, W, H, y + 1, x);
    const float ax = (float)(x + 1) - lx, bx = lx - (float)x;
    const float ay = (float)(y + 1) - ly, by = ly - (float)y;
    float t = (c1 * ax) * ay;
    t = t + (c2 * bx) * ay;
    t = t + (c3 * ax) * by;
    t = t + (c2 * bx) * by;
    return t;
  } else if (x < W - 1 && y == H - 1) {
    const float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y, x + 1);
    return c1 * ((float)(x + 1) - lx) + c2 * (lx - (float)x);
  } else if (x == W - 1 && y < H - 1) {
    const float c1 = f_at(img, W, H, y, x), c2 = f_at(img, W, H, y + 1, x);
    return c1 * ((float)(y + 1) - ly) + c2 * (ly - (float)y);
  }
  return f_at(img, W, H, y, x);
}

__global__ __launch_bounds__(256) void k_patch_project(const PatchIn* __restrict__ pin,
                                                       const float* __restrict__ verts,
                                                       const float* __restrict__ colors, Cam cam,
                                                       float* __restrict__ texcoord,
                                                       float* __restrict__ texcolor,
                                                       PatchOut* __restrict__ pout) {
  const PatchIn P = pin[blockIdx.x];
  const int W = cam.W, H = cam.H;
  const float Wf = (float)W, Hf = (float)H;
  float minX = Wf, maxX = 0.0f, minY = Hf, maxY = 0.0f;  // Patch.cpp:46-49
  int dcmp = 0, ccmp = 0, ncau = 0;
  for (int64_t i = P.v0 + threadIdx.x; i < P.v1; i += 256) {
    const float vx = verts[3 * i], vy = verts[3 * i + 1], vz = verts[3 * i + 2];
    float vl[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {  // T_g_l * (v,1), accumulated column by column
      float s = P.T[4 * r] * vx;
      s = s + P.T[4 * r + 1] * vy;
      s = s + P.T[4 * r + 2] * vz;
      s = s + P.T[4 * r + 3] * 1.0f;
      vl[r] = s;
    }
    const float dist = vl[2];
    const float x = vl[0] / vl[2], y = vl[1] / vl[2];
    float cX = (float)((double)(x * cam.fxi + cam.cxi) + 0.5);  // :55-56
    float cY = (float)((double)(y * cam.fyi + cam.cyi) + 0.5);
    if (cX < 0 || cX >= Wf || cY < 0 || cY >= Hf) ncau++;  // :58-62
    if (cX < 0) cX = 0;
    if (cX >= Wf) cX = Wf;
    if (cY < 0) cY = 0;
    if (cY >= Hf) cY = Hf;
    texcoord[2 * i] = cX;
    texcoord[2 * i + 1] = cY;
    minX = minX < cX ? minX : cX; maxX = maxX > cX ? maxX : cX;
    minY = minY < cY ? minY : cY; maxY = maxY > cY ? maxY : cY;
    float tc[3];
    bilinear_rgb(P.rgb, W, H, cX, cY, tc);
#pragma unroll
    for (int k = 0; k < 3; ++k) { tc[k] = tc[k] / 255.0f; texcolor[3 * i + k] = tc[k]; }
    const float dpt = bilinear_f(P.depth, W, H, cX, cY);
    const float d0 = tc[0] - colors[3 * i], d1 = tc[1] - colors[3 * i + 1], d2 = tc[2] - colors[3 * i + 2];
    const float s12 = d1 * d1 + d2 * d2;
    const float nrm = sqrtf(d0 * d0 + s12);
    if ((double)nrm > 0.6) ccmp++;                   // :88
    if ((double)fabsf(dist - dpt) > 0.7) dcmp++;     // :89
  }
  // block reduction (min/max and integer counts are order-independent)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    float t = __shfl_xor(minX, o); minX = t < minX ? t : minX;
    t = __shfl_xor(maxX, o); maxX = t > maxX ? t : maxX;
    t = __shfl_xor(minY, o); minY = t < minY ? t : minY;
    t = __shfl_xor(maxY, o); maxY = t > maxY ? t : maxY;
    dcmp += __shfl_xor(dcmp, o); ccmp += __shfl_xor(ccmp, o); ncau += __shfl_xor(ncau, o);
  }
  __shared__ float sred[4][4];
  __shared__ int sint[4][3];
  __shared__ int sbox[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    sred[w][0] = minX; sred[w][1] = maxX; sred[w][2] = minY; sred[w][3] = maxY;
    sint[w][0] = dcmp; sint[w][1] = ccmp; sint[w][2] = ncau;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      minX = sred[k][0] < minX ? sred[k][0] : minX; maxX = sred[k][1] > maxX ? sred[k][1] : maxX;
      minY = sred[k][2] < minY ? sred[k][2] : minY; maxY = sred[k][3] > maxY ? sred[k][3] : maxY;
      dcmp += sint[k][0]; ccmp += sint[k][1]; ncau += sint[k][2];
    }
    const double nv = (double)(P.v1 - P.v0);
    const bool wrong = ((double)dcmp > 0.3 * nv) || ((double)ccmp > 0.3 * nv);  // :92-96
    PatchOut o;
    int x1 = 0, y1 = 0, bw = 0, bh = 0;
    if (maxX >= minX && maxY >= minY) {  // :98-99, cv::Rect(float..) truncation, intersection
      const int ax = (int)(minX - 2.0f), ay = (int)(minY - 2.0f);
      const int aw = (int)(maxX - minX + 5.0f), ah = (int)(maxY - minY + 5.0f);
      x1 = ax > 0 ? ax : 0; y1 = ay > 0 ? ay : 0;
      const int x2 = (ax + aw) < (W - 1) ? (ax + aw) : (W - 1);
      const int y2 = (ay + ah) < (H - 1) ? (ay + ah) : (H - 1);
      bw = x2 - x1; bh = y2 - y1;
      if (bw <= 0 || bh <= 0) { x1 = y1 = bw = bh = 0; }
    }
    o.bbox[0] = x1; o.bbox[1] = y1; o.bbox[2] = bw; o.bbox[3] = bh;
    o.flags = (ncau > 0 ? 1 : 0) | (wrong ? 2 : 0);
    o.ratio[0] = 1.0f; o.ratio[1] = 1.0f;
    o.n_caution = ncau;
    pout[blockIdx.x] = o;
    sbox[0] = (maxX >= minX && maxY >= minY) ? x1 : 0;
    sbox[1] = (maxX >= minX && maxY >= minY) ? y1 : 0;
  }
  __syncthreads();
  const float bx = (float)sbox[0], by = (float)sbox[1];
  for (int64_t i = P.v0 + threadIdx.x; i < P.v1; i += 256) {  // :100-102
    texcoord[2 * i] -= bx;
    texcoord[2 * i + 1] -= by;
  }
}

// Atlas::UpdateBuffer.  LDS tile = the keyframe ROI (rows x cols x 3 bytes, capped); the slot
// (PW x PH texels) is written as contiguous row segments of the atlas.
constexpr int kMaxRoiBytes = 48 * 1024;

__device__ __forceinline__ int cv_round_f(float v) { return (int)rintf(v); }

__global__ __launch_bounds__(256) void k_atlas_blit(const PatchIn* __restrict__ pin,
                                                    PatchOut* __restrict__ pout, uint8_t* atlas,
                                                    int aw, int ah, int PW, int PH, int W) {
  extern __shared__ __attribute__((aligned(16))) uint8_t roi[];
  const PatchIn P = pin[blockIdx.x];
  PatchOut* O = &pout[blockIdx.x];
  const int bx = O->bbox[0], by = O->bbox[1], cols = O->bbox[2], rows = O->bbox[3];
  if (cols <= 0 || rows <= 0) return;
  float r0 = 1.0f, r1 = 1.0f;
  if (cols > PW) r0 = (float)PW / (float)cols;  // Atlas.cpp:77-80
  if (rows > PH) r1 = (float)PH / (float)rows;
  if (threadIdx.x == 0) { O->ratio[0] = r0; O->ratio[1] = r1; }
  const uint64_t ox = P.texloc % (uint64_t)aw, oy = P.texloc / (uint64_t)aw;
  const size_t astep = (size_t)aw * 3;
  const bool in_lds = (size_t)cols * rows * 3 <= (size_t)kMaxRoiBytes;
  const int rowbytes = cols * 3;
  if (in_lds) {
    for (int i = threadIdx.x; i < rows * rowbytes; i += 256) {
      const int r = i / rowbytes, c = i - r * rowbytes;
      roi[i] = P.rgb[((size_t)(by + r) * W + bx) * 3 + c];
    }
    __syncthreads();
  }
  auto src = [&](int r, int cbyte) -> int {
    return in_lds ? roi[r * rowbytes + cbyte] : P.rgb[((size_t)(by + r) * W + bx) * 3 + cbyte];
  };
  if (r0 < 1 || r1 < 1) {  // cv::resize(image, texroi, texroi.size()) into the FULL slot
    if (ox + PW > (uint64_t)aw || oy + PH > (uint64_t)ah) return;
    const double scale_x = 1.0 / ((double)PW / cols), scale_y = 1.0 / ((double)PH / rows);
    for (int t = threadIdx.x; t < PW * PH; t += 256) {
      const int dy = t / PW, dx = t - dy * PW;
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = (int)floorf(fx);
      fx -= (float)sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
      const int a0 = (short)cv_round_f((1.f - fx) * 2048.f), a1 = (short)cv_round_f(fx * 2048.f);
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = (int)floorf(fy);
      fy -= (float)sy;
      int sy0 = sy, sy1 = sy + 1;
      sy0 = sy0 < 0 ? 0 : (sy0 > rows - 1 ? rows - 1 : sy0);
      sy1 = sy1 < 0 ? 0 : (sy1 > rows - 1 ? rows - 1 : sy1);
      const int b0 = (short)cv_round_f((1.f - fy) * 2048.f), b1 = (short)cv_round_f(fy * 2048.f);
      const int sx1 = sx + 1 < cols ? sx + 1 : sx;
      uint8_t* D = atlas + (oy + dy) * astep + (ox + dx) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int h0 = src(sy0, 3 * sx + k) * a0 + src(sy0, 3 * sx1 + k) * a1;
        const int h1 = src(sy1, 3 * sx + k) * a0 + src(sy1, 3 * sx1 + k) * a1;
        int val = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        val = val < 0 ? 0 : (val > 255 ? 255 : val);
        D[k] = (uint8_t)val;
      }
    }
  } else {  // image.copyTo(texroi) at the slot origin
    if (ox + cols > (uint64_t)aw || oy + rows > (uint64_t)ah) return;
    for (int i = threadIdx.x; i < rows * rowbytes; i += 256) {
      const int r = i / rowbytes, c = i - r * rowbytes;
      atlas[(oy + r) * astep + ox * 3 + c] = (uint8_t)src(r, c);
    }
  }
}

// ---- host side ------------------------------------------------------------------------
int atlas_init(tf_volume* v) {
  AtlasState& a = v->atlas;
  a.aw = v->cfg.atlas_w;
  a.ah = v->cfg.atlas_h;
  a.pw = (uint64_t)floor((double)(4800.0f * v->res));  // Atlas::SetResolution, Atlas.h:62-65
  a.ph = (uint64_t)floor((double)(3600.0f * v->res));
  a.loc_next = 0;
  const size_t bytes = (size_t)a.aw * a.ah * 3;
  hipError_t e = hipMalloc((void**)&a.buf, bytes);  // Atlas.cpp:34: 13824 x 13824 x RGB8 = 573 MB
  if (e != hipSuccess) { set_error("atlas hipMalloc failed"); return TF_ERR_HIP; }
  TF_HIP(hipMemsetAsync(a.buf, 0, bytes, v->stream));  // Atlas.cpp:35-36
  return TF_OK;
}

void atlas_destroy(tf_volume* v) {
  AtlasState& a = v->atlas;
  for (auto& kv : a.keyframes)
    if (kv.second.owned) { hipFree(kv.second.rgb); hipFree(kv.second.depth); }
  a.keyframes.clear();
  if (a.buf) hipFree(a.buf);
  if (a.d_stage) hipFree(a.d_stage);
  if (a.h_stage) hipHostFree(a.h_stage);
  a.buf = nullptr; a.d_stage = nullptr; a.h_stage = nullptr;
  for (int k = 0; k < 4; ++k) {
    if (a.pin_ev[k]) hipEventDestroy(a.pin_ev[k]);
    if (a.pin_host[k]) hipHostFree(a.pin_host[k]);
    if (a.pin_dev[k]) hipFree(a.pin_dev[k]);
    a.pin_ev[k] = nullptr; a.pin_host[k] = nullptr; a.pin_dev[k] = nullptr; a.pin_bytes[k] = 0;
  }
}

int atlas_reset(tf_volume* v) {
  AtlasState& a = v->atlas;
  a.loc_next = 0;
  a.texloc.clear();
  TF_HIP(hipMemsetAsync(a.buf, 0, (size_t)a.aw * a.ah * 3, v->stream));
  return TF_OK;
}

// ---------------------------------------------------------------------------------------
// Chisel::CompensateColor (Structure/Chisel.cpp:198-286).  The reductions over all vertices of a
// cluster and the per-vertex transfer run here; the two 3x3 eigen-decompositions per cluster are
// host work (a few hundred flops).  Reductions are fixed-shape trees (thread-strided partial sums
// in patch order, then an LDS tree), so results do not depend on timing.
// ---------------------------------------------------------------------------------------
struct CcPatch {
  int64_t v0, v1;
  int32_t cluster;  // -1 = skipped (already adjusted)
  int32_t wrong;
};
// pass 0: sums of texcolor / mesh colour -> out[c][0..5], count -> out[c][6];
// pass 1: centred second moments (6 unique entries each) -> out[c][0..11] given mean[c][0..5]
template <int PASS>
__global__ __launch_bounds__(256) void k_cc_reduce(const CcPatch* __restrict__ pt, int64_t np,
                                                   const float* __restrict__ src, const float* __restrict__ tar,
                                                   const float* __restrict__ mean, float* __restrict__ out) {
  constexpr int NV = PASS == 0 ? 7 : 12;
  const int c = blockIdx.x;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
  float m[6] = {0, 0, 0, 0, 0, 0};
  if (PASS == 1)
    for (int i = 0; i < 6; ++i) m[i] = mean[c * 6 + i];
  for (int64_t p = 0; p < np; ++p) {
    if (pt[p].cluster != c || pt[p].wrong) continue;
    for (int64_t k = pt[p].v0 + threadIdx.x; k < pt[p].v1; k += 256) {
      const float s0 = src[3 * k], s1 = src[3 * k + 1], s2 = src[3 * k + 2];
      const float t0 = tar[3 * k], t1 = tar[3 * k + 1], t2 = tar[3 * k + 2];
      if (PASS == 0) {
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
// labs[k] = T (texcolor[k] - mean_src) + mean_tar (Chisel.cpp:274)
__global__ __launch_bounds__(256) void k_cc_apply(const CcPatch* __restrict__ pt, const float* __restrict__ src,
                                                  const float* __restrict__ xf /* per cluster: T[9], mean_src[3], mean_tar[3], ok */,
                                                  float* __restrict__ labs) {
  const CcPatch P = pt[blockIdx.x];
  if (P.cluster < 0 || P.wrong) return;
  const float* X = xf + (size_t)P.cluster * 16;
  if (X[15] == 0.0f) return;  // nothing was learnt for this cluster
  for (int64_t k = P.v0 + threadIdx.x; k < P.v1; k += 256) {
    const float d0 = src[3 * k] - X[9], d1 = src[3 * k + 1] - X[10], d2 = src[3 * k + 2] - X[11];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float a = X[3 * i] * d0;
      a = a + X[3 * i + 1] * d1;
      a = a + X[3 * i + 2] * d2;
      labs[3 * k + i] = a + X[12 + i];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Chisel::DrawMeshes (Structure/Chisel.cpp:288-355): interleaved vertex stream + rebased indices.
// One workgroup per complete() patch; every output element is written once, 48 B per vertex.
// ---------------------------------------------------------------------------------------
struct PvPatch {
  int64_t v0, v1, i0, i1;  // input vertex / index ranges
  int64_t vout, iout;      // output positions (running counts over the complete patches before this one)
  float ox, oy;            // slot origin (Atlas::GetTexLoc)
  float rx, ry;            // Patch::ratio
  int32_t flags;           // bit0 complete, bit1 wrong_mapping, bit2 labs valid
  int32_t pad;
};
__global__ __launch_bounds__(256) void k_pack_vertices(const PvPatch* __restrict__ pt, const float* __restrict__ verts,
                                                       const float* __restrict__ colors, const float* __restrict__ normals,
                                                       const float* __restrict__ texcoord, const float* __restrict__ texcolor,
                                                       const float* __restrict__ labs, const uint32_t* __restrict__ indices,
                                                       float inv_unused, float aw, float ah, float* __restrict__ out_v,
                                                       uint32_t* __restrict__ out_i) {
  const PvPatch P = pt[blockIdx.x];
  if (!(P.flags & 1)) return;
  for (int64_t j = P.i0 + threadIdx.x; j < P.i1; j += 256) out_i[P.iout + (j - P.i0)] = indices[j] + (uint32_t)P.vout;
  for (int64_t k = P.v0 + threadIdx.x; k < P.v1; k += 256) {
    float* o = out_v + 12 * (P.vout + (k - P.v0));
    float tx = texcoord[2 * k], ty = texcoord[2 * k + 1];
    if (P.rx < 1.0f) tx = tx * P.rx;
    if (P.ry < 1.0f) ty = ty * P.ry;
    tx = tx + P.ox;
    ty = ty + P.oy;
    int rgb = (int)(colors[3 * k] * 255.0f);
    rgb = (rgb << 8) + (int)(colors[3 * k + 1] * 255.0f);
    rgb = (rgb << 8) + (int)(colors[3 * k + 2] * 255.0f);
    float adj = 0.0f;
    if (P.flags & 4) {
      const float a0 = labs[3 * k] - texcolor[3 * k], a1 = labs[3 * k + 1] - texcolor[3 * k + 1],
                  a2 = labs[3 * k + 2] - texcolor[3 * k + 2];
      int ad = (int)(a0 * 255.0f) + 255;
      ad = (ad << 9) + (int)(a1 * 255.0f) + 255;
      ad = (ad << 9) + (int)(a2 * 255.0f) + 255;
      adj = (float)ad;
    }
    const float4 q0 = make_float4(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2], 50.0f);
    const float4 q1 = make_float4((float)rgb, adj, tx / aw, ty / ah);
    const float4 q2 = make_float4(normals[3 * k], normals[3 * k + 1], normals[3 * k + 2], (P.flags & 2) ? 1.0f : 0.0f);
    reinterpret_cast<float4*>(o)[0] = q0;
    reinterpret_cast<float4*>(o)[1] = q1;
    reinterpret_cast<float4*>(o)[2] = q2;
  }
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

// Atlas::AddPatch (Atlas.cpp:43-64)
static int add_patch(AtlasState& a, const int32_t id[3], uint64_t* texloc) {
  const uint64_t key = host_pack_id(id);
  bool inserted = false;
  uint64_t* slot = a.texloc.find_or_insert(key, &inserted);
  if (!inserted && *slot != FlatMap64::kNone) { *texloc = *slot; return TF_OK; }  // Patch::clear keeps texloc
  *texloc = a.loc_next;
  uint64_t x = a.loc_next % (uint64_t)a.aw, y = a.loc_next / (uint64_t)a.aw;
  if (x >= (uint64_t)a.aw || y >= (uint64_t)a.ah) {
    *slot = FlatMap64::kNone;  // known id without a slot: a later call tries again (and overflows again)
    set_error("No enough space for texture storage.");  // std::overflow_error text, Atlas.cpp:53
    return TF_ERR_ATLAS_FULL;
  }
  if (x + a.pw >= (uint64_t)a.aw) { x = 0; y += a.ph; }
  else x += a.pw;
  a.loc_next = x + y * (uint64_t)a.aw;
  *slot = *texloc;
  return TF_OK;
}

}  // namespace tf

using namespace tf;

extern "C" {

int tf_keyframe_cache(tf_volume* v, int32_t kf_id, const uint8_t* rgb, const float* depth) {
  if (!v || !rgb || !depth) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  const size_t npix = (size_t)v->cam.W * v->cam.H;
  KeyframeSlot& ks = a.keyframes[kf_id];
  if (!ks.owned) {
    if ((int)a.keyframes.size() > v->cfg.max_keyframes) {
      a.keyframes.erase(kf_id);
      set_error("keyframe cache full (tf_config.max_keyframes)");
      return TF_ERR_CAPACITY;
    }
    ks.rgb = nullptr; ks.depth = nullptr;
    TF_HIP(hipMalloc((void**)&ks.rgb, npix * 3));
    TF_HIP(hipMalloc((void**)&ks.depth, npix * 4));
    ks.owned = true;
  }
  int rc = atlas_stage(v, npix * 7);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  memcpy(hs, rgb, npix * 3);
  memcpy(hs + npix * 3, depth, npix * 4);
  TF_HIP(hipMemcpyAsync(ks.rgb, hs, npix * 3, hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipMemcpyAsync(ks.depth, hs + npix * 3, npix * 4, hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  return TF_OK;
}

int tf_keyframe_cache_device(tf_volume* v, int32_t kf_id, const uint8_t* d_rgb, const float* d_depth) {
  if (!v || !d_rgb || !d_depth) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  auto it = a.keyframes.find(kf_id);
  if (it != a.keyframes.end() && it->second.owned) { hipFree(it->second.rgb); hipFree(it->second.depth); }
  KeyframeSlot ks;
  ks.rgb = const_cast<uint8_t*>(d_rgb);
  ks.depth = const_cast<float*>(d_depth);
  ks.owned = false;
  a.keyframes[kf_id] = ks;
  return TF_OK;
}

int tf_keyframe_release(tf_volume* v, int32_t kf_id) {
  if (!v) { set_error("null handle"); return TF_ERR_INVALID; }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  auto it = a.keyframes.find(kf_id);
  if (it == a.keyframes.end()) return TF_OK;
  TF_HIP(hipStreamSynchronize(v->stream));
  if (it->second.owned) { hipFree(it->second.rgb); hipFree(it->second.depth); }
  a.keyframes.erase(it);
  return TF_OK;
}

int tf_atlas_patch_size(tf_volume* v, int32_t* pw, int32_t* ph) {
  if (!v || !pw || !ph) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  *pw = (int32_t)v->atlas.pw;
  *ph = (int32_t)v->atlas.ph;
  return TF_OK;
}

int tf_atlas_add_patch(tf_volume* v, const int32_t id[3], uint64_t* texloc) {
  if (!v || !id || !texloc) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  return add_patch(v->atlas, id, texloc);
}

int tf_atlas_loc_next(tf_volume* v, uint64_t* loc_next) {
  if (!v || !loc_next) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  *loc_next = v->atlas.loc_next;
  return TF_OK;
}

int tf_patches_update(tf_volume* v, int64_t np, const int32_t* ids, const int32_t* kf_ids,
                      const float* pose_inv16, const int64_t* voff, const float* verts,
                      const float* colors, float* out_texcoord, float* out_texcolor,
                      int32_t* out_bbox, int32_t* out_flags, float* out_ratio, uint64_t* out_texloc,
                      uint64_t out_hot[2]) {
  if (!v || (np > 0 && (!ids || !kf_ids || !pose_inv16 || !voff || !verts || !colors))) {
    set_error("null argument");
    return TF_ERR_INVALID;
  }
  TF_DEV(v);
  AtlasState& a = v->atlas;
  if (np <= 0) {
    if (out_hot) {  // Chisel.cpp:153-154,184-186 with an empty loop
      const uint64_t ls = (uint64_t)a.aw * (uint64_t)a.ah;
      out_hot[0] = (ls / a.aw) * a.aw;
      out_hot[1] = (0 / a.aw + a.ph) * a.aw;
    }
    return TF_OK;
  }
  const int64_t nv = voff[np];
  const size_t o_pin = 0;
  const size_t o_verts = o_pin + sizeof(PatchIn) * (size_t)np;
  const size_t o_cols = o_verts + (size_t)nv * 12;
  const size_t o_tc = o_cols + (size_t)nv * 12;
  const size_t o_tcol = o_tc + (size_t)nv * 8;
  const size_t o_pout = (o_tcol + (size_t)nv * 12 + 15) & ~(size_t)15;
  const size_t total = o_pout + sizeof(PatchOut) * (size_t)np;
  int rc = atlas_stage(v, total);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  uint8_t* ds = reinterpret_cast<uint8_t*>(a.d_stage);
  PatchIn* hp = reinterpret_cast<PatchIn*>(hs + o_pin);
  uint64_t loc_start = (uint64_t)a.aw * (uint64_t)a.ah, loc_end = 0;  // Chisel.cpp:153-154
  int64_t n_ok = np;
  int overflow = 0;
  auto it = a.keyframes.end();
  for (int64_t p = 0; p < np; ++p) {
    uint64_t tl = 0;
    rc = add_patch(a, ids + 3 * p, &tl);  // Chisel.cpp:167-173: overflow aborts GeneratePatches
    if (rc) { n_ok = p; overflow = 1; break; }
    if (p == 0 || kf_ids[p] != kf_ids[p - 1]) it = a.keyframes.find(kf_ids[p]);
    if (it == a.keyframes.end()) {
      set_error("keyframe " + std::to_string(kf_ids[p]) + " is not cached (tf_keyframe_cache)");
      return TF_ERR_INVALID;
    }
    memcpy(hp[p].T, pose_inv16 + 16 * p, 64);
    hp[p].rgb = it->second.rgb;
    hp[p].depth = it->second.depth;
    hp[p].v0 = voff[p];
    hp[p].v1 = voff[p + 1];
    hp[p].texloc = tl;
    if (out_texloc) out_texloc[p] = tl;
    if (tl < loc_start) loc_start = tl;
    if (tl > loc_end) loc_end = tl;
  }
  if (overflow) return TF_ERR_ATLAS_FULL;  // tsdfFusion stops (MobileFusion.cpp:376-379)
  memcpy(hs + o_verts, verts, (size_t)nv * 12);
  memcpy(hs + o_cols, colors, (size_t)nv * 12);
  TF_HIP(hipMemcpyAsync(ds, hs, o_tc, hipMemcpyHostToDevice, v->stream));
  prof_begin(v, TF_PROF_PATCH_PROJECT);
  hipLaunchKernelGGL(k_patch_project, dim3((unsigned)n_ok), dim3(256), 0, v->stream,
                     reinterpret_cast<const PatchIn*>(ds + o_pin),
                     reinterpret_cast<const float*>(ds + o_verts),
                     reinterpret_cast<const float*>(ds + o_cols), v->cam,
                     reinterpret_cast<float*>(ds + o_tc), reinterpret_cast<float*>(ds + o_tcol),
                     reinterpret_cast<PatchOut*>(ds + o_pout));
  prof_end(v);
  prof_begin(v, TF_PROF_ATLAS_BLIT);
  hipLaunchKernelGGL(k_atlas_blit, dim3((unsigned)n_ok), dim3(256), kMaxRoiBytes, v->stream,
                     reinterpret_cast<const PatchIn*>(ds + o_pin),
                     reinterpret_cast<PatchOut*>(ds + o_pout), a.buf, a.aw, a.ah, (int)a.pw,
                     (int)a.ph, v->cam.W);
  prof_end(v);
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(hs + o_tc, ds + o_tc, total - o_tc, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  if (out_texcoord) memcpy(out_texcoord, hs + o_tc, (size_t)nv * 8);
  if (out_texcolor) memcpy(out_texcolor, hs + o_tcol, (size_t)nv * 12);
  const PatchOut* po = reinterpret_cast<const PatchOut*>(hs + o_pout);
  for (int64_t p = 0; p < np; ++p) {
    if (out_bbox) memcpy(out_bbox + 4 * p, po[p].bbox, 16);
    if (out_flags) out_flags[p] = po[p].flags;
    if (out_ratio) { out_ratio[2 * p] = po[p].ratio[0]; out_ratio[2 * p + 1] = po[p].ratio[1]; }
  }
  if (out_hot) {  // Chisel.cpp:184-186
    out_hot[0] = (loc_start / (uint64_t)a.aw) * (uint64_t)a.aw;
    out_hot[1] = (loc_end / (uint64_t)a.aw + a.ph) * (uint64_t)a.aw;
  }
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

// tf_patches_update with the mesh data already resident and the results left in HBM: nothing but the
// per-patch descriptors (112 B each) crosses PCIe and nothing synchronises, so a keyframe's atlas
// update rides in the frame stream.  Slot allocation stays on the host (immediate, in patch order).
int tf_patches_update_device(tf_volume* v, int64_t np, const int32_t* ids, const int32_t* kf_ids,
                             const float* pose_inv16, const int64_t* voff, const float* d_verts,
                             const float* d_colors, float* d_texcoord, float* d_texcolor,
                             tf_patch_out* d_patch_out, uint64_t* out_texloc, uint64_t out_hot[2]) {
  if (!v || (np > 0 && (!ids || !kf_ids || !pose_inv16 || !voff || !d_verts || !d_colors || !d_texcoord ||
                        !d_texcolor || !d_patch_out))) {
    set_error("null argument");
    return TF_ERR_INVALID;
  }
  TF_DEV(v);
  static_assert(sizeof(tf_patch_out) == sizeof(PatchOut), "public and device patch records differ");
  AtlasState& a = v->atlas;
  if (np <= 0) {
    if (out_hot) {
      const uint64_t ls = (uint64_t)a.aw * (uint64_t)a.ah;
      out_hot[0] = (ls / a.aw) * a.aw;
      out_hot[1] = (0 / a.aw + a.ph) * a.aw;
    }
    return TF_OK;
  }
  // descriptor ring: pinned host copies must outlive their asynchronous upload
  constexpr int kRing = 4;
  const size_t bytes = sizeof(PatchIn) * (size_t)np;
  const int slot = a.pin_next;
  a.pin_next = (a.pin_next + 1) % kRing;
  if (a.pin_ev[slot]) TF_HIP(hipEventSynchronize(a.pin_ev[slot]));
  else TF_HIP(hipEventCreateWithFlags(&a.pin_ev[slot], hipEventDisableTiming));
  if (bytes > a.pin_bytes[slot]) {
    if (a.pin_host[slot]) hipHostFree(a.pin_host[slot]);
    if (a.pin_dev[slot]) { TF_HIP(hipStreamSynchronize(v->stream)); hipFree(a.pin_dev[slot]); }
    size_t want = 4096;
    while (want < bytes) want <<= 1;
    TF_HIP(hipHostMalloc(&a.pin_host[slot], want, hipHostMallocDefault));
    TF_HIP(hipMalloc(&a.pin_dev[slot], want));
    a.pin_bytes[slot] = want;
  }
  PatchIn* hp = reinterpret_cast<PatchIn*>(a.pin_host[slot]);
  uint64_t loc_start = (uint64_t)a.aw * (uint64_t)a.ah, loc_end = 0;  // Chisel.cpp:153-154
  auto it = a.keyframes.end();
  for (int64_t p = 0; p < np; ++p) {
    uint64_t tl = 0;
    int rc = add_patch(a, ids + 3 * p, &tl);  // Chisel.cpp:167-173: overflow aborts GeneratePatches
    if (rc) return TF_ERR_ATLAS_FULL;
    if (p == 0 || kf_ids[p] != kf_ids[p - 1]) it = a.keyframes.find(kf_ids[p]);
    if (it == a.keyframes.end()) {
      set_error("keyframe " + std::to_string(kf_ids[p]) + " is not cached (tf_keyframe_cache)");
      return TF_ERR_INVALID;
    }
    memcpy(hp[p].T, pose_inv16 + 16 * p, 64);
    hp[p].rgb = it->second.rgb;
    hp[p].depth = it->second.depth;
    hp[p].v0 = voff[p];
    hp[p].v1 = voff[p + 1];
    hp[p].texloc = tl;
    if (out_texloc) out_texloc[p] = tl;
    if (tl < loc_start) loc_start = tl;
    if (tl > loc_end) loc_end = tl;
  }
  const PatchIn* dp = reinterpret_cast<const PatchIn*>(a.pin_dev[slot]);
  TF_HIP(hipMemcpyAsync(a.pin_dev[slot], hp, bytes, hipMemcpyHostToDevice, v->stream));
  TF_HIP(hipEventRecord(a.pin_ev[slot], v->stream));
  prof_begin(v, TF_PROF_PATCH_PROJECT);
  hipLaunchKernelGGL(k_patch_project, dim3((unsigned)np), dim3(256), 0, v->stream, dp, d_verts, d_colors, v->cam,
                     d_texcoord, d_texcolor, reinterpret_cast<PatchOut*>(d_patch_out));
  prof_end(v);
  prof_begin(v, TF_PROF_ATLAS_BLIT);
  hipLaunchKernelGGL(k_atlas_blit, dim3((unsigned)np), dim3(256), kMaxRoiBytes, v->stream, dp,
                     reinterpret_cast<PatchOut*>(d_patch_out), a.buf, a.aw, a.ah, (int)a.pw, (int)a.ph, v->cam.W);
  prof_end(v);
  TF_HIP(hipGetLastError());
  if (out_hot) {  // Chisel.cpp:184-186
    out_hot[0] = (loc_start / (uint64_t)a.aw) * (uint64_t)a.aw;
    out_hot[1] = (loc_end / (uint64_t)a.aw + a.ph) * (uint64_t)a.aw;
  }
  return TF_OK;
}

int tf_color_compensate(tf_volume* v, int64_t np, const int32_t* frame_ids, const uint8_t* wrong_mapping,
                        uint8_t* has_adjusted, const int64_t* voff, const float* texcolor,
                        const float* meshcolor, float* out_labs, int64_t* out_n_clusters) {
  if (out_n_clusters) *out_n_clusters = 0;
  if (!v || (np > 0 && (!frame_ids || !wrong_mapping || !has_adjusted || !voff || !texcolor || !meshcolor || !out_labs))) {
    set_error("null argument");
    return TF_ERR_INVALID;
  }
  TF_DEV(v);
  if (np <= 0) return TF_OK;
  // clusters by source frame in order of first appearance (Chisel.cpp:199-214)
  std::vector<int32_t> cl((size_t)np, -1), first;
  for (int64_t p = 0; p < np; ++p) {
    if (has_adjusted[p]) continue;
    size_t k = 0;
    for (; k < first.size(); ++k)
      if (frame_ids[first[k]] == frame_ids[p]) break;
    if (k == first.size()) first.push_back((int32_t)p);
    cl[(size_t)p] = (int32_t)k;
  }
  const size_t ncl = first.size();
  if (out_n_clusters) *out_n_clusters = (int64_t)ncl;
  if (!ncl) return TF_OK;
  const int64_t nv = voff[np];
  AtlasState& a = v->atlas;
  const size_t o_pt = 0;
  const size_t o_src = (o_pt + sizeof(CcPatch) * (size_t)np + 15) & ~(size_t)15;
  const size_t o_tar = o_src + (size_t)nv * 12;
  const size_t o_labs = o_tar + (size_t)nv * 12;
  const size_t o_red = (o_labs + (size_t)nv * 12 + 15) & ~(size_t)15;  // [ncl][12] sums / moments
  const size_t o_mean = o_red + ncl * 48;                                // [ncl][6]
  const size_t o_xf = o_mean + ncl * 24;                                 // [ncl][16]
  const size_t total = o_xf + ncl * 64;
  int rc = atlas_stage(v, total);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  uint8_t* ds = reinterpret_cast<uint8_t*>(a.d_stage);
  CcPatch* hp = reinterpret_cast<CcPatch*>(hs + o_pt);
  for (int64_t p = 0; p < np; ++p) {
    hp[p].v0 = voff[p]; hp[p].v1 = voff[p + 1];
    hp[p].cluster = cl[(size_t)p]; hp[p].wrong = wrong_mapping[p] ? 1 : 0;
  }
  memcpy(hs + o_src, texcolor, (size_t)nv * 12);
  memcpy(hs + o_tar, meshcolor, (size_t)nv * 12);
  memcpy(hs + o_labs, out_labs, (size_t)nv * 12);  // entries of untouched patches keep the caller's values
  TF_HIP(hipMemcpyAsync(ds, hs, o_red, hipMemcpyHostToDevice, v->stream));
  const CcPatch* dp = reinterpret_cast<const CcPatch*>(ds + o_pt);
  const float* dsrc = reinterpret_cast<const float*>(ds + o_src);
  const float* dtar = reinterpret_cast<const float*>(ds + o_tar);
  float* dred = reinterpret_cast<float*>(ds + o_red);
  float* hred = reinterpret_cast<float*>(hs + o_red);
  float* hmean = reinterpret_cast<float*>(hs + o_mean);
  float* hxf = reinterpret_cast<float*>(hs + o_xf);
  // computeMeanAndCov (Patch.cpp:342-348): mean, then centred second moments / (N - 1)
  hipLaunchKernelGGL(k_cc_reduce<0>, dim3((unsigned)ncl), dim3(256), 0, v->stream, dp, np, dsrc, dtar,
                     (const float*)nullptr, dred);
  TF_HIP(hipMemcpyAsync(hred, dred, ncl * 48, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  std::vector<float> cnt(ncl);
  for (size_t c = 0; c < ncl; ++c) {
    cnt[c] = hred[c * 12 + 6];
    for (int i = 0; i < 6; ++i) hmean[c * 6 + i] = cnt[c] > 0.0f ? hred[c * 12 + i] / cnt[c] : 0.0f;
  }
  TF_HIP(hipMemcpyAsync(ds + o_mean, hmean, ncl * 24, hipMemcpyHostToDevice, v->stream));
  hipLaunchKernelGGL(k_cc_reduce<1>, dim3((unsigned)ncl), dim3(256), 0, v->stream, dp, np, dsrc, dtar,
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
  hipLaunchKernelGGL(k_cc_apply, dim3((unsigned)np), dim3(256), 0, v->stream, dp, dsrc,
                     reinterpret_cast<const float*>(ds + o_xf), reinterpret_cast<float*>(ds + o_labs));
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(hs + o_labs, ds + o_labs, (size_t)nv * 12, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  memcpy(out_labs, hs + o_labs, (size_t)nv * 12);
  for (int64_t p = 0; p < np; ++p)
    if (cl[(size_t)p] >= 0 && cnt[(size_t)cl[(size_t)p]] > 0.0f) has_adjusted[p] = 1;  // :280
  return TF_OK;
}

int tf_pack_vertices(tf_volume* v, int64_t np, const uint8_t* complete, const uint8_t* wrong_mapping,
                     const uint8_t* labs_valid, const uint64_t* texloc, const float* ratio,
                     const int64_t* voff, const float* verts, const float* colors, const float* normals,
                     const float* texcoord, const float* texcolor, const float* labs, const int64_t* ioff,
                     const uint32_t* indices, float* out_vertices, uint32_t* out_indices,
                     int64_t* out_n_vertices, int64_t* out_n_indices) {
  if (out_n_vertices) *out_n_vertices = 0;
  if (out_n_indices) *out_n_indices = 0;
  if (!v || (np > 0 && (!complete || !wrong_mapping || !labs_valid || !texloc || !ratio || !voff || !verts ||
                        !colors || !normals || !texcoord || !texcolor || !labs || !ioff || !out_vertices))) {
    set_error("null argument");
    return TF_ERR_INVALID;
  }
  TF_DEV(v);
  if (np <= 0) return TF_OK;
  AtlasState& a = v->atlas;
  const int64_t nv = voff[np], ni = ioff[np];
  if (ni > 0 && (!indices || !out_indices)) { set_error("null index argument"); return TF_ERR_INVALID; }
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
  const size_t o_pt = take(sizeof(PvPatch) * (size_t)np);
  const size_t o_verts = take((size_t)nv * 12), o_cols = take((size_t)nv * 12), o_nrm = take((size_t)nv * 12);
  const size_t o_tc = take((size_t)nv * 8), o_tcol = take((size_t)nv * 12), o_labs = take((size_t)nv * 12);
  const size_t o_idx = take((size_t)ni * 4);
  const size_t o_in_end = o;
  const size_t o_outv = take((size_t)nv * 48), o_outi = take((size_t)ni * 4);
  const size_t total = o;
  int rc = atlas_stage(v, total);
  if (rc) return rc;
  TF_HIP(hipStreamSynchronize(v->stream));
  uint8_t* hs = reinterpret_cast<uint8_t*>(a.h_stage);
  uint8_t* ds = reinterpret_cast<uint8_t*>(a.d_stage);
  PvPatch* hp = reinterpret_cast<PvPatch*>(hs + o_pt);
  int64_t vout = 0, iout = 0;
  for (int64_t p = 0; p < np; ++p) {
    PvPatch& P = hp[p];
    P.v0 = voff[p]; P.v1 = voff[p + 1]; P.i0 = ioff[p]; P.i1 = ioff[p + 1];
    P.vout = vout; P.iout = iout;
    P.ox = (float)(texloc[p] % (uint64_t)a.aw);  // Atlas::GetTexLoc (Atlas.cpp:66-69)
    P.oy = (float)(texloc[p] / (uint64_t)a.aw);
    P.rx = ratio[2 * p]; P.ry = ratio[2 * p + 1];
    P.flags = (complete[p] ? 1 : 0) | (wrong_mapping[p] ? 2 : 0) | (labs_valid[p] ? 4 : 0);
    P.pad = 0;
    if (complete[p]) { vout += P.v1 - P.v0; iout += P.i1 - P.i0; }
  }
  memcpy(hs + o_verts, verts, (size_t)nv * 12);
  memcpy(hs + o_cols, colors, (size_t)nv * 12);
  memcpy(hs + o_nrm, normals, (size_t)nv * 12);
  memcpy(hs + o_tc, texcoord, (size_t)nv * 8);
  memcpy(hs + o_tcol, texcolor, (size_t)nv * 12);
  memcpy(hs + o_labs, labs, (size_t)nv * 12);
  if (ni) memcpy(hs + o_idx, indices, (size_t)ni * 4);
  TF_HIP(hipMemcpyAsync(ds, hs, o_in_end, hipMemcpyHostToDevice, v->stream));
  hipLaunchKernelGGL(k_pack_vertices, dim3((unsigned)np), dim3(256), 0, v->stream,
                     reinterpret_cast<const PvPatch*>(ds + o_pt), reinterpret_cast<const float*>(ds + o_verts),
                     reinterpret_cast<const float*>(ds + o_cols), reinterpret_cast<const float*>(ds + o_nrm),
                     reinterpret_cast<const float*>(ds + o_tc), reinterpret_cast<const float*>(ds + o_tcol),
                     reinterpret_cast<const float*>(ds + o_labs), reinterpret_cast<const uint32_t*>(ds + o_idx),
                     0.0f, (float)a.aw, (float)a.ah, reinterpret_cast<float*>(ds + o_outv),
                     reinterpret_cast<uint32_t*>(ds + o_outi));
  TF_HIP(hipGetLastError());
  TF_HIP(hipMemcpyAsync(hs + o_outv, ds + o_outv, total - o_outv, hipMemcpyDeviceToHost, v->stream));
  TF_HIP(hipStreamSynchronize(v->stream));
  memcpy(out_vertices, hs + o_outv, (size_t)vout * 48);
  if (iout) memcpy(out_indices, hs + o_outi, (size_t)iout * 4);
  if (out_n_vertices) *out_n_vertices = vout;
  if (out_n_indices) *out_n_indices = iout;
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
