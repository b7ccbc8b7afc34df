// tf_pre.hip -- frame pre-processing that feeds the path, on images resident in HBM.
//
//   k_pre_normal_map            BasicAPI::extractNormalMapSIMD      (BasicAPI.cpp:849-905)
//   k_pre_refine_depth_normal   BasicAPI::refineDepthUseNormalSIMD  (BasicAPI.cpp:728-781)
//   k_pre_color_valid           BasicAPI::checkColorQuality         (BasicAPI.cpp:783-806)
//   k_pre_color_quality         BasicAPI::estimateColorQuality      (BasicAPI.cpp:815-847)
//   k_pre_refine_newframe       BasicAPI::refineNewframesSIMD       (BasicAPI.cpp:378-443)
//   k_pre_refine_keyframe       BasicAPI::refineKeyframesSIMD       (BasicAPI.cpp:506-636)
//
// All of them are one thread per pixel, HBM-bound streaming kernels (a few image reads, one or two writes).
// Arithmetic follows the vec8 operator order of the reference without FMA.  _mm256_rsqrt_ps (an approximation
// that differs between x86 vendors) is the correctly rounded 1 / sqrt here and in the oracle; pixels the reference
// leaves uninitialised (cv::Mat::create) are zero.  refineKeyframesSIMD updates the keyframe's depth in place,
// 8 pixels at a time in row-major order, and its nearest-neighbour fallback reads that same map: a pixel may see
// values earlier groups have already rewritten.  The device reproduces the sequential result as the fixed point
// of "evaluate every pixel against the current estimate of the earlier groups' results" (dependencies only
// point to earlier groups, so the iteration is exact after chain-depth + 1 rounds; 2-3 in practice).
#include <stdlib.h>
#include <string.h>

#include "tf_devfn.h"
#include "tf_volume.h"

#pragma clang fp contract(off)

namespace tf {

struct PreCam {
  int W, H;
  float fx, fy, cx, cy;
};
struct PreT {
  float t[12];
};

__device__ __forceinline__ float pre_rsqrt(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ int pre_cvt_rne(float x) {  // _mm256_cvtps_epi32 for in-range operands
  return (int)rintf(__builtin_amdgcn_fmed3f(x, -2147483648.0f, 2147483520.0f));
}

__global__ __launch_bounds__(256) void k_pre_normal_map(const float* __restrict__ depth, PreCam c,
                                                        float* __restrict__ normal) {
  const size_t np = (size_t)c.W * c.H;
  // the reference's 8-wide groups start at j = 1, 9, ... while j < W - 10: columns 1 .. jlast + 7
  const int jlast = (c.W - 10 > 1) ? 1 + ((c.W - 12) / 8) * 8 : -8;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (size_t)gridDim.x * 256) {
    const int i = (int)(p / c.W), j = (int)(p - (size_t)i * c.W);
    float nX = 0.0f, nY = 0.0f, nZ = 0.0f;
    if (i >= 1 && i < c.H - 1 && j >= 1 && j <= jlast + 7) {
      const float dr = depth[p + 1], db = depth[p + c.W], dl = depth[p - 1], dt = depth[p - c.W];
      const int j0 = 1 + ((j - 1) & ~7);
      const float xs = ((float)(j - j0) + (float)j0) - c.cx;
      const float ys = (float)i - c.cy;
      const float u3 = dr - dl, v3 = db - dt;
      const float u1 = ((xs * u3 + dr) + dl) / c.fx;
      const float u2 = (ys * u3) / c.fy;
      const float v1 = (xs * v3) / c.fx;
      const float v2 = ((ys * v3 + db) + dt) / c.fy;
      float x = u2 * v3 - u3 * v2;
      float y = u3 * v1 - u1 * v3;
      float z = u1 * v2 - u2 * v1;
      const float nsq = (x * x + y * y) + z * z;
      const bool valid = (u3 < 0.3f) && (u3 > -0.3f) && (v3 < 0.3f) && (v3 > -0.3f) && (nsq > 1e-24f);
      const float r = pre_rsqrt(nsq);
      x = x * r; y = y * r; z = z * r;
      if (valid) { nX = x; nY = y; nZ = z; }
    }
    normal[p] = nX; normal[p + np] = nY; normal[p + 2 * np] = nZ;
  }
}

__global__ __launch_bounds__(256) void k_pre_refine_depth_normal(float* __restrict__ normal, float* __restrict__ depth,
                                                                 PreCam c) {
  const size_t np = (size_t)c.W * c.H;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (size_t)gridDim.x * 256) {
    const int i = (int)(p / c.W), j = (int)(p - (size_t)i * c.W);
    float vX = ((float)j - c.cx) / c.fx, vY = ((float)i - c.cy) / c.fy, vZ = 1.0f;
    const float r = pre_rsqrt((vX * vX + vY * vY) + vZ * vZ);
    vX = vX * r; vY = vY * r; vZ = vZ * r;
    const float q = (vX * normal[p] + vY * normal[p + np]) + vZ * normal[p + 2 * np];
    if (q > -0.1f && q < 0.1f) { depth[p] = 0.0f; normal[p] = 0.0f; normal[p + np] = 0.0f; normal[p + 2 * np] = 0.0f; }
  }
}

__device__ __forceinline__ void pre_view_angle(int i, int j, const PreCam& c, float v[3]) {
  const float x = ((float)j - c.cx) / c.fx, y = ((float)i - c.cy) / c.fy, z = 1.0f;
  const float yz = y * y + z * z;
  const float sq = x * x + yz;
  v[0] = x; v[1] = y; v[2] = z;
  if (sq > 0.0f) { const float n = sqrtf(sq); v[0] = x / n; v[1] = y / n; v[2] = z / n; }
}
__device__ __forceinline__ float pre_dot_tree(const float a[3], float b0, float b1, float b2) {
  const float p0 = a[0] * b0, p1 = a[1] * b1, p2 = a[2] * b2;
  const float s = p1 + p2;
  return p0 + s;
}

__global__ __launch_bounds__(256) void k_pre_color_valid(const float* __restrict__ normal, PreCam c,
                                                         uint8_t* __restrict__ flag) {
  const size_t np = (size_t)c.W * c.H;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (size_t)gridDim.x * 256) {
    const int i = (int)(p / c.W), j = (int)(p - (size_t)i * c.W);
    float v[3];
    pre_view_angle(i, j, c, v);
    const float q = pre_dot_tree(v, normal[p], normal[p + np], normal[p + 2 * np]);
    flag[p] = ((double)fabsf(q) >= 0.2) ? 1 : 0;
  }
}

__device__ __forceinline__ int pre_gray(const uint8_t* __restrict__ rgb, size_t p) {
  return (4899 * (int)rgb[3 * p] + 9617 * (int)rgb[3 * p + 1] + 1868 * (int)rgb[3 * p + 2] + 8192) >> 14;
}
__global__ __launch_bounds__(256) void k_pre_color_quality(const float* __restrict__ depth, const float* __restrict__ normal,
                                                           const uint8_t* __restrict__ rgb, PreCam c,
                                                           float* __restrict__ quality) {
  const size_t np = (size_t)c.W * c.H;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (size_t)gridDim.x * 256) {
    const int i = (int)(p / c.W), j = (int)(p - (size_t)i * c.W);
    const int im = i == 0 ? 1 : i - 1, ip = i == c.H - 1 ? c.H - 2 : i + 1;  // BORDER_REFLECT_101
    const int jm = j == 0 ? 1 : j - 1, jp = j == c.W - 1 ? c.W - 2 : j + 1;
    const int s = pre_gray(rgb, (size_t)ip * c.W + jp) - pre_gray(rgb, (size_t)ip * c.W + jm) -
                  pre_gray(rgb, (size_t)im * c.W + jp) + pre_gray(rgb, (size_t)im * c.W + jm);
    float qv = (float)s;
    if (depth[p] > 0) {
      float v[3];
      pre_view_angle(i, j, c, v);
      const float vq = fabsf(pre_dot_tree(v, normal[p], normal[p + np], normal[p + 2 * np]));
      qv = fabsf(qv) * vq;
    }
    quality[p] = qv;
  }
}

__device__ __forceinline__ void pre_project(const PreT& T, const PreCam& c, int i, int j, float d, float V[3]) {
  const float lx = (((float)j - c.cx) / c.fx) * d, ly = (((float)i - c.cy) / c.fy) * d;
#pragma unroll
  for (int r = 0; r < 3; ++r) V[r] = ((T.t[4 * r] * lx + T.t[4 * r + 1] * ly) + T.t[4 * r + 2] * d) + T.t[4 * r + 3];
}

__global__ __launch_bounds__(256) void k_pre_refine_newframe(const float* __restrict__ depth_ref, float* __restrict__ depth_new,
                                                             PreCam c, PreT T) {
  const size_t np = (size_t)c.W * c.H;
  const float cxh = (float)((double)c.cx + 0.5), cyh = (float)((double)c.cy + 0.5);
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (size_t)gridDim.x * 256) {
    const int i = (int)(p / c.W), j = (int)(p - (size_t)i * c.W);
    const float d = depth_new[p];
    float V[3];
    pre_project(T, c, i, j, d, V);
    const float rx = (V[0] / V[2]) * c.fx + cxh, ry = (V[1] / V[2]) * c.fy + cyh;
    const bool valid = (rx > 1.0f) && (rx < (float)(c.W - 1)) && (ry > 1.0f) && (ry < (float)(c.H - 1));
    float nd = 0.0f;
    if (valid) nd = depth_ref[pre_cvt_rne(floorf(rx) + floorf(ry) * (float)c.W)];
    const float diff = nd - V[2];
    const bool keep = (diff > (-0.05f) * V[2]) && (diff < 0.05f * V[2]);
    depth_new[p] = keep ? d : 0.0f;
  }
}

// one round of the keyframe refinement: est_in = current estimate of the refined keyframe depth (round 0: the
// original), orig = the keyframe's depth / weight before the call
__global__ __launch_bounds__(256) void k_pre_refine_keyframe(const float* __restrict__ orig_d, const float* __restrict__ orig_w,
                                                             const float* __restrict__ depth_new,
                                                             const float* __restrict__ est_in, float* __restrict__ est_out,
                                                             float* __restrict__ w_out, PreCam c, PreT T,
                                                             uint32_t* __restrict__ changed) {
  const size_t np = (size_t)c.W * c.H;
  bool any = false;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (size_t)gridDim.x * 256) {
    const int i = (int)(p / c.W), j = (int)(p - (size_t)i * c.W);
    const float d = orig_d[p];
    float V[3];
    pre_project(T, c, i, j, d, V);
    const float rx = (V[0] / V[2]) * c.fx + c.cx, ry = (V[1] / V[2]) * c.fy + c.cy;
    const bool valid = (rx > 2.0f) && (rx < (float)(c.W - 2)) && (ry > 2.0f) && (ry < (float)(c.H - 2));
    float ul = 0, ur = 0, bl = 0, br = 0, nn = 0;
    const float fxr = floorf(rx), fyr = floorf(ry);
    if (valid) {
      const int q = pre_cvt_rne(fxr + fyr * (float)c.W);
      ul = depth_new[q]; ur = depth_new[q + 1]; bl = depth_new[q + c.W]; br = depth_new[q + c.W + 1];
      const int qn = pre_cvt_rne(floorf(rx + 0.5f) + floorf(ry + 0.5f) * (float)c.W);
      // sequential semantics: groups of 8 pixels in row-major order; an earlier group has already been rewritten
      nn = ((size_t)qn >> 3) < (p >> 3) ? est_in[qn] : orig_d[qn];
    }
    const float dx = rx - fxr, dy = ry - fyr;
    const bool smooth = ((ul - ur) < 0.1f) && ((ul - ur) > -0.1f) && ((ul - bl) < 0.1f) && ((ul - bl) > -0.1f) &&
                        ((ul - br) < 0.1f) && ((ul - br) > -0.1f);
    float bil = ((((1.0f - dx) * (1.0f - dy)) * ul + ((1.0f - dx) * dy) * ur) + (dx * (1.0f - dy)) * bl) + (dx * dy) * br;
    if (!smooth) bil = nn;
    const float diff = bil - V[2];
    const bool ok = (diff > (-0.05f) * V[2]) && (diff < 0.05f * V[2]);
    const float scale = bil / V[2];
    const float X = V[0] * scale - T.t[3], Y = V[1] * scale - T.t[7], Z = V[2] * scale - T.t[11];
    const float vZ = (T.t[2] * X + T.t[6] * Y) + T.t[10] * Z;
    const float w = orig_w[p];
    const float nd = ok ? (d * w + vZ) / (w + 1.0f) : d;
    const float nw = ok ? w + 1.0f : w;
    if (__float_as_uint(nd) != __float_as_uint(est_in[p])) any = true;
    est_out[p] = nd;
    w_out[p] = nw;
  }
  if (__builtin_amdgcn_ballot_w64(any) != 0ull && (threadIdx.x & 63) == 0) atomicOr(changed, 1u);
}

// ---- DatasetWrapper::framePreprocess (Tools/DatasetWrapper.hpp:186-263): the loader's depth pass.  Three launches on
// the handle's stream, nothing returns to the host: (1) u16 -> metres with the maximum-depth cut, and the image's
// min / max (non-negative floats order like their bit patterns); (2) the 4096-bin colour table of cv::bilateralFilter's
// CV_32FC1 path from that range; (3) the filter itself, one thread per pixel over a 16 x 16 tile staged in LDS with
// its halo (BORDER_REFLECT_101), taps in row-major order, and the write-back to the u16 map.  DESIGN.md s.5 states
// which OpenCV algorithm this restates and what is unpinned about it.
constexpr int kBfBins = 1 << 12;
constexpr int kBfMaxRadius = 7;
constexpr int kBfTile = 16;
struct BfTaps {
  int radius, maxk;
  float weight[(2 * kBfMaxRadius + 1) * (2 * kBfMaxRadius + 1)];
  int8_t di[(2 * kBfMaxRadius + 1) * (2 * kBfMaxRadius + 1)], dj[(2 * kBfMaxRadius + 1) * (2 * kBfMaxRadius + 1)];
};
struct BfState {  // device scratch header
  uint32_t min_bits, max_bits;
  float scale_index;
  uint32_t copy;  // max - min < FLT_EPSILON: the filter is the identity
};

__global__ __launch_bounds__(256) void k_pre_depth_metres(uint16_t* __restrict__ depth, size_t np, float cut, float depth_scale,
                                                          float* __restrict__ metres, BfState* st) {
  uint32_t mn = 0x7F800000u, mx = 0u;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (size_t)gridDim.x * 256) {
    uint16_t z = depth[p];
    if ((float)z > cut) { z = 0; depth[p] = 0; }
    const float m = (float)z / depth_scale;
    metres[p] = m;
    const uint32_t b = __float_as_uint(m);
    mn = b < mn ? b : mn;
    mx = b > mx ? b : mx;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t a = (uint32_t)__shfl_xor((int)mn, o), b = (uint32_t)__shfl_xor((int)mx, o);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&st->min_bits, mn); atomicMax(&st->max_bits, mx); }
}

__global__ __launch_bounds__(256) void k_pre_bilateral_table(BfState* st, float* __restrict__ lut, double gauss_color_coeff) {
  const float mn = __uint_as_float(st->min_bits), mx = __uint_as_float(st->max_bits);
  const bool copy = fabs((double)mn - (double)mx) < 1.1920928955078125e-7;  // FLT_EPSILON
  const float len = (float)((double)mx - (double)mn);
  const float scale_index = (float)kBfBins / len;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { st->scale_index = scale_index; st->copy = copy ? 1u : 0u; }
  if (copy || i >= kBfBins + 2) return;
  // (the reference stops evaluating after the first entry that underflows to 0; exp is monotone, so every later
  // entry evaluates to 0 as well)
  const double val = (double)((float)i / scale_index);
  lut[i] = (float)exp(val * val * gauss_color_coeff);
}

__device__ __forceinline__ int bf_reflect(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
  return p;
}

__global__ __launch_bounds__(kBfTile * kBfTile) void k_pre_bilateral(const float* __restrict__ src, int W, int H, BfTaps taps,
                                                                    const BfState* __restrict__ st,
                                                                    const float* __restrict__ lut, float depth_scale,
                                                                    float* __restrict__ refined, uint16_t* __restrict__ depth) {
  constexpr int kSide = kBfTile + 2 * kBfMaxRadius;
  __shared__ float tile[kSide * kSide];
  const int R = taps.radius, side = kBfTile + 2 * R;
  const int x0 = blockIdx.x * kBfTile, y0 = blockIdx.y * kBfTile;
  for (int q = threadIdx.x; q < side * side; q += kBfTile * kBfTile) {
    const int ty = q / side, tx = q - ty * side;
    tile[ty * kSide + tx] = src[(size_t)bf_reflect(y0 + ty - R, H) * W + bf_reflect(x0 + tx - R, W)];
  }
  __syncthreads();
  const int lx = threadIdx.x & (kBfTile - 1), ly = threadIdx.x / kBfTile;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= W || y >= H) return;
  const float val0 = tile[(ly + R) * kSide + lx + R];
  float out = val0;
  if (!st->copy) {
    const float scale_index = st->scale_index;
    float sum = 0.0f, wsum = 0.0f;
    for (int k = 0; k < taps.maxk; ++k) {
      const float val = tile[(ly + R + taps.di[k]) * kSide + lx + R + taps.dj[k]];
      float alpha = fabsf(val - val0) * scale_index;
      const int idx = (int)floorf(alpha);
      alpha -= (float)idx;
      const float l0 = lut[idx], l1 = lut[idx + 1];
      const float w = taps.weight[k] * (l0 + alpha * (l1 - l0));
      sum += val * w;
      wsum += w;
    }
    out = sum / wsum;
  }
  if (refined) refined[(size_t)y * W + x] = out;
  depth[(size_t)y * W + x] = (uint16_t)(out * depth_scale);  // float -> unsigned short (:250-251)
}

static PreCam pre_cam(const tf_volume* v) {
  PreCam c;
  c.W = v->cam.W; c.H = v->cam.H;
  c.fx = v->fx; c.fy = v->fy; c.cx = v->cx; c.cy = v->cy;  // the untruncated intrinsics (main.cpp passes camera.c_fx ...)
  return c;
}
static dim3 pre_grid(const PreCam& c) {
  const size_t np = (size_t)c.W * c.H;
  size_t b = (np + 255) / 256;
  if (b > 8192) b = 8192;
  return dim3((unsigned)b);
}

}  // namespace tf

using namespace tf;

extern "C" {

int tf_pre_normal_map(tf_volume* v, const float* d_depth, float* d_normal) {
  if (!v || !d_depth || !d_normal) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const PreCam c = pre_cam(v);
  hipLaunchKernelGGL(k_pre_normal_map, pre_grid(c), dim3(256), 0, v->stream, d_depth, c, d_normal);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_pre_refine_depth_normal(tf_volume* v, float* d_normal, float* d_depth) {
  if (!v || !d_depth || !d_normal) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const PreCam c = pre_cam(v);
  hipLaunchKernelGGL(k_pre_refine_depth_normal, pre_grid(c), dim3(256), 0, v->stream, d_normal, d_depth, c);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_pre_color_valid(tf_volume* v, const float* d_normal, uint8_t* d_flag) {
  if (!v || !d_flag || !d_normal) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const PreCam c = pre_cam(v);
  hipLaunchKernelGGL(k_pre_color_valid, pre_grid(c), dim3(256), 0, v->stream, d_normal, c, d_flag);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_pre_color_quality(tf_volume* v, const float* d_depth, const float* d_normal, const uint8_t* d_rgb,
                         float* d_quality) {
  if (!v || !d_depth || !d_normal || !d_rgb || !d_quality) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const PreCam c = pre_cam(v);
  if (c.W < 2 || c.H < 2) { set_error("image too small for a 3x3 derivative"); return TF_ERR_INVALID; }
  hipLaunchKernelGGL(k_pre_color_quality, pre_grid(c), dim3(256), 0, v->stream, d_depth, d_normal, d_rgb, c, d_quality);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_pre_refine_newframe(tf_volume* v, const float* d_depth_ref, float* d_depth_new, const float T_new_to_ref[12]) {
  if (!v || !d_depth_ref || !d_depth_new || !T_new_to_ref) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const PreCam c = pre_cam(v);
  PreT T;
  memcpy(T.t, T_new_to_ref, sizeof(T.t));
  hipLaunchKernelGGL(k_pre_refine_newframe, pre_grid(c), dim3(256), 0, v->stream, d_depth_ref, d_depth_new, c, T);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

int tf_pre_refine_keyframe(tf_volume* v, float* d_depth_ref, float* d_weight_ref, const float* d_depth_new,
                           const float T_ref_to_new[12], int32_t* rounds) {
  if (!v || !d_depth_ref || !d_weight_ref || !d_depth_new || !T_ref_to_new) { set_error("null argument"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const PreCam c = pre_cam(v);
  const size_t np = (size_t)c.W * c.H, bytes = np * sizeof(float);
  // scratch: original depth | estimate A | estimate B | new weight | flag
  int rc = ensure_tmp(v, 4 * bytes + 64);
  if (rc) return rc;
  float* orig = reinterpret_cast<float*>(v->d_tmp);
  float* est[2] = {orig + np, orig + 2 * np};
  float* wout = orig + 3 * np;
  uint32_t* flag = reinterpret_cast<uint32_t*>(orig + 4 * np);
  PreT T;
  memcpy(T.t, T_ref_to_new, sizeof(T.t));
  TF_HIP(hipMemcpyAsync(orig, d_depth_ref, bytes, hipMemcpyDeviceToDevice, v->stream));
  const float* in = orig;  // round 0: nothing rewritten yet
  int k = 0;
  for (;; ++k) {
    if (k >= 256) { set_error("tf_pre_refine_keyframe: the in-place dependency chain did not settle in 256 rounds"); return TF_ERR_INVALID; }
    TF_HIP(hipMemsetAsync(flag, 0, 4, v->stream));
    float* out = est[k & 1];
    hipLaunchKernelGGL(k_pre_refine_keyframe, pre_grid(c), dim3(256), 0, v->stream, orig, d_weight_ref, d_depth_new, in,
                       out, wout, c, T, flag);
    TF_HIP(hipGetLastError());
    uint32_t h = 0;
    TF_HIP(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, v->stream));
    TF_HIP(hipStreamSynchronize(v->stream));
    in = out;
    // round k compared its output with its input estimate: equal everywhere = the fixed point (round 0 compares
    // with the original depth, so a frame that changes nothing ends at once)
    if (!h) break;
  }
  TF_HIP(hipMemcpyAsync(d_depth_ref, in, bytes, hipMemcpyDeviceToDevice, v->stream));
  TF_HIP(hipMemcpyAsync(d_weight_ref, wout, bytes, hipMemcpyDeviceToDevice, v->stream));
  if (rounds) *rounds = k + 1;
  return TF_OK;
}

int tf_pre_frame_depth(tf_volume* v, uint16_t* d_depth, float* d_refined, float maximum_depth, float depth_scale, int d,
                       double sigma_color, double sigma_space) {
  if (!v || !d_depth) { set_error("null argument"); return TF_ERR_INVALID; }
  if (!(depth_scale > 0.0f)) { set_error("depth_scale must be positive"); return TF_ERR_INVALID; }
  TF_DEV(v);
  const PreCam c = pre_cam(v);
  if (sigma_color <= 0) sigma_color = 1;
  if (sigma_space <= 0) sigma_space = 1;
  int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
  if (radius < 1) radius = 1;
  if (radius > kBfMaxRadius) { set_error("bilateral filter: diameter above 15 is not supported"); return TF_ERR_INVALID; }
  BfTaps taps;
  memset(&taps, 0, sizeof(taps));
  taps.radius = radius;
  const double gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
  for (int i = -radius; i <= radius; ++i)
    for (int j = -radius; j <= radius; ++j) {
      const double r = sqrt((double)i * i + (double)j * j);
      if (r > radius) continue;
      taps.weight[taps.maxk] = (float)exp(r * r * gauss_space_coeff);
      taps.di[taps.maxk] = (int8_t)i;
      taps.dj[taps.maxk] = (int8_t)j;
      ++taps.maxk;
    }
  const size_t np = (size_t)c.W * c.H;
  const size_t lut_at = 64, img_at = lut_at + ((kBfBins + 2) * sizeof(float) + 63) / 64 * 64;
  int rc = ensure_tmp(v, img_at + np * sizeof(float));
  if (rc) return rc;
  uint8_t* base = reinterpret_cast<uint8_t*>(v->d_tmp);
  BfState* st = reinterpret_cast<BfState*>(base);
  float* lut = reinterpret_cast<float*>(base + lut_at);
  float* metres = reinterpret_cast<float*>(base + img_at);
  const BfState init = {0x7F800000u, 0u, 0.0f, 0u};
  TF_HIP(hipMemcpyAsync(st, &init, sizeof(init), hipMemcpyHostToDevice, v->stream));
  hipLaunchKernelGGL(k_pre_depth_metres, pre_grid(c), dim3(256), 0, v->stream, d_depth, np, maximum_depth * depth_scale,
                     depth_scale, metres, st);
  hipLaunchKernelGGL(k_pre_bilateral_table, dim3((kBfBins + 2 + 255) / 256), dim3(256), 0, v->stream, st, lut,
                     -0.5 / (sigma_color * sigma_color));
  hipLaunchKernelGGL(k_pre_bilateral, dim3((c.W + kBfTile - 1) / kBfTile, (c.H + kBfTile - 1) / kBfTile), dim3(kBfTile * kBfTile),
                     0, v->stream, metres, c.W, c.H, taps, st, lut, depth_scale, d_refined, d_depth);
  TF_HIP(hipGetLastError());
  return TF_OK;
}

}  // extern "C"
