// FETCH_SIZE / WRITE_SIZE calibration for gfx950 (MI355X_MICROARCH.md, HBM section: only the 16-B/lane streaming
// read is calibrated there).  Each kernel moves a KNOWN number of bytes, far larger than the 256 MiB Infinity
// Cache, in one of the access shapes the fusion kernels use; run under
//     rocprofv3 --pmc FETCH_SIZE  --output-format csv -d <dir> -- tools/calib_fetch
//     rocprofv3 --pmc WRITE_SIZE  --output-format csv -d <dir> -- tools/calib_fetch
// and divide the counter by the bytes printed here (tools/pmc_summary.py does it).
//   hipcc --offload-arch=gfx950 -O2 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> __global__ void __launch_bounds__(256) k_calib_read(const T* __restrict__ p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T v = p[i];
    const unsigned* w = reinterpret_cast<const unsigned*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}
template <typename T> __global__ void __launch_bounds__(256) k_calib_write(T* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T v; unsigned* w = reinterpret_cast<unsigned*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = (unsigned)i + k;
    p[i] = v;
  }
}
// one workgroup per 4-KiB block at a scattered position (the chunk-pool access of K-B / the mesher): read the
// block as 256 lanes x 16 B, optionally write it back (read-modify-write of a voxel chunk)
template <bool WRITE> __global__ void __launch_bounds__(256) k_calib_chunks(float4* __restrict__ pool, const unsigned* __restrict__ perm,
                                                                           unsigned nblk, unsigned* sink) {
  for (unsigned b = blockIdx.x; b < nblk; b += gridDim.x) {
    float4* q = pool + (size_t)perm[b] * 256 + threadIdx.x;
    float4 v = *q;
    if (WRITE) { v.x += 1.f; v.w += 1.f; *q = v; }
    else if (v.x == 123.456f) *sink = 1;
  }
}
// a 4-byte gather per lane at pseudo-random positions of a large image (the patch projection's tap pattern,
// worst case: every lane its own 64-B sector)
__global__ void __launch_bounds__(256) k_calib_gather4(const unsigned* __restrict__ p, size_t n_words, size_t n_gathers, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_gathers; i += (size_t)gridDim.x * 256) {
    size_t h = (i * 0x9E3779B97F4A7C15ull) >> 20;
    acc ^= p[h % n_words];
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30;  // 2 GiB per pass
  void* buf; unsigned* sink; unsigned* perm;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes));
  const unsigned nblk = (unsigned)(bytes / 4096);
  unsigned* hperm = (unsigned*)malloc(4 * (size_t)nblk);
  for (unsigned i = 0; i < nblk; ++i) hperm[i] = (unsigned)(((unsigned long long)i * 2654435761ull) % nblk);  // odd multiplier, nblk a power of two: a permutation
  CK(hipMalloc(&perm, 4 * (size_t)nblk)); CK(hipMemcpy(perm, hperm, 4 * (size_t)nblk, hipMemcpyHostToDevice));
  const int grid = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    k_calib_read<float4><<<grid, 256>>>((const float4*)buf, bytes / 16, sink);
    k_calib_read<float2><<<grid, 256>>>((const float2*)buf, bytes / 8, sink);
    k_calib_read<float><<<grid, 256>>>((const float*)buf, bytes / 4, sink);
    k_calib_write<float4><<<grid, 256>>>((float4*)buf, bytes / 16);
    k_calib_write<float2><<<grid, 256>>>((float2*)buf, bytes / 8);
    k_calib_write<float><<<grid, 256>>>((float*)buf, bytes / 4);
    k_calib_chunks<false><<<grid, 256>>>((float4*)buf, perm, nblk, sink);
    k_calib_chunks<true><<<grid, 256>>>((float4*)buf, perm, nblk, sink);
    k_calib_gather4<<<grid, 256>>>((const unsigned*)buf, bytes / 4, bytes / 64, sink);
  }
  CK(hipDeviceSynchronize());
  printf("{\"bytes_per_launch\": %zu, \"gather4_lanes\": %zu, \"note\": \"k_calib_read/write<T>: bytes; k_calib_chunks<0>: bytes read; "
         "k_calib_chunks<1>: bytes read + bytes written; k_calib_gather4: lanes x 4 B useful, lanes x 64 B of sectors\"}\n", bytes, bytes / 64);
  return 0;
}
