// What rate does the memory system sustain for K-A's access shape?  One wave per scattered 8-KiB chunk (two 4-KiB planes:
// sdf + weight, colour), eight passes of 512 contiguous bytes per plane (lane = 8 bytes), a pass's 64-byte ROWS read and
// written back only where a per-row mask says so (exec-masked buffer-style accesses, as in integrate_body), a persistent
// grid of 7 waves per SIMD striding over the chunk list -- and no arithmetic.  Prints algorithmic bytes / time for several
// fractions of rewritten rows; compare with the streaming RMW of the same volume of data.
//   hipcc --offload-arch=gfx950 -O3 tools/row_rmw_bench.hip -o tools/row_rmw_bench && tools/row_rmw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_rows(uint2* __restrict__ planeA, uint2* __restrict__ planeB, const unsigned* __restrict__ perm,
                                              unsigned nchunks, unsigned thresh /* of 256: rows rewritten */, int both) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
  for (unsigned c = wave; c < nchunks; c += nwaves) {
    const size_t base = (size_t)perm[c] * 512u;  // voxels of 8 bytes per plane
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const unsigned row = (unsigned)g * 8u + (lane >> 3);
      const unsigned h = (perm[c] * 2654435761u + row * 40503u) >> 24;  // per-row pseudo-random byte
      if (h < thresh) {
        const size_t i = base + (size_t)g * 64u + lane;
        uint2 a = planeA[i];
        a.x += 1u;
        planeA[i] = a;
        if (both) { uint2 b = planeB[i]; b.y += 1u; planeB[i] = b; }
      }
    }
  }
}
__global__ void __launch_bounds__(256) k_stream(uint4* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = p[i]; v.x += 1u; p[i] = v; }
}

int main() {
  const unsigned pool = 1u << 19;                  // chunks in the pool: 2 GiB per plane
  const unsigned nlist = 80000;                    // chunks touched per launch (several room frames' worth)
  uint2 *A, *B; unsigned* perm;
  CK(hipMalloc(&A, (size_t)pool * 4096)); CK(hipMalloc(&B, (size_t)pool * 4096));
  CK(hipMemset(A, 1, (size_t)pool * 4096)); CK(hipMemset(B, 1, (size_t)pool * 4096));
  std::vector<unsigned> h(nlist * 16);
  for (unsigned i = 0; i < nlist * 16; ++i) h[i] = (unsigned)(((unsigned long long)(i + 1) * 2654435761ull) % pool);
  CK(hipMalloc(&perm, 4 * h.size())); CK(hipMemcpy(perm, h.data(), 4 * h.size(), hipMemcpyHostToDevice));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int grid = pr.multiProcessorCount * 7;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float pair = 1e9f;  // what an (almost) empty launch costs between the same two events: taken off below
  for (int rep = 0; rep < 8; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rows, dim3(1), dim3(256), 0, 0, A, B, perm, 0u, 0u, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < pair) pair = ms;
  }
  printf("%s, %d CUs, grid %d x 256 (7 waves per SIMD); an empty launch between the two events: %.1f us (taken off in the last column)\n", pr.name, pr.multiProcessorCount, grid, pair * 1e3);
  for (unsigned nl : {nlist, 11000u})  // 80 k chunks (ramp and tail amortised) and one S-room frame's worth
  for (int both = 0; both <= 1; ++both)
    for (unsigned th : {256u, 192u, 128u, 64u, 32u}) {
      float best = 1e9f;
      double rows = 0;
      for (int rep = 0; rep < 6; ++rep) {
        const unsigned* pl = perm + (size_t)(rep % 16) * nlist;  // a different set of chunks every time: nothing is cached
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, 0, A, B, pl, nl, th, both);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      rows = (double)nl * 64.0 * th / 256.0;
      const double bytes = rows * 128.0 * (both ? 2 : 1);  // read + write of 64 B per rewritten row and plane
      printf("%6u chunks, planes %d, %3u/256 of the rows rewritten: %7.1f us, %6.2f TB/s algorithmic (%.0f MB) | without the launch: %6.1f us, %5.2f TB/s\n", nl, both + 1, th, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / 1e6, (best - pair) * 1e3, bytes / ((best - pair) * 1e-3) / 1e12);
    }
  {  // the same amount of data as a streaming read-modify-write
    const size_t n = (size_t)nlist * 8192 / 16;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, reinterpret_cast<uint4*>(A) + (size_t)(rep % 3) * n, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("streaming RMW of the same %.0f MB: %7.1f us, %6.2f TB/s | without the launch: %6.1f us, %5.2f TB/s\n", (double)n * 16 / 1e6, best * 1e3, (double)n * 32 / (best * 1e-3) / 1e12, (best - pair) * 1e3, (double)n * 32 / ((best - pair) * 1e-3) / 1e12);
  }
  return 0;
}
