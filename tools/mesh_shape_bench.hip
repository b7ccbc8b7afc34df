// What does the memory system give a launch of the mesher's size and access shape, with no marching cubes in it?  One
// 128-thread workgroup per surviving chunk (2929 per S-room frame, ten resident per CU): the chunk's own 4 KiB of
// {sdf, weight}, 410 scattered 16-byte halo loads out of its 26 neighbours' chunks (slots known: no dependent probes),
// everything staged through LDS with one barrier, then 4.2 KB of vertices / triangles written to the chunk's mesh block.
//   hipcc --offload-arch=gfx950 -O3 tools/mesh_shape_bench.hip -o tools/mesh_shape_bench && tools/mesh_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(128, 5) k_mesh_shape(const uint4* __restrict__ pool, const unsigned* __restrict__ slots /* [n][27] */,
                                                       uint4* __restrict__ mesh, unsigned nchunks, unsigned halo_loads) {
  __shared__ uint4 stage[256 + 416];
  const unsigned t = threadIdx.x;
  for (unsigned c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const unsigned* sl = slots + (size_t)c * 27;
    const uint4* own = pool + (size_t)sl[0] * 256;        // 4 KiB = 256 x 16 B
    stage[t] = own[t];
    stage[128 + t] = own[128 + t];
    if (halo_loads == 411u) {
      // variant: the two x-neighbours read WHOLE (2 x 256 coalesced granules: the same 64 lines, half the line requests),
      // the other 282 loads as below
      for (unsigned k = t; k < 512u; k += 128) { const unsigned nb = 1u + (k >> 8); const uint4 v = pool[(size_t)sl[nb] * 256 + (k & 255u)]; if ((k & 3u) == (nb == 1u ? 0u : 3u)) stage[256 + (k >> 2)] = v; }
      for (unsigned k = 128u + t; k < 410u; k += 128) {
        unsigned nb, g;
        if (k < 192u) { const unsigned q = k - 128u; nb = 3; g = ((q >> 3) * 8u + ((q >> 2) & 1u)) * 4u + (q & 3u); }
        else if (k < 224u) { const unsigned q = k - 192u; nb = 4; g = ((q >> 2) * 8u + 7u) * 4u + (q & 3u); }
        else if (k < 288u) { nb = 5; g = k - 224u; }
        else if (k < 320u) { nb = 6; g = 224u + (k - 288u); }
        else { const unsigned q = k - 320u; nb = 7u + q % 20u; g = ((q / 20u) * 8u) * 4u; }
        stage[256 + k] = pool[(size_t)sl[nb] * 256 + g];
      }
    } else if (halo_loads == 410u) {
      // the halo as the mesher reads it (x fastest: a 64-byte row holds the 8 voxels of one (z, y); 16-byte granule = 2 voxels):
      // +x / -x: first / last granule of all 64 rows; +y: rows (z, 0) and (z, 1); -y: rows (z, 7); +z: slices 0, 1; -z: slice 7;
      // 12 edge and 8 corner neighbours: a few granules each
      for (unsigned k = t; k < 410u; k += 128) {
        unsigned nb, g;
        if (k < 64u) { nb = 1; g = k * 4u; }                                   // +x: voxels x = 0, 1 of every row
        else if (k < 128u) { nb = 2; g = (k - 64u) * 4u + 3u; }                // -x: voxels x = 6, 7
        else if (k < 192u) { const unsigned q = k - 128u; nb = 3; g = ((q >> 3) * 8u + ((q >> 2) & 1u)) * 4u + (q & 3u); }  // +y: rows (z, 0), (z, 1)
        else if (k < 224u) { const unsigned q = k - 192u; nb = 4; g = ((q >> 2) * 8u + 7u) * 4u + (q & 3u); }               // -y: rows (z, 7)
        else if (k < 288u) { nb = 5; g = k - 224u; }                           // +z: slices z = 0, 1 (64 granules)
        else if (k < 320u) { nb = 6; g = 224u + (k - 288u); }                  // -z: slice z = 7
        else { const unsigned q = k - 320u; nb = 7u + q % 20u; g = ((q / 20u) * 8u) * 4u; }  // edges / corners: row starts
        stage[256 + k] = pool[(size_t)sl[nb] * 256 + g];
      }
    } else {
    for (unsigned k = t; k < halo_loads; k += 128) {      // pairs of voxels at RANDOM granules of the neighbours
      const unsigned nb = 1 + (k * 7u + c) % 26u;
      const unsigned row = (k * 2654435761u + c * 40503u) >> 24;  // 0..255: a 16-byte granule of the neighbour's 4 KiB
      stage[256 + k] = pool[(size_t)sl[nb] * 256 + row];
    }
    }
    __syncthreads();
    // "mesh": 4.2 KB out = 264 x 16 B, made of what was staged (so nothing can be dropped)
    uint4* out = mesh + (size_t)sl[0] * 1280;             // 20 KiB per pool slot
    for (unsigned k = t; k < 264; k += 128) {
      uint4 a = stage[k % 256], b = stage[256 + (k % 410)];
      out[k] = make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
    }
    __syncthreads();
  }
}

int main() {
  const unsigned pool = 1u << 18;                          // 1 GiB of voxel chunks, 5 GiB of mesh blocks
  uint4 *P, *M; unsigned* S;
  CK(hipMalloc(&P, (size_t)pool * 4096)); CK(hipMalloc(&M, (size_t)pool * 20480)); CK(hipMemset(P, 1, (size_t)pool * 4096));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float pair = 1e9f;  // an (almost) empty launch between the same two events
  {
    unsigned* S0; CK(hipMalloc(&S0, 4 * 27));
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_mesh_shape, dim3(1), dim3(128), 0, 0, P, S0, M, 0u, 0u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < pair) pair = ms;
    }
    printf("an empty launch between the two events: %.1f us\n", pair * 1e3);
  }
  for (unsigned n : {2929u, 11200u}) {                     // a room frame's survivors; the hall's
    std::vector<unsigned> h((size_t)n * 27 * 8);
    // neighbours of a chunk sit at unrelated pool slots (the pool is filled in visiting order), every set of chunks is new
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(((unsigned long long)(i + 12345) * 2654435761ull) % pool);
    CK(hipMalloc(&S, 4 * h.size())); CK(hipMemcpy(S, h.data(), 4 * h.size(), hipMemcpyHostToDevice));
    for (unsigned halo : {410u, 411u, 409u, 0u}) {  // the mesher's halo; as many loads at random granules; none
      const int grid = n < 4096u ? (int)n : 4096;
      float best = 1e9f;
      for (int rep = 0; rep < 8; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mesh_shape, dim3(grid), dim3(128), 0, 0, P, S + (size_t)(rep % 8) * n * 27, M, n, halo);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      const double bytes = (double)n * (4096.0 + halo * 16.0 + 264 * 16.0);
      printf("%5u chunks, %3u halo loads each%s: %6.1f us, %5.2f TB/s of requested bytes (%.1f MB)\n", n, halo, halo == 410u ? " (faces / edges as the mesher reads them)" : halo == 411u ? " (the same, x-neighbours read whole)" : halo ? " (random granules)" : "", best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / 1e6);
      printf("        without the launch: %6.1f us\n", (best - pair) * 1e3);
    }
    CK(hipFree(S));
  }
  return 0;
}
