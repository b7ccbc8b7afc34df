// The mesher filter's access shape with no tests in it: one wave per dirty-list entry (8.4 k per S-room frame): the entry
// (16 B), eight 16-byte hash-entry loads at unrelated positions (lanes 0-7), eight 4-byte summary loads that depend on them,
// for about half of the entries the chunk's own 4 KiB (64 lanes x 64 B, dependent on the first probe), one 64-byte record
// out.  2560 workgroups of four waves, as launch_mesh sizes the wave form.
//   hipcc --offload-arch=gfx950 -O3 tools/filter_shape_bench.hip -o tools/filter_shape_bench && tools/filter_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_filter_shape(const uint4* __restrict__ list, const uint4* __restrict__ hent, unsigned hmask,
                                                      const unsigned* __restrict__ summ, const uint4* __restrict__ pool, unsigned pmask,
                                                      uint4* __restrict__ rec, unsigned n, unsigned exact_of_256) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
  for (unsigned e = wave; e < n; e += nwaves) {
    const uint4 id = list[e];                                              // hop 1: the entry
    const unsigned h = (id.x * 2654435761u + (lane & 7u) * 40503u + id.y) & hmask;
    const uint4 he = hent[h];                                              // hop 2: eight hash entries
    const unsigned slot = he.z & pmask;
    unsigned s = summ[slot];                                               // hop 3: their summaries
    for (int o = 1; o < 8; o <<= 1) s |= __shfl_xor(s, o);
    unsigned acc = s;
    const unsigned own = __shfl((int)slot, 0);
    if (((id.x >> 3) & 255u) < exact_of_256) {                             // hop 3': the chunk's own voxels (exact test)
      const uint4* q = pool + (size_t)own * 256;
      uint4 a = q[lane], b = q[64 + lane], c = q[128 + lane], d = q[192 + lane];
      acc ^= a.x ^ b.y ^ c.z ^ d.w;
      for (int o = 1; o < 64; o <<= 1) acc |= __shfl_xor(acc, o);
    }
    if (lane < 4) rec[(size_t)own * 4 + lane] = make_uint4(acc, id.x, id.y, s);  // hop 4: the record (64 B)
  }
}

int main() {
  const unsigned pool_n = 1u << 18, hcap = 1u << 20;
  uint4 *list, *hent, *pool, *rec; unsigned* summ;
  CK(hipMalloc(&pool, (size_t)pool_n * 4096)); CK(hipMemset(pool, 1, (size_t)pool_n * 4096));
  CK(hipMalloc(&hent, (size_t)hcap * 16)); CK(hipMalloc(&summ, (size_t)pool_n * 4)); CK(hipMalloc(&rec, (size_t)pool_n * 64));
  std::vector<uint4> hh(hcap);
  for (unsigned i = 0; i < hcap; ++i) hh[i] = make_uint4(i, 0, (unsigned)(((unsigned long long)i * 2654435761ull) >> 7), 1);
  CK(hipMemcpy(hent, hh.data(), (size_t)hcap * 16, hipMemcpyHostToDevice)); CK(hipMemset(summ, 3, (size_t)pool_n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (unsigned n : {8400u, 69000u}) {
    std::vector<uint4> hl((size_t)n * 8);
    for (size_t i = 0; i < hl.size(); ++i) hl[i] = make_uint4((unsigned)(i * 2654435761ull >> 3), (unsigned)(i * 40503ull), (unsigned)i, 0);
    CK(hipMalloc(&list, hl.size() * 16)); CK(hipMemcpy(list, hl.data(), hl.size() * 16, hipMemcpyHostToDevice));
    for (unsigned ex : {0u, 122u, 256u}) {                                // nobody / 48 % (n_exact / n_dirty of the room stream) / everybody
      const unsigned grid = (n + 3) / 4 < 2560u ? (n + 3) / 4 : 2560u;
      float best = 1e9f;
      for (int rep = 0; rep < 8; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_filter_shape, dim3(grid), dim3(256), 0, 0, list + (size_t)(rep % 8) * n, hent, hcap - 1, summ, pool, pool_n - 1, rec, n, ex);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
      }
      printf("%6u entries, own voxels read for %3u/256 of them: %6.1f us\n", n, ex, best * 1e3);
    }
    CK(hipFree(list));
  }
  // an empty kernel between the same events: what the measurement itself costs
  float best = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_filter_shape, dim3(1), dim3(256), 0, 0, (const uint4*)pool, hent, 0u, summ, pool, 0u, rec, 0u, 0u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  printf("empty launch between the same events: %6.1f us\n", best * 1e3);
  return 0;
}
