// Stress of tf::CopyPool (texturefusion_amd/csrc/tf_copy_pool.h): alternating task counts (a depth-only frame
// followed by a depth + RGBA frame, the keyframe / local-frame pattern), every copy checked byte for byte.
// Built with -fsanitize=thread by tests/test_copy_pool.py: a helper that is scheduled late must never touch
// the task table while the caller rebuilds it.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../texturefusion_amd/csrc/tf_copy_pool.h"

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  const size_t npix = 640 * 480;
  std::vector<unsigned char> a(npix * 4), b(npix * 4), da(npix * 4), db(npix * 4);
  tf::CopyPool pool(3);
  unsigned seed = 12345u;
  for (int it = 0; it < iters; ++it) {
    for (size_t i = 0; i < a.size(); i += 97) { seed = seed * 1664525u + 1013904223u; a[i] = (unsigned char)(seed >> 24); b[i] = (unsigned char)(seed >> 16); }
    void* dst[2] = {da.data(), db.data()};
    const void* src[2] = {a.data(), b.data()};
    // odd calls: one small region (2 tasks); even calls: two full regions (10 tasks)
    size_t nb[2] = {(it & 1) ? (size_t)300000 : npix * 4, npix * 4};
    const int nr = (it & 1) ? 1 : 2;
    pool.copy(dst, src, nb, nr);
    if (memcmp(da.data(), a.data(), nb[0]) != 0 || (nr == 2 && memcmp(db.data(), b.data(), nb[1]) != 0)) {
      printf("COPY MISMATCH at iteration %d\n", it);
      return 1;
    }
  }
  printf("COPY POOL OK\n");
  return 0;
}
