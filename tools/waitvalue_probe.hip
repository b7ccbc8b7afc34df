// Does hipStreamWaitValue32 work on this stack, and what does a cross-stream dependency cost through it compared with an event?
//   hipcc --offload-arch=gfx950 -O2 tools/waitvalue_probe.hip -o tools/waitvalue_probe && tools/waitvalue_probe
// Stream A: [busy kernel ~60 us, its last thread stores `k` into a word] [second kernel ~20 us]
// Stream B: waits for the word (wait-value) or for an event recorded between A's two kernels, then runs a 5-us kernel.
// Reported: A's wall time for 200 rounds (does the dependency cost A anything?), and B's completion.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void busy(int us, uint32_t* word, uint32_t val, uint32_t* ticket) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) {}
  if (word) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(ticket, 1u) == gridDim.x - 1u) { *ticket = 0u; __hip_atomic_store(word, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
}
int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  uint32_t *word = nullptr, *ticket = nullptr;
  CK(hipExtMallocWithFlags((void**)&word, 8, hipMallocSignalMemory));
  CK(hipMalloc((void**)&ticket, 4));
  CK(hipMemset(word, 0, 8));
  CK(hipMemset(ticket, 0, 4));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int R = 200;
  for (int mode = 0; mode < 3; ++mode) {  // 0: no dependency, 1: event, 2: wait-value
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= R; ++r) {
      const uint32_t val = (uint32_t)(mode * 1000 + r);
      hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, A, 60, mode == 2 ? word : nullptr, val, ticket);
      if (mode == 1) { CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0)); }
      if (mode == 2) CK(hipStreamWaitValue32(B, word, val, hipStreamWaitValueGte, 0xFFFFFFFFu));
      hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, A, 20, (uint32_t*)nullptr, 0u, ticket);
      hipLaunchKernelGGL(busy, dim3(64), dim3(64), 0, B, 5, (uint32_t*)nullptr, 0u, ticket);
    }
    CK(hipStreamSynchronize(A));
    const double usA = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
    CK(hipStreamSynchronize(B));
    const double usB = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
    printf("mode %d (%s): stream A %.1f us per round (80 us of kernels), both streams done %.1f\n", mode,
           mode == 0 ? "independent" : mode == 1 ? "event between A's kernels" : "wait-value on a word A's first kernel stores", usA, usB);
  }
  return 0;
}
