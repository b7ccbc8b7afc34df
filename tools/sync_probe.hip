// What one "launch a tiny kernel, wait for it" round trip costs the host, by the way of waiting:
//   hipcc --offload-arch=gfx950 -O2 tools/sync_probe.hip -o tools/sync_probe && tools/sync_probe
// (the call-by-call entry points of the C ABI wait for the device once per call, 13 calls per keyframe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_stamp(unsigned* host_word, unsigned v) {
  if (threadIdx.x == 0) __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  if (mode == 1) hipSetDeviceFlags(hipDeviceScheduleSpin);
  if (mode == 2) hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned* w; hipHostMalloc((void**)&w, 64, hipHostMallocDefault); *w = 0;
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int N = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now_us();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, w, 1u); hipStreamSynchronize(s); }
    double t1 = now_us();
    unsigned seq = *w;
    for (int i = 0; i < N; ++i) {
      const unsigned want = 1000000u + (unsigned)(rep * N + i);
      hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, w, want);
      while (__atomic_load_n(w, __ATOMIC_ACQUIRE) != want) {}
    }
    double t2 = now_us();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, w, 1u); hipEventRecord(ev, s); hipEventSynchronize(ev); }
    double t3 = now_us();
    (void)seq;
    if (rep) printf("mode %d: launch + hipStreamSynchronize %.2f us, launch + spin on a pinned word %.2f us, launch + event record + hipEventSynchronize %.2f us\n",
                    mode, (t1 - t0) / N, (t2 - t1) / N, (t3 - t2) / N);
  }
  return 0;
}
