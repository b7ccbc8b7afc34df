// tf_copy_pool.h -- a few helper threads for the host-side staging copy of tf_integrate_frame_host.
//
// The reference hands over pageable host images (cv::Mat data); they have to be copied into pinned memory before
// the asynchronous H2D transfer, and one thread copies a 2.4 MB frame out of DRAM at ~5 GB/s -- slower than the
// device processes it.  The pool splits a copy into equal parts; the calling thread takes parts too.
#ifndef TF_COPY_POOL_H_
#define TF_COPY_POOL_H_
#include <pthread.h>
#include <sched.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace tf {

class CopyPool {
 public:
  // pin_near: keep the helpers on CPUs next to the calling thread's (the same group of eight logical CPUs, which
  // shares a last-level cache slice on the hosts this runs on): helpers the scheduler parks on another socket halve the
  // copy rate (measured: 37 vs 66-75 us per 2.4 MB frame from run to run).  1 = every helper may run on any CPU of the
  // group (the scheduler moves a helper whose CPU is taken by another process); 2 = one CPU per helper (a helper
  // preempted in the middle of its part then stalls the call for a time slice: on a shared host 1 call in ~1000 took
  // 10+ ms).  Only CPUs the process may use are taken; with fewer than two of them in the group nothing is pinned.
  // spin_us: how long a helper that finished its part keeps polling for the next call before it goes to sleep on the
  // condition variable.  A stream of frames calls every ~100 us and a futex wake-up takes 20-50 us -- most of a staging
  // copy; helpers that are still awake start at once (30-49 -> ~20 us per 2.4 MB frame).  An idle caller costs nothing
  // after spin_us.
  explicit CopyPool(int helpers, int pin_near = 1, int spin_us = 200) : spin_us_(spin_us) {
    std::vector<int> near;
    if (pin_near) {
      cpu_set_t allowed;
      const int me = sched_getcpu();
      if (me >= 0 && sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = (me / 8) * 8; c < (me / 8) * 8 + 8; ++c)
          if ((pin_near == 1 || c != me) && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) near.push_back(c);
      if (near.size() < 2) near.clear();
    }
    for (int i = 0; i < helpers; ++i) {
      workers_.emplace_back([this] { loop(); });
      if (!near.empty()) {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (pin_near == 1) for (int c : near) CPU_SET(c, &set);
        else CPU_SET(near[(size_t)i % near.size()], &set);
        pthread_setaffinity_np(workers_.back().native_handle(), sizeof(set), &set);  // best effort
      }
    }
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  CopyPool(const CopyPool&) = delete;
  CopyPool& operator=(const CopyPool&) = delete;

  // dst[i] <- src[i] for the given regions (up to 4), each split into parts of 128 KiB.
  // The task table is only ever rebuilt under the mutex while no helper is inside pull() (active_ == 0), and a
  // helper only enters pull() after registering under the same mutex: a helper that was notified for call N but is
  // scheduled during call N + 1 either registers before the rebuild (and finds next_ >= size: nothing to do, the
  // rebuild waits for it to leave) or after it (and works on call N + 1's table).
  void copy(void* const* dst, const void* const* src, const size_t* bytes, int regions) {
    size_t ntask = 0;
    for (int r = 0; r < regions; ++r) ntask += (bytes[r] + kPart - 1) / kPart;
    if (!ntask) return;
    if (workers_.empty() || ntask < 2) {
      for (int r = 0; r < regions; ++r) memcpy(dst[r], src[r], bytes[r]);
      return;
    }
    {
      std::unique_lock<std::mutex> g(m_);
      while (active_.load(std::memory_order_acquire) > 0) {  // stragglers of the previous call leave pull() first
        g.unlock();
        std::this_thread::yield();
        g.lock();
      }
      tasks_.clear();
      for (int r = 0; r < regions; ++r)
        for (size_t o = 0; o < bytes[r]; o += kPart)
          tasks_.push_back({static_cast<char*>(dst[r]) + o, static_cast<const char*>(src[r]) + o,
                            bytes[r] - o < kPart ? bytes[r] - o : kPart});
      next_.store(0, std::memory_order_relaxed);
      left_.store((int)tasks_.size(), std::memory_order_release);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    pull();
    while (left_.load(std::memory_order_acquire) > 0) std::this_thread::yield();  // every part has been copied
  }

 private:
  static constexpr size_t kPart = 128u << 10;
  struct Task { char* d; const char* s; size_t n; };
  void pull() {
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= (int)tasks_.size()) return;
      memcpy(tasks_[i].d, tasks_[i].s, tasks_[i].n);
      left_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      if (spin_us_ > 0) {  // poll for the next call for a while (gen_ only ever changes under the mutex; registration below too)
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0; gen_.load(std::memory_order_acquire) == seen; ++spin) {
          __builtin_ia32_pause();
          if ((spin & 255u) == 255u &&
              std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us_)
            break;
        }
      }
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return gen_.load(std::memory_order_relaxed) != seen; });
        seen = gen_.load(std::memory_order_relaxed);
        if (stop_) return;
        active_.fetch_add(1, std::memory_order_acq_rel);  // registered under the mutex: the table is stable from here on
      }
      pull();
      active_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  std::vector<std::thread> workers_;
  std::vector<Task> tasks_;
  std::atomic<int> next_{0}, left_{0}, active_{0};
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<unsigned long> gen_{0};
  bool stop_ = false;
  const int spin_us_;
};

}  // namespace tf
#endif  // TF_COPY_POOL_H_
