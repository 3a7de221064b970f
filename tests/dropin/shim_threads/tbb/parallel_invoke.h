// tests/dropin/shim_threads/tbb/parallel_invoke.h -- TEST INFRASTRUCTURE.  Stand-in for tbb::parallel_invoke(f0, f1) that really runs the
// two functors on two threads, like TBB does: f0 on a persistent worker thread, f1 on the caller.  The engine build of the reference's
// front end (tests/dropin/Makefile) uses it so that LocalTracker's two concurrent match() calls (dvo_slam/src/local_tracker.cpp:180-184)
// reach the engine the way they do under the real TBB -- from two threads at once.  (The CPU twin keeps the serial stand-in of
// oracle/shim/tbb: its numbers are labelled accordingly.)
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace tbb {
namespace shim_detail {
class Worker {
 public:
  static Worker& instance() {
    static Worker w;
    return w;
  }
  void run(std::function<void()> job) {
    std::unique_lock<std::mutex> lock(m_);
    job_ = std::move(job);
    have_job_ = true;
    done_ = false;
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lock(m_);
    cv_.wait(lock, [this] { return done_; });
  }
 private:
  Worker() : thread_([this] { loop(); }) {}
  ~Worker() {
    {
      std::unique_lock<std::mutex> lock(m_);
      quit_ = true;
      cv_.notify_all();
    }
    thread_.join();
  }
  void loop() {
    std::unique_lock<std::mutex> lock(m_);
    for (;;) {
      cv_.wait(lock, [this] { return have_job_ || quit_; });
      if (quit_) return;
      std::function<void()> job = std::move(job_);
      have_job_ = false;
      lock.unlock();
      job();
      lock.lock();
      done_ = true;
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool have_job_ = false, done_ = true, quit_ = false;
  std::thread thread_;
};
}  // namespace shim_detail

template <typename F0, typename F1> void parallel_invoke(const F0& f0, const F1& f1) {
  shim_detail::Worker& w = shim_detail::Worker::instance();
  w.run([&f0] { f0(); });
  f1();
  w.wait();
}
}  // namespace tbb
