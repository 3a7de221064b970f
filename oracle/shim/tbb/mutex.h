// oracle/shim/tbb/mutex.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.
#pragma once
#include <mutex>
namespace tbb {
class mutex {
 public:
  void lock() { m_.lock(); }
  void unlock() { m_.unlock(); }
  bool try_lock() { return m_.try_lock(); }
  class scoped_lock {
   public:
    scoped_lock() : m_(0) {}
    explicit scoped_lock(mutex& m) : m_(&m) { m.lock(); }
    ~scoped_lock() { if (m_) m_->unlock(); }
    void acquire(mutex& m) { m.lock(); m_ = &m; }
    bool try_acquire(mutex& m) { if (!m.try_lock()) return false; m_ = &m; return true; }
    void release() { if (m_) { m_->unlock(); m_ = 0; } }
   private:
    scoped_lock(const scoped_lock&);
    scoped_lock& operator=(const scoped_lock&);
    mutex* m_;
  };
 private:
  std::mutex m_;
};
}  // namespace tbb
