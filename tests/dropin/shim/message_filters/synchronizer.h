// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of message_filters that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <functional>
#include <memory>
#include <boost/bind.hpp>    // (the real header pulls it in; dvo_ros/src/camera_base.cpp relies on that)
#include <message_filters/subscriber.h>
namespace message_filters {
// the test delivers "synchronised" message quadruples by calling signal(); a registered callback receives them
template <typename Policy> class Synchronizer {
 public:
  typedef typename Policy::M0 M0; typedef typename Policy::M1 M1; typedef typename Policy::M2 M2; typedef typename Policy::M3 M3;
  typedef std::function<void(const std::shared_ptr<const M0>&, const std::shared_ptr<const M1>&, const std::shared_ptr<const M2>&, const std::shared_ptr<const M3>&)> Callback;
  template <typename S0, typename S1, typename S2, typename S3> Synchronizer(const Policy&, S0&, S1&, S2&, S3&) : live_(false) {}
  template <typename F> Connection registerCallback(const F& f) { callback_ = f; live_ = true; return Connection(&live_); }
  bool connected() const { return live_; }
  void signal(const std::shared_ptr<const M0>& a, const std::shared_ptr<const M1>& b, const std::shared_ptr<const M2>& c, const std::shared_ptr<const M3>& d) {
    if (live_ && callback_) callback_(a, b, c, d);
  }
 private:
  Callback callback_;
  bool live_;
};
}  // namespace message_filters
