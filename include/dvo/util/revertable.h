// dvo/util/revertable.h -- dvo::util::Revertable<T> (dvo_core/include/dvo/util/revertable.h:29-58): a value with one level
// of undo.  (On the device the same semantic lives in PairState::{initial,estimate}_old, solver_logic.h.)
#pragma once

namespace dvo {
namespace util {

template <typename T>
class Revertable {
 public:
  Revertable() : old_(), value_() {}
  Revertable(const T& value) : old_(), value_(value) {}
  inline const T& operator()() const { return value_; }
  T& update() {
    old_ = value_;
    return value_;
  }
  void revert() { value_ = old_; }

 private:
  T old_, value_;
};

}  // namespace util
}  // namespace dvo
