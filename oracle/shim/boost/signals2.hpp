// oracle/shim/boost/signals2.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  A signal = the list of its slots, called in
// connection order; the combiner sees the results through an input-iterator range that invokes lazily, like boost.signals2.
#pragma once
#include <functional>
#include "bind.hpp"
#include <vector>
namespace boost {
using std::function;
namespace signals2 {
struct connection {};
template <typename R> struct optional_last_value { typedef R result_type; template <typename It> R operator()(It f, It l) const { R r = R(); for (; f != l; ++f) r = *f; return r; } };
template <> struct optional_last_value<void> { typedef void result_type; template <typename It> void operator()(It f, It l) const { for (; f != l; ++f) *f; } };
template <typename Sig, typename Combiner = void> class signal;
template <typename R, typename... A, typename Combiner>
class signal<R(A...), Combiner> {
 public:
  typedef std::function<R(A...)> slot_type;
  connection connect(const slot_type& s) { slots_.push_back(s); return connection(); }
  // invokes slot k with the stored arguments when dereferenced
  struct iterator {
    const std::vector<slot_type>* slots;
    size_t k;
    std::function<R(const slot_type&)> call;
    R operator*() const { return call((*slots)[k]); }
    iterator& operator++() { ++k; return *this; }
    bool operator!=(const iterator& o) const { return k != o.k; }
    bool operator==(const iterator& o) const { return k == o.k; }
  };
  template <typename C = Combiner> typename std::enable_if<!std::is_void<C>::value, typename C::result_type>::type operator()(A... a) const {
    std::function<R(const slot_type&)> call = [&](const slot_type& s) { return s(a...); };
    return C()(iterator{&slots_, 0, call}, iterator{&slots_, slots_.size(), call});
  }
  template <typename C = Combiner> typename std::enable_if<std::is_void<C>::value, void>::type operator()(A... a) const {
    for (size_t k = 0; k < slots_.size(); ++k) slots_[k](a...);
  }
 private:
  std::vector<slot_type> slots_;
};
}  // namespace signals2
}  // namespace boost
