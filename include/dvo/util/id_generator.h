// dvo/util/id_generator.h -- dvo::util::IdGenerator (dvo_core/include/dvo/util/id_generator.h:30-72): "<prefix><n>" ids.
#pragma once

#include <sstream>
#include <string>
#include <vector>

namespace dvo {
namespace util {

class IdGenerator {
 public:
  IdGenerator(const std::string prefix) : prefix_(prefix), next_(0) {}
  const std::vector<std::string>& all() { return issued_; }
  void next(std::string& id) { id = next(); }
  std::string next() {
    std::ostringstream s;
    s << prefix_ << next_++;
    issued_.push_back(s.str());
    return issued_.back();
  }
  void reset() {
    next_ = 0;
    issued_.clear();
  }

 private:
  std::string prefix_;
  std::vector<std::string> issued_;
  int next_;
};

}  // namespace util
}  // namespace dvo
