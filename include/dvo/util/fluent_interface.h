// dvo/util/fluent_interface.h -- FI_ATTRIBUTE of dvo_core/include/dvo/util/fluent_interface.h:26-42: a protected member
// `name_` plus a chaining setter and const / mutable getters called `name`.  The reference's dvo_slam headers
// (keyframe.h, constraints/*.h, config.h) declare their attributes with it, so the facade has to ship it.
#pragma once

#define FI_ATTRIBUTE(FI_TYPE, ATTR_TYPE, ATTR_NAME)                      \
 protected:                                                              \
  ATTR_TYPE ATTR_NAME##_;                                                \
                                                                         \
 public:                                                                 \
  FI_TYPE& ATTR_NAME(ATTR_TYPE const& v) { ATTR_NAME##_ = v; return *this; } \
  ATTR_TYPE const& ATTR_NAME() const { return ATTR_NAME##_; }            \
  ATTR_TYPE& ATTR_NAME() { return ATTR_NAME##_; }
