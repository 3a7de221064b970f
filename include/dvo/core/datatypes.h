// dvo/core/datatypes.h -- typedefs of dvo_core/include/dvo/core/datatypes.h:32-52 for the MI355X engine's facade.
#pragma once

#include <limits>

#include "../compat.h"

namespace dvo {
namespace core {

typedef float IntensityType;
static const IntensityType Invalid = std::numeric_limits<IntensityType>::quiet_NaN();
typedef float DepthType;
static const DepthType InvalidDepth = std::numeric_limits<DepthType>::quiet_NaN();
typedef float NumType;

typedef dvo::compat::Affine3d AffineTransformd;
typedef dvo::compat::Vector6d Vector6d;
typedef dvo::compat::Matrix6d Matrix6d;

}  // namespace core
}  // namespace dvo
