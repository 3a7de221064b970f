#pragma once
#include "edge_se3.h"
