// Stand-in for opm/simulators/linalg/bda/WellContributions.hpp (:60-214) PLUS the accessor INTEGRATION.md's patch adds to
// the reference class (getHostArrays: the host-side vectors the OpenCL path fills, WellContributions.cpp:42-44,
// 215-225); see BdaResult.hpp beside it.  BdaCompat.hpp's class carries the accessor when OPMHIP_USE_OPM_HEADERS is set.
#pragma once
#include "../../../../../BdaCompat.hpp"
