// Stand-in for opm/simulators/linalg/bda/BdaSolver.hpp (:32-92); see BdaResult.hpp beside it.
#pragma once
#include <opm/simulators/linalg/bda/BdaResult.hpp>
#include <opm/simulators/linalg/bda/WellContributions.hpp>
