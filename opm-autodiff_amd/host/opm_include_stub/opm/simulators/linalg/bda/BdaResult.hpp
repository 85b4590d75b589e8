// Stand-in for the reference's opm/simulators/linalg/bda/BdaResult.hpp (:28-40), written here (NOT a copy): it exists so
// that host/Makefile can compile hipSolverBackend.hpp with OPMHIP_USE_OPM_HEADERS - the branch a maintainer of
// opm-simulators uses - under the include paths the reference uses.  Inside an opm-simulators tree the real header is found
// first and this directory is not on the include path.
#pragma once
#include "../../../../../BdaCompat.hpp"
