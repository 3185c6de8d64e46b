// rocalution/rocalution.hpp -- umbrella header (cf. src/rocalution.hpp:31-86 of the reference).
// Subset of the rocALUTION API that forms the preconditioned-Krylov hot path, implemented on the
// MI355X-native backend (librocalution_amd.so).  Existing CG / GMRES / BiCGStab drivers that stay
// inside this subset recompile unchanged:
//     g++ -Iinclude driver.cpp -Lrocalution_amd -lrocalution_amd
#pragma once

#include "base.hpp"
#include "io.hpp"
#include "solvers.hpp"
#include "global.hpp"
#include "distribute.hpp"
