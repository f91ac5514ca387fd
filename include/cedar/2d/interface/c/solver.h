/* Forwarding header: the reference spreads its C interface over several headers
 * (include/cedar/2d/interface/c/solver.h among them); here every declaration lives in <cedar/capi.h>. */
#ifndef CEDAR_AMD_FWD_2D_INTERFACE_C_SOLVER_H
#define CEDAR_AMD_FWD_2D_INTERFACE_C_SOLVER_H
#include <cedar/capi.h>
#endif
