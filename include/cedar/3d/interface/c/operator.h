/* Forwarding header: the reference spreads its C interface over several headers
 * (include/cedar/3d/interface/c/operator.h among them); here every declaration lives in <cedar/capi.h>. */
#ifndef CEDAR_AMD_FWD_3D_INTERFACE_C_OPERATOR_H
#define CEDAR_AMD_FWD_3D_INTERFACE_C_OPERATOR_H
#include <cedar/capi.h>
#endif
