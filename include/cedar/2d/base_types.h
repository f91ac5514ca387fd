/* Forwarding header: the reference spreads its C interface over several headers
 * (include/cedar/2d/base_types.h among them); here every declaration lives in <cedar/capi.h>. */
#ifndef CEDAR_AMD_FWD_2D_BASE_TYPES_H
#define CEDAR_AMD_FWD_2D_BASE_TYPES_H
#include <cedar/capi.h>
#endif
