/* Forwarding header: the reference spreads its C interface over several headers
 * (include/cedar/interface/c/timer.h among them); here every declaration lives in <cedar/capi.h>. */
#ifndef CEDAR_AMD_FWD_INTERFACE_C_TIMER_H
#define CEDAR_AMD_FWD_INTERFACE_C_TIMER_H
#include <cedar/capi.h>
#endif
