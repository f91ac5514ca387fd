// Source compatibility with callers of the reference (include/cedar/2d/relax_stencil.h): the 2D data types of this mirror
// live together in <cedar/2d/types.h>.
#ifndef CEDAR_2D_FWD_RELAX_STENCIL_H
#define CEDAR_2D_FWD_RELAX_STENCIL_H
#include <cedar/2d/types.h>
#endif
