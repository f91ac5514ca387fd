// Source compatibility with callers of the reference (include/cedar/3d/relax_stencil.h): the 3D data types of this mirror
// live together in <cedar/3d/types.h>.
#ifndef CEDAR_3D_FWD_RELAX_STENCIL_H
#define CEDAR_3D_FWD_RELAX_STENCIL_H
#include <cedar/3d/types.h>
#endif
