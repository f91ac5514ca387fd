/* Stencil directions of the 2D operators (reference include/cedar/2d/base_types.h:4-14; Fortran slots
 * src/2d/ftn/BMG_stencils_f90.h:29-48: ko, kw, ks, ksw, knw): the C enumerators come with <cedar/capi.h>,
 * the C++ stencil tags below.  Positive off-diagonals, symmetric half stencil. */
#ifndef CEDAR_2D_BASE_TYPES_H
#define CEDAR_2D_BASE_TYPES_H
#include <cedar/capi.h>
#ifdef __cplusplus
namespace cedar { namespace cdr2 {
enum class five_pt { c = 0, w = 1, s = 2, ndirs = 3 };
enum class nine_pt { c = 0, w = 1, s = 2, sw = 3, nw = 4, ndirs = 5 };
template <class sten> struct stencil_ndirs { static const int value = static_cast<int>(sten::ndirs); };
}}
#endif
#endif
