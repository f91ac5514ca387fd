/* Stencil directions of the 3D operators (reference include/cedar/3d/base_types.h:5-20; Fortran slots
 * src/2d/ftn/BMG_stencils_f90.h:43-64: kp, kpw, kps, kb, kpsw, kpnw, kbw, kbnw, kbn, kbne, kbe, kbse, kbs, kbsw):
 * the C enumerators come with <cedar/capi.h>, the C++ stencil tags below. */
#ifndef CEDAR_3D_BASE_TYPES_H
#define CEDAR_3D_BASE_TYPES_H
#include <cedar/capi.h>
#ifdef __cplusplus
namespace cedar { namespace cdr3 {
enum class seven_pt { p = 0, pw = 1, ps = 2, b = 3, ndirs = 4 };
enum class xxvii_pt { p = 0, pw, ps, b, psw, pnw, bw, bnw, bn, bne, be, bse, bs, bsw, ndirs };
template <class sten> struct stencil_ndirs { static const int value = static_cast<int>(sten::ndirs); };
}}
#endif
#endif
