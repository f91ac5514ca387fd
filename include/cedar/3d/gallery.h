// Gallery operators (reference src/3d/gallery.cc:7-190, include/cedar/3d/gallery.h).
#ifndef CEDAR_3D_GALLERY_H
#define CEDAR_3D_GALLERY_H
#include <cedar/3d/types.h>
namespace cedar { namespace cdr3 {
namespace gallery {
inline stencil_op<seven_pt> diag_diffusion(len_t nx, len_t ny, len_t nz, real_t dx, real_t dy, real_t dz)
{
	stencil_op<seven_pt> so(nx, ny, nz);
	real_t hx = 1.0 / (so.len(0) - 1), hy = 1.0 / (so.len(1) - 1), hz = 1.0 / (so.len(2) - 1);
	real_t xh = hy * hz / hx, yh = hx * hz / hy, zh = hx * hy / hz;
	for (len_t k = 1; k <= nz; k++) for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) {
		if (j >= 2) so(i, j, k, seven_pt::ps) = dy * yh;
		if (i >= 2) so(i, j, k, seven_pt::pw) = dx * xh;
		if (k >= 2) so(i, j, k, seven_pt::b) = dz * zh;
		so(i, j, k, seven_pt::p) = 2.0 * dx * xh + 2.0 * dy * yh + 2.0 * dz * zh;
	}
	return so;
}
inline stencil_op<seven_pt> poisson(len_t nx, len_t ny, len_t nz) { return diag_diffusion(nx, ny, nz, 1.0, 1.0, 1.0); }
inline stencil_op<xxvii_pt> fe(len_t nx, len_t ny, len_t nz)
{
	stencil_op<xxvii_pt> so(nx, ny, nz);
	using X = xxvii_pt;
	for (len_t k = 1; k <= nz; k++) for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) {
		const bool I = i >= 2, J = j >= 2, K = k >= 2;
		if (I) so(i, j, k, X::pw) = 1.0;
		if (J) so(i, j, k, X::ps) = 1.0;
		if (K) so(i, j, k, X::b) = 1.0;
		if (I && J) { so(i, j, k, X::pnw) = 1.0; so(i, j, k, X::psw) = 1.0; }
		if (I && K) { so(i, j, k, X::bw) = 1.0; so(i, j, k, X::be) = 1.0; }
		if (J && K) { so(i, j, k, X::bn) = 1.0; so(i, j, k, X::bs) = 1.0; }
		if (I && J && K) { so(i, j, k, X::bnw) = 1.0; so(i, j, k, X::bne) = 1.0; so(i, j, k, X::bse) = 1.0; so(i, j, k, X::bsw) = 1.0; }
		so(i, j, k, X::p) = 26;
	}
	return so;
}
}
}}
#endif
