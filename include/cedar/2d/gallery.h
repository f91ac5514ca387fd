// Gallery operators (reference src/2d/gallery.cc:7-113, include/cedar/2d/gallery.h): note which ghost-adjacent
// entries stay zero (W only for i >= 2, S only for j >= 2).
#ifndef CEDAR_2D_GALLERY_H
#define CEDAR_2D_GALLERY_H
#include <cedar/2d/types.h>
namespace cedar { namespace cdr2 { namespace gallery {
inline stencil_op<five_pt> diag_diffusion(len_t nx, len_t ny, real_t dx, real_t dy)
{
	stencil_op<five_pt> so(nx, ny);
	real_t hx = 1.0 / (so.len(0) - 1), hy = 1.0 / (so.len(1) - 1);
	real_t xh = hy / hx, yh = hx / hy;
	for (len_t j = 2; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, five_pt::s) = dy * yh;
	for (len_t j = 1; j <= ny; j++) for (len_t i = 2; i <= nx; i++) so(i, j, five_pt::w) = dx * xh;
	for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, five_pt::c) = 2 * dx * xh + 2 * dy * yh;
	return so;
}
inline stencil_op<five_pt> poisson(len_t nx, len_t ny) { return diag_diffusion(nx, ny, 1.0, 1.0); }
inline stencil_op<nine_pt> fe(len_t nx, len_t ny)
{
	stencil_op<nine_pt> so(nx, ny);
	for (len_t j = 2; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, nine_pt::s) = 1.0;
	for (len_t j = 1; j <= ny; j++) for (len_t i = 2; i <= nx; i++) so(i, j, nine_pt::w) = 1.0;
	for (len_t j = 2; j <= ny; j++) for (len_t i = 2; i <= nx; i++) { so(i, j, nine_pt::sw) = 1.0; so(i, j, nine_pt::nw) = 1.0; }
	for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, nine_pt::c) = 8.0;
	return so;
}
}}}
#endif
