// Device-side generators of the reference's gallery operators and example
// right-hand sides, so that benchmark-size problems (512^3 27-point = 15 GB of
// operator) are built in HBM instead of crossing PCIe.
// Restate src/2d/gallery.cc:7-113, src/3d/gallery.cc:7-190 (which entries are
// set: W only for i >= 2, S only for j >= 2, ... in the 0-based-with-ghost
// index) and examples/basic-{2d,3d}-ser/poisson.cc set_problem().
#include "common.h"

namespace cedar_amd {

// which: 0 poisson2 (dx=dy=1), 1 diag_diffusion2, 2 fe2
__global__ void gallery2_kernel(int which, real_t *__restrict__ so, real_t *__restrict__ b, int nx, int ny, double dx, double dy)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1; // 0-based incl. ghost, interior
	if (i > nx) return;
	const int II = nx + 2, JJ = ny + 2;
	const size_t PS = (size_t)II * JJ, x = (size_t)i + (size_t)II * j;
	const double hx = 1.0 / (II - 1), hy = 1.0 / (JJ - 1);
	if (which == 2) {
		if (j >= 2) so[KS * PS + x] = 1.0;
		if (i >= 2) so[KW * PS + x] = 1.0;
		if (i >= 2 && j >= 2) { so[KSW * PS + x] = 1.0; so[KNW * PS + x] = 1.0; }
		so[KO * PS + x] = 8.0;
	} else {
		const double xh = hy / hx, yh = hx / hy;
		if (j >= 2) so[KS * PS + x] = dy * yh;
		if (i >= 2) so[KW * PS + x] = dx * xh;
		so[KO * PS + x] = 2 * dx * xh + 2 * dy * yh;
	}
	if (b) {
		const double pi = 3.14159265358979323846;
		const double xx = i * hx, yy = j * hy;
		b[x] = 8 * (pi * pi) * sin(2 * pi * xx) * sin(2 * pi * yy) * (hx * hy);
	}
}

// which: 10 poisson3, 11 diag_diffusion3, 12 fe3
// (gi0,gj0,gk0): offset of this subdomain inside the global grid, (gnx,gny,gnz): global extents;
// a single-domain run has offsets 0 and global = local extents.
__global__ void gallery3_kernel(int which, real_t *__restrict__ so, real_t *__restrict__ b, int nx, int ny, int nz,
                                double dx, double dy, double dz, int gi0, int gj0, int gk0, int gnx, int gny, int gnz)
{
	const int li = blockIdx.x * blockDim.x + threadIdx.x + 1, lj = blockIdx.y + 1, lk = blockIdx.z + 1;
	if (li > nx) return;
	const int II = nx + 2, JJ = ny + 2, KK = nz + 2;
	const size_t sk = (size_t)II * JJ, PS = sk * KK, x = (size_t)li + (size_t)II * lj + sk * lk;
	const int i = li + gi0, j = lj + gj0, k = lk + gk0; // global 0-based-with-ghost index
	const double hx = 1.0 / (gnx + 1), hy = 1.0 / (gny + 1), hz = 1.0 / (gnz + 1);
	if (which == 12) {
		if (i >= 2) so[KPW * PS + x] = 1.0;
		if (j >= 2) so[KPS * PS + x] = 1.0;
		if (k >= 2) so[KB * PS + x] = 1.0;
		if (i >= 2 && j >= 2) { so[KPNW * PS + x] = 1.0; so[KPSW * PS + x] = 1.0; }
		if (i >= 2 && k >= 2) { so[KBW * PS + x] = 1.0; so[KBE * PS + x] = 1.0; }
		if (j >= 2 && k >= 2) { so[KBN * PS + x] = 1.0; so[KBS * PS + x] = 1.0; }
		if (i >= 2 && j >= 2 && k >= 2) {
			so[KBNW * PS + x] = 1.0; so[KBNE * PS + x] = 1.0; so[KBSE * PS + x] = 1.0; so[KBSW * PS + x] = 1.0;
		}
		so[KP * PS + x] = 26;
	} else {
		const double xh = hy * hz / hx, yh = hx * hz / hy, zh = hx * hy / hz;
		if (j >= 2) so[KPS * PS + x] = dy * yh;
		if (i >= 2) so[KPW * PS + x] = dx * xh;
		if (k >= 2) so[KB * PS + x] = dz * zh;
		so[KP * PS + x] = 2.0 * dx * xh + 2.0 * dy * yh + 2.0 * dz * zh;
	}
	if (b) {
		const double pi = 3.14159265358979323846;
		const double xx = i * hx, yy = j * hy, zz = k * hz;
		b[x] = 12 * (pi * pi) * sin(2 * pi * xx) * sin(2 * pi * yy) * sin(2 * pi * zz) * (hx * hy * hz);
	}
}

// params: diag_diffusion: (dx,dy[,dz]).  `which` + 100 (e.g. 112 = fe3) places the subdomain
// inside a global grid: (gi0,gj0,gk0,gnx,gny,gnz) follow the operator's own parameters
// (3 for diag_diffusion3, else 0).
void gallery_fill(int which, real_t *so, real_t *b, int nx, int ny, int nz, const double *params, hipStream_t st)
{
	double p0 = 1.0, p1 = 1.0, p2 = 1.0;
	int off[6] = { 0, 0, 0, nx, ny, nz };
	if (which >= 100) {
		which -= 100;
		const double *pp = params + (which == 11 ? 3 : 0);
		for (int t = 0; t < 6; t++) off[t] = (int)pp[t];
	}
	if ((which == 1 || which == 11) && params) { p0 = params[0]; p1 = params[1]; if (which == 11) p2 = params[2]; }
	if (which < 10) {
		dim3 grid((nx + 255) / 256, ny);
		hipLaunchKernelGGL(gallery2_kernel, grid, dim3(256), 0, st, which, so, b, nx, ny, p0, p1);
	} else {
		dim3 grid((nx + 127) / 128, ny, nz);
		hipLaunchKernelGGL(gallery3_kernel, grid, dim3(128), 0, st, which, so, b, nx, ny, nz, p0, p1, p2,
		                   off[0], off[1], off[2], off[3], off[4], off[5]);
	}
}

} // namespace cedar_amd
