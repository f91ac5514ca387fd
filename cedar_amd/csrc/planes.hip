// Plane relaxation, the 3D side (include/cedar/3d/relax_planes.h:36-160, src/3d/relax_planes.cc:25-238): the 2D
// operator of a direction's plane solvers, and -- batched over all planes of one colour, which do not couple --
// the copy of the planes into 2D vectors, their right-hand sides (b minus the couplings to the two neighbouring
// planes times the current iterate, term order of copy_rhs) and the copy back.  The 2D solves themselves are
// V-cycles of the resident 2D solver (solver.cpp planes_*).
//   dir 0 = xy planes (2D index (i,j), plane number k), 1 = xz ((i,k), j), 2 = yz ((j,k), i).
// Stacked 2D arrays: plane ipl = beg + 2q of the colour lives in slot q, slot stride = I2*J2 doubles.
#include "common.h"

namespace cedar_amd {

#define SO(i, j, k, s) so[(size_t)((i)-1) + (size_t)II * ((size_t)((j)-1) + (size_t)JJ * ((size_t)((k)-1) + (size_t)KK * (size_t)(s)))]
#define X(i, j, k) x[(size_t)((i)-1) + (size_t)II * ((size_t)((j)-1) + (size_t)JJ * (size_t)((k)-1))]
#define B(i, j, k) b[(size_t)((i)-1) + (size_t)II * ((size_t)((j)-1) + (size_t)JJ * (size_t)((k)-1))]

// copy_coeff (relax_planes.h:80-160) loops over every plane and overwrites the same 2D operator: what the plane
// solvers are built from is the LAST plane's coefficients (k = nz, j = ny, i = nx), ghosts included.
__global__ __launch_bounds__(256) void plane_operator_kernel(int dir, int nst, const real_t *__restrict__ so,
                                                             real_t *__restrict__ so2, int II, int JJ, int KK)
{
	const int I2 = dir == 2 ? JJ : II, J2 = dir == 0 ? JJ : KK;
	const size_t P2 = (size_t)I2 * J2;
	const bool full = nst == 14;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < P2; t += (size_t)gridDim.x * blockDim.x) {
		const int a = 1 + (int)(t % I2), c = 1 + (int)(t / I2);
		int i, j, k, sw, ss, ssw, snw;
		if (dir == 0) { i = a; j = c; k = KK - 1; sw = KPW; ss = KPS; ssw = KPSW; snw = KPNW; }
		else if (dir == 1) { i = a; j = JJ - 1; k = c; sw = KPW; ss = KB; ssw = KBW; snw = KBE; }
		else { i = II - 1; j = a; k = c; sw = KPS; ss = KB; ssw = KBS; snw = KBN; }
		so2[t] = SO(i, j, k, KP);
		so2[P2 + t] = SO(i, j, k, sw);
		so2[2 * P2 + t] = SO(i, j, k, ss);
		if (full) {
			so2[3 * P2 + t] = SO(i, j, k, ssw);
			so2[4 * P2 + t] = SO(i, j, k, snw);
		}
	}
}

void plane_operator(int dir, int nst, const real_t *so, real_t *so2, int II, int JJ, int KK, hipStream_t st)
{
	const int I2 = dir == 2 ? JJ : II, J2 = dir == 0 ? JJ : KK;
	const size_t P2 = (size_t)I2 * J2;
	hipLaunchKernelGGL(plane_operator_kernel, dim3((unsigned)((P2 + 255) / 256 < 4096 ? (P2 + 255) / 256 : 4096)), dim3(256), 0, st,
	                   dir, nst, so, so2, II, JJ, KK);
}

// copy_rhs, src/3d/relax_planes.cc:25-172, at the interior point (i,j,k) of the plane being relaxed
template <int DIR, bool FULL>
__device__ __forceinline__ real_t plane_rhs_at(const real_t *__restrict__ so, const real_t *__restrict__ x,
                                               const real_t *__restrict__ b, int II, int JJ, int KK, int i, int j, int k)
{
	if (DIR == 0) {
		if (!FULL) return B(i, j, k) + SO(i, j, k, KB) * X(i, j, k - 1) + SO(i, j, k + 1, KB) * X(i, j, k + 1);
		return B(i, j, k)
		       + SO(i, j, k, KB) * X(i, j, k - 1)
		       + SO(i, j, k, KBW) * X(i - 1, j, k - 1)
		       + SO(i, j + 1, k, KBNW) * X(i - 1, j + 1, k - 1)
		       + SO(i, j + 1, k, KBN) * X(i, j + 1, k - 1)
		       + SO(i + 1, j + 1, k, KBNE) * X(i + 1, j + 1, k - 1)
		       + SO(i + 1, j, k, KBE) * X(i + 1, j, k - 1)
		       + SO(i + 1, j, k, KBSE) * X(i + 1, j - 1, k - 1)
		       + SO(i, j, k, KBS) * X(i, j - 1, k - 1)
		       + SO(i, j, k, KBSW) * X(i - 1, j - 1, k - 1)
		       + SO(i, j, k + 1, KBE) * X(i - 1, j, k + 1)
		       + SO(i, j + 1, k + 1, KBSE) * X(i - 1, j + 1, k + 1)
		       + SO(i, j + 1, k + 1, KBS) * X(i, j + 1, k + 1)
		       + SO(i + 1, j + 1, k + 1, KBSW) * X(i + 1, j + 1, k + 1)
		       + SO(i + 1, j, k + 1, KBW) * X(i + 1, j, k + 1)
		       + SO(i, j, k + 1, KB) * X(i, j, k + 1)
		       + SO(i + 1, j, k + 1, KBNW) * X(i + 1, j - 1, k + 1)
		       + SO(i, j, k + 1, KBN) * X(i, j - 1, k + 1)
		       + SO(i, j, k + 1, KBNE) * X(i - 1, j - 1, k + 1);
	} else if (DIR == 1) {
		if (!FULL) return B(i, j, k) + SO(i, j, k, KPS) * X(i, j - 1, k) + SO(i, j + 1, k, KPS) * X(i, j + 1, k);
		return B(i, j, k)
		       + SO(i, j + 1, k, KPNW) * X(i - 1, j + 1, k)
		       + SO(i, j + 1, k, KPS) * X(i, j + 1, k)
		       + SO(i + 1, j + 1, k, KPSW) * X(i + 1, j + 1, k)
		       + SO(i, j + 1, k, KBNW) * X(i - 1, j + 1, k - 1)
		       + SO(i, j + 1, k, KBN) * X(i, j + 1, k - 1)
		       + SO(i + 1, j + 1, k, KBNE) * X(i + 1, j + 1, k - 1)
		       + SO(i, j + 1, k + 1, KBSE) * X(i - 1, j + 1, k + 1)
		       + SO(i, j + 1, k + 1, KBS) * X(i, j + 1, k + 1)
		       + SO(i + 1, j + 1, k + 1, KBSW) * X(i + 1, j + 1, k + 1)
		       + SO(i, j, k, KPSW) * X(i - 1, j - 1, k)
		       + SO(i, j, k, KPS) * X(i, j - 1, k)
		       + SO(i + 1, j, k, KPNW) * X(i + 1, j - 1, k)
		       + SO(i, j, k, KBSW) * X(i - 1, j - 1, k - 1)
		       + SO(i, j, k, KBS) * X(i, j - 1, k - 1)
		       + SO(i + 1, j, k, KBSE) * X(i + 1, j - 1, k - 1)
		       + SO(i, j, k + 1, KBNE) * X(i - 1, j - 1, k + 1)
		       + SO(i, j, k + 1, KBN) * X(i, j - 1, k + 1)
		       + SO(i + 1, j, k + 1, KBNW) * X(i + 1, j - 1, k + 1);
	} else {
		if (!FULL) return B(i, j, k) + SO(i, j, k, KPW) * X(i - 1, j, k) + SO(i + 1, j, k, KPW) * X(i + 1, j, k);
		return B(i, j, k)
		       + SO(i, j + 1, k, KPNW) * X(i - 1, j + 1, k)
		       + SO(i, j, k, KPW) * X(i - 1, j, k)
		       + SO(i, j, k, KPSW) * X(i - 1, j - 1, k)
		       + SO(i, j + 1, k, KBNW) * X(i - 1, j + 1, k - 1)
		       + SO(i, j, k, KBW) * X(i - 1, j, k - 1)
		       + SO(i, j, k, KBSW) * X(i - 1, j - 1, k - 1)
		       + SO(i, j + 1, k + 1, KBSE) * X(i - 1, j + 1, k + 1)
		       + SO(i, j, k + 1, KBE) * X(i - 1, j, k + 1)
		       + SO(i, j, k + 1, KBNE) * X(i - 1, j - 1, k + 1)
		       + SO(i + 1, j + 1, k, KPSW) * X(i + 1, j + 1, k)
		       + SO(i + 1, j, k, KPW) * X(i + 1, j, k)
		       + SO(i + 1, j, k, KPNW) * X(i + 1, j - 1, k)
		       + SO(i + 1, j + 1, k, KBNE) * X(i + 1, j + 1, k - 1)
		       + SO(i + 1, j, k, KBE) * X(i + 1, j, k - 1)
		       + SO(i + 1, j, k, KBSE) * X(i + 1, j - 1, k - 1)
		       + SO(i + 1, j + 1, k + 1, KBSW) * X(i + 1, j + 1, k + 1)
		       + SO(i + 1, j, k + 1, KBW) * X(i + 1, j, k + 1)
		       + SO(i + 1, j, k + 1, KBNW) * X(i + 1, j - 1, k + 1);
	}
}

// blockIdx.y = slot q (plane ipl = beg + 2q, 1-based interior number).  x2 gets the whole plane, ghosts included
// (copy32, relax_planes.cc:176-238); b2 its interior (copy_rhs), the ghosts of b2 stay zero.
template <int DIR, bool FULL>
__global__ __launch_bounds__(256) void plane_gather_kernel(const real_t *__restrict__ so, const real_t *__restrict__ x,
                                                           const real_t *__restrict__ b, real_t *__restrict__ x2s,
                                                           real_t *__restrict__ b2s, int II, int JJ, int KK, int beg)
{
	const int I2 = DIR == 2 ? JJ : II, J2 = DIR == 0 ? JJ : KK;
	const size_t P2 = (size_t)I2 * J2;
	const int n = beg + 2 * (int)blockIdx.y + 1; // 1-based grid index of the plane
	real_t *x2 = x2s + P2 * blockIdx.y, *b2 = b2s + P2 * blockIdx.y;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < P2; t += (size_t)gridDim.x * blockDim.x) {
		const int a = 1 + (int)(t % I2), c = 1 + (int)(t / I2);
		const int i = DIR == 2 ? n : a, j = DIR == 0 ? c : DIR == 1 ? n : a, k = DIR == 0 ? n : c;
		x2[t] = X(i, j, k);
		if (a >= 2 && a <= I2 - 1 && c >= 2 && c <= J2 - 1) b2[t] = plane_rhs_at<DIR, FULL>(so, x, b, II, JJ, KK, i, j, k);
	}
}

template <int DIR>
__global__ __launch_bounds__(256) void plane_scatter_kernel(const real_t *__restrict__ x2s, real_t *__restrict__ x,
                                                            int II, int JJ, int KK, int beg)
{
	const int I2 = DIR == 2 ? JJ : II, J2 = DIR == 0 ? JJ : KK;
	const size_t P2 = (size_t)I2 * J2;
	const int n = beg + 2 * (int)blockIdx.y + 1;
	const real_t *x2 = x2s + P2 * blockIdx.y;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < P2; t += (size_t)gridDim.x * blockDim.x) {
		const int a = 1 + (int)(t % I2), c = 1 + (int)(t / I2);
		const int i = DIR == 2 ? n : a, j = DIR == 0 ? c : DIR == 1 ? n : a, k = DIR == 0 ? n : c;
		X(i, j, k) = x2[t];
	}
}
#undef SO
#undef X
#undef B

static dim3 plane_grid(size_t P2, int nslots)
{
	size_t gx = (P2 + 255) / 256;
	if (gx > 1024) gx = 1024;
	return dim3((unsigned)gx, (unsigned)nslots);
}

// all planes beg, beg+2, .. (nslots of them) of direction dir: x -> x2s, right-hand sides -> b2s
void plane_gather(int dir, int nst, const real_t *so, const real_t *x, const real_t *b, real_t *x2s, real_t *b2s,
                  int II, int JJ, int KK, int beg, int nslots, hipStream_t st)
{
	if (nslots <= 0) return;
	const int I2 = dir == 2 ? JJ : II, J2 = dir == 0 ? JJ : KK;
	const dim3 g = plane_grid((size_t)I2 * J2, nslots);
	const bool full = nst == 14;
#define GO(D, F) hipLaunchKernelGGL((plane_gather_kernel<D, F>), g, dim3(256), 0, st, so, x, b, x2s, b2s, II, JJ, KK, beg)
	if (dir == 0) { if (full) GO(0, true); else GO(0, false); }
	else if (dir == 1) { if (full) GO(1, true); else GO(1, false); }
	else { if (full) GO(2, true); else GO(2, false); }
#undef GO
}

void plane_scatter(int dir, const real_t *x2s, real_t *x, int II, int JJ, int KK, int beg, int nslots, hipStream_t st)
{
	if (nslots <= 0) return;
	const int I2 = dir == 2 ? JJ : II, J2 = dir == 0 ? JJ : KK;
	const dim3 g = plane_grid((size_t)I2 * J2, nslots);
	if (dir == 0) hipLaunchKernelGGL(plane_scatter_kernel<0>, g, dim3(256), 0, st, x2s, x, II, JJ, KK, beg);
	else if (dir == 1) hipLaunchKernelGGL(plane_scatter_kernel<1>, g, dim3(256), 0, st, x2s, x, II, JJ, KK, beg);
	else hipLaunchKernelGGL(plane_scatter_kernel<2>, g, dim3(256), 0, st, x2s, x, II, JJ, KK, beg);
}

} // namespace cedar_amd
