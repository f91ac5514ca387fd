// Residual r = b - A x for 5/9-point (2D) and 7/27-point (3D) symmetric stencils,
// plus the deterministic interior sum of squares used for ||r||_2.
// Replaces BMG2_SymStd_residual (src/2d/ftn/BMG2_SymStd_residual.f90:85-119) and
// BMG3_SymStd_residual (src/3d/ftn/BMG3_SymStd_residual.f90:67-121); the norm
// replaces grid_func::lp_norm<2> (include/cedar/2d/grid_func.h:42-53).
//
// Streaming kernels: one lane per grid point, unit stride along i, every
// operand plane read once from HBM (neighbour re-reads are L1/L2 hits):
// 136 algorithmic B/DOF (27-pt), 64 B/DOF (9-pt).  Term order = reference,
// -ffp-contract=off => bit-identical to the reference CPU build.
#include "common.h"

namespace cedar_amd {

// MV = true turns the kernel into the operator application qf = A q of
// BMG2_SymStd_UTILS_matvec (src/2d/ftn/mpi/BMG2_SymStd_UTILS_matvec.f90:84-118): diagonal term first,
// then the same neighbour sequence subtracted; `res` receives A q and `qf` is not read.
template <bool NINE, bool MV = false>
__global__ __launch_bounds__(256) void residual2_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                         const real_t *__restrict__ q, real_t *__restrict__ res,
                                                         int II, int JJ, size_t bstride)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x + 1; // 0-based incl. ghost
	const int j = blockIdx.y + 1;
	if (i > II - 2) return;
	if (!MV) qf += bstride * blockIdx.z; // batch item (common.h Batch)
	q += bstride * blockIdx.z; res += bstride * blockIdx.z;
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t x = (size_t)i + sj * (size_t)j;
	if (MV) {
		real_t s = so[KO * PS + x] * q[x];
		s = s - so[KW * PS + x] * q[x - 1];
		s = s - so[KW * PS + x + 1] * q[x + 1];
		s = s - so[KS * PS + x] * q[x - sj];
		s = s - so[KS * PS + x + sj] * q[x + sj];
		if (NINE) {
			s = s - so[KSW * PS + x] * q[x - 1 - sj];
			s = s - so[KNW * PS + x + 1] * q[x + 1 - sj];
			s = s - so[KNW * PS + x + sj] * q[x - 1 + sj];
			s = s - so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
		}
		res[x] = s;
		return;
	}
	real_t s = qf[x];
	s = s + so[KW * PS + x] * q[x - 1];
	s = s + so[KW * PS + x + 1] * q[x + 1];
	s = s + so[KS * PS + x] * q[x - sj];
	s = s + so[KS * PS + x + sj] * q[x + sj];
	if (NINE) {
		s = s + so[KSW * PS + x] * q[x - 1 - sj];
		s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
		s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
		s = s + so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
	}
	s = s - so[KO * PS + x] * q[x];
	res[x] = s;
}

void matvec2(const real_t *so, const real_t *q, real_t *qf, int II, int JJ, int nstncl, hipStream_t st)
{
	if (II < 3 || JJ < 3) return;
	dim3 grid((II - 2 + 255) / 256, JJ - 2);
	if (nstncl == 5)
		hipLaunchKernelGGL((residual2_kernel<true, true>), grid, dim3(256), 0, st, so, (const real_t *)nullptr, q, qf, II, JJ, (size_t)0);
	else
		hipLaunchKernelGGL((residual2_kernel<false, true>), grid, dim3(256), 0, st, so, (const real_t *)nullptr, q, qf, II, JJ, (size_t)0);
}

void residual9_fast(const real_t *so, const real_t *qf, const real_t *q, real_t *res, int II, int JJ, hipStream_t st, Batch bt);

void residual2(const real_t *so, const real_t *qf, const real_t *q, real_t *res,
               int II, int JJ, int nstncl, hipStream_t st, Batch bt)
{
	if (II < 3 || JJ < 3) return;
	if (nstncl == 5) { // pair-per-lane row kernel (relax2d.hip)
		residual9_fast(so, qf, q, res, II, JJ, st, bt);
		return;
	}
	dim3 grid((II - 2 + 255) / 256, JJ - 2, bt.n);
	if (nstncl == 5)
		hipLaunchKernelGGL(residual2_kernel<true>, grid, dim3(256), 0, st, so, qf, q, res, II, JJ, bt.stride);
	else
		hipLaunchKernelGGL(residual2_kernel<false>, grid, dim3(256), 0, st, so, qf, q, res, II, JJ, bt.stride);
}

// MV: qf = A q of BMG3_SymStd_UTILS_matvec (src/3d/ftn/mpi/BMG3_SymStd_UTILS_matvec.f90:80-127)
template <bool XXVII, bool MV = false>
__global__ __launch_bounds__(256) void residual3_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                         const real_t *__restrict__ q, real_t *__restrict__ res,
                                                         int II, int JJ, int KK, unsigned nrows)
{
	// one workgroup row-segment; rows dealt to XCDs in contiguous k-slabs
	const unsigned L = xcd_remap(blockIdx.x, nrows);
	if (L >= nrows) return;
	const int j = (int)(L % (unsigned)(JJ - 2)) + 1, k = (int)(L / (unsigned)(JJ - 2)) + 1;
	const size_t sj = II, sk = (size_t)II * JJ, PS = sk * KK;
	for (int i = threadIdx.x + 1; i <= II - 2; i += blockDim.x) {
		const size_t x = (size_t)i + sj * (size_t)j + sk * (size_t)k;
		if (MV) {
			real_t s = so[KP * PS + x] * q[x];
			if (XXVII) {
				s = s - so[KPW * PS + x] * q[x - 1];
				s = s - so[KPNW * PS + x + sj] * q[x - 1 + sj];
				s = s - so[KPS * PS + x + sj] * q[x + sj];
				s = s - so[KPSW * PS + x + 1 + sj] * q[x + 1 + sj];
				s = s - so[KPW * PS + x + 1] * q[x + 1];
				s = s - so[KPNW * PS + x + 1] * q[x + 1 - sj];
				s = s - so[KPS * PS + x] * q[x - sj];
				s = s - so[KPSW * PS + x] * q[x - 1 - sj];
				s = s - so[KB * PS + x] * q[x - sk];
				s = s - so[KBW * PS + x] * q[x - 1 - sk];
				s = s - so[KBNW * PS + x + sj] * q[x - 1 + sj - sk];
				s = s - so[KBN * PS + x + sj] * q[x + sj - sk];
				s = s - so[KBNE * PS + x + 1 + sj] * q[x + 1 + sj - sk];
				s = s - so[KBE * PS + x + 1] * q[x + 1 - sk];
				s = s - so[KBSE * PS + x + 1] * q[x + 1 - sj - sk];
				s = s - so[KBS * PS + x] * q[x - sj - sk];
				s = s - so[KBSW * PS + x] * q[x - 1 - sj - sk];
				s = s - so[KB * PS + x + sk] * q[x + sk];
				s = s - so[KBE * PS + x + sk] * q[x - 1 + sk];
				s = s - so[KBSE * PS + x + sj + sk] * q[x - 1 + sj + sk];
				s = s - so[KBS * PS + x + sj + sk] * q[x + sj + sk];
				s = s - so[KBSW * PS + x + 1 + sj + sk] * q[x + 1 + sj + sk];
				s = s - so[KBW * PS + x + 1 + sk] * q[x + 1 + sk];
				s = s - so[KBNW * PS + x + 1 + sk] * q[x + 1 - sj + sk];
				s = s - so[KBN * PS + x + sk] * q[x - sj + sk];
				s = s - so[KBNE * PS + x + sk] * q[x - 1 - sj + sk];
			} else {
				s = s - so[KPW * PS + x] * q[x - 1];
				s = s - so[KPS * PS + x + sj] * q[x + sj];
				s = s - so[KPW * PS + x + 1] * q[x + 1];
				s = s - so[KPS * PS + x] * q[x - sj];
				s = s - so[KB * PS + x] * q[x - sk];
				s = s - so[KB * PS + x + sk] * q[x + sk];
			}
			res[x] = s;
			continue;
		}
		real_t s = qf[x];
		if (XXVII) {
			s = s + so[KPW * PS + x] * q[x - 1];
			s = s + so[KPNW * PS + x + sj] * q[x - 1 + sj];
			s = s + so[KPS * PS + x + sj] * q[x + sj];
			s = s + so[KPSW * PS + x + 1 + sj] * q[x + 1 + sj];
			s = s + so[KPW * PS + x + 1] * q[x + 1];
			s = s + so[KPNW * PS + x + 1] * q[x + 1 - sj];
			s = s + so[KPS * PS + x] * q[x - sj];
			s = s + so[KPSW * PS + x] * q[x - 1 - sj];
			s = s + so[KB * PS + x] * q[x - sk];
			s = s + so[KBW * PS + x] * q[x - 1 - sk];
			s = s + so[KBNW * PS + x + sj] * q[x - 1 + sj - sk];
			s = s + so[KBN * PS + x + sj] * q[x + sj - sk];
			s = s + so[KBNE * PS + x + 1 + sj] * q[x + 1 + sj - sk];
			s = s + so[KBE * PS + x + 1] * q[x + 1 - sk];
			s = s + so[KBSE * PS + x + 1] * q[x + 1 - sj - sk];
			s = s + so[KBS * PS + x] * q[x - sj - sk];
			s = s + so[KBSW * PS + x] * q[x - 1 - sj - sk];
			s = s + so[KB * PS + x + sk] * q[x + sk];
			s = s + so[KBE * PS + x + sk] * q[x - 1 + sk];
			s = s + so[KBSE * PS + x + sj + sk] * q[x - 1 + sj + sk];
			s = s + so[KBS * PS + x + sj + sk] * q[x + sj + sk];
			s = s + so[KBSW * PS + x + 1 + sj + sk] * q[x + 1 + sj + sk];
			s = s + so[KBW * PS + x + 1 + sk] * q[x + 1 + sk];
			s = s + so[KBNW * PS + x + 1 + sk] * q[x + 1 - sj + sk];
			s = s + so[KBN * PS + x + sk] * q[x - sj + sk];
			s = s + so[KBNE * PS + x + sk] * q[x - 1 - sj + sk];
		} else {
			s = s + so[KPW * PS + x] * q[x - 1];
			s = s + so[KPS * PS + x + sj] * q[x + sj];
			s = s + so[KPW * PS + x + 1] * q[x + 1];
			s = s + so[KPS * PS + x] * q[x - sj];
			s = s + so[KB * PS + x] * q[x - sk];
			s = s + so[KB * PS + x + sk] * q[x + sk];
		}
		s = s - so[KP * PS + x] * q[x];
		res[x] = s;
	}
}

void residual27_fast(const real_t *so, const real_t *qf, const real_t *q, real_t *res, int II, int JJ, int KK, hipStream_t st);

void residual3(const real_t *so, const real_t *qf, const real_t *q, real_t *res,
               int II, int JJ, int KK, int nstncl, hipStream_t st)
{
	if (II < 3 || JJ < 3 || KK < 3) return;
	if (nstncl == 14) { // pair-per-lane row kernel (relax3d.hip)
		residual27_fast(so, qf, q, res, II, JJ, KK, st);
		return;
	}
	unsigned nrows = (unsigned)(JJ - 2) * (unsigned)(KK - 2);
	int bs = II - 2 >= 256 ? 256 : (II - 2 > 64 ? 128 : 64);
	if (nstncl == 14)
		hipLaunchKernelGGL(residual3_kernel<true>, dim3(xcd_grid(nrows)), dim3(bs), 0, st, so, qf, q, res, II, JJ, KK, nrows);
	else
		hipLaunchKernelGGL(residual3_kernel<false>, dim3(xcd_grid(nrows)), dim3(bs), 0, st, so, qf, q, res, II, JJ, KK, nrows);
}

void matvec3(const real_t *so, const real_t *q, real_t *qf, int II, int JJ, int KK, int nstncl, hipStream_t st)
{
	if (II < 3 || JJ < 3 || KK < 3) return;
	unsigned nrows = (unsigned)(JJ - 2) * (unsigned)(KK - 2);
	int bs = II - 2 >= 256 ? 256 : (II - 2 > 64 ? 128 : 64);
	if (nstncl == 14)
		hipLaunchKernelGGL((residual3_kernel<true, true>), dim3(xcd_grid(nrows)), dim3(bs), 0, st, so, (const real_t *)nullptr, q, qf, II, JJ, KK, nrows);
	else
		hipLaunchKernelGGL((residual3_kernel<false, true>), dim3(xcd_grid(nrows)), dim3(bs), 0, st, so, (const real_t *)nullptr, q, qf, II, JJ, KK, nrows);
}

// ---------------------------------------------------------------- sum of squares
// Stage 1: NB fixed workgroups, each a fixed strided slice of the rows, wave
// shuffle + LDS tree.  Stage 2: one workgroup reduces the NB partials.  No
// atomics: the result is reproducible run to run.  (The reference sums
// sequentially; the two agree to ~sqrt(N) eps.)
static constexpr int SUMSQ_NB = 2048;

__device__ __forceinline__ real_t block_sum(real_t v, real_t *lds)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
	if (l == 0) lds[w] = v;
	__syncthreads();
	real_t r = 0.0;
	if (threadIdx.x == 0) {
		const int nw = (blockDim.x + 63) >> 6;
		for (int t = 0; t < nw; t++) r += lds[t];
	}
	return r;
}

__global__ __launch_bounds__(256) void sumsq_stage1(const real_t *__restrict__ v, int II, int JJ, int KK,
                                                     real_t *__restrict__ part)
{
	__shared__ real_t lds[4];
	const int nj = JJ - 2, nk = KK == 1 ? 1 : KK - 2;
	const size_t nrows = (size_t)nj * nk;
	real_t acc = 0.0;
	for (size_t r = blockIdx.x; r < nrows; r += gridDim.x) {
		const size_t j = r % nj + 1, k = KK == 1 ? 0 : r / nj + 1;
		const real_t *row = v + (size_t)II * (j + (size_t)JJ * k);
		for (int i = threadIdx.x + 1; i <= II - 2; i += blockDim.x) {
			real_t t = row[i];
			acc += t * t;
		}
	}
	real_t s = block_sum(acc, lds);
	if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sumsq_stage2(const real_t *__restrict__ part, int n, real_t *__restrict__ out)
{
	__shared__ real_t lds[4];
	real_t acc = 0.0;
	for (int i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
	real_t s = block_sum(acc, lds);
	if (threadIdx.x == 0) *out = s;
}

void sumsq_interior(const real_t *v, int II, int JJ, int KK, real_t *scratch, real_t *out, hipStream_t st)
{
	hipLaunchKernelGGL(sumsq_stage1, dim3(SUMSQ_NB), dim3(256), 0, st, v, II, JJ, KK, scratch);
	hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, scratch, SUMSQ_NB, out);
}

} // namespace cedar_amd
