// Coarsest-grid direct solve: band Cholesky set-up and solve.
// Replaces BMG2_SymStd_SETUP_cg_LU (src/2d/ftn/BMG2_SymStd_SETUP_cg_LU.f90:92-118, :236-258),
// BMG2_SymStd_SOLVE_cg (src/2d/ftn/BMG2_SymStd_SOLVE_cg.f90:95-125),
// BMG3_SymStd_SETUP_cg_LU (src/3d/ftn/BMG3_SymStd_SETUP_cg_LU.f90:111-198) and
// BMG3_SymStd_SOLVE_cg (src/3d/ftn/BMG3_SymStd_SOLVE_cg.f90:100-150), including the
// LAPACK they call (DPBTRF/DPBTRS, UPLO='U'; system LAPACK in the reference, not
// vendored).  The coarsest grid holds a few dozen unknowns (16 in 2D, 64 in 3D
// for every configuration here), far below DPBTRF's blocking threshold, so the
// published unblocked DPBTF2 / DTBSV recurrences are restated.  The factorisation
// (set-up, once) runs on one lane in sequential order; the solve (every cycle) runs
// on one wavefront out of LDS (~10 us), on the device: no PCIe round trip inside
// the cycle.  The band is packed by the whole workgroup.
#include "common.h"

namespace cedar_amd {

#define ABD(r, c) abd[(size_t)((r)-1) + (size_t)nabd1 * (size_t)((c)-1)]

// DPBTF2 'U' on AB(ldab, n), bandwidth kd; returns LAPACK INFO.  Run by a whole workgroup (every thread of the block must
// call it): a column step is a pivot, a scaling and the rank-one update A := A - x x^T of the trailing band block, whose
// entries are independent -- each entry receives exactly the operation the sequential loops of DPBTF2 apply to it
// (a + x_r * (-x_c), two roundings), in the same order over the column steps, so the factor is bit-identical to the
// sequential routine; the lanes only share the entries of one step.  (One thread running those loops -- rounds 1-3 -- spends
// 160 s on a 4096 x 1024 band: the coarsest level of an elongated or gathered grid.)
__device__ int dpbtf2_upper_wg(int n, int kd, real_t *ab, int ldab, int *flag /* LDS */)
{
	const int kld = ldab - 1 > 1 ? ldab - 1 : 1;
	if (threadIdx.x == 0) *flag = 0;
	__syncthreads();
	for (int j = 1; j <= n; j++) {
		real_t *dj = ab + (size_t)kd + (size_t)ldab * (size_t)(j - 1); // AB(kd+1, j)
		if (threadIdx.x == 0) {
			const real_t ajj = *dj;
			if (ajj <= 0.0) *flag = j;
			else *dj = sqrt(ajj);
		}
		__syncthreads();
		if (*flag) return *flag;
		const int kn = kd < n - j ? kd : n - j;
		if (kn > 0) {
			const real_t r = 1.0 / *dj;
			real_t *x = ab + (size_t)(kd - 1) + (size_t)ldab * (size_t)j; // AB(kd, j+1), stride kld
			real_t *a = ab + (size_t)kd + (size_t)ldab * (size_t)j;       // AB(kd+1, j+1), lda kld
			for (int t = threadIdx.x; t < kn; t += blockDim.x) x[(size_t)t * kld] = r * x[(size_t)t * kld];
			__syncthreads();
			for (int c = 0; c < kn; c++) {
				const real_t xc = x[(size_t)c * kld];
				if (xc != 0.0) {
					const real_t temp = -1.0 * xc;
					// (x(rr) of this step lies on the super-diagonal line through AB(kd, j+1): never an entry a(rr', c') with rr' <= c')
					for (int rr = threadIdx.x; rr <= c; rr += blockDim.x) a[(size_t)rr + (size_t)c * kld] += x[(size_t)rr * kld] * temp;
				}
			}
		}
		__syncthreads();
	}
	return 0;
}

// DPBTRS 'U', one right-hand side = DTBSV('U','T','N') then DTBSV('U','N','N'), run by one
// wavefront with the band factor and the right-hand side in LDS.  Row-oriented forward sweep /
// column-oriented backward sweep: every b_i receives exactly the same subtractions in the same
// order as in the sequential DTBSV loops, so the result is bit-identical to them; the lanes
// only parallelise the independent updates of one elimination step.
__device__ void dpbtrs_upper_wave(int n, int kd, const real_t *ab, int ldab, real_t *b)
{
	const int lane = threadIdx.x;
	for (int j = 0; j < n; j++) {
		if (lane == 0) b[j] = b[j] / ab[(size_t)kd + (size_t)ldab * j];
		__syncthreads();
		const real_t yj = b[j];
		const int hi = j + kd < n - 1 ? j + kd : n - 1;
		for (int i = j + 1 + lane; i <= hi; i += blockDim.x) b[i] = b[i] - ab[(size_t)(kd + j - i) + (size_t)ldab * i] * yj;
		__syncthreads();
	}
	for (int j = n - 1; j >= 0; j--) {
		const bool nz = b[j] != 0.0;
		__syncthreads();
		if (nz) {
			if (lane == 0) b[j] = b[j] / ab[(size_t)kd + (size_t)ldab * j];
			__syncthreads();
			const real_t xj = b[j];
			const int lo = j - kd > 0 ? j - kd : 0;
			for (int i = j - 1 - lane; i >= lo; i -= blockDim.x) b[i] = b[i] - xj * ab[(size_t)(kd + i - j) + (size_t)ldab * j];
		}
		__syncthreads();
	}
}

// stage the factor in LDS when it fits, then solve
__device__ void band_solve(int n, int kd, const real_t *__restrict__ abd, int ldab, real_t *bbd, real_t *lds, int use_lds)
{
	const real_t *ab = abd;
	real_t *b = bbd;
	if (use_lds) {
		real_t *lab = lds, *lb = lds + (size_t)ldab * n;
		for (int t = threadIdx.x; t < ldab * n; t += blockDim.x) lab[t] = abd[t];
		for (int t = threadIdx.x; t < n; t += blockDim.x) lb[t] = bbd[t];
		ab = lab; b = lb;
	}
	__syncthreads();
	dpbtrs_upper_wave(n, kd, ab, ldab, b);
	if (use_lds) {
		for (int t = threadIdx.x; t < n; t += blockDim.x) bbd[t] = b[t];
		__syncthreads();
	}
}

// ------------------------------------------------------------------ 2D
__global__ __launch_bounds__(256) void setup_cg2_kernel(const real_t *__restrict__ so, int II, int JJ, int nstncl,
                                                         real_t *__restrict__ abd, int nabd1, int nabd2, int *info)
{
	const int I1 = II - 1, J1 = JJ - 1, I2 = I1 - 1;
	const int n = I2 * (J1 - 1);
	const size_t PS = (size_t)II * JJ;
	for (int kk = threadIdx.x + 1; kk <= n; kk += blockDim.x) {
		const int i = (kk - 1) % I2 + 2, j = (kk - 1) / I2 + 2; // 1-based
		const size_t x = (size_t)(i - 1) + (size_t)II * (size_t)(j - 1);
		ABD(II, kk) = so[KO * PS + x];
		ABD(I1, kk) = -so[KW * PS + x];
		ABD(3, kk) = nstncl == 5 ? -so[KNW * PS + x + 1] : 0.0;
		ABD(2, kk) = -so[KS * PS + x];
		ABD(1, kk) = nstncl == 5 ? -so[KSW * PS + x] : 0.0;
	}
	__syncthreads();
	__shared__ int flag;
	const int rc = dpbtf2_upper_wg(n, I1, abd, nabd1, &flag);
	if (threadIdx.x == 0) *info = rc;
}

__global__ __launch_bounds__(64) void solve_cg2_kernel(real_t *__restrict__ q, const real_t *__restrict__ qf, int II, int JJ,
                                                        const real_t *__restrict__ abd, real_t *__restrict__ bbd, int nabd1, int use_lds,
                                                        size_t bstride, int nabd2)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	const int I1 = II - 1, J1 = JJ - 1, I2 = I1 - 1;
	const int n = I2 * (J1 - 1);
	q += bstride * blockIdx.x; qf += bstride * blockIdx.x; bbd += (size_t)nabd2 * blockIdx.x; // batch item (common.h Batch)
	for (int kk = threadIdx.x; kk < n; kk += blockDim.x) {
		const int i = kk % I2 + 1, j = kk / I2 + 1; // 0-based
		bbd[kk] = qf[(size_t)i + (size_t)II * j];
	}
	__syncthreads();
	band_solve(n, I1, abd, nabd1, bbd, lds, use_lds);
	__syncthreads();
	for (int kk = threadIdx.x; kk < n; kk += blockDim.x) {
		const int i = kk % I2 + 1, j = kk / I2 + 1;
		q[(size_t)i + (size_t)II * j] = bbd[kk];
	}
}

void setup_cg2(const real_t *so, int II, int JJ, int nstncl, real_t *abd, int nabd1, int nabd2, int *info, hipStream_t st)
{
	hipLaunchKernelGGL(setup_cg2_kernel, dim3(1), dim3(256), 0, st, so, II, JJ, nstncl, abd, nabd1, nabd2, info);
}

void solve_cg2(real_t *q, const real_t *qf, int II, int JJ, const real_t *abd, real_t *bbd, int nabd1, int nabd2, hipStream_t st,
               Batch bt)
{
	size_t shm = ((size_t)nabd1 * nabd2 + nabd2 + 2) * sizeof(real_t);
	int use_lds = shm <= 60 * 1024;
	hipLaunchKernelGGL(solve_cg2_kernel, dim3(bt.n), dim3(64), use_lds ? shm : 0, st, q, qf, II, JJ, abd, bbd, nabd1, use_lds,
	                   bt.stride, nabd2);
}

// ------------------------------------------------------------------ 3D
__global__ __launch_bounds__(256) void setup_cg3_kernel(const real_t *__restrict__ so, int II, int JJ, int KK, int nstncl,
                                                         real_t *__restrict__ abd, int nabd1, int nabd2, int *info)
{
	const int i1 = II - 1, j1 = JJ - 1, k1 = KK - 1, i2 = i1 - 1;
	const int ibw = i2 * j1 + 1;
	const int nxy = i2 * (j1 - 1), n = nxy * (k1 - 1);
	const size_t sj = II, sk = (size_t)II * JJ, PS = sk * KK;
	const bool full = nstncl == 14;
	for (int kl = threadIdx.x + 1; kl <= n; kl += blockDim.x) {
		const int t = kl - 1;
		const int i = t % i2 + 2, j = (t / i2) % (j1 - 1) + 2, k = t / nxy + 2; // 1-based
		const size_t x = (size_t)(i - 1) + sj * (size_t)(j - 1) + sk * (size_t)(k - 1);
		ABD(ibw + 1, kl) = so[KP * PS + x];
		ABD(ibw, kl) = -so[KPW * PS + x];
		ABD(ibw - i1 + 3, kl) = full ? -so[KPNW * PS + x + 1] : 0.0;
		ABD(ibw - i1 + 2, kl) = -so[KPS * PS + x];
		ABD(ibw - i1 + 1, kl) = full ? -so[KPSW * PS + x] : 0.0;
		ABD(ibw - (j1 - 2) * i2 + 2, kl) = full ? -so[KBNE * PS + x + 1 + sj] : 0.0;
		ABD(ibw - (j1 - 2) * i2 + 1, kl) = full ? -so[KBN * PS + x + sj] : 0.0;
		ABD(ibw - (j1 - 2) * i2, kl) = full ? -so[KBNW * PS + x + sj] : 0.0;
		ABD(ibw - (j1 - 1) * i2 + 2, kl) = full ? -so[KBE * PS + x + 1] : 0.0;
		ABD(ibw - (j1 - 1) * i2 + 1, kl) = -so[KB * PS + x];
		ABD(ibw - (j1 - 1) * i2, kl) = full ? -so[KBW * PS + x] : 0.0;
		ABD(3, kl) = full ? -so[KBSE * PS + x + 1] : 0.0;
		ABD(2, kl) = full ? -so[KBS * PS + x] : 0.0;
		ABD(1, kl) = full ? -so[KBSW * PS + x] : 0.0;
	}
	__syncthreads();
	__shared__ int flag;
	const int rc = dpbtf2_upper_wg(n, ibw, abd, nabd1, &flag);
	if (threadIdx.x == 0) *info = rc;
}

__global__ __launch_bounds__(64) void solve_cg3_kernel(real_t *__restrict__ q, const real_t *__restrict__ qf, int II, int JJ, int KK,
                                                        const real_t *__restrict__ abd, real_t *__restrict__ bbd, int nabd1, int use_lds)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	const int i1 = II - 1, j1 = JJ - 1, k1 = KK - 1, i2 = i1 - 1;
	const int ibw = i2 * j1 + 1;
	const int nxy = i2 * (j1 - 1), n = nxy * (k1 - 1);
	const size_t sj = II, sk = (size_t)II * JJ;
	for (int t = threadIdx.x; t < n; t += blockDim.x) {
		const int i = t % i2 + 1, j = (t / i2) % (j1 - 1) + 1, k = t / nxy + 1; // 0-based
		bbd[t] = qf[(size_t)i + sj * j + sk * k];
	}
	__syncthreads();
	band_solve(n, ibw, abd, nabd1, bbd, lds, use_lds);
	__syncthreads();
	for (int t = threadIdx.x; t < n; t += blockDim.x) {
		const int i = t % i2 + 1, j = (t / i2) % (j1 - 1) + 1, k = t / nxy + 1;
		q[(size_t)i + sj * j + sk * k] = bbd[t];
	}
}

void setup_cg3(const real_t *so, int II, int JJ, int KK, int nstncl, real_t *abd, int nabd1, int nabd2, int *info, hipStream_t st)
{
	hipLaunchKernelGGL(setup_cg3_kernel, dim3(1), dim3(256), 0, st, so, II, JJ, KK, nstncl, abd, nabd1, nabd2, info);
}

void solve_cg3(real_t *q, const real_t *qf, int II, int JJ, int KK, const real_t *abd, real_t *bbd, int nabd1, int nabd2, hipStream_t st)
{
	size_t shm = ((size_t)nabd1 * nabd2 + nabd2 + 2) * sizeof(real_t);
	int use_lds = shm <= 60 * 1024;
	hipLaunchKernelGGL(solve_cg3_kernel, dim3(1), dim3(64), use_lds ? shm : 0, st, q, qf, II, JJ, KK, abd, bbd, nabd1, use_lds);
}

} // namespace cedar_amd
