// 3D periodic boundary conditions (ibc = 1 per_y, 2 per_x, 3 per_xy, 5 per_z, 6 per_xz, 7 per_yz, 8 per_xyz;
// src/2d/ftn/BMG_get_bc.f90:13-20): ghost refreshes, transfers, set-up and the dense coarsest solve around
// the Dirichlet kernels.  Replaces the periodic branches of
//   BMG3_SymStd_relax_GS          (src/3d/ftn/BMG3_SymStd_relax_GS.f90:188-357)   relax3_gs_per (relax3d.hip): row-class
//                                                                                 passes or one launch per colour, each
//                                                                                 followed by wrap3_colour
//   BMG3_SymStd_restrict          (..._restrict.f90:78-103)        ghost refresh (y, x, z), then restrict
//   BMG3_SymStd_interp_add        (..._interp_add.f90:242-286)     interp_add, then ghost refresh
//   BMG3_SymStd_SETUP_interp_OI   (..._SETUP_interp_OI.f90:808-2811)  the Dirichlet formulas with loops started one
//                                                                  coarse point earlier in a periodic direction,
//                                                                  ghost refresh of the weights after every phase
//   BMG3_SymStd_SETUP_ITLI{07,27}_ex (periodic tails)              Galerkin product, then ghost refresh of 14 arrays
//   BMG3_SymStd_SETUP_cg_LU       (..._SETUP_cg_LU.f90:200-619)    dense matrix + DPOTRF
//   BMG3_SymStd_SOLVE_cg          (..._SOLVE_cg.f90:117-212)       DPOTRS, mean removal, ghosts
// Where the reference is self-consistent the results are identical (tests/test_gpu_periodic.py against goldens made
// by the reference's Fortran); DESIGN.md section 6 lists what the reference leaves undefined (the x / y ghost
// loops of interp_add) or assembles wrongly (dense matrix for per_xz / per_xyz / per_xy with nx != ny) and what
// this file does there: the periodic operator itself.
#include "common.h"

namespace cedar_amd {

__host__ __device__ static inline bool per3_x(int ipn) { return ipn == 2 || ipn == 3 || ipn == 6 || ipn == 8; }
__host__ __device__ static inline bool per3_y(int ipn) { return ipn == 1 || ipn == 3 || ipn == 7 || ipn == 8; }
__host__ __device__ static inline bool per3_z(int ipn) { return ipn == 5 || ipn == 6 || ipn == 7 || ipn == 8; }

bool periodic3_code_ok(int ipn) { return ipn == 0 || per3_x(ipn) || per3_y(ipn) || per3_z(ipn); }

// ------------------------------------------------------------------ ghost refreshes
// One workgroup per (plane, array).  Planes k = k0 + kstep*blockIdx.x (0-based).  mode bit 0: y ghosts
//   Q(I,1,K)=Q(I,J1,K), Q(I,JJ,K)=Q(I,2,K), I = 1..II;  bit 1: x ghosts Q(1,J,K)=Q(I1,J,K), Q(II,J,K)=Q(2,J,K) for
// the rows J = j0 + jstep*r < jend (0-based).  xfirst: x before y (the sweep's order, relax_GS.f90:266-276), else y
// before x (restrict.f90:78-95).  Either way the second phase copies what the first one wrote: barrier between.
__global__ __launch_bounds__(256) void wrap3_xy_kernel(real_t *__restrict__ a, int II, int JJ, int KK, int mode, int xfirst,
                                                       int k0, int kstep, int j0, int jstep, int jend)
{
	real_t *p = a + ((size_t)blockIdx.y * KK + (size_t)(k0 + kstep * (int)blockIdx.x)) * (size_t)II * JJ;
	for (int phase = 0; phase < 2; phase++) {
		const bool xphase = (phase == 0) == (xfirst != 0);
		if (xphase && (mode & 2)) {
			for (int j = j0 + jstep * (int)threadIdx.x; j < jend; j += jstep * (int)blockDim.x) {
				p[(size_t)II * j] = p[(size_t)II * j + II - 2];
				p[(size_t)II * j + II - 1] = p[(size_t)II * j + 1];
			}
		} else if (!xphase && (mode & 1)) {
			for (int i = threadIdx.x; i < II; i += blockDim.x) {
				p[i] = p[i + (size_t)II * (JJ - 2)];
				p[i + (size_t)II * (JJ - 1)] = p[i + (size_t)II];
			}
		}
		__syncthreads();
	}
}

// Q(I,J,1)=Q(I,J,K1), Q(I,J,KK)=Q(I,J,2) over the full planes
__global__ __launch_bounds__(256) void wrap3_z_kernel(real_t *__restrict__ a, int II, int JJ, int KK)
{
	const size_t P = (size_t)II * JJ;
	real_t *p = a + (size_t)blockIdx.y * P * KK;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < P; t += (size_t)gridDim.x * blockDim.x) {
		p[t] = p[t + P * (size_t)(KK - 2)];
		p[t + P * (size_t)(KK - 1)] = p[t + P];
	}
}

static void wrap3_z(real_t *a, int II, int JJ, int KK, int narrays, hipStream_t st)
{
	const size_t P = (size_t)II * JJ;
	const unsigned gx = (unsigned)((P + 255) / 256 < 1024 ? (P + 255) / 256 : 1024);
	hipLaunchKernelGGL(wrap3_z_kernel, dim3(gx, narrays), dim3(256), 0, st, a, II, JJ, KK);
}

// full ghost refresh of `narrays` stacked arrays: y, x, then z (restrict.f90:78-103)
void wrap3(real_t *a, int II, int JJ, int KK, int narrays, int ipn, hipStream_t st)
{
	if (narrays <= 0 || II < 3 || JJ < 3 || KK < 3) return;
	const int mode = (per3_y(ipn) ? 1 : 0) | (per3_x(ipn) ? 2 : 0);
	if (mode) hipLaunchKernelGGL(wrap3_xy_kernel, dim3(KK, narrays), dim3(256), 0, st, a, II, JJ, KK, mode, 0, 0, 1, 0, 1, JJ);
	if (per3_z(ipn)) wrap3_z(a, II, JJ, KK, narrays, st);
}

// the refreshes that follow one colour of the sweep (relax_GS.f90:266-276, :318-328): x ghosts of the rows and y
// ghosts of the planes the colour visited.  27-point: rows j = 1 + jb + 2r, planes k = 1 + kb + 2c (0-based);
// 7-point (jb = kb = -1): every interior row and plane.
void wrap3_colour(real_t *q, int II, int JJ, int KK, int jb, int kb, int ipn, hipStream_t st)
{
	const int mode = (per3_y(ipn) ? 1 : 0) | (per3_x(ipn) ? 2 : 0);
	if (!mode) return;
	const int k0 = kb < 0 ? 1 : 1 + kb, kstep = kb < 0 ? 1 : 2;
	const int j0 = jb < 0 ? 1 : 1 + jb, jstep = jb < 0 ? 1 : 2;
	const int nk = (KK - 2 - (k0 - 1) + kstep - 1) / kstep;
	if (nk <= 0) return;
	hipLaunchKernelGGL(wrap3_xy_kernel, dim3(nk, 1), dim3(256), 0, st, q, II, JJ, KK, mode, 1, k0, kstep, j0, jstep, JJ - 1);
}

void wrap3_sweep_end(real_t *q, int II, int JJ, int KK, int ipn, hipStream_t st)
{
	if (per3_z(ipn)) wrap3_z(q, II, JJ, KK, 1, st);
}

// ------------------------------------------------------------------ transfers and set-up
void restrict3_per(real_t *q, real_t *qc, const real_t *ci, int II, int JJ, int KK, int IIC, int JJC, int KKC, int ipn,
                   hipStream_t st)
{
	wrap3(q, II, JJ, KK, 1, ipn, st);
	restrict3(q, qc, ci, II, JJ, KK, IIC, JJC, KKC, st);
}

void interp_add3_per(real_t *q, const real_t *qc, const real_t *so, real_t *res, const real_t *ci,
                     int IIC, int JJC, int KKC, int IIF, int JJF, int KKF, int ipn, hipStream_t st)
{
	interp_add3(q, qc, so, res, ci, IIC, JJC, KKC, IIF, JJF, KKF, st);
	wrap3(q, IIF, JJF, KKF, 1, ipn, st);
}

void setup_interp3_per(const real_t *so, real_t *ci, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int ifd, int ipn,
                       hipStream_t st)
{
	const int ilo = per3_x(ipn) ? 2 : 3, jlo = per3_y(ipn) ? 2 : 3, klo = per3_z(ipn) ? 2 : 3;
	for (int phase = 0; phase < 3; phase++) {
		setup_interp3_phase(so, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, phase, ilo, jlo, klo, st);
		wrap3(ci, IIC, JJC, KKC, 26, ipn, st);
	}
}

void galerkin3_per(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC,
                   int ifd, int ipn, hipStream_t st)
{
	galerkin3(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
	wrap3(soc, IIC, JJC, KKC, 14, ipn, st);
}

// ------------------------------------------------------------------ coarsest grid: dense Cholesky
// slot s of a stencil stored at P couples P+EA[s] with P+EB[s] (BMG3_SymStd_relax_GS.f90:104-131)
__constant__ signed char EA3[14][3] = {
	{ 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, -1, 0 }, { 0, 0, 0 },
	{ 0, -1, 0 }, { 0, -1, 0 }, { -1, -1, 0 }, { -1, 0, 0 }, { -1, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }
};
__constant__ signed char EB3[14][3] = {
	{ 0, 0, 0 }, { -1, 0, 0 }, { 0, -1, 0 }, { 0, 0, -1 }, { -1, -1, 0 }, { -1, 0, 0 }, { -1, 0, -1 },
	{ -1, 0, -1 }, { 0, 0, -1 }, { 0, 0, -1 }, { 0, 0, -1 }, { 0, -1, -1 }, { 0, -1, -1 }, { -1, -1, -1 }
};

#define ABD(r, c) abd[(size_t)(r) + (size_t)nabd1 * (size_t)(c)]

// unknown number of the (1-based) grid point after the periodic wrap; -1 = outside
__device__ __forceinline__ int unknown3(int i, int j, int k, int nx, int ny, int nz, int ipn)
{
	if (i < 2 || i > nx + 1) { if (!per3_x(ipn)) return -1; i = i < 2 ? i + nx : i - nx; }
	if (j < 2 || j > ny + 1) { if (!per3_y(ipn)) return -1; j = j < 2 ? j + ny : j - ny; }
	if (k < 2 || k > nz + 1) { if (!per3_z(ipn)) return -1; k = k < 2 ? k + nz : k - nz; }
	return (i - 2) + nx * ((j - 2) + ny * (k - 2));
}

// One workgroup.  The matrix is assembled by walking the stencil (every coefficient stored at an interior point
// couples two grid points, each mapped to its unknown through the wrap), then factored with the operation order of
// the unblocked DPOTF2 'U': column c of row j is owned by one thread, which accumulates its dot product in the
// reference-BLAS order -- results independent of the thread count.
__global__ __launch_bounds__(1024) void setup_cg3_per_kernel(const real_t *__restrict__ so, int II, int JJ, int KK,
                                                             real_t *__restrict__ abd, int nabd1, int ipn, int *info)
{
	const int nx = II - 2, ny = JJ - 2, nz = KK - 2, N = nx * ny * nz;
	const int tid = threadIdx.x, nt = blockDim.x;
	const size_t PS = (size_t)II * JJ * KK;
	for (size_t t = tid; t < (size_t)N * N; t += nt) ABD(t % N, t / N) = 0.0;
	__syncthreads();
	// an extent of 2 in a periodic direction makes two slots name the same pair: keep the serial order then
	const bool alias = (per3_x(ipn) && nx < 3) || (per3_y(ipn) && ny < 3) || (per3_z(ipn) && nz < 3);
	for (int r = alias ? (tid == 0 ? 0 : N) : tid; r < N; r += alias ? 1 : nt) {
		const int i = 2 + r % nx, j = 2 + (r / nx) % ny, k = 2 + r / (nx * ny);
		const size_t x = (size_t)(i - 1) + (size_t)II * ((size_t)(j - 1) + (size_t)JJ * (size_t)(k - 1));
		ABD(r, r) = so[x];
		for (int s = 1; s < 14; s++) {
			const int X = unknown3(i + EA3[s][0], j + EA3[s][1], k + EA3[s][2], nx, ny, nz, ipn);
			const int Y = unknown3(i + EB3[s][0], j + EB3[s][1], k + EB3[s][2], nx, ny, nz, ipn);
			if (X < 0 || Y < 0) continue;
			const real_t v = -so[(size_t)s * PS + x];
			if (X <= Y) ABD(X, Y) = v;
			else ABD(Y, X) = v;
		}
	}
	__syncthreads();
	__shared__ int rc;
	__shared__ real_t rinv;
	if (tid == 0) rc = 0;
	__syncthreads();
	for (int j = 0; j < N; j++) {
		if (tid == 0) {
			real_t dot = 0.0;
			for (int i = 0; i < j; i++) dot = dot + ABD(i, j) * ABD(i, j);
			real_t ajj = ABD(j, j) - dot;
			if (!(ajj > 0.0)) {
				ABD(j, j) = ajj;
				rc = j + 1;
			} else {
				ajj = sqrt(ajj);
				ABD(j, j) = ajj;
				rinv = 1.0 / ajj;
			}
		}
		__syncthreads();
		if (rc) break;
		const real_t r = rinv;
		for (int c = j + 1 + tid; c < N; c += nt) {
			real_t temp = 0.0;
			for (int i = 0; i < j; i++) temp = temp + ABD(i, c) * ABD(i, j);
			real_t v = ABD(j, c) + (-1.0) * temp;
			ABD(j, c) = r * v;
		}
		__syncthreads();
	}
	if (tid == 0) *info = rc;
}

// DPOTRS 'U' (inv(U^T) then inv(U), dtrsm.f operation order per entry), the mean removal of SOLVE_cg.f90:158-180
// (sequential sum: one lane) and the ghosts (:182-210; one workgroup, barriers between the three directions).
__global__ __launch_bounds__(1024) void solve_cg3_per_kernel(real_t *__restrict__ q, const real_t *__restrict__ qf,
                                                             int II, int JJ, int KK, const real_t *__restrict__ abd,
                                                             real_t *__restrict__ bbd, int nabd1, int ipn)
{
	const int nx = II - 2, ny = JJ - 2, nz = KK - 2, N = nx * ny * nz;
	const int tid = threadIdx.x, nt = blockDim.x;
#define XOF(r) ((size_t)(1 + (r) % nx) + (size_t)II * ((size_t)(1 + ((r) / nx) % ny) + (size_t)JJ * (size_t)(1 + (r) / (nx * ny))))
	for (int r = tid; r < N; r += nt) bbd[r] = qf[XOF(r)];
	__syncthreads();
	// forward: b_i = (b_i - sum_{k<i} U(k,i) b_k) / U(i,i), the sum taken in increasing k for every i
	for (int k = 0; k < N; k++) {
		if (tid == 0) bbd[k] = bbd[k] / ABD(k, k);
		__syncthreads();
		const real_t bk = bbd[k];
		for (int i = k + 1 + tid; i < N; i += nt) bbd[i] = bbd[i] - ABD(k, i) * bk;
		__syncthreads();
	}
	for (int k = N - 1; k >= 0; k--) {
		if (tid == 0 && bbd[k] != 0.0) bbd[k] = bbd[k] / ABD(k, k);
		__syncthreads();
		const real_t bk = bbd[k];
		if (bk != 0.0)
			for (int i = tid; i < k; i += nt) bbd[i] = bbd[i] - bk * ABD(i, k);
		__syncthreads();
	}
	__shared__ real_t cshift;
	if (tid == 0) {
		real_t cint = 0.0, qint = 0.0;
		for (int r = 0; r < N; r++) {
			qint = qint + bbd[r];
			cint = cint + 1;
		}
		cshift = -qint / cint;
	}
	__syncthreads();
	for (int r = tid; r < N; r += nt) q[XOF(r)] = bbd[r] + cshift;
	__syncthreads();
#undef XOF
	const size_t P = (size_t)II * JJ;
	if (per3_x(ipn))
		for (int t = tid; t < JJ * KK; t += nt) {
			real_t *row = q + (size_t)II * t;
			row[0] = row[II - 2];
			row[II - 1] = row[1];
		}
	__syncthreads();
	if (per3_y(ipn))
		for (int t = tid; t < II * KK; t += nt) {
			real_t *p = q + P * (size_t)(t / II) + (t % II);
			p[0] = p[(size_t)II * (JJ - 2)];
			p[(size_t)II * (JJ - 1)] = p[II];
		}
	__syncthreads();
	if (per3_z(ipn))
		for (int t = tid; t < II * JJ; t += nt) {
			q[t] = q[t + P * (size_t)(KK - 2)];
			q[t + P * (size_t)(KK - 1)] = q[t + P];
		}
}
#undef ABD

void setup_cg3_per(const real_t *so, int II, int JJ, int KK, real_t *abd, int nabd1, int ipn, int *info, hipStream_t st)
{
	hipLaunchKernelGGL(setup_cg3_per_kernel, dim3(1), dim3(1024), 0, st, so, II, JJ, KK, abd, nabd1, ipn, info);
}

void solve_cg3_per(real_t *q, const real_t *qf, int II, int JJ, int KK, const real_t *abd, real_t *bbd, int nabd1, int ipn,
                   hipStream_t st)
{
	hipLaunchKernelGGL(solve_cg3_per_kernel, dim3(1), dim3(1024), 0, st, q, qf, II, JJ, KK, abd, bbd, nabd1, ipn);
}

} // namespace cedar_amd
