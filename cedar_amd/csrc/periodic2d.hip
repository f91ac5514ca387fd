// 2D periodic boundary conditions (ibc = 1 per_y, 2 per_x, 3 per_xy; BMG_get_bc.f90:13-16): point
// relaxation, transfers, set-up and the coarsest solve.  Replaces the periodic branches of
//   BMG2_SymStd_relax_GS          (src/2d/ftn/BMG2_SymStd_relax_GS.f90:139-226)
//   BMG2_SymStd_restrict          (..._restrict.f90:94-128)          ghost refresh, then restrict
//   BMG2_SymStd_interp_add        (..._interp_add.f90:139-156)       interp_add, then ghost wraps
//   BMG2_SymStd_SETUP_interp_OI   (..._SETUP_interp_OI.f90:258-618)
//   BMG2_SymStd_SETUP_ITLI_ex     (..._SETUP_ITLI_ex.f90:216-247, :333-364)  Galerkin, then wraps
//   BMG2_SymStd_SETUP_cg_LU       (..._SETUP_cg_LU.f90:148-218, :262-330)    dense matrix + DPOTRF
//   BMG2_SymStd_SOLVE_cg          (..._SOLVE_cg.f90:95-163)          DPOTRS, mean removal, wraps
// Same operation order as the reference => relax / restrict / interp_add / interpolation bit-identical.
// Periodic line relaxation (cyclic tridiagonals, Sherman-Morrison) lives in lines.hip.
#include "common.h"
#include <cfloat>

namespace cedar_amd {

__device__ __forceinline__ bool per_x(int ipn) { return ipn == 2 || ipn == 3; }
__device__ __forceinline__ bool per_y(int ipn) { return ipn == 1 || ipn == 3; }

// ------------------------------------------------------------------ ghost wraps
// mode bit 0: y wrap  Q(I,1)=Q(I,J1), Q(I,JJ)=Q(I,2) for I = 1..II
// mode bit 1: x wrap  Q(1,J)=Q(I1,J), Q(II,J)=Q(2,J) for J = 1..JJ   (after the y wrap: corners follow)
// One workgroup per plane; the two phases are separated by a barrier.
__global__ __launch_bounds__(256) void wrap2_kernel(real_t *__restrict__ q, int II, int JJ, int mode)
{
	real_t *p = q + (size_t)blockIdx.x * II * JJ;
	if (mode & 1)
		for (int i = threadIdx.x; i < II; i += blockDim.x) {
			p[i] = p[i + (size_t)II * (JJ - 2)];
			p[i + (size_t)II * (JJ - 1)] = p[i + (size_t)II];
		}
	__syncthreads();
	if (mode & 2)
		for (int j = threadIdx.x; j < JJ; j += blockDim.x) {
			p[(size_t)II * j] = p[(size_t)II * j + II - 2];
			p[(size_t)II * j + II - 1] = p[(size_t)II * j + 1];
		}
}

void wrap2(real_t *q, int II, int JJ, int nplanes, int do_y, int do_x, hipStream_t st)
{
	const int mode = (do_y ? 1 : 0) | (do_x ? 2 : 0);
	if (mode == 0 || nplanes <= 0) return;
	hipLaunchKernelGGL(wrap2_kernel, dim3(nplanes), dim3(256), 0, st, q, II, JJ, mode);
}

// ------------------------------------------------------------------ relax
// One workgroup per grid row; the row lives in LDS while its colours are relaxed, the x wrap follows
// each colour exactly like the reference (:177-180).  NINE: both i-colours of a row of the current
// j-class (rows of one class do not couple); five point: the one colour (i+j parity) of this pass.
template <bool NINE>
__global__ __launch_bounds__(256) void relax2_per_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                          real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                          int II, int JJ, int jbeg, int jstep, int first, int ipn)
{
	extern __shared__ real_t row[];
	const int j = jbeg + jstep * (int)blockIdx.x; // 1-based
	if (j > JJ - 1) return;
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t r0 = sj * (size_t)(j - 1); // offset of Q(1,j)
	const int I1 = II - 1;
	for (int i = threadIdx.x; i < II; i += blockDim.x) row[i] = q[r0 + i];
	__syncthreads();
	const int ncol = NINE ? 2 : 1;
	for (int c = 0; c < ncol; c++) {
		// nine point: IBEG = first, then the other one (LSTART..LEND, :144-152); five point: IBEG = mod(J+JO,2)+2
		const int ibeg = NINE ? (c == 0 ? first : 5 - first) : (j + first) % 2 + 2;
		for (int i = ibeg + 2 * (int)threadIdx.x; i <= I1; i += 2 * (int)blockDim.x) { // 1-based
			const size_t x = r0 + (size_t)(i - 1);
			real_t s = qf[x];
			s = s + so[KW * PS + x] * row[i - 2];
			s = s + so[KW * PS + x + 1] * row[i];
			s = s + so[KS * PS + x] * q[x - sj];
			s = s + so[KS * PS + x + sj] * q[x + sj];
			if (NINE) {
				s = s + so[KSW * PS + x] * q[x - 1 - sj];
				s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
				s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
				s = s + so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
			}
			row[i - 1] = s * sor[PS + x];
		}
		__syncthreads();
		if (per_x(ipn) && threadIdx.x == 0) {
			row[0] = row[I1 - 1];
			row[II - 1] = row[1];
		}
		__syncthreads();
	}
	const int lo = per_x(ipn) ? 0 : 1, hi = per_x(ipn) ? II : II - 1;
	for (int i = lo + (int)threadIdx.x; i < hi; i += blockDim.x) q[r0 + i] = row[i];
}

int relax2_gs_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                  int II, int JJ, int nstncl, int updown, int ipn, hipStream_t st)
{
	if (II < 3 || JJ < 3) return 0;
	const size_t shm = (size_t)II * sizeof(real_t);
	if (shm > 64 * 1024) return 1; // row does not fit the default LDS window
	const int J1 = JJ - 1;
	const bool down = updown == BMG_DOWN;
	if (nstncl == 5) {
		for (int c = 0; c < 2; c++) {
			const int jbeg = down ? 2 + c : 3 - c; // LSTART..LEND
			if (jbeg > J1) continue;
			const int nrows = (J1 - jbeg) / 2 + 1;
			hipLaunchKernelGGL(relax2_per_kernel<true>, dim3(nrows), dim3(256), shm, st, so, qf, q, sor, II, JJ, jbeg, 2,
			                   down ? 2 : 3, ipn);
		}
	} else {
		for (int c = 0; c < 2; c++) {
			const int jo = down ? 2 + c : 3 - c;
			hipLaunchKernelGGL(relax2_per_kernel<false>, dim3(J1 - 1), dim3(256), shm, st, so, qf, q, sor, II, JJ, 2, 1, jo, ipn);
		}
	}
	if (ipn == 1 || ipn == 3) wrap2(q, II, JJ, 1, 1, 0, st);
	return 0;
}

// ------------------------------------------------------------------ restrict / interp_add / Galerkin wrappers
void restrict2_per(real_t *q, real_t *qc, const real_t *ci, int Nx, int Ny, int Nxc, int Nyc, int ipn, hipStream_t st)
{
	const bool wy = (ipn == 1 || ipn == 3) && Ny / 2 + 1 == Nyc;
	const bool wx = (ipn == 2 || ipn == 3) && Nx / 2 + 1 == Nxc;
	wrap2(q, Nx, Ny, 1, wy, wx, st);
	restrict2(q, qc, ci, Nx, Ny, Nxc, Nyc, st);
}

void interp_add2_per(real_t *q, const real_t *qc, real_t *res, const real_t *so, const real_t *ci,
                     int IIC, int JJC, int IIF, int JJF, int ipn, hipStream_t st)
{
	interp_add2(q, qc, res, so, ci, IIC, JJC, IIF, JJF, st);
	wrap2(q, IIF, JJF, 1, ipn == 1 || ipn == 3, ipn == 2 || ipn == 3, st);
}

void galerkin2_per(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, int ipn,
                   hipStream_t st)
{
	galerkin2(so, soc, ci, IIF, JJF, IIC, JJC, ifd, st);
	wrap2(soc, IIC, JJC, 5, ipn == 1 || ipn == 3, ipn == 2 || ipn == 3, st);
}

// ------------------------------------------------------------------ interpolation set-up
__device__ __forceinline__ real_t rmaxp(real_t a, real_t b) { return a > b ? a : b; }
__device__ __forceinline__ real_t rminp(real_t a, real_t b) { return a < b ? a : b; }
__device__ __forceinline__ real_t lump_p(real_t off, real_t diag, real_t s, real_t ep, real_t eps)
{
	return off + (diag - s) * rmaxp(diag - (1.0 + ep) * s, 0.0) / (fabs(diag - (1.0 + ep) * s) + eps);
}
#define SO(i, j, s) so[(size_t)((i)-1) + (size_t)IIF * ((size_t)((j)-1) + (size_t)JJF * (size_t)(s))]
#define CIW(ic, jc, s) ci[(size_t)((ic)-1) + (size_t)IIC * ((size_t)((jc)-1) + (size_t)JJC * (size_t)(s))]

// the three point formulas (shared with the non-periodic driver of the reference, :322-341, :377-394, :431-466)
__device__ __forceinline__ void ci_xedge(const real_t *__restrict__ so, real_t *ci, int IIF, int JJF, int IIC, int JJC,
                                         int ifd, int i, int j, int ic, int jc)
{
	real_t a, b;
	if (ifd != 1) {
		a = SO(i, j, KW) + SO(i, j, KNW) + SO(i, j + 1, KSW);
		b = SO(i - 1, j, KW) + SO(i - 1, j, KSW) + SO(i - 1, j + 1, KNW);
	} else {
		a = SO(i, j, KW);
		b = SO(i - 1, j, KW);
	}
	const real_t ep = rminp(fabs(a / SO(i - 1, j, KO)), fabs(b / SO(i - 1, j, KO)));
	real_t sum = a + b + SO(i - 1, j, KS) + SO(i - 1, j + 1, KS);
	sum = lump_p(a + b, SO(i - 1, j, KO), sum, ep, DBL_EPSILON);
	sum = 1.0 / sum;
	CIW(ic, jc, LR) = a * sum;
	CIW(ic, jc, LL) = b * sum;
}

__device__ __forceinline__ void ci_yedge(const real_t *__restrict__ so, real_t *ci, int IIF, int JJF, int IIC, int JJC,
                                         int ifd, int i, int j, int ic, int jc)
{
	real_t a, b;
	if (ifd != 1) {
		a = SO(i, j, KS) + SO(i, j, KNW) + SO(i + 1, j, KSW);
		b = SO(i, j - 1, KS) + SO(i, j - 1, KSW) + SO(i + 1, j - 1, KNW);
	} else {
		a = SO(i, j, KS);
		b = SO(i, j - 1, KS);
	}
	const real_t ep = rminp(fabs(a / SO(i, j - 1, KO)), fabs(b / SO(i, j - 1, KO)));
	real_t sum = a + b + SO(i, j - 1, KW) + SO(i + 1, j - 1, KW);
	sum = lump_p(a + b, SO(i, j - 1, KO), sum, ep, DBL_EPSILON);
	sum = 1.0 / sum;
	CIW(ic, jc, LA) = a * sum;
	CIW(ic, jc, LB) = b * sum;
}

__device__ __forceinline__ void ci_centre(const real_t *__restrict__ so, real_t *ci, int IIF, int JJF, int IIC, int JJC,
                                          int ifd, int i, int j, int ic, int jc)
{
	real_t sum, ep, s;
	const real_t d = SO(i - 1, j - 1, KO);
	if (ifd != 1) {
		sum = SO(i - 1, j - 1, KW) + SO(i - 1, j, KNW) + SO(i - 1, j, KS)
		      + SO(i, j, KSW) + SO(i, j - 1, KW) + SO(i, j - 1, KNW)
		      + SO(i - 1, j - 1, KS) + SO(i - 1, j - 1, KSW);
		ep = rminp(rminp(fabs((SO(i - 1, j - 1, KSW) + SO(i - 1, j - 1, KW) + SO(i - 1, j, KNW)) / d),
		                 fabs((SO(i - 1, j, KNW) + SO(i - 1, j, KS) + SO(i, j, KSW)) / d)),
		           rminp(fabs((SO(i, j, KSW) + SO(i, j - 1, KW) + SO(i, j - 1, KNW)) / d),
		                 fabs((SO(i, j - 1, KNW) + SO(i - 1, j - 1, KS) + SO(i - 1, j - 1, KSW)) / d)));
		sum = lump_p(sum, d, sum, ep, DBL_EPSILON);
		s = 1.0 / sum;
		CIW(ic, jc, LSW) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LL) + SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LB)
		                    + SO(i - 1, j - 1, KSW)) * s;
		CIW(ic, jc, LSE) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LR) + SO(i, j - 1, KW) * CIW(ic, jc, LB)
		                    + SO(i, j - 1, KNW)) * s;
		CIW(ic, jc, LNW) = (SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LA) + SO(i - 1, j, KS) * CIW(ic, jc, LL)
		                    + SO(i - 1, j, KNW)) * s;
		CIW(ic, jc, LNE) = (SO(i - 1, j, KS) * CIW(ic, jc, LR) + SO(i, j - 1, KW) * CIW(ic, jc, LA)
		                    + SO(i, j, KSW)) * s;
	} else {
		sum = SO(i - 1, j - 1, KW) + SO(i - 1, j, KS) + SO(i, j - 1, KW) + SO(i - 1, j - 1, KS);
		ep = rminp(rminp(fabs(SO(i - 1, j - 1, KW) / d), fabs(SO(i - 1, j, KS) / d)),
		           rminp(fabs(SO(i, j - 1, KW) / d), fabs(SO(i - 1, j - 1, KS) / d)));
		sum = lump_p(sum, d, sum, ep, DBL_EPSILON);
		s = 1.0 / sum;
		CIW(ic, jc, LSW) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LL) + SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LB)) * s;
		CIW(ic, jc, LSE) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LR) + SO(i, j - 1, KW) * CIW(ic, jc, LB)) * s;
		CIW(ic, jc, LNW) = (SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LA) + SO(i - 1, j, KS) * CIW(ic, jc, LL)) * s;
		CIW(ic, jc, LNE) = (SO(i - 1, j, KS) * CIW(ic, jc, LR) + SO(i, j - 1, KW) * CIW(ic, jc, LA)) * s;
	}
}
#undef SO
#undef CIW

// index bookkeeping of the periodic driver (:279-315).  The reference walks the fine index with
// I <- MAX(MOD(I+2,IIFC), MIN(I+2,3)): it is 2(ic-1) except that the step onto IIFC wraps to 3.
struct PerIdx {
	int IBEGC, IENDC, IBEG_x, IEND_x, IIFC, JBEGC, JENDC, JBEG_y, JEND_y, JJFC;
};
__device__ __forceinline__ int fine_of(int c, int nfc) { const int f = 2 * (c - 1); return f == nfc ? 3 : f; }

__global__ __launch_bounds__(256) void interp2_per_edges(const real_t *__restrict__ so, real_t *ci, int IIF, int JJF,
                                                          int IIC, int JJC, int ifd, int ipn, PerIdx P)
{
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 1, jc = blockIdx.y + 1; // 1-based, every coarse index
	if (ic > IIC) return;
	// x edges: coarse rows 2..JJC-1, plus the two wrapped rows
	if (ic >= P.IBEGC && ic <= P.IENDC) {
		const int i = fine_of(ic, P.IIFC);
		if (jc >= 2 && jc <= JJC - 1) ci_xedge(so, ci, IIF, JJF, IIC, JJC, ifd, i, 2 * (jc - 1), ic, jc);
		else if (per_y(ipn) && jc == JJC) ci_xedge(so, ci, IIF, JJF, IIC, JJC, ifd, i, P.JBEG_y, ic, jc);
		else if (per_y(ipn) && jc == 1) ci_xedge(so, ci, IIF, JJF, IIC, JJC, ifd, i, P.JEND_y, ic, jc);
	}
	// y edges: coarse columns 2..IIC-1, plus the two wrapped columns
	if (jc >= P.JBEGC && jc <= P.JENDC) {
		const int j = fine_of(jc, P.JJFC);
		if (ic >= 2 && ic <= IIC - 1) ci_yedge(so, ci, IIF, JJF, IIC, JJC, ifd, 2 * (ic - 1), j, ic, jc);
		else if (per_x(ipn) && ic == IIC) ci_yedge(so, ci, IIF, JJF, IIC, JJC, ifd, P.IBEG_x, j, ic, jc);
		else if (per_x(ipn) && ic == 1) ci_yedge(so, ci, IIF, JJF, IIC, JJC, ifd, P.IEND_x, j, ic, jc);
	}
}

__global__ __launch_bounds__(256) void interp2_per_centres(const real_t *__restrict__ so, real_t *ci, int IIF, int JJF,
                                                            int IIC, int JJC, int ifd, PerIdx P)
{
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 1, jc = blockIdx.y + 1;
	if (ic < P.IBEGC || ic > P.IENDC || jc < P.JBEGC || jc > P.JENDC) return;
	ci_centre(so, ci, IIF, JJF, IIC, JJC, ifd, fine_of(ic, P.IIFC), fine_of(jc, P.JJFC), ic, jc);
}

void setup_interp2_per(const real_t *so, real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, int ipn, hipStream_t st)
{
	if (IIC < 2 || JJC < 2) return;
	const bool px = ipn == 2 || ipn == 3, py = ipn == 1 || ipn == 3;
	const int IICF = (IIF - 2) / 2 + 3, JJCF = (JJF - 2) / 2 + 3;
	PerIdx P;
	P.IIFC = 2 * (IICF - 2) + 2; P.JJFC = 2 * (JJCF - 2) + 2;
	P.IBEGC = 3; P.IENDC = IICF - 1; P.IBEG_x = 2; P.IEND_x = IIF - 2;
	P.JBEGC = 3; P.JENDC = JJCF - 1; P.JBEG_y = 2; P.JEND_y = JJF - 2;
	if (px) {
		P.IBEGC = 2;
		if (IIC == IICF) { P.IENDC = IICF; P.IBEG_x = 3; P.IEND_x = IIF - 1; }
	}
	if (py) {
		P.JBEGC = 2;
		if (JJC == JJCF) { P.JENDC = JJCF; P.JBEG_y = 3; P.JEND_y = JJF - 1; }
	}
	dim3 grid((IIC + 255) / 256, JJC);
	hipLaunchKernelGGL(interp2_per_edges, grid, dim3(256), 0, st, so, ci, IIF, JJF, IIC, JJC, ifd, ipn, P);
	hipLaunchKernelGGL(interp2_per_centres, grid, dim3(256), 0, st, so, ci, IIF, JJF, IIC, JJC, ifd, P);
}

// ------------------------------------------------------------------ coarsest grid: dense Cholesky
#define ABD(r, c) abd[(size_t)((r)-1) + (size_t)nabd1 * (size_t)((c)-1)]
#define SOC(i, j, s) so[(size_t)((i)-1) + (size_t)II * ((size_t)((j)-1) + (size_t)JJ * (size_t)(s))]
// one lane: the matrix has a few dozen rows.  Entry order of the reference (:152-212); ABD must come in zeroed.
__global__ void setup_cg2_per_kernel(const real_t *__restrict__ so, int II, int JJ, int nstncl, real_t *abd, int nabd1,
                                     int ipn, int *info)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const int I1 = II - 1, J1 = JJ - 1, I2 = I1 - 1;
	const int n = I2 * (J1 - 1);
	const bool nine = nstncl == 5;
	int kk = 1;
	ABD(1, 1) = SOC(2, 2, KO);
	for (int i = 3; i <= I1; i++) {
		kk++;
		ABD(kk, kk) = SOC(i, 2, KO);
		ABD(kk - 1, kk) = -SOC(i, 2, KW);
	}
	if (per_x(ipn)) ABD(kk - I2 + 1, kk) = -SOC(II, 2, KW);
	for (int j = 3; j <= J1; j++) {
		if (per_x(ipn)) ABD(kk, kk + 1) = nine ? -SOC(2, j, KSW) : 0.0;
		for (int i = 2; i <= I1; i++) {
			kk++;
			ABD(kk, kk) = SOC(i, j, KO);
			if (i != 2) {
				ABD(kk - 1, kk) = -SOC(i, j, KW);
				ABD(kk - I2 - 1, kk) = nine ? -SOC(i, j, KSW) : 0.0;
			}
			ABD(kk - I2 + 1, kk) = nine ? -SOC(i + 1, j, KNW) : 0.0;
			ABD(kk - I2, kk) = -SOC(i, j, KS);
		}
		if (per_x(ipn)) {
			ABD(kk - I2 + 1, kk) = -SOC(II, j, KW);
			ABD(kk - 2 * I2 + 1, kk) = nine ? -SOC(II, j, KNW) : 0.0;
		}
	}
	if (per_y(ipn)) {
		kk = kk - I2;
		const int J2 = (J1 - 2) * I2;
		kk++;
		ABD(kk - J2, kk) = -SOC(2, JJ, KS);
		ABD(kk - J2 + 1, kk) = nine ? -SOC(3, JJ, KSW) : 0.0;
		if (ipn == 3) ABD(I2, kk) = nine ? -SOC(2, JJ, KNW) : 0.0;
		for (int i = 3; i <= I1; i++) {
			kk++;
			ABD(kk - J2, kk) = -SOC(i, JJ, KS);
			ABD(kk - J2 - 1, kk) = nine ? -SOC(i, JJ, KNW) : 0.0;
			ABD(kk - J2 + 1, kk) = nine ? -SOC(i + 1, JJ, KSW) : 0.0;
		}
		ABD(kk - J2 + 1, kk) = 0.0;
		if (ipn == 3) ABD(1, kk) = nine ? -SOC(II, JJ, KSW) : 0.0;
		if (per_x(ipn)) ABD(kk - 2 * I2 + 1, kk) = nine ? -SOC(II, J1, KNW) : 0.0;
	}
	// DPOTF2 'U' (dpotf2.f): dot, sqrt, gemv, scal in the reference-BLAS order
	int rc = 0;
	for (int j = 1; j <= n && rc == 0; j++) {
		real_t dot = 0.0;
		for (int i = 1; i <= j - 1; i++) dot = dot + ABD(i, j) * ABD(i, j);
		real_t ajj = ABD(j, j) - dot;
		if (!(ajj > 0.0)) {
			ABD(j, j) = ajj;
			rc = j;
			break;
		}
		ajj = sqrt(ajj);
		ABD(j, j) = ajj;
		const real_t r = 1.0 / ajj;
		for (int c = j + 1; c <= n; c++) {
			real_t temp = 0.0;
			for (int i = 1; i <= j - 1; i++) temp = temp + ABD(i, c) * ABD(i, j);
			ABD(j, c) = ABD(j, c) + (-1.0) * temp;
			ABD(j, c) = r * ABD(j, c);
		}
	}
	*info = rc;
}

__global__ void solve_cg2_per_kernel(real_t *__restrict__ q, const real_t *__restrict__ qf, int II, int JJ,
                                     const real_t *__restrict__ abd, real_t *__restrict__ bbd, int nabd1, int ipn)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const int I1 = II - 1, J1 = JJ - 1, I2 = I1 - 1;
	const int n = I2 * (J1 - 1);
#define Q2(i, j) q[(size_t)((i)-1) + (size_t)II * (size_t)((j)-1)]
	int kk = 0;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++) bbd[kk++] = qf[(size_t)(i - 1) + (size_t)II * (size_t)(j - 1)];
	// DPOTRS 'U': inv(U^T) then inv(U) (dtrsm.f order)
	for (int i = 1; i <= n; i++) {
		real_t temp = bbd[i - 1];
		for (int k = 1; k <= i - 1; k++) temp = temp - ABD(k, i) * bbd[k - 1];
		temp = temp / ABD(i, i);
		bbd[i - 1] = temp;
	}
	for (int k = n; k >= 1; k--) {
		if (bbd[k - 1] != 0.0) {
			bbd[k - 1] = bbd[k - 1] / ABD(k, k);
			for (int i = 1; i <= k - 1; i++) bbd[i - 1] = bbd[i - 1] - bbd[k - 1] * ABD(i, k);
		}
	}
	kk = 0;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++) Q2(i, j) = bbd[kk++];
	// :125-142 the mean of the solution is removed whenever jpn != 0
	real_t cint = 0.0, qint = 0.0;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++) {
			qint = qint + Q2(i, j);
			cint = cint + 1;
		}
	const real_t c = -qint / cint;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++) Q2(i, j) = Q2(i, j) + c;
	if (per_y(ipn))
		for (int i = 2; i <= I1; i++) {
			Q2(i, JJ) = Q2(i, 2);
			Q2(i, 1) = Q2(i, J1);
		}
	if (per_x(ipn))
		for (int j = 2; j <= J1; j++) {
			Q2(II, j) = Q2(2, j);
			Q2(1, j) = Q2(I1, j);
		}
	if (ipn == 3) {
		Q2(1, 1) = Q2(I1, J1);
		Q2(II, 1) = Q2(2, J1);
		Q2(1, JJ) = Q2(I1, 2);
		Q2(II, JJ) = Q2(2, 2);
	}
#undef Q2
}
#undef ABD
#undef SOC

void setup_cg2_per(const real_t *so, int II, int JJ, int nstncl, real_t *abd, int nabd1, int ipn, int *info, hipStream_t st)
{
	hipLaunchKernelGGL(setup_cg2_per_kernel, dim3(1), dim3(64), 0, st, so, II, JJ, nstncl, abd, nabd1, ipn, info);
}

void solve_cg2_per(real_t *q, const real_t *qf, int II, int JJ, const real_t *abd, real_t *bbd, int nabd1, int ipn, hipStream_t st)
{
	hipLaunchKernelGGL(solve_cg2_per_kernel, dim3(1), dim3(64), 0, st, q, qf, II, JJ, abd, bbd, nabd1, ipn);
}

} // namespace cedar_amd
