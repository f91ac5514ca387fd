// Operator-induced interpolation set-up (BoxMG "OI" interpolation).
// Replaces BMG2_SymStd_SETUP_interp_OI (src/2d/ftn/BMG2_SymStd_SETUP_interp_OI.f90:84-256)
// and BMG3_SymStd_SETUP_interp_OI (src/3d/ftn/BMG3_SymStd_SETUP_interp_OI.f90:120-807),
// non-periodic branches.
//
// Parallel structure: one lane per *coarse* index (ic,jc[,kc]) computes the CI
// slots stored at that index.  Slots depend on lower-dimensional ones of the
// neighbouring coarse indices, so the reference's sequential phases become
// dependent launches:  2D: {x-edges, y-edges} -> {cell centres};
// 3D: {x,y,z edges} -> {xy,xz,yz faces} -> {cell centres}.
// Each phase keeps the reference's loop bounds (they reach the coarse ghost
// index IICF1 = IIC for even extents) and its exact formulas -- including the
// as-written asymmetries of the 27-point cell-centre switch (:409-441 of the 3D
// file) -- in the reference's evaluation order, so CI is bit-identical.
// Set-up runs once per solve; these kernels are simple gathers.
#include "common.h"
#include <cfloat>

namespace cedar_amd {

__device__ __forceinline__ real_t rmax(real_t a, real_t b) { return a > b ? a : b; }
__device__ __forceinline__ real_t rmin(real_t a, real_t b) { return a < b ? a : b; }
__device__ __forceinline__ real_t min4(real_t a, real_t b, real_t c, real_t d) { return rmin(rmin(a, b), rmin(c, d)); }
// S <- off + (diag - S) * max(diag - (1+ep) S, 0) / (|diag - (1+ep) S| + eps)
__device__ __forceinline__ real_t lump(real_t off, real_t diag, real_t s, real_t ep, real_t eps)
{
	return off + (diag - s) * rmax(diag - (1.0 + ep) * s, 0.0) / (fabs(diag - (1.0 + ep) * s) + eps);
}

// ======================================================================= 2D
#define SO(i, j, s) so[(size_t)((i)-1) + (size_t)IIF * ((size_t)((j)-1) + (size_t)JJF * (size_t)(s))]
#define CIW(ic, jc, s) ci[(size_t)((ic)-1) + (size_t)IIC * ((size_t)((jc)-1) + (size_t)JJC * (size_t)(s))]

__global__ __launch_bounds__(256) void interp2_edges(const real_t *__restrict__ so, real_t *ci,
        int IIF, int JJF, int IIC, int JJC, int ifd, int ilo, int jlo)
{
	const real_t zeps = DBL_EPSILON;
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 2, jc = blockIdx.y + 2; // 1-based
	const int IIC1 = IIC - 1, JJC1 = JJC - 1;
	const int IICF1 = (IIF - 2) / 2 + 2, JJCF1 = (JJF - 2) / 2 + 2;
	(void)IIC1; (void)JJC1; (void)IICF1; (void)JJCF1;
	const int i = 2 * (ic - 1), j = 2 * (jc - 1);
	real_t a, b, ep, sum, s;
	(void)a; (void)b; (void)s;
	if (ic >= ilo && ic <= IICF1 && jc >= 2 && jc <= JJC1) {

	if (ifd != 1) {
		a = SO(i, j, KW) + SO(i, j, KNW) + SO(i, j + 1, KSW);
		b = SO(i - 1, j, KW) + SO(i - 1, j, KSW) + SO(i - 1, j + 1, KNW);
	} else {
		a = SO(i, j, KW);
		b = SO(i - 1, j, KW);
	}
	ep = rmin(fabs(a / SO(i - 1, j, KO)), fabs(b / SO(i - 1, j, KO)));
	sum = a + b + SO(i - 1, j, KS) + SO(i - 1, j + 1, KS);
	sum = lump(a + b, SO(i - 1, j, KO), sum, ep, zeps);
	sum = 1.0 / sum;
	CIW(ic, jc, LR) = a * sum;
	CIW(ic, jc, LL) = b * sum;
	}
	if (ic >= 2 && ic <= IIC1 && jc >= jlo && jc <= JJCF1) {

	if (ifd != 1) {
		a = SO(i, j, KS) + SO(i, j, KNW) + SO(i + 1, j, KSW);
		b = SO(i, j - 1, KS) + SO(i, j - 1, KSW) + SO(i + 1, j - 1, KNW);
	} else {
		a = SO(i, j, KS);
		b = SO(i, j - 1, KS);
	}
	ep = rmin(fabs(a / SO(i, j - 1, KO)), fabs(b / SO(i, j - 1, KO)));
	sum = a + b + SO(i, j - 1, KW) + SO(i + 1, j - 1, KW);
	sum = lump(a + b, SO(i, j - 1, KO), sum, ep, zeps);
	sum = 1.0 / sum;
	CIW(ic, jc, LA) = a * sum;
	CIW(ic, jc, LB) = b * sum;
	}
}

__global__ __launch_bounds__(256) void interp2_centres(const real_t *__restrict__ so, real_t *ci,
        int IIF, int JJF, int IIC, int JJC, int ifd, int ilo, int jlo)
{
	const real_t zeps = DBL_EPSILON;
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 2, jc = blockIdx.y + 2; // 1-based
	const int IIC1 = IIC - 1, JJC1 = JJC - 1;
	const int IICF1 = (IIF - 2) / 2 + 2, JJCF1 = (JJF - 2) / 2 + 2;
	(void)IIC1; (void)JJC1; (void)IICF1; (void)JJCF1;
	const int i = 2 * (ic - 1), j = 2 * (jc - 1);
	real_t a, b, ep, sum, s;
	(void)a; (void)b; (void)s;
	if (ic >= ilo && ic <= IICF1 && jc >= jlo && jc <= JJCF1) {

	real_t d = SO(i - 1, j - 1, KO);
	if (ifd != 1) {
		sum = SO(i - 1, j - 1, KW) + SO(i - 1, j, KNW) + SO(i - 1, j, KS)
		      + SO(i, j, KSW) + SO(i, j - 1, KW) + SO(i, j - 1, KNW)
		      + SO(i - 1, j - 1, KS) + SO(i - 1, j - 1, KSW);
		ep = rmin(rmin(fabs((SO(i - 1, j - 1, KSW) + SO(i - 1, j - 1, KW)
		                     + SO(i - 1, j, KNW)) / d),
		               fabs((SO(i - 1, j, KNW) + SO(i - 1, j, KS)
		                     + SO(i, j, KSW)) / d)),
		          rmin(fabs((SO(i, j, KSW) + SO(i, j - 1, KW)
		                     + SO(i, j - 1, KNW)) / d),
		               fabs((SO(i, j - 1, KNW) + SO(i - 1, j - 1, KS)
		                     + SO(i - 1, j - 1, KSW)) / d)));
		sum = lump(sum, d, sum, ep, zeps);
		s = 1.0 / sum;
		CIW(ic, jc, LSW) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LL)
		                    + SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LB)
		                    + SO(i - 1, j - 1, KSW)) * s;
		CIW(ic, jc, LSE) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LR)
		                    + SO(i, j - 1, KW) * CIW(ic, jc, LB)
		                    + SO(i, j - 1, KNW)) * s;
		CIW(ic, jc, LNW) = (SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LA)
		                    + SO(i - 1, j, KS) * CIW(ic, jc, LL)
		                    + SO(i - 1, j, KNW)) * s;
		CIW(ic, jc, LNE) = (SO(i - 1, j, KS) * CIW(ic, jc, LR)
		                    + SO(i, j - 1, KW) * CIW(ic, jc, LA)
		                    + SO(i, j, KSW)) * s;
	} else {
		sum = SO(i - 1, j - 1, KW) + SO(i - 1, j, KS) + SO(i, j - 1, KW)
		      + SO(i - 1, j - 1, KS);
		ep = rmin(rmin(fabs(SO(i - 1, j - 1, KW) / d), fabs(SO(i - 1, j, KS) / d)),
		          rmin(fabs(SO(i, j - 1, KW) / d), fabs(SO(i - 1, j - 1, KS) / d)));
		sum = lump(sum, d, sum, ep, zeps);
		s = 1.0 / sum;
		CIW(ic, jc, LSW) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LL)
		                    + SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LB)) * s;
		CIW(ic, jc, LSE) = (SO(i - 1, j - 1, KS) * CIW(ic, jc - 1, LR)
		                    + SO(i, j - 1, KW) * CIW(ic, jc, LB)) * s;
		CIW(ic, jc, LNW) = (SO(i - 1, j - 1, KW) * CIW(ic - 1, jc, LA)
		                    + SO(i - 1, j, KS) * CIW(ic, jc, LL)) * s;
		CIW(ic, jc, LNE) = (SO(i - 1, j, KS) * CIW(ic, jc, LR)
		                    + SO(i, j - 1, KW) * CIW(ic, jc, LA)) * s;
	}
	}
}

// phase 0 = the two edge families, 1 = the cell centres; ilo / jlo = 3 (serial / physical boundary) or 2
// (a neighbouring subdomain owns coarse index 1), as in setup_interp3_phase
void setup_interp2_phase(const real_t *so, real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, int phase,
                         int ilo, int jlo, hipStream_t st)
{
	if (IIC < 2 || JJC < 2) return;
	dim3 grid((IIC - 1 + 255) / 256, JJC - 1); // ic, jc in [2, IIC], [2, JJC]
	if (phase == 0) hipLaunchKernelGGL(interp2_edges, grid, dim3(256), 0, st, so, ci, IIF, JJF, IIC, JJC, ifd, ilo, jlo);
	else hipLaunchKernelGGL(interp2_centres, grid, dim3(256), 0, st, so, ci, IIF, JJF, IIC, JJC, ifd, ilo, jlo);
}

void setup_interp2(const real_t *so, real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, hipStream_t st)
{
	setup_interp2_phase(so, ci, IIF, JJF, IIC, JJC, ifd, 0, 3, 3, st);
	setup_interp2_phase(so, ci, IIF, JJF, IIC, JJC, ifd, 1, 3, 3, st);
}
#undef SO
#undef CIW

// ======================================================================= 3D
#define SO(i, j, k, s) so[(size_t)((i)-1) + (size_t)IIF * ((size_t)((j)-1) + (size_t)JJF * ((size_t)((k)-1) + (size_t)KKF * (size_t)(s)))]
#define CW(ic, jc, kc, s) ci[(size_t)((ic)-1) + (size_t)IIC * ((size_t)((jc)-1) + (size_t)JJC * ((size_t)((kc)-1) + (size_t)KKC * (size_t)(s)))]

__global__ __launch_bounds__(128) void interp3_edges(const real_t *__restrict__ so, real_t *ci,
        int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int ifd, int ilo, int jlo, int klo)
{
	const real_t eMACH = 1.e-13;
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 2, jc = blockIdx.y + 2, kc = blockIdx.z + 2; // 1-based
	if (ic > IIC) return;
	const int iic1 = IIC - 1, jjc1 = JJC - 1, kkc1 = KKC - 1;
	const int iicf1 = (IIF - 2) / 2 + 2, jjcf1 = (JJF - 2) / 2 + 2, kkcf1 = (KKF - 2) / 2 + 2;
	(void)iic1; (void)jjc1; (void)kkc1; (void)iicf1; (void)jjcf1; (void)kkcf1;
	const int i = 2 * (ic - 1), j = 2 * (jc - 1), k = 2 * (kc - 1);
	real_t a, b, c, ep, dnw, dn, dne, dw, de, dsw, ds, dse, dp, sum;
	(void)a; (void)b; (void)c; (void)ep; (void)dnw; (void)dn; (void)dne; (void)dw; (void)de;
	(void)dsw; (void)ds; (void)dse; (void)dp; (void)sum;
	if (kc <= kkc1 && jc <= jjc1 && ic >= ilo && ic <= iicf1) {

	real_t d = SO(i - 1, j, k, KP);
	if (ifd != 1) {
		a = SO(i - 1, j + 1, k, KPNW) + SO(i - 1, j, k, KPW)
		    + SO(i - 1, j, k, KPSW)
		    + SO(i - 1, j + 1, k, KBNW) + SO(i - 1, j, k, KBW)
		    + SO(i - 1, j, k, KBSW) + SO(i - 1, j + 1, k + 1, KBSE)
		    + SO(i - 1, j, k + 1, KBE) + SO(i - 1, j, k + 1, KBNE);
		b = SO(i, j + 1, k, KPSW) + SO(i, j, k, KPW) + SO(i, j, k, KPNW)
		    + SO(i, j + 1, k, KBNE) + SO(i, j, k, KBE) + SO(i, j, k, KBSE)
		    + SO(i, j + 1, k + 1, KBSW) + SO(i, j, k + 1, KBW)
		    + SO(i, j, k + 1, KBNW);
		c = a + b + SO(i - 1, j, k, KPS) + SO(i - 1, j + 1, k, KPS)
		    + SO(i - 1, j + 1, k, KBN) + SO(i - 1, j, k, KB)
		    + SO(i - 1, j, k, KBS)
		    + SO(i - 1, j + 1, k + 1, KBS) + SO(i - 1, j, k + 1, KB)
		    + SO(i - 1, j, k + 1, KBN);
		ep = rmin(fabs(a / d), fabs(b / d));
	} else {
		a = SO(i - 1, j, k, KPW);
		b = SO(i, j, k, KPW);
		ep = rmin(fabs(a / d), fabs(b) / d);
		c = a + b + SO(i - 1, j, k, KPS) + SO(i - 1, j + 1, k, KPS)
		    + SO(i - 1, j, k, KB) + SO(i - 1, j, k + 1, KB);
	}
	c = lump(a + b, d, c, ep, eMACH);
	CW(ic, jc, kc, LXYL) = a / c;
	CW(ic, jc, kc, LXYR) = b / c;
	}
	if (kc <= kkc1 && jc >= jlo && jc <= jjcf1 && ic <= iic1) {

	real_t d = SO(i, j - 1, k, KP);
	if (ifd != 1) {
		a = SO(i, j, k, KPNW) + SO(i, j, k, KPS) + SO(i + 1, j, k, KPSW)
		    + SO(i, j, k, KBNW) + SO(i, j, k, KBN) + SO(i + 1, j, k, KBNE)
		    + SO(i, j, k + 1, KBSE) + SO(i, j, k + 1, KBS)
		    + SO(i + 1, j, k + 1, KBSW);
		b = SO(i, j - 1, k, KPSW) + SO(i, j - 1, k, KPS)
		    + SO(i + 1, j - 1, k, KPNW)
		    + SO(i, j - 1, k, KBSW) + SO(i, j - 1, k, KBS)
		    + SO(i + 1, j - 1, k, KBSE) + SO(i, j - 1, k + 1, KBNE)
		    + SO(i, j - 1, k + 1, KBN) + SO(i + 1, j - 1, k + 1, KBNW);
		ep = rmin(fabs(a / d), fabs(b / d));
		c = a + b + SO(i, j - 1, k, KPW) + SO(i + 1, j - 1, k, KPW)
		    + SO(i, j - 1, k, KBW) + SO(i, j - 1, k, KB)
		    + SO(i + 1, j - 1, k, KBE)
		    + SO(i, j - 1, k + 1, KBE) + SO(i, j - 1, k + 1, KB)
		    + SO(i + 1, j - 1, k + 1, KBW);
	} else {
		a = SO(i, j, k, KPS);
		b = SO(i, j - 1, k, KPS);
		c = a + b + SO(i, j - 1, k, KPW) + SO(i + 1, j - 1, k, KPW)
		    + SO(i, j - 1, k, KB) + SO(i, j - 1, k + 1, KB);
		ep = rmin(fabs(a / d), fabs(b / d));
	}
	c = lump(a + b, d, c, ep, eMACH);
	CW(ic, jc, kc, LXYA) = a / c;
	CW(ic, jc, kc, LXYB) = b / c;
	}
	if (kc >= klo && kc <= kkcf1 && jc <= jjc1 && ic <= iic1) {

	real_t d = SO(i, j, k - 1, KP);
	if (ifd != 1) {
		a = SO(i, j + 1, k, KBSE) + SO(i, j + 1, k, KBS)
		    + SO(i + 1, j + 1, k, KBSW)
		    + SO(i, j, k, KBE) + SO(i, j, k, KB) + SO(i + 1, j, k, KBW)
		    + SO(i, j, k, KBNE) + SO(i, j, k, KBN) + SO(i + 1, j, k, KBNW);
		b = SO(i, j + 1, k - 1, KBNW) + SO(i, j + 1, k - 1, KBN)
		    + SO(i + 1, j + 1, k - 1, KBNE) + SO(i, j, k - 1, KBW)
		    + SO(i, j, k - 1, KB) + SO(i + 1, j, k - 1, KBE)
		    + SO(i, j, k - 1, KBSW) + SO(i, j, k - 1, KBS)
		    + SO(i + 1, j, k - 1, KBSE);
		c = a + b + SO(i, j, k - 1, KPW) + SO(i + 1, j, k - 1, KPW)
		    + SO(i, j + 1, k - 1, KPNW) + SO(i, j + 1, k - 1, KPS)
		    + SO(i + 1, j + 1, k - 1, KPSW) + SO(i, j, k - 1, KPSW)
		    + SO(i, j, k - 1, KPS) + SO(i + 1, j, k - 1, KPNW);
	} else {
		a = SO(i, j, k, KB);
		b = SO(i, j, k - 1, KB);
		c = a + b + SO(i, j, k - 1, KPW) + SO(i + 1, j, k - 1, KPW)
		    + SO(i, j + 1, k - 1, KPS) + SO(i, j, k - 1, KPS);
	}
	ep = rmin(fabs(a / d), fabs(b / d));
	c = lump(a + b, d, c, ep, eMACH);
	CW(ic, jc, kc, LXZA) = a / c;
	CW(ic, jc, kc, LXZB) = b / c;
	}
}

__global__ __launch_bounds__(128) void interp3_faces(const real_t *__restrict__ so, real_t *ci,
        int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int ifd, int ilo, int jlo, int klo)
{
	const real_t eMACH = 1.e-13;
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 2, jc = blockIdx.y + 2, kc = blockIdx.z + 2; // 1-based
	if (ic > IIC) return;
	const int iic1 = IIC - 1, jjc1 = JJC - 1, kkc1 = KKC - 1;
	const int iicf1 = (IIF - 2) / 2 + 2, jjcf1 = (JJF - 2) / 2 + 2, kkcf1 = (KKF - 2) / 2 + 2;
	(void)iic1; (void)jjc1; (void)kkc1; (void)iicf1; (void)jjcf1; (void)kkcf1;
	const int i = 2 * (ic - 1), j = 2 * (jc - 1), k = 2 * (kc - 1);
	real_t a, b, c, ep, dnw, dn, dne, dw, de, dsw, ds, dse, dp, sum;
	(void)a; (void)b; (void)c; (void)ep; (void)dnw; (void)dn; (void)dne; (void)dw; (void)de;
	(void)dsw; (void)ds; (void)dse; (void)dp; (void)sum;
	if (kc <= kkc1 && jc >= jlo && jc <= jjcf1 && ic >= ilo && ic <= iicf1) {

	real_t d = SO(i - 1, j - 1, k, KP);
	if (ifd != 1) {
		dnw = SO(i - 1, j, k, KPNW) + SO(i - 1, j, k, KBNW)
		      + SO(i - 1, j, k + 1, KBSE);
		dn = SO(i - 1, j, k, KPS) + SO(i - 1, j, k, KBN)
		     + SO(i - 1, j, k + 1, KBS);
		dne = SO(i, j, k, KPSW) + SO(i, j, k, KBNE) + SO(i, j, k + 1, KBSW);
		dw = SO(i - 1, j - 1, k, KPW) + SO(i - 1, j - 1, k, KBW)
		     + SO(i - 1, j - 1, k + 1, KBE);
		de = SO(i, j - 1, k, KPW) + SO(i, j - 1, k, KBE)
		     + SO(i, j - 1, k + 1, KBW);
		dsw = SO(i - 1, j - 1, k, KPSW) + SO(i - 1, j - 1, k, KBSW)
		      + SO(i - 1, j - 1, k + 1, KBNE);
		ds = SO(i - 1, j - 1, k, KPS) + SO(i - 1, j - 1, k, KBS)
		     + SO(i - 1, j - 1, k + 1, KBN);
		dse = SO(i, j - 1, k, KPNW) + SO(i, j - 1, k, KBSE)
		      + SO(i, j - 1, k + 1, KBNW);
		ep = min4(fabs((dsw + dw + dnw) / d), fabs((dnw + dn + dne) / d),
		          fabs((dne + de + dse) / d), fabs((dse + ds + dsw) / d));
		dp = dw + dnw + dn + dne + de + dse + ds + dsw;
	} else {
		dn = SO(i - 1, j, k, KPS);
		dw = SO(i - 1, j - 1, k, KPW);
		de = SO(i, j - 1, k, KPW);
		ds = SO(i - 1, j - 1, k, KPS);
		dnw = dne = dsw = dse = 0.0;
		dp = dw + dn + de + ds;
		ep = min4(fabs(dw / d), fabs(dn / d), fabs(de / d), fabs(ds / d));
	}
	sum = d - SO(i - 1, j - 1, k, KB) - SO(i - 1, j - 1, k + 1, KB);
	dp = lump(dp, sum, dp, ep, eMACH);
	dp = 1.0 / dp;
	if (ifd != 1) {
		CW(ic, jc, kc, LXYNW) = dp * (dnw + CW(ic - 1, jc, kc, LXYA) * dw
		                              + CW(ic, jc, kc, LXYL) * dn);
		CW(ic, jc, kc, LXYNE) = dp * (dne + CW(ic, jc, kc, LXYR) * dn
		                              + CW(ic, jc, kc, LXYA) * de);
		CW(ic, jc, kc, LXYSE) = dp * (dse + CW(ic, jc, kc, LXYB) * de
		                              + CW(ic, jc - 1, kc, LXYR) * ds);
		CW(ic, jc, kc, LXYSW) = dp * (dsw + CW(ic, jc - 1, kc, LXYL) * ds
		                              + CW(ic - 1, jc, kc, LXYB) * dw);
	} else {
		CW(ic, jc, kc, LXYNW) = dp * (CW(ic - 1, jc, kc, LXYA) * dw
		                              + CW(ic, jc, kc, LXYL) * dn);
		CW(ic, jc, kc, LXYNE) = dp * (CW(ic, jc, kc, LXYR) * dn
		                              + CW(ic, jc, kc, LXYA) * de);
		CW(ic, jc, kc, LXYSE) = dp * (CW(ic, jc, kc, LXYB) * de
		                              + CW(ic, jc - 1, kc, LXYR) * ds);
		CW(ic, jc, kc, LXYSW) = dp * (CW(ic, jc - 1, kc, LXYL) * ds
		                              + CW(ic - 1, jc, kc, LXYB) * dw);
	}
	}
	if (kc >= klo && kc <= kkcf1 && jc <= jjc1 && ic >= ilo && ic <= iicf1) {

	real_t d = SO(i - 1, j, k - 1, KP);
	if (ifd != 1) {
		dnw = SO(i - 1, j + 1, k, KBSE) + SO(i - 1, j, k, KBE)
		      + SO(i - 1, j, k, KBNE);
		dn = SO(i - 1, j + 1, k, KBS) + SO(i - 1, j, k, KB) + SO(i - 1, j, k, KBN);
		dne = SO(i, j + 1, k, KBSW) + SO(i, j, k, KBW) + SO(i, j, k, KBNW);
		dw = SO(i - 1, j + 1, k - 1, KPNW) + SO(i - 1, j, k - 1, KPW)
		     + SO(i - 1, j, k - 1, KPSW);
		de = SO(i, j + 1, k - 1, KPSW) + SO(i, j, k - 1, KPW)
		     + SO(i, j, k - 1, KPNW);
		dsw = SO(i - 1, j + 1, k - 1, KBNW) + SO(i - 1, j, k - 1, KBW)
		      + SO(i - 1, j, k - 1, KBSW);
		ds = SO(i - 1, j + 1, k - 1, KBN) + SO(i - 1, j, k - 1, KB)
		     + SO(i - 1, j, k - 1, KBS);
		dse = SO(i, j + 1, k - 1, KBNE) + SO(i, j, k - 1, KBE)
		      + SO(i, j, k - 1, KBSE);
		ep = min4(fabs((dsw + dw + dnw) / d), fabs((dnw + dn + dne) / d),
		          fabs((dne + de + dse) / d), fabs((dse + ds + dsw) / d));
		dp = dw + dnw + dn + dne + de + dse + ds + dsw;
	} else {
		dn = SO(i - 1, j, k, KB);
		dw = SO(i - 1, j, k - 1, KPW);
		de = SO(i, j, k - 1, KPW);
		ds = SO(i - 1, j, k - 1, KB);
		dnw = dne = dsw = dse = 0.0;
		dp = dw + dn + de + ds;
		ep = min4(fabs(dw / d), fabs(dn / d), fabs(de / d), fabs(ds / d));
	}
	sum = d - SO(i - 1, j + 1, k - 1, KPS) - SO(i - 1, j, k - 1, KPS);
	dp = lump(dp, sum, dp, ep, eMACH);
	dp = 1.0 / dp;
	if (ifd != 1) {
		CW(ic, jc, kc, LXZNW) = dp * (dnw + CW(ic - 1, jc, kc, LXZA) * dw
		                              + CW(ic, jc, kc, LXYL) * dn);
		CW(ic, jc, kc, LXZNE) = dp * (dne + CW(ic, jc, kc, LXYR) * dn
		                              + CW(ic, jc, kc, LXZA) * de);
		CW(ic, jc, kc, LXZSE) = dp * (dse + CW(ic, jc, kc, LXZB) * de
		                              + CW(ic, jc, kc - 1, LXYR) * ds);
		CW(ic, jc, kc, LXZSW) = dp * (dsw + CW(ic, jc, kc - 1, LXYL) * ds
		                              + CW(ic - 1, jc, kc, LXZB) * dw);
	} else {
		CW(ic, jc, kc, LXZNW) = dp * (CW(ic - 1, jc, kc, LXZA) * dw
		                              + CW(ic, jc, kc, LXYL) * dn);
		CW(ic, jc, kc, LXZNE) = dp * (CW(ic, jc, kc, LXYR) * dn
		                              + CW(ic, jc, kc, LXZA) * de);
		CW(ic, jc, kc, LXZSE) = dp * (CW(ic, jc, kc, LXZB) * de
		                              + CW(ic, jc, kc - 1, LXYR) * ds);
		CW(ic, jc, kc, LXZSW) = dp * (CW(ic, jc, kc - 1, LXYL) * ds
		                              + CW(ic - 1, jc, kc, LXZB) * dw);
	}
	}
	if (kc >= klo && kc <= kkcf1 && jc >= jlo && jc <= jjcf1 && ic <= iic1) {

	real_t d = SO(i, j - 1, k - 1, KP);
	if (ifd != 1) {
		dnw = SO(i, j, k, KBSE) + SO(i, j, k, KBS) + SO(i + 1, j, k, KBSW);
		dn = SO(i, j - 1, k, KBE) + SO(i, j - 1, k, KB) + SO(i + 1, j - 1, k, KBW);
		dne = SO(i, j - 1, k, KBNE) + SO(i, j - 1, k, KBN)
		      + SO(i + 1, j - 1, k, KBNW);
		dw = SO(i, j, k - 1, KPNW) + SO(i, j, k - 1, KPS)
		     + SO(i + 1, j, k - 1, KPSW);
		de = SO(i, j - 1, k - 1, KPSW) + SO(i, j - 1, k - 1, KPS)
		     + SO(i + 1, j - 1, k - 1, KPNW);
		dsw = SO(i, j, k - 1, KBNW) + SO(i, j, k - 1, KBN)
		      + SO(i + 1, j, k - 1, KBNE);
		ds = SO(i, j - 1, k - 1, KBW) + SO(i, j - 1, k - 1, KB)
		     + SO(i + 1, j - 1, k - 1, KBE);
		dse = SO(i, j - 1, k - 1, KBSW) + SO(i, j - 1, k - 1, KBS)
		      + SO(i + 1, j - 1, k - 1, KBSE);
		ep = min4(fabs((dsw + dw + dnw) / d), fabs((dnw + dn + dne) / d),
		          fabs((dne + de + dse) / d), fabs((dse + ds + dsw) / d));
		dp = dw + dnw + dn + dne + de + dse + ds + dsw;
	} else {
		dn = SO(i, j - 1, k, KB);
		dw = SO(i, j, k - 1, KPS);
		de = SO(i, j - 1, k - 1, KPS);
		ds = SO(i, j - 1, k - 1, KB);
		dnw = dne = dsw = dse = 0.0;
		dp = dw + dn + de + ds;
		ep = min4(fabs(dw / d), fabs(dn / d), fabs(de / d), fabs(ds / d));
	}
	sum = d - SO(i, j - 1, k - 1, KPW) - SO(i + 1, j - 1, k - 1, KPW);
	dp = lump(dp, sum, dp, ep, eMACH);
	dp = 1.0 / dp;
	if (ifd != 1) {
		CW(ic, jc, kc, LYZNW) = dp * (dnw + CW(ic, jc, kc, LXZA) * dw
		                              + CW(ic, jc, kc, LXYA) * dn);
		CW(ic, jc, kc, LYZNE) = dp * (dne + CW(ic, jc, kc, LXYB) * dn
		                              + CW(ic, jc - 1, kc, LXZA) * de);
		CW(ic, jc, kc, LYZSE) = dp * (dse + CW(ic, jc - 1, kc, LXZB) * de
		                              + CW(ic, jc, kc - 1, LXYB) * ds);
		CW(ic, jc, kc, LYZSW) = dp * (dsw + CW(ic, jc, kc - 1, LXYA) * ds
		                              + CW(ic, jc, kc, LXZB) * dw);
	} else {
		CW(ic, jc, kc, LYZNW) = dp * (CW(ic, jc, kc, LXZA) * dw
		                              + CW(ic, jc, kc, LXYA) * dn);
		CW(ic, jc, kc, LYZNE) = dp * (CW(ic, jc, kc, LXYB) * dn
		                              + CW(ic, jc - 1, kc, LXZA) * de);
		CW(ic, jc, kc, LYZSE) = dp * (CW(ic, jc - 1, kc, LXZB) * de
		                              + CW(ic, jc, kc - 1, LXYB) * ds);
		CW(ic, jc, kc, LYZSW) = dp * (CW(ic, jc, kc - 1, LXYA) * ds
		                              + CW(ic, jc, kc, LXZB) * dw);
	}
	}
}

__global__ __launch_bounds__(128) void interp3_centres(const real_t *__restrict__ so, real_t *ci,
        int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int ifd, int ilo, int jlo, int klo)
{
	const real_t eMACH = 1.e-13;
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 2, jc = blockIdx.y + 2, kc = blockIdx.z + 2; // 1-based
	if (ic > IIC) return;
	const int iic1 = IIC - 1, jjc1 = JJC - 1, kkc1 = KKC - 1;
	const int iicf1 = (IIF - 2) / 2 + 2, jjcf1 = (JJF - 2) / 2 + 2, kkcf1 = (KKF - 2) / 2 + 2;
	(void)iic1; (void)jjc1; (void)kkc1; (void)iicf1; (void)jjcf1; (void)kkcf1;
	const int i = 2 * (ic - 1), j = 2 * (jc - 1), k = 2 * (kc - 1);
	real_t a, b, c, ep, dnw, dn, dne, dw, de, dsw, ds, dse, dp, sum;
	(void)a; (void)b; (void)c; (void)ep; (void)dnw; (void)dn; (void)dne; (void)dw; (void)de;
	(void)dsw; (void)ds; (void)dse; (void)dp; (void)sum;
	if (kc >= klo && kc <= kkcf1 && jc >= jlo && jc <= jjcf1 && ic >= ilo && ic <= iicf1) {

	real_t d = SO(i - 1, j - 1, k - 1, KP);
	real_t yp, yw;
	if (ifd != 1) {
		yp = SO(i - 1, j - 1, k - 1, KPW)
		     + SO(i - 1, j, k - 1, KPNW)
		     + SO(i - 1, j, k - 1, KPS) + SO(i, j, k - 1, KPSW)
		     + SO(i, j - 1, k - 1, KPW)
		     + SO(i, j - 1, k - 1, KPNW) + SO(i - 1, j - 1, k - 1, KPS)
		     + SO(i - 1, j - 1, k - 1, KPSW) + SO(i - 1, j - 1, k - 1, KB)
		     + SO(i - 1, j - 1, k - 1, KBW) + SO(i - 1, j, k - 1, KBNW)
		     + SO(i - 1, j, k - 1, KBN) + SO(i, j, k - 1, KBNE)
		     + SO(i, j - 1, k - 1, KBE)
		     + SO(i, j - 1, k - 1, KBSE) + SO(i - 1, j - 1, k - 1, KBS)
		     + SO(i - 1, j - 1, k - 1, KBSW) + SO(i - 1, j - 1, k, KB)
		     + SO(i - 1, j - 1, k, KBE) + SO(i - 1, j, k, KBSE)
		     + SO(i - 1, j, k, KBS)
		     + SO(i, j, k, KBSW) + SO(i, j - 1, k, KBW)
		     + SO(i, j - 1, k, KBNW)
		     + SO(i - 1, j - 1, k, KBN) + SO(i - 1, j - 1, k, KBNE);
		yw = min4(fabs(SO(i - 1, j - 1, k - 1, KPW)
		               + SO(i - 1, j, k - 1, KPNW)
		               + SO(i - 1, j, k, KBSE) + SO(i - 1, j - 1, k, KBE)
		               + SO(i - 1, j - 1, k, KBNE)
		               + SO(i - 1, j - 1, k - 1, KPSW) + SO(i - 1, j - 1, k - 1, KBSW)
		               + SO(i - 1, j - 1, k - 1, KBW) + SO(i - 1, j, k - 1, KBNW))
		              / d,
		          fabs(SO(i, j - 1, k - 1, KPW) + SO(i, j, k - 1, KPSW)
		               + SO(i, j, k, KBSW) + SO(i, j - 1, k, KBW)
		               + SO(i, j - 1, k, KBNW) + SO(i, j - 1, k - 1, KPNW)
		               + SO(i, j - 1, k - 1, KBSE)
		               + SO(i, j - 1, k - 1, KBE) + SO(i, j, k - 1, KBNE))
		              / d,
		          fabs(SO(i - 1, j, k - 1, KPS) + SO(i - 1, j, k - 1, KPNW)
		               + SO(i - 1, j, k, KBSE) + SO(i - 1, j, k, KBS)
		               + SO(i, j, k, KBSW)
		               + SO(i, j, k - 1, KPSW) + SO(i, j, k - 1, KBNE)
		               + SO(i - 1, j, k - 1, KBN)
		               + SO(i - 1, j, k - 1, KBNW)),
		          fabs(SO(i - 1, j - 1, k - 1, KPS)
		               + SO(i - 1, j - 1, k - 1, KPSW) + SO(i - 1, j - 1, k, KBNE)
		               + SO(i - 1, j - 1, k, KBN) + SO(i, j - 1, k, KBNW)
		               + SO(i, j - 1, k - 1, KPNW) + SO(i, j - 1, k - 1, KBSE)
		               + SO(i - 1, j - 1, k - 1, KBS) + SO(i, j - 1, k - 1, KBSE))
		              / d);
		yw = rmin(yw,
		          rmin(fabs(SO(i - 1, j - 1, k - 1, KB) + SO(i - 1, j - 1, k - 1, KBW)
		                    + SO(i - 1, j, k - 1, KBNW) + SO(i - 1, j, k - 1, KBN)
		                    + SO(i, j, k - 1, KBNE)
		                    + SO(i, j - 1, k - 1, KBE) + SO(i, j - 1, k - 1, KBSE)
		                    + SO(i - 1, j - 1, k - 1, KBS)
		                    + SO(i - 1, j - 1, k - 1, KBSW)),
		               fabs(SO(i - 1, j - 1, k, KB)
		                    + SO(i - 1, j - 1, k, KBE) + SO(i - 1, j, k, KBSE)
		                    + SO(i - 1, j, k, KBS)
		                    + SO(i, j, k, KBSW) + SO(i, j - 1, k, KBW)
		                    + SO(i, j - 1, k, KBNW)
		                    + SO(i - 1, j - 1, k, KBN) + SO(i - 1, j - 1, k, KBNE))
		                   / d));
		yp = lump(yp, d, yp, yw, eMACH);
		yp = 1.0 / yp;
		CW(ic, jc, kc, LTNW)
		    = yp * (SO(i - 1, j, k, KBSE)
		            + CW(ic - 1, jc, kc, LYZNW)
		              * SO(i - 1, j - 1, k - 1, KPW) + CW(ic - 1, jc, kc, LXZA)
		              * SO(i - 1, j, k - 1, KPNW)
		            + CW(ic, jc, kc, LXZNW) * SO(i - 1, j, k - 1, KPS)
		            + CW(ic - 1, jc, kc, LXYA)
		              * SO(i - 1, j - 1, k, KBE) + CW(ic, jc, kc, LXYL)
		              * SO(i - 1, j, k, KBS)
		            + CW(ic, jc, kc, LXYNW) * SO(i - 1, j - 1, k, KB));
		CW(ic, jc, kc, LTNE)
		    = yp * (SO(i, j, k, KBSW)
		            + CW(ic, jc, kc, LXZNE)
		              * SO(i - 1, j, k - 1, KPS) + CW(ic, jc, kc, LXZA)
		              * SO(i, j, k - 1, KPSW)
		            + CW(ic, jc, kc, LYZNW) * SO(i, j - 1, k - 1, KPW)
		            + CW(ic, jc, kc, LXYR)
		              * SO(i - 1, j, k, KBS) + CW(ic, jc, kc, LXYA)
		              * SO(i, j - 1, k, KBW)
		            + CW(ic, jc, kc, LXYNE) * SO(i - 1, j - 1, k, KB));
		CW(ic, jc, kc, LBNW)
		    = yp * (SO(i - 1, j, k - 1, KBNW)
		            + CW(ic - 1, jc, kc - 1, LXYA) * SO(i - 1, j - 1, k - 1, KBW)
		            + CW(ic, jc, kc - 1, LXYL) * SO(i - 1, j, k - 1, KBN)
		            + CW(ic, jc, kc - 1, LXYNW) * SO(i - 1, j - 1, k - 1, KB)
		            + CW(ic - 1, jc, kc, LYZSW) * SO(i - 1, j - 1, k - 1, KPW)
		            + CW(ic - 1, jc, kc, LXZB) * SO(i - 1, j, k - 1, KPNW)
		            + CW(ic, jc, kc, LXZSW) * SO(i - 1, j, k - 1, KPS));
		CW(ic, jc, kc, LBNE)
		    = yp * (SO(i, j, k - 1, KBNE)
		            + CW(ic, jc, kc - 1, LXYNE)
		              * SO(i - 1, j - 1, k - 1, KB) + CW(ic, jc, kc - 1, LXYR)
		              * SO(i - 1, j, k - 1, KBN)
		            + CW(ic, jc, kc - 1, LXYA) * SO(i, j - 1, k - 1, KBE)
		            + CW(ic, jc, kc, LXZSE)
		              * SO(i - 1, j, k - 1, KPS) + CW(ic, jc, kc, LXZB)
		              * SO(i, j, k - 1, KPSW)
		            + CW(ic, jc, kc, LYZSW) * SO(i, j - 1, k - 1, KPW));
		CW(ic, jc, kc, LBSW)
		    = yp * (SO(i - 1, j - 1, k - 1, KBSW)
		            + CW(ic - 1, jc, kc - 1, LXYB) * SO(i - 1, j - 1, k - 1, KBW)
		            + CW(ic, jc, kc - 1, LXYSW)
		              * SO(i - 1, j - 1, k - 1, KB) + CW(ic, jc - 1, kc - 1, LXYL)
		              * SO(i - 1, j - 1, k - 1, KBS) + CW(ic - 1, jc, kc, LYZSE)
		              * SO(i - 1, j - 1, k - 1, KPW)
		            + CW(ic, jc - 1, kc, LXZSW) * SO(i - 1, j - 1, k - 1, KPS)
		            + CW(ic - 1, jc - 1, kc, LXZB) * SO(i - 1, j - 1, k - 1, KPSW));
		CW(ic, jc, kc, LTSW)
		    = yp * (SO(i - 1, j - 1, k, KBNE)
		            + CW(ic - 1, jc, kc, LXYB) * SO(i - 1, j - 1, k, KBE)
		            + CW(ic, jc, kc, LXYSW) * SO(i - 1, j - 1, k, KB)
		            + CW(ic, jc - 1, kc, LXYL) * SO(i - 1, j - 1, k, KBN)
		            + CW(ic - 1, jc, kc, LYZNE)
		              * SO(i - 1, j - 1, k - 1, KPW) + CW(ic, jc - 1, kc, LXZNW)
		              * SO(i - 1, j - 1, k - 1, KPS)
		            + CW(ic - 1, jc - 1, kc, LXZA) * SO(i - 1, j - 1, k - 1, KPSW));
		CW(ic, jc, kc, LTSE)
		    = yp * (SO(i, j - 1, k, KBNW)
		            + CW(ic, jc - 1, kc, LXYR)
		              * SO(i - 1, j - 1, k, KBN) + CW(ic, jc, kc, LXYSE)
		              * SO(i - 1, j - 1, k, KB)
		            + CW(ic, jc, kc, LXYB) * SO(i, j - 1, k, KBW)
		            + CW(ic, jc - 1, kc, LXZNE)
		              * SO(i - 1, j - 1, k - 1, KPS) + CW(ic, jc, kc, LYZNE)
		              * SO(i, j - 1, k - 1, KPW)
		            + CW(ic, jc - 1, kc, LXZA) * SO(i, j - 1, k - 1, KPNW));
		CW(ic, jc, kc, LBSE)
		    = yp * (SO(i, j - 1, k - 1, KBSE)
		            + CW(ic, jc - 1, kc - 1, LXYR) * SO(i - 1, j - 1, k - 1, KBS)
		            + CW(ic, jc, kc - 1, LXYSE) * SO(i - 1, j - 1, k - 1, KB)
		            + CW(ic, jc, kc - 1, LXYB) * SO(i, j - 1, k - 1, KBE)
		            + CW(ic, jc - 1, kc, LXZSE) * SO(i - 1, j - 1, k - 1, KPS)
		            + CW(ic, jc, kc, LYZSE) * SO(i, j - 1, k - 1, KPW)
		            + CW(ic, jc - 1, kc, LXZB) * SO(i, j - 1, k - 1, KPNW));
	} else {
		dp = SO(i - 1, j - 1, k - 1, KPW) + SO(i - 1, j, k - 1, KPS)
		     + SO(i, j - 1, k - 1, KPW) + SO(i - 1, j - 1, k - 1, KPS)
		     + SO(i - 1, j - 1, k - 1, KB) + SO(i - 1, j - 1, k, KB);
		ep = rmin(min4(fabs(SO(i - 1, j - 1, k - 1, KPW) / d),
		               fabs(SO(i - 1, j, k - 1, KPS) / d),
		               fabs(SO(i, j - 1, k - 1, KPW) / d),
		               fabs(SO(i - 1, j - 1, k - 1, KPS) / d)),
		          rmin(fabs(SO(i - 1, j - 1, k - 1, KB) / d),
		               fabs(SO(i - 1, j - 1, k, KB) / d)));
		dp = (d - dp) * rmax(d - (1.0 + ep) * dp, 0.0)
		     / (fabs(d - (1.0 + ep) * dp) + eMACH) + dp;
		dp = 1.0 / dp;
		CW(ic, jc, kc, LTNW) = dp * (CW(ic - 1, jc, kc, LYZNW)
		                             * SO(i - 1, j - 1, k - 1, KPW)
		                             + CW(ic, jc, kc, LXZNW) * SO(i - 1, j, k - 1, KPS)
		                             + CW(ic, jc, kc, LXYNW) * SO(i - 1, j - 1, k, KB));
		CW(ic, jc, kc, LTNE) = dp * (CW(ic, jc, kc, LXZNE)
		                             * SO(i - 1, j, k - 1, KPS)
		                             + CW(ic, jc, kc, LYZNW) * SO(i, j - 1, k - 1, KPW)
		                             + CW(ic, jc, kc, LXYNE) * SO(i - 1, j - 1, k, KB));
		CW(ic, jc, kc, LBNW) = dp * (CW(ic, jc, kc - 1, LXYNW)
		                             * SO(i - 1, j - 1, k - 1, KB)
		                             + CW(ic - 1, jc, kc, LYZSW) * SO(i - 1, j - 1, k - 1, KPW)
		                             + CW(ic, jc, kc, LXZSW) * SO(i - 1, j, k - 1, KPS));
		CW(ic, jc, kc, LBNE) = dp * (CW(ic, jc, kc - 1, LXYNE)
		                             * SO(i - 1, j - 1, k - 1, KB)
		                             + CW(ic, jc, kc, LXZSE) * SO(i - 1, j, k - 1, KPS)
		                             + CW(ic, jc, kc, LYZSW) * SO(i, j - 1, k - 1, KPW));
		CW(ic, jc, kc, LBSW) = dp * (CW(ic, jc, kc - 1, LXYSW)
		                             * SO(i - 1, j - 1, k - 1, KB)
		                             + CW(ic - 1, jc, kc, LYZSE) * SO(i - 1, j - 1, k - 1, KPW)
		                             + CW(ic, jc - 1, kc, LXZSW) * SO(i - 1, j - 1, k - 1, KPS));
		CW(ic, jc, kc, LTSW) = dp * (CW(ic, jc, kc, LXYSW)
		                             * SO(i - 1, j - 1, k, KB)
		                             + CW(ic - 1, jc, kc, LYZNE) * SO(i - 1, j - 1, k - 1, KPW)
		                             + CW(ic, jc - 1, kc, LXZNW) * SO(i - 1, j - 1, k - 1, KPS));
		CW(ic, jc, kc, LTSE) = dp * (CW(ic, jc, kc, LXYSE)
		                             * SO(i - 1, j - 1, k, KB)
		                             + CW(ic, jc - 1, kc, LXZNE) * SO(i - 1, j - 1, k - 1, KPS)
		                             + CW(ic, jc, kc, LYZNE) * SO(i, j - 1, k - 1, KPW));
		CW(ic, jc, kc, LBSE) = dp * (CW(ic, jc, kc - 1, LXYSE)
		                             * SO(i - 1, j - 1, k - 1, KB)
		                             + CW(ic, jc - 1, kc, LXZSE) * SO(i - 1, j - 1, k - 1, KPS)
		                             + CW(ic, jc, kc, LYZSE) * SO(i, j - 1, k - 1, KPW));
	}
	}
}

// phase 0 = edges, 1 = faces, 2 = cell centres.  (ilo,jlo,klo) are the lower loop bounds of
// the "between two coarse points" directions: 3 in the reference's serial code; 2 on a side
// where a neighbouring subdomain owns the previous coarse point (distributed runs; the fine
// and coarse ghost layers then hold that neighbour's data).
void setup_interp3_phase(const real_t *so, real_t *ci, int IIF, int JJF, int KKF,
                         int IIC, int JJC, int KKC, int ifd, int phase, int ilo, int jlo, int klo, hipStream_t st)
{
	if (IIC < 2 || JJC < 2 || KKC < 2) return;
	dim3 grid((IIC - 1 + 127) / 128, JJC - 1, KKC - 1); // ic,jc,kc in [2,IIC] x [2,JJC] x [2,KKC]
	if (phase == 0)
		hipLaunchKernelGGL(interp3_edges, grid, dim3(128), 0, st, so, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, ilo, jlo, klo);
	else if (phase == 1)
		hipLaunchKernelGGL(interp3_faces, grid, dim3(128), 0, st, so, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, ilo, jlo, klo);
	else
		hipLaunchKernelGGL(interp3_centres, grid, dim3(128), 0, st, so, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, ilo, jlo, klo);
}

void setup_interp3(const real_t *so, real_t *ci, int IIF, int JJF, int KKF,
                   int IIC, int JJC, int KKC, int ifd, hipStream_t st)
{
	for (int phase = 0; phase < 3; phase++)
		setup_interp3_phase(so, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, phase, 3, 3, 3, st);
}
#undef SO
#undef CW

} // namespace cedar_amd
