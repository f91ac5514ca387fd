// Inter-grid transfers: restriction b_c = P^T r and interpolate-and-add
// x += P x_c (+ r/diag at F points).
// Replace BMG2_SymStd_restrict (src/2d/ftn/BMG2_SymStd_restrict.f90:73-92),
// BMG3_SymStd_restrict (src/3d/ftn/BMG3_SymStd_restrict.f90:115-150),
// BMG2_SymStd_interp_add (src/2d/ftn/BMG2_SymStd_interp_add.f90:88-137) and
// BMG3_SymStd_interp_add (src/3d/ftn/BMG3_SymStd_interp_add.f90:88-240).
//
// restrict: one lane per coarse point (gather, unit stride in the coarse row).
// interp_add: the reference scatters from coarse cells; every fine point is
// written exactly once, so the device kernel is one lane per *fine* point that
// picks its formula from its (i,j,k) parity -- a pure gather, no atomics, and
// the in-place RES /= diag pass is fused in (each RES entry is only consumed by
// its own point).  The reference's index ranges are kept exactly, including
// the ghost column/row it touches for even extents (IICF1 = (IIF-2)/2+2).
// Term order = reference, no contraction => bit-identical results.
#include "common.h"

namespace cedar_amd {

// ------------------------------------------------------------------ restrict
__global__ __launch_bounds__(256) void restrict2_kernel(const real_t *__restrict__ q, real_t *__restrict__ qc,
                                                         const real_t *__restrict__ ci, int II, int JJ, int IIC, int JJC,
                                                         size_t bsf, size_t bsc)
{
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 1; // 0-based incl. ghost
	const int jc = blockIdx.y + 1;
	if (ic > IIC - 2) return;
	q += bsf * blockIdx.z; qc += bsc * blockIdx.z; // batch item (common.h Batch)
	const size_t PC = (size_t)IIC * JJC, sc = IIC, sf = II;
	const size_t c = (size_t)ic + sc * jc;
	const size_t f = (size_t)(2 * ic - 1) + sf * (size_t)(2 * jc - 1); // 1-based i = 2(ic1-1) -> 0-based 2*ic-1
	real_t s = ci[LNE * PC + c] * q[f - 1 - sf];
	s = s + ci[LA * PC + c] * q[f - sf];
	s = s + ci[LNW * PC + c + 1] * q[f + 1 - sf];
	s = s + ci[LR * PC + c] * q[f - 1];
	s = s + q[f];
	s = s + ci[LL * PC + c + 1] * q[f + 1];
	s = s + ci[LSE * PC + c + sc] * q[f - 1 + sf];
	s = s + ci[LB * PC + c + sc] * q[f + sf];
	s = s + ci[LSW * PC + c + 1 + sc] * q[f + 1 + sf];
	qc[c] = s;
}

void restrict2(const real_t *q, real_t *qc, const real_t *ci, int II, int JJ, int IIC, int JJC, hipStream_t st, Batch bf, Batch bc)
{
	if (IIC < 3 || JJC < 3) return;
	dim3 grid((IIC - 2 + 255) / 256, JJC - 2, bf.n);
	hipLaunchKernelGGL(restrict2_kernel, grid, dim3(256), 0, st, q, qc, ci, II, JJ, IIC, JJC, bf.stride, bc.stride);
}

__global__ __launch_bounds__(128) void restrict3_kernel(const real_t *__restrict__ q, real_t *__restrict__ qc,
                                                         const real_t *__restrict__ ci, int II, int JJ, int KK,
                                                         int IIC, int JJC, int KKC)
{
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 1;
	const int jc = blockIdx.y + 1, kc = blockIdx.z + 1;
	if (ic > IIC - 2) return;
	const size_t sc = IIC, tc = (size_t)IIC * JJC, PC = tc * KKC;
	const size_t sf = II, tf = (size_t)II * JJ;
	const size_t c = (size_t)ic + sc * jc + tc * kc;
	const size_t f = (size_t)(2 * ic - 1) + sf * (size_t)(2 * jc - 1) + tf * (size_t)(2 * kc - 1);
#define CIv(slot, off) ci[(size_t)(slot)*PC + c + (off)]
	real_t s = CIv(LXYNE, 0) * q[f - 1 - sf];
	s = s + CIv(LXYA, 0) * q[f - sf];
	s = s + CIv(LXYNW, 1) * q[f + 1 - sf];
	s = s + CIv(LXYR, 0) * q[f - 1];
	s = s + q[f];
	s = s + CIv(LXYL, 1) * q[f + 1];
	s = s + CIv(LXYSE, sc) * q[f - 1 + sf];
	s = s + CIv(LXYB, sc) * q[f + sf];
	s = s + CIv(LXYSW, 1 + sc) * q[f + 1 + sf];
	s = s + CIv(LTNE, 0) * q[f - 1 - sf - tf];
	s = s + CIv(LYZNW, 0) * q[f - sf - tf];
	s = s + CIv(LTNW, 1) * q[f + 1 - sf - tf];
	s = s + CIv(LXZNE, 0) * q[f - 1 - tf];
	s = s + CIv(LXZA, 0) * q[f - tf];
	s = s + CIv(LXZNW, 1) * q[f + 1 - tf];
	s = s + CIv(LTSE, sc) * q[f - 1 + sf - tf];
	s = s + CIv(LYZNE, sc) * q[f + sf - tf];
	s = s + CIv(LTSW, 1 + sc) * q[f + 1 + sf - tf];
	s = s + CIv(LBNE, tc) * q[f - 1 - sf + tf];
	s = s + CIv(LYZSW, tc) * q[f - sf + tf];
	s = s + CIv(LBNW, 1 + tc) * q[f + 1 - sf + tf];
	s = s + CIv(LXZSE, tc) * q[f - 1 + tf];
	s = s + CIv(LXZB, tc) * q[f + tf];
	s = s + CIv(LXZSW, 1 + tc) * q[f + 1 + tf];
	s = s + CIv(LBSE, sc + tc) * q[f - 1 + sf + tf];
	s = s + CIv(LYZSE, sc + tc) * q[f + sf + tf];
	s = s + CIv(LBSW, 1 + sc + tc) * q[f + 1 + sf + tf];
#undef CIv
	qc[c] = s;
}

void restrict3(const real_t *q, real_t *qc, const real_t *ci, int II, int JJ, int KK,
               int IIC, int JJC, int KKC, hipStream_t st)
{
	if (IIC < 3 || JJC < 3 || KKC < 3) return;
	dim3 grid((IIC - 2 + 127) / 128, JJC - 2, KKC - 2);
	hipLaunchKernelGGL(restrict3_kernel, grid, dim3(128), 0, st, q, qc, ci, II, JJ, KK, IIC, JJC, KKC);
}

// ------------------------------------------------------------------ interp_add 2D
__global__ __launch_bounds__(256) void interp_add2_kernel(real_t *__restrict__ q, const real_t *__restrict__ qc,
                                                           real_t *__restrict__ res, const real_t *__restrict__ so_diag,
                                                           const real_t *__restrict__ ci,
                                                           int IIC, int JJC, int IIF, int JJF, int imax, int jmax,
                                                           size_t bsf, size_t bsc)
{
	// 1-based fine indices as in the reference
	const int i = blockIdx.x * blockDim.x + threadIdx.x + 2;
	const int j = blockIdx.y + 2;
	if (i > IIF) return;
	q += bsf * blockIdx.z; res += bsf * blockIdx.z; qc += bsc * blockIdx.z; // batch item (common.h Batch)
	const size_t sf = IIF, sc = IIC, PC = (size_t)IIC * JJC;
	const size_t x = (size_t)(i - 1) + sf * (size_t)(j - 1);
	const bool interior = i <= IIF - 1 && j <= JJF - 1;
	real_t r = 0.0;
	if (interior || (i <= imax && j <= jmax)) r = res[x];
	if (interior) {
		r = r / so_diag[x];
		res[x] = r;
	}
	if (i > imax || j > jmax) return;
	const bool io = i & 1, jo = j & 1;
	const int ic = io ? (i + 1) / 2 + 1 : i / 2 + 1, jc = jo ? (j + 1) / 2 + 1 : j / 2 + 1; // 1-based coarse
	const size_t c = (size_t)(ic - 1) + sc * (size_t)(jc - 1);
	real_t v = q[x];
	if (!io && !jo) {
		v = v + qc[c];
	} else if (io && !jo) {
		real_t a = ci[LR * PC + c] * qc[c] + ci[LL * PC + c] * qc[c - 1];
		v = v + a + r;
	} else if (!io && jo) {
		real_t a = ci[LA * PC + c] * qc[c] + ci[LB * PC + c] * qc[c - sc];
		v = v + a + r;
	} else {
		real_t a = ci[LSW * PC + c] * qc[c - 1 - sc] + ci[LNW * PC + c] * qc[c - 1]
		           + ci[LNE * PC + c] * qc[c] + ci[LSE * PC + c] * qc[c - sc];
		v = v + a + r;
	}
	q[x] = v;
}

void interp_add2(real_t *q, const real_t *qc, real_t *res, const real_t *so, const real_t *ci,
                 int IIC, int JJC, int IIF, int JJF, hipStream_t st, Batch bf, Batch bc)
{
	if (IIF < 3 || JJF < 3) return;
	const int imax = 2 * ((IIF - 2) / 2 + 2 - 1), jmax = 2 * ((JJF - 2) / 2 + 2 - 1);
	dim3 grid((IIF - 1 + 255) / 256, JJF - 1, bf.n);
	hipLaunchKernelGGL(interp_add2_kernel, grid, dim3(256), 0, st, q, qc, res, so /* KO plane */, ci,
	                   IIC, JJC, IIF, JJF, imax, jmax, bf.stride, bc.stride);
}

// ------------------------------------------------------------------ interp_add 3D
// One workgroup per fine row (j,k), lane p owns the pair of 1-based columns (2p+2, 2p+3): an even
// (coincides with / lies above a coarse column) and an odd one (between two coarse columns).  The
// (j,k) parities are uniform over the row, so every lane runs the same two formulas: no divergence,
// the fine arrays move as 16-byte pairs, the CI / coarse-q entries of a row as consecutive doubles.
// Expressions and their evaluation order are those of BMG3_SymStd_interp_add.f90:88-240 (scatter in
// the reference, gathered per fine point here); the written index ranges -- including the ghost
// column / row / plane the reference touches for even extents -- are passed in by the host.
template <bool KO, bool JO>
__device__ __forceinline__ void interp_add3_pair(real_t &ve, real_t &vo, real_t re, real_t ro, bool upd_e, bool upd_o,
                                                 const real_t *__restrict__ qc, const real_t *__restrict__ ci,
                                                 size_t ce, size_t sc, size_t tc, size_t PC)
{
	const size_t co = ce + 1; // coarse column of the odd point: (i+1)/2+1
#define CIe(slot) ci[(size_t)(slot)*PC + ce]
#define CIo(slot) ci[(size_t)(slot)*PC + co]
	if (!KO) {
		if (!JO) {
			if (upd_e) ve = ve + qc[ce];
			if (upd_o) {
				real_t a = CIo(LXYR) * qc[co] + CIo(LXYL) * qc[co - 1];
				vo = vo + a + ro;
			}
		} else {
			if (upd_e) {
				real_t a = CIe(LXYA) * qc[ce] + CIe(LXYB) * qc[ce - sc];
				ve = ve + a + re;
			}
			if (upd_o) {
				real_t a = CIo(LXYSW) * qc[co - 1 - sc] + CIo(LXYNW) * qc[co - 1]
				           + CIo(LXYNE) * qc[co] + CIo(LXYSE) * qc[co - sc];
				vo = vo + a + ro;
			}
		}
	} else {
		if (!JO) {
			if (upd_e) ve = ve + CIe(LXZA) * qc[ce] + CIe(LXZB) * qc[ce - tc] + re;
			if (upd_o)
				vo = vo + CIo(LXZNW) * qc[co - 1] + CIo(LXZNE) * qc[co]
				     + CIo(LXZSW) * qc[co - 1 - tc] + CIo(LXZSE) * qc[co - tc] + ro;
		} else {
			if (upd_e)
				ve = ve + CIe(LYZNW) * qc[ce] + CIe(LYZNE) * qc[ce - sc]
				     + CIe(LYZSW) * qc[ce - tc] + CIe(LYZSE) * qc[ce - sc - tc] + re;
			if (upd_o)
				vo = vo + CIo(LTNW) * qc[co - 1] + CIo(LTNE) * qc[co]
				     + CIo(LTSW) * qc[co - 1 - sc] + CIo(LTSE) * qc[co - sc]
				     + CIo(LBNW) * qc[co - 1 - tc] + CIo(LBNE) * qc[co - tc]
				     + CIo(LBSW) * qc[co - 1 - sc - tc] + CIo(LBSE) * qc[co - sc - tc] + ro;
		}
	}
#undef CIe
#undef CIo
}

__global__ __launch_bounds__(256) void interp_add3_kernel(real_t *__restrict__ q, const real_t *__restrict__ qc,
                                                           const real_t *__restrict__ so_diag, real_t *__restrict__ res,
                                                           const real_t *__restrict__ ci,
                                                           int IIC, int JJC, int KKC, int IIF, int JJF, int KKF,
                                                           int imax_e, int imax_all, int jmax, int kmax_e, int kmax_o,
                                                           unsigned nrows)
{
	const unsigned L = xcd_remap(blockIdx.x, nrows);
	if (L >= nrows) return;
	const int j = (int)(L % (unsigned)(JJF - 1)) + 2, k = (int)(L / (unsigned)(JJF - 1)) + 2; // 1-based, 2..JJF / 2..KKF
	const size_t sf = IIF, tf = (size_t)IIF * JJF;
	const size_t sc = IIC, tc = (size_t)IIC * JJC, PC = tc * KKC;
	const bool jo = j & 1, ko = k & 1;
	const int jc = jo ? (j + 1) / 2 + 1 : j / 2 + 1, kc = ko ? (k + 1) / 2 + 1 : k / 2 + 1;
	const bool row_in_range = j <= jmax && k <= (ko ? kmax_o : kmax_e);
	const bool row_interior = j <= JJF - 1 && k <= KKF - 1;
	const size_t rowf = sf * (size_t)(j - 1) + tf * (size_t)(k - 1);
	const size_t rowc = sc * (size_t)(jc - 1) + tc * (size_t)(kc - 1);
	for (int p = threadIdx.x; 2 * p + 2 <= IIF; p += blockDim.x) {
		const int ie = 2 * p + 2, io = ie + 1; // 1-based
		const bool have_o = io <= IIF;
		const size_t x = rowf + (size_t)(ie - 1);
		const bool int_e = row_interior && ie <= IIF - 1, int_o = row_interior && have_o && io <= IIF - 1;
		// written range of this plane type: even planes i in [2, imax_all] (all parities);
		// odd planes: even i up to imax_e, odd i up to imax_all-1
		const bool upd_e = row_in_range && (ko ? ie <= imax_e : ie <= imax_all);
		const bool upd_o = row_in_range && have_o && (ko ? io <= imax_all - 1 : io <= imax_all);
		if (!(int_e || int_o || upd_e || upd_o)) continue;
		real_t re = 0.0, ro = 0.0, de = 1.0, dn = 1.0, ve = 0.0, vo = 0.0;
		if (have_o) {
			d2u t = *reinterpret_cast<const d2u *>(res + x); re = t.x; ro = t.y;
			t = *reinterpret_cast<const d2u *>(so_diag + x); de = t.x; dn = t.y;
			t = *reinterpret_cast<const d2u *>(q + x); ve = t.x; vo = t.y;
		} else {
			re = res[x]; de = so_diag[x]; ve = q[x];
		}
		if (int_e) re = re / de;
		if (int_o) ro = ro / dn;
		if (int_e && int_o) {
			d2u t; t.x = re; t.y = ro;
			*reinterpret_cast<d2u *>(res + x) = t;
		} else {
			if (int_e) res[x] = re;
			if (int_o) res[x + 1] = ro;
		}
		if (!(upd_e || upd_o)) continue;
		const size_t ce = rowc + (size_t)(ie / 2); // 0-based: ic-1 = i/2
		if (ko) {
			if (jo) interp_add3_pair<true, true>(ve, vo, re, ro, upd_e, upd_o, qc, ci, ce, sc, tc, PC);
			else interp_add3_pair<true, false>(ve, vo, re, ro, upd_e, upd_o, qc, ci, ce, sc, tc, PC);
		} else {
			if (jo) interp_add3_pair<false, true>(ve, vo, re, ro, upd_e, upd_o, qc, ci, ce, sc, tc, PC);
			else interp_add3_pair<false, false>(ve, vo, re, ro, upd_e, upd_o, qc, ci, ce, sc, tc, PC);
		}
		if (upd_e && upd_o) {
			d2u t; t.x = ve; t.y = vo;
			*reinterpret_cast<d2u *>(q + x) = t;
		} else {
			if (upd_e) q[x] = ve;
			if (upd_o) q[x + 1] = vo;
		}
	}
}

void interp_add3(real_t *q, const real_t *qc, const real_t *so, real_t *res, const real_t *ci,
                 int IIC, int JJC, int KKC, int IIF, int JJF, int KKF, hipStream_t st)
{
	if (IIF < 3 || JJF < 3 || KKF < 3) return;
	const int iicf1 = (IIF - 2) / 2 + 2, jjcf1 = (JJF - 2) / 2 + 2, kkcf1 = (KKF - 2) / 2 + 2;
	const int imax_all = 2 * (iicf1 - 1);   // even planes, and odd-i bound + 1 on odd planes
	const int imax_e = 2 * (IIC - 2);       // even i on odd planes: ic = 2..iic1
	const int jmax = 2 * (jjcf1 - 1);
	const int kmax_e = 2 * (KKC - 2);       // coarse planes kc = 2..kkc1
	const int kmax_o = 2 * (kkcf1 - 1) - 1; // odd planes kc = 3..kkcf1
	unsigned nrows = (unsigned)(JJF - 1) * (unsigned)(KKF - 1);
	int bs = IIF / 2 >= 256 ? 256 : (IIF / 2 > 64 ? 128 : 64); // one lane per column pair
	hipLaunchKernelGGL(interp_add3_kernel, dim3(xcd_grid(nrows)), dim3(bs), 0, st, q, qc, so /* KP plane */, res, ci,
	                   IIC, JJC, KKC, IIF, JJF, KKF, imax_e, imax_all, jmax, kmax_e, kmax_o, nrows);
}

// ------------------------------------------------------------------ halo pack / unpack
// Copies up to 26 sub-boxes of a (planes x KK x JJ x II) array to / from one contiguous buffer in a
// single launch (blockIdx.y = box): the pack and unpack steps of the ghost-layer exchange that
// replaces the MSG gather/scatter index lists of the reference (src/2d/ftn/mpi/mpi_msg.F:483-550).
struct BoxTable {
	int n;
	int i0[26], j0[26], k0[26], ni[26], nj[26], nk[26];
	int sj[26], sk[26]; // step between the rows / planes of a box (1: dense; 2: one row class / one k-parity)
	unsigned long long off[26]; // offset of the box (first plane) in the buffer, in doubles
};

__global__ __launch_bounds__(256) void box_copy_kernel(real_t *__restrict__ arr, int II, int JJ, int KK, int nplanes,
                                                        BoxTable tab, real_t *__restrict__ buf, int unpack)
{
	const int bx = blockIdx.y;
	const int ni = tab.ni[bx], nj = tab.nj[bx], nk = tab.nk[bx];
	const size_t nbox = (size_t)ni * nj * nk, ntot = nbox * (size_t)nplanes;
	const size_t PS = (size_t)II * JJ * KK;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < ntot; t += (size_t)gridDim.x * blockDim.x) {
		const size_t pl = t / nbox, r = t % nbox;
		const int i = (int)(r % ni), j = (int)((r / ni) % nj), k = (int)(r / ((size_t)ni * nj));
		const size_t a = pl * PS + (size_t)(tab.i0[bx] + i) +
		                 (size_t)II * ((size_t)(tab.j0[bx] + j * tab.sj[bx]) + (size_t)JJ * (size_t)(tab.k0[bx] + k * tab.sk[bx]));
		const size_t b = tab.off[bx] * (size_t)nplanes + t;
		if (unpack) arr[a] = buf[b];
		else buf[b] = arr[a];
	}
}

void box_copy(real_t *arr, int II, int JJ, int KK, int nplanes, int nboxes, const int *boxes /* 6 (8: + row step, plane step) per box */,
              const unsigned long long *offsets, real_t *buf, int unpack, hipStream_t st, int strided)
{
	if (nboxes <= 0) return;
	BoxTable tab;
	tab.n = nboxes;
	size_t maxn = 1;
	const int w = strided ? 8 : 6;
	for (int b = 0; b < nboxes && b < 26; b++) {
		tab.i0[b] = boxes[w * b]; tab.j0[b] = boxes[w * b + 1]; tab.k0[b] = boxes[w * b + 2];
		tab.ni[b] = boxes[w * b + 3]; tab.nj[b] = boxes[w * b + 4]; tab.nk[b] = boxes[w * b + 5];
		tab.sj[b] = strided ? boxes[w * b + 6] : 1; tab.sk[b] = strided ? boxes[w * b + 7] : 1;
		tab.off[b] = offsets[b];
		size_t n = (size_t)tab.ni[b] * tab.nj[b] * tab.nk[b] * nplanes;
		if (n > maxn) maxn = n;
	}
	unsigned gx = (unsigned)((maxn + 255) / 256);
	if (gx > 2048) gx = 2048;
	hipLaunchKernelGGL(box_copy_kernel, dim3(gx, nboxes), dim3(256), 0, st, arr, II, JJ, KK, nplanes, tab, buf, unpack);
}

} // namespace cedar_amd
