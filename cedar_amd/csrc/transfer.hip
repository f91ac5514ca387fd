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
                                                         const real_t *__restrict__ ci, int II, int JJ, int IIC, int JJC)
{
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 1; // 0-based incl. ghost
	const int jc = blockIdx.y + 1;
	if (ic > IIC - 2) return;
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

void restrict2(const real_t *q, real_t *qc, const real_t *ci, int II, int JJ, int IIC, int JJC, hipStream_t st)
{
	if (IIC < 3 || JJC < 3) return;
	dim3 grid((IIC - 2 + 255) / 256, JJC - 2);
	hipLaunchKernelGGL(restrict2_kernel, grid, dim3(256), 0, st, q, qc, ci, II, JJ, IIC, JJC);
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
                                                           int IIC, int JJC, int IIF, int JJF, int imax, int jmax)
{
	// 1-based fine indices as in the reference
	const int i = blockIdx.x * blockDim.x + threadIdx.x + 2;
	const int j = blockIdx.y + 2;
	if (i > IIF) return;
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
                 int IIC, int JJC, int IIF, int JJF, hipStream_t st)
{
	if (IIF < 3 || JJF < 3) return;
	const int imax = 2 * ((IIF - 2) / 2 + 2 - 1), jmax = 2 * ((JJF - 2) / 2 + 2 - 1);
	dim3 grid((IIF - 1 + 255) / 256, JJF - 1);
	hipLaunchKernelGGL(interp_add2_kernel, grid, dim3(256), 0, st, q, qc, res, so /* KO plane */, ci,
	                   IIC, JJC, IIF, JJF, imax, jmax);
}

// ------------------------------------------------------------------ interp_add 3D
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
	for (int i = threadIdx.x + 2; i <= IIF; i += blockDim.x) {
		const size_t x = (size_t)(i - 1) + sf * (size_t)(j - 1) + tf * (size_t)(k - 1);
		const bool interior = i <= IIF - 1 && j <= JJF - 1 && k <= KKF - 1;
		const bool io = i & 1;
		// written range of this plane type: even planes i in [2, imax_all] (all parities);
		// odd planes: even i up to imax_e, odd i up to imax_all-1
		const bool upd = row_in_range && (ko ? (io ? i <= imax_all - 1 : i <= imax_e) : i <= imax_all);
		real_t r = 0.0;
		if (interior || upd) r = res[x];
		if (interior) {
			r = r / so_diag[x];
			res[x] = r;
		}
		if (!upd) continue;
		const int ic = io ? (i + 1) / 2 + 1 : i / 2 + 1;
		const size_t c = (size_t)(ic - 1) + sc * (size_t)(jc - 1) + tc * (size_t)(kc - 1);
#define CIv(slot) ci[(size_t)(slot)*PC + c]
		real_t v = q[x];
		if (!ko) {
			if (!io && !jo) {
				v = v + qc[c];
			} else if (io && !jo) {
				real_t a = CIv(LXYR) * qc[c] + CIv(LXYL) * qc[c - 1];
				v = v + a + r;
			} else if (!io && jo) {
				real_t a = CIv(LXYA) * qc[c] + CIv(LXYB) * qc[c - sc];
				v = v + a + r;
			} else {
				real_t a = CIv(LXYSW) * qc[c - 1 - sc] + CIv(LXYNW) * qc[c - 1]
				           + CIv(LXYNE) * qc[c] + CIv(LXYSE) * qc[c - sc];
				v = v + a + r;
			}
		} else {
			if (!io && !jo) {
				v = v + CIv(LXZA) * qc[c] + CIv(LXZB) * qc[c - tc] + r;
			} else if (!io && jo) {
				v = v + CIv(LYZNW) * qc[c] + CIv(LYZNE) * qc[c - sc]
				    + CIv(LYZSW) * qc[c - tc] + CIv(LYZSE) * qc[c - sc - tc] + r;
			} else if (io && !jo) {
				v = v + CIv(LXZNW) * qc[c - 1] + CIv(LXZNE) * qc[c]
				    + CIv(LXZSW) * qc[c - 1 - tc] + CIv(LXZSE) * qc[c - tc] + r;
			} else {
				v = v + CIv(LTNW) * qc[c - 1] + CIv(LTNE) * qc[c]
				    + CIv(LTSW) * qc[c - 1 - sc] + CIv(LTSE) * qc[c - sc]
				    + CIv(LBNW) * qc[c - 1 - tc] + CIv(LBNE) * qc[c - tc]
				    + CIv(LBSW) * qc[c - 1 - sc - tc] + CIv(LBSE) * qc[c - sc - tc] + r;
			}
		}
#undef CIv
		q[x] = v;
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
	int bs = IIF >= 256 ? 256 : (IIF > 64 ? 128 : 64);
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
		const size_t a = pl * PS + (size_t)(tab.i0[bx] + i) + (size_t)II * ((size_t)(tab.j0[bx] + j) + (size_t)JJ * (size_t)(tab.k0[bx] + k));
		const size_t b = tab.off[bx] * (size_t)nplanes + t;
		if (unpack) arr[a] = buf[b];
		else buf[b] = arr[a];
	}
}

void box_copy(real_t *arr, int II, int JJ, int KK, int nplanes, int nboxes, const int *boxes /* 6 per box */,
              const unsigned long long *offsets, real_t *buf, int unpack, hipStream_t st)
{
	if (nboxes <= 0) return;
	BoxTable tab;
	tab.n = nboxes;
	size_t maxn = 1;
	for (int b = 0; b < nboxes && b < 26; b++) {
		tab.i0[b] = boxes[6 * b]; tab.j0[b] = boxes[6 * b + 1]; tab.k0[b] = boxes[6 * b + 2];
		tab.ni[b] = boxes[6 * b + 3]; tab.nj[b] = boxes[6 * b + 4]; tab.nk[b] = boxes[6 * b + 5];
		tab.off[b] = offsets[b];
		size_t n = (size_t)tab.ni[b] * tab.nj[b] * tab.nk[b] * nplanes;
		if (n > maxn) maxn = n;
	}
	unsigned gx = (unsigned)((maxn + 255) / 256);
	if (gx > 2048) gx = 2048;
	hipLaunchKernelGGL(box_copy_kernel, dim3(gx, nboxes), dim3(256), 0, st, arr, II, JJ, KK, nplanes, tab, buf, unpack);
}

} // namespace cedar_amd
