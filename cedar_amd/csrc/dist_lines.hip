// Distributed zebra line relaxation (2D multi-GPU): set-up recurrence of a line segment and the small per-line kernels
// of a solve.  What it replaces in the reference: src/2d/ftn/mpi/BMG2_SymStd_relax_lines_x.f90:163-307 / _y.f90 with the
// multilevel tridiagonal solver of include/cedar/2d/mpi/ml_relax.h (209 lines): there an interface system of the
// segments' end points is reduced level by level over the ranks of a line; here a line cut by the ranks of a row (x
// lines) or column (y lines) of the rank grid is a chain of segments and both DPTTRS sweeps are first-order affine
// recurrences:
//   forward   y_i = rhs_i - e'_{i-1} y_{i-1}      =>  y_i = y0_i + pf_i * y_in,   pf_i = prod_{k<=i} (-e'_{k-1})
//   backward  x_i = y_i / d'_i - e'_i x_{i+1}     =>  x_i = x0_i + pb_i * x_in,   pb_i = prod_{k>=i} (-e'_k)
// every rank runs its segment from a zero carry (affine_lines, lines.hip: the scan kernel of the single-GPU line solve),
// the ranks of the line exchange (value leaving the segment, product of its multipliers), each composes the carry that
// enters its segment and adds carry x running product -- the decomposition the scan kernel uses between the tiles of
// one line, with the rank in the role of the tile.  Round 2 did the set-up recurrence, the running products and the
// carry composition with torch tensor operations (cedar_amd/dist2d.py); these kernels take torch out of the rank process.
#include "common.h"
#include "dist_lines.h"

namespace cedar_amd {

// One lane per line: d'_i = d_i - e_{i-1}^2 / d'_{i-1}, e'_i = e_i / d'_i along this rank's segment (DPTTRF,
// BMG2_SymStd_SETUP_lines_x.f90:68-87), started from the last pivot of the previous segment (piv_in) when there is one.
// Lines are stored by colour (LineFactors): line l -> colour l & 1, index l >> 1.
__global__ void dist_lines_factor_kernel(const real_t *__restrict__ A, int II, int JJ, int dir, int npos, int nl,
                                         const real_t *__restrict__ piv_in, int has_prev, int has_next, LineFactors F,
                                         real_t *__restrict__ piv_out)
{
	const int l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= nl) return;
	const size_t PS = (size_t)II * JJ;
	// entry of plane s at (line l, position i), i = 0 .. npos (position i sits at array index 1 + i)
	const size_t base = dir == 0 ? (size_t)(1 + l) * II + 1 : (size_t)II + 1 + l;
	const size_t step = dir == 0 ? 1 : (size_t)II;
	const real_t *diag = A + base, *cpl = A + (size_t)(dir == 0 ? KW : KS) * PS + base;
	const size_t o = F.line_offset(l);
	real_t prev = has_prev ? piv_in[l] : 0.0, pf = 1.0;
	for (int i = 0; i < npos; i++) {
		const real_t e = -cpl[i * step]; // e_{i-1} of position i (reference sign), SETUP_lines_x.f90:77
		real_t en = 0.0, dpi;
		if (i == 0 && !has_prev) dpi = diag[0];
		else { en = e / prev; dpi = diag[i * step] - en * e; }
		F.dp[o + i] = dpi;
		F.af[o + i] = -en;
		if (i > 0) F.ab[o + i - 1] = -en; // e'_{i-1} seen from position i-1: its coupling to the next unknown
		pf = has_prev ? (i == 0 ? -en : -en * pf) : 0.0;
		F.pf[o + i] = pf;
		prev = dpi;
	}
	piv_out[l] = prev;
	const real_t e_out = has_next ? (-cpl[(size_t)npos * step]) / prev : 0.0; // scaled coupling leaving the segment
	F.ab[o + npos - 1] = -e_out;
	real_t pb = 1.0;
	for (int i = npos - 1; i >= 0; i--) {
		pb = has_next ? (i == npos - 1 ? F.ab[o + i] : F.ab[o + i] * pb) : 0.0;
		F.pb[o + i] = pb;
	}
}

void dist_lines_factor(const real_t *A, int II, int JJ, int dir, int npos, int nl, const real_t *piv_in, int has_prev,
                       int has_next, const LineFactors &F, real_t *piv_out, hipStream_t st)
{
	if (nl <= 0 || npos <= 0) return;
	hipLaunchKernelGGL(dist_lines_factor_kernel, dim3((nl + 63) / 64), dim3(64), 0, st, A, II, JJ, dir, npos, nl, piv_in,
	                   has_prev, has_next, F, piv_out);
}

// out[l] = (v[l][pos], p[l][pos]): what leaves a segment and the product of its multipliers
__global__ void dist_lines_pick_kernel(const real_t *__restrict__ v, const real_t *__restrict__ p, int nlines, int ld, int pos,
                                       real_t *__restrict__ out)
{
	const int l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= nlines) return;
	out[2 * l] = v[(size_t)l * ld + pos];
	out[2 * l + 1] = p[(size_t)l * ld + pos];
}

void dist_lines_pick(const real_t *v, const real_t *p, int nlines, int ld, int pos, real_t *out, hipStream_t st)
{
	if (nlines <= 0) return;
	hipLaunchKernelGGL(dist_lines_pick_kernel, dim3((nlines + 127) / 128), dim3(128), 0, st, v, p, nlines, ld, pos, out);
}

// carry entering segment `seg`: forward: c = 0; for r = 0 .. seg-1: c = val_r + prod_r * c;
//                               backward: c = 0; for r = nseg-1 .. seg+1: c = val_r + prod_r * c.   parts: (nseg, nlines, 2)
__global__ void dist_lines_compose_kernel(const real_t *__restrict__ parts, int nseg, int seg, int nlines, int backward,
                                          real_t *__restrict__ carry)
{
	const int l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= nlines) return;
	real_t c = 0.0;
	if (!backward)
		for (int r = 0; r < seg; r++) c = parts[((size_t)r * nlines + l) * 2] + parts[((size_t)r * nlines + l) * 2 + 1] * c;
	else
		for (int r = nseg - 1; r > seg; r--) c = parts[((size_t)r * nlines + l) * 2] + parts[((size_t)r * nlines + l) * 2 + 1] * c;
	carry[l] = c;
}

void dist_lines_compose(const real_t *parts, int nseg, int seg, int nlines, int backward, real_t *carry, hipStream_t st)
{
	if (nlines <= 0) return;
	hipLaunchKernelGGL(dist_lines_compose_kernel, dim3((nlines + 127) / 128), dim3(128), 0, st, parts, nseg, seg, nlines,
	                   backward, carry);
}

} // namespace cedar_amd
