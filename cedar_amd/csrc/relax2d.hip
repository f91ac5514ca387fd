// 2D point relaxation: 9-point 4-colour and 5-point red-black Gauss-Seidel.
// Replaces BMG2_SymStd_relax_GS (reference src/2d/ftn/BMG2_SymStd_relax_GS.f90:76-137).
//
// 9-point: the reference's loop nest visits, for each row parity, row by row,
// the even-i then the odd-i points (DOWN; reversed for UP).  Rows of equal
// parity do not couple, and the second i-colour of a row only needs the first
// i-colour *of the same row*, so a workgroup that owns whole rows relaxes both
// i-colours in one pass: 2 launches per sweep, unit-stride row streams
// (64 algorithmic B/DOF), any row length.
// Bit-identical to the reference CPU build (term order kept, no contraction).
#include "common.h"

namespace cedar_amd {

__device__ __forceinline__ real_t gs9_mem(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                          const real_t *__restrict__ q, size_t sj, size_t PS, size_t x)
{
	real_t s = qf[x];
	s = s + so[KW * PS + x] * q[x - 1];
	s = s + so[KW * PS + x + 1] * q[x + 1];
	s = s + so[KS * PS + x] * q[x - sj];
	s = s + so[KS * PS + x + sj] * q[x + sj];
	s = s + so[KSW * PS + x] * q[x - 1 - sj];
	s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
	s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
	s = s + so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
	return s;
}

// EFIRST: even 1-based i first (DOWN in 2D), else odd i first (UP).
// One workgroup per grid row; lanes stride over the row's (i_e, i_o) pairs.
// Phase 1 relaxes the first i-colour of the whole row in place, phase 2 the
// second one (it reads the fresh first-colour values back through L1/L2).
// In-place is safe: within one launch only rows of one parity are written and
// a row's relaxation reads rows j-1, j+1 (other parity) and itself.
template <int BS, bool EFIRST>
__global__ __launch_bounds__(BS) void relax9_rows(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                   real_t *q, const real_t *__restrict__ sor,
                                                   int II, int JJ, int jb, int nrows)
{
	const unsigned L = xcd_remap(blockIdx.x, (unsigned)nrows);
	if (L >= (unsigned)nrows) return;
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t row = (size_t)(1 + jb + 2 * (int)L) * sj;
	const int first = EFIRST ? 1 : 2; // 0-based offset of the first colour's first point
#pragma unroll 1
	for (int phase = 0; phase < 2; phase++) {
		const int start = phase == 0 ? first : 3 - first;
		for (int i = start + 2 * (int)threadIdx.x; i <= II - 2; i += 2 * BS) {
			const size_t x = row + i;
			q[x] = gs9_mem(so, qf, q, sj, PS, x) * sor[PS + x];
		}
		__syncthreads(); // workgroup-scope release/acquire: phase 2 sees phase 1's stores
	}
}

// 5-point red-black, one colour per launch (relax_GS.f90:120-135): colour = mod(j+jo,2)
__global__ __launch_bounds__(256) void relax5_colour(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                      real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                      int II, int JJ, int jo)
{
	const int j1 = blockIdx.y + 2; // 1-based
	const int a = blockIdx.x * blockDim.x + threadIdx.x;
	const int i1 = (j1 + jo) % 2 + 2 + 2 * a;
	if (i1 > II - 1) return;
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t x = (size_t)(i1 - 1) + sj * (size_t)(j1 - 1);
	real_t s = qf[x];
	s = s + so[KW * PS + x] * q[x - 1];
	s = s + so[KW * PS + x + 1] * q[x + 1];
	s = s + so[KS * PS + x] * q[x - sj];
	s = s + so[KS * PS + x + sj] * q[x + sj];
	q[x] = s * sor[PS + x];
}

void relax2_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
               int II, int JJ, int nstncl, int updown, hipStream_t st)
{
	if (II < 3 || JJ < 3) return;
	const bool down = (updown == BMG_DOWN);
	if (nstncl == 5) {
		for (int c = 0; c < 2; c++) {
			int jb = down ? c : 1 - c;  // DOWN: rows J=2,4,.. first (LSTART=2)
			int nrows = (JJ - 2 - jb + 1) / 2;
			if (nrows <= 0) continue;
			unsigned grid = xcd_grid((unsigned)nrows);
			const int npairs = (II - 2 + 1) / 2;
			if (npairs <= 64) {
				if (down) hipLaunchKernelGGL((relax9_rows<64, true>), dim3(grid), dim3(64), 0, st, so, qf, q, sor, II, JJ, jb, nrows);
				else hipLaunchKernelGGL((relax9_rows<64, false>), dim3(grid), dim3(64), 0, st, so, qf, q, sor, II, JJ, jb, nrows);
			} else {
				if (down) hipLaunchKernelGGL((relax9_rows<256, true>), dim3(grid), dim3(256), 0, st, so, qf, q, sor, II, JJ, jb, nrows);
				else hipLaunchKernelGGL((relax9_rows<256, false>), dim3(grid), dim3(256), 0, st, so, qf, q, sor, II, JJ, jb, nrows);
			}
		}
	} else {
		for (int c = 0; c < 2; c++) {
			int jo = down ? 2 + c : 3 - c; // LSTART..LEND
			dim3 grid(((II - 2 + 1) / 2 + 255) / 256, JJ - 2);
			hipLaunchKernelGGL(relax5_colour, grid, dim3(256), 0, st, so, qf, q, sor, II, JJ, jo);
		}
	}
}

} // namespace cedar_amd
