// Small 2D levels (at most 64 x 64 unknowns, Dirichlet): each half of a V-cycle visit in ONE launch
// (multilevel.h:170-218; BMG2_SymStd_relax_lines_x.f90:75-176, relax_lines_y.f90:77-176, relax_GS.f90:80-135,
// residual.f90:90-98, restrict.f90, interp_add.f90).
//
// On such a level every kernel of the cycle is latency: two colour launches per line direction plus the two transposes of
// a y sweep (and one of the right-hand side per visit), residual, restriction, a fill, interpolation -- ~7 us each with a
// few hundred points of work; a line-xy V(2,1) visit was some twenty launches.  Plane relaxation runs one such 2D cycle per
// plane colour and 3D level, so its V-cycle was bound by these launches (profiles/r02_plane_relaxation_vcycle.log: 1900
// kernels of ~7 us per cycle at 256^3).
// Here one workgroup owns the level (one per batch item) and the unknowns sit in LDS for the whole launch:
//   visit_pre_small_kernel    pre-smoothing sweeps, residual, its restriction to the coarse right-hand side, coarse x := 0
//   visit_post_small_kernel   interpolation-and-add of the coarse correction, post-smoothing sweeps
//   lines_small_kernel / points_small_kernel   the sweeps alone (smooth() outside a V-cycle visit)
// Line sweeps: per colour the right-hand sides of its lines are formed by all lanes (reference term order), the factors of
// those lines are staged beside them, and each line is solved by ONE lane sequentially -- DPTTRS's own order, so a line solve
// is the reference's to the bit where the scan of the big-level kernels re-associates.  Point sweeps, residual, restriction
// and interpolation use the expressions of the per-launch kernels: bit-identical to them.
#include "common.h"

namespace cedar_amd {

namespace {
constexpr int SMALL_MAX = 64; // unknowns per direction
constexpr int LP = 33;        // lines of one colour (<= 32) + 1: conflict-free [unknown][line] layout

// kind: 1 = x lines, 2 = y lines, 3 = both (DOWN: x then y, UP: y then x); qs = the level in LDS, ghosts included
template <bool NINE>
__device__ __forceinline__ void lines_sweeps(real_t *qs, real_t *ys, real_t *dsv, real_t *esv, const real_t *__restrict__ so,
                                             const real_t *__restrict__ qf, const real_t *__restrict__ sorx,
                                             const real_t *__restrict__ sory, int II, int JJ, int kind, int updown, int nsweeps)
{
	const int tid = threadIdx.x, NT = blockDim.x;
	const size_t PS = (size_t)II * JJ;
	const int nsteps = kind == 3 ? 4 : 2;
	for (int sw = 0; sw < nsweeps; sw++) {
		for (int step = 0; step < nsteps; step++) {
			const bool xdir = kind == 1 || (kind == 3 && ((updown == BMG_DOWN) == (step < 2)));
			const int c = step & 1;
			const int cb = (updown == BMG_DOWN) ? 1 - c : c; // DOWN: lines 3, 5, .. (1-based) first
			const int n = xdir ? II - 2 : JJ - 2;
			const int nlines = ((xdir ? JJ : II) - 2 - cb + 1) / 2;
			if (nlines <= 0) continue; // uniform
			// right-hand sides and factors of the colour's lines
			for (int idx = tid; idx < nlines * n; idx += NT) {
				int t, l, i, j;
				if (xdir) { t = idx % n; l = idx / n; i = t + 1; j = 1 + cb + 2 * l; }
				else { l = idx % nlines; t = idx / nlines; i = 1 + cb + 2 * l; j = t + 1; }
				const int x = i + II * j;
				real_t s = qf[x];
				if (xdir) { // relax_lines_x.f90:106-111
					s = s + so[KS * PS + x] * qs[x - II];
					s = s + so[KS * PS + x + II] * qs[x + II];
					if (NINE) {
						s = s + so[KSW * PS + x] * qs[x - 1 - II];
						s = s + so[KNW * PS + x + 1] * qs[x + 1 - II];
						s = s + so[KNW * PS + x + II] * qs[x - 1 + II];
						s = s + so[KSW * PS + x + 1 + II] * qs[x + 1 + II];
					}
					dsv[t * LP + l] = sorx[(size_t)II * j + 1 + t];
					esv[t * LP + l] = t < n - 1 ? sorx[PS + (size_t)II * j + 2 + t] : 0.0;
				} else { // relax_lines_y.f90:103-107
					s = s + so[KW * PS + x] * qs[x - 1];
					s = s + so[KW * PS + x + 1] * qs[x + 1];
					if (NINE) {
						s = s + so[KSW * PS + x] * qs[x - 1 - II];
						s = s + so[KNW * PS + x + 1] * qs[x + 1 - II];
						s = s + so[KNW * PS + x + II] * qs[x - 1 + II];
						s = s + so[KSW * PS + x + 1 + II] * qs[x + 1 + II];
					}
					dsv[t * LP + l] = sory[(size_t)JJ * i + 1 + t]; // SOR(JJ,II,2)
					esv[t * LP + l] = t < n - 1 ? sory[PS + (size_t)JJ * i + 2 + t] : 0.0;
				}
				ys[t * LP + l] = s;
			}
			__syncthreads();
			// DPTTRS per line: y_t = y_t - e_{t-1} y_{t-1};  x_n = y_n / d_n, x_t = y_t / d_t - e_t x_{t+1}.
			// The two recurrences run on one lane per line; the divisions between them do not depend on the recurrence
			// and are done by all lanes (a double division is ~30 instructions: in the chain it was most of the kernel)
			if (tid < nlines) {
				const int l = tid;
				real_t v = ys[l];
#pragma unroll 8
				for (int t = 1; t < n; t++) {
					v = (-esv[(t - 1) * LP + l]) * v + ys[t * LP + l];
					ys[t * LP + l] = v;
				}
			}
			__syncthreads();
			for (int idx = tid; idx < nlines * n; idx += NT) {
				const int l = idx % nlines, t = idx / nlines;
				ys[t * LP + l] = ys[t * LP + l] / dsv[t * LP + l];
			}
			__syncthreads();
			if (tid < nlines) {
				const int l = tid;
				real_t v = ys[(n - 1) * LP + l];
#pragma unroll 8
				for (int t = n - 2; t >= 0; t--) {
					v = (-esv[t * LP + l]) * v + ys[t * LP + l];
					ys[t * LP + l] = v;
				}
			}
			__syncthreads();
			for (int idx = tid; idx < nlines * n; idx += NT) {
				int t, l, i, j;
				if (xdir) { t = idx % n; l = idx / n; i = t + 1; j = 1 + cb + 2 * l; }
				else { l = idx % nlines; t = idx / nlines; i = 1 + cb + 2 * l; j = t + 1; }
				qs[i + II * j] = ys[t * LP + l];
			}
			__syncthreads();
		}
	}
}

__device__ __forceinline__ void level_load(real_t *qs, const real_t *__restrict__ q, int n)
{
	for (int x = threadIdx.x; x < n; x += blockDim.x) qs[x] = q[x];
	__syncthreads();
}

__device__ __forceinline__ void level_store_interior(const real_t *qs, real_t *__restrict__ q, int II, int JJ)
{
	for (int idx = threadIdx.x; idx < (II - 2) * (JJ - 2); idx += blockDim.x) {
		const int x = 1 + idx % (II - 2) + II * (1 + idx / (II - 2));
		q[x] = qs[x];
	}
}

template <bool NINE>
__global__ __launch_bounds__(1024) void lines_small_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                           real_t *__restrict__ q, const real_t *__restrict__ sorx,
                                                           const real_t *__restrict__ sory, int II, int JJ, int kind,
                                                           int updown, int nsweeps, size_t bstride)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	const int PSi = II * JJ;
	real_t *qs = lds, *ys = qs + PSi, *dsv = ys + SMALL_MAX * LP, *esv = dsv + SMALL_MAX * LP;
	qf += bstride * blockIdx.x; q += bstride * blockIdx.x; // batch item (common.h Batch)
	level_load(qs, q, PSi);
	lines_sweeps<NINE>(qs, ys, dsv, esv, so, qf, sorx, sory, II, JJ, kind, updown, nsweeps);
	level_store_interior(qs, q, II, JJ);
}
} // namespace

namespace {
// Point Gauss-Seidel on such a level (BMG2_SymStd_relax_GS.f90:80-135): the colours of every sweep of the visit in one
// launch.  Same expression per point as relax9_rows / relax5_colour (gs9_mem's term order, times the stored
// reciprocal), and a colour's points do not read one another: bit-identical to the per-colour launches.
template <bool NINE>
__device__ __forceinline__ void points_sweeps(real_t *qs, const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                              const real_t *__restrict__ sor, int II, int JJ, int updown, int nsweeps)
{
	const int tid = threadIdx.x, NT = blockDim.x;
	const size_t PS = (size_t)II * JJ;
	const bool down = updown == BMG_DOWN;
	const int nxh = (II - 2 + 1) / 2; // points of one i-parity per row, at most
	for (int sw = 0; sw < nsweeps; sw++) {
		for (int step = 0; step < (NINE ? 4 : 2); step++) {
			if (NINE) {
				// relax2_sweep9: row class jb = c (DOWN) / 1-c (UP), c = 0, 1; in a row the even 1-based i first on the
				// way down (0-based offset ie = 2p+1), the odd ones first on the way up
				const int c = step >> 1, jb = down ? c : 1 - c;
				const int ib = (down == ((step & 1) == 0)) ? 1 : 2; // 0-based first point of the i-colour
				const int nrows = (JJ - 2 - jb + 1) / 2;
				for (int idx = tid; idx < nrows * nxh; idx += NT) {
					const int i = ib + 2 * (idx % nxh), j = 1 + jb + 2 * (idx / nxh);
					if (i > II - 2) continue;
					const int x = i + II * j;
					real_t s = qf[x];
					s = s + so[KW * PS + x] * qs[x - 1];
					s = s + so[KW * PS + x + 1] * qs[x + 1];
					s = s + so[KS * PS + x] * qs[x - II];
					s = s + so[KS * PS + x + II] * qs[x + II];
					s = s + so[KSW * PS + x] * qs[x - 1 - II];
					s = s + so[KNW * PS + x + 1] * qs[x + 1 - II];
					s = s + so[KNW * PS + x + II] * qs[x - 1 + II];
					s = s + so[KSW * PS + x + 1 + II] * qs[x + 1 + II];
					qs[x] = s * sor[PS + x];
				}
			} else {
				// relax2_gs: colours LSTART..LEND = 2, 3 on the way down, 3, 2 on the way up; colour jo holds the points
				// i1 = mod(j1 + jo, 2) + 2 + 2a of the rows j1 = 2 .. JJ-1 (1-based)
				const int jo = down ? 2 + step : 3 - step;
				for (int idx = tid; idx < (JJ - 2) * nxh; idx += NT) {
					const int j1 = 2 + idx / nxh, i1 = (j1 + jo) % 2 + 2 + 2 * (idx % nxh);
					if (i1 > II - 1) continue;
					const int x = (i1 - 1) + II * (j1 - 1);
					real_t s = qf[x];
					s = s + so[KW * PS + x] * qs[x - 1];
					s = s + so[KW * PS + x + 1] * qs[x + 1];
					s = s + so[KS * PS + x] * qs[x - II];
					s = s + so[KS * PS + x + II] * qs[x + II];
					qs[x] = s * sor[PS + x];
				}
			}
			__syncthreads();
		}
	}
}

template <bool NINE>
__global__ __launch_bounds__(1024) void points_small_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                            real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                            int II, int JJ, int updown, int nsweeps, size_t bstride)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	qf += bstride * blockIdx.x; q += bstride * blockIdx.x;
	level_load(lds, q, II * JJ);
	points_sweeps<NINE>(lds, so, qf, sor, II, JJ, updown, nsweeps);
	level_store_interior(lds, q, II, JJ);
}

// One visit of a small level, first half (multilevel.h:170-199 up to the recursion): pre-smoothing (DOWN), residual,
// restriction of it to the coarse right-hand side, coarse x := 0 -- the expressions of residual2_kernel / residual9_rows and
// restrict2_kernel.  kind: 0 = point relaxation (sorx = reciprocals), 1 / 2 / 3 = lines.
template <bool NINE>
__global__ __launch_bounds__(1024) void visit_pre_small_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                               real_t *__restrict__ q, real_t *__restrict__ res,
                                                               const real_t *__restrict__ sorx, const real_t *__restrict__ sory,
                                                               int II, int JJ, int kind, int nsweeps,
                                                               const real_t *__restrict__ ci, real_t *__restrict__ cb,
                                                               real_t *__restrict__ cx, int IIC, int JJC, size_t bsf, size_t bsc)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	const int PSi = II * JJ;
	real_t *qs = lds, *rs = qs + PSi, *ys = rs + PSi, *dsv = ys + SMALL_MAX * LP, *esv = dsv + SMALL_MAX * LP;
	qf += bsf * blockIdx.x; q += bsf * blockIdx.x; res += bsf * blockIdx.x;
	cb += bsc * blockIdx.x; cx += bsc * blockIdx.x;
	const int tid = threadIdx.x, NT = blockDim.x;
	const size_t PS = (size_t)PSi;
	for (int x = tid; x < PSi; x += NT) rs[x] = res[x]; // the ghost cells of res are read by the restriction
	level_load(qs, q, PSi);
	if (kind == 0) points_sweeps<NINE>(qs, so, qf, sorx, II, JJ, BMG_DOWN, nsweeps);
	else lines_sweeps<NINE>(qs, ys, dsv, esv, so, qf, sorx, sory, II, JJ, kind, BMG_DOWN, nsweeps);
	for (int idx = tid; idx < (II - 2) * (JJ - 2); idx += NT) {
		const int x = 1 + idx % (II - 2) + II * (1 + idx / (II - 2));
		q[x] = qs[x];
		real_t s = qf[x]; // BMG2_SymStd_residual.f90:90-98
		s = s + so[KW * PS + x] * qs[x - 1];
		s = s + so[KW * PS + x + 1] * qs[x + 1];
		s = s + so[KS * PS + x] * qs[x - II];
		s = s + so[KS * PS + x + II] * qs[x + II];
		if (NINE) {
			s = s + so[KSW * PS + x] * qs[x - 1 - II];
			s = s + so[KNW * PS + x + 1] * qs[x + 1 - II];
			s = s + so[KNW * PS + x + II] * qs[x - 1 + II];
			s = s + so[KSW * PS + x + 1 + II] * qs[x + 1 + II];
		}
		s = s - so[KO * PS + x] * qs[x];
		rs[x] = s;
		res[x] = s;
	}
	__syncthreads();
	if (IIC >= 3 && JJC >= 3) {
		const size_t PC = (size_t)IIC * JJC;
		for (int idx = tid; idx < (IIC - 2) * (JJC - 2); idx += NT) { // BMG2_SymStd_restrict.f90, as restrict2_kernel
			const int ic = 1 + idx % (IIC - 2), jc = 1 + idx / (IIC - 2);
			const int c = ic + IIC * jc, f = (2 * ic - 1) + II * (2 * jc - 1);
			real_t s = ci[LNE * PC + c] * rs[f - 1 - II];
			s = s + ci[LA * PC + c] * rs[f - II];
			s = s + ci[LNW * PC + c + 1] * rs[f + 1 - II];
			s = s + ci[LR * PC + c] * rs[f - 1];
			s = s + rs[f];
			s = s + ci[LL * PC + c + 1] * rs[f + 1];
			s = s + ci[LSE * PC + c + IIC] * rs[f - 1 + II];
			s = s + ci[LB * PC + c + IIC] * rs[f + II];
			s = s + ci[LSW * PC + c + 1 + IIC] * rs[f + 1 + II];
			cb[c] = s;
		}
	}
	for (int c = tid; c < IIC * JJC; c += NT) cx[c] = 0.0; // coarse_x.set(0.0)
}

// second half of the visit: interpolation-and-add of the coarse correction (BMG2_SymStd_interp_add.f90, as
// interp_add2_kernel: res becomes res / diagonal on the way), post-smoothing (UP)
template <bool NINE>
__global__ __launch_bounds__(1024) void visit_post_small_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                                real_t *__restrict__ q, real_t *__restrict__ res,
                                                                const real_t *__restrict__ sorx, const real_t *__restrict__ sory,
                                                                int II, int JJ, int kind, int nsweeps,
                                                                const real_t *__restrict__ ci, const real_t *__restrict__ cx,
                                                                int IIC, int JJC, size_t bsf, size_t bsc)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	const int PSi = II * JJ;
	real_t *qs = lds, *ys = qs + PSi, *dsv = ys + SMALL_MAX * LP, *esv = dsv + SMALL_MAX * LP;
	qf += bsf * blockIdx.x; q += bsf * blockIdx.x; res += bsf * blockIdx.x;
	cx += bsc * blockIdx.x;
	const int tid = threadIdx.x, NT = blockDim.x;
	level_load(qs, q, PSi);
	{
		const int IIF = II, JJF = JJ;
		const int imax = 2 * ((IIF - 2) / 2 + 2 - 1), jmax = 2 * ((JJF - 2) / 2 + 2 - 1);
		const size_t PC = (size_t)IIC * JJC;
		for (int idx = tid; idx < (IIF - 1) * (JJF - 1); idx += NT) {
			const int i = 2 + idx % (IIF - 1), j = 2 + idx / (IIF - 1); // 1-based fine indices as in the reference
			const int x = (i - 1) + IIF * (j - 1);
			const bool interior = i <= IIF - 1 && j <= JJF - 1;
			real_t r = 0.0;
			if (interior || (i <= imax && j <= jmax)) r = res[x];
			if (interior) {
				r = r / so[x]; // KO plane
				res[x] = r;
			}
			if (i > imax || j > jmax) continue;
			const bool io = i & 1, jo = j & 1;
			const int ic = io ? (i + 1) / 2 + 1 : i / 2 + 1, jc = jo ? (j + 1) / 2 + 1 : j / 2 + 1; // 1-based coarse
			const int c = (ic - 1) + IIC * (jc - 1);
			real_t v = qs[x];
			if (!io && !jo) {
				v = v + cx[c];
			} else if (io && !jo) {
				real_t a = ci[LR * PC + c] * cx[c] + ci[LL * PC + c] * cx[c - 1];
				v = v + a + r;
			} else if (!io && jo) {
				real_t a = ci[LA * PC + c] * cx[c] + ci[LB * PC + c] * cx[c - IIC];
				v = v + a + r;
			} else {
				real_t a = ci[LSW * PC + c] * cx[c - 1 - IIC] + ci[LNW * PC + c] * cx[c - 1]
				           + ci[LNE * PC + c] * cx[c] + ci[LSE * PC + c] * cx[c - IIC];
				v = v + a + r;
			}
			qs[x] = v; // every point is written by one lane and read by none in this loop
		}
	}
	__syncthreads();
	if (kind == 0) points_sweeps<NINE>(qs, so, qf, sorx, II, JJ, BMG_UP, nsweeps);
	else lines_sweeps<NINE>(qs, ys, dsv, esv, so, qf, sorx, sory, II, JJ, kind, BMG_UP, nsweeps);
	for (int x = tid; x < PSi; x += NT) q[x] = qs[x]; // with the ghost column / row interp_add touches for even extents
}
} // namespace

// kind: 0 point (sorx: the reciprocals), 1 / 2 / 3 lines; pre != 0: first half of the visit, else the second
void visit_small(int pre, const real_t *so, const real_t *qf, real_t *q, real_t *res, const real_t *sorx, const real_t *sory,
                 int II, int JJ, int nstncl, int kind, int nsweeps, const real_t *ci, real_t *cb, real_t *cx, int IIC, int JJC,
                 hipStream_t st, Batch bf, Batch bc)
{
	// per DEVICE: the attribute belongs to the function on the current device, and a process may switch devices
	// (cedar_amd_set_device); 121 KB for a 66 x 66 level with line relaxation
	static bool attr_dev[64] = {false};
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	bool &attr = attr_dev[dev_ & 63];
	if (!attr) {
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)visit_pre_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)visit_pre_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)visit_post_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)visit_post_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
		attr = true;
	}
	const size_t lines = kind ? 3 * (size_t)SMALL_MAX * LP : 0;
	const size_t shm = ((size_t)II * JJ * (pre ? 2 : 1) + lines) * sizeof(real_t);
	const int work = ((II > JJ ? II : JJ) - 2) * (((II < JJ ? II : JJ) - 2 + 1) / 2);
	const int nthr = work > 512 ? 1024 : work > 256 ? 512 : 256;
	if (pre) {
		if (nstncl == 5)
			hipLaunchKernelGGL(visit_pre_small_kernel<true>, dim3(bf.n), dim3(nthr), shm, st, so, qf, q, res, sorx, sory, II, JJ, kind,
			                   nsweeps, ci, cb, cx, IIC, JJC, bf.stride, bc.stride);
		else
			hipLaunchKernelGGL(visit_pre_small_kernel<false>, dim3(bf.n), dim3(nthr), shm, st, so, qf, q, res, sorx, sory, II, JJ, kind,
			                   nsweeps, ci, cb, cx, IIC, JJC, bf.stride, bc.stride);
	} else {
		if (nstncl == 5)
			hipLaunchKernelGGL(visit_post_small_kernel<true>, dim3(bf.n), dim3(nthr), shm, st, so, qf, q, res, sorx, sory, II, JJ, kind,
			                   nsweeps, ci, cx, IIC, JJC, bf.stride, bc.stride);
		else
			hipLaunchKernelGGL(visit_post_small_kernel<false>, dim3(bf.n), dim3(nthr), shm, st, so, qf, q, res, sorx, sory, II, JJ, kind,
			                   nsweeps, ci, cx, IIC, JJC, bf.stride, bc.stride);
	}
}

void relax_points_small(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int nstncl,
                        int updown, int nsweeps, hipStream_t st, Batch bt)
{
	if (nsweeps <= 0) return;
	const size_t shm = (size_t)II * JJ * sizeof(real_t); // 35 KB for a 66 x 66 level
	const int work = ((II - 2 + 1) / 2) * ((JJ - 2 + 1) / 2);
	const int nthr = work > 512 ? 1024 : work > 256 ? 512 : 256;
	if (nstncl == 5)
		hipLaunchKernelGGL(points_small_kernel<true>, dim3(bt.n), dim3(nthr), shm, st, so, qf, q, sor, II, JJ, updown, nsweeps, bt.stride);
	else
		hipLaunchKernelGGL(points_small_kernel<false>, dim3(bt.n), dim3(nthr), shm, st, so, qf, q, sor, II, JJ, updown, nsweeps, bt.stride);
}

// Dirichlet levels of at most 64 x 64 unknowns (CEDAR_AMD_LINES_SMALL=0: the per-colour kernels everywhere)
bool lines_small_ok(int II, int JJ)
{
	const char *e = getenv("CEDAR_AMD_LINES_SMALL"); // read per call: a solver records its choice in its cycle graph
	const bool off = e && atoi(e) == 0;
	return !off && II >= 3 && JJ >= 3 && II - 2 <= SMALL_MAX && JJ - 2 <= SMALL_MAX;
}

// kind: 1 = x lines (sorx), 2 = y lines (sory), 3 = line-xy; nsweeps sweeps in the direction `updown`
void relax_lines_small(const real_t *so, const real_t *qf, real_t *q, const real_t *sorx, const real_t *sory,
                       int II, int JJ, int nstncl, int kind, int updown, int nsweeps, hipStream_t st, Batch bt)
{
	if (nsweeps <= 0) return;
	const size_t shm = ((size_t)II * JJ + 3 * (size_t)SMALL_MAX * LP) * sizeof(real_t);
	static bool attr_dev[64] = {false}; // per device, see visit_small; 86 KB for a 66 x 66 level
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	bool &attr = attr_dev[dev_ & 63];
	if (!attr) {
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)lines_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)lines_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
		attr = true;
	}
	// lanes for the right-hand sides: up to 32 lines x 64 unknowns per colour
	const int work = ((II > JJ ? II : JJ) - 2) * (((II < JJ ? II : JJ) - 2 + 1) / 2);
	const int nthr = work > 512 ? 1024 : work > 256 ? 512 : 256;
	if (nstncl == 5)
		hipLaunchKernelGGL(lines_small_kernel<true>, dim3(bt.n), dim3(nthr), shm, st, so, qf, q, sorx, sory, II, JJ, kind, updown,
		                   nsweeps, bt.stride);
	else
		hipLaunchKernelGGL(lines_small_kernel<false>, dim3(bt.n), dim3(nthr), shm, st, so, qf, q, sorx, sory, II, JJ, kind, updown,
		                   nsweeps, bt.stride);
}

} // namespace cedar_amd
