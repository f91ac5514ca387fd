// Zebra line relaxation in 2D (x- and y-lines) and its factorisation set-up.
// Replaces BMG2_SymStd_SETUP_lines_x/_y (src/2d/ftn/BMG2_SymStd_SETUP_lines_x.f90:68-87,
// ..._y.f90:69-87), BMG2_SymStd_relax_lines_x (..._relax_lines_x.f90:82-162) and
// BMG2_SymStd_relax_lines_y (..._relax_lines_y.f90:77-168), plus the LAPACK
// calls inside them (DPTTRF / DPTTRS; system LAPACK in the reference, not
// vendored -- the published netlib recurrences are restated here:
//   factor: e_i <- e_i/d_i ; d_{i+1} <- d_{i+1} - e_i*(e_i d_i)
//   solve : b_i <- b_i - b_{i-1} e_{i-1} ;  b_n <- b_n/d_n ; b_i <- b_i/d_i - b_{i+1} e_i ).
//
// Set-up (once per solve): one lane per line runs the DPTTRF recurrence
// sequentially in the reference's operation order => bit-identical factors.
//
// Relaxation (hot path, 64 algorithmic B/DOF per direction): all lines of one
// colour are independent.  One workgroup owns one line held in LDS (<= 64 KB);
// both DPTTRS sweeps are first-order affine recurrences y_i = a_i y_{i-1} + c_i.
// Every lane owns 8 consecutive unknowns (sequential, reference order); the
// lanes' composite maps are combined with a Kogge-Stone scan over wavefront
// shuffles, a 4-entry cross-wave fix-up in LDS and a scalar carry between
// tiles of 2048 unknowns.  The scan re-associates the recurrence, so
// results agree with the sequential DPTTRS to rounding (like one vendor LAPACK
// against another), not bit-for-bit; tolerances are stated in tests/.
// y-lines are strided in memory: their right-hand sides are gathered through
// an LDS tile transpose into a line-contiguous scratch (the same transposed
// layout the reference uses for the y factors, SOR(JJ,II,2)), solved by the
// same kernel, and scattered back.
#include "common.h"

namespace cedar_amd {

// ------------------------------------------------------------------ set-up
__global__ void lines_fill_x(const real_t *__restrict__ so, real_t *__restrict__ sor, int II, int JJ)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1;
	if (i > II - 2) return;
	const size_t PS = (size_t)II * JJ, x = (size_t)i + (size_t)II * j;
	sor[PS + x] = -so[KW * PS + x];
	sor[x] = so[KO * PS + x];
}

__global__ void lines_fill_y(const real_t *__restrict__ so, real_t *__restrict__ sor, int II, int JJ)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1;
	if (i > II - 2) return;
	const size_t PS = (size_t)II * JJ, x = (size_t)i + (size_t)II * j;
	const size_t xt = (size_t)j + (size_t)JJ * i; // SOR(JJ,II,2)
	sor[PS + xt] = -so[KS * PS + x];
	sor[xt] = so[KO * PS + x];
}

// one lane per line; d = sor(:,line,1) from index 1 (0-based), e = sor(:,line,2) from index 2
__global__ void lines_factor(real_t *__restrict__ sor, int n /*unknowns*/, int ld /*line stride*/, int nlines, size_t PS)
{
	const int l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= nlines) return;
	real_t *d = sor + (size_t)ld * (l + 1) + 1;
	real_t *e = sor + PS + (size_t)ld * (l + 1) + 2;
	real_t di = d[0];
	for (int i = 0; i < n - 1; i++) {
		if (di <= 0.0) return; // DPTTRF: INFO = i+1, factorisation stops
		const real_t ei = e[i];
		const real_t en = ei / di;
		e[i] = en;
		di = d[i + 1] - en * ei;
		d[i + 1] = di;
	}
}

// periodic lines: the wrap-around coupling is folded into the two end diagonals before the
// factorisation (SETUP_lines_x.f90:76-83, SETUP_lines_y.f90:78-85); the solves undo it with a
// Sherman-Morrison correction (relax_lines_x.f90:209-226)
__global__ void lines_fold_x(const real_t *__restrict__ so, real_t *__restrict__ sor, int II, int JJ)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x + 1; // 0-based interior row
	if (j > JJ - 2) return;
	const size_t PS = (size_t)II * JJ, r = (size_t)II * j;
	sor[r + 1] = sor[r + 1] + so[KW * PS + r + 1];
	sor[r + II - 2] = sor[r + II - 2] + so[KW * PS + r + II - 1];
}

__global__ void lines_fold_y(const real_t *__restrict__ so, real_t *__restrict__ sor, int II, int JJ)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x + 1;
	if (i > II - 2) return;
	const size_t PS = (size_t)II * JJ, c = (size_t)JJ * i; // SOR(JJ,II,2): column i
	sor[c + 1] = sor[c + 1] + so[KS * PS + (size_t)i + (size_t)II];
	sor[c + JJ - 2] = sor[c + JJ - 2] + so[KS * PS + (size_t)i + (size_t)II * (JJ - 1)];
}

void setup_lines_x(const real_t *so, real_t *sor, int II, int JJ, hipStream_t st, int fold)
{
	if (II < 3 || JJ < 3) return;
	dim3 grid((II - 2 + 255) / 256, JJ - 2);
	hipLaunchKernelGGL(lines_fill_x, grid, dim3(256), 0, st, so, sor, II, JJ);
	if (fold) hipLaunchKernelGGL(lines_fold_x, dim3((JJ - 2 + 63) / 64), dim3(64), 0, st, so, sor, II, JJ);
	hipLaunchKernelGGL(lines_factor, dim3((JJ - 2 + 63) / 64), dim3(64), 0, st, sor, II - 2, II, JJ - 2, (size_t)II * JJ);
}

void setup_lines_y(const real_t *so, real_t *sor, int II, int JJ, hipStream_t st, int fold)
{
	if (II < 3 || JJ < 3) return;
	dim3 grid((II - 2 + 255) / 256, JJ - 2);
	hipLaunchKernelGGL(lines_fill_y, grid, dim3(256), 0, st, so, sor, II, JJ);
	if (fold) hipLaunchKernelGGL(lines_fold_y, dim3((II - 2 + 63) / 64), dim3(64), 0, st, so, sor, II, JJ);
	hipLaunchKernelGGL(lines_factor, dim3((II - 2 + 63) / 64), dim3(64), 0, st, sor, JJ - 2, JJ, II - 2, (size_t)II * JJ);
}

// ------------------------------------------------------------------ affine scan
// Every lane owns CH consecutive unknowns of the line.  A sweep over a tile of BS*CH unknowns is
//   (1) lane-local: compose the CH maps  y -> a*y + c  of the chunk            (sequential, exact order)
//   (2) workgroup scan of the BS composites: Kogge-Stone over wavefront shuffles, cross-wave fix-up
//       through LDS, scalar carry from the previous tile
//   (3) lane-local: re-run the chunk from the value entering it.
// Only step (2) re-associates the recurrence.
constexpr int CH = 8;
__device__ __forceinline__ int lpad(int i) { return i + (i >> 3); } // LDS index: one pad per 8 -> stride-9 chunks, no bank conflicts

// (p[0], p[1]) with one 16-byte load (8-byte aligned)
__device__ __forceinline__ void ldpair2(const real_t *__restrict__ p, real_t *out)
{
	const d2u v = *reinterpret_cast<const d2u *>(p);
	out[0] = v.x; out[1] = v.y;
}

template <int BS>
__device__ __forceinline__ void affine_scan(real_t a, real_t c, real_t carry, real_t *wa, real_t *wc,
                                            real_t &y_in, real_t &y_last)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const real_t ap = __shfl_up(a, off, 64), cp = __shfl_up(c, off, 64);
		if (lane >= off) {
			c = a * cp + c;
			a = a * ap;
		}
	}
	real_t ae = __shfl_up(a, 1, 64), ce = __shfl_up(c, 1, 64); // composite of the lanes before this one
	if (lane == 0) { ae = 1.0; ce = 0.0; }
	constexpr int NW = BS / 64;
	if (NW > 1) {
		if (lane == 63) { wa[w] = a; wc[w] = c; }
		__syncthreads();
		real_t y = carry;
		for (int u = 0; u < w; u++) y = wa[u] * y + wc[u];
		carry = y;
	}
	y_in = ae * carry + ce;
	y_last = a * carry + c;
}

// Solve L D L^T x = y for the line held in LDS (padded index lpad(i)); d[0..n), e[0..n-1) in HBM.
template <int BS>
__device__ __forceinline__ void line_pttrs(real_t *y, int n, const real_t *__restrict__ d, const real_t *__restrict__ e,
                                           real_t *wa, real_t *wc, real_t *carry_slot)
{
	// forward: y_i = y_i - e_{i-1} y_{i-1}
	real_t carry = 0.0;
	for (int base = 0; base < n; base += BS * CH) {
		const int i0 = base + (int)threadIdx.x * CH;
		real_t a[CH], c[CH];
#pragma unroll
		for (int m = 0; m < CH; m++) {
			const int i = i0 + m;
			if (i < n) { c[m] = y[lpad(i)]; a[m] = i > 0 ? -e[i - 1] : 0.0; }
			else { c[m] = 0.0; a[m] = 0.0; }
		}
		real_t A = 1.0, Cc = 0.0;
#pragma unroll
		for (int m = 0; m < CH; m++) { Cc = a[m] * Cc + c[m]; A = a[m] * A; }
		real_t v, last;
		affine_scan<BS>(A, Cc, carry, wa, wc, v, last);
#pragma unroll
		for (int m = 0; m < CH; m++) {
			v = a[m] * v + c[m];
			if (i0 + m < n) y[lpad(i0 + m)] = v;
		}
		if (threadIdx.x == BS - 1) *carry_slot = last;
		__syncthreads();
		carry = *carry_slot;
		__syncthreads();
	}
	// backward: x_i = y_i/d_i - e_i x_{i+1}; reversed position r <-> i = n-1-r
	carry = 0.0;
	for (int base = 0; base < n; base += BS * CH) {
		const int r0 = base + (int)threadIdx.x * CH;
		real_t a[CH], c[CH];
#pragma unroll
		for (int m = 0; m < CH; m++) {
			const int r = r0 + m, i = n - 1 - r;
			if (r < n) { c[m] = y[lpad(i)] / d[i]; a[m] = r > 0 ? -e[i] : 0.0; }
			else { c[m] = 0.0; a[m] = 0.0; }
		}
		real_t A = 1.0, Cc = 0.0;
#pragma unroll
		for (int m = 0; m < CH; m++) { Cc = a[m] * Cc + c[m]; A = a[m] * A; }
		real_t v, last;
		affine_scan<BS>(A, Cc, carry, wa, wc, v, last);
#pragma unroll
		for (int m = 0; m < CH; m++) {
			v = a[m] * v + c[m];
			if (r0 + m < n) y[lpad(n - 1 - r0 - m)] = v;
		}
		if (threadIdx.x == BS - 1) *carry_slot = last;
		__syncthreads();
		carry = *carry_slot;
		__syncthreads();
	}
}

// The same solve on a scan-ordered copy of the factors (resident solver, lines longer than a wavefront tile).
// line_pttrs reads e / d where the scan needs them: lane L owns unknowns 8L .. 8L+7 of a tile, so one wave
// instruction gathers 64 x 8 bytes spread over 4 KB -- 32 cache lines for 512 useful bytes -- and the sweeps of a
// line wait on eight such gathers.  Measured at 8192^2 (profiles/r02_experiment_line_relax.log): 0.52 of the 1.34 ms
// of an x sweep is this solve, 0.22 ms of it the gathers alone.  The copy stores, per line and tile of BS*CH unknowns,
// the forward multipliers a = -e(i-1), the backward multipliers a = -e(i) and the pivots d(i) in the order the lanes
// consume them: [tile][kind][m/2][lane][m%2], so every load is a coalesced 16-byte-per-lane stream, and the next
// tile's factors are requested before the current tile is scanned.  Same values, same chunks, same scan =>
// bit-identical to line_pttrs.
template <int BS> __host__ __device__ inline size_t pf_tiles(int n) { return (size_t)(n + BS * CH - 1) / (BS * CH); }
template <int BS> __host__ __device__ inline size_t pf_line_doubles(int n) { return pf_tiles<BS>(n) * 3 * CH * BS; }
template <int BS> __device__ __forceinline__ size_t pf_off(int t, int kind, int m, int lane)
{
	return ((((size_t)t * 3 + kind) * (CH / 2) + (m >> 1)) * BS + lane) * 2 + (m & 1);
}

template <int BS>
__global__ __launch_bounds__(BS) void lines_permute_kernel(const real_t *__restrict__ sor, real_t *__restrict__ pf,
                                                            int n, int ld, size_t PS)
{
	const int l = blockIdx.x;
	const real_t *d = sor + (size_t)ld * (l + 1) + 1, *e = sor + PS + (size_t)ld * (l + 1) + 2;
	real_t *out = pf + (size_t)l * pf_line_doubles<BS>(n);
	const int nt = (int)pf_tiles<BS>(n);
	for (int t = 0; t < nt; t++)
		for (int m = 0; m < CH; m++) {
			const int r = t * BS * CH + (int)threadIdx.x * CH + m, i = n - 1 - r; // forward position r, backward unknown i
			out[pf_off<BS>(t, 0, m, threadIdx.x)] = (r < n && r > 0) ? -e[r - 1] : 0.0;
			out[pf_off<BS>(t, 1, m, threadIdx.x)] = (r < n && r > 0) ? -e[i] : 0.0;
			out[pf_off<BS>(t, 2, m, threadIdx.x)] = r < n ? d[i] : 1.0;
		}
}

template <int BS>
__device__ __forceinline__ void pf_load(const real_t *__restrict__ pfl, int t, int kind, real_t (&v)[CH])
{
#pragma unroll
	for (int mp = 0; mp < CH / 2; mp++) {
		const d2u w = *reinterpret_cast<const d2u *>(pfl + pf_off<BS>(t, kind, 2 * mp, threadIdx.x));
		v[2 * mp] = w.x; v[2 * mp + 1] = w.y;
	}
}

template <int BS>
__device__ __forceinline__ void line_pttrs_pf(real_t *y, int n, const real_t *__restrict__ pfl,
                                              real_t *wa, real_t *wc, real_t *carry_slot)
{
	const int nt = (int)pf_tiles<BS>(n);
	real_t a[CH], an[CH], dn[CH], dd[CH];
	pf_load<BS>(pfl, 0, 0, a);
	real_t carry = 0.0;
	for (int t = 0; t < nt; t++) {
		if (t + 1 < nt) pf_load<BS>(pfl, t + 1, 0, an); // next tile, or ...
		else { pf_load<BS>(pfl, 0, 1, an); pf_load<BS>(pfl, 0, 2, dn); } // ... the first backward tile
		const int i0 = t * BS * CH + (int)threadIdx.x * CH;
		real_t c[CH];
#pragma unroll
		for (int m = 0; m < CH; m++) c[m] = i0 + m < n ? y[lpad(i0 + m)] : 0.0;
		real_t A = 1.0, Cc = 0.0;
#pragma unroll
		for (int m = 0; m < CH; m++) { Cc = a[m] * Cc + c[m]; A = a[m] * A; }
		real_t v, last;
		affine_scan<BS>(A, Cc, carry, wa, wc, v, last);
#pragma unroll
		for (int m = 0; m < CH; m++) {
			v = a[m] * v + c[m];
			if (i0 + m < n) y[lpad(i0 + m)] = v;
		}
		if (threadIdx.x == BS - 1) *carry_slot = last;
		__syncthreads();
		carry = *carry_slot;
		__syncthreads();
#pragma unroll
		for (int m = 0; m < CH; m++) a[m] = an[m];
	}
#pragma unroll
	for (int m = 0; m < CH; m++) dd[m] = dn[m];
	carry = 0.0;
	for (int t = 0; t < nt; t++) {
		if (t + 1 < nt) { pf_load<BS>(pfl, t + 1, 1, an); pf_load<BS>(pfl, t + 1, 2, dn); }
		const int r0 = t * BS * CH + (int)threadIdx.x * CH;
		real_t c[CH];
#pragma unroll
		for (int m = 0; m < CH; m++) {
			const int r = r0 + m;
			c[m] = r < n ? y[lpad(n - 1 - r)] / dd[m] : 0.0;
		}
		real_t A = 1.0, Cc = 0.0;
#pragma unroll
		for (int m = 0; m < CH; m++) { Cc = a[m] * Cc + c[m]; A = a[m] * A; }
		real_t v, last;
		affine_scan<BS>(A, Cc, carry, wa, wc, v, last);
#pragma unroll
		for (int m = 0; m < CH; m++) {
			v = a[m] * v + c[m];
			if (r0 + m < n) y[lpad(n - 1 - r0 - m)] = v;
		}
		if (threadIdx.x == BS - 1) *carry_slot = last;
		__syncthreads();
		carry = *carry_slot;
		__syncthreads();
#pragma unroll
		for (int m = 0; m < CH; m++) { a[m] = an[m]; dd[m] = dn[m]; }
	}
}

// Generic first-order recurrence over line-contiguous data, one workgroup per line:
//   forward  (reverse = 0): y_i = a_i * y_{i-1} + c_i,  i = 0..n-1,   y_{-1} = 0
//   backward (reverse = 1): y_i = a_i * y_{i+1} + c_i,  i = n-1..0,   y_n    = 0
// with c_i = y_i on entry, divided by div_i when div != nullptr.  The two sweeps of DPTTRS are the cases
// (a = -e', div = nullptr) and (a = -e'_next, div = d', reverse); the domain-decomposed line relaxation
// (cedar_amd/dist2d.py) runs them per segment and joins the segments' carries across ranks.
template <int BS>
__global__ __launch_bounds__(BS) void affine_lines_kernel(real_t *__restrict__ y, const real_t *__restrict__ a,
                                                           const real_t *__restrict__ div, int n, int ld, int reverse)
{
	__shared__ real_t wa[4], wc[4], cs;
	real_t *line = y + (size_t)blockIdx.x * ld;
	const real_t *al = a + (size_t)blockIdx.x * ld;
	const real_t *dl = div ? div + (size_t)blockIdx.x * ld : nullptr;
	real_t carry = 0.0;
	for (int base = 0; base < n; base += BS * CH) {
		const int r0 = base + (int)threadIdx.x * CH;
		real_t am[CH], cm[CH];
#pragma unroll
		for (int m = 0; m < CH; m++) {
			const int r = r0 + m, i = reverse ? n - 1 - r : r;
			if (r < n) { cm[m] = dl ? line[i] / dl[i] : line[i]; am[m] = al[i]; }
			else { cm[m] = 0.0; am[m] = 0.0; }
		}
		real_t A = 1.0, Cc = 0.0;
#pragma unroll
		for (int m = 0; m < CH; m++) { Cc = am[m] * Cc + cm[m]; A = am[m] * A; }
		real_t v, last;
		affine_scan<BS>(A, Cc, carry, wa, wc, v, last);
#pragma unroll
		for (int m = 0; m < CH; m++) {
			v = am[m] * v + cm[m];
			const int r = r0 + m;
			if (r < n) line[reverse ? n - 1 - r : r] = v;
		}
		if (threadIdx.x == BS - 1) cs = last;
		__syncthreads();
		carry = cs;
		__syncthreads();
	}
}

void affine_lines(real_t *y, const real_t *a, const real_t *div, int nlines, int n, int ld, int reverse, hipStream_t st)
{
	if (nlines <= 0 || n <= 0) return;
	if (n <= 512) hipLaunchKernelGGL(affine_lines_kernel<64>, dim3(nlines), dim3(64), 0, st, y, a, div, n, ld, reverse);
	else hipLaunchKernelGGL(affine_lines_kernel<256>, dim3(nlines), dim3(256), 0, st, y, a, div, n, ld, reverse);
}

// doubles of LDS a line of n unknowns needs (padded line + scan scratch)
static inline size_t line_lds_doubles(int n) { return (size_t)n + (size_t)(n >> 3) + 40; } // line + 16 + 16 wave aggregates + carry

// ------------------------------------------------------------------ x-lines
// Sherman-Morrison closure of a cyclic line held in LDS: y holds the solution of the folded system;
// a second solve with the rank-one column u (-c_first at the first, -c_last at the last unknown),
// alpha = u_1 + u_n, beta = (y_1 + y_n) / (1 + alpha), x = y - beta u (relax_lines_x.f90:212-226).
// On return `y` holds u and every lane holds beta; ys[t] (lane-strided) kept the first solution.
template <int BS>
__device__ __forceinline__ real_t line_sherman_morrison(real_t *y, int n, const real_t *__restrict__ d,
                                                        const real_t *__restrict__ e, real_t cfirst, real_t clast,
                                                        real_t *wa, real_t *wc, real_t *cs, real_t &y1, real_t &yn)
{
	y1 = y[lpad(0)];
	yn = y[lpad(n - 1)];
	__syncthreads();
	for (int t = threadIdx.x; t < n; t += BS) y[lpad(t)] = 0.0;
	__syncthreads();
	if (threadIdx.x == 0) {
		y[lpad(0)] = -cfirst;
		y[lpad(n - 1)] = -clast; // n == 1: the second assignment wins, as in the reference
	}
	__syncthreads();
	line_pttrs<BS>(y, n, d, e, wa, wc, cs);
	const real_t alpha = y[lpad(0)] + y[lpad(n - 1)];
	real_t beta = y1 + yn;
	beta = beta / (1.0 + alpha);
	return beta;
}

// YT: the arrays are the transposed ones of a y-line sweep (setup_lines_yt): rows are the y lines, and the two
// cross terms are added in relax_lines_y.f90's order
template <int BS, bool NINE, bool SM = false, bool YT = false, bool PERM = false>
__global__ __launch_bounds__(BS) void relax_lines_x_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                            real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                            int II, int JJ, int jb, int nlines, int dbg,
                                                            const real_t *__restrict__ pf, size_t bstride)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	const int npad = (II - 2) + ((II - 2) >> 3) + 1;
	real_t *y = lds, *wa = lds + npad, *wc = wa + 16, *cs = wc + 16;
	const unsigned L = xcd_remap(blockIdx.x, (unsigned)nlines);
	if (L >= (unsigned)nlines) return;
	qf += bstride * blockIdx.y; q += bstride * blockIdx.y; // batch item (common.h Batch)
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t row = (size_t)(1 + jb + 2 * (int)L) * sj;
	const int n = II - 2;
	// right-hand side (relax_lines_x.f90:106-111 / :128-129), reference term order.
	// Lines longer than a wavefront tile: every lane forms the right-hand side of RU pairs of unknowns per pass
	// with 16-byte loads, all 11 x RU loads of a pass requested before the first is used.  With one point per lane
	// and pass (below) a wave waits out one HBM round trip per 64 unknowns -- 72 % of the wave cycles of this
	// kernel were such waits (profiles/r02_lines_sq_counters.txt).
	int tdone = 0;
	if (dbg & 2) tdone = n; // experiment: no right-hand side (LDS left as is)
	else if (BS >= 256) {
		constexpr int RU = 4;
		const int npair = n >> 1;
		for (int pb0 = 0; pb0 < npair; pb0 += BS * RU) {
			real_t f[RU][2], ks[RU][2], ksn[RU][2], ksw[RU][2], knwe[RU][2], knwn[RU][2], kswne[RU][2], qs[RU][4], qn[RU][4];
#pragma unroll
			for (int u = 0; u < RU; u++) {
				const int pr = pb0 + u * BS + (int)threadIdx.x;
				if (pr < npair) {
					const size_t x = row + 1 + 2 * (size_t)pr;
					ldpair2(qf + x, f[u]);
					ldpair2(so + KS * PS + x, ks[u]);
					ldpair2(so + KS * PS + x + sj, ksn[u]);
					ldpair2(q + x - 1 - sj, &qs[u][0]); ldpair2(q + x + 1 - sj, &qs[u][2]);
					ldpair2(q + x - 1 + sj, &qn[u][0]); ldpair2(q + x + 1 + sj, &qn[u][2]);
					if (NINE) {
						ldpair2(so + KSW * PS + x, ksw[u]);
						ldpair2(so + KNW * PS + x + 1, knwe[u]);
						ldpair2(so + KNW * PS + x + sj, knwn[u]);
						ldpair2(so + KSW * PS + x + 1 + sj, kswne[u]);
					}
				}
			}
#pragma unroll
			for (int u = 0; u < RU; u++) {
				const int pr = pb0 + u * BS + (int)threadIdx.x;
				if (pr < npair) {
#pragma unroll
					for (int h = 0; h < 2; h++) {
						real_t s = f[u][h];
						s = s + ks[u][h] * qs[u][1 + h];
						s = s + ksn[u][h] * qn[u][1 + h];
						if (NINE) {
							s = s + ksw[u][h] * qs[u][h];
							if (YT) {
								s = s + knwn[u][h] * qn[u][h];
								s = s + knwe[u][h] * qs[u][2 + h];
							} else {
								s = s + knwe[u][h] * qs[u][2 + h];
								s = s + knwn[u][h] * qn[u][h];
							}
							s = s + kswne[u][h] * qn[u][2 + h];
						}
						y[lpad(2 * pr + h)] = s;
					}
				}
			}
		}
		tdone = npair * 2; // an odd last unknown goes through the scalar loop
	}
	for (int t = tdone + threadIdx.x; t < n; t += BS) {
		const size_t x = row + 1 + t;
		real_t s = qf[x];
		s = s + so[KS * PS + x] * q[x - sj];
		s = s + so[KS * PS + x + sj] * q[x + sj];
		if (NINE) {
			s = s + so[KSW * PS + x] * q[x - 1 - sj];
			if (YT) { // (i+1,j-1) is (row+1, col-1) of the transposed arrays and comes first
				s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
				s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
			} else {
				s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
				s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
			}
			s = s + so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
		}
		y[lpad(t)] = s;
	}
	__syncthreads();
	if (dbg & 1) { /* experiment: no solve */ }
	else if (PERM) line_pttrs_pf<BS>(y, n, pf + (size_t)(jb + 2 * (int)L) * pf_line_doubles<BS>(n), wa, wc, cs);
	else line_pttrs<BS>(y, n, sor + row + 1, sor + PS + row + 2, wa, wc, cs);
	for (int t = threadIdx.x; t < n; t += BS) q[row + 1 + t] = y[lpad(t)];
	if (SM) {
		real_t y1, yn;
		const real_t beta = line_sherman_morrison<BS>(y, n, sor + row + 1, sor + PS + row + 2, so[KW * PS + row + 1],
		                                              so[KW * PS + row + II - 1], wa, wc, cs, y1, yn);
		for (int t = threadIdx.x; t < n; t += BS) q[row + 1 + t] = q[row + 1 + t] - beta * y[lpad(t)]; // same lane wrote it
	}
}

// lanes per line: one wavefront up to 512 unknowns, else CEDAR_AMD_LINE_BS (256 / 512 / 1024).  The size fixes the
// association of the scan, so every launcher of a line solve takes it from here.
static int line_bs(int n)
{
	if (n <= 512) return 64;
	static const int e = getenv("CEDAR_AMD_LINE_BS") ? atoi(getenv("CEDAR_AMD_LINE_BS")) : 256;
	return (e == 512 || e == 1024) ? e : 256;
}

// a line must fit the 160 KB of LDS (up to ~18,000 unknowns); longer lines are refused through the
// host's print_error callback like every other unsupported request, the sweep is skipped
extern "C" void print_error(char *msg);
static bool lds_ok(int n, const char *who)
{
	if (line_lds_doubles(n) * sizeof(real_t) > 160 * 1024 - 512) {
		char buf[160];
		snprintf(buf, sizeof(buf), "%s: a line of %d unknowns does not fit the 160 KB LDS of a CU; sweep skipped", who, n);
		print_error(buf);
		return false;
	}
	return true;
}

template <int BS, bool NINE, bool SM, bool YT, bool PERM>
static void launch_x_kp(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int jb, int nlines,
                        hipStream_t st, const real_t *pf, Batch bt)
{
	size_t shm = line_lds_doubles(II - 2) * sizeof(real_t);
	auto k = relax_lines_x_kernel<BS, NINE, SM, YT, PERM>;
	if (shm > 64 * 1024) CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
	const char *ed = getenv("CEDAR_AMD_LINE_DBG"); // timing experiments only (1: no solve, 2: no right-hand side)
	hipLaunchKernelGGL(k, dim3(xcd_grid(nlines), bt.n), dim3(BS), shm, st, so, qf, q, sor, II, JJ, jb, nlines, ed ? atoi(ed) : 0, pf,
	                   bt.stride);
}

// pf != nullptr: the scan-ordered factor copy of this line direction (lines_permute), Dirichlet lines of the 256-lane kernel
template <int BS, bool NINE, bool SM, bool YT = false>
static void launch_x_k(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int jb, int nlines,
                       hipStream_t st, const real_t *pf, Batch bt)
{
	if (!SM && pf && BS >= 256) launch_x_kp<BS, NINE, false, YT, true>(so, qf, q, sor, II, JJ, jb, nlines, st, pf, bt);
	else launch_x_kp<BS, NINE, SM, YT, false>(so, qf, q, sor, II, JJ, jb, nlines, st, nullptr, bt);
}

template <int BS>
static void launch_x_n(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ,
                       int nstncl, int jb, hipStream_t st, bool sm, bool yt, const real_t *pf, Batch bt)
{
	int nlines = (JJ - 2 - jb + 1) / 2;
	if (nlines <= 0) return;
	if (yt) { // Dirichlet only
		if (nstncl == 5) launch_x_k<BS, true, false, true>(so, qf, q, sor, II, JJ, jb, nlines, st, pf, bt);
		else launch_x_k<BS, false, false, true>(so, qf, q, sor, II, JJ, jb, nlines, st, pf, bt);
	} else if (nstncl == 5) {
		if (sm) launch_x_k<BS, true, true>(so, qf, q, sor, II, JJ, jb, nlines, st, pf, bt);
		else launch_x_k<BS, true, false>(so, qf, q, sor, II, JJ, jb, nlines, st, pf, bt);
	} else {
		if (sm) launch_x_k<BS, false, true>(so, qf, q, sor, II, JJ, jb, nlines, st, pf, bt);
		else launch_x_k<BS, false, false>(so, qf, q, sor, II, JJ, jb, nlines, st, pf, bt);
	}
}

static void launch_x(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ,
                     int nstncl, int jb, hipStream_t st, bool sm = false, bool yt = false, const real_t *pf = nullptr,
                     Batch bt = Batch())
{
	switch (line_bs(II - 2)) {
	case 64: launch_x_n<64>(so, qf, q, sor, II, JJ, nstncl, jb, st, sm, yt, pf, bt); break;
	case 512: launch_x_n<512>(so, qf, q, sor, II, JJ, nstncl, jb, st, sm, yt, pf, bt); break;
	case 1024: launch_x_n<1024>(so, qf, q, sor, II, JJ, nstncl, jb, st, sm, yt, pf, bt); break;
	default: launch_x_n<256>(so, qf, q, sor, II, JJ, nstncl, jb, st, sm, yt, pf, bt);
	}
}

// ipn: 0 Dirichlet; 1 periodic in y only: ordinary solves, one y wrap after the sweep (relax_lines_x.f90:75-176);
// 2 / 3 periodic in x (/ and y): cyclic lines, y then x wrap after each colour (:178-300)
// scan-ordered factor copy for the lines of `sor` (n unknowns per line, line stride ld, nlines lines): 0 doubles when
// the line kernel would not use it (lines of at most 512 unknowns run on one wavefront)
size_t lines_permuted_doubles(int n, int nlines)
{
	switch (line_bs(n)) {
	case 256: return pf_line_doubles<256>(n) * (size_t)nlines;
	case 512: return pf_line_doubles<512>(n) * (size_t)nlines;
	case 1024: return pf_line_doubles<1024>(n) * (size_t)nlines;
	default: return 0;
	}
}

void lines_permute(const real_t *sor, real_t *pf, int n, int ld, int nlines, size_t PS, hipStream_t st)
{
	if (nlines <= 0 || n <= 0) return;
	switch (line_bs(n)) {
	case 256: hipLaunchKernelGGL(lines_permute_kernel<256>, dim3(nlines), dim3(256), 0, st, sor, pf, n, ld, PS); break;
	case 512: hipLaunchKernelGGL(lines_permute_kernel<512>, dim3(nlines), dim3(512), 0, st, sor, pf, n, ld, PS); break;
	case 1024: hipLaunchKernelGGL(lines_permute_kernel<1024>, dim3(nlines), dim3(1024), 0, st, sor, pf, n, ld, PS); break;
	default: break;
	}
}

void relax_lines_x(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int nstncl, int updown, hipStream_t st, int ipn, const real_t *pf, Batch bt)
{
	if (II < 3 || JJ < 3) return;
	if (!lds_ok(II - 2, "relax_lines_x")) return;
	const bool sm = ipn == 2 || ipn == 3;
	for (int c = 0; c < 2; c++) {
		// DOWN: lines J = 3,5,.. first (0-based rows 2,4,.. => jb = 1), then J = 2,4,..
		int jb = (updown == BMG_DOWN) ? 1 - c : c;
		launch_x(so, qf, q, sor, II, JJ, nstncl, jb, st, sm, false, sm ? nullptr : pf, bt);
		if (sm) wrap2(q, II, JJ, 1, ipn == 3, 1, st);
	}
	if (ipn == 1) wrap2(q, II, JJ, 1, 1, 0, st);
}

// ------------------------------------------------------------------ y-lines
// gather: bt[l*ldt + (j-1)] = rhs(i_l, j) for the colour's lines i_l = 1+ib+2l (0-based), j = 1..JJ-2
template <bool NINE>
__global__ __launch_bounds__(256) void ylines_rhs_T(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                     const real_t *__restrict__ q, real_t *__restrict__ bt,
                                                     int II, int JJ, int ib, int nlines, int ldt, int lofs)
{
	__shared__ real_t tile[32][33];
	const int l0 = blockIdx.x * 32, j0 = blockIdx.y * 32; // tile origin (line index within the chunk, j-1)
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
	const size_t sj = II, PS = (size_t)II * JJ;
	for (int r = ty; r < 32; r += 8) {
		const int l = l0 + tx, j = j0 + r + 1;
		real_t s = 0.0;
		if (l < nlines && j <= JJ - 2) {
			const size_t x = (size_t)(1 + ib + 2 * (lofs + l)) + sj * (size_t)j;
			// relax_lines_y.f90:103-107 / :133-134, reference term order
			s = qf[x];
			s = s + so[KW * PS + x] * q[x - 1];
			s = s + so[KW * PS + x + 1] * q[x + 1];
			if (NINE) {
				s = s + so[KSW * PS + x] * q[x - 1 - sj];
				s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
				s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
				s = s + so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
			}
		}
		tile[r][tx] = s;
	}
	__syncthreads();
	for (int r = ty; r < 32; r += 8) {
		const int l = l0 + r, jj = j0 + tx; // jj = j-1
		if (l < nlines && jj < JJ - 2) bt[(size_t)l * ldt + jj] = tile[tx][r];
	}
}

template <int BS, bool SM = false>
__global__ __launch_bounds__(BS) void ylines_solve(real_t *__restrict__ bt, const real_t *__restrict__ sor,
                                                    int II, int JJ, int ib, int nlines, int ldt,
                                                    const real_t *__restrict__ so, int lofs)
{
	extern __shared__ __attribute__((aligned(16))) real_t lds[];
	const int n = JJ - 2;
	const int npad = n + (n >> 3) + 1;
	real_t *y = lds, *wa = lds + npad, *wc = wa + 16, *cs = wc + 16;
	const unsigned L = blockIdx.x;
	if (L >= (unsigned)nlines) return;
	const size_t PS = (size_t)II * JJ;
	const int i = 1 + ib + 2 * (lofs + (int)L); // 0-based line position
	real_t *line = bt + (size_t)L * ldt;
	for (int t = threadIdx.x; t < n; t += BS) y[lpad(t)] = line[t];
	__syncthreads();
	// SOR(JJ,II,2): d = SOR(2.., i) , e = SOR(3.., i, 2)
	line_pttrs<BS>(y, n, sor + (size_t)JJ * i + 1, sor + PS + (size_t)JJ * i + 2, wa, wc, cs);
	for (int t = threadIdx.x; t < n; t += BS) line[t] = y[lpad(t)];
	if (SM) { // relax_lines_y.f90:196-207: u = -SO(I,2,KS) e_first - SO(I,JJ,KS) e_last
		real_t y1, yn;
		const real_t beta = line_sherman_morrison<BS>(y, n, sor + (size_t)JJ * i + 1, sor + PS + (size_t)JJ * i + 2,
		                                              so[KS * PS + (size_t)i + (size_t)II], so[KS * PS + (size_t)i + (size_t)II * (JJ - 1)],
		                                              wa, wc, cs, y1, yn);
		for (int t = threadIdx.x; t < n; t += BS) line[t] = line[t] - beta * y[lpad(t)];
	}
}

__global__ __launch_bounds__(256) void ylines_scatter_T(const real_t *__restrict__ bt, real_t *__restrict__ q,
                                                         int II, int JJ, int ib, int nlines, int ldt, int lofs)
{
	__shared__ real_t tile[32][33];
	const int l0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
	const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
	for (int r = ty; r < 32; r += 8) {
		const int l = l0 + r, jj = j0 + tx;
		tile[r][tx] = (l < nlines && jj < JJ - 2) ? bt[(size_t)l * ldt + jj] : 0.0;
	}
	__syncthreads();
	for (int r = ty; r < 32; r += 8) {
		const int l = l0 + tx, j = j0 + r + 1;
		if (l < nlines && j <= JJ - 2) q[(size_t)(1 + ib + 2 * (lofs + l)) + (size_t)II * j] = tile[tx][r];
	}
}

template <int BS>
static void launch_ysolve_n(real_t *bt, const real_t *sor, int II, int JJ, int ib, int nlines, int ldt, const real_t *so,
                            int lofs, size_t shm, bool sm, hipStream_t st)
{
	if (sm) {
		auto k = ylines_solve<BS, true>;
		if (shm > 64 * 1024) CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
		hipLaunchKernelGGL(k, dim3(nlines), dim3(BS), shm, st, bt, sor, II, JJ, ib, nlines, ldt, so, lofs);
	} else {
		auto k = ylines_solve<BS, false>;
		if (shm > 64 * 1024) CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
		hipLaunchKernelGGL(k, dim3(nlines), dim3(BS), shm, st, bt, sor, II, JJ, ib, nlines, ldt, so, lofs);
	}
}

static void launch_ysolve(real_t *bt, const real_t *sor, int II, int JJ, int ib, int nlines, int ldt, const real_t *so,
                          int lofs, size_t shm, bool sm, hipStream_t st)
{
	switch (line_bs(JJ - 2)) {
	case 64: launch_ysolve_n<64>(bt, sor, II, JJ, ib, nlines, ldt, so, lofs, shm, sm, st); break;
	case 512: launch_ysolve_n<512>(bt, sor, II, JJ, ib, nlines, ldt, so, lofs, shm, sm, st); break;
	case 1024: launch_ysolve_n<1024>(bt, sor, II, JJ, ib, nlines, ldt, so, lofs, shm, sm, st); break;
	default: launch_ysolve_n<256>(bt, sor, II, JJ, ib, nlines, ldt, so, lofs, shm, sm, st);
	}
}

size_t ylines_scratch_doubles(int II, int JJ)
{
	const int n = JJ - 2;
	return (size_t)((II - 2 + 1) / 2) * (size_t)((n + 15) & ~15) + 16;
}

// ipn: 0 Dirichlet; 2 periodic in x only: ordinary solves, one x wrap after the sweep (relax_lines_y.f90:77-176);
// 1 / 3 periodic in y (/ and x): cyclic lines, y then x wrap after each colour (:178-300)
void relax_lines_y(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, real_t *bt,
                   int II, int JJ, int nstncl, int updown, hipStream_t st, int ipn)
{
	const bool sm = ipn == 1 || ipn == 3;
	if (II < 3 || JJ < 3) return;
	if (!lds_ok(JJ - 2, "relax_lines_y")) return;
	const int n = JJ - 2;
	const int ldt = (n + 15) & ~15;
	const size_t shm = line_lds_doubles(n) * sizeof(real_t);
	// the lines of a colour go through gather -> solve -> scatter in chunks whose line-contiguous scratch
	// (chunk * ldt doubles) stays in the Infinity Cache between the three kernels
	const char *ce = getenv("CEDAR_AMD_YCHUNK");
	const int chunk_lines = ce ? atoi(ce) : 512; // 8192^2: 20.6 (whole colour) -> 19.8 ms per V-cycle (profiles/r01_experiment_yline_chunks.log)
	for (int c = 0; c < 2; c++) {
		int ib = (updown == BMG_DOWN) ? 1 - c : c; // DOWN: I = 3,5,.. first
		int nall = (II - 2 - ib + 1) / 2;
		if (nall <= 0) continue;
		const int step = chunk_lines > 0 ? chunk_lines : nall;
		for (int lofs = 0; lofs < nall; lofs += step) {
			const int nlines = nall - lofs < step ? nall - lofs : step;
			dim3 tg((nlines + 31) / 32, (n + 31) / 32);
			if (nstncl == 5)
				hipLaunchKernelGGL(ylines_rhs_T<true>, tg, dim3(256), 0, st, so, qf, q, bt, II, JJ, ib, nlines, ldt, lofs);
			else
				hipLaunchKernelGGL(ylines_rhs_T<false>, tg, dim3(256), 0, st, so, qf, q, bt, II, JJ, ib, nlines, ldt, lofs);
			launch_ysolve(bt, sor, II, JJ, ib, nlines, ldt, so, lofs, shm, sm, st);
			hipLaunchKernelGGL(ylines_scatter_T, tg, dim3(256), 0, st, bt, q, II, JJ, ib, nlines, ldt, lofs);
		}
		if (sm) wrap2(q, II, JJ, 1, 1, ipn == 3, st);
	}
	if (ipn == 2) wrap2(q, II, JJ, 1, 0, 1, st);
}

// ---- y-lines on transposed arrays (the solver's resident path, Dirichlet): the operator planes the right-hand
// side needs are kept transposed (setup_lines_yt), q is transposed around the sweep, and both colours run through
// the x-line kernel -- rows of the transposed arrays are the y lines, and SOR(JJ,II,2) already is the x layout of
// the transposed grid.  Same right-hand-side term order, same scan: bit-identical to relax_lines_y.  8192^2:
// 2 x 1534 us (gather + solve + scatter per colour) -> 2 x 750 us + two transposes of q.
// out(j,i) = in(i,j), whole arrays with ghosts: in is (II fast, JJ), out is (JJ fast, II)
__global__ __launch_bounds__(256) void transpose2_kernel(const real_t *__restrict__ in, real_t *__restrict__ out, int II, int JJ,
                                                         size_t bstride)
{
	__shared__ real_t tile[64][65];
	in += bstride * blockIdx.z; out += bstride * blockIdx.z; // batch item (common.h Batch)
	const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
	for (int r = ty; r < 64; r += 4) {
		const int i = i0 + tx, j = j0 + r;
		if (i < II && j < JJ) tile[r][tx] = in[(size_t)i + (size_t)II * j];
	}
	__syncthreads();
	for (int r = ty; r < 64; r += 4) {
		const int i = i0 + r, j = j0 + tx;
		if (i < II && j < JJ) out[(size_t)j + (size_t)JJ * i] = tile[tx][r];
	}
}

// the same with 16-byte accesses on both sides (even extents, 16-byte aligned arrays): a lane moves two neighbouring
// columns of a row in and two neighbouring rows of a column out
__global__ __launch_bounds__(256) void transpose2_pairs_kernel(const real_t *__restrict__ in, real_t *__restrict__ out, int II, int JJ,
                                                               size_t bstride)
{
	__shared__ real_t tile[64][65];
	in += bstride * blockIdx.z; out += bstride * blockIdx.z;
	const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
	const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5; // 32 pairs x 8
	for (int r = r0; r < 64; r += 8) {
		const int i = i0 + 2 * c, j = j0 + r;
		if (i < II && j < JJ) {
			const d2u v = *reinterpret_cast<const d2u *>(in + (size_t)i + (size_t)II * j);
			tile[r][2 * c] = v.x; tile[r][2 * c + 1] = v.y;
		}
	}
	__syncthreads();
	for (int r = r0; r < 64; r += 8) {
		const int i = i0 + r, j = j0 + 2 * c;
		if (i < II && j < JJ) {
			d2u v; v.x = tile[2 * c][r]; v.y = tile[2 * c + 1][r];
			*reinterpret_cast<d2u *>(out + (size_t)j + (size_t)JJ * i) = v;
		}
	}
}

void transpose2(const real_t *in, real_t *out, int II, int JJ, hipStream_t st, Batch bt)
{
	// CEDAR_AMD_TRANSPOSE_PAIRS=0: 8-byte accesses (round 2).  8192^2, 49 transposes of a line-xy V-cycle: 2.63 -> 2.28 ms; a
	// 64 x 128 tile (1 KB output rows, 66 KB of LDS) measured 2.65 ms and was dropped
	const char *e = getenv("CEDAR_AMD_TRANSPOSE_PAIRS");
	const bool pairs = !(e && atoi(e) == 0) && !(II & 1) && !(JJ & 1) && !(bt.stride & 1) && !(((uintptr_t)in | (uintptr_t)out) & 15);
	if (pairs) hipLaunchKernelGGL(transpose2_pairs_kernel, dim3((II + 63) / 64, (JJ + 63) / 64, bt.n), dim3(256), 0, st, in, out, II, JJ, bt.stride);
	else hipLaunchKernelGGL(transpose2_kernel, dim3((II + 63) / 64, (JJ + 63) / 64, bt.n), dim3(256), 0, st, in, out, II, JJ, bt.stride);
}

// sot (JJ fast, II, nstncl planes): plane KS = KW^T, KW = KS^T, KSW = KSW^T, KNW = KNW^T; KO is not read by the sweep
void setup_lines_yt(const real_t *so, real_t *sot, int II, int JJ, int nstncl, hipStream_t st)
{
	const size_t PS = (size_t)II * JJ;
	transpose2(so + KW * PS, sot + KS * PS, II, JJ, st);
	transpose2(so + KS * PS, sot + KW * PS, II, JJ, st);
	if (nstncl == 5) {
		transpose2(so + KSW * PS, sot + KSW * PS, II, JJ, st);
		transpose2(so + KNW * PS, sot + KNW * PS, II, JJ, st);
	}
}

// qft = transposed right-hand side (the caller keeps it while qf is unchanged), qt = scratch for the transposed q
void relax_lines_yt(const real_t *sot, const real_t *qft, real_t *q, real_t *qt, const real_t *sor,
                    int II, int JJ, int nstncl, int updown, hipStream_t st, const real_t *pf, Batch bt)
{
	if (II < 3 || JJ < 3) return;
	if (!lds_ok(JJ - 2, "relax_lines_y")) return;
	transpose2(q, qt, II, JJ, st, bt);
	for (int c = 0; c < 2; c++) {
		const int ib = (updown == BMG_DOWN) ? 1 - c : c; // DOWN: I = 3,5,.. first
		const int nlines = (II - 2 - ib + 1) / 2;
		if (nlines <= 0) continue;
		// transposed grid: JJ is the fast extent, II the number of rows
		launch_x(sot, qft, qt, sor, JJ, II, nstncl, ib, st, false, true, pf, bt);
	}
	transpose2(qt, q, JJ, II, st, bt);
}

// ---- pieces of the domain-decomposed line relaxation (cedar_amd/dist2d.py): right-hand sides of the
// lines of one zebra colour into a line-contiguous buffer, the carry correction, and the way back.
template <bool NINE>
__global__ __launch_bounds__(256) void xlines_rhs_kernel(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                          const real_t *__restrict__ q, real_t *__restrict__ out,
                                                          int II, int JJ, int lb, int ld)
{
	const int l = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x; // line of the colour, position
	if (t >= II - 2) return;
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t x = (size_t)(1 + lb + 2 * l) * sj + 1 + t;
	real_t s = qf[x]; // relax_lines_x.f90:106-111 / :128-129, reference term order
	s = s + so[KS * PS + x] * q[x - sj];
	s = s + so[KS * PS + x + sj] * q[x + sj];
	if (NINE) {
		s = s + so[KSW * PS + x] * q[x - 1 - sj];
		s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
		s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
		s = s + so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
	}
	out[(size_t)l * ld + t] = s;
}

__global__ __launch_bounds__(256) void xlines_store_kernel(const real_t *__restrict__ in, real_t *__restrict__ q,
                                                            int II, int lb, int ld)
{
	const int l = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= II - 2) return;
	q[(size_t)(1 + lb + 2 * l) * II + 1 + t] = in[(size_t)l * ld + t];
}

// y[l][i] = y[l][i] + p[l][i] * c[l]: the carry entering a line segment times the running product of its multipliers
__global__ __launch_bounds__(256) void lines_carry_kernel(real_t *__restrict__ y, const real_t *__restrict__ p,
                                                           const real_t *__restrict__ c, int n, int ld)
{
	const int l = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const size_t x = (size_t)l * ld + t;
	y[x] = y[x] + p[x] * c[l];
}

// dir 0: x lines (rows 1+lb, 3+lb, ..), dir 1: y lines (columns 1+lb, 3+lb, ..); out / in: (lines, positions), ld = positions
void lines_rhs2(const real_t *so, const real_t *qf, const real_t *q, real_t *out, int II, int JJ, int nstncl, int dir, int lb,
                hipStream_t st)
{
	const int nlines = dir == 0 ? (JJ - 2 - lb + 1) / 2 : (II - 2 - lb + 1) / 2;
	const int n = dir == 0 ? II - 2 : JJ - 2;
	if (nlines <= 0 || n <= 0) return;
	if (dir == 0) {
		dim3 grid((n + 255) / 256, nlines);
		if (nstncl == 5) hipLaunchKernelGGL(xlines_rhs_kernel<true>, grid, dim3(256), 0, st, so, qf, q, out, II, JJ, lb, n);
		else hipLaunchKernelGGL(xlines_rhs_kernel<false>, grid, dim3(256), 0, st, so, qf, q, out, II, JJ, lb, n);
	} else {
		dim3 tg((nlines + 31) / 32, (n + 31) / 32);
		if (nstncl == 5) hipLaunchKernelGGL(ylines_rhs_T<true>, tg, dim3(256), 0, st, so, qf, q, out, II, JJ, lb, nlines, n, 0);
		else hipLaunchKernelGGL(ylines_rhs_T<false>, tg, dim3(256), 0, st, so, qf, q, out, II, JJ, lb, nlines, n, 0);
	}
}

void lines_store2(const real_t *in, real_t *q, int II, int JJ, int dir, int lb, hipStream_t st)
{
	const int nlines = dir == 0 ? (JJ - 2 - lb + 1) / 2 : (II - 2 - lb + 1) / 2;
	const int n = dir == 0 ? II - 2 : JJ - 2;
	if (nlines <= 0 || n <= 0) return;
	if (dir == 0) hipLaunchKernelGGL(xlines_store_kernel, dim3((n + 255) / 256, nlines), dim3(256), 0, st, in, q, II, lb, n);
	else hipLaunchKernelGGL(ylines_scatter_T, dim3((nlines + 31) / 32, (n + 31) / 32), dim3(256), 0, st, in, q, II, JJ, lb, nlines, n, 0);
}

void lines_carry(real_t *y, const real_t *p, const real_t *c, int nlines, int n, int ld, hipStream_t st)
{
	if (nlines <= 0 || n <= 0) return;
	hipLaunchKernelGGL(lines_carry_kernel, dim3((n + 255) / 256, nlines), dim3(256), 0, st, y, p, c, n, ld);
}

} // namespace cedar_amd
