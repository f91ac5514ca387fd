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
#include "relax3_psum.h"

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

struct C9 {
	real_t w, s, sw;        // stored at X:        kw, ks, ksw
	real_t s_n, nw_n;       // stored at X+(0,1):  ks, knw
	real_t w_e, nw_e;       // stored at X+(1,0):  kw, knw
	real_t sw_ne;           // stored at X+(1,1):  ksw
};

// qq[dj+1][di+1]; term order of BMG2_SymStd_relax_GS.f90:98-107 (= residual.f90:90-98)
__device__ __forceinline__ real_t offdiag9(real_t qf, const C9 &c, const real_t (&qq)[3][3])
{
	real_t s = qf;
	s = s + c.w * qq[1][0];
	s = s + c.w_e * qq[1][2];
	s = s + c.s * qq[0][1];
	s = s + c.s_n * qq[2][1];
	s = s + c.sw * qq[0][0];
	s = s + c.nw_e * qq[0][2];
	s = s + c.nw_n * qq[2][0];
	s = s + c.sw_ne * qq[2][2];
	return s;
}

__device__ __forceinline__ void ldpair2(const real_t *__restrict__ p, bool two, real_t &a, real_t &b)
{
	if (two) {
		d2u v = *reinterpret_cast<const d2u *>(p);
		a = v.x; b = v.y;
	} else {
		a = p[0]; b = 0.0;
	}
}

// all operands of the pair (ie, io) of row `row`: 16-byte loads; `two`: element io+1 inside the row
__device__ __forceinline__ void load_pair9(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                           const real_t *__restrict__ q, size_t row, size_t sj, size_t PS,
                                           int ie, int io, bool two, C9 &ce, C9 &co,
                                           real_t (&qe)[3][3], real_t (&qo)[3][3], real_t &qfe, real_t &qfo)
{
	real_t a, b;
	ldpair2(so + KW * PS + row + ie, true, a, b); ce.w = a; co.w = b; ce.w_e = b;
	ldpair2(so + KS * PS + row + ie, true, a, b); ce.s = a; co.s = b;
	ldpair2(so + KSW * PS + row + ie, true, a, b); ce.sw = a; co.sw = b;
	ldpair2(so + KS * PS + row + sj + ie, true, a, b); ce.s_n = a; co.s_n = b;
	ldpair2(so + KNW * PS + row + sj + ie, true, a, b); ce.nw_n = a; co.nw_n = b;
	ldpair2(so + KNW * PS + row + io, two, a, b); ce.nw_e = a; co.nw_e = b;
	ldpair2(so + KSW * PS + row + sj + io, two, a, b); ce.sw_ne = a; co.sw_ne = b;
	co.w_e = two ? so[KW * PS + row + io + 1] : 0.0;
	ldpair2(qf + row + ie, true, qfe, qfo);
#pragma unroll
	for (int dj = 0; dj < 3; dj++) {
		const real_t *r = q + row + (ptrdiff_t)(dj - 1) * (ptrdiff_t)sj;
		real_t w0, w1, w2, w3;
		ldpair2(r + ie - 1, true, w0, w1);
		ldpair2(r + io, two, w2, w3);
		qe[dj][0] = w0; qe[dj][1] = w1; qe[dj][2] = w2;
		qo[dj][0] = w1; qo[dj][1] = w2; qo[dj][2] = w3;
	}
}

// EFIRST: even 1-based i first (DOWN in 2D), else odd i first (UP).
// One workgroup per grid row; lane p of chunk c owns the pair (i_e,i_o) = (2P+2, 2P+3), P = c*BS+p.
// Within a chunk the fresh first-colour values reach the neighbouring lane through LDS; between
// chunks one value is carried: the chunks of a row are walked downwards for EFIRST (the last odd
// point of a chunk needs the first even point of the next chunk, already relaxed) and upwards
// otherwise.  In-place is safe: a launch writes rows of one parity only, and inside a row every
// value that a later chunk still has to read "old" has not been written yet.
template <int BS, bool EFIRST>
__device__ __forceinline__ void relax9_row_task(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                int II, size_t sj, size_t PS, size_t row, real_t *xch, real_t *carry_s)
{
	const int npairs = (II - 2 + 1) / 2;
	const int nchunks = (npairs + BS - 1) / BS;
	const int t = threadIdx.x;
	for (int cc = 0; cc < nchunks; cc++) {
		const int c = EFIRST ? nchunks - 1 - cc : cc;
		const int p = c * BS + t;
		const int ie = 2 * p + 1, io = ie + 1;
		const bool e_ok = ie <= II - 2, o_ok = io <= II - 2, two = io + 1 <= II - 1;
		C9 ce, co;
		real_t qe[3][3], qo[3][3], qfe = 0, qfo = 0, sre = 0, sro = 0, e_new = 0, o_new = 0;
		if (e_ok) {
			load_pair9(so, qf, q, row, sj, PS, ie, io, two, ce, co, qe, qo, qfe, qfo);
			ldpair2(sor + PS + row + ie, true, sre, sro);
		}
		const real_t carry = *carry_s; // written by the previous chunk iteration (unused in the first)
		if (EFIRST) {
			if (e_ok) { e_new = offdiag9(qfe, ce, qe) * sre; xch[t] = e_new; }
			__syncthreads();
			if (o_ok) {
				qo[1][0] = e_new;
				if (io + 1 <= II - 2) qo[1][2] = (t < BS - 1) ? xch[t + 1] : carry; // next pair's fresh even point
				o_new = offdiag9(qfo, co, qo) * sro;
			}
			if (t == 0) *carry_s = e_new; // first even point of this chunk, for the chunk below
		} else {
			if (o_ok) { o_new = offdiag9(qfo, co, qo) * sro; xch[t + 1] = o_new; }
			__syncthreads();
			if (e_ok) {
				if (p > 0) qe[1][0] = (t > 0) ? xch[t] : carry; // previous pair's fresh odd point
				if (o_ok) qe[1][2] = o_new;
				e_new = offdiag9(qfe, ce, qe) * sre;
			}
			if (t == BS - 1) *carry_s = o_new;
		}
		if (e_ok) {
			if (o_ok) {
				d2u v; v.x = e_new; v.y = o_new;
				*reinterpret_cast<d2u *>(q + row + ie) = v;
			} else
				q[row + ie] = e_new;
		}
		__syncthreads(); // stores + carry visible before the next chunk loads / reads them
	}
}

// rows j = j0 + jstep * L (0-based incl. ghost), L < nrows; the row classes use j0 = 1 + jb, jstep = 2
template <int BS, bool EFIRST>
__global__ __launch_bounds__(BS) void relax9_rows(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                   real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                   int II, int JJ, int j0, int jstep, int nrows, size_t bstride)
{
	__shared__ real_t xch[BS + 2];
	__shared__ real_t carry_s;
	const unsigned L = xcd_remap(blockIdx.x, (unsigned)nrows);
	if (L >= (unsigned)nrows) return;
	qf += bstride * blockIdx.y; q += bstride * blockIdx.y; // batch item (common.h Batch)
	const size_t sj = II, PS = (size_t)II * JJ;
	relax9_row_task<BS, EFIRST>(so, qf, q, sor, II, sj, PS, (size_t)(j0 + jstep * (int)L) * sj, xch, &carry_s);
}

// Band-fused sweep: both row classes in ONE launch.  The sweep relaxes the rows of class F (parity jbF) before those
// of class S; an S row reads the fresh values of its two F neighbours and nothing else changes under it.  A workgroup
// owns a run of consecutive F rows [f0, f1) and walks F(f0), F(f0+1), S between them, F(f0+2), S, ...: the three
// operator rows both classes need (ks, ksw, knw of the upper row: 3 of the 8 slot-rows a row task reads) and the q rows
// are used again one task later instead of being streamed once per launch.  The S row between two runs has its F
// neighbours in different workgroups and is left to a small second launch (relax9_rows, jstep = 2*frun).  Same
// arithmetic per point on the same values => identical to the two-launch order.  (2D analogue of relax27_plane.)
template <int BS, bool EFIRST>
__global__ __launch_bounds__(BS) void relax9_band(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                   real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                   int II, int JJ, int jbF, int frun, int nrun, size_t bstride)
{
	__shared__ real_t xch[BS + 2];
	__shared__ real_t carry_s;
	const unsigned run = xcd_remap(blockIdx.x, (unsigned)nrun);
	if (run >= (unsigned)nrun) return;
	qf += bstride * blockIdx.y; q += bstride * blockIdx.y; // batch item (common.h Batch)
	const size_t sj = II, PS = (size_t)II * JJ;
	const int nF = (JJ - 2 - jbF + 1) / 2, nS = (JJ - 2 - (1 - jbF) + 1) / 2;
	const int f0 = (int)run * frun, f1 = min(nF, f0 + frun);
	for (int f = f0; f < f1; f++) {
		relax9_row_task<BS, EFIRST>(so, qf, q, sor, II, sj, PS, (size_t)(1 + jbF + 2 * f) * sj, xch, &carry_s);
		// the S row both of whose F neighbours are now done (a missing neighbour = ghost row):
		//   jbF = 0: S row g (j = 2+2g) lies between F rows g, g+1  -> after F(f): g = f-1 (f > f0)
		//   jbF = 1: S row g (j = 1+2g) lies between F rows g-1, g  -> after F(f): g = f   (f > f0, or f = 0)
		const int g = jbF ? f : f - 1;
		const bool have = jbF ? (f > f0 || f == 0) : (f > f0);
		if (have && g >= 0 && g < nS) // (the row task ends with a barrier: the F stores are visible)
			relax9_row_task<BS, EFIRST>(so, qf, q, sor, II, sj, PS, (size_t)(2 - jbF + 2 * g) * sj, xch, &carry_s);
	}
	if (f1 == nF && f1 > f0) { // the S row beyond the last F row (its other neighbour is the ghost row)
		const int g = jbF ? nF : nF - 1;
		if (g < nS) relax9_row_task<BS, EFIRST>(so, qf, q, sor, II, sj, PS, (size_t)(2 - jbF + 2 * g) * sj, xch, &carry_s);
	}
}

// ------------------------------------------------------------------ band-fused sweep with inter-row partial sums
// The 2D analogue of relax3d_psum.hip, with the partial sums kept in LDS.  north_star's contract (histories within 1e-10),
// not bit for bit: BMG2_SymStd_relax_GS stays on the reference order (relax2_gs).
// An S row reads, besides its own row, the three operator rows (ks, ksw, knw) that couple it to the F row below and the
// three that couple it to the F row above -- rows the two F tasks of the same workgroup have just streamed for their own
// updates -- plus the two fresh F rows.  Here each F task multiplies its fresh values with those coefficients while it
// holds them in registers and adds the products into an LDS row T of the S row above / below (position-wise: own column,
// then the column to the right, then to the left, a barrier between the three so that every T entry has one writer at a
// time and a fixed order of additions); the S task between two F rows of the run then computes
//     q = (qf + w q(i-1) + w_e q(i+1) + T) / diag
// from ONE operator row (kw), 1/diag, qf and its own row: 5 row streams instead of 14.  S rows whose two F neighbours do
// not both belong to the run (rows between runs: second small launch; row 1 / the row beyond the last F row: in place) keep
// the reference order.  LDS: two T rows of II doubles (the S row being filled and the one being consumed).
template <int BS, bool EFIRST>
__device__ __forceinline__ void relax9_row_task_F(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                  real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                  int II, size_t sj, size_t PS, size_t row, real_t *xch, real_t *carry_s,
                                                  real_t *Tup, real_t *Tdn)
{
	const int npairs = (II - 2 + 1) / 2;
	const int nchunks = (npairs + BS - 1) / BS;
	const int t = threadIdx.x;
	if (Tup) { // this F row is the first contributor of the S row above: start its sums from zero
		for (int i = t; i < II; i += BS) Tup[i] = 0.0;
		__syncthreads();
	}
	for (int cc = 0; cc < nchunks; cc++) {
		const int c = EFIRST ? nchunks - 1 - cc : cc;
		const int p = c * BS + t;
		const int ie = 2 * p + 1, io = ie + 1;
		const bool e_ok = ie <= II - 2, o_ok = io <= II - 2, two = io + 1 <= II - 1;
		C9 ce, co;
		real_t qe[3][3], qo[3][3], qfe = 0, qfo = 0, sre = 0, sro = 0, e_new = 0, o_new = 0;
		if (e_ok) {
			load_pair9(so, qf, q, row, sj, PS, ie, io, two, ce, co, qe, qo, qfe, qfo);
			ldpair2(sor + PS + row + ie, true, sre, sro);
		}
		const real_t ghostL = qe[1][0], ghostR = o_ok ? qo[1][2] : qe[1][2]; // old values of the ghost columns (p = 0 / last lane)
		const real_t carry = *carry_s;
		if (EFIRST) {
			if (e_ok) { e_new = offdiag9(qfe, ce, qe) * sre; xch[t] = e_new; }
			__syncthreads();
			if (o_ok) {
				qo[1][0] = e_new;
				if (io + 1 <= II - 2) qo[1][2] = (t < BS - 1) ? xch[t + 1] : carry;
				o_new = offdiag9(qfo, co, qo) * sro;
			}
			if (t == 0) *carry_s = e_new;
		} else {
			if (o_ok) { o_new = offdiag9(qfo, co, qo) * sro; xch[t + 1] = o_new; }
			__syncthreads();
			if (e_ok) {
				if (p > 0) qe[1][0] = (t > 0) ? xch[t] : carry;
				if (o_ok) qe[1][2] = o_new;
				e_new = offdiag9(qfe, ce, qe) * sre;
			}
			if (t == BS - 1) *carry_s = o_new;
		}
		if (e_ok) {
			if (o_ok) {
				d2u v; v.x = e_new; v.y = o_new;
				*reinterpret_cast<d2u *>(q + row + ie) = v;
			} else
				q[row + ie] = e_new;
		}
		// ---- contributions of this chunk's sources to the S rows above (Tup) and below (Tdn).  Sources: e, o (fresh); the
		// right ghost column as `o` when the row ends on e (old value, only its contribution to the left counts); the left
		// ghost column with lane p = 0, the right one with the lane of the last interior point when that is an o.
		const real_t so_v = o_ok ? o_new : ghostR;
		const bool o_src = o_ok || (e_ok && io == II - 1);
		if (e_ok) { // own column
			if (Tup) { Tup[ie] += e_new * ce.s_n; if (o_ok) Tup[io] += o_new * co.s_n; }
			if (Tdn) { Tdn[ie] += e_new * ce.s; if (o_ok) Tdn[io] += o_new * co.s; }
		}
		__syncthreads();
		if (e_ok) { // the column to the right of each source (targets io and io+1; lane 0 also: ghost column 0 -> column 1)
			if (Tup) {
				if (o_ok) Tup[io] += e_new * ce.sw_ne;
				if (o_ok && io + 1 <= II - 2) Tup[io + 1] += o_new * co.sw_ne;
				if (p == 0) Tup[1] += ghostL * so[KSW * PS + row + sj + 1];
			}
			if (Tdn) {
				if (o_ok) Tdn[io] += e_new * ce.nw_e;
				if (o_ok && io + 1 <= II - 2) Tdn[io + 1] += o_new * co.nw_e;
				if (p == 0) Tdn[1] += ghostL * so[KNW * PS + row + 1];
			}
		}
		__syncthreads();
		if (e_ok) { // the column to the left of each source (targets ie-1 and ie; the last lane also: ghost column II-1 -> II-2)
			if (Tup) {
				if (ie - 1 >= 1) Tup[ie - 1] += e_new * ce.nw_n;
				if (o_src) Tup[ie] += so_v * co.nw_n;
				if (o_ok && io == II - 2) Tup[io] += ghostR * so[KNW * PS + row + sj + (size_t)(II - 1)];
			}
			if (Tdn) {
				if (ie - 1 >= 1) Tdn[ie - 1] += e_new * ce.sw;
				if (o_src) Tdn[ie] += so_v * co.sw;
				if (o_ok && io == II - 2) Tdn[io] += ghostR * so[KSW * PS + row + (size_t)(II - 1)];
			}
		}
		__syncthreads(); // stores, carry and sums visible before the next chunk
	}
}

// S row from its partial sums: term order qf, w, w_e (relax_GS.f90:98-99), then T
template <int BS, bool EFIRST>
__device__ __forceinline__ void relax9_row_task_S(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                  real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                  int II, size_t PS, size_t row, real_t *xch, real_t *carry_s, const real_t *T)
{
	const int npairs = (II - 2 + 1) / 2;
	const int nchunks = (npairs + BS - 1) / BS;
	const int t = threadIdx.x;
	for (int cc = 0; cc < nchunks; cc++) {
		const int c = EFIRST ? nchunks - 1 - cc : cc;
		const int p = c * BS + t;
		const int ie = 2 * p + 1, io = ie + 1;
		const bool e_ok = ie <= II - 2, o_ok = io <= II - 2, two = io + 1 <= II - 1;
		real_t we = 0, wo = 0, wee = 0, weo = 0, qfe = 0, qfo = 0, sre = 0, sro = 0, e_new = 0, o_new = 0;
		real_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, te = 0, to = 0;
		if (e_ok) {
			ldpair2(so + KW * PS + row + ie, true, we, wo);
			wee = wo;
			weo = two ? so[KW * PS + row + io + 1] : 0.0;
			ldpair2(qf + row + ie, true, qfe, qfo);
			ldpair2(sor + PS + row + ie, true, sre, sro);
			ldpair2(q + row + ie - 1, true, w0, w1);
			ldpair2(q + row + io, two, w2, w3);
			te = T[ie];
			to = o_ok ? T[io] : 0.0;
		}
		const real_t carry = *carry_s;
		if (EFIRST) {
			if (e_ok) { e_new = (((qfe + we * w0) + wee * w2) + te) * sre; xch[t] = e_new; }
			__syncthreads();
			if (o_ok) {
				real_t east = w3;
				if (io + 1 <= II - 2) east = (t < BS - 1) ? xch[t + 1] : carry;
				o_new = (((qfo + wo * e_new) + weo * east) + to) * sro;
			}
			if (t == 0) *carry_s = e_new;
		} else {
			if (o_ok) { o_new = (((qfo + wo * w1) + weo * w3) + to) * sro; xch[t + 1] = o_new; }
			__syncthreads();
			if (e_ok) {
				real_t west = w0;
				if (p > 0) west = (t > 0) ? xch[t] : carry;
				e_new = (((qfe + we * west) + wee * (o_ok ? o_new : w2)) + te) * sre;
			}
			if (t == BS - 1) *carry_s = o_new;
		}
		if (e_ok) {
			if (o_ok) {
				d2u v; v.x = e_new; v.y = o_new;
				*reinterpret_cast<d2u *>(q + row + ie) = v;
			} else
				q[row + ie] = e_new;
		}
		__syncthreads();
	}
}

template <int BS, bool EFIRST>
__global__ __launch_bounds__(BS) void relax9_band_psum(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                        real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                        int II, int JJ, int jbF, int frun, int nrun, size_t bstride)
{
	extern __shared__ real_t lds_band[];
	const int IIp = (II + 1) & ~1;
	real_t *T0 = lds_band, *T1 = lds_band + IIp, *xch = lds_band + 2 * IIp, *carry_s = xch + BS + 2;
	const unsigned run = xcd_remap(blockIdx.x, (unsigned)nrun);
	if (run >= (unsigned)nrun) return;
	qf += bstride * blockIdx.y; q += bstride * blockIdx.y; // batch item (common.h Batch)
	const size_t sj = II, PS = (size_t)II * JJ;
	const int nF = (JJ - 2 - jbF + 1) / 2, nS = (JJ - 2 - (1 - jbF) + 1) / 2;
	const int f0 = (int)run * frun, f1 = min(nF, f0 + frun);
	for (int f = f0; f < f1; f++) {
		// F(f) starts the sums of the S row between F(f) and F(f+1) (slot f & 1) and completes those of the S row between
		// F(f-1) and F(f) (slot (f-1) & 1), when that neighbour belongs to the run
		real_t *Tup = f + 1 <= f1 - 1 ? ((f & 1) ? T1 : T0) : nullptr;
		real_t *Tdn = f - 1 >= f0 ? (((f - 1) & 1) ? T1 : T0) : nullptr;
		relax9_row_task_F<BS, EFIRST>(so, qf, q, sor, II, sj, PS, (size_t)(1 + jbF + 2 * f) * sj, xch, carry_s, Tup, Tdn);
		const int g = jbF ? f : f - 1;
		const bool have = jbF ? (f > f0 || f == 0) : (f > f0);
		if (have && g >= 0 && g < nS) {
			const size_t rowS = (size_t)(2 - jbF + 2 * g) * sj;
			if (f > f0) relax9_row_task_S<BS, EFIRST>(so, qf, q, sor, II, PS, rowS, xch, carry_s, ((f - 1) & 1) ? T1 : T0);
			else relax9_row_task<BS, EFIRST>(so, qf, q, sor, II, sj, PS, rowS, xch, carry_s); // row 1 below the first F row (jbF = 1)
		}
	}
	if (f1 == nF && f1 > f0) { // the S row beyond the last F row (its other neighbour is the ghost row): reference order
		const int g = jbF ? nF : nF - 1;
		if (g < nS) relax9_row_task<BS, EFIRST>(so, qf, q, sor, II, sj, PS, (size_t)(2 - jbF + 2 * g) * sj, xch, carry_s);
	}
}

// 9-point residual, pair per lane, 16-byte loads (BMG2_SymStd_residual.f90:88-99)
__global__ __launch_bounds__(256) void residual9_rows(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                       const real_t *__restrict__ q, real_t *__restrict__ res,
                                                       int II, int JJ, unsigned nrows, size_t bstride)
{
	const unsigned L = xcd_remap(blockIdx.x, nrows);
	if (L >= nrows) return;
	qf += bstride * blockIdx.y; q += bstride * blockIdx.y; res += bstride * blockIdx.y; // batch item (common.h Batch)
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t row = (size_t)(L + 1) * sj;
	for (int p = threadIdx.x; 2 * p + 1 <= II - 2; p += blockDim.x) {
		const int ie = 2 * p + 1, io = ie + 1;
		const bool o_ok = io <= II - 2, two = io + 1 <= II - 1;
		C9 ce, co;
		real_t qe[3][3], qo[3][3], qfe, qfo, de, dn;
		load_pair9(so, qf, q, row, sj, PS, ie, io, two, ce, co, qe, qo, qfe, qfo);
		ldpair2(so + row + ie, true, de, dn); // KO plane
		const real_t re = offdiag9(qfe, ce, qe) - de * qe[1][1];
		if (o_ok) {
			const real_t ro = offdiag9(qfo, co, qo) - dn * qo[1][1];
			d2u v; v.x = re; v.y = ro;
			*reinterpret_cast<d2u *>(res + row + ie) = v;
		} else
			res[row + ie] = re;
	}
}

void residual9_fast(const real_t *so, const real_t *qf, const real_t *q, real_t *res, int II, int JJ, hipStream_t st, Batch bt)
{
	unsigned nrows = (unsigned)(JJ - 2);
	hipLaunchKernelGGL(residual9_rows, dim3(xcd_grid(nrows), bt.n), dim3((II - 2) / 2 >= 256 ? 256 : 64), 0, st, so, qf, q, res, II, JJ,
	                   nrows, bt.stride);
}

// 5-point red-black, one colour per launch (relax_GS.f90:120-135): colour = mod(j+jo,2)
__global__ __launch_bounds__(256) void relax5_colour(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                                                      real_t *__restrict__ q, const real_t *__restrict__ sor,
                                                      int II, int JJ, int jo, size_t bstride)
{
	qf += bstride * blockIdx.z; q += bstride * blockIdx.z; // batch item (common.h Batch)
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

// one row class (jb = parity of the 0-based row minus 1) of the nine-point sweep, both i-colours;
// efirst: even 1-based i first (the DOWN order of the 2D sweep).  Domain-decomposed runs exchange
// halos between row classes.
template <int BS>
static void launch_rows9(bool efirst, const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ,
                         int j0, int jstep, int nrows, hipStream_t st, Batch bt = Batch())
{
	if (nrows <= 0) return;
	const dim3 grid(xcd_grid((unsigned)nrows), bt.n);
	if (efirst) hipLaunchKernelGGL((relax9_rows<BS, true>), grid, dim3(BS), 0, st, so, qf, q, sor, II, JJ, j0, jstep, nrows, bt.stride);
	else hipLaunchKernelGGL((relax9_rows<BS, false>), grid, dim3(BS), 0, st, so, qf, q, sor, II, JJ, j0, jstep, nrows, bt.stride);
}

static void pass9(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int jb, bool efirst,
                  hipStream_t st, Batch bt)
{
	const int nrows = (JJ - 2 - jb + 1) / 2;
	if ((II - 2 + 1) / 2 <= 64) launch_rows9<64>(efirst, so, qf, q, sor, II, JJ, 1 + jb, 2, nrows, st, bt);
	else launch_rows9<256>(efirst, so, qf, q, sor, II, JJ, 1 + jb, 2, nrows, st, bt);
}

void relax2_pass9(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                  int II, int JJ, int jb, int efirst, hipStream_t st)
{
	pass9(so, qf, q, sor, II, JJ, jb, efirst != 0, st, Batch());
}

// F rows per workgroup of the band-fused sweep; 0 = two launches per sweep (one per row class).
// CEDAR_AMD_FRUN2 overrides (0 = never; n = runs of n rows wherever the grid has >= 4 runs per class).
static int band_frun(int II, int JJ)
{
	const char *e = getenv("CEDAR_AMD_FRUN2"); // read per call: the tests switch it between cases
	const int ny = JJ - 2;
	if (e) {
		const int frun = atoi(e);
		return (frun <= 0 || ny < 8 * frun) ? 0 : frun;
	}
	// measured (profiles/r02_experiment_band_fused_relax9.log): 4096^2 -4.5 % at runs of 4, slower from 8 on (a 2D grid has
	// only ny/2 row tasks per class: long runs leave compute units idle), slower at 2048^2 and below at any run length
	return (ny >= 4096 && II - 2 >= 2048) ? 4 : 0;
}

// whole nine-point sweep
static void relax2_sweep9(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, bool down, hipStream_t st,
                          Batch bt)
{
	const int frun = band_frun(II, JJ);
	if (frun == 0 || (II - 2 + 1) / 2 <= 64) {
		for (int c = 0; c < 2; c++) // DOWN: rows J=2,4,.. first (LSTART=2), even 1-based i first
			pass9(so, qf, q, sor, II, JJ, down ? c : 1 - c, down, st, bt);
		return;
	}
	const int jbF = down ? 0 : 1;
	const int nF = (JJ - 2 - jbF + 1) / 2, nrun = (nF + frun - 1) / frun;
	const dim3 grid(xcd_grid((unsigned)nrun), bt.n);
	if (down) hipLaunchKernelGGL((relax9_band<256, true>), grid, dim3(256), 0, st, so, qf, q, sor, II, JJ, jbF, frun, nrun, bt.stride);
	else hipLaunchKernelGGL((relax9_band<256, false>), grid, dim3(256), 0, st, so, qf, q, sor, II, JJ, jbF, frun, nrun, bt.stride);
	// S rows between runs: jbF = 0: j = 2 frun (r+1); jbF = 1: j = 1 + 2 frun (r+1), r = 0 .. nrun-2
	launch_rows9<256>(down, so, qf, q, sor, II, JJ, (jbF ? 1 : 0) + 2 * frun, 2 * frun, nrun - 1, st, bt);
}

// The sweep with inter-row partial sums (relax9_band_psum): rows of at most 4098 doubles (two T rows of LDS beside a second
// workgroup), at least 8 runs.  CEDAR_AMD_FRUN2 gives the run length as for the band-fused sweep.
static int band_psum_frun(int II, int JJ)
{
	const char *e = getenv("CEDAR_AMD_FRUN2");
	const int ny = JJ - 2;
	if ((II - 2 + 1) / 2 <= 64 || (size_t)II * 2 * sizeof(real_t) > 68 * 1024) return 0;
	if (e) {
		const int frun = atoi(e);
		return (frun < 2 || ny < 8 * frun) ? 0 : frun;
	}
	// measured (profiles/r03_psum2d_run_length.log): 4096^2 0.310 -> 0.299 / 0.260 / 0.320 ms per sweep for runs of 2 / 4 / 8 rows
	// (a 2D grid has only ny/2 F rows: runs of 8 leave half the compute units idle), 2048^2 and 1024^2 slower than the two
	// row-class launches at every run length => the finest level of config 2 only
	return (ny >= 4096 && II - 2 >= 2048) ? 4 : 0;
}

bool relax2_psum_wanted(int II, int JJ)
{
	const char *e = getenv("CEDAR_AMD_PSUM");
	if (e && atoi(e) == 0) return false;
	return band_psum_frun(II, JJ) > 0;
}

void relax2_gs9_psum(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int updown, hipStream_t st,
                     Batch bt)
{
	const bool down = updown == BMG_DOWN;
	const int frun = band_psum_frun(II, JJ);
	if (frun == 0) {
		relax2_sweep9(so, qf, q, sor, II, JJ, down, st, bt);
		return;
	}
	const int jbF = down ? 0 : 1;
	const int nF = (JJ - 2 - jbF + 1) / 2, nrun = (nF + frun - 1) / frun;
	const size_t shm = ((size_t)((II + 1) & ~1) * 2 + 256 + 2 + 2) * sizeof(real_t);
	static bool attr_dev[64] = {false};
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	if (!attr_dev[dev_ & 63]) {
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)relax9_band_psum<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)relax9_band_psum<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
		attr_dev[dev_ & 63] = true;
	}
	const dim3 grid(xcd_grid((unsigned)nrun), bt.n);
	if (down) hipLaunchKernelGGL((relax9_band_psum<256, true>), grid, dim3(256), shm, st, so, qf, q, sor, II, JJ, jbF, frun, nrun, bt.stride);
	else hipLaunchKernelGGL((relax9_band_psum<256, false>), grid, dim3(256), shm, st, so, qf, q, sor, II, JJ, jbF, frun, nrun, bt.stride);
	launch_rows9<256>(down, so, qf, q, sor, II, JJ, (jbF ? 1 : 0) + 2 * frun, 2 * frun, nrun - 1, st, bt);
}

// recompute the points of column icol (0-based incl. ghost) on the rows of class jb: used after the
// x-neighbour's fresh first colour arrived (the update of a point does not read its own old value)
__global__ void relax9_column(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                              real_t *__restrict__ q, const real_t *__restrict__ sor,
                              int II, int JJ, int icol, int jb)
{
	const int nrows = (JJ - 2 - jb + 1) / 2;
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nrows) return;
	const size_t sj = II, PS = (size_t)II * JJ;
	const size_t x = (size_t)icol + sj * (size_t)(1 + jb + 2 * t);
	real_t s = qf[x];
	s = s + so[KW * PS + x] * q[x - 1];
	s = s + so[KW * PS + x + 1] * q[x + 1];
	s = s + so[KS * PS + x] * q[x - sj];
	s = s + so[KS * PS + x + sj] * q[x + sj];
	s = s + so[KSW * PS + x] * q[x - 1 - sj];
	s = s + so[KNW * PS + x + 1] * q[x + 1 - sj];
	s = s + so[KNW * PS + x + sj] * q[x - 1 + sj];
	s = s + so[KSW * PS + x + 1 + sj] * q[x + 1 + sj];
	q[x] = s * sor[PS + x];
}

void relax2_fixup9(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int icol, int jb, hipStream_t st)
{
	const int nrows = (JJ - 2 - jb + 1) / 2;
	if (nrows <= 0) return;
	hipLaunchKernelGGL(relax9_column, dim3((nrows + 127) / 128), dim3(128), 0, st, so, qf, q, sor, II, JJ, icol, jb);
}

// one colour of the five-point red-black sweep: jo in {2,3}, points with mod(i+j+jo,2) == 0 (1-based)
static void colour5(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int jo, hipStream_t st, Batch bt)
{
	if (II < 3 || JJ < 3) return;
	dim3 grid(((II - 2 + 1) / 2 + 255) / 256, JJ - 2, bt.n);
	hipLaunchKernelGGL(relax5_colour, grid, dim3(256), 0, st, so, qf, q, sor, II, JJ, jo, bt.stride);
}

void relax2_colour5(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                    int II, int JJ, int jo, hipStream_t st)
{
	colour5(so, qf, q, sor, II, JJ, jo, st, Batch());
}

void relax2_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
               int II, int JJ, int nstncl, int updown, hipStream_t st, Batch bt)
{
	if (II < 3 || JJ < 3) return;
	const bool down = (updown == BMG_DOWN);
	if (nstncl == 5) {
		relax2_sweep9(so, qf, q, sor, II, JJ, down, st, bt);
	} else {
		for (int c = 0; c < 2; c++)
			colour5(so, qf, q, sor, II, JJ, down ? 2 + c : 3 - c /* LSTART..LEND */, st, bt);
	}
}

} // namespace cedar_amd
