// 27-point 8-colour Gauss-Seidel with inter-plane partial sums (the resident solver's sweep on big levels).
// Replaces BMG3_SymStd_relax_GS (reference src/3d/ftn/BMG3_SymStd_relax_GS.f90:80-138) under north_star's contract
// (residual histories within 1e-10 relative), NOT bit for bit: the 26-term sum of :104-131 is re-associated for the
// points of the second k-parity.  The bit-identical sweep (relax3d.hip) stays the drop-in and the fallback.
//
// Why.  The sweep relaxes the planes of one k-parity (A) and then the planes between them (B).  The nine inter-plane
// slot-rows stored on a plane couple it to the plane below; a row task of an A plane k reads them for its own update
// (18 slot-rows: plane k's for k-1, plane k+1's for k+1), and in the reference order the B planes k-1 and k+1 read
// the same 18 again half a sweep later: the operator's inter-plane part crosses HBM twice per sweep (2.18 of 5.55 ms
// at 512^3).  Here the A task, while it holds those coefficients in registers, also forms what
// the B points will need from them:
//     Tb(Y) = sum over the nine points X of plane k below Y of  coefficient(X,Y) * q_new(X)     (Y in plane k+1)
//     Tt(Y) = the same for the nine points above Y                                             (Y in plane k-1)
// -- the coupling of X to Y is one stored entry, read by X's task as X's coefficient towards Y -- and the B task
// computes  q(Y) = (qf + eight in-plane terms + Tb + Tt) / diag  from two partial-sum rows instead of 18 slot-rows
// and six q rows.  Same products as the reference, summed in a different order.
//
// How.  A task (row j of plane k, both i-colours, lane p owns the pair (2p+1, 2p+2)) multiplies its two fresh values
// with their 18 inter-plane coefficients: contributions to the rows j-1, j, j+1 of the planes k+1 and k-1.  Products
// for a neighbouring lane's points travel through LDS; each lane adds what belongs to its own pair into LDS
// accumulators of the target rows (a ring of four rows per side: a target row collects from three consecutive source
// rows, and the plane-fused walk F, F, S, F, S, .. completes two target rows at every S task).  A target row is
// complete only if its three source rows belong to the same workgroup's run: the rows at the ends of a run, the rows
// between runs, rows 1 and ny, and the B planes next to a ghost plane keep the reference order (exact row tasks in
// small launches before and after the partial-sum launch of the B planes) -- 3 rows in 2*frun.
// Ghost columns contribute like interior points (the lane at either end of the row multiplies the ghost's old value).
//
// Scratch: one array T of the vector's size per level: Tb of B plane k' in plane k' of T, Tt in plane k'+1 (the slot of
// the A plane above, which needs none).
#include "common.h"
#include "relax27_dev.h"
#include "relax3_psum.h"
#include <vector>

namespace cedar_amd {

// geometry of the plane-fused walk (relax27_plane): F rows j = 1+jbF+2f (f < nF) first, S rows j = 2-jbF+2g between them
struct PsumGeom { int jbF, frun, nF, nS, nrun; };

__host__ __device__ static inline PsumGeom psum_geom(int JJ, int jbF, int frun)
{
	PsumGeom g;
	g.jbF = jbF; g.frun = frun;
	g.nF = (JJ - 2 - jbF + 1) / 2;
	g.nS = (JJ - 2 - (1 - jbF) + 1) / 2;
	g.nrun = (g.nF + frun - 1) / frun;
	return g;
}

// rows lo .. hi one workgroup of the walk relaxes (run `run` of a plane)
__host__ __device__ static inline void psum_run_range(const PsumGeom &g, int JJ, int run, int &lo, int &hi)
{
	const int f0 = run * g.frun, f1 = (f0 + g.frun < g.nF) ? f0 + g.frun : g.nF;
	lo = 1 + g.jbF + 2 * f0;
	hi = 1 + g.jbF + 2 * (f1 - 1);
	if (g.jbF && f0 == 0) lo = 1;                    // the S row 1 goes with run 0
	if (f1 == g.nF && hi + 1 <= JJ - 2) hi = hi + 1; // the S row beyond the last F row goes with the last run
}

// a row of a B plane has complete partial sums iff its three source rows were relaxed by one workgroup
__host__ __device__ static inline bool psum_row_ok(const PsumGeom &g, int JJ, int j)
{
	int f = (j - 1 - g.jbF) >> 1;
	if (f < 0) f = 0;
	if (f >= g.nF) f = g.nF - 1;
	int lo, hi;
	psum_run_range(g, JJ, f / g.frun, lo, hi);
	return lo + 1 <= j && j <= hi - 1;
}

// Which planes take part.  nbr: bit 0 / 1 = a neighbouring RANK owns the plane below / above this box (slab
// decomposition; 0 on a single GPU).  An A plane next to such a ghost plane waits for the halo and runs as a small piece
// of its own (row kernels): it leaves no partial sums, and the B plane beside it keeps the reference order, like the B
// planes next to any ghost plane (their other neighbour plane is not an A plane of this box).
__host__ __device__ static inline bool psum_a_gives(int k, int KK, int nbr)
{
	return k >= 1 && k <= KK - 2 && !(k == 1 && (nbr & 1)) && !(k == KK - 2 && (nbr & 2));
}
__host__ __device__ static inline bool psum_b_takes(int kp, int KK, int nbr)
{
	return kp - 1 >= 1 && kp + 1 <= KK - 2 && psum_a_gives(kp - 1, KK, nbr) && psum_a_gives(kp + 1, KK, nbr);
}

// X's coefficient towards the point (DI, DJ) of the plane above (SIDE 0) / below (SIDE 1): see offdiag27
template <int SIDE, int DI, int DJ>
__device__ __forceinline__ real_t kcoef(const C27 &c)
{
	if (SIDE == 0) {
		if (DI == 0 && DJ == 0) return c.b_t;
		if (DI == -1 && DJ == 0) return c.be_t;
		if (DI == -1 && DJ == 1) return c.bse_nt;
		if (DI == 0 && DJ == 1) return c.bs_nt;
		if (DI == 1 && DJ == 1) return c.bsw_net;
		if (DI == 1 && DJ == 0) return c.bw_et;
		if (DI == 1 && DJ == -1) return c.bnw_et;
		if (DI == 0 && DJ == -1) return c.bn_t;
		return c.bne_t; // (-1,-1)
	}
	if (DI == 0 && DJ == 0) return c.b;
	if (DI == -1 && DJ == 0) return c.bw;
	if (DI == -1 && DJ == 1) return c.bnw_n;
	if (DI == 0 && DJ == 1) return c.bn_n;
	if (DI == 1 && DJ == 1) return c.bne_ne;
	if (DI == 1 && DJ == 0) return c.be_e;
	if (DI == 1 && DJ == -1) return c.bse_e;
	if (DI == 0 && DJ == -1) return c.bs;
	return c.bsw; // (-1,-1)
}

// LDS of one workgroup of relax27_planeA
template <int BS>
struct PsumLds {
	real_t xch[2][BS + 2];      // first colour's fresh values (relax27_row_task)
	real_t exP[6][BS + 1];      // products for the previous lane's second point: lane p writes [p], lane p reads [p+1]
	real_t exN[6][BS + 1];      // products for the next lane's first point:      lane p writes [p+1], lane p reads [p]
	real_t acc[4][2][2 * BS];   // accumulators: [target row & 3][side][point of the row]
};

// A task: relax row (j,k) like relax27_row_task (same arithmetic, same values), then add this row's contributions to
// the partial sums of the rows j-1, j, j+1 of the planes k+1 (side 0, Tb) and k-1 (side 1, Tt).
//   ISF: an F row of the walk -- the first contributor of the targets j and j+1 (assign), the second of j-1 (add);
//        an S row adds to all three and completes the targets j-1 and j, which are stored if lo+1 <= row <= hi-1.
//   stU / stD: the B plane above / below takes partial sums (it is not next to a ghost plane)
template <int BS, bool EFIRST, bool NT>
__device__ __forceinline__ void relax27_row_task_A(const Op3 &A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                                                   real_t *__restrict__ T, int II, size_t sj, size_t sk, size_t j, size_t k,
                                                   PsumLds<BS> &S, int t, bool isf, int lo, int hi, bool stU, bool stD,
                                                   unsigned skm)
{
	const size_t row = j * sj + k * sk, rowA = j * A.SJ + k * A.SK;
	const int p = threadIdx.x;
	const int ie = 2 * p + 1, io = 2 * p + 2;
	const bool e_ok = ie <= II - 2;
	const bool o_ok = io <= II - 2;
	const bool two = io + 1 <= II - 1;
	real_t *xch = S.xch[t & 1];

	real_t e_new = 0.0, o_new = 0.0;
	C27 ce, co;
	real_t qe[3][3][3], qo[3][3][3];
	real_t qfe = 0, qfo = 0, sre = 0, sro = 0;

	if (e_ok) {
		load_pair27<NT, false, NT, 0>(A, qf, q, rowA, row, sj, sk, ie, io, two, ce, co, qe, qo, qfe, qfo);
		real_t a_, b_;
		ldpair(A.sor + j * A.rSJ + k * A.rSK + ie, true, a_, b_); sre = a_; sro = b_;
	}
	// old values of the ghost columns (sources of contributions like any other point of the row)
	const real_t ghostL = qe[1][1][0];                    // lane 0: q(0,j,k)
	const real_t ghostR = o_ok ? qo[1][1][2] : qe[1][1][2]; // the lane holding the last interior point: q(II-1,j,k)

	bool ske = false, sko = false; // points relaxed ahead of the launch keep their value (relax27_dev.h skip27_lane)
	if (skm) skip27_lane(skm, p, (II - 2 + 1) / 2, ske, sko);
	if (EFIRST) {
		if (e_ok) {
			e_new = ske ? qe[1][1][1] : offdiag27(qfe, ce, qe) * sre;
			xch[p] = e_new;
		}
		__syncthreads();
		if (o_ok) {
			qo[1][1][0] = e_new;
			if (io + 1 <= II - 2) qo[1][1][2] = xch[p + 1];
			o_new = sko ? qo[1][1][1] : offdiag27(qfo, co, qo) * sro;
		}
	} else {
		if (o_ok) {
			o_new = sko ? qo[1][1][1] : offdiag27(qfo, co, qo) * sro;
			xch[p + 1] = o_new;
		}
		__syncthreads();
		if (e_ok) {
			if (p > 0) qe[1][1][0] = xch[p];
			if (o_ok) qe[1][1][2] = o_new;
			e_new = ske ? qe[1][1][1] : offdiag27(qfe, ce, qe) * sre;
		}
	}
	if (e_ok) {
		if (o_ok) {
			d2u v; v.x = e_new; v.y = o_new;
			*reinterpret_cast<d2u *>(q + row + ie) = v;
		} else {
			q[row + ie] = e_new;
		}
	}
	if (!stU && !stD) return; // (uniform) no B plane next to this one takes partial sums

	// ---- contributions.  Sources of this lane: e (fresh), o (fresh; the right ghost column's old value when the row
	// ends on e).  own_e / own_o: what the lane's two sources give its own two targets; sendP / sendN: what they give
	// the previous lane's second point and the next lane's first point.
	const real_t se = e_new, so_ = o_ok ? o_new : ghostR;
	const bool last_o = o_ok && io == II - 2; // the right ghost column is this lane's "next" source
	real_t own_e[6], own_o[6];
	const real_t *__restrict__ sop = A.so;
	const size_t PS = A.SS, aj = A.SJ, ak = A.SK;
	if (e_ok) {
#define GROUP(SIDE, DJ)                                                                                           \
	{                                                                                                             \
		constexpr int g_ = SIDE * 3 + (DJ + 1);                                                                   \
		own_e[g_] = se * kcoef<SIDE, 0, DJ>(ce);                                                                  \
		if (o_ok || io == II - 1) own_e[g_] = own_e[g_] + so_ * kcoef<SIDE, -1, DJ>(co);                          \
		S.exP[g_][p] = se * kcoef<SIDE, -1, DJ>(ce);                                                              \
		if (o_ok) {                                                                                               \
			own_o[g_] = se * kcoef<SIDE, 1, DJ>(ce) + so_ * kcoef<SIDE, 0, DJ>(co);                               \
			S.exN[g_][p + 1] = so_ * kcoef<SIDE, 1, DJ>(co);                                                      \
		}                                                                                                         \
	}
		GROUP(0, -1) GROUP(0, 0) GROUP(0, 1) GROUP(1, -1) GROUP(1, 0) GROUP(1, 1)
#undef GROUP
		if (p == 0) {
			// left ghost column X = (0,j,k): its coefficients towards column 1 (DI = +1) are entries stored at i = 1
			S.exN[0][0] = ghostL * sop[KBNW * PS + rowA + ak + 1];      // side 0, DJ -1: bnw_et
			S.exN[1][0] = ghostL * sop[KBW * PS + rowA + ak + 1];       // side 0, DJ  0: bw_et
			S.exN[2][0] = ghostL * sop[KBSW * PS + rowA + aj + ak + 1]; // side 0, DJ +1: bsw_net
			S.exN[3][0] = ghostL * sop[KBSE * PS + rowA + 1];           // side 1, DJ -1: bse_e
			S.exN[4][0] = ghostL * sop[KBE * PS + rowA + 1];            // side 1, DJ  0: be_e
			S.exN[5][0] = ghostL * sop[KBNE * PS + rowA + aj + 1];      // side 1, DJ +1: bne_ne
		}
		if (last_o) {
			// right ghost column X = (II-1,j,k): its coefficients towards column II-2 (DI = -1), stored at i = II-1
			const size_t x = (size_t)(II - 1);
			S.exP[0][p + 1] = ghostR * sop[KBNE * PS + rowA + ak + x];      // side 0, DJ -1: bne_t
			S.exP[1][p + 1] = ghostR * sop[KBE * PS + rowA + ak + x];       // side 0, DJ  0: be_t
			S.exP[2][p + 1] = ghostR * sop[KBSE * PS + rowA + aj + ak + x]; // side 0, DJ +1: bse_nt
			S.exP[3][p + 1] = ghostR * sop[KBSW * PS + rowA + x];           // side 1, DJ -1: bsw
			S.exP[4][p + 1] = ghostR * sop[KBW * PS + rowA + x];            // side 1, DJ  0: bw
			S.exP[5][p + 1] = ghostR * sop[KBNW * PS + rowA + aj + x];      // side 1, DJ +1: bnw_n
		}
	}
	__syncthreads();
	if (e_ok) {
#pragma unroll
		for (int g = 0; g < 6; g++) {
			const int side = g / 3, dj = g % 3 - 1;
			real_t *a = S.acc[((int)j + dj) & 3][side] + 2 * p;
			const real_t te = (S.exN[g][p] + own_e[g]);
			const bool add = !isf || dj < 0;
			a[0] = add ? a[0] + te : te;
			if (o_ok) {
				const real_t to = own_o[g] + S.exP[g][p + 1];
				a[1] = add ? a[1] + to : to;
			}
		}
		if (!isf) {
			// targets j-1 and j are complete: Tb of plane k+1 lives in plane k+1 of T, Tt of plane k-1 in plane k
#pragma unroll
			for (int c = 0; c < 2; c++) {
				const int r = (int)j - 1 + c;
				if (r < lo + 1 || r > hi - 1) continue;
#pragma unroll
				for (int side = 0; side < 2; side++) {
					if (side == 0 ? !stU : !stD) continue;
					const real_t *a = S.acc[r & 3][side] + 2 * p;
					real_t *dst = T + (size_t)r * sj + (side == 0 ? k + 1 : k) * sk + ie;
					if (o_ok) {
						d2u v; v.x = a[0]; v.y = a[1];
						*reinterpret_cast<d2u *>(dst) = v;
					} else
						dst[0] = a[0];
				}
			}
		}
	}
}

// A launch: the plane-fused walk of relax27_plane over the planes of the first k-parity, with the partial sums
template <int BS, bool EFIRST, bool NT>
__global__ __launch_bounds__(BS) void relax27_planeA(const Op3 A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                                                      real_t *__restrict__ T, int II, int JJ, int KK, int kb, int kr0,
                                                      int nrk, int nbr, PsumGeom gm, PsumSkip skp)
{
	__shared__ PsumLds<BS> S;
	const unsigned nblk = (unsigned)nrk * (unsigned)gm.nrun;
	const unsigned L = xcd_remap(blockIdx.x, nblk);
	if (L >= nblk) return;
	const int kr = (int)(L / (unsigned)gm.nrun), run = (int)(L % (unsigned)gm.nrun);
	const int jbF = gm.jbF, nF = gm.nF, nS = gm.nS;
	const int f0 = run * gm.frun, f1 = min(nF, f0 + gm.frun);
	int lo, hi;
	psum_run_range(gm, JJ, run, lo, hi);
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	const int ki = 1 + kb + 2 * (kr + kr0);
	const size_t k = (size_t)ki;
	const bool stU = psum_b_takes(ki + 1, KK, nbr), stD = psum_b_takes(ki - 1, KK, nbr);
	int t = 0;
	for (int f = f0; f < f1; f++) {
		relax27_row_task_A<BS, EFIRST, NT>(A, qf, q, T, II, sj, sk, (size_t)(1 + jbF + 2 * f), k, S, t, true, lo, hi, stU, stD,
		                                   psum_skip_mask(skp, 1 + jbF + 2 * f, true));
		t++;
		const int g = jbF ? f : f - 1;
		const bool have = jbF ? (f > f0 || f == 0) : (f > f0);
		if (have && g >= 0 && g < nS) {
			__syncthreads();
			relax27_row_task_A<BS, EFIRST, NT>(A, qf, q, T, II, sj, sk, (size_t)(2 - jbF + 2 * g), k, S, t, false, lo, hi, stU, stD,
			                                   psum_skip_mask(skp, 2 - jbF + 2 * g, false));
			t++;
		}
	}
	if (f1 == nF && f1 > f0) {
		const int g = jbF ? nF : nF - 1;
		if (g < nS) {
			__syncthreads();
			relax27_row_task_A<BS, EFIRST, NT>(A, qf, q, T, II, sj, sk, (size_t)(2 - jbF + 2 * g), k, S, t, false, lo, hi, stU, stD,
			                                   psum_skip_mask(skp, 2 - jbF + 2 * g, false));
			t++;
		}
	}
}

// B task: row (j,k) of a plane of the second k-parity from its partial sums.  Term order: qf, the eight in-plane terms
// in the reference's order (relax_GS.f90:104-112), then Tb, then Tt.
template <int BS, bool EFIRST, bool NT>
__device__ __forceinline__ void relax27_row_task_B(const Op3 &A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                                                   const real_t *__restrict__ T, int II, size_t sj, size_t sk, size_t j,
                                                   size_t k, real_t *xch, unsigned skm)
{
	const size_t row = j * sj + k * sk, rowA = j * A.SJ + k * A.SK;
	const int p = threadIdx.x;
	const int ie = 2 * p + 1, io = 2 * p + 2;
	const bool e_ok = ie <= II - 2;
	const bool o_ok = io <= II - 2;
	const bool two = io + 1 <= II - 1;
	const real_t *__restrict__ so = A.so;
	const size_t PS = A.SS, aj = A.SJ;

	real_t pw_e = 0, pw_o = 0, ps_e = 0, ps_o = 0, psw_e = 0, psw_o = 0, pnwn_e = 0, pnwn_o = 0, psn_e = 0, psn_o = 0;
	real_t pswne_e = 0, pswne_o = 0, pwe_e = 0, pwe_o = 0, pnwe_e = 0, pnwe_o = 0;
	real_t qfe = 0, qfo = 0, sre = 0, sro = 0, tbe = 0, tbo = 0, tte = 0, tto = 0;
	real_t w[3][4]; // q(ie-1 .. io+1) of the rows j-1, j, j+1
	if (e_ok) {
		// [i] pattern: (value at ie, value at io); KPW is this task's alone, the others are shared with the task of row j-1 / j+1
		ldpair_so<NT>(so + KPW * PS + rowA + ie, true, pw_e, pw_o);
		ldpair(so + KPS * PS + rowA + ie, true, ps_e, ps_o);
		ldpair(so + KPSW * PS + rowA + ie, true, psw_e, psw_o);
		ldpair(so + KPNW * PS + rowA + aj + ie, true, pnwn_e, pnwn_o);
		ldpair(so + KPS * PS + rowA + aj + ie, true, psn_e, psn_o);
		// [i+1] pattern: (value at io, value at io+1)
		ldpair(so + KPSW * PS + rowA + aj + io, two, pswne_e, pswne_o);
		ldpair_so<NT>(so + KPW * PS + rowA + io, two, pwe_e, pwe_o);
		ldpair(so + KPNW * PS + rowA + io, two, pnwe_e, pnwe_o);
		ldpair(A.sor + j * A.rSJ + k * A.rSK + ie, true, sre, sro);
		ldpair(qf + row + ie, true, qfe, qfo);
		ldpair_so<NT>(T + row + ie, true, tbe, tbo);
		ldpair_so<NT>(T + row + sk + ie, true, tte, tto);
#pragma unroll
		for (int dj = 0; dj < 3; dj++) {
			const real_t *r = q + row + (ptrdiff_t)(dj - 1) * (ptrdiff_t)sj;
			ldpair(r + ie - 1, true, w[dj][0], w[dj][1]);
			ldpair(r + io, two, w[dj][2], w[dj][3]);
		}
	}
	// s(X) for X = e: west = w[.][0], centre column = w[.][1], east = w[.][2]; for X = o shift by one
#define INPLANE(qf_, pw_, pnwn_, psn_, pswne_, pwe_, pnwe_, ps_, psw_, W, C, E, tb_, tt_) \
	((((((((((qf_ + pw_ * W(1)) + pnwn_ * W(2)) + psn_ * C(2)) + pswne_ * E(2)) + pwe_ * E(1)) + pnwe_ * E(0)) + ps_ * C(0)) + psw_ * W(0)) + tb_) + tt_)
	real_t e_new = 0.0, o_new = 0.0;
	bool ske = false, sko = false;
	if (skm) skip27_lane(skm, p, (II - 2 + 1) / 2, ske, sko);
	const real_t e_old = w[1][1], o_old = w[1][2];
	if (EFIRST) {
		if (e_ok) {
#define W_(d) w[d][0]
#define C_(d) w[d][1]
#define E_(d) w[d][2]
			e_new = INPLANE(qfe, pw_e, pnwn_e, psn_e, pswne_e, pwe_e, pnwe_e, ps_e, psw_e, W_, C_, E_, tbe, tte) * sre;
#undef W_
#undef C_
#undef E_
			if (ske) e_new = e_old;
			xch[p] = e_new;
		}
		__syncthreads();
		if (o_ok) {
			w[1][1] = e_new;
			if (io + 1 <= II - 2) w[1][3] = xch[p + 1];
#define W_(d) w[d][1]
#define C_(d) w[d][2]
#define E_(d) w[d][3]
			o_new = INPLANE(qfo, pw_o, pnwn_o, psn_o, pswne_o, pwe_o, pnwe_o, ps_o, psw_o, W_, C_, E_, tbo, tto) * sro;
#undef W_
#undef C_
#undef E_
			if (sko) o_new = o_old;
		}
	} else {
		if (o_ok) {
#define W_(d) w[d][1]
#define C_(d) w[d][2]
#define E_(d) w[d][3]
			o_new = INPLANE(qfo, pw_o, pnwn_o, psn_o, pswne_o, pwe_o, pnwe_o, ps_o, psw_o, W_, C_, E_, tbo, tto) * sro;
#undef W_
#undef C_
#undef E_
			if (sko) o_new = o_old;
			xch[p + 1] = o_new;
		}
		__syncthreads();
		if (e_ok) {
			if (p > 0) w[1][0] = xch[p];
			if (o_ok) w[1][2] = o_new;
#define W_(d) w[d][0]
#define C_(d) w[d][1]
#define E_(d) w[d][2]
			e_new = INPLANE(qfe, pw_e, pnwn_e, psn_e, pswne_e, pwe_e, pnwe_e, ps_e, psw_e, W_, C_, E_, tbe, tte) * sre;
#undef W_
#undef C_
#undef E_
			if (ske) e_new = e_old;
		}
	}
#undef INPLANE
	if (e_ok) {
		if (o_ok) {
			d2u v; v.x = e_new; v.y = o_new;
			*reinterpret_cast<d2u *>(q + row + ie) = v;
		} else {
			q[row + ie] = e_new;
		}
	}
}

// B launch: the walk over the planes kr0 .. kr0+nrk-1 of the second k-parity; only rows with complete partial sums
// (the others were / will be relaxed by relax27_rows_sel in the reference order)
template <int BS, bool EFIRST, bool NT>
__global__ __launch_bounds__(BS) void relax27_planeB(const Op3 A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                                                      const real_t *__restrict__ T, int II, int JJ, int KK, int kb, int kr0,
                                                      int nrk, PsumGeom gm, PsumSkip skp)
{
	__shared__ real_t xch[2][BS + 2];
	const unsigned nblk = (unsigned)nrk * (unsigned)gm.nrun;
	const unsigned L = xcd_remap(blockIdx.x, nblk);
	if (L >= nblk) return;
	const int kr = (int)(L / (unsigned)gm.nrun), run = (int)(L % (unsigned)gm.nrun);
	const int jbF = gm.jbF, nF = gm.nF, nS = gm.nS;
	const int f0 = run * gm.frun, f1 = min(nF, f0 + gm.frun);
	int lo, hi;
	psum_run_range(gm, JJ, run, lo, hi);
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	const size_t k = (size_t)(1 + kb + 2 * (kr + kr0));
	int t = 0;
	for (int f = f0; f < f1; f++) {
		const int jf = 1 + jbF + 2 * f;
		const unsigned mf = psum_skip_mask(skp, jf, true);
		if (jf >= lo + 1 && jf <= hi - 1 && !(mf & 0x100u)) {
			if (t) __syncthreads();
			relax27_row_task_B<BS, EFIRST, NT>(A, qf, q, T, II, sj, sk, (size_t)jf, k, xch[t & 1], mf);
			t++;
		}
		const int g = jbF ? f : f - 1;
		const bool have = jbF ? (f > f0 || f == 0) : (f > f0);
		const int js = 2 - jbF + 2 * g;
		const unsigned ms = psum_skip_mask(skp, js, false);
		if (have && g >= 0 && g < nS && js >= lo + 1 && js <= hi - 1 && !(ms & 0x100u)) {
			if (t) __syncthreads();
			relax27_row_task_B<BS, EFIRST, NT>(A, qf, q, T, II, sj, sk, (size_t)js, k, xch[t & 1], ms);
			t++;
		}
	}
	// (the S row beyond the last F row is row hi: never a partial-sum row)
}

// Rows of the B planes that keep the reference order, in one launch per row class.
//   cls 0 (before relax27_planeB): F rows; cls 1 (after it): S rows.
//   Workgroups [0, nx*nrows): every row of the class in the nx <= 2 planes x0, x1 that take no partial sums at all; then per
//   partial-sum plane kr in [elo, ehi) ncand candidates: cls 0: the first and the last F row of every run; cls 1: the S row
//   after every run but the last, row 1 (jbF = 1), the S row beyond the last F.
template <int BS, bool EFIRST, bool NT>
__global__ __launch_bounds__(BS) void relax27_rows_sel(const Op3 A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                                                        int II, int JJ, int KK, int kb, int elo, int ehi, int x0, int x1,
                                                        PsumGeom gm, int cls, PsumSkip skp)
{
	__shared__ real_t xch[BS + 2];
	const int nrows = cls ? gm.nS : gm.nF;
	const int nx = (x0 >= 0 ? 1 : 0) + (x1 >= 0 ? 1 : 0);
	int w = (int)blockIdx.x;
	int kr, j;
	if (w < nx * nrows) {
		const int tpl = w / nrows, r = w % nrows;
		kr = (tpl == 0 && x0 >= 0) ? x0 : x1;
		j = (cls ? 2 - gm.jbF : 1 + gm.jbF) + 2 * r;
	} else {
		w -= nx * nrows;
		const int ncand = cls ? gm.nrun + 1 : 2 * gm.nrun;
		kr = elo + w / ncand;
		const int c = w % ncand;
		if (kr >= ehi) return;
		if (cls == 0) {
			const int run = c >> 1, end = c & 1;
			const int f0 = run * gm.frun, f1 = min(gm.nF, f0 + gm.frun);
			if (end && f1 - 1 == f0) return; // a run of one row: same row twice
			j = 1 + gm.jbF + 2 * (end ? f1 - 1 : f0);
			if (psum_row_ok(gm, JJ, j)) return;
		} else {
			if (c < gm.nrun - 1) {
				const int f1 = (c + 1) * gm.frun; // first F row of the next run; the S row below it
				j = 1 + gm.jbF + 2 * f1 - 1;
			} else if (c == gm.nrun - 1) {
				if (!gm.jbF) return;
				j = 1;
			} else {
				j = 1 + gm.jbF + 2 * (gm.nF - 1) + 1;
				if (j > JJ - 2) return;
			}
			if (j < 1 || j > JJ - 2) return;
		}
	}
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	const unsigned m = psum_skip_mask(skp, j, cls == 0);
	if (m & 0x100u) return;
	relax27_row_task<BS, EFIRST, NT, NT, 0, false, true>(A, qf, q, II, sj, sk, (size_t)j, (size_t)(1 + kb + 2 * kr), xch, m);
}

// rows j = j0 + jstep*jr of all planes of parity kb, reference order (the S rows between the runs of the A launch)
template <int BS, bool EFIRST, bool NT>
__global__ __launch_bounds__(BS) void relax27_rows_between(const Op3 A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                                                            int II, int JJ, int KK, int j0, int jstep, int nrj, int kb, int kr0,
                                                            int nrk, PsumSkip skp)
{
	__shared__ real_t xch[BS + 2];
	const int w = (int)blockIdx.x;
	if (w >= nrj * nrk) return;
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	const int j = j0 + jstep * (w % nrj);
	const unsigned m = psum_skip_mask(skp, j, false); // S rows of the walk
	if (m & 0x100u) return;
	relax27_row_task<BS, EFIRST, NT, NT, 0, false, true>(A, qf, q, II, sj, sk, (size_t)j, (size_t)(1 + kb + 2 * (kr0 + w / nrj)), xch, m);
}

// A phase on the planes kr0 .. kr0+nrk-1 of the first k-parity kb: the plane-fused walk with partial sums, then the S rows
// between runs (reference order)
template <int BS, bool EFIRST>
static void phase_a(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int kb, int kr0, int nrk,
                    int nbr, const PsumGeom &gm, hipStream_t st, const PsumSkip &skp)
{
	if (nrk <= 0) return;
	hipLaunchKernelGGL((relax27_planeA<BS, EFIRST, true>), dim3(xcd_grid((unsigned)nrk * (unsigned)gm.nrun)), dim3(BS), 0, st,
	                   A, qf, q, T, II, JJ, KK, kb, kr0, nrk, nbr, gm, skp);
	if (gm.nrun > 1)
		hipLaunchKernelGGL((relax27_rows_between<BS, EFIRST, true>), dim3((unsigned)((gm.nrun - 1) * nrk)), dim3(BS), 0, st,
		                   A, qf, q, II, JJ, KK, (gm.jbF ? 1 : 0) + 2 * gm.frun, 2 * gm.frun, gm.nrun - 1, kb, kr0, nrk, skp);
}

// B phase on the planes kr0 .. kr0+nrk-1 of the second k-parity kb: the planes that take partial sums form a contiguous
// range; at most one plane at either end of the piece keeps the reference order altogether
template <int BS, bool EFIRST>
static void phase_b(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int kb, int kr0, int nrk,
                    int nbr, const PsumGeom &gm, hipStream_t st, const PsumSkip &skp)
{
	if (nrk <= 0) return;
	int elo = kr0, ehi = kr0 + nrk, x0 = -1, x1 = -1;
	while (elo < ehi && !psum_b_takes(1 + kb + 2 * elo, KK, nbr)) elo++;
	while (ehi > elo && !psum_b_takes(1 + kb + 2 * (ehi - 1), KK, nbr)) ehi--;
	// planes of the piece outside [elo, ehi): in a piece of the sweep there is at most one at either end; more (a piece
	// made of planes next to ghost planes only) go through the reference-order rows two at a time
	std::vector<int> exact;
	for (int kr = kr0; kr < kr0 + nrk; kr++)
		if (kr < elo || kr >= ehi) exact.push_back(kr);
	const int nel = ehi - elo;
	size_t done = 0;
	do {
		x0 = done < exact.size() ? exact[done] : -1;
		x1 = done + 1 < exact.size() ? exact[done + 1] : -1;
		const bool first = done == 0;
		done += 2;
		const int nxp = (x0 >= 0 ? 1 : 0) + (x1 >= 0 ? 1 : 0);
		for (int cls = 0; cls < 2; cls++) {
			if (cls == 1 && first && nel > 0)
				hipLaunchKernelGGL((relax27_planeB<BS, EFIRST, true>), dim3(xcd_grid((unsigned)nel * (unsigned)gm.nrun)), dim3(BS), 0, st,
				                   A, qf, q, T, II, JJ, KK, kb, elo, nel, gm, skp);
			const int nrows = cls ? gm.nS : gm.nF, ncand = cls ? gm.nrun + 1 : 2 * gm.nrun;
			const int nwg = nxp * nrows + (first ? nel * ncand : 0);
			if (nwg > 0)
				hipLaunchKernelGGL((relax27_rows_sel<BS, EFIRST, true>), dim3((unsigned)nwg), dim3(BS), 0, st,
				                   A, qf, q, II, JJ, KK, kb, first ? elo : 0, first ? ehi : 0, x0, x1, gm, cls, skp);
		}
	} while (done < exact.size());
}

template <int BS, bool EFIRST>
static void sweep_psum(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int frun, hipStream_t st)
{
	const bool up = EFIRST; // UP: even i, j-parity 0 rows, k-parity 0 planes first; DOWN the reverse
	const int jbF = up ? 0 : 1, kbA = up ? 0 : 1, kbB = 1 - kbA;
	const PsumGeom gm = psum_geom(JJ, jbF, frun);
	const PsumSkip none = psum_skip_none();
	phase_a<BS, EFIRST>(A, qf, q, T, II, JJ, KK, kbA, 0, (KK - 2 - kbA + 1) / 2, 0, gm, st, none);
	phase_b<BS, EFIRST>(A, qf, q, T, II, JJ, KK, kbB, 0, (KK - 2 - kbB + 1) / 2, 0, gm, st, none);
}

// One k-parity of planes of a sweep, the unit between two halo exchanges of a slab decomposition (relax3_planes27): the
// planes kr0 .. kr0+nrk-1 of parity kb; nbr: which ghost planes belong to a neighbouring rank.
void relax3_planes27_psum(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int kb, int up, int kr0,
                          int nrk, int nbr, int frun, hipStream_t st, const PsumSkip *skip)
{
	const int npairs = (II - 2 + 1) / 2;
	const PsumGeom gm = psum_geom(JJ, up ? 0 : 1, frun);
	const PsumSkip skp = skip ? *skip : psum_skip_none();
	const bool first_parity = up ? kb == 0 : kb == 1;
#define PSUM_PH(B)                                                                                                  \
	do {                                                                                                            \
		if (first_parity) {                                                                                         \
			if (up) phase_a<B, true>(A, qf, q, T, II, JJ, KK, kb, kr0, nrk, nbr, gm, st, skp);                          \
			else phase_a<B, false>(A, qf, q, T, II, JJ, KK, kb, kr0, nrk, nbr, gm, st, skp);                            \
		} else {                                                                                                    \
			if (up) phase_b<B, true>(A, qf, q, T, II, JJ, KK, kb, kr0, nrk, nbr, gm, st, skp);                          \
			else phase_b<B, false>(A, qf, q, T, II, JJ, KK, kb, kr0, nrk, nbr, gm, st, skp);                            \
		}                                                                                                           \
	} while (0)
	if (npairs <= 64) PSUM_PH(64);
	else if (npairs <= 128) PSUM_PH(128);
	else PSUM_PH(256);
#undef PSUM_PH
}

// the level can take the partial-sum sweep: rows of at most 512 points (LDS of relax27_planeA), runs of frun F rows
bool relax3_psum_ok(int II, int JJ, int KK, int frun)
{
	return frun > 0 && II >= 3 && JJ >= 3 && KK >= 3 && (II - 2 + 1) / 2 <= 256;
}

void relax3_gs27_psum(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int updown, int frun,
                      hipStream_t st)
{
	const int npairs = (II - 2 + 1) / 2;
	const bool up = updown == BMG_UP;
#define PSUM_BS(B)                                                                   \
	do {                                                                             \
		if (up) sweep_psum<B, true>(A, qf, q, T, II, JJ, KK, frun, st);              \
		else sweep_psum<B, false>(A, qf, q, T, II, JJ, KK, frun, st);                \
	} while (0)
	if (npairs <= 64) PSUM_BS(64);
	else if (npairs <= 128) PSUM_BS(128);
	else PSUM_BS(256);
#undef PSUM_BS
}

} // namespace cedar_amd
