// Device pieces of the 27-point sweep shared by relax3d.hip (reference order, bit-identical) and
// relax3d_psum.hip (inter-plane partial sums): coefficient set of a point, term order, pair loads, row task.
#pragma once
#include "common.h"

namespace cedar_amd {

// ------------------------------------------------------------------ 27-pt
// coefficients seen from one grid point X=(i,j,k); names = slot _ where stored
struct C27 {
	real_t pw, ps, psw, b, bw, bs, bsw;       // stored at X
	real_t pnw_n, ps_n, bnw_n, bn_n;          // stored at X+(0,1,0)
	real_t b_t, be_t, bn_t, bne_t;            // stored at X+(0,0,1)
	real_t bse_nt, bs_nt;                     // stored at X+(0,1,1)
	real_t psw_ne, bne_ne;                    // stored at X+(1,1,0)
	real_t pw_e, pnw_e, be_e, bse_e;          // stored at X+(1,0,0)
	real_t bsw_net;                           // stored at X+(1,1,1)
	real_t bw_et, bnw_et;                     // stored at X+(1,0,1)
};

// qq[dk+1][dj+1][di+1]; term order of BMG3_SymStd_relax_GS.f90:104-131
__device__ __forceinline__ real_t offdiag27(real_t qf, const C27 &c, const real_t (&qq)[3][3][3])
{
	real_t s = qf;
	s = s + c.pw * qq[1][1][0];
	s = s + c.pnw_n * qq[1][2][0];
	s = s + c.ps_n * qq[1][2][1];
	s = s + c.psw_ne * qq[1][2][2];
	s = s + c.pw_e * qq[1][1][2];
	s = s + c.pnw_e * qq[1][0][2];
	s = s + c.ps * qq[1][0][1];
	s = s + c.psw * qq[1][0][0];
	s = s + c.b * qq[0][1][1];
	s = s + c.bw * qq[0][1][0];
	s = s + c.bnw_n * qq[0][2][0];
	s = s + c.bn_n * qq[0][2][1];
	s = s + c.bne_ne * qq[0][2][2];
	s = s + c.be_e * qq[0][1][2];
	s = s + c.bse_e * qq[0][0][2];
	s = s + c.bs * qq[0][0][1];
	s = s + c.bsw * qq[0][0][0];
	s = s + c.b_t * qq[2][1][1];
	s = s + c.be_t * qq[2][1][0];
	s = s + c.bse_nt * qq[2][2][0];
	s = s + c.bs_nt * qq[2][2][1];
	s = s + c.bsw_net * qq[2][2][2];
	s = s + c.bw_et * qq[2][1][2];
	s = s + c.bnw_et * qq[2][0][2];
	s = s + c.bn_t * qq[2][0][1];
	s = s + c.bne_t * qq[2][0][0];
	return s;
}


// load the pair (ptr[0], ptr[1]); `two` false -> only ptr[0] is inside the row
__device__ __forceinline__ void ldpair(const real_t *__restrict__ p, bool two, real_t &a, real_t &b)
{
	if (two) {
		d2u v = *reinterpret_cast<const d2u *>(p);
		a = v.x; b = v.y;
	} else {
		a = p[0]; b = 0.0;
	}
}

// operator rows are used by exactly one workgroup of a launch: stream them past the caches
// (non-temporal) so that the q rows, which neighbouring workgroups share, stay resident
template <bool NT>
__device__ __forceinline__ void ldpair_so(const real_t *__restrict__ p, bool two, real_t &a, real_t &b)
{
	if (!NT) { ldpair(p, two, a, b); return; }
	if (two) {
		d2u v = __builtin_nontemporal_load(reinterpret_cast<const d2u *>(p));
		a = v.x; b = v.y;
	} else {
		a = __builtin_nontemporal_load(p); b = 0.0;
	}
}

// all operands of the pair (ie, io) of one row: 26 coefficients per point, qf, and the 3x3 q rows
// (offsets ie-1 .. io+1); 16-byte loads, `two` = element io+1 is still inside the row
// Load policy (which operator streams bypass the caches with non-temporal loads):
//   NT  = rows only this task reads in this launch (relax) -- streamed;
//   NTP = the three slots whose rows two row tasks of one plane share (kps, kpsw, kpnw): the plane-fused
//         relax pass keeps them cacheable so that the second task finds them in L2 / Infinity Cache;
//   NTO = the task's own row (offset 0) where that is its LAST use in the launch (residual: the rows at
//         j+1 / k+1 are read again by the neighbouring task, the own row is not).
// WI ("what if", experiments only, results wrong): bit 0 = every q row read from the task's own row, bit 1 = the
// slot-rows of plane k+1 read from plane k, bit 2 = every inter-plane slot read from KPW -- the loads then hit
// the caches and the timing shows what removing that traffic would be worth.  8 = the k-pair walk of relax27_plane.
#define WI_SLOT(slot) ((size_t)(((WI & 4) && ((slot) == KB || (slot) >= KBW)) ? KPW : (slot)))
template <bool NT, bool NTP = NT, bool NTO = NT, int WI = 0>
__device__ __forceinline__ void load_pair27(const Op3 &A, const real_t *__restrict__ qf,
                                            const real_t *__restrict__ q, size_t rowA, size_t row, size_t sj, size_t sk,
                                            int ie, int io, bool two, C27 &ce, C27 &co,
                                            real_t (&qe)[3][3][3], real_t (&qo)[3][3][3], real_t &qfe, real_t &qfo)
{
	// operator entry (slot, i, j+dj, k+dk) = A.so[slot*A.SS + rowA + dj*A.SJ + dk*A.SK + i]; vectors use row, sj, sk
	const real_t *__restrict__ so = A.so;
	const size_t PS = A.SS, aj = A.SJ, ak = (WI & 2) ? 0 : A.SK;
	if (WI & 1) { sj = 0; sk = 0; }
	// ---- [i]-pattern streams: (value at ie, value at io)
#define LD_I_(N, slot, off, fe, fo)                                                    \
{                                                                                  \
	real_t a_, b_;                                                                 \
	ldpair_so<N>(so + WI_SLOT(slot)*PS + rowA + (off) + ie, true, a_, b_);         \
	ce.fe = a_; co.fo = b_;                                                        \
}
#define LD_I(slot, off, fe, fo) LD_I_(NT, slot, off, fe, fo)
#define LD_IO(slot, fe, fo) LD_I_(NTO, slot, 0, fe, fo)
#define LD_IS(slot, off, fe, fo) LD_I_(NTP, slot, off, fe, fo)
#define LD_ISO(slot, fe, fo) LD_I_((NTP && NTO), slot, 0, fe, fo)
	LD_IO(KPW, pw, pw) LD_ISO(KPS, ps, ps) LD_ISO(KPSW, psw, psw) LD_IO(KB, b, b)
	LD_IO(KBW, bw, bw) LD_IO(KBS, bs, bs) LD_IO(KBSW, bsw, bsw)
	LD_IS(KPNW, aj, pnw_n, pnw_n) LD_IS(KPS, aj, ps_n, ps_n) LD_I(KBNW, aj, bnw_n, bnw_n) LD_I(KBN, aj, bn_n, bn_n)
	LD_I(KB, ak, b_t, b_t) LD_I(KBE, ak, be_t, be_t) LD_I(KBN, ak, bn_t, bn_t) LD_I(KBNE, ak, bne_t, bne_t)
	LD_I(KBSE, aj + ak, bse_nt, bse_nt) LD_I(KBS, aj + ak, bs_nt, bs_nt)
#undef LD_I
#undef LD_IO
#undef LD_IS
#undef LD_ISO
#undef LD_I_
	// ---- [i+1]-pattern streams: (value at ie+1 = io, value at io+1)
#define LD_IP_(N, slot, off, f)                                                        \
{                                                                                  \
	real_t a_, b_;                                                                 \
	ldpair_so<N>(so + WI_SLOT(slot)*PS + rowA + (off) + io, two, a_, b_);          \
	ce.f = a_; co.f = b_;                                                          \
}
#define LD_IP(slot, off, f) LD_IP_(NT, slot, off, f)
#define LD_IPO(slot, f) LD_IP_(NTO, slot, 0, f)
#define LD_IPS(slot, off, f) LD_IP_(NTP, slot, off, f)
#define LD_IPSO(slot, f) LD_IP_((NTP && NTO), slot, 0, f)
	LD_IPS(KPSW, aj, psw_ne) LD_IP(KBNE, aj, bne_ne)
	LD_IPO(KPW, pw_e) LD_IPSO(KPNW, pnw_e) LD_IPO(KBE, be_e) LD_IPO(KBSE, bse_e)
	LD_IP(KBSW, aj + ak, bsw_net)
	LD_IP(KBW, ak, bw_et) LD_IP(KBNW, ak, bnw_et)
#undef LD_IP
#undef LD_IPO
#undef LD_IPS
#undef LD_IPSO
#undef LD_IP_
	{
		real_t a_, b_;
		ldpair(qf + row + ie, true, a_, b_); qfe = a_; qfo = b_;
	}
	// ---- q windows: offsets ie-1 .. io+1 of the nine rows
#pragma unroll
	for (int dk = 0; dk < 3; dk++)
#pragma unroll
		for (int dj = 0; dj < 3; dj++) {
			const real_t *r = q + row + (ptrdiff_t)(dj - 1) * (ptrdiff_t)sj + (ptrdiff_t)(dk - 1) * (ptrdiff_t)sk;
			real_t w0, w1, w2, w3;
			ldpair(r + ie - 1, true, w0, w1);
			ldpair(r + io, two, w2, w3);
			qe[dk][dj][0] = w0; qe[dk][dj][1] = w1; qe[dk][dj][2] = w2;
			qo[dk][dj][0] = w1; qo[dk][dj][1] = w2; qo[dk][dj][2] = w3;
		}
}

// Points a launch leaves as they are (already relaxed by the boundary-first chain of this k-parity, dist3.cpp): `skm`
// bits 0..3 = the points at offsets 1, 2, 3, 4 of the row (lane 0: e, o; lane 1: e, o), bits 4..7 = the last four interior
// points (lane P-2: e, o; lane P-1: e, o; P = pairs per row, even extents), bit 8 = the whole row.  A skipped point
// enters every later use (second colour, partial sums, store) with the value it has.
__device__ __forceinline__ void skip27_lane(unsigned skm, int p, int P, bool &ske, bool &sko)
{
	if (skm & 0x100u) { ske = sko = true; return; }
	const int slot = p == 0 ? 0 : p == 1 ? 2 : p == P - 2 ? 4 : p == P - 1 ? 6 : -1;
	ske = slot >= 0 && ((skm >> slot) & 1u);
	sko = slot >= 0 && ((skm >> (slot + 1)) & 1u);
}

// one row task: lane p relaxes the pair (ie, io) = (2p+1, 2p+2) of the row at offset `row`, both
// i-colours, in place.  xch: BS+2 doubles of LDS for the first colour's fresh values.  Every wave of
// the workgroup must call it (one __syncthreads inside).
//   EFIRST = true : colour with even 1-based i (i = 2,4,..) first  (UP order)
//   EFIRST = false: odd i first                                     (DOWN order)
//   PERX: the row is periodic in x with an EVEN number of points: the ghost refresh the reference performs between and
//   after the two colours (q(1) = q(nx+1), q(nx+2) = q(2), BMG3_SymStd_relax_GS.f90:266-269) happens here -- the last
//   point of the second colour takes the fresh first point from LDS, and the two ghost cells are written at the end.
//   MSK / skm: points of the row that were relaxed AHEAD of this launch (boundary-first chain of a rank grid with an
//   x / y split, dist3.cpp) keep their value -- see skip27_lane.
template <int BS, bool EFIRST, bool NT, bool NTP = NT, int WI = 0, bool PERX = false, bool MSK = false>
__device__ __forceinline__ void relax27_row_task(const Op3 &A, const real_t *__restrict__ qf,
                                                 real_t *__restrict__ q, int II, size_t sj, size_t sk,
                                                 size_t j, size_t k, real_t *xch, unsigned skm = 0)
{
	const size_t row = j * sj + k * sk, rowA = j * A.SJ + k * A.SK;
	const int p = threadIdx.x;
	const int ie = 2 * p + 1, io = 2 * p + 2;     // 0-based offsets of the pair in the row
	const bool e_ok = ie <= II - 2;                // interior?
	const bool o_ok = io <= II - 2;
	const bool two = io + 1 <= II - 1;             // element io+1 still inside the row

	real_t e_new = 0.0, o_new = 0.0;
	C27 ce, co;
	real_t qe[3][3][3], qo[3][3][3];
	real_t qfe = 0, qfo = 0, sre = 0, sro = 0;

	if (e_ok) {
		load_pair27<NT, NTP, NT, WI>(A, qf, q, rowA, row, sj, sk, ie, io, two, ce, co, qe, qo, qfe, qfo);
		real_t a_, b_;
		ldpair(A.sor + j * A.rSJ + k * A.rSK + ie, true, a_, b_); sre = a_; sro = b_;
	}

	bool ske = false, sko = false;
	if (MSK) skip27_lane(skm, p, (II - 2 + 1) / 2, ske, sko);
	if (EFIRST) {
		if (e_ok) {
			e_new = (MSK && ske) ? qe[1][1][1] : offdiag27(qfe, ce, qe) * sre;
			xch[p] = e_new;
		}
		__syncthreads();
		if (o_ok) {
			qo[1][1][0] = e_new;
			if (io + 1 <= II - 2) qo[1][1][2] = xch[p + 1]; // next pair's fresh e (else ghost: old value)
			else if (PERX) qo[1][1][2] = xch[0];            // periodic: the ghost was refreshed with the row's first point
			o_new = (MSK && sko) ? qo[1][1][1] : offdiag27(qfo, co, qo) * sro;
		}
		if (PERX && o_ok && io == II - 2) { // the two ghost cells: left = last point, right = first point
			q[row] = o_new;
			q[row + II - 1] = xch[0];
		}
	} else {
		if (o_ok) {
			o_new = (MSK && sko) ? qo[1][1][1] : offdiag27(qfo, co, qo) * sro;
			xch[p + 1] = o_new;
		}
		__syncthreads();
		if (e_ok) {
			if (p > 0) qe[1][1][0] = xch[p]; // previous pair's fresh o (p == 0: ghost column)
			else if (PERX) qe[1][1][0] = xch[(II - 2) / 2]; // periodic: the ghost was refreshed with the row's last point
			if (o_ok) qe[1][1][2] = o_new;
			e_new = (MSK && ske) ? qe[1][1][1] : offdiag27(qfe, ce, qe) * sre;
		}
		if (PERX && p == 0 && e_ok) {
			q[row] = xch[(II - 2) / 2];
			q[row + II - 1] = e_new;
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
}

} // namespace cedar_amd
