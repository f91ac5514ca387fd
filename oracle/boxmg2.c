/* TEST INFRASTRUCTURE ONLY -- see boxmg.h.
 *
 * 2D BoxMG kernels restated in C from the reference's Fortran
 * (src/2d/ftn/ *.f90).  Index macros are 1-based so that every expression can
 * be compared term by term with the cited lines; term order inside each sum is
 * the reference's, and the file is compiled with -ffp-contract=off, so results
 * are expected to be bit-identical to the flang build of the reference.
 */
#include <math.h>
#include <float.h>
#include <string.h>
#include "boxmg.h"

#define F2(a, II, i, j) (a)[(size_t)((i)-1) + (size_t)(II) * (size_t)((j)-1)]
#define S2(a, II, JJ, i, j, s) (a)[(size_t)((i)-1) + (size_t)(II) * ((size_t)((j)-1) + (size_t)(JJ) * (size_t)(s))]

static inline real_t rmax(real_t a, real_t b) { return a > b ? a : b; }
static inline real_t rmin(real_t a, real_t b) { return a < b ? a : b; }

/* src/2d/ftn/BMG2_SymStd_SETUP_recip.f90:63-67 */
void orc2_setup_recip(const real_t *so, real_t *sor, len_t II, len_t JJ)
{
	for (len_t j = 2; j <= JJ - 1; j++)
		for (len_t i = 2; i <= II - 1; i++)
			S2(sor, II, JJ, i, j, 1) = 1.0 / S2(so, II, JJ, i, j, KO); /* msor = 2 */
}

#define SO(i, j, s) S2(so, II, JJ, i, j, s)
#define Q(i, j) F2(q, II, i, j)
#define QF(i, j) F2(qf, II, i, j)

static inline real_t gs9(const real_t *so, const real_t *qf, const real_t *q, const real_t *sor,
                         len_t II, len_t JJ, len_t i, len_t j)
{
	return (QF(i, j)
	        + SO(i, j, KW) * Q(i - 1, j)
	        + SO(i + 1, j, KW) * Q(i + 1, j)
	        + SO(i, j, KS) * Q(i, j - 1)
	        + SO(i, j + 1, KS) * Q(i, j + 1)
	        + SO(i, j, KSW) * Q(i - 1, j - 1)
	        + SO(i + 1, j, KNW) * Q(i + 1, j - 1)
	        + SO(i, j + 1, KNW) * Q(i - 1, j + 1)
	        + SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1))
	       * S2(sor, II, JJ, i, j, 1);
}

static inline real_t gs5(const real_t *so, const real_t *qf, const real_t *q, const real_t *sor,
                         len_t II, len_t JJ, len_t i, len_t j)
{
	return (QF(i, j)
	        + SO(i, j, KW) * Q(i - 1, j)
	        + SO(i + 1, j, KW) * Q(i + 1, j)
	        + SO(i, j, KS) * Q(i, j - 1)
	        + SO(i, j + 1, KS) * Q(i, j + 1))
	       * S2(sor, II, JJ, i, j, 1);
}

/* src/2d/ftn/BMG2_SymStd_relax_GS.f90:76-137 (non-periodic branch).
 * The binding passes irelax_sym = BMG_RELAX_SYM (include/cedar/2d/relax.h:98):
 * DOWN sweeps colours starting at 2 then 3, UP 3 then 2. */
void orc2_relax_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   len_t II, len_t JJ, int ifd, int updown)
{
	int I1 = (int)II - 1, J1 = (int)JJ - 1;
	int lstart, lend, lstride;
	if (updown == BMG_DOWN) { lstart = 2; lend = 3; lstride = 1; }
	else { lstart = 3; lend = 2; lstride = -1; }

	if (ifd != 1) {
		/* 9-point, four colours, row-interleaved loop nest (:93-114) */
		for (int jbeg = lstart; jbeg != lend + lstride; jbeg += lstride)
			for (int j = jbeg; j <= J1; j += 2)
				for (int ibeg = lstart; ibeg != lend + lstride; ibeg += lstride)
					for (int i = ibeg; i <= I1; i += 2)
						Q(i, j) = gs9(so, qf, q, sor, II, JJ, i, j);
	} else {
		/* 5-point red-black (:120-135) */
		for (int jo = lstart; jo != lend + lstride; jo += lstride)
			for (int j = 2; j <= J1; j++)
				for (int i = (j + jo) % 2 + 2; i <= I1; i += 2)
					Q(i, j) = gs5(so, qf, q, sor, II, JJ, i, j);
	}
}

/* one colour of the sweep: what a domain-decomposed run does between two halo exchanges
 * (src/2d/ftn/mpi/BMG2_SymStd_relax_GS.f90).  Nine point: pts = ib + 2 jb, the points with
 * i = 2+ib, 4+ib, .. on the rows j = 2+jb, 4+jb, ..; five point: pts = jo in {2,3}. */
void orc2_relax_colour(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, int ifd, int pts)
{
	int I1 = (int)II - 1, J1 = (int)JJ - 1;
	if (ifd != 1) {
		for (int j = 2 + pts / 2; j <= J1; j += 2)
			for (int i = 2 + pts % 2; i <= I1; i += 2)
				Q(i, j) = gs9(so, qf, q, sor, II, JJ, i, j);
	} else {
		for (int j = 2; j <= J1; j++)
			for (int i = (j + pts) % 2 + 2; i <= I1; i += 2)
				Q(i, j) = gs5(so, qf, q, sor, II, JJ, i, j);
	}
}

/* recompute the nine-point points of 1-based column i on the rows of class jb */
void orc2_relax_column(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, int i, int jb)
{
	int J1 = (int)JJ - 1;
	for (int j = 2 + jb; j <= J1; j += 2)
		Q(i, j) = gs9(so, qf, q, sor, II, JJ, i, j);
}

/* src/2d/ftn/BMG2_SymStd_SETUP_lines_x.f90:68-87 */
void orc2_setup_lines_x(const real_t *so, real_t *sor, len_t II, len_t JJ)
{
	for (len_t j = 2; j <= JJ - 1; j++)
		for (len_t i = 2; i <= II - 1; i++) {
			S2(sor, II, JJ, i, j, 1) = -SO(i, j, KW);
			S2(sor, II, JJ, i, j, 0) = SO(i, j, KO);
		}
	for (len_t j = 2; j <= JJ - 1; j++)
		orc_dpttrf((int)II - 2, &S2(sor, II, JJ, 2, j, 0), &S2(sor, II, JJ, 3, j, 1));
}

/* src/2d/ftn/BMG2_SymStd_SETUP_lines_y.f90:69-87 -- SOR is transposed: SOR(JJ,II,2) */
void orc2_setup_lines_y(const real_t *so, real_t *sor, len_t II, len_t JJ)
{
#define SORT(j, i, s) (sor)[(size_t)((j)-1) + (size_t)JJ * ((size_t)((i)-1) + (size_t)II * (size_t)(s))]
	for (len_t i = 2; i <= II - 1; i++)
		for (len_t j = 2; j <= JJ - 1; j++) {
			SORT(j, i, 1) = -SO(i, j, KS);
			SORT(j, i, 0) = SO(i, j, KO);
		}
	for (len_t i = 2; i <= II - 1; i++)
		orc_dpttrf((int)JJ - 2, &SORT(2, i, 0), &SORT(3, i, 1));
}

/* src/2d/ftn/BMG2_SymStd_relax_lines_x.f90:82-162: DOWN relaxes lines J=3,5,.. then 2,4,.. */
void orc2_relax_lines_x(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                        len_t II, len_t JJ, int ifd, int updown)
{
	int I1 = (int)II - 1, J1 = (int)JJ - 1;
	int jstart, jend, jstride;
	if (updown == BMG_DOWN) { jstart = 3; jend = 2; jstride = -1; }
	else { jstart = 2; jend = 3; jstride = 1; }

	for (int jbeg = jstart; jbeg != jend + jstride; jbeg += jstride) {
		if (ifd != 1) {
			for (int j = jbeg; j <= J1; j += 2)
				for (int i = 2; i <= I1; i++)
					Q(i, j) = QF(i, j) + SO(i, j, KS) * Q(i, j - 1) + SO(i, j + 1, KS)
					          * Q(i, j + 1) + SO(i, j, KSW) * Q(i - 1, j - 1) + SO(i + 1, j, KNW)
					          * Q(i + 1, j - 1) + SO(i, j + 1, KNW) * Q(i - 1, j + 1)
					          + SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1);
			for (int j = jbeg; j <= J1; j += 2)
				orc_dpttrs(I1 - 1, &S2(sor, II, JJ, 2, j, 0), &S2(sor, II, JJ, 3, j, 1), &Q(2, j));
		} else {
			for (int j = jbeg; j <= J1; j += 2) {
				for (int i = 2; i <= I1; i++)
					Q(i, j) = QF(i, j) + SO(i, j, KS) * Q(i, j - 1) + SO(i, j + 1, KS)
					          * Q(i, j + 1);
				orc_dpttrs(I1 - 1, &S2(sor, II, JJ, 2, j, 0), &S2(sor, II, JJ, 3, j, 1), &Q(2, j));
			}
		}
	}
}

/* src/2d/ftn/BMG2_SymStd_relax_lines_y.f90:77-168; B is scratch of length >= 2*JJ */
void orc2_relax_lines_y(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                        real_t *b, len_t II, len_t JJ, int ifd, int updown)
{
	int I1 = (int)II - 1, J1 = (int)JJ - 1;
	int istart, iend, istride;
	if (updown == BMG_DOWN) { istart = 3; iend = 2; istride = -1; }
	else { istart = 2; iend = 3; istride = 1; }

	for (int ibeg = istart; ibeg != iend + istride; ibeg += istride) {
		if (ifd != 1) {
			for (int i = ibeg; i <= I1; i += 2) {
				for (int j = 2; j <= J1; j++)
					b[j - 1] = QF(i, j) + SO(i, j, KW) * Q(i - 1, j) + SO(i + 1, j, KW)
					           * Q(i + 1, j) + SO(i, j, KSW) * Q(i - 1, j - 1) + SO(i + 1, j, KNW)
					           * Q(i + 1, j - 1) + SO(i, j + 1, KNW) * Q(i - 1, j + 1)
					           + SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1);
				orc_dpttrs(J1 - 1, &SORT(2, i, 0), &SORT(3, i, 1), &b[1]);
				for (int j = 2; j <= J1; j++)
					Q(i, j) = b[j - 1];
			}
		} else {
			for (int j = 2; j <= J1; j++)
				for (int i = ibeg; i <= I1; i += 2)
					Q(i, j) = QF(i, j) + SO(i, j, KW) * Q(i - 1, j) + SO(i + 1, j, KW)
					          * Q(i + 1, j);
			for (int i = ibeg; i <= I1; i += 2) {
				for (int j = 2; j <= J1; j++)
					b[j - 1] = Q(i, j);
				orc_dpttrs(J1 - 1, &SORT(2, i, 0), &SORT(3, i, 1), &b[1]);
				for (int j = 2; j <= J1; j++)
					Q(i, j) = b[j - 1];
			}
		}
	}
}
#undef SORT

/* src/2d/ftn/BMG2_SymStd_residual.f90:85-119 */
void orc2_residual(const real_t *so, const real_t *qf, const real_t *q, real_t *res,
                   len_t II, len_t JJ, int ifd)
{
	for (len_t j = 2; j <= JJ - 1; j++)
		for (len_t i = 2; i <= II - 1; i++) {
			if (ifd != 1)
				F2(res, II, i, j) = QF(i, j)
				                    + SO(i, j, KW) * Q(i - 1, j)
				                    + SO(i + 1, j, KW) * Q(i + 1, j)
				                    + SO(i, j, KS) * Q(i, j - 1)
				                    + SO(i, j + 1, KS) * Q(i, j + 1)
				                    + SO(i, j, KSW) * Q(i - 1, j - 1)
				                    + SO(i + 1, j, KNW) * Q(i + 1, j - 1)
				                    + SO(i, j + 1, KNW) * Q(i - 1, j + 1)
				                    + SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1)
				                    - SO(i, j, KO) * Q(i, j);
			else
				F2(res, II, i, j) = QF(i, j)
				                    + SO(i, j, KW) * Q(i - 1, j)
				                    + SO(i + 1, j, KW) * Q(i + 1, j)
				                    + SO(i, j, KS) * Q(i, j - 1)
				                    + SO(i, j + 1, KS) * Q(i, j + 1)
				                    - SO(i, j, KO) * Q(i, j);
		}
}

/* qf = A q: src/2d/ftn/mpi/BMG2_SymStd_UTILS_matvec.f90:84-118, on the serial array layout
 * (the MPI flavour's SO carries one more ghost; the arithmetic per point is the same).
 * Parity: the MPI Fortran is not part of oracle/_ref (needs mpif.h/MSG); pinned through the
 * identity matvec(q) + residual(0, q) = 0 with the _ref-pinned residual and against a dense
 * assembly of the operator (tests/test_bmg_capi.py). */
void orc2_matvec(const real_t *so, const real_t *q, real_t *qf, len_t II, len_t JJ, int ifd)
{
#define QO(i, j) F2(qf, II, i, j)
	for (len_t j = 2; j <= JJ - 1; j++)
		for (len_t i = 2; i <= II - 1; i++) {
			if (ifd != 1)
				QO(i, j) = SO(i, j, KO) * Q(i, j)
				           - SO(i, j, KW) * Q(i - 1, j)
				           - SO(i + 1, j, KW) * Q(i + 1, j)
				           - SO(i, j, KS) * Q(i, j - 1)
				           - SO(i, j + 1, KS) * Q(i, j + 1)
				           - SO(i, j, KSW) * Q(i - 1, j - 1)
				           - SO(i + 1, j, KNW) * Q(i + 1, j - 1)
				           - SO(i, j + 1, KNW) * Q(i - 1, j + 1)
				           - SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1);
			else
				QO(i, j) = SO(i, j, KO) * Q(i, j)
				           - SO(i, j, KW) * Q(i - 1, j)
				           - SO(i + 1, j, KW) * Q(i + 1, j)
				           - SO(i, j, KS) * Q(i, j - 1)
				           - SO(i, j + 1, KS) * Q(i, j + 1);
		}
#undef QO
}
#undef Q
#undef QF
#undef SO

#define CI(ic, jc, s) S2(ci, IIC, JJC, ic, jc, s)
#define QC(ic, jc) F2(qc, IIC, ic, jc)

/* src/2d/ftn/BMG2_SymStd_restrict.f90:73-92 */
void orc2_restrict(const real_t *q, real_t *qc, const real_t *ci,
                   len_t II, len_t JJ, len_t IIC, len_t JJC)
{
	(void)JJ;
#define Q(i, j) F2(q, II, i, j)
	for (len_t jc = 2; jc <= JJC - 1; jc++) {
		len_t j = 2 * (jc - 1);
		for (len_t ic = 2; ic <= IIC - 1; ic++) {
			len_t i = 2 * (ic - 1);
			QC(ic, jc) = CI(ic, jc, LNE) * Q(i - 1, j - 1)
			             + CI(ic, jc, LA) * Q(i, j - 1)
			             + CI(ic + 1, jc, LNW) * Q(i + 1, j - 1)
			             + CI(ic, jc, LR) * Q(i - 1, j)
			             + Q(i, j)
			             + CI(ic + 1, jc, LL) * Q(i + 1, j)
			             + CI(ic, jc + 1, LSE) * Q(i - 1, j + 1)
			             + CI(ic, jc + 1, LB) * Q(i, j + 1)
			             + CI(ic + 1, jc + 1, LSW) * Q(i + 1, j + 1);
		}
	}
#undef Q
}

/* src/2d/ftn/BMG2_SymStd_interp_add.f90:88-137.  NB: RES is divided by the
 * diagonal in place first (:101-105); for even nx the loops reach the coarse
 * ghost column IC = IIC and write the fine ghost Q(IIF,.) exactly like the
 * reference does (IICF1 = (IIF-2)/2+2). */
void orc2_interp_add(real_t *q, const real_t *qc, real_t *res, const real_t *so,
                     const real_t *ci, len_t IIC, len_t JJC, len_t IIF, len_t JJF)
{
#define Q(i, j) F2(q, IIF, i, j)
#define RES(i, j) F2(res, IIF, i, j)
	int IICF1 = ((int)IIF - 2) / 2 + 2, JJCF1 = ((int)JJF - 2) / 2 + 2;
	real_t a, aq;

	for (len_t j = 2; j <= JJF - 1; j++)
		for (len_t i = 2; i <= IIF - 1; i++)
			RES(i, j) = RES(i, j) / S2(so, IIF, JJF, i, j, KO);

	int j = 2, i = 2;
	Q(2, j) = Q(2, j) + QC(2, 2);
	for (int ic = 3; ic <= IICF1; ic++) {
		i += 2;
		Q(i, j) = Q(i, j) + QC(ic, 2);
		a = CI(ic, 2, LR) * QC(ic, 2) + CI(ic, 2, LL) * QC(ic - 1, 2);
		Q(i - 1, j) = Q(i - 1, j) + a + RES(i - 1, j);
	}
	for (int jc = 3; jc <= JJCF1; jc++) {
		j += 2;
		i = 2;
		Q(2, j) = Q(2, j) + QC(2, jc);
		aq = CI(2, jc, LA) * QC(2, jc) + CI(2, jc, LB) * QC(2, jc - 1);
		Q(2, j - 1) = Q(2, j - 1) + aq + RES(2, j - 1);
		for (int ic = 3; ic <= IICF1; ic++) {
			i += 2;
			Q(i, j) = Q(i, j) + QC(ic, jc);
			a = CI(ic, jc, LR) * QC(ic, jc) + CI(ic, jc, LL) * QC(ic - 1, jc);
			Q(i - 1, j) = Q(i - 1, j) + a + RES(i - 1, j);
			aq = CI(ic, jc, LA) * QC(ic, jc) + CI(ic, jc, LB) * QC(ic, jc - 1);
			Q(i, j - 1) = Q(i, j - 1) + aq + RES(i, j - 1);
			a = CI(ic, jc, LSW) * QC(ic - 1, jc - 1) + CI(ic, jc, LNW) * QC(ic - 1, jc)
			    + CI(ic, jc, LNE) * QC(ic, jc) + CI(ic, jc, LSE) * QC(ic, jc - 1);
			Q(i - 1, j - 1) = Q(i - 1, j - 1) + a + RES(i - 1, j - 1);
		}
	}
#undef Q
#undef RES
}

/* The lumping switch shared by every phase of the interpolation set-up:
 *   S <- off + (diag - S) * max(diag - (1+ep) S, 0) / (|diag - (1+ep) S| + eps)
 * src/2d/ftn/BMG2_SymStd_SETUP_interp_OI.f90:122-125 */
static inline real_t lump(real_t off, real_t diag, real_t s, real_t ep, real_t zeps)
{
	return off + (diag - s) * rmax(diag - (1.0 + ep) * s, 0.0) / (fabs(diag - (1.0 + ep) * s) + zeps);
}

/* ---- operator-induced interpolation: the three point formulas of BMG2_SymStd_SETUP_interp_OI.f90,
 * shared by the non-periodic (:84-256) and the periodic (:258-618) index drivers.  (i,j) = fine
 * point the formula is centred on, (ic,jc) = coarse entry that receives the weights. */
#define SO(i, j, s) S2(so, IIF, JJF, i, j, s)
#define CIW(ic, jc, s) S2(ci, IIC, JJC, ic, jc, s)
static void ci_xedge(const real_t *so, real_t *ci, len_t IIF, len_t JJF, len_t IIC, len_t JJC,
                     int ifd, int i, int j, int ic, int jc)
{
	const real_t zeps = DBL_EPSILON; /* EPSILON(1.D0), :70 */
	real_t a, b, ep, sum;
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

static void ci_yedge(const real_t *so, real_t *ci, len_t IIF, len_t JJF, len_t IIC, len_t JJC,
                     int ifd, int i, int j, int ic, int jc)
{
	const real_t zeps = DBL_EPSILON;
	real_t a, b, ep, sum;
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

static void ci_centre(const real_t *so, real_t *ci, len_t IIF, len_t JJF, len_t IIC, len_t JJC,
                      int ifd, int i, int j, int ic, int jc)
{
	const real_t zeps = DBL_EPSILON;
	real_t ep, sum, s;
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
#undef CIW

/* src/2d/ftn/BMG2_SymStd_SETUP_interp_OI.f90:84-256 (non-periodic).  phase_mask: 1 = the two edge
 * families, 2 = the cell centres; ilo / jlo: first coarse index of the "between two coarse points"
 * loops, 3 in the serial code, 2 on a side where a neighbouring subdomain owns coarse index 1 (the
 * domain-decomposed driver, as in the 3D variant). */
void orc2_setup_interp_ex(const real_t *so, real_t *ci, len_t IIF, len_t JJF,
                          len_t IIC, len_t JJC, int ifd, int phase_mask, int ilo, int jlo)
{
	int IIC1 = (int)IIC - 1, JJC1 = (int)JJC - 1;
	int IICF1 = ((int)IIF - 2) / 2 + 2, JJCF1 = ((int)JJF - 2) / 2 + 2;

	if (phase_mask & 1) {
		/* x-edges: fine points between two coarse points on a coarse row (:112-130 / :196-213) */
		for (int jc = 2; jc <= JJC1; jc++)
			for (int ic = ilo; ic <= IICF1; ic++)
				ci_xedge(so, ci, IIF, JJF, IIC, JJC, ifd, 2 * (ic - 1), 2 * (jc - 1), ic, jc);
		/* y-edges (:131-149 / :214-231) */
		for (int jc = jlo; jc <= JJCF1; jc++)
			for (int ic = 2; ic <= IIC1; ic++)
				ci_yedge(so, ci, IIF, JJF, IIC, JJC, ifd, 2 * (ic - 1), 2 * (jc - 1), ic, jc);
	}
	/* cell centres (:150-188 / :232-255) */
	if (phase_mask & 2)
		for (int jc = jlo; jc <= JJCF1; jc++)
			for (int ic = ilo; ic <= IICF1; ic++)
				ci_centre(so, ci, IIF, JJF, IIC, JJC, ifd, 2 * (ic - 1), 2 * (jc - 1), ic, jc);
}

void orc2_setup_interp(const real_t *so, real_t *ci, len_t IIF, len_t JJF,
                       len_t IIC, len_t JJC, int ifd)
{
	orc2_setup_interp_ex(so, ci, IIF, JJF, IIC, JJC, ifd, 3, 3, 3);
}

/* the fine index the periodic driver pairs with the next coarse index: I <- MAX(MOD(I+2,IIFC), MIN(I+2,3))
 * (:327, :378 ...): steps by two and wraps past the last fine point onto index 3 */
static int per_next(int i, int nfc)
{
	int a = (i + 2) % nfc, b = i + 2 < 3 ? i + 2 : 3;
	return a > b ? a : b;
}

/* src/2d/ftn/BMG2_SymStd_SETUP_interp_OI.f90:258-618: periodic in x (ipn 2), y (1) or both (3) */
void orc2_setup_interp_per(const real_t *so, real_t *ci, len_t IIF, len_t JJF,
                           len_t IIC, len_t JJC, int ifd, int ipn)
{
	const int per_x = ipn == 2 || ipn == 3, per_y = ipn == 1 || ipn == 3;
	const int IIC1 = (int)IIC - 1, JJC1 = (int)JJC - 1;
	const int IIF1 = (int)IIF - 1, IIF2 = (int)IIF - 2, JJF1 = (int)JJF - 1, JJF2 = (int)JJF - 2;
	const int IICF = ((int)IIF - 2) / 2 + 3, JJCF = ((int)JJF - 2) / 2 + 3;
	const int IICF1 = IICF - 1, JJCF1 = JJCF - 1;
	const int IIFC = 2 * (IICF - 2) + 2, JJFC = 2 * (JJCF - 2) + 2;
	int IBEG = 2, IBEGC = 3, IBEG_x = 2, IEND_x = IIF2, IENDC = IICF1, JENDC = JJCF1;
	int JBEG = 2, JBEGC = 3, JBEG_y = 2, JEND_y = JJF2;
	int i, j;
	if (per_x) { /* :294-302 */
		IBEG = 0; IBEGC = 2;
		if ((int)IIC == IICF) { IENDC = IICF; IBEG_x = 3; IEND_x = IIF1; }
	}
	if (per_y) { /* :307-315 */
		JBEG = 0; JBEGC = 2;
		if ((int)JJC == JJCF) { JENDC = JJCF; JBEG_y = 3; JEND_y = JJF1; }
	}
	/* x-edges on the coarse rows (:322-341 / :475-492) */
	j = 0;
	for (int jc = 2; jc <= JJC1; jc++) {
		j += 2;
		i = IBEG;
		for (int ic = IBEGC; ic <= IENDC; ic++) {
			i = per_next(i, IIFC);
			ci_xedge(so, ci, IIF, JJF, IIC, JJC, ifd, i, j, ic, jc);
		}
	}
	if (per_y) { /* the two wrapped coarse rows (:343-375 / :495-526) */
		i = IBEG;
		for (int ic = IBEGC; ic <= IENDC; ic++) {
			i = per_next(i, IIFC);
			ci_xedge(so, ci, IIF, JJF, IIC, JJC, ifd, i, JBEG_y, ic, (int)JJC);
		}
		i = IBEG;
		for (int ic = IBEGC; ic <= IENDC; ic++) {
			i = per_next(i, IIFC);
			ci_xedge(so, ci, IIF, JJF, IIC, JJC, ifd, i, JEND_y, ic, 1);
		}
	}
	/* y-edges on the coarse columns (:377-394 / :529-546) */
	j = JBEG;
	for (int jc = JBEGC; jc <= JENDC; jc++) {
		j = per_next(j, JJFC);
		i = 0;
		for (int ic = 2; ic <= IIC1; ic++) {
			i += 2;
			ci_yedge(so, ci, IIF, JJF, IIC, JJC, ifd, i, j, ic, jc);
		}
	}
	if (per_x) { /* the two wrapped coarse columns (:396-428 / :548-579) */
		j = JBEG;
		for (int jc = JBEGC; jc <= JENDC; jc++) {
			j = per_next(j, JJFC);
			ci_yedge(so, ci, IIF, JJF, IIC, JJC, ifd, IBEG_x, j, (int)IIC, jc);
		}
		j = JBEG;
		for (int jc = JBEGC; jc <= JENDC; jc++) {
			j = per_next(j, JJFC);
			ci_yedge(so, ci, IIF, JJF, IIC, JJC, ifd, IEND_x, j, 1, jc);
		}
	}
	/* cell centres (:431-466 / :581-608) */
	j = JBEG;
	for (int jc = JBEGC; jc <= JENDC; jc++) {
		j = per_next(j, JJFC);
		i = IBEG;
		for (int ic = IBEGC; ic <= IENDC; ic++) {
			i = per_next(i, IIFC);
			ci_centre(so, ci, IIF, JJF, IIC, JJC, ifd, i, j, ic, jc);
		}
	}
	(void)IIF1; (void)JJF1;
}

/* Galerkin coarse operator A_c = P^T A P.
 * src/2d/ftn/BMG2_SymStd_SETUP_ITLI_ex.f90:94-214 (9-pt fine), :252-331 (5-pt fine).
 *
 * The reference writes the triple product as closed-form sums per coarse slot.
 * This restatement evaluates the same triple product generically: with
 * F(c) = 2(c-1) the fine index of coarse point c,
 *   P(F(c)+d, c) = 1 (d = 0) or the CI entry that interp_add applies to that
 *   fine point (src/2d/ftn/BMG2_SymStd_interp_add.f90:108-136), and
 *   A(f,f) = SO(f,KO), A(f,g) = -SO(.,slot) for the 8 neighbours,
 *   SOC(c,KO) = (P^T A P)(c,c), SOC(c,slot) = -(P^T A P)(c, c - delta_slot).
 * The set of (SO, CI) entries read is exactly the reference's (ghost entries
 * included); only the order of the additions differs, so agreement with the
 * reference is to rounding (checked <= 1e-13 relative against the goldens),
 * not bit-for-bit. */
static inline real_t pw2(const real_t *ci, len_t IIC, len_t JJC, int ic, int jc, int dx, int dy)
{
	/* weight with which coarse (ic,jc) contributes to fine F(ic,jc)+(dx,dy) */
	if (dx == 0 && dy == 0) return 1.0;
	if (dy == 0) return dx < 0 ? CI(ic, jc, LR) : CI(ic + 1, jc, LL);
	if (dx == 0) return dy < 0 ? CI(ic, jc, LA) : CI(ic, jc + 1, LB);
	if (dx < 0 && dy < 0) return CI(ic, jc, LNE);
	if (dx > 0 && dy < 0) return CI(ic + 1, jc, LNW);
	if (dx < 0 && dy > 0) return CI(ic, jc + 1, LSE);
	return CI(ic + 1, jc + 1, LSW);
}

/* stored (positive) coupling between fine (i,j) and (i+dx,j+dy), (dx,dy) != 0.
 * Returns 0 when the storage location lies outside the array (the reference's
 * closed forms never form such a term). */
static inline real_t aoff2(const real_t *so, len_t IIF, len_t JJF, int ifd, int i, int j, int dx, int dy)
{
	int si, sj, slot;
	if (dy == 0) { slot = KW; si = dx < 0 ? i : i + 1; sj = j; }
	else if (dx == 0) { slot = KS; si = i; sj = dy < 0 ? j : j + 1; }
	else if (dx < 0 && dy < 0) { slot = KSW; si = i; sj = j; }
	else if (dx > 0 && dy > 0) { slot = KSW; si = i + 1; sj = j + 1; }
	else if (dx > 0 && dy < 0) { slot = KNW; si = i + 1; sj = j; }
	else { slot = KNW; si = i; sj = j + 1; }
	if (ifd == 1 && slot >= KSW) return 0.0;
	if (si < 1 || si > (int)IIF || sj < 1 || sj > (int)JJF) return 0.0;
	return SO(si, sj, slot);
}

void orc2_galerkin(const real_t *so, real_t *soc, const real_t *ci, len_t IIF, len_t JJF,
                   len_t IIC, len_t JJC, int ifd)
{
	/* slot s of SOC at coarse P couples coarse P+a_s and P+b_s
	 * (KNW couples P-(0,1) with P-(1,0), src/2d/ftn/BMG2_SymStd_relax_GS.f90:104-105) */
	static const int ax[5] = { 0, 0, 0, 0, 0 }, ay[5] = { 0, 0, 0, 0, -1 };
	static const int bx[5] = { 0, -1, 0, -1, -1 }, by[5] = { 0, 0, -1, -1, 0 };
	for (int jc = 2; jc <= (int)JJC - 1; jc++) {
		for (int ic = 2; ic <= (int)IIC - 1; ic++) {
			for (int s = 0; s < 5; s++) {
				int ic1 = ic + ax[s], jc1 = jc + ay[s];
				int ic2 = ic + bx[s], jc2 = jc + by[s];
				int i1 = 2 * (ic1 - 1), j1 = 2 * (jc1 - 1);
				int i2 = 2 * (ic2 - 1), j2 = 2 * (jc2 - 1);
				real_t acc = 0.0;
				for (int dy = -1; dy <= 1; dy++)
					for (int dx = -1; dx <= 1; dx++) {
						int fi = i1 + dx, fj = j1 + dy; /* f1 in N(c1) */
						real_t row = 0.0;              /* (A P)(f1, c2) */
						int any = 0;
						for (int ey = -1; ey <= 1; ey++)
							for (int ex = -1; ex <= 1; ex++) {
								int rx = fi + ex - i2, ry = fj + ey - j2;
								if (rx < -1 || rx > 1 || ry < -1 || ry > 1) continue;
								real_t p2 = pw2(ci, IIC, JJC, ic2, jc2, rx, ry);
								if (ex == 0 && ey == 0) {
									if (fi < 1 || fi > (int)IIF || fj < 1 || fj > (int)JJF) continue;
									row += SO(fi, fj, KO) * p2;
								} else
									row -= aoff2(so, IIF, JJF, ifd, fi, fj, ex, ey) * p2;
								any = 1;
							}
						if (any)
							acc += pw2(ci, IIC, JJC, ic1, jc1, dx, dy) * row;
					}
				S2(soc, IIC, JJC, ic, jc, s) = (s == KO) ? acc : -acc;
			}
		}
	}
}
#undef SO

/* src/2d/ftn/BMG2_SymStd_SETUP_cg_LU.f90:92-118 (NStncl = 5) and :236-258 (NStncl = 3):
 * pack the coarsest operator into LAPACK upper band storage, then DPBTRF('U'). */
int orc2_setup_cg(const real_t *so, len_t II, len_t JJ, int nstncl,
                  real_t *abd, len_t nabd1, len_t nabd2)
{
#define ABD(r, c) abd[(size_t)((r)-1) + (size_t)nabd1 * (size_t)((c)-1)]
#define SO(i, j, s) S2(so, II, JJ, i, j, s)
	int I1 = (int)II - 1, J1 = (int)JJ - 1, I2 = I1 - 1;
	int n = I2 * (J1 - 1), kk = 0;
	(void)nabd2;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++) {
			kk++;
			ABD(II, kk) = SO(i, j, KO);
			ABD(I1, kk) = -SO(i, j, KW);
			ABD(3, kk) = (nstncl == 5) ? -SO(i + 1, j, KNW) : 0.0;
			ABD(2, kk) = -SO(i, j, KS);
			ABD(1, kk) = (nstncl == 5) ? -SO(i, j, KSW) : 0.0;
		}
	return orc_dpbtrf_upper(n, I1, abd, (int)nabd1);
#undef SO
}

/* src/2d/ftn/BMG2_SymStd_SOLVE_cg.f90:95-125 */
int orc2_solve_cg(real_t *q, const real_t *qf, len_t II, len_t JJ,
                  const real_t *abd, real_t *bbd, len_t nabd1, len_t nabd2)
{
	int I1 = (int)II - 1, J1 = (int)JJ - 1, kk = 0;
	(void)nabd2;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++)
			bbd[kk++] = F2(qf, II, i, j);
	orc_dpbtrs_upper(kk, I1, abd, (int)nabd1, bbd);
	kk = 0;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++)
			F2(q, II, i, j) = bbd[kk++];
	return 0;
#undef ABD
}

/* include/cedar/2d/grid_func.h:42-53: lp_norm<2> = pow(sum pow(v,2), 1/2), interior, i fastest */
real_t orc_l2_norm2(const real_t *v, len_t II, len_t JJ)
{
	real_t result = 0;
	for (len_t j = 2; j <= JJ - 1; j++)
		for (len_t i = 2; i <= II - 1; i++)
			result += F2(v, II, i, j) * F2(v, II, i, j);
	return sqrt(result);
}
