/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's 2D periodic branches
 * (ibc = BMG_BCs_def_per_y 1, _per_x 2, _per_xy 3; include/cedar/2d/ftn/BMG_parameters_c.h:193-196),
 * point and line relaxation.  Each function cites the Fortran it follows; pinned bit for bit against
 * oracle/_ref (tests/test_oracle_periodic.py), except the dense Cholesky (vendor LAPACK vs the
 * unblocked netlib order restated in lapack_mini.c: rounding-level differences).
 * The interpolation set-up lives in boxmg2.c (orc2_setup_interp_per) next to the formulas it shares
 * with the non-periodic driver. */
#include "boxmg.h"
#include <math.h>
#include <string.h>

/* 1-based Fortran indexing, as in boxmg2.c */
#define F2(a, II, i, j) (a)[(size_t)((i)-1) + (size_t)(II) * (size_t)((j)-1)]
#define S2(a, II, JJ, i, j, s) (a)[(size_t)((i)-1) + (size_t)(II) * ((size_t)((j)-1) + (size_t)(JJ) * (size_t)(s))]

#define PER_X(ipn) ((ipn) == 2 || (ipn) == 3)
#define PER_Y(ipn) ((ipn) == 1 || (ipn) == 3)

/* Q(I,1)=Q(I,J1); Q(I,JJ)=Q(I,2), I = 1..II */
static void wrap_y(real_t *q, len_t II, len_t JJ)
{
	for (len_t i = 1; i <= II; i++) {
		F2(q, II, i, 1) = F2(q, II, i, JJ - 1);
		F2(q, II, i, JJ) = F2(q, II, i, 2);
	}
}

/* Q(1,J)=Q(I1,J); Q(II,J)=Q(2,J), J = 1..JJ */
static void wrap_x(real_t *q, len_t II, len_t JJ)
{
	for (len_t j = 1; j <= JJ; j++) {
		F2(q, II, 1, j) = F2(q, II, II - 1, j);
		F2(q, II, II, j) = F2(q, II, 2, j);
	}
}

/* src/2d/ftn/BMG2_SymStd_relax_GS.f90:139-226.  Note the 9-point order: row by row, both i-colours
 * of a row back to back with the x wrap after each; the y wrap only once, after the sweep. */
void orc2_relax_gs_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, int ifd, int updown, int ipn)
{
#define SO(i, j, s) S2(so, II, JJ, i, j, s)
#define Q(i, j) F2(q, II, i, j)
#define QF(i, j) F2(qf, II, i, j)
	const int I1 = (int)II - 1, J1 = (int)JJ - 1;
	const int lstart = updown == 0 ? 2 : 3, lend = updown == 0 ? 3 : 2, lstride = updown == 0 ? 1 : -1;
	if (ifd != 1) {
		for (int jbeg = lstart; jbeg != lend + lstride; jbeg += lstride) {
			int jend = 2 * ((J1 - jbeg) / 2) + jbeg;
			for (int j = jbeg; j <= jend; j += 2)
				for (int ibeg = lstart; ibeg != lend + lstride; ibeg += lstride) {
					int iend = 2 * ((I1 - ibeg) / 2) + ibeg;
					for (int i = ibeg; i <= iend; i += 2)
						Q(i, j) = (QF(i, j)
						           + SO(i, j, KW) * Q(i - 1, j)
						           + SO(i + 1, j, KW) * Q(i + 1, j)
						           + SO(i, j, KS) * Q(i, j - 1)
						           + SO(i, j + 1, KS) * Q(i, j + 1)
						           + SO(i, j, KSW) * Q(i - 1, j - 1)
						           + SO(i + 1, j, KNW) * Q(i + 1, j - 1)
						           + SO(i, j + 1, KNW) * Q(i - 1, j + 1)
						           + SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1))
						          * S2(sor, II, JJ, i, j, 1); /* msor = 2 */
					if (PER_X(ipn)) {
						Q(1, j) = Q(I1, j);
						Q(II, j) = Q(2, j);
					}
				}
		}
	} else {
		for (int jo = lstart; jo != lend + lstride; jo += lstride)
			for (int j = 2; j <= J1; j++) {
				int ibeg = (j + jo) % 2 + 2;
				int iend = 2 * ((I1 - ibeg) / 2) + ibeg;
				for (int i = ibeg; i <= iend; i += 2)
					Q(i, j) = (QF(i, j)
					           + SO(i, j, KW) * Q(i - 1, j)
					           + SO(i + 1, j, KW) * Q(i + 1, j)
					           + SO(i, j, KS) * Q(i, j - 1)
					           + SO(i, j + 1, KS) * Q(i, j + 1))
					          * S2(sor, II, JJ, i, j, 1); /* msor = 2 */
				if (PER_X(ipn)) {
					Q(1, j) = Q(I1, j);
					Q(II, j) = Q(2, j);
				}
			}
	}
	if (PER_Y(ipn)) wrap_y(q, II, JJ);
#undef SO
#undef Q
#undef QF
}

/* src/2d/ftn/BMG2_SymStd_restrict.f90:94-128: the fine vector gets its periodic ghosts refreshed
 * (only when the fine extent is even: Ny/2+1 == Nyc) before the ordinary restriction */
void orc2_restrict_per(real_t *q, real_t *qc, const real_t *ci,
                       len_t II, len_t JJ, len_t IIC, len_t JJC, int ipn)
{
	if (PER_Y(ipn) && JJ / 2 + 1 == JJC)
		for (len_t i = 1; i <= II; i++) {
			F2(q, II, i, 1) = F2(q, II, i, JJ - 1);
			F2(q, II, i, JJ) = F2(q, II, i, 2);
		}
	if (PER_X(ipn) && II / 2 + 1 == IIC)
		for (len_t j = 1; j <= JJ; j++) {
			F2(q, II, 1, j) = F2(q, II, II - 1, j);
			F2(q, II, II, j) = F2(q, II, 2, j);
		}
	orc2_restrict(q, qc, ci, II, JJ, IIC, JJC);
}

/* src/2d/ftn/BMG2_SymStd_interp_add.f90:139-156: ordinary interpolate-and-add, then y and x wraps */
void orc2_interp_add_per(real_t *q, const real_t *qc, real_t *res, const real_t *so,
                         const real_t *ci, len_t IIC, len_t JJC, len_t IIF, len_t JJF, int ipn)
{
	orc2_interp_add(q, qc, res, so, ci, IIC, JJC, IIF, JJF);
	if (PER_Y(ipn)) wrap_y(q, IIF, JJF);
	if (PER_X(ipn)) wrap_x(q, IIF, JJF);
}

/* src/2d/ftn/BMG2_SymStd_SETUP_ITLI_ex.f90:216-247 / :333-364: the five coarse planes get their
 * periodic ghosts after the ordinary Galerkin product */
void orc2_galerkin_per(const real_t *so, real_t *soc, const real_t *ci, len_t IIF, len_t JJF,
                       len_t IIC, len_t JJC, int ifd, int ipn)
{
	orc2_galerkin(so, soc, ci, IIF, JJF, IIC, JJC, ifd);
	const size_t P = (size_t)IIC * JJC;
	if (PER_Y(ipn))
		for (int s = 0; s < 5; s++) {
			real_t *pl = soc + (size_t)s * P;
			for (len_t ic = 1; ic <= IIC; ic++) {
				F2(pl, IIC, ic, JJC) = F2(pl, IIC, ic, 2);
				F2(pl, IIC, ic, 1) = F2(pl, IIC, ic, JJC - 1);
			}
		}
	if (PER_X(ipn))
		for (int s = 0; s < 5; s++) {
			real_t *pl = soc + (size_t)s * P;
			for (len_t jc = 1; jc <= JJC; jc++) {
				F2(pl, IIC, IIC, jc) = F2(pl, IIC, 2, jc);
				F2(pl, IIC, 1, jc) = F2(pl, IIC, IIC - 1, jc);
			}
		}
}

/* src/2d/ftn/BMG2_SymStd_SETUP_cg_LU.f90:148-218 (nine point) / :262-330 (five point): the coarsest
 * operator as a dense symmetric matrix (upper triangle, ABD(nabd1, n) with nabd1 >= n), DPOTRF.
 * ibc > 0 (definite) only: the C++ layer never passes the indefinite codes (BMG_get_bc.f90:13-20). */
int orc2_setup_cg_per(const real_t *so, len_t II, len_t JJ, int nstncl, real_t *abd, len_t nabd1, int ipn)
{
#define ABD(r, c) abd[(size_t)((r)-1) + (size_t)nabd1 * (size_t)((c)-1)]
#define SO(i, j, s) S2(so, II, JJ, i, j, s)
	const int I1 = (int)II - 1, J1 = (int)JJ - 1, I2 = I1 - 1;
	const int n = I2 * (J1 - 1);
	const int nine = nstncl == 5;
	int kk = 1;
	ABD(1, 1) = SO(2, 2, KO);
	for (int i = 3; i <= I1; i++) {
		kk++;
		ABD(kk, kk) = SO(i, 2, KO);
		ABD(kk - 1, kk) = -SO(i, 2, KW);
	}
	if (PER_X(ipn)) ABD(kk - I2 + 1, kk) = -SO(II, 2, KW);
	for (int j = 3; j <= J1; j++) {
		if (PER_X(ipn)) ABD(kk, kk + 1) = nine ? -SO(2, j, KSW) : 0.0;
		for (int i = 2; i <= I1; i++) {
			kk++;
			ABD(kk, kk) = SO(i, j, KO);
			if (i != 2) {
				ABD(kk - 1, kk) = -SO(i, j, KW);
				ABD(kk - I2 - 1, kk) = nine ? -SO(i, j, KSW) : 0.0;
			}
			ABD(kk - I2 + 1, kk) = nine ? -SO(i + 1, j, KNW) : 0.0;
			ABD(kk - I2, kk) = -SO(i, j, KS);
		}
		if (PER_X(ipn)) {
			ABD(kk - I2 + 1, kk) = -SO(II, j, KW);
			ABD(kk - 2 * I2 + 1, kk) = nine ? -SO(II, j, KNW) : 0.0;
		}
	}
	if (PER_Y(ipn)) {
		kk = kk - I2;
		const int J2 = (J1 - 2) * I2;
		kk++;
		ABD(kk - J2, kk) = -SO(2, JJ, KS);
		ABD(kk - J2 + 1, kk) = nine ? -SO(3, JJ, KSW) : 0.0;
		if (ipn == 3) ABD(I2, kk) = nine ? -SO(2, JJ, KNW) : 0.0;
		for (int i = 3; i <= I1; i++) {
			kk++;
			ABD(kk - J2, kk) = -SO(i, JJ, KS);
			ABD(kk - J2 - 1, kk) = nine ? -SO(i, JJ, KNW) : 0.0;
			ABD(kk - J2 + 1, kk) = nine ? -SO(i + 1, JJ, KSW) : 0.0;
		}
		ABD(kk - J2 + 1, kk) = 0.0;
		if (ipn == 3) ABD(1, kk) = nine ? -SO(II, JJ, KSW) : 0.0;
		if (PER_X(ipn)) ABD(kk - 2 * I2 + 1, kk) = nine ? -SO(II, J1, KNW) : 0.0;
	}
	return orc_dpotrf_upper(n, abd, (int)nabd1);
#undef SO
}

/* src/2d/ftn/BMG2_SymStd_SOLVE_cg.f90:95-163: DPOTRS, then (jpn != 0) the mean of the solution is
 * removed, then the periodic ghosts incl. the four corners for per_xy */
int orc2_solve_cg_per(real_t *q, const real_t *qf, len_t II, len_t JJ,
                      const real_t *abd, real_t *bbd, len_t nabd1, int ipn)
{
	const int I1 = (int)II - 1, J1 = (int)JJ - 1;
	int kk = 0;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++)
			bbd[kk++] = F2(qf, II, i, j);
	orc_dpotrs_upper(kk, abd, (int)nabd1, bbd);
	kk = 0;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++)
			F2(q, II, i, j) = bbd[kk++];
	real_t cint = 0.0, qint = 0.0;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++) {
			qint = qint + F2(q, II, i, j);
			cint = cint + 1;
		}
	const real_t c = -qint / cint;
	for (int j = 2; j <= J1; j++)
		for (int i = 2; i <= I1; i++)
			F2(q, II, i, j) = F2(q, II, i, j) + c;
	if (PER_Y(ipn))
		for (int i = 2; i <= I1; i++) {
			F2(q, II, i, JJ) = F2(q, II, i, 2);
			F2(q, II, i, 1) = F2(q, II, i, J1);
		}
	if (PER_X(ipn))
		for (int j = 2; j <= J1; j++) {
			F2(q, II, II, j) = F2(q, II, 2, j);
			F2(q, II, 1, j) = F2(q, II, I1, j);
		}
	if (ipn == 3) {
		F2(q, II, 1, 1) = F2(q, II, I1, J1);
		F2(q, II, II, 1) = F2(q, II, 2, J1);
		F2(q, II, 1, JJ) = F2(q, II, I1, 2);
		F2(q, II, II, JJ) = F2(q, II, 2, 2);
	}
	return 0;
#undef ABD
}

/* ------------------------------------------------------------------ periodic line relaxation
 * A line that closes on itself is a cyclic tridiagonal system; the reference folds the wrap-around
 * coupling into the two end diagonals at set-up and corrects each solve with the Sherman-Morrison
 * formula (a second DPTTRS with the rank-one column).  Lines in the non-periodic direction use the
 * ordinary solve; only the ghost wraps differ. */
#define SO(i, j, s) S2(so, II, JJ, i, j, s)
#define Q(i, j) F2(q, II, i, j)
#define QF(i, j) F2(qf, II, i, j)
#define SORT(j, i, s) (sor)[(size_t)((j)-1) + (size_t)JJ * ((size_t)((i)-1) + (size_t)II * (size_t)(s))]

/* src/2d/ftn/BMG2_SymStd_SETUP_lines_x.f90:68-87 */
void orc2_setup_lines_x_per(const real_t *so, real_t *sor, len_t II, len_t JJ, int ipn)
{
	for (len_t j = 2; j <= JJ - 1; j++)
		for (len_t i = 2; i <= II - 1; i++) {
			S2(sor, II, JJ, i, j, 1) = -SO(i, j, KW);
			S2(sor, II, JJ, i, j, 0) = SO(i, j, KO);
		}
	if (PER_X(ipn))
		for (len_t j = 2; j <= JJ - 1; j++) {
			S2(sor, II, JJ, 2, j, 0) = S2(sor, II, JJ, 2, j, 0) + SO(2, j, KW);
			S2(sor, II, JJ, II - 1, j, 0) = S2(sor, II, JJ, II - 1, j, 0) + SO(II, j, KW);
		}
	for (len_t j = 2; j <= JJ - 1; j++)
		orc_dpttrf((int)II - 2, &S2(sor, II, JJ, 2, j, 0), &S2(sor, II, JJ, 3, j, 1));
}

/* src/2d/ftn/BMG2_SymStd_SETUP_lines_y.f90:69-92 (SOR transposed) */
void orc2_setup_lines_y_per(const real_t *so, real_t *sor, len_t II, len_t JJ, int ipn)
{
	for (len_t i = 2; i <= II - 1; i++)
		for (len_t j = 2; j <= JJ - 1; j++) {
			SORT(j, i, 1) = -SO(i, j, KS);
			SORT(j, i, 0) = SO(i, j, KO);
		}
	if (PER_Y(ipn))
		for (len_t i = 2; i <= II - 1; i++) {
			SORT(2, i, 0) = SORT(2, i, 0) + SO(i, 2, KS);
			SORT(JJ - 1, i, 0) = SORT(JJ - 1, i, 0) + SO(i, JJ, KS);
		}
	for (len_t i = 2; i <= II - 1; i++)
		orc_dpttrf((int)JJ - 2, &SORT(2, i, 0), &SORT(3, i, 1));
}

/* src/2d/ftn/BMG2_SymStd_relax_lines_x.f90: ipn = per_y -> :75-176 (ordinary solves, one y wrap at the
 * end); ipn = per_x / per_xy -> :178-300 (Sherman-Morrison, y then x wrap after each colour).
 * b: scratch of II doubles. */
void orc2_relax_lines_x_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, real_t *b,
                            len_t II, len_t JJ, int ifd, int updown, int ipn)
{
	const int I1 = (int)II - 1, J1 = (int)JJ - 1;
	const int jstart = updown == BMG_DOWN ? 3 : 2, jend = updown == BMG_DOWN ? 2 : 3, jstride = updown == BMG_DOWN ? -1 : 1;
	if (!PER_X(ipn)) {
		orc2_relax_lines_x(so, qf, q, sor, II, JJ, ifd, updown);
		if (ipn == 1) wrap_y(q, II, JJ);
		return;
	}
	for (int jbeg = jstart; jbeg != jend + jstride; jbeg += jstride) {
		for (int j = jbeg; j <= J1; j += 2) {
			for (int i = 2; i <= I1; i++) {
				if (ifd != 1)
					Q(i, j) = QF(i, j) + SO(i, j, KS) * Q(i, j - 1) + SO(i, j + 1, KS)
					          * Q(i, j + 1) + SO(i, j, KSW) * Q(i - 1, j - 1) + SO(i + 1, j, KNW)
					          * Q(i + 1, j - 1) + SO(i, j + 1, KNW) * Q(i - 1, j + 1)
					          + SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1);
				else
					Q(i, j) = QF(i, j) + SO(i, j, KS) * Q(i, j - 1) + SO(i, j + 1, KS) * Q(i, j + 1);
			}
			orc_dpttrs(I1 - 1, &S2(sor, II, JJ, 2, j, 0), &S2(sor, II, JJ, 3, j, 1), &Q(2, j));
			for (int i = 2; i <= I1; i++) b[i - 1] = 0.0;
			b[2 - 1] = -SO(2, j, KW);
			b[I1 - 1] = -SO(II, j, KW);
			orc_dpttrs(I1 - 1, &S2(sor, II, JJ, 2, j, 0), &S2(sor, II, JJ, 3, j, 1), &b[1]);
			real_t alpha = b[2 - 1] + b[I1 - 1];
			real_t beta = Q(2, j) + Q(I1, j);
			beta = beta / (1.0 + alpha);
			for (int i = 2; i <= I1; i++) Q(i, j) = Q(i, j) - beta * b[i - 1];
		}
		if (PER_Y(ipn)) wrap_y(q, II, JJ);
		wrap_x(q, II, JJ);
	}
}

/* src/2d/ftn/BMG2_SymStd_relax_lines_y.f90: ipn = per_x -> :77-176 (ordinary solves, one x wrap at the
 * end); per_y / per_xy -> :178-300.  b: scratch of 2*JJ doubles. */
void orc2_relax_lines_y_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, real_t *b,
                            len_t II, len_t JJ, int ifd, int updown, int ipn)
{
	const int I1 = (int)II - 1, J1 = (int)JJ - 1;
	const int istart = updown == BMG_DOWN ? 3 : 2, iend = updown == BMG_DOWN ? 2 : 3, istride = updown == BMG_DOWN ? -1 : 1;
	if (!PER_Y(ipn)) {
		orc2_relax_lines_y(so, qf, q, sor, b, II, JJ, ifd, updown);
		if (ipn == 2) wrap_x(q, II, JJ);
		return;
	}
	real_t *b2 = b + JJ; /* B(JJ+J) */
	for (int ibeg = istart; ibeg != iend + istride; ibeg += istride) {
		if (ifd == 1) /* five point: the right-hand sides of the whole colour first, in place (:262-268) */
			for (int j = 2; j <= J1; j++)
				for (int i = ibeg; i <= I1; i += 2)
					Q(i, j) = QF(i, j) + SO(i, j, KW) * Q(i - 1, j) + SO(i + 1, j, KW) * Q(i + 1, j);
		for (int i = ibeg; i <= I1; i += 2) {
			for (int j = 2; j <= J1; j++) {
				if (ifd != 1)
					b[j - 1] = QF(i, j) + SO(i, j, KW) * Q(i - 1, j) + SO(i + 1, j, KW)
					           * Q(i + 1, j) + SO(i, j, KSW) * Q(i - 1, j - 1) + SO(i + 1, j, KNW)
					           * Q(i + 1, j - 1) + SO(i, j + 1, KNW) * Q(i - 1, j + 1)
					           + SO(i + 1, j + 1, KSW) * Q(i + 1, j + 1);
				else
					b[j - 1] = Q(i, j);
			}
			orc_dpttrs(J1 - 1, &SORT(2, i, 0), &SORT(3, i, 1), &b[1]);
			for (int j = 2; j <= J1; j++) b2[j - 1] = 0.0;
			b2[2 - 1] = -SO(i, 2, KS);
			b2[J1 - 1] = -SO(i, JJ, KS);
			orc_dpttrs(J1 - 1, &SORT(2, i, 0), &SORT(3, i, 1), &b2[1]);
			real_t alpha = b2[2 - 1] + b2[J1 - 1];
			real_t beta = b[2 - 1] + b[J1 - 1];
			beta = beta / (1.0 + alpha);
			for (int j = 2; j <= J1; j++) Q(i, j) = b[j - 1] - beta * b2[j - 1];
		}
		wrap_y(q, II, JJ);
		if (PER_X(ipn)) wrap_x(q, II, JJ);
	}
}
#undef SO
#undef Q
#undef QF
#undef SORT
