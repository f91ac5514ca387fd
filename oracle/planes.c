/* TEST INFRASTRUCTURE ONLY -- see boxmg.h.
 *
 * Restatement of the reference's serial plane relaxation (include/cedar/3d/relax_planes.h:36-250,
 * src/3d/relax_planes.cc:25-238): red-black over the planes of one direction, each plane "relaxed" by a 2D BoxMG
 * solve (cdr2::solver with the plane configuration, include/cedar/kernel_params.h:32, src/kernel_params.cc:72-78:
 * line-xy relaxation and max-iter 1 unless "plane-config" says otherwise) of
 *      A2 x2 = b - (couplings to the two neighbouring planes) x .
 *
 * Two properties of the reference that this restatement keeps on purpose:
 *   - copy_coeff (relax_planes.h:80-160) takes no plane index: it loops over ALL planes and overwrites the same 2D
 *     operator, so every plane solver of a direction is built from the coefficients of the LAST plane (k = nz for xy,
 *     j = ny for xz, i = nx for yz).  Invisible for constant coefficients (all the reference's own test uses,
 *     test/3d/test_planes.cc); kept because a drop-in must give the reference's iterates.  Since the np solvers are
 *     then identical, one 2D hierarchy serves every plane here.
 *   - the 2D diagonal is the full 3D diagonal and the off-plane couplings go to the right-hand side with the current
 *     iterate (copy_rhs, term order of relax_planes.cc:25-172).
 *
 * Parity: pinned by composition (the 2D solver is pinned against the reference's Fortran, tests/test_oracle.py) and by
 * the reference's own known-answer test (test/3d/test_planes.cc: planes solved to convergence must equal the exact 2D
 * solves; tests/test_oracle_planes.py restates it with scipy).  The C++ that orchestrates it cannot be built here
 * (Boost, nlohmann/json).
 */
#include <stdlib.h>
#include <string.h>
#include "boxmg.h"

#define F3(a, II, JJ, i, j, k) (a)[(size_t)((i)-1) + (size_t)(II) * ((size_t)((j)-1) + (size_t)(JJ) * (size_t)((k)-1))]
#define S3(a, II, JJ, KK, i, j, k, s) \
	(a)[(size_t)((i)-1) + (size_t)(II) * ((size_t)((j)-1) + (size_t)(JJ) * ((size_t)((k)-1) + (size_t)(KK) * (size_t)(s)))]
#define F2(a, II, i, j) (a)[(size_t)((i)-1) + (size_t)(II) * (size_t)((j)-1)]
#define S2(a, II, JJ, i, j, s) (a)[(size_t)((i)-1) + (size_t)(II) * ((size_t)((j)-1) + (size_t)(JJ) * (size_t)(s))]

struct orc_planes {
	int dir;            /* 0 xy, 1 xz, 2 yz */
	int nst;            /* 3D stencil planes: 4 or 14 */
	len_t II, JJ, KK;   /* 3D extents with ghosts */
	len_t I2, J2;       /* 2D extents with ghosts */
	int np;             /* planes */
	int maxiter;
	real_t tol;
	orc_ml *ml2;
	real_t *x2, *b2, *rel;
};

/* copy_coeff, relax_planes.h:80-160: what survives its loop over the planes is the last one */
static void plane_operator(const orc_planes *p, const real_t *so, real_t *so2)
{
	const len_t II = p->II, JJ = p->JJ, KK = p->KK, I2 = p->I2, J2 = p->J2;
	const int full = p->nst == 14;
#define SO(i, j, k, s) S3(so, II, JJ, KK, i, j, k, s)
#define T(a, b, s) S2(so2, I2, J2, a, b, s)
	if (p->dir == 0) {
		const int k = (int)KK - 1;
		for (int j = 1; j <= (int)JJ; j++)
			for (int i = 1; i <= (int)II; i++) {
				T(i, j, KO) = SO(i, j, k, KP);
				T(i, j, KW) = SO(i, j, k, KPW);
				T(i, j, KS) = SO(i, j, k, KPS);
				if (full) {
					T(i, j, KSW) = SO(i, j, k, KPSW);
					T(i, j, KNW) = SO(i, j, k, KPNW);
				}
			}
	} else if (p->dir == 1) {
		const int j = (int)JJ - 1;
		for (int k = 1; k <= (int)KK; k++)
			for (int i = 1; i <= (int)II; i++) {
				T(i, k, KO) = SO(i, j, k, KP);
				T(i, k, KW) = SO(i, j, k, KPW);
				T(i, k, KS) = SO(i, j, k, KB);
				if (full) {
					T(i, k, KSW) = SO(i, j, k, KBW);
					T(i, k, KNW) = SO(i, j, k, KBE);
				}
			}
	} else {
		const int i = (int)II - 1;
		for (int k = 1; k <= (int)KK; k++)
			for (int j = 1; j <= (int)JJ; j++) {
				T(j, k, KO) = SO(i, j, k, KP);
				T(j, k, KW) = SO(i, j, k, KPS);
				T(j, k, KS) = SO(i, j, k, KB);
				if (full) {
					T(j, k, KSW) = SO(i, j, k, KBS);
					T(j, k, KNW) = SO(i, j, k, KBN);
				}
			}
	}
#undef T
}

/* cfg = { relaxation (ORC_RELAX_*), nrelax_pre, nrelax_post, max_iter, min_coarse }; NULL = the reference's default
 * plane configuration (line-xy, 2, 1, 1, 3; tol 1e-8) */
orc_planes *orc3_planes_create(int dir, const real_t *so, len_t II, len_t JJ, len_t KK, int nst, const int *cfg, real_t tol)
{
	static const int dflt[5] = { ORC_RELAX_LINE_XY, 2, 1, 1, 3 };
	if (!cfg) { cfg = dflt; tol = 1e-8; }
	orc_planes *p = (orc_planes *)calloc(1, sizeof(orc_planes));
	p->dir = dir; p->nst = nst;
	p->II = II; p->JJ = JJ; p->KK = KK;
	p->I2 = dir == 2 ? JJ : II;
	p->J2 = dir == 0 ? JJ : KK;
	p->np = (int)(dir == 0 ? KK : dir == 1 ? JJ : II) - 2;
	p->maxiter = cfg[3]; p->tol = tol;
	const int nst2 = nst == 14 ? 5 : 3;
	const size_t P2 = (size_t)p->I2 * p->J2;
	real_t *so2 = (real_t *)calloc(P2 * nst2, sizeof(real_t));
	plane_operator(p, so, so2);
	p->ml2 = orc_ml_create(2, p->I2 - 2, p->J2 - 2, 1, nst2, so2, cfg[0], cfg[1], cfg[2], cfg[4], -1);
	free(so2);
	p->x2 = (real_t *)calloc(P2, sizeof(real_t));
	p->b2 = (real_t *)calloc(P2, sizeof(real_t));
	p->rel = (real_t *)calloc((size_t)p->maxiter + 2, sizeof(real_t));
	return p;
}

void orc3_planes_destroy(orc_planes *p)
{
	if (!p) return;
	orc_ml_destroy(p->ml2);
	free(p->x2); free(p->b2); free(p->rel); free(p);
}

/* copy_rhs, src/3d/relax_planes.cc:25-172 (term order kept); ipl = 1-based interior plane number */
void orc3_plane_rhs(int dir, int nst, const real_t *so, const real_t *x, const real_t *b, real_t *b2,
                    len_t II, len_t JJ, len_t KK, int ipl)
{
#define X(i, j, k) F3(x, II, JJ, i, j, k)
#define B(i, j, k) F3(b, II, JJ, i, j, k)
	const int I1 = (int)II - 1, J1 = (int)JJ - 1, K1 = (int)KK - 1;
	const int full = nst == 14;
	if (dir == 0) {
		const int k = ipl + 1;
		for (int j = 2; j <= J1; j++)
			for (int i = 2; i <= I1; i++)
				F2(b2, II, i, j) = !full
					? B(i, j, k) + SO(i, j, k, KB) * X(i, j, k - 1) + SO(i, j, k + 1, KB) * X(i, j, k + 1)
					: B(i, j, k)
					  + SO(i, j, k, KB) * X(i, j, k - 1)
					  + SO(i, j, k, KBW) * X(i - 1, j, k - 1)
					  + SO(i, j + 1, k, KBNW) * X(i - 1, j + 1, k - 1)
					  + SO(i, j + 1, k, KBN) * X(i, j + 1, k - 1)
					  + SO(i + 1, j + 1, k, KBNE) * X(i + 1, j + 1, k - 1)
					  + SO(i + 1, j, k, KBE) * X(i + 1, j, k - 1)
					  + SO(i + 1, j, k, KBSE) * X(i + 1, j - 1, k - 1)
					  + SO(i, j, k, KBS) * X(i, j - 1, k - 1)
					  + SO(i, j, k, KBSW) * X(i - 1, j - 1, k - 1)
					  + SO(i, j, k + 1, KBE) * X(i - 1, j, k + 1)
					  + SO(i, j + 1, k + 1, KBSE) * X(i - 1, j + 1, k + 1)
					  + SO(i, j + 1, k + 1, KBS) * X(i, j + 1, k + 1)
					  + SO(i + 1, j + 1, k + 1, KBSW) * X(i + 1, j + 1, k + 1)
					  + SO(i + 1, j, k + 1, KBW) * X(i + 1, j, k + 1)
					  + SO(i, j, k + 1, KB) * X(i, j, k + 1)
					  + SO(i + 1, j, k + 1, KBNW) * X(i + 1, j - 1, k + 1)
					  + SO(i, j, k + 1, KBN) * X(i, j - 1, k + 1)
					  + SO(i, j, k + 1, KBNE) * X(i - 1, j - 1, k + 1);
	} else if (dir == 1) {
		const int j = ipl + 1;
		for (int k = 2; k <= K1; k++)
			for (int i = 2; i <= I1; i++)
				F2(b2, II, i, k) = !full
					? B(i, j, k) + SO(i, j, k, KPS) * X(i, j - 1, k) + SO(i, j + 1, k, KPS) * X(i, j + 1, k)
					: B(i, j, k)
					  + SO(i, j + 1, k, KPNW) * X(i - 1, j + 1, k)
					  + SO(i, j + 1, k, KPS) * X(i, j + 1, k)
					  + SO(i + 1, j + 1, k, KPSW) * X(i + 1, j + 1, k)
					  + SO(i, j + 1, k, KBNW) * X(i - 1, j + 1, k - 1)
					  + SO(i, j + 1, k, KBN) * X(i, j + 1, k - 1)
					  + SO(i + 1, j + 1, k, KBNE) * X(i + 1, j + 1, k - 1)
					  + SO(i, j + 1, k + 1, KBSE) * X(i - 1, j + 1, k + 1)
					  + SO(i, j + 1, k + 1, KBS) * X(i, j + 1, k + 1)
					  + SO(i + 1, j + 1, k + 1, KBSW) * X(i + 1, j + 1, k + 1)
					  + SO(i, j, k, KPSW) * X(i - 1, j - 1, k)
					  + SO(i, j, k, KPS) * X(i, j - 1, k)
					  + SO(i + 1, j, k, KPNW) * X(i + 1, j - 1, k)
					  + SO(i, j, k, KBSW) * X(i - 1, j - 1, k - 1)
					  + SO(i, j, k, KBS) * X(i, j - 1, k - 1)
					  + SO(i + 1, j, k, KBSE) * X(i + 1, j - 1, k - 1)
					  + SO(i, j, k + 1, KBNE) * X(i - 1, j - 1, k + 1)
					  + SO(i, j, k + 1, KBN) * X(i, j - 1, k + 1)
					  + SO(i + 1, j, k + 1, KBNW) * X(i + 1, j - 1, k + 1);
	} else {
		const int i = ipl + 1;
		for (int k = 2; k <= K1; k++)
			for (int j = 2; j <= J1; j++)
				F2(b2, JJ, j, k) = !full
					? B(i, j, k) + SO(i, j, k, KPW) * X(i - 1, j, k) + SO(i + 1, j, k, KPW) * X(i + 1, j, k)
					: B(i, j, k)
					  + SO(i, j + 1, k, KPNW) * X(i - 1, j + 1, k)
					  + SO(i, j, k, KPW) * X(i - 1, j, k)
					  + SO(i, j, k, KPSW) * X(i - 1, j - 1, k)
					  + SO(i, j + 1, k, KBNW) * X(i - 1, j + 1, k - 1)
					  + SO(i, j, k, KBW) * X(i - 1, j, k - 1)
					  + SO(i, j, k, KBSW) * X(i - 1, j - 1, k - 1)
					  + SO(i, j + 1, k + 1, KBSE) * X(i - 1, j + 1, k + 1)
					  + SO(i, j, k + 1, KBE) * X(i - 1, j, k + 1)
					  + SO(i, j, k + 1, KBNE) * X(i - 1, j - 1, k + 1)
					  + SO(i + 1, j + 1, k, KPSW) * X(i + 1, j + 1, k)
					  + SO(i + 1, j, k, KPW) * X(i + 1, j, k)
					  + SO(i + 1, j, k, KPNW) * X(i + 1, j - 1, k)
					  + SO(i + 1, j + 1, k, KBNE) * X(i + 1, j + 1, k - 1)
					  + SO(i + 1, j, k, KBE) * X(i + 1, j, k - 1)
					  + SO(i + 1, j, k, KBSE) * X(i + 1, j - 1, k - 1)
					  + SO(i + 1, j + 1, k + 1, KBSW) * X(i + 1, j + 1, k + 1)
					  + SO(i + 1, j, k + 1, KBW) * X(i + 1, j, k + 1)
					  + SO(i + 1, j, k + 1, KBNW) * X(i + 1, j - 1, k + 1);
	}
#undef X
#undef B
}
#undef SO

/* copy32 / copy23, relax_planes.cc:176-238: whole planes, ghosts included */
static void plane_copy(const orc_planes *p, real_t *x, real_t *x2, int ipl, int to3d)
{
	const len_t II = p->II, JJ = p->JJ, KK = p->KK;
	for (int b = 1; b <= (int)p->J2; b++)
		for (int a = 1; a <= (int)p->I2; a++) {
			real_t *e3 = p->dir == 0 ? &F3(x, II, JJ, a, b, ipl + 1) : p->dir == 1 ? &F3(x, II, JJ, a, ipl + 1, b)
			                                                                        : &F3(x, II, JJ, ipl + 1, a, b);
			real_t *e2 = &F2(x2, p->I2, a, b);
			if (to3d) *e3 = *e2;
			else *e2 = *e3;
		}
	(void)KK;
}

/* relax_planes, relax_planes.h:36-72: DOWN = odd planes (1, 3, ..) then even; UP = even then odd */
void orc3_planes_relax(orc_planes *p, const real_t *so, real_t *x, const real_t *b, int updown)
{
	const int lstart = updown == BMG_DOWN ? 1 : 2, lend = updown == BMG_DOWN ? 3 : 0, lstride = updown == BMG_DOWN ? 1 : -1;
	for (int beg = lstart; beg != lend; beg += lstride)
		for (int ipl = beg; ipl < p->np + 1; ipl += 2) {
			plane_copy(p, x, p->x2, ipl, 0);
			orc3_plane_rhs(p->dir, p->nst, so, x, b, p->b2, p->II, p->JJ, p->KK, ipl);
			orc_ml_solve(p->ml2, p->b2, p->x2, p->maxiter, p->tol, p->rel); /* multilevel.h:277-298 */
			plane_copy(p, x, p->x2, ipl, 1);
		}
}
