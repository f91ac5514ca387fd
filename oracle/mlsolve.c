/* TEST INFRASTRUCTURE ONLY -- see boxmg.h.
 *
 * Restatement of the reference's multilevel orchestration for the serial
 * solvers: level sizing and allocation (include/cedar/2d/solver.h:57-116,
 * include/cedar/3d/solver.h:54-123), the set-up loop
 * (include/cedar/multilevel.h:243-265), one V-cycle
 * (include/cedar/cycle/vcycle.h:57-115) and the solve loop with its residual
 * norms (include/cedar/multilevel.h:268-298).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "boxmg.h"

typedef struct {
	len_t nx, ny, nz;     /* interior extents */
	len_t II, JJ, KK;     /* with ghosts (KK = 1 in 2D) */
	int nst;              /* stencil planes of A */
	size_t npts;
	real_t *A;            /* owned */
	real_t *P;            /* interpolation from this level to the finer one (levels >= 1) */
	real_t *x, *b, *res;  /* level 0: x and b are the caller's */
	real_t *SOR0, *SOR1;
	orc_planes *pl[3];    /* plane relaxation: xy, xz, yz solvers of this level (relax_planes.h:164-246) */
} orc_level;

struct orc_ml {
	int nd, nlev, relax, nrelax_pre, nrelax_post, cycle;
	int pcfg[5]; real_t ptol; int has_pcfg; /* "plane-config" (src/kernel_params.cc:72-78) */
	int ibc;              /* 0 Dirichlet; 1 per_y, 2 per_x, 3 per_xy; 3D also 5 per_z, 6 per_xz, 7 per_yz, 8 per_xyz
	                       * (BMG_get_bc.f90:13-20) */
	orc_level *lv;
	real_t *ABD, *bbd;
	real_t *work;         /* line-relaxation scratch B (2*(II+JJ) of level 0), separate as in the reference */
	len_t nabd1, nabd2;
};

static real_t *zalloc(size_t n) { return (real_t *)calloc(n ? n : 1, sizeof(real_t)); }

/* include/cedar/2d/solver.h:57-73: float nxc = (nx-1)/(1<<ng) + 1 with unsigned
 * integer division; loop while min(...) >= min_coarse */
static int compute_num_levels(int nd, len_t nx, len_t ny, len_t nz, int min_coarse)
{
	int ng = 0;
	float m;
	do {
		ng++;
		float nxc = (float)((nx - 1) / (1u << ng) + 1);
		float nyc = (float)((ny - 1) / (1u << ng) + 1);
		m = nxc < nyc ? nxc : nyc;
		if (nd == 3) {
			float nzc = (float)((nz - 1) / (1u << ng) + 1);
			m = m < nzc ? m : nzc;
		}
	} while (m >= (float)min_coarse);
	return ng;
}

static void level_init(orc_level *L, int nd, len_t nx, len_t ny, len_t nz, int nst, int with_P)
{
	memset(L, 0, sizeof(*L));
	L->nx = nx; L->ny = ny; L->nz = nd == 3 ? nz : 1;
	L->II = nx + 2; L->JJ = ny + 2; L->KK = nd == 3 ? nz + 2 : 1;
	L->nst = nst;
	L->npts = (size_t)L->II * L->JJ * L->KK;
	L->A = zalloc(L->npts * nst);
	L->res = zalloc(L->npts);
	L->SOR0 = zalloc(L->npts * 2);
	L->SOR1 = zalloc(L->npts * 2);
	if (with_P) {
		L->P = zalloc(L->npts * (nd == 3 ? 26 : 8));
		L->x = zalloc(L->npts);
		L->b = zalloc(L->npts);
	}
}

orc_ml *orc_ml_create(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                      int relax, int nrelax_pre, int nrelax_post, int min_coarse,
                      int num_levels)
{
	return orc_ml_create_bc(nd, nx, ny, nz, nstencil, so, relax, nrelax_pre, nrelax_post, min_coarse, num_levels, 0);
}

/* ibc != 0: the kernels' periodic branches; the coarsest operator becomes a dense matrix,
 * ABD(nxc*nyc, nxc*nyc) (include/cedar/2d/solver.h:110-114) or ABD(nxc*nyc*nzc, nxc*nyc*nzc)
 * (include/cedar/3d/solver.h:118-121).  3D: point relaxation, V-cycle. */
orc_ml *orc_ml_create_bc(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                         int relax, int nrelax_pre, int nrelax_post, int min_coarse,
                         int num_levels, int ibc)
{
	return orc_ml_create_ex(nd, nx, ny, nz, nstencil, so, relax, nrelax_pre, nrelax_post, min_coarse, num_levels, ibc, NULL, 0.0);
}

/* relax = ORC_RELAX_PLANE_* (3D): plane_cfg / plane_tol = the 2D solvers' configuration (orc3_planes_create) */
orc_ml *orc_ml_create_ex(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                         int relax, int nrelax_pre, int nrelax_post, int min_coarse,
                         int num_levels, int ibc, const int *plane_cfg, real_t plane_tol)
{
	if (relax >= ORC_RELAX_PLANE_XY && (nd != 3 || ibc != 0)) return NULL;
	if (ibc != 0 && nd == 3 && !(ibc == 1 || ibc == 2 || ibc == 3 || (ibc >= 5 && ibc <= 8))) return NULL;
	orc_ml *ml = (orc_ml *)calloc(1, sizeof(orc_ml));
	ml->ibc = ibc;
	if (plane_cfg) { memcpy(ml->pcfg, plane_cfg, sizeof(ml->pcfg)); ml->ptol = plane_tol; ml->has_pcfg = 1; }
	ml->nd = nd; ml->relax = relax;
	ml->nrelax_pre = nrelax_pre; ml->nrelax_post = nrelax_post;
	int nlev = compute_num_levels(nd, nx, ny, nz, min_coarse);
	if (num_levels > 0 && num_levels <= nlev) nlev = num_levels; /* multilevel.h:247-254 */
	ml->nlev = nlev;
	ml->lv = (orc_level *)calloc((size_t)nlev, sizeof(orc_level));

	level_init(&ml->lv[0], nd, nx, ny, nz, nstencil, 0);
	memcpy(ml->lv[0].A, so, ml->lv[0].npts * nstencil * sizeof(real_t));
	/* setup_space: coarse extent = (len_t)((n-1)/2. + 1) */
	for (int l = 1; l < nlev; l++) {
		orc_level *F = &ml->lv[l - 1];
		len_t nxc = (len_t)((F->nx - 1) / 2. + 1);
		len_t nyc = (len_t)((F->ny - 1) / 2. + 1);
		len_t nzc = nd == 3 ? (len_t)((F->nz - 1) / 2. + 1) : 1;
		level_init(&ml->lv[l], nd, nxc, nyc, nzc, nd == 3 ? 14 : 5, 1);
	}
	orc_level *C = &ml->lv[nlev - 1];
	if (nd == 2) { ml->nabd1 = ibc ? C->nx * C->ny : C->nx + 2; ml->nabd2 = C->nx * C->ny; }
	else { ml->nabd1 = ibc ? C->nx * C->ny * C->nz : C->nx * (C->ny + 1) + 2; ml->nabd2 = C->nx * C->ny * C->nz; }
	ml->ABD = zalloc((size_t)ml->nabd1 * ml->nabd2);
	ml->bbd = zalloc(ml->nabd2);
	ml->work = zalloc(2 * ((size_t)ml->lv[0].II + ml->lv[0].JJ) + 8);

	/* setup loop: interp -> operator -> relax for l = 0 .. nlev-2 */
	for (int l = 0; l < nlev - 1; l++) {
		orc_level *F = &ml->lv[l], *K = &ml->lv[l + 1];
		int ifd = (nd == 2) ? (F->nst == 3) : (F->nst == 4);
		if (nd == 2 && ibc) {
			orc2_setup_interp_per(F->A, K->P, F->II, F->JJ, K->II, K->JJ, ifd, ibc);
			orc2_galerkin_per(F->A, K->A, K->P, F->II, F->JJ, K->II, K->JJ, ifd, ibc);
			switch (relax) {
			case ORC_RELAX_POINT: orc2_setup_recip(F->A, F->SOR0, F->II, F->JJ); break;
			case ORC_RELAX_LINE_X: orc2_setup_lines_x_per(F->A, F->SOR0, F->II, F->JJ, ibc); break;
			case ORC_RELAX_LINE_Y: orc2_setup_lines_y_per(F->A, F->SOR0, F->II, F->JJ, ibc); break;
			default:
				orc2_setup_lines_x_per(F->A, F->SOR0, F->II, F->JJ, ibc);
				orc2_setup_lines_y_per(F->A, F->SOR1, F->II, F->JJ, ibc);
			}
		} else if (nd == 2) {
			orc2_setup_interp(F->A, K->P, F->II, F->JJ, K->II, K->JJ, ifd);
			orc2_galerkin(F->A, K->A, K->P, F->II, F->JJ, K->II, K->JJ, ifd);
			switch (relax) {
			case ORC_RELAX_POINT: orc2_setup_recip(F->A, F->SOR0, F->II, F->JJ); break;
			case ORC_RELAX_LINE_X: orc2_setup_lines_x(F->A, F->SOR0, F->II, F->JJ); break;
			case ORC_RELAX_LINE_Y: orc2_setup_lines_y(F->A, F->SOR0, F->II, F->JJ); break;
			default:
				orc2_setup_lines_x(F->A, F->SOR0, F->II, F->JJ);
				orc2_setup_lines_y(F->A, F->SOR1, F->II, F->JJ);
			}
		} else if (ibc) {
			orc3_setup_interp_per(F->A, K->P, F->II, F->JJ, F->KK, K->II, K->JJ, K->KK, ifd, ibc);
			orc3_galerkin_per(F->A, K->A, K->P, F->II, F->JJ, F->KK, K->II, K->JJ, K->KK, ifd, ibc);
			orc3_setup_recip(F->A, F->SOR0, F->II, F->JJ, F->KK);
		} else {
			orc3_setup_interp(F->A, K->P, F->II, F->JJ, F->KK, K->II, K->JJ, K->KK, ifd);
			orc3_galerkin(F->A, K->A, K->P, F->II, F->JJ, F->KK, K->II, K->JJ, K->KK, ifd);
			if (relax >= ORC_RELAX_PLANE_XY) { /* multilevel.h:149-159 */
				for (int d = 0; d < 3; d++)
					if (relax == ORC_RELAX_PLANE_XYZ || relax == ORC_RELAX_PLANE_XY + d)
						F->pl[d] = orc3_planes_create(d, F->A, F->II, F->JJ, F->KK, F->nst, ml->has_pcfg ? ml->pcfg : NULL, ml->ptol);
			} else
				orc3_setup_recip(F->A, F->SOR0, F->II, F->JJ, F->KK);
		}
	}
	/* setup_cg_solve (multilevel.h:95-103) */
	if (nd == 2 && ibc) orc2_setup_cg_per(C->A, C->II, C->JJ, C->nst, ml->ABD, ml->nabd1, ibc);
	else if (nd == 2) orc2_setup_cg(C->A, C->II, C->JJ, C->nst, ml->ABD, ml->nabd1, ml->nabd2);
	else if (ibc) orc3_setup_cg_per(C->A, C->II, C->JJ, C->KK, C->nst, ml->ABD, ml->nabd1, ibc);
	else orc3_setup_cg(C->A, C->II, C->JJ, C->KK, C->nst, ml->ABD, ml->nabd1, ml->nabd2);
	return ml;
}

void orc_ml_destroy(orc_ml *ml)
{
	if (!ml) return;
	for (int l = 0; l < ml->nlev; l++) {
		orc_level *L = &ml->lv[l];
		free(L->A); free(L->P); free(L->res); free(L->SOR0); free(L->SOR1);
		for (int d = 0; d < 3; d++) orc3_planes_destroy(L->pl[d]);
		if (l > 0) { free(L->x); free(L->b); }
	}
	free(ml->lv); free(ml->ABD); free(ml->bbd); free(ml->work); free(ml);
}

int orc_ml_nlevels(const orc_ml *ml) { return ml->nlev; }

void orc_ml_level_dims(const orc_ml *ml, int lvl, len_t *nx, len_t *ny, len_t *nz)
{
	*nx = ml->lv[lvl].nx; *ny = ml->lv[lvl].ny; *nz = ml->lv[lvl].nz;
}

const real_t *orc_ml_level_array(const orc_ml *ml, int lvl, const char *what, size_t *len)
{
	const orc_level *L = &ml->lv[lvl];
	if (!strcmp(what, "A")) { *len = L->npts * L->nst; return L->A; }
	if (!strcmp(what, "P")) { *len = L->P ? L->npts * (ml->nd == 3 ? 26 : 8) : 0; return L->P; }
	if (!strcmp(what, "SOR0")) { *len = L->npts * 2; return L->SOR0; }
	if (!strcmp(what, "SOR1")) { *len = L->npts * 2; return L->SOR1; }
	if (!strcmp(what, "ABD")) { *len = (size_t)ml->nabd1 * ml->nabd2; return ml->ABD; }
	*len = 0;
	return NULL;
}

static void residual(const orc_ml *ml, const orc_level *L, const real_t *x, const real_t *b, real_t *r)
{
	if (ml->nd == 2) orc2_residual(L->A, b, x, r, L->II, L->JJ, L->nst == 3);
	else orc3_residual(L->A, b, x, r, L->II, L->JJ, L->KK, L->nst == 4);
}

/* multilevel.h:165-222: pre = DOWN sweeps (line-xy: x then y), post = UP (y then x) */
static void smooth(const orc_ml *ml, orc_level *L, real_t *x, const real_t *b, int updown, int n)
{
	for (int it = 0; it < n; it++) {
		if (ml->nd == 3 && ml->ibc) {
			orc3_relax_gs_per(L->A, b, x, L->SOR0, L->II, L->JJ, L->KK, L->nst == 4, updown, ml->ibc);
			continue;
		}
		if (ml->nd == 3 && ml->relax >= ORC_RELAX_PLANE_XY) { /* multilevel.h:179-189, :208-218 */
			static const int down[3] = { 0, 2, 1 }, up[3] = { 1, 2, 0 }; /* xy, yz, xz / xz, yz, xy */
			for (int t = 0; t < 3; t++) {
				const int d = updown == BMG_DOWN ? down[t] : up[t];
				if (L->pl[d]) orc3_planes_relax(L->pl[d], L->A, x, b, updown);
			}
			continue;
		}
		if (ml->nd == 3) {
			orc3_relax_gs(L->A, b, x, L->SOR0, L->II, L->JJ, L->KK, L->nst == 4, updown);
			continue;
		}
		int ifd = L->nst == 3;
		if (ml->ibc) { /* L->res doubles as the line scratch, as in the non-periodic y-line call below */
			switch (ml->relax) {
			case ORC_RELAX_POINT: orc2_relax_gs_per(L->A, b, x, L->SOR0, L->II, L->JJ, ifd, updown, ml->ibc); break;
			case ORC_RELAX_LINE_X: orc2_relax_lines_x_per(L->A, b, x, L->SOR0, ml->work, L->II, L->JJ, ifd, updown, ml->ibc); break;
			case ORC_RELAX_LINE_Y: orc2_relax_lines_y_per(L->A, b, x, L->SOR0, ml->work, L->II, L->JJ, ifd, updown, ml->ibc); break;
			default:
				if (updown == BMG_DOWN) {
					orc2_relax_lines_x_per(L->A, b, x, L->SOR0, ml->work, L->II, L->JJ, ifd, updown, ml->ibc);
					orc2_relax_lines_y_per(L->A, b, x, L->SOR1, ml->work, L->II, L->JJ, ifd, updown, ml->ibc);
				} else {
					orc2_relax_lines_y_per(L->A, b, x, L->SOR1, ml->work, L->II, L->JJ, ifd, updown, ml->ibc);
					orc2_relax_lines_x_per(L->A, b, x, L->SOR0, ml->work, L->II, L->JJ, ifd, updown, ml->ibc);
				}
			}
			continue;
		}
		switch (ml->relax) {
		case ORC_RELAX_POINT: orc2_relax_gs(L->A, b, x, L->SOR0, L->II, L->JJ, ifd, updown); break;
		case ORC_RELAX_LINE_X: orc2_relax_lines_x(L->A, b, x, L->SOR0, L->II, L->JJ, ifd, updown); break;
		case ORC_RELAX_LINE_Y: orc2_relax_lines_y(L->A, b, x, L->SOR0, ml->work, L->II, L->JJ, ifd, updown); break;
		default:
			if (updown == BMG_DOWN) {
				orc2_relax_lines_x(L->A, b, x, L->SOR0, L->II, L->JJ, ifd, updown);
				orc2_relax_lines_y(L->A, b, x, L->SOR1, ml->work, L->II, L->JJ, ifd, updown);
			} else {
				orc2_relax_lines_y(L->A, b, x, L->SOR1, ml->work, L->II, L->JJ, ifd, updown);
				orc2_relax_lines_x(L->A, b, x, L->SOR0, L->II, L->JJ, ifd, updown);
			}
		}
	}
}

static void coarse_solve(orc_ml *ml, real_t *x, const real_t *b)
{
	orc_level *C = &ml->lv[ml->nlev - 1];
	if (ml->nd == 2 && ml->ibc) orc2_solve_cg_per(x, b, C->II, C->JJ, ml->ABD, ml->bbd, ml->nabd1, ml->ibc);
	else if (ml->nd == 2) orc2_solve_cg(x, b, C->II, C->JJ, ml->ABD, ml->bbd, ml->nabd1, ml->nabd2);
	else if (ml->ibc) orc3_solve_cg_per(x, b, C->II, C->JJ, C->KK, ml->ABD, ml->bbd, ml->nabd1, ml->ibc);
	else orc3_solve_cg(x, b, C->II, C->JJ, C->KK, ml->ABD, ml->bbd, ml->nabd1, ml->nabd2);
}

/* vcycle.h:57-115 */
static void ncycle(orc_ml *ml, int lvl, real_t *x, const real_t *b)
{
	orc_level *L = &ml->lv[lvl], *K = &ml->lv[lvl + 1];
	smooth(ml, L, x, b, BMG_DOWN, ml->nrelax_pre);
	residual(ml, L, x, b, L->res);
	if (ml->nd == 2 && ml->ibc) orc2_restrict_per(L->res, K->b, K->P, L->II, L->JJ, K->II, K->JJ, ml->ibc);
	else if (ml->nd == 2) orc2_restrict(L->res, K->b, K->P, L->II, L->JJ, K->II, K->JJ);
	else if (ml->ibc) orc3_restrict_per(L->res, K->b, K->P, L->II, L->JJ, L->KK, K->II, K->JJ, K->KK, ml->ibc);
	else orc3_restrict(L->res, K->b, K->P, L->II, L->JJ, L->KK, K->II, K->JJ, K->KK);
	memset(K->x, 0, K->npts * sizeof(real_t)); /* coarse_x.set(0.0) */
	if (lvl + 1 == ml->nlev - 1) coarse_solve(ml, K->x, K->b);
	else ncycle(ml, lvl + 1, K->x, K->b);
	if (ml->nd == 2 && ml->ibc) orc2_interp_add_per(x, K->x, L->res, L->A, K->P, K->II, K->JJ, L->II, L->JJ, ml->ibc);
	else if (ml->nd == 2) orc2_interp_add(x, K->x, L->res, L->A, K->P, K->II, K->JJ, L->II, L->JJ);
	else if (ml->ibc) orc3_interp_add_per(x, K->x, L->A, L->res, K->P, K->II, K->JJ, K->KK, L->II, L->JJ, L->KK, ml->ibc);
	else orc3_interp_add(x, K->x, L->A, L->res, K->P, K->II, K->JJ, K->KK, L->II, L->JJ, L->KK);
	smooth(ml, L, x, b, BMG_UP, ml->nrelax_post);
}

/* include/cedar/cycle/fcycle.h:49-83 */
static void fmg_cycle(orc_ml *ml, int lvl, real_t *x, const real_t *b)
{
	if (lvl == ml->nlev - 1) {
		coarse_solve(ml, x, b);
		return;
	}
	orc_level *L = &ml->lv[lvl], *K = &ml->lv[lvl + 1];
	/* periodic: the same kernels as the V-cycle; the periodic restriction refreshes the ghosts of the vector it restricts
	 * (restrict.f90:78-103), here the right-hand side itself, as the reference's binding does through its const_cast */
	if (ml->nd == 2 && ml->ibc) orc2_restrict_per((real_t *)b, K->b, K->P, L->II, L->JJ, K->II, K->JJ, ml->ibc);
	else if (ml->nd == 2) orc2_restrict(b, K->b, K->P, L->II, L->JJ, K->II, K->JJ);
	else if (ml->ibc) orc3_restrict_per((real_t *)b, K->b, K->P, L->II, L->JJ, L->KK, K->II, K->JJ, K->KK, ml->ibc);
	else orc3_restrict(b, K->b, K->P, L->II, L->JJ, L->KK, K->II, K->JJ, K->KK);
	fmg_cycle(ml, lvl + 1, K->x, K->b);
	memset(x, 0, L->npts * sizeof(real_t));
	memset(L->res, 0, L->npts * sizeof(real_t));
	if (ml->nd == 2 && ml->ibc) orc2_interp_add_per(x, K->x, L->res, L->A, K->P, K->II, K->JJ, L->II, L->JJ, ml->ibc);
	else if (ml->nd == 2) orc2_interp_add(x, K->x, L->res, L->A, K->P, K->II, K->JJ, L->II, L->JJ);
	else if (ml->ibc) orc3_interp_add_per(x, K->x, L->A, L->res, K->P, K->II, K->JJ, K->KK, L->II, L->JJ, L->KK, ml->ibc);
	else orc3_interp_add(x, K->x, L->A, L->res, K->P, K->II, K->JJ, K->KK, L->II, L->JJ, L->KK);
	ncycle(ml, lvl, x, b);
}

void orc_ml_set_cycle(orc_ml *ml, int cycle) { ml->cycle = cycle; }

void orc_ml_vcycle(orc_ml *ml, real_t *x, const real_t *b)
{
	if (ml->nlev == 1) coarse_solve(ml, x, b); /* vcycle.h:37-38 */
	else if (ml->cycle == 1) fmg_cycle(ml, 0, x, b);
	else ncycle(ml, 0, x, b);
}

static real_t l2(const orc_ml *ml, const orc_level *L, const real_t *v)
{
	return ml->nd == 2 ? orc_l2_norm2(v, L->II, L->JJ) : orc_l2_norm3(v, L->II, L->JJ, L->KK);
}

/* multilevel.h:277-298 */
int orc_ml_solve(orc_ml *ml, const real_t *b, real_t *x, int maxiter, real_t tol, real_t *rel)
{
	orc_level *L = &ml->lv[0];
	residual(ml, L, x, b, L->res);
	real_t res0 = l2(ml, L, L->res);
	rel[0] = res0;
	int it = 0;
	for (it = 0; it < maxiter; it++) {
		orc_ml_vcycle(ml, x, b);
		residual(ml, L, x, b, L->res);
		real_t r = l2(ml, L, L->res) / res0;
		rel[it + 1] = r;
		if (r < tol) { it++; break; }
	}
	return it;
}
