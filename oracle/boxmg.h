/* TEST INFRASTRUCTURE ONLY -- never linked, loaded or called by the product
 * path (cedar_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Plain-C restatement of the BoxMG serial hot path of OVGULIU/cedar
 * (Fortran kernels under src/2d/ftn and src/3d/ftn, orchestration from
 * include/cedar/multilevel.h and include/cedar/cycle/vcycle.h).  Every
 * function cites the reference file:line it follows.
 *
 * Parity status: PINNED.  Each routine is checked element-wise against the
 * reference's own Fortran compiled in the build container (oracle/_ref,
 * recipe in oracle/Makefile) through the golden vectors under tests/golden/
 * (generator: oracle/gen_golden.py).
 *
 * Conventions (identical to the reference):
 *   - real_t = double, len_t = unsigned int  (include/cedar/types.h:43-44)
 *   - Fortran order, first index fastest, one ghost layer on every side;
 *     II = nx+2 etc.  Pointers address the first element including ghosts.
 *   - symmetric half stencils with positive off-diagonals
 *     (src/2d/ftn/BMG_stencils_f90.h:29-71).
 *   - Dirichlet ("definite", ibc = 0) boundaries; the periodic codes live in boxmg2_per.c / boxmg3_per.c.
 * Compiled with -ffp-contract=off so that every expression rounds exactly
 * like the (FMA-free, -O2) flang build of the reference.
 */
#ifndef ORACLE_BOXMG_H
#define ORACLE_BOXMG_H

#include <stddef.h>

typedef double real_t;
typedef unsigned int len_t;

enum { BMG_DOWN = 0, BMG_UP = 1 };

/* 2D stencil slots (0-based plane index = Fortran slot - 1) */
enum { KO = 0, KW = 1, KS = 2, KSW = 3, KNW = 4 };
/* 2D interpolation slots */
enum { LL = 0, LR = 1, LA = 2, LB = 3, LSW = 4, LNW = 5, LNE = 6, LSE = 7 };

/* 3D stencil slots */
enum { KP = 0, KPW = 1, KPS = 2, KB = 3, KPSW = 4, KPNW = 5, KBW = 6, KBNW = 7,
       KBN = 8, KBNE = 9, KBE = 10, KBSE = 11, KBS = 12, KBSW = 13 };
/* 3D interpolation slots */
enum { LXYL = 0, LXYR = 1, LXYA = 2, LXYB = 3, LXZA = 4, LXZB = 5,
       LXYNE = 6, LXYSE = 7, LXYSW = 8, LXYNW = 9, LXZSW = 10, LXZNW = 11,
       LXZNE = 12, LXZSE = 13, LYZSW = 14, LYZNW = 15, LYZNE = 16, LYZSE = 17,
       LBSW = 18, LBNW = 19, LBNE = 20, LBSE = 21,
       LTSW = 22, LTNW = 23, LTNE = 24, LTSE = 25 };

/* ---- 2D kernels (boxmg2.c) ---- */
void orc2_setup_recip(const real_t *so, real_t *sor, len_t II, len_t JJ);
void orc2_relax_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   len_t II, len_t JJ, int ifd, int updown);
void orc2_setup_lines_x(const real_t *so, real_t *sor, len_t II, len_t JJ);
void orc2_setup_lines_y(const real_t *so, real_t *sor, len_t II, len_t JJ);
void orc2_relax_lines_x(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                        len_t II, len_t JJ, int ifd, int updown);
void orc2_relax_lines_y(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                        real_t *b, len_t II, len_t JJ, int ifd, int updown);
void orc2_residual(const real_t *so, const real_t *qf, const real_t *q, real_t *res,
                   len_t II, len_t JJ, int ifd);
void orc2_matvec(const real_t *so, const real_t *q, real_t *qf, len_t II, len_t JJ, int ifd);
void orc3_matvec(const real_t *so, const real_t *q, real_t *qf, len_t II, len_t JJ, len_t KK, int ifd);
void orc2_restrict(const real_t *q, real_t *qc, const real_t *ci,
                   len_t II, len_t JJ, len_t IIC, len_t JJC);
void orc2_interp_add(real_t *q, const real_t *qc, real_t *res, const real_t *so,
                     const real_t *ci, len_t IIC, len_t JJC, len_t IIF, len_t JJF);
void orc2_setup_interp(const real_t *so, real_t *ci, len_t IIF, len_t JJF,
                       len_t IIC, len_t JJC, int ifd);
void orc2_setup_interp_ex(const real_t *so, real_t *ci, len_t IIF, len_t JJF,
                          len_t IIC, len_t JJC, int ifd, int phase_mask, int ilo, int jlo);
void orc2_relax_colour(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, int ifd, int pts);
void orc2_relax_column(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, int i, int jb);
void orc2_galerkin(const real_t *so, real_t *soc, const real_t *ci, len_t IIF, len_t JJF,
                   len_t IIC, len_t JJC, int ifd);
int orc2_setup_cg(const real_t *so, len_t II, len_t JJ, int nstncl,
                  real_t *abd, len_t nabd1, len_t nabd2);
int orc2_solve_cg(real_t *q, const real_t *qf, len_t II, len_t JJ,
                  const real_t *abd, real_t *bbd, len_t nabd1, len_t nabd2);

/* ---- 2D periodic branches (boxmg2_per.c; ipn = 1 per_y, 2 per_x, 3 per_xy) ---- */
void orc2_setup_interp_per(const real_t *so, real_t *ci, len_t IIF, len_t JJF,
                           len_t IIC, len_t JJC, int ifd, int ipn);
void orc2_relax_gs_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, int ifd, int updown, int ipn);
void orc2_restrict_per(real_t *q, real_t *qc, const real_t *ci,
                       len_t II, len_t JJ, len_t IIC, len_t JJC, int ipn);
void orc2_interp_add_per(real_t *q, const real_t *qc, real_t *res, const real_t *so,
                         const real_t *ci, len_t IIC, len_t JJC, len_t IIF, len_t JJF, int ipn);
void orc2_galerkin_per(const real_t *so, real_t *soc, const real_t *ci, len_t IIF, len_t JJF,
                       len_t IIC, len_t JJC, int ifd, int ipn);
void orc2_setup_lines_x_per(const real_t *so, real_t *sor, len_t II, len_t JJ, int ipn);
void orc2_setup_lines_y_per(const real_t *so, real_t *sor, len_t II, len_t JJ, int ipn);
void orc2_relax_lines_x_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, real_t *b,
                            len_t II, len_t JJ, int ifd, int updown, int ipn);
void orc2_relax_lines_y_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, real_t *b,
                            len_t II, len_t JJ, int ifd, int updown, int ipn);
int orc2_setup_cg_per(const real_t *so, len_t II, len_t JJ, int nstncl, real_t *abd, len_t nabd1, int ipn);
int orc2_solve_cg_per(real_t *q, const real_t *qf, len_t II, len_t JJ,
                      const real_t *abd, real_t *bbd, len_t nabd1, int ipn);

/* ---- 3D kernels (boxmg3.c) ---- */
void orc3_setup_recip(const real_t *so, real_t *sor, len_t II, len_t JJ, len_t KK);
void orc3_relax_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   len_t II, len_t JJ, len_t KK, int ifd, int updown);
void orc3_relax_colour(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, len_t KK, int ifd, int pts);
void orc3_relax_colour_part(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                            len_t II, len_t JJ, len_t KK, int pts, int part);
void orc3_relax_column(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, len_t KK, int i, int jb, int kb);
void orc3_relax_row(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                    len_t II, len_t JJ, len_t KK, int ifd, int pts, int j, int k);
void orc3_setup_interp_ex(const real_t *so, real_t *ci, len_t IIF, len_t JJF, len_t KKF,
                          len_t IIC, len_t JJC, len_t KKC, int ifd, int phase_mask, int ilo, int jlo, int klo);
void orc3_residual(const real_t *so, const real_t *qf, const real_t *q, real_t *res,
                   len_t II, len_t JJ, len_t KK, int ifd);
void orc3_restrict(const real_t *q, real_t *qc, const real_t *ci,
                   len_t II, len_t JJ, len_t KK, len_t IIC, len_t JJC, len_t KKC);
void orc3_interp_add(real_t *q, const real_t *qc, const real_t *so, real_t *res,
                     const real_t *ci, len_t IIC, len_t JJC, len_t KKC,
                     len_t IIF, len_t JJF, len_t KKF);
void orc3_setup_interp(const real_t *so, real_t *ci, len_t IIF, len_t JJF, len_t KKF,
                       len_t IIC, len_t JJC, len_t KKC, int ifd);
void orc3_galerkin(const real_t *so, real_t *soc, const real_t *ci,
                   len_t IIF, len_t JJF, len_t KKF, len_t IIC, len_t JJC, len_t KKC, int ifd);
int orc3_setup_cg(const real_t *so, len_t II, len_t JJ, len_t KK, int nstncl,
                  real_t *abd, len_t nabd1, len_t nabd2);
int orc3_solve_cg(real_t *q, const real_t *qf, len_t II, len_t JJ, len_t KK,
                  const real_t *abd, real_t *bbd, len_t nabd1, len_t nabd2);

/* ---- 3D periodic boundary conditions (boxmg3_per.c); ipn as BMG_get_bc.f90 gives it ---- */
int orc3_per_x(int ipn);
int orc3_per_y(int ipn);
int orc3_per_z(int ipn);
void orc3_wrap(real_t *a, len_t II, len_t JJ, len_t KK, int nplanes, int ipn);
void orc3_relax_gs_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                       len_t II, len_t JJ, len_t KK, int ifd, int updown, int ipn);
void orc3_restrict_per(real_t *q, real_t *qc, const real_t *ci, len_t II, len_t JJ, len_t KK,
                       len_t IIC, len_t JJC, len_t KKC, int ipn);
void orc3_interp_add_per(real_t *q, const real_t *qc, const real_t *so, real_t *res, const real_t *ci,
                         len_t IIC, len_t JJC, len_t KKC, len_t IIF, len_t JJF, len_t KKF, int ipn);
void orc3_setup_interp_per(const real_t *so, real_t *ci, len_t IIF, len_t JJF, len_t KKF,
                           len_t IIC, len_t JJC, len_t KKC, int ifd, int ipn);
void orc3_galerkin_per(const real_t *so, real_t *soc, const real_t *ci, len_t IIF, len_t JJF, len_t KKF,
                       len_t IIC, len_t JJC, len_t KKC, int ifd, int ipn);
int orc3_setup_cg_per(const real_t *so, len_t II, len_t JJ, len_t KK, int nstncl, real_t *abd, len_t nabd1, int ipn);
int orc3_solve_cg_per(real_t *q, const real_t *qf, len_t II, len_t JJ, len_t KK,
                      const real_t *abd, real_t *bbd, len_t nabd1, int ipn);

/* ---- LAPACK subset (lapack_mini.c): reference-LAPACK 3.x algorithms ---- */
int orc_dpttrf(int n, real_t *d, real_t *e);
void orc_dpttrs(int n, const real_t *d, const real_t *e, real_t *b);
int orc_dpbtrf_upper(int n, int kd, real_t *ab, int ldab);
void orc_dpbtrs_upper(int n, int kd, const real_t *ab, int ldab, real_t *b);
int orc_dpotrf_upper(int n, real_t *a, int lda);
void orc_dpotrs_upper(int n, const real_t *a, int lda, real_t *b);

/* ---- norms (include/cedar/2d/grid_func.h:42-53, src/2d/grid_func.cc:118-134) ---- */
real_t orc_l2_norm2(const real_t *v, len_t II, len_t JJ);
real_t orc_l2_norm3(const real_t *v, len_t II, len_t JJ, len_t KK);
real_t orc_inf_norm3(const real_t *v, len_t II, len_t JJ, len_t KK);

/* ---- multilevel driver (mlsolve.c) ---- */
enum { ORC_RELAX_POINT = 0, ORC_RELAX_LINE_X = 1, ORC_RELAX_LINE_Y = 2, ORC_RELAX_LINE_XY = 3,
       ORC_RELAX_PLANE_XY = 4, ORC_RELAX_PLANE_XZ = 5, ORC_RELAX_PLANE_YZ = 6, ORC_RELAX_PLANE_XYZ = 7 };

typedef struct orc_ml orc_ml;

/* nd = 2 or 3; nstencil = 3|5 (2D) or 4|14 (3D); so is copied. */
orc_ml *orc_ml_create(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                      int relax, int nrelax_pre, int nrelax_post, int min_coarse,
                      int num_levels);
orc_ml *orc_ml_create_bc(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                         int relax, int nrelax_pre, int nrelax_post, int min_coarse,
                         int num_levels, int ibc);
orc_ml *orc_ml_create_ex(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                         int relax, int nrelax_pre, int nrelax_post, int min_coarse,
                         int num_levels, int ibc, const int *plane_cfg, real_t plane_tol);
void orc_ml_destroy(orc_ml *ml);

/* ---- plane relaxation (planes.c) ---- */
typedef struct orc_planes orc_planes;
orc_planes *orc3_planes_create(int dir, const real_t *so, len_t II, len_t JJ, len_t KK, int nst, const int *cfg, real_t tol);
void orc3_planes_destroy(orc_planes *p);
void orc3_plane_rhs(int dir, int nst, const real_t *so, const real_t *x, const real_t *b, real_t *b2,
                    len_t II, len_t JJ, len_t KK, int ipl);
void orc3_planes_relax(orc_planes *p, const real_t *so, real_t *x, const real_t *b, int updown);
int orc_ml_nlevels(const orc_ml *ml);
void orc_ml_level_dims(const orc_ml *ml, int lvl, len_t *nx, len_t *ny, len_t *nz);
/* raw access to a level's arrays for parity tests: what = "A","P","SOR0","SOR1","ABD" */
const real_t *orc_ml_level_array(const orc_ml *ml, int lvl, const char *what, size_t *len);
void orc_ml_set_cycle(orc_ml *ml, int cycle); /* 0 = V (default), 1 = F (fcycle.h) */
void orc_ml_vcycle(orc_ml *ml, real_t *x, const real_t *b); /* cycle->run(x, b) */
/* multilevel::solve: returns the number of cycles run; rel[0] = initial ||r||_2,
 * rel[1..] = ||r_i||_2 / ||r_0||_2 after each cycle */
int orc_ml_solve(orc_ml *ml, const real_t *b, real_t *x, int maxiter, real_t tol, real_t *rel);

#endif
