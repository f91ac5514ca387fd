/* cedar_amd -- C ABI of the MI355X-native BoxMG V-cycle hot path.
 *
 * Two layers, both plain C (pointers + sizes, no C++/torch types):
 *
 * 1. Kernel drop-ins.  Exactly the `extern "C"` symbols that Cedar's C++
 *    binding classes call into its Fortran library;
 *    same names, argument order, by-value/by-pointer conventions and array
 *    layouts (FP64, Fortran order, one ghost layer, pointer to the first
 *    element including ghosts).  Each array argument may be a host pointer
 *    (the library stages it through HBM: correct, PCIe-bound) or a device
 *    pointer obtained from cedar_amd_malloc/hipMalloc (operated on in place).
 *    Linking Cedar against libcedar_amd.so instead of its Fortran objects
 *    therefore routes the whole hot path to the GPU without source changes.
 *    Boundary codes (jpn / ibc, what BMG_get_bc returns for grid.periodic): 0 definite and the definite
 *    periodic codes -- 2D 1 (y), 2 (x), 3 (xy); 3D also 5 (z), 6 (xz), 7 (yz), 8 (xyz), with even extents in the
 *    periodic directions for the two 3D set-up routines.  The indefinite codes (< 0) report through print_error
 *    and return.  3D: where the reference's periodic Fortran is not self-consistent (the ghost loops of
 *    BMG3_SymStd_interp_add.f90:253-272, dense-matrix entries of SETUP_cg_LU for xz / xyz / xy with nx != ny) the
 *    library applies the periodic operator itself; DESIGN.md section 6 has the evidence.
 *
 * 2. Handle API.  Device-resident hierarchy, modelled on Cedar's own C
 *    interface (include/cedar/2d/interface/c/solver.h: bmg2_solver_create /
 *    _run / _destroy): create uploads the fine operator and runs the whole
 *    set-up phase on the GPU, run/vcycle keep every level in HBM.
 *
 * Errors: like the reference, kernels report through the host-supplied
 * callback `print_error(char*)` (src/2d/ftn/ModInterface.f90:4-9); if the
 * host program does not define it the library's weak default writes to stderr.
 * Threading: none (one solver per process/rank, like the reference).
 */
#ifndef CEDAR_AMD_H
#define CEDAR_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double real_t;
typedef unsigned int len_t;

/* ------------------------------------------------------------------ 1. drop-ins */

/* replaces src/2d/ftn/BMG_get_bc.f90:1-24 (decl. include/cedar/2d/relax.h:24) */
void BMG_get_bc(int per_mask, int *ibc);

/* include/cedar/2d/relax.h:13  <- src/2d/ftn/BMG2_SymStd_SETUP_recip.f90 */
void BMG2_SymStd_SETUP_recip(real_t *so, real_t *sor, len_t nx, len_t ny, int nstncl, int nsor_v);
/* include/cedar/2d/relax.h:14-15 <- BMG2_SymStd_SETUP_lines_x/_y.f90 */
void BMG2_SymStd_SETUP_lines_x(real_t *SO, real_t *SOR, len_t Nx, len_t Ny, int NStncl, int JPN);
void BMG2_SymStd_SETUP_lines_y(real_t *SO, real_t *SOR, len_t Nx, len_t Ny, int NStncl, int JPN);
/* include/cedar/2d/relax.h:16-17 <- BMG2_SymStd_relax_GS.f90 */
void BMG2_SymStd_relax_GS(int k, real_t *SO, real_t *QF, real_t *Q, real_t *SOR, len_t II, len_t JJ,
                          int kf, int ifd, int nstncl, int nsorv, int irelax_sym, int updown, int jpn);
/* include/cedar/2d/relax.h:18-23 <- BMG2_SymStd_relax_lines_x/_y.f90 */
void BMG2_SymStd_relax_lines_x(int k, real_t *SO, real_t *QF, real_t *Q, real_t *SOR, real_t *B,
                               len_t II, len_t JJ, int kf, int ifd, int nstencil, int irelax_sym,
                               int updown, int jpn);
void BMG2_SymStd_relax_lines_y(int k, real_t *SO, real_t *QF, real_t *Q, real_t *SOR, real_t *B,
                               len_t II, len_t JJ, int kf, int ifd, int nstencil, int irelax_sym,
                               int updown, int jpn);
/* include/cedar/2d/residual.h:9-10 <- BMG2_SymStd_residual.f90 (everything by pointer) */
void BMG2_SymStd_residual(int *k, real_t *SO, real_t *QF, real_t *Q, real_t *RES, len_t *II, len_t *JJ,
                          int *kf, int *ifd, int *nstncl, int *ibc, int *irelax, int *irelax_sym,
                          int *updown);
/* src/2d/restrict.cc:6-7 <- BMG2_SymStd_restrict.f90 */
void BMG2_SymStd_restrict(real_t *Q, real_t *QC, real_t *CI, int Nx, int Ny, int Nxc, int Nyc, int jpn);
/* src/2d/interp.cc:8-9 <- BMG2_SymStd_interp_add.f90 */
void BMG2_SymStd_interp_add(real_t *Q, real_t *QC, real_t *RES, real_t *SO, real_t *CI,
                            len_t IIC, len_t JJC, len_t IIF, len_t JJF, int nstncl, int jpn);
/* src/2d/interp.cc:10-12 <- BMG2_SymStd_SETUP_interp_OI.f90 */
void BMG2_SymStd_SETUP_interp_OI(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf,
                                 len_t iic, len_t jjc, int ifd, int nstncl, int jpn, int irelax);
/* include/cedar/2d/coarsen.h:10-12 <- BMG2_SymStd_SETUP_ITLI_ex.f90 */
void BMG2_SymStd_SETUP_ITLI_ex(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf,
                               len_t iic, len_t jjc, int ifd, int nstncl, int ipn);
/* include/cedar/2d/solve_cg.h:10 <- BMG2_SymStd_SETUP_cg_LU.f90 (everything by pointer) */
void BMG2_SymStd_SETUP_cg_LU(real_t *so, len_t *ii, len_t *jj, int *nstncl, real_t *abd,
                             len_t *nabd1, len_t *nabd2, int *ibc);
/* src/2d/solve_cg.cc:6 <- BMG2_SymStd_SOLVE_cg.f90 */
void BMG2_SymStd_SOLVE_cg(real_t *q, real_t *qf, len_t ii, len_t jj, real_t *abd, real_t *bbd,
                          len_t nabd1, len_t nabd2, int ibc);

/* include/cedar/3d/relax.h:11-16 */
void BMG3_SymStd_SETUP_recip(real_t *so, real_t *sor, len_t nx, len_t ny, len_t nz, int nstencl, int nsorv);
void BMG3_SymStd_relax_GS(int kg, real_t *so, real_t *qf, real_t *q, real_t *sor,
                          len_t ii, len_t jj, len_t kk, int ifd, int nstncl, int nsorv,
                          int irelax_sym, int updown, int jpn);
/* include/cedar/3d/residual.h:10-13 */
void BMG3_SymStd_residual(int kg, int NOG, int ifd, real_t *q, real_t *qf, real_t *so, real_t *RES,
                          len_t ii, len_t jj, len_t kk, int NStncl);
/* src/3d/restrict.cc:7-10 */
void BMG3_SymStd_restrict(real_t *q, real_t *qc, real_t *ci, len_t nx, len_t ny, len_t nz,
                          len_t nxc, len_t nyc, len_t nzc, int jpn);
/* src/3d/interp.cc:7-11 (NB: so,res order differs from 2D) */
void BMG3_SymStd_interp_add(real_t *q, real_t *qc, real_t *so, real_t *res, real_t *ci,
                            len_t iic, len_t jjc, len_t kkc, len_t iif, len_t jjf, len_t kkf,
                            int NStncl, int jpn);
/* include/cedar/3d/interp.h:11-15 (yo is scratch of the reference and is not touched) */
void BMG3_SymStd_SETUP_interp_OI(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                 len_t iic, len_t jjc, len_t kkc, int ifd, int nstncl, int irelax,
                                 int jpn, real_t *yo);
/* include/cedar/3d/coarsen.h:12-16 */
void BMG3_SymStd_SETUP_ITLI07_ex(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                 len_t iic, len_t jjc, len_t kkc, int ipn);
void BMG3_SymStd_SETUP_ITLI27_ex(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                 len_t iic, len_t jjc, len_t kkc, int ipn);
/* include/cedar/3d/solve_cg.h:10-11, src/3d/solve_cg.cc:6-9 */
void BMG3_SymStd_SETUP_cg_LU(real_t *so, len_t ii, len_t jj, len_t kk, int NStncl, real_t *abd,
                             len_t nabd1, len_t nabd2, int ibc);
void BMG3_SymStd_SOLVE_cg(real_t *q, real_t *qf, len_t ii, len_t jj, len_t kk, real_t *abd,
                          real_t *bbd, len_t nabd1, len_t nabd2, int ibc);

/* ------------------------------------------------------------------ device memory helpers */
int cedar_amd_device_count(void);                 /* 0 when no GPU is visible (no HIP context made) */
int cedar_amd_set_device(int dev);                /* 0 on success */
void *cedar_amd_malloc(size_t bytes);             /* zero-filled HBM */
void cedar_amd_free(void *dptr);
void cedar_amd_memcpy_h2d(void *dst, const void *src, size_t bytes);
void cedar_amd_memcpy_d2h(void *dst, const void *src, size_t bytes);
void cedar_amd_memcpy_d2d(void *dst, const void *src, size_t bytes);
void cedar_amd_memset(void *dst, int value, size_t bytes);
void cedar_amd_sync(void);
/* all launches of this library go to this stream (default: the null stream) */
void cedar_amd_set_stream(void *hip_stream);
void *cedar_amd_get_stream(void);
/* ||v||_2 over the interior (grid_func::lp_norm<2>); v host or device; KK = 1 for 2D */
double cedar_amd_l2norm(const real_t *v, len_t II, len_t JJ, len_t KK);

/* qf = A q on the interior (kernels::matvec; arithmetic of src/2d/ftn/mpi/BMG2_SymStd_UTILS_matvec.f90:84-118
 * and src/3d/ftn/mpi/BMG3_SymStd_UTILS_matvec.f90:80-127, on the serial array layout of this header:
 * SO(II,JJ[,KK],nstncl)); nstncl = 3|5 (2D), 4|14 (3D); host or device pointers. */
void cedar_amd_matvec2(const real_t *so, const real_t *q, real_t *qf, len_t II, len_t JJ, int nstncl);
void cedar_amd_matvec3(const real_t *so, const real_t *q, real_t *qf, len_t II, len_t JJ, len_t KK, int nstncl);

/* device-side gallery (src/2d/gallery.cc, src/3d/gallery.cc); `so`/`b` device or host.
 * which: 0 poisson2, 1 diag_diffusion2(dx,dy), 2 fe2, 10 poisson3, 11 diag_diffusion3, 12 fe3;
 * b (may be NULL) receives the examples' rhs (examples/basic-2d-ser/poisson.cc:15-37). */
void cedar_amd_gallery(int which, real_t *so, real_t *b, len_t nx, len_t ny, len_t nz,
                       const double *params);

/* ------------------------------------------------------------------ 1b. pieces for domain-decomposed runs
 * The MPI flavour of the reference interleaves halo exchanges with the colours of a sweep and
 * with the phases of the interpolation set-up (src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147,
 * src/3d/ftn/mpi/BMG3_SymStd_SETUP_interp_OI.f90:418-1074).  These entry points expose the same
 * granularity so that a host (cedar_amd/dist.py) can place RCCL exchanges between them.
 * Arrays: host or device, local subdomain incl. one ghost layer, Cedar layout. */
/* one row class (jb,kb in {0,1}: parity of 1-based j,k minus 2) of the 27-point sweep: both i-colours,
 * efirst != 0: even 1-based i first (the UP order) */
void cedar_amd_relax3_pass(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                           int jb, int kb, int efirst);
/* the same restricted to a part of the class rows: part 1 = rows none of whose j,k neighbours is a
 * ghost row (they do not read the y/z ghost layers and may run while those are being exchanged),
 * part 2 = the remaining shell, part 0 = all.  Rows of a class do not couple: 1 then 2 equals 0.
 * part may carry, shifted left by 4, the faces of the box that have a neighbouring rank (bit 0 -y, 1 +y, 2 -z,
 * 3 +z): rows next to a face without one read no exchanged ghost and count as interior; no bit set = all four. */
void cedar_amd_relax3_pass_part(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                                int jb, int kb, int efirst, int part);
/* both row classes of the planes of k-parity kb in sweep order (up != 0: the UP order): the unit between two
 * halo exchanges on a slab decomposition (rank grid 1 x 1 x pz), run with the plane-fused kernel on big
 * levels.  part: 0 = all planes of the parity, 1 = planes whose k-neighbours are both owned, 2 = the others;
 * the -z / +z bits of the face mask of cedar_amd_relax3_pass_part apply (part | sides << 4). */
void cedar_amd_relax3_planes(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                             int kb, int up, int part);
/* Register the 27-point operator `so` (device pointer, with its SETUP_recip output `sor`) with the library: on levels
 * with at least 320 rows the pieces above and BMG3_SymStd_relax_GS / _residual then read a row-interleaved solve copy
 *  instead of the fourteen Cedar-layout planes, and cedar_amd_relax3_planes runs the sweep with
 * inter-plane partial sums (cedar_amd_relax3_gs_psum below; levels with at least 160 rows, CEDAR_AMD_PSUM=0: never) on a
 * scratch vector kept with the registration.  Returns bit 0: a solve copy was made, bit 1: the scratch was.  Call again
 * after the operator changed; release before freeing it.  The resident solver (section 2) does this by itself. */
int cedar_amd_relax3_prepare(const real_t *so, const real_t *sor, len_t ii, len_t jj, len_t kk);
/* the same with the partial-sum sweep registered from psum_min_rows rows on (runs of 8 rows below 160 rows) for
 * cedar_amd_relax3_planes_masked: on a rank grid with an x / y split the alternative to it is not four row-class launches but
 * four passes with an exchange after each, and the sweep pays from 128 rows on (what cedar_amd_dist3_* registers) */
int cedar_amd_relax3_prepare_rows(const real_t *so, const real_t *sor, len_t ii, len_t jj, len_t kk, int psum_min_rows);
void cedar_amd_relax3_release(const real_t *so);
/* One 27-point sweep (BMG3_SymStd_relax_GS.f90:80-138, Dirichlet) with INTER-PLANE PARTIAL SUMS: the planes of the
 * first k-parity are relaxed in the reference's order and leave, per point of the planes between them, the sums of the
 * nine products towards the plane below and above; the second k-parity adds those two sums to its eight in-plane terms
 * instead of re-reading eighteen operator slot-rows.  Same products as the reference, the 26-term sum of :104-131
 * re-associated for the second k-parity: results agree with BMG3_SymStd_relax_GS to rounding (not bit for bit); this is
 * the sweep the resident solver runs on levels with at least 320 rows (CEDAR_AMD_PSUM=0: reference order).  scratch: a
 * device array of the vector's size, or NULL (the library takes one from its pool).  Run length: CEDAR_AMD_FRUN.
 * Returns 1, or 0 when the level cannot take it (rows longer than 512 points, fewer than four runs per plane) and the
 * reference-order sweep was run instead. */
int cedar_amd_relax3_gs_psum(real_t *so, real_t *qf, real_t *q, real_t *sor, real_t *scratch, len_t ii, len_t jj, len_t kk,
                             int updown);
/* The 2D analogue for nine-point operators (BMG2_SymStd_relax_GS.f90:89-114): the band-fused sweep whose S rows take the
 * six couplings to the two neighbouring F rows as ONE partial-sum row kept in LDS by the workgroup that has just relaxed
 * those F rows.  Same contract as cedar_amd_relax3_gs_psum; the resident solver runs it on levels with at least 4096 rows
 * (rows of 2048 to 4350 points; CEDAR_AMD_FRUN2 = run length).  Returns 1, or 0 when the reference-order sweep ran. */
int cedar_amd_relax2_gs_psum(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int updown);
/* recompute column icol (0-based incl. ghost) of that row class after its x-neighbour column changed */
void cedar_amd_relax3_fixup(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                            int icol, int jb, int kb);
/* Boundary-first pieces of a k-parity of planes on a rank grid with an x / y split (what cedar_amd_dist3_* runs where the
 * level takes the partial-sum sweep; the MPI flavour of the reference exchanges after every colour,
 * src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147).  The few columns and rows whose values a neighbouring rank waits for,
 * or which wait for a neighbour's, are relaxed stage by stage in the reference order with the exchanges between the
 * stages; one launch then relaxes everything else of the parity and leaves those points as they are:
 *   _rows: rows j0, j0+jstep, .. (nrj rows, 0-based incl. ghost) of every plane of parity kb, both i-colours;
 *   _cols: the points of the listed columns (0-based incl. ghost, relaxed in the order given) in every row of class
 *          (jb,kb) except the rows xrow0 / xrow1 (-1: none);
 *   _planes_masked: cedar_amd_relax3_planes for all planes of the parity, with the points named by the masks skipped --
 *          cols_f / cols_s for the first / second row class of the sweep order: bits 0..3 = columns 1..4, bits 4..7 = the
 *          last four owned columns; rows[3]: whole rows (-1: unused).  Needs even extents, at least 8 columns and the
 *          registration of cedar_amd_relax3_prepare with its scratch; returns 0 (nothing done) otherwise. */
void cedar_amd_relax3_rows(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk, int j0, int jstep,
                           int nrj, int kb, int efirst);
void cedar_amd_relax3_cols(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk, int jb, int kb,
                           int ncol, const int *cols, int xrow0, int xrow1);
int cedar_amd_relax3_planes_masked(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk, int kb,
                                   int up, unsigned cols_f, unsigned cols_s, const int *rows);
/* _cols on a dense copy of the six operator columns next to an x face (device pointers only; at least 12 columns per row):
 * _strip_build fills out[_strip_doubles(jj,kk)] for the low (side 0: columns 0..5) or the high (side 1: ii-6..ii-1) face from
 * the operator and its SETUP_recip output; _cols_strip takes the copies of the sides its columns lie on (the other may be
 * NULL).  Same arithmetic as _cols, a fraction of its memory traffic. */
size_t cedar_amd_relax3_strip_doubles(len_t jj, len_t kk);
void cedar_amd_relax3_strip_build(const real_t *so, const real_t *sor, len_t ii, len_t jj, len_t kk, int side, real_t *out);
void cedar_amd_relax3_cols_strip(const real_t *strip_lo, const real_t *strip_hi, real_t *qf, real_t *q, len_t ii, len_t jj, len_t kk,
                                 int jb, int kb, int ncol, const int *cols, int xrow0, int xrow1);
/* one colour of the 7-point red-black sweep (pts = 0|1, BMG3_SymStd_relax_GS.f90:155-184) */
void cedar_amd_relax3_colour7(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                              int pts);
/* one phase (0 edges, 1 faces, 2 centres) of BMG3_SymStd_SETUP_interp_OI with lower loop bounds
 * ilo,jlo,klo (3 = serial / physical boundary, 2 = a neighbouring subdomain owns coarse point 1) */
void cedar_amd_setup_interp3_phase(real_t *so, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                   len_t iic, len_t jjc, len_t kkc, int ifd, int nstncl, int phase,
                                   int ilo, int jlo, int klo);

/* the 2D counterparts (src/2d/ftn/mpi/BMG2_SymStd_relax_GS.f90, ..._SETUP_interp_OI.f90): one row class of
 * the nine-point sweep (jb in {0,1}; efirst != 0: even 1-based i first = the 2D DOWN order), the column
 * fix-up after the x-neighbour's first colour arrived, one colour of the five-point sweep (jo in {2,3}),
 * one phase (0 edges, 1 centres) of the interpolation set-up with lower loop bounds ilo, jlo (3 | 2) */
void cedar_amd_relax2_pass(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int jb, int efirst);
void cedar_amd_relax2_fixup(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int icol, int jb);
void cedar_amd_relax2_colour5(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int jo);
void cedar_amd_setup_interp2_phase(real_t *so, real_t *ci, len_t iif, len_t jjf, len_t iic, len_t jjc,
                                   int ifd, int nstncl, int phase, int ilo, int jlo);
/* first-order recurrences over `nlines` line-contiguous vectors (leading dimension ld), in place, device
 * pointers: forward y_i = a_i y_{i-1} + c_i (reverse = 0) or backward y_i = a_i y_{i+1} + c_i (reverse = 1)
 * from a zero carry, c_i = y_i on entry (divided by div_i when div != NULL).  The two sweeps of DPTTRS per
 * line segment for the domain-decomposed line relaxation (src/2d/ftn/mpi/BMG2_SymStd_relax_lines_x.f90). */
void cedar_amd_affine_lines(real_t *y, const real_t *a, const real_t *div, int nlines, int n, int ld, int reverse);
/* around it (device pointers): right-hand sides b - (off-line part of A) x of the lines of zebra colour lb
 * (0-based interior parity) in direction dir (0 = x lines, 1 = y lines) into out[line][position]
 * (relax_lines_x.f90:104-109, relax_lines_y.f90:103-107); y[l][i] += p[l][i] * c[l] (the carry entering a
 * segment times the running product of its multipliers); and the solved lines back into q */
void cedar_amd_lines_rhs2(const real_t *so, const real_t *qf, const real_t *q, real_t *out, len_t ii, len_t jj,
                          int nstncl, int dir, int lb);
void cedar_amd_lines_carry(real_t *y, const real_t *p, const real_t *c, int nlines, int n, int ld);
void cedar_amd_lines_store2(const real_t *in, real_t *q, len_t ii, len_t jj, int dir, int lb);
/* pack (unpack = 0) / unpack (1) up to 26 sub-boxes of a device array (nplanes x kk x jj x ii) to /
 * from one contiguous device buffer in one launch: boxes = {i0,j0,k0,ni,nj,nk} per box (0-based incl.
 * ghost), offsets[b] = start of box b in the buffer in doubles per plane (box b occupies
 * offsets[b]*nplanes .. ).  Device pointers only. */
void cedar_amd_box_copy(real_t *arr, len_t ii, len_t jj, len_t kk, int nplanes, int nboxes,
                        const int *boxes, const unsigned long long *offsets, real_t *buf, int unpack);
/* the same with boxes = {i0,j0,k0,ni,nj,nk,sj,sk}: row j0 + j*sj, plane k0 + k*sk (the rows of one class / planes of one
 * k-parity of a face: what one stage of the boundary-first chain has changed) */
void cedar_amd_box_copy_strided(real_t *arr, len_t ii, len_t jj, len_t kk, int nplanes, int nboxes,
                                const int *boxes, const unsigned long long *offsets, real_t *buf, int unpack);

/* ------------------------------------------------------------------ 2. handle API */
typedef struct cedar_amd_solver cedar_amd_solver;

enum { CEDAR_AMD_RELAX_POINT = 0, CEDAR_AMD_RELAX_LINE_X = 1, CEDAR_AMD_RELAX_LINE_Y = 2,
       CEDAR_AMD_RELAX_LINE_XY = 3,
       /* 3D plane relaxation, include/cedar/multilevel.h:149-159 ("plane-xy" .. "plane-xyz", src/multilevel_settings.cc:10-13) */
       CEDAR_AMD_RELAX_PLANE_XY = 4, CEDAR_AMD_RELAX_PLANE_XZ = 5, CEDAR_AMD_RELAX_PLANE_YZ = 6, CEDAR_AMD_RELAX_PLANE_XYZ = 7 };

typedef struct {
	int relaxation;   /* solver.relaxation   (default point)  src/multilevel_settings.cc:15-28 */
	int nrelax_pre;   /* solver.cycle.nrelax-pre  (2)   :38 */
	int nrelax_post;  /* solver.cycle.nrelax-post (1)   :39 */
	int num_levels;   /* solver.num-levels (-1 = automatic) :40 */
	int max_iter;     /* solver.max-iter (10) :41 */
	double tol;       /* solver.tol (1e-8) :42 */
	int min_coarse;   /* solver.min_coarse (3) :43 */
	int cycle;        /* solver.cycle.type: 0 = "v" (default), 1 = "f" (include/cedar/cycle/fcycle.h:49-83) :30-36 */
	int ibc;          /* boundary code of BMG_get_bc(per_mask) from grid.periodic (src/kernel_params.cc): 0 definite,
	                     1 periodic in y, 2 in x, 3 in xy; 3D also 5 z, 6 xz, 7 yz, 8 xyz (V-cycle; 3D: even extents in
	                     the periodic directions on every level that is coarsened) */
	/* "plane-config" of plane relaxation (src/kernel_params.cc:72-78: line-xy, max-iter 1 unless the configuration
	 * carries its own block; the other keys default like the solver's) */
	int plane_relaxation, plane_nrelax_pre, plane_nrelax_post, plane_max_iter, plane_min_coarse;
	double plane_tol;
} cedar_amd_settings;

void cedar_amd_default_settings(cedar_amd_settings *s);

/* nd = 2|3; nstencil = 3|5 (2D five/nine point), 4|14 (3D seven/xxvii point).
 * `so` (host or device) is the fine operator in Cedar's layout; it is copied
 * unless own_device_so != 0 and so is a device pointer, in which case the
 * solver uses it in place and the caller must keep it alive (level 0 holds a
 * reference in the reference too, include/cedar/level.h:25,31). */
cedar_amd_solver *cedar_amd_solver_create(int nd, len_t nx, len_t ny, len_t nz, int nstencil,
                                          const real_t *so, int own_device_so,
                                          const cedar_amd_settings *settings);
void cedar_amd_solver_destroy(cedar_amd_solver *s);
int cedar_amd_solver_nlevels(const cedar_amd_solver *s);
void cedar_amd_solver_level_dims(const cedar_amd_solver *s, int lvl, len_t *nx, len_t *ny, len_t *nz);
/* copy a level array to the host for inspection: what = "A","P","SOR0","SOR1","ABD","res", and on coarse
 * levels "x","b" (level 0 works on the caller's x and b);
 * returns the number of doubles written (0 if absent); out may be NULL to query. */
size_t cedar_amd_solver_get(const cedar_amd_solver *s, int lvl, const char *what, real_t *out);
/* replace a set-up product of a level (what = "A","P","SOR0","SOR1","ABD"; `in` host or device, the array's full size):
 * the reference's `levels` container exposes these as public members (include/cedar/level.h:14-41).  Copies the solver
 * derives from the array are rebuilt.  Returns the number of doubles taken (0: no such array). */
size_t cedar_amd_solver_set(cedar_amd_solver *s, int lvl, const char *what, const real_t *in);
/* one V-cycle, cycle->run(x,b): x,b host or device */
void cedar_amd_solver_vcycle(cedar_amd_solver *s, real_t *x, const real_t *b);
/* multilevel::solve(b,x): rel[0] = ||r0||_2, rel[i] = ||r_i||_2/||r0||_2; returns cycles run */
int cedar_amd_solver_solve(cedar_amd_solver *s, const real_t *b, real_t *x, real_t *rel);
/* timing helper for benchmarks: n V-cycles back to back on device-resident x,b; returns
 * elapsed milliseconds measured with HIP events on the library's stream */
float cedar_amd_solver_time_vcycles(cedar_amd_solver *s, real_t *x_dev, const real_t *b_dev, int n);
/* n relax sweeps on level 0 alternating DOWN/UP (the roofline microbenchmark);
 * returns elapsed milliseconds (HIP events on the library's stream) */
float cedar_amd_solver_time_relax(cedar_amd_solver *s, real_t *x_dev, const real_t *b_dev, int n);
/* n launches of one level-0 kernel of the cycle besides the sweep: op 1 = residual, 2 = restriction of the residual to
 * level 1, 3 = interpolation-and-add from level 1 (overwrites x and the residual: a timing aid, not a solver step);
 * returns elapsed milliseconds (HIP events on the library's stream) */
float cedar_amd_solver_time_op(cedar_amd_solver *s, real_t *x_dev, const real_t *b_dev, int op, int n);

/* plane relaxation as a kernel of its own -- kernels::plane_relax<stypes, rdir>::setup(so) / run(so, x, b, dir)
 * (include/cedar/kernels/plane_relax.h:10-33; include/cedar/3d/relax_planes.h:164-246, src/3d/relax_planes.cc).
 * dir 0 = xy planes, 1 = xz, 2 = yz; plane_settings = the 2D solvers' configuration (NULL: the reference's default
 * plane-config, line-xy relaxation and one cycle per plane).  so / x / b host or device.  Like the reference, every
 * plane solver of a direction is built from the coefficients of the LAST plane (copy_coeff, relax_planes.h:80-160). */
typedef struct cedar_amd_planes cedar_amd_planes;
cedar_amd_planes *cedar_amd_planes_create(int dir, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                                          const cedar_amd_settings *plane_settings);
void cedar_amd_planes_run(cedar_amd_planes *p, const real_t *so, real_t *x, const real_t *b, int updown);
void cedar_amd_planes_destroy(cedar_amd_planes *p);

/* ------------------------------------------------------------------ 3. rank-to-rank transport (RCCL over xGMI)
 * What the reference's MPI flavour does through its MSG library and MPI collectives -- the ghost-layer exchange
 * after each colour / residual / interp_add (src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147,
 * src/3d/mpi/msg_exchanger.cc:188-197, src/2d/ftn/mpi/mpi_msg.F:425-550), the norm all-reduce
 * (include/cedar/3d/mpi/grid_func.h:41) and the coarse gather (include/cedar/3d/mpi/redist_solver.h:221-224) --
 * as RCCL calls issued by the library itself on its current stream (cedar_amd_set_stream).  One communicator per
 * process / GPU; the 128-byte unique id is made by rank 0 and handed to every rank by the launcher (any channel:
 * cedar_amd/comm.py uses a TCP socket).  librccl.so.1 is loaded on first use; all buffers are device pointers.
 * Functions returning int return 0 on success and report through print_error otherwise. */
#define CEDAR_AMD_COMM_ID_BYTES 128
typedef struct cedar_amd_comm cedar_amd_comm;
int cedar_amd_comm_available(void);                      /* 1 if librccl.so.1 could be loaded */
const char *cedar_amd_comm_why_unavailable(void);
int cedar_amd_comm_unique_id(void *id128);               /* ncclGetUniqueId */
cedar_amd_comm *cedar_amd_comm_create(const void *id128, int rank, int world);  /* ncclCommInitRank on the current device */
void cedar_amd_comm_destroy(cedar_amd_comm *c);
/* the launcher's part in compiled code, for hosts without Python or MPI: rank 0 makes the unique id and serves it over TCP
 * on MASTER_ADDR next to MASTER_PORT, the other ranks fetch it; the handshake carries a job tag (MASTER_PORT, world size,
 * CEDAR_AMD_RUN_ID / TORCHELASTIC_RUN_ID), the wire format is that of cedar_amd/comm.py.  _bootstrap = _bootstrap_id +
 * cedar_amd_comm_create; returns NULL / non-zero on failure (reported through print_error). */
int cedar_amd_comm_bootstrap_id(void *id128, int rank, int world);
cedar_amd_comm *cedar_amd_comm_bootstrap(int rank, int world);
int cedar_amd_comm_rank(const cedar_amd_comm *c);
int cedar_amd_comm_size(const cedar_amd_comm *c);
/* one grouped point-to-point exchange: ncclGroupStart; ncclRecv x nrecv; ncclSend x nsend; ncclGroupEnd (counts in doubles) */
int cedar_amd_comm_exchange(cedar_amd_comm *c, int nsend, const int *speer, const real_t *const *sbuf, const size_t *scount,
                            int nrecv, const int *rpeer, real_t *const *rbuf, const size_t *rcount);
int cedar_amd_comm_allreduce_sum(cedar_amd_comm *c, real_t *buf, size_t n);     /* in place */
int cedar_amd_comm_allreduce_max(cedar_amd_comm *c, real_t *buf, size_t n);
int cedar_amd_comm_allgather(cedar_amd_comm *c, const real_t *send, real_t *recv, size_t count); /* recv: world*count */
int cedar_amd_comm_broadcast(cedar_amd_comm *c, real_t *buf, size_t count, int root);
/* streams for the overlapped halo exchange: a non-blocking side stream; `waiter` waits for what is queued on `waited` now */
void *cedar_amd_stream_create(void);
void cedar_amd_stream_destroy(void *stream);
void cedar_amd_stream_wait(void *waiter, void *waited);
void cedar_amd_device_sync(void);
/* Releases the device scratch the library keeps between calls (the ring of row sums of the 3D Galerkin product: 4.4 GB
 * after a 512^3 set-up; it is re-allocated by the next product that needs it).  Waits for the device first. */
void cedar_amd_release_scratch(void);
/* HIP events on the library's current stream: record returns a new event; elapsed waits for e1 */
void *cedar_amd_event_record(void);
float cedar_amd_event_elapsed_ms(void *e0, void *e1);
void cedar_amd_event_destroy(void *event);

const char *cedar_amd_version(void);

/* ------------------------------------------------------------------ 4. the domain-decomposed 3D solver (one rank per GPU)
 * cdr3::mpi::solver of the reference (include/cedar/3d/mpi/solver.h:76-89,232-233) for Dirichlet problems, point
 * relaxation, V(pre,post): Cartesian block decomposition with one ghost layer, global-parity colouring, a ghost
 * exchange after every row class of a sweep / residual / interp_add and between the phases of the set-up
 * (src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147, ..._residual.f90:130, ..._interp_add.f90:308,
 * ..._SETUP_interp_OI.f90:418-1074, ..._SETUP_ITLI27_ex.f90:1803), norm all-reduce (3d/mpi/grid_func.h:41), coarse levels
 * gathered onto every rank (in place of 3d/mpi/redist_solver.h:221-224).  The whole cycle is orchestrated below this ABI
 * (cedar_amd/csrc/dist3.cpp); a rank process only creates the handle and calls it.  rank = k*px*py + j*px + i
 * (src/3d/util/topo.cc:82-84); every local extent must stay even on the distributed levels.
 * Transport: `comm` (section 3, RCCL) -- or `transport`, a caller-supplied table, the counterpart of the reference's
 * halo_exchanger plug-in (include/cedar/kernel.h:25-37, kernel_manager::add_halo): exchange has the contract of
 * cedar_amd_comm_exchange (device buffers, ordered with the library's current stream), allgather that of
 * cedar_amd_comm_allgather, allreduce_sum sums n HOST doubles in place over the ranks.  Each returns 0 on success. */
typedef struct {
	void *ctx;
	int (*exchange)(void *ctx, int nsend, const int *speer, const real_t *const *sbuf, const size_t *scount,
	                int nrecv, const int *rpeer, real_t *const *rbuf, const size_t *rcount);
	int (*allgather)(void *ctx, const real_t *send, real_t *recv, size_t count);
	int (*allreduce_sum)(void *ctx, double *host_values, int n);
} cedar_amd_transport;
/* a transport that talks to itself (device-to-device copies in place of send / receive): one rank of a `world`-rank grid
 * on a single GPU, for measuring what a rank costs beside its messages; ghost values are meaningless */
void cedar_amd_transport_loopback(cedar_amd_transport *out, int world);
typedef struct cedar_amd_dist3 cedar_amd_dist3;
/* the rank grid used when pgrid is NULL: 1x1xN z slabs up to 4 ranks, 2x2x2 for 8 (BASELINE config 5), else the most
 * cubic factorisation */
void cedar_amd_dist3_rank_grid(int world, int pgrid[3]);
/* A_local: DEVICE array (nstencil, nz+2, ny+2, nx+2) of this rank's part of the global operator (entries coupling to a
 * neighbouring rank present; its ghost layers are filled here); it must outlive the handle.  settings: nrelax_pre /
 * nrelax_post / max_iter / tol / min_coarse are used.  agglomerate_below (0 = 64): a level with at most that many
 * points per direction and rank is gathered and continued on the single-domain solver.  overlap_min (0 = 96): levels
 * with at least that many points per direction run the y/z halo of a row class on a side stream under the interior
 * rows of the next.  Collective: every rank of the communicator calls it. */
cedar_amd_dist3 *cedar_amd_dist3_create(cedar_amd_comm *comm, const cedar_amd_transport *transport, int rank, int world,
                                        const int pgrid[3], real_t *A_local, len_t nx, len_t ny, len_t nz, int nstencil,
                                        const cedar_amd_settings *settings, int agglomerate_below, int overlap_min);
void cedar_amd_dist3_destroy(cedar_amd_dist3 *d);
int cedar_amd_dist3_nlevels(const cedar_amd_dist3 *d);            /* levels of the global hierarchy */
int cedar_amd_dist3_distributed_levels(const cedar_amd_dist3 *d); /* of which this many are distributed (the last one gathered) */
/* distributed levels of a rank grid with an x / y split that relax with the partial-sum sweep behind a boundary-first
 * chain (cedar_amd_relax3_planes_masked; CEDAR_AMD_DIST_CHAIN=0: none, the reference-order row-class passes everywhere) */
int cedar_amd_dist3_chain_levels(const cedar_amd_dist3 *d);
/* one V-cycle on this rank's device-resident x, b (local boxes incl. ghost layer) */
void cedar_amd_dist3_vcycle(cedar_amd_dist3 *d, real_t *x_dev, real_t *b_dev);
/* mpi::solver::solve: rel[0] = ||r0||_2, rel[i] = ||r_i||_2 / ||r0||_2 (global norms); returns the cycles run */
int cedar_amd_dist3_solve(cedar_amd_dist3 *d, real_t *b_dev, real_t *x_dev, real_t *rel);
/* n level-0 sweeps (alternating DOWN / UP) with their halo exchanges; elapsed ms by HIP events */
float cedar_amd_dist3_time_relax(cedar_amd_dist3 *d, real_t *x_dev, real_t *b_dev, int n);

/* ------------------------------------------------------------------ 4b. the domain-decomposed 2D solver
 * cdr2::mpi::solver of the reference (include/cedar/2d/mpi/solver.h) for Dirichlet problems on a px x py rank grid
 * (rank = j*px + i), V(pre,post), point relaxation or zebra line relaxation in x, y or both with the lines cut by the
 * ranks of a row / column of the grid -- the reference's distributed tridiagonal solves
 * (src/2d/ftn/mpi/BMG2_SymStd_relax_lines_x.f90:163-307, include/cedar/2d/mpi/ml_relax.h) as a chain of segments whose
 * carries the ranks of a line hand each other (cedar_amd/csrc/dist_lines.hip).  Orchestrated below this ABI
 * (cedar_amd/csrc/dist2.cpp), no torch in a rank process; transport and arguments as for cedar_amd_dist3_create
 * (settings->relaxation selects the smoother).  A_local: device array (nstencil = 3 | 5, ny+2, nx+2). */
typedef struct cedar_amd_dist2 cedar_amd_dist2;
void cedar_amd_dist2_rank_grid(int world, int pgrid[2]);   /* 1x1, 1x2, 2x2, 2x4: y is split first */
cedar_amd_dist2 *cedar_amd_dist2_create(cedar_amd_comm *comm, const cedar_amd_transport *transport, int rank, int world,
                                        const int pgrid[2], real_t *A_local, len_t nx, len_t ny, int nstencil,
                                        const cedar_amd_settings *settings, int agglomerate_below);
void cedar_amd_dist2_destroy(cedar_amd_dist2 *d);
int cedar_amd_dist2_nlevels(const cedar_amd_dist2 *d);
void cedar_amd_dist2_vcycle(cedar_amd_dist2 *d, real_t *x_dev, real_t *b_dev);
int cedar_amd_dist2_solve(cedar_amd_dist2 *d, real_t *b_dev, real_t *x_dev, real_t *rel);
float cedar_amd_dist2_time_relax(cedar_amd_dist2 *d, real_t *x_dev, real_t *b_dev, int n);

/* Cedar's C interface (include/cedar/capi.h: bmg2_* / bmg3_*) with nprocx * nprocy [* nprocz] > 1 runs on the two drivers
 * above, as the reference runs it on its MPI solvers (src/2d/interface/c/solver.cc:10-60).  The MPI_Comm argument is not
 * dereferenced: the rank comes from the launcher's environment (RANK / PMI_RANK / OMPI_COMM_WORLD_RANK / SLURM_PROCID and
 * the matching size variables) or from cedar_amd_bmg_set_rank; the transport is an RCCL communicator the interface
 * bootstraps itself (cedar_amd_comm_bootstrap) unless a table is handed in. */
void cedar_amd_bmg_set_rank(int rank, int world);
void cedar_amd_bmg_set_transport(const cedar_amd_transport *tp);   /* NULL: back to RCCL */

#ifdef __cplusplus
}
#endif
#endif
