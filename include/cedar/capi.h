/* Cedar's C interface (bmg2_* / bmg3_* / bmg_timer_save) on the MI355X library.
 *
 * Same entry points, argument meaning and value conventions as the reference's
 * include/cedar/capi.h and the headers it pulls in
 *   include/cedar/{2d,3d}/interface/c/{topo,operator,solver}.h, include/cedar/interface/c/timer.h,
 *   include/cedar/{2d,3d}/base_types.h
 * implemented in cedar_amd/csrc/bmg_capi.cpp from the behaviour of
 *   src/{2d,3d}/interface/c/{topo,operator,solver}.cc, src/interface/c/timer.cc.
 * A C caller written against the reference links unchanged against libcedar_amd.so.
 *
 * Scope: the reference implements this interface on its MPI solver.  This library is one process per GPU: a topology
 * with nprocx * nprocy [* nprocz] = 1 runs on the device-resident solver, a larger one on the domain-decomposed
 * drivers (cedar_amd_dist2_* / cedar_amd_dist3_*, include/cedar_amd.h section 4) -- every rank must then own the same
 * local extents, even on every level that stays distributed.  The communicator argument is accepted and not
 * dereferenced (the library links no MPI): the rank comes from the launcher's environment (RANK / PMI_RANK /
 * OMPI_COMM_WORLD_RANK / SLURM_PROCID with the matching size variables) or cedar_amd_bmg_set_rank, the transport is an
 * RCCL communicator the interface bootstraps over MASTER_ADDR / MASTER_PORT, or a table handed in with
 * cedar_amd_bmg_set_transport.  A process grid that does not match the launched ranks reports through print_error()
 * and yields NULL.  As in the reference a rank sets the entries whose STORAGE location lies in its local array, ghost
 * layer included (coordinates of ghost vertices are taken).
 *
 * Conventions kept from the reference:
 *  - grid coordinates are 0-based global vertex indices; the operator is given vertex
 *    based with the usual signs (negative off-diagonals); bmgN_operator_set flips the
 *    sign of the off-diagonal entries IN THE CALLER'S `vals` ARRAY (the reference does)
 *    and moves E/N/NE/SE/NW entries to the neighbour that stores them in BoxMG's
 *    symmetric layout (src/2d/interface/c/operator.cc:29-66).  In 3D `dir` already names
 *    a storage slot and no such move is made (src/3d/interface/c/operator.cc:27-49);
 *  - x and b of bmgN_operator_apply / bmgN_solver_run are interior-only arrays, i fastest;
 *  - bmgN_solver_create reads ./config.json (solver.* keys, src/multilevel_settings.cc:11-44),
 *    bmgN_solver_run starts from x = 0 and runs multilevel::solve (max-iter / tol).
 */
#ifndef CEDAR_CAPI_H
#define CEDAR_CAPI_H

/* MPI_Comm: use the caller's <mpi.h> when it was included first, else an int stand-in
 * (MPICH ABI); the value is never used on the single-rank path. */
#if !defined(MPI_VERSION) && !defined(CEDAR_AMD_MPI_COMM_DEFINED)
#define CEDAR_AMD_MPI_COMM_DEFINED
typedef int MPI_Comm;
#define MPI_COMM_WORLD ((MPI_Comm)0x44000000)
#define MPI_COMM_SELF ((MPI_Comm)0x44000001)
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* include/cedar/2d/base_types.h:4-14 (storage slots 0..4 = C, W, S, SW, NW) */
typedef enum { BMG2_C = 0, BMG2_W = 1, BMG2_S = 2, BMG2_SW = 3, BMG2_NW = 4,
               BMG2_SE = 5, BMG2_N = 6, BMG2_NE = 7, BMG2_E = 8 } bmg2_dir;
/* include/cedar/3d/base_types.h:5-20 */
typedef enum { BMG3_P = 0, BMG3_PW = 1, BMG3_PS = 2, BMG3_B = 3, BMG3_PSW = 4, BMG3_PNW = 5,
               BMG3_BW = 6, BMG3_BNW = 7, BMG3_BN = 8, BMG3_BNE = 9, BMG3_BE = 10,
               BMG3_BSE = 11, BMG3_BS = 12, BMG3_BSW = 13 } cdr3_dir;

/* ---- topology: include/cedar/2d/interface/c/topo.h:13-22, 3d/.../topo.h:13-25 */
typedef struct bmg2_topology *bmg2_topo;
typedef struct bmg3_topology *bmg3_topo;
bmg2_topo bmg2_topo_create(MPI_Comm comm, unsigned int ngx, unsigned int ngy,
                           unsigned int lnx[], unsigned int lny[], int nprocx, int nprocy);
bmg3_topo bmg3_topo_create(MPI_Comm comm, unsigned int ngx, unsigned int ngy, unsigned int ngz,
                           unsigned int lnx[], unsigned int lny[], unsigned int lnz[],
                           int nprocx, int nprocy, int nprocz);

/* ---- operator: include/cedar/2d/interface/c/operator.h:12-30, 3d/.../operator.h:12-31 */
typedef struct bmg2_op *bmg2_operator;
typedef struct bmg3_op *bmg3_operator;
typedef struct { unsigned int i, j; bmg2_dir dir; } grid_coord_2d;
typedef struct { unsigned int i, j, k; cdr3_dir dir; } grid_coord_3d;

bmg2_operator bmg2_operator_create(bmg2_topo topo);
void bmg2_operator_set(bmg2_operator, unsigned int nvals, grid_coord_2d coords[], double vals[]);
void bmg2_operator_apply(bmg2_operator, const double *x, double *b); /* b = A x */
void bmg2_operator_dump(bmg2_operator);                              /* writes op<cx>-<cy>.txt */
void bmg2_operator_destroy(bmg2_operator);

bmg3_operator bmg3_operator_create(bmg3_topo topo);
void bmg3_operator_set(bmg3_operator, unsigned int nvals, grid_coord_3d coords[], double vals[]);
void bmg3_operator_apply(bmg3_operator, const double *x, double *b);
void bmg3_operator_dump(bmg3_operator);
void bmg3_operator_destroy(bmg3_operator);

/* ---- solver: include/cedar/2d/interface/c/solver.h:10-15, 3d/.../solver.h */
typedef struct bmg2_slv *bmg2_solver;
typedef struct bmg3_slv *bmg3_solver;
bmg2_solver bmg2_solver_create(bmg2_operator *op);
void bmg2_solver_run(bmg2_solver, double *x, const double *b);
void bmg2_solver_destroy(bmg2_solver);
bmg3_solver bmg3_solver_create(bmg3_operator *op);
void bmg3_solver_run(bmg3_solver, double *x, const double *b);
void bmg3_solver_destroy(bmg3_solver);

/* ---- timer: include/cedar/interface/c/timer.h:9 */
void bmg_timer_save(const char *fname);

#ifdef __cplusplus
}
#endif
#endif
