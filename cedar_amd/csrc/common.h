// cedar_amd -- MI355X-native BoxMG V-cycle hot path.
// Shared declarations for the HIP kernels (gfx950 only) and the C-ABI.
//
// Data layout (identical to the reference so that the drop-in boundary needs
// no conversion): FP64, Fortran order (first index fastest), one ghost layer on
// every side, II = nx+2 ...; stencil operators are SoA "planes", slot s at
// offset s*II*JJ[*KK] (include/cedar/array.h:67-74, stencil_op_nd.h:41-78 of
// the reference).  All device code is compiled with -ffp-contract=off and keeps
// the reference's term order, so the solve-phase kernels round exactly like
// the reference's FMA-free CPU build.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace cedar_amd {

typedef double real_t;
typedef unsigned int len_t;

enum { BMG_DOWN = 0, BMG_UP = 1 };
// 2D slots (src/2d/ftn/BMG_stencils_f90.h:29-37, 0-based)
enum { KO = 0, KW = 1, KS = 2, KSW = 3, KNW = 4 };
enum { LL = 0, LR = 1, LA = 2, LB = 3, LSW = 4, LNW = 5, LNE = 6, LSE = 7 };
// 3D slots (:43-64)
enum { KP = 0, KPW = 1, KPS = 2, KB = 3, KPSW = 4, KPNW = 5, KBW = 6, KBNW = 7,
       KBN = 8, KBNE = 9, KBE = 10, KBSE = 11, KBS = 12, KBSW = 13 };
enum { LXYL = 0, LXYR = 1, LXYA = 2, LXYB = 3, LXZA = 4, LXZB = 5,
       LXYNE = 6, LXYSE = 7, LXYSW = 8, LXYNW = 9, LXZSW = 10, LXZNW = 11,
       LXZNE = 12, LXZSE = 13, LYZSW = 14, LYZNW = 15, LYZNE = 16, LYZSE = 17,
       LBSW = 18, LBNW = 19, LBNE = 20, LBSE = 21,
       LTSW = 22, LTNW = 23, LTNE = 24, LTSE = 25 };

#define CEDAR_HIP_CHECK(expr)                                                         \
	do {                                                                              \
		hipError_t e_ = (expr);                                                       \
		if (e_ != hipSuccess) {                                                       \
			fprintf(stderr, "[cedar_amd] HIP error %s at %s:%d: %s\n",                \
			        hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_)); \
			abort();                                                                  \
		}                                                                             \
	} while (0)

// 8-byte-aligned pair of doubles: rows start at arbitrary parity, the hardware
// handles the unaligned dwordx4.
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

// View of a 3D operator for the solve-phase kernels (27-point relax / residual): entry (slot, i, j, k) =
// so[slot*SS + j*SJ + k*SK + i], reciprocal diagonal (SOR plane msor) = sor[j*rSJ + k*rSK + i].
//   Cedar layout (the boundary):  SS = II*JJ*KK, SJ = II, SK = II*JJ; sor = SOR + II*JJ*KK, same strides.
//   row-interleaved copy (solver-internal, op3_ilv): the NS3 = 16 slot-rows of grid row (j,k) are contiguous
//   (14 operator slots, 1/diag, one pad row), rows padded to RS doubles (a multiple of 16 => every slot-row
//   starts on a 128-byte line): SS = RS, SJ = 16*RS, SK = 16*RS*JJ, sor = so + 14*RS.
struct Op3 {
	const real_t *so;
	size_t SS, SJ, SK;
	const real_t *sor;
	size_t rSJ, rSK;
};
enum { NS3 = 16, ILV_SOR = 14 };
static inline Op3 op3_cedar(const real_t *so, const real_t *sor, int II, int JJ, int KK)
{
	const size_t sk = (size_t)II * JJ, PS = sk * (size_t)KK;
	return Op3{so, PS, (size_t)II, sk, sor ? sor + PS : nullptr, (size_t)II, sk};
}
static inline size_t ilv_row_stride(int II) { return ((size_t)II + 15) / 16 * 16; }
static inline size_t ilv_doubles(int II, int JJ, int KK) { return ilv_row_stride(II) * NS3 * (size_t)JJ * KK; }
static inline Op3 op3_ilv(const real_t *a, int II, int JJ, int KK)
{
	const size_t RS = ilv_row_stride(II);
	(void)KK;
	return Op3{a, RS, NS3 * RS, NS3 * RS * (size_t)JJ, a + ILV_SOR * RS, NS3 * RS, NS3 * RS * (size_t)JJ};
}
// register / drop a row-interleaved solve copy for an operator outside a resident solver (relax3d.hip); prepare
// returns 1 when a copy was built (levels with at least min_rows rows that fit the card's free memory)
int relax3_prepare(const real_t *so, const real_t *sor, int II, int JJ, int KK, int min_rows, hipStream_t st);
void relax3_release(const real_t *so);
// build the row-interleaved copy from the Cedar-layout operator (14 slots) and 1/diag plane (relax3d.hip)
void ilv_build(const real_t *so, const real_t *sor_msor, real_t *ilv, int II, int JJ, int KK, hipStream_t st);

// A batch of independent problems on ONE operator (plane relaxation: the planes of one colour, solver.cpp): the 2D
// solve-phase kernels take the batch item from blockIdx.y (row kernels with a one-dimensional grid) or blockIdx.z and
// offset their VECTOR arguments by `stride` doubles per item; operator arrays are shared.  n = 1: a single problem.
struct Batch {
	int n = 1;
	size_t stride = 0;
};

// ---- kernel launchers (device pointers, asynchronous on `st`) ----
// relax3d.hip
void setup_recip(const real_t *so_diag, real_t *sor_msor, size_t II, size_t JJ, size_t KK, hipStream_t st);
void relax3_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
               int II, int JJ, int KK, int nstncl, int updown, hipStream_t st);
// 27-point sweep / residual on an operator view (the solver's row-interleaved copy)
// T: scratch of the vector's size for the partial-sum sweep (relax3d_psum.hip), nullptr = reference order always
void relax3_gs27_op(const Op3 &A, const real_t *qf, real_t *q, int II, int JJ, int KK, int updown, hipStream_t st,
                    real_t *T = nullptr);
void residual27_op(const Op3 &A, const real_t *qf, const real_t *q, real_t *res, int II, int JJ, int KK, hipStream_t st);
void relax3_planes27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                     int II, int JJ, int KK, int kb, int up, int part, hipStream_t st);
void relax3_pass27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int KK, int jb, int kb, int efirst, hipStream_t st, int part = 0);
void relax3_fixup27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                    int II, int JJ, int KK, int icol, int jb, int kb, hipStream_t st);
void relax3_colour7(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                    int II, int JJ, int KK, int pts, hipStream_t st);
// relax2d.hip
void relax2_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
               int II, int JJ, int nstncl, int updown, hipStream_t st, Batch bt = Batch());
// residual.hip
void residual2(const real_t *so, const real_t *qf, const real_t *q, real_t *res,
               int II, int JJ, int nstncl, hipStream_t st, Batch bt = Batch());
void residual3(const real_t *so, const real_t *qf, const real_t *q, real_t *res,
               int II, int JJ, int KK, int nstncl, hipStream_t st);
// pieces of the 2D sweep / set-up for domain-decomposed runs (relax2d.hip, setup_interp.hip)
void relax2_pass9(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                  int II, int JJ, int jb, int efirst, hipStream_t st);
void relax2_fixup9(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int icol, int jb, hipStream_t st);
void relax2_colour5(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                    int II, int JJ, int jo, hipStream_t st);
void setup_interp2_phase(const real_t *so, real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, int phase,
                         int ilo, int jlo, hipStream_t st);
void lines_rhs2(const real_t *so, const real_t *qf, const real_t *q, real_t *out, int II, int JJ, int nstncl, int dir, int lb,
                hipStream_t st);
void lines_store2(const real_t *in, real_t *q, int II, int JJ, int dir, int lb, hipStream_t st);
void lines_carry(real_t *y, const real_t *p, const real_t *c, int nlines, int n, int ld, hipStream_t st);
void affine_lines(real_t *y, const real_t *a, const real_t *div, int nlines, int n, int ld, int reverse, hipStream_t st);
// 2D periodic boundary conditions (periodic2d.hip); ipn = 1 per_y, 2 per_x, 3 per_xy
void wrap2(real_t *q, int II, int JJ, int nplanes, int do_y, int do_x, hipStream_t st);
int relax2_gs_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                  int II, int JJ, int nstncl, int updown, int ipn, hipStream_t st); // 1 = rows too long
void restrict2_per(real_t *q, real_t *qc, const real_t *ci, int Nx, int Ny, int Nxc, int Nyc, int ipn, hipStream_t st);
void interp_add2_per(real_t *q, const real_t *qc, real_t *res, const real_t *so, const real_t *ci,
                     int IIC, int JJC, int IIF, int JJF, int ipn, hipStream_t st);
void galerkin2_per(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, int ipn,
                   hipStream_t st);
void setup_interp2_per(const real_t *so, real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, int ipn, hipStream_t st);
void setup_cg2_per(const real_t *so, int II, int JJ, int nstncl, real_t *abd, int nabd1, int ipn, int *info, hipStream_t st);
void solve_cg2_per(real_t *q, const real_t *qf, int II, int JJ, const real_t *abd, real_t *bbd, int nabd1, int ipn, hipStream_t st);
// 3D periodic boundary conditions (periodic3d.hip, relax3d.hip); ipn = 1 y, 2 x, 3 xy, 5 z, 6 xz, 7 yz, 8 xyz
bool periodic3_code_ok(int ipn);
void wrap3(real_t *a, int II, int JJ, int KK, int narrays, int ipn, hipStream_t st);
void wrap3_colour(real_t *q, int II, int JJ, int KK, int jb, int kb, int ipn, hipStream_t st);
void wrap3_sweep_end(real_t *q, int II, int JJ, int KK, int ipn, hipStream_t st);
void relax3_gs_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int KK, int nstncl, int updown, int ipn, hipStream_t st, int ghosts_consistent = 0);
void restrict3_per(real_t *q, real_t *qc, const real_t *ci, int II, int JJ, int KK, int IIC, int JJC, int KKC, int ipn,
                   hipStream_t st);
void interp_add3_per(real_t *q, const real_t *qc, const real_t *so, real_t *res, const real_t *ci,
                     int IIC, int JJC, int KKC, int IIF, int JJF, int KKF, int ipn, hipStream_t st);
void setup_interp3_per(const real_t *so, real_t *ci, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int ifd, int ipn,
                       hipStream_t st);
void galerkin3_per(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC,
                   int ifd, int ipn, hipStream_t st);
void setup_cg3_per(const real_t *so, int II, int JJ, int KK, real_t *abd, int nabd1, int ipn, int *info, hipStream_t st);
void solve_cg3_per(real_t *q, const real_t *qf, int II, int JJ, int KK, const real_t *abd, real_t *bbd, int nabd1, int ipn,
                   hipStream_t st);
// plane relaxation, the 3D side (planes.hip); dir 0 xy, 1 xz, 2 yz; stacked 2D arrays, slot q = plane beg + 2q
void plane_operator(int dir, int nst, const real_t *so, real_t *so2, int II, int JJ, int KK, hipStream_t st);
void plane_gather(int dir, int nst, const real_t *so, const real_t *x, const real_t *b, real_t *x2s, real_t *b2s,
                  int II, int JJ, int KK, int beg, int nslots, hipStream_t st);
void plane_scatter(int dir, const real_t *x2s, real_t *x, int II, int JJ, int KK, int beg, int nslots, hipStream_t st);
// qf = A q (operator application with Cedar's sign convention), residual.hip
void matvec2(const real_t *so, const real_t *q, real_t *qf, int II, int JJ, int nstncl, hipStream_t st);
void matvec3(const real_t *so, const real_t *q, real_t *qf, int II, int JJ, int KK, int nstncl, hipStream_t st);
// sum of squares over the interior -> *out (deterministic two-stage tree); scratch >= 4096 doubles
void sumsq_interior(const real_t *v, int II, int JJ, int KK, real_t *scratch, real_t *out, hipStream_t st);
// transfer.hip
void restrict2(const real_t *q, real_t *qc, const real_t *ci, int II, int JJ, int IIC, int JJC, hipStream_t st,
               Batch bf = Batch(), Batch bc = Batch()); // batch strides of the fine / coarse vectors
void restrict3(const real_t *q, real_t *qc, const real_t *ci, int II, int JJ, int KK,
               int IIC, int JJC, int KKC, hipStream_t st);
void interp_add2(real_t *q, const real_t *qc, real_t *res, const real_t *so, const real_t *ci,
                 int IIC, int JJC, int IIF, int JJF, hipStream_t st, Batch bf = Batch(), Batch bc = Batch());
void interp_add3(real_t *q, const real_t *qc, const real_t *so, real_t *res, const real_t *ci,
                 int IIC, int JJC, int KKC, int IIF, int JJF, int KKF, hipStream_t st);
void box_copy(real_t *arr, int II, int JJ, int KK, int nplanes, int nboxes, const int *boxes,
              const unsigned long long *offsets, real_t *buf, int unpack, hipStream_t st, int strided = 0);
// setup_interp.hip
void setup_interp2(const real_t *so, real_t *ci, int IIF, int JJF, int IIC, int JJC, int ifd, hipStream_t st);
void setup_interp3(const real_t *so, real_t *ci, int IIF, int JJF, int KKF,
                   int IIC, int JJC, int KKC, int ifd, hipStream_t st);
void setup_interp3_phase(const real_t *so, real_t *ci, int IIF, int JJF, int KKF,
                         int IIC, int JJC, int KKC, int ifd, int phase, int ilo, int jlo, int klo, hipStream_t st);
// galerkin.hip
void galerkin2(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int IIC, int JJC,
               int ifd, hipStream_t st);
void galerkin3_fused27(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                       int IIC, int JJC, int KKC, hipStream_t st);
void galerkin3_fused7(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                      int IIC, int JJC, int KKC, hipStream_t st);
bool galerkin3_tiled(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                     int IIC, int JJC, int KKC, int ifd, hipStream_t st);
bool galerkin3_rows(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                    int IIC, int JJC, int KKC, int ifd, hipStream_t st);
void galerkin3_rows_release(); // frees the kept ring of row sums
bool galerkin3_rows_pairs(const real_t *so, int IIF); // the row sums can read the operator as aligned pairs
void galerkin3(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
               int IIC, int JJC, int KKC, int ifd, hipStream_t st);
// util.hip: zero fill by a kernel of the library.  hipMemsetAsync is not used on solver data: under the HIP runtime
// that PyTorch loads first (the multi-GPU path always imports torch) it left non-zero bit patterns in the cleared
// arrays (profiles/r01_memset_under_torch_runtime.log)
void zero_fill(real_t *p, size_t n, hipStream_t st);
// lines.hip
void setup_lines_x(const real_t *so, real_t *sor, int II, int JJ, hipStream_t st, int fold = 0);
void setup_lines_y(const real_t *so, real_t *sor, int II, int JJ, hipStream_t st, int fold = 0);
void relax_lines_x(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int nstncl, int updown, hipStream_t st, int ipn = 0, const real_t *pf = nullptr,
                   Batch bt = Batch()); // batches: Dirichlet lines only
// scan-ordered copy of the line factors for the resident solver (lines.hip line_pttrs_pf); 0 doubles = not used
size_t lines_permuted_doubles(int n, int nlines);
void lines_permute(const real_t *sor, real_t *pf, int n, int ld, int nlines, size_t PS, hipStream_t st);
// scratch: line-contiguous buffer of ylines_scratch_doubles(II,JJ) doubles in HBM
size_t ylines_scratch_doubles(int II, int JJ);
void relax_lines_y(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, real_t *scratch,
                   int II, int JJ, int nstncl, int updown, hipStream_t st, int ipn = 0);
// y-lines on transposed arrays (lines.hip): transposed operator planes, transposed right-hand side, scratch for q^T
void transpose2(const real_t *in, real_t *out, int II, int JJ, hipStream_t st, Batch bt = Batch());
// lines_small.hip: all line sweeps of one visit of a level of at most 64 x 64 unknowns in one launch
bool lines_small_ok(int II, int JJ);
// one half of a V-cycle visit of such a level in one launch: pre != 0: pre-smoothing, residual, restriction, coarse x := 0;
// else interpolation-and-add, post-smoothing.  kind: 0 point relaxation (sorx = reciprocals), 1 / 2 / 3 = x / y / xy lines
void visit_small(int pre, const real_t *so, const real_t *qf, real_t *q, real_t *res, const real_t *sorx, const real_t *sory,
                 int II, int JJ, int nstncl, int kind, int nsweeps, const real_t *ci, real_t *cb, real_t *cx, int IIC, int JJC,
                 hipStream_t st, Batch bf = Batch(), Batch bc = Batch());
void relax_points_small(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int nstncl,
                        int updown, int nsweeps, hipStream_t st, Batch bt = Batch());
void relax_lines_small(const real_t *so, const real_t *qf, real_t *q, const real_t *sorx, const real_t *sory,
                       int II, int JJ, int nstncl, int kind, int updown, int nsweeps, hipStream_t st, Batch bt = Batch());
void setup_lines_yt(const real_t *so, real_t *sot, int II, int JJ, int nstncl, hipStream_t st);
void relax_lines_yt(const real_t *sot, const real_t *qft, real_t *q, real_t *qt, const real_t *sor,
                    int II, int JJ, int nstncl, int updown, hipStream_t st, const real_t *pf = nullptr, Batch bt = Batch());
// cgsolve.hip
void setup_cg2(const real_t *so, int II, int JJ, int nstncl, real_t *abd, int nabd1, int nabd2, int *info, hipStream_t st);
void solve_cg2(real_t *q, const real_t *qf, int II, int JJ, const real_t *abd, real_t *bbd, int nabd1, int nabd2, hipStream_t st,
               Batch bt = Batch()); // bbd: nabd2 doubles per batch item
void setup_cg3(const real_t *so, int II, int JJ, int KK, int nstncl, real_t *abd, int nabd1, int nabd2, int *info, hipStream_t st);
void solve_cg3(real_t *q, const real_t *qf, int II, int JJ, int KK, const real_t *abd, real_t *bbd, int nabd1, int nabd2, hipStream_t st);
// gallery.hip (device-side generators of the reference's gallery operators)
void gallery_fill(int which, real_t *so, real_t *b, int nx, int ny, int nz, const double *params, hipStream_t st);

// XCD-aware block remap: consecutive logical blocks land on the same XCD
// (blocks b and b+8 share an XCD under round-robin dispatch; speed only).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk)
{
	unsigned chunk = (nblk + 7u) >> 3;
	return (b & 7u) * chunk + (b >> 3);
}
static inline unsigned xcd_grid(unsigned nblk) { return ((nblk + 7u) >> 3) << 3; }

// Row (jr,kr) owned by logical workgroup L: rows are walked in 2^tjl x 2^tkl (j,k) tiles so that
// the ~64 workgroups an XCD has in flight form a compact patch of rows and find each other's rows
// (q rows j+-1,k+-1; operator rows stored at j+1 / k+1) in that XCD's L2 instead of HBM.
// tkl = 0 degenerates to the plain j-fastest walk.
struct TileShape { unsigned tjl, tkl; };
__device__ __forceinline__ bool tile_rows(unsigned L, unsigned nj, unsigned nk, TileShape ts, unsigned &jr, unsigned &kr)
{
	const unsigned ntj = (nj + (1u << ts.tjl) - 1u) >> ts.tjl;
	const unsigned tile = L >> (ts.tjl + ts.tkl), w = L & ((1u << (ts.tjl + ts.tkl)) - 1u);
	jr = ((tile % ntj) << ts.tjl) + (w & ((1u << ts.tjl) - 1u));
	kr = ((tile / ntj) << ts.tkl) + (w >> ts.tjl);
	return jr < nj && kr < nk;
}
__host__ __device__ static inline unsigned tile_blocks(unsigned nj, unsigned nk, TileShape ts)
{
	return (((nj + (1u << ts.tjl) - 1u) >> ts.tjl) * ((nk + (1u << ts.tkl) - 1u) >> ts.tkl)) << (ts.tjl + ts.tkl);
}
// default shapes (measured on MI355X, profiles/): overridable for experiments with
// CEDAR_AMD_TILE_RELAX="tjl,tkl" / CEDAR_AMD_TILE_RESID="tjl,tkl"
TileShape tile_shape_relax();
TileShape tile_shape_resid();

} // namespace cedar_amd
