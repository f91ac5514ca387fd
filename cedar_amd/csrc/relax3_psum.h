// Host-side entry points of the 27-point sweep variants (relax3d.hip, relax3d_psum.hip).
#pragma once
#include "common.h"

namespace cedar_amd {

// F rows per workgroup of the plane-fused walk on a level with JJ-2 rows (0 = row-class launches), relax3d.hip
int relax3_plane_frun(int JJ);
// relax3d_psum.hip: 27-point sweep with inter-plane partial sums (north_star's 1e-10 contract, not bit for bit)
bool relax3_psum_ok(int II, int JJ, int KK, int frun);
bool relax3_psum_wanted(int II, int JJ, int KK); // the level takes it by default (CEDAR_AMD_PSUM, run length, row length)
void relax3_gs27_psum(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int updown, int frun,
                      hipStream_t st);
// Points of a k-parity of planes that a launch leaves as they are: relaxed ahead of it by the boundary-first chain of a
// rank grid with an x / y split (dist3.cpp).  Per row class of the walk a mask over the first and the last four points
// of a row (relax27_dev.h skip27_lane), and up to three whole rows.
struct PsumSkip {
	unsigned colsF, colsS; // F rows / S rows of the walk
	int rows[3];           // -1: unused
};
static inline PsumSkip psum_skip_none()
{
	PsumSkip s;
	s.colsF = s.colsS = 0; s.rows[0] = s.rows[1] = s.rows[2] = -1;
	return s;
}
__host__ __device__ static inline unsigned psum_skip_mask(const PsumSkip &s, int j, bool isf)
{
	if (j == s.rows[0] || j == s.rows[1] || j == s.rows[2]) return 0x100u;
	return isf ? s.colsF : s.colsS;
}

// one k-parity of planes (kr0 .. kr0+nrk-1 of parity kb) of a partial-sum sweep: the unit between two halo exchanges of a slab
// decomposition; nbr bit 0 / 1: the ghost plane below / above belongs to a neighbouring rank
void relax3_planes27_psum(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int kb, int up, int kr0,
                          int nrk, int nbr, int frun, hipStream_t st, const PsumSkip *skip = nullptr);
// the same through the operator's registration (relax3_prepare), every plane of the parity, in order (no halo in flight):
// the launch of a rank grid with an x / y split after its boundary-first chain.  false = the level has no partial-sum sweep.
bool relax3_planes27_masked(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int KK, int kb,
                            int up, const PsumSkip &skip, hipStream_t st);
// relax3_prepare with the partial-sum sweep registered from psum_min_rows rows on (runs of 8 rows below the levels that take
// it anyway): relax3_planes27_masked uses the run length fixed here
int relax3_prepare_rows(const real_t *so, const real_t *sor, int II, int JJ, int KK, int min_rows, int psum_min_rows, hipStream_t st);
// boundary-first chain pieces (relax3d.hip), reference order:
//   rows j0, j0+jstep, .. (nrj of them) of every plane of parity kb, both i-colours;
//   the points of the listed columns (0-based offsets, relaxed in the order given) in every row of class jb but xrow0 / xrow1
void relax3_rows27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int KK, int j0, int jstep,
                   int nrj, int kb, int efirst, hipStream_t st);
void relax3_cols27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int KK, int jb, int kb,
                   int ncol, const int *cols, int xrow0, int xrow1, hipStream_t st);
// the column stage on a dense copy of the six operator columns next to an x face (side 0 low / 1 high; out: relax3_strip_doubles)
size_t relax3_strip_doubles(int JJ, int KK);
void relax3_strip_build(const real_t *so, const real_t *sor, int II, int JJ, int KK, int side, real_t *out, hipStream_t st);
void relax3_cols27_strip(const real_t *strip_lo, const real_t *strip_hi, const real_t *qf, real_t *q, int II, int JJ, int KK, int jb,
                         int kb, int ncol, const int *cols, int xrow0, int xrow1, hipStream_t st);
// run length of the partial-sum sweep on a level with JJ-2 rows (0 = the level keeps the reference order), relax3d.hip
int relax3_psum_frun(int JJ);

// relax2d.hip: nine-point sweep with inter-row partial sums kept in LDS (the 2D analogue; same contract)
bool relax2_psum_wanted(int II, int JJ);
void relax2_gs9_psum(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int updown, hipStream_t st,
                     Batch bt = Batch());

} // namespace cedar_amd
