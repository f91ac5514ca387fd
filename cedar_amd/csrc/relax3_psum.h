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
// one k-parity of planes (kr0 .. kr0+nrk-1 of parity kb) of a partial-sum sweep: the unit between two halo exchanges of a slab
// decomposition; nbr bit 0 / 1: the ghost plane below / above belongs to a neighbouring rank
void relax3_planes27_psum(const Op3 &A, const real_t *qf, real_t *q, real_t *T, int II, int JJ, int KK, int kb, int up, int kr0,
                          int nrk, int nbr, int frun, hipStream_t st);
// run length of the partial-sum sweep on a level with JJ-2 rows (0 = the level keeps the reference order), relax3d.hip
int relax3_psum_frun(int JJ);

// relax2d.hip: nine-point sweep with inter-row partial sums kept in LDS (the 2D analogue; same contract)
bool relax2_psum_wanted(int II, int JJ);
void relax2_gs9_psum(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int updown, hipStream_t st,
                     Batch bt = Batch());

} // namespace cedar_amd
