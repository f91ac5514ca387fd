// Factors of the line segments a rank owns (distributed line relaxation, dist_lines.hip / dist2.cpp), stored by colour:
// the lines of colour 0 (l = 0, 2, ..) first, then colour 1; each array (lines, npos), row stride npos.
#pragma once
#include "common.h"

namespace cedar_amd {

struct LineFactors {
	real_t *dp = nullptr; // pivots d'_i
	real_t *af = nullptr; // -e'_{i-1}: multiplier of the forward sweep at position i (0 at the first unknown of a line)
	real_t *ab = nullptr; // -e'_i:     multiplier of the backward sweep at position i (towards i+1; the last one crosses to the next segment)
	real_t *pf = nullptr, *pb = nullptr; // running products of af (from the left) and ab (from the right); 0 without a neighbour on that side
	int nl = 0, npos = 0;
	__host__ __device__ size_t colour_offset(int lb) const { return lb ? (size_t)((nl + 1) / 2) * npos : 0; }
	__host__ __device__ size_t line_offset(int l) const { return colour_offset(l & 1) + (size_t)(l >> 1) * npos; }
	__host__ __device__ int colour_lines(int lb) const { return (nl - lb + 1) / 2; }
};

void dist_lines_factor(const real_t *A, int II, int JJ, int dir, int npos, int nl, const real_t *piv_in, int has_prev,
                       int has_next, const LineFactors &F, real_t *piv_out, hipStream_t st);
void dist_lines_pick(const real_t *v, const real_t *p, int nlines, int ld, int pos, real_t *out, hipStream_t st);
void dist_lines_compose(const real_t *parts, int nseg, int seg, int nlines, int backward, real_t *carry, hipStream_t st);

} // namespace cedar_amd
