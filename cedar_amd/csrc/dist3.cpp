// C-ABI layer 4: the domain-decomposed 3D solver -- one rank per GPU -- below the C ABI.
//
// What this replaces in the reference: cdr3::mpi::solver and the MPI flavour of the kernels,
//   include/cedar/3d/mpi/solver.h:76-89,232-233 (halo of b before the solve, the multilevel driver on local boxes),
//   src/3d/mpi/msg_exchanger.cc:188-197 (exchange_func: one ghost layer to every neighbouring rank),
//   src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147 (global-parity colouring, halo after every colour),
//   ..._residual.f90:130, ..._interp_add.f90:308, ..._SETUP_interp_OI.f90:418-1074, ..._SETUP_ITLI27_ex.f90:1803,
//   include/cedar/3d/mpi/grid_func.h:41 (norm all-reduce), include/cedar/3d/mpi/redist_solver.h:221-224 (coarse gather).
// Round 2 kept this orchestration in Python (cedar_amd/dist.py: one ctypes call per kernel piece, 3.3 ms of host
// enqueue per 2x2x2 cycle); here it is compiled code: a rank process creates a handle and calls vcycle / solve, Python
// keeps only the rank bootstrap (unique id over TCP).  Same design rule as dist.py -- serial equivalence by
// construction: every rank runs the serial kernels on its box (owned points + one ghost layer), ghost layers hold the
// owner's current values whenever a kernel reads them, local extents stay even on every distributed level so local
// and global parities coincide; the N-rank residual history equals the single-domain history (the reference's own
// criterion, test/3d/mpi/test_relax.cc:56-59).
//
// Transport: the library's RCCL communicator (comm.cpp) -- or a caller-supplied table of three functions, the
// counterpart of the reference's halo_exchanger plug-in (include/cedar/kernel.h:25-37, kernel_manager::add_halo):
// the one-GPU rehearsal hands in a host-staged transport so that several ranks can share a card (RCCL refuses that).
#include "dist_common.h"
#include <cmath>

using namespace cedar_amd;
using namespace cedar_amd::dist;

namespace {

struct DLevel {
	int n[3] = {0, 0, 0};
	int II = 0, JJ = 0, KK = 0, nst = 14;
	size_t npts = 0;
	real_t *A = nullptr, *P = nullptr, *x = nullptr, *b = nullptr, *res = nullptr, *sor = nullptr;
	bool ownA = true, overlap = false;
	real_t *strip[2] = {nullptr, nullptr}; // dense copies of the operator columns next to the low / high x face (chain levels)
	bool chain = false; // x / y split and the level takes the partial-sum sweep: boundary-first chain + one masked launch per k-parity
	Halo halo;
};

} // namespace

struct cedar_amd_dist3 : RankCtx {
	int pre = 2, post = 1, max_iter = 10, min_coarse = 3, overlap_min = 96, agglomerate_below = 64;
	double tol = 1e-8;
	int sides = 0;
	int nlev_global = 1, la = 0;
	std::vector<DLevel> lv;
	int cn[3] = {0, 0, 0};
	int gII = 0, gJJ = 0, gKK = 0;
	real_t *gA = nullptr, *gx = nullptr, *gb = nullptr, *cs_tmp = nullptr;
	cedar_amd_solver *serial = nullptr;
	std::map<long, std::pair<real_t *, real_t *>> gbuf;
};

namespace {

// ---- gather of a level onto every rank (replaces the reference's redistribution solver)
void gather_into(cedar_amd_dist3 *d, real_t *local, int lII, int lJJ, int lKK, int nplanes, real_t *glob)
{
	const int nx = d->cn[0], ny = d->cn[1], nz = d->cn[2];
	const size_t blk = (size_t)nx * ny * nz;
	auto it = d->gbuf.find(nplanes);
	if (it == d->gbuf.end())
		it = d->gbuf.emplace((long)nplanes, std::make_pair(dmalloc(blk * nplanes), dmalloc(blk * nplanes * d->world))).first;
	real_t *sb = it->second.first, *rb = it->second.second;
	const int own[6] = {1, 1, 1, nx, ny, nz};
	const unsigned long long zero = 0;
	cedar_amd_box_copy(local, lII, lJJ, lKK, nplanes, 1, own, &zero, sb, 0);
	tp_allgather(d, sb, rb, blk * nplanes);
	// unpack every rank's block at its place; the box table of one launch holds 26 boxes
	for (int r0 = 0; r0 < d->world; r0 += 26) {
		const int nb = d->world - r0 < 26 ? d->world - r0 : 26;
		int boxes[26 * 6];
		unsigned long long offs[26];
		for (int i = 0; i < nb; i++) {
			const int r = r0 + i, ci = r % d->p[0], cj = (r / d->p[0]) % d->p[1], ck = r / (d->p[0] * d->p[1]);
			const int b[6] = {1 + ci * nx, 1 + cj * ny, 1 + ck * nz, nx, ny, nz};
			memcpy(boxes + 6 * i, b, sizeof(b));
			offs[i] = (unsigned long long)r * blk;
		}
		cedar_amd_box_copy(glob, d->gII, d->gJJ, d->gKK, nplanes, nb, boxes, offs, rb, 1);
	}
}

// ---- cycle pieces
// One k-parity of planes on a rank grid with an x / y split, partial-sum sweep (relax3d_psum.hip).  The plane-fused launch
// relaxes both row classes and both i-colours of a plane in one go, while the reference's MPI sweep exchanges after every
// colour (src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147).  Boundary-first: the points next to a neighbouring rank, closed
// under "needs the fresh value of" within the four colours of the parity, are relaxed ahead in the reference order, stage
// by stage with the exchanges between -- a few columns and rows; the launch then relaxes everything else and leaves those
// points as they are (every point of the rest that a chain point neighbours comes LATER in the colour order, or the chain
// would contain it, so the rest sees exactly the values the serial sweep sees).
//   colours of a parity in sweep order: F rows first colour (c1), F rows c2, S rows c1, S rows c2;
//   side P of x: its boundary column is c1 (UP: the low side), side Q: c2.   d = distance of a column from the boundary.
//   chain, side Q: F c1 {d1,d3}, F c2 {d0,d2}, S c1 {d1}, S c2 {d0};   side P: F c1 {d0,d2}, F c2 {d1}, S c1 {d0}
//   side P of y: its boundary row is an F row (UP: the low side): that row;   side Q: F row d1, S row d0 (whole rows)
// Both colours of a row class run in one kernel with the ghost column of side Q still stale: its c2 column d0 is
// recomputed after the exchange (nobody has read it in between) -- three exchanges within the rank's z layer per parity
// (one without an x split), each restricted to what the stage changed, then the launch, then one exchange with everything else.
void chain_parity(cedar_amd_dist3 *d, DLevel &L, real_t *x, real_t *b, int kb, bool up)
{
	const int nx = L.n[0], ny = L.n[1];
	const int jbF = up ? 0 : 1, jbS = 1 - jbF;
	int colsF[8], colsS[8], fixc[2], nF = 0, nS = 0, nfix = 0;
	unsigned mF = 0, mS = 0;
	int c1[2][2], c2[2][2], n1[2] = {0, 0}, n2[2] = {0, 0}; // per side: columns of the F stage by colour
	for (int side = 0; side < 2; side++) { // 0 = low x, 1 = high x
		if (!has_nb(d, 0, side ? +1 : -1)) continue;
		const bool isP = (side == 0) == up;
		auto col = [&](int dd) { return side ? nx - dd : 1 + dd; };
		auto bit = [&](int dd) { return 1u << (side ? 7 - dd : dd); };
		if (isP) {
			c1[side][n1[side]++] = col(0); c1[side][n1[side]++] = col(2); c2[side][n2[side]++] = col(1);
			mF |= bit(0) | bit(1) | bit(2);
			colsS[nS++] = col(0);
			mS |= bit(0);
		} else {
			c1[side][n1[side]++] = col(1); c1[side][n1[side]++] = col(3); c2[side][n2[side]++] = col(2); c2[side][n2[side]++] = col(0);
			mF |= bit(0) | bit(1) | bit(2) | bit(3);
			colsS[nS++] = col(1); colsS[nS++] = col(0); // c1 then c2 (two sides: the low side's c2 before the high side's c1 is harmless, they do not couple)
			mS |= bit(0) | bit(1);
			fixc[nfix++] = col(0);
		}
	}
	for (int side = 0; side < 2; side++) for (int t = 0; t < n1[side]; t++) colsF[nF++] = c1[side][t];
	for (int side = 0; side < 2; side++) for (int t = 0; t < n2[side]; t++) colsF[nF++] = c2[side][t];
	int rowsF[2], rowS = -1, nrF = 0;
	for (int side = 0; side < 2; side++) { // 0 = low y, 1 = high y
		if (!has_nb(d, 1, side ? +1 : -1)) continue;
		const bool isP = (side == 0) == up;
		auto row = [&](int dd) { return side ? ny - dd : 1 + dd; };
		if (isP) rowsF[nrF++] = row(0);
		else { rowsF[nrF++] = row(1); rowS = row(0); }
	}
	const int skip[3] = {nrF > 0 ? rowsF[0] : -1, nrF > 1 ? rowsF[1] : -1, rowS};
	const bool xs = d->p[0] > 1;
	auto cols = [&](int jb, int n, const int *cl, int x0, int x1) {
		if (n <= 0) return;
		if (L.strip[0] || L.strip[1]) cedar_amd_relax3_cols_strip(L.strip[0], L.strip[1], b, x, L.II, L.JJ, L.KK, jb, kb, n, cl, x0, x1);
		else cedar_amd_relax3_cols(L.A, b, x, L.sor, L.II, L.JJ, L.KK, jb, kb, n, cl, x0, x1);
	};
	// what a stage has changed: rows of one class in the planes of the parity; side Q's fixed column travels one way only
	const int jparF = (1 + jbF) & 1, jparS = (1 + jbS) & 1, kpar = (1 + kb) & 1, qdir = up ? +1 : -1;
	auto layer = [](const int *o) { return o[2] == 0; };
	auto to_q = [&](const int *o) { return o[2] == 0 && o[0] == qdir; };
	auto to_p = [&](const int *o) { return o[2] == 0 && o[0] == -qdir; };
	// the fixed column also sits in the boundary row a y neighbour has already received: everything but the messages across side P
	auto not_p = [&](const int *o) { return o[2] == 0 && o[0] != -qdir; };
	auto not_q = [&](const int *o) { return o[2] == 0 && o[0] != qdir; };
	// F rows
	if (nrF) cedar_amd_relax3_rows(L.A, b, x, L.sor, L.II, L.JJ, L.KK, rowsF[0], nrF > 1 ? rowsF[1] - rowsF[0] : 2, nrF, kb, up);
	cols(jbF, nF, colsF, skip[0], skip[1]);
	auto rowsF_only = [&](const int *) { return jparF; };
	auto rowsS_only = [&](const int *) { return jparS; };
	halo_exchange_sub(d, L.halo, L.II, L.JJ, L.KK, x, layer, layer, rowsF_only, kpar);
	if (xs) {
		cols(jbF, nfix, fixc, -1, -1);
		halo_exchange_sub(d, L.halo, L.II, L.JJ, L.KK, x, not_p, not_q, rowsF_only, kpar);
	}
	// S rows
	if (rowS >= 0) cedar_amd_relax3_rows(L.A, b, x, L.sor, L.II, L.JJ, L.KK, rowS, 2, 1, kb, up);
	cols(jbS, nS, colsS, rowS, -1);
	if (xs) {
		halo_exchange_sub(d, L.halo, L.II, L.JJ, L.KK, x, to_p, to_q, rowsS_only, kpar); // side P's first colour to the neighbour's side Q
		cols(jbS, nfix, fixc, -1, -1);
	}
	// The launch reads no ghost cell of the rank's own z layer: only the points of a boundary column / row do, and those are
	// chain points (its ghost-column sources feed partial sums of column d0 in the planes of the other parity -- chain points
	// again, relaxed from the operator).  The S rows therefore travel with the exchange across z after the launch.
	if (!cedar_amd_relax3_planes_masked(L.A, b, x, L.sor, L.II, L.JJ, L.KK, kb, up, mF, mS, skip)) {
		char m[] = "cedar_amd_dist3: the masked launch refused a level set up for it";
		print_error(m);
	}
	auto all = [](const int *) { return true; };
	auto rows_last = [&](const int *o) { return o[2] == 0 ? jparS : -1; };
	halo_exchange_sub(d, L.halo, L.II, L.JJ, L.KK, x, all, all, rows_last, kpar);
}

void smooth(cedar_amd_dist3 *d, DLevel &L, real_t *x, real_t *b, int updown, int nsweeps)
{
	const bool up = updown == BMG_UP;
	for (int it = 0; it < nsweeps; it++) {
		if (L.chain) {
			for (int c = 0; c < 2; c++) chain_parity(d, L, x, b, up ? c : 1 - c, up);
			continue;
		}
		if (L.nst == 4) {
			for (int c = 0; c < 2; c++) {
				cedar_amd_relax3_colour7(L.A, b, x, L.sor, L.II, L.JJ, L.KK, up ? c : 1 - c);
				halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 0);
			}
			continue;
		}
		if (d->p[0] == 1 && d->p[1] == 1) {
			// slab decomposition: a whole k-parity of planes (both row classes: the plane-fused kernel on big levels)
			// between two exchanges; its planes next to a ghost plane wait for the previous parity's halo, the others run
			// under it
			for (int c = 0; c < 2; c++) {
				const int kb = up ? c : 1 - c;
				if (L.overlap) {
					cedar_amd_relax3_planes(L.A, b, x, L.sor, L.II, L.JJ, L.KK, kb, up, 1 | (d->sides << 4));
					side_wait(d);
					cedar_amd_relax3_planes(L.A, b, x, L.sor, L.II, L.JJ, L.KK, kb, up, 2 | (d->sides << 4));
					void *m = side_begin(d);
					halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 0);
					side_end(d, m);
				} else {
					cedar_amd_relax3_planes(L.A, b, x, L.sor, L.II, L.JJ, L.KK, kb, up, 0);
					halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 0);
				}
			}
			continue;
		}
		for (int c = 0; c < 4; c++) {
			const int cc = up ? c : 3 - c, jb = cc & 1, kb = cc >> 1;
			if (L.overlap) {
				// interior rows first: they read no y/z ghost, whose exchange (previous pass) may still be in flight on the
				// side stream; then join and relax the shell rows
				cedar_amd_relax3_pass_part(L.A, b, x, L.sor, L.II, L.JJ, L.KK, jb, kb, up, 1 | (d->sides << 4));
				side_wait(d);
				cedar_amd_relax3_pass_part(L.A, b, x, L.sor, L.II, L.JJ, L.KK, jb, kb, up, 2 | (d->sides << 4));
			} else
				cedar_amd_relax3_pass_part(L.A, b, x, L.sor, L.II, L.JJ, L.KK, jb, kb, up, 0);
			if (d->p[0] > 1) {
				// the second i-colour of the column next to an x neighbour needs that neighbour's fresh first colour
				if (halo_exchange_x(d, L.halo, L.II, L.JJ, L.KK, x, up)) cedar_amd_relax3_fixup(L.A, b, x, L.sor, L.II, L.JJ, L.KK, up ? L.n[0] : 1, jb, kb);
			}
			if (L.overlap) {
				halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 1); // x ghosts are read by every row of the next pass: in order
				void *m = side_begin(d);
				halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 2);
				side_end(d, m);
			} else
				halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 0);
		}
	}
	side_wait(d);
}

// levels la.. : gather the right-hand side, one single-domain cycle (or the direct solve) from a zero initial guess,
// keep the own block + ghosts straight from the global solution
void coarse_solve(cedar_amd_dist3 *d, DLevel &C, real_t *x, real_t *b)
{
	gather_into(d, b, C.II, C.JJ, C.KK, 1, d->gb);
	cedar_amd_memset(d->gx, 0, (size_t)d->gII * d->gJJ * d->gKK * sizeof(real_t));
	cedar_amd_solver_vcycle(d->serial, d->gx, d->gb);
	const int nx = d->cn[0], ny = d->cn[1], nz = d->cn[2];
	const unsigned long long zero = 0;
	const int from[6] = {d->coord[0] * nx, d->coord[1] * ny, d->coord[2] * nz, nx + 2, ny + 2, nz + 2};
	const int to[6] = {0, 0, 0, nx + 2, ny + 2, nz + 2};
	cedar_amd_box_copy(d->gx, d->gII, d->gJJ, d->gKK, 1, 1, from, &zero, d->cs_tmp, 0);
	cedar_amd_box_copy(x, C.II, C.JJ, C.KK, 1, 1, to, &zero, d->cs_tmp, 1);
}

void cycle(cedar_amd_dist3 *d, int l, real_t *x, real_t *b)
{
	DLevel &L = d->lv[l], &K = d->lv[l + 1];
	smooth(d, L, x, b, BMG_DOWN, d->pre);
	BMG3_SymStd_residual(1, 1, L.nst == 4, x, b, L.A, L.res, L.II, L.JJ, L.KK, L.nst);
	halo_exchange(d, L.halo, L.II, L.JJ, L.KK, L.res, 1, 0);
	BMG3_SymStd_restrict(L.res, K.b, K.P, L.II, L.JJ, L.KK, K.II, K.JJ, K.KK, 0);
	cedar_amd_memset(K.x, 0, K.npts * sizeof(real_t));
	if (l + 1 == (int)d->lv.size() - 1) coarse_solve(d, K, K.x, K.b);
	else cycle(d, l + 1, K.x, K.b);
	BMG3_SymStd_interp_add(x, K.x, L.A, L.res, K.P, K.II, K.JJ, K.KK, L.II, L.JJ, L.KK, L.nst, 0);
	halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 0);
	smooth(d, L, x, b, BMG_UP, d->post);
}

void vcycle(cedar_amd_dist3 *d, real_t *x, real_t *b)
{
	if (d->lv.size() == 1) coarse_solve(d, d->lv[0], x, b);
	else cycle(d, 0, x, b);
}

double norm(cedar_amd_dist3 *d, DLevel &L, const real_t *r)
{
	const double v = cedar_amd_l2norm(r, L.II, L.JJ, L.KK);
	return std::sqrt(tp_allreduce_sum(d, v * v));
}

// set-up: multilevel.h:243-265 with the MPI flavour's ghost updates
void setup(cedar_amd_dist3 *d)
{
	const int lo[3] = {has_nb(d, 0, -1) ? 2 : 3, has_nb(d, 1, -1) ? 2 : 3, has_nb(d, 2, -1) ? 2 : 3};
	DLevel &L0 = d->lv[0];
	halo_exchange(d, L0.halo, L0.II, L0.JJ, L0.KK, L0.A, L0.nst, 0);
	for (size_t l = 0; l + 1 < d->lv.size(); l++) {
		DLevel &F = d->lv[l], &K = d->lv[l + 1];
		for (int phase = 0; phase < 3; phase++) {
			cedar_amd_setup_interp3_phase(F.A, K.P, F.II, F.JJ, F.KK, K.II, K.JJ, K.KK, F.nst == 4, F.nst, phase, lo[0], lo[1], lo[2]);
			halo_exchange(d, K.halo, K.II, K.JJ, K.KK, K.P, 26, 0);
		}
		if (F.nst == 4) BMG3_SymStd_SETUP_ITLI07_ex(F.A, K.A, K.P, F.II, F.JJ, F.KK, K.II, K.JJ, K.KK, 0);
		else BMG3_SymStd_SETUP_ITLI27_ex(F.A, K.A, K.P, F.II, F.JJ, F.KK, K.II, K.JJ, K.KK, 0);
		halo_exchange(d, K.halo, K.II, K.JJ, K.KK, K.A, 14, 0);
		BMG3_SymStd_SETUP_recip(F.A, F.sor, F.II, F.JJ, F.KK, F.nst, 2);
		// slab decomposition: its sweeps are the plane-fused passes of the single-GPU solver, which read the
		// row-interleaved solve copy where one is registered (worth 7 %; neutral on rank grids with an x / y split)
		if (F.nst == 14 && d->p[0] == 1 && d->p[1] == 1) cedar_amd_relax3_prepare(F.A, F.sor, F.II, F.JJ, F.KK);
		// x / y split: where the level takes the partial-sum sweep (scratch registered: bit 1), the boundary-first chain
		// (chain_parity); CEDAR_AMD_DIST_CHAIN=0 keeps the reference-order row-class passes
		else if (F.nst == 14 && F.n[0] >= 8 && F.n[1] >= 8 && !(getenv("CEDAR_AMD_DIST_CHAIN") && !atoi(getenv("CEDAR_AMD_DIST_CHAIN"))) &&
		         F.n[1] >= (getenv("CEDAR_AMD_DIST_CHAIN_MIN") ? atoi(getenv("CEDAR_AMD_DIST_CHAIN_MIN")) : 0))
			F.chain = (cedar_amd_relax3_prepare_rows(F.A, F.sor, F.II, F.JJ, F.KK,
			                                         getenv("CEDAR_AMD_DIST_PSUM_MIN") ? atoi(getenv("CEDAR_AMD_DIST_PSUM_MIN")) : 128) & 2) != 0;
		if (F.chain && F.n[0] >= 12 && !(getenv("CEDAR_AMD_DIST_STRIP") && !atoi(getenv("CEDAR_AMD_DIST_STRIP"))))
			for (int side = 0; side < 2; side++)
				if (has_nb(d, 0, side ? +1 : -1)) {
					F.strip[side] = dmalloc(cedar_amd_relax3_strip_doubles(F.JJ, F.KK));
					cedar_amd_relax3_strip_build(F.A, F.sor, F.II, F.JJ, F.KK, side, F.strip[side]);
				}
	}
	// level la: the global operator on every rank; the single-domain device-resident solver takes over from there
	DLevel &C = d->lv.back();
	for (int t = 0; t < 3; t++) d->cn[t] = C.n[t];
	d->gII = C.n[0] * d->p[0] + 2; d->gJJ = C.n[1] * d->p[1] + 2; d->gKK = C.n[2] * d->p[2] + 2;
	const size_t gp = (size_t)d->gII * d->gJJ * d->gKK;
	d->gA = dmalloc(gp * C.nst);
	gather_into(d, C.A, C.II, C.JJ, C.KK, C.nst, d->gA);
	d->gx = dmalloc(gp);
	d->gb = dmalloc(gp);
	d->cs_tmp = dmalloc((size_t)(C.n[0] + 2) * (C.n[1] + 2) * (C.n[2] + 2));
	cedar_amd_settings st;
	cedar_amd_default_settings(&st);
	st.nrelax_pre = d->pre; st.nrelax_post = d->post; st.min_coarse = d->min_coarse;
	st.num_levels = d->nlev_global - d->la;
	d->serial = cedar_amd_solver_create(3, (len_t)(d->gII - 2), (len_t)(d->gJJ - 2), (len_t)(d->gKK - 2), C.nst, d->gA, 1, &st);
}

} // namespace

namespace {
// loop-back transport: every message of the rank grid is packed, copied device to device in place of the send / receive,
// and unpacked -- the ghost values are meaningless, the work per cycle is that of one rank of the grid.  A measuring aid
// (tools/dist_overhead.py: what one rank costs beside its messages) for boxes with a single GPU.
int loop_world = 1;
int loop_exchange(void *, int ns, const int *, const real_t *const *sbuf, const size_t *scount, int nr, const int *,
                  real_t *const *rbuf, const size_t *rcount)
{
	// a grouped send / receive is ONE launch on the RCCL transport: where the receive side mirrors the send side (the halo
	// buffers do) the stand-in is one copy of the span the messages cover, else one per message
	bool mirror = ns == nr && ns > 0;
	const real_t *lo = ns ? sbuf[0] : nullptr, *hi = lo;
	for (int i = 0; i < ns && mirror; i++) {
		mirror = scount[i] == rcount[i] && (rbuf[i] - rbuf[0]) == (sbuf[i] - sbuf[0]);
		if (sbuf[i] < lo) lo = sbuf[i];
		if (sbuf[i] + scount[i] > hi) hi = sbuf[i] + scount[i];
	}
	if (mirror) {
		cedar_amd_memcpy_d2d(rbuf[0] + (lo - sbuf[0]), lo, (size_t)(hi - lo) * sizeof(real_t));
		return 0;
	}
	for (int i = 0; i < ns && i < nr; i++)
		cedar_amd_memcpy_d2d(rbuf[i], sbuf[i], (scount[i] < rcount[i] ? scount[i] : rcount[i]) * sizeof(real_t));
	return 0;
}
int loop_allgather(void *, const real_t *send, real_t *recv, size_t count)
{
	for (int r = 0; r < loop_world; r++) cedar_amd_memcpy_d2d(recv + (size_t)r * count, send, count * sizeof(real_t));
	return 0;
}
int loop_allreduce(void *, double *v, int n)
{
	for (int i = 0; i < n; i++) v[i] *= loop_world;
	return 0;
}
} // namespace

extern "C" {

void cedar_amd_transport_loopback(cedar_amd_transport *out, int world)
{
	loop_world = world;
	out->ctx = nullptr;
	out->exchange = loop_exchange;
	out->allgather = loop_allgather;
	out->allreduce_sum = loop_allreduce;
}

void cedar_amd_dist3_rank_grid(int world, int pgrid[3])
{
	// 1 -> 1x1x1, 2 -> 1x1x2, 4 -> 1x1x4 (z slabs), 8 -> 2x2x2 (BASELINE config 5); otherwise the most cubic
	// factorisation with pz >= py >= px (cedar_amd/dist.py rank_grid)
	if (world <= 4) { pgrid[0] = 1; pgrid[1] = 1; pgrid[2] = world < 1 ? 1 : world; return; }
	long best0 = 0, best1 = 0;
	bool have = false;
	for (int pz = 1; pz <= world; pz++) {
		if (world % pz) continue;
		for (int py = pz; py <= world / pz; py++) {
			if ((world / pz) % py) continue;
			const int px = world / pz / py;
			if (px < py) continue;
			const long k0 = px - pz, k1 = px;
			if (!have || k0 < best0 || (k0 == best0 && k1 < best1)) {
				have = true; best0 = k0; best1 = k1;
				pgrid[0] = pz; pgrid[1] = py; pgrid[2] = px; // reversed: the largest factor along z
			}
		}
	}
}

cedar_amd_dist3 *cedar_amd_dist3_create(cedar_amd_comm *comm, const cedar_amd_transport *transport, int rank, int world,
                                        const int pgrid[3], real_t *A_local, len_t nx, len_t ny, len_t nz, int nstencil,
                                        const cedar_amd_settings *settings, int agglomerate_below, int overlap_min)
{
	if (!A_local || !is_device_ptr(A_local) || (nstencil != 14 && nstencil != 4) || world < 1 || rank < 0 || rank >= world) {
		char m[] = "cedar_amd_dist3_create: A_local must be a device array of a 7- or 27-point operator, 0 <= rank < world";
		print_error(m);
		return nullptr;
	}
	if (world > 1 && !comm && !(transport && transport->exchange && transport->allgather && transport->allreduce_sum)) {
		char m[] = "cedar_amd_dist3_create: more than one rank needs a communicator (cedar_amd_comm_create) or a transport table";
		print_error(m);
		return nullptr;
	}
	cedar_amd_dist3 *d = new cedar_amd_dist3;
	d->comm = comm;
	if (transport && transport->exchange) { d->tp = *transport; d->has_tp = true; }
	d->rank = rank; d->world = world;
	if (pgrid) { d->p[0] = pgrid[0]; d->p[1] = pgrid[1]; d->p[2] = pgrid[2]; }
	else cedar_amd_dist3_rank_grid(world, d->p);
	if (d->p[0] * d->p[1] * d->p[2] != world) {
		char m[] = "cedar_amd_dist3_create: the rank grid does not multiply to the world size";
		print_error(m);
		delete d;
		return nullptr;
	}
	d->coord[0] = rank % d->p[0]; d->coord[1] = (rank / d->p[0]) % d->p[1]; d->coord[2] = rank / (d->p[0] * d->p[1]);
	cedar_amd_settings st;
	if (settings) st = *settings;
	else cedar_amd_default_settings(&st);
	d->pre = st.nrelax_pre; d->post = st.nrelax_post; d->max_iter = st.max_iter; d->tol = st.tol; d->min_coarse = st.min_coarse;
	d->agglomerate_below = agglomerate_below > 0 ? agglomerate_below : 64;
	d->overlap_min = overlap_min > 0 ? overlap_min : 96;
	d->sides = (int)has_nb(d, 1, -1) | (int)has_nb(d, 1, +1) << 1 | (int)has_nb(d, 2, -1) << 2 | (int)has_nb(d, 2, +1) << 3;
	d->scal = dmalloc(8);
	int n[3] = {(int)nx, (int)ny, (int)nz};
	// number of levels from the GLOBAL extents (include/cedar/3d/solver.h:54-72)
	int ng = 0;
	for (;;) {
		ng++;
		int m = 1 << 30;
		for (int t = 0; t < 3; t++) {
			const int g = n[t] * d->p[t], c = (g - 1) / (1 << ng) + 1;
			if (c < m) m = c;
		}
		if (m < d->min_coarse) break;
	}
	d->nlev_global = ng;
	// distributed levels 0 .. la; level la is gathered and handed to the single-domain solver
	int la = ng - 1, m[3] = {n[0], n[1], n[2]};
	for (int l = 1; l < ng; l++) {
		int mn = 1 << 30;
		for (int t = 0; t < 3; t++) {
			m[t] = d->p[t] == 1 ? (int)((m[t] - 1) / 2.0 + 1) : m[t] / 2;
			if (m[t] < mn) mn = m[t];
		}
		if (mn <= d->agglomerate_below) { la = l; break; }
	}
	d->la = ng > 1 ? (la > 1 ? la : 1) : 0;
	for (int l = 0; l <= d->la; l++) {
		DLevel L;
		for (int t = 0; t < 3; t++) L.n[t] = n[t];
		for (int t = 0; t < 3; t++)
			if (d->p[t] > 1 && l < d->la && (n[t] & 1)) {
				char msg[160];
				snprintf(msg, sizeof(msg), "cedar_amd_dist3_create: level %d: local extent %d along a split direction must be even", l, n[t]);
				print_error(msg);
				cedar_amd_dist3_destroy(d);
				return nullptr;
			}
		L.II = n[0] + 2; L.JJ = n[1] + 2; L.KK = n[2] + 2;
		L.npts = (size_t)L.II * L.JJ * L.KK;
		d->lv.push_back(L);
		DLevel &R = d->lv.back();
		halo_init(d, R.halo, n);
		R.overlap = !R.halo.grp[2].idx.empty() && n[0] >= d->overlap_min && n[1] >= d->overlap_min && n[2] >= d->overlap_min;
		R.res = dmalloc(R.npts);
		R.sor = dmalloc(2 * R.npts);
		if (l == 0) {
			R.A = A_local; R.ownA = false; R.nst = nstencil;
		} else {
			R.nst = 14;
			R.A = dmalloc(14 * R.npts);
			R.P = dmalloc(26 * R.npts);
			R.x = dmalloc(R.npts);
			R.b = dmalloc(R.npts);
		}
		for (int t = 0; t < 3; t++) n[t] = d->p[t] == 1 ? (int)((n[t] - 1) / 2.0 + 1) : n[t] / 2;
	}
	setup(d);
	if (!d->serial) {
		cedar_amd_dist3_destroy(d);
		return nullptr;
	}
	launch_check("cedar_amd_dist3_create");
	return d;
}

void cedar_amd_dist3_destroy(cedar_amd_dist3 *d)
{
	if (!d) return;
	cedar_amd_device_sync();
	if (d->serial) cedar_amd_solver_destroy(d->serial);
	for (DLevel &L : d->lv) {
		if (L.ownA) cedar_amd_free(L.A);
		else if (L.A) cedar_amd_relax3_release(L.A); // the caller's operator: only its registered solve copy goes
		cedar_amd_free(L.strip[0]); cedar_amd_free(L.strip[1]);
		cedar_amd_free(L.P); cedar_amd_free(L.x); cedar_amd_free(L.b); cedar_amd_free(L.res); cedar_amd_free(L.sor);
		for (auto &kv : L.halo.bufs) { cedar_amd_free(kv.second.first); cedar_amd_free(kv.second.second); }
	}
	for (auto &kv : d->gbuf) { cedar_amd_free(kv.second.first); cedar_amd_free(kv.second.second); }
	cedar_amd_free(d->gA); cedar_amd_free(d->gx); cedar_amd_free(d->gb); cedar_amd_free(d->cs_tmp); cedar_amd_free(d->scal);
	if (d->side) cedar_amd_stream_destroy(d->side);
	delete d;
}

int cedar_amd_dist3_nlevels(const cedar_amd_dist3 *d) { return d ? d->nlev_global : 0; }
int cedar_amd_dist3_distributed_levels(const cedar_amd_dist3 *d) { return d ? (int)d->lv.size() : 0; }
int cedar_amd_dist3_chain_levels(const cedar_amd_dist3 *d)
{
	int n = 0;
	if (d) for (const DLevel &L : d->lv) n += L.chain ? 1 : 0;
	return n;
}

void cedar_amd_dist3_vcycle(cedar_amd_dist3 *d, real_t *x, real_t *b)
{
	if (!d) return;
	vcycle(d, x, b);
	launch_check("cedar_amd_dist3_vcycle");
}

// multilevel::solve (multilevel.h:277-298) after mpi::solver::solve's halo of the iterate (3d/mpi/solver.h:76-89);
// rel[0] = ||r0||_2, rel[i] = ||r_i||_2 / ||r0||_2; returns the number of cycles run
int cedar_amd_dist3_solve(cedar_amd_dist3 *d, real_t *b, real_t *x, real_t *rel)
{
	if (!d) return 0;
	DLevel &L = d->lv[0];
	halo_exchange(d, L.halo, L.II, L.JJ, L.KK, x, 1, 0);
	BMG3_SymStd_residual(1, 1, L.nst == 4, x, b, L.A, L.res, L.II, L.JJ, L.KK, L.nst);
	const double r0 = norm(d, L, L.res);
	rel[0] = r0;
	int it = 0;
	while (it < d->max_iter) {
		vcycle(d, x, b);
		BMG3_SymStd_residual(1, 1, L.nst == 4, x, b, L.A, L.res, L.II, L.JJ, L.KK, L.nst);
		const double r = norm(d, L, L.res) / r0;
		rel[++it] = r;
		if (r < d->tol) break;
	}
	launch_check("cedar_amd_dist3_solve");
	return it;
}

// n level-0 relax sweeps alternating DOWN / UP with their halo exchanges (the roofline microbenchmark of the
// decomposed path); elapsed milliseconds by HIP events on the library's stream
float cedar_amd_dist3_time_relax(cedar_amd_dist3 *d, real_t *x, real_t *b, int n)
{
	if (!d) return 0.f;
	void *e0 = cedar_amd_event_record();
	for (int i = 0; i < n; i++) smooth(d, d->lv[0], x, b, (i & 1) ? BMG_UP : BMG_DOWN, 1);
	void *e1 = cedar_amd_event_record();
	const float ms = cedar_amd_event_elapsed_ms(e0, e1);
	cedar_amd_event_destroy(e0);
	cedar_amd_event_destroy(e1);
	return ms;
}

} // extern "C"
