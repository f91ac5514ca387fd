// C-ABI layer 2: device-resident multilevel hierarchy and V-cycle.
// Restates the orchestration of the reference's serial solvers on top of the
// HIP kernels; every level array lives in HBM for the whole solve, only norms
// (one double per iteration) cross PCIe.
//   level sizing / allocation : include/cedar/2d/solver.h:57-116, include/cedar/3d/solver.h:54-123
//   set-up loop               : include/cedar/multilevel.h:243-265 (interp -> Galerkin -> relax set-up)
//   V-cycle                   : include/cedar/cycle/vcycle.h:57-115
//   smoothers                 : include/cedar/multilevel.h:165-222 (pre = DOWN, post = UP; line-xy: x,y / y,x)
//   solve loop                : include/cedar/multilevel.h:268-298
// The V-cycle is a fixed launch sequence for a given (x,b) pair; it is captured
// once into a hipGraph and replayed, which removes the per-launch host cost
// that dominates the coarse levels (a 2D 4096^2 cycle is ~130 launches).
#include "../../include/cedar_amd.h"
#include "common.h"
#include "stage.h"
#include "relax3_psum.h"
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

using namespace cedar_amd;

namespace {

struct Level {
	int nx = 0, ny = 0, nz = 1;
	int II = 0, JJ = 0, KK = 1;
	int nst = 0;
	size_t npts = 0;
	real_t *A = nullptr;
	bool ownA = true;
	real_t *P = nullptr;
	real_t *x = nullptr, *b = nullptr, *res = nullptr;
	real_t *SOR0 = nullptr, *SOR1 = nullptr;
	real_t *yscr = nullptr; // y-line scratch
	// y-lines on transposed arrays (Dirichlet; lines.hip relax_lines_yt): transposed operator planes, transposed
	// right-hand side, scratch for the transposed x; bt_fresh: bt holds the transpose of this visit's b
	real_t *At = nullptr, *bt = nullptr, *xt = nullptr;
	mutable bool bt_fresh = false;
	// scan-ordered copies of the line factors (lines.hip line_pttrs_pf): PFx of the x-line factors, PFy of the y-line
	// factors when the y sweeps run on the transposed arrays; nullptr = the kernels read SOR directly
	real_t *PFx = nullptr, *PFy = nullptr;
	// 3D 27-point: row-interleaved solve copy of A and 1/diag (common.h Op3) read by relax and residual
	real_t *Ailv = nullptr;
	// 3D 27-point: partial-sum scratch of the relax sweep (relax3d_psum.hip), one vector
	real_t *T = nullptr;
	// the set-up products (A, P, SOR, At, PF*) belong to another solver of the same operator (plane relaxation:
	// every plane solver of a direction is built from the same 2D operator); this level owns only its vectors
	bool shared = false;
	// plane relaxation (3D): the plane solvers of this level, one set per direction in use (xy, xz, yz)
	struct PlaneSet *pl[3] = {nullptr, nullptr, nullptr};
};

// include/cedar/3d/relax_planes.h:164-246.  The reference keeps one 2D solver per plane; copy_coeff gives all of
// them the same operator (planes.hip), so one hierarchy is set up and the others share its set-up products.  Planes
// of one colour are independent: instance q serves the planes 2q+1 and 2q+2 (one of each colour) on fixed 2D vectors
// (slot q of x2s / b2s), so its V-cycle is captured into a hipGraph once and replayed; the replays of a colour are
// spread over a few streams.
struct PlaneSet {
	int dir = 0, np = 0, I2 = 0, J2 = 0;
	size_t P2 = 0;
	real_t *so2 = nullptr, *x2s = nullptr, *b2s = nullptr;
	std::vector<cedar_amd_solver *> inst;
};

real_t *dalloc_raw(size_t n) // not cleared: the caller writes every element
{
	void *p = nullptr;
	CEDAR_HIP_CHECK(hipMalloc(&p, (n ? n : 1) * sizeof(real_t)));
	return static_cast<real_t *>(p);
}

real_t *dalloc(size_t n)
{
	void *p = nullptr;
	size_t bytes = (n ? n : 1) * sizeof(real_t);
	CEDAR_HIP_CHECK(hipMalloc(&p, bytes));
	zero_fill(static_cast<real_t *>(p), bytes / sizeof(real_t), current_stream());
	return static_cast<real_t *>(p);
}

} // namespace

// cedar_amd_solver_create with a batch capacity (plane relaxation builds its 2D solvers with one: planes_setup)
static cedar_amd_solver *solver_create(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so, int own_device_so,
                                       const cedar_amd_settings *settings, int batch);

struct cedar_amd_solver {
	int nd = 2;
	cedar_amd_settings st;
	std::vector<Level> lv;
	real_t *ABD = nullptr, *bbd = nullptr;
	int nabd1 = 0, nabd2 = 0;
	int *dinfo = nullptr;
	real_t *red = nullptr; // reduction scratch (4096 partials + result)
	// captured V-cycle
	hipGraphExec_t gexec = nullptr;
	const real_t *gx = nullptr, *gb = nullptr;
	int gnb = 1; // batch count the graph was captured for
	hipStream_t gstream = nullptr;
	bool use_graph = true;
	// batch of independent right-hand sides on this one hierarchy (2D, Dirichlet, V-cycle; plane relaxation runs the
	// planes of a colour as one batch): every level's vectors hold nb_alloc items, a cycle works on the first nb
	int nb_alloc = 1, nb = 1;
	// a second captured cycle: the two colours of a plane sweep may differ by one plane
	hipGraphExec_t gexec2 = nullptr;
	int gnb2 = 0;
	bool shared_abd = false; // ABD belongs to the solver this one was cloned from
	// plane relaxation: side streams and their fork / join events
	std::vector<hipStream_t> pstreams;
	std::vector<hipEvent_t> pevents;
	hipEvent_t pfork = nullptr;
};

namespace {

int compute_num_levels(int nd, len_t nx, len_t ny, len_t nz, int min_coarse)
{
	// float nxc = (nx-1)/(1<<ng) + 1 with unsigned integer division; do { } while (min >= min_coarse)
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

void level_init(Level &L, int nd, int nx, int ny, int nz, int nst, bool coarse, bool lines_y, bool lines_yt, int nb = 1)
{
	L.nx = nx; L.ny = ny; L.nz = nd == 3 ? nz : 1;
	L.II = nx + 2; L.JJ = ny + 2; L.KK = nd == 3 ? nz + 2 : 1;
	L.nst = nst;
	L.npts = (size_t)L.II * L.JJ * L.KK;
	L.res = dalloc(L.npts * nb);
	L.SOR0 = dalloc(L.npts * 2);
	if (lines_y) {
		L.SOR1 = dalloc(L.npts * 2);
		L.yscr = dalloc(ylines_scratch_doubles(L.II, L.JJ));
	}
	if (lines_yt) {
		L.At = dalloc(L.npts * nst);
		L.bt = dalloc(L.npts * nb);
		L.xt = dalloc(L.npts * nb);
	}
	if (coarse) {
		L.A = dalloc(L.npts * nst);
		L.P = dalloc(L.npts * (nd == 3 ? 26 : 8));
		L.x = dalloc(L.npts * nb);
		L.b = dalloc(L.npts * nb);
	}
}

// clear a level array inside the cycle.  The library's own kernel; CEDAR_AMD_DEBUG_MEMSET=1 switches back to
// hipMemsetAsync to reproduce the runtime defect recorded in DESIGN.md section 7 (tools/memset_rootcause.py).
void clear(real_t *p, size_t n, hipStream_t st)
{
	static const bool dbg = getenv("CEDAR_AMD_DEBUG_MEMSET") && atoi(getenv("CEDAR_AMD_DEBUG_MEMSET")) != 0;
	if (dbg) CEDAR_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(real_t), st));
	else zero_fill(p, n, st);
}

// Row-interleaved solve copy of a 27-point level (common.h Op3): read by relax and residual instead of the fourteen
// Cedar-layout planes.  Measured (profiles/r02_experiment_ilv_layout_ab.log, r02_experiment_ilv_levels.log): 512^3 relax
// sweep -7 %, V-cycle -3.8 % with the copy on level 0; slower on levels of 256^3 and below => same threshold as the
// plane-fused pass.  CEDAR_AMD_ILV: 0 never, 1 every 27-point level, n >= 2 levels with at least n rows; default 320.
// The copy costs 16/14 of the operator again (17.9 GB at 512^3): levels whose copy does not fit beside a 10 % reserve
// of the card stay on the Cedar layout (1024^3 on one GPU).
bool ilv_wanted(const Level &L)
{
	const char *e = getenv("CEDAR_AMD_ILV");
	const int mode = e ? atoi(e) : 320;
	if (mode <= 0 || (L.II - 2 + 1) / 2 > 512) return false;
	if (!(mode == 1 || L.ny >= mode)) return false;
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
	return ilv_doubles(L.II, L.JJ, L.KK) * sizeof(real_t) + tot / 10 < fr;
}

void residual(const cedar_amd_solver *s, const Level &L, const real_t *x, const real_t *b, real_t *r, hipStream_t st)
{
	if (s->nd == 2) residual2(L.A, b, x, r, L.II, L.JJ, L.nst, st, Batch{s->nb, L.npts});
	else if (L.Ailv) residual27_op(op3_ilv(L.Ailv, L.II, L.JJ, L.KK), b, x, r, L.II, L.JJ, L.KK, st);
	else residual3(L.A, b, x, r, L.II, L.JJ, L.KK, L.nst, st);
}

// one y-line sweep: on the transposed arrays when the level keeps them
void lines_y(const Level &L, real_t *x, const real_t *b, const real_t *sor, int updown, int ipn, hipStream_t st, Batch bt = Batch())
{
	if (!L.At) { // (batches always keep the transposed arrays: cedar_amd_solver_create)
		relax_lines_y(L.A, b, x, sor, L.yscr, L.II, L.JJ, L.nst, updown, st, ipn);
		return;
	}
	if (!L.bt_fresh) {
		transpose2(b, L.bt, L.II, L.JJ, st, bt);
		L.bt_fresh = true;
	}
	relax_lines_yt(L.At, L.bt, x, L.xt, sor, L.II, L.JJ, L.nst, updown, st, L.PFy, bt);
}

// ------------------------------------------------------------------ plane relaxation
void cycle_on(cedar_amd_solver *s, real_t *x, const real_t *b, hipStream_t st);
bool graph_prepare(cedar_amd_solver *s, real_t *x, const real_t *b, hipStream_t st);
bool graph_ready(const cedar_amd_solver *s, const real_t *x, const real_t *b);
double l2_dev(cedar_amd_solver *s, const Level &L, const real_t *v);
void residual(const cedar_amd_solver *s, const Level &L, const real_t *x, const real_t *b, real_t *r, hipStream_t st);

// a solver on the same operator as `proto` that shares its set-up products and owns only its vectors
cedar_amd_solver *clone_vectors(const cedar_amd_solver *proto)
{
	cedar_amd_solver *c = new cedar_amd_solver;
	c->nd = proto->nd; c->st = proto->st; c->use_graph = proto->use_graph;
	c->lv = proto->lv; // pointers to the shared products; the vectors are replaced below
	for (size_t l = 0; l < c->lv.size(); l++) {
		Level &L = c->lv[l];
		L.shared = true; L.ownA = false;
		L.res = dalloc(L.npts);
		if (L.yscr) L.yscr = dalloc(ylines_scratch_doubles(L.II, L.JJ));
		if (L.bt) { L.bt = dalloc(L.npts); L.xt = dalloc(L.npts); }
		L.bt_fresh = false;
		if (l > 0) { L.x = dalloc(L.npts); L.b = dalloc(L.npts); }
	}
	c->ABD = proto->ABD; c->shared_abd = true;
	c->nabd1 = proto->nabd1; c->nabd2 = proto->nabd2;
	c->bbd = dalloc(c->nabd2);
	c->red = dalloc(4100);
	CEDAR_HIP_CHECK(hipMalloc((void **)&c->dinfo, 64));
	return c;
}

// kman->setup<plane_relax<dir>>(so): the 2D operator, one full 2D set-up, vector-only clones for the other instances
PlaneSet *planes_setup(int dir, const real_t *so3, int II, int JJ, int KK, int nst, const cedar_amd_settings &pst, hipStream_t st)
{
	PlaneSet *ps = new PlaneSet;
	ps->dir = dir;
	ps->I2 = dir == 2 ? JJ : II;
	ps->J2 = dir == 0 ? JJ : KK;
	ps->np = (dir == 0 ? KK : dir == 1 ? JJ : II) - 2;
	ps->P2 = (size_t)ps->I2 * ps->J2;
	const int nst2 = nst == 14 ? 5 : 3, ninst = (ps->np + 1) / 2;
	ps->so2 = dalloc_raw(ps->P2 * nst2);
	plane_operator(dir, nst, so3, ps->so2, II, JJ, KK, st);
	ps->x2s = dalloc(ps->P2 * ninst);
	ps->b2s = dalloc(ps->P2 * ninst); // ghost entries stay zero: plane_gather writes interiors only
	// one cycle per plane (the reference's default plane configuration): the planes of a colour run as ONE batch through
	// the 2D kernels (common.h Batch) -- the hierarchy is shared anyway, only the vectors differ.  CEDAR_AMD_PLANE_BATCH=0,
	// or a plane configuration with max-iter > 1 (per-plane early exit), keeps one solver instance per pair of planes.
	const char *eb = getenv("CEDAR_AMD_PLANE_BATCH");
	const bool want_batch = pst.max_iter == 1 && !(eb && atoi(eb) == 0);
	cedar_amd_solver *first = solver_create(2, ps->I2 - 2, ps->J2 - 2, 1, nst2, ps->so2, 1, &pst, want_batch ? ninst : 1);
	if (!first) {
		(void)hipFree(ps->so2); (void)hipFree(ps->x2s); (void)hipFree(ps->b2s);
		delete ps;
		return nullptr;
	}
	ps->inst.push_back(first);
	if (first->nb_alloc < ninst)
		for (int q = 1; q < ninst; q++) ps->inst.push_back(clone_vectors(first));
	return ps;
}

void planes_destroy(PlaneSet *ps)
{
	if (!ps) return;
	for (size_t q = ps->inst.size(); q-- > 0;) cedar_amd_solver_destroy(ps->inst[q]); // clones before the owner of the products
	(void)hipFree(ps->so2); (void)hipFree(ps->x2s); (void)hipFree(ps->b2s);
	delete ps;
}

// relax_planes (relax_planes.h:36-72): DOWN = the odd planes (1, 3, ..), then the even ones; UP = even, then odd.
// Per colour: gather all its planes and right-hand sides (one launch), one 2D solve per plane, scatter (one launch).
// Plane configuration max-iter 1 (the reference's default): the solve is one V-cycle whatever the norms are, replayed
// from the instance's graph on a side stream; otherwise multilevel::solve's loop with its host-side tolerance test.
void planes_relax(cedar_amd_solver *s3, PlaneSet &ps, const Level &L, real_t *x, const real_t *b, int updown, hipStream_t st)
{
	const int order[2] = {updown == BMG_DOWN ? 1 : 2, updown == BMG_DOWN ? 2 : 1};
	for (int c = 0; c < 2; c++) {
		const int beg = order[c];
		const int n = beg > ps.np ? 0 : (ps.np - beg) / 2 + 1;
		if (n == 0) continue;
		plane_gather(ps.dir, L.nst, L.A, x, b, ps.x2s, ps.b2s, L.II, L.JJ, L.KK, beg, n, st);
		const int maxit = ps.inst[0]->st.max_iter;
		if (maxit == 1 && ps.inst[0]->nb_alloc >= n) { // the colour as one batch: one graph replay on the solver's stream
			ps.inst[0]->nb = n;
			cycle_on(ps.inst[0], ps.x2s, ps.b2s, st);
		} else if (maxit == 1) {
			const int S = (int)s3->pstreams.size() < n ? (int)s3->pstreams.size() : n;
			if (S <= 1) {
				for (int q = 0; q < n; q++) cycle_on(ps.inst[q], ps.x2s + ps.P2 * q, ps.b2s + ps.P2 * q, st);
			} else {
				// Every instance's graph is recorded and instantiated BEFORE the first replay of this colour is launched,
				// with the device idle: capture / hipGraphInstantiate never run beside replays in flight on the side streams
				// (the first visit used to interleave them; DESIGN.md section 6, the rocprofv3 abort of round 2).
				bool missing = false;
				for (int q = 0; q < n; q++)
					missing = missing || !graph_ready(ps.inst[q], ps.x2s + ps.P2 * q, ps.b2s + ps.P2 * q);
				if (missing) {
					CEDAR_HIP_CHECK(hipDeviceSynchronize());
					for (int q = 0; q < n; q++)
						if (ps.inst[q]->use_graph) graph_prepare(ps.inst[q], ps.x2s + ps.P2 * q, ps.b2s + ps.P2 * q, s3->pstreams[q % S]);
				}
				CEDAR_HIP_CHECK(hipEventRecord(s3->pfork, st));
				for (int t = 0; t < S; t++) CEDAR_HIP_CHECK(hipStreamWaitEvent(s3->pstreams[t], s3->pfork, 0));
				for (int q = 0; q < n; q++) cycle_on(ps.inst[q], ps.x2s + ps.P2 * q, ps.b2s + ps.P2 * q, s3->pstreams[q % S]);
				for (int t = 0; t < S; t++) {
					CEDAR_HIP_CHECK(hipEventRecord(s3->pevents[t], s3->pstreams[t]));
					CEDAR_HIP_CHECK(hipStreamWaitEvent(st, s3->pevents[t], 0));
				}
			}
		} else {
			for (int q = 0; q < n; q++) { // multilevel.h:277-298
				cedar_amd_solver *p = ps.inst[q];
				Level &L2 = p->lv[0];
				real_t *x2 = ps.x2s + ps.P2 * q;
				const real_t *b2 = ps.b2s + ps.P2 * q;
				residual(p, L2, x2, b2, L2.res, st);
				const double res0 = l2_dev(p, L2, L2.res);
				for (int it = 0; it < maxit; it++) {
					cycle_on(p, x2, b2, st);
					residual(p, L2, x2, b2, L2.res, st);
					if (l2_dev(p, L2, L2.res) / res0 < p->st.tol) break;
				}
			}
		}
		plane_scatter(ps.dir, ps.x2s, x, L.II, L.JJ, L.KK, beg, n, st);
	}
}

void smooth(const cedar_amd_solver *s, const Level &L, real_t *x, const real_t *b, int updown, int n, hipStream_t st)
{
	if (s->nd == 2 && s->st.ibc == 0 && s->st.relaxation >= CEDAR_AMD_RELAX_LINE_X && s->st.relaxation <= CEDAR_AMD_RELAX_LINE_XY
	    && lines_small_ok(L.II, L.JJ)) {
		// a small level: every sweep of this visit in one launch, the level resident in LDS (lines_small.hip)
		const int kind = s->st.relaxation == CEDAR_AMD_RELAX_LINE_X ? 1 : s->st.relaxation == CEDAR_AMD_RELAX_LINE_Y ? 2 : 3;
		relax_lines_small(L.A, b, x, L.SOR0, kind == 2 ? L.SOR0 : L.SOR1, L.II, L.JJ, L.nst, kind, updown, n, st, Batch{s->nb, L.npts});
		return;
	}
	if (s->nd == 2 && s->st.ibc == 0 && s->st.relaxation == CEDAR_AMD_RELAX_POINT && lines_small_ok(L.II, L.JJ)) {
		relax_points_small(L.A, b, x, L.SOR0, L.II, L.JJ, L.nst, updown, n, st, Batch{s->nb, L.npts});
		return;
	}
	for (int it = 0; it < n; it++) {
		if (s->nd == 3 && s->st.relaxation >= CEDAR_AMD_RELAX_PLANE_XY) { // multilevel.h:179-189, :208-218
			static const int down[3] = {0, 2, 1}, up[3] = {1, 2, 0}; // xy, yz, xz on the way down; xz, yz, xy on the way up
			for (int t = 0; t < 3; t++) {
				const int d = updown == BMG_DOWN ? down[t] : up[t];
				if (L.pl[d]) planes_relax(const_cast<cedar_amd_solver *>(s), *L.pl[d], L, x, b, updown, st);
			}
			continue;
		}
		if (s->nd == 3 && s->st.ibc) {
			// the ghosts of x hold the periodic image after any sweep, after interp_add, and on a coarse level (x starts
			// from zero); the first pre-smoothing sweep on level 0 sees the caller's x and makes no such assumption
			const bool consistent = it > 0 || updown == BMG_UP || &L != &s->lv[0];
			relax3_gs_per(L.A, b, x, L.SOR0, L.II, L.JJ, L.KK, L.nst, updown, s->st.ibc, st, consistent);
			continue;
		}
		if (s->nd == 3) {
			if (L.Ailv) relax3_gs27_op(op3_ilv(L.Ailv, L.II, L.JJ, L.KK), b, x, L.II, L.JJ, L.KK, updown, st, L.T);
			else if (L.T) relax3_gs27_op(op3_cedar(L.A, L.SOR0, L.II, L.JJ, L.KK), b, x, L.II, L.JJ, L.KK, updown, st, L.T);
			else relax3_gs(L.A, b, x, L.SOR0, L.II, L.JJ, L.KK, L.nst, updown, st);
			continue;
		}
		const int ipn = s->st.ibc;
		if (ipn && s->st.relaxation == CEDAR_AMD_RELAX_POINT) { // periodic point relaxation
			relax2_gs_per(L.A, b, x, L.SOR0, L.II, L.JJ, L.nst, updown, ipn, st);
			continue;
		}
		const Batch bt{s->nb, L.npts};
		switch (s->st.relaxation) {
		case CEDAR_AMD_RELAX_POINT:
			// nine-point levels of at least 4096 rows: the band-fused sweep with inter-row partial sums (relax2d.hip)
			if (L.nst == 5 && relax2_psum_wanted(L.II, L.JJ)) relax2_gs9_psum(L.A, b, x, L.SOR0, L.II, L.JJ, updown, st, bt);
			else relax2_gs(L.A, b, x, L.SOR0, L.II, L.JJ, L.nst, updown, st, bt);
			break;
		case CEDAR_AMD_RELAX_LINE_X: relax_lines_x(L.A, b, x, L.SOR0, L.II, L.JJ, L.nst, updown, st, ipn, L.PFx, bt); break;
		case CEDAR_AMD_RELAX_LINE_Y: lines_y(L, x, b, L.SOR0, updown, ipn, st, bt); break;
		default:
			if (updown == BMG_DOWN) {
				relax_lines_x(L.A, b, x, L.SOR0, L.II, L.JJ, L.nst, updown, st, ipn, L.PFx, bt);
				lines_y(L, x, b, L.SOR1, updown, ipn, st, bt);
			} else {
				lines_y(L, x, b, L.SOR1, updown, ipn, st, bt);
				relax_lines_x(L.A, b, x, L.SOR0, L.II, L.JJ, L.nst, updown, st, ipn, L.PFx, bt);
			}
		}
	}
}

void coarse_solve(cedar_amd_solver *s, real_t *x, const real_t *b, hipStream_t st)
{
	const Level &C = s->lv.back();
	if (s->nd == 2 && s->st.ibc) solve_cg2_per(x, b, C.II, C.JJ, s->ABD, s->bbd, s->nabd1, s->st.ibc, st);
	else if (s->nd == 2) solve_cg2(x, b, C.II, C.JJ, s->ABD, s->bbd, s->nabd1, s->nabd2, st, Batch{s->nb, C.npts});
	else if (s->st.ibc) solve_cg3_per(x, b, C.II, C.JJ, C.KK, s->ABD, s->bbd, s->nabd1, s->st.ibc, st);
	else solve_cg3(x, b, C.II, C.JJ, C.KK, s->ABD, s->bbd, s->nabd1, s->nabd2, st);
}

void ncycle(cedar_amd_solver *s, int lvl, real_t *x, const real_t *b, hipStream_t st)
{
	Level &L = s->lv[lvl], &K = s->lv[lvl + 1];
	L.bt_fresh = false; // b of this visit has not been transposed yet
	// a small 2D level: each half of the visit is one launch with the level resident in LDS (lines_small.hip)
	const bool small = s->nd == 2 && s->st.ibc == 0 && s->st.relaxation <= CEDAR_AMD_RELAX_LINE_XY && lines_small_ok(L.II, L.JJ);
	const int kind = s->st.relaxation == CEDAR_AMD_RELAX_POINT ? 0 : s->st.relaxation == CEDAR_AMD_RELAX_LINE_X ? 1
	               : s->st.relaxation == CEDAR_AMD_RELAX_LINE_Y ? 2 : 3;
	const real_t *sory = kind == 2 ? L.SOR0 : L.SOR1;
	if (small) {
		visit_small(1, L.A, b, x, L.res, L.SOR0, sory, L.II, L.JJ, L.nst, kind, s->st.nrelax_pre, K.P, K.b, K.x, K.II, K.JJ, st,
		            Batch{s->nb, L.npts}, Batch{s->nb, K.npts});
	} else {
	smooth(s, L, x, b, BMG_DOWN, s->st.nrelax_pre, st);
	residual(s, L, x, b, L.res, st);
	if (s->nd == 2 && s->st.ibc) restrict2_per(L.res, K.b, K.P, L.II, L.JJ, K.II, K.JJ, s->st.ibc, st);
	else if (s->nd == 2) restrict2(L.res, K.b, K.P, L.II, L.JJ, K.II, K.JJ, st, Batch{s->nb, L.npts}, Batch{s->nb, K.npts});
	else if (s->st.ibc) restrict3_per(L.res, K.b, K.P, L.II, L.JJ, L.KK, K.II, K.JJ, K.KK, s->st.ibc, st);
	else restrict3(L.res, K.b, K.P, L.II, L.JJ, L.KK, K.II, K.JJ, K.KK, st);
	clear(K.x, K.npts * (size_t)s->nb, st); // coarse_x.set(0.0)
	}
	if (lvl + 1 == (int)s->lv.size() - 1) coarse_solve(s, K.x, K.b, st);
	else ncycle(s, lvl + 1, K.x, K.b, st);
	if (small) {
		visit_small(0, L.A, b, x, L.res, L.SOR0, sory, L.II, L.JJ, L.nst, kind, s->st.nrelax_post, K.P, nullptr, K.x, K.II, K.JJ, st,
		            Batch{s->nb, L.npts}, Batch{s->nb, K.npts});
		return;
	}
	if (s->nd == 2 && s->st.ibc) interp_add2_per(x, K.x, L.res, L.A, K.P, K.II, K.JJ, L.II, L.JJ, s->st.ibc, st);
	else if (s->nd == 2) interp_add2(x, K.x, L.res, L.A, K.P, K.II, K.JJ, L.II, L.JJ, st, Batch{s->nb, L.npts}, Batch{s->nb, K.npts});
	else if (s->st.ibc) interp_add3_per(x, K.x, L.A, L.res, K.P, K.II, K.JJ, K.KK, L.II, L.JJ, L.KK, s->st.ibc, st);
	else interp_add3(x, K.x, L.A, L.res, K.P, K.II, K.JJ, K.KK, L.II, L.JJ, L.KK, st);
	smooth(s, L, x, b, BMG_UP, s->st.nrelax_post, st);
}

// full-multigrid cycle, include/cedar/cycle/fcycle.h:49-83: restrict the right-hand side itself down
// to the coarsest level, solve there, and on the way up interpolate (x = P x_c: interp_add onto
// x = 0 with a zero "residual") and run one V-cycle from that level.
void fmg_cycle(cedar_amd_solver *s, int lvl, real_t *x, const real_t *b, hipStream_t st)
{
	if (lvl == (int)s->lv.size() - 1) {
		coarse_solve(s, x, b, st);
		return;
	}
	Level &L = s->lv[lvl], &K = s->lv[lvl + 1];
	// periodic: the kernels of the V-cycle; the periodic restriction refreshes the ghosts of the vector it restricts
	// (restrict.f90:78-103) -- here the right-hand side itself, which the reference's binding reaches through a const_cast
	const int ibc = s->st.ibc;
	if (s->nd == 2 && ibc) restrict2_per(const_cast<real_t *>(b), K.b, K.P, L.II, L.JJ, K.II, K.JJ, ibc, st);
	else if (s->nd == 2) restrict2(b, K.b, K.P, L.II, L.JJ, K.II, K.JJ, st);
	else if (ibc) restrict3_per(const_cast<real_t *>(b), K.b, K.P, L.II, L.JJ, L.KK, K.II, K.JJ, K.KK, ibc, st);
	else restrict3(b, K.b, K.P, L.II, L.JJ, L.KK, K.II, K.JJ, K.KK, st);
	fmg_cycle(s, lvl + 1, K.x, K.b, st);
	zero_fill(x, L.npts, st);
	zero_fill(L.res, L.npts, st);
	if (s->nd == 2 && ibc) interp_add2_per(x, K.x, L.res, L.A, K.P, K.II, K.JJ, L.II, L.JJ, ibc, st);
	else if (s->nd == 2) interp_add2(x, K.x, L.res, L.A, K.P, K.II, K.JJ, L.II, L.JJ, st);
	else if (ibc) interp_add3_per(x, K.x, L.A, L.res, K.P, K.II, K.JJ, K.KK, L.II, L.JJ, L.KK, ibc, st);
	else interp_add3(x, K.x, L.A, L.res, K.P, K.II, K.JJ, K.KK, L.II, L.JJ, L.KK, st);
	ncycle(s, lvl, x, b, st);
}

void cycle_launch(cedar_amd_solver *s, real_t *x, const real_t *b, hipStream_t st)
{
	if (s->lv.size() == 1) coarse_solve(s, x, b, st); // vcycle.h:37-38, fcycle.h:42-43
	else if (s->st.cycle == 1) fmg_cycle(s, 0, x, b, st);
	else ncycle(s, 0, x, b, st);
}

// run one V-cycle on device pointers: graph replay when possible
void cycle_on(cedar_amd_solver *s, real_t *x, const real_t *b, hipStream_t st);
void cycle_dev(cedar_amd_solver *s, real_t *x, const real_t *b) { cycle_on(s, x, b, current_stream()); }

// make s->gexec the instantiated graph of one cycle on (x, b) with the current batch count; returns true if it had to be
// recorded.  `st`: the stream earlier replays of this solver's graph were launched on (drained before a stale graph goes).
bool graph_prepare(cedar_amd_solver *s, real_t *x, const real_t *b, hipStream_t st)
{
	if (s->gexec && (s->gx != x || s->gb != b)) {
		CEDAR_HIP_CHECK(hipStreamSynchronize(st)); // a replay of the old graph may still be running
		CEDAR_HIP_CHECK(hipGraphExecDestroy(s->gexec));
		s->gexec = nullptr;
		if (s->gexec2) CEDAR_HIP_CHECK(hipGraphExecDestroy(s->gexec2));
		s->gexec2 = nullptr;
	}
	if (s->gexec && s->gnb != s->nb) { // the other captured batch count (the two colours of a plane sweep)
		if (s->gexec2 && s->gnb2 == s->nb) {
			std::swap(s->gexec, s->gexec2);
			std::swap(s->gnb, s->gnb2);
		} else {
			if (s->gexec2) CEDAR_HIP_CHECK(hipGraphExecDestroy(s->gexec2));
			s->gexec2 = s->gexec; s->gnb2 = s->gnb;
			s->gexec = nullptr;
		}
	}
	if (s->gexec) return false;
	{
		// one capture stream for the process: captures are recorded synchronously (thread-local mode), and plane
		// relaxation keeps hundreds of small solvers whose graphs are all recorded through here
		static hipStream_t capture_stream = nullptr;
		if (!capture_stream) CEDAR_HIP_CHECK(hipStreamCreateWithFlags(&capture_stream, hipStreamNonBlocking));
		s->gstream = capture_stream;
		// one eager cycle first is NOT wanted (it would change x); capture records without executing
		CEDAR_HIP_CHECK(hipStreamSynchronize(st));
		hipGraph_t g = nullptr;
		CEDAR_HIP_CHECK(hipStreamBeginCapture(s->gstream, hipStreamCaptureModeThreadLocal));
		cycle_launch(s, x, b, s->gstream);
		CEDAR_HIP_CHECK(hipStreamEndCapture(s->gstream, &g));
		CEDAR_HIP_CHECK(hipGraphInstantiate(&s->gexec, g, nullptr, nullptr, 0));
		CEDAR_HIP_CHECK(hipGraphDestroy(g));
		s->gx = x; s->gb = b; s->gnb = s->nb;
	}
	return true;
}

bool graph_ready(const cedar_amd_solver *s, const real_t *x, const real_t *b)
{
	return !s->use_graph || (s->gexec && s->gx == x && s->gb == b && s->gnb == s->nb);
}

void cycle_on(cedar_amd_solver *s, real_t *x, const real_t *b, hipStream_t st)
{
	if (!s->use_graph) {
		cycle_launch(s, x, b, st);
		return;
	}
	graph_prepare(s, x, b, st);
	CEDAR_HIP_CHECK(hipGraphLaunch(s->gexec, st));
}

double l2_dev(cedar_amd_solver *s, const Level &L, const real_t *v)
{
	sumsq_interior(v, L.II, L.JJ, L.KK, s->red, s->red + 4096, current_stream());
	double ss = 0;
	CEDAR_HIP_CHECK(hipMemcpyAsync(&ss, s->red + 4096, sizeof(double), hipMemcpyDeviceToHost, current_stream()));
	CEDAR_HIP_CHECK(hipStreamSynchronize(current_stream()));
	return std::sqrt(ss);
}

} // namespace

extern "C" {

void cedar_amd_default_settings(cedar_amd_settings *s)
{
	s->relaxation = CEDAR_AMD_RELAX_POINT;
	s->nrelax_pre = 2;
	s->nrelax_post = 1;
	s->num_levels = -1;
	s->max_iter = 10;
	s->tol = 1e-8;
	s->min_coarse = 3;
	s->cycle = 0;
	s->ibc = 0;
	s->plane_relaxation = CEDAR_AMD_RELAX_LINE_XY; // src/kernel_params.cc:72-78
	s->plane_nrelax_pre = 2;
	s->plane_nrelax_post = 1;
	s->plane_max_iter = 1;
	s->plane_min_coarse = 3;
	s->plane_tol = 1e-8;
}

cedar_amd_solver *cedar_amd_solver_create(int nd, len_t nx, len_t ny, len_t nz, int nstencil,
                                          const real_t *so, int own_device_so,
                                          const cedar_amd_settings *settings)
{
	return solver_create(nd, nx, ny, nz, nstencil, so, own_device_so, settings, 1);
}

} // extern "C"

static cedar_amd_solver *solver_create(int nd, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so, int own_device_so,
                                       const cedar_amd_settings *settings, int batch)
{
	hipStream_t st = current_stream();
	cedar_amd_solver *s = new cedar_amd_solver;
	s->nd = nd;
	s->nb_alloc = batch < 1 ? 1 : batch;
	if (settings) s->st = *settings;
	else cedar_amd_default_settings(&s->st);
	const bool planes = s->st.relaxation >= CEDAR_AMD_RELAX_PLANE_XY && s->st.relaxation <= CEDAR_AMD_RELAX_PLANE_XYZ;
	if ((nd == 3 && s->st.relaxation != CEDAR_AMD_RELAX_POINT && !planes) || (nd == 2 && planes)
	    || (planes && (s->st.ibc != 0 || s->st.plane_relaxation >= CEDAR_AMD_RELAX_PLANE_XY))) {
		char msg[] = "cedar_amd_solver_create: relaxation must be point / line-x / line-y / line-xy in 2D and point or "
		             "plane-xy / -xz / -yz / -xyz in 3D (planes: Dirichlet boundaries, a 2D relaxation in the plane "
		             "configuration); point relaxation is used";
		print_error(msg);
		s->st.relaxation = CEDAR_AMD_RELAX_POINT;
	}
	if (s->st.relaxation >= CEDAR_AMD_RELAX_PLANE_XY) {
		// the plane solves of a colour run on side streams and replay graphs of their own: the 3D cycle is launched
		// eagerly (capturing it would nest graph launches)
		s->use_graph = false;
		const char *es = getenv("CEDAR_AMD_PLANE_STREAMS");
		int S = es ? atoi(es) : 8;
		if (S < 1) S = 1;
		if (S > 64) S = 64;
		s->pstreams.resize(S);
		s->pevents.resize(S);
		for (int t = 0; t < S; t++) {
			CEDAR_HIP_CHECK(hipStreamCreateWithFlags(&s->pstreams[t], hipStreamNonBlocking));
			CEDAR_HIP_CHECK(hipEventCreateWithFlags(&s->pevents[t], hipEventDisableTiming));
		}
		CEDAR_HIP_CHECK(hipEventCreateWithFlags(&s->pfork, hipEventDisableTiming));
	}
	if (s->st.ibc != 0) {
		// periodic boundary conditions (V- and F-cycles).  2D: ibc 1..3, point relaxation keeps a row in the default LDS
		// window; 3D: ibc 1..3, 5..8 (BMG_get_bc.f90:13-20), even extents in the periodic directions on every level
		// that is coarsened (checked below, once the level sizes are known)
		const bool ok2 = nd == 2 && s->st.ibc >= 1 && s->st.ibc <= 3
		                 && (s->st.relaxation != CEDAR_AMD_RELAX_POINT || (size_t)(nx + 2) * sizeof(real_t) <= 64 * 1024);
		const bool ok3 = nd == 3 && s->st.ibc > 0 && periodic3_code_ok(s->st.ibc);
		if (!ok2 && !ok3) {
			char msg[] = "cedar_amd_solver_create: periodic boundary conditions are implemented for the definite periodic codes "
			             "(2D: ibc 1..3, point relaxation: rows up to 8190 points; 3D: ibc 1..3, 5..8); no solver created";
			print_error(msg);
			delete s;
			return nullptr;
		}
	}
	if (const char *e = getenv("CEDAR_AMD_NO_GRAPH")) s->use_graph = s->use_graph && !(e[0] == '1');
	int nlev = compute_num_levels(nd, nx, ny, nz, s->st.min_coarse);
	if (s->st.num_levels > 0) {
		if (s->st.num_levels > nlev) {
			char msg[] = "too many levels specified";
			print_error(msg);
		} else
			nlev = s->st.num_levels;
	}
	s->lv.resize(nlev);
	const bool ly = nd == 2 && (s->st.relaxation == CEDAR_AMD_RELAX_LINE_XY || s->st.relaxation == CEDAR_AMD_RELAX_LINE_Y);
	// y-lines run on transposed arrays unless periodic (the wraps and the cyclic closure stay on relax_lines_y);
	// CEDAR_AMD_YLINES_TRANSPOSED=0 keeps the gather / solve / scatter pipeline for cross-checks
	const char *eyt = getenv("CEDAR_AMD_YLINES_TRANSPOSED");
	const bool lyt = ly && s->st.ibc == 0 && !(eyt && atoi(eyt) == 0);
	if (s->nb_alloc > 1 && (nd != 2 || s->st.ibc != 0 || s->st.cycle != 0 || (ly && !lyt))) s->nb_alloc = 1; // batches: see the struct
	level_init(s->lv[0], nd, (int)nx, (int)ny, (int)nz, nstencil, false, ly, lyt, s->nb_alloc);
	Level &F0 = s->lv[0];
	if (own_device_so && is_device_ptr(so)) {
		F0.A = const_cast<real_t *>(so);
		F0.ownA = false;
	} else {
		F0.A = dalloc(F0.npts * nstencil);
		CEDAR_HIP_CHECK(hipMemcpyAsync(F0.A, so, F0.npts * nstencil * sizeof(real_t),
		                               is_device_ptr(so) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
	}
	for (int l = 1; l < nlev; l++) {
		const Level &F = s->lv[l - 1];
		int nxc = (int)((F.nx - 1) / 2. + 1), nyc = (int)((F.ny - 1) / 2. + 1);
		int nzc = nd == 3 ? (int)((F.nz - 1) / 2. + 1) : 1;
		level_init(s->lv[l], nd, nxc, nyc, nzc, nd == 3 ? 14 : 5, true, ly, lyt, s->nb_alloc);
	}
	const Level &C = s->lv.back();
	// periodic: the coarsest operator is stored dense (include/cedar/2d/solver.h:110-114)
	if (nd == 2) { s->nabd1 = s->st.ibc ? C.nx * C.ny : C.nx + 2; s->nabd2 = C.nx * C.ny; }
	else { s->nabd1 = s->st.ibc ? C.nx * C.ny * C.nz : C.nx * (C.ny + 1) + 2; s->nabd2 = C.nx * C.ny * C.nz; } // 3d/solver.h:118-121
	if (nd == 3 && s->st.ibc) {
		const int c = s->st.ibc;
		const bool px = c == 2 || c == 3 || c == 6 || c == 8, py = c == 1 || c == 3 || c == 7 || c == 8, pz = c >= 5;
		bool ok = (size_t)s->nabd2 <= 2048; // dense factor in ONE workgroup: n^2 doubles, n^3/3 flops (2048: ~3 GFlop, about a second)
		for (int l = 0; l + 1 < nlev; l++)
			ok = ok && !(px && (s->lv[l].nx & 1)) && !(py && (s->lv[l].ny & 1)) && !(pz && (s->lv[l].nz & 1));
		if (!ok) {
			char msg[] = "cedar_amd_solver_create: 3D periodic boundary conditions need an even extent in every periodic "
			             "direction on each level that is coarsened (choose the extents or num_levels accordingly) and at "
			             "most 2048 unknowns on the coarsest level (it is factored densely by one workgroup: use more levels); no solver created";
			print_error(msg);
			cedar_amd_solver_destroy(s); // releases the levels allocated so far
			return nullptr;
		}
	}
	s->ABD = dalloc((size_t)s->nabd1 * s->nabd2);
	s->bbd = dalloc((size_t)s->nabd2 * s->nb_alloc);
	s->red = dalloc(4100);
	CEDAR_HIP_CHECK(hipMalloc((void **)&s->dinfo, 64));
	CEDAR_HIP_CHECK(hipMemsetAsync(s->dinfo, 0, 64, st));

	for (int l = 0; l < nlev - 1; l++) {
		Level &F = s->lv[l], &K = s->lv[l + 1];
		if (nd == 2) {
			int ifd = F.nst == 3;
			if (s->st.ibc) {
				setup_interp2_per(F.A, K.P, F.II, F.JJ, K.II, K.JJ, ifd, s->st.ibc, st);
				galerkin2_per(F.A, K.A, K.P, F.II, F.JJ, K.II, K.JJ, ifd, s->st.ibc, st);
			} else {
				setup_interp2(F.A, K.P, F.II, F.JJ, K.II, K.JJ, ifd, st);
				galerkin2(F.A, K.A, K.P, F.II, F.JJ, K.II, K.JJ, ifd, st);
			}
			switch (s->st.relaxation) {
			case CEDAR_AMD_RELAX_POINT: setup_recip(F.A, F.SOR0 + F.npts, F.II, F.JJ, 1, st); break;
			case CEDAR_AMD_RELAX_LINE_X: setup_lines_x(F.A, F.SOR0, F.II, F.JJ, st, s->st.ibc == 2 || s->st.ibc == 3); break;
			case CEDAR_AMD_RELAX_LINE_Y: setup_lines_y(F.A, F.SOR0, F.II, F.JJ, st, s->st.ibc == 1 || s->st.ibc == 3); break;
			default:
				setup_lines_x(F.A, F.SOR0, F.II, F.JJ, st, s->st.ibc == 2 || s->st.ibc == 3);
				setup_lines_y(F.A, F.SOR1, F.II, F.JJ, st, s->st.ibc == 1 || s->st.ibc == 3);
			}
			if (F.At) setup_lines_yt(F.A, F.At, F.II, F.JJ, F.nst, st);
			if (s->st.ibc == 0 && !(getenv("CEDAR_AMD_LINE_PERM") && atoi(getenv("CEDAR_AMD_LINE_PERM")) == 0)) {
				// scan-ordered factor copies for the long-line kernel (Dirichlet; periodic lines keep the SOR reads)
				const int rx = s->st.relaxation;
				if (rx == CEDAR_AMD_RELAX_LINE_X || rx == CEDAR_AMD_RELAX_LINE_XY) {
					const size_t nd_ = lines_permuted_doubles(F.II - 2, F.JJ - 2);
					if (nd_) { F.PFx = dalloc_raw(nd_); lines_permute(F.SOR0, F.PFx, F.II - 2, F.II, F.JJ - 2, F.npts, st); }
				}
				if (F.At) { // y factors: SOR(JJ,II,2) in SOR0 (line-y) or SOR1 (line-xy)
					const real_t *sy = rx == CEDAR_AMD_RELAX_LINE_Y ? F.SOR0 : F.SOR1;
					const size_t nd_ = lines_permuted_doubles(F.JJ - 2, F.II - 2);
					if (nd_) { F.PFy = dalloc_raw(nd_); lines_permute(sy, F.PFy, F.JJ - 2, F.JJ, F.II - 2, F.npts, st); }
				}
			}
		} else {
			int ifd = F.nst == 4;
			if (s->st.ibc) {
				setup_interp3_per(F.A, K.P, F.II, F.JJ, F.KK, K.II, K.JJ, K.KK, ifd, s->st.ibc, st);
				galerkin3_per(F.A, K.A, K.P, F.II, F.JJ, F.KK, K.II, K.JJ, K.KK, ifd, s->st.ibc, st);
			} else {
				setup_interp3(F.A, K.P, F.II, F.JJ, F.KK, K.II, K.JJ, K.KK, ifd, st);
				galerkin3(F.A, K.A, K.P, F.II, F.JJ, F.KK, K.II, K.JJ, K.KK, ifd, st);
			}
			if (s->st.relaxation >= CEDAR_AMD_RELAX_PLANE_XY) { // multilevel.h:149-159
				cedar_amd_settings pst;
				cedar_amd_default_settings(&pst);
				pst.relaxation = s->st.plane_relaxation; pst.nrelax_pre = s->st.plane_nrelax_pre;
				pst.nrelax_post = s->st.plane_nrelax_post; pst.max_iter = s->st.plane_max_iter;
				pst.tol = s->st.plane_tol; pst.min_coarse = s->st.plane_min_coarse;
				for (int d = 0; d < 3; d++)
					if (s->st.relaxation == CEDAR_AMD_RELAX_PLANE_XYZ || s->st.relaxation == CEDAR_AMD_RELAX_PLANE_XY + d)
						F.pl[d] = planes_setup(d, F.A, F.II, F.JJ, F.KK, F.nst, pst, st);
				continue;
			}
			setup_recip(F.A, F.SOR0 + F.npts, F.II, F.JJ, F.KK, st);
			if (F.nst == 14 && !s->st.ibc && ilv_wanted(F)) {
				F.Ailv = dalloc_raw(ilv_doubles(F.II, F.JJ, F.KK));
				ilv_build(F.A, F.SOR0 + F.npts, F.Ailv, F.II, F.JJ, F.KK, st);
			}
			// partial-sum scratch of the relax sweep: T is read only where the sweep wrote it, not cleared
			if (F.nst == 14 && !s->st.ibc && relax3_psum_wanted(F.II, F.JJ, F.KK)) F.T = dalloc_raw(F.npts);
		}
	}
	if (nd == 2 && s->st.ibc) setup_cg2_per(C.A, C.II, C.JJ, C.nst, s->ABD, s->nabd1, s->st.ibc, s->dinfo, st);
	else if (nd == 2) setup_cg2(C.A, C.II, C.JJ, C.nst, s->ABD, s->nabd1, s->nabd2, s->dinfo, st);
	else if (s->st.ibc) setup_cg3_per(C.A, C.II, C.JJ, C.KK, s->ABD, s->nabd1, s->st.ibc, s->dinfo, st);
	else setup_cg3(C.A, C.II, C.JJ, C.KK, C.nst, s->ABD, s->nabd1, s->nabd2, s->dinfo, st);
	int info = 0;
	CEDAR_HIP_CHECK(hipMemcpyAsync(&info, s->dinfo, sizeof(int), hipMemcpyDeviceToHost, st));
	CEDAR_HIP_CHECK(hipStreamSynchronize(st));
	if (info != 0) {
		char msg[] = "Coarse grid Cholesky decomp failed!";
		print_error(msg);
	}
	launch_check("cedar_amd_solver_create");
	return s;
}

extern "C" {

void cedar_amd_solver_destroy(cedar_amd_solver *s)
{
	if (!s) return;
	CEDAR_HIP_CHECK(hipDeviceSynchronize());
	if (s->gexec) CEDAR_HIP_CHECK(hipGraphExecDestroy(s->gexec));
	if (s->gexec2) CEDAR_HIP_CHECK(hipGraphExecDestroy(s->gexec2));
	for (size_t l = 0; l < s->lv.size(); l++) {
		Level &L = s->lv[l];
		for (int d = 0; d < 3; d++) planes_destroy(L.pl[d]);
		if (L.ownA) (void)hipFree(L.A);
		if (!L.shared) {
			(void)hipFree(L.P); (void)hipFree(L.SOR0); (void)hipFree(L.SOR1); (void)hipFree(L.At); (void)hipFree(L.Ailv);
			(void)hipFree(L.PFx); (void)hipFree(L.PFy);
		}
		(void)hipFree(L.res); (void)hipFree(L.yscr); (void)hipFree(L.bt); (void)hipFree(L.xt); (void)hipFree(L.T);
		if (l > 0) { (void)hipFree(L.x); (void)hipFree(L.b); }
	}
	if (!s->shared_abd) (void)hipFree(s->ABD);
	(void)hipFree(s->bbd); (void)hipFree(s->red); (void)hipFree(s->dinfo);
	for (auto t : s->pstreams) (void)hipStreamDestroy(t);
	for (auto e : s->pevents) (void)hipEventDestroy(e);
	if (s->pfork) (void)hipEventDestroy(s->pfork);
	delete s;
}

// every entry point below tolerates the NULL that cedar_amd_solver_create returns for unsupported settings:
// it reports through print_error and does nothing (the reference never aborts either)
static bool null_handle(const void *s, const char *who)
{
	if (s) return false;
	char msg[160];
	snprintf(msg, sizeof(msg), "%s: NULL solver handle (cedar_amd_solver_create reported why no solver was created)", who);
	print_error(msg);
	return true;
}

int cedar_amd_solver_nlevels(const cedar_amd_solver *s) { return null_handle(s, "cedar_amd_solver_nlevels") ? 0 : (int)s->lv.size(); }

void cedar_amd_solver_level_dims(const cedar_amd_solver *s, int lvl, len_t *nx, len_t *ny, len_t *nz)
{
	*nx = *ny = *nz = 0;
	if (null_handle(s, "cedar_amd_solver_level_dims") || lvl < 0 || lvl >= (int)s->lv.size()) return;
	*nx = s->lv[lvl].nx; *ny = s->lv[lvl].ny; *nz = s->lv[lvl].nz;
}

size_t cedar_amd_solver_get(const cedar_amd_solver *s, int lvl, const char *what, real_t *out)
{
	if (null_handle(s, "cedar_amd_solver_get") || lvl < 0 || lvl >= (int)s->lv.size()) return 0;
	const Level &L = s->lv[lvl];
	const real_t *src = nullptr;
	size_t n = 0;
	if (!strcmp(what, "A")) { src = L.A; n = L.npts * L.nst; }
	else if (!strcmp(what, "P")) { src = L.P; n = L.P ? L.npts * (s->nd == 3 ? 26 : 8) : 0; }
	else if (!strcmp(what, "SOR0")) { src = L.SOR0; n = L.npts * 2; }
	else if (!strcmp(what, "SOR1")) { src = L.SOR1; n = L.SOR1 ? L.npts * 2 : 0; }
	else if (!strcmp(what, "ABD")) { src = s->ABD; n = (size_t)s->nabd1 * s->nabd2; }
	else if (!strcmp(what, "res")) { src = L.res; n = L.npts; }
	else if (!strcmp(what, "x")) { src = L.x; n = L.x ? L.npts : 0; } // coarse levels only: level 0 uses the caller's
	else if (!strcmp(what, "b")) { src = L.b; n = L.b ? L.npts : 0; }
	if (out && n) cedar_amd_memcpy_d2h(out, src, n * sizeof(real_t));
	return n;
}

// levels[lvl].A / .P / .SOR / ABD are public members of the reference's solver (include/cedar/level.h:14-41,
// include/cedar/2d/solver.h:56): a caller may replace a set-up product.  Solver-internal copies that derive from the
// array (row-interleaved solve copy, transposed y-line planes, scan-ordered line factors) are rebuilt.
size_t cedar_amd_solver_set(cedar_amd_solver *s, int lvl, const char *what, const real_t *in)
{
	if (null_handle(s, "cedar_amd_solver_set") || lvl < 0 || lvl >= (int)s->lv.size() || !in) return 0;
	Level &L = s->lv[lvl];
	hipStream_t st = current_stream();
	real_t *dst = nullptr;
	size_t n = 0;
	if (!strcmp(what, "A")) { dst = L.A; n = L.npts * L.nst; }
	else if (!strcmp(what, "P")) { dst = L.P; n = L.P ? L.npts * (s->nd == 3 ? 26 : 8) : 0; }
	else if (!strcmp(what, "SOR0")) { dst = L.SOR0; n = L.npts * 2; }
	else if (!strcmp(what, "SOR1")) { dst = L.SOR1; n = L.SOR1 ? L.npts * 2 : 0; }
	else if (!strcmp(what, "ABD")) { dst = s->ABD; n = (size_t)s->nabd1 * s->nabd2; }
	if (!dst || !n) return 0;
	CEDAR_HIP_CHECK(hipDeviceSynchronize()); // no cycle in flight while a product changes
	if (is_device_ptr(in)) cedar_amd_memcpy_d2d(dst, in, n * sizeof(real_t));
	else cedar_amd_memcpy_h2d(dst, in, n * sizeof(real_t));
	if (L.Ailv && (!strcmp(what, "A") || !strcmp(what, "SOR0")))
		ilv_build(L.A, L.SOR0 + L.npts, L.Ailv, L.II, L.JJ, L.KK, st);
	if (L.At && !strcmp(what, "A")) setup_lines_yt(L.A, L.At, L.II, L.JJ, L.nst, st);
	if (L.PFx && !strcmp(what, "SOR0")) lines_permute(L.SOR0, L.PFx, L.nx, L.II, L.ny, L.npts, st);
	if (L.PFy && !strcmp(what, s->st.relaxation == CEDAR_AMD_RELAX_LINE_Y ? "SOR0" : "SOR1"))
		lines_permute(s->st.relaxation == CEDAR_AMD_RELAX_LINE_Y ? L.SOR0 : L.SOR1, L.PFy, L.ny, L.JJ, L.nx, L.npts, st);
	CEDAR_HIP_CHECK(hipStreamSynchronize(st));
	launch_check("cedar_amd_solver_set");
	return n;
}

void cedar_amd_solver_vcycle(cedar_amd_solver *s, real_t *x, const real_t *b)
{
	if (null_handle(s, "cedar_amd_solver_vcycle")) return;
	const Level &L = s->lv[0];
	Staged sx(x, L.npts, true, true), sb(b, L.npts, true, false);
	if (sx.staged() || sb.staged()) {
		// staging buffers change from call to call: launch eagerly
		cycle_launch(s, sx.get(), sb.get(), current_stream());
	} else
		cycle_dev(s, sx.get(), sb.get());
	launch_check("cedar_amd_solver_vcycle");
}

int cedar_amd_solver_solve(cedar_amd_solver *s, const real_t *b, real_t *x, real_t *rel)
{
	if (null_handle(s, "cedar_amd_solver_solve")) return 0;
	Level &L = s->lv[0];
	Staged sx(x, L.npts, true, true), sb(b, L.npts, true, false);
	hipStream_t st = current_stream();
	residual(s, L, sx.get(), sb.get(), L.res, st);
	const double res0 = l2_dev(s, L, L.res);
	rel[0] = res0;
	int it = 0;
	for (it = 0; it < s->st.max_iter; it++) {
		cycle_dev(s, sx.get(), sb.get());
		residual(s, L, sx.get(), sb.get(), L.res, st);
		const double r = l2_dev(s, L, L.res) / res0;
		rel[it + 1] = r;
		if (r < s->st.tol) { it++; break; }
	}
	launch_check("cedar_amd_solver_solve");
	return it;
}

// ---- plane relaxation as a kernel of its own: kernels::plane_relax<stypes, rdir>::setup / run
// (include/cedar/kernels/plane_relax.h:10-33, include/cedar/3d/relax_planes.h:164-246)
struct cedar_amd_planes {
	cedar_amd_solver *host = nullptr; // carries the side streams
	PlaneSet *ps = nullptr;
	Level L;                          // extents and stencil of the 3D operator
};

cedar_amd_planes *cedar_amd_planes_create(int dir, len_t nx, len_t ny, len_t nz, int nstencil, const real_t *so,
                                          const cedar_amd_settings *plane_settings)
{
	if (dir < 0 || dir > 2 || (nstencil != 4 && nstencil != 14)) {
		char msg[] = "cedar_amd_planes_create: dir must be 0 (xy), 1 (xz) or 2 (yz), nstencil 4 or 14";
		print_error(msg);
		return nullptr;
	}
	hipStream_t st = current_stream();
	cedar_amd_settings pst;
	if (plane_settings) pst = *plane_settings;
	else { // the reference's default plane configuration, src/kernel_params.cc:72-78
		cedar_amd_default_settings(&pst);
		pst.relaxation = CEDAR_AMD_RELAX_LINE_XY;
		pst.max_iter = 1;
	}
	cedar_amd_planes *p = new cedar_amd_planes;
	p->L.nx = (int)nx; p->L.ny = (int)ny; p->L.nz = (int)nz;
	p->L.II = (int)nx + 2; p->L.JJ = (int)ny + 2; p->L.KK = (int)nz + 2;
	p->L.nst = nstencil;
	p->L.npts = (size_t)p->L.II * p->L.JJ * p->L.KK;
	p->host = new cedar_amd_solver;
	p->host->nd = 3;
	const char *es = getenv("CEDAR_AMD_PLANE_STREAMS");
	int S = es ? atoi(es) : 8;
	S = S < 1 ? 1 : S > 64 ? 64 : S;
	p->host->pstreams.resize(S);
	p->host->pevents.resize(S);
	for (int t = 0; t < S; t++) {
		CEDAR_HIP_CHECK(hipStreamCreateWithFlags(&p->host->pstreams[t], hipStreamNonBlocking));
		CEDAR_HIP_CHECK(hipEventCreateWithFlags(&p->host->pevents[t], hipEventDisableTiming));
	}
	CEDAR_HIP_CHECK(hipEventCreateWithFlags(&p->host->pfork, hipEventDisableTiming));
	{
		Staged sso(so, p->L.npts * nstencil, true, false);
		p->ps = planes_setup(dir, sso.get(), p->L.II, p->L.JJ, p->L.KK, nstencil, pst, st);
	}
	if (!p->ps) {
		cedar_amd_solver_destroy(p->host);
		delete p;
		return nullptr;
	}
	return p;
}

void cedar_amd_planes_run(cedar_amd_planes *p, const real_t *so, real_t *x, const real_t *b, int updown)
{
	if (null_handle(p, "cedar_amd_planes_run")) return;
	Staged sso(so, p->L.npts * p->L.nst, true, false), sx(x, p->L.npts, true, true), sb(b, p->L.npts, true, false);
	Level L = p->L;
	L.A = sso.get();
	planes_relax(p->host, *p->ps, L, sx.get(), sb.get(), updown, current_stream());
}

void cedar_amd_planes_destroy(cedar_amd_planes *p)
{
	if (!p) return;
	CEDAR_HIP_CHECK(hipDeviceSynchronize());
	planes_destroy(p->ps);
	cedar_amd_solver_destroy(p->host);
	delete p;
}

float cedar_amd_solver_time_vcycles(cedar_amd_solver *s, real_t *x_dev, const real_t *b_dev, int n)
{
	if (null_handle(s, "cedar_amd_solver_time_vcycles")) return 0.f;
	hipStream_t st = current_stream();
	hipEvent_t e0, e1;
	CEDAR_HIP_CHECK(hipEventCreate(&e0));
	CEDAR_HIP_CHECK(hipEventCreate(&e1));
	CEDAR_HIP_CHECK(hipEventRecord(e0, st));
	for (int i = 0; i < n; i++) cycle_dev(s, x_dev, b_dev);
	CEDAR_HIP_CHECK(hipEventRecord(e1, st));
	CEDAR_HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CEDAR_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	return ms;
}

float cedar_amd_solver_time_relax(cedar_amd_solver *s, real_t *x_dev, const real_t *b_dev, int n)
{
	if (null_handle(s, "cedar_amd_solver_time_relax")) return 0.f;
	hipStream_t st = current_stream();
	const Level &L = s->lv[0];
	hipEvent_t e0, e1;
	CEDAR_HIP_CHECK(hipEventCreate(&e0));
	CEDAR_HIP_CHECK(hipEventCreate(&e1));
	CEDAR_HIP_CHECK(hipEventRecord(e0, st));
	for (int i = 0; i < n; i++) {
		L.bt_fresh = false; // every timed sweep pays for its own transpose of b (a V(2,1) visit pays two per three sweeps)
		smooth(s, L, x_dev, b_dev, (i & 1) ? BMG_UP : BMG_DOWN, 1, st);
	}
	CEDAR_HIP_CHECK(hipEventRecord(e1, st));
	CEDAR_HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CEDAR_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	return ms;
}

// n launches of one level-0 transfer / residual kernel: op 1 = residual (res := b - A x), 2 = restriction of res to level 1,
// 3 = interpolation-and-add from level 1 (overwrites x and res: timing only)
float cedar_amd_solver_time_op(cedar_amd_solver *s, real_t *x_dev, const real_t *b_dev, int op, int n)
{
	if (null_handle(s, "cedar_amd_solver_time_op")) return 0.f;
	if (s->lv.size() < 2 || op < 1 || op > 3) return 0.f;
	hipStream_t st = current_stream();
	Level &L = s->lv[0], &K = s->lv[1];
	const int ibc = s->st.ibc;
	hipEvent_t e0, e1;
	CEDAR_HIP_CHECK(hipEventCreate(&e0));
	CEDAR_HIP_CHECK(hipEventCreate(&e1));
	CEDAR_HIP_CHECK(hipEventRecord(e0, st));
	for (int i = 0; i < n; i++) {
		if (op == 1) residual(s, L, x_dev, b_dev, L.res, st);
		else if (op == 2) {
			if (s->nd == 2 && ibc) restrict2_per(L.res, K.b, K.P, L.II, L.JJ, K.II, K.JJ, ibc, st);
			else if (s->nd == 2) restrict2(L.res, K.b, K.P, L.II, L.JJ, K.II, K.JJ, st, Batch{s->nb, L.npts}, Batch{s->nb, K.npts});
			else if (ibc) restrict3_per(L.res, K.b, K.P, L.II, L.JJ, L.KK, K.II, K.JJ, K.KK, ibc, st);
			else restrict3(L.res, K.b, K.P, L.II, L.JJ, L.KK, K.II, K.JJ, K.KK, st);
		} else {
			if (s->nd == 2 && ibc) interp_add2_per(x_dev, K.x, L.res, L.A, K.P, K.II, K.JJ, L.II, L.JJ, ibc, st);
			else if (s->nd == 2) interp_add2(x_dev, K.x, L.res, L.A, K.P, K.II, K.JJ, L.II, L.JJ, st, Batch{s->nb, L.npts}, Batch{s->nb, K.npts});
			else if (ibc) interp_add3_per(x_dev, K.x, L.A, L.res, K.P, K.II, K.JJ, K.KK, L.II, L.JJ, L.KK, ibc, st);
			else interp_add3(x_dev, K.x, L.A, L.res, K.P, K.II, K.JJ, K.KK, L.II, L.JJ, L.KK, st);
		}
	}
	CEDAR_HIP_CHECK(hipEventRecord(e1, st));
	CEDAR_HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CEDAR_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	launch_check("cedar_amd_solver_time_op");
	return ms;
}

} // extern "C"
