// C-ABI layer 4b: the domain-decomposed 2D solver -- one rank per GPU on a px x py rank grid -- below the C ABI.
//
// What this replaces in the reference: cdr2::mpi::solver and the 2D MPI flavour
// (include/cedar/2d/mpi/solver.h, src/2d/ftn/mpi/BMG2_SymStd_relax_GS.f90:102-171, ..._residual.f90, ..._interp_add.f90,
// ..._SETUP_interp_OI.f90, ..._SETUP_ITLI_ex.f90) with its distributed line relaxation
// (src/2d/ftn/mpi/BMG2_SymStd_relax_lines_x.f90:163-307 / _y.f90 over include/cedar/2d/mpi/ml_relax.h).
// Round 2 ran this orchestration on torch tensors over torch.distributed (cedar_amd/dist2d.py, _torch_dist.py); here it
// is compiled code on the library's own transport (dist_common.h: RCCL communicator or a caller-supplied table), so a 2D
// rank process needs no torch either.  Design rule as in 3D: serial equivalence by construction -- local extents stay
// even on every distributed level, ghost layers hold the owner's current values whenever a kernel reads them, the N-rank
// residual history equals the single-domain history on the same global problem.
//
// Point relaxation: the fused nine-point row pass relaxes both i-colours of a row class; with px > 1 the second colour's
// boundary column needs the x-neighbour's fresh first colour (one x-face exchange + column fix-up).  Five-point
// operators relax one red-black colour per exchange.  Line relaxation: dist_lines.hip.
#include "dist_common.h"
#include "dist_lines.h"
#include <cmath>

using namespace cedar_amd;
using namespace cedar_amd::dist;

namespace {

struct Lines2 { // the segments of the lines of one direction this rank owns (DistLines of dist2d.py)
	int dir = 0, nseg = 1, seg = 0, npos = 0, nl = 0;
	LineFactors F;
	real_t *work = nullptr;  // (lines of a colour, npos): right-hand sides -> y -> x
	real_t *pick = nullptr;  // (nseg, lines of a colour, 2): what the segments of a line hand each other
	real_t *carry = nullptr; // (lines of a colour)
	real_t *piv = nullptr;   // (nl) x 2: pivot entering / leaving the segment (set-up)
};

struct DLevel2 {
	int n[2] = {0, 0};
	int II = 0, JJ = 0, nst = 5;
	size_t npts = 0;
	real_t *A = nullptr, *P = nullptr, *x = nullptr, *b = nullptr, *res = nullptr, *sor = nullptr;
	bool ownA = true;
	Halo halo;
	Lines2 *lx = nullptr, *ly = nullptr;
};

} // namespace

struct cedar_amd_dist2 : RankCtx {
	int relax = CEDAR_AMD_RELAX_POINT;
	int pre = 2, post = 1, max_iter = 10, min_coarse = 3, agglomerate_below = 64;
	double tol = 1e-8;
	int nlev_global = 1, la = 0;
	std::vector<DLevel2> lv;
	int cn[2] = {0, 0};
	int gII = 0, gJJ = 0;
	real_t *gA = nullptr, *gx = nullptr, *gb = nullptr, *cs_tmp = nullptr;
	cedar_amd_solver *serial = nullptr;
	std::map<long, std::pair<real_t *, real_t *>> gbuf;
};

namespace {

void exch(cedar_amd_dist2 *d, DLevel2 &L, real_t *arr, int nplanes) { halo_exchange(d, L.halo, L.II, L.JJ, 1, arr, nplanes, 0); }

void gather_into(cedar_amd_dist2 *d, real_t *local, int lII, int lJJ, int nplanes, real_t *glob)
{
	const int nx = d->cn[0], ny = d->cn[1];
	const size_t blk = (size_t)nx * ny;
	auto it = d->gbuf.find(nplanes);
	if (it == d->gbuf.end())
		it = d->gbuf.emplace((long)nplanes, std::make_pair(dmalloc(blk * nplanes), dmalloc(blk * nplanes * d->world))).first;
	real_t *sb = it->second.first, *rb = it->second.second;
	const int own[6] = {1, 1, 0, nx, ny, 1};
	const unsigned long long zero = 0;
	cedar_amd_box_copy(local, lII, lJJ, 1, nplanes, 1, own, &zero, sb, 0);
	tp_allgather(d, sb, rb, blk * nplanes);
	for (int r0 = 0; r0 < d->world; r0 += 26) {
		const int nb = d->world - r0 < 26 ? d->world - r0 : 26;
		int boxes[26 * 6];
		unsigned long long offs[26];
		for (int i = 0; i < nb; i++) {
			const int r = r0 + i, ci = r % d->p[0], cj = r / d->p[0];
			const int b[6] = {1 + ci * nx, 1 + cj * ny, 0, nx, ny, 1};
			memcpy(boxes + 6 * i, b, sizeof(b));
			offs[i] = (unsigned long long)r * blk;
		}
		cedar_amd_box_copy(glob, d->gII, d->gJJ, 1, nplanes, nb, boxes, offs, rb, 1);
	}
}

// ---- distributed line relaxation
int line_rank(const cedar_amd_dist2 *d, int dir, int s) // rank of segment s of this rank's lines
{
	return dir == 0 ? rank_of(d, s, d->coord[1], 0) : rank_of(d, d->coord[0], s, 0);
}

Lines2 *lines_setup(cedar_amd_dist2 *d, DLevel2 &L, int dir)
{
	Lines2 *S = new Lines2;
	S->dir = dir;
	S->nseg = d->p[dir];
	S->seg = d->coord[dir];
	S->npos = L.n[dir];
	S->nl = L.n[1 - dir];
	const size_t tot = (size_t)S->nl * S->npos;
	S->F.nl = S->nl; S->F.npos = S->npos;
	S->F.dp = dmalloc(tot); S->F.af = dmalloc(tot); S->F.ab = dmalloc(tot); S->F.pf = dmalloc(tot); S->F.pb = dmalloc(tot);
	const int nc = (S->nl + 1) / 2;
	S->work = dmalloc((size_t)nc * S->npos);
	S->pick = dmalloc((size_t)S->nseg * nc * 2);
	S->carry = dmalloc(nc);
	S->piv = dmalloc(2 * (size_t)S->nl);
	const bool has_prev = S->seg > 0, has_next = S->seg < S->nseg - 1;
	// the rank owning segment r waits for the last pivot of segment r-1: a pipeline of nseg steps, once per level
	if (has_prev) {
		const int peer = line_rank(d, dir, S->seg - 1);
		real_t *rp = S->piv;
		const size_t cnt = (size_t)S->nl;
		if (tp_exchange(d, 0, nullptr, nullptr, nullptr, 1, &peer, &rp, &cnt)) { char m[] = "cedar_amd_dist2: pivot receive failed"; print_error(m); }
	}
	dist_lines_factor(L.A, L.II, L.JJ, dir, S->npos, S->nl, S->piv, has_prev, has_next, S->F, S->piv + S->nl, current_stream());
	if (has_next) {
		const int peer = line_rank(d, dir, S->seg + 1);
		const real_t *sp = S->piv + S->nl;
		const size_t cnt = (size_t)S->nl;
		if (tp_exchange(d, 1, &peer, &sp, &cnt, 0, nullptr, nullptr, nullptr)) { char m[] = "cedar_amd_dist2: pivot send failed"; print_error(m); }
	}
	return S;
}

void lines_free(Lines2 *S)
{
	if (!S) return;
	cedar_amd_free(S->F.dp); cedar_amd_free(S->F.af); cedar_amd_free(S->F.ab); cedar_amd_free(S->F.pf); cedar_amd_free(S->F.pb);
	cedar_amd_free(S->work); cedar_amd_free(S->pick); cedar_amd_free(S->carry); cedar_amd_free(S->piv);
	delete S;
}

// the ranks of a line hand each other (value leaving the segment, product of its multipliers): every segment's pair ends
// up in pick[(segment, line, 2)] on every rank of the line
void lines_share(cedar_amd_dist2 *d, Lines2 &S, int nlc)
{
	if (S.nseg == 1) return;
	int peer[8];
	const real_t *sp[8];
	real_t *rp[8];
	size_t cnt[8];
	int m = 0;
	for (int r = 0; r < S.nseg && m < 8; r++) {
		if (r == S.seg) continue;
		peer[m] = line_rank(d, S.dir, r);
		sp[m] = S.pick + (size_t)S.seg * nlc * 2;
		rp[m] = S.pick + (size_t)r * nlc * 2;
		cnt[m] = (size_t)nlc * 2;
		m++;
	}
	if (tp_exchange(d, m, peer, sp, cnt, m, peer, rp, cnt)) { char msg[] = "cedar_amd_dist2: line carry exchange failed"; print_error(msg); }
}

// one zebra sweep: DOWN relaxes lines 3,5,.. then 2,4,.. (1-based), UP the reverse (relax_lines_x.f90:82-97); halo
// exchange after each colour
void lines_relax(cedar_amd_dist2 *d, DLevel2 &L, Lines2 &S, real_t *x, real_t *b, int updown)
{
	hipStream_t st = current_stream();
	const int n = S.npos;
	for (int c = 0; c < 2; c++) {
		const int lb = updown == BMG_DOWN ? 1 - c : c; // 0-based interior line parity
		const int nlc = S.F.colour_lines(lb);
		if (nlc > 0) {
			const size_t o = S.F.colour_offset(lb);
			cedar_amd_lines_rhs2(L.A, b, x, S.work, L.II, L.JJ, L.nst, S.dir, lb);
			// forward sweep from a zero carry, then the carry entering this segment
			cedar_amd_affine_lines(S.work, S.F.af + o, nullptr, nlc, n, n, 0);
			if (S.nseg > 1) {
				dist_lines_pick(S.work, S.F.pf + o, nlc, n, n - 1, S.pick + (size_t)S.seg * nlc * 2, st);
				lines_share(d, S, nlc);
				if (S.seg > 0) {
					dist_lines_compose(S.pick, S.nseg, S.seg, nlc, 0, S.carry, st);
					cedar_amd_lines_carry(S.work, S.F.pf + o, S.carry, nlc, n, n);
				}
			}
			// backward sweep from a zero carry, then the carry entering from the right
			cedar_amd_affine_lines(S.work, S.F.ab + o, S.F.dp + o, nlc, n, n, 1);
			if (S.nseg > 1) {
				dist_lines_pick(S.work, S.F.pb + o, nlc, n, 0, S.pick + (size_t)S.seg * nlc * 2, st);
				lines_share(d, S, nlc);
				if (S.seg < S.nseg - 1) {
					dist_lines_compose(S.pick, S.nseg, S.seg, nlc, 1, S.carry, st);
					cedar_amd_lines_carry(S.work, S.F.pb + o, S.carry, nlc, n, n);
				}
			}
			cedar_amd_lines_store2(S.work, x, L.II, L.JJ, S.dir, lb);
		}
		exch(d, L, x, 1);
	}
}

// ---- cycle pieces
void smooth(cedar_amd_dist2 *d, DLevel2 &L, real_t *x, real_t *b, int updown, int nsweeps)
{
	const bool down = updown == BMG_DOWN;
	for (int it = 0; it < nsweeps; it++) {
		if (d->relax != CEDAR_AMD_RELAX_POINT) {
			// multilevel.h:165-222: pre = DOWN sweeps (line-xy: x then y), post = UP (y then x)
			if (d->relax == CEDAR_AMD_RELAX_LINE_X) lines_relax(d, L, *L.lx, x, b, updown);
			else if (d->relax == CEDAR_AMD_RELAX_LINE_Y) lines_relax(d, L, *L.ly, x, b, updown);
			else if (down) { lines_relax(d, L, *L.lx, x, b, updown); lines_relax(d, L, *L.ly, x, b, updown); }
			else { lines_relax(d, L, *L.ly, x, b, updown); lines_relax(d, L, *L.lx, x, b, updown); }
			continue;
		}
		if (L.nst == 3) {
			for (int c = 0; c < 2; c++) {
				cedar_amd_relax2_colour5(L.A, b, x, L.sor, L.II, L.JJ, down ? 2 + c : 3 - c);
				exch(d, L, x, 1);
			}
			continue;
		}
		for (int c = 0; c < 2; c++) {
			const int jb = down ? c : 1 - c; // DOWN: rows J = 2,4,.. first, even 1-based i first
			cedar_amd_relax2_pass(L.A, b, x, L.sor, L.II, L.JJ, jb, down);
			if (d->p[0] > 1 && halo_exchange_x(d, L.halo, L.II, L.JJ, 1, x, down))
				cedar_amd_relax2_fixup(L.A, b, x, L.sor, L.II, L.JJ, down ? L.n[0] : 1, jb);
			exch(d, L, x, 1);
		}
	}
}

void residual(DLevel2 &L, real_t *x, real_t *b)
{
	int k = 0, kf = 0, ifd = L.nst == 3, nst = L.nst, zero = 0;
	len_t II = (len_t)L.II, JJ = (len_t)L.JJ;
	BMG2_SymStd_residual(&k, L.A, b, x, L.res, &II, &JJ, &kf, &ifd, &nst, &zero, &zero, &zero, &zero);
}

void coarse_solve(cedar_amd_dist2 *d, DLevel2 &C, real_t *x, real_t *b)
{
	gather_into(d, b, C.II, C.JJ, 1, d->gb);
	cedar_amd_memset(d->gx, 0, (size_t)d->gII * d->gJJ * sizeof(real_t));
	cedar_amd_solver_vcycle(d->serial, d->gx, d->gb);
	const int nx = d->cn[0], ny = d->cn[1];
	const unsigned long long zero = 0;
	const int from[6] = {d->coord[0] * nx, d->coord[1] * ny, 0, nx + 2, ny + 2, 1};
	const int to[6] = {0, 0, 0, nx + 2, ny + 2, 1};
	cedar_amd_box_copy(d->gx, d->gII, d->gJJ, 1, 1, 1, from, &zero, d->cs_tmp, 0);
	cedar_amd_box_copy(x, C.II, C.JJ, 1, 1, 1, to, &zero, d->cs_tmp, 1);
}

void cycle(cedar_amd_dist2 *d, int l, real_t *x, real_t *b)
{
	DLevel2 &L = d->lv[l], &K = d->lv[l + 1];
	smooth(d, L, x, b, BMG_DOWN, d->pre);
	residual(L, x, b);
	exch(d, L, L.res, 1);
	BMG2_SymStd_restrict(L.res, K.b, K.P, L.II, L.JJ, K.II, K.JJ, 0);
	cedar_amd_memset(K.x, 0, K.npts * sizeof(real_t));
	if (l + 1 == (int)d->lv.size() - 1) coarse_solve(d, K, K.x, K.b);
	else cycle(d, l + 1, K.x, K.b);
	BMG2_SymStd_interp_add(x, K.x, L.res, L.A, K.P, K.II, K.JJ, L.II, L.JJ, L.nst, 0);
	exch(d, L, x, 1);
	smooth(d, L, x, b, BMG_UP, d->post);
}

void vcycle(cedar_amd_dist2 *d, real_t *x, real_t *b)
{
	if (d->lv.size() == 1) coarse_solve(d, d->lv[0], x, b);
	else cycle(d, 0, x, b);
}

double norm(cedar_amd_dist2 *d, DLevel2 &L, const real_t *r)
{
	const double v = cedar_amd_l2norm(r, L.II, L.JJ, 1);
	return std::sqrt(tp_allreduce_sum(d, v * v));
}

void setup(cedar_amd_dist2 *d)
{
	const int lo[2] = {has_nb(d, 0, -1) ? 2 : 3, has_nb(d, 1, -1) ? 2 : 3};
	DLevel2 &L0 = d->lv[0];
	exch(d, L0, L0.A, L0.nst);
	for (size_t l = 0; l + 1 < d->lv.size(); l++) {
		DLevel2 &F = d->lv[l], &K = d->lv[l + 1];
		for (int phase = 0; phase < 2; phase++) {
			cedar_amd_setup_interp2_phase(F.A, K.P, F.II, F.JJ, K.II, K.JJ, F.nst == 3, F.nst, phase, lo[0], lo[1]);
			exch(d, K, K.P, 8);
		}
		BMG2_SymStd_SETUP_ITLI_ex(F.A, K.A, K.P, F.II, F.JJ, K.II, K.JJ, F.nst == 3, F.nst, 0);
		exch(d, K, K.A, 5);
		if (d->relax == CEDAR_AMD_RELAX_POINT) BMG2_SymStd_SETUP_recip(F.A, F.sor, F.II, F.JJ, F.nst, 2);
		if (d->relax == CEDAR_AMD_RELAX_LINE_X || d->relax == CEDAR_AMD_RELAX_LINE_XY) F.lx = lines_setup(d, F, 0);
		if (d->relax == CEDAR_AMD_RELAX_LINE_Y || d->relax == CEDAR_AMD_RELAX_LINE_XY) F.ly = lines_setup(d, F, 1);
	}
	DLevel2 &C = d->lv.back();
	d->cn[0] = C.n[0]; d->cn[1] = C.n[1];
	d->gII = C.n[0] * d->p[0] + 2; d->gJJ = C.n[1] * d->p[1] + 2;
	const size_t gp = (size_t)d->gII * d->gJJ;
	d->gA = dmalloc(gp * C.nst);
	gather_into(d, C.A, C.II, C.JJ, C.nst, d->gA);
	d->gx = dmalloc(gp);
	d->gb = dmalloc(gp);
	d->cs_tmp = dmalloc((size_t)(C.n[0] + 2) * (C.n[1] + 2));
	cedar_amd_settings st;
	cedar_amd_default_settings(&st);
	st.relaxation = d->relax;
	st.nrelax_pre = d->pre; st.nrelax_post = d->post; st.min_coarse = d->min_coarse;
	st.num_levels = d->nlev_global - d->la;
	d->serial = cedar_amd_solver_create(2, (len_t)(d->gII - 2), (len_t)(d->gJJ - 2), 1, C.nst, d->gA, 1, &st);
}

} // namespace

extern "C" {

void cedar_amd_dist2_rank_grid(int world, int pgrid[2])
{
	// 1 -> 1x1, 2 -> 1x2, 4 -> 2x2, 8 -> 2x4: y is split first (y faces are contiguous rows and the fused row pass needs
	// no x fix-up while px = 1)
	int best = 1 << 30;
	pgrid[0] = 1; pgrid[1] = world < 1 ? 1 : world;
	for (int py = 1; py <= world; py++) {
		if (world % py) continue;
		const int px = world / py;
		if (px > py) continue;
		if (py - px < best) { best = py - px; pgrid[0] = px; pgrid[1] = py; }
	}
}

cedar_amd_dist2 *cedar_amd_dist2_create(cedar_amd_comm *comm, const cedar_amd_transport *transport, int rank, int world,
                                        const int pgrid[2], real_t *A_local, len_t nx, len_t ny, int nstencil,
                                        const cedar_amd_settings *settings, int agglomerate_below)
{
	if (!A_local || !is_device_ptr(A_local) || (nstencil != 5 && nstencil != 3) || world < 1 || rank < 0 || rank >= world) {
		char m[] = "cedar_amd_dist2_create: A_local must be a device array of a 5- or 9-point operator, 0 <= rank < world";
		print_error(m);
		return nullptr;
	}
	if (world > 1 && !comm && !(transport && transport->exchange && transport->allgather && transport->allreduce_sum)) {
		char m[] = "cedar_amd_dist2_create: more than one rank needs a communicator (cedar_amd_comm_create) or a transport table";
		print_error(m);
		return nullptr;
	}
	cedar_amd_dist2 *d = new cedar_amd_dist2;
	d->comm = comm;
	if (transport && transport->exchange) { d->tp = *transport; d->has_tp = true; }
	d->rank = rank; d->world = world;
	int pg[2];
	if (pgrid) { pg[0] = pgrid[0]; pg[1] = pgrid[1]; }
	else cedar_amd_dist2_rank_grid(world, pg);
	d->p[0] = pg[0]; d->p[1] = pg[1]; d->p[2] = 1;
	if (d->p[0] * d->p[1] != world || d->p[0] > 8 || d->p[1] > 8) {
		char m[] = "cedar_amd_dist2_create: the rank grid must multiply to the world size (at most 8 ranks per direction)";
		print_error(m);
		delete d;
		return nullptr;
	}
	d->coord[0] = rank % d->p[0]; d->coord[1] = rank / d->p[0]; d->coord[2] = 0;
	cedar_amd_settings st;
	if (settings) st = *settings;
	else cedar_amd_default_settings(&st);
	if (st.relaxation < CEDAR_AMD_RELAX_POINT || st.relaxation > CEDAR_AMD_RELAX_LINE_XY) {
		char m[] = "cedar_amd_dist2_create: relaxation must be point / line-x / line-y / line-xy";
		print_error(m);
		delete d;
		return nullptr;
	}
	d->relax = st.relaxation;
	d->pre = st.nrelax_pre; d->post = st.nrelax_post; d->max_iter = st.max_iter; d->tol = st.tol; d->min_coarse = st.min_coarse;
	d->agglomerate_below = agglomerate_below > 0 ? agglomerate_below : 64;
	d->scal = dmalloc(8);
	int n[2] = {(int)nx, (int)ny};
	int ng = 0;
	for (;;) { // include/cedar/2d/solver.h:57-73 on the GLOBAL extents
		ng++;
		int m = 1 << 30;
		for (int t = 0; t < 2; t++) {
			const int g = n[t] * d->p[t], c = (g - 1) / (1 << ng) + 1;
			if (c < m) m = c;
		}
		if (m < d->min_coarse) break;
	}
	d->nlev_global = ng;
	int la = ng - 1, m[2] = {n[0], n[1]};
	for (int l = 1; l < ng; l++) {
		int mn = 1 << 30;
		for (int t = 0; t < 2; t++) {
			m[t] = d->p[t] == 1 ? (int)((m[t] - 1) / 2.0 + 1) : m[t] / 2;
			if (m[t] < mn) mn = m[t];
		}
		if (mn <= d->agglomerate_below) { la = l; break; }
	}
	d->la = ng > 1 ? (la > 1 ? la : 1) : 0;
	for (int l = 0; l <= d->la; l++) {
		for (int t = 0; t < 2; t++)
			if (d->p[t] > 1 && l < d->la && (n[t] & 1)) {
				char msg[160];
				snprintf(msg, sizeof(msg), "cedar_amd_dist2_create: level %d: local extent %d along a split direction must be even", l, n[t]);
				print_error(msg);
				cedar_amd_dist2_destroy(d);
				return nullptr;
			}
		d->lv.emplace_back();
		DLevel2 &R = d->lv.back();
		R.n[0] = n[0]; R.n[1] = n[1];
		R.II = n[0] + 2; R.JJ = n[1] + 2;
		R.npts = (size_t)R.II * R.JJ;
		const int n3[3] = {n[0], n[1], -1};
		halo_init(d, R.halo, n3);
		R.res = dmalloc(R.npts);
		R.sor = dmalloc(2 * R.npts);
		if (l == 0) {
			R.A = A_local; R.ownA = false; R.nst = nstencil;
		} else {
			R.nst = 5;
			R.A = dmalloc(5 * R.npts);
			R.P = dmalloc(8 * R.npts);
			R.x = dmalloc(R.npts);
			R.b = dmalloc(R.npts);
		}
		for (int t = 0; t < 2; t++) n[t] = d->p[t] == 1 ? (int)((n[t] - 1) / 2.0 + 1) : n[t] / 2;
	}
	setup(d);
	if (!d->serial) {
		cedar_amd_dist2_destroy(d);
		return nullptr;
	}
	launch_check("cedar_amd_dist2_create");
	return d;
}

void cedar_amd_dist2_destroy(cedar_amd_dist2 *d)
{
	if (!d) return;
	cedar_amd_device_sync();
	if (d->serial) cedar_amd_solver_destroy(d->serial);
	for (DLevel2 &L : d->lv) {
		if (L.ownA) cedar_amd_free(L.A);
		cedar_amd_free(L.P); cedar_amd_free(L.x); cedar_amd_free(L.b); cedar_amd_free(L.res); cedar_amd_free(L.sor);
		for (auto &kv : L.halo.bufs) { cedar_amd_free(kv.second.first); cedar_amd_free(kv.second.second); }
		lines_free(L.lx);
		lines_free(L.ly);
	}
	for (auto &kv : d->gbuf) { cedar_amd_free(kv.second.first); cedar_amd_free(kv.second.second); }
	cedar_amd_free(d->gA); cedar_amd_free(d->gx); cedar_amd_free(d->gb); cedar_amd_free(d->cs_tmp); cedar_amd_free(d->scal);
	if (d->side) cedar_amd_stream_destroy(d->side);
	delete d;
}

int cedar_amd_dist2_nlevels(const cedar_amd_dist2 *d) { return d ? d->nlev_global : 0; }

void cedar_amd_dist2_vcycle(cedar_amd_dist2 *d, real_t *x, real_t *b)
{
	if (!d) return;
	vcycle(d, x, b);
	launch_check("cedar_amd_dist2_vcycle");
}

int cedar_amd_dist2_solve(cedar_amd_dist2 *d, real_t *b, real_t *x, real_t *rel)
{
	if (!d) return 0;
	DLevel2 &L = d->lv[0];
	exch(d, L, x, 1);
	residual(L, x, b);
	const double r0 = norm(d, L, L.res);
	rel[0] = r0;
	int it = 0;
	while (it < d->max_iter) {
		vcycle(d, x, b);
		residual(L, x, b);
		const double r = norm(d, L, L.res) / r0;
		rel[++it] = r;
		if (r < d->tol) break;
	}
	launch_check("cedar_amd_dist2_solve");
	return it;
}

float cedar_amd_dist2_time_relax(cedar_amd_dist2 *d, real_t *x, real_t *b, int n)
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
