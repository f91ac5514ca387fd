// C-ABI layer 3: Cedar's C interface (bmg2_* / bmg3_* / bmg_timer_save), declared in
// include/cedar/capi.h.  Behaviour follows the reference's src/{2d,3d}/interface/c/{topo,operator,
// solver}.cc and src/interface/c/timer.cc; the solver underneath is the device-resident handle API
// of solver.cpp (single rank: one process drives one GPU).
//
// The operator is assembled entry by entry on the host exactly like the reference does (the caller
// sets individual stencil entries), uploaded once by bmgN_solver_create / on first apply.
#include "../../include/cedar_amd.h"
#include "../../include/cedar/capi.h"
#include "../../include/cedar/config.h"
#include "dist_common.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <string>
#include <vector>

namespace {

using clk = std::chrono::steady_clock;
std::map<std::string, double> g_timers; // seconds, accumulated per phase name
std::map<std::string, int> g_counts;

struct scoped_timer {
	std::string name;
	clk::time_point t0;
	explicit scoped_timer(const char *n) : name(n), t0(clk::now()) {}
	~scoped_timer()
	{
		g_timers[name] += std::chrono::duration<double>(clk::now() - t0).count();
		g_counts[name]++;
	}
};

void report(const std::string &msg)
{
	std::vector<char> buf(msg.begin(), msg.end());
	buf.push_back(0);
	print_error(buf.data());
}

// ---- who this process is among the ranks, and how it talks to them
int g_rank = -1, g_world = -1;
cedar_amd_comm *g_comm = nullptr;
cedar_amd_transport g_tp{};
bool g_has_tp = false;

int env_int(const char *const names[], int n, int dflt)
{
	for (int i = 0; i < n; i++)
		if (const char *e = getenv(names[i])) return atoi(e);
	return dflt;
}

void rank_world(int &rank, int &world)
{
	static const char *const rk[] = {"RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID"};
	static const char *const wd[] = {"WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE", "SLURM_NTASKS"};
	rank = g_rank >= 0 ? g_rank : env_int(rk, 4, 0);
	world = g_world >= 1 ? g_world : env_int(wd, 4, 1);
}

// transport of a multi-rank topology: the table handed in, else the RCCL communicator (made once per process)
bool transport(int rank, int world, cedar_amd_comm *&comm, const cedar_amd_transport *&tp)
{
	comm = nullptr; tp = nullptr;
	if (world == 1) return true;
	if (g_has_tp) { tp = &g_tp; return true; }
	if (!g_comm) g_comm = cedar_amd_comm_bootstrap(rank, world);
	comm = g_comm;
	return comm != nullptr;
}

// grid_topo of the reference (src/2d/interface/c/topo.cc:10-66): process grid, this rank's place and extents
struct topo_t {
	int nd;
	unsigned ng[3]; // global interior extents
	unsigned nl[3]; // local interior extents
	unsigned is[3]; // 1-based global index of the first local point (src/2d/interface/c/topo.cc:38-45)
	int nproc[3], coord[3];
	int rank, world;
	bool uniform; // every rank owns the same extents (what the domain-decomposed drivers take)
};

struct op_t {
	topo_t topo;
	int nst;                // 5 (nine_pt) or 14 (xxvii_pt): the interface always uses the full stencil
	std::vector<real_t> so; // Cedar layout: (nst, [KK,] JJ, II), i fastest
	real_t *dso = nullptr, *dx = nullptr, *db = nullptr;
	bool dirty = true;
	cedar_amd::dist::Halo *halo = nullptr; // multi-rank: ghost exchange of x for operator_apply
	size_t pts() const { return (size_t)(topo.nl[0] + 2) * (topo.nl[1] + 2) * (topo.nd == 3 ? topo.nl[2] + 2 : 1); }
	~op_t()
	{
		cedar_amd_free(dso);
		cedar_amd_free(dx);
		cedar_amd_free(db);
		if (halo) {
			for (auto &kv : halo->bufs) { cedar_amd_free(kv.second.first); cedar_amd_free(kv.second.second); }
			delete halo;
		}
	}
	void upload()
	{
		if (!dso) {
			dso = static_cast<real_t *>(cedar_amd_malloc(so.size() * sizeof(real_t)));
			dx = static_cast<real_t *>(cedar_amd_malloc(pts() * sizeof(real_t)));
			db = static_cast<real_t *>(cedar_amd_malloc(pts() * sizeof(real_t)));
		}
		if (dirty) cedar_amd_memcpy_h2d(dso, so.data(), so.size() * sizeof(real_t));
		dirty = false;
	}
};

struct slv_t {
	op_t *op;
	cedar_amd_solver *h = nullptr;
	cedar_amd_dist2 *d2 = nullptr;
	cedar_amd_dist3 *d3 = nullptr;
	cedar_amd_settings st;
	std::vector<real_t> xg, bg, rel;
};

// rank context of a topology for the shared halo machinery (dist_common.h)
bool rank_ctx(const topo_t &t, cedar_amd::dist::RankCtx &c)
{
	cedar_amd_comm *comm;
	const cedar_amd_transport *tp;
	if (!transport(t.rank, t.world, comm, tp)) return false;
	c.comm = comm;
	if (tp) { c.tp = *tp; c.has_tp = true; }
	c.rank = t.rank; c.world = t.world;
	for (int d = 0; d < 3; d++) { c.p[d] = t.nproc[d]; c.coord[d] = t.coord[d]; }
	return true;
}

topo_t *make_topo(int nd, const unsigned ng[3], unsigned *const ln[3], const int np[3])
{
	int rank, world;
	rank_world(rank, world);
	int nproc = 1;
	for (int d = 0; d < nd; d++) nproc *= np[d] < 1 ? 1 : np[d];
	if (nproc != world || rank >= world) {
		char buf[240];
		snprintf(buf, sizeof(buf), "bmg_topo_create: the process grid has %d ranks but the launcher started %d (this is rank %d: RANK / "
		         "WORLD_SIZE, PMI_*, OMPI_COMM_WORLD_* or cedar_amd_bmg_set_rank); one rank per GPU", nproc, world, rank);
		report(buf);
		return nullptr;
	}
	auto *t = new topo_t();
	t->nd = nd;
	t->rank = rank; t->world = world;
	// rank = (k * nprocy + j) * nprocx + i (src/2d/interface/c/topo.cc:35-36, src/3d/interface/c/topo.cc)
	int r = rank;
	t->uniform = true;
	for (int d = 0; d < 3; d++) {
		const int npd = d < nd ? np[d] : 1;
		t->nproc[d] = npd;
		t->coord[d] = r % npd;
		r /= npd;
		t->ng[d] = d < nd ? ng[d] : 1;
		t->nl[d] = d < nd ? ln[d][t->coord[d]] : 1;
		t->is[d] = 1;
		for (int i = 0; d < nd && i < t->coord[d]; i++) t->is[d] += ln[d][i];
		for (int i = 0; d < nd && i < npd; i++) t->uniform = t->uniform && ln[d][i] == ln[d][0];
	}
	return t;
}

op_t *make_op(topo_t *t)
{
	if (!t) return nullptr;
	auto *o = new op_t();
	o->topo = *t;
	o->nst = t->nd == 2 ? 5 : 14;
	o->so.assign(o->pts() * o->nst, 0.0);
	return o;
}

// gather/scatter between the caller's interior-only arrays and ghosted grid functions
void interior_copy(const topo_t &t, std::vector<real_t> &g, const double *in, double *out)
{
	const size_t II = t.nl[0] + 2, JJ = t.nl[1] + 2;
	const unsigned nk = t.nd == 3 ? t.nl[2] : 1;
	size_t idx = 0;
	for (unsigned k = 0; k < nk; k++)
		for (unsigned j = 0; j < t.nl[1]; j++) {
			real_t *row = g.data() + (t.nd == 3 ? (k + 1) * II * JJ : 0) + (j + 1) * II + 1;
			if (in) memcpy(row, in + idx, t.nl[0] * sizeof(real_t));
			else memcpy(out + idx, row, t.nl[0] * sizeof(real_t));
			idx += t.nl[0];
		}
}

void apply(op_t *o, const double *x, double *b)
{
	scoped_timer tm("matvec");
	const topo_t &t = o->topo;
	std::vector<real_t> g(o->pts(), 0.0);
	interior_copy(t, g, x, nullptr);
	o->upload();
	cedar_amd_memcpy_h2d(o->dx, g.data(), g.size() * sizeof(real_t));
	if (t.world > 1) { // the ghost layer of x from the neighbouring ranks (the reference's mpi::stencil_op::apply exchanges it too)
		cedar_amd::dist::RankCtx c;
		if (!rank_ctx(t, c)) return;
		if (!o->halo) {
			o->halo = new cedar_amd::dist::Halo;
			const int n3[3] = {(int)t.nl[0], (int)t.nl[1], t.nd == 3 ? (int)t.nl[2] : -1};
			cedar_amd::dist::halo_init(&c, *o->halo, n3);
		}
		cedar_amd::dist::halo_exchange(&c, *o->halo, (int)t.nl[0] + 2, (int)t.nl[1] + 2, t.nd == 3 ? (int)t.nl[2] + 2 : 1, o->dx, 1, 0);
	}
	cedar_amd_memset(o->db, 0, g.size() * sizeof(real_t));
	if (t.nd == 2) cedar_amd_matvec2(o->dso, o->dx, o->db, t.nl[0] + 2, t.nl[1] + 2, o->nst);
	else cedar_amd_matvec3(o->dso, o->dx, o->db, t.nl[0] + 2, t.nl[1] + 2, t.nl[2] + 2, o->nst);
	cedar_amd_memcpy_d2h(g.data(), o->db, g.size() * sizeof(real_t));
	interior_copy(t, g, nullptr, b);
}

slv_t *make_solver(op_t *o)
{
	if (!o) return nullptr;
	scoped_timer tm("setup");
	auto *s = new slv_t();
	s->op = o;
	cedar::config conf("config.json"); // src/2d/interface/c/operator.cc:20, include/cedar/2d/mpi/solver.h ctor
	cedar::ml_settings ms;
	ms.init(conf);
	cedar_amd_default_settings(&s->st);
	s->st.relaxation = static_cast<int>(ms.relaxation);
	s->st.nrelax_pre = ms.nrelax_pre;
	s->st.nrelax_post = ms.nrelax_post;
	s->st.num_levels = ms.num_levels;
	s->st.max_iter = ms.maxiter;
	s->st.tol = ms.tol;
	s->st.min_coarse = ms.min_coarse;
	s->st.cycle = ms.cycle;
	o->upload();
	const topo_t &t = o->topo;
	if (t.world > 1) {
		// mpi::solver on the rank grid (src/2d/interface/c/solver.cc:10-22): the domain-decomposed drivers, one rank per GPU
		cedar_amd_comm *comm;
		const cedar_amd_transport *tp;
		if (!t.uniform) {
			report("bmg_solver_create: the domain-decomposed solver takes the same local extents on every rank");
			delete s;
			return nullptr;
		}
		if (!transport(t.rank, t.world, comm, tp)) {
			delete s;
			return nullptr;
		}
		if (t.nd == 2) s->d2 = cedar_amd_dist2_create(comm, tp, t.rank, t.world, t.nproc, o->dso, t.nl[0], t.nl[1], o->nst, &s->st, 0);
		else s->d3 = cedar_amd_dist3_create(comm, tp, t.rank, t.world, t.nproc, o->dso, t.nl[0], t.nl[1], t.nl[2], o->nst, &s->st, 0, 0);
		if (!s->d2 && !s->d3) {
			delete s;
			return nullptr;
		}
	} else {
		s->h = cedar_amd_solver_create(t.nd, t.nl[0], t.nl[1], t.nd == 3 ? t.nl[2] : 1, o->nst, o->dso, 1, &s->st);
		if (!s->h) {
			delete s;
			return nullptr;
		}
	}
	s->xg.assign(o->pts(), 0.0);
	s->bg.assign(o->pts(), 0.0);
	s->rel.assign(s->st.max_iter + 2, 0.0);
	return s;
}

void run(slv_t *s, double *x, const double *b)
{
	if (!s) return;
	scoped_timer tm("solve");
	const topo_t &t = s->op->topo;
	interior_copy(t, s->bg, b, nullptr);
	std::fill(s->xg.begin(), s->xg.end(), 0.0); // sol.set(0.0), src/2d/interface/c/solver.cc:43
	if (s->h) cedar_amd_solver_solve(s->h, s->bg.data(), s->xg.data(), s->rel.data());
	else {
		op_t *o = s->op;
		const size_t bytes = s->bg.size() * sizeof(real_t);
		cedar_amd_memcpy_h2d(o->db, s->bg.data(), bytes);
		cedar_amd_memset(o->dx, 0, bytes);
		if (s->d2) cedar_amd_dist2_solve(s->d2, o->db, o->dx, s->rel.data());
		else cedar_amd_dist3_solve(s->d3, o->db, o->dx, s->rel.data());
		cedar_amd_memcpy_d2h(s->xg.data(), o->dx, bytes);
	}
	interior_copy(t, s->xg, nullptr, x);
}

void dump(op_t *o)
{
	// debug listing, one row of the vertex-based stencil (natural signs) per line; the reference
	// prints the same quantities through operator<< (src/2d/mpi/stencil_op.cc:32-59)
	const topo_t &t = o->topo;
	std::string name = "op" + std::to_string(t.coord[0]) + "-" + std::to_string(t.coord[1]);
	if (t.nd == 3) name += "-" + std::to_string(t.coord[2]);
	std::ofstream f(name + ".txt", std::ios::out | std::ios::trunc);
	f << std::setprecision(7) << std::scientific;
	const long II = t.nl[0] + 2, JJ = t.nl[1] + 2, KK = t.nd == 3 ? t.nl[2] + 2 : 1;
	const size_t P = (size_t)II * JJ * KK;
	auto S = [&](int s, long i, long j, long k) { return o->so[s * P + (size_t)i + II * ((size_t)j + JJ * (size_t)k)]; };
	if (t.nd == 2) {
		for (long j = 1; j <= (long)t.nl[1]; j++)
			for (long i = 1; i <= (long)t.nl[0]; i++) {
				f << std::setw(4) << (j - 1) * (long)t.ng[0] + (i - 1) << " " << std::setw(4) << i << ", " << std::setw(4) << j << ", "
				  << -S(4, i, j + 1, 0) << " " << -S(2, i, j + 1, 0) << " " << -S(3, i + 1, j + 1, 0) << " "
				  << -S(1, i, j, 0) << " " << S(0, i, j, 0) << " " << -S(1, i + 1, j, 0) << " "
				  << -S(3, i, j, 0) << " " << -S(2, i, j, 0) << " " << -S(4, i + 1, j, 0) << '\n';
			}
	} else {
		for (long k = 1; k <= (long)t.nl[2]; k++)
			for (long j = 1; j <= (long)t.nl[1]; j++)
				for (long i = 1; i <= (long)t.nl[0]; i++) {
					f << std::setw(4) << ((k - 1) * (long)t.ng[1] + (j - 1)) * (long)t.ng[0] + (i - 1) << " " << std::setw(4) << i << ", "
					  << std::setw(4) << j << ", " << std::setw(4) << k << ",";
					for (int s = 0; s < 14; s++) f << " " << (s ? -S(s, i, j, k) : S(s, i, j, k));
					f << '\n';
				}
	}
}

} // namespace

extern "C" {

// who this process is (overrides the launcher's environment) and, optionally, how it talks to the other ranks (a transport
// table in place of the RCCL communicator the interface would bootstrap itself) -- include/cedar_amd.h section 4
void cedar_amd_bmg_set_rank(int rank, int world) { g_rank = rank; g_world = world; }
void cedar_amd_bmg_set_transport(const cedar_amd_transport *tp)
{
	g_has_tp = tp && tp->exchange;
	if (g_has_tp) g_tp = *tp;
}

bmg2_topo bmg2_topo_create(MPI_Comm, unsigned int ngx, unsigned int ngy, unsigned int lnx[], unsigned int lny[],
                           int nprocx, int nprocy)
{
	const unsigned ng[3] = { ngx, ngy, 1 };
	unsigned *const ln[3] = { lnx, lny, nullptr };
	const int np[3] = { nprocx, nprocy, 1 };
	return reinterpret_cast<bmg2_topo>(make_topo(2, ng, ln, np));
}

bmg3_topo bmg3_topo_create(MPI_Comm, unsigned int ngx, unsigned int ngy, unsigned int ngz, unsigned int lnx[],
                           unsigned int lny[], unsigned int lnz[], int nprocx, int nprocy, int nprocz)
{
	const unsigned ng[3] = { ngx, ngy, ngz };
	unsigned *const ln[3] = { lnx, lny, lnz };
	const int np[3] = { nprocx, nprocy, nprocz };
	return reinterpret_cast<bmg3_topo>(make_topo(3, ng, ln, np));
}

bmg2_operator bmg2_operator_create(bmg2_topo topo) { return reinterpret_cast<bmg2_operator>(make_op(reinterpret_cast<topo_t *>(topo))); }
bmg3_operator bmg3_operator_create(bmg3_topo topo) { return reinterpret_cast<bmg3_operator>(make_op(reinterpret_cast<topo_t *>(topo))); }

void bmg2_operator_set(bmg2_operator op, unsigned int nvals, grid_coord_2d coords[], double vals[])
{
	auto *o = reinterpret_cast<op_t *>(op);
	if (!o) return;
	const topo_t &t = o->topo;
	const size_t II = t.nl[0] + 2, JJ = t.nl[1] + 2, P = II * JJ;
	for (unsigned n = 0; n < nvals; n++) {
		// 0-based array index incl. the ghost cell: coords - is + 2 in the reference's 1-based terms
		size_t ci = (size_t)coords[n].i - t.is[0] + 2, cj = (size_t)coords[n].j - t.is[1] + 2;
		int dir = coords[n].dir;
		if (dir != BMG2_C) vals[n] = -1 * vals[n]; // positive off-diagonals; the caller's array is modified, as in the reference
		switch (dir) { // vertex based input -> symmetric storage
		case BMG2_SE: ci++; dir = BMG2_NW; break;
		case BMG2_N: cj++; dir = BMG2_S; break;
		case BMG2_NE: ci++; cj++; dir = BMG2_SW; break;
		case BMG2_E: ci++; dir = BMG2_W; break;
		case BMG2_NW: cj++; break;
		default: break;
		}
		if (ci >= II || cj >= JJ || dir < 0 || dir > 4) {
			report("bmg2_operator_set: entry outside the local grid ignored");
			continue;
		}
		o->so[dir * P + ci + II * cj] = vals[n];
	}
	o->dirty = true;
}

void bmg3_operator_set(bmg3_operator op, unsigned int nvals, grid_coord_3d coords[], double vals[])
{
	auto *o = reinterpret_cast<op_t *>(op);
	if (!o) return;
	const topo_t &t = o->topo;
	const size_t II = t.nl[0] + 2, JJ = t.nl[1] + 2, KK = t.nl[2] + 2, P = II * JJ * KK;
	for (unsigned n = 0; n < nvals; n++) {
		const size_t ci = (size_t)coords[n].i - t.is[0] + 2, cj = (size_t)coords[n].j - t.is[1] + 2,
		             ck = (size_t)coords[n].k - t.is[2] + 2;
		const int dir = coords[n].dir;
		if (dir != BMG3_P) vals[n] = -1 * vals[n];
		if (ci >= II || cj >= JJ || ck >= KK || dir < 0 || dir > 13) {
			report("bmg3_operator_set: entry outside the local grid ignored");
			continue;
		}
		o->so[dir * P + ci + II * (cj + JJ * ck)] = vals[n];
	}
	o->dirty = true;
}

void bmg2_operator_apply(bmg2_operator op, const double *x, double *b) { if (op) apply(reinterpret_cast<op_t *>(op), x, b); }
void bmg3_operator_apply(bmg3_operator op, const double *x, double *b) { if (op) apply(reinterpret_cast<op_t *>(op), x, b); }
void bmg2_operator_dump(bmg2_operator op) { if (op) dump(reinterpret_cast<op_t *>(op)); }
void bmg3_operator_dump(bmg3_operator op) { if (op) dump(reinterpret_cast<op_t *>(op)); }
void bmg2_operator_destroy(bmg2_operator op) { delete reinterpret_cast<op_t *>(op); }
void bmg3_operator_destroy(bmg3_operator op) { delete reinterpret_cast<op_t *>(op); }

bmg2_solver bmg2_solver_create(bmg2_operator *op) { return reinterpret_cast<bmg2_solver>(op ? make_solver(reinterpret_cast<op_t *>(*op)) : nullptr); }
bmg3_solver bmg3_solver_create(bmg3_operator *op) { return reinterpret_cast<bmg3_solver>(op ? make_solver(reinterpret_cast<op_t *>(*op)) : nullptr); }
void bmg2_solver_run(bmg2_solver s, double *x, const double *b) { run(reinterpret_cast<slv_t *>(s), x, b); }
void bmg3_solver_run(bmg3_solver s, double *x, const double *b) { run(reinterpret_cast<slv_t *>(s), x, b); }

static void destroy_solver(slv_t *s)
{
	if (!s) return;
	if (s->h) cedar_amd_solver_destroy(s->h);
	if (s->d2) cedar_amd_dist2_destroy(s->d2);
	if (s->d3) cedar_amd_dist3_destroy(s->d3);
	delete s;
}
void bmg2_solver_destroy(bmg2_solver s) { destroy_solver(reinterpret_cast<slv_t *>(s)); }
void bmg3_solver_destroy(bmg3_solver s) { destroy_solver(reinterpret_cast<slv_t *>(s)); }

// src/interface/c/timer.cc:8-12 -> cedar::timer_save: one JSON object, seconds and call counts per phase
void bmg_timer_save(const char *fname)
{
	std::ofstream f(fname, std::ios::out | std::ios::trunc);
	f << "{";
	bool first = true;
	for (auto &kv : g_timers) {
		f << (first ? "" : ",") << "\n  \"" << kv.first << "\": {\"seconds\": " << std::setprecision(9) << kv.second
		  << ", \"calls\": " << g_counts[kv.first] << "}";
		first = false;
	}
	f << "\n}\n";
}

} // extern "C"
