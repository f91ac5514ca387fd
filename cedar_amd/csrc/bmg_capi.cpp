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
#include "common.h"
#include "stage.h"
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

// grid_topo of the reference reduced to what a single rank needs
struct topo_t {
	int nd;
	unsigned ng[3]; // global interior extents
	unsigned nl[3]; // local interior extents
	unsigned is[3]; // 1-based global index of the first local point (src/2d/interface/c/topo.cc:38-45)
	int nproc[3], coord[3];
};

struct op_t {
	topo_t topo;
	int nst;                // 5 (nine_pt) or 14 (xxvii_pt): the interface always uses the full stencil
	std::vector<real_t> so; // Cedar layout: (nst, [KK,] JJ, II), i fastest
	real_t *dso = nullptr, *dx = nullptr, *db = nullptr;
	bool dirty = true;
	size_t pts() const { return (size_t)(topo.nl[0] + 2) * (topo.nl[1] + 2) * (topo.nd == 3 ? topo.nl[2] + 2 : 1); }
	~op_t()
	{
		cedar_amd_free(dso);
		cedar_amd_free(dx);
		cedar_amd_free(db);
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
	cedar_amd_solver *h;
	cedar_amd_settings st;
	std::vector<real_t> xg, bg, rel;
};

topo_t *make_topo(int nd, const unsigned ng[3], unsigned *const ln[3], const int np[3])
{
	for (int d = 0; d < nd; d++) {
		if (np[d] != 1) {
			report("bmg_topo_create: this library runs one rank per GPU; the C interface serves nproc = 1 per direction "
			       "(multi-GPU runs use the domain-decomposed driver, DESIGN.md section 7)");
			return nullptr;
		}
	}
	auto *t = new topo_t();
	t->nd = nd;
	for (int d = 0; d < 3; d++) {
		t->ng[d] = d < nd ? ng[d] : 1;
		t->nl[d] = d < nd ? ln[d][0] : 1; // coord = 0: the first entry of the per-process extent list
		t->is[d] = 1;
		t->nproc[d] = 1;
		t->coord[d] = 0;
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
	s->h = cedar_amd_solver_create(t.nd, t.nl[0], t.nl[1], t.nd == 3 ? t.nl[2] : 1, o->nst, o->dso, 1, &s->st);
	if (!s->h) {
		delete s;
		return nullptr;
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
	cedar_amd_solver_solve(s->h, s->bg.data(), s->xg.data(), s->rel.data());
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
	cedar_amd_solver_destroy(s->h);
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
