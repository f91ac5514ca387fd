// The reference's examples/basic-3d-ser/periodic.cc written against this repository's mirror of Cedar's C++
// surface: a seven-point Poisson problem that is periodic in the directions "grid.periodic" names, solved on the
// MI355X.  The operator carries the periodic image in its ghost layers and the mesh width counts one interval
// fewer in a periodic direction, as in the reference's create_op (:15-125), set_problem (:128-193) and
// set_solution (:196-228).  Extents in periodic directions must stay even on every level that is coarsened.
//   make -C examples && (cd examples && ./ser-periodic-3d)        # reads ./periodic-config-3d.json
#include <array>
#include <cmath>
#include <cedar/3d/solver.h>

using namespace cedar;
using namespace cedar::cdr3;

static stencil_op<seven_pt> create_op(len_t nx, len_t ny, len_t nz, std::array<bool, 3> per)
{
	stencil_op<seven_pt> so(nx, ny, nz);
	so.set(0);
	const len_t mx = nx - (per[0] ? 1 : 0), my = ny - (per[1] ? 1 : 0), mz = nz - (per[2] ? 1 : 0);
	const real_t hx = 1.0 / (mx + 1), hy = 1.0 / (my + 1), hz = 1.0 / (mz + 1);
	const real_t xh = hy * hz / hx, yh = hx * hz / hy, zh = hx * hy / hz;
	const len_t l = so.shape(0), m = so.shape(1), n = so.shape(2);
	const len_t ibeg = per[0] ? 1 : 2, jbeg = per[1] ? 1 : 2, kbeg = per[2] ? 1 : 2;
	for (len_t k = 1; k <= n; k++) for (len_t j = jbeg; j <= m; j++) for (len_t i = 1; i <= l; i++) so(i, j, k, seven_pt::ps) = 1.0 * yh;
	for (len_t k = 1; k <= n; k++) for (len_t j = 1; j <= m; j++) for (len_t i = ibeg; i <= l; i++) so(i, j, k, seven_pt::pw) = 1.0 * xh;
	for (len_t k = kbeg; k <= n; k++) for (len_t j = 1; j <= m; j++) for (len_t i = 1; i <= l; i++) so(i, j, k, seven_pt::b) = 1.0 * zh;
	for (auto k : so.range(2)) for (auto j : so.range(1)) for (auto i : so.range(0)) so(i, j, k, seven_pt::p) = 2.0 * xh + 2.0 * yh + 2.0 * zh;
	const seven_pt dirs[4] = { seven_pt::p, seven_pt::pw, seven_pt::ps, seven_pt::b };
	if (per[0])
		for (auto k : so.grange(2)) for (auto j : so.grange(1)) for (auto d : dirs) {
			so(ibeg - 1, j, k, d) = so(l, j, k, d);
			so(l + 1, j, k, d) = so(ibeg, j, k, d);
		}
	if (per[1])
		for (auto k : so.grange(2)) for (auto i : so.grange(0)) for (auto d : dirs) {
			so(i, jbeg - 1, k, d) = so(i, m, k, d);
			so(i, m + 1, k, d) = so(i, jbeg, k, d);
		}
	if (per[2])
		for (auto j : so.grange(1)) for (auto i : so.grange(0)) for (auto d : dirs) {
			so(i, j, kbeg - 1, d) = so(i, j, n, d);
			so(i, j, n + 1, d) = so(i, j, kbeg, d);
		}
	return so;
}

static void mesh(const grid_func & g, std::array<bool, 3> per, real_t & hx, real_t & hy, real_t & hz)
{
	hx = 1.0 / (g.len(0) - 2 - (per[0] ? 1 : 0) + 1);
	hy = 1.0 / (g.len(1) - 2 - (per[1] ? 1 : 0) + 1);
	hz = 1.0 / (g.len(2) - 2 - (per[2] ? 1 : 0) + 1);
}

static void set_problem(grid_func & b, std::array<bool, 3> per)
{
	const double pi = M_PI;
	b.set(0);
	real_t hx, hy, hz;
	mesh(b, per, hx, hy, hz);
	const real_t h2 = hx * hy * hz;
	for (auto k : b.range(2)) for (auto j : b.range(1)) for (auto i : b.range(0))
		b(i, j, k) = 12 * (pi * pi) * sin(2 * pi * (i * hx)) * sin(2 * pi * (j * hy)) * sin(2 * pi * (k * hz)) * h2;
	if (per[0])
		for (auto k : b.grange(2)) for (auto j : b.grange(1)) { b(0, j, k) = b(b.shape(0), j, k); b(b.shape(0) + 1, j, k) = b(1, j, k); }
	if (per[1])
		for (auto k : b.grange(2)) for (auto i : b.grange(0)) { b(i, 0, k) = b(i, b.shape(1), k); b(i, b.shape(1) + 1, k) = b(i, 1, k); }
	if (per[2])
		for (auto j : b.grange(1)) for (auto i : b.grange(0)) { b(i, j, 0) = b(i, j, b.shape(2)); b(i, j, b.shape(2) + 1) = b(i, j, 1); }
}

int main()
{
	auto conf = std::make_shared<config>("periodic-config-3d.json");
	auto params = build_kernel_params(*conf);
	auto ndofs = conf->getvec<len_t>("grid.n");
	const len_t nx = ndofs.size() > 0 ? ndofs[0] : 32, ny = ndofs.size() > 1 ? ndofs[1] : 32, nz = ndofs.size() > 2 ? ndofs[2] : 32;

	auto so = create_op(nx, ny, nz, params->periodic);
	grid_func b(nx, ny, nz);
	set_problem(b, params->periodic);

	solver<seven_pt> bmg(so, conf);
	auto sol = bmg.solve(b);

	const double pi = M_PI;
	real_t hx, hy, hz, err = 0;
	mesh(sol, params->periodic, hx, hy, hz);
	for (auto k : sol.range(2)) for (auto j : sol.range(1)) for (auto i : sol.range(0))
		err = std::max(err, std::abs(sol(i, j, k) - sin(2 * pi * (i * hx)) * sin(2 * pi * (j * hy)) * sin(2 * pi * (k * hz))));
	log::status << "periodic: " << params->periodic[0] << " " << params->periodic[1] << " " << params->periodic[2]
	            << "  levels: " << bmg.nlevels() << std::endl;
	log::status << "Solution norm: " << err << std::endl;
	log::status << "Finished Test" << std::endl;
	return (bmg.history.back() < 1e-6 && err < 1e-2) ? 0 : 1;
}
