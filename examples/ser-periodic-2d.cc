// The reference's examples/basic-2d-ser/periodic.cc written against this repository's mirror of
// Cedar's C++ surface: a five-point Poisson problem that is periodic in the directions
// config.json's "grid.periodic" names (the reference ships periodic-config.json: [true, false],
// V(1,1), point relaxation), solved on the MI355X.  The operator carries the periodic image in its
// ghost columns / rows and the mesh width counts one interval fewer in a periodic direction, as in
// the reference's create_op (:17-84), set_problem (:87-125) and set_solution (:128-150).
//   make -C examples && (cd examples && ./ser-periodic-2d)        # reads ./periodic-config.json
#include <array>
#include <cmath>
#include <cedar/2d/solver.h>

using namespace cedar;
using namespace cedar::cdr2;

static stencil_op<five_pt> create_op(len_t nx, len_t ny, std::array<bool, 3> periodic)
{
	stencil_op<five_pt> so(nx, ny);
	so.set(0);
	const len_t mx = nx - (periodic[0] ? 1 : 0), my = ny - (periodic[1] ? 1 : 0);
	const real_t hx = 1.0 / (mx + 1), hy = 1.0 / (my + 1), xh = hy / hx, yh = hx / hy;
	const len_t l = so.shape(0), m = so.shape(1);
	const len_t ibeg = periodic[0] ? 1 : 2, jbeg = periodic[1] ? 1 : 2;
	for (len_t j = jbeg; j <= m; j++) for (len_t i = 1; i <= l; i++) so(i, j, five_pt::s) = 1.0 * yh;
	for (len_t j = 1; j <= m; j++) for (len_t i = ibeg; i <= l; i++) so(i, j, five_pt::w) = 1.0 * xh;
	for (auto j : so.range(1)) for (auto i : so.range(0)) so(i, j, five_pt::c) = 2 * xh + 2 * yh;
	const five_pt dirs[3] = { five_pt::c, five_pt::w, five_pt::s };
	if (periodic[0])
		for (auto j : so.range(1))
			for (auto d : dirs) {
				so(ibeg - 1, j, d) = so(l, j, d);
				so(l + 1, j, d) = so(ibeg, j, d);
			}
	if (periodic[1])
		for (auto i : so.range(0))
			for (auto d : dirs) {
				so(i, jbeg - 1, d) = so(i, m, d);
				so(i, m + 1, d) = so(i, jbeg, d);
			}
	return so;
}

static void set_problem(grid_func & b, std::array<bool, 3> periodic)
{
	const double pi = M_PI;
	b.set(0);
	const len_t nx = b.len(0) - 2 - (periodic[0] ? 1 : 0), ny = b.len(1) - 2 - (periodic[1] ? 1 : 0);
	const real_t hx = 1.0 / (nx + 1), hy = 1.0 / (ny + 1), h2 = hx * hy;
	for (auto j : b.range(1)) for (auto i : b.range(0))
		b(i, j) = 8 * (pi * pi) * sin(2 * pi * (i * hx)) * sin(2 * pi * (j * hy)) * h2;
	if (periodic[0])
		for (auto j : b.grange(1)) { b(0, j) = b(b.shape(0), j); b(b.shape(0) + 1, j) = b(1, j); }
	if (periodic[1])
		for (auto i : b.grange(0)) { b(i, 0) = b(i, b.shape(1)); b(i, b.shape(1) + 1) = b(i, 1); }
}

static void set_solution(grid_func & q, std::array<bool, 3> periodic)
{
	const double pi = M_PI;
	const len_t nx = q.len(0) - 2 - (periodic[0] ? 1 : 0), ny = q.len(1) - 2 - (periodic[1] ? 1 : 0);
	const real_t hx = 1.0 / (nx + 1), hy = 1.0 / (ny + 1);
	for (auto j : q.range(1)) for (auto i : q.range(0)) q(i, j) = sin(2 * pi * (i * hx)) * sin(2 * pi * (j * hy));
}

int main()
{
	auto conf = std::make_shared<config>("periodic-config.json");
	auto params = build_kernel_params(*conf);
	auto ndofs = conf->getvec<len_t>("grid.n");
	const len_t nx = ndofs.size() > 0 ? ndofs[0] : 300, ny = ndofs.size() > 1 ? ndofs[1] : 300;

	auto so = create_op(nx, ny, params->periodic);
	grid_func b(nx, ny);
	set_problem(b, params->periodic);

	solver<five_pt> bmg(so, conf);
	auto sol = bmg.solve(b);

	grid_func exact_sol(sol.shape(0), sol.shape(1));
	set_solution(exact_sol, params->periodic);
	auto diff = exact_sol - sol;
	log::status << "periodic: " << params->periodic[0] << " " << params->periodic[1] << "  levels: " << bmg.nlevels() << std::endl;
	log::status << "Solution norm: " << diff.inf_norm() << std::endl;
	log::status << "Finished Test" << std::endl;
	return std::abs(diff.inf_norm()) < 1e-2 ? 0 : 1;
}
