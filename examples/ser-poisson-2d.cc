// The reference's examples/basic-2d-ser/poisson.cc (ser-poisson-2d) written against this
// repository's mirror of Cedar's C++ surface: same calls, the solve runs on the MI355X.
//   make -C examples      (g++ -std=c++17 -Iinclude ... -lcedar_amd)
// Reads ./config.json like the reference ("grid.n", "solver.*"); defaults 512x512, V(2,1).
#include <cmath>
#include <cedar/2d/solver.h>

using namespace cedar;
using namespace cedar::cdr2;

static void set_problem(grid_func & b)
{
	const double pi = M_PI;
	auto rhs = [pi](real_t x, real_t y) { return 8 * (pi * pi) * sin(2 * pi * x) * sin(2 * pi * y); };
	b.set(0);
	real_t hx = 1.0 / (b.len(0) - 1), hy = 1.0 / (b.len(1) - 1), h2 = hx * hy;
	for (auto j : b.range(1)) for (auto i : b.range(0)) b(i, j) = rhs(i * hx, j * hy) * h2;
}

static void set_solution(grid_func & q)
{
	const double pi = M_PI;
	real_t hx = 1.0 / (q.len(0) - 1), hy = 1.0 / (q.len(1) - 1);
	for (auto j : q.grange(1)) for (auto i : q.grange(0)) q(i, j) = sin(2 * pi * i * hx) * sin(2 * pi * j * hy);
}

int main()
{
	config conf;
	auto ndofs = conf.getvec<len_t>("grid.n");
	len_t nx = ndofs.size() > 0 ? ndofs[0] : 512, ny = ndofs.size() > 1 ? ndofs[1] : 512;

	auto so = gallery::poisson(nx, ny);
	grid_func b(nx, ny);
	set_problem(b);

	solver<five_pt> bmg(so);
	auto sol = bmg.solve(b);

	grid_func exact_sol(sol.shape(0), sol.shape(1));
	set_solution(exact_sol);
	auto diff = exact_sol - sol;
	log::status << "Levels: " << bmg.nlevels() << std::endl;
	log::status << "Solution norm: " << diff.inf_norm() << std::endl;
	log::status << "Finished Test" << std::endl;
	return std::abs(diff.inf_norm()) < 1e-3 ? 0 : 1;
}
