// The reference's examples/basic-3d-ser/poisson.cc against this repository's C++ mirror; also
// exercises the per-kernel "hip" bindings through the kernel_manager (one relax sweep + residual).
#include <cmath>
#include <cedar/3d/solver.h>

using namespace cedar;
using namespace cedar::cdr3;

static void set_problem(grid_func & b)
{
	const double pi = M_PI;
	b.set(0);
	real_t hx = 1.0 / (b.len(0) - 1), hy = 1.0 / (b.len(1) - 1), hz = 1.0 / (b.len(2) - 1), h3 = hx * hy * hz;
	for (auto k : b.range(2)) for (auto j : b.range(1)) for (auto i : b.range(0))
		b(i, j, k) = 12 * (pi * pi) * sin(2 * pi * i * hx) * sin(2 * pi * j * hy) * sin(2 * pi * k * hz) * h3;
}

int main()
{
	config conf;
	auto ndofs = conf.getvec<len_t>("grid.n");
	len_t nx = ndofs.size() > 0 ? ndofs[0] : 64, ny = ndofs.size() > 1 ? ndofs[1] : 64, nz = ndofs.size() > 2 ? ndofs[2] : 64;
	auto so = gallery::poisson(nx, ny, nz);
	grid_func b(nx, ny, nz);
	set_problem(b);

	solver<seven_pt> bmg(so);
	auto sol = bmg.solve(b);

	// drive two kernels by hand through the registry, like a Cedar user with a custom cycle would
	auto kman = bmg.get_kernels();
	relax_stencil sor(nx, ny, nz);
	grid_func res(nx, ny, nz);
	kman->setup<kernels::point_relax<stypes>>(so, sor);
	kman->run<kernels::point_relax<stypes>>(so, sol, b, sor, cycle::Dir::DOWN);
	kman->run<kernels::residual<stypes>>(so, sol, b, res);
	log::status << "Levels: " << bmg.nlevels() << std::endl;
	log::status << "Residual l2 after one extra sweep: " << res.lp_norm<2>() << std::endl;
	real_t hx = 1.0 / (nx + 1);
	real_t err = 0;
	const double pi = M_PI;
	for (auto k : sol.range(2)) for (auto j : sol.range(1)) for (auto i : sol.range(0))
		err = std::max(err, std::abs(sol(i, j, k) - sin(2 * pi * i * hx) * sin(2 * pi * j / (ny + 1.)) * sin(2 * pi * k / (nz + 1.))));
	log::status << "Solution norm: " << err << std::endl;
	return (bmg.history.back() < 1e-6 && err < 5e-3) ? 0 : 1;
}
