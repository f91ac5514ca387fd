// The "hip" kernels behind the mirror's kernel_manager (what Cedar's multilevel calls through
// kman->setup<T>/run<T>): one point-relaxation sweep pair and a residual on gallery::fe, dumped for
// tests/test_cxx_api.py to compare bit for bit with the oracle.  Needs a GPU.
#include <fstream>
#include <cedar/2d/solver.h>

using namespace cedar;
using namespace cedar::cdr2;

template <class A> static void dump(const std::string & path, const A & a)
{
	std::ofstream f(path, std::ios::binary);
	f.write(reinterpret_cast<const char *>(a.data()), static_cast<std::streamsize>(a.size() * sizeof(real_t)));
}

int main(int argc, char ** argv)
{
	const std::string out = argc > 1 ? argv[1] : ".";
	config conf(out + "/config.json");
	auto kman = build_kernel_manager(conf);
	const len_t nx = 37, ny = 22;
	auto so = gallery::fe(nx, ny);
	grid_func x(nx, ny), b(nx, ny), r(nx, ny);
	for (auto j : x.range(1)) for (auto i : x.range(0)) { x(i, j) = 0.01 * i - 0.02 * j; b(i, j) = 1.0 / (1 + i + j); }
	relax_stencil sor(nx, ny);
	const int nst = stencil_ndirs<nine_pt>::value;
	kman->setup<kernels::point_relax>(so.data(), nst, sor);
	kman->run<kernels::point_relax>(so.data(), nst, x, b, sor, cycle::Dir::DOWN);
	kman->run<kernels::point_relax>(so.data(), nst, x, b, sor, cycle::Dir::UP);
	kman->run<kernels::residual>(so.data(), nst, x, b, r);
	dump(out + "/x.bin", x);
	dump(out + "/r.bin", r);
	return 0;
}
