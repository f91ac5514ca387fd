// The C++ plugin boundary on the GPU, driven by tests/test_cxx_api.py:
//  1. the registered "hip" kernels through kman->setup<T>/run<T> with the reference's signatures
//     (const stencil_op<nine_pt>&, ...): one point-relaxation sweep pair and a residual on gallery::fe, dumped for a
//     bit-for-bit comparison with the oracle;
//  2. a kernel written against the reference's abstract class (include/cedar/kernels/point_relax.h:31-71 overload
//     pairs, registered with add<T,impl>("user") and selected with set<T>("user")) is really run by solver::solve,
//     and the orchestrated solve reproduces the device-resident one;
//  3. solver.levels: host views of the device-resident hierarchy equal the hierarchy set up through the manager;
//  4. the same orchestrated / resident agreement for the 3D solver;
//  5. and for 3D plane relaxation (plane_relax<rdir> kernels through the registry).
#include <cstdio>
#include <fstream>
#include <cedar/2d/solver.h>
#include <cedar/3d/solver.h>

using namespace cedar;

template <class A> static void dump(const std::string & path, const A & a)
{
	std::ofstream f(path, std::ios::binary);
	f.write(reinterpret_cast<const char *>(a.data()), static_cast<std::streamsize>(a.size() * sizeof(real_t)));
}

// a user's point relaxation with the reference's exact virtual signatures; it counts its calls and leaves the
// arithmetic to the library's binding
struct user_relax : public kernels::point_relax<cdr2::stypes> {
	using cdr2_op5 = cdr2::stencil_op<cdr2::five_pt>;
	using cdr2_op9 = cdr2::stencil_op<cdr2::nine_pt>;
	explicit user_relax(std::shared_ptr<kernel_params> p) { inner.add_params(p); }
	void setup(const cdr2_op5 & so, cdr2::relax_stencil & sor) override { nsetup++; inner.setup(so, sor); }
	void setup(const cdr2_op9 & so, cdr2::relax_stencil & sor) override { nsetup++; inner.setup(so, sor); }
	void run(const cdr2_op5 & so, cdr2::grid_func & x, const cdr2::grid_func & b, const cdr2::relax_stencil & sor, cycle::Dir cdir) override
	{ nrun++; inner.run(so, x, b, sor, cdir); }
	void run(const cdr2_op9 & so, cdr2::grid_func & x, const cdr2::grid_func & b, const cdr2::relax_stencil & sor, cycle::Dir cdir) override
	{ nrun++; inner.run(so, x, b, sor, cdir); }
	cdr2::rbgs inner;
	static int nsetup, nrun;
};
int user_relax::nsetup = 0;
int user_relax::nrun = 0;

static void print_hist(const char * key, const std::vector<real_t> & h, bool last = false)
{
	std::printf("\"%s\": [", key);
	for (std::size_t i = 0; i < h.size(); i++) std::printf("%s%.17g", i ? ", " : "", h[i]);
	std::printf("]%s", last ? "" : ", ");
}

int main(int argc, char ** argv)
{
	using namespace cedar::cdr2;
	const std::string out = argc > 1 ? argv[1] : ".";
	auto conf = std::make_shared<config>(out + "/config.json");
	log::status.on = false;
	// ---- 1. per-kernel bindings
	{
		auto kman = build_kernel_manager(*conf);
		const len_t nx = 37, ny = 22;
		auto so = gallery::fe(nx, ny);
		grid_func x(nx, ny), b(nx, ny), r(nx, ny);
		for (auto j : x.range(1)) for (auto i : x.range(0)) { x(i, j) = 0.01 * i - 0.02 * j; b(i, j) = 1.0 / (1 + i + j); }
		relax_stencil sor(nx, ny);
		kman->setup<kernels::point_relax<stypes>>(so, sor);
		kman->run<kernels::point_relax<stypes>>(so, x, b, sor, cycle::Dir::DOWN);
		kman->run<kernels::point_relax<stypes>>(so, x, b, sor, cycle::Dir::UP);
		kman->run<kernels::residual<stypes>>(so, x, b, r);
		dump(out + "/x.bin", x);
		dump(out + "/r.bin", r);
	}
	// ---- 2. + 3. user kernel selected on a solver
	const len_t nx = 45, ny = 38;
	auto so = gallery::poisson(nx, ny);
	grid_func b(nx, ny);
	for (auto j : b.range(1)) for (auto i : b.range(0)) b(i, j) = 1e-3 * ((int)((i * 7 + j * 3) % 11) - 5);
	solver<five_pt> bmg(so, conf);
	const bool resident_before = bmg.resident();
	auto x1 = bmg.solve(b);
	auto h_res = bmg.history;
	const std::size_t nlev = bmg.nlevels();
	stencil_op<nine_pt> A1 = bmg.levels.get(1).A; // host view of the device-resident level 1
	auto kman = bmg.get_kernels();
	kman->add<kernels::point_relax<stypes>, user_relax>("user", kman->get_params());
	kman->set<kernels::point_relax<stypes>>("user");
	const bool resident_after = bmg.resident();
	auto x2 = bmg.solve(b);
	auto h_user = bmg.history;
	bool same_A1 = A1.size() == bmg.levels.get(1).A.size();
	for (std::size_t i = 0; same_A1 && i < A1.size(); i++) same_A1 = A1.data()[i] == bmg.levels.get(1).A.data()[i];
	real_t dx = 0, xm = 0;
	for (auto j : x1.range(1)) for (auto i : x1.range(0)) { dx = std::max(dx, std::abs(x1(i, j) - x2(i, j))); xm = std::max(xm, std::abs(x1(i, j))); }
	// ---- 4. 3D: orchestrated driver with the library's kernels against the resident solver
	std::vector<real_t> h3_res, h3_orc;
	{
		auto so3 = cdr3::gallery::fe(14, 12, 10);
		cdr3::grid_func b3(14, 12, 10);
		for (auto k : b3.range(2)) for (auto j : b3.range(1)) for (auto i : b3.range(0)) b3(i, j, k) = 1e-3 * ((int)((i * 7 + j * 3 + k * 5) % 11) - 5);
		cdr3::solver<cdr3::xxvii_pt> s3(so3, conf);
		s3.solve(b3);
		h3_res = s3.history;
		s3.force_orchestrated = true;
		s3.solve(b3);
		h3_orc = s3.history;
	}
	// ---- 5. 3D plane relaxation ("plane-xyz" with the default plane configuration): the plane_relax<rdir> kernels through
	//         the registry (orchestrated) against the resident solver
	std::vector<real_t> hp_res, hp_orc;
	{
		auto pconf = std::make_shared<config>(config::empty_tag());
		pconf->set("solver.relaxation", "plane-xyz");
		pconf->set("solver.max-iter", 4);
		auto so3 = cdr3::gallery::fe(13, 12, 11);
		cdr3::grid_func b3(13, 12, 11);
		for (auto k : b3.range(2)) for (auto j : b3.range(1)) for (auto i : b3.range(0)) b3(i, j, k) = 1e-3 * ((int)((i * 7 + j * 3 + k * 5) % 11) - 5);
		cdr3::solver<cdr3::xxvii_pt> s3(so3, pconf);
		s3.solve(b3);
		hp_res = s3.history;
		s3.force_orchestrated = true;
		s3.solve(b3);
		hp_orc = s3.history;
	}
	std::printf("{\"resident_before\": %d, \"resident_after\": %d, \"nlevels\": %zu, \"user_setup_calls\": %d, \"user_run_calls\": %d, "
	            "\"same_level1_operator\": %d, \"x_diff\": %.17g, \"x_max\": %.17g, ",
	            (int)resident_before, (int)resident_after, nlev, user_relax::nsetup, user_relax::nrun, (int)same_A1, dx, xm);
	print_hist("hist_resident", h_res);
	print_hist("hist_user", h_user);
	print_hist("hist3_resident", h3_res);
	print_hist("hist3_orchestrated", h3_orc);
	print_hist("hist3_planes_resident", hp_res);
	print_hist("hist3_planes_orchestrated", hp_orc, true);
	std::printf("}\n");
	return 0;
}
