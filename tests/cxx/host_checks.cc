// Host-side checks of the C++ mirror of Cedar's API (include/cedar): config reader, arrays and grid
// functions, the gallery builders and the named-kernel registry.  No device call is made.
// Driven by tests/test_cxx_api.py: argv[1] = output directory; prints one JSON object.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <cedar/2d/solver.h>
#include <cedar/3d/solver.h>

using namespace cedar;

template <class A> static void dump(const std::string & path, const A & a)
{
	std::ofstream f(path, std::ios::binary);
	f.write(reinterpret_cast<const char *>(a.data()), static_cast<std::streamsize>(a.size() * sizeof(real_t)));
}

// a registry entry that never touches the GPU
struct probe_kernel : kernel_base {
	static std::string name() { return "probe"; }
	virtual int run(int v) = 0;
};
struct probe_a : probe_kernel { int run(int v) override { return v + 1; } };
struct probe_b : probe_kernel {
	explicit probe_b(int scale) : s(scale) {}
	int run(int v) override { return v * s; }
	int s;
};
struct probe_runner : kernel_base { // run<T> returns void in the reference too: result through an argument
	static std::string name() { return "probe runner"; }
	void run(int v, int * out) { *out = v; }
};

int main(int argc, char ** argv)
{
	const std::string out = argc > 1 ? argv[1] : ".";
	config conf(out + "/config.json");
	auto n = conf.getvec<len_t>("grid.n");
	auto per = conf.getvec<int>("grid.periodic");
	ml_settings st;
	st.init(conf);
	auto params = build_kernel_params(conf);

	// gallery builders against tests/problems.py
	dump(out + "/poisson2.bin", cdr2::gallery::poisson(9, 7));
	dump(out + "/diag2.bin", cdr2::gallery::diag_diffusion(11, 6, 1.0, 1e-4));
	dump(out + "/fe2.bin", cdr2::gallery::fe(8, 10));
	dump(out + "/poisson3.bin", cdr3::gallery::poisson(6, 7, 5));
	dump(out + "/fe3.bin", cdr3::gallery::fe(5, 6, 7));

	// grid_func norms: inf_norm is the SIGNED entry of largest magnitude (src/2d/grid_func.cc:118-134)
	cdr2::grid_func g(4, 3);
	g(1, 1) = 0.5; g(4, 3) = -3.0; g(2, 2) = 2.0;
	g(0, 0) = 100.0; // ghost: not part of any norm
	const real_t inf = g.inf_norm(), l2 = g.lp_norm<2>();

	// registry: first added implementation is selected, set<T> switches, unknown names are reported
	kernel_manager km(params);
	km.add<probe_kernel, probe_a>("system");
	km.add<probe_kernel, probe_b>("hip", 7);
	const int r_first = km.get_ptr<probe_kernel>()->run(5);
	km.set<probe_kernel>("hip");
	const int r_hip = km.get_ptr<probe_kernel>()->run(5);
	km.set<probe_kernel>("does-not-exist"); // logs an error, keeps "hip"
	const int r_after_bad = km.get_ptr<probe_kernel>()->run(5);
	km.add<probe_runner, probe_runner>("system");
	int via_run = 0;
	km.run<probe_runner>(42, &via_run);

	std::printf("{\"nx\": %u, \"ny\": %u, \"periodic\": [%d, %d], \"per_mask\": %d, \"relaxation\": %d, \"pre\": %d, \"post\": %d, "
	            "\"maxiter\": %d, \"tol\": %.17g, \"cycle\": %d, \"inf_norm\": %.17g, \"l2\": %.17g, "
	            "\"r_first\": %d, \"r_hip\": %d, \"r_after_bad\": %d, \"via_run\": %d}\n",
	            n.size() > 0 ? n[0] : 0, n.size() > 1 ? n[1] : 0, per.size() > 0 ? per[0] : -1, per.size() > 1 ? per[1] : -1,
	            params->per_mask(), static_cast<int>(st.relaxation), st.nrelax_pre, st.nrelax_post, st.maxiter, st.tol, st.cycle,
	            inf, l2, r_first, r_hip, r_after_bad, via_run);
	return 0;
}
