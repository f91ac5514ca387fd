// Multilevel driver (reference include/cedar/multilevel.h:30-310, include/cedar/cycle/vcycle.h:44-115,
// include/cedar/cycle/fcycle.h:49-83) in two modes behind one interface:
//
//   resident      every selected kernel is the library's own ("hip"): the hierarchy is set up and cycled on the
//                 device by the handle API of libcedar_amd.so (one hipGraph replay per cycle), only x, b and the
//                 norms cross PCIe.  This is the product path.
//   orchestrated  some kernel was registered by the user and selected with set<T>(name): the driver below runs the
//                 reference's own sequence -- setup_interp / coarsen_op / relax set-up per level, V- or F-cycle,
//                 residual norms -- on host arrays through kernel_manager::run<T>(), so that kernel is really
//                 executed (the "hip" kernels beside it stage their arrays through HBM per call).
//
// `levels` gives the reference's view of the hierarchy (levels.get(i).A / .P / .SOR ...); in resident mode the
// host arrays are filled from HBM on first access.
#ifndef CEDAR_MULTILEVEL_H
#define CEDAR_MULTILEVEL_H
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>
#include <cedar/kernel_manager.h>
#include <cedar/kernels/coarsen_op.h>
#include <cedar/kernels/interp_add.h>
#include <cedar/kernels/line_relax.h>
#include <cedar/kernels/point_relax.h>
#include <cedar/kernels/residual.h>
#include <cedar/kernels/restrict.h>
#include <cedar/kernels/setup_interp.h>
#include <cedar/kernels/solve_cg.h>
#include <cedar/level.h>
extern "C" {
#include <cedar_amd.h>
}

namespace cedar {

// child (CRTP, as in the reference) supplies: compute_num_levels(fop), setup_space(nlevels),
// setup_relax_level(level), smooth(level, A, x, b, dir, n), create_handle(settings) and download(level index, level)
template <class level_container, class fsten, class child> class multilevel {
public:
	template <class sten> using level_t = typename level_container::template level_t<sten>;
	using stypes = typename level_t<fsten>::stypes;
	template <class sten> using stencil_op = typename stypes::template stencil_op<sten>;
	using full_sten = typename stypes::full_sten;
	using grid_func = typename stypes::grid_func;
	using coarsen_op = kernels::coarsen_op<stypes>;
	using interp_add = kernels::interp_add<stypes>;
	using residual = kernels::residual<stypes>;
	using restriction = kernels::restriction<stypes>;
	using setup_prolong = kernels::setup_interp<stypes>;
	using solve_cg = kernels::solve_cg<stypes>;
	using point_relax = kernels::point_relax<stypes>;
	using conf_ptr = std::shared_ptr<config>;

	multilevel(stencil_op<fsten> & fop) : multilevel(fop, std::make_shared<config>("config.json")) {}
	multilevel(stencil_op<fsten> & fop, conf_ptr cfg) : levels(fop), conf(cfg)
	{
		settings.init(*conf);
		levels.touch = [this]() { this->host_levels(); };
	}
	virtual ~multilevel()
	{
		cedar_amd_solver_destroy(h);
		delete[] bbd;
	}
	multilevel(const multilevel &) = delete;

	std::shared_ptr<kernel_manager> get_kernels() { return kman; }
	config & get_config() { return *conf; }
	std::size_t nlevels() { return h ? (std::size_t)cedar_amd_solver_nlevels(h) : nlev; }
	// true while the device-resident path serves solve() / vcycle()
	bool resident() const { return h != nullptr && !force_orchestrated && kman->all_selected("hip"); }

	virtual grid_func solve(const grid_func & b)
	{
		grid_func x = grid_func::zeros_like(b);
		solve(b, x);
		return x;
	}

	virtual void solve(const grid_func & b, grid_func & x)
	{
		history.clear();
		if (resident()) {
			std::vector<real_t> rel(settings.maxiter + 1);
			int n = cedar_amd_solver_solve(h, b.data(), x.data(), rel.data());
			log::info << "Initial residual l2 norm: " << rel[0] << std::endl;
			for (int i = 0; i < n; i++) log::status << "Iteration " << i << " relative l2 norm: " << rel[i + 1] << std::endl;
			history.assign(rel.begin(), rel.begin() + n + 1);
			return;
		}
		if (!ready()) return;
		// reference include/cedar/multilevel.h:277-298
		auto & level = levels.template get<fsten>(0);
		kman->template run<residual>(level.A, x, b, level.res);
		real_t res0_l2 = level.res.template lp_norm<2>();
		log::info << "Initial residual l2 norm: " << res0_l2 << std::endl;
		history.push_back(res0_l2);
		for (int i = 0; i < settings.maxiter; i++) {
			cycle_run(x, b);
			kman->template run<residual>(level.A, x, b, level.res);
			real_t rel_l2 = level.res.template lp_norm<2>() / res0_l2;
			log::status << "Iteration " << i << " relative l2 norm: " << rel_l2 << std::endl;
			history.push_back(rel_l2);
			if (rel_l2 < settings.tol) break;
		}
	}

	void vcycle(grid_func & x, const grid_func & b)
	{
		if (resident()) { cedar_amd_solver_vcycle(h, x.data(), b.data()); return; }
		if (ready()) cycle_run(x, b);
	}

	// the reference's set-up loop through the kernel manager (multilevel.h:243-265) on host arrays
	void setup(stencil_op<fsten> & fop)
	{
		const bool outer = in_touch;
		in_touch = true; // the levels handed out below are being built: no lazy download underneath
		setup_impl(fop);
		in_touch = outer;
	}

	level_container levels;
	std::vector<real_t> history; // [||r0||, rel_1, ...] of the last solve
	bool force_orchestrated = false; // run the orchestrated driver even when every kernel is "hip"

protected:
	void setup_impl(stencil_op<fsten> & fop)
	{
		std::size_t num_levels = static_cast<child *>(this)->compute_num_levels(fop);
		if (settings.num_levels > 0) {
			if (static_cast<std::size_t>(settings.num_levels) > num_levels) log::error << "too many levels specified" << std::endl;
			else num_levels = settings.num_levels;
		}
		nlev = num_levels;
		if (!space_ready) { static_cast<child *>(this)->setup_space(num_levels); space_ready = true; }
		for (std::size_t i = 0; i + 1 < num_levels; ++i) {
			setup_interp(i);
			setup_operator(i);
			setup_relax(i);
		}
		setup_cg_solve();
		host_setup_done = true;
	}
	// host arrays of every level exist and hold the hierarchy (downloaded, or set up through the kernel manager)
	void host_levels()
	{
		if (host_setup_done || in_touch) return;
		in_touch = true;
		if (h && kman->all_selected("hip")) {
			nlev = cedar_amd_solver_nlevels(h);
			if (!space_ready) { static_cast<child *>(this)->setup_space(nlev); space_ready = true; }
			for (std::size_t l = 0; l < nlev; l++) static_cast<child *>(this)->download(l);
			host_setup_done = true;
		} else {
			setup(levels.fine.A);
		}
		in_touch = false;
	}
	bool ready()
	{
		// a kernel selection made after construction invalidates the downloaded hierarchy only in so far as set-up
		// kernels changed; redo the set-up through the manager so that user set-up kernels are honoured too
		const bool all_hip = kman->all_selected("hip");
		if (!host_setup_done || (!all_hip && !setup_through_manager)) {
			in_touch = true;
			host_setup_done = false;
			setup(levels.fine.A);
			setup_through_manager = true;
			in_touch = false;
		}
		return true;
	}
	void cycle_run(grid_func & x, const grid_func & b)
	{
		if (settings.cycle == 1) { // F-cycle (fcycle.h:38-47)
			if (nlev == 1) coarse_solver(x, b);
			else fmg_cycle(0, x, b);
		} else {
			if (nlev == 1) coarse_solver(x, b);
			else ncycle(0, x, b);
		}
	}
	void ncycle(std::size_t lvl, grid_func & x, const grid_func & b)
	{
		if (lvl == 0) ncycle_helper(levels.template get<fsten>(0), lvl, x, b);
		else ncycle_helper(levels.get(lvl), lvl, x, b);
	}
	// vcycle.h:57-115
	template <class sten> void ncycle_helper(level_t<sten> & level, std::size_t lvl, grid_func & x, const grid_func & b)
	{
		auto & A = level.A;
		level.presmoother(A, x, b);
		grid_func & res = level.res;
		kman->template run<residual>(A, x, b, res);
		auto & coarse_level = levels.get(lvl + 1);
		auto & coarse_b = coarse_level.b;
		auto & coarse_x = coarse_level.x;
		kman->template run<restriction>(coarse_level.R, res, coarse_b);
		coarse_x.set(0.0);
		if (lvl + 1 == nlev - 1) coarse_solver(coarse_x, coarse_b);
		else ncycle(lvl + 1, coarse_x, coarse_b);
		kman->template run<interp_add>(coarse_level.P, coarse_x, res, x);
		level.postsmoother(A, x, b);
	}
	// fcycle.h:49-83
	void fmg_cycle(std::size_t lvl, grid_func & x, const grid_func & b)
	{
		if (lvl == nlev - 1) { coarse_solver(x, b); return; }
		auto & coarse_level = levels.get(lvl + 1);
		kman->template run<restriction>(coarse_level.R, b, coarse_level.b);
		fmg_cycle(lvl + 1, coarse_level.x, coarse_level.b);
		x.set(0.0);
		if (lvl == 0) {
			auto & level = levels.template get<fsten>(0);
			level.res.set(0.0);
			kman->template run<interp_add>(coarse_level.P, coarse_level.x, level.res, x);
		} else {
			auto & level = levels.get(lvl);
			level.res.set(0.0);
			kman->template run<interp_add>(coarse_level.P, coarse_level.x, level.res, x);
		}
		ncycle(lvl, x, b);
	}
	void setup_interp(std::size_t lvl)
	{
		auto & P = levels.get(lvl + 1).P;
		auto & cop = levels.get(lvl + 1).A;
		if (lvl == 0) kman->template run<setup_prolong>(levels.template get<fsten>(0).A, cop, P);
		else kman->template run<setup_prolong>(levels.get(lvl).A, cop, P);
	}
	void setup_operator(std::size_t lvl)
	{
		auto & P = levels.get(lvl + 1).P;
		auto & cop = levels.get(lvl + 1).A;
		if (lvl == 0) kman->template run<coarsen_op>(P, levels.template get<fsten>(0).A, cop);
		else kman->template run<coarsen_op>(P, levels.get(lvl).A, cop);
	}
	void setup_relax(std::size_t lvl)
	{
		if (lvl == 0) setup_relax_helper(levels.template get<fsten>(0));
		else setup_relax_helper(levels.get(lvl));
	}
	template <class sten> void setup_relax_helper(level_t<sten> & level)
	{
		child * c = static_cast<child *>(this);
		c->setup_relax_level(level);
		level.presmoother = [c, &level](const stencil_op<sten> & A, grid_func & x, const grid_func & b) {
			c->smooth(level, A, x, b, cycle::Dir::DOWN);
		};
		level.postsmoother = [c, &level](const stencil_op<sten> & A, grid_func & x, const grid_func & b) {
			c->smooth(level, A, x, b, cycle::Dir::UP);
		};
	}
	virtual void setup_cg_solve()
	{
		if (nlev == 1) // single level: the fine operator itself is factorised (vcycle.h:37-38)
			kman->template setup<solve_cg>(levels.template get<fsten>(0).A, ABD);
		else
			kman->template setup<solve_cg>(levels.get(nlev - 1).A, ABD);
		auto kernels = get_kernels();
		coarse_solver = [this, kernels](grid_func & x, const grid_func & b) { kernels->template run<solve_cg>(x, b, ABD, bbd); };
	}

	conf_ptr conf;
	ml_settings settings;
	std::shared_ptr<kernel_manager> kman;
	std::function<void(grid_func & x, const grid_func & b)> coarse_solver;
	grid_func ABD;
	real_t * bbd = nullptr;
	cedar_amd_solver * h = nullptr;
	std::size_t nlev = 0;
	bool space_ready = false, host_setup_done = false, setup_through_manager = false, in_touch = false;
};
}
#endif
