// cedar::cdr2::solver (reference include/cedar/2d/solver.h:20-122 over include/cedar/multilevel.h): same
// constructor, solve / vcycle / levels / get_kernels / get_config / give_op.  The hierarchy lives in HBM behind
// the handle API of libcedar_amd.so while every selected kernel is "hip"; a kernel the caller registers and selects
// switches the solver to the reference's per-kernel orchestration (include/cedar/multilevel.h of this mirror).
// Ownership as in the reference: level 0 refers to the caller's operator, which must outlive the solver unless
// give_op() hands it over.  Errors go to log::error, nothing throws.
#ifndef CEDAR_2D_SOLVER_H
#define CEDAR_2D_SOLVER_H
#include <algorithm>
#include <array>
#include <cedar/multilevel.h>
#include <cedar/2d/gallery.h>
#include <cedar/2d/kernel_manager.h>

namespace cedar { namespace cdr2 {

// include/cedar/2d/solver.h:20-37
template <class sten> struct level2 : public level<sten, stypes> {
	using parent = level<sten, stypes>;
	level2(len_t nx, len_t ny) : parent::level(nx, ny)
	{
		this->SOR = {{relax_stencil(nx, ny), relax_stencil(nx, ny)}};
		this->R.associate(&this->P);
	}
	level2(stencil_op<sten> & A) : parent::level(A)
	{
		this->res = grid_func(A.shape(0), A.shape(1));
		this->SOR = {{relax_stencil(A.shape(0), A.shape(1)), relax_stencil(A.shape(0), A.shape(1))}};
	}
};

template <class fsten> class solver : public multilevel<level_container<level2, fsten, nine_pt>, fsten, solver<fsten>> {
public:
	using parent = multilevel<level_container<level2, fsten, nine_pt>, fsten, solver<fsten>>;
	template <class sten> using level_t = level2<sten>;
	explicit solver(stencil_op<fsten> & fop) : parent(fop) { init(fop); }
	solver(stencil_op<fsten> & fop, std::shared_ptr<config> conf) : parent(fop, conf) { init(fop); }

	// include/cedar/2d/solver.h:57-73 (float arithmetic on unsigned integer quotients)
	std::size_t compute_num_levels(stencil_op<fsten> & fop)
	{
		float nxc, nyc;
		int ng = 0;
		auto min_coarse = this->settings.min_coarse;
		auto nx = fop.shape(0), ny = fop.shape(1);
		do {
			ng++;
			nxc = (nx - 1) / (1u << ng) + 1;
			nyc = (ny - 1) / (1u << ng) + 1;
		} while (std::min(nxc, nyc) >= min_coarse);
		return ng;
	}
	// include/cedar/2d/solver.h:75-116
	void setup_space(std::size_t nlevels)
	{
		auto params = this->kman->get_params();
		len_t nx = this->levels.fine.A.shape(0), ny = this->levels.fine.A.shape(1);
		for (std::size_t i = 0; i + 1 < nlevels; i++) {
			len_t nxc = (nx - 1) / 2. + 1, nyc = (ny - 1) / 2. + 1;
			this->levels.add(nxc, nyc);
			nx = nxc; ny = nyc;
		}
		len_t abd_len_0 = nx + 2;
		if (params->periodic[0] || params->periodic[1]) abd_len_0 = nx * ny;
		this->ABD = grid_func(abd_len_0, nx * ny, 0);
		this->bbd = new real_t[this->ABD.len(1)];
	}
	// multilevel.h:149-166
	template <class sten> void setup_relax_level(level2<sten> & level)
	{
		using rt = ml_settings::relax_type;
		auto & km = this->kman;
		switch (this->settings.relaxation) {
		case rt::point: km->template setup<kernels::point_relax<stypes>>(level.A, level.SOR[0]); break;
		case rt::line_x: km->template setup<kernels::line_relax<stypes, relax_dir::x>>(level.A, level.SOR[0]); break;
		case rt::line_y: km->template setup<kernels::line_relax<stypes, relax_dir::y>>(level.A, level.SOR[0]); break;
		case rt::line_xy:
			km->template setup<kernels::line_relax<stypes, relax_dir::x>>(level.A, level.SOR[0]);
			km->template setup<kernels::line_relax<stypes, relax_dir::y>>(level.A, level.SOR[1]);
			break;
		default: log::error << "cdr2::solver: plane relaxation is a 3D smoother" << std::endl;
		}
	}
	// multilevel.h:170-222: pre = DOWN (line-xy: x then y), post = UP (y then x)
	template <class sten> void smooth(level2<sten> & level, const stencil_op<sten> & A, grid_func & x, const grid_func & b, cycle::Dir dir)
	{
		using rt = ml_settings::relax_type;
		using lx = kernels::line_relax<stypes, relax_dir::x>;
		using ly = kernels::line_relax<stypes, relax_dir::y>;
		auto & km = this->kman;
		const int n = dir == cycle::Dir::DOWN ? this->settings.nrelax_pre : this->settings.nrelax_post;
		for (int i = 0; i < n; i++) {
			switch (this->settings.relaxation) {
			case rt::point: km->template run<kernels::point_relax<stypes>>(A, x, b, level.SOR[0], dir); break;
			case rt::line_x: km->template run<lx>(A, x, b, level.SOR[0], level.res, dir); break;
			case rt::line_y: km->template run<ly>(A, x, b, level.SOR[0], level.res, dir); break;
			case rt::line_xy:
				if (dir == cycle::Dir::DOWN) {
					km->template run<lx>(A, x, b, level.SOR[0], level.res, dir);
					km->template run<ly>(A, x, b, level.SOR[1], level.res, dir);
				} else {
					km->template run<ly>(A, x, b, level.SOR[1], level.res, dir);
					km->template run<lx>(A, x, b, level.SOR[0], level.res, dir);
				}
				break;
			default: break;
			}
		}
	}
	// host view of level l from the device-resident hierarchy
	void download(std::size_t l)
	{
		auto fetch = [&](const char * what, real_t * dst, std::size_t n) {
			if (cedar_amd_solver_get(this->h, (int)l, what, nullptr) == n) cedar_amd_solver_get(this->h, (int)l, what, dst);
		};
		if (l == 0) {
			auto & L = this->levels.fine;
			fetch("SOR0", L.SOR[0].data(), L.SOR[0].size());
			fetch("SOR1", L.SOR[1].data(), L.SOR[1].size());
			return;
		}
		auto & L = this->levels.coarse[l - 1];
		fetch("A", L.A.data(), L.A.size());
		fetch("P", L.P.data(), L.P.size());
		fetch("SOR0", L.SOR[0].data(), L.SOR[0].size());
		fetch("SOR1", L.SOR[1].data(), L.SOR[1].size());
		if (l == 1) { L.P.fine_is_five = std::is_same<fsten, five_pt>::value; set_fine(L.P, this->levels.fine.A); }
		else { L.P.fine_is_five = false; L.P.fine_op_nine = &this->levels.coarse[l - 2].A; }
	}
	void give_op(std::unique_ptr<stencil_op<fsten>> fop) { fop_ref = std::move(fop); }

protected:
	static void set_fine(prolong_op & P, stencil_op<five_pt> & A) { P.fine_op_five = &A; }
	static void set_fine(prolong_op & P, stencil_op<nine_pt> & A) { P.fine_op_nine = &A; }
	void init(stencil_op<fsten> & fop)
	{
		this->kman = build_kernel_manager(*this->conf);
		cedar_amd_settings st;
		cedar_amd_default_settings(&st);
		st.relaxation = static_cast<int>(this->settings.relaxation);
		st.nrelax_pre = this->settings.nrelax_pre;
		st.nrelax_post = this->settings.nrelax_post;
		st.num_levels = this->settings.num_levels;
		st.max_iter = this->settings.maxiter;
		st.tol = this->settings.tol;
		st.min_coarse = this->settings.min_coarse;
		st.cycle = this->settings.cycle;
		BMG_get_bc(this->kman->get_params()->per_mask(), &st.ibc); // grid.periodic -> boundary code, as every reference binding does
		this->h = cedar_amd_solver_create(2, fop.shape(0), fop.shape(1), 1, stencil_ndirs<fsten>::value, fop.data(), 0, &st);
		if (!this->h)
			log::error << "cdr2::solver: the device-resident solver could not be created for these settings (reported above); "
			              "falling back to the per-kernel orchestration" << std::endl;
	}
	std::unique_ptr<stencil_op<fsten>> fop_ref;
};
}}
#endif
