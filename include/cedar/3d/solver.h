// cedar::cdr3::solver (reference include/cedar/3d/solver.h:18-130 over include/cedar/multilevel.h): same constructor,
// solve / vcycle / levels / get_kernels / get_config / give_op; device-resident while every selected kernel is "hip",
// per-kernel orchestration when the caller selects a kernel of its own (see include/cedar/multilevel.h of this mirror).
#ifndef CEDAR_3D_SOLVER_H
#define CEDAR_3D_SOLVER_H
#include <algorithm>
#include <array>
#include <cedar/multilevel.h>
#include <cedar/3d/gallery.h>
#include <cedar/3d/kernel_manager.h>

namespace cedar { namespace cdr3 {

// include/cedar/3d/solver.h:18-38
template <class sten> struct level3 : public level<sten, stypes> {
	using parent = level<sten, stypes>;
	level3(len_t nx, len_t ny, len_t nz) : parent::level(nx, ny, nz)
	{
		this->SOR = {{relax_stencil(nx, ny, nz), relax_stencil(nx, ny, nz)}};
		this->R.associate(&this->P);
	}
	level3(stencil_op<sten> & A) : parent::level(A)
	{
		this->res = grid_func(A.shape(0), A.shape(1), A.shape(2));
		this->SOR = {{relax_stencil(A.shape(0), A.shape(1), A.shape(2)), relax_stencil(A.shape(0), A.shape(1), A.shape(2))}};
	}
};

template <class fsten> class solver : public multilevel<level_container<level3, fsten, xxvii_pt>, fsten, solver<fsten>> {
public:
	using parent = multilevel<level_container<level3, fsten, xxvii_pt>, fsten, solver<fsten>>;
	template <class sten> using level_t = level3<sten>;
	explicit solver(stencil_op<fsten> & fop) : parent(fop) { init(fop); }
	solver(stencil_op<fsten> & fop, std::shared_ptr<config> conf) : parent(fop, conf) { init(fop); }

	// include/cedar/3d/solver.h:54-72
	std::size_t compute_num_levels(stencil_op<fsten> & fop)
	{
		float nxc, nyc, nzc;
		int ng = 0;
		auto min_coarse = this->settings.min_coarse;
		auto nx = fop.shape(0), ny = fop.shape(1), nz = fop.shape(2);
		do {
			ng++;
			nxc = (nx - 1) / (1u << ng) + 1;
			nyc = (ny - 1) / (1u << ng) + 1;
			nzc = (nz - 1) / (1u << ng) + 1;
		} while (std::min({nxc, nyc, nzc}) >= min_coarse);
		return ng;
	}
	// include/cedar/3d/solver.h:74-123
	void setup_space(std::size_t nlevels)
	{
		auto params = this->kman->get_params();
		len_t nx = this->levels.fine.A.shape(0), ny = this->levels.fine.A.shape(1), nz = this->levels.fine.A.shape(2);
		for (std::size_t i = 0; i + 1 < nlevels; i++) {
			len_t nxc = (nx - 1) / 2. + 1, nyc = (ny - 1) / 2. + 1, nzc = (nz - 1) / 2. + 1;
			this->levels.add(nxc, nyc, nzc);
			nx = nxc; ny = nyc; nz = nzc;
		}
		len_t abd_len_0 = nx * (ny + 1) + 2;
		if (params->periodic[0] || params->periodic[1] || params->periodic[2]) abd_len_0 = nx * ny * nz;
		this->ABD = grid_func::matrix(abd_len_0, nx * ny * nz);
		this->bbd = new real_t[this->ABD.len(1)];
	}
	template <relax_dir d> using plane_relax = kernels::plane_relax<stypes, d>;
	using rt = ml_settings::relax_type;
	// include/cedar/multilevel.h:145-160
	template <class sten> void setup_relax_level(level3<sten> & level)
	{
		const auto r = this->settings.relaxation;
		if (r == rt::point) this->kman->template setup<kernels::point_relax<stypes>>(level.A, level.SOR[0]);
		else if (!this->settings.planes()) log::error << "cdr3::solver: relaxation must be point or plane-xy / -xz / -yz / -xyz" << std::endl;
		if (r == rt::plane_xy || r == rt::plane_xyz) this->kman->template setup<plane_relax<relax_dir::xy>>(level.A);
		if (r == rt::plane_xz || r == rt::plane_xyz) this->kman->template setup<plane_relax<relax_dir::xz>>(level.A);
		if (r == rt::plane_yz || r == rt::plane_xyz) this->kman->template setup<plane_relax<relax_dir::yz>>(level.A);
	}
	// include/cedar/multilevel.h:165-222: plane-xyz runs xy, yz, xz on the way down and xz, yz, xy on the way up
	template <class sten> void smooth(level3<sten> & level, const stencil_op<sten> & A, grid_func & x, const grid_func & b, cycle::Dir dir)
	{
		const int n = dir == cycle::Dir::DOWN ? this->settings.nrelax_pre : this->settings.nrelax_post;
		const auto r = this->settings.relaxation;
		for (int i = 0; i < n; i++) {
			if (!this->settings.planes()) {
				this->kman->template run<kernels::point_relax<stypes>>(A, x, b, level.SOR[0], dir);
			} else if (r == rt::plane_xyz && dir == cycle::Dir::DOWN) {
				this->kman->template run<plane_relax<relax_dir::xy>>(A, x, b, dir);
				this->kman->template run<plane_relax<relax_dir::yz>>(A, x, b, dir);
				this->kman->template run<plane_relax<relax_dir::xz>>(A, x, b, dir);
			} else if (r == rt::plane_xyz) {
				this->kman->template run<plane_relax<relax_dir::xz>>(A, x, b, dir);
				this->kman->template run<plane_relax<relax_dir::yz>>(A, x, b, dir);
				this->kman->template run<plane_relax<relax_dir::xy>>(A, x, b, dir);
			} else if (r == rt::plane_xy) this->kman->template run<plane_relax<relax_dir::xy>>(A, x, b, dir);
			else if (r == rt::plane_xz) this->kman->template run<plane_relax<relax_dir::xz>>(A, x, b, dir);
			else this->kman->template run<plane_relax<relax_dir::yz>>(A, x, b, dir);
		}
	}
	void download(std::size_t l)
	{
		auto fetch = [&](const char * what, real_t * dst, std::size_t n) {
			if (cedar_amd_solver_get(this->h, (int)l, what, nullptr) == n) cedar_amd_solver_get(this->h, (int)l, what, dst);
		};
		if (l == 0) {
			fetch("SOR0", this->levels.fine.SOR[0].data(), this->levels.fine.SOR[0].size());
			return;
		}
		auto & L = this->levels.coarse[l - 1];
		fetch("A", L.A.data(), L.A.size());
		fetch("P", L.P.data(), L.P.size());
		fetch("SOR0", L.SOR[0].data(), L.SOR[0].size());
		if (l == 1) { L.P.fine_is_seven = std::is_same<fsten, seven_pt>::value; set_fine(L.P, this->levels.fine.A); }
		else { L.P.fine_is_seven = false; L.P.fine_op_xxvii = &this->levels.coarse[l - 2].A; }
	}
	void give_op(std::unique_ptr<stencil_op<fsten>> fop) { fop_ref = std::move(fop); }

protected:
	static void set_fine(prolong_op & P, stencil_op<seven_pt> & A) { P.fine_op_seven = &A; }
	static void set_fine(prolong_op & P, stencil_op<xxvii_pt> & A) { P.fine_op_xxvii = &A; }
	void init(stencil_op<fsten> & fop)
	{
		this->kman = build_kernel_manager(*this->conf);
		cedar_amd_settings st;
		cedar_amd_default_settings(&st);
		st.nrelax_pre = this->settings.nrelax_pre;
		st.nrelax_post = this->settings.nrelax_post;
		st.num_levels = this->settings.num_levels;
		st.max_iter = this->settings.maxiter;
		st.tol = this->settings.tol;
		st.min_coarse = this->settings.min_coarse;
		st.cycle = this->settings.cycle;
		st.relaxation = static_cast<int>(this->settings.relaxation); // point 0, plane-xy .. plane-xyz 4..7 as in cedar_amd.h
		{
			cedar_amd_settings pst = plane_settings(*this->kman->get_params()->plane_config);
			st.plane_relaxation = pst.relaxation; st.plane_nrelax_pre = pst.nrelax_pre; st.plane_nrelax_post = pst.nrelax_post;
			st.plane_max_iter = pst.max_iter; st.plane_min_coarse = pst.min_coarse; st.plane_tol = pst.tol;
		}
		BMG_get_bc(this->kman->get_params()->per_mask(), &st.ibc);
		this->h = cedar_amd_solver_create(3, fop.shape(0), fop.shape(1), fop.shape(2), stencil_ndirs<fsten>::value, fop.data(), 0, &st);
		if (!this->h)
			log::error << "cdr3::solver: the device-resident solver could not be created for these settings (reported above)" << std::endl;
	}
	std::unique_ptr<stencil_op<fsten>> fop_ref;
};
}}
#endif
