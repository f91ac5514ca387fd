// cedar::cdr3 -- 3D data types, gallery, kernel bindings and solver over libcedar_amd.so.
// Mirrors the reference's include/cedar/3d/{base_types,grid_func,stencil_op,relax_stencil,prolong_op,
// gallery,relax,residual,interp,restrict,coarsen,solve_cg,solver}.h (Boost-free).
#ifndef CEDAR_3D_SOLVER_H
#define CEDAR_3D_SOLVER_H
#include <cmath>
#include <cedar/array.h>
#include <cedar/kernel_manager.h>
extern "C" {
#include <cedar_amd.h>
}

namespace cedar { namespace cdr3 {

enum class seven_pt { p = 0, pw = 1, ps = 2, b = 3, ndirs = 4 };
enum class xxvii_pt { p = 0, pw, ps, b, psw, pnw, bw, bnw, bn, bne, be, bse, bs, bsw, ndirs };
template <class sten> struct stencil_ndirs { static const int value = static_cast<int>(sten::ndirs); };

class grid_func : public array<real_t, 3> {
public:
	grid_func() {}
	grid_func(len_t nx, len_t ny, len_t nz) { this->reshape(nx + 2, ny + 2, nz + 2); }
	len_t shape(int d) const { return this->len(d) - 2; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(1, this->len(d) - 1); }
	range_t<len_t> grange(int d) const { return cedar::range<len_t>(0, this->len(d)); }
	static grid_func zeros_like(const grid_func & o) { return grid_func(o.shape(0), o.shape(1), o.shape(2)); }
	static grid_func ones(len_t nx, len_t ny, len_t nz) { grid_func g(nx, ny, nz); g.set(1.0); return g; }
	real_t inf_norm() const
	{
		real_t cmax = 0;
		for (auto k : range(2)) for (auto j : range(1)) for (auto i : range(0))
			if (std::abs(cmax) < std::abs((*this)(i, j, k))) cmax = (*this)(i, j, k);
		return cmax;
	}
	template <int p> real_t lp_norm() const
	{
		real_t r = 0;
		for (auto k : range(2)) for (auto j : range(1)) for (auto i : range(0)) r += std::pow((*this)(i, j, k), p);
		return std::pow(r, 1. / p);
	}
	grid_func & operator-=(const grid_func & o)
	{
		for (auto k : range(2)) for (auto j : range(1)) for (auto i : range(0)) (*this)(i, j, k) -= o(i, j, k);
		return *this;
	}
	friend grid_func operator-(grid_func a, const grid_func & b) { return a -= b; }
};

template <class sten> class stencil_op : public array<real_t, 4> {
public:
	stencil_op() {}
	stencil_op(len_t nx, len_t ny, len_t nz) { this->reshape(nx + 2, ny + 2, nz + 2, static_cast<len_t>(stencil_ndirs<sten>::value)); }
	real_t & operator()(len_t i, len_t j, len_t k, sten d) { return array<real_t, 4>::operator()(i, j, k, static_cast<len_t>(d)); }
	const real_t & operator()(len_t i, len_t j, len_t k, sten d) const { return array<real_t, 4>::operator()(i, j, k, static_cast<len_t>(d)); }
	len_t shape(int d) const { return this->len(d) - 2; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(1, this->len(d) - 1); }
};
class relax_stencil : public array<real_t, 4> {
public:
	relax_stencil() {}
	relax_stencil(len_t nx, len_t ny, len_t nz) { this->reshape(nx + 2, ny + 2, nz + 2, 2u); }
};
class prolong_op : public array<real_t, 4> {
public:
	prolong_op() {}
	prolong_op(len_t nx, len_t ny, len_t nz) { this->reshape(nx + 2, ny + 2, nz + 2, 26u); }
	real_t * fine_op = nullptr; int fine_nst = 0;
};

// ---------------------------------------------------------------- gallery (src/3d/gallery.cc:7-190)
namespace gallery {
inline stencil_op<seven_pt> diag_diffusion(len_t nx, len_t ny, len_t nz, real_t dx, real_t dy, real_t dz)
{
	stencil_op<seven_pt> so(nx, ny, nz);
	real_t hx = 1.0 / (so.len(0) - 1), hy = 1.0 / (so.len(1) - 1), hz = 1.0 / (so.len(2) - 1);
	real_t xh = hy * hz / hx, yh = hx * hz / hy, zh = hx * hy / hz;
	for (len_t k = 1; k <= nz; k++) for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) {
		if (j >= 2) so(i, j, k, seven_pt::ps) = dy * yh;
		if (i >= 2) so(i, j, k, seven_pt::pw) = dx * xh;
		if (k >= 2) so(i, j, k, seven_pt::b) = dz * zh;
		so(i, j, k, seven_pt::p) = 2.0 * dx * xh + 2.0 * dy * yh + 2.0 * dz * zh;
	}
	return so;
}
inline stencil_op<seven_pt> poisson(len_t nx, len_t ny, len_t nz) { return diag_diffusion(nx, ny, nz, 1.0, 1.0, 1.0); }
inline stencil_op<xxvii_pt> fe(len_t nx, len_t ny, len_t nz)
{
	stencil_op<xxvii_pt> so(nx, ny, nz);
	using X = xxvii_pt;
	for (len_t k = 1; k <= nz; k++) for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) {
		const bool I = i >= 2, J = j >= 2, K = k >= 2;
		if (I) so(i, j, k, X::pw) = 1.0;
		if (J) so(i, j, k, X::ps) = 1.0;
		if (K) so(i, j, k, X::b) = 1.0;
		if (I && J) { so(i, j, k, X::pnw) = 1.0; so(i, j, k, X::psw) = 1.0; }
		if (I && K) { so(i, j, k, X::bw) = 1.0; so(i, j, k, X::be) = 1.0; }
		if (J && K) { so(i, j, k, X::bn) = 1.0; so(i, j, k, X::bs) = 1.0; }
		if (I && J && K) { so(i, j, k, X::bnw) = 1.0; so(i, j, k, X::bne) = 1.0; so(i, j, k, X::bse) = 1.0; so(i, j, k, X::bsw) = 1.0; }
		so(i, j, k, X::p) = 26;
	}
	return so;
}
}

// ---------------------------------------------------------------- abstract kernels + "hip" bindings
// (include/cedar/3d/relax.h:44-100, residual.h:36-60, src/3d/interp.cc:15-45, src/3d/restrict.cc,
//  include/cedar/3d/coarsen.h:36-60, interp.h:36-80, solve_cg.h)
namespace kernels {
struct point_relax : kernel_base {
	static std::string name() { return "point relaxation"; }
	virtual void setup(real_t * so, int nst, relax_stencil & sor) = 0;
	virtual void run(real_t * so, int nst, grid_func & x, const grid_func & b, const relax_stencil & sor, cycle::Dir d) = 0;
};
struct residual : kernel_base {
	static std::string name() { return "residual"; }
	virtual void run(real_t * so, int nst, const grid_func & x, const grid_func & b, grid_func & r) = 0;
};
struct restriction : kernel_base {
	static std::string name() { return "restriction"; }
	virtual void run(const prolong_op & P, const grid_func & fine, grid_func & coarse) = 0;
};
struct interp_add : kernel_base {
	static std::string name() { return "interpolate and add"; }
	virtual void run(const prolong_op & P, const grid_func & coarse, const grid_func & residual, grid_func & fine) = 0;
};
struct setup_interp : kernel_base {
	static std::string name() { return "setup interpolation"; }
	virtual void run(real_t * fop, int nst, len_t iif, len_t jjf, len_t kkf, prolong_op & P) = 0;
};
struct coarsen_op : kernel_base {
	static std::string name() { return "coarsen operator"; }
	virtual void run(const prolong_op & P, real_t * fop, int nst, len_t iif, len_t jjf, len_t kkf, stencil_op<xxvii_pt> & cop) = 0;
};
}
namespace hip {
template <class T> T * mut(const T * p) { return const_cast<T *>(p); }
struct rbgs : kernels::point_relax {
	void setup(real_t * so, int nst, relax_stencil & sor) override
	{ BMG3_SymStd_SETUP_recip(so, sor.data(), sor.len(0), sor.len(1), sor.len(2), nst, 2); }
	void run(real_t * so, int nst, grid_func & x, const grid_func & b, const relax_stencil & sor, cycle::Dir d) override
	{
		int jpn; BMG_get_bc(params->per_mask(), &jpn);
		BMG3_SymStd_relax_GS(1, so, mut(b.data()), x.data(), mut(sor.data()), x.len(0), x.len(1), x.len(2),
		                     nst == 4, nst, 2, 1, d == cycle::Dir::UP ? 1 : 0, jpn);
	}
};
struct residual_hip : kernels::residual {
	void run(real_t * so, int nst, const grid_func & x, const grid_func & b, grid_func & r) override
	{ BMG3_SymStd_residual(1, 1, nst == 4, mut(x.data()), mut(b.data()), so, r.data(), r.len(0), r.len(1), r.len(2), nst); }
};
struct restrict_hip : kernels::restriction {
	void run(const prolong_op & P, const grid_func & fine, grid_func & coarse) override
	{
		int ibc; BMG_get_bc(params->per_mask(), &ibc);
		BMG3_SymStd_restrict(mut(fine.data()), coarse.data(), mut(P.data()), fine.len(0), fine.len(1), fine.len(2),
		                     coarse.len(0), coarse.len(1), coarse.len(2), ibc);
	}
};
struct interp_hip : kernels::interp_add {
	void run(const prolong_op & P, const grid_func & coarse, const grid_func & residual, grid_func & fine) override
	{
		int ibc; BMG_get_bc(params->per_mask(), &ibc);
		BMG3_SymStd_interp_add(fine.data(), mut(coarse.data()), P.fine_op, mut(residual.data()), mut(P.data()),
		                       coarse.len(0), coarse.len(1), coarse.len(2), fine.len(0), fine.len(1), fine.len(2), P.fine_nst, ibc);
	}
};
struct setup_interp_hip : kernels::setup_interp {
	void run(real_t * fop, int nst, len_t iif, len_t jjf, len_t kkf, prolong_op & P) override
	{
		int jpn; BMG_get_bc(params->per_mask(), &jpn);
		P.fine_op = fop; P.fine_nst = nst;
		BMG3_SymStd_SETUP_interp_OI(fop, nullptr, P.data(), iif, jjf, kkf, P.len(0), P.len(1), P.len(2), nst == 4, nst, 1, jpn, nullptr);
	}
};
struct galerkin : kernels::coarsen_op {
	void run(const prolong_op & P, real_t * fop, int nst, len_t iif, len_t jjf, len_t kkf, stencil_op<xxvii_pt> & cop) override
	{
		int ipn; BMG_get_bc(params->per_mask(), &ipn);
		auto f = nst == 4 ? BMG3_SymStd_SETUP_ITLI07_ex : BMG3_SymStd_SETUP_ITLI27_ex;
		f(fop, cop.data(), mut(P.data()), iif, jjf, kkf, cop.len(0), cop.len(1), cop.len(2), ipn);
	}
};
}
inline std::shared_ptr<kernel_manager> build_kernel_manager(config & conf)
{
	auto km = std::make_shared<kernel_manager>(conf);
	km->add<kernels::point_relax, hip::rbgs>("hip");
	km->add<kernels::residual, hip::residual_hip>("hip");
	km->add<kernels::restriction, hip::restrict_hip>("hip");
	km->add<kernels::interp_add, hip::interp_hip>("hip");
	km->add<kernels::setup_interp, hip::setup_interp_hip>("hip");
	km->add<kernels::coarsen_op, hip::galerkin>("hip");
	return km;
}

// ---------------------------------------------------------------- solver (include/cedar/3d/solver.h:39-130)
template <class fsten> class solver {
public:
	explicit solver(stencil_op<fsten> & fop) : solver(fop, std::make_shared<config>("config.json")) {}
	solver(stencil_op<fsten> & fop, std::shared_ptr<config> cfg) : conf(cfg)
	{
		settings.init(*conf);
		kman = build_kernel_manager(*conf);
		cedar_amd_settings st; cedar_amd_default_settings(&st);
		st.nrelax_pre = settings.nrelax_pre; st.nrelax_post = settings.nrelax_post;
		st.num_levels = settings.num_levels; st.max_iter = settings.maxiter; st.tol = settings.tol;
		st.min_coarse = settings.min_coarse;
		st.cycle = settings.cycle;
		if (settings.relaxation != ml_settings::relax_type::point)
			log::error << "3D: only point relaxation is implemented on the GPU path" << std::endl;
		if (kman->get_params()->per_mask() != 0)
			log::error << "3D: periodic boundaries are not implemented on the GPU path (2D only); solving the Dirichlet problem" << std::endl;
		h = cedar_amd_solver_create(3, fop.shape(0), fop.shape(1), fop.shape(2), stencil_ndirs<fsten>::value, fop.data(), 0, &st);
	}
	~solver() { cedar_amd_solver_destroy(h); }
	solver(const solver &) = delete;
	grid_func solve(const grid_func & b) { grid_func x = grid_func::zeros_like(b); solve(b, x); return x; }
	void solve(const grid_func & b, grid_func & x)
	{
		std::vector<real_t> rel(settings.maxiter + 1);
		int n = cedar_amd_solver_solve(h, b.data(), x.data(), rel.data());
		for (int i = 0; i < n; i++) log::status << "Iteration " << i << " relative l2 norm: " << rel[i + 1] << std::endl;
		history.assign(rel.begin(), rel.begin() + n + 1);
	}
	void vcycle(grid_func & x, const grid_func & b) { cedar_amd_solver_vcycle(h, x.data(), b.data()); }
	std::size_t nlevels() { return cedar_amd_solver_nlevels(h); }
	std::shared_ptr<kernel_manager> get_kernels() { return kman; }
	config & get_config() { return *conf; }
	std::vector<real_t> history;
protected:
	std::shared_ptr<config> conf;
	ml_settings settings;
	std::shared_ptr<kernel_manager> kman;
	cedar_amd_solver * h = nullptr;
};
}}
#endif
