// cedar::cdr2 -- 2D data types, gallery, kernel bindings and solver over libcedar_amd.so.
// Mirrors (Boost-free) the reference's include/cedar/2d/{base_types,grid_func,stencil_op,
// relax_stencil,prolong_op,gallery,relax,residual,interp,restrict,coarsen,solve_cg,solver}.h.
// Index convention of operator(): 0-based including the ghost cell, as in the reference
// (interior = 1..nx), so user code such as examples/basic-2d-ser/poisson.cc ports unchanged.
#ifndef CEDAR_2D_SOLVER_H
#define CEDAR_2D_SOLVER_H
#include <cmath>
#include <functional>
#include <cedar/array.h>
#include <cedar/kernel_manager.h>
extern "C" {
#include <cedar_amd.h>
}

namespace cedar { namespace cdr2 {

enum class five_pt { c = 0, w = 1, s = 2, ndirs = 3 };
enum class nine_pt { c = 0, w = 1, s = 2, sw = 3, nw = 4, ndirs = 5 };
template <class sten> struct stencil_ndirs { static const int value = static_cast<int>(sten::ndirs); };

template <class Derived, unsigned short ND> class grid_quantity : public array<real_t, ND> {
public:
	using array<real_t, ND>::array;
	len_t shape(int d) const { return this->len(d) - 2; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(1, this->len(d) - 1); }
	range_t<len_t> grange(int d) const { return cedar::range<len_t>(0, this->len(d)); }
};

class grid_func : public grid_quantity<grid_func, 2> {
public:
	grid_func() {}
	grid_func(len_t nx, len_t ny) { this->reshape(nx + 2, ny + 2); }
	static grid_func zeros(len_t nx, len_t ny) { return grid_func(nx, ny); }
	static grid_func ones(len_t nx, len_t ny) { grid_func g(nx, ny); g.set(1.0); return g; }
	static grid_func zeros_like(const grid_func & o) { return grid_func(o.shape(0), o.shape(1)); }
	static grid_func ones_like(const grid_func & o) { return ones(o.shape(0), o.shape(1)); }
	// reference src/2d/grid_func.cc:118-134: signed value of the entry of largest magnitude
	real_t inf_norm() const
	{
		real_t cmax = 0;
		for (auto j : range(1)) for (auto i : range(0)) if (std::abs(cmax) < std::abs((*this)(i, j))) cmax = (*this)(i, j);
		return cmax;
	}
	// reference include/cedar/2d/grid_func.h:42-53 (sequential sum on the host copy)
	template <int p> real_t lp_norm() const
	{
		real_t r = 0;
		for (auto j : range(1)) for (auto i : range(0)) r += std::pow((*this)(i, j), p);
		return std::pow(r, 1. / p);
	}
	grid_func & operator-=(const grid_func & o)
	{
		for (auto j : range(1)) for (auto i : range(0)) (*this)(i, j) -= o(i, j);
		return *this;
	}
	friend grid_func operator-(grid_func a, const grid_func & b) { return a -= b; }
};

template <class sten> class stencil_op : public array<real_t, 3> {
public:
	stencil_op() {}
	stencil_op(len_t nx, len_t ny) { this->reshape(nx + 2, ny + 2, static_cast<len_t>(stencil_ndirs<sten>::value)); }
	real_t & operator()(len_t i, len_t j, sten d) { return array<real_t, 3>::operator()(i, j, static_cast<len_t>(d)); }
	const real_t & operator()(len_t i, len_t j, sten d) const { return array<real_t, 3>::operator()(i, j, static_cast<len_t>(d)); }
	len_t shape(int d) const { return this->len(d) - 2; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(1, this->len(d) - 1); }
};
class relax_stencil : public array<real_t, 3> {
public:
	relax_stencil() {}
	relax_stencil(len_t nx, len_t ny) { this->reshape(nx + 2, ny + 2, 2u); }
};
enum class inter_dir { L = 0, R, A, B, SW, NW, NE, SE, ndirs };
class prolong_op : public array<real_t, 3> {
public:
	prolong_op() {}
	prolong_op(len_t nx, len_t ny) { this->reshape(nx + 2, ny + 2, 8u); }
	real_t * fine_op = nullptr; int fine_nst = 0; // reference keeps fine_op_five / fine_op_nine pointers
};
struct restrict_op { prolong_op * P = nullptr; void associate(prolong_op * p) { P = p; } prolong_op & getP() { return *P; } };

// ---------------------------------------------------------------- gallery (src/2d/gallery.cc:7-113)
namespace gallery {
inline stencil_op<five_pt> diag_diffusion(len_t nx, len_t ny, real_t dx, real_t dy)
{
	stencil_op<five_pt> so(nx, ny);
	real_t hx = 1.0 / (so.len(0) - 1), hy = 1.0 / (so.len(1) - 1);
	real_t xh = hy / hx, yh = hx / hy;
	for (len_t j = 2; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, five_pt::s) = dy * yh;
	for (len_t j = 1; j <= ny; j++) for (len_t i = 2; i <= nx; i++) so(i, j, five_pt::w) = dx * xh;
	for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, five_pt::c) = 2 * dx * xh + 2 * dy * yh;
	return so;
}
inline stencil_op<five_pt> poisson(len_t nx, len_t ny) { return diag_diffusion(nx, ny, 1.0, 1.0); }
inline stencil_op<nine_pt> fe(len_t nx, len_t ny)
{
	stencil_op<nine_pt> so(nx, ny);
	for (len_t j = 2; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, nine_pt::s) = 1.0;
	for (len_t j = 1; j <= ny; j++) for (len_t i = 2; i <= nx; i++) so(i, j, nine_pt::w) = 1.0;
	for (len_t j = 2; j <= ny; j++) for (len_t i = 2; i <= nx; i++) { so(i, j, nine_pt::sw) = 1.0; so(i, j, nine_pt::nw) = 1.0; }
	for (len_t j = 1; j <= ny; j++) for (len_t i = 1; i <= nx; i++) so(i, j, nine_pt::c) = 8.0;
	return so;
}
}

// ---------------------------------------------------------------- abstract kernels (include/cedar/kernels/*.h)
namespace kernels {
struct point_relax : kernel_base {
	static std::string name() { return "point relaxation"; }
	virtual void setup(real_t * so, int nst, relax_stencil & sor) = 0;
	virtual void run(real_t * so, int nst, grid_func & x, const grid_func & b, const relax_stencil & sor, cycle::Dir d) = 0;
};
template <relax_dir rdir> struct line_relax : kernel_base {
	static std::string name() { return "line relaxation"; }
	virtual void setup(real_t * so, int nst, relax_stencil & sor) = 0;
	virtual void run(real_t * so, int nst, grid_func & x, const grid_func & b, const relax_stencil & sor, grid_func & res, cycle::Dir d) = 0;
};
struct residual : kernel_base {
	static std::string name() { return "residual"; }
	virtual void run(real_t * so, int nst, const grid_func & x, const grid_func & b, grid_func & r) = 0;
};
struct restriction : kernel_base {
	static std::string name() { return "restriction"; }
	virtual void run(const restrict_op & R, const grid_func & fine, grid_func & coarse) = 0;
};
struct interp_add : kernel_base {
	static std::string name() { return "interpolate and add"; }
	virtual void run(const prolong_op & P, const grid_func & coarse, const grid_func & residual, grid_func & fine) = 0;
};
struct setup_interp : kernel_base {
	static std::string name() { return "setup interpolation"; }
	virtual void run(real_t * fop, int nst, len_t iif, len_t jjf, prolong_op & P) = 0;
};
struct coarsen_op : kernel_base {
	static std::string name() { return "coarsen operator"; }
	virtual void run(const prolong_op & P, real_t * fop, int nst, len_t iif, len_t jjf, stencil_op<nine_pt> & cop) = 0;
};
struct solve_cg : kernel_base {
	static std::string name() { return "coarse grid solve"; }
	virtual void setup(stencil_op<nine_pt> & so, array<real_t, 2> & ABD) = 0;
	virtual void run(grid_func & x, const grid_func & b, const array<real_t, 2> & ABD, real_t * bbd) = 0;
};
}

// ---------------------------------------------------------------- "hip" implementations: argument
// marshalling exactly as the reference's binding classes (include/cedar/2d/relax.h:59-102,
// residual.h:36-60, src/2d/interp.cc:20-106, src/2d/restrict.cc:15-30, coarsen.h:34-61, solve_cg.h:32-55)
namespace hip {
template <class T> T * mut(const T * p) { return const_cast<T *>(p); }
struct rbgs : kernels::point_relax {
	void setup(real_t * so, int nst, relax_stencil & sor) override
	{ BMG2_SymStd_SETUP_recip(so, sor.data(), sor.len(0), sor.len(1), nst, 2); }
	void run(real_t * so, int nst, grid_func & x, const grid_func & b, const relax_stencil & sor, cycle::Dir d) override
	{
		int ibc; BMG_get_bc(params->per_mask(), &ibc);
		BMG2_SymStd_relax_GS(1, so, mut(b.data()), x.data(), mut(sor.data()), x.len(0), x.len(1), 1, nst == 3, nst, 2,
		                     1 /*BMG_RELAX_SYM*/, d == cycle::Dir::UP ? 1 : 0, ibc);
	}
};
template <relax_dir rdir> struct lines : kernels::line_relax<rdir> {
	void setup(real_t * so, int nst, relax_stencil & sor) override
	{
		int jpn; BMG_get_bc(this->params->per_mask(), &jpn);
		if (rdir == relax_dir::x) BMG2_SymStd_SETUP_lines_x(so, sor.data(), sor.len(0), sor.len(1), nst, jpn);
		else BMG2_SymStd_SETUP_lines_y(so, sor.data(), sor.len(0), sor.len(1), nst, jpn);
	}
	void run(real_t * so, int nst, grid_func & x, const grid_func & b, const relax_stencil & sor, grid_func & res, cycle::Dir d) override
	{
		int ibc; BMG_get_bc(this->params->per_mask(), &ibc);
		auto f = rdir == relax_dir::x ? BMG2_SymStd_relax_lines_x : BMG2_SymStd_relax_lines_y;
		f(1, so, mut(b.data()), x.data(), mut(sor.data()), res.data(), x.len(0), x.len(1), 1, nst == 3, nst, 1,
		  d == cycle::Dir::UP ? 1 : 0, ibc);
	}
};
struct residual_hip : kernels::residual {
	void run(real_t * so, int nst, const grid_func & x, const grid_func & b, grid_func & r) override
	{
		int k = 0, kf = 0, ifd = nst == 3, ibc, z = 0; len_t ii = r.len(0), jj = r.len(1);
		BMG_get_bc(params->per_mask(), &ibc);
		BMG2_SymStd_residual(&k, so, mut(b.data()), mut(x.data()), r.data(), &ii, &jj, &kf, &ifd, &nst, &ibc, &z, &z, &z);
	}
};
struct restrict_hip : kernels::restriction {
	void run(const restrict_op & R, const grid_func & fine, grid_func & coarse) override
	{
		int ibc; BMG_get_bc(params->per_mask(), &ibc);
		BMG2_SymStd_restrict(mut(fine.data()), coarse.data(), R.P->data(), fine.len(0), fine.len(1), coarse.len(0), coarse.len(1), ibc);
	}
};
struct interp_hip : kernels::interp_add {
	void run(const prolong_op & P, const grid_func & coarse, const grid_func & residual, grid_func & fine) override
	{
		int ibc; BMG_get_bc(params->per_mask(), &ibc);
		BMG2_SymStd_interp_add(fine.data(), mut(coarse.data()), mut(residual.data()), P.fine_op, mut(P.data()),
		                       coarse.len(0), coarse.len(1), fine.len(0), fine.len(1), P.fine_nst, ibc);
	}
};
struct setup_interp_hip : kernels::setup_interp {
	void run(real_t * fop, int nst, len_t iif, len_t jjf, prolong_op & P) override
	{
		int jpn; BMG_get_bc(params->per_mask(), &jpn);
		P.fine_op = fop; P.fine_nst = nst;
		BMG2_SymStd_SETUP_interp_OI(fop, nullptr, P.data(), iif, jjf, P.len(0), P.len(1), nst == 3, nst, jpn, 0);
	}
};
struct galerkin : kernels::coarsen_op {
	void run(const prolong_op & P, real_t * fop, int nst, len_t iif, len_t jjf, stencil_op<nine_pt> & cop) override
	{
		int ipn; BMG_get_bc(params->per_mask(), &ipn);
		BMG2_SymStd_SETUP_ITLI_ex(fop, cop.data(), mut(P.data()), iif, jjf, cop.len(0), cop.len(1), nst == 3, nst, ipn);
	}
};
struct solve_cg_hip : kernels::solve_cg {
	void setup(stencil_op<nine_pt> & so, array<real_t, 2> & ABD) override
	{
		len_t nx = so.len(0), ny = so.len(1), n1 = ABD.len(0), n2 = ABD.len(1); int nst = 5, ibc;
		BMG_get_bc(params->per_mask(), &ibc);
		BMG2_SymStd_SETUP_cg_LU(so.data(), &nx, &ny, &nst, ABD.data(), &n1, &n2, &ibc);
	}
	void run(grid_func & x, const grid_func & b, const array<real_t, 2> & ABD, real_t * bbd) override
	{
		int ibc; BMG_get_bc(params->per_mask(), &ibc);
		BMG2_SymStd_SOLVE_cg(x.data(), mut(b.data()), x.len(0), x.len(1), mut(ABD.data()), bbd, ABD.len(0), ABD.len(1), ibc);
	}
};
}

// reference src/2d/kernel_manager.cc:16-42 registers "system"; here: "hip"
inline std::shared_ptr<kernel_manager> build_kernel_manager(config & conf)
{
	auto km = std::make_shared<kernel_manager>(conf);
	km->add<kernels::point_relax, hip::rbgs>("hip");
	km->add<kernels::line_relax<relax_dir::x>, hip::lines<relax_dir::x>>("hip");
	km->add<kernels::line_relax<relax_dir::y>, hip::lines<relax_dir::y>>("hip");
	km->add<kernels::residual, hip::residual_hip>("hip");
	km->add<kernels::restriction, hip::restrict_hip>("hip");
	km->add<kernels::interp_add, hip::interp_hip>("hip");
	km->add<kernels::setup_interp, hip::setup_interp_hip>("hip");
	km->add<kernels::coarsen_op, hip::galerkin>("hip");
	km->add<kernels::solve_cg, hip::solve_cg_hip>("hip");
	return km;
}

// ---------------------------------------------------------------- solver (include/cedar/2d/solver.h:39-122,
// include/cedar/multilevel.h:80-92,268-308).  The hierarchy is device resident behind the handle API;
// get_kernels() exposes the per-kernel "hip" bindings for callers that drive kernels themselves.
template <class fsten> class solver {
public:
	explicit solver(stencil_op<fsten> & fop) : solver(fop, std::make_shared<config>("config.json")) {}
	solver(stencil_op<fsten> & fop, std::shared_ptr<config> cfg) : conf(cfg), fop_(fop)
	{
		settings.init(*conf);
		kman = build_kernel_manager(*conf);
		cedar_amd_settings st; cedar_amd_default_settings(&st);
		st.relaxation = static_cast<int>(settings.relaxation);
		st.nrelax_pre = settings.nrelax_pre; st.nrelax_post = settings.nrelax_post;
		st.num_levels = settings.num_levels; st.max_iter = settings.maxiter; st.tol = settings.tol;
		st.min_coarse = settings.min_coarse;
		st.cycle = settings.cycle;
		BMG_get_bc(kman->get_params()->per_mask(), &st.ibc); // grid.periodic -> boundary code, as every reference binding does
		h = cedar_amd_solver_create(2, fop.shape(0), fop.shape(1), 1, stencil_ndirs<fsten>::value, fop.data(), 0, &st);
	}
	~solver() { cedar_amd_solver_destroy(h); }
	solver(const solver &) = delete;
	grid_func solve(const grid_func & b) { grid_func x = grid_func::zeros_like(b); solve(b, x); return x; }
	void solve(const grid_func & b, grid_func & x)
	{
		std::vector<real_t> rel(settings.maxiter + 1);
		int n = cedar_amd_solver_solve(h, b.data(), x.data(), rel.data());
		log::info << "Initial residual l2 norm: " << rel[0] << std::endl;
		for (int i = 0; i < n; i++) log::status << "Iteration " << i << " relative l2 norm: " << rel[i + 1] << std::endl;
		history.assign(rel.begin(), rel.begin() + n + 1);
	}
	void vcycle(grid_func & x, const grid_func & b) { cedar_amd_solver_vcycle(h, x.data(), b.data()); }
	std::size_t nlevels() { return cedar_amd_solver_nlevels(h); }
	std::shared_ptr<kernel_manager> get_kernels() { return kman; }
	config & get_config() { return *conf; }
	void give_op(std::unique_ptr<stencil_op<fsten>> fop) { fop_ref = std::move(fop); }
	std::vector<real_t> history; // [||r0||, rel_1, ...] of the last solve
protected:
	std::shared_ptr<config> conf;
	ml_settings settings;
	std::shared_ptr<kernel_manager> kman;
	stencil_op<fsten> & fop_;
	std::unique_ptr<stencil_op<fsten>> fop_ref;
	cedar_amd_solver * h = nullptr;
};
}}
#endif
