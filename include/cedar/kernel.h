// Base of every kernel implementation (reference include/cedar/kernel.h:10-37): the type aliases of the solver's
// type list, the shared kernel_params and the halo exchanger the MPI flavour hands to its kernels.
#ifndef CEDAR_KERNEL_H
#define CEDAR_KERNEL_H
#include <functional>
#include <memory>
#include <string>
#include <cedar/config.h>
#include <cedar/solver_types.h>

namespace cedar {
// reference include/cedar/halo_exchanger_base.h: what a kernel may call back between colours.  Single-rank solvers
// never set one; the multi-GPU path exchanges halos inside the library (include/cedar_amd.h section 3).
class halo_exchanger_base {
public:
	virtual ~halo_exchanger_base() {}
	virtual void exchange_func(int k, real_t * gf) = 0;
	virtual void exchange_sten(int k, real_t * so) = 0;
};

// what the registry stores: every kernel<...> derives from it
struct kernel_base {
	virtual ~kernel_base() {}
	void add_params(std::shared_ptr<kernel_params> p) { this->params = p; }
	void add_halo(halo_exchanger_base * h) { this->halof = h; }
protected:
	std::shared_ptr<kernel_params> params;
	halo_exchanger_base * halof = nullptr;
};

template <class solver_types> class kernel : public kernel_base {
public:
	template <class sten> using stencil_op = typename solver_types::template stencil_op<sten>;
	using comp_sten = typename solver_types::comp_sten;
	using full_sten = typename solver_types::full_sten;
	using grid_func = typename solver_types::grid_func;
	using prolong_op = typename solver_types::prolong_op;
	using restrict_op = typename solver_types::restrict_op;
	using relax_stencil = typename solver_types::relax_stencil;
};
}
#endif
