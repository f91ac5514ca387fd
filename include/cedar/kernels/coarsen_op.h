// Abstract Galerkin coarsening cop = P^T fop P (reference include/cedar/kernels/coarsen_op.h:5-46).
#ifndef CEDAR_COARSEN_OP_H
#define CEDAR_COARSEN_OP_H
#include <cedar/kernel.h>

namespace cedar { namespace kernels {
template <class solver_types> class coarsen_op : public kernel<solver_types> {
public:
	template <class sten> using stencil_op = typename kernel<solver_types>::template stencil_op<sten>;
	using comp_sten = typename kernel<solver_types>::comp_sten;
	using full_sten = typename kernel<solver_types>::full_sten;
	using grid_func = typename kernel<solver_types>::grid_func;
	using prolong_op = typename kernel<solver_types>::prolong_op;
	using run_comp_t = std::function<void(const prolong_op &, const stencil_op<comp_sten> &, stencil_op<full_sten> &)>;
	using run_full_t = std::function<void(const prolong_op &, const stencil_op<full_sten> &, stencil_op<full_sten> &)>;

	coarsen_op() {}
	coarsen_op(run_comp_t rcomp, run_full_t rfull) : run_comp(rcomp), run_full(rfull) {}
	const static std::string name() { return "coarsen operator"; }
	virtual void run(const prolong_op & P, const stencil_op<comp_sten> & fop, stencil_op<full_sten> & cop)
	{
		if (run_comp) run_comp(P, fop, cop);
		else log::error << name() << ": routine not provided" << std::endl;
	}
	virtual void run(const prolong_op & P, const stencil_op<full_sten> & fop, stencil_op<full_sten> & cop)
	{
		if (run_full) run_full(P, fop, cop);
		else log::error << name() << ": routine not provided" << std::endl;
	}
protected:
	run_comp_t run_comp;
	run_full_t run_full;
};
}}
#endif
