// cedar::cdr2 data types (Boost-free mirror of the reference's include/cedar/2d/{grid_func,stencil_op,relax_stencil,
// prolong_op,restrict_op,types}.h and include/cedar/grid_quantity.h).  Index convention of operator(): 0-based
// including the ghost cell (interior = 1..nx), first index fastest, one ghost layer -- the layout the
// BMG2_SymStd_* kernels take (include/cedar_amd.h), so data() goes to the library without conversion.
#ifndef CEDAR_2D_TYPES_H
#define CEDAR_2D_TYPES_H
#include <cmath>
#include <cedar/array.h>
#include <cedar/solver_types.h>
#include <cedar/2d/base_types.h>

namespace cedar { namespace cdr2 {

class grid_func : public array<real_t, 2> {
public:
	grid_func() {}
	grid_func(len_t nx, len_t ny, unsigned int nghosts = 1) : ng(nghosts) { this->reshape(nx + 2 * nghosts, ny + 2 * nghosts); }
	static grid_func zeros(len_t nx, len_t ny) { return grid_func(nx, ny); }
	static grid_func ones(len_t nx, len_t ny) { grid_func g(nx, ny); g.set(1.0); return g; }
	static grid_func zeros_like(const grid_func & o) { return grid_func(o.shape(0), o.shape(1), o.ng); }
	static grid_func ones_like(const grid_func & o) { grid_func g = zeros_like(o); g.set(1.0); return g; }
	len_t shape(int d) const { return this->len(d) - 2 * ng; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(ng, this->len(d) - ng); }
	range_t<len_t> grange(int d) const { return cedar::range<len_t>(0, this->len(d)); }
	// reference src/2d/grid_func.cc:118-134: signed value of the entry of largest magnitude
	real_t inf_norm() const
	{
		real_t cmax = 0;
		for (auto j : range(1)) for (auto i : range(0)) if (std::abs(cmax) < std::abs((*this)(i, j))) cmax = (*this)(i, j);
		return cmax;
	}
	// reference include/cedar/2d/grid_func.h:42-53 (sequential sum, i fastest)
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
private:
	unsigned int ng = 1;
};

template <class sten> class stencil_op : public array<real_t, 3> {
public:
	stencil_op() {}
	stencil_op(len_t nx, len_t ny) { this->reshape(nx + 2, ny + 2, static_cast<len_t>(stencil_ndirs<sten>::value)); }
	using array<real_t, 3>::operator();
	real_t & operator()(len_t i, len_t j, sten d) { return array<real_t, 3>::operator()(i, j, static_cast<len_t>(d)); }
	const real_t & operator()(len_t i, len_t j, sten d) const { return array<real_t, 3>::operator()(i, j, static_cast<len_t>(d)); }
	len_t shape(int d) const { return this->len(d) - 2; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(1, this->len(d) - 1); }
	range_t<len_t> grange(int d) const { return cedar::range<len_t>(0, this->len(d)); }
};

class relax_stencil : public array<real_t, 3> {
public:
	relax_stencil() {}
	relax_stencil(len_t nx, len_t ny) { this->reshape(nx + 2, ny + 2, 2u); }
};

enum class inter_dir { L = 0, R = 1, A = 2, B = 3, SW = 4, NW = 5, NE = 6, SE = 7, ndirs };
template <> struct stencil_ndirs<inter_dir> { static const int value = 8; };
// include/cedar/2d/prolong_op.h:14-28: the interpolation operator also remembers the fine operator
// (setup_interp stores it) because interp_add divides the residual by the fine diagonal
class prolong_op : public stencil_op<inter_dir> {
public:
	prolong_op() {}
	prolong_op(len_t nx, len_t ny) : stencil_op<inter_dir>(nx, ny) {}
	stencil_op<five_pt> * fine_op_five = nullptr;
	stencil_op<nine_pt> * fine_op_nine = nullptr;
	grid_func * residual = nullptr;
	bool fine_is_five = false;
};
class restrict_op {
public:
	restrict_op() : P(nullptr) {}
	restrict_op(prolong_op * P) : P(P) {}
	void associate(prolong_op * P) { this->P = P; }
	prolong_op & getP() { return *P; }
	const prolong_op & getP() const { return *P; }
private:
	prolong_op * P;
};

using stypes = solver_types<stencil_op, five_pt, nine_pt, grid_func, prolong_op, restrict_op, relax_stencil>;
}}
#endif
