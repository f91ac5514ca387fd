// cedar::cdr3 data types (Boost-free mirror of the reference's include/cedar/3d/{grid_func,stencil_op,relax_stencil,
// prolong_op,restrict_op,types}.h).  0-based indices including the ghost cell, first index fastest, one ghost layer:
// the layout the BMG3_SymStd_* kernels take (include/cedar_amd.h).
#ifndef CEDAR_3D_TYPES_H
#define CEDAR_3D_TYPES_H
#include <cmath>
#include <cedar/array.h>
#include <cedar/solver_types.h>
#include <cedar/3d/base_types.h>

namespace cedar { namespace cdr3 {

class grid_func : public array<real_t, 3> {
public:
	grid_func() {}
	grid_func(len_t nx, len_t ny, len_t nz, unsigned int nghosts = 1) : ng(nghosts)
	{ this->reshape(nx + 2 * nghosts, ny + 2 * nghosts, nz + 2 * nghosts); }
	// the band matrix of the coarsest solve is kept in a two-dimensional grid_func without ghosts in the reference
	// (include/cedar/3d/solver.h:118-122: grid_func(abd_len_0, n, 0)): here a (len0, len1, 1) array
	static grid_func matrix(len_t n0, len_t n1) { grid_func g; g.ng = 0; g.reshape(n0, n1, 1u); return g; }
	len_t shape(int d) const { return this->len(d) - 2 * ng; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(ng, this->len(d) - ng); }
	range_t<len_t> grange(int d) const { return cedar::range<len_t>(0, this->len(d)); }
	static grid_func zeros(len_t nx, len_t ny, len_t nz) { return grid_func(nx, ny, nz); }
	static grid_func zeros_like(const grid_func & o) { return grid_func(o.shape(0), o.shape(1), o.shape(2), o.ng); }
	static grid_func ones(len_t nx, len_t ny, len_t nz) { grid_func g(nx, ny, nz); g.set(1.0); return g; }
	static grid_func ones_like(const grid_func & o) { grid_func g = zeros_like(o); g.set(1.0); return g; }
	// src/3d/grid_func.cc: signed entry of largest magnitude
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
private:
	unsigned int ng = 1;
};

template <class sten> class stencil_op : public array<real_t, 4> {
public:
	stencil_op() {}
	stencil_op(len_t nx, len_t ny, len_t nz) { this->reshape(nx + 2, ny + 2, nz + 2, static_cast<len_t>(stencil_ndirs<sten>::value)); }
	using array<real_t, 4>::operator();
	real_t & operator()(len_t i, len_t j, len_t k, sten d) { return array<real_t, 4>::operator()(i, j, k, static_cast<len_t>(d)); }
	const real_t & operator()(len_t i, len_t j, len_t k, sten d) const { return array<real_t, 4>::operator()(i, j, k, static_cast<len_t>(d)); }
	len_t shape(int d) const { return this->len(d) - 2; }
	range_t<len_t> range(int d) const { return cedar::range<len_t>(1, this->len(d) - 1); }
	range_t<len_t> grange(int d) const { return cedar::range<len_t>(0, this->len(d)); }
};

// src/3d/relax_stencil.cc:8-9 allocates (nx+3)^3 x 2 in the reference; the kernels index it as (II,JJ,KK,2)
class relax_stencil : public array<real_t, 4> {
public:
	relax_stencil() {}
	relax_stencil(len_t nx, len_t ny, len_t nz) { this->reshape(nx + 2, ny + 2, nz + 2, 2u); }
};

enum class inter_dir { XYL = 0, XYR = 1, XYA = 2, XYB = 3, XZA = 4, XZB = 5, XYNE = 6, XYSE = 7, XYSW = 8, XYNW = 9, XZSW = 10,
                       XZNW = 11, XZNE = 12, XZSE = 13, YZSW = 14, YZNW = 15, YZNE = 16, YZSE = 17, BSW = 18, BNW = 19, BNE = 20,
                       BSE = 21, TSW = 22, TNW = 23, TNE = 24, TSE = 25, ndirs };
template <> struct stencil_ndirs<inter_dir> { static const int value = 26; };
// include/cedar/3d/prolong_op.h:20-31
class prolong_op : public stencil_op<inter_dir> {
public:
	prolong_op() {}
	prolong_op(len_t nx, len_t ny, len_t nz) : stencil_op<inter_dir>(nx, ny, nz) {}
	stencil_op<seven_pt> * fine_op_seven = nullptr;
	stencil_op<xxvii_pt> * fine_op_xxvii = nullptr;
	grid_func * residual = nullptr;
	bool fine_is_seven = false;
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

using stypes = solver_types<stencil_op, seven_pt, xxvii_pt, grid_func, prolong_op, restrict_op, relax_stencil>;
}}
#endif
