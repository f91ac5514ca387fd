// Per-level data of a multilevel solve (reference include/cedar/level.h:14-41) and the container that holds the
// fine level (caller's operator, stencil fsten) and the coarse levels (full stencil)
// (reference include/cedar/2d/level_container.h, include/cedar/3d/level_container.h).
#ifndef CEDAR_LEVEL_H
#define CEDAR_LEVEL_H
#include <array>
#include <deque>
#include <functional>
#include <type_traits>
#include <cedar/types.h>

namespace cedar {
template <class sten, class solver_types> struct level {
	template <class stencil> using stencil_op = typename solver_types::template stencil_op<stencil>;
	using grid_func = typename solver_types::grid_func;
	using prolong_op = typename solver_types::prolong_op;
	using restrict_op = typename solver_types::restrict_op;
	using relax_stencil = typename solver_types::relax_stencil;
	using stypes = solver_types;

	level(stencil_op<sten> & A) : A(A) {}
	template <class... Args> level(Args &&... args)
		: Adata(std::forward<Args>(args)...), A(Adata), P(std::forward<Args>(args)...), x(std::forward<Args>(args)...),
		  res(std::forward<Args>(args)...), b(std::forward<Args>(args)...) {}
	level(const level &) = delete;
	stencil_op<sten> Adata;
	stencil_op<sten> & A;
	prolong_op P;
	restrict_op R;
	grid_func x;
	grid_func res;
	grid_func b;
	std::array<relax_stencil, 2> SOR;
	std::function<void(const stencil_op<sten> & A, grid_func & x, const grid_func & b)> presmoother;
	std::function<void(const stencil_op<sten> & A, grid_func & x, const grid_func & b)> postsmoother;
};

// level 0 = fine (stencil fsten, refers to the caller's operator), levels 1.. = coarse (full stencil).
// get(i) returns a full-stencil level, get<fsten>(0) the fine one -- the reference's access pattern.
// `touch` (set by the solver) makes the host-side arrays exist before anything is handed out: the solver keeps its
// hierarchy in HBM and materialises host views only on demand.
template <template <class> class level_tmpl, class fsten, class full_sten> class level_container {
public:
	template <class sten> using level_t = level_tmpl<sten>;
	template <class sten> using stencil_op = typename level_tmpl<fsten>::template stencil_op<sten>;
	explicit level_container(stencil_op<fsten> & fop) : fine(fop) {}
	void init(std::size_t) {}
	template <class... Args> void add(Args &&... args) { coarse.emplace_back(std::forward<Args>(args)...); }
	std::size_t size() { if (touch) touch(); return 1 + coarse.size(); }
	template <class sten = full_sten> level_tmpl<sten> & get(std::size_t i)
	{
		if (touch) touch();
		if constexpr (std::is_same<sten, fsten>::value) {
			if (i == 0) return fine;
			if constexpr (std::is_same<sten, full_sten>::value) return coarse[i - 1];
			else { log::error << "coarse operators use the full stencil!" << std::endl; return fine; }
		} else {
			if (i == 0) { log::error << "fine grid operator uses the compact stencil!" << std::endl; return coarse[0]; }
			return coarse[i - 1];
		}
	}
	std::function<void()> touch;
	level_tmpl<fsten> fine;
	std::deque<level_tmpl<full_sten>> coarse; // deque: references to levels stay valid while levels are added
};
}
#endif
