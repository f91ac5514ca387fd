// N-dimensional array, first index fastest (reference include/cedar/array.h:15-17,67-74).
#ifndef CEDAR_ARRAY_H
#define CEDAR_ARRAY_H
#include <array>
#include <cassert>
#include <cedar/types.h>

namespace cedar {
template <class T, unsigned short ND> class array {
public:
	array() { ext.fill(0); str.fill(0); }
	template <class... E> explicit array(E... e) { reshape(e...); }
	template <class... E> void reshape(E... e)
	{
		static_assert(sizeof...(E) == ND, "extent count");
		len_t tmp[ND] = {static_cast<len_t>(e)...};
		std::size_t n = 1;
		for (unsigned short d = 0; d < ND; d++) { ext[d] = tmp[d]; str[d] = n; n *= tmp[d]; }
		vec.assign(n, T(0));
	}
	template <class... I> T & operator()(I... i) { return vec[off(i...)]; }
	template <class... I> const T & operator()(I... i) const { return vec[off(i...)]; }
	len_t len(unsigned short d) const { return ext[d]; }
	std::size_t stride(unsigned short d) const { return str[d]; }
	T * data() { return vec.data(); }
	const T * data() const { return vec.data(); }
	std::size_t size() const { return vec.size(); }
	void set(T v) { for (auto & x : vec) x = v; }
protected:
	template <class... I> std::size_t off(I... i) const
	{
		static_assert(sizeof...(I) == ND, "index count");
		std::size_t idx[ND] = {static_cast<std::size_t>(i)...};
		std::size_t o = 0;
		for (unsigned short d = 0; d < ND; d++) o += idx[d] * str[d];
		return o;
	}
	std::vector<T> vec;
	std::array<len_t, ND> ext;
	std::array<std::size_t, ND> str;
};
}
#endif
