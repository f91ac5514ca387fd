// Boost-free mirror of the part of Cedar's public C++ surface that the BoxMG hot path needs
// (reference include/cedar/types.h:43-44: real_t = double, len_t = unsigned int).
// Everything in include/cedar/ is host-side glue over the C ABI of libcedar_amd.so
// (include/cedar_amd.h); no arithmetic of the hot path lives here.
#ifndef CEDAR_TYPES_H
#define CEDAR_TYPES_H
#include <cstddef>
#include <iostream>
#include <vector>

namespace cedar {
using real_t = double;
using len_t = unsigned int;

// half-open integer range usable in range-for, like cedar::range(a, b)
template <class T> class range_t {
public:
	struct it {
		T v;
		T operator*() const { return v; }
		it & operator++() { ++v; return *this; }
		bool operator!=(const it & o) const { return v != o.v; }
	};
	range_t() : b_(0), e_(0) {}
	range_t(T b, T e) : b_(b), e_(e) {}
	it begin() const { return it{b_}; }
	it end() const { return it{e_}; }
	T front() const { return b_; }
	T back() const { return e_ - 1; }
private:
	T b_, e_;
};
template <class T> range_t<T> range(T b, T e) { return range_t<T>(b, e); }
template <class T> range_t<T> range(T e) { return range_t<T>(T(0), e); }

// levelled log streams (reference src/util/log.cc); status/info/error go to stdout/stderr
namespace log {
struct stream {
	std::ostream * os;
	bool on;
	bool active() const { return on; }
	template <class T> stream & operator<<(const T & v) { if (on) (*os) << v; return *this; }
	stream & operator<<(std::ostream & (*f)(std::ostream &)) { if (on) (*os) << f; return *this; }
};
inline stream status{&std::cout, true};
inline stream info{&std::cout, false};
inline stream error{&std::cerr, true};
inline stream debug{&std::cout, false};
}
namespace cycle { enum class Dir { DOWN = 0, UP = 1 }; }
enum class relax_dir { x, y, xy, xz, yz }; // include/cedar/types.h:25
template <relax_dir rdir> struct relax_dir_name { static constexpr const char * value = ""; };
template <> struct relax_dir_name<relax_dir::xy> { static constexpr const char * value = "xy"; };
template <> struct relax_dir_name<relax_dir::xz> { static constexpr const char * value = "xz"; };
template <> struct relax_dir_name<relax_dir::yz> { static constexpr const char * value = "yz"; };
}
#endif
