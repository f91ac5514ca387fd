// Minimal reader for Cedar's config.json (reference src/config.cc, include/cedar/config.h:76-103):
// dotted-path getters with defaults.  Supports the JSON subset Cedar's configs use
// (objects, arrays, strings, numbers, booleans).
#ifndef CEDAR_CONFIG_H
#define CEDAR_CONFIG_H
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <cedar/types.h>

namespace cedar {
class config {
public:
	config() : config("config.json") {}
	explicit config(const std::string & fname)
	{
		std::ifstream f(fname);
		if (!f) { log::status << "config: " << fname << " not found, using defaults" << std::endl; return; }
		std::stringstream ss; ss << f.rdbuf();
		txt = ss.str(); pos = 0;
		parse_value("");
	}
	template <class T> T get(const std::string & path, T dflt) const
	{
		auto it = scalars.find(path);
		if (it == scalars.end()) return dflt;
		return conv<T>(it->second);
	}
	template <class T> T get(const std::string & path) const { return get<T>(path, T()); }
	template <class T> std::vector<T> getvec(const std::string & path) const
	{
		std::vector<T> out;
		for (std::size_t i = 0;; i++) {
			auto it = scalars.find(path + "." + std::to_string(i));
			if (it == scalars.end()) break;
			out.push_back(conv<T>(it->second));
		}
		return out;
	}
	template <class T> void set(const std::string & path, T v) { std::ostringstream o; o << v; scalars[path] = o.str(); }
	// the sub-object at `path` as a configuration of its own (reference config::getconf); nullptr if absent
	std::shared_ptr<config> getconf(const std::string & path) const
	{
		auto sub = std::make_shared<config>(empty_tag());
		const std::string pre = path + ".";
		for (auto & kv : scalars)
			if (kv.first.compare(0, pre.size(), pre) == 0) sub->scalars[kv.first.substr(pre.size())] = kv.second;
		return sub->scalars.empty() ? nullptr : sub;
	}
	struct empty_tag {};
	explicit config(empty_tag) {}
private:
	template <class T> static T conv(const std::string & s)
	{
		// JSON booleans read as numbers / bools (grid.periodic is an array of true/false)
		std::istringstream i(s == "true" ? "1" : s == "false" ? "0" : s);
		T v{}; i >> v; return v;
	}
	void ws() { while (pos < txt.size() && std::isspace((unsigned char)txt[pos])) pos++; }
	std::string parse_string()
	{
		std::string s; pos++;
		while (pos < txt.size() && txt[pos] != '"') { if (txt[pos] == '\\') pos++; s += txt[pos++]; }
		pos++;
		return s;
	}
	void parse_value(const std::string & path)
	{
		ws();
		if (pos >= txt.size()) return;
		char c = txt[pos];
		if (c == '{') {
			pos++; ws();
			while (pos < txt.size() && txt[pos] != '}') {
				ws(); std::string k = parse_string(); ws(); pos++; /* ':' */
				parse_value(path.empty() ? k : path + "." + k);
				ws(); if (txt[pos] == ',') pos++; ws();
			}
			pos++;
		} else if (c == '[') {
			pos++; ws(); std::size_t i = 0;
			while (pos < txt.size() && txt[pos] != ']') {
				parse_value(path + "." + std::to_string(i++));
				ws(); if (txt[pos] == ',') pos++; ws();
			}
			pos++;
		} else if (c == '"') {
			scalars[path] = parse_string();
		} else {
			std::size_t b = pos;
			while (pos < txt.size() && txt[pos] != ',' && txt[pos] != '}' && txt[pos] != ']' && !std::isspace((unsigned char)txt[pos])) pos++;
			std::string tok = txt.substr(b, pos - b);
			if (tok == "true") tok = "1";
			if (tok == "false") tok = "0";
			scalars[path] = tok;
		}
	}
	std::map<std::string, std::string> scalars;
	std::string txt; std::size_t pos = 0;
};
template <> inline std::string config::conv<std::string>(const std::string & s) { return s; }

// reference src/kernel_params.cc:63-70 (periodic flags; relax_symmetric = definite = true)
struct kernel_params {
	std::array<bool, 3> periodic{{false, false, false}};
	bool relax_symmetric = true, definite = true;
	std::shared_ptr<config> plane_config; // include/cedar/kernel_params.h:32
	int per_mask() const { int m = 0; for (int i = 0; i < 3; i++) if (periodic[i]) m |= 1 << i; return m; }
};
inline std::shared_ptr<kernel_params> build_kernel_params(config & conf)
{
	auto p = std::make_shared<kernel_params>();
	auto per = conf.getvec<int>("grid.periodic");
	for (std::size_t i = 0; i < per.size() && i < 3; i++) p->periodic[i] = per[i] != 0;
	p->plane_config = conf.getconf("plane-config"); // src/kernel_params.cc:72-78
	if (p->plane_config == nullptr) {
		p->plane_config = std::make_shared<config>(config::empty_tag());
		p->plane_config->set("solver.relaxation", "line-xy");
		p->plane_config->set("solver.max-iter", 1);
	}
	return p;
}

// reference src/multilevel_settings.cc:15-61
struct ml_settings {
	enum class relax_type { point, line_x, line_y, line_xy, plane_xy, plane_xz, plane_yz, plane_xyz };
	bool planes() const { return relaxation >= relax_type::plane_xy; }
	relax_type relaxation = relax_type::point;
	int nrelax_pre = 2, nrelax_post = 1, num_levels = -1, maxiter = 10, min_coarse = 3;
	int cycle = 0; // 0 = v, 1 = f
	real_t tol = 1e-8;
	void init(config & conf)
	{
		auto r = conf.get<std::string>("solver.relaxation", "point");
		if (r == "point") relaxation = relax_type::point;
		else if (r == "line-x") relaxation = relax_type::line_x;
		else if (r == "line-y") relaxation = relax_type::line_y;
		else if (r == "line-xy") relaxation = relax_type::line_xy;
		else if (r == "plane-xy") relaxation = relax_type::plane_xy; // src/multilevel_settings.cc:10-13
		else if (r == "plane-xz") relaxation = relax_type::plane_xz;
		else if (r == "plane-yz") relaxation = relax_type::plane_yz;
		else if (r == "plane-xyz") relaxation = relax_type::plane_xyz;
		else log::error << "invalid relaxation type: " << r << std::endl;
		auto cyc = conf.get<std::string>("solver.cycle.type", "v");
		if (cyc == "v") cycle = 0;
		else if (cyc == "f") cycle = 1;
		else log::error << "invalid cycle type: " << cyc << std::endl;
		nrelax_pre = conf.get<int>("solver.cycle.nrelax-pre", 2);
		nrelax_post = conf.get<int>("solver.cycle.nrelax-post", 1);
		num_levels = conf.get<int>("solver.num-levels", -1);
		maxiter = conf.get<int>("solver.max-iter", 10);
		tol = conf.get<real_t>("solver.tol", 1e-8);
		min_coarse = conf.get<int>("solver.min_coarse", 3);
	}
};
}
#endif
