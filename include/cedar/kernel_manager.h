// Named-implementation registry for the abstract kernels (reference include/cedar/kernel_manager.h:22-95,
// include/cedar/type_list.h:63-123): add<T,impl>(name[, ctor args]), set<T>(name), setup<T>(...), run<T>(...),
// add_halo().  The reference registers its Fortran bindings as "system"; this build registers "hip"
// (bindings over the BMG2_/BMG3_SymStd_* entry points of libcedar_amd.so).  The first implementation added for a
// kernel is the selected one until set<T>() names another.
#ifndef CEDAR_KERNEL_MANAGER_H
#define CEDAR_KERNEL_MANAGER_H
#include <map>
#include <memory>
#include <string>
#include <typeindex>
#include <utility>
#include <cedar/config.h>
#include <cedar/kernel.h>

namespace cedar {
class kernel_manager {
public:
	explicit kernel_manager(std::shared_ptr<kernel_params> p) : params(p) {}
	explicit kernel_manager(config & conf) : params(build_kernel_params(conf)) {}
	template <class T, class impl, class... Args> void add(const std::string & name, Args &&... args)
	{
		auto k = std::make_shared<impl>(std::forward<Args>(args)...);
		k->add_params(params);
		k->add_halo(halof);
		impls[std::type_index(typeid(T))][name] = k;
		if (!selected.count(std::type_index(typeid(T)))) selected[std::type_index(typeid(T))] = name;
	}
	template <class T> void set(const std::string & name)
	{
		auto & m = impls[std::type_index(typeid(T))];
		if (!m.count(name)) { log::error << "kernel <" << T::name() << "> has no implementation named " << name << std::endl; return; }
		selected[std::type_index(typeid(T))] = name;
	}
	template <class T> T * get_ptr()
	{
		auto ti = std::type_index(typeid(T));
		auto s = selected.find(ti);
		if (s == selected.end()) return nullptr;
		return dynamic_cast<T *>(impls[ti][s->second].get());
	}
	template <class T> T & get(const std::string & name) { return *dynamic_cast<T *>(impls[std::type_index(typeid(T))][name].get()); }
	// name of the selected implementation ("" if the kernel has none)
	template <class T> std::string selected_name()
	{
		auto s = selected.find(std::type_index(typeid(T)));
		return s == selected.end() ? std::string() : s->second;
	}
	// true when every kernel that has implementations runs the one called `name`
	bool all_selected(const std::string & name) const
	{
		for (auto & s : selected) if (s.second != name) return false;
		return true;
	}
	template <class T, class... Args> void setup(Args &&... args)
	{
		auto k = get_ptr<T>();
		log::debug << "setup kernel <" << T::name() << ">" << std::endl;
		if (!k) { log::error << "kernel not found: " << T::name() << std::endl; return; }
		k->setup(std::forward<Args>(args)...);
	}
	template <class T, class... Args> void run(Args &&... args)
	{
		auto k = get_ptr<T>();
		log::debug << "running kernel <" << T::name() << ">" << std::endl;
		if (!k) { log::error << "kernel not found: " << T::name() << std::endl; return; }
		k->run(std::forward<Args>(args)...);
	}
	// hand the halo exchanger to every registered implementation (reference kernel_manager.h:60-95)
	void add_halo(halo_exchanger_base * h)
	{
		halof = h;
		for (auto & m : impls) for (auto & k : m.second) k.second->add_halo(h);
	}
	std::shared_ptr<kernel_params> get_params() { return params; }
private:
	std::shared_ptr<kernel_params> params;
	halo_exchanger_base * halof = nullptr;
	std::map<std::type_index, std::map<std::string, std::shared_ptr<kernel_base>>> impls;
	std::map<std::type_index, std::string> selected;
};
}
#endif
