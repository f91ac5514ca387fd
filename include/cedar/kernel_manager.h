// Named-implementation registry for the abstract kernels (reference include/cedar/kernel_manager.h:22-69,
// include/cedar/type_list.h:63-123): add<T,impl>(name), set<T>(name), setup<T>(...), run<T>(...).
// The reference registers its Fortran bindings as "system"; this build registers "hip".
#ifndef CEDAR_KERNEL_MANAGER_H
#define CEDAR_KERNEL_MANAGER_H
#include <map>
#include <memory>
#include <string>
#include <typeindex>
#include <utility>
#include <cedar/config.h>

namespace cedar {
// base of every kernel implementation: carries the shared kernel_params (reference include/cedar/kernel.h:25-37)
struct kernel_base {
	virtual ~kernel_base() {}
	void add_params(std::shared_ptr<kernel_params> p) { params = p; }
	std::shared_ptr<kernel_params> params;
};

class kernel_manager {
public:
	explicit kernel_manager(std::shared_ptr<kernel_params> p) : params(p) {}
	explicit kernel_manager(config & conf) : params(build_kernel_params(conf)) {}
	template <class T, class impl, class... Args> void add(const std::string & name, Args &&... args)
	{
		auto k = std::make_shared<impl>(std::forward<Args>(args)...);
		k->add_params(params);
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
	template <class T, class... Args> void setup(Args &&... args)
	{
		auto k = get_ptr<T>();
		if (!k) { log::error << "kernel not found: " << T::name() << std::endl; return; }
		k->setup(std::forward<Args>(args)...);
	}
	template <class T, class... Args> void run(Args &&... args)
	{
		auto k = get_ptr<T>();
		if (!k) { log::error << "kernel not found: " << T::name() << std::endl; return; }
		k->run(std::forward<Args>(args)...);
	}
	std::shared_ptr<kernel_params> get_params() { return params; }
private:
	std::shared_ptr<kernel_params> params;
	std::map<std::type_index, std::map<std::string, std::shared_ptr<kernel_base>>> impls;
	std::map<std::type_index, std::string> selected;
};
}
#endif
