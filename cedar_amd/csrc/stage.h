// Host/device pointer staging for the C-ABI drop-ins.
// An array argument of a BMG*_SymStd_* entry point may live in host memory
// (what Cedar's std::vector-backed arrays give) or in HBM.  `Staged` resolves
// that once per call: device pointers are used in place, host pointers are
// mirrored in a pooled HBM buffer (copied in on construction when `in`, copied
// back on destruction when `out`).
#pragma once
#include "common.h"
#include <map>
#include <vector>

extern "C" void print_error(char *msg); // host-overridable error callback (capi.cpp)

namespace cedar_amd {

hipStream_t current_stream();
bool is_device_ptr(const void *p);
void *pool_get(size_t bytes);
void pool_put(void *p, size_t bytes);
// A kernel launch that the runtime rejects (grid dimension above 65535 rows, more LDS than requested with
// hipFuncSetAttribute, ...) does not return an error from hipLaunchKernelGGL: it leaves one for hipGetLastError and
// the kernel simply does not run.  Every C-ABI entry point ends with this check (the drop-ins through ~Staged, the
// handle API explicitly), so a rejected launch is reported through print_error instead of leaving stale results.
void launch_check(const char *who);

class Staged {
public:
	Staged(const real_t *p, size_t n, bool in, bool out)
	    : host_(const_cast<real_t *>(p)), n_(n), out_(out), owned_(false), dev_(nullptr)
	{
		if (p == nullptr || n == 0) return;
		if (is_device_ptr(p)) {
			dev_ = host_;
			return;
		}
		owned_ = true;
		dev_ = static_cast<real_t *>(pool_get(n * sizeof(real_t)));
		if (in)
			CEDAR_HIP_CHECK(hipMemcpyAsync(dev_, host_, n * sizeof(real_t), hipMemcpyHostToDevice, current_stream()));
	}
	~Staged()
	{
		launch_check("cedar_amd kernel launch");
		if (!owned_) return;
		if (out_) {
			CEDAR_HIP_CHECK(hipMemcpyAsync(host_, dev_, n_ * sizeof(real_t), hipMemcpyDeviceToHost, current_stream()));
		}
		CEDAR_HIP_CHECK(hipStreamSynchronize(current_stream()));
		pool_put(dev_, n_ * sizeof(real_t));
	}
	real_t *get() const { return dev_; }
	bool staged() const { return owned_; }
	Staged(const Staged &) = delete;
	Staged &operator=(const Staged &) = delete;

private:
	real_t *host_;
	size_t n_;
	bool out_, owned_;
	real_t *dev_;
};

} // namespace cedar_amd
