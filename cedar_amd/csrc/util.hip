// Small device utilities of the library.
#include "common.h"

namespace cedar_amd {

__global__ __launch_bounds__(256) void zero_fill_kernel(real_t *__restrict__ p, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

void zero_fill(real_t *p, size_t n, hipStream_t st)
{
	if (!n) return;
	size_t blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
}

} // namespace cedar_amd
