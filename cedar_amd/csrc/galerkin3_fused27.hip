// fused-launch 3D Galerkin product, 27-point fine operator (see galerkin3_fused.inc)
#include "galerkin3_fused.inc"

namespace cedar_amd {
void galerkin3_fused27(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
        int IIC, int JJC, int KKC, hipStream_t st)
{
	launch_fused<false>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, st);
}
} // namespace cedar_amd
