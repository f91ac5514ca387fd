// slots [10, 11, 12, 13] of the compile-time specialised 3D Galerkin product (see galerkin3_unrolled.inc)
#include "galerkin3_unrolled.inc"

namespace cedar_amd {
void galerkin3_part3(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                     int IIC, int JJC, int KKC, int ifd, hipStream_t st)
{
	launch_slot<10>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
	launch_slot<11>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
	launch_slot<12>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
	launch_slot<13>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
}
} // namespace cedar_amd
