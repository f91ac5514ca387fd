// slots [6, 7, 8, 9] of the compile-time specialised 3D Galerkin product (see galerkin3_unrolled.inc)
#include "galerkin3_unrolled.inc"

namespace cedar_amd {
void galerkin3_part2(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                     int IIC, int JJC, int KKC, int ifd, hipStream_t st)
{
	launch_slot<6>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
	launch_slot<7>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
	launch_slot<8>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
	launch_slot<9>(so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, ifd, st);
}
} // namespace cedar_amd
