// 3D Galerkin coarse operator with the fine operator staged through LDS.
//
// The slot kernels (galerkin3_unrolled.inc) give a workgroup 128 coarse points of one coarse row and read the fine
// operator straight from memory: a coarse point reaches 6 x 6 x 6 fine points, neighbouring rows and planes of
// coarse points live in other workgroups, and the fourteen slots are fourteen launches -- the 512^3 product moves
// ~430 GB (PMC) against 15 GB of operator.  Here one workgroup owns a tile of 4 x 4 x 2 coarse points, loads the
// 12 x 12 x 8 fine points they reach (14 planes, 129 KB of the 160 KB LDS, zeros outside the grid) once, and
// evaluates all fourteen slots of its 32 points from it: rap_slot<S> itself is reused unchanged, called with the
// tile as its "fine grid" (local coarse index = global - tile origin + 3) and the interpolation array shifted to
// match.  Entries outside the grid are stored as zeros, which is what the bound tests of the CHECK variant
// amount to; same terms in the same order otherwise.
#include "galerkin3_unrolled.inc"

namespace cedar_amd {

namespace {
constexpr int TCX = 4, TCY = 4, TCZ = 2;                              // coarse points of a tile
constexpr int TFX = 2 * TCX + 4, TFY = 2 * TCY + 4, TFZ = 2 * TCZ + 4; // fine points they reach: 2(c-1)-3 .. 2(c-1)+2
constexpr int NPT = TCX * TCY * TCZ;
constexpr int NTHR = 14 * NPT;
}

template <int S, bool SEVEN>
__device__ __forceinline__ void tile_slot(const real_t *__restrict__ tile, const real_t *__restrict__ ci_l, real_t *__restrict__ soc,
                                          int IIC, int JJC, int KKC, int ic, int jc, int kc, int il, int jl, int kl)
{
	const real_t v = rap_slot<S, SEVEN, false>(tile, ci_l, TFX, TFY, TFZ, IIC, JJC, KKC, il, jl, kl);
	soc[(size_t)(ic - 1) + (size_t)IIC * ((size_t)(jc - 1) + (size_t)JJC * ((size_t)(kc - 1) + (size_t)KKC * (size_t)S))] = v;
}

template <bool SEVEN>
__global__ __launch_bounds__(NTHR) void galerkin3_tiled_kernel(const real_t *__restrict__ so, real_t *__restrict__ soc,
                                                                const real_t *__restrict__ ci, int IIF, int JJF, int KKF,
                                                                int IIC, int JJC, int KKC)
{
	extern __shared__ __attribute__((aligned(16))) real_t tile[];
	constexpr int NPL = SEVEN ? 4 : 14; // fine operator planes
	const int ic0 = 2 + TCX * (int)blockIdx.x, jc0 = 2 + TCY * (int)blockIdx.y, kc0 = 2 + TCZ * (int)blockIdx.z;
	const int i0 = 2 * (ic0 - 1) - 3, j0 = 2 * (jc0 - 1) - 3, k0 = 2 * (kc0 - 1) - 3; // 1-based fine origin of the tile
	const size_t PF = (size_t)IIF * JJF * KKF;
	for (int t = threadIdx.x; t < NPL * TFZ * TFY * TFX; t += NTHR) {
		const int x = t % TFX, y = (t / TFX) % TFY, z = (t / (TFX * TFY)) % TFZ, s = t / (TFX * TFY * TFZ);
		const int gi = i0 + x, gj = j0 + y, gk = k0 + z;
		real_t v = 0.0;
		if (gi >= 1 && gi <= IIF && gj >= 1 && gj <= JJF && gk >= 1 && gk <= KKF)
			v = so[(size_t)(gi - 1) + (size_t)IIF * ((size_t)(gj - 1) + (size_t)JJF * (size_t)(gk - 1)) + PF * (size_t)s];
		tile[t] = v;
	}
	__syncthreads();
	const int p = threadIdx.x % NPT, s = threadIdx.x / NPT;
	const int px = p % TCX, py = (p / TCX) % TCY, pz = p / (TCX * TCY);
	const int ic = ic0 + px, jc = jc0 + py, kc = kc0 + pz;
	if (ic > IIC - 1 || jc > JJC - 1 || kc > KKC - 1) return;
	// local coarse index c - c0 + 3: F(local) = 2(local-1) = F(c) - (tile origin - 1)
	const int il = px + 3, jl = py + 3, kl = pz + 3;
	const real_t *ci_l = ci + ((ptrdiff_t)(ic0 - 3) + (ptrdiff_t)IIC * ((ptrdiff_t)(jc0 - 3) + (ptrdiff_t)JJC * (ptrdiff_t)(kc0 - 3)));
#define SLOT(Sv) case Sv: tile_slot<Sv, SEVEN>(tile, ci_l, soc, IIC, JJC, KKC, ic, jc, kc, il, jl, kl); break;
	switch (s) {
		SLOT(0) SLOT(1) SLOT(2) SLOT(3) SLOT(4) SLOT(5) SLOT(6) SLOT(7) SLOT(8) SLOT(9) SLOT(10) SLOT(11) SLOT(12) SLOT(13)
	}
#undef SLOT
}

// returns false when this variant does not serve the request (the caller runs the slot kernels)
bool galerkin3_tiled(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                     int IIC, int JJC, int KKC, int ifd, hipStream_t st)
{
	if (ifd == 1 || IIC < 3 || JJC < 3 || KKC < 3) return false;
	const size_t shm = (size_t)14 * TFZ * TFY * TFX * sizeof(real_t);
	auto k = galerkin3_tiled_kernel<false>;
	static bool once = false;
	if (!once) {
		CEDAR_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
		once = true;
	}
	dim3 grid((IIC - 2 + TCX - 1) / TCX, (JJC - 2 + TCY - 1) / TCY, (KKC - 2 + TCZ - 1) / TCZ);
	hipLaunchKernelGGL(k, grid, dim3(NTHR), shm, st, so, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC);
	return true;
}

} // namespace cedar_amd
