// 3D Galerkin coarse operator in two stages (BMG3_SymStd_SETUP_ITLI27_ex.f90:84-1888, ITLI07_ex.f90:84-1005).
//
// The one-stage kernels (galerkin3_unrolled.inc) evaluate, for the coarse entry (slot S at coarse point C) that
// couples c1 = C+A[S] with c2 = C+B[S],
//        +- sum_{f1 in N(c1)}  P(f1,c1) * row(f1,c2),     row(f1,c2) = sum_{f2 in N(f1) ∩ N(c2)} +-A(f1,f2) P(f2,c2)
// and recompute row(f1,c2) for every (S, C) that meets it -- up to eight coarse points share a fine point.  row(f1,c2)
// depends on f1 and c2 only.  Stage 1 computes it once per fine point and neighbouring coarse point into
// T (27 planes on the fine grid: per direction c2 is one of the <= 3 coarse indices whose support meets N(f1)),
// stage 2 contracts T with P^T.  The inner sums run in the one-stage kernels' order, so the coarse operators are
// bit-identical, with 8x fewer products on the operator planes.  T for the whole grid would cost 27 fine arrays
// (29 GB at 512^3), so the product runs in slabs of coarse planes through a scratch of at most
// CEDAR_AMD_GALERKIN_SCRATCH_MB (default 2048) that the library keeps; neighbouring slabs recompute the one fine
// plane they share.
//
// EXPERIMENTAL, off by default (CEDAR_AMD_GALERKIN_TWOSTAGE=1): at 512^3 stage 1 takes 74 ms against 61 ms for the
// fourteen one-stage kernels -- it needs 170-256 VGPRs (the 27 operator entries of two fine points and up to 125
// weights in flight), which leaves one or two waves per SIMD to hide its ~230 dependent loads per lane.  Stage 2
// (11 ms) is at its T-read bandwidth.  What is left to try: one fine point per lane and a runtime loop over the
// candidate planes (profiles/r01_experiment_galerkin_twostage.log).
#include "galerkin3_unrolled.inc"

namespace cedar_amd {

// T is indexed by the fine index 0 .. IIF in every direction: N(c1) of the first coarse point reaches fine
// index 0, whose row picks up the operator entries stored at index 1
// and holds the KT fine planes kof .. kof+KT-1 of the current slab
#define TS3(i, j, k, s) T[(size_t)(i) + (size_t)(IIF + 1) * ((size_t)(j) + (size_t)(JJF + 1) * ((size_t)((k) - kof) + (size_t)KT * (size_t)(s)))]

// T plane index of the coarse candidate in one direction.  g = f1 - F(c2) (fine units):
//   f1 coincides with a coarse index (even):  g = +2, 0, -2  ->  0, 1, 2
//   f1 lies between two coarse indices (odd): g = +1, -1     ->  0, 1
__host__ __device__ constexpr int tslot(int g) { return (g & 1) ? (1 - g) / 2 : (2 - g) / 2; }

// row(f1, c2) of one fine point whose index parities are (PX,PY,PZ) (1 = odd = between two coarse indices) for all
// its coarse candidates.  All nine loops sit in one function and unroll over compile-time bounds, so every bound
// test and table look-up folds as in rap_slot.
// CHECK = false for fine points at least two from every face: every operator entry is stored in range and every
// candidate is an interior coarse point, so the tests fold away (the sums keep their order).
template <int PX, int PY, int PZ, bool SEVEN, bool CHECK>
__device__ __forceinline__ void t_point(const real_t *__restrict__ so, const real_t *__restrict__ ci, real_t *__restrict__ T,
                                        int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int kof, int KT, int fi, int fj, int fk)
{
	// coarse index at or just below the fine index: F(c) = 2(c-1)
	const int cx = (fi - PX) / 2 + 1, cy = (fj - PY) / 2 + 1, cz = (fk - PZ) / 2 + 1;
	constexpr int NX = PX ? 2 : 3, NY = PY ? 2 : 3, NZ = PZ ? 2 : 3;
#pragma unroll
	for (int tz = 0; tz < NZ; tz++)
#pragma unroll
		for (int ty = 0; ty < NY; ty++)
#pragma unroll
			for (int tx = 0; tx < NX; tx++) {
				// even: c2 = c-1, c, c+1 (g = 2, 0, -2); odd: c2 = c, c+1 (g = 1, -1)
				const int c2x = cx + (PX ? tx : tx - 1), c2y = cy + (PY ? ty : ty - 1), c2z = cz + (PZ ? tz : tz - 1);
				const int gx = PX ? 1 - 2 * tx : 2 - 2 * tx, gy = PY ? 1 - 2 * ty : 2 - 2 * ty, gz = PZ ? 1 - 2 * tz : 2 - 2 * tz;
				// stage 2 pairs interior coarse points only: c2 in [1, IIC-1]; CI is read at c2 and c2+1
				if (CHECK && (c2x < 1 || c2x > IIC - 1 || c2y < 1 || c2y > JJC - 1 || c2z < 1 || c2z > KKC - 1)) continue;
				real_t row = 0.0;
#pragma unroll
				for (int ez = -1; ez <= 1; ez++)
#pragma unroll
					for (int ey = -1; ey <= 1; ey++)
#pragma unroll
						for (int ex = -1; ex <= 1; ex++) {
							const int rx = gx + ex, ry = gy + ey, rz = gz + ez; // f2 relative to F(c2)
							if (rx < -1 || rx > 1 || ry < -1 || ry > 1 || rz < -1 || rz > 1) continue;
							const g3::OFE o = g3::off_entry(ex, ey, ez);
							if (SEVEN && o.slot > KB) continue;
							const int si = fi + o.sx, sj = fj + o.sy, sk = fk + o.sz;
							if (CHECK && (si < 1 || si > IIF || sj < 1 || sj > JJF || sk < 1 || sk > KKF)) continue; // = the CHECK of rap_slot
							const g3::PWE w = g3::PW[rz + 1][ry + 1][rx + 1];
							const real_t p2 = w.slot < 0 ? 1.0 : CI3(c2x + w.ox, c2y + w.oy, c2z + w.oz, w.slot);
							if (o.slot == 0) row += SO3(si, sj, sk, KP) * p2;
							else row -= SO3(si, sj, sk, o.slot) * p2;
						}
				TS3(fi, fj, fk, tz * 9 + ty * 3 + tx) = row;
				// candidates one after the other: without the fence the scheduler hoists every weight load of the
				// point (up to 125 values) and the kernel drops to one wave per SIMD
				if (tx == NX - 1) __builtin_amdgcn_sched_barrier(0);
			}
}

// a fine index at least two from both faces whose coarse candidates are all interior
__device__ __forceinline__ bool t_inner(int f, int IIF, int IIC) { return f >= 2 && f <= IIF - 1 && f <= 2 * IIC - 6; }

// stage 1: one workgroup per fine row (j,k), lane p owns the columns (2p, 2p+1) = (even, odd); the rows are
// fj = PY, PY+2, ..; the planes fk = k0, k0+2, .. < kof+KT (k0 of parity PZ)
template <int PY, int PZ, bool SEVEN>
__global__ __launch_bounds__(128) void rap_stage1(const real_t *__restrict__ so, const real_t *__restrict__ ci,
                                                   real_t *__restrict__ T, int IIF, int JJF, int KKF,
                                                   int IIC, int JJC, int KKC, int kof, int KT, int k0)
{
	const int fj = PY + 2 * (int)blockIdx.y, fk = k0 + 2 * (int)blockIdx.z;
	if (fj > JJF || fk > KKF || fk >= kof + KT) return;
	const bool rin = t_inner(fj, JJF, JJC) && t_inner(fk, KKF, KKC);
	for (int p = blockIdx.x * blockDim.x + threadIdx.x; 2 * p <= IIF; p += gridDim.x * blockDim.x) {
		const int ie = 2 * p, io = ie + 1;
		if (rin && t_inner(ie, IIF, IIC) && t_inner(io, IIF, IIC)) {
			t_point<0, PY, PZ, SEVEN, false>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, ie, fj, fk);
			t_point<1, PY, PZ, SEVEN, false>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, io, fj, fk);
		} else {
			t_point<0, PY, PZ, SEVEN, true>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, ie, fj, fk);
			if (io <= IIF) t_point<1, PY, PZ, SEVEN, true>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, io, fj, fk);
		}
	}
}

// stage 2: slot S of the coarse point (ic,jc,kc):  +- sum_d P(f1,c1) T(f1,c2), d in the order of rap_slot
template <int S>
__device__ __forceinline__ real_t rap_contract(const real_t *__restrict__ T, const real_t *__restrict__ ci,
                                               int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int kof, int KT,
                                               int ic, int jc, int kc)
{
	constexpr g3::V3 a = g3::A[S], b = g3::B[S];
	const int c1x = ic + a.x, c1y = jc + a.y, c1z = kc + a.z;
	const int f1x = 2 * (c1x - 1), f1y = 2 * (c1y - 1), f1z = 2 * (c1z - 1);
	real_t acc = 0.0;
#pragma unroll
	for (int dz = -1; dz <= 1; dz++)
#pragma unroll
		for (int dy = -1; dy <= 1; dy++)
#pragma unroll
			for (int dx = -1; dx <= 1; dx++) {
				const int gx = 2 * (a.x - b.x) + dx, gy = 2 * (a.y - b.y) + dy, gz = 2 * (a.z - b.z) + dz;
				if (gx < -2 || gx > 2 || gy < -2 || gy > 2 || gz < -2 || gz > 2) continue; // N(f1) misses N(c2)
				const int fi = f1x + dx, fj = f1y + dy, fk = f1z + dz;
				if (fi < 0 || fj < 0 || fk < 0) continue; // below the first coarse point: no operator entry, row = 0
				const real_t row = TS3(fi, fj, fk, tslot(gz) * 9 + tslot(gy) * 3 + tslot(gx));
				const g3::PWE w = g3::PW[dz + 1][dy + 1][dx + 1];
				const real_t p1 = w.slot < 0 ? 1.0 : CI3(c1x + w.ox, c1y + w.oy, c1z + w.oz, w.slot);
				acc += p1 * row;
			}
	return (S == KP) ? acc : -acc;
}

// coarse planes kc0 .. kc0+gridDim.z-1
__global__ __launch_bounds__(128) void rap_stage2(const real_t *__restrict__ T, real_t *__restrict__ soc,
                                                   const real_t *__restrict__ ci, int IIF, int JJF, int KKF,
                                                   int IIC, int JJC, int KKC, int kof, int KT, int kc0)
{
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 2, jc = blockIdx.y + 2, kc = blockIdx.z + kc0;
	if (ic > IIC - 1) return;
	const size_t c = (size_t)(ic - 1) + (size_t)IIC * ((size_t)(jc - 1) + (size_t)JJC * (size_t)(kc - 1));
	const size_t PC = (size_t)IIC * JJC * KKC;
#define SLOT(Sv) soc[c + PC * (size_t)(Sv)] = rap_contract<Sv>(T, ci, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, ic, jc, kc);
	SLOT(0) SLOT(1) SLOT(2) SLOT(3) SLOT(4) SLOT(5) SLOT(6) SLOT(7) SLOT(8) SLOT(9) SLOT(10) SLOT(11) SLOT(12) SLOT(13)
#undef SLOT
}

template <int PY, int PZ>
static void launch_stage1(const real_t *so, const real_t *ci, real_t *T, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC,
                          int kof, int KT, int ifd, hipStream_t st)
{
	const int k0 = kof + ((kof & 1) != PZ); // first plane of parity PZ in the slab
	const int khi = (kof + KT - 1 < KKF) ? kof + KT - 1 : KKF;
	const int nrj = (JJF - PY) / 2 + 1, nrk = (khi - k0) / 2 + 1;
	if (nrj <= 0 || khi < k0) return;
	const int npairs = IIF / 2 + 1;
	dim3 grid((npairs + 127) / 128, nrj, nrk);
	if (ifd == 1)
		hipLaunchKernelGGL((rap_stage1<PY, PZ, true>), grid, dim3(128), 0, st, so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, k0);
	else
		hipLaunchKernelGGL((rap_stage1<PY, PZ, false>), grid, dim3(128), 0, st, so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, k0);
}

constexpr bool slots_stay_in_plane()
{
	for (int s = 0; s < 14; s++)
		if (g3::A[s].z != 0) return false;
	return true;
}

// the scratch is kept by the library and only grows (launches on one stream reuse it in order)
static real_t *g_scratch = nullptr;
static size_t g_scratch_bytes = 0;

// returns false when the scratch cannot be had (the caller then runs the one-stage kernels)
bool galerkin3_twostage(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                        int IIC, int JJC, int KKC, int ifd, hipStream_t st)
{
	if (KKC < 3) return false;
	const char *e = getenv("CEDAR_AMD_GALERKIN_SCRATCH_MB");
	const size_t cap = (size_t)((e && atol(e) > 0) ? atol(e) : 2048) << 20;
	const size_t plane = (size_t)27 * (IIF + 1) * (JJF + 1) * sizeof(real_t); // one fine plane of T
	// c1 lies in the plane of C for every slot (A[S].z = 0), so a slab of S coarse planes from kc0 reads the fine
	// planes 2(kc0-1)-1 .. 2(kc0+S-2)+1: 2S+1 of them
	static_assert(slots_stay_in_plane(), "the slab window assumes A[S].z = 0");
	const int ncz = KKC - 2;
	int S = (cap / plane >= 9) ? (int)((cap / plane - 1) / 2) : 4;
	if (S > ncz) S = ncz;
	const int KT = (2 * S + 1 < KKF + 1) ? 2 * S + 1 : KKF + 1;
	const size_t need = plane * (size_t)KT;
	if (need > g_scratch_bytes) {
		CEDAR_HIP_CHECK(hipStreamSynchronize(st));
		if (g_scratch) CEDAR_HIP_CHECK(hipFree(g_scratch));
		g_scratch = nullptr, g_scratch_bytes = 0;
		if (hipMalloc((void **)&g_scratch, need) != hipSuccess) {
			(void)hipGetLastError();
			return false;
		}
		g_scratch_bytes = need;
	}
	real_t *T = g_scratch;
	for (int kc0 = 2; kc0 <= KKC - 1; kc0 += S) {
		const int ns = (kc0 + S - 1 <= KKC - 1) ? S : KKC - kc0;
		int kof = 2 * (kc0 - 1) - 1;
		if (kof + KT > KKF + 1) kof = KKF + 1 - KT; // the last slab: keep the window inside the fine planes
		launch_stage1<0, 0>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, ifd, st);
		launch_stage1<1, 0>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, ifd, st);
		launch_stage1<0, 1>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, ifd, st);
		launch_stage1<1, 1>(so, ci, T, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, ifd, st);
		dim3 grid((IIC - 2 + 127) / 128, JJC - 2, ns);
		hipLaunchKernelGGL(rap_stage2, grid, dim3(128), 0, st, T, soc, ci, IIF, JJF, KKF, IIC, JJC, KKC, kof, KT, kc0);
	}
	return true;
}

#undef TS3
} // namespace cedar_amd
