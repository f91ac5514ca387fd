// 3D Galerkin coarse operator through row sums kept per coarse point
// (BMG3_SymStd_SETUP_ITLI27_ex.f90:84-1888, ITLI07_ex.f90:84-1005; table-driven product of galerkin3_unrolled.inc).
//
// The one-stage kernels evaluate, for the coarse entry (slot S at coarse point C) that couples c1 = C+A[S] with
// c2 = C+B[S],
//      +- sum_{f1 in N(c1)} P(f1,c1) * row(f1,c2),      row(f1,c2) = sum_{f2 in N(f1) ∩ N(c2)} +-A(f1,f2) P(f2,c2),
// and recompute row(f1,c2) for every (S, C) that meets it: ~4500 operator loads and as many weight loads per coarse
// point, all of them through the caches (55 ms at 512^3, 430 GB of fabric traffic for a 15 GB operator).
//
// row(f1,c2) depends on f1 and c2 only, and for a given c2 it is non-zero for the 5 x 5 x 5 fine points around F(c2).
//   stage 1  one lane per coarse point c2: the 27 weights P(.,c2) are loaded once into registers, then the 125 row
//            sums are formed one after the other (729 operator loads per lane, independent of each other) and stored
//            to T[g][c2] -- coalesced, 125 doubles per coarse point;
//   stage 2  one lane per coarse point C: each of its 14 slots contracts 27 row sums with the weights of c1.
// The additions run in the order of rap_slot (same inner loop over f2, same outer loop over f1), so the coarse
// operator is bit-identical to the one-stage kernels'.  T would be 125 coarse arrays (17 GB at 512^3): the product
// runs over slabs of coarse planes through a ring of T planes (no row sum is computed twice), sized by
// CEDAR_AMD_GALERKIN_SLAB (coarse planes per slab, default 32: 4.4 GB at 512^3).
// Measured at 512^3, 27-point (profiles/r02_experiment_galerkin_rows.log): 54.9 ms -> 27.8 ms with 8-byte operator
// loads (stage 1 18.6 ms + 5.7 ms for the shell of coarse points next to a face, stage 2 4 ms).  Stage 1 is bound by
// L2 bandwidth: a lane's operator loads are 16 bytes apart (fine index = 2 x coarse index), so every 8-byte load
// instruction touches eight lines and uses half of each.  row_group reads aligned pairs instead (a third fewer lines
// per term) and drops the out-of-grid terms of the shell by a select instead of six tests per term:
// 27.8 -> 18.0 ms (stage 1 11.8 ms, shell 2.3 ms, stage 2 4 ms); 7-point 13.7 (one-stage) -> 12.1 ms.
#include "galerkin3_unrolled.inc"
#include <utility>

namespace cedar_amd {

// T[g][slot plane][c2y][c2x], g = (gz+2)*25 + (gy+2)*5 + (gx+2); the plane of c2z is c2z % NB
#define TR(g, x, y, z) T[(size_t)(g) * gstride + (size_t)((x)-1) + (size_t)IIC * ((size_t)((y)-1) + (size_t)JJC * (size_t)((z) % NB))]

// one row sum: G = (gz+2)*25 + (gy+2)*5 + (gx+2) is a template parameter so that, once the three small loops over f2
// are unrolled, every index into the weight array is a constant and the array lives in registers
template <bool SEVEN, bool CHECK, int G>
__device__ __forceinline__ void row_sum(const real_t *__restrict__ so, const real_t *__restrict__ ci, real_t *__restrict__ T,
                                        size_t gstride, int NB, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC,
                                        int c2x, int c2y, int c2z, const real_t (&w)[3][3][3])
{
	constexpr int gz = G / 25 - 2, gy = (G / 5) % 5 - 2, gx = G % 5 - 2;
	const int fi = 2 * (c2x - 1) + gx;
	int fj = 2 * (c2y - 1) + gy, fk = 2 * (c2z - 1) + gz; // uniform over the workgroup (one coarse row per workgroup)
	// the row and plane offsets are re-derived for every row sum: left to itself the compiler shares them across all
	// 125 row sums, keeps some five hundred row bases alive and spills
	if constexpr (!CHECK) asm volatile("" : "+s"(fj), "+s"(fk)); // interior kernel only: there the row is uniform
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
				const g3::PWE we = g3::PW[rz + 1][ry + 1][rx + 1];
				// interior coarse points keep the 27 weights in registers; the few next to a face re-read them
				const real_t p2 = !CHECK ? w[rz + 1][ry + 1][rx + 1]
				                         : (we.slot < 0 ? 1.0 : CI3(c2x + we.ox, c2y + we.oy, c2z + we.oz, we.slot));
				if (o.slot == 0) row += SO3(si, sj, sk, KP) * p2;
				else row -= SO3(si, sj, sk, o.slot) * p2;
			}
	TR(G, c2x, c2y, c2z) = row;
	// one row after the other: without the fence the scheduler hoists the operator loads of many rows (729 are
	// independent) and the kernel drops to two waves per SIMD with spills
	__builtin_amdgcn_sched_barrier(0);
}

template <bool SEVEN, bool CHECK, int... Gs>
__device__ __forceinline__ void row_sums(const real_t *__restrict__ so, const real_t *__restrict__ ci, real_t *__restrict__ T,
                                         size_t gstride, int NB, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC,
                                         int c2x, int c2y, int c2z, std::integer_sequence<int, Gs...>)
{
	real_t w[3][3][3];
	if (!CHECK) {
#pragma unroll
		for (int rz = 0; rz < 3; rz++)
#pragma unroll
			for (int ry = 0; ry < 3; ry++)
#pragma unroll
				for (int rx = 0; rx < 3; rx++) {
					const g3::PWE e = g3::PW[rz][ry][rx];
					w[rz][ry][rx] = e.slot < 0 ? 1.0 : CI3(c2x + e.ox, c2y + e.oy, c2z + e.oz, e.slot);
				}
	}
	(row_sum<SEVEN, CHECK, Gs>(so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, c2x, c2y, c2z, w), ...);
}

// The five row sums gx = -2..2 of one (gy, gz), with the operator read as aligned pairs.  For a given offset
// (ex,ey,ez) three of the five rows have a term (gx+ex in -1..1), and their operator entries are three neighbours in
// x of one slot: two 16-byte loads cover them, and a wave's loads are then contiguous (a lane's entries are 16 bytes
// apart: the 8-byte loads of row_sum touch eight lines per instruction and use half of each).  Needs an even row
// length and a 16-byte aligned operator (pair starts are the odd fine indices).  Every row sum adds its terms in the
// order of row_sum, so the result is the same to the bit.
//   MODE 0  coarse points whose 5^3 neighbourhood lies inside the fine arrays: no bound tests
//   MODE 1  whole coarse rows next to a y or z face: the tests on the fine row and plane are uniform branches
//   MODE 2  the points next to an x face of the other rows, packed (every lane its own row)
// In modes 1 and 2 a term whose entry lies below the first fine index is dropped by selecting 0.0 for the product
// (above the last index there is none: 2(IIC-2)+2 = IIF for an even IIF).  row_sum skips such a term; adding +-0.0
// instead leaves every sum as it is: a sum that starts at +0.0 and only adds or subtracts never becomes -0.0.
// A pair lies either wholly below index 1 or wholly inside the row, so the address of a dropped pair is clamped to
// the start of the operator (for the first row the pair would otherwise lie in front of the allocation).
template <bool SEVEN, int GYZ, int MODE>
__device__ __forceinline__ void row_group(const real_t *__restrict__ so, real_t *__restrict__ T, size_t gstride, int NB,
                                          int IIF, int JJF, int KKF, int IIC, int JJC, int c2x, int c2y, int c2z,
                                          const real_t (&w)[3][3][3])
{
	constexpr int gz = GYZ / 5 - 2, gy = GYZ % 5 - 2;
	int fj = 2 * (c2y - 1) + gy, fk = 2 * (c2z - 1) + gz;
	// as in row_sum: keep the row bases of other groups out of registers (the row is uniform in modes 0 and 1)
	if constexpr (MODE != 2) asm volatile("" : "+s"(fj), "+s"(fk));
	else asm volatile("" : "+v"(fj), "+v"(fk));
	real_t row[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int ez = -1; ez <= 1; ez++) {
		const int rz = gz + ez;
		if (rz < -1 || rz > 1) continue;
#pragma unroll
		for (int ey = -1; ey <= 1; ey++) {
			const int ry = gy + ey;
			if (ry < -1 || ry > 1) continue;
#pragma unroll
			for (int ex = -1; ex <= 1; ex++) {
				const g3::OFE o = g3::off_entry(ex, ey, ez);
				if (SEVEN && o.slot > KB) continue;
				const int sj = fj + o.sy, sk = fk + o.sz;
				if constexpr (MODE == 1)
					if (sj < 1 || sj > JJF || sk < 1 || sk > KKF) continue; // = the CHECK of rap_slot, uniform here
				// rows gx = -1-ex .. 1-ex; their entries sit at fine x = 2(c2x-1) + xo, xo = lo .. lo+2
				const int lo = -1 - ex + o.sx;
				const int ps = (lo & 1) ? lo : lo - 1; // first pair starts at the odd index at or below lo
				real_t v[4];
				if constexpr (MODE == 0) {
					const real_t *p0 = &SO3(2 * (c2x - 1) + ps, sj, sk, o.slot);
					const double2 a = *reinterpret_cast<const double2 *>(p0), b = *reinterpret_cast<const double2 *>(p0 + 2);
					v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
				} else {
					const long long i0 = (long long)(2 * (c2x - 1) + ps - 1) +
					                     (long long)IIF * ((long long)(sj - 1) + (long long)JJF * ((long long)(sk - 1) + (long long)KKF * o.slot));
					const long long ia = i0 < 0 ? 0 : i0, ib = i0 + 2 < 0 ? 0 : i0 + 2;
					const double2 a = *reinterpret_cast<const double2 *>(so + ia), b = *reinterpret_cast<const double2 *>(so + ib);
					v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
				}
#pragma unroll
				for (int gx = -1 - ex; gx <= 1 - ex; gx++) {
					real_t t = v[gx + o.sx - ps] * w[rz + 1][ry + 1][gx + ex + 1];
					if constexpr (MODE != 0) t = (2 * (c2x - 1) + gx + o.sx >= 1) ? t : 0.0;
					if (o.slot == 0) row[gx + 2] += t;
					else row[gx + 2] -= t;
				}
			}
			// one (ey, ez) at a time, six pair loads in flight: the empty asm pins the sums here (the arithmetic is
			// otherwise sunk below the loads of the whole group, which then spill)
			asm volatile("" : "+v"(row[0]), "+v"(row[1]), "+v"(row[2]), "+v"(row[3]), "+v"(row[4]));
			__builtin_amdgcn_sched_barrier(0);
		}
	}
#pragma unroll
	for (int gx = -2; gx <= 2; gx++) TR((gz + 2) * 25 + (gy + 2) * 5 + (gx + 2), c2x, c2y, c2z) = row[gx + 2];
	__builtin_amdgcn_sched_barrier(0);
}

template <bool SEVEN, int MODE, int... GYZs>
__device__ __forceinline__ void row_groups(const real_t *__restrict__ so, const real_t *__restrict__ ci, real_t *__restrict__ T,
                                           size_t gstride, int NB, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC,
                                           int c2x, int c2y, int c2z, std::integer_sequence<int, GYZs...>)
{
	real_t w[3][3][3];
#pragma unroll
	for (int rz = 0; rz < 3; rz++)
#pragma unroll
		for (int ry = 0; ry < 3; ry++)
#pragma unroll
			for (int rx = 0; rx < 3; rx++) {
				const g3::PWE e = g3::PW[rz][ry][rx];
				w[rz][ry][rx] = e.slot < 0 ? 1.0 : CI3(c2x + e.ox, c2y + e.oy, c2z + e.oz, e.slot);
			}
	(row_group<SEVEN, GYZs, MODE>(so, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, c2x, c2y, c2z, w), ...);
}

// the operator entries of the 5^3 fine points around F(c) are stored at fine indices 2(c-1)-2 .. 2(c-1)+3
__device__ __forceinline__ bool rows_inner(int c, int IIF) { return c >= 3 && 2 * (c - 1) + 3 <= IIF; }

// stage 1: coarse points c2 = (1..IIC-1, 1..JJC-1, z0 .. z0+gridDim.z-1).  The coarse points whose 5^3 neighbourhood
// lies inside the fine arrays (all but a shell two points thick) take the path without bound tests and with the
// weights in registers (rap_rows_interior); the shell gets kernels of its own -- one kernel for both would be
// register-allocated for the larger path, and a wave with a single shell lane would walk all of it.
// PAIRS: the operator is read as aligned pairs (row_group); the caller checks the row length and the alignment
template <bool SEVEN, bool PAIRS>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 4)))
void rap_rows_interior(const real_t *__restrict__ so, const real_t *__restrict__ ci, real_t *__restrict__ T, size_t gstride,
                       int NB, int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int z0)
{
	const int c2x = blockIdx.x * blockDim.x + threadIdx.x + 1, c2y = blockIdx.y + 1, c2z = blockIdx.z + z0;
	if (c2x > IIC - 1 || !(rows_inner(c2x, IIF) && rows_inner(c2y, JJF) && rows_inner(c2z, KKF))) return;
	if constexpr (PAIRS)
		row_groups<SEVEN, 0>(so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, c2x, c2y, c2z, std::make_integer_sequence<int, 25>());
	else
		row_sums<SEVEN, false>(so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, c2x, c2y, c2z, std::make_integer_sequence<int, 125>());
}

// shell, part 1: whole coarse rows next to a y or z face.  compact: blockIdx.y runs over the rows next to a y face only
// (c2y = 1, 2 and the last ones) on planes away from the z faces; otherwise over every row of the planes given
// (the caller passes planes next to a z face)
template <bool SEVEN, bool PAIRS>
__global__ __launch_bounds__(128) void rap_rows_shell_rows(const real_t *__restrict__ so, const real_t *__restrict__ ci,
                                                            real_t *__restrict__ T, size_t gstride, int NB,
                                                            int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int z0, int compact)
{
	const int c2x = blockIdx.x * blockDim.x + threadIdx.x + 1, c2z = blockIdx.z + z0;
	int c2y = blockIdx.y + 1;
	if (compact) {
		if (!rows_inner(c2z, KKF)) return;
		const int hi0 = (JJF - 3) / 2 + 2;
		c2y = (int)blockIdx.y < 2 ? 1 + (int)blockIdx.y : hi0 + ((int)blockIdx.y - 2);
		if (c2y > JJC - 1 || rows_inner(c2y, JJF)) return;
	}
	if (c2x > IIC - 1) return;
	if constexpr (PAIRS)
		row_groups<SEVEN, 1>(so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, c2x, c2y, c2z, std::make_integer_sequence<int, 25>());
	else
		row_sums<SEVEN, true>(so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, c2x, c2y, c2z, std::make_integer_sequence<int, 125>());
}

// shell, part 2: the few points next to an x face of every interior row, packed: lane -> (point e of the row, row)
template <bool SEVEN, bool PAIRS>
__global__ __launch_bounds__(128) void rap_rows_shell_x(const real_t *__restrict__ so, const real_t *__restrict__ ci,
                                                         real_t *__restrict__ T, size_t gstride, int NB,
                                                         int IIF, int JJF, int KKF, int IIC, int JJC, int KKC, int z0, int nz)
{
	const int hi0 = (IIF - 3) / 2 + 2;           // first c2x above the interior
	const int ne = 2 + (IIC - 1 - hi0 + 1);      // c2x = 1, 2 and hi0 .. IIC-1
	const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	const size_t total = (size_t)ne * (JJC - 1) * nz;
	if (t >= total) return;
	const int e = (int)(t % ne);
	const int c2y = (int)((t / ne) % (JJC - 1)) + 1, c2z = (int)(t / ((size_t)ne * (JJC - 1))) + z0;
	if (!(rows_inner(c2y, JJF) && rows_inner(c2z, KKF))) return; // whole rows: part 1
	const int c2x = e < 2 ? 1 + e : hi0 + (e - 2);
	if (c2x > IIC - 1 || rows_inner(c2x, IIF)) return;
	if constexpr (PAIRS)
		row_groups<SEVEN, 2>(so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, c2x, c2y, c2z, std::make_integer_sequence<int, 25>());
	else
		row_sums<SEVEN, true>(so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, c2x, c2y, c2z, std::make_integer_sequence<int, 125>());
}

// stage 2: slot S of the coarse point (ic,jc,kc):  +- sum_d P(f1,c1) row(f1,c2), d in the order of rap_slot
template <int S>
__device__ __forceinline__ real_t rows_contract(const real_t *__restrict__ T, size_t gstride, int NB,
                                                const real_t *__restrict__ ci, int IIC, int JJC, int KKC, int ic, int jc, int kc)
{
	constexpr g3::V3 a = g3::A[S], b = g3::B[S];
	const int c1x = ic + a.x, c1y = jc + a.y, c1z = kc + a.z;
	const int c2x = ic + b.x, c2y = jc + b.y, c2z = kc + b.z;
	real_t acc = 0.0;
#pragma unroll
	for (int dz = -1; dz <= 1; dz++)
#pragma unroll
		for (int dy = -1; dy <= 1; dy++)
#pragma unroll
			for (int dx = -1; dx <= 1; dx++) {
				const int gx = 2 * (a.x - b.x) + dx, gy = 2 * (a.y - b.y) + dy, gz = 2 * (a.z - b.z) + dz;
				if (gx < -2 || gx > 2 || gy < -2 || gy > 2 || gz < -2 || gz > 2) continue; // N(f1) misses N(c2)
				const real_t row = TR((gz + 2) * 25 + (gy + 2) * 5 + (gx + 2), c2x, c2y, c2z);
				const g3::PWE w = g3::PW[dz + 1][dy + 1][dx + 1];
				const real_t p1 = w.slot < 0 ? 1.0 : CI3(c1x + w.ox, c1y + w.oy, c1z + w.oz, w.slot);
				acc += p1 * row;
			}
	return (S == KP) ? acc : -acc;
}

// coarse planes kc0 .. kc0+gridDim.z-1
__global__ __launch_bounds__(128) void rap_rows_stage2(const real_t *__restrict__ T, size_t gstride, int NB,
                                                        real_t *__restrict__ soc, const real_t *__restrict__ ci,
                                                        int IIC, int JJC, int KKC, int kc0)
{
	const int ic = blockIdx.x * blockDim.x + threadIdx.x + 2, jc = blockIdx.y + 2, kc = blockIdx.z + kc0;
	if (ic > IIC - 1) return;
	const size_t c = (size_t)(ic - 1) + (size_t)IIC * ((size_t)(jc - 1) + (size_t)JJC * (size_t)(kc - 1));
	const size_t PC = (size_t)IIC * JJC * KKC;
#define SLOT(Sv) soc[c + PC * (size_t)(Sv)] = rows_contract<Sv>(T, gstride, NB, ci, IIC, JJC, KKC, ic, jc, kc);
	SLOT(0) SLOT(1) SLOT(2) SLOT(3) SLOT(4) SLOT(5) SLOT(6) SLOT(7) SLOT(8) SLOT(9) SLOT(10) SLOT(11) SLOT(12) SLOT(13)
#undef SLOT
}
#undef TR

constexpr bool rows_b_z_is_0_or_minus_1()
{
	for (int s = 0; s < 14; s++)
		if (g3::B[s].z != 0 && g3::B[s].z != -1) return false;
	return true;
}

// the ring is kept by the library while it is at most 6 GB (4.4 GB for slabs of 32 planes at 512^3); a larger one is
// released after the product (allocating and releasing 17 GB per product costs more than the larger slab gains)
static real_t *g_rows = nullptr;
static size_t g_rows_bytes = 0;

// pair starts are the odd fine indices of every row of every slot: even row length, 16-byte aligned base
// (CEDAR_AMD_GALERKIN_PAIRS=0: 8-byte loads, for A/B runs)
bool galerkin3_rows_pairs(const real_t *so, int IIF)
{
	const char *ep = getenv("CEDAR_AMD_GALERKIN_PAIRS");
	return (!ep || atoi(ep) != 0) && IIF % 2 == 0 && ((uintptr_t)so & 15) == 0;
}

// cedar_amd_release_scratch: the caller has drained the device
void galerkin3_rows_release()
{
	if (g_rows) (void)hipFree(g_rows);
	g_rows = nullptr, g_rows_bytes = 0;
}

// returns false when the ring cannot be had (the caller then runs the one-stage kernels)
bool galerkin3_rows(const real_t *so, real_t *soc, const real_t *ci, int IIF, int JJF, int KKF,
                    int IIC, int JJC, int KKC, int ifd, hipStream_t st)
{
	static_assert(rows_b_z_is_0_or_minus_1(), "a slab of coarse planes kc0 .. reads the row sums of the planes kc0-1 ..");
	if (IIC < 3 || JJC < 3 || KKC < 3) return false;
	const char *e = getenv("CEDAR_AMD_GALERKIN_SLAB");
	int S = (e && atoi(e) > 0) ? atoi(e) : 32;
	const int ncz = KKC - 2;
	if (S > ncz) S = ncz;
	const int NB = S + 1; // planes of c2 alive at once: kc0-1 .. kc0+S-1
	const size_t gstride = (size_t)IIC * JJC * NB;
	const size_t need = gstride * 125 * sizeof(real_t);
	if (need > g_rows_bytes) {
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) != hipSuccess || need + tot / 20 > fr + g_rows_bytes) return false;
		CEDAR_HIP_CHECK(hipStreamSynchronize(st));
		if (g_rows) CEDAR_HIP_CHECK(hipFree(g_rows));
		g_rows = nullptr, g_rows_bytes = 0;
		if (hipMalloc((void **)&g_rows, need) != hipSuccess) {
			(void)hipGetLastError();
			return false;
		}
		g_rows_bytes = need;
	}
	real_t *T = g_rows;
	const dim3 blk(128);
	const bool pairs = galerkin3_rows_pairs(so, IIF);
	int done = 0; // row sums exist for the c2 planes 1 .. done
	for (int kc0 = 2; kc0 <= KKC - 1; kc0 += S) {
		const int ns = (kc0 + S - 1 <= KKC - 1) ? S : KKC - kc0;
		const int zhi = kc0 + ns - 1; // c2 planes needed: kc0-1 .. zhi
		const int z0 = done + 1 > kc0 - 1 ? done + 1 : kc0 - 1;
		if (zhi >= z0) {
			const dim3 g1((IIC - 1 + 127) / 128, JJC - 1, zhi - z0 + 1);
			const int nzs = zhi - z0 + 1;
			const int ne = 2 + (IIC - 1 - ((IIF - 3) / 2 + 2) + 1);
			const dim3 gx((unsigned)(((size_t)(ne > 0 ? ne : 1) * (JJC - 1) * nzs + 127) / 128));
			const int ney = 2 + (JJC - 1 - ((JJF - 3) / 2 + 2) + 1);
			const dim3 gy((IIC - 1 + 127) / 128, ney > 0 ? ney : 1, nzs);
			// planes of this slab next to a z face: c2z = 1, 2 at the bottom, hiz .. KKC-1 at the top
			const int hiz = (KKF - 3) / 2 + 2;
			const int lo_a = z0, lo_b = zhi < 2 ? zhi : 2;                 // [lo_a, lo_b] if z0 <= 2
			const int hi_a = z0 > hiz ? z0 : hiz, hi_b = zhi;              // [hi_a, hi_b] if zhi >= hiz
#define STAGE1(SV, PV)                                                                                                              \
	hipLaunchKernelGGL((rap_rows_interior<SV, PV>), g1, blk, 0, st, so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, z0);        \
	hipLaunchKernelGGL((rap_rows_shell_rows<SV, PV>), gy, blk, 0, st, so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, z0, 1);   \
	if (lo_a <= lo_b)                                                                                                               \
		hipLaunchKernelGGL((rap_rows_shell_rows<SV, PV>), dim3(g1.x, g1.y, lo_b - lo_a + 1), blk, 0, st, so, ci, T, gstride, NB,      \
		                   IIF, JJF, KKF, IIC, JJC, KKC, lo_a, 0);                                                                  \
	if (hi_a <= hi_b && hi_a > lo_b)                                                                                                \
		hipLaunchKernelGGL((rap_rows_shell_rows<SV, PV>), dim3(g1.x, g1.y, hi_b - hi_a + 1), blk, 0, st, so, ci, T, gstride, NB,      \
		                   IIF, JJF, KKF, IIC, JJC, KKC, hi_a, 0);                                                                  \
	hipLaunchKernelGGL((rap_rows_shell_x<SV, PV>), gx, blk, 0, st, so, ci, T, gstride, NB, IIF, JJF, KKF, IIC, JJC, KKC, z0, nzs);
			if (ifd == 1) {
				if (pairs) { STAGE1(true, true) } else { STAGE1(true, false) }
			} else {
				if (pairs) { STAGE1(false, true) } else { STAGE1(false, false) }
			}
#undef STAGE1
			done = zhi;
		}
		const dim3 g2((IIC - 2 + 127) / 128, JJC - 2, ns);
		hipLaunchKernelGGL(rap_rows_stage2, g2, blk, 0, st, T, gstride, NB, soc, ci, IIC, JJC, KKC, kc0);
	}
	if (g_rows_bytes > ((size_t)6 << 30)) {
		CEDAR_HIP_CHECK(hipStreamSynchronize(st));
		CEDAR_HIP_CHECK(hipFree(g_rows));
		g_rows = nullptr, g_rows_bytes = 0;
	}
	return true;
}

} // namespace cedar_amd
