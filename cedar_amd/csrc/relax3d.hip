// 3D point relaxation: 27-point 8-colour and 7-point red-black Gauss-Seidel.
// Replaces BMG3_SymStd_relax_GS / BMG3_SymStd_SETUP_recip
// (reference src/3d/ftn/BMG3_SymStd_relax_GS.f90:80-187, SETUP_recip.f90:63-69).
//
// 27-point design (HBM-bound, 136 algorithmic B/DOF):
//   The eight colours are (i,j,k) parities swept in the order i fastest.  Two
//   consecutive colours differ only in i-parity and couple only through
//   i+-1 neighbours *of the same grid row*, so one workgroup that owns a whole
//   row (j,k) can relax colour 2c+1 and then colour 2c+2 of that row without
//   any other workgroup's data: 4 launches per sweep instead of 8, every row
//   stream is read with unit stride (no stride-2 half-line waste), and the
//   row is written once.  Each lane owns the adjacent pair (i_e, i_o) =
//   (2p+2, 2p+3); the freshly relaxed first-colour values travel to the
//   neighbouring lane through a 4 KB LDS row.
//   Rows are dealt to XCDs in contiguous k-slabs so that the q rows shared by
//   neighbouring workgroups (j+-1, k+-1) are re-read from that XCD's L2.
//
// Arithmetic: the 26 products are added in the reference's order and the file
// is compiled with -ffp-contract=off: results are bit-identical to the
// reference's CPU build.
#include "common.h"
#include "relax27_dev.h"
#include "relax3_psum.h"
#include <map>

namespace cedar_amd {

static TileShape env_tile(const char *name, unsigned tjl, unsigned tkl)
{
	TileShape ts{tjl, tkl};
	if (const char *e = getenv(name)) {
		unsigned a, b;
		if (sscanf(e, "%u,%u", &a, &b) == 2 && a <= 8 && b <= 8) { ts.tjl = a; ts.tkl = b; }
	}
	return ts;
}
TileShape tile_shape_relax() { static TileShape ts = env_tile("CEDAR_AMD_TILE_RELAX", 4, 4); return ts; }
TileShape tile_shape_resid() { return env_tile("CEDAR_AMD_TILE_RESID", 4, 4); } // read per call (interleaved A/B runs)

// ------------------------------------------------------------------ recip
__global__ void recip_kernel(const real_t *__restrict__ d, real_t *__restrict__ r,
                             int II, int JJ, int KK)
{
	size_t n = (size_t)II * JJ * KK;
	for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n;
	     idx += (size_t)gridDim.x * blockDim.x) {
		int i = (int)(idx % II);
		size_t t = idx / II;
		int j = (int)(t % JJ);
		int k = (int)(t / JJ);
		bool in = i >= 1 && i <= II - 2 && j >= 1 && j <= JJ - 2 && (KK == 1 || (k >= 1 && k <= KK - 2));
		if (in) r[idx] = 1.0 / d[idx];
	}
}

void setup_recip(const real_t *so_diag, real_t *sor_msor, size_t II, size_t JJ, size_t KK, hipStream_t st)
{
	size_t n = II * JJ * KK;
	unsigned grid = (unsigned)((n + 255) / 256);
	if (grid > 8192) grid = 8192;
	hipLaunchKernelGGL(recip_kernel, dim3(grid), dim3(256), 0, st, so_diag, sor_msor, (int)II, (int)JJ, (int)KK);
}

// row-interleaved solve copy of a 27-point operator (common.h Op3): one workgroup per grid row, unit-stride both ways
__global__ __launch_bounds__(256) void ilv_build_kernel(const real_t *__restrict__ so, const real_t *__restrict__ sor,
                                                         real_t *__restrict__ out, int II, int JJ, int KK, size_t RS)
{
	const size_t row = blockIdx.x; // j + JJ*k
	const size_t PS = (size_t)II * JJ * KK;
	const real_t *src = so + row * (size_t)II;
	real_t *dst = out + row * (size_t)NS3 * RS;
	for (int s = 0; s < NS3; s++) {
		const real_t *from = s < 14 ? src + (size_t)s * PS : (s == ILV_SOR ? sor + row * (size_t)II : nullptr);
		for (size_t i = threadIdx.x; i < RS; i += 256)
			dst[(size_t)s * RS + i] = (from && i < (size_t)II) ? from[i] : 0.0;
	}
}

void ilv_build(const real_t *so, const real_t *sor_msor, real_t *ilv, int II, int JJ, int KK, hipStream_t st)
{
	hipLaunchKernelGGL(ilv_build_kernel, dim3((unsigned)((size_t)JJ * KK)), dim3(256), 0, st, so, sor_msor, ilv, II, JJ, KK,
	                   ilv_row_stride(II));
}

// Solve copies registered for operators that live outside a resident solver (the per-rank arrays of the
// domain-decomposed solver, cedar_amd/dist.py): the per-piece entry points below look the operator pointer up and read
// the row-interleaved copy when there is one.  The caller re-registers after it changed the operator.
// T: partial-sum scratch of the sweep (relax3d_psum.hip), one vector; frun: its run length on this operator (0: none)
struct IlvReg { real_t *ilv, *T; int II, JJ, KK, frun; };
static std::map<const real_t *, IlvReg> g_ilv;

// psum_min_rows > 0: the partial-sum sweep from that many rows on (relax3_prepare_rows: the distributed driver on rank grids
// with an x / y split, where the alternative is not four row-class launches but four passes with an exchange after each)
static int prepare_impl(const real_t *so, const real_t *sor, int II, int JJ, int KK, int min_rows, int psum_min_rows, hipStream_t st)
{
	relax3_release(so);
	size_t fr = 0, tot = 0;
	const size_t bytes = ilv_doubles(II, JJ, KK) * sizeof(real_t);
	const bool want_ilv = min_rows >= 0 && JJ - 2 >= min_rows && (II - 2 + 1) / 2 <= 512 && hipMemGetInfo(&fr, &tot) == hipSuccess
	                      && bytes + tot / 10 < fr;
	int frun = relax3_psum_frun(JJ);
	const char *ep = getenv("CEDAR_AMD_PSUM"), *ef = getenv("CEDAR_AMD_FRUN");
	if (!frun && psum_min_rows > 0 && JJ - 2 >= psum_min_rows && !ef) frun = 8;
	const bool want_T = !(ep && atoi(ep) == 0) && relax3_psum_ok(II, JJ, KK, frun);
	if (!want_ilv && !want_T) return 0;
	IlvReg r{nullptr, nullptr, II, JJ, KK, want_T ? frun : 0};
	if (want_ilv) {
		CEDAR_HIP_CHECK(hipMalloc((void **)&r.ilv, bytes));
		ilv_build(so, sor + (size_t)II * JJ * KK, r.ilv, II, JJ, KK, st);
	}
	if (want_T) CEDAR_HIP_CHECK(hipMalloc((void **)&r.T, (size_t)II * JJ * KK * sizeof(real_t)));
	g_ilv[so] = r;
	return (want_ilv ? 1 : 0) | (want_T ? 2 : 0);
}

int relax3_prepare(const real_t *so, const real_t *sor, int II, int JJ, int KK, int min_rows, hipStream_t st)
{
	return prepare_impl(so, sor, II, JJ, KK, min_rows, 0, st);
}

int relax3_prepare_rows(const real_t *so, const real_t *sor, int II, int JJ, int KK, int min_rows, int psum_min_rows, hipStream_t st)
{
	return prepare_impl(so, sor, II, JJ, KK, min_rows, psum_min_rows, st);
}

void relax3_release(const real_t *so)
{
	auto it = g_ilv.find(so);
	if (it == g_ilv.end()) return;
	CEDAR_HIP_CHECK(hipDeviceSynchronize());
	(void)hipFree(it->second.ilv);
	(void)hipFree(it->second.T);
	g_ilv.erase(it);
}

static const IlvReg *reg_lookup(const real_t *so, int II, int JJ, int KK)
{
	if (g_ilv.empty()) return nullptr;
	auto it = g_ilv.find(so);
	if (it != g_ilv.end() && it->second.II == II && it->second.JJ == JJ && it->second.KK == KK) return &it->second;
	return nullptr;
}

static Op3 op3_lookup(const real_t *so, const real_t *sor, int II, int JJ, int KK)
{
	const IlvReg *r = reg_lookup(so, II, JJ, KK);
	if (r && r->ilv) return op3_ilv(r->ilv, II, JJ, KK);
	return op3_cedar(so, sor, II, JJ, KK);
}

// direct-from-memory evaluation at one point (generic path; x = vector offset of (i,j,k), xa = operator offset)
__device__ __forceinline__ real_t offdiag27_mem(const Op3 &A, const real_t *__restrict__ qf,
                                                const real_t *q, size_t II, size_t JJ, // (q: relax27_cols reads what it wrote)
                                                size_t x, size_t xa)
{
	const size_t sj = II, sk = II * JJ;
	const size_t PS = A.SS, aj = A.SJ, ak = A.SK;
	const real_t *__restrict__ so = A.so;
	C27 c;
	c.pw = so[KPW * PS + xa]; c.ps = so[KPS * PS + xa]; c.psw = so[KPSW * PS + xa];
	c.b = so[KB * PS + xa]; c.bw = so[KBW * PS + xa]; c.bs = so[KBS * PS + xa]; c.bsw = so[KBSW * PS + xa];
	c.pnw_n = so[KPNW * PS + xa + aj]; c.ps_n = so[KPS * PS + xa + aj];
	c.bnw_n = so[KBNW * PS + xa + aj]; c.bn_n = so[KBN * PS + xa + aj];
	c.b_t = so[KB * PS + xa + ak]; c.be_t = so[KBE * PS + xa + ak];
	c.bn_t = so[KBN * PS + xa + ak]; c.bne_t = so[KBNE * PS + xa + ak];
	c.bse_nt = so[KBSE * PS + xa + aj + ak]; c.bs_nt = so[KBS * PS + xa + aj + ak];
	c.psw_ne = so[KPSW * PS + xa + 1 + aj]; c.bne_ne = so[KBNE * PS + xa + 1 + aj];
	c.pw_e = so[KPW * PS + xa + 1]; c.pnw_e = so[KPNW * PS + xa + 1];
	c.be_e = so[KBE * PS + xa + 1]; c.bse_e = so[KBSE * PS + xa + 1];
	c.bsw_net = so[KBSW * PS + xa + 1 + aj + ak];
	c.bw_et = so[KBW * PS + xa + 1 + ak]; c.bnw_et = so[KBNW * PS + xa + 1 + ak];
	real_t qq[3][3][3];
#pragma unroll
	for (int dk = 0; dk < 3; dk++)
#pragma unroll
		for (int dj = 0; dj < 3; dj++)
#pragma unroll
			for (int di = 0; di < 3; di++)
				qq[dk][dj][di] = q[x + (di - 1) + (ptrdiff_t)(dj - 1) * (ptrdiff_t)sj + (ptrdiff_t)(dk - 1) * (ptrdiff_t)sk];
	return offdiag27(qf[x], c, qq);
}

// generic fallback: one colour per launch, one thread per colour point
__global__ void relax27_colour(const Op3 A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                               int II, int JJ, int KK, int ib, int jb, int kb)
{
	int ni = (II - 2 - ib + 1) / 2, nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
	size_t n = (size_t)ni * nj * nk;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
		int a = (int)(t % ni);
		size_t r = t / ni;
		int b = (int)(r % nj), c = (int)(r / nj);
		const size_t i = (size_t)(1 + ib + 2 * a), j = (size_t)(1 + jb + 2 * b), k = (size_t)(1 + kb + 2 * c);
		size_t x = i + (size_t)II * (j + (size_t)JJ * k);
		q[x] = offdiag27_mem(A, qf, q, II, JJ, x, i + j * A.SJ + k * A.SK) * A.sor[i + j * A.rSJ + k * A.rSK];
	}
}

// fast path: one workgroup = one grid row, both i-colours.  Rows j = j0 + jstep*jr, jr < nrj, of the
// planes k = 1 + kb + 2*(kr + kr0), kr < nrk.
template <int BS, bool EFIRST, bool NT, bool PERX = false>
__global__ __launch_bounds__(BS) void relax27_rows(const Op3 A, const real_t *__restrict__ qf,
                                                    real_t *__restrict__ q,
                                                    int II, int JJ, int KK, int j0, int jstep, int kb, int nrj, int nrk,
                                                    TileShape ts, int kr0)
{
	__shared__ real_t xch[BS + 2];
	const unsigned nblk = tile_blocks((unsigned)nrj, (unsigned)nrk, ts);
	const unsigned L = xcd_remap(blockIdx.x, nblk);
	unsigned jr, kr;
	if (L >= nblk || !tile_rows(L, (unsigned)nrj, (unsigned)nrk, ts, jr, kr)) return; // whole workgroup leaves together
	const size_t j = (size_t)(j0 + jstep * (int)jr), k = (size_t)(1 + kb + 2 * ((int)kr + kr0)); // 0-based incl. ghost
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	relax27_row_task<BS, EFIRST, NT, NT, 0, PERX>(A, qf, q, II, sj, sk, j, k, xch);
}

// The shell of a row class (rows next to a face shared with another rank) is up to four thin rectangles of
// (row, plane) pairs: one launch over their union instead of one latency-bound launch each.
struct ShellRects {
	int n;
	int j0[4], nrj[4], kr0[4], start[5]; // rows j0 + 2 jr (jr < nrj) of the planes kr0 + kr; start = first workgroup of a rectangle
};

template <int BS, bool EFIRST, bool NT>
__global__ __launch_bounds__(BS) void relax27_rows_shell(const Op3 A, const real_t *__restrict__ qf,
                                                          real_t *__restrict__ q,
                                                          int II, int JJ, int KK, int kb, ShellRects rc)
{
	__shared__ real_t xch[BS + 2];
	const int b = (int)blockIdx.x;
	int r = 0;
#pragma unroll
	for (int t = 1; t < 4; t++)
		if (t < rc.n && b >= rc.start[t]) r = t;
	const int w = b - rc.start[r];
	const size_t j = (size_t)(rc.j0[r] + 2 * (w % rc.nrj[r])), k = (size_t)(1 + kb + 2 * (rc.kr0[r] + w / rc.nrj[r]));
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	relax27_row_task<BS, EFIRST, NT>(A, qf, q, II, sj, sk, j, k, xch);
}

// Plane-fused pass: both row classes of the planes of one k-parity in ONE launch.  In a plane the
// sweep relaxes the rows of class F (j-parity jbF) before those of class S; an S row reads the fresh
// values of its two F neighbours j-1, j+1 and nothing else of this plane class changes under it.  A
// workgroup owns a run of consecutive F rows [f0, f1) of one plane and walks it in the order
//   F(f0), F(f0+1), S between them, F(f0+2), S, ...
// so that the operator rows both classes need (kps, kpsw, kpnw of the plane: 3 of the 25 slot-rows a
// row task reads) and the nine q rows are re-used from L2 a few microseconds after their first use
// instead of being streamed again by a second launch.  The S row between two runs has its F
// neighbours in different workgroups: it is left to a small second launch (relax27_rows over those
// rows, j0 = first such row, jstep = 2*frun).  Same arithmetic per point, same values read =>
// results identical to the four-launch order.
template <int BS, bool EFIRST, bool NT, int WI = 0>
__global__ __launch_bounds__(BS) void relax27_plane(const Op3 A, const real_t *__restrict__ qf,
                                                     real_t *__restrict__ q,
                                                     int II, int JJ, int KK, int jbF, int kb, int nrk, int frun, int nrun,
                                                     int kr0)
{
	__shared__ real_t xch[2][BS + 2];
	const unsigned nblk = (unsigned)nrk * (unsigned)nrun;
	const unsigned L = xcd_remap(blockIdx.x, nblk);
	if (L >= nblk) return;
	const int kr = (int)(L / (unsigned)nrun), run = (int)(L % (unsigned)nrun);
	const int nF = (JJ - 2 - jbF + 1) / 2, nS = (JJ - 2 - (1 - jbF) + 1) / 2;
	const int f0 = run * frun, f1 = min(nF, f0 + frun);
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	const size_t k = (size_t)(1 + kb + 2 * (kr + kr0)); // planes kr0 .. kr0+nrk-1 of parity kb
	int t = 0;
	for (int f = f0; f < f1; f++) {
		// F row f: j = 1 + jbF + 2 f
		relax27_row_task<BS, EFIRST, NT, false, WI>(A, qf, q, II, sj, sk, (size_t)(1 + jbF + 2 * f), k, xch[t & 1]);
		t++;
		// the S row both of whose F neighbours are now done (a missing neighbour = ghost row):
		//   jbF = 0: S row g (j = 2+2g) lies between F rows g, g+1  -> after F(f): g = f-1 (f > f0)
		//   jbF = 1: S row g (j = 1+2g) lies between F rows g-1, g  -> after F(f): g = f   (f > f0, or f = 0)
		const int g = jbF ? f : f - 1;
		const bool have = jbF ? (f > f0 || f == 0) : (f > f0);
		if (have && g >= 0 && g < nS) {
			__syncthreads(); // F stores of every wave visible before the S loads
			relax27_row_task<BS, EFIRST, NT, false, WI>(A, qf, q, II, sj, sk, (size_t)(2 - jbF + 2 * g), k, xch[t & 1]);
			t++;
		}
	}
	// the S row beyond the last F row of the plane (its other neighbour is the ghost row)
	if (f1 == nF && f1 > f0) {
		const int g = jbF ? nF : nF - 1;
		if (g < nS) {
			// jbF = 0: g = nF-1 was not reached in the loop (needs F(nF) which does not exist)
			// jbF = 1: g = nF    likewise
			__syncthreads();
			relax27_row_task<BS, EFIRST, NT, false, WI>(A, qf, q, II, sj, sk, (size_t)(2 - jbF + 2 * g), k, xch[t & 1]);
			t++;
		}
	}
	if (WI & 8) {
		// what-if "k-pair pass" (timing / traffic only, the dependencies between workgroups are ignored): the rows
		// of this run in the plane below right after the run -- the nine inter-plane slot-rows both planes read are
		// then at most one run of row tasks apart
		for (int r = 2 * f0; r < 2 * f1 && r < JJ - 2; r++) {
			__syncthreads();
			relax27_row_task<BS, EFIRST, NT, false, (WI & 7)>(A, qf, q, II, sj, sk, (size_t)(1 + r), k - 1, xch[t & 1]);
			t++;
		}
	}
}

// ------------------------------------------------------------------ 7-pt
// src/3d/ftn/BMG3_SymStd_relax_GS.f90:155-184: colour = mod(i+j+k+pts,2) in 1-based indices
__global__ void relax7_colour(const real_t *__restrict__ so, const real_t *__restrict__ qf,
                              real_t *__restrict__ q, const real_t *__restrict__ sor,
                              int II, int JJ, int KK, int pts)
{
	const int nxh = (II - 2 + 1) / 2; // max points of one colour in a row
	size_t n = (size_t)nxh * (JJ - 2) * (KK - 2);
	const size_t sj = II, sk = (size_t)II * JJ, PS = sk * KK;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
		int a = (int)(t % nxh);
		size_t r = t / nxh;
		int j1 = 2 + (int)(r % (JJ - 2)), k1 = 2 + (int)(r / (JJ - 2)); // 1-based
		int i1 = (j1 + k1 + pts) % 2 + 2 + 2 * a;
		if (i1 > II - 1) continue;
		size_t x = (size_t)(i1 - 1) + sj * (size_t)(j1 - 1) + sk * (size_t)(k1 - 1);
		real_t s = qf[x];
		s = s + so[KPW * PS + x] * q[x - 1];
		s = s + so[KPS * PS + x + sj] * q[x + sj];
		s = s + so[KPW * PS + x + 1] * q[x + 1];
		s = s + so[KPS * PS + x] * q[x - sj];
		s = s + so[KB * PS + x] * q[x - sk];
		s = s + so[KB * PS + x + sk] * q[x + sk];
		q[x] = s * sor[PS + x];
	}
}

// 27-point residual with the same lane layout as the relax row kernel: lane p owns the pair
// (2p+1, 2p+2) of its row, every stream is read with 16-byte loads, one 16-byte store.
// (BMG3_SymStd_residual.f90:77-104; bit-identical term order.)
template <int BS, bool NT>
__global__ __launch_bounds__(BS) void residual27_rows(const Op3 A, const real_t *__restrict__ qf,
                                                       const real_t *__restrict__ q, real_t *__restrict__ res,
                                                       int II, int JJ, int KK, unsigned nblk, TileShape ts)
{
	const unsigned L = xcd_remap(blockIdx.x, nblk);
	unsigned jr, kr;
	if (L >= nblk || !tile_rows(L, (unsigned)(JJ - 2), (unsigned)(KK - 2), ts, jr, kr)) return;
	const size_t j = (size_t)jr + 1, k = (size_t)kr + 1;
	const size_t sj = (size_t)II, sk = (size_t)II * JJ;
	const size_t row = j * sj + k * sk, rowA = j * A.SJ + k * A.SK;
	for (int p = threadIdx.x; 2 * p + 1 <= II - 2; p += BS) {
		const int ie = 2 * p + 1, io = 2 * p + 2;
		const bool o_ok = io <= II - 2, two = io + 1 <= II - 1;
		C27 ce, co;
		real_t qe[3][3][3], qo[3][3][3], qfe, qfo, de, dn;
		// NT: the own row's operator entries are streamed (last use in this launch), the rows at j+1 / k+1
		// stay cacheable for the neighbouring task that reads them as its own
		load_pair27<false, false, NT>(A, qf, q, rowA, row, sj, sk, ie, io, two, ce, co, qe, qo, qfe, qfo);
		ldpair_so<NT>(A.so + rowA + ie, true, de, dn); // KP plane
		const real_t re = offdiag27(qfe, ce, qe) - de * qe[1][1][1];
		if (o_ok) {
			const real_t ro = offdiag27(qfo, co, qo) - dn * qo[1][1][1];
			d2u v; v.x = re; v.y = ro;
			*reinterpret_cast<d2u *>(res + row + ie) = v;
		} else
			res[row + ie] = re;
	}
}

void residual27_op(const Op3 &A, const real_t *qf, const real_t *q, real_t *res, int II, int JJ, int KK, hipStream_t st)
{
	const Op3 so = A;
	const TileShape ts = tile_shape_resid();
	unsigned nrows = tile_blocks((unsigned)(JJ - 2), (unsigned)(KK - 2), ts);
	const int npairs = (II - 2 + 1) / 2;
	const char *e = getenv("CEDAR_AMD_RESID_NT");
	const bool nt = e ? atoi(e) != 0 : false;
	if (npairs <= 64) hipLaunchKernelGGL((residual27_rows<64, false>), dim3(xcd_grid(nrows)), dim3(64), 0, st, so, qf, q, res, II, JJ, KK, nrows, ts);
	else if (npairs <= 128) hipLaunchKernelGGL((residual27_rows<128, false>), dim3(xcd_grid(nrows)), dim3(128), 0, st, so, qf, q, res, II, JJ, KK, nrows, ts);
	else if (nt) hipLaunchKernelGGL((residual27_rows<256, true>), dim3(xcd_grid(nrows)), dim3(256), 0, st, so, qf, q, res, II, JJ, KK, nrows, ts);
	else hipLaunchKernelGGL((residual27_rows<256, false>), dim3(xcd_grid(nrows)), dim3(256), 0, st, so, qf, q, res, II, JJ, KK, nrows, ts);
}

void residual27_fast(const real_t *so, const real_t *qf, const real_t *q, real_t *res, int II, int JJ, int KK, hipStream_t st)
{
	residual27_op(op3_lookup(so, nullptr, II, JJ, KK), qf, q, res, II, JJ, KK, st);
}

static inline unsigned cap_grid(size_t n, unsigned bs)
{
	size_t g = (n + bs - 1) / bs;
	if (g > 16384) g = 16384;
	if (g < 1) g = 1;
	return (unsigned)g;
}

// rows j = j0 + jstep*jr (jr < nrj) of the planes kr0 .. kr0+nrk-1 (in units of planes of parity kb)
template <int BS>
static void launch_rows_at(bool efirst, const Op3 &A, const real_t *qf, real_t *q,
                           int II, int JJ, int KK, int j0, int jstep, int nrj, int kb, int nrk, hipStream_t st, int kr0 = 0)
{
	if (nrj <= 0 || nrk <= 0) return;
	const TileShape ts = tile_shape_relax();
	unsigned grid = xcd_grid(tile_blocks((unsigned)nrj, (unsigned)nrk, ts));
	// non-temporal operator loads: measured -1.8 % per launch at 512^3 (profiles/r01_experiment_nt_loads.log)
	static const bool nt = getenv("CEDAR_AMD_NT") ? atoi(getenv("CEDAR_AMD_NT")) != 0 : true;
	if (efirst) {
		if (nt) hipLaunchKernelGGL((relax27_rows<BS, true, true>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, j0, jstep, kb, nrj, nrk, ts, kr0);
		else hipLaunchKernelGGL((relax27_rows<BS, true, false>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, j0, jstep, kb, nrj, nrk, ts, kr0);
	} else {
		if (nt) hipLaunchKernelGGL((relax27_rows<BS, false, true>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, j0, jstep, kb, nrj, nrk, ts, kr0);
		else hipLaunchKernelGGL((relax27_rows<BS, false, false>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, j0, jstep, kb, nrj, nrk, ts, kr0);
	}
}

// the rows of class (jb,kb), periodic in x (even number of points per row)
template <int BS>
static void launch_rows_perx(bool efirst, const Op3 &A, const real_t *qf, real_t *q, int II, int JJ, int KK, int jb, int kb,
                             hipStream_t st)
{
	const int nrj = (JJ - 2 - jb + 1) / 2, nrk = (KK - 2 - kb + 1) / 2;
	if (nrj <= 0 || nrk <= 0) return;
	const TileShape ts = tile_shape_relax();
	const unsigned grid = xcd_grid(tile_blocks((unsigned)nrj, (unsigned)nrk, ts));
	if (efirst) hipLaunchKernelGGL((relax27_rows<BS, true, true, true>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, 1 + jb, 2, kb, nrj, nrk, ts, 0);
	else hipLaunchKernelGGL((relax27_rows<BS, false, true, true>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, 1 + jb, 2, kb, nrj, nrk, ts, 0);
}

// the rows of class (jb,kb)
template <int BS>
static void launch_rows(bool efirst, const Op3 &A, const real_t *qf, real_t *q,
                        int II, int JJ, int KK, int jb, int kb, hipStream_t st)
{
	launch_rows_at<BS>(efirst, A, qf, q, II, JJ, KK, 1 + jb, 2, (JJ - 2 - jb + 1) / 2, kb, (KK - 2 - kb + 1) / 2, st);
}

// plane-fused pass over the planes of parity kb: F rows (parity jbF) and the S rows between them in
// one launch, the S rows between two workgroups' runs in a second small one (see relax27_plane)
template <int BS>
static void launch_plane(bool efirst, const Op3 &A, const real_t *qf, real_t *q,
                         int II, int JJ, int KK, int jbF, int kb, int frun, hipStream_t st, int kr0 = 0, int nrk_sub = -1)
{
	const int nF = (JJ - 2 - jbF + 1) / 2;
	const int nrk = nrk_sub >= 0 ? nrk_sub : (KK - 2 - kb + 1) / 2;
	if (nrk <= 0) return;
	const int nrun = (nF + frun - 1) / frun;
	static const bool nt = getenv("CEDAR_AMD_NT") ? atoi(getenv("CEDAR_AMD_NT")) != 0 : true;
	const unsigned grid = xcd_grid((unsigned)nrk * (unsigned)nrun);
	if (BS == 256) { // experiments: CEDAR_AMD_WHATIF=1..7 (see load_pair27), plane-fused launch only
		const char *ew = getenv("CEDAR_AMD_WHATIF");
		const int wi = ew ? atoi(ew) : 0;
#define WI_CASE(W)                                                                                                          \
	case W:                                                                                                                 \
		if (efirst) hipLaunchKernelGGL((relax27_plane<256, true, true, W>), dim3(grid), dim3(256), 0, st, A, qf, q, II, JJ, KK, jbF, kb, nrk, frun, nrun, kr0); \
		else hipLaunchKernelGGL((relax27_plane<256, false, true, W>), dim3(grid), dim3(256), 0, st, A, qf, q, II, JJ, KK, jbF, kb, nrk, frun, nrun, kr0);      \
		break;
		if (wi) {
			switch (wi) { WI_CASE(1) WI_CASE(2) WI_CASE(3) WI_CASE(4) WI_CASE(5) WI_CASE(8) default: break; }
			launch_rows_at<BS>(efirst, A, qf, q, II, JJ, KK, (jbF ? 1 : 0) + 2 * frun, 2 * frun, nrun - 1, kb, nrk, st, kr0);
			return;
		}
#undef WI_CASE
	}
	if (efirst) {
		if (nt) hipLaunchKernelGGL((relax27_plane<BS, true, true>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, jbF, kb, nrk, frun, nrun, kr0);
		else hipLaunchKernelGGL((relax27_plane<BS, true, false>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, jbF, kb, nrk, frun, nrun, kr0);
	} else {
		if (nt) hipLaunchKernelGGL((relax27_plane<BS, false, true>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, jbF, kb, nrk, frun, nrun, kr0);
		else hipLaunchKernelGGL((relax27_plane<BS, false, false>), dim3(grid), dim3(BS), 0, st, A, qf, q, II, JJ, KK, jbF, kb, nrk, frun, nrun, kr0);
	}
	// S rows between runs: jbF = 0: j = 2 frun (r+1); jbF = 1: j = 1 + 2 frun (r+1), r = 0 .. nrun-2
	launch_rows_at<BS>(efirst, A, qf, q, II, JJ, KK, (jbF ? 1 : 0) + 2 * frun, 2 * frun, nrun - 1, kb, nrk, st, kr0);
}

// F rows per workgroup of the plane-fused pass; 0 = four launches per sweep (one per row class).
// Measured on MI355X (profiles/r01_experiment_plane_fused_relax.log): 512^3 -3..-7.5 % for runs of
// 8..64 rows, 448^3 -9 % and 384^3 -11 % at 8..16, 320^3 -5 % at 8, 256^3 and below +3..+11 % (the whole level sits closer to the
// Infinity Cache and the four-launch order already re-reads from it) => fused only on big levels.
// CEDAR_AMD_FRUN overrides (0 = never; n = runs of n rows wherever a plane has >= 4 runs).
int relax3_plane_frun(int JJ)
{
	const char *e = getenv("CEDAR_AMD_FRUN"); // read per call: the tests switch it between cases
	const int ny = JJ - 2;
	if (e) {
		const int frun = atoi(e);
		return (frun <= 0 || ny < 4 * frun) ? 0 : frun;
	}
	// runs of 8 rows are within 1 % of the best run length from 320 to 512; with the row-interleaved solve copy runs of 16
	// are another 1 % shorter at 384 and 512 (half as many rows between runs; interleaved A/B on two boxes,
	// profiles/r02_experiment_run_length.log)
	return ny >= 384 ? 16 : ny >= 320 ? 8 : 0;
}

// one row class (jb,kb) of the 27-point sweep, both i-colours (distributed runs exchange halos
// between row classes).  part: 0 = every row of the class; 1 = the rows none of whose neighbours is a
// ghost row (2 <= j <= ny-1, 2 <= k <= nz-1 in 1-based interior numbering 1..n); 2 = the others (the
// shell).  Rows of one class do not couple, so part 1 then part 2 equals part 0; the interior rows do
// not read the y/z ghost layers and can run while those are still being exchanged.
// sides: which faces of the box have a neighbouring rank (bit 0 -y, 1 +y, 2 -z, 3 +z); rows next to a face
// without one read no exchanged ghost and count as interior.
template <int BS>
static void launch_part(bool efirst, const Op3 &A, const real_t *qf, real_t *q,
                        int II, int JJ, int KK, int jb, int kb, int part, int sides, hipStream_t st)
{
	const int nrj = (JJ - 2 - jb + 1) / 2, nrk = (KK - 2 - kb + 1) / 2;
	if (nrj <= 0 || nrk <= 0) return;
	if (part == 0) {
		launch_rows<BS>(efirst, A, qf, q, II, JJ, KK, jb, kb, st);
		return;
	}
	// class rows j = 1+jb+2 jr (0-based incl. ghost): j = 1 is in the class iff jb = 0, j = ny iff it has the class parity
	const int jlo = (jb == 0 && (sides & 1)) ? 1 : 0, klo = (kb == 0 && (sides & 4)) ? 1 : 0;
	const int jhi = (1 + jb + 2 * (nrj - 1) == JJ - 2 && (sides & 2)) ? nrj - 1 : nrj;
	const int khi = (1 + kb + 2 * (nrk - 1) == KK - 2 && (sides & 8)) ? nrk - 1 : nrk;
	const int nji = jhi - jlo > 0 ? jhi - jlo : 0, nki = khi - klo > 0 ? khi - klo : 0;
	if (part == 1) {
		launch_rows_at<BS>(efirst, A, qf, q, II, JJ, KK, 1 + jb + 2 * jlo, 2, nji, kb, nki, st, klo);
		return;
	}
	if (nji == 0 || nki == 0) { // no interior: the shell is the whole class
		launch_rows<BS>(efirst, A, qf, q, II, JJ, KK, jb, kb, st);
		return;
	}
	// shell = planes below klo / from khi (all rows), and in the planes between: rows below jlo / from jhi
	ShellRects rc;
	rc.n = 0;
	rc.start[0] = 0;
	auto add = [&](int j0, int nj, int kr0, int nk) {
		if (nj <= 0 || nk <= 0) return;
		rc.j0[rc.n] = j0; rc.nrj[rc.n] = nj; rc.kr0[rc.n] = kr0;
		rc.start[rc.n + 1] = rc.start[rc.n] + nj * nk;
		rc.n++;
	};
	add(1 + jb, nrj, 0, klo);
	add(1 + jb, nrj, khi, nrk - khi);
	add(1 + jb, jlo, klo, nki);
	add(1 + jb + 2 * jhi, nrj - jhi, klo, nki);
	if (rc.n == 0) return;
	for (int t = rc.n; t < 4; t++) { rc.j0[t] = 0; rc.nrj[t] = 1; rc.kr0[t] = 0; rc.start[t + 1] = rc.start[rc.n]; }
	static const bool nt = getenv("CEDAR_AMD_NT") ? atoi(getenv("CEDAR_AMD_NT")) != 0 : true;
	const dim3 grid((unsigned)rc.start[rc.n]);
	if (efirst) {
		if (nt) hipLaunchKernelGGL((relax27_rows_shell<BS, true, true>), grid, dim3(BS), 0, st, A, qf, q, II, JJ, KK, kb, rc);
		else hipLaunchKernelGGL((relax27_rows_shell<BS, true, false>), grid, dim3(BS), 0, st, A, qf, q, II, JJ, KK, kb, rc);
	} else {
		if (nt) hipLaunchKernelGGL((relax27_rows_shell<BS, false, true>), grid, dim3(BS), 0, st, A, qf, q, II, JJ, KK, kb, rc);
		else hipLaunchKernelGGL((relax27_rows_shell<BS, false, false>), grid, dim3(BS), 0, st, A, qf, q, II, JJ, KK, kb, rc);
	}
}

void relax3_pass27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int KK, int jb, int kb, int efirst, hipStream_t st, int part_sides)
{
	const Op3 A = op3_lookup(so, sor, II, JJ, KK);
	// part_sides = part | sides << 4 (include/cedar_amd.h); no side bit set = every face has a neighbour
	const int part = part_sides & 3, sides = ((part_sides >> 4) & 15) ? (part_sides >> 4) & 15 : 15;
	const int npairs = (II - 2 + 1) / 2;
	if (npairs <= 64) launch_part<64>(efirst, A, qf, q, II, JJ, KK, jb, kb, part, sides, st);
	else if (npairs <= 128) launch_part<128>(efirst, A, qf, q, II, JJ, KK, jb, kb, part, sides, st);
	else if (npairs <= 256) launch_part<256>(efirst, A, qf, q, II, JJ, KK, jb, kb, part, sides, st);
	else if (npairs <= 512) launch_part<512>(efirst, A, qf, q, II, JJ, KK, jb, kb, part, sides, st);
	else {
		if (part == 1) return; // rows too long for the row kernel: everything goes with the shell
		for (int c = 0; c < 2; c++) {
			int ib = efirst ? c : 1 - c;
			int ni = (II - 2 - ib + 1) / 2, nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
			if (ni <= 0 || nj <= 0 || nk <= 0) continue;
			hipLaunchKernelGGL(relax27_colour, dim3(cap_grid((size_t)ni * nj * nk, 256)), dim3(256), 0, st,
			                   A, qf, q, II, JJ, KK, ib, jb, kb);
		}
	}
}

// recompute the points of column `icol` (0-based incl. ghost) in the rows of class (jb,kb):
// used after a halo update of the neighbouring first-colour column (distributed runs)
__global__ void relax27_column(const Op3 A, const real_t *__restrict__ qf, real_t *__restrict__ q,
                               int II, int JJ, int KK, int icol, int jb, int kb)
{
	int nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nj * nk) return;
	const size_t i = (size_t)icol, j = (size_t)(1 + jb + 2 * (t % nj)), k = (size_t)(1 + kb + 2 * (t / nj));
	size_t x = i + (size_t)II * (j + (size_t)JJ * k);
	q[x] = offdiag27_mem(A, qf, q, II, JJ, x, i + j * A.SJ + k * A.SK) * A.sor[i + j * A.rSJ + k * A.rSK];
}

void relax3_fixup27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                    int II, int JJ, int KK, int icol, int jb, int kb, hipStream_t st)
{
	const Op3 A = op3_lookup(so, sor, II, JJ, KK);
	int nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
	if (nj <= 0 || nk <= 0) return;
	hipLaunchKernelGGL(relax27_column, dim3((nj * nk + 127) / 128), dim3(128), 0, st, A, qf, q, II, JJ, KK, icol, jb, kb);
}

// ---- boundary-first chain of a rank grid with an x / y split (dist3.cpp smooth): the few columns and rows next to a
// neighbouring rank are relaxed stage by stage ahead of the big launch of a k-parity, which then leaves them as they are
struct ChainCols {
	int n, col[8], xrow[2];
};

// one thread per (row of class jb, plane of parity kb): the row's points in the listed columns, in order, through memory
// (a later point of the list reads the earlier ones' fresh values like any other neighbour)
__global__ void relax27_cols(const Op3 A, const real_t *__restrict__ qf, real_t *q, int II, int JJ, int KK, int jb, int kb,
                             ChainCols cc)
{
	const int nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nj * nk) return;
	const int jr = 1 + jb + 2 * (t % nj);
	if (jr == cc.xrow[0] || jr == cc.xrow[1]) return;
	const size_t j = (size_t)jr, k = (size_t)(1 + kb + 2 * (t / nj));
	for (int c = 0; c < cc.n; c++) {
		const size_t i = (size_t)cc.col[c];
		const size_t x = i + (size_t)II * (j + (size_t)JJ * k);
		q[x] = offdiag27_mem(A, qf, q, II, JJ, x, i + j * A.SJ + k * A.SK) * A.sor[i + j * A.rSJ + k * A.rSK];
	}
}

// The same from a dense copy of the operator's six columns next to an x face (relax3_strip_build): the column kernel
// above fetches a 128-byte line for every 8-byte entry it uses (55 lines per point; 0.1 ms per stage at 512^3, 2.5 ms per
// V-cycle); here the 25 coefficient segments of a point and 1/diag are 48-byte pieces of an array laid out
// [slot][k][j parity][j / 2][6 columns], so the rows of a stage share lines.  q and the right-hand side stay where they are.
// Window: offsets 0..5 (low side) or II-6..II-1 (high side); slot 14 = 1/diag.
struct StripRef {
	const real_t *lo, *hi;
};
constexpr int STRIP_W = 6, STRIP_SLOTS = 15;

// rows of one j-parity are contiguous: the lanes of a stage (rows of one class) read 48-byte pieces back to back, from the
// own class for entries stored at row j and from the other class for those stored at row j+1
__host__ __device__ __forceinline__ size_t strip_index(int JJ, int KK, size_t slot, size_t j, size_t k, int w)
{
	const size_t JH = ((size_t)JJ + 1) / 2;
	return ((((slot * KK + k) * 2 + (j & 1)) * JH + (j >> 1)) * STRIP_W) + w;
}
__device__ __forceinline__ real_t strip_at(const real_t *__restrict__ S, int JJ, int KK, int slot, size_t j, size_t k, int w)
{
	return S[strip_index(JJ, KK, (size_t)slot, j, k, w)];
}

__global__ void relax27_cols_strip(StripRef sr, const real_t *__restrict__ qf, real_t *q, int II, int JJ, int KK, int jb, int kb,
                                   ChainCols cc)
{
	const int nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nj * nk) return;
	const int jr = 1 + jb + 2 * (t % nj);
	if (jr == cc.xrow[0] || jr == cc.xrow[1]) return;
	const size_t j = (size_t)jr, k = (size_t)(1 + kb + 2 * (t / nj));
	const size_t sj = II, sk = (size_t)II * JJ;
	for (int cidx = 0; cidx < cc.n; cidx++) {
		const int i = cc.col[cidx];
		const bool low = i < II / 2;
		const real_t *__restrict__ S = low ? sr.lo : sr.hi;
		const int w = low ? i : i - (II - STRIP_W);
		C27 c;
#define SA(slot, dj, dk, dw) strip_at(S, JJ, KK, slot, j + dj, k + dk, w + dw)
		c.pw = SA(KPW, 0, 0, 0); c.ps = SA(KPS, 0, 0, 0); c.psw = SA(KPSW, 0, 0, 0);
		c.b = SA(KB, 0, 0, 0); c.bw = SA(KBW, 0, 0, 0); c.bs = SA(KBS, 0, 0, 0); c.bsw = SA(KBSW, 0, 0, 0);
		c.pnw_n = SA(KPNW, 1, 0, 0); c.ps_n = SA(KPS, 1, 0, 0); c.bnw_n = SA(KBNW, 1, 0, 0); c.bn_n = SA(KBN, 1, 0, 0);
		c.b_t = SA(KB, 0, 1, 0); c.be_t = SA(KBE, 0, 1, 0); c.bn_t = SA(KBN, 0, 1, 0); c.bne_t = SA(KBNE, 0, 1, 0);
		c.bse_nt = SA(KBSE, 1, 1, 0); c.bs_nt = SA(KBS, 1, 1, 0);
		c.psw_ne = SA(KPSW, 1, 0, 1); c.bne_ne = SA(KBNE, 1, 0, 1);
		c.pw_e = SA(KPW, 0, 0, 1); c.pnw_e = SA(KPNW, 0, 0, 1); c.be_e = SA(KBE, 0, 0, 1); c.bse_e = SA(KBSE, 0, 0, 1);
		c.bsw_net = SA(KBSW, 1, 1, 1);
		c.bw_et = SA(KBW, 0, 1, 1); c.bnw_et = SA(KBNW, 0, 1, 1);
		const real_t rd = SA(14, 0, 0, 0);
#undef SA
		const size_t x = (size_t)i + sj * j + sk * k;
		real_t qq[3][3][3];
#pragma unroll
		for (int dk = 0; dk < 3; dk++)
#pragma unroll
			for (int dj = 0; dj < 3; dj++)
#pragma unroll
				for (int di = 0; di < 3; di++)
					qq[dk][dj][di] = q[x + (di - 1) + (ptrdiff_t)(dj - 1) * (ptrdiff_t)sj + (ptrdiff_t)(dk - 1) * (ptrdiff_t)sk];
		q[x] = offdiag27(qf[x], c, qq) * rd;
	}
}

// dense copy of the six columns next to the low (side 0) / high (side 1) x face: out[15][KK][JJ][6] (slot 14 = 1/diag, sor's second plane)
__global__ void strip_build_kernel(const real_t *__restrict__ so, const real_t *__restrict__ sor, real_t *__restrict__ out,
                                   int II, int JJ, int KK, int w0)
{
	const size_t n = (size_t)STRIP_SLOTS * KK * JJ * STRIP_W, PS = (size_t)II * JJ * KK;
	for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
		const int w = (int)(t % STRIP_W);
		const size_t r = t / STRIP_W, j = r % JJ, k = (r / JJ) % KK, slot = r / ((size_t)JJ * KK);
		const size_t x = (size_t)(w0 + w) + (size_t)II * (j + (size_t)JJ * k);
		out[strip_index(JJ, KK, slot, j, k, w)] = slot < 14 ? so[slot * PS + x] : sor[PS + x]; // SOR plane msor = 1/diag (op3_cedar)
	}
}

size_t relax3_strip_doubles(int JJ, int KK) { return (size_t)STRIP_SLOTS * KK * 2 * (((size_t)JJ + 1) / 2) * STRIP_W; }

void relax3_strip_build(const real_t *so, const real_t *sor, int II, int JJ, int KK, int side, real_t *out, hipStream_t st)
{
	if (II < 2 * STRIP_W) return;
	hipLaunchKernelGGL(strip_build_kernel, dim3(cap_grid(relax3_strip_doubles(JJ, KK), 256)), dim3(256), 0, st, so, sor, out, II, JJ,
	                   KK, side ? II - STRIP_W : 0);
}

void relax3_cols27_strip(const real_t *strip_lo, const real_t *strip_hi, const real_t *qf, real_t *q, int II, int JJ, int KK, int jb,
                         int kb, int ncol, const int *cols, int xrow0, int xrow1, hipStream_t st)
{
	const int nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
	if (nj <= 0 || nk <= 0 || ncol <= 0) return;
	ChainCols cc;
	cc.n = ncol < 8 ? ncol : 8;
	for (int c = 0; c < 8; c++) cc.col[c] = c < cc.n ? cols[c] : 1;
	cc.xrow[0] = xrow0; cc.xrow[1] = xrow1;
	StripRef sr;
	sr.lo = strip_lo; sr.hi = strip_hi;
	hipLaunchKernelGGL(relax27_cols_strip, dim3((nj * nk + 63) / 64), dim3(64), 0, st, sr, qf, q, II, JJ, KK, jb, kb, cc);
}

void relax3_cols27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int KK, int jb, int kb,
                   int ncol, const int *cols, int xrow0, int xrow1, hipStream_t st)
{
	const Op3 A = op3_lookup(so, sor, II, JJ, KK);
	const int nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
	if (nj <= 0 || nk <= 0 || ncol <= 0) return;
	ChainCols cc;
	cc.n = ncol < 8 ? ncol : 8;
	for (int c = 0; c < 8; c++) cc.col[c] = c < cc.n ? cols[c] : 1;
	cc.xrow[0] = xrow0; cc.xrow[1] = xrow1;
	hipLaunchKernelGGL(relax27_cols, dim3((nj * nk + 63) / 64), dim3(64), 0, st, A, qf, q, II, JJ, KK, jb, kb, cc);
}

void relax3_rows27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int KK, int j0, int jstep,
                   int nrj, int kb, int efirst, hipStream_t st)
{
	const Op3 A = op3_lookup(so, sor, II, JJ, KK);
	const int nrk = (KK - 2 - kb + 1) / 2, npairs = (II - 2 + 1) / 2;
	if (nrj <= 0 || nrk <= 0) return;
	if (npairs <= 64) launch_rows_at<64>(efirst, A, qf, q, II, JJ, KK, j0, jstep, nrj, kb, nrk, st, 0);
	else if (npairs <= 128) launch_rows_at<128>(efirst, A, qf, q, II, JJ, KK, j0, jstep, nrj, kb, nrk, st, 0);
	else if (npairs <= 256) launch_rows_at<256>(efirst, A, qf, q, II, JJ, KK, j0, jstep, nrj, kb, nrk, st, 0);
	else launch_rows_at<512>(efirst, A, qf, q, II, JJ, KK, j0, jstep, nrj, kb, nrk, st, 0);
}

void relax3_colour7(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                    int II, int JJ, int KK, int pts, hipStream_t st)
{
	size_t n = (size_t)((II - 2 + 1) / 2) * (JJ - 2) * (KK - 2);
	hipLaunchKernelGGL(relax7_colour, dim3(cap_grid(n, 256)), dim3(256), 0, st, so, qf, q, sor, II, JJ, KK, pts);
}

// Both row classes of the planes of parity kb, in sweep order (UP: j-parity 0 rows first, even i first;
// DOWN the reverse): what a slab-decomposed run (rank grid 1 x 1 x pz) does between two halo exchanges --
// the second row class needs nothing from another rank.  part: 0 all planes of the parity, 1 = those
// with both k-neighbours owned (they read no ghost plane), 2 = the first / last owned plane.
template <int BS>
static void planes_bs(bool up, const Op3 &A, const real_t *qf, real_t *q,
                      int II, int JJ, int KK, int kb, int kr0, int nrk, hipStream_t st)
{
	if (nrk <= 0) return;
	const int jbF = up ? 0 : 1;
	const int frun = relax3_plane_frun(JJ);
	// the fused kernel gives one workgroup a run of frun rows: a piece of one or two planes (the shell of a slab)
	// would occupy a fraction of the CUs for eight row tasks in a row -- such pieces take the row kernels
	const int nF = (JJ - 2 - jbF + 1) / 2;
	if (frun > 0 && (size_t)nrk * (size_t)((nF + frun - 1) / frun) >= 256) {
		launch_plane<BS>(up, A, qf, q, II, JJ, KK, jbF, kb, frun, st, kr0, nrk);
		return;
	}
	for (int c = 0; c < 2; c++) {
		const int jb = c == 0 ? jbF : 1 - jbF;
		launch_rows_at<BS>(up, A, qf, q, II, JJ, KK, 1 + jb, 2, (JJ - 2 - jb + 1) / 2, kb, nrk, st, kr0);
	}
}

void relax3_planes27(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                     int II, int JJ, int KK, int kb, int up, int part_sides, hipStream_t st)
{
	const Op3 A = op3_lookup(so, sor, II, JJ, KK);
	const int part = part_sides & 3, sides = ((part_sides >> 4) & 15) ? (part_sides >> 4) & 15 : 15;
	const int nrk = (KK - 2 - kb + 1) / 2;
	if (II < 3 || JJ < 3 || nrk <= 0) return;
	const int npairs = (II - 2 + 1) / 2;
	const int klo = (kb == 0 && (sides & 4)) ? 1 : 0;
	const int khi = (1 + kb + 2 * (nrk - 1) == KK - 2 && (sides & 8)) ? nrk - 1 : nrk;
	const int nki = khi - klo > 0 ? khi - klo : 0;
	// (kr0, count) pieces of the requested part
	int pieces[2][2] = { { 0, nrk }, { 0, 0 } };
	if (part == 1) { pieces[0][0] = klo; pieces[0][1] = nki; }
	else if (part == 2 && nki > 0) { pieces[0][0] = 0; pieces[0][1] = klo; pieces[1][0] = khi; pieces[1][1] = nrk - khi; }
	// partial sums (relax3d_psum.hip) where the operator was registered with a scratch vector: a k-parity of planes is the
	// A or the B phase of the sweep; planes next to a neighbouring rank's ghost plane leave / take none (they wait for the halo
	// and run as small pieces in the reference order)
	const IlvReg *reg = reg_lookup(so, II, JJ, KK);
	const int pfrun = relax3_psum_frun(JJ);
	const bool psum = reg && reg->T && npairs <= 256 && relax3_psum_wanted(II, JJ, KK);
	const int nbr = ((sides & 4) ? 1 : 0) | ((sides & 8) ? 2 : 0);
	const bool first_parity = up ? kb == 0 : kb == 1;
	for (auto &pc : pieces) {
		if (pc[1] <= 0) continue;
		if (psum) {
			const int nrun = ((JJ - 2 - (up ? 0 : 1) + 1) / 2 + pfrun - 1) / pfrun;
			bool gives = false; // does any plane of an A piece leave partial sums?  (a piece of shell planes does not)
			for (int kr = pc[0]; kr < pc[0] + pc[1] && !gives; kr++) {
				const int k = 1 + kb + 2 * kr;
				gives = !(k == 1 && (nbr & 1)) && !(k == KK - 2 && (nbr & 2));
			}
			if (!first_parity || gives || (size_t)pc[1] * (size_t)nrun >= 256) {
				relax3_planes27_psum(A, qf, q, reg->T, II, JJ, KK, kb, up, pc[0], pc[1], nbr, pfrun, st);
				continue;
			}
		}
		if (npairs <= 64) planes_bs<64>(up, A, qf, q, II, JJ, KK, kb, pc[0], pc[1], st);
		else if (npairs <= 128) planes_bs<128>(up, A, qf, q, II, JJ, KK, kb, pc[0], pc[1], st);
		else if (npairs <= 256) planes_bs<256>(up, A, qf, q, II, JJ, KK, kb, pc[0], pc[1], st);
		else if (npairs <= 512) planes_bs<512>(up, A, qf, q, II, JJ, KK, kb, pc[0], pc[1], st);
		else if (part != 1) { // rows too long for the row kernels: whole colours, everything with the shell
			for (int c = 0; c < 4; c++) {
				const int jb = (c >> 1) == 0 ? (up ? 0 : 1) : (up ? 1 : 0), ib = (c & 1) == 0 ? (up ? 0 : 1) : (up ? 1 : 0);
				int ni = (II - 2 - ib + 1) / 2, nj = (JJ - 2 - jb + 1) / 2;
				if (ni <= 0 || nj <= 0) continue;
				hipLaunchKernelGGL(relax27_colour, dim3(cap_grid((size_t)ni * nj * nrk, 256)), dim3(256), 0, st,
				                   A, qf, q, II, JJ, KK, ib, jb, kb);
			}
			break;
		}
	}
}

bool relax3_planes27_masked(const real_t *so, const real_t *qf, real_t *q, const real_t *sor, int II, int JJ, int KK, int kb,
                            int up, const PsumSkip &skip, hipStream_t st)
{
	const Op3 A = op3_lookup(so, sor, II, JJ, KK);
	const IlvReg *reg = reg_lookup(so, II, JJ, KK);
	const int npairs = (II - 2 + 1) / 2, nrk = (KK - 2 - kb + 1) / 2;
	if (!(reg && reg->T && reg->frun > 0 && npairs >= 4 && npairs <= 256)) return false;
	if (nrk > 0) relax3_planes27_psum(A, qf, q, reg->T, II, JJ, KK, kb, up, 0, nrk, 0, reg->frun, st, &skip);
	return true;
}

// Run length of the partial-sum sweep (relax3d_psum.hip).  A longer run leaves fewer rows in the reference order (3 in
// 2*frun) but fewer workgroups per launch.  Measured, ms per sweep for runs of 8 / 16 / 32 rows against the reference-order
// sweep of that level (profiles/r03_psum_run_length.log; single runs scatter by +-5 % with the placement of the level):
//   512^3 4.70 / 4.23 / 4.47 (and 4.75 / 4.53 / 4.40 / 4.37 for 8 / 16 / 32 / 64 on another box) against 6.17;
//   448^3 3.19 / 3.21 / 3.45 against 4.20;  384^3 2.04 / 1.96 / 2.08;  320^3 1.27 / 1.28 / 1.32 against 1.58;
//   256^3 0.71 / 0.65 / 0.66 against 0.79 (0.69 for the four row-class launches);  224^3 0.47 / 0.47 / 0.52 against 0.53;
//   192^3 0.30 / 0.31 / 0.45 against 0.34;  128^3 slower than the row-class launches for every length.
// CEDAR_AMD_FRUN overrides as for the plane-fused walk (n = runs of n rows wherever a plane has at least 4 of them).
int relax3_psum_frun(int JJ)
{
	const char *e = getenv("CEDAR_AMD_FRUN"); // read per call: the tests switch it between cases
	const int ny = JJ - 2;
	if (e) {
		const int frun = atoi(e);
		return (frun <= 0 || ny < 4 * frun) ? 0 : frun;
	}
	return ny >= 480 ? 32 : ny >= 224 ? 16 : ny >= 160 ? 8 : 0;
}

// CEDAR_AMD_PSUM: 0 = reference order everywhere; unset / 1 = partial sums on the levels relax3_psum_frun names
bool relax3_psum_wanted(int II, int JJ, int KK)
{
	const char *e = getenv("CEDAR_AMD_PSUM"); // read per call: the tests switch it between cases
	if (e && atoi(e) == 0) return false;
	return relax3_psum_ok(II, JJ, KK, relax3_psum_frun(JJ));
}

void relax3_gs27_op(const Op3 &A, const real_t *qf, real_t *q, int II, int JJ, int KK, int updown, hipStream_t st, real_t *T)
{
	if (II < 3 || JJ < 3 || KK < 3) return;
	const char *ew0 = getenv("CEDAR_AMD_WHATIF"), *ec0 = getenv("CEDAR_AMD_KCHUNK");
	if (T && relax3_psum_wanted(II, JJ, KK) && !(ew0 && atoi(ew0)) && !(ec0 && atoi(ec0))) {
		relax3_gs27_psum(A, qf, q, T, II, JJ, KK, updown, relax3_psum_frun(JJ), st);
		return;
	}
	{
		const bool up = (updown == BMG_UP);
		const int npairs = (II - 2 + 1) / 2;
		const int frun = relax3_plane_frun(JJ);
		if (npairs <= 512 && frun > 0) {
			// plane-fused: UP planes of parity 0 then 1, in a plane j-parity 0 rows first; DOWN the reverse
			const char *ew = getenv("CEDAR_AMD_WHATIF");
			const bool kpair = ew && atoi(ew) == 8 && npairs > 128 && npairs <= 256; // what-if: ONE launch over plane pairs
			// experiment (bit-identical): planes in chunks -- first-parity planes [.., m], then the second-parity planes
			// between them, and so on (a second-parity plane only needs its two first-parity neighbours done, a first-parity
			// plane must run before its second-parity neighbours): the nine inter-plane slot planes both parities read are
			// then a few planes apart instead of a whole sweep half, i.e. within reach of the Infinity Cache
			const char *ec = getenv("CEDAR_AMD_KCHUNK");
			const int kchunk = ec ? atoi(ec) : 0;
			if (kchunk > 0 && !kpair && npairs > 128 && npairs <= 256) {
				const int kbF = up ? 0 : 1, kbS = 1 - kbF, jbF = up ? 0 : 1;
				const int nF = (KK - 2 - kbF + 1) / 2, nS = (KK - 2 - kbS + 1) / 2;
				int fdone = 0;
				for (int c0 = 0; c0 < nS; c0 += kchunk) {
					const int hi = c0 + kchunk < nS ? c0 + kchunk : nS;
					int needF = up ? hi + 1 : hi; // UP: S(r) lies between F(r), F(r+1); DOWN: between F(r-1), F(r)
					if (needF > nF) needF = nF;
					if (needF > fdone) { launch_plane<256>(up, A, qf, q, II, JJ, KK, jbF, kbF, frun, st, fdone, needF - fdone); fdone = needF; }
					launch_plane<256>(up, A, qf, q, II, JJ, KK, jbF, kbS, frun, st, c0, hi - c0);
				}
				if (fdone < nF) launch_plane<256>(up, A, qf, q, II, JJ, KK, jbF, kbF, frun, st, fdone, nF - fdone);
				return;
			}
			for (int c = 0; c < 2; c++) {
				const int kb = kpair ? 1 : (up ? c : 1 - c), jbF = up ? 0 : 1;
				if (kpair && c == 1) break;
				if (npairs <= 64) launch_plane<64>(up, A, qf, q, II, JJ, KK, jbF, kb, frun, st);
				else if (npairs <= 128) launch_plane<128>(up, A, qf, q, II, JJ, KK, jbF, kb, frun, st);
				else if (npairs <= 256) launch_plane<256>(up, A, qf, q, II, JJ, KK, jbF, kb, frun, st);
				else launch_plane<512>(up, A, qf, q, II, JJ, KK, jbF, kb, frun, st);
			}
		} else if (npairs <= 512) {
			// colour pairs in sweep order: UP (j,k) parities 00,10,01,11 with even-i first
			for (int c = 0; c < 4; c++) {
				int cc = up ? c : 3 - c;
				int jb = cc & 1, kb = cc >> 1;
				if (npairs <= 64) launch_rows<64>(up, A, qf, q, II, JJ, KK, jb, kb, st);
				else if (npairs <= 128) launch_rows<128>(up, A, qf, q, II, JJ, KK, jb, kb, st);
				else if (npairs <= 256) launch_rows<256>(up, A, qf, q, II, JJ, KK, jb, kb, st);
				else launch_rows<512>(up, A, qf, q, II, JJ, KK, jb, kb, st);
			}
		} else {
			for (int c = 0; c < 8; c++) {
				int pts = up ? c : 7 - c;
				int ib = pts & 1, jb = (pts >> 1) & 1, kb = (pts >> 2) & 1;
				int ni = (II - 2 - ib + 1) / 2, nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
				if (ni <= 0 || nj <= 0 || nk <= 0) continue;
				hipLaunchKernelGGL(relax27_colour, dim3(cap_grid((size_t)ni * nj * nk, 256)), dim3(256), 0, st,
				                   A, qf, q, II, JJ, KK, ib, jb, kb);
			}
		}
	}
}

void relax3_gs(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
               int II, int JJ, int KK, int nstncl, int updown, hipStream_t st)
{
	if (II < 3 || JJ < 3 || KK < 3) return;
	if (nstncl == 14) {
		relax3_gs27_op(op3_lookup(so, sor, II, JJ, KK), qf, q, II, JJ, KK, updown, st);
	} else {
		// 7-point: UP = colours 0,1; DOWN = 1,0 (:144-153)
		for (int c = 0; c < 2; c++) {
			int pts = (updown == BMG_UP) ? c : 1 - c;
			size_t n = (size_t)((II - 2 + 1) / 2) * (JJ - 2) * (KK - 2);
			hipLaunchKernelGGL(relax7_colour, dim3(cap_grid(n, 256)), dim3(256), 0, st, so, qf, q, sor, II, JJ, KK, pts);
		}
	}
}

// Periodic sweep (BMG3_SymStd_relax_GS.f90:188-357): the colours of the Dirichlet sweep, one launch each, every
// colour followed by the ghost refreshes the reference performs while it walks that colour (x ghosts of a row when
// the row is done, y ghosts of a plane when the plane is done; points of one colour never read each other nor a
// ghost refreshed under the same colour, so refreshing after the launch gives the same values), the z ghost planes
// only once the sweep is over (:279-286) -- until then they hold the previous sweep's values, as in the reference.
// ghosts_consistent: the caller knows that the y ghost rows of q hold the periodic image on entry (true inside the
// cycle after the first sweep on a level; not for arrays handed in from outside) -- see the fast path below.
void relax3_gs_per(const real_t *so, const real_t *qf, real_t *q, const real_t *sor,
                   int II, int JJ, int KK, int nstncl, int updown, int ipn, hipStream_t st, int ghosts_consistent)
{
	if (II < 3 || JJ < 3 || KK < 3) return;
	const bool up = (updown == BMG_UP);
	const bool px = ipn == 2 || ipn == 3 || ipn == 6 || ipn == 8, py = ipn == 1 || ipn == 3 || ipn == 7 || ipn == 8, pz = ipn >= 5;
	if (nstncl == 14 && !(px && (II & 1)) && (!py || ghosts_consistent) && !(py && (JJ & 1)) && !(pz && (KK & 1))
	    && (II - 2 + 1) / 2 <= 512) {
		// x not periodic, even extents in the periodic directions (every level the solver relaxes on): the two i-colours of
		// a row class run back to back in the row kernel.  The y refresh the reference performs between them copies rows
		// 2 and ny+1, which belong to different classes when ny is even: it rewrites ghost rows with the values they
		// already hold -- provided they held the periodic image on entry (ghosts_consistent) -- so refreshing once per
		// class gives the same values.  The z ghosts are refreshed at the end of the sweep in either path.
		for (int c = 0; c < 4; c++) {
			const int cc = up ? c : 3 - c, jb = cc & 1, kb = cc >> 1;
			if (px) { // the row kernel refreshes the x ghosts of its row itself (relax27_row_task PERX)
				const Op3 A = op3_cedar(so, sor, II, JJ, KK);
				const int npairs = (II - 2 + 1) / 2;
				if (npairs <= 64) launch_rows_perx<64>(up, A, qf, q, II, JJ, KK, jb, kb, st);
				else if (npairs <= 128) launch_rows_perx<128>(up, A, qf, q, II, JJ, KK, jb, kb, st);
				else if (npairs <= 256) launch_rows_perx<256>(up, A, qf, q, II, JJ, KK, jb, kb, st);
				else launch_rows_perx<512>(up, A, qf, q, II, JJ, KK, jb, kb, st);
			} else
				relax3_pass27(so, qf, q, sor, II, JJ, KK, jb, kb, up, st);
			wrap3_colour(q, II, JJ, KK, jb, kb, px ? (ipn == 2 ? 0 : ipn == 3 ? 1 : ipn == 6 ? 5 : 7) : ipn, st); // y only: x is done
		}
		wrap3_sweep_end(q, II, JJ, KK, ipn, st);
		return;
	}
	if (nstncl == 14) {
		const Op3 A = op3_cedar(so, sor, II, JJ, KK);
		for (int c = 0; c < 8; c++) {
			const int pts = up ? c : 7 - c;
			const int ib = pts & 1, jb = (pts >> 1) & 1, kb = (pts >> 2) & 1;
			const int ni = (II - 2 - ib + 1) / 2, nj = (JJ - 2 - jb + 1) / 2, nk = (KK - 2 - kb + 1) / 2;
			if (nj <= 0 || nk <= 0) continue;
			if (ni > 0)
				hipLaunchKernelGGL(relax27_colour, dim3(cap_grid((size_t)ni * nj * nk, 256)), dim3(256), 0, st,
				                   A, qf, q, II, JJ, KK, ib, jb, kb);
			wrap3_colour(q, II, JJ, KK, jb, kb, ipn, st);
		}
	} else {
		for (int c = 0; c < 2; c++) {
			const int pts = up ? c : 1 - c;
			const size_t n = (size_t)((II - 2 + 1) / 2) * (JJ - 2) * (KK - 2);
			hipLaunchKernelGGL(relax7_colour, dim3(cap_grid(n, 256)), dim3(256), 0, st, so, qf, q, sor, II, JJ, KK, pts);
			wrap3_colour(q, II, JJ, KK, -1, -1, ipn, st);
		}
	}
	wrap3_sweep_end(q, II, JJ, KK, ipn, st);
}

} // namespace cedar_amd
