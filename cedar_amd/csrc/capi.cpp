// C-ABI layer 1: the BMG2_/BMG3_SymStd_* drop-in entry points, the device
// memory helpers and the library-wide runtime state (stream, staging pool,
// error callback).  Declarations + reference citations: include/cedar_amd.h.
#include "../../include/cedar_amd.h"
#include "common.h"
#include "stage.h"
#include "relax3_psum.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>

using namespace cedar_amd;

// ------------------------------------------------------------------ error callback
// The reference's kernels import print_error from the host program
// (src/2d/ftn/ModInterface.f90:4-9).  Same contract here: a host definition wins
// over this weak default.
extern "C" __attribute__((weak)) void print_error(char *msg)
{
	fprintf(stderr, "[cedar_amd] %s\n", msg);
}

namespace cedar_amd {

static hipStream_t g_stream = nullptr;
hipStream_t current_stream() { return g_stream; }

bool is_device_ptr(const void *p)
{
	hipPointerAttribute_t attr;
	hipError_t e = hipPointerGetAttributes(&attr, p);
	if (e != hipSuccess) {
		(void)hipGetLastError(); // plain host memory: clear the sticky error
		return false;
	}
	return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// tiny size-bucketed pool so that per-call staging does not hipMalloc every time
static std::multimap<size_t, void *> g_pool;
static size_t g_pool_bytes = 0;
static const size_t POOL_CAP = (size_t)8 << 30;

void *pool_get(size_t bytes)
{
	auto it = g_pool.lower_bound(bytes);
	if (it != g_pool.end() && it->first <= bytes + bytes / 4 + 4096) {
		void *p = it->second;
		g_pool_bytes -= it->first;
		g_pool.erase(it);
		return p;
	}
	void *p = nullptr;
	CEDAR_HIP_CHECK(hipMalloc(&p, bytes));
	return p;
}

void pool_put(void *p, size_t bytes)
{
	// the true capacity of a recycled block is unknown; blocks are re-keyed by
	// the size they were last requested with (always <= capacity)
	if (g_pool_bytes + bytes > POOL_CAP) {
		CEDAR_HIP_CHECK(hipFree(p));
		return;
	}
	g_pool.emplace(bytes, p);
	g_pool_bytes += bytes;
}

void launch_check(const char *who)
{
	const hipError_t e = hipGetLastError(); // clears it
	if (e == hipSuccess) return;
	char buf[256];
	snprintf(buf, sizeof(buf), "%s: a kernel launch was rejected by the HIP runtime (%s: %s); results are incomplete", who,
	         hipGetErrorName(e), hipGetErrorString(e));
	print_error(buf);
}

static void report(const char *msg)
{
	char buf[256];
	strncpy(buf, msg, sizeof(buf) - 1);
	buf[sizeof(buf) - 1] = 0;
	print_error(buf);
}

// 2D kernels with a periodic branch on the GPU path: |jpn| in {1 per_y, 2 per_x, 3 per_xy}.  Returns
// -1 after reporting when the code is not served (indefinite variants, 3D codes), else |jpn|.
static int bc2(int jpn, const char *who)
{
	if (jpn >= 0 && jpn <= 3) return jpn;
	char buf[200];
	snprintf(buf, sizeof(buf), "%s: boundary code %d is not implemented on the GPU path (0 definite, 1..3 periodic y/x/xy)", who, jpn);
	report(buf);
	return -1;
}

// 3D boundary code of BMG_get_bc (0, 1 y, 2 x, 3 xy, 5 z, 6 xz, 7 yz, 8 xyz); the indefinite (negative) codes
// and -4 are not implemented
static bool bc3(int jpn, const char *who)
{
	if (jpn >= 0 && periodic3_code_ok(jpn)) return true;
	char buf[200];
	snprintf(buf, sizeof(buf), "%s: boundary code %d is not implemented on the GPU path (0 and the definite periodic codes are)",
	         who, jpn);
	report(buf);
	return false;
}

// the periodic interpolation set-up and Galerkin product coarsen a periodic direction by pairs: its extent must
// be even (the reference's example uses even extents; odd ones are refused rather than guessed)
static bool even_periodic(int ipn, len_t iif, len_t jjf, len_t kkf, const char *who)
{
	const bool px = ipn == 2 || ipn == 3 || ipn == 6 || ipn == 8, py = ipn == 1 || ipn == 3 || ipn == 7 || ipn == 8,
	           pz = ipn >= 5 && ipn <= 8;
	if ((px && (iif & 1)) || (py && (jjf & 1)) || (pz && (kkf & 1))) {
		char buf[200];
		snprintf(buf, sizeof(buf), "%s: a periodic direction needs an even extent (got %u x %u x %u, code %d)", who,
		         (unsigned)iif - 2, (unsigned)jjf - 2, (unsigned)kkf - 2, ipn);
		report(buf);
		return false;
	}
	return true;
}

} // namespace cedar_amd

extern "C" {

const char *cedar_amd_version(void) { return "cedar_amd 0.1 (gfx950)"; }

// src/2d/ftn/BMG_get_bc.f90:11-22 with include/cedar/2d/ftn/BMG_parameters_c.h values
void BMG_get_bc(int per_mask, int *ibc)
{
	// values: src/3d/ftn/BMG_parameters_f90.h:345-360
	static const int bcmap[8] = { 0 /*definite*/, 2 /*per_x*/, 1 /*per_y*/, 3 /*per_xy*/,
		                          5 /*per_z*/, 6 /*per_xz*/, 7 /*per_yz*/, 8 /*per_xyz*/ };
	*ibc = bcmap[per_mask & 7];
}

// ------------------------------------------------------------------ memory helpers
int cedar_amd_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	return n;
}
int cedar_amd_set_device(int dev) { return hipSetDevice(dev) == hipSuccess ? 0 : 1; }
void cedar_amd_memset(void *dst, int value, size_t bytes);
void *cedar_amd_malloc(size_t bytes)
{
	void *p = nullptr;
	CEDAR_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
	cedar_amd_memset(p, 0, bytes);
	return p;
}
void cedar_amd_free(void *p)
{
	if (!p) return;
	relax3_release(static_cast<const real_t *>(p)); // an operator registered with cedar_amd_relax3_prepare goes with its copy
	CEDAR_HIP_CHECK(hipFree(p));
}
void cedar_amd_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
	CEDAR_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, current_stream()));
	CEDAR_HIP_CHECK(hipStreamSynchronize(current_stream()));
}
void cedar_amd_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
	CEDAR_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, current_stream()));
	CEDAR_HIP_CHECK(hipStreamSynchronize(current_stream()));
}
void cedar_amd_memcpy_d2d(void *dst, const void *src, size_t bytes)
{
	CEDAR_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, current_stream()));
}
void cedar_amd_memset(void *dst, int value, size_t bytes)
{
	// zeroing whole doubles (every use the library itself makes) goes through its own kernel, see common.h zero_fill
	if (value == 0 && bytes % sizeof(real_t) == 0 && (uintptr_t)dst % sizeof(real_t) == 0)
		zero_fill(static_cast<real_t *>(dst), bytes / sizeof(real_t), current_stream());
	else
		CEDAR_HIP_CHECK(hipMemsetAsync(dst, value, bytes, current_stream()));
}
void cedar_amd_sync(void) { CEDAR_HIP_CHECK(hipStreamSynchronize(current_stream())); }
void cedar_amd_set_stream(void *s) { cedar_amd::g_stream = static_cast<hipStream_t>(s); }
void *cedar_amd_get_stream(void) { return cedar_amd::g_stream; }

void cedar_amd_matvec2(const real_t *so, const real_t *q, real_t *qf, len_t II, len_t JJ, int nstncl)
{
	const size_t P = (size_t)II * JJ;
	Staged sso(so, P * nstncl, true, false), sq(q, P, true, false), sqf(qf, P, true, true);
	matvec2(sso.get(), sq.get(), sqf.get(), (int)II, (int)JJ, nstncl, current_stream());
}

void cedar_amd_matvec3(const real_t *so, const real_t *q, real_t *qf, len_t II, len_t JJ, len_t KK, int nstncl)
{
	const size_t P = (size_t)II * JJ * KK;
	Staged sso(so, P * nstncl, true, false), sq(q, P, true, false), sqf(qf, P, true, true);
	matvec3(sso.get(), sq.get(), sqf.get(), (int)II, (int)JJ, (int)KK, nstncl, current_stream());
}

double cedar_amd_l2norm(const real_t *v, len_t II, len_t JJ, len_t KK)
{
	size_t n = (size_t)II * JJ * KK;
	Staged sv(v, n, true, false);
	real_t *scratch = static_cast<real_t *>(pool_get(4100 * sizeof(real_t)));
	sumsq_interior(sv.get(), (int)II, (int)JJ, (int)KK, scratch, scratch + 4096, current_stream());
	double ss = 0;
	cedar_amd_memcpy_d2h(&ss, scratch + 4096, sizeof(double));
	pool_put(scratch, 4100 * sizeof(real_t));
	return std::sqrt(ss);
}

void cedar_amd_gallery(int which, real_t *so, real_t *b, len_t nx, len_t ny, len_t nz, const double *params)
{
	static const int nst_of[13] = { 3, 3, 5, 0, 0, 0, 0, 0, 0, 0, 4, 4, 14 };
	const int which_in = which;
	if (which >= 100) which -= 100; /* placed inside a global grid, see gallery.hip */
	if (which < 0 || which > 12 || nst_of[which] == 0) {
		report("cedar_amd_gallery: unknown operator");
		return;
	}
	bool d3 = which >= 10;
	size_t npts = (size_t)(nx + 2) * (ny + 2) * (d3 ? nz + 2 : 1);
	Staged sso(so, npts * nst_of[which], false, true);
	Staged sb(b, npts, false, true);
	zero_fill(sso.get(), npts * nst_of[which], current_stream());
	if (b) zero_fill(sb.get(), npts, current_stream());
	gallery_fill(which_in, sso.get(), sb.get(), (int)nx, (int)ny, d3 ? (int)nz : 1, params, current_stream());
}

// ------------------------------------------------------------------ 2D drop-ins
void BMG2_SymStd_SETUP_recip(real_t *so, real_t *sor, len_t nx, len_t ny, int nstncl, int nsor_v)
{
	(void)nsor_v;
	size_t P = (size_t)nx * ny; // nx,ny are the array extents incl. ghosts (so.len())
	Staged sso(so, P * nstncl, true, false), ssor(sor, P * 2, true, true);
	setup_recip(sso.get() + KO * P, ssor.get() + P, nx, ny, 1, current_stream());
}

void BMG2_SymStd_relax_GS(int k, real_t *SO, real_t *QF, real_t *Q, real_t *SOR, len_t II, len_t JJ,
                          int kf, int ifd, int nstncl, int nsorv, int irelax_sym, int updown, int jpn)
{
	(void)nsorv;
	const int ipn = bc2(jpn, "BMG2_SymStd_relax_GS");
	if (ipn < 0) return;
	size_t P = (size_t)II * JJ;
	// reference branch: 9-point when K < KF or IFD != 1 (relax_GS.f90:89)
	int nst_eff = (k < kf || ifd != 1) ? 5 : 3;
	if (nst_eff > nstncl) nst_eff = nstncl;
	// NONSYM always uses the DOWN ordering (relax_GS.f90:78-87)
	int ud = (irelax_sym == 0) ? BMG_DOWN : updown;
	Staged sso(SO, P * nstncl, true, false), sqf(QF, P, true, false), sq(Q, P, true, true), ssor(SOR, P * 2, true, false);
	if (ipn == 0)
		relax2_gs(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)II, (int)JJ, nst_eff, ud, current_stream());
	else if (relax2_gs_per(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)II, (int)JJ, nst_eff, ud, ipn, current_stream()))
		report("BMG2_SymStd_relax_GS: periodic rows longer than 8192 points are not supported");
}

void BMG2_SymStd_residual(int *k, real_t *SO, real_t *QF, real_t *Q, real_t *RES, len_t *II, len_t *JJ,
                          int *kf, int *ifd, int *nstncl, int *ibc, int *irelax, int *irelax_sym, int *updown)
{
	(void)ibc; (void)irelax; (void)irelax_sym; (void)updown;
	size_t P = (size_t)(*II) * (*JJ);
	int nst_eff = (*k < *kf || *ifd != 1) ? 5 : 3;
	if (nst_eff > *nstncl) nst_eff = *nstncl;
	Staged sso(SO, P * (*nstncl), true, false), sqf(QF, P, true, false), sq(Q, P, true, false), sr(RES, P, true, true);
	residual2(sso.get(), sqf.get(), sq.get(), sr.get(), (int)*II, (int)*JJ, nst_eff, current_stream());
}

void BMG2_SymStd_SETUP_lines_x(real_t *SO, real_t *SOR, len_t Nx, len_t Ny, int NStncl, int JPN)
{
	const int ipn = bc2(JPN, "BMG2_SymStd_SETUP_lines_x");
	if (ipn < 0) return;
	size_t P = (size_t)Nx * Ny;
	Staged sso(SO, P * NStncl, true, false), ssor(SOR, P * 2, true, true);
	setup_lines_x(sso.get(), ssor.get(), (int)Nx, (int)Ny, current_stream(), ipn == 2 || ipn == 3);
}

void BMG2_SymStd_SETUP_lines_y(real_t *SO, real_t *SOR, len_t Nx, len_t Ny, int NStncl, int JPN)
{
	const int ipn = bc2(JPN, "BMG2_SymStd_SETUP_lines_y");
	if (ipn < 0) return;
	size_t P = (size_t)Nx * Ny;
	Staged sso(SO, P * NStncl, true, false), ssor(SOR, P * 2, true, true);
	setup_lines_y(sso.get(), ssor.get(), (int)Nx, (int)Ny, current_stream(), ipn == 1 || ipn == 3);
}

void BMG2_SymStd_relax_lines_x(int k, real_t *SO, real_t *QF, real_t *Q, real_t *SOR, real_t *B,
                               len_t II, len_t JJ, int kf, int ifd, int nstencil, int irelax_sym,
                               int updown, int jpn)
{
	(void)B;
	const int ipn = bc2(jpn, "BMG2_SymStd_relax_lines_x");
	if (ipn < 0) return;
	size_t P = (size_t)II * JJ;
	int nst_eff = (k < kf || ifd != 1) ? 5 : 3;
	if (nst_eff > nstencil) nst_eff = nstencil;
	int ud = (irelax_sym == 0) ? BMG_DOWN : updown;
	Staged sso(SO, P * nstencil, true, false), sqf(QF, P, true, false), sq(Q, P, true, true), ssor(SOR, P * 2, true, false);
	relax_lines_x(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)II, (int)JJ, nst_eff, ud, current_stream(), ipn);
}

void BMG2_SymStd_relax_lines_y(int k, real_t *SO, real_t *QF, real_t *Q, real_t *SOR, real_t *B,
                               len_t II, len_t JJ, int kf, int ifd, int nstencil, int irelax_sym,
                               int updown, int jpn)
{
	(void)B; // the reference's 2*JJ per-line scratch is too small for a whole colour: own HBM scratch
	const int ipn = bc2(jpn, "BMG2_SymStd_relax_lines_y");
	if (ipn < 0) return;
	size_t P = (size_t)II * JJ;
	int nst_eff = (k < kf || ifd != 1) ? 5 : 3;
	if (nst_eff > nstencil) nst_eff = nstencil;
	int ud = (irelax_sym == 0) ? BMG_DOWN : updown;
	size_t nscr = ylines_scratch_doubles((int)II, (int)JJ);
	real_t *scr = static_cast<real_t *>(pool_get(nscr * sizeof(real_t)));
	{
		Staged sso(SO, P * nstencil, true, false), sqf(QF, P, true, false), sq(Q, P, true, true), ssor(SOR, P * 2, true, false);
		relax_lines_y(sso.get(), sqf.get(), sq.get(), ssor.get(), scr, (int)II, (int)JJ, nst_eff, ud, current_stream(), ipn);
	}
	CEDAR_HIP_CHECK(hipStreamSynchronize(current_stream()));
	pool_put(scr, nscr * sizeof(real_t));
}

void BMG2_SymStd_restrict(real_t *Q, real_t *QC, real_t *CI, int Nx, int Ny, int Nxc, int Nyc, int jpn)
{
	const int ipn = bc2(jpn, "BMG2_SymStd_restrict");
	if (ipn < 0) return;
	size_t P = (size_t)Nx * Ny, PC = (size_t)Nxc * Nyc;
	// the periodic branch refreshes the fine vector's ghosts in place (restrict.f90:100-111)
	Staged sq(Q, P, true, ipn != 0), sqc(QC, PC, true, true), sci(CI, PC * 8, true, false);
	if (ipn == 0) restrict2(sq.get(), sqc.get(), sci.get(), Nx, Ny, Nxc, Nyc, current_stream());
	else restrict2_per(sq.get(), sqc.get(), sci.get(), Nx, Ny, Nxc, Nyc, ipn, current_stream());
}

void BMG2_SymStd_interp_add(real_t *Q, real_t *QC, real_t *RES, real_t *SO, real_t *CI,
                            len_t IIC, len_t JJC, len_t IIF, len_t JJF, int nstncl, int jpn)
{
	const int ipn = bc2(jpn, "BMG2_SymStd_interp_add");
	if (ipn < 0) return;
	size_t P = (size_t)IIF * JJF, PC = (size_t)IIC * JJC;
	Staged sq(Q, P, true, true), sqc(QC, PC, true, false), sr(RES, P, true, true),
	    sso(SO, P * nstncl, true, false), sci(CI, PC * 8, true, false);
	if (ipn == 0)
		interp_add2(sq.get(), sqc.get(), sr.get(), sso.get(), sci.get(), (int)IIC, (int)JJC, (int)IIF, (int)JJF, current_stream());
	else
		interp_add2_per(sq.get(), sqc.get(), sr.get(), sso.get(), sci.get(), (int)IIC, (int)JJC, (int)IIF, (int)JJF, ipn,
		                current_stream());
}

void BMG2_SymStd_SETUP_interp_OI(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf,
                                 len_t iic, len_t jjc, int ifd, int nstncl, int jpn, int irelax)
{
	(void)soc; (void)irelax;
	const int ipn = bc2(jpn, "BMG2_SymStd_SETUP_interp_OI");
	if (ipn < 0) return;
	size_t P = (size_t)iif * jjf, PC = (size_t)iic * jjc;
	Staged sso(so, P * nstncl, true, false), sci(ci, PC * 8, true, true);
	if (ipn == 0) setup_interp2(sso.get(), sci.get(), (int)iif, (int)jjf, (int)iic, (int)jjc, ifd, current_stream());
	else setup_interp2_per(sso.get(), sci.get(), (int)iif, (int)jjf, (int)iic, (int)jjc, ifd, ipn, current_stream());
}

void BMG2_SymStd_SETUP_ITLI_ex(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf,
                               len_t iic, len_t jjc, int ifd, int nstncl, int ipn)
{
	const int bc = bc2(ipn, "BMG2_SymStd_SETUP_ITLI_ex");
	if (bc < 0) return;
	size_t P = (size_t)iif * jjf, PC = (size_t)iic * jjc;
	Staged sso(so, P * nstncl, true, false), ssoc(soc, PC * 5, true, true), sci(ci, PC * 8, true, false);
	if (bc == 0) galerkin2(sso.get(), ssoc.get(), sci.get(), (int)iif, (int)jjf, (int)iic, (int)jjc, ifd, current_stream());
	else galerkin2_per(sso.get(), ssoc.get(), sci.get(), (int)iif, (int)jjf, (int)iic, (int)jjc, ifd, bc, current_stream());
}

static void cg_info(int *dinfo, const char *who)
{
	int info = 0;
	cedar_amd_memcpy_d2h(&info, dinfo, sizeof(int));
	if (info != 0) report(who);
}

void BMG2_SymStd_SETUP_cg_LU(real_t *so, len_t *ii, len_t *jj, int *nstncl, real_t *abd,
                             len_t *nabd1, len_t *nabd2, int *ibc)
{
	const int bc = bc2(*ibc, "BMG2_SymStd_SETUP_cg_LU");
	if (bc < 0) return;
	size_t P = (size_t)(*ii) * (*jj), NA = (size_t)(*nabd1) * (*nabd2);
	Staged sso(so, P * (*nstncl), true, false), sabd(abd, NA, true, true);
	int *dinfo = static_cast<int *>(pool_get(64));
	if (bc == 0) setup_cg2(sso.get(), (int)*ii, (int)*jj, *nstncl, sabd.get(), (int)*nabd1, (int)*nabd2, dinfo, current_stream());
	else setup_cg2_per(sso.get(), (int)*ii, (int)*jj, *nstncl, sabd.get(), (int)*nabd1, bc, dinfo, current_stream());
	cg_info(dinfo, "Coarse grid Cholesky decomp failed!");
	pool_put(dinfo, 64);
}

void BMG2_SymStd_SOLVE_cg(real_t *q, real_t *qf, len_t ii, len_t jj, real_t *abd, real_t *bbd,
                          len_t nabd1, len_t nabd2, int ibc)
{
	const int bc = bc2(ibc, "BMG2_SymStd_SOLVE_cg");
	if (bc < 0) return;
	size_t P = (size_t)ii * jj, NA = (size_t)nabd1 * nabd2;
	Staged sq(q, P, true, true), sqf(qf, P, true, false), sabd(abd, NA, true, false), sb(bbd, nabd2, false, true);
	if (bc == 0) solve_cg2(sq.get(), sqf.get(), (int)ii, (int)jj, sabd.get(), sb.get(), (int)nabd1, (int)nabd2, current_stream());
	else solve_cg2_per(sq.get(), sqf.get(), (int)ii, (int)jj, sabd.get(), sb.get(), (int)nabd1, bc, current_stream());
}

// ------------------------------------------------------------------ domain-decomposition pieces
void cedar_amd_relax3_pass(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                           int jb, int kb, int efirst)
{
	size_t P = (size_t)ii * jj * kk;
	Staged sso(so, P * 14, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax3_pass27(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, jb, kb, efirst, current_stream());
}

void cedar_amd_relax3_pass_part(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                                int jb, int kb, int efirst, int part)
{
	size_t P = (size_t)ii * jj * kk;
	Staged sso(so, P * 14, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax3_pass27(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, jb, kb, efirst, current_stream(), part);
}

void cedar_amd_relax2_pass(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int jb, int efirst)
{
	size_t P = (size_t)ii * jj;
	Staged sso(so, P * 5, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax2_pass9(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, jb, efirst, current_stream());
}

void cedar_amd_relax2_fixup(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int icol, int jb)
{
	size_t P = (size_t)ii * jj;
	Staged sso(so, P * 5, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax2_fixup9(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, icol, jb, current_stream());
}

void cedar_amd_relax2_colour5(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int jo)
{
	size_t P = (size_t)ii * jj;
	Staged sso(so, P * 3, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax2_colour5(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, jo, current_stream());
}

void cedar_amd_setup_interp2_phase(real_t *so, real_t *ci, len_t iif, len_t jjf, len_t iic, len_t jjc,
                                   int ifd, int nstncl, int phase, int ilo, int jlo)
{
	size_t P = (size_t)iif * jjf, PC = (size_t)iic * jjc;
	Staged sso(so, P * nstncl, true, false), sci(ci, PC * 8, true, true);
	setup_interp2_phase(sso.get(), sci.get(), (int)iif, (int)jjf, (int)iic, (int)jjc, ifd, phase, ilo, jlo, current_stream());
}

void cedar_amd_lines_rhs2(const real_t *so, const real_t *qf, const real_t *q, real_t *out, len_t ii, len_t jj,
                          int nstncl, int dir, int lb)
{
	lines_rhs2(so, qf, q, out, (int)ii, (int)jj, nstncl, dir, lb, current_stream());
}

void cedar_amd_lines_carry(real_t *y, const real_t *p, const real_t *c, int nlines, int n, int ld)
{
	lines_carry(y, p, c, nlines, n, ld, current_stream());
}

void cedar_amd_lines_store2(const real_t *in, real_t *q, len_t ii, len_t jj, int dir, int lb)
{
	lines_store2(in, q, (int)ii, (int)jj, dir, lb, current_stream());
}

void cedar_amd_affine_lines(real_t *y, const real_t *a, const real_t *div, int nlines, int n, int ld, int reverse)
{
	affine_lines(y, a, div, nlines, n, ld, reverse, current_stream());
}

void cedar_amd_relax3_planes(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                             int kb, int up, int part)
{
	size_t P = (size_t)ii * jj * kk;
	Staged sso(so, P * 14, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax3_planes27(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, kb, up, part, current_stream());
}

void cedar_amd_relax3_fixup(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk,
                            int icol, int jb, int kb)
{
	size_t P = (size_t)ii * jj * kk;
	Staged sso(so, P * 14, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax3_fixup27(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, icol, jb, kb, current_stream());
}

void cedar_amd_relax3_rows(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk, int j0, int jstep,
                           int nrj, int kb, int efirst)
{
	size_t P = (size_t)ii * jj * kk;
	Staged sso(so, P * 14, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax3_rows27(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, j0, jstep, nrj, kb, efirst, current_stream());
}

void cedar_amd_relax3_cols(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk, int jb, int kb,
                           int ncol, const int *cols, int xrow0, int xrow1)
{
	size_t P = (size_t)ii * jj * kk;
	Staged sso(so, P * 14, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax3_cols27(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, jb, kb, ncol, cols, xrow0, xrow1,
	              current_stream());
}

size_t cedar_amd_relax3_strip_doubles(len_t jj, len_t kk) { return relax3_strip_doubles((int)jj, (int)kk); }

void cedar_amd_relax3_strip_build(const real_t *so, const real_t *sor, len_t ii, len_t jj, len_t kk, int side, real_t *out)
{
	relax3_strip_build(so, sor, (int)ii, (int)jj, (int)kk, side, out, current_stream());
}

void cedar_amd_relax3_cols_strip(const real_t *strip_lo, const real_t *strip_hi, real_t *qf, real_t *q, len_t ii, len_t jj, len_t kk,
                                 int jb, int kb, int ncol, const int *cols, int xrow0, int xrow1)
{
	relax3_cols27_strip(strip_lo, strip_hi, qf, q, (int)ii, (int)jj, (int)kk, jb, kb, ncol, cols, xrow0, xrow1, current_stream());
}

int cedar_amd_relax3_planes_masked(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk, int kb,
                                   int up, unsigned cols_f, unsigned cols_s, const int *rows)
{
	if (!is_device_ptr(so) || !is_device_ptr(q) || !is_device_ptr(qf) || !is_device_ptr(sor)) return 0; // registered operators only
	if (((ii - 2) & 1) || ((jj - 2) & 1) || ii - 2 < 8) return 0;
	PsumSkip sk = psum_skip_none();
	sk.colsF = cols_f & 0xffu; sk.colsS = cols_s & 0xffu;
	for (int t = 0; t < 3; t++) sk.rows[t] = rows ? rows[t] : -1;
	return relax3_planes27_masked(so, qf, q, sor, (int)ii, (int)jj, (int)kk, kb, up, sk, current_stream()) ? 1 : 0;
}

int cedar_amd_relax3_prepare(const real_t *so, const real_t *sor, len_t ii, len_t jj, len_t kk)
{
	if (!is_device_ptr(so) || !is_device_ptr(sor)) return 0; // staged host arrays change address from call to call
	const char *e = getenv("CEDAR_AMD_ILV");
	const int mode = e ? atoi(e) : 320;
	// mode <= 0: no solve copy (the partial-sum scratch is still registered where the sweep wants it)
	return relax3_prepare(so, sor, (int)ii, (int)jj, (int)kk, mode <= 0 ? -1 : mode == 1 ? 0 : mode, current_stream());
}

int cedar_amd_relax3_prepare_rows(const real_t *so, const real_t *sor, len_t ii, len_t jj, len_t kk, int psum_min_rows)
{
	if (!is_device_ptr(so) || !is_device_ptr(sor)) return 0;
	const char *e = getenv("CEDAR_AMD_ILV");
	const int mode = e ? atoi(e) : 320;
	return relax3_prepare_rows(so, sor, (int)ii, (int)jj, (int)kk, mode <= 0 ? -1 : mode == 1 ? 0 : mode, psum_min_rows, current_stream());
}

void cedar_amd_relax3_release(const real_t *so) { relax3_release(so); }

int cedar_amd_relax2_gs_psum(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, int updown)
{
	const size_t P = (size_t)ii * jj;
	Staged sso(so, P * 5, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	const int took = relax2_psum_wanted((int)ii, (int)jj) ? 1 : 0;
	relax2_gs9_psum(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, updown, current_stream());
	return took;
}

int cedar_amd_relax3_gs_psum(real_t *so, real_t *qf, real_t *q, real_t *sor, real_t *scratch, len_t ii, len_t jj, len_t kk,
                             int updown)
{
	const size_t P = (size_t)ii * jj * kk;
	const int frun = relax3_psum_frun((int)jj);
	Staged sso(so, P * 14, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	if (!relax3_psum_ok((int)ii, (int)jj, (int)kk, frun)) {
		relax3_gs(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, 14, updown, current_stream());
		return 0;
	}
	real_t *T = scratch && is_device_ptr(scratch) ? scratch : static_cast<real_t *>(pool_get(P * sizeof(real_t)));
	relax3_gs27_psum(op3_cedar(sso.get(), ssor.get(), (int)ii, (int)jj, (int)kk), sqf.get(), sq.get(), T, (int)ii, (int)jj,
	                 (int)kk, updown, frun, current_stream());
	if (T != scratch) {
		CEDAR_HIP_CHECK(hipStreamSynchronize(current_stream()));
		pool_put(T, P * sizeof(real_t));
	}
	return 1;
}

void cedar_amd_relax3_colour7(real_t *so, real_t *qf, real_t *q, real_t *sor, len_t ii, len_t jj, len_t kk, int pts)
{
	size_t P = (size_t)ii * jj * kk;
	Staged sso(so, P * 4, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	relax3_colour7(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, pts, current_stream());
}

void cedar_amd_setup_interp3_phase(real_t *so, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                   len_t iic, len_t jjc, len_t kkc, int ifd, int nstncl, int phase,
                                   int ilo, int jlo, int klo)
{
	size_t P = (size_t)iif * jjf * kkf, PC = (size_t)iic * jjc * kkc;
	Staged sso(so, P * nstncl, true, false), sci(ci, PC * 26, true, true);
	setup_interp3_phase(sso.get(), sci.get(), (int)iif, (int)jjf, (int)kkf, (int)iic, (int)jjc, (int)kkc, ifd,
	                    phase, ilo, jlo, klo, current_stream());
}

void cedar_amd_box_copy(real_t *arr, len_t ii, len_t jj, len_t kk, int nplanes, int nboxes,
                        const int *boxes, const unsigned long long *offsets, real_t *buf, int unpack)
{
	box_copy(arr, (int)ii, (int)jj, (int)kk, nplanes, nboxes, boxes, offsets, buf, unpack, current_stream());
}

void cedar_amd_box_copy_strided(real_t *arr, len_t ii, len_t jj, len_t kk, int nplanes, int nboxes,
                                const int *boxes, const unsigned long long *offsets, real_t *buf, int unpack)
{
	box_copy(arr, (int)ii, (int)jj, (int)kk, nplanes, nboxes, boxes, offsets, buf, unpack, current_stream(), 1);
}

// ------------------------------------------------------------------ 3D drop-ins
void BMG3_SymStd_SETUP_recip(real_t *so, real_t *sor, len_t nx, len_t ny, len_t nz, int nstencl, int nsorv)
{
	(void)nsorv;
	size_t P = (size_t)nx * ny * nz;
	Staged sso(so, P * nstencl, true, false), ssor(sor, P * 2, true, true);
	setup_recip(sso.get() + KP * P, ssor.get() + P, nx, ny, nz, current_stream());
}

void BMG3_SymStd_relax_GS(int kg, real_t *so, real_t *qf, real_t *q, real_t *sor,
                          len_t ii, len_t jj, len_t kk, int ifd, int nstncl, int nsorv,
                          int irelax_sym, int updown, int jpn)
{
	(void)kg; (void)nsorv;
	if (!bc3(jpn, "BMG3_SymStd_relax_GS")) return;
	size_t P = (size_t)ii * jj * kk;
	int nst_eff = (ifd != 1) ? 14 : 4;
	if (nst_eff > nstncl) nst_eff = nstncl;
	// NONSYM always sweeps in the UP order in 3D (relax_GS.f90:85-94)
	int ud = (irelax_sym == 0) ? BMG_UP : updown;
	Staged sso(so, P * nstncl, true, false), sqf(qf, P, true, false), sq(q, P, true, true), ssor(sor, P * 2, true, false);
	if (jpn)
		relax3_gs_per(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, nst_eff, ud, jpn, current_stream());
	else
		relax3_gs(sso.get(), sqf.get(), sq.get(), ssor.get(), (int)ii, (int)jj, (int)kk, nst_eff, ud, current_stream());
}

void BMG3_SymStd_residual(int kg, int NOG, int ifd, real_t *q, real_t *qf, real_t *so, real_t *RES,
                          len_t ii, len_t jj, len_t kk, int NStncl)
{
	size_t P = (size_t)ii * jj * kk;
	int nst_eff = (kg < NOG || ifd != 1) ? 14 : 4; // residual.f90:67
	if (nst_eff > NStncl) nst_eff = NStncl;
	Staged sso(so, P * NStncl, true, false), sqf(qf, P, true, false), sq(q, P, true, false), sr(RES, P, true, true);
	residual3(sso.get(), sqf.get(), sq.get(), sr.get(), (int)ii, (int)jj, (int)kk, nst_eff, current_stream());
}

void BMG3_SymStd_restrict(real_t *q, real_t *qc, real_t *ci, len_t nx, len_t ny, len_t nz,
                          len_t nxc, len_t nyc, len_t nzc, int jpn)
{
	if (!bc3(jpn, "BMG3_SymStd_restrict")) return;
	size_t P = (size_t)nx * ny * nz, PC = (size_t)nxc * nyc * nzc;
	// periodic: the fine vector gets its ghosts refreshed first (restrict.f90:78-103), so it is an output too
	Staged sq(q, P, true, jpn != 0), sqc(qc, PC, true, true), sci(ci, PC * 26, true, false);
	if (jpn)
		restrict3_per(sq.get(), sqc.get(), sci.get(), (int)nx, (int)ny, (int)nz, (int)nxc, (int)nyc, (int)nzc, jpn, current_stream());
	else
		restrict3(sq.get(), sqc.get(), sci.get(), (int)nx, (int)ny, (int)nz, (int)nxc, (int)nyc, (int)nzc, current_stream());
}

void BMG3_SymStd_interp_add(real_t *q, real_t *qc, real_t *so, real_t *res, real_t *ci,
                            len_t iic, len_t jjc, len_t kkc, len_t iif, len_t jjf, len_t kkf,
                            int NStncl, int jpn)
{
	if (!bc3(jpn, "BMG3_SymStd_interp_add")) return;
	size_t P = (size_t)iif * jjf * kkf, PC = (size_t)iic * jjc * kkc;
	Staged sq(q, P, true, true), sqc(qc, PC, true, false), sso(so, P * NStncl, true, false),
	    sr(res, P, true, true), sci(ci, PC * 26, true, false);
	if (jpn)
		interp_add3_per(sq.get(), sqc.get(), sso.get(), sr.get(), sci.get(), (int)iic, (int)jjc, (int)kkc,
		                (int)iif, (int)jjf, (int)kkf, jpn, current_stream());
	else
		interp_add3(sq.get(), sqc.get(), sso.get(), sr.get(), sci.get(), (int)iic, (int)jjc, (int)kkc,
		            (int)iif, (int)jjf, (int)kkf, current_stream());
}

void BMG3_SymStd_SETUP_interp_OI(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                 len_t iic, len_t jjc, len_t kkc, int ifd, int nstncl, int irelax,
                                 int jpn, real_t *yo)
{
	(void)soc; (void)irelax; (void)yo;
	if (!bc3(jpn, "BMG3_SymStd_SETUP_interp_OI") || !even_periodic(jpn, iif, jjf, kkf, "BMG3_SymStd_SETUP_interp_OI")) return;
	size_t P = (size_t)iif * jjf * kkf, PC = (size_t)iic * jjc * kkc;
	Staged sso(so, P * nstncl, true, false), sci(ci, PC * 26, true, true);
	if (jpn)
		setup_interp3_per(sso.get(), sci.get(), (int)iif, (int)jjf, (int)kkf, (int)iic, (int)jjc, (int)kkc, ifd, jpn, current_stream());
	else
		setup_interp3(sso.get(), sci.get(), (int)iif, (int)jjf, (int)kkf, (int)iic, (int)jjc, (int)kkc, ifd, current_stream());
}

static void itli3(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                  len_t iic, len_t jjc, len_t kkc, int ipn, int nst)
{
	if (!bc3(ipn, "BMG3_SymStd_SETUP_ITLI_ex") || !even_periodic(ipn, iif, jjf, kkf, "BMG3_SymStd_SETUP_ITLI_ex")) return;
	size_t P = (size_t)iif * jjf * kkf, PC = (size_t)iic * jjc * kkc;
	Staged sso(so, P * nst, true, false), ssoc(soc, PC * 14, true, true), sci(ci, PC * 26, true, false);
	if (ipn)
		galerkin3_per(sso.get(), ssoc.get(), sci.get(), (int)iif, (int)jjf, (int)kkf, (int)iic, (int)jjc, (int)kkc,
		              nst == 4, ipn, current_stream());
	else
		galerkin3(sso.get(), ssoc.get(), sci.get(), (int)iif, (int)jjf, (int)kkf, (int)iic, (int)jjc, (int)kkc,
		          nst == 4, current_stream());
}

void BMG3_SymStd_SETUP_ITLI07_ex(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                 len_t iic, len_t jjc, len_t kkc, int ipn)
{
	itli3(so, soc, ci, iif, jjf, kkf, iic, jjc, kkc, ipn, 4);
}

void BMG3_SymStd_SETUP_ITLI27_ex(real_t *so, real_t *soc, real_t *ci, len_t iif, len_t jjf, len_t kkf,
                                 len_t iic, len_t jjc, len_t kkc, int ipn)
{
	itli3(so, soc, ci, iif, jjf, kkf, iic, jjc, kkc, ipn, 14);
}

void BMG3_SymStd_SETUP_cg_LU(real_t *so, len_t ii, len_t jj, len_t kk, int NStncl, real_t *abd,
                             len_t nabd1, len_t nabd2, int ibc)
{
	if (!bc3(ibc, "BMG3_SymStd_SETUP_cg_LU")) return;
	if ((NStncl != 14 && NStncl != 4) || (ibc && NStncl != 14)) { // the periodic branch knows 14 only (:235, :596)
		report("Cholesky decomp failed! (incorrect NStncl)");
		return;
	}
	size_t P = (size_t)ii * jj * kk, NA = (size_t)nabd1 * nabd2;
	if (ibc && ((size_t)nabd1 < (size_t)(ii - 2) * (jj - 2) * (kk - 2) || (size_t)nabd2 < (size_t)(ii - 2) * (jj - 2) * (kk - 2))) {
		report("BMG3_SymStd_SETUP_cg_LU: the periodic coarsest operator is dense, ABD(n,n) (include/cedar/3d/solver.h:118-121)");
		return;
	}
	Staged sso(so, P * NStncl, true, false), sabd(abd, NA, true, true);
	int *dinfo = static_cast<int *>(pool_get(64));
	if (ibc)
		setup_cg3_per(sso.get(), (int)ii, (int)jj, (int)kk, sabd.get(), (int)nabd1, ibc, dinfo, current_stream());
	else
		setup_cg3(sso.get(), (int)ii, (int)jj, (int)kk, NStncl, sabd.get(), (int)nabd1, (int)nabd2, dinfo, current_stream());
	cg_info(dinfo, "Coarse grid Cholesky decomp failed!");
	pool_put(dinfo, 64);
}

void BMG3_SymStd_SOLVE_cg(real_t *q, real_t *qf, len_t ii, len_t jj, len_t kk, real_t *abd,
                          real_t *bbd, len_t nabd1, len_t nabd2, int ibc)
{
	if (!bc3(ibc, "BMG3_SymStd_SOLVE_cg")) return;
	size_t P = (size_t)ii * jj * kk, NA = (size_t)nabd1 * nabd2;
	Staged sq(q, P, true, true), sqf(qf, P, true, false), sabd(abd, NA, true, false), sb(bbd, nabd2, false, true);
	if (ibc)
		solve_cg3_per(sq.get(), sqf.get(), (int)ii, (int)jj, (int)kk, sabd.get(), sb.get(), (int)nabd1, ibc, current_stream());
	else
		solve_cg3(sq.get(), sqf.get(), (int)ii, (int)jj, (int)kk, sabd.get(), sb.get(), (int)nabd1, (int)nabd2, current_stream());
}

} // extern "C"
