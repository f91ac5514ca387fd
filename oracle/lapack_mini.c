/* TEST INFRASTRUCTURE ONLY -- see boxmg.h.
 *
 * The four LAPACK routines the BoxMG hot path calls.  LAPACK is a third-party
 * dependency of the reference that is NOT vendored under /root/reference
 * (system LAPACK, unpinned: CMakeLists.txt:49-50).  These restate the
 * published reference-LAPACK (netlib 3.x) algorithms:
 *   DPTTRF  -- L D L^T of an SPD tridiagonal        (dpttrf.f)
 *   DPTTRS  -- solve with that factorisation         (dpttrs.f -> dptts2.f)
 *   DPBTRF  -- band Cholesky, UPLO='U'; the coarsest grids here have N <= 64,
 *              below DPBTRF's block size (NB = 32 from ILAENV for N <= NBMAX
 *              path "use unblocked code" when NB <= 1 or NB > KD) -- so the
 *              unblocked DPBTF2 order is restated             (dpbtf2.f)
 *   DPBTRS  -- two DTBSV sweeps                      (dpbtrs.f, dtbsv.f)
 * Call sites in the reference: src/2d/ftn/BMG2_SymStd_SETUP_lines_x.f90:86,
 * relax_lines_x.f90:115,145, SETUP_cg_LU.f90:118, SOLVE_cg.f90:104 and the
 * 3D equivalents.  Element-wise parity of these steps is pinned by goldens
 * generated with the image's LAPACK (MKL) through oracle/_ref; agreement is
 * to rounding (vendor LAPACKs may reorder), tolerance stated in the tests.
 */
#include <math.h>
#include "boxmg.h"

int orc_dpttrf(int n, real_t *d, real_t *e)
{
	for (int i = 0; i < n - 1; i++) {
		if (d[i] <= 0.0) return i + 1;
		real_t ei = e[i];
		e[i] = ei / d[i];
		d[i + 1] = d[i + 1] - e[i] * ei;
	}
	if (n > 0 && d[n - 1] <= 0.0) return n;
	return 0;
}

void orc_dpttrs(int n, const real_t *d, const real_t *e, real_t *b)
{
	if (n <= 0) return;
	for (int i = 1; i < n; i++)
		b[i] = b[i] - b[i - 1] * e[i - 1];
	b[n - 1] = b[n - 1] / d[n - 1];
	for (int i = n - 2; i >= 0; i--)
		b[i] = b[i] / d[i] - b[i + 1] * e[i];
}

#define AB(r, c) ab[(size_t)((r)-1) + (size_t)ldab * (size_t)((c)-1)]

/* DPBTF2, UPLO = 'U': A = U^T U, AB(kd+1+i-j, j) = A(i,j) */
int orc_dpbtrf_upper(int n, int kd, real_t *ab, int ldab)
{
	int kld = ldab - 1 > 1 ? ldab - 1 : 1;
	for (int j = 1; j <= n; j++) {
		real_t ajj = AB(kd + 1, j);
		if (ajj <= 0.0) return j;
		ajj = sqrt(ajj);
		AB(kd + 1, j) = ajj;
		int kn = kd < n - j ? kd : n - j;
		if (kn > 0) {
			/* DSCAL(kn, 1/ajj, AB(kd,j+1), kld): row j of U right of the diagonal */
			real_t r = 1.0 / ajj;
			real_t *x = &AB(kd, j + 1);
			for (int t = 0; t < kn; t++)
				x[(size_t)t * kld] = r * x[(size_t)t * kld];
			/* DSYR('U', kn, -1, x, kld, AB(kd+1,j+1), kld) */
			real_t *a = &AB(kd + 1, j + 1);
			for (int c = 0; c < kn; c++) {
				real_t xc = x[(size_t)c * kld];
				if (xc != 0.0) {
					real_t temp = -1.0 * xc;
					for (int rr = 0; rr <= c; rr++)
						a[(size_t)rr + (size_t)c * kld] += x[(size_t)rr * kld] * temp;
				}
			}
		}
	}
	return 0;
}

/* DPBTRS, UPLO='U', NRHS=1: DTBSV('U','T','N') then DTBSV('U','N','N') */
void orc_dpbtrs_upper(int n, int kd, const real_t *ab, int ldab, real_t *b)
{
	int kplus1 = kd + 1;
	/* x := inv(U^T) x */
	for (int j = 1; j <= n; j++) {
		real_t temp = b[j - 1];
		int l = kplus1 - j;
		int i0 = j - kd > 1 ? j - kd : 1;
		for (int i = i0; i <= j - 1; i++)
			temp = temp - AB(l + i, j) * b[i - 1];
		temp = temp / AB(kplus1, j);
		b[j - 1] = temp;
	}
	/* x := inv(U) x */
	for (int j = n; j >= 1; j--) {
		if (b[j - 1] != 0.0) {
			int l = kplus1 - j;
			b[j - 1] = b[j - 1] / AB(kplus1, j);
			real_t temp = b[j - 1];
			int i0 = j - kd > 1 ? j - kd : 1;
			for (int i = j - 1; i >= i0; i--)
				b[i - 1] = b[i - 1] - temp * AB(l + i, j);
		}
	}
}

/* ---- dense Cholesky for the periodic coarse solve (BMG2_SymStd_SETUP_cg_LU.f90:214, SOLVE_cg.f90:107)
 * DPOTF2 / DPOTRS, UPLO = 'U', NRHS = 1, reference-BLAS operation order (dpotf2.f, dtrsm.f). */
#define A(r, c) a[(size_t)((r)-1) + (size_t)lda * (size_t)((c)-1)]
int orc_dpotrf_upper(int n, real_t *a, int lda)
{
	for (int j = 1; j <= n; j++) {
		real_t dot = 0.0;
		for (int i = 1; i <= j - 1; i++) dot = dot + A(i, j) * A(i, j);
		real_t ajj = A(j, j) - dot;
		if (ajj <= 0.0 || ajj != ajj) {
			A(j, j) = ajj;
			return j;
		}
		ajj = sqrt(ajj);
		A(j, j) = ajj;
		if (j < n) {
			/* DGEMV('T', j-1, n-j, -1, A(1,j+1), lda, A(1,j), 1, 1, A(j,j+1), lda) */
			for (int c = j + 1; c <= n; c++) {
				real_t temp = 0.0;
				for (int i = 1; i <= j - 1; i++) temp = temp + A(i, c) * A(i, j);
				A(j, c) = A(j, c) + (-1.0) * temp;
			}
			/* DSCAL(n-j, 1/ajj, A(j,j+1), lda) */
			real_t r = 1.0 / ajj;
			for (int c = j + 1; c <= n; c++) A(j, c) = r * A(j, c);
		}
	}
	return 0;
}

void orc_dpotrs_upper(int n, const real_t *a, int lda, real_t *b)
{
	/* DTRSM('L','U','T','N'): b := inv(U^T) b */
	for (int i = 1; i <= n; i++) {
		real_t temp = b[i - 1];
		for (int k = 1; k <= i - 1; k++) temp = temp - A(k, i) * b[k - 1];
		temp = temp / A(i, i);
		b[i - 1] = temp;
	}
	/* DTRSM('L','U','N','N'): b := inv(U) b */
	for (int k = n; k >= 1; k--) {
		if (b[k - 1] != 0.0) {
			b[k - 1] = b[k - 1] / A(k, k);
			for (int i = 1; i <= k - 1; i++) b[i - 1] = b[i - 1] - b[k - 1] * A(i, k);
		}
	}
}
#undef A
