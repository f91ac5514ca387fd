/* TEST INFRASTRUCTURE ONLY (see oracle/Makefile).
 *
 * Host-side callbacks of the BoxMG Fortran ABI.  The Fortran kernels declare
 * them as imports (reference src/2d/ftn/ModInterface.f90:4-24): the *calling
 * program* supplies print_error / ftimer_begin / ftimer_end.  In the reference
 * the caller is Cedar's C++ layer (src/2d/ftn/interface.cc:8-21); here the
 * caller is the golden-vector generator, so it supplies them itself.
 * print_error only fires when a LAPACK factorisation reports INFO != 0.
 */
#include <stdio.h>

static int n_errors = 0;

void print_error(char *msg)
{
	++n_errors;
	fprintf(stderr, "[cedar_ref] %s\n", msg);
}

void ftimer_begin(char *label) { (void)label; }
void ftimer_end(char *label) { (void)label; }

int cedar_ref_error_count(void) { return n_errors; }
