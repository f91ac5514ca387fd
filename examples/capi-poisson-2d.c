/* A plain C caller of Cedar's C interface (include/cedar/capi.h), linked against libcedar_amd.so:
 * 5-point Poisson on the unit square given vertex based, the README right-hand side
 * (reference examples/basic-2d-ser/poisson.cc:15-37), b = A x check and a solve.
 *   make capi-poisson-2d && ./capi-poisson-2d [n]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <cedar/capi.h>

int main(int argc, char **argv)
{
	unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 200;
	unsigned ln[1] = { n };
	const double h = 1.0 / (n + 1), pi = 3.14159265358979323846;

	bmg2_topo topo = bmg2_topo_create(MPI_COMM_WORLD, n, n, ln, ln, 1, 1);
	bmg2_operator op = bmg2_operator_create(topo);

	/* natural signs: 4/h^2 on the diagonal, -1/h^2 to W, E, S, N inside the domain */
	size_t cap = (size_t)5 * n * n, nv = 0;
	grid_coord_2d *co = malloc(cap * sizeof *co);
	double *va = malloc(cap * sizeof *va);
	for (unsigned j = 0; j < n; j++)
		for (unsigned i = 0; i < n; i++) {
			co[nv] = (grid_coord_2d){ i, j, BMG2_C }; va[nv++] = 4.0 / (h * h);
			if (i > 0)     { co[nv] = (grid_coord_2d){ i, j, BMG2_W }; va[nv++] = -1.0 / (h * h); }
			if (j > 0)     { co[nv] = (grid_coord_2d){ i, j, BMG2_S }; va[nv++] = -1.0 / (h * h); }
			/* E and N name the same couplings from the other end: set them too, as a vertex-based caller would */
			if (i + 1 < n) { co[nv] = (grid_coord_2d){ i, j, BMG2_E }; va[nv++] = -1.0 / (h * h); }
			if (j + 1 < n) { co[nv] = (grid_coord_2d){ i, j, BMG2_N }; va[nv++] = -1.0 / (h * h); }
		}
	bmg2_operator_set(op, (unsigned)nv, co, va);

	double *b = malloc((size_t)n * n * sizeof *b), *x = malloc((size_t)n * n * sizeof *x),
	       *ax = malloc((size_t)n * n * sizeof *ax);
	for (unsigned j = 0; j < n; j++)
		for (unsigned i = 0; i < n; i++) {
			double X = (i + 1) * h, Y = (j + 1) * h;
			b[(size_t)j * n + i] = 8 * pi * pi * sin(2 * pi * X) * sin(2 * pi * Y);
		}

	bmg2_solver slv = bmg2_solver_create(&op);
	bmg2_solver_run(slv, x, b);

	bmg2_operator_apply(op, x, ax);
	double r2 = 0, b2 = 0, emax = 0;
	for (unsigned j = 0; j < n; j++)
		for (unsigned i = 0; i < n; i++) {
			size_t p = (size_t)j * n + i;
			double X = (i + 1) * h, Y = (j + 1) * h, e = fabs(x[p] - sin(2 * pi * X) * sin(2 * pi * Y));
			r2 += (b[p] - ax[p]) * (b[p] - ax[p]);
			b2 += b[p] * b[p];
			if (e > emax) emax = e;
		}
	printf("n = %u  ||b - A x|| / ||b|| = %.3e  max |x - u_exact| = %.3e (discretisation error ~ h^2 = %.1e)\n",
	       n, sqrt(r2 / b2), emax, h * h);
	bmg_timer_save("capi-poisson-2d-timings.json");
	bmg2_solver_destroy(slv);
	bmg2_operator_destroy(op);
	free(co); free(va); free(b); free(x); free(ax);
	return sqrt(r2 / b2) < 1e-6 ? 0 : 1;
}
