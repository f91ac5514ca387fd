// temporary: launchers not yet implemented abort loudly
#include "common.h"
namespace cedar_amd {
#define TODO(name) { fprintf(stderr, "[cedar_amd] " name " not implemented yet\n"); abort(); }
void restrict2(const real_t *, real_t *, const real_t *, int, int, int, int, hipStream_t) TODO("restrict2")
void restrict3(const real_t *, real_t *, const real_t *, int, int, int, int, int, int, hipStream_t) TODO("restrict3")
void interp_add2(real_t *, const real_t *, real_t *, const real_t *, const real_t *, int, int, int, int, hipStream_t) TODO("interp_add2")
void interp_add3(real_t *, const real_t *, const real_t *, real_t *, const real_t *, int, int, int, int, int, int, hipStream_t) TODO("interp_add3")
void setup_interp2(const real_t *, real_t *, int, int, int, int, int, hipStream_t) TODO("setup_interp2")
void setup_interp3(const real_t *, real_t *, int, int, int, int, int, int, int, hipStream_t) TODO("setup_interp3")
void galerkin2(const real_t *, real_t *, const real_t *, int, int, int, int, int, hipStream_t) TODO("galerkin2")
void galerkin3(const real_t *, real_t *, const real_t *, int, int, int, int, int, int, int, hipStream_t) TODO("galerkin3")
void setup_lines_x(const real_t *, real_t *, int, int, hipStream_t) TODO("setup_lines_x")
void setup_lines_y(const real_t *, real_t *, int, int, hipStream_t) TODO("setup_lines_y")
void relax_lines_x(const real_t *, const real_t *, real_t *, const real_t *, int, int, int, int, hipStream_t) TODO("relax_lines_x")
void relax_lines_y(const real_t *, const real_t *, real_t *, const real_t *, int, int, int, int, hipStream_t) TODO("relax_lines_y")
void setup_cg2(const real_t *, int, int, int, real_t *, int, int, int *, hipStream_t) TODO("setup_cg2")
void solve_cg2(real_t *, const real_t *, int, int, const real_t *, real_t *, int, int, hipStream_t) TODO("solve_cg2")
void setup_cg3(const real_t *, int, int, int, int, real_t *, int, int, int *, hipStream_t) TODO("setup_cg3")
void solve_cg3(real_t *, const real_t *, int, int, int, const real_t *, real_t *, int, int, hipStream_t) TODO("solve_cg3")
void gallery_fill(int, real_t *, real_t *, int, int, int, const double *, hipStream_t) TODO("gallery_fill")
}
