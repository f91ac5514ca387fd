"""GPU parity tests proper: every BMG2_/BMG3_SymStd_* drop-in, called through the
C-ABI with host arrays (staged through HBM by the library), against
  (a) the golden vectors produced by the reference's own Fortran, and
  (b) the oracle on the same seeded inputs.

Tolerances (same as tests/test_oracle.py):
  bit-exact for relax / residual / restrict / interp_add / recip / interpolation
  set-up / line factorisation (the device code keeps the reference's operation
  order and is built with -ffp-contract=off);
  1e-13 (relative to the array's max-abs) for the Galerkin product (different
  association), the coarse Cholesky (vendor LAPACK in the golden) and the line
  solves, whose device algorithm is a parallel scan of the DPTTRS recurrence
  (1e-12: it re-associates ~n roundings per line).
"""
import numpy as np
import pytest

import cases
import problems as pb

pytestmark = pytest.mark.gpu

EXACT = {"recip", "relax0", "relax1", "residual", "setup_lines_x", "setup_lines_y",
         "interp", "restrict", "interp_add_q", "interp_add_res"}
RTOL = {"relax_lines_x0": 1e-12, "relax_lines_x1": 1e-12, "relax_lines_y0": 1e-12, "relax_lines_y1": 1e-12}


@pytest.fixture(scope="module")
def K():
    from cedar_amd import capi
    assert capi.device_count() >= 1, "no GPU visible"
    return capi.Kernels()


def check(name, got, want):
    key = name.split("/")[-1]
    if key in EXACT:
        assert np.array_equal(got, want), f"{name}: not bit-identical, max diff {np.max(np.abs(got - want))}"
    else:
        scale = np.max(np.abs(want)) + 1e-300
        tol = RTOL.get(key, 1e-13)
        assert np.max(np.abs(got - want)) <= tol * scale, f"{name}: {np.max(np.abs(got - want)) / scale}"


@pytest.mark.parametrize("case", cases.CASES_2D, ids=lambda c: c[0])
def test_kernels_2d_vs_golden(K, golden, case):
    out = cases.kernel_suite_2d(K, case)
    for k, v in out.items():
        check(f"{case[0]}/{k}", v, golden["k2"][f"{case[0]}/{k}"])


@pytest.mark.parametrize("case", cases.CASES_3D, ids=lambda c: c[0])
def test_kernels_3d_vs_golden(K, golden, case):
    out = cases.kernel_suite_3d(K, case)
    for k, v in out.items():
        check(f"{case[0]}/{k}", v, golden["k3"][f"{case[0]}/{k}"])


def test_reference_style_sweeps(K, golden):
    out = cases.sweep_suite(K)
    for k, v in out.items():
        assert np.array_equal(v, golden["sweeps"][k]), k


# larger / ragged shapes against the oracle (no golden): odd and even extents, rows longer
# than one workgroup pass, extents that exercise every fast-path width of the row kernels
EXTRA_2D = [("x_300x7_9", 300, 7, 5), ("x_1030x5_9", 1030, 5, 5), ("x_129x130_5", 129, 130, 3),
            ("x_3x3_9", 3, 3, 5), ("x_4x9_5", 4, 9, 3),
            # lines longer than one scan tile (2048 unknowns) and than one wavefront tile (512)
            ("x_4500x6_9", 4500, 6, 5), ("x_5x2300_9", 5, 2300, 5), ("x_600x7_5", 600, 7, 3)]
EXTRA_3D = [("x_70x9x8_27", 70, 9, 8, 14), ("x_130x6x7_27", 130, 6, 7, 14), ("x_258x5x6_27", 258, 5, 6, 14),
            ("x_33x34x35_7", 33, 34, 35, 4), ("x_3x3x3_27", 3, 3, 3, 14), ("x_4x5x3_7", 4, 5, 3, 4)]


@pytest.mark.parametrize("case", EXTRA_2D, ids=lambda c: c[0])
def test_kernels_2d_vs_oracle(K, oracle, case):
    got, want = cases.kernel_suite_2d(K, case), cases.kernel_suite_2d(oracle, case)
    for k in want:
        check(f"{case[0]}/{k}", got[k], want[k])


@pytest.mark.parametrize("case", EXTRA_3D, ids=lambda c: c[0])
def test_kernels_3d_vs_oracle(K, oracle, case):
    got, want = cases.kernel_suite_3d(K, case), cases.kernel_suite_3d(oracle, case)
    for k in want:
        check(f"{case[0]}/{k}", got[k], want[k])


@pytest.mark.parametrize("frun", [1, 2, 3, 4, 0])
@pytest.mark.parametrize("shape", [(20, 16, 5), (9, 17, 4), (70, 18, 6), (130, 19, 3), (11, 33, 7)], ids=str)
def test_plane_fused_relax_matches_four_pass_order(K, oracle, monkeypatch, shape, frun):
    """the plane-fused 27-point pass (relax27_plane + the deferred rows between runs) for every run
    length, odd and even ny/nz, both sweep directions: bit-identical to the reference order"""
    import problems as pb
    monkeypatch.setenv("CEDAR_AMD_FRUN", str(frun))
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    so = pb.random_op(g, 14, 41, zero_ghost=False)
    qf, q0 = pb.uniform(g, 42, -1, 1), pb.uniform(g, 43, -1, 1)
    sor = np.zeros((2,) + g)
    oracle.setup_recip3(so, sor)
    for ud in (0, 1):
        want, got = q0.copy(), q0.copy()
        oracle.relax3(so, qf, want, sor, ud)
        K.relax3(so, qf, got, sor, ud)
        assert np.array_equal(got, want), (shape, frun, ud, np.max(np.abs(got - want)))


@pytest.mark.parametrize("frun", [2, 3, 4, 8])
@pytest.mark.parametrize("shape", [(20, 16, 5), (9, 17, 4), (70, 18, 6), (130, 19, 3), (11, 33, 7), (257, 34, 8), (512, 20, 9),
                                   (3, 16, 16), (4, 9, 3)], ids=str)
def test_partial_sum_relax_matches_reference_order_to_rounding(K, oracle, monkeypatch, shape, frun):
    """cedar_amd_relax3_gs_psum (relax3d_psum.hip): the 27-point sweep whose second k-parity takes its inter-plane
    terms as two partial sums left by the first -- same products as BMG3_SymStd_relax_GS.f90:104-131, re-associated.
    Every run length, odd and even nx / ny / nz (ghost columns as sources, rows at run ends, planes next to a ghost
    plane), non-zero ghost cells and ghost operator entries, both directions, three sweeps in a row.
    Tolerance: 2e-14 of max|q| per sweep (a few roundings of a 27-term sum; the reference order is kept bit for bit
    by BMG3_SymStd_relax_GS, tested above)."""
    import problems as pb
    monkeypatch.setenv("CEDAR_AMD_FRUN", str(frun))
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    so = pb.random_op(g, 14, 71, zero_ghost=False)
    qf, q0 = pb.uniform(g, 72, -1, 1), pb.uniform(g, 73, -1, 1)
    sor = np.zeros((2,) + g)
    oracle.setup_recip3(so, sor)
    took = 0
    for ud in (0, 1):
        want, got = q0.copy(), q0.copy()
        for sweep in range(3):
            oracle.relax3(so, qf, want, sor, ud)
            took += K.relax3_psum(so, qf, got, sor, ud)
            scale = np.max(np.abs(want))
            assert np.max(np.abs(got - want)) <= 2e-14 * (sweep + 1) * scale, (shape, frun, ud, sweep, np.max(np.abs(got - want)) / scale)
        # ghost cells are never written
        m = np.ones(g, bool)
        m[1:-1, 1:-1, 1:-1] = False
        assert np.array_equal(got[m], q0[m])
    assert took == (6 if ny >= 4 * frun else 0)


@pytest.mark.parametrize("sides", [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1), (1, 1, 1, 1), (1, 0, 0, 1)], ids=str)
@pytest.mark.parametrize("shape,strip", [((16, 24, 6), True), ((8, 16, 5), False), ((130, 20, 4), True), ((512, 16, 4), True),
                                         ((512, 24, 3), False)], ids=str)
def test_boundary_first_pieces_reorder_the_sweep_without_changing_it(oracle, monkeypatch, shape, strip, sides):
    """cedar_amd_relax3_rows / _cols / _cols_strip / _planes_masked (the pieces of the distributed sweep on rank grids with
    an x / y split, dist3.cpp chain_parity) on ONE box without neighbours: the columns and rows that a neighbour on the
    chosen sides (-x, +x, -y, +y) would make chain points are relaxed ahead, stage by stage, the rest by the masked
    partial-sum launch.  The order is a valid order of the same Gauss-Seidel sweep, so the result equals the reference
    sweep to the rounding of the partial sums -- both directions, two sweeps in a row, with and without the dense column
    copy, neighbours on one, two and all four sides."""
    import ctypes as C
    import problems as pb
    from cedar_amd import capi
    lib = capi.lib
    monkeypatch.setenv("CEDAR_AMD_FRUN", "2")
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    II, JJ, KK = nx + 2, ny + 2, nz + 2
    so_h = pb.random_op(g, 14, 91, zero_ghost=False)
    qf_h, q0 = pb.uniform(g, 92, -1, 1), pb.uniform(g, 93, -1, 1)
    sor_h = np.zeros((2,) + g)
    oracle.setup_recip3(so_h, sor_h)
    so, sor, qf = (capi.DeviceArray.from_numpy(a) for a in (so_h, sor_h, qf_h))
    VP, U = C.c_void_p, C.c_uint
    lib.cedar_amd_relax3_prepare.argtypes = [VP, VP, U, U, U]
    lib.cedar_amd_relax3_release.argtypes = [VP]
    lib.cedar_amd_relax3_rows.argtypes = [VP, VP, VP, VP, U, U, U] + [C.c_int] * 5
    lib.cedar_amd_relax3_cols.argtypes = [VP, VP, VP, VP, U, U, U, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.cedar_amd_relax3_cols_strip.argtypes = [VP, VP, VP, VP, U, U, U, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.cedar_amd_relax3_planes_masked.argtypes = [VP, VP, VP, VP, U, U, U, C.c_int, C.c_int, U, U, C.POINTER(C.c_int)]
    lib.cedar_amd_relax3_strip_doubles.restype = C.c_size_t
    lib.cedar_amd_relax3_strip_doubles.argtypes = [U, U]
    lib.cedar_amd_relax3_strip_build.argtypes = [VP, VP, U, U, U, C.c_int, VP]
    assert lib.cedar_amd_relax3_prepare(so.ptr, sor.ptr, II, JJ, KK) & 2
    strips = [None, None]
    if strip:
        for side in (0, 1):
            if sides[side]:
                strips[side] = capi.DeviceArray((lib.cedar_amd_relax3_strip_doubles(JJ, KK),))
                lib.cedar_amd_relax3_strip_build(so.ptr, sor.ptr, II, JJ, KK, side, strips[side].ptr)

    def cols(q, jb, kb, cl, x0=-1, x1=-1):
        if not cl:
            return
        arr = (C.c_int * len(cl))(*cl)
        if strip:
            lib.cedar_amd_relax3_cols_strip(strips[0].ptr if strips[0] else None, strips[1].ptr if strips[1] else None, qf.ptr, q.ptr,
                                            II, JJ, KK, jb, kb, len(cl), arr, x0, x1)
        else:
            lib.cedar_amd_relax3_cols(so.ptr, qf.ptr, q.ptr, sor.ptr, II, JJ, KK, jb, kb, len(cl), arr, x0, x1)

    def parity(q, kb, up):  # dist3.cpp chain_parity without the exchanges
        jbF = 0 if up else 1
        c1, c2, colsS, fixc, mF, mS = [], [], [], [], 0, 0
        for side in (0, 1):
            if not sides[side]:
                continue
            col = (lambda d: nx - d) if side else (lambda d: 1 + d)
            bit = (lambda d: 1 << (7 - d)) if side else (lambda d: 1 << d)
            if (side == 0) == up:  # side P
                c1 += [col(0), col(2)]; c2 += [col(1)]; colsS += [col(0)]
                mF |= bit(0) | bit(1) | bit(2); mS |= bit(0)
            else:
                c1 += [col(1), col(3)]; c2 += [col(2), col(0)]; colsS += [col(1), col(0)]; fixc += [col(0)]
                mF |= bit(0) | bit(1) | bit(2) | bit(3); mS |= bit(0) | bit(1)
        rowsF, rowS = [], -1
        for side in (0, 1):
            if not sides[2 + side]:
                continue
            row = (lambda d: ny - d) if side else (lambda d: 1 + d)
            if (side == 0) == up:
                rowsF.append(row(0))
            else:
                rowsF.append(row(1)); rowS = row(0)
        skip = [rowsF[0] if rowsF else -1, rowsF[1] if len(rowsF) > 1 else -1, rowS]
        if rowsF:
            lib.cedar_amd_relax3_rows(so.ptr, qf.ptr, q.ptr, sor.ptr, II, JJ, KK, rowsF[0], rowsF[1] - rowsF[0] if len(rowsF) > 1 else 2,
                                      len(rowsF), kb, int(up))
        cols(q, jbF, kb, c1 + c2, skip[0], skip[1])
        cols(q, jbF, kb, fixc)
        if rowS >= 0:
            lib.cedar_amd_relax3_rows(so.ptr, qf.ptr, q.ptr, sor.ptr, II, JJ, KK, rowS, 2, 1, kb, int(up))
        cols(q, 1 - jbF, kb, colsS, rowS, -1)
        cols(q, 1 - jbF, kb, fixc)
        assert lib.cedar_amd_relax3_planes_masked(so.ptr, qf.ptr, q.ptr, sor.ptr, II, JJ, KK, kb, int(up), mF, mS, (C.c_int * 3)(*skip)) == 1

    try:
        for ud in (0, 1):  # BMG_DOWN, BMG_UP
            up = ud == 1
            want = q0.copy()
            q = capi.DeviceArray.from_numpy(q0)
            for sweep in range(2):
                oracle.relax3(so_h, qf_h, want, sor_h, 1 if up else 0)
                for c in range(2):
                    parity(q, c if up else 1 - c, up)
                got = q.numpy()
                scale = np.max(np.abs(want))
                assert np.max(np.abs(got - want)) <= 2e-14 * (sweep + 1) * scale, (shape, sides, ud, sweep, np.max(np.abs(got - want)) / scale)
    finally:
        lib.cedar_amd_relax3_release(so.ptr)


@pytest.mark.parametrize("frun", [2, 3, 4])
@pytest.mark.parametrize("shape", [(300, 48), (131, 33), (1100, 25), (260, 24), (514, 40), (1025, 64), (4096, 33)], ids=str)
def test_partial_sum_relax9_matches_reference_order_to_rounding(K, oracle, monkeypatch, shape, frun):
    """cedar_amd_relax2_gs_psum (relax9_band_psum): S rows between two F rows of a run take their six inter-row terms as
    one LDS partial-sum row; rows of one and of several chunks, odd and even nx / ny (ghost columns as sources, row 1 and
    the last row in the reference order), non-zero ghosts, both directions, three sweeps in a row: 2e-14 of max|q| per
    sweep against the oracle (BMG2_SymStd_relax_GS itself stays bit for bit, tested above)"""
    import problems as pb
    monkeypatch.setenv("CEDAR_AMD_FRUN2", str(frun))
    nx, ny = shape
    g = (ny + 2, nx + 2)
    so = pb.random_op(g, 5, 81, zero_ghost=False)
    qf, q0 = pb.uniform(g, 82, -1, 1), pb.uniform(g, 83, -1, 1)
    sor = np.zeros((2,) + g)
    oracle.setup_recip2(so, sor)
    took = 0
    for ud in (0, 1):
        want, got = q0.copy(), q0.copy()
        for sweep in range(3):
            oracle.relax2(so, qf, want, sor, ud)
            took += K.relax2_psum(so, qf, got, sor, ud)
            scale = np.max(np.abs(want))
            assert np.max(np.abs(got - want)) <= 2e-14 * (sweep + 1) * scale, (shape, frun, ud, sweep, np.max(np.abs(got - want)) / scale)
        m = np.ones(g, bool)
        m[1:-1, 1:-1] = False
        assert np.array_equal(got[m], q0[m])
    assert took == (6 if (ny >= 8 * frun and nx > 128) else 0)


@pytest.mark.parametrize("frun", [1, 2, 3, 0])
@pytest.mark.parametrize("shape", [(300, 48), (131, 33), (1100, 25), (260, 24)], ids=str)
def test_band_fused_relax9_matches_two_pass_order(K, oracle, monkeypatch, shape, frun):
    """the band-fused nine-point sweep (relax9_band + the deferred rows between runs) for several run lengths, odd and
    even ny, rows of one and of several chunks, both sweep directions: bit-identical to the reference order"""
    import problems as pb
    monkeypatch.setenv("CEDAR_AMD_FRUN2", str(frun))
    nx, ny = shape
    g = (ny + 2, nx + 2)
    so = pb.random_op(g, 5, 61, zero_ghost=False)
    qf, q0 = pb.uniform(g, 62, -1, 1), pb.uniform(g, 63, -1, 1)
    sor = np.zeros((2,) + g)
    oracle.setup_recip2(so, sor)
    for ud in (0, 1):
        want, got = q0.copy(), q0.copy()
        oracle.relax2(so, qf, want, sor, ud)
        K.relax2(so, qf, got, sor, ud)
        assert np.array_equal(got, want), (shape, frun, ud, np.max(np.abs(got - want)))


@pytest.mark.parametrize("shape", [(20, 16, 5), (9, 7, 4), (70, 18, 6), (11, 3, 7), (12, 9, 2)], ids=str)
def test_partial_row_class_passes_compose_to_the_full_pass(shape):
    """cedar_amd_relax3_pass_part: interior rows (part 1) then shell rows (part 2) of a row class equal the
    whole class (part 0) bit for bit, for every class and both i-colour orders, odd and even extents and
    grids too thin to have an interior -- the split the multi-GPU driver overlaps halo exchanges with"""
    import ctypes as C
    from cedar_amd import capi
    import problems as pb
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    so = pb.random_op(g, 14, 51, zero_ghost=False)
    qf, q0 = pb.uniform(g, 52, -1, 1), pb.uniform(g, 53, -1, 1)
    sor = np.zeros((2,) + g)
    capi.Kernels().setup_recip3(so, sor)
    u = C.c_uint
    f = capi.lib.cedar_amd_relax3_pass_part
    for jb in (0, 1):
        for kb in (0, 1):
            for efirst in (0, 1):
                whole = q0.copy()
                f(capi._p(so), capi._p(qf), capi._p(whole), capi._p(sor), u(nx + 2), u(ny + 2), u(nz + 2), jb, kb, efirst, 0)
                # with every face mask (which faces have a neighbouring rank: bit 0 -y, 1 +y, 2 -z, 3 +z; 0 = all)
                for sides in (0, 1, 2, 4, 8, 5, 10, 15):
                    parts = q0.copy()
                    for part in (1, 2):
                        f(capi._p(so), capi._p(qf), capi._p(parts), capi._p(sor), u(nx + 2), u(ny + 2), u(nz + 2), jb, kb, efirst,
                          part | (sides << 4))
                    assert np.array_equal(whole, parts), (shape, jb, kb, efirst, sides)
                assert not np.array_equal(whole, q0) or ((ny - jb + 1) // 2 == 0 or (nz - kb + 1) // 2 == 0)


@pytest.mark.parametrize("frun", [0, 1, 3])
@pytest.mark.parametrize("shape", [(20, 16, 6), (9, 17, 5), (70, 18, 4), (11, 33, 7)], ids=str)
def test_plane_parity_passes_compose_to_the_sweep(K, oracle, monkeypatch, shape, frun):
    """cedar_amd_relax3_planes (the unit of a slab-decomposed run): the two k-parities in sweep order equal
    the full sweep, interior planes + shell planes equal the whole parity; with and without the plane-fused
    kernel (CEDAR_AMD_FRUN), both directions; against the oracle bit for bit"""
    import ctypes as C
    from cedar_amd import capi
    import problems as pb
    monkeypatch.setenv("CEDAR_AMD_FRUN", str(frun))
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    so = pb.random_op(g, 14, 61, zero_ghost=False)
    qf, q0 = pb.uniform(g, 62, -1, 1), pb.uniform(g, 63, -1, 1)
    sor = np.zeros((2,) + g)
    oracle.setup_recip3(so, sor)
    u = C.c_uint
    f = capi.lib.cedar_amd_relax3_planes
    for up in (0, 1):
        want = q0.copy()
        oracle.relax3(so, qf, want, sor, up)
        whole = q0.copy()
        for c in range(2):
            kb = c if up else 1 - c
            f(capi._p(so), capi._p(qf), capi._p(whole), capi._p(sor), u(nx + 2), u(ny + 2), u(nz + 2), kb, up, 0)
        assert np.array_equal(whole, want), (shape, frun, up)
        for sides in (0, 4, 8, 12):  # which z faces have a neighbouring rank (0 = both)
            parts = q0.copy()
            for c in range(2):
                kb = c if up else 1 - c
                for part in (1, 2):
                    f(capi._p(so), capi._p(qf), capi._p(parts), capi._p(sor), u(nx + 2), u(ny + 2), u(nz + 2), kb, up,
                      part | (sides << 4))
            assert np.array_equal(parts, want), (shape, frun, up, sides)


def test_device_pointers_are_used_in_place(K):
    """the same entry points accept HBM pointers (no staging): results identical"""
    from cedar_amd import capi
    import problems as pb
    g = (12, 11, 14)
    so = pb.random_op(g, 14, 5)
    qf, q0 = pb.uniform(g, 6, -1, 1), pb.uniform(g, 7, -1, 1)
    sor = np.zeros((2,) + g)
    K.setup_recip3(so, sor)
    want = q0.copy()
    K.relax3(so, qf, want, sor, 1)
    d = [capi.DeviceArray.from_numpy(a) for a in (so, qf, q0, sor)]
    K.relax3(d[0], d[1], d[2], d[3], 1)
    assert np.array_equal(d[2].numpy(), want)


def test_unserved_boundary_codes_are_refused_loudly(K, capfd):
    """boundary codes the GPU path does not serve (the indefinite variants < 0) report through print_error and
    leave q untouched; the definite periodic codes are served (tests/test_gpu_periodic.py, test_gpu_periodic3d.py)"""
    import ctypes as C
    from cedar_amd import capi
    import problems as pb
    g = (8, 8)
    so, q = pb.random_op(g, 5, 1), pb.uniform(g, 2)
    q0 = q.copy()
    sor, qf = np.zeros((2,) + g), np.zeros(g)
    capi.lib.BMG2_SymStd_relax_GS(1, capi._p(so), capi._p(qf), capi._p(q), capi._p(sor), C.c_uint(8), C.c_uint(8),
                                  1, 0, 5, 2, 1, 0, -3)
    assert np.array_equal(q, q0)
    assert "boundary code -3 is not implemented" in capfd.readouterr().err
    g3 = (6, 6, 6)
    so3, q3 = pb.random_op(g3, 14, 1), pb.uniform(g3, 2)
    q30 = q3.copy()
    sor3, qf3 = np.zeros((2,) + g3), np.zeros(g3)
    capi.lib.BMG3_SymStd_relax_GS(1, capi._p(so3), capi._p(qf3), capi._p(q3), capi._p(sor3), C.c_uint(6), C.c_uint(6), C.c_uint(6),
                                  0, 14, 2, 1, 0, -8)
    assert np.array_equal(q3, q30)
    assert "boundary code -8 is not implemented" in capfd.readouterr().err


@pytest.mark.parametrize("shape,nst", [((9, 8, 7), 14), ((13, 9, 10), 4), ((33, 20, 17), 14), ((17, 12, 31), 4), ((3, 4, 5), 14),
                                       ((130, 6, 9), 14), ((8, 8, 8), 14), ((10, 7, 9), 4), ((34, 21, 16), 14), ((4, 4, 4), 14),
                                       ((6, 5, 4), 4), ((258, 6, 7), 14), ((132, 9, 6), 4)], ids=str)
def test_row_sum_galerkin_is_bit_identical_to_the_one_stage_kernels(K, monkeypatch, shape, nst):
    """the row-sum product (galerkin3_rows.hip, the default for 27-point fine operators) keeps the one-stage kernels'
    summation order: same coarse operator bit for bit (signs of zeros included), with the whole grid in one slab and
    with slabs of 1 and 3 coarse planes (ring of row-sum planes); an even nx takes the paired operator loads
    (row_group, with the terms outside the grid dropped by a select), CEDAR_AMD_GALERKIN_PAIRS=0 the 8-byte loads"""
    import problems as pb
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, nst, 5, zero_ghost=False)
    ci = pb.uniform((26,) + gc, 6, -1, 1)
    out = []
    for rows, slab, pairs in (("0", "32", "1"), ("1", "1000", "1"), ("1", "1", "1"), ("1", "3", "1"), ("1", "3", "0")):
        monkeypatch.setenv("CEDAR_AMD_GALERKIN_ROWS", rows)
        monkeypatch.setenv("CEDAR_AMD_GALERKIN_SLAB", slab)
        monkeypatch.setenv("CEDAR_AMD_GALERKIN_PAIRS", pairs)
        soc = np.zeros((14,) + gc)
        K.galerkin3(so, soc, ci)
        out.append(soc)
    assert np.any(out[0] != 0)
    for o in out[1:]:
        assert np.array_equal(out[0].view(np.int64), o.view(np.int64))


def _random_cases(nd, count, seed, lo, hi):
    rng = np.random.default_rng(seed)
    out = []
    for c in range(count):
        dims = [int(v) for v in rng.integers(lo, hi + 1, size=nd)]
        if nd == 2:
            nst = int(rng.choice([3, 5]))
            out.append(("rnd_%s_%d" % ("x".join(map(str, dims)), nst), dims[0], dims[1], nst))
        else:
            nst = int(rng.choice([4, 14]))
            out.append(("rnd_%s_%d" % ("x".join(map(str, dims)), nst), dims[0], dims[1], dims[2], nst))
    return out


@pytest.mark.parametrize("case", _random_cases(2, 16, 20261004, 3, 150), ids=lambda c: c[0])
def test_kernels_2d_random_shapes_vs_oracle(K, oracle, case):
    """seeded random extents (odd, even, down to 3) and both stencils through every 2D kernel of the suite"""
    got, want = cases.kernel_suite_2d(K, case), cases.kernel_suite_2d(oracle, case)
    for k in want:
        check(f"{case[0]}/{k}", got[k], want[k])


@pytest.mark.parametrize("case", _random_cases(3, 16, 20261005, 3, 36), ids=lambda c: c[0])
def test_kernels_3d_random_shapes_vs_oracle(K, oracle, case):
    got, want = cases.kernel_suite_3d(K, case), cases.kernel_suite_3d(oracle, case)
    for k in want:
        check(f"{case[0]}/{k}", got[k], want[k])


@pytest.mark.parametrize("shape", [(9, 8, 7), (12, 12, 12), (33, 20, 17), (20, 33, 40)], ids=str)
def test_lds_tiled_galerkin_is_bit_identical_to_the_slot_kernels(K, monkeypatch, shape):
    """the opt-in LDS-tiled product (galerkin3_tiled.hip) evaluates the same rap_slot<S> from a staged tile"""
    import problems as pb
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, 14, 5, zero_ghost=False)
    ci = pb.uniform((26,) + gc, 6, -1, 1)
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("CEDAR_AMD_GALERKIN_TILED", flag)
        soc = np.zeros((14,) + gc)
        K.galerkin3(so, soc, ci)
        out.append(soc)
    assert np.any(out[0] != 0) and np.array_equal(out[0], out[1])


@pytest.mark.parametrize("nx,ny,relax", [(4500, 40, "line-xy"), (3000, 36, "line-x"), (44, 2300, "line-y"), (2049, 700, "line-xy")], ids=str)
def test_line_solve_on_scan_ordered_factors_is_bit_identical(monkeypatch, nx, ny, relax):
    """the resident solver runs long lines on a scan-ordered copy of the factors (lines.hip line_pttrs_pf): same
    values, same chunks, same scan => the same bits as the kernels that read SOR directly (CEDAR_AMD_LINE_PERM=0)"""
    from cedar_amd import capi
    so, b = pb.aniso9(nx, ny), pb.rhs2(nx, ny)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("CEDAR_AMD_LINE_PERM", flag)
        s = capi.Solver(so, relax=relax, max_iter=3)
        x = np.zeros_like(b)
        h = s.solve(b, x)
        s.close()
        outs[flag] = (h, x)
    assert np.array_equal(outs["0"][0], outs["1"][0]) and np.array_equal(outs["0"][1], outs["1"][1])
    assert outs["1"][0][-1] < 0.5


def test_registered_solve_copy_gives_the_same_bits(K, monkeypatch):
    """cedar_amd_relax3_prepare: relax / residual on the row-interleaved copy of a registered device operator equal the
    Cedar-layout kernels bit for bit; freeing the operator through the library drops the registration"""
    import ctypes as C
    from cedar_amd import capi
    monkeypatch.setenv("CEDAR_AMD_ILV", "1")
    g = (11, 20, 37)
    so_h, b_h, x_h = pb.random_op(g, 14, 5), pb.uniform(g, 6, -1, 1), pb.uniform(g, 7, -1, 1)
    so, b = capi.DeviceArray.from_numpy(so_h), capi.DeviceArray.from_numpy(b_h)
    sor = capi.DeviceArray((2,) + g)
    K.setup_recip3(so, sor)
    outs = []
    for prepared in (False, True):
        if prepared:
            capi.lib.cedar_amd_relax3_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint]
            assert capi.lib.cedar_amd_relax3_prepare(so.ptr, sor.ptr, g[2], g[1], g[0]) & 1  # bit 0: solve copy, bit 1: partial-sum scratch
        x, r = capi.DeviceArray.from_numpy(x_h), capi.DeviceArray(g)
        K.relax3(so, b, x, sor, 0)
        K.relax3(so, b, x, sor, 1)
        K.residual3(so, b, x, r)
        outs.append((x.numpy(), r.numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    so.free()  # cedar_amd_free releases the registration with the operator


def test_release_scratch_between_galerkin_products(K):
    """cedar_amd_release_scratch frees the kept ring of row sums; the next product allocates it again"""
    import problems as pb
    from cedar_amd import capi
    g = (12, 14, 18)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, 14, 3, zero_ghost=False)
    ci = pb.uniform((26,) + gc, 4, -1, 1)
    a, b = np.zeros((14,) + gc), np.zeros((14,) + gc)
    K.galerkin3(so, a, ci)
    capi.release_scratch()
    capi.release_scratch()  # nothing kept: no-op
    K.galerkin3(so, b, ci)
    assert np.any(a != 0) and np.array_equal(a, b)
