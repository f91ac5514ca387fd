"""GPU parity of the device-resident multilevel solver (handle API of the C-ABI):
hierarchy (A, P, SOR, ABD per level) and iteration-for-iteration residual norms
against the golden histories of the reference's Fortran and against the oracle.

Tolerance on histories: north_star's 1e-10 relative, with the rounding floor of
r = b - A x as absolute tolerance (tests/cases.py HIST_ATOL; see test_oracle.py).
"""
import numpy as np
import pytest

import cases
import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from cedar_amd import capi
    assert capi.device_count() >= 1, "no GPU visible"
    return capi


@pytest.mark.parametrize("name", list(cases.SOLVES), ids=str)
def test_solve_history_vs_reference_golden(capi, golden, name):
    mk_op, mk_rhs, st = cases.SOLVES[name]
    gold = golden["solves"][name]
    so, b = mk_op(), mk_rhs()
    s = capi.Solver(so, **st)
    assert s.nlevels() == gold["nlevels"]
    for l in range(s.nlevels()):
        nx, ny, nz = s.dims(l)
        want = gold["level_dims"][l]
        assert [nx + 2, ny + 2, nz + 2][: len(want)] == want
    x = np.zeros_like(b)
    h = s.solve(b, x)
    s.close()
    want = [float(gold["res0_l2"])] + [float(v) for v in gold["rel_l2"]]
    assert len(h) == len(want)
    np.testing.assert_allclose(h, want, rtol=1e-10, atol=cases.HIST_ATOL.get(name, 1e-14))


@pytest.mark.parametrize("name", ["varcoef9_513_v21", "varcoef9_200x120_v21", "varcoef9_1024_v21", "varcoef9_200x120_f21"], ids=str)
def test_solve_history_with_partial_sum_relax9_vs_reference_golden(capi, golden, monkeypatch, name):
    """the resident 2D solver with the nine-point partial-sum sweep (relax9_band_psum; by default on levels with >= 4096
    rows -- none of the golden cases -- here FORCED onto every level with at least 8 runs of 2 rows): the reference's
    residual histories to 1e-10 relative.  Absolute floor 5e-14 (x ||r0||) instead of the 1e-14 of the reference-order
    solver: re-associating six terms of every second row's update perturbs x by ~1e-16 |x| per sweep on EVERY level, and
    thirty sweeps of that show as 2e-14 ||r0|| on the 1024^2 case (measured) -- rounding noise of the iterate itself,
    where the reference-order kernels contribute none (they are bit-identical) and only the set-up differs."""
    monkeypatch.setenv("CEDAR_AMD_FRUN2", "2")
    monkeypatch.setenv("CEDAR_AMD_PSUM", "1")
    mk_op, mk_rhs, st = cases.SOLVES[name]
    gold = golden["solves"][name]
    so, b = mk_op(), mk_rhs()
    s = capi.Solver(so, **st)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    s.close()
    want = [float(gold["res0_l2"])] + [float(v) for v in gold["rel_l2"]]
    np.testing.assert_allclose(h, want, rtol=1e-10, atol=5e-14)
    monkeypatch.setenv("CEDAR_AMD_PSUM", "0")
    s = capi.Solver(so, **st)
    x2 = np.zeros_like(b)
    s.solve(b, x2)
    s.close()
    assert not np.array_equal(x, x2) and np.max(np.abs(x - x2)) <= 1e-12 * np.max(np.abs(x2))


HIER = {
    "fe27_24x20x17_v21": (lambda: pb.fe3(24, 20, 17), lambda: pb.rhs3(24, 20, 17), dict(relax="point")),
    "fe27_40x33x50_v21": (lambda: pb.fe3(40, 33, 50), lambda: pb.rhs3(40, 33, 50), dict(relax="point")),
    "varcoef9_72x50_v21": (lambda: pb.varcoef9(72, 50), lambda: pb.rhs2(72, 50), dict(relax="point")),
}


@pytest.mark.parametrize("name", list(HIER), ids=str)
def test_solve_phase_on_the_reference_hierarchy(capi, name):
    """Where the late-cycle deviation of the histories comes from.  The reference's OWN set-up
    products (coarse operators, interpolation, relaxation data, factored coarsest operator: tests/golden/hier_*.npz,
    oracle/gen_golden.py hierarchy) are uploaded into the resident solver (cedar_amd_solver_set) and the SOLVE PHASE runs
    on the device.  No absolute floor here:
      * 3D 27-point: every cycle of the reference history to 1e-12 relative (measured: bit for bit) -- relax, residual,
        restriction and interpolation-and-add reproduce the reference's arithmetic exactly at every level; what the
        histories of test_solve_history_vs_reference_golden show beyond cycle 5 is the set-up (Galerkin association);
      * 2D: the same up to the coarsest-grid triangular solve, where the reference's LAPACK (MKL DPBTRS in this image;
        the reference pins none, CMakeLists.txt:49-50) and the netlib operation order the library restates differ by
        3e-16 of the coarse solution (oracle/gen_golden.py pins this: the only solve-phase kernel whose output differs
        on identical inputs) -- enough for 1e-10 relative once the residual has fallen seven decades, so the last
        cycles are held to the floor of the other history tests and the first four to 1e-12."""
    import os
    mk_op, mk_rhs, st = HIER[name]
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "hier_%s.npz" % name))
    so, b = mk_op(), mk_rhs()
    s = capi.Solver(so, **st)
    nlev = int(fx["nlev"])
    assert s.nlevels() == nlev
    x = np.zeros_like(b)
    h_own = s.solve(b, x)
    for l in range(nlev):
        if l > 0:
            s.set_array(l, "A", fx["A%d" % l])
            s.set_array(l, "P", fx["P%d" % l])
        if l < nlev - 1:
            s.set_array(l, "SOR0", fx["SOR0_%d" % l])
    s.set_array(0, "ABD", fx["abd"])
    x = np.zeros_like(b)
    h = s.solve(b, x)
    s.close()
    want = fx["hist"]
    assert len(h) == len(want)
    dev = np.abs(np.array(h) - want) / want
    dev_own = np.abs(np.array(h_own) - want) / want
    if so.ndim == 4:
        assert np.all(dev <= 1e-12), dev
        # the library's own set-up is what moves the late cycles (same solve kernels, same right-hand side)
        assert dev_own[-1] > 10 * max(dev[-1], 1e-16) or dev_own[-1] <= 1e-12, (dev_own, dev)
    else:
        assert np.all(dev[:5] <= 1e-12), dev
        np.testing.assert_allclose(h, want, rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("frun", [2, 4])
@pytest.mark.parametrize("name", ["fe27_40x33x50_v21", "fe27_65_v21", "fe27_129_v21", "fe27_40x33x50_f21"], ids=str)
def test_solve_history_with_partial_sum_relax_vs_reference_golden(capi, golden, monkeypatch, name, frun):
    """the resident solver with the partial-sum relax sweep (relax3d_psum.hip; by default on levels with >= 320 rows,
    here forced onto every level with at least 4 runs of `frun` rows): the reference's residual histories with the
    SAME tolerances as the reference-order solver above"""
    monkeypatch.setenv("CEDAR_AMD_FRUN", str(frun))
    monkeypatch.setenv("CEDAR_AMD_PSUM", "1")
    mk_op, mk_rhs, st = cases.SOLVES[name]
    gold = golden["solves"][name]
    so, b = mk_op(), mk_rhs()
    s = capi.Solver(so, **st)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    s.close()
    want = [float(gold["res0_l2"])] + [float(v) for v in gold["rel_l2"]]
    np.testing.assert_allclose(h, want, rtol=1e-10, atol=cases.HIST_ATOL.get(name, 1e-14))
    # and it is not the reference-order sweep in disguise: some iterate differs in the last bits
    monkeypatch.setenv("CEDAR_AMD_PSUM", "0")
    s = capi.Solver(so, **st)
    x2 = np.zeros_like(b)
    h2 = s.solve(b, x2)
    s.close()
    np.testing.assert_allclose(h2, want, rtol=1e-10, atol=cases.HIST_ATOL.get(name, 1e-14))
    assert not np.array_equal(x, x2)
    assert np.max(np.abs(x - x2)) <= 1e-12 * np.max(np.abs(x2))


@pytest.mark.parametrize("name", ["varcoef9_200x120_v21", "fe27_40x33x50_v21", "poisson7_64_v21",
                                  "stretch5_800x200_linex", "poisson5_400_v11"], ids=str)
def test_hierarchy_vs_oracle(capi, oracle, name):
    """every level's operator, interpolation and relaxation data against the oracle's set-up"""
    mk_op, _, st = cases.SOLVES[name]
    so = mk_op()
    s = capi.Solver(so, **st)
    ml = oracle.ml_create(so, **st)
    try:
        assert s.nlevels() == ml.nlevels()
        for l in range(s.nlevels()):
            assert s.dims(l) == ml.dims(l)
            for what in ("A", "P", "SOR0"):
                a, w = s.array(l, what), ml.array(l, what)
                if w is None:
                    continue
                if l == s.nlevels() - 1 and what == "SOR0":
                    continue  # coarsest level has no relaxation set-up
                scale = np.max(np.abs(w)) + 1e-300
                assert np.max(np.abs(a - w)) <= 1e-12 * scale, (l, what, np.max(np.abs(a - w)) / scale)
        a, w = s.array(0, "ABD"), ml.array(0, "ABD")
        assert np.max(np.abs(a - w)) <= 1e-12 * np.max(np.abs(w))
    finally:
        s.close()
        ml.close()


def test_vcycle_device_resident_and_graph_replay(capi, oracle):
    """cycle->run(x,b) on HBM-resident x,b (hipGraph replay) == oracle V-cycle, several cycles"""
    so, b = pb.fe3(33, 30, 29), pb.rhs3(33, 30, 29)
    s = capi.Solver(so)
    ml = oracle.ml_create(so)
    dx, db = capi.DeviceArray(b.shape), capi.DeviceArray.from_numpy(b)
    x = np.zeros_like(b)
    for _ in range(4):
        s.vcycle(dx, db)
        ml.vcycle(x, b)
        got = dx.numpy()
        assert np.max(np.abs(got - x)) <= 1e-12 * np.max(np.abs(x))
    s.close()
    ml.close()


def test_reference_acceptance_2d(capi):
    """test/2d/test_poisson.cc:64-93: 200^2, defaults: ||r||_2 < 1e-8 within 10 cycles, error < 1e-4"""
    so, b = pb.poisson2(200, 200), pb.rhs2(200, 200)
    s = capi.Solver(so)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    s.close()
    assert h[-1] * h[0] < 1e-8
    assert np.max(np.abs((pb.exact2(200, 200) - x)[1:-1, 1:-1])) < 1e-4


def test_gallery_on_device_matches_reference_generators(capi):
    """device-side gallery == numpy restatement of src/{2d,3d}/gallery.cc (bit-exact operators)"""
    for name, n, ref in (("poisson2", (37, 20), pb.poisson2(37, 20)), ("fe2", (9, 13), pb.fe2(9, 13)),
                         ("poisson3", (9, 10, 11), pb.poisson3(9, 10, 11)), ("fe3", (7, 8, 9), pb.fe3(7, 8, 9))):
        so, b = capi.gallery(name, n)
        assert np.array_equal(so.numpy(), ref), name
    so, _ = capi.gallery("diag_diffusion2", (30, 12), params=(1.0, 1e-4), with_rhs=False)
    assert np.array_equal(so.numpy(), pb.diag_diffusion2(30, 12, 1.0, 1e-4))
    _, b = capi.gallery("poisson2", (40, 40))
    np.testing.assert_allclose(b.numpy(), pb.rhs2(40, 40), rtol=1e-14, atol=1e-18)


def test_round_trip_properties_at_benchmark_scale(capi):
    """size-independent properties on a grid too large for goldens (256^3 27-pt, on device):
    (1) linearity of the residual in x, (2) restrict is the transpose of interpolation:
    <P^T r, e_c> = <r, P e_c>, (3) a relax sweep leaves ghost cells untouched."""
    n = 192
    so, b = capi.gallery("fe3", (n, n, n))
    s = capi.Solver(so, share_operator=True)
    K = capi.Kernels()
    g = (n + 2,) * 3
    x = capi.DeviceArray.from_numpy(pb.uniform(g, 11, -1, 1))
    r1, r2 = capi.DeviceArray(g), capi.DeviceArray(g)
    zero = capi.DeviceArray(g)
    K.residual3(so, b, x, r1)       # b - A x
    K.residual3(so, zero, x, r2)    # -A x
    d = r1.numpy() - r2.numpy() - b.numpy()
    assert np.max(np.abs(d[1:-1, 1:-1, 1:-1])) <= 1e-12
    # ghost cells survive a sweep bit-for-bit
    sor = capi.DeviceArray((2,) + g)
    K.setup_recip3(so, sor)
    x0 = x.numpy()
    K.relax3(so, b, x, sor, 1)
    x1 = x.numpy()
    m = pb.interior_mask(g)
    assert np.array_equal(x1[~m], x0[~m])
    assert not np.array_equal(x1[m], x0[m])
    s.close()


@pytest.mark.parametrize("shape,relax,op,cycle", [((130, 77), "line-xy", "aniso9", "v"), ((64, 301), "line-xy", "aniso9", "v"),
                                                  ((200, 800), "line-y", "stretch5", "v"), ((700, 90), "line-y", "aniso9", "v"),
                                                  ((96, 130), "line-xy", "aniso9", "f"), ((257, 33), "line-y", "stretch5", "f")], ids=str)
def test_y_lines_on_transposed_arrays_match_the_gather_pipeline(capi, monkeypatch, shape, relax, op, cycle):
    """the resident solver runs y-line sweeps through the x-line kernel on transposed arrays (lines.hip
    relax_lines_yt); same right-hand-side term order and the same scan as relax_lines_y, so the iterates are
    bit-identical to the gather / solve / scatter pipeline (CEDAR_AMD_YLINES_TRANSPOSED=0), which the kernel
    tests pin against the reference"""
    nx, ny = shape
    so = pb.aniso9(nx, ny) if op == "aniso9" else pb.diag_diffusion2(nx, ny, 1e-2, 1.0)
    b = pb.rhs2(nx, ny)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("CEDAR_AMD_YLINES_TRANSPOSED", flag)
        s = capi.Solver(so, relax=relax, nrelax_pre=2, nrelax_post=1, cycle=cycle)
        x = np.zeros_like(b)
        h = s.solve(b, x)
        s.close()
        out[flag] = (np.array(h), x)
    d = np.argwhere(out["0"][1] != out["1"][1])
    assert len(d) == 0, (len(d), d[:6].tolist(), [(float(out["0"][1][tuple(i)]), float(out["1"][1][tuple(i)])) for i in d[:6]])
    assert np.array_equal(out["0"][0], out["1"][0])


@pytest.mark.parametrize("shape,relax,op,cycle", [((64, 64), "line-xy", "aniso9", "v"), ((40, 33), "line-xy", "stretch5", "v"),
                                                  ((17, 64), "line-x", "aniso9", "v"), ((63, 9), "line-y", "aniso9", "v"),
                                                  ((130, 77), "line-xy", "aniso9", "v"), ((96, 130), "line-xy", "aniso9", "f"),
                                                  ((3, 5), "line-xy", "aniso9", "v")], ids=str)
def test_small_level_line_sweeps_match_the_per_colour_kernels(capi, oracle, monkeypatch, shape, relax, op, cycle):
    """levels of at most 64 x 64 unknowns run every line sweep of a visit in one launch with one lane per line solve
    (lines_small.hip, DPTTRS's order); the per-colour kernels (CEDAR_AMD_LINES_SMALL=0) solve the same lines with a
    scan: same iterates up to the association of the line solves, and both follow the oracle's history"""
    nx, ny = shape
    so = pb.aniso9(nx, ny) if op == "aniso9" else pb.diag_diffusion2(nx, ny, 1e-2, 1.0)
    b = pb.rhs2(nx, ny)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("CEDAR_AMD_LINES_SMALL", flag)
        s = capi.Solver(so, relax=relax, nrelax_pre=2, nrelax_post=1, cycle=cycle)
        x = np.zeros_like(b)
        h = s.solve(b, x)
        s.close()
        out[flag] = (np.array(h), x)
    ml = oracle.ml_create(so, relax=relax, cycle=cycle)
    xo = np.zeros_like(b)
    ho = np.array(ml.solve(b, xo))
    ml.close()
    for flag in ("0", "1"):
        assert len(out[flag][0]) == len(ho)
        np.testing.assert_allclose(out[flag][0], ho, rtol=1e-10, atol=1e-12)
        assert np.max(np.abs(out[flag][1] - xo)) <= 1e-10 * np.max(np.abs(xo))
    assert np.max(np.abs(out["0"][1] - out["1"][1])) <= 1e-11 * np.max(np.abs(xo))


@pytest.mark.parametrize("shape,op,cycle", [((64, 64), "aniso9", "v"), ((40, 33), "stretch5", "v"), ((130, 77), "aniso9", "v"),
                                            ((96, 130), "stretch5", "f"), ((3, 5), "aniso9", "v"), ((512, 512), "stretch5", "v")], ids=str)
def test_small_level_point_sweeps_are_bit_identical_to_the_per_colour_kernels(capi, monkeypatch, shape, op, cycle):
    """levels of at most 64 x 64 unknowns run the colours of every point sweep of a visit in one launch
    (lines_small.hip points_small_kernel): same expression per point, so the iterates equal those of the per-colour
    launches (CEDAR_AMD_LINES_SMALL=0) bit for bit"""
    nx, ny = shape
    so = pb.aniso9(nx, ny) if op == "aniso9" else pb.diag_diffusion2(nx, ny, 1e-2, 1.0)
    b = pb.rhs2(nx, ny)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("CEDAR_AMD_LINES_SMALL", flag)
        s = capi.Solver(so, relax="point", nrelax_pre=2, nrelax_post=1, cycle=cycle, max_iter=4)
        x = np.zeros_like(b)
        h = s.solve(b, x)
        s.close()
        out[flag] = (np.array(h), x)
    assert np.array_equal(out["0"][1].view(np.int64), out["1"][1].view(np.int64))
    assert np.array_equal(out["0"][0], out["1"][0])
    assert len(out["1"][0]) >= 2 and np.all(np.isfinite(out["1"][0]))  # (point relaxation does not converge on aniso9: not the point)


def _random_2d_solver_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        nx, ny = (int(v) for v in rng.integers(3, 75, size=2))
        relax = str(rng.choice(["point", "line-x", "line-y", "line-xy"]))
        op = str(rng.choice(["aniso9", "stretch5", "fe2"]))
        cycle = str(rng.choice(["v", "v", "f"]))
        out.append((nx, ny, relax, op, cycle))
    return out


@pytest.mark.parametrize("nx,ny,relax,op,cycle", _random_2d_solver_cases(24, 20261004), ids=str)
def test_random_small_2d_solvers_follow_the_oracle(capi, oracle, nx, ny, relax, op, cycle):
    """seeded random extents around the 64 x 64 limit of the one-launch-per-visit kernels (levels on both sides of it in
    one hierarchy, odd and even extents, lines of one to 74 unknowns): same level count and residual history as the oracle"""
    so = pb.aniso9(nx, ny) if op == "aniso9" else pb.fe2(nx, ny) if op == "fe2" else pb.diag_diffusion2(nx, ny, 1e-2, 1.0)
    b = pb.rhs2(nx, ny)
    s = capi.Solver(so, relax=relax, cycle=cycle, max_iter=5)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    nl = s.nlevels()
    s.close()
    ml = oracle.ml_create(so, relax=relax, cycle=cycle)
    assert nl == ml.nlevels()
    xo = np.zeros_like(b)
    ho = ml.solve(b, xo, maxiter=5)
    ml.close()
    assert len(h) == len(ho)
    np.testing.assert_allclose(h, ho, rtol=1e-9, atol=1e-12)
    assert np.max(np.abs(x - xo)) <= 1e-9 * max(np.max(np.abs(xo)), 1e-300)
