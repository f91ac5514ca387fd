"""3D periodic boundary conditions on the GPU path, through the C ABI with the boundary code the
reference's bindings pass (jpn / ibc = 1, 2, 3, 5..8), against golden vectors from the reference's Fortran
(tests/golden/periodic3d.npz, solves_periodic3d.json: everything the reference defines consistently, see
oracle/boxmg3_per.c) and against the oracle.  Bit-exact for relax / restrict / interp_add / interpolation and the
dense Cholesky (same operation order as the oracle's unblocked LAPACK); rounding-level for the Galerkin product."""
import json
import os

import numpy as np
import pytest

import cases
import problems as pb
from test_oracle_periodic3d import EXACT, check_kernels3

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def K():
    from cedar_amd import capi
    assert capi.device_count() >= 1, "no GPU visible"
    return capi.Kernels()


@pytest.fixture(scope="module")
def gper3():
    return np.load(os.path.join(HERE, "golden", "periodic3d.npz"))


@pytest.mark.parametrize("case", cases.CASES_PER3, ids=lambda c: c[0])
def test_periodic3_kernels_vs_golden(K, gper3, case):
    check_kernels3(case[0], cases.kernel_suite_per3(K, case), gper3)


@pytest.mark.parametrize("case", cases.CG_PER3, ids=lambda c: c[0])
def test_periodic3_coarse_solve_vs_golden(K, gper3, case):
    check_kernels3(case[0], cases.coarse_solve_per3(K, case), gper3)


EXTRA3 = [("y40x36x30_27_xyz", 40, 36, 30, 14, 8), ("y130x6x8_27_x", 130, 6, 8, 14, 2), ("y6x130x8_7_y", 6, 130, 8, 4, 1),
          ("y9x6x66_27_z", 9, 6, 66, 14, 5), ("y34x32x36_7_xz", 34, 32, 36, 4, 6), ("y4x4x4_27_yz", 4, 4, 4, 14, 7),
          ("y600x4x6_27_xy", 600, 4, 6, 14, 3)]


@pytest.mark.parametrize("case", EXTRA3, ids=lambda c: c[0])
def test_periodic3_kernels_vs_oracle(K, oracle, case):
    got, want = cases.kernel_suite_per3(K, case), cases.kernel_suite_per3(oracle, case)
    for k in want:
        if k in EXACT:
            assert np.array_equal(got[k], want[k]), (case[0], k, np.max(np.abs(got[k] - want[k])))
        else:
            assert np.max(np.abs(got[k] - want[k])) <= 1e-13 * np.max(np.abs(want[k])), (case[0], k)


CG_EXTRA3 = [("e4x4x4_xyz", 4, 4, 4, 8), ("e5x3x4_xz", 5, 3, 4, 6), ("e6x4x3_xy", 6, 4, 3, 3), ("e8x8x8_xyz", 8, 8, 8, 8),
             ("e2x4x4_x", 2, 4, 4, 2)]


@pytest.mark.parametrize("case", CG_EXTRA3, ids=lambda c: c[0])
def test_periodic3_coarse_solve_vs_oracle(K, oracle, case):
    """the codes whose dense matrix the reference assembles wrongly (per_xz, per_xyz, per_xy with nx != ny): the
    periodic operator itself, bit for bit like the oracle (same DPOTF2 / DPOTRS operation order)"""
    got, want = cases.coarse_solve_per3(K, case), cases.coarse_solve_per3(oracle, case)
    assert np.array_equal(got["abd_upper"], want["abd_upper"]), np.max(np.abs(got["abd_upper"] - want["abd_upper"]))
    assert np.array_equal(got["q"], want["q"])


@pytest.mark.parametrize("name", list(cases.SOLVES_PER3), ids=str)
def test_periodic3_solve_history(name, oracle):
    """device-resident solver with a 3D periodic code: residual history against the reference-driven golden
    (per_z) and, for every code, iteration for iteration against the oracle; hierarchy against the oracle"""
    from cedar_amd import capi
    mk_op, mk_rhs, st = cases.SOLVES_PER3[name]
    so, b = mk_op(), mk_rhs()
    s = capi.Solver(so, **st)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    gold_all = json.load(open(os.path.join(HERE, "golden", "solves_periodic3d.json")))
    if name in gold_all:
        gold = gold_all[name]
        assert s.nlevels() == gold["nlevels"]
        want = [float(gold["res0_l2"])] + [float(v) for v in gold["rel_l2"]]
        assert len(h) == len(want)
        np.testing.assert_allclose(h, want, rtol=1e-10, atol=1e-14)
        inner = x[1:-1, 1:-1, 1:-1]
        assert abs(float(np.sqrt(np.cumsum((inner * inner).ravel())[-1])) - float(gold["x_l2"])) <= 1e-11 * float(gold["x_l2"])
    ml = oracle.ml_create(so, **st)
    xo = np.zeros_like(b)
    ho = ml.solve(b, xo, maxiter=10, tol=1e-8)
    assert len(h) == len(ho)
    np.testing.assert_allclose(h, ho, rtol=1e-10, atol=1e-14)
    assert np.max(np.abs(x - xo)) <= 1e-11 * np.max(np.abs(xo))
    assert np.array_equal(x, pb.wrap3(x.copy(), pb.per3_of(st["ibc"])))
    for lvl in range(1, s.nlevels()):
        P, Po = s.array(lvl, "P"), ml.array(lvl, "P")
        assert np.array_equal(P, Po) or np.max(np.abs(P - Po)) <= 1e-13 * np.max(np.abs(Po)), lvl
        A, Ao = s.array(lvl, "A"), ml.array(lvl, "A")
        assert np.max(np.abs(A - Ao)) <= 1e-12 * np.max(np.abs(Ao)), lvl
    ml.close()


def test_periodic3_refusals(capfd):
    """odd extent in a periodic direction on a level that is coarsened, the indefinite codes: reported through
    print_error, no solver (periodic F-cycles run since round 3: cases.SOLVES_PER3 *_f21)"""
    from cedar_amd import capi
    for so, kw in [(pb.periodic_random_op3(10, 16, 16, 14, (1, 0, 0), 1), dict(ibc=2)),    # nx 10 -> 5 -> 3: level 5 is coarsened
                   (pb.periodic_random_op3(8, 8, 8, 14, (1, 0, 0), 1), dict(ibc=-2))]:
        with pytest.raises(RuntimeError):
            capi.Solver(so, **kw)
        assert "periodic" in capfd.readouterr().err
    # the kernel drop-ins refuse an odd periodic extent in the set-up routines
    so = pb.periodic_random_op3(9, 8, 8, 14, (1, 0, 0), 1)
    gc = pb.coarse_shape(so.shape[1:])
    ci = np.zeros((26,) + gc)
    capi.Kernels().setup_interp3(so, ci, ibc=2)
    assert "even extent" in capfd.readouterr().err and not ci.any()
