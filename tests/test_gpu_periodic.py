"""2D periodic boundary conditions on the GPU path, through the C ABI with the boundary
code the reference's bindings pass (jpn / ibc = 1, 2, 3), against the golden vectors the reference's
Fortran produced (tests/golden/periodic2d.npz, solves_periodic.json) and against the oracle.
Bit-exact for relax / restrict / interp_add / interpolation; rounding-level for what goes through
the Galerkin product and the dense Cholesky."""
import json
import os

import numpy as np
import pytest

import cases
import problems as pb
from test_oracle_periodic import EXACT, check_kernels

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def K():
    from cedar_amd import capi
    assert capi.device_count() >= 1, "no GPU visible"
    return capi.Kernels()


@pytest.fixture(scope="module")
def gper():
    return np.load(os.path.join(HERE, "golden", "periodic2d.npz"))


@pytest.mark.parametrize("case", cases.CASES_PER, ids=lambda c: c[0])
def test_periodic_kernels_vs_golden(K, gper, case):
    check_kernels(case[0], cases.kernel_suite_per(K, case), gper)


@pytest.mark.parametrize("case", cases.CG_PER, ids=lambda c: c[0])
def test_periodic_coarse_solve_vs_golden(K, gper, case):
    check_kernels(case[0], cases.coarse_solve_per(K, case), gper)


EXTRA = [("x300x7_9_x", 300, 7, 5, 2), ("x1030x5_9_xy", 1030, 5, 5, 3), ("x129x130_5_y", 129, 130, 3, 1),
         ("x4x4_9_xy", 4, 4, 5, 3), ("x5x4_5_x", 5, 4, 3, 2), ("x600x33_5_xy", 600, 33, 3, 3)]


@pytest.mark.parametrize("case", EXTRA, ids=lambda c: c[0])
def test_periodic_kernels_vs_oracle(K, oracle, case):
    got, want = cases.kernel_suite_per(K, case), cases.kernel_suite_per(oracle, case)
    for k in want:
        if k in EXACT:
            assert np.array_equal(got[k], want[k]), (case[0], k, np.max(np.abs(got[k] - want[k])))
        else:
            assert np.max(np.abs(got[k] - want[k])) <= 1e-13 * np.max(np.abs(want[k])), (case[0], k)


@pytest.mark.parametrize("name", list(cases.SOLVES_PER), ids=str)
def test_periodic_solve_history_vs_reference_golden(name, oracle):
    """device-resident solver with ibc != 0 (hipGraph V-cycle): residual history of the reference's
    periodic example problem and of wrapped random operators, iteration for iteration"""
    from cedar_amd import capi
    gold = json.load(open(os.path.join(HERE, "golden", "solves_periodic.json")))[name]
    mk_op, mk_rhs, st = cases.SOLVES_PER[name]
    so, b = mk_op(), mk_rhs()
    s = capi.Solver(so, **st)
    assert s.nlevels() == gold["nlevels"]
    x = np.zeros_like(b)
    h = s.solve(b, x)
    want = [float(gold["res0_l2"])] + [float(v) for v in gold["rel_l2"]]
    assert len(h) == len(want)
    np.testing.assert_allclose(h, want, rtol=1e-10, atol=1e-12 if "line" in name else 1e-14)
    inner = x[1:-1, 1:-1]
    assert abs(float(np.sqrt(np.cumsum((inner * inner).ravel())[-1])) - float(gold["x_l2"])) <= 1e-11 * float(gold["x_l2"])
    # hierarchy against the oracle: interpolation bit-exact, coarse operators to rounding
    ml = oracle.ml_create(so, **st)
    for lvl in range(1, s.nlevels()):
        P, Po = s.array(lvl, "P"), ml.array(lvl, "P")
        assert np.array_equal(P, Po) or np.max(np.abs(P - Po)) <= 1e-13 * np.max(np.abs(Po)), lvl
        A, Ao = s.array(lvl, "A"), ml.array(lvl, "A")
        assert np.max(np.abs(A - Ao)) <= 1e-12 * np.max(np.abs(Ao)), lvl
    ml.close()
    s.close()


def test_periodic_solver_refuses_what_it_does_not_serve(capfd):
    from cedar_amd import capi
    with pytest.raises(RuntimeError):
        capi.Solver(pb.periodic_poisson2(32, 32, (True, False)), ibc=4)  # 4 is no boundary code (BMG_get_bc.f90:13-20)
    assert "periodic boundary conditions are implemented for the definite periodic codes" in capfd.readouterr().err


def test_periodic_long_lines(K, oracle):
    """cyclic lines longer than one scan tile (2048 unknowns) in both directions"""
    for case in (("xl4500x6_9_x", 4500, 6, 5, 2), ("xl5x2300_9_y", 5, 2300, 5, 1), ("xl700x600_5_xy", 700, 600, 3, 3)):
        got, want = cases.kernel_suite_per(K, case), cases.kernel_suite_per(oracle, case)
        for k in want:
            if k.startswith("relax_lines") or k.startswith("setup_lines"):
                tol = 0.0 if k.startswith("setup") else 1e-11
                assert np.max(np.abs(got[k] - want[k])) <= tol * np.max(np.abs(want[k])), (case[0], k)
