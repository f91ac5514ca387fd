"""2D periodic boundary conditions: the oracle's restatement of the reference's periodic
branches against golden vectors produced by the reference's own Fortran (oracle/gen_golden.py
main_periodic; tests/golden/periodic2d.npz, solves_periodic.json).  Kernel inputs carry random ghost
layers (each kernel is a deterministic function of its arrays).  Bit-exact except what goes through
the dense Cholesky (MKL in the golden vs the unblocked netlib order of the oracle) and the Galerkin
product (association)."""
import json
import os

import numpy as np
import pytest

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
EXACT = {"relax0", "relax1", "interp", "restrict_qc", "restrict_q", "interp_add_q", "interp_add_res",
         "setup_lines_x", "setup_lines_y"}


@pytest.fixture(scope="module")
def gper():
    return np.load(os.path.join(HERE, "golden", "periodic2d.npz"))


def check_kernels(name, got, gper):
    for k, v in got.items():
        want = gper[f"{name}/{k}"]
        if k in EXACT:
            assert np.array_equal(v, want), (name, k, np.max(np.abs(v - want)))
        else:
            tol = 1e-12 if (k == "q" or k.startswith("relax_lines")) else 1e-13
            assert np.max(np.abs(v - want)) <= tol * np.max(np.abs(want)), (name, k)


@pytest.mark.parametrize("case", cases.CASES_PER, ids=lambda c: c[0])
def test_periodic_kernels_vs_golden(oracle, gper, case):
    check_kernels(case[0], cases.kernel_suite_per(oracle, case), gper)


@pytest.mark.parametrize("case", cases.CG_PER, ids=lambda c: c[0])
def test_periodic_coarse_solve_vs_golden(oracle, gper, case):
    check_kernels(case[0], cases.coarse_solve_per(oracle, case), gper)


@pytest.mark.parametrize("name", list(cases.SOLVES_PER), ids=str)
def test_periodic_solve_history_vs_golden(oracle, name):
    gold = json.load(open(os.path.join(HERE, "golden", "solves_periodic.json")))[name]
    mk_op, mk_rhs, st = cases.SOLVES_PER[name]
    so, b = mk_op(), mk_rhs()
    ml = oracle.ml_create(so, **st)
    assert ml.nlevels() == gold["nlevels"]
    x = np.zeros_like(b)
    h = ml.solve(b, x, maxiter=10, tol=1e-8)
    ml.close()
    want = [float(gold["res0_l2"])] + [float(v) for v in gold["rel_l2"]]
    assert len(h) == len(want)
    np.testing.assert_allclose(h, want, rtol=1e-10, atol=1e-12 if "line" in name else 1e-14)
    inner = x[1:-1, 1:-1]
    assert abs(float(np.sqrt(np.cumsum((inner * inner).ravel())[-1])) - float(gold["x_l2"])) <= 1e-11 * float(gold["x_l2"])
