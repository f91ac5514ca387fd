"""The C++ mirror of Cedar's API surface (include/cedar: config, arrays, grid functions, gallery,
kernel_manager with the "hip" kernels, cdr2/cdr3::solver) -- drop-in boundary #2 of SURVEY 8b.
CPU: a host-only program built with g++ (config reader, gallery builders vs tests/problems.py, norms,
registry semantics).  GPU: the registered "hip" kernels against the oracle bit for bit, and the
examples (the reference's serial Poisson / periodic programs and a plain-C caller) end to end."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import problems as pb

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FLAGS = ["-std=c++17", "-O1", "-Wall", f"-I{ROOT}/include", f"-L{ROOT}/cedar_amd/lib", "-lcedar_amd", "-L/opt/rocm/lib",
         f"-Wl,-rpath,{ROOT}/cedar_amd/lib", "-Wl,-rpath,/opt/rocm/lib"]


def build(src, exe):
    from cedar_amd import capi  # noqa: F401  (builds libcedar_amd.so if missing)
    subprocess.run(["g++", os.path.join(HERE, "cxx", src)] + FLAGS + ["-o", str(exe)], check=True)


def test_host_side_of_the_cxx_mirror(tmp_path):
    json.dump({"grid": {"n": [300, 200], "periodic": [True, False]},
               "solver": {"relaxation": "line-xy", "cycle": {"nrelax-pre": 3, "nrelax-post": 2, "type": "f"},
                          "max-iter": 7, "tol": 1e-9}}, open(tmp_path / "config.json", "w"))
    exe = tmp_path / "host_checks"
    build("host_checks.cc", exe)
    p = subprocess.run([str(exe), str(tmp_path)], check=True, capture_output=True, text=True)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    assert (got["nx"], got["ny"]) == (300, 200)
    assert got["periodic"] == [1, 0] and got["per_mask"] == 1
    assert (got["relaxation"], got["pre"], got["post"], got["maxiter"], got["cycle"]) == (3, 3, 2, 7, 1)
    assert got["tol"] == 1e-9
    assert got["inf_norm"] == -3.0  # signed entry of largest magnitude, ghosts excluded
    assert abs(got["l2"] - np.sqrt(0.25 + 9 + 4)) < 1e-15
    assert (got["r_first"], got["r_hip"], got["r_after_bad"], got["via_run"]) == (6, 35, 35, 42)
    assert "no implementation named does-not-exist" in p.stderr + p.stdout
    for name, want in (("poisson2", pb.poisson2(9, 7)), ("diag2", pb.diag_diffusion2(11, 6, 1.0, 1e-4)), ("fe2", pb.fe2(8, 10)),
                       ("poisson3", pb.poisson3(6, 7, 5)), ("fe3", pb.fe3(5, 6, 7))):
        a = np.fromfile(tmp_path / f"{name}.bin", dtype=np.float64).reshape(want.shape)
        assert np.array_equal(a, want), name


def test_gpu_side_cxx_programs_compile_and_link(tmp_path):
    """the programs the GPU tests run (kernel registry test, the reference's examples against the mirror, the plain-C
    caller) build and link against the headers and the library here, without a GPU; the plane-config block of a
    configuration reaches the kernel parameters"""
    build("hip_kernels.cc", tmp_path / "hip_kernels")
    ex = os.path.join(ROOT, "examples")
    subprocess.run(["make", "-C", ex], check=True, capture_output=True)
    for exe in ("ser-poisson-2d", "ser-poisson-3d", "ser-periodic-2d", "ser-periodic-3d", "capi-poisson-2d"):
        assert os.path.exists(os.path.join(ex, exe)), exe


@pytest.mark.gpu
def test_registered_hip_kernels_and_user_kernels(tmp_path, oracle):
    """tests/cxx/hip_kernels.cc: the "hip" bindings with the reference's signatures against the oracle bit for bit; a
    kernel written against the reference's abstract point_relax and selected with set<T>("user") is run by
    solver::solve (which leaves the device-resident path for the reference's per-kernel orchestration) and reproduces
    the resident history; solver.levels hands out the hierarchy; same agreement for the 3D solver."""
    json.dump({"solver": {"max-iter": 6}}, open(tmp_path / "config.json", "w"))
    exe = tmp_path / "hip_kernels"
    build("hip_kernels.cc", exe)
    p = subprocess.run([str(exe), str(tmp_path)], check=True, capture_output=True, text=True)
    nx, ny = 37, 22
    g = (ny + 2, nx + 2)
    so = pb.fe2(nx, ny)
    x, b = np.zeros(g), np.zeros(g)
    i = np.arange(1, nx + 1)[None, :]
    j = np.arange(1, ny + 1)[:, None]
    x[1:-1, 1:-1] = 0.01 * i - 0.02 * j
    b[1:-1, 1:-1] = 1.0 / (1 + i + j)
    sor = np.zeros((2,) + g)
    oracle.setup_recip2(so, sor)
    oracle.relax2(so, b, x, sor, 0)
    oracle.relax2(so, b, x, sor, 1)
    r = np.zeros(g)
    oracle.residual2(so, b, x, r)
    assert np.array_equal(np.fromfile(tmp_path / "x.bin").reshape(g), x)
    assert np.array_equal(np.fromfile(tmp_path / "r.bin").reshape(g), r)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    assert got["resident_before"] == 1 and got["resident_after"] == 0
    # 45 x 38 five-point: 5 levels; the user kernel sets up 4 of them and relaxes 3 times per level visit
    assert got["nlevels"] == 5 and got["user_setup_calls"] == 4
    cycles = len(got["hist_user"]) - 1
    assert cycles >= 1 and got["user_run_calls"] == 3 * 4 * cycles
    assert got["same_level1_operator"] == 1
    # same kernels in the same order; only the norm differs (device tree sum vs the host's sequential sum)
    np.testing.assert_allclose(got["hist_user"], got["hist_resident"], rtol=1e-12)
    assert got["x_diff"] <= 1e-14 * got["x_max"]
    np.testing.assert_allclose(got["hist3_orchestrated"], got["hist3_resident"], rtol=1e-12)
    assert len(got["hist3_planes_resident"]) >= 2 and got["hist3_planes_resident"][-1] < 1e-6
    np.testing.assert_allclose(got["hist3_planes_orchestrated"], got["hist3_planes_resident"], rtol=1e-10)
    so3 = pb.fe3(13, 12, 11)
    b3 = np.zeros(so3.shape[1:])
    kk, jj, ii = np.meshgrid(np.arange(1, 12), np.arange(1, 13), np.arange(1, 14), indexing="ij")
    b3[1:-1, 1:-1, 1:-1] = 1e-3 * ((ii * 7 + jj * 3 + kk * 5) % 11 - 5)
    ml3 = oracle.ml_create(so3, relax="plane-xyz")
    x3 = np.zeros_like(b3)
    want3 = ml3.solve(b3, x3, maxiter=4)
    ml3.close()
    np.testing.assert_allclose(got["hist3_planes_resident"], want3, rtol=1e-8, atol=1e-13)
    # and the resident history is the oracle's
    so5 = pb.poisson2(45, 38)
    bb = np.zeros((40, 47))
    jj, ii = np.meshgrid(np.arange(1, 39), np.arange(1, 46), indexing="ij")
    bb[1:-1, 1:-1] = 1e-3 * ((ii * 7 + jj * 3) % 11 - 5)
    ml = oracle.ml_create(so5)
    xo = np.zeros_like(bb)
    want = ml.solve(bb, xo, maxiter=6)
    ml.close()
    np.testing.assert_allclose(got["hist_resident"], want, rtol=1e-10, atol=1e-14)


@pytest.mark.gpu
def test_examples_end_to_end(tmp_path):
    ex = os.path.join(ROOT, "examples")
    subprocess.run(["make", "-C", ex], check=True, capture_output=True)
    json.dump({"grid": {"n": [200, 200]}}, open(tmp_path / "config.json", "w"))
    for cfg in ("periodic-config.json", "periodic-config-3d.json"):
        with open(os.path.join(ex, cfg)) as f:
            open(tmp_path / cfg, "w").write(f.read())
    d3 = tmp_path / "d3"  # no config.json there: the 3D example's default 64^3 grid
    d3.mkdir()
    for exe, args in (("ser-poisson-2d", []), ("ser-poisson-3d", []), ("ser-periodic-2d", []), ("ser-periodic-3d", []),
                      ("capi-poisson-2d", ["200"])):
        p = subprocess.run([os.path.join(ex, exe)] + args, cwd=d3 if exe == "ser-poisson-3d" else tmp_path,
                           capture_output=True, text=True)
        assert p.returncode == 0, (exe, p.stdout[-400:], p.stderr[-400:])
        text = p.stdout + p.stderr
        if exe.startswith("ser-"):
            m = re.search(r"Solution norm: (\S+)", text)
            assert m and abs(float(m.group(1))) < 1e-2, (exe, text[-300:])
