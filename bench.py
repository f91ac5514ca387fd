#!/usr/bin/env python3
"""Headline benchmark: fine-grid DOF/s per BoxMG V-cycle on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" = one V(2,1) cycle (`cycle->run(x,b)` of include/cedar/cycle/vcycle.h:57-115)
on synthetic, HBM-resident data; set-up and the convergence-check residual are
outside the timed region.  Prints ONE JSON line.

Workloads (BASELINE.json configs; synthetic operators built on the device):
  3d27   3D 27-pt gallery::fe,        512^3 per GPU, point 8-colour GS   (default; config 4 / 5,
         the configuration the north-star roofline target is quoted on)
  2d9    2D 9-pt variable coefficient, 4096^2, point 4-colour GS          (config 2)
  2d9l   2D 9-pt anisotropic,          8192^2, zebra line relax x+y       (config 3)
  2d5    2D 5-pt Poisson,              512^2                              (config 1)
Extra keys in the JSON line:
  roofline     dominant kernel (the level-0 relax sweep): algorithmic bytes per launch
               / average launch duration measured live with HIP events on the library's stream
  cpu_baseline the oracle (C restatement of the reference, kind "port") timed on one host
               core on a bounded sample of the same workload
  level0_kernels   N = 1, 3d27: the other level-0 kernels of the cycle (residual27_rows, restrict3, interp_add3) against the
               HBM roofline: algorithmic bytes / launch time from HIP events (cedar_amd_solver_time_op)
  other_workloads  N = 1, default run only: BASELINE configs 2 and 3 (2d9 4096^2 point, 2d9l 8192^2 line-xy) timed the same
               way in this process after the 3D solver has been released: ms_per_step, DOF/s, relax launch time and frac;
               3d27_rank_of_2x2x2_loopback: one rank of config 5 (native distributed driver, loop-back transport: kernels,
               packs and copies of the 8-GPU run per rank, no links) and its ratio to the single-GPU cycle
  ms_per_step_per_allocation  N = 1: the W warm-up + K timed steps are run on every one of the fresh allocations the
               roofline launch time is taken over (default 3), and `value` / `ms_per_step` are the MEDIAN: where the
               operator lands in device memory moves a sweep by +-6 %, and the headline should
               not be the luck of the first allocation.  --allocations 1: the first allocation only.
Multi-GPU (N > 1): one process per GPU.  Started as the driver does (torch.distributed.run sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*) every process is a rank; started bare (`python bench.py --gpus N`) the parent spawns N fresh rank
processes before it touches a GPU itself.  No workload imports torch: the distributed cycles run below the C ABI
(cedar_amd_dist3_* / cedar_amd_dist2_*) and halo, norms and coarse gather are RCCL
calls issued by libcedar_amd.so (cedar_amd/comm.py).  N GPUs requested but fewer visible is an error.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (nd, default n, relax, algorithmic B/DOF of one level-0 relax sweep, launches per sweep, label)
    # 27-pt: the sweep is two plane-fused launches (relax27_plane: the planes of one k-parity, both row
    # classes) plus two small launches over the S rows between workgroup runs (one S row in eight)
    "3d27": (3, 512, "point", 136.0, 2, "3D 27-pt gallery::fe Poisson-type, {n}^3, 8-colour GS V(2,1)"),
    "2d9": (2, 4096, "point", 64.0, 2, "2D 9-pt variable-coefficient, {n}^2, 4-colour GS V(2,1)"),
    "2d9l": (2, 8192, "line-xy", 128.0, 8, "2D 9-pt anisotropic eps=1e-4, {n}^2, zebra line relax x+y V(2,1)"),
    "2d5": (2, 512, "point", 48.0, 2, "2D 5-pt Poisson, {n}^2, red-black GS V(2,1)"),
}


def build_problem(capi, wl, n):
    """operator + rhs in HBM"""
    import problems as pb
    if wl == "3d27":
        return capi.gallery("fe3", (n, n, n))
    if wl == "2d5":
        return capi.gallery("poisson2", (n, n))
    if wl == "2d9":
        so = pb.varcoef9(n, n)
    else:
        so = pb.aniso9(n, n)
    b = pb.rhs2(n, n)
    return capi.DeviceArray.from_numpy(so), capi.DeviceArray.from_numpy(b)


def _cpu_sample(wl):
    import problems as pb
    if wl == "3d27":
        return pb.fe3(96, 96, 96), pb.rhs3(96, 96, 96), "27-pt gallery::fe 96^3"
    if wl == "2d9":
        return pb.varcoef9(1024, 1024), pb.rhs2(1024, 1024), "9-pt variable-coefficient 1024^2"
    if wl == "2d9l":
        return pb.aniso9(1024, 1024), pb.rhs2(1024, 1024), "9-pt anisotropic 1024^2 line-xy"
    return pb.poisson2(512, 512), pb.rhs2(512, 512), "5-pt Poisson 512^2"


def cpu_worker(wl, relax, seconds):
    """one CPU rank: V(2,1) cycles on the bounded sample for `seconds`; prints one JSON line.
    Touches no GPU (runs in its own process, started by cpu_baseline)."""
    from pyoracle import Oracle
    so, b, sample = _cpu_sample(wl)
    dof = float(np.prod([s - 2 for s in b.shape]))
    # preferred: the reference's own Fortran kernels (oracle/_ref, built in the build container and
    # shipped with the snapshot), chained in the reference's cycle order; fallback: the C restatement
    kind, ml, closer, what = "port", None, None, "oracle/liboracle.so (gcc -O2)"
    try:
        from pyoracle import Ref
        from gen_golden import RefML
        ml = RefML(Ref(), so, relax=relax, nrelax_pre=2, nrelax_post=1)
        kind, what = "reference", "oracle/_ref/libcedar_ref.so (reference Fortran, flang -O2, MKL LAPACK)"
    except Exception:
        h = Oracle().ml_create(so, relax=relax)
        ml, closer = h, h.close
    x = np.zeros_like(b)
    ml.vcycle(x, b)  # warm-up
    t0, cycles = time.perf_counter(), 0
    while True:
        ml.vcycle(x, b)
        cycles += 1
        dt = time.perf_counter() - t0
        if dt > seconds or cycles >= 400:
            break
    if closer:
        closer()
    print(json.dumps({"dof_per_s": dof * cycles / dt, "cycles": cycles, "kind": kind, "what": what, "sample": sample}), flush=True)


def cpu_baseline(wl, relax):
    """The reference's CPU path beside the GPU number: the reference is single-threaded
    per process and parallel only through MPI ranks, so (i) one rank on one core and (ii) one rank
    per host core, each on its own block of the bounded sample (weak, halo traffic not charged:
    an upper bound for the CPU).  Workers are child processes; they never touch the GPU."""
    import subprocess

    def launch(nproc, seconds):
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", wl, relax, str(seconds)]
        ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True) for _ in range(nproc)]
        outs = []
        for p in ps:
            o, _ = p.communicate(timeout=600)
            if p.returncode != 0:
                raise RuntimeError("cpu worker failed")
            outs.append(json.loads(o.strip().splitlines()[-1]))
        return outs

    one = launch(1, 8.0)[0]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    allc = launch(cores, 8.0) if cores > 1 else [one]
    total = sum(o["dof_per_s"] for o in allc)
    return {"value": total, "unit": "DOF/s", "cores": cores, "kind": one["kind"],
            "single_core_value": one["dof_per_s"], "cpu_model": cpu_model,
            "sample": f"{one['sample']} per rank, V(2,1) cycles for 8 s, {cores} independent single-threaded ranks "
                      f"(one per host core of {cpu_model}, no halo cost charged); single rank alone: {one['dof_per_s']:.3e} DOF/s; {one['what']}"}


def other_workload(capi, wl, steps=10, warmup=3):
    """one of the 2D BASELINE configs on the resident single-GPU solver: V-cycle time and the level-0 relax launch"""
    nd, n, relax, bytes_per_dof, launches, label = WORKLOADS[wl]
    so, b = build_problem(capi, wl, n)
    capi.sync()
    t0 = time.perf_counter()
    s = capi.Solver(so, relax=relax, share_operator=True)
    capi.sync()
    setup = time.perf_counter() - t0
    x = capi.DeviceArray(b.shape)
    for _ in range(warmup):
        s.vcycle(x, b)
    capi.lib.cedar_amd_device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.vcycle(x, b)
    capi.lib.cedar_amd_device_sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    s.time_relax(x, b, 4)
    lm = s.time_relax(x, b, 40) / (40 * launches)
    dof = float(n) ** nd
    alg = bytes_per_dof * dof / launches
    out = {"workload": label.format(n=n), "ms_per_step": ms, "value": dof / (ms * 1e-3), "unit": "DOF/s", "steps": steps,
           "warmup": warmup, "levels": s.nlevels(), "setup_ms": setup * 1e3,
           "relax_launch_ms": lm, "relax_algorithmic_bytes_per_launch": alg, "relax_launches_per_sweep": launches,
           "relax_achieved_GBps": alg / (lm * 1e-3) / 1e9, "relax_frac": alg / (lm * 1e-3) / 8e12}
    s.close()
    so.free(); b.free(); x.free()
    return out


def rank_of_grid(capi, pgrid=(2, 2, 2), n=512, steps=5, warmup=2):
    """BASELINE config 5 per rank on the one GPU of this run: the native distributed driver (cedar_amd_dist3_*) as the rank in
    the middle of a px x py x pz rank grid with the loop-back transport -- every message of the grid is packed, copied in place
    of the send / receive and unpacked, so the kernels, packs and copies are those of one rank of the 8-GPU run and only the
    links are missing (the ghost values are this rank's own: a cost figure, not a solve)"""
    import ctypes as C
    from cedar_amd.dist3 import DistSolver3
    world = pgrid[0] * pgrid[1] * pgrid[2]
    centre = tuple(min(1, p - 1) for p in pgrid)
    rank = centre[2] * pgrid[0] * pgrid[1] + centre[1] * pgrid[0] + centre[0]
    g = (n + 2, n + 2, n + 2)
    A, b = capi.DeviceArray((14,) + g), capi.DeviceArray(g)
    pp = (C.c_double * 6)(0.0, 0.0, 0.0, float(n), float(n), float(n))
    capi.lib.cedar_amd_gallery(112, A.ptr, b.ptr, n, n, n, pp)
    s = DistSolver3("loopback", rank, world, A, pgrid=pgrid)
    x = capi.DeviceArray(g)
    for _ in range(warmup):
        s.vcycle(x, b)
    capi.lib.cedar_amd_device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.vcycle(x, b)
    t1 = time.perf_counter()
    capi.lib.cedar_amd_device_sync()
    t2 = time.perf_counter()
    out = {"workload": "one rank of a %dx%dx%d rank grid, 3D 27-pt %d^3 per rank, V(2,1), loop-back transport (no links)" % (pgrid + (n,)),
           "ms_per_step": (t2 - t0) / steps * 1e3, "host_enqueue_ms_per_step": (t1 - t0) / steps * 1e3, "steps": steps, "warmup": warmup,
           "levels_on_boundary_first_chain": s.chain_levels}
    s.close()
    A.free(); b.free(); x.free()
    return out


def visible_gpus():
    """number of GPUs this box shows, counted in a child process so that the caller stays free of any HIP state"""
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from cedar_amd import capi; "
                          "print(capi.device_count())" % ROOT], capture_output=True, text=True, timeout=300)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit("bench.py: could not count the GPUs: " + out.stderr[-400:])


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes (this one has not touched a GPU)"""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll all ranks: the first one that fails takes the others down (they would otherwise block forever inside RCCL
    # or the bootstrap waiting for it); an overall deadline bounds a hang of all of them
    deadline = time.time() + float(os.environ.get("CEDAR_AMD_BENCH_DEADLINE", "3000"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            sys.stderr.write("bench.py: ranks still running at the deadline, stopping them\n")
            rc = 124
            break
        time.sleep(0.2)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    sys.exit(rc)


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2], sys.argv[3], float(sys.argv[4]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="3d27", choices=list(WORKLOADS))
    ap.add_argument("--size", type=int, default=0, help="override the per-GPU grid extent")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default 3d27 run: skip the 2d9 / 2d9l timings appended as other_workloads")
    ap.add_argument("--allocations", type=int, default=3,
                    help="N=1: fresh operator allocations the roofline launch time is the median of")
    ap.add_argument("--strong", action="store_true",
                    help="3D, N>1: keep the GLOBAL grid at --size^3 and split it over the ranks (strong scaling); "
                         "default is --size^3 per GPU (weak scaling, what the driver's scaling table uses)")
    args = ap.parse_args()

    rehearsal = os.environ.get("CEDAR_AMD_DIST_BACKEND", "rccl") != "rccl"  # ranks share a GPU over a host-staged transport
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            ngpu = visible_gpus()
            if ngpu < args.gpus and not rehearsal:
                raise SystemExit("bench.py: %d GPUs requested, %d visible" % (args.gpus, ngpu))
            spawn_ranks(args.gpus)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    nd, n_default, relax, bytes_per_dof, launches, label = WORKLOADS[args.workload]
    relax_kernel = "relax27_plane" if nd == 3 else relax
    psum = nd == 3 and world == 1 and os.environ.get("CEDAR_AMD_PSUM", "1") != "0"
    if psum:
        # the resident solver's sweep with inter-plane partial sums (relax3d_psum.hip): its two halves differ (the first
        # k-parity also forms the partial sums, the second reads two partial-sum rows instead of 18 slot-rows), so the
        # roofline unit is the WHOLE sweep: relax27_planeA + relax27_rows_between + relax27_rows_sel + relax27_planeB +
        # relax27_rows_sel, 136 B x n^3 algorithmic bytes
        launches, relax_kernel = 1, "one sweep = relax27_planeA + relax27_planeB + 3 row launches (relax27_rows_between, relax27_rows_sel x2)"
    if world > 4 and nd == 3:  # rank grids with a y split exchange halos after every row class: four launches per sweep
        launches, relax_kernel = 4, "relax27_rows"
    # (2 and 4 GPUs run z slabs: a whole k-parity -- the plane-fused kernel -- between two exchanges)
    n = args.size or n_default

    from cedar_amd import capi
    ndev = capi.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible; cedar_amd has no CPU fallback")
    if world > 1 and ndev < world and not rehearsal:
        raise SystemExit("bench.py: %d GPUs requested, %d visible" % (world, ndev))

    comm = None
    if world > 1:
        local_rank = local_rank % ndev  # rehearsal on a one-GPU box: the ranks share the card
    capi.set_device(local_rank)
    if world > 1:
        from cedar_amd.comm import NativeComm, SocketComm
        comm = SocketComm(rank, world) if rehearsal else NativeComm(rank, world)

    dof = float(n) ** nd
    dsolver, t_setup = None, None
    if world > 1:
        # domain decomposition: one 512^3 block per GPU of a (px,py,pz)*512 global grid (z slabs up to 4 GPUs,
        # 2x2x2 on 8 = BASELINE config 5), halo over RCCL
        from cedar_amd.dist import Topology
        if nd == 3:
            from cedar_amd.dist import DistSolver3, GpuBackend
            be = GpuBackend(comm, local_rank)
            topo = Topology(rank, world)
            ln = [n, n, n]  # local extents
            if args.strong:
                if any(n % topo.p[d] or (n // topo.p[d]) % 16 for d in range(3)):
                    raise SystemExit("bench.py --strong: %d does not split into even local extents over %s" % (n, topo.p))
                ln = [n // topo.p[d] for d in range(3)]
            g = (ln[2] + 2, ln[1] + 2, ln[0] + 2)
            A = be.zeros((14,) + g)
            bt = be.zeros(g)
            import ctypes as C
            place = [float(topo.coord[d] * ln[d]) for d in range(3)] + [float(ln[d] * topo.p[d]) for d in range(3)]
            pp = (C.c_double * 6)(*place)
            capi.lib.cedar_amd_gallery(112, A.ptr, bt.ptr, ln[0], ln[1], ln[2], pp)  # fe3 placed in the global grid
            dof = float(ln[0]) * ln[1] * ln[2]
            # the distributed cycle runs below the C ABI (cedar_amd_dist3_*, cedar_amd/csrc/dist3.cpp): one call per
            # V-cycle from here, no per-kernel Python loop; CEDAR_AMD_DIST_DRIVER=python keeps the Python statement of
            # the same orchestration (cedar_amd/dist.py) for A/B runs
            if os.environ.get("CEDAR_AMD_DIST_DRIVER", "native") == "python":
                dsolver = DistSolver3(be, topo, A)
            else:
                from cedar_amd.dist3 import DistSolver3 as NativeDist3
                dsolver = NativeDist3(comm, rank, world, A, pgrid=topo.p)
            xt = be.zeros(g)
        else:
            # 2D workloads: n^2 per GPU of a (px n) x (py n) global grid on the native driver
            # (cedar_amd_dist2_*, cedar_amd/csrc/dist2.cpp): no torch in the rank process.  The synthetic operators are host
            # generators of the whole grid: every rank builds it and keeps its block.
            import problems as pb
            from cedar_amd.dist3 import DistSolver2 as NativeDist2, rank_grid2
            px, py = rank_grid2(world)
            if 5.0 * (px * n + 2) * (py * n + 2) * 8 > 12e9:
                raise SystemExit("bench.py: the host generator of this 2D workload is too large for %d GPUs at %d^2 per GPU; "
                                 "use --size" % (world, n))
            topo = Topology(rank, world, (px, py, 1))
            gso = {"2d9": pb.varcoef9, "2d9l": pb.aniso9, "2d5": pb.poisson2}[args.workload](px * n, py * n)
            gb = pb.rhs2(px * n, py * n)
            ci, cj = topo.coord[:2]
            sl = (slice(cj * n, cj * n + n + 2), slice(ci * n, ci * n + n + 2))
            m = pb.interior_mask((n + 2, n + 2)).astype(np.float64)
            A = capi.DeviceArray.from_numpy(np.ascontiguousarray(gso[(slice(None),) + sl]) * m)
            bt = capi.DeviceArray.from_numpy(np.ascontiguousarray(gb[sl]) * m)
            del gso, gb
            dsolver = NativeDist2(comm, rank, world, A, pgrid=(px, py), relax=relax)
            xt = capi.DeviceArray(bt.shape)
        so = b = x = None

        class _S:  # minimal adapter so that the timing code below is shared
            def vcycle(self, x_, b_):
                dsolver.vcycle(xt, bt)

            def nlevels(self):
                return dsolver.nlev_global

            def time_relax(self, x_, b_, k):
                if hasattr(dsolver, "time_relax"):
                    return dsolver.time_relax(xt, bt, k)
                from cedar_amd.comm import EventTimer
                t = EventTimer()  # HIP events on the library's stream
                for i in range(k):
                    dsolver._smooth(dsolver.levels[0], xt, bt, i & 1, 1)
                return t.stop()

            def close(self):
                pass
        solver = _S()
    else:
        so, b = build_problem(capi, args.workload, n)
        capi.sync()
        t_setup = time.perf_counter()
        solver = capi.Solver(so, relax=relax, share_operator=True)
        capi.sync()
        t_setup = time.perf_counter() - t_setup
        x = capi.DeviceArray(b.shape)

    def barrier():
        capi.lib.cedar_amd_device_sync()
        if comm is not None:
            comm.barrier()
        capi.lib.cedar_amd_device_sync()

    for _ in range(args.warmup):
        solver.vcycle(x, b)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.vcycle(x, b)
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)

    # dominant kernel: the level-0 relax sweep, HIP events on the library's stream
    nsw = 20 if nd == 3 else 40
    solver.time_relax(x, b, 4)
    ms = solver.time_relax(x, b, nsw)
    per_alloc = [ms / (nsw * launches)]
    step_ms = [elapsed / args.steps * 1e3]
    setups = [] if t_setup is None else [t_setup]
    if world == 1 and args.allocations > 1:
        # the sweep time depends on where the operator allocation lands in HBM (7-14 % between allocations of one
        # process, profiles/r01_allocation_placement_variance.log): quote the MEDIAN over fresh allocations of the
        # operator + hierarchy, not the luck of the first one
        for _ in range(args.allocations - 1):
            so2, b2 = build_problem(capi, args.workload, n)
            capi.sync()
            ts = time.perf_counter()
            s2 = capi.Solver(so2, relax=relax, share_operator=True)
            capi.sync()
            setups.append(time.perf_counter() - ts)
            x2 = capi.DeviceArray(b2.shape)
            # the same W warm-up + K timed steps on this allocation: the headline is the median over the allocations too
            for _ in range(args.warmup):
                s2.vcycle(x2, b2)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                s2.vcycle(x2, b2)
            barrier()
            step_ms.append((time.perf_counter() - t0) / args.steps * 1e3)
            s2.time_relax(x2, b2, 4)
            per_alloc.append(s2.time_relax(x2, b2, nsw) / (nsw * launches))
            s2.close()
            so2.free(); b2.free(); x2.free()
    if world == 1 and setups and args.allocations > 1:
        # a creation that maps device memory the process has not held before also pays the driver's clearing of it (hundreds
        # of ms on a box other processes have used, profiles/r02_setup_time_allocations.log): ALWAYS two more create / release
        # rounds on the resident operator, which reuse released blocks and show the set-up itself
        for _ in range(2):
            capi.sync()
            ts = time.perf_counter()
            s3 = capi.Solver(so, relax=relax, share_operator=True)
            capi.sync()
            setups.append(time.perf_counter() - ts)
            s3.close()
    level0 = None
    if world == 1 and nd == 3 and hasattr(solver, "time_op"):
        # the other level-0 kernels of the cycle against the same roofline (algorithmic bytes: SURVEY 8d / DESIGN section 5)
        level0 = {}
        for op, kname, bpd in (("residual", "residual27_rows", 136.0), ("restrict", "restrict3_kernel", 35.0),
                               ("interp_add", "interp_add3_kernel", 67.0)):
            solver.time_op(x, b, op, 2)
            t = solver.time_op(x, b, op, 10) / 10
            level0[kname] = {"launch_ms": t, "algorithmic_bytes_per_launch": bpd * dof,
                             "achieved": bpd * dof / (t * 1e-3) / 1e9, "unit": "GB/s", "frac": bpd * dof / (t * 1e-3) / 8e12}
    launch_ms = sorted(per_alloc)[len(per_alloc) // 2]
    ms_per_step = sorted(step_ms)[len(step_ms) // 2]
    alg_bytes_launch = bytes_per_dof * dof / launches
    achieved = alg_bytes_launch / (launch_ms * 1e-3) / 1e9
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if n == n_default:  # the PMC passes were taken at the benchmark size
            traffic = pmc.get(args.workload + ("_psum_sweep" if psum else "_rows" if world > 4 else ""), {}).get("hbm_bytes_per_launch")
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "relax sweep level 0 (%s)" % relax_kernel,
                "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes_launch,
                "launch_ms_per_allocation": per_alloc,
                "launch_ms_is": "median over %d fresh allocations of operator + hierarchy" % len(per_alloc)}

    if rank == 0:
        out = {
            "metric": "fine-grid DOF/s per V-cycle",
            "value": dof * world / (ms_per_step * 1e-3),
            "unit": "DOF/s",
            "n_gpus": world,  # == --gpus (checked above)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_per_allocation": step_ms,
            "ms_per_step_is": ("median over %d fresh allocations of operator + hierarchy, each W warm-up + K timed steps"
                               % len(step_ms)) if len(step_ms) > 1 else "the K timed steps",
            "higher_is_better": True,
            "scaling": "strong" if (args.strong and world > 1 and nd == 3) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label.format(n=n) + (", %d GPUs" % world if world > 1 else ""),
                       "grid_per_gpu": ([int(v) for v in ln] if (world > 1 and nd == 3) else [n] * nd), "levels": solver.nlevels(), "cycle": "V(2,1)",
                       "relaxation": relax,
                       "parallelism": "single GPU" if world == 1 else
                       "domain decomposition %s ranks, %s per GPU, halo exchange over %s" %
                       ("x".join(map(str, topo.p[:nd])),
                        "x".join(str(int(v)) for v in ln) if nd == 3 else "%d^2" % n,
                        comm.name)},
            "roofline": roofline,
            # device-side interp + Galerkin + relax set-up + solve copies, host-timed around solver creation.  Quoted: the
            # MINIMUM over the solver creations of this process.  A creation that maps device memory the process has not
            # held before also pays the driver's clearing of that memory (~0.45 s for the 45 GB of a 512^3 solver on a
            # box other processes have used: profiles/r02_setup_time_allocations.log) and, for the very first one,
            # code-object loading; a creation that reuses released blocks shows the set-up itself.  All values are listed.
            "setup_ms": None if not setups else min(setups) * 1e3,
            "setup_ms_is": "minimum over %d solver creations in this process (setup_ms_first = the first creation, what a "
                           "one-shot user pays incl. code-object loading and the driver's clearing of fresh device memory; "
                           "setup_ms_median; all in setup_ms_per_allocation)" % len(setups),
            "setup_ms_first": None if not setups else setups[0] * 1e3,
            "setup_ms_median": None if not setups else sorted(setups)[len(setups) // 2] * 1e3,
            "setup_ms_per_allocation": [v * 1e3 for v in setups],
        }
        if level0:
            out["level0_kernels"] = level0
        if world == 1 and args.workload == "3d27" and not args.size and not args.no_other_workloads:
            # BASELINE configs 2 and 3 in the same line (the driver runs the default command only)
            solver.close()
            solver = None
            so.free(); b.free(); x.free()
            out["other_workloads"] = {wl: other_workload(capi, wl) for wl in ("2d9", "2d9l")}
            # ... and what one rank of BASELINE config 5 (2x2x2 ranks, 512^3 each) costs beside its links
            try:
                rk = rank_of_grid(capi)
                rk["vs_single_gpu_ms_per_step"] = rk["ms_per_step"] / out["ms_per_step"]
            except Exception as e:  # noqa: BLE001 -- an aid beside the headline: its failure must not cost the line
                rk = {"error": "%s: %s" % (type(e).__name__, e)}
            out["other_workloads"]["3d27_rank_of_2x2x2_loopback"] = rk
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, relax)
        print(json.dumps(out), flush=True)
    if solver is not None:
        solver.close()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
