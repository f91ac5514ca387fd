"""The launcher side of the RCCL transport without a GPU: rank 0 hands the 128-byte unique id to the other ranks over a
TCP socket next to MASTER_PORT (cedar_amd/comm.py bootstrap_bytes) -- three processes, a port below it already taken
(as torch.distributed.run's store takes MASTER_PORT itself), late and early joiners."""
import multiprocessing as mp
import os
import socket
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, delay, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    time.sleep(delay)
    from cedar_amd.comm import bootstrap_bytes
    payload = bytes(range(128)) if rank == 0 else None
    q.put((rank, bootstrap_bytes(payload, rank, world, timeout=60.0)))


def test_unique_id_reaches_every_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.listen(1)  # MASTER_PORT itself stays occupied, like under torch.distributed.run
    blocker = socket.socket()
    try:
        blocker.bind(("127.0.0.1", port + 1))  # and so does the first candidate next to it
        blocker.listen(1)
    except OSError:
        blocker = None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_worker, args=(r, world, port, d, q)) for r, d in ((0, 0.5), (1, 0.0), (2, 1.0))]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(30)
    s.close()
    if blocker:
        blocker.close()
    assert all(got[r] == bytes(range(128)) for r in range(world))


def _job_worker(rank, world, port, payload_byte, delay, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    time.sleep(delay)
    from cedar_amd.comm import bootstrap_bytes
    payload = bytes([payload_byte]) * 128 if rank == 0 else None
    q.put((port, rank, bootstrap_bytes(payload, rank, world, timeout=60.0)))


def test_two_jobs_on_neighbouring_ports_do_not_cross():
    """two jobs whose MASTER_PORTs differ by one scan overlapping candidate ports; job B's clients are up before their
    own rank 0 and reach job A's server first: the handshake carries a job tag (MASTER_PORT, world size, run id), so A
    neither answers nor counts them and every rank ends with its own job's payload"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    base = s.getsockname()[1]
    s.close()
    # job A's first candidate (base + 1) is taken, so its rank 0 serves on base + 2 = job B's FIRST candidate
    blocker = socket.socket()
    try:
        blocker.bind(("127.0.0.1", base + 1))
        blocker.listen(1)
    except OSError:
        blocker = None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    jobs = [(base, 0xA1, {0: 0.0, 1: 0.3, 2: 0.6}), (base + 1, 0xB2, {0: 1.5, 1: 0.0, 2: 0.1})]
    ps = [ctx.Process(target=_job_worker, args=(r, world, port, byte, delays[r], q))
          for port, byte, delays in jobs for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(2 * world)]
    for p in ps:
        p.join(30)
    if blocker:
        blocker.close()
    want = {base: bytes([0xA1]) * 128, base + 1: bytes([0xB2]) * 128}
    assert len(got) == 2 * world
    for port, rank, payload in got:
        assert payload == want[port], (port, rank)


def _c_client(rank, world, port, q):
    """a rank that fetches the id through the library's compiled bootstrap (cedar_amd_comm_bootstrap_id)"""
    import ctypes
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from cedar_amd import capi
    buf = ctypes.create_string_buffer(128)
    rc = capi.lib.cedar_amd_comm_bootstrap_id(buf, rank, world)
    q.put((rank, buf.raw if rc == 0 else None))


def test_compiled_bootstrap_speaks_the_python_protocol():
    """cedar_amd_comm_bootstrap_id (comm.cpp: the launcher's part for hosts without Python, used by Cedar's C interface
    with more than one rank) against a Python rank 0: same magic, job tag (its own SHA-1) and framing"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_worker, args=(0, world, port, 0.3, q)),
          ctx.Process(target=_c_client, args=(1, world, port, q)),
          ctx.Process(target=_worker, args=(2, world, port, 0.0, q))]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(30)
    assert all(got[r] == bytes(range(128)) for r in range(world)), {r: (v[:4] if v else v) for r, v in got.items()}
