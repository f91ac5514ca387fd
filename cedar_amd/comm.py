"""Rank-to-rank transport of the domain-decomposed solvers, without torch in the rank process.

`NativeComm`  RCCL over xGMI through the library's own C ABI (include/cedar_amd.h section 3,
              cedar_amd/csrc/comm.cpp): grouped ncclSend/ncclRecv for the halo, ncclAllReduce for the norms,
              ncclAllGather for the coarse levels, all enqueued on the library's current HIP stream.  What it
              replaces in the reference: the MSG halo library and the MPI collectives of the MPI flavour
              (src/3d/mpi/msg_exchanger.cc:188-197, src/2d/ftn/mpi/mpi_msg.F:425-550,
              include/cedar/3d/mpi/grid_func.h:41, include/cedar/3d/mpi/redist_solver.h:221-224).
`SocketComm`  rehearsal transport for boxes where the ranks share ONE GPU (RCCL refuses two ranks per device):
              same interface, messages staged through host memory and TCP sockets on 127.0.0.1.  Test
              infrastructure for the orchestration, never a measured path.

Bootstrap: ranks are ordinary processes (started by bench.py, torch.distributed.run or anything else that sets
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Rank 0 creates the 128-byte RCCL unique id and serves it on a TCP
port next to MASTER_PORT; the others fetch it (`bootstrap_unique_id`).

Buffers are `capi.DeviceArray`s addressed as (array, offset in doubles, count in doubles).
"""
import ctypes as C
import os
import socket
import struct
import time

import numpy as np

from . import capi

lib = capi.lib
ID_BYTES = 128
_MAGIC = b"CEDARAMDUID1"

lib.cedar_amd_comm_available.restype = C.c_int
lib.cedar_amd_comm_why_unavailable.restype = C.c_char_p
lib.cedar_amd_comm_unique_id.argtypes = [C.c_void_p]
lib.cedar_amd_comm_create.restype = C.c_void_p
lib.cedar_amd_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.cedar_amd_comm_destroy.argtypes = [C.c_void_p]
lib.cedar_amd_comm_exchange.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
lib.cedar_amd_comm_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.cedar_amd_comm_allreduce_max.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.cedar_amd_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
lib.cedar_amd_comm_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
lib.cedar_amd_stream_create.restype = C.c_void_p
lib.cedar_amd_stream_destroy.argtypes = [C.c_void_p]
lib.cedar_amd_stream_wait.argtypes = [C.c_void_p, C.c_void_p]
lib.cedar_amd_event_record.restype = C.c_void_p
lib.cedar_amd_event_elapsed_ms.restype = C.c_float
lib.cedar_amd_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
lib.cedar_amd_event_destroy.argtypes = [C.c_void_p]


class EventTimer:
    """milliseconds between two HIP events on the library's current stream: t = EventTimer(); ...; ms = t.stop()"""

    def __init__(self):
        self.e0 = lib.cedar_amd_event_record()

    def stop(self):
        e1 = lib.cedar_amd_event_record()
        ms = lib.cedar_amd_event_elapsed_ms(self.e0, e1)
        lib.cedar_amd_event_destroy(self.e0)
        lib.cedar_amd_event_destroy(e1)
        return float(ms)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _ports():
    base = int(os.environ.get("MASTER_PORT", "29500"))
    # torch.distributed.run keeps its store on MASTER_PORT itself; look next to it
    return [base + 1 + i for i in range(8)]


def _recv_exact(s, n):
    out = b""
    while len(out) < n:
        chunk = s.recv(n - len(out))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        out += chunk
    return out


def _job_tag(world):
    """16 bytes that identify THIS job on a shared host: base MASTER_PORT, world size and a run id when the launcher
    exports one (torch.distributed.run: TORCHELASTIC_RUN_ID; anything: CEDAR_AMD_RUN_ID).  Two jobs with neighbouring
    MASTER_PORTs scan overlapping port ranges; the tag keeps their handshakes apart."""
    import hashlib
    rid = os.environ.get("CEDAR_AMD_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    key = "%s:%d:%s" % (os.environ.get("MASTER_PORT", "29500"), world, rid)
    return hashlib.sha1(key.encode()).digest()[:16]


def bootstrap_bytes(payload, rank, world, timeout=120.0, tag=b""):
    """rank 0 hands `payload` (bytes) to every other rank over TCP on MASTER_ADDR; returns the payload everywhere.
    Handshake: client -> magic + job tag + its rank; rank 0 answers magic + job tag + payload only to ranks 1..world-1 of
    the same job that it has not served yet, and the client checks the echoed tag before it accepts the payload."""
    if world == 1:
        return payload
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    magic = _MAGIC + tag + _job_tag(world)
    if rank == 0:
        srv, err = None, None
        for port in _ports():
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((host, port))
                break
            except OSError as e:
                err, srv = e, None
        if srv is None:
            raise RuntimeError(f"bootstrap: no free port next to MASTER_PORT: {err}")
        srv.listen(world)
        srv.settimeout(timeout)
        served = set()
        while len(served) < world - 1:
            conn, _ = srv.accept()
            with conn:
                conn.settimeout(timeout)
                try:
                    if _recv_exact(conn, len(magic)) != magic:
                        continue  # another job (or a foreign client): not counted, not answered
                    (r,) = struct.unpack("<I", _recv_exact(conn, 4))
                except (ConnectionError, socket.timeout):
                    continue
                if not 1 <= r < world or r in served:
                    continue
                conn.sendall(magic + struct.pack("<I", len(payload)) + payload)
                served.add(r)
        srv.close()
        return payload
    deadline = time.time() + timeout
    while time.time() < deadline:
        for port in _ports():
            try:
                with socket.create_connection((host, port), timeout=2.0) as s:
                    s.settimeout(3.0)  # a foreign service on a candidate port does not answer: move on
                    s.sendall(magic + struct.pack("<I", rank))
                    if _recv_exact(s, len(magic)) != magic:
                        continue  # not this job's rank 0
                    (n,) = struct.unpack("<I", _recv_exact(s, 4))
                    return _recv_exact(s, n)
            except (OSError, ConnectionError):
                continue
        time.sleep(0.1)
    raise TimeoutError("bootstrap: rank 0 did not serve the payload")


class Stream:
    """a non-blocking HIP stream of the library (the side stream of the overlapped halo exchange)"""

    def __init__(self):
        self.h = lib.cedar_amd_stream_create()

    def close(self):
        if self.h:
            lib.cedar_amd_stream_destroy(self.h)
            self.h = None


def _ptr(arr, off):
    return arr.ptr + 8 * int(off)


class NativeComm:
    """RCCL communicator owned by libcedar_amd.so"""
    name = "RCCL"

    def __init__(self, rank=None, world=None):
        r, w = env_rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        if not lib.cedar_amd_comm_available():
            raise RuntimeError("librccl.so.1 could not be loaded: " + lib.cedar_amd_comm_why_unavailable().decode())
        uid = None
        if self.rank == 0:
            buf = C.create_string_buffer(ID_BYTES)
            if lib.cedar_amd_comm_unique_id(buf) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
            uid = buf.raw
        uid = bootstrap_bytes(uid, self.rank, self.world)
        self.h = lib.cedar_amd_comm_create(uid, self.rank, self.world)
        if not self.h:
            raise RuntimeError("ncclCommInitRank failed")
        self._scal = capi.DeviceArray((8,))

    def p2p(self, sends, recvs):
        """sends / recvs: lists of (peer, DeviceArray, offset, count); one ncclGroup on the library's current stream"""
        ns, nr = len(sends), len(recvs)
        if ns + nr == 0:
            return
        IS, IR = C.c_int * max(ns, 1), C.c_int * max(nr, 1)
        PS, PR = C.c_void_p * max(ns, 1), C.c_void_p * max(nr, 1)
        ZS, ZR = C.c_size_t * max(ns, 1), C.c_size_t * max(nr, 1)
        rc = lib.cedar_amd_comm_exchange(
            self.h, ns, IS(*[s[0] for s in sends]), PS(*[_ptr(s[1], s[2]) for s in sends]), ZS(*[s[3] for s in sends]),
            nr, IR(*[r[0] for r in recvs]), PR(*[_ptr(r[1], r[2]) for r in recvs]), ZR(*[r[3] for r in recvs]))
        if rc:
            raise RuntimeError("cedar_amd_comm_exchange failed")

    def allgather(self, send, count, recv):
        if lib.cedar_amd_comm_allgather(self.h, send.ptr, recv.ptr, int(count)):
            raise RuntimeError("cedar_amd_comm_allgather failed")

    def allreduce_sum(self, value):
        self._scal.upload(np.array([value] + [0.0] * 7))
        if lib.cedar_amd_comm_allreduce_sum(self.h, self._scal.ptr, 1):
            raise RuntimeError("cedar_amd_comm_allreduce_sum failed")
        return float(self._scal.numpy()[0])

    def allreduce_max(self, value):
        self._scal.upload(np.array([value] + [0.0] * 7))
        if lib.cedar_amd_comm_allreduce_max(self.h, self._scal.ptr, 1):
            raise RuntimeError("cedar_amd_comm_allreduce_max failed")
        return float(self._scal.numpy()[0])

    def barrier(self):
        self.allreduce_sum(0.0)

    def close(self):
        if self.h:
            lib.cedar_amd_comm_destroy(self.h)
            self.h = None


class SocketComm:
    """Host-staged rehearsal transport (ranks sharing one GPU): a full mesh of TCP connections on 127.0.0.1.
    Blocking and unoptimised on purpose -- it exists to run the orchestration, not to be measured."""
    name = "host-staged sockets (rehearsal)"

    def __init__(self, rank=None, world=None):
        r, w = env_rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.peers = {}
        if self.world == 1:
            return
        host = "127.0.0.1"
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.bind((host, 0))
        srv.listen(self.world)
        my_port = srv.getsockname()[1]
        # every rank learns every port through rank 0 (gather + broadcast over the bootstrap channel)
        table = self._exchange_ports(my_port)
        # lower rank connects to higher rank; the greeting carries the job tag: this listener sits on an ephemeral port that
        # may lie in the candidate range of some rank-0 bootstrap (its clients then knock here and are turned away)
        hello = _MAGIC + _job_tag(self.world)
        for p in range(self.rank + 1, self.world):
            s = socket.create_connection((host, table[p]), timeout=60.0)
            s.sendall(hello + struct.pack("<I", self.rank))
            self.peers[p] = s
        srv.settimeout(120.0)
        while len(self.peers) < self.world - 1:
            conn, _a = srv.accept()
            conn.settimeout(10.0)
            try:
                if _recv_exact(conn, len(hello)) != hello:
                    conn.close()
                    continue
                (p,) = struct.unpack("<I", _recv_exact(conn, 4))
            except (ConnectionError, socket.timeout, OSError):
                conn.close()
                continue
            if not 0 <= p < self.rank or p in self.peers:
                conn.close()
                continue
            self.peers[p] = conn
        srv.close()
        for s in self.peers.values():
            s.settimeout(300.0)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def _exchange_ports(self, my_port):
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        if self.rank == 0:
            srv = None
            for port in _ports():
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((host, port))
                    break
                except OSError:
                    srv = None
            if srv is None:
                raise RuntimeError("SocketComm: no free bootstrap port")
            srv.listen(self.world)
            srv.settimeout(120.0)
            table, conns = {0: my_port}, []
            magic = _MAGIC + _job_tag(self.world)
            while len(table) < self.world:
                conn, _a = srv.accept()
                conn.settimeout(60.0)
                try:
                    if _recv_exact(conn, len(magic)) != magic:
                        conn.close()
                        continue
                    r, p = struct.unpack("<II", _recv_exact(conn, 8))
                except (ConnectionError, socket.timeout):
                    conn.close()
                    continue
                if not 1 <= r < self.world or r in table:  # a rank of another job, or the same rank twice
                    conn.close()
                    continue
                try:
                    conn.sendall(magic)  # at once: the client learns within seconds that it reached ITS rank 0
                except OSError:
                    conn.close()
                    continue
                table[r] = p
                conns.append(conn)
            blob = struct.pack("<%dI" % self.world, *[table[r] for r in range(self.world)])
            for conn in conns:
                conn.sendall(blob)
                conn.close()
            srv.close()
            return [table[r] for r in range(self.world)]
        deadline = time.time() + 120.0
        while time.time() < deadline:
            for port in _ports():
                try:
                    with socket.create_connection((host, port), timeout=2.0) as s:
                        s.settimeout(3.0)  # some other listener on a candidate port (another job, a peer socket): move on
                        magic = _MAGIC + _job_tag(self.world)
                        s.sendall(magic + struct.pack("<II", self.rank, my_port))
                        if _recv_exact(s, len(magic)) != magic:
                            continue  # another job's rank 0
                        s.settimeout(120.0)  # the table comes once every rank has registered
                        blob = _recv_exact(s, 4 * self.world)
                        return list(struct.unpack("<%dI" % self.world, blob))
                except (OSError, ConnectionError):
                    continue
            time.sleep(0.1)
        raise TimeoutError("SocketComm: bootstrap failed")

    @staticmethod
    def _get(arr, off, count):
        h = np.empty(int(count))
        if count:
            capi.sync()
            lib.cedar_amd_memcpy_d2h(h.ctypes.data, _ptr(arr, off), int(count) * 8)
        return h

    @staticmethod
    def _put(arr, off, h):
        if h.size:
            lib.cedar_amd_memcpy_h2d(_ptr(arr, off), h.ctypes.data, h.size * 8)

    def p2p(self, sends, recvs):
        # deadlock-free on blocking sockets: post every send from a helper thread, receive in the caller
        import threading
        out = [(p, self._get(a, o, c).tobytes()) for p, a, o, c in sends]

        def push():
            for p, data in out:
                if p == self.rank:
                    continue
                self.peers[p].sendall(data)
        t = threading.Thread(target=push)
        t.start()
        selfq = [data for p, data in out if p == self.rank]
        for p, a, o, c in recvs:
            data = selfq.pop(0) if p == self.rank else _recv_exact(self.peers[p], int(c) * 8)
            self._put(a, o, np.frombuffer(data, dtype=np.float64).copy())
        t.join()

    def allgather(self, send, count, recv):
        sends = [(p, send, 0, count) for p in range(self.world)]
        recvs = [(p, recv, p * count, count) for p in range(self.world)]
        self.p2p(sends, recvs)

    def _allreduce(self, value, op):
        import threading
        data = struct.pack("<d", float(value))
        others = [p for p in range(self.world) if p != self.rank]

        def push():
            for p in others:
                self.peers[p].sendall(data)
        t = threading.Thread(target=push)
        t.start()
        vals = {self.rank: float(value)}
        for p in others:
            vals[p] = struct.unpack("<d", _recv_exact(self.peers[p], 8))[0]
        t.join()
        acc = vals[0]
        for p in range(1, self.world):  # rank order: every rank forms the same sum bit for bit
            acc = op(acc, vals[p])
        return acc

    def allreduce_sum(self, value):
        return self._allreduce(value, lambda a, b: a + b)

    def allreduce_max(self, value):
        return self._allreduce(value, max)

    def barrier(self):
        self.allreduce_sum(0.0)

    def close(self):
        for s in self.peers.values():
            try:
                s.close()
            except OSError:
                pass
        self.peers = {}
