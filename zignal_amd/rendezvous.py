"""Host-side meeting point of the ranks of one node: barrier, all-gather of small byte strings, and what is
built on them (max / min over ranks, broadcast of rank 0's bytes). Plain localhost TCP and the standard library: the
multi-GPU path has no PyTorch dependency (BASELINE.json north_star), and nothing here touches a GPU.

No counterpart in the reference (single process, zignal/measure/mls.py:338-343). It carries only control traffic: the
timing barrier of bench.py, the max over ranks of a step time, the 128 bootstrap bytes of the RCCL communicator
(zmls_comm_unique_id). Data never goes through it: channels need no exchange, and periods spread over the GPUs are
reduce-scattered by RCCL inside libzmls (csrc/comm.hip).

How the ranks find each other: every rank of a launch shares its parent process (torch.distributed.run's agent, or
bench.py --gpus N itself) and $MASTER_PORT; rank 0 listens on an ephemeral localhost port and publishes it in a file
named after both, the others read the file and connect. Every collective is one round trip to rank 0: a few tens of
microseconds for eight ranks on one host.
"""
import os
import socket
import struct
import tempfile
import time

_MAGIC = b"zmlsrdzv1"


def default_key():
    """Same for every rank of one launch, different between launches: parent pid + $MASTER_PORT (+ the elastic run id)."""
    return "%d_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))


def _send(sock, payload):
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += part
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


class Rendezvous:
    """rank 0 serves, the others hold one connection to it. All ranks must call the same collectives in the same order.
    A rank that does not arrive within `timeout_s` makes the others raise TimeoutError instead of hanging."""

    def __init__(self, rank, world, key=None, timeout_s=600.0, directory=None):
        self.rank, self.world, self.timeout_s = int(rank), int(world), float(timeout_s)
        assert 0 <= self.rank < self.world
        self._peers, self._sock, self._listener, self._file = [], None, None, None
        if self.world == 1:
            return
        key = default_key() if key is None else str(key)
        path = os.path.join(directory or tempfile.gettempdir(), "zmls_rdzv_%d_%s" % (os.getuid(), key))
        deadline = time.monotonic() + self.timeout_s
        if self.rank == 0:
            self._listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self._listener.bind(("127.0.0.1", 0))
            self._listener.listen(self.world)
            self._listener.settimeout(self.timeout_s)
            tmp = "%s.tmp.%d" % (path, os.getpid())
            with open(tmp, "w") as f:
                f.write("%d\n" % self._listener.getsockname()[1])
            os.replace(tmp, path)                                  # published whole or not at all
            self._file = path
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = self._listener.accept()
                conn.settimeout(self.timeout_s)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                hello = _recv(conn)
                if not hello.startswith(_MAGIC + key.encode() + b":"):
                    conn.close()                                   # somebody else's launch, or a port scanner
                    continue
                peers[int(hello.rsplit(b":", 1)[1])] = conn
            self._peers = [peers[r] for r in range(1, self.world)]
            for c in self._peers:
                _send(c, b"ok")
        else:
            while True:
                try:
                    with open(path) as f:
                        port = int(f.read().strip())
                    s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                    s.settimeout(self.timeout_s)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send(s, _MAGIC + key.encode() + b":%d" % self.rank)
                    if _recv(s) == b"ok":
                        self._sock = s
                        break
                    s.close()
                except (FileNotFoundError, ValueError, ConnectionError, OSError):
                    pass                                           # not published yet, or a stale file of an earlier launch
                if time.monotonic() > deadline:
                    raise TimeoutError("rank %d: no rendezvous at %s after %.0f s" % (self.rank, path, self.timeout_s))
                time.sleep(0.01)

    # ---- the one primitive
    def all_gather(self, payload=b""):
        """every rank's bytes, in rank order, on every rank"""
        if self.world == 1:
            return [bytes(payload)]
        try:
            if self.rank == 0:
                parts = [bytes(payload)] + [_recv(c) for c in self._peers]
                blob = b"".join(struct.pack("<I", len(p)) + p for p in parts)
                for c in self._peers:
                    _send(c, blob)
                return parts
            _send(self._sock, bytes(payload))
            blob = _recv(self._sock)
        except socket.timeout:
            raise TimeoutError("rank %d: a rank did not reach the rendezvous within %.0f s" % (self.rank, self.timeout_s)) from None
        parts, off = [], 0
        while off < len(blob):
            (n,) = struct.unpack_from("<I", blob, off)
            parts.append(blob[off + 4:off + 4 + n])
            off += 4 + n
        return parts

    # ---- what bench.py and the RCCL bootstrap need
    def barrier(self):
        self.all_gather(b"")

    def gather_floats(self, value):
        return [struct.unpack("<d", p)[0] for p in self.all_gather(struct.pack("<d", float(value)))]

    def max(self, value):
        return max(self.gather_floats(value))

    def min(self, value):
        return min(self.gather_floats(value))

    def broadcast(self, make_payload):
        """rank 0's bytes (made by calling make_payload there) on every rank"""
        return self.all_gather(make_payload() if self.rank == 0 else b"")[0]

    def close(self):
        for c in self._peers:
            try:
                c.close()
            except OSError:
                pass
        self._peers = []
        for s in (self._sock, self._listener):
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._sock = self._listener = None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
