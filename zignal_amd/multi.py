"""Several GPUs: one process per GPU (no counterpart in the reference, which is single-process and
single-channel, zignal/measure/mls.py:338-343).

Two ways to split a batch of captures:

channel-sharded (default, no exchange)
    Channels are independent, so rank r takes channels `channel_shard(C, P, r)` and calls
    `Plan.impulse_device` on them. Nothing crosses GPUs.

period-sharded (RCCL reduce-scatter over xGMI)
    Every rank holds some of the periods of every channel, e.g. an acquisition striped over the
    devices: rank r holds periods `period_shard(R, P, r)`. trim_and_avgerage (mls.py:284-306) is a sum
    followed by one division and xcorr_circular (mls.py:169-256) is linear, so each rank adds up its own
    periods, the partial sums are reduce-scattered over the channel axis, and rank q transforms channel
    block q. `PeriodSharded.impulse_device` does that through libzmls (RCCL is loaded by the library; this
    module has no PyTorch dependency, the caller brings the 128 bootstrap bytes to every rank:
    `zignal_amd.rendezvous.Rendezvous.broadcast`, `file_exchange` below, MPI, anything).
"""
import ctypes
import os
import time

import numpy as np

from . import _lib as L


# ---------------------------------------------------------------- partitioning (pure host logic)
def channel_shard(n_ch, world, rank):
    """(first, count): contiguous channel range of `rank`; the first n_ch % world ranks get one more."""
    assert world >= 1 and 0 <= rank < world and n_ch >= 0
    base, rem = divmod(n_ch, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def period_shard(repeats, world, rank):
    """Periods of a capture of `repeats` periods striped over `world` ranks: rank r holds global periods
    r, r+world, ... Returns (periods, skip, total): the global period indices this rank holds (in the
    order they sit in its local buffer), how many leading local periods are not averaged (1 on the rank
    holding period 0, which the reference throws away, mls.py:284), and the divisor repeats-1."""
    assert repeats >= 2, "at least two sequences are needed, (repeats=2)"
    assert world >= 1 and 0 <= rank < world
    mine = list(range(rank, repeats, world))
    skip = 1 if mine and mine[0] == 0 else 0
    return mine, skip, repeats - 1


def exchange_block(n_ch, world, rank):
    """(first, count) of the channel block rank `rank` receives from the reduce-scatter; n_ch must split evenly."""
    if n_ch % world:
        raise ValueError("%d channels do not split evenly over %d ranks" % (n_ch, world))
    share = n_ch // world
    return rank * share, share


def send_layout(partial, world):
    """(L, C) partial sums -> the reduce-scatter send buffer (world, L, C/world): block q holds what rank q
    receives. Same layout as zmls_period_sum_device writes (used by host-side tests of the exchange)."""
    Lr, C = partial.shape
    share = exchange_block(C, world, 0)[1]
    return np.ascontiguousarray(partial.reshape(Lr, world, share).transpose(1, 0, 2))


# ---------------------------------------------------------------- bootstrap of the 128 id bytes
def file_exchange(path, rank, make_id, timeout_s=120.0):
    """Dependency-free bootstrap: rank 0 writes the id next to `path` and renames it into place, the
    others wait for the file. Returns the id bytes. The launcher gives every rank the same fresh path."""
    if rank == 0:
        blob = make_id()
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)
        return blob
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == L.COMM_ID_BYTES:
                return blob
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError("no communicator id at %s after %.0f s" % (path, timeout_s))
        time.sleep(0.01)


def rccl_library():
    """(path, version code) of the librccl that libzmls loaded"""
    buf, ver = ctypes.create_string_buffer(512), ctypes.c_int(0)
    L.check(L.lib().zmls_comm_library_info(buf, 512, ctypes.byref(ver)))
    return buf.value.decode(), ver.value


def make_unique_id():
    buf = (ctypes.c_uint8 * L.COMM_ID_BYTES)()
    L.check(L.lib().zmls_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """This rank's end of an RCCL communicator owned by libzmls."""

    def __init__(self, id_bytes, rank, world, device=None):
        assert len(id_bytes) == L.COMM_ID_BYTES
        self.rank, self.world = int(rank), int(world)
        self.device = L.default_device() if device is None else int(device)
        h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(id_bytes)
        L.check(L.lib().zmls_comm_create(ctypes.byref(h), buf, self.rank, self.world, self.device))
        self._h = h

    @classmethod
    def loopback(cls, hub, rank, device=0):
        """This rank's end of a loopback exchange among threads of one process on one GPU (tests: zmls_testing.h)."""
        self = cls.__new__(cls)
        self.rank, self.world, self.device = int(rank), hub.world, int(device)
        h = ctypes.c_void_p()
        L.check(L.lib().zmls_comm_create_loopback(ctypes.byref(h), hub._h, self.rank, self.device))
        self._h = h
        return self

    def close(self):
        if getattr(self, "_h", None):
            L.lib().zmls_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def info(self):
        buf = (ctypes.c_int64 * 8)()
        L.check(L.lib().zmls_comm_info(self._h, buf))
        v = list(buf)
        return {"rank": v[0], "world": v[1], "device": v[2], "chunk_channels": v[3], "staging_bytes": v[4]}

    def reserve(self, plan, chunk_channels):
        L.check(L.lib().zmls_comm_reserve(self._h, plan._h, int(chunk_channels)))

    def reduce_scatter_device(self, d_send, d_recv, recv_count, dtype_code, stream=None):
        L.check(L.lib().zmls_comm_reduce_scatter_device(self._h, d_send, d_recv, int(recv_count), dtype_code, stream))

    def all_gather_device(self, d_send, d_recv, send_count, dtype_code, stream=None):
        L.check(L.lib().zmls_comm_all_gather_device(self._h, d_send, d_recv, int(send_count), dtype_code, stream))


class LoopbackHub:
    """Meeting point of the `world` virtual ranks of a loopback exchange (tests only)."""

    def __init__(self, world):
        self.world = int(world)
        h = ctypes.c_void_p()
        L.check(L.lib().zmls_comm_loopback_hub_create(ctypes.byref(h), self.world))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            L.lib().zmls_comm_loopback_hub_destroy(self._h)
            self._h = None


class PeriodSharded:
    """Impulse responses of captures whose periods are spread over the ranks of `comm`."""

    def __init__(self, plan, comm, chunk_channels):
        assert plan.device == comm.device
        self.plan, self.comm = plan, comm
        comm.reserve(plan, chunk_channels)

    def impulse_device(self, d_x, in_dtype, n_rows, skip_periods, total_periods, n_ch, row_stride, ch_stride,
                       d_h, out_row_stride, out_ch_stride, stream=None):
        """d_x: this rank's periods (n_rows = R_local*L frames, n_ch channels); d_h: (L, n_ch/world), the
        channel block `exchange_block(n_ch, world, rank)`. Collective over the communicator."""
        L.check(L.lib().zmls_impulse_period_sharded_device(
            self.plan._h, self.comm._h, d_x, in_dtype, int(n_rows), int(skip_periods), int(total_periods), int(n_ch),
            int(row_stride), int(ch_stride), d_h, int(out_row_stride), int(out_ch_stride), stream))


# ---------------------------------------------------------------- the steps one by one (tests, custom pipelines)
def period_sum_device(plan, d_x, in_dtype, n_rows, skip_periods, row_stride, ch_stride, n_blocks, block_ch_stride,
                      ch_first, block_ch, d_out, stream=None):
    L.check(L.lib().zmls_period_sum_device(plan._h, d_x, in_dtype, int(n_rows), int(skip_periods), int(row_stride),
                                           int(ch_stride), int(n_blocks), int(block_ch_stride), int(ch_first),
                                           int(block_ch), d_out, stream))


def impulse_sums_device(plan, d_s, total_periods, n_ch, row_stride, ch_stride, d_h, out_row_stride, out_ch_stride,
                        stream=None):
    L.check(L.lib().zmls_impulse_sums_device(plan._h, d_s, int(total_periods), int(n_ch), int(row_stride), int(ch_stride),
                                             d_h, int(out_row_stride), int(out_ch_stride), stream))
