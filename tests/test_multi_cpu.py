"""CPU tier for the multi-GPU paths: the partitioning logic of zignal_amd.multi, and a
world_size-2 rehearsal over gloo of both ways to split a batch. No GPU here, so the two device steps
(period sums, transform) are played by numpy and the oracle; what is under test is everything between
them: who holds which periods, the reduce-scatter send layout, which channel block a rank ends up with,
the divisor, and the bootstrap of the communicator id bytes."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from zignal_amd import multi  # noqa: E402


def test_channel_shard_is_a_partition():
    for C in (0, 1, 7, 64, 513, 4096):
        for P in (1, 2, 3, 8):
            seen = []
            for r in range(P):
                first, n = multi.channel_shard(C, P, r)
                seen.extend(range(first, first + n))
                assert abs(n - C / P) < 1
            assert seen == list(range(C))


def test_period_shard_covers_every_period_once_and_skips_only_period_zero():
    for R in (2, 3, 5, 17, 64):
        for P in (1, 2, 4, 8):
            held, skips = [], 0
            for r in range(P):
                mine, skip, total = multi.period_shard(R, P, r)
                assert total == R - 1
                assert skip == (1 if (mine and mine[0] == 0) else 0)
                skips += skip
                held.extend(mine)
            assert sorted(held) == list(range(R)) and skips == 1
    with pytest.raises(AssertionError):
        multi.period_shard(1, 2, 0)


def test_send_layout_blocks_are_what_each_rank_receives():
    rng = np.random.default_rng(1)
    s = rng.standard_normal((31, 12))
    for P in (1, 2, 3, 4):
        blocks = multi.send_layout(s, P)
        assert blocks.shape == (P, 31, 12 // P) and blocks.flags.c_contiguous
        for q in range(P):
            first, n = multi.exchange_block(12, P, q)
            assert np.array_equal(blocks[q], s[:, first:first + n])
    with pytest.raises(ValueError):
        multi.exchange_block(10, 4, 0)


def test_file_exchange_hands_rank0_bytes_to_the_others(tmp_path):
    path = str(tmp_path / "id.bin")
    blob = bytes(range(128))
    assert multi.file_exchange(path, 0, lambda: blob) == blob
    assert multi.file_exchange(path, 1, None, timeout_s=1.0) == blob
    with pytest.raises(TimeoutError):
        multi.file_exchange(str(tmp_path / "never.bin"), 1, None, timeout_s=0.05)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_over_gloo_channel_sharded_and_period_sharded():
    """Two processes, gloo, CPU tensors: see tests/multi_rank_worker.py for what each rank does."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi_rank_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "rank %d ok" % rank in out
