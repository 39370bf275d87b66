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


def test_rendezvous_collectives_among_threads():
    """zignal_amd.rendezvous: what bench.py --gpus N uses instead of torch.distributed for its barrier, its max over
    ranks and the RCCL id bytes. Ranks as threads here (it is plain sockets); world sizes 1, 2, 5 and 8."""
    import threading
    import time
    from zignal_amd.rendezvous import Rendezvous
    for world in (1, 2, 5, 8):
        out, errs = {}, []
        key = "test_%d_%d" % (os.getpid(), world)

        def rank_main(r):
            try:
                with Rendezvous(r, world, key=key, timeout_s=20.0) as z:
                    z.barrier()
                    got = z.all_gather(bytes([r]) * (r + 1))
                    assert got == [bytes([q]) * (q + 1) for q in range(world)]
                    assert z.max(10.0 + r) == 10.0 + world - 1 and z.min(10.0 + r) == 10.0
                    assert z.gather_floats(r * 0.5) == [q * 0.5 for q in range(world)]
                    blob = z.broadcast(lambda: bytes(range(128)))
                    assert blob == bytes(range(128))
                    for _ in range(50):                      # many rounds in a row stay in step
                        z.barrier()
                    out[r] = z.all_gather(b"x" * 100000)     # larger than one TCP segment
            except Exception as e:  # noqa: BLE001
                errs.append((r, repr(e)))

        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        t0 = time.monotonic()
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        assert not errs, errs
        assert len(out) == world and all(len(v) == world and v[0] == b"x" * 100000 for v in out.values())
        assert time.monotonic() - t0 < 30


def test_rendezvous_times_out_instead_of_hanging(tmp_path):
    from zignal_amd.rendezvous import Rendezvous
    with pytest.raises(TimeoutError):                        # nobody publishes a port
        Rendezvous(1, 2, key="nobody_%d" % os.getpid(), timeout_s=0.3, directory=str(tmp_path))
    import threading
    res = {}

    def lonely():
        try:
            with Rendezvous(0, 2, key="late_%d" % os.getpid(), timeout_s=1.0, directory=str(tmp_path)) as z:
                z.barrier()
        except Exception as e:  # noqa: BLE001
            res["e"] = e
    t = threading.Thread(target=lonely)
    t.start()
    t.join(20)
    assert isinstance(res.get("e"), (TimeoutError, OSError))  # rank 1 never came: rank 0 gives up at accept()


def test_bench_launches_its_own_ranks_and_never_imports_torch():
    """bench.py --gpus 2 without a GPU: two children meet at the rendezvous, each finds no device and says so, the
    launcher's status is theirs. (With a GPU the same launch is tests/test_bench_contract.py's.) Nothing in bench.py or
    the package imports torch: the N > 1 run must not mix PyTorch's bundled ROCm into the process."""
    import re
    for path in [os.path.join(ROOT, "bench.py")] + [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, "zignal_amd")) for f in fs if f.endswith(".py")]:
        src = open(path).read()
        assert not re.search(r"^\s*(import|from)\s+torch", src, re.M), path
    from zignal_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rendezvous-timeout", "30"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == "" and r.stderr.count("no GPU visible") == 2, r.stderr[-2000:]
    code = "import sys; sys.argv=['bench.py']; import bench; print('torch' in sys.modules)"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.stdout.strip() == "False", r.stdout + r.stderr


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
