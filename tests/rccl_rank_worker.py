"""One rank of tests/test_gpu_multi.py::test_two_ranks_on_two_gpus_over_rccl (not collected by pytest): rank r on GPU r.

Every rank builds the same seeded capture, keeps the periods `period_shard` gives it, and calls the fused entry point
collectively (partial period sums, ncclReduceScatter in double-buffered chunks, transform of its channel block); its
block must equal what zmls_impulse_device makes of the whole capture on the same GPU, and the oracle.
usage: rccl_rank_worker.py <rank> <world> <path of the id file>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import c_oracle, mls_oracle  # noqa: E402
from zignal_amd import _lib, multi  # noqa: E402
from zignal_amd.device import DeviceBuffer, Plan, sync  # noqa: E402
from zignal_amd.measure.mls import TAPS  # noqa: E402


def main():
    rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dev = rank
    for N, C, R, chunk, dtype in ((12, 48, 5, 8, "f64"), (16, 64, 4, 32, "f64"), (13, 64, 9, 16, "f32"), (10, 8, 2, 4, "f64")):
        taps = TAPS[N][0]
        p = Plan(N, taps, dtype=dtype, device=dev)
        L, npdt = p.L, p.np_dtype
        x = np.random.default_rng(N * 1000 + C).standard_normal((R * L, C)).astype(npdt)
        blob = multi.file_exchange("%s.%d" % (path, N), rank, multi.make_unique_id)
        comm = multi.Comm(blob, rank, world, dev)
        assert comm.info()["world"] == world
        mine, skip, total = multi.period_shard(R, world, rank)
        local = np.ascontiguousarray(np.concatenate([x[q * L:(q + 1) * L] for q in mine], axis=0)) if mine else np.zeros((1, C), npdt)
        d_x = DeviceBuffer(local.nbytes, dev).upload(local)
        first, share = multi.exchange_block(C, world, rank)
        d_h = DeviceBuffer(L * share * x.itemsize, dev)
        d_h.fill(0xFF)
        ps = multi.PeriodSharded(p, comm, chunk)
        for _ in range(2):                                  # twice on the same staging buffers
            ps.impulse_device(d_x.ptr, p.dtype_code, len(mine) * L, skip, total, C, C, 1, d_h.ptr, share, 1)
        sync(dev)
        got = d_h.download((L, share), npdt).astype(np.float64)
        # the whole capture on this GPU alone
        d_all, d_ref = DeviceBuffer(x.nbytes, dev).upload(x), DeviceBuffer(L * C * x.itemsize, dev)
        p.reserve(C)
        p.impulse_device(d_all.ptr, p.dtype_code, R * L, C, C, 1, d_ref.ptr, C, 1)
        sync(dev)
        want = d_ref.download((L, C), npdt).astype(np.float64)[:, first:first + share]
        peak = np.max(np.abs(want))
        # same samples, added up in another order (partial sums per rank): equal to rounding
        assert np.max(np.abs(got - want)) <= (1e-14 if dtype == "f64" else 1e-6) * peak, "N=%d: differs from the one-GPU result" % N
        ref = c_oracle.sequence(N, taps)
        for c in (0, share - 1):
            h = mls_oracle.get_impulse(N, taps, np.ascontiguousarray(x[:, first + c:first + c + 1], dtype=np.float64), ref=ref)[:, 0]
            assert np.max(np.abs(got[:, c] - h)) <= (1e-12 if dtype == "f64" else 1e-5) * np.max(np.abs(h)), "N=%d: differs from the oracle" % N
        comm.close()
        p.close()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
