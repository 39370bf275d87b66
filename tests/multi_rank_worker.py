"""One rank of the world_size-2 gloo rehearsal started by tests/test_multi_cpu.py (not collected by pytest).

Both ranks build the same seeded multi-channel capture, then

channel-sharded: rank r takes `channel_shard` channels, gets their impulse responses (oracle standing in
    for the GPU), the shards are all-gathered and must equal the oracle on the whole capture.

period-sharded: rank r keeps only the periods `period_shard` gives it, adds them up (numpy standing in for
    zmls_period_sum_device) into the reduce-scatter send layout, the partial sums are summed across ranks
    (gloo has no reduce_scatter on CPU tensors: all_reduce + take block `rank`), rank r divides by the
    total and correlates its block; all-gathered, that must again equal the oracle on the whole capture.

Also: rank 0's 128 id bytes reach rank 1 through a gloo broadcast."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import c_oracle, mls_oracle  # noqa: E402
from zignal_amd import multi  # noqa: E402


def torch_exchange(dist, rank, make_id):
    """the 128 id bytes of rank 0 on every rank, over gloo (a host program that already uses torch.distributed)"""
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(make_id()), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    return bytes(t.numpy().tobytes())


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, taps, R, C = 9, (9, 5), 6, 8
    L = (1 << N) - 1
    rng = np.random.default_rng(77)                      # same capture on every rank
    x = rng.standard_normal((R * L, C))
    ref = c_oracle.sequence(N, taps)
    want = np.stack([mls_oracle.get_impulse(N, taps, np.ascontiguousarray(x[:, c:c + 1]), ref=ref)[:, 0] for c in range(C)], axis=1)

    # ---- id bootstrap
    blob = torch_exchange(dist, rank, lambda: bytes((7 * i + 3) % 256 for i in range(128)))
    assert blob == bytes((7 * i + 3) % 256 for i in range(128))

    # ---- channel-sharded: no exchange on the data path
    first, n = multi.channel_shard(C, world, rank)
    mine = np.stack([mls_oracle.get_impulse(N, taps, np.ascontiguousarray(x[:, c:c + 1]), ref=ref)[:, 0]
                     for c in range(first, first + n)], axis=1)
    parts = [torch.zeros(L, multi.channel_shard(C, world, r)[1], dtype=torch.float64) for r in range(world)]
    dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(mine)))
    got = np.concatenate([p.numpy() for p in parts], axis=1)
    assert np.array_equal(got, want), "channel-sharded result differs"

    # ---- period-sharded
    periods, skip, total = multi.period_shard(R, world, rank)
    local = np.concatenate([x[p * L:(p + 1) * L] for p in periods], axis=0)     # this rank's buffer: its periods back to back
    partial = np.zeros((L, C))
    for i in range(skip, len(periods)):
        partial += local[i * L:(i + 1) * L]
    send = torch.from_numpy(multi.send_layout(partial, world))
    dist.all_reduce(send)                                                        # stands in for ncclReduceScatter(sum)
    bfirst, bn = multi.exchange_block(C, world, rank)
    block = send[rank].numpy() / total
    h_block = np.stack([mls_oracle.xcorr_circular(block[:, c:c + 1].copy(), ref)[:, 0] for c in range(bn)], axis=1)
    parts = [torch.zeros(L, bn, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(h_block)))
    got = np.concatenate([p.numpy() for p in parts], axis=1)
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err <= 1e-12, "period-sharded result off by %.3e" % err

    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
