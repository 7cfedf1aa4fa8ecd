"""Multi-GPU path = partition of independent images, no collective on the data path.
Covered on CPU with two gloo ranks (the driver runs the real 8-GPU bench itself)."""
import os
import socket
import sys

import pytest

from popsift_amd.shard import shard_indices, shard_sizes


def test_shards_partition_the_batch():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            shards = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for s in shards for i in s)
            assert flat == list(range(n))
            assert [len(s) for s in shards] == shard_sizes(n, world)
            assert max(shard_sizes(n, world)) - min(shard_sizes(n, world)) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import numpy as np
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import oracle as O
    from popsift_amd.shard import reduce_stats, shard_indices
    from popsift_amd.synth import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank extracts its shard of a 5-image batch (oracle stands in for the GPU here: this test
    # checks the sharding / reduction logic, not the extraction)
    seeds = [31, 32, 33, 34, 35]
    mine = shard_indices(len(seeds), rank, world)
    feats = descs = 0
    for i in mine:
        a, b = O.Oracle().run(synth(seeds[i], 64, 48)).counts()
        feats += a
        descs += b
    dist.barrier()
    t, c = reduce_stats(dist, 1.0 + rank, [feats, descs, len(mine)])
    q.put((rank, mine, feats, descs, t, c))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_batch(oracle_mod):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, f0, d0, t0, c0), (r1, m1, f1, d1, t1, c1) = res
    assert m0 == [0, 2, 4] and m1 == [1, 3]
    assert t0 == t1 == 2.0                      # MAX over ranks
    assert c0 == c1 == [f0 + f1, d0 + d1, 5.0]  # SUM over ranks
    # the sharded totals equal a single-process run over the whole batch
    from popsift_amd.synth import synth
    O = oracle_mod
    tot = [0, 0]
    for s in (31, 32, 33, 34, 35):
        a, b = O.Oracle().run(synth(s, 64, 48)).counts()
        tot[0] += a
        tot[1] += b
    assert c0[:2] == [float(tot[0]), float(tot[1])]
