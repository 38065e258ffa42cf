"""Multi-GPU path on CPU: images are sharded over ranks with no data-path collective
(SURVEY.md 8e).  world_size-2 gloo run: every image is decoded by exactly one rank, the shards
concatenate to the whole batch, and the harness' max-over-ranks timing reduction works."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

from compeg_amd.sharding import shard, shard_bounds


def test_shard_bounds_cover_exactly_once():
    for n in (0, 1, 7, 8, 9, 256, 2048, 2049):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert shard_bounds(2048, 3, 8) == (768, 1024)   # BASELINE config 4: 256 frames per GPU
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_host_core_shares_of_the_ranks():
    """Every rank gets a disjoint share of its GPU's NUMA node (or of the host, without NUMA information)."""
    from compeg_amd.sharding import cpu_share, parse_cpu_list
    assert parse_cpu_list("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    host = list(range(256))
    node0, node1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    shares = [cpu_share(host, node0 if r < 4 else node1, 4, r % 4) for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    assert sorted(c for s in shares for c in s) == host                      # disjoint, everything used
    assert all(set(shares[r]) <= set(node0 if r < 4 else node1) for r in range(8))
    # no NUMA information: an even cut of what the process may use
    flat = [cpu_share(list(range(8)), None, 2, r) for r in range(2)]
    assert flat == [[0, 1, 2, 3], [4, 5, 6, 7]]
    # a cpuset smaller than the node, more ranks than cores: never empty
    assert cpu_share([3, 5], node0, 4, 3) in ([3], [5])
    assert cpu_share([9], None, 8, 7) == [9]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from compeg_amd.sharding import max_over_ranks, shard
    from oracle import oracle as orc     # the checker stands in for the GPU decoder on CPU
    from tools import synth

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = list(range(7))                                # image i is synthesised from seed i
    mine = shard(batch, rank, world)
    digests = []
    for i in mine:
        jpeg = synth.make_jpeg(64, 32, seed=i, ri=2)
        digests.append((i, hashlib.sha256(orc.ImageData(jpeg).decode().tobytes()).hexdigest()))
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)             # harness-side only, not on the data path
    slowest = max_over_ranks(float(rank + 1))
    dist.barrier()
    if rank == 0:
        flat = [d for part in gathered for d in part]
        np.save(os.path.join(out_dir, "result.npy"), np.array([len(flat), slowest]))
        with open(os.path.join(out_dir, "digests.txt"), "w") as f:
            f.write("\n".join(f"{i} {h}" for i, h in flat))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    import torch.multiprocessing as mp

    from oracle import oracle as orc
    from tools import synth

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n, slowest = np.load(tmp_path / "result.npy")
    assert int(n) == 7 and slowest == 2.0                 # max over ranks
    lines = (tmp_path / "digests.txt").read_text().split("\n")
    assert [int(l.split()[0]) for l in lines] == list(range(7))   # contiguous blocks, in order
    for l in lines:
        i, h = l.split()
        want = orc.ImageData(synth.make_jpeg(64, 32, seed=int(i), ri=2)).decode()
        assert hashlib.sha256(want.tobytes()).hexdigest() == h
