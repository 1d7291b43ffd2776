"""N > 1 path on CPU: two processes, `gloo` backend.  The sharding + gather plumbing of
fmsi_amd/sharding.py (what bench.py and a multi-GPU driver use with backend `nccl`) must reproduce the
single-process byte stream; the per-rank compute here is the CPU oracle (checker standing in for the GPU)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "oracle"))

from fmsi_amd.sharding import gather_to_root, shard_range, shard_sizes, sharded_query  # noqa: E402


def test_shard_range_properties():
    for n in (0, 1, 2, 7, 64, 1000, 12345):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))      # contiguous, ordered
            sizes = shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1                     # balanced


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, prefix, kmers_text, k, n, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py
    orc = oracle_py.Oracle.load(prefix)
    kmers = np.frombuffer(kmers_text, dtype=f"S{k}")

    def compute(part):
        fwd, rc = orc.ternaries_bulk(part.tobytes(), len(part), k, True)
        return (np.where(fwd != -1, fwd, rc) == 1).astype(np.uint8)   # -O schedule, forward strand first

    res = sharded_query(kmers, compute)
    # ragged gather: an item count the world size does not divide
    t = torch.arange(*shard_range(n - 1, rank, world), dtype=torch.int64)
    g = gather_to_root(t, n - 1)
    if rank == 0:
        assert torch.equal(g, torch.arange(n - 1))
        np.save(out_path, res)
    else:
        assert res is None and g is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 5])
def test_ranks_gloo_equal_single_process(tmp_path, world):
    """World sizes 2, 3 and 5 over 4001 queries: shards of unequal length, rank order = output order."""
    import oracle_py
    from conftest import GOLDEN, index_prefix, mixed_kmers
    oracle_py.build_oracle()
    k, n = 31, 4001
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">"))
    kmers = mixed_kmers(ms, k, n, seed=9)
    text = "".join(kmers).encode()
    prefix = index_prefix("sars_k31")
    orc = oracle_py.Oracle.load(prefix)
    fwd, rc = orc.ternaries_bulk(text, n, k, True)
    want = (np.where(fwd != -1, fwd, rc) == 1).astype(np.uint8)
    out = tmp_path / "res.npy"
    mp.spawn(_worker, args=(world, _free_port(), prefix, text, k, n, str(out)), nprocs=world, join=True)
    got = np.load(out)
    assert np.array_equal(got, want)
    assert 0.4 < got.mean() < 0.6
