"""CPU: the N>1 path (row sharding + one all-gather + canonical merge) with world_size 2 and 3 over
gloo.  The per-shard searcher and the merge are the CPU oracle here (injected as the checker's
stand-ins — the product defaults are the HIP kernels); what is under test is the host logic:
shard ranges, global id offsets, the packed collective and the merge order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bergen_amd.sharded import ShardedSearcher, shard_range


class _OracleShard:
    def __init__(self, rows):
        self.rows = rows

    def search(self, queries, k, id_offset=0):
        from oracle import c_oracle
        s, i = c_oracle.canonical_search(np.asarray(queries), self.rows, k, id_offset=id_offset)
        return torch.from_numpy(s), torch.from_numpy(i)


def _oracle_merge(all_s, all_i):
    from oracle import c_oracle
    s, i = c_oracle.merge_topk(all_s.numpy(), all_i.numpy())
    return torch.from_numpy(s), torch.from_numpy(i)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, nq, k, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)  # every rank draws the same corpus, keeps only its shard
        x = rng.standard_normal((n, d)).astype(np.float16)
        x[n // 2] = x[3]  # a tie that straddles shards
        q = rng.standard_normal((nq, d)).astype(np.float16)
        lo, hi = shard_range(n, rank, world)
        searcher = ShardedSearcher(_OracleShard(x[lo:hi]), lo, merge=_oracle_merge)
        res = searcher.search(q, k)
        if rank == 0:
            np.savez(out_path, s=res[0].numpy(), i=res[1].numpy())
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_equals_single_shard(tmp_path, world):
    from oracle import c_oracle, compare
    n, d, nq, k = 1001, 32, 7, 12
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, d, nq, k, out), nprocs=world, join=True)
    got = np.load(out)
    rng = np.random.default_rng(99)
    x = rng.standard_normal((n, d)).astype(np.float16)
    x[n // 2] = x[3]
    q = rng.standard_normal((nq, d)).astype(np.float16)
    want_s, want_i = c_oracle.canonical_search(q, x, k)
    compare.assert_bit_exact(got["s"], got["i"], want_s, want_i, f"world={world}")
