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


class _FailingShard(_OracleShard):
    def search(self, queries, k, id_offset=0):
        raise RuntimeError("shard search exploded")


def _failing_merge(all_s, all_i):
    raise RuntimeError("merge exploded")


def _failure_worker(rank, world, port, mode, broadcast, out_dir):
    """mode "local": rank 1's local search raises; mode "merge": the merge on rank 0 raises.  Every rank must come back from
    search() — nobody may be left waiting in a collective — and every rank must learn of the failure."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        rng = np.random.default_rng(5)
        x = rng.standard_normal((300, 16)).astype(np.float16)
        q = rng.standard_normal((4, 16)).astype(np.float16)
        lo, hi = shard_range(300, rank, world)
        shard = _FailingShard(x[lo:hi]) if mode == "local" and rank == 1 else _OracleShard(x[lo:hi])
        searcher = ShardedSearcher(shard, lo, merge=_failing_merge if mode == "merge" else _oracle_merge)
        try:
            searcher.search(q, 5, broadcast=broadcast)
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
            f.write(msg)
        # the group is still usable: a second, healthy search goes through
        ok = ShardedSearcher(_OracleShard(x[lo:hi]), lo, merge=_oracle_merge).search(q, 5, broadcast=True)
        assert ok is not None and ok[1].shape == (4, 5)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,broadcast", [("local", True), ("local", False), ("merge", True), ("merge", False)])
def test_a_failure_on_one_rank_reaches_every_rank_and_hangs_nobody(tmp_path, mode, broadcast):
    world = 3
    mp.spawn(_failure_worker, args=(world, _free_port(), mode, broadcast, str(tmp_path)), nprocs=world, join=True)
    msgs = [open(tmp_path / f"r{r}.txt").read() for r in range(world)]
    if mode == "local":
        assert "exploded" in msgs[1]                                         # the failing rank raises its own error
        assert all("local search failed on rank(s) [1]" in msgs[r] for r in (0, 2)), msgs   # the gathered status words name it
    else:
        assert "merge of the 3 shards' lists failed on rank 0" in msgs[0] and "exploded" in msgs[0]
        if broadcast:                                                         # the broadcast status word carries it
            assert all("merge of the shards' lists failed on rank 0" in msgs[r] for r in (1, 2)), msgs
        else:                                                                 # rank0-only results: the others got None, as ever
            assert msgs[1] == msgs[2] == "no error"


def _world1_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(17)
        x = rng.standard_normal((500, 24)).astype(np.float16)
        q = rng.standard_normal((9, 24)).astype(np.float16)
        s = ShardedSearcher(_OracleShard(x), 0, merge=_oracle_merge, exercise_collective=True)
        a = s.search(q, 7, broadcast=True)
        b = ShardedSearcher(_OracleShard(x), 0, merge=_oracle_merge).search(q, 7)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert s._status is not None and s._status.tolist() == [0]
        np.savez(out_path, ok=1)
    finally:
        dist.destroy_process_group()


def test_world_size_one_can_exercise_the_collective_path(tmp_path):
    """exercise_collective=True: gather + merge + broadcast in a one-rank group (what tests/test_gpu_nccl.py does under RCCL)."""
    out = str(tmp_path / "ok.npz")
    mp.spawn(_world1_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert os.path.exists(out)
