"""CPU: row-sharded search BEHIND the stage object — `Retrieve(search_rank=, search_world=).retrieve(...)` on world_size 2 and 3
over gloo, against the single-process result of the same stage and against the oracle.  The index classes and the cross-shard
merge are oracle-backed stand-ins injected through Retrieve's class attributes (test infrastructure; the product defaults are the
HIP kernels).  Under test: the shard ranges cut through chunk files, the size check over the WHOLE folder, global row ids,
the packed collective, the merge, the broadcast of the merged lists, the id mapping and the return contract of
reference modules/retrieve.py:52-108 on every rank."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bergen_amd
from bergen_amd.sharded import shard_range


class OracleFlat:
    """FlatIndex's interface over the oracle's canonical search."""
    MAX_K = 248

    def __init__(self, n_rows, dim, metric="ip", device=0):
        self.rows = np.zeros((n_rows, dim), np.float16)
        self.metric = metric
        self.filled = np.zeros(n_rows, bool)

    def upload(self, rows, row0=None):
        x = torch.as_tensor(rows).cpu().numpy().astype(np.float16)
        self.rows[row0:row0 + x.shape[0]] = x
        self.filled[row0:row0 + x.shape[0]] = True

    def finalize(self):
        assert self.filled.all(), "a shard must be filled completely"
        if self.metric == "cos":
            from oracle import c_oracle
            self.rows = c_oracle.l2_normalize_rows(self.rows)
        return self

    def search(self, queries, k, id_offset=0):
        from oracle import c_oracle
        q = torch.as_tensor(queries).cpu().numpy().astype(np.float16)
        if self.metric == "cos":
            q = c_oracle.l2_normalize_rows(q)
        return c_oracle.canonical_search(q, self.rows, k, id_offset=id_offset)

    def close(self):
        pass


class OracleSparse:
    """SparseIndex's interface over the sparse oracle."""
    MAX_K = 120

    def __init__(self, n_rows, vocab, device=0):
        self.n_rows, self.vocab, self.blocks = n_rows, vocab, {}

    def upload(self, rows, row0=None):
        self.blocks[int(row0)] = rows.to_dense().numpy().astype(np.float16)

    def finalize(self):
        parts = [self.blocks[r] for r in sorted(self.blocks)]
        dense = np.concatenate(parts) if parts else np.zeros((0, self.vocab), np.float16)
        assert dense.shape[0] == self.n_rows
        from bergen_amd.sparse import _csr_from_any
        self.csr = _csr_from_any(dense, self.vocab)
        return self

    def search(self, queries, k, id_offset=0):
        from oracle import c_oracle
        q = torch.as_tensor(queries).cpu().numpy().astype(np.float16)
        return c_oracle.sparse_canonical_search(*self.csr, self.vocab, q, k, id_offset=id_offset)

    def close(self):
        pass


def _oracle_merge(all_s, all_i):
    from oracle import c_oracle
    s, i = c_oracle.merge_topk(np.ascontiguousarray(all_s.numpy()), np.ascontiguousarray(all_i.numpy()))
    return torch.from_numpy(s), torch.from_numpy(i)


class _Stage(bergen_amd.Retrieve):
    _dense_index_cls = OracleFlat
    _sparse_index_cls = OracleSparse
    _shard_merge = staticmethod(_oracle_merge)


class _Plug:
    model = torch.nn.Identity()

    def __init__(self, kind):
        self.model_name = "fake/splade-table" if kind == "sparse" else "fake/table-dense"
        self.similarity = bergen_amd.CosineSim() if kind == "cos" else bergen_amd.DotProduct()
        self.sparse = kind == "sparse"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dataset(n, nq):
    import datasets
    return {"doc": datasets.Dataset.from_dict({"id": [f"doc{i}" for i in range(n)]}),
            "query": datasets.Dataset.from_dict({"id": [f"q{i}" for i in range(nq)]})}


def _corpus(kind, n, nq, d):
    rng = np.random.default_rng(5)
    if kind == "sparse":
        x = (rng.random((n, d)) < 0.08) * rng.random((n, d))
        q = (rng.random((nq, d)) < 0.1) * rng.random((nq, d))
        x[n // 2] = x[3]
        return torch.from_numpy(x).half(), torch.from_numpy(q).half()
    x = rng.standard_normal((n, d)).astype(np.float16)
    x[n // 2] = x[3]  # a tie that straddles shards
    return torch.from_numpy(x), torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float16))


def _write_folders(root, kind, x, q, cuts):
    q_path, d_path = os.path.join(root, "q"), os.path.join(root, "d")
    os.makedirs(q_path)
    os.makedirs(d_path)
    edges = [0] + cuts + [x.shape[0]]
    for j in range(len(edges) - 1):  # chunk files that do NOT line up with the shard ranges
        block = x[edges[j]:edges[j + 1]]
        torch.save(block.to_sparse() if kind == "sparse" else block.clone(), os.path.join(d_path, f"embedding_chunk_{10 * (j + 1)}.pt"))
    torch.save(q.to_sparse() if kind == "sparse" else q, os.path.join(q_path, "embedding_chunk_0.pt"))
    return q_path, d_path


def _worker(rank, world, port, root, kind, n, nq, k, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stage = _Stage(init_args=_Plug(kind), batch_size=64, batch_size_sim=7, num_workers=0, search_world="auto",
                       search_results=results)
        assert (stage.search_rank, stage.search_world) == (rank, world)
        out = stage.retrieve(_dataset(n, nq), os.path.join(root, "q"), os.path.join(root, "d"), k)
        lo, hi = shard_range(n, rank, world)
        assert stage._resident[os.path.join(root, "d")][1][4] == (lo, hi)
        if results == "rank0" and rank != 0:
            assert out is None
        else:
            with open(os.path.join(root, f"out{rank}.pkl"), "wb") as f:
                pickle.dump(out, f)
        out2 = stage.retrieve(_dataset(n, nq), os.path.join(root, "q"), os.path.join(root, "d"), k)  # shard stays resident
        assert (out2 is None) == (out is None) and (out is None or torch.equal(out2["score"], out["score"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,results", [("ip", 2, "all"), ("ip", 3, "rank0"), ("cos", 2, "rank0"), ("sparse", 2, "all"),
                                                ("sparse", 3, "rank0")])
def test_sharded_stage_equals_single_process_stage(tmp_path, kind, world, results):
    from oracle import c_oracle, compare
    n, nq, d, k = (600, 17, 97, 9) if kind == "sparse" else (1001, 23, 32, 12)
    x, q = _corpus(kind, n, nq, d)
    root = str(tmp_path)
    q_path, d_path = _write_folders(root, kind, x, q, cuts=[n // 5, n // 5 + 1, (2 * n) // 3])
    single = _Stage(init_args=_Plug(kind), batch_size=64, batch_size_sim=7, num_workers=0).retrieve(_dataset(n, nq), q_path, d_path, k)
    mp.spawn(_worker, args=(world, _free_port(), root, kind, n, nq, k, results), nprocs=world, join=True)
    ranks = range(world) if results == "all" else [0]
    for r in ranks:
        with open(os.path.join(root, f"out{r}.pkl"), "rb") as f:
            got = pickle.load(f)
        # the reference's return contract (retrieve.py:104-108)
        assert isinstance(got["score"], torch.Tensor) and got["score"].dtype == torch.float32 and tuple(got["score"].shape) == (nq, k)
        assert got["q_id"] == [f"q{i}" for i in range(nq)] and isinstance(got["doc_id"][0][0], str)
        assert torch.equal(got["score"], single["score"]) and got["doc_id"] == single["doc_id"], f"rank {r} world {world}"
    # and the single-process stage is the oracle's answer
    if kind == "sparse":
        from bergen_amd.sparse import _csr_from_any
        ws, wi = c_oracle.sparse_canonical_search(*_csr_from_any(x.numpy(), d), d, q.numpy(), k)
    else:
        xq, xd = q.numpy(), x.numpy()
        if kind == "cos":
            xq, xd = c_oracle.l2_normalize_rows(xq), c_oracle.l2_normalize_rows(xd)
        ws, wi = c_oracle.canonical_search(xq, xd, k)
    got_i = np.array([[int(s[3:]) for s in row] for row in single["doc_id"]])
    compare.assert_bit_exact(single["score"].numpy(), got_i, ws, wi, f"stage {kind}")


def test_sharded_stage_size_check_covers_the_whole_folder(tmp_path):
    """A rank whose own rows are all there still raises the reference's IOError when the FOLDER is short (retrieve.py:165-166)."""
    n, nq, d, k = 300, 5, 16, 4
    x, q = _corpus("ip", n, nq, d)
    q_path, d_path = _write_folders(str(tmp_path), "ip", x[:250], q, cuts=[100])
    stage = _Stage(init_args=_Plug("ip"), batch_size=64, num_workers=0, search_rank=0, search_world=2)
    with pytest.raises(IOError, match="Missing 50 documents"):
        stage._resident_index(d_path, n, "ip", rows=shard_range(n, 0, 2))


def test_search_world_arguments():
    with pytest.raises(ValueError):
        bergen_amd.Retrieve(init_args=_Plug("ip"), search_rank=2, search_world=2)
    with pytest.raises(ValueError):
        bergen_amd.Retrieve(init_args=_Plug("ip"), search_results="everyone")
    r = bergen_amd.Retrieve(init_args=_Plug("ip"), search_world="auto")  # no process group: the single-GPU path
    assert (r.search_rank, r.search_world) == (0, 1)
