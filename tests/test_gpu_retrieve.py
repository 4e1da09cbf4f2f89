"""GPU: the Retrieve stage end to end (chunk files on disk -> resident index -> fused search ->
doc-id strings -> .trec), against the oracle.  Uses the reference's file formats and return types."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle, compare

pytestmark = pytest.mark.gpu


class _FakeEncoder(torch.nn.Module):
    def forward(self, v=None):
        return (v,)


class _TableDense:
    """A Dense-shaped plug-in whose 'encoder' looks embeddings up in a table keyed by the text (an int)."""
    model_name = "fake/table-dense"

    def __init__(self, q_table, d_table, similarity):
        self.model = _FakeEncoder()
        self.q_table, self.d_table = q_table, d_table
        self.similarity = similarity

    def collate_fn(self, batch, query_or_doc=None):
        key = 'generated_query' if query_or_doc == "query" else "content"
        return {"idx": torch.tensor([int(s[key]) for s in batch])}

    def __call__(self, query_or_doc, batch):
        t = self.q_table if query_or_doc == "query" else self.d_table
        return {"embedding": t[batch["idx"].cpu()]}


@pytest.mark.parametrize("metric", ["ip", "cos"])
def test_retrieve_end_to_end(tmp_path, metric):
    import datasets
    import bergen_amd
    from bergen_amd import utils
    g = torch.Generator().manual_seed(21)
    n, nq, d, k = 1000, 23, 128, 10
    d_table = torch.randn(n, d, generator=g).half()
    q_table = torch.randn(nq, d, generator=g).half()
    sim = bergen_amd.DotProduct() if metric == "ip" else bergen_amd.CosineSim()
    r = bergen_amd.Retrieve(init_args=_TableDense(q_table, d_table, sim), batch_size=64, batch_size_sim=16, num_workers=0)
    dataset = {
        "doc": datasets.Dataset.from_dict({"id": [f"doc{i}" for i in range(n)], "content": [str(i) for i in range(n)]}),
        "query": datasets.Dataset.from_dict({"id": [f"q{i}" for i in range(nq)], "generated_query": [str(i) for i in range(nq)],
                                             "content": [str(i) for i in range(nq)]}),
    }
    q_path, d_path = str(tmp_path / "q_idx"), str(tmp_path / "d_idx")
    out = r.retrieve(dataset, q_path, d_path, k)
    # reference chunk layout on disk: batch 64 -> save every 2343 batches -> a single final chunk here
    assert os.listdir(d_path) == ["embedding_chunk_15.pt"] and os.listdir(q_path) == ["embedding_chunk_0.pt"]
    assert isinstance(out["score"], torch.Tensor) and out["score"].dtype == torch.float32 and not out["score"].is_cuda
    assert tuple(out["score"].shape) == (nq, k) and out["q_id"] == [f"q{i}" for i in range(nq)]
    assert isinstance(out["doc_id"][0][0], str)          # prepare_dataset_from_ids asserts this (utils.py:131)
    xq, xd = q_table.numpy(), d_table.numpy()
    if metric == "cos":
        xq, xd = c_oracle.l2_normalize_rows(xq), c_oracle.l2_normalize_rows(xd)
    ws, wi = c_oracle.canonical_search(xq, xd, k)
    got_i = np.array([[int(s[3:]) for s in row] for row in out["doc_id"]])
    compare.assert_bit_exact(out["score"].numpy(), got_i, ws, wi, f"Retrieve.retrieve {metric}")
    # second call: index folders exist -> no re-encode, resident index reused
    ix_before = r._resident[d_path][0]
    out2 = r.retrieve(dataset, q_path, d_path, k)
    assert r._resident[d_path][0] is ix_before and torch.equal(out2["score"], out["score"])
    # run file round trip
    run = str(tmp_path / "run.trec")
    utils.write_trec(run, out["q_id"], out["doc_id"], out["score"])
    q_ids, d_ids, scores = utils.load_trec(run)
    assert q_ids == out["q_id"] and d_ids == out["doc_id"]
    assert np.array_equal(np.array(scores, np.float32), out["score"].numpy())
    # an incomplete index raises the reference's IOError
    os.remove(os.path.join(d_path, "embedding_chunk_15.pt"))
    torch.save(d_table[:900], os.path.join(d_path, "embedding_chunk_15.pt"))
    with pytest.raises(IOError, match="Missing 100 documents"):
        r.retrieve(dataset, q_path, d_path, k)
    r.close()


def test_load_collection_and_retrieve_signature():
    import bergen_amd
    g = torch.Generator().manual_seed(3)
    x = torch.randn(700, 64, generator=g).half()
    q = torch.randn(9, 64, generator=g).half()
    r = bergen_amd.Retrieve(init_args=_TableDense(q, x, bergen_amd.DotProduct()), num_workers=0)
    s, i, e = r.load_collection_and_retrieve(q, [x[:300], x[300:]], 7, dataset_size=700)
    ws, wi = c_oracle.canonical_search(q.numpy(), x.numpy(), 7)
    compare.assert_bit_exact(s.numpy(), i.numpy(), ws, wi, "load_collection_and_retrieve")
    assert e is None and s.dtype == torch.float32 and i.dtype == torch.int64
    with pytest.raises(IOError, match="Missing 5 documents"):
        r.load_collection_and_retrieve(q, [x], 7, dataset_size=705)


def test_load_collection_and_retrieve_returns_embeddings_dense_and_sparse():
    """return_embeddings=True (reference retrieve.py:160-163,179-183): the [Q, k, D] rows of the hits — for sparse (SPLADE) chunks
    densified like the reference's `emb_chunk.to_dense()`, only for the rows retrieved."""
    import bergen_amd
    g = torch.Generator().manual_seed(5)
    x = torch.randn(500, 64, generator=g).half()
    q = torch.randn(6, 64, generator=g).half()
    r = bergen_amd.Retrieve(init_args=_TableDense(q, x, bergen_amd.DotProduct()), num_workers=0)
    s, i, e = r.load_collection_and_retrieve(q, [x[:200], x[200:]], 5, dataset_size=500, return_embeddings=True)
    assert tuple(e.shape) == (6, 5, 64) and torch.equal(e, x[i])
    # sparse chunks: non-negative weights over a vocabulary of 300 terms, ~12 per document
    V = 300
    dense_docs = torch.zeros(400, V)
    for row in range(400):
        cols = torch.randperm(V, generator=g)[:12]
        dense_docs[row, cols] = torch.rand(12, generator=g) + 0.05
    dense_docs = dense_docs.half()
    qs = torch.zeros(4, V)
    for row in range(4):
        qs[row, torch.randperm(V, generator=g)[:8]] = torch.rand(8, generator=g) + 0.05
    qs = qs.half()

    class _Sparse(_TableDense):
        sparse = True

    rs = bergen_amd.Retrieve(init_args=_Sparse(qs, dense_docs, bergen_amd.DotProduct()), num_workers=0)
    chunks = [dense_docs[:150].to_sparse(), dense_docs[150:].to_sparse()]
    s2, i2, e2 = rs.load_collection_and_retrieve(qs, chunks, 5, dataset_size=400, return_embeddings=True)
    assert tuple(e2.shape) == (4, 5, V) and torch.equal(e2, dense_docs[i2])
    want = (qs.double() @ dense_docs.double().T).topk(5, dim=1).values.float()
    assert torch.allclose(s2, want, rtol=0, atol=1e-6)


class _DeviceTableDense(_TableDense):
    """The same plug-in with its embeddings coming out of the 'encoder' on the GPU, as a real Dense model's do."""

    def __call__(self, query_or_doc, batch):
        return {"embedding": super().__call__(query_or_doc, batch)["embedding"].cuda()}


def test_resident_on_encode_skips_the_read_back(tmp_path, monkeypatch):
    """resident_on_encode=True: index() copies every encoded batch device-to-device into the resident index while it
    writes the reference's chunk files; retrieve() then searches without loading the document folder (SURVEY H-4)."""
    import datasets
    import bergen_amd
    from bergen_amd import utils
    g = torch.Generator().manual_seed(22)
    n, nq, d, k = 5000, 40, 768, 20
    d_table = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).half()
    q_table = torch.nn.functional.normalize(torch.randn(nq, d, generator=g), dim=1).half()
    dataset = {
        "doc": datasets.Dataset.from_dict({"id": [f"doc{i}" for i in range(n)], "content": [str(i) for i in range(n)]}),
        "query": datasets.Dataset.from_dict({"id": [f"q{i}" for i in range(nq)], "generated_query": [str(i) for i in range(nq)],
                                             "content": [str(i) for i in range(nq)]}),
    }
    q_path, d_path = str(tmp_path / "q_idx"), str(tmp_path / "d_idx")
    r = bergen_amd.Retrieve(init_args=_DeviceTableDense(q_table, d_table, bergen_amd.DotProduct()), batch_size=256, num_workers=0,
                            resident_on_encode=True)
    loaded = []
    real = utils.load_chunk
    monkeypatch.setattr(utils, "load_chunk", lambda f, **kw: (loaded.append(f), real(f, **kw))[1])
    out = r.retrieve(dataset, q_path, d_path, k)
    assert not any(f.startswith(d_path) for f in loaded), loaded          # the document folder was not read back
    assert sorted(os.listdir(d_path)) == ["embedding_chunk_19.pt"]        # ... but it was written, in the reference's layout
    assert torch.equal(torch.load(os.path.join(d_path, "embedding_chunk_19.pt")), d_table)
    ws, wi = c_oracle.canonical_search(q_table.numpy(), d_table.numpy(), k)
    got_i = np.array([[int(s[3:]) for s in row] for row in out["doc_id"]])
    compare.assert_bit_exact(out["score"].numpy(), got_i, ws, wi, "resident_on_encode")
    r.close()
    # a fresh Retrieve finds the folder and loads it (mmap + prefetch thread + pinned staging): same result
    r2 = bergen_amd.Retrieve(init_args=_DeviceTableDense(q_table, d_table, bergen_amd.DotProduct()), batch_size=256, num_workers=0)
    out2 = r2.retrieve(dataset, q_path, d_path, k)
    assert any(f.startswith(d_path) for f in loaded)
    assert torch.equal(out2["score"], out["score"]) and out2["doc_id"] == out["doc_id"]
    r2.close()


@pytest.mark.parametrize("dtype,dim", [(torch.float16, 768), (torch.float32, 768), (torch.float16, 100), (torch.float32, 129)])
def test_host_upload_through_pinned_staging(dtype, dim):
    """FlatIndex.upload of pageable host rows: double-buffered pinned staging (several 64 MiB pieces), with and without the
    fp32 -> fp16 conversion / row padding; the rows must land exactly."""
    import bergen_amd
    n = 200_003 if dim >= 512 else 700_001      # > 2 pieces of 64 MiB either way
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(n, dim, generator=g).to(dtype)
    ix = bergen_amd.FlatIndex(n, dim, metric="ip")
    ix.upload(x[:1000], row0=0)
    ix.upload(x[1000:], row0=1000)
    ix.finalize()
    q = x[[5, n // 2, n - 1]].half()
    s, i = ix.search(q.numpy(), 3)
    ix.close()
    assert i[:, 0].tolist() == [5, n // 2, n - 1]
    ws, wi = c_oracle.canonical_search(q.numpy(), x.half().numpy(), 3)
    compare.assert_bit_exact(s, i, ws, wi, f"pinned staging {dtype} {dim}")


@pytest.mark.parametrize("metric", ["ip", "cos"])
def test_sharded_search_through_the_stage_two_shards_on_one_gpu(tmp_path, monkeypatch, metric):
    """Retrieve(search_rank=r, search_world=2).retrieve(): each 'rank' keeps its row shard of the chunk folder resident
    (the shard boundary cuts through a chunk file), searches it with global row ids through the HIP kernels, the packed
    lists are gathered (the collective is replaced by a copy: both ranks live in this process, on this one GPU) and merged
    by the HIP merge kernel on rank 0.  Result = the single-index stage = the oracle, bit for bit; rank 1 returns None
    (search_results="rank0").  The gloo world-2/3 twin of this test runs in tests/test_retrieve_sharded_gloo.py."""
    import datasets
    import bergen_amd
    from bergen_amd import sharded
    g = torch.Generator().manual_seed(77)
    n, nq, d, k = 30_001, 300, 768, 50
    x = torch.randn(n, d, generator=g).half()
    x[20_000] = x[100]                     # a tie across the shard boundary
    q = torch.randn(nq, d, generator=g).half()
    q_path, d_path = str(tmp_path / "q"), str(tmp_path / "d")
    os.makedirs(q_path)
    os.makedirs(d_path)
    for j, (a, b) in enumerate([(0, 9000), (9000, 21_000), (21_000, n)]):
        torch.save(x[a:b].clone(), os.path.join(d_path, f"embedding_chunk_{10 * (j + 1)}.pt"))
    torch.save(q, os.path.join(q_path, "embedding_chunk_0.pt"))
    dataset = {"doc": datasets.Dataset.from_dict({"id": [f"doc{i}" for i in range(n)]}),
               "query": datasets.Dataset.from_dict({"id": [f"q{i}" for i in range(nq)]})}
    sim = bergen_amd.DotProduct() if metric == "ip" else bergen_amd.CosineSim()

    def stage(**kw):
        return bergen_amd.Retrieve(init_args=_TableDense(q, x, sim), batch_size=64, num_workers=0, **kw)

    sent = {}

    def fake_gather(flat, packed, group=None):
        per = packed.numel()
        sent[fake_gather.rank] = packed.clone()
        for r, p in sent.items():
            flat[r * per:(r + 1) * per].copy_(p)
    monkeypatch.setattr(sharded.dist, "all_gather_into_tensor", fake_gather)
    ranks = [stage(search_rank=r, search_world=2, search_results="rank0") for r in range(2)]
    for _ in range(2):                     # second round: shards resident, gather buffers reused
        fake_gather.rank = 1
        assert ranks[1].retrieve(dataset, q_path, d_path, k) is None
        fake_gather.rank = 0
        out = ranks[0].retrieve(dataset, q_path, d_path, k)
    assert ranks[0]._resident[d_path][0].n_rows == 15_001 and ranks[1]._resident[d_path][0].n_rows == 15_000
    single = stage()
    want = single.retrieve(dataset, q_path, d_path, k)
    assert torch.equal(out["score"], want["score"]) and out["doc_id"] == want["doc_id"] and out["q_id"] == want["q_id"]
    xq, xd = q.numpy(), x.numpy()
    if metric == "cos":
        xq, xd = c_oracle.l2_normalize_rows(xq), c_oracle.l2_normalize_rows(xd)
    ws, wi = c_oracle.canonical_search(xq, xd, k)
    got_i = np.array([[int(s[3:]) for s in row] for row in out["doc_id"]])
    compare.assert_bit_exact(out["score"].numpy(), got_i, ws, wi, f"sharded stage {metric}")
    for r in ranks + [single]:
        r.close()
