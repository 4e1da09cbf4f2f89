"""CPU: host logic of the drop-in — formats, naming, config resolution, chunk writing, id mapping."""
import os

import numpy as np
import pytest
import torch

import bergen_amd
from bergen_amd import utils
from bergen_amd.config import instantiate
from bergen_amd.retrieve import Retrieve, _metric_of
from bergen_amd.sharded import shard_range


def test_write_trec_matches_reference_bytes(golden_dir, tmp_path):
    scores = np.load(os.path.join(golden_dir, "write_trec_scores.npy"))
    q_ids = ["q1", "q2"]
    d_ids = [["12", "7", "24853636"], ["3", "1", "0"]]
    for sc in (scores, torch.from_numpy(scores), scores.tolist()):
        out = tmp_path / "run.trec"
        utils.write_trec(str(out), q_ids, d_ids, sc)
        assert out.read_bytes() == open(os.path.join(golden_dir, "write_trec.trec"), "rb").read()
    q, d, s = utils.load_trec(os.path.join(golden_dir, "write_trec.trec"))
    assert q == q_ids and d == d_ids
    assert s[0] == [84.8125, 83.75, 8.769950866699219]   # float repr of the fp32 value


def test_load_embeddings_reads_reference_written_index(golden_dir):
    """tests/golden/ref_index was written by the reference's own encode_and_save (batch 4, chunk 12)."""
    path = os.path.join(golden_dir, "ref_index")
    files = utils.sorted_chunk_files(path)
    assert [os.path.basename(f) for f in files] == ["embedding_chunk_3.pt", "embedding_chunk_6.pt", "embedding_chunk_7.pt"]
    emb = utils.load_embeddings(path)
    assert emb.dtype == torch.float16 and tuple(emb.shape) == (30, 8)
    assert [utils.load_chunk(f).shape[0] for f in files] == [16, 12, 2]
    want = torch.tensor([[float(i % 7), float(i % 5), 1.0, 0.5 * i, -1.0, 2.0, 0.25, float(i % 3)] for i in range(30)]).half()
    assert torch.equal(emb, want)


def test_chunk_sort_key_uses_all_digits_of_the_path(tmp_path):
    # reference: int(''.join(filter(str.isdigit, path))) over the WHOLE path (utils.py:51)
    d = tmp_path / "idx_v2"
    d.mkdir()
    for i in (292, 1460, 584):
        torch.save(torch.full((1, 2), float(i)).half(), d / f"embedding_chunk_{i}.pt")
    emb = utils.load_embeddings(str(d))
    assert emb[:, 0].tolist() == [292.0, 584.0, 1460.0]


def test_load_embeddings_error_mapping(tmp_path):
    with pytest.raises(RuntimeError, match="No embeddings found"):
        utils.load_embeddings(str(tmp_path / "missing"))
    d = tmp_path / "corrupt"
    d.mkdir()
    (d / "embedding_chunk_0.pt").write_bytes(b"not a torch file")
    with pytest.raises(IOError, match="Embedding index corrupt"):
        utils.load_embeddings(str(d))


def test_path_naming():
    assert utils.get_index_path("indexes", "kilt-100w", "Shitao_RetroMAE_MSMARCO_distill", "doc") == \
        "indexes/kilt-100w_doc_Shitao_RetroMAE_MSMARCO_distill"
    assert utils.get_index_path("indexes", "kilt_nq", "m", "query", dataset_split="dev") == "indexes/kilt_nq_dev_query_m"
    assert utils.get_index_path("indexes", "kilt_nq", "m", "query", "dev", "gen") == "indexes/kilt_nq_dev_query_m.gen"
    assert utils.get_ranking_filename("runs", "kilt_nq", "kilt-100w", "Shitao_RetroMAE_MSMARCO_distill", "dev", 50, "copy") == \
        "runs/run.retrieve.top_50.kilt_nq.kilt-100w.dev.Shitao_RetroMAE_MSMARCO_distill.trec"


def test_instantiate_resolves_reference_targets():
    cfg = {"_target_": "models.retrievers.dense.CosineSim"}
    assert isinstance(instantiate(cfg), bergen_amd.CosineSim)
    assert isinstance(instantiate({"_target_": "models.retrievers.dense.ClsPooler"}), bergen_amd.ClsPooler)
    assert instantiate({"a": 1, "b": {"_target_": "bergen_amd.dense.MeanPooler"}})["a"] == 1
    assert _metric_of(type("M", (), {"similarity": bergen_amd.CosineSim()})()) == "cos"
    assert _metric_of(type("M", (), {"similarity": bergen_amd.DotProduct()})()) == "ip"


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 21_000_000, 24_853_637):
        for g in (1, 2, 3, 4, 8):
            r = [shard_range(n, i, g) for i in range(g)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(hi - lo <= (n + g - 1) // g for lo, hi in r)


def test_poolers_match_reference_math():
    h = torch.randn(3, 5, 8)
    m = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 0, 0, 0, 0]])
    want = (h * m[..., None]).sum(1) / m.sum(1, keepdim=True)
    assert torch.allclose(bergen_amd.MeanPooler.pool(h, m), want, atol=1e-6)
    assert torch.equal(bergen_amd.ClsPooler.pool(h, m), h[:, 0])


class _FakeEncoder(torch.nn.Module):
    def forward(self, v=None):
        return (v,)


class _FakeDense:
    """Deterministic stand-in for a Dense plug-in: text 'i' -> 8-dim fp16 vector."""
    model_name = "fake/dense"
    similarity = bergen_amd.DotProduct()

    def __init__(self):
        self.model = _FakeEncoder()

    def collate_fn(self, batch, query_or_doc=None):
        key = 'generated_query' if query_or_doc == "query" else "content"
        return {"v": torch.tensor([[float(int(s[key]) % 7), float(int(s[key]) % 5), 1.0, 0.5 * int(s[key]), -1.0, 2.0,
                                    0.25, float(int(s[key]) % 3)] for s in batch])}

    def __call__(self, query_or_doc, batch):
        return {"embedding": batch["v"].half()}


def test_encode_and_save_writes_the_reference_layout(tmp_path, golden_dir):
    """Same file names, cadence, dtype and bytes-level content as the reference's own writer."""
    import datasets
    r = Retrieve(init_args=_FakeDense(), batch_size=4, num_workers=0)
    ds = datasets.Dataset.from_dict({"content": [str(i) for i in range(30)]})
    out = tmp_path / "idx"
    r.encode_and_save(ds, save_path=str(out), query_or_doc="doc", chunk_size=12)
    assert sorted(os.listdir(out)) == ["embedding_chunk_3.pt", "embedding_chunk_6.pt", "embedding_chunk_7.pt"]
    ours = utils.load_embeddings(str(out))
    theirs = utils.load_embeddings(os.path.join(golden_dir, "ref_index"))
    assert ours.dtype == theirs.dtype and torch.equal(ours, theirs)
    # continue_batch resume (documentation/indexing.md:43): batches <= continue_batch are skipped
    r2 = Retrieve(init_args=_FakeDense(), batch_size=4, continue_batch=3, num_workers=0)
    out2 = tmp_path / "idx2"
    r2.encode_and_save(ds, save_path=str(out2), query_or_doc="doc", chunk_size=12)
    assert sorted(os.listdir(out2)) == ["embedding_chunk_6.pt", "embedding_chunk_7.pt"]
    assert r.get_clean_model_name() == "fake_dense"
    assert r.get_chunk_path("p", 5) == "p/embedding_chunk_5.pt"


def test_retrieve_defaults_match_reference_signature():
    import inspect
    sig = inspect.signature(Retrieve.__init__)
    assert sig.parameters["batch_size"].default == 128
    assert sig.parameters["batch_size_sim"].default == 1024
    assert sig.parameters["pyserini_num_threads"].default == 1
    assert sig.parameters["continue_batch"].default is None
    rsig = inspect.signature(Retrieve.retrieve)
    assert list(rsig.parameters)[1:] == ["dataset", "query_embeds_path", "doc_embeds_path", "top_k_documents",
                                         "return_docs", "overwrite_index"]


def test_map_doc_ids_only_touches_hits():
    import datasets
    ds = datasets.Dataset.from_dict({"id": [f"d{i}" for i in range(100)], "content": ["x"] * 100})
    idx = torch.tensor([[5, 99, 0], [7, 5, -1]])
    assert Retrieve._map_doc_ids(ds, idx) == [["d5", "d99", "d0"], ["d7", "d5"]]
    assert Retrieve._map_doc_ids({"id": [str(i) for i in range(10)]}, torch.tensor([[3, 1]])) == [["3", "1"]]


def test_multi_process_encoding_writes_one_valid_index(tmp_path):
    """encode_rank / encode_world: every process encodes a contiguous range of batches into the same folder; the union
    equals the single-process index (same rows in the same order under the reference's chunk ordering)."""
    import torch
    import bergen_amd

    class FakeModel:
        model_name = "fake/enc"

        def __init__(self):
            self.model = torch.nn.Identity()

        def collate_fn(self, batch, query_or_doc=None):
            return {"x": torch.tensor([[float(s["content"])] for s in batch])}

        def __call__(self, query_or_doc, kwargs):
            return {"embedding": torch.cat([kwargs["x"], kwargs["x"] * 2], dim=1).half()}

    class DS(torch.utils.data.Dataset):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return {"content": str(i)}

    n, bs = 1003, 16
    one = tmp_path / "one"
    r = bergen_amd.Retrieve(init_args=FakeModel(), batch_size=bs, num_workers=0)
    r.encode_and_save(DS(n), str(one), "doc", chunk_size=160)
    want = bergen_amd.utils.load_embeddings(str(one))
    assert want.shape == (n, 2) and want[:, 0].tolist() == [float(i) for i in range(n)]
    many = tmp_path / "many"
    for rank in range(3):
        rr = bergen_amd.Retrieve(init_args=FakeModel(), batch_size=bs, num_workers=0, encode_rank=rank, encode_world=3)
        rr.encode_and_save(DS(n), str(many), "doc", chunk_size=160)
    got = bergen_amd.utils.load_embeddings(str(many))
    assert torch.equal(got, want)
