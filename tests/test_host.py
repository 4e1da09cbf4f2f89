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


@pytest.mark.parametrize("world", [1, 3])
def test_threaded_tokenisation_writes_the_same_index(tmp_path, world):
    """loader="threads" (default): num_workers threads run collate_fn ahead of the encoder, in order — same chunk files as
    the in-line loop and as the reference's DataLoader worker processes, also over a rank's Subset of the dataset."""
    import datasets
    import time
    ds = datasets.Dataset.from_dict({"content": [str(i) for i in range(203)]})

    class _Slow(_FakeDense):
        def collate_fn(self, batch, query_or_doc=None):
            time.sleep(0.002 * (int(batch[0]["content"]) % 3))  # batches finish out of order on the threads
            return super().collate_fn(batch, query_or_doc)

    outs = {}
    for name, kw in (("inline", dict(num_workers=0)), ("threads", dict(num_workers=3)), ("threads1", dict(num_workers=1)),
                     ("processes", dict(num_workers=2, loader="processes"))):
        path = str(tmp_path / name)
        for rank in range(world):
            r = Retrieve(init_args=_Slow(), batch_size=8, encode_rank=rank, encode_world=world, **kw)
            r.encode_and_save(ds, save_path=path, query_or_doc="doc", chunk_size=40)
        outs[name] = (sorted(os.listdir(path)), utils.load_embeddings(path))
    for name in ("threads", "threads1", "processes"):
        assert outs[name][0] == outs["inline"][0] and torch.equal(outs[name][1], outs["inline"][1]), name
    assert outs["inline"][1].shape[0] == 203
    with pytest.raises(ValueError):
        Retrieve(init_args=_FakeDense(), loader="fibres")


def test_fast_tokenize_equals_the_hf_call():
    """Dense.collate_fn reads the padded ids / type ids / mask straight from the Rust tokenizer's Encodings instead of going
    through HF's Python post-processing (dense.fast_tokenize): the BatchEncoding must equal the reference's call
    `tokenizer(texts, padding="longest", truncation="longest_first", max_length=..., return_tensors='pt')` (dense.py:57) —
    keys, order, dtypes, values — with truncation, an empty string, non-ASCII text and a one-text batch."""
    from bergen_amd.dense import fast_tokenize
    from tests import ut1_fixture
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ut1.npz"))
    tok, _ = ut1_fixture.tokenizer_for([str(w) for w in z["words"]])
    texts = [str(t) for t in z["doc_texts"][:40]] + ["", "x", "caf\u00e9 na\u00efve \u4e2d\u6587 text", "the " * 400]
    for batch, max_len in ((texts, 16), (texts, 128), (texts[:1], 128), (texts[40:41], 8)):
        want = tok(batch, padding="longest", truncation="longest_first", max_length=max_len, return_tensors="pt")
        got = fast_tokenize(tok, batch, max_len)
        assert got is not None and list(got.keys()) == list(want.keys())
        for k in want.keys():
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), (k, max_len)
    assert fast_tokenize(object(), texts, 16) is None  # not a fast tokenizer: the caller falls back to the HF call
    # the ragged / padded form is decided once per tokenizer and never flips: an empty batch decides nothing and is
    # left to the HF call; many threads tokenising at once never see another thread's configuration
    assert tok._bergen_amd_ragged_ok is True
    assert fast_tokenize(tok, [], 16) is None and tok._bergen_amd_ragged_ok is True
    import concurrent.futures
    want = tok(texts, padding="longest", truncation="longest_first", max_length=32, return_tensors="pt")
    with concurrent.futures.ThreadPoolExecutor(8) as pool:
        outs = list(pool.map(lambda _: fast_tokenize(tok, texts, 32), range(48)))
    assert all(o is not None and all(torch.equal(o[k], want[k]) for k in want.keys()) for o in outs)
    # a tokenizer whose single-text encodings carry non-zero type ids stays on the padded form, for good
    tok2, _ = ut1_fixture.tokenizer_for([str(w) for w in z["words"]])
    from tokenizers import processors
    tok2.backend_tokenizer.post_processor = processors.TemplateProcessing(
        single="[CLS]:1 $A:1 [SEP]:1", pair="[CLS]:1 $A:1 [SEP]:1 $B:1 [SEP]:1",
        special_tokens=[("[CLS]", tok2.cls_token_id), ("[SEP]", tok2.sep_token_id)])
    got2 = fast_tokenize(tok2, texts[:5], 16)
    want2 = tok2(texts[:5], padding="longest", truncation="longest_first", max_length=16, return_tensors="pt")
    assert tok2._bergen_amd_ragged_ok is False and all(torch.equal(got2[k], want2[k]) for k in want2.keys())
    assert int(got2["token_type_ids"].max()) == 1


def test_piecewise_tokeniser_threads_give_the_whole_batch_result(monkeypatch):
    """With BERGEN_AMD_TOKENIZER_PIECES=1 on a many-core host the stage cuts every batch into eight pieces, each tokenised serially by one of many threads
    (Retrieve._threaded_batches), and pads the pieces back to the batch's longest row: the batches must equal collate_fn over the
    whole batch — values, shapes, order — and the process-wide settings it borrows (tokenizer parallelism, GIL switch interval)
    must be back afterwards.  A collate_fn whose output is not a set of [B, T] tensors falls back to whole batches."""
    import sys
    from bergen_amd import retrieve as R
    from bergen_amd.dense import fast_tokenize
    from tests import ut1_fixture
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ut1.npz"))
    tok, _ = ut1_fixture.tokenizer_for([str(w) for w in z["words"]])
    texts = [str(t) for t in z["doc_texts"]] * 3 + ["", "the " * 300]
    monkeypatch.setattr(os, "cpu_count", lambda: 64)
    monkeypatch.setenv("BERGEN_AMD_TOKENIZER_PIECES", "1")
    monkeypatch.delenv("TOKENIZERS_PARALLELISM", raising=False)

    class Model:
        model_name = "piecewise"
        tokenizer = tok

        def collate_fn(self, rows, query_or_doc):
            return fast_tokenize(tok, [r["content"] for r in rows], 32)

    class Odd(Model):
        def collate_fn(self, rows, query_or_doc):
            return [r["content"] for r in rows]  # not tensors

    src = [{"content": t} for t in texts]
    for model in (Model(), Odd()):
        stage = R.Retrieve.__new__(R.Retrieve)
        stage.model, stage.batch_size, stage.num_workers = model, 64, 4
        before = sys.getswitchinterval()
        got = list(stage._threaded_batches(src, "doc"))
        assert sys.getswitchinterval() == before and "TOKENIZERS_PARALLELISM" not in os.environ
        assert len(got) == -(-len(src) // 64)
        for j, batch in enumerate(got):
            want = model.collate_fn(src[j * 64:(j + 1) * 64], "doc")
            if isinstance(want, list):
                assert batch == want
            else:
                assert list(batch.keys()) == list(want.keys())
                for key in want.keys():
                    assert torch.equal(batch[key], want[key]), (j, key)


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


# ---- round 2: advisor findings ---------------------------------------------------------------------------------

def test_retrieve_instantiates_any_mapping_config():
    """rag.py builds Retrieve(**retriever_config) from hydra: init_args arrives as an OmegaConf DictConfig, which is a
    Mapping but not a dict.  It must be instantiated (not stored as the model), nested targets included."""
    from collections.abc import Mapping

    class FakeDictConfig(Mapping):  # what a DictConfig looks like to isinstance() when omegaconf is not importable
        def __init__(self, d):
            self._d = {k: FakeDictConfig(v) if isinstance(v, dict) else v for k, v in d.items()}

        def __getitem__(self, k):
            return self._d[k]

        def __iter__(self):
            return iter(self._d)

        def __len__(self):
            return len(self._d)

    cfg = FakeDictConfig({"_target_": "tests.test_host._Plugin", "model_name": "x/y",
                          "pooler": {"_target_": "models.retrievers.dense.MeanPooler"},
                          "similarity": {"_target_": "models.retrievers.dense.CosineSim"}})
    r = Retrieve(init_args=cfg)
    assert type(r.model).__name__ == "_Plugin" and r.model.model_name == "x/y"  # (the module may be imported under two names)
    assert isinstance(r.model.pooler, bergen_amd.MeanPooler) and isinstance(r.model.similarity, bergen_amd.CosineSim)
    assert r.get_clean_model_name() == "x_y"
    built = _Plugin("a", None, None)
    assert Retrieve(init_args=built).model is built  # an already-built plug-in passes through


class _Plugin:
    def __init__(self, model_name, pooler, similarity):
        self.model_name, self.pooler, self.similarity = model_name, pooler, similarity


def test_multi_rank_index_route_encodes_every_range_and_waits(tmp_path):
    """Retrieve.index() from several ranks into one folder: the folder's existence must not make a later rank skip its
    range, a finished rank must not encode twice, and wait_for_index() returns only when every range is there."""
    import datasets
    n, bs = 203, 8
    ds = {"doc": datasets.Dataset.from_dict({"id": [f"d{i}" for i in range(n)], "content": [str(i) for i in range(n)]})}
    folder = str(tmp_path / "shared")
    ranks = [Retrieve(init_args=_FakeDense(), batch_size=bs, num_workers=0, encode_rank=r, encode_world=3) for r in range(3)]
    ranks[1].index(ds, folder, "doc")                       # creates the folder first
    with pytest.raises(TimeoutError):
        ranks[1].wait_for_index(folder, n, timeout_s=0.2, poll_s=0.05)   # ranks 0 and 2 have not written yet
    ranks[0].index(ds, folder, "doc")                       # must NOT be skipped although the folder exists
    ranks[2].index(ds, folder, "doc")
    before = {f: os.path.getmtime(os.path.join(folder, f)) for f in os.listdir(folder)}
    ranks[0].index(ds, folder, "doc")                       # done marker present: no second encode
    assert before == {f: os.path.getmtime(os.path.join(folder, f)) for f in os.listdir(folder)}
    ranks[2].wait_for_index(folder, n, timeout_s=1.0)
    single = str(tmp_path / "single")
    Retrieve(init_args=_FakeDense(), batch_size=bs, num_workers=0).index(ds, single, "doc")
    assert torch.equal(utils.load_embeddings(folder), utils.load_embeddings(single))
    assert not [f for f in os.listdir(folder) if f.endswith(".tmp")]


def test_multi_rank_index_route_reuses_a_complete_folder_whatever_wrote_it(tmp_path):
    """A folder that already holds every row (single-process run, another world size or batch size, merge_indexes, a
    download) has no per-rank marker files: every rank must still take it as the cached index — no re-encode (which would
    append duplicate rows) and no wait (round-2 advisor finding; the reference caches by existence, retrieve.py:40)."""
    import datasets
    n = 203
    ds = {"doc": datasets.Dataset.from_dict({"id": [f"d{i}" for i in range(n)], "content": [str(i) for i in range(n)]})}
    folder = str(tmp_path / "cached")
    Retrieve(init_args=_FakeDense(), batch_size=16, num_workers=0).index(ds, folder, "doc")  # one process, another batch size
    before = {f: os.path.getmtime(os.path.join(folder, f)) for f in os.listdir(folder)}
    ranks = [Retrieve(init_args=_FakeDense(), batch_size=8, num_workers=0, encode_rank=r, encode_world=3) for r in range(3)]
    for r in ranks:
        r.index(ds, folder, "doc")
        r.wait_for_index(folder, n, timeout_s=0.2, poll_s=0.05)
    assert before == {f: os.path.getmtime(os.path.join(folder, f)) for f in os.listdir(folder)}
    assert utils.load_embeddings(folder).shape[0] == n
    # a folder that is NOT complete (and has no marker of this rank) is still encoded into
    short = str(tmp_path / "short")
    os.makedirs(short)
    torch.save(utils.load_embeddings(folder)[:40], os.path.join(short, "embedding_chunk_4.pt"))
    assert Retrieve._folder_rows(short) == 40
    ranks[2].index(ds, short, "doc")
    assert len(os.listdir(short)) == 2


def test_wait_for_index_times_out_on_lack_of_progress_only(tmp_path):
    import threading
    import time
    import datasets
    n, bs = 64, 8
    ds = {"doc": datasets.Dataset.from_dict({"id": [f"d{i}" for i in range(n)], "content": [str(i) for i in range(n)]})}
    folder = str(tmp_path / "slow")
    ranks = [Retrieve(init_args=_FakeDense(), batch_size=bs, num_workers=0, encode_rank=r, encode_world=2) for r in range(2)]
    ranks[0].index(ds, folder, "doc")

    def late():  # progress (an unrelated chunk file) arrives inside the window, the last range after more than one window
        time.sleep(0.25)
        torch.save(torch.zeros(1, 4), os.path.join(folder, "embedding_chunk_900.pt.tmp"))
        os.replace(os.path.join(folder, "embedding_chunk_900.pt.tmp"), os.path.join(folder, "embedding_chunk_900.pt"))
        time.sleep(0.25)
        os.remove(os.path.join(folder, "embedding_chunk_900.pt"))
        ranks[1].index(ds, folder, "doc")

    t = threading.Thread(target=late)
    t.start()
    ranks[0].wait_for_index(folder, n, timeout_s=0.4, poll_s=0.02)  # 0.5 s in total > 0.4 s, but never 0.4 s without a change
    t.join()
    assert utils.load_embeddings(folder).shape[0] == n


def test_dense_call_only_falls_back_for_an_unknown_pooler():
    """Errors of the native forward pass (BH_EINVAL -> ValueError) must propagate; only a pooler the kernels have no mode
    for goes through the unpooled path."""
    calls = []

    class Enc:
        def encode_pooled(self, kwargs, pooler):
            calls.append("pooled")
            raise ValueError("seq_len 9999 > max_position 512")

        def __call__(self, **kw):
            calls.append("unpooled")
            return (torch.zeros(2, 3, 4),)

        def eval(self):
            return self

    class OddPooler:
        @staticmethod
        def pool(hidden, mask):
            return hidden[:, -1]

    batch = {"input_ids": torch.zeros(2, 3, dtype=torch.long), "attention_mask": torch.ones(2, 3, dtype=torch.long)}
    d = bergen_amd.Dense("m", 8, bergen_amd.ClsPooler(), bergen_amd.DotProduct(), model=Enc(), tokenizer=object())
    with pytest.raises(ValueError, match="max_position"):
        d("doc", batch)
    assert calls == ["pooled"]
    calls.clear()
    d2 = bergen_amd.Dense("m", 8, OddPooler(), bergen_amd.DotProduct(), model=Enc(), tokenizer=object())
    assert d2("doc", batch)["embedding"].shape == (2, 4) and calls == ["unpooled"]
    assert d.backend == "hf"


def test_unsupported_encoders_are_reported_not_silent(caplog):
    """A model the HIP forward pass does not cover stays on HF torch with ONE warning naming the reason."""
    import logging
    from types import SimpleNamespace
    from bergen_amd.dense import _native_encoder, _warned
    from bergen_amd.encoder import BertEncoder
    # a BERT with relative_key position embeddings: outside the HIP forward pass (absolute / rotary / ALiBi positions only)
    cfg = SimpleNamespace(model_type="bert", position_embedding_type="relative_key", _name_or_path="org/bert-relative-key", hidden_size=768,
                          num_attention_heads=12, num_hidden_layers=12, intermediate_size=3072, vocab_size=30528, max_position_embeddings=512,
                          type_vocab_size=2, hidden_act="gelu")
    model = SimpleNamespace(config=cfg)
    assert "position_embedding_type 'relative_key'" in BertEncoder.unsupported_reason(model)
    assert BertEncoder.unsupported_reason(SimpleNamespace(config=SimpleNamespace(model_type="llama"))).startswith("model_type 'llama'")
    # a DeBERTa-v2 checkpoint outside deberta-v3's configuration (no relative attention) is refused with the reason
    assert "relative_attention is off" in BertEncoder.unsupported_reason(SimpleNamespace(config=SimpleNamespace(model_type="deberta-v2")))
    relu = SimpleNamespace(config=SimpleNamespace(model_type="bert", hidden_act="relu", hidden_size=768, num_attention_heads=12,
                                                  num_hidden_layers=2, intermediate_size=3072, vocab_size=100,
                                                  max_position_embeddings=64, type_vocab_size=2))
    assert "hidden_act 'relu'" in BertEncoder.unsupported_reason(relu)
    _warned.clear()
    with caplog.at_level(logging.WARNING, logger="bergen_amd"):
        assert _native_encoder(model) is model
        assert _native_encoder(model) is model
    msgs = [r.getMessage() for r in caplog.records if "stays on the HF torch implementation" in r.getMessage()]
    assert len(msgs) == 1 and "bert-relative-key" in msgs[0]


def test_reference_model_families_resolve_to_the_hip_forward_pass():
    """Architectures of the reference's shipped retriever / reranker configs that the kernels cover: BERT with 64- and
    32-dim heads (e5-small-v2.yaml:3, bge-small-en-v1.5.yaml:3, reranker/minilm6.yaml:3), DistilBERT (tasb.yaml:3),
    XLM-R (bge-m3.yaml:3), DeBERTa-v3 (reranker/debertav3.yaml:3: disentangled attention, deberta-v3-large's shape)."""
    from types import SimpleNamespace as NS
    from bergen_amd.encoder import BertEncoder, canonical_config
    bert = dict(hidden_act="gelu", num_hidden_layers=2, intermediate_size=1536, vocab_size=100, max_position_embeddings=64, type_vocab_size=2)
    assert BertEncoder.supports(NS(config=NS(model_type="bert", hidden_size=768, num_attention_heads=12, **bert)))
    small = canonical_config(NS(model_type="bert", hidden_size=384, num_attention_heads=12, **bert))
    assert small["head_dim"] == 32 and small["position_offset"] == 0
    distil = canonical_config(NS(model_type="distilbert", dim=768, n_heads=12, n_layers=6, hidden_dim=3072, activation="gelu",
                                 vocab_size=30522, max_position_embeddings=512))
    assert (distil["hidden_size"], distil["num_hidden_layers"], distil["type_vocab_size"], distil["head_dim"]) == (768, 6, 1, 64)
    xlmr = canonical_config(NS(model_type="xlm-roberta", hidden_size=1024, num_attention_heads=16, num_hidden_layers=24,
                               intermediate_size=4096, hidden_act="gelu", vocab_size=250002, max_position_embeddings=8194,
                               type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5))
    assert xlmr["position_offset"] == 2 and xlmr["layer_norm_eps"] == 1e-5
    deb = canonical_config(NS(model_type="deberta-v2", hidden_size=1024, num_attention_heads=16, num_hidden_layers=24, intermediate_size=4096,
                              hidden_act="gelu", vocab_size=128100, max_position_embeddings=512, type_vocab_size=0, layer_norm_eps=1e-7,
                              relative_attention=True, position_buckets=256, norm_rel_ebd="layer_norm", share_att_key=True,
                              pos_att_type=["p2c", "c2p"], position_biased_input=False, max_relative_positions=-1))
    assert (deb["rel_span"], deb["max_relative_positions"], deb["head_dim"], deb["type_vocab_size"]) == (256, 512, 64, 1)
    assert not BertEncoder.supports(NS(config=NS(model_type="new")))  # (a "new" config without its fields: refused, not guessed)
    # round 6: the remote architectures of gte-base / gte-large-en-v1.5 and jina-embeddings-v2-base-en (their published config.json fields)
    gte = canonical_config(NS(model_type="new", hidden_size=1024, num_attention_heads=16, num_hidden_layers=24, intermediate_size=4096, hidden_act="gelu",
                              vocab_size=30528, max_position_embeddings=8192, type_vocab_size=0, layer_norm_type="layer_norm", layer_norm_eps=1e-12,
                              position_embedding_type="rope", rope_theta=160000, rope_scaling={"type": "ntk", "factor": 2.0}, pack_qkv=True,
                              unpad_inputs=False, use_memory_efficient_attention=False, logn_attention_scale=False, logn_attention_clip1=False))
    assert (gte["ffn_gated"], gte["hidden_act"], gte["rotary_theta"], gte["head_dim"], gte["self_check"]) == (1, "gelu", 320000.0, 64, True)
    jina = canonical_config(NS(model_type="bert", hidden_size=768, num_attention_heads=12, num_hidden_layers=12, intermediate_size=3072, hidden_act="gelu",
                               vocab_size=30528, max_position_embeddings=8192, type_vocab_size=2, layer_norm_eps=1e-12, position_embedding_type="alibi",
                               feed_forward_type="geglu", emb_pooler="mean"))
    assert (jina["alibi"], jina["ffn_gated"], jina["self_check"], jina["head_dim"]) == (1, 1, True, 64)


def test_head_padding_keeps_the_attention_arithmetic():
    """Heads narrower than 64 dims are stored zero-padded to 64 with the query scaled by sqrt(64 / head_dim): the kernel's
    softmax(q k^T / sqrt(64)) v and output projection must equal the original softmax(q k^T / sqrt(head_dim)) v."""
    from bergen_amd.encoder import canonical_state_dict
    torch.manual_seed(0)
    d, nh, hd, T = 128, 4, 32, 9
    pre = "encoder.layer.0.attention."
    sd = {pre + f"self.{n}.weight": torch.randn(d, d) * 0.1 for n in ("query", "key", "value")}
    sd.update({pre + f"self.{n}.bias": torch.randn(d) * 0.1 for n in ("query", "key", "value")})
    sd[pre + "output.dense.weight"] = torch.randn(d, d) * 0.1
    cfg = dict(model_type="bert", hidden_size=d, num_attention_heads=nh, head_dim=hd, type_vocab_size=2)
    sd["embeddings.token_type_embeddings.weight"] = torch.zeros(2, d)
    pad = canonical_state_dict(cfg, sd)
    assert pad[pre + "self.query.weight"].shape == (nh * 64, d) and pad[pre + "output.dense.weight"].shape == (d, nh * 64)
    x = torch.randn(T, d)

    def attend(w, width, scale):
        q = (x @ w[pre + "self.query.weight"].T + w[pre + "self.query.bias"]).view(T, nh, width).transpose(0, 1)
        k = (x @ w[pre + "self.key.weight"].T + w[pre + "self.key.bias"]).view(T, nh, width).transpose(0, 1)
        v = (x @ w[pre + "self.value.weight"].T + w[pre + "self.value.bias"]).view(T, nh, width).transpose(0, 1)
        ctx = torch.softmax(q @ k.transpose(1, 2) * scale, dim=-1) @ v
        return ctx.transpose(0, 1).reshape(T, nh * width) @ w[pre + "output.dense.weight"].T

    assert torch.allclose(attend(sd, hd, hd ** -0.5), attend(pad, 64, 64 ** -0.5), atol=1e-5)


def test_distilbert_and_roberta_state_dicts_get_bert_names():
    from bergen_amd.encoder import canonical_state_dict
    d = 64
    cfg = dict(model_type="distilbert", hidden_size=d, num_attention_heads=1, head_dim=64, type_vocab_size=1)
    z = torch.zeros(1)
    src = {"distilbert.embeddings.word_embeddings.weight": z, "distilbert.embeddings.position_embeddings.weight": z,
           "distilbert.embeddings.LayerNorm.weight": z, "distilbert.embeddings.LayerNorm.bias": z}
    for part in ("attention.q_lin", "attention.k_lin", "attention.v_lin", "attention.out_lin", "sa_layer_norm", "ffn.lin1",
                 "ffn.lin2", "output_layer_norm"):
        src[f"distilbert.transformer.layer.0.{part}.weight"] = z
        src[f"distilbert.transformer.layer.0.{part}.bias"] = z
    got = set(canonical_state_dict(cfg, src))
    want = {"embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight", "embeddings.token_type_embeddings.weight",
            "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"}
    for part in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                 "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"):
        want |= {f"encoder.layer.0.{part}.weight", f"encoder.layer.0.{part}.bias"}
    assert got == want
    rob = canonical_state_dict(dict(model_type="xlm-roberta", hidden_size=d, num_attention_heads=1, head_dim=64, type_vocab_size=1),
                               {"roberta.embeddings.word_embeddings.weight": z, "classifier.dense.weight": z,
                                "classifier.out_proj.weight": z, "lm_head.dense.weight": z,
                                "roberta.embeddings.token_type_embeddings.weight": z})
    assert set(rob) == {"embeddings.word_embeddings.weight", "pooler.dense.weight", "classifier.weight",
                        "embeddings.token_type_embeddings.weight", "cls.predictions.transform.dense.weight"}
    # masked-LM heads (AutoModelForMaskedLM, reference splade.py:17-19): DistilBERT's vocab_* and RoBERTa's lm_head.* are
    # BertForMaskedLM's cls.predictions.* by other names; DistilBERT's sequence-classification head is NOT BertPooler's
    dmlm = canonical_state_dict(cfg, {"vocab_transform.weight": z, "vocab_transform.bias": z, "vocab_layer_norm.weight": z,
                                      "vocab_layer_norm.bias": z, "vocab_projector.weight": z, "vocab_projector.bias": z,
                                      "pre_classifier.weight": z, "classifier.weight": z})
    assert set(dmlm) == {"cls.predictions.transform.dense.weight", "cls.predictions.transform.dense.bias",
                         "cls.predictions.transform.LayerNorm.weight", "cls.predictions.transform.LayerNorm.bias",
                         "cls.predictions.decoder.weight", "cls.predictions.decoder.bias", "embeddings.token_type_embeddings.weight"}
    rmlm = canonical_state_dict(dict(model_type="roberta", hidden_size=d, num_attention_heads=1, head_dim=64, type_vocab_size=1),
                                {"lm_head.dense.bias": z, "lm_head.layer_norm.weight": z, "lm_head.decoder.weight": z,
                                 "lm_head.decoder.bias": z, "lm_head.bias": z})
    assert set(rmlm) == {"cls.predictions.transform.dense.bias", "cls.predictions.transform.LayerNorm.weight",
                         "cls.predictions.decoder.weight", "cls.predictions.decoder.bias", "cls.predictions.bias",
                         "embeddings.token_type_embeddings.weight"}


def test_splade_never_pools_a_headless_native_encoder(monkeypatch, caplog):
    """Round-5 review: a masked-LM checkpoint whose head the converter does not recognise must stay on the HF module (backend
    'hf', reason recorded) — never become a BertEncoder whose hidden-state tuple Splade.__call__ then asks for `.logits`."""
    import logging
    from types import SimpleNamespace
    from bergen_amd import dense as dense_mod
    from bergen_amd.encoder import BertEncoder
    from bergen_amd.splade import Splade

    class Headless(torch.nn.Module):  # an MLM class with unknown head names
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(model_type="bert", _name_or_path="toy/odd-mlm")
            self.odd_head = torch.nn.Linear(4, 4)

    closed = []
    fake = BertEncoder.__new__(BertEncoder)
    fake._h, fake.has_mlm_head = None, False
    monkeypatch.setattr(fake, "close", lambda: closed.append(1), raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(BertEncoder, "unsupported_reason", staticmethod(lambda m: None))
    monkeypatch.setattr(BertEncoder, "from_hf", classmethod(lambda cls, m, device=0: fake))
    model = Headless()
    dense_mod._warned.clear()
    with caplog.at_level(logging.WARNING, logger="bergen_amd"):
        kept = dense_mod._native_encoder(model, need_mlm_head=True)
    assert kept is model and closed == [1], "the head-less conversion must be dropped, the HF module kept"
    assert dense_mod.encoder_backend(kept) == "hf" and "masked-LM head" in kept._bergen_amd_fallback_reason
    assert any("masked-LM head" in r.getMessage() for r in caplog.records)
    with pytest.raises(RuntimeError, match="require_native"):
        dense_mod._native_encoder(model, require_native=True, need_mlm_head=True)
    assert dense_mod._native_encoder(model, need_mlm_head=False) is fake  # (the dense plug-in does not need the head)
    # an INJECTED head-less BertEncoder is a caller error reported by name, not an AttributeError on a tuple
    sp = Splade.__new__(Splade)
    sp.model = sp.query_encoder = fake
    sp.device = torch.device("cpu")
    with pytest.raises(RuntimeError, match="cls.predictions"):
        sp("doc", {"input_ids": torch.tensor([[3, 4]])})


def test_k_is_validated_before_the_index_is_built(tmp_path):
    import datasets
    n = 40
    ds = {"doc": datasets.Dataset.from_dict({"id": [f"d{i}" for i in range(n)], "content": [str(i) for i in range(n)]}),
          "query": datasets.Dataset.from_dict({"id": ["q0"], "generated_query": ["3"]})}
    r = Retrieve(init_args=_FakeDense(), batch_size=8, num_workers=0)
    built = []
    r._resident_index = lambda *a, **k: built.append(1)
    with pytest.raises(ValueError, match=r"top_k_documents=5000 outside 1\.\.4096"):
        r.retrieve(ds, str(tmp_path / "q"), str(tmp_path / "d"), 5000)
    assert not built
    assert bergen_amd.FlatIndex.MAX_K == 4096 and bergen_amd.SparseIndex.MAX_K == 4096


def test_prefetched_keeps_order_and_forwards_errors():
    from bergen_amd import utils
    assert list(utils.prefetched(range(7), lambda i: i * i, depth=2)) == [i * i for i in range(7)]
    assert list(utils.prefetched([], lambda i: i)) == []

    def boom(i):
        if i == 3:
            raise IOError("chunk 3 is corrupt")
        return i
    got = []
    with pytest.raises(IOError, match="chunk 3"):
        for v in utils.prefetched(range(6), boom):
            got.append(v)
    assert got == [0, 1, 2]
    # a consumer that stops early must not leave the worker blocked on a full queue
    it = utils.prefetched(range(100), lambda i: i, depth=1)
    assert next(it) == 0
    it.close()


def test_load_chunk_mmap_reads_the_same_tensor(tmp_path):
    import torch
    from bergen_amd import utils
    x = torch.randn(100, 16).half()
    f = str(tmp_path / "embedding_chunk_3.pt")
    torch.save(x, f)
    assert torch.equal(utils.load_chunk(f, mmap=True), x) and torch.equal(utils.load_chunk(f), x)
    torch.save(x, f, _use_new_zipfile_serialization=False)  # legacy container: cannot be mapped, falls back to a read
    assert torch.equal(utils.load_chunk(f, mmap=True), x)


def test_map_doc_ids_reads_the_arrow_table():
    """Row indices -> id strings: through the Arrow id column (and the dataset's indices mapping, if any), -1 entries
    dropped, same strings as the slow route the reference takes (materialise every id, retrieve.py:58,103)."""
    import datasets
    import torch
    from bergen_amd.retrieve import Retrieve
    n = 5000
    ds = datasets.Dataset.from_dict({"id": [f"doc-{j}" for j in range(n)], "content": ["x"] * n})
    rng = np.random.default_rng(3)
    idx = torch.from_numpy(rng.integers(0, n, size=(17, 9)))
    want = [[f"doc-{int(v)}" for v in row] for row in idx]
    assert Retrieve._map_doc_ids(ds, idx) == want
    assert all(isinstance(v, str) for v in Retrieve._map_doc_ids(ds, idx)[0])
    # a dataset that carries an indices mapping (select / shuffle / filter)
    view = ds.select(range(n - 1, -1, -1))
    assert Retrieve._map_doc_ids(view, idx) == [[f"doc-{n - 1 - int(v)}" for v in row] for row in idx]
    assert Retrieve._map_doc_ids(view, idx) == [[view[int(v)]["id"] for v in row] for row in idx]
    # short index: -1 marks "no such row"
    idx2 = idx.clone()
    idx2[2, 5:] = -1
    idx2[4, :] = -1
    got = Retrieve._map_doc_ids(ds, idx2)
    assert got[2] == want[2][:5] and got[4] == [] and got[0] == want[0]
    # anything with an 'id' column
    assert Retrieve._map_doc_ids({"id": ["a", "b", "c"]}, torch.tensor([[2, 0, -1]])) == [["c", "a"]]


def test_cpu_budget_follows_the_cgroup_quota(tmp_path, monkeypatch):
    """utils.cpu_budget: thread pools are sized for the CPUs the container may use (cgroup v2 cpu.max / v1 cfs quota), not for
    os.cpu_count() — a 1-GPU MI355X pod shows 256 logical CPUs under a quota of 16 and throttles a process that spins on all."""
    import os
    from bergen_amd import utils
    have = len(os.sched_getaffinity(0))
    (tmp_path / "v2").mkdir()
    (tmp_path / "v2" / "cpu.max").write_text("1600000 100000\n")
    assert utils.cpu_budget(str(tmp_path / "v2")) == min(have, 16)
    (tmp_path / "v2" / "cpu.max").write_text("max 100000\n")
    assert utils.cpu_budget(str(tmp_path / "v2")) == have
    (tmp_path / "v2" / "cpu.max").write_text("50000 100000\n")       # half a CPU: still one thread
    assert utils.cpu_budget(str(tmp_path / "v2")) == 1
    (tmp_path / "v1" / "cpu").mkdir(parents=True)
    (tmp_path / "v1" / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")
    (tmp_path / "v1" / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert utils.cpu_budget(str(tmp_path / "v1")) == min(have, 3)
    (tmp_path / "v1" / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")  # v1: no limit
    assert utils.cpu_budget(str(tmp_path / "v1")) == have
    assert utils.cpu_budget(str(tmp_path / "nothing-here")) == have
    # fit_host_pools_to_cpu_budget leaves a user's explicit settings alone
    monkeypatch.setenv("RAYON_NUM_THREADS", "5")
    n = utils.fit_host_pools_to_cpu_budget()
    assert n == utils.cpu_budget() and os.environ["RAYON_NUM_THREADS"] == "5"
