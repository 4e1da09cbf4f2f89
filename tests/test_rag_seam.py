"""CPU, needs the reference tree: the reference's UNMODIFIED ``RAG.retrieve`` (modules/rag.py:296-349) drives
``bergen_amd.Retrieve`` — SURVEY §8a row H14, the caller of the hot path.

What is real here: rag.py's method, the reference's path builders / write_trec / load_trec / get_by_id /
eval_retrieval_kilt, and bergen_amd.Retrieve's index() -> encode_and_save() -> retrieve() route with its chunk files
and id mapping.  What is stubbed: the GPU search behind ``Retrieve._resident_index`` (a numpy oracle shard with the same
interface — there is no GPU in this container; the kernels are pinned by the ``-m gpu`` tests), `rouge` (imported by
rag.py's siblings, absent here) and pytrec_eval (absent: its RelevanceEvaluator API is served by bergen_amd.evaluation,
so that the reference's eval_retrieval_kilt runs end to end)."""
import importlib
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

import bergen_amd
from oracle import numpy_oracle, ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")


class _OracleIndex:
    """The search interface of FlatIndex on the host: canonical scores (fp64 sum -> fp32), order (score desc, row asc)."""

    def __init__(self, rows, metric):
        assert metric == "ip"
        self.rows = rows

    def search(self, q, k, id_offset=0):
        s, i = numpy_oracle.canonical_search(np.asarray(q, dtype=np.float16), self.rows, k)
        return s, i + id_offset

    def close(self):
        pass


class _Encoder:
    """Deterministic toy bi-encoder plug-in: text -> 16-dim fp16 vector (no transformer weights offline)."""
    model_name = "toy/encoder"
    similarity = bergen_amd.DotProduct()

    def __init__(self):
        self.model = torch.nn.Identity()

    @staticmethod
    def _vec(text):
        rng = np.random.default_rng(abs(hash(text)) % (2 ** 31))
        return rng.standard_normal(16).astype(np.float16)

    def collate_fn(self, batch, query_or_doc=None):
        key = 'generated_query' if query_or_doc == "query" else "content"
        return {"v": torch.from_numpy(np.stack([self._vec(row[key]) for row in batch]))}

    def __call__(self, query_or_doc, batch):
        return {"embedding": batch["v"]}


@pytest.fixture()
def ref_rag(monkeypatch):
    ref = ref_import.load()
    if "rouge" not in sys.modules:
        stub = types.ModuleType("rouge")
        stub.Rouge = type("Rouge", (), {})
        monkeypatch.setitem(sys.modules, "rouge", stub)
    rag = importlib.import_module("modules.rag")

    class RelevanceEvaluator:  # pytrec_eval's API on bergen_amd.evaluation's arithmetic
        def __init__(self, qrel, measures):
            self.qrel, self.measures = qrel, set(measures)

        def evaluate(self, run):
            from bergen_amd.evaluation import ranking_metrics
            out = {}
            for q_id, docs in run.items():
                if q_id not in self.qrel:
                    continue
                k = next(int(m.split("_")[1]) for m in self.measures if m.startswith("recall_"))
                out[q_id] = ranking_metrics({q_id: docs}, self.qrel, top_k=k)
            return out

    monkeypatch.setattr(ref.utils.pytrec_eval, "RelevanceEvaluator", RelevanceEvaluator, raising=False)
    return ref, rag


def _dataset(n_docs=120, n_q=7):
    import datasets
    docs = datasets.Dataset.from_dict({"id": [f"p{i}" for i in range(n_docs)], "content": [f"passage {i}" for i in range(n_docs)],
                                       "wikipedia_id": [f"w{i // 3}" for i in range(n_docs)]})
    queries = datasets.Dataset.from_dict({"id": [f"q{i}" for i in range(n_q)], "content": [f"question {i}" for i in range(n_q)],
                                          "generated_query": [f"question {i}" for i in range(n_q)],
                                          "ranking_label": [[f"w{i}"] for i in range(n_q)]})
    docs.id2index = {f"p{i}": i for i in range(n_docs)}
    queries.id2index = {f"q{i}": i for i in range(n_q)}
    return {"doc": docs, "query": queries}


def test_reference_rag_retrieve_drives_our_retrieve(tmp_path, ref_rag, monkeypatch):
    ref, rag = ref_rag
    ds = _dataset()
    retriever = bergen_amd.Retrieve(init_args=_Encoder(), batch_size=16, num_workers=0)
    built = []

    def resident(path, dataset_size, metric, rows=None):
        rows = bergen_amd.utils.load_embeddings(path).numpy()
        assert rows.shape[0] == dataset_size
        built.append(path)
        return _OracleIndex(rows, metric)

    monkeypatch.setattr(retriever, "_resident_index", resident)
    qrels = tmp_path / "qrels"
    qrels.mkdir()
    json.dump({f"q{i}": {f"w{i}": 1} for i in range(7)}, open(qrels / "qrel.toyq.dev.json", "w"))
    for sub in ("runs", "indexes", "exp"):
        (tmp_path / sub).mkdir()
    me = types.SimpleNamespace(
        oracle_provenance=False, runs_folder=str(tmp_path / "runs"), index_folder=str(tmp_path / "indexes"),
        retriever=retriever, query_generator=types.SimpleNamespace(get_clean_model_name=lambda: "copy"),
        overwrite_exp=False, overwrite_index=False, experiment_folder=str(tmp_path / "exp"), datasets={"dev": ds},
        qrels_folder=str(qrels), generation_top_k=5, debug=False)

    q_ids, d_ids, scores = rag.RAG.retrieve(me, ds, "toyq", "toyd", "dev", 10)

    # ---- what the stage returned, against a direct computation
    q_emb = np.stack([_Encoder._vec(f"question {i}") for i in range(7)])
    d_emb = np.stack([_Encoder._vec(f"passage {i}") for i in range(120)])
    want_s, want_i = numpy_oracle.canonical_search(q_emb, d_emb, 10)
    assert q_ids == [f"q{i}" for i in range(7)]
    assert d_ids == [[f"p{j}" for j in row] for row in want_i]
    assert isinstance(d_ids[0][0], str) and torch.is_tensor(scores) and scores.dtype == torch.float32
    assert np.array_equal(scores.numpy().view(np.uint32), want_s.view(np.uint32))

    # ---- files: paths from the reference's builders, run file = reference write_trec of our output, copy in the experiment folder
    run_file = ref.utils.get_ranking_filename(str(tmp_path / "runs"), "toyq", "toyd", "toy_encoder", "dev", 10, "copy")
    assert os.path.exists(run_file) and os.path.exists(os.path.join(me.experiment_folder, os.path.basename(run_file)))
    want_dirs = [ref.utils.get_index_path(str(tmp_path / "indexes"), "toyd", "toy_encoder", "doc"),
                 ref.utils.get_index_path(str(tmp_path / "indexes"), "toyq", "toy_encoder", "query", dataset_split="dev",
                                          query_generator_name="copy")]
    assert sorted(os.path.join(str(tmp_path / "indexes"), d) for d in os.listdir(tmp_path / "indexes")) == sorted(want_dirs)
    assert want_dirs == [bergen_amd.utils.get_index_path(str(tmp_path / "indexes"), "toyd", "toy_encoder", "doc"),
                         bergen_amd.utils.get_index_path(str(tmp_path / "indexes"), "toyq", "toy_encoder", "query",
                                                         dataset_split="dev", query_generator_name="copy")]
    assert built == [ref.utils.get_index_path(str(tmp_path / "indexes"), "toyd", "toy_encoder", "doc")]
    ours_file = tmp_path / "ours.trec"
    bergen_amd.utils.write_trec(str(ours_file), q_ids, d_ids, scores)
    assert open(run_file).read() == open(ours_file).read()
    lq, ld, ls = bergen_amd.utils.load_trec(run_file)
    assert lq == q_ids and ld == d_ids

    # ---- the evaluation hand-off: the reference's eval_retrieval_kilt ran on page ids fetched with get_by_id
    metrics = json.load(open(os.path.join(me.experiment_folder, "eval_dev_ranking_metrics.json")))
    wiki = [ref.utils.get_by_id(ds["doc"], row, "wikipedia_id") for row in d_ids]
    ours = bergen_amd.evaluation.eval_retrieval_kilt(str(tmp_path), str(qrels), "toyq", "toyd", "dev", q_ids, wiki, scores,
                                                     top_k=5, write_trec=False)
    assert metrics == ours and set(metrics) == {"P_1", "recall_5"}

    # ---- cache by existence: a second call reads the run file, neither encodes nor searches
    built.clear()
    monkeypatch.setattr(retriever, "encode_and_save", lambda *a, **k: pytest.fail("re-encoded although the run exists"))
    q2, d2, s2 = rag.RAG.retrieve(me, ds, "toyq", "toyd", "dev", 10)
    assert (q2, d2) == (q_ids, d_ids) and not built
    assert np.allclose(np.asarray(s2, dtype=np.float32), scores.numpy())
    # overwrite_exp forces the search again but the indexes are reused (folders exist)
    me.overwrite_exp = True
    q3, d3, s3 = rag.RAG.retrieve(me, ds, "toyq", "toyd", "dev", 10)
    assert d3 == d_ids and len(built) == 1


@pytest.mark.parametrize("multi_doc", [False, True])
def test_joined_dataset_matches_reference_with_missing_ids_and_oracle_passages(multi_doc):
    """prepare_dataset_from_ids against the live reference where the inputs are awkward: doc ids the corpus does not know
    (both drop them), and oracle passages travelling with the query rows."""
    import datasets
    ref = ref_import.load()
    from bergen_amd import utils as U

    def toy():
        docs = datasets.Dataset.from_dict({"id": [f"d{i}" for i in range(20)], "content": [f"text {i}" for i in range(20)]})
        qs = datasets.Dataset.from_dict({"id": ["a", "b", "c"], "content": ["qa", "qb", "qc"], "label": [["x"], ["y"], ["z"]],
                                         "ranking_label": [["d1"], ["d2"], ["d3"]],
                                         "doc": [["o1", "o2"], ["o3", "o4"], ["o5", "o6"]], "doc_id": [["i1", "i2"], ["i3", "i4"], ["i5", "i6"]]})
        docs.id2index = {f"d{i}": i for i in range(20)}
        qs.id2index = {"a": 0, "b": 1, "c": 2}
        return {"doc": docs, "query": qs}

    q_ids, d_ids = ["c", "a"], [["d5", "nope", "d7"], ["d0", "d19", "zz"]]
    assert ref.utils.prepare_dataset_from_ids(toy(), q_ids, d_ids, multi_doc=multi_doc).to_dict() == \
        U.prepare_dataset_from_ids(toy(), q_ids, d_ids, multi_doc=multi_doc).to_dict()
    assert ref.utils.prepare_dataset_from_ids(toy(), q_ids, d_ids, multi_doc=multi_doc, oracle_provenance=True).to_dict() == \
        U.prepare_dataset_from_ids(toy(), q_ids, d_ids, multi_doc=multi_doc, oracle_provenance=True).to_dict()
