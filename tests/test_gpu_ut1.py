"""UT1 known-answer test: the reference's only offline text fixture — tests/utdata/ut1_docs.tsv (100
passages) x ut1_queries.tsv (10 queries) — through `bergen_amd.Dense` + `bergen_amd.Retrieve` end to end (tokeniser, HIP
encoder, chunk files, resident index, fused search, doc-id strings), against the run the reference's OWN Dense + Retrieve
produced on the same seeded random-init checkpoint (oracle/make_golden_ut1.py -> tests/golden/ut1.npz; texts, ids and the
vocabulary travel inside the fixture, the GPU box has no /root/reference).

Tolerances (floating point, written here): embeddings vs the reference's fp32 pass — cosine >= 0.999 and
max-abs <= 3e-2 * max|ref| per embedding (the encoder tolerance of DESIGN.md); ranking vs the reference's fp32 run — the near-tie
rule of oracle/compare.py with gap 0.15 and |score - ref| <= 0.15 (0.2 against its fp16 run) (the reference's own fp16 pass differs from its fp32 pass
by up to ~0.06 on these scores of ~85-127).  The search itself is pinned bit for bit: the stage's result must equal the
oracle's exact search over the embeddings the stage wrote."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import c_oracle, compare
from tests import ut1_fixture


def _golden():
    return np.load(os.path.join(GOLDEN, "ut1.npz"))


def _checkpoint(tmp_path, z):
    path, checksum = ut1_fixture.build_checkpoint(str(tmp_path / "ut1_ckpt"), [str(w) for w in z["words"]])
    assert abs(checksum - float(z["checksum"])) < 1e-6, "the seeded checkpoint is not the one the reference ran on"
    return path


def _close(got, want, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    for r in range(got.shape[0]):
        cos = float(got[r] @ want[r] / (np.linalg.norm(got[r]) * np.linalg.norm(want[r]) + 1e-30))
        assert cos >= 0.999, f"{what} row {r}: cosine {cos}"
        assert np.abs(got[r] - want[r]).max() <= 3e-2 * np.abs(want[r]).max() + 1e-3, f"{what} row {r}"


def test_ut1_fixture_reproduces_the_reference_embeddings_on_the_cpu(tmp_path):
    """No GPU, no reference tree: the checkpoint rebuilt here is bit-identical to the one the reference ran on (checksum), and
    HF's fp32 forward of it + the reference's pooling formula reproduce the stored reference embeddings."""
    import transformers as T
    z = _golden()
    path = _checkpoint(tmp_path, z)
    tok = T.AutoTokenizer.from_pretrained(path)
    model = T.AutoModel.from_pretrained(path, torch_dtype=torch.float32).eval()
    texts = [str(t) for t in z["query_texts"]]
    batch = tok(texts, padding="longest", truncation="longest_first", max_length=int(z["max_len"]), return_tensors="pt")
    with torch.no_grad():
        hidden = model(**batch)[0]
    cls = hidden[:, 0].numpy()
    assert np.abs(cls - z["q_emb_cls_fp32"]).max() <= 2e-3 * np.abs(z["q_emb_cls_fp32"]).max()  # (stored through fp16 chunk files)


@pytest.mark.gpu
@pytest.mark.parametrize("pooler", ["cls", "mean"])
def test_ut1_end_to_end_against_the_reference_run(tmp_path, pooler):
    import datasets
    import bergen_amd
    from bergen_amd import utils
    z = _golden()
    path = _checkpoint(tmp_path, z)
    k = int(z["k"])
    d_ids, q_ids = [str(v) for v in z["doc_ids"]], [str(v) for v in z["query_ids"]]
    d_texts, q_texts = [str(v) for v in z["doc_texts"]], [str(v) for v in z["query_texts"]]
    pool = bergen_amd.ClsPooler() if pooler == "cls" else bergen_amd.MeanPooler()
    dense = bergen_amd.Dense(model_name=path, max_len=int(z["max_len"]), pooler=pool, similarity=bergen_amd.DotProduct())
    assert dense.backend == "hip", "the checkpoint did not resolve to the hand-written forward pass"
    stage = bergen_amd.Retrieve(init_args=dense, batch_size=16, batch_size_sim=2048, num_workers=0)
    dataset = {"doc": datasets.Dataset.from_dict({"id": d_ids, "content": d_texts}),
               "query": datasets.Dataset.from_dict({"id": q_ids, "content": q_texts, "generated_query": q_texts})}
    q_path, d_path = str(tmp_path / "q"), str(tmp_path / "d")
    out = stage.retrieve(dataset, q_path, d_path, k)
    assert out["q_id"] == q_ids and tuple(out["score"].shape) == (len(q_ids), k) and isinstance(out["doc_id"][0][0], str)
    # chunk layout of the reference: batch 16 -> one final chunk named after the last batch (retrieve.py:135-141)
    assert os.listdir(d_path) == ["embedding_chunk_6.pt"] and os.listdir(q_path) == ["embedding_chunk_0.pt"]
    q_emb, d_emb = utils.load_embeddings(q_path), utils.load_embeddings(d_path)
    assert q_emb.dtype == torch.float16 and tuple(d_emb.shape) == (100, 128)
    # (1) embeddings vs the reference's fp32 pass
    _close(q_emb.float().numpy(), z[f"q_emb_{pooler}_fp32"], f"UT1 query embeddings ({pooler})")
    _close(d_emb.float().numpy(), z[f"d_emb_{pooler}_fp32"], f"UT1 passage embeddings ({pooler})")
    # (2) the search is exact over the embeddings the stage wrote
    row_of = {d: i for i, d in enumerate(d_ids)}
    got_rows = np.array([[row_of[d] for d in row] for row in out["doc_id"]])
    ws, wi = c_oracle.canonical_search(q_emb.numpy(), d_emb.numpy(), k)
    compare.assert_bit_exact(out["score"].numpy(), got_rows, ws, wi, f"UT1 search ({pooler})")
    # (3) the run vs the reference's own run (near-tie rule; both precisions of the reference)
    for prec, gap in (("fp32", 0.15), ("fp16", 0.2)):
        ref_rows = np.array([[row_of[str(d)] for d in row] for row in z[f"run_ids_{pooler}_{prec}"]])
        st = compare.compare_near_tie(out["score"].numpy(), got_rows, z[f"run_scores_{pooler}_{prec}"], ref_rows, gap_tol=gap,
                                      score_tol=gap)
        assert st["queries"] == len(q_ids)
    stage.close()
