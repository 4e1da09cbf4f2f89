"""UT1 known-answer test.  TEST INFRASTRUCTURE ONLY; runs in the build container.

The reference's only offline text fixture — tests/utdata/ut1_docs.tsv (100 passages) x ut1_queries.tsv (10 queries) — goes
through the reference's OWN, unmodified code end to end: models.retrievers.dense.Dense (AutoModel + AutoTokenizer from a
checkpoint directory, dense.py:14-58) inside modules.retrieve.Retrieve (encode_and_save -> chunk files -> load_embeddings ->
load_collection_and_retrieve -> doc-id mapping, retrieve.py:37-185), top-10, CLS and mean pooling.  The checkpoint is the
seeded random-init BERT of tests/ut1_fixture.py (no hub access here).  The reference computes in fp16 (dense.py:16); on this
GPU-less box that is torch's CPU half path.  A second pass with the same weights in fp32 (the precision standard of every
other encoder fixture here) is stored beside it.

Writes tests/golden/ut1.npz: texts, ids, vocabulary, weight checksum, the reference's run (doc ids, scores) and embeddings.
    python -m oracle.make_golden_ut1
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402
from tests import ut1_fixture  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
K = 10


def read_tsv(path):
    rows = [l.rstrip("\n").split("\t") for l in open(path, encoding="utf-8") if l.strip()]
    return [r[0] for r in rows], [r[1] for r in rows]


class _IdListDataset:
    """`dataset['id']` as a plain list, as the datasets release the reference was written against returned it (this image's
    datasets 5.x returns a lazy Column that rejects the 0-d tensor indices of retrieve.py:103); everything else is the HF
    Dataset's own."""

    def __init__(self, ds):
        self._ds = ds

    def __getitem__(self, key):
        return list(self._ds[key]) if key == "id" else self._ds[key]

    def __len__(self):
        return len(self._ds)

    def __getattr__(self, name):
        return getattr(self._ds, name)


def run_reference(ref, ckpt, pooler_name, dataset, work, fp32):
    pooler = ref.dense.ClsPooler() if pooler_name == "cls" else ref.dense.MeanPooler()
    dense = ref.dense.Dense(model_name=ckpt, max_len=ut1_fixture.MAX_LEN, pooler=pooler, similarity=ref.dense.DotProduct())
    if fp32:
        dense.model = dense.model.float()
        dense.query_encoder = dense.model
    r = ref.Retrieve.__new__(ref.Retrieve)  # (the constructor goes through hydra.utils.instantiate, stubbed here)
    r.batch_size, r.batch_size_sim, r.continue_batch, r.pyserini_num_threads, r.model = 16, 2048, None, 1, dense
    tag = f"{pooler_name}_{'fp32' if fp32 else 'fp16'}"
    q_path, d_path = os.path.join(work, f"q_{tag}"), os.path.join(work, f"d_{tag}")
    with torch.no_grad():
        out = r.retrieve(dataset, q_path, d_path, K)
    q_emb = ref.utils.load_embeddings(q_path).float().numpy()
    d_emb = ref.utils.load_embeddings(d_path).float().numpy()
    return out, q_emb, d_emb


def main():
    import datasets
    ref = ref_import.load()
    ut = os.path.join(ref_import.REFERENCE_ROOT, "tests", "utdata")
    d_ids, d_texts = read_tsv(os.path.join(ut, "ut1_docs.tsv"))
    q_ids, q_texts = read_tsv(os.path.join(ut, "ut1_queries.tsv"))
    words = ut1_fixture.words_of(d_texts + q_texts)
    work = tempfile.mkdtemp(prefix="ut1_")
    try:
        ckpt, checksum = ut1_fixture.build_checkpoint(os.path.join(work, "ckpt"), words)
        dataset = {"doc": _IdListDataset(datasets.Dataset.from_dict({"id": d_ids, "content": d_texts})),
                   "query": _IdListDataset(datasets.Dataset.from_dict({"id": q_ids, "content": q_texts, "generated_query": q_texts}))}
        save = dict(doc_ids=np.array(d_ids), doc_texts=np.array(d_texts), query_ids=np.array(q_ids), query_texts=np.array(q_texts),
                    words=np.array(words), checksum=checksum, k=K, max_len=ut1_fixture.MAX_LEN)
        for pooler in ("cls", "mean"):
            for fp32 in (False, True):
                out, q_emb, d_emb = run_reference(ref, ckpt, pooler, dataset, work, fp32)
                tag = f"{pooler}_{'fp32' if fp32 else 'fp16'}"
                assert out["q_id"] == q_ids
                save[f"run_ids_{tag}"] = np.array(out["doc_id"])
                save[f"run_scores_{tag}"] = out["score"].numpy().astype(np.float32)
                save[f"q_emb_{tag}"] = q_emb.astype(np.float32)
                save[f"d_emb_{tag}"] = d_emb.astype(np.float32)
                print(tag, "top-3 of query 0:", out["doc_id"][0][:3], out["score"][0][:3].tolist())
        np.savez_compressed(os.path.join(GOLDEN, "ut1.npz"), **save)
        print("wrote", os.path.join(GOLDEN, "ut1.npz"), "checksum", checksum)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
