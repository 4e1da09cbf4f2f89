"""
Regenerate tests/golden/sparse_small.npz from the REAL reference code.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden_sparse        (build container; needs /root/reference, read-only)

The reference's SPLADE search is ``Splade.similarity_fn`` (models/retrievers/splade.py:55-56,
torch.sparse.mm) driven by ``Retrieve.load_collection_and_retrieve`` (modules/retrieve.py:146-185) over sparse COO
chunks.  Both are imported unmodified and run on CPU in fp32 on a small synthetic SPLADE-like corpus; the fixture
stores the corpus (CSR), the queries (CSR), the reference's outputs and the canonical-oracle outputs.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import synth  # noqa: E402
from oracle import c_oracle, ref_import  # noqa: E402

V, N, NQ, K = 30522, 2500, 12, 50


def main():
    ref = ref_import.load()
    d_ptr, d_terms, d_w = synth.random_sparse_corpus(N, V, seed=41)
    q_ptr, q_terms, q_w = synth.random_sparse_corpus(NQ, V, seed=42, mean_nnz=24, lo=4, hi=64)
    # duplicate a document (exact tie) and plant an empty query overlap case
    d_dense = synth.csr_to_dense(d_ptr, d_terms, d_w, V)
    q_dense = synth.csr_to_dense(q_ptr, q_terms, q_w, V)
    # reference path: sparse COO chunks (as encode_and_save stores them, retrieve.py:138-139), fp32 on CPU
    chunks = [torch.from_numpy(d_dense[a:b]).to_sparse() for a, b in ((0, 1000), (1000, 1900), (1900, N))]
    r = ref.Retrieve.__new__(ref.Retrieve)
    r.batch_size, r.batch_size_sim, r.continue_batch, r.pyserini_num_threads = 64, 128, None, 1
    r.model = types.SimpleNamespace(similarity_fn=lambda q, d: ref.splade.Splade.similarity_fn(None, q, d),
                                    model_name="naver/splade-v3")
    with torch.no_grad():
        s, i, _ = r.load_collection_and_retrieve(torch.from_numpy(q_dense), chunks, K, dataset_size=N)
    can_s, can_i = c_oracle.sparse_canonical_search(d_ptr, d_terms, d_w, V, q_dense.astype(np.float16), K)
    out = os.path.join(ROOT, "tests", "golden", "sparse_small.npz")
    np.savez_compressed(out, vocab=V, k=K, d_indptr=d_ptr, d_terms=d_terms, d_weights=d_w, q_indptr=q_ptr,
                        q_terms=q_terms, q_weights=q_w, ref_scores=s.numpy().astype(np.float32),
                        ref_ids=i.numpy().astype(np.int64), canonical_scores=can_s, canonical_ids=can_i)
    agree = float((i.numpy() == can_i).mean())
    print("wrote", out, os.path.getsize(out), "bytes; reference ids == canonical ids on", agree, "of entries;",
          "max |score diff|", float(np.abs(s.numpy() - can_s).max()))


if __name__ == "__main__":
    main()
