"""
torch-CPU port of the reference's search loop.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

This is the reference's own op sequence — the same torch calls in the same order — so that it
can be timed on the GPU box's host cores (bench.py cpu_baseline, kind="port") where
/root/reference does not exist.  oracle/make_golden.py checks it against the real reference
code in this container.

Reference lines followed:
  DotProduct.sim / CosineSim.sim                    models/retrievers/dense.py:77-89
  Retrieve.load_collection_and_retrieve             modules/retrieve.py:146-185
  query chunking by batch_size_sim                  modules/retrieve.py:81, 92-98
  reference chunk sizes (150000 // batch_size)      modules/retrieve.py:112, 135-141
"""
import torch


def dot_product_sim(q, d):
    return torch.mm(q, d.t())                                             # dense.py:81


def cosine_sim(q, d):
    q = q / (torch.norm(q, dim=-1, keepdim=True) + 1e-9)                  # dense.py:87
    d = d / (torch.norm(d, dim=-1, keepdim=True) + 1e-9)                  # dense.py:88
    return torch.mm(q, d.t())                                             # dense.py:89


@torch.no_grad()
def load_collection_and_retrieve(emb_q, doc_embeds, top_k_documents, dataset_size, similarity_fn=dot_product_sim):
    """emb_q [Bq, d]; doc_embeds: list of chunk tensors.  Returns (scores fp32 [Bq,k], indices int64 [Bq,k])."""
    top_k_scores_list, top_k_indices_list = [], []
    num_emb = 0
    for emb_chunk in doc_embeds:                                          # retrieve.py:152
        scores_q = similarity_fn(emb_q, emb_chunk)                        # retrieve.py:154
        scores_sorted_q, indices_sorted_q = torch.topk(scores_q, top_k_documents, dim=1)   # :157
        top_k_scores_list.append(scores_sorted_q)
        top_k_indices_list.append(indices_sorted_q + num_emb)             # :159
        num_emb += emb_chunk.shape[0]                                     # :164
    if num_emb != dataset_size:                                           # :165-166
        raise IOError(f'!!! Index is not complete. Please re-index. Missing {dataset_size-num_emb} documents in the index. !!!')
    all_top_k_scores = torch.cat(top_k_scores_list, dim=1)                # :169
    all_top_k_indices = torch.cat(top_k_indices_list, dim=1)              # :170
    final_top_k_scores, top_k_indices = torch.topk(all_top_k_scores.float(), top_k_documents, dim=1)   # :175
    final_top_k_indices = torch.gather(all_top_k_indices, 1, top_k_indices)                            # :177
    return final_top_k_scores, final_top_k_indices


def reference_chunk_sizes(n_rows, batch_size=512, chunk_size=150000):
    """Row counts of the embedding_chunk_*.pt files encode_and_save writes (retrieve.py:110-141):
    a chunk is flushed at batch i when i % save_every == 0 and i != 0, or at the last batch."""
    save_every = chunk_size // batch_size                                  # :112
    total_batches = n_rows // batch_size + int(bool(n_rows % batch_size))  # :113
    sizes, pending = [], 0
    for i in range(total_batches):
        pending += min(batch_size, n_rows - i * batch_size)
        if (i % save_every == 0 and i != 0) or i == total_batches - 1:     # :135
            sizes.append(pending)
            pending = 0
    return sizes


@torch.no_grad()
def retrieve(queries, doc_chunks, top_k, batch_size_sim=2048, similarity_fn=dot_product_sim):
    """The query-chunk loop of Retrieve.retrieve (retrieve.py:81, 92-98)."""
    dataset_size = sum(c.shape[0] for c in doc_chunks)
    out_s, out_i = [], []
    for chunk in torch.split(queries, batch_size_sim, dim=0):
        s, i = load_collection_and_retrieve(chunk, doc_chunks, top_k, dataset_size, similarity_fn)
        out_s.append(s)
        out_i.append(i)
    return torch.cat(out_s, dim=0), torch.cat(out_i, dim=0)
