"""
Retrieve — the retrieval stage (seam 1 of SURVEY §8b), signature- and return-compatible with the
reference's modules/retrieve.py:20-197, backed by the resident HBM index and the fused gfx950
search kernels (C ABI: include/bergen_hip.h).

Reference -> here
  Retrieve.__init__                          retrieve.py:21-34
  Retrieve.index                             retrieve.py:37-50
  Retrieve.retrieve                          retrieve.py:52-108
  Retrieve.encode_and_save                   retrieve.py:110-144   (same chunk files: names, cadence, dtype)
  Retrieve.load_collection_and_retrieve      retrieve.py:146-185
  get_clean_model_name / get_chunk_path      retrieve.py:193-197

What changes underneath (SURVEY Appendix A "deliberately differ"):
  * the doc chunks are uploaded to HBM once and stay resident across query sets (the reference
    re-uploads every chunk for every query chunk, retrieve.py:153);
  * scores are the canonical fp32 scores and ties are ordered (score desc, row asc) instead of
    torch.topk's arbitrary order;
  * doc-id strings are produced only for the Q*k hits.
Not carried over: the bm25 / oracle_provenance branches (Pyserini/JVM and a no-op; SURVEY §2 OOS).
"""
import os
import sys

import torch
from torch.utils.data import DataLoader
from tqdm import tqdm

from . import utils
from .config import instantiate
from .index import FlatIndex
from .sparse import SparseIndex

_INCOMPLETE = '!!! Index is not complete. Please re-index. Missing {} documents in the index. !!!'


def _metric_of(model):
    sim = getattr(model, "similarity", None)
    metric = getattr(sim, "metric", None)
    if metric is not None:
        return metric
    name = type(sim).__name__ if not isinstance(sim, type) else sim.__name__
    if name == "CosineSim":
        return "cos"
    if name == "DotProduct":
        return "ip"
    raise ValueError(f"cannot derive the search metric from similarity {sim!r}; expected DotProduct or CosineSim")


def _column(ds, name):
    """dataset['id'] for HF datasets, dicts of lists, or anything indexable by column name."""
    return ds[name]


class Retrieve:
    def __init__(self,
                 init_args=None,
                 batch_size=128,
                 batch_size_sim=1024,
                 pyserini_num_threads=1,
                 continue_batch=None,
                 device=0,
                 num_workers=4,
                 encode_rank=0,
                 encode_world=1):
        # encode_rank / encode_world: multi-GPU encoding.  The reference's only multi-GPU mechanism is
        # torch.nn.DataParallel around the encoder (dense.py:32-35: scatter inputs, re-broadcast every weight and
        # gather [B, T, d] outputs to GPU 0 on every forward).  Here each of `encode_world` processes (one per GPU)
        # encodes a contiguous range of BATCHES of the dataset into the same index folder — no collective; the
        # union of the chunk files is a valid index (same naming, ordering by the integer in the file name).
        self.encode_rank = int(encode_rank)
        self.encode_world = int(encode_world)
        self.continue_batch = continue_batch
        self.batch_size = batch_size
        self.batch_size_sim = batch_size_sim
        self.pyserini_num_threads = pyserini_num_threads
        self.device = device
        self.num_workers = num_workers
        # instantiate model (reference: hydra instantiate, retrieve.py:34); an already-built
        # plug-in object is accepted as well
        self.model = instantiate(init_args) if isinstance(init_args, dict) or init_args is None else init_args
        self._resident = {}  # doc_embeds_path -> (FlatIndex, signature)

    # ------------------------------------------------------------------ indexing (encode)
    def index(self, dataset, index_path, query_or_doc, overwrite_index=False):
        dataset = dataset[query_or_doc]
        # if dataset has not been encoded before (retrieve.py:40)
        if not os.path.exists(index_path) or self.continue_batch != None or overwrite_index:
            if self.model.model_name in ('bm25', 'oracle_provenance'):
                raise NotImplementedError(f"{self.model.model_name} is out of scope of the dense backend (SURVEY §2)")
            dataset = dataset.remove_columns(['id'])
            self._resident.pop(index_path, None)
            _ = self.encode_and_save(dataset, save_path=index_path, query_or_doc=query_or_doc)

    @torch.no_grad()
    def encode_and_save(self, dataset, save_path, query_or_doc, chunk_size=150000):
        """Encode a dataset into embedding_chunk_<last_batch_idx>.pt files (retrieve.py:110-144)."""
        save_every_n_batches = chunk_size // self.batch_size  # batch_size > 150000 -> ZeroDivisionError below, as in the reference
        total_n_batches = len(dataset) // self.batch_size + int(bool(len(dataset) % self.batch_size))
        os.makedirs(save_path, exist_ok=True)
        dataloader = DataLoader(
            dataset,
            batch_size=self.batch_size,
            collate_fn=lambda batch: self.model.collate_fn(batch, query_or_doc),
            num_workers=self.num_workers,
        )
        embs_list = list()
        dev = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.model.model = self.model.model.to(dev)
        # this process's contiguous range of batches [b_lo, b_hi)
        per_rank = -(-total_n_batches // self.encode_world)
        b_lo = min(total_n_batches, self.encode_rank * per_rank)
        b_hi = min(total_n_batches, b_lo + per_rank)
        if self.encode_world > 1:
            from torch.utils.data import Subset
            rows = range(b_lo * self.batch_size, min(len(dataset), b_hi * self.batch_size))
            dataloader = DataLoader(Subset(dataset, rows), batch_size=self.batch_size,
                                    collate_fn=lambda batch: self.model.collate_fn(batch, query_or_doc),
                                    num_workers=self.num_workers)
        for i, batch in tqdm(enumerate(dataloader, start=b_lo), total=b_hi - b_lo,
                             desc=f'Encoding: {self.model.model_name}', file=sys.stderr):
            if self.continue_batch != None:
                if i <= self.continue_batch:
                    continue
            outputs = self.model(query_or_doc, batch)
            emb = outputs['embedding']
            if save_path != None:
                emb = emb.detach().cpu()
            embs_list.append(emb)
            # save chunk (retrieve.py:135-141)
            if i % save_every_n_batches == 0 and i != 0 or i == b_hi - 1:  # (b_hi = total_n_batches for one process)
                chunk_save_path = self.get_chunk_path(save_path, i)
                embs = torch.cat(embs_list)
                if 'splade' in self.model.model_name or getattr(self.model, 'sparse', False):
                    embs = embs.to_sparse()
                torch.save(embs, chunk_save_path)
                embs_list = list()
        self.model.model = self.model.model.to('cpu')
        return None

    # ------------------------------------------------------------------ resident index
    def _build_resident(self, chunk_iter, dataset_size, dim, metric):
        """Upload chunks into a new FlatIndex of dataset_size rows; reference size check."""
        ix = FlatIndex(dataset_size, dim, metric=metric, device=self.device)
        num_emb = 0
        try:
            for emb_chunk in chunk_iter:
                n_c = emb_chunk.shape[0]
                if num_emb + n_c <= dataset_size:
                    ix.upload(emb_chunk, row0=num_emb)
                num_emb += n_c
            if num_emb != dataset_size:  # retrieve.py:165-166
                raise IOError(_INCOMPLETE.format(dataset_size - num_emb))
            ix.finalize()
        except Exception:
            ix.close()
            raise
        return ix

    @staticmethod
    def _dense_chunk(emb_chunk):
        if emb_chunk.is_sparse:
            emb_chunk = emb_chunk.to_dense()
        return emb_chunk

    def _resident_index(self, doc_embeds_path, dataset_size, metric):
        files = utils.sorted_chunk_files(doc_embeds_path)
        signature = (tuple(files), tuple(os.path.getmtime(f) for f in files), dataset_size, metric)
        hit = self._resident.get(doc_embeds_path)
        if hit is not None and hit[1] == signature:
            return hit[0]
        if hit is not None:
            hit[0].close()
            del self._resident[doc_embeds_path]
        if not files:
            raise IOError(_INCOMPLETE.format(dataset_size))

        def chunks():
            for f in tqdm(files, total=len(files), desc='Load embeddings into HBM...'):
                yield self._dense_chunk(utils.load_chunk(f))

        first = utils.load_chunk(files[0])
        dim = first.shape[1]
        sparse = bool(first.is_sparse)
        del first
        if sparse:  # SPLADE chunks (retrieve.py:138-139): keep them sparse, resident CSR index
            ix = self._build_resident_sparse((utils.load_chunk(f) for f in tqdm(files, total=len(files),
                                                                               desc='Load sparse embeddings into HBM...')),
                                             dataset_size, dim)
        else:
            ix = self._build_resident(chunks(), dataset_size, dim, metric)
        self._resident[doc_embeds_path] = (ix, signature)
        return ix

    def _build_resident_sparse(self, chunk_iter, dataset_size, vocab):
        ix = SparseIndex(dataset_size, vocab, device=self.device)
        num_emb = 0
        try:
            for emb_chunk in chunk_iter:
                n_c = emb_chunk.shape[0]
                if num_emb + n_c <= dataset_size:
                    ix.upload(emb_chunk, row0=num_emb)
                num_emb += n_c
            if num_emb != dataset_size:  # retrieve.py:165-166
                raise IOError(_INCOMPLETE.format(dataset_size - num_emb))
            ix.finalize()
        except Exception:
            ix.close()
            raise
        return ix

    # ------------------------------------------------------------------ search
    def retrieve(self, dataset, query_embeds_path, doc_embeds_path, top_k_documents, return_docs=False,
                 overwrite_index=False):
        # index if index doesn't exist (retrieve.py:54-56)
        self.index(dataset, query_embeds_path, query_or_doc='query', overwrite_index=overwrite_index)
        self.index(dataset, doc_embeds_path, query_or_doc='doc', overwrite_index=overwrite_index)

        q_ids = _column(dataset['query'], 'id')
        if self.model.model_name == "bm25":
            raise NotImplementedError("bm25 is out of scope of the dense backend (SURVEY §2)")

        query_embeds = utils.load_embeddings(query_embeds_path)
        sparse_queries = bool(query_embeds.is_sparse)
        if sparse_queries:
            query_embeds = query_embeds.to_dense()  # [Q, vocab], as the reference holds it (retrieve.py:75-76)
        if hasattr(self.model, "model") and hasattr(self.model.model, "to"):
            self.model.model = self.model.model.to('cpu')  # free HBM for the index (retrieve.py:78)

        metric = "sparse" if (sparse_queries or getattr(self.model, "sparse", False)) else _metric_of(self.model)
        index = self._resident_index(doc_embeds_path, dataset_size=len(dataset['doc']), metric=metric)

        # separate query embedding in chunks (retrieve.py:81) — one fused search per chunk
        chunks = torch.split(query_embeds, self.batch_size_sim, dim=0)
        scores_sorted_topk, indices_sorted_topk = list(), list()
        for chunk in tqdm(chunks, desc='Retrieving docs...', total=len(chunks)):
            s, i = index.search(chunk.contiguous(), top_k_documents)
            scores_sorted_topk.append(torch.from_numpy(s))
            indices_sorted_topk.append(torch.from_numpy(i))
        scores_sorted_topk = torch.cat(scores_sorted_topk, dim=0)
        indices_sorted_topk = torch.cat(indices_sorted_topk, dim=0)

        doc_ids = self._map_doc_ids(dataset['doc'], indices_sorted_topk)
        return {
            "score": scores_sorted_topk,
            "q_id": q_ids,
            "doc_id": doc_ids
        }

    @staticmethod
    def _map_doc_ids(doc_dataset, indices):
        """Row indices -> doc-id strings for the Q*k hits only (reference materialises all N ids,
        retrieve.py:58,103)."""
        idx = indices.numpy()
        uniq = sorted(set(int(v) for v in idx.reshape(-1) if v >= 0))
        if hasattr(doc_dataset, "select"):  # HF datasets.Dataset
            ids = doc_dataset.select(uniq)['id'] if uniq else []
        else:
            col = doc_dataset['id']
            ids = [col[i] for i in uniq]
        lut = dict(zip(uniq, ids))
        return [[lut[int(i)] for i in q_idxs if i >= 0] for q_idxs in idx]

    @torch.no_grad()
    def load_collection_and_retrieve(self, emb_q, doc_embeds, top_k_documents, detach_and_cpu=True,
                                     return_embeddings=False, dataset_size=None):
        """Same contract as retrieve.py:146-185 for an explicit list of chunk tensors."""
        num_emb = sum(int(c.shape[0]) for c in doc_embeds)
        if dataset_size is None:
            dataset_size = num_emb
        if num_emb != dataset_size:
            raise IOError(_INCOMPLETE.format(dataset_size - num_emb))
        dim = int(emb_q.shape[1])
        if len(doc_embeds) and doc_embeds[0].is_sparse:
            ix = self._build_resident_sparse(iter(doc_embeds), dataset_size, dim)
        else:
            ix = self._build_resident((self._dense_chunk(c) for c in doc_embeds), dataset_size, dim, _metric_of(self.model))
        try:
            q = emb_q.to_dense() if emb_q.is_sparse else emb_q
            s, i = ix.search(q.detach().cpu().contiguous(), top_k_documents)
        finally:
            ix.close()
        if return_embeddings and len(doc_embeds) and doc_embeds[0].is_sparse:
            raise NotImplementedError("return_embeddings is not supported for sparse indexes")
        final_top_k_scores = torch.from_numpy(s)
        final_top_k_indices = torch.from_numpy(i)
        if return_embeddings:
            all_rows = torch.cat([self._dense_chunk(c).cpu() for c in doc_embeds])
            return final_top_k_scores, final_top_k_indices, all_rows[final_top_k_indices.clamp(min=0)]
        return final_top_k_scores, final_top_k_indices, None

    # ------------------------------------------------------------------ misc (retrieve.py:190-197)
    def tokenize(self, example):
        return self.model.tokenize(example)

    def get_clean_model_name(self):
        return self.model.model_name.replace('/', '_')

    def get_chunk_path(self, save_path, chunk):
        return f'{save_path}/embedding_chunk_{chunk}.pt'

    def close(self):
        for ix, _ in self._resident.values():
            ix.close()
        self._resident.clear()
