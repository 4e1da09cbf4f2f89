"""
Rerank stage + cross-encoder plug-in (the consumer of the retrieval stage's top-k list) — same
constructor kwargs, methods and return dict as the reference, so `modules/rag.py` can drive it unchanged.

Reference -> here
  Rerank.__init__/eval/sort_by_score_indexes/get_clean_model_name   modules/rerank.py:16-68
  Reranker (ABC)                                                    models/rerankers/reranker.py:9-19
  CrossEncoder.__init__/collate_fn/__call__                         models/rerankers/crossencoder.py:13-39
On a gfx950 device a sequence-classification checkpoint of an architecture the kernels cover runs on
bergen_amd.BertEncoder — the hand-written encoder forward pass plus the pooler / classifier head kernel (`classify`, fp32
logits): BERT (BAAI/bge-large-en, MiniLM ...), RoBERTa / XLM-RoBERTa (bge-reranker-v2-m3 ...) and DeBERTa-v2 / v3 with
disentangled attention (naver/trecdl22-crossencoder-debertav3, the reference's default reranker, config/reranker/debertav3.yaml:3;
csrc/attention_rel.hip).  The reference pads every (query, passage) pair to max_len (padding="max_length",
crossencoder.py:30) and runs the padding through the model; the native path packs the attended tokens only.  Any other
architecture stays on its HF module, loudly (`.backend` says which path runs).
Differences: no torch.nn.DataParallel (crossencoder.py:20-21); scores are fp32 (the reference's
are the fp16 logits of an fp16 model); `sort_by_score_indexes` keeps Python's stable sort, i.e. ties stay in
retrieval order, as in the reference.
"""
from abc import ABC, abstractmethod

import torch
from torch.utils.data import DataLoader
from tqdm import tqdm

from . import config as _config
from .dense import _native_encoder


class Reranker(ABC):
    def __init__(self, model_name=None):
        self.model_name = model_name

    @abstractmethod
    def __call__(self, kwargs):
        pass

    @abstractmethod
    def collate_fn(self, batch, query_or_doc=None):
        pass


class CrossEncoder(Reranker):
    def __init__(self, model_name=None, max_len=512, model=None, tokenizer=None):
        self.model_name = model_name
        self.max_len = max_len
        if model is None or tokenizer is None:
            from transformers import AutoModelForSequenceClassification, AutoTokenizer
        if model is None:
            model = AutoModelForSequenceClassification.from_pretrained(self.model_name, low_cpu_mem_usage=True,
                                                                       torch_dtype=torch.float16)
        self.model = _native_encoder(model)
        self.tokenizer = tokenizer if tokenizer is not None else AutoTokenizer.from_pretrained(self.model_name,
                                                                                               max_length=self.max_len)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if hasattr(self.model, "eval"):
            self.model.eval()

    @property
    def backend(self):
        """'hip' when the model runs on the hand-written kernels, 'hf' when it stayed on torch (a warning was logged)."""
        from .dense import encoder_backend
        return encoder_backend(self.model)

    def collate_fn(self, examples, query_or_doc=None):
        question = [e['query'] for e in examples]
        doc = [e['doc'] for e in examples]
        q_id = [e['q_id'] for e in examples]
        d_id = [e['d_id'] for e in examples]
        inp_dict = self.tokenizer(question, doc, padding="max_length", truncation='only_second', max_length=self.max_len,
                                  return_tensors='pt')
        inp_dict['q_id'] = q_id
        inp_dict['d_id'] = d_id
        return inp_dict

    def collate_packed(self, examples):
        """collate_fn for the native path: the same pairs, truncation and special tokens as `collate_fn` (crossencoder.py:24-32), but
        padded to the batch's LONGEST pair instead of max_len — the HIP forward pass packs the attended tokens anyway (padding never
        reaches a kernel), so the logits are the same bits and the host builds / copies [B, longest] instead of [B, max_len] ids."""
        question = [e['query'] for e in examples]
        doc = [e['doc'] for e in examples]
        from .dense import fast_tokenize_pairs
        inp_dict = fast_tokenize_pairs(self.tokenizer, question, doc, self.max_len)  # (the same values without HF's Python post-processing)
        if inp_dict is None:
            inp_dict = self.tokenizer(question, doc, padding=True, truncation='only_second', max_length=self.max_len, return_tensors='pt')
        inp_dict['q_id'] = [e['q_id'] for e in examples]
        inp_dict['d_id'] = [e['d_id'] for e in examples]
        return inp_dict

    @property
    def native(self):
        """True when `__call__` runs the hand-written forward pass + classifier head (BertEncoder.classify)."""
        return bool(getattr(self.model, "num_labels", 0)) and hasattr(self.model, "classify")

    @torch.no_grad()
    def __call__(self, kwargs):
        if getattr(self.model, "num_labels", 0) and hasattr(self.model, "classify"):
            return {"score": self.model.classify(kwargs)}  # host BatchEncoding straight through the C ABI
        kwargs = {k: v.to(self.device) for k, v in kwargs.items()}
        return {"score": self.model(**kwargs).logits}


class Rerank:
    def __init__(self, init_args=None, batch_size=1, launch_pairs=256, num_workers=4):
        """init_args, batch_size: the reference's (modules/rerank.py:17-22; `batch_size` is the yaml's, e.g. 32 in
        config/reranker/debertav3.yaml).  launch_pairs / num_workers (not reference kwargs) shape the NATIVE path only: consecutive
        yaml batches are coalesced into launches of at least `launch_pairs` pairs (a 32-pair batch of a large cross-encoder is 5 k
        tokens: GEMMs of 22 x 4 tiles on 256 CUs), tokenised `num_workers` launches ahead on threads of this process.  The scores do
        not depend on how pairs are batched (tests/test_gpu_rerank.py::test_errors_and_batch_invariance), so the result is the
        reference loop's, bit for bit, whatever the two are set to; launch_pairs <= batch_size restores one launch per yaml batch."""
        self.batch_size = batch_size
        self.launch_pairs = int(launch_pairs)
        self.num_workers = int(num_workers)
        self.init_args = init_args
        self.model = _config.instantiate(self.init_args)  # yaml dict with _target_, or an already built reranker
        self.model_name = self.model.model_name.replace('/', '_')
        self.last_eval_stats = None

    def _eval_native(self, dataset):
        """The reference loop (modules/rerank.py:24-48) as a pipeline for the HIP path: rows [lo, hi) of the dataset -> tokeniser
        threads (padding to the launch's longest pair) -> BertEncoder.classify on >= launch_pairs pairs -> fp32 logits that STAY on
        the device until the last launch; one copy back.  Order of pairs, grouping and sorting are the reference's."""
        import sys
        import time
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        ce = self.model
        n = len(dataset)
        bs = max(1, int(self.batch_size))
        per_launch = bs if self.launch_pairs <= bs else -(-self.launch_pairs // bs) * bs  # whole yaml batches per launch

        def make(lo, hi):
            if hasattr(dataset, "select"):  # HF Dataset: one columnar slice instead of hi - lo row look-ups
                cols = dataset[lo:hi]
                names = list(cols)
                rows = [dict(zip(names, vals)) for vals in zip(*(cols[c] for c in names))]
            else:
                rows = [dataset[j] for j in range(lo, hi)]
            return ce.collate_packed(rows)

        from .utils import fit_host_pools_to_cpu_budget
        fit_host_pools_to_cpu_budget()
        seen_q, seen_d, parts = [], [], []
        flops = 0.0
        kernel_ms = 0.0
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 2e-4))  # (the GPU-driving thread needs the GIL for microseconds between launches)
        t0 = time.perf_counter()
        try:
            with ThreadPoolExecutor(max_workers=max(1, self.num_workers), thread_name_prefix="bergen-rerank-tok") as pool:
                starts = list(range(0, n, per_launch))
                pending = deque()
                nxt = 0
                progress = tqdm(total=len(starts), desc=f'Reranking: {ce.model_name}', file=sys.stderr)
                while nxt < len(starts) or pending:
                    while nxt < len(starts) and len(pending) < max(1, self.num_workers) + 1:
                        lo = starts[nxt]
                        pending.append(pool.submit(make, lo, min(n, lo + per_launch)))
                        nxt += 1
                    batch = pending.popleft().result()
                    seen_q.extend(batch.pop('q_id'))
                    seen_d.extend(batch.pop('d_id'))
                    parts.append(ce.model.classify(batch))  # [pairs, num_labels] fp32 on the device
                    c = ce.model.counters()
                    flops += float(c.get("flops", 0.0))
                    kernel_ms += float(c.get("forward_ms", 0.0))
                    progress.update(1)
                progress.close()
        finally:
            sys.setswitchinterval(old_interval)
        flat = (torch.cat(parts).reshape(-1).cpu() if parts else torch.zeros(0))  # the ONE copy back
        dt = time.perf_counter() - t0
        self.last_eval_stats = {"pairs": n, "launches": len(parts), "pairs_per_launch": per_launch, "seconds": dt,
                                "pairs_per_s": n / dt if dt > 0 else None, "algorithmic_flops": flops, "kernel_ms": kernel_ms,
                                "tokenizer_threads": self.num_workers}
        return flat, seen_q, seen_d

    @torch.no_grad()
    def eval(self, dataset):
        """Scores every (query, passage) row of `dataset` and returns, per query, the passages by descending score:
        {"score": list of 1-D tensors, "doc_id": list of lists, "q_id": list} (modules/rerank.py:24-48)."""
        if getattr(self.model, "native", False) and hasattr(self.model, "collate_packed"):
            flat, seen_q, seen_d = self._eval_native(dataset)
            q_sorted, d_sorted, s_sorted = self.sort_by_score_indexes(flat, seen_q, seen_d)
            return {"score": s_sorted, "doc_id": d_sorted, "q_id": q_sorted}
        target = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.model.model = self.model.model.to(target)
        loader = DataLoader(dataset, batch_size=self.batch_size, collate_fn=self.model.collate_fn)
        seen_q, seen_d, parts = [], [], []
        for batch in tqdm(loader, desc=f'Reranking: {self.model.model_name}'):
            seen_q.extend(batch.pop('q_id'))
            seen_d.extend(batch.pop('d_id'))
            parts.append(self.model(batch)['score'].detach().cpu())
        flat = torch.cat(parts).reshape(-1)
        q_sorted, d_sorted, s_sorted = self.sort_by_score_indexes(flat, seen_q, seen_d)
        self.model.model.to('cpu')
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        return {"score": s_sorted, "doc_id": d_sorted, "q_id": q_sorted}

    def sort_by_score_indexes(self, scores, q_ids, d_ids):
        """Group by query (first-appearance order), order each group by descending score; ties keep the incoming
        (retrieval) order, like Python's stable sort in the reference; the score lists stay ragged
        (modules/rerank.py:50-65)."""
        rows_of = {}
        for row, q in enumerate(q_ids):
            rows_of.setdefault(q, []).append(row)
        out_q, out_d, out_s = [], [], []
        for q, rows in rows_of.items():
            group = scores[torch.as_tensor(rows, dtype=torch.long)]
            order = torch.argsort(group, descending=True, stable=True)
            out_q.append(q)
            out_d.append([d_ids[rows[j]] for j in order.tolist()])
            out_s.append(group[order])
        return out_q, out_d, out_s

    def get_clean_model_name(self):
        return self.model_name
