"""
Dense bi-encoder plug-in (seam 2 of SURVEY §8b) — same yaml schema, attributes and call
signatures as the reference's models/retrievers/dense.py:14-89, so a stock BERGEN `Retrieve`
can run on it and `bergen_amd.retrieve.Retrieve` can run on a stock reference `Dense`.

Reference -> here
  Dense.__init__/__call__/collate_fn/similarity_fn   dense.py:14-62
  MeanPooler.pool / ClsPooler.pool                   dense.py:64-75
  DotProduct.sim / CosineSim.sim                     dense.py:77-89
  Retriever ABC                                      models/retrievers/retriever.py:9-23

Differences: no torch.nn.DataParallel (dense.py:32-35) — multi-GPU encoding
range-partitions the dataset, one process per GPU, weights loaded once per process.
`similarity_fn` exists for API compatibility (it materialises the [Bq, n] matrix the fused
search kernel avoids); `bergen_amd.retrieve.Retrieve` never calls it.
"""
import logging
import os
import threading
from abc import ABC, abstractmethod

import torch

log = logging.getLogger("bergen_amd")
_warned = set()


def encoder_backend(model):
    """'hip' when `model` is the hand-written gfx950 forward pass, 'hf' when it is a torch / transformers module."""
    from .encoder import BertEncoder
    return "hip" if isinstance(model, BertEncoder) else "hf"


def _native_encoder(model, require_native=None, need_mlm_head=False):
    """Move an HF BertModel-architecture encoder onto the hand-written gfx950 forward pass.
    require_native (or BERGEN_AMD_REQUIRE_NATIVE=1 in the environment): raise instead of falling back to the HF module.
    need_mlm_head (the Splade plug-in): a converted encoder without the masked-LM head tensors is of no use to the caller —
    it stays on the HF module (whose `.logits` the plug-in then pools), loudly, like any other uncovered architecture.

    On a GPU box every BERT-architecture checkpoint (all dense retrievers of the reference's
    config/retriever/*.yaml except repllama) runs on bergen_amd.BertEncoder; BERGEN_AMD_ENCODER=hf keeps the
    HF torch module (debugging / A-B comparison).  Other architectures stay on their HF implementation — never
    silently: the reason is logged once per cause (logger "bergen_amd", WARNING) and the plug-ins expose the
    outcome as `.backend` ('hip' | 'hf').
    """
    from .encoder import BertEncoder
    if isinstance(model, BertEncoder):
        return model
    why = None
    if not torch.cuda.is_available():
        why = "no GPU visible"
    elif os.environ.get("BERGEN_AMD_ENCODER", "hip") == "hf":
        why = "BERGEN_AMD_ENCODER=hf"
    else:
        why = BertEncoder.unsupported_reason(model)
    if why is None:
        try:
            enc = BertEncoder.from_hf(model, device=torch.cuda.current_device())
            if not need_mlm_head or enc.has_mlm_head:
                return enc
            enc.close()
            why = f"no masked-LM head tensors recognised in {type(model).__name__}'s state dict (SPLADE pools the head's logits)"
        except (ValueError, TypeError, KeyError) as exc:
            # the configuration looked like one the kernels cover but the checkpoint's tensors do not fit it (names of a remote
            # modelling file, a missing tensor): that is a reason to stay on HF, not a crash of a run that worked before
            why = f"conversion of the checkpoint failed: {type(exc).__name__}: {exc}"
    elif need_mlm_head and not _has_mlm_head_names(model):
        # (no device needed to know: decided from the tensor names, so that a GPU-less box reports the same reason)
        why = f"{why}; no masked-LM head tensors recognised in {type(model).__name__}'s state dict"
    name = getattr(getattr(model, "config", None), "_name_or_path", None) or type(model).__name__
    if require_native is None:
        require_native = os.environ.get("BERGEN_AMD_REQUIRE_NATIVE", "0") not in ("", "0", "false", "False")
    if require_native:
        raise RuntimeError(f"bergen_amd: encoder {name} cannot run on the HIP forward pass ({why}) and require_native is set: "
                           f"refusing to fall back to the HF torch implementation")
    if (name, why) not in _warned:
        _warned.add((name, why))
        log.warning("bergen_amd: encoder %s stays on the HF torch implementation (%s); the HIP forward pass is NOT in use", name, why)
    try:
        model._bergen_amd_fallback_reason = why
    except Exception:  # noqa: BLE001 — (an object that refuses attributes: the log line above is the record)
        pass
    return model


def plugin_backends(plugin):
    """Backends of BOTH encoders of a retriever plug-in (ours or a stock reference one): {'doc': 'hip'|'hf', 'query': 'hip'|'hf'}.
    `.model` is the document encoder, `.query_encoder` the query side (the same object for symmetric models: dense.py:17-20)."""
    doc = getattr(plugin, "model", None)
    qry = getattr(plugin, "query_encoder", None)
    return {"doc": encoder_backend(doc), "query": encoder_backend(qry if qry is not None else doc)}


def _has_mlm_head_names(model):
    """True when `model`'s state dict carries a masked-LM head under one of the names canonical_state_dict maps
    (BERT cls.predictions.*, DistilBERT vocab_*, RoBERTa lm_head.*)."""
    try:
        names = list(model.state_dict().keys())
    except Exception:  # noqa: BLE001
        return False
    return any(n.startswith(("cls.predictions.transform.", "vocab_transform.", "lm_head.dense.")) for n in names)


class Retriever(ABC):
    """Reference models/retrievers/retriever.py:9-23 (as actually called: retrieve.py:129)."""

    def __init__(self, model_name=None):
        self.model_name = model_name

    @abstractmethod
    def __call__(self, query_or_doc, kwargs):
        pass

    @abstractmethod
    def collate_fn(self, batch, query_or_doc=None):
        pass

    @abstractmethod
    def similarity_fn(self, q_embs, doc_embs):
        pass


class MeanPooler:
    """sum_t h_t m_t / sum_t m_t  (reference dense.py:64-69)."""

    @staticmethod
    def pool(hidden, attention_mask):
        keep = attention_mask.bool().unsqueeze(-1)
        summed = torch.where(keep, hidden, torch.zeros((), dtype=hidden.dtype, device=hidden.device)).sum(dim=1)
        return summed / attention_mask.sum(dim=1).unsqueeze(-1)


class ClsPooler:
    """h[:, 0]  (reference dense.py:71-75)."""

    @staticmethod
    def pool(hidden, *unused):
        return hidden[:, 0]


class DotProduct:
    """q @ d^T (reference dense.py:77-81).  metric name understood by FlatIndex: 'ip'."""
    metric = "ip"

    @staticmethod
    def sim(q, d):
        return q @ d.transpose(0, 1)


def _unit_rows(x):
    # (the reference's epsilon, added to the norm: dense.py:87-88)
    return x / (x.norm(dim=-1, keepdim=True) + 1e-9)


class CosineSim:
    """Row-normalised q @ d^T (reference dense.py:83-89).  FlatIndex normalises once at load."""
    metric = "cos"

    @staticmethod
    def sim(q, d):
        return _unit_rows(q) @ _unit_rows(d).transpose(0, 1)


_TOKENIZER_CONFIG_LOCK = threading.Lock()
_FAST_TOKENIZE_WARNED = False


def fast_tokenize(tokenizer, texts, max_len):
    """`tokenizer(texts, padding="longest", truncation="longest_first", max_length=max_len, return_tensors='pt')` for a
    Rust-backed (PreTrainedTokenizerFast) tokenizer, without the Python post-processing of that call: HF configures the
    backend tokenizer's truncation / padding (set_truncation_and_padding), lets it encode the batch — that part is Rust,
    parallel, and releases the GIL — and then rebuilds every Encoding field by field in Python and pads again before the
    tensor conversion, which is 5-10x the encoding time (46 of 50 ms per 512 passages on the GPU box's host).  Here the
    padded ids / type ids / mask are read straight from the backend's Encodings.  Same values as the HF call
    (tests/test_host.py::test_fast_tokenize_equals_the_hf_call); None for a slow (Python) tokenizer — the caller then makes
    the HF call; any other surprise is logged once and answered the same way.
    Thread safety (the stage tokenises on `num_workers` threads, retrieve.py `_threaded_batches`): the backend tokenizer is
    CONFIGURED under a lock — HF's set_truncation_and_padding only mutates it when the requested settings differ from the
    current ones, so after the first call it is a read — and encode_batch, which borrows the backend immutably and releases
    the GIL, runs outside the lock.  A mutation while another thread encodes would raise PyO3's "Already borrowed"."""
    try:
        import numpy as np
        from transformers.tokenization_utils_base import BatchEncoding
        from transformers.utils import PaddingStrategy
        from transformers.tokenization_utils_base import TruncationStrategy
        backend = getattr(tokenizer, "backend_tokenizer", None)
        if backend is None or not getattr(tokenizer, "is_fast", False) or not hasattr(tokenizer, "set_truncation_and_padding"):
            return None
        if len(texts) == 0:
            return None  # (the HF call answers an empty batch; nothing to gain here)
        names = list(getattr(tokenizer, "model_input_names", ["input_ids", "token_type_ids", "attention_mask"]))
        pad_id = getattr(tokenizer, "pad_token_id", None)
        # Ragged form (right padding with a known pad id): the backend only truncates; the padded [B, T] matrix, the mask and
        # the all-zero type ids of single sequences are built here with numpy from the ids alone.  The Python objects of a batch
        # (every id of every field of every Encoding becomes a Python int, under the GIL) are what the tokeniser THREADS of
        # the encode stage fight the forward-pass thread over: ids-only without pad tokens is a fifth of the padded three-field form.
        # Ragged or padded is decided ONCE per tokenizer, under the lock, from a probe encoding (a single text must come back with
        # all-zero type ids — true of every BERT / RoBERTa post-processor template) and never flips afterwards: the backend's
        # configuration is then the same in every call, so no thread reconfigures it while another one encodes (PyO3 "Already
        # borrowed"), and no thread can be handed unpadded rows by a configuration another thread chose.
        with _TOKENIZER_CONFIG_LOCK:
            ragged = getattr(tokenizer, "_bergen_amd_ragged_ok", None)
            if ragged is None:
                ragged = pad_id is not None and getattr(tokenizer, "padding_side", "right") == "right"
                if ragged:
                    probe = backend.encode("a", add_special_tokens=True)
                    ragged = not any(probe.type_ids)
                try:
                    tokenizer._bergen_amd_ragged_ok = bool(ragged)
                except Exception:  # noqa: BLE001 — (an object that refuses attributes: padded form, which needs no memory)
                    ragged = False
            tokenizer.set_truncation_and_padding(padding_strategy=PaddingStrategy.DO_NOT_PAD if ragged else PaddingStrategy.LONGEST,
                                                 truncation_strategy=TruncationStrategy.LONGEST_FIRST,
                                                 max_length=max_len, stride=0, pad_to_multiple_of=None, padding_side=None)
        encode = getattr(backend, "encode_batch_fast", None) or backend.encode_batch  # (_fast: no offsets, the fields used here are the same)
        encs = encode(list(texts), add_special_tokens=True)
        out = {}
        if ragged:
            import itertools
            rows = [e.ids for e in encs]
            lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
            width = int(lens.max())
            mask = np.arange(width, dtype=np.int64)[None, :] < lens[:, None]
            ids = np.full((len(rows), width), int(pad_id), dtype=np.int64)
            ids[mask] = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int64, count=int(lens.sum()))
            out["input_ids"] = torch.from_numpy(ids)
            if "token_type_ids" in names:
                out["token_type_ids"] = torch.zeros((len(rows), width), dtype=torch.int64)
            if "attention_mask" in names:
                out["attention_mask"] = torch.from_numpy(mask.astype(np.int64))
            return BatchEncoding(out)
        if "input_ids" in names or True:
            out["input_ids"] = torch.from_numpy(np.array([e.ids for e in encs], dtype=np.int64))
        if "token_type_ids" in names:
            out["token_type_ids"] = torch.from_numpy(np.array([e.type_ids for e in encs], dtype=np.int64))
        if "attention_mask" in names:
            out["attention_mask"] = torch.from_numpy(np.array([e.attention_mask for e in encs], dtype=np.int64))
        return BatchEncoding(out)
    except Exception as e:  # noqa: BLE001 — the HF call below is the same tokenisation, only slower; say so once
        global _FAST_TOKENIZE_WARNED
        if not _FAST_TOKENIZE_WARNED:
            _FAST_TOKENIZE_WARNED = True
            logging.getLogger("bergen_amd").warning("fast_tokenize fell back to the HF tokenizer call: %s: %s", type(e).__name__, e)
        return None


def fast_tokenize_pairs(tokenizer, first, second, max_len):
    """`tokenizer(first, second, padding=True, truncation='only_second', max_length=max_len, return_tensors='pt')` — the cross-encoder's
    collate (reference models/rerankers/crossencoder.py:24-32, padded to the batch's longest pair instead of max_len) — for a Rust-backed
    tokenizer, without the Python post-processing of the HF call (7x the encoding time for 256 pairs of ~180 tokens: the rerank stage's
    tokeniser threads then feed 5 k pairs/s to kernels that take 8.5 k).  The backend truncates (only the second sequence) and encodes
    the pairs — Rust, parallel, GIL released —; ids, type ids and the mask are built with numpy from the ragged rows, right-padded.
    Same values as the HF call (tests/test_rerank_oracle.py); None when the tokenizer is not a fast one, pads on the left or has no pad
    id — the caller then makes the HF call.  Thread safety as fast_tokenize: configuration under the lock, encoding outside."""
    try:
        import itertools

        import numpy as np
        from transformers.tokenization_utils_base import BatchEncoding, TruncationStrategy
        from transformers.utils import PaddingStrategy
        backend = getattr(tokenizer, "backend_tokenizer", None)
        pad_id = getattr(tokenizer, "pad_token_id", None)
        if (backend is None or not getattr(tokenizer, "is_fast", False) or not hasattr(tokenizer, "set_truncation_and_padding") or
                pad_id is None or getattr(tokenizer, "padding_side", "right") != "right" or len(first) == 0 or len(first) != len(second)):
            return None
        names = list(getattr(tokenizer, "model_input_names", ["input_ids", "token_type_ids", "attention_mask"]))
        with _TOKENIZER_CONFIG_LOCK:
            tokenizer.set_truncation_and_padding(padding_strategy=PaddingStrategy.DO_NOT_PAD, truncation_strategy=TruncationStrategy.ONLY_SECOND,
                                                 max_length=max_len, stride=0, pad_to_multiple_of=None, padding_side=None)
        encode = getattr(backend, "encode_batch_fast", None) or backend.encode_batch
        encs = encode(list(zip(first, second)), add_special_tokens=True)
        rows = [e.ids for e in encs]
        lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
        width, total = int(lens.max()), int(lens.sum())
        mask = np.arange(width, dtype=np.int64)[None, :] < lens[:, None]
        ids = np.full((len(rows), width), int(pad_id), dtype=np.int64)
        ids[mask] = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int64, count=total)
        out = {"input_ids": torch.from_numpy(ids)}
        if "token_type_ids" in names:
            types = np.full((len(rows), width), int(getattr(tokenizer, "pad_token_type_id", 0) or 0), dtype=np.int64)
            types[mask] = np.fromiter(itertools.chain.from_iterable(e.type_ids for e in encs), dtype=np.int64, count=total)
            out["token_type_ids"] = torch.from_numpy(types)
        if "attention_mask" in names:
            out["attention_mask"] = torch.from_numpy(mask.astype(np.int64))
        return BatchEncoding(out)
    except Exception as e:  # noqa: BLE001 — the HF call is the same tokenisation, only slower; say so once
        global _FAST_TOKENIZE_WARNED
        if not _FAST_TOKENIZE_WARNED:
            _FAST_TOKENIZE_WARNED = True
            logging.getLogger("bergen_amd").warning("fast_tokenize_pairs fell back to the HF tokenizer call: %s: %s", type(e).__name__, e)
        return None


class Dense(Retriever):
    """Bi-encoder: tokenizer + transformer encoder + pooler; fp16 embeddings [B, d].

    ``model`` / ``query_encoder`` / ``tokenizer`` may be injected (offline boxes have no HF hub);
    otherwise they are loaded with transformers exactly like the reference (dense.py:16-20).
    The encoder is any module whose call returns a tuple / ModelOutput with the last hidden
    state [B, T, d] first — an HF ``AutoModel`` or ``bergen_amd.encoder.BertEncoder`` (the
    hand-written gfx950 forward pass).
    """

    def __init__(self, model_name, max_len, pooler, similarity, prompt_q=None, prompt_d=None,
                 query_encoder_name=None, model=None, query_encoder=None, tokenizer=None, require_native=None):
        # require_native (not a reference kwarg; also BERGEN_AMD_REQUIRE_NATIVE=1): raise when an encoder cannot run on the HIP
        # forward pass instead of keeping its HF torch module with a warning
        self.model_name = model_name
        if model is None or tokenizer is None:
            from transformers import AutoModel, AutoTokenizer
        if model is None:
            model = AutoModel.from_pretrained(self.model_name, torch_dtype=torch.float16, trust_remote_code=True)
        self.model = _native_encoder(model, require_native)
        if query_encoder is not None:
            self.query_encoder = _native_encoder(query_encoder, require_native)
        elif query_encoder_name:
            self.query_encoder = _native_encoder(AutoModel.from_pretrained(query_encoder_name, torch_dtype=torch.float16,
                                                                           trust_remote_code=True), require_native)
        else:
            self.query_encoder = self.model  # otherwise symmetric (dense.py:19-20)
        self.tokenizer = tokenizer if tokenizer is not None else AutoTokenizer.from_pretrained(self.model_name)
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        for enc in {id(self.model): self.model, id(self.query_encoder): self.query_encoder}.values():
            if hasattr(enc, "eval"):
                enc.eval()
        if self.query_encoder is not self.model:
            self.query_encoder = self.query_encoder.to(self.device)
        self.max_len, self.pooler, self.similarity = max_len, pooler, similarity
        self.prompt_q = prompt_q or ""
        self.prompt_d = prompt_d or ""

    @property
    def backend(self):
        """'hip' when the document encoder runs on the hand-written kernels, 'hf' when it stayed on torch."""
        return encoder_backend(self.model)

    @property
    def backends(self):
        """{'doc': ..., 'query': ...}: an asymmetric plug-in (query_encoder_name) may have ONE side on the HF torch implementation."""
        return plugin_backends(self)

    @property
    def fallback_reason(self):
        """Why an encoder of this plug-in is NOT on the HIP forward pass (None when both are)."""
        for enc in (self.model, getattr(self, "query_encoder", None)):
            if enc is not None and encoder_backend(enc) != "hip":
                return getattr(enc, "_bergen_amd_fallback_reason", "injected torch module")
        return None

    @torch.no_grad()
    def __call__(self, query_or_doc, kwargs):
        encoder = self.model if query_or_doc == "doc" else self.query_encoder
        # fused path: the native encoder takes the HOST BatchEncoding straight through the C ABI, pools on the
        # device and returns the [B, d] embedding (reference dense.py:38-46 in one call)
        if hasattr(encoder, "encode_pooled"):
            from .encoder import pool_mode_or_none
            if pool_mode_or_none(self.pooler) is not None:
                # (errors of the forward pass itself — sequence too long, token id out of range, empty mask — propagate)
                return {"embedding": encoder.encode_pooled(kwargs, self.pooler)}
            # a pooler the kernels do not know: pool the hidden states in torch below
        on_device = {name: t.to(self.device) for name, t in kwargs.items()}
        hidden = encoder(**on_device)[0]
        return {"embedding": self.pooler.pool(hidden, on_device["attention_mask"])}

    def collate_fn(self, batch, query_or_doc=None):
        """Texts of the batch (queries: `generated_query`, documents: `content`) with the configured prompt in front,
        tokenised to the longest of the batch (reference dense.py:49-58)."""
        is_query = query_or_doc == "query"
        prefix = self.prompt_q if is_query else self.prompt_d if query_or_doc == "doc" else ""
        texts = [prefix + row['generated_query' if is_query else "content"] for row in batch]
        fast = fast_tokenize(self.tokenizer, texts, self.max_len)
        if fast is not None:
            return fast
        return self.tokenizer(texts, padding="longest", truncation="longest_first", max_length=self.max_len, return_tensors='pt')

    def similarity_fn(self, q, d):
        """[Bq, dim] x [n, dim] -> [Bq, n] (API compatibility; bergen_amd.Retrieve never materialises this matrix)."""
        return self.similarity.sim(q, d)
