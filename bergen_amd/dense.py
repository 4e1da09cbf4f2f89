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
import os
from abc import ABC, abstractmethod

import torch


def _native_encoder(model):
    """Move an HF BertModel-architecture encoder onto the hand-written gfx950 forward pass.

    On a GPU box every BERT-architecture checkpoint (all dense retrievers of the reference's
    config/retriever/*.yaml except repllama) runs on bergen_amd.BertEncoder; BERGEN_AMD_ENCODER=hf keeps the
    HF torch module (debugging / A-B comparison).  Other architectures stay on their HF implementation.
    """
    from .encoder import BertEncoder
    if isinstance(model, BertEncoder) or not torch.cuda.is_available():
        return model
    if os.environ.get("BERGEN_AMD_ENCODER", "hip") == "hf":
        return model
    if BertEncoder.supports(model):
        return BertEncoder.from_hf(model, device=torch.cuda.current_device())
    return model


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
    def pool(outputs, mask):
        outputs = outputs.masked_fill(~mask[..., None].bool(), 0.)
        return outputs.sum(dim=1) / mask.sum(dim=1)[..., None]


class ClsPooler:
    """h[:, 0]  (reference dense.py:71-75)."""

    @staticmethod
    def pool(outputs, *args):
        return outputs[:, 0]


class DotProduct:
    """q @ d^T (reference dense.py:77-81).  metric name understood by FlatIndex: 'ip'."""
    metric = "ip"

    @staticmethod
    def sim(query_embds, doc_embds):
        return torch.mm(query_embds, doc_embds.t())


class CosineSim:
    """Row-normalised q @ d^T (reference dense.py:83-89).  FlatIndex normalises once at load."""
    metric = "cos"

    @staticmethod
    def sim(query_embds, doc_embds):
        query_embds = query_embds / (torch.norm(query_embds, dim=-1, keepdim=True) + 1e-9)
        doc_embds = doc_embds / (torch.norm(doc_embds, dim=-1, keepdim=True) + 1e-9)
        return torch.mm(query_embds, doc_embds.t())


class Dense(Retriever):
    """Bi-encoder: tokenizer + transformer encoder + pooler; fp16 embeddings [B, d].

    ``model`` / ``query_encoder`` / ``tokenizer`` may be injected (offline boxes have no HF hub);
    otherwise they are loaded with transformers exactly like the reference (dense.py:16-20).
    The encoder is any module whose call returns a tuple / ModelOutput with the last hidden
    state [B, T, d] first — an HF ``AutoModel`` or ``bergen_amd.encoder.BertEncoder`` (the
    hand-written gfx950 forward pass).
    """

    def __init__(self, model_name, max_len, pooler, similarity, prompt_q=None, prompt_d=None,
                 query_encoder_name=None, model=None, query_encoder=None, tokenizer=None):
        self.model_name = model_name
        if model is None or tokenizer is None:
            from transformers import AutoModel, AutoTokenizer
        if model is None:
            model = AutoModel.from_pretrained(self.model_name, torch_dtype=torch.float16, trust_remote_code=True)
        self.model = _native_encoder(model)
        if query_encoder is not None:
            self.query_encoder = _native_encoder(query_encoder)
        elif query_encoder_name:
            self.query_encoder = _native_encoder(AutoModel.from_pretrained(query_encoder_name, torch_dtype=torch.float16,
                                                                           trust_remote_code=True))
        else:
            self.query_encoder = self.model  # otherwise symmetric (dense.py:19-20)
        self.tokenizer = tokenizer if tokenizer is not None else AutoTokenizer.from_pretrained(self.model_name)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if hasattr(self.model, "eval"):
            self.model.eval()
        if self.query_encoder is not self.model:
            self.query_encoder = self.query_encoder.to(self.device)
            if hasattr(self.query_encoder, "eval"):
                self.query_encoder.eval()
        self.max_len = max_len
        self.similarity = similarity
        self.pooler = pooler
        self.prompt_q = "" if prompt_q is None else prompt_q
        self.prompt_d = "" if prompt_d is None else prompt_d

    @torch.no_grad()
    def __call__(self, query_or_doc, kwargs):
        encoder = self.model if query_or_doc == "doc" else self.query_encoder
        # fused path: the native encoder takes the HOST BatchEncoding straight through the C ABI, pools on the
        # device and returns the [B, d] embedding (reference dense.py:38-46 in one call)
        if hasattr(encoder, "encode_pooled"):
            try:
                return {"embedding": encoder.encode_pooled(kwargs, self.pooler)}
            except ValueError:
                pass  # a pooler the kernels do not know: pool the hidden states in torch below
        kwargs = {key: value.to(self.device) for key, value in kwargs.items()}
        outputs = encoder(**kwargs)
        emb = self.pooler.pool(outputs[0], kwargs['attention_mask'])
        return {"embedding": emb}

    def collate_fn(self, batch, query_or_doc=None):
        key = 'generated_query' if query_or_doc == "query" else "content"
        content = [sample[key] for sample in batch]
        if query_or_doc == "query":
            content = ["{}{}".format(self.prompt_q, text) for text in content]
        if query_or_doc == "doc":
            content = ["{}{}".format(self.prompt_d, text) for text in content]
        return self.tokenizer(content, padding="longest", truncation="longest_first", max_length=self.max_len,
                              return_tensors='pt')

    def similarity_fn(self, query_embds, doc_embds):
        return self.similarity.sim(query_embds, doc_embds)
