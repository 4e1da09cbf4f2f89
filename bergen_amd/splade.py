"""
Splade plug-in (seam 2 of SURVEY §8b) — same yaml schema, attributes and call signatures as the reference's
models/retrievers/splade.py:12-56, so that `bergen_amd.Retrieve` (or a stock BERGEN `Retrieve`) can drive it.

Reference -> here
  Splade.__init__/__call__/collate_fn/similarity_fn   models/retrievers/splade.py:12-56
Differences:
  * the reference discards the query-encoder output and runs the document model a second time for every batch
    (splade.py:36-40); here the selected encoder runs once (this only changes results for asymmetric configs);
  * no torch.nn.DataParallel (splade.py:29-32): multi-GPU encoding range-partitions the dataset;
  * `sparse = True` tells `bergen_amd.Retrieve` to keep chunks sparse and search them with the CSR kernel
    (the reference decides by the substring 'splade' in the model name, modules/retrieve.py:138).
On a gfx950 device a BERT-architecture checkpoint (naver/splade-*: BertForMaskedLM) is moved onto
bergen_amd.BertEncoder: encoder layers, masked-LM head and the max-over-tokens pooling run as hand-written HIP
(`encode_splade`; the [B, T, vocab] logits are never materialised).  Other architectures stay on the HF module.
"""
import torch

from .dense import Retriever, _native_encoder


class Splade(Retriever):
    sparse = True

    def __init__(self, model_name, max_len=512, query_encoder_name=None, model=None, query_encoder=None, tokenizer=None):
        self.model_name = model_name
        self.max_len = max_len
        if model is None or tokenizer is None or (query_encoder is None and query_encoder_name):
            from transformers import AutoModelForMaskedLM, AutoTokenizer
        if model is None:
            model = AutoModelForMaskedLM.from_pretrained(self.model_name, low_cpu_mem_usage=True, torch_dtype=torch.float16)
        self.model = _native_encoder(model, need_mlm_head=True)
        if query_encoder is not None:
            self.query_encoder = _native_encoder(query_encoder, need_mlm_head=True)
        elif query_encoder_name:
            self.query_encoder = _native_encoder(AutoModelForMaskedLM.from_pretrained(
                query_encoder_name, torch_dtype=torch.float16, low_cpu_mem_usage=True), need_mlm_head=True)
        else:
            self.query_encoder = self.model  # otherwise symmetric
        self.tokenizer = tokenizer if tokenizer is not None else AutoTokenizer.from_pretrained(self.model_name,
                                                                                               max_length=self.max_len)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if hasattr(self.model, "eval"):
            self.model.eval()
        if self.query_encoder is not self.model:
            self.query_encoder = self.query_encoder.to(self.device)
            if hasattr(self.query_encoder, "eval"):
                self.query_encoder.eval()

    @property
    def backend(self):
        """'hip' when the model runs on the hand-written kernels, 'hf' when it stayed on torch (a warning was logged)."""
        from .dense import encoder_backend
        return encoder_backend(self.model)

    @property
    def backends(self):
        from .dense import plugin_backends
        return plugin_backends(self)

    @property
    def fallback_reason(self):
        from .dense import encoder_backend
        for enc in (self.model, self.query_encoder):
            if encoder_backend(enc) != "hip":
                return getattr(enc, "_bergen_amd_fallback_reason", "injected torch module")
        return None

    @torch.no_grad()
    def __call__(self, query_or_doc, kwargs):
        encoder = self.model if query_or_doc == "doc" else self.query_encoder
        from .encoder import BertEncoder
        if isinstance(encoder, BertEncoder):  # native: host BatchEncoding straight through the C ABI
            # (an injected BertEncoder without the head raises here by name — never `.logits` on its hidden-state tuple)
            return {"embedding": encoder.encode_splade(kwargs)}
        kwargs = {key: value.to(self.device) for key, value in kwargs.items()}
        logits = encoder(**kwargs).logits
        # pooling over hidden representations: max_t log(1 + relu(logit)) * mask   (splade.py:42-43)
        emb, _ = torch.max(torch.log(1 + torch.relu(logits)) * kwargs["attention_mask"].unsqueeze(-1), dim=1)
        return {"embedding": emb}

    def collate_fn(self, batch, query_or_doc=None):
        field = 'generated_query' if query_or_doc == "query" else "content"
        texts = [row[field] for row in batch]
        from .dense import fast_tokenize  # (padding=True / truncation=True are "longest" / "longest_first": the Dense call)
        fast = fast_tokenize(self.tokenizer, texts, self.max_len)
        if fast is not None:
            return fast
        return self.tokenizer(texts, padding=True, truncation=True, max_length=self.max_len, return_tensors='pt')

    def similarity_fn(self, q, d):
        """API compatibility only (materialises [Bq, n]); bergen_amd.Retrieve searches with the CSR kernel."""
        return torch.sparse.mm(q.to_sparse(), d.transpose(0, 1)).to_dense()
