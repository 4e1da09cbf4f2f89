"""
bergen_amd — MI355X-native dense-retrieval backend for BERGEN (naver/bergen).

Scope: ONE hot path, the reference's encode-then-exact-search stage
(modules/retrieve.py + models/retrievers/dense.py), rebuilt from scratch for gfx950 behind the
reference's own Retrieve / Retriever API.  Everything that computes runs in hand-written HIP
kernels reached through the C ABI of include/bergen_hip.h (bergen_amd/lib/libbergen_hip.so);
there is no CPU fallback.

  Retrieve            stage object, drop-in for modules.retrieve.Retrieve
  Dense, MeanPooler, ClsPooler, DotProduct, CosineSim
                      model plug-in, drop-in for models.retrievers.dense.*
  BertEncoder         BERT-architecture forward pass on hand-written HIP kernels (HF AutoModel drop-in)
  FlatIndex           resident HBM index + fused inner-product/top-k search
  SparseIndex, Splade resident CSR index + exact sparse search; SPLADE plug-in (models.retrievers.splade.Splade)
  merge_topk          device merge of per-shard partial top-k lists
  ShardedSearcher     row-sharded multi-GPU search (one process per GPU, RCCL all-gather)
  utils               chunk-file / .trec formats, path naming (reference utils.py)
"""
from . import evaluation, utils  # noqa: F401
from .config import instantiate  # noqa: F401
from .dense import ClsPooler, CosineSim, Dense, DotProduct, MeanPooler, Retriever  # noqa: F401
from .encoder import BertEncoder  # noqa: F401
from .index import FlatIndex, merge_topk  # noqa: F401
from .rerank import CrossEncoder, Rerank, Reranker  # noqa: F401
from .retrieve import Retrieve  # noqa: F401
from .sharded import ShardedSearcher, shard_range  # noqa: F401
from .sparse import SparseIndex  # noqa: F401
from .splade import Splade  # noqa: F401

__version__ = "0.1.0"
