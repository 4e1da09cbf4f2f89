"""
Import the REAL reference modules (read-only, /root/reference).  TEST INFRASTRUCTURE ONLY.

Works only in the build container: the GPU box has no /root/reference, so nothing that runs
there (``-m gpu`` tests, smoke(), bench.py) may import this module.  It is used by
oracle/make_golden.py and by the ``needs_reference`` CPU tests, which skip when the tree is absent.

The reference hard-imports hydra / omegaconf / pytrec_eval, none of which is installed here
(SURVEY §8c), and hard-codes ``.to('cuda')`` (modules/retrieve.py:76,153).  We register three
stub modules and patch Tensor.to to ignore 'cuda' — the reference source itself is not modified.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BERGEN_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "modules", "retrieve.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference's Retrieve, dense module and utils."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # the reference tree is read-only
    import torch

    def _instantiate(cfg, *a, **k):
        raise RuntimeError("hydra.utils.instantiate stub: construct reference objects by hand")

    if "hydra" not in sys.modules:
        hydra = _stub("hydra")
        hydra.utils = _stub("hydra.utils", instantiate=_instantiate)
    if "omegaconf" not in sys.modules:
        _stub("omegaconf", OmegaConf=type("OmegaConf", (), {}), DictConfig=dict)
    if "pytrec_eval" not in sys.modules:
        _stub("pytrec_eval")
    # `.to('cuda')` -> no-op on a box without a GPU (retrieve.py:76,153 hard-code it)
    if not torch.cuda.is_available() and not getattr(torch.Tensor, "_bergen_to_patched", False):
        _orig_to = torch.Tensor.to

        def _to(self, *args, **kwargs):
            args = tuple(a for a in args if not (isinstance(a, str) and a.startswith("cuda")))
            if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
                kwargs.pop("device")
            if not args and not kwargs:
                return self
            return _orig_to(self, *args, **kwargs)

        torch.Tensor.to = _to
        torch.Tensor._bergen_to_patched = True
        # ... and for modules: `self.model.model.to('cuda')` (retrieve.py:124) on a real torch.nn.Module
        _orig_mod_to = torch.nn.Module.to

        def _mod_to(self, *args, **kwargs):
            args = tuple(a for a in args if not (isinstance(a, str) and a.startswith("cuda")))
            if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
                kwargs.pop("device")
            if not args and not kwargs:
                return self
            return _orig_mod_to(self, *args, **kwargs)

        torch.nn.Module.to = _mod_to
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    ref_utils = importlib.import_module("utils")
    ref_retrieve = importlib.import_module("modules.retrieve")
    ref_dense = importlib.import_module("models.retrievers.dense")
    ref_splade = importlib.import_module("models.retrievers.splade")
    ref_rerank = importlib.import_module("modules.rerank")
    ref_crossencoder = importlib.import_module("models.rerankers.crossencoder")
    _loaded = types.SimpleNamespace(utils=ref_utils, retrieve=ref_retrieve, dense=ref_dense, splade=ref_splade,
                                    rerank=ref_rerank, crossencoder=ref_crossencoder, Retrieve=ref_retrieve.Retrieve)
    return _loaded


def make_reference_retrieve(similarity="dot", batch_size=512, batch_size_sim=2048):
    """A reference Retrieve object with a fake model exposing only similarity_fn (no HF weights)."""
    ref = load()
    sim = ref.dense.DotProduct if similarity == "dot" else ref.dense.CosineSim
    r = ref.Retrieve.__new__(ref.Retrieve)
    r.batch_size = batch_size
    r.batch_size_sim = batch_size_sim
    r.continue_batch = None
    r.pyserini_num_threads = 1
    r.model = types.SimpleNamespace(similarity_fn=lambda q, d: sim.sim(q, d), model_name="fake/dense")
    return r
