"""
Tiny `_target_` resolver standing in for hydra.utils.instantiate (hydra is not installed in this
image, SURVEY §5 "Config / flags").  The retriever yaml schema is the reference's
(config/retriever/*.yaml): ``init_args`` = {_target_, model_name, max_len, pooler, similarity,
prompt_q, prompt_d, query_encoder_name}; top-level keys batch_size / batch_size_sim become
Retrieve.__init__ kwargs (reference modules/rag.py:177-181).

Reference `_target_` paths (``models.retrievers.dense.*``) are mapped onto this package so a stock
BERGEN retriever yaml instantiates the MI355X-native classes unchanged.
"""
import importlib
from collections.abc import Mapping

# reference target -> native target
TARGET_ALIASES = {
    "models.retrievers.dense.Dense": "bergen_amd.dense.Dense",
    "models.retrievers.dense.MeanPooler": "bergen_amd.dense.MeanPooler",
    "models.retrievers.dense.ClsPooler": "bergen_amd.dense.ClsPooler",
    "models.retrievers.dense.DotProduct": "bergen_amd.dense.DotProduct",
    "models.retrievers.dense.CosineSim": "bergen_amd.dense.CosineSim",
    "models.retrievers.splade.Splade": "bergen_amd.splade.Splade",
    "modules.retrieve.Retrieve": "bergen_amd.retrieve.Retrieve",
    "models.rerankers.crossencoder.CrossEncoder": "bergen_amd.rerank.CrossEncoder",
    "modules.rerank.Rerank": "bergen_amd.rerank.Rerank",
}


def _locate(path):
    path = TARGET_ALIASES.get(path, path)
    module, _, name = path.rpartition(".")
    return getattr(importlib.import_module(module), name)


def instantiate(cfg, **overrides):
    """Recursively build the object described by a dict with a ``_target_`` key.

    Non-dict values pass through; dicts without ``_target_`` are instantiated member-wise.
    Uses hydra when it is importable and the config is an OmegaConf node.
    """
    if cfg is None:
        return None
    if not isinstance(cfg, dict):
        try:  # OmegaConf DictConfig when hydra is present
            from omegaconf import DictConfig, OmegaConf
            if isinstance(cfg, DictConfig):
                cfg = OmegaConf.to_container(cfg, resolve=True)
        except Exception:
            pass
    if not isinstance(cfg, dict) and isinstance(cfg, Mapping):
        cfg = dict(cfg)  # any other read-only mapping (a DictConfig without omegaconf importable, a MappingProxy ...)
    if not isinstance(cfg, dict):
        return cfg
    if "_target_" not in cfg:
        return {k: instantiate(v) for k, v in cfg.items()}
    kwargs = {k: instantiate(v) for k, v in cfg.items() if k != "_target_"}
    kwargs.update(overrides)
    return _locate(cfg["_target_"])(**kwargs)


def load_retriever_yaml(path):
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)
