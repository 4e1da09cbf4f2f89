"""
BertEncoder — the bi-encoder forward pass on hand-written gfx950 kernels (C ABI: bh_encoder_*).

Drop-in for the HF ``AutoModel`` object the reference stores in ``Dense.model`` / ``Dense.query_encoder``
(models/retrievers/dense.py:16-20) and calls as ``self.model(**kwargs)[0]`` (dense.py:40-44):

  * ``encoder(input_ids=..., attention_mask=..., token_type_ids=...)`` returns a 1-tuple whose first element
    is the last hidden state ``[B, T, d]`` fp16 on the device (zeros at padding positions);
  * ``encoder.encode_pooled(kwargs, pooler)`` fuses the reference's pooler (ClsPooler / MeanPooler,
    dense.py:64-75) into the forward pass and returns the ``[B, d]`` fp16 embedding directly;
  * ``.to(device)``, ``.eval()``, ``.half()`` exist because ``Retrieve`` moves ``model.model`` around
    (modules/retrieve.py:78,124,142); the weights live in HBM inside the library and never move.

Weights come from any HF ``BertModel``-architecture module or state_dict (RetroMAE, contriever, e5, bge ...), from DistilBERT,
RoBERTa / XLM-R and DeBERTa-v2 / v3 checkpoints (renamed onto the same stack, `canonical_state_dict`), and from transformers' native
``NomicBertModel`` (nomic-embed-text-v1.5: rotary positions applied to the Q | K rows after their projection, gate and up rows of the
gated SiLU feed-forward interleaved into one GEMM whose epilogue folds them).
There is no CPU path: constructing an encoder without a gfx950 device raises.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_POOL = {"cls": 0, "mean": 1, "none": 2}


def _pool_mode(pooler):
    """Map a reference-style pooler (class or instance, ours or the reference's) to the fused pooling mode."""
    if isinstance(pooler, str):
        return _POOL[pooler]
    name = pooler.__name__ if isinstance(pooler, type) else type(pooler).__name__
    if name == "ClsPooler":
        return 0
    if name == "MeanPooler":
        return 1
    raise ValueError(f"no fused pooling for {pooler!r}; use encoder(**kwargs)[0] and pool in torch")


def pool_mode_or_none(pooler):
    """The fused pooling mode of `pooler`, or None when the kernels have none for it."""
    try:
        return _pool_mode(pooler)
    except (ValueError, KeyError):
        return None


SUPPORTED_MODEL_TYPES = ("bert", "distilbert", "roberta", "xlm-roberta", "camembert", "deberta-v2", "nomic_bert", "new")

# NomicBert names (transformers modeling_nomic_bert.py: layers.<l>.self_attn / mlp / post_*_layernorm) -> BERT names; the gate and
# up projections of the gated feed-forward become ONE tensor (canonical_state_dict)
_NOMIC = ((".self_attn.q_proj.", ".attention.self.query."), (".self_attn.k_proj.", ".attention.self.key."),
          (".self_attn.v_proj.", ".attention.self.value."), (".self_attn.o_proj.", ".attention.output.dense."),
          (".post_attention_layernorm.", ".attention.output.LayerNorm."), (".mlp.down_proj.", ".output.dense."),
          (".post_mlp_layernorm.", ".output.LayerNorm."))

# DistilBERT names -> BERT names (same post-LN block, no token types, no pooler)
_DISTIL = (("embeddings.word_embeddings.", "embeddings.word_embeddings."),
           ("embeddings.position_embeddings.", "embeddings.position_embeddings."),
           ("embeddings.LayerNorm.", "embeddings.LayerNorm."),
           (".attention.q_lin.", ".attention.self.query."), (".attention.k_lin.", ".attention.self.key."),
           (".attention.v_lin.", ".attention.self.value."), (".attention.out_lin.", ".attention.output.dense."),
           (".sa_layer_norm.", ".attention.output.LayerNorm."), (".ffn.lin1.", ".intermediate.dense."),
           (".ffn.lin2.", ".output.dense."), (".output_layer_norm.", ".output.LayerNorm."))


def _cfg_get(config):
    return (lambda k, dflt=None: config.get(k, dflt)) if isinstance(config, dict) else \
        (lambda k, dflt=None: getattr(config, k, dflt))


def _nomic_hub_config(get):
    """The configuration of the HUB form of NomicBert — what `AutoModel.from_pretrained("nomic-ai/nomic-embed-text-v1.5",
    trust_remote_code=True)` (reference models/retrievers/dense.py:16) builds: the checkpoint's auto_map selects the remote
    `NomicBertModel`, whose config class extends GPT2Config (n_embd / n_head / n_layer / n_inner / n_positions,
    activation_function "swiglu", rotary_emb_*, *_bias flags, prenorm, use_rms_norm) — onto the fields the kernels need.  The remote
    modelling file is not available offline; the field meanings follow the published config.json of that checkpoint and the weight
    mapping transformers itself ships for it (conversion_mapping.py "nomic_bert").  Everything the HIP forward pass does not
    compute is refused by name, so the caller falls back to the HF module instead of computing something else."""
    def need(*names):
        for n in names:
            v = get(n)
            if v is not None:
                return v
        raise ValueError(f"nomic_bert hub configuration without {' / '.join(names)}")
    act = str(get("activation_function", "swiglu"))
    if act != "swiglu":
        raise ValueError(f"nomic_bert activation_function {act!r} (swiglu only: the gated SiLU feed-forward)")
    if float(get("rotary_emb_fraction", 0.0) or 0.0) != 1.0:
        raise ValueError(f"nomic_bert rotary_emb_fraction {get('rotary_emb_fraction')!r} (1.0 only: every head dim rotates; 0 = a position table)")
    for flag in ("rotary_emb_interleaved", "prenorm", "use_rms_norm", "causal", "parallel_block"):
        if get(flag, False):
            raise ValueError(f"nomic_bert {flag} is set (the HIP forward pass is the post-LN, rotate-half, bidirectional block)")
    if get("rotary_emb_scale_base", None):
        raise ValueError("nomic_bert rotary_emb_scale_base is set (xpos scaling)")
    if get("moe_every_n_layers", 0) or get("num_experts", 0) and int(get("num_experts")) > 1:
        raise ValueError("nomic_bert mixture-of-experts feed-forward")
    max_pos = int(need("n_positions", "max_position_embeddings"))
    # (rotary_emb_base: the hub NomicBertConfig class default is 10000; nomic-embed-text-v1.5's config.json SETS 1000)
    if get("rotary_scaling_factor", None):
        # dynamic NTK scaling changes the rotary base only for sequences LONGER than max_trained_positions; up to there the plain
        # table holds, so the encoder simply does not accept longer sequences (bh_encoder_forward: "sequence too long")
        max_pos = min(max_pos, int(get("max_trained_positions", 2048) or 2048))
    return dict(hidden_size=int(need("n_embd", "hidden_size")), num_attention_heads=int(need("n_head", "num_attention_heads")),
                num_hidden_layers=int(need("n_layer", "num_hidden_layers")), intermediate_size=int(need("n_inner", "intermediate_size")),
                hidden_act="silu", type_vocab_size=int(get("type_vocab_size", 2) or 0) or 1,
                layer_norm_eps=float(need("layer_norm_epsilon", "layer_norm_eps")), position_offset=0,
                rotary_theta=float(get("rotary_emb_base", None) or 10000.0), ffn_gated=1, max_position_embeddings_override=max_pos)


def _nomic_hub_names(sd):
    """Hub-form NomicBert tensor names -> transformers' native names, the mapping transformers itself applies when it loads that
    checkpoint into its own class (conversion_mapping.py, "nomic_bert": encoder.layers -> layers, emb_ln -> embeddings.LayerNorm,
    attn.out_proj -> self_attn.o_proj, fc11 -> up_proj, fc12 -> gate_proj, fc2 -> down_proj, norm1 / norm2 -> post_attention /
    post_mlp layernorm, attn.Wqkv chunked along dim 0 into q | k | v).  A dict already in native form passes through."""
    if not any(".attn.Wqkv." in k or k.startswith("emb_ln.") or ".mlp.fc11." in k for k in sd):
        return sd
    out = {}
    for key, t in sd.items():
        if key.startswith("encoder.layers."):
            key = "layers." + key[len("encoder.layers."):]
        if key.startswith("emb_ln."):
            key = "embeddings.LayerNorm." + key[len("emb_ln."):]
        for a, b in ((".attn.out_proj.", ".self_attn.o_proj."), (".mlp.fc11.", ".mlp.up_proj."), (".mlp.fc12.", ".mlp.gate_proj."),
                     (".mlp.fc2.", ".mlp.down_proj."), (".norm1.", ".post_attention_layernorm."), (".norm2.", ".post_mlp_layernorm.")):
            if a in key:
                key = key.replace(a, b)
                break
        if ".attn.Wqkv." in key:
            if t.shape[0] % 3:
                raise ValueError(f"{key}: {tuple(t.shape)} is not q | k | v stacked along dim 0")
            for part, piece in zip(("q_proj", "k_proj", "v_proj"), torch.chunk(t, 3, dim=0)):
                out[key.replace(".attn.Wqkv.", f".self_attn.{part}.")] = piece
            continue
        if ".attn.rotary_emb." in key:
            continue  # (inv_freq buffers: a function of the base, rebuilt in the library)
        out[key] = t
    return out


def canonical_config(config):
    """HF config of a BERT-family model -> the fields the kernels need, under BERT's names.  Raises ValueError with the
    reason when the architecture is outside what the HIP forward pass computes."""
    get = _cfg_get(config)
    mt = get("model_type", "bert") or "bert"
    if mt not in SUPPORTED_MODEL_TYPES:
        raise ValueError(f"model_type {mt!r} (BERT / DistilBERT / RoBERTa-family encoders only)")
    if mt == "deberta-v2":
        # DeBERTa-v2 / v3 (the reference's default reranker, config/reranker/debertav3.yaml:3): the configuration the
        # deberta-v3-* checkpoints ship — disentangled attention with shared keys, log-bucketed relative positions, no
        # absolute positions, no token types, no convolution layer
        pat = get("pos_att_type") or []
        pat = sorted(x.strip() for x in (pat.split("|") if isinstance(pat, str) else pat))
        norm = [x.strip() for x in str(get("norm_rel_ebd", "none")).lower().split("|")]
        why = None
        if not get("relative_attention", False):
            why = "relative_attention is off"
        elif pat != ["c2p", "p2c"]:
            why = f"pos_att_type {pat} (c2p|p2c only)"
        elif not get("share_att_key", False):
            why = "share_att_key is off"
        elif "layer_norm" not in norm:
            why = f"norm_rel_ebd {norm} (layer_norm only)"
        elif int(get("position_buckets", -1) or -1) <= 0:
            why = "position_buckets <= 0"
        elif get("position_biased_input", True):
            why = "position_biased_input (absolute positions on top of the relative ones)"
        elif int(get("conv_kernel_size", 0) or 0) > 0:
            why = "conv_kernel_size > 0"
        elif int(get("embedding_size", get("hidden_size")) or get("hidden_size")) != int(get("hidden_size")):
            why = "embedding_size != hidden_size"
        elif int(get("type_vocab_size", 0) or 0) != 0:
            why = "type_vocab_size > 0"
        elif int(get("pooler_hidden_size", get("hidden_size")) or get("hidden_size")) != int(get("hidden_size")):
            why = "pooler_hidden_size != hidden_size"
        elif str(get("pooler_hidden_act", "gelu")) != "gelu":
            why = f"pooler_hidden_act {get('pooler_hidden_act')!r}"
        if why:
            raise ValueError(f"deberta-v2 configuration outside the HIP forward pass: {why}")
        max_rel = int(get("max_relative_positions", -1) or -1)
        c = dict(hidden_size=get("hidden_size"), num_attention_heads=get("num_attention_heads"),
                 num_hidden_layers=get("num_hidden_layers"), intermediate_size=get("intermediate_size"),
                 hidden_act=get("hidden_act", "gelu"), type_vocab_size=1, layer_norm_eps=get("layer_norm_eps", 1e-7),
                 position_offset=0, rel_span=int(get("position_buckets")),
                 max_relative_positions=max_rel if max_rel >= 1 else int(get("max_position_embeddings")))
    elif mt == "nomic_bert":
        # NomicBert (config/retriever/nomic-embed-text-v1.5.yaml; transformers modeling_nomic_bert.py): BERT's post-LN block with
        # rotary positions instead of a position table, bias-free projections and a gated SiLU feed-forward
        if get("n_embd") is not None or get("activation_function") is not None or get("rotary_emb_fraction") is not None:
            c = _nomic_hub_config(get)
        else:
            rope = get("rope_parameters") or {}
            rope_get = rope.get if isinstance(rope, dict) else (lambda k, dflt=None: getattr(rope, k, dflt))
            rtype = rope_get("rope_type", "default") or "default"
            if rtype != "default":
                raise ValueError(f"nomic_bert rope_type {rtype!r} (default only)")
            theta = float(rope_get("rope_theta", None) or get("rope_theta", None) or get("rotary_emb_base", None) or 1000.0)
            if str(get("hidden_act", "silu")) != "silu":
                raise ValueError(f"nomic_bert hidden_act {get('hidden_act')!r} (silu only)")
            for field in ("hidden_size", "num_attention_heads", "num_hidden_layers", "intermediate_size"):
                if get(field) is None:
                    raise ValueError(f"nomic_bert configuration without {field} (neither transformers' native form nor the hub's)")
            hd_cfg = get("head_dim", None)
            if hd_cfg not in (None, int(get("hidden_size")) // int(get("num_attention_heads"))):
                raise ValueError(f"nomic_bert head_dim {hd_cfg} != hidden_size / num_attention_heads")
            c = dict(hidden_size=get("hidden_size"), num_attention_heads=get("num_attention_heads"),
                     num_hidden_layers=get("num_hidden_layers"), intermediate_size=get("intermediate_size"),
                     hidden_act="silu", type_vocab_size=get("type_vocab_size", 2), layer_norm_eps=get("layer_norm_eps", 1e-12),
                     position_offset=0, rotary_theta=theta, ffn_gated=1)
    elif mt == "new":
        # Alibaba-NLP/gte-base-en-v1.5, gte-large-en-v1.5 (config/retriever/gte-*-en-v1.5.yaml): the REMOTE architecture "new"
        # (hub repository Alibaba-NLP/new-impl) — BERT's post-LN block with rotary positions (NTK-scaled), a packed biased q | k | v
        # projection and a GELU-gated feed-forward.  No copy of that modelling file exists offline: the mapping follows its published
        # description (oracle/new_oracle.py: parity unpinned) and every conversion is PROBED against the caller's own HF module
        # (`self_check`, BertEncoder.from_hf) — a mismatch keeps the run on that module, loudly.
        why = None
        if str(get("position_embedding_type", "rope")) != "rope":
            why = f"position_embedding_type {get('position_embedding_type')!r} (rope only)"
        elif str(get("layer_norm_type", "layer_norm")) != "layer_norm":
            why = f"layer_norm_type {get('layer_norm_type')!r}"
        elif get("logn_attention_scale", False):
            why = "logn_attention_scale is set"
        elif str(get("hidden_act", "gelu")) != "gelu":
            why = f"hidden_act {get('hidden_act')!r} (erf-GELU gate only)"
        scaling = get("rope_scaling", None)
        sget = (scaling.get if isinstance(scaling, dict) else (lambda k, dflt=None: getattr(scaling, k, dflt))) if scaling else None
        theta, scale = float(get("rope_theta", 10000.0) or 10000.0), 1.0
        hd_new = int(get("hidden_size")) // max(1, int(get("num_attention_heads")))
        if sget is not None and why is None:
            if str(sget("type", "")) != "ntk" or sget("mixed_b", None) is not None:
                why = f"rope_scaling {scaling!r} (type 'ntk' without mixed_b only)"
            else:
                # NTKScalingRotaryEmbedding builds its cos / sin cache once for max_position_embeddings * factor positions, i.e.
                # always in the scaled regime: base' = base * factor, inv_freq / factor^(2 / dim)
                factor = float(sget("factor", 1.0))
                theta, scale = theta * factor, factor ** (-2.0 / hd_new)
        if why:
            raise ValueError(f"'new' (gte) configuration outside the HIP forward pass: {why}")
        c = dict(hidden_size=get("hidden_size"), num_attention_heads=get("num_attention_heads"), num_hidden_layers=get("num_hidden_layers"),
                 intermediate_size=get("intermediate_size"), hidden_act="gelu", type_vocab_size=int(get("type_vocab_size", 0) or 0) or 1,
                 layer_norm_eps=get("layer_norm_eps", 1e-12), position_offset=0, rotary_theta=theta, rotary_scale=scale, ffn_gated=1,
                 self_check=True)
    elif mt == "distilbert":
        c = dict(hidden_size=get("dim"), num_attention_heads=get("n_heads"), num_hidden_layers=get("n_layers"),
                 intermediate_size=get("hidden_dim"), hidden_act=get("activation", "gelu"), type_vocab_size=1,
                 layer_norm_eps=1e-12, position_offset=0)
        if get("sinusoidal_pos_embds", False) not in (False, None):
            pass  # (the sinusoids are stored in the position table like learned ones)
    else:
        c = dict(hidden_size=get("hidden_size"), num_attention_heads=get("num_attention_heads"),
                 num_hidden_layers=get("num_hidden_layers"), intermediate_size=get("intermediate_size"),
                 hidden_act=get("hidden_act", "gelu"), type_vocab_size=get("type_vocab_size", 2),
                 layer_norm_eps=get("layer_norm_eps", 1e-12),
                 # RoBERTa numbers positions from padding_idx + 1 (modeling_roberta.create_position_ids_from_input_ids)
                 position_offset=0 if mt == "bert" else int(get("pad_token_id", 1)) + 1)
        pet = get("position_embedding_type", "absolute")
        if pet == "alibi" and mt == "bert":
            # jinaai/jina-embeddings-v2-* (config/retriever/jina-embeddings-v2-base-en.yaml): the REMOTE class JinaBertModel (hub
            # repository jinaai/jina-bert-implementation, model_type "bert") — BERT's post-LN block with symmetric ALiBi attention
            # biases instead of a position table and (feed_forward_type "geglu") a GELU-gated feed-forward without input bias.  Like
            # the "new" class above it cannot be pinned offline: mapped from its published description, probed against the caller's
            # module at conversion (self_check).
            fft = str(get("feed_forward_type", "original"))
            if fft not in ("original", "geglu"):
                raise ValueError(f"feed_forward_type {fft!r} (original / geglu)")
            c.update(alibi=1, self_check=True, ffn_gated=1 if fft == "geglu" else 0, jina=True)
        elif pet not in (None, "absolute"):
            raise ValueError(f"position_embedding_type {pet!r}")
    c.update(vocab_size=get("vocab_size"), max_position_embeddings=c.pop("max_position_embeddings_override", None) or get("max_position_embeddings"),
             model_type=mt)
    if not (c["hidden_act"] == "gelu" or (c["hidden_act"] == "silu" and c.get("ffn_gated"))):
        raise ValueError(f"hidden_act {c['hidden_act']!r} (erf-GELU, or SiLU as the gate of a gated feed-forward)")
    d, nh = int(c["hidden_size"]), int(c["num_attention_heads"])
    hd = d // max(1, nh)
    if d != nh * hd or hd % 8 != 0 or not 8 <= hd <= 64:
        raise ValueError(f"head dim {d / max(1, nh):g} (multiples of 8 up to 64)")
    if mt in ("deberta-v2", "nomic_bert", "new") and hd != 64:
        raise ValueError(f"{mt} with head dim {hd} (64 only)")
    if d % 64 != 0 or d > 2048 or nh * 64 > 2048:
        raise ValueError(f"hidden_size {d} with {nh} heads (multiple of 64, heads * 64 <= 2048)")
    c["head_dim"] = hd
    return c


def canonical_state_dict(cfg, state_dict):
    """Rename a BERT-family state dict to BertModel's names, drop what the forward pass does not use, pad heads
    narrower than 64 dims (see bh_encoder_config.head_dim).  Returns {name: tensor}."""
    mt, d, nh, hd = cfg["model_type"], int(cfg["hidden_size"]), int(cfg["num_attention_heads"]), int(cfg["head_dim"])
    out = {}
    if mt == "nomic_bert":
        stripped = {}
        for name, t in state_dict.items():
            for pre in ("nomic_bert.", "bert.", "model."):
                if name.startswith(pre):
                    name = name[len(pre):]
            stripped[name] = t
        state_dict = _nomic_hub_names(stripped)
    for name, t in state_dict.items():
        key = name
        for pre in ("bert.", "distilbert.", "roberta.", "deberta.", "nomic_bert.", "model."):
            if key.startswith(pre):
                key = key[len(pre):]
        if key.endswith("position_ids") or key.endswith("token_type_ids"):
            continue
        if mt == "deberta-v2":
            for a, b in ((".attention.self.query_proj.", ".attention.self.query."), (".attention.self.key_proj.", ".attention.self.key."),
                         (".attention.self.value_proj.", ".attention.self.value.")):
                if a in key:
                    key = key.replace(a, b)
                    break
            if key.startswith(("lm_predictions.", "mask_predictions.", "cls.")):
                continue
        elif mt == "nomic_bert":
            if key.endswith("inv_freq"):
                continue  # (the rotary frequencies are a function of theta: rebuilt in the library)
            if key.startswith("layers."):
                key = "encoder.layer." + key[len("layers."):]
            for a, b in _NOMIC:
                if a in key:
                    key = key.replace(a, b)
                    break
            if key.startswith(("cls.", "classifier.", "pooler.")):
                continue  # task heads of NomicBertFor*: the retriever reads the last hidden state (dense.py:40-44)
        elif mt == "new":
            if key.startswith("new."):
                key = key[len("new."):]
            if key.endswith("inv_freq") or key.startswith(("pooler.", "lm_head.", "classifier.")):
                continue  # (rotary frequencies are rebuilt in the library; the retriever reads the last hidden state, dense.py:40-44)
            for a, b in ((".attention.o_proj.", ".attention.output.dense."), (".attn_ln.", ".attention.output.LayerNorm."),
                         (".mlp.down_proj.", ".output.dense."), (".mlp_ln.", ".output.LayerNorm."),
                         (".attention.q_proj.", ".attention.self.query."), (".attention.k_proj.", ".attention.self.key."),
                         (".attention.v_proj.", ".attention.self.value.")):
                if a in key:
                    key = key.replace(a, b)
                    break
            if ".attention.qkv_proj." in key:  # pack_qkv: rows (or bias entries) [q | k | v]
                if t.shape[0] % 3:
                    raise ValueError(f"{key}: {tuple(t.shape)} is not q | k | v stacked along dim 0")
                for part, piece in zip(("query", "key", "value"), torch.chunk(t, 3, dim=0)):
                    out[key.replace(".attention.qkv_proj.", f".attention.self.{part}.")] = piece
                continue
        elif mt == "distilbert":
            if key.startswith("transformer.layer."):
                key = "encoder.layer." + key[len("transformer.layer."):]
            for a, b in _DISTIL:
                if a in key:
                    key = key.replace(a, b)
                    break
            # DistilBertForMaskedLM's head (modeling_distilbert.py: vocab_transform -> activation -> vocab_layer_norm ->
            # vocab_projector) is BertForMaskedLM's cls.predictions by other names — the doc model of the reference's
            # config/retriever/splade-efficient.yaml:3 reaches it through AutoModelForMaskedLM (models/retrievers/splade.py:17-19)
            for a, b in (("vocab_transform.", "cls.predictions.transform.dense."),
                         ("vocab_layer_norm.", "cls.predictions.transform.LayerNorm."),
                         ("vocab_projector.", "cls.predictions.decoder.")):
                if key.startswith(a):
                    key = b + key[len(a):]
                    break
            if key.startswith(("pre_classifier.", "classifier.")):
                # DistilBertForSequenceClassification pools with Linear + ReLU, not BertPooler's tanh: classify() must not exist
                continue
        elif mt == "bert" and cfg.get("jina"):
            for a, b in ((".mlp.wo.", ".output.dense."), (".mlp.layernorm.", ".output.LayerNorm.")):
                if a in key:
                    key = key.replace(a, b)
                    break
        elif mt != "bert":
            # RobertaClassificationHead = dense + tanh + out_proj on the <s> token: BertPooler + classifier by another name
            if key.startswith("classifier.dense."):
                key = "pooler.dense." + key[len("classifier.dense."):]
            elif key.startswith("classifier.out_proj."):
                key = "classifier." + key[len("classifier.out_proj."):]
            elif key.startswith("lm_head."):
                # RobertaLMHead (dense -> gelu -> layer_norm -> decoder, + bias) = BertForMaskedLM's cls.predictions renamed
                for a, b in (("lm_head.dense.", "cls.predictions.transform.dense."),
                             ("lm_head.layer_norm.", "cls.predictions.transform.LayerNorm."),
                             ("lm_head.decoder.", "cls.predictions.decoder.")):
                    if key.startswith(a):
                        key = b + key[len(a):]
                        break
                if key == "lm_head.bias":
                    key = "cls.predictions.bias"
                if key.startswith("lm_head."):
                    continue
        out[key] = t
    if mt == "distilbert" or "embeddings.token_type_embeddings.weight" not in out:
        out["embeddings.token_type_embeddings.weight"] = torch.zeros(int(cfg["type_vocab_size"]), d, dtype=torch.float16)
    if mt == "nomic_bert":
        # no position table (rotary), no biases: zeros where the BERT-shaped stack expects them; gate and up rows as one tensor
        nl, f = int(cfg["num_hidden_layers"]), int(cfg["intermediate_size"])
        out["embeddings.position_embeddings.weight"] = torch.zeros(int(cfg["max_position_embeddings"]), d, dtype=torch.float16)
        for l in range(nl):
            pre = f"encoder.layer.{l}."
            gate, up = out.pop(pre + "mlp.gate_proj.weight", None), out.pop(pre + "mlp.up_proj.weight", None)
            if gate is None or up is None:
                raise ValueError(f"nomic_bert state dict lacks layers.{l}.mlp.gate_proj / up_proj")
            # rows interleaved (gate j, up j): one GEMM then yields (gate, up) column pairs, folded in its epilogue
            out[pre + "intermediate.dense.weight"] = torch.stack([gate.detach().float(), up.detach().float()], dim=1).reshape(2 * f, d)
            gb, ub = out.pop(pre + "mlp.gate_proj.bias", None), out.pop(pre + "mlp.up_proj.bias", None)
            if gb is not None or ub is not None:  # (mlp_fc1_bias checkpoints: interleaved like the rows they belong to)
                gb = torch.zeros(f) if gb is None else gb.detach().float()
                ub = torch.zeros(f) if ub is None else ub.detach().float()
                out[pre + "intermediate.dense.bias"] = torch.stack([gb, ub], dim=1).reshape(2 * f)
            for name, n in (("attention.self.query", d), ("attention.self.key", d), ("attention.self.value", d),
                            ("attention.output.dense", d), ("intermediate.dense", 2 * f), ("output.dense", d)):
                out.setdefault(pre + name + ".bias", torch.zeros(n, dtype=torch.float16))
    if cfg.get("jina"):
        # ALiBi: a zero position table; gated_layers' rows are [gated | non-gated] (JinaBertGLUMLP: act(h[:, :f]) * h[:, f:]): interleaved
        # (gate j, up j); it has no bias
        nl, f = int(cfg["num_hidden_layers"]), int(cfg["intermediate_size"])
        out["embeddings.position_embeddings.weight"] = torch.zeros(int(cfg["max_position_embeddings"]), d, dtype=torch.float16)
        for l in range(nl if cfg.get("ffn_gated") else 0):
            pre = f"encoder.layer.{l}."
            gl = out.pop(pre + "mlp.gated_layers.weight", None)
            if gl is None or gl.shape[0] != 2 * f:
                raise ValueError(f"jina state dict lacks encoder.layer.{l}.mlp.gated_layers.weight of {2 * f} rows")
            out[pre + "intermediate.dense.weight"] = torch.stack([gl[:f].detach().float(), gl[f:].detach().float()], dim=1).reshape(2 * f, d)
            out[pre + "intermediate.dense.bias"] = torch.zeros(2 * f, dtype=torch.float16)
    if mt == "new":
        # rotary positions: a zero position table; up_gate_proj's rows are [up | gate] (NewGatedMLP splits the output in that order):
        # interleaved (gate j, up j) like NomicBert's; it has no bias
        nl, f = int(cfg["num_hidden_layers"]), int(cfg["intermediate_size"])
        out["embeddings.position_embeddings.weight"] = torch.zeros(int(cfg["max_position_embeddings"]), d, dtype=torch.float16)
        for l in range(nl):
            pre = f"encoder.layer.{l}."
            ug = out.pop(pre + "mlp.up_gate_proj.weight", None)
            if ug is None or ug.shape[0] != 2 * f:
                raise ValueError(f"'new' state dict lacks encoder.layer.{l}.mlp.up_gate_proj.weight of {2 * f} rows")
            up, gate = ug[:f].detach().float(), ug[f:].detach().float()
            out[pre + "intermediate.dense.weight"] = torch.stack([gate, up], dim=1).reshape(2 * f, d)
            ugb = out.pop(pre + "mlp.up_gate_proj.bias", None)
            if ugb is not None:
                out[pre + "intermediate.dense.bias"] = torch.stack([ugb[f:].detach().float(), ugb[:f].detach().float()], dim=1).reshape(2 * f)
            else:
                out[pre + "intermediate.dense.bias"] = torch.zeros(2 * f, dtype=torch.float16)
            for name, n in (("attention.self.query", d), ("attention.self.key", d), ("attention.self.value", d),
                            ("attention.output.dense", d), ("output.dense", d)):
                out.setdefault(pre + name + ".bias", torch.zeros(n, dtype=torch.float16))
    if mt == "deberta-v2":  # word embeddings only (position_biased_input = False): a zero position table
        out["embeddings.position_embeddings.weight"] = torch.zeros(int(cfg["max_position_embeddings"]), d, dtype=torch.float16)
        if "encoder.rel_embeddings.weight" in out:  # the attention uses the first 2 * position_buckets rows
            out["encoder.rel_embeddings.weight"] = out["encoder.rel_embeddings.weight"][:2 * int(cfg["rel_span"])]
    if hd != 64:
        scale = (64.0 / hd) ** 0.5
        for key in list(out):
            t = out[key]
            if ".attention.self." in key and key.endswith(".weight"):      # [nh*hd, d] -> [nh*64, d]
                w = t.detach().float().reshape(nh, hd, d)
                if ".query." in key:
                    w = w * scale
                p = torch.zeros(nh, 64, d)
                p[:, :hd] = w
                out[key] = p.reshape(nh * 64, d)
            elif ".attention.self." in key and key.endswith(".bias"):      # [nh*hd] -> [nh*64]
                bvec = t.detach().float().reshape(nh, hd)
                if ".query." in key:
                    bvec = bvec * scale
                p = torch.zeros(nh, 64)
                p[:, :hd] = bvec
                out[key] = p.reshape(nh * 64)
            elif key.endswith("attention.output.dense.weight"):            # [d, nh*hd] -> [d, nh*64]
                w = t.detach().float().reshape(d, nh, hd)
                p = torch.zeros(d, nh, 64)
                p[:, :, :hd] = w
                out[key] = p.reshape(d, nh * 64)
    return out


class BertEncoder:
    def __init__(self, config, state_dict, device=0):
        """config: HF config of a BERT-family encoder (BertConfig, DistilBertConfig, RobertaConfig / XLMRobertaConfig; or a
        dict with the same field names); state_dict: the matching HF state dict (task-model prefixes tolerated)."""
        canon = canonical_config(config)
        state_dict = canonical_state_dict(canon, state_dict)
        get = canon.get
        self._h = None
        _lib.init(device)
        self.device_index = device
        self.config = config
        self.hidden_size = int(get("hidden_size"))
        self.vocab_size = int(get("vocab_size"))
        self.has_mlm_head = False
        cfg = _lib.bh_encoder_config(
            n_layers=int(get("num_hidden_layers")), hidden=self.hidden_size, n_heads=int(get("num_attention_heads")),
            intermediate=int(get("intermediate_size")), vocab_size=int(get("vocab_size")),
            max_position=int(get("max_position_embeddings")), type_vocab_size=int(get("type_vocab_size")),
            activation=1 if (get("ffn_gated") and get("hidden_act") == "silu") else 0, ln_eps=float(get("layer_norm_eps", 1e-12)),
            head_dim=int(get("head_dim")), position_offset=int(get("position_offset", 0)),
            rotary_theta=float(get("rotary_theta", 0.0) or 0.0), ffn_gated=int(get("ffn_gated", 0) or 0),
            rotary_scale=float(get("rotary_scale", 0.0) or 0.0), alibi=int(get("alibi", 0) or 0))
        self.needs_self_check = bool(get("self_check", False))
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().bh_encoder_create(ctypes.byref(h), ctypes.byref(cfg)))
        self._h = h
        self.disentangled = canon.get("model_type") == "deberta-v2"
        if self.disentangled:
            # DeBERTa-v2 / v3: relative-position tensors, the ContextPooler's GELU, and the index table t(delta) — the log
            # bucket of transformers' make_log_bucket_position (modeling_deberta_v2.py:57-69), evaluated with the same torch
            # float32 operations (its ceil() sits on float32 logarithms: another libm could move a bucket boundary)
            span, max_rel, L = int(canon["rel_span"]), int(canon["max_relative_positions"]), int(get("max_position_embeddings"))
            _lib.check(_lib.lib().bh_encoder_set_option(self._h, b"rel_attention_span", span))
            _lib.check(_lib.lib().bh_encoder_set_option(self._h, b"cls_activation", 1))
            rel = torch.arange(-(L - 1), L, dtype=torch.long)
            mid = span // 2
            abs_pos = torch.where((rel < mid) & (rel > -mid), torch.tensor(mid - 1).type_as(rel), torch.abs(rel))
            log_pos = torch.ceil(torch.log(abs_pos / mid) / torch.log(torch.tensor((max_rel - 1) / mid)) * (mid - 1)) + mid
            bucket = torch.where(abs_pos <= mid, rel.type_as(log_pos), log_pos * torch.sign(rel)).to(torch.long)
            table = torch.clamp(bucket + span, 0, 2 * span - 1).to(torch.int32).contiguous()
            _lib.check(_lib.lib().bh_encoder_set_rel_index(self._h, ctypes.c_void_p(table.data_ptr()), int(table.numel())))
        n_set = 0
        classifier = {}
        for name, t in state_dict.items():
            key = name
            for pre in ("bert.", "nomic_bert.", "model."):
                if key.startswith(pre):
                    key = key[len(pre):]
            if key.endswith("position_ids") or key.endswith("token_type_ids"):
                continue
            if key.startswith("classifier."):
                classifier[key] = t
                continue
            if key.startswith("pooler."):
                classifier[key] = t  # BertPooler only matters under a classifier (the dense path reads outputs[0], dense.py:40-44)
                continue
            if key.startswith("cls.predictions."):
                self.has_mlm_head = True  # BertForMaskedLM checkpoint: SPLADE pooling available (encode_splade)
            elif not (key.startswith("embeddings.") or key.startswith("encoder.layer.") or
                      (self.disentangled and key in ("encoder.rel_embeddings.weight", "encoder.LayerNorm.weight", "encoder.LayerNorm.bias"))):
                continue
            a = t.detach().to("cpu")
            if a.dtype not in (torch.float16, torch.float32):
                a = a.float()
            a = a.contiguous()
            code = _lib.BH_F16 if a.dtype == torch.float16 else _lib.BH_F32
            _lib.check(_lib.lib().bh_encoder_set_tensor(self._h, key.encode(), ctypes.c_void_p(a.data_ptr()), code,
                                                        a.numel()))
            n_set += 1
        self.num_labels = 0
        if "classifier.weight" in classifier:  # BertForSequenceClassification checkpoint: classify() available
            for key, t in classifier.items():
                a = t.detach().to("cpu")
                if a.dtype not in (torch.float16, torch.float32):
                    a = a.float()
                a = a.contiguous()
                code = _lib.BH_F16 if a.dtype == torch.float16 else _lib.BH_F32
                _lib.check(_lib.lib().bh_encoder_set_tensor(self._h, key.encode(), ctypes.c_void_p(a.data_ptr()), code,
                                                            a.numel()))
                n_set += 1
            self.num_labels = int(classifier["classifier.weight"].shape[0])
        _lib.check(_lib.lib().bh_encoder_commit(self._h))
        self.n_tensors = n_set

    @classmethod
    def from_hf(cls, model, device=0):
        """Build from an instantiated HF BertModel (weights are copied into HBM once).  An architecture whose mapping could not be
        pinned offline (`needs_self_check`: the remote "new" class of gte-*-en-v1.5) is PROBED against `model` itself before it is
        handed out: ValueError when the two disagree (the caller keeps the HF module)."""
        enc = cls(model.config, model.state_dict(), device=device)
        if enc.needs_self_check:
            try:
                enc.self_check(model)
            except Exception:
                enc.close()
                raise
        return enc

    @torch.no_grad()
    def self_check(self, model, batch=4, seq_len=24, seed=1234):
        """A probe batch through the HF module `model` (wherever it lives, in its own dtype) and through this encoder: the last hidden
        states of the attended tokens must agree — cosine >= 0.995 per sequence and max |diff| <= 5e-2 x max |reference| (two fp16
        forward passes of the same weights).  Raises ValueError with the figures otherwise.  Used where the name / arithmetic mapping
        of an architecture follows a description instead of a pinned source (oracle/new_oracle.py)."""
        g = torch.Generator().manual_seed(seed)
        T = int(min(seq_len, int(getattr(self.config, "max_position_embeddings", seq_len) if not isinstance(self.config, dict)
                                 else self.config.get("max_position_embeddings", seq_len))))
        ids = torch.randint(5, max(6, self.vocab_size - 1), (batch, T), generator=g)
        lens = torch.tensor([T] + [max(2, T - 5 * (i + 1)) for i in range(batch - 1)])
        mask = (torch.arange(T)[None, :] < lens[:, None]).long()
        ids = ids * mask
        try:
            p = next(model.parameters())
            dev, was_training = p.device, model.training
        except StopIteration:
            dev, was_training = torch.device("cpu"), False
        model.eval()
        try:
            try:
                want = model(input_ids=ids.to(dev), attention_mask=mask.to(dev))[0].float().cpu()
            except Exception:  # noqa: BLE001 — e.g. an fp16 module still on the CPU (Dense loads with torch_dtype=float16): probe it on the GPU
                if dev.type == "cuda" or not torch.cuda.is_available():
                    raise
                gpu = torch.device("cuda", self.device_index)
                model.to(gpu)
                try:
                    want = model(input_ids=ids.to(gpu), attention_mask=mask.to(gpu))[0].float().cpu()
                finally:
                    model.to(dev)
        except Exception as exc:  # noqa: BLE001 — a module that cannot be probed cannot vouch for the mapping: stay on it
            raise ValueError(f"self-check could not run the HF module on the probe batch: {type(exc).__name__}: {exc}") from exc
        finally:
            if was_training:
                model.train()
        got = self(input_ids=ids, attention_mask=mask)[0].float().cpu()
        worst_cos, worst_abs = 1.0, 0.0
        for b in range(batch):
            w, h = want[b, :int(lens[b])].reshape(-1), got[b, :int(lens[b])].reshape(-1)
            cos = float(torch.dot(w, h) / (w.norm() * h.norm() + 1e-30))
            worst_cos = min(worst_cos, cos)
            worst_abs = max(worst_abs, float((w - h).abs().max() / (w.abs().max() + 1e-30)))
        self.self_check_result = {"min_cosine": worst_cos, "max_rel_abs_diff": worst_abs}
        if not (worst_cos >= 0.995 and worst_abs <= 5e-2):
            raise ValueError(f"self-check against the HF module failed: min cosine {worst_cos:.5f}, max |diff| / max |ref| {worst_abs:.4f} "
                             f"(the HIP mapping of this architecture does not reproduce the module's forward pass)")
        return self.self_check_result

    @staticmethod
    def unsupported_reason(model):
        """None when the HIP forward pass covers this HF model's architecture, else a short reason."""
        cfg = getattr(model, "config", None)
        if cfg is None:
            return "no .config"
        try:
            canonical_config(cfg)
        except (ValueError, TypeError) as exc:
            return str(exc)
        return None

    @staticmethod
    def supports(model):
        return BertEncoder.unsupported_reason(model) is None

    # -- nn.Module-like surface used by Retrieve / Dense -------------------------------------------
    def to(self, *args, **kwargs):
        return self

    def eval(self):
        return self

    def half(self):
        return self

    def set_option(self, name, value):
        _lib.check(_lib.lib().bh_encoder_set_option(self._h, name.encode(), int(value)))

    # -- forward -----------------------------------------------------------------------------------
    def _forward(self, input_ids, attention_mask, token_type_ids, pool, l2_normalize=False):
        ids = torch.as_tensor(input_ids).to("cpu", torch.int64).contiguous()
        if ids.ndim != 2:
            raise ValueError(f"input_ids must be [B, T], got {tuple(ids.shape)}")
        B, T = ids.shape
        keep = [ids]
        if self.disentangled:
            token_type_ids = None  # DeBERTa-v3 has no token-type embeddings (type_vocab_size = 0): HF ignores the ids too

        def host(x):
            if x is None:
                return None
            x = torch.as_tensor(x).to("cpu", torch.int64).contiguous()
            if tuple(x.shape) != (B, T):
                raise ValueError(f"expected [{B}, {T}], got {tuple(x.shape)}")
            keep.append(x)
            return ctypes.c_void_p(x.data_ptr())

        mask_p, type_p = host(attention_mask), host(token_type_ids)
        dev = torch.device("cuda", self.device_index)
        shape = (B, T, self.hidden_size) if pool == 2 else (B, self.vocab_size) if pool == 3 else \
            (B, self.num_labels) if pool == 4 else (B, self.hidden_size)
        out = torch.empty(shape, dtype=torch.float32 if pool == 4 else torch.float16, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # the library runs on its own stream
        _lib.init(self.device_index)
        _lib.check(_lib.lib().bh_encoder_forward(self._h, ctypes.c_void_p(ids.data_ptr()), mask_p, type_p, B, T, pool,
                                                 int(bool(l2_normalize)), ctypes.c_void_p(out.data_ptr()), 1))
        del keep
        return out

    def __call__(self, input_ids=None, attention_mask=None, token_type_ids=None, **unused):
        """HF-style call: returns (last_hidden_state [B, T, d] fp16,)."""
        return (self._forward(input_ids, attention_mask, token_type_ids, 2),)

    def encode_pooled(self, kwargs, pooler, l2_normalize=False):
        """Fused forward + pooling: [B, d] fp16 on the device (reference dense.py:40-46 in one call)."""
        return self._forward(kwargs["input_ids"], kwargs.get("attention_mask"), kwargs.get("token_type_ids"),
                             _pool_mode(pooler), l2_normalize)

    def encode_splade(self, kwargs):
        """Fused forward + masked-LM head + SPLADE pooling: [B, vocab] fp16 = max_t log(1 + relu(logits)) over the
        attended tokens (reference models/retrievers/splade.py:36-43).  The [B, T, vocab] logits are never materialised."""
        if not self.has_mlm_head:
            raise RuntimeError("this encoder was built without cls.predictions.* weights (not a BertForMaskedLM checkpoint)")
        return self._forward(kwargs["input_ids"], kwargs.get("attention_mask"), kwargs.get("token_type_ids"), 3)

    def classify(self, kwargs):
        """Forward + BertPooler + classifier: [B, num_labels] fp32 logits on the device (the `.logits` of HF
        BertForSequenceClassification; reference models/rerankers/crossencoder.py:34-38)."""
        if not self.num_labels:
            raise RuntimeError("this encoder was built without classifier.* weights (not a sequence-classification checkpoint)")
        return self._forward(kwargs["input_ids"], kwargs.get("attention_mask"), kwargs.get("token_type_ids"), 4)

    def counters(self):
        c = _lib.bh_encoder_counters()
        _lib.check(_lib.lib().bh_encoder_counters_get(self._h, ctypes.byref(c)))
        return {name: getattr(c, name) for name, _ in c._fields_}

    def close(self):
        if self._h is not None:
            _lib.lib().bh_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- op-level wrappers (device tensors) used by the parity tests and micro-benchmarks --------------------

def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def gemm_f16(a, w, bias=None, bias_mode=1, residual=None, gelu=False, variant=0, out=None, repeats=1):
    """out[M, N] = a[M, K] @ w[N, K]^T (+bias) (+residual) (GELU) on the HIP kernel.  Returns (out, avg_ms).
    gelu="swiglu": w's rows are (gate, up) pairs and out is [M, N / 2] = silu(gate) * up (the gated fold of the persistent kernel)."""
    assert a.is_cuda and w.is_cuda and a.dtype == torch.float16 and w.dtype == torch.float16
    M, K = a.shape
    N = w.shape[0]
    fold = gelu in ("swiglu", "geglu")  # gated fold, SiLU or erf-GELU gate
    gelu = (2 if gelu == "swiglu" else 3) if fold else int(bool(gelu))
    if out is None:
        out = torch.empty((M, N // 2 if fold else N), dtype=torch.float16, device=a.device)
    ms = ctypes.c_float(0)
    torch.cuda.synchronize(a.device)
    _lib.check(_lib.lib().bh_op_gemm_f16(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), _p(bias),
                                         bias_mode if bias is not None else 0, _p(residual),
                                         residual.stride(0) if residual is not None else 0, M, N, K, int(gelu),
                                         variant, repeats, ctypes.byref(ms)))
    return out, float(ms.value)


def attention(qk, vt, seq_off, seq_len, n_heads, max_len):
    """Packed varlen attention: qk [tokens, 2d], vt [d, ld] -> ctx [tokens, d] (rows outside sequences zero)."""
    assert qk.is_cuda and vt.is_cuda
    d = n_heads * 64
    ctx = torch.zeros((qk.shape[0], d), dtype=torch.float16, device=qk.device)
    so = torch.as_tensor(seq_off, dtype=torch.int64, device=qk.device)
    sl = torch.as_tensor(seq_len, dtype=torch.int32, device=qk.device)
    torch.cuda.synchronize(qk.device)
    _lib.check(_lib.lib().bh_op_attention(_p(qk), qk.stride(0), _p(vt), vt.stride(0), _p(ctx), ctx.stride(0), _p(so),
                                          _p(sl), int(so.numel()), n_heads, int(max_len)))
    return ctx


def layernorm(x, gamma, beta, eps):
    out = torch.empty_like(x)
    torch.cuda.synchronize(x.device)
    _lib.check(_lib.lib().bh_op_layernorm(_p(x), _p(out), x.shape[0], x.shape[1], float(eps), _p(gamma), _p(beta)))
    return out


def rotary(qk, pos, n_heads, theta, max_pos=None):
    """Rotary positions IN PLACE on packed [Q | K] rows: qk [rows, 2 * n_heads * 64] fp16, pos [rows] int32 (device tensors)."""
    assert qk.is_cuda and pos.is_cuda and qk.dtype == torch.float16 and pos.dtype == torch.int32 and qk.is_contiguous()
    assert qk.shape[1] == 2 * n_heads * 64 and pos.shape[0] == qk.shape[0]
    torch.cuda.synchronize(qk.device)
    _lib.check(_lib.lib().bh_op_rotary(_p(qk), qk.shape[0], n_heads, _p(pos), float(theta),
                                       int(max_pos if max_pos is not None else int(pos.max().item()) + 1)))
    return qk


def swiglu(gu):
    """silu(gate) * up over gu [rows, 2 f] fp16 ((gate, up) column pairs: 2 j, 2 j + 1) -> [rows, f] fp16."""
    assert gu.is_cuda and gu.dtype == torch.float16 and gu.is_contiguous() and gu.shape[1] % 16 == 0
    out = torch.empty((gu.shape[0], gu.shape[1] // 2), dtype=torch.float16, device=gu.device)
    torch.cuda.synchronize(gu.device)
    _lib.check(_lib.lib().bh_op_swiglu(_p(gu), _p(out), gu.shape[0], gu.shape[1] // 2))
    return out


def gated_act(gu, act="silu"):
    """act(gate) * up over gu [rows, 2 f] fp16 ((gate, up) column pairs) -> [rows, f] fp16; act "silu" (= swiglu) or "gelu" (erf)."""
    assert gu.is_cuda and gu.dtype == torch.float16 and gu.is_contiguous() and gu.shape[1] % 16 == 0
    out = torch.empty((gu.shape[0], gu.shape[1] // 2), dtype=torch.float16, device=gu.device)
    torch.cuda.synchronize(gu.device)
    _lib.check(_lib.lib().bh_op_gated_act(_p(gu), _p(out), gu.shape[0], gu.shape[1] // 2, {"silu": 0, "gelu": 1}[act]))
    return out


def permlane_mode():
    return int(_lib.lib().bh_gemm_permlane_mode())
