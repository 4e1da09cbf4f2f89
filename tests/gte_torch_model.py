"""A torch implementation of the "new" encoder architecture (Alibaba-NLP/new-impl: gte-base-en-v1.5 / gte-large-en-v1.5), written
from the published description of that REMOTE modelling file, which is not available offline (see oracle/new_oracle.py: parity
unpinned).  Test infrastructure: it plays the part of the HF module a user's `AutoModel.from_pretrained(..., trust_remote_code=True)`
returns — same config fields, same tensor names, same call convention (`model(input_ids=..., attention_mask=...)[0]`) — for the
conversion and self-check tests, and is the second, independently written restatement the numpy oracle is compared with."""
import math
from types import SimpleNamespace

import torch
from torch import nn


def new_config(**kw):
    cfg = dict(model_type="new", vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
               hidden_act="gelu", max_position_embeddings=128, type_vocab_size=0, layer_norm_type="layer_norm", layer_norm_eps=1e-12,
               position_embedding_type="rope", rope_theta=500000.0, rope_scaling={"type": "ntk", "factor": 2.0}, pack_qkv=True,
               unpad_inputs=False, use_memory_efficient_attention=False, logn_attention_scale=False, logn_attention_clip1=False,
               _name_or_path="toy/gte-tiny")
    cfg.update(kw)
    return SimpleNamespace(**cfg)


class _Rotary(nn.Module):
    """cos / sin of RotaryEmbedding (factor None) or NTKScalingRotaryEmbedding with mixed_b None, whose cache is built for max_pos * factor
    positions, i.e. always in the scaled regime.  No buffers: the frequencies are recomputed in forward (nothing for a loader to mishandle)."""

    def __init__(self, dim, max_pos, base, factor):
        super().__init__()
        self.dim, self.base, self.factor = int(dim), float(base), (None if factor is None else float(factor))

    def forward(self, seq_len):
        base = self.base * self.factor if self.factor is not None else self.base
        inv_freq = 1.0 / (base ** (torch.arange(0, self.dim, 2).float() / self.dim))
        if self.factor is not None:
            inv_freq = inv_freq / self.factor ** (2 / self.dim)
        t = torch.arange(seq_len, dtype=torch.float32)
        freqs = torch.outer(t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos(), emb.sin()


def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size)
        if c.type_vocab_size > 0:
            self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        sc = c.rope_scaling
        self.rotary_emb = _Rotary(c.hidden_size // c.num_attention_heads, c.max_position_embeddings, c.rope_theta,
                                  None if sc is None else sc["factor"])


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.qkv_proj = nn.Linear(c.hidden_size, 3 * c.hidden_size, bias=True)
        self.o_proj = nn.Linear(c.hidden_size, c.hidden_size, bias=True)


class _MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.up_gate_proj = nn.Linear(c.hidden_size, 2 * c.intermediate_size, bias=False)
        self.down_proj = nn.Linear(c.intermediate_size, c.hidden_size, bias=True)


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _Attention(c)
        self.mlp = _MLP(c)
        self.attn_ln = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.mlp_ln = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class TorchNewModel(nn.Module):
    def __init__(self, config, seed=0, break_it=None):
        super().__init__()
        self.config = config
        self.break_it = break_it  # tests: "swap_gate" computes act(up) * gate — a module the HIP mapping must NOT reproduce
        torch.manual_seed(seed)
        self.embeddings = _Embeddings(config)
        self.encoder = _Encoder(config)
        with torch.no_grad():
            g = torch.Generator().manual_seed(seed + 1)
            for name, p in self.named_parameters():
                if "LayerNorm.weight" in name or name.endswith("_ln.weight"):
                    p.copy_(1.0 + 0.05 * torch.randn(p.shape, generator=g))
                elif "word_embeddings" in name or "token_type" in name:
                    p.copy_(0.5 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
                p.copy_(p.half().float())

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, **unused):
        c = self.config
        B, T = input_ids.shape
        dt = self.embeddings.word_embeddings.weight.dtype
        x = self.embeddings.word_embeddings(input_ids)
        if c.type_vocab_size > 0:
            x = x + self.embeddings.token_type_embeddings(torch.zeros_like(input_ids) if token_type_ids is None else token_type_ids)
        x = self.embeddings.LayerNorm(x)
        cos, sin = self.embeddings.rotary_emb(T)
        cos, sin = cos.to(x.device, dt)[None, None], sin.to(x.device, dt)[None, None]
        nh = c.num_attention_heads
        dh = c.hidden_size // nh
        bias = torch.zeros(B, 1, 1, T, dtype=dt, device=x.device)
        if attention_mask is not None:
            bias = bias.masked_fill(attention_mask[:, None, None, :] == 0, torch.finfo(dt).min)
        for layer in self.encoder.layer:
            q, k, v = layer.attention.qkv_proj(x).split(c.hidden_size, dim=-1)
            q, k, v = (t.view(B, T, nh, dh).transpose(1, 2) for t in (q, k, v))
            q = q * cos + _rotate_half(q) * sin
            k = k * cos + _rotate_half(k) * sin
            s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + bias
            p = torch.softmax(s, dim=-1)
            ctx = torch.matmul(p, v).transpose(1, 2).reshape(B, T, c.hidden_size)
            x = layer.attn_ln(x + layer.attention.o_proj(ctx))
            up, gate = layer.mlp.up_gate_proj(x).split(c.intermediate_size, dim=-1)
            h = (torch.nn.functional.gelu(up) * gate) if self.break_it == "swap_gate" else (torch.nn.functional.gelu(gate) * up)
            x = layer.mlp_ln(x + layer.mlp.down_proj(h))
        return (x,)


# ---- JinaBert (jina-embeddings-v2): BERT attention + symmetric ALiBi bias, GELU-gated feed-forward; same role as TorchNewModel ----------

def jina_config(**kw):
    cfg = dict(model_type="bert", vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
               hidden_act="gelu", max_position_embeddings=128, type_vocab_size=2, layer_norm_eps=1e-12, position_embedding_type="alibi",
               feed_forward_type="geglu", emb_pooler="mean", _name_or_path="toy/jina-tiny")
    cfg.update(kw)
    return SimpleNamespace(**cfg)


def _alibi_head_slopes(n_heads):
    def get_slopes_power_of_2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        ratio = start
        return [start * ratio ** i for i in range(n)]
    if math.log2(n_heads).is_integer():
        return get_slopes_power_of_2(n_heads)
    closest_power_of_2 = 2 ** math.floor(math.log2(n_heads))
    return get_slopes_power_of_2(closest_power_of_2) + _alibi_head_slopes(2 * closest_power_of_2)[0::2][: n_heads - closest_power_of_2]


class TorchJinaBert(nn.Module):
    def __init__(self, config, seed=0):
        super().__init__()
        c = self.config = config
        torch.manual_seed(seed)
        self.embeddings = nn.Module()
        self.embeddings.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size)
        self.embeddings.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.embeddings.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.encoder = nn.Module()
        layers = []
        for _ in range(c.num_hidden_layers):
            L = nn.Module()
            L.attention = nn.Module()
            L.attention.self = nn.Module()
            L.attention.self.query = nn.Linear(c.hidden_size, c.hidden_size)
            L.attention.self.key = nn.Linear(c.hidden_size, c.hidden_size)
            L.attention.self.value = nn.Linear(c.hidden_size, c.hidden_size)
            L.attention.output = nn.Module()
            L.attention.output.dense = nn.Linear(c.hidden_size, c.hidden_size)
            L.attention.output.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
            L.mlp = nn.Module()
            L.mlp.gated_layers = nn.Linear(c.hidden_size, 2 * c.intermediate_size, bias=False)
            L.mlp.wo = nn.Linear(c.intermediate_size, c.hidden_size)
            L.mlp.layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
            layers.append(L)
        self.encoder.layer = nn.ModuleList(layers)
        with torch.no_grad():
            g = torch.Generator().manual_seed(seed + 1)
            for name, p in self.named_parameters():
                if "LayerNorm.weight" in name or name.endswith("layernorm.weight"):
                    p.copy_(1.0 + 0.05 * torch.randn(p.shape, generator=g))
                elif "embeddings" in name and "LayerNorm" not in name:
                    p.copy_(0.5 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
                p.copy_(p.half().float())

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, **unused):
        c = self.config
        B, T = input_ids.shape
        dt = self.embeddings.word_embeddings.weight.dtype
        tt = torch.zeros_like(input_ids) if token_type_ids is None else token_type_ids
        x = self.embeddings.LayerNorm(self.embeddings.word_embeddings(input_ids) + self.embeddings.token_type_embeddings(tt))
        nh = c.num_attention_heads
        dh = c.hidden_size // nh
        pos = torch.arange(T, device=x.device)
        rel = (pos[None, :] - pos[:, None]).abs().to(dt)
        alibi = -torch.tensor(_alibi_head_slopes(nh), dtype=dt, device=x.device)[:, None, None] * rel[None]
        bias = alibi[None].expand(B, -1, -1, -1).clone()
        if attention_mask is not None:
            bias = bias + torch.zeros(B, 1, 1, T, dtype=dt, device=x.device).masked_fill(attention_mask[:, None, None, :] == 0, torch.finfo(dt).min)
        for L in self.encoder.layer:
            q, k, v = (m(x).view(B, T, nh, dh).transpose(1, 2) for m in (L.attention.self.query, L.attention.self.key, L.attention.self.value))
            p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + bias, dim=-1)
            ctx = torch.matmul(p, v).transpose(1, 2).reshape(B, T, c.hidden_size)
            x = L.attention.output.LayerNorm(L.attention.output.dense(ctx) + x)
            h = L.mlp.gated_layers(x)
            h = torch.nn.functional.gelu(h[..., : c.intermediate_size]) * h[..., c.intermediate_size:]
            x = L.mlp.layernorm(L.mlp.wo(h) + x)
        return (x,)


# ---- a REMOTE-CODE checkpoint directory of the "new" class: what `AutoModel.from_pretrained(path, trust_remote_code=True)` (reference
# models/retrievers/dense.py:16) loads for gte-*-en-v1.5, offline — configuration_new.py / modeling_new.py beside the weights, auto_map in config.json

_CONFIGURATION_PY = """
from transformers import PretrainedConfig


class NewConfig(PretrainedConfig):
    model_type = "new"

    def __init__(self, vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, hidden_act="gelu",
                 max_position_embeddings=128, type_vocab_size=0, layer_norm_type="layer_norm", layer_norm_eps=1e-12, position_embedding_type="rope",
                 rope_theta=500000.0, rope_scaling=None, pack_qkv=True, unpad_inputs=False, use_memory_efficient_attention=False,
                 logn_attention_scale=False, logn_attention_clip1=False, **kwargs):
        super().__init__(**kwargs)
        self.vocab_size, self.hidden_size, self.num_hidden_layers, self.num_attention_heads = vocab_size, hidden_size, num_hidden_layers, num_attention_heads
        self.intermediate_size, self.hidden_act, self.max_position_embeddings, self.type_vocab_size = intermediate_size, hidden_act, max_position_embeddings, type_vocab_size
        self.layer_norm_type, self.layer_norm_eps, self.position_embedding_type = layer_norm_type, layer_norm_eps, position_embedding_type
        self.rope_theta, self.rope_scaling, self.pack_qkv, self.unpad_inputs = rope_theta, rope_scaling, pack_qkv, unpad_inputs
        self.use_memory_efficient_attention, self.logn_attention_scale, self.logn_attention_clip1 = use_memory_efficient_attention, logn_attention_scale, logn_attention_clip1
"""

_MODELING_TAIL = """

from transformers import PreTrainedModel

from .configuration_new import NewConfig


class NewModel(PreTrainedModel):
    config_class = NewConfig
    base_model_prefix = "new"

    def __init__(self, config, add_pooling_layer=False):
        super().__init__(config)
        self.embeddings = _Embeddings(config)
        self.encoder = _Encoder(config)
        self.break_it = None
        self.post_init()

    def _init_weights(self, module):
        pass

    forward = TorchNewModel.forward
"""


def write_remote_code_checkpoint(path, config_kwargs=None, seed=0):
    """Save a seeded TorchNewModel as a checkpoint directory with its modelling code beside the weights (fp16 safetensors, config.json with
    auto_map).  Returns the reference module (fp32, eval) whose weights were written."""
    import json
    import os
    from safetensors.torch import save_file
    cfg = new_config(**(config_kwargs or {}))
    model = TorchNewModel(cfg, seed=seed).eval()
    os.makedirs(path, exist_ok=True)
    here = open(os.path.abspath(__file__)).read()
    body = here[: here.index("# ---- JinaBert")]  # the "new" classes only
    open(os.path.join(path, "configuration_new.py"), "w").write(_CONFIGURATION_PY)
    open(os.path.join(path, "modeling_new.py"), "w").write(body + _MODELING_TAIL)
    conf = {k: v for k, v in vars(cfg).items() if not k.startswith("_")}
    conf.update(architectures=["NewModel"], auto_map={"AutoConfig": "configuration_new.NewConfig", "AutoModel": "modeling_new.NewModel"},
                torch_dtype="float16")
    json.dump(conf, open(os.path.join(path, "config.json"), "w"), indent=1)
    save_file({k: v.half().contiguous() for k, v in model.state_dict().items()}, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    return model


_JINA_CONFIGURATION_PY = """
from transformers import PretrainedConfig


class JinaBertConfig(PretrainedConfig):
    model_type = "bert"

    def __init__(self, vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, hidden_act="gelu",
                 max_position_embeddings=128, type_vocab_size=2, layer_norm_eps=1e-12, position_embedding_type="alibi", feed_forward_type="geglu",
                 emb_pooler="mean", **kwargs):
        super().__init__(**kwargs)
        self.vocab_size, self.hidden_size, self.num_hidden_layers, self.num_attention_heads = vocab_size, hidden_size, num_hidden_layers, num_attention_heads
        self.intermediate_size, self.hidden_act, self.max_position_embeddings, self.type_vocab_size = intermediate_size, hidden_act, max_position_embeddings, type_vocab_size
        self.layer_norm_eps, self.position_embedding_type, self.feed_forward_type, self.emb_pooler = layer_norm_eps, position_embedding_type, feed_forward_type, emb_pooler
"""

_JINA_MODELING_TAIL = """

from transformers import PreTrainedModel

from .configuration_bert import JinaBertConfig


class JinaBertModel(PreTrainedModel):
    config_class = JinaBertConfig
    base_model_prefix = "bert"

    def __init__(self, config, add_pooling_layer=False):
        super().__init__(config)
        inner = TorchJinaBert(config, seed=0)
        self.embeddings = inner.embeddings
        self.encoder = inner.encoder
        self.post_init()

    def _init_weights(self, module):
        pass

    forward = TorchJinaBert.forward
"""


def write_jina_remote_code_checkpoint(path, config_kwargs=None, seed=0):
    """The same for JinaBert (jina-embeddings-v2): model_type "bert" whose auto_map names the remote JinaBertModel."""
    import json
    import os
    from safetensors.torch import save_file
    cfg = jina_config(**(config_kwargs or {}))
    model = TorchJinaBert(cfg, seed=seed).eval()
    os.makedirs(path, exist_ok=True)
    here = open(os.path.abspath(__file__)).read()
    head = here[: here.index("def new_config(")]                                    # imports
    body = here[here.index("# ---- JinaBert"): here.index("# ---- a REMOTE-CODE checkpoint directory")]  # the JinaBert classes only
    open(os.path.join(path, "configuration_bert.py"), "w").write(_JINA_CONFIGURATION_PY)
    open(os.path.join(path, "modeling_bert.py"), "w").write(head + body + _JINA_MODELING_TAIL)
    conf = {k: v for k, v in vars(cfg).items() if not k.startswith("_")}
    conf.update(architectures=["JinaBertModel"], auto_map={"AutoConfig": "configuration_bert.JinaBertConfig", "AutoModel": "modeling_bert.JinaBertModel"},
                torch_dtype="float16")
    json.dump(conf, open(os.path.join(path, "config.json"), "w"), indent=1)
    save_file({k: v.half().contiguous() for k, v in model.state_dict().items()}, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    return model
