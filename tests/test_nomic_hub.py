"""CPU: the HUB form of NomicBert — what the reference's `AutoModel.from_pretrained("nomic-ai/nomic-embed-text-v1.5",
trust_remote_code=True)` (models/retrievers/dense.py:16, config/retriever/nomic-embed-text-v1.5.yaml) builds: the checkpoint's
auto_map selects a remote `NomicBertModel` whose tensors are called attn.Wqkv / attn.out_proj / mlp.fc11 / fc12 / fc2 / norm1 /
norm2 / emb_ln and whose config extends GPT2Config (n_embd, n_inner, activation_function "swiglu", rotary_emb_*).  The remote
modelling file does not exist offline; what pins the name mapping here is transformers' OWN loader: a checkpoint written with the
hub's tensor names loads into transformers' native NomicBertModel through the conversion mapping transformers ships for it
(conversion_mapping.py "nomic_bert"), and bergen_amd.encoder must arrive at the same tensors."""
import json
import os

import numpy as np
import pytest
import torch

from bergen_amd import encoder

from test_nomic_oracle import load_tiny

HUB_CFG_FIELDS = dict(model_type="nomic_bert", activation_function="swiglu", rotary_emb_fraction=1.0, rotary_emb_base=1000,
                      rotary_emb_interleaved=False, rotary_emb_scale_base=None, rotary_scaling_factor=2, max_trained_positions=2048,
                      qkv_proj_bias=False, mlp_fc1_bias=False, mlp_fc2_bias=False, prenorm=False, use_rms_norm=False, causal=False,
                      layer_norm_epsilon=1e-12, type_vocab_size=2, n_positions=8192,
                      auto_map={"AutoModel": "nomic-ai/nomic-bert-2048--modeling_hf_nomic_bert.NomicBertModel"})


def hub_config(cfg):
    """The tiny fixture's geometry under the hub config's field names."""
    return dict(HUB_CFG_FIELDS, n_embd=cfg["hidden_size"], n_head=cfg["num_attention_heads"], n_layer=cfg["num_hidden_layers"],
                n_inner=cfg["intermediate_size"], vocab_size=cfg["vocab_size"])


def to_hub_names(sd):
    """transformers-native NomicBert state dict (numpy) -> the hub checkpoint's names (inverse of the published mapping)."""
    out, qkv = {}, {}
    for k, v in sd.items():
        if k.startswith("layers."):
            k = "encoder." + k
        k = k.replace("embeddings.LayerNorm.", "emb_ln.")
        k = (k.replace(".self_attn.o_proj.", ".attn.out_proj.").replace(".mlp.up_proj.", ".mlp.fc11.").replace(".mlp.gate_proj.", ".mlp.fc12.")
             .replace(".mlp.down_proj.", ".mlp.fc2.").replace(".post_attention_layernorm.", ".norm1.").replace(".post_mlp_layernorm.", ".norm2."))
        hit = [p for p in ("q_proj", "k_proj", "v_proj") if f".self_attn.{p}." in k]
        if hit:
            qkv.setdefault(k.replace(f".self_attn.{hit[0]}.", ".attn.Wqkv."), {})[hit[0]] = v
        else:
            out[k] = v
    for k, parts in qkv.items():
        out[k] = np.concatenate([parts["q_proj"], parts["k_proj"], parts["v_proj"]], axis=0)
    return out


def test_hub_names_map_onto_the_same_tensors_as_the_native_form():
    cfg, sd, _ = load_tiny()
    hub = to_hub_names(sd)
    assert any(".attn.Wqkv.weight" in k for k in hub) and "emb_ln.weight" in hub and not any("self_attn" in k for k in hub)
    canon_native = encoder.canonical_config(dict(cfg, model_type="nomic_bert"))
    canon_hub = encoder.canonical_config(hub_config(cfg))
    for key in ("hidden_size", "num_attention_heads", "num_hidden_layers", "intermediate_size", "rotary_theta", "ffn_gated", "head_dim",
                "type_vocab_size", "layer_norm_eps", "vocab_size"):
        assert canon_hub[key] == canon_native[key], key
    assert canon_hub["max_position_embeddings"] == 2048  # min(n_positions, max_trained_positions): dynamic NTK never engages below it
    a = encoder.canonical_state_dict(canon_native, {k: torch.from_numpy(v) for k, v in sd.items()})
    b = encoder.canonical_state_dict(canon_hub, {k: torch.from_numpy(v) for k, v in hub.items()})
    pos = "embeddings.position_embeddings.weight"  # (a zero table sized by max_position_embeddings: 128 vs 2048 rows)
    assert set(a) == set(b)
    for k in a:
        if k != pos:
            assert torch.equal(a[k].float(), b[k].float()), k
    assert not b[pos].any() and b[pos].shape[0] == 2048
    # prefixes of task models and the rotary buffers of the remote class are tolerated
    hub2 = {"model." + k: v for k, v in hub.items()}
    hub2["model.encoder.layers.0.attn.rotary_emb.inv_freq"] = np.ones(32, np.float32)
    c = encoder.canonical_state_dict(canon_hub, {k: torch.from_numpy(v) for k, v in hub2.items()})
    assert set(c) == set(b)


def test_hub_names_agree_with_transformers_own_conversion(tmp_path):
    """A checkpoint directory with the HUB's tensor names, loaded by transformers' native NomicBertModel (its loader renames and
    splits by conversion_mapping.py): the tensors it ends up with are the ones encoder._nomic_hub_names produces."""
    from safetensors.numpy import save_file
    from transformers import NomicBertConfig, NomicBertModel
    cfg, sd, _ = load_tiny()
    hub = {k: np.ascontiguousarray(v.astype(np.float32)) for k, v in to_hub_names(sd).items()}
    path = str(tmp_path / "hub_named")
    os.makedirs(path)
    hf_cfg = NomicBertConfig(**{k: v for k, v in cfg.items() if k != "rope_theta"}, pad_token_id=1,
                             rope_parameters={"rope_type": "default", "rope_theta": float(cfg["rope_theta"])})
    hf_cfg.save_pretrained(path)
    save_file(hub, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    model = NomicBertModel.from_pretrained(path)
    theirs = {k: v.detach().float() for k, v in model.state_dict().items() if "inv_freq" not in k and "position_ids" not in k and "token_type_ids" not in k}
    ours = encoder._nomic_hub_names({k: torch.from_numpy(v) for k, v in hub.items()})
    assert set(theirs) == set(ours), (sorted(set(theirs) ^ set(ours))[:6])
    for k in theirs:
        assert torch.equal(theirs[k], ours[k].float()), k
    # ... and equal to the fixture the oracle is pinned on (nothing was lost in the round trip)
    for k, v in sd.items():
        assert np.array_equal(ours[k].numpy(), v.astype(np.float32)), k


def test_hub_config_refuses_what_the_kernels_do_not_compute():
    cfg, _, _ = load_tiny()
    good = hub_config(cfg)
    encoder.canonical_config(good)
    for bad in (dict(activation_function="gelu"), dict(rotary_emb_fraction=0.0), dict(rotary_emb_fraction=0.5), dict(rotary_emb_interleaved=True),
                dict(prenorm=True), dict(use_rms_norm=True), dict(rotary_emb_scale_base=512.0), dict(n_inner=None), dict(causal=True)):
        with pytest.raises(ValueError):
            encoder.canonical_config(dict(good, **bad))


def test_remote_style_config_object_is_a_reason_not_a_crash():
    """a GPT2Config-derived nomic_bert config whose geometry cannot be read (no n_inner) made unsupported_reason()
    return None and the conversion raise TypeError out of Dense.__init__.  Now: the hub form is read field by field, and what
    cannot be read is a ValueError -> a fallback reason."""
    from transformers import GPT2Config

    class RemoteStyleConfig(GPT2Config):
        model_type = "nomic_bert"

    cfg, _, _ = load_tiny()
    hc = hub_config(cfg)
    full = RemoteStyleConfig(**{k: v for k, v in hc.items() if k not in ("model_type", "auto_map")})
    canon = encoder.canonical_config(full)
    assert canon["hidden_size"] == cfg["hidden_size"] and canon["intermediate_size"] == cfg["intermediate_size"] and canon["ffn_gated"] == 1

    class Model:
        config = RemoteStyleConfig(n_embd=128, n_head=2, n_layer=2, vocab_size=100)  # GPT2's defaults: n_inner None, gelu_new, no rotary fields

    assert encoder.BertEncoder.unsupported_reason(Model()) is not None


def test_conversion_failure_falls_back_with_a_reason_and_require_native_raises(monkeypatch):
    from bergen_amd import dense

    class Cfg:
        model_type = "bert"
        hidden_size, num_attention_heads, num_hidden_layers, intermediate_size = 128, 2, 2, 512
        vocab_size, max_position_embeddings, type_vocab_size, hidden_act, layer_norm_eps = 100, 64, 2, "gelu", 1e-12
        _name_or_path = "some/bert"

    class Model:
        config = Cfg()

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)

    def boom(model, device=0):
        raise KeyError("encoder.layer.0.attention.self.query.weight")

    monkeypatch.setattr(encoder.BertEncoder, "from_hf", staticmethod(boom))
    m = Model()
    assert dense._native_encoder(m) is m and "conversion of the checkpoint failed" in m._bergen_amd_fallback_reason
    with pytest.raises(RuntimeError, match="require_native"):
        dense._native_encoder(Model(), require_native=True)
    monkeypatch.setenv("BERGEN_AMD_REQUIRE_NATIVE", "1")
    with pytest.raises(RuntimeError, match="refusing to fall back"):
        dense._native_encoder(Model())


def test_retrieve_require_native_refuses_an_hf_backend():
    import bergen_amd

    class Plug:
        model_name = "Alibaba-NLP/gte-base-en-v1.5"
        backend = "hf"
        fallback_reason = "model_type 'new' (BERT / DistilBERT / RoBERTa-family encoders only)"
        model = torch.nn.Identity()

    with pytest.raises(RuntimeError, match="require_native=True.*gte-base.*model_type 'new'"):
        bergen_amd.Retrieve(init_args=Plug(), require_native=True)
    bergen_amd.Retrieve(init_args=Plug()).close()  # default: the stage starts, the warning was the plug-in's


def test_retrieve_require_native_checks_the_query_encoder_too():
    """Round-5 advisor finding: an asymmetric plug-in whose QUERY encoder fell back to HF passed the check (only `.backend`, the
    document side, was read), and a plug-in without a `backend` attribute counted as native."""
    import bergen_amd
    from bergen_amd.encoder import BertEncoder

    native = BertEncoder.__new__(BertEncoder)
    native._h = None

    class Asym:
        model_name = "toy/asymmetric"
        backend = "hip"                      # what the old check read
        model = native
        query_encoder = torch.nn.Identity()  # the side that stayed on torch

    class Silent:  # no `backend` attribute at all, a torch module inside
        model_name = "toy/silent"
        model = torch.nn.Identity()

    for plug in (Asym(), Silent()):
        with pytest.raises(RuntimeError, match="require_native=True"):
            bergen_amd.Retrieve(init_args=plug, require_native=True)
        r = bergen_amd.Retrieve(init_args=plug)
        assert r.backend == "hf"
        r.close()

    class Sym:
        model_name = "toy/native"
        model = native

    r = bergen_amd.Retrieve(init_args=Sym(), require_native=True)
    assert r.backend == "hip"
    r.close()
