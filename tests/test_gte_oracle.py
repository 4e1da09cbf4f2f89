"""CPU: the "new" (gte-*-en-v1.5) architecture — configuration and tensor mapping of bergen_amd.encoder, and the numpy oracle against
the independently written torch implementation (tests/gte_torch_model.py).  PARITY UNPINNED against the real remote modelling file
(not available offline: oracle/new_oracle.py says why); the product's own safeguard is the self-check of tests/test_gpu_gte.py."""
import numpy as np
import pytest
import torch

from bergen_amd import encoder
from oracle import bert_oracle, new_oracle

from gte_torch_model import TorchNewModel, new_config


def _batch(cfg, B=5, T=31, seed=3):
    rng = np.random.default_rng(seed)
    lens = rng.integers(3, T + 1, size=B)
    lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(5, cfg.vocab_size, size=(B, T)).astype(np.int64) * mask
    return ids, mask


@pytest.mark.parametrize("scaling", [{"type": "ntk", "factor": 2.0}, None])
def test_oracle_agrees_with_the_torch_restatement(scaling):
    cfg = new_config(rope_scaling=scaling)
    model = TorchNewModel(cfg, seed=5).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    ids, mask = _batch(cfg)
    want = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))[0].numpy()
    got = new_oracle.new_forward(sd, vars(cfg), ids, mask)
    m = mask.astype(bool)
    np.testing.assert_allclose(got[m], want[m], rtol=0, atol=2e-4)


def test_configuration_maps_onto_the_library_fields():
    c = encoder.canonical_config(new_config(hidden_size=768, num_attention_heads=12, num_hidden_layers=12, intermediate_size=3072,
                                            vocab_size=30528, max_position_embeddings=8192))  # gte-base-en-v1.5's config.json
    assert c["ffn_gated"] == 1 and c["hidden_act"] == "gelu" and c["head_dim"] == 64 and c["type_vocab_size"] == 1 and c["self_check"]
    assert c["rotary_theta"] == 1_000_000.0 and abs(c["rotary_scale"] - 2.0 ** (-2.0 / 64)) < 1e-12
    plain = encoder.canonical_config(new_config(rope_scaling=None, rope_theta=10000.0))
    assert plain["rotary_theta"] == 10000.0 and plain["rotary_scale"] == 1.0
    for bad, why in ((dict(position_embedding_type="absolute"), "rope only"), (dict(layer_norm_type="rms_norm"), "layer_norm_type"),
                     (dict(logn_attention_scale=True), "logn_attention_scale"), (dict(hidden_act="relu"), "hidden_act"),
                     (dict(rope_scaling={"type": "yarn", "factor": 2.0}), "rope_scaling"),
                     (dict(rope_scaling={"type": "ntk", "factor": 2.0, "mixed_b": 0.5}), "rope_scaling"),
                     (dict(num_attention_heads=4), "head dim")):
        with pytest.raises(ValueError, match=why):
            encoder.canonical_config(new_config(**bad))


def test_state_dict_maps_onto_the_bert_shaped_stack_and_reproduces_the_module():
    """canonical_state_dict's renaming / splitting / interleaving, checked ARITHMETICALLY: the BERT-shaped tensors it produces, run
    through a numpy forward pass with rotary positions and a (gate, up)-interleaved GELU-gated feed-forward — the library's layout —
    must give the torch module's hidden states."""
    cfg = new_config()
    model = TorchNewModel(cfg, seed=9).eval()
    canon = encoder.canonical_config(cfg)
    csd = {k: v.detach().float().numpy().astype(np.float64) for k, v in encoder.canonical_state_dict(canon, model.state_dict()).items()}
    d, f, nh = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
    assert csd["encoder.layer.0.intermediate.dense.weight"].shape == (2 * f, d) and not csd["embeddings.position_embeddings.weight"].any()
    assert not any("qkv_proj" in k or "up_gate" in k or "attn_ln" in k or "inv_freq" in k for k in csd)
    ids, mask = _batch(cfg, seed=4)
    want = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))[0].numpy()
    B, T = ids.shape
    dh = d // nh
    x = csd["embeddings.word_embeddings.weight"][ids] + csd["embeddings.token_type_embeddings.weight"][np.zeros_like(ids)] \
        + csd["embeddings.position_embeddings.weight"][np.arange(T)][None]
    x = bert_oracle._ln(x, csd["embeddings.LayerNorm.weight"], csd["embeddings.LayerNorm.bias"], 1e-12)
    # the library's table: angle = t * rotary_scale * rotary_theta^(-2j / 64)
    ang = np.arange(T)[:, None] * canon["rotary_scale"] * canon["rotary_theta"] ** (-2.0 * np.arange(32)[None, :] / 64.0)
    cos, sin = np.cos(np.concatenate([ang, ang], -1)), np.sin(np.concatenate([ang, ang], -1))
    rot = lambda t: np.concatenate([-t[..., 32:], t[..., :32]], -1)
    neg = np.where(mask[:, None, None, :] != 0, 0.0, -np.inf)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{l}."
        lin = lambda t, n: t @ csd[p + n + ".weight"].T + csd[p + n + ".bias"]
        q, k, v = (lin(x, "attention.self." + n).reshape(B, T, nh, dh).transpose(0, 2, 1, 3) for n in ("query", "key", "value"))
        q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
        s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(dh) + neg
        pr = np.exp(s - s.max(-1, keepdims=True))
        pr /= pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        x = bert_oracle._ln(lin(ctx, "attention.output.dense") + x, csd[p + "attention.output.LayerNorm.weight"], csd[p + "attention.output.LayerNorm.bias"], 1e-12)
        gu = lin(x, "intermediate.dense")
        h = new_oracle.geglu_ref(gu.reshape(-1, 2 * f)).reshape(B, T, f)
        x = bert_oracle._ln(lin(h, "output.dense") + x, csd[p + "output.LayerNorm.weight"], csd[p + "output.LayerNorm.bias"], 1e-12)
    m = mask.astype(bool)
    np.testing.assert_allclose(x[m], want[m], rtol=0, atol=2e-4)


# ---- JinaBert (jina-embeddings-v2): same standing — parity unpinned, two restatements compared, mapping checked arithmetically ----------

def test_jina_oracle_agrees_with_the_torch_restatement_and_the_mapping():
    from gte_torch_model import TorchJinaBert, jina_config
    cfg = jina_config(num_attention_heads=2)
    model = TorchJinaBert(cfg, seed=15).eval()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    ids, mask = _batch(cfg, seed=6)
    want = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))[0].numpy()
    got = new_oracle.jina_forward(sd, vars(cfg), ids, mask)
    m = mask.astype(bool)
    np.testing.assert_allclose(got[m], want[m], rtol=0, atol=2e-4)
    np.testing.assert_allclose(new_oracle.alibi_slopes(12), [2.0 ** -(i + 1) for i in range(8)] + [2.0 ** -(0.5 * (2 * i + 1)) for i in range(4)], rtol=1e-12)
    canon = encoder.canonical_config(cfg)
    assert canon["alibi"] == 1 and canon["ffn_gated"] == 1 and canon["self_check"] and canon["model_type"] == "bert"
    csd = encoder.canonical_state_dict(canon, model.state_dict())
    f, d = cfg.intermediate_size, cfg.hidden_size
    w = csd["encoder.layer.1.intermediate.dense.weight"]
    gl = model.state_dict()["encoder.layer.1.mlp.gated_layers.weight"]
    assert torch.equal(w[0::2], gl[:f]) and torch.equal(w[1::2], gl[f:]), "rows must interleave (gated j, non-gated j)"
    assert "encoder.layer.0.output.dense.weight" in csd and "encoder.layer.0.output.LayerNorm.bias" in csd
    assert not any(".mlp." in k for k in csd) and not csd["embeddings.position_embeddings.weight"].any()
    plain = encoder.canonical_config(jina_config(feed_forward_type="original"))
    assert plain["alibi"] == 1 and plain["ffn_gated"] == 0
    with pytest.raises(ValueError, match="feed_forward_type"):
        encoder.canonical_config(jina_config(feed_forward_type="glu"))
    with pytest.raises(ValueError, match="position_embedding_type"):
        encoder.canonical_config(jina_config(model_type="roberta", pad_token_id=1))
