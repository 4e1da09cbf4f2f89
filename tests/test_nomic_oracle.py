"""CPU: the NomicBert oracle (oracle/nomic_oracle.py: rotary positions, gated SiLU feed-forward) against the golden fixture produced
by HF NomicBertModel itself and by the reference's own Dense on a checkpoint directory (oracle/make_golden_nomic.py), plus the
host-side mapping of such a checkpoint onto the HIP encoder's canonical tensors — everything that needs no GPU."""
import os

import numpy as np
import pytest

from oracle import nomic_oracle

from conftest import GOLDEN


def load_tiny():
    z = np.load(os.path.join(GOLDEN, "nomic_tiny.npz"))
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        v = str(v)
        cfg[str(k)] = v if k == "hidden_act" else (float(v) if "." in v or "e" in v else int(v))
    sd = {k[3:]: z[k].astype(np.float32) for k in z.files if k.startswith("w::")}
    return cfg, sd, z


def test_oracle_matches_hf_nomic_bert_hidden_states():
    cfg, sd, z = load_tiny()
    h = nomic_oracle.nomic_forward(sd, cfg, z["input_ids"], z["attention_mask"], z["token_type_ids"])
    m = z["attention_mask"] != 0
    # HF ran in fp32, the oracle in fp64: agreement to fp32 round-off on every real token
    assert np.abs(h[m] - z["hf_hidden"][m]).max() < 3e-5
    assert np.abs(z["hf_hidden"][m]).max() > 1.0  # (not a degenerate comparison)
    e = nomic_oracle.encode(sd, cfg, z["input_ids"], z["attention_mask"], z["token_type_ids"])
    assert np.abs(e - z["ref_mean"]).max() < 2e-5  # the reference's own MeanPooler on HF's hidden states


def test_oracle_matches_the_reference_dense_on_the_checkpoint_directory():
    """The reference's Dense (AutoModel + AutoTokenizer from the directory, prompts, MeanPooler) produced these embeddings from
    TEXT; the oracle gets the token ids its collate_fn made.  fp32 pass: round-off; the reference's native fp16 pass: fp16 noise."""
    cfg, sd, z = load_tiny()
    for side in ("doc", "query"):
        ids, mask = z[f"ref_{side}_input_ids"], z[f"ref_{side}_attention_mask"]
        e = nomic_oracle.encode(sd, cfg, ids, mask)
        assert np.abs(e - z[f"ref_{side}_emb_fp32"]).max() < 3e-5, side
        assert np.abs(e - z[f"ref_{side}_emb_fp16"]).max() < 1e-2, side
    q = nomic_oracle.encode(sd, cfg, z["ref_query_input_ids"], z["ref_query_attention_mask"], l2_normalize=True)
    d = nomic_oracle.encode(sd, cfg, z["ref_doc_input_ids"], z["ref_doc_attention_mask"], l2_normalize=True)
    assert np.abs(q @ d.T - z["ref_cosine_fp32"]).max() < 1e-4  # CosineSim.sim (dense.py:83-89)


def test_rotation_depends_on_the_position_only_through_differences():
    """Rotary attention scores depend on i - j only: shifting every position by a constant leaves the layer output unchanged —
    the property that lets the packed HIP encoder number each sequence's tokens from 0."""
    cos, sin = nomic_oracle.rotary_tables(40, 64, 1000.0)
    rng = np.random.default_rng(0)
    q, k = rng.standard_normal((12, 64)), rng.standard_normal((12, 64))
    rot = lambda x, p: x * cos[p] + nomic_oracle.rotate_half(x) * sin[p]
    p0 = np.arange(12)
    s0 = rot(q, p0) @ rot(k, p0).T
    s1 = rot(q, p0 + 17) @ rot(k, p0 + 17).T
    assert np.abs(s0 - s1).max() < 1e-9


def test_padding_does_not_change_real_tokens():
    cfg, sd, z = load_tiny()
    ids, mask = z["input_ids"], z["attention_mask"]
    h = nomic_oracle.nomic_forward(sd, cfg, ids, mask)
    b = int(np.argmin(mask.sum(1)))
    n = int(mask[b].sum())
    alone = nomic_oracle.nomic_forward(sd, cfg, ids[b:b + 1, :n], mask[b:b + 1, :n])
    assert np.abs(alone[0] - h[b, :n]).max() < 1e-9


def test_op_references():
    rng = np.random.default_rng(1)
    gu = rng.standard_normal((5, 16))
    want = gu[:, 0::2] / (1 + np.exp(-gu[:, 0::2])) * gu[:, 1::2]  # (gate, up) column pairs
    assert np.allclose(nomic_oracle.swiglu_ref(gu), want)
    qk = rng.standard_normal((3, 2 * 2 * 64))
    pos = np.array([0, 5, 9])
    out = nomic_oracle.rotary_ref(qk, pos, 2, 1000.0)
    assert np.allclose(out[0], qk[0])  # position 0: identity
    x = qk[1].reshape(4, 64)
    ang = 5 * 1000.0 ** (-np.arange(32) / 32.0)
    want = np.concatenate([x[:, :32] * np.cos(ang) - x[:, 32:] * np.sin(ang), x[:, 32:] * np.cos(ang) + x[:, :32] * np.sin(ang)], -1)
    assert np.allclose(out[1].reshape(4, 64), want)


def test_checkpoint_maps_onto_the_bert_shaped_stack():
    """encoder.canonical_config / canonical_state_dict on a NomicBert checkpoint (no GPU needed): rotary and gating flags, the
    renamed tensors, zero position table and biases, and gate / up rows INTERLEAVED (row 2 j = gate j, row 2 j + 1 = up j — the
    layout the GEMM's fold epilogue and bh_swiglu_kernel read; include/bergen_hip.h bh_encoder_config.ffn_gated)."""
    import torch
    from bergen_amd import encoder
    cfg, sd, z = load_tiny()
    canon = encoder.canonical_config(dict(cfg, model_type="nomic_bert"))
    assert canon["rotary_theta"] == 1000.0 and canon["ffn_gated"] == 1 and canon["head_dim"] == 64 and canon["position_offset"] == 0
    out = encoder.canonical_state_dict(canon, {k: torch.from_numpy(v) for k, v in sd.items()})
    d, f = cfg["hidden_size"], cfg["intermediate_size"]
    w1 = out["encoder.layer.1.intermediate.dense.weight"].numpy()
    assert w1.shape == (2 * f, d)
    assert np.array_equal(w1[0::2], sd["layers.1.mlp.gate_proj.weight"]) and np.array_equal(w1[1::2], sd["layers.1.mlp.up_proj.weight"])
    assert np.array_equal(out["encoder.layer.0.attention.self.query.weight"].numpy(), sd["layers.0.self_attn.q_proj.weight"])
    assert np.array_equal(out["encoder.layer.0.output.dense.weight"].numpy(), sd["layers.0.mlp.down_proj.weight"])
    assert np.array_equal(out["encoder.layer.1.output.LayerNorm.bias"].numpy(), sd["layers.1.post_mlp_layernorm.bias"])
    assert not out["embeddings.position_embeddings.weight"].any() and out["embeddings.position_embeddings.weight"].shape == (cfg["max_position_embeddings"], d)
    for name, n in (("attention.self.query", d), ("intermediate.dense", 2 * f), ("output.dense", d)):
        b = out[f"encoder.layer.0.{name}.bias"]
        assert b.shape == (n,) and not b.any()
    assert not any("mlp." in k or "self_attn." in k or k.startswith("layers.") for k in out)
    # configurations outside the HIP forward pass are refused with a reason (the plug-in then stays on HF torch, loudly)
    for bad in (dict(hidden_act="gelu"), dict(rope_parameters={"rope_type": "yarn", "rope_theta": 1000.0}), dict(num_attention_heads=4)):
        with pytest.raises(ValueError):
            encoder.canonical_config(dict(cfg, model_type="nomic_bert", **bad))
