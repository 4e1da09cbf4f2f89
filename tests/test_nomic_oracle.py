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
