"""CPU: the encoder oracle (oracle/bert_oracle.py) against the golden fixture produced by HF BertModel itself
and the reference's own poolers (oracle/make_golden_encoder.py), plus the host-side encoder plumbing that needs
no GPU."""
import os

import numpy as np
import pytest

from oracle import bert_oracle

from conftest import GOLDEN


def load_tiny():
    z = np.load(os.path.join(GOLDEN, "bert_tiny.npz"))
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        v = str(v)
        cfg[str(k)] = v if k == "hidden_act" else (float(v) if "." in v or "e" in v else int(v))
    sd = {k[3:]: z[k].astype(np.float32) for k in z.files if k.startswith("w::")}
    return cfg, sd, z


def test_oracle_matches_hf_bert_hidden_states():
    cfg, sd, z = load_tiny()
    h = bert_oracle.bert_forward(sd, cfg, z["input_ids"], z["attention_mask"], z["token_type_ids"])
    m = z["attention_mask"] != 0
    # HF ran in fp32, the oracle in fp64: agreement to fp32 round-off on every real token
    assert np.abs(h[m] - z["hf_hidden"][m]).max() < 2e-5


def test_oracle_poolers_match_reference_poolers():
    cfg, sd, z = load_tiny()
    h = bert_oracle.bert_forward(sd, cfg, z["input_ids"], z["attention_mask"], z["token_type_ids"])
    assert np.abs(bert_oracle.mean_pool(h, z["attention_mask"]) - z["ref_mean"]).max() < 2e-5
    assert np.abs(bert_oracle.cls_pool(h) - z["ref_cls"]).max() < 2e-5
    e = bert_oracle.encode(sd, cfg, z["input_ids"], z["attention_mask"], z["token_type_ids"], pooler="mean",
                           l2_normalize=True)
    assert np.allclose(np.linalg.norm(e, axis=1), 1.0)


def test_padding_does_not_change_real_tokens():
    """The property the packed HIP encoder relies on: masked keys have zero weight, so a sequence encoded alone
    equals the same sequence inside a padded batch."""
    cfg, sd, z = load_tiny()
    ids, mask, types = z["input_ids"], z["attention_mask"], z["token_type_ids"]
    h = bert_oracle.bert_forward(sd, cfg, ids, mask, types)
    b = int(np.argmin(mask.sum(1)))
    n = int(mask[b].sum())
    alone = bert_oracle.bert_forward(sd, cfg, ids[b:b + 1, :n], mask[b:b + 1, :n], types[b:b + 1, :n])
    assert np.abs(alone[0] - h[b, :n]).max() < 1e-9


def test_op_references():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((5, 64))
    w = rng.standard_normal((7, 64))
    bias = rng.standard_normal(7)
    assert np.allclose(bert_oracle.gemm_ref(a, w, bias, 1), a @ w.T + bias)
    brow = rng.standard_normal(5)
    assert np.allclose(bert_oracle.gemm_ref(a, w, brow, 2), a @ w.T + brow[:, None])
    g = bert_oracle.gemm_ref(np.array([[1.0]]), np.array([[1.0]]), gelu=True)
    assert abs(g[0, 0] - 0.8413447460685429) < 1e-12  # gelu(1) = Phi(1)
    # attention over packed rows equals the padded computation
    nh, T = 2, 11
    qk = rng.standard_normal((24, 4 * 64))
    vt = rng.standard_normal((2 * 64, 24))
    ctx = bert_oracle.attention_ref(qk, vt, [0, 16], [T, 5], nh)
    assert np.all(ctx[T:16] == 0) and np.all(ctx[21:] == 0)
    q = qk[:T, :64]
    k = qk[:T, 128:192]
    s = q @ k.T / 8
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    assert np.allclose(ctx[:T, :64], p @ vt[:64, :T].T)


def test_random_bert_is_deterministic_and_fp16_exact():
    cfg, _, _ = load_tiny()
    a = bert_oracle.random_bert(cfg, 3)
    b = bert_oracle.random_bert(cfg, 3)
    for k in a:
        assert np.array_equal(a[k], b[k])
        assert np.array_equal(a[k], a[k].astype(np.float16).astype(np.float32))


def test_encoder_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bergen_amd import BertEncoder, _lib
    cfg, sd, _ = load_tiny()
    with pytest.raises(_lib.BergenHipError):
        BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()})


def test_encoder_abi_argument_validation_needs_no_device():
    import ctypes
    from bergen_amd import _lib
    lib = _lib.lib()
    h = ctypes.c_void_p()
    assert lib.bh_encoder_create(ctypes.byref(h), None) == _lib.BH_EINVAL
    bad = _lib.bh_encoder_config(n_layers=1, hidden=100, n_heads=2, intermediate=256, vocab_size=10, max_position=8,
                                 type_vocab_size=2, activation=0, ln_eps=1e-12)
    assert lib.bh_encoder_create(ctypes.byref(h), ctypes.byref(bad)) == _lib.BH_EUNSUPPORTED
    assert lib.bh_encoder_forward(None, None, None, None, 1, 1, 0, 0, None, 0) == _lib.BH_EINVAL
    assert lib.bh_op_gemm_f16(None, 0, None, 0, None, 0, None, 0, None, 0, 1, 1, 64, 0, 0, 1, None) == _lib.BH_EINVAL


def test_dense_fuses_pooling_with_a_native_style_encoder():
    """Dense.__call__ hands the HOST batch to encoder.encode_pooled when the encoder offers it."""
    import torch
    from bergen_amd import ClsPooler, Dense, DotProduct

    class FakeNative:
        def __init__(self):
            self.calls = []

        def encode_pooled(self, kwargs, pooler):
            self.calls.append((sorted(kwargs), pooler))
            return torch.ones(kwargs["input_ids"].shape[0], 4)

        def to(self, *a, **k):
            return self

    enc = FakeNative()
    d = Dense("fake/model", 16, ClsPooler, DotProduct, model=enc, tokenizer=object())
    out = d("doc", {"input_ids": torch.zeros(3, 5, dtype=torch.long), "attention_mask": torch.ones(3, 5, dtype=torch.long)})
    assert out["embedding"].shape == (3, 4) and enc.calls[0][1] is ClsPooler
