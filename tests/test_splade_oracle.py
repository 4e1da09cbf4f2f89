"""CPU: the numpy MLM-head / SPLADE-pooling oracle against HF BertForMaskedLM + the reference's Splade.__call__
(tests/golden/splade_tiny.npz, made by oracle/make_golden_splade.py)."""
import os

import numpy as np
import pytest

from oracle import bert_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "splade_tiny.npz")


def load(tag):
    z = np.load(GOLD)
    cfg = {k: (float(v) if "." in v or "e-" in v else int(v)) if v.replace(".", "").replace("e-", "").isdigit() else v
           for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    sd = {k[3:]: z[k].astype(np.float32) for k in z.files if k.startswith("w::")}
    sd.update({k.split("::", 1)[1]: z[k].astype(np.float32) for k in z.files if k.startswith(f"w_{tag}::")})
    return z, cfg, sd


@pytest.mark.parametrize("tag", ["untied", "tied"])
def test_oracle_matches_hf_and_reference(tag):
    z, cfg, sd = load(tag)
    assert ("cls.predictions.decoder.weight" in sd) == (tag == "untied")
    ids, mask, types = z["input_ids"], z["attention_mask"], z["token_type_ids"]
    h = bert_oracle.bert_forward(sd, cfg, ids, mask, types)
    logits = bert_oracle.mlm_logits(sd, cfg, h)
    # HF ran in fp32; the oracle in fp64 on the same (fp16-representable) weights
    np.testing.assert_allclose(logits[:2], z[f"hf_logits_{tag}"], rtol=0, atol=2e-4)
    emb = bert_oracle.splade_pool(logits, mask)
    ref = z[f"ref_emb_{tag}"]
    np.testing.assert_allclose(emb, ref, rtol=0, atol=2e-4)
    # same support (a term is active iff some attended token has a positive logit), up to fp32 noise at zero
    assert ((emb > 1e-4) == (ref > 1e-4)).mean() > 0.999
    assert 0.005 < (ref > 0).mean() < 0.5  # sparse-ish, like a trained SPLADE model


def test_max_commutes_with_log1p_relu():
    """The HIP path pools max_t relu(logit) and applies log(1 + .) afterwards."""
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((3, 7, 50)) * 2
    mask = np.array([[1] * 7, [1] * 4 + [0] * 3, [1] + [0] * 6])
    a = bert_oracle.splade_pool(logits, mask)
    m = np.where((mask != 0)[..., None], np.maximum(logits, 0), 0).max(1)
    np.testing.assert_array_equal(a, np.log1p(m))
