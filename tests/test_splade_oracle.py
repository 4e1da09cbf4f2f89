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


# ---- DistilBertForMaskedLM / RobertaForMaskedLM heads (config/retriever/splade-efficient.yaml: DistilBERT-MLM checkpoints) ----

def load_alt(kind):
    """tests/golden/splade_tiny_<kind>.npz (oracle/make_golden_splade.py main_alt): HF config dict (+ model_type), the HF state
    dict under HF's own names, inputs, the reference Splade.__call__'s embedding and two sequences of HF logits."""
    import ast
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"splade_tiny_{kind}.npz"))
    cfg = {str(k): ast.literal_eval(str(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    cfg["model_type"] = str(z["model_type"])
    sd = {k[3:]: z[k] for k in z.files if k.startswith("w::")}
    return z, cfg, sd


def canonical_alt(cfg, sd):
    """The product's own renaming (bergen_amd.encoder.canonical_config / canonical_state_dict — pure host code) applied to the HF
    names, as numpy fp32 under BertForMaskedLM's names: what BertEncoder hands the library, and what the oracle reads."""
    import torch
    from bergen_amd.encoder import canonical_config, canonical_state_dict
    canon = canonical_config(cfg)
    csd = canonical_state_dict(canon, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return canon, {k: v.detach().float().numpy() for k, v in csd.items()}


@pytest.mark.parametrize("kind", ["distilbert", "roberta"])
def test_other_mlm_heads_map_onto_the_bert_head(kind):
    """HF's DistilBERT / RoBERTa masked-LM heads are BertForMaskedLM's transform + decoder by other names: renamed by the product's
    canonical_state_dict and evaluated by the BERT oracle they must reproduce HF's logits and the reference Splade's vectors."""
    z, cfg, sd = load_alt(kind)
    canon, csd = canonical_alt(cfg, sd)
    for name in ("cls.predictions.transform.dense.weight", "cls.predictions.transform.dense.bias",
                 "cls.predictions.transform.LayerNorm.weight", "cls.predictions.transform.LayerNorm.bias",
                 "cls.predictions.decoder.weight"):
        assert name in csd, (kind, name, sorted(k for k in csd if not k.startswith("encoder.")))
    assert "cls.predictions.decoder.bias" in csd or "cls.predictions.bias" in csd
    assert not any(k.startswith(("vocab_", "lm_head.")) for k in csd)
    if "cls.predictions.bias" in csd:  # (HF keeps the decoder bias under both names; the library accepts either)
        csd.setdefault("cls.predictions.decoder.bias", csd["cls.predictions.bias"])
    ids, mask = z["input_ids"], z["attention_mask"]
    ocfg = dict(canon, hidden_act="gelu")
    off = int(canon.get("position_offset", 0))
    if off:  # RoBERTa numbers a left-aligned sequence's tokens pad_token_id + 1 + t (the library's position_offset): the oracle
        csd["embeddings.position_embeddings.weight"] = csd["embeddings.position_embeddings.weight"][off:]  # reads arange(T)
    h = bert_oracle.bert_forward(csd, ocfg, ids, mask, np.zeros_like(ids))
    logits = bert_oracle.mlm_logits(csd, ocfg, h)
    np.testing.assert_allclose(logits[:2][mask[:2] != 0], z["hf_logits"][mask[:2] != 0], rtol=0, atol=3e-4)
    emb = bert_oracle.splade_pool(logits, mask)
    np.testing.assert_allclose(emb, z["ref_emb"], rtol=0, atol=3e-4)
    assert 0.005 < (z["ref_emb"] > 0).mean() < 0.5
