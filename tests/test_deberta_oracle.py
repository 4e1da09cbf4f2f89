"""CPU: the DeBERTa-v2/v3 cross-encoder oracle (oracle/deberta_oracle.py) against the golden fixture produced by HF's own
DebertaV2ForSequenceClassification driven through the reference's CrossEncoder.__call__ (oracle/make_golden_deberta.py), and
the host-side conversion of a DeBERTa state dict to the encoder's canonical names."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import deberta_oracle


def load_golden():
    z = np.load(os.path.join(GOLDEN, "deberta_tiny.npz"))
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        v = str(v)
        cfg[str(k)] = (v == "True") if v in ("True", "False") else (float(v) if any(c in v for c in ".e") and v.replace(".", "").replace("e", "").replace("-", "").isdigit()
                                                                   else int(v) if v.lstrip("-").isdigit() else v)
    sd = {k[3:]: z[k].astype(np.float32) for k in z.files if k.startswith("w::")}
    return z, cfg, sd


def test_oracle_reproduces_the_hf_scores_and_hidden_states():
    z, cfg, sd = load_golden()
    hidden = deberta_oracle.deberta_forward(sd, cfg, z["input_ids"], z["attention_mask"])
    real = z["attention_mask"] != 0
    assert np.abs(hidden[real] - z["ref_hidden"][real]).max() < 2e-5
    assert np.abs(deberta_oracle.seqcls_logits(sd, hidden) - z["ref_score_1"]).max() < 2e-5
    sd3 = dict(sd)
    sd3.update({k[5:]: z[k].astype(np.float32) for k in z.files if k.startswith("w_3::")})
    assert np.abs(deberta_oracle.cross_encode(sd3, cfg, z["input_ids"], z["attention_mask"]) - z["ref_score_3"]).max() < 2e-5


def test_relative_index_table_properties():
    """t(delta) = clamp(bucket(delta) + span, 0, 2 span - 1): linear inside (-span/2, span/2), monotone, odd around the
    centre (the symmetry the kernels use to index c2p and p2c with ONE table), clamped at the ends."""
    for span, max_rel, L in ((16, 64, 60), (256, 512, 512), (256, 512, 300)):
        t = deberta_oracle.relative_index_table(L, span, max_rel)
        c = L - 1
        assert len(t) == 2 * L - 1 and t[c] == span
        half = min(span // 2, L - 1)
        assert np.array_equal(t[c - half + 1:c + half], span + np.arange(-half + 1, half))
        assert (np.diff(t) >= 0).all() and t.min() >= 0 and t.max() <= 2 * span - 1
        d = np.arange(1, L)
        assert np.array_equal(np.clip(2 * span - t[c - d], 0, 2 * span - 1), t[c + d])


def test_padding_does_not_change_real_tokens():
    z, cfg, sd = load_golden()
    ids, mask = z["input_ids"][:3], z["attention_mask"][:3]
    a = deberta_oracle.cross_encode(sd, cfg, ids, mask)
    wide = np.pad(ids, ((0, 0), (0, 4))), np.pad(mask, ((0, 0), (0, 4)))
    cfg2 = dict(cfg)
    assert np.abs(a - deberta_oracle.cross_encode(sd, cfg2, *wide)).max() < 1e-9


def test_unsupported_deberta_configurations_are_refused():
    from bergen_amd.encoder import canonical_config
    base = dict(model_type="deberta-v2", vocab_size=100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                max_position_embeddings=64, type_vocab_size=0, hidden_act="gelu", relative_attention=True, position_buckets=16,
                norm_rel_ebd="layer_norm", share_att_key=True, pos_att_type="p2c|c2p", position_biased_input=False)
    assert canonical_config(base)["rel_span"] == 16
    for key, val in (("relative_attention", False), ("pos_att_type", "c2p"), ("share_att_key", False), ("norm_rel_ebd", "none"),
                     ("position_buckets", -1), ("position_biased_input", True), ("conv_kernel_size", 3), ("type_vocab_size", 2)):
        with pytest.raises(ValueError):
            canonical_config(dict(base, **{key: val}))


def test_deberta_state_dict_is_renamed_to_the_encoder_names():
    import torch
    from bergen_amd.encoder import canonical_config, canonical_state_dict
    z, cfg, sd = load_golden()
    c = canonical_config(dict(cfg, model_type="deberta-v2"))
    hf = {("deberta." + k if not k.startswith(("pooler.", "classifier.")) else k): torch.from_numpy(v) for k, v in sd.items()}
    out = canonical_state_dict(c, hf)
    assert "encoder.layer.0.attention.self.query.weight" in out and "encoder.layer.1.attention.self.value.bias" in out
    assert not [k for k in out if "_proj" in k]
    assert tuple(out["encoder.rel_embeddings.weight"].shape) == (2 * cfg["position_buckets"], cfg["hidden_size"])
    assert float(out["embeddings.position_embeddings.weight"].abs().sum()) == 0.0 and "pooler.dense.weight" in out
