"""GPU parity tests of the DeBERTa-v2 / v3 cross-encoder (the reference's default reranker, config/reranker/debertav3.yaml:3):
BertEncoder.classify with disentangled attention (attention_rel.hip) through the C ABI against the golden fixture produced by HF's
DebertaV2ForSequenceClassification driven through the reference's CrossEncoder.__call__ (tests/golden/deberta_tiny.npz), against
the fp64 oracle (oracle/deberta_oracle.py) at deberta-v3's own geometry (256 position buckets, sequences well into the
logarithmic bucket range), and end to end from a checkpoint directory.
Floating point: |logit - ref| <= 3e-2 absolute (fp16 storage, fp32 accumulation; logits are O(1))."""
import numpy as np
import pytest
import torch

from oracle import deberta_oracle

from test_deberta_oracle import load_golden

pytestmark = pytest.mark.gpu


def _cfg(cfg):
    c = dict(cfg)
    c["model_type"] = "deberta-v2"
    return c


def _native(cfg, sd):
    from bergen_amd import BertEncoder
    return BertEncoder(_cfg(cfg), {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=0)


def _kw(ids, mask):
    return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
            "token_type_ids": torch.from_numpy(mask.copy())}  # (token types are ignored, as HF ignores them at type_vocab_size 0)


@pytest.mark.parametrize("labels", [1, 3])
def test_golden_fixture(labels):
    z, cfg, sd = load_golden()
    if labels == 3:
        sd.update({k[5:]: z[k].astype(np.float32) for k in z.files if k.startswith("w_3::")})
    enc = _native(cfg, sd)
    assert enc.num_labels == labels and enc.disentangled
    got = enc.classify(_kw(z["input_ids"], z["attention_mask"]))
    assert got.dtype == torch.float32 and tuple(got.shape) == (z["input_ids"].shape[0], labels)
    err = np.abs(got.cpu().numpy() - z[f"ref_score_{labels}"]).max()
    assert err <= 3e-2, err
    # hidden states of the real tokens (pool 2) against HF's
    hid = enc(**_kw(z["input_ids"], z["attention_mask"]))[0].float().cpu().numpy()
    real = z["attention_mask"] != 0
    assert np.abs(hid[real] - z["ref_hidden"][real]).max() <= 5e-2
    one = enc.classify({k: v[2:3] for k, v in _kw(z["input_ids"], z["attention_mask"]).items()}).cpu().numpy()
    assert np.abs(one - got.cpu().numpy()[2:3]).max() <= 1e-6  # batch-composition invariant
    enc.close()


def test_deberta_v3_geometry_against_oracle():
    """position_buckets = 256 / max_position_embeddings = 512 (deberta-v3-*), 4 heads of 64, pairs of up to 400 tokens: relative
    distances far beyond the linear half of the bucket table, several 32-key blocks, the last one ragged."""
    cfg = dict(vocab_size=2000, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=512, type_vocab_size=0, layer_norm_eps=1e-7, hidden_act="gelu", relative_attention=True,
               position_buckets=256, norm_rel_ebd="layer_norm", share_att_key=True, pos_att_type="p2c|c2p",
               position_biased_input=False, max_relative_positions=-1)
    sd = deberta_oracle.random_deberta(cfg, seed=91, num_labels=1)
    rng = np.random.default_rng(92)
    lens = np.array([400, 333, 257, 64, 31, 5])
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, cfg["vocab_size"], size=(len(lens), T)).astype(np.int64) * mask
    enc = _native(cfg, sd)
    got = enc.classify(_kw(ids, mask)).cpu().numpy()
    ref = deberta_oracle.cross_encode(sd, cfg, ids, mask)
    err = np.abs(got - ref).max()
    print(f"deberta-v3 geometry: max abs err {err:.4g}, logits {ref.ravel()[:3]}")
    assert err <= 3e-2
    # the position GEMMs of all heads in one launch per term (the default at 2 span = 512 columns) and one launch per head
    # compute the same tiles with the same kernel: identical logits, bit for bit
    enc.set_option("rel_batched_gemm", 0)
    per_head = enc.classify(_kw(ids, mask)).cpu().numpy()
    assert np.array_equal(per_head.view(np.uint32), got.view(np.uint32))
    enc.close()
