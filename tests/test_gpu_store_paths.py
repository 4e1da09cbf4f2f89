"""GPU: the output paths that leave the kernels as whole cache lines, against the plain paths — every one of them is pure data
movement, so the outputs must be identical bit for bit.  Options gemm_full_line_stores (0 = 32-byte row pieces straight from the MFMA
layout, 1 = row-major outputs through LDS as 128-byte lines, 2 = the default since round 5: also the blocked V^T output and the gated
fold's 64-byte rows) and attention_rel_wide_stores (DeBERTa attention: 16-byte context stores, default on).  Round 4 finished level 2 in its last GPU
seconds and left it off; round 5 ran it through the whole GPU suite and flipped the defaults (profiles/r05a_ab_full_line_level2.txt: BERT-base forward
14.40 -> 14.29 ms, NomicBert 20.49 -> 19.15 ms, identical embeddings)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def test_blocked_vt_and_fold_through_lds_are_bit_identical():
    """gemm_full_line_stores = 2: the blocked V^T output of the persistent GEMM as 8 x 128-byte rows per store instruction, and the
    gated fold's 64-byte row pieces through LDS.  BERT-base shape (V^T path) and NomicBert shape (fold path), two micro-batches:
    embeddings identical to level 1, bit for bit."""
    from bergen_amd import BertEncoder, _lib, synth
    rng = np.random.default_rng(7)
    B, T = 96, 200
    lens = rng.integers(30, T + 1, size=B)
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    base = dict(vocab_size=3000, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    try:
        for arch in ("bert", "nomic"):
            cfg = dict(base)
            if arch == "nomic":
                cfg.update(model_type="nomic_bert", hidden_act="silu", rope_theta=1000.0)
                sd = synth.random_nomic(cfg, seed=3, scale=0.03)
            else:
                sd = synth.random_bert(cfg, seed=3)
            enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
            ids = rng.integers(1, cfg["vocab_size"], size=(B, T)).astype(np.int64) * mask
            kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
            outs = {}
            for level in (1, 2, 0, 2):
                _lib.set_option("gemm_full_line_stores", level)
                outs.setdefault(level, []).append(enc.encode_pooled(kw, "mean").clone())
            assert enc.counters()["packed_rows"] >= 8192
            assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[2][0], outs[2][1]) and torch.equal(outs[0][0], outs[1][0]), arch
            assert bool(torch.isfinite(outs[2][0].float()).all())
            enc.close()
    finally:
        _lib.set_option("gemm_full_line_stores", 2)


def test_deberta_attention_wide_stores_are_bit_identical():
    """attention_rel_wide_stores = 1: the disentangled attention's context rows as 16-byte stores (the exchange attention.hip has)."""
    from bergen_amd import _lib
    from oracle import deberta_oracle
    from test_gpu_deberta import _kw, _native
    cfg = dict(vocab_size=2000, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=512, type_vocab_size=0, layer_norm_eps=1e-7, hidden_act="gelu", relative_attention=True,
               position_buckets=256, norm_rel_ebd="layer_norm", share_att_key=True, pos_att_type="p2c|c2p",
               position_biased_input=False, max_relative_positions=-1)
    sd = deberta_oracle.random_deberta(cfg, seed=91, num_labels=1)
    rng = np.random.default_rng(93)
    lens = np.array([400, 333, 257, 64, 31, 5, 320, 1])
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, cfg["vocab_size"], size=(len(lens), T)).astype(np.int64) * mask
    enc = _native(cfg, sd)
    try:
        _lib.set_option("attention_rel_wide_stores", 0)
        narrow = enc.classify(_kw(ids, mask)).cpu().numpy()
        _lib.set_option("attention_rel_wide_stores", 1)
        wide = enc.classify(_kw(ids, mask)).cpu().numpy()
        assert np.array_equal(narrow.view(np.uint32), wide.view(np.uint32))
        assert np.abs(wide - deberta_oracle.cross_encode(sd, cfg, ids, mask)).max() <= 3e-2
    finally:
        _lib.set_option("attention_rel_wide_stores", 1)
        enc.close()
