"""GPU parity tests of the bi-encoder forward pass: every HIP kernel through the C ABI against the fp64 oracle
(oracle/bert_oracle.py) and the HF-BertModel golden fixture (tests/golden/bert_tiny.npz).

Floating point: fp16 storage, fp32 accumulation.  Tolerances (written here, DESIGN.md "Numerics contract"):
  GEMM       |got - ref| <= 1.5e-3 * |ref| + 1.5e-3 * rms(ref)   (one fp16 rounding of an fp32-accumulated sum)
  attention  |got - ref| <= 4e-3 * max|ref|                      (probabilities rounded to fp16 before P.V)
  encoder    cosine(embedding, oracle) >= 0.999 and max-abs <= 3e-2 * max|ref|
"""
import os

import numpy as np
import pytest
import torch

from oracle import bert_oracle

from conftest import GOLDEN
from test_encoder_oracle import load_tiny

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def h16(x):
    return torch.from_numpy(np.asarray(x, np.float16)).to(DEV)


def rnd16(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float16)


def assert_gemm_close(got, ref, what):
    got = got.float().cpu().numpy().astype(np.float64)
    rms = float(np.sqrt((ref ** 2).mean()))
    err = np.abs(got - ref)
    bound = 1.5e-3 * np.abs(ref) + 1.5e-3 * rms
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())} / {bad.size} elements off; worst err {err.max():.4g} at "
                           f"{np.unravel_index(err.argmax(), err.shape)}, rms {rms:.3g}")


def test_permlane_probe():
    from bergen_amd import encoder
    assert encoder.permlane_mode() in (0, 1)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_gemm_variants_plain(variant):
    """Asymmetric operands, sizes that are not tile multiples (ragged M and N edges)."""
    from bergen_amd import encoder
    rng = np.random.default_rng(100 + variant)
    M, N, K = 300, 328, 192
    a, w = rnd16(rng, M, K), rnd16(rng, N, K)
    out, _ = encoder.gemm_f16(h16(a), h16(w), variant=variant)
    assert_gemm_close(out, bert_oracle.gemm_ref(a, w), f"variant {variant} plain")


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_gemm_epilogues(variant):
    from bergen_amd import encoder
    rng = np.random.default_rng(200 + variant)
    M, N, K = 515, 384, 256
    a, w = rnd16(rng, M, K, scale=0.5), rnd16(rng, N, K, scale=0.5)
    bias_c, bias_r = rnd16(rng, N), rnd16(rng, M)
    res = rnd16(rng, M, N)
    out, _ = encoder.gemm_f16(h16(a), h16(w), bias=h16(bias_c), bias_mode=1, variant=variant)
    assert_gemm_close(out, bert_oracle.gemm_ref(a, w, bias_c, 1), "bias per column")
    out, _ = encoder.gemm_f16(h16(a), h16(w), bias=h16(bias_r), bias_mode=2, variant=variant)
    assert_gemm_close(out, bert_oracle.gemm_ref(a, w, bias_r, 2), "bias per row")
    out, _ = encoder.gemm_f16(h16(a), h16(w), bias=h16(bias_c), residual=h16(res), variant=variant)
    assert_gemm_close(out, bert_oracle.gemm_ref(a, w, bias_c, 1, res), "bias + residual")
    out, _ = encoder.gemm_f16(h16(a), h16(w), bias=h16(bias_c), gelu=True, variant=variant)
    assert_gemm_close(out, bert_oracle.gemm_ref(a, w, bias_c, 1, gelu=True), "bias + gelu")


def test_gemm_persistent_many_tiles_per_block():
    """Persistent kernel with several tiles per workgroup (deferred stores, cross-tile prefetch), K from one stage
    (fewer stages than deferred-store slots) to 48 stages; every epilogue it supports."""
    from bergen_amd import encoder
    rng = np.random.default_rng(77)
    # (the last two shapes have more tiles than CUs and a partial last round: the balanced variants split them)
    for (M, N, K) in [(2048, 1024, 64), (1536, 1280, 320), (4096 + 1024, 4096, 128), (768, 256 * 100, 192)]:
        a, w = rnd16(rng, M, K, scale=0.5), rnd16(rng, N, K, scale=0.2)
        bc, br = rnd16(rng, N), rnd16(rng, M)
        for kw, ref in [(dict(), bert_oracle.gemm_ref(a, w)),
                        (dict(bias=h16(bc)), bert_oracle.gemm_ref(a, w, bc, 1)),
                        (dict(bias=h16(br), bias_mode=2), bert_oracle.gemm_ref(a, w, br, 2)),
                        (dict(bias=h16(bc), gelu=True), bert_oracle.gemm_ref(a, w, bc, 1, gelu=True))]:
            for variant in (7, 8, 10, 0):
                out, _ = encoder.gemm_f16(h16(a), h16(w), variant=variant, **kw)
                assert_gemm_close(out, ref, f"persistent v{variant} {M}x{N}x{K} {sorted(kw)}")


def test_gemm_full_line_stores_are_bit_identical():
    """Option gemm_full_line_stores (gemm_f16_persist.h PST bit 32): the persistent kernel's outputs pass through a wave-private
    LDS buffer and leave as whole 128-byte lines (8 rows per store instruction) instead of 32-byte pieces of 32 rows.  Only the
    route changes: every epilogue it covers must give the same bits as the direct stores, over shapes with several tiles per
    workgroup, ragged edges (handled by the other kernels) and K from one stage to 48."""
    from bergen_amd import _lib, encoder
    rng = np.random.default_rng(35)
    try:
        _lib.set_option("gemm_mfma16", 0)  # (gemm_f16_persist.h's paths: gemm_f16_p16.h, the default for these epilogues, has the full-line route only)
        for (M, N, K) in [(2048, 1024, 64), (4096 + 256, 3072, 768), (1024, 768, 3072), (2560 + 40, 1536 + 24, 192)]:
            a, w = rnd16(rng, M, K, scale=0.5), rnd16(rng, N, K, scale=0.2)
            bc = rnd16(rng, N)
            for kw, ref in [(dict(), bert_oracle.gemm_ref(a, w)), (dict(bias=h16(bc)), bert_oracle.gemm_ref(a, w, bc, 1)),
                            (dict(bias=h16(bc), gelu=True), bert_oracle.gemm_ref(a, w, bc, 1, gelu=True))]:
                outs = []
                for on in (0, 1, 2):
                    _lib.set_option("gemm_full_line_stores", on)
                    out, _ = encoder.gemm_f16(h16(a), h16(w), **kw)
                    outs.append(out.clone())
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), f"{M}x{N}x{K} {sorted(kw)}"
                assert_gemm_close(outs[1], ref, f"full-line stores {M}x{N}x{K} {sorted(kw)}")
    finally:
        _lib.set_option("gemm_full_line_stores", 2)  # (the library defaults)
        _lib.set_option("gemm_mfma16", 1)


def test_gemm_mfma16_kernel_against_oracle():
    """Option gemm_mfma16 (gemm_f16_p16.h): the bias and bias + GELU projections on v_mfma_f32_16x16x32_f16 — another LDS image,
    fragment pattern and epilogue route than gemm_f16_persist.h, the same contract.  Shapes with several tiles per workgroup, ragged
    edges (the other kernels take those strips), K from one stage to 48; against the fp64 oracle and against the production kernel
    (same tolerance: the two matrix instructions add their k terms in different groupings)."""
    from bergen_amd import _lib, encoder
    rng = np.random.default_rng(36)
    try:
        for (M, N, K) in [(256, 256, 64), (2048, 1024, 64), (4096 + 256, 3072, 768), (1024, 768, 3072), (2560 + 40, 1536 + 24, 192),
                          (768, 256 * 100, 128)]:
            a, w = rnd16(rng, M, K, scale=0.5), rnd16(rng, N, K, scale=0.2)
            bc = rnd16(rng, N)
            for kw, ref in [(dict(), bert_oracle.gemm_ref(a, w)), (dict(bias=h16(bc)), bert_oracle.gemm_ref(a, w, bc, 1)),
                            (dict(bias=h16(bc), gelu=True), bert_oracle.gemm_ref(a, w, bc, 1, gelu=True))]:
                outs = []
                for on in (0, 1):
                    _lib.set_option("gemm_mfma16", on)
                    out, _ = encoder.gemm_f16(h16(a), h16(w), variant=7, **kw)
                    outs.append(out.clone())
                assert_gemm_close(outs[1], ref, f"mfma16 {M}x{N}x{K} {sorted(kw)}")
                assert_gemm_close(outs[1], outs[0].float().cpu().numpy().astype(np.float64), f"mfma16 vs production {M}x{N}x{K} {sorted(kw)}")
    finally:
        _lib.set_option("gemm_mfma16", 1)  # (the library default)


def test_gemm_tail_split_gives_the_same_bits():
    """gemm_f16_p16.h's tail split: when the last round of an XCD's tiles would occupy at most half (a quarter) of its workgroups, those
    tiles are cut into 2 (4) sub-tiles along the tokens.  Only who computes an output element changes, not the k-steps it sums or their
    order: bit-identical to the run with the option off, for tile counts whose per-XCD remainders hit every case (no remainder, <= 8, <= 16,
    > 16, fewer tiles than workgroups, a single tile), K from one stage to 48, both epilogues."""
    from bergen_amd import _lib, encoder
    rng = np.random.default_rng(37)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    g8 = n_cu // 8
    cases = []  # (row tiles, column tiles, K)
    for tiles, K in [(1, 64), (8, 128), (8 * 3, 768), (8 * (g8 // 4), 192), (8 * (g8 // 4) + 8, 64), (8 * (g8 // 2), 768), (8 * (g8 // 2) + 8, 128),
                     (n_cu, 64), (n_cu + 8, 768), (n_cu + 8 * (g8 // 4) - 3, 256), (n_cu + 8 * (g8 // 2) + 5, 3072), (2 * n_cu + 16, 192)]:
        tn = 3 if tiles % 3 == 0 else 2 if tiles % 2 == 0 else 1
        cases.append((tiles // tn, tn, K))
    cases.append((130, 6, 768))  # the bench batch's micro-batch through the Q | K projection: 780 tiles
    try:
        for mode in (1,):
            _lib.set_option("gemm_mfma16", mode)
            for (tm, tn, K) in cases:
                M, N = 256 * tm, 256 * tn
                a, w = rnd16(rng, M, K, scale=0.5), rnd16(rng, N, K, scale=0.2)
                bc = rnd16(rng, N)
                for kw in (dict(bias=h16(bc)), dict(bias=h16(bc), gelu=True)):
                    outs = []
                    for on in (0, 1):
                        _lib.set_option("gemm_tail_split", on)
                        out, _ = encoder.gemm_f16(h16(a), h16(w), variant=7, **kw)
                        outs.append(out.clone())
                    assert torch.equal(outs[0], outs[1]), f"tail split changes bits: {tm} x {tn} tiles, K {K}, {sorted(kw)}"
                assert_gemm_close(outs[1], bert_oracle.gemm_ref(a, w, bc, 1, gelu=True), f"tail split {tm} x {tn} tiles, K {K}")
    finally:
        _lib.set_option("gemm_mfma16", 1)  # (the library default)
        _lib.set_option("gemm_tail_split", 1)


def test_gemm_alternating_loader_teams():
    """Variant 33 (gemm_f16_persist.h PST 16): deferred stores with the two four-wave teams of a workgroup alternating
    between refilling the LDS ring and storing — a wave skips the vmcnt wait of a stage it did not load in.  Several tiles per
    workgroup (4096 x 8192 = 512 tiles on 256 CUs: every store schedule runs across a tile boundary), stage counts 8 / 12 /
    48 (even, >= 8: the alternating kernel), 6 and 9 (falls back to the burst kernel), every epilogue; repeated launches must
    agree bit for bit (a missing wait shows up as a rare wrong tile, not as a steady error)."""
    from bergen_amd import encoder
    rng = np.random.default_rng(33)
    for (M, N, K) in [(4096, 8192, 512), (2048 + 256, 6144, 768), (1024, 2048, 3072), (1024, 1024, 384), (512, 512, 576)]:
        a, w = rnd16(rng, M, K, scale=0.5), rnd16(rng, N, K, scale=0.2)
        bc, br = rnd16(rng, N), rnd16(rng, M)
        for kw, ref in [(dict(), bert_oracle.gemm_ref(a, w)),
                        (dict(bias=h16(bc)), bert_oracle.gemm_ref(a, w, bc, 1)),
                        (dict(bias=h16(br), bias_mode=2), bert_oracle.gemm_ref(a, w, br, 2)),
                        (dict(bias=h16(bc), gelu=True), bert_oracle.gemm_ref(a, w, bc, 1, gelu=True))]:
            first = None
            for rep in range(3):
                out, _ = encoder.gemm_f16(h16(a), h16(w), variant=33, **kw)
                assert_gemm_close(out, ref, f"alternating teams {M}x{N}x{K} {sorted(kw)} run {rep}")
                if first is None:
                    first = out.clone()
                assert torch.equal(out, first), f"run-to-run difference {M}x{N}x{K} {sorted(kw)}"
            base, _ = encoder.gemm_f16(h16(a), h16(w), variant=7, **kw)
            assert torch.equal(base, first), "variant 33 must produce the bits of variant 7"


def test_gemm_identity_and_odd_n():
    """A = I picks out rows of B^T (catches transposed / permuted output maps); N not a multiple of 4."""
    from bergen_amd import encoder
    rng = np.random.default_rng(5)
    K = 128
    a = np.eye(K, dtype=np.float16)
    w = rnd16(rng, 77, K)
    out, _ = encoder.gemm_f16(h16(a), h16(w), out=torch.zeros((K, 80), dtype=torch.float16, device=DEV)[:, :77])
    # strided output view: ldc = 80 (multiple of 8), N = 77
    assert np.array_equal(out.cpu().numpy(), w.T)


def test_gemm_bert_shapes_auto_variant():
    from bergen_amd import encoder
    rng = np.random.default_rng(6)
    M = 1000
    for (N, K, gelu) in [(1536, 768, False), (3072, 768, True), (768, 3072, False)]:
        a, w, b = rnd16(rng, M, K, scale=0.3), rnd16(rng, N, K, scale=0.05), rnd16(rng, N)
        out, _ = encoder.gemm_f16(h16(a), h16(w), bias=h16(b), gelu=gelu)
        assert_gemm_close(out, bert_oracle.gemm_ref(a, w, b, 1, gelu=gelu), f"N={N} K={K}")
    # V^T orientation: weights as the row operand, bias per row
    x, wv, bv = rnd16(rng, 520, 768, scale=0.3), rnd16(rng, 768, 768, scale=0.05), rnd16(rng, 768)
    out, _ = encoder.gemm_f16(h16(wv), h16(x), bias=h16(bv), bias_mode=2)
    assert_gemm_close(out, bert_oracle.gemm_ref(wv, x, bv, 2), "V^T projection")


def _pack(lens):
    off, cur = [], 0
    for n in lens:
        off.append(cur)
        cur = (cur + n + 7) // 8 * 8
    return off, (cur + 32 + 255) // 256 * 256


@pytest.mark.parametrize("lens", [[1], [5, 31, 32, 33], [64, 100, 7, 256], [512, 3, 129]])
def test_attention_against_oracle(lens):
    from bergen_amd import encoder
    rng = np.random.default_rng(sum(lens))
    nh = 2
    d = nh * 64
    off, rows = _pack(lens)
    qk = rnd16(rng, rows, 2 * d)
    vt = rnd16(rng, d, rows)
    # make the softmax peaky for some queries: large q.k on a few keys
    qk[off[0], :64] *= 6
    ctx = encoder.attention(h16(qk), h16(vt), off, lens, nh, max(lens)).float().cpu().numpy()
    ref = bert_oracle.attention_ref(qk, vt, off, lens, nh)
    assert np.isfinite(ctx).all()
    assert np.abs(ctx - ref).max() <= 4e-3 * np.abs(ref).max(), np.abs(ctx - ref).max()
    # rows that belong to no sequence are never written
    keep = np.zeros(rows, bool)
    for o, n in zip(off, lens):
        keep[o:o + n] = True
    assert np.all(ctx[~keep] == 0)


def test_layernorm_against_oracle():
    """Both LayerNorm kernels: the general one-wave-per-row kernel and (option ln_small, the default; 768-wide rows) the 32-register kernel
    that fits beside a persistent GEMM workgroup — another summation tree, the same fp32 mathematics."""
    from bergen_amd import _lib, encoder
    rng = np.random.default_rng(9)
    try:
        for small in (1, 0):
            _lib.set_option("ln_small", small)
            for d in (128, 512, 768, 1024):
                for rows in (37, 4, 1):
                    x = rnd16(rng, rows, d, scale=3.0)
                    g, b = rnd16(rng, d) + np.float16(1), rnd16(rng, d)
                    got = encoder.layernorm(h16(x), h16(g), h16(b), 1e-12).float().cpu().numpy()
                    ref = bert_oracle.layernorm_ref(x, g, b, 1e-12)
                    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-3, (small, d, rows)
        # the 32-register kernel is as accurate as the general one: mean |error| against the fp64 reference over 2 048 rows of 768
        x = rnd16(rng, 2048, 768, scale=3.0)
        g, b = rnd16(rng, 768) + np.float16(1), rnd16(rng, 768)
        ref = bert_oracle.layernorm_ref(x, g, b, 1e-12)
        errs = {}
        for small in (1, 0):
            _lib.set_option("ln_small", small)
            errs[small] = float(np.abs(encoder.layernorm(h16(x), h16(g), h16(b), 1e-12).float().cpu().numpy() - ref).mean())
        assert errs[1] <= 1.05 * errs[0] + 1e-7, errs  # (equal to ~1e-3 relative: both are the fp16 rounding of the same fp32 values; the truncation bug was 2.4x)
    finally:
        _lib.set_option("ln_small", 1)


def _native(cfg, sd):
    from bergen_amd import BertEncoder
    return BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=0)


def _check_embeddings(got, ref, what):
    got = got.float().cpu().numpy().astype(np.float64)
    cos = (got * ref).sum(-1) / (np.linalg.norm(got, axis=-1) * np.linalg.norm(ref, axis=-1))
    assert cos.min() >= 0.999, f"{what}: min cosine {cos.min():.6f}"
    assert np.abs(got - ref).max() <= 3e-2 * np.abs(ref).max(), f"{what}: max abs err {np.abs(got - ref).max():.4g}"
    return float(cos.min()), float(np.abs(got - ref).max())


def test_encoder_matches_hf_golden_fixture():
    """tests/golden/bert_tiny.npz: outputs of HF BertModel + the reference's own poolers (CPU, fp32)."""
    cfg, sd, z = load_tiny()
    enc = _native(cfg, sd)
    ids, mask, types = (torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "token_type_ids"))
    hidden = enc(input_ids=ids, attention_mask=mask, token_type_ids=types)[0]
    m = z["attention_mask"] != 0
    got = hidden.float().cpu().numpy()
    assert got.shape == z["hf_hidden"].shape
    assert np.all(got[~m] == 0)
    _check_embeddings(hidden[torch.from_numpy(m).to(DEV)], z["hf_hidden"][m].astype(np.float64), "hidden states")
    kw = {"input_ids": ids, "attention_mask": mask, "token_type_ids": types}
    _check_embeddings(enc.encode_pooled(kw, "cls"), z["ref_cls"].astype(np.float64), "cls pooling")
    _check_embeddings(enc.encode_pooled(kw, "mean"), z["ref_mean"].astype(np.float64), "mean pooling")
    c = enc.counters()
    assert c["real_tokens"] == int(m.sum()) and c["batch"] == ids.shape[0] and c["forward_ms"] > 0
    enc.close()


def test_encoder_bert_base_shape_against_oracle():
    """12 x 768 x 12 heads x 3072 (RetroMAE / contriever shape), seeded random weights, ragged batch."""
    cfg = dict(vocab_size=2000, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=256, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=21)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=6, max_len=70, seed=22)
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
          "token_type_ids": torch.from_numpy(types)}
    ref_h = bert_oracle.bert_forward(sd, cfg, ids, mask, types)
    cos_c, err_c = _check_embeddings(enc.encode_pooled(kw, "cls"), bert_oracle.cls_pool(ref_h), "cls")
    cos_m, err_m = _check_embeddings(enc.encode_pooled(kw, "mean"), bert_oracle.mean_pool(ref_h, mask), "mean")
    got_n = enc.encode_pooled(kw, "mean", l2_normalize=True).float().cpu().numpy()
    assert np.allclose(np.linalg.norm(got_n, axis=1), 1.0, atol=2e-3)
    print(f"bert-base shape: cls cos {cos_c:.6f} err {err_c:.4g}; mean cos {cos_m:.6f} err {err_m:.4g}")
    enc.close()


def test_encoder_is_batch_composition_invariant():
    """Packing must not leak between sequences: a sequence encoded alone == inside a ragged batch, bit for bit
    (same kernels, same per-row arithmetic order)."""
    cfg, sd, z = load_tiny()
    enc = _native(cfg, sd)
    ids, mask, types = z["input_ids"], z["attention_mask"], z["token_type_ids"]
    kw = lambda s: {"input_ids": torch.from_numpy(ids[s]), "attention_mask": torch.from_numpy(mask[s]),
                    "token_type_ids": torch.from_numpy(types[s])}
    full = enc.encode_pooled(kw(slice(None)), "mean").cpu().numpy()
    for b in (0, 3, ids.shape[0] - 1):
        alone = enc.encode_pooled(kw(slice(b, b + 1)), "mean").cpu().numpy()
        assert np.array_equal(alone[0].view(np.uint16), full[b].view(np.uint16)), b
    # reversed batch order
    rev = enc.encode_pooled({k: torch.flip(v, [0]) for k, v in kw(slice(None)).items()}, "mean").cpu().numpy()
    assert np.array_equal(rev[::-1].view(np.uint16), full.view(np.uint16))
    enc.close()


def test_encoder_edge_cases_and_errors():
    cfg, sd, z = load_tiny()
    enc = _native(cfg, sd)
    one = torch.tensor([[5]])
    e1 = enc.encode_pooled({"input_ids": one}, "cls")  # mask None = all ones, single token
    ref = bert_oracle.encode(sd, cfg, one.numpy(), np.ones((1, 1), np.int64), pooler="cls")
    _check_embeddings(e1, ref, "single token")
    T = cfg["max_position_embeddings"]
    long_ids = torch.randint(1, cfg["vocab_size"], (2, T), generator=torch.Generator().manual_seed(1))
    ref = bert_oracle.encode(sd, cfg, long_ids.numpy(), np.ones((2, T), np.int64), pooler="mean")
    _check_embeddings(enc.encode_pooled({"input_ids": long_ids}, "mean"), ref, "max_position length")
    with pytest.raises(ValueError):  # all-zero mask
        enc.encode_pooled({"input_ids": one, "attention_mask": torch.zeros(1, 1, dtype=torch.long)}, "cls")
    with pytest.raises(ValueError):  # token id out of range
        enc.encode_pooled({"input_ids": torch.tensor([[cfg["vocab_size"]]])}, "cls")
    with pytest.raises(ValueError):  # longer than the position table
        enc.encode_pooled({"input_ids": torch.ones(1, T + 1, dtype=torch.long)}, "cls")
    with pytest.raises(ValueError):  # CLS pooling with the first token masked
        enc.encode_pooled({"input_ids": torch.ones(1, 3, dtype=torch.long),
                           "attention_mask": torch.tensor([[0, 1, 1]])}, "cls")
    enc.close()


def test_dense_plugin_runs_on_the_native_encoder(tmp_path):
    """Dense + Retrieve end to end on the HIP encoder and the HIP search: encode_and_save -> resident index ->
    fused search; the ranking must agree with an exact search over the oracle's embeddings."""
    import bergen_amd
    cfg, sd, _ = load_tiny()

    class ToyTokenizer:
        """whitespace 'tokenizer' with HF call semantics (padding='longest', truncation)."""

        def __call__(self, texts, padding=None, truncation=None, max_length=None, return_tensors=None):
            rows = [[1] + [2 + (hash_(w) % (cfg["vocab_size"] - 2)) for w in t.split()][:max_length - 1] for t in texts]
            T = max(len(r) for r in rows)
            ids = torch.tensor([r + [0] * (T - len(r)) for r in rows])
            mask = torch.tensor([[1] * len(r) + [0] * (T - len(r)) for r in rows])
            return {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}

    def hash_(w):
        v = 0
        for ch in w:
            v = (v * 131 + ord(ch)) % 1000003
        return v

    rng = np.random.default_rng(3)
    words = [f"w{i}" for i in range(300)]
    docs = [" ".join(rng.choice(words, size=int(rng.integers(3, 40)))) for _ in range(200)]
    queries = [" ".join(d.split()[:6]) for d in docs[:9]]
    enc = _native(cfg, sd)
    model = bergen_amd.Dense("toy/bert-tiny", 48, bergen_amd.MeanPooler, bergen_amd.DotProduct, model=enc,
                             tokenizer=ToyTokenizer())

    class DS(dict):
        pass

    class Col:
        def __init__(self, rows):
            self.rows = rows

        def __len__(self):
            return len(self.rows)

        def __getitem__(self, i):
            if isinstance(i, str):
                return [r[i] for r in self.rows]
            return self.rows[i]

        def remove_columns(self, cols):
            return Col([{k: v for k, v in r.items() if k not in cols} for r in self.rows])

    ds = {"doc": Col([{"id": str(i), "content": t} for i, t in enumerate(docs)]),
          "query": Col([{"id": f"q{i}", "generated_query": t} for i, t in enumerate(queries)])}
    r = bergen_amd.Retrieve(init_args=model, batch_size=64, batch_size_sim=4, num_workers=0)
    out = r.retrieve(ds, str(tmp_path / "q"), str(tmp_path / "d"), 10)
    r.close()
    tok = ToyTokenizer()
    bd, bq = tok(docs, max_length=48), tok(queries, max_length=48)
    ed = bert_oracle.encode(sd, cfg, bd["input_ids"].numpy(), bd["attention_mask"].numpy(), pooler="mean")
    eq = bert_oracle.encode(sd, cfg, bq["input_ids"].numpy(), bq["attention_mask"].numpy(), pooler="mean")
    want = np.argsort(-(eq @ ed.T), axis=1)[:, :10]
    got = np.array([[int(x) for x in row] for row in out["doc_id"]])
    overlap = np.mean([len(set(g) & set(w)) / 10 for g, w in zip(got, want)])
    assert overlap >= 0.9, overlap
    assert (got[:, 0] == want[:, 0]).mean() >= 0.8
    assert out["score"].shape == (len(queries), 10)


def test_encoder_bert_large_shape_against_oracle():
    """e5-large-v2 architecture (BASELINE configs[4]): 1024 hidden x 16 heads x 4096 FFN (4 of its 24 layers, to keep
    the fp64 oracle fast), mean pooling — d = 1024 exercises the blocked V^T layout and the 1024-wide row kernels."""
    cfg = dict(vocab_size=1500, hidden_size=1024, num_hidden_layers=4, num_attention_heads=16, intermediate_size=4096,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=51)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=5, max_len=150, seed=52, min_len=100)
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
          "token_type_ids": torch.from_numpy(types)}
    ref_h = bert_oracle.bert_forward(sd, cfg, ids, mask, types)
    _check_embeddings(enc.encode_pooled(kw, "mean"), bert_oracle.mean_pool(ref_h, mask), "bert-large mean")
    _check_embeddings(enc.encode_pooled(kw, "cls"), bert_oracle.cls_pool(ref_h), "bert-large cls")
    enc.close()


def test_encoder_large_batch():
    """> 65 536 packed tokens: every projection has more 256x256 tiles than the chip has CUs (several tiles per
    persistent workgroup, partial last round), the transposed V projection writes its blocked layout across them."""
    cfg = dict(vocab_size=800, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=128, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=61)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=760, max_len=128, seed=62, min_len=60)
    assert int(mask.sum()) > 66000
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
          "token_type_ids": torch.from_numpy(types)}
    got = enc.encode_pooled(kw, "mean")
    assert enc.counters()["packed_rows"] > 65536 + 256
    ref = bert_oracle.encode(sd, cfg, ids, mask, types, pooler="mean")
    _check_embeddings(got, ref, "large batch")
    enc.close()


def test_micro_batches_give_identical_outputs():
    """The layer stack over one, two, three or four micro-batches on as many streams (option micro_batches; default 2 from 8 192
    packed rows on): the same kernels on disjoint row ranges, so every output — CLS / mean pooled, hidden states, classification
    logits — must be BIT-identical to the single-stream forward pass, for ragged lengths whose split points do not fall on tile
    boundaries; and the default must match the oracle."""
    cfg = dict(vocab_size=900, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
               max_position_embeddings=160, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=71)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=301, max_len=150, seed=72, min_len=5)
    assert int(mask.sum()) > 17000
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(types)}
    outs = {}
    for n_mb in (1, 2, 3, 4, 2):
        enc.set_option("micro_batches", n_mb)
        outs[n_mb] = (enc.encode_pooled(kw, "mean").cpu(), enc.encode_pooled(kw, "cls").cpu(), enc(**kw)[0].cpu())
    for n_mb in (2, 3, 4):
        for got, want, what in zip(outs[n_mb], outs[1], ("mean", "cls", "hidden states")):
            assert torch.equal(got, want), f"micro_batches={n_mb}: {what} differs from the single-stream forward pass"
    ref = bert_oracle.encode(sd, cfg, ids, mask, types, pooler="mean")
    _check_embeddings(outs[2][0], ref, "two micro-batches vs the oracle")
    # a batch too small to split (< 8 192 packed rows) runs on one stream whatever the option says
    small = {k_: v[:20] for k_, v in kw.items()}
    a = enc.encode_pooled(small, "mean").cpu()
    enc.set_option("micro_batches", 1)
    assert torch.equal(a, enc.encode_pooled(small, "mean").cpu())
    enc.close()


def test_encoder_with_outlier_features_against_oracle():
    """Trained BERT-family checkpoints carry a few 'massive' hidden features (LayerNorm gains / biases an order of
    magnitude above the rest, present at every layer) and attention heads with very peaked softmax; seeded random
    weights have neither.  Inject both into the random weights and hold the fp16-storage / fp32-accumulate forward pass
    to the oracle — with the error bound taken over the ORDINARY features too, so that an outlier of 20 cannot hide an
    error of 0.5 on a feature of size 1."""
    cfg = dict(vocab_size=2000, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=256, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=51)
    hot = [77, 308, 381]                      # the outlier features
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    names = ["embeddings.LayerNorm"] + [f"encoder.layer.{l}.{n}" for l in range(12) for n in ("attention.output.LayerNorm", "output.LayerNorm")]
    for n in names:
        w, b = sd[n + ".weight"].copy(), sd[n + ".bias"].copy()
        w[hot] *= np.array([12.0, -9.0, 15.0], np.float32)
        b[hot] += np.array([8.0, -6.0, 11.0], np.float32)
        sd[n + ".weight"], sd[n + ".bias"] = f16(w), f16(b)
    for l in range(12):
        p = f"encoder.layer.{l}."
        sd[p + "output.dense.bias"][hot] += np.array([5.0, -4.0, 3.0], np.float32)
        sd[p + "output.dense.bias"] = f16(sd[p + "output.dense.bias"])
        q = sd[p + "attention.self.query.weight"].copy()
        q[:64] *= 6.0                          # head 0: logits six times larger -> near one-hot attention
        sd[p + "attention.self.query.weight"] = f16(q)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=5, max_len=90, seed=52)
    ref_h = bert_oracle.bert_forward(sd, cfg, ids, mask, types)
    assert np.abs(ref_h[..., hot]).max() > 10 * np.abs(np.delete(ref_h, hot, axis=-1)).mean()   # the outliers are there
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(types)}
    for mode, ref in (("cls", bert_oracle.cls_pool(ref_h)), ("mean", bert_oracle.mean_pool(ref_h, mask))):
        got = enc.encode_pooled(kw, mode).float().cpu().numpy().astype(np.float64)
        _check_embeddings(torch.from_numpy(got), ref, f"outliers, {mode}")
        ordinary = np.ones(768, bool)
        ordinary[hot] = False
        err = np.abs(got - ref)
        bound = 3e-2 * np.abs(ref[:, ordinary]).max()
        assert err[:, ordinary].max() <= bound, f"{mode}: ordinary features off by {err[:, ordinary].max():.4g} (bound {bound:.4g})"
        assert (err[:, hot] <= 3e-2 * np.abs(ref[:, hot]).max()).all(), f"{mode}: outlier features off by {err[:, hot].max():.4g}"
    enc.close()


def test_fused_layernorm_matches_the_separate_layernorm_kernels_and_the_oracle():
    """Option ln_fused (encoder.hip; OFF by default: correct, bit-reproducible and 2 % slower than the separate kernels — profiles/r05b_*):
    for batches whose GEMMs fill the chip the LayerNorm passes between the GEMMs disappear —
    the output-projection / FFN-down epilogues add the residual, store the pre-LayerNorm sum and its per-row (sum, sum of squares); the
    Q | K, V^T and FFN-up GEMMs read that tensor against weights folded with the LayerNorm's gain and normalise in their epilogue
    (LN(z) W^T + b = rstd (z W'^T - mean c) + b').  Same mathematics, different rounding points — tolerances, written here:
      fused vs separate kernels   cosine >= 0.9999 per embedding, |difference| <= 1.5e-2 * max|embedding|   (fp16 round-off of 3 layers)
      fused vs the fp32 oracle    the encoder bound of this file (cosine >= 0.999, 3e-2 * max)
    and the counters must SAY which path ran (a batch too small for the persistent GEMMs keeps the separate kernels)."""
    cfg = dict(vocab_size=2000, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=256, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=81)
    # LayerNorm gains and shifts far from (1, 0), and a hidden state with a mean far from 0: what the fold has to get right
    rng = np.random.default_rng(82)
    for k in list(sd):
        if k.endswith("LayerNorm.weight"):
            sd[k] = (1.0 + 0.5 * rng.standard_normal(sd[k].shape)).astype(np.float32)
        elif k.endswith("LayerNorm.bias"):
            sd[k] = (0.3 * rng.standard_normal(sd[k].shape) + 0.2).astype(np.float32)
        elif k.endswith("output.dense.bias"):
            sd[k] = (sd[k] + 0.5).astype(np.float32)   # shifts the mean of z = dense(x) + residual
    ids, mask, types = bert_oracle.random_batch(cfg, batch=270, max_len=200, seed=83, min_len=30)
    assert int(mask.sum()) > 28000
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(types)}
    out = {}
    for fused in (1, 0):
        enc.set_option("ln_fused", fused)
        out[fused] = (enc.encode_pooled(kw, "mean").float().cpu().numpy(), enc.encode_pooled(kw, "cls").float().cpu().numpy(),
                      enc(**kw)[0].float().cpu().numpy())
        assert enc.counters()["ln_fused"] == fused, "the counters must report the path that ran"
    for a, b, what in zip(out[1], out[0], ("mean", "cls", "hidden states")):
        a2, b2 = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
        keep = np.abs(b2).sum(1) > 0   # (padding positions of the hidden-state output are zero in both)
        cos = (a2[keep] * b2[keep]).sum(1) / (np.linalg.norm(a2[keep], axis=1) * np.linalg.norm(b2[keep], axis=1))
        assert cos.min() >= 0.9999, f"{what}: min cosine fused vs separate {cos.min():.6f}"
        assert np.abs(a2 - b2).max() <= 1.5e-2 * np.abs(b2).max(), f"{what}: max |difference| {np.abs(a2 - b2).max():.4g}"
        assert np.isfinite(a2).all()
    # the oracle on a subset of the sequences (the forward pass is batch-composition invariant; 270 sequences would take minutes)
    sub = slice(0, 24)
    ref = bert_oracle.encode(sd, cfg, ids[sub], mask[sub], types[sub], pooler="mean")
    _check_embeddings(torch.from_numpy(out[1][0][sub]), ref, "fused LayerNorm vs the oracle")
    _check_embeddings(torch.from_numpy(out[0][0][sub]), ref, "separate LayerNorm vs the oracle")
    # bit-identical across micro-batch counts and run to run (fixed statistic slots, no atomics)
    enc.set_option("ln_fused", 1)
    enc.set_option("micro_batches", 1)
    one = enc.encode_pooled(kw, "mean").cpu()
    assert enc.counters()["ln_fused"] == 1
    enc.set_option("micro_batches", 2)
    two = enc.encode_pooled(kw, "mean").cpu()
    assert torch.equal(one, two) and torch.equal(two, enc.encode_pooled(kw, "mean").cpu())
    # a batch whose GEMMs do not fill the chip keeps the separate kernels whatever the option says — and says so
    small = {k_: v[:12] for k_, v in kw.items()}
    e_small = enc.encode_pooled(small, "mean").float().cpu().numpy()
    assert enc.counters()["ln_fused"] == 0
    _check_embeddings(torch.from_numpy(e_small), bert_oracle.encode(sd, cfg, ids[:12], mask[:12], types[:12], pooler="mean"), "small batch")
    enc.close()


def test_fused_layernorm_on_the_other_stacks():
    """The fused path under the things that sit around it: RoBERTa-style position offsets, the SPLADE head and the classification
    head (both read the LAST layer's output, the one LayerNorm pass that stays), a d = 1024 / 16-head geometry (16 statistic slices per row)."""
    rng = np.random.default_rng(91)
    cfg = dict(vocab_size=1500, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
               max_position_embeddings=200, type_vocab_size=2, layer_norm_eps=1e-5, hidden_act="gelu")
    sd = bert_oracle.random_bert(cfg, seed=92)
    synth_mod = __import__("bergen_amd.synth", fromlist=["x"])
    synth_mod.random_cls_head(cfg, seed=93, num_labels=1, sd=sd)
    ids, mask, types = bert_oracle.random_batch(cfg, batch=190, max_len=180, seed=94, min_len=40)
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(types)}
    res = {}
    for fused in (1, 0):
        enc.set_option("ln_fused", fused)
        res[fused] = (enc.encode_pooled(kw, "mean").float().cpu().numpy(), enc.classify(kw).float().cpu().numpy())
        assert enc.counters()["ln_fused"] == fused
    a, b = res[1][0], res[0][0]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert cos.min() >= 0.9999 and np.abs(a - b).max() <= 1.5e-2 * np.abs(b).max(), (cos.min(), np.abs(a - b).max())
    assert np.abs(res[1][1] - res[0][1]).max() <= 2e-2 * max(1.0, np.abs(res[0][1]).max())
    sub = slice(0, 10)
    _check_embeddings(torch.from_numpy(a[sub]), bert_oracle.encode(sd, cfg, ids[sub], mask[sub], types[sub], pooler="mean"), "d = 1024 fused vs oracle")
    enc.close()
    del rng
