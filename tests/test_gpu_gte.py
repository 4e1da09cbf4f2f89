"""GPU: the "new" encoder architecture of Alibaba-NLP/gte-base-en-v1.5 / gte-large-en-v1.5 (config/retriever/gte-*-en-v1.5.yaml; reached by
the reference through AutoModel(trust_remote_code=True), models/retrievers/dense.py:16) on the HIP forward pass: NTK-scaled rotary
positions, packed biased q | k | v, GELU-gated feed-forward (fused into the GEMM epilogue or folded by the standalone kernel).
PARITY UNPINNED against the real remote modelling file (oracle/new_oracle.py); what IS tested: the kernels against the numpy
restatement and the torch restatement (tests/gte_torch_model.py), and the conversion SELF-CHECK that guards a real run — a module whose
arithmetic the mapping does not reproduce must stay on HF, loudly.

Floating point (fp16 storage, fp32 accumulation); tolerance as for the other encoders: cosine >= 0.999 per embedding and
max |diff| <= 3e-2 * max |ref|."""
import logging

import numpy as np
import pytest
import torch

from oracle import new_oracle

from gte_torch_model import TorchNewModel, new_config
from test_gte_oracle import _batch

pytestmark = pytest.mark.gpu


def _close(got, want, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    cos = (got * want).sum(-1) / (np.linalg.norm(got, axis=-1) * np.linalg.norm(want, axis=-1) + 1e-30)
    assert cos.min() >= 0.999, f"{what}: min cosine {cos.min()}"
    assert np.abs(got - want).max() <= 3e-2 * np.abs(want).max(), f"{what}: max abs {np.abs(got - want).max()} vs {np.abs(want).max()}"


@pytest.mark.parametrize("scaling", [{"type": "ntk", "factor": 2.0}, None])
def test_tiny_model_matches_the_oracle_and_passes_its_self_check(scaling):
    from bergen_amd import BertEncoder
    cfg = new_config(rope_scaling=scaling)
    model = TorchNewModel(cfg, seed=5).eval()
    enc = BertEncoder.from_hf(model, device=0)
    assert enc.needs_self_check and enc.self_check_result["min_cosine"] >= 0.995
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    ids, mask = _batch(cfg)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    hid = enc(**kw)[0].float().cpu().numpy()
    want = new_oracle.new_forward(sd, vars(cfg), ids, mask)
    m = mask.astype(bool)
    _close(hid[m], want[m], "hidden states")
    assert not hid[~m].any()
    for pooler in ("cls", "mean"):
        got = enc.encode_pooled(kw, pooler).float().cpu().numpy()
        _close(got, new_oracle.encode(sd, vars(cfg), ids, mask, pooler=pooler), f"pooled {pooler}")
    enc.close()


def test_gated_gelu_fold_fused_and_standalone():
    """bh_op_gated_act(act = gelu) vs the fp64 reference, and the GEMM's fused GELU-gated epilogue (16x16x32 kernel) vs the fp64 fold of
    the fp64 product and vs plain GEMM + standalone fold (same GELU arithmetic; the unfused form rounds gate and up to fp16 first)."""
    from bergen_amd import encoder
    rng = np.random.default_rng(8)
    M, K, F = 512, 768, 1024
    a = torch.from_numpy((rng.standard_normal((M, K)) * 0.5).astype(np.float16)).cuda()
    w = torch.from_numpy((rng.standard_normal((2 * F, K)) * 0.05).astype(np.float16)).cuda()
    b = torch.from_numpy((rng.standard_normal(2 * F) * 0.1).astype(np.float16)).cuda()
    gu, _ = encoder.gemm_f16(a, w, bias=b, bias_mode=1)
    folded = encoder.gated_act(gu.contiguous(), "gelu")
    ref = new_oracle.geglu_ref(gu.float().cpu().numpy())
    assert np.abs(folded.float().cpu().numpy() - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())
    fused, _ = encoder.gemm_f16(a, w, bias=b, bias_mode=1, gelu="geglu")
    assert fused.shape == (M, F)
    # the fused form rounds once (fp32 product -> fold -> fp16), the unfused one twice (gate and up to fp16 first): fp16 round-off apart,
    # like NomicBert's SiLU fold (tests/test_gpu_nomic.py); the fused output against the fp64 fold of the fp64 product
    exact = new_oracle.geglu_ref(a.float().cpu().numpy().astype(np.float64) @ w.float().cpu().numpy().astype(np.float64).T + b.float().cpu().numpy())
    tol = 2e-3 * max(1.0, np.abs(exact).max())
    assert np.abs(fused.float().cpu().numpy() - exact).max() <= tol
    assert np.abs(fused.float().cpu().numpy() - folded.float().cpu().numpy()).max() <= tol
    silu = encoder.gated_act(gu.contiguous(), "silu")
    assert torch.equal(silu.view(torch.int16), encoder.swiglu(gu.contiguous()).view(torch.int16))


def test_base_width_layers_take_the_fused_path_and_match_the_oracle():
    """Two layers at gte-base's width (768 x 12 heads, d_ff 3072) on 64 x ~128 tokens: rows and tiles that put the feed-forward on the
    fused GELU-gated epilogue; batch-composition invariance (alone vs in the batch) bit for bit."""
    from bergen_amd import BertEncoder
    cfg = new_config(hidden_size=768, num_attention_heads=12, intermediate_size=3072, vocab_size=2000, max_position_embeddings=512)
    sd = new_oracle.random_new(vars(cfg), seed=12)
    rng = np.random.default_rng(13)
    B, T = 64, 160
    lens = np.clip(np.rint(rng.normal(128, 20, size=B)), 16, T).astype(np.int64)
    lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(5, cfg.vocab_size, size=(B, T)).astype(np.int64) * mask
    enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    got = enc.encode_pooled(kw, "cls")
    want = new_oracle.encode(sd, vars(cfg), ids[:6], mask[:6], pooler="cls")
    _close(got[:6].float().cpu().numpy(), want, "gte-base-width CLS embeddings")
    # a sequence alone runs the unfused fold (fewer than half a chip of tiles): fp16 round-off apart from the fused batch, same bound
    alone = enc.encode_pooled({k: v[3:4] for k, v in kw.items()}, "cls")
    _close(alone.float().cpu().numpy(), want[3:4], "one sequence alone (unfused fold)")
    # NTK-scaled rotary positions in the Q | K GEMM's epilogue vs the standalone kernel on all rows: the same bits
    from bergen_amd import _lib
    try:
        _lib.set_option("gemm_rotary_fused", 0)
        standalone = enc.encode_pooled(kw, "cls")
    finally:
        _lib.set_option("gemm_rotary_fused", 1)
    assert torch.equal(standalone.view(torch.int16), got.view(torch.int16)), "fused and standalone rotary must agree bit for bit"
    # with the fused fold switched off the batch and the lone sequence agree bit for bit (batch-composition invariance of one path)
    enc.set_option("ffn_fused", 0)
    b0 = enc.encode_pooled(kw, "cls")
    a0 = enc.encode_pooled({k: v[3:4] for k, v in kw.items()}, "cls")
    assert torch.equal(a0.view(torch.int16)[0], b0.view(torch.int16)[3])
    enc.close()


def test_a_module_the_mapping_does_not_reproduce_stays_on_hf(caplog):
    """The safeguard of an unpinned mapping: a "new" module whose feed-forward differs from the description (act(up) * gate) fails the
    probe — from_hf raises, _native_encoder keeps the HF module with the reason, require_native refuses."""
    from bergen_amd import BertEncoder, dense
    model = TorchNewModel(new_config(), seed=6, break_it="swap_gate").eval()
    with pytest.raises(ValueError, match="self-check"):
        BertEncoder.from_hf(model, device=0)
    dense._warned.clear()
    with caplog.at_level(logging.WARNING, logger="bergen_amd"):
        kept = dense._native_encoder(model)
    assert kept is model and "self-check" in kept._bergen_amd_fallback_reason
    with pytest.raises(RuntimeError, match="require_native"):
        dense._native_encoder(model, require_native=True)
    good = dense._native_encoder(TorchNewModel(new_config(), seed=6).eval())
    assert isinstance(good, BertEncoder)
    good.close()


def test_dense_plugin_runs_gte_on_the_hip_path():
    """bergen_amd.Dense with an injected "new" module + tokenizer stand-in: backend 'hip', ClsPooler + CosineSim as in the yaml."""
    import bergen_amd
    cfg = new_config()
    model = TorchNewModel(cfg, seed=7).eval()

    class Tok:
        pad_token_id = 0
        padding_side = "right"

        def __call__(self, texts, **kw):
            rows = [[1] + [5 + (hash(w) % 500) for w in t.split()][:30] + [2] for t in texts]
            T = max(len(r) for r in rows)
            return {"input_ids": torch.tensor([r + [0] * (T - len(r)) for r in rows]),
                    "attention_mask": torch.tensor([[1] * len(r) + [0] * (T - len(r)) for r in rows])}

    d = bergen_amd.Dense(model_name="Alibaba-NLP/gte-base-en-v1.5", max_len=32, pooler=bergen_amd.ClsPooler(), similarity=bergen_amd.CosineSim(),
                         model=model, tokenizer=Tok(), require_native=True)
    assert d.backend == "hip" and d.backends == {"doc": "hip", "query": "hip"}
    batch = Tok()(["alpha beta gamma", "delta", "epsilon zeta eta theta iota"])
    emb = d("doc", batch)["embedding"].float().cpu().numpy()
    want = model(**batch)[0][:, 0].numpy()
    _close(emb, want, "Dense gte CLS embedding")


# ---- JinaBert (jina-embeddings-v2-base-en.yaml): ALiBi biases in the attention kernel + the GELU-gated feed-forward ----------

@pytest.mark.parametrize("heads,hidden", [(2, 128), (12, 768)])
def test_jina_alibi_model_matches_the_oracle_and_passes_its_self_check(heads, hidden):
    """heads = 12: the slopes of a head count that is not a power of two (8 + every other one of 16), as in jina-embeddings-v2-base."""
    from bergen_amd import BertEncoder
    from gte_torch_model import TorchJinaBert, jina_config
    cfg = jina_config(num_attention_heads=heads, hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=2, max_position_embeddings=256)
    model = TorchJinaBert(cfg, seed=21).eval()
    enc = BertEncoder.from_hf(model, device=0)
    assert enc.needs_self_check and enc.self_check_result["min_cosine"] >= 0.995
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(5)
    B, T = 6, 150  # longer than 128 tokens: the 8-wave attention instantiation; a short batch below: the 4-wave one
    lens = np.array([T, 140, 97, 64, 33, 5])
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(5, cfg.vocab_size, size=(B, T)).astype(np.int64) * mask
    types = (np.arange(T)[None, :] >= (lens[:, None] // 2)).astype(np.int64) * mask
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(types)}
    hid = enc(**kw)[0].float().cpu().numpy()
    want = new_oracle.jina_forward(sd, vars(cfg), ids, mask, types)
    m = mask.astype(bool)
    _close(hid[m], want[m], f"jina hidden states ({heads} heads)")
    short = {k: v[2:, :97] for k, v in kw.items()}
    got = enc.encode_pooled(short, "mean").float().cpu().numpy()
    hs = new_oracle.jina_forward(sd, vars(cfg), ids[2:, :97], mask[2:, :97], types[2:, :97])
    ms = mask[2:, :97].astype(np.float64)[..., None]
    _close(got, (hs * ms).sum(1) / ms.sum(1), "jina mean-pooled (4-wave attention)")
    enc.close()


def test_dense_loads_a_remote_code_checkpoint_onto_the_hip_path(tmp_path):
    """The reference's own route for gte-*-en-v1.5: `AutoModel.from_pretrained(model_name, trust_remote_code=True)` (models/retrievers/dense.py:16)
    on a checkpoint directory that carries its modelling code (configuration_new.py / modeling_new.py, auto_map in config.json) — here the torch
    restatement of tests/gte_torch_model.py written out as such a directory, loaded offline.  bergen_amd.Dense must put it on the HIP forward pass
    (self-check against the loaded remote module passed) and return that module's CLS embeddings."""
    import transformers as T
    import bergen_amd
    from gte_torch_model import write_remote_code_checkpoint
    from test_gpu_hf_path import TEXTS, WORDS
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    vocab = ["[CLS]", "[PAD]", "[SEP]", "[UNK]", "[MASK]"] + WORDS + ["?", ".", ","]
    t = Tokenizer(models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))
    t.normalizer = normalizers.BertNormalizer(lowercase=True)
    t.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    t.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1", special_tokens=[("[CLS]", 0), ("[SEP]", 2)])
    tok = T.PreTrainedTokenizerFast(tokenizer_object=t, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]", mask_token="[MASK]",
                                    model_input_names=["input_ids", "attention_mask"])
    path = str(tmp_path / "gte-remote")
    ref = write_remote_code_checkpoint(path, dict(vocab_size=len(vocab), max_position_embeddings=64), seed=11)
    tok.save_pretrained(path)
    dense = bergen_amd.Dense(model_name=path, max_len=32, pooler=bergen_amd.ClsPooler(), similarity=bergen_amd.CosineSim(), require_native=True)
    assert dense.backend == "hip" and dense.model.needs_self_check and dense.model.self_check_result["min_cosine"] >= 0.995
    batch = dense.collate_fn([{"content": x} for x in TEXTS], "doc")
    got = dense("doc", batch)["embedding"].float().cpu().numpy()
    want = ref(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"])[0][:, 0].numpy()
    _close(got, want, "Dense(model_name=<remote-code checkpoint>) CLS embeddings")


def test_dense_loads_a_jina_remote_code_checkpoint_onto_the_hip_path(tmp_path):
    """The same route for jina-embeddings-v2 (config/retriever/jina-embeddings-v2-base-en.yaml: MeanPooler + CosineSim): model_type "bert" whose
    auto_map names a remote JinaBertModel (ALiBi, GELU-gated feed-forward)."""
    import transformers as T
    import bergen_amd
    from gte_torch_model import write_jina_remote_code_checkpoint
    from test_gpu_hf_path import TEXTS, WORDS
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    vocab = ["[CLS]", "[PAD]", "[SEP]", "[UNK]", "[MASK]"] + WORDS + ["?", ".", ","]
    t = Tokenizer(models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))
    t.normalizer = normalizers.BertNormalizer(lowercase=True)
    t.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    t.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1", special_tokens=[("[CLS]", 0), ("[SEP]", 2)])
    tok = T.PreTrainedTokenizerFast(tokenizer_object=t, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]", mask_token="[MASK]",
                                    model_input_names=["input_ids", "token_type_ids", "attention_mask"])
    path = str(tmp_path / "jina-remote")
    ref = write_jina_remote_code_checkpoint(path, dict(vocab_size=len(vocab), max_position_embeddings=64), seed=13)
    tok.save_pretrained(path)
    dense = bergen_amd.Dense(model_name=path, max_len=32, pooler=bergen_amd.MeanPooler(), similarity=bergen_amd.CosineSim(), require_native=True)
    assert dense.backend == "hip" and dense.model.needs_self_check and dense.model.self_check_result["min_cosine"] >= 0.995
    batch = dense.collate_fn([{"content": x} for x in TEXTS], "doc")
    got = dense("doc", batch)["embedding"].float().cpu().numpy()
    hidden = ref(**{k: v for k, v in batch.items()})[0]
    want = bergen_amd.MeanPooler.pool(hidden, batch["attention_mask"]).numpy()
    _close(got, want, "Dense(model_name=<jina remote-code checkpoint>) mean embeddings")
