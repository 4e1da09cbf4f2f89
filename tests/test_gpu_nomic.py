"""GPU parity tests of the NomicBert path (config/retriever/nomic-embed-text-v1.5.yaml): rotary positions and the gated SiLU
feed-forward on the HIP kernels, through the C ABI, against the fp64 oracle (oracle/nomic_oracle.py) and the golden fixture that
HF NomicBertModel and the reference's own Dense produced (tests/golden/nomic_tiny.npz, oracle/make_golden_nomic.py).

Floating point (fp16 storage, fp32 math); tolerances, written here:
  rotary     |got - ref| <= 1e-3 * max|ref|    (one fp16 rounding of an fp32 rotation; angles from a table built in double)
  swiglu     |got - ref| <= 2e-3 * |ref| + 1e-3
  encoder    cosine(embedding, oracle) >= 0.999 and max-abs <= 3e-2 * max|ref|   (the encoder bound of DESIGN.md)
"""
import numpy as np
import pytest
import torch

from oracle import nomic_oracle

import nomic_fixture
from test_gpu_encoder import DEV, _check_embeddings, _native, h16
from test_nomic_oracle import load_tiny

pytestmark = pytest.mark.gpu


def test_rotary_kernel_matches_oracle():
    from bergen_amd import encoder
    rng = np.random.default_rng(5)
    for n_heads, rows in ((2, 77), (12, 1000), (16, 333)):
        qk = (rng.standard_normal((rows, 2 * n_heads * 64)) * 2).astype(np.float16)
        pos = rng.integers(0, 2048, size=rows).astype(np.int32)
        pos[:3] = (0, 1, 2047)
        got = encoder.rotary(h16(qk), torch.from_numpy(pos).to(DEV), n_heads, 1000.0, max_pos=2048).float().cpu().numpy()
        ref = nomic_oracle.rotary_ref(qk, pos, n_heads, 1000.0)
        assert np.array_equal(got[0], qk[0].astype(np.float32))  # position 0 is the identity, bit for bit
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), (n_heads, rows, np.abs(got - ref).max())


def test_swiglu_kernel_matches_oracle():
    from bergen_amd import encoder
    rng = np.random.default_rng(6)
    gu = (rng.standard_normal((515, 2 * 3072)) * 3).astype(np.float16)
    gu[0, 0:16:2] = (-60000, -30, -11, 0, 11, 30, 60000, 1)  # gates at the ends of the range: silu -> 0 or the identity, never NaN
    gu[0, 1:16:2] = 1
    got = encoder.swiglu(h16(gu)).float().cpu().numpy()
    ref = nomic_oracle.swiglu_ref(gu)
    assert np.isfinite(got[:, 8:]).all() and not np.isnan(got).any()
    ok = np.isfinite(ref) & (np.abs(ref) < 60000)
    assert (np.abs(got - ref)[ok] <= 2e-3 * np.abs(ref)[ok] + 1e-3).all(), np.abs(got - ref)[ok].max()


def test_gemm_with_the_gated_fold_in_its_epilogue_matches_oracle():
    """BH_EPI_SWIGLU of the persistent GEMM: weight rows are (gate, up) pairs, the epilogue writes silu(gate) * up — [M][N / 2] —
    and the [M][N] product never exists.  Against the fp64 product folded by the oracle, and against the unfused pair of kernels
    (plain GEMM, then bh_swiglu_kernel) on the same operands: the fused form rounds once (fp32 product -> fold -> fp16), the
    unfused one twice, so the two agree to fp16 round-off, not bit for bit."""
    from bergen_amd import encoder
    from oracle import bert_oracle
    rng = np.random.default_rng(8)
    for (M, N, K) in ((512, 512, 128), (1024, 6144, 768), (2304, 1024, 192)):
        a = (rng.standard_normal((M, K)) * 0.5).astype(np.float16)
        w = (rng.standard_normal((N, K)) * 0.2).astype(np.float16)
        b = (rng.standard_normal(N) * 0.5).astype(np.float16)
        out, _ = encoder.gemm_f16(h16(a), h16(w), bias=h16(b), gelu="swiglu")
        assert tuple(out.shape) == (M, N // 2)
        ref = nomic_oracle.swiglu_ref(bert_oracle.gemm_ref(a, w, b, 1))
        got = out.float().cpu().numpy().astype(np.float64)
        rms = float(np.sqrt((ref ** 2).mean()))
        assert (np.abs(got - ref) <= 2e-3 * np.abs(ref) + 2e-3 * rms).all(), (M, N, K, np.abs(got - ref).max(), rms)
        plain, _ = encoder.gemm_f16(h16(a), h16(w), bias=h16(b))
        two_step = encoder.swiglu(plain).float().cpu().numpy()
        assert np.abs(two_step - got).max() <= 4e-3 * np.abs(ref).max() + 1e-3


def test_encoder_matches_hf_nomic_fixture():
    """Hidden states of HF NomicBertModel and the reference's MeanPooler on them (CPU, fp32), 6 right-padded sequences."""
    cfg, sd, z = load_tiny()
    enc = _native(dict(cfg, model_type="nomic_bert"), sd)
    ids, mask, types = (torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "token_type_ids"))
    hidden = enc(input_ids=ids, attention_mask=mask, token_type_ids=types)[0]
    m = z["attention_mask"] != 0
    got = hidden.float().cpu().numpy()
    assert got.shape == z["hf_hidden"].shape and np.all(got[~m] == 0)
    _check_embeddings(hidden[torch.from_numpy(m).to(DEV)], z["hf_hidden"][m].astype(np.float64), "hidden states")
    kw = {"input_ids": ids, "attention_mask": mask, "token_type_ids": types}
    _check_embeddings(enc.encode_pooled(kw, "mean"), z["ref_mean"].astype(np.float64), "mean pooling")
    # a sequence alone == inside the batch, bit for bit (rotary angles are by token index, not by packed row)
    full = enc.encode_pooled(kw, "mean").cpu().numpy()
    for b in (1, 3):
        alone = enc.encode_pooled({k: v[b:b + 1] for k, v in kw.items()}, "mean").cpu().numpy()
        assert np.array_equal(alone[0].view(np.uint16), full[b].view(np.uint16)), b
    enc.close()


def test_encoder_nomic_embed_shape_against_oracle():
    """nomic-embed-text-v1.5's geometry (768 x 12 heads x 3072 gated, theta 1000) with 3 layers; 96 sequences of up to 300 tokens:
    more than 8 192 packed rows, so the layer stack runs as two micro-batches on two streams (rotary rows offset per half) —
    and once more as one, bit-identical."""
    cfg = dict(vocab_size=3000, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="silu", rope_theta=1000.0,
               model_type="nomic_bert", rope_parameters={"rope_type": "default", "rope_theta": 1000.0})
    sd = nomic_oracle.random_nomic(cfg, seed=41, scale=0.03)
    sd = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    rng = np.random.default_rng(42)
    B, T = 96, 300
    lens = rng.integers(20, T + 1, size=B)
    lens[:2] = (T, 1)
    ids = rng.integers(5, cfg["vocab_size"], size=(B, T)).astype(np.int64)
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    enc = _native(cfg, sd)
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    got = enc.encode_pooled(kw, "mean")
    assert enc.counters()["packed_rows"] >= 8192
    check = [0, 1, 2, 47, 48, 95]  # (the oracle runs these sequences alone: padding does not change real tokens)
    ref = np.stack([nomic_oracle.encode(sd, cfg, ids[b:b + 1, :lens[b]], mask[b:b + 1, :lens[b]])[0] for b in check])
    cos, err = _check_embeddings(got[torch.tensor(check, device=DEV)], ref, "nomic-embed shape, mean pooling")
    enc.set_option("micro_batches", 1)
    one = enc.encode_pooled(kw, "mean")
    assert torch.equal(one, got), "one micro-batch and two must agree bit for bit"
    enc.set_option("micro_batches", 2)
    # rotary positions in the Q | K GEMM's epilogue (rows of whole tiles; round 6) and by the standalone kernel (all rows): the same bits
    from bergen_amd import _lib
    try:
        _lib.set_option("gemm_rotary_fused", 0)
        standalone = enc.encode_pooled(kw, "mean")
    finally:
        _lib.set_option("gemm_rotary_fused", 1)
    assert torch.equal(standalone, got), "fused and standalone rotary must agree bit for bit"
    # the unfused feed-forward (plain GEMM into [rows][2 dff] + the fold kernel: what small batches run) meets the same bound
    enc.set_option("ffn_fused", 0)
    unfused = enc.encode_pooled(kw, "mean")
    cos_u, err_u = _check_embeddings(unfused[torch.tensor(check, device=DEV)], ref, "nomic-embed shape, unfused feed-forward")
    print(f"nomic-embed shape: fused cos {cos:.6f} err {err:.4g}; unfused cos {cos_u:.6f} err {err_u:.4g}")
    enc.close()


def test_dense_from_a_nomic_checkpoint_directory_matches_the_reference_dense(tmp_path):
    """bergen_amd.Dense on the checkpoint directory the reference's Dense produced the fixture from (rebuilt from the stored
    weights): AutoModel resolves to NomicBertModel, the plug-in puts it on the HIP path, and query / document embeddings of the
    fixture's texts — prompts, tokenizer, mean pooling — agree with the reference's own fp32 pass; so do the cosine scores."""
    import bergen_amd
    cfg, sd, z = load_tiny()
    ckpt = nomic_fixture.build_checkpoint(str(tmp_path / "ckpt"), sd, cfg, words=[str(w) for w in z["words"]])
    dense = bergen_amd.Dense(model_name=ckpt, max_len=nomic_fixture.MAX_LEN, pooler=bergen_amd.MeanPooler(), similarity=bergen_amd.CosineSim(),
                             prompt_q="search_query: ", prompt_d="search_document: ")
    assert dense.backend == "hip", "the NomicBert checkpoint did not resolve to the hand-written forward pass"
    embs = {}
    for side, texts, field in (("doc", z["doc_texts"], "content"), ("query", z["query_texts"], "generated_query")):
        batch = dense.collate_fn([{field: str(t)} for t in texts], side)
        assert np.array_equal(batch["input_ids"].numpy(), z[f"ref_{side}_input_ids"]), side  # same prompts, same tokenizer
        got = dense(side, batch)["embedding"]
        embs[side] = got.float().cpu().numpy().astype(np.float64)
        _check_embeddings(got, z[f"ref_{side}_emb_fp32"].astype(np.float64), f"{side} embeddings vs the reference's Dense")
    q = embs["query"] / np.linalg.norm(embs["query"], axis=1, keepdims=True)
    d = embs["doc"] / np.linalg.norm(embs["doc"], axis=1, keepdims=True)
    assert np.abs(q @ d.T - z["ref_cosine_fp32"]).max() < 5e-3
    assert np.array_equal(np.argmax(q @ d.T, axis=1), np.argmax(z["ref_cosine_fp32"], axis=1))


def test_hub_form_checkpoint_runs_on_the_hip_path_with_identical_embeddings():
    """The hub form of nomic-embed-text-v1.5 (remote-code tensor names attn.Wqkv / mlp.fc11 / fc12 / fc2 / norm1 / norm2 / emb_ln, GPT2-style
    config: what reference dense.py:16 builds with trust_remote_code=True) engages the same kernels as transformers' native form: the
    committed nomic_tiny fixture with its tensors renamed gives the SAME embeddings, bit for bit, and a model object carrying that
    config + state dict is accepted by Dense's selection (backend 'hip')."""
    from bergen_amd import BertEncoder, dense
    from test_nomic_hub import hub_config, to_hub_names
    cfg, sd, z = load_tiny()
    native = _native(dict(cfg, model_type="nomic_bert"), sd)
    hub = BertEncoder(hub_config(cfg), {k: torch.from_numpy(np.asarray(v)) for k, v in to_hub_names(sd).items()}, device=0)
    kw = {k: torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "token_type_ids")}
    a, b = native.encode_pooled(kw, "mean"), hub.encode_pooled(kw, "mean")
    assert torch.equal(a, b)
    ref = nomic_oracle.encode(sd, cfg, z["input_ids"], z["attention_mask"], z["token_type_ids"])
    _check_embeddings(b, ref, "hub-form NomicBert vs oracle")
    native.close()
    hub.close()

    class RemoteModel:  # what the remote class hands over: .config + .state_dict()
        class config:
            pass

        def state_dict(self):
            return {k: torch.from_numpy(np.asarray(v)) for k, v in to_hub_names(sd).items()}

    for k, v in hub_config(cfg).items():
        setattr(RemoteModel.config, k, v)
    enc = dense._native_encoder(RemoteModel(), require_native=True)
    assert dense.encoder_backend(enc) == "hip"
    assert torch.equal(enc.encode_pooled(kw, "mean"), a)
    enc.close()
