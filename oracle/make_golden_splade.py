"""
Regenerate tests/golden/splade_tiny.npz from HF ``BertForMaskedLM`` and the reference's ``Splade.__call__``.
TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden_splade        (build container; needs transformers + /root/reference, CPU only)

The reference encodes SPLADE vectors with ``AutoModelForMaskedLM.from_pretrained(...)`` and pools with
``max_t log(1 + relu(logits)) * mask`` (models/retrievers/splade.py:17,34-47).  No checkpoint is available
offline, so the pin is a seeded random-weight BertForMaskedLM (fp32, CPU, eval mode) run through the REAL HF
forward and the REAL, unmodified ``Splade.__call__`` (constructed without __init__, which would download a
checkpoint; its hard-coded ``.to('cuda')`` is neutralised by oracle/ref_import.py's Tensor.to patch).  Two
heads are stored: an untied decoder and a tied one (decoder = word embeddings, HF's default).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bert_oracle, ref_import  # noqa: E402

CFG = dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
           max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")


def hf_model(sd_np, tied):
    from transformers import BertConfig, BertForMaskedLM
    model = BertForMaskedLM(BertConfig(**CFG, attn_implementation="eager", tie_word_embeddings=tied)).eval()
    sd = {"bert." + k if not k.startswith("cls.") else k: torch.from_numpy(v) for k, v in sd_np.items()}
    sd["cls.predictions.bias"] = sd["cls.predictions.decoder.bias"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m and "token_type_ids" not in m
               and not (tied and m == "cls.predictions.decoder.weight")]
    assert not missing, missing
    assert not unexpected, unexpected
    if tied:
        assert model.cls.predictions.decoder.weight.data_ptr() == model.bert.embeddings.word_embeddings.weight.data_ptr()
    return model


def main():
    assert ref_import.available(), "needs /root/reference"
    ref = ref_import.load()
    out = {}
    ids, mask, types = bert_oracle.random_batch(CFG, batch=9, max_len=41, seed=22)
    for tag, tied in (("untied", False), ("tied", True)):
        sd_np = bert_oracle.random_bert(CFG, seed=21)
        bert_oracle.random_mlm_head(CFG, seed=23, tied=tied, sd=sd_np)
        model = hf_model(sd_np, tied)
        sp = object.__new__(ref.splade.Splade)  # skip __init__ (it downloads a checkpoint)
        sp.model = model
        sp.query_encoder = model
        kwargs = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
                  "token_type_ids": torch.from_numpy(types)}
        with torch.no_grad():
            emb = sp("doc", kwargs)["embedding"]
            logits = model(**kwargs).logits
        out[f"ref_emb_{tag}"] = emb.numpy().astype(np.float32)
        out[f"hf_logits_{tag}"] = logits[:2].numpy().astype(np.float32)  # first two sequences only (fixture size)
        for k, v in sd_np.items():
            if k.startswith("cls."):
                out[f"w_{tag}::" + k] = v.astype(np.float16)
        if not tied:
            for k, v in sd_np.items():
                if not k.startswith("cls."):
                    out["w::" + k] = v.astype(np.float16)
    path = os.path.join(ROOT, "tests", "golden", "splade_tiny.npz")
    np.savez_compressed(path, cfg_keys=np.array(list(CFG.keys())), cfg_vals=np.array([str(v) for v in CFG.values()]),
                        input_ids=ids, attention_mask=mask, token_type_ids=types, **out)
    print("wrote", path, os.path.getsize(path), "bytes; density",
          {t: float((out[f'ref_emb_{t}'] > 0).mean()) for t in ("untied", "tied")})


# ---- DistilBertForMaskedLM / RobertaForMaskedLM: the other masked-LM heads AutoModelForMaskedLM can hand the reference's Splade ----
# config/retriever/splade-efficient.yaml:3-4 names naver/efficient-splade-VI-BT-large-{doc,query}: DistilBertForMaskedLM checkpoints.
# The fixture stores the HF state dict under HF's OWN names (vocab_transform / vocab_layer_norm / vocab_projector; lm_head.*), so the
# tests also pin bergen_amd.encoder.canonical_state_dict's renaming onto BertForMaskedLM's cls.predictions.*.

ALT = {
    "distilbert": dict(vocab_size=1000, dim=128, n_heads=2, n_layers=2, hidden_dim=512, max_position_embeddings=64,
                       activation="gelu", sinusoidal_pos_embds=False, dropout=0.0, attention_dropout=0.0, pad_token_id=0),
    "roberta": dict(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                    max_position_embeddings=70, type_vocab_size=1, layer_norm_eps=1e-5, hidden_act="gelu", pad_token_id=1,
                    bos_token_id=0, eos_token_id=2),
}


def alt_model(kind, seed):
    import transformers as T
    torch.manual_seed(seed)
    if kind == "distilbert":
        model = T.DistilBertForMaskedLM(T.DistilBertConfig(**ALT[kind], attn_implementation="eager"))
    else:
        model = T.RobertaForMaskedLM(T.RobertaConfig(**ALT[kind], attn_implementation="eager"))
    model = model.eval()
    with torch.no_grad():  # HF initialises biases to 0 and LayerNorm gains to 1: give every head tensor content, fp16-representable
        g = torch.Generator().manual_seed(seed + 1)
        for name, p in model.named_parameters():
            if name.endswith(".bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "norm" in name.lower() and name.endswith(".weight"):
                p.copy_(1.0 + torch.randn(p.shape, generator=g) * 0.05)
            elif name.endswith(".weight"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.08)
            p.copy_(p.half().float())
        head_bias = model.vocab_projector.bias if kind == "distilbert" else model.lm_head.bias
        head_bias.copy_((head_bias - 2.2).half().float())  # sparse-ish output, like a trained SPLADE model
    return model


def main_alt():
    assert ref_import.available(), "needs /root/reference"
    ref = ref_import.load()
    for kind, seed in (("distilbert", 41), ("roberta", 43)):
        model = alt_model(kind, seed)
        cfg = ALT[kind]
        rng = np.random.default_rng(seed + 2)
        B, T = 7, 37
        lens = rng.integers(3, T + 1, size=B)
        lens[0] = T
        ids = rng.integers(3, cfg["vocab_size"], size=(B, T)).astype(np.int64)
        mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
        ids[mask == 0] = cfg["pad_token_id"]
        sp = object.__new__(ref.splade.Splade)  # skip __init__ (it downloads a checkpoint)
        sp.model = model
        sp.query_encoder = model
        kwargs = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        with torch.no_grad():
            emb = sp("doc", kwargs)["embedding"]
            logits = model(**kwargs).logits
        out = {"w::" + k: v.detach().numpy().astype(np.float16) for k, v in model.state_dict().items()
               if v.dtype.is_floating_point}
        path = os.path.join(ROOT, "tests", "golden", f"splade_tiny_{kind}.npz")
        np.savez_compressed(path, cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([repr(v) for v in cfg.values()]),
                            model_type=np.array(kind), input_ids=ids, attention_mask=mask,
                            ref_emb=emb.numpy().astype(np.float32), hf_logits=logits[:2].numpy().astype(np.float32), **out)
        print("wrote", path, os.path.getsize(path), "bytes; density", float((emb > 0).float().mean()))


if __name__ == "__main__":
    if "--alt" not in sys.argv:  # (the BERT fixture is regenerated only on request of the default invocation: it is seeded)
        main()
    main_alt()
