"""
Regenerate tests/golden/nomic_tiny.npz from HF ``NomicBertModel`` itself, driven through the reference's unmodified
``Dense``.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden_nomic        (build container; needs transformers, CPU only)

config/retriever/nomic-embed-text-v1.5.yaml: Dense(model_name, max_len 256, MeanPooler, CosineSim, prompts
"search_query: " / "search_document: ").  No checkpoint is available offline, so the pin is a seeded random-weight
NomicBertModel (rotary positions, gated SiLU feed-forward; transformers' native modeling_nomic_bert.py) written as a checkpoint
directory with a WordPiece tokenizer, then
  (1) the model's own forward (fp32, eager attention, eval mode) on a right-padded batch of random token ids -> hidden states,
      and the reference's MeanPooler on them;
  (2) the reference's ``Dense(model_name=<that directory>, ...)`` — AutoModel.from_pretrained(..., torch_dtype=float16,
      trust_remote_code=True) + AutoTokenizer, dense.py:14-35 — asked for query and document embeddings of a few texts through
      its own collate_fn (prompt prefix, padding, truncation) and __call__, once as loaded (fp16) and once with the model cast to
      fp32.
The fixture stores the weights (fp16-rounded: what both sides load), inputs, texts, vocabulary and outputs.
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nomic_oracle, ref_import  # noqa: E402
from tests import nomic_fixture  # noqa: E402


def main():
    from transformers import NomicBertModel
    cfg = dict(nomic_fixture.CFG)
    tok, vocab = nomic_fixture.tokenizer()
    cfg["vocab_size"] = vocab
    sd_np = {k: v.astype(np.float16).astype(np.float32) for k, v in nomic_oracle.random_nomic(cfg, seed=31).items()}
    work = tempfile.mkdtemp(prefix="nomic_")
    try:
        ckpt = nomic_fixture.build_checkpoint(os.path.join(work, "ckpt"), sd_np, cfg)
        model = NomicBertModel.from_pretrained(ckpt, attn_implementation="eager").float().eval()
        rng = np.random.default_rng(32)
        B, T = 6, 41
        lens = np.array([41, 17, 33, 8, 40, 25])
        ids = rng.integers(5, vocab, size=(B, T)).astype(np.int64)
        mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
        ids[mask == 0] = 1
        types = np.zeros((B, T), np.int64)
        with torch.no_grad():
            hidden = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), token_type_ids=torch.from_numpy(types))[0]
        assert ref_import.available(), "this fixture is generated in the build container, where /root/reference exists"
        ref = ref_import.load()
        mean = ref.dense.MeanPooler.pool(hidden, torch.from_numpy(mask))
        save = dict(cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([str(v) for v in cfg.values()]),
                    input_ids=ids, attention_mask=mask, token_type_ids=types, hf_hidden=hidden.numpy().astype(np.float32),
                    ref_mean=mean.numpy().astype(np.float32), words=np.array(nomic_fixture.WORDS),
                    doc_texts=np.array(nomic_fixture.DOCS), query_texts=np.array(nomic_fixture.QUERIES),
                    **{"w::" + k: v.astype(np.float16) for k, v in sd_np.items()})
        # (2) the reference's own plug-in on the checkpoint directory
        dense = ref.dense.Dense(model_name=ckpt, max_len=nomic_fixture.MAX_LEN, pooler=ref.dense.MeanPooler(), similarity=ref.dense.CosineSim(),
                                prompt_q="search_query: ", prompt_d="search_document: ")
        assert type(dense.model).__name__ == "NomicBertModel", type(dense.model)
        for tag, cast in (("fp16", False), ("fp32", True)):
            if cast:
                dense.model = dense.model.float()
                dense.query_encoder = dense.model
            for side, texts, field in (("doc", nomic_fixture.DOCS, "content"), ("query", nomic_fixture.QUERIES, "generated_query")):
                batch = dense.collate_fn([{field: t, "content": t, "generated_query": t} for t in texts], side)
                with torch.no_grad():
                    emb = dense(side, {k: v for k, v in batch.items()})["embedding"]
                save[f"ref_{side}_emb_{tag}"] = emb.float().numpy().astype(np.float32)
                if cast:
                    save[f"ref_{side}_input_ids"] = batch["input_ids"].numpy().astype(np.int64)
                    save[f"ref_{side}_attention_mask"] = batch["attention_mask"].numpy().astype(np.int64)
        sim = ref.dense.CosineSim.sim(torch.from_numpy(save["ref_query_emb_fp32"]), torch.from_numpy(save["ref_doc_emb_fp32"]))
        save["ref_cosine_fp32"] = sim.numpy().astype(np.float32)
        out = os.path.join(ROOT, "tests", "golden", "nomic_tiny.npz")
        np.savez_compressed(out, **save)
        print("wrote", out, os.path.getsize(out), "bytes; |fp16 - fp32| of the reference's own document embeddings:",
              float(np.abs(save["ref_doc_emb_fp16"] - save["ref_doc_emb_fp32"]).max()))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
