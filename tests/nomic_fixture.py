"""Shared by oracle/make_golden_nomic.py (build container: runs HF NomicBertModel and the REFERENCE's Dense on this checkpoint)
and the tests (which rebuild the same checkpoint directory from the weights stored in tests/golden/nomic_tiny.npz): a tiny
NomicBert geometry (64-dim heads: nomic-embed-text-v1.5 is 12 x 64), a WordPiece tokenizer over a fixed word list, a few texts."""
import os

CFG = dict(vocab_size=0, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
           max_position_embeddings=128, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="silu", rope_theta=1000.0)
MAX_LEN = 48

DOCS = ["The Eiffel Tower was completed in 1889 for the World's Fair in Paris.",
        "Photosynthesis converts light, water and carbon dioxide into sugar and oxygen.",
        "Mount Everest, on the border of Nepal and China, is the highest mountain above sea level.",
        "The mitochondrion is the organelle that produces most of a cell's chemical energy.",
        "Ada Lovelace wrote what is often called the first computer program, for Babbage's Analytical Engine.",
        "The Amazon river carries more water than any other river on Earth.",
        "A short one.",
        "Retrieval augmented generation feeds the passages a retriever finds to a language model as context, "
        "which lets the model answer from documents it was never trained on and cite where an answer came from."]
QUERIES = ["who wrote the first computer program", "highest mountain in the world", "what does photosynthesis produce",
           "when was the eiffel tower built"]


def _words():
    from collections import Counter
    from tokenizers import normalizers, pre_tokenizers
    norm, pre = normalizers.BertNormalizer(lowercase=True), pre_tokenizers.BertPreTokenizer()
    c = Counter()
    for t in DOCS + QUERIES + ["search_query: ", "search_document: "]:
        c.update(w for w, _ in pre.pre_tokenize_str(norm.normalize_str(t)))
    return [w for w, _ in sorted(c.items(), key=lambda kv: (-kv[1], kv[0]))]


WORDS = _words()


def tokenizer(words=None):
    """-> (PreTrainedTokenizerFast, vocabulary size): the tokenizer of tests/ut1_fixture.py over this fixture's words."""
    try:
        from ut1_fixture import tokenizer_for          # (pytest: tests/ is on sys.path)
    except ImportError:
        from tests.ut1_fixture import tokenizer_for    # (python -m oracle.make_golden_nomic from the repo root)
    return tokenizer_for(list(WORDS if words is None else words))


def build_checkpoint(path, sd_np, cfg, words=None):
    """Write an HF checkpoint directory (config.json with model_type nomic_bert, safetensors weights in fp16 — what the
    reference's Dense loads, dense.py:16 — and the tokenizer files) from a state dict of numpy arrays.  -> path"""
    import torch
    from transformers import NomicBertConfig, NomicBertModel
    tok, vocab = tokenizer(words)
    assert vocab == cfg["vocab_size"], (vocab, cfg["vocab_size"])
    hf = NomicBertConfig(**{k: v for k, v in cfg.items() if k != "rope_theta"}, pad_token_id=1,
                         rope_parameters={"rope_type": "default", "rope_theta": float(cfg["rope_theta"])})
    model = NomicBertModel(hf).eval()
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    assert not unexpected and not [m for m in missing if "position_ids" not in m and "token_type_ids" not in m and "inv_freq" not in m], (missing, unexpected)
    os.makedirs(path, exist_ok=True)
    model.half().save_pretrained(path)
    tok.save_pretrained(path)
    return path
