"""Shared by oracle/make_golden_ut1.py (build container: runs the REFERENCE on this checkpoint) and tests/test_gpu_ut1.py (GPU
box: runs bergen_amd on the same checkpoint): a seeded random-init BERT + a WordPiece tokenizer over the UT1 vocabulary,
written as an HF checkpoint directory.  torch's CPU generator is deterministic, so both sides get the same weights; the
golden file carries a checksum to prove it."""
import torch

SEED = 20260926
MAX_LEN = 128


def words_of(texts, limit=3000):
    """Lower-cased word / punctuation vocabulary of the texts (BertPreTokenizer units), most frequent first."""
    from collections import Counter
    from tokenizers import normalizers, pre_tokenizers
    norm, pre = normalizers.BertNormalizer(lowercase=True), pre_tokenizers.BertPreTokenizer()
    c = Counter()
    for t in texts:
        c.update(w for w, _ in pre.pre_tokenize_str(norm.normalize_str(t)))
    return [w for w, _ in sorted(c.items(), key=lambda kv: (-kv[1], kv[0]))[:limit]]


def tokenizer_for(words):
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = ["[CLS]", "[PAD]", "[SEP]", "[UNK]", "[MASK]"] + list(words)
    t = Tokenizer(models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))
    t.normalizer = normalizers.BertNormalizer(lowercase=True)
    t.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    t.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                     special_tokens=[("[CLS]", 0), ("[SEP]", 2)])
    tok = PreTrainedTokenizerFast(tokenizer_object=t, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]",
                                  mask_token="[MASK]", model_input_names=["input_ids", "token_type_ids", "attention_mask"])
    return tok, len(vocab)


def build_checkpoint(path, words):
    """-> (path, checksum of all weights in float64).  BERT, 2 layers x 128 hidden x 2 heads (64-dim heads, the geometry of
    retromae.yaml's encoder), erf-GELU, fp16 weights as the reference loads them (dense.py:16)."""
    import transformers as T
    tok, vocab = tokenizer_for(words)
    torch.manual_seed(SEED)
    cfg = T.BertConfig(vocab_size=vocab, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
                       max_position_embeddings=MAX_LEN + 2, hidden_act="gelu", hidden_dropout_prob=0.0,
                       attention_probs_dropout_prob=0.0, pad_token_id=1)
    model = T.BertModel(cfg).half().eval()
    # (random LayerNorm / bias parameters too: an init of ones and zeros would leave those code paths untested)
    g = torch.Generator().manual_seed(SEED + 1)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("LayerNorm.weight"):
                p.copy_((1.0 + 0.1 * torch.randn(p.shape, generator=g)).half())
            elif name.endswith("bias"):
                p.copy_((0.05 * torch.randn(p.shape, generator=g)).half())
            elif "embeddings" not in name:
                p.copy_((p.float() * 4.0).half())  # (std 0.08: scores that differ in more than the last bits)
    model.save_pretrained(path)
    tok.save_pretrained(path)
    checksum = float(sum(p.double().sum() for p in model.state_dict().values() if p.is_floating_point()))
    return path, checksum
