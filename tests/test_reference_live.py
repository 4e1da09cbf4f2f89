"""CPU, build container only: the REAL reference code (imported unmodified from /root/reference)
against the oracle and against this package's host logic.  Skipped where the tree is absent."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle, compare, ref_import, ref_port
from tests.conftest import gap_tolerance

pytestmark = pytest.mark.needs_reference


def test_reference_search_vs_oracle_random_chunks():
    r = ref_import.make_reference_retrieve("dot")
    g = torch.Generator().manual_seed(5)
    q = torch.randn(33, 128, generator=g).half()
    x = torch.randn(5000, 128, generator=g).half()
    chunks = list(torch.split(x.float(), [1234, 2000, 1766]))
    rs, ri, _ = r.load_collection_and_retrieve(q.float(), chunks, 25, dataset_size=5000)
    cs, ci = c_oracle.canonical_search(q.numpy(), x.numpy(), 25)
    st = compare.compare_near_tie(cs, ci, rs.numpy(), ri.numpy(), gap_tol=gap_tolerance(q.numpy(), x.numpy()))
    assert st["exact_id_queries"] >= 31, st
    ps, pi = ref_port.load_collection_and_retrieve(q.float(), chunks, 25, 5000)
    assert torch.equal(pi, ri) and torch.equal(ps, rs)


def test_reference_cosine_vs_port():
    r = ref_import.make_reference_retrieve("cos")
    g = torch.Generator().manual_seed(6)
    q, x = torch.randn(5, 64, generator=g), torch.randn(300, 64, generator=g)
    rs, ri, _ = r.load_collection_and_retrieve(q, [x], 10, dataset_size=300)
    ps, pi = ref_port.load_collection_and_retrieve(q, [x], 10, 300, similarity_fn=ref_port.cosine_sim)
    assert torch.equal(pi, ri) and torch.equal(ps, rs)


def test_trec_writer_byte_identical(tmp_path):
    from bergen_amd import utils
    ref = ref_import.load()
    g = torch.Generator().manual_seed(1)
    scores = torch.randn(7, 9, generator=g).sort(dim=1, descending=True).values * 50
    scores[0, 0] = 84.8125
    q_ids = [f"q{i}" for i in range(7)]
    d_ids = [[str(int(v)) for v in torch.randint(0, 24853637, (9,), generator=g)] for _ in range(7)]
    a, b = tmp_path / "a.trec", tmp_path / "b.trec"
    ref.utils.write_trec(str(a), q_ids, d_ids, scores)
    utils.write_trec(str(b), q_ids, d_ids, scores.numpy())
    assert a.read_bytes() == b.read_bytes()
    assert ref.utils.load_trec(str(b)) == utils.load_trec(str(a))


def test_reference_reads_our_chunks_and_naming(tmp_path):
    from bergen_amd import utils
    ref = ref_import.load()
    d = tmp_path / "ours"
    d.mkdir()
    parts = [torch.randn(5, 4).half(), torch.randn(3, 4).half(), torch.randn(2, 4).half()]
    for idx, p in zip((292, 584, 600), parts):
        torch.save(p, d / f"embedding_chunk_{idx}.pt")
    assert torch.equal(ref.utils.load_embeddings(str(d)), utils.load_embeddings(str(d)))
    for args in [("indexes", "kilt-100w", "a_b", "doc"), ("indexes", "kilt_nq", "a_b", "query", "dev", "gen")]:
        assert ref.utils.get_index_path(*args) == utils.get_index_path(*args)
    a = ("runs", "kilt_nq", "kilt-100w", "a_b", "dev", 50, "copy")
    assert ref.utils.get_ranking_filename(*a) == utils.get_ranking_filename(*a)


def test_reference_golden_runs_pin_format_only():
    """The shipped runs/*.trec round-trip through our reader: 6 fields, 50 lines/query, non-increasing."""
    from bergen_amd import utils
    runs = os.path.join(ref_import.REFERENCE_ROOT, "runs")
    f = os.path.join(runs, "run.retrieve.top_50.sciq.kilt-100w.dev.Shitao_RetroMAE_MSMARCO_distill.trec")
    if not os.path.exists(f):
        pytest.skip("run file absent")
    q, d, s = utils.load_trec(f)
    assert len(q) > 0 and all(len(x) == 50 for x in d)
    assert all(all(a >= b for a, b in zip(x, x[1:])) for x in s)


def test_reference_dense_on_a_remote_code_checkpoint_vs_the_new_oracle(tmp_path):
    """The reference's route for gte-*-en-v1.5 end to end on CPU: its UNMODIFIED `Dense(model_name=<directory>)` calls
    `AutoModel.from_pretrained(..., trust_remote_code=True)` (models/retrievers/dense.py:16), which imports the modelling code that lies
    beside the weights — here tests/gte_torch_model.py's restatement of the "new" class written out as such a checkpoint (the real remote
    file is not available offline: parity with IT stays unpinned, oracle/new_oracle.py).  The reference's ClsPooler output must equal the
    numpy oracle's on the same token ids."""
    import transformers as T
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from oracle import new_oracle
    from gte_torch_model import write_remote_code_checkpoint
    ref = ref_import.load()
    words = ["the", "a", "of", "river", "city", "music", "science", "history", "water", "energy", "planet", "what", "is", "capital"]
    vocab = ["[CLS]", "[PAD]", "[SEP]", "[UNK]", "[MASK]"] + words
    t = Tokenizer(models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))
    t.normalizer = normalizers.BertNormalizer(lowercase=True)
    t.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    t.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1", special_tokens=[("[CLS]", 0), ("[SEP]", 2)])
    tok = T.PreTrainedTokenizerFast(tokenizer_object=t, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]", mask_token="[MASK]",
                                    model_input_names=["input_ids", "attention_mask"])
    path = str(tmp_path / "gte-remote")
    cfg_kw = dict(vocab_size=len(vocab), max_position_embeddings=64)
    torch_ref = write_remote_code_checkpoint(path, cfg_kw, seed=17)
    tok.save_pretrained(path)
    dense = ref.dense.Dense(model_name=path, max_len=32, pooler=ref.dense.ClsPooler(), similarity=ref.dense.CosineSim())
    assert type(dense.model).__name__ == "NewModel" and dense.model.config.model_type == "new"
    dense.model = dense.model.float()  # (the reference loads fp16; fp32 on the CPU for a tight comparison)
    dense.query_encoder = dense.model
    texts = ["the capital of the city", "what is music", "water energy planet river", "history"]
    batch = dense.collate_fn([{"content": x} for x in texts], "doc")
    with torch.no_grad():
        emb = dense("doc", {k: v for k, v in batch.items()})["embedding"].float().numpy()
    sd = {k: v.numpy() for k, v in torch_ref.state_dict().items()}
    cfg = dict(vars(torch_ref.config))
    want = new_oracle.encode(sd, cfg, batch["input_ids"].numpy(), batch["attention_mask"].numpy(), pooler="cls")
    np.testing.assert_allclose(emb, want, rtol=0, atol=2e-4)
