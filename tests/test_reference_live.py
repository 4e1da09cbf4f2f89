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
