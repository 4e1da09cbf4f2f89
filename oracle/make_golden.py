"""
Regenerate tests/golden/* from the REAL reference code.  TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference, read-only):   python -m oracle.make_golden
The reference modules are imported unmodified (oracle/ref_import.py); nothing is copied from
them.  Outputs are small .npz / .trec / .pt fixtures; the inputs of the large case are NOT
stored — tests regenerate them from the recorded torch seed (same image on the GPU box).

Fixtures
  kat_small.npz         Q=4, N=17, d=16 hand-checkable case incl. exact ties and a tie across a
                        chunk boundary: inputs, reference-code outputs, canonical outputs.
  config1.npz           BASELINE.json configs[0] (S1 of SURVEY §8d): seed-0 Gaussian,
                        Q=1000 x N=100000 x d=768, IP top-50.  Reference-code outputs on the raw
                        fp32 data and on the fp16-rounded data (the embedding dtype, dense.py:16),
                        plus canonical-oracle outputs on the fp16-rounded data.
  cosine_small.npz      CosineSim reference-code outputs + canonical outputs, Q=16, N=3000, d=64.
  write_trec.trec       bytes written by the reference's utils.write_trec for a small result.
  ref_index/            an index folder written by the reference's own Retrieve.encode_and_save
                        driven by a fake encoder (chunk naming / cadence: retrieve.py:135-141).
"""
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, ref_import  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def s1_inputs():
    """S1: one generator, seed 0; q drawn first, then d."""
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1000, 768, generator=g)
    d = torch.randn(100000, 768, generator=g)
    return q, d


def kat_small_inputs():
    """Small integers / halves so every score is exact in every precision; rows 3, 7 and 12 are
    identical (three-way tie), rows 8 and 9 (either side of the 9|8 chunk split) tie as well."""
    rng = np.random.default_rng(1234)
    x = rng.integers(-3, 4, size=(17, 16)).astype(np.float32) * 0.5
    q = rng.integers(-2, 3, size=(4, 16)).astype(np.float32)
    x[7] = x[3]
    x[12] = x[3]
    x[9] = x[8]
    return q, x


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = ref_import.load()
    r = ref_import.make_reference_retrieve("dot", batch_size=512, batch_size_sim=2048)

    # ---- kat_small ---------------------------------------------------------------------
    q, x = kat_small_inputs()
    qt, xt = torch.from_numpy(q), torch.from_numpy(x)
    rs, ri, _ = r.load_collection_and_retrieve(qt, [xt[:9], xt[9:]], 5, dataset_size=17)
    cs, ci = c_oracle.canonical_search(q.astype(np.float16), x.astype(np.float16), 5)
    np.savez(os.path.join(GOLDEN, "kat_small.npz"), q=q, x=x, chunk_rows=np.array([9, 8]), k=5,
             ref_scores=rs.numpy(), ref_ids=ri.numpy(), canon_scores=cs, canon_ids=ci)
    print("kat_small: ref ids\n", ri.numpy(), "\ncanonical ids\n", ci)

    # ---- config 1 ----------------------------------------------------------------------
    q, d = s1_inputs()
    sizes = [50000, 50000]
    rs32, ri32, _ = r.load_collection_and_retrieve(q, list(torch.split(d, sizes)), 50, dataset_size=100000)
    qh, dh = q.half().float(), d.half().float()
    rsh, rih, _ = r.load_collection_and_retrieve(qh, list(torch.split(dh, sizes)), 50, dataset_size=100000)
    print("config1: reference done; canonical oracle (brute force fp64, ~1-2 min)...")
    cs, ci = c_oracle.canonical_search(q.half().numpy(), d.half().numpy(), 50)
    np.savez_compressed(os.path.join(GOLDEN, "config1.npz"), seed=0, nq=1000, n=100000, d=768, k=50,
                        ref_fp32_scores=rs32.numpy(), ref_fp32_ids=ri32.numpy().astype(np.int32),
                        ref_h_scores=rsh.numpy(), ref_h_ids=rih.numpy().astype(np.int32),
                        canon_h_scores=cs, canon_h_ids=ci.astype(np.int32),
                        checksum_q=float(q.double().sum()), checksum_d=float(d.double().sum()))
    print("config1: id agreement canonical vs reference(fp16-valued):",
          float((ci == rih.numpy()).all(axis=1).mean()))

    # ---- cosine ------------------------------------------------------------------------
    rc = ref_import.make_reference_retrieve("cos")
    g = torch.Generator().manual_seed(11)
    q = torch.randn(16, 64, generator=g).half()
    x = (torch.randn(3000, 64, generator=g) * torch.rand(3000, 1, generator=g) * 3).half()
    rs, ri, _ = rc.load_collection_and_retrieve(q.float(), [x.float()], 20, dataset_size=3000)
    xn = c_oracle.l2_normalize_rows(x.numpy())
    qn = c_oracle.l2_normalize_rows(q.numpy())
    cs, ci = c_oracle.canonical_search(qn, xn, 20)
    np.savez(os.path.join(GOLDEN, "cosine_small.npz"), q=q.numpy(), x=x.numpy(), k=20, ref_scores=rs.numpy(),
             ref_ids=ri.numpy(), canon_scores=cs, canon_ids=ci)

    # ---- write_trec --------------------------------------------------------------------
    scores = torch.tensor([[84.8125, 83.75, 8.769950866699219], [1.5, 0.333251953125, -2.0]], dtype=torch.float32)
    ref.utils.write_trec(os.path.join(GOLDEN, "write_trec.trec"), ["q1", "q2"],
                         [["12", "7", "24853636"], ["3", "1", "0"]], scores)
    np.save(os.path.join(GOLDEN, "write_trec_scores.npy"), scores.numpy())

    # ---- an index folder written by the reference's encode_and_save ---------------------
    import datasets

    class FakeEncoderModel(torch.nn.Module):
        def forward(self, x):
            return x

        def to(self, *a, **k):  # the reference hard-codes .to('cuda') (retrieve.py:124)
            return self

    class FakeDense:
        """Embeds text 'i' as a deterministic 8-dim fp16 vector; only what encode_and_save touches."""
        model_name = "fake/dense"

        def __init__(self):
            self.model = FakeEncoderModel()

        def collate_fn(self, batch, query_or_doc=None):
            key = 'generated_query' if query_or_doc == "query" else "content"
            return {"v": torch.tensor([[float(int(s[key]) % 7), float(int(s[key]) % 5), 1.0, 0.5 * int(s[key]),
                                        -1.0, 2.0, 0.25, float(int(s[key]) % 3)] for s in batch])}

        def __call__(self, query_or_doc, batch):
            return {"embedding": batch["v"].half()}

    out = os.path.join(GOLDEN, "ref_index")
    shutil.rmtree(out, ignore_errors=True)
    r2 = ref_import.make_reference_retrieve()
    r2.model = FakeDense()
    r2.batch_size = 4
    ds = datasets.Dataset.from_dict({"content": [str(i) for i in range(30)]})
    # chunk_size=12 -> save_every_n_batches = 3: files at batch idx 3, 6 and the last (7)
    r2.encode_and_save(ds, save_path=out, query_or_doc="doc", chunk_size=12)
    print("ref_index files:", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
