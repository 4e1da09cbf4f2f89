"""SPLADE search (BASELINE configs[3]) on its own: python profiles/splade_search.py [docs] [option=value ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import SparseIndex, _lib, synth  # noqa: E402


def main():
    docs = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    _lib.init(0)
    sweep = ("", [None])
    for kv in sys.argv[2:]:
        k_, v_ = kv.split("=")
        if k_ in ("distinct", "power"):
            continue
        if "," in v_:
            sweep = (k_, [int(x) for x in v_.split(",")])  # e.g. sparse_ablate=0,1,4,16 (bench-only kernel ablations)
        else:
            _lib.set_option(k_, int(v_))
    V, block = 30522, 1_000_000
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    # the corpus of bench.py's splade leg: distinct 1 M-document blocks (term sets from 3 independent draws, fresh weights per
    # block) — or, `distinct=0`, round 2's single block repeated (every document has 20 exact duplicates: flatters the filter)
    distinct = not any(kv == "distinct=0" for kv in sys.argv[2:])
    term_sets = [synth.random_sparse_corpus_device(min(block, docs), V, seed=4 + 1000 * j, device=dev) for j in range(3 if distinct else 1)]
    t_gen = time.perf_counter() - t0
    ix = SparseIndex(docs, V, device=0)
    done, b = 0, 0
    t0 = time.perf_counter()
    while done < docs:
        indptr, terms, w0 = term_sets[b % len(term_sets)]
        m = min(len(indptr) - 1, docs - done)
        nnz_b = int(indptr[m])
        if b < len(term_sets) or not distinct:
            w = w0[:nnz_b]
        else:
            gw = torch.Generator(device=dev).manual_seed(40_000 + b)
            w = torch.log1p(torch.empty(nnz_b, device=dev).exponential_(1.0, generator=gw)).half().clamp_(min=0.01).cpu().numpy()
        ix.upload((indptr[:m + 1], terms[:nnz_b], w))
        done += m
        b += 1
    ix.finalize()
    t_up = time.perf_counter() - t0
    qp, qt, qw = synth.random_sparse_corpus_fast(256, V, seed=5, mean_nnz=24, lo=4, hi=64)
    q = synth.csr_to_dense(qp, qt, qw, V).astype(np.float16)
    for val in sweep[1]:
        if val is not None:
            _lib.set_option(sweep[0], val)
            print(sweep[0], '=', val)
        ix.search(q[:64], 50)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            ix.search(q, 50)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, ix.counters())
        dt, c = best
        gbps = c["algorithmic_bytes"] / (c["scan_ms"] * 1e-3) / 1e9
        power = None
        if any(kv == "power=1" for kv in sys.argv[2:]):  # board power / shader clock while the search repeats for ~3 s
            import bench
            smp = bench.PowerSampler()
            smp.start()
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < 3.0:
                ix.search(q, 50)
            power = smp.stop(skip_s=0.5)
            power["cap_watts"] = smp.cap_watts()
        print({"power": power, "docs": docs, "nnz": int(ix.nnz), "gen_s": round(t_gen, 2), "upload_s": round(t_up, 2), "queries_per_s": round(256 / dt, 1),
               "scan_ms_per_pass": round(c["scan_ms"] / c["n_passes"], 3), "passes": c["n_passes"], "GB_per_s": round(gbps, 1), "frac": round(gbps / 8000, 4),
               "wall_ms": round(dt * 1e3, 2), "scan_ms": round(c["scan_ms"], 2), "merge_ms": round(c["merge_ms"], 2)})
    ix.close()


if __name__ == "__main__":
    main()
