#!/usr/bin/env python
"""
bench.py — BASELINE.json's headline metric on MI355X: exact inner-product top-50 search of the
kilt_nq-dev-sized query set (2 837 queries) over a KILT-100w-sized corpus (21 M x 768 fp16),
whole-job queries/s with the index already resident in HBM, plus the fraction of the HBM roofline
of the dominant kernel and the reference's own CPU path timed beside it.

  python bench.py --gpus 1 --steps 5 --warmup 1            (driver: N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (driver: N>1)

A "step" = one pass of the hot path over the whole query batch: queries (device resident) ->
per-rank fused scan + merge/rescore over the rank's row shard -> [N>1: one RCCL all-gather of the
partial top-k lists + canonical merge on rank 0].  The corpus is row-sharded across ranks
(SURVEY §8e), so total work is fixed as N grows: scaling = "strong".

Synthetic inputs: corpus rows ~ N(0, I) generated on the device in 1M-row blocks
(block b <- seed 1000+b), L2-normalised, cast to fp16; queries seed 2; for each query 5 positives
normalize(q + 0.3*noise) planted at rows drawn with seed 3.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` and
`cpu_baseline` objects.  oracle/ is imported by the `cpu_baseline` leg only; the parity gates here recompute canonical
scores with numpy (parity proper lives in tests/).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (no 2:1 sparsity), same guide


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--n-rows", type=int, default=21_000_000, help="corpus rows (whole job)")
    p.add_argument("--queries", type=int, default=2837, help="kilt_nq dev size")
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--k", type=int, default=50)
    p.add_argument("--query-tile", type=int, default=None, help="128 or 256 (library default if omitted)")
    p.add_argument("--query-split", type=int, default=None, help="1 or 2 (paired workgroups share the corpus stream)")
    p.add_argument("--pair-window", type=int, default=None, help="query_split 2: max tiles a workgroup runs ahead of its partner")
    p.add_argument("--share-threshold", type=int, default=None)
    p.add_argument("--nontemporal", type=int, default=None)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--full-list-queries", type=int, default=32,
                   help="queries whose COMPLETE top-k lists the parity gate recomputes over the whole corpus, spread over the first, a middle, the last full and the tail pass (N = 1; 0 = off)")
    p.add_argument("--no-other-kernels", action="store_true", help="skip the secondary figures of the earlier scan kernels")
    p.add_argument("--no-larger-k", action="store_true", help="skip the k = 100 / 200 / 1000 legs at the headline geometry")
    p.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] geometry (d = 1024, top-200) search leg")
    p.add_argument("--no-real-size", action="store_true", help="skip the headline search at KILT-100w's real row count (24 853 637 rows)")
    p.add_argument("--no-certificate-leg", action="store_true", help="skip the certificate / fall-back leg on the clustered, non-unit-norm corpus")
    p.add_argument("--encode-stage-passages", type=int, default=32768, help="passages of the Retrieve.encode_and_save leg (0 = skip)")
    p.add_argument("--no-stage", action="store_true", help="skip the Retrieve.retrieve-level leg at the headline size (resident index -> doc-id strings)")
    p.add_argument("--folder-stage", action="store_true",
                   help="also run Retrieve.retrieve on a reference-layout folder of --stage-rows documents (first call loads the folder); off by "
                        "default: its small launches of the headline kernel would blur the kernel's average in a rocprofv3 --stats run")
    p.add_argument("--stage-rows", type=int, default=2_100_000, help="documents of the Retrieve.retrieve-level leg")
    p.add_argument("--cpu-sample-rows", type=int, default=3_150_000,
                   help="rows of the cpu_baseline sample (scaled linearly to --n-rows): ~5 s per pass on the 16 CPUs of a GPU pod, best of 3")
    p.add_argument("--cpu-sample-queries", type=int, default=1000)
    p.add_argument("--no-encoder", action="store_true", help="skip the passages-encoded/s leg")
    p.add_argument("--no-power-leg", action="store_true", help="skip the board power / shader clock sampling behind the timed region")
    p.add_argument("--enc-batch", type=int, default=512, help="sequences per encoder step (retromae.yaml batch_size)")
    p.add_argument("--enc-steps", type=int, default=10, help="timed forward passes of every encoder leg (after one warm-up)")
    p.add_argument("--no-splade", action="store_true", help="skip the SPLADE legs (configs[3]: MLM-head encode, sparse search)")
    p.add_argument("--splade-docs", type=int, default=21_000_000, help="documents of the synthetic SPLADE corpus (S4: 21 M, ~180 terms each)")
    p.add_argument("--splade-queries", type=int, default=2837, help="queries of the SPLADE search leg (kilt_nq dev size; 64 per tile pass)")
    p.add_argument("--splade-gate-queries", type=int, default=16, help="queries whose complete top-k lists the SPLADE leg recomputes over the whole corpus")
    p.add_argument("--splade-term-seeds", type=int, default=3,
                   help="independent draws of the S4 term-set recipe behind the SPLADE corpus blocks (every block gets fresh weights)")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="transport of the N > 1 run: nccl (= RCCL over xGMI, one GPU per rank: the driver's run) or gloo — a TEST mode for "
                        "boxes with fewer GPUs than ranks: ranks share GPUs (LOCAL_RANK modulo the device count) and the partial lists are "
                        "gathered through host buffers; every other line of the run is the same")
    p.add_argument("--sweep", action="store_true", help="also time kernel variants (written to gpurun_out/sweep.json)")
    p.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"),
                   help="optional PMC-derived HBM bytes per scan launch (written by profiles/collect_pmc.py)")
    return p.parse_args()


def make_queries(nq, dim, device):
    g = torch.Generator(device=device).manual_seed(2)
    q = torch.randn(nq, dim, generator=g, device=device)
    return torch.nn.functional.normalize(q, dim=1).half()


BLOCK = 1_000_000


def plant_table(nq, n_total):
    gp = torch.Generator().manual_seed(3)
    return torch.randint(0, n_total, (nq, 5), generator=gp)  # global rows, same on every rank


def corpus_block(b, dim, queries, plant_rows, n_total, device, scaled=False):
    """Block b (rows [b * BLOCK, ...)) of the synthetic corpus as fp16 device rows, plus the (query, j) plants inside it —
    the one recipe behind the index fill, the re-scoring gate and the streaming full-list gate.  scaled: every row is
    stretched by its own factor in [0.25, 4.25) (a cosine index has to undo that at finalize)."""
    b0 = b * BLOCK
    m = min(BLOCK, n_total - b0)
    g = torch.Generator(device=device).manual_seed(1000 + b)
    rows = torch.nn.functional.normalize(torch.randn(m, dim, generator=g, device=device), dim=1)
    sel = ((plant_rows >= b0) & (plant_rows < b0 + m)).nonzero().tolist()
    for qi, j in sel:  # (in this order: a later plant on the same row wins)
        r = int(plant_rows[qi, j])
        gn = torch.Generator(device=device).manual_seed(7_000_000 + qi * 5 + j)
        noise = torch.randn(dim, generator=gn, device=device) * (0.3 / dim ** 0.5)
        base = queries[qi].float()
        if scaled:  # (the cosine legs' queries are not unit length; the headline recipe is unchanged)
            base = torch.nn.functional.normalize(base, dim=0)
        rows[r - b0] = torch.nn.functional.normalize(base + noise, dim=0)
    if scaled:
        rows = rows * (0.25 + 4.0 * torch.rand(m, 1, generator=g, device=device))
    return rows.half(), sel


def canonical_unit_rows(x):
    """The library's cosine normalisation (bergen_amd/csrc/convert.hip, restating CosineSim of reference dense.py:87-88) in plain
    torch IEEE operations, no kernel of this repository: n2 = sum of x_j^2 in fp64 (exact in any order: fp16 squares are
    multiples of 2^-48 and the sum stays far below 2^5, so no addition rounds), inv = 1 / sqrt(n2) (two correctly rounded fp64
    operations), y_j = fp16(fp32(fp64(x_j) * inv)) — product rounded to fp64, then to fp32, then to fp16, each to nearest even.
    Zero rows stay zero."""
    x64 = x.double()
    n2 = (x64 * x64).sum(dim=1, keepdim=True)
    inv = torch.where(n2 > 0, 1.0 / torch.sqrt(n2), torch.zeros_like(n2))
    return (x64 * inv).float().half()


def fill_shard(ix, lo, hi, dim, queries, n_total, device, scaled=False):
    """Upload rows [lo, hi) of the synthetic corpus into ix (local row = global row - lo)."""
    nq = queries.shape[0]
    plant_rows = plant_table(nq, n_total)
    planted = []
    b_first, b_last = lo // BLOCK, (max(hi, lo + 1) - 1) // BLOCK
    for b in range(b_first, b_last + 1):
        b0 = b * BLOCK
        if n_total - b0 <= 0:
            break
        rows, sel = corpus_block(b, dim, queries, plant_rows, n_total, device, scaled=scaled)
        a, e = max(lo, b0), min(hi, b0 + rows.shape[0])
        if e > a:
            ix.upload(rows[a - b0:e - b0].contiguous(), row0=a - lo)
            for qi, j in sel:
                r = int(plant_rows[qi, j])
                if a <= r < e:
                    planted.append((qi, r))
        del rows
    return planted, plant_rows


def streamed_full_lists(check, queries, dim, k, plant_rows, n_total, device, metric="ip", scaled=False):
    """The COMPLETE top-k lists of the queries `check` (an int n = the first n, or a list of query indices), computed without
    any of this repository's kernels: the corpus is regenerated block by block, every block is scored with a float64 matrix
    product (torch / rocBLAS), rounded to fp32, and the rows that can still be in the top k are kept; the union is ordered
    (score desc, row asc) at the end.  metric "cos": queries and rows first go through canonical_unit_rows (plain torch).
    Why a float64 GEMM in ANY summation order gives the canonical score (= fp32 of the sequential fp64 sum) here: fp16
    values are multiples of 2^-24, so every product and every partial sum is a multiple of 2^-48; with
    sum_j |q_j x_j| <= |q| |x| < 16 (checked below; the rows are unit-norm) every partial sum has fewer than 53
    significant bits — no fp64 operation rounds, whatever its order."""
    idx = list(range(check)) if isinstance(check, int) else [int(c) for c in check]
    n_check = len(idx)
    qsel = queries[torch.as_tensor(idx, device=queries.device)]
    if metric == "cos":
        qsel = canonical_unit_rows(qsel)
    q64 = qsel.double()
    q_norm = float(q64.norm(dim=1).max())
    keep_s, keep_i = [[] for _ in range(n_check)], [[] for _ in range(n_check)]
    for b in range((n_total + BLOCK - 1) // BLOCK):
        rows, _ = corpus_block(b, dim, queries, plant_rows, n_total, device, scaled=scaled)
        if metric == "cos":
            rows = canonical_unit_rows(rows)
        x64 = rows.double()
        assert q_norm * float(x64.norm(dim=1).max()) < 16.0, "exactness argument of the float64 gate does not hold"
        sc = (q64 @ x64.T).float()  # [n_check, m], fp32 (RNE) of the exact sum
        kth = torch.topk(sc, min(k, sc.shape[1]), dim=1).values[:, -1:]
        hit = (sc >= kth).nonzero()  # every row that ties the block's k-th stays in
        vals = sc[hit[:, 0], hit[:, 1]].cpu().tolist()
        for (a, r), v in zip(hit.cpu().tolist(), vals):
            keep_s[a].append(v)
            keep_i[a].append(b * BLOCK + r)
        del rows, x64, sc
    want_s = np.empty((n_check, k), np.float32)
    want_i = np.empty((n_check, k), np.int64)
    for a in range(n_check):
        cs, ci = np.asarray(keep_s[a], np.float32), np.asarray(keep_i[a], np.int64)
        order = np.lexsort((ci, -cs.astype(np.float64)))[:k]
        want_s[a], want_i[a] = cs[order], ci[order]
    return want_s, want_i


def gate_queries(nq, counters, n_check):
    """Query indices for the full-list gate, spread over the passes of the search that `counters` describes: the first pass, a
    middle one, the last full pass of the main kernel and the LAST pass (the 128-query tail kernel when the search had one) —
    both ends of each tile.  With paired launches (counter paired_launches) passes 2j and 2j + 1 share a launch, on different
    halves of the grid: the second half of the first and of the last paired launch is covered as well."""
    tile = int(counters["query_tile"])
    n_pass = int(counters["n_passes"])
    n_pair = int(counters.get("paired_launches", 0))
    starts = {0, (n_pass // 2) * tile, max(0, n_pass - 2) * tile, (n_pass - 1) * tile}
    if n_pair > 0:
        starts |= {tile, (2 * n_pair - 1) * tile}
    starts = sorted(starts)
    starts = [s0 for s0 in starts if s0 < nq]
    per = max(1, n_check // len(starts))
    idx = []
    for s0 in starts:
        e0 = min(nq, s0 + tile)
        span = list(range(s0, min(e0, s0 + (per + 1) // 2))) + list(range(max(s0, e0 - per // 2), e0))
        idx += span
    return sorted(set(idx))[:max(n_check, len(starts))]


def full_list_gate_fn(check, res_s, res_i, queries, dim, k, plant_rows, n_total, device, metric="ip", scaled=False):
    """-> (ok, record): the search's lists of the queries `check` against streamed_full_lists, ids and fp32 scores bit for bit."""
    t0 = time.perf_counter()
    full_s, full_i = streamed_full_lists(check, queries, dim, k, plant_rows, n_total, device, metric=metric, scaled=scaled)
    ok = bool(np.array_equal(full_i, res_i[check]) and np.array_equal(full_s.view(np.uint32), res_s[check].view(np.uint32)))
    return ok, {"queries": len(check), "query_indices": [int(c) for c in check], "rows": n_total,
                "ids_and_fp32_scores_bit_exact": ok, "seconds": time.perf_counter() - t0,
                "how": ("float64 GEMM per 1M-row block (exact for unit-norm fp16 data), fp32 round, (score desc, row asc)"
                        + ("; rows and queries normalised by plain torch fp64 / fp32 / fp16 operations first" if metric == "cos" else ""))}


def cpu_baseline(args, dim, k):
    """The reference's op sequence (torch.mm + torch.topk per chunk + host merge, fp32) on the host cores,
    on a bounded sample, scaled linearly in N to the full corpus."""
    from oracle import ref_port
    from bergen_amd.utils import cpu_budget
    cores = cpu_budget()  # the CPUs this container may keep busy (affinity mask cut to the cgroup quota), not the host's count
    torch.set_num_threads(cores)
    nq, n = args.cpu_sample_queries, args.cpu_sample_rows
    g = torch.Generator().manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g), dim=1).half().float()
    chunks = []
    for c in ref_port.reference_chunk_sizes(n, batch_size=512):
        chunks.append(torch.nn.functional.normalize(torch.randn(c, dim, generator=g), dim=1).half().float())
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        ref_port.retrieve(q, chunks, k, batch_size_sim=2048)
        best = min(best, time.perf_counter() - t0)
    scale = args.n_rows / n
    qps_full = nq / (best * scale)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {
        # `cores` = the threads torch was given: the container's CPU budget (bergen_amd.utils.cpu_budget: logical CPUs in the
        # affinity mask, cut to the cgroup quota — 16 of the 256 the MI355X boxes show); "host_logical_cpus" = os.cpu_count()
        "value": qps_full, "unit": "queries/s", "cores": cores, "cores_are": "logical CPUs the container may use (cgroup quota) = torch intra-op threads",
        "host_logical_cpus": os.cpu_count(), "kind": "port",
        "sample": (f"oracle/ref_port.py (the reference's torch.mm+torch.topk chunk loop, fp32, batch_size_sim=2048, "
                   f"chunks 150016/149504 rows) on Q={nq} x N={n} x d={dim}, best of 3 = {best:.3f} s, "
                   f"scaled x{scale:.1f} linearly in N to N={args.n_rows}; CPU: {model}; torch {torch.__version__}"),
        "measured_seconds": best,
    }


def encoder_leg(args, device_index, arch="bert"):
    """passages-encoded/s: the bi-encoder forward pass (BASELINE configs[1] encoder = RetroMAE = BERT-base
    architecture, CLS pooling) on a batch of synthetic passages (lengths ~ clipped-Normal(130, 30)
    in [16, 256], random token ids, seeded random-init weights — no checkpoint exists offline).  A step = one
    forward pass of --enc-batch passages, token ids on the host (as a tokenizer leaves them), embeddings left
    in HBM (where the index consumes them).  Roofline = MFMA: algorithmic flops over REAL tokens
    (12 x (T x 14.16 MFLOP + 4 d sum len^2)) / forward time / 2.5 PFLOP/s.
    arch="nomic": the same batch through nomic-embed-text-v1.5's architecture (config/retriever/nomic-embed-text-v1.5.yaml:
    NomicBert 12 x 768 x 12 heads, rotary positions, gated SiLU feed-forward of 3072 — 18.9 MFLOP of projections per token and
    layer instead of 14.2 —, mean pooling); returned under "nomic_encode"."""
    from bergen_amd import BertEncoder, synth
    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    if arch == "nomic":
        cfg.update(model_type="nomic_bert", hidden_act="silu", vocab_size=30528, max_position_embeddings=2048, rope_theta=1000.0)
        sd = synth.random_nomic(cfg, seed=33, scale=0.02)
    elif arch == "gte":
        # Alibaba-NLP/gte-base-en-v1.5 (config/retriever/gte-base-en-v1.5.yaml): the remote "new" class — NTK-scaled rotary positions,
        # packed biased q | k | v, GELU-gated feed-forward (one [2 d_ff] GEMM folded in its epilogue), CLS pooling
        cfg = dict(model_type="new", vocab_size=30528, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                   hidden_act="gelu", max_position_embeddings=8192, type_vocab_size=0, layer_norm_type="layer_norm", layer_norm_eps=1e-12,
                   position_embedding_type="rope", rope_theta=500000.0, rope_scaling={"type": "ntk", "factor": 2.0})
        sd = synth.random_new(cfg, seed=37, scale=0.02)
    elif arch == "jina":
        # jinaai/jina-embeddings-v2-base-en (config/retriever/jina-embeddings-v2-base-en.yaml): remote JinaBert — symmetric ALiBi attention
        # biases, GELU-gated feed-forward, mean pooling
        cfg = dict(model_type="bert", vocab_size=30528, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                   hidden_act="gelu", max_position_embeddings=8192, type_vocab_size=2, layer_norm_eps=1e-12, position_embedding_type="alibi",
                   feed_forward_type="geglu")
        sd = synth.random_jina(cfg, seed=39, scale=0.02)
    elif arch == "e5_large":
        # BASELINE configs[4]'s encoder (config/retriever/e5-large-v2.yaml:1-10): bert-large shape, MeanPooler,
        # batch_size 512, max_len 256 — the same synthetic passages as the BERT-base leg
        cfg.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
        sd = synth.random_bert(cfg, seed=35)
    else:
        sd = synth.random_bert(cfg, seed=31)
    pooler = "cls" if arch in ("bert", "gte") else "mean"
    enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=device_index)
    rng = np.random.default_rng(6)
    lens = np.clip(np.rint(rng.normal(130, 30, size=args.enc_batch)), 16, 256).astype(np.int64)
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, cfg["vocab_size"], size=(args.enc_batch, T)).astype(np.int64) * mask
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    enc.encode_pooled(kw, pooler)  # warm-up (workspace allocation, kernel attribute setup)
    torch.cuda.synchronize()
    fwd_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.enc_steps):
        emb = enc.encode_pooled(kw, pooler)
        fwd_ms += enc.counters()["forward_ms"]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c = enc.counters()
    ok = bool(torch.isfinite(emb.float()).all()) and tuple(emb.shape) == (args.enc_batch, cfg["hidden_size"])
    enc.close()
    achieved = c["flops"] / (fwd_ms / args.enc_steps * 1e-3) / 1e12
    if arch == "e5_large":
        return {"e5_large_encode": {
            "workload": f"configs[4] encoder: bert-large shape (24x1024x16 heads, d_ff 4096) forward + mean pool (e5-large-v2.yaml), "
                        f"{args.enc_batch} synthetic passages/step, {int(c['real_tokens'])} real tokens, fp16 storage / fp32 accumulate, "
                        f"random-init weights",
            "passages_per_s": args.enc_batch * args.enc_steps / wall, "steps": args.enc_steps, "ms_per_step_kernels": fwd_ms / args.enc_steps,
            "packed_rows": int(c["packed_rows"]), "finite_and_shaped": ok,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "algorithmic_flops_per_step": c["flops"]}}}
    if arch in ("gte", "jina"):
        what = ("the 'new' class of gte-base-en-v1.5 (12x768x12 heads, NTK rotary, GELU-gated d_ff 3072) forward + CLS pool" if arch == "gte" else
                "JinaBert of jina-embeddings-v2-base-en (12x768x12 heads, ALiBi, GELU-gated d_ff 3072) forward + mean pool")
        return {arch + "_encode": {
            "workload": f"{what}, {args.enc_batch} synthetic passages/step, {int(c['real_tokens'])} real tokens, fp16 storage / fp32 accumulate, "
                        f"random-init weights (remote architectures: parity unpinned offline, conversions self-checked — oracle/new_oracle.py)",
            "passages_per_s": args.enc_batch * args.enc_steps / wall, "ms_per_step_kernels": fwd_ms / args.enc_steps, "finite_and_shaped": ok,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "algorithmic_flops_per_step": c["flops"]}}}
    if arch == "nomic":
        return {"nomic_encode": {
            "workload": f"NomicBert (12x768x12 heads, rotary positions, gated SiLU d_ff 3072) forward + mean pool, {args.enc_batch} synthetic "
                        f"passages/step, {int(c['real_tokens'])} real tokens, fp16 storage / fp32 accumulate, random-init weights",
            "passages_per_s": args.enc_batch * args.enc_steps / wall, "ms_per_step_kernels": fwd_ms / args.enc_steps,
            "finite_and_shaped": ok,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "algorithmic_flops_per_step": c["flops"]}}}
    return {
        "passages_per_s": args.enc_batch * args.enc_steps / wall,
        "encoder": {"workload": f"BERT-base (12x768x12 heads, d_ff 3072) forward + CLS pool, {args.enc_batch} synthetic "
                                f"passages/step, {int(c['real_tokens'])} real tokens (reference pads to {ids.size}), fp16 "
                                f"storage / fp32 accumulate, random-init weights",
                    "steps": args.enc_steps, "ms_per_step_wall": wall / args.enc_steps * 1e3,
                    "ms_per_step_kernels": fwd_ms / args.enc_steps, "packed_rows": int(c["packed_rows"]),
                    "finite_and_shaped": ok},
        "encoder_roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": None,
                             "algorithmic_flops_per_step": c["flops"],
                             # the matrix pipe's busy share inside the three GEMM instantiations of a BERT layer (SQ counters of the
                             # encoder-only command, profiles/sq_counters.json): FFN-up (bias + GELU), the bias-per-column projections
                             # (Q|K, attention output, FFN-down), the blocked V^T projection
                             "mfma_busy_frac": {key: (lambda b: b["mfma_busy_frac"] if b else None)(
                                 pmc_mfma_busy("bh_gemm_f16_p16kernel", lambda nm, e=epi: nm.replace(" ", "").startswith(f"bh_gemm_f16_p16kernel<{e},")))
                                 for key, epi in (("ffn_up_gelu", 9), ("bias_col_projections", 1), ("vt_projection", 2))}},
    }


def splade_legs(args, device_index):
    """BASELINE configs[3] (SPLADE), both halves, as secondary figures beside the headline (N = 1 only):
    * encode: BERT-base + tied masked-LM head over 30 522 terms + max-over-tokens pooling on the encoder leg's batch
      (`BertEncoder.encode_splade`; the [B, T, vocab] logits are never materialised);
    * search: SURVEY §8d S4 as stated — synthetic CSR corpus of 21 M documents (V = 30 522, Poisson(180) terms per document
      clipped to [16, 400], Zipf(1.1) term ids drawn without replacement; 21 distinct 1 M-document blocks), 64-query tiles,
      top-k, ALL 2 837 queries of the kilt_nq dev size (45 tile passes); roofline = HBM with algorithmic bytes nnz*4 + (N+1)*8 per
      tile pass.  Parity gate at full size: the complete lists of 16 queries recomputed over all 21 M documents (SparseStreamGate)."""
    from bergen_amd import BertEncoder, SparseIndex, synth
    out = {}
    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = synth.random_bert(cfg, seed=31)
    synth.random_mlm_head(cfg, seed=32, tied=True, sd=sd, bias_mean=-3.0)
    enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=device_index)
    rng = np.random.default_rng(6)
    lens = np.clip(np.rint(rng.normal(130, 30, size=args.enc_batch)), 16, 256).astype(np.int64)
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, cfg["vocab_size"], size=(args.enc_batch, T)).astype(np.int64) * mask
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    enc.encode_splade(kw)
    torch.cuda.synchronize()
    ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.enc_steps):
        emb = enc.encode_splade(kw)
        ms += enc.counters()["forward_ms"]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c = enc.counters()
    out["splade_encode"] = {
        "passages_per_s": args.enc_batch * args.enc_steps / wall, "ms_per_step_kernels": ms / args.enc_steps,
        "tflops": c["flops"] / (ms / args.enc_steps * 1e-3) / 1e12,
        "workload": f"BERT-base + tied MLM head (30522 terms) + SPLADE pooling, {args.enc_batch} synthetic passages/step, "
                    f"{int(c['real_tokens'])} real tokens, random-init weights",
        "finite_and_shaped": bool(torch.isfinite(emb.float()).all()) and tuple(emb.shape) == (args.enc_batch, 30522)}
    del emb
    enc.close()
    V = 30522
    dev = torch.device("cuda", device_index)
    # 21 DISTINCT 1 M-document blocks (synth.sparse_bench_blocks: term sets from --splade-term-seeds independent draws of the S4
    # recipe used in turn, fresh weights per block — no two documents share their scores; round 2 repeated one block 21 times,
    # which flattered the threshold filter).  tests/test_gpu_sparse.py::test_full_size_sparse iterates the same generator and
    # checks the same search against the oracle; the gate below recomputes complete lists WITHOUT the oracle and without any
    # kernel of this repository while the blocks stream by.
    nq = args.splade_queries
    qp, qt, qw = synth.random_sparse_corpus_fast(nq, V, seed=5, mean_nnz=24, lo=4, hi=64)
    q = synth.csr_to_dense(qp, qt, qw, V).astype(np.float16)
    gate = SparseStreamGate(q, sparse_gate_queries(nq, args.splade_gate_queries), args.k, dev)
    n_blocks = (args.splade_docs + 999_999) // 1_000_000
    ix = SparseIndex(args.splade_docs, V, device=device_index)
    t_gen = time.perf_counter()
    gate_s = 0.0
    for b, row0, indptr, terms, w in synth.sparse_bench_blocks(args.splade_docs, V, dev, term_seeds=args.splade_term_seeds):
        ix.upload((indptr, terms, w))
        t0 = time.perf_counter()
        gate.add_block(row0, indptr, terms, w)
        gate_s += time.perf_counter() - t0
    build_s = time.perf_counter() - t_gen - gate_s
    ix.finalize()
    ix.search(q[:64], args.k)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        res_s, res_i = ix.search(q, args.k)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, ix.counters())
    dt, c = best
    gbps = c["algorithmic_bytes"] / (c["scan_ms"] * 1e-3) / 1e9
    t0 = time.perf_counter()
    want_s, want_i = gate.result()
    gate_s += time.perf_counter() - t0
    gi = gate.queries
    exact = bool(np.array_equal(res_i[gi], want_i) and np.array_equal(res_s[gi].view(np.uint32), want_s.view(np.uint32)))
    d_s, d_i = np.diff(res_s, axis=1), np.diff(res_i, axis=1)
    ordered = bool((d_s <= 0).all() and np.all((d_s < 0) | (d_i > 0)))
    out["splade_search"] = {
        "queries_per_s": nq / dt, "queries": nq, "docs": args.splade_docs, "nnz": int(ix.nnz), "passes": int(c["n_passes"]),
        "scan_ms_per_pass": c["scan_ms"] / c["n_passes"], "search_ms": dt * 1e3, "scan_ms": c["scan_ms"],
        "corpus": f"{n_blocks} distinct 1M-document blocks: term sets from {min(args.splade_term_seeds, n_blocks)} independent draws "
                  f"of SURVEY §8d S4, fresh weights per block ({build_s:.1f} s to draw and upload)",
        "roofline": {"bound": "hbm", "kernel": "bh_csr_scan_mfma_kernel", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": gbps / HBM_PEAK_GBPS, "traffic": pmc_traffic(args.traffic_json, "bh_csr_scan_mfma_kernel", args.splade_docs, V),
                     "algorithmic_bytes_per_launch": c["algorithmic_bytes"] / c["n_passes"]},
        "full_list_gate": {"queries": len(gi), "query_indices": [int(x) for x in gi], "docs": args.splade_docs,
                           "ids_and_fp32_scores_bit_exact": exact, "canonical_order_all_queries": ordered, "seconds": gate_s,
                           "candidates_per_block": gate.width, "how": SparseStreamGate.HOW},
        "parity_check": "pass" if exact and ordered else "FAIL"}
    ix.close()
    return out


def sparse_gate_queries(nq, n_check=16):
    """Queries whose complete lists the SPLADE gate recomputes: from the first, a middle and the last 64-query tile, the first
    and the last query of each included (tests/test_gpu_sparse.py picks the same way)."""
    n_tiles = -(-nq // 64)
    tiles = sorted({0, n_tiles // 2, n_tiles - 1})
    share = [n_check // len(tiles) + (1 if j < n_check % len(tiles) else 0) for j in range(len(tiles))]
    picks = set()
    for t, cnt in zip(tiles, share):
        lo, hi = 64 * t, min(nq, 64 * t + 64)
        picks.update(np.linspace(lo, hi - 1, max(cnt, 2)).astype(int).tolist())
    return sorted(picks)


class SparseStreamGate:
    """Complete canonical top-k lists of a few queries over a CSR corpus that streams by block by block — recomputed with plain
    torch / numpy operations, no kernel of this repository and no oracle (parity proper, against oracle/sparse_oracle.c, is
    tests/test_gpu_sparse.py::test_full_size_sparse on the same corpus).
    Per block and query: every document's score as a float64 index_add_ of q[term] * weight (any summation order: within 1e-9
    of the canonical value), the block's `width` best by that score are the candidates; their CANONICAL scores — fp32 of the
    float64 sum in increasing term order (include/bergen_hip.h: bh_sparse_search) — are recomputed sequentially on the host.  A
    member of the global top k is a member of its block's top k, and it is among the candidates unless the unordered sum moved
    it by more than `width - k` ranks: checked (the last candidate's score must lie below the k-th canonical score by more than
    the error bound, or the block is re-done four times as wide)."""
    HOW = ("float64 index_add_ of q[term] * weight per 1M-document block (torch, on the device) -> candidates per query and block; "
           "canonical scores of the candidates = fp32 of the float64 sum in term order (numpy cumsum), (score desc, row asc), cut k")

    def __init__(self, q_dense_f16, queries, k, device, width=None):
        self.queries = list(queries)
        self.k = int(k)
        self.device = device
        self.width = int(width) if width else self.k + 14
        self.qd_host = q_dense_f16[self.queries].astype(np.float64)
        self.qd = torch.from_numpy(self.qd_host).to(device)
        self.cands = [[] for _ in self.queries]       # per query: (canonical fp32 score, global row)

    def add_block(self, row0, indptr, terms, w):
        m = len(indptr) - 1
        if m == 0:
            return
        dev = self.device
        lens = torch.from_numpy(np.diff(indptr)).to(dev)
        doc = torch.repeat_interleave(torch.arange(m, device=dev), lens)
        t_d = torch.from_numpy(terms).to(dev).long()
        w_d = torch.from_numpy(w).to(dev).double()
        for a in range(len(self.queries)):
            approx = torch.zeros(m, dtype=torch.float64, device=dev).index_add_(0, doc, self.qd[a][t_d] * w_d)
            width = min(self.width, m)
            while True:
                top = torch.topk(approx, width)
                rows = top.indices.cpu().numpy()
                exact = np.empty(width, np.float32)
                for j, r in enumerate(rows):
                    lo, hi = indptr[r], indptr[r + 1]
                    acc = np.cumsum(self.qd_host[a, terms[lo:hi]] * w[lo:hi].astype(np.float64))
                    exact[j] = np.float32(acc[-1] if hi > lo else 0.0)
                if width == m or width <= self.k:
                    break
                kth = np.sort(exact)[::-1][self.k - 1]
                if float(top.values[-1]) + 1e-6 < float(kth):
                    break
                if width >= 16384:  # (a block where thousands of documents tie with the k-th score: not a corpus this gate is for)
                    raise RuntimeError(f"SparseStreamGate: {width} candidates do not separate the top {self.k} of query {self.queries[a]}")
                width = min(m, width * 4)
                self.width = max(self.width, width)
            self.cands[a].append((exact, rows.astype(np.int64) + row0))
        del doc, t_d, w_d, lens

    def result(self):
        out_s = np.full((len(self.queries), self.k), -np.inf, np.float32)
        out_i = np.full((len(self.queries), self.k), -1, np.int64)
        for a, parts in enumerate(self.cands):
            s = np.concatenate([p[0] for p in parts])
            r = np.concatenate([p[1] for p in parts])
            order = np.lexsort((r, -s.astype(np.float64)))[:self.k]
            out_s[a, :len(order)] = s[order]
            out_i[a, :len(order)] = r[order]
        return out_s, out_i


def retrieve_stage_leg(args, device_index):
    """The whole stage as RAG.retrieve calls it (reference modules/rag.py:322-330 -> modules/retrieve.py:52-108):
    `Retrieve.retrieve(dataset, query_folder, doc_folder, k)` on index folders in the reference's layout — read the
    query embeddings, bring the document folder into HBM (first call) or find it resident (second call), ONE search for
    the whole query set, map the hit rows to doc-id strings.  A tenth of the headline corpus (the folders are written
    here, outside the timed calls; the encode half is the encoder leg's business)."""
    import shutil
    import tempfile

    import datasets

    import bergen_amd
    n, nq, dim, k = args.stage_rows, args.queries, args.dim, args.k
    root = tempfile.mkdtemp(prefix="bergen_stage_")
    try:
        q_path, d_path = os.path.join(root, "queries"), os.path.join(root, "docs")
        os.makedirs(q_path)
        os.makedirs(d_path)
        g = torch.Generator().manual_seed(11)
        block = torch.nn.functional.normalize(torch.randn(150_000, dim, generator=g), dim=1).half()
        done, i = 0, 0
        while done < n:  # chunk files named after their last batch, 150 k rows each (the block repeated: sizes matter)
            m = min(150_000, n - done)
            i += m // 512 + 1
            torch.save(block[:m].clone(), os.path.join(d_path, f"embedding_chunk_{i}.pt"))
            done += m
        torch.save(torch.nn.functional.normalize(torch.randn(nq, dim, generator=g), dim=1).half(),
                   os.path.join(q_path, "embedding_chunk_0.pt"))
        dataset = {"doc": datasets.Dataset.from_dict({"id": [f"d{j}" for j in range(n)]}),
                   "query": datasets.Dataset.from_dict({"id": [f"q{j}" for j in range(nq)]})}

        class _Plug:  # the folders exist: the model is only asked for its name and similarity
            model_name = "bench/precomputed"
            similarity = bergen_amd.DotProduct()
            model = torch.nn.Identity()

        r = bergen_amd.Retrieve(init_args=_Plug(), batch_size=512, batch_size_sim=2048, num_workers=0, device=device_index)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = r.retrieve(dataset, q_path, d_path, k)
        cold = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = r.retrieve(dataset, q_path, d_path, k)
        warm = time.perf_counter() - t0
        ok = (tuple(out["score"].shape) == (nq, k) and len(out["doc_id"]) == nq and isinstance(out["doc_id"][0][0], str)
              and bool((out["score"][:, :-1] >= out["score"][:, 1:]).all()))
        r.close()
        return {"workload": f"Retrieve.retrieve: {nq} queries, {n} x {dim} fp16 documents in {len(os.listdir(d_path))} chunk files, top-{k}",
                "first_call_seconds": cold, "first_call_queries_per_s": nq / cold,
                "resident_call_seconds": warm, "resident_call_queries_per_s": nq / warm, "shaped_and_sorted": ok}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def config5_leg(args, local_rank, device):
    """BASELINE configs[4] geometry on ONE GPU: e5-large-v2-sized vectors (d = 1024) under the similarity that model's yaml
    names — CosineSim (config/retriever/e5-large-v2.yaml:5-10; the index normalises every row once at finalize) —, top-200, a
    21 M-row synthetic datastore (the stated configuration shards it over 8 GPUs).  The rows and queries are NOT
    unit length (each stretched by its own factor), so the finalize-time normalisation does real work at full size.  Parity gate
    as for the headline: planted positives on top, sorted lists, and the COMPLETE top-200 lists of queries from every pass
    against the kernel-free float64 recomputation (streamed_full_lists, metric "cos")."""
    import bergen_amd
    dim, k, nq, n = 1024, 200, 1000, args.n_rows
    gq = torch.Generator(device=device).manual_seed(12)
    queries = (make_queries(nq, dim, device).float() * (0.5 + 3.0 * torch.rand(nq, 1, generator=gq, device=device))).half()
    ix = bergen_amd.FlatIndex(n, dim, metric="cos", device=local_rank)
    planted, plant_rows = fill_shard(ix, 0, n, dim, queries, n, device, scaled=True)
    t0 = time.perf_counter()
    ix.finalize()
    torch.cuda.synchronize()
    finalize_s = time.perf_counter() - t0
    res = ix.search(queries, k)
    torch.cuda.synchronize()
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    scan_ms = pair_ms = tail_ms = bal_ms = 0.0
    for _ in range(steps):
        res = ix.search(queries, k)
        host = (torch.as_tensor(res[0]).cpu(), torch.as_tensor(res[1]).cpu())
        cc = ix.counters()
        scan_ms += cc["scan_ms"]
        pair_ms += cc.get("paired_scan_ms", 0.0)
        bal_ms += cc.get("balanced_scan_ms", 0.0)
        tail_ms += cc.get("tail_scan_ms", 0.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    c = ix.counters()
    i_np = host[1].numpy()
    s_np = host[0].numpy()
    ok = bool((np.diff(s_np, axis=1) <= 0).all())
    owner = {}
    for qi in range(nq):
        for j in range(5):
            owner[int(plant_rows[qi, j])] = qi
    for qi in range(nq):
        mine = set(r for r in (int(v) for v in plant_rows[qi].tolist()) if owner[r] == qi)
        ok &= set(i_np[qi, :len(mine)].tolist()) == mine
    gate = None
    if args.full_list_queries > 0:
        check = gate_queries(nq, c, min(16, args.full_list_queries))
        gate_ok, gate = full_list_gate_fn(check, s_np, i_np, queries, dim, k, plant_rows, n, device, metric="cos", scaled=True)
        ok &= gate_ok
    roof = scan_roofline({"scan_ms": scan_ms, "tail_scan_ms": tail_ms, "paired_scan_ms": pair_ms, "balanced_scan_ms": bal_ms}, c, steps, n, dim, k, args.traffic_json)
    ix.close()
    return {"workload": f"configs[4] geometry: {nq} queries x {n} x {dim} fp16, cosine (rows normalised once at finalize), top-{k}, one GPU",
            "queries_per_s": nq / dt, "finalize_seconds": finalize_s,
            "ms_per_step": dt * 1e3, "query_tile": c["query_tile"], "passes_per_step": c["n_passes"], "k_padded": c["k_padded"],
            "roofline": roof,
            "uncertified_queries": c.get("uncertified_queries", 0), "full_list_gate": gate, "parity_check": "pass" if ok else "FAIL"}


def real_size_leg(args, local_rank, device):
    """The headline search at the REAL row count of KILT-100w: 24 853 637 passages (BASELINE's "21M" is nominal) —
    38.2 GB of fp16 rows, same queries / k / data recipe / parity gate as the headline (complete lists of queries from the
    first, a middle, the last full and the tail pass)."""
    import bergen_amd
    dim, k, nq, n = args.dim, args.k, args.queries, 24_853_637
    queries = make_queries(nq, dim, device)
    ix = bergen_amd.FlatIndex(n, dim, metric="ip", device=local_rank)
    planted, plant_rows = fill_shard(ix, 0, n, dim, queries, n, device)
    ix.finalize()
    torch.cuda.synchronize()
    res = ix.search(queries, k, host=True)
    steps = max(1, min(args.steps, 3))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scan_ms = tail_ms = pair_ms = bal_ms = 0.0
    for _ in range(steps):
        res = ix.search(queries, k, host=True)
        cc = ix.counters()
        scan_ms += cc["scan_ms"]
        tail_ms += cc.get("tail_scan_ms", 0.0)
        pair_ms += cc.get("paired_scan_ms", 0.0)
        bal_ms += cc.get("balanced_scan_ms", 0.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    c = ix.counters()
    s_np, i_np = res[0].numpy().copy(), res[1].numpy().copy()
    ok = bool((np.diff(s_np, axis=1) <= 0).all())
    owner = {}
    for qi in range(nq):
        for j in range(5):
            owner[int(plant_rows[qi, j])] = qi
    for qi in range(nq):
        mine = set(r for r in (int(v) for v in plant_rows[qi].tolist()) if owner[r] == qi)
        ok &= set(i_np[qi, :len(mine)].tolist()) == mine
    gate = None
    if args.full_list_queries > 0:
        check = gate_queries(nq, c, min(16, args.full_list_queries))
        gate_ok, gate = full_list_gate_fn(check, s_np, i_np, queries, dim, k, plant_rows, n, device)
        ok &= gate_ok
    roof = scan_roofline({"scan_ms": scan_ms, "tail_scan_ms": tail_ms, "paired_scan_ms": pair_ms, "balanced_scan_ms": bal_ms}, c, steps, n, dim, k, args.traffic_json)
    ix.close()
    return {"workload": f"configs[1] at KILT-100w's real row count: {nq} queries x {n} x {dim} fp16, top-{k}, one GPU",
            "queries_per_s": nq / dt, "ms_per_step": dt * 1e3, "query_tile": c["query_tile"], "passes_per_step": c["n_passes"],
            "roofline": roof,
            "uncertified_queries": c.get("uncertified_queries", 0), "full_list_gate": gate, "parity_check": "pass" if ok else "FAIL"}


def retrieve_stage_full_leg(args, stage, index, queries, want_scores, want_rows, n_total, k):
    """`Retrieve.retrieve` — the call RAG.retrieve makes (modules/rag.py:322-329) — at the HEADLINE size, on the headline's own
    resident index (filled on the device: a 32 GB folder is not written here; the folder-read path is the `retrieve_stage` leg
    and profiles/load_path.py): the real query-folder read (embedding_chunk_0.pt, as the reference writes it), one search
    for the whole query set, the real id mapping over a 21 M-row Arrow id column, the reference's return dict.  What the
    headline `value` leaves out is exactly this call's host side."""
    import shutil
    import tempfile

    import datasets
    import pyarrow as pa
    import pyarrow.compute as pc
    nq, dim = queries.shape
    root = tempfile.mkdtemp(prefix="bergen_stage_full_")
    try:
        q_path, d_path = os.path.join(root, "queries"), os.path.join(root, "docs")
        os.makedirs(q_path)
        os.makedirs(d_path)  # (empty: the stage adopts the resident index for this path)
        torch.save(queries.cpu(), os.path.join(q_path, "embedding_chunk_0.pt"))
        ids = pc.cast(pa.array(np.arange(n_total, dtype=np.int64)), pa.string())  # KILT ids are the stringified row (dataset_processor.py:336)
        dataset = {"doc": datasets.Dataset(datasets.table.InMemoryTable(pa.table({"id": ids}))),
                   "query": datasets.Dataset.from_dict({"id": [f"q{j}" for j in range(nq)]})}
        stage.adopt_resident_index(d_path, index, n_total, "ip", rows=None)
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = stage.retrieve(dataset, q_path, d_path, k)
            times.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        stage._map_doc_ids(dataset["doc"], torch.from_numpy(want_rows))
        map_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        from bergen_amd import utils
        utils.load_embeddings(q_path)
        load_q_s = time.perf_counter() - t0
        ok = (torch.equal(out["score"], torch.from_numpy(want_scores)) and
              all(out["doc_id"][q][j] == str(int(want_rows[q, j])) for q in (0, nq // 2, nq - 1) for j in (0, k - 1)) and
              len(out["doc_id"]) == nq and isinstance(out["doc_id"][0][0], str))
        best = min(times)
        return {"workload": f"Retrieve.retrieve: {nq} queries (folder read) x {n_total} x {dim} resident fp16 documents, top-{k}, "
                            f"doc-id strings from a {n_total}-row Arrow column",
                "seconds": best, "queries_per_s": nq / best, "all_calls_seconds": times,
                "host_tail": {"query_folder_read_s": load_q_s, "id_mapping_s": map_s},
                "same_scores_and_ids_as_the_headline_search": bool(ok)}
    finally:
        stage._resident.pop(os.path.join(root, "docs"), None)  # (the index stays with the caller)
        shutil.rmtree(root, ignore_errors=True)


def toy_wordpiece(vocab_size=30522, seed=12):
    """A WordPiece tokenizer over a toy vocabulary of letter-only pseudo-words (no tokenizer file exists offline): one token per
    word, BERT's normaliser / pre-tokeniser / [CLS] A [SEP] B [SEP] template.  Returns (tokenizer, words)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    rng = np.random.default_rng(seed)
    syl = ["ka", "mi", "to", "ra", "ne", "lo", "si", "du", "pe", "ga", "vo", "ti", "ma", "re", "ku", "ban", "ter", "lin", "sor", "pad",
           "an", "el", "ion", "st", "qu"]
    pool = set()
    while len(pool) < vocab_size - 5:  # letter-only pseudo-words of 1-4 syllables (word-like lengths for the tokeniser)
        pool.add("".join(syl[j] for j in rng.integers(0, len(syl), size=int(rng.integers(1, 5)))))
    words = sorted(pool)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words
    t = Tokenizer(models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))
    t.normalizer = normalizers.BertNormalizer(lowercase=True)
    t.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    t.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                     special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    tok = PreTrainedTokenizerFast(tokenizer_object=t, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]",
                                  mask_token="[MASK]", model_input_names=["input_ids", "token_type_ids", "attention_mask"])
    return tok, words


def encode_stage_leg(args, device_index):
    """`Retrieve.encode_and_save` (reference modules/retrieve.py:110-144) as a stage: text passages -> tokeniser in DataLoader
    worker processes (collate_fn, padding='longest') -> HIP forward pass -> D2H of every batch -> chunk files in the reference's
    layout.  Synthetic KILT-like passages (100 words + a 4-word title, one WordPiece token per word from a 30 522-entry toy
    vocabulary; no real text or tokenizer file exists offline), BERT-base random-init weights, batch 512 (retromae.yaml).
    Tokenisation runs ahead of the GPU on 4 / 16 threads of this process (bergen_amd's default loader), on 4 DataLoader worker
    processes (the reference's, retrieve.py:114-118) or in line; passages/s is end to end (the best is reported on top), the
    tokeniser-only rate of the same batches (one call at a time) beside it."""
    import shutil
    import tempfile

    import datasets

    import bergen_amd
    from bergen_amd import BertEncoder, synth
    n_pass = args.encode_stage_passages
    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    tok, words = toy_wordpiece(cfg["vocab_size"], seed=12)
    rng = np.random.default_rng(13)
    lens = np.clip(np.rint(rng.normal(104, 24, size=n_pass)), 12, 250).astype(np.int64)
    picks = rng.integers(0, len(words), size=int(lens.sum()))
    texts, o = [], 0
    for n in lens.tolist():
        texts.append(" ".join(words[j] for j in picks[o:o + n].tolist()))
        o += n
    sd = synth.random_bert(cfg, seed=31)
    enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=device_index)
    dense = bergen_amd.Dense(model_name="bench/bert-base-random", max_len=256, pooler=bergen_amd.ClsPooler(),
                             similarity=bergen_amd.DotProduct(), model=enc, tokenizer=tok)
    root = tempfile.mkdtemp(prefix="bergen_encode_stage_")
    res = {"workload": f"Retrieve.encode_and_save: {n_pass} synthetic passages (~{int(lens.mean())} words), WordPiece tokeniser, "
                       f"BERT-base random-init on the HIP path, batch 512, chunk files written"}
    try:
        ds = datasets.Dataset.from_dict({"content": texts})
        # tokeniser alone, one process
        t0 = time.perf_counter()
        for b0 in range(0, n_pass, 512):
            dense.collate_fn([{"content": x} for x in texts[b0:b0 + 512]], "doc")
        res["tokenizer_only_passages_per_s_one_process"] = n_pass / (time.perf_counter() - t0)
        # "threads": the stage's default loader — whole batches on 4 / 16 threads of this process, each call spread over the cores by the
        # tokenizer's own pool; "threads_pieces": every batch tokenised in eight pieces by up to 32 threads, each piece serially
        # (BERGEN_AMD_TOKENIZER_PIECES=1); "processes": the reference's DataLoader workers; "inline": none
        for loader, workers in (("threads", 4), ("threads_pieces", 4), ("threads", 16), ("processes", 4), ("inline", 0)):
            stage = bergen_amd.Retrieve(init_args=dense, batch_size=512, num_workers=workers, device=device_index,
                                        loader="processes" if loader == "processes" else "threads", require_native=True)
            res["backend"] = stage.backend  # ('hip': require_native would have refused anything else)
            path = os.path.join(root, f"idx_{loader}{workers}")
            if loader == "threads_pieces":
                os.environ["BERGEN_AMD_TOKENIZER_PIECES"] = "1"
            else:
                os.environ.pop("BERGEN_AMD_TOKENIZER_PIECES", None)
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                stage.encode_and_save(ds, save_path=path, query_or_doc="doc")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                rows = sum(int(torch.load(os.path.join(path, f)).shape[0]) for f in os.listdir(path))
                st = getattr(stage, "last_encode_stats", None) or {}
                res[f"workers_{loader}_{workers}"] = {"passages_per_s": n_pass / dt, "seconds": dt, "rows_written": rows,
                                                      "chunk_files": len(os.listdir(path)),
                                                      "steady_state_passages_per_s": st.get("steady_state_rows_per_s"),
                                                      "first_batch_seconds": st.get("first_batch_seconds"),
                                                      "last_chunk_write_seconds": st.get("last_chunk_write_seconds")}
            except Exception as exc:
                res[f"workers_{loader}_{workers}"] = {"error": repr(exc)}
            shutil.rmtree(path, ignore_errors=True)
        os.environ.pop("BERGEN_AMD_TOKENIZER_PIECES", None)
        good = [v["passages_per_s"] for kk, v in res.items() if kk.startswith("workers_") and "passages_per_s" in v]
        res["passages_per_s"] = max(good) if good else None
        return res
    finally:
        enc.close()
        shutil.rmtree(root, ignore_errors=True)


def make_cross_encoder(deberta, device_index):
    """A random-init cross-encoder of the reference's two reranker families on the HIP path: DeBERTa-v3-large shape (disentangled
    attention, 256 position buckets: config/reranker/debertav3.yaml) or BERT-large shape (BAAI/bge-reranker-large ...).  Returns (encoder, cfg)."""
    from bergen_amd import BertEncoder, synth
    deb = deberta
    cfg = dict(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = synth.random_bert(cfg, seed=61)
    synth.random_cls_head(cfg, seed=62, num_labels=1, sd=sd)
    if deb:
        g = np.random.default_rng(63)
        cfg.update(model_type="deberta-v2", type_vocab_size=0, layer_norm_eps=1e-7, relative_attention=True, position_buckets=256,
                   norm_rel_ebd="layer_norm", share_att_key=True, pos_att_type="p2c|c2p", position_biased_input=False,
                   max_relative_positions=-1)
        sd = {k.replace(".attention.self.query.", ".attention.self.query_proj.").replace(".attention.self.key.", ".attention.self.key_proj.")
              .replace(".attention.self.value.", ".attention.self.value_proj."): v for k, v in sd.items()
              if not k.startswith(("embeddings.position_embeddings", "embeddings.token_type_embeddings"))}
        sd["encoder.rel_embeddings.weight"] = (g.standard_normal((512, 1024)) * 0.02).astype(np.float16).astype(np.float32)
        sd["encoder.LayerNorm.weight"] = np.ones(1024, np.float32)
        sd["encoder.LayerNorm.bias"] = np.zeros(1024, np.float32)
    enc = BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=device_index)
    return enc, cfg


def rerank_leg(args, device_index):
    """The rerank stage's model call (reference models/rerankers/crossencoder.py:34-38) at the shapes of the
    reference's two reranker families, random-init weights: a DeBERTa-v3-large-shaped cross-encoder (24 x 1024, 16 heads, 256
    position buckets: naver/trecdl22-crossencoder-debertav3, config/reranker/debertav3.yaml:3) with disentangled attention,
    and a BERT-large-shaped one (BAAI/bge-reranker-large ...), on 32 (query, passage) pairs of ~180 attended tokens (the
    reference pads each pair to max_len 256 and runs the padding through the model).  pairs/s from the forward's HIP events."""
    from bergen_amd import BertEncoder, synth
    out = {}
    rng = np.random.default_rng(17)
    lens = np.clip(np.rint(rng.normal(180, 40, size=32)), 32, 256).astype(np.int64)
    T = 256
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    lens32, mask32 = lens, mask
    # config/reranker/bge.yaml (BAAI/bge-reranker-large, an XLM-R-large-shaped encoder = BERT-large's layer stack) runs at batch_size 256:
    # the same model shape on 256 pairs — GEMMs that fill the chip, unlike the 32-pair batch of debertav3.yaml
    lens256 = np.clip(np.rint(rng.normal(180, 40, size=256)), 32, 256).astype(np.int64)
    mask256 = (np.arange(T)[None, :] < lens256[:, None]).astype(np.int64)
    for name, deb in (("deberta_v3_large_shape", True), ("bert_large_shape", False), ("bert_large_shape_256_pairs", False)):
        lens, mask = (lens256, mask256) if name.endswith("256_pairs") else (lens32, mask32)
        n_pairs = len(lens)
        enc, cfg = make_cross_encoder(deb, device_index)
        ids = rng.integers(1, cfg["vocab_size"], size=(n_pairs, T)).astype(np.int64) * mask
        kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        enc.classify(kw)
        best = 1e9
        for _ in range(3):
            logits = enc.classify(kw)
            best = min(best, enc.counters()["forward_ms"])
        flops = float(enc.counters()["flops"])  # the forward's algorithmic flops over the attended tokens (projections + attention)
        tf = flops / (best * 1e-3) / 1e12
        out[name] = {"pairs_per_s": n_pairs / (best * 1e-3), "pairs": n_pairs, "forward_ms": best, "attended_tokens": int(lens.sum()),
                     "backend": "hip", "finite": bool(torch.isfinite(logits).all()),
                     "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
                                  "algorithmic_flops_per_step": flops,
                                  "note": ("32 pairs = 5.4 k tokens: 22 x 4 tiles of 256 x 256 per N = 1024 projection on 256 CUs - a launch- and "
                                           "latency-bound shape, the fraction says how far from the matrix peak such a batch sits") if n_pairs == 32 else
                                          "256 pairs (config/reranker/bge.yaml batch_size): GEMMs that fill the chip"}}
        if n_pairs == 32:
            try:
                out[name]["through_rerank_eval"] = rerank_eval_leg(enc, cfg, n_pairs)
            except Exception as e:  # noqa: BLE001 — a leg must not take the line down
                out[name]["through_rerank_eval"] = {"error": f"{type(e).__name__}: {e}"}
        enc.close()
    return out


_RERANK_TOK = []


def rerank_eval_leg(enc, cfg, yaml_batch):
    """The rerank STAGE (reference modules/rerank.py:24-48) at the yaml's batch size: `Rerank(init_args=CrossEncoder, batch_size=32).eval`
    over 64 queries x 50 retrieved passages = 3 200 (query, passage) text pairs (~180 tokens each: a 12-word question + a KILT-like
    passage, toy WordPiece vocabulary), tokeniser included.  The stage coalesces the yaml batches into launches of >= 256 pairs,
    tokenises ahead on threads and copies the scores back once (bergen_amd/rerank.py); the same stage with one launch per yaml batch
    (launch_pairs = 1: the reference's loop shape) is timed beside it and must return the SAME bits."""
    import bergen_amd
    if not _RERANK_TOK:
        _RERANK_TOK.append(toy_wordpiece(cfg["vocab_size"], seed=12))
    tok, words = _RERANK_TOK[0]
    rng = np.random.default_rng(23)
    n_q, per_q = 64, 50
    data = []
    for qi in range(n_q):
        q = " ".join(words[j] for j in rng.integers(0, len(words), size=12))
        for di in range(per_q):
            n = int(np.clip(np.rint(rng.normal(165, 40)), 20, 240))
            data.append({"query": q, "doc": " ".join(words[j] for j in rng.integers(0, len(words), size=n)), "q_id": f"q{qi}", "d_id": f"d{qi}_{di}"})
    ce = bergen_amd.CrossEncoder("bench/cross-encoder-random", max_len=256, model=enc, tokenizer=tok)
    res = {"pairs": len(data), "yaml_batch_size": yaml_batch, "backend": ce.backend}
    outs = {}
    for label, launch_pairs in (("coalesced_256", 256), ("one_launch_per_yaml_batch", 1)):
        stage = bergen_amd.Rerank(init_args=ce, batch_size=yaml_batch, launch_pairs=launch_pairs, num_workers=4)
        stage.eval(data[:per_q * 8])  # warm-up (workspace growth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs[label] = stage.eval(data)
        dt = time.perf_counter() - t0
        st = stage.last_eval_stats
        tf = st["algorithmic_flops"] / dt / 1e12
        res[label] = {"pairs_per_s": len(data) / dt, "seconds": dt, "launches": st["launches"], "kernel_ms": st["kernel_ms"],
                      "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
                                   "frac_kernels_only": st["algorithmic_flops"] / (st["kernel_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                   "algorithmic_flops": st["algorithmic_flops"]}}
    a, b = outs["coalesced_256"], outs["one_launch_per_yaml_batch"]
    res["identical_to_per_batch_loop"] = bool(a["q_id"] == b["q_id"] and a["doc_id"] == b["doc_id"] and
                                              all(torch.equal(x, y) for x, y in zip(a["score"], b["score"])))
    res["pairs_per_s"] = res["coalesced_256"]["pairs_per_s"]
    res["frac"] = res["coalesced_256"]["roofline"]["frac"]
    return res


def certificate_leg(args, local_rank, device):
    """The exactness certificate and its fall-back on a corpus that is NOT unit-norm Gaussian: rows with
    RetroMAE-like norms (|x| ~ U(10, 14); one row in 10 000 at 3x that: the certificate's bound uses the corpus-wide maximum),
    2 % of the rows near-duplicates of their predecessor (templated passages: relative noise 1e-3 or 1e-2), and for a tenth of
    the queries a planted cluster of 100 near-duplicates of one well-scoring row (half of them at noise 1e-3, half at 1e-2) —
    the situation in which more than KP - k rows lie within MFMA rounding error of the k-th score.  Reported: how many
    queries the certificate could not prove, how many filter passes and re-scored rows the fall-back took, its wall time,
    against the time of the same search.  Gate: canonical scores of the returned ids recomputed with numpy, and the complete
    lists of 4 clustered + 4 ordinary queries against a float64 GEMM over the whole corpus."""
    import bergen_amd
    dim, k, nq, n = args.dim, args.k, args.queries, args.n_rows
    g = torch.Generator(device=device).manual_seed(77)
    queries = (torch.nn.functional.normalize(torch.randn(nq, dim, generator=g, device=device), dim=1) * 12.0).half()
    clustered = torch.arange(0, nq, 10)  # every tenth query
    gp = torch.Generator().manual_seed(78)
    cluster_at = torch.randint(0, n - 200, (len(clustered),), generator=gp)  # first row of each planted cluster

    def block_rows(b):
        b0 = b * BLOCK
        m = min(BLOCK, n - b0)
        gb = torch.Generator(device=device).manual_seed(5000 + b)
        base = torch.nn.functional.normalize(torch.randn(m, dim, generator=gb, device=device), dim=1)
        norms = 10.0 + 4.0 * torch.rand(m, 1, generator=gb, device=device)
        norms = torch.where(torch.rand(m, 1, generator=gb, device=device) < 1e-4, norms * 3.0, norms)
        rows = base * norms
        dup = torch.rand(m, generator=gb, device=device) < 0.02
        dup[0] = False
        eps = torch.where(torch.rand(m, 1, generator=gb, device=device) < 0.5, 1e-3, 1e-2)
        src = torch.arange(m, device=device)
        src[dup] -= 1
        noise = torch.randn(m, dim, generator=gb, device=device) * (eps * norms / dim ** 0.5)
        rows = torch.where(dup[:, None], rows[src] + noise, rows)
        for j, (qi, r0) in enumerate(zip(clustered.tolist(), cluster_at.tolist())):
            if b0 <= r0 < b0 + m - 100:
                gc = torch.Generator(device=device).manual_seed(9000 + j)
                centre = torch.nn.functional.normalize(queries[qi].float() / 12.0 + 0.3 * torch.randn(dim, generator=gc, device=device) / dim ** 0.5,
                                                       dim=0) * 12.0  # cosine ~0.96 with its query: the cluster IS the query's top 100
                e = torch.cat([torch.full((50, 1), 1e-3, device=device), torch.full((50, 1), 1e-2, device=device)])
                rows[r0 - b0:r0 - b0 + 100] = centre[None, :] + torch.randn(100, dim, generator=gc, device=device) * (e * 12.0 / dim ** 0.5)
        return rows.half()

    ix = bergen_amd.FlatIndex(n, dim, metric="ip", device=local_rank)
    for b in range((n + BLOCK - 1) // BLOCK):
        ix.upload(block_rows(b), row0=b * BLOCK)
    ix.finalize()
    torch.cuda.synchronize()
    ix.search(queries, k)
    t0 = time.perf_counter()
    s_dev, i_dev = ix.search(queries, k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = ix.counters()
    s_np, i_np = s_dev.cpu().numpy(), i_dev.cpu().numpy()
    ix.close()
    # gate: (a) canonical scores of returned ids (sequential fp64), (b) complete lists of 8 queries vs a float64 GEMM
    ok = bool((np.diff(s_np, axis=1) <= 0).all())
    check = clustered[:4].tolist() + [1, 2, 3, 4]
    q64 = queries[check].double()
    keep_s, keep_i = [[] for _ in check], [[] for _ in check]
    for b in range((n + BLOCK - 1) // BLOCK):
        x64 = block_rows(b).double()
        sc = (q64 @ x64.T).float()
        kth = torch.topk(sc, k, dim=1).values[:, -1:]
        hit = (sc >= kth).nonzero()
        vals = sc[hit[:, 0], hit[:, 1]].cpu().tolist()
        for (a, r), v in zip(hit.cpu().tolist(), vals):
            keep_s[a].append(v)
            keep_i[a].append(b * BLOCK + r)
        del x64, sc
    full_ok = True
    for a, qi in enumerate(check):
        cs, ci = np.asarray(keep_s[a], np.float32), np.asarray(keep_i[a], np.int64)
        order = np.lexsort((ci, -cs.astype(np.float64)))[:k]
        full_ok &= bool(np.array_equal(ci[order], i_np[qi]) and np.array_equal(cs[order].view(np.uint32), s_np[qi].view(np.uint32)))
    ok &= full_ok
    in_cluster = float(np.mean([np.isin(i_np[qi], np.arange(r0, r0 + 100)).sum() for qi, r0 in zip(clustered.tolist(), cluster_at.tolist())]))
    return {"workload": f"{nq} queries (|q| = 12; every tenth with a planted cluster of 100 near-duplicate rows) x {n} x {dim} fp16, "
                        f"|x| ~ U(10, 14) with 1e-4 of the rows at 3x, 2 % near-duplicate rows, top-{k}",
            "uncertified_queries": int(c.get("uncertified_queries", 0)), "uncertified_fraction": c.get("uncertified_queries", 0) / nq,
            "fallback_filter_passes": int(c.get("exact_passes", 0)), "fallback_rows_rescored": int(c.get("exact_rows_rescored", 0)),
            "fallback_ms": c.get("exact_ms", 0.0), "scan_ms": c["scan_ms"], "normal_pass_ms": c["scan_ms"] / c["n_passes"],
            "search_ms_wall": dt * 1e3, "queries_per_s": nq / dt, "top_k_rows_inside_planted_cluster_mean": in_cluster,
            "full_lists_of_8_queries_bit_exact_vs_float64_gemm": full_ok, "parity_check": "pass" if ok else "FAIL"}


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def _pci_address(device):
    """'0000:bb:dd.f' of HIP device `device` (hipDeviceGetPCIBusId), or None."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        return buf.value.decode().lower()
    except OSError:
        return None


class PowerSampler:
    """Side thread: (t, socket watts, shader clock MHz) of THIS process's GPU every `period` s from its hwmon node in sysfs,
    falling back to amd-smi (slower: one process per sample).  Why bench.py cares: the headline scan runs AT the board's
    power cap (1 400 W) with the shader clock pulled down to ~1.5 GHz (profiles/r03_power_clock.json) — the time of a pass is
    set by the energy it takes, which a roofline of HBM bytes or MFMA flops alone does not show."""

    def __init__(self, period=0.05):
        import glob
        import threading
        self.period = period
        # the hwmon node of THIS process's GPU (the box has eight; the HIP device is found by its PCI address)
        self.hw = None
        bdf = _pci_address(0)
        cands = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")) if bdf else []
        for h in cands:
            if _read(os.path.join(h, "power1_average")) or _read(os.path.join(h, "power1_input")):
                self.hw = h
                break
        self.pci = bdf
        self.dev = os.path.dirname(os.path.dirname(self.hw)) if self.hw else None
        self.source = "hwmon " + self.hw if self.hw else "amd-smi"
        self.samples = []
        self._threading = threading
        self._stop = threading.Event()
        self._thread = None

    def cap_watts(self):
        v = _read(os.path.join(self.hw, "power1_cap")) if self.hw else None
        return int(v) / 1e6 if v else None

    def _one(self):
        if self.hw:
            p = _read(os.path.join(self.hw, "power1_average")) or _read(os.path.join(self.hw, "power1_input"))
            f = _read(os.path.join(self.hw, "freq1_input"))
            mhz = int(f) / 1e6 if f else None
            if mhz is None:
                dpm = _read(os.path.join(self.dev, "pp_dpm_sclk")) or ""
                for line in dpm.splitlines():
                    if line.endswith("*"):
                        mhz = float(line.split(":")[1].strip().rstrip("*").strip().lower().rstrip("mhz"))
            return (int(p) / 1e6 if p else None), mhz
        try:
            import subprocess
            out = subprocess.run(["amd-smi", "metric", "-g", "0", "-p", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(out)
            j = j[0] if isinstance(j, list) else j
            pw = j.get("power", {}).get("socket_power", {})
            pw = pw.get("value") if isinstance(pw, dict) else pw
            clk = j.get("clock", {}).get("gfx_0", {}).get("clk", {})
            clk = clk.get("value") if isinstance(clk, dict) else clk
            return (float(pw) if pw not in (None, "N/A") else None), (float(clk) if clk not in (None, "N/A") else None)
        except Exception:
            return None, None

    def start(self):
        self.samples = []
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                w, f = self._one()
                self.samples.append((time.perf_counter(), w, f))
                self._stop.wait(self.period)
        self._thread = self._threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self, skip_s=0.5):
        self._stop.set()
        self._thread.join()
        if not self.samples:
            return {}
        import statistics
        t0 = self.samples[0][0]
        ws = [w for t, w, f in self.samples if w is not None and t - t0 >= skip_s]
        fs = [f for t, w, f in self.samples if f is not None and t - t0 >= skip_s]
        out = {"samples": len(ws)}
        if ws:
            out.update(watts_mean=round(statistics.mean(ws), 1), watts_max=round(max(ws), 1))
        if fs:
            out.update(sclk_mhz_mean=round(statistics.mean(fs)), sclk_mhz_min=round(min(fs)), sclk_mhz_max=round(max(fs)))
        return out


def scan_kernel_name(query_tile):
    """Kernel that serves a query tile of this width (bergen_amd/csrc/index.hip)."""
    return {256: "bh_scan_topk256_kernel", 192: "bh_scan_topk192_kernel"}.get(query_tile, "bh_scan_topk_kernel")


def scan_roofline(acc, c, steps, n_rows, dim, k, traffic_json, dim_padded=None):
    """The `roofline` object of a dense search: the DOMINANT kernel launch of the step against the HBM roofline, with the
    algorithmic bytes of SURVEY §8d — per pass of one query tile  N*d*2 + Bq*d*2 + Bq*k*12  — times the passes ONE launch serves.
    `acc` = scan_ms / tail_scan_ms / paired_scan_ms summed over `steps` searches, `c` = the last search's counters.
    Three kinds of launch can make up a step (bergen_amd/csrc/index.hip):
      * PAIRED launches (option pair256, the default): two query-tile passes in one launch, each on half the grid; partner
        workgroups of one XCD walk the same tiles, so a corpus line leaves HBM once per launch and the second reader takes it
        from L2.  Units per launch = 2 passes: `algorithmic_bytes_per_launch` is twice the per-pass figure, while the HBM bytes
        such a launch has to move are only  N*d*2 + 2*(Bq*d*2 + Bq*k*12)  (`hbm_bytes_needed_per_launch`; `traffic` = PMC).
      * one unpaired launch when the number of full passes is odd, and
      * the tail pass (<= 128 queries left) on the 128-query kernel.
    Each kind is timed by its own pair of HIP events on the launch stream (counters paired_scan_ms / tail_scan_ms / scan_ms)."""
    tile = c["query_tile"]
    has_tail = c.get("tail_query_tile", 0) != 0 and c["n_passes"] > 1
    n_pair = c.get("paired_launches", 0)
    n_single = c["n_passes"] - (1 if has_tail else 0) - 2 * n_pair
    per_pass = n_rows * dim * 2.0 + tile * dim * 2.0 + tile * k * 12.0
    tail_ms = acc.get("tail_scan_ms", 0.0) if has_tail else 0.0
    pair_ms = acc.get("paired_scan_ms", 0.0) if n_pair else 0.0
    single_ms = acc["scan_ms"] - tail_ms - pair_ms
    # the BALANCED launch (option balance_tail: the remainder of the query set as one more paired launch with idle waves) is a paired
    # launch with fewer queries than two tiles: reported beside the dominant launch, kept out of its average
    bal_q = int(c.get("balanced_queries", 0) or 0)
    bal_ms = acc.get("balanced_scan_ms", 0.0) if bal_q else 0.0
    n_pair_full = n_pair - (1 if bal_q else 0)
    pair_ms_all, n_pair_all = pair_ms, n_pair
    if bal_q and n_pair_full > 0:
        pair_ms -= bal_ms
        n_pair = n_pair_full
    name = scan_kernel_name(tile) if not (tile == 128 and c.get("shader_mhz", 0) != 0) else "bh_scan_topk256_kernel"  # (d = 1024: 128-query tile of the 256 kernel)
    dp = dim_padded or dim
    if n_pair:
        units, launches, avg = 2, n_pair * steps, pair_ms / (n_pair * steps)
        traffic = pmc_traffic(traffic_json, name, n_rows, dp, variant="paired")
        kind = f"paired: two {tile}-query passes per launch (template ABL = 128), partner workgroups of one XCD share the corpus stream through L2"
    else:
        units, launches, avg = 1, n_single * steps, single_ms / max(1, n_single * steps)
        traffic = pmc_traffic(traffic_json, name, n_rows, dp)
        kind = f"one {tile}-query pass per launch"
    per_launch = units * per_pass
    achieved = per_launch / (avg * 1e-3) / 1e9
    flops = 2.0 * units * tile * n_rows * dim
    hbm_needed = n_rows * dim * 2.0 + units * (tile * dim * 2.0 + tile * k * 12.0)
    out = {"bound": "hbm", "kernel": name, "launch": kind, "passes_per_launch": units,
           "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
           "algorithmic_bytes_per_launch": per_launch, "algorithmic_bytes_per_pass": per_pass, "avg_launch_ms": avg, "launches": launches,
           # what the launch must take from / bring to HBM when the corpus is read once per LAUNCH, and the rate that corresponds to
           "hbm_bytes_needed_per_launch": hbm_needed, "hbm_frac_of_needed_bytes": hbm_needed / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           # the same launch against the other roofline: a tile of 256 queries is 256 flop per corpus byte (ridge ~310)
           "mfma_tflops": flops / (avg * 1e-3) / 1e12, "mfma_frac": flops / (avg * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
           "shader_mhz": c.get("shader_mhz", 0.0)}
    # `frac` follows SURVEY §8d's accounting (bytes per PASS x passes a launch serves: a paired launch counts the corpus twice although
    # it leaves HBM once).  The figure that says how close the launch is to what physically binds it is the larger of the HBM rate of
    # the bytes it must move and its MFMA rate:
    out["algorithmic_frac"] = out["frac"]
    out["frac_binding"] = max(out["hbm_frac_of_needed_bytes"], out["mfma_frac"])
    out["binding_resource"] = "mfma" if out["mfma_frac"] >= out["hbm_frac_of_needed_bytes"] else "hbm"
    # `bound` names what physically binds the launch (round-5 review: the line said "hbm" beside binding_resource "mfma"); `achieved`,
    # `peak`, `unit` and `frac` stay SURVEY §8d's byte accounting (bound_of_frac says so), the MFMA side is mfma_tflops / mfma_frac /
    # mfma_busy_frac (the matrix pipe's busy share by the SQ counters, profiles/sq_counters.json)
    out["bound"] = out["binding_resource"]
    out["bound_of_frac"] = "hbm (SURVEY 8d bytes per pass x passes per launch)"
    want = ("128" if n_pair else "0")
    busy = pmc_mfma_busy("bh_scan_topk256_kernel", lambda nm: f"<{dp // 32}," in nm.replace(" ", "") and
                         nm.replace(" ", "").split(",")[6] == want) if name == "bh_scan_topk256_kernel" else None
    out["mfma_busy_frac"] = busy["mfma_busy_frac"] if busy else None
    out["mfma"] = {"achieved": out["mfma_tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": out["mfma_frac"],
                   "busy_frac": out["mfma_busy_frac"], "busy": busy}
    # the two rooflines side by side, each self-consistent (achieved / peak / unit / frac): `hbm` by SURVEY §8d's algorithmic bytes (the
    # top-level achieved / peak / unit / frac repeat it: the contract's accounting and north_star's ">= 50 % of HBM peak" figure), `mfma`
    # above; `bound` says which of the two physically binds the launch
    out["hbm"] = {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "accounting": "SURVEY 8d bytes per pass x passes per launch",
                  "traffic": traffic, "frac_of_bytes_it_must_move": out["hbm_frac_of_needed_bytes"]}
    if n_pair and n_single > 0:
        a1 = single_ms / (n_single * steps)
        out["unpaired_launch"] = {"kernel": name, "passes_per_launch": 1, "launches": n_single * steps, "avg_launch_ms": a1,
                                  "frac": per_pass / (a1 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                  "traffic": pmc_traffic(traffic_json, name, n_rows, dp),
                                  "what": "an odd number of full passes leaves one that runs alone"}
    if bal_q:
        # (a kernel trace — rocprofv3 --kernel-trace --stats — averages over EVERY launch of this kernel instantiation, the balanced one included)
        out["avg_launch_ms_over_all_launches_of_this_kernel"] = pair_ms_all / max(1, n_pair_all * steps)
        out["launches_of_this_kernel"] = n_pair_all * steps
        out["balanced_launch"] = {"kernel": name, "queries": bal_q, "avg_launch_ms": bal_ms / steps, "launches": steps,
                                  "what": "the queries left behind the last full pair of passes, cut into two passes of about half each: one paired launch, "
                                          "waves without a query skip the tiles' MFMAs (index.hip option balance_tail)"}
    out["tail_pass"] = ({"kernel": "bh_scan_topk_kernel", "query_tile": 128, "avg_launch_ms": tail_ms / steps,
                         "what": "the last pass of a step (<= 128 queries left) runs on the 128-query kernel"} if has_tail else None)
    return out


KERNEL_SOURCES = {"bh_scan_topk256_kernel": "scan_topk256.hip", "bh_scan_topk192_kernel": "scan_topk192.hip",
                  "bh_scan_topk_kernel": "scan_topk.hip", "bh_csr_scan_mfma_kernel": "csr_mfma.hip"}


def pmc_traffic(path, kernel, n_rows, dim_padded, variant=None):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/hbm_traffic.json, keyed "<kernel>@<dim>";
    `variant` "paired" = the paired launch of the 256-query kernel, keyed "<kernel>/paired@<dim>"), or
    None when that file holds no entry for this kernel at this geometry — or when the kernel's source file has changed since
    the counters were collected (the entry carries the file's sha256): a stale measurement is not reported."""
    import hashlib
    try:
        ent = json.load(open(path)).get("kernels", {}).get(f"{kernel}/{variant}@{dim_padded}" if variant else f"{kernel}@{dim_padded}")
        if not ent or ent.get("n_rows") != n_rows:
            return None
        src = os.path.join(ROOT, "bergen_amd", "csrc", KERNEL_SOURCES[kernel])
        if ent.get("source_sha16") != hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]:
            return None
        return ent.get("hbm_bytes_per_launch")
    except Exception:
        return None


def _lib_path():
    from bergen_amd import _lib
    return _lib.LIB_PATH


def _bh_version():
    from bergen_amd import _lib
    return int(_lib.lib().bh_version())


def pmc_mfma_busy(kernel_prefix, match=None, path=None):
    """`mfma_busy_frac` of a kernel instantiation from the committed SQ-counter summary (profiles/sq_counters.json, made by
    profiles/make_sq_counters.py from rocprofv3 --pmc passes on a GPU box): the share of a launch's shader-clock cycles in which a
    SIMD's matrix pipe was busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8).  `match` picks the instantiation (a
    predicate on the demangled name).  None when there is no entry or when the kernel's source has changed since the counters were
    collected (each entry carries the sha256 of its source files): a stale measurement is not reported."""
    import hashlib
    try:
        ents = json.load(open(path or os.path.join(ROOT, "profiles", "sq_counters.json"))).get("kernels", {})
        for name, ent in ents.items():
            if not name.startswith(kernel_prefix) or (match is not None and not match(name)):
                continue
            h = hashlib.sha256()
            for f in ent.get("source_files", []):
                h.update(open(os.path.join(ROOT, "bergen_amd", "csrc", f), "rb").read())
            if ent.get("source_sha16") != h.hexdigest()[:16]:
                return None
            return {"mfma_busy_frac": ent.get("mfma_busy_frac"), "kernel": name, "wait_any_frac": ent.get("wait_any_frac"),
                    "lds_conflict_frac": ent.get("lds_conflict_frac"), "from": "profiles/" + str(ent.get("from")),
                    "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), rocprofv3 --pmc under the profiler's clock"}
    except Exception:
        return None
    return None


class HipEnv:
    """Where the bench runs: one MI355X per rank, RCCL between the ranks, the HIP library underneath.  bench.py itself only
    ever builds this one (there is no CPU mode); tests/bench_standin.py drives run() with an oracle-backed stand-in so that
    the multi-rank control flow (sharding, gather, merge, barrier / max-over-ranks timing) is exercised without GPUs."""
    backend = "nccl"
    merge = None  # ShardedSearcher's default: the HIP merge kernel
    results_to_host = True

    def __init__(self, local_rank, backend="nccl"):
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
        self.backend = backend
        if backend != "nccl":  # (test mode: more ranks than GPUs)
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.device = torch.device("cuda", local_rank)

    def init_dist(self, rank, world):
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=self.device)
        else:
            dist.init_process_group(self.backend, rank=rank, world_size=world)

    def init_library(self, args):
        from bergen_amd import _lib
        _lib.init(self.local_rank)
        for name in ("query_tile", "query_split", "pair_window", "share_threshold", "nontemporal"):
            if getattr(args, name) is not None:
                _lib.set_option(name, getattr(args, name))

    def make_index(self, n_rows, dim):
        import bergen_amd
        return bergen_amd.FlatIndex(n_rows, dim, metric="ip", device=self.local_rank)

    def describe_backend(self):
        return {"search": "hip", "library": os.path.relpath(_lib_path(), ROOT), "bh_version": _bh_version(),
                "require_native": os.environ.get("BERGEN_AMD_REQUIRE_NATIVE") == "1"}

    def make_stage(self, rank, world):
        """The stage object BERGEN builds (modules/rag.py:177-181), here with the row-sharded search switched on: the
        timed step is its search_rows() — the search half of Retrieve.retrieve."""
        import bergen_amd

        class _Plug:  # the index is filled on the device: the model is only asked for its name and similarity
            model_name = "bench/synthetic"
            similarity = bergen_amd.DotProduct()
            model = torch.nn.Identity()

        stage = bergen_amd.Retrieve(init_args=_Plug(), device=self.local_rank, search_rank=rank, search_world=world,
                                    search_results="rank0")
        if self.merge is not None:
            stage._shard_merge = self.merge
        return stage

    def sync(self):
        torch.cuda.synchronize()


def rank_line(env, rank, world, lo, hi, dim, build_s):
    """One stderr line per rank once its shard is resident — device, rows, transport version, free HBM — so that a multi-GPU run
    that fails later (a collective that never returns, a rank on the wrong device) can be diagnosed from the tail of its log."""
    try:
        import torch.distributed as dist
        backend = str(dist.get_backend()) if world > 1 and dist.is_initialized() else "none"
        ver = ""
        if backend == "nccl":
            try:
                ver = " rccl " + ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001
                ver = f" rccl version unavailable ({type(e).__name__})"
        dev = getattr(env, "device", None)
        name, free = "cpu", ""
        if dev is not None and dev.type == "cuda":
            name = torch.cuda.get_device_name(dev)
            f, t = torch.cuda.mem_get_info(dev)
            free = f", HBM free {f / 2**30:.0f} of {t / 2**30:.0f} GiB"
        print(f"[bench rank {rank}/{world}] pid {os.getpid()} device {dev} ({name}), rows [{lo}, {hi}) x {dim} resident after {build_s:.1f} s, "
              f"transport {backend}{ver}{free}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset')}",
              file=sys.stderr, flush=True)
    except Exception as e:  # noqa: BLE001 — a diagnostic must never take the run down
        print(f"[bench rank {rank}/{world}] (rank line failed: {e})", file=sys.stderr, flush=True)


def launch_ranks_if_needed(args, script=None):
    """`python bench.py --gpus N` typed WITHOUT a launcher (N > 1, no WORLD_SIZE in the environment): start the N ranks here —
    the same `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script>
    <flags>` command the driver uses, one process per GPU — relay rank 0's JSON line and exit with the launcher's status.
    Returns False when this process is itself a rank (or N = 1) and should run the bench."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return False
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.path.abspath(script or sys.argv[0])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL's dmabuf IPC; exported on the GPU boxes already)
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse_args()
    # a driver-run number must never come from a silent HF-torch fallback: every plug-in this script builds refuses one
    # (bergen_amd.dense._native_encoder; the stage legs also pass Retrieve(require_native=True))
    os.environ["BERGEN_AMD_REQUIRE_NATIVE"] = "1"
    launch_ranks_if_needed(args)
    run(args, HipEnv(int(os.environ.get("LOCAL_RANK", "0")), backend=args.dist_backend))


def run(args, env):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = env.device
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        env.init_dist(rank, world)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks "
                         f"(start it as `python bench.py --gpus N`, or under torch.distributed.run with --nproc-per-node N)")

    import bergen_amd
    # Host-side thread pools (torch intra-op, OpenMP, the tokenizer's rayon pool) sized for the CPUs this container may use — 16 of
    # the 256 logical CPUs a 1-GPU MI355X pod shows — instead of the host's count: pools of 256 threads spin through the cgroup's
    # CPU quota and the kernel then freezes the whole process for the rest of the 100 ms period (round 4: 18 ms of host-side stall
    # per 100 ms search step on one box, `nr_throttled` counting up; profiles/r04n_step_gap.txt).  Per rank: its share of the budget.
    cpu_budget = bergen_amd.utils.fit_host_pools_to_cpu_budget()
    if world > 1:
        torch.set_num_threads(max(1, cpu_budget // world))
    env.init_library(args)
    if world == 1:
        from bergen_amd import _lib  # (the single-GPU secondary legs switch library options)

    dim, k, nq, n_total = args.dim, args.k, args.queries, args.n_rows
    lo, hi = bergen_amd.shard_range(n_total, rank, world)
    queries = make_queries(nq, dim, device)
    t0 = time.perf_counter()
    ix = env.make_index(hi - lo, dim)
    planted, plant_rows = fill_shard(ix, lo, hi, dim, queries, n_total, device)
    ix.finalize()
    env.sync()
    build_s = time.perf_counter() - t0
    rank_line(env, rank, world, lo, hi, dim, build_s)
    # The search runs THROUGH THE STAGE: Retrieve.search_rows is the search half of Retrieve.retrieve (the code a BERGEN
    # pipeline reaches through modules/rag.py:322-329) — single GPU: fused scan + merge, lists written into pinned host
    # memory; N GPUs: the same on this rank's row shard with global row ids, one all-gather of the packed partial lists
    # (RCCL), canonical merge on rank 0, D2H there.  The shard was filled on the device, so the stage adopts it instead
    # of reading a 32 GB folder.
    corpus_key = "bench://synthetic-corpus"
    stage = env.make_stage(rank, world)
    stage.adopt_resident_index(corpus_key, ix, n_total, "ip", rows=(lo, hi) if world > 1 else None)

    def step():
        # search_seconds includes the D2H of the result lists: [Q, k] fp32 + int64, 1.7 MB at Q = 2 837
        r = stage.search_rows(queries, corpus_key, k, "ip", n_total)  # (CPU tensors on rank 0, None elsewhere)
        return r, r

    def barrier():
        env.sync()
        if world > 1:
            dist.barrier()
        env.sync()

    for _ in range(args.warmup):
        res, res_host = step()
    scan_ms = merge_ms = kernel_total_ms = tail_ms = pair_ms = bal_ms = 0.0
    uncertified = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, res_host = step()
        c = ix.counters()
        scan_ms += c["scan_ms"]
        tail_ms += c.get("tail_scan_ms", 0.0)
        pair_ms += c.get("paired_scan_ms", 0.0)
        bal_ms += c.get("balanced_scan_ms", 0.0)
        merge_ms += c["merge_ms"]
        kernel_total_ms += c["total_ms"]
        uncertified += c.get("uncertified_queries", 0)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if getattr(env, "backend", "nccl") == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    c = ix.counters()

    # ---- board power and shader clock under the headline load (rank 0's GPU, N = 1; outside the timed region): the same
    # step repeated for ~1.5 s with the hwmon node of this GPU sampled from a side thread
    power = None
    if rank == 0 and world == 1 and not args.no_power_leg:
        try:
            smp = PowerSampler()
            smp.start()
            t_p = time.perf_counter()
            n_p = 0
            while time.perf_counter() - t_p < 1.5:
                step()
                n_p += 1
            power = smp.stop(skip_s=0.4)
            power.update(cap_watts=smp.cap_watts(), source=smp.source, steps=n_p,
                         what="socket power / shader clock while the headline step repeats (after the timed region)")
        except Exception as e:  # (no readable sensor in this container: the leg reports that, the bench line stands)
            power = {"error": f"{type(e).__name__}: {e}"}

    # ---- parity gate (rank 0): planted positives on top + canonical scores of returned ids -------
    parity = "skipped"
    full_list_gate = None
    if rank == 0:
        s_np, i_np = res_host[0].numpy(), res_host[1].numpy()
        ok = bool((np.diff(s_np, axis=1) <= 0).all())
        owner = {}  # row -> query whose plant was written last (plants can collide on a row)
        for qi in range(nq):
            for j in range(5):
                owner[int(plant_rows[qi, j])] = qi
        for qi in range(nq):
            mine = set(r for r in (int(v) for v in plant_rows[qi].tolist()) if owner[r] == qi)
            ok &= set(i_np[qi, :len(mine)].tolist()) == mine
        if world == 1:
            # re-score the first 4 queries' hits on the CPU from regenerated rows: the canonical
            # scores must match bit-for-bit
            got_rows = _regenerate_rows(i_np[:4].reshape(-1), dim, queries, plant_rows, n_total, device)
            # (canonical score = fp32 of the SEQUENTIAL fp64 sum of the exact fp16 x fp16 products; cumsum is sequential)
            qf = queries[:4].cpu().numpy().astype(np.float64)
            xf = got_rows.astype(np.float64).reshape(4, k, dim)
            want = np.cumsum(qf[:, None, :] * xf, axis=-1)[..., -1].astype(np.float32)
            ok &= bool(np.array_equal(want.view(np.uint32), s_np[:4].view(np.uint32)))
            # ... and the COMPLETE lists of the first queries against a kernel-free recomputation over all n_total rows
            if args.full_list_queries > 0:
                check = gate_queries(nq, c, min(args.full_list_queries, nq))
                full_ok, full_list_gate = full_list_gate_fn(check, s_np, i_np, queries, dim, k, plant_rows, n_total, device)
                ok &= full_ok
        parity = "pass" if ok else "FAIL"

    if rank == 0:
        # The roofline object is the DOMINANT launch's (scan_roofline): paired launches of the 256-query kernel when there are
        # any, with the unpaired launch and the 128-query tail pass reported beside it.
        roof = scan_roofline({"scan_ms": scan_ms, "tail_scan_ms": tail_ms, "paired_scan_ms": pair_ms, "balanced_scan_ms": bal_ms}, c, args.steps, hi - lo, dim, k,
                             args.traffic_json)
        roof["power"] = power  # the launch is power-bound before it is HBM- or MFMA-bound: see PowerSampler
        out = {
            "metric": "queries/sec (value) + passages-encoded/sec (passages_per_s), KILT-100w-sized corpus (21M x 768 fp16) "
                      "top-50, kilt_nq-dev-sized query set; % of HBM / MFMA roofline",
            "value": nq / (elapsed / args.steps),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16 storage, f32 MFMA accumulate, f64 canonical re-score",
            "data": "synthetic (unit-norm Gaussian rows, 5 planted positives per query; SURVEY §8d S2)",
            "config": {
                "workload": f"configs[1] search half: {nq} queries x {n_total} x {dim} fp16, top-{k}, "
                            f"corpus row-sharded over {world} GPU(s), resident in HBM",
                "query_tile": c["query_tile"], "passes_per_step": c["n_passes"], "workgroups": c["n_workgroups"],
                "rows_per_gpu": hi - lo, "parallelism": f"row-shard x{world}" + (" + RCCL all-gather of partial top-k" if world > 1 else ""),
                # what computed the numbers of this line: the HIP library (there is no other search path in bergen_amd), and — for the
                # encoder / stage / rerank legs — the hand-written forward pass: bench.py runs with BERGEN_AMD_REQUIRE_NATIVE=1, under
                # which every plug-in REFUSES to fall back to an HF torch module
                "backend": env.describe_backend(),
            },
            "roofline": roof,
            "kernel_ms_per_step": {"scan": scan_ms / args.steps, "merge_rescore": merge_ms / args.steps,
                                   "stream_total": kernel_total_ms / args.steps},
            "index_build_seconds": build_s,
            "host": {"cpu_budget": cpu_budget, "host_logical_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads()},
            # queries (summed over the timed steps) whose exactness the certificate could not prove from the scan's
            # candidate lists and that took the exact fall-back scan (bergen_amd/csrc/certify.hip); inside the timed region
            "uncertified_queries": uncertified,
            "parity_check": parity,
            "full_list_gate": full_list_gate,
        }
        if world == 1 and not args.no_stage:
            try:
                out["retrieve_stage_full"] = retrieve_stage_full_leg(args, stage, ix, queries, res_host[0].numpy().copy(),
                                                                     res_host[1].numpy().copy(), n_total, k)
                stage.adopt_resident_index(corpus_key, ix, n_total, "ip", rows=None)
            except Exception as exc:
                out["retrieve_stage_full"] = {"error": repr(exc)}
        if world == 1 and args.query_split is None and not args.no_other_kernels:
            # secondary figures: the same search on the earlier scan kernels (library option scan_kernel), with result equality
            out["other_kernels"] = []
            try:
                for kern, label in ((2, "192-query tile, one wave per SIMD"), (0, "128-query tile, one wave per SIMD")):
                    _lib.set_option("scan_kernel", kern)
                    s_alt, i_alt = ix.search(queries, k)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        s_alt, i_alt = ix.search(queries, k)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / args.steps
                    cb = ix.counters()
                    out["other_kernels"].append({
                        "scan_kernel": kern, "what": label, "kernel": scan_kernel_name(cb["query_tile"]),
                        "queries_per_s": nq / dt, "query_tile": cb["query_tile"], "passes_per_step": cb["n_passes"],
                        "avg_launch_ms": cb["scan_ms"] / cb["n_passes"],
                        "roofline_frac": cb["algorithmic_bytes"] / cb["n_passes"] / (cb["scan_ms"] / cb["n_passes"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "traffic": pmc_traffic(args.traffic_json, scan_kernel_name(cb["query_tile"]), hi - lo, dim),
                        "same_results_as_headline": bool(torch.equal(torch.as_tensor(s_alt).cpu(), torch.as_tensor(res[0]).cpu()) and
                                                         torch.equal(torch.as_tensor(i_alt).cpu(), torch.as_tensor(res[1]).cpu()))})
            except Exception as exc:  # a secondary figure must never cost the headline line
                out["secondary_error"] = repr(exc)
            finally:
                _lib.set_option("scan_kernel", 3)
        if world == 1 and args.query_split is None and not args.no_larger_k:
            # secondary figures: larger top_k_documents at the headline geometry — candidate lists of 128 / 256 entries
            # (k = 100, 200) and the range-by-range search behind k > 248 (k = 1000, 256 queries) — each checked against
            # complete oracle lists of 4 queries (float64 GEMM over the regenerated corpus, see streamed_full_lists)
            try:
                want_s, want_i = streamed_full_lists(4, queries, dim, 1000, plant_rows, n_total, device)
                out["larger_k"] = []
                for k2, nq2 in ((100, nq), (200, nq), (1000, min(nq, 256))):
                    q2 = queries[:nq2]
                    s2, i2 = ix.search(q2, k2)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    s2, i2 = ix.search(q2, k2)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    cb = ix.counters()
                    s2n, i2n = torch.as_tensor(s2).cpu().numpy(), torch.as_tensor(i2).cpu().numpy()
                    exact = bool(np.array_equal(i2n[:4], want_i[:, :k2]) and np.array_equal(s2n[:4].view(np.uint32), want_s[:, :k2].view(np.uint32)))
                    out["larger_k"].append({"k": k2, "queries": nq2, "queries_per_s": nq2 / dt, "k_padded": cb["k_padded"], "query_tile": cb["query_tile"],
                                            "passes": cb["n_passes"], "scan_ms_per_pass": cb["scan_ms"] / max(1, cb["n_passes"]),
                                            "uncertified_queries": cb.get("uncertified_queries", 0),
                                            "full_lists_of_4_queries_bit_exact": exact})
                    if not exact:
                        out["parity_check"] = "FAIL (larger_k leg)"
            except Exception as exc:
                out["larger_k"] = {"error": repr(exc)}
        if args.sweep and world == 1:
            out["sweep"] = sweep(ix, queries, k, args)
        if world == 1 and not args.no_config5:
            ix.close()
            try:
                out["config5"] = config5_leg(args, local_rank, device)
                if out["config5"]["parity_check"] != "pass":
                    out["parity_check"] = "FAIL (config5 leg)"
            except Exception as exc:
                out["config5"] = {"error": repr(exc)}
        if world == 1 and not args.no_real_size:
            ix.close()
            try:
                out["real_size"] = real_size_leg(args, local_rank, device)
                if out["real_size"]["parity_check"] != "pass":
                    out["parity_check"] = "FAIL (real_size leg)"
            except Exception as exc:
                out["real_size"] = {"error": repr(exc)}
        if world == 1 and not args.no_certificate_leg:
            ix.close()
            try:
                out["certificate"] = certificate_leg(args, local_rank, device)
            except Exception as exc:
                out["certificate"] = {"error": repr(exc)}
        if world == 1 and args.folder_stage:
            ix.close()
            try:
                out["retrieve_stage"] = retrieve_stage_leg(args, local_rank)
            except Exception as exc:
                out["retrieve_stage"] = {"error": repr(exc)}
        if not args.no_encoder and world == 1:
            ix.close()  # the search index is no longer needed: give the HBM back before the encoder leg
            try:
                out.update(encoder_leg(args, local_rank))
                out.update(encoder_leg(args, local_rank, arch="nomic"))
                out.update(encoder_leg(args, local_rank, arch="e5_large"))
                out.update(encoder_leg(args, local_rank, arch="gte"))
                out.update(encoder_leg(args, local_rank, arch="jina"))
            except Exception as exc:
                out["encoder_error"] = repr(exc)
        if not args.no_encoder and world == 1 and args.encode_stage_passages > 0:
            try:
                out["encode_stage"] = encode_stage_leg(args, local_rank)
            except Exception as exc:
                out["encode_stage"] = {"error": repr(exc)}
        if not args.no_encoder and world == 1:
            try:
                out["rerank"] = rerank_leg(args, local_rank)
            except Exception as exc:
                out["rerank"] = {"error": repr(exc)}
        if not args.no_splade and world == 1:
            ix.close()
            try:
                out.update(splade_legs(args, local_rank))
                if out.get("splade_search", {}).get("parity_check") != "pass":
                    out["parity_check"] = "FAIL (splade_search leg)"
            except Exception as exc:
                out["splade_error"] = repr(exc)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, dim, k)
            except Exception as exc:
                out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": bergen_amd.utils.cpu_budget(), "kind": "port", "sample": "failed: " + repr(exc)}
        out["roofline"]["secondary"] = secondary_summary(out)
        print(json.dumps(out), flush=True)
    stage.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and parity == "FAIL":
        sys.exit(3)


def secondary_summary(out):
    """The other legs' headline figures as flat scalars INSIDE the roofline object: the driver's record keeps `roofline`, `config` and
    `cpu_baseline` in full and only the key names of everything else, so the second half of BASELINE.json's metric (passages
    encoded per second) and the configs[3] / configs[4] figures were not visible in BENCH_r04.json.  None = that leg did not run."""
    def get(*path):
        cur = out
        for key in path:
            if not isinstance(cur, dict) or key not in cur:
                return None
            cur = cur[key]
        return cur
    sec = {
        "passages_per_s": get("passages_per_s"), "encoder_frac": get("encoder_roofline", "frac"),
        "encoder_ms_per_512": get("encoder", "ms_per_step_kernels"), "encoder_steps": get("encoder", "steps"),
        "encode_stage_passages_per_s": get("encode_stage", "workers_threads_4", "passages_per_s"),
        "e5_large_passages_per_s": get("e5_large_encode", "passages_per_s"), "e5_large_frac": get("e5_large_encode", "roofline", "frac"),
        "nomic_passages_per_s": get("nomic_encode", "passages_per_s"), "nomic_frac": get("nomic_encode", "roofline", "frac"),
        "gte_passages_per_s": get("gte_encode", "passages_per_s"), "gte_frac": get("gte_encode", "roofline", "frac"),
        "jina_passages_per_s": get("jina_encode", "passages_per_s"), "jina_frac": get("jina_encode", "roofline", "frac"),
        "rerank_stage_deberta_pairs_per_s": get("rerank", "deberta_v3_large_shape", "through_rerank_eval", "pairs_per_s"),
        "rerank_stage_deberta_frac": get("rerank", "deberta_v3_large_shape", "through_rerank_eval", "frac"),
        "rerank_stage_bert_large_pairs_per_s": get("rerank", "bert_large_shape", "through_rerank_eval", "pairs_per_s"),
        "rerank_stage_bert_large_frac": get("rerank", "bert_large_shape", "through_rerank_eval", "frac"),
        "splade_queries_per_s": get("splade_search", "queries_per_s"), "splade_frac": get("splade_search", "roofline", "frac"),
        "splade_queries": get("splade_search", "queries"), "splade_full_list_gate": get("splade_search", "full_list_gate", "ids_and_fp32_scores_bit_exact"),
        "splade_encode_passages_per_s": get("splade_encode", "passages_per_s"),
        "config5_queries_per_s": get("config5", "queries_per_s"), "config5_frac": get("config5", "roofline", "frac"),
        "config5_frac_binding": get("config5", "roofline", "frac_binding"), "config5_parity": get("config5", "parity_check"),
        "real_size_queries_per_s": get("real_size", "queries_per_s"), "real_size_frac": get("real_size", "roofline", "frac"),
        "retrieve_stage_queries_per_s": get("retrieve_stage_full", "queries_per_s"),
        "rerank_deberta_pairs_per_s": get("rerank", "deberta_v3_large_shape", "pairs_per_s"),
        "rerank_deberta_frac": get("rerank", "deberta_v3_large_shape", "roofline", "frac"),
        "rerank_bert_pairs_per_s": get("rerank", "bert_large_shape", "pairs_per_s"),
        "rerank_bert_frac": get("rerank", "bert_large_shape", "roofline", "frac"),
        "rerank_bert_256_pairs_per_s": get("rerank", "bert_large_shape_256_pairs", "pairs_per_s"),
        "rerank_bert_256_frac": get("rerank", "bert_large_shape_256_pairs", "roofline", "frac"),
        "parity_check_all_legs": get("parity_check"),
    }
    return {k_: (round(v, 4) if isinstance(v, float) else v) for k_, v in sec.items()}


def _regenerate_rows(global_rows, dim, queries, plant_rows, n_total, device):
    """Rebuild specific corpus rows on the device exactly as fill_shard made them -> numpy fp16 [len, dim]."""
    out = np.empty((len(global_rows), dim), np.float16)
    by_block = {}
    for pos, r in enumerate(global_rows.tolist()):
        by_block.setdefault(r // BLOCK, []).append((pos, r))
    for b, items in by_block.items():
        rows, _ = corpus_block(b, dim, queries, plant_rows, n_total, device)
        for pos, r in items:
            out[pos] = rows[r - b * BLOCK].cpu().numpy()
        del rows
    return out


def sweep(ix, queries, k, args):
    """Kernel-variant timings for tuning (not part of the headline number)."""
    from bergen_amd import _lib
    res = []
    for split in (2, 1):
        for share in (1, 0):
            for nt in (1, 0):
                _lib.set_option("query_split", split)
                _lib.set_option("share_threshold", share)
                _lib.set_option("nontemporal", nt)
                ix.search(queries, k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ix.search(queries, k)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                c = ix.counters()
                per = c["algorithmic_bytes"] / c["n_passes"]
                res.append({"query_split": split, "share": share, "nt": nt, "qps": queries.shape[0] / dt,
                            "n_passes": c["n_passes"],
                            "scan_ms_per_pass": c["scan_ms"] / c["n_passes"], "merge_ms_per_pass": c["merge_ms"] / c["n_passes"],
                            "scan_GBps": per / (c["scan_ms"] / c["n_passes"] * 1e-3) / 1e9})
    _lib.set_option("query_split", args.query_split if args.query_split is not None else 1)
    _lib.set_option("query_tile", 128)
    _lib.set_option("share_threshold", 1)
    _lib.set_option("nontemporal", 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w"), indent=1)
    return res


if __name__ == "__main__":
    main()
