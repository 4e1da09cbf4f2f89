"""GPU: the row-sharded stage with TWO PROCESSES on one GPU (gloo carries the collective through host buffers; RCCL refuses two
ranks on one device) — the real `bergen_amd.Retrieve`, the real HIP indexes and merge kernel in every process, chunk folders cut
across the shard boundary, dense (ip / cos) and sparse: every rank's return dict equals the single-process stage's and the oracle's
lists.  What the 8-GPU run adds on top of this is RCCL as the transport."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_retrieve_sharded_gloo import _Plug, _corpus, _dataset, _write_folders

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, root, kind, n, nq, k, results):
    import bergen_amd
    from bergen_amd.sharded import shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stage = bergen_amd.Retrieve(init_args=_Plug(kind), batch_size=64, batch_size_sim=7, num_workers=0, search_world="auto",
                                    search_results=results, device=0)
        out = stage.retrieve(_dataset(n, nq), os.path.join(root, "q"), os.path.join(root, "d"), k)
        lo, hi = shard_range(n, rank, world)
        assert stage._resident[os.path.join(root, "d")][1][4] == (lo, hi)
        if results == "rank0" and rank != 0:
            assert out is None
        else:
            with open(os.path.join(root, f"out{rank}.pkl"), "wb") as f:
                pickle.dump(out, f)
        dist.barrier()
        stage.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,results", [("ip", 2, "all"), ("cos", 2, "rank0"), ("sparse", 2, "all"), ("ip", 3, "rank0")])
def test_two_processes_one_gpu_sharded_stage(tmp_path, kind, world, results):
    import bergen_amd
    from oracle import c_oracle, compare
    n, nq, d, k = (3000, 17, 300, 9) if kind == "sparse" else (40_001, 70, 768, 50)
    x, q = _corpus(kind, n, nq, d)
    root = str(tmp_path)
    q_path, d_path = _write_folders(root, kind, x, q, cuts=[n // 5, n // 5 + 1, (2 * n) // 3])
    single_stage = bergen_amd.Retrieve(init_args=_Plug(kind), batch_size=64, batch_size_sim=7, num_workers=0, device=0)
    single = single_stage.retrieve(_dataset(n, nq), q_path, d_path, k)
    single_stage.close()
    mp.spawn(_worker, args=(world, _free_port(), root, kind, n, nq, k, results), nprocs=world, join=True)
    for r in (range(world) if results == "all" else [0]):
        with open(os.path.join(root, f"out{r}.pkl"), "rb") as f:
            got = pickle.load(f)
        assert got["q_id"] == [f"q{i}" for i in range(nq)] and isinstance(got["doc_id"][0][0], str)
        assert torch.equal(got["score"], single["score"]) and got["doc_id"] == single["doc_id"], f"rank {r} of {world}"
    if kind == "sparse":
        from bergen_amd.sparse import _csr_from_any
        ws, wi = c_oracle.sparse_canonical_search(*_csr_from_any(x.numpy(), d), d, q.numpy(), k)
    else:
        xq, xd = q.numpy(), x.numpy()
        if kind == "cos":
            xq, xd = c_oracle.l2_normalize_rows(xq), c_oracle.l2_normalize_rows(xd)
        ws, wi = c_oracle.canonical_search(xq, xd, k)
    got_i = np.array([[int(s[3:]) for s in row] for row in single["doc_id"]])
    compare.assert_bit_exact(single["score"].numpy(), got_i, ws, wi, f"stage {kind}")


def test_bench_multi_rank_run_on_one_gpu():
    """`python bench.py --gpus 2 --dist-backend gloo` typed by hand on a one-GPU box: bench.py starts its two ranks itself (the
    driver's own torch.distributed.run command), both on cuda:0, and runs exactly its N > 1 path — shard fill on the device, the
    stage's search_rows on each shard, the gather of the packed lists (through host buffers here, RCCL in the driver's run), the HIP
    merge on rank 0, barrier + max-over-ranks timing, ONE JSON line with the parity gate on the merged lists."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for world in (2, 3):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--dist-backend", "gloo", "--steps", "2",
                              "--warmup", "1", "--n-rows", "300011", "--queries", "300", "--no-cpu-baseline"],
                             capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        r = json.loads(lines[0])
        assert r["n_gpus"] == world and r["parity_check"] == "pass" and r["scaling"] == "strong"
        assert r["config"]["rows_per_gpu"] == -(-300011 // world) and f"row-shard x{world}" in r["config"]["parallelism"]
        assert r["value"] > 0 and abs(r["value"] - 300 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
        assert r["roofline"]["frac"] > 0 and r["uncertified_queries"] == 0
