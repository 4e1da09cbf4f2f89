"""Drives bench.run() — the bench's whole control flow: shard ranges, corpus fill, ShardedSearcher (one all-gather of the
packed top-k lists, merge on rank 0), barrier + max-over-ranks timing, the parity gate, the JSON line — on CPU stand-ins
under torchrun with the gloo backend.  The stand-in index is backed by the oracle (test infrastructure: this file lives
under tests/); bench.py itself only ever builds its HIP environment.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tests/bench_standin.py --gpus 2 --steps 1 --warmup 1 --n-rows 20000 --queries 33 ...
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import c_oracle  # noqa: E402


class OracleIndex:
    """FlatIndex's interface (upload / finalize / search / counters / close) over the oracle's canonical search."""

    def __init__(self, n_rows, dim):
        self.rows = np.zeros((n_rows, dim), np.float16)
        self.dim = dim
        self._c = {}

    def upload(self, rows, row0=None):
        r0 = 0 if row0 is None else int(row0)
        self.rows[r0:r0 + rows.shape[0]] = torch.as_tensor(rows).cpu().numpy().astype(np.float16)
        return self

    def finalize(self):
        return self

    def search(self, queries, k, id_offset=0, out=None):
        q = torch.as_tensor(queries).cpu().numpy().astype(np.float16)
        s, i = c_oracle.canonical_search(q, self.rows, k, id_offset=id_offset)
        nq = q.shape[0]
        self._c = {"n_passes": 1, "query_tile": 256, "n_workgroups": 256, "scan_ms": 1.0, "merge_ms": 0.1, "total_ms": 1.1,
                   "algorithmic_bytes": float(self.rows.size * 2 + nq * self.dim * 2 + nq * k * 12), "uncertified_queries": 0,
                   "shader_mhz": 0.0}
        s, i = torch.from_numpy(s), torch.from_numpy(i)
        if out is not None:
            out[0].copy_(s)
            out[1].copy_(i)
            return out
        return s, i

    def counters(self):
        return dict(self._c)

    def close(self):
        pass


def oracle_merge(scores, ids):
    s, i = c_oracle.merge_topk(np.ascontiguousarray(scores.numpy()), np.ascontiguousarray(ids.numpy()))
    return torch.from_numpy(s), torch.from_numpy(i)


class OracleEnv:
    backend = "gloo"
    merge = staticmethod(oracle_merge)
    results_to_host = False

    def __init__(self, local_rank):
        self.local_rank = local_rank
        self.device = torch.device("cpu")

    def init_dist(self, rank, world):
        import torch.distributed as dist
        dist.init_process_group(self.backend, rank=rank, world_size=world)

    def init_library(self, args):
        pass

    def make_index(self, n_rows, dim):
        return OracleIndex(n_rows, dim)

    def describe_backend(self):
        return {"search": "oracle stand-in (tests/bench_standin.py: control-flow test, not a measurement)"}

    make_stage = bench.HipEnv.make_stage  # the bench's own stage construction (Retrieve with search_rank / search_world)

    def sync(self):
        pass


if __name__ == "__main__":
    _args = bench.parse_args()
    bench.launch_ranks_if_needed(_args, script=__file__)  # `--gpus N` without a launcher starts its own ranks (bench.main does the same)
    bench.run(_args, OracleEnv(int(os.environ.get("LOCAL_RANK", "0"))))
