"""
Row-sharded multi-GPU exact search (BASELINE.json configs[2]).

The reference has no multi-GPU search at all (its only multi-GPU mechanism is
torch.nn.DataParallel around the encoder, models/retrievers/dense.py:32-35); it does however
already decompose the search over chunk files sequentially and merge partial top-k lists
(modules/retrieve.py:152-177).  Here the same decomposition runs in parallel:

  * one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI);
  * rank r holds the contiguous row range shard_range(N, r, G) resident in its HBM;
  * queries are replicated; each rank returns its local top-k with GLOBAL row ids (id_offset);
  * ONE all-gather of the packed [Q, k] (fp32 score, int64 id) lists — 12*Q*k bytes per rank,
    latency-bound on xGMI, no ring all-reduce anywhere — then the canonical merge on rank 0.

Because every shard computes canonical scores (a function of the query and the row only), the
merged result is bit-identical for any number of shards.
"""
import torch
import torch.distributed as dist


def shard_range(n_rows, rank, world_size):
    """Contiguous rows [lo, hi) of rank `rank`: ceil(N/G) per rank, last ranks may be short/empty."""
    per = (n_rows + world_size - 1) // world_size
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi


class ShardedSearcher:
    """local_index: object with .search(queries, k, id_offset) -> (scores [Q,k] f32, ids [Q,k] i64)
    as torch tensors on `device` (a bergen_amd.FlatIndex fed device tensors does exactly that).
    merge: callable([G,Q,k] scores, [G,Q,k] ids) -> ([Q,k], [Q,k]); default = the HIP merge kernel.
    """

    def __init__(self, local_index, row_lo, rank=None, world_size=None, merge=None, group=None, dst=0):
        self.local_index = local_index
        self.row_lo = int(row_lo)
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self.dst = dst
        if merge is None:
            from .index import merge_topk as merge  # HIP kernel; no CPU fallback
        self.merge = merge

    def search(self, queries, k):
        """All ranks call this with the same queries.  Returns (scores, ids) on rank `dst`, None elsewhere."""
        scores, ids = self.local_index.search(queries, k, id_offset=self.row_lo)
        scores = torch.as_tensor(scores)
        ids = torch.as_tensor(ids)
        nq = scores.shape[0]
        if self.world_size == 1:
            return scores, ids
        # pack (score, id) into one byte buffer -> a single collective per search
        packed = torch.cat([scores.contiguous().view(torch.uint8).reshape(-1),
                            ids.contiguous().view(torch.uint8).reshape(-1)])
        flat = torch.empty(self.world_size * packed.numel(), dtype=torch.uint8, device=packed.device)
        dist.all_gather_into_tensor(flat, packed, group=self.group)  # 1-D in/out: valid for RCCL and gloo
        gathered = flat.view(self.world_size, packed.numel())
        if self.rank != self.dst:
            return None
        nbs = nq * k * 4
        all_s = gathered[:, :nbs].contiguous().view(torch.float32).reshape(self.world_size, nq, k)
        all_i = gathered[:, nbs:].contiguous().view(torch.int64).reshape(self.world_size, nq, k)
        out_s, out_i = self.merge(all_s, all_i)
        return torch.as_tensor(out_s), torch.as_tensor(out_i)
