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


def _takes(fn, name):
    import inspect
    try:
        return name in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


class ShardedSearcher:
    """local_index: object with .search(queries, k, id_offset) -> (scores [Q,k] f32, ids [Q,k] i64)
    as torch tensors on `device` (a bergen_amd.FlatIndex fed device tensors does exactly that).
    merge: callable([G,Q,k] scores, [G,Q,k] ids) -> ([Q,k], [Q,k]); default = the HIP merge kernel.
    """

    def __init__(self, local_index, row_lo, rank=None, world_size=None, merge=None, group=None, dst=0, device=None,
                 exercise_collective=False):
        # device: where the collective's buffers live when the queries are host arrays (numpy / CPU tensors): RCCL moves
        # device memory only, gloo host memory.  None = the queries' own device.
        # exercise_collective: run the gather / merge / broadcast path even at world size 1 (RCCL accepts a one-rank group: the
        # one-GPU test of the transport, tests/test_gpu_nccl.py) instead of returning the local lists directly.
        self.local_index = local_index
        self.row_lo = int(row_lo)
        self.group = group
        self.device = torch.device(device) if device is not None else None
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self.dst = dst
        self.exercise_collective = bool(exercise_collective)
        if merge is None:
            from .index import merge_topk as merge  # HIP kernel; no CPU fallback
        self.merge = merge
        self._buf = {}
        self._status = None
        self._folded = None
        self._search_takes_out = _takes(local_index.search, "out")
        self._merge_takes_out = _takes(merge, "out")
        # gloo moves host memory only: under it the collective's buffers live on the host whatever the queries' device (the local
        # lists are copied down, gathered and merged there — bh_merge_topk takes host lists), e.g. several processes sharing one
        # GPU, or a host without RCCL.  RCCL ("nccl") gathers device memory.
        try:
            self._host_collective = (self.world_size > 1 or self.exercise_collective) and dist.is_initialized() and \
                str(dist.get_backend(group)) == "gloo"
        except Exception:  # noqa: BLE001
            self._host_collective = False

    def _buffers(self, nq, k, device):
        """Per (nq, k) buffers, allocated once: this rank's packed (scores | ids | status word) lists — the local search writes
        straight into them —, the gathered lists of all ranks, their dense [G, nq, k] forms for the merge and the merged result."""
        key = (nq, k, str(device))
        buf = self._buf.get(key)
        if buf is None:
            nbs = nq * k * 4
            ids_at = (nbs + 7) // 8 * 8  # int64 view needs an 8-byte offset
            status_at = ids_at + nq * k * 8
            per = status_at + 8
            packed = torch.empty(per, dtype=torch.uint8, device=device)
            buf = {
                "per": per, "ids_at": ids_at, "status_at": status_at, "packed": packed,
                "scores": packed[:nbs].view(torch.float32).view(nq, k),
                "ids": packed[ids_at:status_at].view(torch.int64).view(nq, k),
                "status": packed[status_at:].view(torch.int64),   # 0 = fine, 1 = this rank's local search failed, 2 = the merge failed
                "flat": torch.empty(self.world_size * per, dtype=torch.uint8, device=device),
            }
            if self.rank == self.dst:
                buf["all_s"] = torch.empty((self.world_size, nq, k), dtype=torch.float32, device=device)
                buf["all_i"] = torch.empty((self.world_size, nq, k), dtype=torch.int64, device=device)
                buf["out"] = (torch.empty((nq, k), dtype=torch.float32, device=device),
                              torch.empty((nq, k), dtype=torch.int64, device=device))
            if len(self._buf) >= 4:  # a handful of query-set shapes at most
                self._buf.pop(next(iter(self._buf)))
            self._buf[key] = buf
        return buf

    def search(self, queries, k, broadcast=False, check=True):
        """All ranks call this with the same queries.  Returns (scores, ids) on rank `dst`, None elsewhere — or, with
        broadcast=True, the merged lists on EVERY rank (one more small collective: a pipeline that runs the same script on
        every rank, as BERGEN under torchrun would, needs the result everywhere).  The result tensors are reused by the
        next search of the same shape: copy them if they must outlive it.
        Failures travel OUT OF BAND in a status word behind the packed lists: a rank whose local search raised still takes part in
        the collectives (nobody is left waiting in them), marks its word, and raises afterwards; every other rank sees the word
        in the gathered buffer, rank `dst` folds a failed merge into the word it broadcasts.  check=True reads the word before
        returning (one 8-byte device-to-host copy, i.e. a synchronisation); check=False leaves the results asynchronous — call
        check_last() when they are moved to the host."""
        if self.world_size == 1 and not self.exercise_collective:
            scores, ids = self.local_index.search(queries, k, id_offset=self.row_lo)
            self._status = None
            return torch.as_tensor(scores), torch.as_tensor(ids)
        nq = int(queries.shape[0])
        on_device = isinstance(queries, torch.Tensor) and queries.is_cuda
        device = queries.device if on_device else (self.device or torch.device("cpu"))
        if self._host_collective:
            device = torch.device("cpu")
        buf = self._buffers(nq, k, device)
        local_error = None
        try:
            if self._search_takes_out and on_device and not self._host_collective:
                # (out= is the device-queries path of FlatIndex.search: the local lists land in the packed send buffer)
                self.local_index.search(queries, k, id_offset=self.row_lo, out=(buf["scores"], buf["ids"]))
            else:
                scores, ids = self.local_index.search(queries, k, id_offset=self.row_lo)
                buf["scores"].copy_(torch.as_tensor(scores))
                buf["ids"].copy_(torch.as_tensor(ids))
            buf["status"].zero_()
        except Exception as e:  # noqa: BLE001 — the collectives below must still be entered, or the other ranks wait for ever
            local_error = e
            buf["status"].fill_(1)
        # (score, id) lists + status word in one byte buffer -> a single collective per search
        dist.all_gather_into_tensor(buf["flat"], buf["packed"], group=self.group)  # 1-D in/out: valid for RCCL and gloo
        gathered = buf["flat"].view(self.world_size, buf["per"])
        self._status = gathered[:, buf["status_at"]:].contiguous().view(torch.int64).view(-1)    # one word per rank
        self._folded = None
        if self.rank != self.dst:
            res = self._broadcast(buf, None) if broadcast else None
        else:
            nbs = nq * k * 4
            buf["all_s"].view(self.world_size, nq * k).copy_(gathered[:, :nbs].view(torch.float32))
            buf["all_i"].view(self.world_size, nq * k).copy_(gathered[:, buf["ids_at"]:buf["status_at"]].view(torch.int64))
            merged, merge_error = None, None
            try:
                if self._merge_takes_out:
                    out_s, out_i = self.merge(buf["all_s"], buf["all_i"], out=buf["out"])
                else:
                    out_s, out_i = self.merge(buf["all_s"], buf["all_i"])
                merged = (torch.as_tensor(out_s), torch.as_tensor(out_i))
            except Exception as e:  # noqa: BLE001
                merge_error = e
            if broadcast:  # the other ranks are already waiting in the broadcast: a failure reaches them through the status word
                res = self._broadcast(buf, merged, merge_failed=merge_error is not None, gathered_status=self._status)
            else:
                res = merged
            if merge_error is not None and local_error is None:
                raise RuntimeError(f"merge of the {self.world_size} shards' lists failed on rank {self.rank}: {merge_error}") from merge_error
        if local_error is not None:
            raise local_error
        if check:
            self.check_last()
        return res

    def check_last(self):
        """Raise if the last search failed anywhere (a rank's local search, or the merge on rank `dst`).  Reads one status word
        per rank from the gathered buffer (+ the word rank `dst` broadcast): a device-to-host copy, free when the results are
        being copied anyway."""
        if self._status is None:
            return
        words = self._status if self._folded is None else torch.cat([self._status, self._folded.view(-1)])
        st = words.cpu().tolist()
        bad = [r for r, v in enumerate(st[:self.world_size]) if v == 1]
        if bad:
            raise RuntimeError(f"sharded search: the local search failed on rank(s) {bad} (see their errors); the merged lists are invalid")
        if any(v == 2 for v in st):
            raise RuntimeError(f"sharded search: the merge of the shards' lists failed on rank {self.dst} (see its error)")

    def _broadcast(self, buf, merged, merge_failed=False, gathered_status=None):
        """Merged lists from rank `dst` to every rank, through the packed send buffer (its local lists are spent).  The status
        word travels with them: rank `dst` writes max(every rank's word, 2 if its merge failed) before sending."""
        if merged is not None:
            buf["scores"].copy_(merged[0])
            buf["ids"].copy_(merged[1])
        if self.rank == self.dst:
            if merge_failed:
                buf["status"].fill_(2)
            elif gathered_status is not None:
                buf["status"].copy_(gathered_status.max().view(1))
        src = dist.get_global_rank(self.group, self.dst) if self.group is not None else self.dst
        dist.broadcast(buf["packed"], src=src, group=self.group)
        self._folded = buf["status"]
        return buf["scores"], buf["ids"]
