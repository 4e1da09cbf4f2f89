"""
Retrieve — the retrieval stage (seam 1 of SURVEY §8b), signature- and return-compatible with the
reference's modules/retrieve.py:20-197, backed by the resident HBM index and the fused gfx950
search kernels (C ABI: include/bergen_hip.h).

Reference -> here
  Retrieve.__init__                          retrieve.py:21-34
  Retrieve.index                             retrieve.py:37-50
  Retrieve.retrieve                          retrieve.py:52-108
  Retrieve.encode_and_save                   retrieve.py:110-144   (same chunk files: names, cadence, dtype)
  Retrieve.load_collection_and_retrieve      retrieve.py:146-185
  get_clean_model_name / get_chunk_path      retrieve.py:193-197

What changes underneath (SURVEY Appendix A "deliberately differ"):
  * the doc chunks are uploaded to HBM once and stay resident across query sets (the reference
    re-uploads every chunk for every query chunk, retrieve.py:153);
  * scores are the canonical fp32 scores and ties are ordered (score desc, row asc) instead of
    torch.topk's arbitrary order;
  * doc-id strings are produced only for the Q*k hits.
Not carried over: the bm25 / oracle_provenance branches (Pyserini/JVM and a no-op; SURVEY §2 OOS).
"""
import os
import sys

import torch
from torch.utils.data import DataLoader
from tqdm import tqdm

from . import utils
from .config import instantiate
from .index import FlatIndex
from .sharded import ShardedSearcher, shard_range
from .sparse import SparseIndex

_INCOMPLETE = '!!! Index is not complete. Please re-index. Missing {} documents in the index. !!!'


def _metric_of(model):
    sim = getattr(model, "similarity", None)
    metric = getattr(sim, "metric", None)
    if metric is not None:
        return metric
    name = type(sim).__name__ if not isinstance(sim, type) else sim.__name__
    if name == "CosineSim":
        return "cos"
    if name == "DotProduct":
        return "ip"
    raise ValueError(f"cannot derive the search metric from similarity {sim!r}; expected DotProduct or CosineSim")


def _column(ds, name):
    """dataset['id'] for HF datasets, dicts of lists, or anything indexable by column name."""
    return ds[name]


class Retrieve:
    def __init__(self,
                 init_args=None,
                 batch_size=128,
                 batch_size_sim=1024,
                 pyserini_num_threads=1,
                 continue_batch=None,
                 device=0,
                 num_workers=4,
                 encode_rank=0,
                 encode_world=1,
                 resident_on_encode=False,
                 search_rank=None,
                 search_world=None,
                 search_results="all",
                 loader="threads",
                 require_native=False):
        # encode_rank / encode_world: multi-GPU encoding.  The reference's only multi-GPU mechanism is
        # torch.nn.DataParallel around the encoder (dense.py:32-35: scatter inputs, re-broadcast every weight and
        # gather [B, T, d] outputs to GPU 0 on every forward).  Here each of `encode_world` processes (one per GPU)
        # encodes a contiguous range of BATCHES of the dataset into the same index folder — no collective; the
        # union of the chunk files is a valid index (same naming, ordering by the integer in the file name).
        # resident_on_encode (opt-in, single process, dense): while index() encodes the documents, every batch is ALSO
        # copied device-to-device into the resident HBM index, so that retrieve() searches it without reading the folder
        # back (the reference writes the chunk files and then loads them again, retrieve.py:110-144 -> 153).  The folder is
        # written all the same: it is the cache the next run finds.
        # search_rank / search_world: row-sharded multi-GPU SEARCH behind this same stage object (BASELINE configs[2] / [4];
        # the reference has no multi-GPU search).  One process per GPU, every process calls retrieve() with the
        # same arguments; process r keeps rows shard_range(N, r, world) of the document folder resident in its HBM, runs the
        # same fused search with id_offset = its first row, ONE all-gather (RCCL over xGMI) brings the partial top-k lists
        # to rank 0, which merges them (bergen_amd/sharded.py).  search_world="auto" takes rank and world size from an
        # initialised torch.distributed; None / 1 = the single-GPU path.  search_results: "all" = every rank returns the
        # same dict (one more small broadcast; what a pipeline run under torchrun needs), "rank0" = rank 0 returns the dict,
        # the other ranks None.
        if search_world == "auto":
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            search_world = dist.get_world_size() if on else 1
            search_rank = dist.get_rank() if on else 0
        self.search_world = 1 if search_world is None else int(search_world)
        self.search_rank = 0 if search_rank is None else int(search_rank)
        if not 0 <= self.search_rank < self.search_world:
            raise ValueError(f"search_rank={self.search_rank} outside 0..{self.search_world - 1}")
        if search_results not in ("all", "rank0"):
            raise ValueError(f"search_results={search_results!r}: expected 'all' or 'rank0'")
        self.search_results = search_results
        self.search_group = None  # a torch.distributed process group, if the search ranks are not the default group
        # loader: how encode_and_save tokenises ahead of the GPU.  "processes" = the reference's DataLoader worker processes
        # (retrieve.py:114-118).  "threads" (default) = `num_workers` THREADS of this process run collate_fn on the coming
        # batches, in order: the Rust tokenizers release the GIL and keep their own parallelism, which a forked DataLoader
        # worker loses ("the current process just got forked ... disabling parallelism"), nothing is pickled, no worker
        # start-up per index() call.  Measured with bench.py's encode_stage leg.
        if loader not in ("threads", "processes"):
            raise ValueError(f"loader={loader!r}: expected 'threads' or 'processes'")
        self.loader = loader
        self.resident_on_encode = bool(resident_on_encode)
        self.encode_rank = int(encode_rank)
        self.encode_world = int(encode_world)
        self.continue_batch = continue_batch
        self.batch_size = batch_size
        self.batch_size_sim = batch_size_sim
        self.pyserini_num_threads = pyserini_num_threads
        self.device = device
        self.num_workers = num_workers
        # instantiate model (reference: hydra instantiate, retrieve.py:34).  rag.py hands over an OmegaConf DictConfig;
        # config.instantiate converts that (and any other mapping) and passes an already-built plug-in object through
        self.model = instantiate(init_args)
        # require_native: a run on an architecture the HIP forward pass does not cover (gte-*-v1.5, jina-v2, repllama ...) keeps its
        # HF torch encoder with one warning by default; with this switch the stage refuses to start instead (the plug-ins take the
        # same switch themselves — Dense(require_native=True), BERGEN_AMD_REQUIRE_NATIVE=1 — and then fail at load time)
        # BOTH encoders are checked (an asymmetric Dense / Splade whose query encoder fell back must not pass), from the encoder
        # objects themselves: a plug-in that does not say what it runs on (a stock reference one) is judged by its `.model`
        if require_native:
            from .dense import plugin_backends
            sides = plugin_backends(self.model)
            if any(v != "hip" for v in sides.values()):
                raise RuntimeError(f"bergen_amd.Retrieve(require_native=True): the encoder of {getattr(self.model, 'model_name', self.model)!r} "
                                   f"runs on the HF torch implementation {sides} "
                                   f"({getattr(self.model, 'fallback_reason', None) or 'see the warning above'})")
        # host-side thread pools (torch intra-op, OpenMP, the tokenizer's rayon pool) follow the container's CPU quota, not the host's
        # CPU count (utils.cpu_budget: a GPU pod that shows 256 CPUs under a quota of 16 gets frozen by the CFS throttle otherwise);
        # explicit OMP_NUM_THREADS / RAYON_NUM_THREADS settings of the user win
        self.cpu_budget = utils.fit_host_pools_to_cpu_budget()
        self._resident = {}  # doc_embeds_path -> (FlatIndex, signature)
        self._searchers = {}

    # ------------------------------------------------------------------ indexing (encode)
    def index(self, dataset, index_path, query_or_doc, overwrite_index=False):
        """Encode `dataset[query_or_doc]` into `index_path` unless the folder already exists (cache by existence;
        `continue_batch` and `overwrite_index` force the encode) — reference retrieve.py:37-50."""
        if self.encode_world > 1:
            # Several processes fill ONE folder: its existence says nothing about this rank's range (the first rank to
            # get here creates it).  A folder that already holds every row — written by a single process, another world
            # size or batch size, merge_indexes or a download — is a cached index for every rank (the reference caches by
            # existence, retrieve.py:40); otherwise this rank is done when the chunk file named after ITS last batch exists.
            n_rows = len(dataset[query_or_doc])
            n_batches = (n_rows + self.batch_size - 1) // self.batch_size
            b_lo, b_hi = self._batch_range(n_batches)
            if self.continue_batch is None and not overwrite_index:
                if b_hi <= b_lo or os.path.exists(self.get_chunk_path(index_path, b_hi - 1)):
                    return
                if self._folder_rows(index_path) == n_rows:
                    return
        else:
            have = os.path.exists(index_path)
            if have and self.continue_batch is None and not overwrite_index:
                return
        if self.model.model_name in ('bm25', 'oracle_provenance'):
            raise NotImplementedError(f"{self.model.model_name} is out of scope of the dense backend (SURVEY §2)")
        self._resident.pop(index_path, None)
        split = dataset[query_or_doc].remove_columns(['id'])
        self.encode_and_save(split, save_path=index_path, query_or_doc=query_or_doc)

    def _flush_chunk(self, save_path, last_batch, pieces):
        """One chunk file, named after the LAST batch it holds; sparse COO for SPLADE (retrieve.py:135-141,196-197)."""
        block = torch.cat(pieces)
        if getattr(self.model, 'sparse', False) or 'splade' in self.model.model_name:
            block = block.to_sparse()
        # written under a temporary name and renamed: other processes (multi-rank encoding, a concurrent reader) take the
        # existence of a chunk file as "complete"
        final = self.get_chunk_path(save_path, last_batch)
        torch.save(block, final + '.tmp')
        os.replace(final + '.tmp', final)

    @torch.no_grad()
    def encode_and_save(self, dataset, save_path, query_or_doc, chunk_size=150000):
        """Encode a dataset into embedding_chunk_<last_batch_idx>.pt files with the reference's cadence: a chunk is
        closed after batch i when i is a non-zero multiple of chunk_size // batch_size, and after the last batch
        (retrieve.py:110-144; first chunk 150000 // B + 1 batches, later ones 150000 // B)."""
        cadence = chunk_size // self.batch_size  # 0 for batch_size > chunk_size: ZeroDivisionError below, as in the reference
        n_batches = (len(dataset) + self.batch_size - 1) // self.batch_size
        os.makedirs(save_path, exist_ok=True)
        # this process's contiguous range of batches [b_lo, b_hi): everything for a single process
        b_lo, b_hi = self._batch_range(n_batches)
        source = dataset
        if self.encode_world > 1:
            from torch.utils.data import Subset
            source = Subset(dataset, range(b_lo * self.batch_size, min(len(dataset), b_hi * self.batch_size)))
        if self.loader == "threads" and self.num_workers > 0:
            loader = self._threaded_batches(source, query_or_doc)
        else:
            loader = DataLoader(source, batch_size=self.batch_size, num_workers=self.num_workers,
                                collate_fn=lambda rows: self.model.collate_fn(rows, query_or_doc))
        self.model.model = self.model.model.to('cuda' if torch.cuda.is_available() else 'cpu')
        pieces = []
        progress = tqdm(enumerate(loader, start=b_lo), total=b_hi - b_lo, desc=f'Encoding: {self.model.model_name}',
                        file=sys.stderr)
        direct = (self.resident_on_encode and query_or_doc == 'doc' and self.encode_world == 1 and self.continue_batch is None
                  and torch.cuda.is_available() and not getattr(self.model, 'sparse', False) and 'splade' not in self.model.model_name)
        resident, row = None, 0
        # chunk files are written BEHIND the encoder: torch.cat + torch.save of a 150 k-row chunk (230 MB) takes a few tenths of a
        # second in which the GPU would idle every 5 s of encoding; one writer thread (chunks stay in order), joined before the
        # call returns — a chunk's existence still means "complete" (temporary name + rename in _flush_chunk)
        from concurrent.futures import ThreadPoolExecutor
        writer = ThreadPoolExecutor(max_workers=1, thread_name_prefix="bergen-chunk-writer")
        writes = []
        try:
            import time as _time
            t_start = _time.perf_counter()
            t_first = t_last = None
            rows_first = rows_done = 0
            for i, batch in progress:
                if self.continue_batch is not None and i <= self.continue_batch:
                    continue  # resume: batches up to continue_batch were saved by an earlier run
                emb = self.model(query_or_doc, batch)['embedding'].detach()
                t_last = _time.perf_counter()
                rows_done += int(emb.shape[0])
                if t_first is None:
                    t_first, rows_first = t_last, rows_done
                if direct and emb.is_cuda and emb.ndim == 2:
                    if resident is None:
                        resident = FlatIndex(len(dataset), emb.shape[1], metric=_metric_of(self.model), device=self.device)
                    resident.upload(emb.contiguous(), row0=row)  # device to device, no host round trip
                    row += emb.shape[0]
                pieces.append(emb.cpu())
                if (i != 0 and i % cadence == 0) or i == b_hi - 1:
                    writes.append(writer.submit(self._flush_chunk, save_path, i, pieces))
                    pieces = []
            for w in writes:
                w.result()  # (re-raises a failed write here)
            # where the call's time went (bench.py's encode_stage leg reads it): the rate between the first and the last batch
            # leaving the encoder is the pipeline's steady state; what surrounds it — tokenising the first batch, writing the last
            # chunk — is paid once per call, whatever the corpus size
            t_end = _time.perf_counter()
            self.last_encode_stats = {
                "rows": rows_done, "seconds": t_end - t_start,
                "first_batch_seconds": (t_first - t_start) if t_first is not None else None,
                "last_chunk_write_seconds": (t_end - t_last) if t_last is not None else None,
                "steady_state_rows_per_s": ((rows_done - rows_first) / (t_last - t_first)) if t_first is not None and t_last > t_first else None}
            if resident is not None:
                if row != len(dataset):
                    raise IOError(_INCOMPLETE.format(len(dataset) - row))
                resident.finalize()
                files = utils.sorted_chunk_files(save_path)
                signature = (tuple(files), tuple(os.path.getmtime(f) for f in files), len(dataset), _metric_of(self.model), None)
                old = self._resident.pop(save_path, None)
                if old is not None:
                    old[0].close()
                self._resident[save_path] = (resident, signature)
                resident = None
        finally:
            writer.shutdown(wait=True)
            if resident is not None:
                resident.close()
        self.model.model = self.model.model.to('cpu')

    def _threaded_batches(self, source, query_or_doc):
        """collate_fn over consecutive batches of `source`, in order, `num_workers` batches being tokenised at any time on
        threads of this process while the caller runs the forward pass of an earlier one."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        n, bs = len(source), self.batch_size

        def make(lo, hi):  # rows [lo, hi) of the source through the model's collate_fn (tokeniser)
            rows = [source[j] for j in range(lo, hi)] if not hasattr(source, "select") else None
            if rows is None:
                cols = source[lo:hi]  # HF Dataset slice: dict of column lists
                names = list(cols)
                rows = [dict(zip(names, vals)) for vals in zip(*(cols[c] for c in names))]
            return self.model.collate_fn(rows, query_or_doc)

        pad_id = getattr(getattr(self.model, "tokenizer", None), "pad_token_id", None)

        def merge(parts):
            """Pieces of ONE batch, each padded to its own longest row, as the batch padded to ITS longest row — what
            collate_fn over the whole batch returns (padding="longest", right side)."""
            if len(parts) == 1:
                return parts[0]
            width = max(int(p["input_ids"].shape[1]) for p in parts)
            out = {}
            for key in parts[0].keys():
                fill = int(pad_id) if key == "input_ids" else 0
                out[key] = torch.cat([torch.nn.functional.pad(p[key], (0, width - int(p[key].shape[1])), value=fill) for p in parts])
            return type(parts[0])(out)

        # The tokeniser threads hold the GIL while they turn token ids into Python objects; the thread that drives the GPU
        # needs it for microseconds between two forward passes (the ctypes call itself runs without it).  CPython hands the
        # GIL over only every `switchinterval` (5 ms by default): a quarter of a 17 ms forward pass lost per hand-over, and more
        # tokeniser threads meant more of them (round 3: 16 threads slower than 4).  0.2 ms while the loader runs.
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 2e-4))
        # How many threads, and who parallelises.  Default: whole batches on `num_workers` threads, each call spread over the cores
        # by the Rust tokenizer's own pool (rayon) — with the ids-only fast path and the short switch interval above, 4 threads
        # feed the GPU at about its own rate (round 4, 256-core host: 28-30 k passages/s end to end, 34 k between the first and
        # the last batch; 16 threads 24 k: the callers share one pool).  BERGEN_AMD_TOKENIZER_PIECES=1 selects the other scheme —
        # every batch cut into eight PIECES, each tokenised serially (TOKENIZERS_PARALLELISM=false is read per call) by one of up
        # to 32 threads and merged back in order: no shared pool, but eight times the Python calls per batch under the GIL; it
        # measured 25-27 k on the same hosts.
        # (pools sized for the CPUs the container may use, not for the host's: utils.cpu_budget)
        from .utils import fit_host_pools_to_cpu_budget
        budget = fit_host_pools_to_cpu_budget()
        n_threads = max(int(self.num_workers), min(32, budget // 4))
        tok = getattr(self.model, "tokenizer", None)
        serial = (n_threads >= 8 and pad_id is not None and getattr(tok, "padding_side", "right") == "right"
                  and os.environ.get("BERGEN_AMD_TOKENIZER_PIECES", "0") == "1")
        n_pieces = 8 if serial else 1
        old_par = os.environ.get("TOKENIZERS_PARALLELISM")
        if serial:
            os.environ["TOKENIZERS_PARALLELISM"] = "false"
        else:
            n_threads = int(self.num_workers)

        pieces = [n_pieces]  # (1 from the moment a batch's pieces turn out not to be mergeable)

        def submit(pool, b0):
            hi = min(n, b0 + bs)
            step = -(-(hi - b0) // pieces[0])
            return b0, [pool.submit(make, lo, min(hi, lo + step)) for lo in range(b0, hi, step)]

        def mergeable(parts):
            return all(hasattr(p, "keys") and "input_ids" in p and all(torch.is_tensor(p[k_]) and p[k_].ndim == 2 for k_ in p.keys())
                       for p in parts)

        try:
            with ThreadPoolExecutor(max_workers=n_threads, thread_name_prefix="bergen-tokenize") as pool:
                pending, starts = deque(), iter(range(0, n, bs))
                ahead = max(2, 2 * n_threads // n_pieces)  # batches in flight
                for b0 in starts:
                    pending.append(submit(pool, b0))
                    if len(pending) >= ahead:
                        break
                while pending:
                    b0, futures = pending.popleft()
                    parts = [f.result() for f in futures]
                    if len(parts) > 1 and not mergeable(parts):  # (a collate_fn of another kind: whole batches from here on)
                        pieces[0] = 1
                        parts = [make(b0, min(n, b0 + bs))]
                    nxt = next(starts, None)
                    if nxt is not None:
                        pending.append(submit(pool, nxt))
                    yield merge(parts)
        finally:
            sys.setswitchinterval(old_interval)
            if serial:
                if old_par is None:
                    os.environ.pop("TOKENIZERS_PARALLELISM", None)
                else:
                    os.environ["TOKENIZERS_PARALLELISM"] = old_par

    def _batch_range(self, n_batches, rank=None):
        """Contiguous range of batches [b_lo, b_hi) that process `rank` of `encode_world` encodes."""
        rank = self.encode_rank if rank is None else rank
        share = -(-n_batches // self.encode_world)
        b_lo = min(n_batches, rank * share)
        return b_lo, min(n_batches, b_lo + share)

    @staticmethod
    def _folder_rows(index_path):
        """Rows held by the chunk files of a folder (files are mapped, not read), or -1 if a file cannot be opened —
        e.g. one another process is still writing under its temporary name."""
        try:
            return sum(int(utils.load_chunk(f, mmap=True).shape[0]) for f in utils.sorted_chunk_files(index_path))
        except Exception:
            return -1

    def wait_for_index(self, index_path, n_rows, timeout_s=None, poll_s=0.5):
        """Multi-process encoding: block until every rank's last chunk file is in `index_path` (each rank writes it after
        all its other chunks), or the folder holds all n_rows rows whatever wrote it.  No collective is involved — the
        ranks only share the folder.  A large corpus takes hours to encode, so there is no limit on the total wait; what
        raises TimeoutError is LACK OF PROGRESS: no new chunk file for `timeout_s` seconds (default:
        BERGEN_AMD_INDEX_WAIT_S or 3600)."""
        if self.encode_world <= 1:
            return
        import time
        n_batches = (n_rows + self.batch_size - 1) // self.batch_size
        need = []
        for r in range(self.encode_world):
            b_lo, b_hi = self._batch_range(n_batches, r)
            if b_hi > b_lo:
                need.append(self.get_chunk_path(index_path, b_hi - 1))
        limit = float(os.environ.get("BERGEN_AMD_INDEX_WAIT_S", "3600")) if timeout_s is None else timeout_s
        last_change, seen = time.time(), -1
        while True:
            missing = [f for f in need if not os.path.exists(f)]
            if not missing:
                return
            have = len(utils.sorted_chunk_files(index_path))
            if have != seen:
                if seen >= 0 or have:  # something was written since the last look
                    last_change = time.time()
                seen = have
                if self._folder_rows(index_path) == n_rows:
                    return
            if time.time() - last_change > limit:
                raise TimeoutError(f"index {index_path}: no new chunk file for {limit:.0f} s, still waiting for "
                                   f"{len(missing)} rank(s), e.g. {missing[0]}")
            time.sleep(poll_s)

    # ------------------------------------------------------------------ resident index
    # The index classes and the cross-shard merge are attributes so that the CPU tests of the multi-rank path can drive
    # this stage over gloo with oracle-backed stand-ins; the product never sets them (no CPU fallback).
    _dense_index_cls = FlatIndex
    _sparse_index_cls = SparseIndex
    _shard_merge = None  # ShardedSearcher's default: the HIP merge kernel

    @staticmethod
    def _rows_of(emb_chunk, a, b):
        """Rows [a, b) of a chunk tensor (dense, or sparse COO as SPLADE chunks are stored)."""
        if a == 0 and b == emb_chunk.shape[0]:
            return emb_chunk
        if not emb_chunk.is_sparse:
            return emb_chunk[a:b]
        t = emb_chunk.coalesce()
        idx = t.indices()
        keep = (idx[0] >= a) & (idx[0] < b)
        return torch.sparse_coo_tensor(torch.stack([idx[0][keep] - a, idx[1][keep]]), t.values()[keep],
                                       (b - a, t.shape[1])).coalesce()

    def _build_resident(self, chunk_iter, dataset_size, dim, metric, rows=None, sparse=False):
        """Upload rows [lo, hi) (default: all) of the chunk sequence into a new index of hi - lo rows.  The size check is
        the reference's (retrieve.py:165-166) and always covers the WHOLE folder, whatever part of it this process keeps."""
        lo, hi = (0, dataset_size) if rows is None else rows
        ix = (self._sparse_index_cls(hi - lo, dim, device=self.device) if sparse
              else self._dense_index_cls(hi - lo, dim, metric=metric, device=self.device))
        num_emb = 0
        try:
            for emb_chunk in chunk_iter:
                n_c = emb_chunk.shape[0]
                a, b = max(lo, num_emb), min(hi, num_emb + n_c)
                if b > a and num_emb + n_c <= dataset_size:
                    ix.upload(self._rows_of(emb_chunk, a - num_emb, b - num_emb), row0=a - lo)
                num_emb += n_c
            if num_emb != dataset_size:  # retrieve.py:165-166
                raise IOError(_INCOMPLETE.format(dataset_size - num_emb))
            ix.finalize()
        except Exception:
            ix.close()
            raise
        return ix

    @staticmethod
    def _dense_chunk(emb_chunk):
        if emb_chunk.is_sparse:
            emb_chunk = emb_chunk.to_dense()
        return emb_chunk

    def _resident_index(self, doc_embeds_path, dataset_size, metric, rows=None):
        """The resident index of this process: the whole folder, or rows [lo, hi) of it (sharded search)."""
        files = utils.sorted_chunk_files(doc_embeds_path)
        signature = (tuple(files), tuple(os.path.getmtime(f) for f in files), dataset_size, metric, rows)
        hit = self._resident.get(doc_embeds_path)
        if hit is not None and hit[1] == signature:
            return hit[0]
        if hit is not None:
            hit[0].close()
            del self._resident[doc_embeds_path]
        if not files:
            raise IOError(_INCOMPLETE.format(dataset_size))

        first = utils.load_chunk(files[0], mmap=True)
        dim = first.shape[1]
        sparse = bool(first.is_sparse)
        del first

        def chunks():
            # chunk i + 1 is read (mapped) on a worker thread while chunk i goes through the pinned staging buffers
            loaded = utils.prefetched(files, lambda f: utils.load_chunk(f, mmap=not sparse), depth=1)
            what = 'sparse embeddings' if sparse else 'embeddings'
            for emb in tqdm(loaded, total=len(files), desc=f'Load {what} into HBM...'):
                yield emb  # SPLADE chunks (retrieve.py:138-139) stay sparse: resident CSR index

        ix = self._build_resident(chunks(), dataset_size, dim, metric, rows=rows, sparse=sparse)
        self._resident[doc_embeds_path] = (ix, signature)
        return ix

    # ------------------------------------------------------------------ search
    def retrieve(self, dataset, query_embeds_path, doc_embeds_path, top_k_documents, return_docs=False,
                 overwrite_index=False):
        # index if index doesn't exist (retrieve.py:54-56)
        self.index(dataset, query_embeds_path, query_or_doc='query', overwrite_index=overwrite_index)
        self.index(dataset, doc_embeds_path, query_or_doc='doc', overwrite_index=overwrite_index)

        q_ids = _column(dataset['query'], 'id')
        if self.model.model_name == "bm25":
            raise NotImplementedError("bm25 is out of scope of the dense backend (SURVEY §2)")
        # several encoding processes share the folders: wait until every rank's range is there
        self.wait_for_index(query_embeds_path, len(dataset['query']))
        self.wait_for_index(doc_embeds_path, len(dataset['doc']))

        query_embeds = utils.load_embeddings(query_embeds_path)
        sparse_queries = bool(query_embeds.is_sparse)
        if sparse_queries:
            query_embeds = query_embeds.to_dense()  # [Q, vocab], as the reference holds it (retrieve.py:75-76)
        if hasattr(self.model, "model") and hasattr(self.model.model, "to"):
            self.model.model = self.model.model.to('cpu')  # free HBM for the index (retrieve.py:78)

        metric = "sparse" if (sparse_queries or getattr(self.model, "sparse", False)) else _metric_of(self.model)
        # k up to 4096 is served (one fused search up to 248 dense / 120 sparse, range by range above that); a larger k is
        # refused BEFORE the index is read and uploaded (the reference accepts any k; INTEGRATION.md "Limits")
        k_max = self._sparse_index_cls.MAX_K if metric == "sparse" else self._dense_index_cls.MAX_K
        if not 0 < int(top_k_documents) <= k_max:
            raise ValueError(f"top_k_documents={top_k_documents} outside 1..{k_max} supported by the {metric} search kernels")
        found = self.search_rows(query_embeds, doc_embeds_path, int(top_k_documents), metric, len(dataset['doc']))
        if found is None:  # search_results="rank0" on another rank
            return None
        return {"score": found[0], "q_id": q_ids, "doc_id": self._map_doc_ids(dataset['doc'], found[1])}

    def adopt_resident_index(self, doc_embeds_path, index, dataset_size, metric, rows=None):
        """Register an index that is ALREADY resident in HBM as the index of `doc_embeds_path` (rows = this process's
        (lo, hi) shard of it, None = all): retrieve() / search_rows() then search it without reading the folder — for
        callers that filled HBM some other way (device-to-device from an encoder, a synthetic corpus generated on the
        device).  The stage owns the index from here on (close() frees it)."""
        old = self._resident.pop(doc_embeds_path, None)
        if old is not None and old[0] is not index:
            old[0].close()
        self._resident[doc_embeds_path] = (index, ("adopted", dataset_size, metric, rows))

    def search_rows(self, query_embeds, doc_embeds_path, top_k_documents, metric, n_docs):
        """The search half of retrieve(): query embeddings [Q, d] (host tensor as load_embeddings returns it, or a device
        tensor) -> (scores fp32 [Q, k], rows int64 [Q, k]) as CPU tensors, canonical order — the reference's
        load_collection_and_retrieve over the whole folder (retrieve.py:81-101) without its per-chunk H2D copies.
        search_world > 1: every rank calls it; this rank searches rows shard_range(n_docs, rank, world) with global row
        ids, ONE all-gather of the partial lists, canonical merge on rank 0 (ShardedSearcher); with
        search_results="rank0" the other ranks get None."""
        k = int(top_k_documents)
        sparse = metric == "sparse"
        on_gpu = torch.cuda.is_available()
        device = torch.device("cuda", self.device) if on_gpu else torch.device("cpu")
        rows = shard_range(n_docs, self.search_rank, self.search_world) if self.search_world > 1 else None
        hit = self._resident.get(doc_embeds_path)
        if hit is not None and hit[1] == ("adopted", n_docs, metric, rows):
            index = hit[0]
        else:
            index = self._resident_index(doc_embeds_path, dataset_size=n_docs, metric=metric, rows=rows)
        # ONE search call for the whole query set: the index is resident, the library walks it once per query TILE
        # (256 / 192 / 128 queries, chosen by the kernel).  The reference's batch_size_sim split (retrieve.py:81) bounded
        # its [Bq, n] score matrix, which does not exist here; splitting by it (1024 = 4 x 256 exactly, but 5.3 x 192)
        # could only add corpus passes.  Sparse search keeps the split: its host side builds per-tile term tables.
        pieces = query_embeds.split(self.batch_size_sim, dim=0) if sparse else [query_embeds]
        if self.search_world == 1:
            parts = []
            for part in (tqdm(pieces, total=len(pieces), desc='Retrieving docs...') if sparse else pieces):
                part = part.contiguous()
                if part.is_cuda and not sparse:
                    s, i = index.search(part, k, host=True)  # the merge kernel writes the lists into pinned host memory
                    # (the index reuses those pinned buffers for its next search of this shape: hand out copies — 1.7 MB
                    # for the headline search)
                    parts.append((s.clone(), i.clone()))
                else:
                    s, i = index.search(part, k)
                    parts.append((torch.as_tensor(s), torch.as_tensor(i)))
        else:
            searcher = self._searcher(index, rows[0], device)
            everywhere = self.search_results == "all"
            parts = []
            for part in pieces:
                part = part.contiguous()
                if on_gpu and not sparse and not part.is_cuda:
                    part = part.to(device)  # (sparse search takes host queries: its host side builds the term tables)
                res = searcher.search(part, k, broadcast=everywhere, check=False)
                searcher.check_last()  # (the status words come down with — or, on ranks without results, instead of — the lists)
                if res is not None:  # (the searcher reuses its result buffers: .cpu() of a device tensor copies, host tensors are cloned)
                    parts.append((res[0].cpu(), res[1].cpu()) if res[0].is_cuda else (res[0].clone(), res[1].clone()))
            if not parts:
                return None
        if len(parts) == 1:
            return parts[0]
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])

    def _searcher(self, index, row_lo, device):
        """One ShardedSearcher per resident shard (its gather buffers are allocated once per query-set shape)."""
        key = id(index)
        hit = self._searchers.get(key)
        if hit is None or hit[0] is not index:
            self._searchers = {key: (index, ShardedSearcher(index, row_lo, rank=self.search_rank, world_size=self.search_world,
                                                            merge=self._shard_merge, group=self.search_group, device=device))}
            hit = self._searchers[key]
        return hit[1]

    @staticmethod
    def _ids_at(doc_dataset, rows):
        """The 'id' strings of the given rows (any order, repeats allowed) as a numpy object array.  An HF `datasets.Dataset` is
        read through its Arrow table: one `take` on the id column (through the indices mapping if the dataset carries one)
        and ONE conversion of the taken strings — `to_numpy(zero_copy_only=False)`, which builds the Python strings in C;
        `to_pylist()` builds a pyarrow Scalar per element first and took 4x as long as everything else together.  For the
        141 850 hits of 2 837 x 50 over 21 M documents: take 12 ms + strings 8 ms (+ 4 ms to nest them per query), where
        `Dataset.select(rows)['id']` took seconds; what remains is CPython creating 141 850 str objects."""
        import numpy as np
        if len(rows) == 0:
            return np.empty(0, dtype=object)
        table = getattr(doc_dataset, "data", None)
        if table is not None and hasattr(table, "column") and hasattr(doc_dataset, "_indices"):
            import pyarrow as pa
            take = pa.array(rows, type=pa.int64())
            if doc_dataset._indices is not None:
                take = doc_dataset._indices.column(0).take(take)
            got = table.column("id").take(take)
            if isinstance(got, pa.ChunkedArray):
                got = got.combine_chunks()
            return got.to_numpy(zero_copy_only=False)
        if hasattr(doc_dataset, "select"):
            return np.array(list(doc_dataset.select([int(r) for r in rows])['id']), dtype=object)
        col = doc_dataset['id']
        return np.array([col[int(r)] for r in rows], dtype=object)

    @staticmethod
    def _map_doc_ids(doc_dataset, indices):
        """Row indices -> doc-id strings for the Q*k hits only (reference materialises all N ids,
        retrieve.py:58,103).  -1 entries (an index with fewer than k rows) are dropped from their query's list."""
        import numpy as np
        idx = indices.numpy()
        nq, k = idx.shape
        flat = idx.reshape(-1)
        short = bool((flat < 0).any())
        ids = Retrieve._ids_at(doc_dataset, np.where(flat < 0, 0, flat) if short else flat)  # one take for all hits
        rows = ids.reshape(nq, k).tolist()
        if short:
            rows = [[v for v, r in zip(row, rr.tolist()) if r >= 0] for row, rr in zip(rows, idx)]
        return rows

    @torch.no_grad()
    def load_collection_and_retrieve(self, emb_q, doc_embeds, top_k_documents, detach_and_cpu=True,
                                     return_embeddings=False, dataset_size=None):
        """Same contract as retrieve.py:146-185 for an explicit list of chunk tensors."""
        num_emb = sum(int(c.shape[0]) for c in doc_embeds)
        if dataset_size is None:
            dataset_size = num_emb
        if num_emb != dataset_size:
            raise IOError(_INCOMPLETE.format(dataset_size - num_emb))
        dim = int(emb_q.shape[1])
        if len(doc_embeds) and doc_embeds[0].is_sparse:
            ix = self._build_resident(iter(doc_embeds), dataset_size, dim, "sparse", sparse=True)
        else:
            ix = self._build_resident((self._dense_chunk(c) for c in doc_embeds), dataset_size, dim, _metric_of(self.model))
        try:
            q = emb_q.to_dense() if emb_q.is_sparse else emb_q
            s, i = ix.search(q.detach().cpu().contiguous(), top_k_documents)
        finally:
            ix.close()
        final_top_k_scores = torch.from_numpy(s)
        final_top_k_indices = torch.from_numpy(i)
        if return_embeddings and len(doc_embeds) and doc_embeds[0].is_sparse:
            # the reference densifies every chunk (`emb_chunk.to_dense()`, retrieve.py:160-163) and gathers; here only the
            # retrieved rows of each sparse chunk are densified: [Q, k, vocab] like the reference's result
            flat = final_top_k_indices.clamp(min=0).reshape(-1)
            out = torch.zeros((flat.numel(), dim), dtype=doc_embeds[0].dtype)
            off = 0
            for c in doc_embeds:
                n_c = int(c.shape[0])
                sel = ((flat >= off) & (flat < off + n_c)).nonzero().reshape(-1)
                if sel.numel():
                    out[sel] = c.cpu().coalesce().index_select(0, flat[sel] - off).to_dense()
                off += n_c
            return final_top_k_scores, final_top_k_indices, out.reshape(tuple(final_top_k_indices.shape) + (dim,))
        if return_embeddings:
            all_rows = torch.cat([self._dense_chunk(c).cpu() for c in doc_embeds])
            return final_top_k_scores, final_top_k_indices, all_rows[final_top_k_indices.clamp(min=0)]
        return final_top_k_scores, final_top_k_indices, None

    # ------------------------------------------------------------------ misc (retrieve.py:190-197)
    def tokenize(self, example):
        return self.model.tokenize(example)

    @property
    def backend(self):
        """'hip' when the plug-in's encoder runs on the hand-written kernels, 'hf' when it stayed on torch."""
        from .dense import plugin_backends
        return "hip" if all(v == "hip" for v in plugin_backends(self.model).values()) else "hf"

    def get_clean_model_name(self):
        return self.model.model_name.replace('/', '_')

    def get_chunk_path(self, save_path, chunk):
        return f'{save_path}/embedding_chunk_{chunk}.pt'

    def close(self):
        for ix, _ in self._resident.values():
            ix.close()
        self._resident.clear()
        self._searchers.clear()  # (their gather buffers belong to the indexes just closed)
