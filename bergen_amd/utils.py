"""
Index / run-file formats of the retrieval path, wire-compatible with the reference's utils.py.

Reference functions mirrored (naver/bergen utils.py):
  load_embeddings        utils.py:48-64     chunk files -> one tensor, error mapping
  write_trec / load_trec utils.py:220-224, 244-259
  get_index_path         utils.py:349-352
  get_ranking_filename   utils.py:358-363
  chunk ordering key     utils.py:51 == modules/retrieve.py:85 (int of ALL digits in the path)
"""
import glob
import os
from collections import defaultdict

import torch


def cpu_budget(cgroup_root="/sys/fs/cgroup"):
    """CPUs this PROCESS may actually keep busy: the scheduler affinity mask, cut down to the container's CPU quota (cgroup v2
    `cpu.max` = "<quota> <period>", v1 `cpu/cpu.cfs_quota_us` / `cpu.cfs_period_us`) — not `os.cpu_count()`, which is the host's.
    A GPU pod typically sees every logical CPU of its host (256 on the MI355X boxes) under a quota of a few of them (16 there):
    thread pools sized for the host (torch intra-op, OpenMP, the Rust tokenizer's rayon pool) then burn the quota of a 100 ms
    period in a few milliseconds of spinning and the kernel freezes the WHOLE container until the period ends — tens of
    milliseconds of host-side stall between two launches (round 4: 18 ms on a 100 ms search step, `nr_throttled` counting up in
    `cpu.stat`).  Host-side pools of this package and of bench.py are sized with this number."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open(os.path.join(cgroup_root, "cpu.max")).read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")).read())
            period = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")).read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def fit_host_pools_to_cpu_budget():
    """Size the host-side thread pools this process has not sized yet to cpu_budget(): torch's intra-op pool (set directly), and —
    through the environment, for libraries that read it when they start their pool — OpenMP (`OMP_NUM_THREADS`) and the Rust
    tokenizer's rayon pool (`RAYON_NUM_THREADS`).  Values the user already exported are left alone.  Returns the budget.
    Called by bench.py at start-up and by Retrieve before its tokenizer threads start; see cpu_budget for why."""
    n = cpu_budget()
    if n < (os.cpu_count() or n):
        if "OMP_NUM_THREADS" not in os.environ and torch.get_num_threads() > n:
            torch.set_num_threads(n)
        os.environ.setdefault("OMP_NUM_THREADS", str(n))
        os.environ.setdefault("RAYON_NUM_THREADS", str(n))
    return n


def chunk_sort_key(path):
    """The reference orders chunk files by the integer formed from every digit of the path."""
    return int(''.join(filter(str.isdigit, path)))


def sorted_chunk_files(index_path):
    return sorted(glob.glob(f'{index_path}/*.pt'), key=chunk_sort_key)


def load_chunk(path, mmap=False):
    """torch.load of one embedding_chunk_*.pt (dense fp16 tensor, or sparse COO for SPLADE).  mmap=True maps the file
    instead of reading it into a fresh allocation (the load path copies the rows out of the mapping straight into its pinned
    staging buffers); files written by an old torch (no zip container) fall back to a plain read."""
    if mmap:
        try:
            return torch.load(path, map_location='cpu', weights_only=True, mmap=True)
        except (RuntimeError, ValueError):
            pass
    return torch.load(path, map_location='cpu', weights_only=True)


def prefetched(items, load, depth=1):
    """Yield load(item) for every item, in order, with up to `depth` loads running ahead on a worker thread: reading /
    deserialising chunk i + 1 overlaps the upload of chunk i (torch.load and the upload both release the GIL)."""
    import queue
    import threading
    items = list(items)
    q = queue.Queue(maxsize=max(1, depth))
    stop = threading.Event()

    def work():
        try:
            for it in items:
                if stop.is_set():
                    return
                q.put(("ok", load(it)))
        except BaseException as exc:  # delivered to the consumer
            q.put(("err", exc))

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        for _ in items:
            kind, val = q.get()
            if kind == "err":
                raise val
            yield val
    finally:
        stop.set()
        while t.is_alive():  # unblock a producer waiting on a full queue
            try:
                q.get_nowait()
            except queue.Empty:
                t.join(0.01)


def load_embeddings(index_path):
    """Same contract and error mapping as reference utils.py:48-64."""
    try:
        embeds = [load_chunk(f) for f in sorted_chunk_files(index_path)]
        if not embeds:  # newer torch raises ValueError (not RuntimeError) for an empty cat: keep the intent
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
        embeds = embeds[0] if len(embeds) == 1 else torch.concat(embeds)  # (a query folder is one chunk: no copy)
    except RuntimeError:
        # torch.cat(): expected a non-empty list of Tensors --> embeddings were not found
        raise RuntimeError("No embeddings found. Check .trec run file name if you are running oracle provenance.")
    except Exception as e:
        print("Exception occured: ", e)
        raise IOError(f'Embedding index corrupt. Please delete folder "{index_path}" and run again.')
    return embeds


def write_trec(fname, q_ids, d_ids, scores):
    """6 tab-separated fields, literal q0 / run, 1-based rank (reference utils.py:220-224).

    The reference formats a 0-dim fp32 tensor, i.e. the shortest repr of the fp32 value widened
    to a Python float; ``float(score)`` reproduces that for tensors, numpy scalars and floats.
    """
    rows = scores.tolist() if hasattr(scores, 'tolist') else scores  # (one conversion: element-wise float(tensor[i][j]) is 5x slower)
    with open(fname, 'w') as fout:
        for i, q_id in enumerate(q_ids):
            row = rows[i].tolist() if hasattr(rows[i], 'tolist') else rows[i]
            fout.write(''.join(f'{q_id}\tq0\t{d_id}\t{rank}\t{float(score)}\trun\n'
                               for rank, (d_id, score) in enumerate(zip(d_ids[i], row), 1)))


def load_trec(fname):
    """Reference utils.py:244-259: -> (q_ids, d_ids per query, scores per query), file order."""
    trec_dict = defaultdict(list)
    with open(fname) as fin:
        for l in fin:
            q_id, _, d_id, _, score, _ = l.split('\t')
            trec_dict[q_id].append((d_id, score))
    q_ids, d_ids, scores = list(), list(), list()
    for q_id in trec_dict:
        q_ids.append(q_id)
        d_ids.append([d for d, _ in trec_dict[q_id]])
        scores.append([float(s) for _, s in trec_dict[q_id]])
    return q_ids, d_ids, scores


def get_index_path(index_folder, dataset_name, model_name, query_or_doc, dataset_split='', query_generator_name='copy'):
    """Reference utils.py:349-352."""
    dataset_split = dataset_split + '_' if dataset_split != '' else ''
    query_gen_add = "" if query_generator_name == "copy" or query_or_doc == "doc" else f".{query_generator_name}"
    return os.path.join(index_folder, f'{dataset_name}_{dataset_split}{query_or_doc}_{model_name}{query_gen_add}')


def get_ranking_filename(runs_folder, query_dataset, doc_dataset, retriever_name, dataset_split, retrieve_top_k,
                         query_generator_name):
    """Reference utils.py:358-363 (the oracle_provenance branch is out of scope)."""
    query_gen_add = "" if query_generator_name == "copy" else f".{query_generator_name}"
    return (f'{runs_folder}/run.retrieve.top_{retrieve_top_k}.{query_dataset}.{doc_dataset}.{dataset_split}.'
            f'{retriever_name}{query_gen_add}.trec')


def merge_indexes(in_paths, out_path):
    """Merge several index folders into one by symlinking their chunk files under renumbered names — the
    reference's scripts/multilingual/merge_indexes.py:37-44: chunks of each input keep their order (integer of
    the digits in the file name), chunk i of an input becomes ``embedding_chunk_{base + i}.pt`` and the next
    input starts at the last new index + 1.  Returns the list of (link, target) pairs created."""
    if os.path.exists(out_path) and len(os.listdir(out_path)) > 0:
        raise FileExistsError(f"Folder {out_path} already exists and is not empty, exiting")
    for in_path in in_paths:
        if not os.path.exists(in_path) or len(os.listdir(in_path)) == 0:
            raise FileNotFoundError(f"All indexes for merging should be precomputed. Index {in_path} does not exist")
    os.makedirs(out_path, exist_ok=True)
    made = []
    current_global_index = 0
    for in_path in in_paths:
        newidx = current_global_index
        for chunk in sorted(os.listdir(in_path), key=lambda x: int(''.join(filter(str.isdigit, x)))):
            idx = int(''.join(filter(str.isdigit, chunk)))
            newidx = current_global_index + idx
            link = os.path.join(out_path, f"embedding_chunk_{newidx}.pt")
            os.symlink(os.path.abspath(os.path.join(in_path, chunk)), link)
            made.append((link, os.path.join(in_path, chunk)))
        current_global_index = newidx + 1
    return made


# ---- doc-id mapping + context fetch ----------------------------------------------------------
class IdIndex:
    """Compact id -> row table for a corpus' string ids.

    The reference keeps ``dataset.id2index`` as a pickled Python dict (modules/dataset_processor.py:88-100): 24.85 M
    str keys for KILT-100w, several GB of objects, and `retrieve.py:58` additionally materialises the id column as a
    Python list.  Here the ids live in ONE fixed-width numpy bytes array, sorted once; lookups are vectorised binary
    searches.  Dict-like enough (`in`, `[]`, `keys()`, `len`) to be dropped in as ``dataset.id2index``."""

    def __init__(self, ids):
        import numpy as np
        arr = np.asarray([i.encode() if isinstance(i, str) else i for i in ids], dtype=np.bytes_)
        self._order = np.argsort(arr, kind="stable")
        self._sorted = arr[self._order]
        if len(arr) > 1 and bool((self._sorted[1:] == self._sorted[:-1]).any()):
            # a dict keeps the LAST row of a duplicated id (dataset_processor.get_index_to_id)
            keep = np.ones(len(arr), bool)
            keep[:-1] = self._sorted[1:] != self._sorted[:-1]
            self._sorted, self._order = self._sorted[keep], self._order[keep]
        self._np = np

    @classmethod
    def from_ids(cls, ids):
        return cls(ids)

    def __len__(self):
        return len(self._sorted)

    def get_many(self, ids):
        """-> (rows int64 [len(ids)], found bool [len(ids)]); rows of missing ids are -1."""
        np = self._np
        if len(ids) == 0 or len(self._sorted) == 0:
            return np.full(len(ids), -1, np.int64), np.zeros(len(ids), bool)
        q = np.asarray([i.encode() if isinstance(i, str) else i for i in ids], dtype=np.bytes_)
        pos = np.searchsorted(self._sorted, q)
        pos_c = np.minimum(pos, len(self._sorted) - 1)
        found = self._sorted[pos_c] == q
        rows = np.where(found, self._order[pos_c], -1).astype(np.int64)
        return rows, found

    def __contains__(self, id_):
        return bool(self.get_many([id_])[1][0])

    def __getitem__(self, id_):
        rows, found = self.get_many([id_])
        if not found[0]:
            raise KeyError(id_)
        return int(rows[0])

    def keys(self):
        return (k.decode() for k in self._sorted)


def get_by_id(dataset, ids, field=None):
    """Reference utils.py:37-45: rows (or a field of the rows) of the ids that exist, in the order given.
    `dataset.id2index` may be the reference's dict or an IdIndex."""
    if not isinstance(ids, list):
        ids = [ids]
    table = dataset.id2index
    if isinstance(table, IdIndex):
        rows, found = table.get_many(ids)
        idxs = [int(r) for r in rows[found]]
    else:
        idxs = [table[id_] for id_ in ids if id_ in table]
    if field is not None:
        sel = dataset[idxs]
        return sel[field] if field in sel else []
    return idxs


def _rows_of_ids(table, ids):
    """Rows of the ids that exist in `table` (IdIndex or the reference's dict), in the order given."""
    if isinstance(table, IdIndex):
        rows, found = table.get_many(ids)
        return [int(r) for r in rows[found]]
    return [table[i] for i in ids if i in table]


def prepare_dataset_from_ids(dataset, q_ids, d_ids, multi_doc=False, query_field="content", oracle_provenance=False):
    """Join (query ids, ranked doc ids) back to text — the dataset the rerank and generation stages consume.

    Behaviour of the reference's utils.prepare_dataset_from_ids (utils.py:116-178) — same rows in the same order, same
    column names (its `ranking_labels` spelling on the joined rows included), same refusal of non-string ids — built
    column-wise: every hit of every query goes through ONE id lookup (`IdIndex.get_many`, a vectorised binary search)
    and ONE batched read of the passage texts, instead of two lookups and one dataset read per query."""
    import datasets
    queries = dataset['query']
    if q_ids is None and d_ids is None:  # no ranking: the queries with whatever labels the split carries
        cols = {'query': queries[query_field], 'q_id': queries['id']}
        for name in ('label', 'ranking_label'):
            if name in queries.features:
                cols[name] = queries[name]
        return datasets.Dataset.from_dict(cols)

    from_query_rows = bool(oracle_provenance) and "doc" in queries.features  # oracle passages travel with the query
    if not from_query_rows:
        assert isinstance(d_ids[0][0], str), f"{d_ids[0]}"
        assert isinstance(next(iter(dataset['doc'].id2index.keys())), str), \
            "Dataset id type is not string, real index retrieval will fail and retrieve nothing. Please convert to string in dataset_processor!"

    q_text = get_by_id(queries, q_ids, query_field)
    q_label = get_by_id(queries, q_ids, 'label')
    q_ranking = get_by_id(queries, q_ids, 'ranking_label')

    # ---- passages of every query: ids as ranked, texts and corpus rows of the ids the corpus knows
    if from_query_rows:
        hit_ids = [get_by_id(queries, q, 'doc_id')[0] for q in q_ids]
        hit_text = [get_by_id(queries, q, 'doc')[0] for q in q_ids]
        hit_rows = [[None] * len(ids) for ids in hit_ids]
    else:
        corpus = dataset['doc']
        hit_ids = [list(ids) for ids in d_ids[:len(q_ids)]]
        hit_rows = [_rows_of_ids(corpus.id2index, ids) for ids in hit_ids]
        flat = [r for rows in hit_rows for r in rows]
        texts = corpus[flat]['content'] if flat else []
        hit_text, at = [], 0
        for rows in hit_rows:
            hit_text.append(texts[at:at + len(rows)])
            at += len(rows)

    cols = defaultdict(list)
    if multi_doc:  # one row per query, its passages as lists
        cols['doc'], cols['query'], cols['q_id'] = hit_text, [q_text[i] for i in range(len(q_ids))], list(q_ids)
        cols['d_id'], cols['d_idx'] = hit_ids, hit_rows
        if len(q_label) > 0:
            cols['label'] = [q_label[i] for i in range(len(q_ids))]
        if len(q_ranking) > 0:
            cols['ranking_labels'] = [q_ranking[i] for i in range(len(q_ids))]
    else:  # one row per (query, passage)
        for i, q_id in enumerate(q_ids):
            n = min(len(hit_ids[i]), len(hit_text[i]), len(hit_rows[i]))
            cols['d_id'] += hit_ids[i][:n]
            cols['d_idx'] += hit_rows[i][:n]
            cols['doc'] += hit_text[i][:n]
            cols['query'] += [q_text[i]] * n
            cols['q_id'] += [q_id] * n
            if len(q_label) > 0:
                cols['label'] += [q_label[i]] * n
            if len(q_ranking) > 0:
                cols['ranking_labels'] += [q_ranking[i]] * n
        if not cols:
            cols = {'d_id': [], 'd_idx': [], 'doc': [], 'query': [], 'q_id': []}
    return datasets.Dataset.from_dict(dict(cols))
