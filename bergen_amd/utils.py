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


def chunk_sort_key(path):
    """The reference orders chunk files by the integer formed from every digit of the path."""
    return int(''.join(filter(str.isdigit, path)))


def sorted_chunk_files(index_path):
    return sorted(glob.glob(f'{index_path}/*.pt'), key=chunk_sort_key)


def load_chunk(path):
    """torch.load of one embedding_chunk_*.pt (dense fp16 tensor, or sparse COO for SPLADE)."""
    return torch.load(path, map_location='cpu', weights_only=True)


def load_embeddings(index_path):
    """Same contract and error mapping as reference utils.py:48-64."""
    try:
        embeds = [load_chunk(f) for f in sorted_chunk_files(index_path)]
        if not embeds:  # newer torch raises ValueError (not RuntimeError) for an empty cat: keep the intent
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
        embeds = torch.concat(embeds)
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
    with open(fname, 'w') as fout:
        for i, q_id in enumerate(q_ids):
            for rank, (d_id, score) in enumerate(zip(d_ids[i], scores[i])):
                fout.write(f'{q_id}\tq0\t{d_id}\t{rank+1}\t{float(score)}\trun\n')


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
