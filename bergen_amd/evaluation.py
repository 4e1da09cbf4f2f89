"""
Ranking evaluation of a retrieval run — the step immediately after the hot path.

Mirrors the reference's ``utils.eval_retrieval_kilt`` (utils.py:263-300): passages are mapped to their wiki page
ids by the caller, duplicate page ids keep the maximum passage score (maxP, utils.py:279-285), the deduplicated
run is optionally written as ``eval_<split>_[re]ranking_run.trec`` (utils.py:287-293, literal ``QO`` column) and
P@1 / recall@k are averaged over the queries present in BOTH the run and the qrels (utils.py:295-297) into
``eval_<split>_[re]ranking_metrics.json`` (utils.py:299-300).

The reference computes the two measures with ``pytrec_eval`` (third-party, unpinned, not installed in this
image); they are restated here with trec_eval's definitions:
  * documents are ranked by score descending, ties by document id descending (trec_eval's comp_sim_docno); the score is compared
    as a C float (trec_eval stores `float sim`; pytrec_eval narrows the Python float on the way in), so two scores that differ
    only beyond fp32 tie;
  * relevant = judged with relevance >= 1;
  * P_1 = [top-ranked document is relevant];  recall_k = (relevant documents in the top k) / (relevant documents).
"""
import json
import os
from collections import defaultdict


def get_qrel_ranking_filename(qrels_folder, dataset_name, split, debug=False):
    """Reference utils.py:345-347."""
    dataset_name = dataset_name.replace('_debug', '') if debug else dataset_name
    return f'{qrels_folder}/qrel.{dataset_name}.{split}.json'


def max_passage_run(query_ids, doc_ids, scores):
    """{q_id: {page_id: best score}} keeping the maximum-scoring passage per page (utils.py:276-285)."""
    run = defaultdict(dict)
    for qi, q_id in enumerate(query_ids):
        for doc_id, score in zip(doc_ids[qi], scores[qi]):
            score = float(score)
            if doc_id not in run[q_id] or score >= run[q_id][doc_id]:
                run[q_id][doc_id] = score
    return run


def _as_c_float(x):
    import struct
    try:
        return struct.unpack("f", struct.pack("f", float(x)))[0]
    except OverflowError:  # beyond fp32's range: +-inf, as a C cast gives
        return float("inf") if x > 0 else float("-inf")


def ranking_metrics(run, qrel, top_k=5):
    """Mean P_1 and recall_{top_k} over the queries in both `run` and `qrel` (trec_eval definitions)."""
    p1_sum = rec_sum = 0.0
    n = 0
    for q_id, docs in run.items():
        judged = qrel.get(q_id)
        if judged is None:
            continue
        rel = {d for d, r in judged.items() if r >= 1}
        ranked = sorted(docs.items(), key=lambda kv: str(kv[0]).encode(), reverse=True)  # ties: document id descending (strcmp: bytes) ...
        ranked.sort(key=lambda kv: _as_c_float(kv[1]), reverse=True)                      # ... under a stable sort by the fp32 score
        top = [d for d, _ in ranked]
        p1_sum += 1.0 if top and top[0] in rel else 0.0
        rec_sum += (sum(1 for d in top[:top_k] if d in rel) / len(rel)) if rel else 0.0
        n += 1
    n = max(1, n)
    return {'P_1': p1_sum / n, f'recall_{top_k}': rec_sum / n}


def eval_retrieval_kilt(experiment_folder, qrels_folder, query_dataset_name, doc_dataset_name, split, query_ids,
                        doc_ids, scores, top_k=5, reranking=False, debug=False, write_trec=True):
    """Same signature, files and early-return rules as the reference (utils.py:263-300); returns the metrics dict
    (the reference returns None)."""
    scores = scores.tolist() if hasattr(scores, "tolist") else scores
    reranking_str = 're' if reranking else ''
    qrels_file = get_qrel_ranking_filename(qrels_folder, query_dataset_name, split, debug)
    if not os.path.exists(qrels_file):
        return None
    qrel = json.load(open(qrels_file))
    if "doc_dataset_name" in qrel:
        if qrel["doc_dataset_name"] != doc_dataset_name:
            return None
        qrel.pop("doc_dataset_name")
    run = max_passage_run(query_ids, doc_ids, scores)
    if write_trec:
        with open(f'{experiment_folder}/eval_{split}_{reranking_str}ranking_run.trec', 'w') as trec_out:
            for q_id, scores_dict in run.items():
                ordered = sorted(scores_dict.items(), key=lambda item: item[1], reverse=True)
                for i, (doc_id, score) in enumerate(ordered):
                    trec_out.write(f'{q_id}\tQO\t{doc_id}\t{i+1}\t{score}\trun\n')
    mean_metrics = ranking_metrics(run, qrel, top_k)
    with open(f"{experiment_folder}/eval_{split}_{reranking_str}ranking_metrics.json", 'w') as fp:  # utils.write_dict
        json.dump(mean_metrics, fp, indent=2)
    return mean_metrics
