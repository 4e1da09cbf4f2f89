/*
 * trec_eval_oracle.c — CPU restatement of the two trec_eval measures BERGEN's ranking evaluation asks for.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product path (bergen_amd/) may call into this file (see oracle/__init__.py).
 *
 * What it restates (naver/bergen @ 2026-01-30): utils.eval_retrieval_kilt (utils.py:263-300) hands its max-passage run to
 *     pytrec_eval.RelevanceEvaluator(qrel, {'P_1', f'recall_{top_k}'}).evaluate(run)            utils.py:275,294
 * and averages the per-topic values over the topics that come back (utils.py:295-296).  pytrec_eval is third-party (unpinned in
 * the reference's requirements.txt, absent from /root/reference and from this image; release 0.5 vendors trec_eval 9.0.x), so the
 * PUBLISHED algorithm of trec_eval 9.0 is restated here, function by function:
 *   form_res_rels.c  te_form_res_rels / comp_sim_docno : a topic's retrieved documents are ranked by sim DESCENDING, ties by docno
 *                    DESCENDING (strcmp(ptr2->docno, ptr1->docno)); sim is a C float (the text-results struct stores `float sim`,
 *                    pytrec_eval narrows the Python float on the way in); a retrieved document without a judgment counts as not
 *                    relevant; relevant = judged with rel >= relevance_level (1); num_rel = judged documents with rel >= 1,
 *                    retrieved or not.
 *   m_P.c            P_k = (relevant among the first k ranks) / k — ranks past the end of a short list count as not relevant.
 *   m_recall.c       recall_k = (relevant among the first min(k, num_ret) ranks) / num_rel, and 0 when num_rel == 0.
 *   topics           only topics present in BOTH the run and the qrels are evaluated (pytrec_eval iterates the run's topics and
 *                    skips those without qrels); a topic with judgments but no relevant document is evaluated (both measures 0)
 *                    and COUNTS in the reference's mean.
 * Parity status: UNPINNED — no (run, qrels, value) triple of the reference pins these numbers offline (its shipped runs name
 * passage rows, its qrels name Wikipedia pages, the page map needs the HF hub; pytrec_eval itself cannot be installed here).  What
 * this file gives is an INDEPENDENT second implementation in another language that bergen_amd/evaluation.py is property-tested
 * against (tests/test_eval_oracle.py), incl. the duplicate-topics quirk of scripts/kilt_generate_qrels.py:38,58-62.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float sim;
    const char* docno;
} ret_t;

static int comp_sim_docno(const void* a, const void* b) {
    const ret_t* p1 = (const ret_t*)a;
    const ret_t* p2 = (const ret_t*)b;
    if (p1->sim > p2->sim) return -1;
    if (p1->sim < p2->sim) return 1;
    return strcmp(p2->docno, p1->docno);
}

/* One topic.  docnos / sims: the retrieved documents (unique docnos: the reference's run is a dict); judged / rels: the topic's
 * qrels.  Writes P_k at cut-off 1 and recall at cut-off k. */
void oracle_trec_eval_topic(const char** docnos, const double* sims, int n_ret, const char** judged, const int* rels, int n_judged,
                            int k, double* p_1, double* recall_k) {
    ret_t* r = (ret_t*)malloc((size_t)(n_ret > 0 ? n_ret : 1) * sizeof(ret_t));
    for (int i = 0; i < n_ret; ++i) {
        r[i].sim = (float)sims[i];
        r[i].docno = docnos[i];
    }
    qsort(r, (size_t)n_ret, sizeof(ret_t), comp_sim_docno);
    long num_rel = 0;
    for (int j = 0; j < n_judged; ++j)
        if (rels[j] >= 1) ++num_rel;
    long rel_at_1 = 0, rel_at_k = 0;
    for (int i = 0; i < n_ret && i < (k > 1 ? k : 1); ++i) {
        int rel = 0;
        for (int j = 0; j < n_judged; ++j)
            if (strcmp(judged[j], r[i].docno) == 0) {
                rel = rels[j] >= 1;
                break;
            }
        if (i < 1) rel_at_1 += rel;
        if (i < k) rel_at_k += rel;
    }
    *p_1 = (double)rel_at_1 / 1.0;
    *recall_k = num_rel ? (double)rel_at_k / (double)num_rel : 0.0;
    free(r);
}
