"""Seeded synthetic QA-subgraph records in the reference's on-disk schema.

The reference ships no `*.graph.adj.pk` data and there is no network, so every workload in this
repo (parity fixtures, tests, bench) starts from records produced here.  One record is exactly the
dict that the reference's preprocessing emits per (question, answer-choice) pair
(reference utils/graph.py:338): ``{'adj', 'concepts', 'qmask', 'amask', 'cid2score'}`` with

* ``adj``      scipy COO bool matrix of shape (n_rel * m, m); entry (rel * m + src, tgt) is a KG edge,
* ``concepts`` int array [m] of unique concept ids, ordered Q concepts, A concepts, then others,
* ``qmask`` / ``amask`` bool [m] prefix masks,
* ``cid2score`` dict concept id -> LM relevance score, key -1 is the context node; or None (MedQA).

Shapes follow SURVEY.md section 8(d): CSQA-like graphs have <= 200 node slots and 400..2000 directed
edges after the loader mirrors them; the config-1 shape is 99 concepts, 8 Q + 2 A, 390 KG edges.
"""
from collections import OrderedDict

import numpy as np
from scipy.sparse import coo_matrix


def make_record(rng, n_concepts, n_q, n_a, n_kg_edges, n_rel=17, n_concept_vocab=100000,
                zipf=False, with_scores=True):
    """One synthetic record; `n_kg_edges` unique (rel, src, tgt) triples with src != tgt."""
    m = int(n_concepts)
    assert m >= n_q + n_a and n_q >= 1 and n_a >= 0
    concepts = rng.choice(n_concept_vocab - 1, size=m, replace=False).astype(np.int32)
    ar = np.arange(m)
    qmask = ar < n_q
    amask = (ar >= n_q) & (ar < n_q + n_a)
    if zipf:
        w = 1.0 / np.arange(1, m + 1)
        p = w / w.sum()
    else:
        p = None
    triples = set()
    max_unique = n_rel * m * (m - 1)
    n_kg_edges = min(int(n_kg_edges), max_unique // 2)
    while len(triples) < n_kg_edges:
        need = n_kg_edges - len(triples)
        src = rng.choice(m, size=2 * need + 8, p=p)
        tgt = rng.choice(m, size=2 * need + 8, p=p)
        rel = rng.integers(0, n_rel, size=2 * need + 8)
        for r, s, t in zip(rel, src, tgt):
            if s != t:
                triples.add((int(r), int(s), int(t)))
                if len(triples) == n_kg_edges:
                    break
    tri = np.array(sorted(triples), dtype=np.int64).reshape(-1, 3)
    row = tri[:, 0] * m + tri[:, 1]
    col = tri[:, 2]
    adj = coo_matrix((np.ones(len(row), dtype=bool), (row, col)), shape=(n_rel * m, m))
    if with_scores:
        # LM scores are negative MLM losses; the extra (non Q/A) concepts are stored from high to low score
        # ... stored as multiples of 1/64: the reference's score normalisation (modeling_qagnn.py:160-167) divides by a row sum
        # of |score|, and sin(1.1^j * score) with 1.1^j up to 1.2e4 amplifies a 1-ulp difference of that sum (summation order:
        # CPU vs GPU, fp32 vs fp64) into 1e-3-level feature changes.  Sums of <= 200 such values are exact in fp32 in ANY
        # order, so every implementation sees bit-identical normalised scores and parity can be stated tightly.
        sc = -np.round((20.0 + 40.0 * rng.random(m)) * 64.0) / 64.0
        sc[n_q + n_a:] = np.sort(sc[n_q + n_a:])[::-1]
        cid2score = OrderedDict()
        cid2score[-1] = float(max(sc.max(), -20.0) + 1.0)  # context node = the highest score
        for c, s in zip(concepts, sc):
            cid2score[int(c)] = float(s)
    else:
        cid2score = None
    return {'adj': adj, 'concepts': concepts, 'qmask': qmask, 'amask': amask, 'cid2score': cid2score}


def make_records(n_samples, seed=0, shape='csqa', n_rel=17, n_concept_vocab=100000, zipf=False):
    """A list of `n_samples` records.

    shape:
      'config1'  SURVEY 8(d) config 1: 99 concepts (100 node slots, all real), 8 Q + 2 A, 390 KG edges
      'csqa'     40..199 concepts, 200..1000 KG edges (loader doubles them: 400..2000 directed edges)
      'csqa_max' 199 concepts, 990 KG edges (the <=200 nodes / <=2k edges worst case of north_star)
      'medqa'    100..199 concepts, ~1500 KG edges, no node scores (cid2score None), 15 relations
      'tiny'     3..12 concepts, 0..15 KG edges (edge cases: empty graphs, 1 concept, ...)
      'hub'      249 concepts of which 60 Q + 25 A (the context node gets 85 out-edges and 85 in-edges: a > 64-degree segment),
                 2 900 Zipf-distributed KG edges = ~5.8 k directed edges with hub concepts -- the loader's comment case of a
                 249-node graph (reference utils/data_utils.py:103) that it truncates to n = 200 node slots (:117)
    """
    rng = np.random.default_rng(seed)
    recs = []
    for _ in range(n_samples):
        if shape == 'config1':
            recs.append(make_record(rng, 99, 8, 2, 390, n_rel, n_concept_vocab, zipf))
        elif shape == 'csqa':
            m = int(rng.integers(40, 200))
            nq = int(rng.integers(2, 12))
            na = int(rng.integers(1, 4))
            e = int(rng.integers(200, 1001))
            recs.append(make_record(rng, m, nq, na, e, n_rel, n_concept_vocab, zipf))
        elif shape == 'csqa_max':
            recs.append(make_record(rng, 199, 8, 2, 990, n_rel, n_concept_vocab, zipf))
        elif shape == 'medqa':
            m = int(rng.integers(100, 200))
            recs.append(make_record(rng, m, int(rng.integers(3, 20)), int(rng.integers(1, 4)),
                                    1500, 15, n_concept_vocab, zipf, with_scores=False))
        elif shape == 'hub':
            recs.append(make_record(rng, 249, 60, 25, 2900, n_rel, n_concept_vocab, zipf=True))
        elif shape == 'tiny':
            m = int(rng.integers(1, 13))
            nq = int(rng.integers(1, max(2, m // 2 + 1)))
            na = int(rng.integers(0, max(1, min(3, m - nq + 1))))
            e = int(rng.integers(0, 16)) if m > 1 else 0
            recs.append(make_record(rng, m, nq, na, e, n_rel, n_concept_vocab, zipf))
        else:
            raise ValueError(f'unknown shape {shape!r}')
    return recs
