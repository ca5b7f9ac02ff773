"""Baseline query strategies against vectors produced by the reference's own code (tests/golden/make_goldens.py ->
ref_strategies.npz): Coreset distance / k-centre greedy, softmax-entropy frame value, top-N selection rule."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), 'golden', 'ref_strategies.npz')


def test_coreset_distances_and_furthest_first_match_reference():
    from pcdet.query_strategies.coreset_sampling import pairwise_squared_distances, furthest_first
    g = np.load(G)
    X, S = torch.from_numpy(g['cs_X']), torch.from_numpy(g['cs_S'])
    np.testing.assert_allclose(pairwise_squared_distances(X, S).numpy(), g['cs_dist'], rtol=1e-6, atol=1e-6)
    assert furthest_first(X, S, 9) == g['cs_pick'].tolist()          # index-exact incl. the duplicated row
    assert furthest_first(X, S, 0) == []


def test_entropy_value_and_selection_rule_match_reference():
    from pcdet.query_strategies.pool_eval import softmax_entropy, PoolEvalStrategy
    g = np.load(G)
    parts = np.split(g['ent_logits'], np.cumsum(g['ent_counts'])[:-1])
    vals = torch.stack([softmax_entropy(torch.from_numpy(p)) for p in parts])
    np.testing.assert_allclose(vals.numpy(), g['ent_vals'], rtol=1e-6, atol=1e-7)
    s = PoolEvalStrategy.__new__(PoolEvalStrategy)
    s.pairs = [(i, None) for i in range(len(parts))]
    assert s.top_n_ascending(vals, 3) == g['ent_selected'].tolist()


def test_factory_lists_reference_strategies():
    from pcdet import query_strategies as q
    assert {'random', 'entropy', 'badge', 'coreset', 'montecarlo', 'confidence', 'crb'} <= set(q.names())
