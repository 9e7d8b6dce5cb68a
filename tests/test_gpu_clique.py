"""TEASER++'s inlier selection on caller-supplied graphs (lgicp_teaser_clique: core numbers by
peeling, then the greedy clique heuristic, both on the device) against the CPU oracle's restatement
of the same heuristic.  The graphs aim at the peeling kernel's corners: more vertices leaving in one
step than its hand-over list holds, chains that peel one vertex at a time, rows wider than one wave,
values that need every bit plane."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _symmetric(n, p, rng):
    a = np.triu(rng.random((n, n)) < p, 1)
    return a | a.T


def _plant(adj, members):
    adj[np.ix_(members, members)] = True
    adj[members, members] = False


def _graphs():
    rng = np.random.default_rng(2024)
    g = {}
    g["dense_700"] = _symmetric(700, 0.5, rng)
    a = np.zeros((6000, 6000), bool)                       # 5,000 isolated vertices leave at level 0
    a[5000:, 5000:] = _symmetric(1000, 0.05, rng)
    _plant(a, np.arange(5100, 5140))
    g["isolated_5000_plus_clique_40"] = a
    a = _symmetric(4500, 0.02, rng)
    _plant(a, rng.choice(4500, 60, replace=False))
    g["planted_60_in_4500"] = a
    a = np.zeros((3000, 3000), bool)                       # a path: 1,500 steps of two vertices
    i = np.arange(2999)
    a[i, i + 1] = a[i + 1, i] = True
    g["path_3000"] = a
    g["no_edges_100"] = np.zeros((100, 100), bool)
    a = np.zeros((2500, 2500), bool)                       # one level holds 2,400 vertices: a ring
    i = np.arange(2400)
    a[i, (i + 1) % 2400] = a[(i + 1) % 2400, i] = True
    _plant(a, np.arange(2400, 2500))
    g["ring_2400_plus_clique_100"] = a
    a = _symmetric(16500, 0.0012, rng)                     # five waves per row, 15 bit planes
    _plant(a, rng.choice(16500, 30, replace=False))
    g["sparse_16500"] = a
    return g


def test_device_clique_equals_oracle_on_supplied_graphs():
    from lamp_amd import capi
    c = capi.Context(init_method=capi.INIT_IDENTITY)
    try:
        for name, adj in _graphs().items():
            assert np.array_equal(adj, adj.T) and not adj.diagonal().any()
            want = np.sort(O.teaser_clique(adj.astype(np.uint8)))
            got = np.sort(c.teaser_clique(adj))
            assert np.array_equal(got, want), (name, len(got), len(want))
            if len(got) > 1:
                assert adj[np.ix_(got, got)].sum() == len(got) * (len(got) - 1), name
    finally:
        c.close()
