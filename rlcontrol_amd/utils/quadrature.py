"""Clenshaw-Curtis nodes and weights on [-1, 1] for the KL agents' action integral.

The reference takes them from quadpy (``quadpy.c1.clenshaw_curtis(N_param)``, reversekl_network.py:64-72,
forwardkl_network.py:60-70), which is not installed here; the rule itself is classical.  With n = N - 1 intervals:
    x_k = -cos(k pi / n),  k = 0..n
    w_k = (c_k / n) * (1 - sum_{j=1}^{floor(n/2)} b_j / (4 j^2 - 1) * cos(2 j k pi / n)),
    c_k = 1 at the two end points and 2 inside,  b_j = 1 if 2 j == n else 2.
The agents drop the two end points (tanh never reaches +-1) and scale the nodes by ``action_max``.

Parity unpinned against quadpy itself: the library is absent and the reference holds no fixture for the rule, so the
point ORDER (ascending here, as the reference's own comments imply) and the last bits of the fp64 values before the fp32
cast are checked only against a second in-repo construction (Waldvogel's FFT, oracle/kl_torch.py) and polynomial
exactness (tests/test_kl.py).
"""
import numpy as np


def clenshaw_curtis(n_points):
    """(points, weights) float64, ascending points; integrates polynomials up to degree n_points - 1 exactly."""
    n_points = int(n_points)
    if n_points < 2:
        raise ValueError("Clenshaw-Curtis needs at least 2 points")
    n = n_points - 1
    k = np.arange(n_points, dtype=np.float64)
    points = -np.cos(np.pi * k / n)
    j = np.arange(1, n // 2 + 1, dtype=np.float64)
    b = np.where(2 * j == n, 1.0, 2.0)
    series = (b / (4.0 * j * j - 1.0))[None, :] * np.cos(2.0 * np.pi * np.outer(k, j) / n)
    c = np.where((k == 0) | (k == n), 1.0, 2.0)
    weights = c / n * (1.0 - series.sum(axis=1))
    return points, weights


def interior_action_nodes(n_points, action_max):
    """Nodes and weights as the KL networks hold them for a 1-D action: end points cut, nodes cast to fp32 and
    scaled by action_max (a float64 numpy scalar in the reference, so the product is formed in float64 and cast
    back: reversekl_network.py:69-71), weights cast to fp32."""
    x, w = clenshaw_curtis(n_points)
    actions = (x[1:-1].astype(np.float32).astype(np.float64) * float(action_max)).astype(np.float32)
    return actions, w[1:-1].astype(np.float32)


def sparse_grid_action_nodes(l_param, action_dim, action_max, interior=True):
    """The action_dim > 1 rule of the KL networks (reversekl_network.py:78-108, forwardkl_network.py likewise): Smolyak's
    combination of nested Clenshaw-Curtis line rules.  Level 0 is the midpoint rule (point 0, weight 2), level i >= 1 the
    Clenshaw-Curtis rule on 2^i + 1 points with its two end points cut; every multi-index k with
    l - action_dim <= |k| <= l - 1 contributes the tensor product of its levels' rules with the coefficient
    (-1)^(l - |k| + 1) * binom(action_dim - 1, |k| + action_dim - l).  Points repeat across multi-indices and are kept
    as separate nodes, as the reference keeps them; weights are of either sign.
    Returns (actions [K, action_dim] fp32 scaled by action_max per dimension, weights [K] fp32).
    interior=False keeps the end points (the complete Smolyak rule, for the exactness test only)."""
    import itertools
    from math import comb
    l, A = int(l_param), int(action_dim)
    if A < 2:
        raise ValueError("the sparse grid is the action_dim > 1 branch; use interior_action_nodes for one dimension")
    if l < A:
        raise ValueError("l_param %d < action_dim %d leaves no multi-index" % (l, A))
    cut = slice(1, -1) if interior else slice(None)
    points, weights = [np.array([0.0])], [np.array([2.0])]
    for i in range(1, l):
        x, w = clenshaw_curtis(2 ** i + 1)
        points.append(x[cut])
        weights.append(w[cut])
    amax = np.broadcast_to(np.asarray(action_max, np.float64).reshape(-1), (A,))
    acts, wts = [], []
    for k in itertools.product(range(l), repeat=A):
        sk = sum(k)
        if sk + A < l or sk + A > l + A - 1:
            continue
        coeff = (-1.0) ** (l - sk + 1) * comb(A - 1, sk + A - l)
        for j in itertools.product(*[range(len(points[ki])) for ki in k]):
            # torch.tensor([...], float32) first, the product with action_max afterwards (reversekl_network.py:101-106)
            acts.append([np.float32(points[k[i]][j[i]]) for i in range(A)])
            wts.append(coeff * np.prod([weights[k[i]][j[i]] for i in range(A)]))
    actions = (np.asarray(acts, np.float32).astype(np.float64) * amax[None, :]).astype(np.float32)
    return actions, np.asarray(wts, np.float64).astype(np.float32)
