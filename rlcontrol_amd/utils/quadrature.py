"""Clenshaw-Curtis nodes and weights on [-1, 1] for the KL agents' action integral.

The reference takes them from quadpy (``quadpy.c1.clenshaw_curtis(N_param)``, reversekl_network.py:64-72,
forwardkl_network.py:60-70), which is not installed here; the rule itself is classical.  With n = N - 1 intervals:
    x_k = -cos(k pi / n),  k = 0..n
    w_k = (c_k / n) * (1 - sum_{j=1}^{floor(n/2)} b_j / (4 j^2 - 1) * cos(2 j k pi / n)),
    c_k = 1 at the two end points and 2 inside,  b_j = 1 if 2 j == n else 2.
The agents drop the two end points (tanh never reaches +-1) and scale the nodes by ``action_max``.
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
