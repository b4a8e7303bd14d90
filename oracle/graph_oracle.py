"""CPU oracle for the hand-skeleton adjacency tensors (numpy, float64).

TEST INFRASTRUCTURE ONLY (see oracle/stgcn_oracle.py header for who may import it).

Restates, in loop form so that it is independent of the vectorised product code
in ``st-gcn-altformer_amd/stgcn_amd/graphs.py``:

* ``edge_matrix``        <- ``graph/tools.py:5-9``   (edge (i,j) sets M[j,i] = 1)
* ``norm_columns``       <- ``graph/tools.py:12-20`` (A @ diag(1/colsum), 0-safe)
* ``norm_symmetric``     <- ``graph/tools.py:23-31`` (D^-1/2 A D^-1/2)
* ``labeling``           <- ``graph/tools.py:34-69`` + dispatch in
  ``graph/SHRE_graph.py:33-52`` / ``graph/LMDHG_graph.py:63-83``
* bone lists             <- ``graph/SHRE_graph.py:4-10`` (22 joints, 21 bones) and
  ``graph/LMDHG_graph.py:4-40`` (46 joints, 50 bones); written here as joint
  chains, the edge (parent, child) runs along each chain.

Pinned by ``tests/golden/graphs.npz`` (made from the imported reference by
``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import numpy as np

# chains of joints; consecutive entries (p, c) of a chain form the inward edge (p, c)
_CHAINS = {
    # one hand, 22 joints: wrist(0) -> palm(1) -> five fingers
    "SHRE": (22, [
        [0, 2, 3, 4, 5],
        [0, 1, 6, 7, 8, 9],
        [1, 10, 11, 12, 13],
        [1, 14, 15, 16, 17],
        [1, 18, 19, 20, 21],
    ]),
    # two hands, 23 joints each (second hand = first + 23)
    "LMDHG": (46, [
        [0, 1, 2, 3, 4, 5, 6],
        [1, 3, 7, 8, 9, 10],
        [1, 19, 20, 21, 22],
        [2, 19],
        [7, 11, 12, 13, 14],
        [11, 15, 16, 17, 18],
        [15, 19],
    ]),
}


def bones(name: str):
    """(num_joints, inward edge list) for 'SHRE' or 'LMDHG'."""
    V, chains = _CHAINS[name]
    hands = [0] if name == "SHRE" else [0, 23]
    edges = []
    for off in hands:
        for ch in chains:
            for p, c in zip(ch[:-1], ch[1:]):
                edges.append((p + off, c + off))
    return V, edges


def edge_matrix(edges, V):
    M = np.zeros((V, V), dtype=np.float64)
    for i, j in edges:
        M[j, i] = 1.0
    return M


def norm_columns(M):
    V = M.shape[0]
    out = np.zeros_like(M)
    for col in range(V):
        s = M[:, col].sum()
        if s > 0:
            for row in range(V):
                out[row, col] = M[row, col] * (s ** -1)
    return out


def norm_symmetric(M):
    V = M.shape[0]
    d = np.zeros(V)
    for col in range(V):
        s = M[:, col].sum()
        d[col] = s ** -0.5 if s > 0 else 0.0
    out = np.zeros_like(M)
    for r in range(V):
        for c in range(V):
            out[r, c] = d[r] * M[r, c] * d[c]
    return out


def labeling(name: str, mode: str):
    V, inward = bones(name)
    selfl = [(i, i) for i in range(V)]
    outward = [(j, i) for i, j in inward]
    neighbor = inward + outward
    if mode == "uniform":
        return norm_columns(edge_matrix(neighbor + selfl, V))
    if mode == "distance*":
        return edge_matrix(selfl, V) - norm_columns(edge_matrix(neighbor, V))
    if mode == "distance":
        return np.stack([edge_matrix(selfl, V), norm_columns(edge_matrix(neighbor, V))])
    if mode == "spatial":
        return np.stack([edge_matrix(selfl, V),
                         norm_columns(edge_matrix(inward, V)),
                         norm_columns(edge_matrix(outward, V))])
    if mode == "DAD":
        return norm_symmetric(edge_matrix(neighbor + selfl, V))
    if mode == "DLD":
        return edge_matrix(selfl, V) - norm_symmetric(edge_matrix(neighbor, V))
    raise ValueError(mode)
