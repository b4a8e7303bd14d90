"""Hand-skeleton graphs and their adjacency tensors (host side, numpy float64).

Mirrors the reference's ``graph`` package: ``graph/tools.py:5-69`` (matrix builders),
``graph/SHRE_graph.py`` (22-joint single hand, SHREC'17 / DHG-14/28) and
``graph/LMDHG_graph.py`` (46 joints = two 23-joint hands).  Built once at model construction;
the only consumer on the hot path is ``unit_agcn`` which receives ``Graph('spatial').A`` cast to
fp32 (ST_GCN_AltFormer.py:33-35).

Skeletons are stored as kinematic chains; an inward bone (parent, child) follows each chain from
the wrist outwards, exactly the edge set of the reference's ``inward`` lists.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

Edge = Tuple[int, int]

_ONE_HAND_22 = (  # wrist 0, palm 1, then thumb..pinky (4 joints each)
    (0, 2, 3, 4, 5),
    (0, 1, 6, 7, 8, 9),
    (1, 10, 11, 12, 13),
    (1, 14, 15, 16, 17),
    (1, 18, 19, 20, 21),
)
_ONE_HAND_23 = (  # Leap-Motion style hand of LMDHG
    (0, 1, 2, 3, 4, 5, 6),
    (1, 3, 7, 8, 9, 10),
    (1, 19, 20, 21, 22),
    (2, 19),
    (7, 11, 12, 13, 14),
    (11, 15, 16, 17, 18),
    (15, 19),
)

SKELETONS: Dict[str, Tuple[int, Tuple[Tuple[int, ...], ...], Tuple[int, ...]]] = {
    # name: (joints, chains of one hand, joint offset of every hand)
    "SHRE": (22, _ONE_HAND_22, (0,)),
    "LMDHG": (46, _ONE_HAND_23, (0, 23)),
}


def inward_bones(name: str) -> List[Edge]:
    _, chains, offsets = SKELETONS[name]
    return [(p + o, c + o) for o in offsets for ch in chains for p, c in zip(ch[:-1], ch[1:])]


def edge2mat(link: Sequence[Edge], num_node: int) -> np.ndarray:
    """Dense 0/1 matrix with M[j, i] = 1 for every edge (i, j)  (graph/tools.py:5-9)."""
    M = np.zeros((num_node, num_node))
    if len(link):
        src, dst = np.asarray(link, dtype=np.int64).T
        M[dst, src] = 1
    return M


def _inv_degree(M: np.ndarray, power: float) -> np.ndarray:
    deg = M.sum(axis=0)
    out = np.zeros_like(deg)
    nz = deg > 0
    out[nz] = deg[nz] ** power
    return out


def normalize_digraph(A: np.ndarray) -> np.ndarray:
    """A D^-1 with D the column sums; empty columns stay zero  (graph/tools.py:12-20)."""
    return A @ np.diag(_inv_degree(A, -1.0))


def normalize_undigraph(A: np.ndarray) -> np.ndarray:
    """D^-1/2 A D^-1/2  (graph/tools.py:23-31)."""
    d = np.diag(_inv_degree(A, -0.5))
    return d @ A @ d


def get_uniform_graph(num_node, self_link, neighbor):
    return normalize_digraph(edge2mat(list(neighbor) + list(self_link), num_node))


def get_uniform_distance_graph(num_node, self_link, neighbor):
    return edge2mat(self_link, num_node) - normalize_digraph(edge2mat(neighbor, num_node))


def get_distance_graph(num_node, self_link, neighbor):
    return np.stack((edge2mat(self_link, num_node), normalize_digraph(edge2mat(neighbor, num_node))))


def get_spatial_graph(num_node, self_link, inward, outward):
    """(3,V,V): identity, normalised inward bones, normalised outward bones  (graph/tools.py:53-58)."""
    return np.stack((edge2mat(self_link, num_node),
                     normalize_digraph(edge2mat(inward, num_node)),
                     normalize_digraph(edge2mat(outward, num_node))))


def get_DAD_graph(num_node, self_link, neighbor):
    return normalize_undigraph(edge2mat(list(neighbor) + list(self_link), num_node))


def get_DLD_graph(num_node, self_link, neighbor):
    return edge2mat(self_link, num_node) - normalize_undigraph(edge2mat(neighbor, num_node))


class HandGraph:
    """``Graph(labeling_mode)`` of the reference (SHRE_graph.py:12-52 / LMDHG_graph.py:42-83)."""

    skeleton = "SHRE"

    def __init__(self, labeling_mode: str = "uniform"):
        self.num_node = SKELETONS[self.skeleton][0]
        self.self_link = [(i, i) for i in range(self.num_node)]
        self.inward = inward_bones(self.skeleton)
        self.outward = [(j, i) for (i, j) in self.inward]
        self.neighbor = self.inward + self.outward
        self.A = self.get_adjacency_matrix(labeling_mode)

    def get_adjacency_matrix(self, labeling_mode=None):
        if labeling_mode is None:
            return self.A
        n, sl, nb = self.num_node, self.self_link, self.neighbor
        builders = {
            "uniform": lambda: get_uniform_graph(n, sl, nb),
            "distance*": lambda: get_uniform_distance_graph(n, sl, nb),
            "distance": lambda: get_distance_graph(n, sl, nb),
            "spatial": lambda: get_spatial_graph(n, sl, self.inward, self.outward),
            "DAD": lambda: get_DAD_graph(n, sl, nb),
            "DLD": lambda: get_DLD_graph(n, sl, nb),
        }
        if labeling_mode not in builders:
            raise ValueError()
        return builders[labeling_mode]()


class SHREGraph(HandGraph):
    skeleton = "SHRE"


class LMDHGGraph(HandGraph):
    skeleton = "LMDHG"
