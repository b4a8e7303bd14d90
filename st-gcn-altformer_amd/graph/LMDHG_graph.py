"""Drop-in for ``graph/LMDHG_graph.py``: 46-joint two-hand graph (LMDHG)."""
from stgcn_amd.graphs import LMDHGGraph as Graph, inward_bones

num_node = 46
self_link = [(i, i) for i in range(num_node)]
inward = inward_bones("LMDHG")
outward = [(j, i) for (i, j) in inward]
neighbor = inward + outward
