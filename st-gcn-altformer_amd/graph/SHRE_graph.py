"""Drop-in for ``graph/SHRE_graph.py``: 22-joint single-hand graph (SHREC'17, DHG-14/28)."""
from stgcn_amd.graphs import SHREGraph as Graph, inward_bones

num_node = 22
self_link = [(i, i) for i in range(num_node)]
inward = inward_bones("SHRE")
outward = [(j, i) for (i, j) in inward]
neighbor = inward + outward
