"""Drop-in for ``graph/tools.py`` of the reference; implementation in stgcn_amd/graphs.py."""
from stgcn_amd.graphs import (edge2mat, get_DAD_graph, get_DLD_graph, get_distance_graph,  # noqa: F401
                              get_spatial_graph, get_uniform_distance_graph, get_uniform_graph,
                              normalize_digraph, normalize_undigraph)
