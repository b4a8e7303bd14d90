"""Drop-in for the reference's ``graph`` package: ``import_class('graph.SHRE')`` / ``'graph.LMDHG'``
resolve here (graph/__init__.py:1-2 of the reference)."""
from .SHRE_graph import Graph as SHRE
from .LMDHG_graph import Graph as LMDHG
