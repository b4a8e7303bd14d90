"""Drop-in for ``model/unit_agcn.py`` of the reference: ``from model.unit_agcn import unit_agcn``."""
from stgcn_amd.modules import unit_agcn  # noqa: F401
